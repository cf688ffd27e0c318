"""The N>1 path on CPU: world_size-2 gloo run of the shard + gather logic bench.py uses."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, E_total, q):
    import sys
    sys.path.insert(0, REPO)
    from auv_sim_amd import _lib, distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = D.shard_range(E_total, rank, world)
    n = hi - lo
    E_pad = -(-E_total // world)  # equal-size blocks for the gather
    summ = np.zeros(E_pad, dtype=_lib.SUMMARY_DTYPE)
    summ["status"] = -99
    rng = np.random.default_rng(100 + rank)
    for i in range(n):
        e = lo + i  # global episode id
        summ[i]["status"] = 0
        summ[i]["best_leaf"] = e
        summ[i]["best_path_len"] = 3 + (e % 5)
        summ[i]["best_cost"] = [-(e + 0.5), 0, 0, 0]
    lens = torch.from_numpy(np.where(summ["status"] == 0, summ["best_path_len"], 0).astype(np.int64))
    paths = torch.zeros((int(lens.sum()), 7), dtype=torch.float64)
    pos = 0
    for i in range(n):
        L = int(lens[i])
        paths[pos:pos + L, 0] = lo + i
        paths[pos:pos + L, 1] = torch.arange(L, dtype=torch.float64)
        pos += L
    rec = D.gather_records(D.summaries_to_tensor(summ, "cpu"))
    all_len, all_paths = D.gather_paths(paths, lens)
    allsumm = D.tensor_to_summaries(rec, _lib.SUMMARY_DTYPE)
    ok = True
    seen = []
    for r in range(world):
        rlo, rhi = D.shard_range(E_total, r, world)
        pos = 0
        for i in range(rhi - rlo):
            e = rlo + i
            s = allsumm[r, i]
            ok &= int(s["best_leaf"]) == e and float(s["best_cost"][0]) == -(e + 0.5)
            L = int(all_len[r, i])
            ok &= L == 3 + (e % 5)
            seg = all_paths[r][pos:pos + L]
            ok &= bool((seg[:, 0] == e).all()) and bool((seg[:, 1] == torch.arange(L, dtype=torch.float64)).all())
            pos += L
            seen.append(e)
        ok &= pos == all_paths[r].shape[0]
    ok &= seen == list(range(E_total))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_everything():
    from auv_sim_amd import distributed as D
    for n in (0, 1, 7, 512, 513):
        for w in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_two_rank_gather_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 11, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]
