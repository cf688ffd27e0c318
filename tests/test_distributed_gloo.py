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
    from auv_sim_amd import _lib, _prrt_lib, distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    G = D.TorchGather()
    lo, hi = D.shard_range(E_total, rank, world)
    n = hi - lo  # uneven when E_total % world != 0: rank 0 holds one episode more
    summ = np.zeros(n, dtype=_lib.SUMMARY_DTYPE)
    psumm = np.zeros(n, dtype=_prrt_lib.PRRT_SUMMARY_DTYPE)  # the Planner_RRT record goes through the same helpers
    for i in range(n):
        e = lo + i  # global episode id
        summ[i]["best_leaf"] = e
        summ[i]["best_path_len"] = 3 + (e % 5)
        summ[i]["best_cost"] = [-(e + 0.5), 0, 0, 0]
        psumm[i]["steps"] = 100 + e
        psumm[i]["done"] = e % 2
        psumm[i]["path_len"] = (2 + e % 3) if e % 2 else 0
        psumm[i]["arc"] = [e, 0, 0, 0, 0, e + 0.25]
    lens = torch.from_numpy(summ["best_path_len"].astype(np.int64))
    paths = torch.zeros((int(lens.sum()), 7), dtype=torch.float64)
    pos = 0
    for i in range(n):
        L = int(lens[i])
        paths[pos:pos + L, 0] = lo + i
        paths[pos:pos + L, 1] = torch.arange(L, dtype=torch.float64)
        pos += L
    plens = torch.from_numpy(psumm["path_len"].astype(np.int64))
    ppaths = torch.zeros((int(plens.sum()), 5), dtype=torch.float64)
    pos = 0
    for i in range(n):
        L = int(plens[i])
        ppaths[pos:pos + L, 0] = lo + i
        pos += L
    rec = G.gather_records(D.summaries_to_tensor(summ, "cpu"))
    all_len, all_paths = G.gather_paths(paths, lens)
    prec = G.gather_records(D.summaries_to_tensor(psumm, "cpu"))
    pall_len, pall_paths = G.gather_paths(ppaths, plens)
    ok = True
    seen = []
    for r in range(world):
        rlo, rhi = D.shard_range(E_total, r, world)
        rs = D.tensor_to_summaries(rec[r], _lib.SUMMARY_DTYPE)
        ps = D.tensor_to_summaries(prec[r], _prrt_lib.PRRT_SUMMARY_DTYPE)
        ok &= len(rs) == rhi - rlo and len(ps) == rhi - rlo and len(all_len[r]) == rhi - rlo
        pos = ppos = 0
        for i in range(rhi - rlo):
            e = rlo + i
            ok &= int(rs[i]["best_leaf"]) == e and float(rs[i]["best_cost"][0]) == -(e + 0.5)
            ok &= int(ps[i]["steps"]) == 100 + e and float(ps[i]["arc"][5]) == e + 0.25
            L = int(all_len[r][i])
            ok &= L == 3 + (e % 5)
            seg = all_paths[r][pos:pos + L]
            ok &= bool((seg[:, 0] == e).all()) and bool((seg[:, 1] == torch.arange(L, dtype=torch.float64)).all())
            pos += L
            PL = int(pall_len[r][i])
            ok &= PL == ((2 + e % 3) if e % 2 else 0)
            ok &= bool((pall_paths[r][ppos:ppos + PL, 0] == e).all())
            ppos += PL
            seen.append(e)
        ok &= pos == all_paths[r].shape[0] and ppos == pall_paths[r].shape[0]
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


import pytest


@pytest.mark.parametrize("E_total", [11, 12, 1])  # uneven split, even split, one rank with nothing
def test_two_rank_gather_gloo(E_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, E_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]
