"""The multi-rank code of the C-ABI gather (auv_sim_amd/csrc/gather_host.h: byte counts, prefix offsets, grouped broadcasts, the
grouped ncclSend / ncclRecv of the gather to a root, the gather stream and its events) with world sizes 2 and 4 ON ONE GPU.

RCCL refuses two ranks on one device and the pool has one GPU per box, so the real collectives have only ever run at world size
1 here.  This test binds a stand-in instead (tests/mock_rccl/mock_rccl.cpp through AUVP_RCCL_LIBRARY: the same entry points, bytes
moved between the processes over Unix sockets, every operation ordered behind the stream it is given) and runs everything of
the library ABOVE those entry points exactly as a multi-GPU job would: one process per rank, a communicator on the planner handle,
device pointers in and out.  What it cannot show is RCCL's own behaviour; tests/test_rccl_abi.py pins the signatures."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK_SRC = os.path.join(REPO, "tests", "mock_rccl", "mock_rccl.cpp")


@pytest.fixture(scope="module")
def mock_lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("mock_rccl") / "libmock_rccl.so")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O1", "-o", out, MOCK_SRC, "-lpthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return out


def _worker(rank, world, id_path, E_total, mock, q):
    import time
    os.environ["AUVP_RCCL_LIBRARY"] = mock       # read when the library binds RCCL (first use)
    sys.path.insert(0, REPO)
    import torch
    from auv_sim_amd import _lib, distributed as D, synth
    torch.cuda.set_device(0)                       # every rank on the one GPU
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0)
    assert D.RcclGather.library() == mock, D.RcclGather.library()

    def exchange(mine):  # rank 0 writes the 128-byte id to a file, the others wait for it
        if rank == 0:
            with open(id_path + ".tmp", "wb") as f:
                f.write(mine)
            os.replace(id_path + ".tmp", id_path)
            return mine
        for _ in range(1200):
            if os.path.exists(id_path):
                return open(id_path, "rb").read()
            time.sleep(0.05)
        return b""
    g = D.RcclGather(ctx, rank, world, exchange)
    ok = g.info() == (world, rank, world)
    lo, hi = D.shard_range(E_total, rank, world)
    n = hi - lo
    sizes = D.shard_sizes(E_total, world)
    rec = torch.zeros((n, 112), dtype=torch.uint8, device=dev)
    for i in range(n):
        rec[i] = (lo + i) % 251
    lens = torch.tensor([2 + ((lo + i) % 4) for i in range(n)], dtype=torch.int64, device=dev)
    paths = torch.zeros((int(lens.sum().item()) if n else 0, 7), dtype=torch.float64, device=dev)
    pos = 0
    for i in range(n):
        L = int(lens[i])
        paths[pos:pos + L, 0] = lo + i
        paths[pos:pos + L, 5] = torch.arange(L, dtype=torch.float64, device=dev)
        pos += L

    def check_all(recs, lns, pths):
        good = True
        for r in range(world):
            rlo, rhi = D.shard_range(E_total, r, world)
            good &= recs[r].shape == (rhi - rlo, 112) and len(lns[r]) == rhi - rlo
            p = 0
            for i in range(rhi - rlo):
                e = rlo + i
                L = int(lns[r][i])
                good &= bool((recs[r][i] == e % 251).all()) and L == 2 + (e % 4)
                seg = pths[r][p:p + L]
                good &= bool((seg[:, 0] == e).all()) and bool((seg[:, 5] == torch.arange(L, dtype=torch.float64, device=dev)).all())
                p += L
            good &= p == pths[r].shape[0]
        return bool(good)
    # ---- the all-gathers: every rank ends up with every record (variable-length: counts, then one grouped broadcast per rank)
    out = g.gather_records(rec)
    all_len, blocks = g.gather_paths(paths, lens)
    ok &= check_all(out, all_len, blocks)
    send = torch.full((16,), rank + 1, dtype=torch.uint8, device=dev)
    recv = torch.zeros(16 * world, dtype=torch.uint8, device=dev)
    ctx._chk(g.L.auvp_gather(ctx.h, C.c_void_p(send.data_ptr()), 16, C.c_void_p(recv.data_ptr())))
    ok &= all(bool((recv[16 * r:16 * r + 16] == r + 1).all()) for r in range(world))
    # auvp_gather_var in one call, with a buffer that is large enough on every rank / too small on one rank (collective decision)
    counts = (C.c_int64 * world)()
    nbytes = paths.numel() * 8
    total = sum(s_ * 0 for s_ in sizes)  # (sizes only fix the shape; the byte total comes from the count phase)
    ctx._chk(g.L.auvp_gather_var(ctx.h, C.c_void_p(paths.data_ptr()) if nbytes else None, nbytes, None, 0, counts))
    total = sum(counts)
    big = torch.zeros(max(total, 1), dtype=torch.uint8, device=dev)
    ctx._chk(g.L.auvp_gather_var(ctx.h, C.c_void_p(paths.data_ptr()) if nbytes else None, nbytes, C.c_void_p(big.data_ptr()), total, counts))
    ok &= bytes(big[:total].cpu().numpy().tobytes()) == b"".join(bytes(b.cpu().numpy().tobytes()) for b in blocks)
    short_cap = total if rank != world - 1 else max(total - 8, 0)
    rc = g.L.auvp_gather_var(ctx.h, C.c_void_p(paths.data_ptr()) if nbytes else None, nbytes, C.c_void_p(big.data_ptr()), short_cap, counts)
    ok &= (rc == -2) if total >= 8 else (rc == 0)   # every rank returns AUVP_ERR_CAPACITY, none is left waiting in the payload phase
    # ---- the gather TO A ROOT, every rank as the root in turn: grouped ncclSend / ncclRecv on the gather stream
    for root in range(world):
        ticket = g.root_begin([rec, lens.reshape(-1, 1), paths], rows=[sizes, sizes, None], root=root)
        got = g.root_end(ticket)
        if rank != root:
            ok &= got is None and g.last_root_bytes == rec.numel() + 8 * n + paths.numel() * 8
            continue
        recs, lns, pths = got
        ok &= check_all(recs, [l.reshape(-1) for l in lns], pths)
    # ---- ... ordered behind the planner stream: the records of a batch that is still running when the transfer is enqueued.
    # Seeds are global episode ids, so rank 0 can plan every rank's episodes itself and compare what it received.
    world_ = synth.make_world(seed=1, n_obstacles=64)
    ctx.set_world(world_["obstacles"], world_["habitats"], world_["polygon"], world_["bins"], world_["cells"], world_["prob"])
    E_loc = 6
    init = np.zeros((E_loc, 6))
    init[:, 0], init[:, 1] = world_["start"]
    seeds = np.arange(rank * E_loc, (rank + 1) * E_loc, dtype=np.uint64)
    ctx.rrt_prepare(init, seeds, 1500)
    ctx.rrt_run()
    view = D.device_records(ctx.L.auvp_rrt_summaries_dev(ctx.h), E_loc, _lib.SUMMARY_DTYPE.itemsize, dev)
    ticket = g.root_begin([view], rows=[[E_loc] * world], root=0)      # enqueued; nothing waited for
    got = g.root_end(ticket)
    if rank == 0:
        mine = D.tensor_to_summaries(torch.cat(got[0]), _lib.SUMMARY_DTYPE)
        init_all = np.zeros((E_loc * world, 6))
        init_all[:, 0], init_all[:, 1] = world_["start"]
        ref = ctx.rrt_explore_batch(init_all, np.arange(E_loc * world, dtype=np.uint64), 1500)
        ok &= np.array_equal(mine["n_nodes"], ref["n_nodes"]) and np.array_equal(mine["rng_after"], ref["rng_after"])
        ok &= np.array_equal(mine["best_cost"], ref["best_cost"])
    g.close()
    q.put((rank, bool(ok)))


@pytest.mark.parametrize("world,E_total", [(2, 9), (2, 1), (4, 11), (4, 2)])   # uneven shards; ranks with nothing to send
def test_gather_c_code_with_several_ranks_on_one_gpu(tmp_path, mock_lib, world, E_total):
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_worker, args=(r, world, str(tmp_path / "rccl_id"), E_total, mock_lib, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    assert sorted(res) == [(r, True) for r in range(world)], sorted(res)
