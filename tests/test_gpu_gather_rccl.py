"""The RCCL gather of the C-ABI (auvp_comm_init / auvp_gather / auvp_gather_var, include/auvplan.h) on the one GPU this
box has: a world_size-1 communicator exercises the whole call path (run-time binding of librccl, communicator on the
handle, collectives on the handle's stream, two-phase variable-length gather); the N > 1 arithmetic of the callers is
covered by tests/test_distributed_gloo.py."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rccl_gather_single_rank_round_trip():
    import torch
    from auv_sim_amd import _lib, distributed as D
    ctx = _lib.Context(0)
    assert D.RcclGather.usable()
    g = D.RcclGather(ctx, 0, 1, lambda mine: mine)
    assert g.info() == (1, 0, 1)  # world size given; rank and rank count as the communicator reports them
    # ONE RCCL per process: torch (imported above) has mapped its bundled librccl, and the C-ABI must have bound that image
    lib = D.RcclGather.library()
    mapped = [l.split()[-1] for l in open("/proc/self/maps") if "librccl.so" in l]
    assert lib and mapped and lib in mapped and len(set(mapped)) == 1, (lib, sorted(set(mapped)))
    dev = torch.device("cuda", 0)
    rec = torch.arange(7 * 112, dtype=torch.uint8, device=dev).reshape(7, 112)
    out = g.gather_records(rec)
    assert len(out) == 1 and torch.equal(out[0], rec)
    lens = torch.tensor([3, 0, 5], dtype=torch.int64, device=dev)
    paths = torch.arange(8 * 7, dtype=torch.float64, device=dev).reshape(8, 7)
    all_len, blocks = g.gather_paths(paths, lens)
    assert torch.equal(all_len[0], lens) and torch.equal(blocks[0], paths)
    empty = g.gather_records(torch.zeros((0, 112), dtype=torch.uint8, device=dev))
    assert empty[0].shape == (0, 112)
    assert g.take_ms() is not None
    # fixed-size entry point + device-resident summaries view
    init = np.zeros((4, 6))
    ctx.set_world(polygon=[[-50.0, -50.0], [50.0, -50.0], [50.0, 50.0], [-50.0, 50.0]])
    summ = ctx.rrt_explore_batch(init, [1, 2, 3, 4], 50, max_traj_time=20.0)
    view = D.device_records(ctx.L.auvp_rrt_summaries_dev(ctx.h), 4, _lib.SUMMARY_DTYPE.itemsize, dev)
    got = D.tensor_to_summaries(g.gather_records(view)[0], _lib.SUMMARY_DTYPE)
    assert np.array_equal(got["n_nodes"], summ["n_nodes"]) and np.array_equal(got["rng_after"], summ["rng_after"])
    recv = torch.empty_like(view)
    ctx._chk(g.L.auvp_gather(ctx.h, C.c_void_p(view.data_ptr()), view.numel(), C.c_void_p(recv.data_ptr())))
    assert torch.equal(recv, view)
    # ---- round 6: the gather to a root on the handle's gather stream (auvp_gather_blocks_root_async / auvp_gather_wait).  With
    # one rank the transfer is the root's own block -- a device copy on the gather stream, ordered behind the planner stream's
    # work by an event -- so the stream / event / ticket plumbing runs; ncclSend / ncclRecv need a second GPU (below)
    summ = ctx.rrt_explore_batch(init, [5, 6, 7, 8], 400, max_traj_time=20.0)       # the kernels whose results are sent ...
    view = D.device_records(ctx.L.auvp_rrt_summaries_dev(ctx.h), 4, _lib.SUMMARY_DTYPE.itemsize, dev)
    ticket = g.root_begin([view, lens.reshape(-1, 1), paths], rows=[[4], [3], None], root=0)  # ... enqueued without waiting
    with pytest.raises(RuntimeError):
        g.root_begin([view], rows=[[4]], root=0)                                      # one ticket at a time
    recs, lns, pths = g.root_end(ticket)
    got = D.tensor_to_summaries(recs[0], _lib.SUMMARY_DTYPE)
    assert np.array_equal(got["n_nodes"], summ["n_nodes"]) and np.array_equal(got["rng_after"], summ["rng_after"])
    assert torch.equal(lns[0].reshape(-1), lens) and torch.equal(pths[0], paths)
    assert g.last_root_bytes == view.numel() + lens.numel() * 8 + paths.numel() * 8 and g.last_root_ms >= 0.0
    assert g.root_end({"root": 0, "items": []}) == []                                # nothing pending: a no-op
    bad = (C.c_int64 * 1)(16)
    assert g.L.auvp_gather_blocks_root_async(ctx.h, 3, C.c_void_p(view.data_ptr()), None, 0, bad) == -1   # root outside the communicator
    assert g.L.auvp_gather_blocks_root_async(ctx.h, 0, C.c_void_p(view.data_ptr()), C.c_void_p(recv.data_ptr()), 8, bad) == -2  # short buffer
    g.close()


def _two_rank_worker(rank, world, id_path, E_total, q):
    """one process per GPU: the C-ABI gather with a real two-rank RCCL communicator (no torch.distributed involved)"""
    import os
    import sys
    import time
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from auv_sim_amd import _lib, distributed as D
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    ctx = _lib.Context(rank)

    def exchange(mine):  # rank 0 writes the 128-byte id to a file, the others wait for it
        if rank == 0:
            with open(id_path + ".tmp", "wb") as f:
                f.write(mine)
            os.replace(id_path + ".tmp", id_path)
            return mine
        for _ in range(600):
            if os.path.exists(id_path):
                return open(id_path, "rb").read()
            time.sleep(0.1)
        return b""
    g = D.RcclGather(ctx, rank, world, exchange)
    ok = g.info() == (world, rank, world)
    lo, hi = D.shard_range(E_total, rank, world)
    n = hi - lo
    rec = torch.zeros((n, 112), dtype=torch.uint8, device=dev)
    for i in range(n):
        rec[i] = (lo + i) % 251
    out = g.gather_records(rec)
    lens = torch.tensor([2 + ((lo + i) % 4) for i in range(n)], dtype=torch.int64, device=dev)
    paths = torch.zeros((int(lens.sum().item()), 7), dtype=torch.float64, device=dev)
    pos = 0
    for i in range(n):
        L = int(lens[i])
        paths[pos:pos + L, 0] = lo + i
        pos += L
    all_len, blocks = g.gather_paths(paths, lens)
    for r in range(world):
        rlo, rhi = D.shard_range(E_total, r, world)
        ok &= out[r].shape == (rhi - rlo, 112) and len(all_len[r]) == rhi - rlo
        pos = 0
        for i in range(rhi - rlo):
            e = rlo + i
            ok &= bool((out[r][i] == e % 251).all())
            L = int(all_len[r][i])
            ok &= L == 2 + (e % 4) and bool((blocks[r][pos:pos + L, 0] == e).all())
            pos += L
    # the equal-block entry point too
    send = torch.full((16,), rank + 1, dtype=torch.uint8, device=dev)
    recv = torch.zeros(16 * world, dtype=torch.uint8, device=dev)
    ctx._chk(g.L.auvp_gather(ctx.h, C.c_void_p(send.data_ptr()), 16, C.c_void_p(recv.data_ptr())))
    ok &= all(bool((recv[16 * r:16 * r + 16] == r + 1).all()) for r in range(world))
    # the gather to a root: grouped ncclSend / ncclRecv on the gather stream, both roots
    sizes = D.shard_sizes(E_total, world)
    for root in range(world):
        ticket = g.root_begin([rec, lens.reshape(-1, 1), paths], rows=[sizes, sizes, None], root=root)
        got = g.root_end(ticket)
        if rank != root:
            ok &= got is None
            continue
        recs, lns, pths = got
        for r in range(world):
            rlo, rhi = D.shard_range(E_total, r, world)
            ok &= recs[r].shape == (rhi - rlo, 112) and lns[r].shape == (rhi - rlo, 1)
            pos = 0
            for i in range(rhi - rlo):
                e = rlo + i
                L = int(lns[r][i, 0])
                ok &= bool((recs[r][i] == e % 251).all()) and L == 2 + (e % 4) and bool((pths[r][pos:pos + L, 0] == e).all())
                pos += L
    g.close()
    q.put((rank, bool(ok)))


@pytest.mark.parametrize("E_total", [9, 1])  # uneven shards; one rank with nothing to send
def test_rccl_gather_two_ranks(tmp_path, E_total):
    """needs two GPUs (skipped on the one-GPU boxes of this pool): two processes, one GPU each, a real two-rank communicator
    through auvp_comm_init, the count exchange + grouped broadcasts of the variable-length gather with different roots"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_two_rank_worker, args=(r, 2, str(tmp_path / "rccl_id"), E_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]
