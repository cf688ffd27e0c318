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
    g.close()
