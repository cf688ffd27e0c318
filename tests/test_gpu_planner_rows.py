"""prrt_rows_kernel (four Planner_RRT episodes per wavefront, persistent rows fed from a work counter) against
prrt_kernel (one episode per wavefront) and the checker: bit-identical trees, bucket lists, counters, generator positions
and paths -- planning(max_step) and generate_one_node stepping, freq up to the kernel's limit, batches that are not a
multiple of the wave / workgroup shape, episodes that finish early next to ones that run the whole budget."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from auv_sim_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def _fields_equal(a, b):
    return all(np.array_equal(a[n], b[n]) for n in a.dtype.names)


def _plan(ctx, w, starts, goals, seeds, max_step, rows, monkeypatch, **kw):
    from auv_sim_amd._prrt_lib import PlannerBatch
    monkeypatch.setenv("AUVP_PRRT_ROWS", "1" if rows else "0")
    monkeypatch.setenv("AUVP_PRRT_LAT", "0")
    pb = PlannerBatch(ctx, starts, goals, w["rect"], max_step, seeds=seeds, **kw)
    s = pb.plan().copy()
    ctx.L.auvp_prrt_last_kernel.restype = __import__("ctypes").c_char_p
    assert ctx.L.auvp_prrt_last_kernel(ctx.h).decode() == ("prrt_rows_kernel" if rows else "prrt_kernel")
    return pb, s


@pytest.mark.parametrize("n_ep,freq,max_step,n_obst", [(37, 10, 300, 256), (64, 15, 200, 64), (5, 3, 150, 256), (130, 10, 120, 200)])
def test_rows_equals_one_episode_kernel(ctx, orc, n_ep, freq, max_step, n_obst, monkeypatch):
    from auv_sim_amd import synth
    from oracle import orc_planner as op
    w = synth.make_rect_world(seed=3, n_obstacles=n_obst)
    ctx.set_world(obstacles=w["obstacles"])
    rng = np.random.default_rng(n_ep)
    starts = np.tile(np.array([w["start"][0], w["start"][1], 0.0, 0.0]), (n_ep, 1))
    starts[:, 2] = rng.uniform(-3.0, 3.0, n_ep)
    # goals all over the box: some are reached within a few steps, some never
    goals = np.column_stack([rng.uniform(w["rect"][0] + 5, w["rect"][2] - 5, n_ep), rng.uniform(w["rect"][1] + 5, w["rect"][3] - 5, n_ep)])
    goals[0] = [w["start"][0] + 6.0, w["start"][1] + 1.0]
    seeds = np.arange(n_ep, dtype=np.uint64) + 11
    kw = dict(freq=freq, cell=5, subs=2)
    pa, a = _plan(ctx, w, starts, goals, seeds, max_step, False, monkeypatch, **kw)
    ta = [pa.tree(e, a[e]) for e in range(n_ep)]
    ga = [pa.grid(e) for e in range(n_ep)]
    paths_a = pa.paths(a)
    pb, b = _plan(ctx, w, starts, goals, seeds, max_step, True, monkeypatch, **kw)
    assert (a["status"] >= 0).all() and _fields_equal(a, b)
    assert (a["done"] == 1).any() and (a["done"] == 0).any()  # both kinds of episode are in the batch
    paths_b = pb.paths(b)
    for e in range(n_ep):
        tb = pb.tree(e, b[e])
        for k in ta[e]:
            assert np.array_equal(ta[e][k], tb[k]), (e, k)
        gb = pb.grid(e)
        assert np.array_equal(ga[e][0], gb[0]) and np.array_equal(ga[e][1], gb[1])
        assert np.array_equal(paths_a[e], paths_b[e])
    for e in (0, n_ep // 2, n_ep - 1):
        r = op.planning(w["obstacles"], w["rect"], starts[e], goals[e], int(seeds[e]), max_step, freq, 5, 2, kind="portable")
        assert (b[e]["steps"], bool(b[e]["done"]), b[e]["n_nodes"], b[e]["n_points"]) == (r["steps"], bool(r["done"]), r["n_nodes"], r["n_points"])
        assert b[e]["rng_after"] == r["rng_after"]


def test_rows_step_mode_equals_one_episode_kernel(ctx, monkeypatch):
    """generate_one_node stepping (caller-chosen buckets, empty buckets, skipped episodes) through both kernels"""
    from auv_sim_amd import synth
    from auv_sim_amd._prrt_lib import PlannerBatch
    w = synth.make_rect_world(seed=5, n_obstacles=128)
    ctx.set_world(obstacles=w["obstacles"])
    E, n_steps = 45, 60
    starts = np.tile(np.array([w["start"][0], w["start"][1], 0.3, 0.0]), (E, 1))
    goals = np.tile(w["goal"], (E, 1))
    seeds = np.arange(E, dtype=np.uint64)
    monkeypatch.setenv("AUVP_PRRT_LAT", "0")
    out = {}
    for rows in (False, True):
        monkeypatch.setenv("AUVP_PRRT_ROWS", "1" if rows else "0")
        pb = PlannerBatch(ctx, starts, goals, w["rect"], n_steps + 4, seeds=seeds, freq=10, cell=5, subs=1)
        rng = np.random.default_rng(2)
        log = []
        for i in range(n_steps):
            occ = [pb.grid(e)[0] for e in (0, 1)]
            buckets = np.array([int(rng.choice(pb.grid(e)[0])) if rng.random() < 0.8 else int(rng.integers(0, pb.rows * pb.cols))
                                for e in range(E)], dtype=np.int32)
            buckets[i % E] = -1  # this episode sits the step out
            s = pb.step(buckets)
            log.append(s.copy())
        out[rows] = (log, [pb.tree(e, log[-1][e]) for e in range(E)])
    for sa, sb in zip(out[False][0], out[True][0]):
        assert _fields_equal(sa, sb)
    for ta, tb in zip(out[False][1], out[True][1]):
        for k in ta:
            assert np.array_equal(ta[k], tb[k]), k
