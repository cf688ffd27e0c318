"""prrt_pipe_kernel (four wavefronts per Planner_RRT episode, a speculative pipeline over the steps: planner_pipe_kernel.h)
against prrt_kernel and the checker: bit-identical trees, bucket lists, counters, generator state / position and paths -- and
the planning can be continued by generate_one_node steps of the other kernel.  (Rounds 3-4 also had a two / three-wavefront
prrt_duo_kernel here; removed in round 5.)"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from auv_sim_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def _fields_equal(a, b):
    return all(np.array_equal(a[n], b[n]) for n in a.dtype.names)


def _plan(ctx, w, starts, goals, seeds, max_step, duo, monkeypatch, **kw):
    """duo: 0 = prrt_kernel, 4 = prrt_pipe_kernel with four wavefronts per episode, 5 = with the draw wavefront (round 6)"""
    from auv_sim_amd._prrt_lib import PlannerBatch
    monkeypatch.setenv("AUVP_PRRT_ROWS", "0")
    monkeypatch.setenv("AUVP_PRRT_PIPE", "1" if duo >= 4 else "0")
    monkeypatch.setenv("AUVP_PRRT_PIPE_DRAW", "1" if duo == 5 else "0")
    pb = PlannerBatch(ctx, starts, goals, w["rect"], max_step, seeds=seeds, **kw)
    s = pb.plan().copy()
    ctx.L.auvp_prrt_last_kernel.restype = C.c_char_p
    assert ctx.L.auvp_prrt_last_kernel(ctx.h).decode() == {0: "prrt_kernel", 4: "prrt_pipe_kernel", 5: "prrt_pipe_kernel"}[duo]
    assert ctx.last_launch()[1] % {0: 64, 4: 256, 5: 320}[duo] == 0   # threads per workgroup: 4 / 5 wavefronts per episode
    assert ctx.pipeline_fallbacks()[0] == 0
    return pb, s


@pytest.mark.parametrize("waves", [4, 5])
@pytest.mark.parametrize("n_ep,freq,max_step,n_obst,subs", [(37, 10, 400, 256, 2), (64, 15, 250, 64, 1), (5, 3, 300, 256, 4), (1, 10, 2000, 256, 1),
                                                           (130, 30, 150, 128, 2), (9, 10, 1, 64, 1)])
def test_duo_equals_one_wavefront_per_episode(ctx, orc, n_ep, freq, max_step, n_obst, subs, waves, monkeypatch):
    from auv_sim_amd import synth
    from oracle import orc_planner as op
    w = synth.make_rect_world(seed=3, n_obstacles=n_obst)
    ctx.set_world(obstacles=w["obstacles"])
    rng = np.random.default_rng(n_ep)
    starts = np.tile(np.array([w["start"][0], w["start"][1], 0.0, 0.0]), (n_ep, 1))
    starts[:, 2] = rng.uniform(-3.0, 3.0, n_ep)
    goals = np.column_stack([rng.uniform(w["rect"][0] + 5, w["rect"][2] - 5, n_ep), rng.uniform(w["rect"][1] + 5, w["rect"][3] - 5, n_ep)])
    if n_ep > 1:
        goals[0] = [w["start"][0] + 6.0, w["start"][1] + 1.0]
    seeds = np.arange(n_ep, dtype=np.uint64) + 11
    kw = dict(freq=freq, cell=5, subs=subs)
    pa, a = _plan(ctx, w, starts, goals, seeds, max_step, 0, monkeypatch, **kw)
    ta = [pa.tree(e, a[e]) for e in range(n_ep)]
    ga = [pa.grid(e) for e in range(n_ep)]
    paths_a = pa.paths(a)
    # continue every unfinished episode by three generate_one_node steps (one-wavefront kernel): the generator state it finds
    nxt = np.array([int(g[0][0]) if len(g[0]) else 0 for g in ga], dtype=np.int32)
    cont_a = [pa.step(nxt).copy() for _ in range(3)][-1]
    pb, b = _plan(ctx, w, starts, goals, seeds, max_step, waves, monkeypatch, **kw)
    assert (a["status"] >= 0).all(), np.unique(a["status"])
    assert _fields_equal(a, b), [n for n in a.dtype.names if not np.array_equal(a[n], b[n])]
    paths_b = pb.paths(b)
    for e in range(n_ep):
        tb = pb.tree(e, b[e])
        for k in ta[e]:
            assert np.array_equal(ta[e][k], tb[k]), (e, k)
        gb = pb.grid(e)
        assert np.array_equal(ga[e][0], gb[0]) and np.array_equal(ga[e][1], gb[1])
        assert np.array_equal(paths_a[e], paths_b[e])
    cont_b = [pb.step(nxt).copy() for _ in range(3)][-1]
    assert _fields_equal(cont_a, cont_b)
    for e in sorted({0, n_ep // 2, n_ep - 1}):
        r = op.planning(w["obstacles"], w["rect"], starts[e], goals[e], int(seeds[e]), max_step, freq, 5, subs, kind="portable")
        assert (b[e]["steps"], bool(b[e]["done"]), b[e]["n_nodes"], b[e]["n_points"]) == (r["steps"], bool(r["done"]), r["n_nodes"], r["n_points"])
        assert b[e]["rng_after"] == r["rng_after"]


@pytest.mark.parametrize("next_lds", ["0", "1"])
def test_pipeline_continues_a_tree_with_and_without_the_link_mirror(ctx, next_lds, monkeypatch):
    """prrt_pipe_kernel keeps the member lists' next links in LDS where they fit (AUVP_PRRT_NEXT_LDS): a launch that finds a
    partly grown tree copies the links it starts from; the mirror on and off give the one-wavefront kernel's result"""
    from auv_sim_amd import synth
    from auv_sim_amd._prrt_lib import PlannerBatch
    w = synth.make_rect_world(seed=5, n_obstacles=128)
    ctx.set_world(obstacles=w["obstacles"])
    n_ep, max_step = 24, 500
    rng = np.random.default_rng(77)
    starts = np.tile(np.array([w["start"][0], w["start"][1], 0.0, 0.0]), (n_ep, 1))
    starts[:, 2] = rng.uniform(-3.0, 3.0, n_ep)
    goals = np.column_stack([rng.uniform(w["rect"][0] + 5, w["rect"][2] - 5, n_ep), rng.uniform(w["rect"][1] + 5, w["rect"][3] - 5, n_ep)])
    seeds = np.arange(n_ep, dtype=np.uint64) + 501
    res = []
    for pipe in (0, 1):
        monkeypatch.setenv("AUVP_PRRT_ROWS", "0")
        monkeypatch.setenv("AUVP_PRRT_PIPE", str(pipe))
        monkeypatch.setenv("AUVP_PRRT_NEXT_LDS", next_lds)
        pb = PlannerBatch(ctx, starts, goals, w["rect"], max_step, seeds=seeds, freq=10, cell=5, subs=2)
        for _ in range(40):  # a tree of up to 41 nodes grown by generate_one_node steps from the start's bucket
            g = [pb.grid(e) for e in range(n_ep)]
            pb.step(np.array([int(x[0][-1]) if len(x[0]) else 0 for x in g], dtype=np.int32))
        s = pb.plan().copy()
        ctx.L.auvp_prrt_last_kernel.restype = C.c_char_p
        assert ctx.L.auvp_prrt_last_kernel(ctx.h).decode() == ("prrt_pipe_kernel" if pipe else "prrt_kernel")
        res.append((s, [pb.tree(e, s[e]) for e in range(n_ep)], pb.paths(s)))
    (a, ta, pa_), (b, tb, pb_) = res
    assert (a["n_nodes"] > 41).any()
    assert _fields_equal(a, b), [n for n in a.dtype.names if not np.array_equal(a[n], b[n])]
    for e in range(n_ep):
        for k in ta[e]:
            assert np.array_equal(ta[e][k], tb[e][k]), (e, k)
        assert np.array_equal(pa_[e], pb_[e])
