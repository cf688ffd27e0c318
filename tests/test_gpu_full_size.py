"""BASELINE.json's full sizes (10 000-iteration budget, 256 obstacles, 200x200-cell grid; 512 x 2000-step
Planner_RRT; 1024 A* instances; thousands of particle filters).  The CPU checker is too slow to replay
whole batches at these sizes, so parity is carried by
  * the checker on a SAMPLE of the batch (bit for bit), and
  * size-independent properties over the WHOLE batch: a launch is deterministic, an episode's result does
    not depend on the batch it runs in (seed = episode id), and the tree / list invariants of the algorithm
    hold for every episode."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from auv_sim_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def _fields_equal(a, b, skip=()):
    return all(np.array_equal(a[n], b[n]) for n in a.dtype.names if n not in skip)


@pytest.mark.parametrize("n_obstacles,rows", [(256, "1"), (256, "0"), (64, "1")])
def test_rrt_exploring_full_budget(ctx, orc, n_obstacles, rows, monkeypatch):
    """the headline world (256 obstacles) through both expansion kernels -- four episodes per wavefront
    (rrt_rows_kernel, AUVP_ROWS=1, the default) and one (rrt_explore_kernel) -- and BASELINE configs[1] as written
    (64 obstacles), all at the full 10 000-iteration budget on the 200x200-cell grid"""
    from auv_sim_amd import synth
    monkeypatch.setenv("AUVP_ROWS", rows)
    n_iter, E = 10000, 512
    world = synth.make_world(seed=2, n_obstacles=n_obstacles, box=(-1000.0, -1000.0, 1000.0, 1000.0), cell=10.0, n_bins=10,
                             bin_len=50, n_habitats=10)
    assert len(world["cells"]) == 40000
    ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    init = np.zeros((E, 6))
    init[:, 0], init[:, 1] = world["start"]
    seeds = np.arange(E, dtype=np.uint64)
    kw = dict(mode="timebin", freq=30, bin_interval=5, v=2, max_traj_time=500.0, weights=(-3, -3, -4))
    s1 = ctx.rrt_explore_batch(init, seeds, n_iter, **kw).copy()
    paths1 = ctx.paths(s1)
    trees = {e: ctx.tree(e, s1[e]) for e in (0, 3, 511)}
    # -- properties over the whole batch
    assert (s1["status"] == 0).all() and (s1["iters_run"] == n_iter).all()
    assert (s1["n_nodes"] >= 2).all() and (s1["n_nodes"] <= n_iter + 1).all()
    assert (s1["n_leaves"] > 0).all() and np.isfinite(s1["best_cost"]).all()
    assert (np.abs(s1["best_cost"][:, 1:].sum(axis=1) - s1["best_cost"][:, 0]) < 1e-9).all()
    assert len({tuple(r) for r in s1["best_cost"].tolist()}) > E // 2  # the seeds really differ
    for e, t in trees.items():
        n = len(t["parent"])
        assert t["parent"][0] == -1 and (t["parent"][1:] < np.arange(1, n)).all() and (t["parent"][1:] >= 0).all()
        assert (t["nodes"][1:, 3] >= t["nodes"][t["parent"][1:], 3]).all()          # traj time never decreases
        assert (t["nodes"][1:, 5] >= t["nodes"][t["parent"][1:], 5]).all()          # nor does the length
        assert np.array_equal(t["pt_off"][1:], np.cumsum(t["pt_cnt"])[:-1])         # points stored contiguously
        assert int(t["pt_cnt"].sum()) == int(s1[e]["n_points"])
        p = paths1[e]
        assert np.array_equal(p[0, :2], init[e, :2]) and (np.diff(p[:, 4]) >= 0).all()
        assert p[-1, 4] >= 500.0 - 30 and p[-1, 6] == s1[e]["best_length"]          # a qualifying leaf (:158)
    # -- deterministic
    s2 = ctx.rrt_explore_batch(init, seeds, n_iter, **kw)
    assert _fields_equal(s1, s2)
    if rows == "1":
        # (round 6) ... and through both forms of the four-episode kernel at the full budget, every field: the second batch with a
        # parameter block reads the numbers rrt_stream_kernel wrote ahead; with the option off the generator runs inside the kernel
        assert ctx.last_rrt_kernel() == "rrt_rows_stream_kernel" and ctx.last_stream_len() > 440000 and ctx.pipeline_fallbacks()[0] == 0
        monkeypatch.setenv("AUVP_ROWS_STREAM", "0")
        s2c = ctx.rrt_explore_batch(init, seeds, n_iter, **kw)
        assert ctx.last_rrt_kernel() == "rrt_rows_kernel" and _fields_equal(s2, s2c)
        monkeypatch.delenv("AUVP_ROWS_STREAM")
    # -- an episode does not depend on its batch
    pick = np.array([0, 3, 200, 511])
    s3 = ctx.rrt_explore_batch(init[pick], seeds[pick], n_iter, **kw)
    assert _fields_equal(s1[pick], s3)
    assert ctx.last_launch_parts()[2] == (4 if rows == "1" else 1)  # the kernel the case names really ran
    for k, e in enumerate(pick):
        assert np.array_equal(ctx.paths(s3)[k], paths1[e])
    # -- the checker on a sample, bit for bit
    w = orc.WorldArrays(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    for e in (3, 511):
        r = orc.rrt_explore(w, int(seeds[e]), n_iter, mode="timebin", init=init[e], kind="portable", max_traj_time=500.0)
        s = s1[e]
        assert (s["status"], s["n_nodes"], s["n_points"], s["n_leaves"]) == (r["status"], r["n_nodes"], r["n_points"], r["n_leaves"])
        assert s["rng_after"] == r["rng_after"] and int(s["n_draw32"]) == int(r["n_draw32"])
        assert np.array_equal(np.array(s["best_cost"]), r["best_cost"]) and s["best_length"] == r["best_length"]
        assert np.array_equal(trees[e]["parent"], r["parent"]) and np.array_equal(trees[e]["nodes"], r["nodes"])
        assert np.array_equal(paths1[e], r["path"])


def test_rrt_exploring_nn_full_budget(ctx, orc):
    """nearest-neighbour parent sampling (plan_time=False, rrt_dubins.py:333-343,505-513) at the full 10 000-iteration
    budget on the headline world: every iteration scans the whole tree (streaming x,y mirror)"""
    from auv_sim_amd import synth
    n_iter, E = 10000, 320
    world = synth.make_world(seed=2, n_obstacles=256, box=(-1000.0, -1000.0, 1000.0, 1000.0), cell=10.0, n_bins=10,
                             bin_len=50, n_habitats=10)
    ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    init = np.zeros((E, 6))
    init[:, 0], init[:, 1] = world["start"]
    seeds = np.arange(E, dtype=np.uint64)
    kw = dict(mode="nn", freq=30, bin_interval=5, v=2, max_traj_time=500.0, weights=(-3, -3, -4))
    s1 = ctx.rrt_explore_batch(init, seeds, n_iter, **kw).copy()
    trees = {e: ctx.tree(e, s1[e]) for e in (1, 319)}
    assert (s1["status"] >= 0).all() and (s1["iters_run"] == n_iter).all()
    assert (s1["n_nodes"] >= 2).all() and (s1["n_nodes"] <= n_iter + 1).all()
    # len(mps_list) summed over the scans is bounded by the final tree size and by the triangle of a tree that accepts everything
    assert (s1["nn_scanned"] <= s1["n_nodes"].astype(np.uint64) * n_iter).all()
    assert (s1["nn_scanned"] >= s1["n_nodes"].astype(np.uint64) * (s1["n_nodes"].astype(np.uint64) - 1) // 2).all()
    for e, t in trees.items():
        n = len(t["parent"])
        assert t["parent"][0] == -1 and (t["parent"][1:] < np.arange(1, n)).all() and (t["parent"][1:] >= 0).all()
        assert (t["nodes"][t["parent"][1:], 3] <= 500.0).all()                      # no parent beyond max_traj_time (:138)
        plan_it = t["nodes"][1:, 4]
        assert int(s1[e]["nn_scanned"]) == int(n_iter + (n_iter - 1 - plan_it).sum())
    s2 = ctx.rrt_explore_batch(init, seeds, n_iter, **kw)
    assert _fields_equal(s1, s2)
    pick = np.array([1, 100, 319])
    s3 = ctx.rrt_explore_batch(init[pick], seeds[pick], n_iter, **kw)
    assert _fields_equal(s1[pick], s3)
    w = orc.WorldArrays(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    for e in (1, 319):
        r = orc.rrt_explore(w, int(seeds[e]), n_iter, mode="nn", init=init[e], kind="portable", max_traj_time=500.0)
        s = s1[e]
        assert (s["status"], s["n_nodes"], s["n_points"], s["n_leaves"]) == (r["status"], r["n_nodes"], r["n_points"], r["n_leaves"])
        assert s["rng_after"] == r["rng_after"] and int(s["n_draw32"]) == int(r["n_draw32"])
        assert np.array_equal(np.array(s["best_cost"]), r["best_cost"]) and s["best_length"] == r["best_length"]
        assert np.array_equal(trees[e]["parent"], r["parent"]) and np.array_equal(trees[e]["nodes"], r["nodes"])


def test_planner_rrt_config4(ctx, orc):
    from auv_sim_amd import synth
    from auv_sim_amd._prrt_lib import PlannerBatch
    from oracle import orc_planner as op
    n_ep, max_step = 512, 2000
    w = synth.make_rect_world(seed=3, n_obstacles=256)
    ctx.set_world(obstacles=w["obstacles"])
    starts = np.tile(np.array([w["start"][0], w["start"][1], 0.0, 0.0]), (n_ep, 1))
    goals = np.tile(w["goal"], (n_ep, 1))
    seeds = np.arange(n_ep, dtype=np.uint64)
    a = PlannerBatch(ctx, starts, goals, w["rect"], max_step, seeds=seeds, freq=10, cell=5, subs=1).plan().copy()
    b = PlannerBatch(ctx, starts, goals, w["rect"], max_step, seeds=seeds, freq=10, cell=5, subs=1).plan()
    assert (a["status"] >= 0).all() and _fields_equal(a, b)
    assert ((a["steps"] == max_step) | (a["done"] == 1)).all() and (a["n_nodes"] <= a["steps"] + 1).all()
    pick = np.array([0, 7, 300])
    c = PlannerBatch(ctx, starts[pick], goals[pick], w["rect"], max_step, seeds=seeds[pick], freq=10, cell=5, subs=1).plan()
    assert _fields_equal(a[pick], c)
    for e in (7, 300):
        r = op.planning(w["obstacles"], w["rect"], starts[e], goals[e], int(seeds[e]), max_step, 10, 5, 1, kind="portable")
        assert (a[e]["steps"], bool(a[e]["done"]), a[e]["n_nodes"], a[e]["n_points"]) == (r["steps"], bool(r["done"]), r["n_nodes"], r["n_points"])
        assert a[e]["rng_after"] == r["rng_after"]


def test_astar_config3(ctx, orc):
    from auv_sim_amd import _astar_lib, synth
    from oracle import orc_astar as oa
    n_inst = 1024
    w = synth.make_world(seed=12, n_obstacles=64, obst_radius=(2.0, 6.0), n_habitats=10, hab_radius=(10.0, 25.0))
    ctx.set_world(w["obstacles"], w["habitats"], w["polygon"], w["bins"], w["cells"], w["prob"])
    rng = np.random.default_rng(3)
    starts = np.column_stack([-290.0 + 10.0 * rng.integers(0, 19, n_inst), -90.0 + 10.0 * rng.integers(0, 19, n_inst)])
    limits = rng.choice([100.0, 200.0, 300.0], n_inst)
    kw = dict(limits=limits, weights=(0, 10, 10, 100), velocity=1.0, cap_nodes=20000)
    a = _astar_lib.run_batch(ctx, "astar_fixLenSOG", starts, **kw)
    b = _astar_lib.run_batch(ctx, "astar_fixLenSOG", starts, **kw)
    assert all(x["status"] >= 0 for x in a)
    for x, y in zip(a, b):
        assert (x["found"], x["n_nodes"], x["n_children"]) == (y["found"], y["n_nodes"], y["n_children"])
        assert np.array_equal(x["path"], y["path"])
    for x, lim in zip(a, limits):
        if x["found"]:
            assert abs(x["node_path"][-1][6] - lim) <= 10                     # stop rule: |pathLen - limit| <= 10
            d = np.diff(np.asarray(x["path"])[:, :2], axis=0)
            assert set(np.round(np.hypot(d[:, 0], d[:, 1]), 6).tolist()) <= {10.0, round(10 * 2 ** 0.5, 6)}
    for e in (0, 17, 600, 1023):
        o = oa.run("astar_fixLenSOG", starts[e], obstacles=w["obstacles"], habitats=w["habitats"], polygon=w["polygon"],
                   bins=w["bins"], cells=w["cells"], prob=w["prob"], limit=float(limits[e]), weights=(0, 10, 10, 100),
                   velocity=1.0, cap_nodes=20000, kind="portable")
        assert (a[e]["found"], a[e]["n_nodes"], a[e]["n_children"]) == (o["found"], o["n_nodes"], o["n_children"])
        assert np.array_equal(a[e]["path"], o["path"]) and np.array_equal(a[e]["cost_list"], o["cost_list"])


def test_particle_filters_at_scale(ctx, orc):
    from auv_sim_amd import _pf_lib
    from oracle import orc_pf
    rng = np.random.default_rng(4)
    F, N, S, A = 2048, 1000, 10, 2
    shark0 = rng.uniform(-500, 500, size=(F, 2))
    meas = np.zeros((S, F, A, 5))
    meas[..., 0:2] = shark0[None, :, None, :] + rng.uniform(-150, 150, size=(S, F, A, 2))
    meas[..., 2] = rng.uniform(-np.pi, np.pi, size=(S, F, A))
    meas[..., 3] = rng.uniform(0, 200, size=(S, F, A))
    meas[..., 4] = rng.uniform(-np.pi, np.pi, size=(S, F, A))
    shark = shark0[None] + rng.uniform(-20, 20, size=(S, F, 2))
    key0, _ = _pf_lib.np_seed_state(0)
    mts = np.stack([np.roll(key0, f) ^ np.uint32(f) for f in range(F)])
    b = _pf_lib.FilterBatch(ctx, F, N).create(shark0, mts, 624).run(meas=meas, shark_xy=shark)
    part, obj = b.particles()
    mean, err, ll = b.estimates()
    st, nd = b.status()
    assert (st == 0).all()
    assert ((ll >= N) & (ll <= 5 * N)).all()                                       # 1..5 copies per particle
    assert ((obj >= 0) & (obj < ll[-1][:, None])).all()
    assert (part[:, :, 4].max(axis=1) <= 1.0).all() and (part[:, :, 4] > 0).all()  # normalised weights
    assert (np.abs(part[:, :, 3]) <= np.pi).all() and (part[:, :, 2] <= 5.0).all() and (part[:, :, 2] >= 0).all()
    assert np.allclose(part[:, :, 0].mean(axis=1), mean[-1, :, 0], rtol=1e-12, atol=1e-9)
    assert np.allclose(err[-1], np.hypot(mean[-1, :, 0] - shark[-1, :, 0], mean[-1, :, 1] - shark[-1, :, 1]), rtol=1e-12)
    # a filter does not depend on its batch, and equals the checker
    for f in (0, 1033, 2047):
        one = _pf_lib.FilterBatch(ctx, 1, N).create(shark0[f:f + 1], mts[f:f + 1], 624).run(meas=meas[:, f:f + 1],
                                                                                            shark_xy=shark[:, f:f + 1])
        p1, o1 = one.particles()
        assert np.array_equal(p1[0], part[f]) and np.array_equal(o1[0], obj[f])
        ref = orc_pf.run(N, meas[:, f], shark[:, f], shark0[f], mts[f], 624, kind="portable")
        assert np.array_equal(part[f], ref["resampled"][-1]) and np.array_equal(mean[:, f], ref["mean"])
        assert int(nd[f]) == ref["n_draw32"]
