"""Edge cases the goldens do not reach, HIP vs the CPU checker (portable build), bit-for-bit:
large obstacle lists (the J = 8 / 16 register layouts), steer chunking (freq > 63) in Planner_RRT,
empty worlds, single-iteration budgets, A* with no obstacles."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from auv_sim_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("O,mode", [(400, "timebin"), (900, "timebin"), (130, "nn"), (1, "plantime")])
def test_exploring_many_obstacles(ctx, orc, O, mode):
    from auv_sim_amd import synth
    world = synth.make_world(seed=40 + O, n_obstacles=O, obst_radius=(0.5, 2.5))
    ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    w = orc.WorldArrays(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    E, n_iter = 5, 500
    init = np.zeros((E, 6))
    init[:, 0], init[:, 1] = world["start"]
    seeds = np.arange(70, 70 + E, dtype=np.uint64)
    summ = ctx.rrt_explore_batch(init, seeds, n_iter, mode=mode)
    for e in range(E):
        r = orc.rrt_explore(w, int(seeds[e]), n_iter, mode=mode, init=init[e], kind="portable")
        s = summ[e]
        assert s["status"] == r["status"] and s["n_nodes"] == r["n_nodes"] and s["rng_after"] == r["rng_after"]
        t = ctx.tree(e, s)
        assert np.array_equal(t["parent"], r["parent"]) and np.array_equal(t["nodes"], r["nodes"])
        if r["status"] == 0:
            assert np.array_equal(np.array(s["best_cost"]), r["best_cost"])


def test_exploring_empty_world_and_tiny_budget(ctx, orc):
    poly = [[-50.0, -50.0], [50.0, -50.0], [50.0, 50.0], [-50.0, 50.0]]
    ctx.set_world(polygon=poly)
    w = orc.WorldArrays(polygon=poly)
    init = np.zeros((3, 6))
    for n_iter in (1, 2, 40):
        summ = ctx.rrt_explore_batch(init, [5, 6, 7], n_iter, max_traj_time=20.0, bin_interval=5)
        for e in range(3):
            r = orc.rrt_explore(w, 5 + e, n_iter, init=init[e], max_traj_time=20.0, bin_interval=5, kind="portable")
            s = summ[e]
            assert s["status"] == r["status"] and s["n_nodes"] == r["n_nodes"] and s["rng_after"] == r["rng_after"]
            if r["status"] == 0:
                assert np.array_equal(np.array(s["best_cost"]), r["best_cost"])  # no grid, no habitats: all zeros


def test_exploring_without_boundary_polygon(ctx, orc):
    """V = 0 (the Python default polygon=None): no point is within an empty boundary, so every expansion is
    rejected -- the checker's answer -- and the kernel must return that instead of dividing by zero; the
    collision probe says "not free" for any path; a 1- or 2-vertex boundary is an argument error."""
    from auv_sim_amd import _lib, synth
    world = synth.make_world(seed=3, n_obstacles=8)
    ctx.set_world(world["obstacles"], world["habitats"], None, world["bins"], world["cells"], world["prob"])
    w = orc.WorldArrays(world["obstacles"], world["habitats"], None, world["bins"], world["cells"], world["prob"])
    init = np.zeros((2, 6))
    init[:, 0], init[:, 1] = world["start"]
    summ = ctx.rrt_explore_batch(init, [1, 2], 50)
    for e in range(2):
        r = orc.rrt_explore(w, 1 + e, 50, init=init[e], kind="portable")
        assert r["n_nodes"] == 1 and r["status"] == 1
        s = summ[e]
        assert s["status"] == r["status"] and s["n_nodes"] == 1 and s["iters_run"] == 50 and s["rng_after"] == r["rng_after"]
    assert ctx.check_collision([[[0.0, 0.0], [1.0, 1.0]]])[0] == False  # noqa: E712
    assert not orc.check_collision(w, [[0.0, 0.0], [1.0, 1.0]], kind="portable")
    for bad in ([[0.0, 0.0]], [[0.0, 0.0], [1.0, 0.0]]):
        with pytest.raises(_lib.AuvpError):
            ctx.set_world(world["obstacles"], None, bad)


def test_planner_chunked_steer_and_many_subsections(ctx, orc):
    from auv_sim_amd import synth
    from auv_sim_amd._prrt_lib import PlannerBatch
    from oracle import orc_planner as op
    w = synth.make_rect_world(seed=2, n_obstacles=600, size=150.0, start=(12.0, 12.0), goal=(130.0, 120.0), obst_radius=(0.5, 1.5))
    ctx.set_world(obstacles=w["obstacles"])
    E = 6
    starts = np.tile(np.array([12.0, 12.0, 2.9, 0.0]), (E, 1))
    goals = np.tile(w["goal"], (E, 1))
    seeds = np.arange(900, 900 + E, dtype=np.uint64)
    pb = PlannerBatch(ctx, starts, goals, w["rect"], 400, seeds=seeds, freq=80, cell=3, subs=16)
    summ = pb.plan()
    paths = pb.paths(summ)
    for e in range(E):
        r = op.planning(w["obstacles"], w["rect"], starts[e], goals[e], int(seeds[e]), 400, 80, 3, 16, kind="portable")
        s = summ[e]
        assert s["status"] == r["status"] == 0 and s["steps"] == r["steps"] and bool(s["done"]) == r["done"]
        t = pb.tree(e, s)
        assert np.array_equal(t["parent"], r["parent"]) and np.array_equal(t["nodes"], r["nodes"][:, :4])
        assert np.array_equal(t["node_bucket"], r["node_bucket"]) and s["rng_after"] == r["rng_after"]
        if r["done"]:
            assert np.array_equal(paths[e], r["path"])


def test_astar_without_obstacles(ctx, orc):
    from auv_sim_amd import _astar_lib as al
    from oracle import orc_astar as oa
    ctx.set_world()
    res = al.run_batch(ctx, "astar", [(0.0, 0.0), (30.0, 10.0)], goals=[(90.0, 40.0), (30.0, 10.0)], box=(0, 0, 100, 100),
                       exp_log=True)
    for e, (s, g) in enumerate([((0.0, 0.0), (90.0, 40.0)), ((30.0, 10.0), (30.0, 10.0))]):
        o = oa.run("astar", s, goal=g, box=(0, 0, 100, 100), kind="portable")
        assert res[e]["found"] and o["found"]
        assert np.array_equal(res[e]["path"], o["path"]) and np.array_equal(res[e]["expansions"], o["expansions"])
    assert len(res[1]["path"]) == 1  # start == goal: popped immediately


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_cell_lookup_index_is_exact_for_arbitrary_cell_lists(ctx, orc, seed, monkeypatch):
    """cost.py:181-184 first-match scan (with its `x <= maxy` comparison) over cell lists that are NOT grids:
    overlapping rectangles in random order, degenerate cells, query points exactly on minx / maxx / maxy values.
    The region index (default), the bucket scan (index budget forced to 0) and the CPU checker must agree bit
    for bit."""
    rng = np.random.default_rng(seed)
    C, T = 300, 3
    x0 = np.round(rng.uniform(-60, 60, C), 1)
    y0 = np.round(rng.uniform(-60, 60, C), 1)
    cells = np.stack([x0, y0, x0 + np.round(rng.uniform(0, 40, C), 1), y0 + np.round(rng.uniform(0, 40, C), 1)], axis=1)
    cells[::17, 2] = cells[::17, 0]            # zero-width cells
    cells[5::29, 3] = cells[5::29, 0] - 1.0    # maxy < minx: can never match (x <= maxy fails)
    bins = np.array([[0.0, 50.0], [50.0, 100.0], [100.0, 150.0]])
    prob = rng.uniform(0, 0.3, size=(T, C))
    habitats = np.array([[10.0, 5.0, 12.0], [-30.0, -20.0, 8.0]])
    edges = np.concatenate([cells[:, 0], cells[:, 2], cells[:, 3]])
    paths, los, his, tots, ws = [], [], [], [], []
    for _ in range(40):
        n = int(rng.integers(1, 150))
        x = rng.uniform(-70, 110, n)
        on = rng.random(n) < 0.4
        x[on] = rng.choice(edges, on.sum())    # exactly on a breakpoint of the index
        y = rng.uniform(-70, 110, n)
        yon = rng.random(n) < 0.2
        y[yon] = rng.choice(cells[:, 1], yon.sum())
        t = rng.uniform(-10, 160, n)
        paths.append(np.stack([x, y, t], axis=1))
        lo = int(rng.integers(0, T))
        los.append(lo)
        his.append(int(rng.integers(lo, T + 1)))
        tots.append(float(rng.choice([0.0, 37.5, 200.0])))
        ws.append([-3.0, -3.0, -4.0])
    w = orc.WorldArrays(None, habitats, None, bins, cells, prob)
    want = np.array([orc.cost(w, lo, hi, p, tt, ww, kind="portable") for p, lo, hi, tt, ww in zip(paths, los, his, tots, ws)])
    ctx.set_world(None, habitats, None, bins, cells, prob)
    got_index = ctx.cost_paths(paths, los, his, tots, ws)
    monkeypatch.setenv("AUVP_RG_MAX_ENTRIES", "0")
    ctx.set_world(None, habitats, None, bins, cells, prob)
    got_scan = ctx.cost_paths(paths, los, his, tots, ws)
    monkeypatch.delenv("AUVP_RG_MAX_ENTRIES")
    assert np.array_equal(got_index, want)
    assert np.array_equal(got_scan, want)
    assert (want[:, 3] != 0).sum() > 10


def test_exploring_with_unordered_overlapping_bins(ctx, orc):
    """shark dict whose time bins are neither sorted nor disjoint: a leaf's sub-dict (rrt_dubins.py:160-165) then has
    gaps in dict order, and an element's bin is its first match in that order"""
    from auv_sim_amd import synth
    world = synth.make_world(seed=31, n_obstacles=64, n_bins=6, bin_len=50)
    bins = np.array([[100.0, 150.0], [0.0, 50.0], [30.0, 120.0], [250.0, 300.0], [50.0, 100.0], [150.0, 250.0]])
    prob = world["prob"][:6]
    ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], bins, world["cells"], prob)
    w = orc.WorldArrays(world["obstacles"], world["habitats"], world["polygon"], bins, world["cells"], prob)
    E, n_iter = 6, 900
    init = np.zeros((E, 6))
    init[:, 0], init[:, 1] = world["start"]
    init[:, 3] = [0.0, 10.0, 40.0, 60.0, 0.0, 95.0]   # start times inside different bins
    seeds = np.arange(50, 50 + E, dtype=np.uint64)
    kw = dict(mode="timebin", max_traj_time=260.0)
    summ = ctx.rrt_explore_batch(init, seeds, n_iter, leaf_log=True, **kw)
    n_checked = 0
    for e in range(E):
        r = orc.rrt_explore(w, int(seeds[e]), n_iter, init=init[e], kind="portable", **kw)
        s = summ[e]
        assert s["status"] == r["status"] and s["n_nodes"] == r["n_nodes"] and s["n_leaves"] == r["n_leaves"]
        lc, li = ctx.leaf_log(e, s)
        assert np.array_equal(lc, r["leaf_cost"][:len(lc)]) and np.array_equal(li, r["leaf_iter"][:len(li)])
        if r["status"] == 0:
            assert np.array_equal(np.array(s["best_cost"]), r["best_cost"])
            n_checked += 1
    assert n_checked >= 3


@pytest.mark.parametrize("case", [
    dict(freq=1, n_iter=400),                                            # at most 0 sub-arcs: every node sits on its parent
    dict(freq=2, n_iter=400, max_traj_time=20.0),
    dict(freq=63, n_iter=250, max_traj_time=300.0),                      # one steer chunk of up to 62 sub-arcs
    dict(freq=90, n_iter=200, max_traj_time=300.0),                      # more than one chunk per steer
    dict(freq=30, n_iter=500, weights=(-0.37, -2.25, -1.7)),             # non-integer w2: c1 by repeated addition
    dict(freq=30, n_iter=500, bin_interval=7.5, max_traj_time=130.0),    # fractional bin keys
    dict(freq=30, n_iter=500, v=0.7, dist_to_end=5.0, diff_max=2.0, min_dist=1.5, max_traj_time=400.0),
    dict(freq=12, n_iter=500, mode="nn", max_traj_time=150.0),
    dict(freq=12, n_iter=500, mode="plantime", max_traj_time=150.0),
], ids=lambda c: "-".join("%s%s" % (k[:4], v) for k, v in c.items() if k != "n_iter"))
def test_exploring_parameter_corners(ctx, orc, case):
    """unusual parameter combinations of RRT(...) / exploring(...), each episode equal to its checker run"""
    from auv_sim_amd import synth
    case = dict(case)
    n_iter = case.pop("n_iter")
    world = synth.make_world(seed=41, n_obstacles=64)
    ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    w = orc.WorldArrays(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    E = 5
    init = np.zeros((E, 6))
    init[:, 0], init[:, 1] = world["start"]
    init[:, 2] = np.linspace(-2.5, 2.5, E)
    seeds = np.arange(900, 900 + E, dtype=np.uint64)
    summ = ctx.rrt_explore_batch(init, seeds, n_iter, leaf_log=True, **case)
    for e in range(E):
        r = orc.rrt_explore(w, int(seeds[e]), n_iter, init=init[e], kind="portable", **case)
        s = summ[e]
        assert s["status"] == r["status"], (e, s["status"], r["status"])
        assert (s["n_nodes"], s["n_points"], s["n_leaves"]) == (r["n_nodes"], r["n_points"], r["n_leaves"])
        assert s["rng_after"] == r["rng_after"] and int(s["n_draw32"]) == int(r["n_draw32"])
        t = ctx.tree(e, s)
        assert np.array_equal(t["parent"], r["parent"]) and np.array_equal(t["nodes"], r["nodes"])
        assert np.array_equal(t["points"], r["points"])
        lc, li = ctx.leaf_log(e, s)
        assert np.array_equal(lc, r["leaf_cost"][:len(lc)])
        if r["status"] == 0:
            assert np.array_equal(np.array(s["best_cost"]), r["best_cost"])


@pytest.mark.parametrize("H,V,T", [(64, 64, 64), (0, 3, 1), (33, 17, 5)])
def test_exploring_table_sizes(ctx, orc, H, V, T):
    """the shared LDS tables are sized by the world: the largest accepted world (64 habitats, a 64-vertex boundary,
    64 time bins), the smallest, and an odd one -- every episode equal to its checker run, with and without the logs"""
    from auv_sim_amd import synth
    world = synth.make_world(seed=77, n_obstacles=96, n_bins=T, bin_len=max(1, 320 // T), n_habitats=H)
    x0, y0, x1, y1 = world["box"]
    cxm, cym, rad = 0.5 * (x0 + x1), 0.5 * (y0 + y1), 0.55 * min(x1 - x0, y1 - y0)
    ang = 2.0 * np.pi * np.arange(V) / V
    poly = np.stack([cxm + rad * np.cos(ang), cym + rad * np.sin(ang)], axis=1)  # a V-gon: the safe-box shortcut is off
    ctx.set_world(world["obstacles"], world["habitats"], poly, world["bins"], world["cells"], world["prob"])
    w = orc.WorldArrays(world["obstacles"], world["habitats"], poly, world["bins"], world["cells"], world["prob"])
    E, n_iter = 6, 700
    init = np.zeros((E, 6))
    init[:, 0], init[:, 1] = world["start"]
    seeds = np.arange(300, 300 + E, dtype=np.uint64)
    kw = dict(max_traj_time=150.0)
    for logs in (False, True):
        summ = ctx.rrt_explore_batch(init, seeds, n_iter, leaf_log=logs, **kw)
        for e in range(E):
            r = orc.rrt_explore(w, int(seeds[e]), n_iter, init=init[e], kind="portable", **kw)
            s = summ[e]
            assert (s["status"], s["n_nodes"], s["n_points"], s["n_leaves"]) == (r["status"], r["n_nodes"], r["n_points"], r["n_leaves"])
            assert s["rng_after"] == r["rng_after"]
            t = ctx.tree(e, s)
            assert np.array_equal(t["parent"], r["parent"]) and np.array_equal(t["nodes"], r["nodes"])
            if r["status"] == 0:
                assert np.array_equal(np.array(s["best_cost"]), r["best_cost"])


def test_options_are_per_handle_and_read_the_environment_once(monkeypatch):
    """auvp_set_option / auvp_get_option / auvp_unset_option (include/auvplan.h): unset = the measured default; AUVP_<NAME> in the
    environment is an option's INITIAL value, read by auvp_create and by nothing else; options of one handle do not leak into
    another; an unknown name is AUVP_ERR_ARG"""
    import ctypes as C
    from auv_sim_amd import _lib
    raw = getattr(_lib.load(), "_lib", _lib.load())   # (the suite's environment-sync proxy is bypassed: the plain library)
    monkeypatch.setenv("AUVP_ROWS", "1")
    monkeypatch.delenv("AUVP_TRIO", raising=False)
    a = _lib.Context(0)
    monkeypatch.setenv("AUVP_ROWS", "0")        # after auvp_create: not seen by `a`
    b = _lib.Context(0)

    def get(ctx, name):
        s, v = C.c_int32(-1), C.c_int64(-1)
        assert raw.auvp_get_option(ctx.h, name.encode(), C.byref(s), C.byref(v)) == 0
        return (s.value, v.value)
    assert get(a, "ROWS") == (1, 1) and get(b, "ROWS") == (1, 0) and get(a, "TRIO")[0] == 0
    assert raw.auvp_set_option(a.h, b"TRIO", 1) == 0 and get(a, "TRIO") == (1, 1) and get(b, "TRIO")[0] == 0
    assert raw.auvp_unset_option(a.h, b"TRIO") == 0 and get(a, "TRIO")[0] == 0
    assert raw.auvp_set_option(a.h, b"NO_SUCH_OPTION", 1) == -1 and b"unknown option" in raw.auvp_last_error(a.h)
    assert a.pipeline_fallbacks() == (0, 0)
    a.close()
    b.close()
