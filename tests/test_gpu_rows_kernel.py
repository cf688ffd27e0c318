"""rrt_rows_kernel (four RRT.exploring episodes per wavefront) against rrt_explore_kernel (one per wavefront) and the
CPU checker: every field of every summary, every tree, every path point and every best path must be identical.
AUVP_ROWS=0 forces the one-episode kernel; the default picks the rows kernel whenever its limits allow
(time-bin sampling, freq <= 30, <= 256 obstacles, no iteration log)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from auv_sim_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def _hexagon(cx, cy, r):
    a = np.linspace(0.0, 2.0 * np.pi, 7)[:-1] + 0.3
    return np.column_stack([cx + r * np.cos(a), cy + r * np.sin(a)])


CASES = {
    "o64_default": dict(world=dict(seed=51, n_obstacles=64), E=37, n_iter=900, kw={}),
    "o256_dense_box": dict(world=dict(seed=2, n_obstacles=256), E=21, n_iter=900, kw={}),
    "dense_big_obstacles": dict(world=dict(seed=4, n_obstacles=64, obst_radius=(4.0, 9.0)), E=9, n_iter=700, kw=dict(freq=12)),
    "hexagon_boundary": dict(world=dict(seed=52, n_obstacles=40), E=10, n_iter=800, kw={}, poly="hex"),
    "freq15_one_pass": dict(world=dict(seed=53, n_obstacles=64), E=6, n_iter=700, kw=dict(freq=15)),
    "freq16_two_passes": dict(world=dict(seed=53, n_obstacles=64), E=6, n_iter=700, kw=dict(freq=16)),
    "freq30_long_steps": dict(world=dict(seed=54, n_obstacles=64), E=6, n_iter=600,
                              kw=dict(freq=30, dist_to_end=5.0, diff_max=2.0, min_dist=1.5, v=0.7, max_traj_time=400.0)),
    "freq1": dict(world=dict(seed=55, n_obstacles=16), E=5, n_iter=300, kw=dict(freq=1)),
    "short_horizon_bin_reset": dict(world=dict(seed=3, n_obstacles=64, n_bins=4), E=7, n_iter=800,
                                    kw=dict(max_traj_time=120.0, bin_interval=7.5, weights=(-0.37, -2.25, -1.7))),
    # three bins of ~800 members each: the member lists run past their direct-mapped head into the 64-entry chunks
    "few_bins_chunked_lists": dict(world=dict(seed=59, n_obstacles=32), E=6, n_iter=2500,
                                   kw=dict(max_traj_time=60.0, bin_interval=20.0)),
    # 50 / 20: the last regular key (60) is past max_traj_time and is reset to one member at every insertion
    "few_bins_last_key_reset": dict(world=dict(seed=60, n_obstacles=32), E=5, n_iter=1500,
                                    kw=dict(max_traj_time=50.0, bin_interval=20.0)),
    # a box in positive coordinates: the reference's `x <= maxy` cell test (cost.py:180) leaves most points without a cell,
    # many qualifying leaves cost exactly the same, and the FIRST of them must win (strict `<`, rrt_dubins.py:167)
    "positive_box_exact_cost_ties": dict(world=dict(seed=61, n_obstacles=40, box=(0.0, 0.0, 280.0, 180.0), cell=14.0, n_habitats=0),
                                         E=6, n_iter=1500, kw=dict(max_traj_time=150.0, weights=(3.0, 3.0, 4.0))),
    "positive_box_ties_with_habitats": dict(world=dict(seed=62, n_obstacles=40, box=(0.0, 0.0, 280.0, 180.0), cell=14.0,
                                                       hab_radius=(20.0, 40.0)),
                                            E=6, n_iter=1500, kw=dict(max_traj_time=150.0)),
    # a product grid with uneven column widths and row heights: the separable cell index starts from wrong guesses
    "uneven_product_grid": dict(world=dict(seed=64, n_obstacles=48), E=6, n_iter=900, kw={}, uneven=True),
    "one_episode": dict(world=dict(seed=56, n_obstacles=64), E=1, n_iter=1200, kw={}),
    "no_habitats_no_grid": dict(world=dict(seed=57, n_obstacles=30, n_habitats=0), E=5, n_iter=500, kw={}, strip_grid=True),
    "point_capacity_overflow": dict(world=dict(seed=58, n_obstacles=8), E=6, n_iter=400, kw=dict(points_per_iter=3.0)),
}


@pytest.mark.parametrize("name", list(CASES))
def test_rows_kernel_equals_one_episode_kernel_and_checker(ctx, orc, name, monkeypatch):
    from auv_sim_amd import synth
    c = CASES[name]
    world = synth.make_world(**c["world"])
    poly = world["polygon"]
    if c.get("poly") == "hex":
        x0, y0, x1, y1 = world["box"]
        poly = _hexagon(0.5 * (x0 + x1), 0.5 * (y0 + y1), 0.55 * (x1 - x0))
    if c.get("uneven"):
        from conftest import uneven_grid
        world = uneven_grid(world, 78)
    bins, cells, prob = world["bins"], world["cells"], world["prob"]
    if c.get("strip_grid"):
        bins, cells, prob = None, None, None
    ctx.set_world(world["obstacles"], world["habitats"], poly, bins, cells, prob)
    E, n_iter, kw = c["E"], c["n_iter"], dict(c["kw"])
    init = np.zeros((E, 6))
    init[:, 0], init[:, 1] = world["start"]
    init[:, 2] = np.linspace(-3.0, 3.0, E)
    seeds = np.arange(4000, 4000 + E, dtype=np.uint64)
    out = {}
    for rows in ("1", "0"):
        monkeypatch.setenv("AUVP_ROWS", rows)
        summ = ctx.rrt_explore_batch(init, seeds, n_iter, **kw).copy()
        assert ctx.last_launch_parts()[2] == (4 if rows == "1" else 1)
        trees = [ctx.tree(e, summ[e]) for e in range(E)]
        paths = ctx.paths(summ)
        out[rows] = (summ, trees, paths)
    sa, ta, pa = out["1"]
    sb, tb, pb = out["0"]
    failed = sa["status"] < 0
    for f in sa.dtype.names:
        # an episode that stopped with a capacity error reports where it stopped; the position of its random stream at
        # that moment depends on how the kernel cuts a steer into passes and is not part of the contract
        keep = ~failed if f in ("rng_after", "n_draw32") else np.ones(E, bool)
        if f == "n_candidates":
            # a diagnostic (obstacles that passed the conservative cull): the four-episode kernel culls dense worlds with the
            # tight box of the path points, the one-episode kernel with the reach square -- never fewer candidates there
            assert (sa[f] <= sb[f]).all()
            continue
        assert np.array_equal(sa[f][keep], sb[f][keep]), f
    for e in range(E):
        for k in ("nodes", "parent", "pt_off", "pt_cnt", "points"):
            assert np.array_equal(ta[e][k], tb[e][k]), (e, k)
        assert np.array_equal(pa[e], pb[e])
    if name == "point_capacity_overflow":
        assert (sa["status"] == -2).any()   # the capacity error is reported, identically, by both kernels
        return
    okw = {k: v for k, v in kw.items() if k != "points_per_iter"}
    w = orc.WorldArrays(world["obstacles"], world["habitats"], poly, bins, cells, prob)
    for e in range(min(E, 8)):
        r = orc.rrt_explore(w, int(seeds[e]), n_iter, init=init[e], kind="portable", **okw)
        s = sa[e]
        assert (s["status"], s["n_nodes"], s["n_points"], s["n_leaves"]) == (r["status"], r["n_nodes"], r["n_points"], r["n_leaves"])
        assert s["rng_after"] == r["rng_after"] and int(s["n_draw32"]) == int(r["n_draw32"])
        assert np.array_equal(ta[e]["parent"], r["parent"]) and np.array_equal(ta[e]["nodes"], r["nodes"])
        assert np.array_equal(ta[e]["points"], r["points"])
        if r["status"] == 0:
            assert np.array_equal(np.array(s["best_cost"]), r["best_cost"]) and np.array_equal(pa[e], r["path"])


def test_rows_kernel_continues_a_global_random_state(ctx, orc, monkeypatch):
    """E = 1 with a generator handed over mid-block (random.getstate()): what RRT.exploring(seed=None) does"""
    import random
    from auv_sim_amd import synth
    world = synth.make_world(seed=59, n_obstacles=64)
    ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    rnd = random.Random(123)
    for _ in range(77):
        rnd.random()
    st = rnd.getstate()[1]
    words, idx = np.array(st[:624], dtype=np.uint32).reshape(1, 624), np.array([st[624]], dtype=np.int32)
    init = np.zeros((1, 6))
    init[0, 0], init[0, 1] = world["start"]
    res = {}
    for rows in ("1", "0"):
        monkeypatch.setenv("AUVP_ROWS", rows)
        res[rows] = ctx.rrt_explore_batch(init, (words, idx), 700).copy()
    for f in res["1"].dtype.names:
        if f != "n_candidates":  # (cull diagnostic: depends on the kernel's cull box, see above)
            assert np.array_equal(res["1"][f], res["0"][f]), f
    n = int(res["1"][0]["n_draw32"])
    rnd.getrandbits(32 * n)
    assert rnd.random() == float(res["1"][0]["rng_after"])


def test_leaf_pass_without_pruning_gives_the_same_answers(ctx, orc, monkeypatch):
    """trees too large for the ancestor bit set are swept whole by rrt_leaf_kernel; AUVP_LEAF_SWEEP_ALL forces that path"""
    from auv_sim_amd import synth
    world = synth.make_world(seed=63, n_obstacles=64)
    ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    E, n_iter = 9, 1200
    init = np.zeros((E, 6))
    init[:, 0], init[:, 1] = world["start"]
    seeds = np.arange(7000, 7000 + E, dtype=np.uint64)
    a = ctx.rrt_explore_batch(init, seeds, n_iter, leaf_log=True).copy()
    la = [ctx.leaf_log(e, a[e]) for e in range(E)]
    pa = ctx.paths(a)
    monkeypatch.setenv("AUVP_LEAF_SWEEP_ALL", "1")
    b = ctx.rrt_explore_batch(init, seeds, n_iter, leaf_log=True).copy()
    lb = [ctx.leaf_log(e, b[e]) for e in range(E)]
    pb = ctx.paths(b)
    for f in a.dtype.names:
        assert np.array_equal(a[f], b[f]), f
    for e in range(E):
        assert np.array_equal(pa[e], pb[e])
        assert np.array_equal(la[e][0], lb[e][0]) and np.array_equal(la[e][1], lb[e][1])  # per-leaf costs, creation iteration
    w = orc.WorldArrays(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    for e in range(3):
        r = orc.rrt_explore(w, int(seeds[e]), n_iter, init=init[e], kind="portable")
        assert (b[e]["status"], b[e]["n_leaves"], b[e]["best_leaf"]) == (r["status"], r["n_leaves"], r["best_leaf"])
        if r["status"] == 0:
            assert np.array_equal(np.array(b[e]["best_cost"]), r["best_cost"]) and np.array_equal(pb[e], r["path"])
