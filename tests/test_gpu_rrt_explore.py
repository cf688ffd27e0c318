"""GPU parity tests for RRT.exploring and its building blocks -- all through the C-ABI
(libauvplan.so via auv_sim_amd._lib).  Run on the MI355X box: pytest -m gpu.

Comparison chain (DESIGN.md "Numerics"):
  HIP kernel  ==  oracle(portable math)     bit-for-bit, every float and every index
  HIP kernel  ~=  golden (reference)        decisions exact, floats <= 1e-9, path cost <= 1e-6
"""
import glob
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_world

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from auv_sim_amd import _lib
    c = _lib.Context(0)  # raises when the HIP extension or the GPU is missing: no fallback
    yield c
    c.close()


def test_device_random_stream(ctx):
    import random
    for seed in (0, 7, 1234, 2 ** 32 + 5, 2 ** 63 + 12345):
        r = random.Random(seed)
        want = np.array([r.random() for _ in range(5000)])
        got = ctx.random_stream(seed, 5000)
        assert np.array_equal(got, want)


def test_device_sincos_bit_exact(ctx, orc):
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-60, 60, 200000), rng.uniform(-1e6, 1e6, 20000), rng.uniform(-1e-3, 1e-3, 2000),
                        np.array([0.0, np.pi / 4, -np.pi / 4, np.pi / 2, np.pi, 3 * np.pi / 4, 1e9, -1e9])])
    s, c = ctx.sincos(x)
    L = orc.lib("portable")
    ws = np.array([L.orc_sin(float(v)) for v in x[:20000]])
    wc = np.array([L.orc_cos(float(v)) for v in x[:20000]])
    assert np.array_equal(s[:20000], ws) and np.array_equal(c[:20000], wc)
    # and within 1 ulp of libm everywhere in the working range
    m = np.abs(x) < 1e6
    assert np.max(np.abs(s[m] - np.sin(x[m])) / np.spacing(np.abs(np.sin(x[m])) + 1e-300)) <= 1.0
    assert np.max(np.abs(c[m] - np.cos(x[m])) / np.spacing(np.abs(np.cos(x[m])) + 1e-300)) <= 1.0


def test_device_math_bit_exact(ctx, orc):
    """auvp_math_dev: the device build of auvp_math.h / auvp_exp.h against the host build of the same headers (the portable
    checker library) and, for the short division / square root of the steer (auvp_div_plain / auvp_sqrt_plain:
    rrt_dubins.py:268-281), against the IEEE operation on operands across the magnitudes a steer can produce"""
    rng = np.random.default_rng(11)
    n = 400000
    # the steer's quotients: 2 dist / (-2 diff), 2 dist / (2 radius), movement / velocity -- and generic pairs over 2^+-200
    dist, diff = rng.uniform(0, 2, n), rng.uniform(-0.5, 0.5, n)
    diff[:48] = 2.0 ** -np.arange(1, 49) * 0.25
    diff[48:96] = -(2.0 ** -np.arange(1, 49)) * 0.25
    s1, s2 = dist + diff, dist - diff
    ok = (-s1 + s2) != 0.0  # (a zero divisor is the reference's ZeroDivisionError: not a value to compare)
    dist, diff, s1, s2 = dist[ok], diff[ok], s1[ok], s2[ok]
    n = len(dist)
    radius = (s1 + s2) / (-s1 + s2)
    mv, vt = rng.uniform(0, 2, n), rng.uniform(0, 4, n) + 2.0 ** -53
    mv[:1000] = 0.0
    ga = rng.uniform(1, 2, n) * 2.0 ** rng.integers(-200, 200, n) * rng.choice([-1.0, 1.0], n)
    gb = rng.uniform(1, 2, n) * 2.0 ** rng.integers(-200, 200, n) * rng.choice([-1.0, 1.0], n)
    for a, b in ((s1 + s2, -s1 + s2), (s1 + s2, 2 * radius), (mv, vt), (ga, gb)):
        with np.errstate(all="ignore"):
            want = a / b
        got = ctx.math("div_plain", a, b)
        bad = np.flatnonzero(got != want)
        assert len(bad) == 0, (len(bad), a[bad[:4]], b[bad[:4]], got[bad[:4]], want[bad[:4]])
        assert np.array_equal(ctx.math("div", a, b), want)
    x = np.concatenate([rng.uniform(0, 8, n), rng.uniform(1, 2, n) * 2.0 ** rng.integers(-400, 400, n), np.zeros(8),
                        rng.uniform(0, 1, 1000) ** 8 * 1e-30])
    assert np.array_equal(ctx.math("sqrt_plain", x), np.sqrt(x))
    assert np.array_equal(ctx.math("sqrt", x), np.sqrt(x))
    assert not np.signbit(ctx.math("sqrt_plain", np.zeros(4))).any()
    # the elementary functions: device == host build of the same header, bit for bit
    L = orc.lib("portable")
    m = 20000
    ya, xa = rng.uniform(-300, 300, m), rng.uniform(-300, 300, m)
    ya[:200] = rng.uniform(-1e-9, 1e-9, 200); xa[200:400] = rng.uniform(-1e-9, 1e-9, 200); ya[400:420] = 0.0; xa[420:440] = 0.0
    assert np.array_equal(ctx.math("atan2", ya, xa), np.array([L.orc_atan2(float(p), float(q)) for p, q in zip(ya, xa)]))
    assert np.array_equal(ctx.math("hypot", ya, xa), np.array([L.orc_hypot(float(p), float(q)) for p, q in zip(ya, xa)]))
    from oracle import orc_pf
    z = np.concatenate([rng.uniform(-60, 5, m), -rng.uniform(0, 1, 2000) ** 4 * 700, np.array([0.0, -0.0, 1.0, -745.0, 709.0])])
    assert np.array_equal(ctx.math("pow_e", z), orc_pf.pow_e(z, "portable"))
    xs = rng.uniform(-60, 60, m)
    s, c = ctx.math("sincos", xs)
    assert np.array_equal(s, np.array([L.orc_sin(float(v)) for v in xs])) and np.array_equal(c, np.array([L.orc_cos(float(v)) for v in xs]))


def test_collision_golden(ctx):
    g = json.load(open(os.path.join(GOLDEN, "g5_collision.json")))
    for poly, cases in ((g["rect"], g["cases"]), (g["penta"], g["penta_cases"])):
        # the world (obstacle list) differs per case: group cases by obstacle list
        for c in cases:
            ctx.set_world(obstacles=c["obs"], polygon=poly)
            assert bool(ctx.check_collision([c["pts"]])[0]) == c["free"]


def test_collision_random_vs_oracle(ctx, orc):
    rng = np.random.default_rng(5)
    for trial in range(6):
        O = [0, 1, 63, 64, 65, 300][trial]
        obs = np.column_stack([rng.uniform(0, 200, O), rng.uniform(0, 200, O), rng.uniform(0.5, 12, O)])
        poly = [[0, 0], [200, 0], [200, 200], [0, 200]]
        ctx.set_world(obstacles=obs, polygon=poly)
        w = orc.WorldArrays(obstacles=obs, polygon=poly)
        paths = [rng.uniform(-5, 205, (rng.integers(1, 40), 2)) for _ in range(300)]
        got = ctx.check_collision(paths)
        want = np.array([orc.check_collision(w, p, kind="portable") for p in paths])
        assert np.array_equal(got, want)


def test_cost_golden(ctx, orc):
    from auv_sim_amd import synth
    g = json.load(open(os.path.join(GOLDEN, "g4_cost.json")))
    for c in g["cases"]:
        world = synth.make_world(seed=c["world_seed"], n_obstacles=4, n_habitats=c["n_habitats"], cell=c["cell"],
                                 n_bins=c["n_bins"])
        ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
        out = ctx.cost_paths([c["pts"]], [c["bin_lo"]], [c["bin_hi"]], [c["total"]], [c["weights"]])[0]
        w = orc.WorldArrays(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
        want = orc.cost(w, c["bin_lo"], c["bin_hi"], c["pts"], c["total"], c["weights"], kind="portable")
        assert np.array_equal(out, want)                       # bit-exact vs the checker
        np.testing.assert_allclose(out, c["out"], rtol=1e-12, atol=1e-6)  # vs the reference


def test_cost_irregular_cells_vs_oracle(ctx, orc):
    """cell lists that are not a regular grid: the x-bucket index must still give the first match"""
    rng = np.random.default_rng(9)
    for trial in range(4):
        C = [1, 7, 200, 1500][trial]
        x0 = rng.uniform(-50, 50, C)
        y0 = rng.uniform(-50, 50, C)
        cells = np.column_stack([x0, y0, x0 + rng.uniform(0, 30, C), y0 + rng.uniform(0, 30, C)])
        bins = np.array([[0.0, 50.0], [50.0, 100.0], [100.0, 150.0]])
        prob = rng.uniform(0, 1, (3, C))
        hab = np.column_stack([rng.uniform(-50, 50, 5), rng.uniform(-50, 50, 5), rng.uniform(2, 20, 5)])
        ctx.set_world(habitats=hab, bins=bins, cells=cells, prob=prob)
        w = orc.WorldArrays(habitats=hab, bins=bins, cells=cells, prob=prob)
        paths = [np.column_stack([rng.uniform(-70, 90, n), rng.uniform(-70, 90, n), rng.uniform(-10, 160, n)])
                 for n in rng.integers(1, 200, 40)]
        lo = rng.integers(0, 2, 40)
        hi = lo + rng.integers(0, 3, 40)
        hi = np.minimum(hi, 3)
        tot = rng.uniform(0, 300, 40)
        wts = np.tile([-3.0, -0.1, -4.5], (40, 1))
        out = ctx.cost_paths(paths, lo, hi, tot, wts)
        for i in range(40):
            want = orc.cost(w, int(lo[i]), int(hi[i]), paths[i], tot[i], wts[i], kind="portable")
            assert np.array_equal(out[i], want), (trial, i)


G3 = sorted(glob.glob(os.path.join(GOLDEN, "g3_*.npz")))


def _args(g):
    return dict(mode=str(g["mode"]), freq=int(g["freq"]), bin_interval=int(g["bin_interval"]), v=int(g["v"]),
                max_traj_time=float(g["max_traj_time"]), weights=g["weights"], dist_to_end=float(g["dist_to_end"]),
                diff_max=float(g["diff_max"]))


@pytest.mark.parametrize("path", G3, ids=[os.path.basename(p)[:-4] for p in G3])
def test_exploring_vs_golden_and_oracle(ctx, orc, path):
    g = np.load(path)
    gw = golden_world(g)
    ctx.set_world(gw["obstacles"], gw["habitats"], gw["polygon"], gw["bins"], gw["cells"], gw["prob"])
    init = np.array([[g["start"][0], g["start"][1], 0, 0, 0, 0]], dtype=np.float64)
    n_iter = int(g["n_iter"])
    summ = ctx.rrt_explore_batch(init, [int(g["seed"])], n_iter, iter_log=True, leaf_log=True, **_args(g))
    s = summ[0]
    assert s["status"] == 0 and s["iters_run"] == n_iter
    t = ctx.tree(0, s)
    il = ctx.iter_log(0)
    lc, li = ctx.leaf_log(0, s)
    p = ctx.paths(summ)[0]
    # ---- against the reference's golden vectors: decisions exact, floats to 1e-9, cost to 1e-6
    assert s["n_nodes"] == len(g["nodes"])
    assert np.array_equal(t["parent"], g["parent"])
    assert np.array_equal(t["pt_cnt"][1:] + 1, g["npath"][1:])
    ran = il["it_parent"] >= 0
    assert np.array_equal(il["it_parent"][ran], g["it_parent"])
    assert np.array_equal(il["it_accepted"][ran], g["it_accepted"])
    assert np.array_equal(il["it_npath"][ran], g["it_npath"])
    assert s["rng_after"] == float(g["rng_after"])
    assert np.array_equal(np.cumsum(ran)[li] - 1, g["leaf_iter"])
    np.testing.assert_allclose(t["nodes"], g["nodes"], rtol=1e-9, atol=1e-9)
    if "points" in g.files:
        np.testing.assert_allclose(t["points"], g["points"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(lc, g["leaf_cost"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(np.array(s["best_cost"]), g["res_cost"], rtol=0, atol=1e-6)
    assert abs(s["best_length"] - float(g["res_path_length"])) <= 1e-9 * max(1.0, abs(float(g["res_path_length"])))
    assert p.shape == g["res_path"].shape
    np.testing.assert_allclose(p, g["res_path"], rtol=1e-9, atol=1e-9)
    if "bin_sizes" in g.files:
        bs = ctx.bin_sizes(0)
        assert np.array_equal(bs, g["bin_sizes"][:len(bs)])
    # ---- against the CPU checker built with the same portable math: bit-for-bit
    w = orc.WorldArrays(gw["obstacles"], gw["habitats"], gw["polygon"], gw["bins"], gw["cells"], gw["prob"])
    r = orc.rrt_explore(w, int(g["seed"]), n_iter, init=init[0], kind="portable", **_args(g))
    assert np.array_equal(t["nodes"], r["nodes"])
    assert np.array_equal(t["points"], r["points"])
    assert np.array_equal(t["pt_off"], r["pt_off"])
    assert np.array_equal(il["it_parent"], r["it_parent"])
    assert np.array_equal(il["it_accepted"], r["it_accepted"])
    assert np.array_equal(lc, r["leaf_cost"]) and np.array_equal(li, r["leaf_iter"])
    assert np.array_equal(np.array(s["best_cost"]), r["best_cost"])
    assert s["best_leaf"] == r["best_leaf"] and s["best_length"] == r["best_length"]
    assert np.array_equal(p, r["path"])


@pytest.mark.parametrize("mode,O", [("timebin", 64), ("timebin", 256), ("nn", 64), ("plantime", 64)])
def test_exploring_batch_vs_oracle(ctx, orc, mode, O):
    """many independent episodes in one launch: every episode equals its own checker run"""
    from auv_sim_amd import synth
    world = synth.make_world(seed=21, n_obstacles=O)
    ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    w = orc.WorldArrays(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    E, n_iter = 37, 700
    init = np.zeros((E, 6))
    init[:, 0], init[:, 1] = world["start"]
    init[:, 2] = np.linspace(-3, 3, E)
    seeds = np.arange(1000, 1000 + E, dtype=np.uint64)
    summ = ctx.rrt_explore_batch(init, seeds, n_iter, mode=mode)
    paths = ctx.paths(summ)
    for e in range(E):
        r = orc.rrt_explore(w, int(seeds[e]), n_iter, mode=mode, init=init[e], kind="portable")
        s = summ[e]
        assert s["status"] == r["status"], (e, s["status"], r["status"])
        assert s["n_nodes"] == r["n_nodes"] and s["n_points"] == r["n_points"] and s["n_leaves"] == r["n_leaves"]
        assert s["rng_after"] == r["rng_after"]
        t = ctx.tree(e, s)
        assert np.array_equal(t["parent"], r["parent"])
        assert np.array_equal(t["nodes"], r["nodes"])
        assert np.array_equal(t["points"], r["points"])
        if r["status"] == 0:
            assert np.array_equal(np.array(s["best_cost"]), r["best_cost"])
            assert np.array_equal(paths[e], r["path"])


def test_exploring_no_qualifying_leaf(ctx):
    from auv_sim_amd import synth, _lib
    world = synth.make_world(seed=1, n_obstacles=64)
    ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    init = np.array([[world["start"][0], world["start"][1], 0, 0, 0, 0]])
    summ = ctx.rrt_explore_batch(init, [1], 5)
    assert summ[0]["status"] == _lib.NO_QUALIFYING_LEAF and summ[0]["best_leaf"] == -1
