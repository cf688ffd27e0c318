"""rrt_duo_kernel (two wavefronts per episode: a helper produces the half of an iteration that depends only on the random
stream one iteration ahead, rrt_duo_kernel.h) and rrt_trio_kernel (three: stream, geometry, tree -- a pipeline over the
iterations with speculative stages, rrt_trio_kernel.h) against rrt_explore_kernel and the checker, bit for bit: summaries,
trees, path points, bin sizes, returned paths -- including the stream position the episode ends at."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = {
    "sparse_o64": dict(world=dict(seed=3, n_obstacles=64), E=9, n_iter=1500, kw=dict()),
    "dense_o256_tight_cull": dict(world=dict(seed=2, n_obstacles=256), E=6, n_iter=1200, kw=dict()),
    "one_episode": dict(world=dict(seed=5, n_obstacles=64), E=1, n_iter=3000, kw=dict()),
    "freq8_short_bins": dict(world=dict(seed=7, n_obstacles=128), E=5, n_iter=900, kw=dict(freq=8, bin_interval=2.5, max_traj_time=90.0)),
    "few_bins": dict(world=dict(seed=11, n_obstacles=32), E=4, n_iter=700, kw=dict(bin_interval=50.0, max_traj_time=200.0)),
    "many_episodes": dict(world=dict(seed=13, n_obstacles=64), E=300, n_iter=400, kw=dict()),
    "tiny_budgets": dict(world=dict(seed=17, n_obstacles=64), E=3, n_iter=2, kw=dict()),
}


@pytest.fixture(scope="module")
def ctx():
    from auv_sim_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("kernel", ["duo", "trio", "quad"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_helper_wavefronts_equal_one_wavefront_per_episode(ctx, orc, monkeypatch, name, kernel):
    from auv_sim_amd import synth
    c = CASES[name]
    world = synth.make_world(**c["world"])
    ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    E, n_iter, kw = c["E"], c["n_iter"], dict(c["kw"])
    init = np.zeros((E, 6))
    init[:, 0], init[:, 1] = world["start"]
    init[:, 2] = np.linspace(-3.0, 3.0, E)
    seeds = np.arange(900, 900 + E, dtype=np.uint64)
    out = {}
    monkeypatch.setenv("AUVP_ROWS", "0")
    for duo in ("1", "0"):
        monkeypatch.setenv("AUVP_DUO", duo if kernel == "duo" else "0")
        monkeypatch.setenv("AUVP_TRIO", duo if kernel in ("trio", "quad") else "0")
        monkeypatch.setenv("AUVP_QUAD", "1" if kernel == "quad" else "0")
        summ = ctx.rrt_explore_batch(init, seeds, n_iter, **kw).copy()
        assert ctx.last_rrt_kernel().startswith("rrt_%s_kernel" % ("trio" if kernel == "quad" else kernel) if duo == "1" else "rrt_explore_kernel")
        sample = range(E) if E <= 16 else range(0, E, 37)
        trees = {e: ctx.tree(e, summ[e]) for e in sample}
        bins = {e: ctx.bin_sizes(e) for e in sample} if hasattr(ctx, "bin_sizes") else {}
        out[duo] = (summ, trees, ctx.paths(summ), bins)
    sa, ta, pa, ba = out["1"]
    sb, tb, pb, bb = out["0"]
    assert (sa["status"] >= 0).all(), np.unique(sa["status"])
    for f in sa.dtype.names:
        assert np.array_equal(sa[f], sb[f]), f
    for e in ta:
        for k in ("nodes", "parent", "pt_off", "pt_cnt", "points"):
            assert np.array_equal(ta[e][k], tb[e][k]), (e, k)
        if ba:
            assert np.array_equal(ba[e], bb[e])
    for e in range(E):
        assert np.array_equal(pa[e], pb[e])
    w = orc.WorldArrays(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    for e in list(ta)[:3]:
        r = orc.rrt_explore(w, int(seeds[e]), n_iter, init=init[e], kind="portable", **kw)
        s = sa[e]
        assert (s["status"], s["n_nodes"], s["n_points"], s["n_leaves"]) == (r["status"], r["n_nodes"], r["n_points"], r["n_leaves"])
        assert s["rng_after"] == r["rng_after"] and int(s["n_draw32"]) == int(r["n_draw32"])
        assert np.array_equal(ta[e]["parent"], r["parent"]) and np.array_equal(ta[e]["nodes"], r["nodes"])
