"""get_closest_mps (path_planning/rrt_dubins.py:505-513) as the planner's streaming scan: the FIRST node with the
smallest RN(sqrt(dx**2 + dy**2)).  The scan ranks by the squared distance and falls back to a sqrt per node only when two
different squared distances could round to the same root; both paths are pinned here against the reference's rule
evaluated with numpy (IEEE fp64: x**2 = x*x exactly rounded, correctly rounded sqrt), through the C-ABI probe
auvp_nn_closest_batch, and the whole exploring loop in nearest-neighbour mode against the checker."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from auv_sim_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def reference_closest(xy, q):
    """the reference loop: strict <, first wins"""
    dx, dy = q[0] - xy[:, 0], q[1] - xy[:, 1]
    return int(np.argmin(np.sqrt(dx * dx + dy * dy)))  # argmin returns the first minimum


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 511, 512, 513, 1025, 5000, 10001])
def test_random_tables(ctx, n):
    rng = np.random.default_rng(n)
    xy = rng.uniform(-1000, 1000, size=(n, 2))
    q = rng.uniform(-1000, 1000, size=(257, 2))
    want = np.array([reference_closest(xy, qq) for qq in q])
    got, slow = ctx.nn_closest(xy, q)
    assert np.array_equal(got, want) and not slow.any()
    got2, slow2 = ctx.nn_closest(xy, q, force_exact=True)
    assert np.array_equal(got2, want) and slow2.all()


def test_duplicated_positions_first_index_wins(ctx):
    """a steer with zero sub-arcs copies its parent's position (rrt_dubins.py:258-262 with n = 0): the earlier node wins"""
    rng = np.random.default_rng(5)
    base = rng.uniform(-100, 100, size=(700, 2))
    src = rng.integers(0, 700, size=900)
    xy = np.concatenate([base, base[src]])          # every copy comes after its original
    perm_tail = rng.permutation(900)
    xy[700:] = xy[700:][perm_tail]
    q = np.concatenate([base[rng.integers(0, 700, 200)] + rng.normal(0, 1e-3, (200, 2)), rng.uniform(-100, 100, (100, 2))])
    want = np.array([reference_closest(xy, qq) for qq in q])
    got, slow = ctx.nn_closest(xy, q)
    assert np.array_equal(got, want)
    assert (want < 700).all() and not slow.any()    # exact ties are settled without the fallback
    # the same position three times in ONE lane's subsequence (indices 5, 69, 133) and across lanes
    xy2 = rng.uniform(-50, 50, size=(300, 2))
    xy2[69] = xy2[5]; xy2[133] = xy2[5]; xy2[70] = xy2[5]
    got, _ = ctx.nn_closest(xy2, xy2[5][None] + 1e-9)
    assert got[0] == 5


def _near_tie_cases(rng, n_cases):
    """pairs of points whose squared distances from the origin differ by one ulp while their rounded roots agree"""
    out = []
    while len(out) < n_cases:
        s = rng.uniform(1.5, 2.0)
        a = np.array([s, 0.0])
        v1 = s * s
        y = np.sqrt(np.spacing(v1))
        b = np.array([s, y])
        v2 = s * s + y * y
        if v2 != v1 and np.sqrt(v2) == np.sqrt(v1):
            out.append((a, b))
    return out


def test_near_ties_take_the_reference_path(ctx):
    """different d2, same RN(sqrt(d2)): the reference keeps the FIRST of them even if its d2 is the larger one"""
    rng = np.random.default_rng(11)
    for a, b in _near_tie_cases(rng, 12):
        for n_fill, pos_b, pos_a in ((0, 0, 1), (200, 17, 150), (1500, 1300, 40), (600, 64 + 9, 9)):
            n = max(n_fill, 2)
            xy = rng.uniform(5, 50, size=(n, 2))     # far from the origin
            xy[pos_b], xy[pos_a] = b, a               # b has the larger d2
            q = np.zeros((1, 2))
            want = reference_closest(xy, q[0])
            assert want == min(pos_a, pos_b)
            got, slow = ctx.nn_closest(xy, q)
            assert got[0] == want and slow[0]
            got, _ = ctx.nn_closest(xy, q, force_exact=True)
            assert got[0] == want


def test_near_tie_before_duplicated_minimum_in_one_lane(ctx):
    """positions b, a, a at indices i, i + 64, i + 128 -- ONE lane's subsequence of the scan -- with d2(b) one ulp above d2(a)
    and equal rounded roots: the lane ends with its smallest value held twice, and the near-tie must still be seen (the second
    smallest DISTINCT value is what the lane tracks).  The reference keeps index i (strict < on the rounded root)."""
    rng = np.random.default_rng(13)
    for a, b in _near_tie_cases(rng, 8):
        for n, i in ((200, 5), (700, 3), (1400, 512 + 60), (130, 1)):
            for order in ((b, a, a), (a, b, a), (a, a, b), (b, a, a, a)):
                xy = rng.uniform(5, 50, size=(max(n, i + 64 * len(order) + 1), 2))
                for k, p in enumerate(order):
                    xy[i + 64 * k] = p
                q = np.zeros((1, 2))
                want = reference_closest(xy, q[0])
                assert want == i
                got, slow = ctx.nn_closest(xy, q)
                assert got[0] == want and slow[0], (n, i, len(order))
                got, _ = ctx.nn_closest(xy, q, force_exact=True)
                assert got[0] == want
    # the same duplicated minimum WITHOUT a near-tie stays on the fast path
    xy = rng.uniform(5, 50, size=(300, 2))
    xy[7] = xy[71] = xy[135] = np.array([1.75, 0.0])
    got, slow = ctx.nn_closest(xy, np.zeros((1, 2)))
    assert got[0] == 7 and not slow[0]


def test_exploring_nn_forced_exact_equals_scan(ctx, orc, monkeypatch):
    """the exploring loop in nearest-neighbour mode: scan path == forced-exact path == checker, bit for bit"""
    from auv_sim_amd import synth
    world = synth.make_world(seed=4, n_obstacles=64)
    ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    E, n_iter = 12, 1500
    init = np.zeros((E, 6))
    init[:, 0], init[:, 1] = world["start"]
    seeds = np.arange(E, dtype=np.uint64) + 100
    a = ctx.rrt_explore_batch(init, seeds, n_iter, mode="nn").copy()
    ta = [ctx.tree(e, a[e]) for e in range(E)]
    monkeypatch.setenv("AUVP_NN_EXACT", "1")
    b = ctx.rrt_explore_batch(init, seeds, n_iter, mode="nn").copy()
    monkeypatch.delenv("AUVP_NN_EXACT")
    for n in a.dtype.names:
        assert np.array_equal(a[n], b[n]), n
    w = orc.WorldArrays(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    for e in range(E):
        r = orc.rrt_explore(w, int(seeds[e]), n_iter, mode="nn", init=init[e], kind="portable")
        assert (a[e]["n_nodes"], a[e]["n_points"], a[e]["rng_after"]) == (r["n_nodes"], r["n_points"], r["rng_after"])
        assert np.array_equal(ta[e]["parent"], r["parent"]) and np.array_equal(ta[e]["nodes"], r["nodes"])
        # len(mps_list) summed over the scans: the root is there for every iteration, node m from the iteration after its own
        plan_it = ta[e]["nodes"][1:, 4]
        assert int(a[e]["nn_scanned"]) == int(a[e]["iters_run"] + (a[e]["iters_run"] - 1 - plan_it).sum())
