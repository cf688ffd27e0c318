"""rrt_rows_stream_kernel (round 6; option ROWS_STREAM): rrt_rows_kernel with its random() numbers generated AHEAD by a launch of
its own (rrt_stream_kernel: one wavefront per episode) and read from HBM through a 256-entry LDS ring.  Same operations on the
same values: every summary field, tree, path point and best path must equal the classic kernel's (and the checker's); a stream
that turns out too short is a declared status and the batch is redone with the generator inside the kernel.  Without the option
the host takes this path from the second batch with a parameter block on, on whatever world (the stream's length comes from what
earlier batches with those parameters drew)."""
import os
import random
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu  # (the generator check at the end needs no GPU but lives with its kernel)
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    from auv_sim_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def _run(ctx, init, seeds, n_iter, stream, cap=None, **kw):
    ctx.set_option("ROWS", 1)
    ctx.set_option("ROWS_STREAM", stream)
    ctx.set_option("ROWS_STREAM_CAP", cap)
    try:
        summ = ctx.rrt_explore_batch(init, seeds, n_iter, **kw).copy()
        kernel, stream_ms = ctx.last_rrt_kernel(), ctx.last_stream_ms()
        trees = [ctx.tree(e, summ[e]) for e in range(len(summ))]
        paths = ctx.paths(summ)
    finally:
        ctx.set_option("ROWS", None)
        ctx.set_option("ROWS_STREAM", None)
        ctx.set_option("ROWS_STREAM_CAP", None)
    return summ, trees, paths, kernel, stream_ms


def _same(a, b):
    sa, ta, pa = a[:3]
    sb, tb, pb = b[:3]
    for f in sa.dtype.names:
        assert np.array_equal(sa[f], sb[f]), f
    for e in range(len(sa)):
        for k in ("nodes", "parent", "pt_off", "pt_cnt", "points"):
            assert np.array_equal(ta[e][k], tb[e][k]), (e, k)
        assert np.array_equal(pa[e], pb[e])


CASES = {
    "o256_bench_world": dict(world=dict(seed=2, n_obstacles=256, box=(-1000.0, -1000.0, 1000.0, 1000.0), cell=10.0, n_bins=10, bin_len=50, n_habitats=10),
                             E=53, n_iter=1500, kw={}),
    "o64_default": dict(world=dict(seed=51, n_obstacles=64), E=37, n_iter=900, kw={}),
    "freq30_two_full_passes": dict(world=dict(seed=54, n_obstacles=64), E=6, n_iter=600,
                                   kw=dict(freq=30, dist_to_end=5.0, diff_max=2.0, min_dist=1.5, v=0.7, max_traj_time=400.0)),
    # the leaves looked at after EVERY iteration: ~92 numbers per iteration here, twice the stream's default length -- the stream
    # default length (no earlier batch with these parameters on this context: 46.5 per iteration + 4 096), which every episode runs
    # past: the batch redone; then the stream sized for it (option ROWS_STREAM_CAP)
    "freq1_default_length_falls_back": dict(world=dict(seed=55, n_obstacles=16), E=5, n_iter=300, kw=dict(freq=1), falls_back=5),
    "freq1": dict(world=dict(seed=55, n_obstacles=16), E=5, n_iter=300, kw=dict(freq=1), cap=64000),
    "short_horizon_bin_reset": dict(world=dict(seed=3, n_obstacles=64, n_bins=4), E=7, n_iter=800,
                                    kw=dict(max_traj_time=120.0, bin_interval=7.5, weights=(-0.37, -2.25, -1.7))),
    # three bins: most selection rounds find their first bins empty and redraw -- the stream is consumed in bursts of 14
    "few_bins_many_redraws": dict(world=dict(seed=59, n_obstacles=32), E=6, n_iter=2500, kw=dict(max_traj_time=60.0, bin_interval=20.0)),
    "concave_boundary": dict(world=dict(seed=8, n_obstacles=64, polygon="notch"), E=9, n_iter=1200, kw={}),
    "one_episode": dict(world=dict(seed=56, n_obstacles=64), E=1, n_iter=1200, kw={}),
    "point_capacity_overflow": dict(world=dict(seed=58, n_obstacles=8), E=6, n_iter=400, kw=dict(points_per_iter=3.0)),
}


@pytest.mark.parametrize("name", list(CASES))
def test_stream_kernel_equals_the_classic_rows_kernel_and_the_checker(ctx, orc, name):
    from auv_sim_amd import synth
    c = CASES[name]
    world = synth.make_world(**c["world"])
    ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    E, n_iter, kw = c["E"], c["n_iter"], dict(c["kw"])
    init = np.zeros((E, 6))
    init[:, 0], init[:, 1] = world["start"]
    init[:, 2] = np.linspace(-3.0, 3.0, E)
    seeds = np.arange(4000, 4000 + E, dtype=np.uint64)
    a = _run(ctx, init, seeds, n_iter, 1, cap=c.get("cap"), **kw)
    redone = ctx.pipeline_fallbacks()[0]
    b = _run(ctx, init, seeds, n_iter, 0, **kw)
    assert b[3] == "rrt_rows_kernel" and b[4] == 0.0
    if c.get("falls_back"):
        assert a[3] == "rrt_rows_kernel" and redone == c["falls_back"]
    else:
        assert a[3] == "rrt_rows_stream_kernel" and a[4] > 0.0 and redone == 0
    if name == "point_capacity_overflow":
        # (an episode that stopped with a capacity error reports where it stopped: identical here, both kernels cut a steer into
        # the same passes)
        assert (a[0]["status"] == -2).any()
    _same(a, b)
    if name == "point_capacity_overflow":
        return
    w = orc.WorldArrays(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    for e in sorted({0, E // 2, E - 1}):
        r = orc.rrt_explore(w, int(seeds[e]), n_iter, init=init[e], kind="portable", **kw)
        s = a[0][e]
        assert (s["status"], s["n_nodes"], s["n_points"], s["n_leaves"]) == (r["status"], r["n_nodes"], r["n_points"], r["n_leaves"])
        assert s["rng_after"] == r["rng_after"] and int(s["n_draw32"]) == int(r["n_draw32"])
        assert np.array_equal(a[1][e]["nodes"], r["nodes"]) and np.array_equal(a[1][e]["points"], r["points"])


@pytest.mark.parametrize("consumed", [77, 78, 311, 312, 0])   # random() calls before the hand-over: odd / even word positions, mid-cycle, none
def test_stream_of_a_continued_generator(ctx, consumed):
    """the generator handed over mid-cycle (RRT.exploring(seed=None) continuing Python's global stream): rrt_stream_kernel's three
    paths -- the words the state still holds, pairs that are the blocks' pairs, pairs that straddle the lanes (an odd number of
    32-bit outputs consumed: random.getrandbits(32) once)"""
    from auv_sim_amd import synth
    world = synth.make_world(seed=59, n_obstacles=64)
    ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    E = 5
    words, idx = [], []
    for e in range(E):
        rnd = random.Random(900 + e)
        for _ in range(consumed + 3 * e):
            rnd.random()
        if e % 2 == 1:
            rnd.getrandbits(32)          # one 32-bit output: the next random() starts at an odd word of the cycle
        st = rnd.getstate()[1]
        words.append(st[:624])
        idx.append(st[624])
    words, idx = np.array(words, dtype=np.uint32), np.array(idx, dtype=np.int32)
    init = np.zeros((E, 6))
    init[:, 0], init[:, 1] = world["start"]
    a = _run(ctx, init, (words, idx), 700, 1)
    b = _run(ctx, init, (words, idx), 700, 0)
    assert a[3] == "rrt_rows_stream_kernel" and b[3] == "rrt_rows_kernel"
    _same(a, b)


def test_a_stream_that_is_too_short_is_redone_with_the_generator_inside_the_kernel(ctx):
    """option ROWS_STREAM_CAP: 1 024 numbers for 600 iterations (~27 000 needed): every episode runs past its stream (AUVP_ERR_STREAM,
    -10), the host redoes the batch on rrt_rows_kernel inside the same call and counts it; with PIPE_FALLBACK = 0 the status stays"""
    from auv_sim_amd import synth
    world = synth.make_world(seed=51, n_obstacles=64)
    ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    E = 11
    init = np.zeros((E, 6))
    init[:, 0], init[:, 1] = world["start"]
    seeds = np.arange(E, dtype=np.uint64) + 1
    b = _run(ctx, init, seeds, 600, 0)
    try:
        before = ctx.pipeline_fallbacks()[1]
        a = _run(ctx, init, seeds, 600, 1, cap=1024)
        assert a[3] == "rrt_rows_kernel"                       # the kernel that produced the results
        assert ctx.pipeline_fallbacks() == (E, before + E)
        _same(a, b)
        ctx.set_option("PIPE_FALLBACK", 0)
        ctx.set_option("ROWS", 1)
        ctx.set_option("ROWS_STREAM", 1)
        ctx.set_option("ROWS_STREAM_CAP", 1024)
        s = ctx.rrt_explore_batch(init, seeds, 600)
        assert (s["status"] == -10).all() and (s["iters_run"] < 600).all() and ctx.last_rrt_kernel() == "rrt_rows_stream_kernel"
    finally:
        for k in ("ROWS_STREAM_CAP", "PIPE_FALLBACK", "ROWS", "ROWS_STREAM"):
            ctx.set_option(k, None)


def test_random_batches_with_streams_of_every_length(ctx):
    """40 random batches (worlds, sizes, budgets, leaf intervals, continued generators), the stream's length drawn between a
    twentieth and twice what the episodes draw: whether the stream kernel finishes or the batch is redone, the results are the
    classic kernel's"""
    from auv_sim_amd import synth
    rnd = random.Random(20261004)
    n_redone = n_stream = 0
    for case in range(40):
        world = synth.make_world(seed=rnd.randrange(1000), n_obstacles=rnd.choice([0, 8, 64, 200]), n_bins=rnd.choice([4, 10]),
                                 polygon=rnd.choice([None, "notch", "catalina"]))
        ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
        E, n_iter = rnd.randrange(1, 60), rnd.randrange(16, 1200)
        kw = dict(freq=rnd.choice([1, 7, 30, 100]), max_traj_time=rnd.choice([60.0, 200.0, 500.0]), bin_interval=rnd.choice([5, 7.5, 20.0]))
        init = np.zeros((E, 6))
        init[:, 0], init[:, 1] = world["start"]
        init[:, 2] = [rnd.uniform(-3.1, 3.1) for _ in range(E)]
        if case % 3 == 0:
            words, idx = [], []
            for e in range(E):
                g = random.Random(rnd.randrange(1 << 30))
                for _ in range(rnd.randrange(700)):
                    g.random()
                if rnd.random() < 0.5:
                    g.getrandbits(32)
                st = g.getstate()[1]
                words.append(st[:624])
                idx.append(st[624])
            seeds = (np.array(words, dtype=np.uint32), np.array(idx, dtype=np.int32))
        else:
            seeds = np.array([rnd.randrange(1 << 32) for _ in range(E)], dtype=np.uint64)
        b = _run(ctx, init, seeds, n_iter, 0, **kw)
        drawn = int(b[0]["n_draw32"].max()) // 2
        cap = max(64, int(drawn * rnd.choice([0.05, 0.5, 0.98, 1.0, 1.0, 1.02, 2.0]))) if case % 4 else None
        a = _run(ctx, init, seeds, n_iter, 1, cap=cap, **kw)
        redone = ctx.pipeline_fallbacks()[0]
        assert (a[3] == "rrt_rows_kernel") == (redone > 0), (case, a[3], redone)
        n_redone += redone > 0
        n_stream += redone == 0
        _same(a, b)
    assert n_redone >= 5 and n_stream >= 15, (n_redone, n_stream)


def test_without_the_option_the_second_batch_with_a_parameter_block_takes_the_stream():
    """the host's choice (auvplan.hip rrt_run_pass): rrt_rows_kernel for the first batch -- rrt_leaf_kernel reports how many numbers
    its busiest episode drew -- then rrt_stream_kernel + rrt_rows_stream_kernel with a stream of that length + 3 % + 1 024, on
    whatever world (the draw count is the parameters'); back to rrt_rows_kernel when a parameter changes, for a batch more than
    four times the size of the largest observed one, and below 1 000 iterations"""
    from auv_sim_amd import _lib, synth
    ctx = _lib.Context(0)
    try:
        world = synth.make_world(seed=51, n_obstacles=64)
        ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
        ctx.set_option("ROWS", 1)               # (small batches: the four-episode kernel by option; the stream choice is the host's)

        def batch(E, n_iter, seed0, **kw):
            init = np.zeros((E, 6))
            init[:, 0], init[:, 1] = world["start"]
            s = ctx.rrt_explore_batch(init, np.arange(seed0, seed0 + E, dtype=np.uint64), n_iter, **kw).copy()
            return s, ctx.last_rrt_kernel(), ctx.last_stream_len(), [ctx.tree(e, s[e]) for e in (0, E - 1)]

        s1, k1, l1, _ = batch(40, 1200, 100)
        assert k1 == "rrt_rows_kernel" and l1 == 0
        s2, k2, l2, t2 = batch(40, 1200, 500)
        busiest = int(s1["n_draw32"].max() + 1) // 2
        assert k2 == "rrt_rows_stream_kernel" and busiest < l2 <= busiest + busiest * 3 // 100 + 1024 + 63
        assert ctx.pipeline_fallbacks()[0] == 0 and (s2["status"] >= 0).all()
        ctx.set_option("ROWS_STREAM", 0)
        s2c, k2c, _, t2c = batch(40, 1200, 500)
        ctx.set_option("ROWS_STREAM", None)
        assert k2c == "rrt_rows_kernel"
        for f in s2.dtype.names:
            assert np.array_equal(s2[f], s2c[f]), f
        for a, b in zip(t2, t2c):
            assert all(np.array_equal(a[k], b[k]) for k in a)
        assert batch(40, 1200, 900)[1] == "rrt_rows_stream_kernel"
        # another world, the same parameters: the stream, sized from the batches on the first world -- and enough
        first_world = world
        world = synth.make_world(seed=8, n_obstacles=200, polygon="notch")
        ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
        s_o, k_o, _, _ = batch(40, 1200, 900)
        assert k_o == "rrt_rows_stream_kernel" and ctx.pipeline_fallbacks()[0] == 0 and (s_o["status"] >= 0).all()
        # (the pass's time is its three launches: a stream buffer that had to grow is not in it)
        parts = ctx.last_launch_parts()
        assert ctx.last_kernel_ms() <= ctx.last_stream_ms() + parts[0] + parts[1] + 0.5
        world = first_world
        ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
        assert batch(40, 1200, 900, freq=7)[1] == "rrt_rows_kernel"          # another parameter set ...
        assert batch(40, 1200, 901, freq=7)[1] == "rrt_rows_stream_kernel"   # ... seen once
        assert batch(200, 1200, 0, freq=7)[1] == "rrt_rows_kernel"           # five times the episodes of the observed batch
        assert batch(200, 1200, 7, freq=7)[1] == "rrt_rows_stream_kernel"
        assert batch(40, 900, 0)[1] == "rrt_rows_kernel" and batch(40, 900, 1)[1] == "rrt_rows_kernel"   # short budgets: never
        assert batch(40, 1200, 900)[1] == "rrt_rows_stream_kernel"            # the host remembers the last four parameter blocks ...
        for fr in (3, 4, 5, 6):
            assert batch(40, 1200, 1, freq=fr)[1] == "rrt_rows_kernel"
        assert batch(40, 1200, 900)[1] == "rrt_rows_kernel"                   # ... and the default one has gone now
        # a stream sized from a batch that drew less: the episodes that run past it are redone, and the redo reports the new figure
        s_lo, k_lo, _, _ = batch(40, 1200, 100)
        assert k_lo == "rrt_rows_stream_kernel"
        ctx.set_option("ROWS_STREAM_CAP", 20000)
        s_hi, k_hi, _, _ = batch(40, 1200, 100)
        ctx.set_option("ROWS_STREAM_CAP", None)
        assert k_hi == "rrt_rows_kernel" and ctx.pipeline_fallbacks()[0] > 0
        for f in s_lo.dtype.names:
            assert np.array_equal(s_lo[f], s_hi[f]), f
        assert batch(40, 1200, 100)[1] == "rrt_rows_stream_kernel" and ctx.pipeline_fallbacks()[0] == 0
    finally:
        ctx.close()


def test_a_caller_that_replans_on_a_changing_world_gets_the_stream_from_its_second_call_on():
    """RRT.replanning's shape (rrt_dubins.py:297-331) as a batch caller would have it: the same parameters every call, the world
    changed between calls -- habitats removed as they are visited, the shark probabilities updated, obstacles moved.  The first call
    runs rrt_rows_kernel, every later one the stream path (sized from the calls before, whatever their world), none is redone, and
    every call's results are the classic kernel's"""
    from auv_sim_amd import _lib, synth
    ctx = _lib.Context(0)
    try:
        ctx.set_option("ROWS", 1)
        rnd = np.random.default_rng(5)
        base = synth.make_world(seed=21, n_obstacles=128, n_habitats=12)
        E, n_iter = 96, 2500
        init = np.zeros((E, 6))
        init[:, 0], init[:, 1] = base["start"]
        kernels, redone = [], 0
        habitats, obstacles, prob = base["habitats"].copy(), base["obstacles"].copy(), base["prob"].copy()
        for call in range(6):
            if call:
                habitats = habitats[:-1]                                               # removeHabitat: one visited habitat less
                prob = np.roll(prob, 7 * call, axis=1) * (0.9 + 0.02 * call)           # SharkUpdate: the occupancy grid moves on
                obstacles = obstacles + np.array([0.37 * call, -0.21 * call, 0.0])     # the obstacles drift
            ctx.set_world(obstacles, habitats, base["polygon"], base["bins"], base["cells"], prob)
            init[:, 2] = rnd.uniform(-3, 3, E)
            seeds = rnd.integers(0, 1 << 40, E).astype(np.uint64)
            got = ctx.rrt_explore_batch(init, seeds, n_iter).copy()
            kernels.append(ctx.last_rrt_kernel())
            redone += ctx.pipeline_fallbacks()[0]
            trees = [ctx.tree(e, got[e]) for e in (0, E - 1)]
            ctx.set_option("ROWS_STREAM", 0)
            ref = ctx.rrt_explore_batch(init, seeds, n_iter).copy()
            assert ctx.last_rrt_kernel() == "rrt_rows_kernel"
            for f in got.dtype.names:
                assert np.array_equal(got[f], ref[f]), (call, f)
            for e, t in zip((0, E - 1), trees):
                r = ctx.tree(e, ref[e])
                assert all(np.array_equal(t[k], r[k]) for k in t), (call, e)
            ctx.set_option("ROWS_STREAM", None)
        assert kernels == ["rrt_rows_kernel"] + ["rrt_rows_stream_kernel"] * 5 and redone == 0, (kernels, redone)
    finally:
        ctx.close()


def test_the_generated_header_is_what_the_generator_writes():
    """rrt_rows_stream_kernel.h is generated from rrt_rows_kernel.h (tools/gen_rows_stream_kernel.py): a change to the classic
    kernel's body that was not carried over fails here"""
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "gen_rows_stream_kernel.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
