"""BASELINE.json configs[4] ("particleFilter.py 100k particles x rrt_dubins replan per step over real
sharkTrackingData.csv"): the composed path particle filter -> per-particle Planner_RRT replan, device resident
(auv_sim_amd.tracking.ParticleReplanner: auvp_pf_run -> auvp_prrt_replan_particles -> auvp_prrt_plan).

Parity: every stage against the CPU checker (portable math) on the same inputs -- the filter's particles after each
tracking step (orc_pf), the goals derived from them, and the planner episodes (orc_planner with the same seed) -- on a
small batch for every episode and on the full per-GPU size (25 filters x 500 particles = 12 500 episodes x 200 steps)
for a sample plus size-independent properties.  Inputs: the reference's recorded shark tracks
(tests/golden/shark_tracking_xy.npz, made from data/sharkTrackingData.csv by make_shark_track_fixture.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
RECT = (0.0, 0.0, 200.0, 200.0)
START = (20.0, 20.0)


@pytest.fixture(scope="module")
def tracks():
    return np.load(os.path.join(GOLDEN, "shark_tracking_xy.npz"))["xy"]


def _world():
    from auv_sim_amd import synth
    return synth.make_rect_world(seed=3, n_obstacles=256)


def _check_filter_step(orc_pf, rp, s, states, kind="portable"):
    """advance the checker's filters by tracking step s and compare with the device particles"""
    dev, obj = rp.filters.particles()
    for f in range(rp.F):
        st = states[f]
        r = orc_pf.run(rp.N, rp.meas[s:s + 1, f], rp.shark[s:s + 1, f], rp.tracks[f, 0], st["mt"], st["pos"],
                       init=st["particles"], init_obj=st["obj"], kind=kind)
        st["particles"], st["obj"], st["mt"], st["pos"] = r["resampled"][-1], r["choice"][-1], r["mt"], r["mt_pos"]
        assert np.array_equal(dev[f], st["particles"]), (s, f)
        assert np.array_equal(obj[f], st["obj"]), (s, f)


def _checker_filters(orc_pf, rp, filter_seeds):
    from auv_sim_amd import _pf_lib
    states = []
    for f in range(rp.F):
        mt, pos = _pf_lib.np_seed_state(int(filter_seeds[f]))
        r = orc_pf.run(rp.N, np.zeros((0, 1, 5)), np.zeros((0, 2)), rp.tracks[f, 0], mt, pos, kind="portable")
        states.append({"particles": r["created"], "obj": np.arange(rp.N, dtype=np.int32), "mt": r["mt"], "pos": r["mt_pos"]})
    return states


def _expected_goals(rp, particles):
    """the same two operations per axis as the device (mul, add; contraction off on both sides), then the clamp"""
    g = np.zeros((rp.F, rp.N, 2))
    for f in range(rp.F):
        g[f, :, 0] = particles[f][:, 0] * rp.xform[f, 0] + rp.xform[f, 1]
        g[f, :, 1] = particles[f][:, 1] * rp.xform[f, 2] + rp.xform[f, 3]
    g[..., 0] = np.clip(g[..., 0], rp.clamp[0], rp.clamp[2])
    g[..., 1] = np.clip(g[..., 1], rp.clamp[1], rp.clamp[3])
    return g.reshape(-1, 2)


def _check_episode(op, w, rp, summ, e, goal, seed):
    r = op.planning(w["obstacles"], RECT, [START[0], START[1], 0.0, 0.0], goal, seed, rp.kw["max_step"], rp.kw["freq"],
                    rp.kw["cell"], rp.kw["subs"], kind="portable")
    s = summ[e]
    assert s["status"] == r["status"] and s["steps"] == r["steps"] and bool(s["done"]) == r["done"], (e, s, r["steps"])
    assert s["n_nodes"] == r["n_nodes"] and s["n_points"] == r["n_points"] and s["rng_after"] == r["rng_after"]
    t = rp.planner.tree(e, s)
    assert np.array_equal(t["parent"], r["parent"]) and np.array_equal(t["nodes"], r["nodes"][:, :4])
    return r


def test_composed_path_small_every_episode(orc, tracks):
    from auv_sim_amd import _lib, tracking
    from oracle import orc_pf, orc_planner as op
    w = _world()
    ctx = _lib.Context(0)
    ctx.set_world(obstacles=w["obstacles"])
    F, N = 3, 40
    seeds = [11, 12, 13]
    rp = tracking.ParticleReplanner(ctx, tracks[[0, 5, 9], :6], N, RECT, START, seeds, max_step=200, episode_seed_base=1000)
    states = _checker_filters(orc_pf, rp, seeds)
    for s in range(3):
        summ = rp.step(s)
        _check_filter_step(orc_pf, rp, s, states)
        goals = rp.planner.goals()
        assert np.array_equal(goals, _expected_goals(rp, [st["particles"] for st in states]))
        paths = rp.planner.paths(summ)
        for e in range(F * N):
            r = _check_episode(op, w, rp, summ, e, goals[e], rp.episode_seed(s, e))
            if r["done"]:
                assert np.array_equal(paths[e], r["path"])


def test_composed_path_full_size(orc, tracks):
    """12 500 episodes (25 filters x 500 particles) x 200 steps per tracking step, two tracking steps"""
    from auv_sim_amd import _lib, tracking
    from oracle import orc_pf, orc_planner as op
    w = _world()
    ctx = _lib.Context(0)
    ctx.set_world(obstacles=w["obstacles"])
    F, N = 25, 500
    seeds = list(range(100, 100 + F))
    rp = tracking.ParticleReplanner(ctx, tracks[np.arange(F) % 32, :4], N, RECT, START, seeds, max_step=200)
    states = _checker_filters(orc_pf, rp, seeds)
    rng = np.random.default_rng(0)
    for s in range(2):
        summ = rp.step(s)
        assert len(summ) == 12500
        _check_filter_step(orc_pf, rp, s, states)
        goals = rp.planner.goals()
        assert np.array_equal(goals, _expected_goals(rp, [st["particles"] for st in states]))
        # properties that do not depend on the size
        assert (summ["status"] >= 0).all()
        assert ((summ["steps"] >= 1) & (summ["steps"] <= 200)).all()
        assert ((summ["done"] != 0) | (summ["steps"] == 200)).all()   # an episode stops early only when it is done
        assert (summ["n_nodes"] <= summ["steps"] + 1).all()
        assert ((goals[:, 0] >= rp.clamp[0]) & (goals[:, 0] <= rp.clamp[2]) & (goals[:, 1] >= rp.clamp[1]) &
                (goals[:, 1] <= rp.clamp[3])).all()
        paths = rp.planner.paths(summ)
        done = np.nonzero(summ["done"])[0]
        for e in done[:50]:
            assert len(paths[e]) == summ[e]["path_len"] > 0
            assert np.array_equal(paths[e][-1, :2], np.array(START)) or np.array_equal(paths[e][0, :2], np.array(START))
        for e in rng.choice(12500, size=10, replace=False):
            _check_episode(op, w, rp, summ, int(e), goals[e], rp.episode_seed(s, int(e)))


def test_results_do_not_depend_on_the_sharding(tracks):
    """episodes are seeded by their global id: a rank that holds filters 2..3 of 4 gets what the single-GPU run got"""
    from auv_sim_amd import _lib, tracking
    w = _world()
    ctx = _lib.Context(0)
    ctx.set_world(obstacles=w["obstacles"])
    N, seeds = 64, [5, 6, 7, 8]
    whole = tracking.ParticleReplanner(ctx, tracks[:4, :3], N, RECT, START, seeds, max_step=120)
    a = whole.step(0).copy()
    part = tracking.ParticleReplanner(ctx, tracks[2:4, :3], N, RECT, START, seeds[2:], max_step=120, episode_offset=2 * N,
                                      episodes_total=4 * N)
    b = part.step(0)
    for k in ("status", "steps", "done", "n_nodes", "n_points", "rng_after", "path_len"):
        assert np.array_equal(a[k][2 * N:], b[k]), k
