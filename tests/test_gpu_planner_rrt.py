"""GPU parity tests for Planner_RRT (gym_rrt/envs/rrt_dubins.py) through the C-ABI.

  HIP planning()  ==  oracle(portable math)   bit-for-bit
  HIP planning()  ~=  golden (reference)      decisions exact, floats <= 1e-9
  HIP generate_one_node stepping (bucket chosen on the host from Python's global random stream,
      as RRTEnv / planning() do)  ==  golden
"""
import glob
import os
import random

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

G2 = sorted(glob.glob(os.path.join(GOLDEN, "g2_*.npz")))
ST = ("st_bucket", "st_picked", "st_accepted", "st_done", "st_npath", "st_arc_n", "st_arc_free")


@pytest.fixture(scope="module")
def ctx():
    from auv_sim_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def _batch(ctx, g, seeds=None, mt_states=None, E=1, step_log=True):
    from auv_sim_amd._prrt_lib import PlannerBatch
    ctx.set_world(obstacles=g["obstacles"])
    st = list(g["start"]) + [0.0] * (4 - len(g["start"]))
    starts = np.tile(np.array(st, dtype=np.float64), (E, 1))
    goals = np.tile(np.asarray(g["goal"], dtype=np.float64), (E, 1))
    return PlannerBatch(ctx, starts, goals, g["rect"], int(g["max_step"]), seeds=seeds, mt_states=mt_states,
                        freq=int(g["freq"]), cell=int(g["cell"]), subs=int(g["subs"]), exp_rate=float(g["exp_rate"]),
                        dist_to_end=float(g["dist_to_end"]), diff_max=float(g["diff_max"]), step_log=step_log)


def _check_against_golden(pb, s, g, e=0):
    assert s["status"] == 0
    assert s["steps"] == int(g["steps"]) and bool(s["done"]) == bool(g["done"])
    assert s["n_nodes"] == len(g["nodes"])
    t = pb.tree(e, s)
    assert np.array_equal(t["parent"], g["parent"])
    assert np.array_equal(t["pt_cnt"][1:] + 1, g["npath"][1:])
    np.testing.assert_allclose(t["nodes"], g["nodes"][:, :4], rtol=1e-9, atol=1e-9)
    if "points" in g.files:
        np.testing.assert_allclose(t["points"], g["points"], rtol=1e-9, atol=1e-9)
    occ, cnt = pb.grid(e)
    assert np.array_equal(occ, g["occupied"]) and np.array_equal(cnt, g["bucket_counts"])
    return t


@pytest.mark.parametrize("path", G2, ids=[os.path.basename(p)[:-4] for p in G2])
def test_planning_vs_golden_and_oracle(ctx, orc, path):
    from test_oracle_planner_golden import run_oracle
    g = np.load(path)
    pb = _batch(ctx, g, seeds=[int(g["seed"])])
    summ = pb.plan()
    s = summ[0]
    t = _check_against_golden(pb, s, g)
    log = pb.step_log(0, int(s["steps"]))
    for i, k in enumerate(ST):
        assert np.array_equal(log[:, i], g[k].astype(np.int32)), k
    assert s["rng_after"] == float(g["rng_after"])
    p = pb.paths(summ)[0]
    if "path" in g.files:
        assert p.shape == g["path"].shape
        np.testing.assert_allclose(p, g["path"], rtol=1e-9, atol=1e-9)
    # bit-for-bit against the checker built on the same portable math
    r = run_oracle(g, "portable")
    assert np.array_equal(t["nodes"], r["nodes"][:, :4])
    assert np.array_equal(t["points"], r["points"])
    assert np.array_equal(t["node_bucket"], r["node_bucket"])
    assert np.array_equal(t["pt_off"], r["pt_off"])
    if r["done"]:
        assert np.array_equal(p, r["path"])
    assert int(s["n_draw32"]) == int(r["n_draw32"])


def test_planning_batch_vs_oracle(ctx, orc):
    from auv_sim_amd import synth
    from auv_sim_amd._prrt_lib import PlannerBatch
    from oracle import orc_planner as op
    w = synth.make_rect_world(seed=9, n_obstacles=256)
    ctx.set_world(obstacles=w["obstacles"])
    E = 19
    starts = np.tile(np.array([w["start"][0], w["start"][1], 0.3, 0.0]), (E, 1))
    starts[:, 2] = np.linspace(-3.0, 3.0, E)
    goals = np.tile(w["goal"], (E, 1))
    seeds = np.arange(500, 500 + E, dtype=np.uint64)
    pb = PlannerBatch(ctx, starts, goals, w["rect"], 600, seeds=seeds, freq=10, cell=5, subs=4)
    summ = pb.plan()
    paths = pb.paths(summ)
    for e in range(E):
        r = op.planning(w["obstacles"], w["rect"], starts[e], goals[e], int(seeds[e]), 600, 10, 5, 4, kind="portable")
        s = summ[e]
        assert s["status"] == r["status"] == 0
        assert s["steps"] == r["steps"] and bool(s["done"]) == r["done"] and s["n_nodes"] == r["n_nodes"]
        t = pb.tree(e, s)
        assert np.array_equal(t["parent"], r["parent"]) and np.array_equal(t["nodes"], r["nodes"][:, :4])
        assert np.array_equal(t["points"], r["points"])
        assert s["rng_after"] == r["rng_after"]
        if r["done"]:
            assert np.array_equal(paths[e], r["path"])


@pytest.mark.parametrize("name", ["g2_main_s4", "g2_o64_100m", "g2_main_subs8"])
def test_generate_one_node_stepping_reproduces_planning(ctx, name):
    """planning() = loop of random.choice(occupied) + generate_one_node: drive the device one step
    at a time, choosing the bucket on the host from Python's global stream and handing the stream to
    the device for the step (what the drop-in does for RRTEnv.step)."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    random.seed(int(g["seed"]))

    def state():
        ver, internal, _ = random.getstate()
        return np.array(internal[:624], dtype=np.uint32).reshape(1, 624), np.array([internal[624]], dtype=np.int32)

    pb = _batch(ctx, g, mt_states=state())
    occupied = [int(pb.grid(0)[0][0])]
    n_steps = min(int(g["steps"]), 400)
    drawn_before = 0
    for i in range(n_steps):
        b = random.choice(occupied)
        assert b == int(g["st_bucket"][i])
        summ = pb.step([b], mt_states=state())
        s = summ[0]
        n = int(s["n_draw32"])  # outputs consumed since the generator state was handed over
        if n:
            random.getrandbits(32 * n)
        assert bool(s["last_accepted"]) == bool(g["st_accepted"][i])
        assert bool(s["done"]) == bool(g["st_done"][i])
        if s["last_accepted"]:
            occ, _ = pb.grid(0)
            occupied = [int(x) for x in occ]
    s = pb.summaries()[0]
    t = pb.tree(0, s)
    n = int(s["n_nodes"])
    assert np.array_equal(t["parent"], g["parent"][:n])
    np.testing.assert_allclose(t["nodes"], g["nodes"][:n, :4], rtol=1e-9, atol=1e-9)
    if n_steps == int(g["steps"]):
        assert random.random() == float(g["rng_after"])


def test_empty_bucket_and_errors(ctx):
    g = np.load(os.path.join(GOLDEN, "g2_main_s0.npz"))
    pb = _batch(ctx, g, seeds=[1])
    occ, cnt = pb.grid(0)
    empty = int(np.argmin(cnt))
    assert cnt[empty] == 0
    s = pb.step([empty])[0]      # generate_one_node on an empty cell: (False, None), nothing drawn
    assert s["status"] == 0 and s["steps"] == 1 and s["last_accepted"] == 0 and s["n_nodes"] == 1 and s["n_draw32"] == 0
