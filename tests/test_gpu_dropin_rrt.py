"""The Python drop-in boundary (auv_sim_amd.rrt_dubins.RRT / auv_sim_amd.cost) exercised the way a
reference caller would: Motion_plan_state lists in, result dict / linked Motion_plan_state objects
out, global `random` stream in and out.  Compared with the golden vectors captured from the
reference (tests/golden/make_golden.py)."""
import json
import os
import random

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


class _Cell:
    def __init__(self, b):
        self.bounds = tuple(float(v) for v in b)


class _Poly:
    """shapely.geometry.Polygon look-alike: .exterior.coords and .bounds"""

    class _Ext:
        def __init__(self, pts):
            self.coords = list(pts) + [pts[0]]

    def __init__(self, pts):
        self.exterior = _Poly._Ext([tuple(p) for p in pts])
        xs, ys = [p[0] for p in pts], [p[1] for p in pts]
        self.bounds = (min(xs), min(ys), max(xs), max(ys))


def _reference_style_inputs(g):
    from auv_sim_amd.motion_plan_state import Motion_plan_state as MPS
    obstacles = [MPS(o[0], o[1], size=o[2]) for o in g["obstacles"].tolist()]
    habitats = [MPS(h[0], h[1], size=h[2]) for h in g["habitats"].tolist()]
    cell_list = [_Cell(c) for c in g["cells"].tolist()]
    shark = {}
    for t, b in enumerate(g["bins"].tolist()):
        shark[(int(b[0]), int(b[1]))] = {cell_list[i].bounds: p for i, p in enumerate(g["prob"][t].tolist())}
    poly = _Poly(g["polygon"].tolist())
    start = MPS(float(g["start"][0]), float(g["start"][1]))
    return obstacles, habitats, cell_list, shark, poly, start


@pytest.mark.parametrize("name", ["g3_tb_o64_i500", "g3_nn_o64_i500", "g3_pt_o64_i500", "g3_tb_short_traj"])
def test_exploring_dropin_matches_reference(name):
    from auv_sim_amd.rrt_dubins import RRT
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    obstacles, habitats, cell_list, shark, poly, start = _reference_style_inputs(g)
    rrt = RRT(poly, obstacles, shark, cell_list, dist_to_end=float(g["dist_to_end"]), diff_max=float(g["diff_max"]),
              freq=int(g["freq"]))
    mode = str(g["mode"])
    n_iter = int(g["n_iter"])
    random.seed(int(g["seed"]))  # exactly what a reference user would do
    res = rrt.exploring(start, habitats, float(n_iter), int(g["bin_interval"]), int(g["v"]), int(g["shark_interval"]),
                        traj_time_stamp=(mode == "timebin"), max_plan_time=float(n_iter),
                        max_traj_time=float(g["max_traj_time"]), plan_time=(mode != "nn"),
                        weights=[int(w) for w in g["weights"]], max_iter=n_iter)
    # the global stream has advanced exactly as the reference's did
    assert random.random() == float(g["rng_after"])
    assert set(res.keys()) == {"path length", "path", "cost"}
    assert abs(res["path length"] - float(g["res_path_length"])) < 1e-9
    tot, parts = res["cost"]
    np.testing.assert_allclose([tot] + parts, g["res_cost"], rtol=0, atol=1e-6)
    course, split = res["path"]
    assert course[0] is start
    got = np.array([[p.x, p.y, p.theta, p.v, p.traj_time_stamp, p.plan_time_stamp, p.length] for p in course])
    assert got.shape == g["res_path"].shape
    np.testing.assert_allclose(got, g["res_path"], rtol=1e-9, atol=1e-9)
    assert [list(k) for k in split.keys()] == g["res_split_keys"].tolist()
    assert [len(v) for v in split.values()] == g["res_split_counts"].tolist()
    # tree as linked objects
    nodes = rrt.mps_list
    assert len(nodes) == len(g["nodes"]) and nodes[0] is start
    idx = {id(n): i for i, n in enumerate(nodes)}
    assert [(-1 if n.parent is None else idx[id(n.parent)]) for n in nodes] == g["parent"].tolist()
    assert [len(n.path) for n in nodes[1:]] == g["npath"][1:].tolist()
    assert all(n.path[0] is n.parent for n in nodes[1:])


def test_exploring_raises_like_reference_without_leaf():
    from auv_sim_amd.rrt_dubins import RRT
    g = np.load(os.path.join(GOLDEN, "g3_tb_o64_i500.npz"))
    obstacles, habitats, cell_list, shark, poly, start = _reference_style_inputs(g)
    rrt = RRT(poly, obstacles, shark, cell_list)
    with pytest.raises(TypeError):
        rrt.exploring(start, habitats, 5.0, 5, 2, 50, traj_time_stamp=True, max_plan_time=5.0, max_traj_time=500.0,
                      plan_time=True, weights=[-3, -3, -4], max_iter=5, seed=1)


def test_exploring_batch_seeded_equals_single_calls():
    from auv_sim_amd.rrt_dubins import RRT
    g = np.load(os.path.join(GOLDEN, "g3_tb_o64_i500.npz"))
    obstacles, habitats, cell_list, shark, poly, start = _reference_style_inputs(g)
    rrt = RRT(poly, obstacles, shark, cell_list)
    kw = dict(traj_time_stamp=True, max_plan_time=5.0, max_traj_time=500.0, plan_time=True, weights=[-3, -3, -4],
              max_iter=800)
    batch = rrt.exploring_batch([start] * 5, habitats, 5.0, 5, 2, 50, seeds=[3, 4, 5, 6, 7], **kw)
    for s, b in zip([3, 4, 5, 6, 7], batch):
        one = rrt.exploring(start, habitats, 5.0, 5, 2, 50, seed=s, **kw)
        assert one["cost"] == b["cost"] and one["path length"] == b["path length"]
        assert [(p.x, p.y) for p in one["path"][0]] == [(p.x, p.y) for p in b["path"][0]]


def test_cost_function_dropin():
    from auv_sim_amd import synth
    from auv_sim_amd.cost import habitat_shark_cost_func
    from auv_sim_amd.motion_plan_state import Motion_plan_state as MPS
    g = json.load(open(os.path.join(GOLDEN, "g4_cost.json")))
    for c in g["cases"][:12]:
        world = synth.make_world(seed=c["world_seed"], n_obstacles=4, n_habitats=c["n_habitats"], cell=c["cell"],
                                 n_bins=c["n_bins"])
        cells = [tuple(r) for r in world["cells"].tolist()]
        keys = [(int(b[0]), int(b[1])) for b in world["bins"].tolist()]
        shark = {k: {cells[i]: p for i, p in enumerate(world["prob"][t].tolist())} for t, k in enumerate(keys)}
        sub = {k: shark[k] for k in keys[c["bin_lo"]:c["bin_hi"]]}
        habitats = [MPS(h[0], h[1], size=h[2]) for h in world["habitats"].tolist()]
        path = [MPS(p[0], p[1], traj_time_stamp=p[2]) for p in c["pts"]]
        if not sub:
            continue  # an empty dict has no cell order to pack; covered through the C-ABI test
        res = habitat_shark_cost_func(path, c["total"], habitats, sub, c["weights"])
        np.testing.assert_allclose([res[0]] + res[1], c["out"], rtol=1e-12, atol=1e-6)


@pytest.mark.parametrize("name", ["g13_replan_a", "g13_replan_b"])
def test_replanning_matches_reference(name):
    """RRT.replanning (rrt_dubins.py:51-90) vs the reference's own multi-round run (G13): rounds start at
    non-zero traj_time_stamps, reuse one RRT object, remove habitats between rounds and continue one
    global `random` stream.  max_iter = the iterations the reference's loop ran per round."""
    from auv_sim_amd.rrt_dubins import RRT
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    obstacles, habitats, cell_list, shark, poly, start = _reference_style_inputs(g)
    rrt = RRT(poly, obstacles, shark, cell_list)
    iters = g["iters_per_round"].tolist()
    assert len(set(iters)) == 1
    random.seed(int(g["seed"]))
    traj, time_dict, cost = rrt.replanning(start, habitats, float(g["plan_time_budget"]), float(g["traj_time_length"]),
                                           float(g["replan_time_interval"]), [-3, -3, -4], max_iter=iters[0])
    assert random.random() == float(g["rng_after"])
    got = np.array([[p.x, p.y, p.theta, p.v, p.traj_time_stamp, p.length] for p in traj])
    assert got.shape == g["traj"].shape
    np.testing.assert_allclose(got, g["traj"], rtol=1e-9, atol=1e-9)
    assert list(time_dict.keys()) == list(range(1, len(iters) + 1))
    assert [len(time_dict[k][0]) for k in time_dict] == g["round_len"].tolist()
    assert [len(time_dict[k][1]) for k in time_dict] == g["round_habitats"].tolist()
    # the caller's habitat list was mutated like the reference's (removeHabitat :604-610)
    assert [[h.x, h.y, h.size] for h in habitats] == g["habitats_left"].tolist()
    np.testing.assert_allclose([cost[0]] + list(cost[1]), g["cost"], rtol=0, atol=1e-6)


def test_world_survives_replanning(orc):
    """One RRT object: replanning(), then exploring() and check_collision() again.  The object's obstacles and
    boundary must still be on the device (round-1 advisor finding: the final cost call of replanning used to
    replace the world of the planner's own context): the second exploring equals the CPU checker run on the full
    world, and a path through an obstacle is still reported as colliding."""
    from auv_sim_amd.rrt_dubins import RRT
    from auv_sim_amd.motion_plan_state import Motion_plan_state as MPS
    g = np.load(os.path.join(GOLDEN, "g13_replan_a.npz"))
    obstacles, habitats, cell_list, shark, poly, start = _reference_style_inputs(g)
    rrt = RRT(poly, obstacles, shark, cell_list)
    random.seed(int(g["seed"]))
    rrt.replanning(start, list(habitats), float(g["plan_time_budget"]), float(g["traj_time_length"]),
                   float(g["replan_time_interval"]), [-3, -3, -4], max_iter=int(g["iters_per_round"][0]))
    assert rrt._ctx.world_sizes["O"] == len(obstacles) and rrt._ctx.world_sizes["V"] == len(g["polygon"])
    n_iter = 700
    res = rrt.exploring(start, habitats, float(n_iter), 5, 2, 50, traj_time_stamp=True, max_plan_time=float(n_iter),
                        max_traj_time=200.0, plan_time=True, weights=[-3, -3, -4], max_iter=n_iter, seed=11)
    w = orc.WorldArrays(g["obstacles"], g["habitats"], g["polygon"], g["bins"], g["cells"], g["prob"])
    r = orc.rrt_explore(w, 11, n_iter, init=[start.x, start.y, 0, 0, 0, 0], max_traj_time=200.0, kind="portable")
    assert r["status"] == 0
    assert [res["cost"][0]] + res["cost"][1] == r["best_cost"].tolist()
    assert len(rrt.mps_list) == r["n_nodes"]
    o = obstacles[0]
    through = MPS(o.x, o.y)
    through.path = [MPS(o.x - 1.0, o.y), MPS(o.x, o.y)]
    assert rrt.check_collision(through) is False
