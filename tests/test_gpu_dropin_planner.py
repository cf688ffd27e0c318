"""Planner_RRT drop-in (auv_sim_amd.planner_rrt) used the way gym_rrt's RRTEnv / main() use it."""
import os
import random

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _mk(g, max_nodes=2100):
    from auv_sim_amd.motion_plan_state import Motion_plan_state as MPS
    from auv_sim_amd.planner_rrt import Planner_RRT
    obstacles = [MPS(o[0], o[1], size=o[2]) for o in g["obstacles"].tolist()]
    bnd = [MPS(float(g["rect"][0]), float(g["rect"][1])), MPS(float(g["rect"][2]), float(g["rect"][3]))]
    st = g["start"].tolist()
    start = MPS(st[0], st[1], z=-5.0, theta=st[2] if len(st) > 2 else 0.0)
    goal = MPS(float(g["goal"][0]), float(g["goal"][1]), z=-5.0)
    return Planner_RRT(start, goal, bnd, obstacles, [], exp_rate=float(g["exp_rate"]), freq=int(g["freq"]),
                       cell_side_length=int(g["cell"]), subsections_in_cell=int(g["subs"]), max_nodes=max_nodes), start


@pytest.mark.parametrize("name", ["g2_main_s4", "g2_o64_100m"])
def test_planning_dropin_matches_reference(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    random.seed(int(g["seed"]))
    rrt, start = _mk(g)
    path, step, secs = rrt.planning(max_step=int(g["max_step"]))
    assert step == int(g["steps"])
    assert random.random() == float(g["rng_after"])
    got = np.array([[p.x, p.y, p.theta, p.traj_time_stamp, p.length] for p in path])
    assert got.shape == g["path"].shape
    np.testing.assert_allclose(got, g["path"], rtol=1e-9, atol=1e-9)
    # attributes RRTEnv reads
    assert rrt.mps_list[0] is start and len(rrt.mps_list) == len(g["nodes"])
    counts = [len(sub.node_array) for row in rrt.env_grid for gc in row for sub in gc.subsection_cells]
    assert counts == g["bucket_counts"].tolist()
    S, ncols = int(g["subs"]), int(g["grid_cols"])
    occ = [(r * ncols + c) * S + k for (r, c, k) in rrt.occupied_grid_cells_array]
    assert occ == g["occupied"].tolist()
    idx = {id(n): i for i, n in enumerate(rrt.mps_list)}
    assert [(-1 if n.parent is None else idx[id(n.parent)]) for n in rrt.mps_list] == g["parent"].tolist()


def test_generate_one_node_contract():
    g = np.load(os.path.join(GOLDEN, "g2_main_s0.npz"))
    random.seed(3)
    rrt, start = _mk(g)
    r, c, k = rrt.occupied_grid_cells_array[0]
    cell = rrt.env_grid[r][c].subsection_cells[k]
    assert cell.node_array == [start]
    empty = rrt.env_grid[0][0].subsection_cells[0]
    assert rrt.generate_one_node(empty, step_num=0) == (False, None)
    outcomes = set()
    for i in range(60):
        done, out = rrt.generate_one_node(cell, step_num=i)
        assert done is False or isinstance(out, list)
        if out is not None and not done:
            assert out.parent is start or out.parent in rrt.mps_list
            assert out.path[0] is out.parent and out.rl_state_id == i
        outcomes.add((done, out is None))
    assert (False, False) in outcomes  # at least one accepted node


def test_plan_batch_seeded_matches_golden():
    from auv_sim_amd.planner_rrt import plan_batch
    g = np.load(os.path.join(GOLDEN, "g2_main_s4.npz"))
    res, pb = plan_batch([g["start"].tolist()] * 3, [g["goal"].tolist()] * 3, g["rect"].tolist(),
                         [tuple(o) for o in g["obstacles"].tolist()], seeds=[4, 0, 4], max_step=int(g["max_step"]),
                         freq=int(g["freq"]), cell_side_length=int(g["cell"]), subsections_in_cell=int(g["subs"]))
    assert res[0]["done"] and res[0]["steps"] == int(g["steps"])
    np.testing.assert_allclose(res[0]["path"], g["path"], rtol=1e-9, atol=1e-9)
    assert np.array_equal(res[0]["path"], res[2]["path"])
    g0 = np.load(os.path.join(GOLDEN, "g2_main_s0.npz"))
    assert res[1]["done"] == bool(g0["done"]) and res[1]["steps"] == int(g0["steps"]) and res[1]["n_nodes"] == len(g0["nodes"])
