"""The A* drop-in modules used like their reference counterparts (same constructor / method
signatures, Motion_plan_state lists in and out), compared with the goldens."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _mps(rows, **kw):
    from auv_sim_amd.motion_plan_state import Motion_plan_state as MPS
    return [MPS(r[0], r[1], size=r[2]) if len(r) > 2 else MPS(r[0], r[1]) for r in rows]


def test_astar_config1():
    from auv_sim_amd.astar import astar
    g = np.load(os.path.join(GOLDEN, "g1_astar_cfg1.npz"))
    obs = _mps(g["obstacles"].tolist())
    box = g["box"].tolist()
    bnd = _mps([(box[0], box[1]), (box[2], box[3])])
    solver = astar((0, 0), (490, 490), obs, bnd)
    path = solver.astar(obs, (0, 0), (490, 490))
    assert [(p.x, p.y) for p in path] == [tuple(r) for r in g["path"].tolist()]
    assert isinstance(path[0].x, int)  # int lattice in, int positions out (as the reference)


def test_astar_real():
    from auv_sim_amd.astar_real import astar
    g = np.load(os.path.join(GOLDEN, "g1_real_1.npz"))
    obs, bnd = _mps(g["obstacles"].tolist()), _mps(g["polygon"].tolist())
    start, goal = tuple(g["start"].tolist()), tuple(g["goal"].tolist())
    path = astar(start, goal, obs, bnd).astar(obs, bnd)
    assert [(p.x, p.y) for p in path] == [tuple(r) for r in g["path"].tolist()]


@pytest.mark.parametrize("name", ["g6_fixlen_1", "g6_fixlen_4"])  # a pentagon; (round 6) a CONCAVE outline: the centroid fan's quirk
def test_astar_fixlen_mutates_habitat_list_like_reference(name):
    from auv_sim_amd.astar_fixLen import astar
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    obs, hab, bnd = _mps(g["obstacles"].tolist()), _mps(g["habitats"].tolist()), _mps(g["polygon"].tolist())
    start = tuple(g["start"].tolist())
    res = astar(start, obs, bnd).astar(hab, obs, bnd, start, float(g["limit"]), g["weights"].tolist())
    assert [(p.x, p.y) for p in res[0]] == [tuple(r) for r in g["path"].tolist()]
    assert res[1] == g["cost_list"].tolist()
    assert [[h.x, h.y, h.size] for h in hab] == g["habitats_left"].tolist()


@pytest.mark.parametrize("name", ["g6_sog_0", "g6_sog_1", "g6_sog_2", "g6_sog_3", "g6_sog_4", "g6_sog_5"])  # 3-5 (round 6): the Catalina outline, a concave outline
def test_astar_fixlen_sog(name):
    from auv_sim_amd.astar_fixLenSOG import astar
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    obs, hab, bnd = _mps(g["obstacles"].tolist()), _mps(g["habitats"].tolist()), _mps(g["polygon"].tolist())
    cells = [tuple(c) for c in g["cells"].tolist()]
    shark = {(int(b[0]), int(b[1])): {cells[i]: p for i, p in enumerate(g["prob"][t].tolist())}
             for t, b in enumerate(g["bins"].tolist())}
    start = tuple(g["start"].tolist())
    res = astar(start, obs, bnd, hab, shark, {}, float(g["velocity"])).astar(float(g["limit"]), g["weights"].tolist(), {})
    if not bool(g["found"]):
        assert res is None
        return
    assert set(res.keys()) == {"path length", "path", "cost", "cost list", "node"}
    assert res["path length"] == int(g["path_length"])
    assert [[p.x, p.y, p.traj_time_stamp] for p in res["path"]] == g["path"].tolist()
    assert res["cost"] == float(g["cost"]) and res["cost list"] == g["cost_list"].tolist()
    got = [[n.position[0], n.position[1], n.g, n.h, n.f, n.cost, n.pathLen, n.time_stamp] for n in res["node"]]
    assert got == g["node_path"].tolist()
    assert all(n.parent is (res["node"][i - 1] if i else None) for i, n in enumerate(res["node"]))


def test_astar_fixlen_solver_reuse_keeps_visited_bitmap():
    """the reference's self.visited_nodes persists across astar() calls on one solver object"""
    from auv_sim_amd.astar_fixLen import astar
    g = np.load(os.path.join(GOLDEN, "h6_fixlen_twice.npz"))
    obs, bnd = _mps(g["obstacles"].tolist()), _mps(g["polygon"].tolist())
    solver = astar(tuple(g["start0"].tolist()), obs, bnd)
    for k in (0, 1):
        hab = _mps(g["habitats"].tolist())
        start = tuple(g["start%d" % k].tolist())
        res = solver.astar(hab, obs, bnd, start, float(g["limit%d" % k]), g["weights"].tolist())
        assert (res is not None) == bool(g["found%d" % k])
        if res is not None:
            assert [(p.x, p.y) for p in res[0]] == [tuple(r) for r in g["path%d" % k].tolist()]
        assert int(solver.visited_nodes.sum()) == int(g["visited_count%d" % k])


@pytest.mark.parametrize("stem", ["AUVGrid_prob_500_straight", "AUVGrid_prob_500_turn"])
def test_astar_fixlen_sog_on_the_reference_shark_csv(tmp_path, stem, orc):
    """row a12 end to end: the reference's shark_data CSV -> this module's createSharkGrid (drops each row's last value,
    astar_fixLenSOG.py:46) -> astar_fixLenSOG on the GPU, against the checker run on arrays built from what the REFERENCE's
    loader returned for the same file (G15) -- a non-grid cell list (the last row is one cell short), so the search takes the
    cell sweep"""
    import gzip
    from auv_sim_amd.astar_fixLenSOG import astar, createSharkGrid
    from auv_sim_amd.motion_plan_state import Motion_plan_state as MPS
    from oracle import orc_astar as oa
    g = np.load(os.path.join(GOLDEN, "g15_shark_grid_csv.npz"))
    p = tmp_path / (stem + ".csv")
    p.write_bytes(gzip.open(os.path.join(GOLDEN, "shark_data", stem + ".csv.gz")).read())

    class Cell:
        def __init__(self, b):
            self.bounds = b
    # 987 = 47 x 21 cells of 8 m inside the window the hard-coded visited bitmap covers (x + 500, y + 200 index a 600 x 600
    # array, astar_fixLenSOG.py:655-657)
    cells = [Cell((-300.0 + 8.0 * c, -100.0 + 8.0 * r, -292.0 + 8.0 * c, -92.0 + 8.0 * r)) for r in range(21) for c in range(47)]
    shark = createSharkGrid(str(p), cells)
    pre = stem + "_sog_"
    n = int(g[pre + "lens"][0])
    assert all(len(v) == n for v in shark.values()) and n in (986, 985)
    rng = np.random.default_rng(8)
    obstacles = np.column_stack([rng.uniform(-280, 60, 24), rng.uniform(-80, 50, 24), rng.uniform(2, 6, 24)])
    habitats = np.column_stack([rng.uniform(-280, 60, 8), rng.uniform(-80, 50, 8), rng.uniform(10, 25, 8)])
    poly = [(-300.0, -100.0), (76.0, -100.0), (76.0, 68.0), (-300.0, 68.0)]
    bins = g[pre + "keys"].astype(np.float64)
    cell_arr = np.array([c.bounds for c in cells[:n]])
    prob = g[pre + "vals"].reshape(len(bins), n)
    obs, hab, bnd = _mps(obstacles.tolist()), _mps(habitats.tolist()), [MPS(x, y) for x, y in poly]
    # the reference's `sharkGrid == {}` fallback (:127-132) with the cell list and the CSV handed in
    fb = astar((-250.0, -50.0), obs, bnd, hab, {}, {}, 1.0, cell_list=cells, shark_csv=str(p))
    assert list(fb.sharkGrid.keys()) == list(shark.keys()) and fb.sharkGrid[list(shark.keys())[0]] == shark[list(shark.keys())[0]]
    with pytest.raises(ValueError):
        astar((-250.0, -50.0), obs, bnd, hab, {}, {}, 1.0)
    for start, limit in (((-250.0, -50.0), 200.0), ((20.0, 30.0), 300.0), ((-100.0, 0.0), 100.0)):
        res = astar(start, obs, bnd, hab, shark, {}, 1.0).astar(limit, [0, 10, 10, 100], {})
        o = oa.run("astar_fixLenSOG", np.array(start), obstacles=obstacles, habitats=habitats, polygon=np.array(poly), bins=bins,
                   cells=cell_arr, prob=prob, limit=limit, weights=(0, 10, 10, 100), velocity=1.0, cap_nodes=200000, kind="portable")
        assert (res is not None) == o["found"]
        if res is not None:
            assert [[q.x, q.y, q.traj_time_stamp] for q in res["path"]] == o["smooth_path"].tolist()
            assert res["cost list"] == o["cost_list"].tolist()
