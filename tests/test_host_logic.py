"""Host-side logic that needs no GPU: shark-grid CSV loader / packer, synthetic worlds, record type."""
import os

import numpy as np
import pytest


class _Cell:
    def __init__(self, b):
        self.bounds = tuple(b)


def test_create_shark_grid_roundtrip(tmp_path):
    from auv_sim_amd.rrt_dubins import createSharkGrid, pack_shark_grid
    cells = [_Cell((10.0 * i, 0.0, 10.0 * i + 10.0, 10.0)) for i in range(5)]
    p = tmp_path / "grid.csv"
    # the reference's wire format (path_planning/shark_data/*.csv): header `time bin,grid`
    p.write_text('time bin,grid\n"(0, 50)","[0.1, 0.2, 0.0, 0.4, 0.5]"\n"(50, 100)","[0.5, 0.25, 0.125, 0.0, 1.0]"\n')
    g = createSharkGrid(str(p), cells)
    assert list(g.keys()) == [(0, 50), (50, 100)]
    assert list(g[(0, 50)].keys()) == [c.bounds for c in cells]
    assert g[(50, 100)][cells[2].bounds] == 0.125
    bins, cl, prob = pack_shark_grid(g)
    assert bins.tolist() == [[0.0, 50.0], [50.0, 100.0]]
    assert cl.shape == (5, 4) and prob.shape == (2, 5) and prob[1, 4] == 1.0


@pytest.mark.parametrize("stem", ["AUVGrid_prob_200_straight", "AUVGrid_prob_500_straight", "AUVGrid_prob_500_turn"])
@pytest.mark.parametrize("twin", ["rrt", "sog"])
def test_create_shark_grid_matches_the_reference_on_its_own_csv(tmp_path, stem, twin):
    """both createSharkGrid twins (rrt_dubins.py:612-630; astar_fixLenSOG.py:31-49 drops the last value of each row) on the
    reference's own shark_data/*.csv, against what the reference's loaders returned for them (G15)"""
    import gzip
    from conftest import GOLDEN
    from auv_sim_amd import astar_fixLenSOG, rrt_dubins
    g = np.load(os.path.join(GOLDEN, "g15_shark_grid_csv.npz"))
    p = tmp_path / (stem + ".csv")
    p.write_bytes(gzip.open(os.path.join(GOLDEN, "shark_data", stem + ".csv.gz")).read())
    cells = [_Cell((float(i), 0.5 * i, float(i) + 1.0, 0.5 * i + 1.0)) for i in range(1200)]  # the stand-ins G15 used
    fn = rrt_dubins.createSharkGrid if twin == "rrt" else astar_fixLenSOG.createSharkGrid
    got = fn(str(p), cells)
    pre = "%s_%s_" % (stem, twin)
    keys = list(got.keys())
    assert np.array_equal(np.array(keys, dtype=np.int64).reshape(-1, 2), g[pre + "keys"])
    assert [len(got[k]) for k in keys] == g[pre + "lens"].tolist()
    assert np.array_equal(np.concatenate([np.array(list(got[k].values())) for k in keys]), g[pre + "vals"])
    assert list(got[keys[0]].keys())[0] == tuple(g[pre + "first_cell"]) and list(got[keys[-1]].keys())[-1] == tuple(g[pre + "last_cell"])
    n = int(g[pre + "lens"][0])
    assert n == {"AUVGrid_prob_200_straight": 987, "AUVGrid_prob_500_straight": 987, "AUVGrid_prob_500_turn": 986}[stem] - (twin == "sog")
    # and the packed arrays the device tables are built from
    bins, cl, prob = rrt_dubins.pack_shark_grid(got)
    assert bins.shape == (len(keys), 2) and cl.shape == (n, 4) and prob.shape == (len(keys), n)
    assert np.array_equal(prob.ravel(), g[pre + "vals"])


def test_pack_shark_grid_rejects_inconsistent_cell_order():
    from auv_sim_amd.rrt_dubins import pack_shark_grid
    a, b = (0.0, 0.0, 1.0, 1.0), (1.0, 0.0, 2.0, 1.0)
    with pytest.raises(NotImplementedError):
        pack_shark_grid({(0, 1): {a: 0.1, b: 0.2}, (1, 2): {b: 0.1, a: 0.2}})


def test_synth_worlds_are_reproducible():
    from auv_sim_amd import synth
    w1, w2 = synth.make_world(seed=5, n_obstacles=64), synth.make_world(seed=5, n_obstacles=64)
    assert all(np.array_equal(w1[k], w2[k]) for k in ("obstacles", "habitats", "cells", "prob"))
    assert w1["cells"].shape == (400, 4) and w1["prob"].shape == (10, 400)
    # obstacles leave the start clear
    d = np.hypot(w1["obstacles"][:, 0] - w1["start"][0], w1["obstacles"][:, 1] - w1["start"][1])
    assert (d > w1["obstacles"][:, 2] + 5.0).all()


def test_motion_plan_state_signature():
    from auv_sim_amd.motion_plan_state import Motion_plan_state as MPS
    m = MPS(1, 2, size=3)
    assert (m.x, m.y, m.z, m.theta, m.size, m.parent, m.path, m.length) == (1, 2, 0, 0, 3, None, [], 0)
    assert repr(MPS(1, 2)) == "MPS: [x=1, y=2]" and "size=3" in repr(MPS(1, 2, z=-5, size=3))
    assert MPS(1, 2, length=4.0).length == 4.0 and MPS(0, 0, rl_state_id=7).rl_state_id == 7


def test_habitat_grid_matches_reference_layout():
    """auv_sim_amd.habitatGrid vs G11 (captured from habitatGrid.py): id layout, habitat list, lookups"""
    import contextlib
    import io
    import os
    import numpy as np
    from conftest import GOLDEN
    from auv_sim_amd.habitatGrid import HabitatGrid
    g = np.load(os.path.join(GOLDEN, "g11_habitat_grid.npz"))
    k = 0
    while f"c{k}_args" in g:
        ex, ey, sx, sy, hs, cs = g[f"c{k}_args"].tolist()
        grid = HabitatGrid(ex, ey, sx, sy, habitat_side_length=hs, cell_side_length=cs)
        assert np.array_equal(grid.habitat_id_grid, g[f"c{k}_ids"])
        xy = np.array([[[c.x, c.y] for c in row] for row in grid.habitat_cell_grid])
        assert np.array_equal(xy, g[f"c{k}_cell_xy"])
        habs = np.array([[h.x, h.y, h.id, h.side_length] for h in grid.habitat_array], dtype=np.float64)
        assert np.array_equal(habs, g[f"c{k}_habitats"])
        with contextlib.redirect_stdout(io.StringIO()):
            inside = [grid.inside_habitat(p) for p in g[f"c{k}_query"]]
        assert [(-1 if c is False else c.habitat_id) for c in inside] == g[f"c{k}_inside_id"].tolist()
        assert [int(grid.within_habitat_env(p)) for p in g[f"c{k}_query"]] == g[f"c{k}_within"].tolist()
        k += 1
    assert k == 4


def _g12_world(c):
    from auv_sim_amd import synth
    from auv_sim_amd.motion_plan_state import Motion_plan_state as MPS
    world = synth.make_world(seed=c["world_seed"], n_obstacles=3, n_habitats=c["n_habitats"], cell=c["cell"],
                             n_bins=c["n_bins"])
    cells = [tuple(r) for r in world["cells"].tolist()]
    keys = [(int(b[0]), int(b[1])) for b in world["bins"].tolist()]
    shark = {k: {cells[i]: p for i, p in enumerate(world["prob"][t].tolist())} for t, k in enumerate(keys)}
    habitats = [MPS(h[0], h[1], size=h[2]) for h in world["habitats"].tolist()]
    return shark, habitats


def test_cost_of_edge_and_point_form_match_reference():
    """Cost.cost_of_edge (cost.py:66) and habitat_shark_cost_point (path_planning/cost.py:209) vs G12"""
    import json
    import os
    import types
    from conftest import GOLDEN
    from auv_sim_amd.cost import Cost, habitat_shark_cost_point
    from auv_sim_amd.motion_plan_state import Motion_plan_state as MPS
    g = json.load(open(os.path.join(GOLDEN, "g12_cost_twins.json")))
    cal = Cost.__new__(Cost)  # host arithmetic only: no device context needed
    for c in g["edges"]:
        _, habitats = _g12_world(c)
        node = types.SimpleNamespace(position=tuple(c["pos"]))
        r = cal.cost_of_edge(node, habitats[:c["cut"]], habitats[c["cut"]:], c["weights"])
        assert [float(r[0]), r[1], r[2]] == c["out"]
    for c in g["points"]:
        shark, habitats = _g12_world(c)
        grid = shark[list(shark.keys())[c["bin"]]]
        visited = [False for _ in habitats]
        for p, want in zip(c["pts"], c["out"]):
            got, visited = habitat_shark_cost_point(MPS(p[0], p[1], traj_time_stamp=p[2]), habitats, visited, grid, c["weights"])
            assert float(got) == want
        assert not any(visited)
