"""oracle/orc_astar.c (the four runnable A* variants) pinned against the G1 / G6 goldens captured from
the reference's path_planning/astar*.py (tests/golden/make_golden.py g1 g6).  No GPU needed."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN

FILES = sorted(glob.glob(os.path.join(GOLDEN, "g1_*.npz")) + glob.glob(os.path.join(GOLDEN, "g6_*.npz")))


def astar_kwargs(g):
    v = str(g["variant"])
    kw = dict(obstacles=g["obstacles"])
    if v == "astar":
        kw.update(goal=g["goal"], box=g["box"])
    elif v == "astar_real":
        kw.update(goal=g["goal"], polygon=g["polygon"])
    elif v == "astar_fixLen":
        kw.update(polygon=g["polygon"], habitats=g["habitats"], limit=float(g["limit"]), weights=g["weights"])
    else:
        kw.update(polygon=g["polygon"], habitats=g["habitats"], bins=g["bins"], cells=g["cells"], prob=g["prob"],
                  limit=float(g["limit"]), weights=g["weights"], velocity=float(g["velocity"]))
    return v, kw


def check_result(r, g):
    """r: dict with found, expansions, path, cost_list, node_path, smooth_path, hab_left, visited_count"""
    assert r["status"] == 0
    assert r["found"] == bool(g["found"])
    assert np.array_equal(r["expansions"], g["expansions"])  # pop order and every g/h/f/cost/pathLen/time_stamp
    if bool(g["found"]):
        if "path_length" in g.files:  # fixLenSOG result dict
            assert len(r["smooth_path"]) == int(g["path_length"])
            assert np.array_equal(r["smooth_path"], g["path"])
            assert np.array_equal(r["cost_list"], g["cost_list"])
            assert r["cost_list"][0] == float(g["cost"])
            assert np.array_equal(r["node_path"], g["node_path"])
        else:
            assert np.array_equal(r["path"][:, :2], g["path"])
            if "cost_list" in g.files:
                assert np.array_equal(r["cost_list"], g["cost_list"])
    if "visited_count" in g.files:
        assert r["visited_count"] == int(g["visited_count"])
    if "habitats_left" in g.files:  # the reference pops from the caller's habitat list
        assert np.array_equal(g["habitats"][r["hab_left"]], g["habitats_left"])


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[:-4] for p in FILES])
@pytest.mark.parametrize("kind", ["libm", "portable"])
def test_astar_matches_reference(orc, path, kind):
    from oracle import orc_astar as oa
    g = np.load(path)
    v, kw = astar_kwargs(g)
    check_result(oa.run(v, g["start"], kind=kind, **kw), g)
