"""oracle/orc_planner.c (Planner_RRT restatement) pinned against the G2 goldens captured from the
reference's gym_rrt/envs/rrt_dubins.py (tests/golden/make_golden.py g2).  No GPU needed."""
import glob
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, libm_matches_golden

G2 = sorted(glob.glob(os.path.join(GOLDEN, "g2_*.npz")))
ST = ("st_bucket", "st_picked", "st_accepted", "st_done", "st_npath", "st_arc_n", "st_arc_free")


def run_oracle(g, kind):
    from oracle import orc_planner as op
    return op.planning(g["obstacles"], g["rect"], g["start"], g["goal"], int(g["seed"]), int(g["max_step"]),
                       int(g["freq"]), int(g["cell"]), int(g["subs"]), float(g["exp_rate"]), float(g["dist_to_end"]),
                       float(g["diff_max"]), kind=kind)


@pytest.mark.parametrize("path", G2, ids=[os.path.basename(p)[:-4] for p in G2])
@pytest.mark.parametrize("kind", ["libm", "portable"])
def test_planner_rrt_matches_reference(orc, path, kind):
    g = np.load(path)
    r = run_oracle(g, kind)
    assert r["status"] == 0
    assert r["steps"] == int(g["steps"]) and r["done"] == bool(g["done"])
    assert r["n_nodes"] == len(g["nodes"])
    assert np.array_equal(r["parent"], g["parent"])
    assert np.array_equal(r["pt_cnt"][1:] + 1, g["npath"][1:])
    for k in ST:
        assert np.array_equal(r[k], g[k]), k
    assert np.array_equal(r["occupied"], g["occupied"])
    assert np.array_equal(r["bucket_counts"], g["bucket_counts"])
    assert r["grid_rows"] == int(g["grid_rows"]) and r["grid_cols"] == int(g["grid_cols"])
    assert r["rng_after"] == float(g["rng_after"])
    if kind == "libm" and libm_matches_golden():
        assert np.array_equal(r["nodes"], g["nodes"])
        assert hashlib.sha256(np.ascontiguousarray(r["points"]).tobytes()).hexdigest() == str(g["points_sha"])
        if "path" in g.files:
            assert np.array_equal(r["path"], g["path"])
    else:
        np.testing.assert_allclose(r["nodes"], g["nodes"], rtol=1e-9, atol=1e-9)
        if "points" in g.files:
            np.testing.assert_allclose(r["points"], g["points"], rtol=1e-9, atol=1e-9)
        if "path" in g.files:
            assert r["path"].shape == g["path"].shape
            np.testing.assert_allclose(r["path"], g["path"], rtol=1e-9, atol=1e-9)
