"""bench.py's bookkeeping that needs no GPU: the algorithmic-byte formulas and the rule that counter
traffic from the committed PMC passes is only attached to a launch that ran the same kernels."""
import json
import os

import numpy as np

from conftest import REPO


def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_rrt_bytes_formula():
    b = _bench()
    from auv_sim_amd import _lib
    s = np.zeros(2, dtype=_lib.SUMMARY_DTYPE)
    s["iters_run"] = [100, 50]
    s["n_nodes"] = [11, 6]          # 10 + 5 accepted
    s["n_points"] = [70, 30]
    s["leaf_elems"] = [200, 0]
    s["nn_scanned"] = [1000, 0]
    want = 150 * 52 + 15 * 60 + 100 * 56 + 200 * 32 + 1000 * 16
    assert b.rrt_bytes(s) == want
    r = b.roofline(8e9, 1.0, "k")   # 8 GB in 1 ms = 8 TB/s = the peak
    assert abs(r["frac"] - 1.0) < 1e-12 and r["traffic"] is None and r["bound"] == "hbm"


def test_pmc_traffic_only_for_the_same_kernels(tmp_path, monkeypatch):
    b = _bench()
    f = tmp_path / "pmc.json"
    f.write_text(json.dumps({"tag": "t", "measurements": {"headline": {
        "kernels": {"rrt_rows_kernel": {}, "rrt_leaf_kernel": {}}, "hbm_bytes_raw": 100.0, "hbm_bytes_fetch_x2": 150.0, "units": 10.0}}}))
    monkeypatch.setattr(b, "PMC_FILE", str(f))
    t = b.pmc_traffic("headline", ["rrt_rows_kernel", "rrt_leaf_kernel"], 10.0)
    assert t["traffic"] == 150.0 and t["traffic_raw"] == 100.0
    t = b.pmc_traffic("headline", ["rrt_rows_kernel", "rrt_leaf_kernel"], 20.0)   # other batch: scaled by the work units
    assert t["traffic"] == 300.0 and "scaled" in t["traffic_source"]
    t = b.pmc_traffic("headline", ["rrt_explore_kernel", "rrt_leaf_kernel"], 10.0)  # another kernel ran: no number
    assert t["traffic"] is None and t["traffic_raw"] is None and "not comparable" in t["traffic_source"]
    assert b.pmc_traffic("nothing", ["x"], 1.0)["traffic"] is None


def test_committed_pmc_file_has_the_measurements_bench_asks_for():
    j = json.load(open(os.path.join(REPO, "profiles", "pmc_latest.json")))
    for k in ("headline", "astar", "planner_rrt", "rrt_nn", "rrt_nn_long_horizon", "config5"):
        m = j["measurements"][k]
        assert m["hbm_bytes_fetch_x2"] >= m["hbm_bytes_raw"] > 0 and m["kernels"]
    assert set(j["measurements"]["headline"]["kernels"]) == {"rrt_rows_kernel", "rrt_leaf_kernel"}
    assert set(j["measurements"]["config5"]["kernels"]) == {"prrt_rows_kernel"}
