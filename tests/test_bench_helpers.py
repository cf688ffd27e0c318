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


def test_per_kernel_bytes_and_the_binding_roof(tmp_path, monkeypatch):
    """the expansion kernel is billed its own share of B_exp, the leaf pass its COMPULSORY bytes (every tree element once),
    and `bound` names the roof that binds: vector issue when its fraction exceeds the HBM fraction"""
    b = _bench()
    from auv_sim_amd import _lib
    s = np.zeros(2, dtype=_lib.SUMMARY_DTYPE)
    s["iters_run"] = [100, 50]
    s["n_nodes"] = [11, 6]
    s["n_points"] = [70, 30]
    s["leaf_elems"] = [200, 0]
    s["nn_scanned"] = [1000, 0]
    assert b.rrt_expand_bytes(s) == 150 * 52 + 15 * 60 + 100 * 56 + 1000 * 16
    st = {"nodes_visited": 9, "points_visited": 40, "elements_resummed": 12, "leaves_resummed": 1}
    assert b.rrt_leaf_bytes(s, st) == 17 * 17 + 9 * 128 + 40 * 24 + 12 * 24
    # the share billed to the leaf pass never exceeds what the 8(d) walk formula bills when leaves share ancestors
    assert b.rrt_expand_bytes(s) + 200 * 32 == b.rrt_bytes(s)
    r = b.roofline(8e8, 1.0, "k", valu_issue_frac=0.74)   # 0.1 of the HBM peak, 0.74 of the issue slots
    assert r["bound"] == "valu_issue" and abs(r["frac"] - 0.1) < 1e-12 and r["valu_issue_frac"] == 0.74
    r = b.roofline(7e9, 1.0, "k", valu_issue_frac=0.2)
    assert r["bound"] == "hbm"
    monkeypatch.setitem(b.HBM_MEASURED, "read_GBps", 5000.0)
    r = b.roofline(4e9, 1.0, "k")
    assert abs(r["frac_of_measured"] - 0.8) < 1e-12 and abs(r["frac"] - 0.5) < 1e-12 and r["hbm_measured_GBps"] == 5000.0
    f = tmp_path / "pmc.json"
    f.write_text(json.dumps({"tag": "t", "measurements": {"headline": {
        "kernels": {"rrt_rows_kernel": {"per_launch": {"SQ_INSTS_VALU": 1024.0 * 100, "GRBM_GUI_ACTIVE": 8.0 * 1000, "FETCH_SIZE": 1.0, "WRITE_SIZE": 2.0}}},
        "per_launch": {"SQ_INSTS_VALU": 1024.0 * 150, "GRBM_GUI_ACTIVE": 8.0 * 2000}, "units": 10.0}}}))
    monkeypatch.setattr(b, "PMC_FILE", str(f))
    assert abs(b.pmc_valu_issue("headline", "rrt_rows_kernel") - 0.4) < 1e-12
    assert abs(b.pmc_valu_issue("headline") - 0.3) < 1e-12
    assert b.pmc_valu_issue("nothing") is None
    assert b.pmc_kernel_traffic("headline", "rrt_rows_kernel", 20.0) == (2 * (2 * 1024.0 + 2048.0), 2 * (1024.0 + 2048.0))


def test_workload_string_fits_the_drivers_record():
    """the driver's record cuts config.workload at 120 characters: the parent-sampling mode must survive"""
    src = open(os.path.join(REPO, "bench.py")).read()
    assert '"workload": wl[:120]' in src
    wl = "RRT.exploring %d obst %dx%d cells %d iters x %d episodes/GPU, %s parent sampling" % (256, 200, 200, 10000, 12288, "plantime")
    assert len(wl) <= 120
