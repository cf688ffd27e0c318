"""bench.py's bookkeeping that needs no GPU: the algorithmic-byte formulas and the rule that counter
traffic from the committed PMC passes is only attached to a launch that ran the same kernels."""
import json
import os
import sys

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
    monkeypatch.setattr(sys.modules["bench_sides.common"], "PMC_FILE", str(f))  # (the counter passes are read by bench_sides/common.py)
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
    # (the headline's pass is three launches from the second batch with a parameter block on: the generator ahead, the expansion, the leaves)
    assert set(j["measurements"]["headline"]["kernels"]) == {"rrt_stream_kernel", "rrt_rows_stream_kernel", "rrt_leaf_kernel"}
    assert set(j["measurements"]["config5"]["kernels"]) == {"prrt_rows_kernel"}
    # ONE profile run, taken on ONE state of the kernel sources (tools/digest_profile.py drops what was taken on other sources):
    # round 5's file mixed the tags r5e and r5d
    tags = {m.get("from_tag") for m in j["measurements"].values()}
    shas = {m.get("csrc_sha") for m in j["measurements"].values()}
    assert len(tags) == 1 and tags == {j["tag"]} and len(shas) == 1 and None not in shas, (tags, shas)
    # ... and that state is the tree's: the counters bench.py attaches to its rooflines were taken on THESE kernel sources
    # (tools/digest_profile.py csrc_sha: names and bytes of auv_sim_amd/csrc/*.h, *.hip); after a change there, re-run
    # tools/profile_bench.sh + tools/digest_profile.py
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(REPO, "auv_sim_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    assert shas == {h.hexdigest()[:16]}, "profiles/pmc_latest.json was taken on other kernel sources"


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
    monkeypatch.setattr(sys.modules["bench_sides.common"], "PMC_FILE", str(f))  # (the counter passes are read by bench_sides/common.py)
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


def test_final_line_is_compact_and_complete(tmp_path, monkeypatch):
    """the driver parses the LAST stdout line: <= 4 KB, flat scalars, with roofline.frac and cpu_baseline.value; the full
    record (every side measurement) goes to bench_sides.json.  Canned input: round 4's full 23.8 KB record."""
    b = _bench()
    full = json.load(open(os.path.join(REPO, "profiles", "r4e", "bench.json")))
    assert len(json.dumps(full)) > 20000
    monkeypatch.setattr(b, "SIDES_FILE", str(tmp_path / "bench_sides.json"))
    import io
    import contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        b.emit(full)
    lines = buf.getvalue().strip().split("\n")
    assert len(lines) == 1 and len(lines[0]) <= 4096
    line = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["value"] > 0 and line["unit"] == "expansions/s" and line["dtype"] == "f64" and line["vs_baseline"] is None
    assert len(line["config"]["workload"]) <= 120 and "model" not in line["config"]
    r = line["roofline"]
    assert r["bound"] in ("hbm", "valu_issue", "latency") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6 and "traffic" in r
    assert all(not isinstance(v, (dict, list)) for v in r.values())            # flat scalars only
    assert "nn_long_frac" in r and "side_astar_cells_per_s" in r and "leaf_kernel_ms" in r
    c = line["cpu_baseline"]
    assert c["value"] > 0 and c["cores"] == 1 and c["kind"] == "port" and c["sample"] and c["reference_value"] > 0
    assert all(not isinstance(v, (dict, list)) for v in c.values())
    assert line["sides_file"] == "bench_sides.json"
    assert json.load(open(tmp_path / "bench_sides.json")) == full               # nothing is lost: the sides are in the file
    # values keep 7 significant digits
    assert abs(line["value"] / full["value"] - 1.0) < 1e-6


def test_final_line_survives_oversized_fields_and_missing_parts():
    b = _bench()
    out = {"metric": "m", "value": 1.0, "unit": "expansions/s", "n_gpus": 8, "steps": 2, "warmup": 1, "ms_per_step": 3.0,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "w" * 500, "gather_note": "x" * 9000, "rccl_comm_info_per_rank": [[8, i, 8] for i in range(8)]},
           "roofline": dict({"bound": "hbm", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 1.25e-4, "traffic": None,
                             "kernel": "k" * 300, "traffic_source": "s" * 5000}, **{k: "y" * 200 for k in b.ROOF_KEEP[6:]}),
           "cpu_baseline": None}
    line = b.compact_line(out, None)
    s = json.dumps(line, separators=(",", ":"))
    assert len(s) <= 4096 and json.loads(s)["roofline"]["frac"] == 1.25e-4 and line["cpu_baseline"] is None
    assert "gather_note" not in line["config"] and "traffic_source" not in line["roofline"] and line["roofline"]["traffic"] is None


def test_bound_names_the_roof_that_binds():
    b = _bench()
    r = b.roofline(8e7, 1.0, "k")                                                  # 0.01 of HBM, NO counters: not labelled "latency"
    assert r["bound"] == "hbm" and "vector-issue share unknown" in r["bound_note"]  # (an issue-bound kernel would be mislabelled)
    assert b.roofline(8e7, 1.0, "k", valu_issue_frac=0.1)["bound"] == "latency"
    assert b.roofline(8e7, 1.0, "k", valu_issue_frac=0.6)["bound"] == "valu_issue"
    assert b.roofline(6e9, 1.0, "k", valu_issue_frac=0.3)["bound"] == "hbm"


def test_compact_string_never_exceeds_the_limit_and_never_raises():
    """emit() used to assert on the line length: a long free-text field must cost the field, not the result"""
    b = _bench()
    out = {"metric": "m", "value": 2.5, "unit": "expansions/s", "n_gpus": 8, "steps": 2, "warmup": 1, "ms_per_step": 3.0,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "w" * 500, "gather": "g" * 3000, "gather_mode": "m" * 3000, "parallelism": "p" * 3000,
                      "gather_bytes_per_rank": 1.2e8, "gather_ms_over_step": 0.03},
           "roofline": {"bound": "valu_issue", "achieved": 926.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.116, "traffic": 1.0e11},
           "cpu_baseline": {"value": 3235.0, "unit": "expansions/s", "cores": 1, "kind": "port", "sample": "s" * 4000}}
    s = b.compact_string(out, None)
    line = json.loads(s)
    assert len(s) <= b.LINE_LIMIT and line["value"] == 2.5 and line["roofline"]["frac"] == 0.116
    assert line["config"]["gather_bytes_per_rank"] == 1.2e8 and line["config"]["gather_ms_over_step"] == 0.03
