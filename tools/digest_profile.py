#!/usr/bin/env python3
"""Copy the judged summaries of a tools/profile_bench.sh run from gpurun_out/prof_<tag>/ into profiles/<tag>/ (tracked)
and refresh profiles/pmc_latest.json (read by bench.py for roofline.traffic).

Per kernel: the mean of every counter over the kernel's dispatches in the counter passes, and HBM bytes per launch =
2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes).  The x2 is MI355X_MICROARCH.md's gfx950 correction (FETCH_SIZE tallies
128-B requests at 64 B); the guide calibrates it for 16-B-per-lane streaming reads and calls other widths
uncalibrated, so the raw sum is kept next to it."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(REPO, "gpurun_out", "prof_" + tag)
dst = os.path.join(REPO, "profiles", tag)
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "trace_main", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, os.path.basename(f)))
for f in ("bench.json", "trace_bench.json", "trace_main_bench.json"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
for f in glob.glob(os.path.join(src, "trace", "*kernel_stats.csv")) + glob.glob(os.path.join(src, "trace", "*domain_stats.csv")):
    shutil.copy(f, os.path.join(dst, os.path.basename(f)))

# (kernel key, substring of the kernel name, pass-directory prefix, bench json of that pass, path to its work units)
# the headline pass is two launches: the expansion kernel (rows or one-episode variant) and the leaf pass; their counters
# are summed into "rrt_exploring" (what bench.py's roofline.traffic refers to) and kept separately as well
KERNELS = [("rrt_rows_kernel", "rrt_rows_kernel", "pmc_", None), ("rrt_explore_kernel", "rrt_explore_kernel", "pmc_", None),
           ("rrt_leaf_kernel", "rrt_leaf_kernel", "pmc_", None),
           ("astar_kernel", "astar_kernel", "pmc_astar_", "astar"),
           ("prrt_kernel", "prrt_kernel", "pmc_planner_rrt_", "planner_rrt")]
out = {"tag": tag, "kernels": {}}
for key, needle, prefix, side in KERNELS:
    pmc = collections.defaultdict(list)
    for d in glob.glob(os.path.join(src, prefix + "*")):
        if not os.path.isdir(d):
            continue
        base = os.path.basename(d)[len(prefix):]
        if side is None and (base.startswith("astar_") or base.startswith("planner_rrt_")):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if needle in r["Kernel_Name"]:
                    pmc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not pmc:
        continue
    # the headline pass launches the kernel once (--steps 1 --warmup 0); the side passes launch it warm-up + steps
    # times with identical work, so the mean is the per-launch figure either way
    summary = {k: sum(v) / len(v) for k, v in pmc.items()}
    rec = {"per_launch": summary, "dispatches_seen": {k: len(v) for k, v in pmc.items()}}
    if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
        rd, wr = summary["FETCH_SIZE"] * 1024.0, summary["WRITE_SIZE"] * 1024.0
        rec["hbm_read_bytes_raw"] = rd
        rec["hbm_write_bytes"] = wr
        rec["hbm_bytes_per_launch_raw"] = rd + wr
        rec["hbm_bytes_per_launch"] = 2 * rd + wr
    # work units of the profiled launch, so bench.py can scale the traffic to a different batch
    try:
        if side is None:
            j = json.loads(open(os.path.join(src, "pmc_FETCH_SIZE.json")).read().strip().splitlines()[-1])
            rec["units"] = j["expansions_per_s_kernel_only"] * j["roofline"]["kernel_ms"] * 1e-3
            rec["algorithmic_bytes_per_launch"] = j["roofline"]["algorithmic_bytes_per_launch"]
        else:
            j = json.loads(open(os.path.join(src, prefix + "FETCH_SIZE.json")).read().strip().splitlines()[-1])[side]
            rec["units"] = j["cells_per_step"] if side == "astar" else j["planner_steps_per_step"]
            rec["algorithmic_bytes_per_launch"] = j["roofline"]["algorithmic_bytes_per_launch"]
    except Exception as e:
        rec["units_error"] = str(e)
    if "SQ_INSTS_VALU" in summary and rec.get("units"):
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
            if c in summary:
                rec[c.lower() + "_per_unit"] = summary[c] / rec["units"]
    if "SQ_THREAD_CYCLES_VALU" in summary and "SQ_ACTIVE_INST_VALU" in summary and summary["SQ_ACTIVE_INST_VALU"] > 0:
        # lanes enabled in the exec mask per VALU instruction, out of 64 (normalisation checked on a plain copy kernel,
        # which reads 0.96): exec-mask occupancy, an upper bound of the lanes doing useful work
        rec["valu_exec_mask_occupancy"] = summary["SQ_THREAD_CYCLES_VALU"] / (summary["SQ_ACTIVE_INST_VALU"] * 64.0)
    out["kernels"][key] = rec
parts = [out["kernels"][k] for k in ("rrt_rows_kernel", "rrt_explore_kernel", "rrt_leaf_kernel") if k in out["kernels"]]
if parts:
    tot = {"per_launch": {}, "kernels": [k for k in ("rrt_rows_kernel", "rrt_explore_kernel", "rrt_leaf_kernel") if k in out["kernels"]]}
    for c in set().union(*[p["per_launch"].keys() for p in parts]):
        tot["per_launch"][c] = sum(p["per_launch"].get(c, 0.0) for p in parts)
    for f in ("hbm_read_bytes_raw", "hbm_write_bytes", "hbm_bytes_per_launch_raw", "hbm_bytes_per_launch"):
        if all(f in p for p in parts):
            tot[f] = sum(p[f] for p in parts)
    for f in ("units", "algorithmic_bytes_per_launch"):
        if f in parts[0]:
            tot[f] = parts[0][f]
    out["kernels"]["rrt_exploring"] = tot
json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
json.dump(out, open(os.path.join(REPO, "profiles", "pmc_latest.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
