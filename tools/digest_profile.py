#!/usr/bin/env python3
"""Copy the judged summaries of a tools/profile_bench.sh run from gpurun_out/prof_<tag>/ into
profiles/<tag>/ (tracked) and refresh profiles/pmc_latest.json (read by bench.py for roofline.traffic)."""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "rrt_explore"
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
pmc = collections.defaultdict(list)
for f in glob.glob(os.path.join(src, "pmc_*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]:
            pmc[r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {k: sum(v) / len(v) for k, v in pmc.items()}
out = {"tag": tag, "kernel": kernel, "per_launch": summary}
if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
    # rocprofv3 reports both in KiB.  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE counts 64 B per 128-B
    # request for wide (16 B/lane) streaming reads -> x2; this kernel's reads are 8-B scattered/gather
    # accesses, for which the guide calls the counter uncalibrated, so both figures are kept.
    rd, wr = summary["FETCH_SIZE"] * 1024.0, summary["WRITE_SIZE"] * 1024.0
    out["hbm_read_bytes_raw"] = rd
    out["hbm_write_bytes"] = wr
    out["hbm_bytes_per_launch"] = rd + wr
    out["hbm_bytes_per_launch_fetch_x2"] = 2 * rd + wr
    out["note"] = "FETCH_SIZE/WRITE_SIZE KiB->bytes; fetch x2 correction applies to 16-B/lane streaming reads only"
json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
json.dump(out, open(os.path.join(REPO, "profiles", "pmc_latest.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
