#!/usr/bin/env python3
"""Copy the judged summaries of a tools/profile_bench.sh run from gpurun_out/prof_<tag>/ into profiles/<tag>/ (tracked)
and refresh profiles/pmc_latest.json (read by bench.py for roofline.traffic).

Per measurement (the headline and every profiled side measurement): the counters of the kernels that measurement
launches, averaged over the dispatches of its counter passes and summed over its kernels, and the HBM bytes per launch in
two explicitly named forms:
  hbm_bytes_raw        FETCH_SIZE + WRITE_SIZE (KiB -> bytes), as the counters read
  hbm_bytes_fetch_x2   2 x FETCH_SIZE + WRITE_SIZE: MI355X_MICROARCH.md's gfx950 correction (FETCH_SIZE tallies 128-B
                       requests at 64 B).  The guide calibrates it for 16-B-per-lane streaming reads (the nearest-neighbour
                       scan is exactly that) and calls other access widths uncalibrated (the tree kernels' 8-B scattered
                       reads): bench.py reports both.
A measurement records the kernel names its counters came from; bench.py only attaches the traffic to a roofline whose
launch ran the same kernels."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
KEEP_STALE = "--keep-stale" in sys.argv[2:]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha():
    """SHA-256 over the kernel sources (auv_sim_amd/csrc/*, names and bytes): what a counter pass was taken on"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(REPO, "auv_sim_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if not f.endswith((".h", ".hip")):
            continue
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


CSRC = csrc_sha()
src = os.path.join(REPO, "gpurun_out", "prof_" + tag)
dst = os.path.join(REPO, "profiles", tag)
os.makedirs(dst, exist_ok=True)
for d in glob.glob(os.path.join(src, "trace*")):
    if os.path.isdir(d):
        for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True) + glob.glob(os.path.join(d, "**", "*domain_stats.csv"), recursive=True):
            shutil.copy(f, os.path.join(dst, os.path.basename(f)))
for f in glob.glob(os.path.join(src, "*.json")):
    if os.path.basename(f).startswith(("bench", "trace")):
        d = os.path.join(dst, os.path.basename(f))
        if not os.path.exists(d) or os.path.getmtime(f) > os.path.getmtime(d):  # (a later bench line may have been put there by hand)
            shutil.copy2(f, d)

# measurement key (= pass-directory name pmc@<key>@<counter>; "headline" or the side's name in the bench JSON) ->
#   (kernel-name needles, field of the side's JSON holding its work units)
RRT = ("rrt_rows_kernel", "rrt_rows_stream_kernel", "rrt_stream_kernel", "rrt_explore_kernel", "rrt_leaf_kernel")
MEAS = {
    "headline": (RRT, None),
    "astar": (("astar_kernel",), "cells_per_step"),
    "planner_rrt": (("prrt_kernel", "prrt_rows_kernel", "prrt_pipe_kernel", "prrt_duo_kernel"), "planner_steps_per_step"),
    "rrt_nn": (RRT, "iters_per_launch"),
    "rrt_nn_long_horizon": (RRT, "iters_per_launch"),
    "config5": (("prrt_kernel", "prrt_rows_kernel"), "planner_steps_per_tracking_step"),
    "single_episode": (("rrt_trio_kernel", "rrt_duo_kernel", "rrt_explore_kernel", "rrt_leaf_kernel"), "iters"),
    "rrt_1024_replicas": (("rrt_trio_kernel", "rrt_duo_kernel", "rrt_explore_kernel", "rrt_rows_kernel", "rrt_leaf_kernel"), "iters_per_launch"),
    "particle_filter": (("pf_step_kernel",), None),
    "shark_grid": (("sog_count_kernel", "sog_occ_kernel", "sog_grid_kernel", "sog_grid_tile_kernel", "sog_grid_tile_c_kernel"), None),
}


def short(name):
    """auvp::rrt_explore_kernel<4, 2, false>(...) -> rrt_explore_kernel (the exact function name: prrt_rows_kernel is not
    rrt_rows_kernel)"""
    import re
    m = re.search(r"auvp::([A-Za-z0-9_]+)", name)
    return m.group(1) if m else name.split("(")[0][:60]


out = {"tag": tag, "measurements": {}}
for key, (needles, units_field) in MEAS.items():
    side = None if key == "headline" else key
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(glob.glob(os.path.join(src, "pmc@%s@*" % key))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if short(r["Kernel_Name"]) in needles:
                    per_kernel[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not per_kernel:
        continue
    rec = {"kernels": {}, "per_launch": collections.defaultdict(float)}
    if "rrt_rows_stream_kernel" in per_kernel and "rrt_rows_kernel" in per_kernel:
        # the first batch with a parameter block runs rrt_rows_kernel (and reports how many random numbers the episodes
        # draw), the following ones rrt_stream_kernel + rrt_rows_stream_kernel: the measurement is of those
        del per_kernel["rrt_rows_kernel"]
        rec["first_batch_kernel_dropped"] = "rrt_rows_kernel"
    for kn, ctrs in per_kernel.items():
        # launches per measurement step: a kernel that runs L times per step shows L x (warm-up + steps) dispatches; the
        # mean per dispatch x L would need L -- every kernel here runs once per launch of its measurement
        km = {c: sum(v) / len(v) for c, v in ctrs.items()}
        rec["kernels"][kn] = {"per_launch": km, "dispatches_seen": {c: len(v) for c, v in ctrs.items()}}
        for c, v in km.items():
            rec["per_launch"][c] += v
    rec["per_launch"] = dict(rec["per_launch"])
    s = rec["per_launch"]
    if "FETCH_SIZE" in s and "WRITE_SIZE" in s:
        rd, wr = s["FETCH_SIZE"] * 1024.0, s["WRITE_SIZE"] * 1024.0
        rec["hbm_read_bytes_raw"], rec["hbm_write_bytes"] = rd, wr
        rec["hbm_bytes_raw"] = rd + wr
        rec["hbm_bytes_fetch_x2"] = 2 * rd + wr
    try:
        if side is None:
            full = os.path.join(src, "pmc@headline@FETCH_SIZE.sides.json")  # the full record (the stdout line is compact)
            if os.path.exists(full):
                j = json.load(open(full))
            else:
                j = json.loads(open(os.path.join(src, "pmc@headline@FETCH_SIZE.json")).read().strip().splitlines()[-1])
            rec["units"] = j["expansions_per_step"]
            # whole pass (expansion + leaf launch): SURVEY 8(d) B_exp x expansions; per kernel: roofline.algorithmic_bytes_per_launch
            # (expansion kernel) and roofline.leaf_compulsory_bytes
            rec["algorithmic_bytes_per_launch"] = j["roofline"]["pass_8d_bytes"]
            rec["expansion_kernel_algorithmic_bytes"] = j["roofline"]["algorithmic_bytes_per_launch"]
            rec["leaf_kernel_compulsory_bytes"] = j["roofline"]["leaf_compulsory_bytes"]
        else:
            j = json.loads(open(os.path.join(src, "pmc@%s@FETCH_SIZE.json" % key)).read().strip().splitlines()[-1])[side]
            if units_field:
                rec["units"] = j[units_field]
            if "roofline" in j:
                rec["algorithmic_bytes_per_launch"] = j["roofline"]["algorithmic_bytes_per_launch"]
    except Exception as e:
        rec["units_error"] = "%s: %s" % (type(e).__name__, e)
    if "SQ_INSTS_VALU" in s and rec.get("units"):
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
            if c in s:
                rec[c.lower() + "_per_unit"] = s[c] / rec["units"]
    for kn, kr in rec["kernels"].items():
        km = kr["per_launch"]
        if "SQ_THREAD_CYCLES_VALU" in km and km.get("SQ_ACTIVE_INST_VALU", 0) > 0:
            # lanes enabled in the exec mask per VALU instruction, out of 64 (a plain copy kernel reads 0.96)
            kr["valu_exec_mask_occupancy"] = km["SQ_THREAD_CYCLES_VALU"] / (km["SQ_ACTIVE_INST_VALU"] * 64.0)
        if "SQ_INSTS_VALU" in km and rec.get("units"):
            kr["sq_insts_valu_per_unit"] = km["SQ_INSTS_VALU"] / rec["units"]
        if "SQ_INSTS_SALU" in km and rec.get("units"):
            kr["sq_insts_salu_per_unit"] = km["SQ_INSTS_SALU"] / rec["units"]
        if km.get("SQ_WAVE_CYCLES", 0) > 0 and "SQ_WAIT_ANY" in km:
            # share of the resident wavefronts' cycles spent waiting (any reason): latency kernels sit here
            kr["wait_any_share"] = km["SQ_WAIT_ANY"] / km["SQ_WAVE_CYCLES"]
        if km.get("SQ_WAVE_CYCLES", 0) > 0 and "SQ_ACTIVE_INST_VALU" in km:
            kr["valu_active_share"] = km["SQ_ACTIVE_INST_VALU"] / km["SQ_WAVE_CYCLES"]
    out["measurements"][key] = rec
json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
# pmc_latest.json keeps the measurements this run did not profile (a run may cover the headline and a few sides only), each
# with the tag of the run it came from
latest_path = os.path.join(REPO, "profiles", "pmc_latest.json")
# (round 6) ... but NOT when the kernel sources changed since: a measurement taken on other sources is dropped (its numbers stay
# in profiles/<its tag>/pmc_summary.json), unless --keep-stale.  Every measurement carries the hash of the sources it was taken on.
latest = {"tag": tag, "csrc_sha": CSRC, "measurements": {}}
dropped = []
if os.path.exists(latest_path):
    try:
        old = json.load(open(latest_path))
        for k, v in old.get("measurements", {}).items():
            v.setdefault("from_tag", old.get("tag"))
            if k not in out["measurements"] and v.get("csrc_sha") != CSRC and not KEEP_STALE:
                dropped.append("%s (from %s)" % (k, v.get("from_tag")))
                continue
            latest["measurements"][k] = v
    except (ValueError, OSError):
        pass
for k, v in out["measurements"].items():
    v = dict(v)
    v["from_tag"] = tag
    v["csrc_sha"] = CSRC
    latest["measurements"][k] = v
if dropped:
    print("digest_profile: dropped from pmc_latest.json (taken on other kernel sources; re-profile or --keep-stale): " + ", ".join(dropped), file=sys.stderr)
json.dump(latest, open(latest_path, "w"), indent=1)
print(json.dumps({k: {f: v.get(f) for f in ("hbm_bytes_raw", "hbm_bytes_fetch_x2", "algorithmic_bytes_per_launch", "units", "sq_insts_valu_per_unit")}
                  for k, v in out["measurements"].items()}, indent=1))
