#!/usr/bin/env python3
"""bench.py -- RRT-Dubins node expansions/s on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic input: E independent RRT.exploring episodes
(path_planning/rrt_dubins.py:92) x `--iters` expansions each, on the 256-obstacle 200x200-cell Catalina-like grid of
SURVEY.md 8(d) config 2, followed by the extraction of every episode's best path and -- for N > 1 -- the RCCL gather
of the result records (libauvplan.so's own auvp_gather entry points, on the planner's HIP stream).  Inputs (world
tables, start states, seeded MT19937 states) are resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W          (N > 1: spawns its own N ranks, one per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  value = expansions of ALL ranks / max-over-ranks time.  The other configurations of
BASELINE.json ride along as side measurements in the same line (each with its own step time, kernel time, roofline
and -- at N = 1 -- CPU baseline); for N > 1 the A* batch (config 3) and the Planner_RRT batch (config 4) are sharded
over the ranks like the headline and gathered the same way.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
REF_TIMING = [os.path.join(REPO, "profiles", f) for f in ("r3_reference_timing.json", "r2_reference_timing.json")]
PMC_FILE = os.path.join(REPO, "profiles", "pmc_latest.json")


# ----------------------------------------------------------------------------------------------------------------
# algorithmic bytes, always from the launch's own counters
# ----------------------------------------------------------------------------------------------------------------
def rrt_bytes(summ):
    """B_exp: 48 parent read + 4 bin-index read per expansion; 52 node write + 8 bin append per accepted node;
    56 per stored path point; (24 + 8) per path element walked by the cost function; nearest-neighbour sampling adds
    16 B (x, y) per node of mps_list per iteration ("NN mode adds 16 N per iteration"), counted by the
    kernel as the sum of len(mps_list) over its scans."""
    iters = float(summ["iters_run"].sum())
    nodes = float((summ["n_nodes"] - 1).sum())
    pts = float(summ["n_points"].sum())
    walked = float(summ["leaf_elems"].sum())
    scanned = float(summ["nn_scanned"].sum())
    return iters * (48 + 4) + nodes * (52 + 8) + pts * 56 + walked * (24 + 8) + scanned * 16


def rrt_expand_bytes(summ):
    """the expansion kernel's share of B_exp: 48 parent read + 4 bin-index read per iteration; 52 node write +
    8 bin append per accepted node; 56 per stored path point; nearest-neighbour sampling: 16 per node scanned"""
    iters = float(summ["iters_run"].sum())
    nodes = float((summ["n_nodes"] - 1).sum())
    return iters * 52 + nodes * 60 + float(summ["n_points"].sum()) * 56 + float(summ["nn_scanned"].sum()) * 16


def rrt_leaf_bytes(summ, st):
    """COMPULSORY bytes of the leaf pass (rrt_leaf_kernel), every tree element at most once -- SURVEY 8(d) bills q L 32 bytes
    per expansion for the leaf->root walks of the qualifying leaves, but a path element's cost term does not depend on the
    leaf, so the pass evaluates each element of the visited part of the tree ONCE:
      every node: parent link + qualifying flag (the backward marking sweep)                     16 + 1
      every visited node (a qualifying leaf or an ancestor of one): link record, x y t length, the parent's running
        sums read, its own term and sums written                                              16 + 32 + 32 + 16 + 32
      every path point of a visited node: x, y, t                                                     24
      every element re-summed in the reference's order (the record setters): x, y, t + its node's share    24
    `st` = ctx.last_leaf_stats() of the same launch."""
    return (float(summ["n_nodes"].sum()) * 17 + st["nodes_visited"] * 128.0 + st["points_visited"] * 24.0 +
            st["elements_resummed"] * 24.0)


def planner_bytes(summ):
    """Planner_RRT step: 48 parent read + 4 bucket-index read per step; 52 node write + 8 bucket append per accepted
    node; 56 per stored path point (goal-arc points are transient)."""
    steps = float(summ["steps"].sum())
    nodes = float((summ["n_nodes"] - 1).sum())
    pts = float(summ["n_points"].sum())
    return steps * (48 + 4) + nodes * (52 + 8) + pts * 56


def astar_bytes(summ, variant):
    """per child cell: node write 68 (44 for astar.py) + visited flag 1 + SOG 16 (cell prob + top-n prefix); per pop:
    8 bytes per open-list entry the min-f scan reads (sum of len(open_list) over the pops, counted by the kernel)."""
    cells = float(summ["n_children"].sum())
    scanned = float(summ["open_scanned"].sum())
    per_cell = {"astar": 44.0, "astar_real": 44.0, "astar_fixLen": 69.0, "astar_fixLenSOG": 85.0}[variant]
    return cells * per_cell + scanned * 8.0


LATENCY_FRAC, LATENCY_VALU = 0.05, 0.25  # below both: `bound` = "latency"
N_SIMD = 1024  # 256 CUs x 4 SIMDs (MI355X_MICROARCH.md)
HBM_MEASURED = {"read_GBps": None, "copy_GBps": None}  # filled once per run by measure_hbm() (auvp_hbm_probe)


def measure_hbm(ctx, n_bytes=4 << 30, reps=3):
    """the MEASURED HBM roof of this GPU (north star: "fraction of the measured HBM roofline"): a timed streaming read of
    4 GiB with eight 16-byte loads in flight per lane (the nearest-neighbour scan's access shape) and a 16-byte copy, HIP
    events on the planner's stream (libauvplan.so: auvp_hbm_probe)"""
    try:
        r, c = ctx.hbm_probe(n_bytes, reps)
        HBM_MEASURED.update(read_GBps=r, copy_GBps=c, bytes=int(n_bytes), reps=int(reps))
    except Exception as e:  # the probe must not cost the headline line
        HBM_MEASURED.update(error="%s: %s" % (type(e).__name__, e))
    return HBM_MEASURED


def _pmc():
    try:
        return json.load(open(PMC_FILE))
    except Exception:
        return None


def pmc_valu_issue(meas, kernel=None):
    """fraction of the chip's vector-issue slots the profiled launch used: SQ_INSTS_VALU x 4 cycles / (1 024 SIMDs x
    GRBM_GUI_ACTIVE / 8 XCDs) -- a wave64 instruction occupies its SIMD-32 for >= 4 cycles when one wave issues back to back
    (fp64 and transcendental instructions take longer, so this is a LOWER bound of the pipe's occupancy).  From the committed
    PMC passes (profiles/pmc_latest.json); `kernel`: one kernel of the measurement, None: all of them."""
    pj = _pmc()
    try:
        m = pj["measurements"][meas]
        c = m["kernels"][kernel]["per_launch"] if kernel else m["per_launch"]
        return 4.0 * float(c["SQ_INSTS_VALU"]) / (N_SIMD * float(c["GRBM_GUI_ACTIVE"]) / 8.0)
    except Exception:
        return None


def pmc_latency(meas, kernel_prefix, units_now):
    """counters of a LATENCY measurement (a few dependent chains: one episode, 1 024 replicas) from the committed PMC passes:
    vector / scalar instructions per expansion, the share of the resident wavefronts' cycles spent waiting, HBM bytes per
    launch.  {} when the committed passes do not hold that kernel."""
    pj = _pmc()
    try:
        m = pj["measurements"][meas]
        kn = [k for k in m["kernels"] if k.startswith(kernel_prefix)]
        if not kn:
            return {}
        k = m["kernels"][kn[0]]
        c = k["per_launch"]
        u0 = float(m.get("units") or 0.0)
        sc = units_now / u0 if (u0 > 0 and units_now) else 1.0
        out = {"pmc_kernel": kn[0], "valu_per_expansion": k.get("sq_insts_valu_per_unit"), "salu_per_expansion": k.get("sq_insts_salu_per_unit"),
               "wait_any_share": k.get("wait_any_share"), "valu_active_share": k.get("valu_active_share")}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            f, w = 1024.0 * float(c["FETCH_SIZE"]), 1024.0 * float(c["WRITE_SIZE"])
            out["traffic"], out["traffic_raw"] = (2.0 * f + w) * sc, (f + w) * sc
        out["traffic_source"] = "profiles/%s pmc@%s" % (m.get("from_tag") or pj.get("tag", "?"), meas)
        return {k2: v for k2, v in out.items() if v is not None}
    except Exception:
        return {}


SHADER_GHZ = 2.4  # MI355X_MICROARCH.md: engine clock the latency figures are quoted in


def pmc_valu_issue_est(kernel, units_now, kernel_ms):
    """vector-issue fraction of a launch that has no counter pass of its own, from the instructions per work unit the same
    kernel showed in ANY committed pass (a property of the kernel and the workload's shape) and this launch's time:
    valu_per_unit x units x 4 cycles / (1 024 SIMDs x time x SHADER_GHZ).  None when no committed pass ran that kernel."""
    pj = _pmc()
    try:
        for m in pj["measurements"].values():
            k = m["kernels"].get(kernel)
            if k and k.get("sq_insts_valu_per_unit"):
                return 4.0 * float(k["sq_insts_valu_per_unit"]) * units_now / (N_SIMD * kernel_ms * 1e-3 * SHADER_GHZ * 1e9)
    except Exception:
        pass
    return None


def pmc_kernel_traffic(meas, kernel, units_now):
    """(2 x FETCH_SIZE + WRITE_SIZE, FETCH_SIZE + WRITE_SIZE) of ONE kernel of a profiled measurement, bytes per launch"""
    pj = _pmc()
    try:
        m = pj["measurements"][meas]
        c = m["kernels"][kernel]["per_launch"]
        f, w = 1024.0 * float(c["FETCH_SIZE"]), 1024.0 * float(c["WRITE_SIZE"])
        u0 = float(m.get("units") or 0.0)
        sc = units_now / u0 if (u0 > 0 and units_now) else 1.0
        return (2.0 * f + w) * sc, (f + w) * sc
    except Exception:
        return None, None


def roofline(abytes, k_ms, kernel, traffic=None, valu_issue_frac=None, **extra):
    """`traffic`: the dict pmc_traffic() returns (or None).  `bound` names the roof that binds: "hbm" unless the kernel's
    vector-issue fraction (pmc_valu_issue) exceeds its HBM fraction -- then "valu_issue" (fp64 VALU issue; no MFMA work on
    this path).  achieved / peak / frac are always the HBM figures (GB/s against the 8 TB/s datasheet peak);
    frac_of_measured is against this GPU's measured streaming read rate."""
    ach = abytes / (k_ms * 1e-3) / 1e9
    frac = ach / HBM_PEAK_GBS
    r = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac,
         "traffic": None, "kernel": kernel, "kernel_ms": k_ms, "algorithmic_bytes_per_launch": abytes}
    if HBM_MEASURED.get("read_GBps"):
        r["hbm_measured_GBps"] = HBM_MEASURED["read_GBps"]
        r["frac_of_measured"] = ach / HBM_MEASURED["read_GBps"]
    if valu_issue_frac is not None:
        r["valu_issue_frac"] = valu_issue_frac
        if valu_issue_frac > frac:
            r["bound"] = "valu_issue"
    # neither roof is near: a launch of a few dependent chains (one episode, 1 024 replicas, a dense small tree) is bound
    # by the latency of its serial chain, not by a throughput roof -- say so instead of "hbm" at a fraction of a few percent
    # (only with the counters in hand: without a vector-issue figure an issue-bound kernel would be mislabelled -- it then stays
    # "hbm" with its small fraction and a note)
    if frac < LATENCY_FRAC and valu_issue_frac is not None and valu_issue_frac < LATENCY_VALU:
        r["bound"] = "latency"
    elif frac < LATENCY_FRAC and valu_issue_frac is None:
        r["bound_note"] = "no PMC pass for this kernel: far from the HBM roof, vector-issue share unknown"
    r.update(traffic or {})
    r.update(extra)
    return r


def pmc_traffic(meas, kernels_ran, units_now):
    """HBM bytes per launch of measurement `meas` from the committed PMC passes (tools/profile_bench.sh ->
    profiles/pmc_latest.json; FETCH_SIZE and WRITE_SIZE collected in separate passes), scaled by the work units when this
    run's batch differs from the profiled one.  Returns {"traffic": 2 x FETCH_SIZE + WRITE_SIZE (the microarchitecture
    guide's gfx950 correction, calibrated for 16-B/lane streaming reads), "traffic_raw": FETCH_SIZE + WRITE_SIZE,
    "traffic_source": ...}; all None when the profiled launch ran other kernels than this one (`kernels_ran`)."""
    none = {"traffic": None, "traffic_raw": None, "traffic_source": None}
    try:
        pj = json.load(open(PMC_FILE))
        m = pj["measurements"][meas]
        if set(kernels_ran) != set(m["kernels"].keys()):
            none["traffic_source"] = "profiles/%s profiled %s, this launch ran %s: not comparable" % (
                m.get("from_tag") or pj.get("tag", "?"), sorted(m["kernels"].keys()), sorted(kernels_ran))
            return none
        x2, raw = float(m["hbm_bytes_fetch_x2"]), float(m["hbm_bytes_raw"])
        src = "profiles/%s pmc@%s (traffic = 2 x FETCH_SIZE + WRITE_SIZE, traffic_raw = FETCH_SIZE + WRITE_SIZE; separate passes)" % (
            m.get("from_tag") or pj.get("tag", "?"), meas)
        u0 = float(m.get("units") or 0.0)
        if u0 > 0 and units_now and abs(u0 - units_now) > 0.5:
            x2 *= units_now / u0
            raw *= units_now / u0
            src += ", scaled x%.3f by work units" % (units_now / u0)
        return {"traffic": x2, "traffic_raw": raw, "traffic_source": src}
    except Exception:
        return none


def recorded_reference(key):
    """the reference Python's own timing on this workload, RECORDED in the build container (tests/experiments/ref_timing.py,
    profiles/r2_reference_timing.json, r3_reference_timing.json) -- the reference cannot run on the GPU box"""
    for f in REF_TIMING:
        try:
            r = json.load(open(f)).get(key)
            if r:
                return r
        except Exception:
            pass
    return None


# ----------------------------------------------------------------------------------------------------------------
# worlds
# ----------------------------------------------------------------------------------------------------------------
def bench_world(obstacles, grid):
    from auv_sim_amd import synth
    half = 0.5 * grid * 10.0
    return synth.make_world(seed=2, n_obstacles=obstacles, box=(-half, -half, half, half), cell=10.0, n_bins=10, bin_len=50,
                            n_habitats=10)


# ----------------------------------------------------------------------------------------------------------------
# CPU baselines (the checker under oracle/, libm build = the restatement pinned to the reference goldens)
# ----------------------------------------------------------------------------------------------------------------
def cpu_baseline(world, n_iter, args):
    from oracle import orc
    orc.build()
    w = orc.WorldArrays(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    init = [world["start"][0], world["start"][1], 0, 0, 0, 0]
    done, t_used, eps = 0, 0.0, 0
    budget = float(args.cpu_seconds)
    while t_used < budget and eps < 64:
        t0 = time.perf_counter()
        r = orc.rrt_explore(w, eps, n_iter, mode=args.mode, init=init, kind="libm", want_path=False)
        t_used += time.perf_counter() - t0
        done += r["iters_run"]
        eps += 1
    out = {"value": done / t_used, "unit": "expansions/s", "cores": 1, "kind": "port",
           "sample": "%d episodes x %d iterations of the same workload (seeds 0..%d), oracle/ libm build, %.1f s"
                     % (eps, n_iter, eps - 1, t_used)}
    ref = recorded_reference("config2_rrt_exploring_o%d" % args.obstacles)
    if ref:
        out["reference_recorded"] = {
            "value": ref["ref_expansions_per_s_1proc"], "unit": "expansions/s", "cores": 1, "kind": "reference",
            "where": "build container (8 vCPU Xeon 2.1 GHz), NOT this box; tests/experiments/ref_timing.py",
            "many_cores": {k: v for k, v in ref.items() if k.startswith("ref_expansions_per_s_") and k.endswith("proc")},
            "port_over_reference_same_container": ref.get("port_over_ref")}
    return out


def _cpu_episode(job):
    world, seed, n_iter, mode = job
    from oracle import orc
    w = orc.WorldArrays(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    init = [world["start"][0], world["start"][1], 0, 0, 0, 0]
    return orc.rrt_explore(w, seed, n_iter, mode=mode, init=init, kind="libm", want_path=False)["iters_run"]


def cpu_baseline_all_cores(world, n_iter, args):
    """the same checker on ALL usable host cores, one episode per core.  Forks: must precede any HIP initialisation."""
    import multiprocessing as mp
    from oracle import orc
    orc.build()
    cores, how = effective_cores()
    jobs = [(world, 1000 + s, n_iter, args.mode) for s in range(cores)]
    with mp.get_context("fork").Pool(cores) as pool:
        pool.map(_cpu_episode, [(world, 0, 10, args.mode)] * cores, chunksize=1)
        t0 = time.perf_counter()
        done = sum(pool.map(_cpu_episode, jobs, chunksize=1))
        dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "expansions/s", "cores": cores, "kind": "port",
            "sample": "%d episodes x %d iterations, one per usable core (%s; os.cpu_count() = %d), oracle/ libm build, %.1f s"
                      % (cores, n_iter, how, os.cpu_count() or 1, dt)}


def effective_cores(cap=64):
    n, how = len(os.sched_getaffinity(0)), "sched_getaffinity"
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max" and int(float(q) / float(p)) < n:
            n, how = max(1, int(float(q) / float(p))), "cgroup cpu.max"
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and q // p < n:
                n, how = max(1, q // p), "cgroup cfs quota"
        except (OSError, ValueError):
            pass
    if n > cap:
        n, how = cap, how + ", capped at %d" % cap
    return n, how



# ----------------------------------------------------------------------------------------------------------------
# the ONE line the driver parses: <= 4 KB of flat scalars; everything else goes to bench_sides.json next to bench.py
# ----------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 4096
SIDES_FILE = os.environ.get("AUVP_BENCH_SIDES", os.path.join(REPO, "bench_sides.json"))
ROOF_KEEP = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "algorithmic_bytes_per_launch",
             "traffic_raw", "valu_issue_frac", "hbm_measured_GBps", "frac_of_measured", "bytes_per_expansion",
             "leaf_kernel_ms", "leaf_compulsory_bytes", "leaf_frac", "leaf_valu_issue_frac", "leaf_traffic_raw",
             "pass_kernel_ms", "pass_8d_frac", "pass_traffic_raw",
             "nn_long_kernel_ms", "nn_long_alg_bytes", "nn_long_achieved_GBps", "nn_long_frac", "nn_long_frac_of_measured",
             "nn_long_traffic", "nn_long_traffic_raw", "nn_long_episodes", "nn_long_valu_issue_frac",
             "nn_long_hbm_only_GBps", "nn_long_hbm_only_frac", "nn_long_hbm_only_frac_of_measured",
             "side_astar_cells_per_s", "side_planner_steps_per_s", "side_config5_steps_per_s",
             "side_replicas_expansions_per_s", "side_single_episode_us_per_expansion", "side_pf_particle_steps_per_s",
             "side_shark_grid_cells_per_s")
CONFIG_KEEP = ("workload", "episodes_per_gpu", "iters", "obstacles", "cells", "parallelism", "gather", "rccl_ranks_seen",
               "gather_mode", "gather_bytes_per_rank", "gather_ms", "gather_ms_over_step")
TOP_KEEP = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")


def _short(v, digits=7):
    """scalars only, floats to `digits` significant digits (the full-precision values are in bench_sides.json)"""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, (float, np.floating)):
        return float("%.*g" % (digits, float(v)))
    if isinstance(v, np.integer):
        return int(v)
    return None


def compact_line(out, sides_file=None):
    """The final stdout line: headline scalars, a FLAT `roofline` (scalars only), a flat `cpu_baseline`; <= LINE_LIMIT bytes.
    Keys are dropped from the end of ROOF_KEEP (never `frac` / `achieved`) should a value ever push the line over."""
    line = {k: _short(out.get(k)) for k in TOP_KEEP}
    cfg = out.get("config") or {}
    line["config"] = {k: _short(cfg.get(k)) for k in CONFIG_KEEP if k in cfg}
    if isinstance(line["config"].get("workload"), str):
        line["config"]["workload"] = line["config"]["workload"][:120]
    roof = out.get("roofline") or {}
    line["roofline"] = {k: _short(roof[k]) for k in ROOF_KEEP if k in roof and (k == "traffic" or _short(roof[k]) is not None)}
    if isinstance(line["roofline"].get("kernel"), str):
        line["roofline"]["kernel"] = line["roofline"]["kernel"][:48]
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        c = {k: _short(cb.get(k)) for k in ("value", "unit", "cores", "kind")}
        c["sample"] = str(cb.get("sample", ""))[:160]
        rr = cb.get("reference_recorded")
        if isinstance(rr, dict):  # the reference Python itself, timed in the build container (it cannot travel)
            c["reference_value"] = _short(rr.get("value"))
            c["reference_cores"] = _short(rr.get("cores"))
        allc = out.get("cpu_baseline_all_cores")
        if isinstance(allc, dict):
            c["all_cores_value"] = _short(allc.get("value"))
            c["all_cores"] = _short(allc.get("cores"))
        line["cpu_baseline"] = c
    else:
        line["cpu_baseline"] = None
    line["sides_file"] = os.path.basename(sides_file) if sides_file else None
    drop = [k for k in reversed(ROOF_KEEP) if k not in ("bound", "achieved", "peak", "unit", "frac", "traffic")]
    while len(json.dumps(line, separators=(",", ":"))) > LINE_LIMIT and drop:
        line["roofline"].pop(drop.pop(0), None)
    return line


def emit(out):
    """full record -> SIDES_FILE (and AUVP_BENCH_VERBOSE=1: stderr); compact line -> the LAST line of stdout"""
    sides_file = SIDES_FILE
    try:
        with open(sides_file, "w") as f:
            json.dump(out, f, indent=1)
            f.write("\n")
    except OSError as e:
        print("bench: could not write %s: %s" % (sides_file, e), file=sys.stderr)
        sides_file = None
    if os.environ.get("AUVP_BENCH_VERBOSE") == "1":
        print(json.dumps(out), file=sys.stderr)
    sys.stderr.flush()
    sys.stdout.flush()
    print(compact_string(out, sides_file), flush=True)


def compact_string(out, sides_file):
    """the compact line as a string of at most LINE_LIMIT bytes: should a string field ever push it over (compact_line already
    drops roofline keys from the end), the free-text fields are cut, then dropped -- a long line must not cost the result"""
    line = compact_line(out, sides_file)
    s = json.dumps(line, separators=(",", ":"))
    for cut in (80, 40, 0):
        if len(s) <= LINE_LIMIT:
            break
        if isinstance(line.get("cpu_baseline"), dict):
            line["cpu_baseline"]["sample"] = str(line["cpu_baseline"].get("sample", ""))[:cut]
        for k in ("workload", "gather", "gather_mode", "parallelism"):
            if isinstance(line["config"].get(k), str):
                line["config"][k] = line["config"][k][:max(cut, 24)]
        s = json.dumps(line, separators=(",", ":"))
    if len(s) > LINE_LIMIT:
        line["cpu_baseline"] = None
        s = json.dumps(line, separators=(",", ":"))
    return s


# ----------------------------------------------------------------------------------------------------------------
# distributed plumbing
# ----------------------------------------------------------------------------------------------------------------
class Ranks:
    """rank bookkeeping + the result gather.  N = 1: everything is a no-op."""

    def __init__(self, ctx, rank, world, dev, use_rccl=True):
        self.ctx, self.rank, self.world, self.dev = ctx, rank, world, dev
        self.gather, self.gather_note = None, None
        self.rccl_info, self.rccl_library = None, None
        self.cpu_group = not use_rccl  # gloo: collectives on host tensors
        if world > 1 and not use_rccl:
            from auv_sim_amd import distributed as D
            self.gather = _HostGather(D.TorchGather(), dev)
            self.gather_note = "AUVP_BENCH_ONE_GPU: gloo transport"
        elif world > 1:
            import torch.distributed as dist
            from auv_sim_amd import distributed as D

            def exchange(mine):
                box = [mine]
                dist.broadcast_object_list(box, src=0)
                return box[0]
            import torch
            # pre-flight on every rank, agreed on by all, BEFORE the collective communicator initialisation
            pre = torch.tensor([0 if D.RcclGather.usable() else 1], device=dev)
            dist.all_reduce(pre)
            if int(pre.item()) == 0:
                try:
                    self.gather = D.RcclGather(ctx, rank, world, exchange)
                except Exception as e:
                    # every rank agreed that RCCL is reachable, so the others are inside (or about to enter)
                    # ncclCommInitRank: falling back on this rank alone would leave them waiting there.  End the job: the
                    # launcher (spawn_ranks below / torchrun) takes the other ranks down and reports the failure.
                    print("bench: rank %d: communicator initialisation failed (%s); aborting the job" % (rank, e), file=sys.stderr, flush=True)
                    os._exit(3)
                self.rccl_info = self.gather.info()
                self.rccl_library = D.RcclGather.library()
            else:
                self.gather_note = "auvp_gather unavailable (RCCL not reachable through the C-ABI on %d rank(s))" % int(pre.item())
            flag = torch.tensor([0 if self.gather is not None else 1], device=dev)
            dist.all_reduce(flag)
            if int(flag.item()) != 0:  # every rank uses the same transport
                if self.gather is not None:
                    self.gather.close()
                self.gather = D.TorchGather()

    def _t(self, vals, dtype=None):
        import torch
        return torch.tensor(vals, dtype=dtype or torch.float64, device="cpu" if self.cpu_group else self.dev)

    def sync(self):
        import torch
        torch.cuda.synchronize()
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    def max_time(self, dt):
        if self.world == 1:
            return dt
        import torch
        import torch.distributed as dist
        t = self._t([dt])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, v):
        if self.world == 1:
            return float(v)
        import torch
        import torch.distributed as dist
        t = self._t([float(v)])
        dist.all_reduce(t)
        return float(t.item())

    def all(self, obj):
        if self.world == 1:
            return [obj]
        import torch.distributed as dist
        out = [None] * self.world
        dist.all_gather_object(out, obj)
        return out

    def gather_records(self, ptr, n, itemsize):
        """all ranks' fixed-stride records, read straight from the planner's device buffer"""
        if self.world == 1:
            return None
        from auv_sim_amd import distributed as D
        return self.gather.gather_records(D.device_records(ptr, n, itemsize, self.dev))

    def gather_host_records(self, arr):
        if self.world == 1:
            return None
        from auv_sim_amd import distributed as D
        return self.gather.gather_records(D.summaries_to_tensor(arr, self.dev))

    def gather_ms(self):
        return self.gather.take_ms() if self.gather is not None else None


class _HostGather:
    """TorchGather over a CPU process group (gloo) for device tensors: staged through host memory"""

    name = "torch.distributed (gloo, staged through the host)"

    def __init__(self, inner, dev):
        self.inner, self.dev = inner, dev

    def gather_records(self, records):
        return [t.to(self.dev) for t in self.inner.gather_records(records.cpu())]

    def gather_paths(self, paths, lengths):
        lens, blocks = self.inner.gather_paths(paths.cpu(), lengths.cpu())
        return [l.to(self.dev) for l in lens], [b.to(self.dev) for b in blocks]

    def take_ms(self):
        return None

    def root_begin(self, tensors, rows=None, root=0):
        return self.inner.root_begin([t.cpu() for t in tensors], rows=rows, root=root)

    def root_end(self, ticket):
        out = self.inner.root_end(ticket)
        self.last_root_ms, self.last_root_bytes = None, self.inner.last_root_bytes
        return out

    def close(self):
        pass


def timed_steps(ranks, step, steps, warmup, finish=None):
    """`finish`: what the last step left in flight (the overlapped gather of its results) -- waited for INSIDE the timed region"""
    for _ in range(warmup):
        step()
    if finish:
        finish()
    ranks.sync()
    t0 = time.perf_counter()
    last = None
    for _ in range(steps):
        last = step()
    if finish:
        finish()
    ranks.sync()
    return ranks.max_time(time.perf_counter() - t0), last


# ----------------------------------------------------------------------------------------------------------------
# side measurements
# ----------------------------------------------------------------------------------------------------------------
RRT_KW = dict(freq=30, bin_interval=5, v=2, max_traj_time=500.0, weights=(-3, -3, -4))


SIDE_KEEP = ("bound", "achieved", "peak", "unit", "frac", "kernel", "kernel_ms", "algorithmic_bytes_per_launch", "frac_of_measured",
             "valu_issue_frac", "traffic", "traffic_raw", "bytes_per_expansion", "leaf_kernel_ms", "leaf_compulsory_bytes", "leaf_frac",
             "pass_kernel_ms", "pass_8d_frac")


def rrt_pass_rooflines(ctx, summ, meas, exp_ms, leaf_ms, kname, compact=False, **extra):
    """One pass of RRT.exploring = two launches.  Returns the roofline of the DOMINANT kernel (the tree expansion: its own
    algorithmic bytes over its own HIP-event time) with the leaf pass and the whole-pass SURVEY 8(d) figure beside it as
    flat scalars (the driver's record keeps scalars of this object, not nested dicts):
      leaf_*      rrt_leaf_kernel against its COMPULSORY bytes (rrt_leaf_bytes: every tree element once)
      pass_8d_*   B_exp of SURVEY 8(d) x expansions over both launches -- the figure of rounds 1-3; it bills the leaf pass
                  32 bytes per element of EVERY qualifying leaf's path, which the pass never moves (it evaluates an element
                  once), so it overstates the bandwidth of that launch; kept for continuity, labelled"""
    iters = float(summ["iters_run"].sum())
    st = ctx.last_leaf_stats()
    a_exp, a_leaf, a_8d = rrt_expand_bytes(summ), rrt_leaf_bytes(summ, st), rrt_bytes(summ)
    whole = pmc_traffic(meas, [kname, "rrt_leaf_kernel"], iters)
    comparable = whole["traffic"] is not None
    tx2, traw = pmc_kernel_traffic(meas, kname, iters) if comparable else (None, None)
    lx2, lraw = pmc_kernel_traffic(meas, "rrt_leaf_kernel", iters) if comparable else (None, None)
    vi = pmc_valu_issue(meas, kname) if comparable else None
    if vi is None:  # (a side batch of a kernel that has a pass elsewhere: its instructions per expansion over this launch's time)
        vi = pmc_valu_issue_est(kname, iters, exp_ms)
    lvi = pmc_valu_issue(meas, "rrt_leaf_kernel") if comparable else None
    leaf_ach = a_leaf / (leaf_ms * 1e-3) / 1e9 if leaf_ms > 0 else 0.0
    pass_ach = a_8d / ((exp_ms + leaf_ms) * 1e-3) / 1e9
    r = roofline(a_exp, exp_ms, kname, {"traffic": tx2, "traffic_raw": traw, "traffic_source": whole["traffic_source"]},
                 valu_issue_frac=vi, bytes_per_expansion=a_exp / iters,
                 leaf_kernel="rrt_leaf_kernel", leaf_kernel_ms=leaf_ms, leaf_compulsory_bytes=a_leaf, leaf_achieved=leaf_ach,
                 leaf_frac=leaf_ach / HBM_PEAK_GBS, leaf_valu_issue_frac=lvi, leaf_traffic=lx2, leaf_traffic_raw=lraw,
                 leaf_nodes_visited=st["nodes_visited"], leaf_points_visited=st["points_visited"],
                 leaf_elements_resummed=st["elements_resummed"], leaf_bound="latency (scattered 264-B runs; DESIGN.md)",
                 pass_kernel_ms=exp_ms + leaf_ms, pass_8d_bytes=a_8d, pass_8d_bytes_per_expansion=a_8d / iters,
                 pass_8d_achieved=pass_ach, pass_8d_frac=pass_ach / HBM_PEAK_GBS,
                 pass_traffic=whole["traffic"], pass_traffic_raw=whole["traffic_raw"],
                 pass_8d_note="SURVEY 8(d) B_exp x expansions / both launches: bills q L 32 B of leaf->root walks the leaf pass does not move",
                 **extra)
    if HBM_MEASURED.get("read_GBps"):
        r["leaf_frac_of_measured"] = leaf_ach / HBM_MEASURED["read_GBps"]
    if compact:  # side measurements: the scalars that matter, no prose (the headline's object explains the fields)
        r = {k: r[k] for k in SIDE_KEEP if k in r}
    return r


def bench_single_episode(ctx, world, args, reps=3):
    """SURVEY 8(d) config 2 latency test: ONE episode on one GPU (a serial chain: one wavefront busy)."""
    ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    init = np.zeros((1, 6))
    init[0, 0], init[0, 1] = world["start"]
    ms = []
    for i in range(reps + 1):
        summ = ctx.rrt_explore_batch(init, np.array([7], dtype=np.uint64), args.iters, mode=args.mode, **RRT_KW)
        if i:
            ms.append(ctx.last_kernel_ms())
    k_ms = float(np.mean(ms))
    iters = float(summ[0]["iters_run"])
    exp_ms = ctx.last_launch_parts()[0]
    kname = ctx.last_rrt_kernel()
    # a latency measurement: ONE dependent chain (3-4 wavefronts of one CU) -- no throughput roof applies; what is reported is
    # the chain's length in shader clocks per iteration, with the counters of the committed pass beside it
    roof = dict({"bound": "latency", "kernel": kname, "kernel_ms": exp_ms, "clocks_per_iteration": exp_ms * 1e-3 * SHADER_GHZ * 1e9 / iters,
                 "achieved": rrt_expand_bytes(summ) / (exp_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": rrt_expand_bytes(summ) / (exp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None},
                **pmc_latency("single_episode", kname.split("<")[0], iters))
    return {"metric": "single-episode latency (seed 7)", "kernel": kname, "kernel_ms": k_ms, "iters": int(iters),
            "expansions_per_s": iters / (k_ms * 1e-3), "us_per_expansion": 1e3 * k_ms / iters, "roofline": roof}


def _rrt_batch(ctx, world, n_ep, args, reps=2, cpu_seconds=0.0, mode=None, kw=None, meas="-"):
    """`meas`: key of this measurement's committed counter passes in profiles/pmc_latest.json ("-": none)"""
    mode = mode or args.mode
    kw = kw or RRT_KW
    ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    init = np.zeros((n_ep, 6))
    init[:, 0], init[:, 1] = world["start"]
    ctx.rrt_prepare(init, np.arange(n_ep, dtype=np.uint64), args.iters, mode=mode, **kw)
    ms, ems, lms = [], [], []
    for i in range(reps + 1):
        ctx.rrt_run()
        if i:
            ms.append(ctx.last_kernel_ms())
            ems.append(ctx.last_launch_parts()[0])
            lms.append(ctx.last_launch_parts()[1])
    summ = ctx.summaries()
    if (summ["status"] < 0).any():
        return {"error": "episode status %s" % np.unique(summ["status"])}
    k_ms = float(np.mean(ms))
    iters = float(summ["iters_run"].sum())
    out = {"value": iters / (k_ms * 1e-3), "unit": "expansions/s", "episodes": n_ep, "kernel_ms": k_ms, "mode": mode,
           "kernel": ctx.last_rrt_kernel(), "iters_per_launch": iters,
           "accept_rate": float((summ["n_nodes"] - 1).sum()) / iters,
           "cull_candidates_per_expansion": float(summ["n_candidates"].sum()) / iters,
           "qualifying_leaves_per_episode": float(summ["n_leaves"].mean()),
           "roofline": rrt_pass_rooflines(ctx, summ, meas, float(np.mean(ems)), float(np.mean(lms)), ctx.last_rrt_kernel(), compact=True)}
    if cpu_seconds > 0:
        from oracle import orc
        w = orc.WorldArrays(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
        init0 = [world["start"][0], world["start"][1], 0, 0, 0, 0]
        t0, done, eps = time.perf_counter(), 0, 0
        while time.perf_counter() - t0 < cpu_seconds and eps < 16:
            done += orc.rrt_explore(w, eps, args.iters, mode=mode, init=init0, kind="libm", want_path=False, **kw)["iters_run"]
            eps += 1
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": done / dt, "unit": "expansions/s", "cores": 1, "kind": "port",
                               "sample": "%d episodes x %d iterations (seeds 0..%d), oracle/ libm build, %.1f s" % (eps, args.iters, eps - 1, dt)}
    return out


def bench_rrt_o64(ctx, args, n_ep=None):
    """BASELINE configs[1] as written: 64 obstacles (the headline uses 256), same 200x200-cell grid and 10k budget."""
    out = _rrt_batch(ctx, bench_world(64, args.grid), n_ep or args.episodes_fit, args)
    out["metric"] = "RRT.exploring expansions/s, 64 obstacles, %dx%d cells" % (args.grid, args.grid)
    ref = recorded_reference("config2_rrt_exploring_o64")
    if ref:
        out["reference_recorded"] = {"value": ref["ref_expansions_per_s_1proc"], "unit": "expansions/s", "cores": 1,
                                     "where": "build container, tests/experiments/ref_timing.py"}
    return out


def bench_rrt_replicas(ctx, args, n_ep=1024):
    """SURVEY 8(d) config 2, throughput test: 1 024 replicas of the 64-obstacle episode (seeds 0..1023): as many latency chains
    as the chip has SIMDs -- rrt_duo_kernel, two wavefronts per episode (one per SIMD and a helper beside it)."""
    out = _rrt_batch(ctx, bench_world(64, args.grid), n_ep, args)
    out["metric"] = "RRT.exploring expansions/s, %d replicas, 64 obstacles, %dx%d cells" % (n_ep, args.grid, args.grid)
    if "roofline" in out:
        # one latency chain per SIMD: the chain's length per iteration is the figure; counters of the committed pass beside it
        r = out["roofline"]
        r["clocks_per_iteration"] = r["kernel_ms"] * 1e-3 * SHADER_GHZ * 1e9 / (out["iters_per_launch"] / n_ep)
        r.update(pmc_latency("rrt_1024_replicas", out["kernel"].split("<")[0], out["iters_per_launch"]))
        vi = pmc_valu_issue("rrt_1024_replicas", r.get("pmc_kernel")) if r.get("pmc_kernel") else None
        if vi is not None:
            r["valu_issue_frac"] = vi
        # three wavefronts per episode, one episode per SIMD: every SIMD runs one chain; whichever of chain latency and vector
        # issue is the larger share names the bound
        r["bound"] = "valu_issue" if (vi is not None and vi >= 0.5) else "latency"
    return out


def bench_rrt_nn(ctx, args, with_cpu, n_ep=None, long_horizon=False):
    """The nearest-neighbour parent selection of RRT.exploring (plan_time=False: get_random_mps + get_closest_mps,
    rrt_dubins.py:333-343,505-513) at the full 10 000-iteration budget on the headline world: every iteration reads x, y of
    every node of the episode's tree (16 B each) -- the part of the path that streams memory.
      rrt_nn               the headline's parameters (max_traj_time = 500 s): the parent's time stamp rule (:138-139) rejects
                           ~94 % of the samples, the trees stop at ~550 nodes, so the x,y mirrors of all episodes (36 MB)
                           are served by L2 / Infinity Cache
      rrt_nn_long_horizon  max_traj_time = 20 000 s: the 10k-node budget is what ends the tree (~9 600 nodes, ~77 KB per
                           scan on average); run on the headline's batch (12 288 episodes x 158 KB of x,y mirror = 1.9 GB,
                           7.5 x the 256 MB Infinity Cache, so the cache cannot serve the scans): HBM"""
    world = bench_world(args.obstacles, args.grid)
    kw = dict(RRT_KW, max_traj_time=20000.0) if long_horizon else RRT_KW
    if n_ep is None:
        n_ep = args.episodes_fit if long_horizon else min(4096, args.episodes_fit)
    out = _rrt_batch(ctx, world, n_ep, args, reps=2, cpu_seconds=6.0 if with_cpu else 0.0, mode="nn", kw=kw,
                     meas="rrt_nn_long_horizon" if long_horizon else "rrt_nn")
    if "error" in out:
        return out
    out["metric"] = "RRT.exploring expansions/s, nearest-neighbour sampling, %d obstacles, %dx%d cells, max_traj_time %g s" % (
        args.obstacles, args.grid, args.grid, kw["max_traj_time"])
    summ = ctx.summaries()
    scanned = float(summ["nn_scanned"].sum())
    out["nodes_per_tree"] = float(summ["n_nodes"].mean())
    out["nodes_scanned_per_iteration"] = scanned / float(summ["iters_run"].sum())
    out["scan_bytes_per_launch"] = 16.0 * scanned
    out["xy_mirror_working_set_bytes"] = 16.0 * float(summ["n_nodes"].sum())
    out["scan_GBps"] = 16.0 * scanned / (ctx.last_launch_parts()[0] * 1e-3) / 1e9
    out["iters_per_launch"] = float(summ["iters_run"].sum())
    if long_horizon and "roofline" in out:
        # An HBM-ONLY figure.  FETCH_SIZE counts Infinity-Cache hits as memory reads (MI355X_MICROARCH.md, HBM), so no counter
        # separates the two; what separates them is the working set: while the trees are small the co-resident episodes' x,y
        # mirrors fit the 256 MiB cache (the whole-launch rate is 1.05-1.07 x the streaming read rate
        # measured in the same run).  The trees of a seed are the same whatever the budget, so the same batch run to HALF the
        # budget is the first half of the full launch, and the DIFFERENCE of the two launches is the second half alone: from
        # there on every episode scans >= half a full tree per iteration, and between two scans of one mirror the other
        # resident episodes (MODE 2 keeps 5 wavefronts per SIMD = 20 episodes per CU) stream >= `late_resident_bytes` -- well
        # past the cache -- through it.
        import argparse
        half = argparse.Namespace(**vars(args))
        half.iters = args.iters // 2
        o2 = _rrt_batch(ctx, world, n_ep, half, reps=2, mode="nn", kw=kw, meas="-")
        if "roofline" in o2:
            r1, r2 = out["roofline"], o2["roofline"]
            d_bytes = r1["algorithmic_bytes_per_launch"] - r2["algorithmic_bytes_per_launch"]
            d_ms = r1["kernel_ms"] - r2["kernel_ms"]
            s2 = ctx.summaries()
            n_cu = 256
            resident = min(n_ep, 20 * n_cu)
            late = {"late_segment": "iterations %d..%d of the same batch (full launch minus a launch of the first %d)" % (half.iters, args.iters, half.iters),
                    "late_kernel_ms": d_ms, "late_alg_bytes": d_bytes, "late_achieved_GBps": d_bytes / (d_ms * 1e-3) / 1e9,
                    "late_frac": d_bytes / (d_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "late_nodes_per_tree_at_start": float(s2["n_nodes"].mean()),
                    "late_resident_bytes": 16.0 * float(s2["n_nodes"].mean()) * resident,
                    "first_half_kernel_ms": r2["kernel_ms"], "first_half_achieved_GBps": r2["achieved"]}
            if HBM_MEASURED.get("read_GBps"):
                late["late_frac_of_measured"] = late["late_achieved_GBps"] / HBM_MEASURED["read_GBps"]
            out["hbm_only"] = late
            r1.update({"hbm_only_GBps": late["late_achieved_GBps"], "hbm_only_frac": late["late_frac"],
                       "hbm_only_frac_of_measured": late.get("late_frac_of_measured"),
                       "whole_launch_label": "HBM + Infinity Cache (the first iterations' mirrors fit the 256 MiB cache)"})
    ref = recorded_reference(("rrt_exploring_nn_long_o%d" if long_horizon else "rrt_exploring_nn_o%d") % args.obstacles)
    if ref:
        out["reference_recorded"] = {"value": ref["ref_expansions_per_s_1proc"], "unit": "expansions/s", "cores": 1,
                                     "where": "build container, tests/experiments/ref_timing.py", "sample": ref.get("sample")}
    return out


def bench_rrt_dense(ctx, args, with_cpu, n_ep=None):
    """Worlds where the exact collision test actually runs (the headline's 256 obstacles in 4 km^2 are sparse: the cull
    leaves well under one candidate per expansion).  (i) the G3 fixture world: 256 obstacles of r = 1-3 m in a 200 m box,
    400 cells -- the reference accepts ~56 % there and spends 93 % of its time in check_collision; (ii) a Catalina-sized
    workspace (path_planning/catalina.py:67-119: ~550 x 345 m): 560 x 350 m, 14 m cells (1 000 cells, the reference's
    split gives 987), 256 obstacles with the Catalina radii spread (4-26 m obstacles scaled down to stay plannable: 2-8 m)."""
    from auv_sim_amd import synth
    out = {}
    n_ep = n_ep or args.episodes_fit
    w1 = synth.make_world(seed=2, n_obstacles=256)
    out["g3_box_200m_o256"] = _rrt_batch(ctx, w1, n_ep, args, cpu_seconds=4.0 if with_cpu else 0.0)
    out["g3_box_200m_o256"]["world"] = "200 m x 200 m box, 400 cells, 256 obstacles r = 1-3 m (the G3 golden world)"
    w2 = synth.make_world(seed=5, n_obstacles=256, box=(0.0, 0.0, 560.0, 350.0), cell=14.0, obst_radius=(2.0, 8.0),
                          hab_radius=(20.0, 55.0))
    out["catalina_560x350_o256"] = _rrt_batch(ctx, w2, n_ep, args, cpu_seconds=4.0 if with_cpu else 0.0)
    out["catalina_560x350_o256"]["world"] = "560 m x 350 m, 1 000 cells of 14 m, 256 obstacles r = 2-8 m, habitats r = 20-55 m"
    return out


def astar_inputs(n_inst):
    from auv_sim_amd import synth
    w = synth.make_world(seed=12, n_obstacles=64, obst_radius=(2.0, 6.0), n_habitats=10, hab_radius=(10.0, 25.0))
    rng = np.random.default_rng(3)
    starts = np.column_stack([-290.0 + 10.0 * rng.integers(0, 19, n_inst), -90.0 + 10.0 * rng.integers(0, 19, n_inst)])
    limits = rng.choice([100.0, 200.0, 300.0], n_inst)
    return w, starts, limits


def bench_astar(ctx, ranks, with_cpu, n_inst=1024, steps=5, warmup=1, variants=True):
    """BASELINE config 3: 1024 independent astar_fixLenSOG searches (starts on the 10 m lattice, pathLenLimit in
    {100,200,300}) over one shared world: 64 obstacles, 10 habitats, rectangle polygon, 20x20-cell shark grid x 10 bins.
    A step = the search launch (including whatever reset the batch needs) + the path/smoothing launch + the result
    download (+ the gather for N > 1); cells/s = child cells evaluated / step time.  For N > 1 the
    instances are block-sharded over the ranks."""
    from auv_sim_amd import _astar_lib, distributed as D
    w, starts, limits = astar_inputs(n_inst)
    lo, hi = D.shard_range(n_inst, ranks.rank, ranks.world)
    ctx.set_world(w["obstacles"], w["habitats"], w["polygon"], w["bins"], w["cells"], w["prob"])
    wts = (0, 10, 10, 100)
    kms, gms = [], []

    def step():
        r = _astar_lib.run_batch_arrays(ctx, "astar_fixLenSOG", starts[lo:hi], limits=limits[lo:hi], weights=wts, velocity=1.0,
                                        cap_nodes=20000)
        kms.append(r["batch_ms"])
        if ranks.world > 1:
            ranks.gather_host_records(r["summ"])
            gms.append(ranks.gather_ms())
        return r
    dt, r = timed_steps(ranks, step, steps, warmup)
    summ = r["summ"]
    if (summ["status"] < 0).any():
        return {"error": "instance status %s" % np.unique(summ["status"][summ["status"] < 0])}
    cells = ranks.sum(summ["n_children"].sum())
    k_ms = float(np.mean(kms[-steps:]))
    abytes = astar_bytes(summ, "astar_fixLenSOG")
    traffic = pmc_traffic("astar", ["astar_kernel"], float(summ["n_children"].sum()))
    out = {"metric": "A* cells/s (astar_fixLenSOG, child cells evaluated)", "value": cells * steps / dt, "unit": "cells/s",
           "ms_per_step": 1e3 * dt / steps, "steps": steps, "instances": n_inst, "instances_this_rank": hi - lo,
           "cells_per_step": cells, "expansions_per_step": ranks.sum(summ["n_expansions"].sum()),
           "found": int(ranks.sum(summ["found"].sum())),
           "search_launch_ms": k_ms, "search_launch_ms_per_rank": ranks.all(k_ms),
           "gather_ms_per_rank": ranks.all(float(np.mean([g for g in gms[-steps:] if g is not None])) if gms and gms[-1] is not None else None),
           "cells_per_s_search_launch_only": float(summ["n_children"].sum()) / (k_ms * 1e-3),
           "config": "%d x astar_fixLenSOG, 64 obstacles, 10 habitats, 400-cell grid x 10 bins, limits 100/200/300" % n_inst,
           "roofline": roofline(abytes, k_ms, "astar_kernel", traffic, valu_issue_frac=pmc_valu_issue("astar") if traffic["traffic"] is not None else None,
                                bytes_per_cell=abytes / max(float(summ["n_children"].sum()), 1.0),
                                note="one wave per instance at 1 wave/SIMD: a latency measurement, not a bandwidth one")}
    if ranks.world == 1 and variants:
        # SURVEY 8(d) config 3 also asks for the same batch through astar_fixLen (no grid) and astar.astar (start -> goal
        # pairs on the 50x50 lattice of config 1): reported side by side, labelled
        def variant(name, st, **k):
            _astar_lib.run_batch_arrays(ctx, name, st, **k)
            t0 = time.perf_counter()
            rr = None
            for _ in range(3):
                rr = _astar_lib.run_batch_arrays(ctx, name, st, **k)
            vdt = (time.perf_counter() - t0) / 3
            s = rr["summ"]
            if (s["status"] < 0).any():
                return {"error": "instance status %s" % np.unique(s["status"][s["status"] < 0])}
            return {"value": float(s["n_children"].sum()) / vdt, "unit": "cells/s", "cells_per_step": int(s["n_children"].sum()),
                    "found": int(s["found"].sum()), "ms_per_step": 1e3 * vdt, "search_launch_ms": rr["batch_ms"],
                    "roofline": roofline(astar_bytes(s, name), rr["batch_ms"], "astar_kernel")}
        from auv_sim_amd import synth
        out["variants"] = {"astar_fixLen": variant("astar_fixLen", starts, limits=limits, weights=(0, 10, 10), cap_nodes=20000)}
        lw = synth.make_lattice_world(seed=11, n_obstacles=30, r_range=(10, 22))
        ctx.set_world(lw["obstacles"], None, None, None, None, None)
        rng = np.random.default_rng(3)
        lst = np.column_stack([10.0 * rng.integers(0, 20, n_inst), 10.0 * rng.integers(0, 20, n_inst)])
        out["variants"]["astar"] = variant("astar", lst, goals=np.tile([490.0, 490.0], (n_inst, 1)), box=lw["box"], cap_nodes=60000)
        # config 3 gives every SIMD ONE wavefront (1 024 instances on 1 024 SIMDs): a latency measurement.  The same search
        # with the instance list repeated until every CU holds its three workgroups (12 waves) shows what the kernel does
        # when the chip is full -- labelled, not the config-3 number.
        n_sat = 12 * n_inst
        ctx.set_world(w["obstacles"], w["habitats"], w["polygon"], w["bins"], w["cells"], w["prob"])
        out["variants"]["astar_fixLenSOG_x12_instances"] = variant(
            "astar_fixLenSOG", np.tile(starts, (12, 1)), limits=np.tile(limits, 12), weights=wts, velocity=1.0, cap_nodes=20000)
        out["variants"]["astar_fixLenSOG_x12_instances"]["instances"] = n_sat
    if with_cpu:
        from oracle import orc_astar as oa
        t0, c, n = time.perf_counter(), 0, 0
        while time.perf_counter() - t0 < 5.0 and n < n_inst:
            rr = oa.run("astar_fixLenSOG", starts[n], obstacles=w["obstacles"], habitats=w["habitats"], polygon=w["polygon"],
                        bins=w["bins"], cells=w["cells"], prob=w["prob"], limit=float(limits[n]), weights=wts, velocity=1.0,
                        cap_nodes=20000, kind="libm")
            c += rr["n_children"]
            n += 1
        cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": c / cdt, "unit": "cells/s", "cores": 1, "kind": "port",
                               "sample": "first %d of the %d instances, oracle/ libm build, %.1f s" % (n, n_inst, cdt)}
        ref = recorded_reference("config3_astar_fixLenSOG")
        if ref:
            out["cpu_baseline"]["reference_recorded"] = {"value": ref["ref_cells_per_s_1proc"], "unit": "cells/s", "cores": 1,
                                                         "many_cores": {k: v for k, v in ref.items() if k.startswith("ref_cells_per_s_") and k != "ref_cells_per_s_1proc"},
                                                         "where": "build container, tests/experiments/ref_timing.py"}
    return out


def bench_rrt_env(local_rank, n_env=512, n_steps=60):
    """the batched RRTEnv (gym_rrt/envs/rrt_env.py:182-295) -- n_env environments of config 4's world stepped
    together with a random occupied-bucket policy; one step = bucket upload + generate_one_node launch + observation kernel +
    the observation dict on the HOST (rrt_grid [E, buckets, 4] f64, has_node, node counts: ~39 MB per step at 512 x 1 600
    buckets, i.e. a PCIe figure), and the same with the observation left on the device (observation_to_device)."""
    import torch
    from auv_sim_amd import synth
    from auv_sim_amd.motion_plan_state import Motion_plan_state as MPS
    from auv_sim_amd.rrt_env import RRTEnvBatch, R_CREATE_NODE
    w = synth.make_rect_world(seed=3, n_obstacles=256)
    obstacles = [MPS(o[0], o[1], size=o[2]) for o in w["obstacles"].tolist()]
    bnd = [MPS(float(w["rect"][0]), float(w["rect"][1])), MPS(float(w["rect"][2]), float(w["rect"][3]))]
    auv, shark = MPS(float(w["start"][0]), float(w["start"][1]), z=-5.0), MPS(float(w["goal"][0]), float(w["goal"][1]), z=-5.0)
    env = RRTEnvBatch(auv, shark, bnd, 5, 1, obstacles, seeds=list(range(n_env)), max_nodes=n_steps + 8, freq=10, device=local_rank)
    rng = np.random.default_rng(5)

    def policy(st):
        # a random bucket among those that hold a node (what an agent that respects the action mask does)
        # (vectorised: a Python loop over the environments took most of the step it was meant to drive)
        has = st["has_node"] != 0
        cnt = has.sum(axis=1)
        k = (rng.random(len(has)) * np.maximum(cnt, 1)).astype(np.int64)
        pick = (np.cumsum(has, axis=1, dtype=np.int32) > k[:, None]).argmax(axis=1)
        return np.where(cnt > 0, pick, 0).astype(np.int64)
    st = env.reset()
    st, _, _, _ = env.step(policy(st))  # warm-up (allocations)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    created = 0
    for _ in range(n_steps):
        st, reward, done, _ = env.step(policy(st))
        created += int((reward == R_CREATE_NODE).sum())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nb = env.n_buckets
    out = {"metric": "RRTEnv steps/s (batched env, host observation dict)", "value": n_env * n_steps / dt, "unit": "env-steps/s",
           "envs": n_env, "steps": n_steps, "ms_per_batched_step": 1e3 * dt / n_steps, "buckets": nb,
           "observation_bytes_per_step": int(n_env * nb * (32 + 8 + 8)), "nodes_created": created,
           "note": "includes the numpy policy on the host and the download of the full observation arrays"}
    # the same environments with NOTHING crossing PCIe (what row f1 is for): ONE launch per step -- generate_one_node for every
    # live environment with the stand-in agent's pick made inside the launch, the outcome (reward, done flag) written by it
    # and the observation arrays updated in place (a step changes one bucket per environment) -- enqueued back to back on the
    # planner's stream; one wait at the end.  Beside it: the same loop with the observation arrays rewritten whole every step
    # (two launches; what the reference env rebuilds after every node), and the one-launch step as a hipGraph of 16 steps.
    def device_loop(observe, n2, graph_steps=0):
        env2 = RRTEnvBatch(auv, shark, bnd, 5, 1, obstacles, seeds=list(range(n_env)), max_nodes=4 * n_steps + 64, freq=10, device=local_rank)
        env2.reset()
        d = env2.device_buffers()

        def one_step():
            env2.step_device(agent_seed=5, observe=observe)
        for _ in range(3):
            one_step()
        env2.sync()
        if graph_steps:
            gid = env2.capture_step(lambda: [one_step() for _ in range(graph_steps)])
            env2.replay(gid, 1)
            env2.sync()
            enqueue = lambda: env2.replay(gid, n2 // graph_steps)
        else:
            enqueue = lambda: [one_step() for _ in range(n2)]
        t0 = time.perf_counter()
        dev_ms = env2.timed(enqueue)
        wall = time.perf_counter() - t0
        env2.sync()
        return {"value": n_env * n2 / wall, "unit": "env-steps/s", "steps": n2, "ms_per_batched_step": 1e3 * wall / n2,
                "device_ms_per_batched_step": dev_ms / n2, "envs_still_running_at_end": int((d["done"] == 0).sum().item()),
                "nodes_in_all_trees": int(d["num_nodes"].sum().item())}
    n2 = 3 * n_steps - (3 * n_steps) % 16
    one = device_loop("delta", n2)
    full = device_loop(True, n2)
    graph = device_loop("delta", n2, graph_steps=16)
    # algorithmic bytes of one batched step, one-launch form: the planner step (~0.33 KB per environment) + the
    # changed observation entries (8 + 8 + 8 B) + reward / flags / bucket (8 + 1 + 1 + 4 B); full rewrite: + 52 B per bucket
    ab_one = n_env * (330.0 + 24 + 14)
    ab_full = float(n_env) * nb * (32 + 8 + 8 + 4) + n_env * 330.0
    out["device_resident"] = dict(one, **{
        "metric": "RRTEnv steps/s, device-resident loop, ONE launch per step (agent + generate_one_node + outcome + in-place observation update)",
        "envs": n_env, "launches_per_step": 1,
        "full_observation_rewrite": dict(full, launches_per_step=2,
                                         roofline=roofline(ab_full, full["device_ms_per_batched_step"], "prrt_kernel (step mode) + prrt_observation_kernel",
                                                           note="the observation rewrite (49 MB per step) is the HBM-sized term")),
        "hipgraph_replay": dict(graph, steps_per_graph=16, note="the one-launch step captured 16 x on the planner's stream and replayed"),
        "roofline": roofline(ab_one, one["device_ms_per_batched_step"], "prrt_kernel<4,true> (step mode: agent + generate_one_node + outcome + observation delta)",
                             note="kernel_ms = HIP-event time of the enqueued loop / steps (device time, not host wall time); 512 waves, "
                                  "one dependent fp64 chain each: a latency figure")})
    return out


def bench_planner(ctx, ranks, with_cpu, n_ep=512, max_step=2000, steps=5, warmup=1):
    """BASELINE config 4: 512 Planner_RRT.planning(max_step=2000) episodes, 200 m x 200 m rectangle, 256 obstacles, cell
    5 m, 1 theta subsection, freq 10, start (20,20) -> goal (170,180), seed = global episode id; block-sharded over the
    ranks (64 per GPU at N = 8) with the gather of the summary records and final paths.  A step = batch creation
    (seeding, tree planting) + planning launch + path extraction (+ gather)."""
    from auv_sim_amd import synth, distributed as D
    from auv_sim_amd._prrt_lib import PlannerBatch, PRRT_SUMMARY_DTYPE
    w = synth.make_rect_world(seed=3, n_obstacles=256)
    ctx.set_world(obstacles=w["obstacles"])
    lo, hi = D.shard_range(n_ep, ranks.rank, ranks.world)
    n = hi - lo
    starts = np.tile(np.array([w["start"][0], w["start"][1], 0.0, 0.0]), (n, 1))
    goals = np.tile(w["goal"], (n, 1))
    seeds = np.arange(lo, hi, dtype=np.uint64)
    kms, gms = [], []

    def step():
        pb = PlannerBatch(ctx, starts, goals, w["rect"], max_step, seeds=seeds, freq=10, cell=5, subs=1)
        summ = pb.plan()
        kms.append(ctx.last_kernel_ms())
        paths = pb.paths(summ)
        if ranks.world > 1:
            import torch
            ranks.gather_records(pb.L.auvp_prrt_summaries_dev(ctx.h), n, PRRT_SUMMARY_DTYPE.itemsize)
            lens = torch.from_numpy(np.where(summ["done"] != 0, summ["path_len"], 0).astype(np.int64)).to(ranks.dev)
            flat = torch.from_numpy(np.concatenate(paths) if len(paths) else np.zeros((0, 5))).to(ranks.dev)
            ranks.gather.gather_paths(flat, lens)
            gms.append(ranks.gather_ms())
        return summ
    dt, summ = timed_steps(ranks, step, steps, warmup)
    if (summ["status"] < 0).any():
        return {"error": "episode status %s" % np.unique(summ["status"])}
    tot_steps = ranks.sum(summ["steps"].sum())
    k_ms = float(np.mean(kms[-steps:]))
    abytes = planner_bytes(summ)
    traffic = pmc_traffic("planner_rrt", [ctx.prrt_last_kernel()], float(summ["steps"].sum()))
    out = {"metric": "Planner_RRT steps/s (generate_one_node calls)", "value": tot_steps * steps / dt, "unit": "steps/s",
           "ms_per_step": 1e3 * dt / steps, "steps": steps, "episodes": n_ep, "episodes_this_rank": n,
           "planner_steps_per_step": tot_steps, "episodes_done": int(ranks.sum(summ["done"].sum())),
           "plan_launch_ms": k_ms, "plan_launch_ms_per_rank": ranks.all(k_ms),
           "gather_ms_per_rank": ranks.all(float(np.mean([g for g in gms[-steps:] if g is not None])) if gms and gms[-1] is not None else None),
           "steps_per_s_plan_launch_only": float(summ["steps"].sum()) / (k_ms * 1e-3),
           "config": "%d x Planner_RRT.planning(2000), 200 m env, 256 obstacles, cell 5 m, freq 10" % n_ep,
           "roofline": roofline(abytes, k_ms, ctx.prrt_last_kernel(), traffic,
                                valu_issue_frac=pmc_valu_issue("planner_rrt") if traffic["traffic"] is not None else None,
                                bytes_per_step=abytes / max(float(summ["steps"].sum()), 1.0),
                                note="512 waves on 1 024 SIMDs: a latency measurement")}
    if with_cpu:
        from oracle import orc_planner as op
        t0, c, k = time.perf_counter(), 0, 0
        while time.perf_counter() - t0 < 5.0 and k < n:
            r = op.planning(w["obstacles"], w["rect"], starts[k], goals[k], int(seeds[k]), max_step, 10, 5, 1, kind="libm")
            c += r["steps"]
            k += 1
        cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": c / cdt, "unit": "steps/s", "cores": 1, "kind": "port",
                               "sample": "first %d of the %d episodes, oracle/ libm build, %.1f s" % (k, n_ep, cdt)}
        ref = recorded_reference("config4_planner_rrt")
        if ref:
            out["cpu_baseline"]["reference_recorded"] = {"value": ref["ref_steps_per_s_1proc"], "unit": "steps/s", "cores": 1,
                                                         "many_cores": {k2: v for k2, v in ref.items() if k2.startswith("ref_steps_per_s_") and k2 != "ref_steps_per_s_1proc"},
                                                         "where": "build container, tests/experiments/ref_timing.py"}
    return out


def bench_config5(ctx, ranks, n_filters=25, n_particles=500, max_step=200, track_steps=6):
    """BASELINE config 5 as written: particle filters over the reference's recorded shark tracks
    (data/sharkTrackingData.csv -> tests/golden/shark_tracking_xy.npz), one Planner_RRT replan per particle and tracking
    step, device resident (auv_sim_amd.tracking).  Per GPU: 25 filters x 500 particles = 12 500 episodes x 200 planner
    steps per tracking step (100 000 particles over 8 GPUs); filter f of rank r tracks shark (r * 25 + f) mod 32 with
    np.random.seed(r * 25 + f); episode seeds follow the global episode id."""
    from auv_sim_amd import synth, tracking
    path = os.path.join(REPO, "tests", "golden", "shark_tracking_xy.npz")
    if not os.path.exists(path):
        return {"error": "tests/golden/shark_tracking_xy.npz missing"}
    xy = np.load(path)["xy"]
    w = synth.make_rect_world(seed=3, n_obstacles=256)
    ctx.set_world(obstacles=w["obstacles"])
    gf = ranks.rank * n_filters + np.arange(n_filters)
    E = n_filters * n_particles
    rp = tracking.ParticleReplanner(ctx, xy[gf % 32, :track_steps + 1], n_particles, w["rect"], w["start"], gf, max_step=max_step,
                                    episode_offset=ranks.rank * E, episodes_total=ranks.world * E)
    rp.step(0)  # warm-up (also sizes every buffer)
    ranks.sync()
    t0 = time.perf_counter()
    pf_ms, plan_ms, steps_done, done = [], [], 0, 0
    for s in range(1, track_steps + 1):
        summ = rp.step(s)
        pf_ms.append(rp.pf_ms)
        plan_ms.append(rp.plan_ms)
        steps_done += int(summ["steps"].sum())
        done = int(summ["done"].sum())
        if ranks.world > 1:
            from auv_sim_amd._prrt_lib import PRRT_SUMMARY_DTYPE
            ranks.gather_records(rp.planner.L.auvp_prrt_summaries_dev(ctx.h), E, PRRT_SUMMARY_DTYPE.itemsize)
        if (summ["status"] < 0).any():
            return {"error": "episode status %s" % np.unique(summ["status"])}
    ranks.sync()
    dt = ranks.max_time(time.perf_counter() - t0)
    st, _ = rp.filters.status()
    if (st != 0).any():
        return {"error": "filter status %s" % np.unique(st)}
    total = ranks.sum(steps_done)
    abytes = planner_bytes(summ)  # of the last tracking step's plan launch
    steps_last = float(summ["steps"].sum())
    kname = ctx.prrt_last_kernel()
    c5_traffic = pmc_traffic("config5", [kname], steps_last)
    return {"metric": "config 5: Planner_RRT steps/s, one replan per particle hypothesis per tracking step",
            "value": total / dt, "unit": "steps/s", "ms_per_tracking_step": 1e3 * dt / track_steps,
            "planner_steps_per_tracking_step": steps_last, "steps_per_s_plan_launch_only": steps_last / (plan_ms[-1] * 1e-3),
            "roofline": roofline(abytes, plan_ms[-1], kname, c5_traffic,
                                 valu_issue_frac=pmc_valu_issue("config5") if c5_traffic["traffic"] is not None else None,
                                 bytes_per_step=abytes / max(steps_last, 1.0),
                                 note="the last tracking step's plan launch; ~3 waves per SIMD, each step a chain of dependent "
                                      "fp64 sequences (atan2 / sincos / divisions) and tree reads: issue and latency bound, not HBM"),
            "episodes_per_gpu": E, "filters_per_gpu": n_filters, "particles_per_filter": n_particles, "max_step": max_step,
            "tracking_steps": track_steps, "episodes_done_last_step": done,
            "episode_replans_per_s": ranks.world * E * track_steps / dt,
            "filter_ms": float(np.mean(pf_ms)), "plan_launch_ms": float(np.mean(plan_ms)),
            "data": "recorded shark tracks of the reference (32 sharks x 815 samples), noise-free range/bearing from two fixed AUVs"}


def bench_particle_filter(device, with_cpu, n_filters=4096, n_particles=1000, n_steps=20, n_auv=2, reps=3):
    """F shark particle filters x 1000 particles x S steps of create_and_update + update_weights
    + particleMean/meanError (robotSim.py:665-701) in one launch; filter f continues np.random.seed(f)."""
    from auv_sim_amd import _lib, _pf_lib
    ctx = _lib.Context(device)
    rng = np.random.default_rng(4)
    F, N, S, A = n_filters, n_particles, n_steps, n_auv
    shark0 = rng.uniform(-500, 500, size=(F, 2))
    meas = np.zeros((S, F, A, 5))
    meas[..., 0:2] = shark0[None, :, None, :] + rng.uniform(-150, 150, size=(S, F, A, 2))
    meas[..., 2] = rng.uniform(-np.pi, np.pi, size=(S, F, A))
    meas[..., 3] = rng.uniform(0, 200, size=(S, F, A))
    meas[..., 4] = rng.uniform(-np.pi, np.pi, size=(S, F, A))
    shark = shark0[None] + rng.uniform(-20, 20, size=(S, F, 2))
    key0, _ = _pf_lib.np_seed_state(0)
    mts = np.stack([_pf_lib.np_seed_state(f)[0] if f < 64 else np.roll(key0, f) ^ np.uint32(f) for f in range(F)])
    ms = []
    for i in range(reps + 1):
        b = _pf_lib.FilterBatch(ctx, F, N).create(shark0, mts, 624)
        b.run(meas=meas, shark_xy=shark)
        if i:
            ms.append(ctx.last_kernel_ms())
    st, nd = b.status()
    if (st != 0).any():
        return {"error": "filter status %s" % np.unique(st)}
    k_ms = float(np.mean(ms))
    units = float(F) * N * S
    abytes = F * N * 2 * 44.0 + S * F * (A * 40.0 + 28.0)  # particles in and out once per launch; measurements in, estimates out per step
    pf_traffic = pmc_traffic("particle_filter", ["pf_step_kernel"], None)
    out = {"metric": "particle filter particle-steps/s (create_and_update + update_weights)", "value": units / (k_ms * 1e-3),
           "unit": "particle-steps/s", "filters": F, "particles": N, "steps": S, "auvs": A, "kernel_ms": k_ms,
           "draws32_per_filter_step": float(nd.mean()) / S,
           "config": "%d filters x %d particles x %d steps, %d AUV measurements per step" % (F, N, S, A),
           "roofline": roofline(abytes, k_ms, "pf_step_kernel", pf_traffic,
                                valu_issue_frac=pmc_valu_issue("particle_filter", "pf_step_kernel") if pf_traffic["traffic"] is not None else None,
                                note="state is LDS resident across the steps of a launch; four wavefronts per SIMD of dependent fp64 "
                                     "chains (atan2, exponentials, MT19937 blocks, ordered sums) between ~45 workgroup barriers per "
                                     "step: issue and latency bound, not HBM")}
    if with_cpu:
        from oracle import orc_pf
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < 4.0 and n < F:
            orc_pf.run(N, meas[:, n], shark[:, n], shark0[n], mts[n], 624, kind="libm")
            n += 1
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n * N * S / dt, "unit": "particle-steps/s", "cores": 1, "kind": "port",
                               "sample": "first %d of the %d filters, oracle/ libm build, %.1f s" % (n, F, dt)}
    return out


def bench_shark_grid(device, with_cpu, n_side=200, n_sharks=32, n_pts=3000, reps=3):
    """SharkOccupancyGrid.convert, 10 m cells over 2 km x 2 km, 32 sharks x 3000 points, 10 bins of
    30 s, detection range 50 m (the reference's constructor arguments at rrt_dubins.py:68)."""
    from auv_sim_amd import _lib
    from auv_sim_amd.sharkOccupancyGrid import convert_arrays
    ctx = _lib.Context(device)
    rng = np.random.default_rng(7)
    cs, n = 10.0, n_side
    box = (0.0, 0.0, cs * n, cs * n)
    cx, cy = np.meshgrid(np.arange(n), np.arange(n))
    cells = np.stack([cx.ravel() * cs, cy.ravel() * cs, (cx.ravel() + 1) * cs, (cy.ravel() + 1) * cs], axis=1)
    traj_len = np.full(n_sharks, n_pts, dtype=np.int32)
    t = np.tile(np.arange(1, n_pts + 1) * 0.1, n_sharks)
    pts = np.stack([rng.uniform(1, cs * n - 2, len(t)), rng.uniform(1, cs * n - 2, len(t)), t], axis=1)
    ms = []
    for i in range(reps + 1):
        bins, grids = convert_arrays(ctx, cells, box, cs, 30.0, 50.0, traj_len, pts)
        if i:
            ms.append(ctx.last_kernel_ms())
    k_ms = float(np.mean(ms))
    T, G = grids.shape[0], grids.shape[1] * grids.shape[2]
    units = float(T) * G
    # compulsory traffic: the points once (24 B), the per-(bin, shark, cell) occupancy written and read once (8 + 8 B; the
    # 81-cell window re-reads of the disc stencil come from LDS tiles and are NOT counted), the output once (8 B)
    abytes = len(pts) * 24.0 + T * n_sharks * G * (4.0 + 8.0 + 8.0) + units * 8.0
    out = {"metric": "SharkOccupancyGrid.convert output cells/s", "value": units / (k_ms * 1e-3), "unit": "grid cells/s",
           "bins": int(T), "grid": [int(grids.shape[1]), int(grids.shape[2])], "sharks": n_sharks, "kernel_ms": k_ms,
           "config": "%dx%d cells of 10 m, %d sharks x %d points, %d bins, detect range 50 m" % (n, n, n_sharks, n_pts, T),
           "roofline": roofline(abytes, k_ms, "sog_count/occ/grid_kernel",
                                pmc_traffic("shark_grid", ["sog_count_kernel", "sog_occ_kernel", "sog_grid_tile_c_kernel"], None),
                                note="compulsory bytes only (three launches: count, occupancy, window sums); the 81-cell disc "
                                     "windows are summed from LDS tiles, one LDS read per four of the %.1f G ordered fp64 additions"
                                     % (units * n_sharks * 81 / 1e9))}
    if with_cpu:
        from oracle import orc_sog
        sub = 2
        k = sub * n_pts
        t0 = time.perf_counter()
        r = orc_sog.convert(cells, box, cs, 30.0, 50.0, traj_len[:sub], pts[:k], kind="libm")
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": float(len(r["grids"])) * G * (sub / float(n_sharks)) / dt, "unit": "grid cells/s",
                               "cores": 1, "kind": "port",
                               "sample": "%d of the %d sharks (cost is linear in sharks; value scaled by %d/%d), %.1f s"
                                         % (sub, n_sharks, sub, n_sharks, dt)}
    return out


# ----------------------------------------------------------------------------------------------------------------
def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (fresh children, one per GPU, before
    anything in this process touches HIP), relay rank 0's JSON line, exit with the worst child's code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # a rank that dies (e.g. its communicator initialisation failed) must not leave the others waiting inside a
    # collective: the first non-zero exit ends the job (these are our own children: exact PIDs, no patterns)
    rc, live = 0, list(procs)
    while live:
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            rc = max(rc, abs(r))
            if r != 0:
                for q in live:
                    q.terminate()
        time.sleep(0.05)
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-extra", action="store_true", help="skip the side measurements")
    ap.add_argument("--only", default="", help="comma list of side measurements to run INSTEAD of the headline "
                                                "(astar, planner_rrt, config5, rrt_dense, ...): profiling passes")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--episodes", type=int, default=12288,
                    help="episodes per GPU per step (12288 = one 48-episode workgroup of the four-episodes-per-wavefront kernel "
                         "on each of the 256 CUs; 8.8 MB of tree storage each = 108 GB)")
    ap.add_argument("--iters", type=int, default=10000, help="expansion budget per episode (10k-node budget)")
    ap.add_argument("--obstacles", type=int, default=256)
    ap.add_argument("--grid", type=int, default=200, help="grid is grid x grid cells of 10 m")
    ap.add_argument("--mode", default="timebin", choices=["timebin", "nn", "plantime"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="config 3 only, without the other A* variants (counter passes)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world_size
    only = [s for s in args.only.split(",") if s]
    args.episodes_fit = args.episodes

    world = bench_world(args.obstacles, args.grid)
    with_cpu = world_size == 1 and not args.no_cpu
    cpu_all = None
    if with_cpu and not only:
        cpu_all = cpu_baseline_all_cores(world, args.iters, args)  # forks: must precede any HIP initialisation

    import torch
    import torch.distributed as dist
    from auv_sim_amd import _lib

    # AUVP_BENCH_ONE_GPU=1 (plumbing test on a single-GPU box): every rank uses GPU 0 and the ranks talk over gloo -- RCCL
    # cannot put two ranks on one device -- so everything but the RCCL transport itself is exercised
    one_gpu = os.environ.get("AUVP_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world_size)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=dev)

    ctx = _lib.Context(local_rank)
    ranks = Ranks(ctx, rank, world_size, dev, use_rccl=not one_gpu)
    if rank == 0:
        measure_hbm(ctx)  # 4 GiB streaming read + copy, a few ms: the measured roof every roofline object quotes
    sides = {
        "single_episode": lambda: bench_single_episode(ctx, world, args),
        "rrt_64_obstacles": lambda: bench_rrt_o64(ctx, args),
        "rrt_1024_replicas": lambda: bench_rrt_replicas(ctx, args),
        "rrt_dense": lambda: bench_rrt_dense(ctx, args, with_cpu),
        "rrt_nn": lambda: bench_rrt_nn(ctx, args, with_cpu),
        "rrt_nn_long_horizon": lambda: bench_rrt_nn(ctx, args, with_cpu, long_horizon=True),
        "astar": lambda: bench_astar(ctx, ranks, with_cpu, variants=not args.no_variants),
        "rrt_env": lambda: bench_rrt_env(local_rank),
        "planner_rrt": lambda: bench_planner(ctx, ranks, with_cpu),
        "config5": lambda: bench_config5(ctx, ranks),
        "shark_grid": lambda: bench_shark_grid(local_rank, with_cpu),
        "particle_filter": lambda: bench_particle_filter(local_rank, with_cpu),
    }
    sharded = ("astar", "planner_rrt", "config5")  # these run on every rank; the others on rank 0's GPU only
    if only:
        out = {}
        for name in only:
            if name in sharded or rank == 0:
                r = sides[name]()
                if rank == 0:
                    out[name] = r
        if rank == 0:
            print(json.dumps(out))
        if world_size > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    E = args.episodes
    # the trees need ~8.8 MB of HBM per 10 000-iteration episode (nodes 1.1 MB, path points 7.5 MB, time-bin lists 0.2 MB): if
    # this GPU has less free than the requested batch needs, run the largest multiple of 6144 episodes that fits -- every
    # rank the same
    free_b, _total_b = torch.cuda.mem_get_info(dev)
    fit = int(0.92 * free_b / (9.2e6 * max(args.iters, 1) / 10000.0))
    if fit < E:
        fit = max(6144 * (fit // 6144), min(E, 1024))
    e_fit = ranks._t([min(E, fit)], torch.int64)
    if world_size > 1:
        dist.all_reduce(e_fit, op=dist.ReduceOp.MIN)
    if int(e_fit.item()) < E:
        print("bench: %d episodes per GPU do not fit the free HBM (%.0f GB); running %d" % (E, free_b / 1e9, int(e_fit.item())),
              file=sys.stderr)
        E = int(e_fit.item())
    args.episodes_fit = E  # the side measurements that fill the GPU use the same batch size
    init = np.zeros((E, 6))
    init[:, 0], init[:, 1] = world["start"]
    seeds = np.arange(rank * E, (rank + 1) * E, dtype=np.uint64)  # global episode id = seed
    ctx.rrt_prepare(init, seeds, args.iters, mode=args.mode, **RRT_KW)
    kms, gms, parts = [], [], []

    def step():
        """expansion + leaf pass + best-path extraction (+ RCCL gather of the result records for N > 1)"""
        ctx.rrt_run()
        kms.append(ctx.last_kernel_ms())
        parts.append(ctx.last_launch_parts())
        summ = ctx.summaries()
        lens = np.where(summ["best_leaf"] >= 0, summ["best_path_len"], 0).astype(np.int64)
        off = np.zeros(E + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        total = int(off[-1])
        paths = torch.empty((max(total, 1), 7), dtype=torch.float64, device=dev)
        ctx.paths_dev(off, paths.data_ptr())  # best paths stay in HBM
        if world_size > 1 and gather_mode == "all":
            # (AUVP_BENCH_GATHER=all) every rank receives every record: the fixed-stride result records straight from the
            # planner's device buffer + the variable-length paths, waited for inside the step
            ranks.gather_records(ctx.L.auvp_rrt_summaries_dev(ctx.h), E, _lib.SUMMARY_DTYPE.itemsize)
            lens_dev = torch.from_numpy(lens).to(dev)
            ranks.gather.gather_paths(paths[:total], lens_dev)
            gms.append(ranks.gather_ms())
            gbytes.append(E * _lib.SUMMARY_DTYPE.itemsize + 8 * E + total * 56)
        elif world_size > 1:
            # the gather north_star names: final paths (+ the result records and lengths) to rank 0, each rank's block once, over
            # that rank's own xGMI link, ENQUEUED on the handle's gather stream: step k's transfer runs under step k + 1's
            # kernels.  What is sent must outlive the step: the records are copied out of the planner's buffer (1.4 MB), the
            # paths tensor is this step's own.  The previous step's ticket is ended first (one ticket at a time).
            finish_gather()
            from auv_sim_amd import distributed as D
            recs = D.device_records(ctx.L.auvp_rrt_summaries_dev(ctx.h), E, _lib.SUMMARY_DTYPE.itemsize, dev).clone()
            lens_dev = torch.from_numpy(lens).to(dev).reshape(-1, 1)
            torch.cuda.current_stream().synchronize()  # (torch's stream, not the planner's: the two copies above)
            in_flight.append((ranks.gather.root_begin([recs, lens_dev, paths[:total]], rows=[[E] * world_size, [E] * world_size, None], root=0),
                              (recs, lens_dev, paths)))
        return summ

    gather_mode = os.environ.get("AUVP_BENCH_GATHER", "root")
    in_flight, gbytes, gathered = [], [], []

    def finish_gather():
        while in_flight:
            ticket, _keep = in_flight.pop(0)
            got = ranks.gather.root_end(ticket)
            gms.append(getattr(ranks.gather, "last_root_ms", None))
            gbytes.append(getattr(ranks.gather, "last_root_bytes", None))
            if got is not None:  # rank 0: every rank's records, lengths and paths of that step
                gathered[:] = [sum(int(b.shape[0]) for b in got[0]), sum(int(b.shape[0]) for b in got[2])]

    dt, summ = timed_steps(ranks, step, args.steps, args.warmup, finish=finish_gather if world_size > 1 and gather_mode != "all" else None)
    bad = summ["status"] < 0
    if bad.any():
        sys.exit("device error status in %d episodes: %s" % (int(bad.sum()), np.unique(summ["status"][bad])))
    iters_local = float(summ["iters_run"].sum())
    iters_per_step = ranks.sum(iters_local)
    acc_per_step = ranks.sum(float((summ["n_nodes"] - 1).sum()))
    value = iters_per_step * args.steps / dt
    k_ms = float(np.mean(kms[-args.steps:]))
    k_all = ranks.all(k_ms)
    g_all = ranks.all(float(np.mean([g for g in gms[-args.steps:] if g is not None])) if gms and gms[-1] is not None else None)
    gb_all = ranks.all(float(np.mean([g for g in gbytes[-args.steps:] if g is not None])) if gbytes and gbytes[-1] is not None else None)
    rccl_info_all = ranks.all(ranks.rccl_info) if world_size > 1 else None
    out = None
    if rank == 0:
        grid, block, lds = ctx.last_launch()
        exp_ms = float(np.mean([p[0] for p in parts[-args.steps:]]))
        leaf_ms = float(np.mean([p[1] for p in parts[-args.steps:]]))
        per_wave = parts[-1][2]
        kname = ctx.last_rrt_kernel()
        wl = "RRT.exploring %d obst %dx%d cells %d iters x %d episodes/GPU, %s parent sampling" % (
            args.obstacles, args.grid, args.grid, args.iters, E, args.mode)
        out = {
            "metric": "RRT-Dubins node expansions/s (RRT.exploring)", "value": value, "unit": "expansions/s",
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl[:120],
                       "reference": "path_planning/rrt_dubins.py:92 RRT.exploring, Catalina-like synthetic grid (SURVEY 8(d) config 2, O=256)",
                       "episodes_per_gpu": E, "iters": args.iters, "obstacles": args.obstacles,
                       "cells": int(len(world["cells"])), "parallelism": "episodes sharded x%d" % world_size,
                       "gather": ranks.gather.name if ranks.gather is not None else None, "gather_note": ranks.gather_note,
                       "gather_mode": None if world_size == 1 else ("all-gather in step" if gather_mode == "all" else "to rank 0, overlapped"),
                       "gather_bytes_per_rank": None if not gb_all or gb_all[-1] is None else float(np.max([g for g in gb_all if g is not None])),
                       "gather_ms": None if not g_all or g_all[-1] is None else float(np.max([g for g in g_all if g is not None])),
                       "gather_ms_over_step": None if not g_all or g_all[-1] is None else
                       float(np.max([g for g in g_all if g is not None])) / (1e3 * dt / args.steps),
                       # self-check of a multi-GPU run: (world size given, rank, RANK COUNT AS THE RCCL COMMUNICATOR REPORTS IT)
                       # of every rank, and the RCCL image the C-ABI bound
                       "rccl_comm_info_per_rank": rccl_info_all, "rccl_ranks_seen": (rccl_info_all[0][2] if rccl_info_all and rccl_info_all[0] else None),
                       "rccl_library": ranks.rccl_library},
            # one pass of the path = two launches on the handle's stream: the tree expansion (the dominant kernel: this
            # object's achieved / frac are ITS algorithmic bytes over ITS HIP-event time) and the leaf pass (leaf_*: compulsory
            # bytes); pass_8d_* = the whole-pass SURVEY 8(d) figure of earlier rounds, labelled
            "roofline": rrt_pass_rooflines(ctx, summ, "headline", exp_ms, leaf_ms, kname, episodes_per_wavefront=per_wave,
                                           launch_grid=grid, launch_block=block, launch_lds_bytes=lds,
                                           hbm_measured_copy_GBps=HBM_MEASURED.get("copy_GBps"),
                                           note="expansion kernel: fp64 VALU issue binds (valu_issue_frac), HBM does not; "
                                                "leaf pass: latency bound (DESIGN.md 4)"),
            "expansions_per_s_kernel_only": iters_local / (k_ms * 1e-3),
            "expansions_per_s_expansion_kernel_only": iters_local / (exp_ms * 1e-3),
            "kernel_ms_per_rank": k_all, "gather_ms_per_rank": g_all,
            # the result gather of a multi-GPU step (N = 1: none): where the records go, what one rank sends per step, the
            # transfer's own stream time against the step (root mode: it runs under the NEXT step's kernels, so this is not
            # time added to the step), and what rank 0 held after the last step (records, path elements of ALL ranks)
            "gather_mode": None if world_size == 1 else ("all ranks (all-gather, inside the step)" if gather_mode == "all" else
                                                          "to rank 0 (ncclSend / ncclRecv on the gather stream, overlapped with the next step)"),
            "gather_bytes_per_rank": gb_all, "gather_ms_over_step": (None if not g_all or g_all[-1] is None else
                                                                     float(np.max([g for g in g_all if g is not None])) / (1e3 * dt / args.steps)),
            "gather_root_received": gathered or None,
            "expansions_per_step": iters_per_step, "accepted_nodes_per_step": acc_per_step,  # summed over ALL ranks
            "accepted_nodes_per_episode": float((summ["n_nodes"] - 1).mean()),
            "qualifying_leaves_per_episode": float(summ["n_leaves"].mean()),
            "cull_candidates_per_expansion": float(summ["n_candidates"].sum()) / iters_local,
        }
        out["hbm_probe"] = dict(HBM_MEASURED, peak_spec_GBps=HBM_PEAK_GBS,
                                note="auvp_hbm_probe: best of `reps` launches; read = 8 x 16 B loads in flight per lane, copy = bytes read + written")
        if with_cpu:
            out["cpu_baseline"] = cpu_baseline(world, args.iters, args)
            out["cpu_baseline_all_cores"] = cpu_all
        else:
            out["cpu_baseline"] = None
    if not args.no_extra:
        # the other configurations of the path: configs 3, 4 and 5 on every rank (sharded), the rest on rank 0's GPU
        for name in ("single_episode", "rrt_64_obstacles", "rrt_1024_replicas", "rrt_dense", "rrt_nn", "rrt_nn_long_horizon", "astar", "planner_rrt", "rrt_env", "config5", "shark_grid",
                     "particle_filter"):
            if name in sharded or rank == 0:
                try:
                    r = sides[name]()
                except Exception as e:  # a side measurement must not cost the headline line
                    if name in sharded and world_size > 1:
                        raise
                    r = {"error": "%s: %s" % (type(e).__name__, e)}
                if rank == 0:
                    out[name] = r
        if rank == 0 and isinstance(out.get("rrt_nn_long_horizon"), dict) and "roofline" in out["rrt_nn_long_horizon"]:
            # the headline's parent sampling (time bins) reads one node per iteration; the SAME path with nearest-neighbour
            # parent sampling streams the whole tree every iteration -- that side measurement is where the HBM roofline
            # fraction of the path is visible.  Flat scalars: the driver's record keeps scalars of `roofline` only.
            side = out["rrt_nn_long_horizon"]
            nl = side["roofline"]
            out["roofline"].update({
                "nn_long_kernel": "rrt_explore_kernel<4,2,false> (nearest-neighbour parent sampling, max_traj_time 20000 s)",
                "nn_long_episodes": side["episodes"], "nn_long_kernel_ms": nl["kernel_ms"],
                "nn_long_alg_bytes": nl["algorithmic_bytes_per_launch"], "nn_long_achieved_GBps": nl["achieved"],
                "nn_long_frac": nl["frac"], "nn_long_frac_of_measured": nl.get("frac_of_measured"),
                "nn_long_traffic": nl.get("traffic"), "nn_long_traffic_raw": nl.get("traffic_raw"),
                "nn_long_mirror_bytes": side["xy_mirror_working_set_bytes"], "nn_long_iters_per_s": side["value"],
                "nn_long_valu_issue_frac": nl.get("valu_issue_frac"),
                # the second half of that launch alone (mirrors far past the Infinity Cache): the HBM-only figure; nn_long_frac
                # above is HBM + Infinity Cache
                "nn_long_hbm_only_GBps": nl.get("hbm_only_GBps"), "nn_long_hbm_only_frac": nl.get("hbm_only_frac"),
                "nn_long_hbm_only_frac_of_measured": nl.get("hbm_only_frac_of_measured")})
        if rank == 0:
            # the other configurations as flat scalars too (value of each side measurement; details in its own object)
            for name, key in (("astar", "side_astar_cells_per_s"), ("planner_rrt", "side_planner_steps_per_s"),
                              ("config5", "side_config5_steps_per_s"), ("rrt_1024_replicas", "side_replicas_expansions_per_s"),
                              ("single_episode", "side_single_episode_us_per_expansion"),
                              ("particle_filter", "side_pf_particle_steps_per_s"), ("shark_grid", "side_shark_grid_cells_per_s")):
                v = out.get(name)
                if isinstance(v, dict):
                    out["roofline"][key] = v.get("us_per_expansion" if name == "single_episode" else "value")
    try:
        if world_size > 1:
            # the process group goes first: whatever the communicator prints when it is torn down comes BEFORE the result line,
            # and the other ranks are past their last collective when rank 0 prints.  A teardown that raises must not cost the
            # measured result: the line is printed whatever happens here.
            try:
                dist.barrier()
                dist.destroy_process_group()
            except Exception as e:
                print("bench: rank %d: process-group teardown failed (%s: %s); the result line follows" % (rank, type(e).__name__, e),
                      file=sys.stderr, flush=True)
    finally:
        if rank == 0:
            if world_size > 1:
                time.sleep(0.5)  # (after the timed region and every collective: lets the other ranks' exit output drain first)
            emit(out)


if __name__ == "__main__":
    main()
