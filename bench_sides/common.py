"""bench_sides/common.py -- what the headline (bench.py) and every side measurement share: the algorithmic-byte formulas of
SURVEY.md 8(d), the roofline object, the committed counter passes (profiles/pmc_latest.json), the synthetic bench world, the
CPU baselines (the oracle, timed on the host cores -- the only place besides tests/ that touches oracle/), the timed loop.
Split out of bench.py in round 6 (it had grown to 1 500 lines); bench.py re-exports these names."""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
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


SIDE_KEEP = ("bound", "achieved", "peak", "unit", "frac", "kernel", "kernel_ms", "algorithmic_bytes_per_launch", "frac_of_measured", "stream_kernel_ms",
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
    # the random numbers generated ahead (rrt_stream_kernel + rrt_rows_stream_kernel: every batch after the first with a
    # parameter set): a third launch.  Its bytes are derived data, not part of B_exp -- `achieved` stays on the SURVEY 8(d)
    # bytes; what the stream moves is reported beside it
    stream_ms, stream_len = ctx.last_stream_ms(), ctx.last_stream_len()
    streamed = kname == "rrt_rows_stream_kernel" and stream_len > 0
    if streamed:
        extra = dict(extra, stream_kernel="rrt_stream_kernel", stream_kernel_ms=stream_ms,
                     stream_bytes_written=8.0 * stream_len * len(summ), stream_bytes_read=4.0 * float(summ["n_draw32"].sum()),
                     stream_write_GBps=8.0 * stream_len * len(summ) / (stream_ms * 1e-3) / 1e9 if stream_ms > 0 else None)
    whole = pmc_traffic(meas, [kname, "rrt_leaf_kernel"] + (["rrt_stream_kernel"] if streamed else []), iters)
    comparable = whole["traffic"] is not None
    tx2, traw = pmc_kernel_traffic(meas, kname, iters) if comparable else (None, None)
    lx2, lraw = pmc_kernel_traffic(meas, "rrt_leaf_kernel", iters) if comparable else (None, None)
    vi = pmc_valu_issue(meas, kname) if comparable else None
    if vi is None:  # (a side batch of a kernel that has a pass elsewhere: its instructions per expansion over this launch's time)
        vi = pmc_valu_issue_est(kname, iters, exp_ms)
    lvi = pmc_valu_issue(meas, "rrt_leaf_kernel") if comparable else None
    leaf_ach = a_leaf / (leaf_ms * 1e-3) / 1e9 if leaf_ms > 0 else 0.0
    pass_ms = exp_ms + leaf_ms + (stream_ms if streamed else 0.0)
    pass_ach = a_8d / (pass_ms * 1e-3) / 1e9
    r = roofline(a_exp, exp_ms, kname, {"traffic": tx2, "traffic_raw": traw, "traffic_source": whole["traffic_source"]},
                 valu_issue_frac=vi, bytes_per_expansion=a_exp / iters,
                 leaf_kernel="rrt_leaf_kernel", leaf_kernel_ms=leaf_ms, leaf_compulsory_bytes=a_leaf, leaf_achieved=leaf_ach,
                 leaf_frac=leaf_ach / HBM_PEAK_GBS, leaf_valu_issue_frac=lvi, leaf_traffic=lx2, leaf_traffic_raw=lraw,
                 leaf_nodes_visited=st["nodes_visited"], leaf_points_visited=st["points_visited"],
                 leaf_elements_resummed=st["elements_resummed"], leaf_bound="latency (scattered 264-B runs; DESIGN.md)",
                 pass_kernel_ms=pass_ms, pass_8d_bytes=a_8d, pass_8d_bytes_per_expansion=a_8d / iters,
                 pass_8d_achieved=pass_ach, pass_8d_frac=pass_ach / HBM_PEAK_GBS,
                 pass_traffic=whole["traffic"], pass_traffic_raw=whole["traffic_raw"],
                 pass_8d_note="SURVEY 8(d) B_exp x expansions / both launches: bills q L 32 B of leaf->root walks the leaf pass does not move",
                 **extra)
    if HBM_MEASURED.get("read_GBps"):
        r["leaf_frac_of_measured"] = leaf_ach / HBM_MEASURED["read_GBps"]
    if compact:  # side measurements: the scalars that matter, no prose (the headline's object explains the fields)
        r = {k: r[k] for k in SIDE_KEEP if k in r}
    return r


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
