"""bench_sides/rrt.py -- side measurements of RRT.exploring: one episode, 64 obstacles, 1 024 replicas, nearest-neighbour sampling, dense worlds  (split out of bench.py in round 6)"""
import json
import os
import sys
import time

import numpy as np

from .common import *  # noqa: F401,F403
from .common import _rrt_batch  # noqa: F401


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
    # (AUVP_BENCH_PROFILE=1, set by tools/profile_bench.sh: the half-budget launches below run the same kernel and would be averaged
    # into the profiler's per-kernel duration and counters of this measurement; the profile is of the full launch alone)
    if long_horizon and "roofline" in out and os.environ.get("AUVP_BENCH_PROFILE") != "1":
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
