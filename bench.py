#!/usr/bin/env python3
"""bench.py -- RRT-Dubins node expansions/s on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic input: E independent
RRT.exploring episodes (path_planning/rrt_dubins.py:92) x `--iters` expansions each, on the
256-obstacle 200x200-cell Catalina-like grid of SURVEY.md 8(d) config 2, followed by the extraction
of every episode's best path and -- for N > 1 -- the RCCL gather of the result records.  Inputs
(world tables, start states, seeded MT19937 states) are resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  value = expansions of ALL ranks / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(summ, n_iter):
    """SURVEY.md 8(d) B_exp, evaluated with the launch's own counts (not an estimate):
    48 parent read + 4 bin-index read per expansion; 52 node write + 8 bin append per accepted node;
    56 per stored path point; (24 + 8) per path element walked by the cost function."""
    iters = float(summ["iters_run"].sum())
    nodes = float((summ["n_nodes"] - 1).sum())
    pts = float(summ["n_points"].sum())
    walked = float(summ["leaf_elems"].sum())
    return iters * (48 + 4) + nodes * (52 + 8) + pts * 56 + walked * (24 + 8)


def cpu_baseline(world, n_iter, args):
    """The CPU checker (oracle/, libm build = the restatement pinned to the reference goldens) timed
    on ONE host core over a bounded sample of the same workload."""
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
    return {"value": done / t_used, "unit": "expansions/s", "cores": 1, "kind": "port",
            "sample": "%d episodes x %d iterations of the same workload (seeds 0..%d), oracle/ libm build, %.1f s"
                      % (eps, n_iter, eps - 1, t_used)}


def _cpu_episode(job):
    """pool worker: one oracle episode (runs in a forked child that never touches the GPU)"""
    world, seed, n_iter, mode = job
    from oracle import orc
    w = orc.WorldArrays(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    init = [world["start"][0], world["start"][1], 0, 0, 0, 0]
    return orc.rrt_explore(w, seed, n_iter, mode=mode, init=init, kind="libm", want_path=False)["iters_run"]


def cpu_baseline_all_cores(world, n_iter, args):
    """the same CPU checker on ALL host cores, one episode per core (episodes are the natural
    parallel unit on the CPU too).  Must be called before this process initialises HIP (it forks)."""
    import multiprocessing as mp
    from oracle import orc
    orc.build()
    cores, how = effective_cores()
    jobs = [(world, 1000 + s, n_iter, args.mode) for s in range(cores)]
    with mp.get_context("fork").Pool(cores) as pool:
        pool.map(_cpu_episode, [(world, 0, 10, args.mode)] * cores, chunksize=1)  # start the workers, load the library
        t0 = time.perf_counter()
        done = sum(pool.map(_cpu_episode, jobs, chunksize=1))
        dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "expansions/s", "cores": cores, "kind": "port",
            "sample": "%d episodes x %d iterations, one per usable core (%s; os.cpu_count() = %d), oracle/ libm build, %.1f s"
                      % (cores, n_iter, how, os.cpu_count() or 1, dt)}


def effective_cores(cap=64):
    """cores this process may actually use: scheduler affinity, clipped by the cgroup CPU quota, capped so the
    bounded sample stays bounded"""
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


def bench_single_episode(ctx, world, args, reps=3):
    """SURVEY 8(d) config 2 latency test: ONE episode on one GPU (a serial chain: one wavefront busy)."""
    init = np.zeros((1, 6))
    init[0, 0], init[0, 1] = world["start"]
    ms = []
    for i in range(reps + 1):
        summ = ctx.rrt_explore_batch(init, np.array([7], dtype=np.uint64), args.iters, mode=args.mode, freq=30, bin_interval=5,
                                     v=2, max_traj_time=500.0, weights=(-3, -3, -4))
        if i:
            ms.append(ctx.last_kernel_ms())
    k_ms = float(np.mean(ms))
    return {"metric": "single-episode latency (seed 7)", "kernel_ms": k_ms, "iters": int(summ[0]["iters_run"]),
            "expansions_per_s": float(summ[0]["iters_run"]) / (k_ms * 1e-3), "us_per_expansion": 1e3 * k_ms / float(summ[0]["iters_run"])}


def bench_rrt_o64(ctx, args, n_ep=6144, reps=2):
    """BASELINE configs[1] as written: 64 obstacles (the headline uses 256), same 200x200-cell grid and 10k budget."""
    from auv_sim_amd import synth
    half = 0.5 * args.grid * 10.0
    world = synth.make_world(seed=2, n_obstacles=64, box=(-half, -half, half, half), cell=10.0, n_bins=10, bin_len=50,
                             n_habitats=10)
    ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    init = np.zeros((n_ep, 6))
    init[:, 0], init[:, 1] = world["start"]
    ctx.rrt_prepare(init, np.arange(n_ep, dtype=np.uint64), args.iters, mode=args.mode, freq=30, bin_interval=5, v=2,
                    max_traj_time=500.0, weights=(-3, -3, -4))
    ms = []
    for i in range(reps + 1):
        ctx.rrt_run()
        if i:
            ms.append(ctx.last_kernel_ms())
    summ = ctx.summaries()
    k_ms = float(np.mean(ms))
    return {"metric": "RRT.exploring expansions/s, 64 obstacles", "value": float(summ["iters_run"].sum()) / (k_ms * 1e-3),
            "unit": "expansions/s", "episodes": n_ep, "kernel_ms": k_ms, "accepted_nodes_per_episode": float((summ["n_nodes"] - 1).mean())}


def bench_config5(ctx, n_ep=12500, max_step=200, reps=2):
    """every particle hypothesis of a shark position becomes the goal of one
    Planner_RRT episode with a 200-step budget; 12 500 episodes per GPU (100 000 over 8 GPUs)."""
    from auv_sim_amd import synth
    from auv_sim_amd._prrt_lib import PlannerBatch
    w = synth.make_rect_world(seed=3, n_obstacles=256)
    ctx.set_world(obstacles=w["obstacles"])
    rng = np.random.default_rng(5)
    starts = np.tile(np.array([w["start"][0], w["start"][1], 0.0, 0.0]), (n_ep, 1))
    goals = np.clip(np.asarray(w["goal"])[None, :] + rng.normal(0.0, 25.0, size=(n_ep, 2)), w["rect"][0] + 5, w["rect"][2] - 5)
    seeds = np.arange(n_ep, dtype=np.uint64)
    ms, steps, done = [], 0, 0
    for i in range(reps + 1):
        pb = PlannerBatch(ctx, starts, goals, w["rect"], max_step, seeds=seeds, freq=10, cell=5, subs=1)
        summ = pb.plan()
        if i:
            ms.append(ctx.last_kernel_ms())
        steps, done = int(summ["steps"].sum()), int(summ["done"].sum())
        if (summ["status"] < 0).any():
            return {"error": "episode status %s" % np.unique(summ["status"])}
    k_ms = float(np.mean(ms))
    return {"metric": "config 5: Planner_RRT steps/s, one episode per particle hypothesis", "value": steps / (k_ms * 1e-3),
            "unit": "steps/s", "episodes": n_ep, "max_step": max_step, "steps_per_launch": steps, "episodes_done": done,
            "kernel_ms": k_ms, "episodes_per_s": n_ep / (k_ms * 1e-3)}


def bench_astar(ctx, with_cpu, n_inst=1024, reps=3):
    """BASELINE config 3: 1024 independent astar_fixLenSOG searches (starts on the 10 m lattice,
    pathLenLimit in {100,200,300}) over one shared world: 64 obstacles, 10 habitats, rectangle polygon,
    20x20-cell shark grid (the reference's hard-coded 600x600 visited window bounds the workspace).
    cells/s = neighbour cells that passed the bounds test / time."""
    from auv_sim_amd import _astar_lib, synth
    w = synth.make_world(seed=12, n_obstacles=64, obst_radius=(2.0, 6.0), n_habitats=10, hab_radius=(10.0, 25.0))
    ctx.set_world(w["obstacles"], w["habitats"], w["polygon"], w["bins"], w["cells"], w["prob"])
    rng = np.random.default_rng(3)
    starts = np.column_stack([-290.0 + 10.0 * rng.integers(0, 19, n_inst), -90.0 + 10.0 * rng.integers(0, 19, n_inst)])
    limits = rng.choice([100.0, 200.0, 300.0], n_inst)
    wts = (0, 10, 10, 100)
    kw = dict(limits=limits, weights=wts, velocity=1.0, cap_nodes=20000)
    _astar_lib.run_batch(ctx, "astar_fixLenSOG", starts, **kw)
    ms, cells, exps, found = [], 0, 0, 0
    for _ in range(reps):
        res = _astar_lib.run_batch(ctx, "astar_fixLenSOG", starts, **kw)
        ms.append(ctx.last_kernel_ms())
        cells = sum(r["n_children"] for r in res)
        exps = sum(r["n_expansions"] for r in res)
        found = sum(r["found"] for r in res)
        bad = [r["status"] for r in res if r["status"] < 0]
        if bad:
            return {"error": "instance status %s" % sorted(set(bad))}
    k_ms = float(np.mean(ms))
    out = {"metric": "A* cells/s (astar_fixLenSOG, child cells evaluated)", "value": cells / (k_ms * 1e-3), "unit": "cells/s",
           "instances": n_inst, "cells_per_launch": cells, "expansions_per_launch": exps, "found": found, "kernel_ms": k_ms,
           "config": "1024 x astar_fixLenSOG, 64 obstacles, 10 habitats, 400-cell grid x 10 bins, limits 100/200/300",
           # ~0.1 KB algorithmic HBM bytes per child cell (node write 68 + visited 1 + SOG 16 + scan share)
           "roofline": {"bound": "hbm", "achieved": cells * 100.0 / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": cells * 100.0 / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None}}
    # SURVEY 8(d) config 3 also asks for the same batch through astar_fixLen (no grid) and astar.astar (start -> goal
    # pairs on the 50x50 lattice of config 1): reported side by side, labelled
    def variant(name, st, **k):
        _astar_lib.run_batch(ctx, name, st, **k)
        vms, vc, vf = [], 0, 0
        for _ in range(reps):
            res = _astar_lib.run_batch(ctx, name, st, **k)
            vms.append(ctx.last_kernel_ms())
            vc, vf = sum(r["n_children"] for r in res), sum(r["found"] for r in res)
            if any(r["status"] < 0 for r in res):
                return {"error": "instance status %s" % sorted({r["status"] for r in res if r["status"] < 0})}
        return {"value": vc / (float(np.mean(vms)) * 1e-3), "unit": "cells/s", "cells_per_launch": vc, "found": vf,
                "kernel_ms": float(np.mean(vms))}
    out["variants"] = {"astar_fixLen": variant("astar_fixLen", starts, limits=limits, weights=(0, 10, 10), cap_nodes=20000)}
    lw = synth.make_lattice_world(seed=11, n_obstacles=30, r_range=(10, 22))
    ctx.set_world(lw["obstacles"], None, None, None, None, None)
    lst = np.column_stack([10.0 * rng.integers(0, 20, n_inst), 10.0 * rng.integers(0, 20, n_inst)])
    out["variants"]["astar"] = variant("astar", lst, goals=np.tile([490.0, 490.0], (n_inst, 1)), box=lw["box"], cap_nodes=60000)
    if with_cpu:
        from oracle import orc_astar as oa
        t0, c, n = time.perf_counter(), 0, 0
        while time.perf_counter() - t0 < 5.0 and n < n_inst:
            r = oa.run("astar_fixLenSOG", starts[n], obstacles=w["obstacles"], habitats=w["habitats"], polygon=w["polygon"],
                       bins=w["bins"], cells=w["cells"], prob=w["prob"], limit=float(limits[n]), weights=wts, velocity=1.0,
                       cap_nodes=20000, kind="libm")
            c += r["n_children"]
            n += 1
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": c / dt, "unit": "cells/s", "cores": 1, "kind": "port",
                               "sample": "first %d of the %d instances, oracle/ libm build, %.1f s" % (n, n_inst, dt)}
    return out


def bench_planner(ctx, with_cpu, n_ep=512, max_step=2000, reps=3):
    """BASELINE config 4: 512 Planner_RRT.planning(max_step=2000) episodes, 200 m x 200 m rectangle, 256
    obstacles, cell 5 m, 1 theta subsection, freq 10, start (20,20) -> goal (170,180), seed = episode id."""
    from auv_sim_amd import synth
    from auv_sim_amd._prrt_lib import PlannerBatch
    w = synth.make_rect_world(seed=3, n_obstacles=256)
    ctx.set_world(obstacles=w["obstacles"])
    starts = np.tile(np.array([w["start"][0], w["start"][1], 0.0, 0.0]), (n_ep, 1))
    goals = np.tile(w["goal"], (n_ep, 1))
    seeds = np.arange(n_ep, dtype=np.uint64)
    ms, steps, done = [], 0, 0
    for i in range(reps + 1):
        pb = PlannerBatch(ctx, starts, goals, w["rect"], max_step, seeds=seeds, freq=10, cell=5, subs=1)
        summ = pb.plan()
        if i:
            ms.append(ctx.last_kernel_ms())
        steps, done = int(summ["steps"].sum()), int(summ["done"].sum())
        if (summ["status"] < 0).any():
            return {"error": "episode status %s" % np.unique(summ["status"])}
    k_ms = float(np.mean(ms))
    out = {"metric": "Planner_RRT steps/s (generate_one_node calls)", "value": steps / (k_ms * 1e-3), "unit": "steps/s",
           "episodes": n_ep, "steps_per_launch": steps, "episodes_done": done, "kernel_ms": k_ms,
           "config": "512 x Planner_RRT.planning(2000), 200 m env, 256 obstacles, cell 5 m, freq 10",
           # ~0.33 KB algorithmic HBM bytes per step
           "roofline": {"bound": "hbm", "achieved": steps * 330.0 / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": steps * 330.0 / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None}}
    if with_cpu:
        from oracle import orc_planner as op
        t0, c, n = time.perf_counter(), 0, 0
        while time.perf_counter() - t0 < 5.0 and n < n_ep:
            r = op.planning(w["obstacles"], w["rect"], starts[n], goals[n], n, max_step, 10, 5, 1, kind="libm")
            c += r["steps"]
            n += 1
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": c / dt, "unit": "steps/s", "cores": 1, "kind": "port",
                               "sample": "first %d of the %d episodes, oracle/ libm build, %.1f s" % (n, n_ep, dt)}
    return out


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
    # per launch the particle state (5 doubles + 1 id) is read once and written once (it lives in LDS across the
    # S steps); per step and particle the per-AUV weights make one 8-byte round trip through L2/HBM scratch
    abytes = F * N * (2 * 44.0 + S * A * 24.0) + S * F * A * 40.0
    out = {"metric": "particle filter particle-steps/s (create_and_update + update_weights)", "value": units / (k_ms * 1e-3),
           "unit": "particle-steps/s", "filters": F, "particles": N, "steps": S, "auvs": A, "kernel_ms": k_ms,
           "draws32_per_filter_step": float(nd.mean()) / S,
           "config": "%d filters x %d particles x %d steps, %d AUV measurements per step" % (F, N, S, A),
           "roofline": {"bound": "hbm", "achieved": abytes / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": abytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "note": "state is LDS resident across the steps of a launch: barrier/latency bound, not HBM bound"}}
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
    window = 81  # cells of the radius-5 disc each output cell sums per shark
    abytes = T * n_sharks * G * (4 + 8.0) + units * (n_sharks * window * 8.0 + 8.0)
    out = {"metric": "SharkOccupancyGrid.convert output cells/s", "value": units / (k_ms * 1e-3), "unit": "grid cells/s",
           "bins": int(T), "grid": [int(grids.shape[1]), int(grids.shape[2])], "sharks": n_sharks, "kernel_ms": k_ms,
           "config": "%dx%d cells of 10 m, %d sharks x %d points, %d bins, detect range 50 m" % (n, n, n_sharks, n_pts, T),
           "roofline": {"bound": "hbm", "achieved": abytes / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": abytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "note": "algorithmic bytes count every window read; they are L2 hits, so achieved can exceed HBM traffic"}}
    if with_cpu:
        from oracle import orc_sog
        sub = 2  # sharks in the CPU sample (the scalar port scans cell_list per point like the reference)
        k = sub * n_pts
        t0 = time.perf_counter()
        r = orc_sog.convert(cells, box, cs, 30.0, 50.0, traj_len[:sub], pts[:k], kind="libm")
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": float(len(r["grids"])) * G * (sub / float(n_sharks)) / dt, "unit": "grid cells/s",
                               "cores": 1, "kind": "port",
                               "sample": "%d of the %d sharks (cost is linear in sharks; value scaled by %d/%d), %.1f s"
                                         % (sub, n_sharks, sub, n_sharks, dt)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-extra", action="store_true", help="skip the A* / Planner_RRT side measurements")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--episodes", type=int, default=12288,
                    help="episodes per GPU per step (6144 wavefronts are resident at once, 6 per SIMD: 12288 = two rounds, "
                         "the second one back-fills as episodes of the first finish; 16.9 MB of tree storage each = 208 GB)")
    ap.add_argument("--iters", type=int, default=10000, help="expansion budget per episode (10k-node budget)")
    ap.add_argument("--obstacles", type=int, default=256)
    ap.add_argument("--grid", type=int, default=200, help="grid is grid x grid cells of 10 m")
    ap.add_argument("--mode", default="timebin", choices=["timebin", "nn", "plantime"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size != args.gpus:
        if world_size == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        args.gpus = world_size

    from auv_sim_amd import synth
    half = 0.5 * args.grid * 10.0
    world = synth.make_world(seed=2, n_obstacles=args.obstacles, box=(-half, -half, half, half), cell=10.0,
                             n_bins=10, bin_len=50, n_habitats=10)
    cpu_all = None
    if world_size == 1 and not args.no_cpu:
        cpu_all = cpu_baseline_all_cores(world, args.iters, args)  # forks: must precede any HIP initialisation

    import torch
    import torch.distributed as dist
    from auv_sim_amd import _lib
    from auv_sim_amd import distributed as D

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=dev)

    ctx = _lib.Context(local_rank)
    ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    E = args.episodes
    # the trees need ~17.5 MB of HBM per 10 000-iteration episode: if this GPU has less free than the requested batch
    # needs, run the largest whole number of resident rounds (6144 episodes) that fits -- every rank the same
    free_b, _total_b = torch.cuda.mem_get_info(dev)
    fit = int(0.92 * free_b / (17.5e6 * max(args.iters, 1) / 10000.0))
    if fit < E:
        fit = max(6144 * (fit // 6144), min(E, 1024))
    e_fit = torch.tensor([min(E, fit)], dtype=torch.int64, device=dev)
    if world_size > 1:
        dist.all_reduce(e_fit, op=dist.ReduceOp.MIN)
    if int(e_fit.item()) < E:
        print("bench: %d episodes per GPU do not fit the free HBM (%.0f GB); running %d" % (E, free_b / 1e9, int(e_fit.item())),
              file=sys.stderr)
        E = int(e_fit.item())
    init = np.zeros((E, 6))
    init[:, 0], init[:, 1] = world["start"]
    seeds = np.arange(rank * E, (rank + 1) * E, dtype=np.uint64)  # global episode id = seed
    ctx.rrt_prepare(init, seeds, args.iters, mode=args.mode, freq=30, bin_interval=5, v=2, max_traj_time=500.0,
                    weights=(-3, -3, -4))

    def step():
        """kernel + best-path extraction (+ RCCL gather of the result records for N > 1)"""
        ctx.rrt_run()
        ms = ctx.last_kernel_ms()
        summ = ctx.summaries()
        lens = np.where(summ["best_leaf"] >= 0, summ["best_path_len"], 0).astype(np.int64)
        off = np.zeros(E + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        total = int(off[-1])
        paths = torch.empty((max(total, 1), 7), dtype=torch.float64, device=dev)
        ctx.paths_dev(off, paths.data_ptr())  # best paths stay in HBM
        if world_size > 1:
            # RCCL gather of the fixed-stride result records + the variable-length paths (two-phase)
            D.gather_records(D.summaries_to_tensor(summ, dev))
            D.gather_paths(paths[:total], torch.from_numpy(lens).to(dev))
        return ms, summ

    def fence():
        torch.cuda.synchronize()
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    kms = []
    summ = None
    for _ in range(args.steps):
        ms, summ = step()
        kms.append(ms)
    fence()
    dt = time.perf_counter() - t0
    if world_size > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    bad = summ["status"] < 0
    if bad.any():
        sys.exit("device error status in %d episodes: %s" % (int(bad.sum()), np.unique(summ["status"][bad])))
    iters_per_step = float(summ["iters_run"].sum()) * world_size  # identical budget on every rank
    value = iters_per_step * args.steps / dt
    if rank == 0:
        k_ms = float(np.mean(kms))
        abytes = algorithmic_bytes(summ, args.iters)
        achieved = abytes / (k_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        pmc = os.path.join(REPO, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                # PMC passes of tools/profile_bench.sh on the default workload (one wave = one episode, 10 000 iterations);
                # a different batch is scaled by its number of expansions and says so
                pj = json.load(open(pmc))
                waves = float(pj["per_launch"].get("SQ_WAVES", 0.0))
                traffic = pj.get("hbm_bytes_per_launch")
                traffic_src = "profiles/%s/pmc_summary.json (FETCH_SIZE + WRITE_SIZE passes)" % pj.get("tag", "?")
                if traffic is not None and waves > 0 and (int(waves) != E or args.iters != 10000):
                    scale = (E * args.iters) / (waves * 10000.0)
                    traffic *= scale
                    traffic_src += ", scaled x%.3f from %d episodes x 10000 iterations" % (scale, int(waves))
            except Exception:
                traffic, traffic_src = None, None
        grid, block, lds = ctx.last_launch()
        out = {
            "metric": "RRT-Dubins node expansions/s (RRT.exploring)", "value": value, "unit": "expansions/s",
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "rrt_dubins.py RRT.exploring, %d obstacles, %dx%d-cell Catalina-like grid, "
                                   "%d-iteration budget per episode, %d episodes per GPU per step, %s parent sampling"
                                   % (args.obstacles, args.grid, args.grid, args.iters, E, args.mode),
                       "episodes_per_gpu": E, "iters": args.iters, "obstacles": args.obstacles,
                       "cells": int(len(world["cells"])), "parallelism": "episodes sharded x%d" % world_size},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "rrt_explore_kernel", "kernel_ms": k_ms, "algorithmic_bytes_per_launch": abytes,
                         "bytes_per_expansion": abytes / float(summ["iters_run"].sum()),
                         "launch": {"grid": grid, "block": block, "lds_bytes": lds},
                         "note": "fp64-VALU/latency bound at these sizes, not HBM bound (DESIGN.md)"},
            "expansions_per_s_kernel_only": float(summ["iters_run"].sum()) / (k_ms * 1e-3),
            "accepted_nodes_per_episode": float((summ["n_nodes"] - 1).mean()),
            "qualifying_leaves_per_episode": float(summ["n_leaves"].mean()),
        }
        if world_size == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(world, args.iters, args)
            out["cpu_baseline_all_cores"] = cpu_all
        else:
            out["cpu_baseline"] = None
        if not args.no_extra:
            # the other two planner families of the path, per GPU (rank 0's device), outside the timed region
            out["single_episode"] = bench_single_episode(ctx, world, args)
            out["rrt_64_obstacles"] = bench_rrt_o64(ctx, args)
            out["astar"] = bench_astar(ctx, with_cpu=(world_size == 1 and not args.no_cpu))
            out["planner_rrt"] = bench_planner(ctx, with_cpu=(world_size == 1 and not args.no_cpu))
            # the callers either side of the planners
            out["config5"] = bench_config5(ctx)
            out["shark_grid"] = bench_shark_grid(local_rank, with_cpu=(world_size == 1 and not args.no_cpu))
            out["particle_filter"] = bench_particle_filter(local_rank, with_cpu=(world_size == 1 and not args.no_cpu))
        print(json.dumps(out))
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
