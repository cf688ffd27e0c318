"""bench_sides/planner.py -- Planner_RRT: the batched RRTEnv, config 4 (512 episodes), config 5 (one replan per particle)  (split out of bench.py in round 6)"""
import json
import os
import sys
import time

import numpy as np

from .common import *  # noqa: F401,F403
from .common import _rrt_batch  # noqa: F401


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
