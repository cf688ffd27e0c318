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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--episodes", type=int, default=4096, help="episodes per GPU per step")
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

    import torch
    import torch.distributed as dist
    from auv_sim_amd import _lib, synth
    from auv_sim_amd import distributed as D

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=dev)

    half = 0.5 * args.grid * 10.0
    world = synth.make_world(seed=2, n_obstacles=args.obstacles, box=(-half, -half, half, half), cell=10.0,
                             n_bins=10, bin_len=50, n_habitats=10)
    ctx = _lib.Context(local_rank)
    ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    E = args.episodes
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
        traffic = None
        pmc = os.path.join(REPO, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
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
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
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
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
