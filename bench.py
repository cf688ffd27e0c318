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
from bench_sides.common import *  # noqa: E402,F401,F403  (byte formulas, roofline, counter passes, CPU baselines, timed_steps, _rrt_batch)
from bench_sides.common import _rrt_batch, _pmc, _cpu_episode  # noqa: E402,F401
from bench_sides.rrt import bench_single_episode, bench_rrt_o64, bench_rrt_replicas, bench_rrt_nn, bench_rrt_dense  # noqa: E402
from bench_sides.astar import astar_inputs, bench_astar  # noqa: E402,F401
from bench_sides.planner import bench_rrt_env, bench_planner, bench_config5  # noqa: E402
from bench_sides.filters import bench_particle_filter, bench_shark_grid  # noqa: E402


LINE_LIMIT = 4096
SIDES_FILE = os.environ.get("AUVP_BENCH_SIDES", os.path.join(REPO, "bench_sides.json"))
ROOF_KEEP = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "algorithmic_bytes_per_launch", "stream_kernel_ms", "stream_write_GBps", "stream_bytes_read",
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

    def __init__(self, ctx, rank, world, dev, use_rccl=True, cpu_group=None):
        """`cpu_group`: torch.distributed runs over gloo (host tensors); by default exactly when the C-ABI transport is not
        used.  AUVP_BENCH_ONE_GPU=1 with AUVP_RCCL_LIBRARY set (a stand-in RCCL: tests/mock_rccl) has both: the ranks' own
        bookkeeping over gloo, the result gather through libauvplan.so's auvp_gather* entry points."""
        self.ctx, self.rank, self.world, self.dev = ctx, rank, world, dev
        self.gather, self.gather_note = None, None
        self.rccl_info, self.rccl_library = None, None
        self.cpu_group = (not use_rccl) if cpu_group is None else cpu_group  # gloo: collectives on host tensors
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
            pre = self._t([0 if D.RcclGather.usable() else 1], torch.int64)
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
            flag = self._t([0 if self.gather is not None else 1], torch.int64)
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
    # (one GPU + a stand-in RCCL named by AUVP_RCCL_LIBRARY: the C-ABI transport runs too -- tests/test_gpu_bench_two_ranks.py)
    ranks = Ranks(ctx, rank, world_size, dev, use_rccl=(not one_gpu) or bool(os.environ.get("AUVP_RCCL_LIBRARY")), cpu_group=one_gpu)
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
        """expansion + leaf pass + best-path extraction (+ RCCL gather of the result records for N > 1).  From the second step on the
        library runs the pass as three launches -- the episodes' random numbers generated ahead (rrt_stream_kernel), the expansion
        reading them (rrt_rows_stream_kernel), the leaves -- and ALL THREE are inside the step: the stream is written again in
        every pass, nothing is kept from the step before but the buffer and its length"""
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
