#!/usr/bin/env python3
"""Reference-Python CPU timing on the BENCH workloads (SURVEY.md 8(d) "CPU baseline").

TEST INFRASTRUCTURE, build container only: imports the reference planners from /root/reference through the
stub/virtual-clock recipes of tests/golden/make_golden.py (never shipped, never run on the GPU box), runs them
UNWRAPPED (no logging hooks) on exactly the worlds bench.py uses, with 1 process and with one process per core,
and writes profiles/r2_reference_timing.json, which BASELINE.md tabulates and bench.py quotes next to its live
`cpu_baseline` (marked "recorded").  The same script times the C port (oracle/, libm build) on the same inputs in
the same container so the port/reference ratio is stated.

  config 1  astar.astar, 50x50 lattice, 10 obstacles                       (plumbing)
  config 2  RRT.exploring, 10 000 iterations, 200x200 cells, O = 256 (headline) and O = 64 (as written)
  config 3  astar_fixLenSOG, sample of the 1 024-instance batch
  config 4  Planner_RRT.planning(2000), 200 m env, O = 256, sample of the 512 episodes

usage: python tests/experiments/ref_timing.py [--procs 8] [--only c1,c2,c3,c4] [--c2-iters 10000]
"""
import argparse
import contextlib
import io
import json
import multiprocessing as mp
import os
import random
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))


def _mg():
    import make_golden as mg  # the import recipes (stubs, sys.path orders) live there
    return mg


def bench_world(obstacles):
    from auv_sim_amd import synth
    return synth.make_world(seed=2, n_obstacles=obstacles, box=(-1000.0, -1000.0, 1000.0, 1000.0), cell=10.0, n_bins=10,
                            bin_len=50, n_habitats=10)


# ---------------------------------------------------------------- config 2
def ref_exploring(job):
    obstacles, seed, n_iter = job[:3]
    mode, max_traj = (job[3], job[4]) if len(job) > 3 else ("timebin", 500.0)
    mg = _mg()
    world = bench_world(obstacles)
    rrt_mod, mpsm, _ = mg.import_rrt()
    MPS = mpsm.Motion_plan_state
    obs, habitats, poly, cell_list, shark = mg.ref_world(world, MPS)
    rrt_mod.time = mg.refstubs.VirtualClock()
    rrt = rrt_mod.RRT(poly, obs, shark, cell_list, dist_to_end=2, diff_max=0.5, freq=30)
    random.seed(seed)
    start = MPS(float(world["start"][0]), float(world["start"][1]))
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        try:
            res = rrt.exploring(start, habitats, float(n_iter), 5, 2, 50, traj_time_stamp=(mode == "timebin"),
                                max_plan_time=float(n_iter), max_traj_time=max_traj, plan_time=(mode != "nn"), weights=[-3, -3, -4])
            cost = float(res["cost"][0])
        except TypeError:
            cost = None
    dt = time.perf_counter() - t0
    return {"seconds": dt, "iters": n_iter, "nodes": len(rrt.mps_list), "cost": cost, "seed": seed}


def port_exploring(job):
    obstacles, seed, n_iter = job[:3]
    mode, max_traj = (job[3], job[4]) if len(job) > 3 else ("timebin", 500.0)
    from oracle import orc
    world = bench_world(obstacles)
    w = orc.WorldArrays(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    init = [world["start"][0], world["start"][1], 0, 0, 0, 0]
    t0 = time.perf_counter()
    r = orc.rrt_explore(w, seed, n_iter, mode=mode, init=init, kind="libm", want_path=False, max_traj_time=max_traj)
    dt = time.perf_counter() - t0
    return {"seconds": dt, "iters": r["iters_run"], "nodes": r["n_nodes"], "cost": float(r["best_cost"][0]), "seed": seed}


# ---------------------------------------------------------------- config 3
def astar_inputs(n_inst=1024):
    from auv_sim_amd import synth
    w = synth.make_world(seed=12, n_obstacles=64, obst_radius=(2.0, 6.0), n_habitats=10, hab_radius=(10.0, 25.0))
    rng = np.random.default_rng(3)
    starts = np.column_stack([-290.0 + 10.0 * rng.integers(0, 19, n_inst), -90.0 + 10.0 * rng.integers(0, 19, n_inst)])
    limits = rng.choice([100.0, 200.0, 300.0], n_inst)
    return w, starts, limits


def ref_sog(job):
    idx = job
    mg = _mg()
    w, starts, limits = astar_inputs()
    mod, mpsm = mg.import_astar("astar_fixLenSOG")
    MPS = mpsm.Motion_plan_state
    obstacles, habitats, poly, cell_list, shark = mg.ref_world(w, MPS)
    mod.splitCell = lambda polygon, size: cell_list
    bnd = [MPS(p[0], p[1]) for p in w["polygon"].tolist()]
    from oracle import orc_astar as oa
    out = []
    for i in idx:
        start = (float(starts[i][0]), float(starts[i][1]))
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            solver = mod.astar(start, obstacles, bnd, habitats, shark, {}, 1)
            solver.astar(float(limits[i]), [0, 10, 10, 100], {})
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        r = oa.run("astar_fixLenSOG", starts[i], obstacles=w["obstacles"], habitats=w["habitats"], polygon=w["polygon"],
                   bins=w["bins"], cells=w["cells"], prob=w["prob"], limit=float(limits[i]), weights=(0, 10, 10, 100),
                   velocity=1.0, cap_nodes=20000, kind="libm")
        dp = time.perf_counter() - t1
        out.append({"instance": int(i), "seconds": dt, "cells": int(r["n_children"]), "expansions": int(r["n_expansions"]),
                    "port_seconds": dp})
    return out


# ---------------------------------------------------------------- config 4
def ref_planner(job):
    seed, max_step = job
    mg = _mg()
    from auv_sim_amd import synth
    w = synth.make_rect_world(seed=3, n_obstacles=256)
    mod, mpsm = mg.import_gym_rrt()
    MPS = mpsm.Motion_plan_state
    mod.time = mg.refstubs.VirtualClock()
    obs = [MPS(o[0], o[1], size=o[2]) for o in w["obstacles"].tolist()]
    bnd = [MPS(w["rect"][0], w["rect"][1]), MPS(w["rect"][2], w["rect"][3])]
    s = MPS(float(w["start"][0]), float(w["start"][1]), z=-5.0, theta=0.0)
    g = MPS(float(w["goal"][0]), float(w["goal"][1]), z=-5.0, theta=0.0)
    with contextlib.redirect_stdout(io.StringIO()):
        rrt = mod.Planner_RRT(s, g, bnd, obs, [], freq=10, cell_side_length=5, subsections_in_cell=1)
        random.seed(seed)
        t0 = time.perf_counter()
        path, step, _ = rrt.planning(max_step=max_step)
        dt = time.perf_counter() - t0
    from oracle import orc_planner as op
    t1 = time.perf_counter()
    r = op.planning(w["obstacles"], w["rect"], [w["start"][0], w["start"][1], 0.0, 0.0], w["goal"], seed, max_step, 10, 5, 1,
                    kind="libm")
    dp = time.perf_counter() - t1
    assert r["steps"] == step, (r["steps"], step)
    return {"seed": seed, "seconds": dt, "steps": int(step), "port_seconds": dp}


# ---------------------------------------------------------------- config 1
def ref_astar_cfg1(_):
    mg = _mg()
    from auv_sim_amd import synth
    w = synth.make_lattice_world(seed=0, n_obstacles=10)
    mod, mpsm = mg.import_astar("astar")
    MPS = mpsm.Motion_plan_state
    obs = [MPS(o[0], o[1], size=o[2]) for o in w["obstacles"].tolist()]
    box = w["box"].tolist()
    bnd = [MPS(box[0], box[1]), MPS(box[2], box[3])]
    best = None
    for _rep in range(5):
        solver = mod.astar((0, 0), (490, 490), obs, bnd)
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            path = solver.astar(obs, (0, 0), (490, 490))
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    from oracle import orc_astar as oa
    t1 = time.perf_counter()
    r = oa.run("astar", np.array([0.0, 0.0]), obstacles=w["obstacles"], goal=np.array([490.0, 490.0]), box=w["box"], cap_nodes=60000,
               kind="libm")
    dp = time.perf_counter() - t1
    return {"seconds": best, "path_nodes": len(path), "cells": int(r["n_children"]), "expansions": int(r["n_expansions"]),
            "port_seconds": dp}


def pool_map(fn, jobs, procs):
    if procs == 1:
        t0 = time.perf_counter()
        res = [fn(j) for j in jobs]
        return res, time.perf_counter() - t0
    with mp.get_context("fork").Pool(procs) as pool:
        t0 = time.perf_counter()
        res = pool.map(fn, jobs, chunksize=1)
        return res, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--only", default="c1,c3,c4,c2")
    ap.add_argument("--c2-iters", type=int, default=10000)
    ap.add_argument("--out", default=os.path.join(REPO, "profiles", "r2_reference_timing.json"))
    args = ap.parse_args()
    from oracle import orc
    orc.build()
    P = args.procs
    out = {"host": "build container: %d vCPU (os.cpu_count()), CPython %s, numpy %s" % (os.cpu_count(), sys.version.split()[0], np.__version__),
           "note": "reference = /root/reference imported with third-party stubs + virtual clock (tests/golden/make_golden.py); "
                   "port = oracle/ libm build (bit-identical to the reference on the goldens); same inputs, same container",
           "procs_many": P}
    if os.path.exists(args.out):
        try:
            out.update({k: v for k, v in json.load(open(args.out)).items() if k.startswith("config") or k.startswith("rrt_")})
        except Exception:
            pass
    only = args.only.split(",")
    if "c1" in only:
        out["config1_astar"] = ref_astar_cfg1(None)
        r = out["config1_astar"]
        r["ref_cells_per_s"] = r["cells"] / r["seconds"]
        r["port_cells_per_s"] = r["cells"] / r["port_seconds"]
        print("config1", r, flush=True)
    if "c3" in only:
        n_sample = 32
        idx = list(range(n_sample))
        one, t_one = pool_map(ref_sog, [idx[:8]], 1)
        many, t_many = pool_map(ref_sog, [idx[k::P] for k in range(P)], P)
        flat1 = [x for part in one for x in part]
        flatm = [x for part in many for x in part]
        out["config3_astar_fixLenSOG"] = {
            "sample_1proc": "instances 0..7 of the 1024", "sample_many": "instances 0..31, one slice per process",
            "ref_cells_per_s_1proc": sum(x["cells"] for x in flat1) / sum(x["seconds"] for x in flat1),
            "ref_cells_per_s_%dproc" % P: sum(x["cells"] for x in flatm) / t_many,
            "port_cells_per_s_1proc": sum(x["cells"] for x in flat1) / sum(x["port_seconds"] for x in flat1),
            "mean_cells_per_instance": float(np.mean([x["cells"] for x in flatm])),
            "mean_ref_seconds_per_instance": float(np.mean([x["seconds"] for x in flatm])),
        }
        print("config3", out["config3_astar_fixLenSOG"], flush=True)
    if "c4" in only:
        one, t_one = pool_map(ref_planner, [(0, 2000)], 1)
        many, t_many = pool_map(ref_planner, [(s, 2000) for s in range(P)], P)
        out["config4_planner_rrt"] = {
            "sample_1proc": "episode seed 0", "sample_many": "episodes seed 0..%d, one per process" % (P - 1),
            "ref_steps_per_s_1proc": one[0]["steps"] / one[0]["seconds"],
            "ref_steps_per_s_%dproc" % P: sum(x["steps"] for x in many) / t_many,
            "port_steps_per_s_1proc": one[0]["steps"] / one[0]["port_seconds"],
            "steps": [x["steps"] for x in many],
        }
        print("config4", out["config4_planner_rrt"], flush=True)
    if "c2" in only:
        n_iter = args.c2_iters
        for O in (64, 256):
            key = "config2_rrt_exploring_o%d" % O
            one, t_one = pool_map(ref_exploring, [(O, 7, n_iter)], 1)
            pone, _ = pool_map(port_exploring, [(O, 7, n_iter)], 1)
            rec = {"iters": n_iter, "sample_1proc": "seed 7 (SURVEY 8(d) config 2)",
                   "ref_expansions_per_s_1proc": n_iter / one[0]["seconds"], "ref_seconds_1proc": one[0]["seconds"],
                   "port_expansions_per_s_1proc": pone[0]["iters"] / pone[0]["seconds"],
                   "ref_nodes": one[0]["nodes"], "port_nodes": pone[0]["nodes"], "ref_cost": one[0]["cost"], "port_cost": pone[0]["cost"]}
            rec["port_over_ref"] = rec["port_expansions_per_s_1proc"] / rec["ref_expansions_per_s_1proc"]
            out[key] = rec
            print(key, rec, flush=True)
            json.dump(out, open(args.out, "w"), indent=1)
            if O == 256:
                many, t_many = pool_map(ref_exploring, [(O, s, n_iter) for s in range(P)], P)
                rec["sample_many"] = "seeds 0..%d, one episode per process" % (P - 1)
                rec["ref_expansions_per_s_%dproc" % P] = P * n_iter / t_many
                pmany, tp_many = pool_map(port_exploring, [(O, s, n_iter) for s in range(P)], P)
                rec["port_expansions_per_s_%dproc" % P] = sum(x["iters"] for x in pmany) / tp_many
                print(key, "many", rec, flush=True)
    if "nn" in only:
        # round 3: nearest-neighbour parent sampling (plan_time=False, traj_time_stamp=False) on the headline world, at the
        # headline's max_traj_time and with a horizon long enough for the 10k-node budget to end the tree (bench.py rrt_nn*)
        n_iter = args.c2_iters
        for key, max_traj in (("rrt_exploring_nn_o256", 500.0), ("rrt_exploring_nn_long_o256", 20000.0)):
            one, _ = pool_map(ref_exploring, [(256, 7, n_iter, "nn", max_traj)], 1)
            pone, _ = pool_map(port_exploring, [(256, 7, n_iter, "nn", max_traj)], 1)
            rec = {"iters": n_iter, "sample": "seed 7, %d iterations, max_traj_time %g" % (n_iter, max_traj),
                   "ref_expansions_per_s_1proc": n_iter / one[0]["seconds"], "ref_seconds_1proc": one[0]["seconds"],
                   "port_expansions_per_s_1proc": pone[0]["iters"] / pone[0]["seconds"],
                   "ref_nodes": one[0]["nodes"], "port_nodes": pone[0]["nodes"], "ref_cost": one[0]["cost"], "port_cost": pone[0]["cost"]}
            rec["port_over_ref"] = rec["port_expansions_per_s_1proc"] / rec["ref_expansions_per_s_1proc"]
            out[key] = rec
            print(key, rec, flush=True)
            json.dump(out, open(args.out, "w"), indent=1)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
