"""bench_sides/astar.py -- the A* variants (BASELINE config 3 + config 1)  (split out of bench.py in round 6)"""
import json
import os
import sys
import time

import numpy as np

from .common import *  # noqa: F401,F403
from .common import _rrt_batch  # noqa: F401


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
