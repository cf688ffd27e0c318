#!/usr/bin/env python3
"""Latency runs with one (rrt_explore_kernel), two (rrt_duo_kernel) and three (rrt_trio_kernel) wavefronts per episode: one
episode, 64 / 256 / 1 024 episodes of the headline world, and config 2's 1 024 replicas (64 obstacles).  Run on a GPU box."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from auv_sim_amd import _lib  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import env_options  # noqa: E402  (tests/env_options.py: AUVP_<NAME> in os.environ steers live contexts -- this process only)
env_options.install()

ctx = _lib.Context(0)
iters = 10000
os.environ["AUVP_ROWS"] = "0"
for obst, E in ((256, 1), (256, 64), (256, 256), (256, 1024), (64, 1024), (256, 2048), (256, 4096)):
    world = bench.bench_world(obst, 200)
    ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    init = np.zeros((E, 6))
    init[:, 0], init[:, 1] = world["start"]
    res = []
    for duo, trio, quad in (("0", "0", "0"), ("1", "0", "0"), ("0", "1", "0"), ("0", "1", "1")):
        os.environ["AUVP_DUO"], os.environ["AUVP_TRIO"], os.environ["AUVP_QUAD"] = duo, trio, quad
        ctx.rrt_prepare(init, np.arange(E, dtype=np.uint64) + 7, iters, mode="timebin", **bench.RRT_KW)
        ms = []
        for i in range(3):
            ctx.rrt_run()
            if i:
                ms.append(ctx.last_launch_parts()[0])
        s = ctx.summaries()
        assert (s["status"] >= 0).all(), np.unique(s["status"])
        res.append((ctx.last_rrt_kernel(), float(np.mean(ms)), float(s["iters_run"].sum())))
        if os.environ.get("AUVPLAN_LIBRARY") and trio == "1":
            w = s["nn_scanned"].astype(np.uint64)
            n_it = float(s["iters_run"].sum())
            print("   diag build (trio): %.3f flushes per iteration; clocks at work per iteration: H %.0f, M %.0f, T %.0f"
                  % (float(s["n_candidates"].sum()) / n_it, 256.0 * float((w & np.uint64(0xfffff)).sum()) / n_it,
                     256.0 * float(((w >> np.uint64(20)) & np.uint64(0xfffff)).sum()) / n_it, 256.0 * float((w >> np.uint64(40)).sum()) / n_it))
    print("O=%d E=%d: " % (obst, E) + " | ".join("%s %.2f ms = %.1f M exp/s (%.2f us)" % (k.replace("rrt_", "").replace("_kernel", "").replace("<4 wavefronts>", "4"), m, n / m / 1e3, 1e3 * m / iters)
                                                  for k, m, n in res))
