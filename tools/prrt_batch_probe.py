#!/usr/bin/env python3
"""Planner_RRT.planning (config 4's world and parameters) against the batch size for the three kernels -- prrt_pipe_kernel (four /
five wavefronts per episode), prrt_kernel (one), prrt_rows_kernel (four episodes per wavefront) -- and the host's choice
(planner_rrt_host.h: pipe up to 4 episodes per CU, one wavefront up to 8, rows above).  python tools/prrt_batch_probe.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auv_sim_amd import _lib, synth  # noqa: E402
from auv_sim_amd._prrt_lib import PlannerBatch  # noqa: E402

ctx = _lib.Context(0)
w = synth.make_rect_world(seed=3, n_obstacles=256)
ctx.set_world(obstacles=w["obstacles"])
CHOICES = (("host", {}), ("pipe", dict(PRRT_PIPE=1, PRRT_ROWS=0)), ("one", dict(PRRT_PIPE=0, PRRT_ROWS=0, PRRT_LAT=0)),
           ("one-lat", dict(PRRT_PIPE=0, PRRT_ROWS=0, PRRT_LAT=1)), ("rows", dict(PRRT_PIPE=0, PRRT_ROWS=1)))
print("episodes   " + "   ".join(n for n, _ in CHOICES) + "      M steps/s (plan launch ms, kernel)")
for E in (256, 512, 1024, 1536, 2048, 3072, 4096, 8192):
    starts = np.tile(np.array([w["start"][0], w["start"][1], 0.0, 0.0]), (E, 1))
    goals = np.tile(w["goal"], (E, 1))
    seeds = np.arange(E, dtype=np.uint64)
    out = []
    for name, opts in CHOICES:
        for k in ("PRRT_PIPE", "PRRT_ROWS", "PRRT_LAT"):
            ctx.set_option(k, opts.get(k))
        ms, steps = [], 0
        try:
            for i in range(3):
                pb = PlannerBatch(ctx, starts, goals, w["rect"], 2000, seeds=seeds, freq=10, cell=5, subs=1)
                s = pb.plan()
                if i:
                    ms.append(ctx.last_kernel_ms())
                steps = int(s["steps"].sum())
            out.append("%5.0f (%.2f, %s)" % (steps / (np.mean(ms) * 1e-3) / 1e6, np.mean(ms), ctx.prrt_last_kernel()[:16]))
        except Exception as e:  # (a forced kernel outside its limits)
            out.append("  --  (%s)" % str(e)[:30])
    print("%8d   " % E + "   ".join(out))
