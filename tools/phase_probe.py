#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock breakdown of the RRT.exploring kernel (AUVP_FLAG_PHASE_CLOCKS)."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auv_sim_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--episodes", type=int, default=1024)
ap.add_argument("--iters", type=int, default=2000)
ap.add_argument("--obstacles", type=int, default=256)
ap.add_argument("--grid", type=int, default=200)
ap.add_argument("--mode", default="timebin")
ap.add_argument("--clocks", type=int, default=1)
ap.add_argument("--lib", default=None, help="alternative libauvplan.so (kernel experiments)")
a = ap.parse_args()
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)
half = 0.5 * a.grid * 10.0
world = synth.make_world(seed=2, n_obstacles=a.obstacles, box=(-half, -half, half, half), cell=10.0)
ctx = _lib.Context(0)
ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
E = a.episodes
init = np.zeros((E, 6)); init[:, 0], init[:, 1] = world["start"]
ctx.rrt_prepare(init, np.arange(E, dtype=np.uint64), a.iters, mode=a.mode, phase_clocks=bool(a.clocks))
ctx.rrt_run(); ctx.rrt_run()
ms = ctx.last_kernel_ms()
s = ctx.summaries()
print("E=%d iters=%d kernel %.2f ms -> %.3f Mexp/s; per-wave us/iter %.2f" % (E, a.iters, ms, E * a.iters / ms / 1e3, ms * 1e3 / a.iters))
print("nodes/ep %.0f leaves/ep %.0f leaf_elems/leaf %.0f status %s" % (s["n_nodes"].mean(), s["n_leaves"].mean(), s["leaf_elems"].sum() / max(1, s["n_leaves"].sum()), np.unique(s["status"])))
if a.clocks:
    pc = ctx.phase_clocks().astype(np.float64)
    names = ["select", "steer", "collision", "accept", "cost"]
    tot = pc.sum(axis=1).mean()
    for i, n in enumerate(names):
        print("  %-10s %12.0f clk/episode  %6.1f%%  %8.1f clk/iter" % (n, pc[:, i].mean(), 100 * pc[:, i].mean() / tot, pc[:, i].mean() / a.iters))
    print("  total %.0f clk/iter" % (tot / a.iters))
