#!/usr/bin/env python3
"""Diagnostic for an instrumented planner build (six phase-clock totals in summary.arc[0..5] of the episodes that did
not finish): AUVPLAN_LIBRARY=<instrumented .so> python tools/prrt_clk.py"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auv_sim_amd import _lib, synth
from auv_sim_amd._prrt_lib import PlannerBatch
ctx = _lib.Context(0)
w = synth.make_rect_world(seed=3, n_obstacles=256)
ctx.set_world(obstacles=w["obstacles"])
n_ep, max_step = 512, 2000
starts = np.tile(np.array([w["start"][0], w["start"][1], 0.0, 0.0]), (n_ep, 1))
goals = np.tile(w["goal"], (n_ep, 1))
for _ in range(2):
    s = PlannerBatch(ctx, starts, goals, w["rect"], max_step, seeds=np.arange(n_ep, dtype=np.uint64), freq=10, cell=5, subs=1).plan()
print("kernel %.2f ms" % ctx.last_kernel_ms())
a = s["arc"][s["done"] == 0]
names = ["bucket choice (randbelow, occupied/bcount reads)", "node pick (bucket-id scan)", "steer", "collision", "insert", "goal arc"]
tot = a[:, :6].mean(axis=0).sum()
for i, n in enumerate(names):
    print("%-52s %10.0f clk/step  %5.1f %%" % (n, a[:, i].mean() / max_step, 100 * a[:, i].mean() / tot))
print("episodes not done: %d of %d; total %.0f clk/step" % (len(a), n_ep, tot / max_step))
