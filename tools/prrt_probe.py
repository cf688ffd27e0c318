#!/usr/bin/env python3
"""Diagnostic: Planner_RRT config-4 timing (512 episodes x 2000 steps) for kernel experiments
(AUVPLAN_LIBRARY=<.so> swaps the library)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from auv_sim_amd import _lib
ctx = _lib.Context(0)
r = bench.bench_planner(ctx, False, reps=2)
print("config4 kernel %.2f ms, steps %d, done %d" % (r["kernel_ms"], r["steps_per_launch"], r["episodes_done"]))
r = bench.bench_config5(ctx, reps=1)
print("config5 kernel %.2f ms, steps %d" % (r["kernel_ms"], r["steps_per_launch"]))
