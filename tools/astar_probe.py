#!/usr/bin/env python3
"""Diagnostic: A* config-3 timing for kernel experiments (AUVPLAN_LIBRARY=<.so> swaps the library)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from auv_sim_amd import _lib
r = bench.bench_astar(_lib.Context(0), False, reps=3)
print("SOG kernel %.3f ms cells %d found %d | fixLen %.3f ms | astar %.3f ms" % (
    r["kernel_ms"], r["cells_per_launch"], r["found"], r["variants"]["astar_fixLen"]["kernel_ms"], r["variants"]["astar"]["kernel_ms"]))
