#!/usr/bin/env python3
"""per-trip time of rrt_rows_stream_kernel / rrt_rows_kernel against the iteration budget (= the size of the trees the parent
reads come from): is the kernel waiting for its parents?  python tools/stream_iters_probe.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from auv_sim_amd import _lib  # noqa: E402

ctx = _lib.Context(0)
w = bench.bench_world(256, 200)
ctx.set_world(w["obstacles"], w["habitats"], w["polygon"], w["bins"], w["cells"], w["prob"])
E = 12288
init = np.zeros((E, 6))
init[:, 0], init[:, 1] = w["start"]
for stream in (1, 0):
    for iters in (250, 500, 1000, 2000, 5000, 10000):
        ctx.set_option("ROWS", 1)
        ctx.set_option("ROWS_STREAM", stream)
        ctx.rrt_prepare(init, np.arange(E, dtype=np.uint64), iters, mode="timebin", **bench.RRT_KW)
        ms = []
        for _ in range(3):
            ctx.rrt_run()
            ms.append(ctx.last_launch_parts()[0])
        print(ctx.last_rrt_kernel(), "iters %5d expansion %.2f ms = %.2f us per trip (stream launch %.2f ms)" % (iters, np.mean(ms[1:]), 1e3 * np.mean(ms[1:]) / iters, ctx.last_stream_ms()))
