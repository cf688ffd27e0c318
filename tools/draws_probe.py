#!/usr/bin/env python3
"""random() numbers RRT.exploring draws per iteration on the bench world (the figure the pre-generated stream's length is set
from: auvplan.hip, option ROWS_STREAM).  python tools/draws_probe.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from auv_sim_amd import _lib  # noqa: E402

ctx = _lib.Context(0)
w = bench.bench_world(256, 200)
ctx.set_world(w["obstacles"], w["habitats"], w["polygon"], w["bins"], w["cells"], w["prob"])
for iters in (300, 1000, 10000):
    E = 4096
    init = np.zeros((E, 6))
    init[:, 0], init[:, 1] = w["start"]
    ctx.rrt_prepare(init, np.arange(E, dtype=np.uint64), iters, mode="timebin", **bench.RRT_KW)
    ctx.rrt_run()
    s = ctx.summaries()
    d = s["n_draw32"].astype(np.float64) / 2 / iters
    print(iters, "numbers per iteration: mean %.3f min %.3f max %.3f; most in one episode %.0f" % (d.mean(), d.min(), d.max(), s["n_draw32"].max() / 2))
