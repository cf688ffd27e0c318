#!/usr/bin/env python3
"""rrt_rows_stream_kernel (random numbers generated ahead) against rrt_rows_kernel: identical summaries / trees on a small batch,
then the headline batch's launch times.  Run on a GPU box: python tools/stream_probe.py [episodes] [iters]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from auv_sim_amd import _lib  # noqa: E402

ctx = _lib.Context(0)
w = bench.bench_world(256, 200)
ctx.set_world(w["obstacles"], w["habitats"], w["polygon"], w["bins"], w["cells"], w["prob"])


def run(E, iters, stream, reps=1):
    init = np.zeros((E, 6))
    init[:, 0], init[:, 1] = w["start"]
    ctx.set_option("ROWS", 1)
    ctx.set_option("ROWS_STREAM", stream)
    ctx.rrt_prepare(init, np.arange(E, dtype=np.uint64), iters, mode="timebin", **bench.RRT_KW)
    ms = []
    for _ in range(reps):
        ctx.rrt_run()
        ms.append((ctx.last_kernel_ms(), ctx.last_stream_ms()) + tuple(ctx.last_launch_parts()[:2]))
    return ctx.summaries().copy(), ctx.last_rrt_kernel(), ms


E, iters = 96, 1500
a, ka, _ = run(E, iters, 0)
ta = [ctx.tree(e, a[e]) for e in (0, 5, E - 1)]
b, kb, _ = run(E, iters, 1)
tb = [ctx.tree(e, b[e]) for e in (0, 5, E - 1)]
print(ka, kb, "fallbacks", ctx.pipeline_fallbacks())
bad = [n for n in a.dtype.names if not np.array_equal(a[n], b[n])]
print("summary fields that differ:", bad)
for x, y in zip(ta, tb):
    print("tree equal:", all(np.array_equal(x[k], y[k]) for k in x))
E = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
for stream in (0, 1, 0, 1):
    s, k, ms = run(E, iters, stream, reps=3)
    print(k, "E=%d iters=%d" % (E, iters), "total / stream / expansion / leaf ms:", ["%.2f %.2f %.2f %.2f" % m for m in ms], "status", np.unique(s["status"]),
          "fallbacks", ctx.pipeline_fallbacks()[0])
