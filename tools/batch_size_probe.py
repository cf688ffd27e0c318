#!/usr/bin/env python3
"""RRT.exploring expansions/s against the batch size for the two expansion kernels (AUVP_ROWS=0: one episode per wavefront,
AUVP_ROWS=1: four), on the headline world: where the host's choice between them (auvplan.hip: rows above 24 episodes per CU)
sits.  Run on a GPU box: python tools/batch_size_probe.py [iters]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from auv_sim_amd import _lib  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import env_options  # noqa: E402  (tests/env_options.py: AUVP_<NAME> in os.environ steers live contexts -- this process only)
env_options.install()

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
world = bench.bench_world(256, 200)
ctx = _lib.Context(0)
ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
print("episodes   host's choice   one-episode   rows (generator inside)   rows (numbers generated ahead)      M expansions/s (kernel ms)")
CHOICES = (("default", {}), ("one", dict(ROWS=0, DUO=0, TRIO=0)), ("rows", dict(ROWS=1, DUO=0, TRIO=0, ROWS_STREAM=0)),
           ("stream", dict(ROWS=1, DUO=0, TRIO=0, ROWS_STREAM=1)))
for E in (1024, 2048, 3072, 4096, 5120, 6144, 8192, 10240, 12288):
    init = np.zeros((E, 6))
    init[:, 0], init[:, 1] = world["start"]
    out = []
    for name, opts in CHOICES:
        for k in ("ROWS", "DUO", "TRIO", "ROWS_STREAM"):
            ctx.set_option(k, opts.get(k))
        ctx.rrt_prepare(init, np.arange(E, dtype=np.uint64), iters, mode="timebin", **bench.RRT_KW)
        ms = []
        for i in range(4):
            ctx.rrt_run()
            if i > 1:
                ms.append(ctx.last_kernel_ms())
        out.append((E * iters / (np.mean(ms) * 1e-3) / 1e6, np.mean(ms), ctx.last_rrt_kernel()))
    print("%8d  " % E + "   ".join("%6.0f (%.1f, %s)" % o for o in out))
