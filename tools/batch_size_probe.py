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
print("episodes  one-episode M/s  rows M/s   (kernel ms)")
for E in (1024, 2048, 3072, 4096, 6144, 8192, 10240, 12288):
    init = np.zeros((E, 6))
    init[:, 0], init[:, 1] = world["start"]
    out = []
    for rows in ("0", "1"):
        os.environ["AUVP_ROWS"] = rows
        ctx.rrt_prepare(init, np.arange(E, dtype=np.uint64), iters, mode="timebin", **bench.RRT_KW)
        ms = []
        for i in range(3):
            ctx.rrt_run()
            if i:
                ms.append(ctx.last_kernel_ms())
        out.append((E * iters / (np.mean(ms) * 1e-3) / 1e6, np.mean(ms)))
    os.environ.pop("AUVP_ROWS")
    print("%8d  %10.0f  %10.0f   (%.1f / %.1f)" % (E, out[0][0], out[1][0], out[0][1], out[1][1]))
