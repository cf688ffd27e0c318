#!/usr/bin/env python3
"""Break-even measurement for packing the four rows' sub-arcs of rrt_rows_kernel into one 64-lane steer pass
(profiles/r4_rows_packing.md).  Run on a GPU box:

  python tools/rows_pass_probe.py                      # product library: freq 30 / 15 / 8 (passes per trip 1.92 / 1.0 / 1.0)
  AUVPLAN_LIBRARY=build_tmp/libauvplan_pad128.so python tools/rows_pass_probe.py 30   # an experiment build with
        AUVP_ROWS_PAD extra vector instructions per steer pass (hipcc ... -DAUVP_ROWS_PAD=128; results unchanged)

Prints one JSON line per freq: rrt_rows_kernel ms on the headline batch (12 288 episodes x 10 000 iterations), accepted
nodes and path points per episode (the trees are the same for every padding)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from auv_sim_amd import _lib  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import env_options  # noqa: E402  (tests/env_options.py: AUVP_<NAME> in os.environ steers live contexts -- this process only)
env_options.install()

freqs = [int(a) for a in sys.argv[1:]] or [30, 15, 8]
E, iters = 12288, 10000
world = bench.bench_world(256, 200)
ctx = _lib.Context(0)
ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
init = np.zeros((E, 6))
init[:, 0], init[:, 1] = world["start"]
for freq in freqs:
    kw = dict(bench.RRT_KW, freq=freq)
    ctx.rrt_prepare(init, np.arange(E, dtype=np.uint64), iters, mode="timebin", **kw)
    ms = []
    for i in range(3):
        ctx.rrt_run()
        if i:
            ms.append(ctx.last_launch_parts()[0])
    s = ctx.summaries()
    assert ctx.last_launch_parts()[2] == 4 and (s["status"] >= 0).all()
    # steer passes per trip of four rows: pass 0 runs if any row has n > 0, pass 1 if any row has n > 15 (n = floor(U(0, freq)))
    p0 = 1.0 - (1.0 / freq) ** 4
    p1 = 1.0 - (min(16, freq) / float(freq)) ** 4
    print(json.dumps({"library": os.path.basename(os.environ.get("AUVPLAN_LIBRARY", "libauvplan.so")), "freq": freq,
                      "rows_kernel_ms": float(np.mean(ms)), "passes_per_trip": p0 + p1,
                      "accepted_per_episode": float((s["n_nodes"] - 1).mean()), "points_per_episode": float(s["n_points"].mean()),
                      "draws32_per_episode": float(s["n_draw32"].mean())}))
