"""Section clocks of rrt_trio_kernel's H wavefront (library built with tools/instrument/rrt_trio_h_patch.py and -DAUVP_DUO_DIAG)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from auv_sim_amd import _lib  # noqa: E402

ctx = _lib.Context(0)
iters = 10000
os.environ["AUVP_ROWS"], os.environ["AUVP_DUO"], os.environ["AUVP_TRIO"], os.environ["AUVP_QUAD"] = "0", "0", "1", "0"
names = ["packet start (position record, acquire)", "generation (ensure 128 words)", "window: 60 bin candidates, first hit", "node pick, sub-arc count",
         "window to LDS (+ beyond 64)", "taken predicate over the window", "parent record request + fixed point", "draw pick-up, packet arrays, advance", "packet scalars + release + post"]
for obst, E in ((256, 1), (64, 1024)):
    world = bench.bench_world(obst, 200)
    ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    init = np.zeros((E, 6))
    init[:, 0], init[:, 1] = world["start"]
    ctx.rrt_prepare(init, np.arange(E, dtype=np.uint64) + 7, iters, mode="timebin", **bench.RRT_KW)
    ctx.rrt_run(); ctx.rrt_run()
    s = ctx.summaries()
    print("O=%d E=%d %s %.2f ms" % (obst, E, ctx.last_rrt_kernel(), ctx.last_launch_parts()[0]))
    nd = ctx.tree(0, s[0])["nodes"]
    vals = [nd[0][0], nd[0][1], nd[0][2], nd[0][3], nd[0][5], nd[1][0], nd[1][1], nd[1][2], nd[1][3], nd[1][5]]
    n_it = float(s[0]["iters_run"])
    for i, nm in enumerate(names):
        print("   %-48s %7.0f clocks/iteration" % (nm, vals[i] / n_it))
    print("   sum %.0f" % (sum(vals[:9]) / n_it))
