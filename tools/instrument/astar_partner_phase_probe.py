import sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from auv_sim_amd import _lib, _astar_lib
ctx = _lib.Context(0)
w, starts, limits = bench.astar_inputs(1024)
ctx.set_world(w["obstacles"], w["habitats"], w["polygon"], w["bins"], w["cells"], w["prob"])
for _ in range(2):
    res = _astar_lib.run_batch(ctx, "astar_fixLenSOG", starts, limits=limits, weights=(0, 10, 10, 100), velocity=1.0, cap_nodes=20000, exp_log=True)
print("launch ms", ctx.last_kernel_ms())
i = int(np.argmax([len(r["expansions"]) for r in res]))
r = res[i]
n_exp = len(r["expansions"])
ph = np.array(r["expansions"][2], dtype=np.float64)
names = ["waiting for a node", "box read + position / length / time stamp arithmetic", "round trip 1 (edge tables, time bin)", "bounds check + round trip 2 (rows / columns)",
         "ballots, key, time bin", "round trip 3 (prob / topn)", "results + release + post", "-"]
for n, v in zip(names, ph):
    print("%-56s %8.0f clocks/expansion" % (n, v / n_exp))
print("busy per expansion", ph[1:].sum() / n_exp)
