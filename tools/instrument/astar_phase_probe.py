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
ph = np.array(list(r["expansions"][0]) + list(r["expansions"][1])[:4], dtype=np.float64)
names = ["pop: scan", "pop: reduce", "pop: list fix + record load", "bounds (polygon fan) + cellinfo request", "collision + child mask", "children: grid key",
         "children: sqrt, time bin, cellinfo wait", "children: prob / topn loads", "children: stores, list append", "-", "-", "loop overhead"]
print("critical instance", i, "expansions", n_exp, "mean expansions", np.mean([len(x["expansions"]) for x in res]))
for n, v in zip(names, ph):
    print("%-44s %8.0f clocks/expansion  %5.1f %%" % (n, v / n_exp, 100 * v / ph.sum()))
print("total clocks/expansion", ph.sum() / n_exp)
