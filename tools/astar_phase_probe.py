"""Phase clocks of astar_kernel on the config-3 inputs, from an instrumented build (PH() stamps patched into a copy of
astar_kernel.h; the eight totals of an instance land in row 0 of its expansion log):
AUVPLAN_LIBRARY=<instrumented .so> python tools/astar_phase_probe.py"""
import sys
import numpy as np
sys.path.insert(0, '.')
import bench
from auv_sim_amd import _lib, _astar_lib
ctx = _lib.Context(0)
w, starts, limits = bench.astar_inputs(1024)
ctx.set_world(w["obstacles"], w["habitats"], w["polygon"], w["bins"], w["cells"], w["prob"])
variant = sys.argv[1] if len(sys.argv) > 1 else "astar_fixLenSOG"
wts = (0, 10, 10, 100) if variant == "astar_fixLenSOG" else (0, 10, 10)
for _ in range(2):
    res = _astar_lib.run_batch(ctx, variant, starts, limits=limits, weights=wts, velocity=1.0, cap_nodes=20000, exp_log=True)
print(variant, "launch ms", ctx.last_kernel_ms())
ms = ctx.last_kernel_ms() if hasattr(ctx, "last_kernel_ms") else float("nan")
i = int(np.argmax([r["n_expansions"] if "n_expansions" in r else len(r["expansions"]) for r in res]))
r = res[i]
n_exp = len(r["expansions"])
ph = np.array(r["expansions"][0], dtype=np.float64)
names = ["pop (scan + reduce)", "popped node load", "bounds (polygon fan)", "collision + child mask", "children: stores, list append",
         "children: grid key", "children: sqrt, time bin, cellinfo", "children: prob/topn loads"]
print("critical instance", i, "expansions", n_exp)
for n, v in zip(names, ph):
    print("%-40s %8.0f ticks/expansion  %5.1f %%" % (n, v / n_exp, 100 * v / ph.sum()))
print("total ticks/expansion", ph.sum() / n_exp)
