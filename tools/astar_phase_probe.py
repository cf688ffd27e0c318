"""Phase clocks of astar_kernel (config 3 inputs) from an instrumented build: AUVPLAN_LIBRARY=<.so built with the PH()
stamps> python tools/astar_phase_probe.py -- the instrumented build reuses summary fields for the five phase totals."""
import sys
import numpy as np
sys.path.insert(0, '.')
import bench
from auv_sim_amd import _lib, _astar_lib
ctx = _lib.Context(0)
w, starts, limits = bench.astar_inputs(1024)
ctx.set_world(w["obstacles"], w["habitats"], w["polygon"], w["bins"], w["cells"], w["prob"])
for _ in range(2):
    r = _astar_lib.run_batch_arrays(ctx, "astar_fixLenSOG", starts, limits=limits, weights=(0, 10, 10, 100), velocity=1.0, cap_nodes=20000)
s = r["summ"]
i = int(np.argmax(s["n_expansions"]))
ph = np.array([s["open_scanned"][i] & 0xffffffff, s["open_scanned"][i] >> 32, s["visited_count"][i], s["n_hab_left"][i], s["smooth_len"][i]], dtype=np.float64) * 16
print("launch ms", r["batch_ms"], "critical instance", i, "expansions", s["n_expansions"][i], "children", s["n_children"][i])
names = ["pop (scan + reduce)", "popped node load", "bounds (polygon fan)", "collision + child mask", "children (cell, tables, stores, list)"]
for n, v in zip(names, ph):
    print("%-40s %8.0f ticks/expansion  %5.1f %%" % (n, v / s["n_expansions"][i], 100 * v / ph.sum()))
print("total ticks/expansion", ph.sum() / s["n_expansions"][i], "(100 MHz s_memtime ticks => us x 100)")
