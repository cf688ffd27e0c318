"""Diagnostic: astar_fixLenSOG throughput against the number of instances per launch (config 3 uses 1024)."""
import sys, numpy as np
sys.path.insert(0, '.')
from auv_sim_amd import _lib, _astar_lib, synth
ctx = _lib.Context(0)
w = synth.make_world(seed=12, n_obstacles=64, obst_radius=(2.0, 6.0), n_habitats=10, hab_radius=(10.0, 25.0))
ctx.set_world(w["obstacles"], w["habitats"], w["polygon"], w["bins"], w["cells"], w["prob"])
for n_inst in (1024, 4096, 8192, 16384):
    rng = np.random.default_rng(3)
    starts = np.column_stack([-290.0 + 10.0 * rng.integers(0, 19, n_inst), -90.0 + 10.0 * rng.integers(0, 19, n_inst)])
    limits = rng.choice([100.0, 200.0, 300.0], n_inst)
    kw = dict(limits=limits, weights=(0, 10, 10, 100), velocity=1.0, cap_nodes=20000)
    _astar_lib.run_batch(ctx, "astar_fixLenSOG", starts, **kw)
    res = _astar_lib.run_batch(ctx, "astar_fixLenSOG", starts, **kw)
    ms = ctx.last_kernel_ms()
    cells = sum(r["n_children"] for r in res)
    print(n_inst, "%.2f ms" % ms, "%.0f Mcells/s" % (cells / ms / 1e3), "launch", ctx.last_launch())
