"""Timing of the expansion + leaf launches on the dense side-measurement worlds (bench.py rrt_dense) and the headline world."""
import sys
import numpy as np
sys.path.insert(0, '.')
from auv_sim_amd import _lib, synth
ctx = _lib.Context(0)
E = 12288
def run(name, w):
    ctx.set_world(w["obstacles"], w["habitats"], w["polygon"], w["bins"], w["cells"], w["prob"])
    init = np.zeros((E, 6)); init[:, 0], init[:, 1] = w["start"]
    ctx.rrt_prepare(init, np.arange(E, dtype=np.uint64), 10000, mode="timebin", freq=30, bin_interval=5, v=2, max_traj_time=500.0, weights=(-3, -3, -4))
    ctx.rrt_run(); ctx.rrt_run()
    s = ctx.summaries()
    print(name, "parts", ctx.last_launch_parts(), "nodes", s["n_nodes"].mean(), "cand/exp", s["n_candidates"].sum() / s["iters_run"].sum(), flush=True)
run("g3 dense", synth.make_world(seed=2, n_obstacles=256))
run("catalina", synth.make_world(seed=5, n_obstacles=256, box=(0.0, 0.0, 560.0, 350.0), cell=14.0, obst_radius=(2.0, 8.0), hab_radius=(20.0, 55.0)))
if len(sys.argv) > 1:
    run("headline", synth.make_world(seed=2, n_obstacles=256, box=(-1000.0, -1000.0, 1000.0, 1000.0)))
