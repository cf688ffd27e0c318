import sys, numpy as np
sys.path.insert(0, '.')
from auv_sim_amd import _lib, synth
ctx = _lib.Context(0)
E = 12288
w = synth.make_world(seed=2, n_obstacles=256, box=(-1000.0, -1000.0, 1000.0, 1000.0))
ctx.set_world(w["obstacles"], w["habitats"], w["polygon"], w["bins"], w["cells"], w["prob"])
init = np.zeros((E, 6)); init[:, 0], init[:, 1] = w["start"]
ctx.rrt_prepare(init, np.arange(E, dtype=np.uint64), 10000, mode="timebin", freq=30, bin_interval=5, v=2, max_traj_time=500.0, weights=(-3, -3, -4))
ctx.rrt_run(); ctx.rrt_run()
s = ctx.summaries()
print("parts", ctx.last_launch_parts())
ph = np.array([s["rng_after"], s["best_cost"][:, 1], s["best_cost"][:, 2], s["best_cost"][:, 3], s["best_length"], s["leaf_elems"].astype(float), s["n_draw32"].astype(float)]).mean(axis=1)
names = ["0 mark (backward sweep)", "1 queue fill", "2 records, prefix, owner table", "3 point loop", "4 node term", "5 parent sums + rounds + stores", "6 ranking + re-summation"]
for n, v in zip(names, ph): print("%-36s %10.0f ticks/episode %5.1f %%" % (n, v, 100 * v / ph.sum()))
