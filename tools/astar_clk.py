import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
from auv_sim_amd import _lib, _astar_lib, synth
import ctypes as C
ctx = _lib.Context(0)
n_inst = 1024
w = synth.make_world(seed=12, n_obstacles=64, obst_radius=(2.0, 6.0), n_habitats=10, hab_radius=(10.0, 25.0))
ctx.set_world(w["obstacles"], w["habitats"], w["polygon"], w["bins"], w["cells"], w["prob"])
rng = np.random.default_rng(3)
starts = np.column_stack([-290.0 + 10.0 * rng.integers(0, 19, n_inst), -90.0 + 10.0 * rng.integers(0, 19, n_inst)])
limits = rng.choice([100.0, 200.0, 300.0], n_inst)
L = _astar_lib._bind()
p = _astar_lib.AstarParams(); p.variant, p.cap_nodes, p.velocity = 3, 20000, 1.0
for i, v in enumerate((0, 10, 10, 100)): p.w[i] = v
for rep in range(2):
    ctx._chk(L.auvp_astar_batch(ctx.h, n_inst, _lib._p(_lib._f64(starts)), None, _lib._p(_lib._f64(limits)), C.byref(p), 0))
summ = np.zeros(n_inst, dtype=_astar_lib.ASTAR_SUMMARY_DTYPE)
ctx._chk(L.auvp_astar_summaries(ctx.h, summ.ctypes.data_as(C.c_void_p)))
print("kernel ms", ctx.last_kernel_ms())
ne = summ["n_expansions"]; nn = summ["n_nodes"]
print("expansions: mean %.0f max %d ; nodes mean %.0f max %d" % (ne.mean(), ne.max(), nn.mean(), nn.max()))
i = int(np.argmax(summ["_p0"] + summ["_p1"] + summ["smooth_len"]))
for name, col in (("pop-scan", "_p0"), ("neighbours", "smooth_len"), ("children", "_p1")):
    print("%-10s mean %8.0f Kclk  slowest-instance %8d Kclk" % (name, summ[col].mean(), summ[col][i]))
print("slowest instance: expansions %d nodes %d found %d" % (ne[i], nn[i], summ["found"][i]))
