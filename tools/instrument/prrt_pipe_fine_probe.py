import os, sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from auv_sim_amd import _lib, synth
from auv_sim_amd._prrt_lib import PlannerBatch
ctx = _lib.Context(0)
w = synth.make_rect_world(seed=3, n_obstacles=256)
ctx.set_world(obstacles=w["obstacles"])
names = ["H pre", "H randbelow1", "H occupied+bucket word", "H randbelow2", "H hops", "H record issue + n_total draw", "H sub-arc draws", "H record wait+store", "H publish",
         "S copy", "S theta chain", "S sincos", "S dx dy dt", "S prefix", "S points + sync", "S collision", "S publish"]
for n in (1, 512):
    starts = np.tile(np.array([w["start"][0], w["start"][1], 0.0, 0.0]), (n, 1))
    goals = np.tile(w["goal"], (n, 1))
    seeds = np.arange(n, dtype=np.uint64)
    pb = PlannerBatch(ctx, starts, goals, w["rect"], 2000, seeds=seeds, freq=10, cell=5, subs=1)
    s = pb.plan()
    print("E=%d %s %.2f ms" % (n, ctx.prrt_last_kernel(), ctx.last_kernel_ms()))
    e = 0
    assert not s[e]["done"]
    f = pb.tree(e, s[e])["points"][:8].ravel()
    st = float(s[e]["steps"])
    for i, nm in enumerate(names):
        print("   %-32s %7.0f clocks/step" % (nm, f[i] / st))
    continue
    print("   arcs evaluated per step %.2f, with samples %.2f, chunks of 64 samples per arc with samples %.2f; G eval per arc %.0f" % (f[24] / st, f[25] / st, f[26] / max(f[25], 1), f[19] / max(f[24], 1)))
