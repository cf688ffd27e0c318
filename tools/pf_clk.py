#!/usr/bin/env python3
"""Diagnostic for an instrumented particle-filter build (sub-phase clocks in err[0..5])."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auv_sim_amd import _lib, _pf_lib
ctx = _lib.Context(0)
rng = np.random.default_rng(4)
F, N, S, A = 4096, 1000, 20, 2
shark0 = rng.uniform(-500, 500, size=(F, 2))
meas = np.zeros((S, F, A, 5))
meas[..., 0:2] = shark0[None, :, None, :] + rng.uniform(-150, 150, size=(S, F, A, 2))
meas[..., 2] = rng.uniform(-np.pi, np.pi, size=(S, F, A))
meas[..., 3] = rng.uniform(0, 200, size=(S, F, A))
meas[..., 4] = rng.uniform(-np.pi, np.pi, size=(S, F, A))
shark = shark0[None] + rng.uniform(-20, 20, size=(S, F, 2))
key0, _ = _pf_lib.np_seed_state(0)
mts = np.stack([np.roll(key0, f) ^ np.uint32(f) for f in range(F)])
b = _pf_lib.FilterBatch(ctx, F, N).create(shark0, mts, 624).run(meas=meas, shark_xy=shark)
print("kernel %.2f ms" % ctx.last_kernel_ms())
_, err, _ = b.estimates()
for i, n in enumerate(["update", "weights+max", "normalize+scan", "choice", "gather", "mean"]):
    print("%-16s %9.0f clk/step" % (n, err[i].mean() / S))
