import numpy as np, sys
sys.path.insert(0, '/root/repo')
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
if "--clocks" in sys.argv:  # a -DAUVP_PF_DIAG build (AUVPLAN_LIBRARY=<.so>): s_memtime clocks per section and filter-step
    Fx = int(sys.argv[sys.argv.index("--clocks") + 1]) if len(sys.argv) > sys.argv.index("--clocks") + 1 else 512
    b = _pf_lib.FilterBatch(ctx, Fx, N).create(shark0[:Fx], mts[:Fx], 624)
    b.run(meas=meas[:, :Fx], shark_xy=shark[:, :Fx])
    _, err, _ = b.estimates()
    names = ["update: words", "update: objects", "weights: per-AUV", "weights: sum, classes, scan", "correct: index draws", "correct: gather", "mean"]
    tot = err[:7].mean(axis=1).sum() / S
    for k, nm in enumerate(names):
        print("%-30s %8.0f clocks per filter-step (%.0f %%)" % (nm, err[k].mean() / S, 100 * err[k].mean() / S / tot))
    print("total %.0f; launch %.2f ms" % (tot, ctx.last_kernel_ms()))
    sys.exit(0)
for name, ph in [("all", 7), ("update", 1), ("weights", 2), ("mean", 4), ("upd+w", 3)]:
    for rep in range(2):
        b = _pf_lib.FilterBatch(ctx, F, N).create(shark0, mts, 624)
        b.run(meas=meas, shark_xy=shark, phases=ph)
    print(name, "%.2f ms" % ctx.last_kernel_ms())
for Fx in (512, 1024, 2048):
    b = _pf_lib.FilterBatch(ctx, Fx, N).create(shark0[:Fx], mts[:Fx], 624)
    b.run(meas=meas[:, :Fx], shark_xy=shark[:, :Fx]); b.run(meas=meas[:, :Fx], shark_xy=shark[:, :Fx])
    print("F", Fx, "%.2f ms" % ctx.last_kernel_ms())
