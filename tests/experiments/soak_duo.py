#!/usr/bin/env python3
"""Randomised sweep of the helper-wavefront kernels (rrt_duo_kernel, rrt_trio_kernel) against rrt_explore_kernel: random
worlds, planner parameters, batch sizes and budgets, every summary field and a sample of trees bit for bit.  The speculative
stages (redo / new epoch, the start-of-episode phase where most bins are empty) depend on timing between wavefronts, so the
sweep repeats every case.  With a diagnostic build (AUVPLAN_LIBRARY=auv_sim_amd/libauvplan_diag.so) AUVP_DIAG_JITTER delays
one stage's hand-overs and AUVP_DIAG_SPIN makes the bounded waits run out: the episodes the pipeline fallback redid are counted
in the last line.  usage: python tests/experiments/soak_duo.py <cases> <seed>"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from auv_sim_amd import _lib, synth  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import env_options  # noqa: E402  (tests/env_options.py: AUVP_<NAME> in os.environ steers live contexts -- this process only)
env_options.install()

n_cases, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
ctx = _lib.Context(0)
os.environ["AUVP_ROWS"] = "0"
bad = 0
fallbacks = 0
for c in range(n_cases):
    n_obst = int(rng.choice([8, 64, 128, 256]))
    # (round 6: every fourth world has the reference's 5-vertex Catalina outline or a concave one instead of the rectangle)
    world = synth.make_world(seed=int(rng.integers(1, 10_000)), n_obstacles=n_obst,
                             polygon=[None, None, None, None, None, None, "catalina", "notch"][int(rng.integers(0, 8))])
    ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    E = int(rng.choice([1, 2, 3, 7, 33, 130]))
    n_iter = int(rng.choice([1, 5, 60, 400, 1500]))
    kw = dict(freq=int(rng.choice([1, 4, 10, 30])), bin_interval=float(rng.choice([2.5, 5.0, 20.0])),
              max_traj_time=float(rng.choice([40.0, 200.0, 500.0])))
    init = np.zeros((E, 6))
    init[:, 0], init[:, 1] = world["start"]
    init[:, 2] = rng.uniform(-3, 3, E)
    seeds = rng.integers(0, 2 ** 40, E).astype(np.uint64)
    ref = None
    for kern, rep in (("explore", 1), ("duo", 2), ("trio", 3), ("quad", 3)):
        os.environ["AUVP_DUO"] = "1" if kern == "duo" else "0"
        os.environ["AUVP_TRIO"] = "1" if kern in ("trio", "quad") else "0"
        os.environ["AUVP_QUAD"] = "1" if kern == "quad" else "0"
        for _ in range(rep):
            s = ctx.rrt_explore_batch(init, seeds, n_iter, **kw).copy()
            redone = ctx.pipeline_fallbacks()[0]
            fallbacks += redone
            want = "rrt_explore_kernel" if redone else "rrt_%s_kernel" % ("trio" if kern == "quad" else kern)
            assert ctx.last_rrt_kernel().startswith(want), (ctx.last_rrt_kernel(), want)
            trees = [ctx.tree(e, s[e]) for e in range(min(E, 3))]
            if ref is None:
                ref = (s, trees)
                continue
            ok = all(np.array_equal(s[f], ref[0][f]) for f in s.dtype.names)
            ok = ok and all(np.array_equal(t[k], r[k]) for t, r in zip(trees, ref[1]) for k in ("nodes", "parent", "pt_off", "pt_cnt", "points"))
            if not ok:
                bad += 1
                print("MISMATCH case %d %s: O=%d E=%d iters=%d %s" % (c, kern, n_obst, E, n_iter, kw))
print("%d cases, %d mismatches, %d episodes redone by the pipeline fallback" % (n_cases, bad, fallbacks))
sys.exit(1 if bad else 0)
