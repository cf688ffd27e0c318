#!/usr/bin/env python3
"""Randomised sweep of prrt_pipe_kernel (four wavefronts per Planner_RRT episode) against prrt_kernel: random worlds,
goals near and far (so that plannings end at every stage: the take-back of a step's insert when the arc of the step before is
free depends on timing between the wavefronts), planner parameters and budgets; every summary field, trees, bucket lists and
paths bit for bit, and the planning continued by generate_one_node steps.  Every case is repeated.
With a diagnostic build (AUVPLAN_LIBRARY=auv_sim_amd/libauvplan_diag.so) AUVP_DIAG_JITTER delays one stage's hand-overs and
AUVP_DIAG_SPIN makes the bounded waits run out: the episodes redone by the pipeline fallback are counted in the last line.
usage: python tests/experiments/soak_planner_duo.py <cases> <seed>"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from auv_sim_amd import _lib, synth  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import env_options  # noqa: E402  (tests/env_options.py: AUVP_<NAME> in os.environ steers live contexts -- this process only)
env_options.install()
from auv_sim_amd._prrt_lib import PlannerBatch  # noqa: E402

n_cases, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
ctx = _lib.Context(0)
os.environ["AUVP_PRRT_ROWS"] = "0"
bad = 0
FALLBACKS = [0]
for c in range(n_cases):
    n_obst = int(rng.choice([8, 64, 128, 256]))
    # round 6: every third world is translated (the bucket grid ignores the origin: wrapped indexes, unbucketed nodes, and the
    # reference's IndexError = status -1 with the offending node stored: the two kernels must agree on all of it)
    origin = (0.0, 0.0) if rng.random() < 0.67 else (round(float(rng.uniform(-260.0, 120.0)), 2), round(float(rng.uniform(-260.0, 120.0)), 2))
    w = synth.make_rect_world(seed=int(rng.integers(1, 10_000)), n_obstacles=n_obst, origin=origin)
    ctx.set_world(obstacles=w["obstacles"])
    E = int(rng.choice([1, 2, 5, 33, 130, 600]))
    max_step = int(rng.choice([1, 2, 7, 80, 500, 2000])) if E < 600 else int(rng.choice([1, 7, 80]))
    kw = dict(freq=int(rng.choice([1, 3, 10, 15, 30])), cell=int(rng.choice([2, 5, 10])), subs=int(rng.choice([1, 2, 4, 8])))
    starts = np.tile(np.array([w["start"][0], w["start"][1], 0.0, 0.0]), (E, 1))
    starts[:, 2] = rng.uniform(-3.0, 3.0, E)
    r0, r1, r2, r3 = w["rect"]
    goals = np.column_stack([rng.uniform(r0 + 5, r2 - 5, E), rng.uniform(r1 + 5, r3 - 5, E)])
    near = rng.random(E) < 0.4  # goals a few steps away: the planning ends early, at any stage of the pipeline
    goals[near] = starts[near, :2] + rng.uniform(-25, 25, (int(near.sum()), 2))
    seeds = rng.integers(0, 2 ** 40, E).astype(np.uint64)
    ref = None
    for waves, rep in ((0, 1), (4, 4)):
        os.environ["AUVP_PRRT_PIPE"] = "1" if waves == 4 else "0"
        for r_i in range(rep):
            # (round 6: the pipeline has a fifth wavefront for the sub-arc draws where three episodes fit a workgroup; every
            # other repetition forces the four-wavefront form, which larger batches still get)
            os.environ["AUVP_PRRT_PIPE_DRAW"] = "1" if r_i % 2 == 0 else "0"
            os.environ["AUVP_PRRT_BUCKET_LDS"] = "1" if (r_i // 2) % 2 == 0 else "0"   # (the bucket table's LDS mirror on / off)
            pb = PlannerBatch(ctx, starts, goals, w["rect"], max_step, seeds=seeds, **kw)
            s = pb.plan().copy()
            want = {0: "prrt_kernel", 4: "prrt_pipe_kernel"}[waves]
            # (a batch the pipeline's limits exclude, or whose failed episodes were redone, reports prrt_kernel)
            assert ctx.prrt_last_kernel() in (want, "prrt_kernel"), ctx.prrt_last_kernel()
            FALLBACKS[0] += ctx.pipeline_fallbacks()[0]
            sample = sorted(set(rng.integers(0, E, 4).tolist())) if ref is None else ref[3]
            trees = [pb.tree(e, s[e]) for e in sample]
            grids = [pb.grid(e) for e in sample]
            paths = pb.paths(s)
            nxt = np.array([int(pb.grid(e)[0][0]) if len(pb.grid(e)[0]) else 0 for e in range(min(E, 8))] + [0] * max(0, E - 8), dtype=np.int32)
            cont = [pb.step(nxt).copy() for _ in range(2)][-1]
            if ref is None:
                ref = (s, trees, grids, sample, paths, cont)
                if (s["status"] < 0).any():
                    print("case %d: the one-wavefront kernel reports status %s" % (c, np.unique(s["status"])))
                continue
            diff = [n for n in s.dtype.names if not np.array_equal(ref[0][n], s[n])]
            diff += ["cont." + n for n in cont.dtype.names if not np.array_equal(ref[5][n], cont[n])]
            for i, e in enumerate(sample):
                diff += ["tree%d.%s" % (e, k) for k in trees[i] if not np.array_equal(ref[1][i][k], trees[i][k])]
                if not (np.array_equal(ref[2][i][0], grids[i][0]) and np.array_equal(ref[2][i][1], grids[i][1])):
                    diff.append("grid%d" % e)
            diff += ["path%d" % e for e in range(E) if not np.array_equal(ref[4][e], paths[e])][:3]
            if diff:
                bad += 1
                print("MISMATCH case %d waves %d E=%d max_step=%d obst=%d %s origin %s: %s" % (c, waves, E, max_step, n_obst, kw, origin, diff[:8]))
    if c % 10 == 9:
        print("  %d cases, %d mismatches (last: E=%d max_step=%d done %d)" % (c + 1, bad, E, max_step, int(ref[0]["done"].sum())), flush=True)
print("planner soak: %d cases, %d mismatches, %d episodes redone by the pipeline fallback" % (n_cases, bad, FALLBACKS[0]))
sys.exit(1 if bad else 0)
