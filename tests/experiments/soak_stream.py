#!/usr/bin/env python3
"""Randomised sweep of the pre-generated random stream (rrt_stream_kernel + rrt_rows_stream_kernel) against rrt_rows_kernel and
the one-episode kernel: random worlds (rectangles, the Catalina outline, a concave one), parameters, batch
sizes, budgets, fresh seeds and continued generators at every alignment, and the stream's length drawn around what the batch
draws -- whether the stream kernel finishes or the batch is redone, every summary field and a sample of trees bit for bit.
Every fifth case leaves the choice to the host (no option: second batch on the world).
usage: python tests/experiments/soak_stream.py <cases> <seed>"""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from auv_sim_amd import _lib, synth  # noqa: E402

n_cases, seed = int(sys.argv[1]), int(sys.argv[2])
rnd = random.Random(seed)
ctx = _lib.Context(0)
bad = n_stream = n_redone = n_auto = 0


def run(init, seeds, n_iter, kw, rows, stream, cap=None):
    ctx.set_option("ROWS", rows)
    ctx.set_option("DUO", 0)      # (small batches: not the latency kernels)
    ctx.set_option("TRIO", 0)
    ctx.set_option("ROWS_STREAM", stream)
    ctx.set_option("ROWS_STREAM_CAP", cap)
    s = ctx.rrt_explore_batch(init, seeds, n_iter, **kw).copy()
    E = len(s)
    trees = [ctx.tree(e, s[e]) for e in sorted({0, E // 2, E - 1})]
    return s, trees, ctx.last_rrt_kernel(), ctx.pipeline_fallbacks()[0]


for c in range(n_cases):
    world = synth.make_world(seed=rnd.randrange(10_000), n_obstacles=rnd.choice([0, 8, 64, 200, 256]), n_bins=rnd.choice([4, 10]),
                             polygon=rnd.choice([None, None, "catalina", "notch"]))
    ctx.set_world(world["obstacles"], world["habitats"], world["polygon"], world["bins"], world["cells"], world["prob"])
    E, n_iter = rnd.choice([1, 3, 17, 48, 49, 130, 400]), rnd.choice([16, 60, 400, 1000, 1500, 3000])
    kw = dict(freq=rnd.choice([1, 4, 10, 30]), bin_interval=rnd.choice([2.5, 5.0, 20.0]), max_traj_time=rnd.choice([40.0, 200.0, 500.0]))
    init = np.zeros((E, 6))
    init[:, 0], init[:, 1] = world["start"]
    init[:, 2] = [rnd.uniform(-3, 3) for _ in range(E)]
    if c % 3 == 0:
        words, idx = [], []
        for e in range(E):
            g = random.Random(rnd.randrange(1 << 30))
            for _ in range(rnd.randrange(700)):
                g.random()
            if rnd.random() < 0.5:
                g.getrandbits(32)
            st = g.getstate()[1]
            words.append(st[:624])
            idx.append(st[624])
        seeds = (np.array(words, dtype=np.uint32), np.array(idx, dtype=np.int32))
    else:
        seeds = np.array([rnd.randrange(1 << 40) for _ in range(E)], dtype=np.uint64)
    ref = run(init, seeds, n_iter, kw, 1, 0)
    one = run(init, seeds, n_iter, kw, 0, 0) if c % 4 == 0 else None
    drawn = int(ref[0]["n_draw32"].max() + 1) // 2
    if c % 5 == 4:
        got = run(init, seeds, n_iter, kw, 1, None)      # the host's choice: the batch above was the first on this world
        n_auto += got[2] == "rrt_rows_stream_kernel"
        # (where the four-episode kernel is not eligible -- K too large for its LDS plan -- the batch above ran the one-episode kernel)
        if ref[2] == "rrt_rows_kernel" and (got[2] == "rrt_rows_stream_kernel") != (n_iter >= 1000):
            bad += 1
            print("case", c, "host's choice:", got[2], "at", n_iter, "iterations")
    else:
        cap = max(64, int(drawn * rnd.choice([0.05, 0.5, 0.98, 1.0, 1.0, 1.02, 2.0]))) if c % 2 else None
        got = run(init, seeds, n_iter, kw, 1, 1, cap)
    n_redone += got[3] > 0
    n_stream += got[2] == "rrt_rows_stream_kernel"
    for other, label in ((got, "stream"), (one, "one-episode")):
        if other is None:
            continue
        failed = ref[0]["status"] < 0
        for f in ref[0].dtype.names:
            if f == "n_candidates" and label == "one-episode":
                continue
            keep = ~failed if (f in ("rng_after", "n_draw32") and label == "one-episode") else np.ones(E, bool)
            if not np.array_equal(ref[0][f][keep], other[0][f][keep]):
                bad += 1
                print("case", c, label, "field", f, "differs")
        for a, b in zip(ref[1], other[1]):
            if not all(np.array_equal(a[k], b[k]) for k in a):
                bad += 1
                print("case", c, label, "tree differs")
for k in ("ROWS", "DUO", "TRIO", "ROWS_STREAM", "ROWS_STREAM_CAP"):
    ctx.set_option(k, None)
print("%d cases: %d finished on the stream kernel (%d by the host's choice), %d redone with the generator inside; %d mismatches" % (n_cases, n_stream, n_auto, n_redone, bad))
