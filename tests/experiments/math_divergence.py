#!/usr/bin/env python3
"""CPU-only experiment (test infrastructure): how often does the portable fp64 math of the kernels
(auv_sim_amd/csrc/auvp_math.h: sin/cos within 1 ulp of glibc, ~90 % bit-equal) change a DECISION of RRT.exploring?

The HIP kernels are bit-identical to the CPU checker built with that math (liboracle_portable.so); the reference
computes with glibc (liboracle_libm.so reproduces its goldens bit for bit).  One flipped `movement >= 0.5` or
`d2 <= T` would fork the whole tree, so this runs both builds over many seeds on the bench world
(256 obstacles, 200x200 cells, 10 000 iterations, time-bin sampling) and compares every decision:
accepted flags, parents, node count, draw count (rng_after), qualifying leaves, best leaf.

usage: python tests/experiments/math_divergence.py [--seeds 2048] [--iters 10000] [--procs 6] [--out FILE]
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def _world(obstacles):
    from auv_sim_amd import synth
    return synth.make_world(seed=2, n_obstacles=obstacles, box=(-1000.0, -1000.0, 1000.0, 1000.0), cell=10.0, n_bins=10,
                            bin_len=50, n_habitats=10)


_W = {}


def _episode(job):
    seed, iters, obstacles = job
    from oracle import orc
    if obstacles not in _W:
        w = _world(obstacles)
        _W[obstacles] = (w, orc.WorldArrays(w["obstacles"], w["habitats"], w["polygon"], w["bins"], w["cells"], w["prob"]))
    world, wa = _W[obstacles]
    init = [world["start"][0], world["start"][1], 0, 0, 0, 0]
    a = orc.rrt_explore(wa, seed, iters, init=init, kind="libm", want_path=False)
    b = orc.rrt_explore(wa, seed, iters, init=init, kind="portable", want_path=False)
    same = (a["n_nodes"] == b["n_nodes"] and a["rng_after"] == b["rng_after"] and a["best_leaf"] == b["best_leaf"]
            and a["n_leaves"] == b["n_leaves"] and np.array_equal(a["parent"], b["parent"])
            and np.array_equal(a["it_accepted"], b["it_accepted"]) and np.array_equal(a["it_parent"], b["it_parent"])
            and np.array_equal(a["leaf_iter"], b["leaf_iter"]))
    first = -1
    if not same:
        n = min(len(a["it_accepted"]), len(b["it_accepted"]))
        d = np.nonzero((a["it_accepted"][:n] != b["it_accepted"][:n]) | (a["it_parent"][:n] != b["it_parent"][:n]))[0]
        first = int(d[0]) if len(d) else n
    k = min(a["n_nodes"], b["n_nodes"])
    node_diff = float(np.abs(a["nodes"][:k] - b["nodes"][:k]).max()) if same else float("nan")
    bit_equal = bool(np.array_equal(a["nodes"], b["nodes"])) if same else False
    cost_diff = float(np.abs(a["best_cost"] - b["best_cost"]).max()) if same and a["status"] == 0 else float("nan")
    return seed, same, first, node_diff, cost_diff, bit_equal, a["n_nodes"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=2048)
    ap.add_argument("--iters", type=int, default=10000)
    ap.add_argument("--obstacles", type=int, default=256)
    ap.add_argument("--procs", type=int, default=6)
    ap.add_argument("--out", default=os.path.join(REPO, "profiles", "r2_math_divergence.json"))
    args = ap.parse_args()
    from oracle import orc
    orc.build()
    t0 = time.time()
    with mp.get_context("fork").Pool(args.procs) as pool:
        res = pool.map(_episode, [(s, args.iters, args.obstacles) for s in range(args.seeds)], chunksize=4)
    div = [r for r in res if not r[1]]
    nd = np.array([r[3] for r in res if r[1]])
    cd = np.array([r[4] for r in res if r[1] and r[4] == r[4]])
    out = {
        "what": "liboracle_libm (glibc math = the reference) vs liboracle_portable (auvp_math.h = the HIP kernels), "
                "RRT.exploring on the bench world, every decision compared",
        "world": "synth.make_world(seed=2, n_obstacles=%d, box=+-1000 m, cell=10 m): 40 000 cells, 10 bins" % args.obstacles,
        "episodes": args.seeds, "iters_per_episode": args.iters, "seeds": "0..%d" % (args.seeds - 1),
        "expansions_compared": int(args.seeds) * int(args.iters),
        "episodes_with_a_different_decision": len(div),
        "divergent": [{"seed": r[0], "first_different_iteration": r[2]} for r in div],
        "episodes_bit_identical_in_every_node_float": int(sum(1 for r in res if r[5])),
        "max_abs_node_float_difference_over_non_divergent_episodes": float(nd.max()) if len(nd) else None,
        "max_abs_best_cost_difference": float(cd.max()) if len(cd) else None,
        "mean_nodes_per_episode": float(np.mean([r[6] for r in res])),
        "wall_seconds": time.time() - t0, "procs": args.procs,
    }
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
