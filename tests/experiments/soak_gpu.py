#!/usr/bin/env python3
"""Randomised GPU-vs-checker soak (run by hand on a GPU box: python tests/experiments/soak_gpu.py [n_cases] [seed] [big]).
Every case draws a world and planner parameters at random and compares, bit for bit, the four-episode-per-wavefront
kernel, the one-episode kernel and the CPU checker (RRT.exploring, time-bin sampling; nearest-neighbour -- scan and fallback --
and plan-time sampling), the astar_fixLenSOG / astar_fixLen searches and Planner_RRT.planning (latency, throughput and
four-episodes-per-wavefront kernels) with the checker.  Prints one line per failure and a summary; exit code 1 on any mismatch."""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from auv_sim_amd import _lib, _astar_lib, synth  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import env_options  # noqa: E402  (tests/env_options.py: AUVP_<NAME> in os.environ steers live contexts -- this process only)
env_options.install()
from auv_sim_amd._prrt_lib import PlannerBatch  # noqa: E402
from oracle import orc, orc_astar as oa, orc_planner as op  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"  # fewer, longer RRT episodes (member-list chunks, thousands of leaves)
ctx = _lib.Context(0)
fails = 0


def rrt_case(i):
    global fails
    neg = rng.random() < 0.6
    size = rng.choice([120.0, 200.0, 400.0, 900.0])
    x0, y0 = (-300.0, -100.0) if neg else (rng.choice([0.0, 15.0]), rng.choice([0.0, 40.0]))
    cell = rng.choice([7.0, 10.0, 14.0, 25.0])
    nob = rng.choice([0, 3, 40, 64, 130, 256])
    w = synth.make_world(seed=rng.randrange(10 ** 6), n_obstacles=nob, box=(x0, y0, x0 + size, y0 + size * rng.choice([0.6, 1.0])),
                         cell=cell, n_habitats=rng.choice([0, 1, 10, 40]), obst_radius=(1.0, rng.choice([3.0, 9.0])),
                         hab_radius=(5.0, rng.choice([15.0, 50.0])), n_bins=rng.choice([1, 4, 10]), bin_len=rng.choice([20, 50]),
                         # round 6: a third of the worlds have a boundary that is NOT a rectangle (the reference's 5-vertex
                         # Catalina outline, a concave 8-vertex one): the polygon crossing test instead of the rectangle shortcut
                         polygon=rng.choice([None, None, None, None, "catalina", "notch"]))
    kw = dict(freq=rng.choice([1, 7, 15, 16, 29, 30]), dist_to_end=rng.choice([0.5, 2.0, 5.0]), diff_max=rng.choice([0.1, 0.5, 2.0]),
              min_dist=rng.choice([0.0, 0.5, 1.5]), bin_interval=rng.choice([2.5, 5.0, 20.0]), v=rng.choice([0.7, 2.0]),
              max_traj_time=rng.choice([40.0, 120.0, 500.0]),
              weights=(rng.choice([-3.0, 0.0, 2.5]), rng.choice([-3.0, -0.37, 4.0]), rng.choice([-4.0, 0.0, 1.7])))
    E, n_iter = (rng.choice([1, 5, 9]), rng.choice([200, 700, 1500])) if not BIG else (2, rng.choice([4000, 8000]))
    ctx.set_world(w["obstacles"], w["habitats"], w["polygon"], w["bins"], w["cells"], w["prob"])
    init = np.zeros((E, 6)); init[:, 0], init[:, 1] = w["start"]; init[:, 2] = np.linspace(-3, 3, E)
    seeds = np.array([rng.randrange(2 ** 40) for _ in range(E)], dtype=np.uint64)
    res = {}
    for rows in ("1", "0"):
        os.environ["AUVP_ROWS"] = rows
        for tight in ("0", "1"):
            os.environ["AUVP_TIGHT_CULL"] = tight
            s = ctx.rrt_explore_batch(init, seeds, n_iter, **kw).copy()
            res[rows + tight] = (s, [ctx.tree(e, s[e]) for e in range(E)], ctx.paths(s))
    os.environ.pop("AUVP_ROWS"); os.environ.pop("AUVP_TIGHT_CULL")
    wo = orc.WorldArrays(w["obstacles"], w["habitats"], w["polygon"], w["bins"], w["cells"], w["prob"])
    for e in range(E):
        r = orc.rrt_explore(wo, int(seeds[e]), n_iter, init=init[e], kind="portable", **kw)
        for key, (s, t, p) in res.items():
            ok = (s[e]["status"], s[e]["n_nodes"], s[e]["n_points"], s[e]["n_leaves"], s[e]["best_leaf"]) == \
                 (r["status"], r["n_nodes"], r["n_points"], r["n_leaves"], r["best_leaf"])
            ok = ok and np.array_equal(t[e]["parent"], r["parent"]) and np.array_equal(t[e]["nodes"], r["nodes"]) and \
                np.array_equal(t[e]["points"], r["points"]) and s[e]["rng_after"] == r["rng_after"]
            if ok and r["status"] == 0:
                ok = np.array_equal(np.array(s[e]["best_cost"]), r["best_cost"]) and np.array_equal(p[e], r["path"])
            if not ok:
                fails += 1
                print("RRT MISMATCH case", i, "episode", e, "variant rows/tight", key, kw, "obst", nob)


def rrt_modes_case(i):
    """the other parent-sampling modes of RRT.exploring: nearest neighbour (streaming scan and its sqrt-per-node fallback) and
    plan-time bisect, against the checker"""
    global fails
    size = rng.choice([120.0, 300.0, 900.0])
    nob = rng.choice([0, 40, 256])
    w = synth.make_world(seed=rng.randrange(10 ** 6), n_obstacles=nob, box=(-300.0, -100.0, -300.0 + size, -100.0 + size),
                         cell=rng.choice([10.0, 25.0]), n_habitats=rng.choice([0, 10]), obst_radius=(1.0, rng.choice([3.0, 9.0])),
                         polygon=rng.choice([None, None, "catalina", "notch"]))
    kw = dict(freq=rng.choice([1, 7, 30, 45]), dist_to_end=rng.choice([0.5, 2.0, 5.0]), diff_max=rng.choice([0.1, 0.5]),
              min_dist=rng.choice([0.0, 0.5]), v=rng.choice([0.7, 2.0]), max_traj_time=rng.choice([40.0, 500.0, 5000.0]),
              weights=(rng.choice([-3.0, 2.5]), rng.choice([-3.0, 4.0]), rng.choice([-4.0, 1.7])))
    E, n_iter = rng.choice([1, 6]), rng.choice([300, 1200, 2500])
    ctx.set_world(w["obstacles"], w["habitats"], w["polygon"], w["bins"], w["cells"], w["prob"])
    init = np.zeros((E, 6)); init[:, 0], init[:, 1] = w["start"]; init[:, 2] = np.linspace(-3, 3, E)
    seeds = np.array([rng.randrange(2 ** 40) for _ in range(E)], dtype=np.uint64)
    wo = orc.WorldArrays(w["obstacles"], w["habitats"], w["polygon"], w["bins"], w["cells"], w["prob"])
    for mode, env in (("nn", {}), ("nn", {"AUVP_NN_EXACT": "1"}), ("plantime", {})):
        os.environ.update(env)
        s = ctx.rrt_explore_batch(init, seeds, n_iter, mode=mode, **kw).copy()
        trees = [ctx.tree(e, s[e]) for e in range(E)]
        for k in env:
            os.environ.pop(k)
        for e in range(E):
            r = orc.rrt_explore(wo, int(seeds[e]), n_iter, mode=mode, init=init[e], kind="portable", **kw)
            ok = (s[e]["status"], s[e]["n_nodes"], s[e]["n_points"], s[e]["n_leaves"], s[e]["best_leaf"]) == \
                 (r["status"], r["n_nodes"], r["n_points"], r["n_leaves"], r["best_leaf"])
            ok = ok and np.array_equal(trees[e]["parent"], r["parent"]) and np.array_equal(trees[e]["nodes"], r["nodes"]) and \
                s[e]["rng_after"] == r["rng_after"]
            if ok and r["status"] == 0:
                ok = np.array_equal(np.array(s[e]["best_cost"]), r["best_cost"])
            if not ok:
                fails += 1
                print("RRT MODE MISMATCH case", i, "episode", e, mode, env, kw, "obst", nob)


def astar_case(i):
    global fails
    cell = rng.choice([5.0, 10.0, 14.0, 20.0])
    w = synth.make_world(seed=rng.randrange(10 ** 6), n_obstacles=rng.choice([0, 20, 64]), obst_radius=(2.0, rng.choice([4.0, 8.0])),
                         n_habitats=rng.choice([0, 5, 12]), hab_radius=(8.0, 25.0), cell=cell)
    off = rng.choice([0.0, 0.0, 0.37])
    starts = np.array([(-290.0 + 10.0 * rng.randrange(0, 10) + off, -90.0 + 10.0 * rng.randrange(0, 10) + off) for _ in range(6)])
    variant = rng.choice(["astar_fixLenSOG", "astar_fixLenSOG", "astar_fixLen"])
    kw = dict(obstacles=w["obstacles"], polygon=w["polygon"], habitats=w["habitats"], limit=rng.choice([60.0, 150.0, 260.0]),
              weights=(0, rng.choice([3, 10]), rng.choice([0, 10]), rng.choice([1, 100])))
    if variant == "astar_fixLenSOG":
        kw.update(bins=w["bins"], cells=w["cells"], prob=w["prob"], velocity=rng.choice([1.0, 1.0, 0.8]))
    ctx.set_world(kw.get("obstacles"), kw.get("habitats"), kw.get("polygon"), kw.get("bins"), kw.get("cells"), kw.get("prob"))
    res = _astar_lib.run_batch(ctx, variant, starts, limits=np.full(len(starts), kw["limit"]), velocity=kw.get("velocity", 1.0),
                               weights=kw["weights"], exp_log=True)
    for e, r in enumerate(res):
        o = oa.run(variant, starts[e], kind="portable", cap_nodes=20000, **kw)
        if r["status"] < 0 or o["status"] < 0:
            # an input the reference raises on (e.g. int(dist_left) > number of cells: IndexError in get_top_n_prob): both sides
            # must refuse it (the checker and the C-ABI number their error codes differently), at the same expansion
            ok = r["status"] < 0 and o["status"] < 0 and abs(len(r["expansions"]) - len(o["expansions"])) <= 1
        else:
            ok = r["found"] == o["found"] and r["n_nodes"] == o["n_nodes"] and r["n_children"] == o["n_children"]
            ok = ok and np.array_equal(r["expansions"], o["expansions"]) and np.array_equal(r["path"], o["path"]) and \
                np.array_equal(r["cost_list"], o["cost_list"]) and np.array_equal(r["smooth_path"], o["smooth_path"]) and \
                np.array_equal(r["hab_left"], o["hab_left"])
        if not ok:
            fails += 1
            print("A* MISMATCH case", i, variant, "instance", e, "cell", cell, "off", off, kw["limit"], kw["weights"])


def planner_case(i):
    global fails
    size = rng.choice([100.0, 200.0])
    # round 6: every third world is translated -- the bucket grid ignores the origin (gym_rrt/envs/rrt_dubins.py:115-116), so
    # indexes wrap (negative origin), run past the grid (positive origin: unbucketed nodes) or raise (the checker's
    # ORC_ERR_ARG = the C-ABI's AUVP_ERR_ARG: both sides must stop at the same step with the same tree)
    origin = rng.choice([(0.0, 0.0), (0.0, 0.0), (round(rng.uniform(-1.3, 0.6) * size, 2), round(rng.uniform(-1.3, 0.6) * size, 2))])
    w = synth.make_rect_world(seed=rng.randrange(10 ** 6), n_obstacles=rng.choice([0, 30, 100, 256]), size=size, origin=origin)
    ctx.set_world(obstacles=w["obstacles"])
    n_ep, max_step = 4, rng.choice([60, 300, 900])
    freq, cell, subs = rng.choice([3, 10, 25]), rng.choice([2.0, 5.0]), rng.choice([1, 4, 8])
    starts = np.tile(np.array([w["start"][0], w["start"][1], rng.uniform(-3, 3), 0.0]), (n_ep, 1))
    goals = np.tile(w["goal"], (n_ep, 1))
    seeds = np.array([rng.randrange(2 ** 40) for _ in range(n_ep)], dtype=np.uint64)
    goals[1] = [w["start"][0] + rng.uniform(3, 12), w["start"][1] + rng.uniform(-3, 3)]  # a goal that is reached early
    # the one-episode kernel (latency and throughput instantiations) and, where its limits allow, four episodes per wavefront
    variants = [("lat", {"AUVP_PRRT_ROWS": "0", "AUVP_PRRT_LAT": "1"}), ("thr", {"AUVP_PRRT_ROWS": "0", "AUVP_PRRT_LAT": "0"})]
    if freq <= 15 and len(w["obstacles"]) <= 256:
        variants.append(("rows", {"AUVP_PRRT_ROWS": "1", "AUVP_PRRT_LAT": "0"}))
    ref = [op.planning(w["obstacles"], w["rect"], starts[e], goals[e], int(seeds[e]), max_step, freq=freq, cell=cell, subs=subs,
                       kind="portable") for e in range(n_ep)]
    for name, env in variants:
        os.environ.update(env)
        pb = PlannerBatch(ctx, starts, goals, w["rect"], max_step, seeds=seeds, freq=freq, cell=cell, subs=subs)
        s = pb.plan()
        trees = [pb.tree(e, s[e]) for e in range(n_ep)]
        for k in env:
            os.environ.pop(k)
        for e in range(n_ep):
            r = ref[e]
            st = {-1: -2, -2: -1}.get(int(s["status"][e]), int(s["status"][e]))  # (the two sides number ARG / CAPACITY differently)
            ok = (st, int(s["n_nodes"][e]), int(s["steps"][e]), bool(s["done"][e])) == \
                 (r["status"], r["n_nodes"], r["steps"], r["done"]) and float(s["rng_after"][e]) == r["rng_after"]
            ok = ok and np.array_equal(trees[e]["nodes"], r["nodes"][:, :4]) and np.array_equal(trees[e]["parent"], r["parent"])
            if not ok:
                fails += 1
                print("PLANNER MISMATCH case", i, "episode", e, name, freq, cell, subs, max_step, "origin", origin, "status", int(s["status"][e]), r["status"])


for i in range(n_cases):
    rrt_case(i)
    rrt_modes_case(i)
    astar_case(i)
    try:
        planner_case(i)
    except AttributeError as e:  # checker API differs: report once
        if i == 0:
            print("planner soak skipped:", e)
print("soak: %d cases, %d mismatches" % (n_cases, fails))
sys.exit(1 if fails else 0)
