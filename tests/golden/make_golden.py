#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing the reference planners.

TEST INFRASTRUCTURE.  Runs ONLY in the build container (needs /root/reference); the GPU box and the
test-suite read the committed .npz/.json fixtures, never this script's imports.  No reference source
text is written anywhere: fixtures hold inputs and outputs only.

usage:  python tests/golden/make_golden.py [g7 g5 g4 g3 g2 g1 g6 ...]   (default: all)
"""
import contextlib
import hashlib
import importlib
import io
import json
import math
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = HERE  # where the fixtures are written: this directory, or a scratch directory in --check mode
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("AUVP_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(HERE, "_refstubs"))

import install as refstubs  # noqa: E402
from auv_sim_amd import synth  # noqa: E402


def _purge(names):
    for n in list(sys.modules):
        if n in names or any(n.startswith(p + ".") for p in names):
            del sys.modules[n]


_SHARED = {"cost", "motion_plan_state", "catalina", "sharkOccupancyGrid", "sharkEstimate",
           "path_planning", "rrt_dubins", "astar", "astar_real", "astar_fixLen", "astar_fixLenSOG"}


def import_rrt():
    """path_planning/rrt_dubins.py with sys.path=[path_planning/, root]."""
    refstubs.install()
    _purge(_SHARED)
    saved = list(sys.path)
    sys.path[:0] = [os.path.join(REF, "path_planning"), REF]
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            mod = importlib.import_module("rrt_dubins")
            mps = importlib.import_module("motion_plan_state")
            cost = importlib.import_module("cost")
    finally:
        sys.path[:] = saved
    return mod, mps, cost


def import_astar(name):
    """astar*.py need root cost.Cost: sys.path=[root, path_planning/]."""
    refstubs.install()
    _purge(_SHARED)
    saved = list(sys.path)
    sys.path[:0] = [REF, os.path.join(REF, "path_planning")]
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            mod = importlib.import_module(name)
            mps = importlib.import_module("motion_plan_state")
    finally:
        sys.path[:] = saved
    return mod, mps


def import_gym_rrt():
    refstubs.install()
    _purge({"gym_rrt"})
    saved = list(sys.path)
    sys.path[:0] = [REF]
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            mod = importlib.import_module("gym_rrt.envs.rrt_dubins")
            mps = importlib.import_module("gym_rrt.envs.motion_plan_state_rrt")
    finally:
        sys.path[:] = saved
    return mod, mps


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def save_npz(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print("wrote", path, os.path.getsize(path), "bytes")


# --------------------------------------------------------------------------------------------
# G7: CPython random known answers (stdlib only)
# --------------------------------------------------------------------------------------------
def g7():
    out = {}
    for seed in (0, 1, 7, 1234, 2 ** 32 + 5, 2 ** 63 + 12345):
        r = random.Random(seed)
        out[str(seed)] = {
            "random": [r.random() for _ in range(5)],
            "getrandbits32": [r.getrandbits(32) for _ in range(4)],
            "uniform_1_101": r.uniform(1, 101),
            "uniform_m05_05": r.uniform(-0.5, 0.5),
            "choice_range100": [r.choice(range(100)) for _ in range(8)],
            "choice_range3": [r.choice(range(3)) for _ in range(8)],
            "choice_range1": [r.choice(range(1)) for _ in range(3)],
            "after": r.random(),
        }
        # long stream crossing several 624-word refills
        r = random.Random(seed)
        stream = np.array([r.random() for _ in range(2000)])
        out[str(seed)]["stream2000_sha"] = sha(stream)
        out[str(seed)]["stream_1999"] = float(stream[1999])
    # libm known answers of this container: lets the tests decide whether bit-exact equality with
    # the reference's math.sin/cos/pow can be asserted on the machine running them
    xs = [0.1 * k - 7.0 for k in range(141)]
    out["libm"] = {
        "x": xs,
        "sin": [math.sin(x) for x in xs],
        "cos": [math.cos(x) for x in xs],
        "atan2_x_1": [math.atan2(x, 1.0) for x in xs],
        "pow2": [x ** 2 for x in xs],
    }
    with open(os.path.join(OUT, "g7_random_kat.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote g7_random_kat.json")


# --------------------------------------------------------------------------------------------
# helpers to build reference-side objects from a synth world
# --------------------------------------------------------------------------------------------
def ref_world(world, MPS):
    obstacles = [MPS(o[0], o[1], size=o[2]) for o in world["obstacles"].tolist()]
    habitats = [MPS(h[0], h[1], size=h[2]) for h in world["habitats"].tolist()]
    poly = refstubs.Polygon([tuple(p) for p in world["polygon"].tolist()])
    cell_list = [refstubs.CellStub(*c) for c in world["cells"].tolist()]
    shark = {}
    for t, b in enumerate(world["bins"].tolist()):
        key = (int(b[0]), int(b[1]))
        shark[key] = {cell_list[i].bounds: p for i, p in enumerate(world["prob"][t].tolist())}
    return obstacles, habitats, poly, cell_list, shark


def world_arrays(world):
    return {k: world[k] for k in ("obstacles", "habitats", "polygon", "bins", "cells", "prob")}


# --------------------------------------------------------------------------------------------
# G5: check_collision order dependence (prefix-min quirk)
# --------------------------------------------------------------------------------------------
def g5():
    rrt_mod, mpsm, _ = import_rrt()
    MPS = mpsm.Motion_plan_state
    poly = refstubs.Polygon([(0, 0), (200, 0), (200, 200), (0, 200)])
    cases = []
    rng = random.Random(55)

    def run(points, obstacles, polygon=poly):
        rrt = rrt_mod.RRT(polygon, [MPS(o[0], o[1], size=o[2]) for o in obstacles], {}, [])
        node = MPS(points[-1][0], points[-1][1])
        node.path = [MPS(p[0], p[1]) for p in points]
        return bool(rrt.check_collision(node, rrt.obstacle_list))

    hand = [
        ([(50.0, 50.0)], [(52.0, 50.0, 1.0), (120.0, 50.0, 5.0)]),
        ([(50.0, 50.0)], [(120.0, 50.0, 5.0), (52.0, 50.0, 1.0)]),
        ([(50.0, 50.0), (60.0, 50.0)], [(70.0, 50.0, 1.0), (61.5, 50.0, 1.0), (150.0, 150.0, 9.5)]),
        ([(10.0, 10.0), (250.0, 10.0)], [(100.0, 100.0, 1.0)]),       # leaves the polygon
        ([(10.0, 10.0)], []),
        ([(100.0, 100.0)], [(100.0, 103.0, 3.0)]),                    # d == size -> collision
    ]
    for pts, obs in hand:
        cases.append((pts, obs, run(pts, obs)))
    for _ in range(200):
        npt = rng.randint(1, 14)
        nob = rng.randint(0, 12)
        pts = [(rng.uniform(-10, 210), rng.uniform(-10, 210)) for _ in range(npt)]
        if rng.random() < 0.7:
            pts = [(min(max(x, 1.0), 199.0), min(max(y, 1.0), 199.0)) for x, y in pts]
        obs = [(rng.uniform(0, 200), rng.uniform(0, 200), rng.uniform(0.5, 30.0)) for _ in range(nob)]
        cases.append((pts, obs, run(pts, obs)))
    # non-rectangular polygon (Catalina-like pentagon)
    penta = refstubs.Polygon([(0, 0), (180, -20), (240, 90), (120, 200), (-30, 120)])
    pcases = []
    for _ in range(200):
        pts = [(rng.uniform(-40, 250), rng.uniform(-30, 210)) for _ in range(rng.randint(1, 6))]
        pcases.append((pts, [], run(pts, [], penta)))
    out = {"rect": [[0, 0], [200, 0], [200, 200], [0, 200]],
           "penta": [[0, 0], [180, -20], [240, 90], [120, 200], [-30, 120]],
           "cases": [{"pts": p, "obs": o, "free": r} for p, o, r in cases],
           "penta_cases": [{"pts": p, "obs": o, "free": r} for p, o, r in pcases]}
    with open(os.path.join(OUT, "g5_collision.json"), "w") as f:
        json.dump(out, f)
    print("wrote g5_collision.json", sum(c[2] for c in cases), "free of", len(cases),
          "| penta", sum(c[2] for c in pcases), "of", len(pcases))


# --------------------------------------------------------------------------------------------
# G4: habitat_shark_cost_func on fixed paths
# --------------------------------------------------------------------------------------------
def g4():
    _, mpsm, cost = import_rrt()
    MPS = mpsm.Motion_plan_state
    rng = random.Random(44)
    cases = []
    for k in range(24):
        world = synth.make_world(seed=100 + k, n_obstacles=4, n_habitats=(0 if k == 5 else 6 + k % 5),
                                 cell=10.0 if k % 3 else 20.0, n_bins=4 + k % 4)
        _, habitats, _, _, shark = ref_world(world, MPS)
        x0, y0, x1, y1 = world["box"].tolist()
        T = len(world["bins"])
        npt = rng.randint(1, 120)
        pts = []
        for _ in range(npt):
            # includes points outside the box, on cell edges, and out-of-bin timestamps
            x = rng.uniform(x0 - 15, x1 + 15)
            y = rng.uniform(y0 - 15, y1 + 15)
            if rng.random() < 0.15:
                x = float(round(x / 10.0) * 10.0)
            if rng.random() < 0.15:
                y = float(round(y / 10.0) * 10.0)
            t = rng.uniform(-20.0, 50.0 * T + 40.0)
            if rng.random() < 0.1:
                t = float(50 * rng.randint(0, T))
            pts.append((x, y, t))
        # sub-dict of bins as exploring builds it (any subset, order preserved)
        keys = list(shark.keys())
        lo = rng.randint(0, len(keys) - 1)
        hi = rng.randint(lo, len(keys))
        sub = {kk: shark[kk] for kk in keys[lo:hi]}
        total = rng.choice([0.0, rng.uniform(1.0, 500.0)])
        weights = rng.choice([[-3, -3, -4], [-1, -1, -1], [-0.5, -2.25, -1.75]])
        path = [MPS(p[0], p[1], traj_time_stamp=p[2]) for p in pts]
        res = cost.habitat_shark_cost_func(path, total, habitats, sub, weights)
        cases.append({
            "world_seed": 100 + k, "n_habitats": len(habitats), "cell": 10.0 if k % 3 else 20.0,
            "n_bins": T, "pts": pts, "bin_lo": lo, "bin_hi": hi, "total": total, "weights": weights,
            "out": [float(res[0])] + [float(c) for c in res[1]],
        })
    with open(os.path.join(OUT, "g4_cost.json"), "w") as f:
        json.dump({"cases": cases}, f)
    print("wrote g4_cost.json")


# --------------------------------------------------------------------------------------------
# G3: RRT.exploring under the virtual clock
# --------------------------------------------------------------------------------------------
def run_exploring(seed, world, n_iter, mode, freq=30, bin_interval=5, v=2, shark_interval=50,
                  max_traj_time=500.0, weights=(-3, -3, -4), dist_to_end=2, diff_max=0.5,
                  keep_points=True, count_boundary=False):
    rrt_mod, mpsm, cost_mod = import_rrt()
    MPS = mpsm.Motion_plan_state
    obstacles, habitats, poly, cell_list, shark = ref_world(world, MPS)
    clock = refstubs.VirtualClock()
    rrt_mod.time = clock
    rrt = rrt_mod.RRT(poly, obstacles, shark, cell_list, dist_to_end=dist_to_end, diff_max=diff_max,
                      freq=freq)
    log = {"parent": [], "accepted": [], "npath": [], "leaf_cost": [], "leaf_iter": []}
    index_of = {}
    orig_steer = rrt.steer
    orig_cc = rrt.check_collision
    orig_cost = rrt_mod.habitat_shark_cost_func
    cur = {}

    def steer(m, *a, **k):
        # the parent's index in mps_list at pick time
        for i, n in enumerate(rrt.mps_list[len(index_of):], start=len(index_of)):
            index_of[id(n)] = i
        cur["parent"] = index_of[id(m)]
        new = orig_steer(m, *a, **k)
        cur["npath"] = len(new.path)
        return new

    far = refstubs.Polygon([(-1e9, -1e9), (1e9, -1e9), (1e9, 1e9), (-1e9, 1e9)])

    def check_collision(m, obs):
        r = orig_cc(m, obs)
        if not r:
            # a rejection the boundary alone decided: the same path is free once the polygon test cannot fail (no draws involved)
            keep, rrt.boundary_poly = rrt.boundary_poly, far
            log["boundary_rejects"] = log.get("boundary_rejects", 0) + (1 if orig_cc(m, obs) else 0)
            rrt.boundary_poly = keep
        log["parent"].append(cur["parent"])
        log["npath"].append(cur["npath"])
        log["accepted"].append(bool(r))
        return r

    def cost_func(path, total, habs, sd, w):
        r = orig_cost(path, total, habs, sd, w)
        log["leaf_cost"].append([float(r[0])] + [float(c) for c in r[1]] + [float(len(path)), float(len(sd))])
        log["leaf_iter"].append(len(log["accepted"]) - 1)
        return r

    rrt.steer = steer
    rrt.check_collision = check_collision
    rrt_mod.habitat_shark_cost_func = cost_func
    random.seed(seed)
    start = MPS(float(world["start"][0]), float(world["start"][1]))
    traj_ts = mode in ("timebin",)
    plan_time = mode in ("timebin", "plantime")
    err = None
    try:
        res = rrt.exploring(start, habitats, float(n_iter), bin_interval, v, shark_interval,
                            traj_time_stamp=traj_ts, max_plan_time=float(n_iter),
                            max_traj_time=max_traj_time, plan_time=plan_time, weights=list(weights))
    except TypeError as e:  # no qualifying leaf: opt_path is None (rrt_dubins.py:174)
        res = None
        err = str(e)
    rng_after = random.random()
    nodes = rrt.mps_list
    for i, n in enumerate(nodes):
        index_of[id(n)] = i
    node_arr = np.array([[n.x, n.y, n.theta, n.traj_time_stamp, n.plan_time_stamp, n.length] for n in nodes])
    parent = np.array([-1 if n.parent is None else index_of[id(n.parent)] for n in nodes], dtype=np.int32)
    npath = np.array([len(n.path) for n in nodes], dtype=np.int32)
    pts = []
    for n in nodes[1:]:
        for p in n.path[1:]:
            pts.append([p.x, p.y, p.theta, p.v, p.traj_time_stamp, p.plan_time_stamp, p.length])
    pts = np.array(pts, dtype=np.float64).reshape(-1, 7)
    out = {
        "seed": seed, "n_iter": n_iter, "mode": mode, "freq": freq, "bin_interval": bin_interval, "v": v,
        "shark_interval": shark_interval, "max_traj_time": max_traj_time, "weights": list(weights),
        "dist_to_end": dist_to_end, "diff_max": diff_max,
        "iters_run": len(log["accepted"]),
        "nodes": node_arr, "parent": parent, "npath": npath,
        "it_parent": np.array(log["parent"], dtype=np.int32),
        "it_accepted": np.array(log["accepted"], dtype=np.int8),
        "it_npath": np.array(log["npath"], dtype=np.int32),
        "leaf_cost": np.array(log["leaf_cost"], dtype=np.float64).reshape(-1, 6),
        "leaf_iter": np.array(log["leaf_iter"], dtype=np.int32),
        "rng_after": rng_after,
        "points_sha": sha(pts),
        "n_points": len(pts),
        "error": err or "",
    }
    if count_boundary:
        out["boundary_rejects"] = int(log.get("boundary_rejects", 0))
    if keep_points:
        out["points"] = pts
    if res is not None:
        path = res["path"][0]
        out["res_path_length"] = float(res["path length"])
        out["res_cost"] = np.array([float(res["cost"][0])] + [float(c) for c in res["cost"][1]])
        out["res_path"] = np.array([[p.x, p.y, p.theta, p.v, p.traj_time_stamp, p.plan_time_stamp, p.length]
                                    for p in path])
        split = res["path"][1]
        out["res_split_keys"] = np.array([[k[0], k[1]] for k in split.keys()], dtype=np.float64).reshape(-1, 2)
        out["res_split_counts"] = np.array([len(vv) for vv in split.values()], dtype=np.int32)
        # bin contents (time-bin mode): sizes per regular key, in key order
    if traj_ts:
        keys = sorted(k for k in rrt.time_bin.keys())
        out["bin_keys"] = np.array(keys, dtype=np.float64)
        out["bin_sizes"] = np.array([len(rrt.time_bin[k]) for k in keys], dtype=np.int32)
    return out


def g3():
    specs = [
        # name, seed, world kwargs, n_iter, mode, extra
        ("g3_tb_o64_i500", 7, dict(seed=1, n_obstacles=64), 500, "timebin", {}),
        ("g3_tb_o64_i2000", 7, dict(seed=1, n_obstacles=64), 2000, "timebin", {}),
        ("g3_tb_o256_i500", 11, dict(seed=2, n_obstacles=256), 500, "timebin", {}),
        ("g3_nn_o64_i500", 5, dict(seed=1, n_obstacles=64), 500, "nn", {}),
        ("g3_nn_o256_i2000", 5, dict(seed=2, n_obstacles=256), 2000, "nn", {"keep_points": False}),
        ("g3_pt_o64_i500", 3, dict(seed=1, n_obstacles=64), 500, "plantime", {}),
        ("g3_tb_short_traj", 9, dict(seed=3, n_obstacles=64, n_bins=4), 800, "timebin",
         {"max_traj_time": 120.0, "shark_interval": 30, "weights": (-1, -1, -1)}),
        ("g3_tb_dense", 13, dict(seed=4, n_obstacles=64, obst_radius=(4.0, 9.0)), 600, "timebin",
         {"freq": 12}),
        ("g3_tb_o256_i10000", 7, dict(seed=2, n_obstacles=256), 10000, "timebin", {"keep_points": False}),
        # edge cases (commit e55714b; their specs were missing from this list until round 5): a horizon that is not a multiple
        # of the bin interval, so that inserts beyond it reset the overflow bin (rrt_dubins.py:148-151); more sub-arcs per
        # steer than a wavefront has lanes (freq 100 / 70); a world without obstacles and habitats, nearest-neighbour sampling
        ("g3_tb_binreset", 21, dict(seed=5, n_obstacles=40, n_bins=3), 900, "timebin", {"max_traj_time": 122.0, "shark_interval": 30}),
        ("g3_tb_freq100", 22, dict(seed=6, n_obstacles=64), 300, "timebin", {"freq": 100, "max_traj_time": 300.0}),
        ("g3_nn_freq70_noobs", 23, dict(seed=7, n_obstacles=0, n_habitats=0), 300, "nn", {"freq": 70, "max_traj_time": 200.0}),
        # the bench world itself (bench.py: 256 obstacles, 200x200 cells of 10 m = 40 000 cells, 10 bins) with a short
        # horizon so the reference's linear cell scan (path_planning/cost.py:181-184, ~20 000 dict entries per path
        # point) finishes: pins the device's cell index at the headline size against the reference.  The 4.5 MB of
        # world tables are not stored: the world is synth.make_world(**world_kwargs) and its SHA-256 is.
        ("g3_tb_c40000", 7, dict(seed=2, n_obstacles=256, box=(-1000.0, -1000.0, 1000.0, 1000.0), cell=10.0, n_bins=10,
                                 bin_len=50, n_habitats=10), 1500, "timebin",
         {"max_traj_time": 120.0, "shark_interval": 30, "keep_points": False, "store_world": False}),
        # round 3: nearest-neighbour parent sampling on the bench world -- the headline's horizon (the time-stamp rule of
        # :138-139 rejects most samples, ~100 qualifying leaves) and a horizon long enough for the tree to keep growing
        # (every iteration scans a longer list; duplicated positions and near-equidistant nodes decide get_closest_mps)
        ("g3_nn_c40000_i3000", 7, dict(seed=2, n_obstacles=256, box=(-1000.0, -1000.0, 1000.0, 1000.0), cell=10.0, n_bins=10,
                                       bin_len=50, n_habitats=10), 3000, "nn",
         {"max_traj_time": 500.0, "keep_points": False, "store_world": False}),
        ("g3_nn_long_c40000_i4000", 7, dict(seed=2, n_obstacles=256, box=(-1000.0, -1000.0, 1000.0, 1000.0), cell=10.0, n_bins=10,
                                            bin_len=50, n_habitats=10), 4000, "nn",
         {"max_traj_time": 20000.0, "keep_points": False, "store_world": False}),
        # round 6: whole episodes on boundaries that are NOT rectangles -- the device's polygon crossing test instead of its
        # rectangle shortcut, against Point.within(self.boundary_poly) (:545-546) -- `boundary_rejects` counts the steers the
        # boundary alone rejected (>= 100 in each).  "catalina": the reference's 5-vertex workspace outline
        # (path_planning/catalina.py:71-73) over the bench's Catalina-sized side world (560 x 350 m, 1 000 cells of 14 m, 256
        # obstacles r = 2-8 m; bench.py: catalina_560x350_o256), "notch": a concave 8-vertex outline; and that side world
        # itself with its rectangle.  The catalina worlds' tables are rebuilt from world_kwargs (SHA-256 stored).
        ("g3_tb_cat_penta", 31, dict(seed=5, n_obstacles=256, box=(0.0, 0.0, 560.0, 350.0), cell=14.0, obst_radius=(2.0, 8.0),
                                     hab_radius=(20.0, 55.0), polygon="catalina", start=(150.0, 160.0)), 2500, "timebin",
         {"keep_points": False, "store_world": False, "count_boundary": True}),
        ("g3_nn_cat_penta", 32, dict(seed=5, n_obstacles=256, box=(0.0, 0.0, 560.0, 350.0), cell=14.0, obst_radius=(2.0, 8.0),
                                     hab_radius=(20.0, 55.0), polygon="catalina", start=(150.0, 160.0)), 2000, "nn",
         {"max_traj_time": 300.0, "keep_points": False, "store_world": False, "count_boundary": True}),
        ("g3_tb_notch", 33, dict(seed=8, n_obstacles=64, polygon="notch"), 2500, "timebin", {"count_boundary": True, "keep_points": False}),
        ("g3_nn_notch", 34, dict(seed=8, n_obstacles=64, polygon="notch"), 2000, "nn",
         {"max_traj_time": 300.0, "count_boundary": True, "keep_points": False}),
        ("g3_tb_cat_rect", 35, dict(seed=5, n_obstacles=256, box=(0.0, 0.0, 560.0, 350.0), cell=14.0, obst_radius=(2.0, 8.0),
                                    hab_radius=(20.0, 55.0)), 2500, "timebin",
         {"keep_points": False, "store_world": False, "count_boundary": True}),
    ]
    only = os.environ.get("AUVP_G3_ONLY")
    for name, seed, wk, n_iter, mode, extra in specs:
        if only and name not in only.split(","):
            continue
        extra = dict(extra)
        store_world = extra.pop("store_world", True)
        world = synth.make_world(**wk)
        out = run_exploring(seed, world, n_iter, mode, **extra)
        meta = {"world_kwargs": json.dumps(wk)}
        wa = world_arrays(world)
        if not store_world:
            meta["world_sha"] = sha(*[wa[k] for k in ("obstacles", "habitats", "polygon", "bins", "cells", "prob")])
            wa = {}
        save_npz(name + ".npz", **wa, start=world["start"], **meta, **out)
        print(name, "iters", out["iters_run"], "nodes", len(out["nodes"]), "leaves", len(out["leaf_iter"]),
              "err", out["error"], "cost", out.get("res_cost"), "boundary rejects", out.get("boundary_rejects"))


# --------------------------------------------------------------------------------------------
# G2: Planner_RRT.planning (gym_rrt/envs/rrt_dubins.py:162) -- step bounded, deterministic under seed
# --------------------------------------------------------------------------------------------
def main_layout_obstacles():
    """the deterministic 23-obstacle layout of gym_rrt/envs/rrt_dubins.py main() (:611-644)"""
    obs = []
    x, y = 15.0, 15.0
    for _ in range(5):
        obs.append((x, y, 3.0)); x -= 3.0; y += 3.0
    x, y = 3.0, 33.0
    for _ in range(2):
        obs.append((x, y, 3.0)); y += 6.0
    x, y = 9.0, 39.0
    for _ in range(4):
        obs.append((x, y, 3.0)); x += 6.0
    x, y = 25.0, 24.0
    for _ in range(8):
        obs.append((x, y, 3.0)); x += 3.0; y -= 3.0
    x, y = 33.0, 39.0
    for _ in range(4):
        obs.append((x, y, 3.0)); x += 3.0; y -= 3.0
    return obs


def run_planner_rrt(seed, rect, start, goal, obstacles, max_step, freq, cell, subs, exp_rate=1, dist_to_end=2,
                    diff_max=0.5, keep_points=True):
    mod, mpsm = import_gym_rrt()
    MPS = mpsm.Motion_plan_state
    clock = refstubs.VirtualClock()
    mod.time = clock
    obs = [MPS(o[0], o[1], size=o[2]) for o in obstacles]
    bnd = [MPS(rect[0], rect[1]), MPS(rect[2], rect[3])]
    s = MPS(start[0], start[1], z=-5.0, theta=start[2] if len(start) > 2 else 0.0)
    g = MPS(goal[0], goal[1], z=-5.0, theta=0.0)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            rrt = mod.Planner_RRT(s, g, bnd, obs, [], exp_rate=exp_rate, dist_to_end=dist_to_end, diff_max=diff_max,
                                  freq=freq, cell_side_length=cell, subsections_in_cell=subs)
    except IndexError as e:
        # add_node_to_grid(start) indexes env_grid with int(y / cell), int(x / cell) of the ABSOLUTE position (:115-127):
        # a start further below / left of 0 than the grid is tall / wide raises in __init__
        return {"seed": seed, "rect": np.array(rect, dtype=np.float64), "start": np.array(start, dtype=np.float64),
                "goal": np.array(goal, dtype=np.float64), "obstacles": np.array(obstacles, dtype=np.float64).reshape(-1, 3),
                "max_step": max_step, "freq": freq, "cell": cell, "subs": subs, "exp_rate": exp_rate,
                "dist_to_end": dist_to_end, "diff_max": diff_max, "error": "IndexError", "error_stage": "init",
                "error_text": str(e), "steps": 0}
    log = {"bucket": [], "picked": [], "accepted": [], "done": [], "npath": [], "arc_n": [], "arc_free": []}
    index_of = {id(s): 0}
    orig_gon = rrt.generate_one_node
    orig_steer = rrt.steer
    orig_cc = rrt.check_collision_free
    orig_conn = rrt.connect_to_goal_curve_alt
    ncols = len(rrt.env_grid[0])
    cur = {}

    def steer(m, *a, **k):
        for i, n in enumerate(rrt.mps_list):
            index_of.setdefault(id(n), i)
        cur["picked"] = index_of[id(m)]
        new = orig_steer(m, *a, **k)
        cur["npath"] = len(new.path)
        cur["phase"] = 0
        return new

    def cc(m, obs_):
        r = orig_cc(m, obs_)
        if cur.get("phase") == 0:
            cur["accepted"] = bool(r)
            cur["phase"] = 1
        else:
            cur["arc_free"] = bool(r)
        return r

    def conn(m, *a, **k):
        f = orig_conn(m, *a, **k)
        cur["arc_n"] = -1 if f is None else len(f.path)
        return f

    rrt.steer = steer
    rrt.check_collision_free = cc
    rrt.connect_to_goal_curve_alt = conn

    def gon(cellobj, step_num=None, min_length=250):
        # bucket id = (row * ncols + col) * S + sub
        bid = -1
        for r_, row in enumerate(rrt.env_grid):
            for c_, gc in enumerate(row):
                for k_, sub in enumerate(gc.subsection_cells):
                    if sub is cellobj:
                        bid = (r_ * ncols + c_) * subs + k_
        cur.clear()
        done, out = orig_gon(cellobj, step_num, min_length)
        log["bucket"].append(bid)
        log["picked"].append(cur["picked"])
        log["accepted"].append(cur["accepted"])
        log["npath"].append(cur["npath"])
        log["arc_n"].append(cur["arc_n"])
        log["arc_free"].append(cur.get("arc_free", False))
        log["done"].append(bool(done))
        return done, out

    rrt.generate_one_node = gon
    random.seed(seed)
    error = ("", "", "")
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            path, step, _ = rrt.planning(max_step=max_step)
    except IndexError as e:
        # planning(): random.choice of an empty occupied list (the start was "out of the habitat environment bound": :118-124,
        # :186), or add_node_to_grid of an accepted node whose index is below -len (:127; the node is in mps_list by then: :229)
        path, step = None, len(log["done"])
        error = ("IndexError", "planning", str(e))
    rng_after = random.random()
    nodes = rrt.mps_list
    for i, n in enumerate(nodes):
        index_of[id(n)] = i
    node_arr = np.array([[n.x, n.y, n.theta, n.traj_time_stamp, n.length] for n in nodes])
    parent = np.array([-1 if n.parent is None else index_of[id(n.parent)] for n in nodes], dtype=np.int32)
    npath = np.array([len(n.path) for n in nodes], dtype=np.int32)
    pts = []
    for n in nodes[1:]:
        for p in n.path[1:]:
            pts.append([p.x, p.y, p.theta, p.traj_time_stamp])
    pts = np.array(pts, dtype=np.float64).reshape(-1, 4)
    # the tuples hold the raw indexes, which are negative where the reference relied on Python's negative list indexing
    # (worlds whose origin is below / left of 0): the bucket they address is index % len
    nrows = len(rrt.env_grid)
    occ = np.array([((r_ % nrows) * ncols + c_ % ncols) * subs + k_ % subs for (r_, c_, k_) in rrt.occupied_grid_cells_array],
                   dtype=np.int32)
    counts = np.array([len(sub.node_array) for row in rrt.env_grid for gc in row for sub in gc.subsection_cells],
                      dtype=np.int32)
    done = bool(log["done"][-1]) if log["done"] else False
    out = {
        "seed": seed, "rect": np.array(rect, dtype=np.float64), "start": np.array(start, dtype=np.float64),
        "goal": np.array(goal, dtype=np.float64), "obstacles": np.array(obstacles, dtype=np.float64).reshape(-1, 3),
        "max_step": max_step, "freq": freq, "cell": cell, "subs": subs, "exp_rate": exp_rate,
        "dist_to_end": dist_to_end, "diff_max": diff_max,
        "steps": step, "done": done, "nodes": node_arr, "parent": parent, "npath": npath,
        "points_sha": sha(pts), "n_points": len(pts),
        "st_bucket": np.array(log["bucket"], dtype=np.int32), "st_picked": np.array(log["picked"], dtype=np.int32),
        "st_accepted": np.array(log["accepted"], dtype=np.int8), "st_done": np.array(log["done"], dtype=np.int8),
        "st_npath": np.array(log["npath"], dtype=np.int32), "st_arc_n": np.array(log["arc_n"], dtype=np.int32),
        "st_arc_free": np.array(log["arc_free"], dtype=np.int8),
        "occupied": occ, "bucket_counts": counts, "grid_rows": len(rrt.env_grid), "grid_cols": ncols,
        "rng_after": rng_after,
    }
    if error[0]:
        out["error"], out["error_stage"], out["error_text"] = error
        # the step the exception interrupted: its steer and first collision test had run (the wrappers saw them)
        out["err_picked"] = cur.get("picked", -1)
        out["err_accepted"] = bool(cur.get("accepted", False))
        out["err_npath"] = cur.get("npath", -1)
    if keep_points:
        out["points"] = pts
    if done:
        out["path"] = np.array([[p.x, p.y, p.theta, p.traj_time_stamp, p.length] for p in path])
    return out


def g2():
    lay = main_layout_obstacles()
    specs = []
    for seed in range(5):
        specs.append(("g2_main_s%d" % seed, seed, (0.0, 0.0, 50.0, 50.0), (10.0, 10.0), (35.0, 45.0), lay, 2000, 10, 5, 1,
                      {}))
    specs.append(("g2_main_subs8", 11, (0.0, 0.0, 50.0, 50.0), (10.0, 10.0, 0.7), (35.0, 45.0), lay, 1500, 50, 2, 8, {}))
    w = synth.make_rect_world(seed=3, n_obstacles=256)
    specs.append(("g2_o256_200m", 5, tuple(w["rect"].tolist()), tuple(w["start"].tolist()), tuple(w["goal"].tolist()),
                  [tuple(o) for o in w["obstacles"].tolist()], 2000, 10, 5, 1, {"keep_points": False}))
    w = synth.make_rect_world(seed=4, n_obstacles=64, size=100.0, start=(10.0, 10.0), goal=(85.0, 80.0),
                              obst_radius=(2.0, 5.0))
    specs.append(("g2_o64_100m", 6, tuple(w["rect"].tolist()), tuple(w["start"].tolist()), tuple(w["goal"].tolist()),
                  [tuple(o) for o in w["obstacles"].tolist()], 1500, 20, 5, 4, {"exp_rate": 0.5}))
    # round 6: rectangles whose origin is not (0, 0).  The bucket grid ignores the origin (:115-116: int(y / cell), int(x / cell)
    # of the absolute position), so nodes land in "wrong" cells, negative indexes wrap to the far end of the Python lists,
    # indexes past the grid return without inserting (:118-124: such nodes are in mps_list and in no bucket) and indexes
    # below -len raise IndexError.  g2_org_*: runs that complete; g2e_*: the reference raises (kept out of the g2_* glob).
    def shifted(dx, dy):
        return ((dx, dy, dx + 50.0, dy + 50.0), (10.0 + dx, 10.0 + dy), (35.0 + dx, 45.0 + dy),
                [(x + dx, y + dy, r) for x, y, r in lay])
    for name, seed, (dx, dy), max_step, freq, cell, subs, extra in (
            ("g2_org_p30_p20", 2, (30.0, 20.0), 2000, 10, 5, 1, {}),        # x >= 50 or y >= 50: not bucketed
            ("g2_org_m50_m30", 3, (-50.0, -30.0), 2000, 10, 5, 2, {}),      # every index negative: all wrap
            ("g2_org_frac", 4, (7.5, -3.25), 2000, 10, 5, 4, {}),           # int() truncates toward zero around y = 0
            ("g2_org_m20_p5", 5, (-20.0, 5.0), 1500, 20, 2, 8, {}),         # cell 2: 25 x 25 cells, columns wrap, rows run out
            ("g2e_init_m200", 1, (-200.0, -200.0), 50, 10, 5, 1, {}),       # IndexError in __init__
            ("g2e_empty_p100", 1, (100.0, 100.0), 50, 10, 5, 1, {}),        # start not bucketed: random.choice([]) at step 0
            ("g2e_mid_m55", 3, (-55.5, -55.5), 2000, 10, 5, 1, {}),          # a node below y = -55 or left of x = -55 raises mid-run
            ("g2e_mid_m55_late", 8, (-55.5, -55.5), 2000, 10, 5, 1, {})):
        rect, start, goal, obs = shifted(dx, dy)
        specs.append((name, seed, rect, start, goal, obs, max_step, freq, cell, subs, extra))
    for org, seed in (((-50.0, -30.0), 7), ((30.0, 20.0), 8)):
        w = synth.make_rect_world(seed=3, n_obstacles=256, origin=org)
        specs.append(("g2_org_o256_%s" % ("m50_m30" if org[0] < 0 else "p30_p20"), seed, tuple(w["rect"].tolist()),
                      tuple(w["start"].tolist()), tuple(w["goal"].tolist()), [tuple(o) for o in w["obstacles"].tolist()],
                      2000, 10, 5, 1, {"keep_points": False}))
    only = os.environ.get("AUVP_G2_ONLY")
    for name, seed, rect, start, goal, obs, max_step, freq, cell, subs, extra in specs:
        if only and name not in only.split(","):
            continue
        out = run_planner_rrt(seed, rect, start, goal, obs, max_step, freq, cell, subs, **extra)
        save_npz(name + ".npz", **out)
        print(name, "steps", out["steps"], "done", out.get("done"), "nodes", len(out.get("nodes", ())),
              "path", None if "path" not in out else out["path"].shape, "error", out.get("error", ""),
              out.get("error_stage", ""), out.get("error_text", ""),
              "unbucketed", None if "nodes" not in out else len(out["nodes"]) - int(np.sum(out["bucket_counts"])))


# --------------------------------------------------------------------------------------------
# G1 / G6: the A* variants (path_planning/astar.py, astar_real.py, astar_fixLen.py, astar_fixLenSOG.py)
# --------------------------------------------------------------------------------------------
def _log_expansions(solver, log):
    """every expansion calls curr_neighbors(current_node, ...) once: record the popped node there"""
    orig = solver.curr_neighbors

    def wrapped(node, *a):
        log.append([float(node.position[0]), float(node.position[1]), float(node.g), float(node.h), float(node.f),
                    float(getattr(node, "cost", 0.0)), float(getattr(node, "pathLen", 0.0)),
                    float(getattr(node, "time_stamp", 0.0))])
        return orig(node, *a)

    solver.curr_neighbors = wrapped


def run_astar_basic(world, start, goal):
    mod, mpsm = import_astar("astar")
    MPS = mpsm.Motion_plan_state
    obs = [MPS(o[0], o[1], size=o[2]) for o in world["obstacles"].tolist()]
    box = world["box"].tolist()
    bnd = [MPS(box[0], box[1]), MPS(box[2], box[3])]
    solver = mod.astar(start, goal, obs, bnd)
    log = []
    _log_expansions(solver, log)
    with contextlib.redirect_stdout(io.StringIO()):
        path = solver.astar(obs, start, goal)
    return {"variant": "astar", "obstacles": world["obstacles"], "box": world["box"], "start": np.array(start, dtype=np.float64),
            "goal": np.array(goal, dtype=np.float64), "found": path is not None,
            "path": np.array([[p.x, p.y] for p in (path or [])], dtype=np.float64).reshape(-1, 2),
            "expansions": np.array(log, dtype=np.float64).reshape(-1, 8)}


def run_astar_real(obstacles, polygon, start, goal):
    mod, mpsm = import_astar("astar_real")
    MPS = mpsm.Motion_plan_state
    obs = [MPS(o[0], o[1], size=o[2]) for o in obstacles]
    bnd = [MPS(p[0], p[1]) for p in polygon]
    solver = mod.astar(start, goal, obs, bnd)
    log = []
    _log_expansions(solver, log)
    with contextlib.redirect_stdout(io.StringIO()):
        path = solver.astar(obs, bnd)
    return {"variant": "astar_real", "obstacles": np.array(obstacles, dtype=np.float64).reshape(-1, 3),
            "polygon": np.array(polygon, dtype=np.float64), "start": np.array(start, dtype=np.float64),
            "goal": np.array(goal, dtype=np.float64), "found": path is not None,
            "path": np.array([[p.x, p.y] for p in (path or [])], dtype=np.float64).reshape(-1, 2),
            "expansions": np.array(log, dtype=np.float64).reshape(-1, 8)}


def run_astar_fixlen(obstacles, habitats, polygon, start, limit, weights):
    mod, mpsm = import_astar("astar_fixLen")
    MPS = mpsm.Motion_plan_state
    obs = [MPS(o[0], o[1], size=o[2]) for o in obstacles]
    hab = [MPS(h[0], h[1], size=h[2]) for h in habitats]
    bnd = [MPS(p[0], p[1]) for p in polygon]
    solver = mod.astar(start, obs, bnd)
    log = []
    _log_expansions(solver, log)
    with contextlib.redirect_stdout(io.StringIO()):
        res = solver.astar(hab, obs, bnd, start, limit, list(weights))
    out = {"variant": "astar_fixLen", "obstacles": np.array(obstacles, dtype=np.float64).reshape(-1, 3),
           "habitats": np.array(habitats, dtype=np.float64).reshape(-1, 3), "polygon": np.array(polygon, dtype=np.float64),
           "start": np.array(start, dtype=np.float64), "limit": float(limit), "weights": np.array(weights, dtype=np.float64),
           "found": res is not None, "expansions": np.array(log, dtype=np.float64).reshape(-1, 8),
           "habitats_left": np.array([[h.x, h.y, h.size] for h in hab], dtype=np.float64).reshape(-1, 3),
           "visited_count": int(solver.visited_nodes.sum())}
    if res is not None:
        out["path"] = np.array([[p.x, p.y] for p in res[0]], dtype=np.float64).reshape(-1, 2)
        out["cost_list"] = np.array([float(c) for c in res[1]], dtype=np.float64)
    return out


def run_astar_sog(world, start, limit, weights, velocity=1):
    mod, mpsm = import_astar("astar_fixLenSOG")
    MPS = mpsm.Motion_plan_state
    obstacles, habitats, poly, cell_list, shark = ref_world(world, MPS)
    mod.splitCell = lambda polygon, size: cell_list  # shapely.ops.split is absent; cells come from the world
    bnd = [MPS(p[0], p[1]) for p in world["polygon"].tolist()]
    with contextlib.redirect_stdout(io.StringIO()):
        solver = mod.astar(start, obstacles, bnd, habitats, shark, {}, velocity)
    log = []
    _log_expansions(solver, log)
    with contextlib.redirect_stdout(io.StringIO()):
        res = solver.astar(limit, list(weights), {})
    out = dict(world_arrays(world))
    out.update({"variant": "astar_fixLenSOG", "start": np.array(start, dtype=np.float64), "limit": float(limit),
                "weights": np.array(weights, dtype=np.float64), "velocity": float(velocity), "found": res is not None,
                "expansions": np.array(log, dtype=np.float64).reshape(-1, 8),
                "visited_count": int(solver.visited_nodes.sum())})
    if res is not None:
        out["path_length"] = int(res["path length"])
        out["path"] = np.array([[p.x, p.y, p.traj_time_stamp] for p in res["path"]], dtype=np.float64).reshape(-1, 3)
        out["cost"] = float(res["cost"])
        out["cost_list"] = np.array([float(c) for c in res["cost list"]], dtype=np.float64)
        out["node_path"] = np.array([[n.position[0], n.position[1], n.g, n.h, n.f, n.cost, n.pathLen, n.time_stamp]
                                     for n in res["node"]], dtype=np.float64).reshape(-1, 8)
    return out


def g1():
    # config 1: 50x50 lattice, 10 obstacles, (0,0) -> (490,490)
    w = synth.make_lattice_world(seed=0, n_obstacles=10)
    save_npz("g1_astar_cfg1.npz", **run_astar_basic(w, (0, 0), (490, 490)))
    for k, (seed, nobs, st, gl) in enumerate([(1, 25, (0, 0), (490, 490)), (2, 40, (0, 490), (490, 0)),
                                              (3, 10, (250, 250), (250, 260)), (4, 60, (0, 0), (300, 470))]):
        w = synth.make_lattice_world(seed=seed, n_obstacles=nobs, r_range=(10, 25))
        out = run_astar_basic(w, st, gl)
        save_npz("g1_astar_%d.npz" % k, **out)
        print("g1", k, "found", out["found"], "path", len(out["path"]), "expansions", len(out["expansions"]))
    # float-valued start (round(., 2) style) in a polygon world: astar_real
    rect = [(-300.0, -100.0), (-100.0, -100.0), (-100.0, 100.0), (-300.0, 100.0)]
    penta = [(-300.0, -100.0), (-120.0, -120.0), (-80.0, 20.0), (-180.0, 110.0), (-320.0, 60.0)]
    for k, (seed, poly, st, gl) in enumerate([(5, rect, (-290.0, -90.0), (-110.0, 90.0)),
                                              (6, penta, (-285.37, -82.11), (-125.37, 57.89)),
                                              (7, rect, (-200.5, 0.25), (-120.5, -79.75))]):
        w = synth.make_world(seed=seed, n_obstacles=24, obst_radius=(3.0, 9.0), start=st)
        obs = [o for o in w["obstacles"].tolist() if (o[0] - gl[0]) ** 2 + (o[1] - gl[1]) ** 2 > (o[2] + 12) ** 2]
        out = run_astar_real(obs, poly, st, gl)
        save_npz("g1_real_%d.npz" % k, **out)
        print("g1 real", k, "found", out["found"], "path", len(out["path"]), "expansions", len(out["expansions"]))


def g6():
    rect = [(-300.0, -100.0), (-100.0, -100.0), (-100.0, 100.0), (-300.0, 100.0)]
    penta = [(-300.0, -100.0), (-120.0, -120.0), (-80.0, 20.0), (-180.0, 110.0), (-320.0, 60.0)]
    notch = [tuple(p) for p in synth.make_world(seed=0, n_obstacles=0, polygon="notch")["polygon"].tolist()]  # (round 6: concave)
    specs = [(21, rect, (-290.0, -90.0), 200, (0, 10, 10), 12), (22, penta, (-280.0, -80.0), 400, (0, 10, 10), 12),
             (23, rect, (-200.0, 0.0), 300, (0, 3.5, 1.25), 20), (24, rect, (-290.5, -90.25), 150, (0, 10, 10), 8),
             (25, notch, (-200.0, -10.0), 300, (0, 10, 10), 12)]
    for k, (seed, poly, st, limit, wts, nobs) in enumerate(specs):
        w = synth.make_world(seed=seed, n_obstacles=nobs, obst_radius=(3.0, 8.0), start=st, n_habitats=8,
                             hab_radius=(10.0, 25.0))
        out = run_astar_fixlen(w["obstacles"].tolist(), w["habitats"].tolist(), poly, st, limit, wts)
        save_npz("g6_fixlen_%d.npz" % k, **out)
        print("g6 fixlen", k, "found", out["found"], "path", len(out.get("path", [])), "expansions", len(out["expansions"]),
              "habitats left", len(out["habitats_left"]))
    # round 6: 3 and 4 = astar_fixLenSOG on boundaries that are not rectangles.  within_bounds (:178-199) is a fan of triangles
    # around the polygon's CENTROID: exact for the convex Catalina outline, reference-specific for the concave one (the wedge
    # cut into the top edge is partly "inside", parts of the polygon far from the centroid are not): the search must follow it
    sog = [(31, (-290.0, -90.0), 200, (0, 10, 10, 100), 16, None), (32, (-200.0, 0.0), 300, (0, 10, 10, 100), 16, None),
           (33, (-290.0, 90.0), 100, (0, 2.5, 1.5, 40.5), 64, None),
           (34, (-200.0, 0.0), 100, (0, 10, 10, 100), 16, "catalina"), (35, (-200.0, -10.0), 300, (0, 10, 10, 100), 16, "notch"),
           (36, (-200.0, 0.0), 100, (0, 10, 10, 100), 16, "notch")]
    for k, (seed, st, limit, wts, nobs, polygon) in enumerate(sog):
        w = synth.make_world(seed=seed, n_obstacles=nobs, obst_radius=(2.0, 6.0), start=st, n_habitats=8,
                             hab_radius=(10.0, 25.0), polygon=polygon)
        out = run_astar_sog(w, st, limit, wts)
        save_npz("g6_sog_%d.npz" % k, **out)
        print("g6 sog", k, "found", out["found"], "path", out.get("path_length"), "expansions", len(out["expansions"]))


def g6b():
    """two astar() calls on ONE astar_fixLen solver: self.visited_nodes carries over (:51,:414-416)"""
    mod, mpsm = import_astar("astar_fixLen")
    MPS = mpsm.Motion_plan_state
    rect = [(-300.0, -100.0), (-100.0, -100.0), (-100.0, 100.0), (-300.0, 100.0)]
    w = synth.make_world(seed=26, n_obstacles=10, obst_radius=(3.0, 8.0), n_habitats=8, hab_radius=(10.0, 25.0))
    obs = [MPS(o[0], o[1], size=o[2]) for o in w["obstacles"].tolist()]
    bnd = [MPS(p[0], p[1]) for p in rect]
    solver = mod.astar((-290.0, -90.0), obs, bnd)
    outs = {}
    for k, (st, limit) in enumerate([((-290.0, -90.0), 150), ((-250.0, -60.0), 120)]):
        hab = [MPS(h[0], h[1], size=h[2]) for h in w["habitats"].tolist()]
        log = []
        solver.curr_neighbors = mod.astar.curr_neighbors.__get__(solver)
        _log_expansions(solver, log)
        with contextlib.redirect_stdout(io.StringIO()):
            res = solver.astar(hab, obs, bnd, st, limit, [0, 10, 10])
        outs["found%d" % k] = res is not None
        outs["start%d" % k] = np.array(st)
        outs["limit%d" % k] = float(limit)
        outs["expansions%d" % k] = np.array(log, dtype=np.float64).reshape(-1, 8)
        outs["path%d" % k] = np.array([[p.x, p.y] for p in (res[0] if res else [])], dtype=np.float64).reshape(-1, 2)
        outs["visited_count%d" % k] = int(solver.visited_nodes.sum())
    save_npz("h6_fixlen_twice.npz", obstacles=w["obstacles"], habitats=w["habitats"], polygon=np.array(rect),
             weights=np.array([0.0, 10.0, 10.0]), **outs)
    print("g6b", outs["found0"], outs["found1"], len(outs["expansions0"]), len(outs["expansions1"]), outs["visited_count0"],
          outs["visited_count1"])


# --------------------------------------------------------------------------------------------
# G8: the caller of Planner_RRT -- gym_rrt/envs/rrt_env.py RRTEnv.reset / step
# --------------------------------------------------------------------------------------------
def g8():
    refstubs.install()
    _purge({"gym_rrt"})
    saved = list(sys.path)
    sys.path[:0] = [REF]
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            envmod = importlib.import_module("gym_rrt.envs.rrt_env")
            mpsm = importlib.import_module("gym_rrt.envs.motion_plan_state_rrt")
            planmod = importlib.import_module("gym_rrt.envs.rrt_dubins")
    finally:
        sys.path[:] = saved
    MPS = mpsm.Motion_plan_state
    planmod.time = refstubs.VirtualClock()
    lay = main_layout_obstacles()
    # round 6: `policy` "planning" = the agent plays Planner_RRT.planning's own rule (random.choice of the occupied list, from the
    # same global stream: :186), so the run ends with a found path -- state["path"] as a LIST whose elements carry the
    # rl_state_id of the step that created them (solveRL-RRT.py:981-1006 reads those) -- and `origin` translates the world
    # (negative bucket indexes in the environment's flat index arithmetic: rrt_env.py:203-213 sees the wrapped cell)
    for name, seed, cell, subs, n_steps, (dx, dy), policy in (
            ("g8_env_s1", 5, 5, 1, 700, (0.0, 0.0), "cycle"), ("g8_env_s8", 6, 5, 4, 400, (0.0, 0.0), "cycle"),
            ("g8_env_plan_s1", 1, 5, 1, 1500, (0.0, 0.0), "planning"), ("g8_env_org_m50_m30", 3, 5, 2, 1500, (-50.0, -30.0), "planning"),
            ("g8_env_org_p30_p20", 9, 5, 4, 500, (30.0, 20.0), "cycle")):
        with contextlib.redirect_stdout(io.StringIO()):
            env = envmod.RRTEnv()
            auv = MPS(x=10.0 + dx, y=10.0 + dy, z=-5.0, theta=0.0)
            shark = MPS(x=35.0 + dx, y=45.0 + dy, z=-5.0, theta=0.0)
            obs = [MPS(x=o[0] + dx, y=o[1] + dy, size=o[2]) for o in lay]
            bnd = [MPS(x=dx, y=dy), MPS(x=50.0 + dx, y=50.0 + dy)]
            random.seed(seed)
            st = env.init_env(auv, shark, bnd, cell, subs, obs)
        grid0 = np.array(st["rrt_grid"], dtype=np.float64)
        choices, rewards, dones, counts = [], [], [], []
        path_kind, node_rec = [], []   # what step() returned as `path`: 0 None (state["path"] keeps its value), 1 a node, 2 a list
        done = False
        ncols_env = len(env.rrt_planner.env_grid[0])
        nrows_env = len(env.rrt_planner.env_grid)
        for i in range(n_steps):
            has = np.flatnonzero(np.asarray(env.state["has_node"]))
            if policy == "planning":
                r_, c_, k_ = random.choice(env.rrt_planner.occupied_grid_cells_array)
                idx = ((r_ % nrows_env) * ncols_env + c_ % ncols_env) * subs + k_ % subs
            else:
                # deterministic policy: cycle through the occupied flat indices, every 7th step an empty one
                idx = int(has[i % len(has)])
            if policy == "cycle" and i % 7 == 3:
                empt = np.flatnonzero(np.asarray(env.state["has_node"]) == 0)
                idx = int(empt[(i * 13) % len(empt)])
                # the reference blocks on input() for an empty cell (rrt_dubins.py:219): answer it
                import builtins
                old_input = builtins.input
                builtins.input = lambda *a: ""
            with contextlib.redirect_stdout(io.StringIO()):
                s, r, done, _ = env.step(idx, i)
            if policy == "cycle" and i % 7 == 3:
                builtins.input = old_input
            choices.append(idx); rewards.append(r); dones.append(bool(done))
            counts.append(np.asarray(s["rrt_grid_num_of_nodes_only"], dtype=np.int64).copy())
            # the `path` of this step: generate_one_node's second value, which state["path"] takes unless it is None (:233-234)
            pth = s["path"]
            if r == envmod.R_INVALID_NODE:
                path_kind.append(0); node_rec.append([np.nan] * 5)
            elif isinstance(pth, list):
                path_kind.append(2); node_rec.append([np.nan] * 5)
            else:
                path_kind.append(1)
                node_rec.append([pth.x, pth.y, pth.theta, pth.traj_time_stamp, -1 if pth.rl_state_id is None else pth.rl_state_id])
            if done:
                break
        last = env.state
        path = last["path"]
        out = {"seed": seed, "cell": cell, "subs": subs, "obstacles": np.array([(o[0] + dx, o[1] + dy, o[2]) for o in lay], dtype=np.float64),
               "rect": np.array([dx, dy, 50.0 + dx, 50.0 + dy]), "start": np.array([10.0 + dx, 10.0 + dy]),
               "goal": np.array([35.0 + dx, 45.0 + dy]), "policy": policy,
               "path_kind": np.array(path_kind, dtype=np.int8), "node_rec": np.array(node_rec, dtype=np.float64).reshape(-1, 5),
               "freq": envmod.RRT_PLANNER_FREQ, "rrt_grid0": grid0, "choices": np.array(choices, dtype=np.int64),
               "rewards": np.array(rewards, dtype=np.int64), "dones": np.array(dones, dtype=np.int8),
               "counts": np.array(counts, dtype=np.int64), "final_rrt_grid": np.array(last["rrt_grid"], dtype=np.float64),
               "final_has_node": np.array(last["has_node"], dtype=np.int64), "rng_after": random.random(),
               "done": bool(done)}
        if done and isinstance(path, list):
            out["path"] = np.array([[p.x, p.y, p.theta, p.traj_time_stamp] for p in path], dtype=np.float64)
            out["path_state_id"] = np.array([-1 if p.rl_state_id is None else p.rl_state_id for p in path], dtype=np.int64)
            out["path0_length"] = float(path[0].length)
            out["cal_length"] = float(env.rrt_planner.cal_length(path))  # (solveRL-RRT.py:1343)
        save_npz(name + ".npz", **out)
        print(name, "steps", len(choices), "done", done, "rewards", {int(k): int((np.array(rewards) == k).sum()) for k in set(rewards)})


# --------------------------------------------------------------------------------------------
# G9: SharkOccupancyGrid.convert (path_planning/sharkOccupancyGrid.py:47-74) -- SURVEY 8(f) f2
# --------------------------------------------------------------------------------------------
def g9():
    refstubs.install()
    _purge(_SHARED)
    saved = list(sys.path)
    sys.path[:0] = [os.path.join(REF, "path_planning"), REF]
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            sog = importlib.import_module("sharkOccupancyGrid")
            mpsm = importlib.import_module("motion_plan_state")
    finally:
        sys.path[:] = saved
    MPS = mpsm.Motion_plan_state
    rng = random.Random(91)
    specs = [("g9_sog_small", (0.0, 0.0, 10.0, 10.0), 2.0, 2.0, 4.0, 1, 50, 0.1),
             ("g9_sog_catalina", (-300.0, -100.0, -100.0, 100.0), 10.0, 50.0, 50.0, 6, 240, 1.0),
             ("g9_sog_fine", (-40.0, -30.0, 35.0, 42.0), 5.0, 20.0, 12.0, 4, 150, 0.5)]
    for name, box, cs, bin_interval, detect, n_sharks, n_pts, dt in specs:
        x0, y0, x1, y1 = box
        ncol, nrow = int(round((x1 - x0) / cs)), int(round((y1 - y0) / cs))
        cells = [refstubs.CellStub(x0 + c * cs, y0 + r * cs, x0 + (c + 1) * cs, y0 + (r + 1) * cs)
                 for r in range(nrow) for c in range(ncol)]
        boundary = refstubs.Polygon([(x0, y0), (x1, y0), (x1, y1), (x0, y1)])
        sharks = {}
        for s in range(1, n_sharks + 1):
            px, py = rng.uniform(x0 + 1, x1 - 1), rng.uniform(y0 + 1, y1 - 1)
            vx, vy = rng.uniform(-0.6, 0.6), rng.uniform(-0.6, 0.6)
            traj = []
            for i in range(1, n_pts + 1):
                px = min(max(px + vx, x0), x1)
                py = min(max(py + vy, y0), y1)
                if rng.random() < 0.05:
                    vx, vy = rng.uniform(-0.6, 0.6), rng.uniform(-0.6, 0.6)
                if rng.random() < 0.03:  # exactly on a cell edge / corner
                    px = x0 + cs * round((px - x0) / cs)
                traj.append(MPS(px, py, traj_time_stamp=dt * i))
            sharks[s] = traj
        grid = sog.SharkOccupancyGrid(cs, boundary, bin_interval, detect, cells)
        with contextlib.redirect_stdout(io.StringIO()):
            arr, celld = grid.convert(sharks)
        keys = list(arr.keys())
        occ0 = np.array(grid.constructSharkOccupancyGrid(grid.timeBinDict[keys[0]][1]), dtype=np.float64)
        with contextlib.redirect_stdout(io.StringIO()):
            auv0 = np.array(grid.constructAUVGrid(occ0.tolist()), dtype=np.float64)
        flat = []
        for s in range(1, n_sharks + 1):
            for p in sharks[s]:
                flat.append([p.x, p.y, p.traj_time_stamp])
        cell_bounds = np.array([c.bounds for c in cells], dtype=np.float64)
        save_npz(name + ".npz", box=np.array(box), cell_size=cs, bin_interval=bin_interval, detect_range=detect,
                 cells=cell_bounds, points=np.array(flat, dtype=np.float64),
                 traj_len=np.array([len(sharks[s]) for s in range(1, n_sharks + 1)], dtype=np.int32),
                 bins=np.array([[k[0], k[1]] for k in keys], dtype=np.float64),
                 grids=np.array([arr[k] for k in keys], dtype=np.float64), occ_bin0_shark1=occ0, auv_bin0_shark1=auv0,
                 cell_keys=np.array([[list(b) for b in celld[k].keys()] for k in keys], dtype=np.float64),
                 cell_vals=np.array([list(celld[k].values()) for k in keys], dtype=np.float64))
        print(name, "bins", len(keys), "grid", np.array(arr[keys[0]]).shape, "cells", len(cells),
              "nonzero per bin", [len(celld[k]) for k in keys][:4])


# --------------------------------------------------------------------------------------------
# G10: particleFilter.py (ParticleFilter.create / create_and_update / update_weights / particleMean /
# meanError) driven the way robotSim.py:665-701 drives it, numpy's global legacy RandomState seeded
# --------------------------------------------------------------------------------------------
def g10():
    import types
    refstubs.install()
    _purge(_SHARED | {"particleFilter", "live3DGraph", "twoDfigure"})
    # plotting-only imports of particleFilter.py:14-15
    sys.modules["live3DGraph"] = types.ModuleType("live3DGraph")
    sys.modules["live3DGraph"].Live3DGraph = object
    sys.modules["twoDfigure"] = types.ModuleType("twoDfigure")
    sys.modules["twoDfigure"].Figure = object
    saved = list(sys.path)
    sys.path[:0] = [REF]
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            pfm = importlib.import_module("particleFilter")
    finally:
        sys.path[:] = saved
    orig_choice = np.random.choice
    specs = [("g10_pf_one_auv", 11, 1, 14, (40.0, -25.0)), ("g10_pf_two_auvs", 12, 2, 14, (-310.0, 120.0)),
             ("g10_pf_three_auvs_far", 4000000123, 3, 10, (900.0, 650.0))]
    for name, seed, n_auv, n_steps, shark0 in specs:
        rng = random.Random(seed * 7 + 1)
        np.random.seed(seed)
        choice_log = []

        def logged_choice(a, *args, **kw):
            x = orig_choice(a, *args, **kw)
            choice_log[-1][1].append(int(x))
            return x
        pfm.random.choice = logged_choice  # numpy.random module attribute, restored below
        try:
            pf = pfm.ParticleFilter(shark0[0], shark0[1], [])
            particles = pf.create()

            def snap(ps):
                return np.array([[p.x_p, p.y_p, p.v_p, p.theta_p, p.weight_p] for p in ps], dtype=np.float64)
            created = snap(particles)
            sx, sy, sth = shark0[0], shark0[1], rng.uniform(-math.pi, math.pi)
            auv = [[shark0[0] + rng.uniform(-120, 120), shark0[1] + rng.uniform(-120, 120), rng.uniform(-math.pi, math.pi)]
                   for _ in range(n_auv)]
            upd, new, meas_all, means, errs, alias, ells = [], [], [], [], [], [], []
            for step in range(n_steps):
                with contextlib.redirect_stdout(io.StringIO()):
                    particles = pf.create_and_update(particles)
                upd.append(snap(particles))
                sth += rng.uniform(-0.3, 0.3)
                sx += 1.2 * math.cos(sth)
                sy += 1.2 * math.sin(sth)
                meas = []
                for a in auv:
                    a[2] += rng.uniform(-0.2, 0.2)
                    a[0] += 1.0 * math.cos(a[2])
                    a[1] += 1.0 * math.sin(a[2])
                    z_range = math.hypot(sx - a[0], sy - a[1]) + rng.gauss(0, 5)
                    z_bearing = pfm.angle_wrap(math.atan2(sy - a[1], sx - a[0]) - a[2]) + rng.gauss(0, 0.1)
                    # robotSim.py:get_all_sharks_sensor_measurements row: x, y, theta, range, bearing, id
                    meas.append([a[0], a[1], a[2], z_range, z_bearing, 1])
                pf.x_shark, pf.y_shark = sx, sy
                choice_log.append((step, []))
                with contextlib.redirect_stdout(io.StringIO()):
                    particles = pf.update_weights(particles, meas)
                    m = pf.particleMean(particles)
                    e = pf.meanError(m[0], m[1])
                new.append(snap(particles))
                first = {}
                alias.append([first.setdefault(id(p), i) for i, p in enumerate(particles)])
                meas_all.append([r[:5] for r in meas])
                means.append(m)
                errs.append(e)
                ells.append([sx, sy])
            st = np.random.get_state()
        finally:
            pfm.random.choice = orig_choice
        save_npz(name + ".npz", seed=np.uint64(seed), shark0=np.array(shark0), created=created,
                 updated=np.array(upd), resampled=np.array(new), measurements=np.array(meas_all, dtype=np.float64),
                 shark_xy=np.array(ells), mean=np.array(means), range_error=np.array(errs),
                 alias_first=np.array(alias, dtype=np.int32),
                 choice=np.array([c[1] for c in choice_log], dtype=np.int32),
                 mt_key=np.asarray(st[1], dtype=np.uint32), mt_pos=np.int32(st[2]))
        print(name, "steps", n_steps, "err", [round(x, 2) for x in errs[:3]], "...", round(errs[-1], 2),
              "max multiplicity", max(np.bincount(a).max() for a in alias))


# --------------------------------------------------------------------------------------------
# G11: habitatGrid.py HabitatGrid (habitat-id layout of the cell grid, inside_habitat, within_habitat_env)
# --------------------------------------------------------------------------------------------
def g11():
    saved = list(sys.path)
    sys.path[:0] = [REF]
    try:
        _purge({"habitatGrid", "habitatCell", "habitat"})
        hg = importlib.import_module("habitatGrid")
    finally:
        sys.path[:] = saved
    rng = random.Random(5)
    out = {}
    for k, (ex, ey, sx, sy, hs, cs) in enumerate([(0, 0, 50, 40, 10, 1), (0.0, 0.0, 60.0, 60.0, 20, 5), (5, 3, 35, 25, 10, 2),
                                                   (0, 0, 500, 500, 50, 10)]):
        g = hg.HabitatGrid(ex, ey, sx, sy, habitat_side_length=hs, cell_side_length=cs)
        ids = np.array([[c.habitat_id for c in row] for row in g.habitat_cell_grid], dtype=np.int32)
        cell_xy = np.array([[[c.x, c.y] for c in row] for row in g.habitat_cell_grid], dtype=np.float64)
        habs = np.array([[h.x, h.y, h.id, h.side_length] for h in g.habitat_array], dtype=np.float64)
        q = np.array([[rng.uniform(0, ex + sx + 8), rng.uniform(0, ey + sy + 8)] for _ in range(200)])
        with contextlib.redirect_stdout(io.StringIO()):
            inside = [g.inside_habitat(p) for p in q]
            within = [g.within_habitat_env(p) for p in q]
        out[f"c{k}_args"] = np.array([ex, ey, sx, sy, hs, cs], dtype=np.float64)
        out[f"c{k}_ids"] = ids
        out[f"c{k}_cell_xy"] = cell_xy
        out[f"c{k}_habitats"] = habs
        out[f"c{k}_query"] = q
        out[f"c{k}_inside_id"] = np.array([(-1 if c is False else c.habitat_id) for c in inside], dtype=np.int32)
        out[f"c{k}_within"] = np.array(within, dtype=np.int8)
    save_npz("g11_habitat_grid.npz", **out)


# --------------------------------------------------------------------------------------------
# G12: the other cost entry points of SURVEY 8(b): root cost.py Cost.habitat_shark_cost_func (4 weights,
# stale-bin behaviour) and Cost.cost_of_edge; path_planning/cost.py habitat_shark_cost_point
# --------------------------------------------------------------------------------------------
def g12():
    import types
    refstubs.install()
    _purge(_SHARED)
    saved = list(sys.path)
    sys.path[:0] = [REF]
    try:
        root_cost = importlib.import_module("cost")
        root_mps = importlib.import_module("motion_plan_state")
    finally:
        sys.path[:] = saved
    _purge(_SHARED)
    _, mpsm, pp_cost = import_rrt()
    MPS = root_mps.Motion_plan_state
    rng = random.Random(1212)
    cal = root_cost.Cost()
    twin, edges, points = [], [], []
    for k in range(16):
        world = synth.make_world(seed=300 + k, n_obstacles=3, n_habitats=(0 if k == 3 else 5 + k % 4),
                                 cell=10.0 if k % 2 else 20.0, n_bins=3 + k % 4)
        _, habitats, _, _, shark = ref_world(world, MPS)
        x0, y0, x1, y1 = world["box"].tolist()
        T = len(world["bins"])
        pts = []
        for i in range(rng.randint(2, 90)):
            x = rng.uniform(x0 - 10, x1 + 10)
            y = rng.uniform(y0 - 10, y1 + 10)
            if rng.random() < 0.15:
                x = float(round(x / 10.0) * 10.0)
            # the first point must fall into a bin (the reference reads an unbound local otherwise);
            # later ones may fall outside every bin and then reuse the previous point's bin
            t = rng.uniform(0.0, 50.0 * T) if (i == 0 or rng.random() < 0.7) else rng.uniform(50.0 * T + 1, 50.0 * T + 90)
            if rng.random() < 0.1:
                t = float(50 * rng.randint(0, T))
            pts.append((x, y, t))
        weights = rng.choice([[1, -3, -3, -4], [0.5, -1, -1, -1], [2, -0.5, -2.25, -1.75]])
        length, peri, total = rng.uniform(10, 900), rng.uniform(400, 3000), rng.uniform(1.0, 500.0)
        path = [MPS(p[0], p[1], traj_time_stamp=p[2]) for p in pts]
        res = cal.habitat_shark_cost_func(path, length, peri, total, habitats, shark, weights)
        twin.append({"world_seed": 300 + k, "n_habitats": len(habitats), "cell": 10.0 if k % 2 else 20.0, "n_bins": T,
                     "pts": pts, "length": length, "peri": peri, "total": total, "weights": weights,
                     "out": [float(res[0])] + [float(c) for c in res[1]]})
        # cost_of_edge: nodes around the habitats, a split of the habitat list into open / closed
        cut = rng.randint(0, len(habitats))
        open_l, closed_l = habitats[:cut], habitats[cut:]
        for _ in range(12):
            if habitats and rng.random() < 0.6:
                h = rng.choice(habitats)
                pos = (h.x + rng.uniform(-1.2, 1.2) * h.size, h.y + rng.uniform(-1.2, 1.2) * h.size)
            else:
                pos = (rng.uniform(x0, x1), rng.uniform(y0, y1))
            w3 = [rng.choice([0, 1, 10]), rng.choice([1, 10, 2.5]), rng.choice([1, 10, 0.75])]
            node = types.SimpleNamespace(position=pos)
            r = cal.cost_of_edge(node, open_l, closed_l, w3)
            edges.append({"world_seed": 300 + k, "n_habitats": len(habitats), "cell": 10.0 if k % 2 else 20.0, "n_bins": T,
                          "cut": cut, "pos": list(pos), "weights": w3, "out": [float(r[0]), int(r[1]), int(r[2])]})
        # habitat_shark_cost_point along the same points (visited list threaded through like performance.py:272)
        if habitats:
            visited = [False for _ in habitats]
            grid = shark[list(shark.keys())[k % T]]
            w3 = rng.choice([[-3, -3, -4], [-1, -1, -1]])
            outs = []
            for p in path[:40]:
                c, visited = pp_cost.habitat_shark_cost_point(p, habitats, visited, grid, w3)
                outs.append(float(c))
            points.append({"world_seed": 300 + k, "n_habitats": len(habitats), "cell": 10.0 if k % 2 else 20.0, "n_bins": T,
                           "bin": k % T, "pts": pts[:40], "weights": w3, "out": outs})
    with open(os.path.join(OUT, "g12_cost_twins.json"), "w") as f:
        json.dump({"twin": twin, "edges": edges, "points": points}, f)
    print("wrote g12_cost_twins.json", len(twin), len(edges), len(points))


# --------------------------------------------------------------------------------------------
# G13: RRT.replanning (rrt_dubins.py:51-90): receding-horizon rounds of exploring on ONE RRT object
# (time_bin persists, habitats are removed between rounds, every round starts at a non-zero
# traj_time_stamp), global `random` stream seeded once
# --------------------------------------------------------------------------------------------
class ReplanClock:
    """replanning passes plot_interval = 0.5 to exploring, so the unit-tick VirtualClock would allow one
    iteration.  This clock restarts at 0 for every exploring invocation (new frame) and advances by `tick`
    per while-test (calls from the exploring frame after the t_start read); other frames read the
    current value.  The iteration count per round is whatever the reference's loop then does
    (it is recorded and becomes the build's max_iter)."""

    def __init__(self, tick):
        self.tick, self.frame_id, self.q, self.now = tick, None, 0, 0.0

    def time(self):
        f = sys._getframe(1)
        if f.f_code.co_name == "exploring":
            if self.frame_id is not f:
                self.frame_id, self.q = f, 0
            self.now = self.tick * max(0, self.q - 1)
            self.q += 1
        return self.now

    def sleep(self, s):
        pass


def g13():
    rrt_mod, mpsm, cost_mod = import_rrt()
    MPS = mpsm.Motion_plan_state
    specs = [("g13_replan_a", 5, dict(seed=1, n_obstacles=64), 2.0, 150.0, 98.0, 1.0 / 256),
             ("g13_replan_b", 21, dict(seed=4, n_obstacles=64, n_habitats=8), 1.5, 120.0, 78.5, 1.0 / 512)]
    for name, seed, wk, budget, traj_len, interval, tick in specs:
        world = synth.make_world(**wk)
        obstacles, habitats, poly, cell_list, shark = ref_world(world, MPS)
        rrt_mod.time = ReplanClock(tick)
        # SharkUpdate / SharkOccupancyGrid are built at :67-68 and never read
        rrt_mod.SharkUpdate = lambda *a, **k: None
        rrt_mod.SharkOccupancyGrid = lambda *a, **k: None
        rrt = rrt_mod.RRT(poly, obstacles, shark, cell_list)
        iters = []
        orig_exploring, orig_steer = rrt.exploring, rrt.steer
        calls = []

        def steer(*a, **k):
            iters[-1] += 1
            return orig_steer(*a, **k)

        def exploring(initial, habs, *a, **k):
            iters.append(0)
            r = orig_exploring(initial, habs, *a, **k)
            calls.append({"t0": initial.traj_time_stamp, "n_hab": len(habs), "max_traj_time": k["max_traj_time"],
                          "cost": [float(r["cost"][0])] + [float(c) for c in r["cost"][1]],
                          "path_length": float(r["path length"]), "n_path": len(r["path"][0])})
            return r
        rrt.steer, rrt.exploring = steer, exploring
        random.seed(seed)
        start = MPS(float(world["start"][0]), float(world["start"][1]))
        hab_in = list(habitats)
        with contextlib.redirect_stdout(io.StringIO()):
            traj, time_dict, cost = rrt.replanning(start, hab_in, budget, traj_len, interval, [-3, -3, -4])
        rng_after = random.random()
        save_npz(name + ".npz", **world_arrays(world), start=world["start"], world_kwargs=json.dumps(wk), seed=seed,
                 plan_time_budget=budget, traj_time_length=traj_len, replan_time_interval=interval,
                 iters_per_round=np.array(iters, dtype=np.int32),
                 traj=np.array([[p.x, p.y, p.theta, p.v, p.traj_time_stamp, p.length] for p in traj], dtype=np.float64),
                 round_len=np.array([len(time_dict[k][0]) for k in time_dict], dtype=np.int32),
                 round_habitats=np.array([len(time_dict[k][1]) for k in time_dict], dtype=np.int32),
                 round_t0=np.array([c["t0"] for c in calls]), round_max_traj_time=np.array([c["max_traj_time"] for c in calls]),
                 round_cost=np.array([c["cost"] for c in calls]), round_path_length=np.array([c["path_length"] for c in calls]),
                 habitats_left=np.array([[h.x, h.y, h.size] for h in hab_in], dtype=np.float64).reshape(-1, 3),
                 cost=np.array([float(cost[0])] + [float(c) for c in cost[1]]), rng_after=rng_after)
        print(name, "rounds", len(calls), "iters", iters, "traj", len(traj), "habitats left", len(hab_in), "cost", cost[0])


# --------------------------------------------------------------------------------------------
# G14: SharkUpdate (path_planning/sharkEstimate.py:8-207) -- SURVEY 8(f) f3: the grid predictor RRT.replanning
# constructs at rrt_dubins.py:67.  Inputs are dense row x column grids (lists of lists); recorded: every method's
# return value AND what it did to its arguments (prediction2 / predictOnAve alias their inputs), plus the exception type
# update() ends with when it runs a second round on its own [init, result] pair.
# --------------------------------------------------------------------------------------------
def g14():
    import copy
    refstubs.install()
    _purge(_SHARED)
    saved = list(sys.path)
    sys.path[:0] = [os.path.join(REF, "path_planning"), REF]
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            se = importlib.import_module("sharkEstimate")
    finally:
        sys.path[:] = saved
    rng = random.Random(141)
    out = {"cases": []}
    specs = [("full_grid", (-300.0, -100.0, -100.0, 100.0), 10.0, None),
             ("ragged_cells", (0.0, 0.0, 47.0, 33.0), 5.0, 0.7),     # ceil() before the division, cells missing
             ("one_row", (2.0, 3.0, 26.0, 7.0), 4.0, None)]
    for name, box, cs, keep in specs:
        x0, y0, x1, y1 = box
        ncol, nrow = int(math.floor((x1 - x0) / cs)), int(math.floor((y1 - y0) / cs))
        cells = [(x0 + c * cs, y0 + r * cs, x0 + (c + 1) * cs, y0 + (r + 1) * cs) for r in range(nrow) for c in range(ncol)]
        if keep is not None:
            cells = [c for c in cells if rng.random() < keep]
            rng.shuffle(cells)                                       # prediction1 depends on the list order
        cell_objs = [refstubs.CellStub(*c) for c in cells]
        boundary = refstubs.Polygon([(x0, y0), (x1, y0), (x1, y1), (x0, y1)])
        upd = se.SharkUpdate(boundary, cs, cell_objs)
        rows = int(math.ceil(y1 - y0) / cs) + 1
        cols = int(math.ceil(x1 - x0) / cs) + 1
        prev = [[0.0 for _ in range(cols)] for _ in range(rows)]
        for c in cell_objs:
            r_, c_ = upd.cellToIndex(c)
            prev[r_][c_] = rng.choice([0.0, 0.0, rng.uniform(0.0, 0.2)])
        inf = [[rng.uniform(0.0, 0.1) for _ in range(cols)] for _ in range(rows)]
        parts = [[rng.randrange(0, 40) for _ in range(cols)] for _ in range(rows)]
        case = {"name": name, "box": list(box), "cell_size": cs, "cells": [list(c) for c in cells], "prev": prev, "inf": inf,
                "particles": parts, "index": [list(upd.cellToIndex(c)) for c in cell_objs]}
        case["prediction1"] = upd.prediction1(copy.deepcopy(prev), 0.6)
        a, b = copy.deepcopy(prev), copy.deepcopy(inf)
        case["prediction2"] = upd.prediction2(a, 0.1, b)
        case["prediction2_arg_after"] = a                          # shallow copy: the argument's rows are the result's rows
        for meth in (1, 2):
            a = copy.deepcopy(prev)
            r = upd.predictOnAve(a, False, meth, 0.6, 0.1)
            case["ave_m%d" % meth] = r
            case["ave_m%d_arg_after" % meth] = a
            r = upd.predictOnAve(None, True, meth, 0.6, 0.1)
            case["ave_exp_m%d" % meth] = r
            a, b = copy.deepcopy(prev), copy.deepcopy(inf)
            r = upd.predictOnHist(a, False, meth, b, 0.6, 0.1)
            case["hist_m%d" % meth] = r
            a, b = None, copy.deepcopy(inf)
            r = upd.predictOnHist(a, True, meth, b, 0.6, 0.1)
            case["hist_exp_m%d" % meth] = r
            case["hist_exp_m%d_inf_after" % meth] = b
        case["correction"] = upd.correction(parts, copy.deepcopy(case["prediction1"]))
        # update(): one round works, the second one feeds the [init, result] pair back in
        for meth in (["ave", 1], ["ave", 2], ["hist", 1], ["hist", 2]):
            key = "update_%s%d" % (meth[0], meth[1])
            grid = {(0, 10): copy.deepcopy(prev)}
            res = upd.update((0, 10), grid, 10, 10, meth)
            case[key + "_one_round"] = {"%d,%d" % k: v for k, v in res.items()}
            grid = {(0, 10): copy.deepcopy(prev)}
            try:
                upd.update((0, 10), grid, 20, 10, meth)
                case[key + "_two_rounds_raises"] = None
            except Exception as e:  # noqa: BLE001
                case[key + "_two_rounds_raises"] = type(e).__name__
        try:
            upd.predictOnAve(copy.deepcopy(prev), False, 3, 0.6, 0.1)
            case["ave_m3_raises"] = None
        except Exception as e:  # noqa: BLE001
            case["ave_m3_raises"] = type(e).__name__
        try:
            upd.correction([[0] * cols for _ in range(rows)], copy.deepcopy(case["prediction1"]))
            case["correction_zero_raises"] = None
        except Exception as e:  # noqa: BLE001
            case["correction_zero_raises"] = type(e).__name__
        out["cases"].append(case)
        print("g14", name, "grid", rows, "x", cols, "cells", len(cells), {k: v for k, v in case.items() if k.endswith("raises")})
    with open(os.path.join(OUT, "g14_shark_update.json"), "w") as f:
        json.dump(out, f)


def g15():
    """createSharkGrid, both twins (path_planning/rrt_dubins.py:612-630 and astar_fixLenSOG.py:31-49, which drops the last
    value of every row, :46), run by the reference itself on its own shark_data/*.csv with `.bounds` stand-in
    cells.  Inputs: the reference's DATA files, stored gzip-compressed under tests/golden/shark_data/ (no source).  Outputs:
    per file and twin the bin keys, the number of cells per bin, every probability, the first / last cell key."""
    import gzip
    import shutil
    rrt_mod, _, _ = import_rrt()
    sog_mod, _ = import_astar("astar_fixLenSOG")

    class Cell:
        def __init__(self, i):
            self.bounds = (float(i), 0.5 * i, float(i) + 1.0, 0.5 * i + 1.0)
    cell_list = [Cell(i) for i in range(1200)]
    src = os.path.join(REF, "path_planning", "shark_data")
    dst = os.path.join(OUT, "shark_data")
    os.makedirs(dst, exist_ok=True)
    out = {}
    for name in sorted(os.listdir(src)):
        if not name.endswith(".csv"):
            continue
        with open(os.path.join(src, name), "rb") as fi, gzip.GzipFile(os.path.join(dst, name + ".gz"), "wb", mtime=0) as fo:
            shutil.copyfileobj(fi, fo)
        stem = name[:-4]
        for twin, fn in (("rrt", rrt_mod.createSharkGrid), ("sog", sog_mod.createSharkGrid)):
            g = fn(os.path.join(src, name), cell_list)
            keys = list(g.keys())
            lens = [len(g[k]) for k in keys]
            vals = np.concatenate([np.array(list(g[k].values()), dtype=np.float64) for k in keys])
            out["%s_%s_keys" % (stem, twin)] = np.array(keys, dtype=np.int64).reshape(-1, 2)
            out["%s_%s_lens" % (stem, twin)] = np.array(lens, dtype=np.int64)
            out["%s_%s_vals" % (stem, twin)] = vals
            out["%s_%s_first_cell" % (stem, twin)] = np.array(list(g[keys[0]].keys())[0])
            out["%s_%s_last_cell" % (stem, twin)] = np.array(list(g[keys[-1]].keys())[-1])
            print("g15", stem, twin, "bins", len(keys), "cells per bin", sorted(set(lens)), "sum", float(vals.sum()))
    np.savez_compressed(os.path.join(OUT, "g15_shark_grid_csv.npz"), **out)


ALL = {"g15": g15, "g7": g7, "g5": g5, "g4": g4, "g3": g3, "g2": g2, "g1": g1, "g6": g6, "g6b": g6b, "g8": g8, "g9": g9,
       "g10": g10, "g11": g11, "g12": g12, "g13": g13, "g14": g14}

# the fast subset `--check` regenerates by default (seconds of reference Python each); AUVP_G3_ONLY narrows g3
FAST = ("g7", "g5", "g4", "g1", "g12", "g14", "g15")
FAST_G3 = "g3_tb_o64_i500,g3_nn_o64_i500,g3_pt_o64_i500,g3_tb_binreset,g3_tb_freq100,g3_nn_freq70_noobs,g3_tb_short_traj,g3_tb_dense"


def _same(a, b):
    if isinstance(a, dict) and isinstance(b, dict):
        return a.keys() == b.keys() and all(_same(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)) and isinstance(b, (list, tuple)):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, float) and isinstance(b, float):
        return a == b or (a != a and b != b)
    return a == b


def compare_dirs(new_dir, old_dir):
    """every file the generators wrote into new_dir against the committed fixture of the same name: every array of an .npz
    (dtype, shape, values; NaNs equal), every value of a .json, the bytes of a .csv.gz's content.  Returns the differences."""
    import gzip
    bad = []
    n_files = 0
    for root, _, files in os.walk(new_dir):
        for f in sorted(files):
            new = os.path.join(root, f)
            old = os.path.join(old_dir, os.path.relpath(new, new_dir))
            n_files += 1
            if not os.path.exists(old):
                bad.append("%s: no committed fixture" % f)
            elif f.endswith(".npz"):
                a, b = np.load(new, allow_pickle=False), np.load(old, allow_pickle=False)
                if sorted(a.files) != sorted(b.files):
                    bad.append("%s: arrays %s" % (f, sorted(set(a.files) ^ set(b.files))))
                    continue
                for k in a.files:
                    x, y = a[k], b[k]
                    if x.dtype != y.dtype or x.shape != y.shape or not np.array_equal(x, y, equal_nan=x.dtype.kind == "f"):
                        bad.append("%s[%s]" % (f, k))
            elif f.endswith(".json"):
                if not _same(json.load(open(new)), json.load(open(old))):
                    bad.append(f)
            elif f.endswith(".gz"):
                if gzip.open(new).read() != gzip.open(old).read():
                    bad.append(f)
    return n_files, bad


def check(which):
    """--check: run the generators into a scratch directory and compare with the committed fixtures (nothing is overwritten)"""
    import tempfile
    global OUT
    fast = not which
    which = which or list(FAST) + ["g3"]
    if fast and "AUVP_G3_ONLY" not in os.environ:
        os.environ["AUVP_G3_ONLY"] = FAST_G3
    with tempfile.TemporaryDirectory(prefix="auvp_golden_") as tmp:
        OUT = tmp
        try:
            for w in which:
                ALL[w]()
        finally:
            OUT = HERE
        n, bad = compare_dirs(tmp, HERE)
    print("golden check: %d regenerated files, %d differences%s" % (n, len(bad), (": " + ", ".join(bad[:12])) if bad else ""))
    return n, bad


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--check":
        # python tests/golden/make_golden.py --check [g1 g3 ...]   (no names: the fast subset; exit status 1 on a difference)
        sys.exit(1 if check(args[1:])[1] else 0)
    which = args or list(ALL)
    for w in which:
        ALL[w]()
