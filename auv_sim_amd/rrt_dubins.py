"""Drop-in for path_planning/rrt_dubins.py: class RRT with the reference's constructor and
`exploring` / `replanning` signatures (rrt_dubins.py:26,51,92), running on the MI355X through
libauvplan.so.  Same inputs (Motion_plan_state lists, shapely-like boundary polygon, sharkGrid dict,
cell_list) and same outputs (dict with "path length" / "path" / "cost"; Motion_plan_state objects
linked by .parent/.path), so `robotSim`-style callers and `performance.summary_1` (:65-73) can switch
the import and nothing else.

Additive, non-reference keyword arguments:
  max_iter  iteration budget.  The reference loops until a wall-clock deadline (:116-118); here the
            budget is the virtual clock of SURVEY.md 8(c): exactly max_iter loop iterations and
            plan_time_stamp == 0-based iteration index.  Default: max_plan_time * RRT.iters_per_second.
  seed      None (default): continue Python's global `random` stream exactly where it stands -- so
            `random.seed(k); RRT(...).exploring(...)` draws what the reference would draw -- and
            advance it by the number of outputs consumed.  int: a private random.Random(seed) stream.
  device    HIP device index (constructor).

There is no CPU path: without libauvplan.so or a GPU the constructor raises.
"""
import math
import random

import numpy as np

from . import _lib
from .motion_plan_state import Motion_plan_state


def _polygon_vertices(boundary):
    """accepts a shapely Polygon (or look-alike with .exterior.coords), a list of (x, y) pairs, or a
    list of Motion_plan_state-like objects"""
    if hasattr(boundary, "exterior"):
        pts = [(float(p[0]), float(p[1])) for p in boundary.exterior.coords]
    else:
        pts = [(float(p.x), float(p.y)) if hasattr(p, "x") else (float(p[0]), float(p[1])) for p in boundary]
    if len(pts) > 1 and pts[0] == pts[-1]:
        pts = pts[:-1]
    return np.array(pts, dtype=np.float64).reshape(-1, 2)


def _circles(objs):
    return np.array([(float(o.x), float(o.y), float(o.size)) for o in objs], dtype=np.float64).reshape(-1, 3)


def pack_shark_grid(sharkGrid):
    """{(t0,t1): {cell.bounds: prob}} -> bins [T,2], cells [C,4], prob [T,C].  The inner dicts' key
    order is the scan order of the cost function (path_planning/cost.py:181), so it defines the cell
    order; every bin must list the same cells in the same order (createSharkGrid builds them that
    way, rrt_dubins.py:612-630)."""
    keys = list(sharkGrid.keys())
    if not keys:
        return np.zeros((0, 2)), np.zeros((0, 4)), np.zeros((0, 0))
    cell_keys = list(sharkGrid[keys[0]].keys())
    for k in keys[1:]:
        if list(sharkGrid[k].keys()) != cell_keys:
            raise NotImplementedError("shark grid bins with differing cell order are not supported")
    bins = np.array([(float(k[0]), float(k[1])) for k in keys], dtype=np.float64).reshape(-1, 2)
    cells = np.array(cell_keys, dtype=np.float64).reshape(-1, 4)
    prob = np.array([[float(sharkGrid[k][c]) for c in cell_keys] for k in keys], dtype=np.float64)
    return bins, cells, prob.reshape(len(keys), len(cell_keys))


def _mt_state_of(rng_state):
    """random.getstate() -> (624 words, index)"""
    ver, internal, _ = rng_state
    if ver != 3 or len(internal) != 625:
        raise RuntimeError("unexpected random.getstate() layout")
    return np.array(internal[:624], dtype=np.uint32), int(internal[624])


class RRT:
    """Goal-less cost-optimising RRT with random-arc steering (reference class of the same name)."""

    iters_per_second = 1000  # the reference's measured loop rate: wall-clock -> budget

    def __init__(self, boundary, obstacles, sharkGrid, cell_list, exp_rate=1, dist_to_end=2, diff_max=0.5,
                 freq=30, device=0):
        self.boundary_poly = boundary
        self.obstacle_list = obstacles
        self.last_path = []
        self.exp_rate = exp_rate
        self.dist_to_end = dist_to_end
        self.diff_max = diff_max
        self.freq = freq
        self.cell_list = cell_list
        self.sharkGrid = sharkGrid
        self.t_start = 0.0
        self._ctx = _lib.Context(device)  # raises without the HIP library / a GPU
        self._bins, self._cells, self._prob = pack_shark_grid(sharkGrid)
        self._poly = _polygon_vertices(boundary)
        self._world_habitats = None
        self._ctx.set_world(_circles(obstacles), None, self._poly, self._bins, self._cells, self._prob)
        self._cost_ctx = None   # cost-only context of replanning() (created on first use)
        self._last = None       # (summary, initial, E index) of the last exploring call
        self._mps_cache = None

    # ------------------------------------------------------------------ reference API
    def exploring(self, initial, habitats, plot_interval, bin_interval, v, shark_interval, traj_time_stamp=False,
                  max_plan_time=5, max_traj_time=200.0, plan_time=True, weights=[-1, -1, -1], max_iter=None,
                  seed=None):
        res = self.exploring_batch([initial], habitats, plot_interval, bin_interval, v, shark_interval,
                                   traj_time_stamp, max_plan_time, max_traj_time, plan_time, weights,
                                   max_iter=max_iter, seeds=None if seed is None else [seed])
        r = res[0]
        if r is None:
            # rrt_dubins.py:174: opt_path is None -> opt_path[1] raises
            raise TypeError("'NoneType' object is not subscriptable")
        return r

    def exploring_batch(self, initials, habitats, plot_interval, bin_interval, v, shark_interval,
                        traj_time_stamp=False, max_plan_time=5, max_traj_time=200.0, plan_time=True,
                        weights=[-1, -1, -1], max_iter=None, seeds=None):
        """E independent exploring() calls in one launch (one wavefront per episode).  Returns a list of
        result dicts (None where the reference would have raised for lack of a qualifying leaf)."""
        E = len(initials)
        if max_iter is None:
            max_iter = max(1, int(math.ceil(max_plan_time * self.iters_per_second)))
        mode = "timebin" if (plan_time and traj_time_stamp) else ("plantime" if plan_time else "nn")
        hab = _circles(habitats)
        self._ctx.set_habitats(hab)
        init = np.array([[float(m.x), float(m.y), float(m.theta), float(m.traj_time_stamp),
                          float(m.plan_time_stamp), float(m.length)] for m in initials], dtype=np.float64)
        use_global = seeds is None
        if use_global:
            if E != 1:
                raise ValueError("seeds are required for a batch (the global random stream is one stream)")
            words, idx = _mt_state_of(random.getstate())
            seed_arg = (words.reshape(1, 624), np.array([idx], dtype=np.int32))
        else:
            seed_arg = np.array([int(s) for s in seeds], dtype=np.uint64)
        summ = self._ctx.rrt_explore_batch(init, seed_arg, int(max_iter), mode=mode, freq=self.freq,
                                           bin_interval=bin_interval, v=v, max_traj_time=max_traj_time,
                                           weights=weights, dist_to_end=self.dist_to_end, diff_max=self.diff_max,
                                           min_dist=0.5, max_plan_time=float(max_iter))  # virtual clock: 1 tick per iteration
        if use_global:
            n = int(summ[0]["n_draw32"])
            if n:
                random.getrandbits(32 * n)  # advance the global stream by what the device consumed
        bad = summ["status"] < 0
        if bad.any():
            e = int(np.argmax(bad))
            raise _lib.AuvpError(int(summ[e]["status"]), "episode %d failed on the device (status %d)"
                                 % (e, int(summ[e]["status"])))
        paths = self._ctx.paths(summ)
        self._last = (summ, list(initials))
        self._mps_cache = None
        self.t_start = 0.0
        out = []
        for e in range(E):
            s = summ[e]
            if s["status"] == _lib.NO_QUALIFYING_LEAF:
                out.append(None)
                continue
            course = self._materialise_course(paths[e], initials[e])
            split = self.splitPath(course, shark_interval, [initials[e].traj_time_stamp, max_traj_time])
            c = [float(x) for x in s["best_cost"]]
            out.append({"path length": float(s["best_length"]), "path": [course, split], "cost": [c[0], c[1:]]})
        return out

    def replanning(self, start, habitats, plan_time_budget, traj_time_length, replan_time_interval, weight,
                   max_iter=None, seed=None):
        """Receding-horizon planning (rrt_dubins.py:51-90): plan a horizon with `exploring`, commit the first
        shark-interval bucket of the best course, drop the habitats that bucket visited, repeat until the
        shark grid's last bin ends.  Returns [committed trajectory, {round: [bucket, habitats at that round]},
        cost of the whole trajectory against the ORIGINAL habitat list].

        Like the reference (:67) the call leaves a `SharkUpdate` over the planner's boundary and cell list in
        `self.sharkEstimate` (sharkEstimate.py, host arithmetic; nothing here reads it, neither does the reference); the
        SharkOccupancyGrid of :68 is likewise never read and is not built (its constructor splits polygons with
        shapely).  The final cost runs on a cost-only device context of its own, so this object's obstacles and
        boundary stay on the device for later exploring() / check_collision() calls."""
        from .cost import habitat_shark_cost_func
        from .sharkEstimate import SharkUpdate
        self.sharkEstimate = SharkUpdate(self.boundary_poly, 10, self.cell_list)
        horizon_end = list(self.sharkGrid.keys())[-1][1]
        round_span = plan_time_budget + replan_time_interval  # also the shark_interval handed to exploring (:80)
        all_habitats = list(habitats)
        stream = random.Random(seed) if seed is not None else None
        committed, rounds = [start], {}
        while committed[-1].traj_time_stamp + round_span < horizon_end:
            t_now = committed[-1].traj_time_stamp
            if traj_time_length + t_now > horizon_end:  # stays clipped for the later rounds too (:76-77)
                traj_time_length = horizon_end - t_now
            plan = self.exploring(committed[-1], habitats, 0.5, 5, 2, round_span, traj_time_stamp=True,
                                  max_plan_time=plan_time_budget, max_traj_time=traj_time_length + t_now,
                                  plan_time=True, weights=weight, max_iter=max_iter,
                                  seed=stream.getrandbits(63) if stream is not None else None)
            buckets = plan["path"][1]
            first_bucket = buckets[next(iter(buckets))]
            committed += first_bucket
            rounds[len(rounds) + 1] = [first_bucket, list(habitats)]
            habitats = self.removeHabitat(habitats, first_bucket)
        if self._cost_ctx is None:
            self._cost_ctx = _lib.Context(self._ctx.device)
        total = habitat_shark_cost_func(committed[1:], committed[-1].traj_time_stamp, all_habitats, self.sharkGrid,
                                        weight=[-3, -3, -4], device_context=self._cost_ctx)
        return [committed[1:], rounds, total]

    # ------------------------------------------------------------------ host-side helpers (reference names)
    def splitPath(self, path, shark_interval, traj_time):
        """Bucket a course by shark interval (rrt_dubins.py:590-602): floor(traj_time[1] / shark_interval)
        closed intervals starting at traj_time[0]; a point goes to the first interval that contains its
        traj_time_stamp (so a point on a shared end belongs to the earlier one) or to none."""
        t0 = traj_time[0]
        edges = [(t0 + k * shark_interval, t0 + (k + 1) * shark_interval)
                 for k in range(math.floor(traj_time[1] / shark_interval))]
        buckets = {e: [] for e in edges}
        for point in path:
            t = point.traj_time_stamp
            hit = next((e for e in edges if e[0] <= t <= e[1]), None)
            if hit is not None:
                buckets[hit].append(point)
        return buckets

    def removeHabitat(self, habitats, path):
        """rrt_dubins.py:604-610: every path point removes the first habitat (current list order) that contains
        it.  Mutates and returns the caller's list, like the reference."""
        for point in path:
            inside = next((h for h in habitats if math.sqrt((point.x - h.x) ** 2 + (point.y - h.y) ** 2) <= h.size), None)
            if inside is not None:
                habitats.remove(inside)
        return habitats

    def check_collision(self, mps, obstacleList=None):
        """rrt_dubins.py:530-549 for one node (its .path points), evaluated on the device against the
        obstacle list given at construction (the reference always passes self.obstacle_list)."""
        if mps is None:
            return False
        pts = [(float(p.x), float(p.y)) for p in mps.path]
        if not pts:
            return True
        return bool(self._ctx.check_collision([pts])[0])

    def _materialise_course(self, arr, initial):
        course = []
        for i in range(len(arr)):
            if i == 0:
                course.append(initial)  # the reference returns the caller's own start object
                continue
            r = arr[i]
            course.append(Motion_plan_state(float(r[0]), float(r[1]), theta=float(r[2]), v=float(r[3]),
                                            traj_time_stamp=float(r[4]), plan_time_stamp=float(r[5]),
                                            length=float(r[6])))
        return course

    @property
    def mps_list(self):
        """The tree of the last exploring() call as linked Motion_plan_state objects (built on first
        access: the tree stays in HBM until someone asks for Python objects)."""
        if self._mps_cache is None:
            if self._last is None:
                return []
            summ, initials = self._last
            t = self._ctx.tree(0, summ[0])
            nodes = [initials[0]]
            for i in range(1, len(t["nodes"])):
                n = t["nodes"][i]
                nodes.append(Motion_plan_state(float(n[0]), float(n[1]), theta=float(n[2]), traj_time_stamp=float(n[3]),
                                               plan_time_stamp=float(n[4]), length=float(n[5])))
            for i in range(1, len(nodes)):
                par = nodes[int(t["parent"][i])]
                nodes[i].parent = par
                path = [par]
                o, c = int(t["pt_off"][i]), int(t["pt_cnt"][i])
                for q in t["points"][o:o + c]:
                    path.append(Motion_plan_state(float(q[0]), float(q[1]), theta=float(q[2]), v=float(q[3]),
                                                  traj_time_stamp=float(q[4]), plan_time_stamp=float(q[5]),
                                                  length=float(q[6])))
                nodes[i].path = path
            self._mps_cache = nodes
        return self._mps_cache


def createSharkGrid(filepath, cell_list):
    """CSV -> {(t0,t1): {cell.bounds: prob}} (rrt_dubins.py:612-630): header `time bin,grid`, one row
    per bin: "(t0, t1)","[p0, p1, ...]"; value i belongs to cell_list[i]."""
    import csv
    out = {}
    with open(filepath, newline="") as f:
        for row in csv.DictReader(f):
            a, b = row["time bin"].split(", ")
            key = (int(a[1:]), int(b[:-1]))
            vals = row["grid"][1:-1].split(", ")
            out[key] = {cell_list[i].bounds: float(vals[i]) for i in range(len(vals))}
    return out
