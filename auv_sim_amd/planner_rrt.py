"""Drop-in for gym_rrt/envs/rrt_dubins.py: class Planner_RRT with the reference's constructor,
`planning` and `generate_one_node` (rrt_dubins.py:34,162,205) and the attributes RRTEnv reads
(rrt_env.py:213-222,250-295,488-493): `env_grid[row][col].subsection_cells[k].node_array/.theta`,
`.x/.y/.side_length`, `mps_list`, `occupied_grid_cells_array`.

The tree, the bucket grid and the collision / goal-arc tests live on the MI355X
(libauvplan.so auvp_prrt_*); the Python objects here are a mirror that is extended from the
device's per-step results, so `RRTEnv.step` keeps working on lists it can index.

RNG: like the reference, draws come from Python's global `random` stream (so `random.seed(k)` before
`planning()` reproduces the reference run); `plan_batch(..., seeds=...)` runs many seeded episodes
in one launch with the bucket choice on the device.  `max_nodes` (additive kwarg) bounds the tree.
Deviations (documented): the two blocking `input("stop")` calls (:151,:219) are not reproduced --
an empty cell gives `(False, None)`.  No CPU path: raises without the library / a GPU.
"""
import math
import random
import time

import numpy as np

from . import _lib
from ._prrt_lib import PlannerBatch
from .motion_plan_state import Motion_plan_state


def angle_wrap(ang):
    """gym_rrt/envs/grid_cell_rrt.py:11-27"""
    while not (-math.pi <= ang <= math.pi):
        ang += (-2 * math.pi) if ang > math.pi else (2 * math.pi)
    return ang


class Grid_cell_RRT:
    """gym_rrt/envs/grid_cell_rrt.py:34-84 (attribute-compatible mirror)"""

    class Subsection_grid_cell_RRT:
        def __init__(self, theta):
            self.theta = theta
            self.node_array = []

        def __repr__(self):
            return "Subsec: theta=" + str(self.theta) + ", node list: " + str(self.node_array)

    def __init__(self, x, y, side_length=1, num_of_subsections=8):
        self.x = x
        self.y = y
        self.side_length = side_length
        self.subsection_cells = []
        self.delta_theta = float(2.0 * np.pi) / float(num_of_subsections)
        theta = 0.0
        for _ in range(num_of_subsections):
            self.subsection_cells.append(self.Subsection_grid_cell_RRT(theta))
            theta = angle_wrap(theta + self.delta_theta)

    def has_node(self):
        return any(s.node_array != [] for s in self.subsection_cells)

    def __repr__(self):
        return "RRT Grid: [x=%s, y=%s, side length=%s], node list: %s" % (self.x, self.y, self.side_length,
                                                                          self.subsection_cells)


def _mt_state():
    ver, internal, _ = random.getstate()
    return np.array(internal[:624], dtype=np.uint32).reshape(1, 624), np.array([internal[624]], dtype=np.int32)


class Planner_RRT:
    def __init__(self, start, goal, boundary, obstacles, habitats, exp_rate=1, dist_to_end=2, diff_max=0.5, freq=50,
                 cell_side_length=2, subsections_in_cell=8, max_nodes=4096, device=0, context=None):
        self.start = start
        self.goal = goal
        self.boundary_point = boundary
        self.cell_side_length = cell_side_length
        self.subsections_in_cell = subsections_in_cell
        self.obstacle_list = obstacles
        self.habitats = habitats
        self.exp_rate = exp_rate
        self.dist_to_end = dist_to_end
        self.diff_max = diff_max
        self.freq = freq
        self.last_path = []
        self.t_start = time.time()
        # (`context`: a caller that builds planner after planner -- RRTEnv.reset -- keeps one device context)
        self._own_ctx = context is None
        self._ctx = _lib.Context(device) if context is None else context  # raises without the HIP library / a GPU
        self._ctx.set_world(obstacles=np.array([(float(o.x), float(o.y), float(o.size)) for o in obstacles],
                                               dtype=np.float64).reshape(-1, 3))
        rect = (float(boundary[0].x), float(boundary[0].y), float(boundary[1].x), float(boundary[1].y))
        self._rect = rect
        st = [[float(start.x), float(start.y), float(start.theta), float(start.traj_time_stamp)]]
        gl = [[float(goal.x), float(goal.y)]]
        kw = dict(freq=freq, cell=cell_side_length, subs=subsections_in_cell, exp_rate=exp_rate, dist_to_end=dist_to_end,
                  diff_max=diff_max)
        self._pb = PlannerBatch(self._ctx, st, gl, rect, int(max_nodes), mt_states=_mt_state(), **kw)
        # ---- Python mirror of discretize_env (:77-93) / add_node_to_grid (:108-159) ----
        self.env_grid = []
        for row in range(int(rect[3] - rect[1]) // int(cell_side_length)):
            self.env_grid.append([])
            for col in range(int(rect[2] - rect[0]) // int(cell_side_length)):
                self.env_grid[row].append(Grid_cell_RRT(boundary[0].x + col * cell_side_length,
                                                        boundary[0].y + row * cell_side_length,
                                                        side_length=cell_side_length,
                                                        num_of_subsections=subsections_in_cell))
        self._ncols = len(self.env_grid[0]) if self.env_grid else 0
        self.occupied_grid_cells_array = []
        self.mps_list = [self.start]
        self._cell_id = {}
        for r, row in enumerate(self.env_grid):
            for c, gc in enumerate(row):
                for k, sub in enumerate(gc.subsection_cells):
                    self._cell_id[id(sub)] = (r * self._ncols + c) * subsections_in_cell + k
        s0 = self._pb.summaries()[0]
        if s0["status"] < 0:
            raise IndexError("start state falls outside the bucket grid")  # reference: IndexError in add_node_to_grid
        t = self._pb.tree(0, s0)
        self._mirror_insert(self.start, int(t["node_bucket"][0]))

    def close(self):
        """release the device context (if this planner created it)"""
        if self._own_ctx and self._ctx is not None:
            self._ctx.close()
        self._ctx = None

    # ------------------------------------------------------------------ mirror helpers
    def _bucket_tuple(self, b):
        S = self.subsections_in_cell
        return (b // S) // self._ncols, (b // S) % self._ncols, b % S

    def _mirror_insert(self, mps, b):
        if b < 0:
            return
        r, c, k = self._bucket_tuple(b)
        arr = self.env_grid[r][c].subsection_cells[k].node_array
        arr.append(mps)
        if len(arr) == 1:
            self.occupied_grid_cells_array.append((r, c, k))

    def _pull_new_node(self, s, step_num):
        """materialise the node the device just accepted (and its path points) as Python objects"""
        me = int(s["last_new_node"])
        t = self._pb.node(0, me, cap_points=int(math.floor(self.freq)) + 4)  # this node only, not the whole tree
        n = t["node"]
        node = Motion_plan_state(float(n[0]), float(n[1]), theta=float(n[2]), traj_time_stamp=float(n[3]),
                                 rl_state_id=step_num)
        par = self.mps_list[t["parent"]]
        node.parent = par
        node.path = [par]
        for q in t["points"]:
            node.path.append(Motion_plan_state(float(q[0]), float(q[1]), theta=float(q[2]), traj_time_stamp=float(q[3]),
                                               rl_state_id=step_num))
        self.mps_list.append(node)
        self._mirror_insert(node, t["bucket"])
        return node

    def _final_path(self, s, step_num):
        """generate_final_course(final_node) (:317-327): the final node and its arc points are new objects of this step
        (rl_state_id = step_num: :375,:421); behind them come the tree's OWN objects -- every ancestor's path points and the
        ancestors themselves, each with the rl_state_id of the step that created it (solveRL-RRT.py:981-1006 reads those)"""
        arr = self._pb.paths(np.array([s]))[0]
        n_new = 1 + int(s["n_arc"])
        path = [Motion_plan_state(float(r[0]), float(r[1]), theta=float(r[2]), traj_time_stamp=float(r[3]),
                                  rl_state_id=step_num) for r in arr[:n_new]]
        m = self.mps_list[-1]
        while m.parent is not None:
            path.extend(reversed(m.path))
            m = m.parent
        if len(path) != len(arr):
            raise _lib.AuvpError(-4, "the host mirror of the tree (%d path elements) and the device's path (%d) disagree"
                                 % (len(path), len(arr)))
        return path, arr

    # ------------------------------------------------------------------ reference API
    def generate_one_node(self, grid_cell, step_num=None, min_length=250):
        """rrt_dubins.py:205-248 -> (done, path | new_node | None)"""
        if grid_cell.node_array == []:
            return False, None
        b = self._cell_id[id(grid_cell)]
        s = self._pb.step([b], mt_states=_mt_state())[0]  # continue Python's global stream on the device
        n = int(s["n_draw32"])
        if n:
            random.getrandbits(32 * n)  # ... and advance it by what the step consumed
        if s["status"] == -1:
            # AUVP_ERR_ARG from a step = the reference's IndexError in add_node_to_grid (:127: int(y / cell) or int(x / cell)
            # below -len on a world whose origin is left of / below 0).  mps_list holds the node by then (:229-230)
            if s["last_accepted"]:
                self._pull_new_node(s, step_num)
            raise IndexError("list index out of range")
        if s["status"] < 0:
            raise _lib.AuvpError(int(s["status"]), "Planner_RRT step failed on the device")
        new_node = self._pull_new_node(s, step_num) if s["last_accepted"] else None
        if s["done"]:
            path, arr = self._final_path(s, step_num)
            path[0].length = float(s["arc"][5])
            path[0].parent = self.mps_list[-1]
            return True, path
        if new_node is not None:
            return False, new_node
        return False, None

    def planning(self, max_step=200, min_length=250, plan_time=True):
        """rrt_dubins.py:162-202 -> (path, step, seconds)"""
        path = []
        step = 0
        start_time = time.time()
        for _ in range(max_step):
            r, c, k = random.choice(self.occupied_grid_cells_array)
            done, path = self.generate_one_node(self.env_grid[r][c].subsection_cells[k])
            step += 1
            if done:
                break
        return path, step, time.time() - start_time

    def cal_length(self, path):
        length = 0
        for i in range(1, len(path)):
            length += math.sqrt((path[i].x - path[i - 1].x) ** 2 + (path[i].y - path[i - 1].y) ** 2)
        return length


def plan_batch(starts, goals, boundary, obstacles, seeds, max_step=200, exp_rate=1, dist_to_end=2, diff_max=0.5, freq=50,
               cell_side_length=2, subsections_in_cell=8, device=0, context=None):
    """Many independent Planner_RRT(...).planning(max_step) runs in one launch (one wavefront each).
    starts: list of Motion_plan_state or [x,y,theta]; seeds[e] plays `random.seed(seeds[e])`.
    Returns (list of result dicts, PlannerBatch)."""
    ctx = context if context is not None else _lib.Context(device)
    ctx.set_world(obstacles=np.array([(float(o.x), float(o.y), float(o.size)) if hasattr(o, "x") else tuple(o)
                                      for o in obstacles], dtype=np.float64).reshape(-1, 3))

    def _s(m):
        return [float(m.x), float(m.y), float(m.theta), float(m.traj_time_stamp)] if hasattr(m, "x") else \
            (list(map(float, m)) + [0.0] * 4)[:4]

    def _g(m):
        return [float(m.x), float(m.y)] if hasattr(m, "x") else [float(m[0]), float(m[1])]

    rect = (float(boundary[0].x), float(boundary[0].y), float(boundary[1].x), float(boundary[1].y)) \
        if hasattr(boundary[0], "x") else tuple(map(float, boundary))
    pb = PlannerBatch(ctx, [_s(m) for m in starts], [_g(m) for m in goals], rect, int(max_step), seeds=seeds, freq=freq,
                      cell=cell_side_length, subs=subsections_in_cell, exp_rate=exp_rate, dist_to_end=dist_to_end,
                      diff_max=diff_max)
    summ = pb.plan()
    paths = pb.paths(summ)
    out = []
    for e in range(len(summ)):
        s = summ[e]
        out.append({"done": bool(s["done"]), "steps": int(s["steps"]), "n_nodes": int(s["n_nodes"]),
                    "status": int(s["status"]), "path": paths[e] if s["done"] else None})
    return out, pb
