"""Drop-in for path_planning/astar_fixLenSOG.py: `astar(start, obstacleList, boundaryList, habitatList,
sharkGrid, shark_dict, AUV_velocity).astar(pathLenLimit, weights, shark_traj)` -> dict with
"path length", "path" (smoothed), "cost", "cost list", "node" -- or None when the open list runs empty
(astar_fixLenSOG.py:111,551,618).  `sharkGrid` is {(t0,t1): {cell.bounds: prob}}.  The reference's fallback for `{}`
(:127-132: `cell_list = splitCell(boundary_poly, 10)`, then createSharkGrid on a CSV path relative to its repo root) needs
shapely for the split; here the caller passes that cell list and the CSV path (`cell_list=`, `shark_csv=`, additive) and the
module's own createSharkGrid loads it the way the reference's does.  `self.visited_nodes` persists across calls, as in the
reference."""
import numpy as np

from . import _astar_common as ac
from ._astar_common import Node
from .motion_plan_state import Motion_plan_state
from .rrt_dubins import pack_shark_grid


def createSharkGrid(filepath, cell_list):
    """CSV -> {(t0,t1): {cell.bounds: prob}}, this module's twin of the loader (astar_fixLenSOG.py:31-49): same wire format
    as rrt_dubins.createSharkGrid (header `time bin,grid`, one row per bin: "(t0, t1)","[p0, p1, ...]"), but the LAST value
    of every row is dropped (`range(len(temp)-1)`, :46 -- "changed to fix outofindex error"), so a 987-value row yields 986
    cells."""
    import csv
    out = {}
    with open(filepath, newline="") as f:
        for row in csv.DictReader(f):
            a, b = row["time bin"].split(", ")
            key = (int(a[1:]), int(b[:-1]))
            vals = row["grid"][1:-1].split(", ")
            out[key] = {cell_list[i].bounds: float(vals[i]) for i in range(len(vals) - 1)}
    return out


class astar:
    def __init__(self, start, obstacleList, boundaryList, habitatList, sharkGrid, shark_dict, AUV_velocity,
                 cap_nodes=200000, device=0, cell_list=None, shark_csv=None):
        if not sharkGrid:
            if cell_list is None or shark_csv is None:
                raise ValueError("sharkGrid is empty: pass cell_list= (the reference's splitCell(boundary_poly, 10)) and shark_csv= "
                                 "(its 'path_planning/shark_data/AUVGrid_prob_500_straight.csv') for the fallback of :127-132")
            self.cell_list = cell_list
            sharkGrid = createSharkGrid(shark_csv, cell_list)
        self.start = start
        self.velocity = AUV_velocity
        self.obstacle_list = obstacleList
        self.boundary_list = boundaryList
        self.habitat_list = habitatList
        self.visited_nodes = np.zeros([600, 600])
        self.sharkGrid = sharkGrid
        self.sharkDict = shark_dict
        self.cap_nodes = cap_nodes
        self._ctx = ac.context(device)
        self._bins, self._cells, self._prob = pack_shark_grid(sharkGrid)

    def _result(self, r, start):
        if not r["found"]:
            return None
        like = start
        nodes = []
        prev = None
        for q in r["node_path"]:  # root -> leaf
            n = Node(prev, ac.position_of((q[0], q[1]), like))
            n.g, n.h, n.f, n.cost, n.pathLen, n.time_stamp = float(q[2]), float(q[3]), float(q[4]), float(q[5]), float(q[6]), int(q[7])
            nodes.append(n)
            prev = n
        smooth = []
        for s in r["smooth_path"]:
            p = ac.position_of((s[0], s[1]), like)
            smooth.append(Motion_plan_state(p[0], p[1], traj_time_stamp=int(s[2])))
        cost = [float(c) for c in r["cost_list"]]
        return {"path length": len(smooth), "path": smooth, "cost": cost[0], "cost list": cost, "node": nodes}

    def astar(self, pathLenLimit, weights, shark_traj):
        self._ctx.set_world(obstacles=ac.circles(self.obstacle_list), habitats=ac.circles(self.habitat_list),
                            polygon=ac.corners(self.boundary_list), bins=self._bins, cells=self._cells, prob=self._prob)
        r = ac.run(self._ctx, "astar_fixLenSOG", [tuple(map(float, self.start))], limits=[float(pathLenLimit)],
                   weights=[float(w) for w in weights], velocity=float(self.velocity), cap_nodes=self.cap_nodes,
                   visited=(self.visited_nodes != 0).astype(np.uint8)[None])[0]
        self.visited_nodes = r["visited"].astype(np.float64)
        return self._result(r, self.start)

    def astar_batch(self, starts, limits, weights):
        self._ctx.set_world(obstacles=ac.circles(self.obstacle_list), habitats=ac.circles(self.habitat_list),
                            polygon=ac.corners(self.boundary_list), bins=self._bins, cells=self._cells, prob=self._prob)
        res = ac.run(self._ctx, "astar_fixLenSOG", [tuple(map(float, s)) for s in starts], limits=[float(v) for v in limits],
                     weights=[float(w) for w in weights], velocity=float(self.velocity), cap_nodes=self.cap_nodes)
        return [self._result(r, starts[i]) for i, r in enumerate(res)]
