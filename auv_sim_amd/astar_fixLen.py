"""Drop-in for path_planning/astar_fixLen.py: `astar(start, obs_lst, boundary).astar(habitat_list, obs_lst,
boundary_list, start, pathLenLimit, weights)` -> [path (Motion_plan_state list), cost list] or None
(astar_fixLen.py:45,286).  Like the reference it REMOVES the habitats the search covered from the
caller's `habitat_list` (the reference aliases and pops it, :310,:193-197).
`self.visited_nodes` persists across calls on one solver object, as in the reference."""
import numpy as np

from . import _astar_common as ac
from ._astar_common import Node  # noqa: F401


class astar:
    def __init__(self, start, obs_lst, boundary, cap_nodes=200000, device=0):
        self.path = []
        self.start = start
        self.obstacle_list = obs_lst
        self.boundary_list = boundary
        self.visited_nodes = np.zeros([550, 600])
        self.cap_nodes = cap_nodes
        self._ctx = ac.context(device)

    def astar(self, habitat_list, obs_lst, boundary_list, start, pathLenLimit, weights):
        self._ctx.set_world(obstacles=ac.circles(obs_lst), habitats=ac.circles(habitat_list), polygon=ac.corners(boundary_list))
        r = ac.run(self._ctx, "astar_fixLen", [tuple(map(float, start))], limits=[float(pathLenLimit)],
                   weights=[float(w) for w in weights], cap_nodes=self.cap_nodes,
                   visited=(self.visited_nodes != 0).astype(np.uint8)[None])[0]
        self.visited_nodes = r["visited"].astype(np.float64)
        keep = set(int(i) for i in r["hab_left"])
        survivors = [h for i, h in enumerate(list(habitat_list)) if i in keep]
        habitat_list[:] = survivors  # same objects, same order as the reference leaves them
        if not r["found"]:
            return None
        return [ac.mps_path(r["path"], start), [float(c) for c in r["cost_list"]]]

    def astar_batch(self, habitat_list, obs_lst, boundary_list, starts, limits, weights):
        """independent searches (each with its own copy of the habitat list); returns list of [path, cost] / None"""
        self._ctx.set_world(obstacles=ac.circles(obs_lst), habitats=ac.circles(habitat_list), polygon=ac.corners(boundary_list))
        res = ac.run(self._ctx, "astar_fixLen", [tuple(map(float, s)) for s in starts], limits=[float(v) for v in limits],
                     weights=[float(w) for w in weights], cap_nodes=self.cap_nodes)
        return [[ac.mps_path(r["path"], starts[i]), [float(c) for c in r["cost_list"]]] if r["found"] else None
                for i, r in enumerate(res)]
