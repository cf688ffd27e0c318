"""Drop-in for path_planning/astar_real.py: `astar(start, goal, obs_lst, boundary).astar(obs_lst,
boundary_list)` -> list of Motion_plan_state or None (astar_real.py:26,144).  Polygon workspace tested
by the reference's centroid triangle fan (:62-95), stop once within 10 m of the goal (:137-142)."""
from . import _astar_common as ac
from ._astar_common import Node  # noqa: F401


class astar:
    def __init__(self, start, goal, obs_lst, boundary, cap_nodes=200000, device=0):
        self.path = []
        self.start = start
        self.goal = goal
        self.obstacle_list = obs_lst
        self.boundary = boundary
        self.cap_nodes = cap_nodes
        self._ctx = ac.context(device)

    def astar(self, obs_lst, boundary_list):
        return self.astar_batch(obs_lst, boundary_list, [self.start], [self.goal])[0]

    def astar_batch(self, obs_lst, boundary_list, starts, goals):
        self._ctx.set_world(obstacles=ac.circles(obs_lst), polygon=ac.corners(boundary_list))
        res = ac.run(self._ctx, "astar_real", [tuple(map(float, s)) for s in starts],
                     goals=[tuple(map(float, g)) for g in goals], cap_nodes=self.cap_nodes)
        return [ac.mps_path(r["path"], starts[i]) if r["found"] else None for i, r in enumerate(res)]
