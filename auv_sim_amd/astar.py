"""Drop-in for path_planning/astar.py: `astar(start, goal, obs_lst, boundary).astar(obs_lst, start, goal)`
-> list of Motion_plan_state (start ... goal) or None (astar.py:21,193).  8-connected 10 m lattice,
squared-distance g and h, exact-goal stop, first-minimum pop, no dedup -- as the reference.
Additive: `cap_nodes`, `device`, and `astar_batch(starts, goals)` for many searches in one launch."""
from . import _astar_common as ac
from ._astar_common import Node  # noqa: F401


class astar:
    def __init__(self, start, goal, obs_lst, boundary, cap_nodes=200000, device=0):
        self.path = []
        self.start = start
        self.goal = goal
        self.obstacle_list = obs_lst
        self.min_bound = boundary[0]
        self.max_bound = boundary[1]
        self.cap_nodes = cap_nodes
        self._ctx = ac.context(device)

    def _box(self):
        return (float(self.min_bound.x), float(self.min_bound.y), float(self.max_bound.x), float(self.max_bound.y))

    def astar(self, obs_lst, start, goal):
        return self.astar_batch(obs_lst, [start], [goal])[0]

    def astar_batch(self, obs_lst, starts, goals):
        self._ctx.set_world(obstacles=ac.circles(obs_lst))
        res = ac.run(self._ctx, "astar", [tuple(map(float, s)) for s in starts], goals=[tuple(map(float, g)) for g in goals],
                     box=self._box(), cap_nodes=self.cap_nodes)
        return [ac.mps_path(r["path"], starts[i]) if r["found"] else None for i, r in enumerate(res)]
