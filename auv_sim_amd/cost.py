"""Drop-in for the cost functions of path_planning/cost.py: habitat_shark_cost_func (:145-207)
evaluated on the MI355X (one wavefront per path, libauvplan.so `auvp_cost_paths`).

    habitat_shark_cost_func(path, total_traj_time, habitats, shark_dict, weight) -> [total, [c0, c1, c2]]

`shark_dict` is the (sub-)dict {(t0,t1): {cell.bounds: prob}} the reference passes; bins are scanned
in dict order, cells in the inner dicts' key order, first match wins, and the cell test keeps the
reference's `x <= maxy` comparison (cost.py:182).  No CPU path: raises without the library / a GPU.
"""
import numpy as np

from . import _lib

_default_ctx = {}


def _context(device=0):
    if device not in _default_ctx:
        _default_ctx[device] = _lib.Context(device)
    return _default_ctx[device]


def habitat_shark_cost_func(path, total_traj_time, habitats, shark_dict, weight, device=0, device_context=None):
    from .rrt_dubins import pack_shark_grid, _circles
    ctx = device_context if device_context is not None else _context(device)
    bins, cells, prob = pack_shark_grid(shark_dict)
    ctx.set_world(None, _circles(habitats), None, bins, cells, prob)
    pts = np.array([(float(m.x), float(m.y), float(m.traj_time_stamp)) for m in path], dtype=np.float64).reshape(-1, 3)
    if len(pts) == 0:
        pts = np.zeros((0, 3))
    out = ctx.cost_paths([pts], [0], [len(bins)], [float(total_traj_time)], [[float(w) for w in weight[:3]]])[0]
    return [float(out[0]), [float(out[1]), float(out[2]), float(out[3])]]


def habitat_shark_cost_batch(paths, total_traj_times, habitats, shark_dict, weight, device=0, device_context=None):
    """many paths against one world in a single launch; returns an [n,4] array (total, c0, c1, c2)"""
    from .rrt_dubins import pack_shark_grid, _circles
    ctx = device_context if device_context is not None else _context(device)
    bins, cells, prob = pack_shark_grid(shark_dict)
    ctx.set_world(None, _circles(habitats), None, bins, cells, prob)
    arrs = [np.array([(float(m.x), float(m.y), float(m.traj_time_stamp)) for m in p], dtype=np.float64).reshape(-1, 3)
            for p in paths]
    n = len(arrs)
    w = np.tile(np.array([float(x) for x in weight[:3]]), (n, 1))
    return ctx.cost_paths(arrs, [0] * n, [len(bins)] * n, [float(t) for t in total_traj_times], w)


class Cost:
    """Drop-in for the root-level cost.py `Cost` (the A* analysis twin):

    cost_of_edge(new_node, habitat_open_list, habitat_closed_list, weights) -> [cost, d_2, d_3]   (cost.py:66-102)
        one node against H habitats: host arithmetic, the same statement the A* kernels evaluate per child
        (astar_kernel variant 2, pinned by G6); offered for callers outside the search loop.
    habitat_shark_cost_func(path, length, peri, total_traj_time, habitats, shark_dict, weight)
        -> [total, [c0, c1, c2, c3]]                                                               (cost.py:151-214)
        the 4-weight form; evaluated on the GPU through auvp_cost_paths.
    """

    def __init__(self, device=0, device_context=None):
        self.cost = 0
        self._device, self._ctx = device, device_context

    def cost_of_edge(self, new_node, habitat_open_list, habitat_closed_list, weights):
        import math
        w2, w3 = weights[1], weights[2]
        x, y = new_node.position[0], new_node.position[1]

        def covered(habitats):
            hit = 0
            for habi in habitats:
                if math.sqrt((x - habi.x) ** 2 + (y - habi.y) ** 2) <= habi.size:
                    hit = 1
            return hit
        d_2 = covered(habitat_open_list + habitat_closed_list)
        d_3 = covered(habitat_closed_list)
        return [- w2 * d_2 - w3 * d_3, d_2, d_3]

    def habitat_shark_cost_func(self, path, length, peri, total_traj_time, habitats, shark_dict, weight):
        """Differences from path_planning/cost.py:145 that are kept: a point whose time stamp lies in no bin reuses
        the bin of the previous point (`temp_time` survives the loop, cost.py:185-190; UnboundLocalError if the
        first point has none), and the normalisation by total_traj_time is unconditional (ZeroDivisionError)."""
        from .rrt_dubins import pack_shark_grid, _circles
        w1, w2, w3, w4 = weight[0], weight[1], weight[2], weight[3]
        ctx = self._ctx if self._ctx is not None else _context(self._device)
        bins, cells, prob = pack_shark_grid(shark_dict)
        pts = np.array([(float(m.x), float(m.y), float(m.traj_time_stamp)) for m in path], dtype=np.float64).reshape(-1, 3)
        stale_t = None
        for i in range(len(pts)):
            if ((pts[i, 2] >= bins[:, 0]) & (pts[i, 2] <= bins[:, 1])).any():
                stale_t = pts[i, 2]
            elif stale_t is None:
                raise UnboundLocalError("local variable 'temp_time' referenced before assignment")
            else:
                pts[i, 2] = stale_t  # the time stamp only selects the bin: reuse the last one that selected any
        if total_traj_time == 0:
            raise ZeroDivisionError("float division by zero")
        ctx.set_world(None, _circles(habitats), None, bins, cells, prob)
        # with total = 1 the kernel returns the raw sums: c1 = sum(w3), c2 = sum(w4 * prob), c0 = w2 * count / H
        out = ctx.cost_paths([pts], [0], [len(bins)], [1.0], [[float(w2), float(w3), float(w4)]])[0]
        cost = [w1 * length / peri, float(out[1]) if len(habitats) != 0 else 0, float(out[2]) / total_traj_time,
                float(out[3]) / total_traj_time]
        return [sum(cost), cost]


def habitat_shark_cost_point(mps, habitats, visited, AUVGrid, weight):
    """path_planning/cost.py:209-242: cost of ONE motion_plan_state (performance.py:272 calls it per trajectory
    point).  O(H + C) host arithmetic on a single point; `visited` is returned unchanged because the reference's
    `visited[i] == True` (:232) is a comparison, not an assignment."""
    import math
    w1, w2, w3 = weight[0], weight[1], weight[2]
    cost = [0 for _ in range(len(weight))]
    for i in range(len(habitats)):
        if math.sqrt((habitats[i].x - mps.x) ** 2 + (habitats[i].y - mps.y) ** 2) <= habitats[i].size:
            if visited[i] == False:  # noqa: E712  (list entries may be numpy bools)
                cost[0] += w1 / len(habitats)
            cost[1] += w2 / len(habitats)
    for cell_bound, prob in AUVGrid.items():
        if mps.x >= cell_bound[0] and mps.x <= cell_bound[2] and mps.y >= cell_bound[1] and mps.x <= cell_bound[3]:
            cost[2] += w3 * prob
            break
    return sum(cost), visited
