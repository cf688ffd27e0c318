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
