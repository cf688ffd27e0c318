"""Shared host code of the four A* drop-in modules (astar.py, astar_real.py, astar_fixLen.py,
astar_fixLenSOG.py): argument packing, result materialisation.  The searches run on the MI355X
(libauvplan.so auvp_astar_*); nothing here computes a path on the host."""
import numpy as np

from . import _astar_lib, _lib
from .motion_plan_state import Motion_plan_state


class Node:
    """a node in the graph (same attributes as the reference's Node classes,
    path_planning/astar.py:8-18, astar_fixLen.py:28-41, astar_fixLenSOG.py:93-106)"""

    def __init__(self, parent=None, position=None):
        self.parent = parent
        self.position = position
        self.g = 0
        self.h = 0
        self.f = 0
        self.cost = 0
        self.pathLen = 0
        self.time_stamp = 0


def circles(objs):
    return np.array([(float(o.x), float(o.y), float(o.size)) for o in objs], dtype=np.float64).reshape(-1, 3)


def corners(boundary):
    return np.array([(float(c.x), float(c.y)) for c in boundary], dtype=np.float64).reshape(-1, 2)


def position_of(pos, like):
    """keep the caller's number type: the reference's positions stay int when start is int"""
    x, y = float(pos[0]), float(pos[1])
    if all(isinstance(v, (int, np.integer)) for v in like) and x == int(x) and y == int(y):
        return (int(x), int(y))
    return (x, y)


def context(device):
    return _lib.Context(device)  # raises without libauvplan.so / a GPU: no CPU fallback


def run(ctx, variant, starts, **kw):
    res = _astar_lib.run_batch(ctx, variant, starts, **kw)
    for e, r in enumerate(res):
        if r["status"] == -2:
            raise MemoryError("A* instance %d exceeded cap_nodes (no dedup in the reference: the open list can "
                              "grow without bound); raise cap_nodes" % e)
        if r["status"] < 0:
            # states in which the reference itself raises (visited-bitmap IndexError, no time bin / no
            # cell for a node -> AttributeError / TypeError, top-n beyond the number of cells)
            raise IndexError("A* instance %d reached a state where the reference raises (status %d)" % (e, r["status"]))
    return res


def mps_path(arr, like):
    out = []
    for r in arr:
        p = position_of((r[0], r[1]), like)
        out.append(Motion_plan_state(p[0], p[1]))
    return out
