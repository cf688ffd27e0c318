"""ctypes front-end for oracle/orc_astar.c (A* variants).  TEST INFRASTRUCTURE."""
import ctypes as C

import numpy as np

from . import orc

_dp, _ip = orc._dp, orc._ip
VARIANTS = {"astar": 0, "astar_real": 1, "astar_fixLen": 2, "astar_fixLenSOG": 3}


class AstarParams(C.Structure):
    _fields_ = [("variant", C.c_int32), ("cap_nodes", C.c_int32), ("start", C.c_double * 2), ("goal", C.c_double * 2),
                ("box", C.c_double * 4), ("limit", C.c_double), ("velocity", C.c_double), ("w", C.c_double * 4)]


class AstarOut(C.Structure):
    _fields_ = [("cap_exp", C.c_int32), ("cap_path", C.c_int32),
                ("n_nodes", C.c_int32), ("n_expansions", C.c_int32), ("n_children", C.c_int32), ("found", C.c_int32),
                ("status", C.c_int32), ("path_len", C.c_int32), ("smooth_len", C.c_int32), ("n_hab_left", C.c_int32),
                ("visited_count", C.c_int32), ("_pad", C.c_int32),
                ("exp_log", _dp), ("path", _dp), ("cost_list", _dp), ("node_path", _dp), ("smooth_path", _dp),
                ("hab_left", _ip)]


def run(variant, start, goal=(0.0, 0.0), obstacles=None, habitats=None, polygon=None, bins=None, cells=None, prob=None,
        box=(0, 0, 0, 0), limit=0.0, velocity=1.0, weights=(0, 0, 0, 0), cap_nodes=200000, kind="libm"):
    L = orc.lib(kind)
    L.orc_astar_run.restype = C.c_int
    L.orc_astar_run.argtypes = [C.POINTER(orc.World), C.POINTER(AstarParams), C.POINTER(AstarOut)]
    w = orc.WorldArrays(obstacles, habitats, polygon, bins, cells, prob)
    p = AstarParams()
    p.variant, p.cap_nodes = VARIANTS[variant], int(cap_nodes)
    p.start[0], p.start[1] = float(start[0]), float(start[1])
    p.goal[0], p.goal[1] = float(goal[0]), float(goal[1])
    for i in range(4):
        p.box[i] = float(box[i])
    wts = list(weights) + [0.0] * (4 - len(weights))
    for i in range(4):
        p.w[i] = float(wts[i])
    p.limit, p.velocity = float(limit), float(velocity)
    cap_exp, cap_path = cap_nodes, 4096
    H = len(w.habitats)
    a = {"exp_log": np.zeros((cap_exp, 8)), "path": np.zeros((cap_path, 3)), "cost_list": np.zeros(cap_path),
         "node_path": np.zeros((cap_path, 8)), "smooth_path": np.zeros((cap_path, 3)), "hab_left": np.zeros(max(H, 1), np.int32)}
    o = AstarOut()
    o.cap_exp, o.cap_path = cap_exp, cap_path
    for k in ("exp_log", "path", "cost_list", "node_path", "smooth_path"):
        setattr(o, k, orc._ptr(a[k]))
    o.hab_left = orc._ptr(a["hab_left"], _ip)
    status = L.orc_astar_run(C.byref(w.c), C.byref(p), C.byref(o))
    return {"status": status, "found": bool(o.found), "n_nodes": o.n_nodes, "n_expansions": o.n_expansions,
            "n_children": o.n_children, "visited_count": o.visited_count,
            "expansions": a["exp_log"][:o.n_expansions], "path": a["path"][:o.path_len],
            "cost_list": a["cost_list"][:o.path_len], "node_path": a["node_path"][:o.path_len],
            "smooth_path": a["smooth_path"][:o.smooth_len], "hab_left": a["hab_left"][:o.n_hab_left]}
