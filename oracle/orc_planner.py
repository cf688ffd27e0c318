"""ctypes front-end for oracle/orc_planner.c (Planner_RRT restatement).  TEST INFRASTRUCTURE."""
import ctypes as C

import numpy as np

from . import orc

_dp, _ip, _bp = orc._dp, orc._ip, orc._bp


class PrrtParams(C.Structure):
    _fields_ = [("start", C.c_double * 4), ("goal", C.c_double * 2), ("rect", C.c_double * 4),
                ("exp_rate", C.c_double), ("dist_to_end", C.c_double), ("diff_max", C.c_double),
                ("freq", C.c_double), ("cell_side_length", C.c_double), ("subsections", C.c_int32),
                ("max_step", C.c_int32)]


class PrrtOut(C.Structure):
    _fields_ = [("cap_nodes", C.c_int32), ("cap_points", C.c_int32), ("cap_path", C.c_int32), ("cap_buckets", C.c_int32),
                ("n_nodes", C.c_int32), ("n_points", C.c_int32), ("steps", C.c_int32), ("done", C.c_int32),
                ("status", C.c_int32), ("n_occ", C.c_int32), ("n_buckets", C.c_int32), ("grid_rows", C.c_int32),
                ("grid_cols", C.c_int32), ("path_len", C.c_int32),
                ("rng_after", C.c_double), ("n_draw32", C.c_uint64),
                ("nodes", _dp), ("parent", _ip), ("pt_off", _ip), ("pt_cnt", _ip), ("node_bucket", _ip), ("points", _dp),
                ("st_bucket", _ip), ("st_picked", _ip), ("st_accepted", _bp), ("st_done", _bp), ("st_npath", _ip),
                ("st_arc_n", _ip), ("st_arc_free", _bp), ("occupied", _ip), ("bucket_counts", _ip), ("path", _dp)]


def planning(obstacles, rect, start, goal, seed, max_step=2000, freq=10, cell=5, subs=1, exp_rate=1, dist_to_end=2,
             diff_max=0.5, kind="libm"):
    L = orc.lib(kind)
    L.orc_prrt_planning.restype = C.c_int
    L.orc_prrt_planning.argtypes = [C.POINTER(orc.World), C.POINTER(PrrtParams), C.c_uint64, C.POINTER(PrrtOut)]
    w = orc.WorldArrays(obstacles=obstacles)
    p = PrrtParams()
    st = list(start) + [0.0] * (4 - len(start))
    for i in range(4):
        p.start[i] = float(st[i])
        p.rect[i] = float(rect[i])
    p.goal[0], p.goal[1] = float(goal[0]), float(goal[1])
    p.exp_rate, p.dist_to_end, p.diff_max, p.freq = float(exp_rate), float(dist_to_end), float(diff_max), float(freq)
    p.cell_side_length, p.subsections, p.max_step = float(cell), int(subs), int(max_step)
    rows = int(rect[3] - rect[1]) // int(cell)
    cols = int(rect[2] - rect[0]) // int(cell)
    nb = max(rows * cols * subs, 1)
    capn, capp, cappath = max_step + 1, max_step * (int(freq) + 1) + 1, 200000
    a = {"nodes": np.zeros((capn, 5)), "parent": np.zeros(capn, np.int32), "pt_off": np.zeros(capn, np.int32),
         "pt_cnt": np.zeros(capn, np.int32), "node_bucket": np.zeros(capn, np.int32), "points": np.zeros((capp, 4)),
         "st_bucket": np.zeros(max_step, np.int32), "st_picked": np.zeros(max_step, np.int32),
         "st_accepted": np.zeros(max_step, np.int8), "st_done": np.zeros(max_step, np.int8),
         "st_npath": np.zeros(max_step, np.int32), "st_arc_n": np.zeros(max_step, np.int32),
         "st_arc_free": np.zeros(max_step, np.int8), "occupied": np.zeros(capn, np.int32),
         "bucket_counts": np.zeros(nb, np.int32), "path": np.zeros((cappath, 5))}
    o = PrrtOut()
    o.cap_nodes, o.cap_points, o.cap_path, o.cap_buckets = capn, capp, cappath, nb
    for k in ("nodes", "points", "path"):
        setattr(o, k, orc._ptr(a[k]))
    for k in ("parent", "pt_off", "pt_cnt", "node_bucket", "st_bucket", "st_picked", "st_npath", "st_arc_n", "occupied",
              "bucket_counts"):
        setattr(o, k, orc._ptr(a[k], _ip))
    for k in ("st_accepted", "st_done", "st_arc_free"):
        setattr(o, k, orc._ptr(a[k], _bp))
    status = L.orc_prrt_planning(C.byref(w.c), C.byref(p), int(seed), C.byref(o))
    n, s = o.n_nodes, o.steps
    return {"status": status, "steps": s, "done": bool(o.done), "n_nodes": n, "n_points": o.n_points,
            "rng_after": o.rng_after, "n_draw32": o.n_draw32, "grid_rows": o.grid_rows, "grid_cols": o.grid_cols,
            "nodes": a["nodes"][:n], "parent": a["parent"][:n], "pt_off": a["pt_off"][:n], "pt_cnt": a["pt_cnt"][:n],
            "node_bucket": a["node_bucket"][:n], "points": a["points"][:o.n_points],
            "st_bucket": a["st_bucket"][:s], "st_picked": a["st_picked"][:s], "st_accepted": a["st_accepted"][:s],
            "st_done": a["st_done"][:s], "st_npath": a["st_npath"][:s], "st_arc_n": a["st_arc_n"][:s],
            "st_arc_free": a["st_arc_free"][:s], "occupied": a["occupied"][:o.n_occ],
            "bucket_counts": a["bucket_counts"][:o.n_buckets], "path": a["path"][:o.path_len]}
