"""ctypes front-end for oracle/orc_sog.c (SharkOccupancyGrid.convert restatement).  TEST INFRASTRUCTURE."""
import ctypes as C
import math

import numpy as np

from . import orc

_dp, _ip = orc._dp, orc._ip


class SogIn(C.Structure):
    _fields_ = [("n_cells", C.c_int32), ("n_sharks", C.c_int32), ("box", C.c_double * 4), ("cell_size", C.c_double),
                ("bin_interval", C.c_double), ("detect_range", C.c_double), ("cells", _dp), ("traj_len", _ip), ("pts", _dp)]


class SogOut(C.Structure):
    _fields_ = [("cap_bins", C.c_int32), ("n_bins", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32), ("bins", _dp),
                ("grids", _dp), ("occ_dbg", _dp), ("auv_dbg", _dp)]


def grid_shape(box, cs):
    return int(math.ceil(box[3] - box[1]) / cs) + 1, int(math.ceil(box[2] - box[0]) / cs) + 1


def convert(cells, box, cell_size, bin_interval, detect_range, traj_len, pts, kind="libm"):
    L = orc.lib(kind)
    L.orc_sog_convert.restype = C.c_int
    L.orc_sog_convert.argtypes = [C.POINTER(SogIn), C.POINTER(SogOut)]
    cells = orc._f64(cells, (-1, 4))
    pts = orc._f64(pts, (-1, 3))
    tl = np.ascontiguousarray(traj_len, dtype=np.int32)
    i = SogIn()
    i.n_cells, i.n_sharks = len(cells), len(tl)
    for k in range(4):
        i.box[k] = float(box[k])
    i.cell_size, i.bin_interval, i.detect_range = float(cell_size), float(bin_interval), float(detect_range)
    i.cells, i.traj_len, i.pts = orc._ptr(cells), orc._ptr(tl, _ip), orc._ptr(pts)
    rows, cols = grid_shape(box, cell_size)
    cap = int(pts[:, 2].max() / bin_interval) + 2 if len(pts) else 1
    bins, grids = np.zeros((cap, 2)), np.zeros((cap, rows, cols))
    occ, auv = np.zeros((rows, cols)), np.zeros((rows, cols))
    o = SogOut()
    o.cap_bins, o.bins, o.grids, o.occ_dbg, o.auv_dbg = cap, orc._ptr(bins), orc._ptr(grids), orc._ptr(occ), orc._ptr(auv)
    status = L.orc_sog_convert(C.byref(i), C.byref(o))
    return {"status": status, "bins": bins[:o.n_bins], "grids": grids[:o.n_bins], "occ": occ, "auv": auv}
