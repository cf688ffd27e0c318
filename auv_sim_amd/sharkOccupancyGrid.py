"""Drop-in for path_planning/sharkOccupancyGrid.py's `SharkOccupancyGrid` (the producer of the
`sharkGrid` dict that astar_fixLenSOG, RRT.createSharkGrid and the cost function consume):

    SharkOccupancyGrid(cell_size, boundary, bin_interval, detect_range, cell_list).convert(shark_dict)
        -> (resultArr, resultCell)          (sharkOccupancyGrid.py:47-74)

    resultArr[(t0, t1)]  = rows x cols list of lists, the AUV detection grid of the time bin
    resultCell[(t0, t1)] = {cell.bounds: value} for the listed cells whose value is non-zero (:282-291)

The three grid passes (occupancy histogram, disc stencil, mean over sharks) run in
csrc/sog_kernels.h through `auvp_sog_convert`; every floating-point sum keeps the reference's order,
so the grids are bit-identical to the reference's.  `boundary` and the cells only need `.bounds`
(cells may also be a [C,4] array of bounds).  The shapely polygon splitting of `splitCell`
(:376-393) is outside the path: `cell_list` must be given, or `boundary` must be an axis-aligned
rectangle (then `splitCell` below tiles it).
"""
import ctypes as C

import numpy as np

from . import _lib

_dp, _ip = _lib._dp, _lib._ip
_ctx_cache = {}
_bound = False


def _bind():
    global _bound
    L = _lib.load()
    if not _bound:
        L.auvp_sog_convert.argtypes = [C.c_void_p, _dp, C.c_int32, _dp, C.c_double, C.c_double, C.c_double, C.c_int32, _ip,
                                       _dp, C.c_int32, _ip, _ip, _ip, _dp, _dp]
        L.auvp_sog_convert.restype = C.c_int
        _bound = True
    return L


def _context(device):
    if device not in _ctx_cache:
        _ctx_cache[device] = _lib.Context(device)
    return _ctx_cache[device]


def _bounds_of(obj):
    return tuple(float(v) for v in (obj.bounds if hasattr(obj, "bounds") else obj))


class _Cell:
    """rectangle stand-in for the shapely cells: carries `.bounds` only"""
    __slots__ = ("bounds",)

    def __init__(self, minx, miny, maxx, maxy):
        self.bounds = (minx, miny, maxx, maxy)


def splitCell(boundary, cell_size):
    """Tiles an axis-aligned rectangular boundary with cell_size squares, row-major from (minx, miny).
    Stands in for the shapely splitter (:376-393) for rectangles only; its cell order is not pinned to
    the reference (the occupancy grids do not depend on the order unless cells overlap)."""
    minx, miny, maxx, maxy = _bounds_of(boundary)
    out = []
    ny = int(np.ceil((maxy - miny) / cell_size))
    nx = int(np.ceil((maxx - minx) / cell_size))
    for r in range(ny):
        for c in range(nx):
            out.append(_Cell(minx + c * cell_size, miny + r * cell_size, minx + (c + 1) * cell_size,
                             miny + (r + 1) * cell_size))
    return out


def convert_arrays(ctx, cells, box, cell_size, bin_interval, detect_range, traj_len, pts):
    """Array form of convert: cells [C,4], pts [sum(traj_len),3] (x, y, traj_time_stamp).
    Returns (bins [T,2], grids [T,rows,cols])."""
    L = _bind()
    cells = _lib._f64(cells, (-1, 4)) if len(cells) else np.zeros((0, 4))
    pts = _lib._f64(pts, (-1, 3)) if len(pts) else np.zeros((0, 3))
    tl = np.ascontiguousarray(traj_len, dtype=np.int32)
    if int(tl.sum()) != len(pts):
        raise ValueError("traj_len does not add up to the number of points")
    b = _lib._f64(np.asarray(box, dtype=np.float64), (4,))
    nb, rows, cols = (np.zeros(1, dtype=np.int32) for _ in range(3))
    args = (ctx.h, _lib._p(cells), len(cells), _lib._p(b), float(cell_size), float(bin_interval), float(detect_range),
            len(tl), _lib._p(tl, _ip), _lib._p(pts))
    rc = L.auvp_sog_convert(*args, 0, _lib._p(nb, _ip), _lib._p(rows, _ip), _lib._p(cols, _ip), None, None)
    if rc != 0 and rc != -2:
        ctx._chk(rc)
    T = int(nb[0])
    bins = np.zeros((T, 2))
    grids = np.zeros((T, int(rows[0]), int(cols[0])))
    if T:
        ctx._chk(L.auvp_sog_convert(*args, T, _lib._p(nb, _ip), _lib._p(rows, _ip), _lib._p(cols, _ip), _lib._p(bins),
                                    _lib._p(grids)))
    return bins, grids


class SharkOccupancyGrid:
    def __init__(self, cell_size, boundary, bin_interval, detect_range, cell_list=None, device=0):
        self.cell_size = cell_size
        if cell_list is not None and len(cell_list):
            self.cell_list = cell_list
        else:
            self.cell_list = splitCell(boundary, cell_size)
        self.bin_interval = bin_interval
        self.detect_range = detect_range
        self.boundary = boundary
        self._ctx = _context(device)
        self._cell_bounds = np.array([_bounds_of(c) for c in self.cell_list], dtype=np.float64).reshape(-1, 4)

    def createBinList(self):
        """(i*bin_interval, (i+1)*bin_interval) for i < floor(longest last time stamp / bin_interval) (:306-319)"""
        import math
        longest_time = 0
        for _, traj in self.data.items():
            if traj[-1].traj_time_stamp > longest_time:
                longest_time = traj[-1].traj_time_stamp
        return [(i * self.bin_interval, (i + 1) * self.bin_interval)
                for i in range(math.floor(longest_time / self.bin_interval))]

    def cellToIndex(self, cell):
        minx, miny, _, _ = _bounds_of(self.boundary)
        lowx, lowy, _, _ = _bounds_of(cell)
        return (int((lowy - miny) / self.cell_size), int((lowx - minx) / self.cell_size))

    def indexToCell(self, row, col):
        minx, miny, _, _ = _bounds_of(self.boundary)
        lowx = (col * self.cell_size) + minx
        lowy = (row * self.cell_size) + miny
        return (lowx, lowy, lowx + self.cell_size, lowy + self.cell_size)

    def convert2DArr(self, arr):
        result = {}
        for cell in self.cell_list:
            row, col = self.cellToIndex(cell)
            if arr[row][col] == 0:
                continue
            result[tuple(cell.bounds) if hasattr(cell, "bounds") else tuple(cell)] = arr[row][col]
        return result

    def convert(self, shark_dict):
        self.data = shark_dict
        self.bin_list = self.createBinList()
        traj_len = [len(t) for t in shark_dict.values()]
        pts = np.array([[p.x, p.y, p.traj_time_stamp] for t in shark_dict.values() for p in t], dtype=np.float64)
        _, grids = convert_arrays(self._ctx, self._cell_bounds, _bounds_of(self.boundary), self.cell_size,
                                  self.bin_interval, self.detect_range, traj_len, pts)
        if len(grids) != len(self.bin_list):
            raise RuntimeError("device bin count %d != createBinList %d" % (len(grids), len(self.bin_list)))
        resultArr, resultCell = {}, {}
        for time, g in zip(self.bin_list, grids):
            grid = g.tolist()
            resultArr[time] = grid
            resultCell[time] = self.convert2DArr(grid)
        return (resultArr, resultCell)
