"""ctypes binding of the A* entry points of libauvplan.so (auvp_astar_*, include/auvplan.h)."""
import ctypes as C

import numpy as np

from . import _lib

_dp, _ip = _lib._dp, _lib._ip
VARIANTS = {"astar": 0, "astar_real": 1, "astar_fixLen": 2, "astar_fixLenSOG": 3}


class AstarParams(C.Structure):
    _fields_ = [("variant", C.c_int32), ("cap_nodes", C.c_int32), ("box", C.c_double * 4), ("velocity", C.c_double),
                ("w", C.c_double * 4)]


ASTAR_SUMMARY_DTYPE = np.dtype([(n, "<i4") for n in ("status", "found", "n_nodes", "n_expansions", "n_children", "path_len",
                                                     "smooth_len", "n_hab_left", "visited_count", "leaf")] +
                               [("open_scanned", "<u8")])
_bound = False


def _bind():
    global _bound
    L = _lib.load()
    if _bound:
        return L
    vp = C.c_void_p
    L.auvp_astar_batch.argtypes = [vp, C.c_int32, _dp, _dp, _dp, C.POINTER(AstarParams), C.c_int32]
    L.auvp_astar_summaries.argtypes = [vp, C.c_void_p]
    L.auvp_astar_paths.argtypes = [vp, C.POINTER(C.c_int64), _dp, _dp, _dp, _dp]
    L.auvp_astar_exp_log.argtypes = [vp, C.c_int32, _dp]
    L.auvp_astar_hab_left.argtypes = [vp, C.c_int32, _ip]
    L.auvp_astar_set_visited.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(C.c_uint8)]
    L.auvp_astar_get_visited.argtypes = [vp, C.c_int32, C.POINTER(C.c_uint8)]
    _bound = True
    return L


def run_batch_arrays(ctx, variant, starts, goals=None, limits=None, box=(0, 0, 0, 0), velocity=1.0, weights=(0, 0, 0, 0),
                     cap_nodes=20000):
    """E searches in one launch + one path-extraction launch; results as arrays (no per-instance Python work):
    summaries [E], offsets [E+1] and the concatenated path / cost_list / node_path / smooth_path rows."""
    L = _bind()
    starts = _lib._f64(starts, (-1, 2))
    E = len(starts)
    p = AstarParams()
    p.variant, p.cap_nodes, p.velocity = VARIANTS[variant], int(cap_nodes), float(velocity)
    wts = list(weights) + [0.0] * (4 - len(weights))
    for i in range(4):
        p.box[i] = float(box[i])
        p.w[i] = float(wts[i])
    g = _lib._f64(goals, (-1, 2)) if goals is not None else None
    lim = _lib._f64(limits).reshape(E) if limits is not None else None
    ctx._chk(L.auvp_astar_batch(ctx.h, E, _lib._p(starts), _lib._p(g) if g is not None else None,
                                _lib._p(lim) if lim is not None else None, C.byref(p), 0))
    batch_ms = ctx.last_kernel_ms()
    summ = np.zeros(E, dtype=ASTAR_SUMMARY_DTYPE)
    ctx._chk(L.auvp_astar_summaries(ctx.h, summ.ctypes.data_as(C.c_void_p)))
    lens = np.where(summ["found"] != 0, summ["path_len"], 0).astype(np.int64)
    off = np.zeros(E + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    n = max(int(off[-1]), 1)
    path, cost, npath, smooth = np.zeros((n, 3)), np.zeros(n), np.zeros((n, 8)), np.zeros((n, 3))
    ctx._chk(L.auvp_astar_paths(ctx.h, off.ctypes.data_as(C.POINTER(C.c_int64)), _lib._p(path), _lib._p(cost), _lib._p(npath),
                                _lib._p(smooth)))
    return dict(summ=summ, off=off, path=path, cost_list=cost, node_path=npath, smooth_path=smooth, batch_ms=batch_ms)


def run_batch(ctx, variant, starts, goals=None, limits=None, box=(0, 0, 0, 0), velocity=1.0, weights=(0, 0, 0, 0),
              cap_nodes=20000, exp_log=False, visited=None):
    """E searches over ctx's world in one launch.  Returns a list of per-instance dicts."""
    L = _bind()
    starts = _lib._f64(starts, (-1, 2))
    E = len(starts)
    p = AstarParams()
    p.variant, p.cap_nodes, p.velocity = VARIANTS[variant], int(cap_nodes), float(velocity)
    wts = list(weights) + [0.0] * (4 - len(weights))
    for i in range(4):
        p.box[i] = float(box[i])
        p.w[i] = float(wts[i])
    g = _lib._f64(goals, (-1, 2)) if goals is not None else None
    lim = _lib._f64(limits).reshape(E) if limits is not None else None
    flags = 1 if exp_log else 0
    if visited is not None:  # [E, vx, 600] uint8: the solver's visited_nodes carried over from earlier calls
        vis = np.ascontiguousarray(visited, dtype=np.uint8)
        ctx._chk(L.auvp_astar_set_visited(ctx.h, E, p.variant, vis.ctypes.data_as(C.POINTER(C.c_uint8))))
        flags |= 8
    ctx._chk(L.auvp_astar_batch(ctx.h, E, _lib._p(starts), _lib._p(g) if g is not None else None,
                                _lib._p(lim) if lim is not None else None, C.byref(p), flags))
    summ = np.zeros(E, dtype=ASTAR_SUMMARY_DTYPE)
    ctx._chk(L.auvp_astar_summaries(ctx.h, summ.ctypes.data_as(C.c_void_p)))
    lens = np.where(summ["found"] != 0, summ["path_len"], 0).astype(np.int64)
    off = np.zeros(E + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    n = max(int(off[-1]), 1)
    path, cost, npath, smooth = np.zeros((n, 3)), np.zeros(n), np.zeros((n, 8)), np.zeros((n, 3))
    ctx._chk(L.auvp_astar_paths(ctx.h, off.ctypes.data_as(C.POINTER(C.c_int64)), _lib._p(path), _lib._p(cost), _lib._p(npath),
                                _lib._p(smooth)))
    ctx._chk(L.auvp_astar_summaries(ctx.h, summ.ctypes.data_as(C.c_void_p)))  # smooth_len is filled by the path pass
    H = ctx.world_sizes.get("H", 0)
    out = []
    for e in range(E):
        s = summ[e]
        a, b = int(off[e]), int(off[e + 1])
        r = {"status": int(s["status"]), "found": bool(s["found"]), "n_nodes": int(s["n_nodes"]),
             "n_expansions": int(s["n_expansions"]), "n_children": int(s["n_children"]),
             "visited_count": int(s["visited_count"]), "path": path[a:b].copy(), "cost_list": cost[a:b].copy(),
             "node_path": npath[a:b].copy(), "smooth_path": smooth[a:a + int(s["smooth_len"])].copy()}
        hl = np.zeros(max(H, 1), np.int32)
        if H:
            ctx._chk(L.auvp_astar_hab_left(ctx.h, e, _lib._p(hl, _ip)))
        r["hab_left"] = hl[:int(s["n_hab_left"])]
        if p.variant >= 2 and visited is not None:
            vb = np.zeros((550 if p.variant == 2 else 600, 600), dtype=np.uint8)
            ctx._chk(L.auvp_astar_get_visited(ctx.h, e, vb.ctypes.data_as(C.POINTER(C.c_uint8))))
            r["visited"] = vb
        if exp_log:
            ex = np.zeros((max(int(s["n_expansions"]), 1), 8))
            ctx._chk(L.auvp_astar_exp_log(ctx.h, e, _lib._p(ex)))
            r["expansions"] = ex[:int(s["n_expansions"])]
        out.append(r)
    return out
