"""ctypes binding of the Planner_RRT entry points of libauvplan.so (auvp_prrt_*, include/auvplan.h)."""
import ctypes as C

import numpy as np

from . import _lib

_dp, _ip = _lib._dp, _lib._ip


class PrrtParams(C.Structure):
    _fields_ = [("rect", C.c_double * 4), ("exp_rate", C.c_double), ("dist_to_end", C.c_double),
                ("diff_max", C.c_double), ("freq", C.c_double), ("cell_side_length", C.c_double),
                ("subsections", C.c_int32), ("max_step", C.c_int32)]


PRRT_SUMMARY_DTYPE = np.dtype([("status", "<i4"), ("n_nodes", "<i4"), ("n_points", "<i4"), ("n_occ", "<i4"),
                               ("steps", "<i4"), ("done", "<i4"), ("path_len", "<i4"), ("last_node", "<i4"),
                               ("last_accepted", "<i4"), ("last_new_node", "<i4"), ("n_arc", "<i4"), ("_pad", "<i4"),
                               ("arc", "<f8", (6,)), ("rng_after", "<f8"), ("n_draw32", "<u8")])

_bound = False


def _bind():
    global _bound
    L = _lib.load()
    if _bound:
        return L
    vp = C.c_void_p
    u64p, u32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    L.auvp_prrt_create_batch.argtypes = [vp, C.c_int32, _dp, _dp, C.POINTER(PrrtParams), u64p, u32p, _ip, C.c_int32]
    L.auvp_prrt_plan.argtypes = [vp]
    L.auvp_prrt_step.argtypes = [vp, _ip, u32p, _ip]
    L.auvp_prrt_summaries.argtypes = [vp, C.c_void_p]
    L.auvp_prrt_paths.argtypes = [vp, C.POINTER(C.c_int64), _dp]
    L.auvp_prrt_tree.argtypes = [vp, C.c_int32, _dp, _ip, _ip, _dp]
    L.auvp_prrt_grid.argtypes = [vp, C.c_int32, _ip, _ip, _ip]
    L.auvp_prrt_node.argtypes = [vp, C.c_int32, C.c_int32, _dp, _ip, _ip, _dp, C.c_int32]
    L.auvp_prrt_replan_particles.argtypes = [vp, _dp, C.POINTER(PrrtParams), _dp, _dp, C.c_uint64, C.c_int32]
    L.auvp_prrt_step_log.argtypes = [vp, C.c_int32, _ip]
    L.auvp_prrt_summaries_dev.argtypes = [vp]
    L.auvp_prrt_summaries_dev.restype = C.c_void_p
    _bound = True
    return L


class PlannerBatch:
    """E Planner_RRT episodes resident on the device of `ctx` (an auv_sim_amd._lib.Context whose world
    holds the obstacle list)."""

    def __init__(self, ctx, starts, goals, rect, max_step, seeds=None, mt_states=None, freq=50, cell=2, subs=8,
                 exp_rate=1, dist_to_end=2, diff_max=0.5, step_log=False):
        self.ctx = ctx
        self.L = _bind()
        starts = _lib._f64(starts, (-1, 4))
        goals = _lib._f64(goals, (-1, 2))
        self.E = len(starts)
        p = PrrtParams()
        for i in range(4):
            p.rect[i] = float(rect[i])
        p.exp_rate, p.dist_to_end, p.diff_max, p.freq = float(exp_rate), float(dist_to_end), float(diff_max), float(freq)
        p.cell_side_length, p.subsections, p.max_step = float(cell), int(subs), int(max_step)
        self.max_step = int(max_step)
        self.rows = int(rect[3] - rect[1]) // int(cell)
        self.cols = int(rect[2] - rect[0]) // int(cell)
        self.subs = int(subs)
        flags = _lib.FLAG_ITER_LOG if step_log else 0
        if seeds is not None:
            sd = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64).reshape(self.E))
            rc = self.L.auvp_prrt_create_batch(ctx.h, self.E, _lib._p(starts), _lib._p(goals), C.byref(p),
                                               sd.ctypes.data_as(C.POINTER(C.c_uint64)), None, None, flags)
        else:
            words = np.ascontiguousarray(np.asarray(mt_states[0], dtype=np.uint32).reshape(self.E, 624))
            idx = np.ascontiguousarray(np.asarray(mt_states[1], dtype=np.int32).reshape(self.E))
            rc = self.L.auvp_prrt_create_batch(ctx.h, self.E, _lib._p(starts), _lib._p(goals), C.byref(p), None,
                                               words.ctypes.data_as(C.POINTER(C.c_uint32)), _lib._p(idx, _ip), flags)
        ctx._chk(rc)

    @classmethod
    def from_particles(cls, ctx, n_episodes, start, rect, max_step, xform, clamp, seed_base, freq=50, cell=2, subs=8,
                       exp_rate=1, dist_to_end=2, diff_max=0.5, step_log=False):
        """One episode per particle of the filter batch resident on `ctx` (auvp_prrt_replan_particles): goals are read
        from the particle state on the device, generators are seeded there as random.seed(seed_base + e)."""
        self = cls.__new__(cls)
        self.ctx, self.L, self.E = ctx, _bind(), int(n_episodes)
        p = PrrtParams()
        for i in range(4):
            p.rect[i] = float(rect[i])
        p.exp_rate, p.dist_to_end, p.diff_max, p.freq = float(exp_rate), float(dist_to_end), float(diff_max), float(freq)
        p.cell_side_length, p.subsections, p.max_step = float(cell), int(subs), int(max_step)
        self.max_step = int(max_step)
        self.rows = int(rect[3] - rect[1]) // int(cell)
        self.cols = int(rect[2] - rect[0]) // int(cell)
        self.subs = int(subs)
        st = _lib._f64((list(start) + [0.0] * 4)[:4])
        xf = _lib._f64(xform, (-1, 4))
        cl = _lib._f64(clamp, (4,))
        ctx._chk(self.L.auvp_prrt_replan_particles(ctx.h, _lib._p(st), C.byref(p), _lib._p(xf), _lib._p(cl), int(seed_base),
                                                   _lib.FLAG_ITER_LOG if step_log else 0))
        return self

    def plan(self):
        self.ctx._chk(self.L.auvp_prrt_plan(self.ctx.h))
        return self.summaries()

    def goals(self):
        """the goals of the resident batch (read back from the device)"""
        out = np.zeros((self.E, 2))
        self.L.auvp_prrt_goals.argtypes = [C.c_void_p, _dp]
        self.ctx._chk(self.L.auvp_prrt_goals(self.ctx.h, _lib._p(out)))
        return out

    def step(self, bucket_ids, mt_states=None):
        b = np.ascontiguousarray(np.asarray(bucket_ids, dtype=np.int32).reshape(self.E))
        if mt_states is None:
            rc = self.L.auvp_prrt_step(self.ctx.h, _lib._p(b, _ip), None, None)
        else:
            words = np.ascontiguousarray(np.asarray(mt_states[0], dtype=np.uint32).reshape(self.E, 624))
            idx = np.ascontiguousarray(np.asarray(mt_states[1], dtype=np.int32).reshape(self.E))
            rc = self.L.auvp_prrt_step(self.ctx.h, _lib._p(b, _ip), words.ctypes.data_as(C.POINTER(C.c_uint32)),
                                       _lib._p(idx, _ip))
        self.ctx._chk(rc)
        return self.summaries()

    def summaries(self):
        out = np.zeros(self.E, dtype=PRRT_SUMMARY_DTYPE)
        self.ctx._chk(self.L.auvp_prrt_summaries(self.ctx.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def paths(self, summ):
        lens = np.where(summ["done"] != 0, summ["path_len"], 0).astype(np.int64)
        off = np.zeros(self.E + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        out = np.zeros((max(int(off[-1]), 1), 5))
        self.ctx._chk(self.L.auvp_prrt_paths(self.ctx.h, off.ctypes.data_as(C.POINTER(C.c_int64)), _lib._p(out)))
        return [out[off[e]:off[e + 1]] for e in range(self.E)]

    def tree(self, ep, s):
        n, npnt = int(s["n_nodes"]), int(s["n_points"])
        nodes = np.zeros((n, 4))
        ni = np.zeros((n, 4), np.int32)
        nb = np.zeros(n, np.int32)
        pts = np.zeros((max(npnt, 1), 4))
        self.ctx._chk(self.L.auvp_prrt_tree(self.ctx.h, ep, _lib._p(nodes), _lib._p(ni, _ip), _lib._p(nb, _ip), _lib._p(pts)))
        return dict(nodes=nodes, step=ni[:, 0].copy(), parent=ni[:, 1].copy(), pt_off=ni[:, 2].copy(),
                    pt_cnt=ni[:, 3].copy(), node_bucket=nb, points=pts[:npnt])

    def node(self, ep, node, cap_points=256):
        """one node record + its path points (what a step-mode caller needs after an accepted step)"""
        nf = np.zeros(4)
        ni = np.zeros(4, np.int32)
        nb = np.zeros(1, np.int32)
        pts = np.zeros((cap_points, 4))
        self.ctx._chk(self.L.auvp_prrt_node(self.ctx.h, ep, int(node), _lib._p(nf), _lib._p(ni, _ip), _lib._p(nb, _ip),
                                            _lib._p(pts), cap_points))
        return dict(node=nf, step=int(ni[0]), parent=int(ni[1]), pt_off=int(ni[2]), pt_cnt=int(ni[3]),
                    bucket=int(nb[0]), points=pts[:int(ni[3])])

    def grid(self, ep):
        dims = np.zeros(4, np.int32)
        self.ctx._chk(self.L.auvp_prrt_grid(self.ctx.h, ep, None, None, _lib._p(dims, _ip)))
        occ = np.zeros(max(int(dims[3]), 1), np.int32)
        cnt = np.zeros(max(int(dims[0] * dims[1] * dims[2]), 1), np.int32)
        self.ctx._chk(self.L.auvp_prrt_grid(self.ctx.h, ep, _lib._p(occ, _ip), _lib._p(cnt, _ip), _lib._p(dims, _ip)))
        return occ[:int(dims[3])], cnt

    def step_log(self, ep, n_steps):
        log = np.zeros((self.max_step, 8), np.int32)
        self.ctx._chk(self.L.auvp_prrt_step_log(self.ctx.h, ep, _lib._p(log, _ip)))
        return log[:n_steps]
