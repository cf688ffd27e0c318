"""ctypes binding of the particle-filter entry points of libauvplan.so (auvp_pf_*, include/auvplan.h)."""
import ctypes as C

import numpy as np

from . import _lib

_dp, _ip = _lib._dp, _lib._ip
_up = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
UPDATE, WEIGHTS, MEAN = 1, 2, 4
_bound = False


def _bind():
    global _bound
    L = _lib.load()
    if _bound:
        return L
    vp, i32 = C.c_void_p, C.c_int32
    L.auvp_pf_create_batch.argtypes = [vp, i32, i32, _dp, _up, _ip]
    L.auvp_pf_set_particles.argtypes = [vp, i32, i32, _dp, _ip, _ip, _up, _ip]
    L.auvp_pf_set_rng.argtypes = [vp, _up, _ip]
    L.auvp_pf_run.argtypes = [vp, i32, i32, i32, _dp, _dp, i32]
    L.auvp_pf_particles.argtypes = [vp, _dp, _ip]
    L.auvp_pf_estimates.argtypes = [vp, _dp, _dp, _ip]
    L.auvp_pf_status.argtypes = [vp, _ip, _u64p]
    L.auvp_pf_rng_state.argtypes = [vp, _up, _ip]
    L.auvp_pf_step_log.argtypes = [vp, _dp, _ip]
    _bound = True
    return L


def np_seed_state(seed):
    """MT19937 key + position right after np.random.seed(seed) (numpy legacy seeding = init_genrand)"""
    mt = np.zeros(624, dtype=np.uint32)
    s = int(seed) & 0xffffffff
    for i in range(624):
        mt[i] = s
        s = (1812433253 * (s ^ (s >> 30)) + i + 1) & 0xffffffff
    return mt, 624


class FilterBatch:
    """F particle filters x N particles resident on one GPU (one per Context: the handle keeps one batch)."""

    def __init__(self, ctx, n_filters, n_particles=1000):
        self.ctx, self.F, self.N = ctx, int(n_filters), int(n_particles)
        self.L = _bind()
        self.S = 0

    def _rng_args(self, mt, pos):
        mt = np.ascontiguousarray(mt, dtype=np.uint32).reshape(self.F, 624)
        pos = np.ascontiguousarray(np.broadcast_to(np.asarray(pos, dtype=np.int32), (self.F,)))
        return mt, pos

    def create(self, shark_xy0, mt, pos=624):
        xy = _lib._f64(shark_xy0, (self.F, 2))
        mt, pos = self._rng_args(mt, pos)
        self.ctx._chk(self.L.auvp_pf_create_batch(self.ctx.h, self.F, self.N, _lib._p(xy), mt.ctypes.data_as(_up),
                                                  _lib._p(pos, _ip)))
        return self

    def set_particles(self, particles, mt, pos, obj=None, list_len=None):
        pa = _lib._f64(particles, (self.F, self.N, 5))
        mt, pos = self._rng_args(mt, pos)
        ob = np.ascontiguousarray(obj, dtype=np.int32).reshape(self.F, self.N) if obj is not None else None
        ll = np.ascontiguousarray(list_len, dtype=np.int32).reshape(self.F) if list_len is not None else None
        self.ctx._chk(self.L.auvp_pf_set_particles(self.ctx.h, self.F, self.N, _lib._p(pa),
                                                   _lib._p(ob, _ip) if ob is not None else None,
                                                   _lib._p(ll, _ip) if ll is not None else None,
                                                   mt.ctypes.data_as(_up), _lib._p(pos, _ip)))
        return self

    def set_rng(self, mt, pos):
        mt, pos = self._rng_args(mt, pos)
        self.ctx._chk(self.L.auvp_pf_set_rng(self.ctx.h, mt.ctypes.data_as(_up), _lib._p(pos, _ip)))

    def run(self, meas=None, shark_xy=None, phases=UPDATE | WEIGHTS | MEAN, n_steps=None, log=False):
        """meas [S,F,A,5], shark_xy [S,F,2]; one launch for all S steps."""
        m = sx = None
        A = 0
        if meas is not None:
            m = np.ascontiguousarray(meas, dtype=np.float64)
            if m.ndim != 4 or m.shape[1] != self.F or m.shape[3] != 5:
                raise ValueError("meas must be [S, F, A, 5]")
            n_steps, A = m.shape[0], m.shape[2]
        if shark_xy is not None:
            sx = np.ascontiguousarray(shark_xy, dtype=np.float64).reshape(-1, self.F, 2)
            n_steps = len(sx) if n_steps is None else n_steps
            if len(sx) != n_steps:
                raise ValueError("shark_xy must be [S, F, 2]")
        self.S = int(n_steps or 1)
        self.ctx._chk(self.L.auvp_pf_run(self.ctx.h, self.S, A, int(phases), _lib._p(m) if m is not None else None,
                                         _lib._p(sx) if sx is not None else None, 1 if log else 0))
        return self

    def particles(self):
        out = np.zeros((self.F, self.N, 5))
        obj = np.zeros((self.F, self.N), dtype=np.int32)
        self.ctx._chk(self.L.auvp_pf_particles(self.ctx.h, _lib._p(out), _lib._p(obj, _ip)))
        return out, obj

    def estimates(self):
        mean, err = np.zeros((self.S, self.F, 2)), np.zeros((self.S, self.F))
        ll = np.zeros((self.S, self.F), dtype=np.int32)
        self.ctx._chk(self.L.auvp_pf_estimates(self.ctx.h, _lib._p(mean), _lib._p(err), _lib._p(ll, _ip)))
        return mean, err, ll

    def status(self):
        st = np.zeros(self.F, dtype=np.int32)
        nd = np.zeros(self.F, dtype=np.uint64)
        self.ctx._chk(self.L.auvp_pf_status(self.ctx.h, _lib._p(st, _ip), nd.ctypes.data_as(_u64p)))
        return st, nd

    def rng_state(self):
        mt = np.zeros((self.F, 624), dtype=np.uint32)
        pos = np.zeros(self.F, dtype=np.int32)
        self.ctx._chk(self.L.auvp_pf_rng_state(self.ctx.h, mt.ctypes.data_as(_up), _lib._p(pos, _ip)))
        return mt, pos

    def step_log(self):
        upd = np.zeros((self.S, self.F, self.N, 5))
        cho = np.zeros((self.S, self.F, self.N), dtype=np.int32)
        self.ctx._chk(self.L.auvp_pf_step_log(self.ctx.h, _lib._p(upd), _lib._p(cho, _ip)))
        return upd, cho
