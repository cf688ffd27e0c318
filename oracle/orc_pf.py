"""ctypes front-end for oracle/orc_pf.c (particleFilter.py restatement).  TEST INFRASTRUCTURE."""
import ctypes as C

import numpy as np

from . import orc

_dp, _ip = orc._dp, orc._ip
_up = C.POINTER(C.c_uint32)


class PfIn(C.Structure):
    _fields_ = [("n_particles", C.c_int32), ("n_steps", C.c_int32), ("n_auv", C.c_int32), ("do_create", C.c_int32),
                ("shark0", C.c_double * 2), ("meas", _dp), ("shark_xy", _dp), ("init", _dp), ("init_obj", _ip), ("mt", _up),
                ("mt_pos", _ip)]


class PfOut(C.Structure):
    _fields_ = [("created", _dp), ("updated", _dp), ("resampled", _dp), ("choice", _ip), ("alias_first", _ip),
                ("list_len", _ip), ("mean", _dp), ("range_error", _dp), ("n_draw32", C.c_uint64)]


def np_seed_state(seed):
    """MT19937 key after np.random.seed(seed) (init_genrand), pos = 624"""
    mt = np.zeros(624, dtype=np.uint32)
    s = int(seed) & 0xffffffff
    for i in range(624):
        mt[i] = s
        s = (1812433253 * (s ^ (s >> 30)) + i + 1) & 0xffffffff
    return mt, 624


def run(n_particles, meas, shark_xy, shark0=(0.0, 0.0), mt=None, mt_pos=624, init=None, init_obj=None, kind="libm"):
    """meas [S,A,5], shark_xy [S,2]; create() first unless `init` [N,5] is given."""
    L = orc.lib(kind)
    L.orc_pf_run.restype = C.c_int
    L.orc_pf_run.argtypes = [C.POINTER(PfIn), C.POINTER(PfOut)]
    meas = np.ascontiguousarray(meas, dtype=np.float64)
    S, A = meas.shape[0], meas.shape[1]
    shark_xy = orc._f64(shark_xy, (S, 2)) if S else np.zeros((0, 2))
    N = int(n_particles)
    mt = np.ascontiguousarray(mt, dtype=np.uint32).copy()
    pos = np.array([mt_pos], dtype=np.int32)
    i = PfIn()
    i.n_particles, i.n_steps, i.n_auv, i.do_create = N, S, A, 0 if init is not None else 1
    i.shark0[0], i.shark0[1] = float(shark0[0]), float(shark0[1])
    i.meas, i.shark_xy = orc._ptr(meas), orc._ptr(shark_xy)
    if init is not None:
        init = orc._f64(init, (N, 5))
        i.init = orc._ptr(init)
        if init_obj is not None:
            init_obj = np.ascontiguousarray(init_obj, dtype=np.int32)
            i.init_obj = orc._ptr(init_obj, _ip)
    i.mt, i.mt_pos = mt.ctypes.data_as(_up), orc._ptr(pos, _ip)
    r = {"created": np.zeros((N, 5)), "updated": np.zeros((S, N, 5)), "resampled": np.zeros((S, N, 5)),
         "choice": np.zeros((S, N), dtype=np.int32), "alias_first": np.zeros((S, N), dtype=np.int32),
         "list_len": np.zeros(S, dtype=np.int32), "mean": np.zeros((S, 2)), "range_error": np.zeros(S)}
    o = PfOut()
    for k, v in r.items():
        setattr(o, k, orc._ptr(v, _ip if v.dtype == np.int32 else _dp))
    r["status"] = L.orc_pf_run(C.byref(i), C.byref(o))
    r["mt"], r["mt_pos"], r["n_draw32"] = mt, int(pos[0]), int(o.n_draw32)
    return r


def np_kat(seed, n, choice_n, kind="libm"):
    L = orc.lib(kind)
    L.orc_np_kat.restype = None
    L.orc_np_kat.argtypes = [C.c_uint32, C.c_int, _dp, _ip, C.c_int32]
    u, c = np.zeros(n), np.zeros(n, dtype=np.int32)
    L.orc_np_kat(seed, n, orc._ptr(u), orc._ptr(c, _ip), choice_n)
    return u, c


def pow_e(z, kind="libm"):
    L = orc.lib(kind)
    L.orc_pow_e.restype = C.c_double
    L.orc_pow_e.argtypes = [C.c_double]
    return np.array([L.orc_pow_e(float(v)) for v in np.asarray(z).ravel()])
