"""ctypes binding of libauvplan.so (include/auvplan.h).

There is no CPU fallback: importing this module without the built library, or creating a context
without a usable MI355X, raises.  Build with `python __graft_entry__.py build` (hipcc, gfx950).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AUVPLAN_LIBRARY") or os.path.join(HERE, "libauvplan.so")  # override: kernel experiments

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_bp = C.POINTER(C.c_int8)


class AuvpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("auvplan error %d: %s" % (code, msg))
        self.code = code


class RRTParams(C.Structure):
    _fields_ = [("dist_to_end", C.c_double), ("diff_max", C.c_double), ("freq", C.c_double),
                ("min_dist", C.c_double), ("bin_interval", C.c_double), ("v", C.c_double),
                ("max_traj_time", C.c_double), ("max_plan_time", C.c_double), ("w", C.c_double * 3),
                ("mode", C.c_int32), ("max_iter", C.c_int32), ("points_per_iter", C.c_double)]


class RRTSummary(C.Structure):
    _fields_ = [("status", C.c_int32), ("n_nodes", C.c_int32), ("n_points", C.c_int32),
                ("n_leaves", C.c_int32), ("best_leaf", C.c_int32), ("best_path_len", C.c_int32),
                ("iters_run", C.c_int32), ("n_candidates", C.c_int32), ("best_cost", C.c_double * 4),
                ("best_length", C.c_double), ("rng_after", C.c_double), ("leaf_elems", C.c_int64), ("n_draw32", C.c_uint64),
                ("nn_scanned", C.c_uint64)]


SUMMARY_DTYPE = np.dtype([("status", "<i4"), ("n_nodes", "<i4"), ("n_points", "<i4"), ("n_leaves", "<i4"),
                          ("best_leaf", "<i4"), ("best_path_len", "<i4"), ("iters_run", "<i4"), ("n_candidates", "<i4"),
                          ("best_cost", "<f8", (4,)), ("best_length", "<f8"), ("rng_after", "<f8"), ("leaf_elems", "<i8"), ("n_draw32", "<u8"),
                          ("nn_scanned", "<u8")])
assert SUMMARY_DTYPE.itemsize == C.sizeof(RRTSummary)

MODES = {"timebin": 0, "plantime": 1, "nn": 2}
FLAG_ITER_LOG, FLAG_LEAF_LOG, FLAG_PHASE_CLOCKS = 1, 2, 4
OK, NO_QUALIFYING_LEAF = 0, 1

_lib = None


def _share_hip_runtime_with_torch():
    """A process must hold ONE HIP/HSA runtime.  PyTorch-ROCm bundles its own libamdhip64.so.7 /
    libhsa-runtime64.so.1 (same SONAMEs as /opt/rocm's); whichever copy is loaded first serves both
    torch and libauvplan.so.  torch does not find its GPU on the system copy, so when torch is
    installed its copies are mapped first (without importing torch).  AUVP_SYSTEM_HIP=1 opts out."""
    import sys
    if os.environ.get("AUVP_SYSTEM_HIP") == "1" or "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        libdir = os.path.join(os.path.dirname(spec.origin), "lib")
        for name in ("libhsa-runtime64.so", "libamdhip64.so"):
            p = os.path.join(libdir, name)
            if os.path.exists(p):
                C.CDLL(p, mode=C.RTLD_GLOBAL)
    except OSError:
        pass  # fall back to the system runtime; torch (if imported later) may then not see the GPU


def load():
    """Load libauvplan.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: build the HIP extension first "
                          "(python -c 'import __graft_entry__ as g; g.build()')" % LIB_PATH)
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.auvp_version.restype = C.c_char_p
    L.auvp_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.auvp_destroy.argtypes = [vp]
    L.auvp_destroy.restype = None
    L.auvp_last_error.argtypes = [vp]
    L.auvp_last_error.restype = C.c_char_p
    L.auvp_world_set.argtypes = [vp, _dp, C.c_int32, _dp, C.c_int32, _dp, C.c_int32, _dp, C.c_int32, _dp,
                                 C.c_int32, _dp]
    L.auvp_rrt_explore_batch.argtypes = [vp, C.c_int32, _dp, C.POINTER(C.c_uint64), C.POINTER(RRTParams), C.c_int32]
    L.auvp_rrt_prepare.argtypes = [vp, C.c_int32, _dp, C.POINTER(C.c_uint64), C.POINTER(RRTParams), C.c_int32]
    L.auvp_rrt_prepare_states.argtypes = [vp, C.c_int32, _dp, C.POINTER(C.c_uint32), _ip, C.POINTER(RRTParams), C.c_int32]
    L.auvp_world_set_habitats.argtypes = [vp, _dp, C.c_int32]
    L.auvp_rrt_run.argtypes = [vp]
    L.auvp_rrt_paths_dev.argtypes = [vp, C.POINTER(C.c_int64), C.c_void_p]
    L.auvp_rrt_summaries.argtypes = [vp, C.c_void_p]
    L.auvp_rrt_paths.argtypes = [vp, C.POINTER(C.c_int64), _dp]
    L.auvp_rrt_tree.argtypes = [vp, C.c_int32, _dp, _ip, _ip, _ip, _dp]
    L.auvp_rrt_iter_log.argtypes = [vp, C.c_int32, _ip, _bp, _ip]
    L.auvp_rrt_leaf_log.argtypes = [vp, C.c_int32, _dp, _ip]
    L.auvp_rrt_bin_sizes.argtypes = [vp, C.c_int32, _ip, _ip]
    L.auvp_rrt_summaries_dev.argtypes = [vp]
    L.auvp_rrt_summaries_dev.restype = C.c_void_p
    L.auvp_check_collision_batch.argtypes = [vp, C.c_int32, _ip, _dp, _bp]
    L.auvp_cost_paths.argtypes = [vp, C.c_int32, _ip, _dp, _ip, _ip, _dp, _dp, _dp]
    L.auvp_sincos_dev.argtypes = [vp, C.c_int32, _dp, _dp, _dp]
    L.auvp_math_dev.argtypes = [vp, C.c_int32, C.c_int32, _dp, _dp, _dp, _dp]
    L.auvp_nn_closest_batch.argtypes = [vp, C.c_int32, _dp, C.c_int32, _dp, C.c_int32, _ip, _ip]
    L.auvp_random_stream_dev.argtypes = [vp, C.c_uint64, C.c_int32, _dp]
    L.auvp_rrt_phase_clocks.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.auvp_last_kernel_ms.argtypes = [vp]
    L.auvp_last_kernel_ms.restype = C.c_double
    L.auvp_last_launch.argtypes = [vp, _ip, _ip, _ip]
    L.auvp_rrt_last_launch_parts.argtypes = [vp, _dp, _dp, _ip]
    L.auvp_rrt_last_leaf_stats.argtypes = [vp, C.POINTER(C.c_int64)]
    L.auvp_rrt_last_kernel.argtypes = [vp]
    L.auvp_rrt_last_kernel.restype = C.c_char_p
    L.auvp_rrt_last_stream_ms.argtypes = [vp]
    L.auvp_rrt_last_stream_ms.restype = C.c_double
    L.auvp_rrt_last_stream_len.argtypes = [vp]
    L.auvp_rrt_last_stream_len.restype = C.c_int64
    L.auvp_hbm_probe.argtypes = [vp, C.c_uint64, C.c_int32, _dp, _dp]
    L.auvp_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.auvp_unset_option.argtypes = [vp, C.c_char_p]
    L.auvp_get_option.argtypes = [vp, C.c_char_p, _ip, C.POINTER(C.c_int64)]
    L.auvp_pipeline_fallbacks.argtypes = [vp, _ip, C.POINTER(C.c_int64)]
    _lib = L
    return L


# the options of a handle (auvp_set_option).  AUVP_<NAME> in the environment is read ONCE per handle, by
# auvp_create.
OPTION_NAMES = ("ROWS", "DUO", "TRIO", "QUAD", "TIGHT_CULL", "NN_EXACT", "LEAF_SWEEP_ALL", "NO_HABITAT_GRID", "RG_MAX_ENTRIES",
                "NO_GRID_INDEX", "PRRT_LAT", "PRRT_PIPE", "PRRT_OBST_LDS", "PRRT_NEXT_LDS", "PRRT_ROWS", "ASTAR_NO_GRID",
                "ASTAR_NO_LIST", "ASTAR_PAIR", "SOG_TILE", "PIPE_FALLBACK", "PRRT_PIPE_DRAW", "PRRT_BUCKET_LDS", "ROWS_STREAM", "ROWS_STREAM_CAP", "ROWS_STREAM_WAVES")


def _f64(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    return a.reshape(shape) if shape is not None else a


def _p(a, t=_dp):
    return a.ctypes.data_as(t)


class Context:
    """One planner context = one HIP device + stream + device-resident world and tree storage."""

    def __init__(self, device=0):
        self.L = load()
        self.h = C.c_void_p()
        rc = self.L.auvp_create(int(device), C.byref(self.h))
        if rc != 0:
            raise AuvpError(rc, "auvp_create(device=%d) failed: no usable HIP device (no CPU fallback)" % device)
        self.device = int(device)
        self.n_episodes = 0
        self.max_iter = 0
        self.world_sizes = {}

    def close(self):
        if getattr(self, "h", None):
            self.L.auvp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise AuvpError(rc, self.L.auvp_last_error(self.h).decode())
        return rc

    # ---- options (include/auvplan.h: auvp_set_option) ----
    def set_option(self, name, value):
        """force a kernel choice / diagnostic switch of this context (`name` without the AUVP_ prefix); None: back to the default"""
        if value is None:
            self._chk(self.L.auvp_unset_option(self.h, name.encode()))
        else:
            self._chk(self.L.auvp_set_option(self.h, name.encode(), int(value)))

    def get_option(self, name):
        """the option's value, None when it is unset (the measured default applies)"""
        s, v = C.c_int32(0), C.c_int64(0)
        self._chk(self.L.auvp_get_option(self.h, name.encode(), C.byref(s), C.byref(v)))
        return int(v.value) if s.value else None

    def pipeline_fallbacks(self):
        """(episodes the last planning call redid on the one-wavefront kernel after AUVP_ERR_PIPELINE, total so far)"""
        a, b = C.c_int32(0), C.c_int64(0)
        self._chk(self.L.auvp_pipeline_fallbacks(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    # ---- world ----
    def set_world(self, obstacles=None, habitats=None, polygon=None, bins=None, cells=None, prob=None):
        ob = _f64(obstacles if obstacles is not None else [], (-1, 3))
        hb = _f64(habitats if habitats is not None else [], (-1, 3))
        pg = _f64(polygon if polygon is not None else [], (-1, 2))
        bn = _f64(bins if bins is not None else [], (-1, 2))
        ce = _f64(cells if cells is not None else [], (-1, 4))
        pr = _f64(prob if prob is not None else [])
        if pr.size != len(bn) * len(ce):
            raise ValueError("prob must be [n_bins, n_cells]")
        self._chk(self.L.auvp_world_set(self.h, _p(ob), len(ob), _p(hb), len(hb), _p(pg), len(pg), _p(bn), len(bn),
                                        _p(ce), len(ce), _p(pr)))
        self.world_sizes = dict(O=len(ob), H=len(hb), V=len(pg), T=len(bn), C=len(ce))

    def set_habitats(self, habitats):
        hb = _f64(habitats if habitats is not None else [], (-1, 3))
        self._chk(self.L.auvp_world_set_habitats(self.h, _p(hb), len(hb)))
        self.world_sizes["H"] = len(hb)

    # ---- RRT.exploring batch ----
    def rrt_explore_batch(self, init, seeds, n_iter, mode="timebin", freq=30, bin_interval=5, v=2,
                          max_traj_time=500.0, weights=(-3, -3, -4), dist_to_end=2, diff_max=0.5, min_dist=0.5,
                          max_plan_time=None, points_per_iter=0.0, iter_log=False, leaf_log=False, phase_clocks=False):
        self.rrt_prepare(init, seeds, n_iter, mode, freq, bin_interval, v, max_traj_time, weights, dist_to_end,
                         diff_max, min_dist, max_plan_time, points_per_iter, iter_log, leaf_log, phase_clocks)
        self.rrt_run()
        return self.summaries()

    def rrt_run(self):
        """launch the kernel on the prepared batch (inputs already resident in HBM); repeatable"""
        self._chk(self.L.auvp_rrt_run(self.h))

    def rrt_prepare(self, init, seeds, n_iter, mode="timebin", freq=30, bin_interval=5, v=2,
                    max_traj_time=500.0, weights=(-3, -3, -4), dist_to_end=2, diff_max=0.5, min_dist=0.5,
                    max_plan_time=None, points_per_iter=0.0, iter_log=False, leaf_log=False, phase_clocks=False):
        init = _f64(init, (-1, 6))
        E = len(init)
        states = None
        if isinstance(seeds, tuple):  # (mt [E,624] uint32, index [E]) as random.getstate() reports
            states = np.ascontiguousarray(np.asarray(seeds[0], dtype=np.uint32).reshape(E, 624))
            sidx = np.ascontiguousarray(np.asarray(seeds[1], dtype=np.int32).reshape(E))
        else:
            seeds = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64).reshape(E))
        p = RRTParams()
        p.dist_to_end, p.diff_max, p.freq, p.min_dist = float(dist_to_end), float(diff_max), float(freq), float(min_dist)
        p.bin_interval, p.v, p.max_traj_time = float(bin_interval), float(v), float(max_traj_time)
        p.max_plan_time = float(n_iter if max_plan_time is None else max_plan_time)
        for i in range(3):
            p.w[i] = float(weights[i])
        p.mode, p.max_iter, p.points_per_iter = MODES[mode], int(n_iter), float(points_per_iter)
        flags = (FLAG_ITER_LOG if iter_log else 0) | (FLAG_LEAF_LOG if leaf_log else 0) | (FLAG_PHASE_CLOCKS if phase_clocks else 0)
        if states is not None:
            self._chk(self.L.auvp_rrt_prepare_states(self.h, E, _p(init), states.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                     _p(sidx, _ip), C.byref(p), flags))
        else:
            self._chk(self.L.auvp_rrt_prepare(self.h, E, _p(init), seeds.ctypes.data_as(C.POINTER(C.c_uint64)),
                                              C.byref(p), flags))
        self.n_episodes, self.max_iter = E, int(n_iter)

    def summaries(self):
        out = np.zeros(self.n_episodes, dtype=SUMMARY_DTYPE)
        self._chk(self.L.auvp_rrt_summaries(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def paths(self, summaries):
        """final course (root -> leaf) of every episode: list of [L,7] arrays"""
        lens = np.where(summaries["best_leaf"] >= 0, summaries["best_path_len"], 0).astype(np.int64)
        off = np.zeros(self.n_episodes + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        out = np.zeros((max(int(off[-1]), 1), 7))
        self._chk(self.L.auvp_rrt_paths(self.h, off.ctypes.data_as(C.POINTER(C.c_int64)), _p(out)))
        return [out[off[e]:off[e + 1]] for e in range(self.n_episodes)]

    def paths_dev(self, offsets, out_dev_ptr):
        """final courses written to a caller-owned device buffer [offsets[E],7] f64 (e.g. a torch tensor's
        data_ptr()) -- stays in HBM for the multi-GPU gather"""
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        self._chk(self.L.auvp_rrt_paths_dev(self.h, off.ctypes.data_as(C.POINTER(C.c_int64)), C.c_void_p(out_dev_ptr)))

    def tree(self, ep, summary):
        n, npnt = int(summary["n_nodes"]), int(summary["n_points"])
        nodes = np.zeros((n, 6))
        parent = np.zeros(n, np.int32)
        pt_off = np.zeros(n, np.int32)
        pt_cnt = np.zeros(n, np.int32)
        points = np.zeros((max(npnt, 1), 7))
        self._chk(self.L.auvp_rrt_tree(self.h, ep, _p(nodes), _p(parent, _ip), _p(pt_off, _ip), _p(pt_cnt, _ip), _p(points)))
        return dict(nodes=nodes, parent=parent, pt_off=pt_off, pt_cnt=pt_cnt, points=points[:npnt])

    def iter_log(self, ep):
        n = self.max_iter
        a, b, c = np.zeros(n, np.int32), np.zeros(n, np.int8), np.zeros(n, np.int32)
        self._chk(self.L.auvp_rrt_iter_log(self.h, ep, _p(a, _ip), _p(b, _bp), _p(c, _ip)))
        return dict(it_parent=a, it_accepted=b, it_npath=c)

    def leaf_log(self, ep, summary):
        n = int(summary["n_leaves"])
        lc, li = np.zeros((max(n, 1), 6)), np.zeros(max(n, 1), np.int32)
        self._chk(self.L.auvp_rrt_leaf_log(self.h, ep, _p(lc), _p(li, _ip)))
        return lc[:n], li[:n]

    def bin_sizes(self, ep):
        k = C.c_int32()
        self._chk(self.L.auvp_rrt_bin_sizes(self.h, ep, None, C.byref(k)))
        s = np.zeros(max(k.value, 1), np.int32)
        self._chk(self.L.auvp_rrt_bin_sizes(self.h, ep, _p(s, _ip), C.byref(k)))
        return s[:k.value]

    # ---- probes ----
    def check_collision(self, paths_xy):
        off = np.zeros(len(paths_xy) + 1, np.int32)
        off[1:] = np.cumsum([len(p) for p in paths_xy])
        pts = _f64(np.concatenate([_f64(p, (-1, 2)) for p in paths_xy]) if len(paths_xy) else [], (-1, 2))
        out = np.zeros(len(paths_xy), np.int8)
        self._chk(self.L.auvp_check_collision_batch(self.h, len(paths_xy), _p(off, _ip), _p(pts), _p(out, _bp)))
        return out.astype(bool)

    def cost_paths(self, paths_xyt, bin_lo, bin_hi, total, weights):
        n = len(paths_xyt)
        off = np.zeros(n + 1, np.int32)
        off[1:] = np.cumsum([len(p) for p in paths_xyt])
        pts = _f64(np.concatenate([_f64(p, (-1, 3)) for p in paths_xyt]), (-1, 3))
        lo = np.ascontiguousarray(bin_lo, dtype=np.int32)
        hi = np.ascontiguousarray(bin_hi, dtype=np.int32)
        tt = _f64(total)
        w = _f64(weights, (n, 3))
        out = np.zeros((n, 4))
        self._chk(self.L.auvp_cost_paths(self.h, n, _p(off, _ip), _p(pts), _p(lo, _ip), _p(hi, _ip), _p(tt), _p(w), _p(out)))
        return out

    def nn_closest(self, xy, queries, force_exact=False):
        """get_closest_mps (rrt_dubins.py:505-513) of every query point against the node list xy -> (index, slow-path flag)"""
        xy, q = _f64(xy, (-1, 2)), _f64(queries, (-1, 2))
        idx, slow = np.zeros(len(q), np.int32), np.zeros(len(q), np.int32)
        self._chk(self.L.auvp_nn_closest_batch(self.h, len(xy), _p(xy), len(q), _p(q), 1 if force_exact else 0, _p(idx, _ip), _p(slow, _ip)))
        return idx, slow.astype(bool)

    def sincos(self, x):
        x = _f64(x).ravel()
        s, c = np.zeros_like(x), np.zeros_like(x)
        self._chk(self.L.auvp_sincos_dev(self.h, len(x), _p(x), _p(s), _p(c)))
        return s, c

    MATH_OPS = {"sincos": 0, "atan2": 1, "pow_e": 2, "div_plain": 3, "sqrt_plain": 4, "hypot": 5, "div": 6, "sqrt": 7}

    def math(self, op, a, b=None):
        """auvp_math_dev: one portable elementary function on the device (bit-exactness probe)"""
        a = _f64(a).ravel()
        b = np.zeros_like(a) if b is None else _f64(b).ravel()
        o0, o1 = np.zeros_like(a), np.zeros_like(a)
        self._chk(self.L.auvp_math_dev(self.h, self.MATH_OPS[op], len(a), _p(a), _p(b), _p(o0), _p(o1)))
        return (o0, o1) if op == "sincos" else o0

    def random_stream(self, seed, n):
        out = np.zeros(n)
        self._chk(self.L.auvp_random_stream_dev(self.h, int(seed), n, _p(out)))
        return out

    def phase_clocks(self):
        out = np.zeros((self.n_episodes, 5), dtype=np.uint64)
        self._chk(self.L.auvp_rrt_phase_clocks(self.h, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    def prrt_last_kernel(self):
        """name of the kernel the last Planner_RRT launch ran (prrt_kernel: one episode per wavefront; prrt_rows_kernel: four)"""
        self.L.auvp_prrt_last_kernel.restype = C.c_char_p
        self.L.auvp_prrt_last_kernel.argtypes = [C.c_void_p]
        return (self.L.auvp_prrt_last_kernel(self.h) or b"").decode()

    def last_kernel_ms(self):
        return float(self.L.auvp_last_kernel_ms(self.h))

    def last_launch_parts(self):
        """(expansion ms, leaf-pass ms, episodes per wavefront) of the last rrt_run"""
        a, b, k = C.c_double(), C.c_double(), C.c_int32()
        self._chk(self.L.auvp_rrt_last_launch_parts(self.h, C.byref(a), C.byref(b), C.byref(k)))
        return a.value, b.value, k.value

    def last_stream_ms(self):
        """ms of the launch that generated the random numbers ahead of the last rrt_run's expansion kernel (0: it did not)"""
        return float(self.L.auvp_rrt_last_stream_ms(self.h))

    def last_stream_len(self):
        """random() numbers per episode that launch wrote (0: none)"""
        return int(self.L.auvp_rrt_last_stream_len(self.h))

    def last_rrt_kernel(self):
        """name of the expansion kernel the last rrt_run launched (rrt_rows_kernel / rrt_explore_kernel / rrt_duo_kernel)"""
        return (self.L.auvp_rrt_last_kernel(self.h) or b"").decode()

    def last_leaf_stats(self):
        """of the last rrt_run's leaf pass, summed over the batch: nodes visited, their path points, path elements re-summed
        in the reference's order, leaves re-summed"""
        out = (C.c_int64 * 4)()
        self._chk(self.L.auvp_rrt_last_leaf_stats(self.h, out))
        return dict(zip(("nodes_visited", "points_visited", "elements_resummed", "leaves_resummed"), (int(v) for v in out)))

    def hbm_probe(self, n_bytes=4 << 30, reps=3):
        """measured HBM streaming rates of this GPU in GB/s: (read, copy [read + written bytes])"""
        r, c = C.c_double(), C.c_double()
        self._chk(self.L.auvp_hbm_probe(self.h, int(n_bytes), int(reps), C.byref(r), C.byref(c)))
        return r.value, c.value

    def last_launch(self):
        g, b, l = C.c_int32(), C.c_int32(), C.c_int32()
        self.L.auvp_last_launch(self.h, C.byref(g), C.byref(b), C.byref(l))
        return g.value, b.value, l.value
