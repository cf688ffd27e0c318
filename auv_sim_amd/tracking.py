"""Shark tracking with one path replan per particle hypothesis (BASELINE.json configs[4]).

The reference chains these by hand in robotSim.py:665-701: every tracking step the particle filter
(particleFilter.py:283-317) moves and re-weights its particles from the AUVs' range/bearing measurements, and a
planner is asked for a path towards the shark estimate.  Config 5 scales that out: EVERY particle (a hypothesis of
where the shark is) becomes the goal of its own Planner_RRT episode (gym_rrt/envs/rrt_dubins.py:162-248), F filters x
N particles per GPU.  Both stages stay on the device: the particle state the filter kernel leaves in HBM is what
the planner batch reads its goals from (auvp_prrt_replan_particles); nothing but the per-step measurements goes in
and nothing but the result records comes out.
"""
import numpy as np

from . import _pf_lib
from ._prrt_lib import PlannerBatch


def range_bearing_measurements(tracks, auv_offsets, auv_theta=0.0):
    """Noise-free sensor model over recorded tracks: AUV a of filter f sits at tracks[f, 0] + auv_offsets[a]
    (x, y, theta fixed) and measures range and bearing to the shark at every sample.
    tracks [F, S, 2] -> meas [S, F, A, 5] = auv x, auv y, auv theta, range, bearing (the Z_shark the reference's
    update_weights reads, particleFilter.py:283-301) and shark_xy [S, F, 2]."""
    tracks = np.asarray(tracks, dtype=np.float64)
    F, S, _ = tracks.shape
    off = np.asarray(auv_offsets, dtype=np.float64).reshape(-1, 2)
    A = len(off)
    meas = np.zeros((S, F, A, 5))
    auv = tracks[:, 0, None, :] + off[None, :, :]          # [F, A, 2]
    d = tracks.transpose(1, 0, 2)[:, :, None, :] - auv[None]  # [S, F, A, 2]
    meas[..., 0:2] = auv[None]
    meas[..., 2] = auv_theta
    meas[..., 3] = np.sqrt(d[..., 0] ** 2 + d[..., 1] ** 2)
    meas[..., 4] = np.arctan2(d[..., 1], d[..., 0]) - auv_theta
    return meas, np.ascontiguousarray(tracks.transpose(1, 0, 2))


def goal_transforms(tracks, rect, margin=20.0):
    """per filter: map the bounding box of its track (grown by 10 %) into the planner rectangle inset by `margin`,
    same scale on both axes.  Returns xform [F, 4] = sx, ox, sy, oy and the clamp rectangle."""
    tracks = np.asarray(tracks, dtype=np.float64)
    lo, hi = tracks.min(axis=1), tracks.max(axis=1)
    span = np.maximum((hi - lo).max(axis=1) * 1.1, 1.0)
    room = min(rect[2] - rect[0], rect[3] - rect[1]) - 2.0 * margin
    s = room / span
    mid = 0.5 * (lo + hi)
    cx, cy = 0.5 * (rect[0] + rect[2]), 0.5 * (rect[1] + rect[3])
    xform = np.stack([s, cx - mid[:, 0] * s, s, cy - mid[:, 1] * s], axis=1)
    clamp = np.array([rect[0] + 5.0, rect[1] + 5.0, rect[2] - 5.0, rect[3] - 5.0])
    return np.ascontiguousarray(xform), clamp


class ParticleReplanner:
    """F filters x N particles on one GPU; step(s) = one tracking step (filter update from the measurements of sample
    s) followed by one Planner_RRT.planning(max_step) per particle towards that particle.

    ctx          auv_sim_amd._lib.Context whose world holds the obstacle list of the planning workspace
    tracks       [F, S, 2] recorded shark positions, one track per filter
    filter_seeds [F] np.random.seed values of the filters (numpy legacy stream, as particleFilter.py draws)
    episode_seed_base  episode e of tracking step s plans with random.seed(episode_seed_base + s * F * N + e)
    """

    def __init__(self, ctx, tracks, n_particles, rect, start, filter_seeds, auv_offsets=((60.0, -40.0), (-50.0, 70.0)),
                 max_step=200, freq=10, cell=5, subs=1, episode_seed_base=0, episode_offset=0, episodes_total=None):
        self.ctx = ctx
        self.tracks = np.asarray(tracks, dtype=np.float64)
        self.F, self.S_total = self.tracks.shape[0], self.tracks.shape[1]
        self.N = int(n_particles)
        self.E = self.F * self.N
        self.rect, self.start = tuple(float(v) for v in rect), tuple(float(v) for v in start)
        self.kw = dict(max_step=int(max_step), freq=freq, cell=cell, subs=subs)
        self.meas, self.shark = range_bearing_measurements(self.tracks, auv_offsets)
        self.xform, self.clamp = goal_transforms(self.tracks, self.rect)
        self.seed_base = int(episode_seed_base)
        self.episode_offset = int(episode_offset)                 # this rank's first global episode id
        self.episodes_total = int(episodes_total or self.E)       # over all ranks: seeds do not depend on the sharding
        mts = np.stack([_pf_lib.np_seed_state(int(s))[0] for s in filter_seeds])
        self.filters = _pf_lib.FilterBatch(ctx, self.F, self.N).create(self.tracks[:, 0, :], mts, 624)
        self.pf_ms = self.replan_ms = self.plan_ms = 0.0
        self.planner = None

    def episode_seed(self, s, e_local=0):
        return self.seed_base + s * self.episodes_total + self.episode_offset + e_local

    def step(self, s):
        """tracking step s: filter update -> goals from particles -> plan.  Returns the planner summaries [E]."""
        self.filters.run(meas=self.meas[s:s + 1], shark_xy=self.shark[s:s + 1])
        self.pf_ms = self.ctx.last_kernel_ms()
        self.planner = PlannerBatch.from_particles(self.ctx, self.E, list(self.start) + [0.0, 0.0], self.rect,
                                                   self.kw["max_step"], self.xform, self.clamp, self.episode_seed(s),
                                                   freq=self.kw["freq"], cell=self.kw["cell"], subs=self.kw["subs"])
        self.replan_ms = self.ctx.last_kernel_ms()
        summ = self.planner.plan()
        self.plan_ms = self.ctx.last_kernel_ms()
        return summ
