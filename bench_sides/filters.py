"""bench_sides/filters.py -- the shark particle filter and SharkOccupancyGrid.convert  (split out of bench.py in round 6)"""
import json
import os
import sys
import time

import numpy as np

from .common import *  # noqa: F401,F403
from .common import _rrt_batch  # noqa: F401


def bench_particle_filter(device, with_cpu, n_filters=4096, n_particles=1000, n_steps=20, n_auv=2, reps=3):
    """F shark particle filters x 1000 particles x S steps of create_and_update + update_weights
    + particleMean/meanError (robotSim.py:665-701) in one launch; filter f continues np.random.seed(f)."""
    from auv_sim_amd import _lib, _pf_lib
    ctx = _lib.Context(device)
    rng = np.random.default_rng(4)
    F, N, S, A = n_filters, n_particles, n_steps, n_auv
    shark0 = rng.uniform(-500, 500, size=(F, 2))
    meas = np.zeros((S, F, A, 5))
    meas[..., 0:2] = shark0[None, :, None, :] + rng.uniform(-150, 150, size=(S, F, A, 2))
    meas[..., 2] = rng.uniform(-np.pi, np.pi, size=(S, F, A))
    meas[..., 3] = rng.uniform(0, 200, size=(S, F, A))
    meas[..., 4] = rng.uniform(-np.pi, np.pi, size=(S, F, A))
    shark = shark0[None] + rng.uniform(-20, 20, size=(S, F, 2))
    key0, _ = _pf_lib.np_seed_state(0)
    mts = np.stack([_pf_lib.np_seed_state(f)[0] if f < 64 else np.roll(key0, f) ^ np.uint32(f) for f in range(F)])
    ms = []
    for i in range(reps + 1):
        b = _pf_lib.FilterBatch(ctx, F, N).create(shark0, mts, 624)
        b.run(meas=meas, shark_xy=shark)
        if i:
            ms.append(ctx.last_kernel_ms())
    st, nd = b.status()
    if (st != 0).any():
        return {"error": "filter status %s" % np.unique(st)}
    k_ms = float(np.mean(ms))
    units = float(F) * N * S
    abytes = F * N * 2 * 44.0 + S * F * (A * 40.0 + 28.0)  # particles in and out once per launch; measurements in, estimates out per step
    pf_traffic = pmc_traffic("particle_filter", ["pf_step_kernel"], None)
    out = {"metric": "particle filter particle-steps/s (create_and_update + update_weights)", "value": units / (k_ms * 1e-3),
           "unit": "particle-steps/s", "filters": F, "particles": N, "steps": S, "auvs": A, "kernel_ms": k_ms,
           "draws32_per_filter_step": float(nd.mean()) / S,
           "config": "%d filters x %d particles x %d steps, %d AUV measurements per step" % (F, N, S, A),
           "roofline": roofline(abytes, k_ms, "pf_step_kernel", pf_traffic,
                                valu_issue_frac=pmc_valu_issue("particle_filter", "pf_step_kernel") if pf_traffic["traffic"] is not None else None,
                                note="state is LDS resident across the steps of a launch; four wavefronts per SIMD of dependent fp64 "
                                     "chains (atan2, exponentials, MT19937 blocks, ordered sums) between ~45 workgroup barriers per "
                                     "step: issue and latency bound, not HBM")}
    if with_cpu:
        from oracle import orc_pf
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < 4.0 and n < F:
            orc_pf.run(N, meas[:, n], shark[:, n], shark0[n], mts[n], 624, kind="libm")
            n += 1
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n * N * S / dt, "unit": "particle-steps/s", "cores": 1, "kind": "port",
                               "sample": "first %d of the %d filters, oracle/ libm build, %.1f s" % (n, F, dt)}
    return out


def bench_shark_grid(device, with_cpu, n_side=200, n_sharks=32, n_pts=3000, reps=3):
    """SharkOccupancyGrid.convert, 10 m cells over 2 km x 2 km, 32 sharks x 3000 points, 10 bins of
    30 s, detection range 50 m (the reference's constructor arguments at rrt_dubins.py:68)."""
    from auv_sim_amd import _lib
    from auv_sim_amd.sharkOccupancyGrid import convert_arrays
    ctx = _lib.Context(device)
    rng = np.random.default_rng(7)
    cs, n = 10.0, n_side
    box = (0.0, 0.0, cs * n, cs * n)
    cx, cy = np.meshgrid(np.arange(n), np.arange(n))
    cells = np.stack([cx.ravel() * cs, cy.ravel() * cs, (cx.ravel() + 1) * cs, (cy.ravel() + 1) * cs], axis=1)
    traj_len = np.full(n_sharks, n_pts, dtype=np.int32)
    t = np.tile(np.arange(1, n_pts + 1) * 0.1, n_sharks)
    pts = np.stack([rng.uniform(1, cs * n - 2, len(t)), rng.uniform(1, cs * n - 2, len(t)), t], axis=1)
    ms = []
    for i in range(reps + 1):
        bins, grids = convert_arrays(ctx, cells, box, cs, 30.0, 50.0, traj_len, pts)
        if i:
            ms.append(ctx.last_kernel_ms())
    k_ms = float(np.mean(ms))
    T, G = grids.shape[0], grids.shape[1] * grids.shape[2]
    units = float(T) * G
    # compulsory traffic: the points once (24 B), the per-(bin, shark, cell) occupancy written and read once (8 + 8 B; the
    # 81-cell window re-reads of the disc stencil come from LDS tiles and are NOT counted), the output once (8 B)
    abytes = len(pts) * 24.0 + T * n_sharks * G * (4.0 + 8.0 + 8.0) + units * 8.0
    out = {"metric": "SharkOccupancyGrid.convert output cells/s", "value": units / (k_ms * 1e-3), "unit": "grid cells/s",
           "bins": int(T), "grid": [int(grids.shape[1]), int(grids.shape[2])], "sharks": n_sharks, "kernel_ms": k_ms,
           "config": "%dx%d cells of 10 m, %d sharks x %d points, %d bins, detect range 50 m" % (n, n, n_sharks, n_pts, T),
           "roofline": roofline(abytes, k_ms, "sog_count/occ/grid_kernel",
                                pmc_traffic("shark_grid", ["sog_count_kernel", "sog_occ_kernel", "sog_grid_tile_c_kernel"], None),
                                note="compulsory bytes only (three launches: count, occupancy, window sums); the 81-cell disc "
                                     "windows are summed from LDS tiles, one LDS read per four of the %.1f G ordered fp64 additions"
                                     % (units * n_sharks * 81 / 1e9))}
    if with_cpu:
        from oracle import orc_sog
        sub = 2
        k = sub * n_pts
        t0 = time.perf_counter()
        r = orc_sog.convert(cells, box, cs, 30.0, 50.0, traj_len[:sub], pts[:k], kind="libm")
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": float(len(r["grids"])) * G * (sub / float(n_sharks)) / dt, "unit": "grid cells/s",
                               "cores": 1, "kind": "port",
                               "sample": "%d of the %d sharks (cost is linear in sharks; value scaled by %d/%d), %.1f s"
                                         % (sub, n_sharks, sub, n_sharks, dt)}
    return out


# ----------------------------------------------------------------------------------------------------------------
