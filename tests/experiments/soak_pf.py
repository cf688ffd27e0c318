"""Randomised GPU-vs-checker sweep of the shark particle filter (pf_step_kernel): particle counts 2 .. 2048, 1 .. 4 AUV
measurements, 1 .. 12 steps, mid-block generator positions, measurements that concentrate the weights (few survivors: long
object-sharing chains) or spread them.  usage: python tests/experiments/soak_pf.py [n_cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from auv_sim_amd import _lib, _pf_lib  # noqa: E402
from oracle import orc_pf  # noqa: E402


def one_case(ctx, rng):
    N = int(rng.choice([2, 3, 17, 64, 100, 255, 256, 257, 500, 1000, 1000, 1024, 1025, 1500, 2048]))
    A, S, F = int(rng.integers(1, 5)), int(rng.integers(1, 13)), int(rng.integers(1, 7))
    seeds = rng.integers(0, 2 ** 32, size=F)
    shark0 = rng.uniform(-500, 500, size=(F, 2))
    spread = float(rng.choice([5.0, 50.0, 200.0, 600.0]))
    meas = np.zeros((S, F, A, 5))
    meas[..., 0:2] = shark0[None, :, None, :] + rng.uniform(-spread, spread, size=(S, F, A, 2))
    meas[..., 2] = rng.uniform(-np.pi, np.pi, size=(S, F, A))
    meas[..., 3] = rng.uniform(0, float(rng.choice([3.0, 60.0, 300.0, 900.0])), size=(S, F, A))
    meas[..., 4] = rng.uniform(-np.pi, np.pi, size=(S, F, A))
    shark = shark0[None] + rng.uniform(-30, 30, size=(S, F, 2))
    mts = np.stack([_pf_lib.np_seed_state(int(s))[0] for s in seeds])
    pos = rng.integers(0, 625, size=F).astype(np.int32)
    b = _pf_lib.FilterBatch(ctx, F, N).create(shark0, mts, pos)
    b.run(meas=meas, shark_xy=shark, log=True)
    upd, cho = b.step_log()
    mean, err, ll = b.estimates()
    final, obj = b.particles()
    st, nd = b.status()
    mt1, pos1 = b.rng_state()
    bad = 0
    for f in range(F):
        ref = orc_pf.run(N, meas[:, f], shark[:, f], shark0[f], mts[f], int(pos[f]), kind="portable")
        if ref["status"] != 0 or st[f] != 0:
            bad += int(ref["status"] != st[f])
            continue
        ok = (np.array_equal(upd[:, f], ref["updated"]) and np.array_equal(cho[:, f], ref["choice"]) and np.array_equal(ll[:, f], ref["list_len"])
              and np.array_equal(final[f], ref["resampled"][-1]) and np.array_equal(mean[:, f], ref["mean"]) and np.array_equal(err[:, f], ref["range_error"])
              and np.array_equal(mt1[f], ref["mt"]) and pos1[f] == ref["mt_pos"] and int(nd[f]) == ref["n_draw32"])
        bad += 0 if ok else 1
    return bad, (N, A, S, F)


def main(n_cases=60, seed=1):
    ctx = _lib.Context(0)
    rng = np.random.default_rng(seed)
    bad = 0
    for c in range(n_cases):
        b, cfg = one_case(ctx, rng)
        if b:
            print("MISMATCH case", c, cfg, b)
        bad += b
    print("pf soak: %d cases, %d mismatching filters" % (n_cases, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(*(int(a) for a in sys.argv[1:3])) else 0)
