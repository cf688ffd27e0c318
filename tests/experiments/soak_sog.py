"""Randomised GPU-vs-checker sweep of SharkOccupancyGrid.convert: grid shapes around the 16 x 64 tile edges, sparse / duplicated /
shuffled cell lists, points on cell and bin edges, window radii 1 .. 14 cells, the three window-sum kernels (AUVP_SOG_TILE).
usage: python tests/experiments/soak_sog.py [n_cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from auv_sim_amd import _lib  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import env_options  # noqa: E402  (tests/env_options.py: AUVP_<NAME> in os.environ steers live contexts -- this process only)
env_options.install()
from auv_sim_amd.sharkOccupancyGrid import convert_arrays  # noqa: E402
from oracle import orc_sog  # noqa: E402


def one_case(ctx, rng):
    cs = float(rng.choice([2.0, 3.0, 4.0, 7.5]))
    ncol, nrow = int(rng.choice([3, 15, 16, 17, 40, 63, 64, 65, 70])), int(rng.choice([2, 15, 16, 17, 33]))
    x0, y0 = float(rng.uniform(-50, 50)), float(rng.uniform(-50, 50))
    box = (x0, y0, x0 + ncol * cs, y0 + nrow * cs)
    cells = np.array([[box[0] + c * cs, box[1] + r * cs, box[0] + (c + 1) * cs, box[1] + (r + 1) * cs] for r in range(nrow) for c in range(ncol)])
    cells = cells[rng.random(len(cells)) < float(rng.choice([0.3, 0.8, 1.0]))]
    if len(cells) == 0:
        cells = np.array([[box[0], box[1], box[0] + cs, box[1] + cs]])
    if rng.random() < 0.5:
        cells = np.concatenate([cells, cells[: int(rng.integers(1, 4))]])
    cells = cells[rng.permutation(len(cells))]
    n_sharks = int(rng.integers(1, 6))
    traj_len = rng.integers(1, 300, size=n_sharks).astype(np.int32)
    pts = []
    for n in traj_len:
        t = np.sort(rng.uniform(0.0, 70.0, size=n))
        x = rng.uniform(box[0] - 5, box[2] + 5, size=n)
        y = rng.uniform(box[1] - 5, box[3] + 5, size=n)
        snap = rng.random(n) < 0.2
        x[snap] = box[0] + cs * np.round((x[snap] - box[0]) / cs)
        snap = rng.random(n) < 0.2
        y[snap] = box[1] + cs * np.round((y[snap] - box[1]) / cs)
        t[rng.random(n) < 0.1] = 20.0
        pts.append(np.stack([x, y, np.sort(t)], axis=1))
    pts = np.concatenate(pts)
    detect = cs * float(rng.integers(1, 15)) - float(rng.choice([0.0, 0.3]))
    ref = orc_sog.convert(cells, box, cs, 10.0, detect, traj_len, pts, kind="portable")
    bad = 0
    for mode in ("0", "1", "2"):
        os.environ["AUVP_SOG_TILE"] = mode
        bins, grids = convert_arrays(ctx, cells, box, cs, 10.0, detect, traj_len, pts)
        bad += 0 if (np.array_equal(bins, ref["bins"]) and np.array_equal(grids, ref["grids"])) else 1
    os.environ.pop("AUVP_SOG_TILE", None)
    return bad, (cs, ncol, nrow, len(cells), n_sharks, detect)


def main(n_cases=40, seed=1):
    ctx = _lib.Context(0)
    rng = np.random.default_rng(seed)
    bad = 0
    for c in range(n_cases):
        b, cfg = one_case(ctx, rng)
        if b:
            print("MISMATCH case", c, cfg, b)
        bad += b
    print("sog soak: %d cases x 3 kernels, %d mismatches" % (n_cases, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(*(int(a) for a in sys.argv[1:3])) else 0)
