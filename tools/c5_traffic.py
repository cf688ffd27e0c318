#!/usr/bin/env python3
"""Config 5 (12 500 Planner_RRT replans x 200 steps per tracking step): where the plan launch's HBM bytes go, array by array.
Run on a GPU box (python tools/c5_traffic.py [--md out.md]).  Runs one warm-up and one measured tracking step of bench.py's
config-5 setup, downloads the result records of every episode and the trees of a sample, and prints per array of
PrrtBuffers (planner_rrt_kernel.h): the bytes the launch touches (algorithmic), the same in whole 64-byte sectors (what a
perfect cache in front of HBM would still move: first read / final write-back of every sector touched), and the bytes if
every access moved its own sector (no cache at all).  Compare with FETCH_SIZE / WRITE_SIZE of the same launch
(profiles/pmc_latest.json@config5)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from auv_sim_amd import _lib, synth, tracking  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--md", default="")
ap.add_argument("--sample", type=int, default=200)
args = ap.parse_args()

F, N, max_step = 25, 500, 200
xy = np.load(os.path.join(bench.REPO, "tests", "golden", "shark_tracking_xy.npz"))["xy"]
w = synth.make_rect_world(seed=3, n_obstacles=256)
ctx = _lib.Context(0)
ctx.set_world(obstacles=w["obstacles"])
gf = np.arange(F)
rp = tracking.ParticleReplanner(ctx, xy[gf % 32, :3], N, w["rect"], w["start"], gf, max_step=max_step)
rp.step(0)
summ = rp.step(1)
E = rp.E
pb = rp.planner
nb = pb.rows * pb.cols * pb.subs
steps = float(summ["steps"].sum())
nodes = float((summ["n_nodes"] - 1).sum())
pts = float(summ["n_points"].sum())
SEC = 64.0


def sectors(n_bytes_contiguous):
    return np.ceil(np.asarray(n_bytes_contiguous, dtype=np.float64) / SEC) * SEC


# --- sample of trees: buckets touched, hops along the member lists, rejected steers' points
rng = np.random.default_rng(0)
sample = rng.choice(E, size=min(args.sample, E), replace=False)
bucket_sec, hop_reads, spec_pts = [], [], []
for e in sample:
    t = pb.tree(int(e), summ[int(e)])
    bk = np.asarray(t["node_bucket"])
    bucket_sec.append(len(np.unique(bk[bk >= 0] // 8)))  # 8-byte words: eight buckets per 64-B sector
scale = E / float(len(sample))
bsec = float(np.sum(bucket_sec)) * scale  # distinct 64-B sectors of the [n_buckets] table an episode's accepted nodes fall into

rows = []


def row(name, what, alg_r, alg_w, sec_r, sec_w, acc_r, acc_w):
    rows.append((name, what, alg_r, alg_w, sec_r, sec_w, acc_r, acc_w))


# generator state: read at pickup, written back at the end (plan mode: the write-back only serves a later continuation)
row("mt + rng_state", "624 words + 4 per episode, read at pickup, stored at the end", E * 2512.0, E * 2512.0, E * 2560.0, E * 2560.0, E * 2560.0, E * 2560.0)
row("summary", "136-B record per episode, read at pickup, stored at the end", E * 136.0, E * 136.0, E * 192.0, E * 192.0, E * 192.0, E * 192.0)
row("goal + start", "16 + 32 B per episode", E * 48.0, 0.0, E * 128.0, 0.0, E * 128.0, 0.0)
# per step: the 8-byte {size, head} word of the chosen bucket and of the new node's bucket (requested before the collision test)
row("buckets ({size | epoch, head}, 8 B per bucket)", "the chosen bucket's and the new node's bucket's word per step; one word stored per accepted node",
    steps * 16.0, nodes * 8.0, 2 * bsec * SEC, 2 * bsec * SEC, steps * 2 * SEC, nodes * SEC)
row("nodes (64-B records)", "parent state 32 B per step, member-list hops 4 B each (~1 per step), the last node's state when the step's node was "
    "rejected, links read by the final path walk; one 64-B record stored per accepted node (one store instruction)",
    steps * 36.0, nodes * 64.0, (nodes + E) * SEC, (nodes + E) * SEC, steps * 2 * SEC, nodes * SEC)
row("occupied", "4 B stored per first member of a bucket (the list itself is read from its LDS copy)", 0.0, float(summ["n_occ"].sum()) * 4.0,
    0.0, np.ceil(summ["n_occ"] * 4 / SEC).sum() * SEC, 0.0, float(summ["n_occ"].sum()) * SEC)
row("points (32-B records)", "x, y, theta, t of every taken sub-arc, stored speculatively per step (rejected steers are overwritten)",
    0.0, pts * 32.0, 0.0, np.ceil(pts / E * 32 / SEC) * SEC * E, 0.0, steps * 3 * SEC)
row("obstacle tile (os_*)", "7 KB shared by every episode: L1 / L2 hits", 0.0, 0.0, 7168.0, 0.0, 0.0, 0.0)

tot = np.array([[r[2], r[3], r[4], r[5], r[6], r[7]] for r in rows]).sum(axis=0)
lines = []
lines.append("| array | what the launch does with it | bytes touched: read / written (MB) | whole 64-B sectors, each once: read / written (MB) | every access its own sector: read / written (MB) |")
lines.append("|---|---|---|---|---|")
for r in rows:
    lines.append("| `%s` | %s | %.1f / %.1f | %.1f / %.1f | %.1f / %.1f |" % (r[0], r[1], r[2] / 1e6, r[3] / 1e6, r[4] / 1e6, r[5] / 1e6, r[6] / 1e6, r[7] / 1e6))
lines.append("| **sum** | | **%.1f / %.1f** | **%.1f / %.1f** | **%.1f / %.1f** |" % tuple(tot / 1e6))
hdr = ("config 5 plan launch (%s): %d episodes, %.0f planner steps, %.0f accepted nodes (%.1f per episode), %.0f stored path points "
       "(%.1f per episode), %d buckets per episode; SURVEY 8(d) algorithmic bytes (bench.planner_bytes) = %.1f MB; "
       "the bucket table (%.1f MB) is not cleared between batches (epoch tag)"
       % (ctx.prrt_last_kernel(), E, steps, nodes, nodes / E, pts, pts / E, nb, bench.planner_bytes(summ) / 1e6, E * nb * 8 / 1e6))
out = hdr + "\n\n" + "\n".join(lines) + "\n"
print(out)
print("plan launch %.3f ms, replan (seeding + planting) %.3f ms, filter %.3f ms" % (rp.plan_ms, rp.replan_ms, rp.pf_ms))
if args.md:
    open(args.md, "w").write(out)
