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
    bucket_sec.append(len(np.unique(bk[bk >= 0] // 16)))
scale = E / float(len(sample))
bsec = float(np.sum(bucket_sec)) * scale  # distinct 64-B sectors of a [n_buckets] int32 array an episode's accepted nodes fall into

rows = []


def row(name, what, alg_r, alg_w, sec_r, sec_w, acc_r, acc_w):
    rows.append((name, what, alg_r, alg_w, sec_r, sec_w, acc_r, acc_w))


# generator state: read at pickup, written back at the end (plan mode: the write-back only serves a later continuation)
row("mt + rng_state", "624 words + 4 per episode, read at pickup, stored at the end", E * 2512.0, E * 2512.0, E * 2560.0, E * 2560.0, E * 2560.0, E * 2560.0)
row("summary", "136-B record per episode, read at pickup, stored at the end", E * 136.0, E * 136.0, E * 192.0, E * 192.0, E * 192.0, E * 192.0)
row("goal + start", "16 + 32 B per episode", E * 48.0, 0.0, E * 128.0, 0.0, E * 128.0, 0.0)
# per step: bucket size + head of the chosen bucket, of the new node's bucket (accepted or not: requested before the collision test)
row("bucket_counts + bucket_head (two arrays)", "4 + 4 B of the chosen bucket and of the new node's bucket per step; 4 + 4 B stored per accepted node",
    steps * 16.0, nodes * 8.0, 2 * bsec * SEC, 2 * bsec * SEC, steps * 4 * SEC, nodes * 2 * SEC)
row("node_next", "member-list hops of the node choice (~1 per step) + 4 B stored per node", steps * 4.0, nodes * 4.0, sectors(nodes * 4 / E).sum() * E / 1.0 if False else np.ceil(nodes / E * 4 / SEC) * SEC * E,
    np.ceil(nodes / E * 4 / SEC) * SEC * E, steps * SEC, nodes * SEC)
row("node_f", "parent record 32 B per step (+ the last node's 24 B when the step's node was rejected); 32 B stored per node",
    steps * 32.0, nodes * 32.0, np.ceil((nodes / E + 1) * 32 / SEC) * SEC * E, np.ceil((nodes / E + 1) * 32 / SEC) * SEC * E, steps * SEC, nodes * SEC)
row("node_i + node_bucket + occupied", "16 + 4 (+ 4 for a first bucket member) B stored per node; links read by the final path walk", 0.0, nodes * 24.0,
    0.0, (np.ceil((nodes / E + 1) * 16 / SEC) + np.ceil((nodes / E + 1) * 4 / SEC) * 2) * SEC * E, 0.0, nodes * 3 * SEC)
row("points (4 SoA columns)", "x, y, theta, t of every taken sub-arc, stored speculatively per step (rejected steers are overwritten)",
    0.0, pts * 32.0, 0.0, 4 * np.ceil(pts / E * 8 / SEC) * SEC * E, 0.0, steps * 4 * SEC)
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
       "memset of bucket_counts before the launch: %.1f MB (not in the launch's counters)"
       % (ctx.prrt_last_kernel(), E, steps, nodes, nodes / E, pts, pts / E, nb, bench.planner_bytes(summ) / 1e6, E * nb * 4 / 1e6))
out = hdr + "\n\n" + "\n".join(lines) + "\n"
print(out)
print("plan launch %.3f ms, replan (seeding + planting) %.3f ms, filter %.3f ms" % (rp.plan_ms, rp.replan_ms, rp.pf_ms))
if args.md:
    open(args.md, "w").write(out)
