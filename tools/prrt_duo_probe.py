#!/usr/bin/env python3
"""Config 4 (512 Planner_RRT.planning(2000) episodes) with one / two / three wavefronts per episode.  With a diagnostic build
(AUVPLAN_LIBRARY=<lib built with -DAUVP_DUO_DIAG>) also the shader clocks per step each wavefront spends at work."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auv_sim_amd import _lib, synth  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import env_options  # noqa: E402  (tests/env_options.py: AUVP_<NAME> in os.environ steers live contexts -- this process only)
env_options.install()
from auv_sim_amd._prrt_lib import PlannerBatch  # noqa: E402

ctx = _lib.Context(0)
w = synth.make_rect_world(seed=3, n_obstacles=256)
ctx.set_world(obstacles=w["obstacles"])
for n in (1, 64, 512, 1024):
    starts = np.tile(np.array([w["start"][0], w["start"][1], 0.0, 0.0]), (n, 1))
    goals = np.tile(w["goal"], (n, 1))
    seeds = np.arange(n, dtype=np.uint64)
    out = []
    for duo, trio, pipe in (("0", "0", "0"), ("1", "0", "0"), ("1", "1", "0"), ("1", "1", "1")):
        os.environ["AUVP_PRRT_DUO"], os.environ["AUVP_PRRT_TRIO"], os.environ["AUVP_PRRT_PIPE"] = duo, trio, pipe
        ms = []
        for i in range(3):
            pb = PlannerBatch(ctx, starts, goals, w["rect"], 2000, seeds=seeds, freq=10, cell=5, subs=1)
            s = pb.plan()
            if i:
                ms.append(ctx.last_kernel_ms())
        out.append("%s %.2f ms (%.2f us/step)" % (ctx.prrt_last_kernel(), np.mean(ms), 1e3 * np.mean(ms) / max(s["steps"].max(), 1)))
        if os.environ.get("AUVPLAN_LIBRARY") and pipe == "1":
            nd = s["done"] == 0
            st = float(s["steps"][nd].sum())
            a = s["arc"][nd]
            print("   diag (%s): clocks per step at work (AUVP_DUO_DIAG=2: waiting): M %.0f, H %.0f, S %.0f, G %.0f, D %.0f; redos per step %.3f; nodes per step %.2f"
                  % (ctx.prrt_last_kernel(), a[:, 0].sum() / st, a[:, 1].sum() / st, a[:, 2].sum() / st, a[:, 3].sum() / st, a[:, 5].sum() / st, a[:, 4].sum() / st,
                     s["n_nodes"][nd].sum() / st))
        elif os.environ.get("AUVPLAN_LIBRARY") and duo == "1":
            nd = s["done"] == 0
            st = float(s["steps"][nd].sum())
            a = s["arc"][nd]
            arc = a[:, 3].sum() / st if trio == "1" else 0.0  # (two wavefronts: main runs the arcs itself)
            print("   diag (%s): per step main works %.0f clocks and waits %.0f for arc verdicts, helper builds for %.0f, arc wavefront %.0f; redos per step %.3f"
                  % (ctx.prrt_last_kernel(), a[:, 0].sum() / st, a[:, 1].sum() / st, a[:, 2].sum() / st, arc, a[:, 4].sum() / st))
    print("E=%d: " % n + " | ".join(out))
