#!/usr/bin/env python3
"""Temporary source patch: s_memtime stamps at the phase boundaries of rrt_leaf_kernel (auv_sim_amd/csrc/rrt_explore_kernel.h),
active under -DAUVP_LEAF_PHASES; the per-phase totals of an episode land in summary fields tools/leaf_phase_probe.py reads.

  cp auv_sim_amd/csrc/rrt_explore_kernel.h /tmp/rek.orig.h
  python tools/instrument/leaf_phase_patch.py
  python __graft_entry__.py variant exp/libauvplan_lp.so -DAUVP_LEAF_PHASES=1
  cp /tmp/rek.orig.h auv_sim_amd/csrc/rrt_explore_kernel.h
  gpurun -- 'AUVPLAN_LIBRARY=exp/libauvplan_lp.so python tools/leaf_phase_probe.py'
Buckets: 0 marking sweep, 2 records / prefix / owner table, 3 point loop, 4 node term, 5 parent sums + stores,
6 ranking + re-summation + queue refill (1 stays ~0)."""
import os
p = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "auv_sim_amd", "csrc", "rrt_explore_kernel.h")
s = open(p).read()


def ins(anchor, text):
    global s
    assert s.count(anchor) == 1, (s.count(anchor), anchor)
    s = s.replace(anchor, text + anchor, 1)


def stamp(k):
    return "\n#ifdef AUVP_LEAF_PHASES\n    { const long long t_ = __builtin_amdgcn_s_memtime(); lp_[%d] += t_ - lp_t; lp_t = t_; }\n#endif\n" % k


ins("  // ---------------------------------------------------------------- 0. which nodes matter",
    "#ifdef AUVP_LEAF_PHASES\n  long long lp_[7] = {0, 0, 0, 0, 0, 0, 0};\n  long long lp_t = __builtin_amdgcn_s_memtime();\n#endif\n")
ins("  // ---------------------------------------------------------------- the sweep: marked nodes in creation order, 64 per pass", stamp(0))
ins("    wave_sync();\n    if (qn == 0) break;", stamp(6))
ins("    const int nlive = qn < 64 ? qn : 64;", stamp(1))
ins("    // Point rounds, two points per lane and round, software-pipelined", stamp(2))
ins("    double own = c_S[lane], ntv = 0.0, ctt = 0.0, nlen = 0.0;", stamp(3))
ins("    // ---------------------------------------------------------------- 2. running sums down the tree", stamp(4))
ins("    // ---------------------------------------------------------------- 3. ranking of the pass's qualifying leaves", stamp(5))
ins("  if (lane == 0) {\n    if (B.leaf_stats) {",
    "#ifdef AUVP_LEAF_PHASES\n  if (lane == 0) { sum.rng_after = (double)lp_[0]; sum.best_cost[1] = (double)lp_[1]; sum.best_cost[2] = (double)lp_[2]; "
    "sum.best_cost[3] = (double)lp_[3]; sum.best_length = (double)lp_[4]; }\n#endif\n")
s = s.replace("    sum.leaf_elems = leaf_elems;",
              "#ifdef AUVP_LEAF_PHASES\n    sum.leaf_elems = lp_[5]; sum.n_draw32 = (unsigned long long)lp_[6];\n#else\n    sum.leaf_elems = leaf_elems;\n#endif", 1)
s = s.replace("    if (best_leaf >= 0) {\n      sum.best_cost[0] = best_tot;", "#ifndef AUVP_LEAF_PHASES\n    if (best_leaf >= 0) {\n      sum.best_cost[0] = best_tot;", 1)
s = s.replace("      sum.status = 1;  // no qualifying leaf: opt_path stays None (:174)\n    }",
              "      sum.status = 1;  // no qualifying leaf: opt_path stays None (:174)\n    }\n#endif", 1)
open(p, "w").write(s)
