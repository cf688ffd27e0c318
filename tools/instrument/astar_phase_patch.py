# temporary phase clocks for astar_kernel (variant 3 path); totals of an instance land in rows 0-1 of its expansion log
import sys
import os
p=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))+'/auv_sim_amd/csrc/astar_kernel.h'
s=open(p).read()
def rep(old,new,cnt=1):
    global s
    assert s.count(old)==cnt,(s.count(old),old[:80])
    s=s.replace(old,new)
rep("  while (n_open > 0) {\n    // ------------------------------------------------------------ pop the first minimum f",
"""  unsigned long long ph[12] = {0,0,0,0,0,0,0,0,0,0,0,0};
  unsigned long long tf = __builtin_amdgcn_s_memtime();
#define PH(i) do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); const unsigned long long tn = __builtin_amdgcn_s_memtime(); ph[i] += tn - tf; tf = tn; __builtin_amdgcn_sched_barrier(0); } while (0)
  while (n_open > 0) {
    PH(11);
    // ------------------------------------------------------------ pop the first minimum f""")
rep("""    // reduce: smaller f wins; on equal f the smaller index wins (list order).""","""    PH(0);
    // reduce: smaller f wins; on equal f the smaller index wins (list order).""")
rep("""    const int cur = uni(bi);
    if (list_ok) {  // the last entry""","""    PH(1);
    const int cur = uni(bi);
    if (list_ok) {  // the last entry""")
rep("""    bool stop;
    if (V == 0) stop = (cxp == gx && cyp == gy);""","""    PH(2);
    bool stop;
    if (V == 0) stop = (cxp == gx && cyp == gy);""")
rep("""    bool hit = false;
    if (inb) {
      auto circles""","""    PH(3);
    bool hit = false;
    if (inb) {
      auto circles""")
rep("""    const int nch = __popc(childmask);
    if (n_nodes + nch > cap) { status = -2; break; }""","""    PH(4);
    const int nch = __popc(childmask);
    if (n_nodes + nch > cap) { status = -2; break; }""")
rep("""      // children of the SOG variant, lane k < 8 = neighbour k: nothing one child does is seen by another (eight""","""      PH(5);
      // children of the SOG variant, lane k < 8 = neighbour k: nothing one child does is seen by another (eight""")
rep("""      double pr = 0.0, tn = 0.0;
      int was = 0;""","""      PH(6);
      double pr = 0.0, tn = 0.0;
      int was = 0;""")
rep("""      const double g_ = ccost - w4 * pr;
      const double h_ = -w2 * dist_left - w3 * (double)H - w4 * tn;""","""      PH(7);
      const double g_ = ccost - w4 * pr;
      const double h_ = -w2 * dist_left - w3 * (double)H - w4 * tn;""")
rep("""      n_open += opened;
      visited_count += opened;
      wave_sync();
    }
    int slot = 0;""","""      n_open += opened;
      visited_count += opened;
      wave_sync();
      PH(8);
    }
    int slot = 0;""")
rep("""  for (int i = lane; i < n_hopen; i += 64) B.hab_left""","""  if (logx && lane == 0) { double* e = B.exp_log + (size_t)ep * P.cap_exp * 8; for (int i = 0; i < 12; i++) e[i] = (double)ph[i]; }
  for (int i = lane; i < n_hopen; i += 64) B.hab_left""")
open(p,'w').write(s)
