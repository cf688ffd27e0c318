import os
p=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))+'/auv_sim_amd/csrc/astar_kernel.h'
s=open(p).read()
def rep(old,new,cnt=1):
    global s
    assert s.count(old)==cnt,(s.count(old),old[:80])
    s=s.replace(old,new)
rep("""  while (n_open > 0) {
    // ------------------------------------------------------------ pop the first minimum f""","""  unsigned long long ph[12] = {0,0,0,0,0,0,0,0,0,0,0,0};
  unsigned long long tf = __builtin_amdgcn_s_memtime();
#define PH(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long tn = __builtin_amdgcn_s_memtime(); ph[i] += tn - tf; tf = tn; __builtin_amdgcn_sched_barrier(0); } while (0)
  while (n_open > 0) {
    PH(7);
    // ------------------------------------------------------------ pop the first minimum f""")
rep("""    const int cur = uni(bi);
    if (list_ok) {  // the last entry""","""    PH(8);
    const int cur = uni(bi);
    if (list_ok) {  // the last entry""")
rep("""    n_exp++;
    if (PAIR) {
      if (lane == 0) { box->cx = cxp; box->cy = cyp; box->clen = clen; }""","""    n_exp++;
    PH(0);
    if (PAIR) {
      if (lane == 0) { box->cx = cxp; box->cy = cyp; box->clen = clen; }""")
rep("""    const int nch = __popc(childmask);
    if (n_nodes + nch > cap) { status = -2; break; }""","""    PH(1);
    const int nch = __popc(childmask);
    if (n_nodes + nch > cap) { status = -2; break; }""")
rep("""          if (status) break;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        const int kk = lane & 7;""","""          if (status) break;
        }
        PH(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        const int kk = lane & 7;""")
rep("""      n_open += opened;
      visited_count += opened;
      wave_sync();
    };""","""      n_open += opened;
      visited_count += opened;
      wave_sync();
      PH(5);
    };""")
rep("""  if (PAIR && lane == 0) lds_poke(&box->stop, 1);""","""  if (logx && lane == 0) { double* e = B.exp_log + (size_t)ep * P.cap_exp * 8; for (int i = 0; i < 12; i++) e[i] = (double)ph[i]; }
  if (PAIR && lane == 0) lds_poke(&box->stop, 1);""")
open(p,'w').write(s)
