import os
p=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))+'/auv_sim_amd/csrc/astar_kernel.h'
s=open(p).read()
def rep(old,new,cnt=1):
    global s
    assert s.count(old)==cnt,(s.count(old),old[:80])
    s=s.replace(old,new)
rep("""  if (PAIR && second) {""","""  if (PAIR && second) {
    unsigned long long ph[12] = {0,0,0,0,0,0,0,0,0,0,0,0};
    unsigned long long tf = __builtin_amdgcn_s_memtime();
#define PHX(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long tn = __builtin_amdgcn_s_memtime(); ph[i] += tn - tf; tf = tn; __builtin_amdgcn_sched_barrier(0); } while (0)""")
rep("""        if (stop) break;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");  // (it reads nothing the searching wavefront writes outside LDS)""","""        if (stop) break;
      }
      PHX(0);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");  // (it reads nothing the searching wavefront writes outside LDS)""")
rep("""      const int ntop = (int)dist_left;
      int key = -1, tb = -1;""","""      const int ntop = (int)dist_left;
      PHX(1);
      int key = -1, tb = -1;""")
rep("""        if (!okx) gc = astar_lower_bound(gx1, W.g_ncol, qx, gc);""","""        PHX(2);
        if (!okx) gc = astar_lower_bound(gx1, W.g_ncol, qx, gc);""")
rep("""        const unsigned long long bm = __ballot(m);
        const unsigned g8 = (unsigned)((bm >> (k8 * 8)) & 0xffull);
        const unsigned cm3 = g8 & 7u, rm3 = (g8 >> 3) & 7u;
        if (cm3 && rm3) key = (gr - 1 + (__ffs((int)rm3) - 1)) * W.g_ncol + (gc - 1 + (__ffs((int)cm3) - 1));
        // the first time bin""","""        PHX(3);
        const unsigned long long bm = __ballot(m);
        const unsigned g8 = (unsigned)((bm >> (k8 * 8)) & 0xffull);
        const unsigned cm3 = g8 & 7u, rm3 = (g8 >> 3) & 7u;
        if (cm3 && rm3) key = (gr - 1 + (__ffs((int)rm3) - 1)) * W.g_ncol + (gc - 1 + (__ffs((int)cm3) - 1));
        // the first time bin""")
rep("""      // ---- round trip 3: the table values
      double pr = 0.0, tn = 0.0;""","""      PHX(4);
      // ---- round trip 3: the table values
      double pr = 0.0, tn = 0.0;""")
rep("""      if (s8 == 0) {
        box->len_[k8] = len_; box->pr[k8] = pr; box->tn[k8] = tn;""","""      __builtin_amdgcn_s_waitcnt(0);
      PHX(5);
      if (s8 == 0) {
        box->len_[k8] = len_; box->pr[k8] = pr; box->tn[k8] = tn;""")
rep("""      if (lane == 0) lds_poke(&box->seq_x, e + 1);
    }
    return;""","""      if (lane == 0) lds_poke(&box->seq_x, e + 1);
      PHX(6);
    }
    if (logx && lane == 0) { double* ee = B.exp_log + (size_t)ep * P.cap_exp * 8 + 16; for (int i = 0; i < 8; i++) ee[i] = (double)ph[i]; }
    return;""")
open(p,'w').write(s)
