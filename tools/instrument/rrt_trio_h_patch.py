# temporary section clocks of rrt_trio_kernel's H wavefront (diag build only): totals land in the episode's first node rows
import os
p=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))+'/auv_sim_amd/csrc/rrt_trio_kernel.h'
s=open(p).read()
def rep(old,new,cnt=1):
    global s
    assert s.count(old)==cnt,(s.count(old),old[:80])
    s=s.replace(old,new)
rep("  unsigned long long sp_drawn[TRIO_RING];\n  uint32_t sp_cslot[TRIO_RING];\n};","  unsigned long long sp_drawn[TRIO_RING];\n  uint32_t sp_cslot[TRIO_RING];\n  unsigned long long fine[16];\n};")
rep("namespace auvp {\n","namespace auvp {\n#define FMARK(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long tn = __builtin_amdgcn_s_memtime(); if (lane_id() == 0) ctl->fine[i] += tn - tf; tf = tn; __builtin_amdgcn_sched_barrier(0); } while (0)\n",1)
rep("      sp_set(k, rng.cslot, rng.drawn);\n      // (the snapshot of `ver`","      unsigned long long tf = t_b0;\n      sp_set(k, rng.cslot, rng.drawn);\n      // (the snapshot of `ver`")
rep("""      for (;;) {
        if (!ensure(128u)) {""","""      FMARK(0);
      for (;;) {
        if (!ensure(128u)) {""")
rep("""        u_me = ring_random_at(rng, (uint32_t)lane);
        const int rbj""","""        FMARK(1);
        u_me = ring_random_at(rng, (uint32_t)lane);
        const int rbj""")
rep("""      if (again) continue;
      TrioPacket* q = packet(k);""","""      if (again) continue;
      FMARK(2);
      TrioPacket* q = packet(k);""")
rep("""        const int n = n_total, nwin = 3 * n;
        u_win[lane] = u_me;""","""        const int n = n_total, nwin = 3 * n;
        FMARK(3);
        u_win[lane] = u_me;""")
rep("""        wave_sync();
        const double* uw = u_win + base;""","""        wave_sync();
        FMARK(4);
        const double* uw = u_win + base;""")
rep("""        double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0, p4 = 0.0;
        if (NW == 3) {""","""        FMARK(5);
        double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0, p4 = 0.0;
        if (NW == 3) {""")
rep("""        const int mypos = 2 * lane + cbelow;""","""        FMARK(6);
        const int mypos = 2 * lane + cbelow;""")
rep("""        ring_advance(rng, (uint32_t)(2 * (base + used)));
      }""","""        ring_advance(rng, (uint32_t)(2 * (base + used)));
        FMARK(7);
      }""")
rep("""      if (lane == 0) duo_poke64(NW == 3 ? &q->tag : &q->tag_s, duo_tag(epoch, k));
#ifdef AUVP_DUO_DIAG""","""      if (lane == 0) duo_poke64(NW == 3 ? &q->tag : &q->tag_s, duo_tag(epoch, k));
      FMARK(8);
#ifdef AUVP_DUO_DIAG""")
rep("""      if (lane == 0) ctl->hist_bin[0] = (int)(diag_h >> 8);""","""      if (lane == 0) ctl->hist_bin[0] = (int)(diag_h >> 8);
      if (lane < 10) nodeF[(lane / 5) * 8 + (lane % 5)] = (double)ctl->fine[lane];""")
rep("      ctl->l_done = NW == 4 ? 0 : 1;","      ctl->l_done = NW == 4 ? 0 : 1;\n      for (int i = 0; i < 16; i++) ctl->fine[i] = 0ull;")
open(p,'w').write(s)
