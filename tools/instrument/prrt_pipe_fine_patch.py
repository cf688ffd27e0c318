import os
p=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))+'/auv_sim_amd/csrc/planner_pipe_kernel.h'
s=open(p).read()
def rep(old,new,cnt=1):
    global s
    assert s.count(old)==cnt,(s.count(old),old[:90])
    s=s.replace(old,new)
rep("  double diag[4];","  double diag[4];\n  unsigned long long fine[32];")
rep("namespace auvp {\n","namespace auvp {\n#define FMARK(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long tn = __builtin_amdgcn_s_memtime(); if (lane_id() == 0) ctl->fine[i] += tn - tf; tf = tn; __builtin_amdgcn_sched_barrier(0); } while (0)\n",1)
rep("      ctl->h_done = 0; ctl->s_done = 0; ctl->g_done = 0;","      ctl->h_done = 0; ctl->s_done = 0; ctl->g_done = 0;\n      for (int i = 0; i < 32; i++) ctl->fine[i] = 0ull;")
rep("""      cur = k;
      sp_set(k, rng.cslot, rng.drawn);""","""      unsigned long long tf = t_h0;
      cur = k;
      sp_set(k, rng.cslot, rng.drawn);""")
rep("""        const uint32_t oi = ppipe_randbelow<true>(rng, (uint32_t)n_occ, rmin, ok1, more);
        fits = ok1;""","""        FMARK(0);
        const uint32_t oi = ppipe_randbelow<true>(rng, (uint32_t)n_occ, rmin, ok1, more);
        fits = ok1;
        FMARK(1);""")
rep("""          const int cnt_b = uni(prrt_bucket_count(bw, epoch_b));
          if (cnt_b == 0) kind = 1;""","""          const int cnt_b = uni(prrt_bucket_count(bw, epoch_b));
          FMARK(2);
          if (cnt_b == 0) kind = 1;""")
rep("""            fits = ok2;
            if (fits) {
              // the rsel-th member""","""            fits = ok2;
            FMARK(3);
            if (fits) {
              // the rsel-th member""")
rep("""              par = pv;
              const double* pr = &nodes[par].x;""","""              par = pv;
              FMARK(4);
              const double* pr = &nodes[par].x;""")
rep("""                n_total = n_total < 0 ? 0 : (n_total > DUO_MAX_FREQ ? DUO_MAX_FREQ : n_total);
                const int n = n_total;""","""                n_total = n_total < 0 ? 0 : (n_total > DUO_MAX_FREQ ? DUO_MAX_FREQ : n_total);
                FMARK(5);
                const int n = n_total;""")
rep("""                  if (lane == 0) { q->px = p0; q->py = p1; q->pth = p2; q->ptt = p3; }
                }""","""                  FMARK(6);
                  if (lane == 0) { q->px = p0; q->py = p1; q->pth = p2; q->ptt = p3; }
                  FMARK(7);
                }""")
rep("""      if (lane == 0) duo_poke64(&q->tagA, duo_tag(epoch, k));
#ifdef AUVP_DUO_DIAG
      diag_h +=""","""      if (lane == 0) duo_poke64(&q->tagA, duo_tag(epoch, k));
      FMARK(8);
#ifdef AUVP_DUO_DIAG
      diag_h +=""")
# S
rep("""      // ---- part A, copied out
      const int a_status = uni(q->status), a_kind = uni(q->kind);
      int n_total = uni(q->n_total);""","""      unsigned long long tf = t_s0;
      // ---- part A, copied out
      const int a_status = uni(q->status), a_kind = uni(q->kind);
      int n_total = uni(q->n_total);""")
rep("""      if (lane == 0) duo_poke64(&q->tagB, 0ull);
      int ok = 0, cnt = 0, have_sc = 0;""","""      if (lane == 0) duo_poke64(&q->tagB, 0ull);
      FMARK(9);
      int ok = 0, cnt = 0, have_sc = 0;""")
rep("""          (void)n;
          double sn, cs;
          auvp_sincos(myth, &sn, &cs);""","""          (void)n;
          FMARK(10);
          double sn, cs;
          auvp_sincos(myth, &sn, &cs);
          FMARK(11);""")
rep("""          double mx = 0.0, my = 0.0, mt_ = 0.0;
          for (unsigned long long tm = tmask; tm; tm &= tm - 1ull) {""","""          FMARK(12);
          double mx = 0.0, my = 0.0, mt_ = 0.0;
          for (unsigned long long tm = tmask; tm; tm &= tm - 1ull) {""")
rep("""          cth = th;
          cnt = __popcll(tmask);""","""          FMARK(13);
          cth = th;
          cnt = __popcll(tmask);""")
rep("""        wave_sync();
        int P_n = cnt + 1;""","""        wave_sync();
        FMARK(14);
        int P_n = cnt + 1;""")
rep("""      if (lane == 0) {
        q->ok = ok; q->cnt = cnt; q->have_sc = have_sc;""","""      FMARK(15);
      if (lane == 0) {
        q->ok = ok; q->cnt = cnt; q->have_sc = have_sc;""")
rep("""      if (lane == 0) duo_poke64(&q->tagB, duo_tag(epoch, k));
#ifdef AUVP_DUO_DIAG""","""      if (lane == 0) duo_poke64(&q->tagB, duo_tag(epoch, k));
      FMARK(16);
#ifdef AUVP_DUO_DIAG""")
rep("""    if (!done) { sum.arc[0] = (double)diag_m;""","""    if (!done) { for (int i = 0; i < 32; i++) ptF[i] = (double)ctl->fine[i]; }
    if (!done) { sum.arc[0] = (double)diag_m;""")
open(p,'w').write(s)
