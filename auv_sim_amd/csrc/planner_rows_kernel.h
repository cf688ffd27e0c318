// planner_rows_kernel.h -- Planner_RRT (gym_rrt/envs/rrt_dubins.py:162-289,374-423) for throughput batches: FOUR episodes
// per wavefront, one 16-lane DPP row each, rows fed from a work counter.
//
// prrt_kernel gives a wavefront to one episode; at freq <= 15 (the environment's planner runs freq = 10, rrt_env.py:28) a
// steer has at most 15 sub-arcs and everything between the bucket draw and the goal arc is one lane's worth of
// bookkeeping.  Here, as in rrt_rows_kernel, what was wave-uniform is row-uniform: every vector instruction serves four
// episodes, ballots are cut into 16-bit row masks, cross-lane reads stay inside a row.
//   bucket / node choice   random.choice = _randbelow (:186,:223): lanes 0..7 of a row temper eight tries at once; the
//                          picked node is reached along the bucket's member list (newest first)
//   steer (:251-289)       lane s = sub-arc s (two draws each, no data-dependent offsets); theta with its angle_wrap and
//                          x / y / t are left-to-right DPP row-shift chains (lane s is final after s + 1 steps): no LDS
//                          scratch at all -- the only LDS an episode owns is its MT19937 state
//   collision (:435-458)   the obstacles cut into 16 slots of 16 along a space-filling curve (WorldDev::os_*): lane rl
//                          keeps slot rl's bounding box, hit slots get the per-obstacle cull against the tight box of the
//                          path points, survivors the exact d2 <= T_i test (lane = path point, lane 15 = the parent's end)
//   goal arc (:374-423)    16 samples per row and pass, same cull, early exit at the first pass that is not free
// Rows are persistent: a row whose episode is finished (goal connected, step budget used, error) stores it and takes
// the next episode id from a device counter, so rows do not wait for their neighbours' episodes and a batch that is
// not a multiple of the resident capacity has no second round.
//
// Bit-identical to prrt_kernel (tree, bucket lists, counters, generator position).  Limits (the host falls back to
// prrt_kernel beyond them): freq <= 15, <= 256 obstacles, no step log.
#ifndef AUVP_PLANNER_ROWS_KERNEL_H
#define AUVP_PLANNER_ROWS_KERNEL_H
#include "planner_rrt_kernel.h"
#include "rrt_rows_kernel.h"

#include "auvp_math_late.h"

namespace auvp {

constexpr int PRW_WAVES = 4;       // waves per workgroup (16 episodes in flight per workgroup)
constexpr int PRW_MAX_FREQ = 15;   // sub-arcs of a steer: lanes 0..14 of the row (lane 15: the parent's end / entry angle)
constexpr int PRW_LDS_PER_EP = 624 * 4;
constexpr int PRW_NO_ARC = -0x7fffffff;

__device__ __forceinline__ uint32_t rows_word(const RowRng& r, uint32_t j) {
  uint32_t k = r.pslot + j;  // pslot < 624, j < avail <= 624: one wrap
  k = k >= 624u ? k - 624u : k;
  return mt_temper(r.s[k]);
}

// LDS per episode: the generator state and, when it fits 16 bits per entry, a copy of the episode's list of occupied buckets
// (random.choice(occupied) starts every step: with the list in LDS the step's chain of dependent global reads is one shorter)
__host__ __device__ inline int prrt_rows_occ_bytes(int n_buckets, int max_step) {
  const int b = ((max_step + 1) * 2 + 15) & ~15;
  return (n_buckets <= 65535 && b <= 832) ? b : 0;
}

// the obstacle slot tables (auvp_types.h: os_x, os_y, os_t [256], os_r [256] float, os_box [16][4]) as an LDS tile behind the
// episodes' blocks, where three workgroups per CU still fit (config 5: 3 x 54 272 B of the CU's 163 840): obstacle_hit runs for the
// collision test and for every 16-sample pass of the goal arc, and each slot it looks into was four dependent L2 reads
constexpr int PRW_OBST_TILE = RW_MAX_OBST * (8 + 8 + 4) + 16 * 32;  // x, y, r, slot boxes.  (t, needed with the exact tests only, stays in
                                                                    // memory: with it three workgroups no longer fit a CU's LDS granules)
template <bool OBST_LDS>
__global__ __launch_bounds__(PRW_WAVES * 64, 3) void prrt_rows_kernel(WorldDev W, PrrtParamsDev P, PrrtBuffers B, int n_episodes,
                                                                      int* __restrict__ work_counter, int work_base, int occ_bytes) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int wave = (int)(threadIdx.x >> 6);
  const int lane = lane_id();
  const int row = lane >> 4, rl = lane & 15, rowbase = lane & 48;
  // obstacle tile (OBST_LDS): x | y | box | r
  double* tl_x = reinterpret_cast<double*>(smem + (size_t)(PRW_WAVES * RW_ROWS) * (PRW_LDS_PER_EP + occ_bytes));
  double* tl_y = tl_x + RW_MAX_OBST;
  double* tl_box = tl_y + RW_MAX_OBST;
  float* tl_r = reinterpret_cast<float*>(tl_box + 16 * 4);
  if (OBST_LDS) {
    for (int i = threadIdx.x; i < RW_MAX_OBST; i += blockDim.x) { tl_x[i] = W.os_x[i]; tl_y[i] = W.os_y[i]; tl_r[i] = W.os_r[i]; }
    for (int i = threadIdx.x; i < 16 * 4; i += blockDim.x) tl_box[i] = W.os_box[i];
    __syncthreads();
  }
  unsigned char* ebase = smem + (size_t)(wave * RW_ROWS + row) * (PRW_LDS_PER_EP + occ_bytes);
  uint32_t* mt = reinterpret_cast<uint32_t*>(ebase);
  uint16_t* occl = reinterpret_cast<uint16_t*>(ebase + PRW_LDS_PER_EP);  // [max_step + 1] when occ_bytes != 0
  const int capn = B.cap_nodes;
  const size_t capp = (size_t)B.cap_points;
  const int nfreq = (int)P.freq;

  RowRng rng;
  rng.s = mt; rng.pslot = 0u; rng.avail = 0u; rng.drawn = 0ull;
  // ---- the row's episode (row-uniform) ----
  int ep = -1;
  bool live = false, more = true;
  int n_nodes = 0, n_points = 0, n_occ = 0, step = 0, done = 0, status = 0, last_accepted = 0, last_new = -1;
  // prev_n_arc: the goal arc's sample count of the episode's previous step; PRW_NO_ARC = none evaluated yet.  In step mode a
  // row is done with its episode after ONE trip (`stepped`): no end-of-step counter, and the caller's bucket is read where it
  // is used -- three row-uniform registers fewer across the step than rounds 3's step_end / step_bucket / have_prev_arc
  int prev_n_arc = PRW_NO_ARC;
  bool stepped = false;

  for (;;) {
    // ---------------------------------------------------------------- rows without an episode take the next one
    if (wave_any(!live && more)) {
      const bool need = !live && more;
      int e = 0;
      if (need && rl == 0) e = atomicAdd(work_counter, 1) - work_base;
      e = row_read(e, rowbase);
      bool skip = false;
      if (need) {
        if ((unsigned)e >= (unsigned)n_episodes) { more = false; }  // (a negative id -- a base ahead of the counter -- ends the row too)
        else {
          ep = e;
          // step mode: an episode whose bucket is < 0 is not touched at all
          skip = P.step_mode && B.step_bucket[e] < 0;
        }
      }
      const bool load = need && more && !skip;
      if (wave_any(load)) {
        wave_sync();
        if (load) {
          const uint32_t* src = B.mt + (size_t)ep * 624;
          for (int i = rl; i < 624; i += 16) mt[i] = src[i];
          const int4 rs = *reinterpret_cast<const int4*>(B.rng_state + 4 * (size_t)ep);
          rng.pslot = (uint32_t)rs.x; rng.avail = (uint32_t)rs.y;
          rng.drawn = ((unsigned long long)(uint32_t)rs.z) | ((unsigned long long)(uint32_t)rs.w << 32);
          const PrrtSummary& sm = B.summary[ep];
          n_nodes = sm.n_nodes; n_points = sm.n_points; n_occ = sm.n_occ; step = sm.steps; done = sm.done; status = sm.status;
          if (occ_bytes) {
            const int32_t* og = B.occupied + (size_t)ep * capn;
            for (int i = rl; i < n_occ; i += 16) occl[i] = (uint16_t)og[i];
          }
          last_accepted = 0; last_new = -1; prev_n_arc = PRW_NO_ARC; stepped = false;
          // whole 16-word blocks are regenerated in place (rrt_rows_kernel.h): the frontier must sit on a block boundary,
          // which every state this kernel or the host's seeding writes does
          if (((rng.pslot + rng.avail) & 15u) != 0u) status = AUVP_ST_GENERATOR;
          live = true;
        }
        wave_sync();
      }
    }
    if (!wave_any(live)) {
      if (wave_any(more)) continue;  // (every row drew an episode the step skips: draw again)
      break;
    }
    // per-episode array bases are formed where they are used, from the episode id alone (an opaque copy keeps the compiler
    // from hoisting seven 64-bit pointers into registers for the whole step)
    auto epq = [&]() { int v = ep < 0 ? 0 : ep; asm volatile("" : "+v"(v)); return (size_t)v; };
#define en (epq() * (size_t)capn)
#define eb (epq() * (size_t)P.n_buckets)
#define nodes (B.nodes + en)
#define buckets (B.buckets + eb)

    // a row takes part in this step if its episode is still running
    bool act = live && status == 0 && !done && (P.step_mode ? !stepped : step < P.max_step);

    // per-row _randbelow(n): getrandbits(bit_length(n)) until < n, one 32-bit output per try; lanes 0..7 try eight at once
    auto randbelow = [&](bool on, uint32_t n) -> uint32_t {
      uint32_t res = 0u;
      bool search = on;
      const int k = 32 - __clz((int)(n | 1u));
      for (;;) {
        rows_ensure(rng, search, 8u, rl);
        uint32_t v = 0xffffffffu;
        if (search && rl < 8) v = rows_word(rng, (uint32_t)rl) >> (32 - k);
        const uint32_t okm = row_ballot(search && rl < 8 && v < n, rowbase);
        const int f = okm ? (__ffs((int)okm) - 1) : 0;
        const uint32_t got = (uint32_t)row_read((int)v, rowbase + f);
        const bool fin = search && okm != 0u;
        if (fin) res = got;
        rows_advance(rng, search, fin ? (uint32_t)(f + 1) : 8u);
        search = search && !fin;
        if (!wave_any(search)) break;
      }
      return res;
    };

    if (wave_any(act)) {
      // ---------------------------------------------------------------- bucket + node choice (:186, :214-223)
      int b = 0;
      if (P.step_mode) {
        if (act) b = B.step_bucket[ep];
        if (act && b >= P.n_buckets) { status = -1; act = false; }
      } else {
        if (act && n_occ == 0) { status = -1; act = false; }
        const uint32_t oi = randbelow(act, (uint32_t)n_occ);
        if (act) b = occ_bytes ? (int)occl[oi] : B.occupied[en + oi];
      }
      int cnt_b = 0, head_b = 0;
      if (act) { const int2 bw = buckets[b]; cnt_b = prrt_bucket_count(bw, B.bucket_epoch); head_b = bw.y; }
      last_accepted = act ? 0 : last_accepted;
      last_new = act ? -1 : last_new;
      // generate_one_node on an empty bucket: (False, None) (:214-220): the step is used up, nothing else happens
      const bool empty_b = act && cnt_b == 0;
      if (empty_b) { step++; stepped = true; act = false; }
      const uint32_t rsel = randbelow(act, (uint32_t)cnt_b);
      // the rsel-th member of the bucket in creation order is count - 1 - rsel steps from the head of its list
      int par = act ? head_b : 0;
      {
        int hops = act ? cnt_b - 1 - (int)rsel : 0;
        while (wave_any(hops > 0)) {
          if (hops > 0) { par = nodes[par].next; hops--; }
        }
      }
      // ---------------------------------------------------------------- steer (:251-289)
      double cx = 0.0, cy = 0.0, cth = 0.0, ctt = 0.0;
      if (act) {
        const double2 a = *reinterpret_cast<const double2*>(&nodes[par].x);
        const double2 c = *reinterpret_cast<const double2*>(&nodes[par].theta);
        cx = a.x; cy = a.y; cth = c.x; ctt = c.y;
      }
      const double px0 = cx, py0 = cy;
      int n = 0;
      {
        rows_ensure(rng, act, 2u, rl);
        const double u = act ? rows_random_at(rng, 0u) : 0.0;
        rows_advance(rng, act, 2u);
        n = act ? (int)auvp_floor(py_uniform(0.0, P.freq, u) / 1) : 0;
      }
      // sub-arc s draws random() numbers 2s (dist) and 2s + 1 (diff): every lane tempers its own two
      rows_ensure(rng, act, (uint32_t)(4 * n), rl);
      const bool active = act && rl < n;
      double radius = 0.0, phi = 0.0;
      bool taken = false;
      if (active) {
        const double dist = py_uniform(0.0, P.dist_to_end, rows_random_at(rng, (uint32_t)(2 * rl)));
        const double diff = py_uniform(-P.diff_max, P.diff_max, rows_random_at(rng, (uint32_t)(2 * rl + 1)));
        taken = auvp_fabs(dist) > auvp_fabs(diff);
        if (taken) {
          const double s1 = dist + diff, s2 = dist - diff;
          radius = auvp_div_plain(s1 + s2, -s1 + s2);
          phi = auvp_div_plain(s1 + s2, 2 * radius);
        }
      }
      rows_advance(rng, act, (uint32_t)(4 * n));
      const uint32_t tmask = row_ballot(taken, rowbase);
      int nmax = 0;  // longest steer among the rows (wave-uniform loop bound)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int nr = __builtin_amdgcn_readlane(n, 16 * r);
        nmax = nr > nmax ? nr : nmax;
      }
      // theta = angle_wrap(theta + phi) for the taken sub-arcs only, left to right: one DPP row shift per step; lane s holds the
      // angle after sub-arc s once s + 1 steps have run (untaken and idle lanes pass their left neighbour's angle on)
      double th = cth;
      for (int s = 0; s < nmax; s++) {
        const double prev = row_prev_f64(th, cth);
        double a = prev + phi;
        // angle_wrap (:425-433): add -+2 pi until inside [-pi, pi] (at most once for |phi| <= pi)
        for (int guard = 0; guard < 64; guard++) {
          // (the vote is the ballot of ONE compare, |a| > pi, over all lanes: a lane that is not taken holds an angle that is
          // wrapped already, and its `a` is not used)
          if (__builtin_amdgcn_fcmp(auvp_fabs(a), AUVP_PI, 2 /* ordered > */) == 0ull) break;
          const bool hi = a > AUVP_PI, lo = a < -AUVP_PI;
          a = hi ? a + (-2 * AUVP_PI) : (lo ? a + (2 * AUVP_PI) : a);
        }
        th = taken ? a : prev;
      }
      double sn, cs;
      auvp_sincos(rl == 15 ? cth : th, &sn, &cs);  // lane 15: the entry angle
      double dx = 0.0, dy = 0.0, dt = 0.0;
      {
        // sin / cos of the previous TAKEN sub-arc's angle (none: the entry angle on lane 15)
        const uint32_t below = tmask & ((1u << rl) - 1u);
        const int prevl = below ? (31 - __clz((int)below)) : 15;
        const double so = row_read_f64(sn, rowbase + prevl), co = row_read_f64(cs, rowbase + prevl);
        if (taken) {
          dx = radius * (sn - so);
          dy = radius * (-cs + co);
          dt = auvp_sqrt_plain(dx * dx + dy * dy) / 1;
        }
      }
      // x += dx; y += dy; t += dt, left to right (untaken sub-arcs add an exact 0.0): the same kind of chain
      double mx = dx, my = dy, mtt = dt;
      for (int s = 0; s < nmax; s++) {
        mx = row_prev_f64(mx, cx) + dx;
        my = row_prev_f64(my, cy) + dy;
        mtt = row_prev_f64(mtt, ctt) + dt;
      }
      const int napp = __popc(tmask);
      if (act && (n_points + napp > (int)capp || napp + 2 > B.max_pts)) { status = -2; act = false; }
      const bool wr = taken && act;
      if (wr) {
        const int rank = __popc(tmask & ((1u << rl) - 1u));
        // speculative: kept only if the node is accepted.  One 32-byte record per point, a steer's points contiguous
        double2* pr = reinterpret_cast<double2*>(B.points + ((size_t)ep * capp + (size_t)(n_points + rank)) * 4);
        pr[0] = make_double2(mx, my); pr[1] = make_double2(th, mtt);
      }
      // the row's state after the steer = the prefix values of its last sub-arc
      // (sin, cos of the row's final angle -- lane n - 1's, or the entry angle's on lane 15 for a steer without sub-arcs --
      // are what the goal arc needs of an accepted node: auvp_sincos of the same argument)
      double fin_sn, fin_cs;
      {
        const int last = rowbase + (n > 0 ? n - 1 : 0);
        const double ex = row_read_f64(mx, last), ey = row_read_f64(my, last), et = row_read_f64(mtt, last), eth = row_read_f64(th, last);
        const int lsc = rowbase + (n > 0 ? n - 1 : 15);
        fin_sn = row_read_f64(sn, lsc); fin_cs = row_read_f64(cs, lsc);
        if (act && n > 0) { cx = ex; cy = ey; ctt = et; cth = eth; }
      }
      const int cnt = napp;

      // where the new node would go (add_node_to_grid :108-159: int(y / cs), int(x / cs) with Python's negative-index wrap,
      // floor(theta / delta_theta)) and that bucket's size and head: requested now, used by the accept below, so the reads
      // travel while the collision test runs
      bool pre_err = false;
      int pre_bk = -1, pre_c = -1, pre_h = -1;
      if (act) {
        pre_bk = prrt_bucket_of(P, cx, cy, cth, pre_err);
        if (pre_bk >= 0) { const int2 bw = buckets[pre_bk]; pre_c = prrt_bucket_count(bw, B.bucket_epoch); pre_h = bw.y; }
      }

      // ---------------------------------------------------------------- check_collision_free (:435-458)
      // closed rectangle (every path point and the parent's end) + obstacles behind the slot / obstacle culls
      bool bad_pt = false;
      {
        const bool mine = wr || (act && rl == 15);
        const double qx = rl == 15 ? px0 : mx, qy = rl == 15 ? py0 : my;
        const bool wx = (qx >= P.rect[0]) && (qx <= P.rect[2]);
        const bool wy = (qy >= P.rect[1]) && (qy <= P.rect[3]);
        bad_pt = mine && !(wx && wy);
      }
      // lanes = the row's points: `pv` marks a lane that holds one
      auto obstacle_hit = [&](bool on, bool pv, double qx, double qy) -> bool {
        // tight box of the row's points
        const double inf = __builtin_inf();
        double mnx = pv ? qx : inf, mxx = pv ? qx : -inf, mny = pv ? qy : inf, mxy = pv ? qy : -inf;
        mnx = row_min_f64(mnx); mxx = row_max_f64(mxx); mny = row_min_f64(mny); mxy = row_max_f64(mxy);
        const double ts = 0x1p-30 * (auvp_fabs(mnx) + auvp_fabs(mxx) + auvp_fabs(mny) + auvp_fabs(mxy) + 1.0);
        const double tcx = (mnx + mxx) * 0.5, tcy = (mny + mxy) * 0.5;
        const double thx = (mxx - mnx) * 0.5 + ts, thy = (mxy - mny) * 0.5 + ts;
        const double4 sbox = OBST_LDS ? reinterpret_cast<const double4*>(tl_box)[rl]
                                      : reinterpret_cast<const double4*>(W.os_box)[rl];  // lane rl: bounding box of obstacle slot rl
        const bool slot_hit = on && !(sbox.z < mnx - ts || sbox.x > mxx + ts || sbox.w < mny - ts || sbox.y > mxy + ts);
        uint32_t sm = row_ballot(slot_hit, rowbase);
        bool hit = false;
        while (wave_any(sm != 0u)) {
          const bool hs_ = sm != 0u;
          const int j0 = hs_ ? 16 * (__ffs((int)sm) - 1) : 0;
          sm &= sm - 1u;
          const int oi = j0 + rl;
          const double oxj = OBST_LDS ? tl_x[oi] : W.os_x[oi], oyj = OBST_LDS ? tl_y[oi] : W.os_y[oi];
          const double orj = (double)(OBST_LDS ? tl_r[oi] : W.os_r[oi]), otj = W.os_t[oi];
          const bool cand = hs_ && !(auvp_fabs(oxj - tcx) > thx + orj || auvp_fabs(oyj - tcy) > thy + orj);
          uint32_t cm = row_ballot(cand, rowbase);
          while (wave_any(cm != 0u)) {
            const bool has = cm != 0u;
            const int cl = has ? (__ffs((int)cm) - 1) : 0;
            cm &= cm - 1u;
            const double ox = row_read_f64(oxj, rowbase + cl), oy = row_read_f64(oyj, rowbase + cl), ot = row_read_f64(otj, rowbase + cl);
            const double ddx = qx - ox, ddy = qy - oy;
            hit |= has && pv && (ddx * ddx + ddy * ddy <= ot);
            if (row_ballot(hit, rowbase) != 0u) cm = 0u;
          }
          if (row_ballot(hit, rowbase) != 0u) sm = 0u;
        }
        return row_ballot(hit, rowbase) != 0u;
      };
      bool ok = false;
      {
        const bool pv = wr || (act && rl == 15);
        const bool hit = obstacle_hit(act, pv, rl == 15 ? px0 : mx, rl == 15 ? py0 : my);
        ok = act && !hit && row_ballot(bad_pt, rowbase) == 0u;
      }
      // ---------------------------------------------------------------- accept: add_node_to_grid (:108-159)
      int me = -1;
      if (ok && n_nodes >= capn) { status = -2; ok = false; act = false; }
      if (ok) {
        me = n_nodes;
        {
          // (pre_err: the reference's IndexError.  mps_list holds the node by then (:229-230): it is stored -- pre_bk is -1, no
          // bucket -- and the episode ends with AUVP_ERR_ARG below, before the goal connection and without counting the step)
          const int bk = pre_bk, c_before = pre_c, h_before = pre_h;
          if (rl < 4) {
            // the node's 64-byte record: lanes 0..3 of the row store one 16-byte quarter each -- ONE store instruction and one
            // full line per accepted node (rounds 1-3: seven stores into six arrays)
            const int nx = (bk >= 0 && c_before > 0) ? h_before : -1;
            int4 q;
            if (rl == 0) q = make_int4(__double2loint(cx), __double2hiint(cx), __double2loint(cy), __double2hiint(cy));
            else if (rl == 1) q = make_int4(__double2loint(cth), __double2hiint(cth), __double2loint(ctt), __double2hiint(ctt));
            else if (rl == 2) q = make_int4(step, par, n_points, cnt);
            else q = make_int4(bk, nx, 0, 0);
            reinterpret_cast<int4*>(&nodes[me])[rl] = q;
          }
          if (rl == 0 && bk >= 0) {
            buckets[bk] = prrt_bucket_word(c_before + 1, me, B.bucket_epoch);
            if (c_before == 0) {  // first node of the bucket (:157-159)
              B.occupied[en + n_occ] = bk;
              if (occ_bytes) occl[n_occ] = (uint16_t)bk;
            }
          }
          if (c_before == 0) n_occ++;
          n_nodes++;
          n_points += cnt;
          last_accepted = 1; last_new = me;
        }
        if (pre_err) { status = -1; ok = false; act = false; }
      }
      // ---------------------------------------------------------------- connect_to_goal_curve_alt(mps_list[-1]) (:374-423)
      // from the LAST list node even if this step's node was rejected (:237); a step that added no node repeats the previous
      // step's evaluation, which was "not free" (or planning would have ended): its result is reused
      const int lastn = n_nodes - 1;
      double lx = cx, ly = cy, th0 = cth;
      const bool eval = act && (ok || prev_n_arc == PRW_NO_ARC);
      if (eval && !ok) {
        const double2 a = *reinterpret_cast<const double2*>(&nodes[lastn].x);
        lx = a.x; ly = a.y; th0 = nodes[lastn].theta;
      }
      int n_arc = act ? prev_n_arc : -1;
      bool free_ = false;
      if (wave_any(eval)) {
        n_arc = eval ? -1 : n_arc;
        const double2 goal = *reinterpret_cast<const double2*>(B.goal + 2 * (size_t)(ep < 0 ? 0 : ep));
        const double gx = goal.x, gy = goal.y;
        const double theta = auvp_atan2_late(gy - ly, gx - lx);
        double diffg = theta - th0;
        for (int guard = 0; guard < 64; guard++) {  // angle_wrap (all lanes: rows that do not evaluate hold angles in range)
          if (__builtin_amdgcn_fcmp(auvp_fabs(diffg), AUVP_PI, 2 /* ordered > */) == 0ull) break;
          const bool hi = diffg > AUVP_PI, lo = diffg < -AUVP_PI;
          diffg = hi ? diffg + (-2 * AUVP_PI) : (lo ? diffg + (2 * AUVP_PI) : diffg);
        }
        bool go = eval && !(auvp_fabs(diffg) > AUVP_PI / 2);
        const double r_G = auvp_hypot(gx - lx, gy - ly);
        const double dphi = theta - th0;  // phi_G - th0 (same atan2 arguments, :387)
        go = go && (dphi != 0);
        double phi2 = dphi;
        for (int guard = 0; guard < 64; guard++) {
          if (__builtin_amdgcn_fcmp(auvp_fabs(phi2), AUVP_PI, 2 /* ordered > */) == 0ull) break;
          const bool hi = phi2 > AUVP_PI, lo = phi2 < -AUVP_PI;
          phi2 = hi ? phi2 + (-2 * AUVP_PI) : (lo ? phi2 + (2 * AUVP_PI) : phi2);
        }
        phi2 = 2 * phi2;
        const double sn0 = auvp_sin(dphi);
        go = go && (sn0 != 0);
        const double radiusg = r_G / (2 * sn0);
        double length = radiusg * phi2;
        if (phi2 > AUVP_PI) { phi2 -= 2 * AUVP_PI; length = -radiusg * phi2; }
        else if (phi2 < -AUVP_PI) { phi2 += 2 * AUVP_PI; length = -radiusg * phi2; }
        const double ang_vel = phi2 / (length / P.exp_rate);
        // sin / cos of th0: an accepted node's angle went through the steer's sincos already; only an arc from an older
        // node (this step's was rejected and no arc has been evaluated yet) needs its own
        double s0 = fin_sn, c0 = fin_cs;
        if (wave_any(eval && !ok)) {
          double s1, c1;
          auvp_sincos(th0, &s1, &c1);
          if (!ok) { s0 = s1; c0 = c1; }
        }
        const double x_C = lx - radiusg * s0;
        const double y_C = ly + radiusg * c0;
        const double ne = auvp_floor(length / P.exp_rate);
        if (go) n_arc = (ne >= 0 && ne < 1e8) ? (int)ne + 1 : 0;
        // sample the arc 16 points per row and pass; a row stops at its first pass that is not free
        bool sampling = go;
        free_ = go;
        for (int i0 = 0; wave_any(sampling && i0 < n_arc); i0 += 16) {
          const bool on = sampling && i0 < n_arc;
          const int nv = on ? ((n_arc - i0) < 16 ? (n_arc - i0) : 16) : 0;
          const int i = i0 + rl;
          const bool pv = on && rl < nv;
          double sa, ca;
          auvp_sincos(ang_vel * i + th0, &sa, &ca);
          const double ax = x_C + radiusg * sa, ay = y_C - radiusg * ca;
          const bool wx = (ax >= P.rect[0]) && (ax <= P.rect[2]);
          const bool wy = (ay >= P.rect[1]) && (ay <= P.rect[3]);
          const bool outside = row_ballot(pv && !(wx && wy), rowbase) != 0u;
          bool hitp = false;
          if (wave_any(on && !outside)) hitp = obstacle_hit(on && !outside, pv, ax, ay);
          if (on && (outside || hitp)) { free_ = false; sampling = false; }
        }
        if (free_) {
          done = 1;
          // len(path): the final node, the arc, every ancestor's points and the ancestors themselves
          int L = 1 + n_arc;
          int m = lastn;
          bool walking = true;
          while (wave_any(walking)) {
            if (walking) {
              const int4 r = *reinterpret_cast<const int4*>(&nodes[m].step);
              if (r.y < 0) walking = false;
              else { L += r.w + 1; m = r.y; }
            }
          }
          if (rl == 0) {
            PrrtSummary& sum = B.summary[ep];
            sum.path_len = L; sum.last_node = lastn; sum.n_arc = n_arc;
            sum.arc[0] = x_C; sum.arc[1] = y_C; sum.arc[2] = radiusg; sum.arc[3] = ang_vel; sum.arc[4] = th0; sum.arc[5] = length;
          }
        }
      }
      if (act) { prev_n_arc = n_arc; stepped = true; step++; }
    }

    // ---------------------------------------------------------------- rows whose episode is finished store it
    const bool fin = live && (status != 0 || done || (P.step_mode ? stepped : step >= P.max_step));
    if (wave_any(fin)) {
      const unsigned long long drawn = rng.drawn;
      wave_sync();
      if (fin) {
        uint32_t* dst = B.mt + (size_t)ep * 624;
        for (int i = rl; i < 624; i += 16) dst[i] = mt[i];
        if (rl == 0) {
          *reinterpret_cast<int4*>(B.rng_state + 4 * (size_t)ep) =
              make_int4((int)rng.pslot, (int)rng.avail, (int)(uint32_t)(drawn & 0xffffffffull), (int)(uint32_t)(drawn >> 32));
        }
      }
      wave_sync();
      // peek the next random() without consuming it (parity probe): generated ahead in LDS only, the stored words above are
      // what the next launch continues from
      RowRng peek = rng;
      rows_ensure(peek, fin, 2u, rl);
      const double after = fin ? rows_random_at(peek, 0u) : 0.0;
      if (fin && rl == 0) {
        PrrtSummary& sum = B.summary[ep];
        sum.status = status; sum.n_nodes = n_nodes; sum.n_points = n_points; sum.n_occ = n_occ; sum.steps = step;
        sum.done = done; sum.last_accepted = last_accepted; sum.last_new_node = last_new;
        sum.rng_after = after; sum.n_draw32 = drawn;
        if (!done) sum.path_len = 0;
      }
      if (fin) { live = false; status = 0; done = 0; }
      wave_sync();
    }
  }
}

#undef en
#undef eb
#undef nodes
#undef buckets
}  // namespace auvp
#endif
