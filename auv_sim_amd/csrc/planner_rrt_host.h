// planner_rrt_host.h -- C-ABI entry points for Planner_RRT (auvp_prrt_*), host side.
// Included at the end of auvplan.hip (same translation unit: uses auvp_handle, fail, HIPCHK, upload).
#ifndef AUVP_PLANNER_RRT_HOST_H
#define AUVP_PLANNER_RRT_HOST_H

static_assert(sizeof(auvp_prrt_summary) == sizeof(auvp::PrrtSummary), "prrt summary layout");

namespace {

struct PrrtState {
  bool ready = false;
  int E = 0;
  auvp::PrrtParamsDev P{};
  auvp::PrrtBuffers B{};
  DevBuf nodes, node_bucket, points, occupied, buckets, mt, rng_state, start, goal, step_bucket, summary, st_log,
      tmp_off, tmp_out, env_done, thetas, seeds;
  // the bucket table's current epoch (PrrtBuffers::bucket_epoch): a new batch on the same allocation takes the next one; the
  // table is cleared when the allocation changes or the 8-bit tag would wrap
  int bucket_epoch = 0;
  void* bucket_alloc = nullptr;
  size_t bucket_alloc_bytes = 0;
  // work counter of the four-episodes-per-wavefront kernel (planner_rows_kernel.h).  It is never reset between launches: a
  // launch hands out ids counter - work_base, and every one of its rows ends with exactly one pull that finds no episode, so
  // the next launch's base is known on the host (no fill launch per plan / step)
  DevBuf snap_sum, snap_rng, snap_mt, redo_mask, redo_count;  // the pipeline fallback's snapshot of a launch's episodes
  DevBuf work;
  int work_base = 0;
  bool work_memset = false;  // set by the first launch under stream capture: the counter is zeroed by a memset node per launch
  bool use_rows = false;  // decided once per batch: the two kernels keep the generator's lazy state in different block phases
  const char* last_kernel = "";
  DevBuf env_err;    // device int32[2]: {status, environment} of the first episode that failed inside the device-resident loop
  // captured step graphs of the device-resident loop (auvp_graph_*).  A graph holds raw device pointers (this batch's
  // buffers and the caller's observation / reward arrays): it dies with the batch it was captured for -- prrt_configure
  // destroys them, and the ids stay taken so that a stale id can never reach another batch's graph.
  std::vector<hipGraphExec_t> graphs;
  void drop_graphs() { for (auto& g : graphs) if (g) { (void)hipGraphExecDestroy(g); g = nullptr; } }
  ~PrrtState() { drop_graphs(); }
  bool thetas_ready = false;
};

PrrtState* prrt_of(auvp_handle* h);

}  // namespace
extern "C" hipError_t auvpi_prrt_rows_launch(int obst_lds, const auvp::WorldDev* W, const auvp::PrrtParamsDev* P, const auvp::PrrtBuffers* B,
                                             int n_episodes, int* work_counter, int work_base, int occ_bytes, int grid, int block, int lds,
                                             hipStream_t stream);
namespace {

// `sync`: wait for the launch and record its HIP-event time (the device-resident loop passes false: it only enqueues)
int prrt_launch(auvp_handle* h, PrrtState& S, int step_mode, bool sync = true, bool one_wave_only = false) {
  S.P.step_mode = step_mode;
  if (!one_wave_only) { h->pipe_clear(); h->pipe_fallback_last = 0; }
  const int nfreq = (int)std::floor(S.P.freq);
  // latency run (at most three waves per SIMD on this GPU) or throughput run: register budget and steer differ (planner_rrt_kernel.h)
  int n_cu_l = 256;
  (void)hipDeviceGetAttribute(&n_cu_l, hipDeviceAttributeMultiprocessorCount, h->device);
  if (n_cu_l <= 0) n_cu_l = 256;
  const bool lat = h->opt_flag(OPT_PRRT_LAT, S.E <= 12 * n_cu_l);  // (12: see prrt_create_batch's choice of the four-episode kernel)
  // latency runs: workgroups small enough that every CU gets one (512 episodes: 256 workgroups of two waves)
  int wg_waves = auvp::RRT_WAVES;
  if (lat) { wg_waves = (S.E + n_cu_l - 1) / n_cu_l; wg_waves = wg_waves < 1 ? 1 : (wg_waves > auvp::RRT_WAVES ? auvp::RRT_WAVES : wg_waves); }
  const int grid = (S.E + wg_waves - 1) / wg_waves;
  const size_t lds = (size_t)wg_waves * auvp::prrt_lds_per_wave(S.B.max_pts, nfreq, lat);
  if (lds > 160 * 1024) return fail(h, AUVP_ERR_ARG, "LDS need %zu B > 160 KiB", lds);
  const int O = h->W.n_obstacles;
  auto launch = [&](auto kern) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(wg_waves * 64), lds, h->stream, h->W, S.P, S.B, S.E);
    return hipGetLastError();
  };
  if (sync) HIPCHK(h, hipEventRecord(h->ev0, h->stream));  // (enqueue-only calls may be under stream capture: no events)
  hipError_t le;
  int grid_used = grid, block_used = wg_waves * 64;
  size_t lds_used = lds;
  S.last_kernel = "prrt_kernel";
  // plan-mode latency runs of at most four episodes per CU: a pipeline of four wavefronts per episode (planner_pipe_kernel.h).
  // Option PRRT_PIPE = 0 / 1 forces the choice where the kernel's limits allow it.  Only where this call waits for the launch:
  // the pipeline is speculative and an episode it gives up on is redone on prrt_kernel below (pipeline fallback).
  const bool use_pipe = sync && !one_wave_only && !S.use_rows && step_mode == 0 && lat && !(S.P.flags & AUVP_FLAG_ITER_LOG) &&
                        nfreq <= auvp::DUO_MAX_FREQ && O <= 256 && S.B.max_pts <= auvp::DUO_CS + 2 && h->opt_flag(OPT_PRRT_PIPE, S.E <= 4 * n_cu_l);
  if (S.use_rows) {
    // persistent rows (four episodes per wavefront) fed from a device counter: as many workgroups as fit the chip at
    // three per CU (one wave per SIMD each), fewer when the batch is smaller
    S.last_kernel = "prrt_rows_kernel";
    const int per_wg = auvp::PRW_WAVES * auvp::RW_ROWS;
    grid_used = std::min((S.E + per_wg - 1) / per_wg, 3 * n_cu_l);
    block_used = auvp::PRW_WAVES * 64;
    const int occ_bytes = auvp::prrt_rows_occ_bytes(S.P.n_buckets, S.P.max_step);
    lds_used = (size_t)per_wg * (auvp::PRW_LDS_PER_EP + occ_bytes);
    // the obstacle slot tables as an LDS tile where three workgroups per CU still fit beside it (option PRRT_OBST_LDS overrides)
    // (LDS is handed out in 1 280-byte granules on this GPU: three workgroups of 53 760 B fit a CU, three of 54 272 B do not)
    auto granules = [](size_t b) { return (b + 1279) / 1280 * 1280; };
    const bool obst_lds = h->opt_flag(OPT_PRRT_OBST_LDS, 3 * granules(lds_used + auvp::PRW_OBST_TILE) <= (size_t)160 * 1024);
    if (obst_lds) lds_used += auvp::PRW_OBST_TILE;
    // The ids a launch hands out are counter - work_base.  Eager launches carry the base as an argument (no memset per
    // launch).  A launch recorded into a hipGraph would freeze that argument while the counter keeps advancing on every
    // replay, so from the first capture on this state zeroes the counter with a memset NODE in front of every launch and
    // passes base 0 -- captured and eager launches alike (a replay may run between any two eager launches).
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(h->stream, &cap);
    if (cap != hipStreamCaptureStatusNone) S.work_memset = true;
    if (!S.work.p) {
      if (cap != hipStreamCaptureStatusNone) return fail(h, AUVP_ERR_STATE, "the rows work counter must exist before a stream capture (run one eager step first)");
      le = S.work.reserve(sizeof(int));
      if (le == hipSuccess) le = hipMemsetAsync(S.work.p, 0, sizeof(int), h->stream);
      S.work_base = 0;
    } else if (S.work_memset || S.work_base > (1 << 30)) {
      le = hipMemsetAsync(S.work.p, 0, sizeof(int), h->stream);
      S.work_base = 0;
    } else le = hipSuccess;
    if (le == hipSuccess) {
      // (the kernel lives in prrt_rows_kernels.hip: a translation unit with its own compiler flags)
      le = auvpi_prrt_rows_launch(obst_lds ? 1 : 0, &h->W, &S.P, &S.B, S.E, S.work.as<int>(), S.work_base, occ_bytes, grid_used, block_used,
                                  (int)lds_used, h->stream);
      // every episode once + one empty pull per row; a failed launch pulled nothing, a memset-fronted one restarts at 0
      if (le == hipSuccess && !S.work_memset) S.work_base += S.E + grid_used * per_wg;
    }
  } else if (use_pipe) {
    // four wavefronts per episode, feed-forward (planner_pipe_kernel.h)
    S.last_kernel = "prrt_pipe_kernel";
    int eps_wg = (S.E + n_cu_l - 1) / n_cu_l;
    eps_wg = eps_wg < 1 ? 1 : (eps_wg > auvp::PPIPE_EP ? auvp::PPIPE_EP : eps_wg);
    // round 6: a FIFTH wavefront per episode takes the sub-arc draws off H, the slowest stage (planner_pipe_kernel.h: D) -- where
    // a workgroup of five-wavefront episodes fits the 1 024-thread limit (at most three episodes per workgroup: up to 768
    // episodes on this GPU; config 4's 512 run two per CU).  Option PRRT_PIPE_DRAW = 0 / 1 forces the choice within that limit.
    const bool draw_wave = eps_wg <= auvp::PPIPE_EP5 && h->opt_flag(OPT_PRRT_PIPE_DRAW, true);
    grid_used = (S.E + eps_wg - 1) / eps_wg;
    block_used = eps_wg * (draw_wave ? 320 : 256);
    // the member lists' next links in LDS where they fit beside the slots (option PRRT_NEXT_LDS = 0 keeps them in memory)
    int next_lds = (size_t)eps_wg * auvp::ppipe_per_episode_bytes(S.B.max_pts, S.B.cap_nodes) <= (size_t)150 * 1024 ? 1 : 0;
    next_lds = next_lds && h->opt_flag(OPT_PRRT_NEXT_LDS, true);
    // round 6: ... and the bucket table + the occupied list, where they fit too (config 4: 1 600 buckets = 19 KB per episode;
    // option PRRT_BUCKET_LDS = 0 keeps them in memory)
    const int occ_cap = auvp::ppipe_occ_entries(S.P.n_buckets, S.B.cap_nodes);
    int bk_lds = (size_t)eps_wg * auvp::ppipe_per_episode_bytes(S.B.max_pts, next_lds ? S.B.cap_nodes : 0, S.P.n_buckets, occ_cap) <= (size_t)150 * 1024 ? 1 : 0;
    bk_lds = bk_lds && h->opt_flag(OPT_PRRT_BUCKET_LDS, true);
    lds_used = (size_t)eps_wg * auvp::ppipe_per_episode_bytes(S.B.max_pts, next_lds ? S.B.cap_nodes : 0, bk_lds ? S.P.n_buckets : 0, bk_lds ? occ_cap : 0);
    next_lds |= bk_lds << 1;  // (one kernel argument: bit 0 the links, bit 1 the bucket table)
    // what the fallback needs to take an episode back to where this launch found it: its record, generator and position
    le = S.snap_sum.reserve((size_t)S.E * sizeof(auvp::PrrtSummary));
    if (le == hipSuccess) le = S.snap_rng.reserve((size_t)S.E * 4 * sizeof(int32_t));
    if (le == hipSuccess) le = S.snap_mt.reserve((size_t)S.E * 624 * sizeof(uint32_t));
    if (le == hipSuccess) le = S.redo_mask.reserve((size_t)S.E);
    HIPCHK(h, le);
    hipLaunchKernelGGL(auvp::prrt_snapshot_kernel, dim3(S.E), dim3(256), 0, h->stream, S.B, S.snap_sum.as<auvp::PrrtSummary>(),
                       S.snap_rng.as<int32_t>(), S.snap_mt.as<uint32_t>());
    HIPCHK(h, hipGetLastError());
    // (ev0 was recorded before the snapshot: the time a caller is told includes it -- every latency-path plan call pays it)
    auto launch_pipe = [&](auto kern) -> hipError_t {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_used);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, dim3(grid_used), dim3(block_used), lds_used, h->stream, h->W, S.P, S.B, S.E, next_lds);
      return hipGetLastError();
    };
    if (draw_wave) {
      if (O <= 64) le = launch_pipe(auvp::prrt_pipe_kernel<1, 5>);
      else if (O <= 128) le = launch_pipe(auvp::prrt_pipe_kernel<2, 5>);
      else le = launch_pipe(auvp::prrt_pipe_kernel<4, 5>);
    } else {
      if (O <= 64) le = launch_pipe(auvp::prrt_pipe_kernel<1, 4>);
      else if (O <= 128) le = launch_pipe(auvp::prrt_pipe_kernel<2, 4>);
      else le = launch_pipe(auvp::prrt_pipe_kernel<4, 4>);
    }
  } else if (lat) {
    if (O <= 64) le = launch(auvp::prrt_kernel<1, true>);
    else if (O <= 128) le = launch(auvp::prrt_kernel<2, true>);
    else if (O <= 256) le = launch(auvp::prrt_kernel<4, true>);
    else if (O <= 512) le = launch(auvp::prrt_kernel<8, true>);
    else le = launch(auvp::prrt_kernel<16, true>);
  } else {
    if (O <= 64) le = launch(auvp::prrt_kernel<1, false>);
    else if (O <= 128) le = launch(auvp::prrt_kernel<2, false>);
    else if (O <= 256) le = launch(auvp::prrt_kernel<4, false>);
    else if (O <= 512) le = launch(auvp::prrt_kernel<8, false>);
    else le = launch(auvp::prrt_kernel<16, false>);
  }
  HIPCHK(h, le);
  if (sync) HIPCHK(h, hipEventRecord(h->ev1, h->stream));
  h->last_grid = grid_used; h->last_block = block_used; h->last_lds = (int)lds_used;
  if (!sync) return AUVP_OK;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  float ms = 0.f;
  HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
  h->last_ms = ms;
  if (use_pipe && h->pipe_failed()) {
    // some episode's pipeline gave up (AUVP_ERR_PIPELINE): those episodes go back to where this launch found them (record,
    // generator, bucket table; the tree is append-only) and are planned by the one-wavefront kernel -- the others, finished, are
    // not touched (redo_mask).  Option PIPE_FALLBACK = 0 leaves the status in the summaries instead.
    h->pipe_clear();
    if (h->opt_flag(OPT_PIPE_FALLBACK, true)) {
      HIPCHK(h, S.redo_count.reserve(sizeof(int32_t)));
      HIPCHK(h, hipMemsetAsync(S.redo_count.p, 0, sizeof(int32_t), h->stream));
      hipLaunchKernelGGL(auvp::prrt_undo_kernel, dim3((S.E + 63) / 64), dim3(64), 0, h->stream, S.P, S.B, S.E, S.snap_sum.as<auvp::PrrtSummary>(),
                         S.snap_rng.as<int32_t>(), S.snap_mt.as<uint32_t>(), S.redo_mask.as<uint8_t>(), S.redo_count.as<int32_t>());
      HIPCHK(h, hipGetLastError());
      int32_t n = 0;
      HIPCHK(h, hipMemcpyAsync(&n, S.redo_count.p, sizeof n, hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
      if (n > 0) {
        h->pipe_fallback_last += n;
        h->pipe_fallback_total += n;
        S.B.redo_mask = S.redo_mask.as<uint8_t>();
        const int rc = prrt_launch(h, S, step_mode, true, true);
        S.B.redo_mask = nullptr;
        h->last_ms += ms;  // (the time the caller waited)
        if (rc != AUVP_OK) return rc;
      }
    }
  }
  return AUVP_OK;
}

}  // namespace

extern "C" {

// parameters + buffers of a batch of E episodes (everything of auvp_prrt_create_batch that does not depend on where
// the starts, goals and generator states come from)
static int prrt_configure(auvp_handle* h, PrrtState& S, int32_t E, const auvp_prrt_params* p, int32_t flags) {
  if (!h->have_world) return fail(h, AUVP_ERR_STATE, "auvp_world_set not called (obstacle list)");
  if (E <= 0 || !p) return fail(h, AUVP_ERR_ARG, "bad batch arguments");
  if (h->W.n_obstacles > 16 * 64) return fail(h, AUVP_ERR_ARG, "n_obstacles %d > 1024", h->W.n_obstacles);
  const int ics = (int)p->cell_side_length;
  if (ics <= 0 || p->subsections <= 0 || p->max_step <= 0 || !(p->freq >= 0)) return fail(h, AUVP_ERR_ARG, "bad params");
  S.ready = false;
  S.drop_graphs();  // they replay launches on the previous batch's buffers
  auvp::PrrtParamsDev& P = S.P;
  for (int i = 0; i < 4; i++) P.rect[i] = p->rect[i];
  P.exp_rate = p->exp_rate; P.dist_to_end = p->dist_to_end; P.diff_max = p->diff_max; P.freq = p->freq;
  P.cell = p->cell_side_length;
  P.S = p->subsections;
  // discretize_env (gym_rrt/envs/rrt_dubins.py:77-93): int(height) // int(cell), int(width) // int(cell)
  P.rows = (int)(p->rect[3] - p->rect[1]) / ics;
  P.cols = (int)(p->rect[2] - p->rect[0]) / ics;
  if (P.rows <= 0 || P.cols <= 0) return fail(h, AUVP_ERR_ARG, "empty grid");
  const double nbd = (double)P.rows * P.cols * P.S;
  if (nbd > 5.0e7) return fail(h, AUVP_ERR_ARG, "too many buckets");
  P.n_buckets = (int)nbd;
  P.max_step = p->max_step; P.flags = flags; P.step_mode = 0;
  P.delta_theta = (double)(2.0 * M_PI) / (double)P.S;  // grid_cell_rrt.py:49
  auvp::PrrtBuffers& B = S.B;
  const int nfreq = (int)std::floor(p->freq);
  B.cap_nodes = p->max_step + 1;
  B.max_pts = nfreq + 3;
  double cp = (double)p->max_step * (double)(nfreq > 0 ? nfreq : 1) + 64;  // every taken sub-arc is stored
  if (cp > 2.0e9) return fail(h, AUVP_ERR_ARG, "point capacity too large");
  B.cap_points = (int32_t)cp;
  if (B.cap_nodes >= (1 << 24)) return fail(h, AUVP_ERR_ARG, "max_step %d: a bucket's size is kept in 24 bits", p->max_step);
  const size_t cn = (size_t)E * B.cap_nodes;
  HIPCHK(h, S.nodes.reserve(cn * sizeof(auvp::PrrtNode)));
  HIPCHK(h, S.node_bucket.reserve(cn * sizeof(int32_t)));
  HIPCHK(h, S.points.reserve((size_t)E * B.cap_points * 4 * sizeof(double)));
  HIPCHK(h, S.occupied.reserve(cn * sizeof(int32_t)));
  HIPCHK(h, S.buckets.reserve((size_t)E * P.n_buckets * sizeof(int2)));
  HIPCHK(h, S.mt.reserve((size_t)E * 624 * sizeof(uint32_t)));
  HIPCHK(h, S.rng_state.reserve((size_t)E * 4 * sizeof(int32_t)));
  HIPCHK(h, S.summary.reserve((size_t)E * sizeof(auvp::PrrtSummary)));
  HIPCHK(h, S.step_bucket.reserve((size_t)E * sizeof(int32_t)));
  HIPCHK(h, S.start.reserve((size_t)E * 4 * sizeof(double)));
  HIPCHK(h, S.goal.reserve((size_t)E * 2 * sizeof(double)));
  B.nodes = S.nodes.as<auvp::PrrtNode>(); B.node_bucket = S.node_bucket.as<int32_t>();
  B.points = S.points.as<double>(); B.occupied = S.occupied.as<int32_t>(); B.buckets = S.buckets.as<int2>();
  B.mt = S.mt.as<uint32_t>(); B.rng_state = S.rng_state.as<int32_t>(); B.summary = S.summary.as<auvp::PrrtSummary>();
  B.step_bucket = S.step_bucket.as<int32_t>();
  B.start = S.start.as<double>(); B.goal = S.goal.as<double>();
  B.st_log = nullptr;
  B.env_flags = 0; B._pad_env = 0; B.env_done = nullptr; B.env_reward = nullptr; B.env_done_out = nullptr;
  B.env_bucket_out = nullptr; B.env_agent_seed = 0ull; B.env_err = nullptr;
  B.pipe_fail = h->pipe_fail_dev; B.redo_mask = nullptr;
  B.env_obs_grid = nullptr; B.env_obs_has = nullptr; B.env_obs_num = nullptr;
  if (flags & AUVP_FLAG_ITER_LOG) {
    HIPCHK(h, S.st_log.reserve((size_t)E * p->max_step * 8 * sizeof(int32_t)));
    HIPCHK(h, hipMemsetAsync(S.st_log.p, 0xff, (size_t)E * p->max_step * 8 * sizeof(int32_t), h->stream));
    B.st_log = S.st_log.as<int32_t>();
  }
  {
    // throughput batches (more than twelve episodes per CU) of the environment's planner shape run four episodes per
    // wavefront (planner_rows_kernel.h); option PRRT_ROWS = 0 / 1 forces the choice where the kernel's limits allow it.
    // (Twelve: re-measured at the end of round 6 on config 4's world, M steps/s one wavefront per episode (latency
    // instantiation) / four episodes per wavefront: 2 048 episodes 276 / 178, 3 072: 334 / 262, 4 096: 311 / 342, 8 192: 359 / 588
    // -- tools/prrt_batch_probe.py; the threshold had been eight per CU.)
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, h->device);
    if (n_cu <= 0) n_cu = 256;
    const bool rows_ok = nfreq <= auvp::PRW_MAX_FREQ && h->W.n_obstacles <= auvp::RW_MAX_OBST && !(flags & AUVP_FLAG_ITER_LOG);
    const bool lat = h->opt_flag(OPT_PRRT_LAT, E <= 12 * n_cu);
    S.use_rows = rows_ok && h->opt_flag(OPT_PRRT_ROWS, !lat);
  }
  return AUVP_OK;
}

// mps_list = [start]; add_node_to_grid(start)  (:53,:108-159): one thread per episode places the start node with the
// kernel's own add_node_to_grid arithmetic (starts / goals / generator states are already on the device)
static int prrt_plant(auvp_handle* h, PrrtState& S, int32_t E) {
  {
    // every bucket of the new batch is empty: a fresh epoch does that without touching the table (E x n_buckets x 8 bytes);
    // the table itself is cleared when its allocation changed or the 8-bit tag wraps (tag 0 is never current)
    const size_t need = (size_t)E * S.P.n_buckets * sizeof(int2);
    if (S.bucket_alloc != S.buckets.p || need > S.bucket_alloc_bytes || S.bucket_epoch >= 255) {
      HIPCHK(h, hipMemsetAsync(S.buckets.p, 0, need, h->stream));
      S.bucket_alloc = S.buckets.p; S.bucket_alloc_bytes = need; S.bucket_epoch = 0;
    }
    S.bucket_epoch++;
    S.B.bucket_epoch = S.bucket_epoch;
  }
  HIPCHK(h, S.env_done.reserve((size_t)E));
  HIPCHK(h, S.env_err.reserve(2 * sizeof(int32_t)));
  S.thetas_ready = false;
  // (the planting launch also clears the device-resident loop's finished flags and error word: no fill launches)
  hipLaunchKernelGGL(auvp::prrt_init_kernel, dim3((E + 255) / 256), dim3(256), 0, h->stream, S.P, S.B, (int)E, S.env_done.as<uint8_t>(),
                     S.env_err.as<int32_t>());
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  S.E = E;
  S.ready = true;
  return AUVP_OK;
}

int auvp_prrt_create_batch(auvp_handle* h, int32_t E, const double* starts, const double* goals,
                           const auvp_prrt_params* p, const uint64_t* seeds, const uint32_t* mt, const int32_t* mt_index,
                           int32_t flags) {
  if (!h) return AUVP_ERR_ARG;
  if (E <= 0 || !starts || !goals || !p || (!seeds && !(mt && mt_index))) return fail(h, AUVP_ERR_ARG, "bad batch arguments");
  HIPCHK(h, hipSetDevice(h->device));
  PrrtState& S = *prrt_of(h);
  int rc;
  if ((rc = prrt_configure(h, S, E, p, flags))) return rc;
  if ((rc = upload(h, S.start, starts, (size_t)E * 4))) return rc;
  if ((rc = upload(h, S.goal, goals, (size_t)E * 2))) return rc;
  // generator states: random.seed(seeds[e]) on the device (init_by_array, 64 generators per workgroup), or the caller's own
  if (seeds) {
    HIPCHK(h, S.seeds.reserve((size_t)E * sizeof(uint64_t)));
    HIPCHK(h, hipMemcpyAsync(S.seeds.p, seeds, (size_t)E * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(auvp::mt_seed_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  auvp::MT_SEED_LDS));
    hipLaunchKernelGGL(auvp::mt_seed_kernel, dim3((E + 63) / 64), dim3(64), auvp::MT_SEED_LDS, h->stream,
                       S.seeds.as<unsigned long long>(), S.B.mt, S.B.rng_state, (int)E);
    HIPCHK(h, hipGetLastError());
  } else {
    std::vector<uint32_t> words((size_t)E * 624);
    std::vector<int32_t> rs((size_t)E * 4, 0);
    for (int e = 0; e < E; e++) {
      memcpy(words.data() + (size_t)e * 624, mt + (size_t)e * 624, 624 * sizeof(uint32_t));
      int idx = mt_index[e] < 0 ? 0 : (mt_index[e] > 624 ? 624 : mt_index[e]);
      rs[4 * (size_t)e] = idx == 624 ? 0 : idx;
      rs[4 * (size_t)e + 1] = 624 - idx;
    }
    if ((rc = upload(h, S.mt, words.data(), words.size()))) return rc;
    if ((rc = upload(h, S.rng_state, rs.data(), rs.size()))) return rc;
  }
  return prrt_plant(h, S, E);
}

int auvp_prrt_plan(auvp_handle* h) {
  if (!h) return AUVP_ERR_ARG;
  PrrtState& S = *prrt_of(h);
  if (!S.ready) return fail(h, AUVP_ERR_STATE, "auvp_prrt_create_batch not called");
  HIPCHK(h, hipSetDevice(h->device));
  return prrt_launch(h, S, 0);
}

int auvp_prrt_step(auvp_handle* h, const int32_t* bucket_ids, const uint32_t* mt, const int32_t* mt_index) {
  if (!h || !bucket_ids) return AUVP_ERR_ARG;
  PrrtState& S = *prrt_of(h);
  if (!S.ready) return fail(h, AUVP_ERR_STATE, "auvp_prrt_create_batch not called");
  HIPCHK(h, hipSetDevice(h->device));
  int rc;
  if ((rc = upload(h, S.step_bucket, bucket_ids, (size_t)S.E))) return rc;
  if (mt && mt_index) {  // continue an external generator (the global `random` state) for this step
    std::vector<int32_t> rs((size_t)S.E * 4, 0);
    for (int e = 0; e < S.E; e++) {
      int idx = mt_index[e] < 0 ? 0 : (mt_index[e] > 624 ? 624 : mt_index[e]);
      rs[4 * (size_t)e] = idx == 624 ? 0 : idx;
      rs[4 * (size_t)e + 1] = 624 - idx;
    }
    if ((rc = upload(h, S.mt, mt, (size_t)S.E * 624))) return rc;
    if ((rc = upload(h, S.rng_state, rs.data(), rs.size()))) return rc;
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return prrt_launch(h, S, 1);
}

int auvp_prrt_summaries(auvp_handle* h, auvp_prrt_summary* out) {
  if (!h || !out) return AUVP_ERR_ARG;
  PrrtState& S = *prrt_of(h);
  if (!S.ready) return fail(h, AUVP_ERR_STATE, "no planner batch");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpy(out, S.B.summary, (size_t)S.E * sizeof(auvp::PrrtSummary), hipMemcpyDeviceToHost));
  return AUVP_OK;
}

int auvp_prrt_goals(auvp_handle* h, double* goals2) {
  if (!h || !goals2) return AUVP_ERR_ARG;
  PrrtState& S = *prrt_of(h);
  if (!S.ready) return fail(h, AUVP_ERR_STATE, "no planner batch");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpy(goals2, S.B.goal, (size_t)S.E * 2 * sizeof(double), hipMemcpyDeviceToHost));
  return AUVP_OK;
}

int auvp_prrt_paths(auvp_handle* h, const int64_t* offsets, double* out) {
  if (!h || !offsets || !out) return AUVP_ERR_ARG;
  PrrtState& S = *prrt_of(h);
  if (!S.ready) return fail(h, AUVP_ERR_STATE, "no planner batch");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t total = (size_t)offsets[S.E];
  int rc;
  if ((rc = upload(h, S.tmp_off, offsets, (size_t)S.E + 1))) return rc;
  HIPCHK(h, S.tmp_out.reserve(std::max<size_t>(total, 1) * 5 * sizeof(double)));
  hipLaunchKernelGGL(auvp::prrt_final_course_kernel, dim3(S.E), dim3(64), 0, h->stream, S.B, S.tmp_off.as<int64_t>(),
                     S.tmp_out.as<double>(), S.E);
  HIPCHK(h, hipGetLastError());
  if (total) HIPCHK(h, hipMemcpyAsync(out, S.tmp_out.p, total * 5 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return AUVP_OK;
}

int auvp_prrt_tree(auvp_handle* h, int32_t ep, double* nodes4, int32_t* node_i4, int32_t* node_bucket, double* points4) {
  if (!h) return AUVP_ERR_ARG;
  PrrtState& S = *prrt_of(h);
  if (!S.ready || ep < 0 || ep >= S.E) return fail(h, AUVP_ERR_STATE, "bad episode");
  HIPCHK(h, hipSetDevice(h->device));
  auvp::PrrtSummary s;
  HIPCHK(h, hipMemcpy(&s, S.B.summary + ep, sizeof s, hipMemcpyDeviceToHost));
  const size_t N = (size_t)s.n_nodes, NP = (size_t)s.n_points, capp = (size_t)S.B.cap_points;
  if (nodes4 || node_i4 || node_bucket) {
    std::vector<auvp::PrrtNode> rec(N);
    HIPCHK(h, hipMemcpy(rec.data(), S.B.nodes + (size_t)ep * S.B.cap_nodes, N * sizeof(auvp::PrrtNode), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < N; i++) {
      const auvp::PrrtNode& r = rec[i];
      if (nodes4) { nodes4[4 * i] = r.x; nodes4[4 * i + 1] = r.y; nodes4[4 * i + 2] = r.theta; nodes4[4 * i + 3] = r.t; }
      if (node_i4) { node_i4[4 * i] = r.step; node_i4[4 * i + 1] = r.parent; node_i4[4 * i + 2] = r.pt_off; node_i4[4 * i + 3] = r.pt_cnt; }
      if (node_bucket) node_bucket[i] = r.bucket;
    }
  }
  if (points4 && NP) HIPCHK(h, hipMemcpy(points4, S.B.points + (size_t)ep * capp * 4, NP * 4 * sizeof(double), hipMemcpyDeviceToHost));
  return AUVP_OK;
}

// one node of one episode: its record and its run of path points (what a step-mode caller needs to materialise the
// node the device just accepted, instead of downloading the whole tree every step)
int auvp_prrt_node(auvp_handle* h, int32_t ep, int32_t node, double* node4, int32_t* node_i4, int32_t* node_bucket,
                   double* points4, int32_t cap_points) {
  if (!h) return AUVP_ERR_ARG;
  PrrtState& S = *prrt_of(h);
  if (!S.ready || ep < 0 || ep >= S.E) return fail(h, AUVP_ERR_STATE, "bad episode");
  if (node < 0 || node >= S.B.cap_nodes) return fail(h, AUVP_ERR_ARG, "node %d outside 0..%d", node, S.B.cap_nodes - 1);
  HIPCHK(h, hipSetDevice(h->device));
  const size_t at = (size_t)ep * S.B.cap_nodes + (size_t)node, capp = (size_t)S.B.cap_points;
  auvp::PrrtNode r;
  HIPCHK(h, hipMemcpy(&r, S.B.nodes + at, sizeof r, hipMemcpyDeviceToHost));
  if (node4) { node4[0] = r.x; node4[1] = r.y; node4[2] = r.theta; node4[3] = r.t; }
  if (node_i4) { node_i4[0] = r.step; node_i4[1] = r.parent; node_i4[2] = r.pt_off; node_i4[3] = r.pt_cnt; }
  if (node_bucket) *node_bucket = r.bucket;
  const int off = r.pt_off, cnt = r.pt_cnt;
  if (points4 && cnt > 0) {
    if (cnt > cap_points) return fail(h, AUVP_ERR_CAPACITY, "node has %d points, buffer holds %d", cnt, cap_points);
    if (off < 0 || (size_t)off + (size_t)cnt > capp) return fail(h, AUVP_ERR_STATE, "node record out of range");
    HIPCHK(h, hipMemcpy(points4, S.B.points + ((size_t)ep * capp + (size_t)off) * 4, (size_t)cnt * 4 * sizeof(double), hipMemcpyDeviceToHost));
  }
  return AUVP_OK;
}

int auvp_prrt_grid(auvp_handle* h, int32_t ep, int32_t* occupied, int32_t* bucket_counts, int32_t* dims4) {
  if (!h) return AUVP_ERR_ARG;
  PrrtState& S = *prrt_of(h);
  if (!S.ready || ep < 0 || ep >= S.E) return fail(h, AUVP_ERR_STATE, "bad episode");
  HIPCHK(h, hipSetDevice(h->device));
  auvp::PrrtSummary s;
  HIPCHK(h, hipMemcpy(&s, S.B.summary + ep, sizeof s, hipMemcpyDeviceToHost));
  if (dims4) { dims4[0] = S.P.rows; dims4[1] = S.P.cols; dims4[2] = S.P.S; dims4[3] = s.n_occ; }
  if (occupied && s.n_occ) HIPCHK(h, hipMemcpy(occupied, S.B.occupied + (size_t)ep * S.B.cap_nodes, (size_t)s.n_occ * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (bucket_counts) {
    std::vector<int2> tab((size_t)S.P.n_buckets);
    HIPCHK(h, hipMemcpy(tab.data(), S.B.buckets + (size_t)ep * S.P.n_buckets, tab.size() * sizeof(int2), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < tab.size(); i++)
      bucket_counts[i] = (int32_t)((uint32_t)tab[i].x >> 24) == S.B.bucket_epoch ? (tab[i].x & 0xffffff) : 0;
  }
  return AUVP_OK;
}

int auvp_prrt_step_log(auvp_handle* h, int32_t ep, int32_t* log8) {
  if (!h || !log8) return AUVP_ERR_ARG;
  PrrtState& S = *prrt_of(h);
  if (!S.ready || ep < 0 || ep >= S.E || !S.B.st_log) return fail(h, AUVP_ERR_STATE, "no step log");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpy(log8, S.B.st_log + (size_t)ep * S.P.max_step * 8, (size_t)S.P.max_step * 8 * sizeof(int32_t), hipMemcpyDeviceToHost));
  return AUVP_OK;
}

const char* auvp_prrt_last_kernel(auvp_handle* h) {
  if (!h) return "";
  return prrt_of(h)->last_kernel;
}

void* auvp_prrt_summaries_dev(auvp_handle* h) {
  if (!h) return nullptr;
  PrrtState& S = *prrt_of(h);
  return S.ready ? (void*)S.B.summary : nullptr;
}

// RRTEnv observation arrays (gym_rrt/envs/rrt_env.py:250-295) for every episode, written to
// caller-owned DEVICE buffers: rrt_grid [E,n_buckets,4] f64 = cell.x, cell.y, subsection.theta,
// len(node_array); has_node [E,n_buckets] i64; num_nodes [E,n_buckets] i64.
static int prrt_observation_enqueue(auvp_handle* h, PrrtState& S, void* rrt_grid_dev, void* has_node_dev, void* num_nodes_dev) {
  if (!S.thetas_ready) {
    // subsection thetas exactly as Grid_cell_RRT builds them (grid_cell_rrt.py:49-55); once per batch
    std::vector<double> th(S.P.S);
    const double pi = M_PI;
    double theta = 0.0;
    for (int i = 0; i < S.P.S; i++) {
      th[i] = theta;
      theta = theta + S.P.delta_theta;
      while (!(-pi <= theta && theta <= pi)) theta += theta > pi ? (-2 * pi) : (2 * pi);
    }
    int rc;
    if ((rc = upload(h, S.thetas, th.data(), th.size()))) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));  // `th` goes out of scope
    S.thetas_ready = true;
  }
  const long long total = (long long)S.E * S.P.n_buckets;
  const int grid = (int)std::min<long long>((total + 255) / 256, 65535LL * 16);
  hipLaunchKernelGGL(auvp::prrt_observation_kernel, dim3(grid), dim3(256), 0, h->stream, S.P, S.B, S.thetas.as<double>(), S.E,
                     reinterpret_cast<double*>(rrt_grid_dev), reinterpret_cast<long long*>(has_node_dev),
                     reinterpret_cast<long long*>(num_nodes_dev));
  HIPCHK(h, hipGetLastError());
  return AUVP_OK;
}

int auvp_prrt_observation_dev(auvp_handle* h, void* rrt_grid_dev, void* has_node_dev, void* num_nodes_dev) {
  if (!h || !rrt_grid_dev) return AUVP_ERR_ARG;
  PrrtState& S = *prrt_of(h);
  if (!S.ready) return fail(h, AUVP_ERR_STATE, "no planner batch");
  HIPCHK(h, hipSetDevice(h->device));
  int rc = prrt_observation_enqueue(h, S, rrt_grid_dev, has_node_dev, num_nodes_dev);
  if (rc != AUVP_OK) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return AUVP_OK;
}

// ---- the device-resident RRTEnv loop: every call below only ENQUEUES on the handle's stream (auvp_stream_sync waits) ----
// One environment step = generate_one_node for every live environment, with the step's outcome (reward, done flag) written
// by the same launch and -- AUVP_ENV_AGENT -- the stand-in agent's pick made inside it.  The observation arrays are either
// rewritten whole by a second launch (E x n_buckets entries, what the reference env rebuilds after every node) or --
// AUVP_ENV_OBS_DELTA -- updated in place by the planner launch itself: a step changes ONE bucket per environment.  (Batches
// the four-episodes-per-wavefront kernel serves keep the agent and the outcome as small launches of their own and always
// rewrite the observation.)
static int prrt_env_step_enqueue(auvp_handle* h, PrrtState& S, int32_t flags, uint64_t agent_seed, int32_t* bucket_ids_dev,
                                 void* rrt_grid_dev, void* has_node_dev, void* num_nodes_dev, int64_t* reward_dev, uint8_t* done_dev) {
  const bool agent = (flags & AUVP_ENV_AGENT) != 0;
  const bool delta = (flags & AUVP_ENV_OBS_DELTA) != 0 && rrt_grid_dev != nullptr;
  auvp::PrrtBuffers& B = S.B;
  B.env_done = S.env_done.as<uint8_t>(); B.env_reward = reinterpret_cast<long long*>(reward_dev); B.env_done_out = done_dev;
  B.env_bucket_out = bucket_ids_dev; B.env_agent_seed = agent_seed; B.env_err = S.env_err.as<int32_t>();
  B.env_obs_grid = reinterpret_cast<double*>(rrt_grid_dev); B.env_obs_has = reinterpret_cast<long long*>(has_node_dev);
  B.env_obs_num = reinterpret_cast<long long*>(num_nodes_dev);
  const int32_t* own = B.step_bucket;
  B.step_bucket = bucket_ids_dev;
  int rc;
  bool full_obs = rrt_grid_dev != nullptr && !delta;
  if (!S.use_rows) {
    B.env_flags = PRRT_ENV_OUTCOME | (agent ? PRRT_ENV_AGENT : 0) | (delta ? PRRT_ENV_DELTA : 0);
    rc = prrt_launch(h, S, 1, false);
  } else {
    B.env_flags = 0;
    full_obs = rrt_grid_dev != nullptr;
    if (agent) {
      hipLaunchKernelGGL(auvp::prrt_policy_random_kernel, dim3((S.E + 255) / 256), dim3(256), 0, h->stream, B, S.E,
                         (unsigned long long)agent_seed, bucket_ids_dev);
      HIPCHK(h, hipGetLastError());
    }
    rc = prrt_launch(h, S, 1, false);
    if (rc == AUVP_OK) {
      hipLaunchKernelGGL(auvp::prrt_env_outcome_kernel, dim3((S.E + 255) / 256), dim3(256), 0, h->stream, B, S.E, bucket_ids_dev);
      rc = hipGetLastError() == hipSuccess ? AUVP_OK : fail(h, AUVP_ERR_HIP, "prrt_env_outcome_kernel launch");
    }
  }
  B.step_bucket = own;
  B.env_flags = 0;
  if (rc != AUVP_OK) return rc;
  if (full_obs) return prrt_observation_enqueue(h, S, rrt_grid_dev, has_node_dev, num_nodes_dev);
  return AUVP_OK;
}

int auvp_prrt_env_step_dev(auvp_handle* h, const int32_t* bucket_ids_dev, void* rrt_grid_dev, void* has_node_dev, void* num_nodes_dev,
                           int64_t* reward_dev, uint8_t* done_dev) {
  return auvp_prrt_env_step_ex_dev(h, 0, 0, const_cast<int32_t*>(bucket_ids_dev), rrt_grid_dev, has_node_dev, num_nodes_dev, reward_dev, done_dev);
}

int auvp_prrt_env_step_ex_dev(auvp_handle* h, int32_t flags, uint64_t agent_seed, int32_t* bucket_ids_dev, void* rrt_grid_dev,
                              void* has_node_dev, void* num_nodes_dev, int64_t* reward_dev, uint8_t* done_dev) {
  if (!h || !bucket_ids_dev || !reward_dev || (flags & ~(AUVP_ENV_AGENT | AUVP_ENV_OBS_DELTA))) return AUVP_ERR_ARG;
  PrrtState& S = *prrt_of(h);
  if (!S.ready) return fail(h, AUVP_ERR_STATE, "auvp_prrt_create_batch not called");
  HIPCHK(h, hipSetDevice(h->device));
  return prrt_env_step_enqueue(h, S, flags, agent_seed, bucket_ids_dev, rrt_grid_dev, has_node_dev, num_nodes_dev, reward_dev, done_dev);
}

int auvp_prrt_policy_random_dev(auvp_handle* h, const int64_t* has_node_dev, uint64_t seed, int32_t* bucket_ids_dev) {
  (void)has_node_dev;  // the agent reads the planner's own list of occupied buckets: the set has_node marks, without the scan
  if (!h || !bucket_ids_dev) return AUVP_ERR_ARG;
  PrrtState& S = *prrt_of(h);
  if (!S.ready) return fail(h, AUVP_ERR_STATE, "no planner batch");
  HIPCHK(h, hipSetDevice(h->device));
  auvp::PrrtBuffers B = S.B;
  B.env_done = S.env_done.as<uint8_t>();
  hipLaunchKernelGGL(auvp::prrt_policy_random_kernel, dim3((S.E + 255) / 256), dim3(256), 0, h->stream, B, S.E, (unsigned long long)seed,
                     bucket_ids_dev);
  HIPCHK(h, hipGetLastError());
  return AUVP_OK;
}

// {status, environment} of the first episode that failed on the device inside the device-resident loop (0: none).  Waits
// for the stream.  The host loop (auvp_prrt_step + summaries) surfaces the same failures as summary.status < 0.
int auvp_prrt_env_check(auvp_handle* h, int32_t* status, int32_t* env) {
  if (!h || !status) return AUVP_ERR_ARG;
  PrrtState& S = *prrt_of(h);
  if (!S.ready) return fail(h, AUVP_ERR_STATE, "no planner batch");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  int32_t e2[2] = {0, 0};
  HIPCHK(h, hipMemcpy(e2, S.env_err.p, sizeof e2, hipMemcpyDeviceToHost));
  *status = e2[0];
  if (env) *env = e2[1];
  return AUVP_OK;
}

int auvp_stream_sync(auvp_handle* h) {
  if (!h) return AUVP_ERR_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return AUVP_OK;
}

void* auvp_stream(auvp_handle* h) { return h ? (void*)h->stream : nullptr; }

// HIP-event time of a region of the handle's stream the CALLER brackets (enqueue-only loops: auvp_prrt_env_step_dev,
// auvp_graph_launch): auvp_stream_mark(h, 0) before, auvp_stream_mark(h, 1) after, auvp_stream_elapsed_ms waits and reads.
int auvp_stream_mark(auvp_handle* h, int32_t which) {
  if (!h || (which != 0 && which != 1)) return AUVP_ERR_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipEventRecord(which ? h->ev1 : h->ev0, h->stream));
  return AUVP_OK;
}

int auvp_stream_elapsed_ms(auvp_handle* h, double* ms) {
  if (!h || !ms) return AUVP_ERR_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipEventSynchronize(h->ev1));
  float f = 0.f;
  HIPCHK(h, hipEventElapsedTime(&f, h->ev0, h->ev1));
  *ms = f;
  return AUVP_OK;
}

// ---- hipGraph capture of a launch-bound inner loop on the handle's stream ------------------------------------------------
// Between auvp_graph_begin and auvp_graph_end every enqueue-only entry point (auvp_prrt_policy_random_dev,
// auvp_prrt_env_step_dev, a caller's own kernels on auvp_stream) is recorded instead of run; auvp_graph_launch replays the
// recorded step n times back to back.  Call the entry points once un-captured first (they allocate and upload on first use).
int auvp_graph_begin(auvp_handle* h) {
  if (!h) return AUVP_ERR_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
  return AUVP_OK;
}

int auvp_graph_end(auvp_handle* h, int32_t* graph_id) {
  if (!h || !graph_id) return AUVP_ERR_ARG;
  PrrtState& S = *prrt_of(h);
  hipGraph_t g = nullptr;
  HIPCHK(h, hipStreamEndCapture(h->stream, &g));
  hipGraphExec_t ge = nullptr;
  hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return fail(h, AUVP_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
  S.graphs.push_back(ge);
  *graph_id = (int32_t)S.graphs.size() - 1;
  return AUVP_OK;
}

int auvp_graph_launch(auvp_handle* h, int32_t graph_id, int32_t n_times) {
  if (!h || n_times < 0) return AUVP_ERR_ARG;
  PrrtState& S = *prrt_of(h);
  if (graph_id < 0 || graph_id >= (int32_t)S.graphs.size()) return fail(h, AUVP_ERR_ARG, "no such graph");
  if (!S.graphs[graph_id]) return fail(h, AUVP_ERR_STATE, "graph %d was captured for an earlier batch (a new batch invalidates its graphs)", graph_id);
  HIPCHK(h, hipSetDevice(h->device));
  for (int i = 0; i < n_times; i++) HIPCHK(h, hipGraphLaunch(S.graphs[graph_id], h->stream));
  return AUVP_OK;
}

// the same for one episode, copied to host arrays
int auvp_prrt_observation(auvp_handle* h, int32_t ep, double* rrt_grid, int64_t* has_node, int64_t* num_nodes) {
  if (!h || !rrt_grid) return AUVP_ERR_ARG;
  PrrtState& S = *prrt_of(h);
  if (!S.ready || ep < 0 || ep >= S.E) return fail(h, AUVP_ERR_STATE, "bad episode");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t nb = (size_t)S.P.n_buckets, tot = (size_t)S.E * nb;
  DevBuf g, hn, nn;
  HIPCHK(h, g.reserve(tot * 4 * sizeof(double)));
  HIPCHK(h, hn.reserve(tot * sizeof(int64_t)));
  HIPCHK(h, nn.reserve(tot * sizeof(int64_t)));
  int rc = auvp_prrt_observation_dev(h, g.p, hn.p, nn.p);
  if (rc != AUVP_OK) return rc;
  HIPCHK(h, hipMemcpy(rrt_grid, g.as<double>() + (size_t)ep * nb * 4, nb * 4 * sizeof(double), hipMemcpyDeviceToHost));
  if (has_node) HIPCHK(h, hipMemcpy(has_node, hn.as<int64_t>() + (size_t)ep * nb, nb * sizeof(int64_t), hipMemcpyDeviceToHost));
  if (num_nodes) HIPCHK(h, hipMemcpy(num_nodes, nn.as<int64_t>() + (size_t)ep * nb, nb * sizeof(int64_t), hipMemcpyDeviceToHost));
  return AUVP_OK;
}

}  // extern "C"
#endif
