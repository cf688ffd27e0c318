// astar_host.h -- C-ABI entry points for the A* variants (auvp_astar_*), host side.
// Included at the end of auvplan.hip (same translation unit: uses auvp_handle, fail, HIPCHK, upload).
#ifndef AUVP_ASTAR_HOST_H
#define AUVP_ASTAR_HOST_H

static_assert(sizeof(auvp_astar_summary) == sizeof(auvp::AstarSummary), "astar summary layout");

namespace {

struct AstarState {
  bool ready = false;
  unsigned world_version = ~0u;
  int E = 0;
  auvp::AstarWorldDev W{};
  auvp::AstarParamsDev P{};
  auvp::AstarBuffers B{};
  DevBuf ox, oy, ot, hab, poly, bins, rcells, prob, topn, gridtab;
  DevBuf start, goal, limit, nodes, node_i, cellinfo, hab_left, exp_log, summary, off, path, cost, npath, smooth;
  long long visited_set_for = -1;  // entry count of a bitmap uploaded by auvp_astar_set_visited, -1 none
  // epoch of the words in `cellinfo` (astar_kernel.h): bumped per batch instead of clearing 1.4 MB per instance;
  // cellinfo_clean_cap = capacity (bytes) that has been zeroed since the buffer was (re)allocated
  uint32_t epoch = 0;
  size_t cellinfo_clean_cap = 0;
};

AstarState* astar_of(auvp_handle* h) {
  if (!h->astar) {
    h->astar = new AstarState();
    h->astar_free = [](void* p) { delete static_cast<AstarState*>(p); };
  }
  return static_cast<AstarState*>(h->astar);
}

// Python round(v, 2): decimal rounding of the exact binary value (float.__round__ -> dtoa mode 3)
double round2_py(double v) {
  char buf[512];
  snprintf(buf, sizeof buf, "%.2f", v);
  return strtod(buf, nullptr);
}

int astar_build_world(auvp_handle* h, AstarState& S) {
  const int O = (int)(h->w_obst.size() / 3), H = (int)(h->w_hab.size() / 3), V = (int)(h->w_poly.size() / 2);
  const int T = (int)(h->w_bins.size() / 2), C = (int)(h->w_cells.size() / 4);
  if (H > auvp::ASTAR_MAX_HAB) return fail(h, AUVP_ERR_ARG, "n_habitats %d > %d", H, auvp::ASTAR_MAX_HAB);
  std::vector<double> ox(O), oy(O), ot(O), hab((size_t)H * 4), rc((size_t)C * 4), topn((size_t)T * (C + 1));
  for (int i = 0; i < O; i++) {
    ox[i] = h->w_obst[3 * i]; oy[i] = h->w_obst[3 * i + 1];
    ot[i] = sq_threshold(h->w_obst[3 * i + 2]);  // plain `d <= obstacle.size` per obstacle (no shared dList here)
  }
  for (int i = 0; i < H; i++) {
    hab[4 * i] = h->w_hab[3 * i]; hab[4 * i + 1] = h->w_hab[3 * i + 1]; hab[4 * i + 2] = h->w_hab[3 * i + 2];
    hab[4 * i + 3] = sq_threshold(h->w_hab[3 * i + 2]);
  }
  for (size_t i = 0; i < rc.size(); i++) rc[i] = round2_py(h->w_cells[i]);
  std::vector<double> tmp(C);
  for (int t = 0; t < T; t++) {
    std::copy(h->w_prob.begin() + (size_t)t * C, h->w_prob.begin() + (size_t)(t + 1) * C, tmp.begin());
    std::sort(tmp.begin(), tmp.end(), [](double a, double b) { return a > b; });
    double tot = 0.0;  // total = 0; total += probabilities[index]  (astar_fixLenSOG.py:526-530)
    topn[(size_t)t * (C + 1)] = 0.0;
    for (int i = 0; i < C; i++) { tot += tmp[i]; topn[(size_t)t * (C + 1) + i + 1] = tot; }
  }
  // Polygon(...).centroid: area-weighted (shoelace) centroid -- shapely is absent; DESIGN.md
  double a2 = 0.0, sx = 0.0, sy = 0.0;
  for (int i = 0; i < V; i++) {
    double x0 = h->w_poly[2 * i], y0 = h->w_poly[2 * i + 1];
    double x1 = h->w_poly[2 * ((i + 1) % V)], y1 = h->w_poly[2 * ((i + 1) % V) + 1];
    double cr = x0 * y1 - x1 * y0;
    a2 += cr;
    sx += (x0 + x1) * cr;
    sy += (y0 + y1) * cr;
  }
  // product grid?  row-major list of ncol x nrow cells whose x interval depends on the column only and whose y interval
  // on the row only, edges ascending, neighbouring intervals touching at most, no degenerate widths (astar_kernel.h)
  int g_ncol = 0, g_nrow = 0;
  std::vector<double> gtab;
  if (C > 0 && !h->opt_on(OPT_ASTAR_NO_GRID)) {
    int nc = 1;
    while (nc < C && rc[4 * (size_t)nc + 1] == rc[1] && rc[4 * (size_t)nc + 3] == rc[3]) nc++;
    if (C % nc == 0) {
      const int nr = C / nc;
      gtab.resize(2 * (size_t)nc + 2 * (size_t)nr);
      double* X0 = gtab.data(); double* X1 = X0 + nc; double* Y0 = X1 + nc; double* Y1 = Y0 + nr;
      for (int c = 0; c < nc; c++) { X0[c] = rc[4 * (size_t)c]; X1[c] = rc[4 * (size_t)c + 2]; }
      for (int r = 0; r < nr; r++) { Y0[r] = rc[4 * (size_t)r * nc + 1]; Y1[r] = rc[4 * (size_t)r * nc + 3]; }
      bool ok = true;
      for (int r = 0; r < nr && ok; r++)
        for (int c = 0; c < nc && ok; c++) {
          const double* q = &rc[4 * ((size_t)r * nc + c)];
          ok = q[0] == X0[c] && q[2] == X1[c] && q[1] == Y0[r] && q[3] == Y1[r];
        }
      auto axis_ok = [](const double* lo, const double* hi, int n) {
        double scale = 1.0;
        for (int i = 0; i < n; i++) {
          if (!std::isfinite(lo[i]) || !std::isfinite(hi[i])) return false;
          scale = std::fmax(scale, std::fmax(std::fabs(lo[i]), std::fabs(hi[i])));
        }
        for (int i = 0; i < n; i++) {
          if (!(hi[i] - lo[i] >= 1e-6 * scale)) return false;
          if (i + 1 < n && !(lo[i + 1] >= hi[i])) return false;
        }
        return true;
      };
      if (ok && axis_ok(X0, X1, nc) && axis_ok(Y0, Y1, nr)) { g_ncol = nc; g_nrow = nr; }
    }
  }
  int rc_;
  if (g_ncol > 0 && (rc_ = upload(h, S.gridtab, gtab.data(), gtab.size()))) return rc_;
  if ((rc_ = upload(h, S.ox, ox.data(), ox.size()))) return rc_;
  if ((rc_ = upload(h, S.oy, oy.data(), oy.size()))) return rc_;
  if ((rc_ = upload(h, S.ot, ot.data(), ot.size()))) return rc_;
  if ((rc_ = upload(h, S.hab, hab.data(), hab.size()))) return rc_;
  if ((rc_ = upload(h, S.poly, h->w_poly.data(), h->w_poly.size()))) return rc_;
  if ((rc_ = upload(h, S.bins, h->w_bins.data(), h->w_bins.size()))) return rc_;
  if ((rc_ = upload(h, S.rcells, rc.data(), rc.size()))) return rc_;
  if ((rc_ = upload(h, S.prob, h->w_prob.data(), h->w_prob.size()))) return rc_;
  if ((rc_ = upload(h, S.topn, topn.data(), topn.size()))) return rc_;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  auvp::AstarWorldDev& W = S.W;
  W.n_obstacles = O; W.n_habitats = H; W.n_poly = V; W.n_bins = T; W.n_cells = C;
  W.ox = S.ox.as<double>(); W.oy = S.oy.as<double>(); W.ot = S.ot.as<double>(); W.hab = S.hab.as<double>();
  W.poly = S.poly.as<double>(); W.bins = S.bins.as<double>(); W.rcells = S.rcells.as<double>();
  W.prob = S.prob.as<double>(); W.topn = S.topn.as<double>();
  W.g_ncol = g_ncol; W.g_nrow = g_nrow;
  W.gx0 = W.gx1 = W.gy0 = W.gy1 = nullptr;
  W.g_inv_dx = W.g_inv_dy = 0.0;
  W.g_x1_0 = W.g_y1_0 = 0.0;
  if (g_ncol > 0) {
    W.gx0 = S.gridtab.as<double>(); W.gx1 = W.gx0 + g_ncol; W.gy0 = W.gx1 + g_ncol; W.gy1 = W.gy0 + g_nrow;
    const double* X1 = gtab.data() + g_ncol; const double* Y1 = gtab.data() + 2 * (size_t)g_ncol + g_nrow;
    W.g_x1_0 = X1[0]; W.g_y1_0 = Y1[0];
    W.g_inv_dx = g_ncol > 1 ? (g_ncol - 1) / (X1[g_ncol - 1] - X1[0]) : 0.0;
    W.g_inv_dy = g_nrow > 1 ? (g_nrow - 1) / (Y1[g_nrow - 1] - Y1[0]) : 0.0;
  }
  W.cx = V > 0 ? sx / (3.0 * a2) : 0.0;
  W.cy = V > 0 ? sy / (3.0 * a2) : 0.0;
  S.world_version = h->world_version;
  return AUVP_OK;
}

}  // namespace

extern "C" {

int auvp_astar_batch(auvp_handle* h, int32_t E, const double* starts, const double* goals, const double* limits,
                     const auvp_astar_params* p, int32_t flags) {
  if (!h) return AUVP_ERR_ARG;
  if (!h->have_world) return fail(h, AUVP_ERR_STATE, "auvp_world_set not called");
  if (E <= 0 || !starts || !p || p->variant < 0 || p->variant > 3) return fail(h, AUVP_ERR_ARG, "bad arguments");
  if (p->variant <= 1 && !goals) return fail(h, AUVP_ERR_ARG, "goals required for astar / astar_real");
  if (p->variant >= 2 && !limits) return fail(h, AUVP_ERR_ARG, "pathLenLimit required for the fixLen variants");
  if (p->variant >= 1 && h->w_poly.size() < 6) return fail(h, AUVP_ERR_ARG, "boundary polygon needs >= 3 vertices");
  if (p->cap_nodes < 16) return fail(h, AUVP_ERR_ARG, "cap_nodes too small");
  HIPCHK(h, hipSetDevice(h->device));
  AstarState& S = *astar_of(h);
  S.ready = false;
  int rc;
  if (S.world_version != h->world_version && (rc = astar_build_world(h, S))) return rc;
  auvp::AstarParamsDev& P = S.P;
  P.variant = p->variant; P.cap_nodes = p->cap_nodes; P.flags = flags;
  if (h->opt_on(OPT_ASTAR_NO_LIST)) P.flags |= AUVP_KFLAG_ASTAR_NO_LIST;
  P.cap_exp = (flags & AUVP_FLAG_ITER_LOG) ? p->cap_nodes : 0;
  for (int i = 0; i < 4; i++) { P.box[i] = p->box[i]; P.w[i] = p->w[i]; }
  P.velocity = p->velocity;
  P.vx = p->variant == 2 ? 550 : 600;  // np.zeros([550, 600]) (astar_fixLen.py:51) / [600, 600] (astar_fixLenSOG.py:117)
  P.vy = 600;
  auvp::AstarBuffers& B = S.B;
  if ((rc = upload(h, S.start, starts, (size_t)E * 2))) return rc;
  B.start = S.start.as<double>();
  B.goal = nullptr; B.limit = nullptr;
  if (p->variant <= 1) { if ((rc = upload(h, S.goal, goals, (size_t)E * 2))) return rc; B.goal = S.goal.as<double>(); }
  else { if ((rc = upload(h, S.limit, limits, (size_t)E))) return rc; B.limit = S.limit.as<double>(); }
  const size_t cn = (size_t)E * p->cap_nodes;
  HIPCHK(h, S.nodes.reserve(cn * 9 * sizeof(double)));
  HIPCHK(h, S.node_i.reserve(cn * sizeof(int32_t)));
  HIPCHK(h, S.summary.reserve((size_t)E * sizeof(auvp::AstarSummary)));
  const int H = S.W.n_habitats;
  HIPCHK(h, S.hab_left.reserve((size_t)E * (H > 0 ? H : 1) * sizeof(int32_t)));
  B.nodes = S.nodes.as<double>(); B.node_i = S.node_i.as<int32_t>(); B.summary = S.summary.as<auvp::AstarSummary>();
  B.hab_left = S.hab_left.as<int32_t>();
  B.cellinfo = nullptr;
  HIPCHK(h, hipEventRecord(h->ev0, h->stream));  // the batch time includes whatever reset the batch needs
  if (p->variant >= 2) {
    if (p->variant == 3 && S.W.n_cells >= 65535) return fail(h, AUVP_ERR_ARG, "n_cells %d does not fit the 16-bit cell-key field", S.W.n_cells);
    const size_t entries = (size_t)E * P.vx * P.vy, vb = entries * sizeof(uint32_t);
    HIPCHK(h, S.cellinfo.reserve(vb));
    if (flags & AUVP_FLAG_KEEP_VISITED) {
      // the reference keeps self.visited_nodes across astar() calls on one solver object
      // (astar_fixLen.py:51, astar_fixLenSOG.py:117): the caller uploaded it with auvp_astar_set_visited,
      // tagged with the current epoch
      if (S.visited_set_for != (long long)entries) return fail(h, AUVP_ERR_STATE, "auvp_astar_set_visited not called for this batch shape");
    } else {
      // a fresh visited array = a new epoch; the words are only cleared when the buffer is new or the tag wraps
      if (S.cellinfo_clean_cap != S.cellinfo.cap || S.epoch >= 255u) {
        HIPCHK(h, hipMemsetAsync(S.cellinfo.p, 0, S.cellinfo.cap, h->stream));
        S.cellinfo_clean_cap = S.cellinfo.cap;
        S.epoch = 0;
      }
      S.epoch++;
    }
    S.visited_set_for = -1;
    P.epoch = S.epoch;
    B.cellinfo = S.cellinfo.as<uint32_t>();
  }
  B.exp_log = nullptr;
  if (P.cap_exp) {
    HIPCHK(h, S.exp_log.reserve((size_t)E * P.cap_exp * 8 * sizeof(double)));
    B.exp_log = S.exp_log.as<double>();
  }
  B.pipe_fail = h->pipe_fail_dev;
  h->pipe_clear();
  h->pipe_fallback_last = 0;
  const int grid = (E + auvp::ASTAR_WAVES - 1) / auvp::ASTAR_WAVES;
  bool pair = false;
  switch (P.variant) {
    case 0: hipLaunchKernelGGL(auvp::astar_kernel<0>, dim3(grid), dim3(auvp::ASTAR_WAVES * 64), 0, h->stream, S.W, P, B, (int)E); break;
    case 1: hipLaunchKernelGGL(auvp::astar_kernel<1>, dim3(grid), dim3(auvp::ASTAR_WAVES * 64), 0, h->stream, S.W, P, B, (int)E); break;
    case 2: hipLaunchKernelGGL(auvp::astar_kernel<2>, dim3(grid), dim3(auvp::ASTAR_WAVES * 64), 0, h->stream, S.W, P, B, (int)E); break;
    default: {
      // latency batches on a product grid: a second wavefront per instance (astar_kernel.h, PAIR); option ASTAR_PAIR = 0 / 1
      // forces the choice.  Not with a visited array the caller carries over (KEEP_VISITED): a batch that had to be repeated on
      // the one-wavefront kernel (pipeline fallback, below) could not get it back.
      int n_cu = 256;
      (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, h->device);
      pair = S.W.g_ncol > 0 && !(flags & AUVP_FLAG_KEEP_VISITED) && h->opt_flag(OPT_ASTAR_PAIR, E <= (size_t)8 * (size_t)(n_cu > 0 ? n_cu : 256));
      if (pair) hipLaunchKernelGGL((auvp::astar_kernel<3, true>), dim3(grid), dim3(auvp::ASTAR_WAVES * 128), 0, h->stream, S.W, P, B, (int)E);
      else hipLaunchKernelGGL(auvp::astar_kernel<3>, dim3(grid), dim3(auvp::ASTAR_WAVES * 64), 0, h->stream, S.W, P, B, (int)E);
      break;
    }
  }
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipEventRecord(h->ev1, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (pair && h->pipe_failed()) {
    // the searching wavefront of some instance stopped waiting for its partner (AUVP_ERR_PIPELINE): the batch is searched again
    // by the one-wavefront kernel, on a fresh epoch of the visited words (same results: tests/test_gpu_astar.py)
    std::vector<auvp::AstarSummary> sm((size_t)E);
    HIPCHK(h, hipMemcpy(sm.data(), B.summary, sm.size() * sizeof(auvp::AstarSummary), hipMemcpyDeviceToHost));
    int n = 0;
    for (const auvp::AstarSummary& r : sm) n += r.status == AUVP_ERR_PIPELINE ? 1 : 0;
    h->pipe_clear();
    if (n > 0 && h->opt_flag(OPT_PIPE_FALLBACK, true)) {
      h->pipe_fallback_last = n;
      h->pipe_fallback_total += n;
      if (S.epoch >= 255u) {
        HIPCHK(h, hipMemsetAsync(S.cellinfo.p, 0, S.cellinfo.cap, h->stream));
        S.epoch = 0;
      }
      S.epoch++;
      P.epoch = S.epoch;
      pair = false;
      hipLaunchKernelGGL(auvp::astar_kernel<3>, dim3(grid), dim3(auvp::ASTAR_WAVES * 64), 0, h->stream, S.W, P, B, (int)E);
      HIPCHK(h, hipGetLastError());
      HIPCHK(h, hipEventRecord(h->ev1, h->stream));  // (the time the caller waited: both searches)
      HIPCHK(h, hipStreamSynchronize(h->stream));
    }
  }
  float ms = 0.f;
  HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
  h->last_ms = ms;
  h->last_grid = grid; h->last_block = auvp::ASTAR_WAVES * (pair ? 128 : 64); h->last_lds = 0;
  S.E = E;
  S.ready = true;
  return AUVP_OK;
}

int auvp_astar_summaries(auvp_handle* h, auvp_astar_summary* out) {
  if (!h || !out) return AUVP_ERR_ARG;
  AstarState& S = *astar_of(h);
  if (!S.ready) return fail(h, AUVP_ERR_STATE, "no A* batch");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpy(out, S.B.summary, (size_t)S.E * sizeof(auvp::AstarSummary), hipMemcpyDeviceToHost));
  return AUVP_OK;
}

int auvp_astar_paths(auvp_handle* h, const int64_t* offsets, double* path3, double* cost_list, double* node_path8,
                     double* smooth3) {
  if (!h || !offsets || !path3) return AUVP_ERR_ARG;
  AstarState& S = *astar_of(h);
  if (!S.ready) return fail(h, AUVP_ERR_STATE, "no A* batch");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t total = (size_t)offsets[S.E], n = std::max<size_t>(total, 1);
  int rc;
  if ((rc = upload(h, S.off, offsets, (size_t)S.E + 1))) return rc;
  HIPCHK(h, S.path.reserve(n * 3 * sizeof(double)));
  HIPCHK(h, S.cost.reserve(n * sizeof(double)));
  HIPCHK(h, S.npath.reserve(n * 8 * sizeof(double)));
  HIPCHK(h, S.smooth.reserve(n * 3 * sizeof(double)));
  hipLaunchKernelGGL(auvp::astar_path_kernel, dim3(S.E), dim3(64), 0, h->stream, S.W, S.P, S.B, S.off.as<int64_t>(),
                     S.path.as<double>(), S.cost.as<double>(), S.npath.as<double>(), S.smooth.as<double>(), S.E);
  HIPCHK(h, hipGetLastError());
  if (total) {
    HIPCHK(h, hipMemcpyAsync(path3, S.path.p, total * 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (cost_list) HIPCHK(h, hipMemcpyAsync(cost_list, S.cost.p, total * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (node_path8) HIPCHK(h, hipMemcpyAsync(node_path8, S.npath.p, total * 8 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (smooth3) HIPCHK(h, hipMemcpyAsync(smooth3, S.smooth.p, total * 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return AUVP_OK;
}

int auvp_astar_exp_log(auvp_handle* h, int32_t ep, double* out8) {
  if (!h || !out8) return AUVP_ERR_ARG;
  AstarState& S = *astar_of(h);
  if (!S.ready || ep < 0 || ep >= S.E || !S.B.exp_log) return fail(h, AUVP_ERR_STATE, "no expansion log");
  HIPCHK(h, hipSetDevice(h->device));
  auvp::AstarSummary s;
  HIPCHK(h, hipMemcpy(&s, S.B.summary + ep, sizeof s, hipMemcpyDeviceToHost));
  const size_t n = (size_t)std::min(s.n_expansions, S.P.cap_exp);
  if (n) HIPCHK(h, hipMemcpy(out8, S.B.exp_log + (size_t)ep * S.P.cap_exp * 8, n * 8 * sizeof(double), hipMemcpyDeviceToHost));
  return AUVP_OK;
}

// visited bitmap in / out (variants 2,3): [E][vx*vy] bytes, vx = 550 (fixLen) or 600 (SOG), vy = 600;
// set before auvp_astar_batch(..., AUVP_FLAG_KEEP_VISITED), get after any batch.  On the device the array is
// the epoch-tagged word array of astar_kernel.h; the byte view of the reference is converted here.
int auvp_astar_set_visited(auvp_handle* h, int32_t E, int32_t variant, const uint8_t* bitmap) {
  if (!h || !bitmap || E <= 0 || variant < 2 || variant > 3) return AUVP_ERR_ARG;
  AstarState& S = *astar_of(h);
  HIPCHK(h, hipSetDevice(h->device));
  const size_t entries = (size_t)E * (variant == 2 ? 550 : 600) * 600;
  HIPCHK(h, S.cellinfo.reserve(entries * sizeof(uint32_t)));
  if (S.cellinfo_clean_cap != S.cellinfo.cap || S.epoch >= 255u) {
    HIPCHK(h, hipMemsetAsync(S.cellinfo.p, 0, S.cellinfo.cap, h->stream));
    S.cellinfo_clean_cap = S.cellinfo.cap;
    S.epoch = 0;
  }
  S.epoch++;
  std::vector<uint32_t> words(entries);
  const uint32_t tag = S.epoch << 24;
  for (size_t i = 0; i < entries; i++) words[i] = bitmap[i] ? (tag | 0x10000u) : 0u;
  HIPCHK(h, hipMemcpyAsync(S.cellinfo.p, words.data(), entries * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  S.visited_set_for = (long long)entries;
  return AUVP_OK;
}

int auvp_astar_get_visited(auvp_handle* h, int32_t ep, uint8_t* bitmap) {
  if (!h || !bitmap) return AUVP_ERR_ARG;
  AstarState& S = *astar_of(h);
  if (!S.ready || ep < 0 || ep >= S.E || !S.B.cellinfo) return fail(h, AUVP_ERR_STATE, "no visited bitmap");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t one = (size_t)S.P.vx * S.P.vy;
  std::vector<uint32_t> words(one);
  HIPCHK(h, hipMemcpy(words.data(), S.B.cellinfo + (size_t)ep * one, one * sizeof(uint32_t), hipMemcpyDeviceToHost));
  const uint32_t tag = S.P.epoch << 24;
  for (size_t i = 0; i < one; i++) bitmap[i] = ((words[i] & 0xff000000u) == tag && (words[i] & 0x10000u)) ? 1 : 0;
  return AUVP_OK;
}

int auvp_astar_hab_left(auvp_handle* h, int32_t ep, int32_t* out) {
  if (!h || !out) return AUVP_ERR_ARG;
  AstarState& S = *astar_of(h);
  if (!S.ready || ep < 0 || ep >= S.E) return fail(h, AUVP_ERR_STATE, "bad instance");
  HIPCHK(h, hipSetDevice(h->device));
  const int H = S.W.n_habitats;
  if (H) HIPCHK(h, hipMemcpy(out, S.B.hab_left + (size_t)ep * H, (size_t)H * sizeof(int32_t), hipMemcpyDeviceToHost));
  return AUVP_OK;
}

}  // extern "C"
#endif
