// auvplan.hip -- C-ABI (include/auvplan.h) + host runtime of libauvplan.so for MI355X / gfx950.
//
// Host side of the hot path: owns the device copy of the world model and the per-episode tree
// storage in HBM, seeds the per-episode MT19937 states the way CPython's random.seed() does,
// launches the persistent expansion kernel (one wavefront per episode) on the handle's own HIP
// stream and times it with HIP events on that stream.  No CPU compute path exists here: if the
// device or the code object is unusable every entry point fails with AUVP_ERR_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/auvplan.h"
#include "auvp_seed.h"
#include "rrt_explore_kernel.h"
#include "rrt_rows_kernel.h"
#include "rrt_rows_stream_kernel.h"  // (the LDS plan; the kernel itself is launched from rows_kernels.hip)
#include "rrt_duo_kernel.h"
#include "rrt_trio_kernel.h"

using namespace auvp;

// rrt_rows_kernel lives in rows_kernels.hip (a translation unit with its own compiler flags); this is its launcher
extern "C" hipError_t auvpi_rrt_stream_launch(const auvp::RrtBuffers* B, int n_episodes, hipStream_t stream);
extern "C" hipError_t auvpi_rrt_rows_stream_launch(const auvp::WorldDev* W, const auvp::RrtParamsDev* P, const auvp::RrtBuffers* B, int n_episodes,
                                                   int grid, int block, int lds_max, int lds, hipStream_t stream);
extern "C" hipError_t auvpi_rrt_rows_launch(const auvp::WorldDev* W, const auvp::RrtParamsDev* P, const auvp::RrtBuffers* B, int n_episodes,
                                            int grid, int block, int lds_max, int lds, hipStream_t stream);

static_assert(sizeof(auvp_rrt_summary) == sizeof(RrtSummary), "summary layout");

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    release();
    hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
    if (e == hipSuccess) cap = bytes ? bytes : 16;
    return e;
  }
  void release() { if (p) { (void)hipFree(p); p = nullptr; cap = 0; } }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

}  // namespace

// Tuning / diagnostic options of a handle (auvp_set_option, include/auvplan.h).  Every kernel choice the host makes has a
// measured default ("auto": the option is unset); an option forces it.  The environment variable AUVP_<NAME> gives an
// option its initial value ONCE, when the handle is created -- no launch path reads the environment.
enum AuvpOpt {
  OPT_ROWS, OPT_DUO, OPT_TRIO, OPT_QUAD, OPT_TIGHT_CULL, OPT_NN_EXACT, OPT_LEAF_SWEEP_ALL, OPT_NO_HABITAT_GRID, OPT_RG_MAX_ENTRIES,
  OPT_NO_GRID_INDEX, OPT_PRRT_LAT, OPT_PRRT_PIPE, OPT_PRRT_OBST_LDS, OPT_PRRT_NEXT_LDS, OPT_PRRT_ROWS, OPT_ASTAR_NO_GRID,
  OPT_ASTAR_NO_LIST, OPT_ASTAR_PAIR, OPT_SOG_TILE, OPT_PIPE_FALLBACK, OPT_PRRT_PIPE_DRAW, OPT_PRRT_BUCKET_LDS, OPT_ROWS_STREAM, OPT_ROWS_STREAM_CAP, OPT_ROWS_STREAM_WAVES, OPT_COUNT
};
static const char* const AUVP_OPT_NAMES[OPT_COUNT] = {
  "ROWS", "DUO", "TRIO", "QUAD", "TIGHT_CULL", "NN_EXACT", "LEAF_SWEEP_ALL", "NO_HABITAT_GRID", "RG_MAX_ENTRIES",
  "NO_GRID_INDEX", "PRRT_LAT", "PRRT_PIPE", "PRRT_OBST_LDS", "PRRT_NEXT_LDS", "PRRT_ROWS", "ASTAR_NO_GRID",
  "ASTAR_NO_LIST", "ASTAR_PAIR", "SOG_TILE", "PIPE_FALLBACK", "PRRT_PIPE_DRAW", "PRRT_BUCKET_LDS", "ROWS_STREAM", "ROWS_STREAM_CAP", "ROWS_STREAM_WAVES"};

struct auvp_handle {
  bool opt_has[OPT_COUNT] = {};
  long long opt_val[OPT_COUNT] = {};
  // option K as a yes / no choice: its value when set, `dflt` (the measured heuristic) otherwise
  bool opt_flag(int k, bool dflt) const { return opt_has[k] ? opt_val[k] != 0 : dflt; }
  bool opt_on(int k) const { return opt_has[k] && opt_val[k] != 0; }
  long long opt_num(int k, long long dflt) const { return opt_has[k] ? opt_val[k] : dflt; }
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_mid = nullptr, ev_pre = nullptr;
  double last_expand_ms = 0.0, last_leaf_ms = 0.0, last_stream_ms = 0.0;
  long long last_stream_len = 0;  // numbers per episode the last pass generated ahead (0: none)
  // what complete passes of RRT.exploring drew, per parameter block (the last four): the most random() numbers of one episode and
  // the largest batch seen (the length of the next batch's pre-generated stream: rrt_run_pass)
  struct Drawn { RrtParamsDev P{}; long long most = 0; int E = 0; unsigned long long used = 0; bool valid = false; };
  Drawn drawn[4];
  unsigned long long drawn_clock = 0;
  Drawn* drawn_find(const RrtParamsDev& P) {
    for (Drawn& d : drawn)
      if (d.valid && memcmp(&d.P, &P, sizeof P) == 0) return &d;
    return nullptr;
  }
  int last_rows = 0;
  const char* last_rrt_kernel = "";
  std::string err;
  // world
  bool have_world = false;
  double obst_area = 0.0;  // area of the bounding box of the obstacle centres (0: fewer than two obstacles / degenerate)
  WorldDev W{};
  DevBuf d_ox, d_oy, d_ot, d_hab, d_habt, d_poly, d_bins, d_cells, d_prob, d_xoff, d_xitems, d_xdata, d_rgfirst, d_rgbp, d_rgoff,
      d_rgpm, d_rgid, d_sgx0, d_sgx1, d_sgy0, d_sgy1, d_sgcol, d_sgrow, d_osx, d_osy, d_ost, d_osr, d_osbox, d_hgmask;
  // rrt batch
  int E = 0;
  RrtParamsDev P{};
  RrtBuffers B{};
  int max_pts = 0;
  DevBuf d_nodes_f, d_nodes_i, d_points, d_bin_items, d_bin_count, d_mt, d_mtidx, d_seeds, d_init, d_summary, d_itlog_i, d_itlog_b,
      d_leaf_c, d_leaf_i, d_phase, d_leaf_stats, d_node_c, d_node_q, d_node_xy, d_tmp0, d_tmp1, d_tmp2, d_tmp3, d_tmp4, d_tmp5, d_stream;
  bool have_batch = false, prepared = false;
  double last_ms = 0.0;
  int last_grid = 0, last_block = 0, last_lds = 0;
  // planner families that live in their own headers keep their state behind an opaque pointer
  void* prrt = nullptr;
  void (*prrt_free)(void*) = nullptr;
  std::vector<double> w_obst, w_hab, w_poly, w_bins, w_cells, w_prob;  // host copy of the world
  unsigned world_version = 0;
  void* astar = nullptr;
  void (*astar_free)(void*) = nullptr;
  void* pf = nullptr;
  void (*pf_free)(void*) = nullptr;
  void* comm = nullptr;  // RCCL communicator state (gather_host.h)
  void (*comm_free)(void*) = nullptr;
  // Pipeline fallback.  The latency kernels (rrt_trio / rrt_duo, prrt_pipe, the paired astar_kernel) are speculative
  // pipelines of several wavefronts per episode whose waits are bounded; an episode whose wait runs out ends with
  // AUVP_ERR_PIPELINE and sets this host-mapped word.  The host then repeats the work on the one-wavefront kernel (same
  // results by construction: tests/test_gpu_duo_kernel.py etc.) -- a caller never sees the status.  Counters: episodes redone.
  int32_t* pipe_fail_host = nullptr;
  int32_t* pipe_fail_dev = nullptr;
  int pipe_fallback_last = 0;
  long long pipe_fallback_total = 0;
  bool pipe_failed() const { return pipe_fail_host && __atomic_load_n(pipe_fail_host, __ATOMIC_ACQUIRE) != 0; }
  void pipe_clear() { if (pipe_fail_host) __atomic_store_n(pipe_fail_host, 0, __ATOMIC_RELEASE); }
};

namespace {

int fail(auvp_handle* h, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (h) h->err = buf;
  return code;
}

#define HIPCHK(h, call)                                                                        \
  do {                                                                                         \
    hipError_t e__ = (call);                                                                   \
    if (e__ != hipSuccess) return fail(h, AUVP_ERR_HIP, "%s: %s", #call, hipGetErrorString(e__)); \
  } while (0)

// CPython random.seed(int) -> init_by_array over the 32-bit limbs of the seed
// (Modules/_randommodule.c; restated from the MT19937 reference algorithm)
void seed_mt(uint64_t seed, uint32_t* mt) {
  uint32_t key[2] = {(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32)};
  const int klen = key[1] ? 2 : 1;
  mt[0] = 19650218u;
  for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
  int i = 1, j = 0;
  for (int k = 624; k; k--) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
    i++; j++;
    if (i >= 624) { mt[0] = mt[623]; i = 1; }
    if (j >= klen) j = 0;
  }
  for (int k = 623; k; k--) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
    i++;
    if (i >= 624) { mt[0] = mt[623]; i = 1; }
  }
  mt[0] = 0x80000000u;
}

// largest double s with RN(sqrt(s)) <= R  (R < 0 -> -1: `d <= size` can never hold)
double sq_threshold(double R) {
  if (!(R >= 0.0)) return -1.0;
  double s = R * R;
  if (std::isinf(s)) return s;
  while (std::sqrt(s) > R) s = std::nextafter(s, -INFINITY);
  for (;;) {
    double n = std::nextafter(s, INFINITY);
    if (std::isinf(n) || std::sqrt(n) > R) break;
    s = n;
  }
  return s;
}

template <class T>
int upload(auvp_handle* h, DevBuf& b, const T* src, size_t n) {
  HIPCHK(h, b.reserve(n * sizeof(T)));
  if (n) HIPCHK(h, hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, h->stream));
  return AUVP_OK;
}


// habitat mask grid (auvp_types.h): 32 x 32 cells over the box of the habitats' cull squares; a habitat is entered into
// every cell its square touches, one cell of margin on each side against rounding in the cell arithmetic
int build_habitat_grid(auvp_handle* h, const double* habitats, int H) {
  WorldDev& W = h->W;
  W.hg_n = 0; W._pad_hg = 0; W.hg_x0 = W.hg_y0 = W.hg_inv_w = W.hg_inv_h = 0.0;
  W.hg_mask = nullptr;
  if (H <= 0 || H > 64 || h->opt_on(OPT_NO_HABITAT_GRID)) return AUVP_OK;
  const int G = 32;
  double x0 = INFINITY, y0 = INFINITY, x1 = -INFINITY, y1 = -INFINITY;
  std::vector<double> rr(H);
  for (int i = 0; i < H; i++) {
    const double t = sq_threshold(habitats[3 * i + 2]);
    rr[i] = t >= 0.0 ? std::sqrt(t) * (1.0 + 0x1p-30) + 0x1p-40 : -1.0;
    const double hx = habitats[3 * i], hy = habitats[3 * i + 1];
    if (!std::isfinite(hx) || !std::isfinite(hy) || !std::isfinite(rr[i])) return AUVP_OK;  // odd input: keep the plain scan
    if (rr[i] < 0.0) continue;  // can never contain a point
    x0 = std::min(x0, hx - rr[i]); x1 = std::max(x1, hx + rr[i]);
    y0 = std::min(y0, hy - rr[i]); y1 = std::max(y1, hy + rr[i]);
  }
  std::vector<unsigned long long> mask((size_t)G * G, 0ull);
  if (x1 >= x0 && y1 >= y0) {
    // grow the box a little so that every square lies strictly inside it
    const double mx = 1e-6 * (std::fabs(x0) + std::fabs(x1) + 1.0), my = 1e-6 * (std::fabs(y0) + std::fabs(y1) + 1.0);
    x0 -= mx; x1 += mx; y0 -= my; y1 += my;
    const double iw = (double)G / (x1 - x0), ih = (double)G / (y1 - y0);
    for (int i = 0; i < H; i++) {
      if (rr[i] < 0.0) continue;
      const double hx = habitats[3 * i], hy = habitats[3 * i + 1];
      int cx0 = (int)std::floor((hx - rr[i] - x0) * iw) - 1, cx1 = (int)std::floor((hx + rr[i] - x0) * iw) + 1;
      int cy0 = (int)std::floor((hy - rr[i] - y0) * ih) - 1, cy1 = (int)std::floor((hy + rr[i] - y0) * ih) + 1;
      cx0 = std::max(cx0, 0); cy0 = std::max(cy0, 0); cx1 = std::min(cx1, G - 1); cy1 = std::min(cy1, G - 1);
      for (int cy = cy0; cy <= cy1; cy++)
        for (int cx = cx0; cx <= cx1; cx++) mask[(size_t)cy * G + cx] |= 1ull << i;
    }
    W.hg_x0 = x0; W.hg_y0 = y0; W.hg_inv_w = iw; W.hg_inv_h = ih;
  } else {
    W.hg_inv_w = W.hg_inv_h = 0.0;  // no habitat can contain anything: every cell empty (floor(nan/inf) guards below)
    W.hg_x0 = W.hg_y0 = 0.0;
  }
  int rc = upload(h, h->d_hgmask, mask.data(), mask.size());
  if (rc != AUVP_OK) return rc;
  if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(h, AUVP_ERR_HIP, "habitat grid upload");
  W.hg_mask = h->d_hgmask.as<unsigned long long>();
  W.hg_n = G;
  return AUVP_OK;
}

}  // namespace

extern "C" {

const char* auvp_version(void) { return "auvplan 0.1 (gfx950)"; }

int auvp_create(int device, auvp_handle** out) {
  if (!out) return AUVP_ERR_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return AUVP_ERR_HIP;
  auvp_handle* h = new auvp_handle();
  h->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess || hipEventCreate(&h->ev_mid) != hipSuccess ||
      hipEventCreate(&h->ev_pre) != hipSuccess) {
    delete h;
    return AUVP_ERR_HIP;
  }
  // options: initial values from AUVP_<NAME>, read here and nowhere else
  for (int k = 0; k < OPT_COUNT; k++) {
    const std::string name = std::string("AUVP_") + AUVP_OPT_NAMES[k];
    if (const char* e = getenv(name.c_str())) { h->opt_has[k] = true; h->opt_val[k] = atoll(e); }
  }
  if (hipHostMalloc(reinterpret_cast<void**>(&h->pipe_fail_host), 64, hipHostMallocMapped) == hipSuccess) {
    h->pipe_fail_host[0] = 0;
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&h->pipe_fail_dev), h->pipe_fail_host, 0) != hipSuccess) h->pipe_fail_dev = nullptr;
  } else {
    h->pipe_fail_host = nullptr;
    (void)hipGetLastError();
  }
#ifdef AUVP_PIPE_DIAG
  {
    // diagnostic build only (libauvplan_diag.so): spin limit and delay injection of the speculative pipelines (auvp_wave.h)
    int cfg[8] = {0, 0, 1, 0, 0, 0, 0, 0};
    if (const char* e = getenv("AUVP_DIAG_SPIN")) cfg[0] = atoi(e);
    if (const char* e = getenv("AUVP_DIAG_JITTER")) (void)sscanf(e, "%d,%d,%d,%d,%d", &cfg[1], &cfg[2], &cfg[3], &cfg[4], &cfg[5]);
    if (hipMemcpyToSymbol(HIP_SYMBOL(auvp_diag_cfg), cfg, sizeof cfg) != hipSuccess) { auvp_destroy(h); return AUVP_ERR_HIP; }
  }
#endif
  *out = h;
  return AUVP_OK;
}

int auvp_set_option(auvp_handle* h, const char* name, int64_t value) {
  if (!h || !name) return AUVP_ERR_ARG;
  for (int k = 0; k < OPT_COUNT; k++)
    if (!strcmp(name, AUVP_OPT_NAMES[k])) { h->opt_has[k] = true; h->opt_val[k] = value; return AUVP_OK; }
  return fail(h, AUVP_ERR_ARG, "unknown option %s", name);
}

int auvp_unset_option(auvp_handle* h, const char* name) {
  if (!h || !name) return AUVP_ERR_ARG;
  for (int k = 0; k < OPT_COUNT; k++)
    if (!strcmp(name, AUVP_OPT_NAMES[k])) { h->opt_has[k] = false; h->opt_val[k] = 0; return AUVP_OK; }
  return fail(h, AUVP_ERR_ARG, "unknown option %s", name);
}

int auvp_get_option(auvp_handle* h, const char* name, int32_t* is_set, int64_t* value) {
  if (!h || !name) return AUVP_ERR_ARG;
  for (int k = 0; k < OPT_COUNT; k++)
    if (!strcmp(name, AUVP_OPT_NAMES[k])) {
      if (is_set) *is_set = h->opt_has[k] ? 1 : 0;
      if (value) *value = h->opt_val[k];
      return AUVP_OK;
    }
  return fail(h, AUVP_ERR_ARG, "unknown option %s", name);
}

int auvp_pipeline_fallbacks(auvp_handle* h, int32_t* last, int64_t* total) {
  if (!h) return AUVP_ERR_ARG;
  if (last) *last = h->pipe_fallback_last;
  if (total) *total = h->pipe_fallback_total;
  return AUVP_OK;
}

void auvp_destroy(auvp_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->prrt && h->prrt_free) h->prrt_free(h->prrt);
  if (h->astar && h->astar_free) h->astar_free(h->astar);
  if (h->pf && h->pf_free) h->pf_free(h->pf);
  if (h->comm && h->comm_free) h->comm_free(h->comm);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->ev_mid) (void)hipEventDestroy(h->ev_mid);
  if (h->ev_pre) (void)hipEventDestroy(h->ev_pre);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  if (h->pipe_fail_host) (void)hipHostFree(h->pipe_fail_host);
  delete h;
}

const char* auvp_last_error(auvp_handle* h) { return h ? h->err.c_str() : "null handle"; }

int auvp_world_set(auvp_handle* h, const double* obstacles, int32_t O, const double* habitats, int32_t H,
                   const double* polygon, int32_t V, const double* bins, int32_t T, const double* cells, int32_t C,
                   const double* prob) {
  if (!h) return AUVP_ERR_ARG;
  if (O < 0 || H < 0 || V < 0 || T < 0 || C < 0) return fail(h, AUVP_ERR_ARG, "negative size");
  if (H > RRT_MAX_HAB) return fail(h, AUVP_ERR_ARG, "n_habitats %d > %d", H, RRT_MAX_HAB);
  if (V > RRT_MAX_POLY) return fail(h, AUVP_ERR_ARG, "n_poly %d > %d", V, RRT_MAX_POLY);
  // a boundary needs at least three vertices (shapely's Polygon constructor raises for fewer); V == 0 means
  // "no boundary" and is accepted for the world-only users (cost probe, A* with its own box)
  if (V == 1 || V == 2) return fail(h, AUVP_ERR_ARG, "boundary polygon with %d vertices (need >= 3, or 0 for none)", V);
  if (T > RRT_MAX_BINS) return fail(h, AUVP_ERR_ARG, "n_bins %d > %d", T, RRT_MAX_BINS);
  HIPCHK(h, hipSetDevice(h->device));
  std::vector<double> ox(O), oy(O), ot(O);
  double run = -INFINITY;
  for (int i = O - 1; i >= 0; i--) {  // suffix max of the radii, list order
    double r = obstacles[3 * i + 2];
    if (r > run) run = r;
    ox[i] = obstacles[3 * i];
    oy[i] = obstacles[3 * i + 1];
    ot[i] = sq_threshold(run);
  }
  // x-bucket index over the cells
  std::vector<int32_t> xoff, xitems;
  std::vector<double> xdata;
  double X0 = 0.0, inv_w = 0.0;
  int NB = 0;
  if (C > 0) {
    double lo = INFINITY, hi = -INFINITY, wmin = INFINITY;
    for (int c = 0; c < C; c++) {
      double a = cells[4 * c], b = std::min(cells[4 * c + 2], cells[4 * c + 3]);
      double wdt = cells[4 * c + 2] - cells[4 * c];
      if (wdt > 0 && wdt < wmin) wmin = wdt;
      if (b < a) continue;
      lo = std::min(lo, a);
      hi = std::max(hi, b);
    }
    if (!(hi >= lo)) { lo = 0.0; hi = 1.0; }
    double span = hi - lo;
    NB = 1;
    if (span > 0 && std::isfinite(wmin)) NB = (int)std::min(8192.0, std::max(1.0, std::ceil(span / wmin)));
    X0 = lo;
    inv_w = span > 0 ? (double)NB / span : 0.0;
    std::vector<std::vector<int32_t>> lists(NB);
    for (int c = 0; c < C; c++) {
      double a = cells[4 * c], b = std::min(cells[4 * c + 2], cells[4 * c + 3]);
      if (b < a) continue;
      int b0 = (int)std::floor((a - X0) * inv_w) - 1, b1 = (int)std::floor((b - X0) * inv_w) + 1;
      b0 = std::max(0, std::min(NB - 1, b0));
      b1 = std::max(0, std::min(NB - 1, b1));
      for (int k = b0; k <= b1; k++) lists[k].push_back(c);
    }
    xoff.resize(NB + 1);
    xoff[0] = 0;
    for (int k = 0; k < NB; k++) {
      xoff[k + 1] = xoff[k] + (int32_t)lists[k].size();
      xitems.insert(xitems.end(), lists[k].begin(), lists[k].end());
    }
    xdata.resize(xitems.size() * 4);
    for (int k = 0; k < NB; k++) {
      double suf = INFINITY;
      for (int i = xoff[k + 1] - 1; i >= xoff[k]; i--) {
        const double* cb = cells + 4 * (size_t)xitems[i];
        suf = std::min(suf, cb[1]);
        xdata[4 * (size_t)i] = cb[0];
        xdata[4 * (size_t)i + 1] = std::min(cb[2], cb[3]);
        xdata[4 * (size_t)i + 2] = cb[1];
        xdata[4 * (size_t)i + 3] = suf;
      }
    }
  } else {
    xoff.assign(2, 0);
  }
  // region index (auvp_types.h): breakpoints, per-region candidate lists with running min of miny
  std::vector<double> bp, rpm;
  std::vector<int32_t> rfirst(NB + 1, 0), roff(2, 0), rid;
  int rg_enabled = 0;
  if (C > 0) {
    bp.reserve((size_t)2 * C);
    for (int c = 0; c < C; c++) {
      const double a = cells[4 * c], b = std::min(cells[4 * c + 2], cells[4 * c + 3]);
      if (b < a || a != a || b != b) continue;
      bp.push_back(a); bp.push_back(b);
    }
    std::sort(bp.begin(), bp.end());
    bp.erase(std::unique(bp.begin(), bp.end()), bp.end());
    const size_t m = bp.size();
    auto bucket_of = [&](double x) {
      const double fb = std::floor((x - X0) * inv_w);
      return fb < 0.0 ? 0 : (fb >= (double)NB ? NB - 1 : (int)fb);
    };
    for (size_t i = 0; i < m; i++) rfirst[bucket_of(bp[i]) + 1]++;
    for (int k = 0; k < NB; k++) rfirst[k + 1] += rfirst[k];
    const size_t R = 2 * m + 1;
    std::vector<int64_t> cnt(R + 1, 0);
    std::vector<int32_t> r0(C, 0), r1(C, -1);
    for (int c = 0; c < C; c++) {
      const double a = cells[4 * c], b = std::min(cells[4 * c + 2], cells[4 * c + 3]);
      if (b < a || a != a || b != b) continue;
      r0[c] = 2 * (int32_t)(std::lower_bound(bp.begin(), bp.end(), a) - bp.begin()) + 1;
      r1[c] = 2 * (int32_t)(std::lower_bound(bp.begin(), bp.end(), b) - bp.begin()) + 1;
      cnt[r0[c]]++; cnt[r1[c] + 1]--;
    }
    int64_t total = 0, run = 0;
    for (size_t r = 0; r < R; r++) { run += cnt[r]; total += run; }
    int64_t budget = 32ll << 20;
    budget = h->opt_num(OPT_RG_MAX_ENTRIES, budget);
    if (total <= budget && R + 1 < (size_t)INT32_MAX) {
      rg_enabled = 1;
      roff.assign(R + 1, 0);
      run = 0;
      for (size_t r = 0; r < R; r++) { run += cnt[r]; roff[r + 1] = roff[r] + (int32_t)run; }
      rpm.resize((size_t)total); rid.resize((size_t)total);
      std::vector<int32_t> fill(roff.begin(), roff.end() - 1);
      for (int c = 0; c < C; c++)
        for (int32_t r = r0[c]; r <= r1[c]; r++) {
          const int32_t k = fill[r]++;
          rid[k] = c;
          rpm[k] = (k > roff[r]) ? std::min(rpm[k - 1], cells[4 * c + 1]) : cells[4 * c + 1];
        }
    } else {
      bp.clear();
      std::fill(rfirst.begin(), rfirst.end(), 0);
    }
  }
  // separable-grid index (auvp_types.h): cell_list row-major over non-decreasing X0/X1 (columns) and Y0/Y1 (rows)
  std::vector<double> sgx0, sgx1, sgy0, sgy1;
  int sg_ncol = 0, sg_nrow = 0;
  if (C > 0) {
    int nc = 1;
    while (nc < C && cells[4 * (size_t)nc + 1] == cells[1] && cells[4 * (size_t)nc + 3] == cells[3]) nc++;
    if (C % nc == 0) {
      const int nr = C / nc;
      bool ok = true;
      sgx0.resize(nc); sgx1.resize(nc); sgy0.resize(nr); sgy1.resize(nr);
      for (int c = 0; c < nc; c++) { sgx0[c] = cells[4 * (size_t)c]; sgx1[c] = cells[4 * (size_t)c + 2]; }
      for (int r = 0; r < nr; r++) { sgy0[r] = cells[4 * (size_t)r * nc + 1]; sgy1[r] = cells[4 * (size_t)r * nc + 3]; }
      for (int r = 0; r < nr && ok; r++)
        for (int c = 0; c < nc; c++) {
          const double* cb = cells + 4 * ((size_t)r * nc + c);
          if (!(cb[0] == sgx0[c] && cb[1] == sgy0[r] && cb[2] == sgx1[c] && cb[3] == sgy1[r])) { ok = false; break; }
        }
      for (int c = 1; c < nc && ok; c++) ok = sgx0[c] >= sgx0[c - 1] && sgx1[c] >= sgx1[c - 1];
      for (int r = 1; r < nr && ok; r++) ok = sgy0[r] >= sgy0[r - 1] && sgy1[r] >= sgy1[r - 1];
      for (int c = 0; c < nc && ok; c++) ok = std::isfinite(sgx0[c]) && std::isfinite(sgx1[c]);
      for (int r = 0; r < nr && ok; r++) ok = std::isfinite(sgy0[r]) && std::isfinite(sgy1[r]);
      if (ok && !h->opt_on(OPT_NO_GRID_INDEX)) { sg_ncol = nc; sg_nrow = nr; }
    }
  }
  double prob_absmax = 0.0;
  bool any_pos = false, any_neg = false, any_nan = false;
  for (size_t i = 0; i < (size_t)T * C; i++) {
    const double a = std::fabs(prob[i]);
    if (a > prob_absmax || a != a) prob_absmax = a;  // a nan poisons the bound: every leaf is then re-summed exactly
    any_pos |= prob[i] > 0; any_neg |= prob[i] < 0; any_nan |= prob[i] != prob[i];
  }
  int rc;
  if ((rc = upload(h, h->d_rgfirst, rfirst.data(), rfirst.size()))) return rc;
  if ((rc = upload(h, h->d_rgbp, bp.data(), bp.size()))) return rc;
  if ((rc = upload(h, h->d_rgoff, roff.data(), roff.size()))) return rc;
  if ((rc = upload(h, h->d_rgpm, rpm.data(), rpm.size()))) return rc;
  if ((rc = upload(h, h->d_rgid, rid.data(), rid.size()))) return rc;
  if ((rc = upload(h, h->d_sgx0, sgx0.data(), (size_t)sg_ncol))) return rc;
  if ((rc = upload(h, h->d_sgx1, sgx1.data(), (size_t)sg_ncol))) return rc;
  if ((rc = upload(h, h->d_sgy0, sgy0.data(), (size_t)sg_nrow))) return rc;
  if ((rc = upload(h, h->d_sgy1, sgy1.data(), (size_t)sg_nrow))) return rc;
  {
    std::vector<double> col((size_t)sg_ncol * 4, 0.0), row((size_t)sg_nrow * 4, 0.0);
    for (int c = 0; c < sg_ncol; c++) { col[4 * (size_t)c] = c ? sgx1[c - 1] : -INFINITY; col[4 * (size_t)c + 1] = sgx1[c]; col[4 * (size_t)c + 2] = sgx0[c]; }
    for (int r = 0; r < sg_nrow; r++) { row[4 * (size_t)r] = r ? sgy1[r - 1] : -INFINITY; row[4 * (size_t)r + 1] = sgy1[r]; row[4 * (size_t)r + 2] = sgy0[r]; }
    if ((rc = upload(h, h->d_sgcol, col.data(), col.size()))) return rc;
    if ((rc = upload(h, h->d_sgrow, row.data(), row.size()))) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  {
    // spatially sorted obstacle tile + slot boxes for rrt_rows_kernel (auvp_types.h)
    const int NS = 256;
    std::vector<double> sx(NS, 0.0), sy(NS, 0.0), st(NS, -1.0), box(16 * 4);
    std::vector<float> sr(NS, -INFINITY);
    {
      double x0 = INFINITY, y0 = INFINITY, x1 = -INFINITY, y1 = -INFINITY;
      for (int i = 0; i < O; i++) { x0 = std::min(x0, ox[i]); x1 = std::max(x1, ox[i]); y0 = std::min(y0, oy[i]); y1 = std::max(y1, oy[i]); }
      h->obst_area = (O > 1 && std::isfinite((x1 - x0) * (y1 - y0))) ? (x1 - x0) * (y1 - y0) : 0.0;
    }
    if (O <= NS) {
      double x0 = INFINITY, y0 = INFINITY, x1 = -INFINITY, y1 = -INFINITY;
      for (int i = 0; i < O; i++) { x0 = std::min(x0, ox[i]); x1 = std::max(x1, ox[i]); y0 = std::min(y0, oy[i]); y1 = std::max(y1, oy[i]); }
      auto spread = [](uint32_t v) { uint64_t x = v & 0xffff; x = (x | (x << 8)) & 0x00ff00ff; x = (x | (x << 4)) & 0x0f0f0f0f;
                                     x = (x | (x << 2)) & 0x33333333; x = (x | (x << 1)) & 0x55555555; return x; };
      std::vector<std::pair<uint64_t, int>> key(O);
      for (int i = 0; i < O; i++) {
        const double fx = x1 > x0 ? (ox[i] - x0) / (x1 - x0) : 0.0, fy = y1 > y0 ? (oy[i] - y0) / (y1 - y0) : 0.0;
        const uint32_t qx = (uint32_t)std::min(65535.0, std::max(0.0, fx * 65535.0)), qy = (uint32_t)std::min(65535.0, std::max(0.0, fy * 65535.0));
        key[i] = {(std::isfinite(fx) && std::isfinite(fy)) ? (spread(qx) | (spread(qy) << 1)) : 0, i};
      }
      std::sort(key.begin(), key.end());
      for (int k = 0; k < O; k++) {
        const int i = key[k].second;
        sx[k] = ox[i]; sy[k] = oy[i]; st[k] = ot[i];
        const double rd = ot[i] >= 0.0 ? std::sqrt(ot[i]) * (1.0 + 0x1p-30) + 0x1p-40 : -INFINITY;
        float rf = (float)rd;
        if ((double)rf < rd) rf = std::nextafter(rf, INFINITY);
        sr[k] = rf;
      }
    }
    for (int s_ = 0; s_ < 16; s_++) {
      double b0 = INFINITY, b1 = INFINITY, b2 = -INFINITY, b3 = -INFINITY;
      for (int k = 16 * s_; k < 16 * s_ + 16; k++) {
        if (!(sr[k] >= 0.0f) && !std::isnan(sx[k])) continue;  // padding / an obstacle that can never collide
        const double r = (double)sr[k];
        if (std::isnan(sx[k]) || std::isnan(sy[k]) || std::isnan(r)) { b0 = b1 = -INFINITY; b2 = b3 = INFINITY; break; }  // never culled
        b0 = std::min(b0, sx[k] - r); b1 = std::min(b1, sy[k] - r); b2 = std::max(b2, sx[k] + r); b3 = std::max(b3, sy[k] + r);
      }
      box[4 * s_] = b0; box[4 * s_ + 1] = b1; box[4 * s_ + 2] = b2; box[4 * s_ + 3] = b3;
    }
    int rc2;
    if ((rc2 = upload(h, h->d_osx, sx.data(), sx.size()))) return rc2;
    if ((rc2 = upload(h, h->d_osy, sy.data(), sy.size()))) return rc2;
    if ((rc2 = upload(h, h->d_ost, st.data(), st.size()))) return rc2;
    if ((rc2 = upload(h, h->d_osr, sr.data(), sr.size()))) return rc2;
    if ((rc2 = upload(h, h->d_osbox, box.data(), box.size()))) return rc2;
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  if ((rc = upload(h, h->d_ox, ox.data(), O))) return rc;
  if ((rc = upload(h, h->d_oy, oy.data(), O))) return rc;
  if ((rc = upload(h, h->d_ot, ot.data(), O))) return rc;
  if ((rc = upload(h, h->d_hab, habitats, (size_t)H * 3))) return rc;
  {
    std::vector<double> ht(H);
    for (int i = 0; i < H; i++) ht[i] = sq_threshold(habitats[3 * i + 2]);
    if ((rc = upload(h, h->d_habt, ht.data(), (size_t)H))) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  if ((rc = upload(h, h->d_poly, polygon, (size_t)V * 2))) return rc;
  if ((rc = upload(h, h->d_bins, bins, (size_t)T * 2))) return rc;
  if ((rc = upload(h, h->d_cells, cells, (size_t)C * 4))) return rc;
  if ((rc = upload(h, h->d_prob, prob, (size_t)T * C))) return rc;
  if ((rc = upload(h, h->d_xoff, xoff.data(), xoff.size()))) return rc;
  if ((rc = upload(h, h->d_xitems, xitems.data(), xitems.size()))) return rc;
  if ((rc = upload(h, h->d_xdata, xdata.data(), xdata.size()))) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));  // host vectors go out of scope
  WorldDev& W = h->W;
  W.n_obstacles = O; W.n_habitats = H; W.n_poly = V; W.n_bins = T; W.n_cells = C; W.n_xbuckets = NB;
  W.ox = h->d_ox.as<double>(); W.oy = h->d_oy.as<double>(); W.ot = h->d_ot.as<double>();
  W.os_x = h->d_osx.as<double>(); W.os_y = h->d_osy.as<double>(); W.os_t = h->d_ost.as<double>(); W.os_r = h->d_osr.as<float>();
  W.os_box = h->d_osbox.as<double>();
  W.hab = h->d_hab.as<double>(); W.hab_t = h->d_habt.as<double>(); W.poly = h->d_poly.as<double>(); W.bins = h->d_bins.as<double>();
  W.cells = h->d_cells.as<double>(); W.prob = h->d_prob.as<double>();
  W.xb_off = h->d_xoff.as<int32_t>(); W.xb_items = h->d_xitems.as<int32_t>(); W.xb_data = h->d_xdata.as<double>();
  W.xb_x0 = X0; W.xb_inv_w = inv_w;
  W.rg_enabled = rg_enabled; W.n_rg_bp = (int32_t)bp.size();
  W.rg_first = h->d_rgfirst.as<int32_t>(); W.rg_bp = h->d_rgbp.as<double>(); W.rg_off = h->d_rgoff.as<int32_t>();
  W.rg_pm = h->d_rgpm.as<double>(); W.rg_id = h->d_rgid.as<int32_t>();
  W.sg_enabled = sg_ncol > 0 ? 1 : 0; W.sg_ncol = sg_ncol; W.sg_nrow = sg_nrow;
  W.sg_col = h->d_sgcol.as<double>(); W.sg_row = h->d_sgrow.as<double>();
  W.sg_x1_0 = sg_ncol > 0 ? sgx1[0] : 0.0; W.sg_y1_0 = sg_nrow > 0 ? sgy1[0] : 0.0;
  {
    bool sorted = T > 0;
    for (int t = 0; t < T && sorted; t++) sorted = std::isfinite(bins[2 * t]) && std::isfinite(bins[2 * t + 1]);
    for (int t = 1; t < T && sorted; t++) sorted = bins[2 * t] >= bins[2 * (t - 1)] && bins[2 * t + 1] >= bins[2 * (t - 1) + 1];
    W.bins_sorted = sorted ? 1 : 0;
    W.bins_t1_0 = T > 0 ? bins[1] : 0.0;
    W.bins_inv_len = (sorted && T > 1 && bins[2 * (T - 1) + 1] > bins[1]) ? (double)(T - 1) / (bins[2 * (T - 1) + 1] - bins[1]) : 0.0;
  }
  W.sg_x0 = h->d_sgx0.as<double>(); W.sg_x1 = h->d_sgx1.as<double>(); W.sg_y0 = h->d_sgy0.as<double>(); W.sg_y1 = h->d_sgy1.as<double>();
  W.sg_inv_dx = (sg_ncol > 1 && sgx1[sg_ncol - 1] > sgx1[0]) ? (double)(sg_ncol - 1) / (sgx1[sg_ncol - 1] - sgx1[0]) : 0.0;
  W.sg_inv_dy = (sg_nrow > 1 && sgy1[sg_nrow - 1] > sgy1[0]) ? (double)(sg_nrow - 1) / (sgy1[sg_nrow - 1] - sgy1[0]) : 0.0;
  W.prob_absmax = prob_absmax;
  W.prob_one_sign = (!any_nan && !(any_pos && any_neg)) ? 1 : 0;
  W._pad_prob = 0;
  if ((rc = build_habitat_grid(h, habitats, H))) return rc;
  double bb[4] = {INFINITY, INFINITY, -INFINITY, -INFINITY};
  for (int i = 0; i < V; i++) {
    bb[0] = std::min(bb[0], polygon[2 * i]); bb[1] = std::min(bb[1], polygon[2 * i + 1]);
    bb[2] = std::max(bb[2], polygon[2 * i]); bb[3] = std::max(bb[3], polygon[2 * i + 1]);
  }
  memcpy(W.bb, bb, sizeof bb);
  // axis-aligned rectangle given by its 4 corners: strict containment in it implies Point.within
  W.has_safe_box = 0;
  if (V == 4) {
    bool rect = true;
    for (int i = 0; i < 4 && rect; i++) {
      const double x0 = polygon[2 * i], y0 = polygon[2 * i + 1], x1 = polygon[2 * ((i + 1) % 4)], y1 = polygon[2 * ((i + 1) % 4) + 1];
      const bool horiz = (y0 == y1) && (x0 != x1), vert = (x0 == x1) && (y0 != y1);
      const double x2 = polygon[2 * ((i + 2) % 4)], y2 = polygon[2 * ((i + 2) % 4) + 1];
      const bool next_vert = (x1 == x2) && (y1 != y2), next_horiz = (y1 == y2) && (x1 != x2);
      rect = (horiz && next_vert) || (vert && next_horiz);
    }
    if (rect) { W.has_safe_box = 1; memcpy(W.safe_box, bb, sizeof bb); }
  }
  // host copies: the A* family derives its own device tables from the same world (astar_host.h)
  h->w_obst.assign(obstacles, obstacles + (size_t)O * 3);
  h->w_hab.assign(habitats, habitats + (size_t)H * 3);
  h->w_poly.assign(polygon, polygon + (size_t)V * 2);
  h->w_bins.assign(bins, bins + (size_t)T * 2);
  h->w_cells.assign(cells, cells + (size_t)C * 4);
  h->w_prob.assign(prob, prob + (size_t)T * C);
  h->world_version++;
  h->have_world = true;
  return AUVP_OK;
}

int auvp_world_set_habitats(auvp_handle* h, const double* habitats, int32_t H) {
  if (!h) return AUVP_ERR_ARG;
  if (!h->have_world) return fail(h, AUVP_ERR_STATE, "auvp_world_set not called");
  if (H < 0 || H > RRT_MAX_HAB) return fail(h, AUVP_ERR_ARG, "n_habitats %d outside 0..%d", H, RRT_MAX_HAB);
  HIPCHK(h, hipSetDevice(h->device));
  int rc;
  if ((rc = upload(h, h->d_hab, habitats, (size_t)H * 3))) return rc;
  std::vector<double> ht(H);
  for (int i = 0; i < H; i++) ht[i] = sq_threshold(habitats[3 * i + 2]);
  if ((rc = upload(h, h->d_habt, ht.data(), (size_t)H))) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->W.n_habitats = H;
  h->W.hab = h->d_hab.as<double>();
  h->W.hab_t = h->d_habt.as<double>();
  if ((rc = build_habitat_grid(h, habitats, H))) return rc;
  h->w_hab.assign(habitats, habitats + (size_t)H * 3);
  h->world_version++;
  return AUVP_OK;
}

int auvp_rrt_explore_batch(auvp_handle* h, int32_t E, const double* init, const uint64_t* seeds,
                           const auvp_rrt_params* p, int32_t flags) {
  int rc = auvp_rrt_prepare(h, E, init, seeds, p, flags);
  if (rc != AUVP_OK) return rc;
  return auvp_rrt_run(h);
}

static int rrt_prepare_impl(auvp_handle* h, int32_t E, const double* init, const uint64_t* seeds, const uint32_t* states,
                            const int32_t* state_index, const auvp_rrt_params* p, int32_t flags);

int auvp_rrt_prepare(auvp_handle* h, int32_t E, const double* init, const uint64_t* seeds,
                     const auvp_rrt_params* p, int32_t flags) {
  if (!seeds) return h ? fail(h, AUVP_ERR_ARG, "seeds is null") : AUVP_ERR_ARG;
  return rrt_prepare_impl(h, E, init, seeds, nullptr, nullptr, p, flags);
}

int auvp_rrt_prepare_states(auvp_handle* h, int32_t E, const double* init, const uint32_t* mt, const int32_t* mt_index,
                            const auvp_rrt_params* p, int32_t flags) {
  if (!mt || !mt_index) return h ? fail(h, AUVP_ERR_ARG, "mt state is null") : AUVP_ERR_ARG;
  return rrt_prepare_impl(h, E, init, nullptr, mt, mt_index, p, flags);
}

static int rrt_prepare_impl(auvp_handle* h, int32_t E, const double* init, const uint64_t* seeds, const uint32_t* states,
                            const int32_t* state_index, const auvp_rrt_params* p, int32_t flags) {
  if (!h) return AUVP_ERR_ARG;
  if (!h->have_world) return fail(h, AUVP_ERR_STATE, "auvp_world_set not called");
  if (E <= 0 || !init || !p) return fail(h, AUVP_ERR_ARG, "bad batch arguments");
  if (p->max_iter <= 0 || !(p->freq >= 0) || p->mode < 0 || p->mode > 2) return fail(h, AUVP_ERR_ARG, "bad params");
  if (p->mode == AUVP_MODE_TIMEBIN && !(p->bin_interval > 0)) return fail(h, AUVP_ERR_ARG, "bin_interval <= 0");
  if (h->W.n_obstacles > 16 * 64) return fail(h, AUVP_ERR_ARG, "n_obstacles %d > 1024", h->W.n_obstacles);
  HIPCHK(h, hipSetDevice(h->device));
  RrtParamsDev& P = h->P;
  P.dist_to_end = p->dist_to_end; P.diff_max = p->diff_max; P.freq = p->freq; P.min_dist = p->min_dist;
  P.bin_interval = p->bin_interval; P.v = p->v; P.max_traj_time = p->max_traj_time; P.max_plan_time = p->max_plan_time;
  P.w[0] = p->w[0]; P.w[1] = p->w[1]; P.w[2] = p->w[2];
  P.mode = p->mode; P.max_iter = p->max_iter; P.flags = flags;
  P.inv_bin_interval = p->bin_interval > 0 ? 1.0 / p->bin_interval : 0.0;
  P.K = p->mode == AUVP_MODE_TIMEBIN ? (int)std::ceil(p->max_traj_time / p->bin_interval) : 0;
  if (P.K > 1 << 20) return fail(h, AUVP_ERR_ARG, "too many time bins (%d)", P.K);
  const int nfreq = (int)std::floor(p->freq);
  h->max_pts = nfreq + 2;
  // the pre-generated random stream of earlier batches (tens of GB, rrt_run_pass) stays with the handle while the batches are of
  // its kind -- a caller alternating between worlds does not pay a hipMalloc per call -- and goes back before a batch of
  // another kind reserves its own buffers
  if (h->d_stream.p) {
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, h->device);
    if (!(P.mode == 0 && P.max_iter >= 1000 && h->opt_flag(OPT_ROWS_STREAM, true) && h->opt_flag(OPT_ROWS, E > 18 * (n_cu > 0 ? n_cu : 256))))
      h->d_stream.release();
  }
  RrtBuffers& B = h->B;
  B.cap_nodes = p->max_iter + 1;
  // Path-point budget.  A steer appends at most n = floor(uniform(0, freq)) points (:258-262) and only an accepted one
  // keeps them, so an episode's total is dominated by a sum of max_iter independent draws of n: mean <= freq/2 each,
  // standard deviation <= freq/sqrt(12).  The default budget is that mean plus EIGHT standard deviations of the sum
  // (exceeded with probability < 1e-15 per episode; the bench world uses 68 % of it), never more than the worst case;
  // points_per_iter overrides it (freq + 1 = the worst case).  Running out is AUVP_ERR_CAPACITY, never a truncation.
  const double n_it = (double)p->max_iter;
  const double worst = n_it * (nfreq > 1 ? nfreq : 1);
  double cp = p->points_per_iter > 0 ? std::ceil(p->points_per_iter * n_it)
                                     : std::ceil(std::fmin(worst, 0.5 * p->freq * n_it + 8.0 * p->freq / std::sqrt(12.0) * std::sqrt(n_it)));
  cp += nfreq + 64;
  if (cp > 2.0e9) return fail(h, AUVP_ERR_ARG, "point capacity too large");
  B.cap_points = (int32_t)cp;
  B.bin_cap = P.K > 0 ? B.cap_nodes : 1;
  // chunked member lists: AUVP_BIN_HEAD direct-mapped members per bin, the rest in 64-entry chunks (auvp_types.h)
  const bool chunks = B.bin_cap > AUVP_BIN_HEAD;
  B.bin_over = chunks ? (B.cap_nodes + 63) / 64 + P.K + 1 : 0;
  B.bin_slots = chunks ? (B.bin_cap - AUVP_BIN_HEAD + 63) / 64 : 0;
  B.bin_stride = (long long)(P.K + 1) * AUVP_BIN_HEAD + (long long)B.bin_over * 64 + (long long)B.bin_slots * (P.K + 1);
  if (B.bin_stride >= (1ll << 31)) return fail(h, AUVP_ERR_ARG, "time-bin lists too large (K=%d, max_iter=%d)", P.K, p->max_iter);
  B.cap_leaves = (flags & AUVP_FLAG_LEAF_LOG) ? B.cap_nodes : 1;
  const size_t cn = (size_t)E * B.cap_nodes, cpnt = (size_t)E * B.cap_points;
  HIPCHK(h, h->d_nodes_f.reserve(cn * 8 * sizeof(double)));
  HIPCHK(h, h->d_nodes_i.reserve(cn * 4 * sizeof(int32_t)));
  HIPCHK(h, h->d_points.reserve(cpnt * 6 * sizeof(double)));
  HIPCHK(h, h->d_bin_items.reserve((size_t)E * (size_t)B.bin_stride * sizeof(int32_t)));
  HIPCHK(h, h->d_bin_count.reserve((size_t)E * (P.K + 1) * sizeof(int32_t)));
  HIPCHK(h, h->d_summary.reserve((size_t)E * sizeof(RrtSummary)));
  HIPCHK(h, h->d_node_c.reserve(cn * 8 * sizeof(int32_t)));
  HIPCHK(h, h->d_node_q.reserve(cn));
  B.node_f = h->d_nodes_f.as<double>();
  B.node_i = h->d_nodes_i.as<int32_t>();
  B.points = h->d_points.as<double>();
  B.node_c = h->d_node_c.as<int32_t>();
  B.node_q = h->d_node_q.as<uint8_t>();
  // nearest-neighbour sampling scans a contiguous x,y mirror of the nodes (auvp_types.h); the other modes never touch it
  B.node_xy = nullptr;
  B.xy_stride = 0;
  if (p->mode == AUVP_MODE_NN) {
    B.xy_stride = rrt_nn_stride(B.cap_nodes);
    HIPCHK(h, h->d_node_xy.reserve((size_t)E * (size_t)B.xy_stride * 2 * sizeof(double)));
    B.node_xy = h->d_node_xy.as<double>();
  }
  B.bin_items = h->d_bin_items.as<int32_t>();
  B.bin_count = h->d_bin_count.as<int32_t>();
  B.summary = h->d_summary.as<RrtSummary>();
  B.it_parent = nullptr; B.it_accepted = nullptr; B.it_npath = nullptr; B.leaf_cost = nullptr; B.leaf_iter = nullptr;
  if (flags & AUVP_FLAG_ITER_LOG) {
    const size_t n = (size_t)E * p->max_iter;
    HIPCHK(h, h->d_itlog_i.reserve(n * 2 * sizeof(int32_t)));
    HIPCHK(h, h->d_itlog_b.reserve(n));
    B.it_parent = h->d_itlog_i.as<int32_t>();
    B.it_npath = B.it_parent + n;
    B.it_accepted = h->d_itlog_b.as<int8_t>();
  }
  if (flags & AUVP_FLAG_LEAF_LOG) {
    HIPCHK(h, h->d_leaf_c.reserve((size_t)E * B.cap_leaves * 6 * sizeof(double)));
    HIPCHK(h, h->d_leaf_i.reserve((size_t)E * B.cap_leaves * sizeof(int32_t)));
    B.leaf_cost = h->d_leaf_c.as<double>();
    B.leaf_iter = h->d_leaf_i.as<int32_t>();
  }
  B.phase_clocks = nullptr;
  HIPCHK(h, h->d_leaf_stats.reserve(8 * sizeof(unsigned long long)));
  B.leaf_stats = h->d_leaf_stats.as<unsigned long long>();
  B.pipe_fail = h->pipe_fail_dev;
  if (flags & AUVP_FLAG_PHASE_CLOCKS) {
    HIPCHK(h, h->d_phase.reserve((size_t)E * 5 * sizeof(unsigned long long)));
    HIPCHK(h, hipMemsetAsync(h->d_phase.p, 0, (size_t)E * 5 * sizeof(unsigned long long), h->stream));
    B.phase_clocks = h->d_phase.as<unsigned long long>();
  }
  // seeds -> MT states, once per batch: random.seed(seeds[e]) on the device (auvp_seed.h)
  int rc;
  B.mt_index = nullptr;
  if (seeds) {
    HIPCHK(h, h->d_mt.reserve((size_t)E * 624 * sizeof(uint32_t)));
    if ((rc = upload(h, h->d_seeds, seeds, (size_t)E))) return rc;
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(auvp::mt_seed_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  auvp::MT_SEED_LDS));
    hipLaunchKernelGGL(auvp::mt_seed_kernel, dim3((E + 63) / 64), dim3(64), auvp::MT_SEED_LDS, h->stream,
                       h->d_seeds.as<unsigned long long>(), h->d_mt.as<uint32_t>(), (int32_t*)nullptr, (int)E);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
  } else {
    if ((rc = upload(h, h->d_mt, states, (size_t)E * 624))) return rc;
    if ((rc = upload(h, h->d_mtidx, state_index, (size_t)E))) return rc;
    B.mt_index = h->d_mtidx.as<int32_t>();
  }
  if ((rc = upload(h, h->d_init, init, (size_t)E * 6))) return rc;
  B.mt = h->d_mt.as<uint32_t>();
  B.init = h->d_init.as<double>();
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->E = E;
  h->prepared = true;
  h->have_batch = false;
  return AUVP_OK;
}

}  // extern "C"

// one pass over the prepared batch: the expansion launch + the leaf pass.  one_wave_only: never a speculative pipeline
// The pre-generated random stream of a batch (rrt_stream_kernel.h): its length in numbers per episode -- from what the previous
// batches with the same parameters drew (`seen`), else 46.5 per iteration + 4 096 -- and its buffer.
static long long rrt_stream_len(const auvp_handle* h, const auvp_handle::Drawn* seen) {
  const long long guess = seen ? seen->most + seen->most * 3 / 100 + 1024 : (long long)(46.5 * (double)h->P.max_iter) + 4096;
  long long cap = h->opt_num(OPT_ROWS_STREAM_CAP, guess);
  cap = cap < 64 ? 64 : cap;
  return (cap + 63) / 64 * 64;
}
// 1: the buffer holds `bytes`; 0: it does not fit the free memory beside a 4 GB margin (or the allocation failed).  A buffer that
// has to grow is taken 6 % larger than asked for: the length follows the busiest episode seen so far, which creeps up by a few
// parts in a thousand from batch to batch, and every re-allocation of tens of GB is a second or two of hipFree + hipMalloc.
// *grew: an allocation happened (the caller re-records its start event: nothing has been launched yet)
static int rrt_stream_reserve(auvp_handle* h, size_t bytes, bool* grew = nullptr) {
  if (grew) *grew = false;
  if (h->d_stream.cap >= bytes) return 1;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return 0; }
  const size_t have = free_b + h->d_stream.cap, margin = (size_t)4 << 30;
  const size_t padded = bytes + bytes / 16;
  const size_t want = have >= padded + margin ? padded : bytes;
  if (have < want + margin || h->d_stream.reserve(want) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  if (grew) *grew = true;
  return 1;
}

// no_stream: the random numbers are generated inside the expansion kernel whatever option ROWS_STREAM says (the stream fallback)
static int rrt_run_pass(auvp_handle* h, bool one_wave_only, bool no_stream = false) {
  const RrtParamsDev& P = h->P;
  const RrtBuffers& B = h->B;
  const int E = h->E;
  const int nfreq = (int)std::floor(P.freq);
  const int O_ = h->W.n_obstacles;
  // Where the obstacles are dense the cull of a steer uses the tight box of its path points instead of the square of
  // its total movement (fewer exact tests for ~100 extra instructions): decided here from the expected number of
  // obstacles inside a typical reach square, 4 (freq dist_to_end / 4)^2 O / (area of the obstacles' bounding box).
  RrtParamsDev PR = P;
  {
    const double reach = 0.25 * P.freq * P.dist_to_end;
    const double lam = h->obst_area > 0.0 ? 4.0 * reach * reach * (double)O_ / h->obst_area : (O_ > 0 ? 1e9 : 0.0);
    if (h->opt_flag(OPT_TIGHT_CULL, lam > 0.5)) PR.flags |= AUVP_KFLAG_TIGHT_CULL;
    if (h->opt_on(OPT_NN_EXACT)) PR.flags |= AUVP_KFLAG_NN_EXACT;
  }
  const int jslots = (O_ <= 64 ? 1 : (O_ <= 128 ? 2 : (O_ <= 256 ? 4 : (O_ <= 512 ? 8 : 16)))) * 64;
  int n_cu_ = 256;
  (void)hipDeviceGetAttribute(&n_cu_, hipDeviceAttributeMultiprocessorCount, h->device);
  if (n_cu_ <= 0) n_cu_ = 256;
  const bool diag = (P.flags & (AUVP_FLAG_ITER_LOG | AUVP_FLAG_LEAF_LOG | AUVP_FLAG_PHASE_CLOCKS)) != 0;
  // Small batches (at most eight episodes per CU: config 2's 1 024 replicas) get workgroups of fewer waves, so that every CU
  // holds one (1 024 episodes: 256 workgroups of four waves = one wave per SIMD, instead of 128 CUs with two per SIMD:
  // 208 -> 229 M expansions/s).  A latency instantiation on top of that -- 124 VGPRs without the 80-register cap, the steer's
  // running sums as 32 unrolled steps with all reads up front -- was bit-identical and SLOWER (4.2 -> 5.7 us per expansion:
  // the loops only run n / 2 ~ 7 trips) and is not kept.
  const bool small_batch = E <= 8 * n_cu_;
  int xw = RRT_X_WAVES;
  if (small_batch) { xw = (E + n_cu_ - 1) / n_cu_; xw = xw < 1 ? 1 : (xw > RRT_X_WAVES ? RRT_X_WAVES : xw); }
  const size_t lds = (size_t)rrt_lds_plan(P.K, h->max_pts, nfreq, jslots,
                                          rrt_tables_bytes(h->W.n_habitats, h->W.n_poly, h->W.n_bins), xw).total;
  if (lds > 160 * 1024) return fail(h, AUVP_ERR_ARG, "LDS need %zu B > 160 KiB (K=%d, freq=%d)", lds, P.K, nfreq);
  const int grid = (E + xw - 1) / xw;
  const int O = h->W.n_obstacles;
  auto launch = [&](auto kern) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(xw * 64), lds, h->stream, h->W, PR, B, (int)E, h->max_pts);
    return hipGetLastError();
  };
  if (B.leaf_stats) HIPCHK(h, hipMemsetAsync(B.leaf_stats, 0, 8 * sizeof(unsigned long long), h->stream));  // (before the timed region)
  HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  hipError_t le = hipSuccess;
  // four episodes per wavefront (rrt_rows_kernel.h) where its limits allow; one episode per wavefront otherwise
  const RowsLdsPlan rp = rrt_rows_lds_plan(P.K, RW_MAX_OBST, rrt_tables_bytes(h->W.n_habitats, h->W.n_poly, h->W.n_bins));
  const bool iter_log = (P.flags & (AUVP_FLAG_ITER_LOG | AUVP_FLAG_PHASE_CLOCKS)) != 0;
  // ... and where it pays: a batch the one-episode kernel can keep resident in one go (6 waves per SIMD = 24 episodes per
  // CU) runs faster there -- the rows kernel would leave the SIMDs with one or two waves.  Measured on MI355X, M
  // expansions/s one-episode vs rows: 4 096 episodes 616 vs 507, 6 144 episodes 704 vs 645, 8 192 episodes 699 vs 849,
  // 10 240 episodes 737 vs 877.  End of round 6 (the rows kernel has lost a quarter of its instructions since), same batches:
  // 4 096 episodes 631 vs 606, 5 120: 609 vs 647, 6 144: 727 vs 773, 8 192: 715 vs 1 014 -- the crossover is between 16 and 20
  // episodes per CU now: rows above 18 (tools/batch_size_probe.py).  Option ROWS = 1 / 0 forces it on (limits permitting) / off.
  const bool rows_ok = P.mode == 0 && !iter_log && nfreq <= RW_MAX_FREQ && O_ <= RW_MAX_OBST && P.max_iter < 65534 &&
                       rp.total <= 160 * 1024;
  const bool use_rows = rows_ok && h->opt_flag(OPT_ROWS, E > 18 * n_cu_);
  int grid_used = grid, block_used = xw * 64, lds_used = (int)lds;
  bool stream_launched = false;
  // latency runs (at most four episodes per CU: one episode, config 2's 1 024 replicas): two wavefronts per episode
  // (rrt_duo_kernel.h).  Option DUO = 1 / 0 forces it on (limits permitting) / off.
  const bool duo_ok = P.mode == 0 && !diag && nfreq <= DUO_MAX_FREQ && nfreq >= 1 && O_ <= 256 && h->max_pts <= 64;
  // Measured (tools/duo_probe.py, M expansions/s one vs two wavefronts per episode): 1 episode 0.25 vs 0.32, 256: 62 vs 80,
  // 1 024: 227 vs 271 (config 2's replicas, 64 obstacles: 241 vs 283), 2 048: 409 vs 435, 4 096: 621 vs 485
  const bool use_duo = duo_ok && !use_rows && !one_wave_only && h->opt_flag(OPT_DUO, E <= 8 * n_cu_);
  // ... and three (rrt_trio_kernel.h: stream, geometry, tree -- a pipeline over the iterations) for at most four episodes per
  // CU.  Measured (tools/duo_probe.py, M expansions/s, one / two / three wavefronts per episode): 1 episode 0.25 / 0.32 / 0.40,
  // 256: 61 / 80 / 95, 1 024: 226 / 271 / 303 (config 2's replicas: 239 / 282 / 309), 2 048: 406 / 434 / 304.
  // Option TRIO = 1 / 0 forces it on (limits permitting) / off; an explicit DUO = 1 takes precedence.
  const bool use_trio = duo_ok && !use_rows && !one_wave_only && h->opt_flag(OPT_TRIO, E <= TRIO_EP * n_cu_ && !h->opt_on(OPT_DUO));
  h->last_rrt_kernel = use_rows ? "rrt_rows_kernel" : (use_trio ? "rrt_trio_kernel" : (use_duo ? "rrt_duo_kernel" : "rrt_explore_kernel"));
  h->last_stream_ms = 0.0;
  h->last_stream_len = 0;
  if (use_trio) {
    int eps_wg = (E + n_cu_ - 1) / n_cu_;
    eps_wg = eps_wg < 1 ? 1 : (eps_wg > TRIO_EP ? TRIO_EP : eps_wg);
    const int jd = O_ <= 64 ? 1 : (O_ <= 128 ? 2 : 4);
    const int dl = trio_lds_bytes(P.K, jd * 64, rrt_tables_bytes(h->W.n_habitats, h->W.n_poly, h->W.n_bins), eps_wg);
    if (dl > 160 * 1024) return fail(h, AUVP_ERR_ARG, "LDS need %d B > 160 KiB (K=%d)", dl, P.K);
    // the parent lookup as a fourth wavefront per episode (rrt_trio_kernel<J, 4>) where every episode has a CU to itself: one
    // episode 2.52 -> 2.49 us per expansion, 256 episodes 95 -> 100 M/s (1 024: 303 -> 262 M/s, so not there).  Option QUAD
    const bool quad = h->opt_flag(OPT_QUAD, E <= n_cu_);
    grid_used = (E + eps_wg - 1) / eps_wg; block_used = eps_wg * (quad ? 256 : 192); lds_used = dl;
    if (quad) h->last_rrt_kernel = "rrt_trio_kernel<4 wavefronts>";
    auto launch_trio = [&](auto kern) -> hipError_t {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, dl);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, dim3(grid_used), dim3(block_used), dl, h->stream, h->W, PR, B, (int)E);
      return hipGetLastError();
    };
    if (quad) le = jd == 1 ? launch_trio(rrt_trio_kernel<1, 4>) : (jd == 2 ? launch_trio(rrt_trio_kernel<2, 4>) : launch_trio(rrt_trio_kernel<4, 4>));
    else le = jd == 1 ? launch_trio(rrt_trio_kernel<1, 3>) : (jd == 2 ? launch_trio(rrt_trio_kernel<2, 3>) : launch_trio(rrt_trio_kernel<4, 3>));
  } else if (use_duo) {
    int eps_wg = (E + n_cu_ - 1) / n_cu_;
    eps_wg = eps_wg < 1 ? 1 : (eps_wg > DUO_EP ? DUO_EP : eps_wg);
    const int jd = O_ <= 64 ? 1 : (O_ <= 128 ? 2 : 4);
    const int dl = duo_lds_bytes(P.K, h->max_pts, jd * 64, rrt_tables_bytes(h->W.n_habitats, h->W.n_poly, h->W.n_bins), eps_wg);
    if (dl > 160 * 1024) return fail(h, AUVP_ERR_ARG, "LDS need %d B > 160 KiB (K=%d)", dl, P.K);
    grid_used = (E + eps_wg - 1) / eps_wg; block_used = eps_wg * 128; lds_used = dl;
    auto launch_duo = [&](auto kern) -> hipError_t {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, dl);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, dim3(grid_used), dim3(block_used), dl, h->stream, h->W, PR, B, (int)E, h->max_pts);
      return hipGetLastError();
    };
    le = jd == 1 ? launch_duo(rrt_duo_kernel<1>) : (jd == 2 ? launch_duo(rrt_duo_kernel<2>) : launch_duo(rrt_duo_kernel<4>));
  } else if (use_rows) {
    // a workgroup of up to 12 waves (48 episodes) fills one CU; a batch that cannot give every CU such a workgroup is
    // spread over all CUs with fewer waves per workgroup instead of leaving CUs idle
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, h->device);
    if (n_cu <= 0) n_cu = 256;
    int wg_waves = (E + RW_ROWS * n_cu - 1) / (RW_ROWS * n_cu);
    wg_waves = wg_waves < 1 ? 1 : (wg_waves > RW_WAVES ? RW_WAVES : wg_waves);
    const RowsLdsPlan rq = rrt_rows_lds_plan(P.K, RW_MAX_OBST, rrt_tables_bytes(h->W.n_habitats, h->W.n_poly, h->W.n_bins), wg_waves);
    const int per_wg = wg_waves * RW_ROWS;
    grid_used = (E + per_wg - 1) / per_wg; block_used = wg_waves * 64; lds_used = rq.total;
    // round 6: the episodes' random() numbers generated AHEAD by a launch of its own (rrt_stream_kernel.h: one wavefront per
    // episode, every lane busy) and read by rrt_rows_stream_kernel -- no generator, no tempering, 2.5 KB less LDS per episode in
    // the expansion kernel.  Measured on the headline batch: the expansion launch 93.7 -> 81.5 ms (953 instead of 1 284 vector
    // instructions per trip), generating 46 GB of numbers ahead 9.4 ms, the pass 99.6 -> 96.9 ms (profiles/r6_rows_stream.md).
    // The stream's length is a bound, and how many numbers an iteration draws depends on the PARAMETERS (44.8 with the bench's,
    // 92 with the leaves looked at every iteration) and hardly on the world (bench world, 64 / 256 obstacles, dense boxes,
    // concave outlines, accept rates 0.54 .. 0.99: the busiest of 1 024 episodes draws 45.56 .. 45.86 per iteration at 10 000
    // iterations, profiles/r6_rows_stream.md): it is set from what earlier batches with the same parameter block drew -- the
    // busiest episode seen + 3 % + 1 024 numbers (rrt_leaf_kernel reports the figure: leaf_stats[4]) -- whatever world they
    // ran on, so a caller that replans on a changing world (replanning: rrt_dubins.py:297-331) is served as well.  The first
    // batch with a parameter block runs rrt_rows_kernel and the following ones this path; an episode that runs past its
    // stream all the same is reported through the mapped flag and auvp_rrt_run redoes the batch with the kernel above (which
    // records the new figure).  Option ROWS_STREAM = 0: never; = 1: also without a previous batch (46.5 numbers
    // per iteration + 4 096), and a stream that does not fit the free memory beside a 4 GB margin is an error instead of a
    // quiet no; ROWS_STREAM_CAP: the length in numbers (tests).
    // (a batch at most four times the size of the one the figure comes from: the busiest of more episodes is busier)
    const auvp_handle::Drawn* seen = h->drawn_find(P);
    if (seen && (long long)seen->E * 4 < E) seen = nullptr;
    bool use_stream = !no_stream && P.max_iter >= 16 && h->opt_flag(OPT_ROWS_STREAM, seen != nullptr && P.max_iter >= 1000);
    long long cap = 0;
    bool grew = false;
    if (use_stream) {
      cap = rrt_stream_len(h, seen);
      if (cap > 0x7fffffffll) use_stream = false;  // (positions are 32-bit in the kernel)
      else if (!rrt_stream_reserve(h, (size_t)E * (size_t)cap * sizeof(double), &grew)) {
        if (h->opt_on(OPT_ROWS_STREAM)) return fail(h, AUVP_ERR_CAPACITY, "ROWS_STREAM = 1: %zu bytes of random stream do not fit the free memory", (size_t)E * (size_t)cap * sizeof(double));
        use_stream = false;
      }
    }
    if (grew) HIPCHK(h, hipEventRecord(h->ev0, h->stream));  // (the allocation is not part of the pass's time: nothing was launched yet)
    if (use_stream) {
      RrtBuffers Bs = B;
      Bs.stream = h->d_stream.as<double>();
      Bs.stream_cap = cap;
      // without the generator's state an episode needs 2.3 KB of LDS instead of 3.3: a CU holds 64 of them -- sixteen wavefronts,
      // four per SIMD -- but that instantiation spills (rows_kernels.hip): twelve at most, like rrt_rows_kernel (option
      // ROWS_STREAM_WAVES: fewer, for experiments)
      int sw = (E + RW_ROWS * n_cu - 1) / (RW_ROWS * n_cu);
      const int sw_max = (int)h->opt_num(OPT_ROWS_STREAM_WAVES, RW_WAVES);  // (the four-per-SIMD form measured 0.93 G expansions/s against 1.16: 100 B of scratch per lane at 128 registers)
      sw = sw < 1 ? 1 : (sw > sw_max ? sw_max : sw);
      sw = sw > RW_WAVES ? RW_WAVES : sw;
      while (sw > 1 && rrt_rows_stream_lds_plan(P.K, RW_MAX_OBST, rrt_tables_bytes(h->W.n_habitats, h->W.n_poly, h->W.n_bins), sw).total > 160 * 1024) sw--;
      const RowsStreamLdsPlan sp = rrt_rows_stream_lds_plan(P.K, RW_MAX_OBST, rrt_tables_bytes(h->W.n_habitats, h->W.n_poly, h->W.n_bins), sw);
      const RowsStreamLdsPlan sp_max = sp;
      grid_used = (E + sw * RW_ROWS - 1) / (sw * RW_ROWS); block_used = sw * 64;
      lds_used = sp.total;
      h->last_rrt_kernel = "rrt_rows_stream_kernel";
      h->last_stream_len = cap;
      le = auvpi_rrt_stream_launch(&Bs, (int)E, h->stream);
      if (le == hipSuccess) le = hipEventRecord(h->ev_pre, h->stream);
      if (le == hipSuccess) le = auvpi_rrt_rows_stream_launch(&h->W, &PR, &Bs, (int)E, grid_used, block_used, sp_max.total, sp.total, h->stream);
      stream_launched = true;
    } else
    le = auvpi_rrt_rows_launch(&h->W, &PR, &B, (int)E, grid_used, block_used, rp.total, rq.total, h->stream);  // (rows_kernels.hip)
  } else {
  // compile-time specialisation: obstacles per lane (J), parent-sampling mode, diagnostics on/off
  const int jsel = O <= 64 ? 0 : (O <= 128 ? 1 : (O <= 256 ? 2 : (O <= 512 ? 3 : 4)));
#define AUVP_LAUNCH_J(JV)                                                                   \
  do {                                                                                      \
    if (P.mode == 0) le = diag ? launch(rrt_explore_kernel<JV, 0, true>) : launch(rrt_explore_kernel<JV, 0, false>); \
    else if (P.mode == 1) le = diag ? launch(rrt_explore_kernel<JV, 1, true>) : launch(rrt_explore_kernel<JV, 1, false>); \
    else le = diag ? launch(rrt_explore_kernel<JV, 2, true>) : launch(rrt_explore_kernel<JV, 2, false>); \
  } while (0)
  switch (jsel) {
    case 0: AUVP_LAUNCH_J(1); break;
    case 1: AUVP_LAUNCH_J(2); break;
    case 2: AUVP_LAUNCH_J(4); break;
    case 3: AUVP_LAUNCH_J(8); break;
    default: AUVP_LAUNCH_J(16); break;
  }
#undef AUVP_LAUNCH_J
  }
  HIPCHK(h, le);
  // the trees are complete: rank the qualifying leaves (same stream, inside the timed region)
  HIPCHK(h, hipEventRecord(h->ev_mid, h->stream));
  {
    // dynamic LDS of the leaf pass: the separable-grid edge tables + one "ancestor of a qualifying leaf" bit per node
    // and episode (trees too large for that are swept whole)
    const int gl = rrt_leaf_grid_lds_bytes(h->W.sg_enabled, h->W.sg_ncol, h->W.sg_nrow);
    // (option LEAF_SWEEP_ALL: no pruning, every node visited -- what trees of more than 131 072 nodes get; for tests)
    const int bm_words = h->opt_on(OPT_LEAF_SWEEP_ALL) ? 0 : rrt_leaf_mark_words(B.cap_nodes);
    const int dyn = gl + RRT_LEAF_WAVES * bm_words * 4;
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(rrt_leaf_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, dyn));
    hipLaunchKernelGGL(rrt_leaf_kernel, dim3((E + RRT_LEAF_WAVES - 1) / RRT_LEAF_WAVES), dim3(RRT_LEAF_WAVES * 64), dyn, h->stream,
                       h->W, P, B, (int)E, bm_words);
  }
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipEventRecord(h->ev1, h->stream));
  // (the most 32-bit outputs one episode drew: eight bytes into the mapped page beside the pipeline flag)
  const bool want_drawn = B.leaf_stats && h->pipe_fail_host;
  if (want_drawn) HIPCHK(h, hipMemcpyAsync(h->pipe_fail_host + 2, B.leaf_stats + 4, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (want_drawn && !h->pipe_failed()) {
    unsigned long long d32 = 0;
    memcpy(&d32, h->pipe_fail_host + 2, sizeof d32);
    if (d32 > 0) {
      // (the most seen with these parameters: worlds that alternate between a little more and a little less do not make every
      // other batch run past its stream; a parameter block not seen before takes the slot used longest ago)
      auvp_handle::Drawn* d = h->drawn_find(P);
      const long long now = (long long)((d32 + 1) / 2);
      if (!d) {
        d = &h->drawn[0];
        for (auvp_handle::Drawn& x : h->drawn)
          if (!x.valid || (d->valid && x.used < d->used)) d = &x;
        *d = auvp_handle::Drawn{};
        d->P = P;
        d->valid = true;
      }
      d->most = d->most > now ? d->most : now;
      d->E = d->E > E ? d->E : E;
      d->used = ++h->drawn_clock;
      // the next batch with these parameters will want its stream: the buffer is taken NOW, in the call that found
      // out (tens of GB: a second of hipMalloc that a later, timed call would pay otherwise)
      if (use_rows && !stream_launched && P.max_iter >= 1000 && h->opt_flag(OPT_ROWS_STREAM, true)) {
        const long long cap_next = rrt_stream_len(h, d);
        if (cap_next <= 0x7fffffffll) (void)rrt_stream_reserve(h, (size_t)E * (size_t)cap_next * sizeof(double));
      }
    }
  }
  float ms = 0.f;
  HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
  h->last_ms = ms;
  HIPCHK(h, hipEventElapsedTime(&ms, stream_launched ? h->ev_pre : h->ev0, h->ev_mid));
  h->last_expand_ms = ms;
  if (stream_launched) {
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev_pre));
    h->last_stream_ms = ms;
  }
  HIPCHK(h, hipEventElapsedTime(&ms, h->ev_mid, h->ev1));
  h->last_leaf_ms = ms;
  h->last_rows = use_rows ? 1 : 0;
  h->last_grid = grid_used; h->last_block = block_used; h->last_lds = lds_used;
  h->have_batch = true;
  return AUVP_OK;
}

// AUVP_ERR_PIPELINE episodes of the pass that just ran (its mapped flag is set): counted from the summaries
static int rrt_count_pipeline_failures(auvp_handle* h, int* n, int* n_stream) {
  std::vector<RrtSummary> s((size_t)h->E);
  HIPCHK(h, hipMemcpy(s.data(), h->B.summary, s.size() * sizeof(RrtSummary), hipMemcpyDeviceToHost));
  *n = 0;
  *n_stream = 0;
  for (const RrtSummary& r : s) { *n += r.status == AUVP_ERR_PIPELINE ? 1 : 0; *n_stream += r.status == AUVP_ERR_STREAM ? 1 : 0; }
  return AUVP_OK;
}

extern "C" {

int auvp_rrt_run(auvp_handle* h) {
  if (!h) return AUVP_ERR_ARG;
  if (!h->prepared) return fail(h, AUVP_ERR_STATE, "auvp_rrt_prepare not called");
  HIPCHK(h, hipSetDevice(h->device));
  h->pipe_clear();
  h->pipe_fallback_last = 0;
  int rc = rrt_run_pass(h, false);
  if (rc != AUVP_OK || !h->pipe_failed()) return rc;
  // a speculative pipeline gave up on some episode: the batch starts from its prepared state in every pass, so the whole
  // pass is repeated on the one-wavefront kernel (option PIPE_FALLBACK = 0: leave the status in the summaries instead)
  int n = 0, n_stream = 0;
  if ((rc = rrt_count_pipeline_failures(h, &n, &n_stream))) return rc;
  h->pipe_clear();
  if (n + n_stream == 0 || !h->opt_flag(OPT_PIPE_FALLBACK, true)) return AUVP_OK;
  h->pipe_fallback_last = n + n_stream;
  h->pipe_fallback_total += n + n_stream;
  const double first_ms = h->last_ms;
  // (episodes that ran past their pre-generated random stream: the same kernel shape with the generator inside; a pipeline that
  // gave up: the one-wavefront kernel)
  rc = n_stream > 0 ? rrt_run_pass(h, false, true) : rrt_run_pass(h, true);
  h->last_ms += first_ms;  // (the time the caller waited)
  return rc;
}

int auvp_rrt_summaries(auvp_handle* h, auvp_rrt_summary* out) {
  if (!h || !out) return AUVP_ERR_ARG;
  if (!h->have_batch) return fail(h, AUVP_ERR_STATE, "no batch has run");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpyAsync(out, h->B.summary, (size_t)h->E * sizeof(RrtSummary), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return AUVP_OK;
}

void* auvp_rrt_summaries_dev(auvp_handle* h) { return (h && h->have_batch) ? (void*)h->B.summary : nullptr; }

int auvp_rrt_paths_dev(auvp_handle* h, const int64_t* offsets, void* out_dev) {
  if (!h || !offsets || !out_dev) return AUVP_ERR_ARG;
  if (!h->have_batch) return fail(h, AUVP_ERR_STATE, "no batch has run");
  HIPCHK(h, hipSetDevice(h->device));
  const int E = h->E;
  HIPCHK(h, h->d_tmp0.reserve((size_t)(E + 1) * sizeof(int64_t)));
  HIPCHK(h, hipMemcpyAsync(h->d_tmp0.p, offsets, (size_t)(E + 1) * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(rrt_final_course_kernel, dim3(E), dim3(64), 0, h->stream, h->B, h->d_tmp0.as<int64_t>(),
                     reinterpret_cast<double*>(out_dev), E);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return AUVP_OK;
}

int auvp_rrt_paths(auvp_handle* h, const int64_t* offsets, double* out) {
  if (!h || !offsets || !out) return AUVP_ERR_ARG;
  if (!h->have_batch) return fail(h, AUVP_ERR_STATE, "no batch has run");
  const size_t total = (size_t)offsets[h->E];
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, h->d_tmp1.reserve(std::max<size_t>(total, 1) * 7 * sizeof(double)));
  int rc = auvp_rrt_paths_dev(h, offsets, h->d_tmp1.p);
  if (rc != AUVP_OK) return rc;
  if (total) HIPCHK(h, hipMemcpyAsync(out, h->d_tmp1.p, total * 7 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return AUVP_OK;
}

int auvp_rrt_tree(auvp_handle* h, int32_t ep, double* nodes6, int32_t* parent, int32_t* pt_off, int32_t* pt_cnt,
                  double* points7) {
  if (!h) return AUVP_ERR_ARG;
  if (!h->have_batch || ep < 0 || ep >= h->E) return fail(h, AUVP_ERR_STATE, "bad episode");
  HIPCHK(h, hipSetDevice(h->device));
  RrtSummary s;
  HIPCHK(h, hipMemcpy(&s, h->B.summary + ep, sizeof s, hipMemcpyDeviceToHost));
  const RrtBuffers& B = h->B;
  const int N = s.n_nodes, NP = s.n_points;
  const size_t capp = (size_t)B.cap_points;
  std::vector<double> nf((size_t)N * 8);
  std::vector<int32_t> ni((size_t)N * 4);
  HIPCHK(h, hipMemcpy(nf.data(), B.node_f + (size_t)ep * B.cap_nodes * 8, nf.size() * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(ni.data(), B.node_i + (size_t)ep * B.cap_nodes * 4, ni.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  for (int i = 0; i < N; i++) {
    if (nodes6) {
      double* d = nodes6 + 6 * (size_t)i;
      const double* f = nf.data() + 8 * (size_t)i;
      d[0] = f[0]; d[1] = f[1]; d[2] = f[2]; d[3] = f[3]; d[4] = (double)ni[4 * (size_t)i]; d[5] = f[4];
    }
    if (parent) parent[i] = ni[4 * (size_t)i + 1];
    if (pt_off) pt_off[i] = ni[4 * (size_t)i + 2];
    if (pt_cnt) pt_cnt[i] = ni[4 * (size_t)i + 3];
  }
  if (points7 && NP > 0) {
    // the points live in two 24-byte record arrays per episode: {x, y, traj_t} and {theta, v, length}
    std::vector<double> ra((size_t)NP * 3), rb((size_t)NP * 3);
    HIPCHK(h, hipMemcpy(ra.data(), B.points + (size_t)ep * 6 * capp, ra.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(rb.data(), B.points + (size_t)ep * 6 * capp + 3 * capp, rb.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int i = 0; i < NP; i++) {
      double* d = points7 + 7 * (size_t)i;
      d[0] = ra[3 * (size_t)i]; d[1] = ra[3 * (size_t)i + 1]; d[2] = rb[3 * (size_t)i]; d[3] = rb[3 * (size_t)i + 1];
      d[4] = ra[3 * (size_t)i + 2]; d[6] = rb[3 * (size_t)i + 2];
    }
    // plan_time_stamp of a path point = iteration of the node that owns it
    for (int m = 0; m < N; m++) {
      const int off = ni[4 * (size_t)m + 2], cnt = ni[4 * (size_t)m + 3], plan = ni[4 * (size_t)m];
      for (int k = 0; k < cnt; k++) points7[7 * (size_t)(off + k) + 5] = (double)plan;
    }
  }
  return AUVP_OK;
}

int auvp_rrt_iter_log(auvp_handle* h, int32_t ep, int32_t* it_parent, int8_t* it_accepted, int32_t* it_npath) {
  if (!h) return AUVP_ERR_ARG;
  if (!h->have_batch || ep < 0 || ep >= h->E || !h->B.it_parent) return fail(h, AUVP_ERR_STATE, "no iteration log");
  HIPCHK(h, hipSetDevice(h->device));
  const size_t n = h->P.max_iter, o = (size_t)ep * n;
  if (it_parent) HIPCHK(h, hipMemcpy(it_parent, h->B.it_parent + o, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (it_accepted) HIPCHK(h, hipMemcpy(it_accepted, h->B.it_accepted + o, n, hipMemcpyDeviceToHost));
  if (it_npath) HIPCHK(h, hipMemcpy(it_npath, h->B.it_npath + o, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  return AUVP_OK;
}

int auvp_rrt_leaf_log(auvp_handle* h, int32_t ep, double* leaf_cost6, int32_t* leaf_iter) {
  if (!h) return AUVP_ERR_ARG;
  if (!h->have_batch || ep < 0 || ep >= h->E || !h->B.leaf_cost) return fail(h, AUVP_ERR_STATE, "no leaf log");
  HIPCHK(h, hipSetDevice(h->device));
  RrtSummary s;
  HIPCHK(h, hipMemcpy(&s, h->B.summary + ep, sizeof s, hipMemcpyDeviceToHost));
  const size_t n = std::min(s.n_leaves, h->B.cap_leaves), o = (size_t)ep * h->B.cap_leaves;
  if (leaf_cost6 && n) HIPCHK(h, hipMemcpy(leaf_cost6, h->B.leaf_cost + o * 6, n * 6 * sizeof(double), hipMemcpyDeviceToHost));
  if (leaf_iter && n) HIPCHK(h, hipMemcpy(leaf_iter, h->B.leaf_iter + o, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  return AUVP_OK;
}

int auvp_rrt_bin_sizes(auvp_handle* h, int32_t ep, int32_t* sizes, int32_t* n_bins) {
  if (!h) return AUVP_ERR_ARG;
  if (!h->have_batch || ep < 0 || ep >= h->E) return fail(h, AUVP_ERR_STATE, "bad episode");
  HIPCHK(h, hipSetDevice(h->device));
  const int K = h->P.K;
  if (n_bins) *n_bins = K;
  if (sizes && K > 0)
    HIPCHK(h, hipMemcpy(sizes, h->B.bin_count + (size_t)ep * (K + 1) + 1, (size_t)K * sizeof(int32_t), hipMemcpyDeviceToHost));
  return AUVP_OK;
}

int auvp_rrt_phase_clocks(auvp_handle* h, uint64_t* out) {
  if (!h || !out) return AUVP_ERR_ARG;
  if (!h->have_batch || !h->B.phase_clocks) return fail(h, AUVP_ERR_STATE, "no phase clocks (AUVP_FLAG_PHASE_CLOCKS)");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpy(out, h->B.phase_clocks, (size_t)h->E * 5 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return AUVP_OK;
}

double auvp_last_kernel_ms(auvp_handle* h) { return h ? h->last_ms : -1.0; }

int auvp_rrt_last_launch_parts(auvp_handle* h, double* expand_ms, double* leaf_ms, int32_t* episodes_per_wave) {
  if (!h) return AUVP_ERR_ARG;
  if (!h->have_batch) return fail(h, AUVP_ERR_STATE, "no batch has run");
  if (expand_ms) *expand_ms = h->last_expand_ms;
  if (leaf_ms) *leaf_ms = h->last_leaf_ms;
  if (episodes_per_wave) *episodes_per_wave = h->last_rows ? 4 : 1;
  return AUVP_OK;
}

const char* auvp_rrt_last_kernel(auvp_handle* h) { return h ? h->last_rrt_kernel : ""; }

double auvp_rrt_last_stream_ms(auvp_handle* h) { return h ? h->last_stream_ms : -1.0; }
int64_t auvp_rrt_last_stream_len(auvp_handle* h) { return h ? (int64_t)h->last_stream_len : -1; }

int auvp_rrt_last_leaf_stats(auvp_handle* h, int64_t* out4) {
  if (!h || !out4) return AUVP_ERR_ARG;
  if (!h->have_batch || !h->B.leaf_stats) return fail(h, AUVP_ERR_STATE, "no batch has run");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpy(out4, h->B.leaf_stats, 4 * sizeof(int64_t), hipMemcpyDeviceToHost));
  return AUVP_OK;
}

int auvp_last_launch(auvp_handle* h, int32_t* grid, int32_t* block, int32_t* lds_bytes) {
  if (!h) return AUVP_ERR_ARG;
  if (grid) *grid = h->last_grid;
  if (block) *block = h->last_block;
  if (lds_bytes) *lds_bytes = h->last_lds;
  return AUVP_OK;
}

}  // extern "C"

#include "probe_kernels.h"
#include "planner_rrt_kernel.h"
#include "planner_rows_kernel.h"
#include "planner_goal_arc.h"
#include "planner_pipe_kernel.h"
#include "planner_rrt_host.h"

namespace {
PrrtState* prrt_of(auvp_handle* h) {
  if (!h->prrt) {
    h->prrt = new PrrtState();
    h->prrt_free = [](void* p) { delete static_cast<PrrtState*>(p); };
  }
  return static_cast<PrrtState*>(h->prrt);
}
}  // namespace

#include "astar_kernel.h"
#include "astar_host.h"
#include "sog_kernels.h"
#include "pf_types.h"
#include "pf_host.h"
#include "compose_host.h"
#include "gather_host.h"
