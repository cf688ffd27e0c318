// probe_kernels.h -- standalone device evaluations of the building blocks (collision test, cost
// function, portable sin/cos, wave-level MT19937).  They run the same device functions the planner
// kernels use, one wavefront per path, so the parity tests can pin each block against the golden
// vectors on its own.  Included at the end of auvplan.hip (same translation unit).
#ifndef AUVP_PROBE_KERNELS_H
#define AUVP_PROBE_KERNELS_H
#include "auvp_exp.h"

namespace auvp {

// RRT.check_collision (path_planning/rrt_dubins.py:530-549): one wave per path
__global__ __launch_bounds__(64) void collision_probe_kernel(WorldDev W, int n_paths, const int32_t* __restrict__ off,
                                                             const double* __restrict__ pts, int8_t* __restrict__ out) {
  __shared__ double poly[RRT_MAX_POLY][2];
  const int lane = lane_id();
  for (int i = lane; i < W.n_poly * 2; i += 64) (&poly[0][0])[i] = W.poly[i];
  __syncthreads();
  const int p = blockIdx.x;
  if (p >= n_paths) return;
  const int b = off[p], e = off[p + 1];
  bool hit = false;
  for (int i = lane; i < W.n_obstacles; i += 64) {
    const double ox = W.ox[i], oy = W.oy[i], ot = W.ot[i];
    for (int k = b; k < e; k++) {
      double dx = pts[2 * k] - ox, dy = pts[2 * k + 1] - oy;
      hit = hit || (dx * dx + dy * dy <= ot);
    }
  }
  const bool outside = any_point_outside(poly, W.n_poly, reinterpret_cast<const double(*)[2]>(pts + 2 * (size_t)b), e - b);
  bool ok = !wave_any(hit) && !outside;
  if (lane == 0) out[p] = ok ? 1 : 0;
}

// habitat_shark_cost_func (path_planning/cost.py:145-207): one wave per path
__global__ __launch_bounds__(64) void cost_probe_kernel(WorldDev W, int n_paths, const int32_t* __restrict__ off,
                                                        const double* __restrict__ pts, const int32_t* __restrict__ blo,
                                                        const int32_t* __restrict__ bhi, const double* __restrict__ total,
                                                        const double* __restrict__ w, double* __restrict__ out) {
  __shared__ __align__(16) unsigned char tables[RRT_WORLD_BYTES + RRT_MAX_HAB * 32 + RRT_MAX_POLY * 16 + RRT_MAX_BINS * 16];
  const RrtTables S = rrt_tables_view(tables, W.n_habitats, W.n_poly);
  const int lane = lane_id();
  __shared__ double term[64];
  rrt_tables_stage(S, W);
  __syncthreads();
  const int p = blockIdx.x;
  if (p >= n_paths) return;
  const int b = off[p], e = off[p + 1];
  const double w1 = w[3 * p], w2 = w[3 * p + 1], w3 = w[3 * p + 2];
  CostAcc acc;
  acc.c2 = 0.0; acc.visited = 0ull; acc.hits = 0;
  for (int s0 = b; s0 < e; s0 += 64) {
    int k = s0 + lane;
    bool valid = k < e;
    int nv = (e - s0) < 64 ? (e - s0) : 64;
    double x = valid ? pts[3 * k] : 0.0, y = valid ? pts[3 * k + 1] : 0.0, t = valid ? pts[3 * k + 2] : 0.0;
    cost_pass(W, S, blo[p], bhi[p], w3, nv, x, y, t, term, acc);
  }
  double c0 = 0.0, c1 = 0.0, c2 = acc.c2;
  if (w2 == auvp_rint(w2) && auvp_fabs(w2) < 1048576.0) c1 = w2 * (double)acc.hits;
  else for (int h = 0; h < acc.hits; h++) c1 = c1 + w2;
  const double tt = total[p];
  if (tt > 0) { c1 = c1 / tt; c2 = c2 / tt; }
  if (W.n_habitats != 0) c0 = w1 * (double)__popcll(acc.visited) / (double)W.n_habitats;
  if (lane == 0) {
    out[4 * p] = ((0.0 + c0) + c1) + c2;
    out[4 * p + 1] = c0; out[4 * p + 2] = c1; out[4 * p + 3] = c2;
  }
}

// get_closest_mps (path_planning/rrt_dubins.py:505-513): one wave per query against one padded x,y table
__global__ __launch_bounds__(64) void nn_probe_kernel(const double2* __restrict__ xy, int n_nodes, int n_queries,
                                                      const double* __restrict__ q, int force_exact, int32_t* __restrict__ out,
                                                      int32_t* __restrict__ slow) {
  const int i = blockIdx.x;
  if (i >= n_queries) return;
  int sl = 0;
  const int r = nn_closest(xy, n_nodes, readfirst_f64(q[2 * i]), readfirst_f64(q[2 * i + 1]), force_exact != 0, &sl);
  if (lane_id() == 0) { out[i] = r; slow[i] = sl; }
}

// ---- HBM streaming probes: the MEASURED roof the bench's bandwidth fractions are quoted against -------------------------
// Read probe: every lane keeps U (8 or 16) 16-byte loads in flight (U = 8 is the access shape of the nearest-neighbour scan:
// 1 KB per wave-instruction, 8 KB per wave and trip), grid-stride over U x 4-KB chunks per workgroup; the values are folded
// into one word per lane that is stored only if it equals a value the memset pattern cannot produce (keeps the loads
// alive).  The host tries both depths at four and eight workgroups per CU and reports the best rate.
template <int U>
__global__ __launch_bounds__(256) void hbm_read_probe_kernel(const uint4* __restrict__ src, unsigned long long n16, uint32_t* __restrict__ sink) {
  const unsigned long long chunk = 256ull * (unsigned long long)U;  // uint4 elements per workgroup and trip
  uint4 acc = make_uint4(0u, 0u, 0u, 0u);
  for (unsigned long long c0 = (unsigned long long)blockIdx.x * chunk; c0 + chunk <= n16; c0 += (unsigned long long)gridDim.x * chunk) {
    uint4 q[U];
#pragma unroll
    for (int u = 0; u < U; u++) q[u] = src[c0 + (unsigned long long)u * 256ull + threadIdx.x];
#pragma unroll
    for (int u = 0; u < U; u++) { acc.x ^= q[u].x; acc.y += q[u].y; acc.z ^= q[u].z; acc.w += q[u].w; }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) sink[blockIdx.x] = acc.x;
}
// Copy probe: 16-byte loads and stores, four in flight per lane; bytes moved = 2 x the buffer half
__global__ __launch_bounds__(256) void hbm_copy_probe_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, unsigned long long n16) {
  const unsigned long long chunk = 256ull * 4ull;
  for (unsigned long long c0 = (unsigned long long)blockIdx.x * chunk; c0 + chunk <= n16; c0 += (unsigned long long)gridDim.x * chunk) {
    uint4 q[4];
#pragma unroll
    for (int u = 0; u < 4; u++) q[u] = src[c0 + (unsigned long long)u * 256ull + threadIdx.x];
#pragma unroll
    for (int u = 0; u < 4; u++) dst[c0 + (unsigned long long)u * 256ull + threadIdx.x] = q[u];
  }
}

__global__ void sincos_probe_kernel(int n, const double* __restrict__ x, double* __restrict__ s, double* __restrict__ c) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) auvp_sincos(x[i], &s[i], &c[i]);
}

// one portable function per op (auvp_math_dev): 0 sin / cos, 1 atan2(a, b), 2 pow(e, a), 3 auvp_div_plain(a, b),
// 4 auvp_sqrt_plain(a), 5 hypot(a, b), 6 a / b, 7 sqrt(a)
__global__ void math_probe_kernel(int op, int n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ o0,
                                  double* __restrict__ o1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = a[i], y = b[i];
  double r0 = 0.0, r1 = 0.0;
  switch (op) {
    case 0: auvp_sincos(x, &r0, &r1); break;
    case 1: r0 = auvp_atan2(x, y); break;
    case 2: r0 = auvp_pow_e(x); break;
    case 3: r0 = auvp_div_plain(x, y); break;
    case 4: r0 = auvp_sqrt_plain(x); break;
    case 5: r0 = auvp_hypot(x, y); break;
    case 6: r0 = x / y; break;
    default: r0 = auvp_sqrt(x); break;
  }
  o0[i] = r0;
  o1[i] = r1;
}

__global__ __launch_bounds__(64) void random_probe_kernel(const uint32_t* __restrict__ mt, int n, double* __restrict__ out) {
  __shared__ uint32_t s[624];
  const int lane = lane_id();
  for (int i = lane; i < 624; i += 64) s[i] = mt[i];
  WaveRng r;
  r.s = s; r.pslot = 0; r.avail = 0; r.drawn = 0ull;
  wave_sync();
  // mix the two access patterns the planners use: single draws and lane-parallel windows
  int done = 0;
  while (done < n) {
    int w = (done % 7 == 0) ? 1 : ((done * 13) % 150 + 1);
    if (w > n - done) w = n - done;
    if (w == 1) {
      double v = rng_next_random(r);
      if (lane == 0) out[done] = v;
    } else {
      rng_ensure(r, (uint32_t)(2 * w));
      for (int j = lane; j < w; j += 64) out[done + j] = rng_random_at(r, (uint32_t)j);
      rng_advance_words(r, (uint32_t)(2 * w));
    }
    done += w;
  }
}

}  // namespace auvp

extern "C" {

int auvp_check_collision_batch(auvp_handle* h, int32_t n_paths, const int32_t* off, const double* pts_xy, int8_t* out_free) {
  if (!h || n_paths < 0 || !off || !out_free) return AUVP_ERR_ARG;
  if (!h->have_world) return fail(h, AUVP_ERR_STATE, "auvp_world_set not called");
  if (n_paths == 0) return AUVP_OK;
  HIPCHK(h, hipSetDevice(h->device));
  const size_t npts = (size_t)off[n_paths];
  int rc;
  if ((rc = upload(h, h->d_tmp0, off, (size_t)n_paths + 1))) return rc;
  if ((rc = upload(h, h->d_tmp1, pts_xy, npts * 2))) return rc;
  HIPCHK(h, h->d_tmp2.reserve((size_t)n_paths));
  hipLaunchKernelGGL(collision_probe_kernel, dim3(n_paths), dim3(64), 0, h->stream, h->W, (int)n_paths,
                     h->d_tmp0.as<int32_t>(), h->d_tmp1.as<double>(), h->d_tmp2.as<int8_t>());
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(out_free, h->d_tmp2.p, (size_t)n_paths, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return AUVP_OK;
}

int auvp_cost_paths(auvp_handle* h, int32_t n_paths, const int32_t* off, const double* pts_xyt, const int32_t* bin_lo,
                    const int32_t* bin_hi, const double* total_traj_time, const double* weights3, double* out4) {
  if (!h || n_paths < 0 || !off || !out4) return AUVP_ERR_ARG;
  if (!h->have_world) return fail(h, AUVP_ERR_STATE, "auvp_world_set not called");
  if (n_paths == 0) return AUVP_OK;
  HIPCHK(h, hipSetDevice(h->device));
  const size_t npts = (size_t)off[n_paths];
  int rc;
  if ((rc = upload(h, h->d_tmp0, off, (size_t)n_paths + 1))) return rc;
  if ((rc = upload(h, h->d_tmp1, pts_xyt, npts * 3))) return rc;
  if ((rc = upload(h, h->d_tmp2, bin_lo, (size_t)n_paths))) return rc;
  if ((rc = upload(h, h->d_tmp3, bin_hi, (size_t)n_paths))) return rc;
  if ((rc = upload(h, h->d_tmp4, total_traj_time, (size_t)n_paths))) return rc;
  if ((rc = upload(h, h->d_tmp5, weights3, (size_t)n_paths * 3))) return rc;
  HIPCHK(h, h->d_leaf_c.reserve((size_t)n_paths * 4 * sizeof(double)));
  hipLaunchKernelGGL(cost_probe_kernel, dim3(n_paths), dim3(64), 0, h->stream, h->W, (int)n_paths, h->d_tmp0.as<int32_t>(),
                     h->d_tmp1.as<double>(), h->d_tmp2.as<int32_t>(), h->d_tmp3.as<int32_t>(), h->d_tmp4.as<double>(),
                     h->d_tmp5.as<double>(), h->d_leaf_c.as<double>());
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(out4, h->d_leaf_c.p, (size_t)n_paths * 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return AUVP_OK;
}

int auvp_nn_closest_batch(auvp_handle* h, int32_t n_nodes, const double* xy, int32_t n_queries, const double* q, int32_t force_exact,
                          int32_t* out_index, int32_t* out_slow) {
  if (!h || n_nodes <= 0 || n_queries < 0 || !xy || !q || !out_index) return AUVP_ERR_ARG;
  if (n_queries == 0) return AUVP_OK;
  HIPCHK(h, hipSetDevice(h->device));
  const size_t stride = (size_t)rrt_nn_stride(n_nodes);
  std::vector<double> pad(stride * 2, INFINITY);  // the scan reads whole blocks: +inf past the end, as in the planner
  memcpy(pad.data(), xy, (size_t)n_nodes * 2 * sizeof(double));
  int rc;
  if ((rc = upload(h, h->d_tmp0, pad.data(), pad.size()))) return rc;
  if ((rc = upload(h, h->d_tmp1, q, (size_t)n_queries * 2))) return rc;
  HIPCHK(h, h->d_tmp2.reserve((size_t)n_queries * sizeof(int32_t)));
  HIPCHK(h, h->d_tmp3.reserve((size_t)n_queries * sizeof(int32_t)));
  hipLaunchKernelGGL(nn_probe_kernel, dim3(n_queries), dim3(64), 0, h->stream, h->d_tmp0.as<double2>(), (int)n_nodes, (int)n_queries,
                     h->d_tmp1.as<double>(), (int)force_exact, h->d_tmp2.as<int32_t>(), h->d_tmp3.as<int32_t>());
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(out_index, h->d_tmp2.p, (size_t)n_queries * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  if (out_slow) HIPCHK(h, hipMemcpyAsync(out_slow, h->d_tmp3.p, (size_t)n_queries * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return AUVP_OK;
}

int auvp_hbm_probe(auvp_handle* h, uint64_t bytes, int32_t reps, double* read_GBps, double* copy_GBps) {
  if (!h || reps <= 0 || !read_GBps) return AUVP_ERR_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  // whole 64-KB chunks; at least 64 MB so that the launch is long against its own start-up
  const unsigned long long chunk_b = 256ull * 16ull * 16ull;
  unsigned long long nb = (bytes < (64ull << 20) ? (64ull << 20) : bytes) / chunk_b * chunk_b;
  DevBuf buf, sink;
  HIPCHK(h, buf.reserve((size_t)nb));
  int n_cu = 256;
  (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, h->device);
  if (n_cu <= 0) n_cu = 256;
  const int grid = n_cu * 8;
  HIPCHK(h, sink.reserve((size_t)grid * sizeof(uint32_t)));
  HIPCHK(h, hipMemsetAsync(buf.p, 0x5a, (size_t)nb, h->stream));
  const unsigned long long n16 = nb / 16ull;
  auto timed = [&](auto enqueue, double moved, double* out) -> int {
    enqueue();  // warm-up (page tables, clocks)
    double best = 0.0;
    for (int r = 0; r < reps; r++) {
      HIPCHK(h, hipEventRecord(h->ev0, h->stream));
      enqueue();
      HIPCHK(h, hipGetLastError());
      HIPCHK(h, hipEventRecord(h->ev1, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
      float ms = 0.f;
      HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
      if (ms > 0.f) best = std::max(best, moved / (ms * 1e-3) / 1e9);
    }
    *out = best;
    return AUVP_OK;
  };
  int rc = AUVP_OK;
  double best_read = 0.0;
  for (int wgs = 4; wgs <= 8 && rc == AUVP_OK; wgs *= 2) {
    const int g = n_cu * wgs;
    double r8 = 0.0, r16 = 0.0;
    rc = timed([&] { hipLaunchKernelGGL(hbm_read_probe_kernel<8>, dim3(g), dim3(256), 0, h->stream, buf.as<uint4>(), n16, sink.as<uint32_t>()); },
               (double)nb, &r8);
    if (rc == AUVP_OK)
      rc = timed([&] { hipLaunchKernelGGL(hbm_read_probe_kernel<16>, dim3(g), dim3(256), 0, h->stream, buf.as<uint4>(), n16 / 4096ull * 4096ull, sink.as<uint32_t>()); },
                 (double)(n16 / 4096ull * 4096ull) * 16.0, &r16);
    best_read = std::max(best_read, std::max(r8, r16));
  }
  *read_GBps = best_read;
  if (rc != AUVP_OK) return rc;
  if (copy_GBps) {
    const unsigned long long half16 = (n16 / 2ull) / 1024ull * 1024ull;
    rc = timed([&] { hipLaunchKernelGGL(hbm_copy_probe_kernel, dim3(grid), dim3(256), 0, h->stream, buf.as<uint4>(), buf.as<uint4>() + half16, half16); },
               2.0 * 16.0 * (double)half16, copy_GBps);
    if (rc != AUVP_OK) return rc;
  }
  return AUVP_OK;
}

int auvp_sincos_dev(auvp_handle* h, int32_t n, const double* x, double* s, double* c) {
  if (!h || n < 0 || !x || !s || !c) return AUVP_ERR_ARG;
  if (n == 0) return AUVP_OK;
  HIPCHK(h, hipSetDevice(h->device));
  int rc;
  if ((rc = upload(h, h->d_tmp0, x, (size_t)n))) return rc;
  HIPCHK(h, h->d_tmp1.reserve((size_t)n * sizeof(double)));
  HIPCHK(h, h->d_tmp2.reserve((size_t)n * sizeof(double)));
  hipLaunchKernelGGL(sincos_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, (int)n, h->d_tmp0.as<double>(),
                     h->d_tmp1.as<double>(), h->d_tmp2.as<double>());
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(s, h->d_tmp1.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(c, h->d_tmp2.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return AUVP_OK;
}

int auvp_math_dev(auvp_handle* h, int32_t op, int32_t n, const double* a, const double* b, double* out0, double* out1) {
  if (!h || n < 0 || op < 0 || op > 7 || !a || !b || !out0 || !out1) return AUVP_ERR_ARG;
  if (n == 0) return AUVP_OK;
  HIPCHK(h, hipSetDevice(h->device));
  int rc;
  if ((rc = upload(h, h->d_tmp0, a, (size_t)n))) return rc;
  if ((rc = upload(h, h->d_tmp1, b, (size_t)n))) return rc;
  HIPCHK(h, h->d_tmp2.reserve((size_t)2 * n * sizeof(double)));
  double* o = h->d_tmp2.as<double>();
  hipLaunchKernelGGL(math_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, (int)op, (int)n, h->d_tmp0.as<double>(),
                     h->d_tmp1.as<double>(), o, o + n);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(out0, o, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(out1, o + n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return AUVP_OK;
}

int auvp_random_stream_dev(auvp_handle* h, uint64_t seed, int32_t n, double* out) {
  if (!h || n < 0 || !out) return AUVP_ERR_ARG;
  if (n == 0) return AUVP_OK;
  HIPCHK(h, hipSetDevice(h->device));
  uint32_t mt[624];
  seed_mt(seed, mt);
  int rc;
  if ((rc = upload(h, h->d_tmp0, mt, 624))) return rc;
  HIPCHK(h, h->d_tmp1.reserve((size_t)n * sizeof(double)));
  hipLaunchKernelGGL(random_probe_kernel, dim3(1), dim3(64), 0, h->stream, h->d_tmp0.as<uint32_t>(), (int)n, h->d_tmp1.as<double>());
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(out, h->d_tmp1.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return AUVP_OK;
}

}  // extern "C"
#endif
