// sog_kernels.h -- SharkOccupancyGrid.convert (path_planning/sharkOccupancyGrid.py:47-74) on gfx950:
// per time bin, the AUV-detection grid = mean over sharks of the disc-stencil sum of each shark's
// occupancy histogram (constructSharkOccupancyGrid :205-241, constructAUVGrid :174-203, constructGrid
// :145-172).  Three elementwise passes, every sum in the reference's order (bit-identical):
//   sog_count_kernel   thread = trajectory point: first time bin that contains it (:243-256), first
//                      cell in cell_list order that contains it (closed rectangle, :264), atomic count
//   sog_occ_kernel     thread = (bin, shark, grid cell): 0.01 prior + count unit adds, / normaliser
//   sog_grid_kernel    thread = (bin, grid cell): per shark the window sum in (i, j) order, summed over
//                      sharks in dict order, / number of sharks
// Included at the end of auvplan.hip.
#ifndef AUVP_SOG_KERNELS_H
#define AUVP_SOG_KERNELS_H

namespace auvp {

struct SogDev {
  int32_t n_cells, n_sharks, n_bins, rows, cols, count, n_pts, n_xbuckets, n_ybuckets;
  double bin_interval;
  const double* pts;        // [n_pts,3] x, y, t
  const int32_t* pt_shark;  // [n_pts]
  const int32_t* cell_rc;   // [C] row*cols + col of each listed cell
  const int32_t* mult;      // [G] how many listed cells map to the grid cell
  // (x, y)-bucket index over the cells (closed containment, first match in cell_list order)
  const int32_t* xb_off;
  const int32_t* xb_items;
  const double* xb_data;    // per item: minx, maxx, miny, maxy
  double xb_x0, xb_inv_w, xb_y0, xb_inv_h;
  int32_t* counts;          // [T][S][G]
  int32_t* npts;            // [T][S]
  double* occ;              // [T][S][G]
  double* grids;            // [T][G]
};

__global__ __launch_bounds__(256) void sog_count_kernel(SogDev D) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= D.n_pts) return;
  const double x = D.pts[3 * (size_t)k], y = D.pts[3 * (size_t)k + 1], t = D.pts[3 * (size_t)k + 2];
  int b = -1;
  for (int q = 0; q < D.n_bins; q++) {
    // bins are (q*bin_interval, (q+1)*bin_interval); a point on a shared edge goes to the earlier bin
    if (t >= q * D.bin_interval && t <= (q + 1) * D.bin_interval) { b = q; break; }
  }
  if (b < 0) return;
  const int s = D.pt_shark[k];
  const size_t G = (size_t)D.rows * D.cols;
  // len(traj) per (bin, shark): a trajectory's points are consecutive and sorted by time, so a wavefront holds one or two
  // keys -- one atomic per key and wavefront instead of one per point on the same address
  {
    const int key = b * D.n_sharks + s, lane = threadIdx.x & 63;
    unsigned long long todo = wave_ballot(1);
    while (todo) {
      const int leader = __ffsll(todo) - 1;
      const int k0 = __shfl(key, leader);
      const unsigned long long same = wave_ballot(key == k0) & todo;
      if (lane == leader) atomicAdd(&D.npts[k0], __popcll(same));
      todo &= ~same;
      if (key == k0) break;
    }
  }
  if (D.n_cells == 0) return;
  double fb = auvp_floor((x - D.xb_x0) * D.xb_inv_w), fy = auvp_floor((y - D.xb_y0) * D.xb_inv_h);
  const int bx = fb < 0.0 ? 0 : (fb >= (double)D.n_xbuckets ? D.n_xbuckets - 1 : (int)fb);
  const int by = fy < 0.0 ? 0 : (fy >= (double)D.n_ybuckets ? D.n_ybuckets - 1 : (int)fy);
  const int bk = by * D.n_xbuckets + bx;
  const int e = D.xb_off[bk + 1];
  for (int i = D.xb_off[bk]; i < e; i++) {
    const double4 d = reinterpret_cast<const double4*>(D.xb_data)[i];
    if (x >= d.x && x <= d.y && y >= d.z && y <= d.w) {
      atomicAdd(&D.counts[((size_t)b * D.n_sharks + s) * G + D.cell_rc[D.xb_items[i]]], 1);
      break;
    }
  }
}

__global__ __launch_bounds__(256) void sog_occ_kernel(SogDev D) {
  const size_t G = (size_t)D.rows * D.cols;
  const size_t total = (size_t)D.n_bins * D.n_sharks * G;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const size_t ts = i / G, g = i % G;
  double v = D.mult[g] > 0 ? 0.01 : 0.0;              // grid[row][col] = 0.01 for listed cells (:255-257)
  const int c = D.counts[i];
  for (int k = 0; k < c; k++) v = v + 1;              // grid[row][col] += 1 per point (:266)
  const double nor = ((double)D.npts[ts] + D.n_cells * 0.01);  // len(traj) + len(cell_list) * 0.01 (:260)
  D.occ[i] = v / nor;
}

__global__ __launch_bounds__(256) void sog_grid_kernel(SogDev D) {
  const size_t G = (size_t)D.rows * D.cols;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)D.n_bins * G) return;
  const int t = (int)(i / G), g = (int)(i % G), row = g / D.cols, col = g % D.cols;
  const int count = D.count, m = D.mult[g];
  const long long c2 = (long long)count * count;
  // The reference scans the 4count x 4count window [row-2count, row+2count) x [col-2count, col+2count) in (i, j)
  // order and keeps d2 <= count^2: only rows |dr| <= count contribute, each over |dc| <= floor(sqrt(count^2 - dr^2)),
  // all of which lie inside the window (count < 2 count).  Visiting exactly those cells in the same order gives the
  // same sequence of additions.
  const int r_lo = max(row - count, 0), r_hi = min(row + count, D.rows - 1);
  double total = 0.0;
  for (int s = 0; s < D.n_sharks; s++) {
    const double* occ = D.occ + ((size_t)t * D.n_sharks + s) * G;
    double a = 0.0;
    for (int rep = 0; rep < (count > 0 ? m : 0); rep++) {  // once per listed cell that maps here (constructAUVGrid's loop
                                                            // over cell_list); count == 0: the window is empty
      for (int rt = r_lo; rt <= r_hi; rt++) {
        const long long rem = c2 - (long long)(rt - row) * (rt - row);
        int w = (int)sqrt((double)rem);  // integer square root, corrected for the rounding of the cast
        while ((long long)w * w > rem) w--;
        while ((long long)(w + 1) * (w + 1) <= rem) w++;
        const int c_lo = max(col - w, 0), c_hi = min(col + w, D.cols - 1);
        const double* orow = occ + (size_t)rt * D.cols;
        for (int ct = c_lo; ct <= c_hi; ct++) a = a + orow[ct];
      }
    }
    total = total + a;  // grid[i][j] + tempAUVGrid[i][j], sharks in dict order (:163-167)
  }
  D.grids[i] = total / D.n_sharks;
}

// The same sums from LDS (round 4): a 256-thread workgroup owns a 16-row x 64-column tile of one time bin, a thread four
// adjacent cells of a row.  Per shark the tile plus a halo of `count` cells is staged in LDS (zeros outside the grid: a + 0.0
// == a for the non-negative occupancies, so padding equals the reference's clipping), double buffered across the sharks; a
// thread walks a window row once with a four-value register window -- one LDS read per four additions -- keeping each cell's
// additions in the reference's (row, column) order.  Tile column j is stored at (j % 4) * Q + j / 4: the 16 threads of a
// tile row read consecutive doubles, and a thread's offset within its row is the same scalar for the whole wavefront.
#define SOG_TH 16
#define SOG_THREADS (16 * SOG_TH)
#define SOG_TW 64
__host__ __device__ constexpr int sog_tile_q(int count) { return (((SOG_TW + 2 * count + 1 + 3) / 4 + 7) / 8) * 8 + 4; }  // Q = 4 mod 8
__host__ __device__ inline size_t sog_tile_bytes(int count) { return (size_t)2 * (SOG_TH + 2 * count) * 4 * sog_tile_q(count) * sizeof(double); }

__global__ __launch_bounds__(SOG_THREADS) void sog_grid_tile_kernel(SogDev D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sog_smem[];
  const int count = D.count, Q = sog_tile_q(count), RS = 4 * Q, H = SOG_TH + 2 * count, P = SOG_TW + 2 * count + 1;
  double* buf0 = reinterpret_cast<double*>(sog_smem);
  double* buf1 = buf0 + (size_t)H * RS;
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int t = blockIdx.z, row0 = blockIdx.y * SOG_TH, col0 = blockIdx.x * SOG_TW;
  const size_t G = (size_t)D.rows * D.cols;
  const int row = row0 + ty, col = col0 + 4 * tx;
  int m[4];
#pragma unroll
  for (int k = 0; k < 4; k++) m[k] = (row < D.rows && col + k < D.cols) ? D.mult[(size_t)row * D.cols + col + k] : 1;
  const bool plain = __syncthreads_and(m[0] == 1 && m[1] == 1 && m[2] == 1 && m[3] == 1);
  int max_m = max(max(m[0], m[1]), max(m[2], m[3]));
  const long long c2 = (long long)count * count;
  auto stage = [&](double* buf, int s) {
    const double* occ = D.occ + ((size_t)t * D.n_sharks + s) * G;
    for (int idx = tid; idx < H * P; idx += SOG_THREADS) {
      const int r = idx / P, j = idx - r * P;
      const int gr = row0 - count + r, gc = col0 - count + j;
      const double v = (gr >= 0 && gr < D.rows && gc >= 0 && gc < D.cols) ? occ[(size_t)gr * D.cols + gc] : 0.0;
      buf[(size_t)r * RS + (j & 3) * Q + (j >> 2)] = v;
    }
  };
  double total[4] = {0.0, 0.0, 0.0, 0.0};
  stage(buf0, 0);
  __syncthreads();
  for (int s = 0; s < D.n_sharks; s++) {
    double* cur = (s & 1) ? buf1 : buf0;
    if (s + 1 < D.n_sharks) stage((s & 1) ? buf0 : buf1, s + 1);
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    const int reps = plain ? 1 : max_m;
    for (int rep = 0; rep < reps; rep++) {
      for (int dr = -count; dr <= count; dr++) {
        const long long rem = c2 - (long long)dr * dr;
        int w = (int)sqrt((double)rem);  // integer square root, corrected for the rounding of the cast
        while ((long long)w * w > rem) w--;
        while ((long long)(w + 1) * (w + 1) <= rem) w++;
        const double* lrow = cur + (size_t)(ty + count + dr) * RS + tx;
        const int o0 = count - w;  // tile column of (col - w) is 4 tx + o0
        auto ld = [&](int o) { return lrow[(o & 3) * Q + (o >> 2)]; };
        double v0 = ld(o0), v1 = ld(o0 + 1), v2 = ld(o0 + 2), v3 = ld(o0 + 3);
        if (plain) {
          for (int i = 0; i <= 2 * w; i++) {
            a[0] = a[0] + v0; a[1] = a[1] + v1; a[2] = a[2] + v2; a[3] = a[3] + v3;
            v0 = v1; v1 = v2; v2 = v3; v3 = ld(o0 + i + 4);
          }
        } else {  // a grid cell listed m times takes its window m times (constructAUVGrid loops over cell_list); m = 0: none
          for (int i = 0; i <= 2 * w; i++) {
            a[0] = rep < m[0] ? a[0] + v0 : a[0]; a[1] = rep < m[1] ? a[1] + v1 : a[1];
            a[2] = rep < m[2] ? a[2] + v2 : a[2]; a[3] = rep < m[3] ? a[3] + v3 : a[3];
            v0 = v1; v1 = v2; v2 = v3; v3 = ld(o0 + i + 4);
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) total[k] = total[k] + a[k];  // grid[i][j] + tempAUVGrid[i][j], sharks in dict order (:163-167)
    __syncthreads();
  }
  if (row < D.rows) {
#pragma unroll
    for (int k = 0; k < 4; k++) if (col + k < D.cols) D.grids[(size_t)t * G + (size_t)row * D.cols + col + k] = total[k] / D.n_sharks;
  }
}

// The tiled kernel with the window radius as a template parameter (counts 1..8: detect ranges up to eight cells): the row
// half-widths, the LDS pitch and every LDS offset are compile-time constants, a window row is read into registers once
// (2 w + 4 values for the thread's four cells) and added from there; the next shark's tile is fetched from memory into
// registers before this shark's sums start and written to the other LDS buffer after them.
template <int C>
struct SogDisc {
  int w[2 * C + 1];
  constexpr SogDisc() : w() {
    for (int dr = -C; dr <= C; dr++) {
      int x = 0;
      while ((x + 1) * (x + 1) <= C * C - dr * dr) x++;
      w[dr + C] = x;
    }
  }
};

template <int C>
__global__ __launch_bounds__(SOG_THREADS, C <= 5 ? 3 : 2) void sog_grid_tile_c_kernel(SogDev D) {  // (workgroups per CU LDS admits / registers)
  extern __shared__ __attribute__((aligned(16))) unsigned char sog_smem[];
  constexpr int Q = sog_tile_q(C), RS = 4 * Q, H = SOG_TH + 2 * C, P = SOG_TW + 2 * C + 1, NST = (H * P + SOG_THREADS - 1) / SOG_THREADS;
  constexpr SogDisc<C> disc{};
  double* buf0 = reinterpret_cast<double*>(sog_smem);
  double* buf1 = buf0 + (size_t)H * RS;
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int t = blockIdx.z, row0 = blockIdx.y * SOG_TH, col0 = blockIdx.x * SOG_TW;
  const size_t G = (size_t)D.rows * D.cols;
  const int row = row0 + ty, col = col0 + 4 * tx;
  int m[4];
#pragma unroll
  for (int k = 0; k < 4; k++) m[k] = (row < D.rows && col + k < D.cols) ? D.mult[(size_t)row * D.cols + col + k] : 1;
  const bool plain = __syncthreads_and(m[0] == 1 && m[1] == 1 && m[2] == 1 && m[3] == 1);
  const int max_m = max(max(m[0], m[1]), max(m[2], m[3]));
  // where this thread's staged elements come from and go to (the same for every shark)
  int src[NST], dst[NST];
#pragma unroll
  for (int q = 0; q < NST; q++) {
    const int idx = tid + SOG_THREADS * q, r = idx / P, j = idx - r * P;
    const int gr = row0 - C + r, gc = col0 - C + j;
    src[q] = (idx < H * P && gr >= 0 && gr < D.rows && gc >= 0 && gc < D.cols) ? gr * D.cols + gc : -1;
    dst[q] = idx < H * P ? r * RS + (j & 3) * Q + (j >> 2) : -1;
  }
  double sv[NST];
  auto fetch = [&](int s) {
    const double* occ = D.occ + ((size_t)t * D.n_sharks + s) * G;
#pragma unroll
    for (int q = 0; q < NST; q++) sv[q] = src[q] >= 0 ? occ[src[q]] : 0.0;
  };
  auto put = [&](double* buf) {
#pragma unroll
    for (int q = 0; q < NST; q++) if (dst[q] >= 0) buf[dst[q]] = sv[q];
  };
  double total[4] = {0.0, 0.0, 0.0, 0.0};
  fetch(0);
  put(buf0);
  __syncthreads();
  for (int s = 0; s < D.n_sharks; s++) {
    const double* cur = (s & 1) ? buf1 : buf0;
    if (s + 1 < D.n_sharks) fetch(s + 1);
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    const double* lbase = cur + (size_t)(ty + C) * RS + tx;
    auto window = [&](auto plain_c, int rep) {
      auto load_row = [&](int dr, double (&v)[2 * C + 4]) {
        const int w = disc.w[dr + C];
        const double* lrow = lbase + dr * RS;
#pragma unroll
        for (int i = 0; i < 2 * C + 4; i++) {
          const int o = C - w + i;  // tile column of (col - w + i) is 4 tx + o
          if (i < 2 * w + 4) v[i] = lrow[(o & 3) * Q + (o >> 2)];
        }
      };
      double vn[2 * C + 4];
      load_row(-C, vn);
#pragma unroll
      for (int dr = -C; dr <= C; dr++) {
        const int w = disc.w[dr + C];
        double v[2 * C + 4];
#pragma unroll
        for (int i = 0; i < 2 * C + 4; i++) v[i] = vn[i];
        if (dr < C) load_row(dr + 1, vn);  // the next row's reads are in flight behind this row's additions; the fence keeps
                                           // the compiler from hoisting every row's reads to the top (299 registers)
#pragma unroll
        for (int i = 0; i < 2 * C + 1; i++) {
          if (i <= 2 * w) {
            if (decltype(plain_c)::value) {
              a[0] = a[0] + v[i]; a[1] = a[1] + v[i + 1]; a[2] = a[2] + v[i + 2]; a[3] = a[3] + v[i + 3];
            } else {  // a grid cell listed m times takes its window m times (constructAUVGrid loops over cell_list); m = 0: none
              a[0] = rep < m[0] ? a[0] + v[i] : a[0]; a[1] = rep < m[1] ? a[1] + v[i + 1] : a[1];
              a[2] = rep < m[2] ? a[2] + v[i + 2] : a[2]; a[3] = rep < m[3] ? a[3] + v[i + 3] : a[3];
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (plain) window(std::true_type{}, 0);
    else for (int rep = 0; rep < max_m; rep++) window(std::false_type{}, rep);
#pragma unroll
    for (int k = 0; k < 4; k++) total[k] = total[k] + a[k];  // grid[i][j] + tempAUVGrid[i][j], sharks in dict order (:163-167)
    if (s + 1 < D.n_sharks) put((s & 1) ? buf0 : buf1);
    __syncthreads();
  }
  if (row < D.rows) {
#pragma unroll
    for (int k = 0; k < 4; k++) if (col + k < D.cols) D.grids[(size_t)t * G + (size_t)row * D.cols + col + k] = total[k] / D.n_sharks;
  }
}

}  // namespace auvp

extern "C" int auvp_sog_convert(auvp_handle* h, const double* cells, int32_t C, const double* box, double cell_size,
                                double bin_interval, double detect_range, int32_t S, const int32_t* traj_len,
                                const double* pts, int32_t cap_bins, int32_t* n_bins, int32_t* rows_out, int32_t* cols_out,
                                double* bins, double* grids) {
  if (!h || !box || !traj_len || !n_bins || !rows_out || !cols_out || C < 0 || S <= 0 || !(cell_size > 0) || !(bin_interval > 0))
    return h ? fail(h, AUVP_ERR_ARG, "bad arguments") : AUVP_ERR_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  const double minx = box[0], miny = box[1], maxx = box[2], maxy = box[3];
  const int cols = (int)(std::ceil(maxx - minx) / cell_size) + 1, rows = (int)(std::ceil(maxy - miny) / cell_size) + 1;
  *rows_out = rows; *cols_out = cols;
  int n_pts = 0;
  double longest = 0;
  std::vector<int32_t> pt_shark;
  for (int s = 0; s < S; s++) {
    for (int k = 0; k < traj_len[s]; k++) pt_shark.push_back(s);
    n_pts += traj_len[s];
    if (traj_len[s] > 0 && pts[3 * (size_t)(n_pts - 1) + 2] > longest) longest = pts[3 * (size_t)(n_pts - 1) + 2];
  }
  const int T = (int)std::floor(longest / bin_interval);  // createBinList (:306-319)
  *n_bins = T;
  if (T > cap_bins) return fail(h, AUVP_ERR_CAPACITY, "%d time bins > capacity %d", T, cap_bins);
  for (int t = 0; t < T && bins; t++) { bins[2 * t] = t * bin_interval; bins[2 * t + 1] = (t + 1) * bin_interval; }
  if (T == 0) return AUVP_OK;
  const size_t G = (size_t)rows * cols;
  std::vector<int32_t> cell_rc(C > 0 ? C : 1), mult(G, 0);
  for (int c = 0; c < C; c++) {
    int col = (int)((cells[4 * c] - minx) / cell_size), row = (int)((cells[4 * c + 1] - miny) / cell_size);  // cellToIndex (:294-299)
    if (col < 0) col += cols;
    if (row < 0) row += rows;
    if (col < 0 || col >= cols || row < 0 || row >= rows) return fail(h, AUVP_ERR_ARG, "cell %d outside the grid (IndexError)", c);
    cell_rc[c] = row * cols + col;
    mult[cell_rc[c]]++;
  }
  // (x, y)-bucket index, closed containment: a bucket lists, in cell_list order, the cells whose closed rectangle can hold a
  // point of the bucket (round 4: the x-only buckets of round 1 held a whole column of cells, ~200 dependent reads per point)
  std::vector<int32_t> xoff(2, 0), xitems;
  std::vector<double> xdata;
  double X0 = 0.0, inv_w = 0.0, Y0 = 0.0, inv_h = 0.0;
  int NBX = 1, NBY = 1;
  if (C > 0) {
    double lo = INFINITY, hi = -INFINITY, wmin = INFINITY, ylo = INFINITY, yhi = -INFINITY, hmin = INFINITY;
    for (int c = 0; c < C; c++) {
      lo = std::min(lo, cells[4 * c]); hi = std::max(hi, cells[4 * c + 2]);
      ylo = std::min(ylo, cells[4 * c + 1]); yhi = std::max(yhi, cells[4 * c + 3]);
      const double wdt = cells[4 * c + 2] - cells[4 * c], hgt = cells[4 * c + 3] - cells[4 * c + 1];
      if (wdt > 0 && wdt < wmin) wmin = wdt;
      if (hgt > 0 && hgt < hmin) hmin = hgt;
    }
    const double span = hi - lo, yspan = yhi - ylo;
    if (span > 0 && std::isfinite(wmin)) NBX = (int)std::min(1024.0, std::max(1.0, std::ceil(span / wmin)));
    if (yspan > 0 && std::isfinite(hmin)) NBY = (int)std::min(1024.0, std::max(1.0, std::ceil(yspan / hmin)));
    auto bucket = [](double v, double v0, double inv, int nb) {
      const int b = (int)std::floor((v - v0) * inv);
      return std::max(0, std::min(nb - 1, b));
    };
    // The bucket size comes from the SMALLEST cell, and a cell is entered into every bucket it overlaps: a list of mixed sizes
    // (one large cell beside many small ones, thin slivers) would make the table and its items O(C x NBX x NBY).  Bounded:
    // the table to max(4 096, 8 C) buckets, the items to 16 C + 4 096 -- the index is halved in both directions until it fits
    // (a coarser index lists more candidate cells per point, in the same order: same result).
    X0 = lo; Y0 = ylo;
    for (;;) {
      inv_w = span > 0 ? (double)NBX / span : 0.0;
      inv_h = yspan > 0 ? (double)NBY / yspan : 0.0;
      if (NBX == 1 && NBY == 1) break;
      bool fits = (size_t)NBX * NBY <= std::max<size_t>(4096, 8 * (size_t)C);
      if (fits) {
        const size_t budget = 16 * (size_t)C + 4096;
        size_t items = 0;
        for (int c = 0; c < C && items <= budget; c++) {
          const size_t nx = (size_t)(bucket(cells[4 * c + 2], X0, inv_w, NBX) - bucket(cells[4 * c], X0, inv_w, NBX) + 1);
          const size_t ny = (size_t)(bucket(cells[4 * c + 3], Y0, inv_h, NBY) - bucket(cells[4 * c + 1], Y0, inv_h, NBY) + 1);
          items += nx * ny;
        }
        fits = items <= budget;
      }
      if (fits) break;
      NBX = std::max(1, NBX / 2);
      NBY = std::max(1, NBY / 2);
    }
    // the device computes the same floor((v - v0) * inv), clamped: a monotone map, so every coordinate inside a cell lands in
    // a bucket between those of the cell's two ends
    std::vector<int32_t> cnt((size_t)NBX * NBY + 1, 0);
    for (int pass = 0; pass < 2; pass++) {
      for (int c = 0; c < C; c++) {
        const int bx0 = bucket(cells[4 * c], X0, inv_w, NBX), bx1 = bucket(cells[4 * c + 2], X0, inv_w, NBX);
        const int by0 = bucket(cells[4 * c + 1], Y0, inv_h, NBY), by1 = bucket(cells[4 * c + 3], Y0, inv_h, NBY);
        for (int by = by0; by <= by1; by++)
          for (int bx = bx0; bx <= bx1; bx++) {
            const size_t k = (size_t)by * NBX + bx;
            if (pass == 0) cnt[k + 1]++;
            else xitems[(size_t)xoff[k] + cnt[k]++] = c;  // cells visited in list order: a bucket's items stay in list order
          }
      }
      if (pass == 0) {
        xoff.assign((size_t)NBX * NBY + 1, 0);
        for (size_t k = 0; k < (size_t)NBX * NBY; k++) xoff[k + 1] = xoff[k] + cnt[k + 1];
        xitems.resize((size_t)xoff.back());
        std::fill(cnt.begin(), cnt.end(), 0);
      }
    }
    xdata.resize(xitems.size() * 4);
    for (size_t i = 0; i < xitems.size(); i++) {
      const double* cb = cells + 4 * (size_t)xitems[i];
      xdata[4 * i] = cb[0]; xdata[4 * i + 1] = cb[2]; xdata[4 * i + 2] = cb[1]; xdata[4 * i + 3] = cb[3];
    }
  }
  DevBuf d_pts, d_ps, d_rc, d_mult, d_xoff, d_xit, d_xd, d_cnt, d_np, d_occ, d_grid;
  int rc;
  if ((rc = upload(h, d_pts, pts, (size_t)n_pts * 3))) return rc;
  if ((rc = upload(h, d_ps, pt_shark.data(), pt_shark.size()))) return rc;
  if ((rc = upload(h, d_rc, cell_rc.data(), cell_rc.size()))) return rc;
  if ((rc = upload(h, d_mult, mult.data(), mult.size()))) return rc;
  if ((rc = upload(h, d_xoff, xoff.data(), xoff.size()))) return rc;
  if ((rc = upload(h, d_xit, xitems.data(), xitems.size()))) return rc;
  if ((rc = upload(h, d_xd, xdata.data(), xdata.size()))) return rc;
  const size_t tsg = (size_t)T * S * G;
  HIPCHK(h, d_cnt.reserve(tsg * sizeof(int32_t)));
  HIPCHK(h, d_np.reserve((size_t)T * S * sizeof(int32_t)));
  HIPCHK(h, d_occ.reserve(tsg * sizeof(double)));
  HIPCHK(h, d_grid.reserve((size_t)T * G * sizeof(double)));
  HIPCHK(h, hipMemsetAsync(d_cnt.p, 0, tsg * sizeof(int32_t), h->stream));
  HIPCHK(h, hipMemsetAsync(d_np.p, 0, (size_t)T * S * sizeof(int32_t), h->stream));
  auvp::SogDev D{};
  D.n_cells = C; D.n_sharks = S; D.n_bins = T; D.rows = rows; D.cols = cols; D.count = (int)std::ceil(detect_range / cell_size);
  D.n_pts = n_pts; D.n_xbuckets = NBX; D.n_ybuckets = NBY; D.bin_interval = bin_interval;
  D.pts = d_pts.as<double>(); D.pt_shark = d_ps.as<int32_t>(); D.cell_rc = d_rc.as<int32_t>(); D.mult = d_mult.as<int32_t>();
  D.xb_off = d_xoff.as<int32_t>(); D.xb_items = d_xit.as<int32_t>(); D.xb_data = d_xd.as<double>();
  D.xb_x0 = X0; D.xb_inv_w = inv_w; D.xb_y0 = Y0; D.xb_inv_h = inv_h;
  D.counts = d_cnt.as<int32_t>(); D.npts = d_np.as<int32_t>(); D.occ = d_occ.as<double>(); D.grids = d_grid.as<double>();
  HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  if (n_pts) hipLaunchKernelGGL(auvp::sog_count_kernel, dim3((n_pts + 255) / 256), dim3(256), 0, h->stream, D);
  hipLaunchKernelGGL(auvp::sog_occ_kernel, dim3((unsigned)((tsg + 255) / 256)), dim3(256), 0, h->stream, D);
  // the LDS-tiled sums where a tile with its halo fits (count <= 24: 150 KB), with the radius as a compile-time constant up to
  // eight cells; option SOG_TILE = 0 forces the per-cell sweep of L2, 2 the tiled kernel with the radius at run time
  const int tmode = (int)h->opt_num(OPT_SOG_TILE, 1);
  const size_t tile_lds = auvp::sog_tile_bytes(D.count);
  const bool tiled = D.count > 0 && tile_lds <= (size_t)150 * 1024 && tmode != 0;
  const dim3 tgrid((cols + SOG_TW - 1) / SOG_TW, (rows + SOG_TH - 1) / SOG_TH, T);
  auto launch_tile = [&](auto kern) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, tgrid, dim3(SOG_THREADS), tile_lds, h->stream, D);
    return hipSuccess;
  };
  if (tiled) {
    const int cc = tmode == 2 ? 0 : D.count;
    switch (cc) {
      case 1: HIPCHK(h, launch_tile(auvp::sog_grid_tile_c_kernel<1>)); break;
      case 2: HIPCHK(h, launch_tile(auvp::sog_grid_tile_c_kernel<2>)); break;
      case 3: HIPCHK(h, launch_tile(auvp::sog_grid_tile_c_kernel<3>)); break;
      case 4: HIPCHK(h, launch_tile(auvp::sog_grid_tile_c_kernel<4>)); break;
      case 5: HIPCHK(h, launch_tile(auvp::sog_grid_tile_c_kernel<5>)); break;
      case 6: HIPCHK(h, launch_tile(auvp::sog_grid_tile_c_kernel<6>)); break;
      case 7: HIPCHK(h, launch_tile(auvp::sog_grid_tile_c_kernel<7>)); break;
      case 8: HIPCHK(h, launch_tile(auvp::sog_grid_tile_c_kernel<8>)); break;
      default: HIPCHK(h, launch_tile(auvp::sog_grid_tile_kernel)); break;
    }
  } else {
    hipLaunchKernelGGL(auvp::sog_grid_kernel, dim3((unsigned)(((size_t)T * G + 255) / 256)), dim3(256), 0, h->stream, D);
  }
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipEventRecord(h->ev1, h->stream));
  if (grids) HIPCHK(h, hipMemcpyAsync(grids, d_grid.p, (size_t)T * G * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  float ms = 0.f;
  HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
  h->last_ms = ms;
  return AUVP_OK;
}
#endif
