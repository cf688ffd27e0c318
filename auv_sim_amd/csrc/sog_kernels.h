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
  int32_t n_cells, n_sharks, n_bins, rows, cols, count, n_pts, n_xbuckets;
  double bin_interval;
  const double* pts;        // [n_pts,3] x, y, t
  const int32_t* pt_shark;  // [n_pts]
  const int32_t* cell_rc;   // [C] row*cols + col of each listed cell
  const int32_t* mult;      // [G] how many listed cells map to the grid cell
  // x-bucket index over the cells (closed containment, first match in cell_list order)
  const int32_t* xb_off;
  const int32_t* xb_items;
  const double* xb_data;    // per item: minx, maxx, miny, maxy
  const double* xb_sufmin;  // per item: min miny over this and the later items of the bucket
  double xb_x0, xb_inv_w;
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
  atomicAdd(&D.npts[b * D.n_sharks + s], 1);
  if (D.n_cells == 0) return;
  double fb = auvp_floor((x - D.xb_x0) * D.xb_inv_w);
  int bk = fb < 0.0 ? 0 : (fb >= (double)D.n_xbuckets ? D.n_xbuckets - 1 : (int)fb);
  const int e = D.xb_off[bk + 1];
  for (int i = D.xb_off[bk]; i < e; i++) {
    if (y < D.xb_sufmin[i]) break;
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
  // x-bucket index, closed containment
  std::vector<int32_t> xoff(2, 0), xitems;
  std::vector<double> xdata, xsuf;
  double X0 = 0.0, inv_w = 0.0;
  int NB = 1;
  if (C > 0) {
    double lo = INFINITY, hi = -INFINITY, wmin = INFINITY;
    for (int c = 0; c < C; c++) {
      lo = std::min(lo, cells[4 * c]); hi = std::max(hi, cells[4 * c + 2]);
      const double wdt = cells[4 * c + 2] - cells[4 * c];
      if (wdt > 0 && wdt < wmin) wmin = wdt;
    }
    const double span = hi - lo;
    if (span > 0 && std::isfinite(wmin)) NB = (int)std::min(8192.0, std::max(1.0, std::ceil(span / wmin)));
    X0 = lo; inv_w = span > 0 ? (double)NB / span : 0.0;
    std::vector<std::vector<int32_t>> lists(NB);
    for (int c = 0; c < C; c++) {
      // the device computes the same floor((x - X0) * inv_w), clamped: a monotone map, so every x in [minx, maxx]
      // lands in a bucket between those of the two ends
      int b0 = (int)std::floor((cells[4 * c] - X0) * inv_w), b1 = (int)std::floor((cells[4 * c + 2] - X0) * inv_w);
      b0 = std::max(0, std::min(NB - 1, b0)); b1 = std::max(0, std::min(NB - 1, b1));
      for (int k = b0; k <= b1; k++) lists[k].push_back(c);
    }
    xoff.assign(NB + 1, 0);
    for (int k = 0; k < NB; k++) { xoff[k + 1] = xoff[k] + (int32_t)lists[k].size(); xitems.insert(xitems.end(), lists[k].begin(), lists[k].end()); }
    xdata.resize(xitems.size() * 4); xsuf.resize(xitems.size());
    for (int k = 0; k < NB; k++) {
      double suf = INFINITY;
      for (int i = xoff[k + 1] - 1; i >= xoff[k]; i--) {
        const double* cb = cells + 4 * (size_t)xitems[i];
        suf = std::min(suf, cb[1]);
        xdata[4 * (size_t)i] = cb[0]; xdata[4 * (size_t)i + 1] = cb[2]; xdata[4 * (size_t)i + 2] = cb[1]; xdata[4 * (size_t)i + 3] = cb[3];
        xsuf[i] = suf;
      }
    }
  }
  DevBuf d_pts, d_ps, d_rc, d_mult, d_xoff, d_xit, d_xd, d_xs, d_cnt, d_np, d_occ, d_grid;
  int rc;
  if ((rc = upload(h, d_pts, pts, (size_t)n_pts * 3))) return rc;
  if ((rc = upload(h, d_ps, pt_shark.data(), pt_shark.size()))) return rc;
  if ((rc = upload(h, d_rc, cell_rc.data(), cell_rc.size()))) return rc;
  if ((rc = upload(h, d_mult, mult.data(), mult.size()))) return rc;
  if ((rc = upload(h, d_xoff, xoff.data(), xoff.size()))) return rc;
  if ((rc = upload(h, d_xit, xitems.data(), xitems.size()))) return rc;
  if ((rc = upload(h, d_xd, xdata.data(), xdata.size()))) return rc;
  if ((rc = upload(h, d_xs, xsuf.data(), xsuf.size()))) return rc;
  const size_t tsg = (size_t)T * S * G;
  HIPCHK(h, d_cnt.reserve(tsg * sizeof(int32_t)));
  HIPCHK(h, d_np.reserve((size_t)T * S * sizeof(int32_t)));
  HIPCHK(h, d_occ.reserve(tsg * sizeof(double)));
  HIPCHK(h, d_grid.reserve((size_t)T * G * sizeof(double)));
  HIPCHK(h, hipMemsetAsync(d_cnt.p, 0, tsg * sizeof(int32_t), h->stream));
  HIPCHK(h, hipMemsetAsync(d_np.p, 0, (size_t)T * S * sizeof(int32_t), h->stream));
  auvp::SogDev D{};
  D.n_cells = C; D.n_sharks = S; D.n_bins = T; D.rows = rows; D.cols = cols; D.count = (int)std::ceil(detect_range / cell_size);
  D.n_pts = n_pts; D.n_xbuckets = NB; D.bin_interval = bin_interval;
  D.pts = d_pts.as<double>(); D.pt_shark = d_ps.as<int32_t>(); D.cell_rc = d_rc.as<int32_t>(); D.mult = d_mult.as<int32_t>();
  D.xb_off = d_xoff.as<int32_t>(); D.xb_items = d_xit.as<int32_t>(); D.xb_data = d_xd.as<double>(); D.xb_sufmin = d_xs.as<double>();
  D.xb_x0 = X0; D.xb_inv_w = inv_w;
  D.counts = d_cnt.as<int32_t>(); D.npts = d_np.as<int32_t>(); D.occ = d_occ.as<double>(); D.grids = d_grid.as<double>();
  HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  if (n_pts) hipLaunchKernelGGL(auvp::sog_count_kernel, dim3((n_pts + 255) / 256), dim3(256), 0, h->stream, D);
  hipLaunchKernelGGL(auvp::sog_occ_kernel, dim3((unsigned)((tsg + 255) / 256)), dim3(256), 0, h->stream, D);
  hipLaunchKernelGGL(auvp::sog_grid_kernel, dim3((unsigned)(((size_t)T * G + 255) / 256)), dim3(256), 0, h->stream, D);
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
