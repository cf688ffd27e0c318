/* orc_sog.c -- CPU restatement of the shark-occupancy / AUV-detection grid construction.
 *
 * TEST INFRASTRUCTURE (oracle/): checker only (see orc_api.h).
 *
 * Follows path_planning/sharkOccupancyGrid.py (paths relative to /root/reference):
 *   SharkOccupancyGrid.convert                     :47-74
 *   createBinList                                  :306-319
 *   splitTraj / convertToTimeBin                   :243-279
 *   constructSharkOccupancyGrid                    :205-241
 *   constructAUVGrid                               :174-203
 *   constructGrid                                  :145-172
 *   cellToIndex                                    :294-299
 * Cells are axis-aligned rectangles (the stand-ins of tests/golden/_refstubs/install.py for the
 * shapely polygons `splitCell` produces, :376-393); `point.within(cell) or cell.touches(point)`
 * (:264) is closed containment.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "orc_api.h"

int orc_sog_convert(const orc_sog_in* in, orc_sog_out* o) {
  const int C = in->n_cells, S = in->n_sharks;
  const double minx = in->box[0], miny = in->box[1], maxx = in->box[2], maxy = in->box[3];
  const double cs = in->cell_size;
  /* grid shape (:134): int(ceil(maxx-minx) / cell_size) + 1 columns, likewise rows */
  const int cols = (int)(ceil(maxx - minx) / cs) + 1, rows = (int)(ceil(maxy - miny) / cs) + 1;
  o->rows = rows; o->cols = cols;
  /* createBinList: longest last time stamp over the sharks */
  double longest = 0;
  int off = 0;
  int* start = (int*)malloc(sizeof(int) * (size_t)(S + 1));
  for (int s = 0; s < S; s++) {
    start[s] = off;
    off += in->traj_len[s];
    if (in->traj_len[s] > 0) {
      double t = in->pts[3 * (size_t)(off - 1) + 2];
      if (t > longest) longest = t;
    }
  }
  start[S] = off;
  const int T = (int)floor(longest / in->bin_interval);
  o->n_bins = T;
  if (T > o->cap_bins) { free(start); return ORC_ERR_CAPACITY; }
  for (int t = 0; t < T; t++) { o->bins[2 * t] = t * in->bin_interval; o->bins[2 * t + 1] = (t + 1) * in->bin_interval; }
  int* crow = (int*)malloc(sizeof(int) * (size_t)(C > 0 ? C : 1));
  int* ccol = (int*)malloc(sizeof(int) * (size_t)(C > 0 ? C : 1));
  int status = ORC_OK;
  for (int c = 0; c < C; c++) {
    ccol[c] = (int)((in->cells[4 * c] - minx) / cs);
    crow[c] = (int)((in->cells[4 * c + 1] - miny) / cs);
    if (ccol[c] < 0) ccol[c] += cols;
    if (crow[c] < 0) crow[c] += rows;
    if (ccol[c] < 0 || ccol[c] >= cols || crow[c] < 0 || crow[c] >= rows) status = ORC_ERR_ARG; /* IndexError */
  }
  const size_t G = (size_t)rows * cols;
  double* occ = (double*)malloc(sizeof(double) * G);
  double* auv = (double*)malloc(sizeof(double) * G);
  const int count = (int)ceil(in->detect_range / cs);
  for (int t = 0; t < T && status == ORC_OK; t++) {
    double* grid = o->grids + (size_t)t * G;
    for (size_t i = 0; i < G; i++) grid[i] = 0.0;
    for (int s = 0; s < S; s++) {
      /* constructSharkOccupancyGrid over the points of shark s that fall into bin t (first matching bin) */
      for (size_t i = 0; i < G; i++) occ[i] = 0.0;
      for (int c = 0; c < C; c++) occ[(size_t)crow[c] * cols + ccol[c]] = 0.01;
      int npt = 0;
      for (int k = start[s]; k < start[s + 1]; k++) {
        const double x = in->pts[3 * (size_t)k], y = in->pts[3 * (size_t)k + 1], tm = in->pts[3 * (size_t)k + 2];
        int b = -1;
        for (int q = 0; q < T; q++) if (tm >= o->bins[2 * q] && tm <= o->bins[2 * q + 1]) { b = q; break; }
        if (b != t) continue;
        npt++;
        for (int c = 0; c < C; c++) {
          const double* cb = in->cells + 4 * (size_t)c;
          if (x >= cb[0] && x <= cb[2] && y >= cb[1] && y <= cb[3]) { occ[(size_t)crow[c] * cols + ccol[c]] += 1; break; }
        }
      }
      const double nor = (npt + C * 0.01);
      for (size_t i = 0; i < G; i++) occ[i] = occ[i] / nor;
      if (t == 0 && s == 0 && o->occ_dbg) memcpy(o->occ_dbg, occ, sizeof(double) * G);
      /* constructAUVGrid: disc of `count` cells around every listed cell, summed in window order */
      for (size_t i = 0; i < G; i++) auv[i] = 0.0;
      for (int c = 0; c < C; c++) {
        const int row = crow[c], col = ccol[c];
        const int row_min = row - 2 * count, col_min = col - 2 * count;
        for (int i = 0; i < 2 * (2 * count); i++) {
          const int rt = row_min + i;
          for (int j = 0; j < 2 * 2 * count; j++) {
            const int ct = col_min + j;
            if (rt >= 0 && rt < rows && ct >= 0 && ct < cols) {
              const long long d2 = (long long)(rt - row) * (rt - row) + (long long)(ct - col) * (ct - col);
              if (d2 <= (long long)count * count) auv[(size_t)row * cols + col] += occ[(size_t)rt * cols + ct];
            }
          }
        }
      }
      if (t == 0 && s == 0 && o->auv_dbg) memcpy(o->auv_dbg, auv, sizeof(double) * G);
      for (size_t i = 0; i < G; i++) grid[i] = grid[i] + auv[i];
    }
    for (size_t i = 0; i < G; i++) grid[i] = grid[i] / S;
  }
  free(start); free(crow); free(ccol); free(occ); free(auv);
  return status;
}
