/* orc_astar.c -- CPU restatement of the four runnable A* variants.
 *
 * TEST INFRASTRUCTURE (oracle/): checker + reported CPU baseline only (see orc_api.h).
 *
 * Follows (paths relative to /root/reference/path_planning):
 *   variant 0  astar.py           astar.astar :193-271, curr_neighbors :79-100, check_collision :60-77
 *   variant 1  astar_real.py      astar.astar :144-222, check_boundary :77-95, near_goal :137-142
 *   variant 2  astar_fixLen.py    astar.astar :286-418, within_bounds :96-118, update_habitat_coverage :182-199,
 *                                 get_indices :272-284; Cost.cost_of_edge  ../cost.py:66-101
 *   variant 3  astar_fixLenSOG.py astar.astar :551-657, findCurrSOG :442-459, get_cell_prob :485-514,
 *                                 get_top_n_prob :516-533, smoothPath :404-440, Walkable :218-242,
 *                                 inside_habitats :304-320
 * Shared quirks reproduced: Node has no __eq__, so `child in closed_list` / `child ==
 * open_node` never fire (no dedup except the visited bitmap of variants 2/3); pop = first minimum f;
 * fixLen's h uses child.pathLen == 0; update_habitat_coverage pops while enumerating.
 * Polygon(...).centroid (shapely, absent) = area-weighted shoelace centroid of the stub
 * (tests/golden/_refstubs/install.py) -- parity unpinned at that boundary.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "orc_api.h"
#include "orc_math.h"

typedef struct {
  double x, y, g, h, f, cost, pathLen;
  double time_stamp;
  int parent, open;
} anode;

static const int NB_A[8][2] = {{0, -10}, {0, 10}, {-10, 0}, {10, 0}, {10, 10}, {10, -10}, {-10, 10}, {-10, -10}};
static const int NB_B[8][2] = {{0, -10}, {0, 10}, {-10, 0}, {10, 0}, {-10, -10}, {-10, 10}, {10, -10}, {10, 10}};

static double sqdist(double ax, double ay, double bx, double by) {
  double dx = fabs(ax - bx), dy = fabs(ay - by);
  return dx * dx + dy * dy;
}

/* same_side (astar_real.py:62-68): np.cross of 2-vectors is a0*b1 - a1*b0 */
static int same_side(double p1x, double p1y, double p2x, double p2y, double ax, double ay, double bx, double by) {
  double ex = bx - ax, ey = by - ay;
  double cp1 = ex * (p1y - ay) - ey * (p1x - ax);
  double cp2 = ex * (p2y - ay) - ey * (p2x - ax);
  return cp1 * cp2 >= 0;
}

static int in_triangle(double px, double py, double ax, double ay, double bx, double by, double cx, double cy) {
  return same_side(px, py, ax, ay, bx, by, cx, cy) && same_side(px, py, bx, by, ax, ay, cx, cy) &&
         same_side(px, py, cx, cy, ax, ay, bx, by);
}

static void centroid(const double* poly, int nv, double* cx, double* cy) {
  double a2 = 0.0, sx = 0.0, sy = 0.0;
  for (int i = 0; i < nv; i++) {
    double x0 = poly[2 * i], y0 = poly[2 * i + 1];
    double x1 = poly[2 * ((i + 1) % nv)], y1 = poly[2 * ((i + 1) % nv) + 1];
    double cr = x0 * y1 - x1 * y0;
    a2 += cr;
    sx += (x0 + x1) * cr;
    sy += (y0 + y1) * cr;
  }
  *cx = sx / (3.0 * a2);
  *cy = sy / (3.0 * a2);
}

static int within_bounds(const double* poly, int nv, double cx, double cy, double px, double py) {
  for (int i = 0; i < nv; i++) {
    int j = (i != nv - 1) ? i + 1 : 0;
    if (in_triangle(px, py, poly[2 * i], poly[2 * i + 1], poly[2 * j], poly[2 * j + 1], cx, cy)) return 1;
  }
  return 0;
}

static int collision_free(const orc_world* w, double px, double py) {
  for (int k = 0; k < w->n_obstacles; k++) {
    double dx = px - w->obstacles[3 * k], dy = py - w->obstacles[3 * k + 1];
    double d = ORC_SQRT(ORC_POW2(dx) + ORC_POW2(dy));
    if (d <= w->obstacles[3 * k + 2]) return 0;
  }
  return 1;
}

static int hab_covers(const double* h, double px, double py) {
  double d = ORC_SQRT(ORC_POW2(px - h[0]) + ORC_POW2(py - h[1]));
  return d <= h[2];
}

/* Python round(v, 2): correctly rounded decimal rounding of the exact binary value */
static double round2(double v) {
  char buf[512];
  snprintf(buf, sizeof buf, "%.2f", v);
  return strtod(buf, NULL);
}

/* numpy index with negative wrap; -1 = IndexError */
static int np_index(double v, int n) {
  int i = (int)v;
  if (i < 0) i += n;
  return (i < 0 || i >= n) ? -1 : i;
}

static int cmp_desc(const void* a, const void* b) {
  double x = *(const double*)a, y = *(const double*)b;
  return (x < y) - (x > y);
}

int orc_astar_run(const orc_world* w, const orc_astar_params* p, orc_astar_out* o) {
  const int V = p->variant;
  const int cap = p->cap_nodes;
  anode* N = (anode*)calloc((size_t)cap, sizeof(anode));
  int n_nodes = 0, n_exp = 0, n_children = 0, status = ORC_OK, found = -1;
  double cx = 0, cy = 0;
  if (V >= 1) centroid(w->polygon, w->n_poly, &cx, &cy);
  /* habitat open / closed lists as index arrays (variants 2, 3) */
  const int H = w->n_habitats;
  int* hopen = (int*)malloc(sizeof(int) * (size_t)(H + 1));
  int* hclosed = (int*)malloc(sizeof(int) * (size_t)(H + 1));
  int n_open = H, n_closed = 0;
  for (int i = 0; i < H; i++) hopen[i] = i;
  /* visited bitmap */
  const int VX = (V == 2) ? 550 : 600, VY = 600;
  unsigned char* visited = (V >= 2) ? (unsigned char*)calloc((size_t)VX * VY, 1) : NULL;
  int visited_count = 0;
  /* SOG tables: rounded cell corners, descending prefix sums of the probabilities per bin */
  const int C = w->n_cells, T = w->n_bins;
  double* rc = NULL;
  double* topn = NULL;
  if (V == 3) {
    rc = (double*)malloc(sizeof(double) * 4 * (size_t)(C > 0 ? C : 1));
    for (int c = 0; c < 4 * C; c++) rc[c] = round2(w->cells[c]);
    topn = (double*)malloc(sizeof(double) * (size_t)T * (size_t)(C + 1));
    double* tmp = (double*)malloc(sizeof(double) * (size_t)(C > 0 ? C : 1));
    for (int t = 0; t < T; t++) {
      memcpy(tmp, w->prob + (size_t)t * C, sizeof(double) * (size_t)C);
      qsort(tmp, (size_t)C, sizeof(double), cmp_desc);
      double tot = 0.0;
      topn[(size_t)t * (C + 1)] = 0.0;
      for (int i = 0; i < C; i++) { tot += tmp[i]; topn[(size_t)t * (C + 1) + i + 1] = tot; }
    }
    free(tmp);
  }
  const double w2 = p->w[1], w3 = p->w[2], w4 = p->w[3];

  N[0].x = p->start[0]; N[0].y = p->start[1]; N[0].parent = -1; N[0].open = 1;
  n_nodes = 1;
  int n_open_nodes = 1;
  while (n_open_nodes > 0) {
    /* first minimum f in list order */
    int cur = -1;
    for (int i = 0; i < n_nodes; i++) {
      if (!N[i].open) continue;
      if (cur < 0 || N[i].f < N[cur].f) cur = i;
    }
    N[cur].open = 0;
    n_open_nodes--;
    int stop;
    if (V == 0) stop = (N[cur].x == p->goal[0] && N[cur].y == p->goal[1]);
    else if (V == 1) stop = sqdist(N[cur].x, N[cur].y, p->goal[0], p->goal[1]) <= 100;
    else stop = fabs(N[cur].pathLen - p->limit) <= 10;
    if (stop) { found = cur; break; }
    if (n_exp < o->cap_exp) {
      double* e = o->exp_log + 8 * (size_t)n_exp;
      e[0] = N[cur].x; e[1] = N[cur].y; e[2] = N[cur].g; e[3] = N[cur].h; e[4] = N[cur].f; e[5] = N[cur].cost;
      e[6] = N[cur].pathLen; e[7] = N[cur].time_stamp;
    }
    n_exp++;
    /* curr_neighbors + collision: children in neighbour order */
    int child[8], nch = 0;
    for (int k = 0; k < 8; k++) {
      const int* d = (V == 0) ? NB_A[k] : NB_B[k];
      double nx = N[cur].x + d[0], ny = N[cur].y + d[1];
      int inb;
      if (V == 0) inb = (nx >= p->box[0] && nx <= p->box[2]) && (ny >= p->box[1] && ny <= p->box[3]);
      else inb = within_bounds(w->polygon, w->n_poly, cx, cy, nx, ny);
      if (!inb) continue;
      n_children++;
      if (!collision_free(w, nx, ny)) continue;
      if (n_nodes >= cap) { status = ORC_ERR_CAPACITY; break; }
      int c = n_nodes++;
      memset(&N[c], 0, sizeof(anode));
      N[c].x = nx; N[c].y = ny; N[c].parent = cur;
      if (V == 2) {
        /* update_habitat_coverage (:182-199): pop while enumerating skips the next element */
        for (int idx = 0; idx < n_open; idx++) {
          if (hab_covers(w->habitats + 3 * hopen[idx], nx, ny)) {
            hclosed[n_closed++] = hopen[idx];
            memmove(hopen + idx, hopen + idx + 1, sizeof(int) * (size_t)(n_open - idx - 1));
            n_open--;
          }
        }
        /* cost_of_edge with the lists as they stand now (cost.py:66-101) */
        int d2 = 0, d3 = 0;
        for (int i = 0; i < n_open; i++) if (hab_covers(w->habitats + 3 * hopen[i], nx, ny)) d2 = 1;
        for (int i = 0; i < n_closed; i++) if (hab_covers(w->habitats + 3 * hclosed[i], nx, ny)) { d2 = 1; d3 = 1; }
        N[c].cost = N[cur].cost + (-w2 * d2 - w3 * d3);
      }
      child[nch++] = c;
    }
    if (status != ORC_OK) break;
    for (int q = 0; q < nch; q++) {
      anode* ch = &N[child[q]];
      if (V <= 1) {
        ch->g = N[cur].g + sqdist(ch->x, ch->y, N[cur].x, N[cur].y);
        ch->h = sqdist(ch->x, ch->y, p->goal[0], p->goal[1]);
        ch->f = ch->g + ch->h;
        ch->open = 1;
        n_open_nodes++;
        continue;
      }
      if (V == 2) {
        int d2 = 0, d3 = 0;
        for (int i = 0; i < n_open; i++) if (hab_covers(w->habitats + 3 * hopen[i], ch->x, ch->y)) d2 = 1;
        for (int i = 0; i < n_closed; i++) if (hab_covers(w->habitats + 3 * hclosed[i], ch->x, ch->y)) { d2 = 1; d3 = 1; }
        ch->g = N[cur].cost - w2 * d2 - w3 * d3;
        ch->cost = ch->g;
        ch->h = -w2 * fabs(p->limit - ch->pathLen) - w3 * n_open; /* pathLen is still 0 here (:401-403) */
        ch->f = ch->g + ch->h;
        ch->pathLen = N[cur].pathLen + sqrt(sqdist(N[cur].x, N[cur].y, ch->x, ch->y));
      } else {
        ch->pathLen = N[cur].pathLen + sqrt(sqdist(N[cur].x, N[cur].y, ch->x, ch->y));
        double dist_left = fabs(p->limit - ch->pathLen);
        ch->time_stamp = (double)(long long)(ch->pathLen / p->velocity);
        int tb = -1;
        for (int t = 0; t < T; t++) {
          if (ch->time_stamp <= w->bins[2 * t + 1] && ch->time_stamp >= w->bins[2 * t]) { tb = t; break; }
        }
        if (tb < 0) { status = ORC_ERR_ARG; break; } /* AttributeError on None in the reference */
        int key = -1;
        for (int c = 0; c < C; c++) {
          const double* r = rc + 4 * (size_t)c;
          double dx = fabs(r[0] - r[2]), dy = fabs(r[1] - r[3]);
          if ((fabs(ch->x - r[0]) <= dx && fabs(ch->x - r[2]) <= dx) && (fabs(ch->y - r[1]) <= dy && fabs(ch->y - r[3]) <= dy)) {
            key = c;
            break;
          }
        }
        if (key < 0) { status = ORC_ERR_ARG; break; } /* `w4 * None` TypeError */
        int ntop = (int)dist_left;
        if (ntop > C) { status = ORC_ERR_ARG; break; } /* IndexError in get_top_n_prob */
        ch->g = N[cur].cost - w4 * w->prob[(size_t)tb * C + key];
        ch->cost = ch->g;
        ch->h = -w2 * dist_left - w3 * H - w4 * topn[(size_t)tb * (C + 1) + ntop];
        ch->f = ch->g + ch->h;
      }
      int xi = np_index(ch->x + 500, VX), yi = np_index(ch->y + 200, VY);
      if (xi < 0 || yi < 0) { status = ORC_ERR_ARG; break; } /* IndexError */
      if (!visited[(size_t)xi * VY + yi]) {
        visited[(size_t)xi * VY + yi] = 1;
        visited_count++;
        ch->open = 1;
        n_open_nodes++;
      }
    }
    if (status != ORC_OK) break;
  }
  o->n_nodes = n_nodes; o->n_expansions = n_exp; o->n_children = n_children; o->found = found >= 0;
  o->visited_count = visited_count; o->path_len = 0; o->smooth_len = 0;
  o->n_hab_left = n_open;
  for (int i = 0; i < n_open && o->hab_left; i++) o->hab_left[i] = hopen[i];
  if (found >= 0 && status == ORC_OK) {
    int L = 0;
    for (int m = found; m >= 0; m = N[m].parent) L++;
    if (L > o->cap_path) status = ORC_ERR_CAPACITY;
    else {
      o->path_len = L;
      int k = 0;
      for (int m = found; m >= 0; m = N[m].parent, k++) {
        o->cost_list[k] = N[m].cost; /* leaf -> root, as appended during the backtrack */
        double* e = o->path + 3 * (size_t)(L - 1 - k);
        e[0] = N[m].x; e[1] = N[m].y; e[2] = round2(N[m].time_stamp);
        double* q = o->node_path + 8 * (size_t)(L - 1 - k);
        q[0] = N[m].x; q[1] = N[m].y; q[2] = N[m].g; q[3] = N[m].h; q[4] = N[m].f; q[5] = N[m].cost;
        q[6] = N[m].pathLen; q[7] = N[m].time_stamp;
      }
      if (V == 3) {
        /* smoothPath (:404-440) over the trajectory; `keep` marks what stays in smoothTraj */
        unsigned char* keep = (unsigned char*)malloc((size_t)L);
        memset(keep, 1, (size_t)L);
        if (L >= 2) {
          int index = 0, check = 0;
          index += 1;
          int curp = index;
          while (index < L - 1) {
            /* Walkable(checkPoint, currentPoint) (:218-242) */
            const double* a = o->path + 3 * (size_t)check;
            const double* b = o->path + 3 * (size_t)curp;
            double sx = a[0], sy = a[1];
            int step_x = (int)(fabs(b[0] - a[0]) / 5), step_y = (int)(fabs(b[1] - a[1]) / 5);
            int walk = 1, guard = 0;
            while (sx <= b[0] && sy <= b[1]) {
              double ix = sx, iy = sy;
              sx += step_x; sy += step_y;
              if (!collision_free(w, ix, iy)) { walk = 0; break; }
              if (++guard > 1000000) { status = ORC_ERR_ARG; break; } /* both steps 0: endless in the reference */
            }
            if (status != ORC_OK) break;
            if (walk) {
              int inside = 0;
              for (int h = 0; h < H; h++) {
                /* inside_habitats uses the module-level euclidean_dist (sqrt of squares of abs) */
                double d = sqrt(sqdist(w->habitats[3 * h], w->habitats[3 * h + 1], b[0], b[1]));
                if (d <= w->habitats[3 * h + 2]) { inside = 1; break; }
              }
              if (!inside) { keep[curp] = 0; index += 1; curp = index; }
              else { index += 1; curp = index; }
            } else {
              check = curp; index += 1; curp = index;
            }
          }
        }
        int s = 0;
        for (int i = 0; i < L; i++) {
          if (keep[i]) { memcpy(o->smooth_path + 3 * (size_t)s, o->path + 3 * (size_t)i, 3 * sizeof(double)); s++; }
        }
        o->smooth_len = s;
        free(keep);
      }
    }
  }
  o->status = status;
  free(N); free(hopen); free(hclosed); free(visited); free(rc); free(topn);
  return status;
}
