/* orc_rrt.c -- CPU restatement of the goal-less RRT ("exploring") and its cost/collision helpers.
 *
 * TEST INFRASTRUCTURE (oracle/): checker + reported CPU baseline only (see orc_api.h).
 *
 * Follows, statement by statement (all paths relative to /root/reference):
 *   RRT.exploring            path_planning/rrt_dubins.py:92-176
 *   RRT.steer                path_planning/rrt_dubins.py:237-295
 *   RRT.check_collision      path_planning/rrt_dubins.py:530-549  (+ get_distance_angle :558-564)
 *   RRT.get_random_mps       path_planning/rrt_dubins.py:333-343
 *   RRT.get_closest_mps      path_planning/rrt_dubins.py:505-513
 *   RRT.get_closest_mps_time path_planning/rrt_dubins.py:515-528
 *   RRT.generate_final_course path_planning/rrt_dubins.py:321-331
 *   habitat_shark_cost_func  path_planning/cost.py:145-207
 * Point.within(polygon) (shapely, absent from the reference tree and from this image) is the
 * even-odd crossing test defined in tests/golden/_refstubs/install.py -- parity at that boundary
 * is pinned by the build itself, not by the reference.
 *
 * The wall-clock loop bound (`while time.time() < t_end`) is replaced by the virtual clock of
 * exactly max_iter loop iterations, plan_time_stamp == 0-based iteration index.
 */
#include <stdlib.h>
#include <string.h>
#include "cpyrandom.h"
#include "orc_api.h"
#include "orc_math.h"

const char* orc_math_name(void) { return ORC_MATH_NAME; }
double orc_sin(double x) { return ORC_SIN(x); }
double orc_cos(double x) { return ORC_COS(x); }
double orc_atan2(double y, double x) { return ORC_ATAN2(y, x); }
double orc_hypot(double x, double y) { return ORC_HYPOT(x, y); }

void orc_rng_kat(uint64_t seed, int n, double* out_random, uint32_t* out_bits32, int nchoice,
                 uint32_t choice_n, uint32_t* out_choice) {
  cpy_rng r;
  cpy_seed_u64(&r, seed);
  for (int i = 0; i < n; i++) out_random[i] = cpy_random(&r);
  for (int i = 0; i < n; i++) out_bits32[i] = cpy_genrand32(&r);
  for (int i = 0; i < nchoice; i++) out_choice[i] = cpy_randbelow(&r, choice_n);
}

/* get_distance_angle(): only the distance is consumed on this path (rrt_dubins.py:538,562) */
static inline double dist2d(double ax, double ay, double bx, double by) {
  double dx = bx - ax, dy = by - ay;
  return ORC_SQRT(ORC_POW2(dx) + ORC_POW2(dy));
}

/* Point(x,y).within(polygon): even-odd crossing number, strict comparisons (stub definition) */
static int point_within(const double* poly, int nv, double x, double y) {
  int inside = 0;
  int j = nv - 1;
  for (int i = 0; i < nv; i++) {
    double xi = poly[2 * i], yi = poly[2 * i + 1];
    double xj = poly[2 * j], yj = poly[2 * j + 1];
    if ((yi > y) != (yj > y)) {
      if (x < (xj - xi) * (y - yi) / (yj - yi) + xi) inside = !inside;
    }
    j = i;
  }
  return inside;
}

/* check_collision (rrt_dubins.py:530-549).  dList is shared by all obstacles, so obstacle k is
 * compared with the minimum distance from any path point to any of obstacles 0..k.
 * Returns 1 when the path is free ("safe"). */
int orc_check_collision(const orc_world* w, int npts, const double* pts_xy) {
  double run_min = INFINITY;
  for (int k = 0; k < w->n_obstacles; k++) {
    double ox = w->obstacles[3 * k], oy = w->obstacles[3 * k + 1], r = w->obstacles[3 * k + 2];
    for (int i = 0; i < npts; i++) {
      double d = dist2d(ox, oy, pts_xy[2 * i], pts_xy[2 * i + 1]);
      if (d < run_min) run_min = d;
    }
    if (run_min <= r) return 0;
  }
  for (int i = 0; i < npts; i++)
    if (!point_within(w->polygon, w->n_poly, pts_xy[2 * i], pts_xy[2 * i + 1])) return 0;
  return 1;
}

/* habitat_shark_cost_func (path_planning/cost.py:145-207) over the sub-dict of bins [bin_lo,bin_hi).
 * out4 = {sum(cost), cost[0], cost[1], cost[2]} */
static void cost_selected(const orc_world* w, const unsigned char* sel, int npts, const double* pts_xyt,
                          double total_traj_time, const double* weights, double* out4);

void orc_cost(const orc_world* w, int bin_lo, int bin_hi, int npts, const double* pts_xyt,
              double total_traj_time, const double* weights, double* out4) {
  unsigned char* sel = (unsigned char*)calloc(w->n_bins > 0 ? w->n_bins : 1, 1);
  for (int b = bin_lo; b < bin_hi && b < w->n_bins; b++) sel[b] = 1;
  cost_selected(w, sel, npts, pts_xyt, total_traj_time, weights, out4);
  free(sel);
}

/* the same over an arbitrary sub-dict: sel[b] != 0 <=> bin b is a key of it (dict order = bin order) */
static void cost_selected(const orc_world* w, const unsigned char* sel, int npts, const double* pts_xyt,
                          double total_traj_time, const double* weights, double* out4) {
  double w1 = weights[0], w2 = weights[1], w3 = weights[2];
  double c0 = 0.0, c1 = 0.0, c2 = 0.0;
  int H = w->n_habitats, C = w->n_cells;
  unsigned char* visited = (unsigned char*)calloc(H > 0 ? H : 1, 1);
  for (int i = 0; i < npts; i++) {
    double x = pts_xyt[3 * i], y = pts_xyt[3 * i + 1], t = pts_xyt[3 * i + 2];
    int tb = -1;
    for (int b = 0; b < w->n_bins; b++) {
      if (sel[b] && t >= w->bins[2 * b] && t <= w->bins[2 * b + 1]) { tb = b; break; }
    }
    if (tb < 0) continue; /* no bin: neither the shark nor the habitat term (cost.py:178-179) */
    const double* pr = w->prob + (size_t)tb * C;
    for (int c = 0; c < C; c++) {
      const double* cb = w->cells + 4 * c;
      /* sic: x is compared with maxy (cost.py:182) */
      if (x >= cb[0] && x <= cb[2] && y >= cb[1] && x <= cb[3]) { c2 += w3 * pr[c]; break; }
    }
    for (int h = 0; h < H; h++) {
      double hx = w->habitats[3 * h], hy = w->habitats[3 * h + 1], hr = w->habitats[3 * h + 2];
      double d = ORC_SQRT(ORC_POW2(hx - x) + ORC_POW2(hy - y));
      if (d <= hr) { visited[h] = 1; c1 += w2; break; }
    }
  }
  if (total_traj_time > 0) {
    c1 = c1 / total_traj_time;
    c2 = c2 / total_traj_time;
  }
  int count = 0;
  for (int h = 0; h < H; h++) count += visited[h];
  if (H != 0) c0 = w1 * count / H;
  free(visited);
  out4[0] = ((0.0 + c0) + c1) + c2; /* sum([c0,c1,c2]) starts from int 0 */
  out4[1] = c0;
  out4[2] = c1;
  out4[3] = c2;
}

typedef struct { int32_t* v; int n, cap; } ivec;
static void ivec_push(ivec* a, int32_t x) {
  if (a->n == a->cap) { a->cap = a->cap ? 2 * a->cap : 8; a->v = (int32_t*)realloc(a->v, sizeof(int32_t) * a->cap); }
  a->v[a->n++] = x;
}

/* leaf -> root element count (generate_final_course) */
static int course_len(const orc_rrt_out* t, int leaf) {
  int n = 1;
  for (int m = leaf; t->parent[m] >= 0; m = t->parent[m]) n += t->pt_cnt[m] + 1;
  return n;
}

static void node_elem(const orc_rrt_out* t, const double* init7, int m, double* e) {
  if (t->parent[m] < 0 && init7) { memcpy(e, init7, 7 * sizeof(double)); return; }
  const double* nd = t->nodes + 6 * (size_t)m;
  e[0] = nd[0]; e[1] = nd[1]; e[2] = nd[2]; e[3] = 0.0; e[4] = nd[3]; e[5] = nd[4]; e[6] = nd[5];
}

int orc_rrt_final_course(const orc_rrt_out* t, const double* init7, int leaf, double* path7, int cap) {
  int n = course_len(t, leaf);
  if (n > cap) return -n;
  /* fill from the back: element n-1 is the leaf, element 0 the root */
  int pos = n - 1;
  node_elem(t, init7, leaf, path7 + 7 * (size_t)pos--);
  for (int m = leaf; t->parent[m] >= 0; m = t->parent[m]) {
    for (int k = t->pt_cnt[m] - 1; k >= 0; k--)
      memcpy(path7 + 7 * (size_t)pos--, t->points + 7 * (size_t)(t->pt_off[m] + k), 7 * sizeof(double));
    node_elem(t, init7, t->parent[m], path7 + 7 * (size_t)pos--);
  }
  return n;
}

int orc_rrt_explore(const orc_world* w, const orc_rrt_params* p, uint64_t seed, orc_rrt_out* o) {
  cpy_rng rng;
  cpy_seed_u64(&rng, seed);
  const int max_iter = p->max_iter;
  if (o->cap_nodes < 1) return ORC_ERR_CAPACITY;
  int status = ORC_OK;

  /* mps_list = [initial] */
  memcpy(o->nodes, p->init, 6 * sizeof(double));
  o->parent[0] = -1; o->pt_off[0] = 0; o->pt_cnt[0] = 0;
  int n_nodes = 1, n_points = 0, n_leaves = 0;
  double opt_cost[4] = {INFINITY, 0, 0, 0};
  int opt_leaf = -1;
  double opt_len = 0.0;

  /* time bins (rrt_dubins.py:110-114): keys bin_interval*i, i = 1..K; start goes to bin 1 */
  int K = 0;
  ivec* bins = NULL;
  if (p->mode == ORC_MODE_TIMEBIN) {
    K = (int)ceil(p->max_traj_time / p->bin_interval);
    bins = (ivec*)calloc((size_t)(K > 0 ? K : 1) + 1, sizeof(ivec));
    ivec_push(&bins[1 > K ? 0 : 1], 0);
  }
  double xmin = INFINITY, ymin = INFINITY, xmax = -INFINITY, ymax = -INFINITY;
  for (int i = 0; i < w->n_poly; i++) {
    double x = w->polygon[2 * i], y = w->polygon[2 * i + 1];
    if (x < xmin) xmin = x;
    if (x > xmax) xmax = x;
    if (y < ymin) ymin = y;
    if (y > ymax) ymax = y;
  }
  int max_pts = (int)p->freq + 2;
  double* path_xy = (double*)malloc(sizeof(double) * 2 * (size_t)(max_pts + 1));
  double* tmp_pts = (double*)malloc(sizeof(double) * 7 * (size_t)(max_pts + 1));
  double* course = NULL;
  unsigned char* bin_sel = (unsigned char*)calloc(w->n_bins > 0 ? w->n_bins : 1, 1);
  int course_cap = 0;
  const double init_t = p->init[3];
  int it;
  for (it = 0; it < max_iter; it++) {
    o->it_parent[it] = -1; o->it_accepted[it] = 0; o->it_npath[it] = 0;
    int par;
    if (p->mode == ORC_MODE_TIMEBIN) {
      int rb = (int)cpy_uniform(&rng, 1.0, (double)(K + 1));
      if (rb > K) { status = ORC_ERR_ARG; break; } /* KeyError in the reference */
      while (bins[rb].n == 0) {
        rb = (int)cpy_uniform(&rng, 1.0, (double)(K + 1));
        if (rb > K) { status = ORC_ERR_ARG; break; }
      }
      if (status < 0) break;
      int ri = (int)cpy_uniform(&rng, 0.0, (double)bins[rb].n);
      par = bins[rb].v[ri];
    } else if (p->mode == ORC_MODE_PLANTIME) {
      double ran_time = cpy_uniform(&rng, 0.0, p->max_plan_time * p->freq);
      int lo = 0, hi = n_nodes; /* list slicing of get_closest_mps_time */
      while (hi - lo > 3) {
        int n = hi - lo;
        double ld = fabs(o->nodes[6 * (size_t)(lo + n / 2 - 1) + 4] - ran_time);
        double rd = fabs(o->nodes[6 * (size_t)(lo + n / 2 + 1) + 4] - ran_time);
        if (ld >= rd) lo += n / 2; else hi = lo + n / 2;
      }
      par = lo;
      if (o->nodes[6 * (size_t)par + 3] > p->max_traj_time) continue;
    } else {
      double rx = cpy_uniform(&rng, xmin, xmax);
      double ry = cpy_uniform(&rng, ymin, ymax);
      (void)cpy_uniform(&rng, -M_PI, M_PI);
      (void)cpy_uniform(&rng, 0.0, 15.0);
      double md = dist2d(o->nodes[0], o->nodes[1], rx, ry);
      par = 0;
      for (int m = 0; m < n_nodes; m++) {
        double d = dist2d(o->nodes[6 * (size_t)m], o->nodes[6 * (size_t)m + 1], rx, ry);
        if (d < md) { md = d; par = m; }
      }
      if (o->nodes[6 * (size_t)par + 3] > p->max_traj_time) continue;
    }

    /* steer (rrt_dubins.py:252-295), min_dist=0.5 velocity=v (call site :141) */
    const double* pn = o->nodes + 6 * (size_t)par;
    double x = pn[0], y = pn[1], th = pn[2], tt = pn[3], len = pn[5];
    const double plan_t = (double)it;
    int n_expand = (int)floor(cpy_uniform(&rng, 0.0, p->freq) / 1);
    int cnt = 0;
    path_xy[0] = pn[0]; path_xy[1] = pn[1];
    for (int s = 0; s < n_expand; s++) {
      double dist = cpy_uniform(&rng, 0.0, p->dist_to_end);
      double diff = cpy_uniform(&rng, -p->diff_max, p->diff_max);
      if (fabs(dist) > fabs(diff)) {
        double s1 = dist + diff, s2 = dist - diff;
        double radius = (s1 + s2) / (-s1 + s2);
        double phi = (s1 + s2) / (2 * radius);
        double ori = th;
        th += phi;
        double dx = radius * (ORC_SIN(th) - ORC_SIN(ori));
        double dy = radius * (-ORC_COS(th) + ORC_COS(ori));
        x += dx;
        y += dy;
        double vt = cpy_uniform(&rng, 0.0, 2 * p->v);
        double movement = ORC_SQRT(ORC_POW2(dx) + ORC_POW2(dy));
        tt += movement / vt;
        len += movement;
        if (movement >= p->min_dist) {
          double* q = tmp_pts + 7 * (size_t)cnt;
          q[0] = x; q[1] = y; q[2] = th; q[3] = vt; q[4] = tt; q[5] = plan_t; q[6] = len;
          cnt++;
          path_xy[2 * cnt] = x; path_xy[2 * cnt + 1] = y;
        }
      }
    }
    o->it_parent[it] = par;
    o->it_npath[it] = cnt + 1;
    int ok = orc_check_collision(w, cnt + 1, path_xy);
    o->it_accepted[it] = (int8_t)ok;
    if (!ok) continue;
    if (n_nodes >= o->cap_nodes || n_points + cnt > o->cap_points) { status = ORC_ERR_CAPACITY; break; }
    int me = n_nodes++;
    double* nn = o->nodes + 6 * (size_t)me;
    nn[0] = x; nn[1] = y; nn[2] = th; nn[3] = tt; nn[4] = plan_t; nn[5] = len;
    o->parent[me] = par; o->pt_off[me] = n_points; o->pt_cnt[me] = cnt;
    memcpy(o->points + 7 * (size_t)n_points, tmp_pts, sizeof(double) * 7 * (size_t)cnt);
    n_points += cnt;
    if (p->mode == ORC_MODE_TIMEBIN) {
      double fi = orc_floordiv(tt, p->bin_interval) + 1.0;
      double curr_bin = fi * p->bin_interval;
      if (curr_bin > p->max_traj_time) {
        /* the key is replaced by an empty list before the append (:149-151).  Keys beyond K are
         * never sampled; a regular key (only when max_traj_time is not a multiple of
         * bin_interval) loses its earlier members. */
        if (fi <= (double)K) { bins[(int)fi].n = 0; ivec_push(&bins[(int)fi], me); }
      } else {
        ivec_push(&bins[(int)fi], me);
      }
    }
    if (tt >= p->max_traj_time - 30) {
      int L = course_len(o, me);
      if (L > course_cap) { course_cap = 2 * L; course = (double*)realloc(course, sizeof(double) * 3 * (size_t)course_cap); }
      /* leaf -> root order, as generate_final_course builds it (order matters for nothing but
       * the summation order of c2) */
      int pos = 0;
      course[0] = x; course[1] = y; course[2] = tt; pos = 1;
      for (int m = me; o->parent[m] >= 0; m = o->parent[m]) {
        for (int k = o->pt_cnt[m] - 1; k >= 0; k--) {
          const double* q = o->points + 7 * (size_t)(o->pt_off[m] + k);
          course[3 * pos] = q[0]; course[3 * pos + 1] = q[1]; course[3 * pos + 2] = q[4]; pos++;
        }
        const double* q = o->nodes + 6 * (size_t)o->parent[m];
        course[3 * pos] = q[0]; course[3 * pos + 1] = q[1]; course[3 * pos + 2] = q[3]; pos++;
      }
      /* shark bins overlapping [initial.t, leaf.t] (:161-166); dict order preserved */
      int nsel = 0;
      for (int b = 0; b < w->n_bins; b++) {
        double b0 = w->bins[2 * b], b1 = w->bins[2 * b + 1];
        bin_sel[b] = (init_t >= b0 && init_t <= b1) || (b0 >= init_t && b1 <= tt) || (tt >= b0 && tt <= b1);
        nsel += bin_sel[b];
      }
      double c4[4];
      cost_selected(w, bin_sel, L, course, tt, p->w, c4);
      if (n_leaves < o->cap_leaves) {
        double* lc = o->leaf_cost + 6 * (size_t)n_leaves;
        lc[0] = c4[0]; lc[1] = c4[1]; lc[2] = c4[2]; lc[3] = c4[3]; lc[4] = (double)L; lc[5] = (double)nsel;
        o->leaf_iter[n_leaves] = it;
      }
      n_leaves++;
      if (c4[0] < opt_cost[0]) { memcpy(opt_cost, c4, sizeof c4); opt_leaf = me; opt_len = len; }
    }
  }
  o->iters_run = it;
  o->n_nodes = n_nodes; o->n_points = n_points; o->n_leaves = n_leaves;
  o->best_leaf = opt_leaf; o->best_length = opt_len;
  memcpy(o->best_cost, opt_cost, sizeof opt_cost);
  o->n_bins = K;
  for (int i = 0; i < K && i < o->cap_bins; i++) o->bin_sizes[i] = bins[i + 1].n;
  o->n_draw32 = rng.n_draw32; /* what exploring consumed; the peek below is a test probe, not part of it */
  o->rng_after = cpy_random(&rng);
  if (status == ORC_OK && opt_leaf < 0) status = ORC_NO_QUALIFYING_LEAF;
  o->status = status;
  if (bins) { for (int i = 0; i <= K; i++) free(bins[i].v); free(bins); }
  free(path_xy); free(tmp_pts); free(course); free(bin_sel);
  return status;
}
