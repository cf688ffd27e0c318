// auvp_types.h -- device-side views of the world model and of the per-episode tree storage.
#ifndef AUVP_TYPES_H
#define AUVP_TYPES_H
#include <stdint.h>

namespace auvp {

// World model in HBM (read-only for the kernels; one copy shared by every episode).
// Obstacles are stored SoA with two derived columns that fold the reference's order-dependent
// collision test (rrt_dubins.py:535-541) into an elementwise one:
//   collision  <=>  exists i:  RN(sqrt(min_p d2(p, i))) <= R_i,   R_i = max_{k >= i} size_k
//              <=>  exists i, p:  d2(p, i) <= T_i,  T_i = largest double s with RN(sqrt(s)) <= R_i
// (sqrt is monotone, so the running min over "obstacles 0..k" of distances equals the sqrt of the
// running min of squared distances, and "some k >= i has size_k >= that" is the suffix max).
struct WorldDev {
  int32_t n_obstacles, n_habitats, n_poly, n_bins, n_cells, n_xbuckets;
  const double* ox;   // [O]
  const double* oy;   // [O]
  const double* ot;   // [O] T_i threshold on the squared distance
  const double* hab;  // [H,3]
  const double* hab_t;  // [H] T(size): `dist <= size` as a test on the squared distance
  const double* poly; // [V,2]
  const double* bins; // [T,2]
  const double* cells;  // [C,4]
  const double* prob;   // [T,C]
  // x-bucket index over the cells for the first-match scan of cost.py:181-184: bucket b lists, in
  // cell_list order, every cell whose [minx, min(maxx,maxy)] meets the bucket's x-range
  const int32_t* xb_off;    // [NB+1]
  const int32_t* xb_items;  // [xb_off[NB]] cell ids
  // per item, in bucket order: {minx, min(maxx,maxy), miny, min over this and all later items of
  // the bucket of miny}; the last column lets the scan stop as soon as no later cell can match
  const double* xb_data;    // [xb_off[NB]][4]
  double xb_x0, xb_inv_w;
  // region index over the same cells (exact, O(log) lookup).  The test `minx <= x <= min(maxx,maxy)` only
  // changes at the breakpoints {minx} U {min(maxx,maxy)}: for x on a breakpoint, or strictly between two
  // consecutive ones ("region"), the set of cells passing it is constant.  Region r lists that set in
  // cell_list order with the running minimum of miny, so the first cell with miny <= y is a binary search
  // on a non-increasing array.  Built when it fits the entry budget (always for grid-like cell lists);
  // otherwise rg_enabled = 0 and the bucket scan above is used.
  int32_t rg_enabled, n_rg_bp;
  const int32_t* rg_first;  // [NB+1] number of breakpoints that fall in earlier x-buckets
  const double* rg_bp;      // [m] breakpoints, ascending
  const int32_t* rg_off;    // [2m+2] candidate range of region r: 2i = (bp[i-1], bp[i]), 2i+1 = {bp[i]}
  const double* rg_pm;      // [total] running min of miny along the region's list
  const int32_t* rg_id;     // [total] cell id
  double bb[4];  // polygon bounds xmin,ymin,xmax,ymax (get_random_mps, :334)
  double safe_box[4];    // when the polygon is an axis-aligned rectangle: its corners (strict interior test)
  int32_t has_safe_box, _pad1;
};

struct RrtParamsDev {
  double dist_to_end, diff_max, freq, min_dist, bin_interval, v, max_traj_time, max_plan_time;
  double w[3];
  int32_t mode, max_iter, K, flags;
  double inv_bin_interval;  // RN(1 / bin_interval): first guess of t // bin_interval (corrected exactly by the remainder)
};

struct RrtSummary {  // must match auvp_rrt_summary in include/auvplan.h
  int32_t status, n_nodes, n_points, n_leaves, best_leaf, best_path_len, iters_run, n_candidates;
  double best_cost[4];
  double best_length;
  double rng_after;
  long long leaf_elems;
  unsigned long long n_draw32;
};

// Per-episode tree storage, episode-major.  Nodes are records (one 16-B + one 64-B access per
// node: they are read one at a time by the whole wave); path points are 48-B records, a node's run contiguous.
struct RrtBuffers {
  int32_t cap_nodes, cap_points, bin_cap, cap_leaves;
  double* node_f;    // [E][cap_nodes][8]  x, y, theta, traj_t, length, -, -, -   (64 B per node)
  int32_t* node_i;   // [E][cap_nodes][4]  plan_iter, parent, pt_off, pt_cnt       (16 B per node)
  double* points;    // [E][cap_points][6] x, y, theta, v, traj_t, length
  // cost-walk acceleration (derived data, never returned):
  //  * a path element's contribution to habitat_shark_cost_func -- w3*prob of its cell in its time bin, and
  //    the habitat it lies in -- does not depend on the leaf that walks over it (its bin is always part of
  //    the leaf's sub-dict), so it is evaluated the first time a walk meets it and kept:
  //    points in pt_term / pt_hab, nodes in node_f[6] / node_f[7]; hab = -2 marks "not evaluated yet"
  //  * anc[n] = the node_i records of n's parent, grandparent, ... (4 levels, one 64-B line), so the
  //    leaf -> root walk follows one dependent load per four ancestors
  double* pt_term;   // [E][cap_points]
  int8_t* pt_hab;    // [E][cap_points]
  int32_t* anc;      // [E][cap_nodes][16]
  int32_t* bin_items;                       // [E][K+1][bin_cap]
  int32_t* bin_count;                       // [E][K+1] (copied out of LDS at the end)
  uint32_t* mt;                             // [E][624] generator state in
  const int32_t* mt_index;                  // [E] position inside the state (624 = fresh seed)
  const double* init;                       // [E][6]
  RrtSummary* summary;                      // [E]
  int32_t* it_parent;                       // optional logs [E][max_iter]
  int8_t* it_accepted;
  int32_t* it_npath;
  double* leaf_cost;  // optional [E][cap_leaves][6]
  int32_t* leaf_iter;
  unsigned long long* phase_clocks;  // optional [E][5] shader clocks per phase
};

}  // namespace auvp
#endif
