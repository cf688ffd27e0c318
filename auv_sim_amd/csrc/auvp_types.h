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
  // separable-grid index: when cell_list is a row-major grid -- cell (r, c) = (X0[c], Y0[r], X1[c], Y1[r]) with all four
  // arrays non-decreasing -- the first-match scan of cost.py:181-184 factorises: the first row r with Y0[r] <= y and
  // x <= Y1[r] (sic: x against maxy) and, independently, the first column c with X0[c] <= x <= X1[c]; each is a
  // lower bound on a sorted array, compared against the table's own doubles (exact).  Takes precedence over the
  // region index (two dependent reads instead of five).
  int32_t sg_enabled, sg_ncol, sg_nrow, bins_sorted;  // bins_sorted: t0 and t1 of the time bins are non-decreasing
  const double* sg_x0;  // [ncol]
  const double* sg_x1;  // [ncol]
  const double* sg_y0;  // [nrow]
  const double* sg_y1;  // [nrow]
  // the same tables interleaved so that checking a guessed index costs one 32-byte read: entry i = {a1[i-1] (or -inf),
  // a1[i], a0[i], 0}: i is the lower bound of v in a1 iff entry.x < v <= entry.y
  const double* sg_col;  // [ncol][4]
  const double* sg_row;  // [nrow][4]
  double bins_inv_len;   // 1 / mean spacing of the bins' upper ends (guess only)
  double sg_x1_0, sg_y1_0, bins_t1_0;  // first entries of X1, Y1 and of the bins' upper ends (origins of the guesses)
  double sg_inv_dx, sg_inv_dy;  // 1 / mean spacing of X1 / Y1: first guess of the lower bound only
  double prob_absmax;  // max |prob|: bounds a path element's cost term (approximate-cost error bound of the leaf pass)
  // 1: no two entries of prob differ in sign, so the shark terms of a path all have one sign and sum|term| = |sum|
  int32_t prob_one_sign, _pad_prob;
  // habitat mask grid: hg_n x hg_n cells over the bounding box of the habitats' discs; cell -> bit set of the habitats
  // whose (slightly inflated) bounding square touches it.  A point outside the box, or in a cell with an empty set, lies
  // in no habitat; otherwise only the set's members are tested, in list order (first match, cost.py:187-191).
  int32_t hg_n, _pad_hg;
  double hg_x0, hg_y0, hg_inv_w, hg_inv_h;
  const unsigned long long* hg_mask;  // [hg_n * hg_n]
  // the obstacles once more, reordered along a space-filling curve and cut into 16 slots of 16 (rrt_rows_kernel; built
  // for <= 256 obstacles).  The collision decision is an OR over (point, obstacle) pairs of d2 <= T_i, so the order in
  // which a kernel looks at them is free once T_i (which encodes the reference's list order) is fixed.
  //   os_x, os_y, os_t [256]  centre and threshold, padded with entries that never collide
  //   os_r [256]              cull radius >= sqrt(T) as float, -inf for padding
  //   os_box [16][4]          per slot: x0, y0, x1, y1 of the union of its obstacles' cull squares (empty slot: inverted)
  const double* os_x;
  const double* os_y;
  const double* os_t;
  const float* os_r;
  const double* os_box;
  double bb[4];  // polygon bounds xmin,ymin,xmax,ymax (get_random_mps, :334)
  double safe_box[4];    // when the polygon is an axis-aligned rectangle: its corners (strict interior test)
  int32_t has_safe_box, _pad1;
};

// kernel-internal bits of RrtParamsDev::flags (above the public AUVP_FLAG_* bits)
#define AUVP_KFLAG_TIGHT_CULL 1024
#define AUVP_KFLAG_NN_EXACT 2048  // nearest-neighbour scan: always rank the reference's way (sqrt per node); tests
#define AUVP_KFLAG_ASTAR_NO_LIST 4096  // A* fixLen variants: pop by the scan of every node's f in memory from the start (what an open set larger than
                                      // the LDS list falls back to); tests (AUVP_ASTAR_NO_LIST=1)

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
  unsigned long long nn_scanned;  // nearest-neighbour mode: sum over the iterations of len(mps_list) (16 B of x,y each)
};

// Per-episode tree storage, episode-major.  Nodes are records (one 16-B + one 64-B access per
// node: they are read one at a time by the whole wave); path points are 48-B records, a node's run contiguous.
#define AUVP_BIN_HEAD 128

struct RrtBuffers {
  int32_t cap_nodes, cap_points, bin_cap, cap_leaves;
  // [E][cap_nodes][8]  x, y, theta, traj_t, length, S, tv, hab   (64 B per node); S/tv/hab: see below
  double* node_f;
  int32_t* node_i;   // [E][cap_nodes][4]  plan_iter, parent, pt_off, pt_cnt       (16 B per node)
  // nearest-neighbour mode only: contiguous mirror of the nodes' x, y (16 B per node; what get_closest_mps :505-513 reads
  // of every node, every iteration), [E][xy_stride] with xy_stride = cap_nodes rounded up to a whole scan block; entries
  // past the tree's end hold +inf (never the nearest)
  double* node_xy;
  long long xy_stride;
  // [E] x { [cap_points][3] x, y, traj_t ; [cap_points][3] theta, v, length }: two 24-byte records per path point -- the
  // leaf pass reads only the first (half the bytes); a node's points are one contiguous run in both
  double* points;
  // habitat_shark_cost_func bookkeeping (derived data, never returned), filled by rrt_leaf_kernel.  A path element's
  // contribution to the cost of a leaf -- w3*prob of its cell in its time bin, and the habitat it lies in -- does not
  // depend on the leaf (its bin is always part of the leaf's sub-dict), so it is evaluated once and summed down the
  // tree:  node_f[6] / node_f[7] = term and habitat of the node's own state;  node_f[5] = S = sum of the terms of every
  // element on the root..node path (any order: an approximation of the reference's ordered sum with a rigorous bound);
  // node_c = {number of elements inside some habitat, number of elements, visited-habitat bit set} of that path
  // (exact).  The leaf pass ranks the qualifying leaves with these and re-sums in the reference's order only where the
  // bound cannot decide.
  int32_t* node_c;   // [E][cap_nodes] 32-byte records {S f64, hits i32 | elements i32, visited mask u64, -}
  // [E][cap_nodes] 1 = the node is a qualifying leaf (traj_t >= max_traj_time - 30, :158), written by the expansion
  // kernels: the leaf pass only visits those and their ancestors
  uint8_t* node_q;
  // time-bin member lists, per episode (bin_stride int32 words): the first AUVP_BIN_HEAD members of every bin
  // direct-mapped [K+1][AUVP_BIN_HEAD]; members beyond that in 64-entry chunks handed out on demand from [bin_over][64],
  // found through the chunk directory [bin_slots][K+1] (slot s of bin b = members AUVP_BIN_HEAD + 64 s .. of b).  Every
  // node is in exactly one bin, so bin_over = ceil(cap_nodes / 64) + K + 1 chunks always suffice.
  int32_t* bin_items;
  int32_t bin_over, bin_slots;
  long long bin_stride;
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
  // optional [4], summed over the episodes of a launch by rrt_leaf_kernel: nodes visited by its sweep (qualifying leaves and
  // their ancestors), path points of those nodes, path elements re-summed in the reference's order, leaves re-summed --
  // the units of the leaf pass's compulsory-traffic figure (bench.py)
  unsigned long long* leaf_stats;
  int32_t* pipe_fail;  // host-mapped word, set to 1 by an episode that ends with AUVP_ST_PIPELINE (null: not reported)
  // round 6: [E][stream_cap] the episodes' random() numbers, generated ahead by rrt_stream_kernel for rrt_rows_stream_kernel
  // (null otherwise)
  double* stream;
  long long stream_cap;
};

}  // namespace auvp
#endif
