/* auvplan.h -- C-ABI of libauvplan.so: the MI355X (gfx950) path-planning hot path.
 *
 * The reference (hmc-lair-shark-tracking/auv-sim) is pure Python and has no FFI; its boundary for
 * this path is the Python planner API.  The Python drop-in classes in auv_sim_amd/
 * keep those signatures and call the entry points below through ctypes.  Each entry point cites the
 * reference interface it replaces (paths relative to the reference tree).
 *
 * Conventions: plain C types; host pointers unless a name ends in _dev; caller-allocated outputs;
 * sizes queried first (two-phase) where variable; integer status (0 ok, >0 reference-defined
 * outcome, <0 error + auvp_last_error()); nothing throws across the ABI; one HIP stream per handle;
 * a handle is thread-compatible (no internal global state), not thread-safe.
 */
#ifndef AUVPLAN_H
#define AUVPLAN_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct auvp_handle auvp_handle;

enum {
  AUVP_OK = 0,
  AUVP_NO_QUALIFYING_LEAF = 1, /* exploring(): opt_path stayed None -> TypeError at rrt_dubins.py:174 */
  AUVP_ERR_ARG = -1,
  AUVP_ERR_CAPACITY = -2, /* a device buffer sized from the budget overflowed; nothing is truncated silently */
  AUVP_ERR_HIP = -3,
  AUVP_ERR_STATE = -4,
  AUVP_ERR_KEY = -5, /* time-bin key outside 1..K: KeyError at rrt_dubins.py:124 */
  AUVP_ERR_COMM = -6, /* RCCL missing or a collective failed (auvp_comm_*, auvp_gather*) */
  /* per-episode statuses a kernel can leave in a summary record (never returned by an entry point): */
  AUVP_ERR_GENERATOR = -7, /* an episode's stored MT19937 block phase does not fit the kernel that picked it up (a batch was
                            * moved between the one-episode and the four-episodes-per-wavefront planner kernels) */
  AUVP_ERR_PIPELINE = -9   /* a bounded wait of a multi-wavefront (speculative) latency kernel ran out.  The reference cannot
                            * fail there (rrt_dubins.py:116-151, gym_rrt/envs/rrt_dubins.py:205-248): the host repeats the
                            * affected work on the one-wavefront kernel inside the same call and counts it
                            * (auvp_pipeline_fallbacks); the status stays visible only with option PIPE_FALLBACK = 0 */
  ,
  AUVP_ERR_STREAM = -10    /* RRT.exploring with the random numbers generated ahead (large time-bin batches from the second
                            * batch with a parameter block on; option ROWS_STREAM): an episode drew more numbers than
                            * the stream holds (the busiest episode of the previous batch + 3 % + 1 024).  The host repeats
                            * the batch with the generator inside the kernel in the same call and counts it
                            * (auvp_pipeline_fallbacks); visible only with option PIPE_FALLBACK = 0 */
};

enum { AUVP_MODE_TIMEBIN = 0, AUVP_MODE_PLANTIME = 1, AUVP_MODE_NN = 2 };
enum { AUVP_FLAG_ITER_LOG = 1, AUVP_FLAG_LEAF_LOG = 2, AUVP_FLAG_PHASE_CLOCKS = 4, AUVP_FLAG_KEEP_VISITED = 8 };

const char* auvp_version(void);
/* create a planner context on HIP device `device`; fails (AUVP_ERR_HIP) when no gfx950 GPU is
 * usable -- there is no CPU fallback */
int auvp_create(int device, auvp_handle** out);
void auvp_destroy(auvp_handle* h);
/* Tuning / diagnostic options of a handle.  Every kernel choice the host makes (how many wavefronts per episode, LDS tiles,
 * culls) has a measured default; an option forces it: `name` without the AUVP_ prefix, e.g. "ROWS", "DUO", "TRIO", "QUAD",
 * "PRRT_ROWS", "PRRT_PIPE", "PRRT_LAT", "ASTAR_PAIR", "SOG_TILE", "TIGHT_CULL", "NN_EXACT", "PIPE_FALLBACK" (the full list:
 * INTEGRATION.md).  The environment variable AUVP_<NAME> gives an option its initial value when the handle is created; no
 * other call reads the environment.  auvp_unset_option returns the choice to the default.  Unknown name: AUVP_ERR_ARG. */
int auvp_set_option(auvp_handle* h, const char* name, int64_t value);
int auvp_unset_option(auvp_handle* h, const char* name);
int auvp_get_option(auvp_handle* h, const char* name, int32_t* is_set, int64_t* value);
/* episodes / instances the last planning call (*last) and all calls so far (*total) repeated on the one-wavefront kernel
 * because a speculative latency kernel ended them with AUVP_ERR_PIPELINE (0 in normal operation) */
int auvp_pipeline_fallbacks(auvp_handle* h, int32_t* last, int64_t* total);
const char* auvp_last_error(auvp_handle* h);

/* World model shared by every episode of a batch.
 * Replaces the Python-side arguments of RRT.__init__ (path_planning/rrt_dubins.py:26: boundary
 * polygon, obstacle list, sharkGrid dict, cell_list) and the `habitats` list of exploring() (:92).
 *   obstacles [O,3] x,y,size in LIST ORDER (check_collision is order dependent, :535-541)
 *   habitats  [H,3]
 *   polygon   [V,2] boundary vertices (Point.within, :545-546)
 *   bins      [T,2] shark-grid time bins in dict order; cells [C,4] bounds in cell_list order;
 *   prob      [T,C] (createSharkGrid, :612-630) */
int auvp_world_set(auvp_handle* h, const double* obstacles, int32_t n_obstacles, const double* habitats,
                   int32_t n_habitats, const double* polygon, int32_t n_poly, const double* bins,
                   int32_t n_bins, const double* cells, int32_t n_cells, const double* prob);

/* replace only the habitat list (exploring() takes it per call, rrt_dubins.py:92) */
int auvp_world_set_habitats(auvp_handle* h, const double* habitats, int32_t n_habitats);

typedef struct {
  double dist_to_end, diff_max, freq; /* RRT.__init__ kwargs, rrt_dubins.py:26 */
  double min_dist;                     /* steer(min_dist=0.5), call site :141 */
  double bin_interval, v;              /* exploring() args, :92 */
  double max_traj_time;
  double max_plan_time;                /* only feeds ran_time in plan-time mode (:129) */
  double w[3];                         /* weights */
  int32_t mode;                        /* AUVP_MODE_*: (plan_time, traj_time_stamp) flags of :121-139 */
  int32_t max_iter;                    /* iteration budget = virtual clock of SURVEY 8(c) */
  double points_per_iter;              /* device path-point capacity per iteration; 0 -> freq/2 + 8 sigma of the sum */
} auvp_rrt_params;

typedef struct {
  int32_t status, n_nodes, n_points, n_leaves, best_leaf, best_path_len, iters_run,
          n_candidates; /* obstacles that survived the bounding-box cull, whole episode (diagnostic) */
  double best_cost[4]; /* sum, c0, c1, c2 (habitat_shark_cost_func, path_planning/cost.py:145) */
  double best_length;  /* "path length" of the returned dict, rrt_dubins.py:171,176 */
  double rng_after;    /* next random() of the episode's stream (parity probe: same draw count) */
  int64_t leaf_elems;  /* sum over qualifying leaves of len(path) (cost-walk reads; bench byte count) */
  uint64_t n_draw32;   /* 32-bit MT19937 outputs consumed: lets the host advance Python's global `random` */
  uint64_t nn_scanned; /* nearest-neighbour sampling: sum over iterations of len(mps_list) scanned by get_closest_mps */
} auvp_rrt_summary;

/* RRT.exploring (path_planning/rrt_dubins.py:92-176) for E independent episodes, one wavefront
 * each.  init [E,6] = x,y,theta,traj_time_stamp,plan_time_stamp,length of `initial`;
 * seeds [E]: the episode draws the stream `random.seed(seeds[e])` would give. */
int auvp_rrt_explore_batch(auvp_handle* h, int32_t n_episodes, const double* init, const uint64_t* seeds,
                           const auvp_rrt_params* params, int32_t flags);
/* the same in two steps: prepare() sizes the HBM buffers and uploads init/seeds (MT states are
 * seeded on the host like random.seed), run() launches the kernel on the prepared batch and can be
 * repeated (every run restarts from the prepared inputs) */
int auvp_rrt_prepare(auvp_handle* h, int32_t n_episodes, const double* init, const uint64_t* seeds,
                     const auvp_rrt_params* params, int32_t flags);
/* as prepare(), but every episode continues an existing generator: mt [E,624] words + mt_index [E]
 * exactly as random.getstate() reports them (drop-in use: the global `random` state) */
int auvp_rrt_prepare_states(auvp_handle* h, int32_t n_episodes, const double* init, const uint32_t* mt,
                            const int32_t* mt_index, const auvp_rrt_params* params, int32_t flags);
int auvp_rrt_run(auvp_handle* h);
int auvp_rrt_summaries(auvp_handle* h, auvp_rrt_summary* out /* [E] */);
/* generate_final_course (:321-331) of every episode's best leaf, reversed to root->leaf (:174).
 * offsets [E+1] = exclusive prefix sum of best_path_len; out [offsets[E],7] =
 * x,y,theta,v,traj_time_stamp,plan_time_stamp,length per element */
int auvp_rrt_paths(auvp_handle* h, const int64_t* offsets, double* out);
/* the same, left in HBM: out_dev is a device pointer to [offsets[E],7] doubles (multi-GPU gather) */
int auvp_rrt_paths_dev(auvp_handle* h, const int64_t* offsets, void* out_dev);
/* whole tree of one episode (tests / RRT.mps_list): nodes [n_nodes,6] x,y,theta,traj_t,plan_t,length */
int auvp_rrt_tree(auvp_handle* h, int32_t episode, double* nodes6, int32_t* parent, int32_t* pt_off,
                  int32_t* pt_cnt, double* points7);
int auvp_rrt_iter_log(auvp_handle* h, int32_t episode, int32_t* it_parent, int8_t* it_accepted,
                      int32_t* it_npath);
int auvp_rrt_leaf_log(auvp_handle* h, int32_t episode, double* leaf_cost6, int32_t* leaf_iter);
int auvp_rrt_bin_sizes(auvp_handle* h, int32_t episode, int32_t* sizes /* [K] */, int32_t* n_bins);

/* device-side result records for the multi-GPU gather: pointer to [E] auvp_rrt_summary in HBM */
void* auvp_rrt_summaries_dev(auvp_handle* h);

/* ---------------------------------------------------------------------------------------------
 * Planner_RRT (gym_rrt/envs/rrt_dubins.py:34): goal-directed RRT driven by the RL environment.
 * Obstacles come from auvp_world_set (list order matters, :445-451); everything else is here.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  double rect[4];          /* boundary corners x0,y0,x1,y1 (boundary[0], boundary[1], :45) */
  double exp_rate, dist_to_end, diff_max, freq, cell_side_length; /* constructor kwargs, :34 */
  int32_t subsections;     /* subsections_in_cell */
  int32_t max_step;        /* node / step capacity of the batch (planning(max_step), :162) */
} auvp_prrt_params;

typedef struct {
  int32_t status, n_nodes, n_points, n_occ, steps, done, path_len, last_node;
  int32_t last_accepted, last_new_node, n_arc, _pad; /* outcome of the most recent step */
  double arc[6];           /* successful goal arc: x_C, y_C, radius, ang_vel, theta_0, length */
  double rng_after;        /* next random() of the episode's stream (not consumed) */
  uint64_t n_draw32;
} auvp_prrt_summary;

/* Planner_RRT.__init__ for E episodes: starts [E,4] x,y,theta,traj_time_stamp; goals [E,2]; the start
 * node is put into the (row, col, theta-subsection) bucket grid (:53,:108-159).  RNG: seeds [E]
 * (random.seed(seed) streams) or, if seeds is NULL, mt [E,624] + mt_index [E] (random.getstate()). */
int auvp_prrt_create_batch(auvp_handle* h, int32_t n_episodes, const double* starts, const double* goals,
                           const auvp_prrt_params* params, const uint64_t* seeds, const uint32_t* mt,
                           const int32_t* mt_index, int32_t flags);
/* Planner_RRT.planning(max_step) (:162-202) for every episode: device-side loop until done / budget */
int auvp_prrt_plan(auvp_handle* h);
/* Planner_RRT.generate_one_node(grid_cell) (:205-248), one step per episode: bucket_ids [E] =
 * (row*cols + col)*subsections + k of the chosen cell (<0 skips the episode).  mt/mt_index optional:
 * continue that generator for this step (the caller's global `random` state) */
int auvp_prrt_step(auvp_handle* h, const int32_t* bucket_ids, const uint32_t* mt, const int32_t* mt_index);
int auvp_prrt_summaries(auvp_handle* h, auvp_prrt_summary* out /* [E] */);
/* the goals of the resident batch, [E,2] (self.goal of every Planner_RRT, :40) */
int auvp_prrt_goals(auvp_handle* h, double* goals2);
/* generate_final_course(final_node) (:317-327) in planning()'s return order (goal end first);
 * offsets [E+1] prefix sums of path_len; out [offsets[E],5] = x,y,theta,traj_time_stamp,length */
int auvp_prrt_paths(auvp_handle* h, const int64_t* offsets, double* out);
/* one episode's tree: nodes4 [n,4] x,y,theta,traj_t; node_i4 [n,4] step,parent,pt_off,pt_cnt;
 * node_bucket [n]; points4 [n_points,4] x,y,theta,traj_t */
int auvp_prrt_tree(auvp_handle* h, int32_t episode, double* nodes4, int32_t* node_i4, int32_t* node_bucket,
                   double* points4);
/* one node of one episode (the node generate_one_node just appended to mps_list, gym_rrt/envs/rrt_dubins.py:236-240):
 * node4 x,y,theta,traj_t; node_i4 step,parent,pt_off,pt_cnt; its bucket; points4 [pt_cnt,4] (cap_points rows
 * available, AUVP_ERR_CAPACITY if the node has more).  O(path points of the node), not O(tree). */
int auvp_prrt_node(auvp_handle* h, int32_t episode, int32_t node, double* node4, int32_t* node_i4, int32_t* node_bucket,
                   double* points4, int32_t cap_points);
/* env_grid state: occupied_grid_cells_array as bucket ids, len(node_array) per bucket, dims4 =
 * rows, cols, subsections, n_occupied */
int auvp_prrt_grid(auvp_handle* h, int32_t episode, int32_t* occupied, int32_t* bucket_counts, int32_t* dims4);
/* per-step log (AUVP_FLAG_ITER_LOG): [max_step,8] bucket, picked, accepted, done, npath, arc_n, arc_free, new_node */
int auvp_prrt_step_log(auvp_handle* h, int32_t episode, int32_t* log8);
void* auvp_prrt_summaries_dev(auvp_handle* h);
/* which kernel the last auvp_prrt_plan / auvp_prrt_step launch ran: "prrt_kernel" (one episode per wavefront) or
 * "prrt_rows_kernel" (four per wavefront: throughput batches of the environment's planner shape, freq <= 15, <= 256
 * obstacles, no step log) */
const char* auvp_prrt_last_kernel(auvp_handle* h);
/* RRTEnv observation arrays (gym_rrt/envs/rrt_env.py:250-295: convert_rrt_grid_to_1D,
 * generate_rrt_grid_has_node_array, convert_rrt_grid_to_1D_num_of_nodes_only) for all episodes into
 * caller-owned DEVICE buffers: rrt_grid [E,n_buckets,4] f64 = cell.x, cell.y, subsection.theta,
 * len(node_array); has_node [E,n_buckets] i64; num_nodes [E,n_buckets] i64 (either may be NULL) */
int auvp_prrt_observation_dev(auvp_handle* h, void* rrt_grid_dev, void* has_node_dev, void* num_nodes_dev);
/* the same for one episode, copied to host arrays */
int auvp_prrt_observation(auvp_handle* h, int32_t episode, double* rrt_grid, int64_t* has_node, int64_t* num_nodes);
/* RRTEnv.step (gym_rrt/envs/rrt_env.py:182-247) for every environment with NOTHING crossing PCIe: bucket_ids_dev [E]
 * (chosen_grid_cell_idx per environment, -1 = skip; DEVICE memory, e.g. an agent's output) -> generate_one_node, then
 * the observation arrays (as auvp_prrt_observation_dev; rrt_grid_dev may be NULL to skip them) and the step's outcome:
 * reward_dev [E] i64 = 300 R_FOUND_PATH / 0 R_CREATE_NODE / -1 R_INVALID_NODE; 0 for an environment that had already
 * finished or that the caller skipped with -1 (neither is touched; a skipped one keeps its done flag), done_dev [E] u8
 * (may be NULL).  An episode that fails on the device (tree / point capacity, a bucket id outside the grid) is flagged
 * done, rewarded 0 and reported by auvp_prrt_env_check -- the host loop (auvp_prrt_step) shows the same failure as
 * summary.status < 0.  Two launches per step (the outcome is written by the planner launch itself, the observation arrays
 * by a second one; see AUVP_ENV_OBS_DELTA below for one).  Only ENQUEUES on the
 * handle's stream: follow with auvp_stream_sync (or order other work on auvp_stream) before reading results. */
int auvp_prrt_env_step_dev(auvp_handle* h, const int32_t* bucket_ids_dev, void* rrt_grid_dev, void* has_node_dev,
                           void* num_nodes_dev, int64_t* reward_dev, uint8_t* done_dev);
/* the same step with options (flags):
 *   AUVP_ENV_AGENT      the stand-in agent of auvp_prrt_policy_random_dev picks INSIDE the planner launch; bucket_ids_dev [E]
 *                       is then an OUTPUT (what it picked, -1 for finished environments)
 *   AUVP_ENV_OBS_DELTA  rrt_grid_dev / has_node_dev / num_nodes_dev hold the observation of the previous step (or of
 *                       auvp_prrt_observation_dev after the reset) and are UPDATED IN PLACE by the planner launch: a step adds
 *                       at most one node per environment, so one bucket's len(node_array) / has_node / node count changes
 *                       (rrt_env.py:250-295 rebuilds all three lists after every node) -- ONE launch per environment step
 * flags = 0 is auvp_prrt_env_step_dev. */
#define AUVP_ENV_AGENT 1
#define AUVP_ENV_OBS_DELTA 2
int auvp_prrt_env_step_ex_dev(auvp_handle* h, int32_t flags, uint64_t agent_seed, int32_t* bucket_ids_dev, void* rrt_grid_dev,
                              void* has_node_dev, void* num_nodes_dev, int64_t* reward_dev, uint8_t* done_dev);
/* stand-in agent for device-resident runs as a launch of its own: every live environment picks, uniformly, one of its
 * occupied buckets -- read from the planner's own list of occupied buckets, the set the observation's has_node array marks
 * (rrt_env.py:250-265); has_node_dev is accepted for compatibility, not read, and may be NULL -- finished environments get
 * -1; its randomness is its own, not the planner's stream: a pure function of (`seed`, environment, the environment's step
 * count in HBM), so a captured graph of one step draws anew at every replay, and calls repeated WITHOUT a step in between
 * (or for an environment whose step did not run) return the same pick -- vary `seed` to redraw.  Enqueues only. */
int auvp_prrt_policy_random_dev(auvp_handle* h, const int64_t* has_node_dev, uint64_t seed, int32_t* bucket_ids_dev);
/* waits for the stream, then: *status = status (< 0) of the first episode that failed on the device inside the
 * device-resident loop since the batch was created, *env (may be NULL) = its index; 0 = none */
int auvp_prrt_env_check(auvp_handle* h, int32_t* status, int32_t* env);
/* the handle's HIP stream (hipStream_t) and a wait for everything enqueued on it */
void* auvp_stream(auvp_handle* h);
int auvp_stream_sync(auvp_handle* h);
/* HIP-event time of a region of the handle's stream that the caller brackets around enqueue-only calls
 * (auvp_prrt_env_step_dev, auvp_graph_launch ...): mark(h, 0) before, mark(h, 1) after; elapsed_ms waits for the second
 * mark and returns the device time between the two (measurement infrastructure: no reference counterpart) */
int auvp_stream_mark(auvp_handle* h, int32_t which);
int auvp_stream_elapsed_ms(auvp_handle* h, double* ms);
/* hipGraph capture of a launch-bound step loop: everything the enqueue-only entry points (auvp_prrt_policy_random_dev,
 * auvp_prrt_env_step_dev, the caller's own kernels on auvp_stream) put on the stream between begin and end is recorded
 * instead of run; auvp_graph_launch replays it n_times back to back (enqueue only).  Every environment's step count lives in
 * HBM, so every replay of the stand-in agent draws anew.  Run one un-captured step first.  A graph holds the device
 * pointers of the batch (and of the caller's arrays) it was captured with: creating a new batch (auvp_prrt_create_batch,
 * auvp_prrt_replan_particles) destroys the handle's graphs, and auvp_graph_launch of such an id fails with AUVP_ERR_STATE. */
int auvp_graph_begin(auvp_handle* h);
int auvp_graph_end(auvp_handle* h, int32_t* graph_id);
int auvp_graph_launch(auvp_handle* h, int32_t graph_id, int32_t n_times);

/* ---------------------------------------------------------------------------------------------
 * A* variants (path_planning/astar.py, astar_real.py, astar_fixLen.py, astar_fixLenSOG.py), one
 * wavefront per search instance, E instances over the world of auvp_world_set (obstacles = obs_lst,
 * habitats = habitat_list, polygon = boundary_list corners, bins/cells/prob = sharkGrid).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t variant;   /* 0 astar.astar (astar.py:193), 1 astar_real (:144), 2 astar_fixLen (:286), 3 astar_fixLenSOG (:551) */
  int32_t cap_nodes; /* Node capacity per instance; overflow -> AUVP_ERR_CAPACITY in the instance's status */
  double box[4];     /* variant 0: boundary[0].x, boundary[0].y, boundary[1].x, boundary[1].y (astar.py:26-27) */
  double velocity;   /* variant 3: AUV_velocity (astar_fixLenSOG.py:111) */
  double w[4];       /* weights (w1 unused by the searches, as in the reference) */
} auvp_astar_params;

typedef struct {
  int32_t status;        /* 0 ok; <0: capacity, or a state in which the reference raises (IndexError / TypeError) */
  int32_t found;         /* 0: open list ran empty -> the reference returns None */
  int32_t n_nodes, n_expansions;
  int32_t n_children;    /* neighbour cells that passed the bounds test (the "cells/s" unit) */
  int32_t path_len, smooth_len, n_hab_left, visited_count, leaf;
  uint32_t open_scanned_lo, open_scanned_hi; /* sum over pops of len(open_list) (the min-f scan, astar.py:206-211) */
} auvp_astar_summary;

/* starts [E,2]; goals [E,2] (variants 0,1) or limits [E] = pathLenLimit (variants 2,3) */
int auvp_astar_batch(auvp_handle* h, int32_t n_instances, const double* starts, const double* goals,
                     const double* limits, const auvp_astar_params* params, int32_t flags);
int auvp_astar_summaries(auvp_handle* h, auvp_astar_summary* out /* [E] */);
/* offsets [E+1] prefix sums of path_len.  path3 [n,3] root->leaf x,y,round(time_stamp,2); cost_list [n]
 * leaf->root ("cost list"); node_path8 [n,8] root->leaf x,y,g,h,f,cost,pathLen,time_stamp ("node");
 * smooth3 [n,3] smoothPath output (variant 3; smooth_len rows per instance, re-read the summaries) */
int auvp_astar_paths(auvp_handle* h, const int64_t* offsets, double* path3, double* cost_list, double* node_path8,
                     double* smooth3);
/* AUVP_FLAG_ITER_LOG: the popped node of every expansion, [n_expansions,8] like node_path8 */
int auvp_astar_exp_log(auvp_handle* h, int32_t instance, double* out8);
/* variant 2: indices of the habitats still in the caller's habitat_list after the call (:310,:193-197) */
int auvp_astar_hab_left(auvp_handle* h, int32_t instance, int32_t* out);
/* self.visited_nodes (astar_fixLen.py:51 [550,600]; astar_fixLenSOG.py:117 [600,600]) persists across
 * astar() calls on one solver object in the reference: upload it before a batch launched with
 * AUVP_FLAG_KEEP_VISITED ([E][vx*600] bytes, row-major [x_pos][y_pos]) and read it back afterwards */
int auvp_astar_set_visited(auvp_handle* h, int32_t n_instances, int32_t variant, const uint8_t* bitmap);
int auvp_astar_get_visited(auvp_handle* h, int32_t instance, uint8_t* bitmap);

/* ---- shark occupancy / AUV detection grids ------------------------------------------------------
 * SharkOccupancyGrid.convert (path_planning/sharkOccupancyGrid.py:47-74): the producer of the
 * `sharkGrid` dict the A* SOG variant and createSharkGrid consume.  cells [C,4] = bounds of
 * cell_list (splitCell output, :376-393) in list order; box = boundary.bounds; traj_len [S] points
 * per shark in dict order; pts [sum(traj_len),3] = x, y, traj_time_stamp.  Outputs: n_bins
 * (createBinList :306-319), rows/cols (:134), bins [n_bins,2], grids [n_bins,rows,cols] = resultArr.
 * A cell whose index falls outside the grid is the reference's IndexError -> AUVP_ERR_ARG.
 * Call once with grids == NULL (or cap_bins == 0 -> AUVP_ERR_CAPACITY with the sizes filled in) to size
 * the output. */
int auvp_sog_convert(auvp_handle* h, const double* cells, int32_t n_cells, const double* box4, double cell_size,
                     double bin_interval, double detect_range, int32_t n_sharks, const int32_t* traj_len,
                     const double* pts_xyt, int32_t cap_bins, int32_t* n_bins, int32_t* rows, int32_t* cols,
                     double* bins, double* grids);

/* ---- shark particle filters ------------------------------------------------------------------------
 * particleFilter.py driven like robotSim.py:665-701, F independent filters x N particles (N <= 2048),
 * one workgroup per filter.  The reference draws from numpy's global legacy RandomState (its `random`
 * is numpy.random, particleFilter.py:8): each filter carries an MT19937 key [624] + position, handed
 * in and read back so a caller can continue numpy's own stream (np.random.get_state / set_state).
 * Particles are rows {x_p, y_p, v_p, theta_p, weight_p}; `obj` is the object id of a list position
 * (positions that hold the same Particle object after `correct` share an id and are moved once per
 * position by create_and_update, like the reference). */
enum { AUVP_PF_UPDATE = 1,  /* ParticleFilter.create_and_update (:277-282) */
       AUVP_PF_WEIGHTS = 2, /* ParticleFilter.update_weights (:285-310): weight, normalize, correct */
       AUVP_PF_MEAN = 4 };  /* particleMean + meanError (:153-177) */
/* ParticleFilter(x, y, ...).create() for F filters (:311-317, Particle.__init__ :44-53) */
int auvp_pf_create_batch(auvp_handle* h, int32_t n_filters, int32_t n_particles, const double* shark_xy0,
                         const uint32_t* mt_key, const int32_t* mt_pos);
/* start from caller-supplied lists instead: particles [F,N,5], obj [F,N] or NULL (all distinct),
 * list_len [F] or NULL */
int auvp_pf_set_particles(auvp_handle* h, int32_t n_filters, int32_t n_particles, const double* particles,
                          const int32_t* obj, const int32_t* list_len, const uint32_t* mt_key, const int32_t* mt_pos);
int auvp_pf_set_rng(auvp_handle* h, const uint32_t* mt_key, const int32_t* mt_pos);
/* n_steps x the phases selected (AUVP_PF_* mask, in the order update, weights, mean) in ONE launch.
 * meas [S,F,n_auv,5] = list_of_range_bearing rows {x, y, theta, [3], [4]} (update_weights passes [3] as
 * the bearing and [4] as the range argument of Particle.weight, :293-295); shark_xy [S,F,2] =
 * self.x_shark / y_shark for meanError.  AUVP_FLAG_ITER_LOG records the per-step log below.
 * Per-filter status: 1 = an angle_wrap recursion deeper than CPython allows / nan, 2 = empty
 * list_of_new_particles (numpy raises ValueError). */
int auvp_pf_run(auvp_handle* h, int32_t n_steps, int32_t n_auv, int32_t phases, const double* meas,
                const double* shark_xy, int32_t flags);
int auvp_pf_particles(auvp_handle* h, double* particles, int32_t* obj);
/* device-resident state for a caller that keeps working on the GPU: SoA [F][5][N] doubles, ids [F][N] */
int auvp_pf_particles_dev(auvp_handle* h, double** particles_soa, int32_t** obj);
/* of the last auvp_pf_run: mean [S,F,2], range_error [S,F], len(list_of_new_particles) [S,F] */
int auvp_pf_estimates(auvp_handle* h, double* mean, double* range_error, int32_t* list_len);
int auvp_pf_status(auvp_handle* h, int32_t* status, uint64_t* n_draw32);
int auvp_pf_rng_state(auvp_handle* h, uint32_t* mt_key, int32_t* mt_pos);
/* AUVP_FLAG_ITER_LOG: updated [S,F,N,5] (the list after create_and_update), choice [S,F,N] (the indices
 * random.choice drew in correct, :248-250) */
int auvp_pf_step_log(auvp_handle* h, double* updated, int32_t* choice);

/* standalone evaluations on the device (parity probes for the building blocks) */
/* RRT.check_collision (:530-549) of n_paths paths; pts [sum(npts),2], path i = pts[off[i]:off[i+1]] */
int auvp_check_collision_batch(auvp_handle* h, int32_t n_paths, const int32_t* off, const double* pts_xy,
                               int8_t* out_free);
/* habitat_shark_cost_func (path_planning/cost.py:145-207) over bins [bin_lo,bin_hi) */
int auvp_cost_paths(auvp_handle* h, int32_t n_paths, const int32_t* off, const double* pts_xyt,
                    const int32_t* bin_lo, const int32_t* bin_hi, const double* total_traj_time,
                    const double* weights3, double* out4);
/* RRT.get_closest_mps (path_planning/rrt_dubins.py:505-513) for n_queries sample points against one node list xy
 * [n_nodes,2]: index of the FIRST node with the smallest RN(sqrt(dx**2 + dy**2)), by the planner's streaming scan.
 * force_exact != 0 ranks with a sqrt per node (the scan's fallback path); out_slow (optional) reports per query whether
 * that path ran */
int auvp_nn_closest_batch(auvp_handle* h, int32_t n_nodes, const double* xy, int32_t n_queries, const double* q,
                          int32_t force_exact, int32_t* out_index, int32_t* out_slow);
/* Measured HBM streaming rates of this GPU (the roof bench.py quotes its bandwidth fractions against, beside the 8 TB/s
 * datasheet figure): a read of `bytes` (>= 64 MB; use >= 4 GB to get past the 256 MB Infinity Cache) with eight 16-byte
 * loads in flight per lane -- the access shape of the nearest-neighbour scan of get_closest_mps
 * (path_planning/rrt_dubins.py:505-513) -- and a 16-byte copy of one half of the buffer onto the other; best of `reps`
 * launches each, HIP events on the handle's stream.  GB/s = 1e9 bytes per second; copy counts bytes read + written.
 * (No reference counterpart: measurement infrastructure of the boundary.) */
int auvp_hbm_probe(auvp_handle* h, uint64_t bytes, int32_t reps, double* read_GBps, double* copy_GBps);
/* portable sin/cos evaluated on the device (bit-exactness probe for auvp_math.h) */
int auvp_sincos_dev(auvp_handle* h, int32_t n, const double* x, double* s, double* c);
/* one portable elementary function of auv_sim_amd/csrc/auvp_math.h / auvp_exp.h evaluated on the device over n operand pairs
 * (bit-exactness probe: the tests compare with the host build of the same header and with the IEEE operation).  op: 0 sin -> out0,
 * cos -> out1 of a (math.sin / math.cos, path_planning/rrt_dubins.py:275-276); 1 atan2(a, b) (gym_rrt/envs/rrt_dubins.py:387);
 * 2 math.e ** a (particleFilter.py:103-116); 3 the steer's short division a / b and 4 its square root (rrt_dubins.py:268-281;
 * operands of unexceptional magnitude: auvp_div_plain / auvp_sqrt_plain); 5 hypot(a, b); 6 a / b and 7 sqrt(a) as compiled. */
int auvp_math_dev(auvp_handle* h, int32_t op, int32_t n, const double* a, const double* b, double* out0, double* out1);
/* CPython random() stream of `seed` generated by the wave-level device MT19937 */
int auvp_random_stream_dev(auvp_handle* h, uint64_t seed, int32_t n, double* out);

/* diagnostic: shader clocks each episode spent in {parent selection, steer, collision, accept, cost
 * walk} (needs AUVP_FLAG_PHASE_CLOCKS); out [E,5] */
int auvp_rrt_phase_clocks(auvp_handle* h, uint64_t* out);
/* HIP-event time (ms) of the last batch kernel on the handle's stream, and its launch geometry */
double auvp_last_kernel_ms(auvp_handle* h);
/* of the last auvp_rrt_run: HIP-event times of its two launches (tree expansion; leaf ranking) and which expansion
 * kernel ran (4 = four episodes per wavefront, rrt_rows_kernel; 1 = rrt_explore_kernel) */
int auvp_rrt_last_launch_parts(auvp_handle* h, double* expand_ms, double* leaf_ms, int32_t* episodes_per_wave);
/* ... and of the launch that generated the episodes' random numbers ahead of it (rrt_stream_kernel; 0 when the last auvp_rrt_run
 * ran the generator inside the expansion kernel).  auvp_last_kernel_ms covers all three launches. */
double auvp_rrt_last_stream_ms(auvp_handle* h);
/* ... and how many random() numbers per episode that launch wrote (8 bytes each; 0: none) */
int64_t auvp_rrt_last_stream_len(auvp_handle* h);
/* name of the expansion kernel the last auvp_rrt_run launched: "rrt_rows_kernel" (four episodes per wavefront: batches of
 * more than 18 episodes per CU; "rrt_rows_stream_kernel": the same with the random numbers generated ahead by
 * rrt_stream_kernel), "rrt_explore_kernel" (one), "rrt_duo_kernel" (two wavefronts per episode: batches of at
 * most four episodes per CU in time-bin mode -- the helper wavefront produces the half of an iteration that depends only on
 * the random stream, rrt_dubins.py:121-127,252-281, one iteration ahead) */
const char* auvp_rrt_last_kernel(auvp_handle* h);
/* of the last auvp_rrt_run's leaf pass (the qualifying-leaf bookkeeping of exploring, rrt_dubins.py:158-171 +
 * cost.py:145-207), summed over the batch: out4 = {nodes its sweep visited (qualifying leaves and their ancestors), path
 * points of those nodes (each evaluated once), path elements re-summed in the reference's leaf->root order, leaves
 * re-summed}: the units of that kernel's compulsory-traffic figure */
int auvp_rrt_last_leaf_stats(auvp_handle* h, int64_t* out4);
int auvp_last_launch(auvp_handle* h, int32_t* grid, int32_t* block, int32_t* lds_bytes);

/* Config 5 composition (BASELINE.json configs[4]): the particle filter of particleFilter.py:283-317 feeding one
 * Planner_RRT replan per particle (gym_rrt/envs/rrt_dubins.py:162-248), without leaving the device.
 * Needs a filter batch on this handle (auvp_pf_create_batch / auvp_pf_run).  Episode e = f * N + p plans from the
 * shared start4 (x, y, theta, traj_time_stamp) to goal = clamp(particle(f, p).xy * scale_f + offset_f):
 *   xform [F,4] = sx, ox, sy, oy per filter (gx = x * sx + ox, gy = y * sy + oy);   clamp4 = x0, y0, x1, y1
 * with the generator of random.seed(seed_base + e), seeded on the device.  Replaces Planner_RRT.__init__ per
 * particle; follow with auvp_prrt_plan() and the auvp_prrt_* readers. */
int auvp_prrt_replan_particles(auvp_handle* h, const double* start4, const auvp_prrt_params* params, const double* xform,
                               const double* clamp4, uint64_t seed_base, int32_t flags);

/* ---------------------------------------------------------------------------------------------------
 * Multi-GPU result gather (SURVEY.md 8(b) "auvp_gather(comm...)", 8(e)).  The reference is single-process;
 * what these replace is the implicit "results are in this process" of its batch callers
 * (path_planning/performance.py:65-73 collects one result per exploring() call in a Python list).  One
 * process per GPU plans a block of the episode index; afterwards every rank holds every record.
 * Collectives are RCCL (bound at run time) on the handle's stream; all buffers are DEVICE pointers.
 *   auvp_comm_unique_id  rank 0 creates the 128-byte id and hands it to the other ranks (any channel)
 *   auvp_comm_init       every rank, same id: joins the communicator (collective)
 *   auvp_gather          all-gather of equally sized blocks: recv_dev [world][bytes_per_rank]
 *   auvp_gather_var      variable-size blocks, two-phase: counts_out[world] always receives every rank's byte
 *                        count; with recv_dev == NULL only that (size query), else the blocks are written back
 *                        to back in rank order.  Whether the payload phase runs is decided COLLECTIVELY (every rank
 *                        also publishes its capacity): it runs only if every rank passed a buffer that holds
 *                        sum(counts); otherwise every rank returns after the counts (AUVP_ERR_CAPACITY on the ranks
 *                        that passed a buffer), none is left waiting
 *   auvp_gather_counts / auvp_gather_blocks   the two phases as separate calls, for a caller that sizes its
 *                        buffer between them (one count exchange instead of two); counts must be the array
 *                        auvp_gather_counts returned, unchanged, on every rank
 *   auvp_comm_available  1 if an RCCL can be bound in this process (no communicator, no bootstrap thread is created)
 *   auvp_comm_library    path of the RCCL image that was bound (an already mapped one -- PyTorch's -- is preferred
 *                        over opening a second image), or the reason none could be
 *   auvp_comm_info       communicator facts for self-checking runs: world size given, rank and rank count as RCCL
 *                        reports them (ncclCommUserRank, ncclCommCount)
 *   auvp_last_gather_ms  HIP-event time of the last gather on the handle's stream */
#define AUVP_COMM_ID_BYTES 128
int auvp_comm_unique_id(uint8_t* id_out /* [AUVP_COMM_ID_BYTES] */);
int auvp_comm_init(auvp_handle* h, int32_t world_size, int32_t rank, const uint8_t* id);
int auvp_comm_destroy(auvp_handle* h);
int auvp_gather(auvp_handle* h, const void* send_dev, size_t bytes_per_rank, void* recv_dev);
int auvp_gather_var(auvp_handle* h, const void* send_dev, int64_t send_bytes, void* recv_dev, int64_t recv_cap_bytes,
                    int64_t* counts_out);
int auvp_gather_counts(auvp_handle* h, int64_t send_bytes, int64_t* counts_out /* [world] */);
int auvp_gather_blocks(auvp_handle* h, const void* send_dev, void* recv_dev, int64_t recv_cap_bytes,
                       const int64_t* counts /* [world] */);
/* Gather to ONE rank -- what north_star asks for ("an RCCL gather of final paths"); the all-gathers above hand every rank every
 * record.  rank r's block lands at the prefix offset of r in the ROOT's recv_dev (the other ranks pass NULL / 0): one group of
 * ncclSend / ncclRecv, every peer over its own xGMI link.  `counts` [world] as auvp_gather_counts returned them, or computed
 * by the caller where the block sizes are known (fixed-stride records of a block partition).
 *   auvp_gather_blocks_root        the transfer, waited for
 *   auvp_gather_blocks_root_async  the same ENQUEUED on the handle's gather stream -- a second stream, ordered behind what the
 *                                  planner stream holds at the call -- so that the transfer of step k overlaps the kernels
 *                                  of step k + 1.  May be called several times (records, lengths, paths); send_dev and the
 *                                  root's recv_dev must stay untouched until auvp_gather_wait.  No other collective of this
 *                                  handle may be issued while transfers are pending.
 *   auvp_gather_wait               waits for the pending transfers; ms_out = their HIP-event time on the gather stream (first
 *                                  enqueue to completion, waiting for the planner stream's event included), bytes_out = what
 *                                  this rank sent (on the root: received, its own block included).  No-op without pending ones.
 * An RCCL without ncclSend / ncclRecv: AUVP_ERR_COMM (world size 1 needs neither). */
int auvp_gather_blocks_root(auvp_handle* h, int32_t root, const void* send_dev, void* recv_dev, int64_t recv_cap_bytes,
                            const int64_t* counts /* [world] */);
int auvp_gather_blocks_root_async(auvp_handle* h, int32_t root, const void* send_dev, void* recv_dev, int64_t recv_cap_bytes,
                                  const int64_t* counts /* [world] */);
int auvp_gather_wait(auvp_handle* h, double* ms_out, int64_t* bytes_out);
int auvp_comm_available(void);
const char* auvp_comm_library(void);
int auvp_comm_info(auvp_handle* h, int32_t* world_size, int32_t* rank, int32_t* rccl_ranks_seen);
double auvp_last_gather_ms(auvp_handle* h);

#ifdef __cplusplus
}
#endif
#endif
