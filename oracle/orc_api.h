/* orc_api.h -- C interface of the CPU checker (restatement of the reference planners).
 *
 * TEST INFRASTRUCTURE (oracle/): a plain-C restatement of the reference algorithm for the hot path,
 * pinned against tests/golden/ (vectors captured from the reference itself, tests/golden/
 * make_golden.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * it, and only as the checker / the reported CPU baseline.  The product (auv_sim_amd/) never links,
 * imports or falls back to anything in this directory.
 */
#ifndef ORC_API_H
#define ORC_API_H
#include <stdint.h>

typedef struct {
  int32_t n_obstacles, n_habitats, n_poly, n_bins, n_cells, _pad;
  const double* obstacles; /* [O,3] x,y,r  (list order matters: prefix-min quirk) */
  const double* habitats;  /* [H,3] x,y,r */
  const double* polygon;   /* [V,2] boundary polygon vertices, or rect {x0,y0,x1,y1} for Planner_RRT */
  const double* bins;      /* [T,2] shark-grid time bins (t0,t1), dict order */
  const double* cells;     /* [C,4] minx,miny,maxx,maxy, cell_list order */
  const double* prob;      /* [T,C] */
} orc_world;

enum { ORC_MODE_TIMEBIN = 0, ORC_MODE_PLANTIME = 1, ORC_MODE_NN = 2 };

typedef struct {
  double init[6];        /* x, y, theta, traj_time_stamp, plan_time_stamp, length of `initial` */
  double dist_to_end, diff_max, freq, min_dist;
  double bin_interval, v, max_traj_time, max_plan_time;
  double w[3];
  int32_t mode, max_iter;
} orc_rrt_params;

typedef struct {
  /* capacities (in) */
  int32_t cap_nodes, cap_points, cap_leaves, cap_bins;
  /* counts (out) */
  int32_t n_nodes, n_points, n_leaves, n_bins, iters_run, best_leaf, status, _pad;
  double best_cost[4];
  double best_length;
  double rng_after;
  uint64_t n_draw32;
  /* arrays (caller allocated) */
  double* nodes;       /* [cap_nodes,6] x,y,theta,traj_t,plan_t,length */
  int32_t* parent;     /* [cap_nodes] */
  int32_t* pt_off;     /* [cap_nodes] first appended path point of the node */
  int32_t* pt_cnt;     /* [cap_nodes] number of appended path points (len(path)-1) */
  double* points;      /* [cap_points,7] x,y,theta,v,traj_t,plan_t,length */
  int32_t* it_parent;  /* [max_iter] parent picked (-1: iteration skipped by `continue`) */
  int8_t* it_accepted; /* [max_iter] */
  int32_t* it_npath;   /* [max_iter] len(new.path) */
  double* leaf_cost;   /* [cap_leaves,6] total,c0,c1,c2,len(path),len(shark sub-dict) */
  int32_t* leaf_iter;  /* [cap_leaves] */
  int32_t* bin_sizes;  /* [cap_bins] len(time_bin[bin_interval*(i+1)]) */
} orc_rrt_out;

enum { ORC_OK = 0, ORC_NO_QUALIFYING_LEAF = 1, ORC_ERR_CAPACITY = -1, ORC_ERR_ARG = -2 };

/* Planner_RRT (gym_rrt/envs/rrt_dubins.py) */
typedef struct {
  double start[4];  /* x, y, theta, traj_time_stamp */
  double goal[2];
  double rect[4];   /* boundary corners x0,y0,x1,y1 */
  double exp_rate, dist_to_end, diff_max, freq, cell_side_length;
  int32_t subsections, max_step;
} orc_prrt_params;

typedef struct {
  int32_t cap_nodes, cap_points, cap_path, cap_buckets;                       /* in */
  int32_t n_nodes, n_points, steps, done, status, n_occ, n_buckets, grid_rows, grid_cols, path_len; /* out */
  double rng_after;
  uint64_t n_draw32;
  double* nodes;          /* [cap_nodes,5] x,y,theta,traj_t,length */
  int32_t* parent;        /* [cap_nodes] */
  int32_t* pt_off;
  int32_t* pt_cnt;
  int32_t* node_bucket;   /* [cap_nodes] bucket id or -1 */
  double* points;         /* [cap_points,4] x,y,theta,traj_t */
  int32_t* st_bucket;     /* per step [max_step] */
  int32_t* st_picked;
  int8_t* st_accepted;
  int8_t* st_done;
  int32_t* st_npath;
  int32_t* st_arc_n;      /* -1: connect_to_goal returned None */
  int8_t* st_arc_free;
  int32_t* occupied;      /* [cap_nodes] occupied_grid_cells_array as bucket ids */
  int32_t* bucket_counts; /* [cap_buckets] */
  double* path;           /* [cap_path,5] generate_final_course order (goal end first) */
} orc_prrt_out;

/* A* variants (path_planning/astar*.py); world: obstacles, habitats, polygon, bins, cells, prob */
typedef struct {
  int32_t variant;   /* 0 astar, 1 astar_real, 2 astar_fixLen, 3 astar_fixLenSOG */
  int32_t cap_nodes;
  double start[2], goal[2];
  double box[4];     /* variant 0: min_bound.x, min_bound.y, max_bound.x, max_bound.y */
  double limit, velocity;
  double w[4];
} orc_astar_params;

typedef struct {
  int32_t cap_exp, cap_path;                                       /* in */
  int32_t n_nodes, n_expansions, n_children, found, status, path_len, smooth_len, n_hab_left, visited_count, _pad;
  double* exp_log;     /* [cap_exp,8] popped node x,y,g,h,f,cost,pathLen,time_stamp */
  double* path;        /* [cap_path,3] root -> leaf: x, y, round(time_stamp,2) */
  double* cost_list;   /* [cap_path] leaf -> root */
  double* node_path;   /* [cap_path,8] root -> leaf node fields */
  double* smooth_path; /* [cap_path,3] variant 3 */
  int32_t* hab_left;   /* [H] indices of the habitats still in habitat_open_list (variant 2) */
} orc_astar_out;

/* SharkOccupancyGrid.convert (path_planning/sharkOccupancyGrid.py:47) */
typedef struct {
  int32_t n_cells, n_sharks;
  double box[4];            /* boundary.bounds */
  double cell_size, bin_interval, detect_range;
  const double* cells;      /* [C,4] bounds, cell_list order */
  const int32_t* traj_len;  /* [S] points per shark (dict order) */
  const double* pts;        /* [sum,3] x, y, traj_time_stamp */
} orc_sog_in;

typedef struct {
  int32_t cap_bins, n_bins, rows, cols;
  double* bins;     /* [cap_bins,2] */
  double* grids;    /* [cap_bins,rows,cols] AUV detection grid per bin (resultArr) */
  double* occ_dbg;  /* [rows,cols] constructSharkOccupancyGrid of (bin 0, shark 0), optional */
  double* auv_dbg;  /* [rows,cols] constructAUVGrid of the same, optional */
} orc_sog_out;

/* ParticleFilter (particleFilter.py) driven like robotSim.py:665-701: create(), then per step
 * create_and_update, update_weights (normalize, correct), particleMean, meanError; numpy legacy
 * RandomState stream (MT19937 key + pos in/out) */
typedef struct {
  int32_t n_particles, n_steps, n_auv, do_create;
  double shark0[2];        /* ParticleFilter(init_x_shark, init_y_shark, ...) */
  const double* meas;      /* [n_steps,n_auv,5] list_of_range_bearing rows: x, y, theta, [3], [4] */
  const double* shark_xy;  /* [n_steps,2] self.x_shark / self.y_shark when meanError runs */
  const double* init;      /* [N,5] x, y, v, theta, weight when !do_create */
  const int32_t* init_obj; /* [N] object id of each list entry when !do_create (aliases share an id), or NULL */
  uint32_t* mt;            /* [624] in/out */
  int32_t* mt_pos;         /* in/out */
} orc_pf_in;

typedef struct {
  double* created;      /* [N,5] (do_create) */
  double* updated;      /* [n_steps,N,5] after create_and_update */
  double* resampled;    /* [n_steps,N,5] the list update_weights returns */
  int32_t* choice;      /* [n_steps,N] index drawn by random.choice(len(list_of_new_particles)) */
  int32_t* alias_first; /* [n_steps,N] first list position holding the same object */
  int32_t* list_len;    /* [n_steps] len(list_of_new_particles) */
  double* mean;         /* [n_steps,2] */
  double* range_error;  /* [n_steps] */
  uint64_t n_draw32;
} orc_pf_out;

#ifdef __cplusplus
extern "C" {
#endif
int orc_pf_run(const orc_pf_in* in, orc_pf_out* out);
void orc_np_kat(uint32_t seed, int n, double* out_uniform, int32_t* out_choice, int32_t choice_n);
double orc_pow_e(double z);
int orc_sog_convert(const orc_sog_in* in, orc_sog_out* out);
int orc_astar_run(const orc_world* w, const orc_astar_params* p, orc_astar_out* o);
const char* orc_math_name(void);
double orc_sin(double x);
double orc_cos(double x);
double orc_atan2(double y, double x);
double orc_hypot(double x, double y);

void orc_rng_kat(uint64_t seed, int n, double* out_random, uint32_t* out_bits32, int nchoice,
                 uint32_t choice_n, uint32_t* out_choice);

int orc_check_collision(const orc_world* w, int npts, const double* pts_xy);
void orc_cost(const orc_world* w, int bin_lo, int bin_hi, int npts, const double* pts_xyt,
              double total_traj_time, const double* weights, double* out4);
int orc_rrt_explore(const orc_world* w, const orc_rrt_params* p, uint64_t seed, orc_rrt_out* out);
/* leaf -> root concatenation (generate_final_course), then reversed to root -> leaf order.
 * returns number of elements written (<= cap), or -needed if cap too small */
int orc_rrt_final_course(const orc_rrt_out* t, const double* init7, int leaf, double* path7, int cap);
/* Planner_RRT(...).planning(max_step) with `random.seed(seed)`; w carries the obstacle list only */
int orc_prrt_planning(const orc_world* w, const orc_prrt_params* p, uint64_t seed, orc_prrt_out* out);
#ifdef __cplusplus
}
#endif
#endif
