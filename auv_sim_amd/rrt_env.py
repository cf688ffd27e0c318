"""Batched counterpart of gym_rrt/envs/rrt_env.py::RRTEnv for E planning episodes at once.

The reference env runs one Planner_RRT and, after every node, rebuilds three O(#buckets) Python lists
(`convert_rrt_grid_to_1D`, `generate_rrt_grid_has_node_array`, `convert_rrt_grid_to_1D_num_of_nodes_only`,
rrt_env.py:227-231,250-295).  Here the E planners, their bucket grids and the observation arrays live on
the MI355X; `step(cells, step_num)` is one `auvp_prrt_step` launch + one elementwise observation kernel
and hands the observations back as numpy arrays (or leaves them in caller-owned device memory).

step() contract per episode (rrt_env.py:182-247): chosen_grid_cell_idx -> (cell, subsection)
= divmod(idx, num_of_subsections); reward R_FOUND_PATH (300) when the goal arc is free, R_CREATE_NODE (0)
when a node was added, R_INVALID_NODE (-1) otherwise.  Episodes that are done are skipped.
RNG: seeded per episode (`seeds[e]` plays random.seed), drawn on the device; `seeds=None` runs ONE
episode on Python's global `random` stream exactly like the reference env (the stream is handed to the
device for each step and advanced by what the step consumed).
"""
import random
import ctypes as C

import numpy as np

from . import _lib
from ._prrt_lib import PlannerBatch, _bind

R_FOUND_PATH = 300
R_CREATE_NODE = 0
R_INVALID_NODE = -1
RRT_PLANNER_FREQ = 10


class RRTEnvBatch:
    def __init__(self, auv_init_pos, shark_init_pos, boundary_array, grid_cell_side_length, num_of_subsections,
                 obstacle_array=(), seeds=(0,), max_nodes=2048, freq=RRT_PLANNER_FREQ, device=0):
        """auv_init_pos / shark_init_pos: one Motion_plan_state (shared) or a list of E of them"""
        self.global_stream = seeds is None
        self.E = 1 if seeds is None else len(seeds)
        self.seeds = None if seeds is None else np.asarray(seeds, dtype=np.uint64)
        self._ctx = _lib.Context(device)
        self._obst = np.array([(float(o.x), float(o.y), float(o.size)) for o in obstacle_array], dtype=np.float64).reshape(-1, 3)
        self.obstacle_array = np.array([[o.x, o.y, o.z, o.size] for o in obstacle_array])
        self._rect = (float(boundary_array[0].x), float(boundary_array[0].y), float(boundary_array[1].x), float(boundary_array[1].y))
        self.cell_side_length = grid_cell_side_length
        self.num_of_subsections = num_of_subsections
        self.max_nodes = int(max_nodes)
        self.freq = freq

        def _many(m):
            return list(m) if isinstance(m, (list, tuple)) else [m] * self.E

        self._auv, self._shark = _many(auv_init_pos), _many(shark_init_pos)
        self._L = _bind()
        self._L.auvp_prrt_observation.argtypes = [C.c_void_p, C.c_int32, _lib._dp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        self._L.auvp_prrt_observation_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self._L.auvp_prrt_env_step_dev.argtypes = [C.c_void_p] * 7
        self._L.auvp_prrt_env_step_ex_dev.argtypes = [C.c_void_p, C.c_int32, C.c_uint64] + [C.c_void_p] * 6
        self._L.auvp_prrt_env_check.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        self._L.auvp_prrt_policy_random_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        self._L.auvp_stream_sync.argtypes = [C.c_void_p]
        self._L.auvp_graph_begin.argtypes = [C.c_void_p]
        self._L.auvp_graph_end.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        self._L.auvp_graph_launch.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        self._pb = None
        self.state = None

    @property
    def n_buckets(self):
        return self._pb.rows * self._pb.cols * self._pb.subs

    def reset(self):
        """rrt_env.py:410-449 for every episode"""
        self._ctx.set_world(obstacles=self._obst)
        starts = [[float(a.x), float(a.y), float(a.theta), float(a.traj_time_stamp)] for a in self._auv]
        goals = [[float(s.x), float(s.y)] for s in self._shark]
        kw = dict(freq=self.freq, cell=self.cell_side_length, subs=self.num_of_subsections)
        if self.global_stream:
            self._pb = PlannerBatch(self._ctx, starts, goals, self._rect, self.max_nodes, mt_states=self._mt_state(), **kw)
        else:
            self._pb = PlannerBatch(self._ctx, starts, goals, self._rect, self.max_nodes, seeds=self.seeds, **kw)
        self._done = np.zeros(self.E, dtype=bool)
        self._dev = None
        self._graphs = set()   # the new batch invalidated the handle's captured graphs (they replay on the old buffers)
        self._mode = None      # "host" (step) or "device" (step_device / replay): one per reset(), see _enter
        self.state = self._observe(None)
        return self.state

    @staticmethod
    def _mt_state():
        ver, internal, _ = random.getstate()
        return np.array(internal[:624], dtype=np.uint32).reshape(1, 624), np.array([internal[624]], dtype=np.int32)

    def _observe(self, summ):
        nb = self.n_buckets
        # device tensors are torch's job (allocation only); one elementwise pass fills all episodes
        import torch
        dev = torch.device("cuda", self._ctx.device)
        if getattr(self, "_obs_dev", None) is None:
            self._obs_dev = (torch.empty((self.E, nb, 4), dtype=torch.float64, device=dev),
                             torch.empty((self.E, nb), dtype=torch.int64, device=dev),
                             torch.empty((self.E, nb), dtype=torch.int64, device=dev))
        g, h, n = self._obs_dev
        self.observation_to_device(g.data_ptr(), h.data_ptr(), n.data_ptr())
        # fresh host arrays every step (the caller may keep them), in page-locked memory from torch's caching host allocator: the
        # 39 MB of a 512-environment observation cross PCIe at the link's rate instead of through a pageable staging copy
        host = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in (g, h, n)]
        for dst, src in zip(host, (g, h, n)):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize(dev)
        grid, has, num = (t.numpy() for t in host)
        st = {
            "auv_pos": np.array([[a.x, a.y, a.z, a.theta] for a in self._auv], dtype=np.float64),
            "shark_pos": np.array([[s.x, s.y, s.z, s.theta] for s in self._shark], dtype=np.float64),
            "obstacles_pos": self.obstacle_array,
            "rrt_grid": grid, "has_node": has, "rrt_grid_num_of_nodes_only": num,
            "path": [None] * self.E if self.state is None else self.state["path"],
        }
        return st

    def observation_to_device(self, rrt_grid_ptr, has_node_ptr=None, num_nodes_ptr=None):
        """write the observation arrays of all episodes into caller-owned device memory (e.g. torch
        tensors' data_ptr()): rrt_grid [E,n_buckets,4] f64, has_node / num_nodes [E,n_buckets] i64"""
        self._ctx._chk(self._L.auvp_prrt_observation_dev(self._ctx.h, C.c_void_p(rrt_grid_ptr),
                                                         C.c_void_p(has_node_ptr) if has_node_ptr else None,
                                                         C.c_void_p(num_nodes_ptr) if num_nodes_ptr else None))

    def _enter(self, mode):
        """The host loop keeps its finished flags and observation arrays on the host, the device-resident loop keeps its own
        in HBM; neither updates the other, so one episode (one reset()) runs in ONE of the two modes."""
        if self._pb is None:
            raise RuntimeError("reset() first")
        if self._mode is None:
            self._mode = mode
        elif self._mode != mode:
            raise RuntimeError("RRTEnvBatch: %s stepping after %s stepping in the same episode; call reset() to switch"
                               % (mode, self._mode))

    def step(self, chosen_grid_cell_idx, step_num=None):
        """chosen_grid_cell_idx [E] (RRTEnv.step's flat index over cells x subsections).
        Returns (state, reward [E], done [E], {})"""
        self._enter("host")
        idx = np.asarray(chosen_grid_cell_idx, dtype=np.int64).reshape(self.E)
        # flat index -> bucket id: (row*cols + col)*S + k is the same flattening RRTEnv uses (:203-213)
        buckets = np.where(self._done, -1, idx).astype(np.int32)
        before = self._pb.summaries()
        if self.global_stream:
            summ = self._pb.step(buckets, mt_states=self._mt_state())
            n = int(summ[0]["n_draw32"]) if buckets[0] >= 0 else 0
            if n:
                random.getrandbits(32 * n)
        else:
            summ = self._pb.step(buckets)
        bad = summ["status"] < 0
        if bad.any():
            raise _lib.AuvpError(int(summ["status"][bad][0]), "planner episode failed on the device")
        active = ~self._done
        done_now = (summ["done"] != 0) & active
        created = (summ["n_nodes"] > before["n_nodes"]) & active
        reward = np.where(done_now, R_FOUND_PATH, np.where(created, R_CREATE_NODE, R_INVALID_NODE)).astype(np.int64)
        reward[~active] = 0
        paths = self._pb.paths(summ) if done_now.any() else None
        self.state = self._observe(summ)
        for e in range(self.E):
            if done_now[e]:
                self.state["path"][e] = paths[e]
            elif created[e]:
                t = int(summ[e]["last_new_node"])
                self.state["path"][e] = t  # id of the node added this step (tree stays on the device)
        self._done |= done_now
        return self.state, reward, self._done.copy(), {}

    # ---- the device-resident loop: nothing of a step crosses PCIe (auvp_prrt_env_step_dev) ----
    def device_buffers(self):
        """torch device tensors the loop writes: rrt_grid [E,nb,4] f64, has_node / num_nodes [E,nb] i64, reward [E] i64,
        done [E] u8, bucket [E] i32 (the agent's choice).  Allocated once per reset()."""
        import torch
        if getattr(self, "_dev", None) is None:
            dev = torch.device("cuda", self._ctx.device)
            nb = self.n_buckets
            self._dev = dict(rrt_grid=torch.empty((self.E, nb, 4), dtype=torch.float64, device=dev),
                             has_node=torch.empty((self.E, nb), dtype=torch.int64, device=dev),
                             num_nodes=torch.empty((self.E, nb), dtype=torch.int64, device=dev),
                             reward=torch.zeros(self.E, dtype=torch.int64, device=dev),
                             done=torch.zeros(self.E, dtype=torch.uint8, device=dev),
                             bucket=torch.zeros(self.E, dtype=torch.int32, device=dev))
            d = self._dev
            self.observation_to_device(d["rrt_grid"].data_ptr(), d["has_node"].data_ptr(), d["num_nodes"].data_ptr())
        return self._dev

    def policy_random_device(self, seed=0):
        """stand-in agent as a launch of its own: a random occupied bucket per live environment (-1 for finished ones),
        chosen on the device from the planner's list of occupied buckets (enqueue only)"""
        self._enter("device")
        d = self.device_buffers()
        self._ctx._chk(self._L.auvp_prrt_policy_random_dev(self._ctx.h, C.c_void_p(d["has_node"].data_ptr()), int(seed),
                                                           C.c_void_p(d["bucket"].data_ptr())))
        return d["bucket"]

    def step_device(self, bucket_dev=None, observe=True, agent_seed=None):
        """RRTEnv.step for all environments with the agent's choices already in device memory (`bucket_dev`: int32 [E] torch
        tensor on this GPU, default: the buffer policy_random_device fills; -1 skips an environment: reward 0, done flag
        unchanged).  `agent_seed` (int): the stand-in agent picks inside the planner launch instead and `bucket` receives its
        choices.  `observe`: True = the observation arrays are rewritten whole by a second launch; "delta" = the planner
        launch updates the one bucket per environment that changed in the arrays of device_buffers() (which hold the previous
        observation): one launch per step; False = no observation.  Enqueues on the planner's stream and returns the device
        tensors; call sync() before reading them from another stream or the host.  The planner's stream is not torch's: an
        agent that wrote `bucket_dev` on torch's current stream synchronises that stream
        (torch.cuda.current_stream().synchronize()) before this call."""
        self._enter("device")
        d = self.device_buffers()
        flags = (1 if agent_seed is not None else 0) | (2 if observe == "delta" else 0)
        b = d["bucket"] if (bucket_dev is None or agent_seed is not None) else bucket_dev
        self._ctx._chk(self._L.auvp_prrt_env_step_ex_dev(
            self._ctx.h, flags, int(agent_seed or 0), C.c_void_p(b.data_ptr()),
            C.c_void_p(d["rrt_grid"].data_ptr()) if observe else None, C.c_void_p(d["has_node"].data_ptr()),
            C.c_void_p(d["num_nodes"].data_ptr()), C.c_void_p(d["reward"].data_ptr()), C.c_void_p(d["done"].data_ptr())))
        return d

    def sync(self):
        """wait for the planner's stream; raises if an episode failed on the device since reset() (the host loop raises the
        same failures from step())"""
        self._ctx._chk(self._L.auvp_stream_sync(self._ctx.h))
        if self._mode == "device":
            st, env = C.c_int32(0), C.c_int32(0)
            self._ctx._chk(self._L.auvp_prrt_env_check(self._ctx.h, C.byref(st), C.byref(env)))
            if st.value < 0:
                raise _lib.AuvpError(int(st.value), "planner episode %d failed on the device" % env.value)

    def timed(self, enqueue):
        """device time in ms (HIP events on the planner's stream) of what `enqueue()` puts on it; waits for it"""
        L = self._L
        L.auvp_stream_mark.argtypes = [C.c_void_p, C.c_int32]
        L.auvp_stream_elapsed_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        self._ctx._chk(L.auvp_stream_mark(self._ctx.h, 0))
        enqueue()
        self._ctx._chk(L.auvp_stream_mark(self._ctx.h, 1))
        ms = C.c_double(0.0)
        self._ctx._chk(L.auvp_stream_elapsed_ms(self._ctx.h, C.byref(ms)))
        return ms.value

    def capture_step(self, enqueue):
        """record what `enqueue()` puts on the planner's stream (policy_random_device, step_device, ...) as a hipGraph of one
        step; returns its id for replay(id, n).  One un-captured step must have run before (first-use allocations).  The
        graph belongs to this episode's batch and device buffers: reset() drops it."""
        self._enter("device")
        gid = C.c_int32(-1)
        self._ctx._chk(self._L.auvp_graph_begin(self._ctx.h))
        try:
            enqueue()
        finally:
            self._ctx._chk(self._L.auvp_graph_end(self._ctx.h, C.byref(gid)))
        self._graphs.add(int(gid.value))
        return int(gid.value)

    def replay(self, graph_id, n_times=1):
        """n_times replays of a captured step, back to back on the planner's stream (enqueue only)"""
        if int(graph_id) not in self._graphs:
            raise _lib.AuvpError(-4, "graph %d was not captured since the last reset()" % int(graph_id))
        self._enter("device")
        self._ctx._chk(self._L.auvp_graph_launch(self._ctx.h, int(graph_id), int(n_times)))

    def tree(self, e):
        s = self._pb.summaries()[e]
        return self._pb.tree(e, s)


# ----------------------------------------------------------------------------------------------------------------------
# the reference-signature environment: what solveRL-RRT.py builds (:651-669) and steps (:711)
# ----------------------------------------------------------------------------------------------------------------------
END_GAME_RADIUS = 3.0
FOLLOWING_RADIUS = 50.0
OBSTACLE_ZONE = 0.0
WALL_ZONE = 10.0
ENV_SIZE = 500.0


class RRTEnv:
    """gym_rrt/envs/rrt_env.py::RRTEnv (:86-449) with its own call sequence -- RRTEnv(), init_env(...), reset(),
    step(chosen_grid_cell_idx, step_num) -- ONE environment on Python's global `random` stream, so a caller of the reference
    switches by changing the import.  `self.rrt_planner` is the drop-in Planner_RRT (its tree lives on the MI355X; each step is
    one generate_one_node launch that continues the global stream on the device and advances it by what the step consumed);
    the three observation arrays the reference rebuilds from Python lists after every node (:227-231,250-295) come from the
    device's observation kernel instead.  state["path"] is what the reference stores (:233-234): the new node (a
    Motion_plan_state with .parent / .path / rl_state_id), the final path as a list of Motion_plan_state whose elements carry
    the rl_state_id of the step that created them, or the previous value when the step added nothing.

    Not here: gym's spaces (gym is not a dependency: `action_space` / `observation_space` are None), rendering and the
    matplotlib live graph (visualisation is out of scope).  For many environments at once: RRTEnvBatch above."""
    metadata = {"render.modes": ["human"]}

    def __init__(self, device=0, max_nodes=4096):
        self.action_space = None
        self.observation_space = None
        self.auv_init_pos = None
        self.shark_init_pos = None
        self.state = None
        self.obstacle_array = []
        self.obstacle_array_for_rendering = []
        self.habitats_array = []
        self.habitats_array_for_rendering = []
        self.visited_unique_habitat_count = 0
        self.rrt_planner = None
        self._device = device
        self._max_nodes = int(max_nodes)
        self._ctx = None

    def init_env(self, auv_init_pos, shark_init_pos, boundary_array, grid_cell_side_length, num_of_subsections,
                 obstacle_array=[], habitat_grid=None):
        """rrt_env.py:132-180"""
        self.auv_init_pos = auv_init_pos
        self.shark_init_pos = shark_init_pos
        self.obstacle_array_for_rendering = obstacle_array
        self.habitats_array_for_rendering = []
        if habitat_grid is not None:
            self.habitat_grid = habitat_grid
            self.habitats_array_for_rendering = habitat_grid.habitat_array
        self.obstacle_array = np.array([[obs.x, obs.y, obs.z, obs.size] for obs in obstacle_array])
        self.boundary_array = boundary_array
        self.cell_side_length = grid_cell_side_length
        self.num_of_subsections = num_of_subsections
        return self.reset()

    def reset(self):
        """rrt_env.py:410-449: a new planner (the start node in its bucket) and the initial observation"""
        from .planner_rrt import Planner_RRT
        self.visited_unique_habitat_count = 0
        self.total_time_in_hab = 0
        a, s = self.auv_init_pos, self.shark_init_pos
        if self._ctx is None:
            self._ctx = _lib.Context(self._device)  # one device context for every planner this environment builds
        self.rrt_planner = Planner_RRT(a, s, self.boundary_array, self.obstacle_array_for_rendering, self.habitats_array_for_rendering,
                                       cell_side_length=self.cell_side_length, freq=RRT_PLANNER_FREQ,
                                       subsections_in_cell=self.num_of_subsections, max_nodes=self._max_nodes, context=self._ctx)
        grid, has, num = self._observation()
        self.state = {
            "auv_pos": np.array([a.x, a.y, a.z, a.theta]),
            "shark_pos": np.array([s.x, s.y, s.z, s.theta]),
            "obstacles_pos": self.obstacle_array,
            "rrt_grid": grid,
            "has_node": has,
            "path": None,
            "rrt_grid_num_of_nodes_only": num,
        }
        return self.state

    def _observation(self):
        pb = self.rrt_planner._pb
        nb = pb.rows * pb.cols * pb.subs
        grid = np.zeros((nb, 4))
        has = np.zeros(nb, dtype=np.int64)
        num = np.zeros(nb, dtype=np.int64)
        L = _bind()
        L.auvp_prrt_observation.argtypes = [C.c_void_p, C.c_int32, _lib._dp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        pb.ctx._chk(L.auvp_prrt_observation(pb.ctx.h, 0, _lib._p(grid), has.ctypes.data_as(C.POINTER(C.c_int64)),
                                            num.ctypes.data_as(C.POINTER(C.c_int64))))
        return grid, has, num

    def step(self, chosen_grid_cell_idx, step_num):
        """rrt_env.py:182-247 -> (state, reward, done, {})"""
        grid_cell_index = chosen_grid_cell_idx // self.num_of_subsections
        subsection_index = chosen_grid_cell_idx % self.num_of_subsections
        ncols = len(self.rrt_planner.env_grid[0])
        row, col = grid_cell_index // ncols, grid_cell_index % ncols
        chosen_grid_cell = self.rrt_planner.env_grid[row][col].subsection_cells[subsection_index]
        done, path = self.rrt_planner.generate_one_node(chosen_grid_cell, step_num)
        self.state["rrt_grid"], self.state["has_node"], self.state["rrt_grid_num_of_nodes_only"] = self._observation()
        if path is not None:
            self.state["path"] = path
        if done and path is not None:
            reward = R_FOUND_PATH
        elif path is not None:
            reward = R_CREATE_NODE
        else:
            reward = R_INVALID_NODE
        return self.state, reward, done, {}

    def close(self):
        self.rrt_planner = None
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None

    # ---- the reference's list builders, kept for callers that hand them a grid (:250-295) ----
    def convert_rrt_grid_to_1D(self, rrt_grid):
        return np.array([[gc.x, gc.y, sub.theta, len(sub.node_array)] for row in rrt_grid for gc in row for sub in gc.subsection_cells])

    def convert_rrt_grid_to_1D_num_of_nodes_only(self, rrt_grid):
        return np.array([len(sub.node_array) for row in rrt_grid for gc in row for sub in gc.subsection_cells])

    def generate_rrt_grid_has_node_array(self, rrt_grid):
        return np.array([0 if len(sub.node_array) == 0 else 1 for row in rrt_grid for gc in row for sub in gc.subsection_cells])

    # ---- geometry helpers of the reference class (:298-379) ----
    def calculate_range(self, a_pos, b_pos):
        delta_x = b_pos[0] - a_pos[0]
        delta_y = b_pos[1] - a_pos[1]
        return np.sqrt(delta_x ** 2 + delta_y ** 2)

    def within_follow_range(self, auv_pos, shark_pos):
        return bool(self.calculate_range(auv_pos, shark_pos) <= FOLLOWING_RADIUS)

    def check_collision(self, auv_pos):
        for obs in self.obstacle_array:
            if self.calculate_range(auv_pos, obs) <= obs[3]:
                return True
        return False

    def check_close_to_obstacles(self, auv_pos):
        for obs in self.obstacle_array:
            if self.calculate_range(auv_pos, obs) <= (obs[3] + OBSTACLE_ZONE):
                return True
        return False

    def check_close_to_walls(self, auv_pos, dist_from_walls_array):
        for dist_from_wall in dist_from_walls_array:
            if dist_from_wall <= WALL_ZONE:
                return True
        return False
