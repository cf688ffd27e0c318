"""Batched RRTEnv counterpart (auv_sim_amd.rrt_env.RRTEnvBatch): its observation arrays and rewards
follow the reference env's step() contract (gym_rrt/envs/rrt_env.py:182-295), checked against the
Planner_RRT goldens by replaying their bucket sequence."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_env_batch_replays_golden_buckets():
    from auv_sim_amd.motion_plan_state import Motion_plan_state as MPS
    from auv_sim_amd.rrt_env import RRTEnvBatch, R_CREATE_NODE, R_FOUND_PATH, R_INVALID_NODE
    g = np.load(os.path.join(GOLDEN, "g2_main_s4.npz"))
    obstacles = [MPS(o[0], o[1], size=o[2]) for o in g["obstacles"].tolist()]
    bnd = [MPS(float(g["rect"][0]), float(g["rect"][1])), MPS(float(g["rect"][2]), float(g["rect"][3]))]
    auv = MPS(float(g["start"][0]), float(g["start"][1]), z=-5.0)
    shark = MPS(float(g["goal"][0]), float(g["goal"][1]), z=-5.0)
    E = 3
    env = RRTEnvBatch(auv, shark, bnd, int(g["cell"]), int(g["subs"]), obstacles, seeds=[4, 4, 9], max_nodes=2100,
                      freq=int(g["freq"]))
    st = env.reset()
    nb = int(g["grid_rows"]) * int(g["grid_cols"]) * int(g["subs"])
    assert st["rrt_grid"].shape == (E, nb, 4) and st["has_node"].shape == (E, nb)
    assert st["has_node"].sum(axis=1).tolist() == [1, 1, 1]
    # cell coordinates and subsection angle of the flat index, as convert_rrt_grid_to_1D lists them
    cols, cs = int(g["grid_cols"]), float(g["cell"])
    b = int(g["occupied"][0])
    assert st["rrt_grid"][0, b, 0] == (b // int(g["subs"])) % cols * cs and st["rrt_grid"][0, b, 3] == 1
    # episodes 0 and 1 share seed 4 = the golden's seed; the golden drew its buckets with the same
    # stream, so replaying them does NOT reproduce the golden tree (the device stream here is not
    # advanced by the bucket draws) -- but the two episodes must agree with each other, and the
    # reward / observation bookkeeping must be consistent with the tree.
    total_reward = np.zeros(E)
    for i in range(120):
        occ = np.flatnonzero(env.state["has_node"][0])
        choice = int(occ[i % len(occ)])
        empty = int(np.flatnonzero(env.state["has_node"][2] == 0)[0])
        before = env.state["rrt_grid_num_of_nodes_only"].sum(axis=1)
        st, reward, done, _ = env.step([choice, choice, empty], step_num=i)
        after = st["rrt_grid_num_of_nodes_only"].sum(axis=1)
        assert reward[2] == R_INVALID_NODE and after[2] == before[2]  # empty cell: (False, None)
        for e in (0, 1):
            if reward[e] == R_CREATE_NODE:
                assert after[e] == before[e] + 1
            elif reward[e] == R_INVALID_NODE:
                assert after[e] == before[e]
            else:
                assert reward[e] == R_FOUND_PATH and done[e]
        assert reward[0] == reward[1] and np.array_equal(st["rrt_grid"][0], st["rrt_grid"][1])
        assert np.array_equal(st["has_node"], (st["rrt_grid_num_of_nodes_only"] > 0).astype(np.int64))
        assert np.array_equal(st["rrt_grid"][:, :, 3], st["rrt_grid_num_of_nodes_only"].astype(np.float64))
        total_reward += reward
        if done[:2].all():
            break
    t0, t1 = env.tree(0), env.tree(1)
    assert np.array_equal(t0["nodes"], t1["nodes"]) and len(t0["nodes"]) == st["rrt_grid_num_of_nodes_only"][0].sum()


@pytest.mark.parametrize("name", ["g8_env_s1", "g8_env_s8"])
def test_env_matches_reference_rrtenv(name):
    """golden = the reference's own RRTEnv.init_env/step run (tests/golden/make_golden.py g8): same
    flat cell indices in, same rewards / node-count observations / RNG position out"""
    import random
    from auv_sim_amd.motion_plan_state import Motion_plan_state as MPS
    from auv_sim_amd.rrt_env import RRTEnvBatch
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    obstacles = [MPS(o[0], o[1], size=o[2]) for o in g["obstacles"].tolist()]
    bnd = [MPS(0.0, 0.0), MPS(50.0, 50.0)]
    auv, shark = MPS(10.0, 10.0, z=-5.0), MPS(35.0, 45.0, z=-5.0)
    random.seed(int(g["seed"]))
    env = RRTEnvBatch(auv, shark, bnd, int(g["cell"]), int(g["subs"]), obstacles, seeds=None, max_nodes=1200, freq=int(g["freq"]))
    st = env.reset()
    assert np.array_equal(st["rrt_grid"][0], g["rrt_grid0"])  # cell.x, cell.y, subsection.theta, count
    for i, idx in enumerate(g["choices"].tolist()):
        st, reward, done, _ = env.step([idx], step_num=i)
        assert reward[0] == g["rewards"][i], i
        assert bool(done[0]) == bool(g["dones"][i])
        assert np.array_equal(st["rrt_grid_num_of_nodes_only"][0], g["counts"][i]), i
    assert np.array_equal(st["rrt_grid"][0], g["final_rrt_grid"]) and np.array_equal(st["has_node"][0], g["final_has_node"])
    assert random.random() == float(g["rng_after"])


def test_env_device_resident_loop_matches_host_loop():
    """auvp_prrt_env_step_dev (bucket ids, observations, rewards and done flags all in HBM) against RRTEnvBatch.step fed the
    same choices: identical rewards, done flags, observation arrays and trees; and the stand-in device agent only ever
    picks occupied buckets (-1 for finished environments)"""
    import torch
    from auv_sim_amd import synth
    from auv_sim_amd.motion_plan_state import Motion_plan_state as MPS
    from auv_sim_amd.rrt_env import RRTEnvBatch
    w = synth.make_rect_world(seed=3, n_obstacles=64)
    obstacles = [MPS(o[0], o[1], size=o[2]) for o in w["obstacles"].tolist()]
    bnd = [MPS(float(w["rect"][0]), float(w["rect"][1])), MPS(float(w["rect"][2]), float(w["rect"][3]))]
    auv, shark = MPS(float(w["start"][0]), float(w["start"][1]), z=-5.0), MPS(float(w["goal"][0]), float(w["goal"][1]), z=-5.0)
    E, n_steps = 24, 150
    seeds = list(range(40, 40 + E))
    host = RRTEnvBatch(auv, shark, bnd, 5, 2, obstacles, seeds=seeds, max_nodes=n_steps + 8, freq=10)
    devenv = RRTEnvBatch(auv, shark, bnd, 5, 2, obstacles, seeds=seeds, max_nodes=n_steps + 8, freq=10)
    st = host.reset()
    devenv.reset()
    d = devenv.device_buffers()
    assert np.array_equal(d["rrt_grid"].cpu().numpy(), st["rrt_grid"])
    rng = np.random.default_rng(1)
    dev = d["bucket"].device
    for i in range(n_steps):
        has = st["has_node"] != 0
        choice = np.array([rng.choice(np.flatnonzero(h)) if (h.any() and rng.random() < 0.9) else int(rng.integers(0, has.shape[1]))
                           for h in has], dtype=np.int32)
        st, reward, done, _ = host.step(choice)
        dd = devenv.step_device(torch.from_numpy(choice).to(dev))
        devenv.sync()
        assert np.array_equal(dd["reward"].cpu().numpy(), reward), i
        assert np.array_equal(dd["done"].cpu().numpy().astype(bool), done), i
        assert np.array_equal(dd["rrt_grid"].cpu().numpy(), st["rrt_grid"])
        assert np.array_equal(dd["has_node"].cpu().numpy(), st["has_node"])
        assert np.array_equal(dd["num_nodes"].cpu().numpy(), st["rrt_grid_num_of_nodes_only"])
        if done.all():
            break
    assert done.any() and (reward == 300).any() or i == n_steps - 1
    for e in (0, E - 1):
        a, b = host.tree(e), devenv.tree(e)
        assert np.array_equal(a["nodes"], b["nodes"]) and np.array_equal(a["parent"], b["parent"])
    # the stand-in agent: an occupied bucket for every live environment, -1 for the finished ones; several calls differ
    picks = []
    for k in range(4):
        b = devenv.policy_random_device(seed=7 + k // 2)   # a pure function of (seed, environment, step count): 0 == 1, 2 == 3
        devenv.sync()
        b = b.cpu().numpy()
        hn = d["has_node"].cpu().numpy()
        dn = d["done"].cpu().numpy().astype(bool)
        assert (b[dn] == -1).all()
        live = np.flatnonzero(~dn)
        assert all(hn[e, b[e]] == 1 for e in live)
        picks.append(b.copy())
    assert np.array_equal(picks[0], picks[1]) and np.array_equal(picks[2], picks[3])   # no step in between: the same draw


def test_stand_in_agent_is_a_pure_function_of_seed_environment_and_step_count():
    """auvp_prrt_policy_random_dev (include/auvplan.h): calls repeated without a step return the same picks, another seed
    redraws, a step redraws; has_node_dev is not read and may be NULL"""
    import ctypes as C
    env = _small_env(E=32, max_nodes=64)
    env.reset()
    d = env.device_buffers()
    for _ in range(12):                                   # grow the trees: several occupied buckets per environment
        env.policy_random_device(seed=5)
        env.step_device()
    env.sync()
    hn = d["has_node"].cpu().numpy()
    dn = d["done"].cpu().numpy().astype(bool)
    many = (~dn) & (hn.sum(axis=1) > 3)
    assert many.sum() >= 8, "the fixture must leave live environments with several occupied buckets"

    def pick(seed, null_has_node=False):
        if null_has_node:
            env._ctx._chk(env._L.auvp_prrt_policy_random_dev(env._ctx.h, C.c_void_p(0), int(seed), C.c_void_p(d["bucket"].data_ptr())))
            b = d["bucket"]
        else:
            b = env.policy_random_device(seed=seed)
        env.sync()
        return b.cpu().numpy().copy()
    a, a2, a3, b = pick(7), pick(7), pick(7, null_has_node=True), pick(8)
    assert np.array_equal(a, a2) and np.array_equal(a, a3)
    assert not np.array_equal(a[many], b[many])
    live = np.flatnonzero(~dn)
    assert all(hn[e, a[e]] == 1 for e in live) and (a[dn] == -1).all()
    env.step_device()                                     # one step: the step count moved, the same seed draws anew
    env.sync()
    dn2 = d["done"].cpu().numpy().astype(bool)
    c = pick(8)
    still = many & ~dn2
    assert still.sum() >= 4 and not np.array_equal(b[still], c[still])


@pytest.mark.parametrize("rows", [False, True])
def test_env_device_loop_as_a_graph_equals_eager(rows, monkeypatch):
    """one step of the device-resident loop (stand-in agent + generate_one_node + observation + outcome) captured as a
    hipGraph and replayed n times == the same n steps enqueued one by one: the loop's step counter lives in HBM, so every
    replay draws anew.  rows: the four-episodes-per-wavefront kernel (AUVP_PRRT_ROWS=1; the default of large batches), whose
    work counter must restart at every replay (a memset node) -- and eager steps AFTER the replays must still be right"""
    if rows:
        monkeypatch.setenv("AUVP_PRRT_ROWS", "1")
    from auv_sim_amd import synth
    from auv_sim_amd.motion_plan_state import Motion_plan_state as MPS
    from auv_sim_amd.rrt_env import RRTEnvBatch
    w = synth.make_rect_world(seed=3, n_obstacles=64)
    obstacles = [MPS(o[0], o[1], size=o[2]) for o in w["obstacles"].tolist()]
    bnd = [MPS(float(w["rect"][0]), float(w["rect"][1])), MPS(float(w["rect"][2]), float(w["rect"][3]))]
    auv, shark = MPS(float(w["start"][0]), float(w["start"][1]), z=-5.0), MPS(float(w["goal"][0]), float(w["goal"][1]), z=-5.0)
    E, n_steps = 40, 90
    out = []
    for graph in (False, True):
        env = RRTEnvBatch(auv, shark, bnd, 5, 1, obstacles, seeds=list(range(E)), max_nodes=n_steps + 8, freq=10)
        env.reset()
        d = env.device_buffers()

        def one_step():
            env.policy_random_device(seed=3)
            env.step_device()
        one_step()  # first-use allocations; also step 0 of both runs
        if rows:
            assert env._ctx.prrt_last_kernel() == "prrt_rows_kernel"
        if graph:
            gid = env.capture_step(one_step)
            env.replay(gid, n_steps - 1 - 6)
            for _ in range(3):          # eager steps between and after replays of the same graph
                one_step()
            env.replay(gid, 2)
            one_step()
        else:
            for _ in range(n_steps - 1):
                one_step()
        env.sync()
        out.append((d["num_nodes"].cpu().numpy().copy(), d["reward"].cpu().numpy().copy(), d["done"].cpu().numpy().copy(),
                    [env.tree(e)["nodes"] for e in (0, E - 1)]))
    a, b = out
    assert a[0].sum() > E * 5  # trees grew
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert all(np.array_equal(x, y) for x, y in zip(a[3], b[3]))


def _small_env(E=16, max_nodes=100, seeds=None):
    from auv_sim_amd import synth
    from auv_sim_amd.motion_plan_state import Motion_plan_state as MPS
    from auv_sim_amd.rrt_env import RRTEnvBatch
    w = synth.make_rect_world(seed=3, n_obstacles=64)
    obstacles = [MPS(o[0], o[1], size=o[2]) for o in w["obstacles"].tolist()]
    bnd = [MPS(float(w["rect"][0]), float(w["rect"][1])), MPS(float(w["rect"][2]), float(w["rect"][3]))]
    auv, shark = MPS(float(w["start"][0]), float(w["start"][1]), z=-5.0), MPS(float(w["goal"][0]), float(w["goal"][1]), z=-5.0)
    return RRTEnvBatch(auv, shark, bnd, 5, 1, obstacles, seeds=seeds or list(range(E)), max_nodes=max_nodes, freq=10)


def test_one_launch_step_equals_the_three_launch_step():
    """The device-resident step in its forms -- agent as its own launch + planner step + observation rewrite (three launches);
    agent inside the planner launch (two); agent inside and the observation arrays updated in place by the planner launch
    (one: a step changes one bucket per environment); the one-launch step captured as a hipGraph, one and eight steps per
    graph -- gives the same picks, rewards, flags, observation arrays and trees"""
    n_steps = 73
    out = []
    for fused, observe, graph in ((False, True, 0), (True, True, 0), (True, "delta", 0), (True, "delta", 1), (True, "delta", 8)):
        env = _small_env(E=24, max_nodes=n_steps + 8)
        env.reset()
        d = env.device_buffers()

        def one_step():
            if fused:
                env.step_device(agent_seed=11, observe=observe)
            else:
                env.policy_random_device(seed=11)
                env.step_device(observe=observe)
        one_step()
        env.sync()
        picks = [d["bucket"].cpu().numpy().copy()]
        if graph:
            gid = env.capture_step(lambda: [one_step() for _ in range(graph)])
            assert (n_steps - 1) % graph == 0
            env.replay(gid, (n_steps - 1) // graph)
        else:
            for _ in range(n_steps - 1):
                one_step()
                env.sync()
                picks.append(d["bucket"].cpu().numpy().copy())
        env.sync()
        out.append((d["num_nodes"].cpu().numpy().copy(), d["reward"].cpu().numpy().copy(), d["done"].cpu().numpy().copy(),
                    d["rrt_grid"].cpu().numpy().copy(), d["has_node"].cpu().numpy().copy(), [env.tree(e)["nodes"] for e in (0, 23)], picks))
    a = out[0]
    assert a[0].sum() > 24 * 5
    assert np.array_equal(a[4], (a[0] > 0).astype(np.int64)) and np.array_equal(a[3][:, :, 3], a[0].astype(np.float64))
    for x in out[1:]:
        for k in range(5):
            assert np.array_equal(a[k], x[k]), k
        assert all(np.array_equal(p, q) for p, q in zip(a[5], x[5]))
    for x in out[1:3]:
        assert len(x[6]) == n_steps and all(np.array_equal(p, q) for p, q in zip(a[6], x[6]))   # the agent's picks, step by step


def test_skipped_and_failed_environments_in_the_device_loop():
    """-1 skips a live environment: reward 0, done flag unchanged, tree untouched; a bucket id outside the grid fails the
    episode on the device: flagged done, rewarded 0, and sync() raises like the host loop's step() does"""
    import torch
    from auv_sim_amd import _lib
    env = _small_env(E=8, max_nodes=64)
    st = env.reset()
    d = env.device_buffers()
    dev = d["bucket"].device
    occ0 = np.array([int(np.flatnonzero(h)[0]) for h in st["has_node"]], dtype=np.int32)
    for i in range(6):
        choice = occ0.copy()
        choice[3] = -1                     # the agent skips environment 3 in every step
        env.step_device(torch.from_numpy(choice).to(dev))
        env.sync()
        assert d["reward"][3].item() == 0 and d["done"][3].item() == 0
    assert int(d["num_nodes"][3].sum().item()) == 1     # only the start node: never stepped
    assert int(d["num_nodes"][0].sum().item()) >= 1
    bad = occ0.copy()
    bad[5] = env.n_buckets + 7             # outside the grid
    env.step_device(torch.from_numpy(bad).to(dev))
    with pytest.raises(_lib.AuvpError):
        env.sync()
    assert d["done"][5].item() == 1 and d["reward"][5].item() == 0
    # the host loop surfaces the same failure from step()
    host = _small_env(E=8, max_nodes=64)
    host.reset()
    with pytest.raises(_lib.AuvpError):
        host.step(bad)


def test_graphs_die_with_their_batch_and_modes_do_not_mix():
    """a captured step replays launches on the batch's buffers: reset() (a new batch) invalidates it, on both sides of the
    C-ABI; and one episode runs either the host loop or the device-resident loop"""
    import ctypes as C
    from auv_sim_amd import _lib
    env = _small_env(E=8, max_nodes=64)
    env.reset()

    def one_step():
        env.step_device(agent_seed=3, observe="delta")
    one_step()
    gid = env.capture_step(one_step)
    env.replay(gid, 3)
    env.sync()
    with pytest.raises(RuntimeError):
        env.step(np.zeros(8, dtype=np.int64))           # host stepping inside a device-resident episode
    env.reset()
    with pytest.raises(_lib.AuvpError):
        env.replay(gid, 1)                               # the wrapper forgot the id ...
    rc = env._L.auvp_graph_launch(env._ctx.h, C.c_int32(gid), C.c_int32(1))
    assert rc == -4                                      # ... and the library destroyed the graph (AUVP_ERR_STATE)
    st, reward, done, _ = env.step(np.array([int(np.flatnonzero(h)[0]) for h in env.state["has_node"]]))
    with pytest.raises(RuntimeError):
        env.step_device(agent_seed=3, observe="delta")   # device stepping inside a host episode
    env.reset()
    one_step()                                           # a fresh episode may choose again
    gid2 = env.capture_step(one_step)
    assert gid2 != gid
    env.replay(gid2, 2)
    env.sync()


G8_ALL = ["g8_env_s1", "g8_env_s8", "g8_env_plan_s1", "g8_env_org_m50_m30", "g8_env_org_p30_p20"]


def _g8_world(g):
    from auv_sim_amd.motion_plan_state import Motion_plan_state as MPS
    obstacles = [MPS(o[0], o[1], size=o[2]) for o in g["obstacles"].tolist()]
    r = g["rect"]
    bnd = [MPS(float(r[0]), float(r[1])), MPS(float(r[2]), float(r[3]))]
    auv = MPS(float(g["start"][0]), float(g["start"][1]), z=-5.0, theta=0.0)
    shark = MPS(float(g["goal"][0]), float(g["goal"][1]), z=-5.0, theta=0.0)
    return auv, shark, bnd, obstacles


@pytest.mark.parametrize("name", G8_ALL)
def test_reference_signature_rrtenv_replays_the_reference_run(name):
    """auv_sim_amd.rrt_env.RRTEnv -- the reference's own call sequence RRTEnv() / init_env(...) / step(idx, step_num)
    (gym_rrt/envs/rrt_env.py:132,182,410; solveRL-RRT.py:651-669,711) -- against the reference's run (G8): observations,
    rewards, done flags, state["path"] as the reference stores it (the new node object; the final path as a list whose elements
    carry the rl_state_id of the step that created them; unchanged when a step adds nothing), the global stream's position.
    Two fixtures are translated worlds (negative / past-the-grid bucket indexes under the environment's flat index)."""
    import random
    from auv_sim_amd.motion_plan_state import Motion_plan_state as MPS
    from auv_sim_amd.rrt_env import RRTEnv
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    auv, shark, bnd, obstacles = _g8_world(g)
    random.seed(int(g["seed"]))
    env = RRTEnv()
    st = env.init_env(auv, shark, boundary_array=bnd, grid_cell_side_length=int(g["cell"]), obstacle_array=obstacles,
                      num_of_subsections=int(g["subs"]))
    assert set(st) == {"auv_pos", "shark_pos", "obstacles_pos", "rrt_grid", "has_node", "path", "rrt_grid_num_of_nodes_only"}
    assert st["path"] is None and np.array_equal(st["rrt_grid"], g["rrt_grid0"])
    assert np.array_equal(st["auv_pos"], [auv.x, auv.y, -5.0, 0.0]) and np.array_equal(st["shark_pos"], [shark.x, shark.y, -5.0, 0.0])
    planning_policy = str(g["policy"]) == "planning"
    prev_path = None
    nrows, ncols, subs = len(env.rrt_planner.env_grid), len(env.rrt_planner.env_grid[0]), int(g["subs"])
    for i, idx in enumerate(g["choices"].tolist()):
        if planning_policy:  # the agent of this fixture draws its choice from the global stream, as planning() does (:186)
            r_, c_, k_ = random.choice(env.rrt_planner.occupied_grid_cells_array)
            assert ((r_ % nrows) * ncols + c_ % ncols) * subs + k_ % subs == idx, i
        st, reward, done, info = env.step(idx, i)
        assert reward == g["rewards"][i] and bool(done) == bool(g["dones"][i]) and info == {}, i
        assert np.array_equal(st["rrt_grid_num_of_nodes_only"], g["counts"][i]), i
        kind = int(g["path_kind"][i])
        if kind == 0:
            assert st["path"] is prev_path
        elif kind == 1:
            node = st["path"]
            assert isinstance(node, MPS) and node is env.rrt_planner.mps_list[-1]
            want = g["node_rec"][i]
            np.testing.assert_allclose([node.x, node.y, node.theta, node.traj_time_stamp], want[:4], rtol=1e-9, atol=1e-9)
            assert node.rl_state_id == int(want[4]) == i
            assert node.parent is node.path[0] and node.parent in env.rrt_planner.mps_list
        else:
            assert isinstance(st["path"], list)
        prev_path = st["path"]
    assert np.array_equal(st["rrt_grid"], g["final_rrt_grid"]) and np.array_equal(st["has_node"], g["final_has_node"])
    if bool(g["done"]):
        path = st["path"]
        assert len(path) == len(g["path"])
        got = np.array([[p.x, p.y, p.theta, p.traj_time_stamp] for p in path])
        np.testing.assert_allclose(got, g["path"], rtol=1e-9, atol=1e-9)
        ids = [-1 if p.rl_state_id is None else p.rl_state_id for p in path]
        assert ids == g["path_state_id"].tolist()
        np.testing.assert_allclose(path[0].length, float(g["path0_length"]), rtol=1e-9)
        np.testing.assert_allclose(env.rrt_planner.cal_length(path), float(g["cal_length"]), rtol=1e-9)
        # the tree's own objects make up the tail of the path (generate_final_course :317-327)
        assert path[-1] is env.rrt_planner.mps_list[0]
    assert random.random() == float(g["rng_after"])
    # reset(): a fresh planner on the same device context, the initial observation again
    random.seed(int(g["seed"]))
    st2 = env.reset()
    assert st2["path"] is None and np.array_equal(st2["rrt_grid"], g["rrt_grid0"])
    st2, reward, done, _ = env.step(int(g["choices"][0]) if not planning_policy else int(g["choices"][0]), 0)
    env.close()


@pytest.mark.parametrize("name", ["g8_env_org_m50_m30", "g8_env_org_p30_p20"])
def test_env_batch_on_a_translated_world(name):
    """RRTEnvBatch.step (global-stream mode) on the translated G8 worlds: same rewards / counts / stream position"""
    import random
    from auv_sim_amd.rrt_env import RRTEnvBatch
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    auv, shark, bnd, obstacles = _g8_world(g)
    random.seed(int(g["seed"]))
    env = RRTEnvBatch(auv, shark, bnd, int(g["cell"]), int(g["subs"]), obstacles, seeds=None, max_nodes=1600, freq=int(g["freq"]))
    st = env.reset()
    assert np.array_equal(st["rrt_grid"][0], g["rrt_grid0"])
    planning_policy = str(g["policy"]) == "planning"
    for i, idx in enumerate(g["choices"].tolist()):
        if planning_policy:
            occ = np.flatnonzero(st["has_node"][0])
            # the reference's occupied list is in creation order; the device's list is the same list
            occ_list = [int(b) for b in env._pb.grid(0)[0]]
            assert sorted(occ_list) == occ.tolist()
            assert random.choice(occ_list) == idx, i
        st, reward, done, _ = env.step([idx], step_num=i)
        assert reward[0] == g["rewards"][i] and bool(done[0]) == bool(g["dones"][i]), i
        assert np.array_equal(st["rrt_grid_num_of_nodes_only"][0], g["counts"][i]), i
    assert np.array_equal(st["rrt_grid"][0], g["final_rrt_grid"]) and np.array_equal(st["has_node"][0], g["final_has_node"])
    assert random.random() == float(g["rng_after"])
