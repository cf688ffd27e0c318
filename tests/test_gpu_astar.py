"""GPU parity tests for the four A* variants through the C-ABI: bit-exact against the goldens captured
from the reference (pop order, every g/h/f/cost/pathLen/time_stamp, path, cost list, smoothed path,
caller-visible habitat-list mutation) and against the CPU checker on a batch of random instances."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_oracle_astar_golden import astar_kwargs, check_result

pytestmark = pytest.mark.gpu
FILES = sorted(glob.glob(os.path.join(GOLDEN, "g1_*.npz")) + glob.glob(os.path.join(GOLDEN, "g6_*.npz")))


@pytest.fixture(scope="module")
def ctx():
    from auv_sim_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def _gpu_run(ctx, v, starts, kw, **extra):
    from auv_sim_amd import _astar_lib as al
    ctx.set_world(kw.get("obstacles"), kw.get("habitats"), kw.get("polygon"), kw.get("bins"), kw.get("cells"), kw.get("prob"))
    E = len(starts)
    goals = np.tile(np.asarray(kw["goal"], dtype=np.float64), (E, 1)) if "goal" in kw else None
    limits = np.full(E, kw["limit"]) if "limit" in kw else None
    return al.run_batch(ctx, v, starts, goals=goals, limits=limits, box=kw.get("box", (0, 0, 0, 0)),
                        velocity=kw.get("velocity", 1.0), weights=kw.get("weights", (0, 0, 0, 0)), **extra)


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[:-4] for p in FILES])
def test_astar_gpu_matches_reference(ctx, path):
    g = np.load(path)
    v, kw = astar_kwargs(g)
    r = _gpu_run(ctx, v, [g["start"]], kw, exp_log=True)[0]
    check_result(r, g)


@pytest.mark.parametrize("variant", ["astar", "astar_real", "astar_fixLen", "astar_fixLenSOG"])
def test_astar_batch_vs_oracle(ctx, orc, variant):
    """many instances (different starts) in one launch over one shared world, each equal to its checker run"""
    from auv_sim_amd import synth
    from oracle import orc_astar as oa
    rng = np.random.default_rng(17)
    if variant == "astar":
        w = synth.make_lattice_world(seed=11, n_obstacles=30, r_range=(10, 22))
        starts = np.array([(10.0 * rng.integers(0, 8), 10.0 * rng.integers(0, 8)) for _ in range(24)])
        kw = dict(obstacles=w["obstacles"], goal=(490.0, 490.0), box=w["box"])
    else:
        w = synth.make_world(seed=12, n_obstacles=64 if variant != "astar_fixLenSOG" else 32, obst_radius=(2.0, 6.0),
                             n_habitats=8, hab_radius=(10.0, 25.0))
        starts = np.array([(-290.0 + 10.0 * rng.integers(0, 6), -90.0 + 10.0 * rng.integers(0, 6)) for _ in range(24)])
        kw = dict(obstacles=w["obstacles"], polygon=w["polygon"])
        if variant == "astar_real":
            kw["goal"] = (-120.0, 80.0)
        else:
            kw.update(habitats=w["habitats"], limit=150.0, weights=(0, 10, 10, 100))
            if variant == "astar_fixLenSOG":
                kw.update(bins=w["bins"], cells=w["cells"], prob=w["prob"], velocity=1.0)
    res = _gpu_run(ctx, variant, starts, kw, exp_log=True)
    n_found = 0
    for e, r in enumerate(res):
        o = oa.run(variant, starts[e], kind="portable", cap_nodes=20000, **kw)
        assert r["status"] == o["status"] == 0, (e, r["status"], o["status"])
        assert r["found"] == o["found"] and r["n_nodes"] == o["n_nodes"] and r["n_children"] == o["n_children"]
        assert np.array_equal(r["expansions"], o["expansions"])
        assert np.array_equal(r["path"], o["path"]) and np.array_equal(r["cost_list"], o["cost_list"])
        assert np.array_equal(r["node_path"], o["node_path"])
        assert np.array_equal(r["smooth_path"], o["smooth_path"])
        assert np.array_equal(r["hab_left"], o["hab_left"]) and r["visited_count"] == o["visited_count"]
        n_found += r["found"]
    assert n_found > 0


@pytest.mark.parametrize("cells_as", ["product_grid", "shuffled_list", "column_major", "grid_14m_float_starts", "uneven_product_grid",
                                      "fine_grid_tables_in_hbm"])
@pytest.mark.parametrize("wavefronts", ["two_per_instance", "one_per_instance"])
def test_sog_cell_lookup_paths(ctx, orc, cells_as, wavefronts, monkeypatch):
    """get_cell_prob (astar_fixLenSOG.py:485-514) = the first cell of the LIST whose closed box holds the point; lattice
    points sit on cell edges, so up to four cells match and the list order decides.  A row-major product grid takes the
    arithmetic lookup (three rows and columns around the lower bounds), any other list the sweep over the cells."""
    from auv_sim_amd import synth
    from oracle import orc_astar as oa
    # (latency batches on a product grid give every instance a second wavefront for what depends on the popped node's position
    # alone, astar_kernel.h PAIR; both forms against the checker)
    monkeypatch.setenv("AUVP_ASTAR_PAIR", "1" if wavefronts == "two_per_instance" else "0")
    rng = np.random.default_rng(23)
    # (80 x 80 cells of 2.5 m: the edge tables no longer fit the kernel's LDS budget and are read from memory)
    cell = 14.0 if cells_as == "grid_14m_float_starts" else (2.5 if cells_as == "fine_grid_tables_in_hbm" else 10.0)
    w = synth.make_world(seed=13, n_obstacles=32, obst_radius=(2.0, 6.0), n_habitats=8, hab_radius=(10.0, 25.0), cell=cell)
    cells, prob = np.array(w["cells"], dtype=np.float64), np.array(w["prob"], dtype=np.float64)
    if cells_as == "uneven_product_grid":
        from conftest import uneven_grid
        w2 = uneven_grid(w, 77)
        cells, prob = w2["cells"], w2["prob"]
    if cells_as == "shuffled_list":
        perm = rng.permutation(len(cells))
        cells, prob = cells[perm], prob[:, perm]
    elif cells_as == "column_major":
        n = int(round(np.sqrt(len(cells))))
        perm = np.arange(len(cells)).reshape(-1, n).T.ravel() if len(cells) == n * n else rng.permutation(len(cells))
        cells, prob = cells[perm], prob[:, perm]
    if cells_as == "grid_14m_float_starts":
        starts = np.array([(-290.0 + 10.0 * rng.integers(0, 6) + 0.37, -90.0 + 10.0 * rng.integers(0, 6) + 0.21) for _ in range(12)])
    else:
        starts = np.array([(-290.0 + 10.0 * rng.integers(0, 6), -90.0 + 10.0 * rng.integers(0, 6)) for _ in range(12)])
    if cells_as == "fine_grid_tables_in_hbm":
        starts = starts[:5]
    kw = dict(obstacles=w["obstacles"], polygon=w["polygon"], habitats=w["habitats"], limit=150.0, weights=(0, 10, 10, 100),
              bins=w["bins"], cells=cells, prob=prob, velocity=1.0)
    res = _gpu_run(ctx, "astar_fixLenSOG", starts, kw, exp_log=True)
    for e, r in enumerate(res):
        o = oa.run("astar_fixLenSOG", starts[e], kind="portable", cap_nodes=20000, **kw)
        assert r["status"] == o["status"], (e, r["status"], o["status"])
        assert r["found"] == o["found"] and r["n_nodes"] == o["n_nodes"] and r["n_children"] == o["n_children"]
        assert np.array_equal(r["expansions"], o["expansions"])
        assert np.array_equal(r["path"], o["path"]) and np.array_equal(r["cost_list"], o["cost_list"])
        assert np.array_equal(r["node_path"], o["node_path"]) and np.array_equal(r["smooth_path"], o["smooth_path"])


def test_astar_capacity_is_an_error_not_a_truncation(ctx):
    from auv_sim_amd import synth
    w = synth.make_lattice_world(seed=0, n_obstacles=10)
    r = _gpu_run(ctx, "astar", [(0.0, 0.0)], dict(obstacles=w["obstacles"], goal=(490.0, 490.0), box=w["box"]), cap_nodes=64)[0]
    assert r["status"] == -2 and not r["found"]


@pytest.mark.parametrize("variant", ["astar_fixLen", "astar_fixLenSOG"])
def test_pop_by_the_scan_of_every_node(ctx, orc, variant, monkeypatch):
    """an open set that outgrows the 768-entry list the pop scans in LDS falls back to the scan of every node's f in memory,
    where a closed node's f is a nan (nodes created closed -- their cell was visited already -- and popped ones).  Cells close
    once visited, so the worlds of these tests never get there: AUVP_ASTAR_NO_LIST=1 takes the fallback from the first pop.
    Both forms of the fixLenSOG kernel (one / two wavefronts per instance)."""
    from auv_sim_amd import synth
    from oracle import orc_astar as oa
    monkeypatch.setenv("AUVP_ASTAR_NO_LIST", "1")
    w = synth.make_world(seed=12, n_obstacles=32, obst_radius=(2.0, 6.0), n_habitats=8, hab_radius=(10.0, 25.0))
    starts = np.array([(-280.0, -80.0), (-250.0, -60.0), (-270.0, -50.0)])
    kw = dict(obstacles=w["obstacles"], polygon=w["polygon"], habitats=w["habitats"], limit=300.0, weights=(0, 10, 10, 100))
    if variant == "astar_fixLenSOG":
        kw.update(bins=w["bins"], cells=w["cells"], prob=w["prob"], velocity=1.0)
    for pair in ("1", "0"):
        monkeypatch.setenv("AUVP_ASTAR_PAIR", pair)
        res = _gpu_run(ctx, variant, starts, kw, exp_log=True)
        for e, r in enumerate(res):
            o = oa.run(variant, starts[e], kind="portable", cap_nodes=20000, **kw)
            assert r["status"] == o["status"], (e, r["status"], o["status"])
            assert r["found"] == o["found"] and r["n_nodes"] == o["n_nodes"] and r["n_children"] == o["n_children"]
            assert np.array_equal(r["expansions"], o["expansions"])
            assert np.array_equal(r["path"], o["path"]) and np.array_equal(r["cost_list"], o["cost_list"])
        if variant == "astar_fixLen":
            break
