"""CPU checker (oracle/) pinned against the golden vectors captured from the reference
(tests/golden/make_golden.py).  No GPU needed."""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_world, libm_matches_golden


def test_rng_known_answers(orc):
    k = json.load(open(os.path.join(GOLDEN, "g7_random_kat.json")))
    for seed, v in k.items():
        if seed == "libm":
            continue
        rnd, bits, _ = orc.rng_kat(int(seed), n=5, nchoice=0)
        assert list(rnd) == v["random"]
        # the KAT drew 5 random() then 4 getrandbits(32)
        assert [int(b) for b in bits[:4]] == v["getrandbits32"]
    # survey 9.6
    rnd, _, _ = orc.rng_kat(7, n=3, nchoice=0)
    assert list(rnd) == [0.32383276483316237, 0.15084917392450192, 0.6509344730398537]


def test_rng_choice_stream(orc):
    import random
    for seed in (0, 7, 99):
        for n in (1, 3, 100, 1000, 65536):
            r = random.Random(seed)
            [r.random() for _ in range(4)]
            [r.getrandbits(32) for _ in range(4)]
            want = [r.choice(range(n)) for _ in range(16)]
            _, _, ch = orc.rng_kat(seed, n=4, nchoice=16, choice_n=n)
            assert [int(c) for c in ch] == want


def test_collision_cases(orc):
    g = json.load(open(os.path.join(GOLDEN, "g5_collision.json")))
    for poly, cases in ((g["rect"], g["cases"]), (g["penta"], g["penta_cases"])):
        for c in cases:
            w = orc.WorldArrays(obstacles=c["obs"], polygon=poly)
            for kind in ("libm", "portable"):
                assert orc.check_collision(w, c["pts"], kind=kind) == c["free"]


def test_collision_order_dependence(orc):
    # [A,B] collides, [B,A] is free
    poly = [[0, 0], [200, 0], [200, 200], [0, 200]]
    a, b = (52.0, 50.0, 1.0), (120.0, 50.0, 5.0)
    assert not orc.check_collision(orc.WorldArrays(obstacles=[a, b], polygon=poly), [(50.0, 50.0)])
    assert orc.check_collision(orc.WorldArrays(obstacles=[b, a], polygon=poly), [(50.0, 50.0)])


def test_cost_cases(orc):
    from auv_sim_amd import synth
    g = json.load(open(os.path.join(GOLDEN, "g4_cost.json")))
    exact = libm_matches_golden()
    for c in g["cases"]:
        world = synth.make_world(seed=c["world_seed"], n_obstacles=4, n_habitats=c["n_habitats"],
                                 cell=c["cell"], n_bins=c["n_bins"])
        w = orc.WorldArrays(world["obstacles"], world["habitats"], world["polygon"], world["bins"],
                            world["cells"], world["prob"])
        for kind in ("libm", "portable"):
            out = orc.cost(w, c["bin_lo"], c["bin_hi"], c["pts"], c["total"], c["weights"], kind=kind)
            if exact and kind == "libm":
                assert list(out) == c["out"]
            else:
                np.testing.assert_allclose(out, c["out"], rtol=1e-12, atol=1e-15)


G3 = sorted(glob.glob(os.path.join(GOLDEN, "g3_*.npz")))


def _run(orc, g, kind):
    gw = golden_world(g)
    w = orc.WorldArrays(gw["obstacles"], gw["habitats"], gw["polygon"], gw["bins"], gw["cells"], gw["prob"])
    init = [g["start"][0], g["start"][1], 0, 0, 0, 0]
    return orc.rrt_explore(w, int(g["seed"]), int(g["n_iter"]), str(g["mode"]), init=init,
                           freq=int(g["freq"]), bin_interval=int(g["bin_interval"]), v=int(g["v"]),
                           max_traj_time=float(g["max_traj_time"]), weights=g["weights"],
                           dist_to_end=float(g["dist_to_end"]), diff_max=float(g["diff_max"]), kind=kind)


@pytest.mark.parametrize("path", G3, ids=[os.path.basename(p)[:-4] for p in G3])
@pytest.mark.parametrize("kind", ["libm", "portable"])
def test_exploring_matches_reference(orc, path, kind):
    g = np.load(path)
    r = _run(orc, g, kind)
    assert r["status"] == 0
    # decisions: bit-exact in both math builds
    assert r["n_nodes"] == len(g["nodes"])
    assert np.array_equal(r["parent"], g["parent"])
    assert np.array_equal(r["pt_cnt"][1:] + 1, g["npath"][1:])  # root has path []
    ran = r["it_parent"] >= 0  # iterations not skipped by `continue`
    assert int(ran.sum()) == int(g["iters_run"])
    assert np.array_equal(r["it_parent"][ran], g["it_parent"])
    assert np.array_equal(r["it_accepted"][ran], g["it_accepted"])
    assert np.array_equal(r["it_npath"][ran], g["it_npath"])
    assert r["rng_after"] == float(g["rng_after"])  # same number of draws consumed
    # golden leaf_iter counts collision-checked iterations only
    assert np.array_equal(np.cumsum(ran)[r["leaf_iter"]] - 1, g["leaf_iter"])
    assert np.array_equal(r["leaf_cost"][:, 4:], g["leaf_cost"][:, 4:])
    if "bin_sizes" in g.files:
        K = len(r["bin_sizes"])
        assert np.array_equal(r["bin_sizes"], g["bin_sizes"][:K])
    assert r["path"].shape == g["res_path"].shape
    if kind == "libm" and libm_matches_golden():
        # same glibc as the capture: every float identical to the reference
        assert np.array_equal(r["nodes"], g["nodes"])
        assert hashlib.sha256(np.ascontiguousarray(r["points"]).tobytes()).hexdigest() == str(g["points_sha"])
        assert np.array_equal(r["leaf_cost"], g["leaf_cost"])
        assert np.array_equal(r["best_cost"], g["res_cost"])
        assert np.array_equal(r["path"], g["res_path"])
        assert r["best_length"] == float(g["res_path_length"])
    else:
        np.testing.assert_allclose(r["nodes"], g["nodes"], rtol=1e-9, atol=1e-9)
        if "points" in g.files:
            np.testing.assert_allclose(r["points"], g["points"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(r["leaf_cost"], g["leaf_cost"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(r["best_cost"], g["res_cost"], rtol=0, atol=1e-6)  # north_star bar
        np.testing.assert_allclose(r["best_cost"], g["res_cost"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(r["path"], g["res_path"], rtol=1e-9, atol=1e-9)
