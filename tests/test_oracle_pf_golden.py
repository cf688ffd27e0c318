"""oracle/orc_pf.c (particleFilter.py restatement incl. numpy's legacy RandomState stream) against the
G10 goldens captured from the reference's own run (tests/golden/make_golden.py g10)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, libm_matches_golden

FILES = sorted(glob.glob(os.path.join(GOLDEN, "g10_pf_*.npz")))
IDS = [os.path.basename(f)[:-4] for f in FILES]


def test_goldens_present():
    assert len(FILES) == 3


def test_numpy_legacy_stream_known_answers(orc):
    """np.random.seed / uniform / choice of the numpy in this image (the stream the reference draws from)"""
    from oracle import orc_pf
    for seed, n in [(1, 1), (5, 1000), (4000000123, 4097), (77, 5000)]:
        np.random.seed(seed)
        u = np.array([np.random.uniform(-150, 150) for _ in range(64)])
        c = np.array([np.random.choice(n) for _ in range(64)])
        uo, co = orc_pf.np_kat(seed, 64, n)
        assert np.array_equal(u, uo) and np.array_equal(c, co)
        mt, pos = orc_pf.np_seed_state(seed)
        np.random.seed(seed)
        st = np.random.get_state()
        assert np.array_equal(st[1], mt) and st[2] == pos


@pytest.mark.parametrize("path", FILES, ids=IDS)
def test_libm_build_is_bit_exact(orc, path):
    if not libm_matches_golden():
        pytest.skip("this machine's libm differs from the one the goldens were captured with")
    from oracle import orc_pf
    g = np.load(path)
    mt, pos = orc_pf.np_seed_state(int(g["seed"]))
    r = orc_pf.run(1000, g["measurements"], g["shark_xy"], g["shark0"], mt, pos, kind="libm")
    assert r["status"] == 0
    for k in ("created", "updated", "resampled", "choice", "alias_first", "mean", "range_error"):
        assert np.array_equal(r[k], g[k]), k
    assert np.array_equal(r["mt"], g["mt_key"]) and r["mt_pos"] == int(g["mt_pos"])


@pytest.mark.parametrize("path", FILES, ids=IDS)
def test_portable_build_same_decisions(orc, path):
    from oracle import orc_pf
    g = np.load(path)
    mt, pos = orc_pf.np_seed_state(int(g["seed"]))
    r = orc_pf.run(1000, g["measurements"], g["shark_xy"], g["shark0"], mt, pos, kind="portable")
    assert r["status"] == 0
    for k in ("choice", "alias_first"):
        assert np.array_equal(r[k], g[k]), k
    assert np.array_equal(r["mt"], g["mt_key"]) and r["mt_pos"] == int(g["mt_pos"])
    for k in ("created", "updated", "resampled", "mean", "range_error"):
        assert np.allclose(r[k], g[k], rtol=1e-12, atol=1e-9), k


def test_restart_from_a_resampled_list_keeps_the_aliasing(orc):
    """splitting a run in two (state + object ids handed back in) gives the same second half"""
    from oracle import orc_pf
    g = np.load(FILES[1])
    mt, pos = orc_pf.np_seed_state(int(g["seed"]))
    full = orc_pf.run(1000, g["measurements"], g["shark_xy"], g["shark0"], mt, pos)
    a = orc_pf.run(1000, g["measurements"][:6], g["shark_xy"][:6], g["shark0"], mt, pos)
    b = orc_pf.run(1000, g["measurements"][6:], g["shark_xy"][6:], g["shark0"], a["mt"], a["mt_pos"],
                   init=a["resampled"][-1], init_obj=a["choice"][-1])
    assert np.array_equal(b["resampled"], full["resampled"][6:])
    assert np.array_equal(b["mt"], full["mt"]) and b["mt_pos"] == full["mt_pos"]
