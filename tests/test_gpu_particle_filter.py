"""The shark particle filter on the GPU (csrc/pf_kernel.h through auvp_pf_*) against the G10 goldens
(the reference's own run) and the CPU checker.  Bars: HIP == checker(portable math) bit-for-bit on
every float, drawn index and the final MT19937 state; HIP vs reference: drawn indices, aliasing and
RNG state exact, floats <= 1e-9 (libm vs portable last-bit differences)."""
import contextlib
import glob
import io
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
FILES = sorted(glob.glob(os.path.join(GOLDEN, "g10_pf_*.npz")))
IDS = [os.path.basename(f)[:-4] for f in FILES]


@pytest.fixture(scope="module")
def ctx():
    from auv_sim_amd import _lib
    return _lib.Context(0)


def _alias_first(choice_row):
    first = {}
    return [first.setdefault(int(x), i) for i, x in enumerate(choice_row)]


@pytest.mark.parametrize("path", FILES, ids=IDS)
def test_one_launch_matches_checker_and_reference(ctx, orc, path):
    from auv_sim_amd import _pf_lib
    from oracle import orc_pf
    g = np.load(path)
    mt, pos = _pf_lib.np_seed_state(int(g["seed"]))
    ref = orc_pf.run(1000, g["measurements"], g["shark_xy"], g["shark0"], mt, pos, kind="portable")
    b = _pf_lib.FilterBatch(ctx, 1, 1000).create([g["shark0"]], mt, pos)
    created, obj = b.particles()
    assert np.array_equal(created[0], ref["created"]) and np.array_equal(obj[0], np.arange(1000))
    assert np.allclose(created[0], g["created"], rtol=0, atol=1e-9)
    b.run(meas=g["measurements"][:, None], shark_xy=g["shark_xy"][:, None], log=True)
    upd, cho = b.step_log()
    mean, err, ll = b.estimates()
    final, obj = b.particles()
    st, nd = b.status()
    mt1, pos1 = b.rng_state()
    assert st[0] == 0
    # == checker, bit for bit
    assert np.array_equal(upd[:, 0], ref["updated"])
    assert np.array_equal(cho[:, 0], ref["choice"])
    assert np.array_equal(ll[:, 0], ref["list_len"])
    assert np.array_equal(mean[:, 0], ref["mean"]) and np.array_equal(err[:, 0], ref["range_error"])
    assert np.array_equal(final[0], ref["resampled"][-1]) and np.array_equal(obj[0], ref["choice"][-1])
    assert np.array_equal(mt1[0], ref["mt"]) and pos1[0] == ref["mt_pos"]
    assert int(nd[0]) == ref["n_draw32"]
    # vs the reference's own run
    assert np.array_equal(cho[:, 0], g["choice"])
    assert [_alias_first(r) for r in cho[:, 0]] == g["alias_first"].tolist()
    assert np.array_equal(mt1[0], g["mt_key"]) and pos1[0] == int(g["mt_pos"])
    assert np.allclose(upd[:, 0], g["updated"], rtol=1e-12, atol=1e-9)
    assert np.allclose(final[0], g["resampled"][-1], rtol=1e-12, atol=1e-9)
    assert np.allclose(mean[:, 0], g["mean"], rtol=1e-12, atol=1e-9)
    assert np.allclose(err[:, 0], g["range_error"], rtol=1e-12, atol=1e-9)


def test_step_by_step_equals_one_launch(ctx):
    """S launches of one step (state, object ids and RNG persistent in HBM) == one launch of S steps,
    and the per-step resampled lists match the reference's"""
    from auv_sim_amd import _pf_lib
    g = np.load(FILES[1])
    mt, pos = _pf_lib.np_seed_state(int(g["seed"]))
    b = _pf_lib.FilterBatch(ctx, 1, 1000).create([g["shark0"]], mt, pos)
    for s in range(len(g["measurements"])):
        b.run(phases=_pf_lib.UPDATE, n_steps=1)
        b.run(meas=g["measurements"][s][None, None], shark_xy=g["shark_xy"][s][None, None],
              phases=_pf_lib.WEIGHTS | _pf_lib.MEAN)
        res, obj = b.particles()
        mean, err, _ = b.estimates()
        assert np.array_equal(obj[0], g["choice"][s])
        assert np.allclose(res[0], g["resampled"][s], rtol=1e-12, atol=1e-9)
        assert np.allclose(mean[0, 0], g["mean"][s], rtol=1e-12, atol=1e-9)
        assert abs(err[0, 0] - g["range_error"][s]) < 1e-9
    mt1, pos1 = b.rng_state()
    assert np.array_equal(mt1[0], g["mt_key"]) and pos1[0] == int(g["mt_pos"])


@pytest.mark.parametrize("n_particles,n_auv,n_filters", [(1000, 2, 24), (300, 1, 16), (2048, 3, 6), (257, 1, 5), (64, 2, 3)])
def test_batches_match_checker(ctx, orc, n_particles, n_auv, n_filters):
    from auv_sim_amd import _pf_lib
    from oracle import orc_pf
    rng = np.random.default_rng(n_particles)
    F, N, A, S = n_filters, n_particles, n_auv, 6
    seeds = rng.integers(0, 2 ** 32, size=F)
    shark0 = rng.uniform(-500, 500, size=(F, 2))
    meas = np.zeros((S, F, A, 5))
    meas[..., 0:2] = shark0[None, :, None, :] + rng.uniform(-200, 200, size=(S, F, A, 2))
    meas[..., 2] = rng.uniform(-np.pi, np.pi, size=(S, F, A))
    meas[..., 3] = rng.uniform(0, 300, size=(S, F, A))   # robotSim rows: [3] = range, [4] = bearing
    meas[..., 4] = rng.uniform(-np.pi, np.pi, size=(S, F, A))
    shark = shark0[None] + rng.uniform(-30, 30, size=(S, F, 2))
    mts = np.stack([_pf_lib.np_seed_state(int(s))[0] for s in seeds])
    pos = rng.integers(0, 625, size=F).astype(np.int32)  # mid-block stream positions too
    b = _pf_lib.FilterBatch(ctx, F, N).create(shark0, mts, pos)
    b.run(meas=meas, shark_xy=shark, log=True)
    upd, cho = b.step_log()
    mean, err, ll = b.estimates()
    final, obj = b.particles()
    st, nd = b.status()
    mt1, pos1 = b.rng_state()
    assert (st == 0).all()
    for f in range(F):
        ref = orc_pf.run(N, meas[:, f], shark[:, f], shark0[f], mts[f], int(pos[f]), kind="portable")
        assert ref["status"] == 0
        assert np.array_equal(upd[:, f], ref["updated"]), f
        assert np.array_equal(cho[:, f], ref["choice"]), f
        assert np.array_equal(ll[:, f], ref["list_len"])
        assert np.array_equal(final[f], ref["resampled"][-1])
        assert np.array_equal(mean[:, f], ref["mean"]) and np.array_equal(err[:, f], ref["range_error"])
        assert np.array_equal(mt1[f], ref["mt"]) and pos1[f] == ref["mt_pos"] and int(nd[f]) == ref["n_draw32"]


def test_restart_from_uploaded_lists_keeps_aliasing(ctx, orc):
    from auv_sim_amd import _pf_lib
    g = np.load(FILES[0])
    mt, pos = _pf_lib.np_seed_state(int(g["seed"]))
    m, sx = g["measurements"][:, None], g["shark_xy"][:, None]
    full = _pf_lib.FilterBatch(ctx, 1, 1000).create([g["shark0"]], mt, pos).run(meas=m, shark_xy=sx)
    want, want_obj = full.particles()
    want_mt, want_pos = full.rng_state()
    a = _pf_lib.FilterBatch(ctx, 1, 1000).create([g["shark0"]], mt, pos).run(meas=m[:5], shark_xy=sx[:5])
    part, obj = a.particles()
    mt5, pos5 = a.rng_state()
    b = _pf_lib.FilterBatch(ctx, 1, 1000).set_particles(part, mt5, pos5, obj=obj).run(meas=m[5:], shark_xy=sx[5:])
    got, got_obj = b.particles()
    got_mt, got_pos = b.rng_state()
    assert np.array_equal(got, want) and np.array_equal(got_obj, want_obj)
    assert np.array_equal(got_mt, want_mt) and got_pos[0] == want_pos[0]
    # without the object ids the aliased positions move independently: a different (non-reference) run
    c = _pf_lib.FilterBatch(ctx, 1, 1000).set_particles(part, mt5, pos5).run(meas=m[5:], shark_xy=sx[5:])
    assert not np.array_equal(c.particles()[0], want)


@pytest.mark.parametrize("path", FILES[:2], ids=IDS[:2])
def test_dropin_class_used_like_robotsim(path):
    """np.random.seed(k) + the reference's call sequence (robotSim.py:665-701) through the drop-in class"""
    from auv_sim_amd.particleFilter import ParticleFilter
    g = np.load(path)
    np.random.seed(int(g["seed"]))
    pf = ParticleFilter(float(g["shark0"][0]), float(g["shark0"][1]), [])
    particles = pf.create()
    assert len(particles) == 1000
    assert np.allclose(particles.as_array(), g["created"], rtol=0, atol=1e-9)
    for s in range(6):
        particles = pf.create_and_update(particles)
        assert np.allclose(particles.as_array(), g["updated"][s], rtol=1e-12, atol=1e-9)
        rows = [list(r) + [1] for r in g["measurements"][s].tolist()]
        pf.x_shark, pf.y_shark = g["shark_xy"][s].tolist()
        with contextlib.redirect_stdout(io.StringIO()):
            particles = pf.update_weights(particles, rows)
            xy = pf.particleMean(particles)
            e = pf.meanError(xy[0], xy[1])
        assert np.allclose(particles.as_array(), g["resampled"][s], rtol=1e-12, atol=1e-9)
        assert np.allclose(xy, g["mean"][s], atol=1e-9) and abs(e - g["range_error"][s]) < 1e-9
    first = next(iter(particles))
    assert abs(first.x_p - g["resampled"][5][0, 0]) < 1e-9 and hasattr(first, "weight_p")
    # numpy's global stream sits where the reference's run left it after 6 steps
    from oracle import orc_pf
    mt, pos = orc_pf.np_seed_state(int(g["seed"]))
    ref = orc_pf.run(1000, g["measurements"][:6], g["shark_xy"][:6], g["shark0"], mt, pos)
    st = np.random.get_state()
    assert np.array_equal(st[1], ref["mt"]) and st[2] == ref["mt_pos"]
    with pytest.raises(ValueError):
        pf.create_and_update(pf._live.__class__(pf))  # a list that is not the resident one


def test_capacity_and_state_errors(ctx):
    from auv_sim_amd import _lib, _pf_lib
    mt, pos = _pf_lib.np_seed_state(1)
    with pytest.raises(_lib.AuvpError):
        _pf_lib.FilterBatch(ctx, 1, 4096).create([[0, 0]], mt, pos)
    with pytest.raises(_lib.AuvpError):
        _pf_lib.FilterBatch(ctx, 1, 100).create([[0, 0]], mt, 700)
