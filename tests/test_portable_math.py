"""auv_sim_amd/csrc/auvp_math.h (host build, through the portable checker library): < 1 ulp vs mpmath,
and last-bit agreement rate with this machine's libm."""
import math

import numpy as np


def _ulp_err(got, want_mp, mp):
    want = float(want_mp)
    u = math.ulp(want) if want != 0 else 5e-324
    return abs(float((mp.mpf(got) - want_mp) / u))


def test_sincos_under_one_ulp(orc):
    import mpmath as mp
    mp.mp.prec = 200
    L = orc.lib("portable")
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-70, 70, 3000), rng.uniform(-1e5, 1e5, 500), rng.uniform(-1e-4, 1e-4, 100),
                         np.array([k * math.pi / 2 for k in range(-40, 41)]),
                         np.array([k * math.pi / 4 for k in range(-40, 41)])])
    worst = 0.0
    for x in xs:
        x = float(x)
        worst = max(worst, _ulp_err(L.orc_sin(x), mp.sin(mp.mpf(x)), mp), _ulp_err(L.orc_cos(x), mp.cos(mp.mpf(x)), mp))
    assert worst < 1.0, worst


def test_sincos_mostly_equal_to_libm(orc):
    L = orc.lib("portable")
    rng = np.random.default_rng(1)
    xs = rng.uniform(-60, 60, 20000)
    s = np.array([L.orc_sin(float(x)) for x in xs])
    c = np.array([L.orc_cos(float(x)) for x in xs])
    ds = np.abs(s - np.sin(xs)) / np.spacing(np.abs(np.sin(xs)))
    dc = np.abs(c - np.cos(xs)) / np.spacing(np.abs(np.cos(xs)))
    assert ds.max() <= 1.0 and dc.max() <= 1.0
    # informational bound: the two agree bit-for-bit on the vast majority of inputs
    assert (ds == 0).mean() > 0.9 and (dc == 0).mean() > 0.9
