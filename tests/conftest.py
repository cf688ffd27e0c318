import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# many tests steer kernel choices through AUVP_<NAME> in os.environ on live contexts (monkeypatch.setenv); the library reads the
# environment once, at auvp_create.  tests/env_options.py wraps the loaded library -- in the test process only, the product
# package has no such hook -- so that the current environment is pushed into the handle's options before every call
import env_options  # noqa: E402

env_options.install()
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def orc():
    """The CPU checker (oracle/), built on demand with gcc."""
    from oracle import orc as _orc
    _orc.build()
    return _orc


def libm_matches_golden():
    """True when this machine's libm reproduces the sin/cos/pow known answers captured next to the
    goldens -- only then can bit-exact equality with the reference floats be asserted."""
    import json
    import math
    k = json.load(open(os.path.join(GOLDEN, "g7_random_kat.json")))["libm"]
    for x, s, c, a, p in zip(k["x"], k["sin"], k["cos"], k["atan2_x_1"], k["pow2"]):
        if math.sin(x) != s or math.cos(x) != c or math.atan2(x, 1.0) != a or x ** 2 != p:
            return False
    return True


def golden_world(g):
    """World tables of a G3-style golden: stored arrays, or -- for the 40 000-cell bench world, whose 4.5 MB of
    tables are not committed -- rebuilt with synth.make_world(**world_kwargs) and checked against the stored SHA-256."""
    import hashlib
    import json
    import numpy as np
    keys = ("obstacles", "habitats", "polygon", "bins", "cells", "prob")
    if "cells" in g.files:
        return {k: g[k] for k in keys}
    from auv_sim_amd import synth
    kw = json.loads(str(g["world_kwargs"]))
    for k in ("box", "start"):
        if k in kw:
            kw[k] = tuple(kw[k])
    w = synth.make_world(**kw)
    h = hashlib.sha256()
    for k in keys:
        h.update(np.ascontiguousarray(w[k]).tobytes())
    assert h.hexdigest() == str(g["world_sha"]), "synth.make_world no longer reproduces the golden's world"
    return {k: w[k] for k in keys}


def uneven_grid(world, seed, lo=4.0, hi=17.0):
    """replace a synth world's uniform cells by a product grid with uneven column widths / row heights over the same box
    (row-major, prob re-drawn): the lower-bound guesses of the grid indexes are then wrong by several cells"""
    import random
    import numpy as np
    rng = random.Random(seed)
    x0, y0, x1, y1 = [float(v) for v in world["box"]]

    def edges(a, b):
        e = [a]
        while e[-1] < b - 1e-9:
            e.append(min(b, e[-1] + round(rng.uniform(lo, hi), 2)))
        if e[-1] - e[-2] < 1.0:  # no sliver at the end
            e.pop(-2)
        return e
    ex, ey = edges(x0, x1), edges(y0, y1)
    cells = [(ex[c], ey[r], ex[c + 1], ey[r + 1]) for r in range(len(ey) - 1) for c in range(len(ex) - 1)]
    T = len(world["bins"])
    w = dict(world)
    w["cells"] = np.array(cells, dtype=np.float64)
    w["prob"] = np.array([[rng.uniform(0.0, 0.3) for _ in cells] for _ in range(T)], dtype=np.float64)
    return w
