"""root cost.py `Cost.habitat_shark_cost_func` (4 weights, a point outside every bin reuses the previous
point's bin) through the device cost kernel, against G12 captured from the reference."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_host_logic import _g12_world

pytestmark = pytest.mark.gpu


def test_cost_class_twin_matches_reference():
    from auv_sim_amd.cost import Cost
    from auv_sim_amd.motion_plan_state import Motion_plan_state as MPS
    g = json.load(open(os.path.join(GOLDEN, "g12_cost_twins.json")))
    cal = Cost()
    stale_cases = 0
    for c in g["twin"]:
        shark, habitats = _g12_world(c)
        path = [MPS(p[0], p[1], traj_time_stamp=p[2]) for p in c["pts"]]
        stale_cases += any(p[2] > 50.0 * c["n_bins"] for p in c["pts"])
        res = cal.habitat_shark_cost_func(path, c["length"], c["peri"], c["total"], habitats, shark, c["weights"])
        assert len(res[1]) == 4
        np.testing.assert_allclose([res[0]] + res[1], c["out"], rtol=1e-13, atol=1e-9)
    assert stale_cases >= 8
    shark, habitats = _g12_world(g["twin"][0])
    with pytest.raises(UnboundLocalError):
        cal.habitat_shark_cost_func([MPS(0, 0, traj_time_stamp=1e9)], 1.0, 1.0, 1.0, habitats, shark, [1, 1, 1, 1])
    with pytest.raises(ZeroDivisionError):
        cal.habitat_shark_cost_func([MPS(0, 0, traj_time_stamp=1.0)], 1.0, 1.0, 0, habitats, shark, [1, 1, 1, 1])
