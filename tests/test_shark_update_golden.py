"""SharkUpdate (auv_sim_amd/sharkEstimate.py) against G14: every method's return value, what it does to its arguments,
and the exceptions the reference ends with (tests/golden/make_golden.py g14 imports the reference to record them)."""
import copy
import json
import os

import pytest

from conftest import GOLDEN


class _Box:
    def __init__(self, *b):
        self.bounds = tuple(b)


CASES = json.load(open(os.path.join(GOLDEN, "g14_shark_update.json")))["cases"]


def _raises(name):
    return {"TypeError": TypeError, "IndexError": IndexError, "UnboundLocalError": UnboundLocalError,
            "ZeroDivisionError": ZeroDivisionError}[name]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_shark_update_matches_reference(case):
    from auv_sim_amd.sharkEstimate import SharkUpdate
    cells = [_Box(*c) for c in case["cells"]]
    upd = SharkUpdate(_Box(*case["box"]), case["cell_size"], cells)
    prev, inf, parts = case["prev"], case["inf"], case["particles"]
    assert [list(upd.cellToIndex(c)) for c in cells] == case["index"]
    assert upd.prediction1(copy.deepcopy(prev), 0.6) == case["prediction1"]
    a, b = copy.deepcopy(prev), copy.deepcopy(inf)
    assert upd.prediction2(a, 0.1, b) == case["prediction2"]
    assert a == case["prediction2_arg_after"] and b == inf
    for meth in (1, 2):
        a = copy.deepcopy(prev)
        assert upd.predictOnAve(a, False, meth, 0.6, 0.1) == case["ave_m%d" % meth]
        assert a == case["ave_m%d_arg_after" % meth]
        assert upd.predictOnAve(None, True, meth, 0.6, 0.1) == case["ave_exp_m%d" % meth]
        a, b = copy.deepcopy(prev), copy.deepcopy(inf)
        assert upd.predictOnHist(a, False, meth, b, 0.6, 0.1) == case["hist_m%d" % meth]
        b = copy.deepcopy(inf)
        assert upd.predictOnHist(None, True, meth, b, 0.6, 0.1) == case["hist_exp_m%d" % meth]
        assert b == case["hist_exp_m%d_inf_after" % meth]
    assert upd.correction(parts, copy.deepcopy(case["prediction1"])) == case["correction"]
    for meth in (["ave", 1], ["ave", 2], ["hist", 1], ["hist", 2]):
        key = "update_%s%d" % (meth[0], meth[1])
        res = upd.update((0, 10), {(0, 10): copy.deepcopy(prev)}, 10, 10, meth)
        assert {"%d,%d" % k: v for k, v in res.items()} == case[key + "_one_round"]
        with pytest.raises(_raises(case[key + "_two_rounds_raises"])):
            upd.update((0, 10), {(0, 10): copy.deepcopy(prev)}, 20, 10, meth)
    with pytest.raises(_raises(case["ave_m3_raises"])):
        upd.predictOnAve(copy.deepcopy(prev), False, 3, 0.6, 0.1)
    with pytest.raises(_raises(case["correction_zero_raises"])):
        upd.correction([[0] * len(prev[0]) for _ in prev], copy.deepcopy(case["prediction1"]))


def test_replanning_can_hold_a_shark_update():
    """rrt_dubins.py:67 builds one from the planner's boundary and cell list"""
    from auv_sim_amd.sharkEstimate import SharkUpdate
    cells = [_Box(0.0, 0.0, 10.0, 10.0), _Box(10.0, 0.0, 20.0, 10.0)]
    upd = SharkUpdate(_Box(0.0, 0.0, 20.0, 10.0), 10, cells)
    assert upd.cellToIndex(cells[1]) == (0, 1) and len(upd.predictOnAve(None, True, 1, 0.6, 0.1)) == 2
