"""oracle/orc_sog.c (SharkOccupancyGrid.convert restatement) against the G9 goldens captured from the
reference (tests/golden/make_golden.py g9).  No libm calls on this path: bit-exact on any host."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN

FILES = sorted(glob.glob(os.path.join(GOLDEN, "g9_sog_*.npz")))


def test_goldens_present():
    assert len(FILES) == 3


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
@pytest.mark.parametrize("kind", ["libm", "portable"])
def test_oracle_matches_reference_grids(orc, path, kind):
    from oracle import orc_sog
    g = np.load(path)
    r = orc_sog.convert(g["cells"], g["box"], float(g["cell_size"]), float(g["bin_interval"]), float(g["detect_range"]),
                        g["traj_len"], g["points"], kind=kind)
    assert r["status"] == 0
    assert np.array_equal(r["bins"], g["bins"])
    assert r["grids"].shape == g["grids"].shape
    assert np.array_equal(r["grids"], g["grids"])  # bit-exact
    assert np.array_equal(r["occ"], g["occ_bin0_shark1"])
    assert np.array_equal(r["auv"], g["auv_bin0_shark1"])


def test_oracle_cell_outside_grid_is_index_error(orc):
    from oracle import orc_sog
    r = orc_sog.convert([[100.0, 0.0, 102.0, 2.0]], (0, 0, 10, 10), 2.0, 2.0, 4.0, [3],
                        [[1, 1, 1.0], [2, 2, 2.0], [3, 3, 4.5]])
    assert r["status"] != 0


def test_oracle_duplicate_cells_add_the_window_twice(orc):
    """two listed cells mapping to one grid cell: constructAUVGrid adds the window sum once per listing"""
    from oracle import orc_sog
    cells = [[0.0, 0.0, 2.0, 2.0], [2.0, 0.0, 4.0, 2.0]]
    pts = [[1.0, 1.0, 0.5], [3.0, 1.0, 1.0], [3.0, 1.5, 2.0]]
    one = orc_sog.convert(cells, (0, 0, 4, 2), 2.0, 2.0, 2.0, [3], pts)
    two = orc_sog.convert(cells + [cells[0]], (0, 0, 4, 2), 2.0, 2.0, 2.0, [3], pts)
    assert one["status"] == 0 and two["status"] == 0
    assert two["grids"][0, 0, 0] > one["grids"][0, 0, 0] * 1.5
