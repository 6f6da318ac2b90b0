"""Column spaces of the mirrored cap kernels (OGG_SYM_MIRROR), replayed on the host by the library itself (ogg_symmetry_coverage: the same
quad_lane / mesh_lane / dq_lane functions the kernels call): whatever the size, every cell and every node column of a row is written exactly
once, with and without symmetry; mirrored, the evaluated share is what DESIGN.md says.  No GPU."""
import ctypes

import numpy as np
import pytest

from ocean_model_grid_generator_amd import _lib as L


def coverage(which, n, symmetry, order=4, lon0=-300.0, lon_dp=80.0):
    cells = np.zeros(n, dtype=np.int32)
    cols = np.zeros(n + 1, dtype=np.int32)
    ev = ctypes.c_long(0)
    L.call("ogg_symmetry_coverage", which, order, n, lon0, lon_dp, symmetry, cells.ctypes.data if which != 1 else None, cols.ctypes.data,
           ctypes.byref(ev))
    return cells, cols, ev.value


SIZES = [4, 8, 12, 60, 64, 180, 240, 252, 360, 720, 1000, 1440, 2880, 3600, 5760, 11520, 23040, 63 * 4, 63 * 8, 62 * 4, 126, 61, 1441]


@pytest.mark.parametrize("which", [0, 1, 2])
@pytest.mark.parametrize("symmetry", [L.SYM_MIRROR, L.SYM_NONE])
def test_every_cell_and_column_written_once(which, symmetry):
    for n in SIZES:
        cells, cols, ev = coverage(which, n, symmetry)
        assert np.all(cols == 1), (which, symmetry, n, np.nonzero(cols != 1)[0][:8].tolist(), cols[cols != 1][:8].tolist())
        if which != 1:
            assert np.all(cells == 1), (which, symmetry, n, np.nonzero(cells != 1)[0][:8].tolist())
        if symmetry == L.SYM_NONE:
            assert ev == n + 1 if which != 2 else ev == n


def test_mirrored_share_of_the_baseline_sizes():
    """1/8 degree: the bipolar quadrature evaluates 30 % of the columns of a row (a quarter + 6 degrees at each of the three fold-line
    neighbourhoods), the mesh 27 %, the displaced-pole quadrature half."""
    for n in (2880, 5760, 11520):
        assert 0.29 < coverage(0, n, L.SYM_MIRROR)[2] / (n + 1.0) < 0.31
        assert 0.26 < coverage(1, n, L.SYM_MIRROR)[2] / (n + 1.0) < 0.29
        assert coverage(2, n, L.SYM_MIRROR)[2] == n // 2


def test_symmetry_is_declined_where_it_does_not_apply():
    # Ni not a multiple of 4: no pole-meridian column; the displaced pole's meridian between two columns: every column evaluated
    assert coverage(0, 1442, L.SYM_MIRROR)[2] == 1443
    assert coverage(1, 1442, L.SYM_MIRROR)[2] == 1443
    assert coverage(2, 1440, L.SYM_MIRROR, lon_dp=80.1)[2] == 1440
    assert coverage(2, 1441, L.SYM_MIRROR)[2] == 1441
    for lon_dp in (80.0, -100.0, 260.0, -300.0, 60.0, 0.0, 59.75):   # meridians on node columns, either half of the row, the seam itself
        cells, cols, ev = coverage(2, 1440, L.SYM_MIRROR, lon_dp=lon_dp)
        assert ev == 720 and np.all(cells == 1) and np.all(cols == 1), lon_dp
    for order in (2, 4):
        cells, cols, ev = coverage(2, 5760, L.SYM_MIRROR, order=order)
        assert ev == 2880 and np.all(cells == 1) and np.all(cols == 1)


def test_default_follows_the_environment(monkeypatch):
    monkeypatch.delenv("OGG_CAP_SYMMETRY", raising=False)
    assert coverage(0, 5760, L.SYM_DEFAULT)[2] < 2000          # mirrored by default
    monkeypatch.setenv("OGG_CAP_SYMMETRY", "0")
    assert coverage(0, 5760, L.SYM_DEFAULT)[2] == 5761
    assert coverage(0, 5760, L.SYM_MIRROR)[2] < 2000           # an explicit request wins
    monkeypatch.setenv("OGG_CAP_SYMMETRY", "1")
    assert coverage(0, 5760, L.SYM_NONE)[2] == 5761
