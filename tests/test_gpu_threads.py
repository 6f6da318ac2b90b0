"""Two Python threads in the host-pointer layer at once (ctypes releases the GIL; the library serialises callers with one mutex per device,
SURVEY 8b "Threading"): the results must be the serial bits, every time."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_two_threads_give_the_serial_bits(hip):
    import contextlib
    import io

    import ocean_model_grid_generator_amd.ocean_grid_generator as ogg
    from oracle import ogg_oracle as orc

    nx, ny, lat0, lon_bp = 720, 120, 64.05895973, -300.0
    rp = float(np.tan(0.5 * (90 - lat0) * np.pi / 180))
    xm, ym = orc.generate_latlon_grid(720, 110, -300.0, 360.0, -78.0, 11.0, ensure_nj_even=False)
    xm, ym = np.ascontiguousarray(xm), np.ascontiguousarray(ym)
    with contextlib.redirect_stdout(io.StringIO()):
        want_q = ogg.bipolar_cap_metrics_quad_fast(5, nx, ny, lat0, lon_bp, rp)
        want_m = ogg.generate_grid_metrics_MIDAS(xm, ym)
        want_d = ogg.displacedPoleCap_metrics_quad(4, 360, 70, -300.0, -78.0, 80.0, 0.2)
    bad = []

    def quad():
        for _ in range(50):
            got = ogg.bipolar_cap_metrics_quad_fast(5, nx, ny, lat0, lon_bp, rp)
            if not all(np.array_equal(a, b) for a, b in zip(got, want_q)):
                bad.append("bipolar quadrature")
                return

    def midas():
        for k in range(50):
            got = ogg.generate_grid_metrics_MIDAS(xm, ym)
            if not all(np.array_equal(a, b) for a, b in zip(got, want_m)):
                bad.append("MIDAS")
                return
            if k % 5 == 0:
                got = ogg.displacedPoleCap_metrics_quad(4, 360, 70, -300.0, -78.0, 80.0, 0.2)
                if not all(np.array_equal(a, b) for a, b in zip(got, want_d)):
                    bad.append("displaced-pole quadrature")
                    return

    with contextlib.redirect_stdout(io.StringIO()):   # (the reference's progress prints)
        ts = [threading.Thread(target=quad), threading.Thread(target=midas)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(600)
    assert not bad and not any(t.is_alive() for t in ts), bad
