"""CPU oracle for the supergrid hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  The product path (``ocean_model_grid_generator_amd``) never imports it and fails loudly when the
HIP library is missing.

What it is: a plain-numpy restatement of the per-cell coordinate-transform and metrics path of the reference
script ``ocean_grid_generator.py`` (cited below as OGG:<line>), written so that every floating-point
operation happens in the same order as in the reference.  It is *vectorised* where the reference loops in
Python (the per-cell quadrature loops OGG:180-184, OGG:591-595 and the column loop of OGG:470-475); the
element-wise operation order is unchanged, so the results are bit-identical.

How it is pinned: the reference imports ``numpypi.numpypi_series`` (OGG:5), a third-party module that is not
vendored, not pinned to a version (setup.cfg:22 names a git branch) and not installed in this image.  The
oracle is therefore pinned against the *unmodified reference file executed under a numpy stand-in for that
module* (``tests/golden/make_golden.py``; fixtures in ``tests/golden``) -- bit-for-bit in the build
container.  Bit-level parity with numpypi's own arithmetic is UNPINNED (its source is unavailable here).

All angles in degrees unless a name ends in ``_rad``.  All arrays float64, C order, shape (nj, ni).
"""
import numpy as np

# OGG:13-16
PI_180 = np.pi / 180.0
RE_DEFAULT = 6371.0e3
HUGE = 1.0e30


# ----------------------------------------------------------------------------------------------
# quadrature helpers (OGG:191-255)
# ----------------------------------------------------------------------------------------------
def quad_positions(n=3):
    """Gauss-Lobatto node weights (wa, wb): node = wb*x_a + wa*x_b.  OGG:191-204."""
    if n == 2:
        return np.array([0.0, 1.0]), np.array([1.0, 0.0])
    if n == 3:
        return np.array([0.0, 0.5, 1.0]), np.array([1.0, 0.5, 0.0])
    if n == 4:
        r5 = 0.5 / np.sqrt(5.0)
        return np.array([0.0, 0.5 - r5, 0.5 + r5, 1.0]), np.array([1.0, 0.5 + r5, 0.5 - r5, 0.0])
    if n == 5:
        r37 = 0.5 * np.sqrt(3.0 / 7.0)
        return (np.array([0.0, 0.5 - r37, 0.5, 0.5 + r37, 1.0]),
                np.array([1.0, 0.5 + r37, 0.5, 0.5 - r37, 0.0]))
    raise Exception("Uncoded order")


_W2D = {4: np.array([1.0, 5.0, 5.0, 1.0]), 5: np.array([9.0, 49.0, 64.0, 49.0, 9.0])}
_D1D = {2: 1.0 / 2.0, 3: 1.0 / 6.0, 4: 1.0 / 12.0, 5: 1.0 / 180.0}


def quad_average(y):
    """1-D Lobatto mean over the LAST axis of ``y`` (length n).  OGG:207-222, vectorised."""
    n = y.shape[-1]
    if n == 2:
        return _D1D[2] * (y[..., 0] + y[..., 1])
    if n == 3:
        return _D1D[3] * (4.0 * y[..., 1] + (y[..., 0] + y[..., 2]))
    if n == 4:
        return _D1D[4] * (5.0 * (y[..., 1] + y[..., 2]) + (y[..., 0] + y[..., 3]))
    if n == 5:
        return _D1D[5] * (64.0 * y[..., 2] + (49.0 * (y[..., 1] + y[..., 3])) + 9.0 * (y[..., 0] + y[..., 4]))
    raise Exception("Uncoded order")


def quad_average_2d(y):
    """2-D Lobatto mean of ``y[..., jj, ii]`` over the last two axes.  OGG:225-255, vectorised;
    the (jj, ii) accumulation order of OGG:242-244 / 250-252 is kept."""
    if y.shape[-1] != y.shape[-2]:
        raise Exception("Input array is not squared!")
    n = y.shape[-1]
    if n == 2:
        d = _D1D[2]
        return d * d * (y[..., 0, 0] + y[..., 0, 1] + y[..., 1, 0] + y[..., 1, 1])
    if n == 3:
        d = _D1D[3]
        return (d * d * (y[..., 0, 0] + y[..., 0, 2] + y[..., 2, 0] + y[..., 2, 2]
                         + 4.0 * (y[..., 0, 1] + y[..., 1, 0] + y[..., 1, 2] + y[..., 2, 1]
                                  + 4.0 * y[..., 1, 1])))
    if n in (4, 5):
        d = _D1D[n]
        w = _W2D[n]
        ysum = 0.0
        for jj in range(n):
            for ii in range(n):
                ysum = ysum + w[ii] * w[jj] * y[..., jj, ii]
        return d * d * ysum
    raise Exception("Uncoded order")


def _lattice_1d(n_cells, order):
    """Concatenated Lobatto nodes of cells 0..n_cells (inclusive): OGG:143-156 / 573-581."""
    a, b = quad_positions(order)
    k = np.arange(0, n_cells + 1, dtype=np.int64)
    # per cell: b*k + a*(k+1); python-int times float array, as in the reference loop
    return (b[None, :] * k[:, None] + a[None, :] * (k[:, None] + 1)).reshape(-1)


# ----------------------------------------------------------------------------------------------
# modular distance, MIDAS metrics, orientation angle (OGG:682-729)
# ----------------------------------------------------------------------------------------------
def mdist(x1, x2):
    """Positive distance modulo 360 (python-sign mod).  OGG:682-684."""
    return np.minimum(np.mod(x1 - x2, 360.0), np.mod(x2 - x1, 360.0))


def generate_grid_metrics_MIDAS(x, y, Re=RE_DEFAULT, latlon_areafix=True):
    """dx (nj+1,ni), dy (nj,ni+1), area (nj,ni) from the 2x2 stencil.  OGG:687-716 (dead roll
    temporaries OGG:704-709 not reproduced)."""
    lv = (0.5 * (y[:, 1:] + y[:, :-1])) * PI_180
    dx_i = mdist(x[:, 1:], x[:, :-1]) * PI_180
    dy_i = (y[:, 1:] - y[:, :-1]) * PI_180
    dx = Re * np.sqrt(dy_i ** 2 + (dx_i * np.cos(lv)) ** 2)
    lu = (0.5 * (y[1:, :] + y[:-1, :])) * PI_180
    dx_j = mdist(x[1:, :], x[:-1, :]) * PI_180
    dy_j = (y[1:, :] - y[:-1, :]) * PI_180
    dy = Re * np.sqrt(dy_j ** 2 + (dx_j * np.cos(lu)) ** 2)
    if latlon_areafix:
        sl = np.sin(lv)
        area = (Re ** 2) * ((0.5 * (dx_i[1:, :] + dx_i[:-1, :])) * (sl[1:, :] - sl[:-1, :]))
    else:
        area = 0.25 * ((dx[1:, :] + dx[:-1, :]) * (dy[:, 1:] + dy[:, :-1]))
    return dx, dy, area


def angle_x(x, y):
    """Grid orientation angle, centred in i, one-sided at row ends.  OGG:719-729."""
    if x.shape != y.shape:
        raise Exception("Input arrays do not have the same shape!")
    ang = np.zeros(x.shape)
    ang[:, 1:-1] = np.arctan2(y[:, 2:] - y[:, :-2], (x[:, 2:] - x[:, :-2]) * np.cos(y[:, 1:-1] * PI_180))
    ang[:, 0] = np.arctan2(y[:, 1] - y[:, 0], (x[:, 1] - x[:, 0]) * np.cos(y[:, 0] * PI_180))
    ang[:, -1] = np.arctan2(y[:, -1] - y[:, -2], (x[:, -1] - x[:, -2]) * np.cos(y[:, -1] * PI_180))
    return ang / PI_180


def metrics_error(dx_, dy_, area_, Ni, lat1, lat2=90, Re=RE_DEFAULT, bipolar=False, displaced_pole=-999):
    """Percent errors of summed area / meridian arc / parallel arc vs the analytic sphere.  OGG:732-770."""
    exact_area = 2 * np.pi * (Re ** 2) * np.abs(np.sin(lat2 * PI_180) - np.sin(lat1 * PI_180))
    exact_lat_arc = np.abs(lat2 - lat1) * PI_180 * Re
    exact_lon_arc = np.cos(lat1 * PI_180) * 2 * np.pi * Re
    lat_arc = np.sum(dy_[:, Ni // 4])
    lon_arc = np.sum(dx_[0, :])
    if lat1 > lat2:
        lon_arc = np.sum(dx_[-1, :])
    if bipolar:
        lon_arc2 = np.sum(dx_[-1, :])
        lon_arc2_error = 100 * (lon_arc2 / 4 - exact_lat_arc) / exact_lat_arc
    area_error = 100 * (np.sum(area_) - exact_area) / exact_area
    lat_arc_error = 100 * (lat_arc - exact_lat_arc) / exact_lat_arc
    lon_arc_error = 100 * (lon_arc - exact_lon_arc) / exact_lon_arc
    if displaced_pole != -999:
        antipole = displaced_pole + Ni // 2
        if displaced_pole > Ni // 2:
            antipole = displaced_pole - Ni // 2
        lat_arc = np.sum(dy_[:, displaced_pole]) + np.sum(dy_[:, antipole])
        lat_arc_error = 100 * (lat_arc - 2.0 * exact_lat_arc) / exact_lat_arc
    if bipolar:
        return area_error, lat_arc_error, lon_arc_error, lon_arc2_error
    return area_error, lat_arc_error, lon_arc_error


# ----------------------------------------------------------------------------------------------
# Mercator and regular lat-lon builders (OGG:292-441, 832-846)
# ----------------------------------------------------------------------------------------------
def y_mercator(Ni, phi_rad):
    """OGG:292-295."""
    R = Ni / (2 * np.pi)
    return R * (np.log((1.0 + np.sin(phi_rad)) / np.cos(phi_rad)))


def phi_mercator(Ni, y):
    """OGG:298-301 (degrees out)."""
    R = Ni / (2 * np.pi)
    return np.arctan(np.sinh(y / R)) * (180 / np.pi)


def y_mercator_rounded(Ni, phi_rad):
    """OGG:309-311 (numpy.round is half-to-even)."""
    yf = y_mercator(Ni, phi_rad)
    return (np.sign(yf) * np.round(np.abs(yf))).astype(int)


def mercator_y_star(Ni, phi_s, phi_n, shift_equator_to_u_point=True, ensure_nj_even=True):
    """Integer Mercator ordinate range after the parity fixes.  OGG:318-334."""
    y_star = y_mercator_rounded(Ni, np.array([phi_s * PI_180, phi_n * PI_180]))
    if y_star[0] % 2 == 0 and shift_equator_to_u_point:
        y_star[0] = y_star[0] - 1
        y_star[1] = y_star[1] - 1
    if (y_star[1] - y_star[0] + 1) % 2 == 0 and ensure_nj_even:
        y_star[1] = y_star[1] - 1
    return int(y_star[0]), int(y_star[1])


def lagrange_interp(x, y, q):
    """4-point Lagrange polynomial through (x[k], y[k]) at q.  OGG:258-270."""
    n0 = (q - x[1]) * (q - x[2]) * (q - x[3]); d0 = (x[0] - x[1]) * (x[0] - x[2]) * (x[0] - x[3])
    n1 = (q - x[0]) * (q - x[2]) * (q - x[3]); d1 = (x[1] - x[0]) * (x[1] - x[2]) * (x[1] - x[3])
    n2 = (q - x[0]) * (q - x[1]) * (q - x[3]); d2 = (x[2] - x[0]) * (x[2] - x[1]) * (x[2] - x[3])
    n3 = (q - x[0]) * (q - x[1]) * (q - x[2]); d3 = (x[3] - x[0]) * (x[3] - x[1]) * (x[3] - x[2])
    return ((n0 / d0) * y[0] + (n3 / d3) * y[3]) + ((n1 / d1) * y[1] + (n2 / d2) * y[2])


def lagrange_interp_6pt(x, y, q):
    """6-point Lagrange polynomial.  OGG:272-289."""
    terms = []
    for k in range(6):
        n = 1.0
        d = 1.0
        first = True
        for m in range(6):
            if m == k:
                continue
            n = (q - x[m]) if first else n * (q - x[m])
            d = (x[k] - x[m]) if first else d * (x[k] - x[m])
            first = False
        terms.append((n / d) * y[k])
    return (terms[0] + terms[5]) + (terms[1] + terms[4]) + (terms[2] + terms[3])


def mercator_axis(Ni, phi_s, phi_n, refineR, shift_equator_to_u_point=True, ensure_nj_even=True,
                  enhanced_equatorial=0):
    """1-D Mercator latitude axis incl. the optional enhanced-equator splice.  OGG:318-428."""
    y0, y1 = mercator_y_star(Ni, phi_s, phi_n, shift_equator_to_u_point, ensure_nj_even)
    phi_M = phi_mercator(Ni, np.arange(y0, y1 + 1))
    if np.searchsorted(phi_M, 0.0) == 0:
        raise Exception("   Ooops: Equator is not in the grid")
    if enhanced_equatorial:
        phi_M = enhance_equator(phi_M, phi_n, refineR, enhanced_equatorial)
    return phi_M


def enhance_equator(phi_M, phi_n, refineR, enhanced_equatorial):
    """Splice: pure Mercator | Lagrange shoulder | uniform band | mirror.  OGG:349-428."""
    use4, use6 = True, False
    phi_enh_d, phi_cub_d = -5.0, -30
    N_cub = 132 * refineR / 2
    dphi_e = 0.13 * 2 / refineR
    N_enh = 40 * refineR / 2
    if refineR == 1 and enhanced_equatorial:
        phi_enh_d, phi_cub_d, N_cub, N_enh = -10, -20, 29, 55
        dphi_e = -phi_enh_d / N_enh / 0.981
    if refineR == 4 and enhanced_equatorial == 8:
        phi_enh_d = -10
        N_enh = 2 * enhanced_equatorial * abs(phi_enh_d) + 1
        phi_cub_d, N_cub = -20, 101
        dphi_e = -phi_enh_d / N_enh
        use4, use6 = False, True
    if refineR == 4 and enhanced_equatorial == 6:
        phi_enh_d = -10
        N_enh = 2 * enhanced_equatorial * abs(phi_enh_d) + 1
        phi_cub_d, N_cub = -20, 101
        dphi_e = -phi_enh_d / N_enh
    jc = np.where(phi_M < phi_cub_d)[0][-1]
    phi1 = phi_M[0:jc]
    phi_e = phi_enh_d
    if use4:
        nodes = [0, 1, N_cub - 2, N_cub - 1]
        vals = [phi_M[jc - 1], phi_M[jc], phi_e - dphi_e, phi_e]
        phi2 = lagrange_interp(nodes, vals, np.arange(N_cub))
    elif use6:
        N_cub = 111
        nodes = [0, 1, 2, N_cub - 3, N_cub - 2, N_cub - 1]
        vals = [phi_M[jc - 1], phi_M[jc], phi_M[jc + 1], phi_e - dphi_e, phi_e, phi_e + dphi_e]
        phi2 = lagrange_interp_6pt(nodes, vals, np.arange(N_cub))
    phi3 = np.concatenate((phi1[0:-1], phi2))
    phi4 = np.linspace(phi3[-1], 0, int(N_enh))
    phi5 = np.concatenate((phi3[0:-1], phi4))
    out = np.concatenate((phi5[0:-1], -phi5[::-1]))
    j_phi_n = np.where(out < phi_n)[0][-1]
    return out[0:j_phi_n]


def generate_mercator_grid(Ni, phi_s, phi_n, lon0_M, lenlon_M, refineR, shift_equator_to_u_point=True,
                           ensure_nj_even=True, enhanced_equatorial=0):
    """x, y of the Mercator sub-grid.  OGG:314-441."""
    phi_M = mercator_axis(Ni, phi_s, phi_n, refineR, shift_equator_to_u_point, ensure_nj_even,
                          enhanced_equatorial)
    lam_M = lon0_M + np.arange(Ni + 1) * lenlon_M / float(Ni)
    if phi_M.shape[0] % 2 == 0 and ensure_nj_even:
        phi_M = phi_M[1:]
    y = np.tile(phi_M.reshape(-1, 1), (1, Ni + 1))
    x = np.tile(lam_M, (phi_M.shape[0], 1))
    return x, y


def generate_latlon_grid(lni, lnj, llon0, llen_lon, llat0, llen_lat, ensure_nj_even=True):
    """Regular lat-lon sub-grid.  OGG:832-846."""
    lon = llon0 + np.arange(lni + 1) * llen_lon / float(lni)
    lat = llat0 + np.arange(lnj + 1) * llen_lat / float(lnj)
    if lat.shape[0] % 2 == 0 and ensure_nj_even:
        lat = lat[1:]
    x = np.tile(lon, (lat.shape[0], 1))
    y = np.tile(lat.reshape(-1, 1), (1, lon.shape[0]))
    return x, y


# ----------------------------------------------------------------------------------------------
# Murray bipolar cap (OGG:33-188)
# ----------------------------------------------------------------------------------------------
def bipolar_projection(lamg, phig, lon_bp, rp, metrics_only=False):
    """Stereographic bipolar projection + inverse scale factors.  OGG:33-100."""
    phig = 90 - 2 * np.arctan(np.tan(0.5 * (90 - phig) * PI_180) / rp) / PI_180
    tmp = mdist(lamg, lon_bp) * PI_180
    sinla = np.sin(tmp)
    sphig = np.sin(phig * PI_180)
    alpha2 = (np.cos(tmp)) ** 2
    beta2_inv = (np.tan(phig * PI_180)) ** 2
    rden = 1.0 / (1.0 + alpha2 * beta2_inv)
    if not metrics_only:
        B = sinla * np.sqrt(rden)
        B = np.where(np.abs(beta2_inv) > HUGE, 0.0, B)
        lamc = np.arcsin(B) / PI_180
        dl = lamg - lon_bp
        lamc = np.where((dl > 90) & (dl <= 180), 180 - lamc, lamc)
        lamc = np.where((dl > 180) & (dl <= 270), 180 + lamc, lamc)
        lamc = np.where((dl > 270), 360 - lamc, lamc)
        lamc = np.where((dl == 90), 90, lamc)
        lamc = np.where((dl == 270), 270, lamc)
        lams = lamc + lon_bp
    A = sinla * sphig
    chic = np.arccos(A)
    phis = 90 - 2 * np.arctan(rp * np.tan(chic / 2)) / PI_180
    rden2 = 1.0 / (1 + (rp * np.tan(chic / 2)) ** 2)
    M_inv = rp * (1 + (np.tan(chic / 2)) ** 2) * rden2
    chig = (90 - phig) * PI_180
    rden2 = 1.0 / (1 + (rp * np.tan(chig / 2)) ** 2)
    N = rp * (1 + (np.tan(chig / 2)) ** 2) * rden2
    N_inv = 1 / N
    cos2phis = (np.cos(phis * PI_180)) ** 2
    h_j_inv = (cos2phis * alpha2 * (1 - alpha2) * beta2_inv * (1 + beta2_inv) * (rden ** 2)
               + M_inv * M_inv * (1 - alpha2) * rden)
    h_j_inv = np.where(np.abs(beta2_inv) > HUGE, M_inv * M_inv, h_j_inv)
    h_j_inv = np.sqrt(h_j_inv) * N_inv
    h_i_inv = (cos2phis * (1 + beta2_inv) * (rden ** 2) + M_inv * M_inv * alpha2 * beta2_inv * rden)
    h_i_inv = np.where(np.abs(beta2_inv) > HUGE, M_inv * M_inv, h_i_inv)
    h_i_inv = np.sqrt(h_i_inv)
    if not metrics_only:
        return lams, phis, h_i_inv, h_j_inv
    return h_i_inv, h_j_inv


def generate_bipolar_cap_mesh(Ni, Nj_ncap, lat0_bp, lon_bp, ensure_nj_even=True):
    """OGG:103-122."""
    if Nj_ncap % 2 != 0 and ensure_nj_even:
        Nj_ncap = Nj_ncap - 1
    lon_g = lon_bp + np.arange(Ni + 1) * 360.0 / float(Ni)
    lamg = np.tile(lon_g, (Nj_ncap + 1, 1))
    latg0 = lat0_bp + np.arange(Nj_ncap + 1) * (90 - lat0_bp) / float(Nj_ncap)
    phig = np.tile(latg0.reshape((Nj_ncap + 1, 1)), (1, Ni + 1))
    rp = np.tan(0.5 * (90 - lat0_bp) * PI_180)
    lams, phis, h_i_inv, h_j_inv = bipolar_projection(lamg, phig, lon_bp, rp)
    h_i_inv = h_i_inv[:, :-1] * 2 * np.pi / float(Ni)
    h_j_inv = h_j_inv[:-1, :] * PI_180 * (90 - lat0_bp) / float(Nj_ncap)
    return lams, phis, h_i_inv, h_j_inv


def bipolar_cap_ij_array(i, j, Ni, Nj_ncap, lat0_bp, lon_bp, rp):
    """Per-index arc lengths (radians) at fractional (i, j).  OGG:125-133."""
    long = lon_bp + i * 360.0 / float(Ni)
    latg = lat0_bp + j * (90 - lat0_bp) / float(Nj_ncap)
    lamg = np.tile(long, (latg.shape[0], 1))
    phig = np.tile(latg.reshape((latg.shape[0], 1)), (1, long.shape[0]))
    h_i_inv, h_j_inv = bipolar_projection(lamg, phig, lon_bp, rp, metrics_only=True)
    h_i_inv = h_i_inv * 2 * np.pi / float(Ni)
    h_j_inv = h_j_inv * (90 - lat0_bp) * PI_180 / float(Nj_ncap)
    return h_i_inv, h_j_inv


def _quad_chunk(dx_r, dy_r, dxq, dyq, daq, c0, c1, per_cell_loop):
    """Lobatto means of one chunk of cell rows.  ``per_cell_loop``: one Python-level call per cell, as the reference does it
    (OGG:176-187 / 585-599) -- the same values, only slower; bench.py uses it for the "reference-shaped" CPU row."""
    dxdy_r = dx_r * dy_r
    if not per_cell_loop:
        daq[c0:c1] = quad_average_2d(dxdy_r.transpose(0, 2, 1, 3))
        dxq[c0:c1] = quad_average(dx_r[:, 0, :, :])
        dyq[c0:c1] = quad_average(dy_r[:, :, :, 0].transpose(0, 2, 1))
        return
    for j in range(c1 - c0):
        for i in range(dx_r.shape[2]):
            daq[c0 + j, i] = quad_average_2d(dxdy_r[j, :, i, :])
            dxq[c0 + j, i] = quad_average(dx_r[j, 0, i, :])
            dyq[c0 + j, i] = quad_average(dy_r[j, :, i, 0])


def bipolar_cap_metrics_quad_fast(order, nx, ny, lat0_bp, lon_bp, rp, Re=RE_DEFAULT, rows_per_chunk=64, j_first=0,
                                  j_last=None, per_cell_loop=False):
    """dx (ny+1,nx), dy (ny,nx+1), area (ny,nx) by Lobatto quadrature of h.  OGG:136-188.
    Evaluated in chunks of cell rows; the per-element arithmetic and the summation order are the
    reference's (the reference's own chunking, OGG:161-172, is likewise semantically irrelevant).  ``j_first`` /
    ``j_last`` restrict the evaluation to cell rows [j_first, j_last) (others stay zero): cells are independent."""
    quad_positions(order)  # raises for uncoded orders
    # OGG:145-147: only the LAST node of a cell is tested against ny (hits cell ny-1 only)
    nodes = _lattice_1d(ny, order).reshape(ny + 1, order)
    nodes[:, -1] = np.where(nodes[:, -1] == ny, ny - 0.001, nodes[:, -1])
    j1d = nodes.reshape(-1)
    i1d = _lattice_1d(nx, order)
    daq = np.zeros([ny + 1, nx + 1])
    dxq = np.zeros([ny + 1, nx + 1])
    dyq = np.zeros([ny + 1, nx + 1])
    j_end = ny + 1 if j_last is None else min(j_last, ny + 1)
    for c0 in range(j_first, j_end, rows_per_chunk):
        c1 = min(j_end, c0 + rows_per_chunk)
        dx, dy = bipolar_cap_ij_array(i1d, j1d[c0 * order:c1 * order], nx, ny, lat0_bp, lon_bp, rp)
        dx_r = dx.reshape(c1 - c0, order, nx + 1, order)
        dy_r = dy.reshape(c1 - c0, order, nx + 1, order)
        _quad_chunk(dx_r, dy_r, dxq, dyq, daq, c0, c1, per_cell_loop)
    return dxq[:, :-1] * Re, dyq[:-1, :] * Re, daq[:-1, :-1] * Re * Re


# ----------------------------------------------------------------------------------------------
# displaced-pole cap (OGG:447-601)
# ----------------------------------------------------------------------------------------------
def monotonic_bounding(x, x_0):
    """Sequential 360-degree unwrap along i.  OGG:470-475.  Vectorised over rows only: the state of
    column i depends on the adjusted column i-1, exactly as in the reference loop."""
    x_im1 = x[:, 0] * 0 + x_0
    for i in range(0, x.shape[1]):
        x[:, i] = np.where(x[:, i] - x_im1[:] > 100, x[:, i] - 360, x[:, i])
        x_im1[:] = x[:, i]
    return x


def displacedPoleCap_projection(lon_grid, lat_grid, z_0, r_joint):
    """OGG:447-467 (numpy complex128 arithmetic, Smith division)."""
    r = np.tan((90 + lat_grid) * PI_180) / r_joint
    e2itheta = np.cos(lon_grid * PI_180) + 1j * np.sin(lon_grid * PI_180)
    e2ithetaprime = (e2itheta - z_0) / (1.0 - np.conj(z_0) * e2itheta)
    z = r * e2ithetaprime
    w = (z + z_0) / (1 + np.conj(z_0) * z)
    lamcDP = np.angle(w, deg=True)
    lamcDP = monotonic_bounding(lamcDP, lon_grid[0, 0])
    rw = np.absolute(w)
    phicDP = -90 + np.arctan(rw * r_joint) / PI_180
    return lamcDP, phicDP


def displacedPoleCap_mesh(i, j, ni, nj, lon0, lat0, lam_pole, r_pole):
    """Mesh at (possibly fractional) index vectors i, j.  OGG:478-506 (du/dv and the never-taken
    excluded_fraction branch not reproduced)."""
    long = lon0 + i * 360.0 / float(ni)
    a = -90.0
    latg = a + j * (lat0 - a) / float(nj)
    lamg = np.tile(long, (latg.shape[0], 1))
    phig = np.tile(latg.reshape((latg.shape[0], 1)), (1, long.shape[0]))
    r_joint = np.tan((90 + lat0) * PI_180)
    z_0 = r_pole * (np.cos(lam_pole * PI_180) + 1j * np.sin(lam_pole * PI_180))
    lams, phis = displacedPoleCap_projection(lamg, phig, z_0, r_joint)
    return lams, phis, lams[0, 0], phis[0, 0]


def generate_displaced_pole_grid(Ni, Nj_scap, lon0, lat0, lon_dp, r_dp):
    """OGG:509-518."""
    return displacedPoleCap_mesh(np.arange(Ni + 1), np.arange(Nj_scap + 1), Ni, Nj_scap, lon0, lat0, lon_dp, r_dp)


def great_arc_distance(j0, i0, j1, i1, nx, ny, lon0, lat0, lon_dp, r_dp):
    """Haversine distance between re-projected nodes.  OGG:522-532."""
    lam0, phi0, _, _ = displacedPoleCap_mesh(i0, j0, nx, ny, lon0, lat0, lon_dp, r_dp)
    lam1, phi1, _, _ = displacedPoleCap_mesh(i1, j1, nx, ny, lon0, lat0, lon_dp, r_dp)
    lam0, phi0 = lam0 * PI_180, phi0 * PI_180
    lam1, phi1 = lam1 * PI_180, phi1 * PI_180
    dphi, dlam = phi1 - phi0, lam1 - lam0
    d = np.sin(0.5 * dphi) ** 2 + np.sin(0.5 * dlam) ** 2 * np.cos(phi0) * np.cos(phi1)
    return 2.0 * np.arcsin(np.sqrt(d))


def _central_difference(ds, order, reps):
    """OGG:539-547 / 554-562: ds is a callable k -> great-arc distance at +-k*eps."""
    ds2 = ds(1.0)
    if order == 2:
        return 0.5 * ds2 * reps
    ds4 = ds(2.0)
    if order == 4:
        return (8.0 * ds2 - ds4) * (1.0 / 12.0) * reps
    ds6 = ds(3.0)
    if order == 6:
        return (45.0 * ds2 - 9.0 * ds4 + ds6) * (1.0 / 60.0) * reps
    raise Exception("order not coded")


def numerical_hi(j, i, nx, ny, lon0, lat0, lon_dp, r_dp, eps, order=6):
    """OGG:535-547.  Note the reference writes i + 2.0*eps etc."""
    def ds(k):  # k*eps: 1.0*eps == eps exactly, 2.0*eps and 3.0*eps as written in the reference
        return great_arc_distance(j, i + k * eps, j, i - k * eps, nx, ny, lon0, lat0, lon_dp, r_dp)
    return _central_difference(ds, order, 1.0 / eps)


def numerical_hj(j, i, nx, ny, lon0, lat0, lon_dp, r_dp, eps, order=6):
    """OGG:550-562."""
    def ds(k):
        return great_arc_distance(j + k * eps, i, j - k * eps, i, nx, ny, lon0, lat0, lon_dp, r_dp)
    return _central_difference(ds, order, 1.0 / eps)


def displacedPoleCap_metrics_quad(order, nx, ny, lon0, lat0, lon_dp, r_dp, Re=RE_DEFAULT, rows_per_chunk=32,
                                  j_first=0, j_last=None, per_cell_loop=False):
    """dx (ny+1,nx), dy (ny,nx+1), area (ny,nx) by quadrature of finite-difference h.  OGG:565-601.
    Cell rows are independent (the unwrap scan runs along i inside a lattice row), so the evaluation is
    chunked by cell rows; ``j_first`` skips cell rows < j_first (left as zeros) -- the rows main() discards
    (OGG:1177-1186)."""
    quad_positions(order)
    j1d = _lattice_1d(ny, order)
    i1d = _lattice_1d(nx, order)
    daq = np.zeros([ny + 1, nx + 1])
    dxq = np.zeros([ny + 1, nx + 1])
    dyq = np.zeros([ny + 1, nx + 1])
    j_end = ny + 1 if j_last is None else min(j_last, ny + 1)
    for c0 in range(j_first, j_end, rows_per_chunk):
        c1 = min(j_end, c0 + rows_per_chunk)
        jj = j1d[c0 * order:c1 * order]
        dx = numerical_hi(jj, i1d, nx, ny, lon0, lat0, lon_dp, r_dp, eps=1e-3, order=order)
        dy = numerical_hj(jj, i1d, nx, ny, lon0, lat0, lon_dp, r_dp, eps=1e-3, order=order)
        dx_r = dx.reshape(c1 - c0, order, nx + 1, order)
        dy_r = dy.reshape(c1 - c0, order, nx + 1, order)
        _quad_chunk(dx_r, dy_r, dxq, dyq, daq, c0, c1, per_cell_loop)
    return dxq[:, :-1] * Re, dyq[:-1, :] * Re, daq[:-1, :-1] * Re * Re


# ----------------------------------------------------------------------------------------------
# whole supergrid: size logic + stitching of main() (OGG:855-1449), without I/O
# ----------------------------------------------------------------------------------------------
def make_supergrid(inverse_resolution, r_dp=0.0, lon_dp=80.0, lat_dp=-99.0, exfracdp=0.49,
                   south_cutoff_row=0, south_cutoff_ang=-90.0, enhanced_equatorial=0, grids="all",
                   match_dy=(), skip_metrics=False, ensure_nj_even=False, shift_equator_to_u_point=True,
                   bipolar_lower_lat=-90.0, mercator_lower_lat=-90.0, mercator_upper_lat=-99.0,
                   south_ocean_lower_lat=-90.0, no_south_cap=False, skip_doughnut_rows=False):
    """Oracle restatement of main()'s numerical content.  Returns a dict with the stitched
    ``x y dx dy area angle_dx`` and the per-sub-grid pieces under ``sub``.  Defaults are the argparse
    defaults (OGG:1456-1524), not main()'s own.  ``skip_doughnut_rows`` evaluates
    the displaced-pole quadrature only for the rows main() keeps (same values for those rows)."""
    doughnut = exfracdp if (exfracdp is not None) else 0.28 * 7 / 4          # OGG:891-892
    calculate_metrics = not skip_metrics
    if r_dp != 0.0 and lat_dp > -90.0:                                          # OGG:904-907
        raise SystemExit(2)
    refineS, refineR = 2, inverse_resolution                                   # OGG:969-974
    lenlon, lon0 = 360, -300.0
    Ni = int(refineR * refineS * lenlon)
    phi_s_Merc, phi_n_Merc = -66.85954725, 64.05895973                         # OGG:987-1001
    if mercator_upper_lat > -90:
        phi_n_Merc = mercator_upper_lat
    if mercator_lower_lat > -90:
        phi_s_Merc = mercator_lower_lat
    if refineR == 2:
        phi_s_Merc, phi_n_Merc = -68.0, 65.0
    if refineR == 1 and enhanced_equatorial:
        phi_s_Merc, phi_n_Merc = -77.8, 60.0

    def want(tok):
        return (tok in grids) or ("all" in grids)

    def metrics_or_minus_one(lam, fn):
        nj1, ni1 = lam.shape
        if calculate_metrics:
            return fn()
        return -np.ones([nj1, ni1 - 1]), -np.ones([nj1 - 1, ni1]), -np.ones([nj1 - 1, ni1 - 1])

    sub = {}
    q = Ni // 4
    if want("mercator"):                                                       # OGG:1003-1030
        lamM, phiM = generate_mercator_grid(Ni, phi_s_Merc, phi_n_Merc, lon0, lenlon, refineR,
                                            shift_equator_to_u_point=shift_equator_to_u_point,
                                            ensure_nj_even=ensure_nj_even,
                                            enhanced_equatorial=enhanced_equatorial)
        angM = angle_x(lamM, phiM)
        dxM, dyM, arM = metrics_or_minus_one(lamM, lambda: generate_grid_metrics_MIDAS(lamM, phiM))
        sub["Merc"] = [lamM, phiM, dxM, dyM, arM, angM]
        dphi_so = phiM[1, q] - phiM[0, q]
        dphi_no = phiM[-1, q] - phiM[-2, q]
        lat0_bp = phiM[-1, q]
    if want("bipolar"):                                                        # OGG:1035-1071
        lon_bp = lon0
        if bipolar_lower_lat > -90:
            lat0_bp = bipolar_lower_lat
        Nj_ncap = int(60 * refineR * refineS)
        if refineR == 2:
            Nj_ncap = 119 * refineS
        if refineR == 1 and enhanced_equatorial:
            Nj_ncap = 154
        if "bp" in match_dy:
            Nj_ncap = int(0.5 + (90.0 - lat0_bp) / dphi_no)
        lamB, phiB, _, _ = generate_bipolar_cap_mesh(Ni, Nj_ncap, lat0_bp, lon_bp, ensure_nj_even=ensure_nj_even)
        rp = np.tan(0.5 * (90 - lat0_bp) * PI_180)
        dxB, dyB, arB = metrics_or_minus_one(
            lamB, lambda: bipolar_cap_metrics_quad_fast(5, phiB.shape[1] - 1, phiB.shape[0] - 1, lat0_bp, lon_bp, rp))
        angB = angle_x(lamB, phiB)
        sub["BP"] = [lamB, phiB, dxB, dyB, arB, angB]
    lat0_SO = -78.0                                                            # OGG:1080-1097
    if south_ocean_lower_lat > -90:
        lat0_SO = south_ocean_lower_lat
    latUp_SO = phiM[0, q]
    lenlat_SO = latUp_SO - lat0_SO
    deltaPhiSO = 1.0 / refineR / refineS
    Nj_SO = int(refineR * 55)
    if refineR == 2 and enhanced_equatorial:
        Nj_SO = 109
    if refineR == 1 and enhanced_equatorial:
        Nj_SO = 0
    if "so" in match_dy:
        Nj_SO = int(0.5 + lenlat_SO / dphi_so)
    if (Nj_SO != 0) and want("so"):                                            # OGG:1100-1110
        lamS, phiS = generate_latlon_grid(Ni, Nj_SO, lon0, lenlon, lat0_SO, lenlat_SO, ensure_nj_even=ensure_nj_even)
        dxS, dyS, arS = metrics_or_minus_one(lamS, lambda: generate_grid_metrics_MIDAS(lamS, phiS))
        angS = angle_x(lamS, phiS)
        sub["SO"] = [lamS, phiS, dxS, dyS, arS, angS]
    lat0_SC = phiS[0, q]                                                       # OGG:1122-1138
    if "p125sc" in match_dy:
        lat0_SC = lat0_SO
    Nj_scap = int(refineR * 40) * 7 // 4
    if no_south_cap or (enhanced_equatorial and refineR in (1, 2)):
        Nj_scap = 0
    if (Nj_scap != 0) and want("sc"):                                          # OGG:1140-1197
        if r_dp == 0.0 and lat_dp <= -90.0:
            Nj_scap = int((lat0_SC + 90.0) / deltaPhiSO)
            lamC, phiC = generate_latlon_grid(Ni, Nj_scap, lon0, lenlon, -90.0, 90 + lat0_SO,
                                              ensure_nj_even=ensure_nj_even)
            angC = angle_x(lamC, phiC)
            dxC, dyC, arC = metrics_or_minus_one(lamC, lambda: generate_grid_metrics_MIDAS(lamC, phiC))
        else:
            if lat_dp > -90:
                r_dp = np.tan((90 + lat_dp) * PI_180) / np.tan((90 + lat0_SC) * PI_180)
            lamC, phiC, _, _ = generate_displaced_pole_grid(Ni, Nj_scap, lon0, lat0_SC, lon_dp, r_dp)
            angC = angle_x(lamC, phiC)
            jmint = 0
            if doughnut != 0.0:
                jmin = np.ceil(doughnut * Nj_scap)
                jmint = int(jmin + np.mod(jmin, 2))
            dxC, dyC, arC = metrics_or_minus_one(
                lamC, lambda: displacedPoleCap_metrics_quad(4, Ni, Nj_scap, lon0, lat0_SC, lon_dp, r_dp,
                                                            j_first=jmint if skip_doughnut_rows else 0))
            pieces = [lamC, phiC, dxC, dyC, arC, angC]
            pieces = [p[jmint:, :] for p in pieces]
            if pieces[1].shape[0] % 2 == 0 and ensure_nj_even:
                pieces = [p[1:, :] for p in pieces]
            lamC, phiC, dxC, dyC, arC, angC = pieces
        sub["SC"] = [lamC, phiC, dxC, dyC, arC, angC]

    # south cuts (OGG:1268-1313)
    cut, jcut = False, 0
    if south_cutoff_row > 0:
        cut, jcut = True, south_cutoff_row - 1
    elif south_cutoff_ang > -90:
        cut, jcut = True, 1 + np.nonzero(sub["SC"][1][:, 0] < south_cutoff_ang)[0][-1]
    if cut:
        if "SC" in sub and jcut < sub["SC"][0].shape[0]:
            if (sub["SC"][1].shape[0] - jcut) % 2 == 0 and ensure_nj_even:
                jcut = jcut + 1
            sub["SC"] = [p[jcut:, :] for p in sub["SC"]]
        elif "SO" in sub:
            n_sc = sub["SC"][0].shape[0]   # like the reference, needs an SC piece to have existed
            del sub["SC"]
            jcut_SO = jcut - n_sc
            if (sub["SO"][4].shape[0] - jcut_SO - 1) % 2 == 0 and ensure_nj_even:
                jcut_SO = jcut_SO + 1
            sub["SO"] = [p[jcut_SO:, :] for p in sub["SO"]]

    # stitching south -> north (OGG:1315-1377): drop the last row of the southern piece for
    # x, y, dx, angle; concatenate dy, area unchanged
    def join(south, north):
        x, y, dx, dy, ar, an = south
        X, Y, DX, DY, AR, AN = north
        return [np.concatenate((x[:-1, :], X), axis=0), np.concatenate((y[:-1, :], Y), axis=0),
                np.concatenate((dx[:-1, :], DX), axis=0), np.concatenate((dy, DY), axis=0),
                np.concatenate((ar, AR), axis=0), np.concatenate((an[:-1, :], AN), axis=0)]

    g = None
    if "SC" in sub and "SO" in sub:
        g = join(sub["SC"], sub["SO"])
    elif "SO" in sub:
        g = list(sub["SO"])
    if "SO" in sub and "Merc" in sub:
        g = join(g, sub["Merc"])
    elif "Merc" in sub:
        g = list(sub["Merc"])
    if "BP" in sub:
        g = join(g, sub["BP"])
        ycol = g[1][:, q]
        if np.any((np.roll(ycol, shift=-1, axis=0) - ycol) == 0):
            raise Exception("lattitude array has repeated values along symmetry meridian!")
    x3, y3, dx3, dy3, area3, angle3 = g
    eq = np.searchsorted(y3[:, q], 0.0)                                        # OGG:1425-1436
    if eq == 0:
        raise Exception("   Ooops: Equator is not in the grid")
    if eq % 2 == 0:
        raise Exception("Ooops: Equator is not going to be a u-point. Use option --south_cutoff_row to one more or on less row from south.")
    if y3.shape[0] % 2 == 0:
        raise Exception("Ooops: The number of j's in the supergrid is not even. Use option --south_cutoff_row to one more or on less row from south.")
    return {"x": x3, "y": y3, "dx": dx3, "dy": dy3, "area": area3, "angle_dx": angle3, "sub": sub, "Ni": Ni}
