// K3 / K4: Murray bipolar Arctic cap -- C-ABI entry points; the kernels are in ogg_bipolar_dev.h.
#include "ogg_bipolar_dev.h"

namespace ogg {
QuadNodes quad_nodes_host(int order) { return make_nodes(order); }
}  // namespace ogg

namespace {
__global__ void libm_check_kernel(int which, long n, const double* __restrict__ x, const double* __restrict__ y, unsigned long long* n_diff) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    bool diff = false;
    if (k < n) {
        double a = 0.0, b = 0.0;
        if (which == 0)
            a = asin_unit(x[k]), b = asin(x[k]);
        else if (which == 1)
            a = atan_lib(x[k]), b = atan(x[k]);
        else if (which == 2)
            a = atan2_lib(y[k], x[k]), b = atan2(y[k], x[k]);
        else if (which == 3)
            a = rcp_ieee_normal(x[k]), b = 1.0 / x[k];
        else if (which == 4)
            a = sqrt_ieee_normal(x[k]), b = sqrt(x[k]);
        else if (which == 5)
            a = div_ieee_normal(y[k], x[k]), b = y[k] / x[k];
        else if (which == 6 || which == 7) {   // the restatements with their coefficients in vector registers (the literal displaced-pole quadrature)
            AtanVgpr c;
            c.load(kAtanRed);
            if (which == 6)
                a = atan_lib_wave(x[k], c), b = atan(x[k]);
            else
                a = atan2_lib(y[k], x[k], c), b = atan2(y[k], x[k]);
            c.keep();
        } else if (which == 9) {              // atan2 for any arguments (infinities, NaNs)
            a = atan2_lib_any(y[k], x[k]), b = atan2(y[k], x[k]);
            if (a != a && b != b) b = a;      // any NaN is the same answer
        } else if (which == 13 || which == 14) {   // mdist from one reduction / from two fma reductions, against numpy.mod's own fmod form
            a = (which == 13) ? mdist_one(x[k], y[k]) : mdist(x[k], y[k]);
            double m1 = fmod(x[k] - y[k], 360.0), m2 = fmod(y[k] - x[k], 360.0);
            m1 = (m1 < 0.0) ? m1 + 360.0 : m1, m2 = (m2 < 0.0) ? m2 + 360.0 : m2;
            m1 = (m1 == 0.0) ? 0.0 : m1, m2 = (m2 == 0.0) ? 0.0 : m2;
            b = fmin(m1, m2);
            if (a != a && b != b) b = a;
        }
        diff = __double_as_longlong(a) != __double_as_longlong(b);
    }
    const unsigned long long m = __ballot(diff);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(n_diff, (unsigned long long)__popcll(m));
}
}  // namespace

extern "C" {

int ogg_libm_check_dev(int which, long n, const double* x, const double* y, unsigned long long* n_diff, void* stream) {
    OGG_REQUIRE(which >= 0 && which <= 14 && which != 8 && !(which >= 10 && which <= 12) && n >= 0 && x && n_diff &&
                    ((which != 2 && which != 5 && which != 7 && which != 9 && which != 13 && which != 14) || y), OGG_EARG,
                "ogg_libm_check: bad argument");
    if (n == 0) return OGG_OK;
    libm_check_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ogg::as_stream(stream)>>>(which, n, x, y, n_diff);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}

int ogg_bipolar_projection_dev(long n, const double* lamg, const double* phig, double lon_bp, double rp, int metrics_only,
                               double* lams, double* phis, double* h_i_inv, double* h_j_inv, void* stream) {
    OGG_REQUIRE(n >= 0 && lamg && phig, OGG_EARG, "ogg_bipolar_projection: bad argument");
    if (n == 0) return OGG_OK;
    if (metrics_only) lams = phis = nullptr;
    bipolar_projection_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ogg::as_stream(stream)>>>(n, lamg, phig, lon_bp, rp, lams,
                                                                                              phis, h_i_inv, h_j_inv);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}

int ogg_bipolar_cap_mesh_angle_sym_dev(long Ni, long Nj, double lat0_bp, double lon_bp, long j0, long nrows, int symmetry, double* lams,
                                       double* phis, double* h_i_inv, double* h_j_inv, double* angle_dx, void* stream) {
    OGG_REQUIRE(Ni > 0 && Nj > 0 && j0 >= 0 && nrows >= 0 && j0 + nrows <= Nj + 1, OGG_ESHAPE,
                "ogg_bipolar_cap_mesh: rows %ld..%ld outside 0..%ld", j0, j0 + nrows, Nj);
    OGG_REQUIRE(lams && phis, OGG_EARG, "ogg_bipolar_cap_mesh: null output");
    if (nrows == 0) return OGG_OK;
    MeshParams m{Ni, Nj, lat0_bp, lon_bp, j0, nrows, lams, phis, h_i_inv, h_j_inv, angle_dx, MESH_ROWS, {}};
    const dim3 grid = mesh_grid(m, ogg::cap_symmetry(symmetry) ? 1 : 0);
    if (h_i_inv || h_j_inv)
        bipolar_mesh_kernel<true><<<grid, 64 * MESH_WAVES, 0, ogg::as_stream(stream)>>>(m);
    else
        bipolar_mesh_kernel<false><<<grid, 64 * MESH_WAVES, 0, ogg::as_stream(stream)>>>(m);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}

int ogg_bipolar_cap_mesh_angle_dev(long Ni, long Nj, double lat0_bp, double lon_bp, long j0, long nrows, double* lams,
                                   double* phis, double* h_i_inv, double* h_j_inv, double* angle_dx, void* stream) {
    return ogg_bipolar_cap_mesh_angle_sym_dev(Ni, Nj, lat0_bp, lon_bp, j0, nrows, OGG_SYM_DEFAULT, lams, phis, h_i_inv, h_j_inv, angle_dx, stream);
}

int ogg_bipolar_cap_mesh_dev(long Ni, long Nj, double lat0_bp, double lon_bp, long j0, long nrows, double* lams,
                             double* phis, double* h_i_inv, double* h_j_inv, void* stream) {
    return ogg_bipolar_cap_mesh_angle_dev(Ni, Nj, lat0_bp, lon_bp, j0, nrows, lams, phis, h_i_inv, h_j_inv, nullptr, stream);
}

int ogg_bipolar_cap_ij_array_dev(long n_i, const double* i, long n_j, const double* j, long Ni, long Nj, double lat0_bp,
                                 double lon_bp, double rp, double* h_i_inv, double* h_j_inv, void* stream) {
    OGG_REQUIRE(n_i > 0 && n_j >= 0 && i && j && h_i_inv && h_j_inv && Ni > 0 && Nj > 0, OGG_EARG,
                "ogg_bipolar_cap_ij_array: bad argument");
    if (n_j == 0) return OGG_OK;
    dim3 grid((unsigned)((n_i + 255) / 256), (unsigned)n_j);
    bipolar_ij_kernel<<<grid, 256, 0, ogg::as_stream(stream)>>>(n_i, i, n_j, j, Ni, Nj, lat0_bp, lon_bp, rp, h_i_inv, h_j_inv);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}

long ogg_bipolar_quad_workspace_bytes(int order, long nx, long ny) {
    if (order < 2 || order > 5 || nx <= 0 || ny <= 0) return 0;
    switch (order) {
        case 2: return (long)quad_workspace_bytes<2>(nx, ny, ny);
        case 3: return (long)quad_workspace_bytes<3>(nx, ny, ny);
        case 4: return (long)quad_workspace_bytes<4>(nx, ny, ny);
        default: return (long)quad_workspace_bytes<5>(nx, ny, ny);
    }
}

int ogg_bipolar_cap_metrics_quad_sym_ws_dev(int order, long nx, long ny, double lat0_bp, double lon_bp, double rp, double Re,
                                            long j0, long n_dx_rows, long n_cell_rows, int symmetry, double* dxq, double* dyq, double* daq,
                                            void* workspace, long workspace_bytes, void* stream) {
    OGG_REQUIRE(order >= 2 && order <= 5, OGG_EORDER, "Uncoded order");
    OGG_REQUIRE(nx > 0 && ny > 0 && dxq && (n_cell_rows <= 0 || (dyq && daq)), OGG_EARG, "ogg_bipolar_cap_metrics_quad: bad argument");
    OGG_REQUIRE(j0 >= 0 && n_cell_rows >= 0 && j0 + n_cell_rows <= ny &&
                    (n_dx_rows == n_cell_rows || (n_dx_rows == n_cell_rows + 1 && j0 + n_cell_rows == ny)),
                OGG_ESHAPE, "ogg_bipolar_cap_metrics_quad: band j0=%ld cell rows=%ld dx rows=%ld of ny=%ld", j0, n_cell_rows,
                n_dx_rows, ny);
    // OGG_BP_GUARD_K: threshold of the exactness guard (see bp_point_fast).  Default 4000; 0 hands every cell to the
    // literal fix-up (slow; used by the tests to compare the two paths).
    double gap = BP_GUARD_K_DEFAULT;
    if (const char* e = getenv("OGG_BP_GUARD_K")) gap = atof(e);
    QuadParams p{};
    p.nx = nx, p.ny = ny, p.lat0_bp = lat0_bp, p.lon_bp = lon_bp, p.rp = rp, p.Re = Re, p.j0 = j0;
    p.dxq = dxq, p.dyq = dyq, p.daq = daq, p.q = make_nodes(order);
    hipStream_t s = ogg::as_stream(stream);
    const int sym = ogg::cap_symmetry(symmetry) ? 1 : 0;
    switch (order) {
        case 2: return launch_quad<2>(p, n_dx_rows, n_cell_rows, gap, sym, workspace, workspace_bytes, s);
        case 3: return launch_quad<3>(p, n_dx_rows, n_cell_rows, gap, sym, workspace, workspace_bytes, s);
        case 4: return launch_quad<4>(p, n_dx_rows, n_cell_rows, gap, sym, workspace, workspace_bytes, s);
        default: return launch_quad<5>(p, n_dx_rows, n_cell_rows, gap, sym, workspace, workspace_bytes, s);
    }
}

int ogg_bipolar_cap_metrics_quad_ws_dev(int order, long nx, long ny, double lat0_bp, double lon_bp, double rp, double Re,
                                        long j0, long n_dx_rows, long n_cell_rows, double* dxq, double* dyq, double* daq,
                                        void* workspace, long workspace_bytes, void* stream) {
    return ogg_bipolar_cap_metrics_quad_sym_ws_dev(order, nx, ny, lat0_bp, lon_bp, rp, Re, j0, n_dx_rows, n_cell_rows, OGG_SYM_DEFAULT, dxq,
                                                   dyq, daq, workspace, workspace_bytes, stream);
}

int ogg_bipolar_cap_metrics_quad_dev(int order, long nx, long ny, double lat0_bp, double lon_bp, double rp, double Re,
                                     long j0, long n_dx_rows, long n_cell_rows, double* dxq, double* dyq, double* daq,
                                     void* stream) {
    return ogg_bipolar_cap_metrics_quad_ws_dev(order, nx, ny, lat0_bp, lon_bp, rp, Re, j0, n_dx_rows, n_cell_rows, dxq, dyq, daq,
                                               nullptr, 0, stream);
}

}  // extern "C"
