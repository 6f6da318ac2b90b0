// Host-pointer entry points of the C ABI (drop-in layer: numpy arrays in, numpy arrays out) and small utilities.
// Each function stages its inputs into device memory, runs the *_dev kernels on the default stream, copies the
// results back and frees its scratch.  There is no CPU compute path: without a usable HIP device every function
// returns OGG_EHIP.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ogg_common.h"

namespace ogg {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// scratch device buffers of one call, freed on scope exit
class DevScratch {
   public:
    ~DevScratch() {
        for (void* p : ptrs_) (void)hipFree(p);
    }
    int alloc(double** out, long n) {
        void* p = nullptr;
        const size_t bytes = (size_t)(n > 0 ? n : 1) * sizeof(double);
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            const int code = (e == hipErrorOutOfMemory) ? OGG_ENOMEM : OGG_EHIP;
            return set_error(code, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        }
        ptrs_.push_back(p);
        *out = static_cast<double*>(p);
        return OGG_OK;
    }
    int upload(double** out, const double* host, long n) {
        if (int e = alloc(out, n)) return e;
        if (n > 0) OGG_HIP_CHECK(hipMemcpy(*out, host, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
        return OGG_OK;
    }

   private:
    std::vector<void*> ptrs_;
};

static int download(double* host, const double* dev, long n) {
    if (n > 0 && host) OGG_HIP_CHECK(hipMemcpy(host, dev, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    return OGG_OK;
}

}  // namespace ogg

using ogg::DevScratch;
using ogg::download;

#define OGG_TRY(expr)          \
    do {                       \
        int e__ = (expr);      \
        if (e__) return e__;   \
    } while (0)

extern "C" {

const char* ogg_last_error(void) { return ogg::g_err; }
const char* ogg_version(void) { return "ogg_hip 0.1 (gfx950)"; }
// sizeof of the descriptor structs of the ABI (0: ogg_latlon_band, 1: ogg_bipolar_band), so that a binding can check its layout
long ogg_abi_sizeof(int which) {
    return which == 0 ? (long)sizeof(ogg_latlon_band)
                      : (which == 1 ? (long)sizeof(ogg_bipolar_band) : (which == 2 ? (long)sizeof(ogg_dpole_band) : -1L));
}

int ogg_device_count(int* count) {
    OGG_REQUIRE(count, OGG_EARG, "ogg_device_count: null pointer");
    *count = 0;
    OGG_HIP_CHECK(hipGetDeviceCount(count));
    return OGG_OK;
}

int ogg_set_device(int device) {
    OGG_HIP_CHECK(hipSetDevice(device));
    return OGG_OK;
}

int ogg_device_name(char* buf, int buflen) {
    OGG_REQUIRE(buf && buflen > 0, OGG_EARG, "ogg_device_name: bad buffer");
    int dev = 0;
    OGG_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    OGG_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    snprintf(buf, (size_t)buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return OGG_OK;
}

int ogg_event_create(void** ev) {
    OGG_REQUIRE(ev, OGG_EARG, "ogg_event_create: null pointer");
    hipEvent_t e;
    OGG_HIP_CHECK(hipEventCreate(&e));
    *ev = e;
    return OGG_OK;
}
int ogg_event_destroy(void* ev) {
    OGG_HIP_CHECK(hipEventDestroy(static_cast<hipEvent_t>(ev)));
    return OGG_OK;
}
int ogg_event_record(void* ev, void* stream) {
    OGG_HIP_CHECK(hipEventRecord(static_cast<hipEvent_t>(ev), ogg::as_stream(stream)));
    return OGG_OK;
}
int ogg_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms) {
    OGG_REQUIRE(ms, OGG_EARG, "ogg_event_elapsed_ms: null pointer");
    OGG_HIP_CHECK(hipEventSynchronize(static_cast<hipEvent_t>(ev_stop)));
    OGG_HIP_CHECK(hipEventElapsedTime(ms, static_cast<hipEvent_t>(ev_start), static_cast<hipEvent_t>(ev_stop)));
    return OGG_OK;
}
int ogg_stream_synchronize(void* stream) {
    OGG_HIP_CHECK(hipStreamSynchronize(ogg::as_stream(stream)));
    return OGG_OK;
}

// ---- Mercator / lat-lon ------------------------------------------------------------------------------------
int ogg_y_mercator_rounded(long Ni, long n, const double* phi_rad, long long* ystar) {
    OGG_REQUIRE(n >= 0 && phi_rad && ystar, OGG_EARG, "ogg_y_mercator_rounded: bad argument");
    DevScratch s;
    double *d_phi, *d_y;
    OGG_TRY(s.upload(&d_phi, phi_rad, n));
    OGG_TRY(s.alloc(&d_y, n));
    OGG_TRY(ogg_y_mercator_rounded_dev(Ni, n, d_phi, reinterpret_cast<long long*>(d_y), nullptr));
    if (n > 0) OGG_HIP_CHECK(hipMemcpy(ystar, d_y, (size_t)n * sizeof(long long), hipMemcpyDeviceToHost));
    return OGG_OK;
}

int ogg_phi_mercator(long Ni, long n, const double* y, double* phi_deg) {
    OGG_REQUIRE(n >= 0 && y && phi_deg, OGG_EARG, "ogg_phi_mercator: bad argument");
    DevScratch s;
    double *d_y, *d_phi;
    OGG_TRY(s.upload(&d_y, y, n));
    OGG_TRY(s.alloc(&d_phi, n));
    OGG_TRY(ogg_phi_mercator_dev(Ni, n, d_y, d_phi, nullptr));
    return download(phi_deg, d_phi, n);
}

int ogg_tile_latlon(long nrows, long ni1, const double* lat1d, const double* lon1d, double* x, double* y) {
    OGG_REQUIRE(nrows >= 0 && ni1 > 0 && lat1d && lon1d && x && y, OGG_EARG, "ogg_tile_latlon: bad argument");
    DevScratch s;
    double *d_lat, *d_lon, *d_x, *d_y;
    OGG_TRY(s.upload(&d_lat, lat1d, nrows));
    OGG_TRY(s.upload(&d_lon, lon1d, ni1));
    OGG_TRY(s.alloc(&d_x, nrows * ni1));
    OGG_TRY(s.alloc(&d_y, nrows * ni1));
    OGG_TRY(ogg_tile_latlon_dev(nrows, ni1, d_lat, d_lon, d_x, d_y, nullptr));
    OGG_TRY(download(x, d_x, nrows * ni1));
    return download(y, d_y, nrows * ni1);
}

int ogg_generate_latlon_grid(long lni, long lnj, double llon0, double llen_lon, double llat0, double llen_lat,
                             int skip_first_row, double* x, double* y) {
    OGG_REQUIRE(lni > 0 && lnj > 0 && x && y, OGG_EARG, "ogg_generate_latlon_grid: bad argument");
    const long skip = skip_first_row ? 1 : 0;
    const long nrows = lnj + 1 - skip, ni1 = lni + 1;
    DevScratch s;
    double *d_lat, *d_lon, *d_x, *d_y;
    OGG_TRY(s.alloc(&d_lat, lnj + 1));
    OGG_TRY(s.alloc(&d_lon, ni1));
    OGG_TRY(s.alloc(&d_x, nrows * ni1));
    OGG_TRY(s.alloc(&d_y, nrows * ni1));
    OGG_TRY(ogg_linear_axis_dev(ni1, llon0, llen_lon, (double)lni, d_lon, nullptr));      // OGG:834
    OGG_TRY(ogg_linear_axis_dev(lnj + 1, llat0, llen_lat, (double)lnj, d_lat, nullptr));  // OGG:835
    OGG_TRY(ogg_tile_latlon_dev(nrows, ni1, d_lat + skip, d_lon, d_x, d_y, nullptr));
    OGG_TRY(download(x, d_x, nrows * ni1));
    return download(y, d_y, nrows * ni1);
}

// ---- MIDAS + angle -----------------------------------------------------------------------------------------
int ogg_grid_metrics_midas(long nj1, long ni1, const double* x, const double* y, double Re, int latlon_areafix, double* dx,
                           double* dy, double* area) {
    OGG_REQUIRE(x && y && dx && dy && area, OGG_EARG, "ogg_grid_metrics_midas: null pointer");
    OGG_REQUIRE(nj1 >= 2 && ni1 >= 2, OGG_ESHAPE, "ogg_grid_metrics_midas: need at least 2x2 points, got %ld x %ld", nj1, ni1);
    DevScratch s;
    double *d_x, *d_y, *d_dx, *d_dy, *d_ar;
    OGG_TRY(s.upload(&d_x, x, nj1 * ni1));
    OGG_TRY(s.upload(&d_y, y, nj1 * ni1));
    OGG_TRY(s.alloc(&d_dx, nj1 * (ni1 - 1)));
    OGG_TRY(s.alloc(&d_dy, (nj1 - 1) * ni1));
    OGG_TRY(s.alloc(&d_ar, (nj1 - 1) * (ni1 - 1)));
    OGG_TRY(ogg_grid_metrics_midas_dev(nj1, ni1, d_x, d_y, nj1, nj1 - 1, Re, latlon_areafix, d_dx, d_dy, d_ar, nullptr, nullptr));
    OGG_TRY(download(dx, d_dx, nj1 * (ni1 - 1)));
    OGG_TRY(download(dy, d_dy, (nj1 - 1) * ni1));
    return download(area, d_ar, (nj1 - 1) * (ni1 - 1));
}

int ogg_angle_x(long nj1, long ni1, const double* x, const double* y, double* angle_dx) {
    OGG_REQUIRE(x && y && angle_dx, OGG_EARG, "ogg_angle_x: null pointer");
    OGG_REQUIRE(nj1 >= 1 && ni1 >= 2, OGG_ESHAPE, "Input arrays do not have the same shape!");
    DevScratch s;
    double *d_x, *d_y, *d_a;
    OGG_TRY(s.upload(&d_x, x, nj1 * ni1));
    OGG_TRY(s.upload(&d_y, y, nj1 * ni1));
    OGG_TRY(s.alloc(&d_a, nj1 * ni1));
    OGG_TRY(ogg_grid_metrics_midas_dev(nj1, ni1, d_x, d_y, nj1, 0, 6371.0e3, 1, nullptr, nullptr, nullptr, d_a, nullptr));
    return download(angle_dx, d_a, nj1 * ni1);
}

// ---- bipolar cap -------------------------------------------------------------------------------------------
int ogg_bipolar_projection(long n, const double* lamg, const double* phig, double lon_bp, double rp, int metrics_only,
                           double* lams, double* phis, double* h_i_inv, double* h_j_inv) {
    OGG_REQUIRE(n >= 0 && lamg && phig && h_i_inv && h_j_inv, OGG_EARG, "ogg_bipolar_projection: bad argument");
    OGG_REQUIRE(metrics_only || (lams && phis), OGG_EARG, "ogg_bipolar_projection: lams/phis required");
    DevScratch s;
    double *d_l, *d_p, *d_ls, *d_ps, *d_hi, *d_hj;
    OGG_TRY(s.upload(&d_l, lamg, n));
    OGG_TRY(s.upload(&d_p, phig, n));
    OGG_TRY(s.alloc(&d_ls, n));
    OGG_TRY(s.alloc(&d_ps, n));
    OGG_TRY(s.alloc(&d_hi, n));
    OGG_TRY(s.alloc(&d_hj, n));
    OGG_TRY(ogg_bipolar_projection_dev(n, d_l, d_p, lon_bp, rp, metrics_only, d_ls, d_ps, d_hi, d_hj, nullptr));
    if (!metrics_only) {
        OGG_TRY(download(lams, d_ls, n));
        OGG_TRY(download(phis, d_ps, n));
    }
    OGG_TRY(download(h_i_inv, d_hi, n));
    return download(h_j_inv, d_hj, n);
}

int ogg_bipolar_cap_mesh(long Ni, long Nj, double lat0_bp, double lon_bp, double* lams, double* phis, double* h_i_inv,
                         double* h_j_inv) {
    OGG_REQUIRE(Ni > 0 && Nj > 0 && lams && phis, OGG_EARG, "ogg_bipolar_cap_mesh: bad argument");
    const long n = (Nj + 1) * (Ni + 1);
    DevScratch s;
    double *d_ls, *d_ps, *d_hi = nullptr, *d_hj = nullptr;
    OGG_TRY(s.alloc(&d_ls, n));
    OGG_TRY(s.alloc(&d_ps, n));
    if (h_i_inv) OGG_TRY(s.alloc(&d_hi, (Nj + 1) * Ni));
    if (h_j_inv) OGG_TRY(s.alloc(&d_hj, Nj * (Ni + 1)));
    OGG_TRY(ogg_bipolar_cap_mesh_dev(Ni, Nj, lat0_bp, lon_bp, 0, Nj + 1, d_ls, d_ps, d_hi, d_hj, nullptr));
    OGG_TRY(download(lams, d_ls, n));
    OGG_TRY(download(phis, d_ps, n));
    if (h_i_inv) OGG_TRY(download(h_i_inv, d_hi, (Nj + 1) * Ni));
    if (h_j_inv) OGG_TRY(download(h_j_inv, d_hj, Nj * (Ni + 1)));
    return OGG_OK;
}

int ogg_bipolar_cap_metrics_quad(int order, long nx, long ny, double lat0_bp, double lon_bp, double rp, double Re, double* dxq,
                                 double* dyq, double* daq) {
    OGG_REQUIRE(order >= 2 && order <= 5, OGG_EORDER, "Uncoded order");
    OGG_REQUIRE(nx > 0 && ny > 0 && dxq && dyq && daq, OGG_EARG, "ogg_bipolar_cap_metrics_quad: bad argument");
    DevScratch s;
    double *d_dx, *d_dy, *d_da;
    OGG_TRY(s.alloc(&d_dx, (ny + 1) * nx));
    OGG_TRY(s.alloc(&d_dy, ny * (nx + 1)));
    OGG_TRY(s.alloc(&d_da, ny * nx));
    OGG_TRY(ogg_bipolar_cap_metrics_quad_dev(order, nx, ny, lat0_bp, lon_bp, rp, Re, 0, ny + 1, ny, d_dx, d_dy, d_da, nullptr));
    OGG_TRY(download(dxq, d_dx, (ny + 1) * nx));
    OGG_TRY(download(dyq, d_dy, ny * (nx + 1)));
    return download(daq, d_da, ny * nx);
}

// ---- displaced pole cap ------------------------------------------------------------------------------------
int ogg_displaced_pole_mesh(long n_i, const double* i, long n_j, const double* j, long ni, long nj, double lon0, double lat0,
                            double lam_pole, double r_pole, double* lams, double* phis) {
    OGG_REQUIRE(n_i > 0 && n_j >= 0 && i && j && lams && phis, OGG_EARG, "ogg_displaced_pole_mesh: bad argument");
    DevScratch s;
    double *d_i, *d_j, *d_l, *d_p;
    OGG_TRY(s.upload(&d_i, i, n_i));
    OGG_TRY(s.upload(&d_j, j, n_j));
    OGG_TRY(s.alloc(&d_l, n_i * n_j));
    OGG_TRY(s.alloc(&d_p, n_i * n_j));
    OGG_TRY(ogg_displaced_pole_mesh_dev(n_i, d_i, n_j, d_j, ni, nj, lon0, lat0, lam_pole, r_pole, d_l, d_p, nullptr));
    OGG_TRY(download(lams, d_l, n_i * n_j));
    return download(phis, d_p, n_i * n_j);
}

int ogg_displaced_pole_numerical_h(long n_i, const double* i, long n_j, const double* j, long nx, long ny, double lon0,
                                   double lat0, double lon_dp, double r_dp, double eps, int fd_order, double* h_i, double* h_j) {
    OGG_REQUIRE(n_i > 0 && n_j >= 0 && i && j && (h_i || h_j), OGG_EARG, "ogg_displaced_pole_numerical_h: bad argument");
    DevScratch s;
    double *d_i, *d_j, *d_hi = nullptr, *d_hj = nullptr;
    OGG_TRY(s.upload(&d_i, i, n_i));
    OGG_TRY(s.upload(&d_j, j, n_j));
    if (h_i) OGG_TRY(s.alloc(&d_hi, n_i * n_j));
    if (h_j) OGG_TRY(s.alloc(&d_hj, n_i * n_j));
    OGG_TRY(ogg_displaced_pole_numerical_h_dev(n_i, d_i, n_j, d_j, nx, ny, lon0, lat0, lon_dp, r_dp, eps, fd_order, d_hi, d_hj, nullptr));
    if (h_i) OGG_TRY(download(h_i, d_hi, n_i * n_j));
    if (h_j) OGG_TRY(download(h_j, d_hj, n_i * n_j));
    return OGG_OK;
}

int ogg_displaced_pole_metrics_quad_form(int arc_form, int order, long nx, long ny, double lon0, double lat0, double lon_dp, double r_dp,
                                         double Re, double* dxq, double* dyq, double* daq) {
    OGG_REQUIRE(order >= 2 && order <= 5, OGG_EORDER, "Uncoded order");
    OGG_REQUIRE(order == 2 || order == 4, OGG_EORDER, "order not coded");
    OGG_REQUIRE(nx > 0 && ny > 0 && dxq && dyq && daq, OGG_EARG, "ogg_displaced_pole_metrics_quad: bad argument");
    DevScratch s;
    double *d_dx, *d_dy, *d_da, *d_ws;
    const long ws_bytes = ogg_displaced_pole_quad_workspace_bytes(order, nx, ny);
    OGG_TRY(s.alloc(&d_dx, (ny + 1) * nx));
    OGG_TRY(s.alloc(&d_dy, ny * (nx + 1)));
    OGG_TRY(s.alloc(&d_da, ny * nx));
    OGG_TRY(s.alloc(&d_ws, (ws_bytes + 7) / 8));
    OGG_TRY(ogg_displaced_pole_metrics_quad_form_ws_dev(arc_form, order, nx, ny, lon0, lat0, lon_dp, r_dp, Re, 0, ny + 1, ny, d_dx, d_dy,
                                                        d_da, d_ws, ws_bytes, nullptr));
    OGG_TRY(download(dxq, d_dx, (ny + 1) * nx));
    OGG_TRY(download(dyq, d_dy, ny * (nx + 1)));
    OGG_TRY(download(daq, d_da, ny * nx));
    int flag = 0;
    OGG_TRY(ogg_workspace_error_flag_dev(d_ws, &flag, nullptr));
    OGG_REQUIRE(flag == 0, OGG_EHIP, "ogg_displaced_pole_metrics_quad: a look-back wait timed out (flag %d)", flag);
    return OGG_OK;
}

int ogg_displaced_pole_metrics_quad(int order, long nx, long ny, double lon0, double lat0, double lon_dp, double r_dp, double Re,
                                    double* dxq, double* dyq, double* daq) {
    return ogg_displaced_pole_metrics_quad_form(OGG_DP_ARC_LITERAL, order, nx, ny, lon0, lat0, lon_dp, r_dp, Re, dxq, dyq, daq);
}

// ---- small element-wise entry points -----------------------------------------------------------------------
int ogg_y_mercator(long Ni, long n, const double* phi_rad, double* y) {
    OGG_REQUIRE(n >= 0 && phi_rad && y, OGG_EARG, "ogg_y_mercator: bad argument");
    DevScratch s;
    double *d_in, *d_out;
    OGG_TRY(s.upload(&d_in, phi_rad, n));
    OGG_TRY(s.alloc(&d_out, n));
    OGG_TRY(ogg_y_mercator_dev(Ni, n, d_in, d_out, nullptr));
    return download(y, d_out, n);
}

int ogg_affine_index(long n, const double* idx, double a0, double len, double denom, double* out) {
    OGG_REQUIRE(n >= 0 && idx && out, OGG_EARG, "ogg_affine_index: bad argument");
    DevScratch s;
    double *d_in, *d_out;
    OGG_TRY(s.upload(&d_in, idx, n));
    OGG_TRY(s.alloc(&d_out, n));
    OGG_TRY(ogg_affine_index_dev(n, d_in, a0, len, denom, d_out, nullptr));
    return download(out, d_out, n);
}

int ogg_mdist(long n, const double* x1, const double* x2, double* out) {
    OGG_REQUIRE(n >= 0 && x1 && x2 && out, OGG_EARG, "ogg_mdist: bad argument");
    DevScratch s;
    double *d_a, *d_b, *d_out;
    OGG_TRY(s.upload(&d_a, x1, n));
    OGG_TRY(s.upload(&d_b, x2, n));
    OGG_TRY(s.alloc(&d_out, n));
    OGG_TRY(ogg_mdist_dev(n, d_a, d_b, d_out, nullptr));
    return download(out, d_out, n);
}

int ogg_haversine(long n, const double* lam0, const double* phi0, const double* lam1, const double* phi1, double* out) {
    OGG_REQUIRE(n >= 0 && lam0 && phi0 && lam1 && phi1 && out, OGG_EARG, "ogg_haversine: bad argument");
    DevScratch s;
    double *d0, *d1, *d2, *d3, *d_out;
    OGG_TRY(s.upload(&d0, lam0, n));
    OGG_TRY(s.upload(&d1, phi0, n));
    OGG_TRY(s.upload(&d2, lam1, n));
    OGG_TRY(s.upload(&d3, phi1, n));
    OGG_TRY(s.alloc(&d_out, n));
    OGG_TRY(ogg_haversine_dev(n, d0, d1, d2, d3, d_out, nullptr));
    return download(out, d_out, n);
}

int ogg_bipolar_cap_ij_array(long n_i, const double* i, long n_j, const double* j, long Ni, long Nj, double lat0_bp,
                             double lon_bp, double rp, double* h_i_inv, double* h_j_inv) {
    OGG_REQUIRE(n_i > 0 && n_j >= 0 && i && j && h_i_inv && h_j_inv, OGG_EARG, "ogg_bipolar_cap_ij_array: bad argument");
    DevScratch s;
    double *d_i, *d_j, *d_hi, *d_hj;
    OGG_TRY(s.upload(&d_i, i, n_i));
    OGG_TRY(s.upload(&d_j, j, n_j));
    OGG_TRY(s.alloc(&d_hi, n_i * n_j));
    OGG_TRY(s.alloc(&d_hj, n_i * n_j));
    OGG_TRY(ogg_bipolar_cap_ij_array_dev(n_i, d_i, n_j, d_j, Ni, Nj, lat0_bp, lon_bp, rp, d_hi, d_hj, nullptr));
    OGG_TRY(download(h_i_inv, d_hi, n_i * n_j));
    return download(h_j_inv, d_hj, n_i * n_j);
}

int ogg_displaced_pole_projection(long nj, long ni, const double* lon_grid, const double* lat_grid, double z0_re, double z0_im,
                                  double r_joint, double x_0, double* lam, double* phi) {
    OGG_REQUIRE(nj >= 0 && ni > 0 && lon_grid && lat_grid && lam && phi, OGG_EARG, "ogg_displaced_pole_projection: bad argument");
    DevScratch s;
    double *d_lon, *d_lat, *d_l, *d_p;
    OGG_TRY(s.upload(&d_lon, lon_grid, nj * ni));
    OGG_TRY(s.upload(&d_lat, lat_grid, nj * ni));
    OGG_TRY(s.alloc(&d_l, nj * ni));
    OGG_TRY(s.alloc(&d_p, nj * ni));
    OGG_TRY(ogg_displaced_pole_projection_dev(nj, ni, d_lon, d_lat, z0_re, z0_im, r_joint, x_0, d_l, d_p, nullptr));
    OGG_TRY(download(lam, d_l, nj * ni));
    return download(phi, d_p, nj * ni);
}

int ogg_monotonic_bounding(long nj, long ni, double* x, double x_0) {
    OGG_REQUIRE(nj >= 0 && ni > 0 && x, OGG_EARG, "ogg_monotonic_bounding: bad argument");
    DevScratch s;
    double* d_x;
    OGG_TRY(s.upload(&d_x, x, nj * ni));
    OGG_TRY(ogg_monotonic_bounding_dev(nj, ni, d_x, x_0, nullptr));
    return download(x, d_x, nj * ni);
}

}  // extern "C"
