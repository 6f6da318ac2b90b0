// K1+K2 fused for sub-grids that are lat-lon BY CONSTRUCTION (Mercator, Southern Ocean, regular southern cap):
// x[j][i] = lon[i], y[j][i] = lat[j] (OGG:430-432, 840-841), so every term of the MIDAS metrics (OGG:695-713) and of
// angle_x (OGG:725-728) factors into a per-row scalar times a per-column scalar.  The kernel reads the two 1-D axes
// (a few thousand values, L2-resident), evaluates the transcendental functions once per ROW in LDS, and streams out
// all six fields: 48 B written per cell, nothing read from HBM -- the algorithmic minimum of SURVEY 8(d).
//
// It is bit-identical to tile_latlon_kernel followed by midas_angle_kernel, because it performs the same operations on
// the same operands and only hoists those that do not depend on i (or on j):
//   dy_i = (lat_j - lat_j) PI/180 = 0, so dx = Re sqrt(0 + (dlam_i cos(lv_j))^2);  lv_j = (0.5 (lat_j+lat_j)) PI/180 = lat_j PI/180
//   dx_j = mdist(lon_i, lon_i) PI/180 = 0, so dy = Re sqrt(dphi_j^2 + 0) is a per-row constant
//   area = Re^2 ((0.5 (dlam_i + dlam_i)) (sin lv_{j+1} - sin lv_j))
//   angle = atan2(+0, (lon_{i+1} - lon_{i-1}) cos(lat_j PI/180)) / (PI/180): the IEEE value of atan2(+0, p) is +0 for
//           p > 0 or p = +0 and pi for p < 0 or p = -0.
// tests/test_gpu_pipeline.py checks the bit-identity against the generic stencil kernel.
#include <cstdlib>

#include "ogg_common.h"
#include "ogg_math.h"

namespace {
using namespace ogg;

constexpr int LF_TX = 256;
constexpr int LF_ROWS = 32;  // maximum rows per workgroup (LDS row table); small bands use fewer so that the grid stays >= ~2000 workgroups

struct FusedParams {
    long n_pt_rows, n_cell_rows, ni1;
    int rows_per_block;
    const double* lat;  // lat[0 .. n_pt_rows-1] (+1 more entry when n_cell_rows == n_pt_rows)
    const double* lon;  // lon[0 .. ni1-1]
    double Re, Re2;
    int metrics;
    double* x;
    double* y;
    double* dx;
    double* dy;
    double* area;
    double* angle;
};

struct RowScalars {
    double lat, sl, cl, dy;
};

__global__ __launch_bounds__(LF_TX) void latlon_fused_kernel(FusedParams p) {
    __shared__ RowScalars s_row[LF_ROWS + 1];
    const int tid = threadIdx.x;
    // Row strips are taken grid-stride: the launch caps the number of resident workgroups (an HBM-write-bound kernel needs
    // only a few waves per SIMD) so that a VALU-bound kernel running on another stream can share the CUs.
    for (long strip = blockIdx.y; strip * p.rows_per_block < p.n_pt_rows; strip += gridDim.y) {
    const long js = strip * p.rows_per_block;
    const long je = (js + p.rows_per_block < p.n_pt_rows) ? js + p.rows_per_block : p.n_pt_rows;
    const int nrows = (int)(je - js);
    // per-row scalars for rows js .. je (row je only when a cell row needs it)
    if (tid <= nrows) {
        const long j = js + tid;
        const bool have = (tid < nrows) || (j - 1 < p.n_cell_rows && j < p.n_pt_rows + (p.n_cell_rows == p.n_pt_rows ? 1 : 0));
        RowScalars r = {0.0, 0.0, 0.0, 0.0};
        if (have) {
            r.lat = p.lat[j];
            const double lv = (0.5 * (r.lat + r.lat)) * kPi180;
            sincos(lv, &r.sl, &r.cl);
        }
        s_row[tid] = r;
    }
    __syncthreads();
    if (p.metrics && tid < nrows) {  // dy of cell row j needs lat_{j+1}
        const long j = js + tid;
        if (j < p.n_cell_rows) {
            const double dyj = (s_row[tid + 1].lat - s_row[tid].lat) * kPi180;
            s_row[tid].dy = p.Re * sqrt(dyj * dyj + 0.0);
        }
    }
    __syncthreads();

    const long i = (long)blockIdx.x * LF_TX + tid;
    const long ni1 = p.ni1, ni = ni1 - 1;
    if (i < ni1) {
    const double lon_c = p.lon[i];
    const double lon_r = p.lon[(i + 1 < ni1) ? i + 1 : i];
    const double lon_l = p.lon[(i > 0) ? i - 1 : 0];
    const double dlam = mdist(lon_r, lon_c) * kPi180;      // OGG:696 (column-only)
    const double hdlam = 0.5 * (dlam + dlam);              // OGG:713
    double xdiff;                                          // OGG:725-727
    if (i == 0)
        xdiff = lon_r - lon_c;
    else if (i == ni)
        xdiff = lon_c - lon_l;
    else
        xdiff = lon_r - lon_l;
    const bool has_r = i < ni;
    const double pi_deg = kPi / kPi180;
    for (int r = 0; r < nrows; ++r) {
        const long j = js + r;
        const RowScalars rs = s_row[r];
        p.x[j * ni1 + i] = lon_c;
        p.y[j * ni1 + i] = rs.lat;
        const double pa = xdiff * rs.cl;
        const bool zero = (pa > 0.0) || (pa == 0.0 && !signbit(pa));
        p.angle[j * ni1 + i] = (pa != pa) ? pa : (zero ? 0.0 / kPi180 : pi_deg);
        if (p.metrics) {
            if (has_r) {
                const double t = dlam * rs.cl;
                p.dx[j * ni + i] = p.Re * sqrt(0.0 + t * t);
            }
            if (j < p.n_cell_rows) {
                p.dy[j * ni1 + i] = rs.dy;
                if (has_r) p.area[j * ni + i] = p.Re2 * (hdlam * (s_row[r + 1].sl - rs.sl));
            }
        }
    }
    }
    __syncthreads();  // the row table is rewritten by the next strip
    }
}

}  // namespace

extern "C" int ogg_latlon_supergrid_dev(long n_pt_rows, long n_cell_rows, long ni1, const double* lat1d, const double* lon1d,
                                        double Re, int metrics, double* x, double* y, double* dx, double* dy, double* area,
                                        double* angle, void* stream) {
    OGG_REQUIRE(n_pt_rows >= 0 && ni1 >= 2 && n_cell_rows >= 0 && n_cell_rows <= n_pt_rows, OGG_ESHAPE,
                "ogg_latlon_supergrid: rows pt=%ld cell=%ld ni1=%ld", n_pt_rows, n_cell_rows, ni1);
    OGG_REQUIRE(lat1d && lon1d && x && y && angle, OGG_EARG, "ogg_latlon_supergrid: null pointer");
    OGG_REQUIRE(!metrics || (dx && (n_cell_rows == 0 || (dy && area))), OGG_EARG, "ogg_latlon_supergrid: null metrics output");
    if (n_pt_rows == 0) return OGG_OK;
    const long gx = (ni1 + LF_TX - 1) / LF_TX;
    long rpb = (n_pt_rows * gx + 2047) / 2048;  // aim at >= 2048 workgroups (8 per CU)
    rpb = rpb < 4 ? 4 : (rpb > LF_ROWS ? LF_ROWS : rpb);
    FusedParams p{n_pt_rows, metrics ? n_cell_rows : 0, ni1, (int)rpb, lat1d, lon1d, Re, pow(Re, 2.0), metrics, x, y, dx, dy, area, angle};
    long gy = (n_pt_rows + rpb - 1) / rpb;
    static const long max_wg = getenv("OGG_FUSED_MAX_WG") ? atol(getenv("OGG_FUSED_MAX_WG")) : 128;  // ~100 workgroups already saturate the HBM write path (measured)
    if (gx * gy > max_wg) gy = (max_wg + gx - 1) / gx;
    dim3 grid((unsigned)gx, (unsigned)gy);
    latlon_fused_kernel<<<grid, LF_TX, 0, ogg::as_stream(stream)>>>(p);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}
