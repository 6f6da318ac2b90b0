// K2: MIDAS stencil metrics (OGG:687-716) fused with the grid orientation angle (OGG:719-729).
//
// HBM-bound: 16 B read + 32 B written per cell.  One thread owns one column i and walks MIDAS_ROWS consecutive
// point rows; the values of row j+1 (x, y, the i-direction arc dx_i and sin/cos of the mid latitude) are carried in
// registers and become row j of the next step, so every point is read from HBM once per row strip and the
// transcendental work per cell is one sincos + one cos + one atan2.  The i-1 / i+1 neighbours of the 2x3 stencil
// come from wavefront shuffles (64 lanes); only lanes 0 and 63 touch memory for their outer neighbour.
#include "ogg_common.h"
#include "ogg_math.h"

namespace {

using namespace ogg;

#ifndef OGG_MIDAS_TX
#define OGG_MIDAS_TX 256
#endif
constexpr int MIDAS_TX = OGG_MIDAS_TX;   // columns per workgroup (4 waves)
constexpr int MIDAS_ROWS = 16;  // maximum point rows per workgroup; small bands use fewer (grid >= ~2000 workgroups)

struct MidasParams {
    long nrows_xy, ni1, n_pt_rows, n_cell_rows;
    int rows_per_block;
    const double* x;
    const double* y;
    double Re, Re2;
    double* dx;
    double* dy;
    double* area;
    double* angle;
};

struct RowVals {
    double xl, xc, xr, yl, yc, yr;
};

struct RowQ {
    double dxi, dyi, cl, sl;
};

OGG_DEV RowVals load_row(const double* __restrict__ x, const double* __restrict__ y, long row, long ni1, long i, int lane) {
    RowVals v;
    const long ic = (i < ni1) ? i : ni1 - 1;
    const double* xrow = x + row * ni1;
    const double* yrow = y + row * ni1;
    v.xc = xrow[ic];
    v.yc = yrow[ic];
    v.xr = wave_next(v.xc);
    v.yr = wave_next(v.yc);
    v.xl = wave_prev(v.xc);
    v.yl = wave_prev(v.yc);
    if (lane == 63) {
        const long ir = (ic + 1 < ni1) ? ic + 1 : ni1 - 1;
        v.xr = xrow[ir];
        v.yr = yrow[ir];
    }
    if (lane == 0) {
        const long il = (ic > 0) ? ic - 1 : 0;
        v.xl = xrow[il];
        v.yl = yrow[il];
    }
    return v;
}

// OGG:695-697 (+ OGG:711): quantities of the i-direction edge (i, i+1) of one row
OGG_DEV RowQ row_quantities(const RowVals& v) {
    RowQ q;
    const double lv = (0.5 * (v.yr + v.yc)) * kPi180;
    q.dxi = mdist(v.xr, v.xc) * kPi180;
    q.dyi = (v.yr - v.yc) * kPi180;
    sincos(lv, &q.sl, &q.cl);
    return q;
}

// OGG:699-702 at one column
OGG_DEV double dy_at(double x_up, double y_up, double x_c, double y_c, double Re) {
    const double lu = (0.5 * (y_up + y_c)) * kPi180;
    const double dxj = mdist(x_up, x_c) * kPi180;
    const double dyj = (y_up - y_c) * kPi180;
    const double t = dxj * cos(lu);
    return Re * sqrt(dyj * dyj + t * t);
}

OGG_DEV double dx_of(const RowQ& q, double Re) {
    const double t = q.dxi * q.cl;
    return Re * sqrt(q.dyi * q.dyi + t * t);
}

// OGG:725-728.  A caller's mesh may hold anything: the arctangent answers infinities and NaNs as the library's does (atan2_lib_any).
OGG_DEV double angle_of(const RowVals& v, long i, long ni1) {
    const double c = cos(v.yc * kPi180);
    double a;
    if (i == 0)
        a = atan2_lib_any(v.yr - v.yc, (v.xr - v.xc) * c);
    else if (i == ni1 - 1)
        a = atan2_lib_any(v.yc - v.yl, (v.xc - v.xl) * c);
    else
        a = atan2_lib_any(v.yr - v.yl, (v.xr - v.xl) * c);
    return div_pi180(a);
}

template <bool METRICS, bool AREAFIX>
__global__ __launch_bounds__(MIDAS_TX) void midas_angle_kernel(MidasParams p) {
    const long v = xcd_contiguous((long)blockIdx.y * gridDim.x + blockIdx.x, (long)gridDim.x * gridDim.y);  // rows slow: one row range per XCD
    const long vbx = v % gridDim.x, vby = v / gridDim.x;
    const long i = vbx * MIDAS_TX + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const long ni1 = p.ni1;
    const long ni = ni1 - 1;
    const bool active = i < ni1;
    const bool has_r = i + 1 < ni1;
    const long js = vby * p.rows_per_block;
    const long je = (js + p.rows_per_block < p.n_pt_rows) ? js + p.rows_per_block : p.n_pt_rows;

    RowVals cur = load_row(p.x, p.y, js, ni1, i, lane);
    RowQ q = {0.0, 0.0, 0.0, 0.0};
    if (METRICS) q = row_quantities(cur);
    for (long j = js; j < je; ++j) {
        double dx_cur = 0.0;
        if (METRICS) {
            dx_cur = dx_of(q, p.Re);
            if (p.dx && active && has_r) p.dx[j * ni + i] = dx_cur;
        }
        if (p.angle && active) p.angle[j * ni1 + i] = angle_of(cur, i, ni1);
        const bool cell_row = METRICS && (j < p.n_cell_rows);
        if (!cell_row && !(j + 1 < je)) break;  // workgroup-uniform
        RowVals up = load_row(p.x, p.y, j + 1, ni1, i, lane);
        RowQ qu = {0.0, 0.0, 0.0, 0.0};
        if (METRICS) qu = row_quantities(up);
        if (cell_row) {
            const double dy_c = dy_at(up.xc, up.yc, cur.xc, cur.yc, p.Re);
            if (p.dy && active) p.dy[j * ni1 + i] = dy_c;
            if (p.area && active && has_r) {
                double a;
                if (AREAFIX) {
                    a = p.Re2 * ((0.5 * (qu.dxi + q.dxi)) * (qu.sl - q.sl));  // OGG:713
                } else {
                    const double dy_r = dy_at(up.xr, up.yr, cur.xr, cur.yr, p.Re);
                    a = 0.25 * ((dx_of(qu, p.Re) + dx_cur) * (dy_r + dy_c));  // OGG:715
                }
                p.area[j * ni + i] = a;
            }
        }
        cur = up;
        q = qu;
    }
}

}  // namespace

extern "C" int ogg_grid_metrics_midas_dev(long nrows_xy, long ni1, const double* x, const double* y, long n_pt_rows,
                                          long n_cell_rows, double Re, int latlon_areafix, double* dx, double* dy,
                                          double* area, double* angle, void* stream) {
    OGG_REQUIRE(x && y, OGG_EARG, "ogg_grid_metrics_midas: null input");
    OGG_REQUIRE(ni1 >= 2 && nrows_xy >= 1, OGG_ESHAPE, "Input arrays do not have the same shape! (%ld x %ld)", nrows_xy, ni1);
    OGG_REQUIRE(n_pt_rows >= 0 && n_pt_rows <= nrows_xy && n_cell_rows >= 0 && n_cell_rows <= n_pt_rows &&
                    (n_cell_rows == 0 || n_cell_rows + 1 <= nrows_xy),
                OGG_ESHAPE, "ogg_grid_metrics_midas: rows xy=%ld pt=%ld cell=%ld inconsistent", nrows_xy, n_pt_rows, n_cell_rows);
    if (n_pt_rows == 0) return OGG_OK;
    const bool metrics = dx || dy || area;
    if (!(dy || area)) n_cell_rows = 0;
    const long gx = (ni1 + MIDAS_TX - 1) / MIDAS_TX;
    long rpb = (n_pt_rows * gx + 2047) / 2048;  // aim at >= 2048 workgroups
    rpb = rpb < (metrics ? 4 : 1) ? (metrics ? 4 : 1) : (rpb > MIDAS_ROWS ? MIDAS_ROWS : rpb);
    MidasParams p{nrows_xy, ni1, n_pt_rows, n_cell_rows, (int)rpb, x, y, Re, pow(Re, 2.0), dx, dy, area, angle};
    dim3 grid((unsigned)gx, (unsigned)((n_pt_rows + rpb - 1) / rpb));
    hipStream_t s = ogg::as_stream(stream);
    if (!metrics)
        midas_angle_kernel<false, true><<<grid, MIDAS_TX, 0, s>>>(p);
    else if (latlon_areafix)
        midas_angle_kernel<true, true><<<grid, MIDAS_TX, 0, s>>>(p);
    else
        midas_angle_kernel<true, false><<<grid, MIDAS_TX, 0, s>>>(p);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}
