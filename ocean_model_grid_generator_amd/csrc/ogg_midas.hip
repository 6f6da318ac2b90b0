// K2: MIDAS stencil metrics (OGG:687-716) fused with the grid orientation angle (OGG:719-729).
//
// 16 B read + 32 B written per cell behind ~425 fp64 instructions (sincos + 2 cos + atan2 + 2 mod 360 + 2 sqrt per point).
//
// midas_tile_kernel (the default): a workgroup of 256 threads STAGES its tile through LDS -- up to R + 1 rows x 258 columns of x and
// y (one halo row above, one halo column either side, columns clamped at the edges of the mesh), all of its global loads issued up
// front and in flight together, one barrier -- and then every thread walks its column up the R rows of the tile out of LDS: the
// values of row j+1 (x, y, the i-direction arc dx_i and sin / cos of the mid latitude) are carried in registers and become row j of
// the next step, i+-1 neighbours are LDS reads.  The walk holds no global load, so a wave never waits on HBM between two rows (the
// streaming kernel below waits for the loads of row j+1 BEHIND the four stores of row j, in-order vmcnt: one dependent chain per
// wave), stores are fire-and-forget, and the loads of one workgroup hide behind the arithmetic of the 3-4 others on its CU.
// midas_angle_kernel (OGG_MIDAS_TILE_ROWS=0): the streaming form of rounds 1-3 -- one thread per column walks up to 16 rows straight
// from global memory, i+-1 neighbours from wavefront shuffles (only lanes 0 and 63 touch memory for their outer neighbour).
// The two evaluate the same expressions in the same order: bit-identical (tests/test_gpu_parity.py).
#include <cstdlib>

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

// ---- the arithmetic of the LDS-staged kernel: row_quantities / dy_at with mdist from ONE mod-360 reduction (mdist_one, ogg_math.h: the
// same bits as the two reductions of mdist, tested on 4e7 pairs) -- 427 -> ~400 instructions per point, 0.295 -> 0.253 ms for the three
// launches of the 1/8 degree grid on one box -- and non-temporal stores (the outputs are never read back: 0.258 -> 0.253 ms).  Measured
// and NOT kept (same box, same bits): sin / cos with their 16 constants as scalar operands (72 scalar registers spilled to vector
// lanes: +9 %), sqrt without its scaling steps behind a wave-uniform branch (+-0).
OGG_DEV RowQ t_row_quantities(const RowVals& v) {
    RowQ q;
    const double lv = (0.5 * (v.yr + v.yc)) * kPi180;
    q.dxi = mdist_one(v.xr, v.xc) * kPi180;
    q.dyi = (v.yr - v.yc) * kPi180;
    sincos(lv, &q.sl, &q.cl);
    return q;
}

OGG_DEV double t_dy_at(double x_up, double y_up, double x_c, double y_c, double Re) {
    const double lu = (0.5 * (y_up + y_c)) * kPi180;
    const double dxj = mdist_one(x_up, x_c) * kPi180;
    const double dyj = (y_up - y_c) * kPi180;
    const double t = dxj * cos(lu);
    return Re * sqrt(dyj * dyj + t * t);
}

OGG_DEV void t_store(double* p, double v) { __builtin_nontemporal_store(v, p); }

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

// ---- the LDS-staged form --------------------------------------------------------------------------------------------------
constexpr int TILE_W = MIDAS_TX + 2;     // columns of a staged row: i0 - 1 .. i0 + MIDAS_TX (clamped to the mesh)
constexpr int TILE_ROWS_MAX = 12;        // instantiations: 4, 6 (default: 28.9 KB, five workgroups per CU), 8, 12 rows per tile

OGG_DEV RowVals lds_row(const double* __restrict__ sx, const double* __restrict__ sy, int r, int t) {
    RowVals v;
    const double* xr = sx + r * TILE_W + t;
    const double* yr = sy + r * TILE_W + t;
    v.xl = xr[0], v.xc = xr[1], v.xr = xr[2];
    v.yl = yr[0], v.yc = yr[1], v.yr = yr[2];
    return v;
}

// five waves per SIMD (<= 96 vector registers: the compiler takes 93 when asked, 97 otherwise -- four waves) = five workgroups of the default
// tile per CU
template <int TR, bool METRICS, bool AREAFIX>
__global__ __launch_bounds__(MIDAS_TX, 5) void midas_tile_kernel(MidasParams p) {
    __shared__ double sx[(TR + 1) * TILE_W];
    __shared__ double sy[(TR + 1) * TILE_W];
    const long v = xcd_contiguous((long)blockIdx.y * gridDim.x + blockIdx.x, (long)gridDim.x * gridDim.y);  // rows slow: one row range per XCD
    const long vbx = v % gridDim.x, vby = v / gridDim.x;
    const int t = threadIdx.x;
    const long i0 = vbx * MIDAS_TX;
    const long i = i0 + t;
    const long ni1 = p.ni1;
    const long ni = ni1 - 1;
    const bool active = i < ni1;
    const bool has_r = i + 1 < ni1;
    const long js = vby * p.rows_per_block;                      // rows_per_block <= TR
    const long je = (js + p.rows_per_block < p.n_pt_rows) ? js + p.rows_per_block : p.n_pt_rows;
    // rows staged: the point rows js .. je-1, and row je when the last of them is a cell row (METRICS)
    const bool top_cell = METRICS && (je - 1 < p.n_cell_rows);
    const int n_stage = (int)(je - js) + (top_cell ? 1 : 0);
    {
        // every load of the tile is issued before the first one is waited for (registers, then LDS)
        constexpr int NIT = ((TR + 1) * TILE_W + MIDAS_TX - 1) / MIDAS_TX;
        const int n = n_stage * TILE_W;
        const double* __restrict__ gx = p.x + js * ni1;
        const double* __restrict__ gy = p.y + js * ni1;
        double xv[NIT], yv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int k = t + it * MIDAS_TX;
            xv[it] = 0.0, yv[it] = 0.0;
            if (k < n) {
                const int r = k / TILE_W, c = k - r * TILE_W;
                long col = i0 - 1 + c;
                col = col < 0 ? 0 : (col > ni1 - 1 ? ni1 - 1 : col);     // the clamps of load_row: il = max(i-1, 0), ir = min(i+1, ni1-1)
                xv[it] = gx[r * ni1 + col];
                yv[it] = gy[r * ni1 + col];
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int k = t + it * MIDAS_TX;
            if (k < n) sx[k] = xv[it], sy[k] = yv[it];
        }
    }
    __syncthreads();
    // The walk.  x and y are re-read from LDS where an expression needs them (a few ds_read_b64 next to ~430 vector instructions per point);
    // what a row hands to the next one is three numbers: its i-direction arc dx_i, sin of its mid latitude (OGG:711-713 take the
    // differences of both across the cell row) and its dx.
    double dxi_c = 0.0, sl_c = 0.0, dxv_c = 0.0;
    if (METRICS) {
        const RowQ q = t_row_quantities(lds_row(sx, sy, 0, t));
        dxi_c = q.dxi, sl_c = q.sl, dxv_c = dx_of(q, p.Re);
    }
    for (long j = js; j < je; ++j) {
        const int r = (int)(j - js);
        if (METRICS && p.dx && active && has_r) t_store(p.dx + j * ni + i, dxv_c);
        if (p.angle && active) t_store(p.angle + j * ni1 + i, angle_of(lds_row(sx, sy, r, t), i, ni1));
        const bool cell_row = METRICS && (j < p.n_cell_rows);
        if (!cell_row && !(j + 1 < je)) break;  // workgroup-uniform
        if (METRICS) {
            const RowQ qu = t_row_quantities(lds_row(sx, sy, r + 1, t));
            const double dxv_u = dx_of(qu, p.Re);
            if (cell_row) {
                const double* xc = sx + r * TILE_W + t + 1;
                const double* yc = sy + r * TILE_W + t + 1;
                const double dy_c = t_dy_at(xc[TILE_W], yc[TILE_W], xc[0], yc[0], p.Re);
                if (p.dy && active) t_store(p.dy + j * ni1 + i, dy_c);
                if (p.area && active && has_r) {
                    double a;
                    if (AREAFIX) {
                        a = p.Re2 * ((0.5 * (qu.dxi + dxi_c)) * (qu.sl - sl_c));  // OGG:713
                    } else {
                        const double dy_r = t_dy_at(xc[TILE_W + 1], yc[TILE_W + 1], xc[1], yc[1], p.Re);
                        a = 0.25 * ((dxv_u + dxv_c) * (dy_r + dy_c));  // OGG:715
                    }
                    t_store(p.area + j * ni + i, a);
                }
            }
            dxi_c = qu.dxi, sl_c = qu.sl, dxv_c = dxv_u;
        }
    }
}

long midas_env(const char* name, long dflt) {
    const char* e = getenv(name);
    return (e && *e) ? atol(e) : dflt;
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
    hipStream_t s = ogg::as_stream(stream);
    // rows per tile of the LDS-staged kernel: 6 -> 7 staged rows x 258 columns x 16 B = 28.9 KB, five workgroups per CU (the
    // kernel's 5 waves per SIMD); fewer for small bands (>= ~2048 workgroups); 0: the streaming kernel
    long tile_rows = midas_env("OGG_MIDAS_TILE_ROWS", 6);
    if (tile_rows > 0) {
        long rpb = (n_pt_rows * gx + 2047) / 2048;
        rpb = rpb < 2 ? 2 : (rpb > tile_rows ? tile_rows : rpb);
        rpb = rpb > TILE_ROWS_MAX ? TILE_ROWS_MAX : rpb;
        if (rpb > n_pt_rows) rpb = n_pt_rows;
        MidasParams p{nrows_xy, ni1, n_pt_rows, n_cell_rows, (int)rpb, x, y, Re, pow(Re, 2.0), dx, dy, area, angle};
        dim3 grid((unsigned)gx, (unsigned)((n_pt_rows + rpb - 1) / rpb));
#define OGG_MIDAS_LAUNCH(TR)                                                            \
    do {                                                                               \
        if (!metrics)                                                                  \
            midas_tile_kernel<TR, false, true><<<grid, MIDAS_TX, 0, s>>>(p);           \
        else if (latlon_areafix)                                                       \
            midas_tile_kernel<TR, true, true><<<grid, MIDAS_TX, 0, s>>>(p);            \
        else                                                                           \
            midas_tile_kernel<TR, true, false><<<grid, MIDAS_TX, 0, s>>>(p);           \
    } while (0)
        // the smallest instantiation that holds the tile (its static LDS sets how many workgroups a CU takes)
        if (rpb <= 4)
            OGG_MIDAS_LAUNCH(4);
        else if (rpb <= 6)
            OGG_MIDAS_LAUNCH(6);
        else if (rpb <= 8)
            OGG_MIDAS_LAUNCH(8);
        else
            OGG_MIDAS_LAUNCH(TILE_ROWS_MAX);
#undef OGG_MIDAS_LAUNCH
        OGG_LAUNCH_CHECK();
        return OGG_OK;
    }
    long rpb = (n_pt_rows * gx + 2047) / 2048;  // aim at >= 2048 workgroups
    rpb = rpb < (metrics ? 4 : 1) ? (metrics ? 4 : 1) : (rpb > MIDAS_ROWS ? MIDAS_ROWS : rpb);
    MidasParams p{nrows_xy, ni1, n_pt_rows, n_cell_rows, (int)rpb, x, y, Re, pow(Re, 2.0), dx, dy, area, angle};
    dim3 grid((unsigned)gx, (unsigned)((n_pt_rows + rpb - 1) / rpb));
    if (!metrics)
        midas_angle_kernel<false, true><<<grid, MIDAS_TX, 0, s>>>(p);
    else if (latlon_areafix)
        midas_angle_kernel<true, true><<<grid, MIDAS_TX, 0, s>>>(p);
    else
        midas_angle_kernel<true, false><<<grid, MIDAS_TX, 0, s>>>(p);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}
