// K3 / K4: Murray bipolar Arctic cap.
//   bipolar_projection                 OGG:33-100   (element-wise kernel + the mesh builder OGG:103-122)
//   bipolar_cap_metrics_quad_fast      OGG:136-188  (+ bipolar_cap_ij_array OGG:125-133, quadrature OGG:191-255)
//
// K4 is fp64-VALU bound (an acos, a tan, an atan, a cos and two sqrt per lattice point; 24 B written per cell).
// One workgroup owns a tile of QT_ROWS x QT_COLS cells.  Lobatto nodes on shared cell edges are bit-identical
// in the reference (node n-1 of cell k == node 0 of cell k+1 == k+1 exactly), so the tile evaluates each
// unique lattice point once -- (n-1)^2 instead of n^2 evaluations per cell -- into LDS, after splitting the
// projection into its row-only part (5 libm calls per lattice row), its column-only part (sincos + fmod per
// lattice column) and the per-point remainder.  Each thread then reduces one cell from LDS in the reference's
// summation order (OGG:216-221, 246-253).  The full lattice (138 M points at 1/8 degree) is never materialised.
#include <cstdlib>

#include "ogg_common.h"
#include "ogg_math.h"

namespace {

using namespace ogg;

// ---- pieces of OGG:41-95 ------------------------------------------------------------------------------
struct BpRow {       // depends on the (fractional) row index only
    double sphig;     // sin(phig*PI_180)                      OGG:44
    double beta2_inv; // tan(phig*PI_180)^2                    OGG:46
    double N_inv;     // OGG:75-78
};
struct BpCol {       // depends on the (fractional) column index only
    double sinla;     // OGG:43
    double alpha2;    // OGG:45
};

OGG_DEV BpRow bp_row(double phig_in, double rp) {
    BpRow r;
    const double phig = 90 - 2 * atan(tan(0.5 * (90 - phig_in) * kPi180) / rp) / kPi180;  // OGG:41
    const double pr = phig * kPi180;
    r.sphig = sin(pr);
    const double t = tan(pr);
    r.beta2_inv = t * t;
    const double chig = (90 - phig) * kPi180;
    const double tg = tan(chig / 2);
    const double rden2 = 1.0 / (1 + (rp * tg) * (rp * tg));
    const double N = rp * (1 + tg * tg) * rden2;
    r.N_inv = 1 / N;
    return r;
}

OGG_DEV BpCol bp_col(double lamg, double lon_bp) {
    BpCol c;
    const double tmp = mdist(lamg, lon_bp) * kPi180;  // OGG:42
    double s, co;
    sincos(tmp, &s, &co);
    c.sinla = s;
    c.alpha2 = co * co;
    return c;
}

// per-point remainder: phis (OGG:68-70) and the inverse scale factors (OGG:72-95)
OGG_DEV void bp_point(const BpRow& r, const BpCol& c, double rp, double& phis, double& h_i_inv, double& h_j_inv, double& rden_out) {
    const double rden = 1.0 / (1.0 + c.alpha2 * r.beta2_inv);  // OGG:47
    const double A = c.sinla * r.sphig;
    const double chic = acos(A);
    const double t = tan(chic / 2);
    const double rpt = rp * t;
    phis = 90 - 2 * atan(rpt) / kPi180;
    const double rden2 = 1.0 / (1 + rpt * rpt);
    const double M_inv = rp * (1 + t * t) * rden2;
    const double cp = cos(phis * kPi180);
    const double cos2phis = cp * cp;
    const double MM = M_inv * M_inv;
    const bool huge = fabs(r.beta2_inv) > kHuge;
    const double rr = rden * rden;
    double hj = cos2phis * c.alpha2 * (1 - c.alpha2) * r.beta2_inv * (1 + r.beta2_inv) * rr + MM * (1 - c.alpha2) * rden;
    if (huge) hj = MM;
    h_j_inv = sqrt(hj) * r.N_inv;
    double hi = cos2phis * (1 + r.beta2_inv) * rr + MM * c.alpha2 * r.beta2_inv * rden;
    if (huge) hi = MM;
    h_i_inv = sqrt(hi);
    rden_out = rden;
}

// Algebraically reduced form of bp_point for the quadrature lattice (metrics only).  With A = sinla*sphig in [0,1]:
//   tan(acos(A)/2)^2 = (1-A)/(1+A),   1 + (rp t)^2 = D/(1+A),  D = (1+A) + rp^2 (1-A)
//   M_inv = rp (1+t^2) / (1+(rp t)^2) = 2 rp / D
//   cos^2(phis PI/180) = sin^2(2 atan(rp t)) = 4 rp^2 (1-A)(1+A) / D^2
// so the acos -> tan -> atan -> cos round trip of OGG:69-79 collapses to one division.  The identities are exact; the
// results differ from the literal sequence only by rounding: <= 1e-14 relative where the cap latitude is >= 1.4 degrees
// from the pole (measured against the oracle on the 1/8 degree lattice; 4e-15 at >= 4 degrees).  Nearer the pole the
// LITERAL sequence loses digits (phis = 90 - small is rounded to 1 ulp of 90 before the cosine), and parity with the
// reference means reproducing that, so lattice rows within OGG_BP_ALG_GAP_DEG (default 2) of the pole keep bp_point.
OGG_DEV void bp_point_fast(const BpRow& r, const BpCol& c, double rp2x4, double rp2, double& h_i_inv, double& h_j_inv) {
    const double a = c.alpha2, b = r.beta2_inv;
    const double a1 = 1 - a, b1 = 1 + b;
    const double rden = 1.0 / (1.0 + a * b);
    const double A = c.sinla * r.sphig;
    const double p1 = 1 + A, m1 = 1 - A;
    const double E = 1.0 / (p1 + rp2 * m1);
    const double MM = rp2x4 * (E * E);            // M_inv^2
    const double cc = MM * (m1 * p1);             // cos^2(phis)
    const double rr = rden * rden;
    double hj = cc * a * a1 * b * b1 * rr + MM * a1 * rden;
    double hi = cc * b1 * rr + MM * a * b * rden;
    if (fabs(b) > kHuge) hj = hi = MM;
    h_j_inv = sqrt(hj) * r.N_inv;
    h_i_inv = sqrt(hi);
}

// lams of OGG:50-64
OGG_DEV double bp_lams(const BpRow& r, const BpCol& c, double rden, double lamg, double lon_bp) {
    double B = c.sinla * sqrt(rden);
    if (fabs(r.beta2_inv) > kHuge) B = 0.0;
    double lamc = asin(B) / kPi180;
    const double dl = lamg - lon_bp;
    if ((dl > 90) && (dl <= 180)) lamc = 180 - lamc;
    if ((dl > 180) && (dl <= 270)) lamc = 180 + lamc;
    if (dl > 270) lamc = 360 - lamc;
    if (dl == 90) lamc = 90;
    if (dl == 270) lamc = 270;
    return lamc + lon_bp;
}

// ---- element-wise projection on arbitrary inputs ---------------------------------------------------------
__global__ void bipolar_projection_kernel(long n, const double* __restrict__ lamg, const double* __restrict__ phig,
                                          double lon_bp, double rp, double* __restrict__ lams, double* __restrict__ phis,
                                          double* __restrict__ hi, double* __restrict__ hj) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double lg = lamg[k];
    const BpRow r = bp_row(phig[k], rp);
    const BpCol c = bp_col(lg, lon_bp);
    double ps, h_i, h_j, rden;
    bp_point(r, c, rp, ps, h_i, h_j, rden);
    if (lams) lams[k] = bp_lams(r, c, rden, lg, lon_bp);
    if (phis) phis[k] = ps;
    if (hi) hi[k] = h_i;
    if (hj) hj[k] = h_j;
}

// ---- mesh builder (OGG:103-122), rows j0 .. j0+nrows-1 ------------------------------------------------------
constexpr int MESH_TX = 256;

__global__ __launch_bounds__(MESH_TX) void bipolar_mesh_kernel(long Ni, long Nj, double lat0_bp, double lon_bp, long j0,
                                                               long nrows, double* __restrict__ lams,
                                                               double* __restrict__ phis, double* __restrict__ hi,
                                                               double* __restrict__ hj) {
    const long i = (long)blockIdx.x * MESH_TX + threadIdx.x;
    const long jl = blockIdx.y;
    if (i > Ni || jl >= nrows) return;
    const long j = j0 + jl;
    const double rp = tan(0.5 * (90 - lat0_bp) * kPi180);                              // OGG:117
    const double lamg = lon_bp + ((double)i * 360.0) / (double)Ni;                     // OGG:113
    const double phig = lat0_bp + ((double)j * (90 - lat0_bp)) / (double)Nj;           // OGG:115
    const BpRow r = bp_row(phig, rp);
    const BpCol c = bp_col(lamg, lon_bp);
    double ps, h_i, h_j, rden;
    bp_point(r, c, rp, ps, h_i, h_j, rden);
    const long ni1 = Ni + 1;
    if (lams) lams[jl * ni1 + i] = bp_lams(r, c, rden, lamg, lon_bp);
    if (phis) phis[jl * ni1 + i] = ps;
    if (hi && i < Ni) hi[jl * Ni + i] = h_i * 2 * kPi / (double)Ni;                    // OGG:119
    if (hj && j < Nj) hj[jl * ni1 + i] = h_j * kPi180 * (90 - lat0_bp) / (double)Nj;   // OGG:120
}

// ---- quadrature metrics -------------------------------------------------------------------------------------
constexpr int QT_ROWS = 4;    // cell rows per workgroup
constexpr int QT_COLS = 64;   // cell columns per workgroup
constexpr int QT_THREADS = QT_ROWS * QT_COLS;

struct QuadParams {
    long nx, ny;
    double lat0_bp, lon_bp, rp, Re;
    long j0;           // first cell row of the band
    long n_cell_rows;  // cell rows of the band evaluated by this launch
    int top_row;       // 1: this launch evaluates only dxq[ny][:] (exact j = ny lattice row) into band row out_row
    long out_row;
    long faithful_from;  // lattice rows (unique index (N-1)*cell + node) >= this use bp_point, the others bp_point_fast
    double* dxq;
    double* dyq;
    double* daq;
    QuadNodes q;
};

template <int N>
OGG_DEV double quad_average_1d(const double* y) {  // OGG:207-222
    if (N == 2) return (1.0 / 2.0) * (y[0] + y[1]);
    if (N == 3) return (1.0 / 6.0) * (4.0 * y[1] + (y[0] + y[2]));
    if (N == 4) return (1.0 / 12.0) * (5.0 * (y[1] + y[2]) + (y[0] + y[3]));
    return (1.0 / 180.0) * (64.0 * y[2] + (49.0 * (y[1] + y[3])) + 9.0 * (y[0] + y[4]));
}

template <int N, typename F>
OGG_DEV double quad_average_2d(F y) {  // y(jj, ii); OGG:225-255
    if (N == 2) {
        const double d = 1.0 / 2.0;
        return d * d * (y(0, 0) + y(0, 1) + y(1, 0) + y(1, 1));
    }
    if (N == 3) {
        const double d = 1.0 / 6.0;
        return d * d * (y(0, 0) + y(0, 2) + y(2, 0) + y(2, 2) + 4.0 * (y(0, 1) + y(1, 0) + y(1, 2) + y(2, 1) + 4.0 * y(1, 1)));
    }
    const double w4[4] = {1.0, 5.0, 5.0, 1.0};
    const double w5[5] = {9.0, 49.0, 64.0, 49.0, 9.0};
    const double d = (N == 4) ? (1.0 / 12.0) : (1.0 / 180.0);
    double ysum = 0.0;
#pragma unroll
    for (int jj = 0; jj < N; ++jj) {
#pragma unroll
        for (int ii = 0; ii < N; ++ii) {
            const double w = (N == 4) ? (w4[ii] * w4[jj]) : (w5[ii] * w5[jj]);
            ysum = ysum + w * y(jj, ii);
        }
    }
    return d * d * ysum;
}

template <int N>
__global__ __launch_bounds__(QT_THREADS) void bipolar_quad_kernel(QuadParams p) {
    constexpr int M = N - 1;               // unique nodes per cell and direction
    constexpr int NR = M * QT_ROWS + 1;    // lattice rows of the tile
    constexpr int NC = M * QT_COLS + 1;    // lattice columns of the tile
    __shared__ double s_dx[NR * NC];
    __shared__ double s_dy[NR * NC];
    __shared__ BpRow s_row[NR];
    __shared__ BpCol s_col[NC];

    const int tid = threadIdx.x;
    const long ci0 = (long)blockIdx.x * QT_COLS;                          // first cell column of the tile
    const long cj0 = p.top_row ? p.ny : p.j0 + (long)blockIdx.y * QT_ROWS;  // first cell row of the tile
    const long cj_end = p.top_row ? p.ny + 1 : p.j0 + p.n_cell_rows;     // one past the last cell row wanted
    const int nrows_cells = (int)((cj_end - cj0 < QT_ROWS) ? (cj_end - cj0) : QT_ROWS);
    const int ncols_cells = (int)((p.nx - ci0 < QT_COLS) ? (p.nx - ci0) : QT_COLS);
    const int nr = p.top_row ? 1 : M * nrows_cells + 1;  // lattice rows needed
    const int nc = M * ncols_cells + 1;                  // lattice columns needed

    // phase 0: row-only and column-only parts of the projection
    for (int l = tid; l < nr + nc; l += QT_THREADS) {
        if (l < nr) {
            const long cell = cj0 + l / M;
            const int node = l % M;
            double jv = lattice_node(p.q, node, cell);
            // OGG:146-147: the last node of cell ny-1 (== ny) is moved to ny-0.001; the first node of cell ny
            // (also == ny, used for dxq[ny]) is not.
            if (!p.top_row && cell == p.ny && node == 0) jv = (double)p.ny - 0.001;
            const double latg = p.lat0_bp + (jv * (90 - p.lat0_bp)) / (double)p.ny;   // OGG:127
            s_row[l] = bp_row(latg, p.rp);
        } else {
            const int lc = l - nr;
            const long cell = ci0 + lc / M;
            const int node = lc % M;
            const double iv = lattice_node(p.q, node, cell);
            const double lon = p.lon_bp + (iv * 360.0) / (double)p.nx;               // OGG:126
            s_col[lc] = bp_col(lon, p.lon_bp);
        }
    }
    __syncthreads();

    // phase 1: per-point remainder, scaled to per-index arc lengths (OGG:131-132)
    const int npts = nr * nc;
    // A lattice row keeps the literal sequence iff its absolute index is >= faithful_from (a function of the row alone,
    // so the result of a cell does not depend on how the cap is cut into tiles or bands).
    const long gu0 = (long)M * cj0;
    const double rp2 = p.rp * p.rp, rp2x4 = 4 * rp2;
    for (int pt = tid; pt < npts; pt += QT_THREADS) {
        const int lr = pt / nc;
        const int lc = pt - lr * nc;
        double h_i, h_j;
        if (p.top_row || gu0 + lr >= p.faithful_from) {
            double phis, rden;
            bp_point(s_row[lr], s_col[lc], p.rp, phis, h_i, h_j, rden);
        } else {
            bp_point_fast(s_row[lr], s_col[lc], rp2x4, rp2, h_i, h_j);
        }
        s_dx[lr * NC + lc] = h_i * 2 * kPi / (double)p.nx;
        s_dy[lr * NC + lc] = h_j * (90 - p.lat0_bp) * kPi180 / (double)p.ny;
    }
    __syncthreads();

    // phase 2: one thread per cell, reference summation order
    const int cr = tid / QT_COLS;
    const int cc = tid % QT_COLS;
    if (cc >= ncols_cells) return;
    const long ci = ci0 + cc;
    if (p.top_row) {
        if (cr != 0) return;
        double yv[N];
#pragma unroll
        for (int ii = 0; ii < N; ++ii) yv[ii] = s_dx[M * cc + ii];
        p.dxq[p.out_row * p.nx + ci] = quad_average_1d<N>(yv) * p.Re;
        return;
    }
    if (cr >= nrows_cells) return;
    const long out_r = cj0 + cr - p.j0;  // band-local output row
    const int r0 = M * cr, c0 = M * cc;
    {
        double yv[N];
#pragma unroll
        for (int ii = 0; ii < N; ++ii) yv[ii] = s_dx[r0 * NC + c0 + ii];
        p.dxq[out_r * p.nx + ci] = quad_average_1d<N>(yv) * p.Re;                     // OGG:183,186
#pragma unroll
        for (int jj = 0; jj < N; ++jj) yv[jj] = s_dy[(r0 + jj) * NC + c0];
        p.dyq[out_r * (p.nx + 1) + ci] = quad_average_1d<N>(yv) * p.Re;               // OGG:184,187
        if (ci == p.nx - 1) {  // column nx: first node column of the cell beyond the grid == right edge of this cell
#pragma unroll
            for (int jj = 0; jj < N; ++jj) yv[jj] = s_dy[(r0 + jj) * NC + c0 + M];
            p.dyq[out_r * (p.nx + 1) + p.nx] = quad_average_1d<N>(yv) * p.Re;
        }
    }
    const double da = quad_average_2d<N>([&](int jj, int ii) {
        return s_dx[(r0 + jj) * NC + c0 + ii] * s_dy[(r0 + jj) * NC + c0 + ii];       // OGG:178
    });
    p.daq[out_r * p.nx + ci] = da * p.Re * p.Re;                                       // OGG:185
}


// ---- bipolar_cap_ij_array (OGG:125-133) at arbitrary fractional indices ---------------------------------------
__global__ void bipolar_ij_kernel(long n_i, const double* __restrict__ iv, long n_j, const double* __restrict__ jv, long Ni,
                                  long Nj, double lat0_bp, double lon_bp, double rp, double* __restrict__ hi,
                                  double* __restrict__ hj) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long r = blockIdx.y;
    if (k >= n_i || r >= n_j) return;
    const double lon = lon_bp + (iv[k] * 360.0) / (double)Ni;
    const double lat = lat0_bp + (jv[r] * (90 - lat0_bp)) / (double)Nj;
    const BpRow row = bp_row(lat, rp);
    const BpCol col = bp_col(lon, lon_bp);
    double phis, h_i, h_j, rden;
    bp_point(row, col, rp, phis, h_i, h_j, rden);
    hi[r * n_i + k] = h_i * 2 * kPi / (double)Ni;
    hj[r * n_i + k] = h_j * (90 - lat0_bp) * kPi180 / (double)Nj;
}

QuadNodes make_nodes(int order) {  // OGG:191-204, host IEEE double
    QuadNodes q{};
    if (order == 2) {
        double a[] = {0.0, 1.0}, b[] = {1.0, 0.0};
        for (int k = 0; k < 2; ++k) q.a[k] = a[k], q.b[k] = b[k];
    } else if (order == 3) {
        double a[] = {0.0, 0.5, 1.0}, b[] = {1.0, 0.5, 0.0};
        for (int k = 0; k < 3; ++k) q.a[k] = a[k], q.b[k] = b[k];
    } else if (order == 4) {
        const double r5 = 0.5 / sqrt(5.0);
        double a[] = {0.0, 0.5 - r5, 0.5 + r5, 1.0}, b[] = {1.0, 0.5 + r5, 0.5 - r5, 0.0};
        for (int k = 0; k < 4; ++k) q.a[k] = a[k], q.b[k] = b[k];
    } else if (order == 5) {
        const double r37 = 0.5 * sqrt(3.0 / 7.0);
        double a[] = {0.0, 0.5 - r37, 0.5, 0.5 + r37, 1.0}, b[] = {1.0, 0.5 + r37, 0.5, 0.5 - r37, 0.0};
        for (int k = 0; k < 5; ++k) q.a[k] = a[k], q.b[k] = b[k];
    }
    return q;
}

template <int N>
int launch_quad(const QuadParams& p0, long n_dx_rows, hipStream_t s) {
    QuadParams p = p0;
    if (p.n_cell_rows > 0) {
        p.top_row = 0;
        dim3 grid((unsigned)((p.nx + QT_COLS - 1) / QT_COLS), (unsigned)((p.n_cell_rows + QT_ROWS - 1) / QT_ROWS));
        bipolar_quad_kernel<N><<<grid, QT_THREADS, 0, s>>>(p);
        OGG_LAUNCH_CHECK();
    }
    if (n_dx_rows > p.n_cell_rows) {
        p.top_row = 1;
        p.out_row = p.n_cell_rows;
        dim3 grid((unsigned)((p.nx + QT_COLS - 1) / QT_COLS), 1);
        bipolar_quad_kernel<N><<<grid, QT_THREADS, 0, s>>>(p);
        OGG_LAUNCH_CHECK();
    }
    return OGG_OK;
}

}  // namespace

namespace ogg {
QuadNodes quad_nodes_host(int order) { return make_nodes(order); }
}  // namespace ogg

extern "C" {

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

int ogg_bipolar_cap_mesh_dev(long Ni, long Nj, double lat0_bp, double lon_bp, long j0, long nrows, double* lams,
                             double* phis, double* h_i_inv, double* h_j_inv, void* stream) {
    OGG_REQUIRE(Ni > 0 && Nj > 0 && j0 >= 0 && nrows >= 0 && j0 + nrows <= Nj + 1, OGG_ESHAPE,
                "ogg_bipolar_cap_mesh: rows %ld..%ld outside 0..%ld", j0, j0 + nrows, Nj);
    if (nrows == 0) return OGG_OK;
    dim3 grid((unsigned)((Ni + 1 + MESH_TX - 1) / MESH_TX), (unsigned)nrows);
    bipolar_mesh_kernel<<<grid, MESH_TX, 0, ogg::as_stream(stream)>>>(Ni, Nj, lat0_bp, lon_bp, j0, nrows, lams, phis, h_i_inv,
                                                                     h_j_inv);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
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

int ogg_bipolar_cap_metrics_quad_dev(int order, long nx, long ny, double lat0_bp, double lon_bp, double rp, double Re,
                                     long j0, long n_dx_rows, long n_cell_rows, double* dxq, double* dyq, double* daq,
                                     void* stream) {
    OGG_REQUIRE(order >= 2 && order <= 5, OGG_EORDER, "Uncoded order");
    OGG_REQUIRE(nx > 0 && ny > 0 && dxq && (n_cell_rows <= 0 || (dyq && daq)), OGG_EARG, "ogg_bipolar_cap_metrics_quad: bad argument");
    OGG_REQUIRE(j0 >= 0 && n_cell_rows >= 0 && j0 + n_cell_rows <= ny &&
                    (n_dx_rows == n_cell_rows || (n_dx_rows == n_cell_rows + 1 && j0 + n_cell_rows == ny)),
                OGG_ESHAPE, "ogg_bipolar_cap_metrics_quad: band j0=%ld cell rows=%ld dx rows=%ld of ny=%ld", j0, n_cell_rows,
                n_dx_rows, ny);
    // OGG_BP_ALG_GAP_DEG: distance from the pole (degrees of cap latitude) below which the literal operation sequence of
    // the reference is kept; "inf" keeps it everywhere.  Default 2.0 (see bp_point_fast).
    double gap = 2.0;
    if (const char* e = getenv("OGG_BP_ALG_GAP_DEG")) gap = atof(e);
    long jf = (long)ceil((double)ny * (1.0 - gap / (90.0 - lat0_bp)));  // first cell row inside the gap
    if (!(jf > 0)) jf = 0;                                               // also catches gap = inf / NaN
    if (jf > ny) jf = ny;
    QuadParams p{nx, ny, lat0_bp, lon_bp, rp, Re, j0, n_cell_rows, 0, 0, (long)(order - 1) * jf, dxq, dyq, daq, make_nodes(order)};
    hipStream_t s = ogg::as_stream(stream);
    switch (order) {
        case 2: return launch_quad<2>(p, n_dx_rows, s);
        case 3: return launch_quad<3>(p, n_dx_rows, s);
        case 4: return launch_quad<4>(p, n_dx_rows, s);
        default: return launch_quad<5>(p, n_dx_rows, s);
    }
}

}  // extern "C"
