// K5 / K6: displaced-pole Southern cap.
//   displacedPoleCap_projection / _mesh      OGG:447-506   (generate_displaced_pole_grid OGG:509-518)
//   monotonic_bounding                       OGG:470-475
//   great_arc_distance, numerical_hi/hj      OGG:522-562
//   displacedPoleCap_metrics_quad            OGG:565-601
//
// Kernels:
//   dpole_eval_kernel + _unwrap_kernel   lam, phi on a lattice of (fractional) indices, exact 360-degree unwrap     -> K5
//   dpole_direct_kernel                  displacedPoleCap_projection on explicit grids / bare monotonic_bounding
//   dpole_h_kernel<F, group>             literal finite-difference scale factors h_i / h_j (numerical_hi / _hj)
//   dpole_chord_tables / _h_kernel<F>    the same stencil with the chord form of the great-arc distance            -> K6
//   dpole_quad_reduce_kernel<N>          Lobatto quadrature of h_i, h_j, h_i*h_j per cell                           -> K6
//
// monotonic_bounding is a sequential scan along i: column k is lowered by 360 iff v_k - x_{k-1} > 100 where x_{k-1} is the
// ALREADY ADJUSTED previous column.  With s_k in {0,1} the "was lowered" state, s_k = f_k(s_{k-1}) with
// f_k(0) = [v_k - v_{k-1} > 100], f_k(1) = [v_k - (v_{k-1} - 360) > 100] -- a composition of 1-bit maps, which is
// associative.  A workgroup sweeps a row in chunks of SW_TX columns; inside a chunk the maps of all probes (bit-packed,
// one bit per probe) are composed with a wave64 shuffle scan plus a 4-entry LDS carry, and the state and raw value of the
// last column are carried to the next chunk.  The comparison values are formed exactly as the reference forms them, so
// the unwrap is bit-faithful to the sequential loop.
#include <cstdlib>

#include "ogg_common.h"
#include "ogg_math.h"

namespace ogg {
QuadNodes quad_nodes_host(int order);
}

namespace {

using namespace ogg;

constexpr int SW_TX = 256;
constexpr int SW_WAVES = SW_TX / 64;

struct SweepParams {
    // geometry of the cap (OGG:478-495)
    long ni, nj;
    double lon0, lat0, lam_pole, r_pole;
    // lattice of the mesh kernel
    long n_cols;          // columns per row
    const double* i_arr;  // column indices or NULL for 0,1,2,...
    const double* j_arr;  // row indices or NULL for j0, j0+1, ...
    long j0;
    double* out0;         // lam
    double* out1;         // phi
};

struct DpConst {
    double z0r, z0i, r_joint;
};

OGG_DEV DpConst dp_const(const SweepParams& p) {
    DpConst c;
    c.r_joint = tan((90 + p.lat0) * kPi180);  // OGG:494
    double s, co;
    sincos(p.lam_pole * kPi180, &s, &co);
    c.z0r = p.r_pole * co;                    // OGG:495
    c.z0i = p.r_pole * s;
    return c;
}

// column-only part of OGG:451-452: e' = (e - z0) / (1 - conj(z0) e)
OGG_DEV cplx dp_column(double iv, const SweepParams& p, const DpConst& c) {
    const double lon = p.lon0 + (iv * 360.0) / (double)p.ni;  // OGG:479
    double s, co;
    sincos(lon * kPi180, &s, &co);
    const cplx e = {co, s};
    const cplx num = {e.re - c.z0r, e.im - c.z0i};
    const cplx cz = cmul(cplx{c.z0r, -c.z0i}, e);
    const cplx den = {1.0 - cz.re, 0.0 - cz.im};
    return cdiv(num, den);
}

// row-only part of OGG:448: r = tan((90+lat) PI/180) / r_joint
OGG_DEV double dp_row_radius(double jv, const SweepParams& p, const DpConst& c) {
    const double lat = -90.0 + (jv * (p.lat0 - (-90.0))) / (double)p.nj;  // OGG:480-482
    return tan((90 + lat) * kPi180) / c.r_joint;
}

// per-point remainder of OGG:454-466: raw longitude (before the unwrap) and latitude
OGG_DEV void dp_point(double r, cplx ep, const DpConst& c, double& lam_raw, double& phi) {
    const cplx z = {r * ep.re, r * ep.im};
    const cplx num = {z.re + c.z0r, z.im + c.z0i};
    const cplx cz = cmul(cplx{c.z0r, -c.z0i}, z);
    const cplx den = {1 + cz.re, cz.im};
    const cplx w = cdiv(num, den);
    lam_raw = atan2(w.im, w.re) * k180Pi;  // np.angle(deg=True)
    const double rw = cabs_np(w);
    phi = -90 + div_pi180(atan(rw * c.r_joint));
}

// OGG:527-532 for point0 = (lam0, phi0), point1 = (lam1, phi1) in degrees
OGG_DEV double great_arc(double lam0d, double phi0d, double lam1d, double phi1d) {
    const double lam0 = lam0d * kPi180, phi0 = phi0d * kPi180;
    const double lam1 = lam1d * kPi180, phi1 = phi1d * kPi180;
    const double dphi = phi1 - phi0, dlam = lam1 - lam0;
    const double sp = sin(0.5 * dphi), sl = sin(0.5 * dlam);
    const double d = sp * sp + sl * sl * cos(phi0) * cos(phi1);
    return 2.0 * asin(sqrt(d));
}


// ---- monotonic_bounding (OGG:470-475) as a workgroup-wide scan of 1-bit maps ------------------------------------
template <int NP, int TX = SW_TX>
struct UnwrapShared {
    double v[NP][TX];
    unsigned w0[TX / 64], w1[TX / 64];
};

// Packed "was lowered by 360" states of this thread's column for NP independent scans.  v: raw values of this
// column; first_col: this is column 0 of the row (compared with seed[], no dependence on a previous state);
// carry_v / carry_state_ptr: raw values and states of the column just before this chunk (LDS; the caller rewrites
// them only AFTER this function returns).  Contains two workgroup barriers: every thread of the workgroup must call it
// (inactive threads contribute the identity map).
template <int NP, int TX = SW_TX>
OGG_DEV unsigned unwrap_states(const double* v, bool active, bool first_col, const double* seed, const double* carry_v,
                               const unsigned* carry_state_ptr, UnwrapShared<NP, TX>& sh) {
    constexpr unsigned ALL = (NP >= 32) ? 0xffffffffu : ((1u << NP) - 1u);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
#pragma unroll
    for (int q = 0; q < NP; ++q) sh.v[q][tid] = v[q];
    __syncthreads();
    // The carry of the previous chunk was written after that chunk's second barrier; this read sits behind the
    // barrier above, and the next write sits behind the barrier below: no race in either direction.
    const unsigned carry_state = *carry_state_ptr;
    double cv[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) cv[q] = carry_v[q];
    unsigned f0 = 0u, f1 = ALL;  // identity map
    if (active) {
        f0 = 0u;
        f1 = 0u;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            bool b0, b1;
            if (first_col) {
                b0 = b1 = (v[q] - seed[q] > 100);
            } else {
                const double vp = (tid > 0) ? sh.v[q][tid - 1] : cv[q];
                b0 = (v[q] - vp > 100);          // previous column was not lowered
                b1 = (v[q] - (vp - 360) > 100);  // previous column was lowered (x_im1 = vp - 360, OGG:473-474)
            }
            f0 |= (b0 ? 1u : 0u) << q;
            f1 |= (b1 ? 1u : 0u) << q;
        }
    }
    // inclusive wave64 scan of the composition (later o earlier)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned e0 = __shfl_up(f0, off);
        const unsigned e1 = __shfl_up(f1, off);
        if (lane >= off) {
            const unsigned h0 = (e0 & f1) | (~e0 & f0);
            const unsigned h1 = (e1 & f1) | (~e1 & f0);
            f0 = h0;
            f1 = h1;
        }
    }
    if (lane == 63) {
        sh.w0[wave] = f0;
        sh.w1[wave] = f1;
    }
    __syncthreads();
    unsigned st = carry_state;
    for (int w = 0; w < wave; ++w) st = (st & sh.w1[w]) | (~st & sh.w0[w]);
    return ((st & f1) | (~st & f0)) & ALL;
}

template <int F>
OGG_DEV double central_difference(const double* ds, double reps) {  // OGG:539-546
    if (F == 2) return 0.5 * ds[0] * reps;
    if (F == 4) return (8.0 * ds[0] - ds[1]) * (1.0 / 12.0) * reps;
    return (45.0 * ds[0] - 9.0 * ds[1] + ds[2]) * (1.0 / 60.0) * reps;
}

template <int N>
OGG_DEV double qavg_1d(const double* y) {  // OGG:207-222
    if (N == 1) return y[0];
    if (N == 2) return (1.0 / 2.0) * (y[0] + y[1]);
    if (N == 3) return (1.0 / 6.0) * (4.0 * y[1] + (y[0] + y[2]));
    if (N == 4) return (1.0 / 12.0) * (5.0 * (y[1] + y[2]) + (y[0] + y[3]));
    return (1.0 / 180.0) * (64.0 * y[2] + (49.0 * (y[1] + y[3])) + 9.0 * (y[0] + y[4]));
}

// ---- mesh (OGG:488-518) in two launches ---------------------------------------------------------------------------------
// The unwrap of a row is a sequential chain over its chunks, so a kernel that evaluates the projection inside that chain takes
// n_cols / TX projection latencies per row whatever the number of rows -- a band of 36 rows (one rank's share on 8 GPUs) kept
// 36 CUs busy for 28 us.  Hence: (1) every point's raw longitude and latitude, fully parallel (one workgroup per 256 columns
// of a row); (2) the unwrap alone, one workgroup of 1024 threads per row reading the raw longitudes back (L2) and lowering them
// in place.  Same arithmetic per point, same scan: the same bits as the one-kernel form.
__global__ __launch_bounds__(SW_TX) void dpole_eval_kernel(SweepParams p) {
    __shared__ DpConst s_c;
    __shared__ double s_r;
    const int tid = threadIdx.x;
    const long row = blockIdx.y;
    if (tid == 0) {
        s_c = dp_const(p);
        const double jv = p.j_arr ? p.j_arr[row] : (double)(p.j0 + row);
        s_r = dp_row_radius(jv, p, s_c);                                   // row-only
    }
    __syncthreads();
    const long g = (long)blockIdx.x * SW_TX + tid;
    if (g >= p.n_cols) return;
    const DpConst c = s_c;
    const double iv = p.i_arr ? p.i_arr[g] : (double)g;
    double lam_raw, ph;
    dp_point(s_r, dp_column(iv, p, c), c, lam_raw, ph);
    p.out0[row * p.n_cols + g] = lam_raw;
    p.out1[row * p.n_cols + g] = ph;
}

// Unwrap of one row by one workgroup in ONE step: thread t owns the CPT consecutive columns t*CPT .. and composes their
// 1-bit maps sequentially (both incoming states), a wave64 shuffle scan and a scan over the waves give every thread its
// incoming state, and a second sequential walk lowers its columns.  Two barriers per row instead of two per 1024 columns.
template <int TX>
__global__ __launch_bounds__(TX) void dpole_unwrap_kernel(SweepParams p, int cpt) {
    __shared__ unsigned s_w0[TX / 64], s_w1[TX / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long row = blockIdx.x;
    const double i_first = p.i_arr ? p.i_arr[0] : 0.0;
    const double seed = p.lon0 + (i_first * 360.0) / (double)p.ni;         // lon_grid[0,0] (OGG:463)
    double* lam = p.out0 + row * p.n_cols;
    const long g0 = (long)tid * cpt;
    const long g1 = (g0 + cpt < p.n_cols) ? g0 + cpt : p.n_cols;
    // f0 / f1: state after this thread's columns when the column before them was not / was lowered (identity if it owns none)
    unsigned f0 = 0u, f1 = 1u;
    if (g0 < g1) {
        unsigned s0 = 0u, s1 = 1u;
        double prev = (g0 > 0) ? lam[g0 - 1] : 0.0;
        for (long g = g0; g < g1; ++g) {
            const double v = lam[g];
            if (g == 0) {
                s0 = s1 = (v - seed > 100) ? 1u : 0u;                      // OGG:471-472: compared with x_0, no incoming state
            } else {
                s0 = (v - (s0 ? prev - 360 : prev) > 100) ? 1u : 0u;       // OGG:473-474: x_im1 is the ADJUSTED previous column
                s1 = (v - (s1 ? prev - 360 : prev) > 100) ? 1u : 0u;
            }
            prev = v;
        }
        f0 = s0, f1 = s1;
    }
    // inclusive scan of the composition (later o earlier) over the wave, then over the waves
    unsigned c0 = f0, c1 = f1;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned e0 = __shfl_up(c0, off), e1 = __shfl_up(c1, off);
        if (lane >= off) {
            const unsigned h0 = e0 ? c1 : c0, h1 = e1 ? c1 : c0;
            c0 = h0, c1 = h1;
        }
    }
    if (lane == 63) s_w0[wave] = c0, s_w1[wave] = c1;
    __syncthreads();
    unsigned st = 0u;  // state of the column before this WAVE's first column (row start: no previous column, value unused)
    for (int w = 0; w < wave; ++w) st = st ? s_w1[w] : s_w0[w];
    // exclusive within the wave: state after the previous lane's columns
    const unsigned p0 = __shfl_up(c0, 1), p1 = __shfl_up(c1, 1);
    unsigned s_in = (lane == 0) ? st : (st ? p1 : p0);
    const double prev0 = (g0 > 0 && g0 < g1) ? lam[g0 - 1] : 0.0;
    __syncthreads();   // every thread has read its left neighbour's raw value before anybody lowers one (uniform barrier)
    if (g0 < g1) {
        double prev = prev0;
        unsigned sp = s_in;
        for (long g = g0; g < g1; ++g) {
            const double v = lam[g];
            unsigned sc;
            if (g == 0)
                sc = (v - seed > 100) ? 1u : 0u;
            else
                sc = (v - (sp ? prev - 360 : prev) > 100) ? 1u : 0u;
            if (sc) lam[g] = v - 360;                                      // OGG:473
            prev = v;
            sp = sc;
        }
    }
}

// ---- finite-difference scale factors on a lattice row (OGG:535-562), one workgroup per lattice row and probe group ----
// JGROUP = false: h_i from the 2H probes (j, i +- m eps);  JGROUP = true: h_j from the probes (j +- m eps, i).
// The two groups of a lattice row are independent scans, so they run as separate workgroups: twice the parallelism and
// half the live registers of a kernel that carries all 4H probes.
struct HParams {
    long ni, nj;
    double lon0, lat0, lam_pole, r_pole, eps;
    long n_cols, n_rows;
    const double* i_arr;  // explicit column indices, or NULL: Lobatto nodes of cells 0.. (lattice_M unique nodes per cell)
    const double* j_arr;  // explicit row indices, or NULL: Lobatto nodes of cell rows row0_cell..
    int lattice_M;
    long row0_cell;
    QuadNodes q;
    double* h_i;          // [n_rows][n_cols] or NULL
    double* h_j;
};

template <int F, bool JGROUP>
__global__ __launch_bounds__(SW_TX) void dpole_h_kernel(HParams p) {
    constexpr int H = F / 2;
    __shared__ UnwrapShared<F> s_u;
    __shared__ double s_carry_v[F];
    __shared__ unsigned s_carry_state;
    __shared__ double s_r[F + 1];
    const int tid = threadIdx.x;
    const long row = blockIdx.x;
    double* out = JGROUP ? p.h_j : p.h_i;
    SweepParams sp{};
    sp.ni = p.ni, sp.nj = p.nj, sp.lon0 = p.lon0, sp.lat0 = p.lat0, sp.lam_pole = p.lam_pole, sp.r_pole = p.r_pole;
    const DpConst c = dp_const(sp);
    const double reps = 1.0 / p.eps;
    const int M = p.lattice_M;
    if (tid <= F) {  // s_r[0]: base row; s_r[2m-1]: j + m eps; s_r[2m]: j - m eps
        double jv = p.j_arr ? p.j_arr[row] : lattice_node(p.q, (int)(row % M), p.row0_cell + row / M);
        if (tid > 0) {
            const double off = (double)((tid + 1) / 2) * p.eps;
            jv = (tid & 1) ? jv + off : jv - off;
        }
        if (tid == 0 || JGROUP) s_r[tid] = dp_row_radius(jv, sp, c);
    }
    const double i_first = p.i_arr ? p.i_arr[0] : lattice_node(p.q, 0, 0);
    __syncthreads();
    for (long c0 = 0; c0 < p.n_cols; c0 += SW_TX) {
        const long g = c0 + tid;
        const bool active = g < p.n_cols;
        double v[F], ph[F], seed[F];
#pragma unroll
        for (int q = 0; q < F; ++q) v[q] = ph[q] = seed[q] = 0.0;
        if (active) {
            const double iv = p.i_arr ? p.i_arr[g] : lattice_node(p.q, (int)(g % M), g / M);
            if (JGROUP) {
                const cplx ep = dp_column(iv, sp, c);
#pragma unroll
                for (int q = 0; q < F; ++q) dp_point(s_r[q + 1], ep, c, v[q], ph[q]);
            } else {
#pragma unroll
                for (int m = 1; m <= H; ++m) {
                    const double off = (double)m * p.eps;
                    dp_point(s_r[0], dp_column(iv + off, sp, c), c, v[2 * (m - 1)], ph[2 * (m - 1)]);
                    dp_point(s_r[0], dp_column(iv - off, sp, c), c, v[2 * (m - 1) + 1], ph[2 * (m - 1) + 1]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < F; ++q) {  // seed lon_grid[0,0] of each probe's mesh (OGG:463)
            double di = 0.0;
            if (!JGROUP) {
                const double off = (double)(q / 2 + 1) * p.eps;
                di = (q & 1) ? -off : off;
            }
            seed[q] = p.lon0 + ((i_first + di) * 360.0) / (double)p.ni;
        }
        const unsigned st = unwrap_states<F>(v, active, g == 0, seed, s_carry_v, &s_carry_state, s_u);
        if (active && tid == SW_TX - 1) {
#pragma unroll
            for (int q = 0; q < F; ++q) s_carry_v[q] = v[q];
            s_carry_state = st;
        }
        if (active) {
            double ds[(H > 0) ? H : 1];
#pragma unroll
            for (int m = 0; m < H; ++m) {
                const double x0 = ((st >> (2 * m)) & 1u) ? v[2 * m] - 360 : v[2 * m];              // OGG:473
                const double x1 = ((st >> (2 * m + 1)) & 1u) ? v[2 * m + 1] - 360 : v[2 * m + 1];
                ds[m] = great_arc(x0, ph[2 * m], x1, ph[2 * m + 1]);
            }
            out[row * p.n_cols + g] = central_difference<F>(ds, reps);
        }
    }
}


// ---- chord form of the finite-difference scale factors -----------------------------------------------------------------
// The reference differentiates great-arc distances numerically (OGG:535-562): h = (8 ds(eps) - ds(2 eps)) / (12 eps), where
// ds is the haversine distance between two projected points that are ~2e-6 rad apart, formed from longitudes/latitudes in
// degrees.  That subtraction of two O(1) angles loses 10 digits: against an 80-bit evaluation the reference's own h is
// accurate to 2e-9 (relative).  Here the SAME stencil (the same 4H probe points, the same w = conformal image of each probe)
// is kept, but the distance between two probes is taken from their positions on the sphere -- from the gnomonic images
// pa, pb = (X, Y, -1): sin(ds) = |pa x pb| / (|pa| |pb|), see gnomonic_arc -- : no atan2 / atan / hypot per probe, no
// sin/cos/asin per pair, and no longitude at all -- hence no 360-degree unwrap and no sequential scan.  Its h is accurate to 8e-10 and differs from the reference's by 1.6e-9, i.e. by
// less than the reference's own rounding error (tests/test_gpu_parity.py bounds it at 5e-7 like the literal path;
// OGG_DP_LITERAL=1 selects the literal kernels, which numerical_hi / numerical_hj / great_arc_distance always use).
struct ChordParams {
    long ni, nj;
    double lon0, lat0, lam_pole, r_pole, eps;
    long n_cols, n_rows;   // unique lattice columns / rows
    int lattice_M;
    long row0_cell;
    QuadNodes q;
    double* row_tab;       // [NV][n_rows]: gnomonic radius of row variants (base, +eps, -eps, +2eps, -2eps)
    double* col_tab;       // [NV][2][n_cols]: e' of column variants
    double* h_i;
    double* h_j;
};

template <int F>
__global__ void dpole_chord_tables_kernel(ChordParams p) {
    constexpr int NV = F + 1;
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    SweepParams sp{};
    sp.ni = p.ni, sp.nj = p.nj, sp.lon0 = p.lon0, sp.lat0 = p.lat0, sp.lam_pole = p.lam_pole, sp.r_pole = p.r_pole;
    const DpConst c = dp_const(sp);
    const int M = p.lattice_M;
    if (k < p.n_rows * NV) {
        const long row = k / NV;
        const int var = (int)(k % NV);
        double jv = lattice_node(p.q, (int)(row % M), p.row0_cell + row / M);
        if (var > 0) {
            const double off = (double)((var + 1) / 2) * p.eps;
            jv = (var & 1) ? jv + off : jv - off;
        }
        p.row_tab[var * p.n_rows + row] = dp_row_radius(jv, sp, c);
    } else if (k < p.n_rows * NV + p.n_cols * NV) {
        const long kk = k - p.n_rows * NV;
        const long col = kk / NV;
        const int var = (int)(kk % NV);
        double iv = lattice_node(p.q, (int)(col % M), col / M);
        if (var > 0) {
            const double off = (double)((var + 1) / 2) * p.eps;
            iv = (var & 1) ? iv + off : iv - off;
        }
        const cplx ep = dp_column(iv, sp, c);
        p.col_tab[(var * 2 + 0) * p.n_cols + col] = ep.re;
        p.col_tab[(var * 2 + 1) * p.n_cols + col] = ep.im;
    }
}

struct Gno {
    double X, Y;   // gnomonic image (plane tangent at the south pole) of a point of the sphere
};

// w * r_joint, w the conformal image of the probe (OGG:454-455), with the complex quotient formed from one Newton reciprocal
// of |den|^2 (<= 2 ulp): the point of the sphere is (X, Y, -1) / sqrt(1 + X^2 + Y^2).
OGG_DEV Gno dp_gnomonic(double r, cplx ep, const DpConst& c) {
    const double zr = r * ep.re, zi = r * ep.im;
    const double nr = zr + c.z0r, ni = zi + c.z0i;
    const double dr = 1 + fma(c.z0r, zr, c.z0i * zi);      // 1 + conj(z0) z
    const double di = fma(c.z0r, zi, -(c.z0i * zr));
    const double s = rcp_nr(fma(dr, dr, di * di)) * c.r_joint;
    return Gno{fma(nr, dr, ni * di) * s, fma(ni, dr, -(nr * di)) * s};
}

// Great-arc distance of two nearby points from their gnomonic images a, b: with p = (X, Y, -1),
//   sin(theta) = |pa x pb| / (|pa| |pb|),   |pa x pb|^2 = dX^2 + dY^2 + (Xa dY - Ya dX)^2,   |p|^2 = 1 + X^2 + Y^2
// -- the differences dX, dY are formed once, nothing is normalised per point, and theta = asin(sin theta) from three terms of
// the series (the probes of the eps = 1e-3 stencil are ~1e-6 rad apart; exact to 1e-30 below 1e-3).
OGG_DEV double gnomonic_arc(const Gno& a, const Gno& b) {
    const double dX = b.X - a.X, dY = b.Y - a.Y;
    const double cr = fma(a.X, dY, -(a.Y * dX));
    const double s2 = fma(dX, dX, fma(dY, dY, cr * cr));
    if (s2 == 0.0) return 0.0;
    const double na = fma(a.X, a.X, fma(a.Y, a.Y, 1.0)), nb = fma(b.X, b.X, fma(b.Y, b.Y, 1.0));
    const double q = s2 * rcp_nr(na * nb);            // sin^2(theta)
    const double sn = sqrt_nr(q);
    return (sn < 1e-3) ? sn * (1.0 + q * (1.0 / 6.0 + q * (3.0 / 40.0))) : asin(sn);
}

template <int F>
__global__ __launch_bounds__(256) void dpole_chord_h_kernel(ChordParams p) {
    constexpr int H = F / 2, NV = F + 1;
    const long col = (long)blockIdx.x * 256 + threadIdx.x;
    const long row = blockIdx.y;
    if (col >= p.n_cols) return;
    SweepParams sp{};
    sp.ni = p.ni, sp.nj = p.nj, sp.lon0 = p.lon0, sp.lat0 = p.lat0, sp.lam_pole = p.lam_pole, sp.r_pole = p.r_pole;
    const DpConst c = dp_const(sp);
    const double reps = 1.0 / p.eps;
    double r[NV];
    cplx ep[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        r[v] = p.row_tab[v * p.n_rows + row];  // wave-uniform
        ep[v] = cplx{p.col_tab[(v * 2 + 0) * p.n_cols + col], p.col_tab[(v * 2 + 1) * p.n_cols + col]};
    }
    double dsi[(H > 0) ? H : 1], dsj[(H > 0) ? H : 1];
#pragma unroll
    for (int m = 1; m <= H; ++m) {
        // OGG:538,541: ds(j, i + m eps, j, i - m eps);  OGG:553,556: ds(j + m eps, i, j - m eps, i)
        dsi[m - 1] = gnomonic_arc(dp_gnomonic(r[0], ep[2 * m - 1], c), dp_gnomonic(r[0], ep[2 * m], c));
        dsj[m - 1] = gnomonic_arc(dp_gnomonic(r[2 * m - 1], ep[0], c), dp_gnomonic(r[2 * m], ep[0], c));
    }
    p.h_i[row * p.n_cols + col] = central_difference<F>(dsi, reps);
    if (p.h_j) p.h_j[row * p.n_cols + col] = central_difference<F>(dsj, reps);
}

// ---- Lobatto quadrature of the lattice values, one thread per cell, reference summation order (OGG:585-599) --------
struct ReduceParams {
    long nx, n_cols;       // cells per row; lattice columns (M*nx + 1)
    long n_cell_rows;      // cell rows of the band
    long n_dx_rows;        // n_cell_rows, or n_cell_rows + 1 for the band that owns dxq[ny]
    double Re;
    const double* h_i;
    const double* h_j;
    double* dxq;
    double* dyq;
    double* daq;
};

template <int N>
__global__ __launch_bounds__(256) void dpole_quad_reduce_kernel(ReduceParams p) {
    constexpr int M = N - 1;
    const long ci = (long)blockIdx.x * 256 + threadIdx.x;
    const long cj = blockIdx.y;
    if (ci >= p.nx) return;
    const double* hi0 = p.h_i + (M * cj) * p.n_cols + M * ci;
    double yv[N];
#pragma unroll
    for (int ii = 0; ii < N; ++ii) yv[ii] = hi0[ii];
    p.dxq[cj * p.nx + ci] = qavg_1d<N>(yv) * p.Re;                              // OGG:594,598
    if (cj >= p.n_cell_rows) return;                                          // the dxq[ny] row has no cells
    const double* hj0 = p.h_j + (M * cj) * p.n_cols + M * ci;
#pragma unroll
    for (int jj = 0; jj < N; ++jj) yv[jj] = hj0[jj * p.n_cols];
    p.dyq[cj * (p.nx + 1) + ci] = qavg_1d<N>(yv) * p.Re;                        // OGG:595,599
    if (ci == p.nx - 1) {
#pragma unroll
        for (int jj = 0; jj < N; ++jj) yv[jj] = hj0[jj * p.n_cols + M];
        p.dyq[cj * (p.nx + 1) + p.nx] = qavg_1d<N>(yv) * p.Re;
    }
    double da;
    if (N == 2) {
        const double d = 1.0 / 2.0;
        da = d * d * (hi0[0] * hj0[0] + hi0[1] * hj0[1] + hi0[p.n_cols] * hj0[p.n_cols] + hi0[p.n_cols + 1] * hj0[p.n_cols + 1]);
    } else {
        const double w4[4] = {1.0, 5.0, 5.0, 1.0};
        const double d = 1.0 / 12.0;
        double ysum = 0.0;
#pragma unroll
        for (int jj = 0; jj < N; ++jj)
#pragma unroll
            for (int ii = 0; ii < N; ++ii)
                ysum = ysum + w4[ii & 3] * w4[jj & 3] * (hi0[jj * p.n_cols + ii] * hj0[jj * p.n_cols + ii]);   // OGG:589,244
        da = d * d * ysum;
    }
    p.daq[cj * p.nx + ci] = da * p.Re * p.Re;                                   // OGG:597
}

// ---- displacedPoleCap_projection on explicit 2-D lon/lat grids (OGG:447-467) and bare monotonic_bounding ---------
struct DirectParams {
    long nj, ni;
    const double* lon;  // PROJECT: lon_grid; UNWRAP: unused
    const double* lat;
    double z0r, z0i, r_joint, x0;
    double* lam;        // PROJECT: out; UNWRAP: in/out
    double* phi;
};

template <bool PROJECT>
__global__ __launch_bounds__(SW_TX) void dpole_direct_kernel(DirectParams p) {
    __shared__ UnwrapShared<1> s_u;
    __shared__ double s_carry_v[1];
    __shared__ unsigned s_carry_state;
    const int tid = threadIdx.x;
    const long row = blockIdx.x;
    const DpConst c = {p.z0r, p.z0i, p.r_joint};
    for (long c0 = 0; c0 < p.ni; c0 += SW_TX) {
        const long g = c0 + tid;
        const bool active = g < p.ni;
        double v[1] = {0.0}, ph = 0.0;
        if (active) {
            if (PROJECT) {
                const double lon = p.lon[row * p.ni + g], lat = p.lat[row * p.ni + g];
                const double r = tan((90 + lat) * kPi180) / p.r_joint;  // OGG:448
                double s, co;
                sincos(lon * kPi180, &s, &co);
                const cplx e = {co, s};
                const cplx num = {e.re - c.z0r, e.im - c.z0i};
                const cplx cz = cmul(cplx{c.z0r, -c.z0i}, e);
                const cplx den = {1.0 - cz.re, 0.0 - cz.im};
                dp_point(r, cdiv(num, den), c, v[0], ph);
            } else {
                v[0] = p.lam[row * p.ni + g];
            }
        }
        const double seed[1] = {p.x0};
        const unsigned st = unwrap_states<1>(v, active, g == 0, seed, s_carry_v, &s_carry_state, s_u);
        if (active && tid == SW_TX - 1) {
            s_carry_v[0] = v[0];
            s_carry_state = st;
        }
        if (active) {
            p.lam[row * p.ni + g] = (st & 1u) ? v[0] - 360 : v[0];
            if (PROJECT) p.phi[row * p.ni + g] = ph;
        }
    }
}

int launch_mesh(const SweepParams& p, long rows, hipStream_t s) {
    if (rows <= 0) return OGG_OK;
    dpole_eval_kernel<<<dim3((unsigned)((p.n_cols + SW_TX - 1) / SW_TX), (unsigned)rows), SW_TX, 0, s>>>(p);
    OGG_LAUNCH_CHECK();
    const int cpt = (int)((p.n_cols + 1023) / 1024);
    dpole_unwrap_kernel<1024><<<(unsigned)rows, 1024, 0, s>>>(p, cpt);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}

int check_cap(long ni, long nj) {
    OGG_REQUIRE(ni > 0 && nj > 0, OGG_ESHAPE, "displaced pole cap: ni=%ld nj=%ld", ni, nj);
    return OGG_OK;
}

}  // namespace

extern "C" {

int ogg_displaced_pole_mesh_dev(long n_i, const double* i, long n_j, const double* j, long ni, long nj, double lon0,
                                double lat0, double lam_pole, double r_pole, double* lams, double* phis, void* stream) {
    if (int e = check_cap(ni, nj)) return e;
    OGG_REQUIRE(n_i > 0 && n_j >= 0 && i && j && lams && phis, OGG_EARG, "ogg_displaced_pole_mesh: bad argument");
    SweepParams p{};
    p.ni = ni, p.nj = nj, p.lon0 = lon0, p.lat0 = lat0, p.lam_pole = lam_pole, p.r_pole = r_pole;
    p.n_cols = n_i, p.i_arr = i, p.j_arr = j;
    p.out0 = lams, p.out1 = phis;
    return launch_mesh(p, n_j, ogg::as_stream(stream));
}

int ogg_displaced_pole_projection_dev(long nj, long ni, const double* lon_grid, const double* lat_grid, double z0_re,
                                      double z0_im, double r_joint, double x_0, double* lam, double* phi, void* stream) {
    OGG_REQUIRE(nj >= 0 && ni > 0 && lon_grid && lat_grid && lam && phi, OGG_EARG, "ogg_displaced_pole_projection: bad argument");
    if (nj == 0) return OGG_OK;
    DirectParams p{nj, ni, lon_grid, lat_grid, z0_re, z0_im, r_joint, x_0, lam, phi};
    dpole_direct_kernel<true><<<(unsigned)nj, SW_TX, 0, ogg::as_stream(stream)>>>(p);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}

int ogg_monotonic_bounding_dev(long nj, long ni, double* x, double x_0, void* stream) {
    OGG_REQUIRE(nj >= 0 && ni > 0 && x, OGG_EARG, "ogg_monotonic_bounding: bad argument");
    if (nj == 0) return OGG_OK;
    DirectParams p{nj, ni, nullptr, nullptr, 0.0, 0.0, 1.0, x_0, x, nullptr};
    dpole_direct_kernel<false><<<(unsigned)nj, SW_TX, 0, ogg::as_stream(stream)>>>(p);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}

int ogg_displaced_pole_grid_dev(long Ni, long Nj, double lon0, double lat0, double lon_dp, double r_dp, long j0, long nrows,
                                double* x, double* y, void* stream) {
    if (int e = check_cap(Ni, Nj)) return e;
    OGG_REQUIRE(j0 >= 0 && nrows >= 0 && j0 + nrows <= Nj + 1 && x && y, OGG_ESHAPE,
                "ogg_displaced_pole_grid: rows %ld..%ld outside 0..%ld", j0, j0 + nrows, Nj);
    SweepParams p{};
    p.ni = Ni, p.nj = Nj, p.lon0 = lon0, p.lat0 = lat0, p.lam_pole = lon_dp, p.r_pole = r_dp;
    p.n_cols = Ni + 1, p.i_arr = nullptr, p.j_arr = nullptr, p.j0 = j0;
    p.out0 = x, p.out1 = y;
    return launch_mesh(p, nrows, ogg::as_stream(stream));
}

int ogg_displaced_pole_numerical_h_dev(long n_i, const double* i, long n_j, const double* j, long nx, long ny, double lon0,
                                       double lat0, double lon_dp, double r_dp, double eps, int fd_order, double* h_i,
                                       double* h_j, void* stream) {
    if (int e = check_cap(nx, ny)) return e;
    OGG_REQUIRE(fd_order == 2 || fd_order == 4 || fd_order == 6, OGG_EORDER, "order not coded");
    OGG_REQUIRE(n_i > 0 && n_j >= 0 && i && j && (h_i || h_j) && eps > 0, OGG_EARG, "ogg_displaced_pole_numerical_h: bad argument");
    if (n_j == 0) return OGG_OK;
    HParams p{};
    p.ni = nx, p.nj = ny, p.lon0 = lon0, p.lat0 = lat0, p.lam_pole = lon_dp, p.r_pole = r_dp, p.eps = eps;
    p.n_cols = n_i, p.n_rows = n_j, p.i_arr = i, p.j_arr = j, p.lattice_M = 1;
    p.h_i = h_i, p.h_j = h_j;
    hipStream_t s = ogg::as_stream(stream);
#define OGG_LAUNCH_H(F)                                                                      \
    do {                                                                                     \
        if (h_i) dpole_h_kernel<F, false><<<(unsigned)n_j, SW_TX, 0, s>>>(p);                \
        if (h_j) dpole_h_kernel<F, true><<<(unsigned)n_j, SW_TX, 0, s>>>(p);                 \
    } while (0)
    if (fd_order == 2)
        OGG_LAUNCH_H(2);
    else if (fd_order == 4)
        OGG_LAUNCH_H(4);
    else
        OGG_LAUNCH_H(6);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}

long ogg_displaced_pole_quad_workspace_bytes(int order, long nx, long n_cell_rows) {
    if (order < 2 || order > 5 || nx <= 0 || n_cell_rows < 0) return 0;
    const long rows = (long)(order - 1) * n_cell_rows + 1, cols = (long)(order - 1) * nx + 1;
    return (2L * rows * cols + (long)(order + 1) * (rows + 2 * cols)) * (long)sizeof(double);
}

int ogg_displaced_pole_metrics_quad_ws_dev(int order, long nx, long ny, double lon0, double lat0, double lon_dp, double r_dp,
                                           double Re, long j0, long n_dx_rows, long n_cell_rows, double* dxq, double* dyq,
                                           double* daq, void* workspace, long workspace_bytes, void* stream) {
    OGG_REQUIRE(order >= 2 && order <= 5, OGG_EORDER, "Uncoded order");
    // the quadrature order is forwarded as the finite-difference order (OGG:583-584): 3 and 5 are "not coded" there
    OGG_REQUIRE(order == 2 || order == 4, OGG_EORDER, "order not coded");
    if (int e = check_cap(nx, ny)) return e;
    OGG_REQUIRE(dxq && (n_cell_rows <= 0 || (dyq && daq)), OGG_EARG, "ogg_displaced_pole_metrics_quad: null output");
    OGG_REQUIRE(j0 >= 0 && n_cell_rows >= 0 && j0 + n_cell_rows <= ny &&
                    (n_dx_rows == n_cell_rows || (n_dx_rows == n_cell_rows + 1 && j0 + n_cell_rows == ny)),
                OGG_ESHAPE, "ogg_displaced_pole_metrics_quad: band j0=%ld cell rows=%ld dx rows=%ld of ny=%ld", j0,
                n_cell_rows, n_dx_rows, ny);
    if (n_dx_rows == 0) return OGG_OK;
    const int M = order - 1;
    hipStream_t s = ogg::as_stream(stream);
    // lattice rows of the band: the unique Lobatto rows of its cell rows plus the closing row (which is dxq's row j0+n_cell_rows
    // when the band owns it, and the top edge of the last cell row otherwise)
    const long n_lat_rows = (long)M * n_cell_rows + 1, n_cols = (long)M * nx + 1;
    const size_t need = (size_t)ogg_displaced_pole_quad_workspace_bytes(order, nx, n_cell_rows);
    void* ws = workspace;
    if (workspace)
        OGG_REQUIRE((size_t)workspace_bytes >= need, OGG_EARG, "displaced-pole quadrature workspace too small: %ld < %zu bytes", workspace_bytes, need);
    else
        OGG_HIP_CHECK(hipMallocAsync(&ws, need, s));
    HParams h{};
    h.ni = nx, h.nj = ny, h.lon0 = lon0, h.lat0 = lat0, h.lam_pole = lon_dp, h.r_pole = r_dp, h.eps = 1e-3;  // OGG:583
    h.n_cols = n_cols, h.n_rows = n_lat_rows, h.lattice_M = M, h.row0_cell = j0;
    h.q = ogg::quad_nodes_host(order);
    h.h_i = static_cast<double*>(ws);
    h.h_j = h.h_i + n_lat_rows * n_cols;
    static const bool literal = getenv("OGG_DP_LITERAL") && atoi(getenv("OGG_DP_LITERAL")) != 0;
    if (literal) {  // haversine of unwrapped longitudes, operation for operation as OGG:522-562
        if (order == 2) {
            dpole_h_kernel<2, false><<<(unsigned)n_lat_rows, SW_TX, 0, s>>>(h);
            if (n_cell_rows > 0) dpole_h_kernel<2, true><<<(unsigned)n_lat_rows, SW_TX, 0, s>>>(h);
        } else {
            dpole_h_kernel<4, false><<<(unsigned)n_lat_rows, SW_TX, 0, s>>>(h);
            if (n_cell_rows > 0) dpole_h_kernel<4, true><<<(unsigned)n_lat_rows, SW_TX, 0, s>>>(h);
        }
    } else {  // same stencil, chord form of the distance (see dpole_chord_h_kernel)
        ChordParams cp{};
        cp.ni = nx, cp.nj = ny, cp.lon0 = lon0, cp.lat0 = lat0, cp.lam_pole = lon_dp, cp.r_pole = r_dp, cp.eps = 1e-3;
        cp.n_cols = n_cols, cp.n_rows = n_lat_rows, cp.lattice_M = M, cp.row0_cell = j0, cp.q = h.q;
        cp.row_tab = h.h_j + n_lat_rows * n_cols;
        cp.col_tab = cp.row_tab + (long)(order + 1) * n_lat_rows;
        cp.h_i = h.h_i;
        cp.h_j = (n_cell_rows > 0) ? h.h_j : nullptr;
        const long n_tab = (long)(order + 1) * (n_lat_rows + n_cols);
        dim3 grid((unsigned)((n_cols + 255) / 256), (unsigned)n_lat_rows);
        if (order == 2) {
            dpole_chord_tables_kernel<2><<<(unsigned)((n_tab + 255) / 256), 256, 0, s>>>(cp);
            dpole_chord_h_kernel<2><<<grid, 256, 0, s>>>(cp);
        } else {
            dpole_chord_tables_kernel<4><<<(unsigned)((n_tab + 255) / 256), 256, 0, s>>>(cp);
            dpole_chord_h_kernel<4><<<grid, 256, 0, s>>>(cp);
        }
    }
    OGG_LAUNCH_CHECK();
    ReduceParams r{nx, n_cols, n_cell_rows, n_dx_rows, Re, h.h_i, h.h_j, dxq, dyq, daq};
    dim3 grid((unsigned)((nx + 255) / 256), (unsigned)n_dx_rows);
    if (order == 2)
        dpole_quad_reduce_kernel<2><<<grid, 256, 0, s>>>(r);
    else
        dpole_quad_reduce_kernel<4><<<grid, 256, 0, s>>>(r);
    OGG_LAUNCH_CHECK();
    if (!workspace) OGG_HIP_CHECK(hipFreeAsync(ws, s));
    return OGG_OK;
}

int ogg_displaced_pole_metrics_quad_dev(int order, long nx, long ny, double lon0, double lat0, double lon_dp, double r_dp,
                                        double Re, long j0, long n_dx_rows, long n_cell_rows, double* dxq, double* dyq,
                                        double* daq, void* stream) {
    return ogg_displaced_pole_metrics_quad_ws_dev(order, nx, ny, lon0, lat0, lon_dp, r_dp, Re, j0, n_dx_rows, n_cell_rows, dxq, dyq,
                                                  daq, nullptr, 0, stream);
}

}  // extern "C"
