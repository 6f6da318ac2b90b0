// K3 / K4: Murray bipolar Arctic cap -- device code (included by ogg_bipolar.hip and by the fused pass, ogg_pass.hip).
//   bipolar_projection                 OGG:33-100   (element-wise kernel + the mesh builder OGG:103-122)
//   bipolar_cap_metrics_quad_fast      OGG:136-188  (+ bipolar_cap_ij_array OGG:125-133, quadrature OGG:191-255)
//
// K4 is fp64-VALU bound (an acos, a tan, an atan, a cos and two sqrt per lattice point; 24 B written per cell).
// The projection is split into its row-only part (5 libm calls per lattice row), its column-only part (sincos + fmod
// per lattice column) -- both tabulated once per call -- and the per-point remainder.  Lobatto nodes on shared cell
// edges are bit-identical in the reference (node n-1 of cell k == node 0 of cell k+1 == k+1 exactly), so a wave that walks
// a strip of cells evaluates every unique lattice point once -- (n-1)^2 instead of n^2 evaluations per cell -- and
// exchanges edge values by wave shuffles; sums follow the reference's order (OGG:216-221, 246-253).  The full lattice
// (138 M points at 1/8 degree) is never materialised.
#pragma once
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
    const double phig = 90 - div_pi180(2 * atan(tan(0.5 * (90 - phig_in) * kPi180) / rp));  // OGG:41
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
    phis = 90 - div_pi180(2 * atan(rpt));
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
// so the acos -> tan -> atan -> cos round trip of OGG:69-79 collapses to one division (returns the SQUARES of h_i_inv and
// of h_j_inv*N; the caller takes the roots and applies the per-index scale factors of OGG:131-132 as multiplications).  The identities are exact; the
// results differ from the literal sequence only by rounding: <= 1e-14 relative where the cap latitude is >= 1.4 degrees
// from the pole (measured against the oracle on the 1/8 degree lattice; 4e-15 at >= 4 degrees).  Nearer the pole the
// LITERAL sequence loses digits (phis = 90 - small is rounded to 1 ulp of 90 before the cosine), and parity with the
// reference means reproducing that: every point carries a guard (below) and the cells with a guarded point are
// re-evaluated with bp_point by a fix-up kernel (about 0.4 % of the cells at 1/8 degree, around the two pole points).
//
// Cleared of its quotients for the instruction count: with P = 1 + a b = 1/rden and q = P D,
//   h_j^2 N^2 = 4 rp^2 Xn / q^2,  Xn = (1-A^2) a(1-a) b(1+b) + (1-a) P = (1-a) + a(1-a) (b + (1-A^2) b(1+b))      (OGG:81-88)
//   h_i^2     = 4 rp^2 Yn / q^2,  Yn = (1-A^2) (1+b) + a b P                                                      (OGG:89-95)
//   D = (1 + rp^2) + (1 - rp^2) A
// so dx = 2 rp sqrt(Yn) / q = 2 rp Yn / sqrt(Yn q^2), dy likewise from Xn, and a point that only feeds the area sum --
// dx dy = 4 rp^2 sqrt(Xn Yn) / q^2 = 4 rp^2 Xn Yn / sqrt(Xn Yn q^4) -- costs ONE reciprocal square root and no reciprocal (Xn Yn q^4
// <= (1+b)^8-ish: in range for b < 1e38, and the exact j = ny row, b = inf, never takes this path).  The callers fold 4 rp^2 into
// their per-row scale factors.
template <bool GUARD>
OGG_DEV bool bp_point_fast(const BpRow& r, double b1, double bb1, const BpCol& c, double a1, double aa1, double rp2p, double rp2m,
                           double guard_kk, double& q, double& Xn, double& Yn) {
    // b1 = 1+b and bb1 = b (1+b) are row-only, a1 = 1-a and aa1 = a (1-a) are column-only: the callers hoist them;
    // rp2p = 1 + rp^2, rp2m = 1 - rp^2
    const double ab = c.alpha2 * r.beta2_inv;
    const double A = c.sinla * r.sphig;
    const double P = 1.0 + ab;                    // 1/rden
    const double D = fma(rp2m, A, rp2p);
    const double mp = fma(-A, A, 1.0);            // (1-A)(1+A), one rounding
    q = P * D;
    Xn = fma(aa1, fma(mp, bb1, r.beta2_inv), a1);
    Yn = fma(ab, P, mp * b1);
    // (the |beta2_inv| > HUGE case of OGG:86,94 only occurs on the exact j = ny row, which never takes this path)
    //
    // Exactness guard.  The literal sequence rounds phis = 90 - 2 atan(rp t)/PI_180 to a multiple of ulp(90) before taking
    // its cosine, which perturbs cos^2(phis) by up to 2.4e-16 / atan(rp t) relative; the algebraic value does not have that
    // perturbation, so where the cos^2 term carries weight w in h^2 the two differ by ~ w * 1.2e-16 / atan(rp t) in h.
    // With atan(u) >= u/(1+u^2) = sqrt(cc)/2 the point is handed to the literal fix-up when w^2 > K * cc, cc = cos^2(phis) =
    // 4 rp^2 (1-A)(1+A) / D^2, i.e. when (ui D)^2 > K 4 rp^2 (1-A)(1+A) Yn^2 (or the same with uj, Xn).  Measured against the oracle
    // on the top 100 cell rows of the 1/8 degree cap
    // (scripts/guard_k_probe.py), worst relative difference of dx / dy / area: every cell literal 5.8e-15 / 4.3e-15 / 3.5e-15
    // (ocml vs the host libm), K = 1000 5.8 / 6.1 / 5.0e-15, K = 4000 (default) 8.4 / 7.7 / 8.9e-15 with a quarter of the fix-up
    // cells, K = 16000 1.5e-14, no guard at all 6.8e-14 / 1.4e-13 / 6.9e-14 (area still within 3.3e-7 m^2).
    // Since w <= 1 a point can only be guarded where cos^2(phis) < 1/K, i.e. (phis <= grid latitude of the row) on the
    // lattice rows with cos^2(lat) < 1/K: the rows below that latitude run the GUARD = false instantiation.
    if (!GUARD) return false;
    const double kc = guard_kk * mp;                    // K 4 rp^2 (1-A)(1+A)
    const double uid = (mp * b1) * D, ujd = (mp * (aa1 * bb1)) * D;   // the cos^2(phis) terms of h_i^2, h_j^2 (up to the common factor), times D
    return (uid * uid > kc * (Yn * Yn)) || (ujd * ujd > kc * (Xn * Xn));
}

// asin(x) for 0 <= x <= 1, the SAME bits as ocml's asin (ROCm 7.2 device library, __ocml_asin_f64: one 12-coefficient polynomial in
// x^2 below 0.5 and in (1-x)/2 above, there with the square root and the reconstruction pi/2 - 2 (s + s p) in double-double) --
// restated operation for operation so that its coefficients can be scalar operands (horner_scalar: 12 instead of 36 vector instructions
// for the polynomial).  Bit-identity with asin() is a test (test_asin_unit_equals_library_asin: 4e7 arguments, the ends, the
// neighbours of 0.5), so the parity numbers of the mesh are those of ocml's asin.
__constant__ double kAsinPoly[12] = {0x1.5555555555380p-3, 0x1.333333336fd5bp-4, 0x1.6db6db41ce4bdp-5, 0x1.f1c72c668963fp-6,
                                     0x1.6e89f0a0adacfp-6, 0x1.1c6c111dccb70p-6, 0x1.c6fa84b77012bp-7, 0x1.8ed60a300c8d2p-7,
                                     0x1.ab3a098a70509p-8, 0x1.4052137024d6ap-6, -0x1.0a5a378a05eafp-6, 0x1.059859fea6a70p-5};

OGG_DEV double asin_unit(double x) {
    const bool hi = x >= 0.5;
    const double h = fma(x, -0.5, 0.5);          // (1 - x) / 2
    const double r = hi ? h : x * x;
    const double p = r * horner_scalar<12>(kAsinPoly, r);
    double v = fma(x, p, x);
    if (hi) {                                    // (a wave of the mesh almost always has such lanes)
        // s = sqrt(h): seed, one coupled step, one residual step
        const double y = __builtin_amdgcn_rsq(h);
        const double g0 = h * y, h0 = y * 0.5;
        const double e = fma(-h0, g0, 0.5);
        const double h1 = fma(h0, e, h0), g1 = fma(g0, e, g0);
        const double d = fma(-g1, g1, h);
        const double s0 = fma(d, h1, g1);
        const bool zero = h == 0.0;
        const double s = zero ? h : s0;
        // tail of the root: (h - s^2) / (2 s) in double-double
        const double ss = s * s;
        const double sse = fma(s, s, -ss);
        const double t0 = h - ss;
        const double t1 = ((h - t0) - ss) - sse;
        const double num = t0 + t1;
        const double den = s * 2.0;
        const double rc0 = __builtin_amdgcn_rcp(den);
        const double rc1 = fma(fma(-den, rc0, 1.0), rc0, rc0);
        const double rc = fma(fma(-den, rc1, 1.0), rc1, rc1);
        const double q0 = num * rc;
        const double q = fma(fma(-den, q0, num), rc, q0);
        const double c = zero ? 0.0 : q;
        const double sh = s + c;
        const double sl = c - (sh - s);
        // (sh + sl) * p, then + (sh + sl), in double-double
        const double ph = p * sh;
        const double pl = fma(sl, p, fma(sh, p, -ph));
        const double a = ph + pl;
        const double al = pl - (a - ph);
        const double b = sh + a;
        const double bl = a - (b - sh);
        const double w = (sl + al) + bl;
        const double u = b + w;
        const double ul = w - (u - b);
        // pi/4 - (u + ul), doubled
        constexpr double pio4 = 0x1.921fb54442d18p-1, pio4_lo = 0x1.1a62633145c07p-55;
        const double z = pio4 - u;
        const double zl = (((pio4 - z) - u) + pio4_lo) - ul;
        const double res = z + zl;
        v = (x == 1.0) ? 0x1.921fb54442d18p+0 : res + res;
    }
    return v;
}

// lamc of OGG:50-53 (degrees, before the root selection of OGG:58-63)
OGG_DEV double bp_lamc(const BpRow& r, const BpCol& c, double sqrt_rden) {   // sqrt_rden: sqrt(rden), IEEE
    double B = c.sinla * sqrt_rden;
    if (fabs(r.beta2_inv) > kHuge) B = 0.0;
    return div_pi180(asin_unit(B));   // B in [0, 1]
}

// lams of OGG:58-64 from lamc
OGG_DEV double bp_lams_select(double lamc, double lamg, double lon_bp) {
    const double dl = lamg - lon_bp;
    if ((dl > 90) && (dl <= 180)) lamc = 180 - lamc;
    if ((dl > 180) && (dl <= 270)) lamc = 180 + lamc;
    if (dl > 270) lamc = 360 - lamc;
    if (dl == 90) lamc = 90;
    if (dl == 270) lamc = 270;
    return lamc + lon_bp;
}

// lams of OGG:50-64
OGG_DEV double bp_lams(const BpRow& r, const BpCol& c, double sqrt_rden, double lamg, double lon_bp) {
    return bp_lams_select(bp_lamc(r, c, sqrt_rden), lamg, lon_bp);
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
    if (lams) lams[k] = bp_lams(r, c, sqrt(rden), lg, lon_bp);
    if (phis) phis[k] = ps;
    if (hi) hi[k] = h_i;
    if (hj) hj[k] = h_j;
}

// ---- the two inverse tangents of the mesh kernel -----------------------------------------------------------------------------------
// ocml's atan and atan2 spend two thirds of their instructions on what this kernel never feeds them (infinities, NaNs, signed zeros,
// |x| > 1, range selection): 83 and 114 instructions for 35 and 44 of fp64 arithmetic.
//
// Series coefficients through scalar registers (ogg_math.h, horner_scalar): a Horner step is one vector instruction, not three.
__constant__ double kAtanOdd[18] = {1.0 / 3.0,  -1.0 / 5.0,  1.0 / 7.0,  -1.0 / 9.0,  1.0 / 11.0, -1.0 / 13.0, 1.0 / 15.0, -1.0 / 17.0, 1.0 / 19.0,
                                    -1.0 / 21.0, 1.0 / 23.0, -1.0 / 25.0, 1.0 / 27.0, -1.0 / 29.0, 1.0 / 31.0, -1.0 / 33.0, 1.0 / 35.0, 0.0};

// r - r^3 (1/3 - r^2/5 + ...): the odd Taylor series of atan with TERMS coefficients
template <int TERMS>
OGG_DEV double atan_series(double r) {
    const double z = r * r;
    return fma(-(r * z), horner_scalar<TERMS>(kAtanOdd, z), r);
}

// atan(u) for 0 <= u <= 0.3 -- u = rp tan(chi/2) <= rp = tan(13 degrees) = 0.23 for every cap main() builds -- from 14 terms of the odd
// Taylor series in Horner form (the first omitted term is below 5e-18 relative): 0.57 ulp at worst (validated on the host against
// atanl on 3e7 arguments; glibc: 0.52).  A wave with any larger argument (a cap that starts south of 56.6 degrees) takes ocml's atan,
// behind one ballot.
OGG_DEV double atan_cap(double u) {
    if (__builtin_expect(__ballot(!(u <= 0.3)) != 0ull, 0)) return atan(u);
    return atan_series<14>(u);
}

// atan2(y, x) for the fused angle_x (finite arguments): octant reduction with ONE reciprocal -- r = (mn - mx) / (mn + mx) when
// mn / mx > tan(pi/8), mn / mx otherwise, so |r| <= tan(pi/8) -- and 17 terms of the odd series: 6e-16 rad at worst (host, 3e7
// arguments), which is 3e-14 degrees where the fused angle is held to 1e-10 (it already takes cos(phi) algebraically).  atan2(0, 0) = 0
// like numpy's arctan2(+0, +0).
OGG_DEV double atan2_angle(double y, double x) {
    const double ax = fabs(x), ay = fabs(y);
    const double mx = fmax(ax, ay), mn = fmin(ax, ay);
    const bool big = mn > 0.41421356237309503 * mx;
    const double num = big ? mn - mx : mn, den = big ? mn + mx : mx;
    double a = atan_series<17>(num * rcp_c3(den));
    a = big ? 0.78539816339744830962 + a : a;          // atan(mn / mx) in [0, pi/4]
    a = (ay > ax) ? 1.57079632679489661923 - a : a;
    a = (x < 0.0) ? 3.14159265358979323846 - a : a;
    a = (mx == 0.0) ? 0.0 : a;
    return copysign(a, y);
}

// ---- mesh builder (OGG:103-122) fused with angle_x (OGG:719-729), rows j0 .. j0+nrows-1 ------------------------
// A wave owns 62 output columns plus one halo column on either side, so the i-1 / i+1 neighbours that angle_x needs
// come from wave shuffles and every lane does the same work.  Row-only factors are computed once per row of the
// workgroup (LDS), column-only factors once per lane.  For the coordinates, tan(acos(A)/2) is evaluated as
// sqrt((1-A)/(1+A)) (exact identity; phis then differs from the literal sequence by at most 1 ulp of 90 degrees, 1.4e-14);
// the scale factors, when requested, follow the literal sequence (bp_point).
constexpr int MESH_WAVES = 4;
constexpr int MESH_OUT = 62;   // output columns per wave
constexpr int MESH_ROWS = 8;       // rows per workgroup: the default (OGG_MESH_ROWS: 1 .. MESH_ROWS_MAX)
constexpr int MESH_ROWS_MAX = 32;  // LDS entries for the row factors

// The columns a mesh launch evaluates: up to four runs of consecutive columns, each a whole number of wave blocks (MESH_OUT columns).
// Without symmetry one run, 0 .. Ni.  With it (the same symmetry as QuadCols, for lams, phis and angle_dx: with lamc the longitude
// before the root selection of OGG:58-63, the images of column i are (180 - lamc) + lon_bp at Ni/2 - i, (180 + lamc) + lon_bp at
// Ni/2 + i, (360 - lamc) + lon_bp at Ni - i -- the reference's own last operations on ITS lamc --, phis is the same, angle_dx changes
// sign under a mirror image and not under the half turn) the columns zf < i < Ni/4 - zm of run 0 = [0, Ni/4 + zm] also write their
// three images; evaluated at their own columns: [0, zf] and its images (the end columns i = 0, Ni take one-sided differences in
// angle_x, OGG:726-727, their images Ni/2 -+ 0 do not), and zm columns either side of the pole meridians Ni/4, 3Ni/4, where asin(B),
// B -> 1, amplifies the last-bit differences between the reference's own columns to 4.9e-12 degrees in x one column from the meridian
// (1/8 degree; 7.2e-12 at 1/16), falling as 1 / distance: 3.4e-13 two degrees away, which is where the zone ends (mesh_cols).
struct MeshCols {
    long lo[4], hi[4];   // columns [lo, hi] of run r
    long w_end[4];       // wave blocks of runs 0 .. r
    long zf, m_hi;       // images for zf < i < m_hi
    int sym;
};

struct MeshLane {
    long i;          // this lane's column (halo lanes beyond the row ends are clamped)
    bool active;     // the wave block exists (wave-uniform)
    bool out;        // writes column i ...
    bool img;        // ... and its three images Ni/2 - i, Ni/2 + i, Ni - i
};

OGG_HD MeshLane mesh_lane(const MeshCols& mc, long Ni, long w, int lane, int out_cols) {   // w: wave block of the launch's column space
    MeshLane m{};
    m.active = w < mc.w_end[3];
    const int run = (w < mc.w_end[0]) ? 0 : ((w < mc.w_end[1]) ? 1 : ((w < mc.w_end[2]) ? 2 : 3));
    const long w0 = (run == 0) ? 0 : mc.w_end[run - 1];
    const long col0 = mc.lo[run] + (w - w0) * out_cols;   // first output column of this wave
    long i = col0 - 1 + lane;
    m.out = m.active && (lane >= 1) && (lane <= out_cols) && (i <= mc.hi[run]);
    m.i = i < 0 ? 0 : (i > Ni ? Ni : i);
    m.img = mc.sym && run == 0 && m.out && m.i > mc.zf && m.i < mc.m_hi;
    return m;
}

struct MeshParams {
    long Ni, Nj;
    double lat0_bp, lon_bp;
    long j0, nrows;
    double *lams, *phis, *hi, *hj, *angle;
    int rows_per_wg;   // 1..MESH_ROWS_MAX
    MeshCols cols;
};

// workgroup (bx, by) of the mesh grid; s_row: MESH_ROWS_MAX entries of LDS.  WITH_H = false leaves the scale factors out of the
// code (the pass never asks for them; with them the kernel needs 183 VGPRs = 2 waves per SIMD).
template <bool WITH_H>
OGG_DEV void bipolar_mesh_body(const MeshParams& m, BpRow* s_row, long bx, long by) {
    const long Ni = m.Ni, Nj = m.Nj, j0 = m.j0, nrows = m.nrows;
    const double lat0_bp = m.lat0_bp, lon_bp = m.lon_bp;
    double* __restrict__ lams = m.lams;
    double* __restrict__ phis = m.phis;
    double* __restrict__ hi = m.hi;
    double* __restrict__ hj = m.hj;
    double* __restrict__ angle = m.angle;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long jl0 = by * m.rows_per_wg;
    const int nr = (int)((nrows - jl0 < m.rows_per_wg) ? (nrows - jl0) : m.rows_per_wg);
    const double rp = tan(0.5 * (90 - lat0_bp) * kPi180);                              // OGG:117
    if (tid < nr) {
        const long j = j0 + jl0 + tid;
        const double phig = lat0_bp + ((double)j * (90 - lat0_bp)) / (double)Nj;       // OGG:115
        s_row[tid] = bp_row(phig, rp);
    }
    __syncthreads();
    const MeshLane ml = mesh_lane(m.cols, Ni, bx * MESH_WAVES + wave, lane, MESH_OUT);
    if (!ml.active) return;                                                // wave-uniform
    const long i = ml.i;
    const bool out = ml.out, img = ml.img;                                 // img: this column also writes its three images
    const long h2 = Ni / 2;
    const double lamg = lon_bp + ((double)i * 360.0) / (double)Ni;                     // OGG:113
    const BpCol c = bp_col(lamg, lon_bp);
    const long ni1 = Ni + 1;
    for (int r = 0; r < nr; ++r) {
        const BpRow row = s_row[r];
        const long jl = jl0 + r, j = j0 + jl;
        // OGG:47, and the square root of OGG:50: IEEE division and root -- the same bits -- without the scaling and special-case steps
        // that 1 <= 1 + a b <= 1e33 does not need (a in [0, 1]; b = tan^2 of a latitude that is at most 90 degrees ROUNDED: <= 2.7e32)
        const double rden = rcp_ieee_normal(1.0 + c.alpha2 * row.beta2_inv);
        const double lamc = bp_lamc(row, c, sqrt_ieee_normal(rden));
        const double lam = bp_lams_select(lamc, lamg, lon_bp);
        const double A = c.sinla * row.sphig;
        const double m1 = 1 - A, p1 = 1 + A;
        const double t = (m1 > 0.0) ? m1 * rsqrt_c3(m1 * p1) : 0.0;                    // sqrt((1-A)/(1+A)) == tan(acos(A)/2), OGG:69-70
        const double u = rp * t;
        const double phi = 90 - div_pi180(2 * atan_cap(u));
        double* __restrict__ rl = lams + jl * ni1;
        double* __restrict__ rph = phis + jl * ni1;
        if (out) {
            rl[i] = lam;
            rph[i] = phi;
        }
        if (img) {                                                                     // OGG:58-64 at the image columns, from this lamc
            rl[h2 - i] = (180 - lamc) + lon_bp, rl[h2 + i] = (180 + lamc) + lon_bp, rl[Ni - i] = (360 - lamc) + lon_bp;
            rph[h2 - i] = phi, rph[h2 + i] = phi, rph[Ni - i] = phi;
        }
        if (WITH_H && (hi || hj)) {
            double ps, h_i, h_j, rd;
            bp_point(row, c, rp, ps, h_i, h_j, rd);
            if (out && hi && i < Ni) {
                const double v = h_i * 2 * kPi / (double)Ni;                                          // OGG:119
                hi[jl * Ni + i] = v;
                if (img) hi[jl * Ni + h2 - i] = v, hi[jl * Ni + h2 + i] = v, hi[jl * Ni + Ni - i] = v;
            }
            if (out && hj && j < Nj) {
                const double v = h_j * kPi180 * (90 - lat0_bp) / (double)Nj;                          // OGG:120
                hj[jl * ni1 + i] = v;
                if (img) hj[jl * ni1 + h2 - i] = v, hj[jl * ni1 + h2 + i] = v, hj[jl * ni1 + Ni - i] = v;
            }
        }
        if (angle) {                                                                   // OGG:725-728
            const double xl = wave_prev(lam), xr = wave_next(lam);
            const double yl = wave_prev(phi), yr = wave_next(phi);
            // cos(phi PI/180) = sin(2 atan u) = 2u/(1+u^2): differs from the cosine of the ROUNDED phi by < 7e-15/(90-phi) relative,
            // three orders below what the last-ulp differences of lam, phi do to their finite differences here
            const double cy = (2 * u) * rcp_c3(1 + u * u);
            double a;
            if (i == 0)
                a = atan2_angle(yr - phi, (xr - lam) * cy);
            else if (i == Ni)
                a = atan2_angle(phi - yl, (lam - xl) * cy);
            else
                a = atan2_angle(yr - yl, (xr - xl) * cy);
            a = div_pi180(a);
            double* __restrict__ ra = angle + jl * ni1;
            if (out) ra[i] = a;
            if (img) ra[h2 - i] = -a, ra[h2 + i] = a, ra[Ni - i] = -a;
        }
    }
}

template <bool WITH_H>
__global__ __launch_bounds__(64 * MESH_WAVES) void bipolar_mesh_kernel(MeshParams m) {
    __shared__ BpRow s_row[MESH_ROWS_MAX];
    bipolar_mesh_body<WITH_H>(m, s_row, blockIdx.x, blockIdx.y);
}

constexpr double BP_SYM_MESH_MERIDIAN_DEG = 2.0;   // MeshCols: degrees of longitude either side of a pole meridian evaluated at their own columns
constexpr long BP_SYM_MESH_FOLD_COLS = 2;          // ... and columns next to the fold lines i = 0, Ni/2, Ni

inline MeshCols mesh_cols(long Ni, int symmetry) {
    MeshCols c{};
    auto blocks = [](long lo, long hi) { return (hi - lo + 1 + MESH_OUT - 1) / MESH_OUT; };
    for (int r = 0; r < 4; ++r) c.lo[r] = 0, c.hi[r] = -1;
    c.lo[0] = 0, c.hi[0] = Ni;
    c.w_end[0] = c.w_end[1] = c.w_end[2] = c.w_end[3] = blocks(0, Ni);
    c.zf = Ni, c.m_hi = 0, c.sym = 0;
    if (!symmetry || Ni % 4 != 0) return c;
    double deg = BP_SYM_MESH_MERIDIAN_DEG;
    if (const char* e = getenv("OGG_BP_SYM_MESH_DEG")) deg = atof(e);
    const long q = Ni / 4, zf = BP_SYM_MESH_FOLD_COLS;
    long zm = (long)ceil(deg * (double)Ni / 360.0);
    if (zm < 2) zm = 2;
    if (zf + 1 >= q - zm) return c;   // nothing left to mirror
    c.sym = 1, c.zf = zf, c.m_hi = q - zm;
    c.lo[0] = 0, c.hi[0] = q + zm;
    c.lo[1] = 2 * q - zf, c.hi[1] = 2 * q + zf;
    c.lo[2] = 3 * q - zm, c.hi[2] = 3 * q + zm;
    c.lo[3] = Ni - zf, c.hi[3] = Ni;
    long w = 0;
    for (int r = 0; r < 4; ++r) w += blocks(c.lo[r], c.hi[r]), c.w_end[r] = w;
    return c;
}

// sets m.rows_per_wg and m.cols.  The column-only factors (sincos + two fmod) cost about as much as half a point, so a wave keeps
// MESH_ROWS rows even when the band is small: fewer rows per workgroup measured slower down to 1/8 of the 1/8 degree cap.
inline dim3 mesh_grid(MeshParams& m, int symmetry) {
    m.cols = mesh_cols(m.Ni, symmetry);
    const long n_waves = m.cols.w_end[3];
    const long gx = (n_waves + MESH_WAVES - 1) / MESH_WAVES;
    long rpw = MESH_ROWS;
    if (const char* e = getenv("OGG_MESH_ROWS")) rpw = atol(e) < 1 ? 1 : (atol(e) > MESH_ROWS_MAX ? MESH_ROWS_MAX : atol(e));
    m.rows_per_wg = (int)rpw;
    return dim3((unsigned)gx, (unsigned)((m.nrows + rpw - 1) / rpw));
}

constexpr double BP_GUARD_K_DEFAULT = 4000.0;  // exactness guard of bp_point_fast; OGG_BP_GUARD_K overrides

// ---- quadrature metrics -------------------------------------------------------------------------------------
struct QuadParams {
    long nx, ny;
    double lat0_bp, lon_bp, rp, Re;
    long j0;           // first cell row of the band (band-local output row 0)
    long top_out_row;  // band-local row of dxq that holds the j = ny row (the band that owns it), else -1
    double guard_k;       // exactness guard of bp_point_fast
    unsigned* ll_claims;  // QUAD_LL_CLAIM_WORDS words behind the fix-up counter: claim counters of the lat-lon strips of a fused pass
                          // (ogg_latlon_fused_dev.h), zeroed with the tables
    unsigned* fix_count;  // number of cells handed to the literal fix-up ...
    unsigned* fix_list;   // ... and their band-local linear indices out_r*nx + ci
    const BpRow* row_tab;  // [(N-1)*ny + 2]: unique lattice rows; entry (N-1)*ny is the ny-0.001 row, the last one is j = ny exactly
    const BpCol* col_tab;  // [(N-1)*nx + 1]: unique lattice columns
    double* dxq;
    double* dyq;
    double* daq;
    const double* top_src;   // NULL, or the j = ny row of dxq as the tables kernel left it in the workspace: the tail kernel copies it to
                             // dxq (a pass whose tables launch runs AHEAD of the stream must not write an output there: ogg_pass.hip)
    QuadNodes q;
    // constants of the algebraic per-point form, filled by plan_quad on the host (IEEE double, the operations a kernel would do) so
    // that they sit in scalar registers: sx = 2 pi / nx, sy = (90 - lat0) PI_180 / ny (OGG:131-132), 1 +- rp^2, 4 rp^2, 2 rp sx,
    // K 4 rp^2
    double sx, sy, rp2p, rp2m, rp2x4, cdx, guard_kk;
};

// The columns a grid of strip workgroups walks: up to three RUNS of consecutive cells, each a whole number of workgroups (QS_WAVES
// strips of QS_CELLS cells; the node column behind the last cell of a run -- it feeds dyq only -- is the halo lane of the run's last
// strip).  Without symmetry there is one run, the cells 0 .. nx-1.  With it (QuadCols::sym): the projection is mirror-symmetric about
// its two pole meridians (lattice columns nx/4, 3nx/4) and about the fold lines (columns 0, nx/2) -- sinla and alpha2 (OGG:42-45) are
// functions of min(d, 360 - d) and of sin / cos^2 of it --, so the cells [0, nx/4) determine the whole row: a lane that owns cell
// i >= z also writes its values to the cells nx/2-1-i, nx/2+i, nx-1-i (dyq, on node columns: i > z to nx/2-i, nx/2+i, nx-i).  The
// reference evaluates every column with its own roundings (lon_bp + i 360/Ni is rounded at magnitude 300 at one end of the row and
// at 60 at the other), and its own results are mirror images of each other only to ~5.7e-14 degrees / (distance from the fold line):
// the z cells next to each fold line (plan_quad: 6 degrees of longitude, where the reference's own asymmetry has fallen to 7e-15
// relative in area, 3e-15 in dx, 5e-15 in dy at 1/8 and 1/16 degree) are therefore evaluated at their own columns: run 1 =
// cells [nx/2-z, nx/2+z), run 2 = [nx-z, nx), and the cells [0, z) of run 0 have no images.
constexpr int QS_CELLS = 63;   // cells per wave strip (lane 63: the halo lane, the first cell of the next strip)
constexpr int QS_WAVES = 4;    // strips per workgroup
constexpr int QS_WG_CELLS = QS_CELLS * QS_WAVES;   // 252 cells = 2016 bytes of a row of one field per workgroup

struct QuadCols {
    long lo[3], hi[3];    // run r: cells [lo, hi), closed by node column hi
    long g_end[3];        // workgroups (along the columns) of runs 0 .. r
    long z;               // sym: cells [0, z) of run 0 (columns [0, z]) have no images
    int sym;
};

// what lane `lane` of wave `wave` of column workgroup `wg` owns in the column space `cs`
struct QuadLane {
    long ci;          // its cell = its node column (the cell's left edge)
    long c0;          // first cell of the workgroup
    int nv;           // cells of the workgroup inside its run (0 .. QS_WG_CELLS); workgroup-uniform
    int run;
    bool active;      // the wave has cells (wave-uniform)
    bool cell_lane;   // owns dxq, daq of cell ci and dyq of node column ci
    bool closing;     // the node column behind the run's last cell: dyq only (the halo lane of the run's last strip)
    bool img_cell;    // the cell also goes to its three mirror images: cells nx/2-1-ci, nx/2+ci, nx-1-ci
    bool img_col;     // the node column to its images: columns nx/2-ci, nx/2+ci, nx-ci
};

OGG_HD QuadLane quad_lane(const QuadCols& cs, long wg, int wave, int lane) {
    QuadLane q{};
    q.run = (wg < cs.g_end[0]) ? 0 : ((wg < cs.g_end[1]) ? 1 : 2);
    const long g0 = (q.run == 0) ? 0 : cs.g_end[q.run - 1];
    const long lo = cs.lo[q.run], hi = cs.hi[q.run];
    q.c0 = lo + (wg - g0) * QS_WG_CELLS;
    const long left = hi - q.c0;
    q.nv = (wg < cs.g_end[2] && left > 0) ? (int)(left < QS_WG_CELLS ? left : QS_WG_CELLS) : 0;
    const long cw = q.c0 + (long)wave * QS_CELLS;          // first cell of this wave
    q.active = q.nv > 0 && cw < hi;
    q.ci = cw + lane;
    q.cell_lane = q.active && lane < QS_CELLS && q.ci < hi;
    q.closing = q.active && lane <= QS_CELLS && q.ci == hi;
    q.img_cell = cs.sym && q.run == 0 && q.cell_lane && q.ci >= cs.z;
    q.img_col = cs.sym && q.run == 0 && (q.cell_lane || q.closing) && q.ci > cs.z;
    return q;
}

// The runs of one row a workgroup writes (element offsets within a row of dxq / daq, of dyq): its own nv cells from c0, and -- mirrored --
// the three images of its cells [zc, c0 + nv) and of its node columns [zp, c0 + nv).  Shared by the kernel and ogg_symmetry_coverage.
struct QuadStores {
    int n_ic, n_ip;         // cells / node columns with images
    int k0, k1;             // their first entry in the workgroup's row (cell zc - c0, column zp - c0)
    long cell_up, cell_dn2, cell_dn4;   // images of the cells: ascending from nx/2 + zc; from nx/2 - (c0+nv) and nx - (c0+nv), LAST cell first
    long col_up, col_dn2, col_dn4;      // of the node columns: from nx/2 + zp; from nx/2 - (c0+nv-1) and nx - (c0+nv-1), last column first
};

OGG_HD QuadStores quad_stores(const QuadCols& cs, const QuadLane& q, long nx, bool mirrored) {
    QuadStores s{};
    const bool sym = mirrored && cs.sym && q.run == 0;
    const long end = q.c0 + q.nv, h2 = nx / 2;
    const long zc = sym ? (cs.z > q.c0 ? cs.z : q.c0) : end;
    const long zp = sym ? (cs.z + 1 > q.c0 ? cs.z + 1 : q.c0) : end;
    s.n_ic = end > zc ? (int)(end - zc) : 0, s.n_ip = end > zp ? (int)(end - zp) : 0;
    s.k0 = (int)(zc - q.c0), s.k1 = (int)(zp - q.c0);
    s.cell_up = h2 + zc, s.cell_dn2 = h2 - end, s.cell_dn4 = nx - end;
    s.col_up = h2 + zp, s.col_dn2 = h2 - (end - 1), s.col_dn4 = nx - (end - 1);
    return s;
}

struct QuadRange {     // the part of the band one grid of strip workgroups evaluates
    long row_begin;    // cell rows [row_begin, row_end)
    long row_end;
    long rows_per_chunk;  // cell rows one wave walks (its first lattice row is recomputed: 1/((N-1)*rows_per_chunk) extra)
    unsigned gx, gy;   // workgroups along the columns (strips of cols / QS_WAVES), along the rows
    QuadCols cols;
};

// row-only and column-only parts of the projection for every unique lattice row / column of the cap (OGG:126-127,
// 41-46, 75-78), evaluated once per call instead of once per tile
template <int N>
OGG_DEV double quad_average_1d(const double* y) {  // OGG:207-222
    if (N == 2) return (1.0 / 2.0) * (y[0] + y[1]);
    if (N == 3) return (1.0 / 6.0) * (4.0 * y[1] + (y[0] + y[2]));
    if (N == 4) return (1.0 / 12.0) * (5.0 * (y[1] + y[2]) + (y[0] + y[3]));
    return (1.0 / 180.0) * (64.0 * y[2] + (49.0 * (y[1] + y[3])) + 9.0 * (y[0] + y[4]));
}

template <int N>
OGG_DEV double quad_weight_1d(int k) {  // w[k] of OGG:240 / 248, as selects (k may be a run-time value)
    if (N == 4) return (k == 0 || k == 3) ? 1.0 : 5.0;
    return (k == 0 || k == 4) ? 9.0 : ((k == 2) ? 64.0 : 49.0);
}

constexpr int QUAD_LL_CLAIM_WORDS = 2048;   // 2 per lat-lon workgroup of a pass: up to 1024 resident workgroups

template <int N>
__host__ __device__ inline unsigned tables_only_blocks(const QuadParams& p) {
    return (unsigned)(((N - 1) * p.ny + 2 + (N - 1) * p.nx + 1 + 255) / 256);
}

// dxq[ny][:] -- the exact j = ny lattice row (OGG:183 for the last row of dxq), literal sequence -- depends on nothing but the cap's
// parameters, so it rides with the tables (ONE lattice column per lane, the cell sums through wave shuffles) instead of waiting for
// the tail launch behind the quadrature, where its four-columns-per-lane chain of libm calls was the longest thing in that launch.
// A wave covers TOP_CELLS<N> cells = TOP_CELLS * (N-1) + 1 <= 64 lattice columns; 4 waves per workgroup.
template <int N>
constexpr int top_cells() { return 63 / (N - 1); }
template <int N>
inline unsigned top_row_blocks(const QuadParams& p) {
    if (p.top_out_row < 0) return 0u;
    const long waves = (p.nx + top_cells<N>() - 1) / top_cells<N>();
    return (unsigned)((waves + 3) / 4);
}

template <int N>
OGG_DEV void bipolar_top_row_body(const QuadParams& p, long w) {   // w: wave index
    constexpr int M = N - 1, CW = top_cells<N>();
    const int lane = threadIdx.x & 63;
    const long c0 = w * CW;
    if (c0 >= p.nx) return;   // wave-uniform
    long u = M * c0 + lane;   // lattice column of this lane (the table's index)
    if (u > M * p.nx) u = M * p.nx;
    const double iv = lattice_node(p.q, (int)(u % M), u / M);
    const double lon = p.lon_bp + (iv * 360.0) / (double)p.nx;                    // OGG:126
    const BpCol c = bp_col(lon, p.lon_bp);
    const double jv = lattice_node(p.q, 0, p.ny);                                 // first node of cell ny: j = ny exactly
    const double latg = p.lat0_bp + (jv * (90 - p.lat0_bp)) / (double)p.ny;       // OGG:127
    const BpRow r = bp_row(latg, p.rp);
    double phis, rden, h_i, h_j;
    bp_point(r, c, p.rp, phis, h_i, h_j, rden);
    const double dx = h_i * 2 * kPi / (double)p.nx;                               // OGG:131
    double y[N];
#pragma unroll
    for (int k = 0; k < N; ++k) y[k] = __shfl_down(dx, k);
    const int cell = lane / M;
    if (lane % M == 0 && cell < CW && c0 + cell < p.nx) p.dxq[p.top_out_row * p.nx + c0 + cell] = quad_average_1d<N>(y) * p.Re;
}

// row-only and column-only parts of the projection for every unique lattice row / column of the cap (OGG:126-127,
// 41-46, 75-78), evaluated once per call instead of once per tile; the workgroups behind them: the j = ny row of dxq
template <int N>
OGG_DEV void bipolar_tables_body(const QuadParams& p, long bx) {
    constexpr int M = N - 1;
    const long nb_tab = tables_only_blocks<N>(p);
    if (bx >= nb_tab) {
        bipolar_top_row_body<N>(p, (bx - nb_tab) * 4 + (threadIdx.x >> 6));
        return;
    }
    BpRow* row_tab = const_cast<BpRow*>(p.row_tab);
    BpCol* col_tab = const_cast<BpCol*>(p.col_tab);
    const long k = bx * blockDim.x + threadIdx.x;
    const long n_rows = M * p.ny + 2, n_cols = M * p.nx + 1;
    if (k == 0) *p.fix_count = 0u;  // the fix-up list of this call starts empty (this kernel precedes the quadrature kernels)
    for (long w = k; w < QUAD_LL_CLAIM_WORDS; w += nb_tab * blockDim.x) p.ll_claims[w] = 0u;   // (and nothing of the pass is claimed)
    if (k < n_rows) {
        double jv;
        if (k == M * p.ny)
            jv = (double)p.ny - 0.001;  // OGG:146-147: last node of cell ny-1
        else if (k == M * p.ny + 1)
            jv = lattice_node(p.q, 0, p.ny);  // first node of cell ny: j = ny exactly
        else
            jv = lattice_node(p.q, (int)(k % M), k / M);
        const double latg = p.lat0_bp + (jv * (90 - p.lat0_bp)) / (double)p.ny;   // OGG:127
        row_tab[k] = bp_row(latg, p.rp);
    } else if (k < n_rows + n_cols) {
        const long u = k - n_rows;
        const double iv = lattice_node(p.q, (int)(u % M), u / M);
        const double lon = p.lon_bp + (iv * 360.0) / (double)p.nx;               // OGG:126
        col_tab[u] = bp_col(lon, p.lon_bp);
    }
}

template <int N>
__global__ __launch_bounds__(256) void bipolar_tables_kernel(QuadParams p) {
    bipolar_tables_body<N>(p, blockIdx.x);
}

template <int N>
inline unsigned tables_blocks(const QuadParams& p) {   // table workgroups + the workgroups of the j = ny row
    return tables_only_blocks<N>(p) + top_row_blocks<N>(p);
}

// A wave owns a vertical strip of QS_CELLS = 63 cells (lane 63 is a halo lane: the first cell of the next strip) and walks up the
// cell rows of its chunk.  Each lane evaluates the (N-1) x (N-1) lattice points of its cell that are not on the cell's
// right or top edge; the right-edge values are the left-edge values of lane+1 (one wave shuffle), the top-edge row is the
// bottom row of the next cell row and is evaluated once and reused.  So every unique lattice point of the strip is
// evaluated exactly once -- (N-1)^2 instead of N^2 evaluations per cell -- with no LDS and no barrier; the row-only
// factors are wave-uniform (scalar loads), the column-only factors stay in registers for the whole walk.  Sums are taken
// in the reference's order (OGG:216-221, 244-253).
//
// The STORES go through LDS (round 5): what a wave holds is a 504-byte piece of a row that starts anywhere, and a store whose ends split
// 128-byte lines with the neighbouring waves writes at 5.3 TB/s where whole aligned lines reach 6.6 (scripts/microbench/mirror_writes.hip;
// with mirrored columns launch B is bound by its writes).  The four strips of a workgroup are 252 consecutive cells: every wave puts its
// row values into LDS and the workgroup writes the row -- and its three mirror images -- as 128-byte-aligned runs, wave w the elements
// [64 w, 64 w + 64) counted from the line boundary at or before the run (wg_store_aligned): two split lines per workgroup, row and field
// instead of eight.

// dst[j] = buf[j0 + sgn j], j = 0 .. n-1 (n <= QS_WG_CELLS), by the four waves of a workgroup: wave w takes the elements of the w-th
// 512 bytes counted from the 128-byte boundary at or before dst, so that every line inside the run is written whole by one wave (the
// up to 11 elements past the fourth 512 bytes -- 252 + 15 > 256 -- go with wave 0)
OGG_DEV void wg_store_aligned(double* __restrict__ dst, int n, const double* buf, int j0, int sgn, int wave, int lane) {
    const int mis = (int)((reinterpret_cast<unsigned long>(dst) >> 3) & 15ul);   // elements of dst[0] past the line boundary
    const int j = 64 * wave + lane - mis;
    if (j >= 0 && j < n) dst[j] = buf[j0 + sgn * j];
    const int j2 = 64 * QS_WAVES + lane - mis;
    if (wave == 0 && j2 < n) dst[j2] = buf[j0 + sgn * j2];
}

struct QuadOutLds {
    double v[2][3][QS_WG_CELLS];   // [buffer][dxq, dyq, daq][cell of the workgroup]: double-buffered, one barrier per cell row
};

template <int N, int MODE>
struct RowEval {
    double dx[N];  // dx at this lane's columns ii = 0..N-2 and, in [N-1], at the right edge (from lane+1); only if want_dx
    double dy0;    // dy at ii = 0 (the lane's left cell edge)
    double pr[N];  // dx*dy at the same N columns (OGG:178)
    int guarded;   // any of these N points failed the exactness guard (always 0 on the literal path)
};

constexpr int QM_FAST = 0, QM_GUARD = 1, QM_LITERAL = 2;  // per-point method of the strip kernel

template <int N, int MODE>
OGG_DEV void eval_lattice_row(const QuadParams& p, const BpRow& r, const BpCol* col, double rp2p, bool want_dx, RowEval<N, MODE>& o) {  // r: factors of the lattice row; r, want_dx wave-uniform
    // want_dx: the row is the bottom edge of a cell row, whose dx feeds dxq (OGG:183).  On the other rows the algebraic path
    // needs dx only inside the product dx*dy, which it then takes as sqrt(hi2*hj2) -- one square root instead of two.
    constexpr int M = N - 1;
    constexpr bool FAITHFUL = (MODE == QM_LITERAL);
    if (FAITHFUL) {
        // one point at a time (the libm calls of several points interleaved cost 225 VGPRs = 2 waves/SIMD); the column
        // entry and the result slot are picked with compare-selects so that no register array is indexed dynamically
#pragma unroll 1
        for (int ii = 0; ii < M; ++ii) {
            BpCol c = col[0];
#pragma unroll
            for (int k = 1; k < M; ++k) {
                c.sinla = (ii == k) ? col[k].sinla : c.sinla;
                c.alpha2 = (ii == k) ? col[k].alpha2 : c.alpha2;
            }
            double phis, rden, h_i, h_j;
            bp_point(r, c, p.rp, phis, h_i, h_j, rden);
            const double dx = h_i * 2 * kPi / (double)p.nx;                        // OGG:131
            const double dy = h_j * (90 - p.lat0_bp) * kPi180 / (double)p.ny;      // OGG:132
#pragma unroll
            for (int k = 0; k < M; ++k) {
                o.dx[k] = (ii == k) ? dx : o.dx[k];
                o.pr[k] = (ii == k) ? dx * dy : o.pr[k];                           // OGG:178
            }
            if (ii == 0) o.dy0 = dy;
        }
    } else {
        const double b1 = 1 + r.beta2_inv, bb1 = r.beta2_inv * b1, nsy = r.N_inv * p.sy;
        // 4 rp^2 folded into the scale factors of OGG:131-132: sqrt(4 rp^2) = 2 rp
        const double cdx = p.cdx, cdy = (2 * p.rp) * nsy, cpr = p.rp2x4 * (p.sx * nsy);
        const double guard_kk = p.guard_kk, rp2m = p.rp2m;   // rp2p: in a vector register (D = fma(rp2m, A, rp2p) may read ONE scalar operand)
        int g_first = 0, g_any = 0;
#pragma unroll
        for (int ii = 0; ii < M; ++ii) {
            double q, Xn, Yn;
            const double a1 = 1 - col[ii].alpha2;
            const bool g = bp_point_fast<MODE == QM_GUARD>(r, b1, bb1, col[ii], a1, col[ii].alpha2 * a1, rp2p, rp2m, guard_kk, q, Xn, Yn);
            if (ii == 0) g_first = g;
            g_any |= (int)g;
            const double q2 = q * q;
            // h_j vanishes identically on the meridians alpha2 == 1 (OGG:81-84), where Xn == 0 exactly: the argument of the reciprocal
            // square root is held above zero, so that Xn rsqrt(..) is 0 * finite = 0 there without a branch around the point
            constexpr double tiny = 1e-290;
            if (ii == 0 || want_dx) {
                const double dx = (Yn * rsqrt_c3(Yn * q2)) * cdx;
                const double dy = (Xn * rsqrt_c3(fmax(Xn * q2, tiny))) * cdy;
                o.dx[ii] = dx;
                o.pr[ii] = dx * dy;
                if (ii == 0) o.dy0 = dy;
            } else {
                const double xy = Xn * Yn;
                o.pr[ii] = (xy * rsqrt_c3(fmax(xy * (q2 * q2), tiny))) * cpr;
            }
        }
        o.guarded = 0;
        if (MODE == QM_GUARD) o.guarded = g_any | wave_next(g_first);  // the right edge is lane+1's first column
    }
    if (FAITHFUL) o.guarded = 0;
    if (FAITHFUL || want_dx) o.dx[M] = wave_next(o.dx[0]);
    o.pr[M] = wave_next(o.pr[0]);
}

template <int N, int MODE>
OGG_DEV void bipolar_quad_body(const QuadParams& p, const QuadRange& rg, QuadOutLds& out, long wgx, long by) {  // wgx: workgroup along the columns
    constexpr int M = N - 1;
    constexpr bool FAITHFUL = (MODE == QM_LITERAL);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const QuadLane ql = quad_lane(rg.cols, wgx, wave, lane);
    if (ql.nv == 0) return;  // workgroup-uniform: nothing of this workgroup lies inside a run
    // (a wave without cells -- the last workgroup of a run may be partly filled -- evaluates nothing but stays for the barriers and the stores)
    const bool active = ql.active;
    const long ci = ql.ci, c0 = ql.c0;
    const int nv = ql.nv;
    const long n_cols_tab = M * p.nx + 1;
    BpCol col[M];
#pragma unroll
    for (int ii = 0; ii < M; ++ii) {
        long u = M * ci + ii;
        if (u > n_cols_tab - 1) u = n_cols_tab - 1;
        col[ii] = p.col_tab[u];
    }
    const bool cell_lane = ql.cell_lane, closing = ql.closing;
    const bool img_col = (MODE == QM_FAST) && ql.img_col;
    const long h2 = p.nx / 2;
    const QuadStores qs = quad_stores(rg.cols, ql, p.nx, MODE == QM_FAST);   // the cells / node columns of this workgroup that have images
    double rp2p = p.rp2p;
    asm volatile("" : "+v"(rp2p));   // held in a vector register for the whole walk (else re-copied from its scalar register per point)
    RowEval<N, MODE> cur;
    const long r0 = rg.row_begin + by * rg.rows_per_chunk;
    const long r1 = (r0 + rg.rows_per_chunk < rg.row_end) ? r0 + rg.rows_per_chunk : rg.row_end;
    if (active) eval_lattice_row<N, MODE>(p, p.row_tab[(long)M * r0], col, rp2p, true, cur);
    const int slot = wave * QS_CELLS + lane;   // this lane's entry of the workgroup's row in LDS
    for (long c = r0; c < r1; ++c) {
        double dxq = 0.0, dyq = 0.0, da = 0.0;
        int guarded = 0;
        if (active) {
            dxq = quad_average_1d<N>(cur.dx) * p.Re;                        // OGG:183,186: bottom-edge row
            guarded = cur.guarded;                                          // bottom-edge row (carried)
            double dyc[N];                                                  // dy down this lane's left edge
            double y2[(N <= 3) ? N * N : 1];
            double ysum = 0.0;
#pragma unroll
            for (int k = 0; k < N; ++k) dyc[k] = 0.0;
#pragma unroll((N <= 3 || MODE != QM_LITERAL) ? N : 1)
            for (int jj = 0; jj < N; ++jj) {
                if (jj > 0) {
                    // the top row of this cell row is the bottom row of the next one.  Every cell-edge lattice row is evaluated in the
                    // edge form (dx and dy separately), also the last one of a chunk, whose dx nobody reads: the area of a cell must not
                    // depend on where the chunks -- hence the bands of a sharded run -- end
                    eval_lattice_row<N, MODE>(p, p.row_tab[(long)M * c + jj], col, rp2p, jj == M, cur);
                    guarded |= cur.guarded;
                }
                const double wj = quad_weight_1d<N>(jj);
#pragma unroll
                for (int ii = 0; ii < N; ++ii) {
                    const double pr = cur.pr[ii];
                    if (N <= 3) {
#pragma unroll
                        for (int k = 0; k < N * N; ++k)
                            if (N <= 3 && k == jj * N + ii) y2[(N <= 3) ? k : 0] = pr;
                    } else {
                        if (FAITHFUL)
                            ysum = ysum + (quad_weight_1d<N>(ii) * wj) * pr;    // OGG:244 / 252
                        else
                            ysum = fma(quad_weight_1d<N>(ii) * wj, pr, ysum);
                    }
                }
#pragma unroll
                for (int k = 0; k < N; ++k) dyc[k] = (jj == k) ? cur.dy0 : dyc[k];
            }
            // cur now holds lattice row (c, N-1) == (c+1, 0): the bottom row of the next cell row
            dyq = quad_average_1d<N>(dyc) * p.Re;                               // OGG:184,187
            if (N == 2) {
                const double d = 1.0 / 2.0;
                da = d * d * (y2[0] + y2[1] + y2[(N <= 3) ? N : 0] + y2[(N <= 3) ? N + 1 : 0]);
            } else if (N == 3) {
                const double d = 1.0 / 6.0;
                auto Y = [&](int a, int b) { return y2[(N == 3) ? a * 3 + b : 0]; };
                da = d * d * (Y(0, 0) + Y(0, 2) + Y(2, 0) + Y(2, 2) + 4.0 * (Y(0, 1) + Y(1, 0) + Y(1, 2) + Y(2, 1) + 4.0 * Y(1, 1)));
            } else {
                const double d = (N == 4) ? (1.0 / 12.0) : (1.0 / 180.0);
                da = d * d * ysum;
            }
            da = da * p.Re * p.Re;                                              // OGG:185
        }
        const long out_r = c - p.j0;
        double* __restrict__ rdx = p.dxq + out_r * p.nx;
        double* __restrict__ rdy = p.dyq + out_r * (p.nx + 1);
        double* __restrict__ rda = p.daq + out_r * p.nx;
        // the node column behind the run's last cell (the halo lane of the run's last strip): dyq only, stored directly
        if (closing) {
            rdy[ci] = dyq;
            if (img_col) rdy[h2 - ci] = dyq, rdy[h2 + ci] = dyq, rdy[p.nx - ci] = dyq;
        }
        if (cell_lane && MODE == QM_GUARD && guarded) p.fix_list[atomicAdd(p.fix_count, 1u)] = (unsigned)(out_r * p.nx + ci);
        // the row of the workgroup through LDS, then whole 128-byte lines to the arrays
        double(*buf)[QS_WG_CELLS] = out.v[(c - r0) & 1];
        if (lane < QS_CELLS) buf[0][slot] = dxq, buf[1][slot] = dyq, buf[2][slot] = da;
        __syncthreads();   // (one barrier per cell row: the other buffer was read a whole cell row of arithmetic ago)
        wg_store_aligned(rdx + c0, nv, buf[0], 0, 1, wave, lane);
        wg_store_aligned(rdy + c0, nv, buf[1], 0, 1, wave, lane);
        wg_store_aligned(rda + c0, nv, buf[2], 0, 1, wave, lane);
        const int last = nv - 1;
        if (qs.n_ic > 0) {   // nx/2 + i ascending; nx/2 - 1 - i and nx - 1 - i descend in i, i.e. ascend from the workgroup's last cell
            wg_store_aligned(rdx + qs.cell_up, qs.n_ic, buf[0], qs.k0, 1, wave, lane);
            wg_store_aligned(rda + qs.cell_up, qs.n_ic, buf[2], qs.k0, 1, wave, lane);
            wg_store_aligned(rdx + qs.cell_dn2, qs.n_ic, buf[0], last, -1, wave, lane);
            wg_store_aligned(rda + qs.cell_dn2, qs.n_ic, buf[2], last, -1, wave, lane);
            wg_store_aligned(rdx + qs.cell_dn4, qs.n_ic, buf[0], last, -1, wave, lane);
            wg_store_aligned(rda + qs.cell_dn4, qs.n_ic, buf[2], last, -1, wave, lane);
        }
        if (qs.n_ip > 0) {   // node columns: nx/2 + i; nx/2 - i and nx - i from the last column
            wg_store_aligned(rdy + qs.col_up, qs.n_ip, buf[1], qs.k1, 1, wave, lane);
            wg_store_aligned(rdy + qs.col_dn2, qs.n_ip, buf[1], last, -1, wave, lane);
            wg_store_aligned(rdy + qs.col_dn4, qs.n_ip, buf[1], last, -1, wave, lane);
        }
    }
}

template <int N, int MODE>
__global__ __launch_bounds__(64 * QS_WAVES) void bipolar_quad_kernel(QuadParams p, QuadRange rg) {
    __shared__ QuadOutLds out;
    bipolar_quad_body<N, MODE>(p, rg, out, blockIdx.x, blockIdx.y);
}

// Literal re-evaluation (bp_point, OGG:41-95 operation for operation) of the cells the guard handed over: half a wave per
// cell, one lane per Lobatto point (N*N <= 25 of 32 lanes), the lane of point 0 gathers the values with shuffles and sums
// in the reference's order; overwrites the cell's dxq, dyq, daq (and dyq[.][nx] for the last cell of a row).  A cell is
// re-evaluated iff one of ITS points is guarded, which depends on the cell alone: the result does not depend on tiling or
// banding.
template <int N>
__global__ __launch_bounds__(64) void bipolar_quad_tail_kernel(QuadParams p) {
    constexpr int M = N - 1;
    __shared__ double sdx[64], sdy[64];
    const unsigned bid = blockIdx.x, nblk = gridDim.x;
    const unsigned count = *p.fix_count;
    const int lane = threadIdx.x, half = lane >> 5, q = lane & 31;
    if (p.top_src && p.top_out_row >= 0)
        for (long i = (long)bid * 64 + lane; i < p.nx; i += (long)nblk * 64) p.dxq[p.top_out_row * p.nx + i] = p.top_src[i];
    const int jj = (q < N * N) ? q / N : 0, ii = (q < N * N) ? q % N : 0;
    for (unsigned k0 = bid * 2; k0 < count; k0 += nblk * 2) {  // wave-uniform trip count
        const unsigned k = k0 + half;
        const bool have = k < count;
        const unsigned lin = have ? p.fix_list[k] : 0u;
        const long out_r = lin / p.nx, ci = lin % p.nx;
        const long cj = p.j0 + out_r;
        double dx = 0.0, dy = 0.0;
        if (have && q < N * N) {
            const BpRow r = p.row_tab[M * cj + jj];
            const BpCol c = p.col_tab[M * ci + ii];
            double phis, rden, h_i, h_j;
            bp_point(r, c, p.rp, phis, h_i, h_j, rden);
            dx = h_i * 2 * kPi / (double)p.nx;                        // OGG:131
            dy = h_j * (90 - p.lat0_bp) * kPi180 / (double)p.ny;      // OGG:132
        }
        // Hand the values to the lane of point 0 through LDS (the barriers also keep the compiler from overlapping the
        // gather with bp_point's register peak -- the kernel is latency-bound and wants occupancy).
        __syncthreads();
        sdx[lane] = dx;
        sdy[lane] = dy;
        __syncthreads();
        if (!have || q != 0) continue;
        const int base = half << 5;
        double dxrow[N], dycol[N], dyright[N], y2[(N <= 3) ? N * N : 1];
        double ysum = 0.0;
#pragma unroll
        for (int a = 0; a < N; ++a) {
#pragma unroll
            for (int b = 0; b < N; ++b) {
                const double vx = sdx[base + a * N + b], vy = sdy[base + a * N + b];
                if (a == 0) dxrow[b] = vx;
                if (b == 0) dycol[a] = vy;
                if (b == N - 1) dyright[a] = vy;
                if (N <= 3)
                    y2[(N <= 3) ? a * N + b : 0] = vx * vy;
                else
                    ysum = ysum + (quad_weight_1d<N>(b) * quad_weight_1d<N>(a)) * (vx * vy);   // OGG:244/252
            }
        }
        p.dxq[out_r * p.nx + ci] = quad_average_1d<N>(dxrow) * p.Re;
        p.dyq[out_r * (p.nx + 1) + ci] = quad_average_1d<N>(dycol) * p.Re;
        if (ci == p.nx - 1) p.dyq[out_r * (p.nx + 1) + p.nx] = quad_average_1d<N>(dyright) * p.Re;
        double da;
        if (N == 2) {
            const double d = 1.0 / 2.0;
            da = d * d * (y2[0] + y2[1] + y2[(N <= 3) ? N : 0] + y2[(N <= 3) ? N + 1 : 0]);
        } else if (N == 3) {
            const double d = 1.0 / 6.0;
            auto Y = [&](int a, int b) { return y2[(N == 3) ? a * 3 + b : 0]; };
            da = d * d * (Y(0, 0) + Y(0, 2) + Y(2, 0) + Y(2, 2) + 4.0 * (Y(0, 1) + Y(1, 0) + Y(1, 2) + Y(2, 1) + 4.0 * Y(1, 1)));
        } else {
            const double d = (N == 4) ? (1.0 / 12.0) : (1.0 / 180.0);
            da = d * d * ysum;
        }
        p.daq[out_r * p.nx + ci] = da * p.Re * p.Re;
    }
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
size_t quad_workspace_bytes(long nx, long ny, long n_cell_rows) {
    constexpr int M = N - 1;
    const size_t tabs = (size_t)(M * ny + 2) * sizeof(BpRow) + (size_t)(M * nx + 1) * sizeof(BpCol);
    const size_t lists = tabs + 16 + QUAD_LL_CLAIM_WORDS * sizeof(unsigned) + (size_t)(n_cell_rows > 0 ? n_cell_rows : 0) * nx * sizeof(unsigned);
    return (lists + 7) / 8 * 8 + (size_t)nx * sizeof(double);   // + a row of dxq (QuadParams::top_src)
}

// Launch plan of one quadrature call: which cell rows run the plain algebraic form, which carry the guard, whether the
// band owns the j = ny row; shared by the function-level entry point and by the fused pass (ogg_pass.hip).
struct QuadPlan {
    QuadParams p;
    bool has_fast, has_guard, has_top;   // has_top: the band owns the j = ny row of dxq (evaluated with the tables)
    QuadRange fast, guard;
    double* top_buf;             // nx doubles at the end of the workspace (QuadParams::top_src)
};

constexpr double BP_SYM_FOLD_DEG = 6.0;   // QuadCols: degrees of longitude next to a fold line that are evaluated at their own columns

// the compact column space of a range of cell rows (QuadCols).  symmetry: the caller asks for mirrored columns; taken when nx is a multiple of
// 4 (the pole meridians are node columns) and the literal zones leave something to mirror
inline QuadCols quad_cols(long nx, int symmetry) {
    QuadCols c{};
    auto wgs = [](long lo, long hi) { return (hi - lo + QS_WG_CELLS - 1) / QS_WG_CELLS; };
    for (int r = 0; r < 3; ++r) c.lo[r] = c.hi[r] = 0;
    c.lo[0] = 0, c.hi[0] = nx;
    c.g_end[0] = c.g_end[1] = c.g_end[2] = wgs(0, nx);
    c.z = nx, c.sym = 0;
    if (!symmetry || nx % 4 != 0) return c;
    double deg = BP_SYM_FOLD_DEG;
    if (const char* e = getenv("OGG_BP_SYM_FOLD_DEG")) deg = atof(e);
    const long q = nx / 4;
    long z = (long)ceil(deg * (double)nx / 360.0);
    if (z < 1) z = 1;
    if (2 * z > q) return c;
    c.sym = 1, c.z = z;
    c.lo[0] = 0, c.hi[0] = q;                    // cells [0, q), closed by column q
    c.lo[1] = 2 * q - z, c.hi[1] = 2 * q + z;    // cells [nx/2 - z, nx/2 + z), closed by column nx/2 + z
    c.lo[2] = nx - z, c.hi[2] = nx;              // cells [nx - z, nx), closed by column nx
    long g = 0;
    for (int r = 0; r < 3; ++r) g += wgs(c.lo[r], c.hi[r]), c.g_end[r] = g;
    return c;
}

template <int N>
int plan_quad(QuadParams p, long n_dx_rows, long n_cell_rows, double guard_k, int symmetry, void* ws, long ws_bytes, QuadPlan& out) {
    constexpr int M = N - 1;
    const long n_rows = M * p.ny + 2, n_cols = M * p.nx + 1;
    const size_t need = quad_workspace_bytes<N>(p.nx, p.ny, n_cell_rows);
    OGG_REQUIRE(ws && (size_t)ws_bytes >= need, OGG_EARG, "bipolar quadrature workspace too small: %ld < %zu bytes", ws_bytes, need);
    BpRow* row_tab = static_cast<BpRow*>(ws);
    BpCol* col_tab = reinterpret_cast<BpCol*>(row_tab + n_rows);
    unsigned* fix_count = reinterpret_cast<unsigned*>(col_tab + n_cols);
    p.row_tab = row_tab;
    p.col_tab = col_tab;
    p.fix_count = fix_count;
    p.ll_claims = fix_count + 4;
    p.fix_list = fix_count + 4 + QUAD_LL_CLAIM_WORDS;
    p.top_src = nullptr;
    out.top_buf = reinterpret_cast<double*>(static_cast<char*>(ws) + need - (size_t)p.nx * sizeof(double));
    p.guard_k = guard_k;
    p.top_out_row = (n_dx_rows > n_cell_rows) ? n_cell_rows : -1;
    const double rp2 = p.rp * p.rp;
    p.sx = (2 * kPi) / (double)p.nx, p.sy = ((90 - p.lat0_bp) * kPi180) / (double)p.ny;
    p.rp2p = 1 + rp2, p.rp2m = 1 - rp2, p.rp2x4 = 4 * rp2;
    p.cdx = (2 * p.rp) * p.sx;
    p.guard_kk = guard_k * p.rp2x4;
    out.p = p;
    // A point can only be guarded where cos^2(phis) < 1/K and phis never exceeds the grid latitude of its lattice row, so the
    // cell rows whose top edge lies below acos(2/sqrt(K)) (a factor 4 of margin on cos^2) run without the guard.
    long jg = 0;  // first cell row that carries the guard
    if (guard_k > 4.0) {
        const double lat_thr = acos(2.0 / sqrt(guard_k)) / kPi180;  // degrees
        jg = (long)floor((double)p.ny * (lat_thr - p.lat0_bp) / (90.0 - p.lat0_bp)) - 1;
    }
    if (jg < 0) jg = 0;
    if (jg > p.ny) jg = p.ny;
    const long lo = p.j0, hi = p.j0 + n_cell_rows;
    auto range = [&](long b, long e, int sym) {
        QuadRange r{};
        r.row_begin = b;
        r.row_end = e;
        r.cols = quad_cols(p.nx, sym);
        r.gx = (unsigned)r.cols.g_end[2];
        long n_strips = 0;   // strips that hold cells
        for (int k = 0; k < 3; ++k) n_strips += (r.cols.hi[k] - r.cols.lo[k] + QS_CELLS - 1) / QS_CELLS;
        // enough waves to fill 1024 SIMDs (x 3 wave slots) more than once, without recomputing more than 1-2 % of the lattice rows (8192: +1.5 %)
        long target = 4096;
        if (const char* ev = getenv("OGG_QUAD_TARGET_WAVES")) target = atol(ev) > 0 ? atol(ev) : target;
        long rpc = ((e - b) * n_strips + target - 1) / target;
        r.rows_per_chunk = rpc < 1 ? 1 : (rpc > 32 ? 32 : rpc);
        r.gy = (unsigned)((e - b + r.rows_per_chunk - 1) / r.rows_per_chunk);
        return r;
    };
    out.has_fast = lo < hi && lo < jg;
    out.has_guard = lo < hi && hi > jg;
    out.has_top = n_dx_rows > n_cell_rows;
    // (the rows that carry the guard -- above 88.2 degrees for K = 4000 -- are evaluated at every column: the reference's own asymmetry
    // reaches 4.6e-12 relative in the cells next to the pole points)
    if (out.has_fast) out.fast = range(lo, hi < jg ? hi : jg, symmetry);
    if (out.has_guard) out.guard = range(lo > jg ? lo : jg, hi, 0);
    return OGG_OK;
}

constexpr unsigned FIXUP_BLOCKS = 2048;

// fix-up of the guarded cells, if this band has rows that carry the guard
template <int N>
int launch_quad_tail(const QuadPlan& q, hipStream_t s) {
    if (!q.has_guard && !(q.p.top_src && q.has_top)) return OGG_OK;
    bipolar_quad_tail_kernel<N><<<FIXUP_BLOCKS, 64, 0, s>>>(q.p);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}

template <int N>
int launch_quad(QuadParams p, long n_dx_rows, long n_cell_rows, double guard_k, int symmetry, void* ext_ws, long ext_ws_bytes, hipStream_t s) {
    // workspace (row/column tables, fix-up counter and list): the caller's (graph-capturable: no allocation at all), or from
    // the stream-ordered allocator (no host synchronisation, safe with concurrent streams)
    const size_t need = quad_workspace_bytes<N>(p.nx, p.ny, n_cell_rows);
    void* ws = ext_ws;
    long ws_bytes = ext_ws_bytes;
    ogg::AsyncScratch own(s);   // returned to the stream-ordered allocator on every exit path
    if (!ext_ws) {
        if (int e = own.alloc(&ws, need)) return e;
        ws_bytes = (long)need;
    }
    QuadPlan q;
    if (int e = plan_quad<N>(p, n_dx_rows, n_cell_rows, guard_k, symmetry, ws, ws_bytes, q)) return e;
    bipolar_tables_kernel<N><<<tables_blocks<N>(q.p), 256, 0, s>>>(q.p);
    OGG_LAUNCH_CHECK();
    if (q.has_fast) {
        bipolar_quad_kernel<N, QM_FAST><<<dim3(q.fast.gx, q.fast.gy), 64 * QS_WAVES, 0, s>>>(q.p, q.fast);
        OGG_LAUNCH_CHECK();
    }
    if (q.has_guard) {
        bipolar_quad_kernel<N, QM_GUARD><<<dim3(q.guard.gx, q.guard.gy), 64 * QS_WAVES, 0, s>>>(q.p, q.guard);
        OGG_LAUNCH_CHECK();
    }
    return launch_quad_tail<N>(q, s);
}

}  // namespace
