// K5 / K6: displaced-pole Southern cap -- device code (included by ogg_dpole.hip and by the fused pass, ogg_pass.hip).
//   displacedPoleCap_projection / _mesh      OGG:447-506   (generate_displaced_pole_grid OGG:509-518)
//   monotonic_bounding                       OGG:470-475
//   great_arc_distance, numerical_hi/hj      OGG:522-562
//   displacedPoleCap_metrics_quad            OGG:565-601
//
// The quadrature (K6) is fp64-VALU bound: per lattice point the reference re-projects 2*order probe points (complex
// division, atan2, hypot, atan each) and takes `order` haversines.  Nothing of the lattice is ever written to HBM here: a wave
// owns 64 consecutive lattice COLUMNS (one per lane) and walks up the lattice rows of its chunk; the Lobatto sums of a cell
// are formed at the lane of the cell's first column from its right-hand neighbours (DPP wave shifts) and carried from lattice
// row to lattice row in registers, in the reference's summation order (OGG:216-221, 244-253, 589-599).  Row-only factors
// (gnomonic radius of the 2*order+1 row variants) are scalar loads from a small table, column-only factors (e' of the
// column variants, OGG:451-452) stay in registers for the whole walk.
//
// monotonic_bounding is a sequential scan along i (column k is lowered by 360 iff v_k - x_{k-1} > 100 with x_{k-1} the ALREADY
// ADJUSTED previous column).  With s_k in {0,1} the "was lowered" state, s_k = f_k(s_{k-1}) with f_k(0) = [v_k - v_{k-1} > 100],
// f_k(1) = [v_k - (v_{k-1} - 360) > 100]: 1-bit maps whose composition is associative.  A wave composes the maps of its 64 columns
// with a shuffle scan, PUBLISHES the composed map of its strip (one 8-byte word per lattice row and strip, relaxed agent-scope
// atomic store) and reads the words of all strips to its left (relaxed agent-scope atomic loads, bypassing L1) to obtain the
// state at its first column: a decoupled look-back in which nobody waits for a predecessor's RESULT, only for its strip map,
// which depends on nothing.  Work items are handed out by an atomic ticket in (rows, strips-fastest) order, so every strip a
// wave can wait for belongs to a workgroup that is already running: no assumption on dispatch order or residency.  The wait
// is software-pipelined by one lattice row (row L+1's probes are evaluated between publishing row L and reading its
// predecessors) and bounded (an error flag is raised instead of spinning forever).  Bit-faithful to the sequential loop.
#pragma once
#include <cstdlib>

#include "ogg_common.h"
#include "ogg_math.h"

// The literal quadrature's LDS-pipelined walk (dpole_quad_literal_ring below) in the ONE configuration that is built, tested and timed: two
// pending rows in an LDS ring, the probes' arctangents restated with their coefficients in 40 vector registers (the library's bits), wave
// priority raised until a row's maps are published, the look-back loads of the row to finish issued after 4 of the row's 2 * order probes.
// What else was built and measured against it on one box -- pending rows in registers, ring depths 1 and 3, scalar and literal
// coefficient sets, priorities by look-back feedback, divisions without scaling with a row-wise fallback, a rolled probe loop, and the
// shader-clock profile of the walk -- is recorded in DESIGN.md 4.2 with its numbers (profiles/r03_dq_profile.txt) and no longer in the source.
constexpr int DQ_RING = 2;        // rows of slack for the look-back = pending rows parked in LDS
constexpr int DQ_LB_AT = 4;       // the look-back loads go out after this many probes of the row being evaluated

namespace {

using namespace ogg;

// ---- pieces of OGG:447-466 ---------------------------------------------------------------------------------------------
struct DpGeom {      // geometry of the cap (OGG:478-495)
    long ni, nj;
    double lon0, lat0, lam_pole, r_pole;
};

struct DpConst {
    double z0r, z0i, r_joint;
};

OGG_DEV DpConst dp_const(const DpGeom& p) {
    DpConst c;
    c.r_joint = tan((90 + p.lat0) * kPi180);  // OGG:494
    double s, co;
    sincos(p.lam_pole * kPi180, &s, &co);
    c.z0r = p.r_pole * co;                    // OGG:495
    c.z0i = p.r_pole * s;
    return c;
}

// column-only part of OGG:451-452: e' = (e - z0) / (1 - conj(z0) e)
OGG_DEV cplx dp_column(double iv, const DpGeom& p, const DpConst& c) {
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
OGG_DEV double dp_row_radius(double jv, const DpGeom& p, const DpConst& c) {
    const double lat = -90.0 + (jv * (p.lat0 - (-90.0))) / (double)p.nj;  // OGG:480-482
    return tan((90 + lat) * kPi180) / c.r_joint;
}

// per-point remainder of OGG:454-466: raw longitude (before the unwrap) and latitude
// w = (z + z0) / (1 + conj(z0) z), z = r e' (OGG:454-460), in numpy's complex arithmetic
OGG_DEV cplx dp_image(double r, cplx ep, const DpConst& c) {
    const cplx z = {r * ep.re, r * ep.im};
    const cplx num = {z.re + c.z0r, z.im + c.z0i};
    const cplx cz = cmul(cplx{c.z0r, -c.z0i}, z);
    const cplx den = {1 + cz.re, cz.im};
    return cdiv(num, den);
}
// Longitude (degrees, not yet unwrapped) and latitude of a point of the cap (OGG:461-466).  Two forms with the SAME bits:
// the device library's atan2 / atan, and their restatement with the coefficients in scalar registers (ogg_math.h: atc loaded by the
// caller).  The mesh takes the second (500 -> 420 instructions per point: 62 -> 56 us at 1/8 degree); the literal quadrature the first
// -- in that kernel, which runs at the register limit, the compiler already keeps the library's coefficients in scalar registers
// and a persistent set of forty only adds spills (1.42 against 1.37 ms).
OGG_DEV void dp_point(double r, cplx ep, const DpConst& c, double& lam_raw, double& phi) {
    const cplx w = dp_image(r, ep, c);
    lam_raw = atan2(w.im, w.re) * k180Pi;  // np.angle(deg=True)
    const double rw = cabs_np(w);
    phi = -90 + div_pi180(atan(rw * c.r_joint));
}
template <class C>
OGG_DEV void dp_point(double r, cplx ep, const DpConst& c, const C& atc, double& lam_raw, double& phi) {
    const cplx w = dp_image(r, ep, c);
    lam_raw = atan2_lib(w.im, w.re, atc) * k180Pi;
    const double rw = cabs_np(w);
    phi = -90 + div_pi180(atan_lib_wave(rw * c.r_joint, atc));   // (the reciprocal of arguments above 1 behind a wave-uniform branch)
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

// sin(x) and asin(x) for |x| < 2^-13 from three terms of their series: the truncation error is below 2^-80 relative, so the
// result is the correctly rounded value up to the rounding of the last fma (<= 0.5000001 ulp) -- inside the <1 ulp band of
// any libm, at 5 instructions instead of ocml's range-reduced sin / table-free asin.  The haversine of two probes that are
// 2e-6 rad apart only ever sees such arguments; anything larger goes to ocml.
// the library functions behind the range-specialised forms below, out of line: inlined, the literals of their polynomials (a few dozen
// doubles) are hoisted out of the quadrature's loop into vector registers for branches that a fine grid never takes
__device__ __attribute__((noinline)) double lib_sin(double x) { return sin(x); }
__device__ __attribute__((noinline)) double lib_asin(double x) { return asin(x); }
__device__ __attribute__((noinline)) double lib_cos(double x) { return cos(x); }

OGG_DEV double sin_tiny(double x) {
    if (fabs(x) < 0x1p-13) {
        const double x2 = x * x;
        return fma(x * x2, fma(x2, 1.0 / 120.0, -1.0 / 6.0), x);
    }
    return lib_sin(x);
}
OGG_DEV double asin_tiny(double x) {
    if (fabs(x) < 0x1p-13) {
        const double x2 = x * x;
        return fma(x * x2, fma(x2, 3.0 / 40.0, 1.0 / 6.0), x);
    }
    return lib_asin(x);
}

// cos(x) for the latitudes of a southern cap: x in (-3 pi/4, -pi/4) and not within 2^-20 of -pi/2.  cos x = sin(x + pi/2): the 33-bit head
// of pi/2 is added exactly, its tail in double-double, and the sum goes through the fdlibm sine kernel with tail -- 0.73 ulp at worst
// (validated on the host against cosl on 2.7e7 arguments; glibc: 0.52, and the two differ in 2 % of the arguments, as ocml and glibc do),
// 20 instructions instead of ocml's range-reduced cosine.  Taken only when the WHOLE wave is in range (one ballot): a cap row always is,
// except next to the pole itself; everything else goes to ocml.
OGG_DEV double cos_cap(double x) {
    constexpr double pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
    const double z = x + pio2_1;
    const bool in_range = (x < -0.7853981633974483) && (x > -2.356194490192345) && (fabs(z) >= 0x1p-20);
    if (__builtin_expect(__ballot(!in_range) != 0ull, 0)) return lib_cos(x);
    const double y0 = z + pio2_1t;
    const double y1 = (z - y0) + pio2_1t;
    constexpr double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                     S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double w = y0 * y0, v = w * y0;
    const double r = S2 + w * (S3 + w * (S4 + w * (S5 + w * S6)));
    return y0 - ((w * (0.5 * y1 - v * r) - y1) - v * S1);
}

// ---- 1-bit state maps, packed NB to a word -------------------------------------------------------------------------------
// A map is (m0, m1): the state after it when the state before it was 0 / 1; identity = (0, all).
OGG_DEV unsigned map_apply(unsigned m0, unsigned m1, unsigned s) { return (s & m1) | (~s & m0); }

// inclusive wave64 scan of the composition (later o earlier) over the lanes
OGG_DEV void map_scan(unsigned& m0, unsigned& m1) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned e0 = __shfl_up(m0, off), e1 = __shfl_up(m1, off);
        if (lane >= off) {
            const unsigned h0 = map_apply(m0, m1, e0), h1 = map_apply(m0, m1, e1);
            m0 = h0, m1 = h1;
        }
    }
}

constexpr unsigned long long LB_VALID = 1ull << 63;
constexpr int LB_SPIN_LIMIT = 1 << 20;   // polls before a wave gives up (seconds): raises *err, never reached in a sane launch

OGG_DEV void lb_publish(unsigned long long* word, unsigned m0, unsigned m1) {
    __hip_atomic_store(word, LB_VALID | (unsigned long long)m0 | ((unsigned long long)m1 << 16), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// State at the first column of strip s of a lattice row: the composition of the published maps of strips 0 .. s-1 (strip 0's
// map is constant, so the result does not depend on a state before it).  row_words: the row's words, one per strip.  Wave-uniform
// arguments; all 64 lanes must call.
OGG_DEV void lb_compose_from(const unsigned long long* row_words, long from, long s, unsigned all, int* err, unsigned& t0, unsigned& t1) {
    const int lane = threadIdx.x & 63;   // composes the maps of strips from .. s-1 (from: a multiple of 64) onto the map (t0, t1)
    for (long base = from; base < s; base += 64) {
        const long k = base + lane;
        const bool have = k < s;
        unsigned long long w = LB_VALID;
        int spins = 0;
        for (;;) {
            if (have) w = __hip_atomic_load(row_words + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__ballot((w & LB_VALID) != 0ull) == ~0ull) break;
            // wave-uniform exits (spins is the same in every lane): give up after LB_SPIN_LIMIT polls, or as soon as another wave
            // has given up (the call has failed anyway: drain the grid quickly)
            ++spins;
            if (spins > LB_SPIN_LIMIT) {
                if (lane == 0) atomicExch(err, 1);
                break;
            }
            if ((spins & 255) == 0 && __builtin_amdgcn_readfirstlane(__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) break;
            __builtin_amdgcn_s_sleep(8);
        }
        unsigned m0 = have ? (unsigned)(w & 0xffffull) & all : 0u;
        unsigned m1 = have ? (unsigned)((w >> 16) & 0xffffull) & all : all;
        // usually at most one strip of the block holds anything but the identity (strip 0, or the strip where the longitude crosses
        // the cut): its map is the block's; otherwise compose them all
        const unsigned long long nonid = __ballot(m0 != 0u || m1 != all);
        unsigned b0, b1;
        if (nonid == 0ull) {
            b0 = 0u, b1 = all;
        } else if ((nonid & (nonid - 1ull)) == 0ull) {
            const int src = 63 - __builtin_clzll(nonid);
            b0 = __shfl(m0, src), b1 = __shfl(m1, src);
        } else {
            map_scan(m0, m1);
            b0 = __shfl(m0, 63), b1 = __shfl(m1, 63);   // the block's composed map
        }
        const unsigned n0 = map_apply(b0, b1, t0), n1 = map_apply(b0, b1, t1);
        t0 = n0, t1 = n1;
    }
}

OGG_DEV unsigned lb_incoming(const unsigned long long* row_words, long s, unsigned all, int* err) {
    unsigned t0 = 0u, t1 = all;
    lb_compose_from(row_words, 0, s, all, err, t0, t1);
    return t0 & all;
}

// The same in two halves, so that the latency of the loads hides behind a lattice row's arithmetic: lb_issue() sends the loads of
// up to LB_BATCH blocks of 64 strips (one word per lane and block) and returns at once; lb_resolve() -- a row's evaluation later --
// polls again only what was not yet published then, composes the maps, and handles strips beyond the batch with fresh loads.  One
// round trip to memory at most in the common case, instead of one per 64 strips one after the other (1/8 degree, order 4: 366 strips).
constexpr int LB_BATCH = 6;

struct LbBatch {
    unsigned long long w[LB_BATCH];
};

OGG_DEV void lb_issue(const unsigned long long* row_words, long s, LbBatch& b) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < LB_BATCH; ++i) {
        if (64 * i >= s) {                              // wave-uniform: no strip in this block
            b.w[i] = LB_VALID | (0xffffull << 16);      // the identity, published
        } else {
            const long k = 64 * i + lane;
            b.w[i] = __hip_atomic_load(row_words + (k < s ? k : s - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // lanes beyond s: word s-1 again
        }
    }
}

OGG_DEV unsigned lb_resolve(const unsigned long long* row_words, long s, unsigned all, LbBatch& b, int* err) {
    const int lane = threadIdx.x & 63;
    int spins = 0;
    for (;;) {
        unsigned long long v = LB_VALID;
#pragma unroll
        for (int i = 0; i < LB_BATCH; ++i) v &= b.w[i];
        if (__ballot((v & LB_VALID) == 0ull) == 0ull) break;
        ++spins;
        if (spins > LB_SPIN_LIMIT) {
            if (lane == 0) atomicExch(err, 1);
            break;
        }
        if ((spins & 255) == 0 && __builtin_amdgcn_readfirstlane(__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) break;
        __builtin_amdgcn_s_sleep(8);
        lb_issue(row_words, s, b);                     // rare: some strip to the left had not published yet
    }
    unsigned t0 = 0u, t1 = all;
#pragma unroll
    for (int i = 0; i < LB_BATCH; ++i) {
        if (64 * i >= s) break;                         // wave-uniform
        const bool have = 64 * i + lane < s;
        const unsigned long long w = b.w[i];
        const unsigned m0 = have ? (unsigned)(w & 0xffffull) & all : 0u, m1 = have ? (unsigned)((w >> 16) & 0xffffull) & all : all;
        // almost every strip of almost every row holds the identity (the constant maps sit at column 0 and where the longitude crosses
        // the cut): compose the few others in strip order, one broadcast each
        unsigned long long nonid = __ballot(m0 != 0u || m1 != all);
        while (nonid != 0ull) {
            const int src = __builtin_ctzll(nonid);
            const unsigned b0 = (unsigned)__builtin_amdgcn_readlane((int)m0, src), b1 = (unsigned)__builtin_amdgcn_readlane((int)m1, src);
            const unsigned n0 = map_apply(b0, b1, t0), n1 = map_apply(b0, b1, t1);
            t0 = n0, t1 = n1;
            nonid &= nonid - 1ull;
        }
    }
    // wider grids than the batch holds (1/16 degree: 732 strips): the rest block by block, as lb_incoming does
    if (s > 64 * LB_BATCH) lb_compose_from(row_words, 64 * LB_BATCH, s, all, err, t0, t1);
    return t0 & all;
}

// The same for ONE scan (1-bit maps).  f(0) <= f(1) always (lowering the previous column can only raise the difference), so a map is
// "constant 0", "constant 1" or the identity, and the state after a run of maps is the value of the LAST constant one: two ballots and
// a count-leading-zeros instead of a 6-step shuffle scan.
OGG_DEV unsigned lb_incoming1(const unsigned long long* row_words, long s, int* err) {
    const int lane = threadIdx.x & 63;
    unsigned st = 0u;
    for (long base = 0; base < s; base += 64) {
        const long k = base + lane;
        const bool have = k < s;
        unsigned long long w = LB_VALID | (1ull << 16);   // identity
        int spins = 0;
        for (;;) {
            if (have) w = __hip_atomic_load(row_words + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__ballot((w & LB_VALID) != 0ull) == ~0ull) break;
            ++spins;
            if (spins > LB_SPIN_LIMIT) {
                if (lane == 0) atomicExch(err, 1);
                break;
            }
            if ((spins & 255) == 0 && __builtin_amdgcn_readfirstlane(__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) break;
            __builtin_amdgcn_s_sleep(8);
        }
        const unsigned m0 = (unsigned)(w & 1ull), m1 = (unsigned)((w >> 16) & 1ull);
        const unsigned long long C = __ballot(have && m0 == m1), V = __ballot(m0 != 0u);
        if (C) st = (unsigned)((V >> (63 - __builtin_clzll(C))) & 1ull);
    }
    return st;
}

// work item of a workgroup from an atomic ticket (see the head of this file); one barrier
OGG_DEV long take_ticket(unsigned* counter, unsigned* s_slot) {
    if (threadIdx.x == 0) *s_slot = atomicAdd(counter, 1u);
    __syncthreads();
    return (long)__builtin_amdgcn_readfirstlane((int)*s_slot);
}

// ---- quadrature of the finite-difference scale factors (OGG:565-601) ----------------------------------------------------
constexpr int DQ_WAVES = 4;       // strips per workgroup
constexpr int DQ_COLS = 63;       // lattice columns a strip owns; lane 63 is the first column of the next strip
constexpr int DQ_RPC_MAX = 32;    // cell rows one wave walks at most (plan_dquad; sizes the workspace and the LDS row table)
constexpr int DP_ARC_LITERAL = 0; // haversine of the unwrapped longitudes, operation for operation as OGG:522-532
constexpr int DP_ARC_CHORD = 1;   // same stencil, distance of two probes from their gnomonic images (see dq_chord_point)

struct DpQuadParams {
    DpGeom g;                    // ni = nx, nj = ny
    double eps, Re;
    long j0;                     // first cell row of the band (band-local output row 0)
    long n_cell_rows, n_dx_rows; // dyq / daq rows; dxq rows (n_cell_rows + 1 on the band that owns row ny)
    long n_rows, n_cols;         // unique lattice rows of the band (M n_cell_rows + 1), lattice columns (M nx + 1)
    long rows_per_chunk;         // cell rows one wave walks (its first lattice row is the last one of the chunk below, recomputed)
    long n_chunks, n_strips, gx; // gx = strip workgroups per chunk
    // The lattice columns the strips walk: all of them, 0 .. n_cols - 1, or (sym, chord form only) the half that determines the rest.
    // The map of OGG:447-467 is mirror-symmetric about the meridian through lon_dp: when that meridian is a node column,
    // i_c = (lon_dp - lon0) ni / 360 an integer, and ni is even, the cells [c_first, c_first + ni/2) with c_first = i_c or i_c - ni/2
    // (the one that does not wrap) are evaluated and cell c also writes its values to cell (2 c_first - 1 - c) mod ni, node column c
    // (dyq) to column (2 c_first - c) mod ni; column ni is the twin of column 0 (the same meridian, lon0 + 360).  The reference's own
    // results are mirror images of each other to the level of its own finite-difference noise (1e-10 at ni = 720, 6e-10 at 2880,
    // uniformly in i: the oracle), which is also how far it is from the exact value of its formula.
    long u_first, n_src_cols;    // first lattice column of the strips, number of lattice columns they cover (M c_first, M ni / 2 + 1)
    long c_first, c_end;         // cells [c_first, c_end) write dxq / daq; node columns [c_first, c_end] write dyq
    int sym;
    QuadNodes q;
    const double* row_tab;       // [NV][n_rows]: gnomonic radius of the row variants (base, +eps, -eps, +2eps, -2eps, ...)
    const double* col_tab;       // [NV][2][n_cols]: e' of the column variants
    unsigned long long* words;   // [n_chunks][M rows_per_chunk + 1][n_strips]: published strip maps (literal form only)
    unsigned* ticket;            // work counter; ticket[1] is the error flag
    double *dxq, *dyq, *daq;
};

// what lane `lane` of strip `strip` owns: its lattice column, and what it writes
struct DqLane {
    long uc, ci;            // lattice column (clamped to the last one), its cell
    bool active;            // the strip exists (wave-uniform)
    bool valid, cell_start;
    bool out_lane;          // writes dxq, dyq, daq of cell ci
    bool dy_edge;           // the node column behind the last cell: dyq only
    long cm, pm;            // sym: mirror image of the cell, of the node column (a column that lands on 0 also writes its twin ni)
};

OGG_HD DqLane dq_lane(const DpQuadParams& p, int M, long strip, int lane, int cols_per_strip) {
    DqLane q{};
    const long u0 = strip * cols_per_strip;
    q.active = u0 < p.n_src_cols - 1;            // else nothing but another strip's halo column (n_src_cols >= 2)
    q.valid = u0 + lane < p.n_src_cols;
    q.uc = p.u_first + (q.valid ? u0 + lane : p.n_src_cols - 1);
    q.ci = q.uc / M;
    q.cell_start = q.valid && (q.uc % M == 0);
    q.out_lane = q.active && q.cell_start && q.ci < p.c_end && lane <= cols_per_strip - M;
    q.dy_edge = q.active && q.cell_start && q.ci == p.c_end;
    long cm = 2 * p.c_first - 1 - q.ci, pm = 2 * p.c_first - q.ci;
    q.cm = cm < 0 ? cm + p.g.ni : cm, q.pm = pm < 0 ? pm + p.g.ni : pm;
    return q;
}

// Row / column tables of one call and the reset of its look-back words; n_words = 0 for the chord form.
template <int N>
OGG_DEV void dpole_quad_tables_body(const DpQuadParams& p, long bx, long n_blocks) {
    constexpr int M = N - 1, NV = N + 1;
    const long k = bx * blockDim.x + threadIdx.x;
    const DpConst c = dp_const(p.g);
    double* row_tab = const_cast<double*>(p.row_tab);
    double* col_tab = const_cast<double*>(p.col_tab);
    if (k == 0) p.ticket[0] = 0u, p.ticket[1] = 0u;
    if (k < p.n_rows * NV) {
        const long row = k / NV;
        const int var = (int)(k % NV);
        double jv = lattice_node(p.q, (int)(row % M), p.j0 + row / M);
        if (var > 0) {
            const double off = (double)((var + 1) / 2) * p.eps;   // OGG:538-543: j + eps, j + 2.0*eps, ...
            jv = (var & 1) ? jv + off : jv - off;
        }
        row_tab[var * p.n_rows + row] = dp_row_radius(jv, p.g, c);
    } else if (k < p.n_rows * NV + p.n_cols * NV) {
        const long kk = k - p.n_rows * NV;
        const long col = kk / NV;
        const int var = (int)(kk % NV);
        double iv = lattice_node(p.q, (int)(col % M), col / M);
        if (var > 0) {
            const double off = (double)((var + 1) / 2) * p.eps;
            iv = (var & 1) ? iv + off : iv - off;
        }
        const cplx ep = dp_column(iv, p.g, c);
        col_tab[(var * 2 + 0) * p.n_cols + col] = ep.re;
        col_tab[(var * 2 + 1) * p.n_cols + col] = ep.im;
    }
    if (p.words) {
        const long n_words = p.n_chunks * (M * p.rows_per_chunk + 1) * p.n_strips;
        for (long w = k; w < n_words; w += n_blocks * blockDim.x) p.words[w] = 0ull;
    }
}

template <int N>
inline long dpole_quad_tables_blocks(const DpQuadParams& p) {
    constexpr int M = N - 1, NV = N + 1;
    const long n_tab = (p.n_rows + p.n_cols) * NV;
    const long n_words = p.words ? p.n_chunks * (M * p.rows_per_chunk + 1) * p.n_strips : 0;
    const long n = n_tab > n_words / 4 ? n_tab : n_words / 4;   // a thread zeroes up to ~4 words
    return (n + 255) / 256;
}

// ---- host side: workspace layout and launch plan of one quadrature call (shared with the fused pass) -----------------------
// workspace: [ticket, error flag, pad (16 B)] [row_tab] [col_tab] [look-back words]
inline long dq_strips(int order, long nx) { return ((long)(order - 1) * nx + DQ_COLS - 1) / DQ_COLS; }

inline size_t dq_workspace_bytes(int order, long nx, long n_cell_rows) {
    const long M = order - 1, NV = order + 1;
    const long n_rows = M * n_cell_rows + 1, n_cols = M * nx + 1;
    // look-back words: n_chunks x (M rows_per_chunk + 1) lattice rows x strips, for the worst rows_per_chunk plan_dquad may choose
    // (the chunks' shared edge rows are stored twice, and the last chunk is padded to a whole one: rows_per_chunk = 1 is NOT always the
    // largest layout -- 5 cell rows in chunks of 2 need 3 x 7 = 21 lattice rows against 5 x 4 = 20)
    const long nr = n_cell_rows > 0 ? n_cell_rows : 1;
    long rows = 0;
    for (long rpc = 1; rpc <= DQ_RPC_MAX; ++rpc) {
        const long need = ((nr + rpc - 1) / rpc) * (M * rpc + 1);
        rows = need > rows ? need : rows;
    }
    const long n_words = rows * dq_strips(order, nx);
    return 16 + (size_t)(NV * (n_rows + 2 * n_cols)) * sizeof(double) + (size_t)n_words * sizeof(unsigned long long);
}

inline int plan_dquad(int arc_form, int order, const DpGeom& g, double Re, long j0, long n_dx_rows, long n_cell_rows, double* dxq,
                      double* dyq, double* daq, void* ws, long ws_bytes, const QuadNodes& q, DpQuadParams& p, int symmetry = 0) {
    const long M = order - 1, NV = order + 1;
    const size_t need = dq_workspace_bytes(order, g.ni, n_cell_rows);
    OGG_REQUIRE(ws && (size_t)ws_bytes >= need, OGG_EARG, "displaced-pole quadrature workspace too small: %ld < %zu bytes", ws_bytes, need);
    p.g = g;
    p.eps = 1e-3;   // OGG:583
    p.Re = Re;
    p.j0 = j0, p.n_cell_rows = n_cell_rows, p.n_dx_rows = n_dx_rows;
    p.n_rows = M * n_cell_rows + 1, p.n_cols = M * g.ni + 1;
    p.n_strips = dq_strips(order, g.ni);
    p.u_first = 0, p.n_src_cols = p.n_cols, p.c_first = 0, p.c_end = g.ni, p.sym = 0;
    if (symmetry && arc_form == DP_ARC_CHORD && g.ni % 2 == 0 && g.ni >= 4) {
        double t = fmod(g.lam_pole - g.lon0, 360.0);
        if (t < 0.0) t += 360.0;
        const double ic = t * (double)g.ni / 360.0;
        if (ic == floor(ic) && ic >= 0.0 && ic <= (double)g.ni) {   // the meridian of the displaced pole is a node column
            long a = (long)ic % g.ni;
            if (a > g.ni / 2) a -= g.ni / 2;
            p.sym = 1, p.c_first = a, p.c_end = a + g.ni / 2;
            p.u_first = M * a, p.n_src_cols = M * (g.ni / 2) + 1;
            p.n_strips = (p.n_src_cols - 1 + DQ_COLS - 1) / DQ_COLS;
        }
    }
    p.gx = (p.n_strips + DQ_WAVES - 1) / DQ_WAVES;
    // Cell rows per chunk.  A wave walks M rpc + 1 lattice rows (the first one recomputed), and the launch runs in ROUNDS of as many
    // waves as the chip holds: the literal form, at 2 waves per SIMD, 2048 -- so the time goes as rounds x lattice rows per wave, and the
    // rpc that minimises it is taken (OM4's cap, 61 x 183 strips: rpc 6 = 2013 waves = ONE round, 0.152 ms; the old "8192 waves" rule
    // chose rpc 2 = 5673 waves = 2.8 rounds, 0.175 ms; the 1/8 degree cap keeps its rpc 13 = 3.9 rounds).  The chord form shares launch B
    // with other roles and keeps the wave-count rule.  OGG_DPQUAD_TARGET_WAVES forces that rule with another count (experiments, tests).
    long rpc;
    const char* ev = getenv("OGG_DPQUAD_TARGET_WAVES");
    if (arc_form == DP_ARC_LITERAL && !(ev && atol(ev) > 0)) {
        const long slots = 2048;
        long best_cost = -1;
        rpc = 1;
        for (long r = 1; r <= DQ_RPC_MAX; ++r) {
            const long waves = ((n_cell_rows > 0 ? n_cell_rows : 1) + r - 1) / r * p.n_strips;
            const long cost = (waves + slots - 1) / slots * (M * r + 1);
            if (best_cost < 0 || cost < best_cost) best_cost = cost, rpc = r;
        }
    } else {
        long target = 8192;   // enough waves to fill 1024 SIMDs several times over without recomputing more than a few % of the lattice rows
        if (ev) target = atol(ev) > 0 ? atol(ev) : target;
        rpc = (n_cell_rows * p.n_strips + target - 1) / target;
    }
    p.rows_per_chunk = rpc < 1 ? 1 : (rpc > DQ_RPC_MAX ? DQ_RPC_MAX : rpc);
    OGG_REQUIRE((size_t)ws_bytes >= 16 + (size_t)(NV * (p.n_rows + 2 * p.n_cols)) * sizeof(double) +
                                        (arc_form == DP_ARC_LITERAL ? (size_t)(((n_cell_rows > 0 ? n_cell_rows : 1) + p.rows_per_chunk - 1) / p.rows_per_chunk *
                                                                               (M * p.rows_per_chunk + 1) * p.n_strips) * sizeof(unsigned long long)
                                                                    : 0),
                OGG_EARG, "displaced-pole quadrature workspace too small for %ld rows per chunk", p.rows_per_chunk);
    p.n_chunks = n_cell_rows > 0 ? (n_cell_rows + p.rows_per_chunk - 1) / p.rows_per_chunk : 1;
    p.q = q;
    p.ticket = static_cast<unsigned*>(ws);
    double* tabs = reinterpret_cast<double*>(static_cast<char*>(ws) + 16);
    p.row_tab = tabs;
    p.col_tab = tabs + NV * p.n_rows;
    p.words = (arc_form == DP_ARC_LITERAL) ? reinterpret_cast<unsigned long long*>(tabs + NV * p.n_rows + NV * 2 * p.n_cols) : nullptr;
    p.dxq = dxq, p.dyq = dyq, p.daq = daq;
    return OGG_OK;
}

// -- chord form ------------------------------------------------------------------------------------------------------------
// The reference differentiates great-arc distances numerically (OGG:535-562): h = (8 ds(eps) - ds(2 eps)) / (12 eps), where
// ds is the haversine distance between two projected points that are ~2e-6 rad apart, formed from longitudes/latitudes in
// degrees.  That subtraction of two O(1) angles loses 10 digits.  The chord form keeps the SAME stencil (the same probe points, the same
// conformal image w of each probe) but takes the distance between two probes from their positions on the sphere -- from vectors pa, pb
// along the two points: sin(ds) = |pa x pb| / (|pa| |pb|) -- : no atan2 / atan / hypot per probe, no sin/cos/asin per pair and no
// longitude at all, hence no unwrap.  Against a 50-digit evaluation of the reference's own formula on 10 500 cells of the 1/8 degree cap
// (tests/golden/truth_table.npz, tests/test_gpu_truth.py, profiles/r04_truth_table.json; max relative error of dx / dy / area):
//     the fp64 reference itself (numpy)   1.41e-9 / 8.96e-10 / 1.21e-9
//     the literal form (this file)         1.37e-9 / 8.50e-10 / 1.20e-9
//     the chord form                       7.61e-10 / 2.74e-10 / 8.26e-10
// i.e. the chord form is CLOSER to what the reference's formula means than the reference's own fp64 evaluation, at a sixth of the
// arithmetic.  Since round 4 it is the default of main() and of the Python entry points (OGG_DP_ARC / dp_arc / arc_form = literal opts
// back into the reference's operation sequence) and of the C entry points without an arc_form argument.
// A probe's point on the sphere in homogeneous form: with w = num / den the conformal image of the probe (OGG:454-455),
// (X, Y) = r_joint w its gnomonic image and (X, Y, -1) a vector along the point, so is
//   (P, Q, -D) = (r_joint Re(num conj(den)), r_joint Im(num conj(den)), -|den|^2)
// -- no division per probe.  num = r e' + z0 and den = 1 + r conj(z0) e' are each one fma per component.
struct Hom {
    double P, Q, D;
};

// (P, Q, -D) scaled by 1 / r_joint: (Re(num conj(den)), Im(num conj(den)), -|den|^2 / r_joint)
OGG_DEV Hom dp_homogeneous(double r, cplx ep, const DpConst& c, double inv_rj) {
    const double cr = fma(c.z0r, ep.re, c.z0i * ep.im), ci = fma(c.z0r, ep.im, -(c.z0i * ep.re));   // conj(z0) e' (column-only: hoisted)
    const double nr = fma(r, ep.re, c.z0r), ni = fma(r, ep.im, c.z0i);
    const double dr = fma(r, cr, 1.0), di = r * ci;
    Hom h;
    h.P = fma(nr, dr, ni * di);
    h.Q = fma(ni, dr, -(nr * di));
    h.D = fma(dr, dr, di * di) * inv_rj;
    return h;
}

// Great-arc distance of two nearby points a, b given along (P, Q, -D): tan(theta) = |a x b| / (a . b); theta = atan(tan theta)
// from three terms of the series (the probes of the eps = 1e-3 stencil are ~1e-6 rad apart; exact to 1e-30 below 1e-3).  One
// reciprocal square root per arc, seed + one Newton step (2^-50: the cross product itself carries 1e-10 of cancellation):
// tan theta = |a x b|^2 / sqrt(|a x b|^2 (a . b)^2).
OGG_DEV double homogeneous_arc(const Hom& a, const Hom& b) {
    const double c1 = fma(a.D, b.Q, -(a.Q * b.D));
    const double c2 = fma(a.P, b.D, -(a.D * b.P));
    const double c3 = fma(a.P, b.Q, -(a.Q * b.P));
    const double cc = fma(c1, c1, fma(c2, c2, c3 * c3));
    if (cc == 0.0) return 0.0;
    const double dot = fma(a.P, b.P, fma(a.Q, b.Q, a.D * b.D));
    const double x = cc * (dot * dot);
    double y = __builtin_amdgcn_rsq(x);
    y = y * fma(-0.5 * x * y, y, 1.5);
    const double t = cc * y;                   // tan(theta)
    const double q = t * t;
    double th = t * fma(q, fma(q, 1.0 / 5.0, -1.0 / 3.0), 1.0);
    // grids so coarse that two probes are more than 1e-3 rad apart (a few dozen columns): the whole wave takes the library atan, behind a
    // wave-uniform branch that the fine grids never enter
    if (__builtin_expect(__ballot(t >= 1e-3) != 0ull, 0)) th = (t >= 1e-3) ? atan_lib(t) : th;   // the library's atan, bit for bit (ogg_math.h)
    return th;
}

// -- one lattice row, one column per lane ---------------------------------------------------------------------------------
// Pair k of the stencil: k < H: (j, i + m eps) / (j, i - m eps), m = k + 1 (h_i, OGG:538,541); k >= H: (j + m eps, i) / (j - m eps, i),
// m = k - H + 1 (h_j, OGG:553,556).  Probe 2k is the "+" point (point 0 of great_arc_distance), probe 2k+1 the "-" point.
template <int F>
struct DqPending {       // literal form: raw longitude and latitude (degrees) of the two probes of each pair
    double va[F], vb[F];
    double pa[F], pb[F];
};

template <int F>
OGG_DEV void dq_probe_pair(int k, const double* r, const cplx* ep, double& ra, double& rb, cplx& epa, cplx& epb) {
    constexpr int H = F / 2;
    // r[v], ep[v]: variant v of the row radius / of e' (0: base, 2m-1: +m eps, 2m: -m eps); selects instead of dynamic indexing
    ra = rb = r[0];
    epa = epb = ep[0];
#pragma unroll
    for (int m = 1; m <= H; ++m) {
        if (k == m - 1) epa = ep[2 * m - 1], epb = ep[2 * m];
        if (k == H + m - 1) ra = r[2 * m - 1], rb = r[2 * m];
    }
}

// phase 1 of the literal form: all probes of this lane's lattice point (OGG:522-526, 454-466), ONE at a time (two projections
// in flight need 60 more registers than one)
template <int F>
OGG_DEV void dq_literal_probes_lib(const double* r, const cplx* ep, const DpConst& c, DqPending<F>& o) {
#pragma unroll    // fully unrolled: no compare-selects to pick a probe's operands and result slots
    for (int q = 0; q < 2 * F; ++q) {
        const int k = q >> 1;
        double ra, rb;
        cplx epa, epb;
        dq_probe_pair<F>(k, r, ep, ra, rb, epa, epb);
        const bool minus = (q & 1) != 0;
        double v, ph;
        dp_point(minus ? rb : ra, minus ? epb : epa, c, v, ph);
#pragma unroll
        for (int t = 0; t < F; ++t) {
            o.va[t] = (q == 2 * t) ? v : o.va[t];
            o.vb[t] = (q == 2 * t + 1) ? v : o.vb[t];
            o.pa[t] = (q == 2 * t) ? ph : o.pa[t];
            o.pb[t] = (q == 2 * t + 1) ? ph : o.pb[t];
        }
    }
}

// the same for the LDS-pipelined walk below (probes Q0 .. Q1-1 of the lattice point): the restated arctangents with their coefficients in
// vector registers (`atc`; the library's bits, ogg_math.h)
typedef AtanVgpr DqAtan;

template <int F, int Q0 = 0, int Q1 = 2 * F>
OGG_DEV void dq_literal_probes(const double* r, const cplx* ep, const DpConst& c, const DqAtan& atc, DqPending<F>& o) {
#pragma unroll
    for (int q = Q0; q < Q1; ++q) {
        const int k = q >> 1;
        double ra, rb;
        cplx epa, epb;
        dq_probe_pair<F>(k, r, ep, ra, rb, epa, epb);
        const bool minus = (q & 1) != 0;
        double v, ph;
        dp_point(minus ? rb : ra, minus ? epb : epa, c, atc, v, ph);
#pragma unroll
        for (int t = 0; t < F; ++t) {
            o.va[t] = (q == 2 * t) ? v : o.va[t];
            o.vb[t] = (q == 2 * t + 1) ? v : o.vb[t];
            o.pa[t] = (q == 2 * t) ? ph : o.pa[t];
            o.pb[t] = (q == 2 * t + 1) ? ph : o.pb[t];
        }
    }
}

// phase 2, once the unwrap states are known: the haversines (OGG:527-532) and the central differences.  st = packed "was lowered
// by 360" states of this lane's 2F probes (bit 2k: "+" probe of pair k)
template <int F>
OGG_DEV void dq_literal_finish(const DqPending<F>& o, unsigned st, double reps, double& hi, double& hj) {
    constexpr int H = F / 2;
    double ds[F];
#pragma unroll
    for (int k = 0; k < F; ++k) {
        double va = o.va[0], vb = o.vb[0], pha = o.pa[0], phb = o.pb[0];
#pragma unroll
        for (int q = 1; q < F; ++q) {
            va = (k == q) ? o.va[q] : va, vb = (k == q) ? o.vb[q] : vb;
            pha = (k == q) ? o.pa[q] : pha, phb = (k == q) ? o.pb[q] : phb;
        }
        const double xa = ((st >> (2 * k)) & 1u) ? va - 360 : va;            // OGG:473
        const double xb = ((st >> (2 * k + 1)) & 1u) ? vb - 360 : vb;
        const double lam0 = xa * kPi180, phi0 = pha * kPi180;               // OGG:527-528
        const double lam1 = xb * kPi180, phi1 = phb * kPi180;
        const double dphi = phi1 - phi0, dlam = lam1 - lam0;
        const double sp = sin_tiny(0.5 * dphi), sl = sin_tiny(0.5 * dlam);
        const double d = sp * sp + sl * sl * cos_cap(phi0) * cos_cap(phi1); // OGG:531
        const double dsk = 2.0 * asin_tiny(sqrt(d));
#pragma unroll
        for (int q = 0; q < F; ++q) ds[q] = (k == q) ? dsk : ds[q];
    }
    hi = central_difference<F>(ds, reps);
    hj = central_difference<F>(ds + H, reps);
}

template <int F>
OGG_DEV void dq_chord_point(const double* r, const cplx* ep, const DpConst& c, double reps, double& hi, double& hj) {
    constexpr int H = F / 2;
    const double inv_rj = 1.0 / c.r_joint;
    double ds[F];
#pragma unroll
    for (int k = 0; k < F; ++k) {
        double ra, rb;
        cplx epa, epb;
        dq_probe_pair<F>(k, r, ep, ra, rb, epa, epb);
        ds[k] = homogeneous_arc(dp_homogeneous(ra, epa, c, inv_rj), dp_homogeneous(rb, epb, c, inv_rj));
    }
    hi = central_difference<F>(ds, reps);
    hj = central_difference<F>(ds + H, reps);
}

template <int N>
OGG_DEV double dq_weight(int k) {  // w[k] of OGG:240 (order 4); order 2 sums the four corners unweighted (OGG:229-232)
    if (N == 4) return (k == 0 || k == 3) ? 1.0 : 5.0;
    return 1.0;
}

// Strip `strip` of chunk `chunk`: all 64 lanes of the wave call; strip, chunk wave-uniform.
template <int N, int ARC>
OGG_DEV void dpole_quad_body(const DpQuadParams& p, long strip, long chunk) {
    constexpr int M = N - 1, F = N, NV = F + 1;
    constexpr unsigned ALL = (1u << (2 * F)) - 1u;
    const int lane = threadIdx.x & 63;
    const DqLane dl = dq_lane(p, M, strip, lane, DQ_COLS);
    if (!dl.active) return;                       // wave-uniform
    const bool valid = dl.valid;
    const long uc = dl.uc, u = dl.uc;             // this lane's lattice column
    const long ci = dl.ci;                        // cell
    const bool out_lane = dl.out_lane, dy_edge = dl.dy_edge;
    const bool sym = (ARC == DP_ARC_CHORD) && p.sym;
    const long cm = dl.cm, pm = dl.pm;
    const DpConst c = dp_const(p.g);
    const double reps = 1.0 / p.eps;
    const double* __restrict__ row_tab = p.row_tab;   // written by the tables kernel of this call, read-only here: scalar loads
    const double* __restrict__ col_tab = p.col_tab;
    cplx ep[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) ep[v] = cplx{col_tab[(v * 2 + 0) * p.n_cols + uc], col_tab[(v * 2 + 1) * p.n_cols + uc]};

    const long r0 = chunk * p.rows_per_chunk;                     // band-local first cell row of the chunk
    const long nc = (p.n_cell_rows - r0 < p.rows_per_chunk) ? p.n_cell_rows - r0 : p.rows_per_chunk;   // cell rows of the chunk (>= 0)
    const long n_lat = M * nc + 1;                                // lattice rows the wave evaluates
    const bool own_top = (p.n_dx_rows > p.n_cell_rows) && (r0 + nc == p.n_cell_rows);
    unsigned long long* words = p.words + (chunk * (M * p.rows_per_chunk + 1)) * p.n_strips;

    // quadrature state of the cell row in progress
    double dyc[N], ysum = 0.0, dxv = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) dyc[k] = 0.0;

    DqPending<F> pend, pend_next;
    unsigned inc0 = 0u, inc1 = ALL, inc0_next = 0u, inc1_next = ALL;   // inclusive strip-local maps of this lane
    double chi = 0.0, chj = 0.0;                                       // chord form: the scale factors themselves
#pragma unroll
    for (int k = 0; k < F; ++k) pend.va[k] = pend.vb[k] = pend.pa[k] = pend.pb[k] = 0.0;
    pend_next = pend;
    // row L + 1 is evaluated between publishing row L's maps and reading its predecessors' (the wait is hidden, the pending state of two
    // rows lives in registers)
#pragma unroll 1
    for (long L = -1; L < n_lat; ++L) {
        // ---- evaluate lattice row L + 1 ------------------------------------------------------------------------------
        if (L + 1 < n_lat) {
            const long row = M * r0 + L + 1;      // band-local lattice row
            double r[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) r[v] = row_tab[v * p.n_rows + row];   // wave-uniform
            if (ARC == DP_ARC_CHORD) {
                dq_chord_point<F>(r, ep, c, reps, pend_next.va[0], pend_next.vb[0]);
            } else {
                dq_literal_probes_lib<F>(r, ep, c, pend_next);
                // maps of this column (OGG:471-474): one bit per probe
                unsigned f0 = 0u, f1 = 0u;
#pragma unroll
                for (int k = 0; k < F; ++k) {
                    const double va = pend_next.va[k], vb = pend_next.vb[k];
                    const double pa = wave_prev(va), pb = wave_prev(vb);
                    bool a0, a1, b0, b1;
                    if (u == 0) {  // column 0 is compared with lon_grid[0,0] of its probe mesh (OGG:463), no state before it
                        constexpr int H = F / 2;
                        const double off = (k < H) ? (double)(k + 1) * p.eps : 0.0;
                        const double i_first = lattice_node(p.q, 0, 0);
                        a0 = a1 = (va - (p.g.lon0 + ((i_first + off) * 360.0) / (double)p.g.ni) > 100);
                        b0 = b1 = (vb - (p.g.lon0 + ((i_first - off) * 360.0) / (double)p.g.ni) > 100);
                    } else {
                        a0 = (va - pa > 100), a1 = (va - (pa - 360) > 100);   // previous column not lowered / lowered
                        b0 = (vb - pb > 100), b1 = (vb - (pb - 360) > 100);
                    }
                    f0 |= ((a0 ? 1u : 0u) << (2 * k)) | ((b0 ? 1u : 0u) << (2 * k + 1));
                    f1 |= ((a1 ? 1u : 0u) << (2 * k)) | ((b1 ? 1u : 0u) << (2 * k + 1));
                }
                // lane 0's column belongs to the strip on the left (its state is the incoming state); columns past the row end: identity
                if ((lane == 0 && u != 0) || !valid) f0 = 0u, f1 = ALL;
                // almost every strip of almost every row holds identity maps only (the constant ones sit at column 0 and where the
                // longitude crosses the cut): the scan is then the identity too
                if (__ballot(f0 != 0u || f1 != ALL) != 0ull) map_scan(f0, f1);
                inc0_next = f0, inc1_next = f1;
                if (lane == 63) lb_publish(words + (L + 1) * p.n_strips + strip, f0, f1);
            }
        }
        // ---- finish lattice row Lf and feed it to the quadrature --------------------------------------------------------
        const long Lf = L;
        if (Lf >= 0) {
            double hi, hj;
            if (ARC == DP_ARC_CHORD) {
                hi = chi, hj = chj;
            } else {
                const unsigned s_in = (strip > 0) ? lb_incoming(words + Lf * p.n_strips, strip, ALL, (int*)(p.ticket + 1)) : 0u;
                dq_literal_finish<F>(pend, map_apply(inc0, inc1, s_in), reps, hi, hj);
            }
            const long k = Lf / M;                // cell row of the chunk this lattice row is the row jj of
            const int jj = (int)(Lf % M);
            const double pr = hi * hj;            // OGG:589
            double ah[N], ap[N];                  // h_i and h_i h_j at this column and its M right-hand neighbours
            ah[0] = hi, ap[0] = pr;
#pragma unroll
            for (int i = 1; i < N; ++i) ap[i] = wave_next(ap[i - 1]);
#pragma unroll
            for (int i = 1; i < N; ++i) ah[i] = (jj == 0) ? wave_next(ah[i - 1]) : 0.0;   // only a cell row's bottom edge feeds dxq
            if (jj == 0 && k > 0) {               // top edge of cell row k - 1: its last Lobatto row
#pragma unroll
                for (int i = 0; i < N; ++i) ysum = ysum + dq_weight<N>(i) * dq_weight<N>(M) * ap[i];   // OGG:244 / 231
                dyc[M] = hj;
                const long out_r = r0 + k - 1;
                const double d = (N == 2) ? (1.0 / 2.0) : (1.0 / 12.0);
                if (out_lane) {
                    const double da = (d * d * ysum) * p.Re * p.Re;                                   // OGG:597
                    p.dxq[out_r * p.g.ni + ci] = dxv;
                    p.daq[out_r * p.g.ni + ci] = da;
                    if (sym) p.dxq[out_r * p.g.ni + cm] = dxv, p.daq[out_r * p.g.ni + cm] = da;
                }
                if (out_lane || dy_edge) {
                    const double dyv = qavg_1d<N>(dyc) * p.Re;                                        // OGG:595,599
                    double* __restrict__ row = p.dyq + out_r * (p.g.ni + 1);
                    row[ci] = dyv;
                    if (sym) {
                        row[pm] = dyv;
                        if (pm == 0) row[p.g.ni] = dyv;
                    }
                }
            }
            if (k < nc) {
                if (jj == 0) {
                    dxv = qavg_1d<N>(ah) * p.Re;                                                       // OGG:594,598
                    ysum = 0.0;
                }
                const double wj = dq_weight<N>(jj);
#pragma unroll
                for (int i = 0; i < N; ++i) ysum = ysum + dq_weight<N>(i) * wj * ap[i];
#pragma unroll
                for (int q = 0; q < N; ++q) dyc[q] = (jj == q) ? hj : dyc[q];
            } else if (own_top && out_lane) {     // L = M nc: the j = ny lattice row, dxq only
                const double dxt = qavg_1d<N>(ah) * p.Re;
                p.dxq[(r0 + nc) * p.g.ni + ci] = dxt;
                if (sym) p.dxq[(r0 + nc) * p.g.ni + cm] = dxt;
            }
        }
        pend = pend_next;
        inc0 = inc0_next, inc1 = inc1_next;
        chi = pend_next.va[0], chj = pend_next.vb[0];
    }
}

// ---- literal form with its pending rows in LDS --------------------------------------------------------------------------------
// The look-back makes every wave wait for the SLOWEST strip to its left, row by row, and the waves of a SIMD then compute and wait in
// phase: with one row of slack (dpole_quad_body above: a second pending row in registers) a quarter of a wave's time is such waiting
// (measured with a shader-clock profile of the walk, profiles/r03_dq_profile.txt: 5.4 k of 23.6 k cycles per lattice row at 1/8 degree, growing from 2.7 k at the left end of a row to 6.5 k
// at the right end; not the latency of the loads -- issuing them early changes nothing).  Here a wave parks what a row's second half needs --
// raw longitude and latitude of the 2 * order probes, 16 doubles per lane, and its strip-local maps -- in a ring of D slots in LDS
// instead of registers and finishes row R - D while the strips to its left have had D rows' time to publish it: D = 2 fits the CU
// (2 workgroups x 4 waves x 2 slots x 8.5 KB = 136 KB of 160 KB) and leaves ONE pending row in registers (32 fewer than before).
template <int N>
constexpr int dq_ring_slot_doubles() { return (4 * N + 1) * 64; }   // [4 F values + the two maps][lane]
template <int N>
constexpr int dq_ring_doubles() { return DQ_WAVES * DQ_RING * dq_ring_slot_doubles<N>(); }

template <int F>
OGG_DEV void dq_literal_finish_ring(const double* slot, unsigned st, double reps, double& hi, double& hj) {   // slot: this lane's column of the slot
    constexpr int H = F / 2;
    double ds[F];
#pragma unroll
    for (int k = 0; k < F; ++k) {
        const double va = slot[(0 * F + k) * 64], vb = slot[(1 * F + k) * 64], pha = slot[(2 * F + k) * 64], phb = slot[(3 * F + k) * 64];
        const double xa = ((st >> (2 * k)) & 1u) ? va - 360 : va;            // OGG:473
        const double xb = ((st >> (2 * k + 1)) & 1u) ? vb - 360 : vb;
        const double lam0 = xa * kPi180, phi0 = pha * kPi180;               // OGG:527-528
        const double lam1 = xb * kPi180, phi1 = phb * kPi180;
        const double dphi = phi1 - phi0, dlam = lam1 - lam0;
        const double sp = sin_tiny(0.5 * dphi), sl = sin_tiny(0.5 * dlam);
        const double d = sp * sp + sl * sl * cos_cap(phi0) * cos_cap(phi1); // OGG:531
        ds[k] = 2.0 * asin_tiny(sqrt(d));
    }
    hi = central_difference<F>(ds, reps);
    hj = central_difference<F>(ds + H, reps);
}

template <int N, int D>
OGG_DEV void dpole_quad_literal_ring(const DpQuadParams& p, long strip, long chunk, double* ring) {   // ring: this WAVE's D slots
    constexpr int M = N - 1, F = N, NV = F + 1, SLOT = dq_ring_slot_doubles<N>();
    constexpr unsigned ALL = (1u << (2 * F)) - 1u;
    const int lane = threadIdx.x & 63;
    const long u0 = strip * DQ_COLS;
    if (u0 >= p.n_cols - 1) return;              // wave-uniform
    const long u = u0 + lane;
    const bool valid = u < p.n_cols;
    const long uc = valid ? u : p.n_cols - 1;
    const long ci = uc / M;
    const bool cell_start = valid && (uc % M == 0);
    const bool out_lane = cell_start && ci < p.g.ni && lane <= 63 - M;
    const bool dy_edge = cell_start && ci == p.g.ni;
    const DpConst c = dp_const(p.g);
    const double reps = 1.0 / p.eps;
    const double* __restrict__ row_tab = p.row_tab;
    const double* __restrict__ col_tab = p.col_tab;
    cplx ep[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) ep[v] = cplx{col_tab[(v * 2 + 0) * p.n_cols + uc], col_tab[(v * 2 + 1) * p.n_cols + uc]};
    const long r0 = chunk * p.rows_per_chunk;
    const long nc = (p.n_cell_rows - r0 < p.rows_per_chunk) ? p.n_cell_rows - r0 : p.rows_per_chunk;
    const long n_lat = M * nc + 1;
    const bool own_top = (p.n_dx_rows > p.n_cell_rows) && (r0 + nc == p.n_cell_rows);
    unsigned long long* words = p.words + (chunk * (M * p.rows_per_chunk + 1)) * p.n_strips;
    int* err = (int*)(p.ticket + 1);
    double* my = ring + lane;

    double dyc[N], ysum = 0.0, dxv = 0.0;        // quadrature state of the cell row in progress
#pragma unroll
    for (int k = 0; k < N; ++k) dyc[k] = 0.0;
    double rn[NV];                                // the radii of the next row to evaluate, loaded one iteration ahead
#pragma unroll
    for (int v = 0; v < NV; ++v) rn[v] = row_tab[v * p.n_rows + M * r0];
    DqAtan atc;
    atc.load(kAtanRed);
#pragma unroll 1
    for (long R = 0; R < n_lat + D; ++R) {
        const long Rf = R - D;                    // the row finished in this iteration
        LbBatch lbb;
        DqPending<F> pend;
        unsigned f0 = 0u, f1 = 0u;
        __builtin_amdgcn_s_setprio(3);   // until the maps are out: other waves wait for them
        if (R < n_lat) {
            // ---- evaluate lattice row R, publish its maps ----------------------------------------------------------------------
            const long row = M * r0 + R;
            double r[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) r[v] = rn[v];
            const long row_n = (R + 1 < n_lat) ? row + 1 : row;
#pragma unroll
            for (int v = 0; v < NV; ++v) rn[v] = row_tab[v * p.n_rows + row_n];   // ahead of the look-back loads below (results return in order)
            constexpr int QS = (DQ_LB_AT > 2 * F) ? 2 * F : DQ_LB_AT;
            dq_literal_probes<F, 0, QS>(r, ep, c, atc, pend);
            if (Rf >= 0 && strip > 0) lb_issue(words + Rf * p.n_strips, strip, lbb);
            dq_literal_probes<F, QS, 2 * F>(r, ep, c, atc, pend);
#pragma unroll
            for (int k = 0; k < F; ++k) {         // maps of this column (OGG:471-474): one bit per probe
                const double va = pend.va[k], vb = pend.vb[k];
                const double pa = wave_prev(va), pb = wave_prev(vb);
                bool a0, a1, b0, b1;
                if (u == 0) {  // column 0 is compared with lon_grid[0,0] of its probe mesh (OGG:463), no state before it
                    constexpr int H = F / 2;
                    const double off = (k < H) ? (double)(k + 1) * p.eps : 0.0;
                    const double i_first = lattice_node(p.q, 0, 0);
                    a0 = a1 = (va - (p.g.lon0 + ((i_first + off) * 360.0) / (double)p.g.ni) > 100);
                    b0 = b1 = (vb - (p.g.lon0 + ((i_first - off) * 360.0) / (double)p.g.ni) > 100);
                } else {
                    a0 = (va - pa > 100), a1 = (va - (pa - 360) > 100);
                    b0 = (vb - pb > 100), b1 = (vb - (pb - 360) > 100);
                }
                f0 |= ((a0 ? 1u : 0u) << (2 * k)) | ((b0 ? 1u : 0u) << (2 * k + 1));
                f1 |= ((a1 ? 1u : 0u) << (2 * k)) | ((b1 ? 1u : 0u) << (2 * k + 1));
            }
            if ((lane == 0 && u != 0) || !valid) f0 = 0u, f1 = ALL;
            if (__ballot(f0 != 0u || f1 != ALL) != 0ull) map_scan(f0, f1);
            if (lane == 63) lb_publish(words + R * p.n_strips + strip, f0, f1);
        }
        __builtin_amdgcn_s_setprio(0);   // the second half of a row is nobody's critical path
        if (Rf >= 0) {
            // ---- finish lattice row Rf, parked D iterations ago, and feed it to the quadrature -----------------------------------
            const double* slot = my + (Rf % D) * SLOT;
            unsigned s_in = 0u;
            if (strip > 0) {
                if (R >= n_lat) lb_issue(words + Rf * p.n_strips, strip, lbb);
                s_in = lb_resolve(words + Rf * p.n_strips, strip, ALL, lbb, err);
            }
            const unsigned long long mw = __double_as_longlong(slot[4 * F * 64]);
            const unsigned inc0 = (unsigned)(mw & 0xffffull), inc1 = (unsigned)((mw >> 16) & 0xffffull);
            double hi, hj;
            dq_literal_finish_ring<F>(slot, map_apply(inc0, inc1, s_in), reps, hi, hj);
            const long Lf = Rf;
            const long k = Lf / M;                // cell row of the chunk this lattice row is the row jj of
            const int jj = (int)(Lf % M);
            const double pr = hi * hj;            // OGG:589
            double ah[N], ap[N];
            ah[0] = hi, ap[0] = pr;
#pragma unroll
            for (int i = 1; i < N; ++i) ap[i] = wave_next(ap[i - 1]);
#pragma unroll
            for (int i = 1; i < N; ++i) ah[i] = (jj == 0) ? wave_next(ah[i - 1]) : 0.0;
            if (jj == 0 && k > 0) {               // top edge of cell row k - 1: its last Lobatto row
#pragma unroll
                for (int i = 0; i < N; ++i) ysum = ysum + dq_weight<N>(i) * dq_weight<N>(M) * ap[i];   // OGG:244 / 231
                dyc[M] = hj;
                const long out_r = r0 + k - 1;
                const double d = (N == 2) ? (1.0 / 2.0) : (1.0 / 12.0);
                if (out_lane) {
                    p.dxq[out_r * p.g.ni + ci] = dxv;
                    p.dyq[out_r * (p.g.ni + 1) + ci] = qavg_1d<N>(dyc) * p.Re;                        // OGG:595,599
                    p.daq[out_r * p.g.ni + ci] = (d * d * ysum) * p.Re * p.Re;                        // OGG:597
                }
                if (dy_edge) p.dyq[out_r * (p.g.ni + 1) + p.g.ni] = qavg_1d<N>(dyc) * p.Re;
            }
            if (k < nc) {
                if (jj == 0) {
                    dxv = qavg_1d<N>(ah) * p.Re;                                                       // OGG:594,598
                    ysum = 0.0;
                }
                const double wj = dq_weight<N>(jj);
#pragma unroll
                for (int i = 0; i < N; ++i) ysum = ysum + dq_weight<N>(i) * wj * ap[i];
#pragma unroll
                for (int q = 0; q < N; ++q) dyc[q] = (jj == q) ? hj : dyc[q];
            } else if (own_top && out_lane) {     // the j = ny lattice row, dxq only
                p.dxq[(r0 + nc) * p.g.ni + ci] = qavg_1d<N>(ah) * p.Re;
            }
        }
        if (R < n_lat) {                          // park row R in the slot that row Rf has just left
            double* slot = my + (R % D) * SLOT;
#pragma unroll
            for (int k = 0; k < F; ++k) {
                slot[(0 * F + k) * 64] = pend.va[k], slot[(1 * F + k) * 64] = pend.vb[k];
                slot[(2 * F + k) * 64] = pend.pa[k], slot[(3 * F + k) * 64] = pend.pb[k];
            }
            slot[4 * F * 64] = __longlong_as_double((long long)((unsigned long long)f0 | ((unsigned long long)f1 << 16)));
        }
    }
    atc.keep();
}

// ---- mesh (OGG:488-518) fused with the unwrap and angle_x (OGG:719-729), rows j0 .. j0+nrows-1 -----------------------------
// A wave owns 62 output columns plus one halo column on either side (the i-1 / i+1 neighbours of angle_x are wave shifts, the
// mesh is never read back); the unwrap state at the wave's first column comes from the strips to its left by the same
// look-back as the quadrature's.  Row-only radius once per row of the workgroup (LDS), column-only e' once per lane.
constexpr int DM_WAVES = 4;
constexpr int DM_OUT = 62;
constexpr int DM_ROWS = 8;

struct DpMeshParams {
    DpGeom g;                    // ni = Ni, nj = Nj
    long j0, nrows;
    double *x, *y, *angle;       // angle may be NULL
    unsigned long long* words;   // [nrows][n_strips]
    unsigned* ticket;            // ticket[0]: work counter, ticket[1]: error flag
    long n_strips, gx;           // strips per row; strip workgroups per row tile
    int rows_per_wg;
};

struct DpMeshLds {
    DpConst c;
    double r[DM_ROWS];
};

OGG_DEV void dpole_mesh_body(const DpMeshParams& m, DpMeshLds& s, long bx, long by) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long Ni = m.g.ni, ni1 = Ni + 1;
    const long jl0 = by * m.rows_per_wg;
    const int nr = (int)((m.nrows - jl0 < m.rows_per_wg) ? (m.nrows - jl0) : m.rows_per_wg);
    if (tid == 0) s.c = dp_const(m.g);
    __syncthreads();
    if (tid < nr) s.r[tid] = dp_row_radius((double)(m.j0 + jl0 + tid), m.g, s.c);
    __syncthreads();
    const long strip = bx * DM_WAVES + wave;
    const long col0 = strip * DM_OUT;                     // first output column of this wave
    if (col0 > Ni) return;                                // wave-uniform
    long i = col0 - 1 + lane;
    const bool in_row = (i >= 0) && (i <= Ni);
    i = i < 0 ? 0 : (i > Ni ? Ni : i);
    const bool out = (lane >= 1) && (lane <= DM_OUT) && in_row;
    const DpConst c = s.c;
    const cplx ep = dp_column((double)i, m.g, c);
    const double seed = m.g.lon0 + (0.0 * 360.0) / (double)Ni;   // lon_grid[0,0] (OGG:463)
    double v_cur = 0.0, ph_cur = 0.0, v_nxt = 0.0, ph_nxt = 0.0;
    unsigned long long cC = 0ull, cV = 0ull, nC = 0ull, nV = 0ull;
#pragma unroll 1
    for (int r = -1; r < nr; ++r) {
        if (r + 1 < nr) {
            AtanCoefs atc;                                // per row: the mesh has waves enough to hide the scalar loads, not registers to spare
            atc.load(kAtanRed);
            dp_point(s.r[r + 1], ep, c, atc, v_nxt, ph_nxt);
            atc.keep();
            const double vp = wave_prev(v_nxt);
            unsigned f0, f1;
            if (lane == 0 || !in_row) {
                f0 = 0u, f1 = 1u;                         // the halo column on the left belongs to the previous strip
            } else if (i == 0) {
                f0 = f1 = (v_nxt - seed > 100) ? 1u : 0u; // OGG:471-472
            } else {
                f0 = (v_nxt - vp > 100) ? 1u : 0u;        // OGG:473-474
                f1 = (v_nxt - (vp - 360) > 100) ? 1u : 0u;
            }
            // constant maps of the wave (C) and their values (V); the strip's own map covers its columns col0 .. col0+61 = lanes 1 .. 62
            nC = __ballot(f0 == f1), nV = __ballot(f0 != 0u);
            if (lane == DM_OUT) {
                const unsigned long long own = nC & 0x7ffffffffffffffeull;
                const unsigned v = own ? (unsigned)((nV >> (63 - __builtin_clzll(own))) & 1ull) : 0u;
                lb_publish(m.words + (jl0 + r + 1) * m.n_strips + strip, own ? v : 0u, own ? v : 1u);
            }
        }
        if (r >= 0) {
            const long jl = jl0 + r;
            const unsigned s_in = (strip > 0) ? lb_incoming1(m.words + jl * m.n_strips, strip, (int*)(m.ticket + 1)) : 0u;
            const unsigned long long upto = cC & (~0ull >> (63 - lane));          // constant maps at or before this lane
            const unsigned st = upto ? (unsigned)((cV >> (63 - __builtin_clzll(upto))) & 1ull) : s_in;
            const double lam = st ? v_cur - 360 : v_cur;  // OGG:473
            const double phi = ph_cur;
            if (out) {
                m.x[jl * ni1 + i] = lam;
                m.y[jl * ni1 + i] = phi;
            }
            if (m.angle) {                                // OGG:725-728, literal
                const double xl = wave_prev(lam), xr = wave_next(lam);
                const double yl = wave_prev(phi), yr = wave_next(phi);
                const double cy = cos(phi * kPi180);
                double a;
                if (i == 0)
                    a = atan2_lib(yr - phi, (xr - lam) * cy);
                else if (i == Ni)
                    a = atan2_lib(phi - yl, (lam - xl) * cy);
                else
                    a = atan2_lib(yr - yl, (xr - xl) * cy);
                if (out) m.angle[jl * ni1 + i] = div_pi180(a);
            }
        }
        v_cur = v_nxt, ph_cur = ph_nxt;
        cC = nC, cV = nV;
    }
}

// reset of the look-back words and the ticket of one mesh call (its own small launch, or part of launch A of the pass)
OGG_DEV void dpole_mesh_reset_body(const DpMeshParams& m, long bx, long n_blocks) {
    const long k = bx * blockDim.x + threadIdx.x;
    if (k == 0) m.ticket[0] = 0u, m.ticket[1] = 0u;
    const long n_words = m.nrows * m.n_strips;
    for (long w = k; w < n_words; w += n_blocks * blockDim.x) m.words[w] = 0ull;
}

// workspace: [ticket, error flag, pad (16 B)] [look-back words: nrows x n_strips]
inline long dm_strips(long Ni) { return (Ni + 1 + DM_OUT - 1) / DM_OUT; }
inline size_t dm_workspace_bytes(long Ni, long nrows) { return 16 + (size_t)(nrows * dm_strips(Ni)) * sizeof(unsigned long long); }

inline int plan_dmesh(const DpGeom& g, long j0, long nrows, double* x, double* y, double* angle, void* ws, long ws_bytes, DpMeshParams& m) {
    const size_t need = dm_workspace_bytes(g.ni, nrows);
    OGG_REQUIRE(ws && (size_t)ws_bytes >= need, OGG_EARG, "displaced-pole mesh workspace too small: %ld < %zu bytes", ws_bytes, need);
    m.g = g;
    m.j0 = j0, m.nrows = nrows;
    m.x = x, m.y = y, m.angle = angle;
    m.ticket = static_cast<unsigned*>(ws);
    m.words = reinterpret_cast<unsigned long long*>(static_cast<char*>(ws) + 16);
    m.n_strips = dm_strips(g.ni);
    m.gx = (m.n_strips + DM_WAVES - 1) / DM_WAVES;
    long rpw = DM_ROWS;
    if (const char* e = getenv("OGG_DPMESH_ROWS")) rpw = atol(e) < 1 ? 1 : (atol(e) > DM_ROWS ? DM_ROWS : atol(e));
    m.rows_per_wg = (int)rpw;
    return OGG_OK;
}
inline long dm_blocks(const DpMeshParams& m) { return m.gx * ((m.nrows + m.rows_per_wg - 1) / m.rows_per_wg); }
inline long dm_reset_blocks(const DpMeshParams& m) { return (m.nrows * m.n_strips / 4 + 256) / 256; }

}  // namespace
