// Device-side scalar helpers shared by all kernels.  gfx950 only.
//
// Arithmetic contract: every kernel in this library is compiled with -ffp-contract=off and follows the operation
// order of the reference script (ocean_grid_generator.py, "OGG:<line>" below) so that the only differences
// from a CPU evaluation of the same formulas are the last-ulp differences of the transcendental functions
// (ocml on the device).  Where numpy itself fuses (its complex multiply is fma(ar,br,-(ai*bi)), its complex
// absolute is a*sqrt(fma(r,r,1))), the explicit fma() is written out here.
#pragma once
#include <hip/hip_runtime.h>

#define OGG_DEV __device__ __forceinline__
#define OGG_HD __host__ __device__ __forceinline__   // index arithmetic shared by a kernel and its host-side check

namespace ogg {

constexpr double kPi = 3.141592653589793;        // numpy.pi
constexpr double kPi180 = kPi / 180.0;            // OGG:13  PI_180
constexpr double k180Pi = 180.0 / kPi;            // OGG:301 (180/np.pi), also np.angle(deg=True)
constexpr double kHuge = 1.0e30;                  // OGG:16
constexpr double kReDefault = 6371.0e3;           // OGG:15

// numpy.mod(a, 360.0): fmod, then + 360 if negative (the result carries the sign of the divisor).  For |a| < 2^40 the same
// value comes from q = floor(a/360) and ONE fma: a - 360 q is exactly representable, and where numpy's sum fmod(a) + 360
// rounds, the fma rounds the same real number.  q from a*(1/360) can be off by one next to a multiple of 360; r then falls
// outside [0, 360) and is redone.  Checked against the fmod form on 2e8 operands (random, and within an ulp of multiples of
// 360).  8 instructions instead of ocml's fmod loop.
OGG_DEV double pymod360(double a) {
    double r;
    if (fabs(a) < 1.0e12) {
        double q = floor(a * (1.0 / 360.0));
        r = fma(-q, 360.0, a);
        if (r < 0.0) {
            q -= 1.0;
            r = fma(-q, 360.0, a);
        } else if (r >= 360.0) {
            // q one too small -- or not: for -2.8e-14 < a < 0 the exact a + 360 ROUNDS to 360.0, which is numpy's answer too
            // (fmod(a) + 360, rounded); the quotient was right then and the retry would come out negative
            const double r2 = fma(-(q + 1.0), 360.0, a);
            r = (r2 >= 0.0) ? r2 : r;
        }
    } else {
        r = fmod(a, 360.0);
        if (r < 0.0) r += 360.0;
    }
    return (r == 0.0) ? 0.0 : r;
}

// a / PI_180 for finite a, correctly rounded -- the same bits as the IEEE division a / kPi180 -- in 3 instructions instead
// of the 11 of a full fp64 divide (division by a constant: q = a rc, then one residual correction; checked against the
// divide instruction on 3e8 random operands over 40 binades).
OGG_DEV double div_pi180(double a) {
    constexpr double rc = 1.0 / kPi180;
    const double q = a * rc;
    return fma(fma(-q, kPi180, a), rc, q);
}

// OGG:682-684 mdist: positive distance modulo 360.
OGG_DEV double mdist(double x1, double x2) {
    return fmin(pymod360(x1 - x2), pymod360(x2 - x1));
}

// mdist from ONE reduction.  x2 - x1 = -(x1 - x2) exactly, fmod is exact and odd, so with f = fmod(|d|, 360) in [0, 360) the two
// numpy.mod of OGG:684 are f and fl(360 - f) (in the order of the sign of d; both 0 when f = 0), and their minimum is
// fmin(f, 360 - f): half the instructions of two reductions, the same bits (ogg_libm_check_dev, which = 13, against the fmod form).
// A wave with |d| >= 1e12, an infinity or a NaN takes mdist, behind one ballot.
OGG_DEV double mdist_one(double x1, double x2) {
    const double a = fabs(x1 - x2);
    if (__builtin_expect(__ballot(!(a < 1.0e12)) != 0ull, 0)) return mdist(x1, x2);
    const double q = floor(a * (1.0 / 360.0));
    const double r = fma(-q, 360.0, a);                      // exact: a - 360 q is representable
    const double qc = (r < 0.0) ? q - 1.0 : q + 1.0;         // the quotient from a * (1/360) can be off by one next to a multiple of 360
    const double rc = fma(-qc, 360.0, a);
    const double f = (r < 0.0 || r >= 360.0) ? rc : r;       // (exact values: r >= 360 means q was one too small)
    return fmin(f, 360.0 - f);
}

// IEEE 1/x and sqrt(x) -- correctly rounded, the SAME bits as the compiler's expansions of `1.0 / x` and `sqrt(x)` -- for operands
// that need none of the scaling and special-case handling those expansions carry (2^-700 <= x <= 2^700): the same seed and the same
// fma steps, without v_div_scale / v_div_fmas / v_div_fixup (7 instead of 11 instructions) and without the range test, the two
// ldexp and the zero / infinity selects of the square root (10 instead of 21).  Bit-identity is a test (ogg_libm_check_dev, 4e7
// operands each).  Used where the reference divides or takes a root of a quantity whose range is known (the cap mesh: 1 + a b in
// [1, 1e33], its reciprocal in [1e-33, 1]; numpy's complex absolute: 1 + r^2 in [1, 2]).
OGG_DEV double rcp_ieee_normal(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    y = fma(fma(-x, y, 1.0), y, y);
    return fma(fma(-x, y, 1.0), y, y);
}

OGG_DEV double sqrt_ieee_normal(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g), h = fma(h, r, h);
    g = fma(fma(-g, g, x), h, g);
    return fma(fma(-g, g, x), h, g);
}

// a / b for operands within 2^-300 .. 2^300 in magnitude (a may also be a signed zero): the compiler's own expansion of the IEEE
// division -- reciprocal seed, two coupled refinements, quotient, one residual correction -- without v_div_scale / v_div_fmas /
// v_div_fixup (9 instead of 11 instructions; the same bits: v_div_scale only acts when an operand is near the ends of the exponent
// range or the exponents differ by 768 or more, without scaling v_div_fmas IS the fma, and of what v_div_fixup repairs only the sign
// of a zero quotient can occur here -- the residual step turns -0 into +0 --, which the quotient's own sign restores).  Bit-identity
// is a test (ogg_libm_check_dev, which = 5).
OGG_DEV double div_ieee_normal(double a, double b) {
    double y = __builtin_amdgcn_rcp(b);
    y = fma(fma(-b, y, 1.0), y, y);
    y = fma(fma(-b, y, 1.0), y, y);
    const double q = a * y;
    return copysign(fma(fma(-b, q, a), y, q), q);
}

struct cplx {
    double re, im;
};

// numpy complex128 multiply as the SIMD loop evaluates it (checked bit-for-bit on the build host).
OGG_DEV cplx cmul(cplx a, cplx b) {
    cplx o;
    o.re = fma(a.re, b.re, -(a.im * b.im));
    o.im = fma(a.re, b.im, a.im * b.re);
    return o;
}

// numpy complex128 divide: Smith's algorithm, no fma.
OGG_DEV cplx cdiv(cplx a, cplx b) {
    cplx o;
    if (fabs(b.re) >= fabs(b.im)) {
        const double rat = b.im / b.re;
        const double scl = 1.0 / (b.re + b.im * rat);
        o.re = (a.re + a.im * rat) * scl;
        o.im = (a.im - a.re * rat) * scl;
    } else {
        const double rat = b.re / b.im;
        const double scl = 1.0 / (b.im + b.re * rat);
        o.re = (a.re * rat + a.im) * scl;
        o.im = (a.im * rat - a.re) * scl;
    }
    return o;
}

// numpy.absolute(complex128): max * sqrt(1 + (min/max)^2) with one fma.
OGG_DEV double cabs_np(cplx w) {
    const double ar = fabs(w.re), ai = fabs(w.im);
    const double a = fmax(ar, ai), b = fmin(ar, ai);
    if (a == 0.0) return 0.0;
    const double r = b / a;
    return a * sqrt_ieee_normal(fma(r, r, 1.0));   // the argument lies in [1, 2]: IEEE sqrt, the same bits, without its scaling / special cases
}
// 1/x and 1/sqrt(x) for NORMAL, positive x from the hardware seed (v_rcp_f64 / v_rsq_f64, relative error e <= 2^-24 measured,
// scripts/microbench/rcp_rsq_accuracy.hip) and ONE third-order step -- 1/x = y (1 + e + e^2 + ...), x^(-1/2) = y (1 + e/2 + 3 e^2/8 + ...)
// with e = 1 - x y (resp. 1 - x y^2) exact from the fma -- which leaves e^3 = 2^-72 of truncation: 0.5 / 1.0 ulp at worst over 1.7e7
// operands of 680 binades (two Newton steps: 0.5 / 1.75 ulp and one / two more instructions; that micro-benchmark keeps them for
// comparison).  No denormal scaling, no fix-up.  Only used where a kernel documents that it departs from the reference's literal
// operation sequence (bp_point_fast, the mesh's tan(acos(A)/2), atan2_angle); everything else uses IEEE division and sqrt.
OGG_DEV double rcp_c3(double x) {
    const double y = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, y, 1.0);
    return fma(y, fma(e, e, e), y);
}

OGG_DEV double rsqrt_c3(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * y), y, 1.0);
    return fma(y * e, fma(e, 0.375, 0.5), y);
}

// Polynomial coefficients as SCALAR operands.  A 64-bit literal cannot be an operand of a gfx9 VALU instruction: the compiler
// materialises each coefficient with two v_mov_b32 in front of its fma, so a Horner step costs three vector instructions instead of
// one (ocml's own polynomials pay the same).  Kept in a __constant__ table the coefficients arrive by scalar loads (one
// s_load_dwordx16 per 8) and feed v_fma_f64 as scalar operands.  Two details make the compiler do it:
//   * the table pointer passes through an empty asm, or a table with internal linkage is folded back into literals;
//   * every coefficient is "used" once more by an empty asm behind the chain: a scalar addend that dies at its fma is otherwise
//     copied into vector registers for the two-address v_fmac_f64 form.
typedef const double __attribute__((address_space(4))) * coef_table_t;

OGG_DEV coef_table_t scalar_table(const double* t) {
    coef_table_t p = (coef_table_t)t;
    asm("" : "+s"(p));
    return p;
}

// N coefficients of a __constant__ table in scalar registers.  load() once, eval() as often as wanted, keep() behind the last use
// (all three in the same function after inlining).  A kernel that evaluates one polynomial many times per iteration of a long loop
// (the displaced-pole quadrature: 16 arctangents per lattice point) loads ONCE in front of the loop and keeps behind it -- loaded per
// use, sixteen sets of forty scalar registers want to be in flight at once and spill into vector-register lanes.
template <int N>
struct ScalarPoly {
    double k[N];
    OGG_DEV void load(const double* table) {
        const coef_table_t c = scalar_table(table);
#pragma unroll
        for (int i = 0; i < N; ++i) k[i] = c[i];
    }
    // k[N-1] z^(N-1) + ... + k[1] z + k[0], Horner
    OGG_DEV double eval(double z) const {
        double p = k[N - 1];
#pragma unroll
        for (int i = N - 2; i >= 0; --i) p = fma(p, z, k[i]);
        return p;
    }
    OGG_DEV void keep() const {
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("" ::"s"(k[i]));
    }
};

// one-shot form: load, evaluate, keep
template <int N>
OGG_DEV double horner_scalar(const double* table, double z) {
    ScalarPoly<N> c;
    c.load(table);
    const double p = c.eval(z);
    c.keep();
    return p;
}

// atan and atan2 with the SAME bits as the device library's (ROCm 7.2 ocml: __ocml_atan_f64, __ocml_atan2_f64 and their shared
// 20-coefficient odd polynomial __ocmlpriv_atanred_f64, read from the library's bitcode), restated operation for operation so that the
// coefficients are scalar operands: 20 instead of 60 vector instructions per polynomial.  atan_lib: all arguments.  atan2_lib: FINITE
// arguments (the library's extra selects for infinities and NaNs are left out; signed zeros and atan2(0, 0) are kept).  Bit-identity
// with atan() / atan2() is tested on 4e7 arguments each (ogg_libm_check_dev), so every parity number measured with ocml's functions
// stands.
static __constant__ double kAtanRed[20] = {-0x1.5555555555523p-2, 0x1.99999999952ccp-3,  -0x1.2492492376b7dp-3, 0x1.c71c717e1913cp-4,
                                    -0x1.745d119378e4fp-4, 0x1.3b13657b87036p-4,  -0x1.110e48b207f05p-4, 0x1.e1bb48427b883p-5,
                                    -0x1.ae5ce6a214619p-5, 0x1.82d5d6ef28734p-5,  -0x1.59976e82d3ff0p-5, 0x1.2c15b5711927ap-5,
                                    -0x1.e9ae6fc27006ap-6, 0x1.67e295f08b19fp-6,  -0x1.c6ea4a57d9582p-7, 0x1.d6d43a595c56fp-8,
                                    -0x1.7952daf56de9bp-9, 0x1.b2bb069efb384p-11, -0x1.3e260bd3237f4p-13, 0x1.ba404b5e68a13p-17};

typedef ScalarPoly<20> AtanCoefs;   // AtanCoefs c; c.load(kAtanRed); ... atan2_lib(y, x, c) ...; c.keep();

// ... and with its coefficients resident in VECTOR registers (40 of them), for a kernel that has vector registers to spare but no scalar
// ones (a wave limited to two per SIMD by its other state): load() once in front of the loop.  The fma is written as the three-address
// VOP3 instruction -- left to itself the compiler takes the two-address v_fmac_f64 and copies the coefficient in front of every step
// (v_mov_b64 + v_fmac_f64: two vector instructions per Horner step; measured 3.8 against 2.1 ns per step, scripts/microbench/horner_issue.hip).
struct AtanVgpr {
    double k[20];
    OGG_DEV void load(const double* table) {
#pragma unroll
        for (int i = 0; i < 20; ++i) {
            k[i] = table[i];
            asm volatile("" : "+v"(k[i]));
        }
    }
    OGG_DEV double eval(double z) const {
        double p;   // the whole chain in one asm statement (the compiler pads every asm statement with wait states of its own)
        asm("v_fma_f64 %0, %21, %1, %20\n\t"
            "v_fma_f64 %0, %0, %1, %19\n\t"
            "v_fma_f64 %0, %0, %1, %18\n\t"
            "v_fma_f64 %0, %0, %1, %17\n\t"
            "v_fma_f64 %0, %0, %1, %16\n\t"
            "v_fma_f64 %0, %0, %1, %15\n\t"
            "v_fma_f64 %0, %0, %1, %14\n\t"
            "v_fma_f64 %0, %0, %1, %13\n\t"
            "v_fma_f64 %0, %0, %1, %12\n\t"
            "v_fma_f64 %0, %0, %1, %11\n\t"
            "v_fma_f64 %0, %0, %1, %10\n\t"
            "v_fma_f64 %0, %0, %1, %9\n\t"
            "v_fma_f64 %0, %0, %1, %8\n\t"
            "v_fma_f64 %0, %0, %1, %7\n\t"
            "v_fma_f64 %0, %0, %1, %6\n\t"
            "v_fma_f64 %0, %0, %1, %5\n\t"
            "v_fma_f64 %0, %0, %1, %4\n\t"
            "v_fma_f64 %0, %0, %1, %3\n\t"
            "v_fma_f64 %0, %0, %1, %2"
            : "=&v"(p)
            : "v"(z), "v"(k[0]), "v"(k[1]), "v"(k[2]), "v"(k[3]), "v"(k[4]), "v"(k[5]), "v"(k[6]), "v"(k[7]), "v"(k[8]), "v"(k[9]), "v"(k[10]),
              "v"(k[11]), "v"(k[12]), "v"(k[13]), "v"(k[14]), "v"(k[15]), "v"(k[16]), "v"(k[17]), "v"(k[18]), "v"(k[19]));
        return p;
    }
    OGG_DEV void keep() const {
#pragma unroll
        for (int i = 0; i < 20; ++i) asm volatile("" ::"v"(k[i]));
    }
};

template <class C>
OGG_DEV double atanred_lib(double v, const C& c) {   // |v| <= 1
    const double t = v * v;
    return fma(v, t * c.eval(t), v);
}

template <class C>
OGG_DEV double atan_lib(double x, const C& c) {
    const double v = fabs(x);
    const bool g = v > 1.0;
    const double a = atanred_lib(g ? 1.0 / v : v, c);
    const double r = g ? fma(0x1.dd9ad336a0500p-1, 0x1.af154eeb562d6p+0, -a) : a;   // pi/2 as an exact product
    return copysign(r, x);
}

// atan_lib with the reciprocal of arguments above 1 behind a wave-uniform branch: a wave whose arguments are all <= 1 in magnitude (the
// tangent of a cap's colatitude: every wave of a southern cap) does not pay for a division whose result a select would discard
template <class C>
OGG_DEV double atan_lib_wave(double x, const C& c) {
    if (__builtin_expect(__ballot(fabs(x) > 1.0) != 0ull, 0)) return atan_lib(x, c);
    return copysign(atanred_lib(fabs(x), c), x);
}

template <class C>
OGG_DEV double atan2_lib(double y, double x, const C& c) {   // finite arguments
    const double ay = fabs(y), ax = fabs(x);
    const double mx = fmax(ax, ay), mn = fmin(ax, ay);
    double a = atanred_lib(mn / mx, c);
    const bool xneg = __double2hiint(x) < 0;     // the sign BIT: -0.0 counts
    a = (ax < ay) ? 0x1.921fb54442d18p+0 - a : a;
    a = xneg ? 0x1.921fb54442d18p+1 - a : a;
    a = (y == 0.0) ? (xneg ? 0x1.921fb54442d18p+1 : 0.0) : a;
    return copysign(a, y);
}

// atan2 for ANY arguments -- infinities and NaNs as the library answers them (both infinite: +-pi/4 or +-3pi/4; a NaN: NaN) -- for
// kernels that take a caller's arrays (the generic stencil kernel behind ogg_angle_x / ogg_grid_metrics_midas): two more selects
template <class C>
OGG_DEV double atan2_lib_any(double y, double x, const C& c) {
    double a = atan2_lib(y, x, c);
    const bool xinf = fabs(x) == __builtin_inf(), yinf = fabs(y) == __builtin_inf();
    const double q = (__double2hiint(x) < 0) ? 0x1.2d97c7f3321d2p+1 : 0x1.921fb54442d18p-1;   // 3 pi / 4, pi / 4
    a = (xinf && yinf) ? copysign(q, y) : a;
    return (x != x || y != y) ? __builtin_nan("") : a;
}

// one-shot forms (coefficients loaded for this one call)
OGG_DEV double atan_lib(double x) {
    AtanCoefs c;
    c.load(kAtanRed);
    const double r = atan_lib(x, c);
    c.keep();
    return r;
}
OGG_DEV double atan2_lib(double y, double x) {
    AtanCoefs c;
    c.load(kAtanRed);
    const double r = atan2_lib(y, x, c);
    c.keep();
    return r;
}
OGG_DEV double atan2_lib_any(double y, double x) {
    AtanCoefs c;
    c.load(kAtanRed);
    const double r = atan2_lib_any(y, x, c);
    c.keep();
    return r;
}

// Neighbour lanes of a wave64 through DPP wave shifts (gfx9: wave_shr:1 = 0x138, wave_shl:1 = 0x130): two VALU moves per
// double instead of two ds_bpermute round trips through the LDS crossbar.  The end lanes keep their own value, like
// __shfl_up / __shfl_down with delta 1.
OGG_DEV double wave_prev(double x) {  // lane l gets the value of lane l-1
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
OGG_DEV double wave_next(double x) {  // lane l gets the value of lane l+1
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
OGG_DEV int wave_next(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x130, 0xf, 0xf, false); }

// Workgroups are handed to the 8 XCDs of the chip round-robin: workgroup b (linear index in dispatch order) runs on XCD
// b % 8.  xcd_contiguous renumbers the n workgroups of a launch so that the workgroups of one XCD get CONSECUTIVE virtual
// indices: with rows as the slow index of the virtual numbering every XCD then writes a contiguous eighth of the rows, which
// the HBM write path rewards (+15 % on strided multi-array writes, scripts/microbench/write_patterns.hip pattern b2).
OGG_DEV long xcd_contiguous(long b, long n) {
    const long x = b % 8, r = n % 8;
    return x * (n / 8) + ((x < r) ? x : r) + b / 8;
}

// Gauss-Lobatto node weights of OGG:191-204, computed on the host in IEEE double and passed by value.
struct QuadNodes {
    double a[5];
    double b[5];
};

// OGG:145 / OGG:155: node k of cell c  ->  b[k]*c + a[k]*(c+1)
OGG_DEV double lattice_node(const QuadNodes& q, int k, long c) {
    return q.b[k] * (double)c + q.a[k] * (double)(c + 1);
}

}  // namespace ogg
