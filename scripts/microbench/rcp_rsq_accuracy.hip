// Raw accuracy of v_rcp_f64 / v_rsq_f64 on gfx950 and of the Newton / third-order variants of ogg_math.h, in ulps of the result
// against long double on the host (operands over 2^-340 .. 2^340: the quadrature takes rsqrt of products up to ~1e105).
// build + run: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I ocean_model_grid_generator_amd/csrc -o /tmp/ra
//              scripts/microbench/rcp_rsq_accuracy.hip && /tmp/ra
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

#include "ogg_math.h"

namespace ogg {   // the Newton variants the kernels used before the third-order steps, kept here for comparison
OGG_DEV double rcp_nr(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}
OGG_DEV double rsqrt_nr(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * fma(-hx * y, y, 1.5);
    return y * fma(-hx * y, y, 1.5);
}
OGG_DEV double sqrt_nr(double x) {   // one coupled step + one residual step
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    const double h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    const double d = fma(-g, g, x);
    return fma(d, h, g);
}
OGG_DEV double sqrt_c3(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double g = x * y;
    const double e = fma(-g, y, 1.0);
    return fma(g * e, fma(e, 0.375, 0.5), g);
}
}  // namespace ogg

constexpr int NV = 9;

__global__ void k(const double* x, double* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    o[i] = __builtin_amdgcn_rcp(v);
    double y = __builtin_amdgcn_rcp(v);
    double e = fma(-v, y, 1.0);
    o[n + i] = fma(y, e, y);                         // one Newton step
    o[2 * (size_t)n + i] = ogg::rcp_nr(v);
    o[3 * (size_t)n + i] = ogg::rcp_c3(v);
    o[4 * (size_t)n + i] = __builtin_amdgcn_rsq(v);
    o[5 * (size_t)n + i] = ogg::rsqrt_nr(v);
    o[6 * (size_t)n + i] = ogg::rsqrt_c3(v);
    o[7 * (size_t)n + i] = ogg::sqrt_nr(v);
    o[8 * (size_t)n + i] = ogg::sqrt_c3(v);
}

int main() {
    const int n = 1 << 24;
    std::vector<double> x(n), o(NV * (size_t)n);
    unsigned long long s = 88172645463325252ULL;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13, s ^= s >> 7, s ^= s << 17;
        double m = 1.0 + (double)(s >> 11) / 9007199254740992.0;
        if (i % 16 == 0) m = 1.0 + (double)((s >> 11) & 1023) / 9007199254740992.0;   // just above a power of two
        if (i % 16 == 1) m = 2.0 - (double)(1 + ((s >> 11) & 1023)) / 9007199254740992.0;   // just below
        x[i] = ldexp(m, (int)((s >> 3) % 681) - 340);
    }
    double *dx, *dout;
    hipMalloc(&dx, n * 8);
    hipMalloc(&dout, NV * (size_t)n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout, n);
    hipMemcpy(o.data(), dout, NV * (size_t)n * 8, hipMemcpyDeviceToHost);
    const char* name[NV] = {"rcp raw", "rcp + 1 Newton", "rcp_nr", "rcp_c3", "rsq raw", "rsqrt_nr", "rsqrt_c3", "sqrt_nr", "sqrt_c3"};
    for (int q = 0; q < NV; ++q) {
        double worst = 0, worst_ulp = 0;
        for (int i = 0; i < n; ++i) {
            const long double xv = x[i];
            const long double ref = (q < 4) ? 1.0L / xv : ((q < 7) ? 1.0L / sqrtl(xv) : sqrtl(xv));
            const double got = o[(size_t)q * n + i];
            const double rel = (double)fabsl(((long double)got - ref) / ref);
            int ex;
            frexp((double)ref, &ex);
            const double ulps = (double)fabsl((long double)got - ref) / ldexp(1.0, ex - 53);
            if (rel > worst) worst = rel;
            if (ulps > worst_ulp) worst_ulp = ulps;
        }
        printf("%-16s max rel error %.3e = 2^%.1f   %.3f ulp\n", name[q], worst, log2(worst), worst_ulp);
    }
    return 0;
}
