// Raw accuracy of v_rcp_f64 / v_rsq_f64 on gfx950 and of the Newton variants used in ogg_math.h.
// build + run: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/ra scripts/microbench/rcp_rsq_accuracy.hip && /tmp/ra
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k(const double* x, double* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double y = __builtin_amdgcn_rcp(v);
    o[i] = y;                                        // raw rcp
    double e = fma(-v, y, 1.0); y = fma(y, e, y);
    o[n + i] = y;                                    // one Newton step
    e = fma(-v, y, 1.0); y = fma(y, e, y);
    o[2 * n + i] = y;                                // two Newton steps
    double r = __builtin_amdgcn_rsq(v);
    o[3 * n + i] = r;                                // raw rsq
    double g = v * r, h = 0.5 * r;
    double rr = fma(-h, g, 0.5);
    g = fma(g, rr, g);
    o[4 * n + i] = g;                                // sqrt: one coupled step
    const double d = fma(-g, g, v);
    o[5 * n + i] = fma(d, h, g);                     // + residual correction (sqrt_nr)
}

int main() {
    const int n = 1 << 22;
    std::vector<double> x(n), o(6 * (size_t)n);
    unsigned long long s = 88172645463325252ULL;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = ldexp(1.0 + (double)(s >> 11) / 9007199254740992.0, (int)(s % 40) - 20); }
    double *dx, *dout; hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * (size_t)n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout, n);
    hipMemcpy(o.data(), dout, 6 * (size_t)n * 8, hipMemcpyDeviceToHost);
    const char* name[6] = {"rcp raw", "rcp + 1 Newton", "rcp + 2 Newton", "rsq raw", "sqrt, 1 coupled step", "sqrt_nr (with residual step)"};
    for (int q = 0; q < 6; ++q) {
        double worst = 0;
        for (int i = 0; i < n; ++i) {
            const long double ref = (q < 3) ? 1.0L / x[i] : ((q == 3) ? 1.0L / sqrtl((long double)x[i]) : sqrtl((long double)x[i]));
            const double rel = (double)fabsl(((long double)o[(size_t)q * n + i] - ref) / ref);
            if (rel > worst) worst = rel;
        }
        printf("%-30s max rel error %.3e = 2^%.1f\n", name[q], worst, log2(worst));
    }
    return 0;
}
