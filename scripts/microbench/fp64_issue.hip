// fp64 VALU issue rate of gfx950 under sustained load: K independent FMA chains per lane, W waves per SIMD.
// Tells what "100 % of the fp64 issue slots" means in wall-clock terms (the effective clock under an fp64-heavy kernel).
// build + run: hipcc --offload-arch=gfx950 -O3 -o /tmp/fi scripts/microbench/fp64_issue.hip && /tmp/fi
#include <hip/hip_runtime.h>
#include <cstdio>

template <int K>
__global__ __launch_bounds__(256) void fma_chain(double* out, double a, double b, int iters) {
    double x[K];
#pragma unroll
    for (int k = 0; k < K; ++k) x[k] = (double)(threadIdx.x + k);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int k = 0; k < K; ++k) x[k] = fma(x[k], a, b);
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) s += x[k];
    out[(long)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K>
void run(int waves_per_simd, double* out) {
    const int iters = 4000;
    const int blocks = 256 /*CUs*/ * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    fma_chain<K><<<blocks, 256>>>(out, 0.999999, 1e-9, iters);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) fma_chain<K><<<blocks, 256>>>(out, 0.999999, 1e-9, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double wave_instr_per_simd = (double)waves_per_simd * iters * 16 * K;
    const double cycles_at_4 = wave_instr_per_simd * 4;
    printf("chains %d, waves/SIMD %d: %.3f ms -> %.2f GHz-equivalent (4 cycles per wave64 fp64 FMA), %.1f TFLOP/s\n", K, waves_per_simd, ms,
           cycles_at_4 / (ms * 1e-3) / 1e9, 2.0 * 64 * wave_instr_per_simd * 1024 / (ms * 1e-3) / 1e12);
}

int main() {
    double* out; hipMalloc(&out, 256L * 8 * 256 * 8);
    run<1>(1, out); run<1>(2, out); run<1>(4, out); run<1>(8, out);
    run<4>(1, out); run<4>(2, out); run<4>(3, out); run<4>(4, out); run<8>(2, out);
    return 0;
}
