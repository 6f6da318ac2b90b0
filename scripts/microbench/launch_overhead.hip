// What a launch costs on this box, whatever is in it: back-to-back kernels on one stream, wall time per kernel (host clock around
// K launches + one synchronize) for (a) an empty kernel of 1, 106 and 1400 workgroups, (b) a kernel whose threads run a dependent
// chain of n fp64 fma (the shape of launch A: ~110 workgroups, one libm chain per thread).
//   hipcc --offload-arch=gfx950 -O3 -o launch_overhead launch_overhead.hip && ./launch_overhead
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void empty_kernel(double* out) {
    if (out && threadIdx.x == 1024) out[0] = 1.0;
}

__global__ void chain_kernel(double* out, int n, double a) {
    double x = threadIdx.x * 1e-9 + 1.0;
    for (int k = 0; k < n; ++k) x = fma(x, a, 1e-9);
    if (x == 123.456) out[0] = x;
}

struct Big {   // the size of the pass kernels' parameter blocks (1.6-2 KB)
    double v[240];
};
__global__ void empty_big_kernel(Big b, double* out) {
    if (out && threadIdx.x == 1024) out[0] = b.v[7];
}
__global__ void empty_ptr_kernel(const Big* b, double* out) {
    if (out && threadIdx.x == 1024) out[0] = b->v[7];
}

template <class F>
double per_launch_us(F launch, int reps) {
    for (int k = 0; k < 50; ++k) launch();
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < reps; ++k) launch();
    hipDeviceSynchronize();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
}

int main() {
    double* d;
    hipMalloc(&d, 4096);
    hipStream_t st;
    hipStreamCreate(&st);
    const int reps = 2000;
    for (int wg : {1, 106, 1400, 8192})
        printf("empty kernel, %5d workgroups of 256: %.2f us per launch\n", wg, per_launch_us([&] { empty_kernel<<<wg, 256, 0, st>>>(d); }, reps));
    for (int n : {100, 500, 1000, 2000, 4000})
        printf("chain of %4d dependent fp64 fma, 106 workgroups of 256: %.2f us per launch\n", n,
               per_launch_us([&] { chain_kernel<<<106, 256, 0, st>>>(d, n, 0.999999); }, reps));
    {   // does the size of the kernel-argument block matter?  1.9 KB by value against a pointer to the same block in device memory
        Big hb{};
        Big* db;
        hipMalloc(&db, sizeof(Big));
        hipMemcpy(db, &hb, sizeof(Big), hipMemcpyHostToDevice);
        for (int wg : {106, 1400})
            printf("empty kernel, %5d workgroups, 1920-byte argument block by value: %.2f us per launch; by pointer: %.2f us\n", wg,
                   per_launch_us([&] { empty_big_kernel<<<wg, 256, 0, st>>>(hb, d); }, reps), per_launch_us([&] { empty_ptr_kernel<<<wg, 256, 0, st>>>(db, d); }, reps));
    }
    // three launches per "pass" against one
    printf("3 x empty(106): %.2f us per triple\n", per_launch_us([&] { for (int k = 0; k < 3; ++k) empty_kernel<<<106, 256, 0, st>>>(d); }, reps));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    hipEventRecord(e0, st);
    for (int k = 0; k < reps; ++k) empty_kernel<<<106, 256, 0, st>>>(d);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("events around %d empty(106): %.2f us per launch\n", reps, ms * 1e3 / reps);
    return 0;
}
