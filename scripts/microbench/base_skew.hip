// Does the relative placement of the six output arrays matter to the lat-lon kernel's store pattern?  Pattern b2/remap 3 of write_patterns.hip
// (six arrays, a workgroup writes 4 KB per row and array, every XCD a contiguous eighth of the row strips, non-temporal 16-byte stores), with
// array f starting at base + f * skew bytes beyond a common 2 MiB-aligned slab layout.
// build + run:  hipcc --offload-arch=gfx950 -O3 -w -o /tmp/bs scripts/microbench/base_skew.hip && /tmp/bs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dbl2 __attribute__((ext_vector_type(2), aligned(8)));
constexpr long NI1 = 5761, ROWS = 3435;
constexpr long N = NI1 * ROWS;
struct Six { double* f[6]; };
__global__ __launch_bounds__(256) void k(Six s, int rows_per_strip, int wg_per_xcd, int nt) {
    const long n_strips = (ROWS + rows_per_strip - 1) / rows_per_strip;
    const int b = blockIdx.x;
    const int v = (b % 8) * wg_per_xcd + b / 8;
    const int total = 8 * wg_per_xcd;
    const long tile = v % 12, lane_strip = v / 12, gy = total / 12;
    const long i0 = (tile * 256 + threadIdx.x) * 2;
    if (i0 + 1 >= NI1) return;
    const long per = (n_strips + gy - 1) / gy;
    const long lanes_x = (gy + 7) / 8, x = lane_strip / lanes_x, l = lane_strip % lanes_x, per_x = (n_strips + 7) / 8;
    for (long k2 = 0; k2 < per; ++k2) {
        const long st = x * per_x + l + k2 * lanes_x;
        if (l + k2 * lanes_x >= per_x || st >= n_strips) break;
        const long j0 = st * rows_per_strip, j1 = (j0 + rows_per_strip < ROWS) ? j0 + rows_per_strip : ROWS;
        for (long j = j0; j < j1; ++j) {
            dbl2 v2; v2.x = (double)j, v2.y = (double)i0;
#pragma unroll
            for (int f = 0; f < 6; ++f) {
                dbl2* p = reinterpret_cast<dbl2*>(s.f[f] + j * NI1 + i0);
                if (nt) __builtin_nontemporal_store(v2, p); else *p = v2;
            }
        }
    }
}
int main() {
    char* slab;
    const size_t per = ((size_t)(N + 16) * 8 + (4u << 20)) & ~((size_t)(2u << 20) - 1);   // 2 MiB-aligned pitch between the arrays
    hipMalloc(&slab, 6 * per + (64u << 20));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double gb6 = 6.0 * N * 8 / 1e9;
    const long skews[] = {0, 128, 256, 512, 1024, 2048, 4096, 4096 + 256, 8192, 16384, 65536, 65536 + 4096, 262144, 1048576 + 4096, 40, 5761 * 8};
    for (int rep = 0; rep < 2; ++rep)
        for (int nt : {1, 0})
            for (long skew : skews) {
                Six s;
                for (int f = 0; f < 6; ++f) s.f[f] = reinterpret_cast<double*>(slab + f * per + f * skew);
                for (int w = 0; w < 3; ++w) k<<<8 * 24, 256>>>(s, 16, 24, nt);
                hipEventRecord(e0);
                for (int w = 0; w < 10; ++w) k<<<8 * 24, 256>>>(s, 16, 24, nt);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("nt %d skew %8ld B per array: %.4f ms  %.0f GB/s\n", nt, skew, ms / 10, gb6 / (ms / 10) * 1e3);
            }
    return 0;
}
