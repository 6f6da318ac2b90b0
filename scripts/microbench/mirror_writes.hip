// How fast can the cap roles' store pattern go?  Three fields of ROWS x NX doubles (the 1/16 degree bipolar cap's dx, dy, area), a wave owns
// W cells of a row and walks RPC rows; no arithmetic.  Variants (mode):
//   0  every column, W = 63, 8-byte stores (what the quadrature strips do)            1  the same, mirrored (own + three images)
//   2  every column, W = 64 (512-byte aligned segments)                               3  W = 64 mirrored
//   4  every column, W = 63, 16-byte stores where a pair is 16-byte aligned            5  W = 64, 16-byte stores (32 lanes store)
//   6  W = 64, 16-byte stores, mirrored                                                7  W = 64, 16-byte non-temporal stores
//   8  every column, W = 63, non-temporal 8-byte stores
// build + run:  hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mw scripts/microbench/mirror_writes.hip && /tmp/mw
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr long NX = 11520, ROWS = 1780;
struct F3 { double* f[3]; };
typedef double dbl2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void st16(double* p, double a, double b, bool nt) {
    dbl2 v; v.x = a, v.y = b;
    if (nt) __builtin_nontemporal_store(v, reinterpret_cast<dbl2*>(p)); else *reinterpret_cast<dbl2*>(p) = v;
}

__global__ __launch_bounds__(256) void k(F3 s, int mode, int rpc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long strip = (long)blockIdx.x * 4 + wave;
    const long q = NX / 4, h2 = NX / 2;
    const bool mirrored = (mode == 1 || mode == 3 || mode == 6);
    const int W = (mode == 0 || mode == 1 || mode == 4 || mode == 8) ? 63 : 64;
    const bool wide = (mode >= 4 && mode <= 7);
    const bool nt = (mode == 7 || mode == 8);
    const long span = mirrored ? q : NX;
    const long ci = strip * W + lane;
    if (strip * W >= span) return;
    const bool on = lane < W && ci < span;
    const long r0 = (long)blockIdx.y * rpc;
    for (long r = r0; r < r0 + rpc && r < ROWS; ++r) {
        const double v = (double)(r * NX + ci);
        const double vn = __shfl_down(v, 1);
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            double* row = s.f[f] + r * NX;
            if (!wide) {
                if (on) {
                    if (nt) __builtin_nontemporal_store(v, row + ci); else row[ci] = v;
                    if (mirrored) row[h2 - 1 - ci] = v, row[h2 + ci] = v, row[NX - 1 - ci] = v;
                }
            } else {
                // pairs (ci even, ci + 1): the even cell's lane stores both; an odd first cell / even last cell of the wave stores alone
                const bool lead = on && (ci & 1) == 0 && lane + 1 < W && ci + 1 < span;
                const bool single = on && !lead && !((ci & 1) == 1 && lane > 0);
                if (lead) st16(row + ci, v, vn, nt);
                if (single) row[ci] = v;
                if (mirrored) {
                    if (lead) {
                        st16(row + h2 + ci, v, vn, nt);
                        st16(row + h2 - 2 - ci, vn, v, nt);          // cells h2-1-ci (v) and h2-2-ci (vn): ascending address order
                        st16(row + NX - 2 - ci, vn, v, nt);
                    }
                    if (single) row[h2 - 1 - ci] = v, row[h2 + ci] = v, row[NX - 1 - ci] = v;
                }
            }
        }
    }
}

int main() {
    F3 s;
    for (int f = 0; f < 3; ++f) hipMalloc(&s.f[f], (ROWS * NX + 64) * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const double gb = 3.0 * ROWS * NX * 8 / 1e9;
    const char* names[] = {"every column, 63, 8 B", "mirrored, 63, 8 B", "every column, 64, 8 B", "mirrored, 64, 8 B", "every column, 63, 16 B pairs", "every column, 64, 16 B",
                           "mirrored, 64, 16 B", "every column, 64, 16 B nt", "every column, 63, 8 B nt"};
    for (int rpc : {8, 2, 24})
        for (int mode = 0; mode < 9; ++mode) {
            const bool mirrored = (mode == 1 || mode == 3 || mode == 6);
            const int W = (mode == 0 || mode == 1 || mode == 4 || mode == 8) ? 63 : 64;
            const long span = mirrored ? NX / 4 : NX;
            const dim3 grid((unsigned)(((span + W - 1) / W + 3) / 4), (unsigned)((ROWS + rpc - 1) / rpc));
            for (int k0 = 0; k0 < 3; ++k0) k<<<grid, 256>>>(s, mode, rpc);
            hipEventRecord(e0);
            for (int k0 = 0; k0 < 20; ++k0) k<<<grid, 256>>>(s, mode, rpc);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("rows/wave %2d mode %d (%-30s): %.4f ms  %.0f GB/s\n", rpc, mode, names[mode], ms / 20, gb / (ms / 20) * 1e3);
        }
    return 0;
}
