// Cold write patterns at 1/16 degree size: which way of handing the lat-lon rows to resident workgroups gets most out of HBM when nothing
// of the previous pass is left on the chip (3.8 GB per sweep)?  All six arrays, 16-byte stores, values from registers.
//   a   one array, contiguous (what a fill does): the ceiling
//   p   the pass's pattern: G resident workgroups, a column tile of 512 * W columns each, R-row strips handed out from one counter per tile
//       (ticket asked at the head of a strip, waited for behind its stores), per row the six fields one after the other; nt / plain stores
//   q   the same, field-major inside a strip (six single-stream walks over R rows)
//   r   units = (strip of R whole rows), one counter: a workgroup writes whole rows (92 KB contiguous per field), field after field per row
//   s   units = (field, strip of R whole rows) in field-major order, one counter: G workgroups always write G * R consecutive rows of ONE field
// build here, run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o scripts/microbench/strip_patterns scripts/microbench/strip_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double dbl2 __attribute__((ext_vector_type(2), aligned(8)));
#ifndef NI1V
#define NI1V 11521
#define ROWSV 6870
#endif
constexpr long NI1 = NI1V, ROWS = ROWSV, N = NI1 * ROWS;
struct Six { double* f[6]; };

__device__ unsigned ticket_ask(unsigned* c) {
    unsigned r; const unsigned one = 1u;
    asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(r) : "v"(c), "v"(one) : "memory");
    return r;
}
template <int LATER> __device__ unsigned ticket_answer(unsigned p) {
    unsigned o;
    asm volatile("s_waitcnt vmcnt(%2)\n\tv_mov_b32 %0, %1" : "=v"(o) : "v"(p), "n"(LATER) : "memory");
    return o;
}
template <bool NT> __device__ void st2(double* q, dbl2 v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<dbl2*>(q)); else *reinterpret_cast<dbl2*>(q) = v;
}

// p / q: W = pairs per thread (1: 512 columns per workgroup, 2: 1024)
template <bool NT, int W, bool FIELD_MAJOR>
__global__ __launch_bounds__(256) void k_pool(Six s, unsigned* cnt, int R) {
    __shared__ int s_claim;
    const long cols = 512 * W, tiles = (NI1 + cols - 1) / cols;
    const long tile = blockIdx.x % tiles;
    const long n_strips = (ROWS + R - 1) / R;
    unsigned* c = cnt + tile;
    if (threadIdx.x == 0) { const unsigned t = atomicAdd(c, 1u); s_claim = t < n_strips ? (int)t : -1; }
    __syncthreads();
    for (;;) {
        const int pick = s_claim;
        if (pick < 0) break;
        unsigned asked;
        if (threadIdx.x == 0) asked = ticket_ask(c);
        __syncthreads();
        const long j0 = (long)pick * R, j1 = (j0 + R < ROWS) ? j0 + R : ROWS;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const long i0 = tile * cols + w * 512 + threadIdx.x * 2;
            if (i0 + 1 < NI1) {
                if (FIELD_MAJOR) {
#pragma unroll 1
                    for (int f = 0; f < 6; ++f)
                        for (long j = j0; j < j1; ++j) { dbl2 v; v.x = (double)j, v.y = (double)i0; st2<NT>(s.f[f] + j * NI1 + i0, v); }
                } else {
                    for (long j = j0; j < j1; ++j) {
                        dbl2 v; v.x = (double)j, v.y = (double)i0;
#pragma unroll
                        for (int f = 0; f < 6; ++f) st2<NT>(s.f[f] + j * NI1 + i0, v);
                    }
                }
            }
        }
        if (threadIdx.x == 0) { const unsigned t = (j1 - j0) * 6 * W >= 16 ? ticket_answer<16>(asked) : ticket_answer<0>(asked); s_claim = t < n_strips ? (int)t : -1; }
        __syncthreads();
    }
}

// r / s: whole rows.  FIELD_UNITS: units are (field, strip) in field-major order; else (strip) with all six fields
template <bool NT, bool FIELD_UNITS>
__global__ __launch_bounds__(256) void k_rows(Six s, unsigned* cnt, int R) {
    __shared__ int s_claim;
    const long n_strips = (ROWS + R - 1) / R, units = FIELD_UNITS ? 6 * n_strips : n_strips;
    if (threadIdx.x == 0) { const unsigned t = atomicAdd(cnt, 1u); s_claim = t < units ? (int)t : -1; }
    __syncthreads();
    for (;;) {
        const int pick = s_claim;
        if (pick < 0) break;
        unsigned asked;
        if (threadIdx.x == 0) asked = ticket_ask(cnt);
        __syncthreads();
        const int f0 = FIELD_UNITS ? (int)(pick / n_strips) : 0, f1 = FIELD_UNITS ? f0 + 1 : 6;
        const long st = FIELD_UNITS ? pick % n_strips : pick;
        const long j0 = st * R, j1 = (j0 + R < ROWS) ? j0 + R : ROWS;
        for (long j = j0; j < j1; ++j)
            for (int f = f0; f < f1; ++f) {
                double* q = s.f[f] + j * NI1;
                for (long i0 = threadIdx.x * 2; i0 + 1 < NI1; i0 += 512) { dbl2 v; v.x = (double)j, v.y = (double)i0; st2<NT>(q + i0, v); }
            }
        if (threadIdx.x == 0) { const unsigned t = ticket_answer<16>(asked); s_claim = t < units ? (int)t : -1; }
        __syncthreads();
    }
}

// t   the s pattern under the constraints of the real kernel: a workgroup is bound to a SEGMENT of up to 6144 columns (12 column pairs per
//     thread: their column quantity sits in registers and is reloaded from a table when the field changes), takes units (band-less here:
//     field, R rows) of its segment from the segment's counter, and fetches the unit's row scalars with a SCALAR load from a row table
//     (constant address space) before its first store; value = column quantity x row scalar
struct RowS { double lat, sl, cl, dy; };
typedef const __attribute__((address_space(4))) RowS* ConstRowS;
template <bool NT>
__global__ __launch_bounds__(256) void k_seg(Six s, unsigned* cnt, int R, const double* __restrict__ col_tab, const RowS* row_tab, int n_seg) {
    __shared__ int s_claim;
    const long seg = blockIdx.x % n_seg;
    const long seg_cols = ((NI1 + n_seg - 1) / n_seg + 511) / 512 * 512;   // multiple of 512
    const long c_lo = seg * seg_cols;
    const long n_strips = (ROWS + R - 1) / R, units = 6 * n_strips;
    unsigned* c = cnt + seg;
    if (threadIdx.x == 0) { const unsigned t = atomicAdd(c, 1u); s_claim = t < units ? (int)t : -1; }
    __syncthreads();
    double cq[24];
    int cur_f = -1;
    for (;;) {
        const int pick = __builtin_amdgcn_readfirstlane(s_claim);
        if (pick < 0) break;
        unsigned asked;
        if (threadIdx.x == 0) asked = ticket_ask(c);
        __syncthreads();
        const int f = (int)(pick / n_strips);
        const long st = pick % n_strips;
        if (f != cur_f) {   // (a drain of this wave's stores: 6 times per sweep)
            cur_f = f;
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const long i0 = c_lo + k * 512 + threadIdx.x * 2;
                const bool in = i0 + 1 < NI1 && k * 512 < seg_cols;
                cq[2 * k] = in ? col_tab[(long)(f % 4) * NI1 + i0] : 0.0, cq[2 * k + 1] = in ? col_tab[(long)(f % 4) * NI1 + i0 + 1] : 0.0;
            }
        }
        const long j0 = st * R, j1 = (j0 + R < ROWS) ? j0 + R : ROWS;
        for (long j = j0; j < j1; ++j) {
            ConstRowS rs = (ConstRowS)(unsigned long long)(row_tab + j);
            const double m = rs->cl + rs->dy;    // one 32-byte scalar load per row
            double* q = s.f[f] + j * NI1 + c_lo + threadIdx.x * 2;
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const long i0 = c_lo + k * 512 + threadIdx.x * 2;
                if (i0 + 1 < NI1 && k * 512 < seg_cols) { dbl2 v; v.x = cq[2 * k] * m, v.y = cq[2 * k + 1] * m; st2<NT>(q + k * 512, v); }
            }
        }
        if (threadIdx.x == 0) { const unsigned t = ticket_answer<8>(asked); s_claim = t < units ? (int)t : -1; }
        __syncthreads();
    }
}
__global__ void k_a(double* a, long n2) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n2) { dbl2 v; v.x = 1.0, v.y = 2.0; reinterpret_cast<dbl2*>(a)[i] = v; }
}
template <class F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int k = 0; k < 2; ++k) f();
    hipEventRecord(e0);
    for (int k = 0; k < 6; ++k) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 6;
}
int main() {
    Six s; for (int f = 0; f < 6; ++f) hipMalloc(&s.f[f], (N + 16) * 8);
    double* big; hipMalloc(&big, 6 * (N + 16) * 8);
    unsigned* cnt; hipMalloc(&cnt, 4096);
    const double gb6 = 6.0 * N * 8 / 1e9;
    printf("NI1 %ld ROWS %ld: %.2f GB per sweep\n", NI1, ROWS, gb6);
    { const long n2 = 3 * N; float ms = timeit([&] { k_a<<<(unsigned)((n2 + 255) / 256), 256>>>(big, n2); }); printf("a  one array contiguous                                  %7.3f ms %7.1f GB/s\n", ms, gb6 / ms * 1e3); }
    auto zero = [&] { hipMemsetAsync(cnt, 0, 4096, 0); };
    const long t1 = (NI1 + 511) / 512, t2 = (NI1 + 1023) / 1024;
    for (int R : {32, 12, 6, 4}) for (int g : {4, 6, 8}) {
        float ms = timeit([&] { zero(); k_pool<true, 1, false><<<(unsigned)(t1 * g), 256>>>(s, cnt, R); });
        printf("p  nt, 512 cols, R %2d, %4ld WGs                          %7.3f ms %7.1f GB/s\n", R, t1 * g, ms, gb6 / ms * 1e3);
    }
    for (int R : {12, 6}) for (int g : {4, 8}) {
        float ms = timeit([&] { zero(); k_pool<false, 1, false><<<(unsigned)(t1 * g), 256>>>(s, cnt, R); });
        printf("p  plain, 512 cols, R %2d, %4ld WGs                       %7.3f ms %7.1f GB/s\n", R, t1 * g, ms, gb6 / ms * 1e3);
    }
    for (int R : {12, 6, 3}) for (int g : {4, 8, 12}) {
        float ms = timeit([&] { zero(); k_pool<true, 2, false><<<(unsigned)(t2 * g), 256>>>(s, cnt, R); });
        printf("p  nt, 1024 cols, R %2d, %4ld WGs                         %7.3f ms %7.1f GB/s\n", R, t2 * g, ms, gb6 / ms * 1e3);
    }
    for (int R : {12, 6}) for (int g : {4, 8}) {
        float ms = timeit([&] { zero(); k_pool<true, 1, true><<<(unsigned)(t1 * g), 256>>>(s, cnt, R); });
        printf("q  nt, 512 cols, field-major, R %2d, %4ld WGs             %7.3f ms %7.1f GB/s\n", R, t1 * g, ms, gb6 / ms * 1e3);
    }
    for (int R : {1, 2, 4}) for (int g : {64, 96, 128, 256}) {
        float ms = timeit([&] { zero(); k_rows<true, false><<<g, 256>>>(s, cnt, R); });
        printf("r  nt, whole rows x 6 fields, R %2d, %4d WGs              %7.3f ms %7.1f GB/s\n", R, g, ms, gb6 / ms * 1e3);
    }
    for (int nt : {1, 0}) for (int R : {1, 2, 4}) for (int g : {64, 96, 128, 256, 512}) {
        float ms = nt ? timeit([&] { zero(); k_rows<true, true><<<g, 256>>>(s, cnt, R); }) : timeit([&] { zero(); k_rows<false, true><<<g, 256>>>(s, cnt, R); });
        printf("s  %s, (field, whole rows) units, R %2d, %4d WGs        %7.3f ms %7.1f GB/s\n", nt ? "nt   " : "plain", R, g, ms, gb6 / ms * 1e3);
    }
    {
        double* col_tab; hipMalloc(&col_tab, 4 * (NI1 + 16) * 8); hipMemset(col_tab, 0, 4 * (NI1 + 16) * 8);
        RowS* row_tab; hipMalloc(&row_tab, (ROWS + 16) * sizeof(RowS)); hipMemset(row_tab, 0, (ROWS + 16) * sizeof(RowS));
        const int n_seg = (int)((NI1 + 6143) / 6144);
        for (int nt : {1, 0}) for (int R : {1, 2, 4}) for (int g : {64, 96, 128, 160, 256}) {
            const int grid = g / n_seg * n_seg;
            float ms = nt ? timeit([&] { zero(); k_seg<true><<<grid, 256>>>(s, cnt, R, col_tab, row_tab, n_seg); })
                          : timeit([&] { zero(); k_seg<false><<<grid, 256>>>(s, cnt, R, col_tab, row_tab, n_seg); });
            printf("t  %s, %d segment(s), column regs + scalar row loads, R %2d, %4d WGs  %7.3f ms %7.1f GB/s\n", nt ? "nt   " : "plain", n_seg, R, grid, ms, gb6 / ms * 1e3);
        }
    }
    return 0;
}
