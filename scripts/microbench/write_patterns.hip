// Write-bandwidth micro-benchmark: how does the HBM write rate of gfx950 depend on the store pattern?
//   a  one array, contiguous, 16-byte stores, short-lived workgroups (what a fill kernel does)
//   b  six arrays, a workgroup writes 4 KB per row and array, row after row (the lat-lon kernel's pattern), G resident workgroups
//   c  six arrays, every workgroup writes ONE contiguous 64 KB chunk of ONE array (field-specialised)
//   d  six arrays, a persistent workgroup writes whole rows (46 KB contiguous per array), G resident workgroups
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/wp scripts/microbench/write_patterns.hip && /tmp/wp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double dbl2 __attribute__((ext_vector_type(2), aligned(8)));
constexpr long NI1 = 5761, ROWS = 3435;   // 19.8 M points like the 1/8 degree lat-lon sub-grids
constexpr long N = NI1 * ROWS;

__global__ void k_a(double* a, long n2) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n2) { dbl2 v; v.x = 1.0, v.y = 2.0; reinterpret_cast<dbl2*>(a)[i] = v; }
}
struct Six { double* f[6]; };
__global__ __launch_bounds__(256) void k_b(Six s, int rows_per_strip) {
    const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i0 + 1 >= NI1) return;
    const long n_strips = (ROWS + rows_per_strip - 1) / rows_per_strip;
    for (long st = blockIdx.y; st < n_strips; st += gridDim.y) {
        const long j0 = st * rows_per_strip, j1 = (j0 + rows_per_strip < ROWS) ? j0 + rows_per_strip : ROWS;
        for (long j = j0; j < j1; ++j) {
            dbl2 v; v.x = (double)j, v.y = (double)i0;
#pragma unroll
            for (int f = 0; f < 6; ++f) *reinterpret_cast<dbl2*>(s.f[f] + j * NI1 + i0) = v;
        }
    }
}
__global__ __launch_bounds__(256) void k_c(Six s, long chunk2) {  // chunk2: dbl2 elements per workgroup
    const long n2 = N / 2, chunks = (n2 + chunk2 - 1) / chunk2;
    const long c = blockIdx.x % chunks; const int f = (int)(blockIdx.x / chunks);
    if (f >= 6) return;
    dbl2* q = reinterpret_cast<dbl2*>(s.f[f]);
    const long e = (c + 1) * chunk2 < n2 ? (c + 1) * chunk2 : n2;
    for (long i = c * chunk2 + threadIdx.x; i < e; i += 256) { dbl2 v; v.x = (double)i, v.y = 1.0; q[i] = v; }
}
__global__ __launch_bounds__(256) void k_d(Six s) {  // whole rows per workgroup
    for (long j = blockIdx.x; j < ROWS; j += gridDim.x) {
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            double* q = s.f[f] + j * NI1;
            for (long i0 = threadIdx.x * 2; i0 + 1 < NI1; i0 += 512) { dbl2 v; v.x = (double)j, v.y = (double)i0; *reinterpret_cast<dbl2*>(q + i0) = v; }
        }
    }
}
// e/f/g: short-lived workgroups, one (strip of R rows) x (512-column tile) each
//   e  one field per workgroup, workgroup order: tile fastest, then strip, then field  (a compact window sweeps field after field)
//   f  one field per workgroup, order: field fastest, then tile, then strip
//   g  all six fields per workgroup, order: tile fastest, then strip
__global__ __launch_bounds__(256) void k_efg(Six s, int R, int mode) {
    const long tiles = 12, strips = (ROWS + R - 1) / R;
    long b = blockIdx.x, t, st; int f0, f1;
    if (mode == 0) { t = b % tiles; st = (b / tiles) % strips; f0 = (int)(b / (tiles * strips)); f1 = f0 + 1; }
    else if (mode == 1) { f0 = (int)(b % 6); f1 = f0 + 1; t = (b / 6) % tiles; st = b / (6 * tiles); }
    else { t = b % tiles; st = b / tiles; f0 = 0; f1 = 6; }
    const long i0 = (t * 256 + threadIdx.x) * 2;
    if (i0 + 1 >= NI1 || st >= strips) return;
    const long j0 = st * R, j1 = (j0 + R < ROWS) ? j0 + R : ROWS;
    for (long j = j0; j < j1; ++j) {
        dbl2 v; v.x = (double)j, v.y = (double)i0;
        for (int f = f0; f < f1; ++f) *reinterpret_cast<dbl2*>(s.f[f] + j * NI1 + i0) = v;
    }
}
// h  units = (field, row), taken in order by G persistent workgroups (unit = wg + k G): the G workgroups always write G
//    consecutive rows of ONE field -- a compact window that sweeps through memory; every pair is loaded from a column table
//    (L2-resident) and stored with an aligned 16-byte store (odd rows peel their first element)
__global__ __launch_bounds__(256) void k_h(Six s, const double* __restrict__ tab, int use_tab) {
    const long units = 6 * ROWS;
    for (long u = blockIdx.x; u < units; u += gridDim.x) {
        const int f = (int)(u / ROWS); const long j = u % ROWS;
        double* q = s.f[f] + j * NI1;
        const int sft = (int)((reinterpret_cast<unsigned long>(q) >> 3) & 1ul);
        if (sft && threadIdx.x == 0) q[0] = use_tab ? tab[0] : 1.0;
        for (long i = sft + 2 * threadIdx.x; i < NI1; i += 512) {
            if (i + 1 < NI1) {
                dbl2 v;
                if (use_tab) { v.x = tab[i] * (double)j; v.y = tab[i + 1] * (double)j; } else { v.x = (double)j; v.y = (double)i; }
                typedef double dbl2a __attribute__((ext_vector_type(2), aligned(16)));
                dbl2a w; w.x = v.x, w.y = v.y;
                *reinterpret_cast<dbl2a*>(q + i) = w;
            } else {
                q[i] = use_tab ? tab[i] : 2.0;
            }
        }
    }
}
// b2: pattern b with an XCD-aware remap: workgroup b runs on XCD b % 8; give every XCD a contiguous eighth of the row strips
__global__ __launch_bounds__(256) void k_b2(Six s, int rows_per_strip, int wg_per_xcd, int remap) {
    const long n_strips = (ROWS + rows_per_strip - 1) / rows_per_strip;
    const int b = blockIdx.x;                       // 12 column tiles x gy strips-in-flight
    int v = b;
    if (remap == 1 || remap == 3) v = (b % 8) * wg_per_xcd + b / 8;   // virtual id: XCD x owns ids [x * wg_per_xcd, (x+1) * wg_per_xcd)
    // remap == 2: contiguous blocks of strips per workgroup, but no per-XCD grouping (v = b)
    const int total = 8 * wg_per_xcd;
    const long tile = v % 12, lane_strip = v / 12, gy = total / 12;
    const long i0 = (tile * 256 + threadIdx.x) * 2;
    if (i0 + 1 >= NI1) return;
    // strips of this virtual workgroup: contiguous block of the strip range when remapped, grid-stride otherwise
    const long per = (n_strips + gy - 1) / gy;
    // remap == 3: XCD x owns a contiguous eighth of the strips; its row-lanes take them grid-stride (one compact window per XCD)
    const long lanes_x = (gy + 7) / 8, x = lane_strip / lanes_x, l = lane_strip % lanes_x, per_x = (n_strips + 7) / 8;
    for (long k = 0; k < per; ++k) {
        long st = remap ? lane_strip * per + k : lane_strip + k * gy;
        if (remap == 3) { st = x * per_x + l + k * lanes_x; if (l + k * lanes_x >= per_x) break; }
        if (st >= n_strips) break;
        const long j0 = st * rows_per_strip, j1 = (j0 + rows_per_strip < ROWS) ? j0 + rows_per_strip : ROWS;
        for (long j = j0; j < j1; ++j) {
            dbl2 v2; v2.x = (double)j, v2.y = (double)i0;
#pragma unroll
            for (int f = 0; f < 6; ++f) *reinterpret_cast<dbl2*>(s.f[f] + j * NI1 + i0) = v2;
        }
    }
}
template <class F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int k = 0; k < 3; ++k) f();
    hipEventRecord(e0);
    for (int k = 0; k < 10; ++k) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 10;
}
int main() {
    Six s; for (int f = 0; f < 6; ++f) hipMalloc(&s.f[f], (N + 16) * 8);
    double* big; hipMalloc(&big, 6 * (N + 16) * 8);
    const double gb6 = 6.0 * N * 8 / 1e9;
    { const long n2 = 3 * N; float ms = timeit([&] { k_a<<<(unsigned)((n2 + 255) / 256), 256>>>(big, n2); }); printf("a  one array contiguous             %7.3f ms %7.1f GB/s\n", ms, gb6 / ms * 1e3); }
    for (int g : {2, 3, 5, 8, 12, 24, 100}) { float ms = timeit([&] { k_b<<<dim3(12, g), 256>>>(s, 16); }); printf("b  6 arrays, 4 KB x rows, %4d WGs    %7.3f ms %7.1f GB/s\n", 12 * g, ms, gb6 / ms * 1e3); }
    for (int wpx : {12, 24, 48, 96}) for (int rm : {0, 1, 2, 3}) { float ms = timeit([&] { k_b2<<<8 * wpx, 256>>>(s, 16, wpx, rm); }); printf("b2 6 arrays, 4 KB x rows, %4d WGs, XCD remap %d %7.3f ms %7.1f GB/s\n", 8 * wpx, rm, ms, gb6 / ms * 1e3); }
    for (long ch : {1024L, 4096L, 16384L}) { const long chunks = (N / 2 + ch - 1) / ch; float ms = timeit([&] { k_c<<<(unsigned)(chunks * 6), 256>>>(s, ch); }); printf("c  6 arrays, field WGs %6ld KB chunk %7.3f ms %7.1f GB/s\n", ch * 16 / 1024, ms, gb6 / ms * 1e3); }
    for (int g : {32, 64, 128, 256, 512, 1024}) { float ms = timeit([&] { k_d<<<g, 256>>>(s); }); printf("d  6 arrays, whole rows, %4d WGs      %7.3f ms %7.1f GB/s\n", g, ms, gb6 / ms * 1e3); }
    double* tab; hipMalloc(&tab, (NI1 + 16) * 8); hipMemset(tab, 0, (NI1 + 16) * 8);
    for (int ut : {0, 1}) for (int g : {48, 60, 84, 120, 240, 512, 2048, 20610}) { float ms = timeit([&] { k_h<<<g, 256>>>(s, tab, ut); }); printf("h  (field,row) units, tab=%d, %5d WGs     %7.3f ms %7.1f GB/s\n", ut, g, ms, gb6 / ms * 1e3); }
    for (int R : {4, 16, 64}) {
        const long strips = (ROWS + R - 1) / R;
        float ms = timeit([&] { k_efg<<<(unsigned)(12 * strips * 6), 256>>>(s, R, 0); }); printf("e  field WGs, %2d rows x 4 KB, field-major   %7.3f ms %7.1f GB/s\n", R, ms, gb6 / ms * 1e3);
        ms = timeit([&] { k_efg<<<(unsigned)(12 * strips * 6), 256>>>(s, R, 1); }); printf("f  field WGs, %2d rows x 4 KB, field-fastest %7.3f ms %7.1f GB/s\n", R, ms, gb6 / ms * 1e3);
        ms = timeit([&] { k_efg<<<(unsigned)(12 * strips), 256>>>(s, R, 2); }); printf("g  6-field WGs, %2d rows x 4 KB, short-lived %7.3f ms %7.1f GB/s\n", R, ms, gb6 / ms * 1e3);
    }
    return 0;
}
