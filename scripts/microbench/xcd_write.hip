// Do the eight XCDs of an MI355X write at the same rate?  W workgroups per XCD, every workgroup streams its own share of a 1 GiB
// buffer; the end time of every workgroup is recorded (100 MHz constant clock) next to the XCD it ran on (HW_REG_XCC_ID), and the mean /
// last end per XCD is printed.  Layouts: "wg"  = workgroup b owns chunk b (the XCDs' chunks interleave), "xcd" = every XCD owns a contiguous
// eighth, "rows" = the lat-lon kernel's pattern (six arrays of 5761-double rows, a workgroup writes 4 KB per row and array, every XCD a
// contiguous eighth of the rows).
// build + run:  hipcc --offload-arch=gfx950 -O3 -w -o /tmp/xw scripts/microbench/xcd_write.hip && /tmp/xw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double dbl2 __attribute__((ext_vector_type(2), aligned(8)));
struct Rec { unsigned long long t0, t1; unsigned xcc; unsigned pad; };

__device__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xf;
}

__global__ __launch_bounds__(256) void stream(double* out, long doubles_per_wg, int w_per_xcd, int layout, Rec* rec, unsigned xcd_mask = 0xff) {
    const unsigned long long t0 = wall_clock64();
    const int b = blockIdx.x;
    const long chunk = layout == 0 ? b : (long)(b % 8) * w_per_xcd + b / 8;
    dbl2* p = reinterpret_cast<dbl2*>(out + chunk * doubles_per_wg);
    dbl2 v; v.x = (double)b, v.y = 1.0;
    if (xcd_mask >> (b % 8) & 1)
        for (long k = threadIdx.x; k < doubles_per_wg / 2; k += 256) __builtin_nontemporal_store(v, p + k);
    __syncthreads();
    if (threadIdx.x == 0) rec[b] = Rec{t0, (unsigned long long)wall_clock64(), xcc_id(), 0u};
}

// the same bytes, handed out in 64 KB pieces from one counter: every XCD takes what it can
__global__ __launch_bounds__(256) void stream_ticket(double* out, long n_pieces, unsigned* ticket, Rec* rec) {
    const unsigned long long t0 = wall_clock64();
    __shared__ unsigned s_t;
    dbl2 v; v.x = (double)blockIdx.x, v.y = 1.0;
    for (;;) {
        if (threadIdx.x == 0) s_t = atomicAdd(ticket, 1u);
        __syncthreads();
        const unsigned t = s_t;
        __syncthreads();
        if (t >= n_pieces) break;
        dbl2* p = reinterpret_cast<dbl2*>(out + (long)t * 8192);
#pragma unroll
        for (int k = 0; k < 16; ++k) __builtin_nontemporal_store(v, p + k * 256 + threadIdx.x);
    }
    if (threadIdx.x == 0) rec[blockIdx.x] = Rec{t0, (unsigned long long)wall_clock64(), xcc_id(), 0u};
}

constexpr long NI1 = 5761, ROWS = 3435;
struct Six { double* f[6]; };
__global__ __launch_bounds__(256) void rows(Six s, int rows_per_strip, int w_per_xcd, Rec* rec, int tilt) {
    const unsigned long long t0 = wall_clock64();
    const int b = blockIdx.x;
    const int x = b % 8, k = b / 8;                    // XCD x: its k-th workgroup
    // XCD x owns the rows [r0, r1); tilt: odd XCDs get (100 - tilt) % of an even one's rows
    const double w_even = 100.0, w_odd = 100.0 - tilt, unit = ROWS / (4 * w_even + 4 * w_odd);
    double before = 0.0;
    for (int y = 0; y < x; ++y) before += (y & 1) ? w_odd : w_even;
    const long r0 = (long)(before * unit), r1 = (x == 7) ? ROWS : (long)((before + ((x & 1) ? w_odd : w_even)) * unit);
    const int gy = w_per_xcd / 12, tile = k % 12, ly = k / 12;
    const long i0 = (tile * 256L + threadIdx.x) * 2;
    if (ly < gy && i0 + 1 < NI1) {
        const long n_strips = (r1 - r0 + rows_per_strip - 1) / rows_per_strip;
        for (long st = ly; st < n_strips; st += gy) {
            const long j0 = r0 + st * rows_per_strip, j1 = (j0 + rows_per_strip < r1) ? j0 + rows_per_strip : r1;
            for (long j = j0; j < j1; ++j) {
                dbl2 v2; v2.x = (double)j, v2.y = (double)i0;
#pragma unroll
                for (int f = 0; f < 6; ++f) __builtin_nontemporal_store(v2, reinterpret_cast<dbl2*>(s.f[f] + j * NI1 + i0));
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) rec[b] = Rec{t0, (unsigned long long)wall_clock64(), xcc_id(), 0u};
}

// the lat-lon pattern with the row strips handed out from counters: scope 8 = every XCD its contiguous eighth of the rows and one counter
// per column tile in it (balances inside an XCD only), 4 = every XCD PAIR (2k, 2k+1) a contiguous quarter, 1 = one pool for the chip.
// Every XCD has workgroups on all twelve column tiles.
__global__ __launch_bounds__(256) void rows_ticket(Six s, int rows_per_strip, unsigned* ticket, Rec* rec, int scope) {
    const unsigned long long t0 = wall_clock64();
    __shared__ unsigned s_t;
    const int x = blockIdx.x % 8, tile = (blockIdx.x / 8) % 12;
    const int region = scope == 8 ? x : scope == 4 ? x / 2 : 0;
    const long r0 = ROWS * region / scope, r1 = ROWS * (region + 1) / scope;
    const long i0 = (tile * 256L + threadIdx.x) * 2;
    const long n_strips = (r1 - r0 + rows_per_strip - 1) / rows_per_strip;
    for (;;) {
        if (threadIdx.x == 0) s_t = atomicAdd(ticket + region * 12 + tile, 1u);
        __syncthreads();
        const long st = s_t;
        __syncthreads();
        if (st >= n_strips) break;
        if (i0 + 1 >= NI1) continue;
        const long j0 = r0 + st * rows_per_strip, j1 = (j0 + rows_per_strip < r1) ? j0 + rows_per_strip : r1;
        for (long j = j0; j < j1; ++j) {
            dbl2 v2; v2.x = (double)j, v2.y = (double)i0;
#pragma unroll
            for (int f = 0; f < 6; ++f) __builtin_nontemporal_store(v2, reinterpret_cast<dbl2*>(s.f[f] + j * NI1 + i0));
        }
    }
    if (threadIdx.x == 0) rec[blockIdx.x] = Rec{t0, (unsigned long long)wall_clock64(), xcc_id(), 0u};
}

static void report(const char* what, const std::vector<Rec>& r, double gb) {
    unsigned long long t0 = ~0ull, t1 = 0;
    for (auto& e : r) { if (e.t0 < t0) t0 = e.t0; if (e.t1 > t1) t1 = e.t1; }
    double sum[8] = {}, last[8] = {}; int n[8] = {}, mism = 0;
    for (size_t b = 0; b < r.size(); ++b) {
        const int x = r[b].xcc & 7;
        if (x != (int)(b % 8)) ++mism;
        const double e = (r[b].t1 - t0) / 100.0;
        sum[x] += e, ++n[x];
        if (e > last[x]) last[x] = e;
    }
    printf("%-38s %7.1f us %5.2f TB/s | mean end (last end) per XCD:", what, (t1 - t0) / 100.0, gb / ((t1 - t0) / 100.0) * 1e3);
    for (int x = 0; x < 8; ++x) printf(" %.0f(%.0f)", n[x] ? sum[x] / n[x] : 0.0, last[x]);
    printf("%s\n", mism ? "  [workgroup b NOT on XCD b % 8]" : "");
}

int main() {
    const long N = 1L << 27;   // 1 GiB of doubles
    double* buf; hipMalloc(&buf, N * 8 + (64 << 20));
    Rec* drec; hipMalloc(&drec, 4096 * sizeof(Rec));
    unsigned* ticket; hipMalloc(&ticket, 512);
    for (unsigned mask : {0x01u, 0x02u, 0x04u, 0x08u, 0x10u, 0x20u, 0x40u, 0x80u, 0x03u, 0x05u, 0x11u, 0x55u, 0xaau, 0x0fu, 0xf0u, 0xffu}) {
        const int w = 32, nwg = 8 * w;
        const long per = N / nwg;
        for (int k = 0; k < 3; ++k) stream<<<nwg, 256>>>(buf, per, w, 1, drec, mask);
        hipDeviceSynchronize();
        std::vector<Rec> r(nwg);
        hipMemcpy(r.data(), drec, nwg * sizeof(Rec), hipMemcpyDeviceToHost);
        char what[64]; snprintf(what, sizeof what, "XCD mask %02x, 32 wg/XCD, 128 MiB each", mask);
        report(what, r, __builtin_popcount(mask) * (N / 8) * 8 / 1e9);
    }
    for (int rep = 0; rep < 3; ++rep) {
        for (int w : {12, 20, 32, 64}) {
            const int nwg = 8 * w;
            for (int k = 0; k < 3; ++k) {
                hipMemset(ticket, 0, 64);
                stream_ticket<<<nwg, 256>>>(buf, N / 8192, ticket, drec);
            }
            hipDeviceSynchronize();
            std::vector<Rec> r(nwg);
            hipMemcpy(r.data(), drec, nwg * sizeof(Rec), hipMemcpyDeviceToHost);
            char what[64]; snprintf(what, sizeof what, "1 GiB, %d wg/XCD, 64 KB tickets", w);
            report(what, r, N * 8 / 1e9);
        }
        for (int w : {12, 20, 32, 64})
            for (int layout : {0, 1}) {
                const int nwg = 8 * w;
                const long per = N / nwg;
                for (int k = 0; k < 3; ++k) stream<<<nwg, 256>>>(buf, per, w, layout, drec);
                hipDeviceSynchronize();
                std::vector<Rec> r(nwg);
                hipMemcpy(r.data(), drec, nwg * sizeof(Rec), hipMemcpyDeviceToHost);
                char what[64]; snprintf(what, sizeof what, "1 GiB, %d wg/XCD, layout %s", w, layout ? "xcd" : "wg");
                report(what, r, N * 8 / 1e9);
            }
        const size_t pitch = ((size_t)(NI1 * ROWS + 16) * 8 + (4u << 20)) & ~((size_t)(2u << 20) - 1);
        char* slab; hipMalloc(&slab, 6 * pitch);
        Six s; for (int f = 0; f < 6; ++f) s.f[f] = reinterpret_cast<double*>(slab + f * pitch);
        for (int w : {12, 24})
            for (int tilt : {0, 14, 20}) {
                for (int k = 0; k < 3; ++k) rows<<<8 * w, 256>>>(s, 16, w, drec, tilt);
                hipDeviceSynchronize();
                std::vector<Rec> r(8 * w);
                hipMemcpy(r.data(), drec, r.size() * sizeof(Rec), hipMemcpyDeviceToHost);
                char what[64]; snprintf(what, sizeof what, "six arrays, %d wg/XCD, odd -%d %%", w, tilt);
                report(what, r, 6.0 * NI1 * ROWS * 8 / 1e9);
            }
        for (int w : {12, 24})
            for (int scope : {8, 4, 1}) {
                for (int k = 0; k < 3; ++k) {
                    hipMemset(ticket, 0, 512);
                    rows_ticket<<<8 * w, 256>>>(s, 16, ticket, drec, scope);
                }
                hipDeviceSynchronize();
                std::vector<Rec> r(8 * w);
                hipMemcpy(r.data(), drec, r.size() * sizeof(Rec), hipMemcpyDeviceToHost);
                char what[64]; snprintf(what, sizeof what, "six arrays, %d wg/XCD, tickets / %d regions", w, scope);
                report(what, r, 6.0 * NI1 * ROWS * 8 / 1e9);
            }
        hipFree(slab);
    }
    return 0;
}
