// Can the latency of the small launch A (tables) be hidden behind the big launch B of the PREVIOUS pass?  A(k) -> B(k) is a true
// dependence, B(k) -> A(k+1) is not once the tables are double-buffered.  B spins for a given time on the wall clock in every workgroup
// (and writes nothing), A is 106 workgroups of a 500-fma chain; wall time per pass over 400 passes for
//   one stream A, B, A, B ... (what a pass does)            |  B alone (the floor)
//   A on a side stream, events both ways                    |  the cost of an event record / of a wait on a completed event on the main stream
//   A on a side stream, unordered (is the overlap itself free?)
//   A of the next pass behind B on the side stream, the host asking hipEventQuery before the next B (how often is it late?)
//   ONE stream, A launched with hipExtAnyOrderLaunch behind B (no barrier bit)
//   stream memory operations: write-value behind A on the side stream, wait-value in front of B on the main stream
// Results of round 3: profiles/r03_microbench_stream_overlap.txt; what they meant for the pass: DESIGN.md section 5.
//   hipcc --offload-arch=gfx950 -O3 -o stream_overlap stream_overlap.hip && ./stream_overlap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>

__global__ void a_kernel(double* tab, int n, double a) {
    double x = threadIdx.x * 1e-9 + 1.0;
    for (int k = 0; k < n; ++k) x = fma(x, a, 1e-9);
    tab[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

__global__ void b_kernel(const double* tab, double* out, long long ticks) {   // wall_clock64: 100 MHz
    const long long t0 = wall_clock64();
    double v = tab[threadIdx.x];
    while (wall_clock64() - t0 < ticks) v = fma(v, 0.999, 1e-9);
    if (v == 123.456) out[0] = v;
}

template <class F>
double per_pass_us(F pass, int reps) {
    for (int k = 0; k < 50; ++k) pass(k);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < reps; ++k) pass(50 + k);
    hipDeviceSynchronize();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
}

int main() {
    double *tab[2], *out;
    hipMalloc(&tab[0], 106 * 256 * 8);
    hipMalloc(&tab[1], 106 * 256 * 8);
    hipMalloc(&out, 4096);
    hipStream_t st, side;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
    hipEvent_t ea[2], eb[2];
    for (int k = 0; k < 2; ++k) {
        hipEventCreateWithFlags(&ea[k], hipEventDisableTiming);
        hipEventCreateWithFlags(&eb[k], hipEventDisableTiming);
        hipEventRecord(eb[k], st);
    }
    const int reps = 400;
    for (int wg : {700, 5600})
        for (long long us : {30, 200}) {
            const long long ticks = us * 100;
            const double one = per_pass_us(
                [&](int k) {
                    a_kernel<<<106, 256, 0, st>>>(tab[0], 500, 0.999999);
                    b_kernel<<<wg, 256, 0, st>>>(tab[0], out, ticks);
                },
                reps);
            const double two = per_pass_us(
                [&](int k) {
                    const int s = k & 1;
                    hipStreamWaitEvent(side, eb[s], 0);   // the last reader of this table buffer: B(k-2)
                    a_kernel<<<106, 256, 0, side>>>(tab[s], 500, 0.999999);
                    hipEventRecord(ea[s], side);
                    hipStreamWaitEvent(st, ea[s], 0);
                    b_kernel<<<wg, 256, 0, st>>>(tab[s], out, ticks);
                    hipEventRecord(eb[s], st);
                },
                reps);
            const double rec_only = per_pass_us(   // what the event record after B costs the main stream
                [&](int k) {
                    b_kernel<<<wg, 256, 0, st>>>(tab[0], out, ticks);
                    hipEventRecord(eb[k & 1], st);
                },
                reps);
            const double wait_only = per_pass_us(   // what the wait before B costs it (the event completed long ago)
                [&](int k) {
                    hipStreamWaitEvent(st, ea[0], 0);
                    b_kernel<<<wg, 256, 0, st>>>(tab[0], out, ticks);
                },
                reps);
            const double side_free = per_pass_us(   // A on the side stream with no ordering at all: is the overlap itself free?
                [&](int k) {
                    a_kernel<<<106, 256, 0, side>>>(tab[1], 500, 0.999999);
                    b_kernel<<<wg, 256, 0, st>>>(tab[0], out, ticks);
                },
                reps);
            const double any_order = per_pass_us(   // ONE stream: B, then A of the next pass WITHOUT the barrier bit (hipExtAnyOrderLaunch)
                [&](int k) {
                    b_kernel<<<wg, 256, 0, st>>>(tab[0], out, ticks);
                    hipExtLaunchKernelGGL(a_kernel, dim3(106), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, tab[1], 500, 0.999999);
                },
                reps);
            const double any_order_pair = per_pass_us(   // two spinning kernels of 100 workgroups: 2 x the spin in order, 1 x if they overlap
                [&](int k) {
                    b_kernel<<<100, 256, 0, st>>>(tab[0], out, ticks);
                    hipExtLaunchKernelGGL(b_kernel, dim3(100), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, (const double*)tab[0], out, ticks);
                },
                reps);
            printf("   one stream, A of the next pass launched any-order behind B: %.2f | two 100-workgroup spinners, the second any-order: %.2f\n", any_order, any_order_pair);
            {   // stream memory operations: the side stream writes k behind A(k), the main stream waits for value >= k in front of B(k)
                static uint64_t* sig = nullptr;
                static uint64_t base = 0;
                if (!sig) {
                    if (hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory) != hipSuccess) printf("   no signal memory\n");
                    else hipMemset(sig, 0, 8);
                }
                if (sig) {
                    hipError_t e1 = hipSuccess, e2 = hipSuccess;
                    const uint64_t b0 = base;
                    const double wv = per_pass_us(
                        [&](int k) {
                            const uint64_t v = b0 + (uint64_t)k + 1;
                            a_kernel<<<106, 256, 0, side>>>(tab[k & 1], 500, 0.999999);
                            hipError_t a = hipStreamWriteValue64(side, sig, v, 0);
                            hipError_t b = hipStreamWaitValue64(st, sig, v, hipStreamWaitValueGte, 0xffffffffffffffffull);
                            if (a != hipSuccess) e1 = a;
                            if (b != hipSuccess) e2 = b;
                            b_kernel<<<wg, 256, 0, st>>>(tab[k & 1], out, ticks);
                        },
                        reps);
                    base += 1000;
                    const double wv_only = per_pass_us(   // the wait alone, on a value reached long ago
                        [&](int k) {
                            hipStreamWaitValue64(st, sig, 1, hipStreamWaitValueGte, 0xffffffffffffffffull);
                            b_kernel<<<wg, 256, 0, st>>>(tab[0], out, ticks);
                        },
                        reps);
                    printf("   A + write-value on the side stream, wait-value + B on the main stream: %.2f (%s / %s) | wait-value on an old value + B: %.2f\n", wv,
                           hipGetErrorName(e1), hipGetErrorName(e2), wv_only);
                }
            }
            hipEvent_t ad[8], hv[2];
            for (auto& e : ad) hipEventCreateWithFlags(&e, hipEventDisableTiming);
            for (auto& e : hv) hipEventCreateWithFlags(&e, hipEventDisableTiming);
            const double ahead_rec = per_pass_us(   // B, then A of the next pass on the side stream with an event behind it
                [&](int k) {
                    b_kernel<<<wg, 256, 0, st>>>(tab[0], out, ticks);
                    a_kernel<<<106, 256, 0, side>>>(tab[1], 500, 0.999999);
                    hipEventRecord(ad[k & 7], side);
                },
                reps);
            int late = 0;
            const double ahead_query = per_pass_us(   // ... and the next pass asks the host whether it has finished
                [&](int k) {
                    if (k > 50 && hipEventQuery(ad[(k - 1) & 7]) != hipSuccess) ++late;
                    b_kernel<<<wg, 256, 0, st>>>(tab[0], out, ticks);
                    a_kernel<<<106, 256, 0, side>>>(tab[1], 500, 0.999999);
                    hipEventRecord(ad[k & 7], side);
                },
                reps);
            const double ahead_full = per_pass_us(   // ... and every 4 passes an event on the main stream that the side stream waits for 4 passes later
                [&](int k) {
                    if (k > 50 && hipEventQuery(ad[(k - 1) & 7]) != hipSuccess) ++late;
                    b_kernel<<<wg, 256, 0, st>>>(tab[0], out, ticks);
                    if ((k + 1) % 4 == 0) hipEventRecord(hv[(k / 4) & 1], st);
                    if ((k + 1) % 4 == 0 && k >= 8) hipStreamWaitEvent(side, hv[((k + 1) / 4) & 1], 0);
                    a_kernel<<<106, 256, 0, side>>>(tab[1], 500, 0.999999);
                    hipEventRecord(ad[k & 7], side);
                },
                reps);
            printf("   A of the next pass on the side stream + event record there %.2f | + host query %.2f | + one event per 4 passes across %.2f (late queries %d)\n",
                   ahead_rec, ahead_query, ahead_full, late);
            printf("   B + event record %.2f | wait on a completed event + B %.2f | A on the side stream unordered %.2f\n", rec_only, wait_only, side_free);
            const double alone = per_pass_us([&](int k) { b_kernel<<<wg, 256, 0, st>>>(tab[0], out, ticks); }, reps);
            // host time of the two-stream form: a burst of 16 passes into an empty queue
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < 16; ++k) {
                const int s = k & 1;
                hipStreamWaitEvent(side, eb[s], 0);
                a_kernel<<<106, 256, 0, side>>>(tab[s], 500, 0.999999);
                hipEventRecord(ea[s], side);
                hipStreamWaitEvent(st, ea[s], 0);
                b_kernel<<<wg, 256, 0, st>>>(tab[s], out, ticks);
                hipEventRecord(eb[s], st);
            }
            const double host = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 16;
            hipDeviceSynchronize();
            printf("B = %4d workgroups spinning %3lld us: one stream A,B %.2f us per pass | two streams %.2f | B alone %.2f | host time of the two-stream form %.2f us per pass\n",
                   wg, us, one, two, alone, host);
        }
    return 0;
}
