// What one Horner step costs a wave on gfx950, by how its coefficient reaches the fma -- and what a second / third wave per SIMD or a
// second independent chain in the same wave buys.  Patterns (all: one dependent fp64 chain per lane unless said otherwise):
//   P0  v_fma_f64 with the coefficient resident in a scalar register pair
//   P1  s_mov_b32 x2 (64-bit literal into a scalar pair) + v_fma_f64            (what the compiler emits for inlined ocml polynomials at high register pressure)
//   P2  s_mov_b32 x2 + s_nop 0 + v_fma_f64                                     (the same through an inline-asm "s" operand)
//   P3  v_mov_b64 (coefficient from a vector register) + v_fmac_f64             (two-address form)
//   P4  two independent chains of P0 interleaved;  P5  two independent chains of P1 interleaved
//   hipcc --offload-arch=gfx950 -O3 -o horner_issue horner_issue.hip && ./horner_issue
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(s) s s s s s s s s s s s s s s s s

template <int P>
__global__ __launch_bounds__(256) void kern(double* out, double z, double k, int iters) {
    double x = threadIdx.x * 1e-9 + 1.0, y = threadIdx.x * 2e-9 + 0.5, kv = k;
    for (int it = 0; it < iters; ++it) {
        if (P == 0) asm volatile(REP16("v_fma_f64 %0, %0, %1, %2\n") : "+v"(x) : "v"(z), "s"(k));
        if (P == 1)
            asm volatile(REP16("s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x3e112e0b\n v_fma_f64 %0, %0, %1, s[20:21]\n") : "+v"(x) : "v"(z) : "s20", "s21");
        if (P == 2)
            asm volatile(REP16("s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x3e112e0b\n s_nop 0\n v_fma_f64 %0, %0, %1, s[20:21]\n")
                         : "+v"(x)
                         : "v"(z)
                         : "s20", "s21");
        if (P == 3)
            asm volatile(REP16("v_mov_b64 %1, %3\n v_fmac_f64 %1, %0, %2\n v_mov_b64 %0, %3\n v_fmac_f64 %0, %1, %2\n") : "+v"(x), "+v"(y) : "v"(z), "v"(kv));
        if (P == 4) asm volatile(REP16("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n") : "+v"(x), "+v"(y) : "v"(z), "s"(k));
        if (P == 5)
            asm volatile(REP16("s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x3e112e0b\n v_fma_f64 %0, %0, %2, s[20:21]\n s_mov_b32 s22, 0x55555555\n s_mov_b32 "
                               "s23, 0x3e112e0b\n v_fma_f64 %1, %1, %2, s[22:23]\n")
                         : "+v"(x), "+v"(y)
                         : "v"(z)
                         : "s20", "s21", "s22", "s23");
    }
    out[(long)blockIdx.x * blockDim.x + threadIdx.x] = x + y;
}

template <int P>
void run(const char* name, int steps_per_rep, int vinstr_per_rep, double* out) {
    const int iters = 4000;
    for (int w : {1, 2, 3, 4}) {
        const int blocks = 256 * w;   // 4 waves per workgroup = one per SIMD of a CU
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        for (int r = 0; r < 3; ++r) kern<P><<<blocks, 256>>>(out, 0.999999, 1e-9, iters);   // ramps the clock
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) kern<P><<<blocks, 256>>>(out, 0.999999, 1e-9, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 5;
        const double ns_per_step_wave = ms * 1e6 / ((double)iters * 16 * steps_per_rep);               // wall time per Horner step of ONE wave
        const double valu_per_simd = (double)w * iters * 16 * vinstr_per_rep;                          // wave64 VALU instructions per SIMD
        printf("%-44s waves/SIMD %d: %6.2f ns per step per wave, %5.2f ns per step per SIMD; VALU instr per ns per SIMD %.3f\n", name, w, ns_per_step_wave,
               ns_per_step_wave / w, valu_per_simd / (ms * 1e6));
    }
}

int main() {
    double* out;
    hipMalloc(&out, 256L * 8 * 256 * 8);
    run<0>("P0 v_fma, coefficient in a scalar pair", 1, 1, out);
    run<1>("P1 s_mov x2 + v_fma", 1, 1, out);
    run<2>("P2 s_mov x2 + s_nop + v_fma", 1, 1, out);
    run<3>("P3 v_mov_b64 + v_fmac", 2, 4, out);
    run<4>("P4 two chains of P0 interleaved", 2, 2, out);
    run<5>("P5 two chains of P1 interleaved", 2, 2, out);
    return 0;
}
