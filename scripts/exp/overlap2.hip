// micro-benchmark (round 4): can the VALU stream of one wave run under the MFMAs of ANOTHER wave of the same SIMD if the MFMA wave
// paces itself (s_nop between MFMAs so that it never waits on a busy matrix pipe), and what do priorities change?
// Workgroup = NW waves; waves 0-3 (one per SIMD) are MFMA waves, the others VALU waves (exp + add mix, the softmax's instruction mix).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MF(C) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(C) : "v"(a), "v"(b))
template <int NOP, int PRIO_M, int PRIO_V>
__global__ __launch_bounds__(1024) void k(long long* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
    f32x16 c0 = {0}, c1 = {0};
    float x0 = 0.001f * threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, r0 = 0, r1 = 0;
    const bool mf = wave < 4;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (mf) {
        if (mode != 2) {
            if (PRIO_M) __builtin_amdgcn_s_setprio(PRIO_M);
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    MF(c0);
                    if (NOP > 0) asm volatile("s_nop %0" : : "n"(NOP - 1));
                    MF(c1);
                    if (NOP > 0) asm volatile("s_nop %0" : : "n"(NOP - 1));
                }
            }
        }
    } else if (mode != 1) {
        if (PRIO_V) __builtin_amdgcn_s_setprio(PRIO_V);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 6; ++u)          // 6 x (4 exp + 4 add) = 48 VALU per iteration, two independent chains
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_add_f32 %4, %4, %0\n v_exp_f32 %2, %2\n v_add_f32 %5, %5, %1\n v_exp_f32 %3, %3\n v_add_f32 %4, %4, %2\n s_nop 0\n v_add_f32 %5, %5, %3"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(r0), "+v"(r1));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = r0 + r1 + x0 + x1 + x2 + x3;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + wave] = (t1 - t0) + (s == 12345.f);
}
template <int NOP, int PM, int PV> void run(int nw, const char* name) {
    long long* out; hipMalloc(&out, 256 * 16 * 8);
    const int iters = 2000;
    long long h[256 * 16];
    printf("%-34s", name);
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL((k<NOP, PM, PV>), dim3(256), dim3(nw * 64), 0, 0, out, iters, mode);
        hipLaunchKernelGGL((k<NOP, PM, PV>), dim3(256), dim3(nw * 64), 0, 0, out, iters, mode);
        hipDeviceSynchronize();
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        double mm = 0, mv = 0; int nm = 0, nv = 0;
        for (int bI = 0; bI < 256; ++bI) for (int w = 0; w < nw; ++w) { if (w < 4) mm += h[bI * 16 + w], ++nm; else mv += h[bI * 16 + w], ++nv; }
        printf("  %s: mfma-wave %6.1f  valu-wave %6.1f cyc/iter |", mode == 0 ? "both" : mode == 1 ? "mfma only" : "valu only", mm / nm / iters, mv / nv / iters);
    }
    printf("\n");
    hipFree(out);
}
int main() {
    printf("per iteration: MFMA wave = 8 MFMAs (256 cycles of matrix pipe); each VALU wave = 48 VALU (24 exp + 24 add)\n");
    run<0, 0, 0>(8, "2 waves/SIMD nop0");
    run<4, 0, 0>(8, "2 waves/SIMD nop4");
    run<6, 0, 0>(8, "2 waves/SIMD nop6");
    run<7, 0, 0>(8, "2 waves/SIMD nop7");
    run<8, 0, 0>(8, "2 waves/SIMD nop8");
    run<10, 0, 0>(8, "2 waves/SIMD nop10");
    run<0, 1, 0>(8, "2 waves/SIMD nop0 prioM");
    run<0, 0, 1>(8, "2 waves/SIMD nop0 prioV");
    run<7, 0, 1>(8, "2 waves/SIMD nop7 prioV");
    run<7, 1, 0>(8, "2 waves/SIMD nop7 prioM");
    run<0, 0, 0>(12, "3 waves/SIMD nop0");
    run<7, 0, 0>(12, "3 waves/SIMD nop7");
    run<8, 0, 1>(12, "3 waves/SIMD nop8 prioV");
    run<7, 1, 0>(12, "3 waves/SIMD nop7 prioM");
    run<0, 0, 0>(16, "4 waves/SIMD nop0");
    run<7, 0, 0>(16, "4 waves/SIMD nop7");
    return 0;
}
