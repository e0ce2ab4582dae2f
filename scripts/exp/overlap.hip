// micro-benchmark: do MFMAs of one wave and VALU work of another wave on the SAME SIMD overlap?
// 512-thread workgroups: waves 0-3 (one per SIMD) run NM MFMAs per iteration, waves 4-7 run NV VALU ops per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>   // 0: both roles, 1: MFMA waves only work, 2: VALU waves only work, 3: each wave does both (MFMA then VALU, independent)
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
    f32x16 c0 = {0}, c1 = {0};
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = 0.001f * (threadIdx.x + i);
    const bool do_m = MODE == 3 || (wave < 4 && MODE != 2), do_v = MODE == 3 || (wave >= 4 && MODE != 1);
    for (int it = 0; it < iters; ++it) {
        if (do_m) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0); }
        }
        if (do_v) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[i], 0.999f, -0.001f)) + x[(i + 1) & 15] * 0.5f;
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// every wave: 8 MFMAs and 8 x 8 x (fma, exp, fma) per iteration, interleaved in program order: 1 MFMA then NV VALU groups
template <int PIN>
__global__ __launch_bounds__(512) void k2(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
    f32x16 c0 = {0}, c1 = {0};
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = 0.001f * (threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (u & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) { const int e = (u & 1) * 8 + i; x[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[e], 0.999f, -0.001f)) + x[(e + 1) & 15] * 0.5f; }
            if (PIN) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 24, 0); }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// attention-like ratio: 8 MFMAs and 8 x (2 fma + 2 exp + 2 add + 1 cvt-ish) = 56 VALU per iteration, interleaved 1 MFMA : 7 VALU
__global__ __launch_bounds__(768) void k3(float* out, int iters, int mode) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
    f32x16 c0 = {0}, c1 = {0};
    float x[16], rs = 0.f;
    for (int i = 0; i < 16; ++i) x[i] = 0.001f * (threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (mode != 2) { if (u & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0); else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0); }
            __builtin_amdgcn_sched_barrier(0);
            if (mode != 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i) { const int e = 2 * u + i; x[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[e], 0.999f, -0.001f)); rs += x[e]; }
                x[(2 * u + 5) & 15] *= 0.5f;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = rs;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
void run3(int threads, int mode, const char* name) {
    float* out; hipMalloc(&out, 4 << 20);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k3, dim3(256), dim3(threads), 0, 0, out, iters, mode);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k3, dim3(256), dim3(threads), 0, 0, out, iters, mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %d waves/SIMD: %.1f ns per iteration per wave-on-SIMD\n", name, threads / 256, ms * 1e6 / iters / (threads / 256));
}
template <int PIN> void run2(int threads, const char* name) {
    float* out; hipMalloc(&out, 4 << 20);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k2<PIN>, dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k2<PIN>, dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %.1f us  (%.1f ns per iteration per wave-on-SIMD: 8 MFMAs + 64 x (fma, exp, fma))\n", name, ms * 1e3, ms * 1e6 / iters / (threads / 256));
}
template <int MODE> float run(const char* name) {
    float* out; hipMalloc(&out, 4 << 20);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %.1f us  (%.1f ns per iteration: 8 MFMAs = 256 cyc; 64 x (fma, exp, fma) VALU)\n", name, ms * 1e3, ms * 1e6 / iters);
    return ms;
}
int main() {
    run<1>("MFMA waves only (4 waves idle)");
    run<2>("VALU waves only (4 waves idle)");
    run<0>("MFMA waves + VALU waves, same SIMDs");
    run<3>("every wave both (2 waves/SIMD)");
    for (int w = 1; w <= 3; ++w) {
        run3(256 * w, 1, "8 MFMA only");
        run3(256 * w, 2, "8 x (2 fma, 2 exp, 2 add, 1 mul) only");
        run3(256 * w, 0, "interleaved 1 MFMA : 7 VALU");
    }
    run2<0>(256, "interleaved in-wave, 1 wave/SIMD");
    run2<1>(256, "interleaved + pinned, 1 wave/SIMD");
    run2<0>(512, "interleaved in-wave, 2 waves/SIMD");
    run2<1>(512, "interleaved + pinned, 2 waves/SIMD");
    return 0;
}
