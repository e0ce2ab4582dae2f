// micro-benchmark: how fast do 32x32x16 bf16 MFMAs issue when (a) chained on one accumulator, (b) alternating between two, (c) four
// independent accumulators -- with 1, 2 or 3 waves per SIMD.  Prints cycles per MFMA per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 ld_tr(const unsigned short* lo, const unsigned short* hi) {
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(lo));
    const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(hi));
    return __builtin_bit_cast(bf16x8, (s16x8_t)__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ int pi_row(int i) { return 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3); }
template <int MODE>
__global__ void k(float* out, int iters, long long* cyc) {
    constexpr int LDR = 72;
    __shared__ __attribute__((aligned(16))) unsigned short Ks[128 * LDR], Vs[128 * LDR];
    for (int i = threadIdx.x; i < 128 * LDR; i += blockDim.x) { Ks[i] = 0x3c00; Vs[i] = 0x3c00; }
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int koff = pi_row(j) * LDR + 8 * h;
    const int voff = (16 * h + 4 * ((lane & 15) >> 2)) * LDR + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // chain of 8 on one accumulator
#pragma unroll
            for (int u = 0; u < 8; ++u) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        } else if (MODE == 1) {   // alternate two accumulators
#pragma unroll
            for (int u = 0; u < 4; ++u) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0); }
        } else if (MODE == 2) {   // four accumulators
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
            }
        } else if (MODE >= 4) {   // attention core with K fragments (MODE 4) and V^T fragments (MODE 5) read from LDS
            const int sub = it & 3;
            f32x16 s = {0};
            const unsigned short* kp = Ks + 32 * sub * LDR + koff;
#pragma unroll
            for (int u = 0; u < 4; ++u) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(kp + 16 * u), b, s, 0, 0, 0);
            const unsigned short* vp = Vs + 32 * sub * LDR + voff;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                bf16x8 p;
                for (int i = 0; i < 8; ++i) p[i] = (__bf16)s[8 * k2 + i];
                if (MODE == 4) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, p, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, p, c1, 0, 0, 0);
                } else {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_tr(vp + (2 * k2) * LDR, vp + (2 * k2 + 1) * LDR), p, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_tr(vp + (2 * k2) * LDR + 32, vp + (2 * k2 + 1) * LDR + 32), p, c1, 0, 0, 0);
                }
            }
        } else {                  // the attention core's pattern: chain of 4 (fresh accumulator), then 2 x 2 alternating
            f32x16 s = {0};
#pragma unroll
            for (int u = 0; u < 4; ++u) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s, 0, 0, 0);
            bf16x8 p;
            for (int i = 0; i < 8; ++i) p[i] = (__bf16)s[i];
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, p, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, p, c1, 0, 0, 0);
            for (int i = 0; i < 8; ++i) p[i] = (__bf16)s[8 + i];
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, p, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, p, c1, 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE> void run(int wps, const char* name) {
    float* out; long long* cyc; hipMalloc(&out, 4 << 20); hipMalloc(&cyc, 8);
    const int iters = 2000, threads = 256 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 8 * wps;     // MFMAs per SIMD
    printf("%-28s waves/SIMD %d: %.1f us, %.1f ns/MFMA/SIMD = %.1f cyc @2.4GHz, s_memtime ticks/MFMA %.1f, %.0f TF/s\n", name, wps, ms * 1e3, ms * 1e6 / n,
           ms * 1e6 / n * 2.4, (double)c / n, 256.0 * 4 * n * 32768 / (ms * 1e-3) / 1e12);
}
int main() {
    for (int w = 1; w <= 3; ++w) {
        run<3>(w, "attention core pattern"); run<4>(w, "core + K frags from LDS"); run<5>(w, "core + K and V frags");
    }
    return 0;
}
