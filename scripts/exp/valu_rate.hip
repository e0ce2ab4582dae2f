// Round 6 micro-benchmark: issue rate of scalar vs packed fp32 FMAs in pure-VALU code, and of four GELU formulations (the A-S 7.1.26 form of
// tc_common.h, packed and scalar, against a clamped odd minimax polynomial of the normal CDF), 4 waves per SIMD, operands in registers.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -I transception_amd/csrc -o scripts/exp/valu_rate scripts/exp/valu_rate.hip && scripts/exp/valu_rate
#include "tc_common.h"
#include <cstdio>
#include <vector>

#define TC_PHI_X0 4.2f
#define TC_PHI_C0 3.9890743800e-01f
#define TC_PHI_C1 -6.6360416413e-02f
#define TC_PHI_C2 9.8301434219e-03f
#define TC_PHI_C3 -1.1141502210e-03f
#define TC_PHI_C4 9.4574729576e-05f
#define TC_PHI_C5 -5.7608060352e-06f
#define TC_PHI_C6 2.3436991163e-07f
#define TC_PHI_C7 -5.6334089611e-09f
#define TC_PHI_C8 5.9982045543e-11f

__device__ __forceinline__ float phi_poly(float x) {               // Phi(x) ~ 0.5 + t P(t^2), t = clamp(x, -X0, X0)
    const float t = __builtin_amdgcn_fmed3f(x, -TC_PHI_X0, TC_PHI_X0), s = t * t;
    float p = TC_PHI_C8;
    p = fmaf(p, s, TC_PHI_C7); p = fmaf(p, s, TC_PHI_C6); p = fmaf(p, s, TC_PHI_C5); p = fmaf(p, s, TC_PHI_C4);
    p = fmaf(p, s, TC_PHI_C3); p = fmaf(p, s, TC_PHI_C2); p = fmaf(p, s, TC_PHI_C1); p = fmaf(p, s, TC_PHI_C0);
    return fmaf(t, p, 0.5f);
}
__device__ __forceinline__ tc_f32x2 phi_poly2(tc_f32x2 x) {
    const tc_f32x2 t = {__builtin_amdgcn_fmed3f(x.x, -TC_PHI_X0, TC_PHI_X0), __builtin_amdgcn_fmed3f(x.y, -TC_PHI_X0, TC_PHI_X0)};
    const tc_f32x2 s = t * t;
    tc_f32x2 p = s * TC_PHI_C8 + TC_PHI_C7;
    p = p * s + TC_PHI_C6; p = p * s + TC_PHI_C5; p = p * s + TC_PHI_C4; p = p * s + TC_PHI_C3; p = p * s + TC_PHI_C2; p = p * s + TC_PHI_C1; p = p * s + TC_PHI_C0;
    return t * p + 0.5f;
}

template <int MODE> __global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, float seed) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = seed * (float)(threadIdx.x % 61 - 30) * 0.1f + 0.01f * e;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {                                 // 8 independent scalar FMA chains
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[e]) : "v"(seed), "v"(0.25f));
        } else if constexpr (MODE == 1) {                          // the same flops as 4 packed chains
            tc_f32x2* q = reinterpret_cast<tc_f32x2*>(v);
            const tc_f32x2 a = {seed, seed}, b = {0.25f, 0.25f};
#pragma unroll
            for (int e = 0; e < 4; ++e) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(q[e]) : "v"(a), "v"(b));
        } else if constexpr (MODE == 2) {                          // A-S, packed (what the kernels run today)
#pragma unroll
            for (int e = 0; e < 8; e += 2) { const tc_f32x2 u = gelu_f2_fast(tc_f32x2{v[e], v[e + 1]}); v[e] = u.x * 1.5f - 0.3f; v[e + 1] = u.y * 1.5f + 0.2f; }
        } else if constexpr (MODE == 3) {                          // A-S, scalar
#pragma unroll
            for (int e = 0; e < 8; ++e) { float pdf; v[e] = v[e] * gelu_cdf_pdf<true>(v[e], pdf) * 1.5f - 0.3f; }
        } else if constexpr (MODE == 4) {                          // polynomial, scalar
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * phi_poly(v[e]) * 1.5f - 0.3f;
        } else if constexpr (MODE == 5) {                          // polynomial, packed
#pragma unroll
            for (int e = 0; e < 8; e += 2) { const tc_f32x2 x = {v[e], v[e + 1]}; const tc_f32x2 u = x * phi_poly2(x); v[e] = u.x * 1.5f - 0.3f; v[e + 1] = u.y * 1.5f + 0.2f; }
        } else if constexpr (MODE == 6) {                          // A-S value + gradient (the backward's pair), packed
#pragma unroll
            for (int e = 0; e < 8; e += 2) { tc_f32x2 pdf; const tc_f32x2 x = {v[e], v[e + 1]}; const tc_f32x2 cdf = gelu_cdf_pdf2_fast(x, pdf); const tc_f32x2 u = x * cdf + (cdf + x * pdf) * 0.1f; v[e] = u.x - 0.3f; v[e + 1] = u.y + 0.2f; }
        } else if constexpr (MODE == 8) {                          // 8 x v_dot2_f32_bf16 (two bf16 products into an fp32 accumulator)
            const unsigned w = __float_as_uint(seed) | 0x3f803f80u;
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(v[e]) : "v"(w), "v"(0x3e803e80u));
        } else if constexpr (MODE == 9) {                          // 8 x v_perm_b32
            unsigned* u = reinterpret_cast<unsigned*>(v);
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[e]) : "v"(0x12345678u), "v"(0x05040100u));
        } else if constexpr (MODE == 10) {                         // 8 x v_mul_lo_u32
            unsigned* u = reinterpret_cast<unsigned*>(v);
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[e]) : "v"(0x9e3779b1u));
        } else if constexpr (MODE == 11) {                         // 8 x v_mul_i32_i24
            int* u = reinterpret_cast<int*>(v);
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(u[e]) : "v"(3));
        } else if constexpr (MODE == 12) {                         // 8 x v_exp_f32
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("v_exp_f32 %0, %0" : "+v"(v[e]));
        } else if constexpr (MODE == 13) {                         // 4 x v_cvt_pk_bf16_f32 (8 elements)
            unsigned* u = reinterpret_cast<unsigned*>(v);
#pragma unroll
            for (int e = 0; e < 8; e += 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[e]) : "v"(v[e]), "v"(v[e + 1]));
        } else {                                                   // polynomial value + exp for the density, scalar
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float x = v[e], cdf = phi_poly(x), pdf = 0.39894228f * __expf(-0.5f * x * x); v[e] = x * cdf + (cdf + x * pdf) * 0.1f - 0.3f; }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> void run(const char* what, float* out) {
    const int iters = 4096, grid = 1024 * 4;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 0.37f);
    hipEventRecord(a, 0);
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 0.37f);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    const double elems = 10.0 * grid * 256 * 8.0 * iters;
    // cycles per wave-instruction-equivalent: 1024 SIMDs at ~2.4 GHz
    printf("%-52s %8.3f ms   %.3f ps per element   %.1f SIMD-cycles per 64 elements at 2.4 GHz\n", what, ms / 10, ms * 1e9 / elems, ms * 1e-3 * 2.4e9 * 1024 / (elems / 64));
}

int main() {
    float* out;
    hipMalloc(&out, 1024 * 4 * 256 * sizeof(float));
    run<0>("8 x v_fma_f32 per iteration", out);
    run<1>("4 x v_pk_fma_f32 per iteration (same flops)", out);
    run<2>("GELU A-S 7.1.26, packed (today)", out);
    run<3>("GELU A-S 7.1.26, scalar", out);
    run<4>("GELU clamped polynomial CDF, scalar", out);
    run<5>("GELU clamped polynomial CDF, packed", out);
    run<6>("GELU + GELU' A-S, packed (today's backward)", out);
    run<7>("GELU + GELU' polynomial CDF + exp density, scalar", out);
    run<8>("8 x v_dot2_f32_bf16", out);
    run<9>("8 x v_perm_b32", out);
    run<10>("8 x v_mul_lo_u32", out);
    run<11>("8 x v_mul_i32_i24", out);
    run<12>("8 x v_exp_f32", out);
    run<13>("4 x v_cvt_pk_bf16_f32 (8 elements)", out);
    return 0;
}
