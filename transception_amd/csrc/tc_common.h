// Shared device helpers for the TransCeption gfx950 kernels.
// One storage type parameter T per kernel: float (parity path) or bf16_t (raw 16-bit storage, fp32 math).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

#include "../../include/transception_hip.h"

typedef unsigned short bf16_t;

#define TC_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
// fp32 -> bf16, round to nearest even, through the hardware conversion (v_cvt_pk_bf16_f32 on gfx950: ONE instruction per pair; the
// integer form -- add 0x7fff + lsb, shift, mask, or -- cost ~5.5 VALU operations per element, a third of the arithmetic of the
// element-wise kernels that store bf16).
typedef float tc_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 tc_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    const tc_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, tc_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }

// IEEE half storage: a distinct 16-bit tag type (bf16_t is `unsigned short`), same raw-bit handling, different conversions
struct f16_t { unsigned short v; };
typedef _Float16 tc_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float h2f(f16_t v) { return (float)__builtin_bit_cast(_Float16, v.v); }
__device__ __forceinline__ unsigned pack2h(float lo, float hi) {
    const tc_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, tc_f16x2));      // round to nearest even
}
__device__ __forceinline__ f16_t f2h(float f) { f16_t r; r.v = __builtin_bit_cast(unsigned short, (_Float16)f); return r; }
// the two halves of a 32-bit word as floats
template <typename T> __device__ __forceinline__ void unpack2(unsigned w, float& lo, float& hi);
template <> __device__ __forceinline__ void unpack2<bf16_t>(unsigned w, float& lo, float& hi) { lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xffff0000u); }
template <> __device__ __forceinline__ void unpack2<f16_t>(unsigned w, float& lo, float& hi) {
    const tc_f16x2 h = __builtin_bit_cast(tc_f16x2, w);
    lo = (float)h.x; hi = (float)h.y;
}
template <typename T> __device__ __forceinline__ unsigned pack2(float lo, float hi);
template <> __device__ __forceinline__ unsigned pack2<bf16_t>(float lo, float hi) { return pack2bf(lo, hi); }
template <> __device__ __forceinline__ unsigned pack2<f16_t>(float lo, float hi) { return pack2h(lo, hi); }

// matrix-core flavour of a 16-bit storage type: MFMA operand vector, its element type and the 32x32x16 instruction
typedef float tc_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 tc_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 tc_f16x8 __attribute__((ext_vector_type(8)));
template <typename H> struct TcHalf;
template <> struct TcHalf<bf16_t> {
    typedef tc_bf16x8 v8; typedef __bf16 e;
    static __device__ __forceinline__ tc_f32x16 mfma(v8 a, v8 b, tc_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct TcHalf<f16_t> {
    typedef tc_f16x8 v8; typedef _Float16 e;
    static __device__ __forceinline__ tc_f32x16 mfma(v8 a, v8 b, tc_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <> __device__ __forceinline__ float ldf<f16_t>(const f16_t* p) { return h2f(*p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }
template <> __device__ __forceinline__ void stf<f16_t>(f16_t* p, float v) { *p = f2h(v); }

// 4-element vector access (16 B for float, 8 B for bf16); caller guarantees alignment.
template <typename T> __device__ __forceinline__ float4 ld4(const T* p);
template <> __device__ __forceinline__ float4 ld4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 ld4<bf16_t>(const bf16_t* p) {
    uint2 r = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u),
                       __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
}
template <> __device__ __forceinline__ float4 ld4<f16_t>(const f16_t* p) {
    const uint2 r = *reinterpret_cast<const uint2*>(p);
    float4 o;
    unpack2<f16_t>(r.x, o.x, o.y); unpack2<f16_t>(r.y, o.z, o.w);
    return o;
}
template <typename T> __device__ __forceinline__ void st4(T* p, float4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, float4 v) {
    uint2 r;
    r.x = pack2bf(v.x, v.y);
    r.y = pack2bf(v.z, v.w);
    *reinterpret_cast<uint2*>(p) = r;
}

template <> __device__ __forceinline__ void st4<f16_t>(f16_t* p, float4 v) {
    uint2 r;
    r.x = pack2h(v.x, v.y);
    r.y = pack2h(v.z, v.w);
    *reinterpret_cast<uint2*>(p) = r;
}

// Sums / maxima over aligned groups of G = 2 .. 64 lanes, every lane of the group receiving the result.  Inside a 16-lane row the
// partner comes through DPP (quad_perm, row_half_mirror, row_mirror: a VALU operand modifier); __shfl_xor compiles to ds_bpermute_b32,
// an LDS-crossbar instruction with ~100 cycles of latency per dependent step -- four of those per LayerNorm row reduction.
template <int CTRL> __device__ __forceinline__ float tc_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int G> __device__ __forceinline__ float tc_group_sum(float v) {
    if (G >= 2) v += tc_dpp<0xB1>(v);                  // quad_perm [1,0,3,2]
    if (G >= 4) v += tc_dpp<0x4E>(v);                  // quad_perm [2,3,0,1]
    if (G >= 8) v += tc_dpp<0x141>(v);                 // row_half_mirror: lane i <-> 7 - i of its half row (both quads hold their sum)
    if (G >= 16) v += tc_dpp<0x140>(v);                // row_mirror: lane i <-> 15 - i of its row
    if (G >= 32) v += __shfl_xor(v, 16, 64);
    if (G >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}
template <int G> __device__ __forceinline__ float tc_group_max(float v) {
    if (G >= 2) v = fmaxf(v, tc_dpp<0xB1>(v));
    if (G >= 4) v = fmaxf(v, tc_dpp<0x4E>(v));
    if (G >= 8) v = fmaxf(v, tc_dpp<0x141>(v));
    if (G >= 16) v = fmaxf(v, tc_dpp<0x140>(v));
    if (G >= 32) v = fmaxf(v, __shfl_xor(v, 16, 64));
    if (G >= 64) v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}

// n / d for small non-negative n and a RUN-TIME d (pixel / tile indices; n + d < 2^22): hipcc expands an unsigned division into a hoisted
// reciprocal and, per quotient, a v_mul_hi_u32, two v_mul_lo_u32 and eight compare / select / subtract corrections -- round 6 found a fifth of
// the tiled MixFFN kernels' vector instructions there.  floor((n + 0.5) / d) in fp32 is exact on that range ((n + 0.5) / d is at least 0.5 / d
// away from an integer, the two roundings move it by < (n / d + 1) 2^-23) and costs a convert, an FMA and a convert; the remainder one
// multiply-add.  (It is the instruction COUNT that matters: scripts/exp/valu_rate.hip measures v_mul_lo_u32, v_mul_i32_i24, v_perm_b32 and
// v_dot2_f32_bf16 all at 4.3 SIMD-cycles per wave instruction against 3.0 for v_fma_f32 and 8.8 for v_exp_f32 -- 32-bit integer multiplies
// are not quarter-rate on this chip, and replacing them by 24-bit ones changed nothing.)
__device__ __forceinline__ int tc_mul24(int a, int b) { return a * b; }                 // index products of small operands (named for the reader)
__device__ __forceinline__ int tc_mad24(int a, int b, int c) { return a * b + c; }
struct SDiv { float inv, half; int d; };
__device__ __forceinline__ SDiv sdiv_make(int d) { SDiv s; s.inv = 1.0f / (float)d; s.half = 0.5f * s.inv; s.d = d; return s; }
__device__ __forceinline__ int sdiv(int n, const SDiv& s) { return (int)fmaf((float)n, s.inv, s.half); }
__device__ __forceinline__ int smod(int n, int q, const SDiv& s) { return n - q * s.d; }
// An index the compiler must treat as new at every use: with divisions this cheap it hoists every loop-invariant quotient out of the tile loop and
// spills it (ffn_bwd_dw_kernel: 22 more scratch words, +12 % time) -- three instructions recomputed beat a scratch reload.
__device__ __forceinline__ int tc_opaque(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ float wave_sum(float v) { return tc_group_sum<64>(v); }
__device__ __forceinline__ float wave_max(float v) { return tc_group_max<64>(v); }

// Exact-erf GELU (nn.GELU() default, MSTr.py:894) and its derivative.  erf through Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7,
// below fp32 resolution of the products it enters): with z = x / sqrt(2),  erf(|z|) = 1 - (a1 t + .. + a5 t^5) exp(-z^2),
// t = 1 / (1 + p |z|).  exp(-z^2) = exp(-x^2/2) is the Gaussian the derivative needs anyway, so value and gradient cost one
// exp + one reciprocal + a dozen FMAs -- libm's erff alone was ~25 instructions and made the GELU LayerNorm kernels VALU-bound.
// FAST: the hardware reciprocal (v_rcp_f32, 1 ulp) instead of the correctly rounded one -- __frcp_rn and `1.0f / x` expand to the IEEE
// division sequence (v_div_scale x 2, v_rcp, four FMAs, v_div_fmas, v_div_fixup: ~10 instructions); the 16-bit storage kernels take
// FAST (the difference is far below their storage rounding), the fp32 parity path keeps the exact form.
template <bool FAST> __device__ __forceinline__ float tc_rcp(float x) { return FAST ? __builtin_amdgcn_rcpf(x) : __frcp_rn(x); }
template <bool FAST = false> __device__ __forceinline__ float gelu_cdf_pdf(float x, float& pdf) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float g = __expf(-z * z);                               // exp(-x^2 / 2)
    const float t = tc_rcp<FAST>(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float half_erfc = 0.5f * poly * t * g;                  // 0.5 * erfc(|z|)
    pdf = 0.39894228040143267794f * g;
    return x >= 0.f ? 1.0f - half_erfc : half_erfc;               // Phi(x)
}
__device__ __forceinline__ float gelu_f(float x) { float pdf; return x * gelu_cdf_pdf(x, pdf); }
// by storage type (gradient): exact reciprocal for fp32, the hardware one for the 16-bit types
template <typename T> __device__ __forceinline__ float gelu_grad_fT(float x) {
    float pdf;
    const float cdf = gelu_cdf_pdf<!std::is_same<T, float>::value>(x, pdf);
    return cdf + x * pdf;
}
template <typename T> __device__ __forceinline__ float sigmoid_fT(float x) { return tc_rcp<!std::is_same<T, float>::value>(1.0f + __expf(-x)); }
// Two elements at a time on the packed fp32 pipe (v_pk_mul / v_pk_fma / v_pk_add: two lanes of arithmetic per issue slot; only
// the two exp and the two reciprocals stay scalar transcendental issues) -- same formula, same rounding per element.
__device__ __forceinline__ tc_f32x2 gelu_cdf_pdf2(tc_f32x2 x, tc_f32x2& pdf) {
    const tc_f32x2 ax = {fabsf(x.x), fabsf(x.y)};
    const tc_f32x2 z = ax * 0.70710678118654752440f;
    const tc_f32x2 nz2 = -z * z;
    const tc_f32x2 g = {__expf(nz2.x), __expf(nz2.y)};
    const tc_f32x2 den = z * 0.3275911f + 1.0f;
    const tc_f32x2 t = {__frcp_rn(den.x), __frcp_rn(den.y)};
    tc_f32x2 poly = t * 1.061405429f + (-1.453152027f);
    poly = poly * t + 1.421413741f;
    poly = poly * t + (-0.284496736f);
    poly = poly * t + 0.254829592f;
    const tc_f32x2 half_erfc = poly * t * g * 0.5f;
    pdf = g * 0.39894228040143267794f;
    const tc_f32x2 one_m = 1.0f - half_erfc;
    return tc_f32x2{x.x >= 0.f ? one_m.x : half_erfc.x, x.y >= 0.f ? one_m.y : half_erfc.y};
}
// The same with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of the correctly rounded one (__frcp_rn expands to the IEEE division
// sequence, ~10 instructions per element): for the 16-bit storage kernels, where the difference is far below the storage rounding.
__device__ __forceinline__ tc_f32x2 gelu_cdf_pdf2_fast(tc_f32x2 x, tc_f32x2& pdf) {
    const tc_f32x2 ax = {fabsf(x.x), fabsf(x.y)};
    const tc_f32x2 z = ax * 0.70710678118654752440f;
    const tc_f32x2 nz2 = -z * z;
    const tc_f32x2 g = {__expf(nz2.x), __expf(nz2.y)};
    const tc_f32x2 den = z * 0.3275911f + 1.0f;
    const tc_f32x2 t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    tc_f32x2 poly = t * 1.061405429f + (-1.453152027f);
    poly = poly * t + 1.421413741f;
    poly = poly * t + (-0.284496736f);
    poly = poly * t + 0.254829592f;
    const tc_f32x2 half_erfc = poly * t * g * 0.5f;
    pdf = g * 0.39894228040143267794f;
    const tc_f32x2 one_m = 1.0f - half_erfc;
    return tc_f32x2{x.x >= 0.f ? one_m.x : half_erfc.x, x.y >= 0.f ? one_m.y : half_erfc.y};
}
__device__ __forceinline__ tc_f32x2 gelu_f2_fast(tc_f32x2 x) { tc_f32x2 pdf; return x * gelu_cdf_pdf2_fast(x, pdf); }
// Forward-only GELU of the 16-bit storage kernels (round 6).  Where only the VALUE is needed the Gaussian is not, and the normal CDF is smooth
// enough for a polynomial:  Phi(x) ~ 0.5 + t P(t^2), t = clamp(x, -4.2, 4.2), P of degree 8 in t^2 fitted for the smallest maximum ABSOLUTE error
// (scripts/exp/fit_phi_poly.py: 1.02e-5 over the whole line in fp32 Horner arithmetic; Phi(-4.2) = 1.3e-5 is the clamp's share).  Twelve full-rate
// instructions per element against sixteen + v_exp_f32 + v_rcp_f32 (8.8 cycles each): 30 against 58 SIMD-cycles per 64 elements
// (scripts/exp/valu_rate.hip).  The error is absolute, not relative: x Phi(x) is off by <= 1.02e-5 |x|, under 0.3 % of a bf16 / 2 % of an fp16
// spacing wherever |GELU| >= 4e-3 -- below that (x < -2.8) the value is within 1.02e-5 |x| of the exact one instead of within an ulp of it.  The
// backward kernels keep the A-S form: they need exp(-x^2 / 2) anyway and get Phi from it for three more instructions.
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
__device__ __forceinline__ float gelu_poly(float x) {
    const float t = __builtin_amdgcn_fmed3f(x, -TC_PHI_X0, TC_PHI_X0), s = t * t;
    float p = fmaf(TC_PHI_C8, s, TC_PHI_C7);
    p = fmaf(p, s, TC_PHI_C6); p = fmaf(p, s, TC_PHI_C5); p = fmaf(p, s, TC_PHI_C4); p = fmaf(p, s, TC_PHI_C3);
    p = fmaf(p, s, TC_PHI_C2); p = fmaf(p, s, TC_PHI_C1); p = fmaf(p, s, TC_PHI_C0);
    return x * fmaf(t, p, 0.5f);
}
__device__ __forceinline__ tc_f32x2 gelu_poly2(tc_f32x2 x) {
    const tc_f32x2 t = {__builtin_amdgcn_fmed3f(x.x, -TC_PHI_X0, TC_PHI_X0), __builtin_amdgcn_fmed3f(x.y, -TC_PHI_X0, TC_PHI_X0)};
    const tc_f32x2 s = t * t;
    tc_f32x2 p = s * TC_PHI_C8 + TC_PHI_C7;
    p = p * s + TC_PHI_C6; p = p * s + TC_PHI_C5; p = p * s + TC_PHI_C4; p = p * s + TC_PHI_C3;
    p = p * s + TC_PHI_C2; p = p * s + TC_PHI_C1; p = p * s + TC_PHI_C0;
    return x * (t * p + 0.5f);
}
__device__ __forceinline__ tc_f32x2 gelu_f2(tc_f32x2 x) { tc_f32x2 pdf; return x * gelu_cdf_pdf2(x, pdf); }
__device__ __forceinline__ tc_f32x2 gelu_grad_f2(tc_f32x2 x) { tc_f32x2 pdf; const tc_f32x2 cdf = gelu_cdf_pdf2(x, pdf); return cdf + x * pdf; }
__device__ __forceinline__ tc_f32x2 gelu_grad_f2_fast(tc_f32x2 x) { tc_f32x2 pdf; const tc_f32x2 cdf = gelu_cdf_pdf2_fast(x, pdf); return cdf + x * pdf; }
__device__ __forceinline__ float gelu_grad_f(float x) {
    float pdf;
    const float cdf = gelu_cdf_pdf(x, pdf);
    return cdf + x * pdf;
}
// by storage type: the exact form for fp32, the polynomial for the 16-bit types
template <typename T> __device__ __forceinline__ float gelu_fT(float x) {
    if constexpr (std::is_same<T, float>::value) return gelu_f(x); else return gelu_poly(x);
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float hswish_f(float x) { return x * fminf(fmaxf(x + 3.0f, 0.0f), 6.0f) * (1.0f / 6.0f); }
__device__ __forceinline__ float hswish_grad_f(float x) {
    return x <= -3.0f ? 0.0f : (x >= 3.0f ? 1.0f : (2.0f * x + 3.0f) * (1.0f / 6.0f));
}
// IFF activation, reference MSTr.py:1270-1286: y * min(silu(y+3)/6, 1)
__device__ __forceinline__ float coordact_f(float y) {
    const float u = y + 3.0f;
    return y * fminf(u * sigmoid_f(u) * (1.0f / 6.0f), 1.0f);
}
__device__ __forceinline__ float coordact_grad_f(float y) {
    const float u = y + 3.0f, s = sigmoid_f(u);
    const float g = u * s * (1.0f / 6.0f);
    if (g >= 1.0f) return 1.0f;
    const float dg = (s + u * s * (1.0f - s)) * (1.0f / 6.0f);
    return g + y * dg;
}

// act codes shared by BN-apply / GEMM epilogues
__device__ __forceinline__ float apply_act(int act, float v) {
    switch (act) {
        case TC_ACT_HSWISH: return hswish_f(v);
        case TC_ACT_COORD: return coordact_f(v);
        case TC_ACT_SIGMOID: return sigmoid_f(v);
        case TC_ACT_GELU: return gelu_f(v);
        case TC_ACT_RELU: return fmaxf(v, 0.f);
        default: return v;
    }
}
__device__ __forceinline__ float act_grad(int act, float z) {
    switch (act) {
        case TC_ACT_HSWISH: return hswish_grad_f(z);
        case TC_ACT_COORD: return coordact_grad_f(z);
        case TC_ACT_SIGMOID: { const float s = sigmoid_f(z); return s * (1.0f - s); }
        case TC_ACT_GELU: return gelu_grad_f(z);
        case TC_ACT_RELU: return z > 0.f ? 1.0f : 0.f;
        default: return 1.0f;
    }
}

static inline int tc_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TC_OK : TC_ERR_LAUNCH;
}
static inline int tc_blocks(long long work, int per_block, int cap = 4096) {
    long long b = (work + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}

#define TC_DISPATCH_DTYPE(dtype, ...)                                  \
    do {                                                               \
        if ((dtype) == TC_F32) { using T = float; __VA_ARGS__; }       \
        else if ((dtype) == TC_BF16) { using T = bf16_t; __VA_ARGS__; }\
        else if ((dtype) == TC_F16) { using T = f16_t; __VA_ARGS__; }  \
        else return TC_ERR_ARG;                                        \
    } while (0)
