// Linear + residual + LayerNorm in one launch (16-bit storage; square Linear, N = K = C in {64, 128, 320}):
//   t = x W^T + b + residual            xn = LayerNorm_C(t) = (t - mean) * rstd * gamma + beta
// The tail of the attention half of an MHCABlock (`proj` + skip, then norm2: MSTr.py:883, 940-942), of a bridge layer (:2288 / 2404-2406)
// and of the decoder's EfficientTransformerBlocks (`reprojection` + skip, then norm2: :141, 167-170): a 7-13 us GEMM whose output a 5-7 us
// LayerNorm launch read straight back.  A workgroup owns RT rows and ALL C output channels, so the row statistics never leave it:
// D^T MFMA tiles (lane = row; wave w: rows (w % RH) * 32.., output channels (w / RH) * 32 NT..).  These launches are short (tens to a few
// hundred workgroups, 4-40 MFMAs a wave), so what they cost is memory round trips: EVERY global load of the workgroup -- its x rows, the
// wave's whole weight panel (KS x NT fragments, registers), the residual rows, bias / gamma / beta -- is issued before the first wait, in the
// order it is needed, and consumed behind counted waits (one round trip per workgroup; hipcc would sink each load to its use).
// bias + residual are added in fp32, t is rounded to the storage type BEFORE the statistics are taken (the backward, and every other reader,
// sees the rounded t), both results leave as whole rows through LDS.  t, xn, mean, rstd are exactly what tc_gemm (+ residual) followed by
// tc_layernorm_fwd leave: their backward entries apply unchanged.
#include "tc_common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct LinLnDev {
    const void *x, *w, *b, *res, *gamma, *beta;
    void *t, *xn;
    float *mean, *rstd;
    long long wstride, gstride;      // elements between the parameter blocks of two weight groups (Linear / LayerNorm)
    int ldx, ldr, ldt, ldn, rows;    // rows per group
    float eps;
};

#define LL_GLOAD128(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr))
#define LL_KEEP128(r) asm volatile("" : "+v"(r))
#define LL_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
template <int N> __device__ __forceinline__ void ll_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N)); }

template <typename T, int C, int RT, int NW>
__global__ __launch_bounds__(NW * 64) void lin_res_ln_kernel(LinLnDev p) {
    constexpr int NTH = NW * 64, RH = RT / 32, CB = NW / RH, NT = C / (CB * 32), KS = C / 16, PT = C + 8, CV = C / 8, NXP = RT * CV / NTH;
    constexpr int KH = KS / 2, NPAR = 3 * CV;
    static_assert(RT * CV % NTH == 0 && C % (CB * 32) == 0 && NW % RH == 0 && NPAR <= NTH && KS * NT + 2 * NXP + 1 < 64, "tiling");
    typedef typename TcHalf<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_[];
    T* xs = reinterpret_cast<T*>(smem_);                  // [RT][PT] x rows, then t
    T* rs = xs + RT * PT;                                 // [RT][PT] residual rows, then xn
    T* par = rs + RT * PT;                                // bias | gamma | beta
    float* st = reinterpret_cast<float*>(par + 3 * C);    // [2][CB][RT]: per channel block, the row's sum / sum of squared deviations
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int g = blockIdx.y;
    const long long row0 = (long long)g * p.rows + (long long)blockIdx.x * RT;
    const int nrow = min(RT, p.rows - (int)blockIdx.x * RT);
    const T* W = reinterpret_cast<const T*>(p.w) + g * p.wstride;
    const int cb = wave / RH, ch0 = cb * (NT * 32), tok = (wave % RH) * 32 + l31;

    // every load of the workgroup, in the order it is consumed
    u32x4 xr[NXP], wr[KS * NT], rr[NXP], pr;
#pragma unroll
    for (int j = 0; j < NXP; ++j) {
        const int i = tid + NTH * j, n = i / CV, c8 = i - n * CV;
        const T* src = reinterpret_cast<const T*>(p.x) + (row0 + (n < nrow ? n : 0)) * p.ldx + c8 * 8;
        LL_GLOAD128(xr[j], src);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const T* src = W + (long long)(ch0 + t * 32 + l31) * C + ks * 16 + hh * 8;
            LL_GLOAD128(wr[ks * NT + t], src);
        }
    {
        const int i = tid < NPAR ? tid : 0, a = i / CV, c8 = i - a * CV;
        const T* src = a == 0 ? reinterpret_cast<const T*>(p.b) + g * p.wstride : (a == 1 ? reinterpret_cast<const T*>(p.gamma) : reinterpret_cast<const T*>(p.beta)) + g * p.gstride;
        src += c8 * 8;
        LL_GLOAD128(pr, src);
    }
#pragma unroll
    for (int j = 0; j < NXP; ++j) {
        const int i = tid + NTH * j, n = i / CV, c8 = i - n * CV;
        const T* src = (p.res ? reinterpret_cast<const T*>(p.res) + (row0 + (n < nrow ? n : 0)) * p.ldr : reinterpret_cast<const T*>(p.x) + row0 * p.ldx) + c8 * 8;
        LL_GLOAD128(rr[j], src);
    }
    tc_f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    ll_wait_vm<KS * NT + 1 + NXP>();                                            // x rows
#pragma unroll
    for (int j = 0; j < NXP; ++j) {
        LL_KEEP128(xr[j]);
        const int i = tid + NTH * j, n = i / CV, c8 = i - n * CV;
        *reinterpret_cast<u32x4*>(xs + n * PT + c8 * 8) = xr[j];
    }
    LL_BARRIER();
    ll_wait_vm<(KS - KH) * NT + 1 + NXP>();                                     // first half of the weight panel
#pragma unroll
    for (int i = 0; i < KH * NT; ++i) LL_KEEP128(wr[i]);
#pragma unroll
    for (int ks = 0; ks < KH; ++ks) {
        const v8 bfr = *reinterpret_cast<const v8*>(xs + tok * PT + ks * 16 + hh * 8);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = TcHalf<T>::mfma(__builtin_bit_cast(v8, wr[ks * NT + t]), bfr, acc[t]);
    }
    ll_wait_vm<1 + NXP>();                                                      // the rest of it
#pragma unroll
    for (int i = KH * NT; i < KS * NT; ++i) LL_KEEP128(wr[i]);
#pragma unroll
    for (int ks = KH; ks < KS; ++ks) {
        const v8 bfr = *reinterpret_cast<const v8*>(xs + tok * PT + ks * 16 + hh * 8);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = TcHalf<T>::mfma(__builtin_bit_cast(v8, wr[ks * NT + t]), bfr, acc[t]);
    }
    ll_wait_vm<0>();                                                            // parameters, residual rows
    LL_KEEP128(pr);
    if (tid < NPAR) *reinterpret_cast<u32x4*>(par + tid * 8) = pr;
#pragma unroll
    for (int j = 0; j < NXP; ++j) {
        LL_KEEP128(rr[j]);
        const int i = tid + NTH * j, n = i / CV, c8 = i - n * CV;
        *reinterpret_cast<u32x4*>(rs + n * PT + c8 * 8) = rr[j];
    }
    LL_BARRIER();                                              // (also: every wave is done reading x -- the tile takes t)
    // t = acc + bias + residual, rounded; the lane's channels: tile t_, register group gq -> ch0 + t_ * 32 + 8 gq + 4 hh + (0..3)
    const bool has_res = p.res != nullptr;
    float s1 = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int c = ch0 + t * 32 + 8 * gq + 4 * hh;
            const float4 bq = ld4<T>(par + c);
            float v0 = acc[t][4 * gq] + bq.x, v1 = acc[t][4 * gq + 1] + bq.y, v2 = acc[t][4 * gq + 2] + bq.z, v3 = acc[t][4 * gq + 3] + bq.w;
            if (has_res) {
                const float4 rq = ld4<T>(rs + tok * PT + c);
                v0 += rq.x; v1 += rq.y; v2 += rq.z; v3 += rq.w;
            }
            const unsigned lo = pack2<T>(v0, v1), hi = pack2<T>(v2, v3);
            *reinterpret_cast<uint2*>(xs + tok * PT + c) = make_uint2(lo, hi);
            unpack2<T>(lo, v0, v1); unpack2<T>(hi, v2, v3);
            acc[t][4 * gq] = v0; acc[t][4 * gq + 1] = v1; acc[t][4 * gq + 2] = v2; acc[t][4 * gq + 3] = v3;
            s1 += (v0 + v1) + (v2 + v3);
        }
    // row statistics: the lane's values + its partner half (same row, other 16 channels of every tile) + the other channel blocks (other waves)
    s1 += __shfl_xor(s1, 32, 64);
    if (hh == 0) st[cb * RT + tok] = s1;
    LL_BARRIER();
    float mean = 0.f;
#pragma unroll
    for (int q = 0; q < CB; ++q) mean += st[q * RT + tok];
    mean *= 1.0f / C;
    float s2 = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = acc[t][r] - mean; s2 += d * d; }
    s2 += __shfl_xor(s2, 32, 64);
    if (hh == 0) st[(CB + cb) * RT + tok] = s2;
    // t leaves as whole rows while the second statistic settles
#pragma unroll
    for (int j = 0; j < NXP; ++j) {
        const int i = tid + NTH * j, n = i / CV, c8 = i - n * CV;
        if (n < nrow) *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(p.t) + (row0 + n) * p.ldt + c8 * 8) = *reinterpret_cast<const u32x4*>(xs + n * PT + c8 * 8);
    }
    LL_BARRIER();
    float var = 0.f;
#pragma unroll
    for (int q = 0; q < CB; ++q) var += st[(CB + q) * RT + tok];
    const float rstd = rsqrtf(var * (1.0f / C) + p.eps);
    if (cb == 0 && hh == 0 && tok < nrow) { p.mean[row0 + tok] = mean; p.rstd[row0 + tok] = rstd; }
    // xn over the residual values this lane alone read
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int c = ch0 + t * 32 + 8 * gq + 4 * hh;
            const float4 gm = ld4<T>(par + C + c), bt = ld4<T>(par + 2 * C + c);
            const float o0 = (acc[t][4 * gq] - mean) * rstd * gm.x + bt.x, o1 = (acc[t][4 * gq + 1] - mean) * rstd * gm.y + bt.y;
            const float o2 = (acc[t][4 * gq + 2] - mean) * rstd * gm.z + bt.z, o3 = (acc[t][4 * gq + 3] - mean) * rstd * gm.w + bt.w;
            *reinterpret_cast<uint2*>(rs + tok * PT + c) = make_uint2(pack2<T>(o0, o1), pack2<T>(o2, o3));
        }
    LL_BARRIER();
#pragma unroll
    for (int j = 0; j < NXP; ++j) {
        const int i = tid + NTH * j, n = i / CV, c8 = i - n * CV;
        if (n < nrow) *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(p.xn) + (row0 + n) * p.ldn + c8 * 8) = *reinterpret_cast<const u32x4*>(rs + n * PT + c8 * 8);
    }
}

template <typename T, int C, int RT, int NW> int linln_launch(const LinLnDev& p, int groups, hipStream_t s) {
    constexpr size_t smem = 2 * ((size_t)2 * RT * (C + 8) + 3 * C) + sizeof(float) * 2 * (NW / (RT / 32)) * RT;
    static bool once = false;
    if (!once) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lin_res_ln_kernel<T, C, RT, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); once = true; }
    hipLaunchKernelGGL((lin_res_ln_kernel<T, C, RT, NW>), dim3((p.rows + RT - 1) / RT, groups), dim3(NW * 64), smem, s, p);
    return tc_launch_status();
}

}  // namespace

extern "C" int tc_linear_ln_supported(int C, int dtype) { return (dtype == TC_BF16 || dtype == TC_F16) && (C == 64 || C == 128 || C == 320); }

extern "C" int tc_linear_ln_fwd(const void* x, int ldx, const void* w, const void* b, long long wstride, const void* res, int ldr, const void* gamma,
                                const void* beta, long long gstride, void* t, int ldt, void* xn, int ldn, float* mean, float* rstd, int groups,
                                int rows, int C, float eps, int dtype, void* stream) {
    if (!x || !w || !b || !gamma || !beta || !t || !xn || !mean || !rstd || groups <= 0 || rows <= 0) return TC_ERR_ARG;
    if (!tc_linear_ln_supported(C, dtype)) return TC_ERR_UNSUPPORTED;
    if (ldx % 8 || ldt % 8 || ldn % 8 || (res && ldr % 8) || wstride % 8 || gstride % 8 ||
        (((uintptr_t)x | (uintptr_t)w | (uintptr_t)t | (uintptr_t)xn | (res ? (uintptr_t)res : 0) | (uintptr_t)b | (uintptr_t)gamma | (uintptr_t)beta) & 15))
        return TC_ERR_ARG;
    LinLnDev p;
    p.x = x; p.w = w; p.b = b; p.res = res; p.gamma = gamma; p.beta = beta; p.t = t; p.xn = xn; p.mean = mean; p.rstd = rstd;
    p.wstride = wstride; p.gstride = gstride; p.ldx = ldx; p.ldr = ldr; p.ldt = ldt; p.ldn = ldn; p.rows = rows; p.eps = eps;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == TC_BF16) {
        if (C == 64) return linln_launch<bf16_t, 64, 64, 4>(p, groups, s);
        if (C == 128) return linln_launch<bf16_t, 128, 64, 4>(p, groups, s);
        return linln_launch<bf16_t, 320, 32, 5>(p, groups, s);
    }
    if (C == 64) return linln_launch<f16_t, 64, 64, 4>(p, groups, s);
    if (C == 128) return linln_launch<f16_t, 128, 64, 4>(p, groups, s);
    return linln_launch<f16_t, 320, 32, 5>(p, groups, s);
}
