// MixFFN_skip backward (autograd of MSTr.py:889-902 with DWConv :21-31) with the hidden maps kept on the chip.
//   forward (mixffn.hip) left:  d = dw3x3(h) + bias + h  (storage type) and stat = (mean, rstd) of LayerNorm_4C(d) per pixel
//   launch 1  ffn_bwd_ln_kernel   rows of 64 pixels:  gpre = dy W2 (MFMA), u = LN(d), gp = gpre (.) GELU'(u), LayerNorm backward -> gd,
//                                 dW2 += dy^T GELU(u) (MFMA, the activation recomputed in registers and staged in LDS only), db2, dgamma, dbeta
//   launch 2  ffn_bwd_dw_kernel   pixel tiles with a one-pixel halo, 128 hidden channels at a time:  h = fc1(x) recomputed (MFMA) on the
//                                 haloed tile, dh = dw3x3^T(gd) + gd, dwd / dbd / db1 sums, dx += dh W1 (MFMA), dW1 += dh^T x (MFMA)
//   launch 3  ffn_bwd_reduce_kernel  the per-workgroup fp32 partial sums of both launches -> the gradient arrays
// Only gd crosses HBM between the launches (op-by-op: gp, a, d twice, h, dh three times).  Workgroups are persistent (one per CU):
// weight-gradient accumulators stay in registers / LDS for the whole launch and leave once, as one partial per workgroup.
// 16-bit storage types (bf16 / fp16), C = 64.
#include "tc_common.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

// k-row permutation of the LDS tiles whose rows are the reduction index of a transpose-read operand (gemm.hip, gemm_krow)
__device__ __forceinline__ int krow(int k) { return (k & ~15) | ((k & 3) << 2) | ((k >> 2) & 3); }
template <typename V8> __device__ __forceinline__ V8 ld_tr(const bf16_t* lo, const bf16_t* hi) {
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(lo));
    const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(hi));
    return __builtin_bit_cast(V8, (s16x8_t)__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}
template <typename H> __device__ __forceinline__ void up8(const uint4& r, float* o) {
    unpack2<H>(r.x, o[0], o[1]); unpack2<H>(r.y, o[2], o[3]); unpack2<H>(r.z, o[4], o[5]); unpack2<H>(r.w, o[6], o[7]);
}
template <typename H> __device__ __forceinline__ uint4 pk8(const float* o) {
    return make_uint4(pack2<H>(o[0], o[1]), pack2<H>(o[2], o[3]), pack2<H>(o[4], o[5]), pack2<H>(o[6], o[7]));
}

// per-(weight group, workgroup) partial sums, fp32: launch 1 [dW2 C x 4C][db2 C][dgamma 4C][dbeta 4C], launch 2 [dW1 4C x C][db1 4C][dwd 4C x 9][dbd 4C]
template <int C> struct BwdPart {
    static constexpr int C4 = 4 * C;
    static constexpr int oW2 = 0, oB2 = oW2 + C * C4, oG = oB2 + C, oBt = oG + C4, n1 = oBt + C4;
    static constexpr int oW1 = n1, oB1 = oW1 + C4 * C, oWd = oB1 + C4, oBd = oWd + 9 * C4, oPG = oBd + C4, oPB = oPG + C, n = oPB + C;   // (+ the fused norm2's dgamma, dbeta)
};

struct FfnBwdDev {
    const void* x; const void* dy; const void* d; const float* stat;
    const void* w1; const void* b1; const void* wd; const void* gamma; const void* beta; const void* w2;
    void* dx; void* gd; float* part;
    long long sdy, wstride;
    int ldx, lddy, lddx, B, H, W, M, acc_dx, nblk;
    int TH, TW, tilesH, tilesW, HW2, HP, MT, IP, MT2, KS, ntiles;
    const void* pre_g; const void* pre_b; float pre_eps;           // LayerNorm(C) ahead of fc1 (applied to x as it is loaded, differentiated where dx leaves), or null
};

#ifdef TC_FFNB_TIMING
// phase stamps (experiment builds only, scripts/exp/ffnb_timing.py): thread 0 of the first 16 workgroups of weight group 0 adds the
// cycles between consecutive stamps to its row of the table
__device__ long long g_ffnb_dbg[2 * 16 * 16];
#define BSTAMP(k) do { if (bs_) { const long long t_ = __builtin_readcyclecounter(); bs_[k] += t_ - bt_; bt_ = t_; } } while (0)
#define BSTAMP_INIT(kern) long long bt_ = __builtin_readcyclecounter(); \
    long long* bs_ = (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 16) ? g_ffnb_dbg + ((kern) * 16 + blockIdx.x) * 16 : nullptr; \
    if (bs_) for (int k_ = 0; k_ < 16; ++k_) bs_[k_] = 0;
#else
#define BSTAMP(k)
#define BSTAMP_INIT(kern)
#endif

// --------------------------------------------------------------------------------------------------------------------- launch 1
template <int C> struct LnCfg {
    static constexpr int C4 = 4 * C, NW = 8, NTH = 512, CW = C4 / NW, NT = CW / 32, KK = C / 16, OB = C / 32, P = 32;   // P = 64 needs ~290 registers (spills, and a spill reload after the prefetch waits for it)
    static constexpr int PH = C4 + 8, PX = C + 8, HC = C4 / 8, XC = C / 8, ND = P * HC / NTH, NY = (P * XC + NTH - 1) / NTH;
    static constexpr int PGG = C4 + 4;                               // fp32 row pitch of the parked gg tile
    static constexpr size_t smem = (size_t)P * PH * 2 + (size_t)P * PX * 2 + (size_t)2 * C4 * 4 + (size_t)NW * P * 8 + (size_t)P * 8 + (size_t)P * PGG * 4 + (size_t)C4 * PX * 2;
    static_assert(NTH % XC == 0 && P * HC % NTH == 0, "strip ownership");
    static_assert(smem <= 160 * 1024, "LDS");
};

template <typename H, int C>
__global__ __launch_bounds__(512, 2) void ffn_bwd_ln_kernel(const FfnBwdDev p) {
    using K = LnCfg<C>;
    using PT = BwdPart<C>;
    using V8 = typename TcHalf<H>::v8;
    constexpr int C4 = K::C4, CW = K::CW, NT = K::NT, KK = K::KK, OB = K::OB, P = K::P, PH = K::PH, PX = K::PX, HC = K::HC, XC = K::XC;
    constexpr int ND = K::ND, NY = K::NY, NTH = K::NTH, NW = K::NW, PGG = K::PGG;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* dt = reinterpret_cast<bf16_t*>(smem);                          // [P][PH]  d, then GELU(LN(d)), then gd; rows k-permuted
    bf16_t* yt = dt + P * PH;                                              // [P][PX]  dy; rows k-permuted
    float* gbs = reinterpret_cast<float*>(yt + P * PX);                    // gamma[C4], beta[C4]
    float2* psum = reinterpret_cast<float2*>(gbs + 2 * C4);                // [NW][P] per-wave LayerNorm-backward row sums
    float2* fst = psum + NW * P;                                           // [P] mean, rstd
    float* ggs = reinterpret_cast<float*>(fst + P);                        // [P][PGG] gp * gamma, parked between the two LayerNorm-backward phases
    bf16_t* w2t = reinterpret_cast<bf16_t*>(ggs + P * PGG);                // [C4][PX] W2^T (rows = hidden channel): A operand of gpre^T = W2^T dy^T

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int gi = lane & 15, gq2 = (lane >> 4) & 1;
    const int g = blockIdx.y, M = p.M;
    BSTAMP_INIT(0);
    const long long wo = (long long)g * p.wstride;
    const H* D = reinterpret_cast<const H*>(p.d) + (long long)g * M * C4;
    H* GD = reinterpret_cast<H*>(p.gd) + (long long)g * M * C4;
    const H* DY = reinterpret_cast<const H*>(p.dy) + (long long)g * p.sdy;
    const float2* ST = reinterpret_cast<const float2*>(p.stat) + (long long)g * M;
    const unsigned short* W2 = reinterpret_cast<const unsigned short*>(p.w2) + wo;

    {
        const H* gm = reinterpret_cast<const H*>(p.gamma) + wo;
        const H* bt = reinterpret_cast<const H*>(p.beta) + wo;
        for (int i = tid; i < C4; i += NTH) { gbs[i] = ldf<H>(gm + i); gbs[C4 + i] = ldf<H>(bt + i); }
    }
    {   // W2 [C][C4] -> W2^T in LDS, once per launch: lanes walk the hidden channels (coalesced 2-byte reads), eight output channels per
        // 16-byte store; every load of the staging is issued before the first store waits for one
        constexpr int NWT = (C4 * XC + NTH - 1) / NTH;
        unsigned short wv[NWT][8];
#pragma unroll
        for (int k = 0; k < NWT; ++k) {
            const int i = min(tid + k * NTH, C4 * XC - 1), og = i / C4, ch = i - og * C4;
#pragma unroll
            for (int e = 0; e < 8; ++e) wv[k][e] = W2[(long long)(og * 8 + e) * C4 + ch];
        }
#pragma unroll
        for (int k = 0; k < NWT; ++k) {
            const int i = tid + k * NTH, og = i / C4, ch = i - og * C4;
            if (i < C4 * XC)
                *reinterpret_cast<uint4*>(w2t + ch * PX + og * 8) = make_uint4((unsigned)wv[k][0] | ((unsigned)wv[k][1] << 16), (unsigned)wv[k][2] | ((unsigned)wv[k][3] << 16),
                                                                                 (unsigned)wv[k][4] | ((unsigned)wv[k][5] << 16), (unsigned)wv[k][6] | ((unsigned)wv[k][7] << 16));
        }
    }

    uint4 dr[ND], yr[NY];
    float2 sr = make_float2(0.f, 0.f);
    auto fetch = [&](int blk) __attribute__((always_inline)) {
        const long long r0 = (long long)blk * P;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int s = tid + i * NTH, px = s / HC, cg = s - px * HC;
            const long long row = r0 + px;
            dr[i] = *reinterpret_cast<const uint4*>(D + (row < M ? row : 0) * C4 + cg * 8);
        }
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            const int s = tid + i * NTH, px = s / XC, cg = s - px * XC;
            const long long row = r0 + px;
            yr[i] = *reinterpret_cast<const uint4*>(DY + ((row < M && s < P * XC) ? row : 0) * p.lddy + cg * 8);
        }
        if (tid < P) { const long long row = r0 + tid; sr = ST[row < M ? row : 0]; }
    };
    float db2p[OB];                                                 // (wave 0) column sums of dy, from the dW2 operand fragments
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) db2p[ob] = 0.f;
    auto put = [&](int blk) __attribute__((always_inline)) {
        const long long r0 = (long long)blk * P;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int s = tid + i * NTH, px = s / HC, cg = s - px * HC;
            *reinterpret_cast<uint4*>(dt + krow(px) * PH + cg * 8) = (r0 + px < M) ? dr[i] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            const int s = tid + i * NTH, px = s / XC, cg = s - px * XC;
            if (s < P * XC) {
                const uint4 v = (r0 + px < M) ? yr[i] : make_uint4(0u, 0u, 0u, 0u);
                *reinterpret_cast<uint4*>(yt + krow(px) * PX + cg * 8) = v;
            }
        }
        if (tid < P) fst[tid] = (r0 + tid < M) ? sr : make_float2(0.f, 0.f);
    };

    float dgam[NT][16], dbet[NT][16];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dgam[nt][r] = 0.f; dbet[nt][r] = 0.f; }
    f32x16 acc2[OB][NT];
#pragma unroll
    for (int ob = 0; ob < OB; ++ob)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[ob][nt][r] = 0.f;

    int blk = blockIdx.x;
    if (blk < p.nblk) fetch(blk);
    __syncthreads();                                               // gamma / beta
    BSTAMP(0);
    for (; blk < p.nblk; blk += gridDim.x) {
        put(blk);
        BSTAMP(1);
        __syncthreads();
        BSTAMP(2);
        if (blk + (int)gridDim.x < p.nblk) fetch(blk + gridDim.x); // lands under this block's arithmetic
        const long long r0 = (long long)blk * P;
        // ---- per 32-pixel half: gpre^T = W2^T dy^T (lane = pixel, registers = 16 of this wave's hidden channels); u = LN(d);
        //      gp = gpre * GELU'(u); a = GELU(u) -> LDS (over d); gg = gp * gamma -> LDS (fp32) and its two row sums; dgamma / dbeta
        uint2 dk[P / 32][NT][4];
#pragma unroll
        for (int pb = 0; pb < P / 32; ++pb) {
            const int px = pb * 32 + l31;
            f32x16 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
            {
                const bf16_t* yp = yt + krow(px) * PX + 8 * hh;
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    const V8 yv = *reinterpret_cast<const V8*>(yp + kk * 16);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[nt] = TcHalf<H>::mfma(*reinterpret_cast<const V8*>(w2t + (wave * CW + nt * 32 + l31) * PX + kk * 16 + 8 * hh), yv, acc[nt]);
                }
            }
            bf16_t* rowp = dt + krow(px) * PH;
            float* ggp = ggs + px * PGG;
            const float2 st = fst[px];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int ch0 = wave * CW + nt * 32 + 8 * gq + 4 * hh;
                    const uint2 dv = *reinterpret_cast<const uint2*>(rowp + ch0);
                    dk[pb][nt][gq] = dv;
                    const float4 g4 = *reinterpret_cast<const float4*>(gbs + ch0), b4 = *reinterpret_cast<const float4*>(gbs + C4 + ch0);
                    float xv[4], av[4], gv4[4];
                    unpack2<H>(dv.x, xv[0], xv[1]); unpack2<H>(dv.y, xv[2], xv[3]);
                    const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const tc_f32x2 x2 = {xv[e], xv[e + 1]}, g2 = {gv[e], gv[e + 1]}, be2 = {bv[e], bv[e + 1]};
                        const tc_f32x2 xh = (x2 - st.x) * st.y;
                        const tc_f32x2 u = xh * g2 + be2;
                        tc_f32x2 pdf;
                        const tc_f32x2 cdf = gelu_cdf_pdf2_fast(u, pdf);
                        const tc_f32x2 a2 = u * cdf, gpr = cdf + u * pdf;
                        const tc_f32x2 gpre = {acc[nt][4 * gq + e], acc[nt][4 * gq + e + 1]};
                        const tc_f32x2 gp = gpre * gpr;
                        dbet[nt][4 * gq + e] += gp.x; dbet[nt][4 * gq + e + 1] += gp.y;
                        const tc_f32x2 gx = gp * xh;
                        dgam[nt][4 * gq + e] += gx.x; dgam[nt][4 * gq + e + 1] += gx.y;
                        const tc_f32x2 gg = gp * g2;
                        s1 += gg.x + gg.y;
                        const tc_f32x2 gh = gg * xh;
                        s2 += gh.x + gh.y;
                        gv4[e] = gg.x; gv4[e + 1] = gg.y;
                        av[e] = a2.x; av[e + 1] = a2.y;
                    }
                    *reinterpret_cast<uint2*>(rowp + ch0) = make_uint2(pack2<H>(av[0], av[1]), pack2<H>(av[2], av[3]));
                    *reinterpret_cast<float4*>(ggp + ch0) = make_float4(gv4[0], gv4[1], gv4[2], gv4[3]);
                }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (hh == 0) psum[wave * P + px] = make_float2(s1, s2);
        }
        BSTAMP(4);
        __syncthreads();
        BSTAMP(5);
        // ---- dW2 += dy^T a over the 64 pixels: both operands by transpose reads (k = pixel = LDS row)
#pragma unroll
        for (int ks = 0; ks < P / 16; ++ks) {
            const int rr = 16 * ks + 2 * hh + 4 * (gi >> 2), cc = 16 * gq2 + 4 * (gi & 3);
            V8 bfr[NT], afr[OB];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bfr[nt] = ld_tr<V8>(dt + rr * PH + wave * CW + nt * 32 + cc, dt + (rr + 1) * PH + wave * CW + nt * 32 + cc);
#pragma unroll
            for (int ob = 0; ob < OB; ++ob) afr[ob] = ld_tr<V8>(yt + rr * PX + ob * 32 + cc, yt + (rr + 1) * PX + ob * 32 + cc);
            if (wave == 0) {
#pragma unroll
                for (int ob = 0; ob < OB; ++ob) {
                    const uint4 q = __builtin_bit_cast(uint4, afr[ob]);
                    float f[8];
                    up8<H>(q, f);
                    db2p[ob] += ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
                }
            }
#pragma unroll
            for (int ob = 0; ob < OB; ++ob)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc2[ob][nt] = TcHalf<H>::mfma(afr[ob], bfr[nt], acc2[ob][nt]);
        }
        BSTAMP(6);
        // ---- LayerNorm backward: gd = rstd * (gg - S1 / 4C - xhat * S2 / 4C), over a in LDS (this wave's columns only)
#pragma unroll
        for (int pb = 0; pb < P / 32; ++pb) {
            const int px = pb * 32 + l31;
            bf16_t* rowp = dt + krow(px) * PH;
            const float2 st = fst[px];
            float S1 = 0.f, S2 = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) { const float2 t = psum[w * P + px]; S1 += t.x; S2 += t.y; }
            const float k1 = S1 * (1.0f / (float)C4), k2 = S2 * (1.0f / (float)C4);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int ch0 = wave * CW + nt * 32 + 8 * gq + 4 * hh;
                    float xv[4], o[4];
                    unpack2<H>(dk[pb][nt][gq].x, xv[0], xv[1]); unpack2<H>(dk[pb][nt][gq].y, xv[2], xv[3]);
                    const float4 g4 = *reinterpret_cast<const float4*>(ggs + px * PGG + ch0);
                    const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xh = (xv[e] - st.x) * st.y;
                        o[e] = st.y * (gg[e] - k1 - xh * k2);
                    }
                    *reinterpret_cast<uint2*>(rowp + ch0) = make_uint2(pack2<H>(o[0], o[1]), pack2<H>(o[2], o[3]));
                }
        }
        BSTAMP(7);
        __syncthreads();
        BSTAMP(8);
        // ---- gd leaves as whole pixel rows
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int s = tid + i * NTH, px = s / HC, cg = s - px * HC;
            if (r0 + px < M) *reinterpret_cast<uint4*>(GD + (r0 + px) * C4 + cg * 8) = *reinterpret_cast<const uint4*>(dt + krow(px) * PH + cg * 8);
        }
        BSTAMP(9);
        __syncthreads();
        BSTAMP(10);
    }
    // ---- this workgroup's partial sums
    float* PB = p.part + ((long long)g * gridDim.x + blockIdx.x) * PT::n;
#pragma unroll
    for (int ob = 0; ob < OB; ++ob)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh, ch = wave * CW + nt * 32 + l31;
                PB[PT::oW2 + o * C4 + ch] = acc2[ob][nt][r];
            }
    // dgamma / dbeta: sums over the 32 pixel lanes of a half-wave, as a transposing butterfly (31 shuffles per 16 values instead of 80):
    // afterwards lane l holds the total of register l & 15
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float t = which ? dbet[nt][r] : dgam[nt][r]; v[r] = t + __shfl_xor(t, 16, 64); }
#pragma unroll
            for (int m = 8, n = 16; m >= 1; m >>= 1, n >>= 1) {
                const bool up = (lane & m) != 0;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < n / 2) {
                        const float keep = up ? v[j + n / 2] : v[j], send = up ? v[j] : v[j + n / 2];
                        v[j] = keep + __shfl_xor(send, m, 64);
                    }
            }
            if (l31 < 16) {
                const int r = l31, ch = wave * CW + nt * 32 + 8 * (r >> 2) + 4 * hh + (r & 3);
                PB[(which ? PT::oBt : PT::oG) + ch] = v[0];
            }
        }
    if (wave == 0) {
#pragma unroll
        for (int ob = 0; ob < OB; ++ob) {
            const float v = db2p[ob] + __shfl_xor(db2p[ob], 32, 64);
            if (hh == 0) PB[PT::oB2 + ob * 32 + l31] = v;
        }
    }
    BSTAMP(11);
}

// --------------------------------------------------------------------------------------------------------------------- launch 2
template <int C> struct DwCfg {
    static constexpr int C4 = 4 * C, NW = 8, NTH = 512, CH = 128, NCH = C4 / CH, KK1 = C / 16, KKC = CH / 16, NB = C / 32, XC = C / 8, GC = CH / 8;
    static constexpr int MPMAX = 144, IPMAX = 128, MT2MAX = IPMAX / 32, TWMAX = 16, THMAX = 8;
    static constexpr int PX = C + 8, PG = CH + 8, PO = C + 4;
    static constexpr int NXR = (MPMAX * XC + NTH - 1) / NTH, NGR = (MPMAX * GC + NTH - 1) / NTH;
    static constexpr size_t o_xs = 0, o_gs = o_xs + (size_t)MPMAX * PX * 2, o_hs = o_gs + (size_t)MPMAX * PG * 2, o_w1 = o_hs + (size_t)(MPMAX + 1) * PG * 2,
                            o_tap = o_w1 + (size_t)C4 * PX * 2, o_b1 = o_tap + (size_t)9 * C4 * 4, o_pre = o_b1 + (size_t)C4 * 4, smem = o_pre + (size_t)4 * C * 4;
    static_assert((size_t)NW * NCH * 22 * 64 * 4 <= o_w1, "the final fold of the depthwise sums aliases the tiles");
    static_assert(MT2MAX * NB <= NW && (CH / 32) * NB <= NW, "one MFMA block per wave");
    static_assert((size_t)IPMAX * PO * 4 <= (size_t)MPMAX * PG * 2, "the fp32 dx stage aliases the gd tile");
    static_assert(smem <= 160 * 1024, "LDS");
};

template <typename H, int C, bool PRE>
__global__ __launch_bounds__(512, 2) void ffn_bwd_dw_kernel(const FfnBwdDev p) {
    using K = DwCfg<C>;
    using PT = BwdPart<C>;
    using V8 = typename TcHalf<H>::v8;
    constexpr int C4 = K::C4, CH = K::CH, NCH = K::NCH, KK1 = K::KK1, KKC = K::KKC, NB = K::NB, XC = K::XC, GC = K::GC, PX = K::PX, PG = K::PG, PO = K::PO;
    constexpr int NTH = K::NTH, MPMAX = K::MPMAX, NXR = K::NXR, NGR = K::NGR, TWMAX = K::TWMAX;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem + K::o_xs);               // [MPMAX][PX]      x on the haloed tile
    bf16_t* gs = reinterpret_cast<bf16_t*>(smem + K::o_gs);               // [MPMAX][PG]      gd, this chunk's channels, haloed tile
    bf16_t* hs = reinterpret_cast<bf16_t*>(smem + K::o_hs);               // [MPMAX + 1][PG]  h (haloed), then dh (inner pixels); last row = zeros
    bf16_t* w1s = reinterpret_cast<bf16_t*>(smem + K::o_w1);              // [C4][PX]         W1, rows k-permuted inside each 16-row group
    float* taps = reinterpret_cast<float*>(smem + K::o_tap);              // [9][C4]
    float* b1s = reinterpret_cast<float*>(smem + K::o_b1);                // [C4]
    float* stg = reinterpret_cast<float*>(smem + K::o_gs);                // fp32 dx stage over the gd tile
    float* pre = reinterpret_cast<float*>(smem + K::o_pre);               // fused norm2: gamma[C], beta[C], then this workgroup's dgamma[C], dbeta[C]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int gi = lane & 15, gq2 = (lane >> 4) & 1;
    const int g = blockIdx.y, M = p.M;
    BSTAMP_INIT(1);
    const long long wo = (long long)g * p.wstride, imgpix = (long long)p.H * p.W;
    const H* X = reinterpret_cast<const H*>(p.x) + (long long)g * M * p.ldx;
    const H* GD = reinterpret_cast<const H*>(p.gd) + (long long)g * M * C4;
    H* DX = reinterpret_cast<H*>(p.dx) + (long long)g * M * p.lddx;
    const H* W1 = reinterpret_cast<const H*>(p.w1) + wo;
    const int TH = p.TH, TW = p.TW, HW2 = p.HW2, HP = p.HP, MT = p.MT, IP = p.IP, Himg = p.H, Wimg = p.W;
    const SDiv dHW2 = sdiv_make(HW2), dTW = sdiv_make(TW), dTLW = sdiv_make(p.tilesW), dTLH = sdiv_make(p.tilesH);      // (tc_common.h: run-time divisors)

    auto stage_params = [&]() __attribute__((always_inline)) {
        const H* wd = reinterpret_cast<const H*>(p.wd) + wo;
        const H* b1 = reinterpret_cast<const H*>(p.b1) + wo;
        // (every global load of the staging in flight before the first LDS store waits for one)
        constexpr int NTAP = (9 * C4 + NTH - 1) / NTH, NW1 = (C4 * XC + NTH - 1) / NTH;
        float tv[NTAP];
        uint4 w1v[NW1];
#pragma unroll
        for (int k = 0; k < NTAP; ++k) { const int i = tid + k * NTH; tv[k] = ldf<H>(wd + (i < 9 * C4 ? i : 0)); }
#pragma unroll
        for (int k = 0; k < NW1; ++k) {
            const int s = min(tid + k * NTH, C4 * XC - 1), r = s / XC, cg = s - r * XC;
            w1v[k] = *reinterpret_cast<const uint4*>(W1 + (long long)r * C + cg * 8);
        }
#pragma unroll
        for (int k = 0; k < NTAP; ++k) { const int i = tid + k * NTH, ch = i / 9, t = i - ch * 9; if (i < 9 * C4) taps[t * C4 + ch] = tv[k]; }
        for (int i = tid; i < C4; i += NTH) b1s[i] = ldf<H>(b1 + i);
        for (int i = tid; i < PG; i += NTH) hs[MPMAX * PG + i] = 0;
        if constexpr (PRE) {
            for (int i = tid; i < C; i += NTH) {
                pre[i] = ldf<H>(reinterpret_cast<const H*>(p.pre_g) + wo + i); pre[C + i] = ldf<H>(reinterpret_cast<const H*>(p.pre_b) + wo + i);
                pre[2 * C + i] = 0.f; pre[3 * C + i] = 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < NW1; ++k) {                            // W1 stays in LDS for the whole launch
            const int s = tid + k * NTH, r = s / XC, cg = s - r * XC;
            if (s < C4 * XC) *reinterpret_cast<uint4*>(w1s + krow(r) * PX + cg * 8) = w1v[k];
        }
    };

    auto tile_org = [&](int tidx, int& b, int& oh0, int& ow0) __attribute__((always_inline)) {
        const int t1 = sdiv(tidx, dTLW), tx = smod(tidx, t1, dTLW);
        b = sdiv(t1, dTLH);
        const int ty = smod(t1, b, dTLH);
        b = __builtin_amdgcn_readfirstlane(b); oh0 = __builtin_amdgcn_readfirstlane(tc_mul24(ty, TH)); ow0 = __builtin_amdgcn_readfirstlane(tc_mul24(tx, TW));
    };
    auto halo_in = [&](int pix, int oh0, int ow0, int& ih, int& iw) __attribute__((always_inline)) {
        pix = tc_opaque(pix);
        const int hy = sdiv(pix, dHW2), hx = smod(pix, hy, dHW2);
        ih = oh0 - 1 + hy; iw = ow0 - 1 + hx;
        return pix < HP && (unsigned)ih < (unsigned)Himg && (unsigned)iw < (unsigned)Wimg;
    };
    uint4 xr[NXR], gr[NGR];
    auto xfetch = [&](int tidx) __attribute__((always_inline)) {
        int b, oh0, ow0;
        tile_org(tidx, b, oh0, ow0);
        const H* xb = X + (long long)b * imgpix * p.ldx;
#pragma unroll
        for (int i = 0; i < NXR; ++i) {
            const int s = tid + i * NTH, pix = s / XC, cg = s - pix * XC;
            int ih, iw;
            const bool ok = halo_in(pix, oh0, ow0, ih, iw);
            xr[i] = *reinterpret_cast<const uint4*>(xb + (ok ? tc_mad24(tc_mad24(ih, Wimg, iw), p.ldx, cg * 8) : 0));
        }
    };
    auto xput = [&](int tidx) __attribute__((always_inline)) {
        int b, oh0, ow0;
        tile_org(tidx, b, oh0, ow0);
#pragma unroll
        for (int i = 0; i < NXR; ++i) {
            const int s = tid + i * NTH, pix = s / XC, cg = s - pix * XC;
            int ih, iw;
            const bool ok = halo_in(pix, oh0, ow0, ih, iw);
            uint4 v = ok ? xr[i] : make_uint4(0u, 0u, 0u, 0u);
            if constexpr (PRE) {                                    // n2 = LayerNorm(x): XC consecutive lanes hold one pixel
                float f[8];
                up8<H>(v, f);
                float sm = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) sm += f[e];
                sm = tc_group_sum<XC>(sm);
                const float mean = sm * (1.0f / C);
                float q2 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { f[e] -= mean; q2 += f[e] * f[e]; }
                q2 = tc_group_sum<XC>(q2);
                const float rstd = rsqrtf(q2 * (1.0f / C) + p.pre_eps);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = f[e] * rstd * pre[cg * 8 + e] + pre[C + cg * 8 + e];
                v = pk8<H>(f);
            }
            if (pix < HP) *reinterpret_cast<uint4*>(xs + tc_mul24(pix, PX) + cg * 8) = v;
        }
    };
    auto gfetch = [&](int tidx, int c) __attribute__((always_inline)) {
        int b, oh0, ow0;
        tile_org(tidx, b, oh0, ow0);
        const H* gb = GD + (long long)b * imgpix * C4 + c * CH;
#pragma unroll
        for (int i = 0; i < NGR; ++i) {
            const int s = tid + i * NTH, pix = s / GC, cg = s - pix * GC;
            int ih, iw;
            const bool ok = halo_in(pix, oh0, ow0, ih, iw);
            gr[i] = *reinterpret_cast<const uint4*>(gb + (ok ? tc_mad24(ih, Wimg, iw) * C4 + cg * 8 : 0));
        }
    };
    auto gput = [&](int tidx) __attribute__((always_inline)) {
        int b, oh0, ow0;
        tile_org(tidx, b, oh0, ow0);
#pragma unroll
        for (int i = 0; i < NGR; ++i) {
            const int s = tid + i * NTH, pix = s / GC, cg = s - pix * GC;
            int ih, iw;
            const bool ok = halo_in(pix, oh0, ow0, ih, iw);
            if (pix < HP) *reinterpret_cast<uint4*>(gs + tc_mul24(pix, PG) + cg * 8) = ok ? gr[i] : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    // halo-tile row of inner pixel q (row-major over the TH x TW inner pixels)
    auto hrow = [&](int q) __attribute__((always_inline)) { q = tc_opaque(q); const int y = sdiv(q, dTW); return tc_mad24(y + 1, HW2, smod(q, y, dTW) + 1); };

    f32x16 accw[NCH];
    tc_f32x2 aw[NCH][11];                                            // this thread's sums over (its row of every tile) x (its channel pair of chunk c): dwd taps, dbd, db1
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accw[c][r] = 0.f;
#pragma unroll
        for (int t = 0; t < 11; ++t) aw[c][t] = tc_f32x2{0.f, 0.f};
    }
    const int cb = wave & 3, nb = wave >> 2;                        // MFMA block of this wave: (hidden block | pixel block, channel block)

    int tidx = blockIdx.x;
    if (tidx < p.ntiles) { xfetch(tidx); gfetch(tidx, 0); }        // (in flight under the parameter staging)
    stage_params();
    __syncthreads();                                               // parameters in LDS
    BSTAMP(0);
    for (; tidx < p.ntiles; tidx += gridDim.x) {
        int b, oh0, ow0;
        tile_org(tidx, b, oh0, ow0);
        const bool more = tidx + (int)gridDim.x < p.ntiles;
        xput(tidx);
        f32x16 accx;
#pragma unroll
        for (int r = 0; r < 16; ++r) accx[r] = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            gput(tidx);
            BSTAMP(1);
            __syncthreads();
            BSTAMP(2);
            if (c + 1 < NCH) gfetch(tidx, c + 1);
            // ---- h = fc1(x) on the haloed tile for this chunk's channels (zero outside the image: the convolution's padding)
            {
                V8 af[KK1];
#pragma unroll
                for (int kk = 0; kk < KK1; ++kk) af[kk] = *reinterpret_cast<const V8*>(w1s + krow(c * CH + cb * 32 + l31) * PX + kk * 16 + 8 * hh);
                for (int mi = nb; mi < MT; mi += 2) {               // (one block at a time: three interleaved chains cost 36 spilled registers)
                    f32x16 acc;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const float4 bv = *reinterpret_cast<const float4*>(b1s + c * CH + cb * 32 + 8 * gq + 4 * hh);
                        acc[4 * gq] = bv.x; acc[4 * gq + 1] = bv.y; acc[4 * gq + 2] = bv.z; acc[4 * gq + 3] = bv.w;
                    }
                    const int pp = mi * 32 + l31;
                    const bf16_t* xp = xs + tc_mul24(pp < HP ? pp : 0, PX) + 8 * hh;      // (rows beyond the haloed tile: columns nobody keeps)
#pragma unroll
                    for (int kk = 0; kk < KK1; ++kk) acc = TcHalf<H>::mfma(af[kk], *reinterpret_cast<const V8*>(xp + kk * 16), acc);
                    int ih, iw;
                    const bool ok = halo_in(pp, oh0, ow0, ih, iw);
                    const int ppG = tc_mul24(pp, PG);
                    if (pp < HP) {
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            uint2 v = make_uint2(pack2<H>(acc[4 * gq], acc[4 * gq + 1]), pack2<H>(acc[4 * gq + 2], acc[4 * gq + 3]));
                            if (!ok) v = make_uint2(0u, 0u);
                            *reinterpret_cast<uint2*>(hs + ppG + cb * 32 + 8 * gq + 4 * hh) = v;
                        }
                    }
                }
            }
            BSTAMP(3);
            __syncthreads();
            BSTAMP(4);
            // ---- depthwise stage: wave = inner row, lane = channel pair (packed fp32 math on the pair).
            //      Pass 1: dwd[t] += gd * h(shifted), dbd += gd.  (barrier: every read of h is done.)
            //      Pass 2: dh = gd + sum_t w[t] gd(shifted the other way) -> the inner pixels' slots of the h tile, db1 += dh.
            //      Both walk the row with the next column's LDS reads in flight under the current column's arithmetic.
            {
                const int y = wave, chl = 2 * lane, chg = c * CH + chl;
                const bool active = y < TH;
                auto ld2 = [&](const bf16_t* q) __attribute__((always_inline)) { tc_f32x2 v; float a, b; unpack2<H>(*reinterpret_cast<const unsigned*>(q), a, b); v.x = a; v.y = b; return v; };
                if (active) {
                    const bf16_t* hb = hs + (y * HW2) * PG + chl;
                    const bf16_t* gcp = gs + ((y + 1) * HW2 + 1) * PG + chl;
                    tc_f32x2 c0[3], c1[3], c2[3];
#pragma unroll
                    for (int r = 0; r < 3; ++r) { c1[r] = ld2(hb + (r * HW2) * PG); c2[r] = ld2(hb + (r * HW2 + 1) * PG); }
                    unsigned nh[3], ng = *reinterpret_cast<const unsigned*>(gcp);
#pragma unroll
                    for (int r = 0; r < 3; ++r) nh[r] = *reinterpret_cast<const unsigned*>(hb + (r * HW2 + 2) * PG);
                    for (int x = 0; x < TW; ++x) {
                        tc_f32x2 gc;
                        { float a, b; unpack2<H>(ng, a, b); gc.x = a; gc.y = b; }          // zero for a pixel outside the image
#pragma unroll
                        for (int r = 0; r < 3; ++r) { c0[r] = c1[r]; c1[r] = c2[r]; float a, b; unpack2<H>(nh[r], a, b); c2[r].x = a; c2[r].y = b; }
                        const int xn = x + 1 < TW ? x + 1 : x;                               // (the last step re-reads its own column)
                        ng = *reinterpret_cast<const unsigned*>(gcp + xn * PG);
#pragma unroll
                        for (int r = 0; r < 3; ++r) nh[r] = *reinterpret_cast<const unsigned*>(hb + (r * HW2 + xn + 2) * PG);
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky) {
                            aw[c][ky * 3] += gc * c0[ky]; aw[c][ky * 3 + 1] += gc * c1[ky]; aw[c][ky * 3 + 2] += gc * c2[ky];
                        }
                        aw[c][9] += gc;
                    }
                }
                BSTAMP(5);
                __syncthreads();
                BSTAMP(6);
                if (active) {
                    const bool rowin = oh0 + y < Himg;
                    tc_f32x2 tp[9];
#pragma unroll
                    for (int t = 0; t < 9; ++t) { const float2 q = *reinterpret_cast<const float2*>(taps + t * C4 + chg); tp[t].x = q.x; tp[t].y = q.y; }
                    const bf16_t* gb = gs + (y * HW2) * PG + chl;
                    bf16_t* dhp = hs + ((y + 1) * HW2 + 1) * PG + chl;
                    tc_f32x2 c0[3], c1[3], c2[3];
#pragma unroll
                    for (int r = 0; r < 3; ++r) { c1[r] = ld2(gb + (r * HW2) * PG); c2[r] = ld2(gb + (r * HW2 + 1) * PG); }
                    unsigned nh[3];
#pragma unroll
                    for (int r = 0; r < 3; ++r) nh[r] = *reinterpret_cast<const unsigned*>(gb + (r * HW2 + 2) * PG);
                    for (int x = 0; x < TW; ++x) {
#pragma unroll
                        for (int r = 0; r < 3; ++r) { c0[r] = c1[r]; c1[r] = c2[r]; float a, b; unpack2<H>(nh[r], a, b); c2[r].x = a; c2[r].y = b; }
                        const int xn = x + 1 < TW ? x + 1 : x;
#pragma unroll
                        for (int r = 0; r < 3; ++r) nh[r] = *reinterpret_cast<const unsigned*>(gb + (r * HW2 + xn + 2) * PG);
                        tc_f32x2 dv = c1[1];
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky) {            // gd at (row 2 - ky, column 2 - kx) of the window meets tap (ky, kx)
                            dv += tp[ky * 3] * c2[2 - ky]; dv += tp[ky * 3 + 1] * c1[2 - ky]; dv += tp[ky * 3 + 2] * c0[2 - ky];
                        }
                        if (!(rowin && ow0 + x < Wimg)) dv = tc_f32x2{0.f, 0.f};
                        aw[c][10] += dv;
                        *reinterpret_cast<unsigned*>(dhp + x * PG) = pack2<H>(dv.x, dv.y);
                    }
                }
            }
            if (c + 1 == NCH && more) { xfetch(tidx + gridDim.x); gfetch(tidx + gridDim.x, 0); }     // (after the register-hungry stage)
            BSTAMP(7);
            __syncthreads();
            BSTAMP(8);
            // ---- dx^T += W1^T dh^T  (rows = input channel, columns = inner pixel);  dW1 += dh^T x  (rows = hidden channel, columns = input channel)
            if (cb < p.MT2) {
                const int q = cb * 32 + l31;
                const bf16_t* dp = hs + tc_mul24(q < IP ? hrow(q) : 0, PG) + 8 * hh;
#pragma unroll
                for (int kk = 0; kk < KKC; ++kk) {
                    const bf16_t* wp = w1s + (c * CH + 16 * kk + 2 * hh + 4 * (gi >> 2)) * PX + nb * 32 + 16 * gq2 + 4 * (gi & 3);
                    accx = TcHalf<H>::mfma(ld_tr<V8>(wp, wp + PX), *reinterpret_cast<const V8*>(dp + kk * 16), accx);
                }
            }
#pragma unroll
            for (int ks = 0; ks < K::IPMAX / 16; ++ks) {                // (steps beyond the tile's pixels multiply the zero row)
                const int qlo = 16 * ks + 8 * hh + (gi >> 2), qhi = qlo + 4, cc = 16 * gq2 + 4 * (gi & 3);
                const int rlo = qlo < IP ? hrow(qlo) : -1, rhi = qhi < IP ? hrow(qhi) : -1;
                const V8 av = ld_tr<V8>(hs + tc_mul24(rlo < 0 ? MPMAX : rlo, PG) + cb * 32 + cc, hs + tc_mul24(rhi < 0 ? MPMAX : rhi, PG) + cb * 32 + cc);
                const V8 bv = ld_tr<V8>(xs + tc_mul24(rlo < 0 ? 0 : rlo, PX) + nb * 32 + cc, xs + tc_mul24(rhi < 0 ? 0 : rhi, PX) + nb * 32 + cc);
                accw[c] = TcHalf<H>::mfma(av, bv, accw[c]);
            }
            BSTAMP(9);
            __syncthreads();                                        // the tiles of this chunk are dead
            BSTAMP(10);
        }
        // ---- dx leaves as whole pixel rows (through LDS, over the gd tile)
        constexpr int NEP = (K::IPMAX * XC + NTH - 1) / NTH;          // trips of the leaving loop
        uint4 xraw[NEP];
        if constexpr (PRE) {                                         // the raw rows the LayerNorm backward needs: in flight under the staging
            const H* xb = X + (long long)b * imgpix * p.ldx;
#pragma unroll
            for (int k = 0; k < NEP; ++k) {
                const int s = k * NTH + tid, q = s < IP * XC ? s / XC : 0, y = sdiv(q, dTW), x = smod(q, y, dTW);
                const bool valid = s < IP * XC && oh0 + y < Himg && ow0 + x < Wimg;
                xraw[k] = *reinterpret_cast<const uint4*>(xb + (valid ? (long long)(oh0 + y) * Wimg + ow0 + x : 0) * p.ldx + (tid % XC) * 8);
            }
        }
        if (cb < p.MT2) {
            float* sp = stg + tc_mul24(cb * 32 + l31, PO) + nb * 32 + 4 * hh;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *reinterpret_cast<float4*>(sp + 8 * gq) = make_float4(accx[4 * gq], accx[4 * gq + 1], accx[4 * gq + 2], accx[4 * gq + 3]);
        }
        __syncthreads();
        if constexpr (!PRE) {
            H* dxb = DX + (long long)b * imgpix * p.lddx;
            for (int s = tid; s < IP * XC; s += NTH) {
                const int q = s / XC, cg = s - q * XC, y = sdiv(q, dTW), x = smod(q, y, dTW);
                if (oh0 + y >= Himg || ow0 + x >= Wimg) continue;
                const float* sq = stg + tc_mul24(q, PO) + cg * 8;
                const float4 v0 = *reinterpret_cast<const float4*>(sq), v1 = *reinterpret_cast<const float4*>(sq + 4);
                float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                H* dst = dxb + (long long)((oh0 + y) * Wimg + ow0 + x) * p.lddx + cg * 8;
                if (p.acc_dx) {
                    float o[8];
                    up8<H>(*reinterpret_cast<const uint4*>(dst), o);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += o[e];
                }
                *reinterpret_cast<uint4*>(dst) = pk8<H>(v);
            }
        } else {
            H* dxb = DX + (long long)b * imgpix * p.lddx;
            const int cg = tid % XC;
            float dgl[8], dbl[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { dgl[e] = 0.f; dbl[e] = 0.f; }
#pragma unroll
            for (int k = 0; k < NEP; ++k) {                         // (whole pixels per lane group, every lane makes every trip)
                const int s = k * NTH + tid, q = s < IP * XC ? s / XC : 0, y = sdiv(q, dTW), x = smod(q, y, dTW);
                const bool valid = s < IP * XC && oh0 + y < Himg && ow0 + x < Wimg;
                const long long pix = valid ? (long long)(oh0 + y) * Wimg + ow0 + x : 0;
                const float* sq = stg + tc_mul24(q, PO) + cg * 8;
                const float4 v0 = *reinterpret_cast<const float4*>(sq), v1 = *reinterpret_cast<const float4*>(sq + 4);
                float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                if constexpr (PRE) {
                    // v = d(n2); the gradient of x = LayerNorm backward of it, from the raw row (still in L2) -- statistics recomputed
                    float f[8];
                    up8<H>(xraw[k], f);
                    float sm = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) sm += f[e];
                    sm = tc_group_sum<XC>(sm);
                    const float mean = sm * (1.0f / C);
                    float q2 = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { f[e] -= mean; q2 += f[e] * f[e]; }
                    q2 = tc_group_sum<XC>(q2);
                    const float rstd = rsqrtf(q2 * (1.0f / C) + p.pre_eps);
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        f[e] *= rstd;                               // xhat
                        if (valid) { dgl[e] += v[e] * f[e]; dbl[e] += v[e]; }
                        v[e] *= pre[cg * 8 + e];
                        s1 += v[e]; s2 += v[e] * f[e];
                    }
                    s1 = tc_group_sum<XC>(s1); s2 = tc_group_sum<XC>(s2);
                    s1 *= 1.0f / C; s2 *= 1.0f / C;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = rstd * (v[e] - s1 - f[e] * s2);
                }
                if (valid) {
                    H* dst = dxb + pix * p.lddx + cg * 8;
                    if (p.acc_dx) {
                        float o[8];
                        up8<H>(*reinterpret_cast<const uint4*>(dst), o);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += o[e];
                    }
                    *reinterpret_cast<uint4*>(dst) = pk8<H>(v);
                }
            }
            if constexpr (PRE) {                                    // lanes of one channel group (stride XC) fold, then one LDS add per wave and channel
#pragma unroll
                for (int e = 0; e < 8; ++e)
#pragma unroll
                    for (int m = XC; m < 64; m <<= 1) { dgl[e] += __shfl_xor(dgl[e], m, 64); dbl[e] += __shfl_xor(dbl[e], m, 64); }
                if (lane < XC) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { atomicAdd(pre + 2 * C + cg * 8 + e, dgl[e]); atomicAdd(pre + 3 * C + cg * 8 + e, dbl[e]); }
                }
            }
        }
        __syncthreads();
        BSTAMP(11);
    }
    // ---- this workgroup's partial sums
    float* PB = p.part + ((long long)g * gridDim.x + blockIdx.x) * PT::n;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = c * CH + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            PB[PT::oW1 + ch * C + nb * 32 + l31] = accw[c][r];
        }
    {   // depthwise sums: the eight waves (rows) of the workgroup fold through LDS (the tiles are dead)
        float* red = reinterpret_cast<float*>(smem);               // [NW][NCH * 22][64]
        __syncthreads();
        if constexpr (PRE) { if (tid < 2 * C) PB[PT::oPG + tid] = pre[2 * C + tid]; }
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int t = 0; t < 11; ++t) {
                red[(wave * (NCH * 22) + c * 22 + 2 * t) * 64 + lane] = aw[c][t].x;
                red[(wave * (NCH * 22) + c * 22 + 2 * t + 1) * 64 + lane] = aw[c][t].y;
            }
        __syncthreads();
        for (int i = tid; i < NCH * 22 * 64; i += NTH) {
            const int v = i >> 6, l = i & 63, c = v / 22, t = (v - c * 22) >> 1, ch = c * CH + 2 * l + (v & 1);
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < K::NW; ++w) sum += red[(w * (NCH * 22) + v) * 64 + l];
            if (t < 9) PB[PT::oWd + ch * 9 + t] = sum;
            else if (t == 9) PB[PT::oBd + ch] = sum;
            else PB[PT::oB1 + ch] = sum;
        }
    }
    BSTAMP(12);
}

// --------------------------------------------------------------------------------------------------------------------- launch 3
struct RedDev { const float* part; float* dst[10]; int off[11]; int nwg, nper; long long wstride; };

// grid (nper / 64, groups), 256 threads = 16 float4 columns x 16 slices of the workgroup list
__global__ __launch_bounds__(256) void ffn_bwd_reduce_kernel(const RedDev p) {
    __shared__ float4 sh[16][17];
    const int e = threadIdx.x & 15, sl = threadIdx.x >> 4, g = blockIdx.y;
    const int q = blockIdx.x * 16 + e;                              // float4 index within the partial
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q * 4 < p.nper) {
        const float* base = p.part + (long long)g * p.nwg * p.nper + (long long)q * 4;
        for (int w = sl; w < p.nwg; w += 16) {
            const float4 v = *reinterpret_cast<const float4*>(base + (long long)w * p.nper);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    sh[sl][e] = s;
    __syncthreads();
    if (sl == 0 && q * 4 < p.nper) {
#pragma unroll
        for (int k = 1; k < 16; ++k) { const float4 v = sh[k][e]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        const int idx = q * 4;
        int seg = 0;
#pragma unroll
        for (int k = 1; k < 10; ++k) seg += idx >= p.off[k];
        float* d = p.dst[seg];
        if (d) {
            d += (long long)g * p.wstride + (idx - p.off[seg]);
            d[0] += s.x; d[1] += s.y; d[2] += s.z; d[3] += s.w;
        }
    }
}

int bwd_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    }
    return n;
}

// pixel tile of launch 2: smallest estimated makespan over the CUs (cycles per tile = fixed + haloed pixels of fc1 / fills + inner pixels)
template <int C>
void dw_pick_tile(int H, int W, long long images, int gxmax, int fth, int ftw, int& TH, int& TW) {
    using K = DwCfg<C>;
    static const int eth = getenv("TC_FFN_BWD_TH") ? atoi(getenv("TC_FFN_BWD_TH")) : 0, etw = getenv("TC_FFN_BWD_TW") ? atoi(getenv("TC_FFN_BWD_TW")) : 0;
    if (!(fth && ftw)) { fth = eth; ftw = etw; }
    double best = 1e30;
    TH = 1; TW = 1;
    for (int th = 1; th <= K::THMAX && th <= H; ++th)
        for (int tw = 1; tw <= K::TWMAX && tw <= W; ++tw) {
            const int hp = (th + 2) * (tw + 2), ip = th * tw;
            if (hp > K::MPMAX || ip > K::IPMAX) continue;
            if (fth && ftw && (th != fth || tw != ftw)) continue;
            const double tiles = (double)images * ((H + th - 1) / th) * ((W + tw - 1) / tw);
            const double rounds = (double)(long long)((tiles + gxmax - 1) / gxmax);
            const double cyc = 6000.0 + ((hp + 31) / 32) * 32 * 40.0 + tw * 450.0 + ((ip + 31) / 32) * 32 * 12.0;
            const double cost = rounds * cyc;
            if (cost < best) { best = cost; TH = th; TW = tw; }
        }
}

bool bwd_args_ok(const TcFfnBwd* f) {
    if (!f || !f->x || !f->dy || !f->d || !f->stat || !f->w1 || !f->b1 || !f->wd || !f->gamma || !f->beta || !f->w2 || !f->dx || !f->gd || !f->part) return false;
    if (!f->dw1 || !f->db1 || !f->dwd || !f->dbd || !f->dgamma || !f->dbeta || !f->dw2 || !f->db2) return false;
    if (f->B < 1 || f->H < 1 || f->W < 1 || f->groups < 1 || f->C != 64) return false;
    auto al = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    if (!al(f->x) || !al(f->dy) || !al(f->d) || !al(f->dx) || !al(f->gd) || !al(f->w1) || !al(f->part) || ((uintptr_t)f->stat & 7)) return false;
    if ((f->ldx & 7) || (f->lddy & 7) || (f->lddx & 7) || (f->sdy & 7) || (f->wstride & 7)) return false;
    const long long mx = f->ldx > f->lddx ? f->ldx : f->lddx;
    return (long long)f->H * f->W * (mx > 4 * f->C ? mx : 4 * f->C) < 0x7fffffffLL && (long long)f->B * f->H * f->W < 0x7fffffffLL;
}

template <typename H, int C, bool PRE>
int ffn_bwd_launch(const TcFfnBwd* f, hipStream_t s) {
    using PT = BwdPart<C>;
    using KL = LnCfg<C>;
    using KD = DwCfg<C>;
    const int ncu = bwd_num_cus();
    int gx = ncu / f->groups;                                       // persistent workgroups: never more than the CUs hold at once
    if (gx < 1) gx = 1;
    if ((long long)f->groups * gx * PT::n > f->part_floats) return TC_ERR_ARG;
    FfnBwdDev p;
    p.x = f->x; p.dy = f->dy; p.d = f->d; p.stat = f->stat; p.w1 = f->w1; p.b1 = f->b1; p.wd = f->wd; p.gamma = f->gamma; p.beta = f->beta; p.w2 = f->w2;
    p.dx = f->dx; p.gd = f->gd; p.part = f->part; p.sdy = f->sdy; p.wstride = f->wstride; p.ldx = f->ldx; p.lddy = f->lddy; p.lddx = f->lddx;
    p.B = f->B; p.H = f->H; p.W = f->W; p.M = f->B * f->H * f->W; p.acc_dx = f->acc_dx;
    p.pre_g = f->pre_gamma; p.pre_b = f->pre_beta; p.pre_eps = f->pre_eps;
    p.nblk = (p.M + KL::P - 1) / KL::P;
    dw_pick_tile<C>(f->H, f->W, f->B, gx, f->tile_h, f->tile_w, p.TH, p.TW);
    if ((p.TH + 2) * (p.TW + 2) > KD::MPMAX || p.TH * p.TW > KD::IPMAX || p.TH > KD::THMAX || p.TW > KD::TWMAX) return TC_ERR_ARG;
    p.tilesH = (f->H + p.TH - 1) / p.TH; p.tilesW = (f->W + p.TW - 1) / p.TW;
    p.HW2 = p.TW + 2; p.HP = (p.TH + 2) * p.HW2; p.MT = (p.HP + 31) / 32; p.IP = p.TH * p.TW; p.MT2 = (p.IP + 31) / 32; p.KS = (p.IP + 15) / 16;
    // tile / pixel indices in fp32 (sdiv: < 2^22), per-image element offsets in 32 bits (< 2^31)
    if ((long long)f->B * p.tilesH * p.tilesW >= (1LL << 22) || (long long)f->H * f->W >= (1LL << 23) || f->ldx >= (1 << 23) ||
        (long long)f->H * f->W * (f->ldx > 4 * C ? f->ldx : 4 * C) >= (1LL << 31)) return TC_ERR_ARG;
    p.ntiles = f->B * p.tilesH * p.tilesW;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)ffn_bwd_ln_kernel<H, C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KL::smem) != hipSuccess) return TC_ERR_LAUNCH;
        if (hipFuncSetAttribute((const void*)ffn_bwd_dw_kernel<H, C, PRE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KD::smem) != hipSuccess) return TC_ERR_LAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL((ffn_bwd_ln_kernel<H, C>), dim3(gx, f->groups), dim3(KL::NTH), KL::smem, s, p);
    if (tc_launch_status() != TC_OK) return TC_ERR_LAUNCH;
    hipLaunchKernelGGL((ffn_bwd_dw_kernel<H, C, PRE>), dim3(gx, f->groups), dim3(KD::NTH), KD::smem, s, p);
    if (tc_launch_status() != TC_OK) return TC_ERR_LAUNCH;
    RedDev r;
    r.part = f->part; r.nwg = gx; r.nper = PT::n; r.wstride = f->wstride;
    float* dsts[10] = {f->dw2, f->db2, f->dgamma, f->dbeta, f->dw1, f->db1, f->dwd, f->dbd, PRE ? f->dpre_gamma : nullptr, PRE ? f->dpre_beta : nullptr};
    const int offs[11] = {PT::oW2, PT::oB2, PT::oG, PT::oBt, PT::oW1, PT::oB1, PT::oWd, PT::oBd, PT::oPG, PT::oPB, PT::n};
    for (int i = 0; i < 10; ++i) r.dst[i] = dsts[i];
    for (int i = 0; i < 11; ++i) r.off[i] = offs[i];
    hipLaunchKernelGGL(ffn_bwd_reduce_kernel, dim3((PT::n / 4 + 15) / 16, f->groups), dim3(256), 0, s, r);
    return tc_launch_status();
}

}  // namespace

#ifdef TC_FFNB_TIMING
extern "C" int tc_ffnb_dbg_read(long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ffnb_dbg), sizeof(long long) * 2 * 16 * 16); }
#endif

extern "C" int tc_ffn_fused_bwd_supported(int C, int dtype) { return C == 64 && (dtype == TC_BF16 || dtype == TC_F16); }

extern "C" long long tc_ffn_fused_bwd_scratch_floats(int C, int groups) {
    if (C != 64 || groups < 1) return 0;
    const int ncu = bwd_num_cus();
    const int gx = ncu / groups < 1 ? 1 : ncu / groups;
    return (long long)groups * gx * BwdPart<64>::n;
}

extern "C" int tc_ffn_fused_bwd(const TcFfnBwd* f, int dtype, void* stream) {
    if (!bwd_args_ok(f) || !tc_ffn_fused_bwd_supported(f->C, dtype)) return TC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const bool pre = f->pre_gamma != nullptr;
    if (pre && (!f->pre_beta || !f->dpre_gamma || !f->dpre_beta)) return TC_ERR_ARG;
    if (dtype == TC_BF16) return pre ? ffn_bwd_launch<bf16_t, 64, true>(f, s) : ffn_bwd_launch<bf16_t, 64, false>(f, s);
    return pre ? ffn_bwd_launch<f16_t, 64, true>(f, s) : ffn_bwd_launch<f16_t, 64, false>(f, s);
}
