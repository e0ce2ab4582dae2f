// MixFFN_skip (MSTr.py:889-902 with DWConv :21-31) as spatially tiled kernels whose hidden maps live in LDS:
//   out = fc2(GELU(LayerNorm_4C(dw3x3(h) + h))) + residual,  h = fc1(x)
// The op-by-op form writes and re-reads three 4C-wide maps per site in the forward pass alone (h, d, a); at C = 64 / 128 they are
// most of the step's memory traffic (profiles/r2_hbm_by_kernel.json).  Here a workgroup owns a TH x TW pixel tile:
//   forward   x halo tile -> fc1 (MFMA) -> h in LDS -> dw3x3 + bias + skip in place -> LayerNorm statistics -> GELU(LN(.)) in place
//             -> fc2 (MFMA) -> + bias + residual -> out.   Only d (what the backward recomputes from) and the row statistics leave.
// 16-bit storage types only (bf16 / fp16, v_mfma_f32_32x32x16); the fp32 parity path keeps the op-by-op kernels.
// Work split inside the 512-thread workgroup: wave w owns the hidden channels [w * 4C/8, (w+1) * 4C/8) through fc1 and the
// depthwise stage (no barrier between them: a wave only touches its own LDS columns), and one 32-pixel x 32-channel block of fc2.
#include "tc_common.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct FfnFwdDev {
    const void* x; const void* w1; const void* b1; const void* wd; const void* bd; const void* gamma; const void* beta;
    const void* w2; const void* b2; const void* res; void* out; void* h; void* d; void* a; float* stat;
    long long sres, sout, wstride;
    int ldx, ldr, ldo;
    int B, H, W;
    int TH, TW, tilesH, tilesW, HW2, HP, MT, IP, MT2, ntiles;
    float eps;
    const void* pre_g; const void* pre_b; float pre_eps;           // LayerNorm(C) applied to x on its way into LDS (the block's norm2), or null
};

template <int C> struct FfnCfg {
    static constexpr int C4 = 4 * C, NW = 8, NTH = 512, CW = C4 / NW, NT1 = CW / 32, KK1 = C / 16, KK2 = C4 / 16, NT2 = C / 32, SG = CW / 8;
    static constexpr int PX = C + 8, PH = C4 + 8, PO = C + 4;                        // LDS row pitches: x tile, hidden tile (elements), fp32 out stage
    static constexpr int MPMAX = C == 64 ? 160 : 96, IPMAX = C == 64 ? 128 : 64;    // halo / inner pixels of a tile
    static constexpr int MT2MAX = IPMAX / 32;
    static constexpr int NXR = (MPMAX * (C / 8) + NTH - 1) / NTH;                   // 16-byte x strips per thread
    static constexpr bool W2LDS = C == 64;                                           // fc2's weights resident in LDS (33 KB at C = 64; 133 KB at C = 128: streamed from L2)
    static constexpr size_t smem = (size_t)MPMAX * PH * 2 + (size_t)MPMAX * PX * 2 + (size_t)10 * C4 * 4 + (size_t)3 * C4 * 4 +
                                   (size_t)NW * IPMAX * 8 + (size_t)IPMAX * 8 + (W2LDS ? (size_t)C * PH * 2 : 0);
    static_assert(NW == MT2MAX * NT2, "one fc2 block per wave");
    static_assert((size_t)IPMAX * PO * 4 <= (size_t)MPMAX * PH * 2, "the fp32 output stage aliases the hidden tile");
    static_assert(smem <= 160 * 1024, "LDS");
};

template <typename H> __device__ __forceinline__ void up8(const uint4& r, float* o) {
    unpack2<H>(r.x, o[0], o[1]); unpack2<H>(r.y, o[2], o[3]); unpack2<H>(r.z, o[4], o[5]); unpack2<H>(r.w, o[6], o[7]);
}
template <typename H> __device__ __forceinline__ uint4 pk8(const float* o) {
    return make_uint4(pack2<H>(o[0], o[1]), pack2<H>(o[2], o[3]), pack2<H>(o[4], o[5]), pack2<H>(o[6], o[7]));
}

#ifdef TC_FFNF_TIMING
// phase stamps (experiment builds only, scripts/exp/ffnb_timing.py --fwd): thread 0 of the first 16 workgroups of weight group 0
__device__ long long g_ffnf_dbg[16 * 16];
#define FSTAMP(k) do { if (fs_) { const long long t_ = __builtin_readcyclecounter(); fs_[k] += t_ - ft_; ft_ = t_; } } while (0)
#define FSTAMP_INIT() long long ft_ = __builtin_readcyclecounter(); \
    long long* fs_ = (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 16) ? g_ffnf_dbg + blockIdx.x * 16 : nullptr; \
    if (fs_) for (int k_ = 0; k_ < 16; ++k_) fs_[k_] = 0;
#else
#define FSTAMP(k)
#define FSTAMP_INIT()
#endif

template <typename H, int C, bool PRE>
__global__ __launch_bounds__(512, 2) void ffn_fused_fwd_kernel(const FfnFwdDev p) {
    using K = FfnCfg<C>;
    using V8 = typename TcHalf<H>::v8;
    constexpr int C4 = K::C4, CW = K::CW, NT1 = K::NT1, KK1 = K::KK1, KK2 = K::KK2, SG = K::SG, PX = K::PX, PH = K::PH, PO = K::PO;
    constexpr int NTH = K::NTH, IPMAX = K::IPMAX, XC = C / 8, HC = C4 / 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* hs = reinterpret_cast<bf16_t*>(smem);                          // [MPMAX][PH]   h, then d, then a (storage type)
    bf16_t* xs = hs + K::MPMAX * PH;                                       // [MPMAX][PX]   x on the haloed tile
    float* wtap = reinterpret_cast<float*>(xs + K::MPMAX * PX);            // [9][C4] taps, [C4] conv bias
    float* gbs = wtap + 10 * C4;                                           // gamma[C4], beta[C4], fc1 bias[C4]
    float2* pst = reinterpret_cast<float2*>(gbs + 3 * C4);                 // [8][IPMAX] per-wave (sum, squared deviations)
    float2* fst = pst + 8 * IPMAX;                                         // [IPMAX] mean, rstd
    bf16_t* w2s = reinterpret_cast<bf16_t*>(fst + IPMAX);                  // [C][PH] fc2 weights (W2LDS)
    float* stg = reinterpret_cast<float*>(smem);                           // fp32 out stage over the (then dead) hidden tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int g = blockIdx.y;
    FSTAMP_INIT();
    const long long wo = (long long)g * p.wstride, imgpix = (long long)p.H * p.W, grow = (long long)g * p.B * imgpix;
    const H* X = reinterpret_cast<const H*>(p.x) + grow * p.ldx;
    const H* W1 = reinterpret_cast<const H*>(p.w1) + wo;
    const H* W2 = reinterpret_cast<const H*>(p.w2) + wo;
    const H* RES = p.res ? reinterpret_cast<const H*>(p.res) + (long long)g * p.sres : nullptr;
    H* OUT = reinterpret_cast<H*>(p.out) + (long long)g * p.sout;
    H* HO = p.h ? reinterpret_cast<H*>(p.h) + grow * C4 : nullptr;
    H* DO = p.d ? reinterpret_cast<H*>(p.d) + grow * C4 : nullptr;
    H* AO = p.a ? reinterpret_cast<H*>(p.a) + grow * C4 : nullptr;
    float2* ST = p.stat ? reinterpret_cast<float2*>(p.stat) + grow : nullptr;
    const int TH = p.TH, TW = p.TW, HW2 = p.HW2, HP = p.HP, MT = p.MT, IP = p.IP, Himg = p.H, Wimg = p.W;
    const SDiv dHW2 = sdiv_make(HW2), dTW = sdiv_make(TW), dRW = sdiv_make((TW + 1) >> 1), dTLW = sdiv_make(p.tilesW), dTLH = sdiv_make(p.tilesH);   // (tc_common.h: run-time divisors)

    // Parameters of this weight group.  Every global load of the prologue is issued before the first LDS store waits for one (the
    // staging loops used to pay one memory round trip per iteration: 10 - 17 k cycles of a 25 - 80 k-cycle launch).
    constexpr int NTAP = (9 * C4 + NTH - 1) / NTH, NPAR = (C4 + NTH - 1) / NTH, NW2 = K::W2LDS ? (C * HC + NTH - 1) / NTH : 1;
    float tv[NTAP], pv[NPAR][4];
    uint4 w2v[NW2];
    {
        const H* wd = reinterpret_cast<const H*>(p.wd) + wo;
        const H* bd = reinterpret_cast<const H*>(p.bd) + wo;
        const H* gm = reinterpret_cast<const H*>(p.gamma) + wo;
        const H* bt = reinterpret_cast<const H*>(p.beta) + wo;
        const H* b1 = reinterpret_cast<const H*>(p.b1) + wo;
#pragma unroll
        for (int k = 0; k < NTAP; ++k) { const int i = tid + k * NTH; tv[k] = ldf<H>(wd + (i < 9 * C4 ? i : 0)); }
#pragma unroll
        for (int k = 0; k < NPAR; ++k) {
            const int i = min(tid + k * NTH, C4 - 1);
            pv[k][0] = ldf<H>(bd + i); pv[k][1] = ldf<H>(gm + i); pv[k][2] = ldf<H>(bt + i); pv[k][3] = ldf<H>(b1 + i);
        }
        if constexpr (K::W2LDS) {
#pragma unroll
            for (int k = 0; k < NW2; ++k) {
                const int i = min(tid + k * NTH, C * HC - 1), o = i / HC, cg = i - o * HC;
                w2v[k] = *reinterpret_cast<const uint4*>(W2 + (long long)o * C4 + cg * 8);
            }
        }
    }
    // fc1 weights of this wave's channels: MFMA operand fragments, resident for the whole launch
    V8 wf1[NT1][KK1];
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
        for (int kk = 0; kk < KK1; ++kk)
            wf1[nt][kk] = *reinterpret_cast<const V8*>(W1 + (long long)(wave * CW + nt * 32 + l31) * C + kk * 16 + 8 * hh);
    float b2v[8];
    {
        const H* b2 = reinterpret_cast<const H*>(p.b2) + wo;
#pragma unroll
        for (int e = 0; e < 8; ++e) b2v[e] = ldf<H>(b2 + (tid % XC) * 8 + e);
    }
    // fc2: wave -> (pixel block mt2, output-channel block nt2)
    const int mt2 = wave % K::MT2MAX, nt2 = wave / K::MT2MAX;

    auto tile_org = [&](int tidx, int& b, int& oh0, int& ow0) __attribute__((always_inline)) {
        const int t1 = sdiv(tidx, dTLW), tx = smod(tidx, t1, dTLW);
        b = sdiv(t1, dTLH);
        const int ty = smod(t1, b, dTLH);
        b = __builtin_amdgcn_readfirstlane(b); oh0 = __builtin_amdgcn_readfirstlane(tc_mul24(ty, TH)); ow0 = __builtin_amdgcn_readfirstlane(tc_mul24(tx, TW));
    };
    // x on the haloed tile: loads issued branch-free (positions outside read the map's first bytes), zeroed on the way into LDS
    uint4 xr[K::NXR];
    auto xin = [&](int s, int oh0, int ow0, int& pix, int& cg, int& off) __attribute__((always_inline)) {
        pix = s / XC; cg = s - pix * XC;
        const int hy = sdiv(pix, dHW2), hx = smod(pix, hy, dHW2), ih = oh0 - 1 + hy, iw = ow0 - 1 + hx;
        const bool ok = pix < HP && (unsigned)ih < (unsigned)Himg && (unsigned)iw < (unsigned)Wimg;
        off = tc_mad24(tc_mad24(ih, Wimg, iw), p.ldx, cg * 8);
        return ok;
    };
    auto xfetch = [&](int tidx) __attribute__((always_inline)) {
        int b, oh0, ow0;
        tile_org(tidx, b, oh0, ow0);
        const H* xb = X + (long long)b * imgpix * p.ldx;
#pragma unroll
        for (int i = 0; i < K::NXR; ++i) {
            int pix, cg, off;
            const bool ok = xin(tid + i * NTH, oh0, ow0, pix, cg, off);
            xr[i] = *reinterpret_cast<const uint4*>(xb + (ok ? off : 0));
        }
    };
    auto xput = [&](int tidx) __attribute__((always_inline)) {
        int b, oh0, ow0;
        tile_org(tidx, b, oh0, ow0);
#pragma unroll
        for (int i = 0; i < K::NXR; ++i) {
            int pix, cg, off;
            const int s = tid + i * NTH;
            const bool ok = xin(s, oh0, ow0, pix, cg, off);
            uint4 v = ok ? xr[i] : make_uint4(0u, 0u, 0u, 0u);
            if constexpr (PRE) {                                   // LayerNorm over the pixel's C channels: XC consecutive lanes hold one pixel
                float f[8], gm[8], bt[8];
                up8<H>(v, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) {                      // (parameter vectors sit at any 2-byte offset of the flat buffer)
                    gm[e] = ldf<H>(reinterpret_cast<const H*>(p.pre_g) + wo + cg * 8 + e);
                    bt[e] = ldf<H>(reinterpret_cast<const H*>(p.pre_b) + wo + cg * 8 + e);
                }
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
                for (int e = 0; e < 8; ++e) f[e] = f[e] * rstd * gm[e] + bt[e];
                v = pk8<H>(f);
            }
            if (s < MT * 32 * XC) *reinterpret_cast<uint4*>(xs + tc_mul24(pix, PX) + cg * 8) = v;
        }
    };

    int tidx = blockIdx.x;
    if (tidx < p.ntiles) xfetch(tidx);                           // (in flight with the parameter loads)
    {   // taps transposed to [tap][channel], fp32; conv bias, gamma, beta, fc1 bias; fc2 weights (W2LDS)
#pragma unroll
        for (int k = 0; k < NTAP; ++k) { const int i = tid + k * NTH, ch = i / 9, t = i - ch * 9; if (i < 9 * C4) wtap[t * C4 + ch] = tv[k]; }
#pragma unroll
        for (int k = 0; k < NPAR; ++k) {
            const int i = tid + k * NTH;
            if (i < C4) { wtap[9 * C4 + i] = pv[k][0]; gbs[i] = pv[k][1]; gbs[C4 + i] = pv[k][2]; gbs[2 * C4 + i] = pv[k][3]; }
        }
        if constexpr (K::W2LDS) {
#pragma unroll
            for (int k = 0; k < NW2; ++k) {
                const int i = tid + k * NTH, o = i / HC, cg = i - o * HC;
                if (i < C * HC) *reinterpret_cast<uint4*>(w2s + o * PH + cg * 8) = w2v[k];
            }
        }
    }
    if (tidx < p.ntiles) xput(tidx);
    __syncthreads();                                             // parameters and the first x tile are in LDS
    FSTAMP(0);
    for (; tidx < p.ntiles; tidx += gridDim.x) {
        int b, oh0, ow0;
        tile_org(tidx, b, oh0, ow0);
        const long long ibase = (long long)b * imgpix;
        const bool more = tidx + (int)gridDim.x < p.ntiles;
        if (more) xfetch(tidx + gridDim.x);                      // lands under this tile's arithmetic
        // ---- fc1 on the haloed tile, this wave's CW hidden channels; rows outside the image are the convolution's zero padding
        for (int mi = 0; mi < MT; ++mi) {
            f32x16 acc[NT1];
#pragma unroll
            for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {                 // the accumulator starts at the bias of the lane's columns
                    const float4 bv = *reinterpret_cast<const float4*>(gbs + 2 * C4 + wave * CW + nt * 32 + 8 * gq + 4 * hh);
                    acc[nt][4 * gq] = bv.x; acc[nt][4 * gq + 1] = bv.y; acc[nt][4 * gq + 2] = bv.z; acc[nt][4 * gq + 3] = bv.w;
                }
            const bf16_t* ap = xs + tc_mul24(mi * 32 + l31, PX) + 8 * hh;
#pragma unroll
            for (int kk = 0; kk < KK1; ++kk) {
                const V8 av = *reinterpret_cast<const V8*>(ap + kk * 16);
#pragma unroll
                for (int nt = 0; nt < NT1; ++nt) acc[nt] = TcHalf<H>::mfma(wf1[nt][kk], av, acc[nt]);     // D^T: lane = pixel row
            }
            const int pp = mi * 32 + l31, hy = sdiv(pp, dHW2), hx = smod(pp, hy, dHW2), ih = oh0 - 1 + hy, iw = ow0 - 1 + hx;
            const bool ok = pp < HP && (unsigned)ih < (unsigned)Himg && (unsigned)iw < (unsigned)Wimg;
            const bool inner = ok && hy >= 1 && hy <= TH && hx >= 1 && hx <= TW;
            const int ppH = tc_mul24(pp, PH);
#pragma unroll
            for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int col = wave * CW + nt * 32 + 8 * gq + 4 * hh;
                    uint2 v = make_uint2(pack2<H>(acc[nt][4 * gq], acc[nt][4 * gq + 1]), pack2<H>(acc[nt][4 * gq + 2], acc[nt][4 * gq + 3]));
                    if (!ok) v = make_uint2(0u, 0u);
                    *reinterpret_cast<uint2*>(hs + ppH + col) = v;
                    if (HO && inner) *reinterpret_cast<uint2*>(HO + (ibase + (long long)ih * Wimg + iw) * C4 + col) = v;
                }
        }
        FSTAMP(1);
        // ---- d = dw3x3(h) + bias + h on the inner pixels, in place: the result of pixel (y, x) goes to halo slot (y, x), which no
        // later pixel of the row-major walk reads (a wave's LDS accesses execute in order; every load of a round precedes its stores)
        {
            // item = (inner row y, pixel pair x0 / x0 + 1, 8-channel group) of this wave's channels, 64 items per round, all rounds in
            // registers at once: a filter row's taps are read from LDS once per tile (not once per item), and every load of the walk
            // precedes its first store
            constexpr int NRD = 2;                 // rounds held in registers at once (C = 128: two passes of two, or the fc1 fragments spill)
            const int sg = lane % SG, chw = wave * CW + sg * 8, RW = (TW + 1) >> 1, nitems_all = TH * RW * SG;
            for (int it00 = 0; it00 < nitems_all; it00 += NRD * 64) {
            const int nitems = min(nitems_all - it00, NRD * 64);
            int yq[NRD], xq[NRD];
            tc_f32x2 o[NRD][2][4];                               // packed fp32 pairs: two channels per issue slot
            auto up8p = [&](const uint4& r, tc_f32x2* v) __attribute__((always_inline)) {
                float a, b;
                unpack2<H>(r.x, a, b); v[0] = tc_f32x2{a, b}; unpack2<H>(r.y, a, b); v[1] = tc_f32x2{a, b};
                unpack2<H>(r.z, a, b); v[2] = tc_f32x2{a, b}; unpack2<H>(r.w, a, b); v[3] = tc_f32x2{a, b};
            };
            {
                const float4 ba = *reinterpret_cast<const float4*>(wtap + 9 * C4 + chw), bb = *reinterpret_cast<const float4*>(wtap + 9 * C4 + chw + 4);
#pragma unroll
                for (int q = 0; q < NRD; ++q) {
                    const int it = q * 64 + lane, run = it < nitems ? (it00 + it) / SG : 0;
                    yq[q] = sdiv(run, dRW); xq[q] = smod(run, yq[q], dRW) * 2;
#pragma unroll
                    for (int r = 0; r < 2; ++r) { o[q][r][0] = tc_f32x2{ba.x, ba.y}; o[q][r][1] = tc_f32x2{ba.z, ba.w}; o[q][r][2] = tc_f32x2{bb.x, bb.y}; o[q][r][3] = tc_f32x2{bb.z, bb.w}; }
                }
            }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                tc_f32x2 tw[3][4];
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float* tp = wtap + (dy * 3 + dx) * C4 + chw;
                    const float4 ta = *reinterpret_cast<const float4*>(tp), tb = *reinterpret_cast<const float4*>(tp + 4);
                    tw[dx][0] = tc_f32x2{ta.x, ta.y}; tw[dx][1] = tc_f32x2{ta.z, ta.w}; tw[dx][2] = tc_f32x2{tb.x, tb.y}; tw[dx][3] = tc_f32x2{tb.z, tb.w};
                }
#pragma unroll
                for (int q = 0; q < NRD; ++q) {
                    if (q * 64 >= nitems) break;
                    tc_f32x2 in[4][4];
#pragma unroll
                    for (int dx = 0; dx < 4; ++dx) up8p(*reinterpret_cast<const uint4*>(hs + tc_mul24(tc_mad24(yq[q] + dy, HW2, xq[q] + dx), PH) + chw), in[dx]);
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                        for (int r = 0; r < 2; ++r)
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[q][r][e] += tw[dx][e] * in[r + dx][e];
                    if (dy == 1) {
#pragma unroll
                        for (int r = 0; r < 2; ++r)
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[q][r][e] += in[r + 1][e];
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < NRD; ++q) {
                if (q * 64 >= nitems) break;
                const bool live = q * 64 + lane < nitems;
                const int y = yq[q], x0 = xq[q];
#pragma unroll
                for (int r = 0; r < 2; ++r) {                    // LayerNorm partials over this wave's CW channels (Chan-mergeable)
                    const tc_f32x2 s4 = (o[q][r][0] + o[q][r][1]) + (o[q][r][2] + o[q][r][3]);
                    float sm = s4.x + s4.y;
                    sm = tc_group_sum<SG>(sm);
                    const float mu = sm * (1.0f / (float)CW);
                    tc_f32x2 q2 = {0.f, 0.f};
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const tc_f32x2 dl = o[q][r][e] - mu; q2 += dl * dl; }
                    float sq = q2.x + q2.y;
                    sq = tc_group_sum<SG>(sq);
                    if (live && sg == 0 && x0 + r < TW) pst[wave * IPMAX + y * TW + x0 + r] = make_float2(sm, sq);
                }
#pragma unroll
                for (int r = 0; r < 2; ++r)
                    if (live && x0 + r < TW)
                        *reinterpret_cast<uint4*>(hs + tc_mul24(tc_mad24(y, HW2, x0 + r), PH) + chw) =
                            make_uint4(pack2<H>(o[q][r][0].x, o[q][r][0].y), pack2<H>(o[q][r][1].x, o[q][r][1].y), pack2<H>(o[q][r][2].x, o[q][r][2].y), pack2<H>(o[q][r][3].x, o[q][r][3].y));
            }
            }
        }
        FSTAMP(2);
        __syncthreads();
        FSTAMP(3);
        // ---- row statistics
        if (tid < IP) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += pst[w * IPMAX + tid].x;
            const float mean = s * (1.0f / (float)C4);
            float m2 = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) { const float2 t = pst[w * IPMAX + tid]; const float dm = t.x * (1.0f / (float)CW) - mean; m2 += t.y + (float)CW * dm * dm; }
            const float rstd = rsqrtf(m2 * (1.0f / (float)C4) + p.eps);
            fst[tid] = make_float2(mean, rstd);
            const int y = sdiv(tid, dTW), x = smod(tid, y, dTW);
            if (ST && oh0 + y < Himg && ow0 + x < Wimg) ST[ibase + (long long)(oh0 + y) * Wimg + ow0 + x] = make_float2(mean, rstd);
        }
        __syncthreads();
        FSTAMP(4);
        // ---- a = GELU(LN(d)) in place; d (and a, when asked for) leave for the backward pass as whole pixel rows
        {
            const int cgx = tid % HC;
            float gm[8], bt[8];
            {
                const float4 a0 = *reinterpret_cast<const float4*>(gbs + cgx * 8), a1 = *reinterpret_cast<const float4*>(gbs + cgx * 8 + 4);
                const float4 c0 = *reinterpret_cast<const float4*>(gbs + C4 + cgx * 8), c1 = *reinterpret_cast<const float4*>(gbs + C4 + cgx * 8 + 4);
                gm[0] = a0.x; gm[1] = a0.y; gm[2] = a0.z; gm[3] = a0.w; gm[4] = a1.x; gm[5] = a1.y; gm[6] = a1.z; gm[7] = a1.w;
                bt[0] = c0.x; bt[1] = c0.y; bt[2] = c0.z; bt[3] = c0.w; bt[4] = c1.x; bt[5] = c1.y; bt[6] = c1.z; bt[7] = c1.w;
            }
            for (int s = tid; s < IP * HC; s += NTH) {
                const int q = s / HC, y = sdiv(q, dTW), x = smod(q, y, dTW);
                bf16_t* cell = hs + tc_mul24(tc_mad24(y, HW2, x), PH) + cgx * 8;
                const uint4 dv = *reinterpret_cast<const uint4*>(cell);
                const float2 st = fst[q];
                float v[8];
                up8<H>(dv, v);
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const tc_f32x2 xv = {v[e], v[e + 1]}, gv = {gm[e], gm[e + 1]}, bv = {bt[e], bt[e + 1]};
                    const tc_f32x2 u = gelu_poly2((xv - st.x) * st.y * gv + bv);
                    v[e] = u.x; v[e + 1] = u.y;
                }
                const uint4 av = pk8<H>(v);
                *reinterpret_cast<uint4*>(cell) = av;
                if (oh0 + y < Himg && ow0 + x < Wimg) {
                    const long long ro = (ibase + (long long)(oh0 + y) * Wimg + ow0 + x) * C4 + cgx * 8;
                    if (DO) *reinterpret_cast<uint4*>(DO + ro) = dv;
                    if (AO) *reinterpret_cast<uint4*>(AO + ro) = av;
                }
            }
        }
        FSTAMP(5);
        __syncthreads();
        FSTAMP(6);
        // ---- fc2: this wave's 32 pixels x 32 output channels over K = 4C
        f32x16 oacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
        if (mt2 < p.MT2) {
            const int q = min(mt2 * 32 + l31, IP - 1), y = sdiv(q, dTW), x = smod(q, y, dTW);
            const bf16_t* ap = hs + tc_mul24(tc_mad24(y, HW2, x), PH) + 8 * hh;
            if constexpr (K::W2LDS) {
                const bf16_t* wp = w2s + (nt2 * 32 + l31) * PH + 8 * hh;
#pragma unroll
                for (int kk = 0; kk < KK2; ++kk)
                    oacc = TcHalf<H>::mfma(*reinterpret_cast<const V8*>(wp + kk * 16), *reinterpret_cast<const V8*>(ap + kk * 16), oacc);
            } else {
                // W2 fragments stream from L2 (shared by every workgroup), four k-steps ahead of their use
                const H* wp = W2 + (long long)(nt2 * 32 + l31) * C4 + 8 * hh;
                V8 bq[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) bq[j] = *reinterpret_cast<const V8*>(wp + j * 16);
                for (int kk = 0; kk < KK2; kk += 4) {
                    V8 nb[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) nb[j] = *reinterpret_cast<const V8*>(wp + min(kk + 4 + j, KK2 - 1) * 16);
#pragma unroll
                    for (int j = 0; j < 4; ++j) oacc = TcHalf<H>::mfma(bq[j], *reinterpret_cast<const V8*>(ap + (kk + j) * 16), oacc);
#pragma unroll
                    for (int j = 0; j < 4; ++j) bq[j] = nb[j];
                }
            }
        }
        FSTAMP(7);
        __syncthreads();                                         // every fragment read of the hidden tile is done: it becomes the out stage
        FSTAMP(8);
        if (mt2 < p.MT2) {
            float* sp = stg + tc_mul24(mt2 * 32 + l31, PO) + nt2 * 32 + 4 * hh;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *reinterpret_cast<float4*>(sp + 8 * gq) = make_float4(oacc[4 * gq], oacc[4 * gq + 1], oacc[4 * gq + 2], oacc[4 * gq + 3]);
        }
        __syncthreads();
        FSTAMP(9);
        {   // + bias + residual, rounded once, whole 16-byte pieces of pixel rows
            const int cg = tid % XC;
            for (int s = tid; s < IP * XC; s += NTH) {
                const int q = s / XC, y = sdiv(q, dTW), x = smod(q, y, dTW);
                if (oh0 + y >= Himg || ow0 + x >= Wimg) continue;
                const long long rg = ibase + (long long)(oh0 + y) * Wimg + ow0 + x;
                const float* sq = stg + tc_mul24(q, PO) + cg * 8;
                const float4 v0 = *reinterpret_cast<const float4*>(sq), v1 = *reinterpret_cast<const float4*>(sq + 4);
                float v[8] = {v0.x + b2v[0], v0.y + b2v[1], v0.z + b2v[2], v0.w + b2v[3], v1.x + b2v[4], v1.y + b2v[5], v1.z + b2v[6], v1.w + b2v[7]};
                if (RES) {
                    float rv[8];
                    up8<H>(*reinterpret_cast<const uint4*>(RES + rg * p.ldr + cg * 8), rv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += rv[e];
                }
                *reinterpret_cast<uint4*>(OUT + rg * p.ldo + cg * 8) = pk8<H>(v);
            }
        }
        FSTAMP(10);
        if (more) xput(tidx + gridDim.x);                        // (the x tile's last reader was this tile's fc1)
        __syncthreads();
        FSTAMP(11);
    }
}

int ffn_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    }
    return n;
}

// Tile shape: the (TH, TW) with the smallest estimated makespan over the CUs (workgroups are persistent, one per CU): cycles per
// tile = fixed (barriers, prologue) + halo rows of fc1 + inner pixels of the element-wise stages and fc2.
template <int C>
void ffn_pick_tile(int H, int W, long long images, int fth, int ftw, int& TH, int& TW) {
    using K = FfnCfg<C>;
    static const int eth = getenv("TC_FFN_TH") ? atoi(getenv("TC_FFN_TH")) : 0, etw = getenv("TC_FFN_TW") ? atoi(getenv("TC_FFN_TW")) : 0;
    if (!(fth && ftw)) { fth = eth; ftw = etw; }
    const double ncu = ffn_num_cus();
    double best = 1e30;
    TH = 1; TW = 2;
    for (int th = 1; th <= 16 && th <= H; ++th)
        for (int tw = 2; tw <= 32; ++tw) {
            if (tw > W && tw != ((W + 1) & ~1)) continue;
            if ((tw & 1) && tw != W && !(fth && ftw)) continue;
            const int hp = (th + 2) * (tw + 2), ip = th * tw;
            if (hp > K::MPMAX || ip > K::IPMAX || th * ((tw + 1) / 2) * K::SG > 256) continue;
            if (fth && ftw && (th != fth || tw != ftw)) continue;
            const double tiles = (double)images * ((H + th - 1) / th) * ((W + tw - 1) / tw);
            const double rounds = (double)(long long)((tiles + ncu - 1) / ncu);
            const double mt = (hp + 31) / 32, cyc = 2500.0 + mt * 32 * (C == 64 ? 14.0 : 50.0) + ip * (C == 64 ? 110.0 : 220.0);
            const double cost = rounds * cyc;
            if (cost < best) { best = cost; TH = th; TW = tw; }
        }
}

template <typename H, int C, bool PRE>
int ffn_fused_fwd_launch(const TcFfnFused* f, hipStream_t s) {
    using K = FfnCfg<C>;
    FfnFwdDev p;
    p.x = f->x; p.w1 = f->w1; p.b1 = f->b1; p.wd = f->wd; p.bd = f->bd; p.gamma = f->gamma; p.beta = f->beta; p.w2 = f->w2; p.b2 = f->b2;
    p.res = f->res; p.out = f->out; p.h = f->h; p.d = f->d; p.a = f->a; p.stat = f->stat;
    p.sres = f->sres; p.sout = f->sout; p.wstride = f->wstride; p.ldx = f->ldx; p.ldr = f->ldr; p.ldo = f->ldo;
    p.B = f->B; p.H = f->H; p.W = f->W; p.eps = f->eps;
    p.pre_g = f->pre_gamma; p.pre_b = f->pre_beta; p.pre_eps = f->pre_eps;
    ffn_pick_tile<C>(f->H, f->W, (long long)f->B * f->groups, f->tile_h, f->tile_w, p.TH, p.TW);
    if ((p.TH + 2) * (p.TW + 2) > K::MPMAX || p.TH * p.TW > K::IPMAX || p.TH * ((p.TW + 1) / 2) * K::SG > 256) return TC_ERR_ARG;
    p.tilesH = (f->H + p.TH - 1) / p.TH; p.tilesW = (f->W + p.TW - 1) / p.TW;
    p.HW2 = p.TW + 2; p.HP = (p.TH + 2) * p.HW2; p.MT = (p.HP + 31) / 32; p.IP = p.TH * p.TW; p.MT2 = (p.IP + 31) / 32;
    const long long nt = (long long)f->B * p.tilesH * p.tilesW;
    // the kernels form tile / pixel indices in fp32 (sdiv: < 2^22) and per-image element offsets in 32 bits (< 2^31)
    if (nt >= (1LL << 22) || (long long)f->H * f->W >= (1LL << 23) || f->ldx >= (1 << 23) || (long long)f->H * f->W * f->ldx >= (1LL << 31)) return TC_ERR_ARG;
    p.ntiles = (int)nt;
    const int ncu = ffn_num_cus();
    int gx = (int)(nt < ncu ? nt : ncu);
    if (f->groups > 1) { gx = (ncu + f->groups - 1) / f->groups; if (gx > nt) gx = (int)nt; if (gx < 1) gx = 1; }
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)ffn_fused_fwd_kernel<H, C, PRE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::smem) != hipSuccess) return TC_ERR_LAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL((ffn_fused_fwd_kernel<H, C, PRE>), dim3(gx, f->groups), dim3(K::NTH), K::smem, s, p);
    return tc_launch_status();
}

bool ffn_fused_args_ok(const TcFfnFused* f) {
    if (!f || !f->x || !f->w1 || !f->b1 || !f->wd || !f->bd || !f->gamma || !f->beta || !f->w2 || !f->b2 || !f->out) return false;
    if (f->B < 1 || f->H < 1 || f->W < 2 || f->groups < 1 || (f->C != 64 && f->C != 128)) return false;
    auto al = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    if (!al(f->x) || !al(f->w1) || !al(f->w2) || !al(f->out) || !al(f->res) || !al(f->h) || !al(f->d) || !al(f->a) || ((uintptr_t)f->stat & 7)) return false;
    if ((f->ldx & 7) || (f->ldo & 7) || (f->res && (f->ldr & 7)) || (f->sres & 7) || (f->sout & 7) || (f->wstride & 7)) return false;
    return (long long)f->H * f->W * (f->ldx > f->ldo ? f->ldx : f->ldo) < 0x7fffffffLL;
}

}  // namespace

#ifdef TC_FFNF_TIMING
extern "C" int tc_ffnf_dbg_read(long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ffnf_dbg), sizeof(long long) * 16 * 16); }
#endif

extern "C" int tc_ffn_fused_supported(int C, int dtype) { return (C == 64 || C == 128) && (dtype == TC_BF16 || dtype == TC_F16); }

extern "C" int tc_ffn_fused_fwd(const TcFfnFused* f, int dtype, void* stream) {
    if (!ffn_fused_args_ok(f) || !tc_ffn_fused_supported(f->C, dtype)) return TC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const bool pre = f->pre_gamma != nullptr;
    if (pre && (!f->pre_beta || f->C != 64)) return TC_ERR_ARG;      // the fused LayerNorm(C) exists for the width whose backward is tiled as well
    if (dtype == TC_BF16)
        return f->C == 128 ? ffn_fused_fwd_launch<bf16_t, 128, false>(f, s) : pre ? ffn_fused_fwd_launch<bf16_t, 64, true>(f, s) : ffn_fused_fwd_launch<bf16_t, 64, false>(f, s);
    return f->C == 128 ? ffn_fused_fwd_launch<f16_t, 128, false>(f, s) : pre ? ffn_fused_fwd_launch<f16_t, 64, true>(f, s) : ffn_fused_fwd_launch<f16_t, 64, false>(f, s);
}
