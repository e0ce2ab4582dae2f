// EfficientAttention (MSTr.py:80-143, one head: :154-155) with its LayerNorm (:166) and residual add (:167) as seven launches instead of
// twenty-two:   out = t + reproj( softmax_c(Q) . ctx ),  ctx = softmax_n(K)^T V per image,  K | Q | V = LN(t) W^T + b.
// The token maps are C = 64 wide (6 MB at 56^2, B = 16): the op-by-op form was twenty launches of 5 - 35 us each that moved K, Q, V and
// their softmaxes through HBM; here nothing N-sized but the input, the output (and one gradient scratch in the backward) touches HBM.
//
// Organisation: ONE WAVE = ONE WORKGROUP = 256 consecutive tokens of one image, walked as eight 32-token blocks.  Everything row-local
// (LayerNorm, the 64 x 64 projections by MFMA, the row softmax of Q, the products with ctx) stays inside the wave in the D^T layout
// (lane = token, registers = channels); everything that reduces over tokens (the column softmax statistics of K, ctx = Ksm^T V, d_ctx,
// the weight gradients) accumulates in that wave's registers over its eight blocks and leaves as one fp32 partial per wave, which a
// small second launch folds per image / per parameter.  No barrier and no atomic anywhere: a wave only ever reads LDS it wrote itself.
//   forward   effatt_kv_kernel   LN, K, V; per-wave column max m, sums S = sum exp(K - m), P = exp(K - m)^T V        -> partials
//             effatt_ctx_kernel  per image: M = max m, Z = sum e^(m - M) S, ctx = sum e^(m - M) P / Z               -> ctx, (M, Z)
//             effatt_out_kernel  LN, Q, row softmax, att = Qsm ctx, out = att Wr^T + br + t
//   backward  effatt_bq_kernel   LN, Q, Qsm, att recomputed; d_att = dout Wr; dQ; g1 = dQ Wq -> scratch; d_ctx, dWr, dWq, dbr, dbq partials
//             effatt_dctx_kernel per image: d_ctx = sum partials, r[c] = sum_c' d_ctx[c][c'] ctx[c][c']  (= sum_n Ksm dKsm: the column
//                                softmax's backward needs no further pass over the tokens)
//             effatt_bkv_kernel  LN, K, V, Ksm recomputed; dV = Ksm d_ctx, dKsm = V d_ctx^T, dK = Ksm (dKsm - r); d_n1 = g1 + dK Wk + dV Wv;
//                                LayerNorm backward + residual -> dt; dWk, dWv, dbk, dbv, dgamma, dbeta partials
//             effatt_fold_kernel partials -> the fp32 gradient arrays
// 16-bit storage types, C = 64.
#include "tc_common.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

constexpr int C = 64, PT = C + 8, TPW = 256, NW = 4, RT = 32 * NW, NRD = TPW / RT;     // channels, LDS row pitch, tokens per workgroup, waves, tokens per round, rounds
constexpr int TOUT = RT;                                                                    // tokens per workgroup of the forward's output launch
constexpr int F_M = 0, F_S = C, F_P = 2 * C, F_N = 2 * C + C * C;                       // forward partial of a workgroup
constexpr int B1_DCTX = 0, B1_DWR = C * C, B1_DWQ = 2 * C * C, B1_DBR = 3 * C * C, B1_DBQ = B1_DBR + C, B1_N = B1_DBQ + C;
constexpr int B2_DWK = 0, B2_DWV = C * C, B2_DBK = 2 * C * C, B2_DBV = B2_DBK + C, B2_DG = B2_DBV + C, B2_DB = B2_DG + C, B2_N = B2_DB + C;
constexpr int SCW = 64 * 33;                                                                // a wave's column-reduction scratch (floats)

struct EffDev {
    const void* t; const void* gamma; const void* beta;
    const void* wk; const void* bk; const void* wq; const void* bq; const void* wv; const void* bv; const void* wr; const void* br;
    void* out; float* ctx; float* kstat; float* part; float* dctx; float* rsum;
    const void* dout; void* dt; void* g1;
    int ldt, ldo, lddo, lddt, acc_dt, B, N, wpi;
    float eps;
};

template <typename V8> __device__ __forceinline__ V8 ld_tr(const bf16_t* lo, const bf16_t* hi) {
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(lo));
    const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(hi));
    return __builtin_bit_cast(V8, (s16x8_t)__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}
template <typename H> __device__ __forceinline__ void up8s(const uint4& r, float* o) {
    unpack2<H>(r.x, o[0], o[1]); unpack2<H>(r.y, o[2], o[3]); unpack2<H>(r.z, o[4], o[5]); unpack2<H>(r.w, o[6], o[7]);
}
// Lane geometry of a wave: D^T tiles put token l31 on the lane and channels cb * 32 + 8 gq + 4 hh + j in register 4 gq + j of acc[cb].
struct Lane { int lane, l31, hh, gi, gq2, wv; };
__device__ __forceinline__ Lane lane_of() {
    Lane L; L.lane = threadIdx.x & 63; L.wv = threadIdx.x >> 6; L.l31 = L.lane & 31; L.hh = L.lane >> 5; L.gi = L.lane & 15; L.gq2 = (L.lane >> 4) & 1;
    return L;
}

__device__ __forceinline__ void zero2(f32x16 (&a)[2]) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[cb][r] = 0.f;
}
// acc^T[o][tok] += sum_k W[o][k] X[tok][k]:  W rows = output channel (k contiguous), X rows = this wave's 32 tokens; 16-byte fragment reads
template <typename H> __device__ __forceinline__ void mm_w(const bf16_t* W, const bf16_t* X, const Lane& L, f32x16 (&acc)[2]) {
    using V8 = typename TcHalf<H>::v8;
#pragma unroll
    for (int kk = 0; kk < C / 16; ++kk) {
        const V8 xv = *reinterpret_cast<const V8*>(X + L.l31 * PT + kk * 16 + 8 * L.hh);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[cb] = TcHalf<H>::mfma(*reinterpret_cast<const V8*>(W + (cb * 32 + L.l31) * PT + kk * 16 + 8 * L.hh), xv, acc[cb]);
    }
}
// acc^T[i][tok] += sum_k W[k][i] X[tok][k]:  the same weight matrix used transposed (rows = the reduction index): transpose reads
template <typename H> __device__ __forceinline__ void mm_wt(const bf16_t* W, const bf16_t* X, const Lane& L, f32x16 (&acc)[2]) {
    using V8 = typename TcHalf<H>::v8;
#pragma unroll
    for (int kk = 0; kk < C / 16; ++kk) {
        const V8 xv = *reinterpret_cast<const V8*>(X + L.l31 * PT + kk * 16 + 8 * L.hh);
        const bf16_t* wp = W + (16 * kk + 8 * L.hh + (L.gi >> 2)) * PT + 16 * L.gq2 + 4 * (L.gi & 3);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[cb] = TcHalf<H>::mfma(ld_tr<V8>(wp + cb * 32, wp + 4 * PT + cb * 32), xv, acc[cb]);
    }
}
// acc[i][j] += sum_tok X[tok][32 ib + i] Y[tok][32 jb + j] over the RT tokens of a round (both tiles in LDS, rows = tokens)
template <typename H> __device__ __forceinline__ void mm_tok(const bf16_t* X, const bf16_t* Y, int ib, int jb, const Lane& L, f32x16& acc) {
    using V8 = typename TcHalf<H>::v8;
    const int col = 16 * L.gq2 + 4 * (L.gi & 3);
#pragma unroll
    for (int ks = 0; ks < RT / 16; ++ks) {
        const int row = 16 * ks + 8 * L.hh + (L.gi >> 2);
        const V8 a = ld_tr<V8>(X + row * PT + ib * 32 + col, X + (row + 4) * PT + ib * 32 + col);
        const V8 b = ld_tr<V8>(Y + row * PT + jb * 32 + col, Y + (row + 4) * PT + jb * 32 + col);
        acc = TcHalf<H>::mfma(a, b, acc);
    }
}
// a D^T pair of tiles -> the wave's 32 rows of an LDS tile in the storage type
template <typename H> __device__ __forceinline__ void put_T(bf16_t* X, const Lane& L, const f32x16 (&acc)[2]) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            *reinterpret_cast<uint2*>(X + L.l31 * PT + cb * 32 + 8 * gq + 4 * L.hh) =
                make_uint2(pack2<H>(acc[cb][4 * gq], acc[cb][4 * gq + 1]), pack2<H>(acc[cb][4 * gq + 2], acc[cb][4 * gq + 3]));
}
// the wave's 32 rows of an LDS tile, added to a D^T pair of tiles
template <typename H> __device__ __forceinline__ void add_T(const bf16_t* X, const Lane& L, f32x16 (&acc)[2]) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const uint2 tv = *reinterpret_cast<const uint2*>(X + L.l31 * PT + cb * 32 + 8 * gq + 4 * L.hh);
            float t0, t1, t2, t3;
            unpack2<H>(tv.x, t0, t1); unpack2<H>(tv.y, t2, t3);
            acc[cb][4 * gq] += t0; acc[cb][4 * gq + 1] += t1; acc[cb][4 * gq + 2] += t2; acc[cb][4 * gq + 3] += t3;
        }
}
// per-channel vector (bias, ...) in the D^T register order, from an fp32 LDS array
__device__ __forceinline__ void get_vecT(const float* v, const Lane& L, f32x16 (&o)[2]) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float4 q = *reinterpret_cast<const float4*>(v + cb * 32 + 8 * gq + 4 * L.hh);
            o[cb][4 * gq] = q.x; o[cb][4 * gq + 1] = q.y; o[cb][4 * gq + 2] = q.z; o[cb][4 * gq + 3] = q.w;
        }
}
// Column reductions over the 32 token lanes through the wave's scratch: every lane parks its 32 per-channel values, then lane j
// gathers channel j.  D^T order: value e = 16 cb + r of lane (l31, hh) is channel cb * 32 + 8 (r >> 2) + 4 hh + (r & 3).
template <bool MAX> __device__ __forceinline__ float colred_T(float* sc, const f32x16 (&a)[2], const Lane& L) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[L.lane * 33 + cb * 16 + r] = a[cb][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    const int cb = L.lane >> 5, c5 = L.lane & 31, hh = (c5 >> 2) & 1, e = cb * 16 + 4 * (c5 >> 3) + (c5 & 3);
    float s = MAX ? -3.0e38f : 0.f;
#pragma unroll 8
    for (int l = 0; l < 32; ++l) { const float v = sc[(hh * 32 + l) * 33 + e]; s = MAX ? fmaxf(s, v) : s + v; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    return s;
}
// row order: value e of lane (l31, hh) is channel hh * 32 + e
__device__ __forceinline__ float colsum_row(float* sc, const float* v, const Lane& L) {
#pragma unroll
    for (int e = 0; e < 32; ++e) sc[L.lane * 33 + e] = v[e];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    const int hh = L.lane >> 5, e = L.lane & 31;
    float s = 0.f;
#pragma unroll 8
    for (int l = 0; l < 32; ++l) s += sc[(hh * 32 + l) * 33 + e];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    return s;
}
// one D tile (rows i, columns j) -> dst[i * C + j]
__device__ __forceinline__ void put_tile(float* dst, int ib, int jb, const f32x16& a, const Lane& L) {
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(ib * 32 + (r & 3) + 8 * (r >> 2) + 4 * L.hh) * C + jb * 32 + L.l31] = a[r];
}

// 32 token rows of a [rows, ld] map -> the wave's rows of an LDS tile (zero rows beyond the image's tokens)
template <typename H> __device__ __forceinline__ void load_tile(bf16_t* X, const H* src, long long row0, int ld, int nvalid, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * i + (lane >> 3), cg = lane & 7;
        const uint4 v = r < nvalid ? *reinterpret_cast<const uint4*>(src + (row0 + r) * ld + cg * 8) : make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(X + r * PT + cg * 8) = v;
    }
}
// the same in two halves: the global loads into registers (issued a round early), the registers into LDS
template <typename H> __device__ __forceinline__ void fetch_tile(uint4 (&v)[4], const H* src, long long row0, int ld, int nvalid, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * i + (lane >> 3), cg = lane & 7;
        v[i] = r < nvalid ? *reinterpret_cast<const uint4*>(src + (row0 + r) * ld + cg * 8) : make_uint4(0u, 0u, 0u, 0u);
    }
}
__device__ __forceinline__ void stash_tile(bf16_t* X, const uint4 (&v)[4], int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(X + (8 * i + (lane >> 3)) * PT + (lane & 7) * 8) = v[i];
}
template <typename H> __device__ __forceinline__ void store_tile(const bf16_t* X, H* dst, long long row0, int ld, int nvalid, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * i + (lane >> 3), cg = lane & 7;
        if (r < nvalid) *reinterpret_cast<uint4*>(dst + (row0 + r) * ld + cg * 8) = *reinterpret_cast<const uint4*>(X + r * PT + cg * 8);
    }
}
// weights [C][C] (storage type) -> LDS [C][PT]; vectors -> fp32 LDS (whole workgroup)
template <typename H> __device__ __forceinline__ void load_w(bf16_t* W, const void* src) {
    const H* s = reinterpret_cast<const H*>(src);
    for (int i = threadIdx.x; i < C * (C / 8); i += 64 * NW) { const int r = i >> 3, cg = i & 7; *reinterpret_cast<uint4*>(W + r * PT + cg * 8) = *reinterpret_cast<const uint4*>(s + r * C + cg * 8); }
}
template <typename H> __device__ __forceinline__ void load_v(float* v, const void* src) { if (threadIdx.x < C) v[threadIdx.x] = ldf<H>(reinterpret_cast<const H*>(src) + threadIdx.x); }
// a [C][C] fp32 matrix -> LDS in the storage type
template <typename H> __device__ __forceinline__ void load_ctx(bf16_t* W, const float* src) {
    for (int i = threadIdx.x; i < C * (C / 4); i += 64 * NW) {
        const int r = i >> 4, c4 = (i & 15) * 4;
        const float4 v = *reinterpret_cast<const float4*>(src + r * C + c4);
        *reinterpret_cast<uint2*>(W + r * PT + c4) = make_uint2(pack2<H>(v.x, v.y), pack2<H>(v.z, v.w));
    }
}

// LayerNorm of the wave's 32 token rows of an LDS tile, in place (lane = token, hh = channel half); xhat kept in registers when asked for
template <typename H, bool KEEP> __device__ __forceinline__ void ln_tile(bf16_t* X, const float* gam, const float* bet, float eps, const Lane& L, float& rstd, float* xh) {
    float v[32];
#pragma unroll
    for (int q = 0; q < 4; ++q) up8s<H>(*reinterpret_cast<const uint4*>(X + L.l31 * PT + L.hh * 32 + q * 8), v + 8 * q);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 32; ++e) s += v[e];
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / C);
    float q2 = 0.f;
#pragma unroll
    for (int e = 0; e < 32; ++e) { const float d = v[e] - mean; q2 += d * d; }
    q2 += __shfl_xor(q2, 32, 64);
    rstd = rsqrtf(q2 * (1.0f / C) + eps);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = L.hh * 32 + q * 8 + e;
            const float xhat = (v[8 * q + e] - mean) * rstd;
            if (KEEP) xh[8 * q + e] = xhat;
            o[e] = xhat * gam[ch] + bet[ch];
        }
        *reinterpret_cast<uint4*>(X + L.l31 * PT + L.hh * 32 + q * 8) = make_uint4(pack2<H>(o[0], o[1]), pack2<H>(o[2], o[3]), pack2<H>(o[4], o[5]), pack2<H>(o[6], o[7]));
    }
}
// (a wave's LDS accesses execute in order; the compiler must not move them across each other: one fence where rows change hands)
__device__ __forceinline__ void lds_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// row softmax of a D^T pair of tiles over the 64 channels of each token (in lane + the partner half-wave)
__device__ __forceinline__ void row_softmax(f32x16 (&a)[2]) {
    float m = a[0][0];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, a[cb][r]);
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float s = 0.f;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { a[cb][r] = __expf(a[cb][r] - m); s += a[cb][r]; }
    s += __shfl_xor(s, 32, 64);
    const float inv = 1.0f / s;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[cb][r] *= inv;
}

struct Where { int b, w; long long row0; int ntok; };
__device__ __forceinline__ Where where_am_i(int wpi, int N, int tpw) {
    Where q;
    q.b = blockIdx.x / wpi; q.w = blockIdx.x - q.b * wpi;
    q.row0 = (long long)q.b * N + (long long)q.w * tpw;
    q.ntok = min(tpw, N - q.w * tpw);
    return q;
}

// ---------------------------------------------------------------------------------------------------------------- forward 1
template <typename H>
__global__ __launch_bounds__(64 * NW) void effatt_kv_kernel(const EffDev p) {
    __shared__ __attribute__((aligned(16))) bf16_t wk[C * PT], wv[C * PT], xt[TPW * PT], ev[2 * RT * PT];
    __shared__ float vec[4 * C], mxs[NW][C], mall[C];              // gamma, beta, bk, bv
    bf16_t* et = ev; bf16_t* vt = ev + RT * PT;
    float* sc = reinterpret_cast<float*>(ev);                      // column-reduction scratch over et | vt, between their uses
    static_assert(NW * SCW * 4 <= 2 * RT * PT * 2, "scratch");
    const Lane L = lane_of();
    const Where q = where_am_i(p.wpi, p.N, TPW);
    const H* T = reinterpret_cast<const H*>(p.t);
    uint4 tf[NRD][4];                                              // the wave's token rows of both rounds: in flight under the parameter staging
#pragma unroll
    for (int rd = 0; rd < NRD; ++rd) fetch_tile<H>(tf[rd], T, q.row0 + (rd * NW + L.wv) * 32, p.ldt, q.ntok - (rd * NW + L.wv) * 32, L.lane);
    load_w<H>(wk, p.wk); load_w<H>(wv, p.wv);
    load_v<H>(vec, p.gamma); load_v<H>(vec + C, p.beta); load_v<H>(vec + 2 * C, p.bk); load_v<H>(vec + 3 * C, p.bv);
    __syncthreads();
    // pass 1: LN of the wave's blocks (kept in LDS), their K^T (kept in registers), the column maximum
    f32x16 kk[NRD][2], mx[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx[cb][r] = -3.0e38f;
#pragma unroll
    for (int rd = 0; rd < NRD; ++rd) {
        const int blk = rd * NW + L.wv, nv = q.ntok - blk * 32;
        bf16_t* X = xt + blk * 32 * PT;
        stash_tile(X, tf[rd], L.lane);
        lds_fence();
        float rstd;
        ln_tile<H, false>(X, vec, vec + C, p.eps, L, rstd, nullptr);
        lds_fence();
        get_vecT(vec + 2 * C, L, kk[rd]);
        mm_w<H>(wk, X, L, kk[rd]);
        const bool ok = L.l31 < nv;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { if (!ok) kk[rd][cb][r] = -3.0e38f; mx[cb][r] = fmaxf(mx[cb][r], kk[rd][cb][r]); }
    }
    mxs[L.wv][L.lane] = colred_T<true>(sc + L.wv * SCW, mx, L);
    __syncthreads();
    if (threadIdx.x < C) mall[threadIdx.x] = fmaxf(fmaxf(mxs[0][threadIdx.x], mxs[1][threadIdx.x]), fmaxf(mxs[2][threadIdx.x], mxs[3][threadIdx.x]));
    __syncthreads();
    get_vecT(mall, L, mx);
    // pass 2: E = exp(K - m), S += E, P += E^T V (wave = one 32 x 32 tile of P over the round's tokens)
    f32x16 ss[2], pp;
    zero2(ss);
#pragma unroll
    for (int r = 0; r < 16; ++r) pp[r] = 0.f;
    const int ib = L.wv >> 1, jb = L.wv & 1;
#pragma unroll
    for (int rd = 0; rd < NRD; ++rd) {
        const int blk = rd * NW + L.wv, nv = q.ntok - blk * 32;
        const bf16_t* X = xt + blk * 32 * PT;
        const bool ok = L.l31 < nv;
        f32x16 vv[2];
        get_vecT(vec + 3 * C, L, vv);
        mm_w<H>(wv, X, L, vv);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = ok ? __expf(kk[rd][cb][r] - mx[cb][r]) : 0.f;
                kk[rd][cb][r] = e; ss[cb][r] += e;
                if (!ok) vv[cb][r] = 0.f;
            }
        put_T<H>(et + L.wv * 32 * PT, L, kk[rd]);
        put_T<H>(vt + L.wv * 32 * PT, L, vv);
        __syncthreads();
        mm_tok<H>(et, vt, ib, jb, L, pp);
        __syncthreads();
    }
    float* PB = p.part + (long long)blockIdx.x * F_N;
    put_tile(PB + F_P, ib, jb, pp, L);
    mxs[L.wv][L.lane] = colred_T<false>(sc + L.wv * SCW, ss, L);
    __syncthreads();
    if (threadIdx.x < C) {
        PB[F_S + threadIdx.x] = mxs[0][threadIdx.x] + mxs[1][threadIdx.x] + mxs[2][threadIdx.x] + mxs[3][threadIdx.x];
        PB[F_M + threadIdx.x] = mall[threadIdx.x];
    }
}

// ---------------------------------------------------------------------------------------------------------------- forward 2
// grid (B, C): thread = (channel c = blockIdx.y, column c'); ctx[b][c][c'], kstat[b] = (M[C], Z[C]).  The partials of an image are read
// sixteen at a time with every load of a batch issued before the first use (a loop over a run-time count issues them one by one).
__global__ __launch_bounds__(64) void effatt_ctx_kernel(const EffDev p) {
    const int b = blockIdx.x, c = blockIdx.y, cc = threadIdx.x;
    const float* PB = p.part + (long long)b * p.wpi * F_N;
    float M = -3.0e38f;
    for (int w0 = 0; w0 < p.wpi; w0 += 16) {
        float mv[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) mv[k] = w0 + k < p.wpi ? PB[(long long)(w0 + k) * F_N + F_M + c] : -3.0e38f;
#pragma unroll
        for (int k = 0; k < 16; ++k) M = fmaxf(M, mv[k]);
    }
    float Z = 0.f, acc = 0.f;
    for (int w0 = 0; w0 < p.wpi; w0 += 16) {
        float mv[16], sv[16], pv[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const bool in = w0 + k < p.wpi;
            const float* q = PB + (long long)(in ? w0 + k : 0) * F_N;
            mv[k] = in ? q[F_M + c] : -3.0e38f; sv[k] = in ? q[F_S + c] : 0.f; pv[k] = in ? q[F_P + c * C + cc] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) { const float f = __expf(mv[k] - M); Z += f * sv[k]; acc += f * pv[k]; }
    }
    p.ctx[((long long)b * C + c) * C + cc] = acc / Z;
    if (cc == 0) { p.kstat[(long long)b * 2 * C + c] = M; p.kstat[(long long)b * 2 * C + C + c] = Z; }
}

// ---------------------------------------------------------------------------------------------------------------- forward 3
// workgroup = TOUT tokens, wave = 32 of them, nothing shared but the weights
template <typename H>
__global__ __launch_bounds__(64 * NW) void effatt_out_kernel(const EffDev p) {
    __shared__ __attribute__((aligned(16))) bf16_t wq[C * PT], wr[C * PT], cx[C * PT], nts[TOUT * PT], sts[TOUT * PT];      // 66 KB: two workgroups per CU
    __shared__ float vec[4 * C];                                   // gamma, beta, bq, br
    const Lane L = lane_of();
    const int wpo = (p.N + TOUT - 1) / TOUT;
    const Where q = where_am_i(wpo, p.N, TOUT);
    load_w<H>(wq, p.wq); load_w<H>(wr, p.wr); load_ctx<H>(cx, p.ctx + (long long)q.b * C * C);
    load_v<H>(vec, p.gamma); load_v<H>(vec + C, p.beta); load_v<H>(vec + 2 * C, p.bq); load_v<H>(vec + 3 * C, p.br);
    const H* T = reinterpret_cast<const H*>(p.t);
    H* O = reinterpret_cast<H*>(p.out);
    bf16_t* nt = nts + L.wv * 32 * PT; bf16_t* st = sts + L.wv * 32 * PT;
    const int nv = q.ntok - L.wv * 32;
    const long long row = q.row0 + L.wv * 32;
    uint4 tf[4];                                                   // t: normalised in LDS, and kept as it is for the residual
    fetch_tile<H>(tf, T, row, p.ldt, nv, L.lane);
    stash_tile(nt, tf, L.lane);
    __syncthreads();
    if (nv <= 0) return;
    float rstd;
    ln_tile<H, false>(nt, vec, vec + C, p.eps, L, rstd, nullptr);
    lds_fence();
    f32x16 a[2];
    get_vecT(vec + 2 * C, L, a);
    mm_w<H>(wq, nt, L, a);
    row_softmax(a);
    put_T<H>(st, L, a);                                            // Qsm
    lds_fence();
    zero2(a);
    mm_wt<H>(cx, st, L, a);                                        // att^T[c'][tok] = sum_c ctx[c][c'] Qsm[tok][c]
    lds_fence();
    put_T<H>(st, L, a);                                            // att (every read of Qsm is done)
    lds_fence();
    get_vecT(vec + 3 * C, L, a);
    mm_w<H>(wr, st, L, a);                                         // out^T[o][tok] = sum_c' Wr[o][c'] att[tok][c']
    lds_fence();
    // fp32 on the way out (one rounding, after the residual): channels 0..31 over the wave's att rows, 32..63 over its n1 rows (a row of
    // PT 16-bit elements holds 36 floats)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(cb ? nt : st) + L.l31 * (PT / 2) + 8 * gq + 4 * L.hh) =
                make_float4(a[cb][4 * gq], a[cb][4 * gq + 1], a[cb][4 * gq + 2], a[cb][4 * gq + 3]);
    lds_fence();
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                  // + t, row-wise
        const int r = 8 * i + (L.lane >> 3), cg = L.lane & 7;
        const float* src = reinterpret_cast<const float*>((cg >> 2) ? nt : st) + r * (PT / 2) + (cg & 3) * 8;
        const float4 o0 = *reinterpret_cast<const float4*>(src), o1 = *reinterpret_cast<const float4*>(src + 4);
        float o[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w}, x[8];
        up8s<H>(tf[i], x);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += x[e];
        if (r < nv) *reinterpret_cast<uint4*>(O + (row + r) * p.ldo + cg * 8) = make_uint4(pack2<H>(o[0], o[1]), pack2<H>(o[2], o[3]), pack2<H>(o[4], o[5]), pack2<H>(o[6], o[7]));
    }
}

// ---------------------------------------------------------------------------------------------------------------- backward 1
template <typename H>
__global__ __launch_bounds__(64 * NW) void effatt_bq_kernel(const EffDev p) {
    __shared__ __attribute__((aligned(16))) bf16_t wq[C * PT], wr[C * PT], cx[C * PT];
    __shared__ __attribute__((aligned(16))) bf16_t tl[6 * RT * PT];                    // n1, Qsm, att, dout, d_att, dQ: [RT tokens][PT] each
    __shared__ float vec[3 * C], vs[NW][2][C];                     // gamma, beta, bq
    bf16_t* nts = tl; bf16_t* qts = tl + RT * PT; bf16_t* ats = tl + 2 * RT * PT; bf16_t* yts = tl + 3 * RT * PT; bf16_t* dats = tl + 4 * RT * PT; bf16_t* dqts = tl + 5 * RT * PT;
    float* sc = reinterpret_cast<float*>(tl);
    const Lane L = lane_of();
    const Where q = where_am_i(p.wpi, p.N, TPW);
    const H* T = reinterpret_cast<const H*>(p.t);
    const H* DY = reinterpret_cast<const H*>(p.dout);
    H* G1 = reinterpret_cast<H*>(p.g1);
    uint4 tf[4], yf[4];                                            // next round's token / gradient rows, a round ahead
    fetch_tile<H>(tf, T, q.row0 + L.wv * 32, p.ldt, q.ntok - L.wv * 32, L.lane);
    fetch_tile<H>(yf, DY, q.row0 + L.wv * 32, p.lddo, q.ntok - L.wv * 32, L.lane);
    load_w<H>(wq, p.wq); load_w<H>(wr, p.wr); load_ctx<H>(cx, p.ctx + (long long)q.b * C * C);
    load_v<H>(vec, p.gamma); load_v<H>(vec + C, p.beta); load_v<H>(vec + 2 * C, p.bq);
    const int wo = L.wv * 32 * PT;
    bf16_t* nt = nts + wo; bf16_t* qt = qts + wo; bf16_t* at = ats + wo; bf16_t* yt = yts + wo; bf16_t* dat = dats + wo; bf16_t* dqt = dqts + wo;
    __syncthreads();
    f32x16 dctx, dwr, dwq, dbr[2], dbq[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dctx[r] = 0.f; dwr[r] = 0.f; dwq[r] = 0.f; }
    zero2(dbr); zero2(dbq);
    const int ib = L.wv >> 1, jb = L.wv & 1;
    for (int rd = 0; rd < NRD; ++rd) {
        const int blk = rd * NW + L.wv, nv = q.ntok - blk * 32;
        const long long row = q.row0 + blk * 32;
        const bool ok = L.l31 < nv;
        stash_tile(nt, tf, L.lane);
        stash_tile(yt, yf, L.lane);                                // (rows beyond the image: zeros -> no contribution)
        if (rd + 1 < NRD) {
            fetch_tile<H>(tf, T, row + NW * 32, p.ldt, nv - NW * 32, L.lane);
            fetch_tile<H>(yf, DY, row + NW * 32, p.lddo, nv - NW * 32, L.lane);
        }
        lds_fence();
        float rstd;
        ln_tile<H, false>(nt, vec, vec + C, p.eps, L, rstd, nullptr);
        lds_fence();
        f32x16 qs[2], a[2];
        get_vecT(vec + 2 * C, L, qs);
        mm_w<H>(wq, nt, L, qs);
        row_softmax(qs);
        put_T<H>(qt, L, qs);                                       // Qsm
        lds_fence();
        zero2(a);
        mm_wt<H>(cx, qt, L, a);
        put_T<H>(at, L, a);                                        // att = Qsm ctx (recomputed for dWr)
        zero2(a);
        mm_wt<H>(wr, yt, L, a);                                    // d_att^T[c'][tok] = sum_o Wr[o][c'] dout[tok][o]
        put_T<H>(dat, L, a);
        lds_fence();
        zero2(a);
        mm_w<H>(cx, dat, L, a);                                    // dQsm^T[c][tok] = sum_c' ctx[c][c'] d_att[tok][c']
        float dot = 0.f;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) dot += qs[cb][r] * a[cb][r];
        dot += __shfl_xor(dot, 32, 64);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { a[cb][r] = ok ? qs[cb][r] * (a[cb][r] - dot) : 0.f; dbq[cb][r] += a[cb][r]; }       // dQ
        put_T<H>(dqt, L, a);
        lds_fence();
        add_T<H>(yt, L, dbr);                                      // dbr: column sums of dout
        zero2(a);
        mm_wt<H>(wq, dqt, L, a);                                   // g1^T[cin][tok] = sum_c Wq[c][cin] dQ[tok][c]
        __syncthreads();
        mm_tok<H>(qts, dats, ib, jb, L, dctx);                     // d_ctx[c][c'] += sum_tok Qsm[tok][c] d_att[tok][c']
        mm_tok<H>(yts, ats, ib, jb, L, dwr);                       // dWr[o][c']   += sum_tok dout[tok][o] att[tok][c']
        mm_tok<H>(dqts, nts, ib, jb, L, dwq);                      // dWq[c][cin]  += sum_tok dQ[tok][c] n1[tok][cin]
        __syncthreads();
        put_T<H>(at, L, a);
        lds_fence();
        store_tile<H>(at, G1, row, C, nv, L.lane);
        lds_fence();
    }
    float* PB = p.part + (long long)blockIdx.x * (B1_N + B2_N);
    put_tile(PB + B1_DCTX, ib, jb, dctx, L); put_tile(PB + B1_DWR, ib, jb, dwr, L); put_tile(PB + B1_DWQ, ib, jb, dwq, L);
    __syncthreads();
    vs[L.wv][0][L.lane] = colred_T<false>(sc + L.wv * SCW, dbr, L);
    vs[L.wv][1][L.lane] = colred_T<false>(sc + L.wv * SCW, dbq, L);
    __syncthreads();
    if (threadIdx.x < 2 * C) {
        const int k = threadIdx.x >> 6, ch = threadIdx.x & 63;
        PB[(k ? B1_DBQ : B1_DBR) + ch] = vs[0][k][ch] + vs[1][k][ch] + vs[2][k][ch] + vs[3][k][ch];
    }
}

// grid (B, C): d_ctx[b][c][c'] = sum over the image's workgroups; r[b][c] = sum_c' d_ctx ctx
__global__ __launch_bounds__(64) void effatt_dctx_kernel(const EffDev p) {
    const int b = blockIdx.x, c = blockIdx.y, cc = threadIdx.x;
    const float* PB = p.part + (long long)b * p.wpi * (B1_N + B2_N);
    float acc = 0.f;
    for (int w0 = 0; w0 < p.wpi; w0 += 16) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = w0 + k < p.wpi ? PB[(long long)(w0 + k) * (B1_N + B2_N) + B1_DCTX + c * C + cc] : 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += v[k];
    }
    p.dctx[((long long)b * C + c) * C + cc] = acc;
    float r = acc * p.ctx[((long long)b * C + c) * C + cc];
    r = wave_sum(r);
    if (cc == 0) p.rsum[(long long)b * C + c] = r;
}

// ---------------------------------------------------------------------------------------------------------------- backward 2
template <typename H>
__global__ __launch_bounds__(64 * NW) void effatt_bkv_kernel(const EffDev p) {
    __shared__ __attribute__((aligned(16))) bf16_t wk[C * PT], wv[C * PT], dcx[C * PT];
    __shared__ __attribute__((aligned(16))) bf16_t tl[6 * RT * PT];                    // n1, Ksm, V, dK, dV, g1: [RT tokens][PT] each
    __shared__ float vec[7 * C], vs[NW][4][C];                     // gamma, beta, bk, bv, M, 1 / Z, r
    bf16_t* nts = tl; bf16_t* dkts = tl + 3 * RT * PT; bf16_t* dvts = tl + 4 * RT * PT;
    float* sc = reinterpret_cast<float*>(tl);
    const Lane L = lane_of();
    const Where q = where_am_i(p.wpi, p.N, TPW);
    load_w<H>(wk, p.wk); load_w<H>(wv, p.wv); load_ctx<H>(dcx, p.dctx + (long long)q.b * C * C);
    load_v<H>(vec, p.gamma); load_v<H>(vec + C, p.beta); load_v<H>(vec + 2 * C, p.bk); load_v<H>(vec + 3 * C, p.bv);
    if (threadIdx.x < C) {
        vec[4 * C + threadIdx.x] = p.kstat[(long long)q.b * 2 * C + threadIdx.x];
        vec[5 * C + threadIdx.x] = 1.0f / p.kstat[(long long)q.b * 2 * C + C + threadIdx.x];
        vec[6 * C + threadIdx.x] = p.rsum[(long long)q.b * C + threadIdx.x];
    }
    const H* T = reinterpret_cast<const H*>(p.t);
    const H* DY = reinterpret_cast<const H*>(p.dout);
    const H* G1 = reinterpret_cast<const H*>(p.g1);
    H* DT = reinterpret_cast<H*>(p.dt);
    uint4 tf[4], gf[4];                                            // next round's token / g1 rows, a round ahead
    fetch_tile<H>(tf, T, q.row0 + L.wv * 32, p.ldt, q.ntok - L.wv * 32, L.lane);
    const int wo = L.wv * 32 * PT;
    bf16_t* nt = tl + wo; bf16_t* kt = tl + RT * PT + wo; bf16_t* vt = tl + 2 * RT * PT + wo; bf16_t* dkt = dkts + wo; bf16_t* dvt = dvts + wo; bf16_t* gt = tl + 5 * RT * PT + wo;
    __syncthreads();
    f32x16 dwk, dwv, dbk[2], dbv[2];
    float dgr[32], dbr[32];                                        // dgamma / dbeta of the lane's token, row order (channels hh * 32 + e)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dwk[r] = 0.f; dwv[r] = 0.f; }
    zero2(dbk); zero2(dbv);
#pragma unroll
    for (int e = 0; e < 32; ++e) { dgr[e] = 0.f; dbr[e] = 0.f; }
    const int ib = L.wv >> 1, jb = L.wv & 1;
    for (int rd = 0; rd < NRD; ++rd) {
        const int blk = rd * NW + L.wv, nv = q.ntok - blk * 32;
        const long long row = q.row0 + blk * 32;
        const bool ok = L.l31 < nv;
        stash_tile(nt, tf, L.lane);
        uint4 yf[4], df[4];
        fetch_tile<H>(gf, G1, row, C, nv, L.lane);                 // g1, dout and (accumulating) dt: used late in the round, requested now
        fetch_tile<H>(yf, DY, row, p.lddo, nv, L.lane);
        if (p.acc_dt) fetch_tile<H>(df, DT, row, p.lddt, nv, L.lane);
        if (rd + 1 < NRD) fetch_tile<H>(tf, T, row + NW * 32, p.ldt, nv - NW * 32, L.lane);
        lds_fence();
        float rstd, xh[32];
        ln_tile<H, true>(nt, vec, vec + C, p.eps, L, rstd, xh);
        lds_fence();
        f32x16 ks[2], a[2], b2[2];
        get_vecT(vec + 2 * C, L, ks);
        mm_w<H>(wk, nt, L, ks);
        get_vecT(vec + 3 * C, L, a);
        mm_w<H>(wv, nt, L, a);
        put_T<H>(vt, L, a);                                        // V
        get_vecT(vec + 4 * C, L, a); get_vecT(vec + 5 * C, L, b2);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) ks[cb][r] = __expf(ks[cb][r] - a[cb][r]) * b2[cb][r];                                    // Ksm
        put_T<H>(kt, L, ks);
        lds_fence();
        zero2(a);
        mm_wt<H>(dcx, kt, L, a);                                   // dV^T[c'][tok]  = sum_c d_ctx[c][c'] Ksm[tok][c]
        zero2(b2);
        mm_w<H>(dcx, vt, L, b2);                                   // dKsm^T[c][tok] = sum_c' d_ctx[c][c'] V[tok][c']
        {
            f32x16 rv[2];
            get_vecT(vec + 6 * C, L, rv);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    b2[cb][r] = ok ? ks[cb][r] * (b2[cb][r] - rv[cb][r]) : 0.f;                                                   // dK
                    if (!ok) a[cb][r] = 0.f;
                    dbk[cb][r] += b2[cb][r]; dbv[cb][r] += a[cb][r];
                }
        }
        put_T<H>(dkt, L, b2);
        put_T<H>(dvt, L, a);
        lds_fence();
        zero2(a);
        mm_wt<H>(wk, dkt, L, a);                                   // d_n1^T[cin][tok] = g1 + sum_c Wk[c][cin] dK[tok][c] + sum_c' Wv[c'][cin] dV[tok][c']
        mm_wt<H>(wv, dvt, L, a);
        lds_fence();
        put_T<H>(kt, L, a);                                        // (Ksm's readers are done): the two GEMM terms, by token rows
        stash_tile(gt, gf, L.lane);
        lds_fence();
        // LayerNorm backward in row order: lane = token, its channels hh * 32 + 0..31
        float dn[32], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            float x[8], y[8];
            up8s<H>(*reinterpret_cast<const uint4*>(kt + L.l31 * PT + L.hh * 32 + qq * 8), x);
            up8s<H>(*reinterpret_cast<const uint4*>(gt + L.l31 * PT + L.hh * 32 + qq * 8), y);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = ok ? x[e] + y[e] : 0.f;             // d_n1
                dgr[qq * 8 + e] += d * xh[qq * 8 + e]; dbr[qq * 8 + e] += d;
                const float dg = d * vec[L.hh * 32 + qq * 8 + e];
                dn[qq * 8 + e] = dg;
                s1 += dg; s2 += dg * xh[qq * 8 + e];
            }
        }
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        const float k1 = s1 * (1.0f / C), k2 = s2 * (1.0f / C);
        lds_fence();
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rstd * (dn[qq * 8 + e] - k1 - xh[qq * 8 + e] * k2);
            *reinterpret_cast<uint4*>(vt + L.l31 * PT + L.hh * 32 + qq * 8) = make_uint4(pack2<H>(o[0], o[1]), pack2<H>(o[2], o[3]), pack2<H>(o[4], o[5]), pack2<H>(o[6], o[7]));
        }
        // dt = LayerNorm backward + the residual's gradient (dout) [+ what dt holds]
        stash_tile(kt, yf, L.lane);
        if (p.acc_dt) stash_tile(gt, df, L.lane);
        lds_fence();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 8 * i + (L.lane >> 3), cg = L.lane & 7;
            float x[8], y[8];
            up8s<H>(*reinterpret_cast<const uint4*>(vt + r * PT + cg * 8), x);
            up8s<H>(*reinterpret_cast<const uint4*>(kt + r * PT + cg * 8), y);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] += y[e];
            if (p.acc_dt) {
                up8s<H>(*reinterpret_cast<const uint4*>(gt + r * PT + cg * 8), y);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] += y[e];
            }
            if (r < nv) *reinterpret_cast<uint4*>(DT + (row + r) * p.lddt + cg * 8) =
                make_uint4(pack2<H>(x[0], x[1]), pack2<H>(x[2], x[3]), pack2<H>(x[4], x[5]), pack2<H>(x[6], x[7]));
        }
        __syncthreads();
        mm_tok<H>(dkts, nts, ib, jb, L, dwk);                      // dWk[c][cin] += sum_tok dK[tok][c] n1[tok][cin]
        mm_tok<H>(dvts, nts, ib, jb, L, dwv);
        __syncthreads();
    }
    float* PB = p.part + (long long)blockIdx.x * (B1_N + B2_N) + B1_N;
    put_tile(PB + B2_DWK, ib, jb, dwk, L); put_tile(PB + B2_DWV, ib, jb, dwv, L);
    vs[L.wv][0][L.lane] = colred_T<false>(sc + L.wv * SCW, dbk, L);
    vs[L.wv][1][L.lane] = colred_T<false>(sc + L.wv * SCW, dbv, L);
    vs[L.wv][2][L.lane] = colsum_row(sc + L.wv * SCW, dgr, L);
    vs[L.wv][3][L.lane] = colsum_row(sc + L.wv * SCW, dbr, L);
    __syncthreads();
    {
        const int k = threadIdx.x >> 6, ch = threadIdx.x & 63;
        const int off = k == 0 ? B2_DBK : k == 1 ? B2_DBV : k == 2 ? B2_DG : B2_DB;
        PB[off + ch] = vs[0][k][ch] + vs[1][k][ch] + vs[2][k][ch] + vs[3][k][ch];
    }
}

// ---------------------------------------------------------------------------------------------------------------- fold
struct FoldDev { const float* part; float* dst[12]; int off[13]; int nwv, nper; };
// grid (nper / 64), 256 threads = 16 float4 columns x 16 slices of the workgroup list
__global__ __launch_bounds__(256) void effatt_fold_kernel(const FoldDev p) {
    __shared__ float4 sh[16][17];
    const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int q = blockIdx.x * 16 + e;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q * 4 < p.nper) {
        const float* base = p.part + (long long)q * 4;
        for (int w0 = sl; w0 < p.nwv; w0 += 16 * 8) {             // eight loads in flight per thread
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = w0 + 16 * k < p.nwv ? *reinterpret_cast<const float4*>(base + (long long)(w0 + 16 * k) * p.nper) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 8; ++k) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
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
        for (int k = 1; k < 12; ++k) seg += idx >= p.off[k];
        float* d = p.dst[seg];
        if (d) { d += idx - p.off[seg]; d[0] += s.x; d[1] += s.y; d[2] += s.z; d[3] += s.w; }
    }
}

bool eff_ok(const TcEffAtt* f, bool bwd) {
    if (!f || !f->t || !f->gamma || !f->beta || !f->wk || !f->bk || !f->wq || !f->bq || !f->wv || !f->bv || !f->wr || !f->br || !f->ctx || !f->kstat || !f->part) return false;
    if (f->C != C || f->B < 1 || f->N < 1 || (f->ldt & 7) || ((uintptr_t)f->t & 15) || ((uintptr_t)f->part & 15) || ((uintptr_t)f->ctx & 15)) return false;
    if (((uintptr_t)f->wk | (uintptr_t)f->wq | (uintptr_t)f->wv | (uintptr_t)f->wr) & 15) return false;      // weight rows are fetched 16 bytes at a time
    if ((long long)f->B * f->N >= 0x7fffffffLL / (f->ldt > C ? f->ldt : C)) return false;
    if (!bwd) return f->out && !(f->ldo & 7) && !((uintptr_t)f->out & 15);
    if (!f->dout || !f->dt || !f->g1 || (f->lddo & 7) || (f->lddt & 7) || ((uintptr_t)f->dout & 15) || ((uintptr_t)f->dt & 15) || ((uintptr_t)f->g1 & 15)) return false;
    return f->dgamma && f->dbeta && f->dwk && f->dbk && f->dwq && f->dbq && f->dwv && f->dbv && f->dwr && f->dbr;
}
long long eff_scratch(int B, int N) {
    const long long nwv = (long long)B * ((N + TPW - 1) / TPW);
    const long long per = (B1_N + B2_N) > F_N ? (B1_N + B2_N) : F_N;
    return nwv * per + (long long)B * C * C + (long long)B * C;
}
EffDev eff_dev(const TcEffAtt* f) {
    EffDev p;
    p.t = f->t; p.gamma = f->gamma; p.beta = f->beta; p.wk = f->wk; p.bk = f->bk; p.wq = f->wq; p.bq = f->bq; p.wv = f->wv; p.bv = f->bv; p.wr = f->wr; p.br = f->br;
    p.out = f->out; p.ctx = f->ctx; p.kstat = f->kstat; p.part = f->part; p.dout = f->dout; p.dt = f->dt; p.g1 = f->g1;
    p.ldt = f->ldt; p.ldo = f->ldo; p.lddo = f->lddo; p.lddt = f->lddt; p.acc_dt = f->acc_dt; p.B = f->B; p.N = f->N; p.eps = f->eps;
    p.wpi = (f->N + TPW - 1) / TPW;
    const long long nwv = (long long)f->B * p.wpi;
    const long long per = (B1_N + B2_N) > F_N ? (B1_N + B2_N) : F_N;
    p.dctx = f->part + nwv * per;
    p.rsum = p.dctx + (long long)f->B * C * C;
    return p;
}

template <typename H> int eff_fwd(const TcEffAtt* f, hipStream_t s) {
    const EffDev p = eff_dev(f);
    const int nwv = f->B * p.wpi;
    hipLaunchKernelGGL((effatt_kv_kernel<H>), dim3(nwv), dim3(64 * NW), 0, s, p);
    if (tc_launch_status() != TC_OK) return TC_ERR_LAUNCH;
    hipLaunchKernelGGL(effatt_ctx_kernel, dim3(f->B, C), dim3(64), 0, s, p);
    if (tc_launch_status() != TC_OK) return TC_ERR_LAUNCH;
    hipLaunchKernelGGL((effatt_out_kernel<H>), dim3(f->B * ((f->N + TOUT - 1) / TOUT)), dim3(64 * NW), 0, s, p);
    return tc_launch_status();
}
template <typename H> int eff_bwd(const TcEffAtt* f, hipStream_t s) {
    const EffDev p = eff_dev(f);
    const int nwv = f->B * p.wpi;
    hipLaunchKernelGGL((effatt_bq_kernel<H>), dim3(nwv), dim3(64 * NW), 0, s, p);
    if (tc_launch_status() != TC_OK) return TC_ERR_LAUNCH;
    hipLaunchKernelGGL(effatt_dctx_kernel, dim3(f->B, C), dim3(64), 0, s, p);
    if (tc_launch_status() != TC_OK) return TC_ERR_LAUNCH;
    hipLaunchKernelGGL((effatt_bkv_kernel<H>), dim3(nwv), dim3(64 * NW), 0, s, p);
    if (tc_launch_status() != TC_OK) return TC_ERR_LAUNCH;
    FoldDev r;
    r.part = f->part; r.nwv = nwv; r.nper = B1_N + B2_N;
    float* dsts[12] = {nullptr, f->dwr, f->dwq, f->dbr, f->dbq, f->dwk, f->dwv, f->dbk, f->dbv, f->dgamma, f->dbeta, nullptr};
    const int offs[13] = {B1_DCTX, B1_DWR, B1_DWQ, B1_DBR, B1_DBQ, B1_N + B2_DWK, B1_N + B2_DWV, B1_N + B2_DBK, B1_N + B2_DBV, B1_N + B2_DG, B1_N + B2_DB, B1_N + B2_N, B1_N + B2_N};
    for (int i = 0; i < 12; ++i) r.dst[i] = dsts[i];
    for (int i = 0; i < 13; ++i) r.off[i] = offs[i];
    hipLaunchKernelGGL(effatt_fold_kernel, dim3((r.nper / 4 + 15) / 16), dim3(256), 0, s, r);
    return tc_launch_status();
}

}  // namespace

extern "C" int tc_effatt_supported(int Cc, int dtype) { return Cc == C && (dtype == TC_BF16 || dtype == TC_F16); }
extern "C" long long tc_effatt_scratch_floats(int Cc, int B, int N) { return (Cc == C && B > 0 && N > 0) ? eff_scratch(B, N) : 0; }
extern "C" int tc_effatt_fwd(const TcEffAtt* f, int dtype, void* stream) {
    if (!eff_ok(f, false) || !tc_effatt_supported(f->C, dtype) || f->part_floats < eff_scratch(f->B, f->N)) return TC_ERR_ARG;
    return dtype == TC_BF16 ? eff_fwd<bf16_t>(f, (hipStream_t)stream) : eff_fwd<f16_t>(f, (hipStream_t)stream);
}
extern "C" int tc_effatt_bwd(const TcEffAtt* f, int dtype, void* stream) {
    if (!eff_ok(f, true) || !tc_effatt_supported(f->C, dtype) || f->part_floats < eff_scratch(f->B, f->N)) return TC_ERR_ARG;
    return dtype == TC_BF16 ? eff_bwd<bf16_t>(f, (hipStream_t)stream) : eff_bwd<f16_t>(f, (hipStream_t)stream);
}
