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

constexpr int C = 64, PT = C + 8, TPW = 256, NBLK = TPW / 32;      // channels, LDS row pitch (elements), tokens per wave, 32-token blocks
constexpr int F_M = 0, F_S = C, F_P = 2 * C, F_N = 2 * C + C * C;                       // forward partial of a wave
constexpr int B1_DCTX = 0, B1_DWR = C * C, B1_DWQ = 2 * C * C, B1_DBR = 3 * C * C, B1_DBQ = B1_DBR + C, B1_N = B1_DBQ + C;
constexpr int B2_DWK = 0, B2_DWV = C * C, B2_DBK = 2 * C * C, B2_DBV = B2_DBK + C, B2_DG = B2_DBV + C, B2_DB = B2_DG + C, B2_N = B2_DB + C;

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
struct Lane { int lane, l31, hh, gi, gq2; };
__device__ __forceinline__ Lane lane_of() { Lane L; L.lane = threadIdx.x & 63; L.l31 = L.lane & 31; L.hh = L.lane >> 5; L.gi = L.lane & 15; L.gq2 = (L.lane >> 4) & 1; return L; }

template <typename H> __device__ __forceinline__ void zero2(f32x16 (&a)[2]) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[cb][r] = 0.f;
}
// acc^T[o][tok] += sum_k W[o][k] X[tok][k]:  W rows = output channel (k contiguous), X rows = tokens; both 16-byte fragment reads
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
// out[i][j] += sum_tok X[tok][i] Y[tok][j] over the 32 tokens of a block (both tiles in LDS, rows = tokens): 2 x 2 tiles of 32 x 32
template <typename H> __device__ __forceinline__ void mm_tok(const bf16_t* X, const bf16_t* Y, const Lane& L, f32x16 (&acc)[2][2]) {
    using V8 = typename TcHalf<H>::v8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int row = 16 * ks + 8 * L.hh + (L.gi >> 2), col = 16 * L.gq2 + 4 * (L.gi & 3);
        V8 a[2], b[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            a[q] = ld_tr<V8>(X + row * PT + q * 32 + col, X + (row + 4) * PT + q * 32 + col);
            b[q] = ld_tr<V8>(Y + row * PT + q * 32 + col, Y + (row + 4) * PT + q * 32 + col);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = TcHalf<H>::mfma(a[i], b[j], acc[i][j]);
    }
}
// a D^T pair of tiles -> the wave's LDS tile [32 tokens][PT] in the storage type
template <typename H> __device__ __forceinline__ void put_T(bf16_t* X, const Lane& L, const f32x16 (&acc)[2]) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
            *reinterpret_cast<uint2*>(X + L.l31 * PT + cb * 32 + 8 * gq + 4 * L.hh) =
                make_uint2(pack2<H>(acc[cb][4 * gq], acc[cb][4 * gq + 1]), pack2<H>(acc[cb][4 * gq + 2], acc[cb][4 * gq + 3]));
}
// per-channel vector (bias, gamma, ...) in the D^T register order, from an fp32 LDS array
__device__ __forceinline__ void get_vecT(const float* v, const Lane& L, f32x16 (&o)[2]) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float4 q = *reinterpret_cast<const float4*>(v + cb * 32 + 8 * gq + 4 * L.hh);
            o[cb][4 * gq] = q.x; o[cb][4 * gq + 1] = q.y; o[cb][4 * gq + 2] = q.z; o[cb][4 * gq + 3] = q.w;
        }
}
__device__ __forceinline__ int chan_of(int cb, int r, int hh) { return cb * 32 + 8 * (r >> 2) + 4 * hh + (r & 3); }
// sum over the 32 token lanes of a half-wave of the 16 registers of one tile: afterwards lane l holds the total of register l & 15
__device__ __forceinline__ float fold16(const f32x16& t, int lane) {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = t[r] + __shfl_xor(t[r], 16, 64);
#pragma unroll
    for (int m = 8, n = 16; m >= 1; m >>= 1, n >>= 1) {
        const bool up = (lane & m) != 0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < n / 2) { const float keep = up ? v[j + n / 2] : v[j], send = up ? v[j] : v[j + n / 2]; v[j] = keep + __shfl_xor(send, m, 64); }
    }
    return v[0];
}
// the column sums of a D^T pair of tiles -> part[chan] (one writer per channel)
__device__ __forceinline__ void put_colsum(float* dst, const f32x16 (&a)[2], const Lane& L) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const float s = fold16(a[cb], L.lane);
        if (L.l31 < 16) dst[chan_of(cb, L.l31, L.hh)] = s;
    }
}
// a 2 x 2 set of D tiles (rows i, columns j) -> part[i * C + j]
__device__ __forceinline__ void put_mat(float* dst, const f32x16 (&a)[2][2], const Lane& L) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * L.hh) * C + j * 32 + L.l31] = a[i][j][r];
}

// 32 token rows of a [rows, ld] map -> LDS tile (zero rows beyond the image's tokens)
template <typename H> __device__ __forceinline__ void load_tile(bf16_t* X, const H* src, long long row0, int ld, int nvalid, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * i + (lane >> 3), cg = lane & 7;
        const uint4 v = r < nvalid ? *reinterpret_cast<const uint4*>(src + (row0 + r) * ld + cg * 8) : make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(X + r * PT + cg * 8) = v;
    }
}
template <typename H> __device__ __forceinline__ void store_tile(const bf16_t* X, H* dst, long long row0, int ld, int nvalid, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * i + (lane >> 3), cg = lane & 7;
        if (r < nvalid) *reinterpret_cast<uint4*>(dst + (row0 + r) * ld + cg * 8) = *reinterpret_cast<const uint4*>(X + r * PT + cg * 8);
    }
}
// weights [C][C] (storage type) -> LDS [C][PT]; vectors -> fp32 LDS
template <typename H> __device__ __forceinline__ void load_w(bf16_t* W, const void* src, int lane) {
    const H* s = reinterpret_cast<const H*>(src);
    for (int i = lane; i < C * (C / 8); i += 64) { const int r = i >> 3, cg = i & 7; *reinterpret_cast<uint4*>(W + r * PT + cg * 8) = *reinterpret_cast<const uint4*>(s + r * C + cg * 8); }
}
template <typename H> __device__ __forceinline__ void load_v(float* v, const void* src, int lane) { v[lane] = ldf<H>(reinterpret_cast<const H*>(src) + lane); }

// LayerNorm of the 32 token rows of an LDS tile, in place (lane = token, hh = channel half); returns xhat in registers when asked for
template <typename H, bool KEEP> __device__ __forceinline__ void ln_tile(bf16_t* X, const float* gam, const float* bet, float eps, const Lane& L, float& mean, float& rstd, float* xh) {
    float v[32];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint4 u = *reinterpret_cast<const uint4*>(X + L.l31 * PT + L.hh * 32 + q * 8);
        unpack2<H>(u.x, v[8 * q], v[8 * q + 1]); unpack2<H>(u.y, v[8 * q + 2], v[8 * q + 3]); unpack2<H>(u.z, v[8 * q + 4], v[8 * q + 5]); unpack2<H>(u.w, v[8 * q + 6], v[8 * q + 7]);
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 32; ++e) s += v[e];
    s += __shfl_xor(s, 32, 64);
    mean = s * (1.0f / C);
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
// (the wave's LDS accesses execute in order; the compiler must not move them across each other: one fence where a tile changes hands)
__device__ __forceinline__ void lds_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }

struct Where { int b, w; long long row0; int ntok; };
__device__ __forceinline__ Where where_am_i(const EffDev& p) {
    Where q;
    q.b = blockIdx.x / p.wpi; q.w = blockIdx.x - q.b * p.wpi;
    q.row0 = (long long)q.b * p.N + (long long)q.w * TPW;
    q.ntok = min(TPW, p.N - q.w * TPW);
    return q;
}

// ---------------------------------------------------------------------------------------------------------------- forward 1
template <typename H>
__global__ __launch_bounds__(64, 1) void effatt_kv_kernel(const EffDev p) {
    __shared__ __attribute__((aligned(16))) bf16_t wk[C * PT], wv[C * PT], xt[TPW * PT], et[32 * PT], vt[32 * PT];
    __shared__ float vec[4 * C];                                   // gamma, beta, bk, bv
    const Lane L = lane_of();
    const Where q = where_am_i(p);
    load_w<H>(wk, p.wk, L.lane); load_w<H>(wv, p.wv, L.lane);
    load_v<H>(vec, p.gamma, L.lane); load_v<H>(vec + C, p.beta, L.lane); load_v<H>(vec + 2 * C, p.bk, L.lane); load_v<H>(vec + 3 * C, p.bv, L.lane);
    const H* T = reinterpret_cast<const H*>(p.t);
    lds_fence();
    f32x16 bkv[2], bvv[2];
    get_vecT(vec + 2 * C, L, bkv); get_vecT(vec + 3 * C, L, bvv);
    // pass 1: LN of every block (kept in LDS), K^T of every block (kept in registers), the running column maximum
    f32x16 kk[NBLK][2];
    f32x16 mx[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx[cb][r] = -3.0e38f;
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
        bf16_t* X = xt + blk * 32 * PT;
        const int nv = q.ntok - blk * 32;
        load_tile<H>(X, T, q.row0 + blk * 32, p.ldt, nv, L.lane);
        lds_fence();
        float mean, rstd;
        ln_tile<H, false>(X, vec, vec + C, p.eps, L, mean, rstd, nullptr);
        lds_fence();
        kk[blk][0] = bkv[0]; kk[blk][1] = bkv[1];
        mm_w<H>(wk, X, L, kk[blk]);
        const bool ok = L.l31 < nv;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { if (!ok) kk[blk][cb][r] = -3.0e38f; mx[cb][r] = fmaxf(mx[cb][r], kk[blk][cb][r]); }
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) mx[cb][r] = fmaxf(mx[cb][r], __shfl_xor(mx[cb][r], m, 64));
    // pass 2: E = exp(K - m), S += E, P += E^T V
    f32x16 ss[2], pp[2][2];
    zero2<H>(ss);
#pragma unroll
    for (int i = 0; i < 2; ++i) zero2<H>(pp[i]);
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
        const bf16_t* X = xt + blk * 32 * PT;
        const int nv = q.ntok - blk * 32;
        const bool ok = L.l31 < nv;
        f32x16 vv[2];
        vv[0] = bvv[0]; vv[1] = bvv[1];
        mm_w<H>(wv, X, L, vv);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = ok ? __expf(kk[blk][cb][r] - mx[cb][r]) : 0.f;
                kk[blk][cb][r] = e; ss[cb][r] += e;
                if (!ok) vv[cb][r] = 0.f;
            }
        put_T<H>(et, L, kk[blk]);
        put_T<H>(vt, L, vv);
        lds_fence();
        mm_tok<H>(et, vt, L, pp);
        lds_fence();
    }
    float* PB = p.part + (long long)blockIdx.x * F_N;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const float s = fold16(ss[cb], L.lane);
        if (L.l31 < 16) { const int ch = chan_of(cb, L.l31, L.hh); PB[F_S + ch] = s; }
#pragma unroll
        for (int r = 0; r < 16; ++r) if (L.l31 == 0) PB[F_M + chan_of(cb, r, L.hh)] = mx[cb][r];
    }
    put_mat(PB + F_P, pp, L);
}

// ---------------------------------------------------------------------------------------------------------------- forward 2
// grid (B, C): thread = (channel c = blockIdx.y, column c'); ctx[b][c][c'], kstat[b] = (M[C], Z[C])
__global__ __launch_bounds__(64) void effatt_ctx_kernel(const EffDev p) {
    const int b = blockIdx.x, c = blockIdx.y, cc = threadIdx.x;
    const float* PB = p.part + (long long)b * p.wpi * F_N;
    float M = -3.0e38f;
    for (int w = 0; w < p.wpi; ++w) M = fmaxf(M, PB[(long long)w * F_N + F_M + c]);
    float Z = 0.f, acc = 0.f;
    for (int w = 0; w < p.wpi; ++w) {
        const float f = __expf(PB[(long long)w * F_N + F_M + c] - M);
        Z += f * PB[(long long)w * F_N + F_S + c];
        acc += f * PB[(long long)w * F_N + F_P + c * C + cc];
    }
    p.ctx[((long long)b * C + c) * C + cc] = acc / Z;
    if (cc == 0) { p.kstat[(long long)b * 2 * C + c] = M; p.kstat[(long long)b * 2 * C + C + c] = Z; }
}

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
// ctx[b] (fp32 [C][C]) -> LDS in the storage type
template <typename H> __device__ __forceinline__ void load_ctx(bf16_t* W, const float* src, int lane) {
    for (int i = lane; i < C * (C / 4); i += 64) {
        const int r = i >> 4, c4 = (i & 15) * 4;
        const float4 v = *reinterpret_cast<const float4*>(src + r * C + c4);
        *reinterpret_cast<uint2*>(W + r * PT + c4) = make_uint2(pack2<H>(v.x, v.y), pack2<H>(v.z, v.w));
    }
}

template <typename H>
__global__ __launch_bounds__(64, 1) void effatt_out_kernel(const EffDev p) {
    __shared__ __attribute__((aligned(16))) bf16_t wq[C * PT], wr[C * PT], cx[C * PT], xt[32 * PT], nt[32 * PT], st[32 * PT];
    __shared__ float vec[4 * C];                                   // gamma, beta, bq, br
    const Lane L = lane_of();
    const Where q = where_am_i(p);
    load_w<H>(wq, p.wq, L.lane); load_w<H>(wr, p.wr, L.lane); load_ctx<H>(cx, p.ctx + (long long)q.b * C * C, L.lane);
    load_v<H>(vec, p.gamma, L.lane); load_v<H>(vec + C, p.beta, L.lane); load_v<H>(vec + 2 * C, p.bq, L.lane); load_v<H>(vec + 3 * C, p.br, L.lane);
    const H* T = reinterpret_cast<const H*>(p.t);
    H* O = reinterpret_cast<H*>(p.out);
    lds_fence();
    f32x16 bqv[2], brv[2];
    get_vecT(vec + 2 * C, L, bqv); get_vecT(vec + 3 * C, L, brv);
    for (int blk = 0; blk < NBLK; ++blk) {
        const int nv = q.ntok - blk * 32;
        if (nv <= 0) break;
        load_tile<H>(xt, T, q.row0 + blk * 32, p.ldt, nv, L.lane);
        lds_fence();
#pragma unroll
        for (int i = 0; i < 4; ++i) {                              // n1 = LN(t) in its own tile: t itself is the residual
            const int r = 8 * i + (L.lane >> 3), cg = L.lane & 7;
            *reinterpret_cast<uint4*>(nt + r * PT + cg * 8) = *reinterpret_cast<const uint4*>(xt + r * PT + cg * 8);
        }
        lds_fence();
        float mean, rstd;
        ln_tile<H, false>(nt, vec, vec + C, p.eps, L, mean, rstd, nullptr);
        lds_fence();
        f32x16 a[2];
        a[0] = bqv[0]; a[1] = bqv[1];
        mm_w<H>(wq, nt, L, a);
        row_softmax(a);
        put_T<H>(st, L, a);                                        // Qsm
        lds_fence();
        zero2<H>(a);
        mm_wt<H>(cx, st, L, a);                                    // att^T[c'][tok] = sum_c ctx[c][c'] Qsm[tok][c]
        lds_fence();
        put_T<H>(st, L, a);                                        // att (every read of Qsm is done)
        lds_fence();
        a[0] = brv[0]; a[1] = brv[1];
        mm_w<H>(wr, st, L, a);                                     // out^T[o][tok] = sum_c' Wr[o][c'] att[tok][c']
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const uint2 tv = *reinterpret_cast<const uint2*>(xt + L.l31 * PT + cb * 32 + 8 * gq + 4 * L.hh);
                float t0, t1, t2, t3;
                unpack2<H>(tv.x, t0, t1); unpack2<H>(tv.y, t2, t3);
                a[cb][4 * gq] += t0; a[cb][4 * gq + 1] += t1; a[cb][4 * gq + 2] += t2; a[cb][4 * gq + 3] += t3;
            }
        lds_fence();
        put_T<H>(st, L, a);
        lds_fence();
        store_tile<H>(st, O, q.row0 + blk * 32, p.ldo, nv, L.lane);
        lds_fence();
    }
}

// ---------------------------------------------------------------------------------------------------------------- backward 1
template <typename H>
__global__ __launch_bounds__(64, 1) void effatt_bq_kernel(const EffDev p) {
    __shared__ __attribute__((aligned(16))) bf16_t wq[C * PT], wr[C * PT], cx[C * PT];
    __shared__ __attribute__((aligned(16))) bf16_t nt[32 * PT], qt[32 * PT], at[32 * PT], yt[32 * PT], dat[32 * PT], dqt[32 * PT];   // n1, Qsm, att, dout, d_att, dQ
    __shared__ float vec[3 * C];                                   // gamma, beta, bq
    const Lane L = lane_of();
    const Where q = where_am_i(p);
    load_w<H>(wq, p.wq, L.lane); load_w<H>(wr, p.wr, L.lane); load_ctx<H>(cx, p.ctx + (long long)q.b * C * C, L.lane);
    load_v<H>(vec, p.gamma, L.lane); load_v<H>(vec + C, p.beta, L.lane); load_v<H>(vec + 2 * C, p.bq, L.lane);
    const H* T = reinterpret_cast<const H*>(p.t);
    const H* DY = reinterpret_cast<const H*>(p.dout);
    H* G1 = reinterpret_cast<H*>(p.g1);
    lds_fence();
    f32x16 bqv[2];
    get_vecT(vec + 2 * C, L, bqv);
    f32x16 dctx[2][2], dwr[2][2], dwq[2][2], dbr[2], dbq[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { zero2<H>(dctx[i]); zero2<H>(dwr[i]); zero2<H>(dwq[i]); }
    zero2<H>(dbr); zero2<H>(dbq);
    for (int blk = 0; blk < NBLK; ++blk) {
        const int nv = q.ntok - blk * 32;
        if (nv <= 0) break;
        load_tile<H>(nt, T, q.row0 + blk * 32, p.ldt, nv, L.lane);
        load_tile<H>(yt, DY, q.row0 + blk * 32, p.lddo, nv, L.lane);          // (rows beyond the image: zeros -> no contribution)
        lds_fence();
        float mean, rstd;
        ln_tile<H, false>(nt, vec, vec + C, p.eps, L, mean, rstd, nullptr);
        lds_fence();
        f32x16 qs[2], a[2];
        qs[0] = bqv[0]; qs[1] = bqv[1];
        mm_w<H>(wq, nt, L, qs);
        row_softmax(qs);
        put_T<H>(qt, L, qs);                                       // Qsm
        lds_fence();
        zero2<H>(a);
        mm_wt<H>(cx, qt, L, a);
        put_T<H>(at, L, a);                                        // att = Qsm ctx (recomputed for dWr)
        zero2<H>(a);
        mm_wt<H>(wr, yt, L, a);                                    // d_att^T[c'][tok] = sum_o Wr[o][c'] dout[tok][o]
        put_T<H>(dat, L, a);
        lds_fence();
        zero2<H>(a);
        mm_w<H>(cx, dat, L, a);                                    // dQsm^T[c][tok] = sum_c' ctx[c][c'] d_att[tok][c']
        float dot = 0.f;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) dot += qs[cb][r] * a[cb][r];
        dot += __shfl_xor(dot, 32, 64);
        const bool ok = L.l31 < nv;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { a[cb][r] = ok ? qs[cb][r] * (a[cb][r] - dot) : 0.f; dbq[cb][r] += a[cb][r]; }       // dQ
        put_T<H>(dqt, L, a);
        lds_fence();
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {                       // dbr: column sums of dout
                const uint2 tv = *reinterpret_cast<const uint2*>(yt + L.l31 * PT + cb * 32 + 8 * gq + 4 * L.hh);
                float t0, t1, t2, t3;
                unpack2<H>(tv.x, t0, t1); unpack2<H>(tv.y, t2, t3);
                dbr[cb][4 * gq] += t0; dbr[cb][4 * gq + 1] += t1; dbr[cb][4 * gq + 2] += t2; dbr[cb][4 * gq + 3] += t3;
            }
        mm_tok<H>(qt, dat, L, dctx);                               // d_ctx[c][c'] += sum_tok Qsm[tok][c] d_att[tok][c']
        mm_tok<H>(yt, at, L, dwr);                                 // dWr[o][c']   += sum_tok dout[tok][o] att[tok][c']
        mm_tok<H>(dqt, nt, L, dwq);                                // dWq[c][cin]  += sum_tok dQ[tok][c] n1[tok][cin]
        zero2<H>(a);
        mm_wt<H>(wq, dqt, L, a);                                   // g1^T[cin][tok] = sum_c Wq[c][cin] dQ[tok][c]
        lds_fence();
        put_T<H>(at, L, a);                                        // (att's readers are done)
        lds_fence();
        store_tile<H>(at, G1, q.row0 + blk * 32, C, nv, L.lane);
        lds_fence();
    }
    float* PB = p.part + (long long)blockIdx.x * (B1_N + B2_N);
    put_mat(PB + B1_DCTX, dctx, L); put_mat(PB + B1_DWR, dwr, L); put_mat(PB + B1_DWQ, dwq, L);
    put_colsum(PB + B1_DBR, dbr, L); put_colsum(PB + B1_DBQ, dbq, L);
}

// grid (B, C): d_ctx[b][c][c'] = sum over the image's waves; r[b][c] = sum_c' d_ctx ctx
__global__ __launch_bounds__(64) void effatt_dctx_kernel(const EffDev p) {
    const int b = blockIdx.x, c = blockIdx.y, cc = threadIdx.x;
    const float* PB = p.part + (long long)b * p.wpi * (B1_N + B2_N);
    float acc = 0.f;
    for (int w = 0; w < p.wpi; ++w) acc += PB[(long long)w * (B1_N + B2_N) + B1_DCTX + c * C + cc];
    p.dctx[((long long)b * C + c) * C + cc] = acc;
    float r = acc * p.ctx[((long long)b * C + c) * C + cc];
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) r += __shfl_xor(r, m, 64);
    if (cc == 0) p.rsum[(long long)b * C + c] = r;
}

// ---------------------------------------------------------------------------------------------------------------- backward 2
template <typename H>
__global__ __launch_bounds__(64, 1) void effatt_bkv_kernel(const EffDev p) {
    __shared__ __attribute__((aligned(16))) bf16_t wk[C * PT], wv[C * PT], dcx[C * PT];
    __shared__ __attribute__((aligned(16))) bf16_t nt[32 * PT], kt[32 * PT], vt[32 * PT], dkt[32 * PT], dvt[32 * PT], gt[32 * PT];  // n1, Ksm, V, dK, dV, g1 / dt
    __shared__ float vec[7 * C];                                   // gamma, beta, bk, bv, M, 1 / Z, r
    const Lane L = lane_of();
    const Where q = where_am_i(p);
    load_w<H>(wk, p.wk, L.lane); load_w<H>(wv, p.wv, L.lane); load_ctx<H>(dcx, p.dctx + (long long)q.b * C * C, L.lane);
    load_v<H>(vec, p.gamma, L.lane); load_v<H>(vec + C, p.beta, L.lane); load_v<H>(vec + 2 * C, p.bk, L.lane); load_v<H>(vec + 3 * C, p.bv, L.lane);
    vec[4 * C + L.lane] = p.kstat[(long long)q.b * 2 * C + L.lane];
    vec[5 * C + L.lane] = 1.0f / p.kstat[(long long)q.b * 2 * C + C + L.lane];
    vec[6 * C + L.lane] = p.rsum[(long long)q.b * C + L.lane];
    const H* T = reinterpret_cast<const H*>(p.t);
    const H* DY = reinterpret_cast<const H*>(p.dout);
    const H* G1 = reinterpret_cast<const H*>(p.g1);
    H* DT = reinterpret_cast<H*>(p.dt);
    lds_fence();
    f32x16 bkv[2], bvv[2], Mv[2], iZ[2], rv[2];
    get_vecT(vec + 2 * C, L, bkv); get_vecT(vec + 3 * C, L, bvv); get_vecT(vec + 4 * C, L, Mv); get_vecT(vec + 5 * C, L, iZ); get_vecT(vec + 6 * C, L, rv);
    f32x16 dwk[2][2], dwv[2][2], dbk[2], dbv[2], dgm[2], dbt[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { zero2<H>(dwk[i]); zero2<H>(dwv[i]); }
    zero2<H>(dbk); zero2<H>(dbv); zero2<H>(dgm); zero2<H>(dbt);
    for (int blk = 0; blk < NBLK; ++blk) {
        const int nv = q.ntok - blk * 32;
        if (nv <= 0) break;
        const bool ok = L.l31 < nv;
        load_tile<H>(nt, T, q.row0 + blk * 32, p.ldt, nv, L.lane);
        load_tile<H>(gt, G1, q.row0 + blk * 32, C, nv, L.lane);
        lds_fence();
        float mean, rstd, xh[32];
        ln_tile<H, true>(nt, vec, vec + C, p.eps, L, mean, rstd, xh);        // xh: this lane's 32 channels hh * 32 + 0..31 (row order)
        lds_fence();
        f32x16 ks[2], vv[2], a[2], b2[2];
        ks[0] = bkv[0]; ks[1] = bkv[1];
        mm_w<H>(wk, nt, L, ks);
        vv[0] = bvv[0]; vv[1] = bvv[1];
        mm_w<H>(wv, nt, L, vv);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) ks[cb][r] = __expf(ks[cb][r] - Mv[cb][r]) * iZ[cb][r];                                   // Ksm
        put_T<H>(kt, L, ks);
        put_T<H>(vt, L, vv);
        lds_fence();
        zero2<H>(a);
        mm_wt<H>(dcx, kt, L, a);                                   // dV^T[c'][tok]  = sum_c d_ctx[c][c'] Ksm[tok][c]
        zero2<H>(b2);
        mm_w<H>(dcx, vt, L, b2);                                   // dKsm^T[c][tok] = sum_c' d_ctx[c][c'] V[tok][c']
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                b2[cb][r] = ok ? ks[cb][r] * (b2[cb][r] - rv[cb][r]) : 0.f;                                                       // dK
                if (!ok) a[cb][r] = 0.f;
                dbk[cb][r] += b2[cb][r]; dbv[cb][r] += a[cb][r];
            }
        put_T<H>(dkt, L, b2);
        put_T<H>(dvt, L, a);
        lds_fence();
        mm_tok<H>(dkt, nt, L, dwk);                                // dWk[c][cin] += sum_tok dK[tok][c] n1[tok][cin]
        mm_tok<H>(dvt, nt, L, dwv);
        zero2<H>(a);
        mm_wt<H>(wk, dkt, L, a);                                   // d_n1^T[cin][tok] = g1 + sum_c Wk[c][cin] dK[tok][c] + sum_c' Wv[c'][cin] dV[tok][c']
        mm_wt<H>(wv, dvt, L, a);
        lds_fence();
        put_T<H>(kt, L, a);                                        // (Ksm's readers are done): the two GEMM terms, token rows
        lds_fence();
        // LayerNorm backward in row order: lane = token, its channels hh * 32 + 0..31
        float dn[32], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const uint4 u = *reinterpret_cast<const uint4*>(kt + L.l31 * PT + L.hh * 32 + qq * 8);
            const uint4 g = *reinterpret_cast<const uint4*>(gt + L.l31 * PT + L.hh * 32 + qq * 8);
            float x[8], y[8];
            unpack2<H>(u.x, x[0], x[1]); unpack2<H>(u.y, x[2], x[3]); unpack2<H>(u.z, x[4], x[5]); unpack2<H>(u.w, x[6], x[7]);
            unpack2<H>(g.x, y[0], y[1]); unpack2<H>(g.y, y[2], y[3]); unpack2<H>(g.z, y[4], y[5]); unpack2<H>(g.w, y[6], y[7]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ch = L.hh * 32 + qq * 8 + e;
                const float d = ok ? x[e] + y[e] : 0.f;             // d_n1
                const float dg = d * vec[ch];
                dn[qq * 8 + e] = dg;
                s1 += dg; s2 += dg * xh[qq * 8 + e];
                // (dgamma / dbeta need column sums: gathered below in the D^T register order from a tile)
            }
        }
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        const float k1 = s1 * (1.0f / C), k2 = s2 * (1.0f / C);
        lds_fence();
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            // dgamma += d_n1 xhat, dbeta += d_n1: park (d_n1 xhat) in dkt and d_n1 in dvt (their readers are done), summed by columns below
            const uint4 u = *reinterpret_cast<const uint4*>(kt + L.l31 * PT + L.hh * 32 + qq * 8);
            const uint4 g = *reinterpret_cast<const uint4*>(gt + L.l31 * PT + L.hh * 32 + qq * 8);
            float x[8], y[8], o[8], w1[8], w2[8];
            unpack2<H>(u.x, x[0], x[1]); unpack2<H>(u.y, x[2], x[3]); unpack2<H>(u.z, x[4], x[5]); unpack2<H>(u.w, x[6], x[7]);
            unpack2<H>(g.x, y[0], y[1]); unpack2<H>(g.y, y[2], y[3]); unpack2<H>(g.z, y[4], y[5]); unpack2<H>(g.w, y[6], y[7]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = ok ? x[e] + y[e] : 0.f;
                w1[e] = d * xh[qq * 8 + e]; w2[e] = d;
                o[e] = ok ? rstd * (dn[qq * 8 + e] - k1 - xh[qq * 8 + e] * k2) : 0.f;
            }
            *reinterpret_cast<uint4*>(dkt + L.l31 * PT + L.hh * 32 + qq * 8) = make_uint4(pack2<H>(w1[0], w1[1]), pack2<H>(w1[2], w1[3]), pack2<H>(w1[4], w1[5]), pack2<H>(w1[6], w1[7]));
            *reinterpret_cast<uint4*>(dvt + L.l31 * PT + L.hh * 32 + qq * 8) = make_uint4(pack2<H>(w2[0], w2[1]), pack2<H>(w2[2], w2[3]), pack2<H>(w2[4], w2[5]), pack2<H>(w2[6], w2[7]));
            *reinterpret_cast<uint4*>(vt + L.l31 * PT + L.hh * 32 + qq * 8) = make_uint4(pack2<H>(o[0], o[1]), pack2<H>(o[2], o[3]), pack2<H>(o[4], o[5]), pack2<H>(o[6], o[7]));
        }
        lds_fence();
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const uint2 u1 = *reinterpret_cast<const uint2*>(dkt + L.l31 * PT + cb * 32 + 8 * gq + 4 * L.hh);
                const uint2 u2 = *reinterpret_cast<const uint2*>(dvt + L.l31 * PT + cb * 32 + 8 * gq + 4 * L.hh);
                float t0, t1, t2, t3;
                unpack2<H>(u1.x, t0, t1); unpack2<H>(u1.y, t2, t3);
                dgm[cb][4 * gq] += t0; dgm[cb][4 * gq + 1] += t1; dgm[cb][4 * gq + 2] += t2; dgm[cb][4 * gq + 3] += t3;
                unpack2<H>(u2.x, t0, t1); unpack2<H>(u2.y, t2, t3);
                dbt[cb][4 * gq] += t0; dbt[cb][4 * gq + 1] += t1; dbt[cb][4 * gq + 2] += t2; dbt[cb][4 * gq + 3] += t3;
            }
        // dt = LayerNorm backward + the residual's gradient (dout) [+ what dt holds]
        load_tile<H>(kt, DY, q.row0 + blk * 32, p.lddo, nv, L.lane);
        if (p.acc_dt) load_tile<H>(gt, DT, q.row0 + blk * 32, p.lddt, nv, L.lane);
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
            if (r < nv) *reinterpret_cast<uint4*>(DT + (q.row0 + blk * 32 + r) * p.lddt + cg * 8) =
                make_uint4(pack2<H>(x[0], x[1]), pack2<H>(x[2], x[3]), pack2<H>(x[4], x[5]), pack2<H>(x[6], x[7]));
        }
        lds_fence();
    }
    float* PB = p.part + (long long)blockIdx.x * (B1_N + B2_N) + B1_N;
    put_mat(PB + B2_DWK, dwk, L); put_mat(PB + B2_DWV, dwv, L);
    put_colsum(PB + B2_DBK, dbk, L); put_colsum(PB + B2_DBV, dbv, L); put_colsum(PB + B2_DG, dgm, L); put_colsum(PB + B2_DB, dbt, L);
}

// ---------------------------------------------------------------------------------------------------------------- fold
struct FoldDev { const float* part; float* dst[12]; int off[13]; int nwv, nper; };
// grid (nper / 64), 256 threads = 16 float4 columns x 16 slices of the wave list
__global__ __launch_bounds__(256) void effatt_fold_kernel(const FoldDev p) {
    __shared__ float4 sh[16][17];
    const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int q = blockIdx.x * 16 + e;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q * 4 < p.nper) {
        const float* base = p.part + (long long)q * 4;
        for (int w = sl; w < p.nwv; w += 16) {
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
        for (int k = 1; k < 12; ++k) seg += idx >= p.off[k];
        float* d = p.dst[seg];
        if (d) { d += idx - p.off[seg]; d[0] += s.x; d[1] += s.y; d[2] += s.z; d[3] += s.w; }
    }
}

bool eff_ok(const TcEffAtt* f, bool bwd) {
    if (!f || !f->t || !f->gamma || !f->beta || !f->wk || !f->bk || !f->wq || !f->bq || !f->wv || !f->bv || !f->wr || !f->br || !f->ctx || !f->kstat || !f->part) return false;
    if (f->C != C || f->B < 1 || f->N < 1 || (f->ldt & 7) || ((uintptr_t)f->t & 15) || ((uintptr_t)f->part & 15) || ((uintptr_t)f->ctx & 15)) return false;
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
    hipLaunchKernelGGL((effatt_kv_kernel<H>), dim3(nwv), dim3(64), 0, s, p);
    if (tc_launch_status() != TC_OK) return TC_ERR_LAUNCH;
    hipLaunchKernelGGL(effatt_ctx_kernel, dim3(f->B, C), dim3(64), 0, s, p);
    if (tc_launch_status() != TC_OK) return TC_ERR_LAUNCH;
    hipLaunchKernelGGL((effatt_out_kernel<H>), dim3(nwv), dim3(64), 0, s, p);
    return tc_launch_status();
}
template <typename H> int eff_bwd(const TcEffAtt* f, hipStream_t s) {
    const EffDev p = eff_dev(f);
    const int nwv = f->B * p.wpi;
    hipLaunchKernelGGL((effatt_bq_kernel<H>), dim3(nwv), dim3(64), 0, s, p);
    if (tc_launch_status() != TC_OK) return TC_ERR_LAUNCH;
    hipLaunchKernelGGL(effatt_dctx_kernel, dim3(f->B, C), dim3(64), 0, s, p);
    if (tc_launch_status() != TC_OK) return TC_ERR_LAUNCH;
    hipLaunchKernelGGL((effatt_bkv_kernel<H>), dim3(nwv), dim3(64), 0, s, p);
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
