// Fused core of FactorAtt_ConvRelPosEnc (MSTr.py:864-877) for one (image, head) per workgroup:
//   ksm = softmax(k, dim=N)            (per channel, over the N tokens of the image)
//   ctx = ksm^T v                      [Ch, Ch]
//   o   = scale * (q ctx) + q (.) convv           convv = the conv-relative-position term crpe(v) (computed by the depthwise kernels)
// and its backward.  N <= ~1024 tokens and Ch = 8/16/40 channels per head: everything a head needs fits LDS in fp32, so the five
// launches of the unfused forward (column-softmax statistics + apply, two tiny batched GEMMs with 16-64x padded tiles, the fma
// epilogue) and the seven of its backward collapse into one launch each, and no intermediate is rounded to the storage type.
// Backward identities used:  dctx = scale q^T do ;  dq = scale do ctx^T + do (.) convv ;  dconvv = do (.) q ;  dv = ksm dctx ;
//   dksm = v dctx^T ;  dk = ksm (.) (dksm - t),  t[i] = sum_n ksm[n,i] dksm[n,i] = sum_j ctx[i,j] dctx[i,j]  (no extra pass over N).
#include "tc_common.h"

namespace {

constexpr int FA_MAXCH = 64;

template <typename T> struct V16;
template <> struct V16<float> { static constexpr int N = 4; };
template <> struct V16<bf16_t> { static constexpr int N = 8; };
template <> struct V16<f16_t> { static constexpr int N = 8; };
template <typename T> __device__ __forceinline__ void unpack16(const uint4& r, float* o);
template <> __device__ __forceinline__ void unpack16<float>(const uint4& r, float* o) {
    o[0] = __uint_as_float(r.x); o[1] = __uint_as_float(r.y); o[2] = __uint_as_float(r.z); o[3] = __uint_as_float(r.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const uint4& r, float* o) {
    o[0] = __uint_as_float(r.x << 16); o[1] = __uint_as_float(r.x & 0xffff0000u); o[2] = __uint_as_float(r.y << 16);
    o[3] = __uint_as_float(r.y & 0xffff0000u); o[4] = __uint_as_float(r.z << 16); o[5] = __uint_as_float(r.z & 0xffff0000u);
    o[6] = __uint_as_float(r.w << 16); o[7] = __uint_as_float(r.w & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack16<f16_t>(const uint4& r, float* o) {
    unpack2<f16_t>(r.x, o[0], o[1]); unpack2<f16_t>(r.y, o[2], o[3]); unpack2<f16_t>(r.z, o[4], o[5]); unpack2<f16_t>(r.w, o[6], o[7]);
}
template <typename T> __device__ __forceinline__ uint4 pack16(const float* o);
template <> __device__ __forceinline__ uint4 pack16<float>(const float* o) {
    return make_uint4(__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3]));
}
template <> __device__ __forceinline__ uint4 pack16<bf16_t>(const float* o) {
    return make_uint4(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7]));
}
template <> __device__ __forceinline__ uint4 pack16<f16_t>(const float* o) {
    return make_uint4(pack2h(o[0], o[1]), pack2h(o[2], o[3]), pack2h(o[4], o[5]), pack2h(o[6], o[7]));
}
// head tile [N][Ch] of a row-strided matrix -> fp32 LDS, 16-byte loads, four in flight per thread (a scalar copy loop keeps ONE
// load in flight and made the first version of this kernel 50 us for 6272 elements)
template <typename T>
__device__ __forceinline__ void fa_load_tile(float* dst, const T* src, int ld, int N, int Ch) {
    constexpr int VEC = V16<T>::N;
    const int nv = Ch / VEC, total = N * nv;
#pragma unroll 4
    for (int i = threadIdx.x; i < total; i += 256) {
        const int n = i / nv, cv = i - n * nv;
        const uint4 r = *reinterpret_cast<const uint4*>(src + (long long)n * ld + cv * VEC);
        unpack16<T>(r, dst + n * Ch + cv * VEC);
    }
}

// column statistics of k over the tile in LDS: e[n,c] <- exp(k - max_c), inv[c] = 1 / sum_n e[n,c]
__device__ __forceinline__ void fa_softmax_cols(float* e, float* cmax, float* cinv, float* red, int N, int Ch) {
    const int tid = threadIdx.x, R = 256 / Ch, c = tid % Ch, r = tid / Ch;
    float m = -INFINITY;
    if (r < R) for (int n = r; n < N; n += R) m = fmaxf(m, e[n * Ch + c]);
    if (r < R) red[r * Ch + c] = m;
    __syncthreads();
    if (tid < Ch) { float v = red[tid]; for (int i = 1; i < R; ++i) v = fmaxf(v, red[i * Ch + tid]); cmax[tid] = v; }
    __syncthreads();
    float s = 0.f;
    if (r < R) {
        const float mc = cmax[c];
        for (int n = r; n < N; n += R) { const float v = __expf(e[n * Ch + c] - mc); e[n * Ch + c] = v; s += v; }
        red[r * Ch + c] = s;
    }
    __syncthreads();
    if (tid < Ch) { float v = 0.f; for (int i = 0; i < R; ++i) v += red[i * Ch + tid]; cinv[tid] = 1.0f / v; }
    __syncthreads();
}

// the same head tile kept in its STORAGE type (bf16 operands lose nothing that way and q / v / do take half the LDS: two
// workgroups per CU at N = 784 instead of one)
template <typename T>
__device__ __forceinline__ void fa_load_tile_raw(T* dst, const T* src, int ld, int N, int Ch) {
    constexpr int VEC = V16<T>::N;
    const int nv = Ch / VEC, total = N * nv;
#pragma unroll 4
    for (int i = threadIdx.x; i < total; i += 256) {
        const int n = i / nv, cv = i - n * nv;
        *reinterpret_cast<uint4*>(dst + n * Ch + cv * VEC) = *reinterpret_cast<const uint4*>(src + (long long)n * ld + cv * VEC);
    }
}
__device__ __forceinline__ float fa_get(const float* p, int i) { return p[i]; }
__device__ __forceinline__ float fa_get(const bf16_t* p, int i) { return bf2f(p[i]); }
__device__ __forceinline__ float fa_get(const f16_t* p, int i) { return h2f(p[i]); }
__device__ __forceinline__ void fa_get8(const float* p, float* o) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void fa_get8(const bf16_t* p, float* o) { unpack16<bf16_t>(*reinterpret_cast<const uint4*>(p), o); }
__device__ __forceinline__ void fa_get8(const f16_t* p, float* o) { unpack16<f16_t>(*reinterpret_cast<const uint4*>(p), o); }

// out[i][j] = sum_n a[n,i] * b[n,j]   (Ch x Ch, Ch a multiple of 8).  A thread owns (row i, 8 consecutive j) for the rows
// n = rl, rl + RL, ...: one read of a[n,i], two 16-byte reads of b[n, j0..j0+7], 8 FMAs; the RL row lanes of an output sit next to
// each other in a wave and are folded with shuffles.  (One thread per output walking all N rows took ~10 us per product.)
template <typename TA, typename TB>
__device__ __forceinline__ void fa_gram(float* out, const TA* a, const TB* b, int N, int Ch) {
    const int tid = threadIdx.x, jb = Ch >> 3, combos = Ch * jb;
    int RL = 256 / combos;                                  // 32 (Ch 8), 8 (Ch 16), 1 (Ch 40)
    RL = RL >= 32 ? 32 : (RL >= 16 ? 16 : (RL >= 8 ? 8 : (RL >= 4 ? 4 : (RL >= 2 ? 2 : 1))));
    const int combo = tid / RL, rl = tid - combo * RL;
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    for (int cb = combo; cb < combos; cb += 256 / RL) {     // Ch = 40: 200 combos, one pass; generic for larger Ch
        const int i = cb / jb, j0 = (cb - i * jb) * 8;
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = 0.f;
        for (int n = rl; n < N; n += RL) {
            const float ai = fa_get(a, n * Ch + i);
            float bv[8];
            fa_get8(b + n * Ch + j0, bv);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += ai * bv[u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {                           // (RL is 1 .. 32, wave-uniform: the in-row steps through DPP)
            float v = acc[u];
            if (RL >= 2) v += tc_dpp<0xB1>(v);
            if (RL >= 4) v += tc_dpp<0x4E>(v);
            if (RL >= 8) v += tc_dpp<0x141>(v);
            if (RL >= 16) v += tc_dpp<0x140>(v);
            if (RL >= 32) v += __shfl_xor(v, 16, 64);
            acc[u] = v;
        }
        if (rl == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) out[i * Ch + j0 + u] = acc[u];
        }
    }
    __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(256) void factor_att_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, int ld,
                                                             const T* __restrict__ convv, int ldc, T* __restrict__ o, int ldo,
                                                             float* __restrict__ stats, int N, int Ch, int heads, float scale) {
    constexpr int VEC = V16<T>::N;
    extern __shared__ float sm[];
    float* e = sm;                       // [N][Ch]  k, then exp(k - max)
    float* ctx = e + N * Ch;             // [Ch][Ch]
    float* cmax = ctx + Ch * Ch;
    float* cinv = cmax + Ch;
    float* red = cinv + Ch;              // [256/Ch][Ch] <= 256
    T* vs = reinterpret_cast<T*>(red + 256);      // [N][Ch] storage type
    T* qs = vs + N * Ch;
    // the heads of one image read neighbouring 16..128-byte pieces of the same qkv rows: numbered onto ONE XCD (index l runs on XCD l % 8)
    const int blk = (gridDim.x & 7) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
    const int bt = blk / heads, hd = blk - bt * heads, tid = threadIdx.x;
    const long long row0 = (long long)bt * N;
    const int col0 = hd * Ch;
    fa_load_tile<T>(e, k + row0 * ld + col0, ld, N, Ch);
    fa_load_tile_raw<T>(vs, v + row0 * ld + col0, ld, N, Ch);
    fa_load_tile_raw<T>(qs, q + row0 * ld + col0, ld, N, Ch);
    __syncthreads();
    fa_softmax_cols(e, cmax, cinv, red, N, Ch);
    fa_gram(ctx, e, vs, N, Ch);
    for (int i = tid; i < Ch * Ch; i += 256) ctx[i] *= cinv[i / Ch];          // rows of ctx carry the softmax normaliser
    if (tid < Ch) { stats[((long long)blk * 2) * Ch + tid] = cmax[tid]; stats[((long long)blk * 2 + 1) * Ch + tid] = cinv[tid]; }
    __syncthreads();
    const int nv = Ch / VEC;
#pragma unroll 2
    for (int i = tid; i < N * nv; i += 256) {
        const int n = i / nv, j0 = (i - n * nv) * VEC;
        float cv[VEC], acc[VEC];
        unpack16<T>(*reinterpret_cast<const uint4*>(convv + (row0 + n) * ldc + col0 + j0), cv);
#pragma unroll
        for (int u = 0; u < VEC; ++u) acc[u] = 0.f;
        const T* qr = qs + n * Ch;
        for (int c8 = 0; c8 < Ch; c8 += 8) {                    // the row's q values eight at a time (one or two 16-byte LDS reads instead of eight scalar ones)
            float q8[8];
            fa_get8(qr + c8, q8);
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) {
                const float* cr = ctx + (c8 + cc) * Ch + j0;
                float cx[VEC];
#pragma unroll
                for (int u = 0; u < VEC; u += 4) *reinterpret_cast<float4*>(cx + u) = *reinterpret_cast<const float4*>(cr + u);
#pragma unroll
                for (int u = 0; u < VEC; ++u) acc[u] += q8[cc] * cx[u];
            }
        }
#pragma unroll
        for (int u = 0; u < VEC; ++u) acc[u] = scale * acc[u] + fa_get(qr, j0 + u) * cv[u];
        *reinterpret_cast<uint4*>(o + (row0 + n) * ldo + col0 + j0) = pack16<T>(acc);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void factor_att_bwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, int ld,
                                                             const T* __restrict__ convv, int ldc, const T* __restrict__ go, int ldgo,
                                                             const float* __restrict__ stats, T* __restrict__ dq, T* __restrict__ dk,
                                                             T* __restrict__ dv, int ldd, int acc_q, int acc_k, int acc_v,
                                                             T* __restrict__ dconvv, int lddc, int N, int Ch, int heads, float scale) {
    constexpr int VEC = V16<T>::N;
    extern __shared__ float sm[];
    float* e = sm;                       // softmax(k) (normalised), fp32
    float* ctx = e + N * Ch;
    float* dctx = ctx + Ch * Ch;
    float* tcol = dctx + Ch * Ch;        // [Ch] (+ padding to 16 bytes)
    float* ctxT = tcol + ((Ch + 3) & ~3);                      // ctx^T, dctx^T: the per-token loop reads VEC consecutive floats of a row
    float* dctxT = ctxT + Ch * Ch;
    T* vs = reinterpret_cast<T*>(dctxT + Ch * Ch);             // v, q, do tiles in the storage type
    T* qs = vs + N * Ch;
    T* gs = qs + N * Ch;
    // the heads of one image read neighbouring 16..128-byte pieces of the same qkv rows: numbered onto ONE XCD (index l runs on XCD l % 8)
    const int blk = (gridDim.x & 7) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
    const int bt = blk / heads, hd = blk - bt * heads, tid = threadIdx.x;
    const long long row0 = (long long)bt * N;
    const int col0 = hd * Ch;
    const float* cmax = stats + ((long long)blk * 2) * Ch;
    const float* cinv = cmax + Ch;
    fa_load_tile<T>(e, k + row0 * ld + col0, ld, N, Ch);
    fa_load_tile_raw<T>(vs, v + row0 * ld + col0, ld, N, Ch);
    fa_load_tile_raw<T>(qs, q + row0 * ld + col0, ld, N, Ch);
    fa_load_tile_raw<T>(gs, go + row0 * ldgo + col0, ldgo, N, Ch);
    __syncthreads();
    if (256 % Ch == 0) {                                        // (Ch = 8, 16, 32, 64: a thread stays in its column -- no division per element)
        const int c = tid % Ch;
        const float mc = cmax[c], ic = cinv[c];
        for (int i = tid; i < N * Ch; i += 256) e[i] = __expf(e[i] - mc) * ic;
    } else {
        for (int i = tid; i < N * Ch; i += 256) { const int c = i % Ch; e[i] = __expf(e[i] - cmax[c]) * cinv[c]; }
    }
    __syncthreads();
    fa_gram(ctx, e, vs, N, Ch);
    fa_gram(dctx, qs, gs, N, Ch);
    for (int i = tid; i < Ch * Ch; i += 256) dctx[i] *= scale;
    __syncthreads();
    if (tid < Ch) { float t = 0.f; for (int j = 0; j < Ch; ++j) t += ctx[tid * Ch + j] * dctx[tid * Ch + j]; tcol[tid] = t; }
    for (int i = tid; i < Ch * Ch; i += 256) { const int c = i / Ch, j = i - c * Ch; ctxT[j * Ch + c] = ctx[i]; dctxT[j * Ch + c] = dctx[i]; }
    __syncthreads();
    const int nv = Ch / VEC;
    for (int i = tid; i < N * nv; i += 256) {
        const int n = i / nv, c0 = (i - n * nv) * VEC;
        const long long r = row0 + n;
        const T* gr = gs + n * Ch;
        const T* vr = vs + n * Ch;
        const float* er = e + n * Ch;
        float a_q[VEC], a_ks[VEC], a_v[VEC], cv[VEC], oc[VEC];
#pragma unroll
        for (int u = 0; u < VEC; ++u) a_q[u] = a_ks[u] = a_v[u] = 0.f;
        for (int j8 = 0; j8 < Ch; j8 += 8) {                    // the row's do / v / ksm values eight at a time (16-byte LDS reads; Ch % 8 == 0)
            float g8[8], v8[8], e8[8];
            fa_get8(gr + j8, g8); fa_get8(vr + j8, v8); fa_get8(er + j8, e8);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j = j8 + jj;
                float cq[VEC], ck[VEC], cw[VEC];              // 16-byte LDS reads (the first version read 3 x VEC scalars per j, two of them strided)
#pragma unroll
                for (int u = 0; u < VEC; u += 4) {
                    *reinterpret_cast<float4*>(cq + u) = *reinterpret_cast<const float4*>(ctxT + j * Ch + c0 + u);
                    *reinterpret_cast<float4*>(ck + u) = *reinterpret_cast<const float4*>(dctxT + j * Ch + c0 + u);
                    *reinterpret_cast<float4*>(cw + u) = *reinterpret_cast<const float4*>(dctx + j * Ch + c0 + u);
                }
#pragma unroll
                for (int u = 0; u < VEC; ++u) {
                    a_q[u] += g8[jj] * cq[u];                 // dq[n,c]  += scale * sum_j do[n,j] ctx[c,j]
                    a_ks[u] += v8[jj] * ck[u];                // dksm[n,c] = sum_j v[n,j] dctx[c,j]
                    a_v[u] += e8[jj] * cw[u];                 // dv[n,c]   = sum_i ksm[n,i] dctx[i,c]
                }
            }
        }
        unpack16<T>(*reinterpret_cast<const uint4*>(convv + r * ldc + col0 + c0), cv);
        T* pq = dq + r * ldd + col0 + c0; T* pk = dk + r * ldd + col0 + c0; T* pv = dv + r * ldd + col0 + c0;
        float old[VEC];
#pragma unroll
        for (int u = 0; u < VEC; ++u) {
            const float gc = fa_get(gr, c0 + u);
            oc[u] = gc * fa_get(qs, n * Ch + c0 + u); a_q[u] = scale * a_q[u] + gc * cv[u]; a_ks[u] = er[c0 + u] * (a_ks[u] - tcol[c0 + u]);
        }
        if (acc_q) { unpack16<T>(*reinterpret_cast<const uint4*>(pq), old);
#pragma unroll
            for (int u = 0; u < VEC; ++u) a_q[u] += old[u]; }
        if (acc_k) { unpack16<T>(*reinterpret_cast<const uint4*>(pk), old);
#pragma unroll
            for (int u = 0; u < VEC; ++u) a_ks[u] += old[u]; }
        if (acc_v) { unpack16<T>(*reinterpret_cast<const uint4*>(pv), old);
#pragma unroll
            for (int u = 0; u < VEC; ++u) a_v[u] += old[u]; }
        *reinterpret_cast<uint4*>(pq) = pack16<T>(a_q);
        *reinterpret_cast<uint4*>(pk) = pack16<T>(a_ks);
        *reinterpret_cast<uint4*>(pv) = pack16<T>(a_v);
        *reinterpret_cast<uint4*>(dconvv + r * lddc + col0 + c0) = pack16<T>(oc);
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The attention half of an MHCABlock in ONE launch (MSTr.py:852-886 + 801-823; 16-bit storage): for one (image, head) per workgroup
//   q | k | v = xn Wqkv_h^T + b_h          by MFMA (D^T tiles: lane = token, registers = 16 of the head's 3 Ch output channels; the
//                                          xn rows and the head's 3 Ch weight rows go from global memory straight into operand registers)
//   convv     = crpe_h(v)                  the head's 3x3 / 5x5 / 7x7 depthwise window over the token grid, from the v tile in LDS
//   o         = scale q (softmax_N(k)^T v) + q (.) convv      exactly as factor_att_fwd_kernel
// replacing the qkv GEMM, the three-window depthwise launch and the factor-attention launch (33-43 us of 5-20 us launches per block)
// by one, with q, k, v, convv never read back from memory in the forward pass.  q | k | v, convv and the softmax statistics are still
// stored (the backward kernels read them), in the storage type and rounded BEFORE they are used, so that the forward consumes the
// same values the backward will see.
// a 16-bit element as a zero-extended 32-bit load (two 16-bit loads into the halves of ONE register order the second behind the first:
// a wait between loads that were meant to be in flight together), converted when it is used
template <typename T> __device__ __forceinline__ unsigned ld_raw16(const T* p) { return (unsigned)*reinterpret_cast<const unsigned short*>(p); }
template <typename T> __device__ __forceinline__ float cvt_raw16(unsigned r);
template <> __device__ __forceinline__ float cvt_raw16<bf16_t>(unsigned r) { return __uint_as_float(r << 16); }
template <> __device__ __forceinline__ float cvt_raw16<f16_t>(unsigned r) { f16_t h; h.v = (unsigned short)r; return h2f(h); }

struct MhcaAttDev {
    const void *xn, *Wqkv, *bqkv, *cw[3], *cb[3];
    void *qkv, *convv, *o;
    float* stats;
    long long gs;
    int ldx, ldq, ldc, ldo, B, N, H, W;
    float scale;
};

// One head's depthwise K x K window over the token grid, v tile [N][CH] (storage type) -> convv tile.  A thread owns XB neighbouring
// pixels of a row and 8 channels: per filter row it reads XB + K - 1 input vectors and the K tap vectors once and feeds XB * K * 8
// FMAs from them (one vector per tap and pixel cost 3 LDS reads per 8 FMAs and made the 7 x 7 heads the launch's critical path);
// out-of-range taps read a clamped address and are zeroed by a select, so the loop body has no branches.
template <int K, int CH, int XB, typename T>
__device__ __forceinline__ void mhca_conv_tile(T* cvs, const T* vs, const float* wcv, const float* bcv, int H, int W) {
    constexpr int P = K / 2, NV = CH / 8, NIN = XB + K - 1;
    const int nxb = (W + XB - 1) / XB, items = H * nxb * NV;
    const SDiv dnxb = sdiv_make(nxb);                                 // (tc_common.h: a run-time divisor of small indices)
    for (int i = threadIdx.x; i < items; i += 256) {
        const int cvi = i % NV, t = i / NV, y = sdiv(t, dnxb), xb = smod(t, y, dnxb), x0 = xb * XB, c0 = cvi * 8;
        float acc[XB][8];
#pragma unroll
        for (int o = 0; o < XB; ++o)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[o][u] = bcv[c0 + u];
#pragma unroll 1
        for (int ky = 0; ky < K; ++ky) {
            const int yy = y + ky - P;
            const bool rv = (unsigned)yy < (unsigned)H;
            const int yc = rv ? yy : y;
            float vin[NIN][8];
#pragma unroll
            for (int j = 0; j < NIN; ++j) {
                const int xx = x0 + j - P;
                const bool ok = rv && (unsigned)xx < (unsigned)W;
                const int xc = (unsigned)xx < (unsigned)W ? xx : x0;
                uint4 r = *reinterpret_cast<const uint4*>(vs + tc_mul24(tc_mad24(yc, W, xc), CH) + c0);
                if (!ok) r = make_uint4(0u, 0u, 0u, 0u);
                unpack16<T>(r, vin[j]);
            }
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                float w8[8];
                fa_get8(wcv + (ky * K + kx) * CH + c0, w8);
#pragma unroll
                for (int o = 0; o < XB; ++o)
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc[o][u] = fmaf(vin[o + kx][u], w8[u], acc[o][u]);
            }
        }
#pragma unroll
        for (int o = 0; o < XB; ++o)
            if (x0 + o < W) *reinterpret_cast<uint4*>(cvs + tc_mul24(tc_mad24(y, W, x0 + o), CH) + c0) = pack16<T>(acc[o]);
    }
}
template <int K, int CH, typename T>
__device__ __forceinline__ void mhca_conv_pick(T* cvs, const T* vs, const float* wcv, const float* bcv, int H, int W) {
    if (W >= 24) mhca_conv_tile<K, CH, 4, T>(cvs, vs, wcv, bcv, H, W);
    else if (W >= 12) mhca_conv_tile<K, CH, 2, T>(cvs, vs, wcv, bcv, H, W);
    else mhca_conv_tile<K, CH, 1, T>(cvs, vs, wcv, bcv, H, W);
}

template <typename T, int C>
__global__ __launch_bounds__(256, 2) void mhca_att_fwd_kernel(MhcaAttDev p) {
    constexpr int CH = C / 8, KS = C / 16, NQ = 3 * CH, NCOL = (NQ + 31) / 32, RSTEP = 4 / NCOL, NV = CH / 8;
    constexpr bool STAGE_X = C >= 320;
    constexpr int XLD = C + 8;
    static_assert(NCOL == 1 || NCOL == 2 || NCOL == 4, "column tiles of a head must divide the four waves");
    typedef typename TcHalf<T>::v8 v8;
    extern __shared__ float sm[];
    const int N = p.N, tid = threadIdx.x;
#ifdef TC_MHCA_TIMING
    long long tstamp[16];
#define MHCA_STAMP(i) tstamp[i] = clock64()
    MHCA_STAMP(0);
#else
#define MHCA_STAMP(i)
#endif
    float* e = sm;                        // [N][CH]  k (rounded to the storage type), then exp(k - max)
    float* ctx = e + N * CH;              // [CH][CH]
    float* cmax = ctx + CH * CH;
    float* cinv = cmax + CH;
    float* red = cinv + CH;               // 256
    float* wcv = red + 256;               // [49][CH] window taps of this head, channel fastest
    float* bcv = wcv + 49 * CH;           // [CH]
    T* vs = reinterpret_cast<T*>(bcv + CH);
    T* qs = vs + N * CH;
    T* cvs = qs + N * CH;                 // convv tile; until the projection is done the same bytes hold the xn staging tiles
    T* xs = STAGE_X ? cvs + N * CH : cvs; // STAGE_X: [N][XLD] xn rows of the image; else 4 wave-private [32][XLD] tiles (aliasing cvs)
    // Workgroup -> (image, head).  Index l runs on XCD l % 8 (each XCD has its own L2): the eight heads of an image, which read the same xn rows,
    // share an XCD; inside an XCD's share the 7 x 7 heads are dispatched first, then the 5 x 5, then the 3 x 3 ones -- the window makes a
    // 7 x 7 head ~10 k cycles longer than a 3 x 3 one, and the CUs that take a second workgroup then pair a long one with a short one.
    int bt, hd;
    if (((gridDim.x >> 3) & 7) == 0) {                 // images a multiple of 8
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, nu = gridDim.x >> 6;       // nu images per XCD
        const int hs = j / nu;
        hd = hs < 3 ? 5 + hs : (hs < 6 ? hs - 1 : hs - 6);
        bt = xcd * nu + (j - hs * nu);
    } else {
        const int blk0 = (gridDim.x & 7) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
        bt = blk0 >> 3; hd = blk0 & 7;
    }
    const int blk = bt * 8 + hd, g = bt / p.B;
    const long long row0 = (long long)bt * N;
    const T* xn = reinterpret_cast<const T*>(p.xn);
    const T* xn_img = xn + row0 * p.ldx;            // this image's first row (uniform); a token's offset below it fits 32 bits
    const T* Wg = reinterpret_cast<const T*>(p.Wqkv) + g * p.gs;
    const T* bg = reinterpret_cast<const T*>(p.bqkv) + g * p.gs;
    // window of this head (MSTr.py:785-799: heads 0-1 3x3, 2-4 5x5, 5-7 7x7), taps transposed into LDS
    const int wi = hd < 2 ? 0 : (hd < 5 ? 1 : 2), K = 3 + 2 * wi, hoff = hd - (wi == 0 ? 0 : (wi == 1 ? 2 : 5));
    // Every global load of the projection phase is issued before the first wait: the window taps (<= 8 per thread), the weight fragments and
    // biases, the xn rows of ALL of the wave's tiles.  (Taps staged by a loop of load -> LDS store, then the weights, then a tile at a time
    // were four to nine dependent memory round trips of ~3.5 k cycles each: two thirds of the launch.)
    const int KK = K * K, ntap = KK * CH;
    // (no lambdas and no "memory"-clobbering barriers in this phase: an array captured by a lambda, or live across such a barrier, is
    // kept in scratch memory, and a scratch store of a loaded value is a wait.  The xn rows are fetched by `asm volatile` loads -- which
    // the compiler can neither sink to their uses nor reorder -- followed by one explicit wait; its own waits for the compiler-visible
    // loads issued before them stay conservative because the memory counter retires in order.)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MHCA_GLOAD128(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr))
#define MHCA_KEEP128(r) asm volatile("" : "+v"(r))
    unsigned tapv[8];                     // (raw until they are parked: a conversion right behind a load is a wait)
    unsigned bcv_v;
#define MHCA_LOAD_TAPS()                                                                                                               \
    {                                                                                                                                  \
        const T* cw = reinterpret_cast<const T*>(p.cw[wi]) + g * p.gs + (long long)hoff * CH * KK;                                     \
        const T* cb = reinterpret_cast<const T*>(p.cb[wi]) + g * p.gs + hoff * CH;                                                     \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) { const int i = tid + 256 * j; tapv[j] = ld_raw16<T>(cw + (i < ntap ? i : 0)); } \
        bcv_v = ld_raw16<T>(cb + (tid < CH ? tid : 0));                                                                                \
    }
#define MHCA_STASH_TAPS()                                                                                                              \
    {                                                                                                                                  \
        const float rkk = 1.0f / (float)KK;                                                                                            \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                                                \
            const int i = tid + 256 * j;                                                                                               \
            if (i < ntap) { const int ch = (int)(((float)i + 0.5f) * rkk), tap = i - ch * KK; wcv[tap * CH + ch] = cvt_raw16<T>(tapv[j]); } \
        }                                                                                                                              \
        if (tid < CH) bcv[tid] = cvt_raw16<T>(bcv_v);                                                                                  \
    }
    // operand fragments / raw biases / destinations of column tile CT (32 of the head's 3 CH output channels): lane l31 supplies weight row
    // j = CT * 32 + l31; register group gq of a D^T tile holds 4 consecutive channels of q, k or v: which tile (0 q, 1 k, 2 v) * 65536 + channel, or -1
#define MHCA_LOAD_W(CT, AF, BIAS, DSTI)                                                                                                \
    {                                                                                                                                  \
        const int j_ = (CT) * 32 + l31, jj_ = j_ < NQ ? j_ : NQ - 1;                                                                   \
        const int wrow_ = (jj_ / CH) * C + hd * CH + (jj_ % CH);                                                                       \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) AF[ks] = *reinterpret_cast<const v8*>(Wg + (long long)wrow_ * C + ks * 16 + hh * 8); \
        _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                                                             \
            const int j0_ = (CT) * 32 + 8 * gq + 4 * hh;                                                                               \
            DSTI[gq] = j0_ < NQ ? ((j0_ / CH) << 16) + (j0_ % CH) : -1;                                                                \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                                            \
                const int ju_ = j0_ + u < NQ ? j0_ + u : NQ - 1;                                                                       \
                BIAS[4 * gq + u] = ld_raw16<T>(bg + (ju_ / CH) * C + hd * CH + (ju_ % CH));                                            \
            }                                                                                                                          \
        }                                                                                                                              \
    }
#define MHCA_PUT(ACC, DSTI, TOK)                                                                                                       \
    if ((TOK) < N) {                                                                                                                   \
        _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                                                             \
            if (DSTI[gq] >= 0) {                                                                                                       \
                const int which = DSTI[gq] >> 16, ch = DSTI[gq] & 0xffff;                                                              \
                const unsigned lo = pack2<T>(ACC[4 * gq], ACC[4 * gq + 1]), hi = pack2<T>(ACC[4 * gq + 2], ACC[4 * gq + 3]);           \
                if (which == 1) {                                                                                                      \
                    float4 f;                                                                                                          \
                    unpack2<T>(lo, f.x, f.y); unpack2<T>(hi, f.z, f.w);                                                                \
                    *reinterpret_cast<float4*>(e + (TOK) * CH + ch) = f;                                                               \
                } else {                                                                                                               \
                    *reinterpret_cast<uint2*>((which == 0 ? qs : vs) + (TOK) * CH + ch) = make_uint2(lo, hi);                          \
                }                                                                                                                      \
            }                                                                                                                          \
        }                                                                                                                              \
    }
    MHCA_STAMP(8);
    // ---- q | k | v of this head for all N tokens
    {
        const int w = tid >> 6, l = tid & 63, l31 = l & 31, hh = l >> 5;
        const int nrt = (N + 31) >> 5;
        if constexpr (STAGE_X) {
            // wide rows (C = 320): wave = column tile; the image's xn rows go through LDS once, fetched by the whole workgroup with every
            // load in flight, rows padded by 16 bytes against bank conflicts
            v8 af[KS];
            unsigned bias[16];
            int dsti[4];
            constexpr int CV = C / 8, XPT = 8;                  // 16-byte pieces per row; pieces per thread in flight at once (8 x 256 threads = 51 rows of C = 320)
            u32x4 xr[XPT];
#define MHCA_FETCH_X(I0)                                                                                                               \
    _Pragma("unroll") for (int j = 0; j < XPT; ++j) {                                                                                  \
        const int i = (I0) + tid + 256 * j, ic = i < N * CV ? i : 0, n = ic / CV, c8 = ic - n * CV;                                    \
        const T* src = xn_img + tc_mul24(n, p.ldx) + c8 * 8;                                                                            \
        MHCA_GLOAD128(xr[j], src);                                                                                                     \
    }
            MHCA_FETCH_X(0);                                    // (the asm loads first: what the compiler hoists out of the loop below -- bias conversions -- then waits behind them, not before them)
            MHCA_LOAD_TAPS();
            MHCA_LOAD_W(w % NCOL, af, bias, dsti);
            for (int i0 = 0;;) {
                asm volatile("s_waitcnt vmcnt(0)");
#pragma unroll
                for (int j = 0; j < XPT; ++j) MHCA_KEEP128(xr[j]);
                if (i0 == 0) MHCA_STASH_TAPS();
#pragma unroll
                for (int j = 0; j < XPT; ++j) {
                    const int i = i0 + tid + 256 * j, n = i / CV, c8 = i - n * CV;
                    if (i < N * CV) *reinterpret_cast<u32x4*>(xs + n * XLD + c8 * 8) = xr[j];
                }
                i0 += 256 * XPT;
                if (i0 >= N * CV) break;
                MHCA_FETCH_X(i0);
            }
            MHCA_STAMP(9);
            __syncthreads();
            for (int rt = w / NCOL; rt < nrt; rt += RSTEP) {
                const int tok = rt * 32 + l31, tc = tok < N ? tok : N - 1;
                const T* xr = xs + tc * XLD + hh * 8;
                tc_f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = cvt_raw16<T>(bias[r]);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc = TcHalf<T>::mfma(af[ks], *reinterpret_cast<const v8*>(xr + ks * 16), acc);
                MHCA_PUT(acc, dsti, tok);
            }
        } else {
            // wave = row tiles w, w + 4, ... against ALL column tiles of the head.  A tile's 32 xn rows are fetched as whole rows (16 bytes
            // per lane, consecutive lanes consecutive bytes -- per-lane row-strided fragment loads touch 32 cache lines per instruction),
            // parked in a wave-private LDS tile (no barrier: a wave reads what it wrote) and read back as operand fragments.  MAXT tiles
            // of a wave are in flight at once: all of them at the 224^2 shapes (25 / 7 row tiles over 4 waves).
            constexpr int LPT = 32 * C / (64 * 8);             // 16-byte loads per lane and tile
            constexpr int MAXT = C <= 64 ? 7 : 2;
            v8 af[NCOL][KS];
            unsigned bias[NCOL][16];
            int dsti[NCOL][4];
            T* xw = xs + w * 32 * XLD;
            const int mine = nrt > w ? (nrt - w + 3) >> 2 : 0;
            u32x4 xr[MAXT][LPT];
#define MHCA_FETCH_T(T0)                                                                                                               \
    _Pragma("unroll") for (int u = 0; u < MAXT; ++u)                                                                                   \
        _Pragma("unroll") for (int i = 0; i < LPT; ++i) {            /* (rows past the image: a clamped row, dropped in MHCA_PUT) */   \
            const int flat = (i * 64 + l) * 8, row = flat / C, col = flat - row * C;                                                   \
            const int tok = (w + 4 * ((T0) + u)) * 32 + row, tc = tok < N ? tok : N - 1;                                               \
            const T* src = xn_img + tc_mul24(tc, p.ldx) + col;                                                                          \
            MHCA_GLOAD128(xr[u][i], src);                                                                                              \
        }
            MHCA_FETCH_T(0);                                    // (the asm loads first: what the compiler hoists out of the loop below -- bias conversions -- then waits behind them, not before them)
            MHCA_LOAD_TAPS();
#pragma unroll
            for (int ct = 0; ct < NCOL; ++ct) MHCA_LOAD_W(ct, af[ct], bias[ct], dsti[ct]);
            for (int t0 = 0;;) {
                asm volatile("s_waitcnt vmcnt(0)");
#pragma unroll
                for (int u = 0; u < MAXT; ++u)
#pragma unroll
                    for (int i = 0; i < LPT; ++i) MHCA_KEEP128(xr[u][i]);
                if (t0 == 0) { MHCA_STASH_TAPS(); MHCA_STAMP(9); }
#pragma unroll
                for (int u = 0; u < MAXT; ++u) {
#pragma unroll
                    for (int i = 0; i < LPT; ++i) {
                        const int flat = (i * 64 + l) * 8, row = flat / C, col = flat - row * C;
                        *reinterpret_cast<u32x4*>(xw + row * XLD + col) = xr[u][i];
                    }
                    const int tok = t0 + u < mine ? (w + 4 * (t0 + u)) * 32 + l31 : N;
                    tc_f32x16 acc[NCOL];
#pragma unroll
                    for (int ct = 0; ct < NCOL; ++ct)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[ct][r] = cvt_raw16<T>(bias[ct][r]);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const v8 bfr = *reinterpret_cast<const v8*>(xw + l31 * XLD + ks * 16 + hh * 8);
#pragma unroll
                        for (int ct = 0; ct < NCOL; ++ct) acc[ct] = TcHalf<T>::mfma(af[ct][ks], bfr, acc[ct]);
                    }
#pragma unroll
                    for (int ct = 0; ct < NCOL; ++ct) MHCA_PUT(acc[ct], dsti[ct], tok);
                }
                t0 += MAXT;
                if (t0 >= mine) break;
                MHCA_FETCH_T(t0);
            }
        }
    }
    __syncthreads();
    MHCA_STAMP(1);
    // ---- q | k | v out for the backward (16-byte rows pieces), the window over v
    {
        T* qo = reinterpret_cast<T*>(p.qkv) + row0 * p.ldq + hd * CH;
        for (int i = tid; i < N * NV; i += 256) {
            const int n = i / NV, c0 = (i - n * NV) * 8;
            T* dst = qo + tc_mul24(n, p.ldq) + c0;
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(qs + n * CH + c0);
            *reinterpret_cast<uint4*>(dst + C) = pack16<T>(e + n * CH + c0);
            *reinterpret_cast<uint4*>(dst + 2 * C) = *reinterpret_cast<const uint4*>(vs + n * CH + c0);
        }
    }
    MHCA_STAMP(2);
    if (K == 3) mhca_conv_pick<3, CH, T>(cvs, vs, wcv, bcv, p.H, p.W);
    else if (K == 5) mhca_conv_pick<5, CH, T>(cvs, vs, wcv, bcv, p.H, p.W);
    else mhca_conv_pick<7, CH, T>(cvs, vs, wcv, bcv, p.H, p.W);
    __syncthreads();
    MHCA_STAMP(3);
    fa_softmax_cols(e, cmax, cinv, red, N, CH);
    MHCA_STAMP(4);
    fa_gram(ctx, e, vs, N, CH);
    MHCA_STAMP(5);
    for (int i = tid; i < CH * CH; i += 256) ctx[i] *= cinv[i / CH];
    if (tid < CH) { p.stats[((long long)blk * 2) * CH + tid] = cmax[tid]; p.stats[((long long)blk * 2 + 1) * CH + tid] = cinv[tid]; }
    __syncthreads();
    {
        T* oo = reinterpret_cast<T*>(p.o) + row0 * p.ldo + hd * CH;
        T* co = reinterpret_cast<T*>(p.convv) + row0 * p.ldc + hd * CH;
        const float scale = p.scale;
#pragma unroll 2
        for (int i = tid; i < N * NV; i += 256) {
            const int n = i / NV, j0 = (i - n * NV) * 8;
            float cv[8], acc[8];
            const uint4 craw = *reinterpret_cast<const uint4*>(cvs + n * CH + j0);
            *reinterpret_cast<uint4*>(co + tc_mul24(n, p.ldc) + j0) = craw;
            unpack16<T>(craw, cv);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = 0.f;
            const T* qr = qs + n * CH;
#pragma unroll
            for (int c8 = 0; c8 < CH; c8 += 8) {
                float q8[8];
                fa_get8(qr + c8, q8);
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) {
                    const float* cr = ctx + (c8 + cc) * CH + j0;
                    const float4 c0 = *reinterpret_cast<const float4*>(cr), c1 = *reinterpret_cast<const float4*>(cr + 4);
                    acc[0] += q8[cc] * c0.x; acc[1] += q8[cc] * c0.y; acc[2] += q8[cc] * c0.z; acc[3] += q8[cc] * c0.w;
                    acc[4] += q8[cc] * c1.x; acc[5] += q8[cc] * c1.y; acc[6] += q8[cc] * c1.z; acc[7] += q8[cc] * c1.w;
                }
            }
            float qj[8];
            fa_get8(qr + j0, qj);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = scale * acc[u] + qj[u] * cv[u];
            *reinterpret_cast<uint4*>(oo + tc_mul24(n, p.ldo) + j0) = pack16<T>(acc);
        }
    }
#ifdef TC_MHCA_TIMING
    MHCA_STAMP(6);
    if (tid == 0) {                       // (experiment builds only: the caller's stats buffer has room for 8 stamps per workgroup behind the statistics)
        long long* dbg = reinterpret_cast<long long*>(p.stats + (long long)gridDim.x * 2 * CH) + (long long)blk * 16;
        for (int i = 0; i < 7; ++i) dbg[i] = tstamp[i];
        dbg[7] = hd;
        for (int i = 8; i < 10; ++i) dbg[i] = tstamp[i];
        dbg[10] = tstamp[9];
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Backward of the same half, one launch per (image, head): factor_att_bwd_kernel's arithmetic + the window's backward, which used to be
// a launch of its own (tc_dwconv_multi mode 3) reading dconvv back from memory:
//   dconvv = do (.) q  stays in LDS;  dv += crpe^T(dconvv) (the window flipped);  dw[c][tap] += sum_n dconvv[n, c] v[n + tap, c],
//   db[c] += sum_n dconvv[n, c]  -- per-head sums over the image, added to the fp32 gradients with one atomic per word and workgroup.
struct MhcaAttBwdDev {
    const void *qkv, *convv, *go, *cw[3];
    const float* stats;
    void* dqkv;
    float *dcw[3], *dcb[3];
    long long gs;
    int ldq, ldc, ldgo, ldd, acc_q, acc_k, acc_v, B, N, H, W;
    float scale;
};

// dw / db of one head's K x K window.  Work item = (filter row ky, 8-channel group, image row y): it walks the W pixels of its row with
// K x 8 accumulators (one filter row), G = 8 / 16 / 32 / 64 lanes (the image rows, padded to a power of two) then fold by DPP / shuffles
// and lane 0 of a group adds the K x 8 (+ 8 for the bias, ky == 0) sums to the gradient arrays.
#ifdef TC_MHCA_TIMING
__device__ long long g_wg_stamps[4096 * 4];
#define WG_STAMP(i) if (threadIdx.x == 0 && it0 == 0) g_wg_stamps[blockIdx.x * 4 + (i)] = clock64()
#else
#define WG_STAMP(i)
#endif
// dw / db of one head's K x K window.  Work item = (filter row ky, 8-channel group, image row y): it walks the W pixels of its row with
// K x 8 accumulators (one filter row) and a K-wide window of v in registers -- the walk is unrolled K pixels at a time so that the
// window's slots are compile-time names: one new v vector and one dconvv vector are read per pixel (reading the K window vectors per
// pixel cost ~230 instructions per pixel for a lone wave per SIMD: the phase was half of the launch).  G = 8 / 16 / 32 / 64 lanes
// (the image rows, padded to a power of two) then fold by DPP / shuffles; lane 0 of a group leaves the K x 8 (+ 8 for the bias,
// ky == 0) sums in the LDS array `dwl` ([CH][K*K] then [CH]), which the caller adds to the gradient arrays with full-width atomics.
template <int K, int CH, int G, typename T>
__device__ __forceinline__ void mhca_conv_wgrad(const T* dc, const T* vs, float* dwl, int H, int W) {
    constexpr int P = K / 2, NV = CH / 8, KK = K * K;
    const int items = K * NV * G;
    for (int it0 = 0; it0 < items; it0 += 256) {
        const int it = it0 + threadIdx.x;
        const int y = it & (G - 1), r = it / G, cvi = r % NV, ky = r / NV, c0 = cvi * 8;
        const bool live = it < items && y < H;
        const int yy = y + ky - P;
        const bool rv = live && (unsigned)yy < (unsigned)H;
        float acc[K][8], accb[8], win[K][8];
#pragma unroll
        for (int kx = 0; kx < K; ++kx)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[kx][u] = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) accb[u] = 0.f;
        WG_STAMP(0);
        const T* dr = dc + ((live ? y : 0) * W) * CH + c0;
        const T* vr = vs + ((rv ? yy : 0) * W) * CH + c0;
        // window slots: at pixel x = xb + j slot (j + i) % K holds v[x - P + i]; before the walk: v[-P .. P-1] (zeros left of the row)
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int xx = i - P;
            uint4 raw = *reinterpret_cast<const uint4*>(vr + (xx > 0 && xx < W ? xx : 0) * CH);
            if (!(rv && xx >= 0 && xx < W)) raw = make_uint4(0u, 0u, 0u, 0u);
            unpack16<T>(raw, win[i]);
        }
        for (int xb = 0; xb < W; xb += K) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int x = xb + j;
                // the newest element v[x + P] replaces v[x - P - 1] in slot (j + K - 1) % K ... which at step j is the slot of window index K - 1
                {
                    const int xx = x + P;
                    uint4 raw = *reinterpret_cast<const uint4*>(vr + (xx < W ? xx : 0) * CH);
                    if (!(rv && xx < W)) raw = make_uint4(0u, 0u, 0u, 0u);
                    unpack16<T>(raw, win[(j + K - 1) % K]);
                }
                uint4 draw = *reinterpret_cast<const uint4*>(dr + (x < W ? x : 0) * CH);
                if (!(live && x < W)) draw = make_uint4(0u, 0u, 0u, 0u);
                float d8[8];
                unpack16<T>(draw, d8);
#pragma unroll
                for (int u = 0; u < 8; ++u) accb[u] += d8[u];
#pragma unroll
                for (int kx = 0; kx < K; ++kx)
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc[kx][u] = fmaf(d8[u], win[(j + kx) % K][u], acc[kx][u]);
            }
        }
        WG_STAMP(1);
#pragma unroll
        for (int kx = 0; kx < K; ++kx)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[kx][u] = tc_group_sum<G>(acc[kx][u]);
        WG_STAMP(2);
        if (it < items && y == 0) {
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
#pragma unroll
                for (int u = 0; u < 8; ++u) dwl[(c0 + u) * KK + ky * K + kx] = acc[kx][u];
        }
        if (ky == 0) {                                           // (uniform per G-lane group)
#pragma unroll
            for (int u = 0; u < 8; ++u) accb[u] = tc_group_sum<G>(accb[u]);
            if (it < items && y == 0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) dwl[CH * KK + c0 + u] = accb[u];
            }
        }
        WG_STAMP(3);
    }
}
#ifdef TC_MHCA_TIMING
extern "C" int tc_dbg_wg_stamps(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wg_stamps), sizeof(long long) * n); }
#endif
template <int K, int CH, typename T>
__device__ __forceinline__ void mhca_conv_wgrad_pick(const T* dc, const T* vs, float* dwl, int H, int W) {
    if (H <= 8) mhca_conv_wgrad<K, CH, 8, T>(dc, vs, dwl, H, W);
    else if (H <= 16) mhca_conv_wgrad<K, CH, 16, T>(dc, vs, dwl, H, W);
    else if (H <= 32) mhca_conv_wgrad<K, CH, 32, T>(dc, vs, dwl, H, W);
    else mhca_conv_wgrad<K, CH, 64, T>(dc, vs, dwl, H, W);
}

template <typename T, int C>
__global__ __launch_bounds__(256, 2) void mhca_att_bwd_kernel(MhcaAttBwdDev p) {
    constexpr int CH = C / 8, NV = CH / 8, VEC = 8;
    extern __shared__ float sm[];
    const int N = p.N, tid = threadIdx.x;
#ifdef TC_MHCA_TIMING
    long long tstamp[8];
    MHCA_STAMP(0);
#endif
    float* e = sm;                        // softmax(k) (normalised), fp32
    float* ctx = e + N * CH;
    float* dctx = ctx + CH * CH;
    float* tcol = dctx + CH * CH;         // [CH] (+ padding to 16 bytes)
    float* ctxT = tcol + ((CH + 3) & ~3);
    float* dctxT = ctxT + CH * CH;
    float* wcv = dctxT + CH * CH;         // [49][CH] the window's taps FLIPPED (the input gradient of a correlation is the correlation with the flipped window)
    float* bz = wcv + 49 * CH;            // [CH] zeros (the window routine adds a bias)
    T* vs = reinterpret_cast<T*>(bz + CH);
    T* qs = vs + N * CH;                  // q, then dconvv = do (.) q in place
    T* gs_ = qs + N * CH;                 // do
    T* dvc = gs_ + N * CH;                // crpe^T(dconvv)
    int bt, hd;
    if (((gridDim.x >> 3) & 7) == 0) {                 // (as in the forward kernel: heads of an image on one XCD, long windows first)
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, nu = gridDim.x >> 6;
        const int hs = j / nu;
        hd = hs < 3 ? 5 + hs : (hs < 6 ? hs - 1 : hs - 6);
        bt = xcd * nu + (j - hs * nu);
    } else {
        const int blk0 = (gridDim.x & 7) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
        bt = blk0 >> 3; hd = blk0 & 7;
    }
    const int blk = bt * 8 + hd, g = bt / p.B;
    const long long row0 = (long long)bt * N;
    const int col0 = hd * CH;
    const int wi = hd < 2 ? 0 : (hd < 5 ? 1 : 2), K = 3 + 2 * wi, hoff = hd - (wi == 0 ? 0 : (wi == 1 ? 2 : 5)), KK = K * K, ntap = KK * CH;
    const T* qg = reinterpret_cast<const T*>(p.qkv) + row0 * p.ldq + col0;
    const T* gg = reinterpret_cast<const T*>(p.go) + row0 * p.ldgo + col0;
    const T* cg = reinterpret_cast<const T*>(p.convv) + row0 * p.ldc + col0;
    // ---- every tile of the head in flight at once: q | k | v | do pieces of up to 4 x 256 (token, 8-channel) items, this thread's convv pieces, the taps
    const int nitem = N * NV;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 cvr[4];
    {
        // (asm volatile loads, one explicit wait: see the forward kernel -- the compiler otherwise sinks every piece's loads to its LDS store,
        // one memory round trip per piece)
        u32x4 rq[4], rk[4], rv[4], rg[4];
#define MHCA_FETCH_B(I0)                                                                                                               \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                                    \
        const int i = (I0) + tid + 256 * j, ic = i < nitem ? i : 0, n = ic / NV, c8 = (ic - n * NV) * 8;                               \
        const T* src = qg + tc_mul24(n, p.ldq) + c8;                                                                                 \
        const T* srk = src + C;                                                                                                        \
        const T* srv = src + 2 * C;                                                                                                    \
        const T* srg = gg + tc_mul24(n, p.ldgo) + c8;                                                                                \
        MHCA_GLOAD128(rq[j], src); MHCA_GLOAD128(rk[j], srk); MHCA_GLOAD128(rv[j], srv); MHCA_GLOAD128(rg[j], srg);                    \
    }
        MHCA_FETCH_B(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + 256 * j, ic = i < nitem ? i : 0, n = ic / NV, c8 = (ic - n * NV) * 8;
            const T* src = cg + tc_mul24(n, p.ldc) + c8;
            MHCA_GLOAD128(cvr[j], src);
        }
        unsigned tapv[8];
        const T* cw = reinterpret_cast<const T*>(p.cw[wi]) + g * p.gs + (long long)hoff * CH * KK;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int i = tid + 256 * j; tapv[j] = ld_raw16<T>(cw + (i < ntap ? i : 0)); }
        const float stv = p.stats[((long long)blk * 2) * CH + (tid < 2 * CH ? tid : 0)];      // cmax[CH] | cinv[CH] of this head, through LDS (tcol / ctxT are free until the grams are done)
        float* cmaxp = ctxT;
        for (int i0 = 0;;) {
            asm volatile("s_waitcnt vmcnt(0)");
#pragma unroll
            for (int j = 0; j < 4; ++j) { MHCA_KEEP128(rq[j]); MHCA_KEEP128(rk[j]); MHCA_KEEP128(rv[j]); MHCA_KEEP128(rg[j]); }
            if (i0 == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) MHCA_KEEP128(cvr[j]);
                const float rkk = 1.0f / (float)KK;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = tid + 256 * j;
                    if (i < ntap) { const int ch = (int)(((float)i + 0.5f) * rkk), tap = i - ch * KK; wcv[(KK - 1 - tap) * CH + ch] = cvt_raw16<T>(tapv[j]); }
                }
                if (tid < CH) bz[tid] = 0.f;
                if (tid < 2 * CH) cmaxp[tid] = stv;
                __syncthreads();
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + tid + 256 * j;
                if (i < nitem) {
                    const int n = i / NV, c8 = (i - n * NV) * 8;
                    *reinterpret_cast<u32x4*>(qs + n * CH + c8) = rq[j];
                    *reinterpret_cast<u32x4*>(vs + n * CH + c8) = rv[j];
                    *reinterpret_cast<u32x4*>(gs_ + n * CH + c8) = rg[j];
                    float k8[8];
                    unpack16<T>(make_uint4(rk[j].x, rk[j].y, rk[j].z, rk[j].w), k8);
#pragma unroll
                    for (int u = 0; u < 8; ++u) k8[u] = __expf(k8[u] - cmaxp[c8 + u]) * cmaxp[CH + c8 + u];
                    *reinterpret_cast<float4*>(e + n * CH + c8) = make_float4(k8[0], k8[1], k8[2], k8[3]);
                    *reinterpret_cast<float4*>(e + n * CH + c8 + 4) = make_float4(k8[4], k8[5], k8[6], k8[7]);
                }
            }
            i0 += 1024;
            if (i0 >= nitem) break;
            MHCA_FETCH_B(i0);
        }
    }
    __syncthreads();
    MHCA_STAMP(1);
    fa_gram(ctx, e, vs, N, CH);
    fa_gram(dctx, qs, gs_, N, CH);
    for (int i = tid; i < CH * CH; i += 256) dctx[i] *= p.scale;
    // dconvv = do (.) q, in place of q (q is not read again: dq's q-term is do (.) convv)
    for (int i = tid; i < nitem; i += 256) {
        float a[8], b[8];
        fa_get8(qs + i * 8, a); fa_get8(gs_ + i * 8, b);
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] *= b[u];
        *reinterpret_cast<uint4*>(qs + i * 8) = pack16<T>(a);
    }
    __syncthreads();
    if (tid < CH) { float t = 0.f; for (int j = 0; j < CH; ++j) t += ctx[tid * CH + j] * dctx[tid * CH + j]; tcol[tid] = t; }
    for (int i = tid; i < CH * CH; i += 256) { const int c = i / CH, j = i - c * CH; ctxT[j * CH + c] = ctx[i]; dctxT[j * CH + c] = dctx[i]; }
    MHCA_STAMP(2);
    // the window's backward on the dconvv tile: input gradient (-> dvc), then the head's weight / bias gradient sums -- parked in the taps' LDS
    // array (the taps are not read again) and added to the fp32 gradients by the whole workgroup: one atomic per word, consecutive lanes
    // consecutive words (lane 0 of every row group issuing its 64 atomics one lane at a time took ~180 cycles per instruction)
    if (K == 3) mhca_conv_pick<3, CH, T>(dvc, qs, wcv, bz, p.H, p.W);
    else if (K == 5) mhca_conv_pick<5, CH, T>(dvc, qs, wcv, bz, p.H, p.W);
    else mhca_conv_pick<7, CH, T>(dvc, qs, wcv, bz, p.H, p.W);
    MHCA_STAMP(3);
    __syncthreads();
    if (K == 3) mhca_conv_wgrad_pick<3, CH, T>(qs, vs, wcv, p.H, p.W);
    else if (K == 5) mhca_conv_wgrad_pick<5, CH, T>(qs, vs, wcv, p.H, p.W);
    else mhca_conv_wgrad_pick<7, CH, T>(qs, vs, wcv, p.H, p.W);
    __syncthreads();
    {
        float* dwg = p.dcw[wi] + g * p.gs + (long long)hoff * CH * KK;
        float* dbg = p.dcb[wi] + g * p.gs + hoff * CH;
        for (int i = tid; i < ntap + CH; i += 256) atomicAdd(i < ntap ? dwg + i : dbg + (i - ntap), wcv[i]);
    }
    __syncthreads();
    MHCA_STAMP(4);
    T* dq0 = reinterpret_cast<T*>(p.dqkv) + row0 * p.ldd + col0;
    auto token = [&](int i, const uint4 cvraw) __attribute__((always_inline)) {
        const int n = i / NV, c0 = (i - n * NV) * VEC;
        const T* gr = gs_ + n * CH;
        const T* vr = vs + n * CH;
        const float* er = e + n * CH;
        float a_q[VEC], a_ks[VEC], a_v[VEC], cv[VEC];
#pragma unroll
        for (int u = 0; u < VEC; ++u) a_q[u] = a_ks[u] = a_v[u] = 0.f;
#pragma unroll 1
        for (int j8 = 0; j8 < CH; j8 += 8) {
            float g8[8], v8[8], e8[8];
            fa_get8(gr + j8, g8); fa_get8(vr + j8, v8); fa_get8(er + j8, e8);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j = j8 + jj;
                float cq[VEC], ck[VEC], cw8[VEC];
#pragma unroll
                for (int u = 0; u < VEC; u += 4) {
                    *reinterpret_cast<float4*>(cq + u) = *reinterpret_cast<const float4*>(ctxT + j * CH + c0 + u);
                    *reinterpret_cast<float4*>(ck + u) = *reinterpret_cast<const float4*>(dctxT + j * CH + c0 + u);
                    *reinterpret_cast<float4*>(cw8 + u) = *reinterpret_cast<const float4*>(dctx + j * CH + c0 + u);
                }
#pragma unroll
                for (int u = 0; u < VEC; ++u) {
                    a_q[u] += g8[jj] * cq[u];                 // dq[n,c]  += scale * sum_j do[n,j] ctx[c,j]   (dctx carries the scale; ctx does not:)
                    a_ks[u] += v8[jj] * ck[u];                // dksm[n,c] = sum_j v[n,j] dctx[c,j]
                    a_v[u] += e8[jj] * cw8[u];                // dv[n,c]   = sum_i ksm[n,i] dctx[i,c]
                }
            }
        }
        unpack16<T>(cvraw, cv);
        float dvc8[8], g0[8];
        fa_get8(dvc + n * CH + c0, dvc8);
        fa_get8(gr + c0, g0);
        T* pq = dq0 + tc_mul24(n, p.ldd) + c0;
        float old[VEC];
#pragma unroll
        for (int u = 0; u < VEC; ++u) {
            a_q[u] = p.scale * a_q[u] + g0[u] * cv[u];
            a_ks[u] = er[c0 + u] * (a_ks[u] - tcol[c0 + u]);
            a_v[u] += dvc8[u];
        }
        if (p.acc_q) { unpack16<T>(*reinterpret_cast<const uint4*>(pq), old);
#pragma unroll
            for (int u = 0; u < VEC; ++u) a_q[u] += old[u]; }
        if (p.acc_k) { unpack16<T>(*reinterpret_cast<const uint4*>(pq + C), old);
#pragma unroll
            for (int u = 0; u < VEC; ++u) a_ks[u] += old[u]; }
        if (p.acc_v) { unpack16<T>(*reinterpret_cast<const uint4*>(pq + 2 * C), old);
#pragma unroll
            for (int u = 0; u < VEC; ++u) a_v[u] += old[u]; }
        *reinterpret_cast<uint4*>(pq) = pack16<T>(a_q);
        *reinterpret_cast<uint4*>(pq + C) = pack16<T>(a_ks);
        *reinterpret_cast<uint4*>(pq + 2 * C) = pack16<T>(a_v);
    };
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
        const int i = tid + 256 * jt;
        const uint4 c = make_uint4(cvr[jt].x, cvr[jt].y, cvr[jt].z, cvr[jt].w);
        if (i < nitem) token(i, c);
    }
    for (int i = tid + 1024; i < nitem; i += 256) {
        const int n = i / NV, c0 = (i - n * NV) * VEC;
        token(i, *reinterpret_cast<const uint4*>(cg + tc_mul24(n, p.ldc) + c0));
    }
#ifdef TC_MHCA_TIMING
    MHCA_STAMP(5);
    if (tid == 0) {                       // (experiment builds only: the caller's dqkv buffer is followed by room for 8 stamps per workgroup -- passed through p.stats' tail)
        long long* dbg = reinterpret_cast<long long*>(const_cast<float*>(p.stats) + (long long)gridDim.x * 2 * CH) + (long long)blk * 8;
        for (int i = 0; i < 6; ++i) dbg[i] = tstamp[i];
        dbg[7] = hd;
    }
#endif
}

template <int C> __host__ size_t mhca_att_bwd_smem(int N) {
    constexpr int CH = C / 8;
    return sizeof(float) * ((size_t)N * CH + (size_t)4 * CH * CH + ((CH + 3) & ~3) + 49 * CH + CH) + 4 * 2 * (size_t)N * CH;
}
template <typename T, int C> int mhca_att_bwd_launch(const MhcaAttBwdDev& p, int Bt, hipStream_t s) {
    const size_t smem = mhca_att_bwd_smem<C>(p.N);
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)mhca_att_bwd_kernel<T, C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((mhca_att_bwd_kernel<T, C>), dim3(Bt * 8), dim3(256), smem, s, p);
    return tc_launch_status();
}

template <int C> __host__ size_t mhca_att_smem(int N) {
    constexpr int CH = C / 8;
    const size_t cvs = 2 * (size_t)N * CH, stage = 2 * (size_t)4 * 32 * (C + 8);           // convv tile / the four xn staging tiles share their bytes
    return sizeof(float) * ((size_t)N * CH + (size_t)CH * CH + 3 * CH + 256 + 49 * CH) + 2 * 2 * (size_t)N * CH +
           (C >= 320 ? cvs + 2 * (size_t)N * (C + 8) : (cvs > stage ? cvs : stage));
}

template <typename T, int C> int mhca_att_launch(const MhcaAttDev& p, int Bt, hipStream_t s) {
    const size_t smem = mhca_att_smem<C>(p.N);
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)mhca_att_fwd_kernel<T, C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((mhca_att_fwd_kernel<T, C>), dim3(Bt * 8), dim3(256), smem, s, p);
    return tc_launch_status();
}

}  // namespace

extern "C" long long tc_factor_att_stats_floats(int Bt, int heads, int Ch) { return (long long)Bt * heads * 2 * Ch; }

extern "C" int tc_factor_att_fwd(const void* q, const void* k, const void* v, int ld, const void* convv, int ldc, void* o, int ldo,
                                 float* stats, int Bt, int N, int heads, int Ch, float scale, int dtype, void* stream) {
    if (!q || !k || !v || !convv || !o || !stats || Bt <= 0 || N <= 0 || heads <= 0 || Ch <= 0 || Ch > FA_MAXCH) return TC_ERR_ARG;
    const int vec = dtype == TC_F32 ? 4 : 8;
    if (Ch % 8 || ld % vec || ldc % vec || ldo % vec || (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)convv | (uintptr_t)o) & 15)) return TC_ERR_ARG;
    const size_t esz = dtype == TC_F32 ? 4 : 2;
    const size_t smem = sizeof(float) * ((size_t)N * Ch + (size_t)Ch * Ch + 2 * Ch + 256) + 2 * esz * N * Ch;
    if (smem > 150 * 1024) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, {
        if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)factor_att_fwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((factor_att_fwd_kernel<T>), dim3(Bt * heads), dim3(256), smem, (hipStream_t)stream, (const T*)q, (const T*)k,
                           (const T*)v, ld, (const T*)convv, ldc, (T*)o, ldo, stats, N, Ch, heads, scale);
    });
    return tc_launch_status();
}

extern "C" int tc_factor_att_bwd(const void* q, const void* k, const void* v, int ld, const void* convv, int ldc, const void* go, int ldgo,
                                 const float* stats, void* dq, void* dk, void* dv, int ldd, int acc_q, int acc_k, int acc_v, void* dconvv,
                                 int lddc, int Bt, int N, int heads, int Ch, float scale, int dtype, void* stream) {
    if (!q || !k || !v || !convv || !go || !stats || !dq || !dk || !dv || !dconvv || Bt <= 0 || N <= 0 || heads <= 0 || Ch <= 0 ||
        Ch > FA_MAXCH)
        return TC_ERR_ARG;
    const int vec = dtype == TC_F32 ? 4 : 8;
    if (Ch % 8 || ld % vec || ldc % vec || ldgo % vec || ldd % vec || lddc % vec ||
        (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)convv | (uintptr_t)go | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv | (uintptr_t)dconvv) & 15))
        return TC_ERR_ARG;
    const size_t esz = dtype == TC_F32 ? 4 : 2;
    const size_t smem = sizeof(float) * ((size_t)N * Ch + (size_t)4 * Ch * Ch + ((Ch + 3) & ~3)) + 3 * esz * N * Ch;
    if (smem > 150 * 1024) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, {
        if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)factor_att_bwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((factor_att_bwd_kernel<T>), dim3(Bt * heads), dim3(256), smem, (hipStream_t)stream, (const T*)q, (const T*)k,
                           (const T*)v, ld, (const T*)convv, ldc, (const T*)go, ldgo, stats, (T*)dq, (T*)dk, (T*)dv, ldd, acc_q, acc_k,
                           acc_v, (T*)dconvv, lddc, N, Ch, heads, scale);
    });
    return tc_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Head of an MHCABlock (MSTr.py:935-940): t1 = t + cpe(t) (ConvPosEnc: depthwise 3x3 + bias, :744-752), xn = LayerNorm(t1) (norm1, eps 1e-6)
// in ONE launch (16-bit storage): the depthwise launch wrote t1, the LayerNorm launch read it back -- a 5 us launch and a memory round
// trip per block for a map that is read once.  Eight lanes own a token (C / 64 vectors of 8 channels each); the 9 taps are read straight
// from global memory (the neighbouring tokens of a workgroup's 32 overlap in the L1), the taps' weights sit transposed in LDS; both
// results, the mean and the reciprocal standard deviation leave exactly as tc_dwconv_fwd + tc_layernorm_fwd leave them (their backward
// entries apply unchanged): t1 is rounded to the storage type before the statistics are taken.
struct DwLnDev {
    const void *x, *w, *b, *gamma, *beta;
    void *t1, *xn;
    float *mean, *rstd;
    long long wstride, gstride;
    int ldx, ldt, ldn, B, H, W, rows_per_group;
    float eps;
};
template <typename T, int C>
__global__ __launch_bounds__(256) void dw_ln_fwd_kernel(DwLnDev p) {
    constexpr int NVL = C / 64;                                  // 8-channel vectors per lane
    __shared__ float wt[9 * C], bs[C], gm[C], bt[C];
    const int tid = threadIdx.x, sub = tid & 7;
    const long long row = (long long)blockIdx.x * 32 + (tid >> 3);
    const long long total = (long long)gridDim.y * 0 + (long long)p.rows_per_group * gridDim.y;
    const int g = blockIdx.y;
    {
        const T* w = reinterpret_cast<const T*>(p.w) + g * p.wstride;
        const T* b = reinterpret_cast<const T*>(p.b) + g * p.wstride;
        const T* ga = reinterpret_cast<const T*>(p.gamma) + g * p.gstride;
        const T* be = reinterpret_cast<const T*>(p.beta) + g * p.gstride;
        for (int i = tid; i < 9 * C; i += 256) { const int ch = i / 9, tap = i - ch * 9; wt[tap * C + ch] = ldf<T>(w + i); }
        for (int i = tid; i < C; i += 256) { bs[i] = ldf<T>(b + i); gm[i] = ldf<T>(ga + i); bt[i] = ldf<T>(be + i); }
    }
    (void)total;
    const bool live = row < p.rows_per_group;
    const long long r = live ? row : 0;
    const int hw = p.H * p.W, img = (int)(r / hw), pix = (int)(r - (long long)img * hw), y = pix / p.W, x = pix - y * p.W;
    const long long grow = (long long)g * p.rows_per_group;
    const T* xb = reinterpret_cast<const T*>(p.x) + (grow + (long long)img * hw) * p.ldx;
    uint4 raw[9][NVL];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int yy = y + ky - 1, xx = x + kx - 1;
            const bool ok = (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
            const T* src = xb + (long long)((ok ? yy : y) * p.W + (ok ? xx : x)) * p.ldx + sub * 8;
#pragma unroll
            for (int v = 0; v < NVL; ++v) {
                raw[ky * 3 + kx][v] = *reinterpret_cast<const uint4*>(src + v * 64);
                if (!ok) raw[ky * 3 + kx][v] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    __syncthreads();
    float t1v[NVL][8];
    float s1 = 0.f;
#pragma unroll
    for (int v = 0; v < NVL; ++v) {
        const int c0 = v * 64 + sub * 8;
        float acc[8], ctr[8];
        unpack16<T>(raw[4][v], ctr);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = bs[c0 + u];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            float xv[8], w8[8];
            unpack16<T>(raw[tap][v], xv);
            fa_get8(wt + tap * C + c0, w8);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = fmaf(xv[u], w8[u], acc[u]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] += ctr[u];
        const uint4 pk = pack16<T>(acc);
        if (live) *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.t1) + (grow + r) * p.ldt + c0) = pk;
        unpack16<T>(pk, t1v[v]);                                 // (the statistics see what the next reader of t1 will see)
#pragma unroll
        for (int u = 0; u < 8; ++u) s1 += t1v[v][u];
    }
    const float mean = tc_group_sum<8>(s1) * (1.0f / C);
    float s2 = 0.f;
#pragma unroll
    for (int v = 0; v < NVL; ++v)
#pragma unroll
        for (int u = 0; u < 8; ++u) { const float d = t1v[v][u] - mean; s2 += d * d; }
    const float rstd = rsqrtf(tc_group_sum<8>(s2) * (1.0f / C) + p.eps);
    if (live) {
#pragma unroll
        for (int v = 0; v < NVL; ++v) {
            const int c0 = v * 64 + sub * 8;
            float o[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) o[u] = (t1v[v][u] - mean) * rstd * gm[c0 + u] + bt[c0 + u];
            *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.xn) + (grow + r) * p.ldn + c0) = pack16<T>(o);
        }
        if (sub == 0) { p.mean[grow + r] = mean; p.rstd[grow + r] = rstd; }
    }
}

// LDS bytes of the fused attention half for width C and N tokens per image (0: unsupported width)
static size_t mhca_att_smem_of(int C, int N) {
    return C == 64 ? mhca_att_smem<64>(N) : C == 128 ? mhca_att_smem<128>(N) : C == 320 ? mhca_att_smem<320>(N) : 0;
}
extern "C" int tc_mhca_att_supported(int C, int N, int dtype) {
    if (dtype != TC_BF16 && dtype != TC_F16) return 0;
    const size_t sm = mhca_att_smem_of(C, N);
    return sm > 0 && sm <= 150 * 1024 && N > 0;
}
extern "C" int tc_mhca_att_fwd(const void* xn, int ldx, const void* Wqkv, const void* bqkv, const void* w3, const void* b3, const void* w5,
                               const void* b5, const void* w7, const void* b7, long long gs, void* qkv, int ldq, void* convv, int ldc,
                               void* o, int ldo, float* stats, int groups, int B, int H, int W, int C, float scale, int dtype, void* stream) {
    if (!xn || !Wqkv || !bqkv || !w3 || !b3 || !w5 || !b5 || !w7 || !b7 || !qkv || !convv || !o || !stats || groups <= 0 || B <= 0 || H <= 0 || W <= 0)
        return TC_ERR_ARG;
    if (!tc_mhca_att_supported(C, H * W, dtype)) return TC_ERR_UNSUPPORTED;
    {   // token offsets inside an image are formed in 32 bits
        const long long n = (long long)H * W, lmax = ldx > ldq ? (ldx > ldc ? ldx : ldc) : (ldq > ldc ? ldq : ldc);
        if (n * (lmax > ldo ? lmax : ldo) >= (1LL << 31) || lmax >= (1 << 23) || ldo >= (1 << 23)) return TC_ERR_ARG;
    }
    if (ldx % 8 || ldq % 8 || ldc % 8 || ldo % 8 || gs % 8 ||
        (((uintptr_t)xn | (uintptr_t)Wqkv | (uintptr_t)qkv | (uintptr_t)convv | (uintptr_t)o) & 15))
        return TC_ERR_ARG;
    MhcaAttDev p;
    p.xn = xn; p.Wqkv = Wqkv; p.bqkv = bqkv; p.cw[0] = w3; p.cw[1] = w5; p.cw[2] = w7; p.cb[0] = b3; p.cb[1] = b5; p.cb[2] = b7;
    p.qkv = qkv; p.convv = convv; p.o = o; p.stats = stats; p.gs = gs; p.ldx = ldx; p.ldq = ldq; p.ldc = ldc; p.ldo = ldo;
    p.B = B; p.N = H * W; p.H = H; p.W = W; p.scale = scale;
    const int Bt = groups * B;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == TC_BF16) {
        if (C == 64) return mhca_att_launch<bf16_t, 64>(p, Bt, s);
        if (C == 128) return mhca_att_launch<bf16_t, 128>(p, Bt, s);
        return mhca_att_launch<bf16_t, 320>(p, Bt, s);
    }
    if (C == 64) return mhca_att_launch<f16_t, 64>(p, Bt, s);
    if (C == 128) return mhca_att_launch<f16_t, 128>(p, Bt, s);
    return mhca_att_launch<f16_t, 320>(p, Bt, s);
}

static size_t mhca_att_bwd_smem_of(int C, int N) {
    return C == 64 ? mhca_att_bwd_smem<64>(N) : C == 128 ? mhca_att_bwd_smem<128>(N) : C == 320 ? mhca_att_bwd_smem<320>(N) : 0;
}
extern "C" int tc_mhca_att_bwd_supported(int C, int N, int dtype) {
    if (dtype != TC_BF16 && dtype != TC_F16) return 0;
    const size_t sm = mhca_att_bwd_smem_of(C, N);
    return sm > 0 && sm <= 79 * 1024 && N > 0;          // (two workgroups per CU: a launch of 3 x 16 images x 8 heads stays one wave of workgroups)
}
extern "C" int tc_mhca_att_bwd(const void* qkv, int ldq, const void* convv, int ldc, const void* go, int ldgo, const float* stats, void* dqkv,
                               int ldd, int acc_q, int acc_k, int acc_v, const void* w3, const void* w5, const void* w7, float* dw3, float* db3,
                               float* dw5, float* db5, float* dw7, float* db7, long long gs, int groups, int B, int H, int W, int C, float scale,
                               int dtype, void* stream) {
    if (!qkv || !convv || !go || !stats || !dqkv || !w3 || !w5 || !w7 || !dw3 || !db3 || !dw5 || !db5 || !dw7 || !db7 || groups <= 0 || B <= 0 ||
        H <= 0 || W <= 0)
        return TC_ERR_ARG;
    if (!tc_mhca_att_bwd_supported(C, H * W, dtype)) return TC_ERR_UNSUPPORTED;
    {
        const long long n = (long long)H * W, l1 = ldq > ldc ? ldq : ldc, l2 = ldgo > ldd ? ldgo : ldd, lmax = l1 > l2 ? l1 : l2;
        if (n * lmax >= (1LL << 31) || lmax >= (1 << 23)) return TC_ERR_ARG;
    }
    if (ldq % 8 || ldc % 8 || ldgo % 8 || ldd % 8 || (((uintptr_t)qkv | (uintptr_t)convv | (uintptr_t)go | (uintptr_t)dqkv) & 15)) return TC_ERR_ARG;
    MhcaAttBwdDev p;
    p.qkv = qkv; p.convv = convv; p.go = go; p.cw[0] = w3; p.cw[1] = w5; p.cw[2] = w7; p.stats = stats; p.dqkv = dqkv;
    p.dcw[0] = dw3; p.dcw[1] = dw5; p.dcw[2] = dw7; p.dcb[0] = db3; p.dcb[1] = db5; p.dcb[2] = db7; p.gs = gs;
    p.ldq = ldq; p.ldc = ldc; p.ldgo = ldgo; p.ldd = ldd; p.acc_q = acc_q; p.acc_k = acc_k; p.acc_v = acc_v;
    p.B = B; p.N = H * W; p.H = H; p.W = W; p.scale = scale;
    const int Bt = groups * B;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == TC_BF16) {
        if (C == 64) return mhca_att_bwd_launch<bf16_t, 64>(p, Bt, s);
        if (C == 128) return mhca_att_bwd_launch<bf16_t, 128>(p, Bt, s);
        return mhca_att_bwd_launch<bf16_t, 320>(p, Bt, s);
    }
    if (C == 64) return mhca_att_bwd_launch<f16_t, 64>(p, Bt, s);
    if (C == 128) return mhca_att_bwd_launch<f16_t, 128>(p, Bt, s);
    return mhca_att_bwd_launch<f16_t, 320>(p, Bt, s);
}

extern "C" int tc_dw_ln_supported(int C, int dtype) { return (dtype == TC_BF16 || dtype == TC_F16) && (C == 64 || C == 128 || C == 320); }
extern "C" int tc_dw_ln_fwd(const void* x, int ldx, const void* w, const void* b, long long wstride, const void* gamma, const void* beta,
                            long long gstride, void* t1, int ldt, void* xn, int ldn, float* mean, float* rstd, int groups, int B, int H, int W,
                            int C, float eps, int dtype, void* stream) {
    if (!x || !w || !b || !gamma || !beta || !t1 || !xn || !mean || !rstd || groups <= 0 || B <= 0 || H <= 0 || W <= 0) return TC_ERR_ARG;
    if (!tc_dw_ln_supported(C, dtype)) return TC_ERR_UNSUPPORTED;
    if (ldx % 8 || ldt % 8 || ldn % 8 || (((uintptr_t)x | (uintptr_t)t1 | (uintptr_t)xn) & 15)) return TC_ERR_ARG;
    DwLnDev p;
    p.x = x; p.w = w; p.b = b; p.gamma = gamma; p.beta = beta; p.t1 = t1; p.xn = xn; p.mean = mean; p.rstd = rstd;
    p.wstride = wstride; p.gstride = gstride; p.ldx = ldx; p.ldt = ldt; p.ldn = ldn; p.B = B; p.H = H; p.W = W;
    p.rows_per_group = B * H * W; p.eps = eps;
    const dim3 grid((p.rows_per_group + 31) / 32, groups);
    hipStream_t s = (hipStream_t)stream;
#define DWLN(TT, CC) hipLaunchKernelGGL((dw_ln_fwd_kernel<TT, CC>), grid, dim3(256), 0, s, p)
    if (dtype == TC_BF16) { if (C == 64) DWLN(bf16_t, 64); else if (C == 128) DWLN(bf16_t, 128); else DWLN(bf16_t, 320); }
    else { if (C == 64) DWLN(f16_t, 64); else if (C == 128) DWLN(f16_t, 128); else DWLN(f16_t, 320); }
#undef DWLN
    return tc_launch_status();
}
