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
        if (smem > 64 * 1024) hipFuncSetAttribute((const void*)factor_att_fwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
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
        if (smem > 64 * 1024) hipFuncSetAttribute((const void*)factor_att_bwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((factor_att_bwd_kernel<T>), dim3(Bt * heads), dim3(256), smem, (hipStream_t)stream, (const T*)q, (const T*)k,
                           (const T*)v, ld, (const T*)convv, ldc, (const T*)go, ldgo, stats, (T*)dq, (T*)dk, (T*)dv, ldd, acc_q, acc_k,
                           acc_v, (T*)dconvv, lddc, N, Ch, heads, scale);
    });
    return tc_launch_status();
}
