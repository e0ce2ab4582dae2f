// LayerNorm (+ exact GELU) and BatchNorm (+ Hardswish / IFF activation / residual) for token-major activations.
// HBM-bound kernels: one wavefront per LayerNorm row with the row held in registers (two-pass statistics via
// wave shuffles); BatchNorm statistics as deterministic per-chunk partials that the apply kernel folds itself.
#include "tc_common.h"
#include <cstdlib>

namespace {

constexpr int LN_MAXC = 2048;
#ifndef LN_FOLD
#define LN_FOLD 8
#endif
constexpr int LN_BWD_MAX_BLOCKS = 4096, LN_BWD_ROWS_PER_GROUP = 4;   // fused dx + parameter-gradient launch
inline int ln_bwd_wg_min() { static const int v = getenv("TC_LN_WG_MIN") ? atoi(getenv("TC_LN_WG_MIN")) : 1024; return v < 16 ? 16 : v; }   // 512 with the fold at the tail; deferred: 256 / 512 / 1024 -> 12.11 / 12.07 / 12.04 ms
inline int ln_bwd_blocks(bool deferred) {   // A/B switch (<= LN_BWD_MAX_BLOCKS, which sizes the scratch)
    // launches that fold at their own tail: 1024 (every block is one more partial to park and fold in-kernel); deferred fold: 4096
    // (1024 / 2048 / 4096 -> 11.87 / 11.85 / 11.84 ms per step)
    static const int v = getenv("TC_LN_BWD_BLOCKS") ? atoi(getenv("TC_LN_BWD_BLOCKS")) : 0;
    const int w = v > 0 ? v : (deferred ? LN_BWD_MAX_BLOCKS : 1024);
    return w < 16 ? 16 : (w > LN_BWD_MAX_BLOCKS ? LN_BWD_MAX_BLOCKS : w);
}

// ---------------------------------------------------------------------------------------------- LayerNorm
// A group of GS lanes (16 / 32 / 64) owns one row, each lane NV float4s of it; a 256-thread block therefore works on
// 256/GS rows at a time (C = 64 rows use 16 lanes: 4 rows per wave instead of 48 idle lanes).  Statistics by xor-shuffles
// inside the group.  The backward keeps per-lane partial dgamma/dbeta over a grid-stride loop of rows, folds the groups
// of a block through LDS and issues one atomic per channel per block (grid capped so the atomics do not serialise).
// Optional source addressing of a LayerNorm whose rows are the pixels of a pixel-shuffled map (PatchExpand / FinalPatchExpand_X4,
// MSTr.py:196-199,222-225): output pixel (b, h p + p1, w p + p2) is the C-wide chunk (p1 p + p2) of row (b, h, w) of the un-shuffled
// [B H W, p p C] matrix -- the shuffle is an address computation here, not a copy (the X4 copy alone was 206 MB of traffic).
struct LnMap { int p, H, W; };
__device__ __forceinline__ long long ln_row_off(int row, const LnMap& m, int ld, int C) {
    if (m.p == 0) return (long long)row * ld;
    const int Wp = m.W * m.p, Hp = m.H * m.p;
    const int ow = row % Wp, t = row / Wp, oh = t % Hp, b = t / Hp;
    return ((long long)(b * m.H + oh / m.p) * m.W + ow / m.p) * ld + ((oh % m.p) * m.p + ow % m.p) * C;
}

template <int GS> __device__ __forceinline__ float group_sum(float v) { return tc_group_sum<GS>(v); }

// RPT consecutive rows per lane group and iteration: their loads are all issued before the first reduction (narrow rows are
// 128-256 bytes: one row per group per iteration left a single 8-byte load per lane in flight and ran at ~2 TB/s).
template <typename T, int GS, int NV, int RPT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ gamma,
                                                     const T* __restrict__ beta, T* __restrict__ y, int ldy,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                     int C, float eps, int act, long long pstride, LnMap map) {
    constexpr int RPB = 256 / GS;
    const int gl = threadIdx.x % GS, gi = threadIdx.x / GS;
    const int sst = (rstd == mean + 1) ? 2 : 1;                  // interleaved statistics: [rows][2] (mean, rstd) pairs
    {   // group = blockIdx.y: `rows` rows each, parameters pstride apart
        const long long g = blockIdx.y;
        x += g * rows * ldx; y += g * rows * ldy; mean += g * rows * sst; rstd += g * rows * sst; gamma += g * pstride; beta += g * pstride;
    }
    const int nv = C >> 2;
    const float invC = 1.0f / (float)C;
    float4 g[NV], b[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = gl + i * GS;
        if (q < nv) { g[i] = ld4<T>(gamma + q * 4); b[i] = ld4<T>(beta + q * 4); }
    }
    for (int row0 = (blockIdx.x * RPB + gi) * RPT; row0 < rows; row0 += gridDim.x * RPB * RPT) {
        float4 v[RPT][NV];
        float mu[RPT], rs[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int row = min(row0 + r, rows - 1);              // rows past the end re-read the last row (no divergent loads)
            const T* xr = x + ln_row_off(row, map, ldx, C);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int q = gl + i * GS;
                v[r][i] = q < nv ? ld4<T>(xr + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) s += v[r][i].x + v[r][i].y + v[r][i].z + v[r][i].w;
            mu[r] = group_sum<GS>(s) * invC;
            float s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int q = gl + i * GS;
                if (q < nv) {
                    const float a = v[r][i].x - mu[r], bb = v[r][i].y - mu[r], c = v[r][i].z - mu[r], d = v[r][i].w - mu[r];
                    s2 += a * a + bb * bb + c * c + d * d;
                }
            }
            rs[r] = rsqrtf(group_sum<GS>(s2) * invC + eps);
        }
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int row = row0 + r;
            if (row >= rows) break;
            if (gl == 0) { mean[(long long)row * sst] = mu[r]; rstd[(long long)row * sst] = rs[r]; }
            T* yr = y + (long long)row * ldy;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int q = gl + i * GS;
                if (q < nv) {
                    float4 o;
                    o.x = (v[r][i].x - mu[r]) * rs[r] * g[i].x + b[i].x; o.y = (v[r][i].y - mu[r]) * rs[r] * g[i].y + b[i].y;
                    o.z = (v[r][i].z - mu[r]) * rs[r] * g[i].z + b[i].z; o.w = (v[r][i].w - mu[r]) * rs[r] * g[i].w + b[i].w;
                    if (act == TC_ACT_GELU) { o.x = gelu_fT<T>(o.x); o.y = gelu_fT<T>(o.y); o.z = gelu_fT<T>(o.z); o.w = gelu_fT<T>(o.w); }
                    st4<T>(yr + q * 4, o);
                }
            }
        }
    }
}

// The parameter-gradient tail of the LayerNorm backward kernels: the workgroup's per-lane-group column sums (red[RPB][2][C], filled by the
// caller) are added up and leave as atomics, as a parked partial row (defer: ln_fold_kernel adds them, one launch per backward leg), or
// through the two-level fold below.
template <int RPB>
__device__ __forceinline__ void ln_bwd_param_tail(const float* red, const int C, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                  float* __restrict__ partial, int* __restrict__ cnt, const int defer) {
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f, bb = 0.f;
#pragma unroll
        for (int w = 0; w < RPB; ++w) { a += red[(w * 2 + 0) * C + c]; bb += red[(w * 2 + 1) * C + c]; }
        if (partial) {                       // parked for the 16-way fold below, or (defer) for ln_fold_kernel: one launch per backward leg
            float* pp = partial + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 2 * C;
            if (defer) { pp[c] = a; pp[C + c] = bb; }
            else {
                __hip_atomic_store(pp + c, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(pp + C + c, bb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            atomicAdd(dgamma + c, a);
            atomicAdd(dbeta + c, bb);
        }
    }
    
    if (!partial || defer) return;
    // Two-level fold (the GEMM split-K fix-up protocol): LN_FOLD consecutive workgroups share an arrival counter (8 measured best of 2-32: the last arriver reads its group serially); the last to arrive
    // adds the group's partial rows and is the only one that touches dgamma / dbeta atomically (32 contributors per word instead of
    // 1024, and no second launch).
    constexpr int FG = LN_FOLD;
    const int grp = blockIdx.x / FG, ngrp = (gridDim.x + FG - 1) / FG, gm = min(FG, (int)gridDim.x - grp * FG);
    __builtin_amdgcn_s_waitcnt(0);                          // vmcnt(0): THIS thread's write-through stores have been acknowledged
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // (the barrier alone does not wait for stores in flight)
    __syncthreads();
    __shared__ int s_last;
    if (threadIdx.x == 0) {
        int* cn = cnt + blockIdx.y * ngrp + grp;
        const int old = atomicAdd(cn, 1);
        s_last = (old == gm - 1);
        if (s_last) atomicExch(cn, 0);
    }
    __syncthreads();
    
    if (!s_last) return;
    const float* pg = partial + ((long long)blockIdx.y * gridDim.x + grp * FG) * 2 * C;
    for (int c = threadIdx.x; c < 2 * C; c += 256) {
        float tmp[FG];
#pragma unroll
        for (int m = 0; m < FG; ++m) tmp[m] = m < gm ? __hip_atomic_load(pg + (long long)m * 2 * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
        float v = 0.f;
#pragma unroll
        for (int m = 0; m < FG; ++m) v += tmp[m];
        atomicAdd((c < C ? dgamma + c : dbeta + (c - C)), v);
    }
    
}

#ifdef TC_LN_TIMING
// phase stamps of ln_bwd_kernel (experiment builds): every 61st workgroup writes its own row: GS, rows, then cycles of
// parameter loads / row loop / LDS partials / parked partial + arrival / fold
__device__ unsigned long long g_ln_dbg[1024 * 8];
__device__ unsigned int g_ln_dbg_n;
#define LSTAMP(k) do { if (ls_ >= 0) { const long long t_ = __builtin_readcyclecounter(); g_ln_dbg[ls_ * 8 + 2 + (k)] = (unsigned long long)(t_ - lt_); lt_ = t_; } } while (0)
#define LSTAMP_INIT() int ls_ = -1; long long lt_ = 0; \
    if (threadIdx.x == 0 && (blockIdx.x + 17 * blockIdx.y) % 61 == 0) { ls_ = (int)(atomicAdd(&g_ln_dbg_n, 1u) & 1023u); \
        g_ln_dbg[ls_ * 8] = GS + 1000ull * C; g_ln_dbg[ls_ * 8 + 1] = (unsigned long long)rows; lt_ = __builtin_readcyclecounter(); }
#else
#define LSTAMP(k)
#define LSTAMP_INIT()
#endif
template <typename T, int GS, int NV, int RPT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ x, int ldx,
                                                     const T* __restrict__ gamma, const T* __restrict__ beta,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     T* __restrict__ dx, int lddx, const T* __restrict__ dres, int ldres,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int C,
                                                     int act, long long pstride, float* __restrict__ partial, int* __restrict__ cnt, LnMap map, int defer) {
    constexpr int RPB = 256 / GS;
    extern __shared__ float red[];           // [RPB][2][C]
    const int gl = threadIdx.x % GS, gi = threadIdx.x / GS;
    {
        const long long g = blockIdx.y;
        dy += g * rows * lddy; x += g * rows * ldx; dx += g * rows * lddx; mean += g * rows; rstd += g * rows;
        if (dres) dres += g * rows * ldres;
        gamma += g * pstride; beta += g * pstride;
        if (dgamma) { dgamma += g * pstride; dbeta += g * pstride; }
    }
    LSTAMP_INIT();
    const int nv = C >> 2;
    float4 g[NV], b[NV], ag[NV], ab[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = gl + i * GS;
        ag[i] = ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < nv) { g[i] = ld4<T>(gamma + q * 4); b[i] = ld4<T>(beta + q * 4); }
    }
    const float invC = 1.0f / (float)C;
    LSTAMP(0);
    // RPT consecutive rows per lane group and iteration, all their loads issued before the first reduction (see ln_fwd_kernel)
    for (int row0 = (blockIdx.x * RPB + gi) * RPT; row0 < rows; row0 += gridDim.x * RPB * RPT) {
        float4 xh[RPT][NV], gg[RPT][NV], rr[RPT][NV];
        float mu[RPT], rs[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int row = min(row0 + r, rows - 1);
            mu[r] = mean[row]; rs[r] = rstd[row];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int q = gl + i * GS;
                const bool in = q < nv;
                xh[r][i] = in ? ld4<T>(x + ln_row_off(row, map, ldx, C) + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                gg[r][i] = in ? ld4<T>(dy + (long long)row * lddy + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (dres) rr[r][i] = in ? ld4<T>(dres + ln_row_off(row, map, ldres, C) + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const bool live = row0 + r < rows;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int q = gl + i * GS;
                if (q < nv) {
                    float4 xv = xh[r][i], d = gg[r][i];
                    xv.x = (xv.x - mu[r]) * rs[r]; xv.y = (xv.y - mu[r]) * rs[r]; xv.z = (xv.z - mu[r]) * rs[r]; xv.w = (xv.w - mu[r]) * rs[r];
                    if (act == TC_ACT_GELU) {
                        d.x *= gelu_grad_fT<T>(xv.x * g[i].x + b[i].x); d.y *= gelu_grad_fT<T>(xv.y * g[i].y + b[i].y);
                        d.z *= gelu_grad_fT<T>(xv.z * g[i].z + b[i].z); d.w *= gelu_grad_fT<T>(xv.w * g[i].w + b[i].w);
                    }
                    if (dgamma && live) {
                        ag[i].x += d.x * xv.x; ag[i].y += d.y * xv.y; ag[i].z += d.z * xv.z; ag[i].w += d.w * xv.w;
                        ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
                    }
                    d.x *= g[i].x; d.y *= g[i].y; d.z *= g[i].z; d.w *= g[i].w;
                    s1 += d.x + d.y + d.z + d.w;
                    s2 += d.x * xv.x + d.y * xv.y + d.z * xv.z + d.w * xv.w;
                    xh[r][i] = xv; gg[r][i] = d;
                }
            }
            s1 = group_sum<GS>(s1) * invC; s2 = group_sum<GS>(s2) * invC;
            if (!live) continue;
            T* dxr = dx + ln_row_off(row0 + r, map, lddx, C);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int q = gl + i * GS;
                if (q < nv) {
                    float4 o;
                    o.x = rs[r] * (gg[r][i].x - s1 - xh[r][i].x * s2); o.y = rs[r] * (gg[r][i].y - s1 - xh[r][i].y * s2);
                    o.z = rs[r] * (gg[r][i].z - s1 - xh[r][i].z * s2); o.w = rs[r] * (gg[r][i].w - s1 - xh[r][i].w * s2);
                    if (dres) { o.x += rr[r][i].x; o.y += rr[r][i].y; o.z += rr[r][i].z; o.w += rr[r][i].w; }
                    st4<T>(dxr + q * 4, o);
                }
            }
        }
    }
    LSTAMP(1);
    if (!dgamma) return;                     // dx-only launch: parameter gradients come from ln_param_grad_kernel
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = gl + i * GS;
        if (q < nv) {
            float* r0 = red + (gi * 2 + 0) * C + q * 4; float* r1 = red + (gi * 2 + 1) * C + q * 4;
            r0[0] = ag[i].x; r0[1] = ag[i].y; r0[2] = ag[i].z; r0[3] = ag[i].w;
            r1[0] = ab[i].x; r1[1] = ab[i].y; r1[2] = ab[i].z; r1[3] = ab[i].w;
        }
    }
    ln_bwd_param_tail<RPB>(red, C, dgamma, dbeta, partial, cnt, defer);
}

// 16-bit rows of 64 / 128 / 256 / 512 channels: a lane owns EIGHT channels (one 16-byte piece) of a row, GS = C / 8 lanes a row, RPT rows per
// lane group in flight -- ln_bwd_kernel's 8-byte pieces left 16-32 bytes per lane in flight and ran the 25 MB maps of the last decoder
// stage at 1.4 TB/s (56 us a launch).  The raw pieces of all RPT rows are requested before the first one is unpacked.  Same arithmetic in the
// same order per row as ln_bwd_kernel (statistics over the group by the same shuffles), same parameter-gradient tail.
typedef unsigned lnu4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ void ln_unpack8(const lnu4& r, float* o) {
    unpack2<T>(r.x, o[0], o[1]); unpack2<T>(r.y, o[2], o[3]); unpack2<T>(r.z, o[4], o[5]); unpack2<T>(r.w, o[6], o[7]);
}
template <typename T, int GS, int RPT>
__global__ __launch_bounds__(256) void ln_bwd_v8_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ x, int ldx,
                                                        const T* __restrict__ gamma, const T* __restrict__ beta,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        T* __restrict__ dx, int lddx, const T* __restrict__ dres, int ldres,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta, int rows,
                                                        long long pstride, float* __restrict__ partial, int* __restrict__ cnt, LnMap map, int defer) {
    constexpr int RPB = 256 / GS, C = GS * 8;
    extern __shared__ float red[];           // [RPB][2][C]
    const int gl = threadIdx.x % GS, gi = threadIdx.x / GS;
    {
        const long long g = blockIdx.y;
        dy += g * rows * lddy; x += g * rows * ldx; dx += g * rows * lddx; mean += g * rows; rstd += g * rows;
        if (dres) dres += g * rows * ldres;
        gamma += g * pstride; beta += g * pstride;
        if (dgamma) { dgamma += g * pstride; dbeta += g * pstride; }
    }
    float g[8], ag[8], ab[8];
    ln_unpack8<T>(*reinterpret_cast<const lnu4*>(gamma + gl * 8), g);
#pragma unroll
    for (int j = 0; j < 8; ++j) ag[j] = ab[j] = 0.f;
    constexpr float invC = 1.0f / (float)C;
    const bool has_res = dres != nullptr;
    for (int row0 = (blockIdx.x * RPB + gi) * RPT; row0 < rows; row0 += gridDim.x * RPB * RPT) {
        lnu4 rx[RPT], rd[RPT], rr[RPT];
        float mu[RPT], rs[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int row = min(row0 + r, rows - 1);
            rx[r] = *reinterpret_cast<const lnu4*>(x + ln_row_off(row, map, ldx, C) + gl * 8);
            rd[r] = *reinterpret_cast<const lnu4*>(dy + (long long)row * lddy + gl * 8);
            if (has_res) rr[r] = *reinterpret_cast<const lnu4*>(dres + ln_row_off(row, map, ldres, C) + gl * 8);
            mu[r] = mean[row]; rs[r] = rstd[row];
        }
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const bool live = row0 + r < rows;
            float xv[8], d[8];
            ln_unpack8<T>(rx[r], xv); ln_unpack8<T>(rd[r], d);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[j] = (xv[j] - mu[r]) * rs[r];
            if (dgamma && live) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { ag[j] += d[j] * xv[j]; ab[j] += d[j]; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] *= g[j];
            // (the 4-wide kernel's order: a lane adds its four values of a piece, pieces in turn)
            s1 = ((d[0] + d[1]) + d[2]) + d[3]; s1 += ((d[4] + d[5]) + d[6]) + d[7];
            s2 = ((d[0] * xv[0] + d[1] * xv[1]) + d[2] * xv[2]) + d[3] * xv[3]; s2 += ((d[4] * xv[4] + d[5] * xv[5]) + d[6] * xv[6]) + d[7] * xv[7];
            s1 = group_sum<GS>(s1) * invC; s2 = group_sum<GS>(s2) * invC;
            if (!live) continue;
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = rs[r] * (d[j] - s1 - xv[j] * s2);
            if (has_res) {
                float q[8];
                ln_unpack8<T>(rr[r], q);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += q[j];
            }
            lnu4 w;
            w.x = pack2<T>(o[0], o[1]); w.y = pack2<T>(o[2], o[3]); w.z = pack2<T>(o[4], o[5]); w.w = pack2<T>(o[6], o[7]);
            *reinterpret_cast<lnu4*>(dx + ln_row_off(row0 + r, map, lddx, C) + gl * 8) = w;
        }
    }
    if (!dgamma) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[(gi * 2 + 0) * C + gl * 8 + j] = ag[j];
        red[(gi * 2 + 1) * C + gl * 8 + j] = ab[j];
    }
    ln_bwd_param_tail<RPB>(red, C, dgamma, dbeta, partial, cnt, defer);
}

// The forward on the same rows (16-bit, C = 64 / 128 / 256 / 512, no activation): eight channels per lane, RPT rows per lane group in flight
// (ln_fwd_kernel's 8-byte pieces stream the 100 MB FinalPatchExpand map at 3.5 TB/s).  Same two-pass statistics, same order per row.
template <typename T, int GS, int RPT>
__global__ __launch_bounds__(256) void ln_fwd_v8_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ gamma, const T* __restrict__ beta,
                                                        T* __restrict__ y, int ldy, float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                        float eps, long long pstride, LnMap map) {
    constexpr int RPB = 256 / GS, C = GS * 8;
    const int gl = threadIdx.x % GS, gi = threadIdx.x / GS;
    const int sst = (rstd == mean + 1) ? 2 : 1;
    {
        const long long g = blockIdx.y;
        x += g * rows * ldx; y += g * rows * ldy; mean += g * rows * sst; rstd += g * rows * sst; gamma += g * pstride; beta += g * pstride;
    }
    float gm[8], bt[8];
    ln_unpack8<T>(*reinterpret_cast<const lnu4*>(gamma + gl * 8), gm);
    ln_unpack8<T>(*reinterpret_cast<const lnu4*>(beta + gl * 8), bt);
    constexpr float invC = 1.0f / (float)C;
    for (int row0 = (blockIdx.x * RPB + gi) * RPT; row0 < rows; row0 += gridDim.x * RPB * RPT) {
        lnu4 raw[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) raw[r] = *reinterpret_cast<const lnu4*>(x + ln_row_off(min(row0 + r, rows - 1), map, ldx, C) + gl * 8);
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            float v[8];
            ln_unpack8<T>(raw[r], v);
            float s = ((v[0] + v[1]) + v[2]) + v[3];
            s += ((v[4] + v[5]) + v[6]) + v[7];
            const float mu = group_sum<GS>(s) * invC;
            float s2 = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[j] - mu; s2 += d * d; }
            const float rs = rsqrtf(group_sum<GS>(s2) * invC + eps);
            const int row = row0 + r;
            if (row >= rows) continue;
            if (gl == 0) { mean[(long long)row * sst] = mu; rstd[(long long)row * sst] = rs; }
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (v[j] - mu) * rs * gm[j] + bt[j];
            lnu4 w;
            w.x = pack2<T>(o[0], o[1]); w.y = pack2<T>(o[2], o[3]); w.z = pack2<T>(o[4], o[5]); w.w = pack2<T>(o[6], o[7]);
            *reinterpret_cast<lnu4*>(y + (long long)row * ldy + gl * 8) = w;
        }
    }
}

// dgamma[c] += sum_rows dz * xhat ; dbeta[c] += sum_rows dz   as a column reduction (thread = 4 channels x row lane): fully
// parallel over rows, a few atomics per block; runs on the weight-gradient stream next to the dx kernel.
template <typename T>
__global__ __launch_bounds__(256) void ln_param_grad_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ x, int ldx,
                                                            const T* __restrict__ gamma, const T* __restrict__ beta,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int C, int act,
                                                            long long pstride) {
    __shared__ float4 sg[16][16], sb[16][16];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = blockIdx.y * 64 + tx * 4;
    {
        const long long g = blockIdx.z;
        dy += g * rows * lddy; x += g * rows * ldx; mean += g * rows; rstd += g * rows;
        gamma += g * pstride; beta += g * pstride; dgamma += g * pstride; dbeta += g * pstride;
    }
    float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = ag;
    if (c < C) {
        const float4 gm = ld4<T>(gamma + c), bt = ld4<T>(beta + c);
        for (int r = blockIdx.x * 16 + ty; r < rows; r += gridDim.x * 16) {
            const float mu = mean[r], rs = rstd[r];
            float4 xv = ld4<T>(x + (long long)r * ldx + c), d = ld4<T>(dy + (long long)r * lddy + c);
            xv.x = (xv.x - mu) * rs; xv.y = (xv.y - mu) * rs; xv.z = (xv.z - mu) * rs; xv.w = (xv.w - mu) * rs;
            if (act == TC_ACT_GELU) {
                d.x *= gelu_grad_fT<T>(xv.x * gm.x + bt.x); d.y *= gelu_grad_fT<T>(xv.y * gm.y + bt.y);
                d.z *= gelu_grad_fT<T>(xv.z * gm.z + bt.z); d.w *= gelu_grad_fT<T>(xv.w * gm.w + bt.w);
            }
            ag.x += d.x * xv.x; ag.y += d.y * xv.y; ag.z += d.z * xv.z; ag.w += d.w * xv.w;
            ab.x += d.x; ab.y += d.y; ab.z += d.z; ab.w += d.w;
        }
    }
    sg[ty][tx] = ag; sb[ty][tx] = ab;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int cc = threadIdx.x & 63, which = threadIdx.x >> 6, q = cc >> 2, e = cc & 3;
        if (blockIdx.y * 64 + cc < C) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float4 t = which ? sb[r][q] : sg[r][q]; v += (e == 0 ? t.x : e == 1 ? t.y : e == 2 ? t.z : t.w); }
            atomicAdd((which ? dbeta : dgamma) + blockIdx.y * 64 + cc, v);
        }
    }
}

// (GS, NV) by row width: quads = C/4
#define TC_LN_DISPATCH(quads, CALL)                                     \
    if ((quads) <= 16) { CALL(16, 1); }                                 \
    else if ((quads) <= 32) { CALL(32, 1); }                            \
    else if ((quads) <= 64) { CALL(64, 1); }                            \
    else if ((quads) <= 128) { CALL(64, 2); }                           \
    else if ((quads) <= 256) { CALL(64, 4); }                           \
    else if ((quads) <= 320) { CALL(64, 5); }                           \
    else { CALL(64, 8); }

// ---------------------------------------------------------------------------------------------- BatchNorm
// scratch layout: shift[C] | S1[nchunk][C] | S2[nchunk][C]
// Row chunks of the statistics pass.  EVERY workgroup of the apply kernels folds all chunks of its 64 channels, so the chunk count is
// a per-workgroup read of nchunk x 512 bytes: at 128 chunks the apply pass of a [12544, 64] map (1.6 MB) read 12.5 MB of partials.
#ifndef BN_MAX_CHUNKS
#define BN_MAX_CHUNKS 128
#endif
inline int bn_chunk_cap() { static const int v = getenv("TC_BN_CHUNKS") ? atoi(getenv("TC_BN_CHUNKS")) : BN_MAX_CHUNKS; return v < 1 ? 1 : (v > 128 ? 128 : v); }
inline int bn_nchunk(int rows) { int n = (rows + 63) / 64; const int cap = bn_chunk_cap(); return n < 1 ? 1 : (n > cap ? cap : n); }

// mode 0 (forward stats):  S1 = sum(x - shift), S2 = sum((x-shift)^2), shift = x[0, c]
// mode 1 (backward sums):  S1 = sum(dz), S2 = sum(dz * xhat), dz = dy * act'(z)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dy, int lddy,
                                                         const T* __restrict__ gamma, const T* __restrict__ beta,
                                                         const float* __restrict__ save_mean,
                                                         const float* __restrict__ save_rstd, float* __restrict__ scratch,
                                                         int rows, int C, int act) {
    // 16 channel quads x 16 row lanes per workgroup (the layout of the apply kernels): 8/16-byte loads instead of one element per lane
    __shared__ float r1[16][64], r2[16][64];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = blockIdx.y * 64 + tx * 4;
    const int nchunk = gridDim.x, chunk = blockIdx.x;
    const int per = (rows + nchunk - 1) / nchunk;
    const int rbeg = chunk * per, rend = min(rows, rbeg + per);
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        if (MODE == 0) {
            const float4 sh = ld4<T>(x + c);
            if (chunk == 0 && ty == 0) { scratch[c] = sh.x; scratch[c + 1] = sh.y; scratch[c + 2] = sh.z; scratch[c + 3] = sh.w; }
            for (int r = rbeg + ty; r < rend; r += 16) {
                const float4 v = ld4<T>(x + (long long)r * ldx + c);
                const float a0 = v.x - sh.x, a1 = v.y - sh.y, a2 = v.z - sh.z, a3 = v.w - sh.w;
                s1[0] += a0; s1[1] += a1; s1[2] += a2; s1[3] += a3;
                s2[0] += a0 * a0; s2[1] += a1 * a1; s2[2] += a2 * a2; s2[3] += a3 * a3;
            }
        } else {
            const float4 mu = *reinterpret_cast<const float4*>(save_mean + c), rs = *reinterpret_cast<const float4*>(save_rstd + c);
            const float4 g = ld4<T>(gamma + c), b = ld4<T>(beta + c);
            const float mus[4] = {mu.x, mu.y, mu.z, mu.w}, rss[4] = {rs.x, rs.y, rs.z, rs.w};
            const float gs[4] = {g.x, g.y, g.z, g.w}, bs[4] = {b.x, b.y, b.z, b.w};
            for (int r = rbeg + ty; r < rend; r += 16) {
                const float4 xv = ld4<T>(x + (long long)r * ldx + c), dv = ld4<T>(dy + (long long)r * lddy + c);
                const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (xs[j] - mus[j]) * rss[j];
                    float d = ds[j];
                    if (act != TC_ACT_NONE) d *= act_grad(act, xh * gs[j] + bs[j]);
                    s1[j] += d; s2[j] += d * xh;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { r1[ty][tx * 4 + j] = s1[j]; r2[ty][tx * 4 + j] = s2[j]; }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int cc = threadIdx.x, ch = blockIdx.y * 64 + cc;
        if (ch < C) {
            float a = 0.f, b2 = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) { a += r1[w][cc]; b2 += r2[w][cc]; }
            scratch[C + chunk * C + ch] = a;
            scratch[C + nchunk * C + chunk * C + ch] = b2;
        }
    }
}

// A quarter (part = 0..3) of the chunk partials of one channel: every workgroup of the apply kernels starts with this fold, and as a loop
// of dependent loads over a run-time count it was a chain of ~nchunk / 4 L2 round trips at the head of each of them (the GEMM epilogue leaves
// one partial per 64-row tile: 196 for a 12544-row map).  Eight loads per array in flight before the first add.
__device__ __forceinline__ void bn_fold_partials(const float* __restrict__ scratch, int nchunk, int C, int ch, int part, float& s1, float& s2) {
    const float* p1 = scratch + C + ch;
    const float* p2 = scratch + C + (long long)nchunk * C + ch;
    int k = part;
    for (; k + 28 < nchunk; k += 32) {
        float a[8], b[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) { a[m] = p1[(long long)(k + 4 * m) * C]; b[m] = p2[(long long)(k + 4 * m) * C]; }
#pragma unroll
        for (int m = 0; m < 8; ++m) { s1 += a[m]; s2 += b[m]; }
    }
    for (; k < nchunk; k += 4) { s1 += p1[(long long)k * C]; s2 += p2[(long long)k * C]; }
}

// The forward statistics: the producers' shift is the running mean (GEMM / RIPM epilogues) or the map's first row (bn_partial_kernel), which
// a distribution shift can leave many standard deviations off the batch mean -- and var = S2 / n - (S1 / n)^2 then cancels.  The per-chunk
// sums are exact to their own fp32 rounding (independent between chunks), so folding them and taking the difference in fp64 keeps the
// relative error of the variance at (mean - shift)^2 / var x 1e-8 instead of x 1e-6 x sqrt(chunks)  (round 6; tests/test_ops_gpu.py::
// test_batchnorm_statistics_far_from_the_shift: batch mean = 50 standard deviations).
__device__ __forceinline__ void bn_fold_partials(const float* __restrict__ scratch, int nchunk, int C, int ch, int part, double& s1, double& s2) {
    const float* p1 = scratch + C + ch;
    const float* p2 = scratch + C + (long long)nchunk * C + ch;
    int k = part;
    for (; k + 28 < nchunk; k += 32) {
        float a[8], b[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) { a[m] = p1[(long long)(k + 4 * m) * C]; b[m] = p2[(long long)(k + 4 * m) * C]; }
#pragma unroll
        for (int m = 0; m < 8; ++m) { s1 += (double)a[m]; s2 += (double)b[m]; }
    }
    for (; k < nchunk; k += 4) { s1 += (double)p1[(long long)k * C]; s2 += (double)p2[(long long)k * C]; }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ gamma,
                                                       const T* __restrict__ beta, float* __restrict__ running_mean,
                                                       float* __restrict__ running_var, const T* __restrict__ res, int ldres,
                                                       T* __restrict__ y, int ldy, float* __restrict__ save_mean,
                                                       float* __restrict__ save_rstd, const float* __restrict__ scratch,
                                                       int nchunk, int rows, int C, float eps, float momentum, int training,
                                                       int act) {
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;       // 16 channel quads x 16 row lanes
    const int c = blockIdx.y * 64 + tx * 4;
    __shared__ double psum[2][4][64];
    __shared__ float stat[2][64];
    if (training) {
        // fold the per-chunk partials once per block: thread (channel = tid % 64, quarter = tid / 64); fp64 sums and difference (see above)
        const int cc = threadIdx.x & 63, part = threadIdx.x >> 6, ch = blockIdx.y * 64 + cc;
        // (fp32 storage = the parity path: its statistics pass shifts by the map's first row, which is well conditioned, and the fp32 fold
        //  reproduces the reference's own fp32 arithmetic more closely -- with the fp64 fold one stem-weight gradient sample of the stage4_se
        //  fixture moved from inside 2e-6 to 4.2e-6 of the reference's value; 16-bit storage, where the GEMM / RIPM epilogues shift by the
        //  running mean, folds in fp64)
        constexpr bool F32FOLD = std::is_same<T, float>::value;
        double s1 = 0.0, s2 = 0.0;
        if (ch < C) {
            if constexpr (F32FOLD) { float f1 = 0.f, f2 = 0.f; bn_fold_partials(scratch, nchunk, C, ch, part, f1, f2); s1 = f1; s2 = f2; }
            else bn_fold_partials(scratch, nchunk, C, ch, part, s1, s2);
        }
        psum[0][part][cc] = s1; psum[1][part][cc] = s2;
        __syncthreads();
        if (threadIdx.x < 64 && ch < C) {
            s1 = (psum[0][0][cc] + psum[0][1][cc]) + (psum[0][2][cc] + psum[0][3][cc]);
            s2 = (psum[1][0][cc] + psum[1][1][cc]) + (psum[1][2][cc] + psum[1][3][cc]);
            float var, mean_;
            if constexpr (F32FOLD) {
                const float f1 = (float)psum[0][0][cc] + (float)psum[0][1][cc] + (float)psum[0][2][cc] + (float)psum[0][3][cc];     // (round 5's order)
                const float f2 = (float)psum[1][0][cc] + (float)psum[1][1][cc] + (float)psum[1][2][cc] + (float)psum[1][3][cc];
                const float m1f = f1 / (float)rows;
                var = fmaxf(f2 / (float)rows - m1f * m1f, 0.f);
                mean_ = scratch[ch] + m1f;
            } else {
                const double m1 = s1 / (double)rows;
                var = (float)fmax(s2 / (double)rows - m1 * m1, 0.0);
                mean_ = (float)((double)scratch[ch] + m1);
            }
            const float rstd_ = rsqrtf(var + eps);
            stat[0][cc] = mean_; stat[1][cc] = rstd_;
            if (blockIdx.x == 0) {
                save_mean[ch] = mean_; save_rstd[ch] = rstd_;
                const float unbiased = rows > 1 ? var * (float)rows / (float)(rows - 1) : var;
                running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * mean_;
                running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * unbiased;
            }
        }
        __syncthreads();
    }
    if (c >= C) return;
    float mu[4], rs[4];
    if (training) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { mu[j] = stat[0][tx * 4 + j]; rs[j] = stat[1][tx * 4 + j]; }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { mu[j] = running_mean[c + j]; rs[j] = rsqrtf(running_var[c + j] + eps); }
    }
    const float4 g = ld4<T>(gamma + c), b = ld4<T>(beta + c);
    for (int r = blockIdx.x * 16 + ty; r < rows; r += gridDim.x * 16) {
        float4 v = ld4<T>(x + (long long)r * ldx + c);
        v.x = (v.x - mu[0]) * rs[0] * g.x + b.x; v.y = (v.y - mu[1]) * rs[1] * g.y + b.y;
        v.z = (v.z - mu[2]) * rs[2] * g.z + b.z; v.w = (v.w - mu[3]) * rs[3] * g.w + b.w;
        if (act != TC_ACT_NONE) { v.x = apply_act(act, v.x); v.y = apply_act(act, v.y); v.z = apply_act(act, v.z); v.w = apply_act(act, v.w); }
        if (res) { const float4 q = ld4<T>(res + (long long)r * ldres + c); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
        st4<T>(y + (long long)r * ldy + c, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ x, int ldx,
                                                           const T* __restrict__ gamma, const T* __restrict__ beta,
                                                           const float* __restrict__ save_mean,
                                                           const float* __restrict__ save_rstd, T* __restrict__ dx, int lddx,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           const float* __restrict__ scratch, int nchunk, int rows, int C,
                                                           int act, int accumulate) {
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = blockIdx.y * 64 + tx * 4;
    __shared__ float psum[2][4][64];
    {
        const int cc = threadIdx.x & 63, part = threadIdx.x >> 6, ch = blockIdx.y * 64 + cc;
        float s1 = 0.f, s2 = 0.f;
        if (ch < C) bn_fold_partials(scratch, nchunk, C, ch, part, s1, s2);
        psum[0][part][cc] = s1; psum[1][part][cc] = s2;
        __syncthreads();
        if (threadIdx.x < 64) {
            s1 = psum[0][0][cc] + psum[0][1][cc] + psum[0][2][cc] + psum[0][3][cc];
            s2 = psum[1][0][cc] + psum[1][1][cc] + psum[1][2][cc] + psum[1][3][cc];
            psum[0][0][cc] = s1; psum[1][0][cc] = s2;
            if (blockIdx.x == 0 && ch < C) { dgamma[ch] += s2; dbeta[ch] += s1; }
        }
        __syncthreads();
    }
    if (c >= C) return;
    float mu[4], rs[4], m1[4], m2[4], gam[4], bet[4];
    const float4 g = ld4<T>(gamma + c), b = ld4<T>(beta + c);
    gam[0] = g.x; gam[1] = g.y; gam[2] = g.z; gam[3] = g.w;
    bet[0] = b.x; bet[1] = b.y; bet[2] = b.z; bet[3] = b.w;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mu[j] = save_mean[c + j]; rs[j] = save_rstd[c + j];
        m1[j] = psum[0][0][tx * 4 + j] / (float)rows; m2[j] = psum[1][0][tx * 4 + j] / (float)rows;
    }
    for (int r = blockIdx.x * 16 + ty; r < rows; r += gridDim.x * 16) {
        const float4 xv = ld4<T>(x + (long long)r * ldx + c);
        const float4 dv = ld4<T>(dy + (long long)r * lddy + c);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
        const float ds[4] = {dv.x, dv.y, dv.z, dv.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (xs[j] - mu[j]) * rs[j];
            float d = ds[j];
            if (act != TC_ACT_NONE) d *= act_grad(act, xh * gam[j] + bet[j]);
            o[j] = gam[j] * rs[j] * (d - m1[j] - xh * m2[j]);
        }
        if (accumulate) { const float4 q = ld4<T>(dx + (long long)r * lddx + c); o[0] += q.x; o[1] += q.y; o[2] += q.z; o[3] += q.w; }
        st4<T>(dx + (long long)r * lddx + c, make_float4(o[0], o[1], o[2], o[3]));
    }
}

// Deferred parameter gradients of the LayerNorms (tc_layernorm_bwd_defer): every backward launch leaves its per-workgroup column sums
// [groups][nblk][dgamma C | dbeta C] in a buffer of its own and returns -- no arrival counter, no last-arriver fold at its tail (3-5 us of a
// 6-11 us launch on the small maps) -- and ONE launch per backward leg adds them up for all sites.  Workgroup = (site, group, 64 columns of
// the 2C): 64 columns x 4 row slices, every load of a slice's batch issued before the first add.
constexpr int LN_FOLD_SITES = 64, LN_FOLD_ROWS = 128;
struct LnFoldSite { const float* part; float* dgamma; float* dbeta; long long pstride; int nblk, C, groups, blk0, rsplit; };
struct LnFoldDev { LnFoldSite s[LN_FOLD_SITES]; int n; };
static_assert(sizeof(LnFoldDev) <= 4096, "kernel argument block");
__global__ __launch_bounds__(256) void ln_fold_kernel(const LnFoldDev q) {
    __shared__ float red[4][64];
    int si = 0;
    for (int i = 1; i < q.n; ++i) if ((int)blockIdx.x >= q.s[i].blk0) si = i;
    const LnFoldSite& t = q.s[si];
    const int cchunks = (2 * t.C + 63) / 64;
    // (site, group, 64 columns, LN_FOLD_ROWS-row range): with up to 4096 partial rows per site one workgroup per column chunk was 30 us of serial adds
    int lin = blockIdx.x - t.blk0;
    const int rs = lin % t.rsplit; lin /= t.rsplit;
    const int g = lin / cchunks, c = (lin - g * cchunks) * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
    const int rend = min(t.nblk, (rs + 1) * LN_FOLD_ROWS);
    float v = 0.f;
    if (c < 2 * t.C) {
        const float* p = t.part + (long long)g * t.nblk * 2 * t.C + c;
        int r = rs * LN_FOLD_ROWS + sl;
        for (; r + 28 < rend; r += 32) {
            float tmp[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) tmp[m] = p[(long long)(r + 4 * m) * 2 * t.C];
#pragma unroll
            for (int m = 0; m < 8; ++m) v += tmp[m];
        }
        for (; r < rend; r += 4) v += p[(long long)r * 2 * t.C];
    }
    red[sl][threadIdx.x & 63] = v;
    __syncthreads();
    if (sl == 0 && c < 2 * t.C) {
        v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        atomicAdd(c < t.C ? t.dgamma + g * t.pstride + c : t.dbeta + g * t.pstride + (c - t.C), v);     // (a parameter used at two sites has two writers)
    }
}

// workgroups of the fused dx + parameter-gradient launch (one place: the launch and the size of a deferred partial buffer)
inline int ln_bwd_nblk(int rows, int C, bool with_params, bool deferred) {
    const int quads = C >> 2;
    const int GS = quads <= 16 ? 16 : (quads <= 32 ? 32 : 64);
    static const int ilp_on = getenv("TC_LN_BWD_ILP") ? atoi(getenv("TC_LN_BWD_ILP")) : 1;
    const bool ilp = ilp_on && quads <= 64 && rows >= 8192;
    int rpg = with_params ? rows / ((256 / GS) * ln_bwd_wg_min()) : 1;   // rows per lane group: >= 512 workgroups before rows are stacked
    rpg = rpg < 1 ? 1 : (rpg > LN_BWD_ROWS_PER_GROUP ? LN_BWD_ROWS_PER_GROUP : rpg);
    if (ilp && rpg < 2) rpg = 2;
    return tc_blocks(rows, (256 / GS) * rpg, with_params ? ln_bwd_blocks(deferred) : 8192);
}

}  // namespace

static int ln_fwd_impl(const void* x, int ldx, const void* gamma, const void* beta, void* y, int ldy,
                       float* mean, float* rstd, int rows, int C, float eps, int act, int groups, long long pstride, int dtype,
                       void* stream, LnMap map) {
    if (!x || !y || !gamma || !beta || !mean || !rstd || rows <= 0 || groups < 1 || C <= 0 || (C & 3) || C > LN_MAXC ||
        (ldx & 3) || (ldy & 3) || (act != TC_ACT_NONE && act != TC_ACT_GELU))
        return TC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    {   // 16-bit rows of 64 / 128 / 256 / 512 channels, every row piece 16-byte aligned, large maps: eight channels per lane (ln_fwd_v8_kernel)
        static const int v8_on = getenv("TC_LN_FWD_V8") ? atoi(getenv("TC_LN_FWD_V8")) : 1;
        const bool al = !((ldx | ldy) & 7) && !(pstride & 7) && !(((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15);
        if (v8_on && (dtype == TC_BF16 || dtype == TC_F16) && act == TC_ACT_NONE && al && rows >= 65536 && (C == 64 || C == 128 || C == 256 || C == 512)) {
            const dim3 grid(tc_blocks(rows, (2048 / C) * 4, 2048), groups);
#define TC_LNF8(T_, GS_) hipLaunchKernelGGL((ln_fwd_v8_kernel<T_, GS_, 4>), grid, dim3(256), 0, s, (const T_*)x, ldx, (const T_*)gamma, (const T_*)beta,   \
                                            (T_*)y, ldy, mean, rstd, rows, eps, pstride, map)
#define TC_LNF8_C(T_) { if (C == 64) TC_LNF8(T_, 8); else if (C == 128) TC_LNF8(T_, 16); else if (C == 256) TC_LNF8(T_, 32); else TC_LNF8(T_, 64); }
            if (dtype == TC_BF16) TC_LNF8_C(bf16_t) else TC_LNF8_C(f16_t)
#undef TC_LNF8_C
#undef TC_LNF8
            return tc_launch_status();
        }
    }
    const int quads = C >> 2;
#define TC_LNF(GS, NV) {                                                                                                                  \
        constexpr int RPT = NV == 1 ? 4 : (NV == 2 ? 2 : 1);       /* narrow rows: several rows in flight per lane group */                   \
        if (rows >= 4096)                                                                                                                     \
            hipLaunchKernelGGL((ln_fwd_kernel<T, GS, NV, RPT>), dim3(tc_blocks(rows, (256 / GS) * RPT, 2048), groups), dim3(256), 0, s,       \
                               (const T*)x, ldx, (const T*)gamma, (const T*)beta, (T*)y, ldy, mean, rstd, rows, C, eps, act, pstride, map);  \
        else                                                                                                                                  \
            hipLaunchKernelGGL((ln_fwd_kernel<T, GS, NV, 1>), dim3(tc_blocks(rows, 256 / GS, 2048), groups), dim3(256), 0, s, (const T*)x,    \
                               ldx, (const T*)gamma, (const T*)beta, (T*)y, ldy, mean, rstd, rows, C, eps, act, pstride, map); }
    TC_DISPATCH_DTYPE(dtype, { TC_LN_DISPATCH(quads, TC_LNF) });
#undef TC_LNF
    return tc_launch_status();
}

extern "C" int tc_layernorm_fwd(const void* x, int ldx, const void* gamma, const void* beta, void* y, int ldy,
                                float* mean, float* rstd, int rows, int C, float eps, int act, int groups, long long pstride, int dtype,
                                void* stream) {
    return ln_fwd_impl(x, ldx, gamma, beta, y, ldy, mean, rstd, rows, C, eps, act, groups, pstride, dtype, stream, LnMap{0, 0, 0});
}

extern "C" int tc_layernorm_ps_fwd(const void* x, int ldx, const void* gamma, const void* beta, void* y, int ldy, float* mean, float* rstd,
                                   int B, int H, int W, int p, int C, float eps, int dtype, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || p < 1 || ldx < p * p * C) return TC_ERR_ARG;
    return ln_fwd_impl(x, ldx, gamma, beta, y, ldy, mean, rstd, B * H * p * W * p, C, eps, TC_ACT_NONE, 1, 0, dtype, stream, LnMap{p, H, W});
}

extern "C" int tc_layernorm_bwd_params(const void* dy, int lddy, const void* x, int ldx, const void* gamma, const void* beta,
                                       const float* mean, const float* rstd, float* dgamma, float* dbeta, int rows, int C, int act,
                                       int groups, long long pstride, int dtype, void* stream);

static int ln_bwd_impl(const void* dy, int lddy, const void* x, int ldx, const void* gamma, const void* beta,
                       const float* mean, const float* rstd, void* dx, int lddx, const void* dres, int ldres,
                       float* dgamma, float* dbeta, int rows, int C, int act, int groups, long long pstride,
                       float* scratch, long long scratch_floats, int dtype, void* stream, LnMap map, float* defer_part = nullptr) {
    if (!dy || !x || !gamma || !beta || !mean || !rstd || !dx || (!dgamma != !dbeta) || rows <= 0 || groups < 1 || C <= 0 || (C & 3) ||
        C > LN_MAXC || (ldx & 3) || (lddy & 3) || (lddx & 3) || (dres && (ldres & 3)))
        return TC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int quads = C >> 2;
    if (dgamma && quads > 256 && map.p == 0) {
        // wide rows (C = 1280 / 2048: 5-8 float4 per lane): the one-pass kernel would hold ~300 VGPRs; a dx-only launch plus the
        // column-reduction parameter kernel is faster there (measured 18 vs 37 us at 784 x 2048)
        const int rc = tc_layernorm_bwd(dy, lddy, x, ldx, gamma, beta, mean, rstd, dx, lddx, dres, ldres, nullptr, nullptr, rows, C, act,
                                        groups, pstride, nullptr, 0, dtype, stream);
        return rc != TC_OK ? rc : tc_layernorm_bwd_params(dy, lddy, x, ldx, gamma, beta, mean, rstd, dgamma, dbeta, rows, C, act, groups,
                                                          pstride, dtype, stream);
    }
    int nblk = 0;
    float* partial = nullptr;
    {   // 16-bit rows of 64 / 128 / 256 / 512 channels, every row piece 16-byte aligned: eight channels per lane (ln_bwd_v8_kernel)
        static const int v8_on = getenv("TC_LN_BWD_V8") ? atoi(getenv("TC_LN_BWD_V8")) : 1;
        const bool al = !((lddy | ldx | lddx | (dres ? ldres : 0)) & 7) && !(pstride & 7) &&
                        !(((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)gamma | (dres ? (uintptr_t)dres : 0)) & 15);
        if (v8_on && (dtype == TC_BF16 || dtype == TC_F16) && act == TC_ACT_NONE && al && (C == 64 || C == 128 || C == 256 || C == 512)) {
            nblk = ln_bwd_nblk(rows, C, dgamma != nullptr, defer_part != nullptr);
            partial = defer_part ? defer_part :
                      (dgamma && scratch && (uintptr_t)scratch % 16 == 0 && scratch_floats >= 4096 + (long long)groups * nblk * 2 * C &&
                       (long long)groups * ((nblk + LN_FOLD - 1) / LN_FOLD) <= 4096) ? scratch + 4096 : nullptr;
            const long long rpi = (long long)nblk * (2048 / C);           // rows the grid takes per iteration with one row per lane group
            const int rpt = rows >= 4 * rpi ? 4 : (rows >= 2 * rpi ? 2 : 1);
#define TC_LNB8(T_, GS_, RPT_)                                                                                                             \
            hipLaunchKernelGGL((ln_bwd_v8_kernel<T_, GS_, RPT_>), dim3(nblk, groups), dim3(256), (size_t)(256 / GS_) * 2 * C * sizeof(float), s,   \
                               (const T_*)dy, lddy, (const T_*)x, ldx, (const T_*)gamma, (const T_*)beta, mean, rstd, (T_*)dx, lddx,            \
                               (const T_*)dres, ldres, dgamma, dbeta, rows, pstride, partial, reinterpret_cast<int*>(scratch), map, defer_part ? 1 : 0)
#define TC_LNB8_R(T_, GS_) { if (rpt == 4) TC_LNB8(T_, GS_, 4); else if (rpt == 2) TC_LNB8(T_, GS_, 2); else TC_LNB8(T_, GS_, 1); }
#define TC_LNB8_C(T_) { if (C == 64) TC_LNB8_R(T_, 8) else if (C == 128) TC_LNB8_R(T_, 16) else if (C == 256) TC_LNB8_R(T_, 32) else TC_LNB8_R(T_, 64) }
            if (dtype == TC_BF16) TC_LNB8_C(bf16_t) else TC_LNB8_C(f16_t)
#undef TC_LNB8_C
#undef TC_LNB8_R
#undef TC_LNB8
            return tc_launch_status();
        }
    }
#define TC_LNB_LAUNCH(GS, NV, RPT)                                                                                                        \
        hipLaunchKernelGGL((ln_bwd_kernel<T, GS, NV, RPT>), dim3(nblk, groups), dim3(256),                                                  \
                                          (size_t)(256 / GS) * 2 * C * sizeof(float), s, (const T*)dy, lddy, (const T*)x, ldx,            \
                                          (const T*)gamma, (const T*)beta, mean, rstd, (T*)dx, lddx, (const T*)dres, ldres, dgamma,      \
                                          dbeta, rows, C, act, pstride, partial, reinterpret_cast<int*>(scratch), map, defer_part ? 1 : 0)
#define TC_LNB(GS, NV) {                                                                                                                  \
        constexpr int RPT = 2;   /* rows in flight per lane group (narrow rows): 14.16 vs 14.19 ms per step; 4 rows: 14.21 vs 14.22 */                                                         \
        static const int ilp_on = getenv("TC_LN_BWD_ILP") ? atoi(getenv("TC_LN_BWD_ILP")) : 1;                                             \
        const bool ilp = ilp_on && NV == 1 && rows >= 8192;                                                                                \
        nblk = ln_bwd_nblk(rows, C, dgamma != nullptr, defer_part != nullptr);                                                             \
        partial = defer_part ? defer_part :                                                                                               \
                  (dgamma && scratch && (uintptr_t)scratch % 16 == 0 && scratch_floats >= 4096 + (long long)groups * nblk * 2 * C &&      \
                   (long long)groups * ((nblk + LN_FOLD - 1) / LN_FOLD) <= 4096) ? scratch + 4096 : nullptr;                                            \
        if (ilp) TC_LNB_LAUNCH(GS, NV, RPT); else TC_LNB_LAUNCH(GS, NV, 1); }
    TC_DISPATCH_DTYPE(dtype, { TC_LN_DISPATCH(quads, TC_LNB) });
#undef TC_LNB
#undef TC_LNB_LAUNCH
    return tc_launch_status();
}

#ifdef TC_LN_TIMING
extern "C" int tc_ln_dbg_read(unsigned long long* dst, int reset) {
    int rc = (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_ln_dbg), sizeof(unsigned long long) * 1024 * 8);
    if (reset) {
        static unsigned long long z[1024 * 8];
        unsigned int zn = 0;
        rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ln_dbg), z, sizeof(z));
        rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ln_dbg_n), &zn, sizeof(zn));
    }
    return rc;
}
#endif
extern "C" int tc_layernorm_bwd(const void* dy, int lddy, const void* x, int ldx, const void* gamma, const void* beta,
                                const float* mean, const float* rstd, void* dx, int lddx, const void* dres, int ldres,
                                float* dgamma, float* dbeta, int rows, int C, int act, int groups, long long pstride,
                                float* scratch, long long scratch_floats, int dtype, void* stream) {
    return ln_bwd_impl(dy, lddy, x, ldx, gamma, beta, mean, rstd, dx, lddx, dres, ldres, dgamma, dbeta, rows, C, act, groups, pstride, scratch,
                       scratch_floats, dtype, stream, LnMap{0, 0, 0});
}

extern "C" int tc_layernorm_ps_bwd(const void* dy, int lddy, const void* x, int ldx, const void* gamma, const void* beta, const float* mean,
                                   const float* rstd, void* dx, int lddx, float* dgamma, float* dbeta, int B, int H, int W, int p, int C,
                                   float* scratch, long long scratch_floats, int dtype, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || p < 1 || ldx < p * p * C || lddx < p * p * C || C > 1024) return TC_ERR_ARG;
    return ln_bwd_impl(dy, lddy, x, ldx, gamma, beta, mean, rstd, dx, lddx, nullptr, 0, dgamma, dbeta, B * H * p * W * p, C, TC_ACT_NONE, 1, 0,
                       scratch, scratch_floats, dtype, stream, LnMap{p, H, W});
}

extern "C" int tc_layernorm_bwd_nblk(int rows, int C) {
    if (rows <= 0 || C <= 0 || (C & 3) || C > LN_MAXC || (C >> 2) > 256) return 0;     // wide rows fold in launches of their own
    return ln_bwd_nblk(rows, C, true, true);
}

extern "C" int tc_layernorm_bwd_defer(const void* dy, int lddy, const void* x, int ldx, const void* gamma, const void* beta,
                                      const float* mean, const float* rstd, void* dx, int lddx, const void* dres, int ldres,
                                      float* dgamma, float* dbeta, int rows, int C, int act, int groups, long long pstride,
                                      float* part, long long part_floats, int dtype, void* stream) {
    const int nblk = tc_layernorm_bwd_nblk(rows, C);
    if (!part || !dgamma || nblk <= 0 || groups < 1 || part_floats < (long long)groups * nblk * 2 * C) return TC_ERR_ARG;
    return ln_bwd_impl(dy, lddy, x, ldx, gamma, beta, mean, rstd, dx, lddx, dres, ldres, dgamma, dbeta, rows, C, act, groups, pstride, nullptr, 0,
                       dtype, stream, LnMap{0, 0, 0}, part);
}

extern "C" int tc_layernorm_fold(const TcLnFold* sites, int n, void* stream) {
    if (!sites || n < 1 || n > LN_FOLD_SITES) return TC_ERR_ARG;
    LnFoldDev q;
    q.n = n;
    int blk = 0;
    for (int i = 0; i < n; ++i) {
        const TcLnFold& t = sites[i];
        if (!t.part || !t.dgamma || !t.dbeta || t.nblk < 1 || t.C <= 0 || t.groups < 1) return TC_ERR_ARG;
        const int rsplit = (t.nblk + LN_FOLD_ROWS - 1) / LN_FOLD_ROWS;
        q.s[i] = LnFoldSite{t.part, t.dgamma, t.dbeta, t.pstride, t.nblk, t.C, t.groups, blk, rsplit};
        blk += t.groups * ((2 * t.C + 63) / 64) * rsplit;
    }
    hipLaunchKernelGGL(ln_fold_kernel, dim3(blk), dim3(256), 0, (hipStream_t)stream, q);
    return tc_launch_status();
}

extern "C" long long tc_layernorm_bwd_scratch_floats(int rows, int C, int groups) {
    if (rows <= 0 || C <= 0 || groups < 1) return 0;
    return 4096 + (long long)groups * 1024 * 2 * C;      // counters + an upper bound over the lane-group shapes (launches that fold at their tail use <= 1024 workgroups)
}

extern "C" int tc_layernorm_bwd_params(const void* dy, int lddy, const void* x, int ldx, const void* gamma, const void* beta,
                                       const float* mean, const float* rstd, float* dgamma, float* dbeta, int rows, int C, int act,
                                       int groups, long long pstride, int dtype, void* stream) {
    if (!dy || !x || !gamma || !beta || !mean || !rstd || !dgamma || !dbeta || rows <= 0 || groups < 1 || C <= 0 || (C & 3) || (ldx & 3) ||
        (lddy & 3))
        return TC_ERR_ARG;
    dim3 grid(tc_blocks(rows, 16 * 8, 128), (C + 63) / 64, groups);
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ln_param_grad_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)dy, lddy,
                                                (const T*)x, ldx, (const T*)gamma, (const T*)beta, mean, rstd, dgamma, dbeta, rows, C, act,
                                                pstride));
    return tc_launch_status();
}

extern "C" long long tc_bn_scratch_floats(int rows, int C) { int n = (rows + 63) / 64; n = n < 1 ? 1 : (n > 128 ? 128 : n); return (long long)C * (1 + 2 * n); }

extern "C" int tc_bn_fwd(const void* x, int ldx, const void* gamma, const void* beta, float* running_mean,
                         float* running_var, const void* res, int ldres, void* y, int ldy, float* save_mean,
                         float* save_rstd, float* partial, int rows, int C, float eps, float momentum, int training,
                         int stats_chunks, int act, int dtype, void* stream) {
    if (!x || !gamma || !beta || !running_mean || !running_var || !y || rows <= 0 || C <= 0 || (C & 3) || (ldx & 3) ||
        (ldy & 3) || (res && (ldres & 3)) || (training && (!save_mean || !save_rstd || !partial)) || stats_chunks < 0 ||
        (stats_chunks > 0 && !training))
        return TC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int nchunk = stats_chunks > 0 ? stats_chunks : bn_nchunk(rows);
    const int cb = (C + 63) / 64;
    TC_DISPATCH_DTYPE(dtype, {
        if (training && stats_chunks == 0) {
            hipLaunchKernelGGL((bn_partial_kernel<T, 0>), dim3(nchunk, cb), dim3(256), 0, s, (const T*)x, ldx, (const T*)nullptr,
                               0, (const T*)gamma, (const T*)beta, (const float*)nullptr, (const float*)nullptr, partial, rows,
                               C, act);
        }
        // rows per workgroup of the apply pass: these maps are a few MB, the pass is bound by its round trips, not by bytes -- more, shorter
        // workgroups win although each folds the chunk partials again (RIPM stages alone: 536 us at 64 rows, 527 at 32, 523 at 16, 594 at 256)
        static const int rpw = getenv("TC_BN_ROWS_PER_WG") ? atoi(getenv("TC_BN_ROWS_PER_WG")) : 16;
        const int rb = tc_blocks(rows, rpw, 2048);
        hipLaunchKernelGGL((bn_apply_kernel<T>), dim3(rb, cb), dim3(256), 0, s, (const T*)x, ldx, (const T*)gamma,
                           (const T*)beta, running_mean, running_var, (const T*)res, ldres, (T*)y, ldy, save_mean, save_rstd,
                           partial, nchunk, rows, C, eps, momentum, training, act);
    });
    return tc_launch_status();
}

extern "C" int tc_bn_bwd(const void* dy, int lddy, const void* x, int ldx, const void* gamma, const void* beta,
                         const float* save_mean, const float* save_rstd, void* dx, int lddx, float* dgamma, float* dbeta,
                         float* partial, int rows, int C, int act, int accumulate, int dtype, void* stream) {
    if (!dy || !x || !gamma || !beta || !save_mean || !save_rstd || !dx || !dgamma || !dbeta || !partial || rows <= 0 ||
        C <= 0 || (C & 3) || (ldx & 3) || (lddy & 3) || (lddx & 3))
        return TC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int nchunk = bn_nchunk(rows);
    const int cb = (C + 63) / 64;
    TC_DISPATCH_DTYPE(dtype, {
        hipLaunchKernelGGL((bn_partial_kernel<T, 1>), dim3(nchunk, cb), dim3(256), 0, s, (const T*)x, ldx, (const T*)dy, lddy,
                           (const T*)gamma, (const T*)beta, save_mean, save_rstd, partial, rows, C, act);
        static const int rpwb = getenv("TC_BN_BWD_ROWS_PER_WG") ? atoi(getenv("TC_BN_BWD_ROWS_PER_WG")) : 16;
        const int rb = tc_blocks(rows, rpwb, 2048);
        hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(rb, cb), dim3(256), 0, s, (const T*)dy, lddy, (const T*)x, ldx,
                           (const T*)gamma, (const T*)beta, save_mean, save_rstd, (T*)dx, lddx, dgamma, dbeta, partial, nchunk,
                           rows, C, act, accumulate);
    });
    return tc_launch_status();
}
