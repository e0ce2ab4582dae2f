// The tail of the last decoder stage (MyDecoderLayer with is_last, MSTr.py:271-290 + FinalPatchExpand_X4 :213-227):
//     rearrange 'b h w (p1 p2 c) -> b (h p1) (w p2) c'  ->  LayerNorm(c)  ->  last_layer = Conv2d(c, n_class, 1)
// as ONE forward launch and ONE backward launch (+ a small fold of the parameter-gradient partials) on 16-bit storage, c = 64.
// The op-by-op form (tc_layernorm_ps_fwd + tc_gemm; tc_gemm_pair + tc_layernorm_ps_bwd) writes the normalised 224^2 x 64 map and reads it
// back -- 103 MB each way at B = 16 -- for a product with K = 64, N = 9; in the backward it writes and re-reads the gradient of that map as
// well (r5 profile: 40 + 21 us forward, 108 + 71 us backward, 0.6 GB of traffic for 14 MB of logits).  Here a row of the map lives in
// eight lanes (eight channels each), the normalised values never leave registers, and the traffic is x in, logits out (forward) and
// dlogits + x in, dx out (backward): the HBM floor of the two maps.
//   forward   mean / rstd by DPP sums over the eight lanes, xn = (x - mean) rstd gamma + beta in fp32 (NOT rounded to the storage type: one
//             rounding fewer in front of the logits than the op-by-op form), logits[c] = sum_ch xn[ch] Wc[c][ch] + bc[c]: 8 x NC FMAs per
//             lane against the lane's slice of Wc held in registers, NC butterfly sums, lane j of the row stores class j
//   backward  xn recomputed from x and the saved statistics; dWc += dl (x) xn, dbc += dl, dxn = dl Wc, dgamma += dxn xhat, dbeta += dxn,
//             LayerNorm backward of dxhat = dxn gamma -> dx (in the un-shuffled layout of x).  The parameter sums stay in registers for the
//             whole launch, leave once per workgroup as fp32 partials, and tc_ln_cls_bwd's second launch folds them into the gradient arrays.
// The pixel shuffle is the address computation of norm.hip's ln_row_off (output pixel -> row / column slice of the expand Linear's output).
#include "tc_common.h"

namespace {

// Output pixel -> element offset of its c channels in the un-shuffled map.  The three divisions (by W p, H p and p) are multiplications by
// M = floor(2^32 / d) + 1 (exact for n * d < 2^32; the host checks): as run-time integer divisions they were ~100 of the ~250 instructions
// a row costs these kernels.
struct PsMap { int p, H, W; unsigned mWp, mHp, mp; };
__device__ __forceinline__ long long ps_row_off(int row, const PsMap& m, int ld, int C) {
    if (m.p == 0) return (long long)row * ld;
    const unsigned Wp = m.W * m.p, Hp = m.H * m.p, r = (unsigned)row;
    const unsigned t = __umulhi(r, m.mWp), ow = r - t * Wp, b = __umulhi(t, m.mHp), oh = t - b * Hp;
    const unsigned ih = __umulhi(oh, m.mp), iw = __umulhi(ow, m.mp), p1 = oh - ih * m.p, p2 = ow - iw * m.p;
    return ((long long)(b * m.H + ih) * m.W + iw) * ld + (p1 * m.p + p2) * C;
}
inline bool ps_map(PsMap& m, int p, int H, int W, long long rows) {
    m.p = p; m.H = H; m.W = W; m.mWp = m.mHp = m.mp = 0;
    if (p == 0) return true;
    if (p < 2) return false;                                        // (M of d = 1 does not fit 32 bits; p = 1 is p = 0 to the callers)
    const unsigned long long Wp = (unsigned long long)W * p, Hp = (unsigned long long)H * p;
    if ((unsigned long long)rows * (Wp > Hp ? Wp : Hp) >= (1ull << 32)) return false;
    m.mWp = (unsigned)((1ull << 32) / Wp + 1); m.mHp = (unsigned)((1ull << 32) / Hp + 1); m.mp = (unsigned)((1ull << 32) / (unsigned)p + 1);
    return true;
}

template <typename T> __device__ __forceinline__ void up8(const uint4& r, float* o) {
    unpack2<T>(r.x, o[0], o[1]); unpack2<T>(r.y, o[2], o[3]); unpack2<T>(r.z, o[4], o[5]); unpack2<T>(r.w, o[6], o[7]);
}

constexpr int LC_C = 64, LC_GS = 8, LC_RPB = 256 / LC_GS;      // channels; lanes per row; rows per workgroup pass

// One pass = 32 consecutive output rows per workgroup; RP passes' loads are issued before the first reduction.
// PL: the logits rows are 16-byte aligned and padded to a multiple of 8 elements (Graph.ln_cls(pad_rows=True)): lane q of the row stores
// the 16-byte piece q (classes 8 q .. 8 q + 7, zeros past NC); otherwise lane j stores class j as a 2-byte element (nine partial-line
// writes per row: measured 56 us for the 224^2 B = 16 map against 3x less with whole pieces).
template <typename T, int NC, int RP, bool PL>
__global__ __launch_bounds__(256) void ln_cls_fwd_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ gamma, const T* __restrict__ beta,
                                                         const T* __restrict__ Wc, const T* __restrict__ bc, T* __restrict__ logits, int ldl,
                                                         float* __restrict__ mean, float* __restrict__ rstd, int rows, float eps, PsMap map) {
    const int gl = threadIdx.x % LC_GS, gi = threadIdx.x / LC_GS;
    // the lane's slice of Wc is read from LDS per row (fp32; the same address across a wave's rows: broadcast) -- in registers it cost 72 VGPRs
    // and a wave per SIMD
    __shared__ float4 wsh4[NC * LC_C / 4];
    float* wsh = reinterpret_cast<float*>(wsh4);
    for (int i = threadIdx.x; i < NC * LC_C; i += 256) wsh[i] = ldf<T>(Wc + i);
    float g[8], b[8], bias[NC];
    up8<T>(*reinterpret_cast<const uint4*>(gamma + gl * 8), g);
    up8<T>(*reinterpret_cast<const uint4*>(beta + gl * 8), b);
#pragma unroll
    for (int c = 0; c < NC; ++c) bias[c] = ldf<T>(bc + c);
    __syncthreads();
    for (long long r0 = (long long)blockIdx.x * LC_RPB * RP; r0 < rows; r0 += (long long)gridDim.x * LC_RPB * RP) {
        uint4 raw[RP];
#pragma unroll
        for (int q = 0; q < RP; ++q) {
            const long long r = r0 + q * LC_RPB + gi;
            raw[q] = *reinterpret_cast<const uint4*>(x + ps_row_off((int)(r < rows ? r : rows - 1), map, ldx, LC_C) + gl * 8);   // (branch-free: rows past the end re-read the last one)
        }
#pragma unroll
        for (int q = 0; q < RP; ++q) {
            const long long r = r0 + q * LC_RPB + gi;
            float v[8];
            up8<T>(raw[q], v);
            float s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            s = tc_group_sum<LC_GS>(s);
            const float mu = s * (1.0f / LC_C);
            float q2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[e] -= mu; q2 = fmaf(v[e], v[e], q2); }
            q2 = tc_group_sum<LC_GS>(q2);
            const float rs = rsqrtf(q2 * (1.0f / LC_C) + eps);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e] * rs, g[e], b[e]);
            float acc[NC];
            int woff = gl * 2;
            asm volatile("" : "+v"(woff));                          // (keeps the loop-invariant LDS reads inside the loop)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float4 w0 = wsh4[c * (LC_C / 4) + woff], w1 = wsh4[c * (LC_C / 4) + woff + 1];
                float a = v[0] * w0.x;
                a = fmaf(v[1], w0.y, a); a = fmaf(v[2], w0.z, a); a = fmaf(v[3], w0.w, a);
                a = fmaf(v[4], w1.x, a); a = fmaf(v[5], w1.y, a); a = fmaf(v[6], w1.z, a); a = fmaf(v[7], w1.w, a);
                acc[c] = tc_group_sum<LC_GS>(a) + bias[c];
            }
            if (r < rows) {
                if (gl == 0) { mean[r] = mu; rstd[r] = rs; }
                T* lp = logits + r * ldl;
                // every lane of the group holds every sum after the butterflies
                if constexpr (PL) {
                    constexpr int NQ = (NC + 7) / 8;
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        if (gl == q) {
                            float o[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = 8 * q + e < NC ? acc[(8 * q + e) < NC ? 8 * q + e : 0] : 0.f;
                            *reinterpret_cast<uint4*>(lp + 8 * q) = make_uint4(pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3]), pack2<T>(o[4], o[5]), pack2<T>(o[6], o[7]));
                        }
                } else {
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                        if ((c & 7) == gl) stf<T>(lp + c, acc[c]);
                }
            }
        }
    }
}

// partial layout per workgroup (fp32): dgamma[64] | dbeta[64] | dWc[NC][64] | dbc[NC]
template <int NC> struct LcPart { static constexpr int oG = 0, oB = 64, oW = 128, oC = 128 + NC * 64, n = 128 + NC * 64 + NC; };

// Registers: the NC x 8 weight-gradient sums + dgamma / dbeta / dbc stay in VGPRs for the whole launch (97 at NC = 9); the lane's slice of Wc
// is read from LDS per row (fp32, two 16-byte reads per class, the same address across a wave's rows: broadcast) -- held in registers as
// well the kernel took 256 VGPRs and ran one wave per SIMD.  The next row's loads are issued before the current row's arithmetic.
// AL: the dlogits rows are 16-byte aligned (padded row stride: the captured training step's layout) and read in 16-byte pieces; otherwise
// (contiguous [rows, ncls], 2 * ncls bytes per row) element by element.
template <typename T, int NC, bool AL>
__global__ __launch_bounds__(256, 2) void ln_cls_bwd_kernel(const T* __restrict__ dl, int lddl, const T* __restrict__ x, int ldx,
                                                            const T* __restrict__ gamma, const T* __restrict__ beta, const T* __restrict__ Wc,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ dx,
                                                            int lddx, float* __restrict__ part, int rows, PsMap map) {
    using PT = LcPart<NC>;
    __shared__ float red[4][PT::n];
    __shared__ float4 wsh4[NC * LC_C / 4];
    float* wsh = reinterpret_cast<float*>(wsh4);
    const int gl = threadIdx.x % LC_GS, gi = threadIdx.x / LC_GS, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < NC * LC_C; i += 256) wsh[i] = ldf<T>(Wc + i);
    float g[8], b[8];
    up8<T>(*reinterpret_cast<const uint4*>(gamma + gl * 8), g);
    up8<T>(*reinterpret_cast<const uint4*>(beta + gl * 8), b);
    float dg[8], db[8], dw[NC][8], dc[NC];
#pragma unroll
    for (int e = 0; e < 8; ++e) { dg[e] = 0.f; db[e] = 0.f; }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        dc[c] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) dw[c][e] = 0.f;
    }
    __syncthreads();
    constexpr int NQ = (NC + 7) / 8;                                // 16-byte pieces of a dlogits row (rows are 16-byte aligned: lddl % 8 == 0)
    const long long step = (long long)gridDim.x * LC_RPB;
    // two rows of this lane group are in flight ahead of the one being worked on (at two waves per SIMD one row ahead left ~24 KB per CU
    // outstanding: 106 us for the 224^2 B = 16 map, latency-bound)
    struct Row { uint4 raw, dq[NQ]; float du[AL ? 1 : NC]; float mu, rs; };
    Row nx0, nx1;
    auto fetch = [&](Row& w, long long r) __attribute__((always_inline)) {
        const long long rc = r < rows ? r : rows - 1;               // (branch-free: rows past the end re-read the last one and are dropped)
        w.raw = *reinterpret_cast<const uint4*>(x + ps_row_off((int)rc, map, ldx, LC_C) + gl * 8);
        if constexpr (AL) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) w.dq[q] = *reinterpret_cast<const uint4*>(dl + rc * lddl + q * 8);
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) w.du[c] = ldf<T>(dl + rc * lddl + c);
        }
        w.mu = mean[rc]; w.rs = rstd[rc];
    };
    long long r = (long long)blockIdx.x * LC_RPB + gi;
    fetch(nx0, r);
    fetch(nx1, r + step);
    for (; r - gi < rows; r += step) {
        const bool ok = r < rows;
        float v[8], d[NC];
        const float mu_ = nx0.mu, rs_ = nx0.rs;
        up8<T>(nx0.raw, v);
        if constexpr (AL) {
            float t8[8 * NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) up8<T>(nx0.dq[q], t8 + 8 * q);
#pragma unroll
            for (int c = 0; c < NC; ++c) d[c] = ok ? t8[c] : 0.f;
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) d[c] = ok ? nx0.du[c] : 0.f;
        }
        nx0 = nx1;
        fetch(nx1, r + 2 * step);                                   // two rows ahead: in flight under the arithmetic of this row and the next
        float xh[8], xn[8], dxn[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { xh[e] = (v[e] - mu_) * rs_; xn[e] = fmaf(xh[e], g[e], b[e]); dxn[e] = 0.f; }
        int woff = gl * 2;
        asm volatile("" : "+v"(woff));                              // (loop-invariant LDS reads would be hoisted: 72 more live registers)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float4 w0 = wsh4[c * (LC_C / 4) + woff], w1 = wsh4[c * (LC_C / 4) + woff + 1];
            const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
            dc[c] += d[c];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                dw[c][e] = fmaf(d[c], xn[e], dw[c][e]);
                dxn[e] = fmaf(d[c], wv[e], dxn[e]);
            }
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            dg[e] = fmaf(dxn[e], xh[e], dg[e]);
            db[e] += dxn[e];
            dxn[e] *= g[e];                                         // dxhat
            s1 += dxn[e]; s2 = fmaf(dxn[e], xh[e], s2);
        }
        s1 = tc_group_sum<LC_GS>(s1) * (1.0f / LC_C);
        s2 = tc_group_sum<LC_GS>(s2) * (1.0f / LC_C);
        if (ok) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rs_ * (dxn[e] - s1 - xh[e] * s2);
            const long long dxo = ps_row_off((int)r, map, lddx, LC_C) + gl * 8;
            *reinterpret_cast<uint4*>(dx + dxo) = make_uint4(pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3]), pack2<T>(o[4], o[5]), pack2<T>(o[6], o[7]));
        }
    }
    // fold the lanes that own the same eight channels (lane bits 3..5), then the four waves through LDS; one partial per workgroup
    auto fold = [&](float vv) __attribute__((always_inline)) {
        vv += __shfl_xor(vv, 8, 64); vv += __shfl_xor(vv, 16, 64); vv += __shfl_xor(vv, 32, 64);
        return vv;
    };
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float a = fold(dg[e]), c2 = fold(db[e]);
        if (lane < 8) { red[wave][PT::oG + gl * 8 + e] = a; red[wave][PT::oB + gl * 8 + e] = c2; }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a = fold(dw[c][e]);
            if (lane < 8) red[wave][PT::oW + c * 64 + gl * 8 + e] = a;
        }
        const float a = fold(dc[c]);                                // every lane of a row added the row's dl[c]: lane 0's fold is the sum over the wave's rows
        if (lane == 0) red[wave][PT::oC + c] = a;
    }
    __syncthreads();
    float* po = part + (long long)blockIdx.x * PT::n;
    for (int i = threadIdx.x; i < PT::n; i += 256) po[i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// sums the per-workgroup partials and ADDS them to the gradient arrays: thread = (entry, one of gridDim.y slices of the partials), eight
// partials in flight, one fp32 atomic per thread (one thread per entry walking all 512 partials took 26 us: a chain of 64 round trips)
template <int NC>
__global__ __launch_bounds__(256) void ln_cls_fold_kernel(const float* __restrict__ part, int nblk_all, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          float* __restrict__ dWc, float* __restrict__ dbc) {
    using PT = LcPart<NC>;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= PT::n) return;
    const int per = (nblk_all + (int)gridDim.y - 1) / (int)gridDim.y, k0 = blockIdx.y * per, nblk = min(nblk_all, k0 + per);
    float s = 0.f;
    int k = k0;
    for (; k + 8 <= nblk; k += 8) {
        float a[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) a[m] = part[(long long)(k + m) * PT::n + i];
        s += ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    for (; k < nblk; ++k) s += part[(long long)k * PT::n + i];
    if (k0 >= nblk_all) return;
    float* dst = i < PT::oB ? dgamma + i : i < PT::oW ? dbeta + (i - PT::oB) : i < PT::oC ? dWc + (i - PT::oW) : dbc + (i - PT::oC);
    atomicAdd(dst, s);
}

inline int lc_nblk(int rows) {
    const long long passes = ((long long)rows + LC_RPB - 1) / LC_RPB;
    return (int)(passes < 512 ? passes : 512);                     // 2 workgroups per CU (256 threads at <= 256 VGPRs)
}
inline bool lc_aligned(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int tc_ln_cls_supported(int C, int ncls, int dtype) {
    return (C == LC_C && (ncls == 2 || ncls == 9) && (dtype == TC_BF16 || dtype == TC_F16)) ? 1 : 0;
}

extern "C" long long tc_ln_cls_scratch_floats(int rows, int ncls) {
    return (long long)lc_nblk(rows) * (128 + (long long)ncls * 64 + ncls);
}

extern "C" int tc_ln_cls_fwd(const void* x, int ldx, const void* gamma, const void* beta, const void* Wc, const void* bc, void* logits, int ldl,
                             float* mean, float* rstd, int B, int H, int W, int p, int C, int ncls, float eps, int dtype, void* stream) {
    if (!tc_ln_cls_supported(C, ncls, dtype) || !x || !gamma || !beta || !Wc || !bc || !logits || !mean || !rstd || B <= 0 || H <= 0 || W <= 0 || p < 0 ||
        ldx < (p ? p * p : 1) * C || (ldx & 7) || ldl < ncls || !lc_aligned(x) || !lc_aligned(gamma) || !lc_aligned(beta))
        return TC_ERR_ARG;
    const long long rows = p ? (long long)B * H * p * W * p : (long long)B * H * W;
    if (rows > 0x7fffffffLL) return TC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    PsMap map;
    if (!ps_map(map, p, H, W, rows)) return TC_ERR_ARG;
    constexpr int RP = 4;                                           // rows of a lane group in flight
    const long long passes = (rows + RP * LC_RPB - 1) / (RP * LC_RPB);
    const int nblk = (int)(passes < 1024 ? passes : 1024);          // 4 workgroups per CU (~100 VGPRs)
    const bool pl = !(ldl & 7) && ldl >= ((ncls + 7) & ~7) && lc_aligned(logits);
#define TC_LCF1(T_, NC_, PL_) hipLaunchKernelGGL((ln_cls_fwd_kernel<T_, NC_, RP, PL_>), dim3(nblk), dim3(256), 0, s, (const T_*)x, ldx, (const T_*)gamma, (const T_*)beta, \
                                                 (const T_*)Wc, (const T_*)bc, (T_*)logits, ldl, mean, rstd, (int)rows, eps, map)
#define TC_LCF(T_, NC_) { if (pl) TC_LCF1(T_, NC_, true); else TC_LCF1(T_, NC_, false); }
    if (dtype == TC_BF16) { if (ncls == 9) TC_LCF(bf16_t, 9) else TC_LCF(bf16_t, 2) }
    else { if (ncls == 9) TC_LCF(f16_t, 9) else TC_LCF(f16_t, 2) }
#undef TC_LCF
#undef TC_LCF1
    return tc_launch_status();
}

extern "C" int tc_ln_cls_bwd(const void* dl, int lddl, const void* x, int ldx, const void* gamma, const void* beta, const void* Wc, const float* mean,
                             const float* rstd, void* dx, int lddx, float* dgamma, float* dbeta, float* dWc, float* dbc, float* scratch,
                             long long scratch_floats, int B, int H, int W, int p, int C, int ncls, int dtype, void* stream) {
    if (!tc_ln_cls_supported(C, ncls, dtype) || !dl || !x || !gamma || !beta || !Wc || !mean || !rstd || !dx || !dgamma || !dbeta || !dWc || !dbc || !scratch ||
        B <= 0 || H <= 0 || W <= 0 || p < 0 || ldx < (p ? p * p : 1) * C || lddx < (p ? p * p : 1) * C || (ldx & 7) || (lddx & 7) || lddl < ncls ||
        !lc_aligned(x) || !lc_aligned(dx) || !lc_aligned(gamma) || !lc_aligned(beta))
        return TC_ERR_ARG;
    const bool al = !(lddl & 7) && lddl >= ((ncls + 7) & ~7) && lc_aligned(dl);
    const long long rows = p ? (long long)B * H * p * W * p : (long long)B * H * W;
    if (rows > 0x7fffffffLL || scratch_floats < tc_ln_cls_scratch_floats((int)rows, ncls)) return TC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    PsMap map;
    if (!ps_map(map, p, H, W, rows)) return TC_ERR_ARG;
    const int nblk = lc_nblk((int)rows);
#define TC_LCB1(T_, NC_, AL_) hipLaunchKernelGGL((ln_cls_bwd_kernel<T_, NC_, AL_>), dim3(nblk), dim3(256), 0, s, (const T_*)dl, lddl, (const T_*)x, ldx, (const T_*)gamma, \
                                                 (const T_*)beta, (const T_*)Wc, mean, rstd, (T_*)dx, lddx, scratch, (int)rows, map)
#define TC_LCB(T_, NC_) { if (al) TC_LCB1(T_, NC_, true); else TC_LCB1(T_, NC_, false);                                                                          \
                          hipLaunchKernelGGL((ln_cls_fold_kernel<NC_>), dim3((LcPart<NC_>::n + 255) / 256, 32), dim3(256), 0, s, scratch, nblk, dgamma, dbeta, dWc, dbc); }
    if (dtype == TC_BF16) { if (ncls == 9) TC_LCB(bf16_t, 9) else TC_LCB(bf16_t, 2) }
    else { if (ncls == 9) TC_LCB(f16_t, 9) else TC_LCB(f16_t, 2) }
#undef TC_LCB
#undef TC_LCB1
    return tc_launch_status();
}
