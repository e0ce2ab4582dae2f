// Segmented single-launch bf16 attention for the Dual Transformer Bridge.
//
// The bridge's 6076 query tokens per image live in a stage-major buffer (four row segments, one per encoder scale, each
// holding B images back to back), while the 784 reduced K/V tokens are image-major.  One launch covers every
// (segment, image, 128-query tile): ~800 workgroups instead of four launches of 50-400, so the chip is filled and the
// launch boundary is paid once.  Versus attention.hip's first bf16 kernels this version also
//   * prefetches the next 128-key K/V fill into registers while the current one is being consumed,
//   * folds the softmax scale into the exp2 argument (one FMA per score), masks keys only in the tail tile,
//   * rescales the running output lazily (only when some row maximum grew by more than 2^8),
//   * allows two workgroups per CU (launch bounds) so one wave's softmax VALU overlaps another's MFMA.
// fp32 storage falls back to the per-segment fp32 kernels of attention.hip (parity path).
#include "../../transception_amd/csrc/tc_common.h"
#include <type_traits>
#ifndef ATT_VAR
#define ATT_VAR 0
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int D = 64, KB = 128, LDR = D + 8, LDTB = KB + 16;   // LDTB: 288-byte rows = 72 words (8 mod 32), see st_t8
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f
#define NEG_BIG (-1.0e30f)
#define RESCALE_THR 8.0f

struct Segs {
    int n;
    int nq[4];        // queries per image in the segment
    int row0[4];      // first row of the segment in the stage-major buffers
    int tile0[5];     // prefix sums of B * ceil(nq/128)   (forward / dQ tiling)
    int t32[5];       // prefix sums of ceil(nq/32)        (dK/dV per-image query tiles)
};

__device__ __forceinline__ int pi_row(int i) { return 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3); }
__device__ __forceinline__ int d_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ bf16x8 ld_frag(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
// two hardware transpose reads (ds_read_b64_tr_b16): lane i of a 16-lane group passes the address of row i>>2, columns
// 4*(i&3).. of a 4 x 16 block and receives column i of it; lo = rows 0-3, hi = rows 4-7 of the lane's 8-deep k-slice
__device__ __forceinline__ bf16x8 ld_frag_tr(const bf16_t* lo, const bf16_t* hi) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lo));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(hi));
    const s16x8 c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, c);
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int o) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (__bf16)v[o + i];
    return r;
}
__device__ __forceinline__ uint4 ld_row8(const bf16_t* base, int ld, int row, int nrows, int c8) {
    return row < nrows ? *reinterpret_cast<const uint4*>(base + (long long)row * ld + c8) : make_uint4(0u, 0u, 0u, 0u);
}
// Transposed LDS store of one 8-element strip: dst[(c8+i)*ldt + col] = v[i].  A wave stores 16 consecutive `col`s (lanes
// 0-15) for four strips c8 = 8g (g = lane>>4).  The element order is rotated by g so that the four lane groups hit rows
// that differ mod 4; with a row stride of 8 (mod 32) words their 8-word spans then fall on disjoint banks.
__device__ __forceinline__ void st_t8(bf16_t* dst, int ldt, int c8, int col, uint4 v, int g) {
    // w = v rotated right by g elements (register-only: a dynamically indexed union goes through scratch memory)
    const bool ds = (g & 2) != 0;
    const unsigned sh = (g & 1) * 16;
    const unsigned a0 = ds ? v.y : v.x, a1 = ds ? v.z : v.y, a2 = ds ? v.w : v.z, a3 = ds ? v.x : v.w;
    unsigned w[4];
    w[0] = __builtin_amdgcn_alignbit(a1, a0, sh); w[1] = __builtin_amdgcn_alignbit(a2, a1, sh);
    w[2] = __builtin_amdgcn_alignbit(a3, a2, sh); w[3] = __builtin_amdgcn_alignbit(a0, a3, sh);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ii = (i + g) & 7;
        dst[(c8 + ii) * ldt + col] = (bf16_t)((i & 1) ? (w[i >> 1] >> 16) : (w[i >> 1] & 0xffffu));
    }
}
// strip owned by a thread in a 128-key x 64-d fill: wave w, iteration it -> 16 keys x 32 d
__device__ __forceinline__ void fill_map(int tid, int it, int& r, int& c8, int& g) {
    const int lane = tid & 63, c = (tid >> 6) * 4 + it;
    g = lane >> 4;
    r = 16 * (c >> 1) + (lane & 15);
    c8 = 32 * (c & 1) + 8 * g;
}
__device__ __forceinline__ float max3f(float a, float b, float c) {      // no canonicalising v_max in front (MFMA outputs)
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// block -> (segment, image, first query row of the tile, queries valid in the tile's segment-image)
__device__ __forceinline__ void locate_tile(const Segs& sg, int t, int& row_base, int& q_local0, int& nq, int& b) {
    int s = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < sg.n && t >= sg.tile0[i]) s = i;
    const int local = t - sg.tile0[s];
    nq = sg.nq[s];
    const int tiles = (nq + 127) >> 7;
    b = local / tiles;
    q_local0 = (local - b * tiles) * 128;
    row_base = sg.row0[s] + b * nq;
}

#if ATT_VAR == 12
// v12 = v11 with V kept row-major (4x4 k-row permutation) and gathered by ds_read_b64_tr_b16; V strips loaded together
// v11: 64 queries per wave (two 32-query sets share every K / V fragment read from LDS), 2 waves = 128 queries per workgroup,
// 128 threads, up to 4 workgroups per CU (2 waves per SIMD, 256 VGPRs each).
__global__ __launch_bounds__(128, 2) void attn_fwd_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                              const bf16_t* __restrict__ V, int ldv, long long skv, bf16_t* __restrict__ O,
                                                              int ldo, float* __restrict__ lse, Segs sg, int Nk, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[KB * LDR];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[KB * LDR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    int row_base, q0, nq, b;
    locate_tile(sg, blockIdx.x, row_base, q0, nq, b);
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    bool ok[2];
    long long qrow[2];
    bf16x8 qf[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ql = q0 + wave * 64 + 32 * t + j;
        ok[t] = ql < nq;
        qrow[t] = (long long)row_base + ql;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint4 v = ok[t] ? *reinterpret_cast<const uint4*>(Q + qrow[t] * ldq + 16 * ks + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
            qf[t][ks] = *reinterpret_cast<const bf16x8*>(&v);
        }
    }
    const float qs = scale * LOG2E;
    f32x16 acc0[2], acc1[2];
    float m[2], lsum[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        m[t] = NEG_BIG; lsum[t] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[t][r] = 0.f; acc1[t][r] = 0.f; }
    }
    const int krow = pi_row(j);
    auto fmap = [&](int it, int& r, int& c8, int& g) {       // 2 waves x 8 iterations cover the 16 (16 keys x 32 d) chunks of a fill
        const int c = wave * 8 + it;
        g = lane >> 4; r = 16 * (c >> 1) + (lane & 15); c8 = 32 * (c & 1) + 8 * g;
    };
    uint4 kr[8];
    auto fetch = [&](int kb0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int r, c8, g; fmap(i, r, c8, g);
            kr[i] = ld_row8(Kb, ldk, kb0 + r, Nk, c8);
        }
    };
    fetch(0);
    for (int kb0 = 0; kb0 < Nk; kb0 += KB) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int r, c8, g; fmap(i, r, c8, g);
            *reinterpret_cast<uint4*>(&Ks[r * LDR + c8]) = kr[i];
        }
        {
            uint4 vr[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int r, c8, g; fmap(i, r, c8, g);
                vr[i] = *reinterpret_cast<const uint4*>(Vb + (long long)min(kb0 + r, Nk - 1) * ldv + c8);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int r, c8, g; fmap(i, r, c8, g);
                const int pr = (r & ~15) | ((r & 3) << 2) | ((r >> 2) & 3);
                *reinterpret_cast<uint4*>(&Vs[pr * LDR + c8]) = vr[i];
            }
        }
        __syncthreads();
        if (kb0 + KB < Nk) fetch(kb0 + KB);
#pragma unroll 1
        for (int sub = 0; sub < KB / 32 && kb0 + 32 * sub < Nk; ++sub) {
            f32x16 s[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
            const bf16_t* kp = Ks + (32 * sub + krow) * LDR + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = ld_frag(kp + 16 * ks);
                s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[0][ks], s[0], 0, 0, 0);
                s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[1][ks], s[1], 0, 0, 0);
            }
            const int kv0 = kb0 + 32 * sub;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (kv0 + 32 > Nk) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) if (kv0 + 16 * h + r >= Nk) s[t][r] = NEG_BIG;
                }
                float mx = s[t][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * qs;
                if (__any(mx > m[t] + RESCALE_THR)) {
                    const float mn = fmaxf(m[t], mx);
                    const float alpha = fast_exp2(m[t] - mn);
                    lsum[t] *= alpha;
                    m[t] = mn;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { acc0[t][r] *= alpha; acc1[t][r] *= alpha; }
                }
                float rs = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[t][r] = fast_exp2(fmaf(s[t][r], qs, -m[t])); rs += s[t][r]; }
                lsum[t] += rs;                                   // per-half partial; halves meet after the loop
            }
            const int gi = lane & 15, gq = (lane >> 4) & 1;
            const bf16_t* vp = Vs + (32 * sub + 16 * h + 4 * (gi >> 2)) * LDR + 16 * gq + 4 * (gi & 3);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const bf16x8 v0 = ld_frag_tr(vp + (2 * k2) * LDR, vp + (2 * k2 + 1) * LDR);
                const bf16x8 v1 = ld_frag_tr(vp + (2 * k2) * LDR + 32, vp + (2 * k2 + 1) * LDR + 32);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const bf16x8 pb = pack8(s[t], 8 * k2);
                    acc0[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pb, acc0[t], 0, 0, 0);
                    acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pb, acc1[t], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float tot = lsum[t] + __shfl_xor(lsum[t], 32, 64);
        if (ok[t]) {
            const float inv = 1.0f / tot;
            bf16_t* orow = O + qrow[t] * ldo;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                st4<bf16_t>(orow + 8 * g + 4 * h, make_float4(acc0[t][4 * g] * inv, acc0[t][4 * g + 1] * inv, acc0[t][4 * g + 2] * inv, acc0[t][4 * g + 3] * inv));
                st4<bf16_t>(orow + 32 + 8 * g + 4 * h, make_float4(acc1[t][4 * g] * inv, acc1[t][4 * g + 1] * inv, acc1[t][4 * g + 2] * inv, acc1[t][4 * g + 3] * inv));
            }
            if (h == 0) lse[qrow[t]] = (m[t] + log2f(tot)) * LN2;
        }
    }
}
#elif ATT_VAR == 11
// v11: 64 queries per wave (two 32-query sets share every K / V fragment read from LDS), 2 waves = 128 queries per workgroup,
// 128 threads, up to 4 workgroups per CU (2 waves per SIMD, 256 VGPRs each).
__global__ __launch_bounds__(128, 2) void attn_fwd_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                              const bf16_t* __restrict__ V, int ldv, long long skv, bf16_t* __restrict__ O,
                                                              int ldo, float* __restrict__ lse, Segs sg, int Nk, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[KB * LDR];
    __shared__ __attribute__((aligned(16))) bf16_t Vt[D * LDTB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    int row_base, q0, nq, b;
    locate_tile(sg, blockIdx.x, row_base, q0, nq, b);
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    bool ok[2];
    long long qrow[2];
    bf16x8 qf[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ql = q0 + wave * 64 + 32 * t + j;
        ok[t] = ql < nq;
        qrow[t] = (long long)row_base + ql;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint4 v = ok[t] ? *reinterpret_cast<const uint4*>(Q + qrow[t] * ldq + 16 * ks + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
            qf[t][ks] = *reinterpret_cast<const bf16x8*>(&v);
        }
    }
    const float qs = scale * LOG2E;
    f32x16 acc0[2], acc1[2];
    float m[2], lsum[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        m[t] = NEG_BIG; lsum[t] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[t][r] = 0.f; acc1[t][r] = 0.f; }
    }
    const int krow = pi_row(j);
    auto fmap = [&](int it, int& r, int& c8, int& g) {       // 2 waves x 8 iterations cover the 16 (16 keys x 32 d) chunks of a fill
        const int c = wave * 8 + it;
        g = lane >> 4; r = 16 * (c >> 1) + (lane & 15); c8 = 32 * (c & 1) + 8 * g;
    };
    uint4 kr[8];
    auto fetch = [&](int kb0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int r, c8, g; fmap(i, r, c8, g);
            kr[i] = ld_row8(Kb, ldk, kb0 + r, Nk, c8);
        }
    };
    fetch(0);
    for (int kb0 = 0; kb0 < Nk; kb0 += KB) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int r, c8, g; fmap(i, r, c8, g);
            *reinterpret_cast<uint4*>(&Ks[r * LDR + c8]) = kr[i];
            st_t8(Vt, LDTB, c8, r, ld_row8(Vb, ldv, kb0 + r, Nk, c8), g);
        }
        __syncthreads();
        if (kb0 + KB < Nk) fetch(kb0 + KB);
#pragma unroll 1
        for (int sub = 0; sub < KB / 32 && kb0 + 32 * sub < Nk; ++sub) {
            f32x16 s[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
            const bf16_t* kp = Ks + (32 * sub + krow) * LDR + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = ld_frag(kp + 16 * ks);
                s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[0][ks], s[0], 0, 0, 0);
                s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[1][ks], s[1], 0, 0, 0);
            }
            const int kv0 = kb0 + 32 * sub;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (kv0 + 32 > Nk) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) if (kv0 + 16 * h + r >= Nk) s[t][r] = NEG_BIG;
                }
                float mx = s[t][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * qs;
                if (__any(mx > m[t] + RESCALE_THR)) {
                    const float mn = fmaxf(m[t], mx);
                    const float alpha = fast_exp2(m[t] - mn);
                    lsum[t] *= alpha;
                    m[t] = mn;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { acc0[t][r] *= alpha; acc1[t][r] *= alpha; }
                }
                float rs = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[t][r] = fast_exp2(fmaf(s[t][r], qs, -m[t])); rs += s[t][r]; }
                lsum[t] += rs;                                   // per-half partial; halves meet after the loop
            }
            const bf16_t* vp = Vt + j * LDTB + 32 * sub + 16 * h;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const bf16x8 v0 = ld_frag(vp + 8 * k2), v1 = ld_frag(vp + 32 * LDTB + 8 * k2);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const bf16x8 pb = pack8(s[t], 8 * k2);
                    acc0[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, pb, acc0[t], 0, 0, 0);
                    acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pb, acc1[t], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float tot = lsum[t] + __shfl_xor(lsum[t], 32, 64);
        if (ok[t]) {
            const float inv = 1.0f / tot;
            bf16_t* orow = O + qrow[t] * ldo;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                st4<bf16_t>(orow + 8 * g + 4 * h, make_float4(acc0[t][4 * g] * inv, acc0[t][4 * g + 1] * inv, acc0[t][4 * g + 2] * inv, acc0[t][4 * g + 3] * inv));
                st4<bf16_t>(orow + 32 + 8 * g + 4 * h, make_float4(acc1[t][4 * g] * inv, acc1[t][4 * g + 1] * inv, acc1[t][4 * g + 2] * inv, acc1[t][4 * g + 3] * inv));
            }
            if (h == 0) lse[qrow[t]] = (m[t] + log2f(tot)) * LN2;
        }
    }
}
#elif ATT_VAR == 13
// v13 = the production kernel with V kept row-major (4x4 k-row permutation), gathered by ds_read_b64_tr_b16, its strips loaded together
__global__ __launch_bounds__(256, 4) void attn_fwd_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                              const bf16_t* __restrict__ V, int ldv, long long skv, bf16_t* __restrict__ O,
                                                              int ldo, float* __restrict__ lse, Segs sg, int Nk, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[KB * LDR];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[KB * LDR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    int row_base, q0, nq, b;
    locate_tile(sg, blockIdx.x, row_base, q0, nq, b);
    const int ql = q0 + wave * 32 + j;                     // query index inside this (segment, image)
    const bool ok = ql < nq;
    const long long qrow = (long long)row_base + ql;
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint4 v = ok ? *reinterpret_cast<const uint4*>(Q + qrow * ldq + 16 * ks + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
        qf[ks] = *reinterpret_cast<const bf16x8*>(&v);
    }
    const float qs = scale * LOG2E;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float m = NEG_BIG, lsum = 0.f;
    const int krow = pi_row(j);
    uint4 kr[4];                                            // K of the next fill is prefetched; V is fetched at its LDS store (keeps the
    auto fetch = [&](int kb0) {                            // kernel within 128 VGPRs: 4 workgroups per CU = all 800 tiles resident at once)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r, c8, g;
            fill_map(tid, i, r, c8, g);
            kr[i] = ld_row8(Kb, ldk, kb0 + r, Nk, c8);
        }
    };
    fetch(0);
    for (int kb0 = 0; kb0 < Nk; kb0 += KB) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r, c8, g;
            fill_map(tid, i, r, c8, g);
            *reinterpret_cast<uint4*>(&Ks[r * LDR + c8]) = kr[i];
        }
        {
            uint4 vr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r, c8, g;
                fill_map(tid, i, r, c8, g);
                vr[i] = *reinterpret_cast<const uint4*>(Vb + (long long)min(kb0 + r, Nk - 1) * ldv + c8);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r, c8, g;
                fill_map(tid, i, r, c8, g);
                const int pr = (r & ~15) | ((r & 3) << 2) | ((r >> 2) & 3);
                *reinterpret_cast<uint4*>(&Vs[pr * LDR + c8]) = vr[i];
            }
        }
        __syncthreads();
        if (kb0 + KB < Nk) fetch(kb0 + KB);
        // S^T tile of sub-tile `sub`: 4 chained MFMAs, issued asynchronously to the matrix pipe
        auto qk = [&](int sub) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const bf16_t* kp = Ks + (32 * sub + krow) * LDR + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(kp + 16 * ks), qf[ks], s, 0, 0, 0);
            return s;
        };
        // online softmax of a finished S^T tile (register VALU) followed by O^T += V^T P^T
        auto softmax_pv = [&](f32x16 s, int sub) {
            const int kv0 = kb0 + 32 * sub;
            if (kv0 + 32 > Nk) {                                // tail tile (wave-uniform)
#pragma unroll
                for (int r = 0; r < 16; ++r) if (kv0 + 16 * h + r >= Nk) s[r] = NEG_BIG;
            }
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * qs;        // scaled maximum of this tile
            if (__any(mx > m + RESCALE_THR)) {                  // lazy rescale: rare once the running maximum has settled
                const float mn = fmaxf(m, mx);
                const float alpha = fast_exp2(m - mn);
                lsum *= alpha;
                m = mn;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] *= alpha; acc1[r] *= alpha; }
            }
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = fast_exp2(fmaf(s[r], qs, -m)); rs += s[r]; }
            rs += __shfl_xor(rs, 32, 64);
            lsum += rs;
            const int gi = lane & 15, gq = (lane >> 4) & 1;
            const bf16_t* vp = Vs + (32 * sub + 16 * h + 4 * (gi >> 2)) * LDR + 16 * gq + 4 * (gi & 3);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const bf16x8 pb = pack8(s, 8 * k2);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + (2 * k2) * LDR, vp + (2 * k2 + 1) * LDR), pb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + (2 * k2) * LDR + 32, vp + (2 * k2 + 1) * LDR + 32), pb, acc1, 0, 0, 0);
            }
        };
#pragma unroll 1
        for (int sub = 0; sub < KB / 32 && kb0 + 32 * sub < Nk; ++sub) softmax_pv(qk(sub), sub);
    }
    if (ok) {
        const float inv = 1.0f / lsum;
        bf16_t* orow = O + qrow * ldo;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<bf16_t>(orow + 8 * g + 4 * h, make_float4(acc0[4 * g] * inv, acc0[4 * g + 1] * inv, acc0[4 * g + 2] * inv, acc0[4 * g + 3] * inv));
            st4<bf16_t>(orow + 32 + 8 * g + 4 * h, make_float4(acc1[4 * g] * inv, acc1[4 * g + 1] * inv, acc1[4 * g + 2] * inv, acc1[4 * g + 3] * inv));
        }
        if (h == 0) lse[qrow] = (m + log2f(lsum)) * LN2;
    }
}

#elif ATT_VAR == 14
// v14 = v13 + branch-free K prefetch
// v13 = the production kernel with V kept row-major (4x4 k-row permutation), gathered by ds_read_b64_tr_b16, its strips loaded together
__global__ __launch_bounds__(256, 4) void attn_fwd_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                              const bf16_t* __restrict__ V, int ldv, long long skv, bf16_t* __restrict__ O,
                                                              int ldo, float* __restrict__ lse, Segs sg, int Nk, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[KB * LDR];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[KB * LDR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    int row_base, q0, nq, b;
    locate_tile(sg, blockIdx.x, row_base, q0, nq, b);
    const int ql = q0 + wave * 32 + j;                     // query index inside this (segment, image)
    const bool ok = ql < nq;
    const long long qrow = (long long)row_base + ql;
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint4 v = ok ? *reinterpret_cast<const uint4*>(Q + qrow * ldq + 16 * ks + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
        qf[ks] = *reinterpret_cast<const bf16x8*>(&v);
    }
    const float qs = scale * LOG2E;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float m = NEG_BIG, lsum = 0.f;
    const int krow = pi_row(j);
    uint4 kr[4];                                            // K of the next fill is prefetched; V is fetched at its LDS store (keeps the
    auto fetch = [&](int kb0) {                            // kernel within 128 VGPRs: 4 workgroups per CU = all 800 tiles resident at once)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r, c8, g;
            fill_map(tid, i, r, c8, g);
            kr[i] = *reinterpret_cast<const uint4*>(Kb + (long long)min(kb0 + r, Nk - 1) * ldk + c8);
        }
    };
    fetch(0);
    for (int kb0 = 0; kb0 < Nk; kb0 += KB) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r, c8, g;
            fill_map(tid, i, r, c8, g);
            *reinterpret_cast<uint4*>(&Ks[r * LDR + c8]) = kr[i];
        }
        {
            uint4 vr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r, c8, g;
                fill_map(tid, i, r, c8, g);
                vr[i] = *reinterpret_cast<const uint4*>(Vb + (long long)min(kb0 + r, Nk - 1) * ldv + c8);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r, c8, g;
                fill_map(tid, i, r, c8, g);
                const int pr = (r & ~15) | ((r & 3) << 2) | ((r >> 2) & 3);
                *reinterpret_cast<uint4*>(&Vs[pr * LDR + c8]) = vr[i];
            }
        }
        __syncthreads();
        if (kb0 + KB < Nk) fetch(kb0 + KB);
        // S^T tile of sub-tile `sub`: 4 chained MFMAs, issued asynchronously to the matrix pipe
        auto qk = [&](int sub) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const bf16_t* kp = Ks + (32 * sub + krow) * LDR + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(kp + 16 * ks), qf[ks], s, 0, 0, 0);
            return s;
        };
        // online softmax of a finished S^T tile (register VALU) followed by O^T += V^T P^T
        auto softmax_pv = [&](f32x16 s, int sub) {
            const int kv0 = kb0 + 32 * sub;
            if (kv0 + 32 > Nk) {                                // tail tile (wave-uniform)
#pragma unroll
                for (int r = 0; r < 16; ++r) if (kv0 + 16 * h + r >= Nk) s[r] = NEG_BIG;
            }
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * qs;        // scaled maximum of this tile
            if (__any(mx > m + RESCALE_THR)) {                  // lazy rescale: rare once the running maximum has settled
                const float mn = fmaxf(m, mx);
                const float alpha = fast_exp2(m - mn);
                lsum *= alpha;
                m = mn;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] *= alpha; acc1[r] *= alpha; }
            }
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = fast_exp2(fmaf(s[r], qs, -m)); rs += s[r]; }
            rs += __shfl_xor(rs, 32, 64);
            lsum += rs;
            const int gi = lane & 15, gq = (lane >> 4) & 1;
            const bf16_t* vp = Vs + (32 * sub + 16 * h + 4 * (gi >> 2)) * LDR + 16 * gq + 4 * (gi & 3);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const bf16x8 pb = pack8(s, 8 * k2);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + (2 * k2) * LDR, vp + (2 * k2 + 1) * LDR), pb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + (2 * k2) * LDR + 32, vp + (2 * k2 + 1) * LDR + 32), pb, acc1, 0, 0, 0);
            }
        };
#pragma unroll 1
        for (int sub = 0; sub < KB / 32 && kb0 + 32 * sub < Nk; ++sub) softmax_pv(qk(sub), sub);
    }
    if (ok) {
        const float inv = 1.0f / lsum;
        bf16_t* orow = O + qrow * ldo;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<bf16_t>(orow + 8 * g + 4 * h, make_float4(acc0[4 * g] * inv, acc0[4 * g + 1] * inv, acc0[4 * g + 2] * inv, acc0[4 * g + 3] * inv));
            st4<bf16_t>(orow + 32 + 8 * g + 4 * h, make_float4(acc1[4 * g] * inv, acc1[4 * g + 1] * inv, acc1[4 * g + 2] * inv, acc1[4 * g + 3] * inv));
        }
        if (h == 0) lse[qrow] = (m + log2f(lsum)) * LN2;
    }
}

#elif ATT_VAR == 15
// v15 = v13 + permlane32 swap for the cross-half max, per-half row sums
// v13 = the production kernel with V kept row-major (4x4 k-row permutation), gathered by ds_read_b64_tr_b16, its strips loaded together
__global__ __launch_bounds__(256, 4) void attn_fwd_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                              const bf16_t* __restrict__ V, int ldv, long long skv, bf16_t* __restrict__ O,
                                                              int ldo, float* __restrict__ lse, Segs sg, int Nk, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[KB * LDR];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[KB * LDR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    int row_base, q0, nq, b;
    locate_tile(sg, blockIdx.x, row_base, q0, nq, b);
    const int ql = q0 + wave * 32 + j;                     // query index inside this (segment, image)
    const bool ok = ql < nq;
    const long long qrow = (long long)row_base + ql;
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint4 v = ok ? *reinterpret_cast<const uint4*>(Q + qrow * ldq + 16 * ks + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
        qf[ks] = *reinterpret_cast<const bf16x8*>(&v);
    }
    const float qs = scale * LOG2E;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float m = NEG_BIG, lsum = 0.f;
    const int krow = pi_row(j);
    uint4 kr[4];                                            // K of the next fill is prefetched; V is fetched at its LDS store (keeps the
    auto fetch = [&](int kb0) {                            // kernel within 128 VGPRs: 4 workgroups per CU = all 800 tiles resident at once)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r, c8, g;
            fill_map(tid, i, r, c8, g);
            kr[i] = ld_row8(Kb, ldk, kb0 + r, Nk, c8);
        }
    };
    fetch(0);
    for (int kb0 = 0; kb0 < Nk; kb0 += KB) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r, c8, g;
            fill_map(tid, i, r, c8, g);
            *reinterpret_cast<uint4*>(&Ks[r * LDR + c8]) = kr[i];
        }
        {
            uint4 vr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r, c8, g;
                fill_map(tid, i, r, c8, g);
                vr[i] = *reinterpret_cast<const uint4*>(Vb + (long long)min(kb0 + r, Nk - 1) * ldv + c8);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r, c8, g;
                fill_map(tid, i, r, c8, g);
                const int pr = (r & ~15) | ((r & 3) << 2) | ((r >> 2) & 3);
                *reinterpret_cast<uint4*>(&Vs[pr * LDR + c8]) = vr[i];
            }
        }
        __syncthreads();
        if (kb0 + KB < Nk) fetch(kb0 + KB);
        // S^T tile of sub-tile `sub`: 4 chained MFMAs, issued asynchronously to the matrix pipe
        auto qk = [&](int sub) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const bf16_t* kp = Ks + (32 * sub + krow) * LDR + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(kp + 16 * ks), qf[ks], s, 0, 0, 0);
            return s;
        };
        // online softmax of a finished S^T tile (register VALU) followed by O^T += V^T P^T
        auto softmax_pv = [&](f32x16 s, int sub) {
            const int kv0 = kb0 + 32 * sub;
            if (kv0 + 32 > Nk) {                                // tail tile (wave-uniform)
#pragma unroll
                for (int r = 0; r < 16; ++r) if (kv0 + 16 * h + r >= Nk) s[r] = NEG_BIG;
            }
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            {
                const unsigned u = __float_as_uint(mx);
                const auto pr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                mx = fmaxf(__uint_as_float(pr[0]), __uint_as_float(pr[1])) * qs;
            }
            if (__any(mx > m + RESCALE_THR)) {                  // lazy rescale: rare once the running maximum has settled
                const float mn = fmaxf(m, mx);
                const float alpha = fast_exp2(m - mn);
                lsum *= alpha;
                m = mn;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] *= alpha; acc1[r] *= alpha; }
            }
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = fast_exp2(fmaf(s[r], qs, -m)); rs += s[r]; }
            lsum += rs;
            const int gi = lane & 15, gq = (lane >> 4) & 1;
            const bf16_t* vp = Vs + (32 * sub + 16 * h + 4 * (gi >> 2)) * LDR + 16 * gq + 4 * (gi & 3);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const bf16x8 pb = pack8(s, 8 * k2);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + (2 * k2) * LDR, vp + (2 * k2 + 1) * LDR), pb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + (2 * k2) * LDR + 32, vp + (2 * k2 + 1) * LDR + 32), pb, acc1, 0, 0, 0);
            }
        };
#pragma unroll 1
        for (int sub = 0; sub < KB / 32 && kb0 + 32 * sub < Nk; ++sub) softmax_pv(qk(sub), sub);
    }
    lsum += __shfl_xor(lsum, 32, 64);
    if (ok) {
        const float inv = 1.0f / lsum;
        bf16_t* orow = O + qrow * ldo;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<bf16_t>(orow + 8 * g + 4 * h, make_float4(acc0[4 * g] * inv, acc0[4 * g + 1] * inv, acc0[4 * g + 2] * inv, acc0[4 * g + 3] * inv));
            st4<bf16_t>(orow + 32 + 8 * g + 4 * h, make_float4(acc1[4 * g] * inv, acc1[4 * g + 1] * inv, acc1[4 * g + 2] * inv, acc1[4 * g + 3] * inv));
        }
        if (h == 0) lse[qrow] = (m + log2f(lsum)) * LN2;
    }
}

#elif ATT_VAR == 16
// v16 = v15 with wave tiles flattened per image: 48 workgroups per image = 768 = 3 per CU exactly
// v15 = v13 + permlane32 swap for the cross-half max, per-half row sums
// v13 = the production kernel with V kept row-major (4x4 k-row permutation), gathered by ds_read_b64_tr_b16, its strips loaded together
__global__ __launch_bounds__(256, 4) void attn_fwd_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                              const bf16_t* __restrict__ V, int ldv, long long skv, bf16_t* __restrict__ O,
                                                              int ldo, float* __restrict__ lse, Segs sg, int Nk, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[KB * LDR];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[KB * LDR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int nwt = sg.t32[sg.n], bpi = (nwt + 3) >> 2;
    const int b = blockIdx.x / bpi, wt = (blockIdx.x - b * bpi) * 4 + wave;
    int sgi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < sg.n && wt >= sg.t32[i]) sgi = i;
    const int nq = sg.nq[sgi];
    const int ql = (wt - sg.t32[sgi]) * 32 + j;
    const bool ok = wt < nwt && ql < nq;
    const long long qrow = (long long)sg.row0[sgi] + (long long)b * nq + ql;
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint4 v = ok ? *reinterpret_cast<const uint4*>(Q + qrow * ldq + 16 * ks + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
        qf[ks] = *reinterpret_cast<const bf16x8*>(&v);
    }
    const float qs = scale * LOG2E;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float m = NEG_BIG, lsum = 0.f;
    const int krow = pi_row(j);
    uint4 kr[4];                                            // K of the next fill is prefetched; V is fetched at its LDS store (keeps the
    auto fetch = [&](int kb0) {                            // kernel within 128 VGPRs: 4 workgroups per CU = all 800 tiles resident at once)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r, c8, g;
            fill_map(tid, i, r, c8, g);
            kr[i] = ld_row8(Kb, ldk, kb0 + r, Nk, c8);
        }
    };
    fetch(0);
    for (int kb0 = 0; kb0 < Nk; kb0 += KB) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r, c8, g;
            fill_map(tid, i, r, c8, g);
            *reinterpret_cast<uint4*>(&Ks[r * LDR + c8]) = kr[i];
        }
        {
            uint4 vr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r, c8, g;
                fill_map(tid, i, r, c8, g);
                vr[i] = *reinterpret_cast<const uint4*>(Vb + (long long)min(kb0 + r, Nk - 1) * ldv + c8);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r, c8, g;
                fill_map(tid, i, r, c8, g);
                const int pr = (r & ~15) | ((r & 3) << 2) | ((r >> 2) & 3);
                *reinterpret_cast<uint4*>(&Vs[pr * LDR + c8]) = vr[i];
            }
        }
        __syncthreads();
        if (kb0 + KB < Nk) fetch(kb0 + KB);
        // S^T tile of sub-tile `sub`: 4 chained MFMAs, issued asynchronously to the matrix pipe
        auto qk = [&](int sub) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const bf16_t* kp = Ks + (32 * sub + krow) * LDR + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(kp + 16 * ks), qf[ks], s, 0, 0, 0);
            return s;
        };
        // online softmax of a finished S^T tile (register VALU) followed by O^T += V^T P^T
        auto softmax_pv = [&](f32x16 s, int sub) {
            const int kv0 = kb0 + 32 * sub;
            if (kv0 + 32 > Nk) {                                // tail tile (wave-uniform)
#pragma unroll
                for (int r = 0; r < 16; ++r) if (kv0 + 16 * h + r >= Nk) s[r] = NEG_BIG;
            }
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            {
                const unsigned u = __float_as_uint(mx);
                const auto pr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                mx = fmaxf(__uint_as_float(pr[0]), __uint_as_float(pr[1])) * qs;
            }
            if (__any(mx > m + RESCALE_THR)) {                  // lazy rescale: rare once the running maximum has settled
                const float mn = fmaxf(m, mx);
                const float alpha = fast_exp2(m - mn);
                lsum *= alpha;
                m = mn;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] *= alpha; acc1[r] *= alpha; }
            }
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = fast_exp2(fmaf(s[r], qs, -m)); rs += s[r]; }
            lsum += rs;
            const int gi = lane & 15, gq = (lane >> 4) & 1;
            const bf16_t* vp = Vs + (32 * sub + 16 * h + 4 * (gi >> 2)) * LDR + 16 * gq + 4 * (gi & 3);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const bf16x8 pb = pack8(s, 8 * k2);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + (2 * k2) * LDR, vp + (2 * k2 + 1) * LDR), pb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + (2 * k2) * LDR + 32, vp + (2 * k2 + 1) * LDR + 32), pb, acc1, 0, 0, 0);
            }
        };
#pragma unroll 1
        for (int sub = 0; sub < KB / 32 && kb0 + 32 * sub < Nk; ++sub) softmax_pv(qk(sub), sub);
    }
    lsum += __shfl_xor(lsum, 32, 64);
    if (ok) {
        const float inv = 1.0f / lsum;
        bf16_t* orow = O + qrow * ldo;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<bf16_t>(orow + 8 * g + 4 * h, make_float4(acc0[4 * g] * inv, acc0[4 * g + 1] * inv, acc0[4 * g + 2] * inv, acc0[4 * g + 3] * inv));
            st4<bf16_t>(orow + 32 + 8 * g + 4 * h, make_float4(acc1[4 * g] * inv, acc1[4 * g + 1] * inv, acc1[4 * g + 2] * inv, acc1[4 * g + 3] * inv));
        }
        if (h == 0) lse[qrow] = (m + log2f(lsum)) * LN2;
    }
}

#elif ATT_VAR >= 30 && ATT_VAR < 40
// v30: ONE workgroup of NW30 = 12 waves (768 threads) per CU: 16 workgroups per image = 256 at B = 16.  The twelve 32-query wave tiles share
// one K/V staging (a third of the L2 -> LDS traffic of the 4-wave workgroups); the 128-key K/V tiles are double-buffered in LDS (72 KB)
// and fetched a whole tile ahead into 3 registers per thread, so a tile costs ONE barrier and no exposed load latency.
// v31 = v30 + the QK^T MFMAs of the next 32-key sub-tile issued before the softmax of the current one (within a 128-key tile)
// v32 = v31 + explicit MFMA / VALU interleave (sched_group_barrier)
#ifndef NW30
#define NW30 12
#endif
#ifndef KB30
#define KB30 128
#endif
#ifdef TIMING
__device__ long long g_dbg[4 * NW30 * 24];
#define TSTAMP(k) do { if ((blockIdx.x & 63) == 5 && blockIdx.x < 256 && lane == 0) g_dbg[((blockIdx.x >> 6) * NW30 + wave) * 24 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define TSTAMP(k)
#endif
constexpr int NT30 = NW30 * 64, NC30 = KB30 * 8, NF30 = (2 * NC30 + NT30 - 1) / NT30;   // NC30 16-byte chunks per K (or V) tile
__global__ __launch_bounds__(NT30, 1) void attn_fwd_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                               const bf16_t* __restrict__ V, int ldv, long long skv, bf16_t* __restrict__ O,
                                                               int ldo, float* __restrict__ lse, Segs sg, int Nk, float scale) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem30[];      // [2 slots][K tile | V tile][KB][LDR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int nwt = sg.t32[sg.n], bpi = (nwt + NW30 - 1) / NW30;
    const int b = blockIdx.x / bpi, wt = (blockIdx.x - b * bpi) * NW30 + wave;
    int sgi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < sg.n && wt >= sg.t32[i]) sgi = i;
    const int nq = sg.nq[sgi];
    const int ql = (wt - sg.t32[sgi]) * 32 + j;
    const bool ok = wt < nwt && ql < nq;
    const long long qrow = (long long)sg.row0[sgi] + (long long)b * nq + ql;
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint4 v = ok ? *reinterpret_cast<const uint4*>(Q + qrow * ldq + 16 * ks + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
        qf[ks] = *reinterpret_cast<const bf16x8*>(&v);
    }
    const float qs = scale * LOG2E;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float m = NEG_BIG, lsum = 0.f;
    const int krow = pi_row(j);
    // staging: chunk id = tid + i * NT30; ids 0..1023 are K (row = id >> 3, 16-byte chunk id & 7), 1024..2047 V
    uint4 st[NF30];
    auto fetch = [&](int kb0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NF30; ++i) {
            const int id = tid + i * NT30, isv = id >= NC30, r = (id - isv * NC30) >> 3, c8 = (id & 7) * 8;
            const bf16_t* src = isv ? Vb : Kb;
            const int ld = isv ? ldv : ldk;
            st[i] = id < 2 * NC30 ? *reinterpret_cast<const uint4*>(src + (long long)min(kb0 + r, Nk - 1) * ld + c8) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto stash = [&](int slot) __attribute__((always_inline)) {
        bf16_t* base = smem30 + slot * (2 * KB30 * LDR);
#pragma unroll
        for (int i = 0; i < NF30; ++i) {
            const int id = tid + i * NT30, isv = id >= NC30, r = (id - isv * NC30) >> 3, c8 = (id & 7) * 8;
            const int pr = isv ? ((r & ~15) | ((r & 3) << 2) | ((r >> 2) & 3)) : r;
            if (id < 2 * NC30) *reinterpret_cast<uint4*>(base + isv * (KB30 * LDR) + pr * LDR + c8) = st[i];
        }
    };
    const int nt = (Nk + KB30 - 1) / KB30;
    TSTAMP(0);
    fetch(0);
    stash(0);
    if (nt > 1) fetch(KB30);
    __syncthreads();
    TSTAMP(1);
    for (int t = 0; t < nt; ++t) {
        const int kb0 = t * KB30;
#if ATT_VAR != 34 && ATT_VAR != 35 && ATT_VAR != 38 && ATT_VAR != 39
#ifndef STG_NOSTASH
        if (t + 1 < nt) stash((t + 1) & 1);
#endif
#ifndef STG_NOFETCH
        if (t + 2 < nt) fetch(kb0 + 2 * KB30);
#endif
        const bf16_t* Ks = smem30 + (t & 1) * (2 * KB30 * LDR);
#else
        const bf16_t* Ks = smem30;
#endif
        const bf16_t* Vs = Ks + KB30 * LDR;
        auto qk = [&](int sub) __attribute__((always_inline)) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const bf16_t* kp = Ks + (32 * sub + krow) * LDR + 8 * h;
#pragma unroll
#if ATT_VAR == 38
            for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[(ks + sub) & 3], qf[ks], s, 0, 0, 0);
#else
            for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(kp + 16 * ks), qf[ks], s, 0, 0, 0);
#endif
            return s;
        };
        auto softmax_pv = [&](f32x16 s, int sub) __attribute__((always_inline)) {
            const int kv0 = kb0 + 32 * sub;
            if (kv0 + 32 > Nk) {                                // tail tile (wave-uniform)
#pragma unroll
                for (int r = 0; r < 16; ++r) if (kv0 + 16 * h + r >= Nk) s[r] = NEG_BIG;
            }
#if ATT_VAR == 33 || ATT_VAR == 35 || ATT_VAR == 38 || ATT_VAR == 39
            (void)qs;
#else
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            {
                const unsigned u = __float_as_uint(mx);
                const auto pr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                mx = fmaxf(__uint_as_float(pr[0]), __uint_as_float(pr[1])) * qs;
            }
            if (__any(mx > m + RESCALE_THR)) {                  // lazy rescale: rare once the running maximum has settled
                const float mn = fmaxf(m, mx);
                const float alpha = fast_exp2(m - mn);
                lsum *= alpha;
                m = mn;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] *= alpha; acc1[r] *= alpha; }
            }
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = fast_exp2(fmaf(s[r], qs, -m)); rs += s[r]; }
            lsum += rs;
#endif
            const int gi = lane & 15, gq = (lane >> 4) & 1;
            const bf16_t* vp = Vs + (32 * sub + 16 * h + 4 * (gi >> 2)) * LDR + 16 * gq + 4 * (gi & 3);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const bf16x8 pb = pack8(s, 8 * k2);
#if ATT_VAR == 38 || ATT_VAR == 39
                (void)vp;
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[k2], pb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[2 + k2], pb, acc1, 0, 0, 0);
#else
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + (2 * k2) * LDR, vp + (2 * k2 + 1) * LDR), pb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + (2 * k2) * LDR + 32, vp + (2 * k2 + 1) * LDR + 32), pb, acc1, 0, 0, 0);
#endif
            }
        };
        const int nsub = min((KB30 + 31) / 32, (min(Nk, kb0 + KB30) - kb0 + 31) / 32);
#if ATT_VAR == 30 || (ATT_VAR >= 33 && ATT_VAR <= 39)
#pragma unroll 1
        for (int sub = 0; sub < nsub; ++sub) {
#ifdef ROTPRIO
            // the three waves of a SIMD (wave, wave + 4, wave + 8) take turns at the top priority, so that none of them lags
            const int pr = ((wave >> 2) + sub + t) % 3;
            if (pr == 0) __builtin_amdgcn_s_setprio(ROTPRIO == 1 ? 2 : 0);
            else if (pr == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(ROTPRIO == 1 ? 0 : 2);
#endif
            softmax_pv(qk(sub), sub);
        }
#else
        f32x16 sc = qk(0);
#pragma unroll 1
        for (int sub = 0; sub < nsub; ++sub) {
            f32x16 sn = sc;
            if (sub + 1 < nsub) sn = qk(sub + 1);
            softmax_pv(sc, sub);
            sc = sn;
        }
#endif
        TSTAMP(2 + 2 * t);
#if ATT_VAR != 34 && ATT_VAR != 35 && ATT_VAR != 38 && ATT_VAR != 39 && !defined(STG_NOBAR)
        __syncthreads();
#endif
        TSTAMP(3 + 2 * t);
    }
    lsum += __shfl_xor(lsum, 32, 64);
    if (ok) {
        const float inv = 1.0f / lsum;
        bf16_t* orow = O + qrow * ldo;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<bf16_t>(orow + 8 * g + 4 * h, make_float4(acc0[4 * g] * inv, acc0[4 * g + 1] * inv, acc0[4 * g + 2] * inv, acc0[4 * g + 3] * inv));
            st4<bf16_t>(orow + 32 + 8 * g + 4 * h, make_float4(acc1[4 * g] * inv, acc1[4 * g + 1] * inv, acc1[4 * g + 2] * inv, acc1[4 * g + 3] * inv));
        }
        if (h == 0) lse[qrow] = (m + log2f(lsum)) * LN2;
    }
    TSTAMP(20);
}
#ifdef TIMING
} extern "C" int tc_dbg_read(long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_dbg), sizeof(long long) * 4 * NW30 * 24); } namespace {
#endif

#elif ATT_VAR >= 90 && ATT_VAR < 92
// v30: ONE workgroup of NW30 = 12 waves (768 threads) per CU: 16 workgroups per image = 256 at B = 16.  The twelve 32-query wave tiles share
// one K/V staging (a third of the L2 -> LDS traffic of the 4-wave workgroups); the 128-key K/V tiles are double-buffered in LDS (72 KB)
// and fetched a whole tile ahead into 3 registers per thread, so a tile costs ONE barrier and no exposed load latency.
// v31 = v30 + the QK^T MFMAs of the next 32-key sub-tile issued before the softmax of the current one (within a 128-key tile)
// v32 = v31 + explicit MFMA / VALU interleave (sched_group_barrier)
#ifndef NW30
#define NW30 12
#endif
#ifndef KB30
#define KB30 128
#endif
#ifdef TIMING
__device__ long long g_dbg[4 * NW30 * 24];
#define TSTAMP(k) do { if ((blockIdx.x & 63) == 5 && blockIdx.x < 256 && lane == 0) g_dbg[((blockIdx.x >> 6) * NW30 + wave) * 24 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define TSTAMP(k)
#endif
constexpr int NT30 = NW30 * 64, NC30 = KB30 * 8, NF30 = (2 * NC30 + NT30 - 1) / NT30;   // NC30 16-byte chunks per K (or V) tile
__global__ __launch_bounds__(NT30, 1) void attn_fwd_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                               const bf16_t* __restrict__ V, int ldv, long long skv, bf16_t* __restrict__ O,
                                                               int ldo, float* __restrict__ lse, Segs sg, int Nk, float scale) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem30[];      // [2 slots][K tile | V tile][KB][LDR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    TSTAMP(18);
    const int nwt = sg.t32[sg.n], bpi = (nwt + NW30 - 1) / NW30;
    const int b = blockIdx.x / bpi, wt = (blockIdx.x - b * bpi) * NW30 + wave;
    int sgi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < sg.n && wt >= sg.t32[i]) sgi = i;
    const int nq = sg.nq[sgi];
    const int ql = (wt - sg.t32[sgi]) * 32 + j;
    const bool ok = wt < nwt && ql < nq;
    const long long qrow = (long long)sg.row0[sgi] + (long long)b * nq + ql;
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    bf16x8 qf[4];
#if ATT_VAR >= 91
    // Q tile of the wave: four coalesced loads (eight lanes per 128-byte row) into a wave-private LDS tile, fragments read back from
    // there -- a per-lane load of 16 bytes at a row stride touches 32 lines per instruction and queues in the address unit
    bf16_t* wtile = smem30 + 2 * (2 * KB30 * LDR) + wave * (32 * LDR);
    {
        const int tq0 = (wt - sg.t32[sgi]) * 32;                  // first query of the tile within its segment-image
        const long long trow0 = (long long)sg.row0[sgi] + (long long)b * nq + tq0;
        uint4 qv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 8 * i + (lane >> 3);
            qv[i] = (wt < nwt && tq0 + r < nq) ? *reinterpret_cast<const uint4*>(Q + (trow0 + r) * ldq + 8 * (lane & 7)) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(wtile + (8 * i + (lane >> 3)) * LDR + 8 * (lane & 7)) = qv[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = ld_frag(wtile + j * LDR + 16 * ks + 8 * h);
    }
#else
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint4 v = ok ? *reinterpret_cast<const uint4*>(Q + qrow * ldq + 16 * ks + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
        qf[ks] = *reinterpret_cast<const bf16x8*>(&v);
    }
#endif
    const float qs = scale * LOG2E;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float m = NEG_BIG, lsum = 0.f;
    const int krow = pi_row(j);
    // staging: chunk id = tid + i * NT30; ids 0..1023 are K (row = id >> 3, 16-byte chunk id & 7), 1024..2047 V
    uint4 st[NF30];
    auto fetch = [&](int kb0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NF30; ++i) {
            const int id = tid + i * NT30, isv = id >= NC30, r = (id - isv * NC30) >> 3, c8 = (id & 7) * 8;
            const bf16_t* src = isv ? Vb : Kb;
            const int ld = isv ? ldv : ldk;
            st[i] = id < 2 * NC30 ? *reinterpret_cast<const uint4*>(src + (long long)min(kb0 + r, Nk - 1) * ld + c8) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto stash = [&](int slot) __attribute__((always_inline)) {
        bf16_t* base = smem30 + slot * (2 * KB30 * LDR);
#pragma unroll
        for (int i = 0; i < NF30; ++i) {
            const int id = tid + i * NT30, isv = id >= NC30, r = (id - isv * NC30) >> 3, c8 = (id & 7) * 8;
            const int pr = isv ? ((r & ~15) | ((r & 3) << 2) | ((r >> 2) & 3)) : r;
            if (id < 2 * NC30) *reinterpret_cast<uint4*>(base + isv * (KB30 * LDR) + pr * LDR + c8) = st[i];
        }
    };
    const int nt = (Nk + KB30 - 1) / KB30;
    TSTAMP(0);
    fetch(0);
    stash(0);
    if (nt > 1) fetch(KB30);
    __syncthreads();
    TSTAMP(1);
    for (int t = 0; t < nt; ++t) {
        const int kb0 = t * KB30;
#if ATT_VAR != 34 && ATT_VAR != 35 && ATT_VAR != 38 && ATT_VAR != 39
#ifndef STG_NOSTASH
        if (t + 1 < nt) stash((t + 1) & 1);
#endif
#ifndef STG_NOFETCH
        if (t + 2 < nt) fetch(kb0 + 2 * KB30);
#endif
        const bf16_t* Ks = smem30 + (t & 1) * (2 * KB30 * LDR);
#else
        const bf16_t* Ks = smem30;
#endif
        const bf16_t* Vs = Ks + KB30 * LDR;
        auto qk = [&](int sub) __attribute__((always_inline)) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const bf16_t* kp = Ks + (32 * sub + krow) * LDR + 8 * h;
#pragma unroll
#if ATT_VAR == 38
            for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[(ks + sub) & 3], qf[ks], s, 0, 0, 0);
#else
            for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(kp + 16 * ks), qf[ks], s, 0, 0, 0);
#endif
            return s;
        };
        auto softmax_pv = [&](f32x16 s, int sub) __attribute__((always_inline)) {
            const int kv0 = kb0 + 32 * sub;
            if (kv0 + 32 > Nk) {                                // tail tile (wave-uniform)
#pragma unroll
                for (int r = 0; r < 16; ++r) if (kv0 + 16 * h + r >= Nk) s[r] = NEG_BIG;
            }
            // Softmax against a reference exponent m that is NOT the running row maximum: any m within fp32's exponent range of
            // the true maximum gives the same quotient, so m is set once (integer-valued, from the first sub-tile) and raised only
            // when a row sum shows that the scores have outgrown it by 2^30 -- one compare per sub-tile instead of a 16-element
            // maximum, a lane swap and a compare.  Integer m: P differs from the max-referenced P by an exact power of two.
            if (m == NEG_BIG) {                                 // first sub-tile of the row (wave-uniform: all rows start together)
                float mx = s[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
                const unsigned u = __float_as_uint(mx);
                const auto pr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                m = ceilf(fmaxf(__uint_as_float(pr[0]), __uint_as_float(pr[1])) * qs);
            }
            f32x16 pv;
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { pv[r] = fast_exp2(fmaf(s[r], qs, -m)); rs += pv[r]; }
            if (__any(!(rs < 1073741824.0f))) {                 // rare: some row grew past 2^30 times its reference; re-reference all rows
                float mx = s[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
                const unsigned u = __float_as_uint(mx);
                const auto pr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                const float mn = fmaxf(m, ceilf(fmaxf(__uint_as_float(pr[0]), __uint_as_float(pr[1])) * qs));
                const float alpha = fast_exp2(m - mn);
                lsum *= alpha;
                m = mn;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] *= alpha; acc1[r] *= alpha; }
                rs = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { pv[r] = fast_exp2(fmaf(s[r], qs, -m)); rs += pv[r]; }
            }
            lsum += rs;
            s = pv;
            const int gi = lane & 15, gq = (lane >> 4) & 1;
            const bf16_t* vp = Vs + (32 * sub + 16 * h + 4 * (gi >> 2)) * LDR + 16 * gq + 4 * (gi & 3);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const bf16x8 pb = pack8(s, 8 * k2);
#if ATT_VAR == 38 || ATT_VAR == 39
                (void)vp;
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[k2], pb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[2 + k2], pb, acc1, 0, 0, 0);
#else
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + (2 * k2) * LDR, vp + (2 * k2 + 1) * LDR), pb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + (2 * k2) * LDR + 32, vp + (2 * k2 + 1) * LDR + 32), pb, acc1, 0, 0, 0);
#endif
            }
        };
        const int nsub = min((KB30 + 31) / 32, (min(Nk, kb0 + KB30) - kb0 + 31) / 32);
#ifdef PAIRSUB
        {   // two sub-tiles' QK^T chains back to back, then their softmax / PV: one MFMA -> VALU drain per pair
            int sub = 0;
#pragma unroll 1
            for (; sub + 1 < nsub; sub += 2) {
                const f32x16 s0 = qk(sub), s1 = qk(sub + 1);
                softmax_pv(s0, sub);
                softmax_pv(s1, sub + 1);
            }
            if (sub < nsub) softmax_pv(qk(sub), sub);
        }
#elif 1
#pragma unroll 1
        for (int sub = 0; sub < nsub; ++sub) {
#ifdef ROTPRIO
            // the three waves of a SIMD (wave, wave + 4, wave + 8) take turns at the top priority, so that none of them lags
            const int pr = ((wave >> 2) + sub + t) % 3;
            if (pr == 0) __builtin_amdgcn_s_setprio(ROTPRIO == 1 ? 2 : 0);
            else if (pr == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(ROTPRIO == 1 ? 0 : 2);
#endif
            softmax_pv(qk(sub), sub);
        }
#else
        f32x16 sc = qk(0);
#pragma unroll 1
        for (int sub = 0; sub < nsub; ++sub) {
            f32x16 sn = sc;
            if (sub + 1 < nsub) sn = qk(sub + 1);
            softmax_pv(sc, sub);
            sc = sn;
        }
#endif
        TSTAMP(2 + 2 * t);
#if ATT_VAR != 34 && ATT_VAR != 35 && ATT_VAR != 38 && ATT_VAR != 39 && !defined(STG_NOBAR)
        __syncthreads();
#endif
        TSTAMP(3 + 2 * t);
    }
    TSTAMP(19);
    lsum += __shfl_xor(lsum, 32, 64);
#if ATT_VAR >= 91
    {   // O tile through the wave-private LDS tile: whole 128-byte rows leave with 16-byte stores
        const float inv = 1.0f / lsum;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<bf16_t>(wtile + j * LDR + 8 * g + 4 * h, make_float4(acc0[4 * g] * inv, acc0[4 * g + 1] * inv, acc0[4 * g + 2] * inv, acc0[4 * g + 3] * inv));
            st4<bf16_t>(wtile + j * LDR + 32 + 8 * g + 4 * h, make_float4(acc1[4 * g] * inv, acc1[4 * g + 1] * inv, acc1[4 * g + 2] * inv, acc1[4 * g + 3] * inv));
        }
        if (ok && h == 0) lse[qrow] = (m + log2f(lsum)) * LN2;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int tq0 = (wt - sg.t32[sgi]) * 32;
        const long long trow0 = (long long)sg.row0[sgi] + (long long)b * nq + tq0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 8 * i + (lane >> 3);
            const uint4 v = *reinterpret_cast<const uint4*>(wtile + r * LDR + 8 * (lane & 7));
            if (wt < nwt && tq0 + r < nq) *reinterpret_cast<uint4*>(O + (trow0 + r) * ldo + 8 * (lane & 7)) = v;
        }
    }
#else
    if (ok) {
        const float inv = 1.0f / lsum;
        bf16_t* orow = O + qrow * ldo;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<bf16_t>(orow + 8 * g + 4 * h, make_float4(acc0[4 * g] * inv, acc0[4 * g + 1] * inv, acc0[4 * g + 2] * inv, acc0[4 * g + 3] * inv));
            st4<bf16_t>(orow + 32 + 8 * g + 4 * h, make_float4(acc1[4 * g] * inv, acc1[4 * g + 1] * inv, acc1[4 * g + 2] * inv, acc1[4 * g + 3] * inv));
        }
        if (h == 0) lse[qrow] = (m + log2f(lsum)) * LN2;
    }
#endif
    TSTAMP(20);
}
#ifdef TIMING
} extern "C" int tc_dbg_read(long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_dbg), sizeof(long long) * 4 * NW30 * 24); } namespace {
#endif

#elif ATT_VAR >= 40 && ATT_VAR < 50
// v40: v30's one 12-wave workgroup per CU, with the matrix work of a wave re-ordered so that no MFMA follows one it depends on:
// the four chained QK^T MFMAs of sub-tile i+1 are interleaved one-for-one with the four PV MFMAs of sub-tile i (dependent MFMAs
// issued back to back are paced by their latency and leave holes in the matrix pipe that no other wave can use).  K/V tiles in a
// 3-slot LDS ring (108 KB), one barrier per 128-key tile; fragment reads issued before the softmax arithmetic that hides them.
#ifndef NW30
#define NW30 12
#endif
#define KB30 128
constexpr int NT30 = NW30 * 64, NC30 = KB30 * 8, NF30 = (2 * NC30 + NT30 - 1) / NT30, SLOT30 = 2 * KB30 * LDR;
__global__ __launch_bounds__(NT30, 1) void attn_fwd_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                               const bf16_t* __restrict__ V, int ldv, long long skv, bf16_t* __restrict__ O,
                                                               int ldo, float* __restrict__ lse, Segs sg, int Nk, float scale) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem30[];      // [3 slots][K tile | V tile][KB30][LDR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int nwt = sg.t32[sg.n], bpi = (nwt + NW30 - 1) / NW30;
    const int b = blockIdx.x / bpi, wt = (blockIdx.x - b * bpi) * NW30 + wave;
    int sgi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < sg.n && wt >= sg.t32[i]) sgi = i;
    const int nq = sg.nq[sgi];
    const int ql = (wt - sg.t32[sgi]) * 32 + j;
    const bool ok = wt < nwt && ql < nq;
    const long long qrow = (long long)sg.row0[sgi] + (long long)b * nq + ql;
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint4 v = ok ? *reinterpret_cast<const uint4*>(Q + qrow * ldq + 16 * ks + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
        qf[ks] = *reinterpret_cast<const bf16x8*>(&v);
    }
    const float qs = scale * LOG2E;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float m = NEG_BIG, lsum = 0.f;
    uint4 st[NF30];
    auto fetch = [&](int kb0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NF30; ++i) {
            const int id = tid + i * NT30, isv = id >= NC30, r = (id - isv * NC30) >> 3, c8 = (id & 7) * 8;
            const bf16_t* src = isv ? Vb : Kb;
            const int ld = isv ? ldv : ldk;
            st[i] = id < 2 * NC30 ? *reinterpret_cast<const uint4*>(src + (long long)min(kb0 + r, Nk - 1) * ld + c8) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto stash = [&](int slot) __attribute__((always_inline)) {
        bf16_t* base = smem30 + slot * SLOT30;
#pragma unroll
        for (int i = 0; i < NF30; ++i) {
            const int id = tid + i * NT30, isv = id >= NC30, r = (id - isv * NC30) >> 3, c8 = (id & 7) * 8;
            const int pr = isv ? ((r & ~15) | ((r & 3) << 2) | ((r >> 2) & 3)) : r;
            if (id < 2 * NC30) *reinterpret_cast<uint4*>(base + isv * (KB30 * LDR) + pr * LDR + c8) = st[i];
        }
    };
    const int nt = (Nk + KB30 - 1) / KB30, nsubs = (Nk + 31) / 32;
    const int koff = pi_row(j) * LDR + 8 * h;                                               // K fragment of this lane within a 32-key sub-tile
    const int voff = (16 * h + 4 * ((lane & 15) >> 2)) * LDR + 16 * ((lane >> 4) & 1) + 4 * (lane & 3) + KB30 * LDR;   // V^T fragment
    fetch(0);
    stash(0);
    if (nt > 1) fetch(KB30);
    __syncthreads();
    bf16x8 kf[4], vf[4];
    f32x16 S;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(smem30 + koff + 16 * ks), qf[ks], S, 0, 0, 0);
    int slot = 0, nslot = 1;                                    // ring slot of the current tile / of the next
#pragma unroll 1
    for (int i = 0; i < nsubs; ++i) {
        const int t = i >> 2, sub = i & 3;
        if (sub == 0) {
            if (t + 1 < nt) stash(nslot);
            if (t + 2 < nt) fetch((t + 2) * KB30);
        }
        const bf16_t* vp = smem30 + slot * SLOT30 + 32 * sub * LDR + voff;
#if ATT_VAR == 40
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            vf[2 * k2] = ld_frag_tr(vp + (2 * k2) * LDR, vp + (2 * k2 + 1) * LDR);
            vf[2 * k2 + 1] = ld_frag_tr(vp + (2 * k2) * LDR + 32, vp + (2 * k2 + 1) * LDR + 32);
        }
#endif
        const bool has_next = i + 1 < nsubs;
        if (sub == 3 && has_next) __syncthreads();              // the next tile (stored during this one) is complete
        const bf16_t* kp = smem30 + (sub == 3 ? nslot : slot) * SLOT30 + 32 * ((sub + 1) & 3) * LDR + koff;
#if ATT_VAR != 42
        if (has_next) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[ks] = ld_frag(kp + 16 * ks);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        // online softmax of S (sub-tile i)
        const int kv0 = 32 * i;
        if (kv0 + 32 > Nk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) if (kv0 + 16 * h + r >= Nk) S[r] = NEG_BIG;
        }
        float mx = S[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, S[r]);
        {
            const unsigned u = __float_as_uint(mx);
            const auto pr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            mx = fmaxf(__uint_as_float(pr[0]), __uint_as_float(pr[1])) * qs;
        }
        if (__any(mx > m + RESCALE_THR)) {
            const float mn = fmaxf(m, mx);
            const float alpha = fast_exp2(m - mn);
            lsum *= alpha;
            m = mn;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] *= alpha; acc1[r] *= alpha; }
        }
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { S[r] = fast_exp2(fmaf(S[r], qs, -m)); rs += S[r]; }
        lsum += rs;
        const bf16x8 pb0 = pack8(S, 0), pb1 = pack8(S, 8);
        __builtin_amdgcn_sched_barrier(0);
#if ATT_VAR != 40
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            vf[2 * k2] = ld_frag_tr(vp + (2 * k2) * LDR, vp + (2 * k2 + 1) * LDR);
            vf[2 * k2 + 1] = ld_frag_tr(vp + (2 * k2) * LDR + 32, vp + (2 * k2 + 1) * LDR + 32);
        }
#endif
#if ATT_VAR == 42
        if (has_next) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[ks] = ld_frag(kp + 16 * ks);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (has_next) {
            f32x16 Sn;
#pragma unroll
            for (int r = 0; r < 16; ++r) Sn[r] = 0.f;
            Sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], Sn, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0], pb0, acc0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            Sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1], qf[1], Sn, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1], pb0, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            Sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2], qf[2], Sn, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[2], pb1, acc0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            Sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[3], qf[3], Sn, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[3], pb1, acc1, 0, 0, 0);
            S = Sn;
        } else {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0], pb0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1], pb0, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[2], pb1, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[3], pb1, acc1, 0, 0, 0);
        }
        if (sub == 3) { slot = nslot; nslot = nslot == 2 ? 0 : nslot + 1; }
    }
    lsum += __shfl_xor(lsum, 32, 64);
    if (ok) {
        const float inv = 1.0f / lsum;
        bf16_t* orow = O + qrow * ldo;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<bf16_t>(orow + 8 * g + 4 * h, make_float4(acc0[4 * g] * inv, acc0[4 * g + 1] * inv, acc0[4 * g + 2] * inv, acc0[4 * g + 3] * inv));
            st4<bf16_t>(orow + 32 + 8 * g + 4 * h, make_float4(acc1[4 * g] * inv, acc1[4 * g + 1] * inv, acc1[4 * g + 2] * inv, acc1[4 * g + 3] * inv));
        }
        if (h == 0) lse[qrow] = (m + log2f(lsum)) * LN2;
    }
}

#elif ATT_VAR >= 60 && ATT_VAR < 70
// v60: ONE wave per SIMD.  A workgroup = 4 waves, each wave owns NQ60 = 3 32-query tiles (96 queries) and the whole 512-register
// file: the three tiles share every K / V^T fragment read (a third of the LDS read traffic), their independent MFMA and softmax
// streams interleave inside the wave (no reliance on the arbitration between co-resident waves, which lets the youngest wave of a
// SIMD lag and makes every barrier wait for it), and the four waves of the workgroup are symmetric, so barriers cost little.
// Software pipeline: the QK^T MFMAs of key sub-tile i+1 are issued inside the exp / PV phase of sub-tile i.
// 16 workgroups per image = 256 at B = 16.  K/V tiles of 128 keys in a 3-slot LDS ring, one barrier per tile.
#ifndef NQ60
#define NQ60 3
#endif
#define KB30 128
constexpr int NT30 = 256, NC30 = KB30 * 8, NF30 = (2 * NC30 + NT30 - 1) / NT30, SLOT30 = 2 * KB30 * LDR;
__global__ __launch_bounds__(256, 1) void attn_fwd_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                              const bf16_t* __restrict__ V, int ldv, long long skv, bf16_t* __restrict__ O,
                                                              int ldo, float* __restrict__ lse, Segs sg, int Nk, float scale) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem30[];      // [3 slots][K tile | V tile][KB30][LDR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int nwt = sg.t32[sg.n], bpi = (nwt + 4 * NQ60 - 1) / (4 * NQ60);
    const int b = blockIdx.x / bpi, wt0 = ((blockIdx.x - b * bpi) * 4 + wave) * NQ60;
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    bool ok[NQ60];
    long long qrow[NQ60];
    bf16x8 qf[NQ60][4];
    f32x16 acc0[NQ60], acc1[NQ60], S[NQ60];
    float m[NQ60], lsum[NQ60];
#pragma unroll
    for (int q = 0; q < NQ60; ++q) {
        const int wt = wt0 + q;
        int sgi = 0;
#pragma unroll
        for (int i = 1; i < 4; ++i) if (i < sg.n && wt >= sg.t32[i]) sgi = i;
        const int nq = sg.nq[sgi];
        const int ql = (wt - sg.t32[sgi]) * 32 + j;
        ok[q] = wt < nwt && ql < nq;
        qrow[q] = (long long)sg.row0[sgi] + (long long)b * nq + ql;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint4 v = ok[q] ? *reinterpret_cast<const uint4*>(Q + qrow[q] * ldq + 16 * ks + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
            qf[q][ks] = *reinterpret_cast<const bf16x8*>(&v);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[q][r] = 0.f; acc1[q][r] = 0.f; }
        m[q] = NEG_BIG; lsum[q] = 0.f;
    }
    const float qs = scale * LOG2E;
    uint4 st[NF30];
    auto fetch = [&](int kb0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NF30; ++i) {
            const int id = tid + i * NT30, isv = id >= NC30, r = (id - isv * NC30) >> 3, c8 = (id & 7) * 8;
            const bf16_t* src = isv ? Vb : Kb;
            const int ld = isv ? ldv : ldk;
            st[i] = *reinterpret_cast<const uint4*>(src + (long long)min(kb0 + r, Nk - 1) * ld + c8);
        }
    };
    auto stash = [&](int slot) __attribute__((always_inline)) {
        bf16_t* base = smem30 + slot * SLOT30;
#pragma unroll
        for (int i = 0; i < NF30; ++i) {
            const int id = tid + i * NT30, isv = id >= NC30, r = (id - isv * NC30) >> 3, c8 = (id & 7) * 8;
            const int pr = isv ? ((r & ~15) | ((r & 3) << 2) | ((r >> 2) & 3)) : r;
            *reinterpret_cast<uint4*>(base + isv * (KB30 * LDR) + pr * LDR + c8) = st[i];
        }
    };
    const int nt = (Nk + KB30 - 1) / KB30, nsubs = (Nk + 31) / 32;
    const int koff = pi_row(j) * LDR + 8 * h;
    const int voff = (16 * h + 4 * ((lane & 15) >> 2)) * LDR + 16 * ((lane >> 4) & 1) + 4 * (lane & 3) + KB30 * LDR;
    fetch(0);
    stash(0);
    if (nt > 1) fetch(KB30);
    __syncthreads();
    {
        bf16x8 kf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[ks] = ld_frag(smem30 + koff + 16 * ks);
#pragma unroll
        for (int q = 0; q < NQ60; ++q) {
#pragma unroll
            for (int r = 0; r < 16; ++r) S[q][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) S[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[q][ks], S[q], 0, 0, 0);
        }
    }
    int slot = 0, nslot = 1;
    // one key sub-tile: NEXT = the QK^T of sub-tile i + 1 is issued here
    auto step = [&](int i, auto next_tag) __attribute__((always_inline)) {
        constexpr bool NEXT = decltype(next_tag)::value;
        const int t = i >> 2, sub = i & 3;
        if (sub == 0) {
            if (t + 1 < nt) stash(nslot);
            if (t + 2 < nt) fetch((t + 2) * KB30);
        }
        if (sub == 3 && NEXT) __syncthreads();                  // the next tile (stored during this one) is complete
        const int kv0 = 32 * i;
        if (kv0 + 32 > Nk) {
#pragma unroll
            for (int q = 0; q < NQ60; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) if (kv0 + 16 * h + r >= Nk) S[q][r] = NEG_BIG;
        }
        float mx[NQ60];
        bool grow = false;
#pragma unroll
        for (int q = 0; q < NQ60; ++q) {
            float v = S[q][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) v = fmaxf(v, S[q][r]);
            const unsigned u = __float_as_uint(v);
            const auto pr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            mx[q] = fmaxf(__uint_as_float(pr[0]), __uint_as_float(pr[1])) * qs;
            grow |= mx[q] > m[q] + RESCALE_THR;
        }
        if (__any(grow)) {                                      // lazy rescale of all tiles of the wave: rare once the maxima have settled
#pragma unroll
            for (int q = 0; q < NQ60; ++q) {
                const float mn = fmaxf(m[q], mx[q]);
                const float alpha = fast_exp2(m[q] - mn);
                lsum[q] *= alpha;
                m[q] = mn;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[q][r] *= alpha; acc1[q][r] *= alpha; }
            }
        }
        const bf16_t* vp = smem30 + slot * SLOT30 + 32 * sub * LDR + voff;
        bf16x8 vf[4], kf[4];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            vf[2 * k2] = ld_frag_tr(vp + (2 * k2) * LDR, vp + (2 * k2 + 1) * LDR);
            vf[2 * k2 + 1] = ld_frag_tr(vp + (2 * k2) * LDR + 32, vp + (2 * k2 + 1) * LDR + 32);
        }
        if (NEXT) {
            const bf16_t* kp = smem30 + (sub == 3 ? nslot : slot) * SLOT30 + 32 * ((sub + 1) & 3) * LDR + koff;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[ks] = ld_frag(kp + 16 * ks);
        }
#pragma unroll
        for (int q = 0; q < NQ60; ++q) {
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { S[q][r] = fast_exp2(fmaf(S[q][r], qs, -m[q])); rs += S[q][r]; }
            lsum[q] += rs;
            const bf16x8 pb0 = pack8(S[q], 0), pb1 = pack8(S[q], 8);
            acc0[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0], pb0, acc0[q], 0, 0, 0);
            acc1[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1], pb0, acc1[q], 0, 0, 0);
            acc0[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[2], pb1, acc0[q], 0, 0, 0);
            acc1[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[3], pb1, acc1[q], 0, 0, 0);
            if (NEXT) {
#pragma unroll
                for (int r = 0; r < 16; ++r) S[q][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) S[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[q][ks], S[q], 0, 0, 0);
            }
        }
        if (sub == 3) { slot = nslot; nslot = nslot == 2 ? 0 : nslot + 1; }
    };
#pragma unroll 1
    for (int i = 0; i + 1 < nsubs; ++i) step(i, std::true_type{});
    step(nsubs - 1, std::false_type{});
#pragma unroll
    for (int q = 0; q < NQ60; ++q) {
        const float ls = lsum[q] + __shfl_xor(lsum[q], 32, 64);
        if (ok[q]) {
            const float inv = 1.0f / ls;
            bf16_t* orow = O + qrow[q] * ldo;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                st4<bf16_t>(orow + 8 * g + 4 * h, make_float4(acc0[q][4 * g] * inv, acc0[q][4 * g + 1] * inv, acc0[q][4 * g + 2] * inv, acc0[q][4 * g + 3] * inv));
                st4<bf16_t>(orow + 32 + 8 * g + 4 * h, make_float4(acc1[q][4 * g] * inv, acc1[q][4 * g + 1] * inv, acc1[q][4 * g + 2] * inv, acc1[q][4 * g + 3] * inv));
            }
            if (h == 0) lse[qrow[q]] = (m[q] + log2f(ls)) * LN2;
        }
    }
}

#elif ATT_VAR >= 70 && ATT_VAR < 80
// v70: v30's 12-wave workgroup per CU with the matrix and softmax work interleaved INSIDE each wave.  Measured on gfx950
// (scripts/exp/overlap.hip): MFMAs of one wave and VALU work of another wave on the same SIMD do not overlap (time = sum), while
// independent VALU instructions placed behind an MFMA in the SAME wave run under it (time = max).  So each loop iteration carries,
// in program order, the max / exp / sum / pack arithmetic of key sub-tile i interleaved with eight MFMAs that do not depend on it:
// PV of sub-tile i-1 and QK^T of sub-tile i+1.  3-slot LDS ring of 128-key K/V tiles, one barrier per tile.
#ifndef NW30
#define NW30 12
#endif
#define KB30 128
constexpr int NT30 = NW30 * 64, NC30 = KB30 * 8, NF30 = (2 * NC30 + NT30 - 1) / NT30, SLOT30 = 2 * KB30 * LDR;
__global__ __launch_bounds__(NT30, 1) void attn_fwd_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                               const bf16_t* __restrict__ V, int ldv, long long skv, bf16_t* __restrict__ O,
                                                               int ldo, float* __restrict__ lse, Segs sg, int Nk, float scale) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem30[];      // [3 slots][K tile | V tile][KB30][LDR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int nwt = sg.t32[sg.n], bpi = (nwt + NW30 - 1) / NW30;
    const int b = blockIdx.x / bpi, wt = (blockIdx.x - b * bpi) * NW30 + wave;
    int sgi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < sg.n && wt >= sg.t32[i]) sgi = i;
    const int nq = sg.nq[sgi];
    const int ql = (wt - sg.t32[sgi]) * 32 + j;
    const bool ok = wt < nwt && ql < nq;
    const long long qrow = (long long)sg.row0[sgi] + (long long)b * nq + ql;
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint4 v = ok ? *reinterpret_cast<const uint4*>(Q + qrow * ldq + 16 * ks + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
        qf[ks] = *reinterpret_cast<const bf16x8*>(&v);
    }
    const float qs = scale * LOG2E;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float m = NEG_BIG, lsum = 0.f;
    uint4 st[NF30];
    auto fetch = [&](int kb0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NF30; ++i) {
            const int id = tid + i * NT30, isv = id >= NC30, r = (id - isv * NC30) >> 3, c8 = (id & 7) * 8;
            const bf16_t* src = isv ? Vb : Kb;
            const int ld = isv ? ldv : ldk;
            st[i] = id < 2 * NC30 ? *reinterpret_cast<const uint4*>(src + (long long)min(kb0 + r, Nk - 1) * ld + c8) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto stash = [&](int slot) __attribute__((always_inline)) {
        bf16_t* base = smem30 + slot * SLOT30;
#pragma unroll
        for (int i = 0; i < NF30; ++i) {
            const int id = tid + i * NT30, isv = id >= NC30, r = (id - isv * NC30) >> 3, c8 = (id & 7) * 8;
            const int pr = isv ? ((r & ~15) | ((r & 3) << 2) | ((r >> 2) & 3)) : r;
            if (id < 2 * NC30) *reinterpret_cast<uint4*>(base + isv * (KB30 * LDR) + pr * LDR + c8) = st[i];
        }
    };
    const int nt = (Nk + KB30 - 1) / KB30, nsubs = (Nk + 31) / 32;
    const int koff = pi_row(j) * LDR + 8 * h;
    const int voff = (16 * h + 4 * ((lane & 15) >> 2)) * LDR + 16 * ((lane >> 4) & 1) + 4 * (lane & 3) + KB30 * LDR;
    fetch(0);
    stash(0);
    if (nt > 1) fetch(KB30);
    __syncthreads();
    bf16x8 vf[4], pb0, pb1;                                     // V^T fragments and packed P of the PREVIOUS sub-tile
    f32x16 SA, SB;
#pragma unroll
    for (int i = 0; i < 8; ++i) { pb0[i] = (__bf16)0.f; pb1[i] = (__bf16)0.f; }
#pragma unroll
    for (int r = 0; r < 16; ++r) SA[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) SA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(smem30 + koff + 16 * ks), qf[ks], SA, 0, 0, 0);
    {
        const bf16_t* vp = smem30 + voff;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            vf[2 * k2] = ld_frag_tr(vp + (2 * k2) * LDR, vp + (2 * k2 + 1) * LDR);
            vf[2 * k2 + 1] = ld_frag_tr(vp + (2 * k2) * LDR + 32, vp + (2 * k2 + 1) * LDR + 32);
        }
    }
    int slot = 0, nslot = 1;
    // iteration i: S = scores of sub-tile i (finished), Sn <- scores of sub-tile i + 1, acc += P(i-1) V(i-1)
    auto step = [&](f32x16& S, f32x16& Sn, int i, auto last_tag) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_tag)::value;
        const int t = i >> 2, sub = i & 3;
        if (sub == 0) {
            if (t + 1 < nt) stash(nslot);
            if (t + 2 < nt) fetch((t + 2) * KB30);
        }
        if (sub == 3 && !LAST) __syncthreads();                 // the next tile (stored during this one) is complete
        bf16x8 kf[4];
        if (!LAST) {
            const bf16_t* kp = smem30 + (sub == 3 ? nslot : slot) * SLOT30 + 32 * ((sub + 1) & 3) * LDR + koff;
#pragma unroll
#if ATT_VAR >= 72
            for (int ks = 0; ks < 4; ++ks) kf[ks] = *(const volatile bf16x8 __attribute__((address_space(3)))*)(kp + 16 * ks);   // volatile: not sunk behind the branch
#else
            for (int ks = 0; ks < 4; ++ks) kf[ks] = ld_frag(kp + 16 * ks);
#endif
        }
        if (LAST) {
            const int kv0 = 32 * i;
            if (kv0 + 32 > Nk) {
#pragma unroll
                for (int r = 0; r < 16; ++r) if (kv0 + 16 * h + r >= Nk) S[r] = NEG_BIG;
            }
        }
        // ---- phase 1: row maxima of S, under the PV MFMAs of the previous sub-tile
#if ATT_VAR >= 74
#define FENCE() __builtin_amdgcn_sched_barrier(0)
        float mx, mxb;
        FENCE();
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0], pb0, acc0, 0, 0, 0);
        FENCE();
        mx = fmaxf(fmaxf(S[0], S[1]), S[2]); mx = fmaxf(fmaxf(mx, S[3]), S[4]);
        FENCE();
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1], pb0, acc1, 0, 0, 0);
        FENCE();
        mxb = fmaxf(fmaxf(S[5], S[6]), S[7]); mxb = fmaxf(fmaxf(mxb, S[8]), S[9]); mx = fmaxf(mx, mxb);
        FENCE();
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[2], pb1, acc0, 0, 0, 0);
        FENCE();
        mxb = fmaxf(fmaxf(S[10], S[11]), S[12]); mxb = fmaxf(fmaxf(mxb, S[13]), S[14]); mx = fmaxf(fmaxf(mx, mxb), S[15]);
        FENCE();
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[3], pb1, acc1, 0, 0, 0);
        FENCE();
        {
            const unsigned u = __float_as_uint(mx);
            const auto pr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            mx = fmaxf(__uint_as_float(pr[0]), __uint_as_float(pr[1])) * qs;
        }
#else
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0], pb0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1], pb0, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[2], pb1, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[3], pb1, acc1, 0, 0, 0);
        float mx = S[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, S[r]);
        {
            const unsigned u = __float_as_uint(mx);
            const auto pr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            mx = fmaxf(__uint_as_float(pr[0]), __uint_as_float(pr[1])) * qs;
        }
#endif
#if ATT_VAR == 71 || ATT_VAR == 73
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x2, 5, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x2, 5, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x2, 5, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
#endif
        if (__any(mx > m + RESCALE_THR)) {                      // lazy rescale: rare once the running maximum has settled
            const float mn = fmaxf(m, mx);
            const float alpha = fast_exp2(m - mn);
            lsum *= alpha;
            m = mn;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] *= alpha; acc1[r] *= alpha; }
        }
        // ---- phase 2: exp / row sum / pack of S, under the QK^T MFMAs of the next sub-tile; V^T fragments of this sub-tile
        const bf16_t* vp = smem30 + slot * SLOT30 + 32 * sub * LDR + voff;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            vf[2 * k2] = ld_frag_tr(vp + (2 * k2) * LDR, vp + (2 * k2 + 1) * LDR);
            vf[2 * k2 + 1] = ld_frag_tr(vp + (2 * k2) * LDR + 32, vp + (2 * k2 + 1) * LDR + 32);
        }
#if ATT_VAR >= 74
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) Sn[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            FENCE();
            if (!LAST) Sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], Sn, 0, 0, 0);
            FENCE();
#pragma unroll
            for (int r = 4 * ks; r < 4 * ks + 4; ++r) { S[r] = fast_exp2(fmaf(S[r], qs, -m)); rs += S[r]; }
            if (ks == 1) pb0 = pack8(S, 0);
            if (ks == 3) pb1 = pack8(S, 8);
        }
        FENCE();
        lsum += rs;
#else
        if (!LAST) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Sn[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) Sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], Sn, 0, 0, 0);
        }
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { S[r] = fast_exp2(fmaf(S[r], qs, -m)); rs += S[r]; }
        lsum += rs;
        pb0 = pack8(S, 0);
        pb1 = pack8(S, 8);
#endif
#if ATT_VAR == 71 || ATT_VAR == 73
        if (!LAST) {
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x2, 14, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x2, 14, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x2, 14, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        }
#endif
        if (sub == 3) { slot = nslot; nslot = nslot == 2 ? 0 : nslot + 1; }
    };
    int i = 0;
#pragma unroll 1
    for (; i + 2 < nsubs; i += 2) { step(SA, SB, i, std::false_type{}); step(SB, SA, i + 1, std::false_type{}); }
    if (i + 2 == nsubs) { step(SA, SB, i, std::false_type{}); step(SB, SA, i + 1, std::true_type{}); }
    else step(SA, SB, i, std::true_type{});
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0], pb0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1], pb0, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[2], pb1, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[3], pb1, acc1, 0, 0, 0);
    lsum += __shfl_xor(lsum, 32, 64);
    if (ok) {
        const float inv = 1.0f / lsum;
        bf16_t* orow = O + qrow * ldo;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<bf16_t>(orow + 8 * g + 4 * h, make_float4(acc0[4 * g] * inv, acc0[4 * g + 1] * inv, acc0[4 * g + 2] * inv, acc0[4 * g + 3] * inv));
            st4<bf16_t>(orow + 32 + 8 * g + 4 * h, make_float4(acc1[4 * g] * inv, acc1[4 * g + 1] * inv, acc1[4 * g + 2] * inv, acc1[4 * g + 3] * inv));
        }
        if (h == 0) lse[qrow] = (m + log2f(lsum)) * LN2;
    }
}

#elif ATT_VAR >= 80 && ATT_VAR < 90
// v80: ONE wave per SIMD, three 32-query tiles per wave (as v60), with the register files and the instruction order under control:
//   * measured on gfx950 (scripts/exp/overlap.hip): an MFMA that finds the matrix pipe busy holds the SIMD's VALU issue, so MFMAs and
//     VALU work of DIFFERENT waves on one SIMD add up, while independent VALU instructions issued behind an MFMA by the SAME wave run
//     under it.  Hence one wave per SIMD and >= 32 cycles of softmax arithmetic behind every MFMA, in program order (sched_barrier fences).
//   * MFMAs are inline asm so that the O accumulators (96 registers) and the Q fragments (48) live in AGPRs and the score tiles in
//     arch VGPRs (hipcc put every MFMA result in AGPRs and copied 140 registers per iteration, v60).  Hazards of asm MFMAs are not
//     tracked by the compiler: results are first read >= one loop phase later, behind explicit s_nops.
//   Per key sub-tile: max of 3 tiles under 8 QK^T MFMAs of the next sub-tile; exp/sum/pack of tile q under the remaining QK^T and
//   the PV MFMAs of tile q-1.  3-slot LDS ring of 128-key K/V tiles, one barrier per tile, 16 workgroups per image.
#define NQ60 3
#define KB30 128
constexpr int NT30 = 256, NC30 = KB30 * 8, NF30 = (2 * NC30 + NT30 - 1) / NT30, SLOT30 = 2 * KB30 * LDR;
#define FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ void mfma_acc(f32x16& c, const bf16x8& a, const bf16x8& b) {        // c (AGPR) += a b
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_s0(f32x16& c, const bf16x8& a, const bf16x8& b) {         // c (VGPR) = a b, b in AGPR
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma_s(f32x16& c, const bf16x8& a, const bf16x8& b) {          // c (VGPR) += a b, b in AGPR
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
}
__global__ __launch_bounds__(256, 1) void attn_fwd_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                              const bf16_t* __restrict__ V, int ldv, long long skv, bf16_t* __restrict__ O,
                                                              int ldo, float* __restrict__ lse, Segs sg, int Nk, float scale) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem30[];      // [3 slots][K tile | V tile][KB30][LDR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int nwt = sg.t32[sg.n], bpi = (nwt + 4 * NQ60 - 1) / (4 * NQ60);
    const int b = blockIdx.x / bpi, wt0 = ((blockIdx.x - b * bpi) * 4 + wave) * NQ60;
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    bool ok[NQ60];
    long long qrow[NQ60];
    bf16x8 qf[NQ60][4];
    f32x16 acc0[NQ60], acc1[NQ60], SA[NQ60], SB[NQ60];
    float m[NQ60], lsum[NQ60];
#pragma unroll
    for (int q = 0; q < NQ60; ++q) {
        const int wt = wt0 + q;
        int sgi = 0;
#pragma unroll
        for (int i = 1; i < 4; ++i) if (i < sg.n && wt >= sg.t32[i]) sgi = i;
        const int nq = sg.nq[sgi];
        const int ql = (wt - sg.t32[sgi]) * 32 + j;
        ok[q] = wt < nwt && ql < nq;
        qrow[q] = (long long)sg.row0[sgi] + (long long)b * nq + ql;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint4 v = ok[q] ? *reinterpret_cast<const uint4*>(Q + qrow[q] * ldq + 16 * ks + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
            qf[q][ks] = *reinterpret_cast<const bf16x8*>(&v);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[q][r] = 0.f; acc1[q][r] = 0.f; }
        m[q] = NEG_BIG; lsum[q] = 0.f;
    }
    const float qs = scale * LOG2E;
    uint4 st[NF30];
    auto fetch = [&](int kb0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NF30; ++i) {
            const int id = tid + i * NT30, isv = id >= NC30, r = (id - isv * NC30) >> 3, c8 = (id & 7) * 8;
            const bf16_t* src = isv ? Vb : Kb;
            const int ld = isv ? ldv : ldk;
            st[i] = *reinterpret_cast<const uint4*>(src + (long long)min(kb0 + r, Nk - 1) * ld + c8);
        }
    };
    auto stash = [&](int slot) __attribute__((always_inline)) {
        bf16_t* base = smem30 + slot * SLOT30;
#pragma unroll
        for (int i = 0; i < NF30; ++i) {
            const int id = tid + i * NT30, isv = id >= NC30, r = (id - isv * NC30) >> 3, c8 = (id & 7) * 8;
            const int pr = isv ? ((r & ~15) | ((r & 3) << 2) | ((r >> 2) & 3)) : r;
            *reinterpret_cast<uint4*>(base + isv * (KB30 * LDR) + pr * LDR + c8) = st[i];
        }
    };
    const int nt = (Nk + KB30 - 1) / KB30, nsubs = (Nk + 31) / 32;
    const int koff = pi_row(j) * LDR + 8 * h;
    const int voff = (16 * h + 4 * ((lane & 15) >> 2)) * LDR + 16 * ((lane >> 4) & 1) + 4 * (lane & 3) + KB30 * LDR;
    // fragment loads of key sub-tile i (tile i >> 2, ring slot (i >> 2) % 3)
    auto kaddr = [&](int i) __attribute__((always_inline)) { return smem30 + ((i >> 2) % 3) * SLOT30 + 32 * (i & 3) * LDR + koff; };
    auto vaddr = [&](int i) __attribute__((always_inline)) { return smem30 + ((i >> 2) % 3) * SLOT30 + 32 * (i & 3) * LDR + voff; };
    bf16x8 kf[4], vf[4];
    auto load_k = [&](int i) __attribute__((always_inline)) {
        const bf16_t* kp = kaddr(i);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[ks] = ld_frag(kp + 16 * ks);
    };
    auto load_v = [&](int i) __attribute__((always_inline)) {
        const bf16_t* vp = vaddr(i);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            vf[2 * k2] = ld_frag_tr(vp + (2 * k2) * LDR, vp + (2 * k2 + 1) * LDR);
            vf[2 * k2 + 1] = ld_frag_tr(vp + (2 * k2) * LDR + 32, vp + (2 * k2 + 1) * LDR + 32);
        }
    };
    fetch(0);
    stash(0);
    if (nt > 1) fetch(KB30);
    __syncthreads();
    load_k(0);
#pragma unroll
    for (int q = 0; q < NQ60; ++q) {
        mfma_s0(SA[q], kf[0], qf[q][0]);
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) mfma_s(SA[q], kf[ks], qf[q][ks]);
    }
    if (nsubs > 1) load_k(1);                                   // (sub-tile 1 is in tile 0)
    // one key sub-tile.  S: finished scores of sub-tile i; Sn: receives the scores of sub-tile i + 1 (kf holds its K fragments)
    auto step = [&](f32x16 (&S)[NQ60], f32x16 (&Sn)[NQ60], int i, auto last_tag) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_tag)::value;
        const int t = i >> 2, sub = i & 3;
        if (sub == 0) {
            if (t + 1 < nt) stash((t + 1) % 3);
            if (t + 2 < nt) fetch((t + 2) * KB30);
        }
        if (sub == 2 && i + 2 < nsubs) __syncthreads();         // tile t + 1 (stored at sub 0) is complete before its first K fragments are read below
        load_v(i);
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");        // asm-MFMA results (S) are read by VALU below
        if (LAST) {
            const int kv0 = 32 * i;
            if (kv0 + 32 > Nk) {
#pragma unroll
                for (int q = 0; q < NQ60; ++q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) if (kv0 + 16 * h + r >= Nk) S[q][r] = NEG_BIG;
            }
        }
        // ---- phase M: row maxima of the three tiles, under the QK^T MFMAs (next sub-tile) of tiles 0 and 1
        float mx[NQ60];
        bool grow = false;
#pragma unroll
        for (int q = 0; q < NQ60; ++q) {
            float v = S[q][0], w;
            FENCE();
            if (!LAST && q < 2) mfma_s0(Sn[q], kf[0], qf[q][0]);
            FENCE();
            v = fmaxf(fmaxf(v, S[q][1]), S[q][2]); v = fmaxf(fmaxf(v, S[q][3]), S[q][4]);
            FENCE();
            if (!LAST && q < 2) mfma_s(Sn[q], kf[1], qf[q][1]);
            FENCE();
            w = fmaxf(fmaxf(S[q][5], S[q][6]), S[q][7]); w = fmaxf(fmaxf(w, S[q][8]), S[q][9]); v = fmaxf(v, w);
            FENCE();
            if (!LAST && q < 2) mfma_s(Sn[q], kf[2], qf[q][2]);
            FENCE();
            w = fmaxf(fmaxf(S[q][10], S[q][11]), S[q][12]); w = fmaxf(fmaxf(w, S[q][13]), S[q][14]); v = fmaxf(fmaxf(v, w), S[q][15]);
            FENCE();
            if (!LAST && q < 2) mfma_s(Sn[q], kf[3], qf[q][3]);
            FENCE();
            const unsigned u = __float_as_uint(v);
            const auto pr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            mx[q] = fmaxf(__uint_as_float(pr[0]), __uint_as_float(pr[1])) * qs;
            grow |= mx[q] > m[q] + RESCALE_THR;
        }
        if (__any(grow)) {                                      // lazy rescale of all tiles of the wave: rare once the maxima have settled
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
#pragma unroll
            for (int q = 0; q < NQ60; ++q) {
                const float mn = fmaxf(m[q], mx[q]);
                const float alpha = fast_exp2(m[q] - mn);
                lsum[q] *= alpha;
                m[q] = mn;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[q][r] *= alpha; acc1[q][r] *= alpha; }
            }
        }
        // ---- phase E: exp / row sum / pack of tile q under 4 MFMAs: QK^T (next) of tile 2 for q = 0, PV of tile q - 1 otherwise
        bf16x8 pb0, pb1, pp0, pp1;
#pragma unroll
        for (int q = 0; q < NQ60; ++q) {
            float rs = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                FENCE();
                if (q == 0) {
                    if (!LAST) { if (g == 0) mfma_s0(Sn[2], kf[0], qf[2][0]); else mfma_s(Sn[2], kf[g], qf[2][g]); }
                } else {
                    if (g & 1) mfma_acc(acc1[q - 1], vf[g], g < 2 ? pp0 : pp1); else mfma_acc(acc0[q - 1], vf[g], g < 2 ? pp0 : pp1);
                }
                FENCE();
#pragma unroll
                for (int r = 4 * g; r < 4 * g + 4; ++r) { S[q][r] = fast_exp2(fmaf(S[q][r], qs, -m[q])); rs += S[q][r]; }
                if (g == 1) pb0 = pack8(S[q], 0);
                if (g == 3) pb1 = pack8(S[q], 8);
            }
            FENCE();
            lsum[q] += rs;
            pp0 = pb0; pp1 = pb1;
        }
        // PV of the last tile; the K fragments of sub-tile i + 2 are fetched under it
        FENCE();
        asm volatile("s_nop 1" ::: "memory");
        mfma_acc(acc0[NQ60 - 1], vf[0], pp0);
        mfma_acc(acc1[NQ60 - 1], vf[1], pp0);
        if (i + 2 < nsubs) load_k(i + 2);
        mfma_acc(acc0[NQ60 - 1], vf[2], pp1);
        mfma_acc(acc1[NQ60 - 1], vf[3], pp1);
        FENCE();
    };
    int i = 0;
#pragma unroll 1
    for (; i + 2 < nsubs; i += 2) { step(SA, SB, i, std::false_type{}); step(SB, SA, i + 1, std::false_type{}); }
    if (i + 2 == nsubs) { step(SA, SB, i, std::false_type{}); step(SB, SA, i + 1, std::true_type{}); }
    else step(SA, SB, i, std::true_type{});
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
#pragma unroll
    for (int q = 0; q < NQ60; ++q) {
        const float ls = lsum[q] + __shfl_xor(lsum[q], 32, 64);
        if (ok[q]) {
            const float inv = 1.0f / ls;
            bf16_t* orow = O + qrow[q] * ldo;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                st4<bf16_t>(orow + 8 * g + 4 * h, make_float4(acc0[q][4 * g] * inv, acc0[q][4 * g + 1] * inv, acc0[q][4 * g + 2] * inv, acc0[q][4 * g + 3] * inv));
                st4<bf16_t>(orow + 32 + 8 * g + 4 * h, make_float4(acc1[q][4 * g] * inv, acc1[q][4 * g + 1] * inv, acc1[q][4 * g + 2] * inv, acc1[q][4 * g + 3] * inv));
            }
            if (h == 0) lse[qrow[q]] = (m[q] + log2f(ls)) * LN2;
        }
    }
}

#elif ATT_VAR >= 92 && ATT_VAR < 100
// v92: v91 (12-wave workgroup per CU, reference-exponent softmax, Q/O through LDS) with a software-pipelined, fenced loop body:
// iteration i carries the exp / sum / pack arithmetic of key sub-tile i in eight groups of 7 VALU instructions, each behind one of
// eight MFMAs that do not depend on it (PV of sub-tile i-1 alternating with QK^T of sub-tile i+1; no MFMA directly follows one it
// depends on).  Measured (scripts/exp/overlap.hip, 3 waves per SIMD): this order costs 216 ns per wave-iteration against 283 for MFMA
// and VALU phases issued apart.  3-slot LDS ring of 128-key K/V tiles, one barrier per tile.
#ifndef NW30
#define NW30 12
#endif
#ifndef KB30
#define KB30 96
#endif
constexpr int NT30 = NW30 * 64, NC30 = KB30 * 8, NF30 = (2 * NC30 + NT30 - 1) / NT30, SLOT30 = 2 * KB30 * LDR, SPT30 = KB30 / 32;
#undef FENCE
#ifdef NOFENCE
#define FENCE()
#else
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
__global__ __launch_bounds__(NT30, 1) void attn_fwd_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                               const bf16_t* __restrict__ V, int ldv, long long skv, bf16_t* __restrict__ O,
                                                               int ldo, float* __restrict__ lse, Segs sg, int Nk, float scale) {
    extern __shared__ __attribute__((aligned(16))) bf16_t smem30[];      // [3 slots][K tile | V tile][KB30][LDR], then NW30 wave tiles [32][LDR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int nwt = sg.t32[sg.n], bpi = (nwt + NW30 - 1) / NW30;
    const int b = blockIdx.x / bpi, wt = (blockIdx.x - b * bpi) * NW30 + wave;
    int sgi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < sg.n && wt >= sg.t32[i]) sgi = i;
    const int nq = sg.nq[sgi];
    const int tq0 = (wt - sg.t32[sgi]) * 32;                    // first query of the wave's tile within its segment-image
    const bool ok = wt < nwt && tq0 + j < nq;
    const long long trow0 = (long long)sg.row0[sgi] + (long long)b * nq + tq0;
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    bf16_t* wtile = smem30 + 3 * SLOT30 + wave * (32 * LDR);
    uint4 st[NF30];
    auto fetch = [&](int kb0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NF30; ++i) {
            const int id = tid + i * NT30, isv = id >= NC30, r = (id - isv * NC30) >> 3, c8 = (id & 7) * 8;
            const bf16_t* src = isv ? Vb : Kb;
            const int ld = isv ? ldv : ldk;
            st[i] = id < 2 * NC30 ? *reinterpret_cast<const uint4*>(src + (long long)min(kb0 + r, Nk - 1) * ld + c8) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto stash = [&](int slot) __attribute__((always_inline)) {
        bf16_t* base = smem30 + slot * SLOT30;
#pragma unroll
        for (int i = 0; i < NF30; ++i) {
            const int id = tid + i * NT30, isv = id >= NC30, r = (id - isv * NC30) >> 3, c8 = (id & 7) * 8;
            const int pr = isv ? ((r & ~15) | ((r & 3) << 2) | ((r >> 2) & 3)) : r;
            if (id < 2 * NC30) *reinterpret_cast<uint4*>(base + isv * (KB30 * LDR) + pr * LDR + c8) = st[i];
        }
    };
    fetch(0);                                                   // the first K/V tile is requested before the Q tile: every wave waits for it
    bf16x8 qf[4];
    {   // Q tile of the wave: coalesced 128-byte rows into the wave-private LDS tile, fragments read back from there
        uint4 qv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 8 * i + (lane >> 3);
            qv[i] = (wt < nwt && tq0 + r < nq) ? *reinterpret_cast<const uint4*>(Q + (trow0 + r) * ldq + 8 * (lane & 7)) : make_uint4(0u, 0u, 0u, 0u);
        }
        stash(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(wtile + (8 * i + (lane >> 3)) * LDR + 8 * (lane & 7)) = qv[i];
    }
    const int nt = (Nk + KB30 - 1) / KB30, nsubs = (Nk + 31) / 32;
    if (nt > 1) fetch(KB30);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = ld_frag(wtile + j * LDR + 16 * ks + 8 * h);
    const float qs = scale * LOG2E;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float m, lsum = 0.f;
    const int koff = pi_row(j) * LDR + 8 * h;
    const int voff = (16 * h + 4 * ((lane & 15) >> 2)) * LDR + 16 * ((lane >> 4) & 1) + 4 * (lane & 3) + KB30 * LDR;
    auto kaddr = [&](int i) __attribute__((always_inline)) { return smem30 + ((i / SPT30) % 3) * SLOT30 + 32 * (i % SPT30) * LDR + koff; };
    auto vaddr = [&](int i) __attribute__((always_inline)) { return smem30 + ((i / SPT30) % 3) * SLOT30 + 32 * (i % SPT30) * LDR + voff; };
    bf16x8 kf[4], vf[4], pb0, pb1;
    auto load_k = [&](int i) __attribute__((always_inline)) {
        const bf16_t* kp = kaddr(i);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[ks] = ld_frag(kp + 16 * ks);
    };
    auto load_v = [&](int i) __attribute__((always_inline)) {
        const bf16_t* vp = vaddr(i);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            vf[2 * k2] = ld_frag_tr(vp + (2 * k2) * LDR, vp + (2 * k2 + 1) * LDR);
            vf[2 * k2 + 1] = ld_frag_tr(vp + (2 * k2) * LDR + 32, vp + (2 * k2 + 1) * LDR + 32);
        }
    };
    auto qk = [&](f32x16& S) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], S, 0, 0, 0);
    };
    auto rowmax = [&](const f32x16& S) __attribute__((always_inline)) {
        float mx = S[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, S[r]);
        const unsigned u = __float_as_uint(mx);
        const auto pr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        return ceilf(fmaxf(__uint_as_float(pr[0]), __uint_as_float(pr[1])) * qs);
    };
    f32x16 SA, SB;
    load_k(0);
    qk(SA);
    m = rowmax(SA);                                             // reference exponent: integer-valued, raised only when a row outgrows it by 2^30
    load_v(0);                                                  // (placeholder fragments for the empty PV of the first iteration)
    if (nsubs > 1) load_k(1);
#pragma unroll
    for (int i = 0; i < 8; ++i) { pb0[i] = (__bf16)0.f; pb1[i] = (__bf16)0.f; }
    // iteration i: S = scores of sub-tile i; kf = K fragments of sub-tile i + 1; vf, pb = V^T fragments and packed P of sub-tile i - 1
    auto step = [&](f32x16& S, f32x16& Sn, int i, auto last_tag) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_tag)::value;
        const int t = i / SPT30, sub = i % SPT30;
        if (sub == 0) {
            if (t + 1 < nt) stash((t + 1) % 3);
            if (t + 2 < nt) fetch((t + 2) * KB30);
        }
        if (LAST) {
            const int kv0 = 32 * i;
            if (kv0 + 32 > Nk) {
#pragma unroll
                for (int r = 0; r < 16; ++r) if (kv0 + 16 * h + r >= Nk) S[r] = NEG_BIG;
            }
        }
        float rs = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            FENCE();
            if (g & 1) {                                        // QK^T of sub-tile i + 1, k-slice g >> 1
                if (!LAST) {
                    if (g == 1) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) Sn[r] = 0.f;
                    }
                    Sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[g >> 1], qf[g >> 1], Sn, 0, 0, 0);
                }
            } else {                                            // PV of sub-tile i - 1
                const int c = g >> 1;
                if (c & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[c], c < 2 ? pb0 : pb1, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[c], c < 2 ? pb0 : pb1, acc0, 0, 0, 0);
            }
            FENCE();
#pragma unroll
            for (int r = 2 * g; r < 2 * g + 2; ++r) { S[r] = fast_exp2(fmaf(S[r], qs, -m)); rs += S[r]; }
            if (g == 3) pb0 = pack8(S, 0);                      // (the PV MFMAs that read the previous pb0 / pb1 were issued at g = 0, 2 / 4, 6)
            if (g == 7) pb1 = pack8(S, 8);
        }
        FENCE();
#ifndef NOSLOW
        if (__any(!(rs < 1073741824.0f))) {                     // rare: re-reference the rows (scores recomputed from the resident K tile)
            f32x16 T;
            const bf16_t* kp = kaddr(i);
#pragma unroll
            for (int r = 0; r < 16; ++r) T[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) T = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(kp + 16 * ks), qf[ks], T, 0, 0, 0);
            if (LAST) {
                const int kv0 = 32 * i;
#pragma unroll
                for (int r = 0; r < 16; ++r) if (kv0 + 16 * h + r >= Nk) T[r] = NEG_BIG;
            }
            const float mn = fmaxf(m, rowmax(T));
            const float alpha = fast_exp2(m - mn);
            lsum *= alpha;
            m = mn;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] *= alpha; acc1[r] *= alpha; }
            rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { S[r] = fast_exp2(fmaf(T[r], qs, -m)); rs += S[r]; }
            pb0 = pack8(S, 0);
            pb1 = pack8(S, 8);
        }
#endif
        lsum += rs;
        // fragments for the next iteration: V^T of this sub-tile, K of sub-tile i + 2 (its tile was stored >= 2 iterations ago)
        load_v(i);
        if (sub == 1 && i + 2 < nsubs) __syncthreads();         // tile t + 1 (stored at sub 0) is complete: sub-tile i + 3 is its first
        if (i + 2 < nsubs) load_k(i + 2);
    };
    int i = 0;
#pragma unroll 1
    for (; i + 2 < nsubs; i += 2) { step(SA, SB, i, std::false_type{}); step(SB, SA, i + 1, std::false_type{}); }
    if (i + 2 == nsubs) { step(SA, SB, i, std::false_type{}); step(SB, SA, i + 1, std::true_type{}); }
    else step(SA, SB, i, std::true_type{});
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0], pb0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1], pb0, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[2], pb1, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[3], pb1, acc1, 0, 0, 0);
    lsum += __shfl_xor(lsum, 32, 64);
    {   // O tile through the wave-private LDS tile: whole 128-byte rows leave with 16-byte stores
        const float inv = 1.0f / lsum;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<bf16_t>(wtile + j * LDR + 8 * g + 4 * h, make_float4(acc0[4 * g] * inv, acc0[4 * g + 1] * inv, acc0[4 * g + 2] * inv, acc0[4 * g + 3] * inv));
            st4<bf16_t>(wtile + j * LDR + 32 + 8 * g + 4 * h, make_float4(acc1[4 * g] * inv, acc1[4 * g + 1] * inv, acc1[4 * g + 2] * inv, acc1[4 * g + 3] * inv));
        }
        if (ok && h == 0) lse[trow0 + j] = (m + log2f(lsum)) * LN2;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
            const int r = 8 * i2 + (lane >> 3);
            const uint4 v = *reinterpret_cast<const uint4*>(wtile + r * LDR + 8 * (lane & 7));
            if (wt < nwt && tq0 + r < nq) *reinterpret_cast<uint4*>(O + (trow0 + r) * ldo + 8 * (lane & 7)) = v;
        }
    }
}

#elif ATT_VAR == 27
// v27 = v16 with the 4-long dependent QK^T MFMA chain split in two independent halves (3 waves per SIMD budget)
// v16 = v15 with wave tiles flattened per image: 48 workgroups per image = 768 = 3 per CU exactly
// v15 = v13 + permlane32 swap for the cross-half max, per-half row sums
// v13 = the production kernel with V kept row-major (4x4 k-row permutation), gathered by ds_read_b64_tr_b16, its strips loaded together
__global__ __launch_bounds__(256, 3) void attn_fwd_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                              const bf16_t* __restrict__ V, int ldv, long long skv, bf16_t* __restrict__ O,
                                                              int ldo, float* __restrict__ lse, Segs sg, int Nk, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[KB * LDR];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[KB * LDR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int nwt = sg.t32[sg.n], bpi = (nwt + 3) >> 2;
    const int b = blockIdx.x / bpi, wt = (blockIdx.x - b * bpi) * 4 + wave;
    int sgi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < sg.n && wt >= sg.t32[i]) sgi = i;
    const int nq = sg.nq[sgi];
    const int ql = (wt - sg.t32[sgi]) * 32 + j;
    const bool ok = wt < nwt && ql < nq;
    const long long qrow = (long long)sg.row0[sgi] + (long long)b * nq + ql;
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint4 v = ok ? *reinterpret_cast<const uint4*>(Q + qrow * ldq + 16 * ks + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
        qf[ks] = *reinterpret_cast<const bf16x8*>(&v);
    }
    const float qs = scale * LOG2E;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float m = NEG_BIG, lsum = 0.f;
    const int krow = pi_row(j);
    uint4 kr[4];                                            // K of the next fill is prefetched; V is fetched at its LDS store (keeps the
    auto fetch = [&](int kb0) {                            // kernel within 128 VGPRs: 4 workgroups per CU = all 800 tiles resident at once)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r, c8, g;
            fill_map(tid, i, r, c8, g);
            kr[i] = ld_row8(Kb, ldk, kb0 + r, Nk, c8);
        }
    };
    fetch(0);
    for (int kb0 = 0; kb0 < Nk; kb0 += KB) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r, c8, g;
            fill_map(tid, i, r, c8, g);
            *reinterpret_cast<uint4*>(&Ks[r * LDR + c8]) = kr[i];
        }
        {
            uint4 vr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r, c8, g;
                fill_map(tid, i, r, c8, g);
                vr[i] = *reinterpret_cast<const uint4*>(Vb + (long long)min(kb0 + r, Nk - 1) * ldv + c8);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r, c8, g;
                fill_map(tid, i, r, c8, g);
                const int pr = (r & ~15) | ((r & 3) << 2) | ((r >> 2) & 3);
                *reinterpret_cast<uint4*>(&Vs[pr * LDR + c8]) = vr[i];
            }
        }
        __syncthreads();
        if (kb0 + KB < Nk) fetch(kb0 + KB);
        // S^T tile of sub-tile `sub`: 4 chained MFMAs, issued asynchronously to the matrix pipe
        auto qk = [&](int sub) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const bf16_t* kp = Ks + (32 * sub + krow) * LDR + 8 * h;
            f32x16 s2;
#pragma unroll
            for (int r = 0; r < 16; ++r) s2[r] = 0.f;
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(kp), qf[0], s, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(kp + 32), qf[2], s2, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(kp + 16), qf[1], s, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(kp + 48), qf[3], s2, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] += s2[r];
            return s;
        };
        // online softmax of a finished S^T tile (register VALU) followed by O^T += V^T P^T
        auto softmax_pv = [&](f32x16 s, int sub) {
            const int kv0 = kb0 + 32 * sub;
            if (kv0 + 32 > Nk) {                                // tail tile (wave-uniform)
#pragma unroll
                for (int r = 0; r < 16; ++r) if (kv0 + 16 * h + r >= Nk) s[r] = NEG_BIG;
            }
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            {
                const unsigned u = __float_as_uint(mx);
                const auto pr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                mx = fmaxf(__uint_as_float(pr[0]), __uint_as_float(pr[1])) * qs;
            }
            if (__any(mx > m + RESCALE_THR)) {                  // lazy rescale: rare once the running maximum has settled
                const float mn = fmaxf(m, mx);
                const float alpha = fast_exp2(m - mn);
                lsum *= alpha;
                m = mn;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] *= alpha; acc1[r] *= alpha; }
            }
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = fast_exp2(fmaf(s[r], qs, -m)); rs += s[r]; }
            lsum += rs;
            const int gi = lane & 15, gq = (lane >> 4) & 1;
            const bf16_t* vp = Vs + (32 * sub + 16 * h + 4 * (gi >> 2)) * LDR + 16 * gq + 4 * (gi & 3);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const bf16x8 pb = pack8(s, 8 * k2);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + (2 * k2) * LDR, vp + (2 * k2 + 1) * LDR), pb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + (2 * k2) * LDR + 32, vp + (2 * k2 + 1) * LDR + 32), pb, acc1, 0, 0, 0);
            }
        };
#pragma unroll 1
        for (int sub = 0; sub < KB / 32 && kb0 + 32 * sub < Nk; ++sub) softmax_pv(qk(sub), sub);
    }
    lsum += __shfl_xor(lsum, 32, 64);
    if (ok) {
        const float inv = 1.0f / lsum;
        bf16_t* orow = O + qrow * ldo;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<bf16_t>(orow + 8 * g + 4 * h, make_float4(acc0[4 * g] * inv, acc0[4 * g + 1] * inv, acc0[4 * g + 2] * inv, acc0[4 * g + 3] * inv));
            st4<bf16_t>(orow + 32 + 8 * g + 4 * h, make_float4(acc1[4 * g] * inv, acc1[4 * g + 1] * inv, acc1[4 * g + 2] * inv, acc1[4 * g + 3] * inv));
        }
        if (h == 0) lse[qrow] = (m + log2f(lsum)) * LN2;
    }
}

#elif ATT_VAR == 25
// v25 = ablation of v16: no softmax arithmetic (P = raw scores)
// v16 = v15 with wave tiles flattened per image: 48 workgroups per image = 768 = 3 per CU exactly
// v15 = v13 + permlane32 swap for the cross-half max, per-half row sums
// v13 = the production kernel with V kept row-major (4x4 k-row permutation), gathered by ds_read_b64_tr_b16, its strips loaded together
__global__ __launch_bounds__(256, 4) void attn_fwd_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                              const bf16_t* __restrict__ V, int ldv, long long skv, bf16_t* __restrict__ O,
                                                              int ldo, float* __restrict__ lse, Segs sg, int Nk, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[KB * LDR];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[KB * LDR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int nwt = sg.t32[sg.n], bpi = (nwt + 3) >> 2;
    const int b = blockIdx.x / bpi, wt = (blockIdx.x - b * bpi) * 4 + wave;
    int sgi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < sg.n && wt >= sg.t32[i]) sgi = i;
    const int nq = sg.nq[sgi];
    const int ql = (wt - sg.t32[sgi]) * 32 + j;
    const bool ok = wt < nwt && ql < nq;
    const long long qrow = (long long)sg.row0[sgi] + (long long)b * nq + ql;
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint4 v = ok ? *reinterpret_cast<const uint4*>(Q + qrow * ldq + 16 * ks + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
        qf[ks] = *reinterpret_cast<const bf16x8*>(&v);
    }
    const float qs = scale * LOG2E;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float m = NEG_BIG, lsum = 0.f;
    const int krow = pi_row(j);
    uint4 kr[4];                                            // K of the next fill is prefetched; V is fetched at its LDS store (keeps the
    auto fetch = [&](int kb0) {                            // kernel within 128 VGPRs: 4 workgroups per CU = all 800 tiles resident at once)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r, c8, g;
            fill_map(tid, i, r, c8, g);
            kr[i] = ld_row8(Kb, ldk, kb0 + r, Nk, c8);
        }
    };
    fetch(0);
    for (int kb0 = 0; kb0 < Nk; kb0 += KB) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r, c8, g;
            fill_map(tid, i, r, c8, g);
            *reinterpret_cast<uint4*>(&Ks[r * LDR + c8]) = kr[i];
        }
        {
            uint4 vr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r, c8, g;
                fill_map(tid, i, r, c8, g);
                vr[i] = *reinterpret_cast<const uint4*>(Vb + (long long)min(kb0 + r, Nk - 1) * ldv + c8);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r, c8, g;
                fill_map(tid, i, r, c8, g);
                const int pr = (r & ~15) | ((r & 3) << 2) | ((r >> 2) & 3);
                *reinterpret_cast<uint4*>(&Vs[pr * LDR + c8]) = vr[i];
            }
        }
        __syncthreads();
        if (kb0 + KB < Nk) fetch(kb0 + KB);
        // S^T tile of sub-tile `sub`: 4 chained MFMAs, issued asynchronously to the matrix pipe
        auto qk = [&](int sub) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const bf16_t* kp = Ks + (32 * sub + krow) * LDR + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(kp + 16 * ks), qf[ks], s, 0, 0, 0);
            return s;
        };
        // online softmax of a finished S^T tile (register VALU) followed by O^T += V^T P^T
        auto softmax_pv = [&](f32x16 s, int sub) {
            lsum = 1.f;
            const int gi = lane & 15, gq = (lane >> 4) & 1;
            const bf16_t* vp = Vs + (32 * sub + 16 * h + 4 * (gi >> 2)) * LDR + 16 * gq + 4 * (gi & 3);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const bf16x8 pb = pack8(s, 8 * k2);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + (2 * k2) * LDR, vp + (2 * k2 + 1) * LDR), pb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + (2 * k2) * LDR + 32, vp + (2 * k2 + 1) * LDR + 32), pb, acc1, 0, 0, 0);
            }
        };
#pragma unroll 1
        for (int sub = 0; sub < KB / 32 && kb0 + 32 * sub < Nk; ++sub) softmax_pv(qk(sub), sub);
    }
    lsum += __shfl_xor(lsum, 32, 64);
    if (ok) {
        const float inv = 1.0f / lsum;
        bf16_t* orow = O + qrow * ldo;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<bf16_t>(orow + 8 * g + 4 * h, make_float4(acc0[4 * g] * inv, acc0[4 * g + 1] * inv, acc0[4 * g + 2] * inv, acc0[4 * g + 3] * inv));
            st4<bf16_t>(orow + 32 + 8 * g + 4 * h, make_float4(acc1[4 * g] * inv, acc1[4 * g + 1] * inv, acc1[4 * g + 2] * inv, acc1[4 * g + 3] * inv));
        }
        if (h == 0) lse[qrow] = (m + log2f(lsum)) * LN2;
    }
}

#elif ATT_VAR == 26
// v26 = ablation of v16: K/V staged for the first fill only (no loads, LDS stores or barriers afterwards)
// v16 = v15 with wave tiles flattened per image: 48 workgroups per image = 768 = 3 per CU exactly
// v15 = v13 + permlane32 swap for the cross-half max, per-half row sums
// v13 = the production kernel with V kept row-major (4x4 k-row permutation), gathered by ds_read_b64_tr_b16, its strips loaded together
__global__ __launch_bounds__(256, 4) void attn_fwd_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                              const bf16_t* __restrict__ V, int ldv, long long skv, bf16_t* __restrict__ O,
                                                              int ldo, float* __restrict__ lse, Segs sg, int Nk, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[KB * LDR];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[KB * LDR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int nwt = sg.t32[sg.n], bpi = (nwt + 3) >> 2;
    const int b = blockIdx.x / bpi, wt = (blockIdx.x - b * bpi) * 4 + wave;
    int sgi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < sg.n && wt >= sg.t32[i]) sgi = i;
    const int nq = sg.nq[sgi];
    const int ql = (wt - sg.t32[sgi]) * 32 + j;
    const bool ok = wt < nwt && ql < nq;
    const long long qrow = (long long)sg.row0[sgi] + (long long)b * nq + ql;
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint4 v = ok ? *reinterpret_cast<const uint4*>(Q + qrow * ldq + 16 * ks + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
        qf[ks] = *reinterpret_cast<const bf16x8*>(&v);
    }
    const float qs = scale * LOG2E;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float m = NEG_BIG, lsum = 0.f;
    const int krow = pi_row(j);
    uint4 kr[4];                                            // K of the next fill is prefetched; V is fetched at its LDS store (keeps the
    auto fetch = [&](int kb0) {                            // kernel within 128 VGPRs: 4 workgroups per CU = all 800 tiles resident at once)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r, c8, g;
            fill_map(tid, i, r, c8, g);
            kr[i] = ld_row8(Kb, ldk, kb0 + r, Nk, c8);
        }
    };
    fetch(0);
    for (int kb0 = 0; kb0 < Nk; kb0 += KB) {
        if (kb0 == 0) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r, c8, g;
            fill_map(tid, i, r, c8, g);
            *reinterpret_cast<uint4*>(&Ks[r * LDR + c8]) = kr[i];
        }
        {
            uint4 vr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r, c8, g;
                fill_map(tid, i, r, c8, g);
                vr[i] = *reinterpret_cast<const uint4*>(Vb + (long long)min(kb0 + r, Nk - 1) * ldv + c8);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r, c8, g;
                fill_map(tid, i, r, c8, g);
                const int pr = (r & ~15) | ((r & 3) << 2) | ((r >> 2) & 3);
                *reinterpret_cast<uint4*>(&Vs[pr * LDR + c8]) = vr[i];
            }
        }
        __syncthreads();
        }
        // S^T tile of sub-tile `sub`: 4 chained MFMAs, issued asynchronously to the matrix pipe
        auto qk = [&](int sub) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const bf16_t* kp = Ks + (32 * sub + krow) * LDR + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(kp + 16 * ks), qf[ks], s, 0, 0, 0);
            return s;
        };
        // online softmax of a finished S^T tile (register VALU) followed by O^T += V^T P^T
        auto softmax_pv = [&](f32x16 s, int sub) {
            const int kv0 = kb0 + 32 * sub;
            if (kv0 + 32 > Nk) {                                // tail tile (wave-uniform)
#pragma unroll
                for (int r = 0; r < 16; ++r) if (kv0 + 16 * h + r >= Nk) s[r] = NEG_BIG;
            }
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            {
                const unsigned u = __float_as_uint(mx);
                const auto pr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                mx = fmaxf(__uint_as_float(pr[0]), __uint_as_float(pr[1])) * qs;
            }
            if (__any(mx > m + RESCALE_THR)) {                  // lazy rescale: rare once the running maximum has settled
                const float mn = fmaxf(m, mx);
                const float alpha = fast_exp2(m - mn);
                lsum *= alpha;
                m = mn;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] *= alpha; acc1[r] *= alpha; }
            }
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = fast_exp2(fmaf(s[r], qs, -m)); rs += s[r]; }
            lsum += rs;
            const int gi = lane & 15, gq = (lane >> 4) & 1;
            const bf16_t* vp = Vs + (32 * sub + 16 * h + 4 * (gi >> 2)) * LDR + 16 * gq + 4 * (gi & 3);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const bf16x8 pb = pack8(s, 8 * k2);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + (2 * k2) * LDR, vp + (2 * k2 + 1) * LDR), pb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + (2 * k2) * LDR + 32, vp + (2 * k2 + 1) * LDR + 32), pb, acc1, 0, 0, 0);
            }
        };
#pragma unroll 1
        for (int sub = 0; sub < KB / 32 && kb0 + 32 * sub < Nk; ++sub) softmax_pv(qk(sub), sub);
    }
    lsum += __shfl_xor(lsum, 32, 64);
    if (ok) {
        const float inv = 1.0f / lsum;
        bf16_t* orow = O + qrow * ldo;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<bf16_t>(orow + 8 * g + 4 * h, make_float4(acc0[4 * g] * inv, acc0[4 * g + 1] * inv, acc0[4 * g + 2] * inv, acc0[4 * g + 3] * inv));
            st4<bf16_t>(orow + 32 + 8 * g + 4 * h, make_float4(acc1[4 * g] * inv, acc1[4 * g + 1] * inv, acc1[4 * g + 2] * inv, acc1[4 * g + 3] * inv));
        }
        if (h == 0) lse[qrow] = (m + log2f(lsum)) * LN2;
    }
}

#elif ATT_VAR == 24
// v24 = v16 with MFMA phases at raised wave priority
// v16 = v15 with wave tiles flattened per image: 48 workgroups per image = 768 = 3 per CU exactly
// v15 = v13 + permlane32 swap for the cross-half max, per-half row sums
// v13 = the production kernel with V kept row-major (4x4 k-row permutation), gathered by ds_read_b64_tr_b16, its strips loaded together
__global__ __launch_bounds__(256, 4) void attn_fwd_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                              const bf16_t* __restrict__ V, int ldv, long long skv, bf16_t* __restrict__ O,
                                                              int ldo, float* __restrict__ lse, Segs sg, int Nk, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[KB * LDR];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[KB * LDR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int nwt = sg.t32[sg.n], bpi = (nwt + 3) >> 2;
    const int b = blockIdx.x / bpi, wt = (blockIdx.x - b * bpi) * 4 + wave;
    int sgi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < sg.n && wt >= sg.t32[i]) sgi = i;
    const int nq = sg.nq[sgi];
    const int ql = (wt - sg.t32[sgi]) * 32 + j;
    const bool ok = wt < nwt && ql < nq;
    const long long qrow = (long long)sg.row0[sgi] + (long long)b * nq + ql;
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint4 v = ok ? *reinterpret_cast<const uint4*>(Q + qrow * ldq + 16 * ks + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
        qf[ks] = *reinterpret_cast<const bf16x8*>(&v);
    }
    const float qs = scale * LOG2E;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float m = NEG_BIG, lsum = 0.f;
    const int krow = pi_row(j);
    uint4 kr[4];                                            // K of the next fill is prefetched; V is fetched at its LDS store (keeps the
    auto fetch = [&](int kb0) {                            // kernel within 128 VGPRs: 4 workgroups per CU = all 800 tiles resident at once)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r, c8, g;
            fill_map(tid, i, r, c8, g);
            kr[i] = ld_row8(Kb, ldk, kb0 + r, Nk, c8);
        }
    };
    fetch(0);
    for (int kb0 = 0; kb0 < Nk; kb0 += KB) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r, c8, g;
            fill_map(tid, i, r, c8, g);
            *reinterpret_cast<uint4*>(&Ks[r * LDR + c8]) = kr[i];
        }
        {
            uint4 vr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r, c8, g;
                fill_map(tid, i, r, c8, g);
                vr[i] = *reinterpret_cast<const uint4*>(Vb + (long long)min(kb0 + r, Nk - 1) * ldv + c8);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r, c8, g;
                fill_map(tid, i, r, c8, g);
                const int pr = (r & ~15) | ((r & 3) << 2) | ((r >> 2) & 3);
                *reinterpret_cast<uint4*>(&Vs[pr * LDR + c8]) = vr[i];
            }
        }
        __syncthreads();
        if (kb0 + KB < Nk) fetch(kb0 + KB);
        // S^T tile of sub-tile `sub`: 4 chained MFMAs, issued asynchronously to the matrix pipe
        auto qk = [&](int sub) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            __builtin_amdgcn_s_setprio(2);
            const bf16_t* kp = Ks + (32 * sub + krow) * LDR + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(kp + 16 * ks), qf[ks], s, 0, 0, 0);
            return s;
        };
        // online softmax of a finished S^T tile (register VALU) followed by O^T += V^T P^T
        auto softmax_pv = [&](f32x16 s, int sub) {
            const int kv0 = kb0 + 32 * sub;
            if (kv0 + 32 > Nk) {                                // tail tile (wave-uniform)
#pragma unroll
                for (int r = 0; r < 16; ++r) if (kv0 + 16 * h + r >= Nk) s[r] = NEG_BIG;
            }
            __builtin_amdgcn_s_setprio(0);
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            {
                const unsigned u = __float_as_uint(mx);
                const auto pr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                mx = fmaxf(__uint_as_float(pr[0]), __uint_as_float(pr[1])) * qs;
            }
            if (__any(mx > m + RESCALE_THR)) {                  // lazy rescale: rare once the running maximum has settled
                const float mn = fmaxf(m, mx);
                const float alpha = fast_exp2(m - mn);
                lsum *= alpha;
                m = mn;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] *= alpha; acc1[r] *= alpha; }
            }
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = fast_exp2(fmaf(s[r], qs, -m)); rs += s[r]; }
            lsum += rs;
            __builtin_amdgcn_s_setprio(2);
            const int gi = lane & 15, gq = (lane >> 4) & 1;
            const bf16_t* vp = Vs + (32 * sub + 16 * h + 4 * (gi >> 2)) * LDR + 16 * gq + 4 * (gi & 3);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const bf16x8 pb = pack8(s, 8 * k2);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + (2 * k2) * LDR, vp + (2 * k2 + 1) * LDR), pb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + (2 * k2) * LDR + 32, vp + (2 * k2 + 1) * LDR + 32), pb, acc1, 0, 0, 0);
            }
        };
#pragma unroll 1
        for (int sub = 0; sub < KB / 32 && kb0 + 32 * sub < Nk; ++sub) softmax_pv(qk(sub), sub);
    }
    lsum += __shfl_xor(lsum, 32, 64);
    if (ok) {
        const float inv = 1.0f / lsum;
        bf16_t* orow = O + qrow * ldo;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<bf16_t>(orow + 8 * g + 4 * h, make_float4(acc0[4 * g] * inv, acc0[4 * g + 1] * inv, acc0[4 * g + 2] * inv, acc0[4 * g + 3] * inv));
            st4<bf16_t>(orow + 32 + 8 * g + 4 * h, make_float4(acc1[4 * g] * inv, acc1[4 * g + 1] * inv, acc1[4 * g + 2] * inv, acc1[4 * g + 3] * inv));
        }
        if (h == 0) lse[qrow] = (m + log2f(lsum)) * LN2;
    }
}

#elif ATT_VAR == 23
// v23 = v16 with the QK^T MFMAs of sub-tile i+1 issued before the softmax of sub-tile i (3 waves per SIMD: 170 VGPRs)
// v16 = v15 with wave tiles flattened per image: 48 workgroups per image = 768 = 3 per CU exactly
// v15 = v13 + permlane32 swap for the cross-half max, per-half row sums
// v13 = the production kernel with V kept row-major (4x4 k-row permutation), gathered by ds_read_b64_tr_b16, its strips loaded together
__global__ __launch_bounds__(256, 3) void attn_fwd_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                              const bf16_t* __restrict__ V, int ldv, long long skv, bf16_t* __restrict__ O,
                                                              int ldo, float* __restrict__ lse, Segs sg, int Nk, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[KB * LDR];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[KB * LDR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int nwt = sg.t32[sg.n], bpi = (nwt + 3) >> 2;
    const int b = blockIdx.x / bpi, wt = (blockIdx.x - b * bpi) * 4 + wave;
    int sgi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < sg.n && wt >= sg.t32[i]) sgi = i;
    const int nq = sg.nq[sgi];
    const int ql = (wt - sg.t32[sgi]) * 32 + j;
    const bool ok = wt < nwt && ql < nq;
    const long long qrow = (long long)sg.row0[sgi] + (long long)b * nq + ql;
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint4 v = ok ? *reinterpret_cast<const uint4*>(Q + qrow * ldq + 16 * ks + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
        qf[ks] = *reinterpret_cast<const bf16x8*>(&v);
    }
    const float qs = scale * LOG2E;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float m = NEG_BIG, lsum = 0.f;
    const int krow = pi_row(j);
    uint4 kr[4];                                            // K of the next fill is prefetched; V is fetched at its LDS store (keeps the
    auto fetch = [&](int kb0) {                            // kernel within 128 VGPRs: 4 workgroups per CU = all 800 tiles resident at once)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r, c8, g;
            fill_map(tid, i, r, c8, g);
            kr[i] = ld_row8(Kb, ldk, kb0 + r, Nk, c8);
        }
    };
    fetch(0);
    for (int kb0 = 0; kb0 < Nk; kb0 += KB) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r, c8, g;
            fill_map(tid, i, r, c8, g);
            *reinterpret_cast<uint4*>(&Ks[r * LDR + c8]) = kr[i];
        }
        {
            uint4 vr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r, c8, g;
                fill_map(tid, i, r, c8, g);
                vr[i] = *reinterpret_cast<const uint4*>(Vb + (long long)min(kb0 + r, Nk - 1) * ldv + c8);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r, c8, g;
                fill_map(tid, i, r, c8, g);
                const int pr = (r & ~15) | ((r & 3) << 2) | ((r >> 2) & 3);
                *reinterpret_cast<uint4*>(&Vs[pr * LDR + c8]) = vr[i];
            }
        }
        __syncthreads();
        if (kb0 + KB < Nk) fetch(kb0 + KB);
        // S^T tile of sub-tile `sub`: 4 chained MFMAs, issued asynchronously to the matrix pipe
        auto qk = [&](int sub) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const bf16_t* kp = Ks + (32 * sub + krow) * LDR + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(kp + 16 * ks), qf[ks], s, 0, 0, 0);
            return s;
        };
        // online softmax of a finished S^T tile (register VALU) followed by O^T += V^T P^T
        auto softmax_pv = [&](f32x16 s, int sub) {
            const int kv0 = kb0 + 32 * sub;
            if (kv0 + 32 > Nk) {                                // tail tile (wave-uniform)
#pragma unroll
                for (int r = 0; r < 16; ++r) if (kv0 + 16 * h + r >= Nk) s[r] = NEG_BIG;
            }
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            {
                const unsigned u = __float_as_uint(mx);
                const auto pr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                mx = fmaxf(__uint_as_float(pr[0]), __uint_as_float(pr[1])) * qs;
            }
            if (__any(mx > m + RESCALE_THR)) {                  // lazy rescale: rare once the running maximum has settled
                const float mn = fmaxf(m, mx);
                const float alpha = fast_exp2(m - mn);
                lsum *= alpha;
                m = mn;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] *= alpha; acc1[r] *= alpha; }
            }
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = fast_exp2(fmaf(s[r], qs, -m)); rs += s[r]; }
            lsum += rs;
            const int gi = lane & 15, gq = (lane >> 4) & 1;
            const bf16_t* vp = Vs + (32 * sub + 16 * h + 4 * (gi >> 2)) * LDR + 16 * gq + 4 * (gi & 3);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const bf16x8 pb = pack8(s, 8 * k2);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + (2 * k2) * LDR, vp + (2 * k2 + 1) * LDR), pb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + (2 * k2) * LDR + 32, vp + (2 * k2 + 1) * LDR + 32), pb, acc1, 0, 0, 0);
            }
        };
        f32x16 scur = qk(0);
#pragma unroll 1
        for (int sub = 0; sub < KB / 32 && kb0 + 32 * sub < Nk; ++sub) {
            f32x16 snext = scur;
            if (sub + 1 < KB / 32 && kb0 + 32 * (sub + 1) < Nk) snext = qk(sub + 1);      // in flight under this tile's softmax
            softmax_pv(scur, sub);
            scur = snext;
        }
    }
    lsum += __shfl_xor(lsum, 32, 64);
    if (ok) {
        const float inv = 1.0f / lsum;
        bf16_t* orow = O + qrow * ldo;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<bf16_t>(orow + 8 * g + 4 * h, make_float4(acc0[4 * g] * inv, acc0[4 * g + 1] * inv, acc0[4 * g + 2] * inv, acc0[4 * g + 3] * inv));
            st4<bf16_t>(orow + 32 + 8 * g + 4 * h, make_float4(acc1[4 * g] * inv, acc1[4 * g + 1] * inv, acc1[4 * g + 2] * inv, acc1[4 * g + 3] * inv));
        }
        if (h == 0) lse[qrow] = (m + log2f(lsum)) * LN2;
    }
}

#elif ATT_VAR == 21
// v21 = v16 with 3 waves per SIMD allowed (170 VGPRs) and V prefetched a fill ahead together with K, all loads branch-free
// v16 = v15 with wave tiles flattened per image: 48 workgroups per image = 768 = 3 per CU exactly
// v15 = v13 + permlane32 swap for the cross-half max, per-half row sums
// v13 = the production kernel with V kept row-major (4x4 k-row permutation), gathered by ds_read_b64_tr_b16, its strips loaded together
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void attn_fwd_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                              const bf16_t* __restrict__ V, int ldv, long long skv, bf16_t* __restrict__ O,
                                                              int ldo, float* __restrict__ lse, Segs sg, int Nk, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[KB * LDR];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[KB * LDR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int nwt = sg.t32[sg.n], bpi = (nwt + 3) >> 2;
    const int b = blockIdx.x / bpi, wt = (blockIdx.x - b * bpi) * 4 + wave;
    int sgi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < sg.n && wt >= sg.t32[i]) sgi = i;
    const int nq = sg.nq[sgi];
    const int ql = (wt - sg.t32[sgi]) * 32 + j;
    const bool ok = wt < nwt && ql < nq;
    const long long qrow = (long long)sg.row0[sgi] + (long long)b * nq + ql;
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint4 v = ok ? *reinterpret_cast<const uint4*>(Q + qrow * ldq + 16 * ks + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
        qf[ks] = *reinterpret_cast<const bf16x8*>(&v);
    }
    const float qs = scale * LOG2E;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float m = NEG_BIG, lsum = 0.f;
    const int krow = pi_row(j);
    uint4 kr[4], vr[4];                                            // K of the next fill is prefetched; V is fetched at its LDS store (keeps the
    auto fetch = [&](int kb0) {                            // kernel within 128 VGPRs: 4 workgroups per CU = all 800 tiles resident at once)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r, c8, g;
            fill_map(tid, i, r, c8, g);
            const long long row = min(kb0 + r, Nk - 1);
            kr[i] = *reinterpret_cast<const uint4*>(Kb + row * ldk + c8);
            vr[i] = *reinterpret_cast<const uint4*>(Vb + row * ldv + c8);
        }
    };
    fetch(0);
    for (int kb0 = 0; kb0 < Nk; kb0 += KB) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r, c8, g;
            fill_map(tid, i, r, c8, g);
            *reinterpret_cast<uint4*>(&Ks[r * LDR + c8]) = kr[i];
        }
        {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r, c8, g;
                fill_map(tid, i, r, c8, g);
                const int pr = (r & ~15) | ((r & 3) << 2) | ((r >> 2) & 3);
                *reinterpret_cast<uint4*>(&Vs[pr * LDR + c8]) = vr[i];
            }
        }
        __syncthreads();
        fetch(min(kb0 + KB, Nk - 1));                      // unconditional: the last one is never used
        // S^T tile of sub-tile `sub`: 4 chained MFMAs, issued asynchronously to the matrix pipe
        auto qk = [&](int sub) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const bf16_t* kp = Ks + (32 * sub + krow) * LDR + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(kp + 16 * ks), qf[ks], s, 0, 0, 0);
            return s;
        };
        // online softmax of a finished S^T tile (register VALU) followed by O^T += V^T P^T
        auto softmax_pv = [&](f32x16 s, int sub) {
            const int kv0 = kb0 + 32 * sub;
            if (kv0 + 32 > Nk) {                                // tail tile (wave-uniform)
#pragma unroll
                for (int r = 0; r < 16; ++r) if (kv0 + 16 * h + r >= Nk) s[r] = NEG_BIG;
            }
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            {
                const unsigned u = __float_as_uint(mx);
                const auto pr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                mx = fmaxf(__uint_as_float(pr[0]), __uint_as_float(pr[1])) * qs;
            }
            if (__any(mx > m + RESCALE_THR)) {                  // lazy rescale: rare once the running maximum has settled
                const float mn = fmaxf(m, mx);
                const float alpha = fast_exp2(m - mn);
                lsum *= alpha;
                m = mn;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] *= alpha; acc1[r] *= alpha; }
            }
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = fast_exp2(fmaf(s[r], qs, -m)); rs += s[r]; }
            lsum += rs;
            const int gi = lane & 15, gq = (lane >> 4) & 1;
            const bf16_t* vp = Vs + (32 * sub + 16 * h + 4 * (gi >> 2)) * LDR + 16 * gq + 4 * (gi & 3);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const bf16x8 pb = pack8(s, 8 * k2);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + (2 * k2) * LDR, vp + (2 * k2 + 1) * LDR), pb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + (2 * k2) * LDR + 32, vp + (2 * k2 + 1) * LDR + 32), pb, acc1, 0, 0, 0);
            }
        };
#pragma unroll 1
        for (int sub = 0; sub < KB / 32 && kb0 + 32 * sub < Nk; ++sub) softmax_pv(qk(sub), sub);
    }
    lsum += __shfl_xor(lsum, 32, 64);
    if (ok) {
        const float inv = 1.0f / lsum;
        bf16_t* orow = O + qrow * ldo;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<bf16_t>(orow + 8 * g + 4 * h, make_float4(acc0[4 * g] * inv, acc0[4 * g + 1] * inv, acc0[4 * g + 2] * inv, acc0[4 * g + 3] * inv));
            st4<bf16_t>(orow + 32 + 8 * g + 4 * h, make_float4(acc1[4 * g] * inv, acc1[4 * g + 1] * inv, acc1[4 * g + 2] * inv, acc1[4 * g + 3] * inv));
        }
        if (h == 0) lse[qrow] = (m + log2f(lsum)) * LN2;
    }
}

#elif ATT_VAR == 17
// v17 = v16 with 160-key fills (5 instead of 7 at Nk = 784)
#ifndef KBF
#define KBF 160
#endif
// v16 = v15 with wave tiles flattened per image: 48 workgroups per image = 768 = 3 per CU exactly
// v15 = v13 + permlane32 swap for the cross-half max, per-half row sums
// v13 = the production kernel with V kept row-major (4x4 k-row permutation), gathered by ds_read_b64_tr_b16, its strips loaded together
__global__ __launch_bounds__(256, 4) void attn_fwd_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                              const bf16_t* __restrict__ V, int ldv, long long skv, bf16_t* __restrict__ O,
                                                              int ldo, float* __restrict__ lse, Segs sg, int Nk, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[KBF * LDR];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[KBF * LDR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int nwt = sg.t32[sg.n], bpi = (nwt + 3) >> 2;
    const int b = blockIdx.x / bpi, wt = (blockIdx.x - b * bpi) * 4 + wave;
    int sgi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < sg.n && wt >= sg.t32[i]) sgi = i;
    const int nq = sg.nq[sgi];
    const int ql = (wt - sg.t32[sgi]) * 32 + j;
    const bool ok = wt < nwt && ql < nq;
    const long long qrow = (long long)sg.row0[sgi] + (long long)b * nq + ql;
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint4 v = ok ? *reinterpret_cast<const uint4*>(Q + qrow * ldq + 16 * ks + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
        qf[ks] = *reinterpret_cast<const bf16x8*>(&v);
    }
    const float qs = scale * LOG2E;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float m = NEG_BIG, lsum = 0.f;
    const int krow = pi_row(j);
    constexpr int NS = KBF / 32;
    uint4 kr[NS];                                            // K of the next fill is prefetched; V is fetched at its LDS store (keeps the
    auto fetch = [&](int kb0) {                            // kernel within 128 VGPRs: 4 workgroups per CU = all 800 tiles resident at once)
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            int r, c8, g;
            { const int c = (tid >> 6) * NS + i; g = lane >> 4; r = 16 * (c >> 1) + (lane & 15); c8 = 32 * (c & 1) + 8 * g; }
            kr[i] = ld_row8(Kb, ldk, kb0 + r, Nk, c8);
        }
    };
    fetch(0);
    for (int kb0 = 0; kb0 < Nk; kb0 += KBF) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            int r, c8, g;
            { const int c = (tid >> 6) * NS + i; g = lane >> 4; r = 16 * (c >> 1) + (lane & 15); c8 = 32 * (c & 1) + 8 * g; }
            *reinterpret_cast<uint4*>(&Ks[r * LDR + c8]) = kr[i];
        }
        {
            uint4 vr[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                int r, c8, g;
                { const int c = (tid >> 6) * NS + i; g = lane >> 4; r = 16 * (c >> 1) + (lane & 15); c8 = 32 * (c & 1) + 8 * g; }
                vr[i] = *reinterpret_cast<const uint4*>(Vb + (long long)min(kb0 + r, Nk - 1) * ldv + c8);
            }
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                int r, c8, g;
                { const int c = (tid >> 6) * NS + i; g = lane >> 4; r = 16 * (c >> 1) + (lane & 15); c8 = 32 * (c & 1) + 8 * g; }
                const int pr = (r & ~15) | ((r & 3) << 2) | ((r >> 2) & 3);
                *reinterpret_cast<uint4*>(&Vs[pr * LDR + c8]) = vr[i];
            }
        }
        __syncthreads();
        if (kb0 + KBF < Nk) fetch(kb0 + KBF);
        // S^T tile of sub-tile `sub`: 4 chained MFMAs, issued asynchronously to the matrix pipe
        auto qk = [&](int sub) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const bf16_t* kp = Ks + (32 * sub + krow) * LDR + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(kp + 16 * ks), qf[ks], s, 0, 0, 0);
            return s;
        };
        // online softmax of a finished S^T tile (register VALU) followed by O^T += V^T P^T
        auto softmax_pv = [&](f32x16 s, int sub) {
            const int kv0 = kb0 + 32 * sub;
            if (kv0 + 32 > Nk) {                                // tail tile (wave-uniform)
#pragma unroll
                for (int r = 0; r < 16; ++r) if (kv0 + 16 * h + r >= Nk) s[r] = NEG_BIG;
            }
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            {
                const unsigned u = __float_as_uint(mx);
                const auto pr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                mx = fmaxf(__uint_as_float(pr[0]), __uint_as_float(pr[1])) * qs;
            }
            if (__any(mx > m + RESCALE_THR)) {                  // lazy rescale: rare once the running maximum has settled
                const float mn = fmaxf(m, mx);
                const float alpha = fast_exp2(m - mn);
                lsum *= alpha;
                m = mn;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] *= alpha; acc1[r] *= alpha; }
            }
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = fast_exp2(fmaf(s[r], qs, -m)); rs += s[r]; }
            lsum += rs;
            const int gi = lane & 15, gq = (lane >> 4) & 1;
            const bf16_t* vp = Vs + (32 * sub + 16 * h + 4 * (gi >> 2)) * LDR + 16 * gq + 4 * (gi & 3);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const bf16x8 pb = pack8(s, 8 * k2);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + (2 * k2) * LDR, vp + (2 * k2 + 1) * LDR), pb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + (2 * k2) * LDR + 32, vp + (2 * k2 + 1) * LDR + 32), pb, acc1, 0, 0, 0);
            }
        };
#pragma unroll 1
        for (int sub = 0; sub < KBF / 32 && kb0 + 32 * sub < Nk; ++sub) softmax_pv(qk(sub), sub);
    }
    lsum += __shfl_xor(lsum, 32, 64);
    if (ok) {
        const float inv = 1.0f / lsum;
        bf16_t* orow = O + qrow * ldo;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<bf16_t>(orow + 8 * g + 4 * h, make_float4(acc0[4 * g] * inv, acc0[4 * g + 1] * inv, acc0[4 * g + 2] * inv, acc0[4 * g + 3] * inv));
            st4<bf16_t>(orow + 32 + 8 * g + 4 * h, make_float4(acc1[4 * g] * inv, acc1[4 * g + 1] * inv, acc1[4 * g + 2] * inv, acc1[4 * g + 3] * inv));
        }
        if (h == 0) lse[qrow] = (m + log2f(lsum)) * LN2;
    }
}

#elif ATT_VAR == 10
// v10: 6 waves (192 queries) per workgroup over image-flattened 32-query wave tiles: 32 workgroups per image = 512 = 2 per CU
// exactly (12 waves per CU), K/V staging shared by 192 queries instead of 128, 96-key stages (3 sub-tiles of 32 keys).
constexpr int NW10 = 6, KB10 = 96, LDT10 = KB10 + 16;     // 224-byte Vt rows = 56 words (24 mod 32) -- see st_t8 note below
__global__ __launch_bounds__(NW10 * 64, 3) void attn_fwd_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                                   const bf16_t* __restrict__ V, int ldv, long long skv, bf16_t* __restrict__ O,
                                                                   int ldo, float* __restrict__ lse, Segs sg, int Nk, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[KB10 * LDR];
    __shared__ __attribute__((aligned(16))) bf16_t Vt[D * LDT10];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int nwt = sg.t32[sg.n], bpi = (nwt + NW10 - 1) / NW10;
    const int b = blockIdx.x / bpi, wt = (blockIdx.x - b * bpi) * NW10 + wave;
    int sgi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < sg.n && wt >= sg.t32[i]) sgi = i;
    const int nq = sg.nq[sgi];
    const int ql = (wt - sg.t32[sgi]) * 32 + j;
    const bool ok = wt < nwt && ql < nq;
    const long long qrow = (long long)sg.row0[sgi] + (long long)b * nq + ql;
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint4 v = ok ? *reinterpret_cast<const uint4*>(Q + qrow * ldq + 16 * ks + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
        qf[ks] = *reinterpret_cast<const bf16x8*>(&v);
    }
    const float qs = scale * LOG2E;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float m = NEG_BIG, lsum = 0.f;
    const int krow = pi_row(j);
    uint4 kr[2], vr[2];
    auto fmap = [&](int it, int& r, int& c8, int& g) {
        const int c = wave * 2 + it;                       // 12 chunks of 16 keys x 32 channels
        g = lane >> 4; r = 16 * (c >> 1) + (lane & 15); c8 = 32 * (c & 1) + 8 * g;
    };
    auto fetch = [&](int kb0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int r, c8, g; fmap(i, r, c8, g);
            const int row = min(kb0 + r, Nk - 1);
            kr[i] = *reinterpret_cast<const uint4*>(Kb + (long long)row * ldk + c8);
            vr[i] = *reinterpret_cast<const uint4*>(Vb + (long long)row * ldv + c8);
        }
    };
    fetch(0);
    for (int kb0 = 0; kb0 < Nk; kb0 += KB10) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int r, c8, g; fmap(i, r, c8, g);
            *reinterpret_cast<uint4*>(&Ks[r * LDR + c8]) = kr[i];
            st_t8(Vt, LDT10, c8, r, vr[i], g);
        }
        __syncthreads();
        if (kb0 + KB10 < Nk) fetch(kb0 + KB10);
#pragma unroll 1
        for (int sub = 0; sub < KB10 / 32 && kb0 + 32 * sub < Nk; ++sub) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const bf16_t* kp = Ks + (32 * sub + krow) * LDR + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(kp + 16 * ks), qf[ks], s, 0, 0, 0);
            const int kv0 = kb0 + 32 * sub;
            if (kv0 + 32 > Nk) {
#pragma unroll
                for (int r = 0; r < 16; ++r) if (kv0 + 16 * h + r >= Nk) s[r] = NEG_BIG;
            }
            float mx = max3f(s[0], s[1], s[2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) mx = max3f(mx, s[r], s[r + 1]);
            mx = fmaxf(mx, s[15]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * qs;
            if (__any(mx > m + RESCALE_THR)) {
                const float mn = fmaxf(m, mx);
                const float alpha = fast_exp2(m - mn);
                lsum *= alpha;
                m = mn;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] *= alpha; acc1[r] *= alpha; }
            }
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = fast_exp2(fmaf(s[r], qs, -m)); rs += s[r]; }
            lsum += rs;
            const bf16_t* vp = Vt + j * LDT10 + 32 * sub + 16 * h;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const bf16x8 pb = pack8(s, 8 * k2);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(vp + 8 * k2), pb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(vp + 32 * LDT10 + 8 * k2), pb, acc1, 0, 0, 0);
            }
        }
    }
    lsum += __shfl_xor(lsum, 32, 64);
    if (ok) {
        const float inv = 1.0f / lsum;
        bf16_t* orow = O + qrow * ldo;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<bf16_t>(orow + 8 * g + 4 * h, make_float4(acc0[4 * g] * inv, acc0[4 * g + 1] * inv, acc0[4 * g + 2] * inv, acc0[4 * g + 3] * inv));
            st4<bf16_t>(orow + 32 + 8 * g + 4 * h, make_float4(acc1[4 * g] * inv, acc1[4 * g + 1] * inv, acc1[4 * g + 2] * inv, acc1[4 * g + 3] * inv));
        }
        if (h == 0) lse[qrow] = (m + log2f(lsum)) * LN2;
    }
}
#elif ATT_VAR >= 5
// v5: wave tiles flattened per image (768 workgroups = 3 per CU exactly at the bench shape), 64-key double-buffered LDS stages
// (one barrier per stage, K and V both prefetched a full stage ahead), two independent S chains per iteration.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void attn_fwd_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                              const bf16_t* __restrict__ V, int ldv, long long skv, bf16_t* __restrict__ O,
                                                              int ldo, float* __restrict__ lse, Segs sg, int Nk, float scale) {
    constexpr int KS = 64, LDK = D + 8, LDV = KS + 16;
#if ATT_VAR >= 9
    constexpr int PV = D + 32;                           // 192-byte rows: the 4 rows of a tr-read block land on disjoint 16-bank windows
    __shared__ __attribute__((aligned(16))) bf16_t Ks[2][KS * LDK];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[2][KS * PV];
#else
    __shared__ __attribute__((aligned(16))) bf16_t Ks[2][KS * LDK];
    __shared__ __attribute__((aligned(16))) bf16_t Vt[2][D * LDV];
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int nwt = sg.t32[sg.n], bpi = (nwt + 3) >> 2;
    const int b = blockIdx.x / bpi, wt = (blockIdx.x - b * bpi) * 4 + wave;
    int sgi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < sg.n && wt >= sg.t32[i]) sgi = i;
    const int nq = sg.nq[sgi];
    const int ql = (wt - sg.t32[sgi]) * 32 + j;
    const bool ok = wt < nwt && ql < nq;
    const long long qrow = (long long)sg.row0[sgi] + (long long)b * nq + ql;
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint4 v = ok ? *reinterpret_cast<const uint4*>(Q + qrow * ldq + 16 * ks + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
        qf[ks] = *reinterpret_cast<const bf16x8*>(&v);
    }
    const float qs = scale * LOG2E;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float m = NEG_BIG, lsum = 0.f;
    const int krow = pi_row(j);
    uint4 kr[2], vr[2];
    auto fmap = [&](int it, int& r, int& c8, int& g) {
#if ATT_VAR >= 9
        const int idx = tid + 256 * it;                   // 8 lanes = one 128-byte row: coalesced loads, conflict-free b128 stores
        g = 0; r = idx >> 3; c8 = (idx & 7) * 8;
#else
        const int c = wave * 2 + it;
        g = lane >> 4; r = 16 * (c >> 1) + (lane & 15); c8 = 32 * (c & 1) + 8 * g;
#endif
    };
    auto fetch = [&](int st) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int r, c8, g; fmap(i, r, c8, g);
            const int row = min(st * KS + r, Nk - 1);       // rows past Nk: a finite duplicate, masked to P = 0 below (no branch, no wait)
            kr[i] = *reinterpret_cast<const uint4*>(Kb + (long long)row * ldk + c8);
            vr[i] = *reinterpret_cast<const uint4*>(Vb + (long long)row * ldv + c8);
        }
    };
    auto put = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int r, c8, g; fmap(i, r, c8, g);
            *reinterpret_cast<uint4*>(&Ks[buf][r * LDK + c8]) = kr[i];
#if ATT_VAR >= 9
            *reinterpret_cast<uint4*>(&Vs[buf][r * PV + c8]) = vr[i];
#else
            st_t8(Vt[buf], LDV, c8, r, vr[i], g);
#endif
        }
    };
    const int nst = (Nk + KS - 1) / KS;
    fetch(0); put(0);
    if (nst > 1) fetch(1);
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
#if ATT_VAR != 7 && ATT_VAR != 8
        if (st + 1 < nst) put(buf ^ 1);
        if (st + 2 < nst) fetch(st + 2);
#endif
        f32x16 sa, sb;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] = 0.f; sb[r] = 0.f; }
        {
            const bf16_t* kpa = Ks[buf] + krow * LDK + 8 * h;
            bf16x8 ka[4], kb[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { ka[ks] = ld_frag(kpa + 16 * ks); kb[ks] = ld_frag(kpa + 32 * LDK + 16 * ks); }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[ks], qf[ks], sa, 0, 0, 0);
                sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb[ks], qf[ks], sb, 0, 0, 0);
            }
        }
        const int kv0 = st * KS;
        if (kv0 + KS > Nk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (kv0 + 16 * h + r >= Nk) sa[r] = NEG_BIG;
                if (kv0 + 32 + 16 * h + r >= Nk) sb[r] = NEG_BIG;
            }
        }
#if ATT_VAR != 6 && ATT_VAR != 8
        float mx = max3f(sa[0], sb[0], sa[1]);
        mx = max3f(mx, sb[1], sa[2]);
#pragma unroll
        for (int r = 2; r < 15; ++r) mx = max3f(mx, sb[r], sa[r + 1]);
        mx = fmaxf(mx, sb[15]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * qs;
        if (__any(mx > m + RESCALE_THR)) {
            const float mn = fmaxf(m, mx);
            const float alpha = fast_exp2(m - mn);
            lsum *= alpha;
            m = mn;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] *= alpha; acc1[r] *= alpha; }
        }
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sa[r] = fast_exp2(fmaf(sa[r], qs, -m)); sb[r] = fast_exp2(fmaf(sb[r], qs, -m));
            rs += sa[r] + sb[r];
        }
        lsum += rs;                                        // the two halves are combined once, after the loop
#else
        lsum = 1.f;
#endif
#if ATT_VAR >= 9
        const bf16_t* vp = Vs[buf] + (16 * h + ((lane & 15) >> 2)) * PV + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {
            const bf16x8 pb = (k2 < 2) ? pack8(sa, 8 * k2) : pack8(sb, 8 * (k2 - 2));
            const int roff = (32 * (k2 >> 1) + 8 * (k2 & 1)) * PV;
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + roff, vp + roff + 4 * PV), pb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag_tr(vp + roff + 32, vp + roff + 4 * PV + 32), pb, acc1, 0, 0, 0);
        }
#else
        const bf16_t* vp = Vt[buf] + j * LDV + 16 * h;
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {
            const bf16x8 pb = (k2 < 2) ? pack8(sa, 8 * k2) : pack8(sb, 8 * (k2 - 2));
            const int off = 32 * (k2 >> 1) + 8 * (k2 & 1);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(vp + off), pb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(vp + 32 * LDV + off), pb, acc1, 0, 0, 0);
        }
#endif
#if ATT_VAR != 7 && ATT_VAR != 8
        __syncthreads();
#endif
    }
    lsum += __shfl_xor(lsum, 32, 64);
    if (ok) {
        const float inv = 1.0f / lsum;
        bf16_t* orow = O + qrow * ldo;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<bf16_t>(orow + 8 * g + 4 * h, make_float4(acc0[4 * g] * inv, acc0[4 * g + 1] * inv, acc0[4 * g + 2] * inv, acc0[4 * g + 3] * inv));
            st4<bf16_t>(orow + 32 + 8 * g + 4 * h, make_float4(acc1[4 * g] * inv, acc1[4 * g + 1] * inv, acc1[4 * g + 2] * inv, acc1[4 * g + 3] * inv));
        }
        if (h == 0) lse[qrow] = (m + log2f(lsum)) * LN2;
    }
}
#else
__global__ __launch_bounds__(256, 4) void attn_fwd_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                              const bf16_t* __restrict__ V, int ldv, long long skv, bf16_t* __restrict__ O,
                                                              int ldo, float* __restrict__ lse, Segs sg, int Nk, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[KB * LDR];
    __shared__ __attribute__((aligned(16))) bf16_t Vt[D * LDTB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    int row_base, q0, nq, b;
    locate_tile(sg, blockIdx.x, row_base, q0, nq, b);
    const int ql = q0 + wave * 32 + j;                     // query index inside this (segment, image)
    const bool ok = ql < nq;
    const long long qrow = (long long)row_base + ql;
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint4 v = ok ? *reinterpret_cast<const uint4*>(Q + qrow * ldq + 16 * ks + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
        qf[ks] = *reinterpret_cast<const bf16x8*>(&v);
    }
    const float qs = scale * LOG2E;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float m = NEG_BIG, lsum = 0.f;
    const int krow = pi_row(j);
    uint4 kr[4];                                            // K of the next fill is prefetched; V is fetched at its LDS store (keeps the
    auto fetch = [&](int kb0) {                            // kernel within 128 VGPRs: 4 workgroups per CU = all 800 tiles resident at once)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r, c8, g;
            fill_map(tid, i, r, c8, g);
            kr[i] = ld_row8(Kb, ldk, kb0 + r, Nk, c8);
        }
    };
    fetch(0);
    for (int kb0 = 0; kb0 < Nk; kb0 += KB) {
#if ATT_VAR == 2 || ATT_VAR == 3 || ATT_VAR == 4
        if (kb0 == 0) {
#endif
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r, c8, g;
            fill_map(tid, i, r, c8, g);
            *reinterpret_cast<uint4*>(&Ks[r * LDR + c8]) = kr[i];
            st_t8(Vt, LDTB, c8, r, ld_row8(Vb, ldv, kb0 + r, Nk, c8), g);
        }
        __syncthreads();
#if ATT_VAR == 2 || ATT_VAR == 3 || ATT_VAR == 4
        }
#else
        if (kb0 + KB < Nk) fetch(kb0 + KB);
#endif
        // S^T tile of sub-tile `sub`: 4 chained MFMAs, issued asynchronously to the matrix pipe
        auto qk = [&](int sub) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const bf16_t* kp = Ks + (32 * sub + krow) * LDR + 8 * h;
#pragma unroll
#if ATT_VAR == 4
            for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[(ks + sub) & 3], qf[ks], s, 0, 0, 0);
#else
            for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(kp + 16 * ks), qf[ks], s, 0, 0, 0);
#endif
            return s;
        };
        // online softmax of a finished S^T tile (register VALU) followed by O^T += V^T P^T
        auto softmax_pv = [&](f32x16 s, int sub) {
            const int kv0 = kb0 + 32 * sub;
            if (kv0 + 32 > Nk) {                                // tail tile (wave-uniform)
#pragma unroll
                for (int r = 0; r < 16; ++r) if (kv0 + 16 * h + r >= Nk) s[r] = NEG_BIG;
            }
#if ATT_VAR == 1 || ATT_VAR == 3 || ATT_VAR == 4
            {
                const bf16_t* vp = Vt + j * LDTB + 32 * sub + 16 * h;
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const bf16x8 pb = pack8(s, 8 * k2);
#if ATT_VAR == 4
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[k2], pb, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[k2 + 2], pb, acc1, 0, 0, 0);
#else
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(vp + 8 * k2), pb, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(vp + 32 * LDTB + 8 * k2), pb, acc1, 0, 0, 0);
#endif
                }
                lsum = 1.f;
                return;
            }
#endif
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * qs;        // scaled maximum of this tile
            if (__any(mx > m + RESCALE_THR)) {                  // lazy rescale: rare once the running maximum has settled
                const float mn = fmaxf(m, mx);
                const float alpha = fast_exp2(m - mn);
                lsum *= alpha;
                m = mn;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] *= alpha; acc1[r] *= alpha; }
            }
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = fast_exp2(fmaf(s[r], qs, -m)); rs += s[r]; }
            rs += __shfl_xor(rs, 32, 64);
            lsum += rs;
            const bf16_t* vp = Vt + j * LDTB + 32 * sub + 16 * h;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const bf16x8 pb = pack8(s, 8 * k2);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(vp + 8 * k2), pb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(vp + 32 * LDTB + 8 * k2), pb, acc1, 0, 0, 0);
            }
        };
#pragma unroll 1
        for (int sub = 0; sub < KB / 32 && kb0 + 32 * sub < Nk; ++sub) softmax_pv(qk(sub), sub);
    }
    if (ok) {
        const float inv = 1.0f / lsum;
        bf16_t* orow = O + qrow * ldo;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<bf16_t>(orow + 8 * g + 4 * h, make_float4(acc0[4 * g] * inv, acc0[4 * g + 1] * inv, acc0[4 * g + 2] * inv, acc0[4 * g + 3] * inv));
            st4<bf16_t>(orow + 32 + 8 * g + 4 * h, make_float4(acc1[4 * g] * inv, acc1[4 * g + 1] * inv, acc1[4 * g + 2] * inv, acc1[4 * g + 3] * inv));
        }
        if (h == 0) lse[qrow] = (m + log2f(lsum)) * LN2;
    }
}

#endif

__global__ __launch_bounds__(256, 2) void attn_bwd_dq_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                                 const bf16_t* __restrict__ V, int ldv, long long skv,
                                                                 const bf16_t* __restrict__ dO, int lddo, const float* __restrict__ lse,
                                                                 const float* __restrict__ delta, bf16_t* __restrict__ dQ, int lddq, Segs sg,
                                                                 int Nk, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[KB * LDR];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[KB * LDR];
    __shared__ __attribute__((aligned(16))) bf16_t Kt[D * LDTB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    int row_base, q0, nq, b;
    locate_tile(sg, blockIdx.x, row_base, q0, nq, b);
    const int ql = q0 + wave * 32 + j;
    const bool ok = ql < nq;
    const long long qrow = (long long)row_base + ql;
    bf16x8 qf[4], dof[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint4 v = ok ? *reinterpret_cast<const uint4*>(Q + qrow * ldq + 16 * ks + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
        const uint4 g = ok ? *reinterpret_cast<const uint4*>(dO + qrow * lddo + 16 * ks + 8 * h) : make_uint4(0u, 0u, 0u, 0u);
        qf[ks] = *reinterpret_cast<const bf16x8*>(&v);
        dof[ks] = *reinterpret_cast<const bf16x8*>(&g);
    }
    const float qs = scale * LOG2E;
    const float l2 = ok ? lse[qrow] * LOG2E : 0.f;
    const float dl = ok ? delta[qrow] : 0.f;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int krow = pi_row(j);
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    uint4 kr[4], vr[4];
    auto fetch = [&](int kb0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r, c8, g;
            fill_map(tid, i, r, c8, g);
            kr[i] = ld_row8(Kb, ldk, kb0 + r, Nk, c8);
            vr[i] = ld_row8(Vb, ldv, kb0 + r, Nk, c8);
        }
    };
    fetch(0);
    for (int kb0 = 0; kb0 < Nk; kb0 += KB) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r, c8, g;
            fill_map(tid, i, r, c8, g);
            *reinterpret_cast<uint4*>(&Ks[r * LDR + c8]) = kr[i];
            st_t8(Kt, LDTB, c8, r, kr[i], g);
            *reinterpret_cast<uint4*>(&Vs[r * LDR + c8]) = vr[i];
        }
        __syncthreads();
        if (kb0 + KB < Nk) fetch(kb0 + KB);
#pragma unroll 1
        for (int sub = 0; sub < KB / 32; ++sub) {
            const int kv0 = kb0 + 32 * sub;
            if (kv0 >= Nk) break;
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
            const bf16_t* kp = Ks + (32 * sub + krow) * LDR + 8 * h;
            const bf16_t* vp = Vs + (32 * sub + krow) * LDR + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(kp + 16 * ks), qf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(vp + 16 * ks), dof[ks], dp, 0, 0, 0);
            }
            const bool tail = kv0 + 32 > Nk;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = fast_exp2(fmaf(s[r], qs, -l2));
                if (tail && kv0 + 16 * h + r >= Nk) p = 0.f;
                s[r] = p * (dp[r] - dl) * scale;
            }
            const bf16_t* kt = Kt + j * LDTB + 32 * sub + 16 * h;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const bf16x8 db = pack8(s, 8 * k2);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(kt + 8 * k2), db, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(kt + 32 * LDTB + 8 * k2), db, acc1, 0, 0, 0);
            }
        }
    }
    if (ok) {
        bf16_t* row = dQ + qrow * lddq;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<bf16_t>(row + 8 * g + 4 * h, make_float4(acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]));
            st4<bf16_t>(row + 32 + 8 * g + 4 * h, make_float4(acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]));
        }
    }
}

// dK/dV: workgroup = (32-key tile, image); its 4 waves stride over ALL 32-query tiles of that image across the segments.
__global__ __launch_bounds__(256) void attn_bwd_dkv_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                               const bf16_t* __restrict__ V, int ldv, long long skv,
                                                               const bf16_t* __restrict__ dO, int lddo, const float* __restrict__ lse,
                                                               const float* __restrict__ delta, float* __restrict__ dkv32, Segs sg, int Nk,
                                                               float scale) {
    // gridDim.z query splits share a (key tile, image): partial dK/dV are added atomically into the fp32 scratch dkv32
    // [B][Nk][128] (dK | dV), which attn_dkv_store_kernel converts to the bf16 outputs.
    constexpr int LDQT = 32 + 16;                  // 96-byte rows = 24 words (8-word spans of rows distinct mod 4 are disjoint)
    constexpr int PER_WAVE_B = (2 * 32 * LDR + 2 * D * LDQT) * 2 + 256;
    constexpr int RED_B = 4 * 2 * 64 * 33 * 4;
    constexpr int SMEM_B = (4 * PER_WAVE_B > RED_B) ? 4 * PER_WAVE_B : RED_B;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_B];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int b = blockIdx.y, kv0 = blockIdx.x * 32;
    bf16_t* Qs = reinterpret_cast<bf16_t*>(smem + wave * PER_WAVE_B);
    bf16_t* dOs = Qs + 32 * LDR;
    bf16_t* Qt = dOs + 32 * LDR;
    bf16_t* dOt = Qt + D * LDQT;
    float* lss = reinterpret_cast<float*>(dOt + D * LDQT);
    float* dls = lss + 32;
    const int key = kv0 + j;
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint4 a = ld_row8(K + b * skv, ldk, key, Nk, 16 * ks + 8 * h);
        const uint4 c = ld_row8(V + b * skv, ldv, key, Nk, 16 * ks + 8 * h);
        kf[ks] = *reinterpret_cast<const bf16x8*>(&a);
        vf[ks] = *reinterpret_cast<const bf16x8*>(&c);
    }
    const float qs = scale * LOG2E;
    f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk0[r] = dk1[r] = dv0[r] = dv1[r] = 0.f; }
    const int qrow = pi_row(j);
    const int ntiles = sg.t32[sg.n];
    uint4 qr[4], gr[4];
    float lr = 0.f, dr = 0.f;
    auto locate = [&](int t, long long& base, int& valid) {       // first row and number of valid rows of 32-query tile t
        int s = 0;
#pragma unroll
        for (int i = 1; i < 4; ++i) if (i < sg.n && t >= sg.t32[i]) s = i;
        const int q0 = (t - sg.t32[s]) * 32;
        base = (long long)sg.row0[s] + (long long)b * sg.nq[s] + q0;
        valid = min(32, sg.nq[s] - q0);
    };
    auto fetch = [&](int t) {
        long long base; int valid;
        locate(t, base, valid);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 16 * (i & 1) + (lane & 15), c8 = 32 * (i >> 1) + 8 * (lane >> 4);
            const bool okr = r < valid;
            qr[i] = okr ? *reinterpret_cast<const uint4*>(Q + (base + r) * ldq + c8) : make_uint4(0u, 0u, 0u, 0u);
            gr[i] = okr ? *reinterpret_cast<const uint4*>(dO + (base + r) * lddo + c8) : make_uint4(0u, 0u, 0u, 0u);
        }
        if (lane < 32) { const bool okr = lane < valid; lr = okr ? lse[base + lane] * LOG2E : 0.f; dr = okr ? delta[base + lane] : 0.f; }
    };
    const int tstep = 4 * gridDim.z, tfirst = blockIdx.z * 4 + wave;
    if (tfirst < ntiles) fetch(tfirst);
    for (int t = tfirst; t < ntiles; t += tstep) {
        long long base; int valid;
        locate(t, base, valid);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 16 * (i & 1) + (lane & 15), g = lane >> 4, c8 = 32 * (i >> 1) + 8 * g;
            *reinterpret_cast<uint4*>(&Qs[r * LDR + c8]) = qr[i];
            *reinterpret_cast<uint4*>(&dOs[r * LDR + c8]) = gr[i];
            st_t8(Qt, LDQT, c8, r, qr[i], g);
            st_t8(dOt, LDQT, c8, r, gr[i], g);
        }
        if (lane < 32) { lss[lane] = lr; dls[lane] = dr; }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        if (t + tstep < ntiles) fetch(t + tstep);
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        const bf16_t* qp = Qs + qrow * LDR + 8 * h;
        const bf16_t* gp = dOs + qrow * LDR + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(qp + 16 * ks), kf[ks], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(gp + 16 * ks), vf[ks], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qi = 16 * h + r;
            const float p = (qi < valid) ? fast_exp2(fmaf(s[r], qs, -lss[qi])) : 0.f;
            dp[r] = p * (dp[r] - dls[qi]) * scale;
            s[r] = p;
        }
        const bf16_t* gt = dOt + j * LDQT + 16 * h;
        const bf16_t* qt = Qt + j * LDQT + 16 * h;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const bf16x8 pb = pack8(s, 8 * k2), db = pack8(dp, 8 * k2);
            dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(gt + 8 * k2), pb, dv0, 0, 0, 0);
            dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(gt + 32 * LDQT + 8 * k2), pb, dv1, 0, 0, 0);
            dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(qt + 8 * k2), db, dk0, 0, 0, 0);
            dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(qt + 32 * LDQT + 8 * k2), db, dk1, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    constexpr int RW = 2 * 64 * 33;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int d = d_row(r, h);
        red[wave * RW + (d) * 33 + j] = dk0[r];
        red[wave * RW + (32 + d) * 33 + j] = dk1[r];
        red[wave * RW + 64 * 33 + (d) * 33 + j] = dv0[r];
        red[wave * RW + 64 * 33 + (32 + d) * 33 + j] = dv1[r];
    }
    __syncthreads();
    for (int f = tid; f < 2 * 32 * D; f += 256) {
        const int which = f / (32 * D), kk = (f % (32 * D)) / D, d = f % D;
        if (kv0 + kk >= Nk) continue;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) v += red[w * RW + which * 64 * 33 + d * 33 + kk];
        atomicAdd(dkv32 + ((long long)b * Nk + kv0 + kk) * 128 + which * 64 + d, v);
    }
}

__global__ __launch_bounds__(256) void attn_dkv_store_kernel(const float* __restrict__ dkv32, bf16_t* __restrict__ dK, int lddk,
                                                             bf16_t* __restrict__ dV, int lddv, long long sdkv, int B, int Nk) {
    const long long n = (long long)B * Nk * 32;                 // float4 groups
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int q = (int)(i & 31), key = (int)((i >> 5) % Nk), b = (int)((i >> 5) / Nk);
        const float4 v = *reinterpret_cast<const float4*>(dkv32 + i * 4);
        bf16_t* dst = (q < 16 ? dK + b * sdkv + (long long)key * lddk + q * 4 : dV + b * sdkv + (long long)key * lddv + (q - 16) * 4);
        st4<bf16_t>(dst, v);
    }
}

__global__ __launch_bounds__(256) void delta_rows_kernel(const bf16_t* __restrict__ O, int ldo, const bf16_t* __restrict__ dO, int lddo,
                                                         float* __restrict__ delta, long long rows) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float s = wave_sum(bf2f(O[row * ldo + lane]) * bf2f(dO[row * lddo + lane]));
    if (lane == 0) delta[row] = s;
}

bool make_segs(Segs& sg, int B, int nseg, const int* nq, long long& total_rows) {
    if (nseg < 1 || nseg > 4) return false;
    sg.n = nseg;
    int row = 0;
    sg.tile0[0] = 0;
    sg.t32[0] = 0;
    for (int i = 0; i < 4; ++i) {
        const int n = i < nseg ? nq[i] : 0;
        if (i < nseg && n <= 0) return false;
        sg.nq[i] = n;
        sg.row0[i] = row;
        row += B * n;
        sg.tile0[i + 1] = sg.tile0[i] + B * ((n + 127) / 128);
        sg.t32[i + 1] = sg.t32[i] + (n + 31) / 32;
    }
    total_rows = row;
    return true;
}

}  // namespace

extern "C" int tc_attn_fwd_seg(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, long long skv, void* O, int ldo,
                               float* lse, int B, int nseg, const int* nq, int Nk, float scale, int dtype, void* stream) {
    Segs sg;
    long long rows;
    if (!Q || !K || !V || !O || !lse || !nq || B <= 0 || Nk <= 0 || !make_segs(sg, B, nseg, nq, rows)) return TC_ERR_ARG;
    if (dtype == TC_F32) {                                      // parity path: one fp32 launch per segment
        for (int i = 0; i < nseg; ++i) {
            const long long off = sg.row0[i];
            const int rc = tc_attn_fwd((const float*)Q + off * ldq, ldq, (long long)nq[i] * ldq, K, ldk, V, ldv, skv, (float*)O + off * ldo, ldo,
                                       (long long)nq[i] * ldo, lse + off, B, nq[i], Nk, scale, dtype, stream);
            if (rc != TC_OK) return rc;
        }
        return TC_OK;
    }
    if (dtype != TC_BF16 || ((ldq | ldk | ldv) & 7) || (skv & 7) || (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V) & 15) || (ldo & 3)) return TC_ERR_ARG;
#if (ATT_VAR >= 60 && ATT_VAR < 70) || (ATT_VAR >= 80 && ATT_VAR < 90)
    {
        const size_t smem = (size_t)3 * SLOT30 * sizeof(bf16_t);
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)attn_fwd_seg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); attr = true; }
        const unsigned g60 = (unsigned)B * ((sg.t32[nseg] + 4 * NQ60 - 1) / (4 * NQ60));
        hipLaunchKernelGGL(attn_fwd_seg_kernel, dim3(g60), dim3(256), smem, (hipStream_t)stream, (const bf16_t*)Q, ldq, (const bf16_t*)K, ldk,
                           (const bf16_t*)V, ldv, skv, (bf16_t*)O, ldo, lse, sg, Nk, scale);
        return tc_launch_status();
    }
#elif (ATT_VAR >= 30 && ATT_VAR < 50) || (ATT_VAR >= 70 && ATT_VAR < 80) || (ATT_VAR >= 90 && ATT_VAR < 100)
    {
        const size_t smem = ((size_t)(((ATT_VAR >= 40 && ATT_VAR < 90) || ATT_VAR >= 92) ? 3 : 2) * 2 * KB30 * LDR + (ATT_VAR >= 91 ? NW30 * 32 * LDR : 0)) * sizeof(bf16_t);
        static_assert(ATT_VAR < 92 || (3 * 2 * KB30 * LDR + NW30 * 32 * LDR) * 2 <= 160 * 1024, "LDS");
        static bool attr = false;
        if (!attr) { hipFuncSetAttribute((const void*)attn_fwd_seg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); attr = true; }
        const unsigned g30 = (unsigned)B * ((sg.t32[nseg] + NW30 - 1) / NW30);
        hipLaunchKernelGGL(attn_fwd_seg_kernel, dim3(g30), dim3(NT30), smem, (hipStream_t)stream, (const bf16_t*)Q, ldq, (const bf16_t*)K, ldk,
                           (const bf16_t*)V, ldv, skv, (bf16_t*)O, ldo, lse, sg, Nk, scale);
        return tc_launch_status();
    }
#elif ATT_VAR == 11 || ATT_VAR == 12
    hipLaunchKernelGGL(attn_fwd_seg_kernel, dim3(sg.tile0[nseg]), dim3(128), 0, (hipStream_t)stream, (const bf16_t*)Q, ldq, (const bf16_t*)K, ldk,
                       (const bf16_t*)V, ldv, skv, (bf16_t*)O, ldo, lse, sg, Nk, scale);
    return tc_launch_status();
#elif ATT_VAR == 10
    const unsigned fwd_grid = (unsigned)B * ((sg.t32[nseg] + 5) / 6);
    hipLaunchKernelGGL(attn_fwd_seg_kernel, dim3(fwd_grid), dim3(384), 0, (hipStream_t)stream, (const bf16_t*)Q, ldq, (const bf16_t*)K, ldk,
                       (const bf16_t*)V, ldv, skv, (bf16_t*)O, ldo, lse, sg, Nk, scale);
    return tc_launch_status();
#elif (ATT_VAR >= 5 && ATT_VAR < 13) || ATT_VAR == 16 || ATT_VAR == 17 || ATT_VAR == 21 || ATT_VAR == 23 || ATT_VAR == 24 || ATT_VAR == 25 || ATT_VAR == 26 || ATT_VAR == 27
    const unsigned fwd_grid = (unsigned)B * ((sg.t32[nseg] + 3) / 4);
#else
    const unsigned fwd_grid = sg.tile0[nseg];
#endif
#if ATT_VAR != 10 && ATT_VAR != 11 && ATT_VAR != 12 && !(ATT_VAR >= 30 && ATT_VAR < 100)
    hipLaunchKernelGGL(attn_fwd_seg_kernel, dim3(fwd_grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)Q, ldq, (const bf16_t*)K, ldk,
                       (const bf16_t*)V, ldv, skv, (bf16_t*)O, ldo, lse, sg, Nk, scale);
    return tc_launch_status();
#endif
}

extern "C" int tc_attn_bwd_seg(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, long long skv, const void* O, int ldo,
                               const void* dO, int lddo, const float* lse, float* delta, float* dkv32, void* dQ, int lddq, void* dK, int lddk,
                               void* dV, int lddv, long long sdkv, int B, int nseg, const int* nq, int Nk, float scale, int dtype,
                               void* stream) {
    Segs sg;
    long long rows;
    if (!Q || !K || !V || !O || !dO || !lse || !delta || !dQ || !dK || !dV || !nq || B <= 0 || Nk <= 0 || !make_segs(sg, B, nseg, nq, rows))
        return TC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == TC_F32) {
        for (int i = 0; i < nseg; ++i) {
            const long long off = sg.row0[i];
            const int rc = tc_attn_bwd((const float*)Q + off * ldq, ldq, (long long)nq[i] * ldq, K, ldk, V, ldv, skv, (const float*)O + off * ldo, ldo,
                                       (long long)nq[i] * ldo, (const float*)dO + off * lddo, lddo, (long long)nq[i] * lddo, lse + off, delta + off,
                                       (float*)dQ + off * lddq, lddq, (long long)nq[i] * lddq, dK, lddk, dV, lddv, sdkv, i > 0, B, nq[i], Nk,
                                       scale, dtype, stream);
            if (rc != TC_OK) return rc;
        }
        return TC_OK;
    }
    if (dtype != TC_BF16 || !dkv32 || ((ldq | ldk | ldv | lddo) & 7) || (skv & 7) ||
        (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)dO) & 15) || ((ldo | lddq | lddk | lddv) & 3) || (sdkv & 3))
        return TC_ERR_ARG;
    if (hipMemsetAsync(dkv32, 0, sizeof(float) * (size_t)B * Nk * 128, s) != hipSuccess) return TC_ERR_LAUNCH;
    hipLaunchKernelGGL(delta_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (const bf16_t*)O, ldo, (const bf16_t*)dO, lddo, delta, rows);
    hipLaunchKernelGGL(attn_bwd_dkv_seg_kernel, dim3((Nk + 31) / 32, B, 4), dim3(256), 0, s, (const bf16_t*)Q, ldq, (const bf16_t*)K, ldk,
                       (const bf16_t*)V, ldv, skv, (const bf16_t*)dO, lddo, lse, delta, dkv32, sg, Nk, scale);
    hipLaunchKernelGGL(attn_dkv_store_kernel, dim3(tc_blocks((long long)B * Nk * 32, 256, 1024)), dim3(256), 0, s, dkv32, (bf16_t*)dK, lddk,
                       (bf16_t*)dV, lddv, sdkv, B, Nk);
    hipLaunchKernelGGL(attn_bwd_dq_seg_kernel, dim3(sg.tile0[nseg]), dim3(256), 0, s, (const bf16_t*)Q, ldq, (const bf16_t*)K, ldk, (const bf16_t*)V,
                       ldv, skv, (const bf16_t*)dO, lddo, lse, delta, (bf16_t*)dQ, lddq, sg, Nk, scale);
    return tc_launch_status();
}
