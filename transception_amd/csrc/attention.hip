// Fused single-head (d = 64) attention for the Dual Transformer Bridge: O = softmax(Q K^T * scale) V with
// spatial-reduction K/V (Nk << Nq), forward and backward, never materialising the Nq x Nk score matrix.
//
// fp32 matrix-core version (v_mfma_f32_32x32x2_f32, exact fp32 products): every wavefront owns 32 queries and
// computes the TRANSPOSED score tile S^T = K Q^T, so that a lane holds one query column and 16 of the 32 keys in
// registers: the online-softmax max/sum are register reductions plus one cross-half shuffle, the probabilities
// feed the second MFMA (O^T += V^T P^T) straight from registers as its B operand, and the running rescale is a
// per-lane scalar.  K rows are read through the permutation pi so that register r of half h is key 16h + r.
// K/V tiles (32 keys) are staged through LDS once per 128-query workgroup.
//
// Backward = delta kernel + dK/dV kernel (workgroup owns a 32-key tile, loops over queries) + dQ kernel
// (workgroup owns 128 queries, loops over keys); scores are recomputed from the saved log-sum-exp.
#include "tc_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int D = 64;        // head dim
constexpr int KT = 32;       // keys per tile
constexpr int LDK = D + 1;   // LDS row stride (floats) for tiles read "row = lane"
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f
#define NEG_BIG (-1.0e30f)

__device__ __forceinline__ int pi_row(int i) { return 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3); }
__device__ __forceinline__ int d_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }   // MFMA 32x32 D row of reg r

// cooperative load of a [32 x 64] row tile (rows row0.., zero beyond nrows) into LDS with row stride lds_ld
template <typename T>
__device__ __forceinline__ void load_tile(float* dst, int lds_ld, const T* src, int ld, int row0, int nrows, int tid, int nthreads) {
    for (int f = tid; f < KT * D / 4; f += nthreads) {
        const int r = f >> 4, c4 = (f & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + r < nrows) v = ld4<T>(src + (long long)(row0 + r) * ld + c4);
        float* d = dst + r * lds_ld + c4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
}

// ------------------------------------------------------------------------------------------------ forward
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ Q, int ldq, long long sq, const T* __restrict__ K, int ldk,
                                                       const T* __restrict__ V, int ldv, long long skv, T* __restrict__ O, int ldo,
                                                       long long so, float* __restrict__ lse, int Nq, int Nk, float scale) {
    __shared__ float Ks[KT * LDK];
    __shared__ float Vs[KT * D];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int q = blockIdx.x * 128 + wave * 32 + j;
    const T* Qb = Q + b * sq;
    const T* Kb = K + b * skv;
    const T* Vb = V + b * skv;
    float qreg[32];
    const float qs = scale * LOG2E;
#pragma unroll
    for (int t = 0; t < 32; ++t) qreg[t] = (q < Nq) ? ldf<T>(Qb + (long long)q * ldq + 2 * t + h) * qs : 0.f;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float m = NEG_BIG, lsum = 0.f;
    const int krow = pi_row(j);
    for (int kv0 = 0; kv0 < Nk; kv0 += KT) {
        __syncthreads();
        load_tile<T>(Ks, LDK, Kb, ldk, kv0, Nk, tid, 256);
        load_tile<T>(Vs, D, Vb, ldv, kv0, Nk, tid, 256);
        __syncthreads();
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const float* kp = Ks + krow * LDK + h;
#pragma unroll
        for (int t = 0; t < 32; ++t) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * t], qreg[t], s, 0, 0, 0);
        float mx = NEG_BIG;
#pragma unroll
        for (int r = 0; r < 16; ++r) { if (kv0 + 16 * h + r >= Nk) s[r] = NEG_BIG; mx = fmaxf(mx, s[r]); }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx);
        const float alpha = exp2f(m - mn);
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = exp2f(s[r] - mn); rs += s[r]; }
        rs += __shfl_xor(rs, 32, 64);
        lsum = lsum * alpha + rs;
        m = mn;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] *= alpha; acc1[r] *= alpha; }
        const float* vp = Vs + (16 * h) * D + j;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[t * D], s[t], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[t * D + 32], s[t], acc1, 0, 0, 0);
        }
    }
    if (q < Nq) {
        const float inv = 1.0f / lsum;
        T* orow = O + b * so + (long long)q * ldo;
#pragma unroll
        for (int g = 0; g < 4; ++g) {            // register group g holds d = 8g + 4h + (0..3)
            st4<T>(orow + 8 * g + 4 * h, make_float4(acc0[4 * g] * inv, acc0[4 * g + 1] * inv, acc0[4 * g + 2] * inv, acc0[4 * g + 3] * inv));
            st4<T>(orow + 32 + 8 * g + 4 * h, make_float4(acc1[4 * g] * inv, acc1[4 * g + 1] * inv, acc1[4 * g + 2] * inv, acc1[4 * g + 3] * inv));
        }
        if (h == 0) lse[(long long)b * Nq + q] = (m + log2f(lsum)) * LN2;
    }
}

// ------------------------------------------------------------------------------------------------ backward
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(const T* __restrict__ O, int ldo, long long so, const T* __restrict__ dO, int lddo,
                                                         long long sdo, float* __restrict__ delta, int Nq, int B) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)B * Nq) return;
    const int b = (int)(row / Nq), q = (int)(row % Nq);
    const float v = ldf<T>(O + b * so + (long long)q * ldo + lane) * ldf<T>(dO + b * sdo + (long long)q * lddo + lane);
    const float s = wave_sum(v);
    if (lane == 0) delta[row] = s;
}

// dQ: workgroup = 128 queries (wave = 32), loop over key tiles.  Same S^T / P^T register layout as forward.
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const T* __restrict__ Q, int ldq, long long sq, const T* __restrict__ K, int ldk,
                                                          const T* __restrict__ V, int ldv, long long skv, const T* __restrict__ dO,
                                                          int lddo, long long sdo, const float* __restrict__ lse,
                                                          const float* __restrict__ delta, T* __restrict__ dQ, int lddq, long long sdq,
                                                          int Nq, int Nk, float scale) {
    __shared__ float Ks[KT * LDK];
    __shared__ float Vs[KT * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int q = blockIdx.x * 128 + wave * 32 + j;
    const bool ok = q < Nq;
    const T* Qb = Q + b * sq;
    const T* dOb = dO + b * sdo;
    float qreg[32], doreg[32];
    const float qs = scale * LOG2E;
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        qreg[t] = ok ? ldf<T>(Qb + (long long)q * ldq + 2 * t + h) * qs : 0.f;
        doreg[t] = ok ? ldf<T>(dOb + (long long)q * lddo + 2 * t + h) : 0.f;
    }
    const float l2 = ok ? lse[(long long)b * Nq + q] * LOG2E : 0.f;
    const float dl = ok ? delta[(long long)b * Nq + q] : 0.f;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int krow = pi_row(j);
    for (int kv0 = 0; kv0 < Nk; kv0 += KT) {
        __syncthreads();
        load_tile<T>(Ks, LDK, K + b * skv, ldk, kv0, Nk, tid, 256);
        load_tile<T>(Vs, LDK, V + b * skv, ldv, kv0, Nk, tid, 256);
        __syncthreads();
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        const float* kp = Ks + krow * LDK + h;
        const float* vp = Vs + krow * LDK + h;
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * t], qreg[t], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[2 * t], doreg[t], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = (kv0 + 16 * h + r < Nk) ? exp2f(s[r] - l2) : 0.f;
            s[r] = p * (dp[r] - dl) * scale;                          // dS^T
        }
        const float* kr = Ks + (16 * h) * LDK + j;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[t * LDK], s[t], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[t * LDK + 32], s[t], acc1, 0, 0, 0);
        }
    }
    if (ok) {
        T* row = dQ + b * sdq + (long long)q * lddq;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<T>(row + 8 * g + 4 * h, make_float4(acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]));
            st4<T>(row + 32 + 8 * g + 4 * h, make_float4(acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]));
        }
    }
}

// dK / dV: workgroup = one 32-key tile (K, V rows held in registers as MFMA B operands), the 4 waves stride over
// 32-query tiles staged in per-wave LDS; S = Q K^T is computed UN-transposed here (lane = key, register = query)
// so that P and dS feed the dV^T / dK^T MFMAs from registers.  Wave partials are reduced through LDS at the end.
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const T* __restrict__ Q, int ldq, long long sq, const T* __restrict__ K, int ldk,
                                                           const T* __restrict__ V, int ldv, long long skv, const T* __restrict__ dO,
                                                           int lddo, long long sdo, const float* __restrict__ lse,
                                                           const float* __restrict__ delta, T* __restrict__ dK, int lddk,
                                                           T* __restrict__ dV, int lddv, long long sdkv, int Nq, int Nk, float scale,
                                                           int accumulate) {
    constexpr int PER_WAVE = 2 * KT * LDK + 64;   // per wave: Qs[32][65] | dOs[32][65] | lse2[32] | dl[32]
    __shared__ float smem[4 * PER_WAVE];          // 66 KiB static LDS (gfx950 allows up to 160 KiB per workgroup)
    static_assert(PER_WAVE >= 2 * 64 * 33, "reduction scratch must fit");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int b = blockIdx.y, kv0 = blockIdx.x * KT;
    float* Qs = smem + wave * PER_WAVE;
    float* dOs = Qs + KT * LDK;
    float* lss = dOs + KT * LDK;
    float* dls = lss + 32;
    const int key = kv0 + j;
    const bool kok = key < Nk;
    float kreg[32], vreg[32];
    const float qs = scale * LOG2E;
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        kreg[t] = kok ? ldf<T>(K + b * skv + (long long)key * ldk + 2 * t + h) * qs : 0.f;
        vreg[t] = kok ? ldf<T>(V + b * skv + (long long)key * ldv + 2 * t + h) : 0.f;
    }
    f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk0[r] = dk1[r] = dv0[r] = dv1[r] = 0.f; }
    const T* Qb = Q + b * sq;
    const T* dOb = dO + b * sdo;
    const int qrow = pi_row(j);
    for (int q0 = wave * 32; q0 < Nq; q0 += 128) {
        // stage this wave's 32-query tile (a wavefront's DS operations execute in order; no cross-wave sharing)
        load_tile<T>(Qs, LDK, Qb, ldq, q0, Nq, lane, 64);
        load_tile<T>(dOs, LDK, dOb, lddo, q0, Nq, lane, 64);
        if (lane < 32) {
            const bool ok = q0 + lane < Nq;
            lss[lane] = ok ? lse[(long long)b * Nq + q0 + lane] * LOG2E : 0.f;
            dls[lane] = ok ? delta[(long long)b * Nq + q0 + lane] : 0.f;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): LDS writes of this wave have landed
        __builtin_amdgcn_wave_barrier();
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        // S[query][key] and dP[query][key]: A = Q / dO rows (read through pi so register r of half h is query 16h + r)
        const float* qp = Qs + qrow * LDK + h;
        const float* gp = dOs + qrow * LDK + h;
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(qp[2 * t], kreg[t], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(gp[2 * t], vreg[t], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ql = 16 * h + r;
            const float p = (q0 + ql < Nq) ? exp2f(s[r] - lss[ql]) : 0.f;
            dp[r] = p * (dp[r] - dls[ql]) * scale;     // dS
            s[r] = p;                                   // P
        }
        // dV^T[d][key] += dO^T[d][q] P[q][key] ; dK^T[d][key] += Q^T[d][q] dS[q][key]
        const float* gr = dOs + (16 * h) * LDK + j;
        const float* qr = Qs + (16 * h) * LDK + j;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            dv0 = __builtin_amdgcn_mfma_f32_32x32x2f32(gr[t * LDK], s[t], dv0, 0, 0, 0);
            dv1 = __builtin_amdgcn_mfma_f32_32x32x2f32(gr[t * LDK + 32], s[t], dv1, 0, 0, 0);
            dk0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qr[t * LDK], dp[t], dk0, 0, 0, 0);
            dk1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qr[t * LDK + 32], dp[t], dk1, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // cross-wave reduction: every wave dumps its [64 d][32 keys] partials, then 256 threads sum and store
    __syncthreads();
    float* red = smem;                          // reuse: [4 waves][2 (dk,dv)][64 d][33]
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
    for (int f = tid; f < 2 * KT * D; f += 256) {
        const int which = f / (KT * D), kk = (f % (KT * D)) / D, d = f % D;
        if (kv0 + kk >= Nk) continue;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) v += red[w * RW + which * 64 * 33 + d * 33 + kk];
        // dK was accumulated against K*scale*log2e-free operands: Q rows are unscaled, dS already carries `scale`
        T* dst = (which == 0 ? dK + b * sdkv + (long long)(kv0 + kk) * lddk : dV + b * sdkv + (long long)(kv0 + kk) * lddv) + d;
        if (accumulate) v += ldf<T>(dst);
        stf<T>(dst, v);
    }
}

}  // namespace

// =====================================================================================================================
// bf16 matrix-core versions (v_mfma_f32_32x32x16_bf16, fp32 accumulate, fp32 softmax statistics).  Same lane ownership as
// the fp32 kernels: S^T = K Q^T puts one query column per lane with keys 16h + r in registers (rows read through pi), so
// P^T / dS^T pack straight into the bf16x8 B operand of the second MFMA; the A operand of that MFMA (V^T, K^T, dO^T, Q^T)
// comes from a transposed LDS tile with one 16-byte read.  128 keys are staged per LDS fill (4 MFMA sub-tiles per barrier).
// =====================================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int KB = 128;          // keys per LDS fill
constexpr int LDR = D + 8;       // bf16 row stride of row-major [key][d] tiles (144 B)
constexpr int LDTB = KB + 8;     // bf16 row stride of transposed [d][key] tiles (272 B)

__device__ __forceinline__ bf16x8 ld_frag(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int o) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (__bf16)v[o + i];
    return r;
}
// 8 consecutive bf16 of a global row (zero beyond the valid rows); rows are 16-byte aligned (ld % 8 == 0 checked on the host)
__device__ __forceinline__ uint4 ld_row8(const bf16_t* base, int ld, int row, int nrows, int c8) {
    return row < nrows ? *reinterpret_cast<const uint4*>(base + (long long)row * ld + c8) : make_uint4(0u, 0u, 0u, 0u);
}
__device__ __forceinline__ void st_t8(bf16_t* dst, int ldt, int c8, int col, uint4 v) {      // dst[(c8+i)*ldt + col] = v[i]
    union { uint4 v; bf16_t e[8]; } u;
    u.v = v;
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[(c8 + i) * ldt + col] = u.e[i];
}

__global__ __launch_bounds__(256) void attn_fwd_bf16_kernel(const bf16_t* __restrict__ Q, int ldq, long long sq, const bf16_t* __restrict__ K,
                                                            int ldk, const bf16_t* __restrict__ V, int ldv, long long skv,
                                                            bf16_t* __restrict__ O, int ldo, long long so, float* __restrict__ lse, int Nq,
                                                            int Nk, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[KB * LDR];
    __shared__ __attribute__((aligned(16))) bf16_t Vt[D * LDTB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int q = blockIdx.x * 128 + wave * 32 + j;
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint4 v = ld_row8(Q + b * sq, ldq, q, Nq, 16 * ks + 8 * h);
        qf[ks] = *reinterpret_cast<const bf16x8*>(&v);
    }
    const float qs = scale * LOG2E;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float m = NEG_BIG, lsum = 0.f;
    const int krow = pi_row(j);
    for (int kb0 = 0; kb0 < Nk; kb0 += KB) {
        __syncthreads();
        for (int f = tid; f < KB * D / 8; f += 256) {
            const int r = f >> 3, c8 = (f & 7) * 8;
            *reinterpret_cast<uint4*>(&Ks[r * LDR + c8]) = ld_row8(Kb, ldk, kb0 + r, Nk, c8);
            st_t8(Vt, LDTB, c8, r, ld_row8(Vb, ldv, kb0 + r, Nk, c8));
        }
        __syncthreads();
#pragma unroll 1
        for (int sub = 0; sub < KB / 32; ++sub) {
            const int kv0 = kb0 + 32 * sub;
            if (kv0 >= Nk) break;
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const bf16_t* kp = Ks + (32 * sub + krow) * LDR + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(kp + 16 * ks), qf[ks], s, 0, 0, 0);
            float mx = NEG_BIG;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = (kv0 + 16 * h + r < Nk) ? s[r] * qs : NEG_BIG; mx = fmaxf(mx, s[r]); }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m, mx);
            const float alpha = exp2f(m - mn);
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = exp2f(s[r] - mn); rs += s[r]; }
            rs += __shfl_xor(rs, 32, 64);
            lsum = lsum * alpha + rs;
            m = mn;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] *= alpha; acc1[r] *= alpha; }
            const bf16_t* vp = Vt + j * LDTB + 32 * sub + 16 * h;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const bf16x8 pb = pack8(s, 8 * k2);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(vp + 8 * k2), pb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(vp + 32 * LDTB + 8 * k2), pb, acc1, 0, 0, 0);
            }
        }
    }
    if (q < Nq) {
        const float inv = 1.0f / lsum;
        bf16_t* orow = O + b * so + (long long)q * ldo;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<bf16_t>(orow + 8 * g + 4 * h, make_float4(acc0[4 * g] * inv, acc0[4 * g + 1] * inv, acc0[4 * g + 2] * inv, acc0[4 * g + 3] * inv));
            st4<bf16_t>(orow + 32 + 8 * g + 4 * h, make_float4(acc1[4 * g] * inv, acc1[4 * g + 1] * inv, acc1[4 * g + 2] * inv, acc1[4 * g + 3] * inv));
        }
        if (h == 0) lse[(long long)b * Nq + q] = (m + log2f(lsum)) * LN2;
    }
}

__global__ __launch_bounds__(256) void attn_bwd_dq_bf16_kernel(const bf16_t* __restrict__ Q, int ldq, long long sq, const bf16_t* __restrict__ K,
                                                               int ldk, const bf16_t* __restrict__ V, int ldv, long long skv,
                                                               const bf16_t* __restrict__ dO, int lddo, long long sdo,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               bf16_t* __restrict__ dQ, int lddq, long long sdq, int Nq, int Nk, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[KB * LDR];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[KB * LDR];
    __shared__ __attribute__((aligned(16))) bf16_t Kt[D * LDTB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int q = blockIdx.x * 128 + wave * 32 + j;
    const bool ok = q < Nq;
    bf16x8 qf[4], dof[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const uint4 v = ld_row8(Q + b * sq, ldq, q, Nq, 16 * ks + 8 * h);
        const uint4 g = ld_row8(dO + b * sdo, lddo, q, Nq, 16 * ks + 8 * h);
        qf[ks] = *reinterpret_cast<const bf16x8*>(&v);
        dof[ks] = *reinterpret_cast<const bf16x8*>(&g);
    }
    const float qs = scale * LOG2E;
    const float l2 = ok ? lse[(long long)b * Nq + q] * LOG2E : 0.f;
    const float dl = ok ? delta[(long long)b * Nq + q] : 0.f;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int krow = pi_row(j);
    for (int kb0 = 0; kb0 < Nk; kb0 += KB) {
        __syncthreads();
        for (int f = tid; f < KB * D / 8; f += 256) {
            const int r = f >> 3, c8 = (f & 7) * 8;
            const uint4 kv = ld_row8(K + b * skv, ldk, kb0 + r, Nk, c8);
            *reinterpret_cast<uint4*>(&Ks[r * LDR + c8]) = kv;
            st_t8(Kt, LDTB, c8, r, kv);
            *reinterpret_cast<uint4*>(&Vs[r * LDR + c8]) = ld_row8(V + b * skv, ldv, kb0 + r, Nk, c8);
        }
        __syncthreads();
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
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = (kv0 + 16 * h + r < Nk) ? exp2f(s[r] * qs - l2) : 0.f;
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
        bf16_t* row = dQ + b * sdq + (long long)q * lddq;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<bf16_t>(row + 8 * g + 4 * h, make_float4(acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]));
            st4<bf16_t>(row + 32 + 8 * g + 4 * h, make_float4(acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]));
        }
    }
}

__global__ __launch_bounds__(256) void attn_bwd_dkv_bf16_kernel(const bf16_t* __restrict__ Q, int ldq, long long sq, const bf16_t* __restrict__ K,
                                                                int ldk, const bf16_t* __restrict__ V, int ldv, long long skv,
                                                                const bf16_t* __restrict__ dO, int lddo, long long sdo,
                                                                const float* __restrict__ lse, const float* __restrict__ delta,
                                                                bf16_t* __restrict__ dK, int lddk, bf16_t* __restrict__ dV, int lddv,
                                                                long long sdkv, int Nq, int Nk, float scale, int accumulate) {
    constexpr int LDQT = 32 + 8;                                         // transposed [d][query] tiles, 80-byte rows
    constexpr int PER_WAVE_B = (2 * 32 * LDR + 2 * D * LDQT) * 2 + 256;  // bytes: Qs | dOs | Qt | dOt | lse2[32] | dl[32]
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
    for (int q0 = wave * 32; q0 < Nq; q0 += 128) {
        for (int f = lane; f < 32 * D / 8; f += 64) {
            const int r = f >> 3, c8 = (f & 7) * 8;
            const uint4 qv = ld_row8(Q + b * sq, ldq, q0 + r, Nq, c8);
            const uint4 gv = ld_row8(dO + b * sdo, lddo, q0 + r, Nq, c8);
            *reinterpret_cast<uint4*>(&Qs[r * LDR + c8]) = qv;
            *reinterpret_cast<uint4*>(&dOs[r * LDR + c8]) = gv;
            st_t8(Qt, LDQT, c8, r, qv);
            st_t8(dOt, LDQT, c8, r, gv);
        }
        if (lane < 32) {
            const bool ok = q0 + lane < Nq;
            lss[lane] = ok ? lse[(long long)b * Nq + q0 + lane] * LOG2E : 0.f;
            dls[lane] = ok ? delta[(long long)b * Nq + q0 + lane] : 0.f;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
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
            const int ql = 16 * h + r;
            const float p = (q0 + ql < Nq) ? exp2f(s[r] * qs - lss[ql]) : 0.f;
            dp[r] = p * (dp[r] - dls[ql]) * scale;
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
        bf16_t* dst = (which == 0 ? dK + b * sdkv + (long long)(kv0 + kk) * lddk : dV + b * sdkv + (long long)(kv0 + kk) * lddv) + d;
        if (accumulate) v += bf2f(*dst);
        *dst = f2bf(v);
    }
}

}  // namespace

static int tc_attn_fwd_bf16_impl(const void* Q, int ldq, long long sq, const void* K, int ldk, const void* V, int ldv, long long skv,
                                     void* O, int ldo, long long so, float* lse, int B, int Nq, int Nk, float scale, hipStream_t s) {
    if (((ldq | ldk | ldv) & 7) || ((sq | skv) & 7) || ((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V) & 15) return TC_ERR_ARG;
    hipLaunchKernelGGL(attn_fwd_bf16_kernel, dim3((Nq + 127) / 128, B), dim3(256), 0, s, (const bf16_t*)Q, ldq, sq, (const bf16_t*)K, ldk,
                       (const bf16_t*)V, ldv, skv, (bf16_t*)O, ldo, so, lse, Nq, Nk, scale);
    return tc_launch_status();
}

static int tc_attn_bwd_bf16_impl(const void* Q, int ldq, long long sq, const void* K, int ldk, const void* V, int ldv, long long skv,
                                     const void* O, int ldo, long long so, const void* dO, int lddo, long long sdo, const float* lse,
                                     float* delta, void* dQ, int lddq, long long sdq, void* dK, int lddk, void* dV, int lddv, long long sdkv,
                                     int accumulate_dkv, int B, int Nq, int Nk, float scale, hipStream_t s) {
    if (((ldq | ldk | ldv | lddo) & 7) || ((sq | skv | sdo) & 7) || ((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)dO) & 15)
        return TC_ERR_ARG;
    hipLaunchKernelGGL((attn_delta_kernel<bf16_t>), dim3((unsigned)(((long long)B * Nq + 3) / 4)), dim3(256), 0, s, (const bf16_t*)O, ldo, so,
                       (const bf16_t*)dO, lddo, sdo, delta, Nq, B);
    hipLaunchKernelGGL(attn_bwd_dkv_bf16_kernel, dim3((Nk + 31) / 32, B), dim3(256), 0, s, (const bf16_t*)Q, ldq, sq, (const bf16_t*)K, ldk,
                       (const bf16_t*)V, ldv, skv, (const bf16_t*)dO, lddo, sdo, lse, delta, (bf16_t*)dK, lddk, (bf16_t*)dV, lddv, sdkv, Nq,
                       Nk, scale, accumulate_dkv);
    hipLaunchKernelGGL(attn_bwd_dq_bf16_kernel, dim3((Nq + 127) / 128, B), dim3(256), 0, s, (const bf16_t*)Q, ldq, sq, (const bf16_t*)K, ldk,
                       (const bf16_t*)V, ldv, skv, (const bf16_t*)dO, lddo, sdo, lse, delta, (bf16_t*)dQ, lddq, sdq, Nq, Nk, scale);
    return tc_launch_status();
}

extern "C" int tc_attn_fwd(const void* Q, int ldq, long long sq, const void* K, int ldk, const void* V, int ldv, long long skv, void* O,
                           int ldo, long long so, float* lse, int B, int Nq, int Nk, float scale, int dtype, void* stream) {
    if (!Q || !K || !V || !O || !lse || B <= 0 || Nq <= 0 || Nk <= 0 || ((ldq | ldk | ldv | ldo) & 3) || ((sq | skv | so) & 3)) return TC_ERR_ARG;
    if (dtype == TC_BF16) return tc_attn_fwd_bf16_impl(Q, ldq, sq, K, ldk, V, ldv, skv, O, ldo, so, lse, B, Nq, Nk, scale, (hipStream_t)stream);
    dim3 grid((Nq + 127) / 128, B);
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((attn_fwd_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)Q, ldq, sq,
                                                (const T*)K, ldk, (const T*)V, ldv, skv, (T*)O, ldo, so, lse, Nq, Nk, scale));
    return tc_launch_status();
}

extern "C" int tc_attn_bwd(const void* Q, int ldq, long long sq, const void* K, int ldk, const void* V, int ldv, long long skv,
                           const void* O, int ldo, long long so, const void* dO, int lddo, long long sdo, const float* lse, float* delta,
                           void* dQ, int lddq, long long sdq, void* dK, int lddk, void* dV, int lddv, long long sdkv, int accumulate_dkv,
                           int B, int Nq, int Nk, float scale, int dtype, void* stream) {
    if (!Q || !K || !V || !O || !dO || !lse || !delta || !dQ || !dK || !dV || B <= 0 || Nq <= 0 || Nk <= 0 ||
        ((ldq | ldk | ldv | ldo | lddo | lddq | lddk | lddv) & 3) || ((sq | skv | so | sdo | sdq | sdkv) & 3))
        return TC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == TC_BF16)
        return tc_attn_bwd_bf16_impl(Q, ldq, sq, K, ldk, V, ldv, skv, O, ldo, so, dO, lddo, sdo, lse, delta, dQ, lddq, sdq, dK, lddk, dV, lddv,
                                     sdkv, accumulate_dkv, B, Nq, Nk, scale, s);
    TC_DISPATCH_DTYPE(dtype, {
        hipLaunchKernelGGL((attn_delta_kernel<T>), dim3((unsigned)(((long long)B * Nq + 3) / 4)), dim3(256), 0, s, (const T*)O, ldo, so,
                           (const T*)dO, lddo, sdo, delta, Nq, B);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<T>), dim3((Nk + KT - 1) / KT, B), dim3(256), 0, s, (const T*)Q, ldq, sq, (const T*)K, ldk,
                           (const T*)V, ldv, skv, (const T*)dO, lddo, sdo, lse, delta, (T*)dK, lddk, (T*)dV, lddv, sdkv, Nq, Nk, scale,
                           accumulate_dkv);
        hipLaunchKernelGGL((attn_bwd_dq_kernel<T>), dim3((Nq + 127) / 128, B), dim3(256), 0, s, (const T*)Q, ldq, sq, (const T*)K, ldk,
                           (const T*)V, ldv, skv, (const T*)dO, lddo, sdo, lse, delta, (T*)dQ, lddq, sdq, Nq, Nk, scale);
    });
    return tc_launch_status();
}

