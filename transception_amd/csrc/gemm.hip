// Batched GEMM for gfx950, used for every Linear / 1x1-conv / bmm of the TransCeption path and their gradients:
//   C[b] = alpha * op(A[b]) op(B[b]) (+bias) (+R) ; optional sigmoid ; optional accumulate / atomic accumulation.
// Two matrix-core paths selected by the storage type:
//   fp32 storage -> v_mfma_f32_32x32x2_f32  (exact fp32 products, the parity path),  K-slab 16
//   bf16 storage -> v_mfma_f32_32x32x16_bf16 (fp32 accumulate),                       K-slab 64
// 256 threads = 4 waves (2 x 2); each wave owns a (BM/2) x (BN/2) sub-tile made of 32x32 MFMA blocks.  Operands are
// staged through LDS with a register prefetch of the next K-slabs: fp32 always "K-inner" (As[BM][BK+pad], transposed on
// the LDS write); bf16 K-inner for K-contiguous operands and as-is ([BK][X+pad], gathered by ds_read_b64_tr_b16) for
// operands stored [K][X].  The bf16 K loop keeps its global loads free of per-lane branches (see strip_raw).  C may be fp32 while A/B are bf16
// (weight gradients accumulate into the fp32 gradient arena); an optional row-sum of op(A) (the bias gradient that
// belongs to a dW GEMM) is produced by the blocks of the first N-tile.
#include "tc_common.h"
#include <algorithm>
#include <type_traits>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// The two 16-bit storage types run the same kernels: tiles move as raw 16-bit words, only the MFMA operand type and the
// conversions differ (H = bf16_t or f16_t).
template <typename H> struct HalfOps;
template <> struct HalfOps<bf16_t> {
    typedef bf16x8 v8;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct HalfOps<f16_t> {
    typedef f16x8 v8;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

namespace {

// Division of a workgroup index by a launch constant as a multiplication (round 6): hipcc expands `n / d` with a run-time d into ~30 VALU
// instructions (v_rcp_iflag_f32 + fix-ups) although both are wave-uniform -- the tile-index prologues of gemm_pair_kernel / gemm_multi_kernel
// held 44 / 53 such sequences, several hundred instructions at the head of every workgroup.  m = floor(2^32 / d) + 1 is exact for
// n * d < 2^32 (the host checks against the largest n the launch can produce; m = 0 keeps the division).
static inline unsigned fdiv_make(long long d, unsigned long long nmax) {
    return (d > 1 && nmax * (unsigned long long)d < (1ull << 32)) ? (unsigned)((1ull << 32) / (unsigned long long)d + 1) : 0u;
}
__device__ __forceinline__ int fdiv(int n, int d, unsigned m) { return d == 1 ? n : (m ? (int)__umulhi((unsigned)n, m) : n / d); }

struct GemmDev {
    const void* A; const void* B; void* C; const void* bias; const void* R;
    int M, N, K, lda, ldb, ldc, ldr;
    int nb2, splitk, kchunk;
    unsigned m_splitk, m_nb2, m_gx;   // fdiv multipliers: blockIdx.z = (b1 * nb2 + b2) * splitk + ks < 65536; m_gx: the grid's x extent (tc_xcd_tile)
    long long sA1, sA2, sB1, sB2, sC1, sC2, sR1, sR2;
    float alpha; int accumulate, act;
    unsigned char vecA, vecB, vecC, atomic;   // (bytes: twelve of these structs + the per-problem arrays of gemm_multi_kernel fill the 4 KiB argument block)
    short vec8C;            // bf16 C rows are 16-byte addressable (LDS-staged, fully coalesced epilogue)
    short xcd;              // XCD-aware tile numbering (tc_xcd_tile)
    float* rowsum;          // optional: rowsum[m] += sum_k op(A)[m][k]  (fp32, atomics)
    long long sBias1, sRow1; // batch-level-1 strides of bias / rowsum (grouped weights)
    float* ws_part; int* ws_cnt; short fix_group, fix_ngroups;   // split-K fix-up workspace (fix_group > 0: enabled; group size 16, <= 4096 groups)
    int bgap_every; long long bgap;   // B stored [K,N] in blocks of bgap_every rows with bgap extra elements between blocks
    struct {                          // MixFFN_skip fusion hooks (TcGemm.ffn_*); mode mirrors the kernels' FFN template parameter
        int mode, nchunk, chunk_n, ldd, sRow1, sPar1;
        float eps;
        const float* part; float* stat; const void* gamma; const void* beta; const void* d; float* part2;
    } ffn;
    // BatchNorm statistics of the stored tile (TcGemm.bn_part / bn_shift; plain products only, so they share the slots of the MixFFN
    // hooks -- the argument block of gemm_multi_kernel holds twelve of these structs in 4 KiB): ffn.mode == GEMM_BN_STATS
    __device__ __host__ float* bn_part() const { return ffn.mode == 4 ? ffn.part2 : nullptr; }
    __device__ __host__ const float* bn_shift() const { return ffn.part; }
};
constexpr int GEMM_BN_STATS = 4;

// GELU(LayerNorm(.)) on raw 8 x bf16 / 4 x fp32 operand strips (the A rows of fc2's forward, the B rows of its weight gradient)
template <typename H> __device__ __forceinline__ void bf8_unpack(const uint4& r, float* o) {
    unpack2<H>(r.x, o[0], o[1]); unpack2<H>(r.y, o[2], o[3]); unpack2<H>(r.z, o[4], o[5]); unpack2<H>(r.w, o[6], o[7]);
}
template <typename H> __device__ __forceinline__ uint4 bf8_pack(const float* o) {
    return make_uint4(pack2<H>(o[0], o[1]), pack2<H>(o[2], o[3]), pack2<H>(o[4], o[5]), pack2<H>(o[6], o[7]));
}
template <typename H> __device__ __forceinline__ uint4 ffn_ln_gelu8(const uint4& v, float mean, float rstd, const float* g, const float* b) {
    float x[8];
    bf8_unpack<H>(v, x);
#pragma unroll
    for (int e = 0; e < 8; e += 2) {                          // pairs on the packed fp32 pipe
        const tc_f32x2 xv = {x[e], x[e + 1]}, gv = {g[e], g[e + 1]}, bv = {b[e], b[e + 1]};
        const tc_f32x2 u = gelu_poly2((xv - mean) * rstd * gv + bv);             // (16-bit storage: the polynomial CDF, tc_common.h)
        x[e] = u.x; x[e + 1] = u.y;
    }
    return bf8_pack<H>(x);
}
__device__ __forceinline__ float4 ffn_ln_gelu4(const float4& v, float mean, float rstd, const float4& g, const float4& b) {
    return make_float4(gelu_f((v.x - mean) * rstd * g.x + b.x), gelu_f((v.y - mean) * rstd * g.y + b.y),
                       gelu_f((v.z - mean) * rstd * g.z + b.z), gelu_f((v.w - mean) * rstd * g.w + b.w));
}
// TC_FFN_LN_A: merge the chunk partials of the tile's BM rows (Chan's formula), LPR lanes per row; result in LDS (and, from the
// workgroups of the first N tile / first K split, in ffn.stat for the backward pass)
template <int BM>
__device__ __forceinline__ void ffn_finalize_stats(const GemmDev& p, int b1, int m0, bool write_out, float2* s_stat) {
    constexpr int LPR = 256 / BM;
    const int tid = threadIdx.x, r = tid / LPR, q = tid % LPR, row = m0 + r;
    const long long rbase = (long long)b1 * p.ffn.sRow1;
    const int nch = p.ffn.nchunk;
    const float cn = (float)p.ffn.chunk_n, inv_cn = 1.0f / cn, invC = 1.0f / (cn * (float)nch);
    const float2* pp = reinterpret_cast<const float2*>(p.ffn.part) + (rbase + (row < p.M ? row : 0)) * nch;
    // this lane's partials are read ONCE, all loads in flight before the first use (two loops of dependent loads over a run-time count were
    // up to 2 x 16 L2 round trips at the head of every workgroup); more than 16 per lane (nch > 16 LPR): the plain loops
    float sm = 0.f, m2 = 0.f;
    constexpr int PMAX = 16;
    if (nch <= PMAX * LPR) {
        float2 t[PMAX];
#pragma unroll
        for (int e = 0; e < PMAX; ++e) t[e] = pp[min(q + e * LPR, nch - 1)];
#pragma unroll
        for (int e = 0; e < PMAX; ++e) if (q + e * LPR < nch) sm += t[e].x;
        sm = tc_group_sum<LPR>(sm);
        const float mean_ = sm * invC;
#pragma unroll
        for (int e = 0; e < PMAX; ++e) if (q + e * LPR < nch) { const float dm = t[e].x * inv_cn - mean_; m2 += t[e].y + cn * dm * dm; }
    } else {
        for (int k = q; k < nch; k += LPR) sm += pp[k].x;
        sm = tc_group_sum<LPR>(sm);
        const float mean_ = sm * invC;
        for (int k = q; k < nch; k += LPR) { const float2 t = pp[k]; const float dm = t.x * inv_cn - mean_; m2 += t.y + cn * dm * dm; }
    }
    const float mean = sm * invC;
    m2 = tc_group_sum<LPR>(m2);
    const float rstd = rsqrtf(m2 * invC + p.ffn.eps);
    if (q == 0) {
        s_stat[r] = make_float2(mean, rstd);
        if (write_out && row < p.M) reinterpret_cast<float2*>(p.ffn.stat)[rbase + row] = make_float2(mean, rstd);
    }
    __syncthreads();
}
// TC_FFN_EP on one value: gp = v * GELU'(xhat gamma + beta); accumulates the two LayerNorm-backward row sums
template <typename T> __device__ __forceinline__ float ffn_ep1(float v, float d, float mean, float rstd, float g, float b, float& s1, float& s2) {
    const float xh = (d - mean) * rstd;
    const float gp = v * gelu_grad_fT<T>(xh * g + b);
    s1 += gp * g; s2 += gp * g * xh;
    return gp;
}

// Split-K fix-up without a second launch.  The `splitk` workgroups of an output tile park their fp32 partial tiles in the
// workspace (plain coalesced stores), groups of `fix_group` consecutive splits share an arrival counter, and the member that
// arrives last sums its group's partials in split order (deterministic) and runs the normal epilogue: directly when the tile
// has one group (any output type, bias / residual / activation applied once), with fp32 atomics when it has several (weight
// gradients with very long K: the same-address atomic chain is then splitk/fix_group long instead of splitk -- contended
// fp32 atomics cost ~0.13 us EACH on this part, which is what made 128-way atomic split-K slow).
// Cross-XCD visibility without flushing L2s: partials move with agent-scope (write-through / L2-bypassing) stores and loads,
// the counter is an agent-scope atomic; an agent-scope fence here would write back and invalidate the whole L2 per workgroup
// (measured 4x slower than no split at all).
template <int TM, int TN>
__device__ __forceinline__ bool splitk_fixup(const GemmDev& p, f32x16 (&acc)[TM][TN], int tile, int ks, int& grp) {
    const int G = p.fix_group, tid = threadIdx.x;
    grp = ks / G;
    const int g0 = grp * G, gm = min(G, p.splitk - g0);
    if (gm == 1) return true;
    constexpr int NV = TM * TN * 16;
    float* part = p.ws_part + ((long long)tile * p.splitk + ks) * (NV * 256) + tid;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __hip_atomic_store(part + ((i * TN + j) * 16 + r) * 256, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);                          // vmcnt(0): THIS thread's write-through stores have been acknowledged
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // (the barrier alone does not wait for stores in flight)
    __syncthreads();
    __shared__ int s_last;
    if (tid == 0) {
        int* c = p.ws_cnt + tile * p.fix_ngroups + grp;
        const int old = atomicAdd(c, 1);
        s_last = (old == gm - 1);
        if (s_last) atomicExch(c, 0);                    // the workspace is left clean for the next launch on this stream
    }
    __syncthreads();
    if (!s_last) return false;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float* q = p.ws_part + ((long long)tile * p.splitk + g0) * (NV * 256) + tid;
#pragma unroll 4
    for (int s_ = 0; s_ < gm; ++s_, q += NV * 256) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[i][j][r] += __hip_atomic_load(q + ((i * TN + j) * 16 + r) * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return true;
}

// ---------------------------------------------------------------------------------------------- shared epilogue
// The MFMAs are issued with swapped operands (D^T = B^T A^T), so a lane owns ONE output row m = lane&31 of its 32x32
// block and register r holds column n = (r&3) + 8*(r>>2) + 4*(lane>>5): four consecutive columns per register group,
// i.e. 8-byte (bf16) / 16-byte (fp32) row-major stores instead of 2-byte ones.
template <typename T, typename TC, int TM, int TN>
__device__ __forceinline__ void epilogue_rows(const GemmDev& p, f32x16 (&acc)[TM][TN], int b1, int b2, bool first_split, bool atomic, int mbase, int nbase, int lane) {
    TC* C = reinterpret_cast<TC*>(p.C) + b1 * p.sC1 + b2 * p.sC2;
    const T* R = p.R ? reinterpret_cast<const T*>(p.R) + b1 * p.sR1 + b2 * p.sR2 : nullptr;
    const T* bias = p.bias ? reinterpret_cast<const T*>(p.bias) + b1 * p.sBias1 : nullptr;
    const int h = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = mbase + i * 32 + (lane & 31);
        if (row >= p.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = nbase + j * 32 + 8 * g + 4 * h;
                if (col >= p.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (p.act == TC_ACT_SCALE ? 1.0f : p.alpha) * acc[i][j][4 * g + e];
                TC* c = C + (long long)row * p.ldc + col;
                if (col + 3 < p.N && p.vecC) {
                    if (bias && first_split) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += ldf<T>(bias + col + e);
                    }
                    if (R && first_split) {
                        const float4 r4 = ld4<T>(R + (long long)row * p.ldr + col);
                        v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                    }
                    if (atomic) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) atomicAdd(reinterpret_cast<float*>(c) + e, v[e]);
                    } else {
                        if (p.act == TC_ACT_SIGMOID) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = sigmoid_f(v[e]);
                        } else if (p.act == TC_ACT_SCALE) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
                        }
                        if (p.accumulate) { const float4 o = ld4<TC>(c); v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w; }
                        st4<TC>(c, make_float4(v[0], v[1], v[2], v[3]));
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (col + e >= p.N) continue;
                        float w = v[e];
                        if (bias && first_split) w += ldf<T>(bias + col + e);
                        if (R && first_split) w += ldf<T>(R + (long long)row * p.ldr + col + e);
                        if (atomic) atomicAdd(reinterpret_cast<float*>(c) + e, w);
                        else {
                            if (p.act == TC_ACT_SIGMOID) w = sigmoid_f(w);
                            else if (p.act == TC_ACT_SCALE) w *= p.alpha;
                            if (p.accumulate) w += ldf<TC>(c + e);
                            stf<TC>(c + e, w);
                        }
                    }
                }
            }
    }
}

// Un-swapped orientation (fp32 C: weight gradients, fp32 storage): register r holds row (r&3) + 8*(r>>2) + 4*(lane>>5), the
// 32 lanes of a half-wave hold 32 consecutive columns -> 128-byte coalesced fp32 stores / atomics.
// EP (fp32 storage only): TC_FFN_EP -- the value becomes gp = v * GELU'(u) and the row sums of gp*gamma, gp*gamma*xhat over this
// workgroup's 64 columns go to ffn.part2[row][ntile]; srow = LDS float2 [2][64] (the two column halves of the tile), mloc0 = first
// tile row of this wave.
template <typename T, typename TC, int TM, int TN, bool EP = false>
__device__ __forceinline__ void epilogue_cols(const GemmDev& p, f32x16 (&acc)[TM][TN], int b1, int b2, bool first_split, bool atomic, int mbase, int nbase, int lane,
                                              float2* srow = nullptr, int mloc0 = 0, int wc = 0, int m0 = 0, int ntile = 0, int ntiles = 0) {
    TC* C = reinterpret_cast<TC*>(p.C) + b1 * p.sC1 + b2 * p.sC2;
    const T* R = p.R ? reinterpret_cast<const T*>(p.R) + b1 * p.sR1 + b2 * p.sR2 : nullptr;
    const T* bias = p.bias ? reinterpret_cast<const T*>(p.bias) + b1 * p.sBias1 : nullptr;
    const long long rbase = EP ? (long long)b1 * p.ffn.sRow1 : 0;
    const T* dmap = EP ? reinterpret_cast<const T*>(p.ffn.d) + rbase * p.ffn.ldd : nullptr;
    const float2* stat = EP ? reinterpret_cast<const float2*>(p.ffn.stat) + rbase : nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        float s1[16], s2[16];
        if constexpr (EP) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s1[r] = s2[r] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = nbase + j * 32 + (lane & 31);
            if (col >= p.N) continue;
            const float bv = (bias && first_split) ? ldf<T>(bias + col) : 0.f;
            float gm = 0.f, bt = 0.f;
            if constexpr (EP) {
                gm = ldf<T>(reinterpret_cast<const T*>(p.ffn.gamma) + (long long)b1 * p.ffn.sPar1 + col);
                bt = ldf<T>(reinterpret_cast<const T*>(p.ffn.beta) + (long long)b1 * p.ffn.sPar1 + col);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= p.M) continue;
                float v = (p.act == TC_ACT_SCALE ? 1.0f : p.alpha) * acc[i][j][r] + bv;
                if (R && first_split) v += ldf<T>(R + (long long)row * p.ldr + col);
                if constexpr (EP) {
                    const float2 st = stat[row];
                    v = ffn_ep1<T>(v, ldf<T>(dmap + (long long)row * p.ffn.ldd + col), st.x, st.y, gm, bt, s1[r], s2[r]);
                }
                TC* c = C + (long long)row * p.ldc + col;
                if (atomic) {
#ifndef TC_DBG_GEMM_NOATOMIC                                       // what-if build (scripts/exp): the weight-gradient atomics dropped, timing only
                    atomicAdd(reinterpret_cast<float*>(c), v);
#endif
                } else {
                    if (p.act == TC_ACT_SIGMOID) v = sigmoid_f(v);
                    else if (p.act == TC_ACT_SCALE) v *= p.alpha;
                    if (p.accumulate) v += ldf<TC>(c);
                    stf<TC>(c, v);
                }
            }
        }
        if constexpr (EP) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s1[r] = tc_group_sum<32>(s1[r]); s2[r] = tc_group_sum<32>(s2[r]);
                if ((lane & 31) == 0) srow[wc * 64 + mloc0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] = make_float2(s1[r], s2[r]);
            }
        }
    }
    if constexpr (EP) {
        __syncthreads();
        const int t = threadIdx.x;
        if (t < 64 && m0 + t < p.M) {
            const float2 a = srow[t], b = srow[64 + t];
            reinterpret_cast<float2*>(p.ffn.part2)[(rbase + m0 + t) * ntiles + ntile] = make_float2(a.x + b.x, a.y + b.y);
        }
    }
}

// bf16 C, plain store (no accumulate): values are finished in registers (alpha, bias, residual, activation) exactly as
// epilogue_rows does, rounded to bf16 once, parked in LDS as the row-major tile and written with 16-byte stores -- 16 consecutive
// lanes cover one 256-byte row of a 128-wide tile.  (epilogue_rows' direct 8-byte stores touch 32 different lines per
// instruction; most GEMMs of this model have K <= 512, so the epilogue is a large part of their time.)
template <typename H, int BM, int BN, int TM, int TN>
__device__ __forceinline__ void epilogue_rows_lds(const GemmDev& p, f32x16 (&acc)[TM][TN], int b1, int b2, int m0, int n0, int wr, int wc,
                                                  int lane, bf16_t* stage_raw) {
    // (requesting the residual tile and the bias before the K loop and parking them in the staging tile was tried: the sampled
    //  workgroups' epilogue got shorter, the step did not -- 14.64 ms either way: other resident workgroups already cover the wait)
    H* stage = reinterpret_cast<H*>(stage_raw);
    constexpr int LDS_ = BN + 8, WM = BM / 2, WN = BN / 2;
    const H* R = p.R ? reinterpret_cast<const H*>(p.R) + b1 * p.sR1 + b2 * p.sR2 : nullptr;
    const H* bias = p.bias ? reinterpret_cast<const H*>(p.bias) + b1 * p.sBias1 : nullptr;
    const int h = lane >> 5;
    __syncthreads();                                         // every wave is done with the operand tiles this overwrites
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int lr = wr * WM + i * 32 + (lane & 31), row = m0 + lr;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int lc = wc * WN + j * 32 + 8 * g + 4 * h, col = n0 + lc;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (p.act == TC_ACT_SCALE ? 1.0f : p.alpha) * acc[i][j][4 * g + e];
                if (row < p.M && col < p.N) {                // N % 8 == 0 on this path: the 4-group is all in or all out
                    if (bias) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += ldf<H>(bias + col + e);
                    }
                    if (R) {
                        const float4 r4 = ld4<H>(R + (long long)row * p.ldr + col);
                        v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                    }
                    if (p.act == TC_ACT_SIGMOID) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = sigmoid_f(v[e]);
                    } else if (p.act == TC_ACT_SCALE) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
                    }
                }
                st4<H>(stage + lr * LDS_ + lc, make_float4(v[0], v[1], v[2], v[3]));
            }
    }
    __syncthreads();
    H* C = reinterpret_cast<H*>(p.C) + b1 * p.sC1 + b2 * p.sC2;
    constexpr int CPR = BN / 8;                              // 16-byte chunks per tile row
#pragma unroll
    for (int it = 0; it < BM * CPR / 256; ++it) {
        const int idx = threadIdx.x + it * 256, lr = idx / CPR, lc = (idx - lr * CPR) * 8;
        const int row = m0 + lr, col = n0 + lc;
        if (row < p.M && col < p.N)
            *reinterpret_cast<uint4*>(C + (long long)row * p.ldc + col) = *reinterpret_cast<const uint4*>(stage + lr * LDS_ + lc);
    }
    if constexpr (BM == 64 && BN == 64) {
        if (float* const bnp = p.bn_part()) {
            // the statistics pass of the BatchNorm that follows, on the ROUNDED tile as it lies in LDS: thread (column, row quarter)
            // adds 16 rows, the quarters meet behind the staging tile, one thread per column writes this row tile's two sums
            float* red = reinterpret_cast<float*>(stage_raw + BM * LDS_);      // [2][4][64] floats behind the 64 x 72 tile
            const int cc = threadIdx.x & 63, part = threadIdx.x >> 6, col = n0 + cc;
            const float sh = (p.bn_shift() && col < p.N) ? p.bn_shift()[col] : 0.f;
            float s1 = 0.f, s2 = 0.f;
            if (col < p.N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = part * 16 + r;
                    const float v = (m0 + lr < p.M) ? ldf<H>(stage + lr * LDS_ + cc) - sh : 0.f;
                    s1 += v; s2 += v * v;
                }
            }
            red[part * 64 + cc] = s1; red[256 + part * 64 + cc] = s2;
            __syncthreads();
            if (threadIdx.x < 64 && col < p.N) {
                const int T = (p.M + BM - 1) / BM, t = m0 / BM;
                bnp[p.N + (long long)t * p.N + col] = red[cc] + red[64 + cc] + red[128 + cc] + red[192 + cc];
                bnp[p.N + (long long)(T + t) * p.N + col] = red[256 + cc] + red[320 + cc] + red[384 + cc] + red[448 + cc];
                if (t == 0) bnp[col] = sh;
            }
        }
    }
}

// TC_FFN_EP form of the epilogue above: the d tile (prefetched before the K loop, rd) is parked beside the C staging tile, every
// value becomes gp = v * GELU'(u) and the row sums for the LayerNorm backward leave through ffn.part2[row][bx].
template <typename H, int BM, int BN, int TM, int TN>
__device__ __forceinline__ void epilogue_rows_lds_ep(const GemmDev& p, f32x16 (&acc)[TM][TN], int b1, int m0, int n0, int wr, int wc, int lane,
                                                     bf16_t* stage_raw, const uint4 rd0, const uint4 rd1, int bx, int gx) {
    constexpr int LDS_ = BN + 8, WM = BM / 2, WN = BN / 2, CPR = BN / 8;
    H* stage = reinterpret_cast<H*>(stage_raw);
    H* dt = stage + BM * LDS_;
    float2* srow = reinterpret_cast<float2*>(dt + BM * LDS_);          // [2][BM]
    const long long rbase = (long long)b1 * p.ffn.sRow1;
    const H* gam = reinterpret_cast<const H*>(p.ffn.gamma) + (long long)b1 * p.ffn.sPar1;
    const H* bet = reinterpret_cast<const H*>(p.ffn.beta) + (long long)b1 * p.ffn.sPar1;
    const float2* stat = reinterpret_cast<const float2*>(p.ffn.stat) + rbase;
    const int h = lane >> 5;
    __syncthreads();                                         // every wave is done with the operand tiles this overwrites
    {
        const int lr = threadIdx.x / CPR, lc = (threadIdx.x % CPR) * 8;
        *reinterpret_cast<uint4*>(dt + lr * LDS_ + lc) = rd0;
        *reinterpret_cast<uint4*>(dt + (lr + 256 / CPR) * LDS_ + lc) = rd1;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int lr = wr * WM + i * 32 + (lane & 31), row = m0 + lr;
        const float2 st = stat[row < p.M ? row : 0];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int lc = wc * WN + j * 32 + 8 * g + 4 * h, col = n0 + lc;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = p.alpha * acc[i][j][4 * g + e];
                if (row < p.M && col < p.N) {                // N % 8 == 0 on this path: the 4-group is all in or all out
                    const float4 d4 = ld4<H>(dt + lr * LDS_ + lc), g4 = ld4<H>(gam + col), b4 = ld4<H>(bet + col);
                    {
                        const tc_f32x2 va = {v[0], v[1]}, vb = {v[2], v[3]};
                        const tc_f32x2 ga = {g4.x, g4.y}, gb = {g4.z, g4.w};
                        const tc_f32x2 xa = (tc_f32x2{d4.x, d4.y} - st.x) * st.y, xb = (tc_f32x2{d4.z, d4.w} - st.x) * st.y;
                        const tc_f32x2 pa = va * gelu_grad_f2_fast(xa * ga + tc_f32x2{b4.x, b4.y});      // (the 16-bit kernel)
                        const tc_f32x2 pb = vb * gelu_grad_f2_fast(xb * gb + tc_f32x2{b4.z, b4.w});
                        const tc_f32x2 qa = pa * ga, qb = pb * gb, ra = qa * xa, rb = qb * xb;
                        s1 += (qa.x + qa.y) + (qb.x + qb.y);
                        s2 += (ra.x + ra.y) + (rb.x + rb.y);
                        v[0] = pa.x; v[1] = pa.y; v[2] = pb.x; v[3] = pb.y;
                    }
                }
                st4<H>(stage + lr * LDS_ + lc, make_float4(v[0], v[1], v[2], v[3]));
            }
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if (h == 0) srow[wc * BM + lr] = make_float2(s1, s2);
    }
    __syncthreads();
    H* C = reinterpret_cast<H*>(p.C) + b1 * p.sC1;
#pragma unroll
    for (int it = 0; it < BM * CPR / 256; ++it) {
        const int idx = threadIdx.x + it * 256, lr = idx / CPR, lc = (idx - lr * CPR) * 8;
        const int row = m0 + lr, col = n0 + lc;
        if (row < p.M && col < p.N)
            *reinterpret_cast<uint4*>(C + (long long)row * p.ldc + col) = *reinterpret_cast<const uint4*>(stage + lr * LDS_ + lc);
    }
    if (threadIdx.x < BM && m0 + threadIdx.x < p.M) {
        const float2 a = srow[threadIdx.x], b = srow[BM + threadIdx.x];
        reinterpret_cast<float2*>(p.ffn.part2)[(rbase + m0 + threadIdx.x) * gx + bx] = make_float2(a.x + b.x, a.y + b.y);
    }
}

// ---------------------------------------------------------------------------------------------- fp32 path
// Load a 4-wide strip of an operand tile.  `trans` = the operand is stored [K, X] (X contiguous).
__device__ __forceinline__ float4 load_strip(const float* base, int ld, int x, int k, int X, int K, bool trans, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!trans) {
        if (x < X) {
            const float* p = base + (long long)x * ld + k;
            if (vec && k + 3 < K) v = *reinterpret_cast<const float4*>(p);
            else {
                if (k + 0 < K) v.x = p[0];
                if (k + 1 < K) v.y = p[1];
                if (k + 2 < K) v.z = p[2];
                if (k + 3 < K) v.w = p[3];
            }
        }
    } else {
        if (k < K) {
            const float* p = base + (long long)k * ld + x;
            if (vec && x + 3 < X) v = *reinterpret_cast<const float4*>(p);
            else {
                if (x + 0 < X) v.x = p[0];
                if (x + 1 < X) v.y = p[1];
                if (x + 2 < X) v.z = p[2];
                if (x + 3 < X) v.w = p[3];
            }
        }
    }
    return v;
}

template <typename TC, int BM, int BN, bool TA, bool TB, int FFN = 0>
__global__ __launch_bounds__(256) void gemm_kernel(GemmDev p) {
    constexpr int BK = 16, LDT = BK + 1;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int SA = BM * BK / 4 / 256, SB = BN * BK / 4 / 256;   // strips per thread
    static_assert(SA >= 1 && SB >= 1, "tile too small");
    static_assert(FFN == 0 || (BM == 64 && BN == 64), "the MixFFN hooks run on 64x64 tiles");
    static_assert(FFN != TC_FFN_LN_A || !TA, "LN_A transforms rows of a row-major A");
    static_assert(FFN != TC_FFN_LN_B || !TB, "LN_B transforms rows of a [K,N] B");
    __shared__ float As[BM * LDT];
    __shared__ float Bs[BN * LDT];
    __shared__ float2 s_ffn[FFN ? 2 * 64 : 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    int z = blockIdx.z;
    const int ks = z % p.splitk; z /= p.splitk;
    const int b2 = z % p.nb2, b1 = z / p.nb2;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = ks * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);

    const float* A = reinterpret_cast<const float*>(p.A) + b1 * p.sA1 + b2 * p.sA2;
    const float* B = reinterpret_cast<const float*>(p.B) + b1 * p.sB1 + b2 * p.sB2;
    const float* gam = FFN ? reinterpret_cast<const float*>(p.ffn.gamma) + (long long)b1 * p.ffn.sPar1 : nullptr;
    const float* bet = FFN ? reinterpret_cast<const float*>(p.ffn.beta) + (long long)b1 * p.ffn.sPar1 : nullptr;
    const float2* fstat = FFN ? reinterpret_cast<const float2*>(p.ffn.stat) + (long long)b1 * p.ffn.sRow1 : nullptr;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // hook state: LN_A -- this thread's strip row statistics (fixed for the whole K loop), gamma/beta of the slab's k range arrive with
    // the slab; LN_B -- gamma/beta of the strip's fixed columns, the statistics of the slab's token rows arrive with the slab
    float a_mean = 0.f, a_rstd = 0.f;
    float4 hg = make_float4(0.f, 0.f, 0.f, 0.f), hb = hg;     // per slab (LN_A) or fixed (LN_B)
    float2 hst = make_float2(0.f, 0.f);                        // LN_B: (mean, rstd) of the slab's token row
    if constexpr (FFN == TC_FFN_LN_A) {
        ffn_finalize_stats<BM>(p, b1, m0, blockIdx.x == 0 && ks == 0, s_ffn);
        const float2 st = s_ffn[tid / (BK / 4)];
        a_mean = st.x; a_rstd = st.y;
    }
    if constexpr (FFN == TC_FFN_LN_B) {
        const int col = n0 + (tid % (BN / 4)) * 4;
        if (col + 3 < p.N) { hg = *reinterpret_cast<const float4*>(gam + col); hb = *reinterpret_cast<const float4*>(bet + col); }
    }

    float4 ra[SA], rb[SB];
    auto fetch = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < SA; ++i) {
            const int f = tid + i * 256;
            if (!TA) { const int row = f / (BK / 4), kq = f % (BK / 4);
                ra[i] = load_strip(A, p.lda, m0 + row, k0 + kq * 4, p.M, kend, false, p.vecA); }
            else { const int k = f / (BM / 4), mq = f % (BM / 4);
                ra[i] = load_strip(A, p.lda, m0 + mq * 4, k0 + k, p.M, kend, true, p.vecA); }
        }
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            const int f = tid + i * 256;
            if (TB) { const int row = f / (BK / 4), kq = f % (BK / 4);          // stored [N,K]
                rb[i] = load_strip(B, p.ldb, n0 + row, k0 + kq * 4, p.N, kend, false, p.vecB); }
            else { const int k = f / (BN / 4), nq = f % (BN / 4);               // stored [K,N] (possibly in gapped row blocks)
                const float* Bk = p.bgap_every ? B + (long long)(k0 / p.bgap_every) * p.bgap : B;
                rb[i] = load_strip(Bk, p.ldb, n0 + nq * 4, k0 + k, p.N, kend, true, p.vecB); }
        }
        if constexpr (FFN == TC_FFN_LN_A) {
            const int k = k0 + (tid % (BK / 4)) * 4;
            if (k + 3 < kend) { hg = *reinterpret_cast<const float4*>(gam + k); hb = *reinterpret_cast<const float4*>(bet + k); }
        }
        if constexpr (FFN == TC_FFN_LN_B) {
            const int k = k0 + tid / (BN / 4);
            if (k < kend) hst = fstat[k];
        }
    };
    auto stage = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < SA; ++i) {
            const int f = tid + i * 256;
            if (!TA) { const int row = f / (BK / 4), kq = f % (BK / 4); float* d = &As[row * LDT + kq * 4];
                float4 v = ra[i];
                if constexpr (FFN == TC_FFN_LN_A) {          // (K is a multiple of 4 here: a strip is all in or all out)
                    if (m0 + row < p.M && k0 + kq * 4 + 3 < kend) {
                        v = ffn_ln_gelu4(v, a_mean, a_rstd, hg, hb);
                        if (p.ffn.d && blockIdx.x == 0)
                            *reinterpret_cast<float4*>(reinterpret_cast<float*>(const_cast<void*>(p.ffn.d)) +
                                                       ((long long)b1 * p.ffn.sRow1 + m0 + row) * p.ffn.ldd + k0 + kq * 4) = v;
                    }
                }
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }
            else { const int k = f / (BM / 4), mq = f % (BM / 4); float* d = &As[(mq * 4) * LDT + k];
                d[0] = ra[i].x; d[LDT] = ra[i].y; d[2 * LDT] = ra[i].z; d[3 * LDT] = ra[i].w; }
        }
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            const int f = tid + i * 256;
            if (TB) { const int row = f / (BK / 4), kq = f % (BK / 4); float* d = &Bs[row * LDT + kq * 4];
                d[0] = rb[i].x; d[1] = rb[i].y; d[2] = rb[i].z; d[3] = rb[i].w; }
            else { const int k = f / (BN / 4), nq = f % (BN / 4); float* d = &Bs[(nq * 4) * LDT + k];
                float4 v = rb[i];
                if constexpr (FFN == TC_FFN_LN_B) {
                    if (k0 + k < kend && n0 + nq * 4 + 3 < p.N) v = ffn_ln_gelu4(v, hst.x, hst.y, hg, hb);
                }
                d[0] = v.x; d[LDT] = v.y; d[2 * LDT] = v.z; d[3 * LDT] = v.w; }
        }
    };

    float rsum = 0.f;
    const bool do_rowsum = p.rowsum && blockIdx.x == 0 && tid < BM;
    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        stage(k0);
        __syncthreads();
        if (k0 + BK < kend) fetch(k0 + BK);
        if (do_rowsum) {
#pragma unroll
            for (int kk = 0; kk < BK; ++kk) rsum += As[tid * LDT + kk];
        }
        const float* ap = &As[(wr * WM + (lane & 31)) * LDT + (lane >> 5)];
        const float* bp = &Bs[(wc * WN + (lane & 31)) * LDT + (lane >> 5)];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = ap[i * 32 * LDT + kk];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = bp[j * 32 * LDT + kk];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    if (do_rowsum && m0 + tid < p.M) atomicAdd(p.rowsum + b1 * p.sRow1 + m0 + tid, rsum);
    bool first = (ks == 0), atomic = p.atomic;
    if (p.fix_group) {
        int grp;
        if (!splitk_fixup<TM, TN>(p, acc, (int)(((blockIdx.z / p.splitk) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x), ks, grp)) return;
        first = (grp == 0); atomic = p.fix_ngroups > 1;
    }
    if constexpr (FFN == TC_FFN_EP)
        epilogue_cols<float, TC, TM, TN, true>(p, acc, b1, b2, first, atomic, m0 + wr * WM, n0 + wc * WN, lane, s_ffn, wr * WM, wc, m0,
                                               (int)blockIdx.x, (int)gridDim.x);
    else
        epilogue_cols<float, TC, TM, TN>(p, acc, b1, b2, first, atomic, m0 + wr * WM, n0 + wc * WN, lane);
}

// ---------------------------------------------------------------------------------------------- bf16 path
// Load an 8-wide strip (16 B) of an operand tile as raw bf16 bits.
// Operand strips (8 bf16 along K, or along M/N for a transposed operand).  The common case -- an aligned operand and a strip
// wholly inside the matrix -- must not sit behind a per-lane branch, and its data must not be touched before it is needed: the
// compiler closes every divergent region that contains a load with s_waitcnt vmcnt(0), and a select on the loaded value waits
// for it as well; either turns the strips of one slab into as many sequential memory round trips (measured: 1.2-1.7 us per
// 64-deep slab).  So strip_raw() issues the 16-byte load unconditionally (strips not wholly inside read the operand's first 16
// bytes, always valid and aligned) and the zeroing of what lies outside happens when the slab is written to LDS, a slab or two
// later.  A workgroup with partially covered strips (K / M / N tails not a multiple of 8) or unaligned operands runs the
// general loop instead, which resolves every strip at fetch time.
__device__ __forceinline__ bool strip_whole(int x, int k, int X, int K, bool trans) {
    return trans ? (k < K && x + 7 < X) : (x < X && k + 7 < K);
}
__device__ __forceinline__ uint4 strip_raw(const bf16_t* base, const bf16_t* safe, int ld, int x, int k, int X, int K, bool trans) {
    const long long off = trans ? (long long)k * ld + x : (long long)x * ld + k;
    return *reinterpret_cast<const uint4*>(strip_whole(x, k, X, K, trans) ? base + off : safe);
}
__device__ __forceinline__ uint4 strip_tail(const bf16_t* base, int ld, int x, int k, int X, int K, bool trans) {
    union { uint4 v; bf16_t e[8]; } u;
    u.v = make_uint4(0u, 0u, 0u, 0u);
    const bf16_t* p = base + (trans ? (long long)k * ld + x : (long long)x * ld + k);
#pragma unroll
    for (int i = 0; i < 8; ++i) if (trans ? (x + i < X) : (k + i < K)) u.e[i] = p[i];
    return u.v;
}
// general path (rare)
__device__ __forceinline__ uint4 load_strip8(const bf16_t* base, int ld, int x, int k, int X, int K, bool trans, bool vec) {
    const bool in = trans ? (k < K) : (x < X);
    if (!in) return make_uint4(0u, 0u, 0u, 0u);
    if (vec && strip_whole(x, k, X, K, trans))
        return *reinterpret_cast<const uint4*>(base + (trans ? (long long)k * ld + x : (long long)x * ld + k));
    return strip_tail(base, ld, x, k, X, K, trans);
}

// Operands whose global layout is [K][X] (transposed: A of a weight-gradient product, B of an input-gradient product) are kept in
// LDS as they come, [64][X+8], one 16-byte store per strip, and their MFMA fragments are gathered with the hardware transpose
// read (ds_read_b64_tr_b16: lane i of a 16-lane group hands in the address of row i>>2 of a 4-row k block, columns 4(i&3).., and
// receives column i of the block).  A transpose read touches 4 k-rows x 32 columns per 32 lanes; with 144-byte rows (36 banks)
// rows p, p+4, p+8, p+12 start 16 banks apart, so the k-rows of every 16-k chunk are stored 4x4-transposed (gemm_krow) and each
// read is conflict-free at the LDS footprint of the K-contiguous layout.  (Before: eight 2-byte LDS stores per strip.)
__device__ __forceinline__ int gemm_krow(int k) { return (k & ~15) | ((k & 3) << 2) | ((k >> 2) & 3); }
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
template <typename V8> __device__ __forceinline__ V8 ld_frag_tr(const bf16_t* lo, const bf16_t* hi) {
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(lo));
    const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(hi));
    return __builtin_bit_cast(V8, (s16x8_t)__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

// Body of one workgroup of the bf16 GEMM: (bx, by, bz) of a (gx, gy, *) grid.  The LDS buffers come from the caller so that the
// two problems of a paired launch (gemm_pair_kernel) share one allocation.
#ifdef TC_GEMM_TIMING
// phase stamps of the bf16 GEMM body (experiment builds only): every 61st workgroup of a launch writes its own row of the table
// (kind = TA*4 + TB*2 + fp32-out + 8*FFN, cycles of setup / K loop / epilogue) -- plain stores, no atomics (same-address atomics
// from every workgroup serialise at ~0.13 us each and were what a first version measured)
__device__ unsigned long long g_gemm_dbg[1024 * 8];
__device__ unsigned int g_gemm_dbg_n;
#define GSTAMP(k) do { if (gs_ >= 0) { const long long t_ = __builtin_readcyclecounter(); g_gemm_dbg[gs_ * 8 + 1 + (k)] = (unsigned long long)(t_ - gt_); gt_ = t_; } } while (0)
#define GSTAMP_INIT() int gs_ = -1; long long gt_ = 0; \
    if (threadIdx.x == 0 && (blockIdx.x + 7 * blockIdx.y + 13 * blockIdx.z) % 61 == 0) { gs_ = (int)(atomicAdd(&g_gemm_dbg_n, 1u) & 1023u); \
        g_gemm_dbg[gs_ * 8] = (TA ? 4 : 0) + (TB ? 2 : 0) + (sizeof(TC) == 4 ? 1 : 0) + 8 * FFN + 1000ull * (unsigned long long)((kend_for_stamp)); gt_ = __builtin_readcyclecounter(); }
#else
#define GSTAMP(k)
#define GSTAMP_INIT()
#endif
template <typename H, typename TC, int BM, int BN, bool TA, bool TB, bool DB, int FFN = 0>
__device__ __forceinline__ void gemm_bf16_body(const GemmDev& p, const int bx, const int by, const int bz, const int gx, const int gy,
                                               bf16_t (*As)[BM * (64 + 8)], bf16_t (*Bs)[BN * (64 + 8)]) {
    constexpr int BK = 64, LDT = BK + 8;                       // 144-byte rows: 16-B aligned, conflict-free b128 fragment reads
    static_assert(FFN == 0 || (BM == 64 && BN == 64 && DB), "the MixFFN hooks run on 64x64 tiles with the double-buffered LDS");
    static_assert(FFN != TC_FFN_LN_A || !TA, "LN_A transforms rows of a row-major A");
    static_assert(FFN != TC_FFN_LN_B || !TB, "LN_B transforms rows of a [K,N] B");
    static_assert(FFN != TC_FFN_EP || sizeof(TC) == 2, "EP writes a bf16 C");
    constexpr bool SWAP = sizeof(TC) == 2;                     // bf16 output: lane owns a row (8-byte stores); fp32 output: coalesced columns
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int SA = BM * BK / 8 / 256, SB = BN * BK / 8 / 256;   // 8-element strips per thread
    static_assert(SA >= 1 && SB >= 1, "tile too small");
    // double-buffered LDS + two register sets: the loads of slab i+2 are in flight while slab i is multiplied and slab i+1
    // is written to the other LDS buffer -- one barrier per slab, two slabs of memory latency covered (small-M GEMMs of this
    // model run ~1 workgroup per CU, so the K loop is latency-bound, not bandwidth-bound)
    // (DB = false: K <= 2 slabs -- one LDS buffer, half the footprint, twice the resident workgroups)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    int z = bz;
    const int zq = fdiv(z, p.splitk, p.m_splitk);
    const int ks = z - zq * p.splitk; z = zq;
    const int b1 = fdiv(z, p.nb2, p.m_nb2), b2 = z - b1 * p.nb2;
    const int m0 = by * BM, n0 = bx * BN;
    const int kbeg = ks * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int kend_for_stamp = kend - kbeg; (void)kend_for_stamp;
    GSTAMP_INIT();

    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A) + b1 * p.sA1 + b2 * p.sA2;
    const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B) + b1 * p.sB1 + b2 * p.sB2;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // strip ownership: K-contiguous operands: consecutive threads walk along K (coalesced 16-B loads, vector LDS writes);
    // transposed operands: 8 consecutive threads cover one k-row of the tile (128 contiguous bytes), one 16-B LDS write each
    auto a_xy = [&](int i, int k0, int& x, int& k) __attribute__((always_inline)) {
        const int f = tid + i * 256;
        if (!TA) { x = m0 + f / (BK / 8); k = k0 + (f % (BK / 8)) * 8; }
        else { k = k0 + f / (BM / 8); x = m0 + (f % (BM / 8)) * 8; }         // 8 lanes = one k-row of the tile: coalesced
    };
    auto b_xy = [&](int i, int k0, int& x, int& k) __attribute__((always_inline)) {
        const int f = tid + i * 256;
        if (TB) { x = n0 + f / (BK / 8); k = k0 + (f % (BK / 8)) * 8; }
        else { k = k0 + f / (BN / 8); x = n0 + (f % (BN / 8)) * 8; }
    };
    auto b_base = [&](int k0) __attribute__((always_inline)) { return (!TB && p.bgap_every) ? B + (long long)(k0 / p.bgap_every) * p.bgap : B; };
    // MixFFN hooks.  LN_A: the statistics of this thread's two strip rows are fixed for the K loop (merged from the chunk partials
    // here), gamma / beta of a slab's k range travel with the slab (rh = {gamma, beta} raw).  LN_B: gamma / beta of the strip's
    // fixed columns are loaded once, the statistics of a slab's two token rows travel with the slab (rh[0] = {mean0, rstd0,
    // mean1, rstd1}).  EP: the d tile under this output tile is requested now and parked in LDS by the epilogue.
    const bf16_t* gam = FFN ? reinterpret_cast<const bf16_t*>(p.ffn.gamma) + (long long)b1 * p.ffn.sPar1 : nullptr;
    const bf16_t* bet = FFN ? reinterpret_cast<const bf16_t*>(p.ffn.beta) + (long long)b1 * p.ffn.sPar1 : nullptr;
    const float2* fstat = FFN ? reinterpret_cast<const float2*>(p.ffn.stat) + (long long)b1 * p.ffn.sRow1 : nullptr;
    __shared__ float2 s_ffn[FFN == TC_FFN_LN_A ? BM : 1];
    float a_mean[SA], a_rstd[SA], fg[8], fb[8];
    bf16_t* aout = (FFN == TC_FFN_LN_A && p.ffn.d && bx == 0) ? reinterpret_cast<bf16_t*>(const_cast<void*>(p.ffn.d)) + (long long)b1 * p.ffn.sRow1 * p.ffn.ldd
                                                               : nullptr;
    uint4 rd0 = make_uint4(0u, 0u, 0u, 0u), rd1 = rd0;         // EP: this thread's two 16-byte pieces of the d tile
    if constexpr (FFN == TC_FFN_LN_A) {
        ffn_finalize_stats<BM>(p, b1, m0, bx == 0 && ks == 0, s_ffn);
#pragma unroll
        for (int i = 0; i < SA; ++i) { const float2 st = s_ffn[(tid + i * 256) / (BK / 8)]; a_mean[i] = st.x; a_rstd[i] = st.y; }
    }
    if constexpr (FFN == TC_FFN_LN_B) {
        const int col = n0 + (tid % (BN / 8)) * 8;
        const int cc = col + 7 < p.N ? col : 0;
        bf8_unpack<H>(*reinterpret_cast<const uint4*>(gam + cc), fg);
        bf8_unpack<H>(*reinterpret_cast<const uint4*>(bet + cc), fb);
    }
    if constexpr (FFN == TC_FFN_EP) {
        const bf16_t* dmap = reinterpret_cast<const bf16_t*>(p.ffn.d) + (long long)b1 * p.ffn.sRow1 * p.ffn.ldd;
        static_assert(BM * BN / 8 / 256 == 2 || FFN != TC_FFN_EP, "two pieces per thread");
        const int lr = tid / (BN / 8), lc = (tid % (BN / 8)) * 8;
        const bool in0 = m0 + lr < p.M && n0 + lc < p.N, in1 = m0 + lr + 256 / (BN / 8) < p.M && n0 + lc < p.N;
        rd0 = *reinterpret_cast<const uint4*>(in0 ? dmap + (long long)(m0 + lr) * p.ffn.ldd + n0 + lc : dmap);
        rd1 = *reinterpret_cast<const uint4*>(in1 ? dmap + (long long)(m0 + lr + 256 / (BN / 8)) * p.ffn.ldd + n0 + lc : dmap);
    }
    // FAST (decided once per workgroup, below): both operands 16-byte aligned and no strip of this workgroup's slabs is partially
    // covered -- the K loop then contains no branch at all around its loads.  Otherwise: the general loader, everything at fetch time.
    auto fetch = [&](auto FT, uint4 (&ra)[SA], uint4 (&rb)[SB], uint4 (&rh)[2], int k0) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(FT)::value;
        int x, k;
#pragma unroll
        for (int i = 0; i < SA; ++i) {
            a_xy(i, k0, x, k);
            if constexpr (FAST) ra[i] = strip_raw(A, A, p.lda, x, k, p.M, kend, TA);
            else ra[i] = load_strip8(A, p.lda, x, k, p.M, kend, TA, p.vecA);
        }
        const bf16_t* Bk = b_base(k0);
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            b_xy(i, k0, x, k);
            if constexpr (FAST) rb[i] = strip_raw(Bk, B, p.ldb, x, k, p.N, kend, !TB);
            else rb[i] = load_strip8(Bk, p.ldb, x, k, p.N, kend, !TB, p.vecB);
        }
        if constexpr (FFN == TC_FFN_LN_A) {
            const int kk = k0 + (tid % (BK / 8)) * 8, kc = kk + 7 < kend ? kk : 0;
            rh[0] = *reinterpret_cast<const uint4*>(gam + kc);
            rh[1] = *reinterpret_cast<const uint4*>(bet + kc);
        }
        if constexpr (FFN == TC_FFN_LN_B) {
            const int t0 = k0 + tid / (BN / 8), t1 = t0 + 256 / (BN / 8);
            const float2 s0 = fstat[t0 < kend ? t0 : 0], s1 = fstat[t1 < kend ? t1 : 0];
            rh[0] = make_uint4(__float_as_uint(s0.x), __float_as_uint(s0.y), __float_as_uint(s1.x), __float_as_uint(s1.y));
        }
    };
    constexpr int PA = BM + 8, PB = BN + 8;                    // row pitch of a transposed-layout slab
    auto stage = [&](auto FT, const uint4 (&ra)[SA], const uint4 (&rb)[SB], const uint4 (&rh)[2], int k0, bf16_t* as, bf16_t* bs) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(FT)::value;
        int x, k;
        float hg[8], hb[8];
        if constexpr (FFN == TC_FFN_LN_A) { bf8_unpack<H>(rh[0], hg); bf8_unpack<H>(rh[1], hb); }
#pragma unroll
        for (int i = 0; i < SA; ++i) {
            const int f = tid + i * 256;
            a_xy(i, k0, x, k);
            uint4 v = ra[i];
            if constexpr (FFN == TC_FFN_LN_A) {              // (K % 8 == 0: strips are all in or all out)
                v = ffn_ln_gelu8<H>(v, a_mean[i], a_rstd[i], hg, hb);
                if (aout && strip_whole(x, k, p.M, kend, TA)) *reinterpret_cast<uint4*>(aout + (long long)x * p.ffn.ldd + k) = v;
            }
            if constexpr (FAST) { if (!strip_whole(x, k, p.M, kend, TA)) v = make_uint4(0u, 0u, 0u, 0u); }
            else if constexpr (FFN == TC_FFN_LN_A) { if (!strip_whole(x, k, p.M, kend, TA)) v = make_uint4(0u, 0u, 0u, 0u); }
            if (!TA) { const int row = f / (BK / 8), kq = f % (BK / 8); *reinterpret_cast<uint4*>(&as[row * LDT + kq * 8]) = v; }
            else { const int kl = f / (BM / 8), mq = f % (BM / 8); *reinterpret_cast<uint4*>(&as[gemm_krow(kl) * PA + mq * 8]) = v; }
        }
        const bf16_t* Bk = b_base(k0);
#pragma unroll
        for (int i = 0; i < SB; ++i) {
            const int f = tid + i * 256;
            b_xy(i, k0, x, k);
            uint4 v = rb[i];
            if constexpr (FFN == TC_FFN_LN_B) {
                static_assert(FFN != TC_FFN_LN_B || SB == 2, "two strips per thread");
                v = ffn_ln_gelu8<H>(v, __uint_as_float(i == 0 ? rh[0].x : rh[0].z), __uint_as_float(i == 0 ? rh[0].y : rh[0].w), fg, fb);
            }
            if constexpr (FAST) { if (!strip_whole(x, k, p.N, kend, !TB)) v = make_uint4(0u, 0u, 0u, 0u); }
            else if constexpr (FFN == TC_FFN_LN_B) { if (!strip_whole(x, k, p.N, kend, !TB)) v = make_uint4(0u, 0u, 0u, 0u); }
            if (TB) { const int row = f / (BK / 8), kq = f % (BK / 8); *reinterpret_cast<uint4*>(&bs[row * LDT + kq * 8]) = v; }
            else { const int kl = f / (BN / 8), nq = f % (BN / 8); *reinterpret_cast<uint4*>(&bs[gemm_krow(kl) * PB + nq * 8]) = v; }
        }
    };
    float rsum = 0.f;
    const bool do_rowsum = p.rowsum && bx == 0 && tid < BM;
    auto compute = [&](const bf16_t* as, const bf16_t* bs) __attribute__((always_inline)) {
        if (do_rowsum) {
#pragma unroll
            for (int kk = 0; kk < BK; ++kk) rsum += ldf<H>(reinterpret_cast<const H*>(TA ? as + kk * PA + tid : as + tid * LDT + kk));
        }
        // K-contiguous slab: lane (row lane&31, k-slice 8*(lane>>5)) reads 16 bytes.  Transposed-layout slab: k block j of a 16-k
        // chunk lies in rows j, j+4, j+8, j+12 of the chunk; the lane's k-slice 8h..8h+7 is blocks 2h and 2h+1.
        const int h = lane >> 5, gi = lane & 15, gq = (lane >> 4) & 1;
        const bf16_t* ap = TA ? &as[(2 * h + 4 * (gi >> 2)) * PA + wr * WM + 16 * gq + 4 * (gi & 3)] : &as[(wr * WM + (lane & 31)) * LDT + 8 * h];
        const bf16_t* bp = !TB ? &bs[(2 * h + 4 * (gi >> 2)) * PB + wc * WN + 16 * gq + 4 * (gi & 3)] : &bs[(wc * WN + (lane & 31)) * LDT + 8 * h];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
            typedef typename HalfOps<H>::v8 V8;
            V8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (TA) a[i] = ld_frag_tr<V8>(ap + kk * PA + i * 32, ap + (kk + 1) * PA + i * 32);
                else a[i] = *reinterpret_cast<const V8*>(ap + i * 32 * LDT + kk);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (!TB) b[j] = ld_frag_tr<V8>(bp + kk * PB + j * 32, bp + (kk + 1) * PB + j * 32);
                else b[j] = *reinterpret_cast<const V8*>(bp + j * 32 * LDT + kk);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = SWAP ? HalfOps<H>::mfma(b[j], a[i], acc[i][j])      // D^T: lane = output row
                                     : HalfOps<H>::mfma(a[i], b[j], acc[i][j]);
        }
    };

    auto kloop = [&](auto FT) __attribute__((always_inline)) {
    // FAST: prefetches are issued unconditionally (a slab beyond kend reads the operands' first bytes and is never used) -- a
    // conditional prefetch makes the compiler wait for ALL outstanding loads, the new slab's included, before the older slab is used
    constexpr bool FAST = decltype(FT)::value;
    uint4 ra0[SA], rb0[SB], ra1[SA], rb1[SB], rh0[2], rh1[2];
    if (!DB) {
        if (kbeg < kend) fetch(FT, ra0, rb0, rh0, kbeg);             // (one register set: a conditional prefetch costs nothing here)
        for (int k0 = kbeg; k0 < kend; k0 += BK) {
            stage(FT, ra0, rb0, rh0, k0, As[0], Bs[0]);
            __syncthreads();
            if (k0 + BK < kend) fetch(FT, ra0, rb0, rh0, k0 + BK);
            compute(As[0], Bs[0]);
            __syncthreads();
        }
    } else if (kbeg < kend) {
        fetch(FT, ra0, rb0, rh0, kbeg);
        if (FAST || kbeg + BK < kend) fetch(FT, ra1, rb1, rh1, kbeg + BK);
        stage(FT, ra0, rb0, rh0, kbeg, As[0], Bs[0]);
        __syncthreads();
        if (FAST || kbeg + 2 * BK < kend) fetch(FT, ra0, rb0, rh0, kbeg + 2 * BK);
        for (int k0 = kbeg;; k0 += 2 * BK) {
            // slab k0 sits in LDS buffer 0; slab k0+BK waits in register set 1; slab k0+2BK is arriving in set 0
            const bool has1 = k0 + BK < kend;
            if (has1) stage(FT, ra1, rb1, rh1, k0 + BK, As[DB ? 1 : 0], Bs[DB ? 1 : 0]);
            compute(As[0], Bs[0]);
            if (!has1) break;
            __syncthreads();
            if (FAST || k0 + 3 * BK < kend) fetch(FT, ra1, rb1, rh1, k0 + 3 * BK);
            const bool has2 = k0 + 2 * BK < kend;
            if (has2) stage(FT, ra0, rb0, rh0, k0 + 2 * BK, As[0], Bs[0]);
            compute(As[DB ? 1 : 0], Bs[DB ? 1 : 0]);
            if (!has2) break;
            __syncthreads();
            if (FAST || k0 + 4 * BK < kend) fetch(FT, ra0, rb0, rh0, k0 + 4 * BK);
        }
    }
    };
    // no partially covered strip anywhere in this workgroup's slabs: K range a multiple of 8 for K-contiguous operands, M / N a
    // multiple of 8 for operands whose strips run along M / N
    const bool fast = p.vecA && p.vecB && ((kend - kbeg) % 8 == 0 || (TA && !TB)) && (!TA || p.M % 8 == 0) && (TB || p.N % 8 == 0);
    GSTAMP(0);
    if (fast) kloop(std::true_type{}); else kloop(std::false_type{});
    GSTAMP(1);
    if (do_rowsum && m0 + tid < p.M) atomicAdd(p.rowsum + b1 * p.sRow1 + m0 + tid, rsum);
    bool first = (ks == 0), atomic = p.atomic;
    if (p.fix_group) {
        int grp;
        if (!splitk_fixup<TM, TN>(p, acc, (zq * gy + by) * gx + bx, ks, grp)) return;
        first = (grp == 0); atomic = p.fix_ngroups > 1;
    }
    if constexpr (FFN == TC_FFN_EP) {
        static_assert(FFN != TC_FFN_EP || 2 * BM * (BN + 8) + 2 * BM * 4 <= 2 * (BM + BN) * (64 + 8), "C stage + d tile + row sums must fit");
        epilogue_rows_lds_ep<H, BM, BN, TM, TN>(p, acc, b1, m0, n0, wr, wc, lane, &As[0][0], rd0, rd1, bx, gx);
    } else if constexpr (SWAP) {
        if (p.vec8C && !p.accumulate && !atomic && first) {
            static_assert((BM * (BN + 8)) <= (DB ? 2 : 1) * (BM + BN) * (64 + 8), "staging tile must fit the operand buffers");
            epilogue_rows_lds<H, BM, BN, TM, TN>(p, acc, b1, b2, m0, n0, wr, wc, lane, &As[0][0]);
        } else {
            epilogue_rows<H, TC, TM, TN>(p, acc, b1, b2, first, atomic, m0 + wr * WM, n0 + wc * WN, lane);
        }
    } else {
        epilogue_cols<H, TC, TM, TN>(p, acc, b1, b2, first, atomic, m0 + wr * WM, n0 + wc * WN, lane);
    }
    GSTAMP(2);
}

// XCD-aware tile numbering.  Workgroups are dealt to the 8 XCDs round-robin in dispatch order (x fastest), so the N-tiles of one
// M-tile -- which all read the same A rows -- land on different XCDs and every L2 fetches that A tile for itself (N = 1280: twenty
// times).  Here the l-th workgroup of an XCD takes the l-th tile of that XCD's contiguous share of the (m, n) tile list, n fastest:
// the readers of an A tile run back to back on one XCD.  A bijection for any tile count (the first T % 8 XCDs get one tile more).
__device__ __forceinline__ void tc_xcd_tile(int& bx, int& by, const int gx, const int gy, const unsigned mgx) {
    const int T = gx * gy, L = bx + gx * by;
    const int xcd = L & 7, slot = L >> 3, q = T >> 3, r = T & 7;
    const int t = xcd * q + min(xcd, r) + slot;
    by = fdiv(t, gx, mgx);
    bx = t - by * gx;
}

// resident workgroups per CU the 64x64 kernels are compiled for (89-90 VGPRs without spills; 6: 67-73 VGPRs, equal end to end; 8 spills:
// 14.54 vs 14.14 ms per step)
#ifndef GEMM_OCC64
#define GEMM_OCC64 5
#endif
template <typename H, typename TC, int BM, int BN, bool TA, bool TB, bool DB, int FFN = 0>
__global__ __launch_bounds__(256, (BM == 64 && BN == 64) ? GEMM_OCC64 : 2) void gemm_bf16_kernel(GemmDev p) {
    // ONE buffer: A slabs first, B slabs behind them -- the epilogue reuses it from the start as its C staging tile
    __shared__ __attribute__((aligned(16))) bf16_t smem[(DB ? 2 : 1) * (BM + BN) * (64 + 8)];
    int tbx = blockIdx.x, tby = blockIdx.y;
    if (p.xcd & 1) tc_xcd_tile(tbx, tby, gridDim.x, gridDim.y, p.m_gx);
    gemm_bf16_body<H, TC, BM, BN, TA, TB, DB, FFN>(p, tbx, tby, blockIdx.z, gridDim.x, gridDim.y,
                                                reinterpret_cast<bf16_t(*)[BM * (64 + 8)]>(smem),
                                                reinterpret_cast<bf16_t(*)[BN * (64 + 8)]>(smem + (DB ? 2 : 1) * BM * (64 + 8)));
}

// One launch for the two gradient GEMMs of a Linear: problem A = dX = dY W (row-major operands, bf16 out), problem B = dW = dY^T X
// (+ row sums = db; fp32 accumulate into the gradient arena).  Both are short, low-occupancy grids (64x64 tiles); side by side in
// one grid they fill the CUs together, where two launches on two streams mostly serialise on this part (measured: kernels from
// different HW queues overlap only at their tails) and pay two dependency gaps.
struct GemmPairDev { GemmDev a, b; int nA, gxA, gyA, gxB, gyB; unsigned mgxA, mgyA, mgxB, mgyB; };
template <typename H>
__global__ __launch_bounds__(256, 2) void gemm_pair_kernel(GemmPairDev q) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * (64 + 64) * (64 + 8)];
    bf16_t(*As)[64 * (64 + 8)] = reinterpret_cast<bf16_t(*)[64 * (64 + 8)]>(smem);
    bf16_t(*Bs)[64 * (64 + 8)] = reinterpret_cast<bf16_t(*)[64 * (64 + 8)]>(smem + 2 * 64 * (64 + 8));
    // the weight-gradient workgroups (long K ranges: 14-26 k cycles against 7-15 k for a dX tile, scripts/exp/gemm_timing.py) are
    // dispatched FIRST, the dX tiles fill in behind them -- the other way round the launch ended on late-started long workgroups
    const int nB = (int)gridDim.x - q.nA;
    int lin = (int)blockIdx.x < nB ? (int)blockIdx.x + q.nA : (int)blockIdx.x - nB;
    if (lin < q.nA) {
        const int l1 = fdiv(lin, q.gxA, q.mgxA);
        int bx = lin - l1 * q.gxA;
        const int bz = fdiv(l1, q.gyA, q.mgyA);
        int by = l1 - bz * q.gyA;
        if (q.a.xcd & 1) tc_xcd_tile(bx, by, q.gxA, q.gyA, q.mgxA);
        if (q.a.ffn.mode == TC_FFN_EP) gemm_bf16_body<H, H, 64, 64, false, false, true, TC_FFN_EP>(q.a, bx, by, bz, q.gxA, q.gyA, As, Bs);
        else gemm_bf16_body<H, H, 64, 64, false, false, true>(q.a, bx, by, bz, q.gxA, q.gyA, As, Bs);
    } else {
        lin -= q.nA;
        if (q.b.xcd & 2) {
            // the tiles of ONE K split read the same rows of dY and X: the split-major tile list is dealt to the XCDs in contiguous
            // shares (this workgroup's index IS blockIdx.x, the weight-gradient problem comes first), so a split's tiles share an L2
            const int xcd = lin & 7, slot = lin >> 3, qq = nB >> 3, r = nB & 7;
            lin = xcd * qq + min(xcd, r) + slot;
        }
        const int l1 = fdiv(lin, q.gxB, q.mgxB), bx = lin - l1 * q.gxB, bz = fdiv(l1, q.gyB, q.mgyB), by = l1 - bz * q.gyB;
        if (q.b.ffn.mode == TC_FFN_LN_B) gemm_bf16_body<H, float, 64, 64, true, false, true, TC_FFN_LN_B>(q.b, bx, by, bz, q.gxB, q.gyB, As, Bs);
        else gemm_bf16_body<H, float, 64, 64, true, false, true>(q.b, bx, by, bz, q.gxB, q.gyB, As, Bs);
    }
}

// Up to 12 independent bf16 GEMMs of the three kinds a Linear produces (y = x W^T: TA=0,TB=1 ; dX = dY W: TA=0,TB=0 ; dW = dY^T X:
// TA=1,TB=0, fp32 C) in one grid of 64x64 tiles -- the four per-scale MixFFNs of a bridge layer are independent chains of small
// GEMMs; level by level their workgroups share the CUs instead of queueing as 4 (forward) or 8 (backward) short launches.
constexpr int GEMM_MULTI_MAX = 12;
static_assert(sizeof(GemmDev) * GEMM_MULTI_MAX + 20 * GEMM_MULTI_MAX + 8 <= 4096, "kernel argument block");
struct GemmMultiDev { GemmDev p[GEMM_MULTI_MAX]; int blk0[GEMM_MULTI_MAX], gx[GEMM_MULTI_MAX], gy[GEMM_MULTI_MAX], kind[GEMM_MULTI_MAX]; unsigned mgy[GEMM_MULTI_MAX]; int n; };
template <typename H>
__global__ __launch_bounds__(256, 4) void gemm_multi_kernel(GemmMultiDev q) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * (64 + 64) * (64 + 8)];
    bf16_t(*As)[64 * (64 + 8)] = reinterpret_cast<bf16_t(*)[64 * (64 + 8)]>(smem);
    bf16_t(*Bs)[64 * (64 + 8)] = reinterpret_cast<bf16_t(*)[64 * (64 + 8)]>(smem + 2 * 64 * (64 + 8));
    int lin = blockIdx.x, i = 0;
    for (int j = 1; j < q.n; ++j) if (lin >= q.blk0[j]) i = j;
    lin -= q.blk0[i];
    if (q.p[i].xcd & 2) {                                    // split-K weight gradients: a split's tiles on one XCD (see gemm_pair_kernel)
        const int ni = (i + 1 < q.n ? q.blk0[i + 1] : (int)gridDim.x) - q.blk0[i];
        const int xcd = lin & 7, slot = lin >> 3, qq = ni >> 3, r = ni & 7;    // local indices l, l + 8, ... share an XCD at any offset
        lin = xcd * qq + min(xcd, r) + slot;
    }
    const int gx = q.gx[i], gy = q.gy[i];
    const unsigned mgx = q.p[i].m_gx;
    const int l1 = fdiv(lin, gx, mgx), bz = fdiv(l1, gy, q.mgy[i]);
    int bx = lin - l1 * gx, by = l1 - bz * gy;
    if ((q.p[i].xcd & 1) && q.kind[i] != 2 && q.kind[i] != 5) tc_xcd_tile(bx, by, gx, gy, mgx);   // (not the split-K weight gradients)
    // a private copy of the one descriptor: with six inlined bodies reading fields through a reference into the 4 KB argument block
    // the compiler stopped forwarding the loads to the kernel-argument segment and copied the whole block to scratch
    const GemmDev p = q.p[i];
    if (q.kind[i] == 0) gemm_bf16_body<H, H, 64, 64, false, true, true>(p, bx, by, bz, gx, gy, As, Bs);
    else if (q.kind[i] == 1) gemm_bf16_body<H, H, 64, 64, false, false, true>(p, bx, by, bz, gx, gy, As, Bs);
    else if (q.kind[i] == 2) gemm_bf16_body<H, float, 64, 64, true, false, true>(p, bx, by, bz, gx, gy, As, Bs);
    else if (q.kind[i] == 3) gemm_bf16_body<H, H, 64, 64, false, true, true, TC_FFN_LN_A>(p, bx, by, bz, gx, gy, As, Bs);
    else if (q.kind[i] == 4) gemm_bf16_body<H, H, 64, 64, false, false, true, TC_FFN_EP>(p, bx, by, bz, gx, gy, As, Bs);
    else gemm_bf16_body<H, float, 64, 64, true, false, true, TC_FFN_LN_B>(p, bx, by, bz, gx, gy, As, Bs);
}

// ---------------------------------------------------------------------------------------------- host side
template <typename T, typename TC, int BM, int BN, bool TA, bool TB>
void launch_one(const GemmDev& d, dim3 grid, hipStream_t s) {
    if constexpr (sizeof(T) == 4) hipLaunchKernelGGL((gemm_kernel<TC, BM, BN, TA, TB>), grid, dim3(256), 0, s, d);
    else if (d.kchunk > 128) hipLaunchKernelGGL((gemm_bf16_kernel<T, TC, BM, BN, TA, TB, true>), grid, dim3(256), 0, s, d);
    else hipLaunchKernelGGL((gemm_bf16_kernel<T, TC, BM, BN, TA, TB, false>), grid, dim3(256), 0, s, d);
}

// the three hooked products (64x64 tiles; gemm_plan has checked operand kinds and output types)
template <typename T>
int launch_ffn(const GemmDev& d, dim3 grid, hipStream_t s) {
    if constexpr (sizeof(T) == 4) {
        if (d.ffn.mode == TC_FFN_LN_A) hipLaunchKernelGGL((gemm_kernel<float, 64, 64, false, true, TC_FFN_LN_A>), grid, dim3(256), 0, s, d);
        else if (d.ffn.mode == TC_FFN_LN_B) hipLaunchKernelGGL((gemm_kernel<float, 64, 64, true, false, TC_FFN_LN_B>), grid, dim3(256), 0, s, d);
        else hipLaunchKernelGGL((gemm_kernel<float, 64, 64, false, false, TC_FFN_EP>), grid, dim3(256), 0, s, d);
    } else {
        if (d.ffn.mode == TC_FFN_LN_A) hipLaunchKernelGGL((gemm_bf16_kernel<T, T, 64, 64, false, true, true, TC_FFN_LN_A>), grid, dim3(256), 0, s, d);
        else if (d.ffn.mode == TC_FFN_LN_B) hipLaunchKernelGGL((gemm_bf16_kernel<T, float, 64, 64, true, false, true, TC_FFN_LN_B>), grid, dim3(256), 0, s, d);
        else hipLaunchKernelGGL((gemm_bf16_kernel<T, T, 64, 64, false, false, true, TC_FFN_EP>), grid, dim3(256), 0, s, d);
    }
    return tc_launch_status();
}

template <typename T, typename TC, int BM, int BN>
int launch(const GemmDev& d, int transA, int transB, dim3 grid, hipStream_t s) {
    if (!transA && transB) launch_one<T, TC, BM, BN, false, true>(d, grid, s);
    else if (!transA && !transB) launch_one<T, TC, BM, BN, false, false>(d, grid, s);
    else if (transA && !transB) launch_one<T, TC, BM, BN, true, false>(d, grid, s);
    else launch_one<T, TC, BM, BN, true, true>(d, grid, s);
    return tc_launch_status();
}

// fills the device-side descriptor and the launch grid; false: bad arguments
template <typename T>
bool gemm_plan(const TcGemm* g, GemmDev& d, dim3& grid, bool& use128_out, bool force64 = false) {
    d.A = g->A; d.B = g->B; d.C = g->C; d.bias = g->bias; d.R = g->R;
    d.M = g->M; d.N = g->N; d.K = g->K; d.lda = g->lda; d.ldb = g->ldb; d.ldc = g->ldc; d.ldr = g->ldr;
    d.nb2 = g->nb2; d.splitk = g->splitk;
    d.sA1 = g->sA1; d.sA2 = g->sA2; d.sB1 = g->sB1; d.sB2 = g->sB2;
    d.sC1 = g->sC1; d.sC2 = g->sC2; d.sR1 = g->sR1; d.sR2 = g->sR2;
    d.alpha = g->alpha; d.accumulate = g->accumulate; d.act = g->act;
    d.rowsum = g->rowsum;
    d.sBias1 = g->sBias1; d.sRow1 = g->sRow1;
    d.bgap_every = g->bgap_every; d.bgap = g->bgap;
    d.ffn.mode = g->ffn_mode; d.ffn.nchunk = g->ffn_nchunk; d.ffn.chunk_n = g->ffn_chunk_n; d.ffn.ldd = g->ffn_ldd;
    d.ffn.sRow1 = (int)g->ffn_sRow1; d.ffn.sPar1 = (int)g->ffn_sPar1; d.ffn.eps = g->ffn_eps;
    d.ffn.part = g->ffn_part; d.ffn.stat = g->ffn_stat; d.ffn.gamma = g->ffn_gamma; d.ffn.beta = g->ffn_beta;
    d.ffn.d = g->ffn_mode == TC_FFN_LN_A ? g->ffn_aout : g->ffn_d;      // one slot: EP reads d there, LN_A may write the activated rows
    d.ffn.part2 = g->ffn_part2;
    if (g->bn_part && !g->ffn_mode) { d.ffn.mode = GEMM_BN_STATS; d.ffn.part2 = g->bn_part; d.ffn.part = g->bn_shift; }
    if (g->ffn_mode || g->bn_part) force64 = true;
    constexpr int VEC = 16 / (int)sizeof(T);                         // elements per 16-byte vector
    auto aligned = [&](const void* ptr, int ld, long long s1, long long s2) {
        return ((uintptr_t)ptr % 16 == 0) && (ld % VEC == 0) && (s1 % VEC == 0) && (s2 % VEC == 0);
    };
    d.vecA = aligned(g->A, g->lda, g->sA1, g->sA2);
    d.vecB = aligned(g->B, g->ldb, g->sB1, g->sB2);
    {   // 4-wide epilogue accesses: C (fp32 or T) and R (T) rows must be 4-element aligned
        const int csz = g->c_f32 ? 4 : (int)sizeof(T);
        const bool cok = ((uintptr_t)g->C % (4 * csz) == 0) && (g->ldc % 4 == 0) && (g->sC1 % 4 == 0) && (g->sC2 % 4 == 0);
        const bool rok = !g->R || (((uintptr_t)g->R % (4 * sizeof(T)) == 0) && (g->ldr % 4 == 0) && (g->sR1 % 4 == 0) && (g->sR2 % 4 == 0));
        d.vecC = cok && rok;
        d.vec8C = !g->c_f32 && sizeof(T) == 2 && cok && rok && ((uintptr_t)g->C % 16 == 0) && (g->ldc % 8 == 0) && (g->sC1 % 8 == 0) &&
                  (g->sC2 % 8 == 0) && (g->N % 8 == 0);
    }
    const int nb = g->nb1 * g->nb2;
    const long long big = (long long)((g->M + 127) / 128) * ((g->N + 127) / 128) * nb;
    // 128x128 tiles only for very large grids: on this model's shapes (K <= 2048, M <= 800k) 64x64 tiles measured 2-3 % faster
    // end to end at every threshold tried (more workgroups in flight per CU, and the dX/dW pair launch needs small tiles)
    static const long long thr128 = getenv("TC_GEMM_THR128") ? atoll(getenv("TC_GEMM_THR128")) : 100000;
    const bool use128 = !force64 && big >= thr128 && g->M >= 96 && g->N >= 96;
    const int BM = use128 ? 128 : 64, BN = use128 ? 128 : 64;
    constexpr int BK = sizeof(T) == 4 ? 16 : 64;
    // split-K plan: the caller's request (weight gradients), or -- with a workspace -- our own for few-tile / long-K
    // products whose K loop would otherwise run serially on a handful of CUs
    constexpr int CNT_BYTES = 16384, PART_BYTES = 64 * 64 * 4, FIX_GROUP = 16;
    const long long tiles64 = (long long)((g->M + 63) / 64) * ((g->N + 63) / 64) * nb;
    const long long slots = (g->ws && g->ws_bytes > CNT_BYTES && (uintptr_t)g->ws % 16 == 0) ? (g->ws_bytes - CNT_BYTES) / PART_BYTES : 0;
    int want = g->splitk;
    bool fix = false;
    if (!use128 && slots > 0) {
#ifndef GEMM_AS_TILES
#define GEMM_AS_TILES 384
#define GEMM_AS_TARGET 512
#define GEMM_AS_KPER 256
#endif
        if (want == 1 && !g->atomic && g->ffn_mode != TC_FFN_EP && !g->bn_part && g->K >= 2 * GEMM_AS_KPER && tiles64 <= GEMM_AS_TILES) {
            long long sk = GEMM_AS_TARGET / tiles64;
            if (sk > g->K / GEMM_AS_KPER) sk = g->K / GEMM_AS_KPER;
            if (sk > FIX_GROUP) sk = FIX_GROUP;
            if (sk >= 2 && tiles64 * sk <= slots) { want = (int)sk; fix = true; }
        } else if (want > 128) {
            // caller-requested split beyond 128 (the classifier's dW: K = 802816): groups of 16 splits fold through the
            // workspace, only the group leaders touch C atomically -- a 64-long atomic chain instead of a 1024-long one.
            // (up to 128 the plain atomic epilogue measured equal or better: the leader's L2-bypassing reads of its group's
            //  partials cost what the shorter chain saves)
            fix = tiles64 * want <= slots && tiles64 * ((want + FIX_GROUP - 1) / FIX_GROUP) <= CNT_BYTES / 4;
        }
    }
    int kchunk = (g->K + want - 1) / want;
    kchunk = (kchunk + BK - 1) / BK * BK;
    d.kchunk = kchunk;
    d.splitk = (g->K + kchunk - 1) / kchunk;
    if (d.splitk < 1) d.splitk = 1;
    d.fix_group = 0; d.fix_ngroups = 1; d.ws_part = nullptr; d.ws_cnt = nullptr;
    if (fix && d.splitk > 1) {
        d.fix_group = FIX_GROUP;
        d.fix_ngroups = (d.splitk + FIX_GROUP - 1) / FIX_GROUP;
        d.ws_cnt = reinterpret_cast<int*>(g->ws);
        d.ws_part = reinterpret_cast<float*>(reinterpret_cast<char*>(g->ws) + CNT_BYTES);
    }
    d.atomic = ((d.splitk > 1 && !d.fix_group) || g->atomic) ? 1 : 0;
    grid = dim3((g->N + BN - 1) / BN, (g->M + BM - 1) / BM, nb * d.splitk);
    d.m_splitk = fdiv_make(d.splitk, 65536); d.m_nb2 = fdiv_make(d.nb2, 65536);
    d.m_gx = fdiv_make(grid.x, (unsigned long long)grid.x * grid.y * grid.z);          // (every linear index the launchers divide by gx is below the block count)
    {   // XCD-aware tile numbering pays where several N-tiles share an A tile and there are enough tiles to spread (A/B: TC_GEMM_XCD=0)
        static const int xcd_on = getenv("TC_GEMM_XCD") ? atoi(getenv("TC_GEMM_XCD")) : 1;
        d.xcd = (xcd_on && grid.x > 1 && (long long)grid.x * grid.y >= 64) ? 1 : 0;
    }
    use128_out = use128;
    if (g->bn_part) {                                               // statistics come from the LDS-staged plain-store epilogue of ONE product
        if (sizeof(T) != 2 || g->c_f32 || g->accumulate || g->atomic || d.splitk != 1 || nb != 1 || !d.vec8C || g->ffn_mode || g->act != TC_ACT_NONE)
            return false;
    }
    if (g->ffn_mode) {
        // the hooked kernels assume whole 16-byte strips everywhere and, for EP, the LDS-staged plain-store epilogue
        if (!d.vecA || !d.vecB || (g->ffn_mode != TC_FFN_LN_B && g->K % VEC) || g->N % VEC || g->nb2 != 1 || (uintptr_t)g->ffn_gamma % 16 || (uintptr_t)g->ffn_beta % 16 ||
            g->ffn_sPar1 % VEC)
            return false;
        if (g->ffn_mode == TC_FFN_LN_A && (g->transA || !g->transB || g->c_f32 || g->K != g->ffn_nchunk * g->ffn_chunk_n ||
                                           (g->ffn_aout && ((uintptr_t)g->ffn_aout % 16 || g->ffn_ldd % VEC))))
            return false;
        if (g->ffn_mode == TC_FFN_LN_B && (!g->transA || g->transB || !(g->c_f32 || sizeof(T) == 4))) return false;
        if (g->ffn_mode == TC_FFN_EP && (g->transA || g->transB || g->c_f32 || g->accumulate || g->atomic || d.splitk != 1 || g->bias || g->R ||
                                         g->act != TC_ACT_NONE || (sizeof(T) == 2 && !d.vec8C) || (uintptr_t)g->ffn_d % 16 || g->ffn_ldd % VEC))
            return false;
    }
    return grid.y <= 65535 && grid.z <= 65535;
}

template <typename T>
int gemm_typed(const TcGemm* g, hipStream_t s) {
    GemmDev d;
    dim3 grid;
    bool use128;
    if (!gemm_plan<T>(g, d, grid, use128)) return TC_ERR_ARG;
    if (g->ffn_mode) return launch_ffn<T>(d, grid, s);
    if (g->c_f32)
        return use128 ? launch<T, float, 128, 128>(d, g->transA, g->transB, grid, s)
                      : launch<T, float, 64, 64>(d, g->transA, g->transB, grid, s);
    return use128 ? launch<T, T, 128, 128>(d, g->transA, g->transB, grid, s)
                  : launch<T, T, 64, 64>(d, g->transA, g->transB, grid, s);
}


}  // namespace

static bool gemm_args_ok(const TcGemm* g) {
    if (!g || !g->A || !g->B || !g->C || g->M <= 0 || g->N <= 0 || g->K <= 0 || g->nb1 < 1 || g->nb2 < 1 || g->splitk < 1) return false;
    if ((g->splitk > 1 || g->atomic) && (!g->accumulate || (g->dtype != TC_F32 && !g->c_f32) || g->act != TC_ACT_NONE)) return false;
    if (g->act != TC_ACT_NONE && g->act != TC_ACT_SIGMOID && g->act != TC_ACT_SCALE) return false;
    if (g->act == TC_ACT_SCALE && g->ffn_mode != TC_FFN_NONE) return false;
    if (g->bgap_every < 0 || (g->bgap_every > 0 && (g->transB || g->bgap_every % 64 || (g->bgap % 8)))) return false;
    if (g->ffn_mode < TC_FFN_NONE || g->ffn_mode > TC_FFN_EP) return false;
    if (g->ffn_mode != TC_FFN_NONE) {
        if (!g->ffn_gamma || !g->ffn_beta || !g->ffn_stat || g->bgap_every) return false;
        if (g->ffn_mode == TC_FFN_LN_A && (!g->ffn_part || g->ffn_nchunk < 1 || g->ffn_chunk_n < 1)) return false;
        if (g->ffn_mode == TC_FFN_EP && (!g->ffn_d || !g->ffn_part2)) return false;
        if (g->ffn_sRow1 < 0 || g->ffn_sRow1 > 0x7fffffffLL || g->ffn_sPar1 < 0 || g->ffn_sPar1 > 0x7fffffffLL) return false;
    }
    return true;
}

#ifdef TC_GEMM_TIMING
extern "C" int tc_gemm_dbg_read(unsigned long long* dst, int reset) {
    int rc = (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_gemm_dbg), sizeof(unsigned long long) * 1024 * 8);
    if (reset) {
        static unsigned long long z[1024 * 8];
        unsigned int zn = 0;
        rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_dbg), z, sizeof(z));
        rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_dbg_n), &zn, sizeof(zn));
    }
    return rc;
}
#endif
extern "C" int tc_gemm(const TcGemm* g, void* stream) {
    if (!gemm_args_ok(g)) return TC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    TC_DISPATCH_DTYPE(g->dtype, return gemm_typed<T>(g, s));
    return TC_ERR_ARG;
}

extern "C" int tc_gemm_pair(const TcGemm* a, const TcGemm* b, void* stream) {
    if (!gemm_args_ok(a) || !gemm_args_ok(b) || a->bn_part || b->bn_part) return TC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if ((a->dtype == TC_BF16 || a->dtype == TC_F16) && b->dtype == a->dtype && !a->transA && !a->transB && !a->c_f32 && b->transA && !b->transB && b->c_f32 &&
        (a->ffn_mode == TC_FFN_NONE || a->ffn_mode == TC_FFN_EP) && (b->ffn_mode == TC_FFN_NONE || b->ffn_mode == TC_FFN_LN_B)) {
        GemmPairDev q;
        dim3 ga, gb;
        bool bigA, bigB;
        if (!gemm_plan<bf16_t>(a, q.a, ga, bigA) || !gemm_plan<bf16_t>(b, q.b, gb, bigB)) return TC_ERR_ARG;     // (sizes only: either 16-bit type)
        const long long nA = (long long)ga.x * ga.y * ga.z, nB = (long long)gb.x * gb.y * gb.z;
        if (!bigA && !bigB && nA + nB < 0x7fffffffLL) {
            q.nA = (int)nA; q.gxA = ga.x; q.gyA = ga.y; q.gxB = gb.x; q.gyB = gb.y;
            q.mgxA = fdiv_make(ga.x, nA + nB); q.mgyA = fdiv_make(ga.y, nA + nB); q.mgxB = fdiv_make(gb.x, nA + nB); q.mgyB = fdiv_make(gb.y, nA + nB);
            static const int xcd_b = getenv("TC_PAIR_XCD_B") ? atoi(getenv("TC_PAIR_XCD_B")) : 1;   // A/B switch
            if (xcd_b && gb.x * gb.y > 1) q.b.xcd |= 2;
            if (a->dtype == TC_BF16) hipLaunchKernelGGL(gemm_pair_kernel<bf16_t>, dim3((unsigned)(nA + nB)), dim3(256), 0, s, q);
            else hipLaunchKernelGGL(gemm_pair_kernel<f16_t>, dim3((unsigned)(nA + nB)), dim3(256), 0, s, q);
            return tc_launch_status();
        }
    }
    const int rc = tc_gemm(a, stream);                       // shapes the paired kernel does not cover: two launches
    return rc != TC_OK ? rc : tc_gemm(b, stream);
}

extern "C" int tc_gemm_multi(const TcGemm* g, int n, void* stream) {
    if (!g || n < 1) return TC_ERR_ARG;
    for (int i = 0; i < n; ++i) if (!gemm_args_ok(g + i) || g[i].bn_part) return TC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    bool ok = n <= GEMM_MULTI_MAX;
    GemmMultiDev q;
    GemmDev plan[GEMM_MULTI_MAX];
    dim3 grids[GEMM_MULTI_MAX];
    int kinds[GEMM_MULTI_MAX], order[GEMM_MULTI_MAX];
    for (int i = 0; ok && i < n; ++i) {
        const TcGemm& t = g[i];
        int kind = -1;
        const bool h16 = (t.dtype == TC_BF16 || t.dtype == TC_F16) && t.dtype == g[0].dtype;     // one 16-bit type per launch
        if (h16 && !t.transA && t.transB && !t.c_f32) kind = t.ffn_mode == TC_FFN_LN_A ? 3 : (t.ffn_mode ? -1 : 0);
        else if (h16 && !t.transA && !t.transB && !t.c_f32) kind = t.ffn_mode == TC_FFN_EP ? 4 : (t.ffn_mode ? -1 : 1);
        else if (h16 && t.transA && !t.transB && t.c_f32) kind = t.ffn_mode == TC_FFN_LN_B ? 5 : (t.ffn_mode ? -1 : 2);
        bool big;
        if (kind < 0 || !gemm_plan<bf16_t>(&t, plan[i], grids[i], big, true)) { ok = false; break; }
        kinds[i] = kind; order[i] = i;
    }
    long long blk = 0;
    if (ok) {
        // workgroups are dispatched in index order: the problems whose workgroups run the longest K loops go first, so the launch
        // does not end on a few long-running stragglers (measured 5-12 % on the bridge's four-scale MixFFN launches)
        std::stable_sort(order, order + n, [&](int a, int b) { return plan[a].kchunk > plan[b].kchunk; });
        static const int xcd_b = getenv("TC_MULTI_XCD_B") ? atoi(getenv("TC_MULTI_XCD_B")) : 1;   // A/B switch
        for (int j = 0; j < n; ++j) {
            const int i = order[j];
            q.p[j] = plan[i]; q.kind[j] = kinds[i]; q.gx[j] = grids[i].x; q.gy[j] = grids[i].y; q.blk0[j] = (int)blk;
            q.mgy[j] = fdiv_make(grids[i].y, (unsigned long long)grids[i].x * grids[i].y * grids[i].z);
            if (xcd_b && (kinds[i] == 2 || kinds[i] == 5) && grids[i].x * grids[i].y > 1) q.p[j].xcd |= 2;
            blk += (long long)grids[i].x * grids[i].y * grids[i].z;
            if (blk > 0x7fffffffLL) ok = false;
        }
    }
    if (ok) {
        q.n = n;
        if (g[0].dtype == TC_BF16) hipLaunchKernelGGL(gemm_multi_kernel<bf16_t>, dim3((unsigned)blk), dim3(256), 0, s, q);
        else hipLaunchKernelGGL(gemm_multi_kernel<f16_t>, dim3((unsigned)blk), dim3(256), 0, s, q);
        return tc_launch_status();
    }
    for (int i = 0; i < n; ++i) {                              // kinds / sizes the merged kernel does not cover: one launch each
        const int rc = tc_gemm(g + i, stream);
        if (rc != TC_OK) return rc;
    }
    return TC_OK;
}
