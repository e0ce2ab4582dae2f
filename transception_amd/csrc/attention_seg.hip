// Segmented single-launch bf16 attention for the Dual Transformer Bridge.
//
// The bridge's 6076 query tokens per image live in a stage-major buffer (four row segments, one per encoder scale, each
// holding B images back to back), while the 784 reduced K/V tokens are image-major.  One launch covers every
// (segment, image, query tile) instead of four launches of 50-400 workgroups, so the chip is filled and the launch boundary is paid
// once.  The kernels prefetch the next 128-key K/V fill into registers while the current one is being consumed, fold the softmax
// scale into the exp2 argument (one FMA per score) and mask keys only in the tail tile; see the comment of each kernel.
// fp32 storage falls back to the per-segment fp32 kernels of attention.hip (parity path).
#include "tc_common.h"
#include <type_traits>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int D = 64, KB = 128, LDR = D + 8;   // LDR: 144-byte rows = 36 words (4 mod 32)
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f
#define NEG_BIG (-1.0e30f)
#define RESCALE_THR 8.0f

struct Segs {
    int n;
    int nq[4];        // queries per image in the segment
    int row0[4];      // first row of the segment in the stage-major buffers
    int t32[5];       // prefix sums of ceil(nq/32): the 32-query wave tiles of an image, numbered through its segments
};

__device__ __forceinline__ int pi_row(int i) { return 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3); }
__device__ __forceinline__ int d_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
template <typename V8> __device__ __forceinline__ V8 ld_frag(const bf16_t* p) { return *reinterpret_cast<const V8*>(p); }
template <typename H> __device__ __forceinline__ typename TcHalf<H>::v8 pack8(const f32x16& v, int o) {
    typename TcHalf<H>::v8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (typename TcHalf<H>::e)v[o + i];
    return r;
}
// [keys][d] tiles kept row-major in LDS and consumed as MFMA A-operands with keys as the k index are gathered by the hardware
// transpose read: lane i of a 16-lane group hands in the address of key row (i >> 2) of a 4-key block, d columns 4 (i & 3).., and
// receives column i.  With 144-byte rows the 4 rows of a read must lie 4 rows apart to start 16 banks apart, so the keys of every
// 16-key group are stored 4x4-transposed (row = (key & ~15) | (key & 3) << 2 | (key >> 2) & 3): conflict-free, one 16-byte LDS
// store per strip instead of eight 2-byte transposed ones.
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
template <typename V8> __device__ __forceinline__ V8 ld_frag_tr(const bf16_t* lo, const bf16_t* hi) {
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(lo));
    const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(hi));
    return __builtin_bit_cast(V8, (s16x8_t)__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ int key_row(int r) { return (r & ~15) | ((r & 3) << 2) | ((r >> 2) & 3); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Forward.  ONE workgroup of FW_NW = 12 waves per CU; each wave owns a 32-query tile (the tiles are numbered through the four scales of
// an image: 190 per image at 224^2 = 16 workgroups per image = 256 at B = 16, one per CU).
//   * The twelve waves share one K/V staging: 128-key tiles, double-buffered in LDS and fetched a whole tile ahead into three
//     registers per thread -- one barrier and no exposed load per tile (4-wave workgroups staged three times the bytes per CU).
//   * Softmax against a reference exponent m that is NOT the running row maximum: any m within fp32's exponent range of the true
//     maximum gives the same quotient, so m is set once (integer-valued, from the row's first sub-tile) and raised only when a row
//     sum shows that the scores have outgrown it by 2^30 (2^14 for fp16 storage of P) -- one compare per 32-key sub-tile instead of
//     a 16-element maximum, a lane swap and a compare (-25 % VALU work: the SIMD issues one vector instruction per 4 cycles whoever it
//     comes from, so every VALU instruction removed is time -- scripts/exp/overlap2.hip, DESIGN.md section 5 "Round 4").  Integer m: P
//     differs from the maximum-referenced P by an exact power of two.
//   * Q and O tiles pass through a wave-private LDS tile so that global memory sees whole 128-byte rows (eight lanes x 16 bytes): a
//     per-lane 16-byte access at a row stride touches 32 lines per instruction and queued in the address unit for ~4 us at each end.
// Workgroups go to the 8 XCDs round-robin by linear index, and each XCD has its own L2.  The workgroups that share operands (the 16
// query-tile groups of one image read the same K / V; the 7 key blocks of one (image, query chunk) read the same Q / dO) are numbered
// so that they land on ONE XCD: index l runs on XCD l % 8 and takes the (l / 8)-th unit of that XCD's contiguous share.
__device__ __forceinline__ int xcd_block(int l, int n) { return (n & 7) ? l : (l & 7) * (n >> 3) + (l >> 3); }
constexpr int FW_NW = 12, FW_NT = FW_NW * 64, FW_NC = KB * 8, FW_NF = (2 * FW_NC + FW_NT - 1) / FW_NT;
constexpr int FW_SLOT = 2 * KB * LDR;                           // one ring slot: K tile | V tile
constexpr size_t FW_SMEM = (size_t)(2 * FW_SLOT + FW_NW * 32 * LDR) * sizeof(bf16_t);
template <typename H>
__global__ __launch_bounds__(FW_NT, 1) void attn_fwd_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                                const bf16_t* __restrict__ V, int ldv, long long skv, bf16_t* __restrict__ O,
                                                                int ldo, float* __restrict__ lse, Segs sg, int Nk, float qs, int wide_o) {
    extern __shared__ __attribute__((aligned(16))) bf16_t fw_smem[];      // [2 slots][K tile | V tile][KB][LDR], then FW_NW wave tiles [32][LDR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int nwt = sg.t32[sg.n], bpi = (nwt + FW_NW - 1) / FW_NW;
    const int bx = xcd_block(blockIdx.x, gridDim.x);
    const int b = bx / bpi, wt = (bx - b * bpi) * FW_NW + wave;
    int sgi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < sg.n && wt >= sg.t32[i]) sgi = i;
    const int nq = sg.nq[sgi];
    const int tq0 = (wt - sg.t32[sgi]) * 32;                    // first query of the wave's tile within its segment-image
    const bool live = wt < nwt, ok = live && tq0 + j < nq;
    const long long trow0 = (long long)sg.row0[sgi] + (long long)b * nq + tq0;
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    typedef typename TcHalf<H>::v8 V8;
    bf16_t* wtile = fw_smem + 2 * FW_SLOT + wave * (32 * LDR);
    V8 qf[4];
    {
        uint4 qv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 8 * i + (lane >> 3);
            qv[i] = (live && tq0 + r < nq) ? *reinterpret_cast<const uint4*>(Q + (trow0 + r) * ldq + 8 * (lane & 7)) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(wtile + (8 * i + (lane >> 3)) * LDR + 8 * (lane & 7)) = qv[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = ld_frag<V8>(wtile + j * LDR + 16 * ks + 8 * h);
    }
    f32x16 acc0, acc1;                                           // (qs = scale * log2(e), or 1 when Q arrives scaled)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float m = NEG_BIG, lsum = 0.f;
    const int krow = pi_row(j);
    // staging: chunk id = tid + i * FW_NT; ids below FW_NC are K (row = id >> 3, 16-byte chunk id & 7), the next FW_NC are V
    uint4 st[FW_NF];
    auto fetch = [&](int kb0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < FW_NF; ++i) {
            const int id = tid + i * FW_NT, isv = id >= FW_NC, r = (id - isv * FW_NC) >> 3, c8 = (id & 7) * 8;
            const bf16_t* src = isv ? Vb : Kb;
            const int ld = isv ? ldv : ldk;
            st[i] = id < 2 * FW_NC ? *reinterpret_cast<const uint4*>(src + (long long)min(kb0 + r, Nk - 1) * ld + c8) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto stash = [&](int slot) __attribute__((always_inline)) {
        bf16_t* base = fw_smem + slot * FW_SLOT;
#pragma unroll
        for (int i = 0; i < FW_NF; ++i) {
            const int id = tid + i * FW_NT, isv = id >= FW_NC, r = (id - isv * FW_NC) >> 3, c8 = (id & 7) * 8;
            if (id < 2 * FW_NC) *reinterpret_cast<uint4*>(base + isv * (KB * LDR) + (isv ? key_row(r) : r) * LDR + c8) = st[i];
        }
    };
    const int nt = (Nk + KB - 1) / KB;
    fetch(0);
    stash(0);
    if (nt > 1) fetch(KB);
    __syncthreads();
    constexpr float REREF = std::is_same<H, f16_t>::value ? 16384.0f : 1073741824.0f;   // P must stay finite in its 16-bit storage type
    for (int t = 0; t < nt; ++t) {
        const int kb0 = t * KB;
        if (t + 1 < nt) stash((t + 1) & 1);                     // (every wave left tile t - 1, the slot's last reader, at the barrier below)
        if (t + 2 < nt) fetch(kb0 + 2 * KB);
        const bf16_t* Ks = fw_smem + (t & 1) * FW_SLOT;
        const bf16_t* Vs = Ks + KB * LDR;
        const int nsub = min(KB / 32, (Nk - kb0 + 31) / 32);
#pragma unroll 1
        for (int sub = 0; sub < nsub; ++sub) {
            // S^T tile: 4 chained MFMAs
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const bf16_t* kp = Ks + (32 * sub + krow) * LDR + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s = TcHalf<H>::mfma(ld_frag<V8>(kp + 16 * ks), qf[ks], s);
            const int kv0 = kb0 + 32 * sub;
            if (kv0 + 32 > Nk) {                                // tail tile (wave-uniform)
#pragma unroll
                for (int r = 0; r < 16; ++r) if (kv0 + 16 * h + r >= Nk) s[r] = NEG_BIG;
            }
            auto rowmax = [&]() __attribute__((always_inline)) {
                float mx = s[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
                const unsigned u = __float_as_uint(mx);
                const auto pr = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                return ceilf(fmaxf(__uint_as_float(pr[0]), __uint_as_float(pr[1])) * qs);
            };
            if (kv0 == 0) m = rowmax();                         // the row's first sub-tile sets the reference exponent
            f32x16 p;
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { p[r] = fast_exp2(fmaf(s[r], qs, -m)); rs += p[r]; }
            if (__any(!(rs < REREF))) {                         // rare: some row outgrew its reference; re-reference the rows of the wave
                const float mn = fmaxf(m, rowmax());
                const float alpha = fast_exp2(m - mn);
                lsum *= alpha;
                m = mn;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] *= alpha; acc1[r] *= alpha; }
                rs = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { p[r] = fast_exp2(fmaf(s[r], qs, -m)); rs += p[r]; }
            }
            lsum += rs;
            // O^T += V^T P^T
            const int gi = lane & 15, gq = (lane >> 4) & 1;
            const bf16_t* vp = Vs + (32 * sub + 16 * h + 4 * (gi >> 2)) * LDR + 16 * gq + 4 * (gi & 3);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const V8 pb = pack8<H>(p, 8 * k2);
                acc0 = TcHalf<H>::mfma(ld_frag_tr<V8>(vp + (2 * k2) * LDR, vp + (2 * k2 + 1) * LDR), pb, acc0);
                acc1 = TcHalf<H>::mfma(ld_frag_tr<V8>(vp + (2 * k2) * LDR + 32, vp + (2 * k2 + 1) * LDR + 32), pb, acc1);
            }
        }
        __syncthreads();
    }
    lsum += __shfl_xor(lsum, 32, 64);
    const float inv = 1.0f / lsum;
    if (ok && h == 0) lse[trow0 + j] = (m + log2f(lsum)) * LN2;
    if (wide_o) {                                               // O tile through the wave's LDS tile: whole rows, 16-byte stores
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<H>(reinterpret_cast<H*>(wtile + j * LDR + 8 * g + 4 * h), make_float4(acc0[4 * g] * inv, acc0[4 * g + 1] * inv, acc0[4 * g + 2] * inv, acc0[4 * g + 3] * inv));
            st4<H>(reinterpret_cast<H*>(wtile + j * LDR + 32 + 8 * g + 4 * h), make_float4(acc1[4 * g] * inv, acc1[4 * g + 1] * inv, acc1[4 * g + 2] * inv, acc1[4 * g + 3] * inv));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 8 * i + (lane >> 3);
            const uint4 v = *reinterpret_cast<const uint4*>(wtile + r * LDR + 8 * (lane & 7));
            if (live && tq0 + r < nq) *reinterpret_cast<uint4*>(O + (trow0 + r) * ldo + 8 * (lane & 7)) = v;
        }
    } else if (ok) {                                            // O rows that are not 16-byte aligned: 8-byte stores per lane
        bf16_t* orow = O + (trow0 + j) * ldo;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<H>(reinterpret_cast<H*>(orow + 8 * g + 4 * h), make_float4(acc0[4 * g] * inv, acc0[4 * g + 1] * inv, acc0[4 * g + 2] * inv, acc0[4 * g + 3] * inv));
            st4<H>(reinterpret_cast<H*>(orow + 32 + 8 * g + 4 * h), make_float4(acc1[4 * g] * inv, acc1[4 * g + 1] * inv, acc1[4 * g + 2] * inv, acc1[4 * g + 3] * inv));
        }
    }
}

// Forward, hand-scheduled (the default for 16-bit storage; the kernel above stays as the generic fallback and the A/B baseline).
// Same decomposition -- one 12-wave workgroup per CU, a 32-query tile per wave -- but the whole key loop of a wave is one
// instruction stream emitted by gen_attn_asm.py (attn_fwd_asm.inc; the reasons and the measurements are in that file's header):
// software-pipelined so that every group of four MFMAs carries softmax arithmetic or fragment reads of another sub-tile, which
// keeps the twelve waves from running their matrix and VALU phases in lock step.  Scores are produced directly as
// s * scale * log2(e) - m: Q is scaled when it is staged and -m is the C operand of the first QK^T MFMA, so the softmax is
// max3 tree + exp2 + add + pack per score.  K / V sub-tiles of 32 keys go through a 12-slot LDS ring (buffer loads six
// sub-tiles ahead: waves 0-3 stage K, waves 4-7 V, one 16-byte chunk per thread; one barrier per FOUR sub-tiles).  This function stages
// Q and the first five sub-tiles, hands the addresses to the stream and stores the O tile / lse the stream leaves in the wave's LDS tile.
#include "attn_fwd_asm.inc"
typedef int tc_i32x4 __attribute__((ext_vector_type(4)));
// LDS: 12 ring slots of one 32-key K | V sub-tile; the twelve 32-query wave tiles (Q in, O out) alias slots 6-11, which the stream
// does not store to before every wave has read its Q fragments and does not read after the barrier in front of the O tiles.
#ifndef TC_AS_AHEAD
#define TC_AS_AHEAD 6                                            // keep in sync with AHEAD of gen_attn_asm.py (TC_ATTN_AHEAD)
#endif
constexpr int AS_NW = 12, AS_VOFF = 32 * LDR * 2, AS_SLOT = 2 * AS_VOFF, AS_NSLOT = 12, AS_AHEAD = TC_AS_AHEAD, AS_WT = 32 * LDR * 2, AS_WT0 = 6 * AS_SLOT,
              AS_LSE0 = AS_NSLOT * AS_SLOT;
static_assert(AS_WT0 + AS_NW * AS_WT <= AS_LSE0, "wave tiles must fit in the ring");
#ifdef TC_ATTN_ASM_TIMING
constexpr size_t AS_SMEM = AS_LSE0 + AS_NW * 32 * sizeof(float) + 8192;
__device__ unsigned long long g_attn_dbg[256 * 12 * 8];
#else
constexpr size_t AS_SMEM = AS_LSE0 + AS_NW * 32 * sizeof(float);
#endif
template <typename H>
__global__ __launch_bounds__(AS_NW * 64, 1) void attn_fwd_asm_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                                     const bf16_t* __restrict__ V, int ldv, long long skv, bf16_t* __restrict__ O,
                                                                     int ldo, float* __restrict__ lse, Segs sg, int Nk, float qs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char as_smem[];
#ifdef TC_ATTN_ASM_TIMING
    const unsigned long long dbg_t0 = __builtin_readcyclecounter();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int nwt = sg.t32[sg.n], bpi = (nwt + AS_NW - 1) / AS_NW;
    const int bx = xcd_block(blockIdx.x, gridDim.x);
    const int b = bx / bpi, wt = (bx - b * bpi) * AS_NW + wave;
    int sgi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < sg.n && wt >= sg.t32[i]) sgi = i;
    const int nq = sg.nq[sgi];
    const int tq0 = (wt - sg.t32[sgi]) * 32;
    const bool live = wt < nwt;
    const long long trow0 = (long long)sg.row0[sgi] + (long long)b * nq + tq0;
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    unsigned char* wtile = as_smem + AS_WT0 + wave * AS_WT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {                               // Q rows (scaled by qs = scale * log2(e) here unless the producer did: qs = 1), whole 128-byte rows in
        const int r = 8 * i + (lane >> 3);
        const uint4 v = (live && tq0 + r < nq) ? *reinterpret_cast<const uint4*>(Q + (trow0 + r) * ldq + 8 * (lane & 7)) : make_uint4(0u, 0u, 0u, 0u);
        unsigned w[4] = {v.x, v.y, v.z, v.w};
        if (qs != 1.0f) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float lo, hi;
                unpack2<H>(w[e], lo, hi);
                w[e] = pack2<H>(lo * qs, hi * qs);
            }
        }
        *reinterpret_cast<uint4*>(wtile + r * (LDR * 2) + 16 * (lane & 7)) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    // staging roles: thread (tid & 255) of waves 0-3 owns the K chunk (row sr, 16-byte column sc) of every sub-tile, of waves 4-7 the V chunk
    const int role = wave >> 2, sr = (tid & 255) >> 3, sc = tid & 7;
    if (role < 2) {
        const bf16_t* src = role ? Vb : Kb;
        const int ld = role ? ldv : ldk;
#pragma unroll
        for (int s = 0; s < AS_AHEAD - 1; ++s) {                // sub-tiles 0..4 into ring slots 0..4 (rows past Nk: a duplicate, masked later)
            const uint4 x = *reinterpret_cast<const uint4*>(src + (long long)min(32 * s + sr, Nk - 1) * ld + 8 * sc);
            *reinterpret_cast<uint4*>(as_smem + s * AS_SLOT + (role ? AS_VOFF + key_row(sr) * (LDR * 2) : sr * (LDR * 2)) + 16 * sc) = x;
        }
    }
    __syncthreads();
    {
        const unsigned lds0 = (unsigned)(uintptr_t)as_smem;     // LDS offset = low half of the flat address
        const int gi = lane & 15, gq = (lane >> 4) & 1, nsub = (Nk + 31) / 32, nv = Nk - 32 * (nsub - 1);
        const unsigned kbase = lds0 + pi_row(j) * (LDR * 2) + 16 * h;
        const unsigned vbase = lds0 + AS_VOFF + (16 * h + 4 * (gi >> 2)) * (LDR * 2) + 32 * gq + 8 * (gi & 3);
        const unsigned wbase = lds0 + (role == 1 ? AS_VOFF + key_row(sr) * (LDR * 2) : sr * (LDR * 2)) + 16 * sc;
        const int ld = role == 1 ? ldv : ldk;
        unsigned goff = (unsigned)(((32 * (AS_AHEAD - 1) + sr) * ld + 8 * sc) * 2);           // sub-tile 5 is the first one the stream loads
        const unsigned qaddr = lds0 + AS_WT0 + wave * AS_WT + j * (LDR * 2) + 16 * h, oaddr = qaddr - 8 * h;
        const unsigned lseaddr = lds0 + AS_LSE0 + (wave * 32 + j) * 4;
        unsigned mask = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) mask |= (16 * h + r >= nv) ? (1u << r) : 0u;
        const unsigned long long pb = (unsigned long long)(uintptr_t)(role == 1 ? Vb : Kb);
        const tc_i32x4 rsrc = {__builtin_amdgcn_readfirstlane((int)(unsigned)pb), __builtin_amdgcn_readfirstlane((int)(unsigned)(pb >> 32)),
                               __builtin_amdgcn_readfirstlane(((Nk - 1) * ld + D) * 2), 0x00020000};
        const int step = __builtin_amdgcn_readfirstlane(32 * ld * 2);
        const int wex = __builtin_amdgcn_readfirstlane(role < 2 ? -1 : 0);
        const long long wexec = ((long long)wex << 32) | (unsigned)wex;
        if (std::is_same<H, f16_t>::value)
            asm volatile(TC_ATTN_FWD_ASM_F16 : "+v"(goff) : "v"(kbase), "v"(vbase), "v"(wbase), "v"(qaddr), "v"(oaddr), "v"(lseaddr), "v"(mask),
                         "s"(rsrc), "s"(nsub), "s"(step), "s"(wexec) : TC_ATTN_FWD_ASM_CLOBBERS);
        else
            asm volatile(TC_ATTN_FWD_ASM_BF16 : "+v"(goff) : "v"(kbase), "v"(vbase), "v"(wbase), "v"(qaddr), "v"(oaddr), "v"(lseaddr), "v"(mask),
                         "s"(rsrc), "s"(nsub), "s"(step), "s"(wexec) : TC_ATTN_FWD_ASM_CLOBBERS);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * i + (lane >> 3);
        const uint4 v = *reinterpret_cast<const uint4*>(wtile + r * (LDR * 2) + 16 * (lane & 7));
        if (live && tq0 + r < nq) *reinterpret_cast<uint4*>(O + (trow0 + r) * ldo + 8 * (lane & 7)) = v;
    }
    if (h == 0 && live && tq0 + j < nq) lse[trow0 + j] = reinterpret_cast<const float*>(as_smem + AS_LSE0)[wave * 32 + j];
#ifdef TC_ATTN_ASM_TIMING
    if (lane == 0 && blockIdx.x < 256) {
        unsigned long long* dst = g_attn_dbg + (blockIdx.x * 12 + wave) * 8;
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(as_smem + AS_LSE0 + wave * 128 + 4096);
        dst[0] = dbg_t0;
        for (int k = 0; k < TC_ATTN_ASM_TIMING; ++k) dst[1 + k] = src[k];
        dst[1 + TC_ATTN_ASM_TIMING] = __builtin_readcyclecounter();
    }
#endif
}

// dQ.  Same organisation as the forward kernel: one 12-wave workgroup per CU, a 32-query tile per wave, K/V tiles double-buffered in
// LDS behind one barrier per tile, Q / dO / dQ tiles through the wave's LDS tile as whole 128-byte rows.
// K is stored ONCE, rows in key_row order: conflict-free both for the 16-byte fragment reads of S^T = K Q^T (row
// key_row(pi_row(j))) and for the transpose reads of dQ^T += K^T dS^T.
// O != nullptr: delta = rowsum(O dO) of the wave's 32 rows is computed HERE from the dO rows the wave loads anyway (+ the O rows) and
// written to `delta` for the dK/dV kernel, which is launched behind this one -- no separate delta pass over O and dO.
// QM = 1 (Q stored as q * scale * log2(e)): -lse * log2(e) is the INITIAL value of the S accumulator (P = exp2(S) directly), dS is
// P (dP - delta) and `scale` is applied once to the dQ accumulators.  (-delta as the initial value of dP as well -- bit 1 -- keeps a
// second 16-register tuple alive through the key loop: 23 spilled registers at the 168 of a 12-wave workgroup, 139 vs 132 us.)
template <typename H, int QM>
__global__ __launch_bounds__(FW_NT, 1) void attn_bwd_dq_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                                   const bf16_t* __restrict__ V, int ldv, long long skv,
                                                                   const bf16_t* __restrict__ O, int ldo,
                                                                   const bf16_t* __restrict__ dO, int lddo, const float* __restrict__ lse,
                                                                   float* __restrict__ delta, bf16_t* __restrict__ dQ, int lddq, Segs sg,
                                                                   int Nk, float scale, float qs, int wide_o, float* __restrict__ nstat) {
    extern __shared__ __attribute__((aligned(16))) bf16_t fw_smem[];      // [2 slots][K tile | V tile][KB][LDR], then FW_NW wave tiles [32][LDR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int nwt = sg.t32[sg.n], bpi = (nwt + FW_NW - 1) / FW_NW;
    const int bx = xcd_block(blockIdx.x, gridDim.x);
    const int b = bx / bpi, wt = (bx - b * bpi) * FW_NW + wave;
    int sgi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < sg.n && wt >= sg.t32[i]) sgi = i;
    const int nq = sg.nq[sgi];
    const int tq0 = (wt - sg.t32[sgi]) * 32;
    const bool live = wt < nwt, ok = live && tq0 + j < nq;
    const long long trow0 = (long long)sg.row0[sgi] + (long long)b * nq + tq0;
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    typedef typename TcHalf<H>::v8 V8;
    bf16_t* wtile = fw_smem + 2 * FW_SLOT + wave * (32 * LDR);
    V8 qf[4], dof[4];
    float dsum[4] = {0.f, 0.f, 0.f, 0.f};                      // O != nullptr: delta of row 8 i + (lane >> 3), complete in every lane of the row
    {   // Q, then dO, through the wave tile (coalesced rows in, fragments out)
        uint4 qv[4], gv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 8 * i + (lane >> 3);
            const bool okr = live && tq0 + r < nq;
            qv[i] = okr ? *reinterpret_cast<const uint4*>(Q + (trow0 + r) * ldq + 8 * (lane & 7)) : make_uint4(0u, 0u, 0u, 0u);
            gv[i] = okr ? *reinterpret_cast<const uint4*>(dO + (trow0 + r) * lddo + 8 * (lane & 7)) : make_uint4(0u, 0u, 0u, 0u);
            if (O) {
                const uint4 ov = okr ? *reinterpret_cast<const uint4*>(O + (trow0 + r) * ldo + 8 * (lane & 7)) : make_uint4(0u, 0u, 0u, 0u);
                const unsigned aw[4] = {ov.x, ov.y, ov.z, ov.w}, gw[4] = {gv[i].x, gv[i].y, gv[i].z, gv[i].w};
                float sd = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a0, a1, g0, g1;
                    unpack2<H>(aw[e], a0, a1);
                    unpack2<H>(gw[e], g0, g1);
                    sd = fmaf(a0, g0, fmaf(a1, g1, sd));
                }
                sd += __shfl_xor(sd, 1, 64);
                sd += __shfl_xor(sd, 2, 64);
                sd += __shfl_xor(sd, 4, 64);
                dsum[i] = sd;
                if (okr && (lane & 7) == 0) delta[trow0 + r] = sd;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(wtile + (8 * i + (lane >> 3)) * LDR + 8 * (lane & 7)) = qv[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = ld_frag<V8>(wtile + j * LDR + 16 * ks + 8 * h);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(wtile + (8 * i + (lane >> 3)) * LDR + 8 * (lane & 7)) = gv[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) dof[ks] = ld_frag<V8>(wtile + j * LDR + 16 * ks + 8 * h);
        if (O) {                                                // row deltas -> lane j through the wave tile (behind the fragment reads)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if ((lane & 7) == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) reinterpret_cast<float*>(wtile)[8 * i + (lane >> 3)] = dsum[i];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    const float l2 = ok ? lse[trow0 + j] * LOG2E : 0.f;         // qs = scale * log2(e), or 1 when Q arrives scaled
    const float dlt = O ? reinterpret_cast<const float*>(wtile)[j] : (ok ? delta[trow0 + j] : 0.f);
    const float dls = ok ? dlt * scale : 0.f;                   // dS = P (dP scale - delta scale)
    const float dl0 = ok ? dlt : 0.f;
    // the hand-scheduled dK/dV stream reads the row statistics per 32-query TILE: [image][tile][-lse log2(e) x 32 | -delta x 32], rows past
    // the end of a segment as (-1e30, 0) so that their P is exactly 0
    if (nstat && live && h == 0) {
        float* st = nstat + ((long long)b * nwt + wt) * 64;
        st[j] = ok ? -l2 : NEG_BIG;
        st[32 + j] = -dl0;
    }
    constexpr bool QSC = QM != 0;                               // QM bit 0: -l2 is the C operand of S; bit 1: -delta the C operand of dP
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int krow = pi_row(j), kprow = key_row(krow);
    uint4 st[FW_NF];
    auto fetch = [&](int kb0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < FW_NF; ++i) {
            const int id = tid + i * FW_NT, isv = id >= FW_NC, r = (id - isv * FW_NC) >> 3, c8 = (id & 7) * 8;
            const bf16_t* src = isv ? Vb : Kb;
            const int ld = isv ? ldv : ldk;
            st[i] = (id < 2 * FW_NC && kb0 + r < Nk) ? *reinterpret_cast<const uint4*>(src + (long long)(kb0 + r) * ld + c8) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto stash = [&](int slot) __attribute__((always_inline)) {
        bf16_t* base = fw_smem + slot * FW_SLOT;
#pragma unroll
        for (int i = 0; i < FW_NF; ++i) {
            const int id = tid + i * FW_NT, isv = id >= FW_NC, r = (id - isv * FW_NC) >> 3, c8 = (id & 7) * 8;
            if (id < 2 * FW_NC) *reinterpret_cast<uint4*>(base + isv * (KB * LDR) + (isv ? r : key_row(r)) * LDR + c8) = st[i];
        }
    };
    const int nt = (Nk + KB - 1) / KB;
    fetch(0);
    stash(0);
    if (nt > 1) fetch(KB);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int kb0 = t * KB;
        if (t + 1 < nt) stash((t + 1) & 1);
        if (t + 2 < nt) fetch(kb0 + 2 * KB);
        const bf16_t* Ks = fw_smem + (t & 1) * FW_SLOT;
        const bf16_t* Vs = Ks + KB * LDR;
        const int nsub = min(KB / 32, (Nk - kb0 + 31) / 32);
        // (issuing the S / dP products of sub-tile sub+1 before the softmax arithmetic of sub-tile sub -- a source-level software
        // pipeline -- needs 32-64 more live accumulator registers than the 168 of a 12-wave workgroup: the compiler spills 28-62
        // registers and the kernel runs 63 us instead of 44)
#pragma unroll 1
        for (int sub = 0; sub < nsub; ++sub) {
            const int kv0 = kb0 + 32 * sub;
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = (QM & 1) ? -l2 : 0.f; dp[r] = (QM & 2) ? -dl0 : 0.f; }
            const bf16_t* kp = Ks + (32 * sub + kprow) * LDR + 8 * h;
            const bf16_t* vp = Vs + (32 * sub + krow) * LDR + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = TcHalf<H>::mfma(ld_frag<V8>(kp + 16 * ks), qf[ks], s);
                dp = TcHalf<H>::mfma(ld_frag<V8>(vp + 16 * ks), dof[ks], dp);
            }
            const bool tail = kv0 + 32 > Nk;                    // keys past Nk were staged as zeros: P is forced to 0 for them
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = (QM & 1) ? fast_exp2(s[r]) : (QSC ? fast_exp2(s[r] - l2) : fast_exp2(fmaf(s[r], qs, -l2)));
                if (tail && kv0 + 16 * h + r >= Nk) p = 0.f;
                s[r] = (QM & 2) ? p * dp[r] : (QSC ? p * (dp[r] - dl0) : p * fmaf(dp[r], scale, -dls));
            }
            const int gi = lane & 15, gq = (lane >> 4) & 1;
            const bf16_t* kt = Ks + (32 * sub + 16 * h + 4 * (gi >> 2)) * LDR + 16 * gq + 4 * (gi & 3);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const V8 db = pack8<H>(s, 8 * k2);
                acc0 = TcHalf<H>::mfma(ld_frag_tr<V8>(kt + (2 * k2) * LDR, kt + (2 * k2 + 1) * LDR), db, acc0);
                acc1 = TcHalf<H>::mfma(ld_frag_tr<V8>(kt + (2 * k2) * LDR + 32, kt + (2 * k2 + 1) * LDR + 32), db, acc1);
            }
        }
        __syncthreads();
    }
    if (QSC) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] *= scale; acc1[r] *= scale; }
    }
    if (wide_o) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<H>(reinterpret_cast<H*>(wtile + j * LDR + 8 * g + 4 * h), make_float4(acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]));
            st4<H>(reinterpret_cast<H*>(wtile + j * LDR + 32 + 8 * g + 4 * h), make_float4(acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 8 * i + (lane >> 3);
            const uint4 v = *reinterpret_cast<const uint4*>(wtile + r * LDR + 8 * (lane & 7));
            if (live && tq0 + r < nq) *reinterpret_cast<uint4*>(dQ + (trow0 + r) * lddq + 8 * (lane & 7)) = v;
        }
    } else if (ok) {
        bf16_t* row = dQ + (trow0 + j) * lddq;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            st4<H>(reinterpret_cast<H*>(row + 8 * g + 4 * h), make_float4(acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]));
            st4<H>(reinterpret_cast<H*>(row + 32 + 8 * g + 4 * h), make_float4(acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]));
        }
    }
}

// dK/dV: workgroup = (NW x 32 keys of one image, one chunk of that image's 32-query tiles).  Every wave keeps the K and V
// fragments of its own 32 keys in registers and accumulates their dK^T / dV^T tiles; the Q / dO rows of a 64-query stage are
// staged ONCE per workgroup (row-major for the S and dP products, transposed for the dV^T / dK^T products) and shared by all
// waves -- the first version staged them per wave (4x the LDS stores and global reads) and needed 288 VGPRs (one wave per SIMD).
// Query chunks (gridDim.z) add their partial sums atomically into the fp32 scratch dkv32 [B][Nk][128] (dK | dV), which
// attn_dkv_store_kernel converts to the bf16 outputs.
#ifdef TC_DKV_TIMING
__device__ unsigned long long g_dkv_dbg[512 * 4];
#define DSTAMP(k) do { if (threadIdx.x == 0) { const long long t_ = __builtin_readcyclecounter(); g_dkv_dbg[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) % 512 * 4 + (k)] = (unsigned long long)(t_ - dt_); dt_ = t_; } } while (0)
#define DSTAMP_INIT() long long dt_ = __builtin_readcyclecounter()
#else
#define DSTAMP(k)
#define DSTAMP_INIT()
#endif
// QSC (Q stored as q * scale * log2(e), `qs` = 1): the stage keeps -lse * log2(e) and -delta, and they are the INITIAL values of
// the S and dP accumulators -- P = exp2(S), dS = P dP' with no multiply-add per score (a third of the kernel's VALU work, which
// shares the SIMD's issue slots with the MFMAs); the factor ln 2 of dK (`scale`) is applied once to the accumulators at the end.
#ifndef DKV_PIPE
#define DKV_PIPE 0
#endif
#ifndef DKV_QS
#define DKV_QS 64                                               // queries per stage of the dK/dV kernel (whole 32-query tiles)
#endif
constexpr int DKV_STAGE_B = 2 * (2 * DKV_QS * (D + 8) + 2 * DKV_QS * 2) * 2;
constexpr int DKV_SMEM_B = DKV_STAGE_B > 4 * D * 33 * 4 ? DKV_STAGE_B : 4 * D * 33 * 4;
template <typename H, int NW, bool QSC>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_dkv_seg_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                                    const bf16_t* __restrict__ V, int ldv, long long skv,
                                                                    const bf16_t* __restrict__ dO, int lddo, const float* __restrict__ lse,
                                                                    const float* __restrict__ delta, float* __restrict__ dkv32, Segs sg, int Nk,
                                                                    float scale, float qs, int tiles_per_chunk) {
    constexpr int QS = DKV_QS, NT = QS / 32, NI = QS * 8 / 256, LDQ = D + 8;
    // Q and dO of a stage are stored ONCE, row-major with the rows of every 16-row group 4x4-transposed (key_row): conflict-free both
    // for the 16-byte fragment reads of S = Q K^T / dP = dO V^T and for the hardware transpose reads (ds_read_b64_tr_b16) that gather
    // the dO^T / Q^T operands of the dV^T / dK^T products -- the first version kept transposed copies written with 2-byte stores
    // two stage buffers: the next 64-query stage is stored (from registers loaded a stage earlier) while the current one is consumed,
    // so a stage costs ONE barrier
    constexpr int STAGE_E = 2 * QS * LDQ + 2 * QS * 2;          // bf16 elements per buffer: Q | dO | lse (fp32) | delta (fp32)
    static_assert(2 * STAGE_E * 2 == DKV_STAGE_B && NW * D * 33 * 4 <= DKV_SMEM_B, "host-side LDS size");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* const sbase = reinterpret_cast<bf16_t*>(smem);
    DSTAMP_INIT();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int lin = xcd_block((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, gridDim.x * gridDim.y * gridDim.z);
    const int bkx = lin % gridDim.x, b = (lin / gridDim.x) % gridDim.y, bkz = lin / (gridDim.x * gridDim.y);
    const int kv0 = (bkx * NW + wave) * 32, key = min(kv0 + j, Nk - 1);
    typedef typename TcHalf<H>::v8 V8;
    V8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {           // keys past Nk: a duplicate row whose results are never stored
        const uint4 a = *reinterpret_cast<const uint4*>(K + b * skv + (long long)key * ldk + 16 * ks + 8 * h);
        const uint4 c = *reinterpret_cast<const uint4*>(V + b * skv + (long long)key * ldv + 16 * ks + 8 * h);
        kf[ks] = *reinterpret_cast<const V8*>(&a);
        vf[ks] = *reinterpret_cast<const V8*>(&c);
    }
    // qs = scale * log2(e) and dS = P (dP - delta) scale multiplies the UNSCALED Q into dK; with Q stored as q * qs (qs passed as 1)
    // the factor of dS is ln 2 = scale / (scale * log2(e)) instead (`scale` is passed as ln 2 by the host in that case)
    f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk0[r] = dk1[r] = dv0[r] = dv1[r] = 0.f; }
    const int qrow = pi_row(j);
    const int ntiles = sg.t32[sg.n];
    const int t_begin = bkz * tiles_per_chunk, t_end = min(ntiles, t_begin + tiles_per_chunk);
    auto locate = [&](int t, long long& base, int& valid) {       // first row and number of valid rows of 32-query tile t
        int s = 0;
#pragma unroll
        for (int i = 1; i < 4; ++i) if (i < sg.n && t >= sg.t32[i]) s = i;
        const int q0 = (t - sg.t32[s]) * 32;
        base = (long long)sg.row0[s] + (long long)b * sg.nq[s] + q0;
        valid = (t < t_end) ? min(32, sg.nq[s] - q0) : 0;
    };
    // staging (threads 0..255): 16-byte chunk c8 of row r of the 64-query stage
    const bool stager = tid < 256;
    uint4 qr[NI], gr[NI];
    float lr = 0.f, dr = 0.f;
    auto fmap = [&](int it, int& r, int& c8, int& g) {           // eight lanes x 16 bytes cover one 128-byte row: whole lines per request
        const int idx = it * 256 + tid;
        g = 0; r = idx >> 3; c8 = (idx & 7) * 8;
    };
    auto fetch = [&](int t0) {
        if (stager) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                int r, c8, g; fmap(i, r, c8, g);
                long long base; int valid;
                locate(t0 + (r >> 5), base, valid);
                const bool okr = (r & 31) < valid;
                qr[i] = okr ? *reinterpret_cast<const uint4*>(Q + (base + (r & 31)) * ldq + c8) : make_uint4(0u, 0u, 0u, 0u);
                gr[i] = okr ? *reinterpret_cast<const uint4*>(dO + (base + (r & 31)) * lddo + c8) : make_uint4(0u, 0u, 0u, 0u);
            }
        }
        if (tid < QS) {
            long long base; int valid;
            locate(t0 + (tid >> 5), base, valid);
            const bool okr = (tid & 31) < valid;
            lr = okr ? lse[base + (tid & 31)] * LOG2E : 1.0e30f;          // rows past the end: P = exp2(s - 1e30) = 0
            dr = okr ? delta[base + (tid & 31)] * (QSC ? 1.0f : scale) : 0.f;
            if (QSC) { lr = -lr; dr = -dr; }
        }
    };
    auto stash = [&](int buf) {
        bf16_t* Qw = sbase + buf * STAGE_E;
        bf16_t* dOw = Qw + QS * LDQ;
        float* lw = reinterpret_cast<float*>(dOw + QS * LDQ);
        if (stager) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                int r, c8, g; fmap(i, r, c8, g);
                *reinterpret_cast<uint4*>(&Qw[key_row(r) * LDQ + c8]) = qr[i];
                *reinterpret_cast<uint4*>(&dOw[key_row(r) * LDQ + c8]) = gr[i];
            }
        }
        if (tid < QS) { lw[tid] = lr; lw[QS + tid] = dr; }
    };
    if (t_begin < t_end) {
        fetch(t_begin);
        stash(0);
        if (t_begin + NT < t_end) fetch(t_begin + NT);
    }
    __syncthreads();
    DSTAMP(0);
    int buf = 0;
    for (int t0 = t_begin; t0 < t_end; t0 += NT, buf ^= 1) {
        if (t0 + NT < t_end) stash(buf ^ 1);                    // (every wave left the stage before, that buffer's last reader, at the barrier below)
        if (t0 + 2 * NT < t_end) fetch(t0 + 2 * NT);
        const bf16_t* Qs = sbase + buf * STAGE_E;
        const bf16_t* dOs = Qs + QS * LDQ;
        const float* lss = reinterpret_cast<const float*>(dOs + QS * LDQ);
        const float* dls = lss + QS;
        // S / dP of 32-query tile qt + 1 are issued BEFORE the exp2 / multiply / pack / dV^T / dK^T work of tile qt (DKV_PIPE, within a
        // stage): two waves per SIMD do not cover each other's dependent chains, the wave's own next tile does
        auto sdp = [&](int qt, f32x16& s, f32x16& dp, f32x16& lq, f32x16& dq) __attribute__((always_inline)) {
            // the 16 log-sum-exps and (scaled) deltas of this lane's query rows: four 16-byte LDS reads each instead of 16 scalar ones
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 a = *reinterpret_cast<const float4*>(lss + 32 * qt + 16 * h + 4 * g);
                const float4 c = *reinterpret_cast<const float4*>(dls + 32 * qt + 16 * h + 4 * g);
                lq[4 * g] = a.x; lq[4 * g + 1] = a.y; lq[4 * g + 2] = a.z; lq[4 * g + 3] = a.w;
                dq[4 * g] = c.x; dq[4 * g + 1] = c.y; dq[4 * g + 2] = c.z; dq[4 * g + 3] = c.w;
            }
            if (QSC) { s = lq; dp = dq; }                        // accumulators start at -lse log2(e) and -delta
            else {
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
            }
            const bf16_t* qp = Qs + (32 * qt + key_row(qrow)) * LDQ + 8 * h;
            const bf16_t* gp = dOs + (32 * qt + key_row(qrow)) * LDQ + 8 * h;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = TcHalf<H>::mfma(ld_frag<V8>(qp + 16 * ks), kf[ks], s);
                dp = TcHalf<H>::mfma(ld_frag<V8>(gp + 16 * ks), vf[ks], dp);
            }
        };
        auto fold = [&](int qt, f32x16& s, f32x16& dp, const f32x16& lq, const f32x16& dq) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (QSC) {
                    const float p = fast_exp2(s[r]);
                    dp[r] *= p;                                  // dS / ln 2 = P (dP - delta)
                    s[r] = p;
                } else {
                    const float p = fast_exp2(fmaf(s[r], qs, -lq[r]));
                    dp[r] = p * fmaf(dp[r], scale, -dq[r]);      // dS = P (dP - delta) scale, delta pre-multiplied by scale at staging
                    s[r] = p;
                }
            }
            const int gi = lane & 15, gq = (lane >> 4) & 1, toff = (32 * qt + 16 * h + 4 * (gi >> 2)) * LDQ + 16 * gq + 4 * (gi & 3);
            const bf16_t* gt = dOs + toff;
            const bf16_t* qtp = Qs + toff;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const V8 pb = pack8<H>(s, 8 * k2), db = pack8<H>(dp, 8 * k2);
                dv0 = TcHalf<H>::mfma(ld_frag_tr<V8>(gt + (2 * k2) * LDQ, gt + (2 * k2 + 1) * LDQ), pb, dv0);
                dv1 = TcHalf<H>::mfma(ld_frag_tr<V8>(gt + (2 * k2) * LDQ + 32, gt + (2 * k2 + 1) * LDQ + 32), pb, dv1);
                dk0 = TcHalf<H>::mfma(ld_frag_tr<V8>(qtp + (2 * k2) * LDQ, qtp + (2 * k2 + 1) * LDQ), db, dk0);
                dk1 = TcHalf<H>::mfma(ld_frag_tr<V8>(qtp + (2 * k2) * LDQ + 32, qtp + (2 * k2 + 1) * LDQ + 32), db, dk1);
            }
        };
#if DKV_PIPE
        f32x16 sv[2], dpv[2], lqv[2], dqv[2];
        sdp(0, sv[0], dpv[0], lqv[0], dqv[0]);
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) {
            if (qt + 1 < NT) sdp(qt + 1, sv[(qt + 1) & 1], dpv[(qt + 1) & 1], lqv[(qt + 1) & 1], dqv[(qt + 1) & 1]);
            fold(qt, sv[qt & 1], dpv[qt & 1], lqv[qt & 1], dqv[qt & 1]);
        }
#else
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) {
            f32x16 s, dp, lq, dq;
            sdp(qt, s, dp, lq, dq);
            fold(qt, s, dp, lq, dq);
        }
#endif
        __syncthreads();
    }
    DSTAMP(1);
    // each wave owns its keys: its [d][key] accumulators pass through LDS so that the stores run along d (whole 256-byte rows).
    // Every query chunk (blockIdx.z) writes its own partial [B][Nk][128]; attn_dkv_store_kernel adds the chunks -- fp32 atomics into
    // one buffer cost this epilogue 20 k of the kernel's 149 k cycles (scripts/exp/dkv_timing.py) plus a zero-fill launch.
    float* red = reinterpret_cast<float*>(smem) + wave * (D * 33);
    float* dpart = dkv32 + (long long)bkz * gridDim.y * Nk * 128;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = d_row(r, h);
            red[d * 33 + j] = which ? dv0[r] : dk0[r] * (QSC ? scale : 1.0f);
            red[(32 + d) * 33 + j] = which ? dv1[r] : dk1[r] * (QSC ? scale : 1.0f);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        for (int f = lane; f < 32 * D; f += 64) {
            const int kk = f >> 6, d = f & 63;
            if (kv0 + kk < Nk) dpart[((long long)b * Nk + kv0 + kk) * 128 + which * 64 + d] = red[d * 33 + kk];
        }
    }
    DSTAMP(2);
}

// dQ, hand-scheduled (gen_dq_asm.py; Q stored scaled, 16-byte addressable rows, at least two key sub-tiles): the forward stream's workgroup
// and ring (attn_fwd_asm_kernel), K rows in key_row order and V rows in natural order.  This function computes the row deltas (and the
// per-tile statistics of the dK/dV stream), stages sub-tiles 0..4, runs the stream and stores the dQ tile it leaves in the wave's LDS tile.
#include "attn_dq_asm.inc"
template <typename H>
__global__ __launch_bounds__(AS_NW * 64, 1) void attn_bwd_dq_asm_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                                        const bf16_t* __restrict__ V, int ldv, long long skv,
                                                                        const bf16_t* __restrict__ O, int ldo, const bf16_t* __restrict__ dO, int lddo,
                                                                        const float* __restrict__ lse, float* __restrict__ delta,
                                                                        bf16_t* __restrict__ dQ, int lddq, Segs sg, int Nk, float scale,
                                                                        float* __restrict__ nstat, long long rows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char as_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int nwt = sg.t32[sg.n], bpi = (nwt + AS_NW - 1) / AS_NW;
    const int bx = xcd_block(blockIdx.x, gridDim.x);
    const int b = bx / bpi, wt = (bx - b * bpi) * AS_NW + wave;
    int sgi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < sg.n && wt >= sg.t32[i]) sgi = i;
    const int nq = sg.nq[sgi];
    const int tq0 = (wt - sg.t32[sgi]) * 32;
    const bool live = wt < nwt, ok = live && tq0 + j < nq;
    const long long trow0 = (long long)sg.row0[sgi] + (long long)b * nq + tq0;
    const bf16_t* Kb = K + b * skv;
    const bf16_t* Vb = V + b * skv;
    unsigned char* wtile = as_smem + AS_WT0 + wave * AS_WT;
    {   // delta = rowsum(O dO) of the wave's rows: eight lanes x 16 bytes per row, the sums meet in the wave tile
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 8 * i + (lane >> 3);
            const bool okr = live && tq0 + r < nq;
            const uint4 gv = okr ? *reinterpret_cast<const uint4*>(dO + (trow0 + r) * lddo + 8 * (lane & 7)) : make_uint4(0u, 0u, 0u, 0u);
            const uint4 ov = okr ? *reinterpret_cast<const uint4*>(O + (trow0 + r) * ldo + 8 * (lane & 7)) : make_uint4(0u, 0u, 0u, 0u);
            const unsigned aw[4] = {ov.x, ov.y, ov.z, ov.w}, gw[4] = {gv.x, gv.y, gv.z, gv.w};
            float sd = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a0, a1, g0, g1;
                unpack2<H>(aw[e], a0, a1);
                unpack2<H>(gw[e], g0, g1);
                sd = fmaf(a0, g0, fmaf(a1, g1, sd));
            }
            sd += __shfl_xor(sd, 1, 64);
            sd += __shfl_xor(sd, 2, 64);
            sd += __shfl_xor(sd, 4, 64);
            if (okr && (lane & 7) == 0) delta[trow0 + r] = sd;
            if ((lane & 7) == 0) reinterpret_cast<float*>(wtile)[r] = sd;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const float dl0 = ok ? reinterpret_cast<const float*>(wtile)[j] : 0.f;
    const float nl2 = ok ? -(lse[trow0 + j] * LOG2E) : 0.f;
    if (nstat && live && h == 0) {
        float* st = nstat + ((long long)b * nwt + wt) * 64;
        st[j] = ok ? nl2 : NEG_BIG;
        st[32 + j] = -dl0;
    }
    const int role = wave >> 2, sr = (tid & 255) >> 3, sc = tid & 7;
    if (role < 2) {
        const bf16_t* src = role ? Vb : Kb;
        const int ld = role ? ldv : ldk;
#pragma unroll
        for (int s = 0; s < AS_AHEAD - 1; ++s) {                // sub-tiles 0..4 into ring slots 0..4 (rows past Nk: a duplicate, masked by the stream)
            const uint4 x = *reinterpret_cast<const uint4*>(src + (long long)min(32 * s + sr, Nk - 1) * ld + 8 * sc);
            *reinterpret_cast<uint4*>(as_smem + s * AS_SLOT + (role ? AS_VOFF + sr * (LDR * 2) : key_row(sr) * (LDR * 2)) + 16 * sc) = x;
        }
    }
    __syncthreads();
    {
        const unsigned lds0 = (unsigned)(uintptr_t)as_smem;
        const int gi = lane & 15, gq = (lane >> 4) & 1, nsub = (Nk + 31) / 32, nv = Nk - 32 * (nsub - 1);
        const unsigned kbase = lds0 + key_row(pi_row(j)) * (LDR * 2) + 16 * h;
        const unsigned vbase = lds0 + AS_VOFF + pi_row(j) * (LDR * 2) + 16 * h;
        const unsigned tbase = lds0 + (16 * h + 4 * (gi >> 2)) * (LDR * 2) + 32 * gq + 8 * (gi & 3);
        const unsigned wbase = lds0 + (role == 1 ? AS_VOFF + sr * (LDR * 2) : key_row(sr) * (LDR * 2)) + 16 * sc;
        const int ld = role == 1 ? ldv : ldk;
        unsigned goff = (unsigned)(((32 * (AS_AHEAD - 1) + sr) * ld + 8 * sc) * 2);
        const long long qrow = live ? trow0 + min(j, max(0, min(32, nq - tq0) - 1)) : 0;     // rows past the tile's end: a valid row whose column is never stored
        const unsigned qoff = (unsigned)((qrow * ldq + 8 * h) * 2), gooff = (unsigned)((qrow * lddo + 8 * h) * 2);
        const unsigned oaddr = lds0 + AS_WT0 + wave * AS_WT + j * (LDR * 2) + 8 * h;
        unsigned mask = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) mask |= (16 * h + r >= nv) ? (1u << r) : 0u;
        auto mk = [](const void* p, long long bytes) {
            const unsigned long long a = (unsigned long long)(uintptr_t)p;
            return tc_i32x4{__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)),
                            __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
        };
        const tc_i32x4 rsrc = mk(role == 1 ? Vb : Kb, ((long long)(Nk - 1) * ld + D) * 2);
        const tc_i32x4 rq = mk(Q, ((rows - 1) * ldq + D) * 2), rg = mk(dO, ((rows - 1) * lddo + D) * 2);
        const int step = __builtin_amdgcn_readfirstlane(32 * ld * 2);
        const int wex = __builtin_amdgcn_readfirstlane(role < 2 ? -1 : 0);
        const long long wexec = ((long long)wex << 32) | (unsigned)wex;
        const int scl = __builtin_amdgcn_readfirstlane(__float_as_int(scale));
        if (std::is_same<H, f16_t>::value)
            asm volatile(TC_ATTN_DQ_ASM_F16 : "+v"(goff) : "v"(kbase), "v"(vbase), "v"(tbase), "v"(wbase), "v"(qoff), "v"(gooff), "v"(oaddr), "v"(nl2), "v"(dl0),
                         "v"(mask), "s"(rsrc), "s"(rq), "s"(rg), "s"(nsub), "s"(step), "s"(wexec), "s"(scl) : TC_ATTN_DQ_ASM_CLOBBERS);
        else
            asm volatile(TC_ATTN_DQ_ASM_BF16 : "+v"(goff) : "v"(kbase), "v"(vbase), "v"(tbase), "v"(wbase), "v"(qoff), "v"(gooff), "v"(oaddr), "v"(nl2), "v"(dl0),
                         "v"(mask), "s"(rsrc), "s"(rq), "s"(rg), "s"(nsub), "s"(step), "s"(wexec), "s"(scl) : TC_ATTN_DQ_ASM_CLOBBERS);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * i + (lane >> 3);
        const uint4 v = *reinterpret_cast<const uint4*>(wtile + r * (LDR * 2) + 16 * (lane & 7));
        if (live && tq0 + r < nq) *reinterpret_cast<uint4*>(dQ + (trow0 + r) * lddq + 8 * (lane & 7)) = v;
    }
}

// dK/dV, hand-scheduled (gen_dkv_asm.py; Q stored scaled, 16-byte addressable rows): the decomposition of attn_bwd_dkv_seg_kernel -- workgroup =
// (4 x 32 keys of one image, one chunk of that image's 32-query tiles), every wave owns 32 keys -- with the 32-query sub-tiles of
// Q | dO | statistics streaming through an 8-slot LDS ring (buffer loads six sub-tiles ahead, one barrier per two sub-tiles).  This
// function stages sub-tiles 0..4 and the parameter block, runs the stream and stores the [d][key] fp32 tiles it leaves in LDS.
#include "attn_dkv_asm.inc"
constexpr int DA_TILE = 32 * LDR * 2, DA_SLOT = 2 * DA_TILE + 256, DA_NSLOT = 8, DA_AHEAD = 6, DA_PRM0 = DA_NSLOT * DA_SLOT, DA_SMEM = DA_PRM0 + 64;
constexpr int DA_RED = D * 33 * 4;
static_assert(4 * 2 * DA_RED <= DA_NSLOT * DA_SLOT, "the accumulator tiles of the four waves fit the ring");
// PAIR = 2: an 8-wave workgroup whose two halves take two query chunks of the same 128 keys -- each half with its own ring and parameter
// block, the barriers of the stream shared (the chunks are equally long by construction: see the host) -- and add their accumulator tiles
// through LDS before the store: half the fp32 partial traffic.
constexpr int DA_HALF = (DA_SMEM + 15) / 16 * 16;
template <typename H, int PAIR>
__global__ __launch_bounds__(256 * PAIR, 3 - PAIR) void attn_bwd_dkv_asm_kernel(const bf16_t* __restrict__ Q, int ldq, const bf16_t* __restrict__ K, int ldk,
                                                                    const bf16_t* __restrict__ V, int ldv, long long skv,
                                                                    const bf16_t* __restrict__ dO, int lddo, const float* __restrict__ nstat,
                                                                    float* __restrict__ dkv32, Segs sg, int Nk, float kscale, int tiles_per_chunk,
                                                                    long long rows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char da_smem_all[];
    const int half = PAIR == 2 ? (int)(threadIdx.x >> 8) : 0;
    unsigned char* const da_smem = da_smem_all + half * DA_HALF;
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int lin = xcd_block((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, gridDim.x * gridDim.y * gridDim.z);
    const int bkx = lin % gridDim.x, b = (lin / gridDim.x) % gridDim.y, bkz = lin / (gridDim.x * gridDim.y);
    const int kv0 = (bkx * 4 + wave) * 32, key = min(kv0 + j, Nk - 1);
    const int ntiles = sg.t32[sg.n];
    // PAIR: every chunk is tiles_per_chunk long; the last one starts early and its first t_mask tiles -- the previous chunk's -- count for
    // nothing: their statistics are staged as the padding (-1e30, 0), so P = exp2(S - 1e30) = 0 exactly (the host keeps t_mask below the 5 staged here)
    const int chunk = PAIR * bkz + half;
    const int t_begin = PAIR == 2 ? min(chunk * tiles_per_chunk, ntiles - tiles_per_chunk) : bkz * tiles_per_chunk;
    const int t_mask = PAIR == 2 ? chunk * tiles_per_chunk - t_begin : 0;
    const int nsub = PAIR == 2 ? tiles_per_chunk : min(ntiles, t_begin + tiles_per_chunk) - t_begin;
    // first row of tile t of this image = A_s + 32 t, s = the segment of t
    int As[4], Ts[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        As[i] = i < sg.n ? sg.row0[i] + b * sg.nq[i] - 32 * sg.t32[i] : 0;
        Ts[i] = i < sg.n ? sg.t32[i] : 0x7fffffff;
    }
    const int sr = tid >> 3, sc = tid & 7;
#pragma unroll
    for (int s = 0; s < DA_AHEAD - 1; ++s) {                    // sub-tiles 0..4 into ring slots 0..4
        const int t = t_begin + s;
        int a = As[0];
#pragma unroll
        for (int i = 1; i < 4; ++i) if (t >= Ts[i]) a = As[i];
        const long long row = (long long)a + 32LL * t + sr;
        const bool in = row >= 0 && row < rows;
        const uint4 qv = in ? *reinterpret_cast<const uint4*>(Q + row * ldq + 8 * sc) : make_uint4(0u, 0u, 0u, 0u);
        const uint4 gv = in ? *reinterpret_cast<const uint4*>(dO + row * lddo + 8 * sc) : make_uint4(0u, 0u, 0u, 0u);
        unsigned char* slot = da_smem + s * DA_SLOT;
        *reinterpret_cast<uint4*>(slot + key_row(sr) * (LDR * 2) + 16 * sc) = qv;
        *reinterpret_cast<uint4*>(slot + DA_TILE + key_row(sr) * (LDR * 2) + 16 * sc) = gv;
        if (tid < 16) {
            uint4 sv = t < ntiles ? *reinterpret_cast<const uint4*>(nstat + ((long long)b * ntiles + t) * 64 + 4 * tid) : make_uint4(0u, 0u, 0u, 0u);
            if (s < t_mask) {                                    // [-lse log2e x 32 | -delta x 32] of a tile that counts for nothing
                const unsigned pad = tid < 8 ? __float_as_uint(-1e30f) : 0u;
                sv = make_uint4(pad, pad, pad, pad);
            }
            *reinterpret_cast<uint4*>(slot + 2 * DA_TILE + 16 * tid) = sv;
        }
    }
    if (tid < 16) {
        int v = 0;
        switch (tid) {
            case 0: v = nsub; break;
            case 1: v = t_begin + DA_AHEAD - 1; break;
            case 2: v = (b * ntiles + t_begin + DA_AHEAD - 1) * 256; break;
            case 3: v = Ts[1]; break;
            case 4: v = Ts[2]; break;
            case 5: v = Ts[3]; break;
            case 6: v = As[0]; break;
            case 7: v = As[1]; break;
            case 8: v = As[2]; break;
            case 9: v = As[3]; break;
            case 10: v = ldq * 2; break;
            case 11: v = lddo * 2; break;
            case 12: v = __float_as_int(kscale); break;
            default: break;
        }
        reinterpret_cast<int*>(da_smem + DA_PRM0)[tid] = v;
    }
    __syncthreads();
    {
        const unsigned lds0 = (unsigned)(uintptr_t)da_smem;
        const int gi = lane & 15, gq = (lane >> 4) & 1;
        const unsigned vq = (unsigned)((sr * ldq + 8 * sc) * 2), vg = (unsigned)((sr * lddo + 8 * sc) * 2), vs = (unsigned)(16 * (tid & 15));
        const unsigned wq = lds0 + key_row(sr) * (LDR * 2) + 16 * sc, ws = lds0 + 2 * DA_TILE + 16 * (tid & 15);
        const unsigned rb = lds0 + key_row(pi_row(j)) * (LDR * 2) + 16 * h;
        const unsigned tb = lds0 + (16 * h + 4 * (gi >> 2)) * (LDR * 2) + 32 * gq + 8 * (gi & 3);
#ifdef TC_DKV_B128STATS
        const unsigned sb = lds0 + 2 * DA_TILE + 64 * h;
#else
        const unsigned sb = lds0 + 2 * DA_TILE + 4 * (16 * h + (lane & 15));
#endif
        const unsigned ko = (unsigned)((key * ldk + 8 * h) * 2), vo = (unsigned)((key * ldv + 8 * h) * 2);
        const unsigned red = lds0 + wave * (2 * DA_RED) + (4 * h * 33 + j) * 4;
        const unsigned prm = lds0 + DA_PRM0;
        auto mk = [](const void* p, long long bytes) {
            const unsigned long long a = (unsigned long long)(uintptr_t)p;
            return tc_i32x4{__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)),
                            __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
        };
        const tc_i32x4 rq = mk(Q, ((rows - 1) * ldq + D) * 2), rg = mk(dO, ((rows - 1) * lddo + D) * 2);
        const tc_i32x4 rs = mk(nstat, (long long)gridDim.y * ntiles * 256);
        const tc_i32x4 rk = mk(K + b * skv, ((long long)(Nk - 1) * ldk + D) * 2), rv = mk(V + b * skv, ((long long)(Nk - 1) * ldv + D) * 2);
        const int smlo = __builtin_amdgcn_readfirstlane(wave == 0 ? 0xffff : 0);
        const long long smask = (long long)(unsigned)smlo;
        if (std::is_same<H, f16_t>::value)
            asm volatile(TC_ATTN_DKV_ASM_F16 : : "v"(vq), "v"(vg), "v"(vs), "v"(wq), "v"(ws), "v"(rb), "v"(tb), "v"(sb), "v"(ko), "v"(vo), "v"(red), "v"(prm),
                         "s"(rq), "s"(rg), "s"(rs), "s"(rk), "s"(rv), "s"(smask) : TC_ATTN_DKV_ASM_CLOBBERS);
        else
            asm volatile(TC_ATTN_DKV_ASM_BF16 : : "v"(vq), "v"(vg), "v"(vs), "v"(wq), "v"(ws), "v"(rb), "v"(tb), "v"(sb), "v"(ko), "v"(vo), "v"(red), "v"(prm),
                         "s"(rq), "s"(rg), "s"(rs), "s"(rk), "s"(rv), "s"(smask) : TC_ATTN_DKV_ASM_CLOBBERS);
    }
    // each wave owns its keys: whole 256-byte rows of its [key][d] partial out of the [d][key] tiles the stream left in LDS
    const float* redw = reinterpret_cast<const float*>(da_smem + wave * (2 * DA_RED));
    float* dpart = dkv32 + (long long)bkz * gridDim.y * Nk * 128;
    if (PAIR == 2) {                                            // the two halves' tiles added; half 0 stores dK, half 1 dV
        __syncthreads();
        const float* red0 = reinterpret_cast<const float*>(da_smem_all + wave * (2 * DA_RED));
        const float* red1 = reinterpret_cast<const float*>(da_smem_all + DA_HALF + wave * (2 * DA_RED));
        const int which = half;
        for (int f = lane; f < 32 * D; f += 64) {
            const int kk = f >> 6, d = f & 63, o = which * (D * 33) + d * 33 + kk;
            if (kv0 + kk < Nk) dpart[((long long)b * Nk + kv0 + kk) * 128 + which * 64 + d] = red0[o] + red1[o];
        }
        return;
    }
#pragma unroll
    for (int which = 0; which < 2; ++which)
        for (int f = lane; f < 32 * D; f += 64) {
            const int kk = f >> 6, d = f & 63;
            if (kv0 + kk < Nk) dpart[((long long)b * Nk + kv0 + kk) * 128 + which * 64 + d] = redw[which * (D * 33) + d * 33 + kk];
        }
}

template <typename H>
__global__ __launch_bounds__(256) void attn_dkv_store_kernel(const float* __restrict__ dkv32, bf16_t* __restrict__ dK, int lddk,
                                                             bf16_t* __restrict__ dV, int lddv, long long sdkv, int B, int Nk, int zs) {
    const long long n = (long long)B * Nk * 32;                 // float4 groups per query chunk
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int q = (int)(i & 31), key = (int)((i >> 5) % Nk), b = (int)((i >> 5) / Nk);
        // the query chunks' partial sums, added as a fixed binary tree over (up to) TC_ATTN_DKV_SPLITS = 8 leaves, absent ones zero: the paired
        // dK/dV kernel hands over leaves already added two by two and must land on the same bits
        float4 t[8];
#pragma unroll
        for (int z = 0; z < 8; ++z) t[z] = z < zs ? *reinterpret_cast<const float4*>(dkv32 + (z * n + i) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 1; w < 8; w *= 2)
#pragma unroll
            for (int z = 0; z < 8; z += 2 * w) { t[z].x += t[z + w].x; t[z].y += t[z + w].y; t[z].z += t[z + w].z; t[z].w += t[z + w].w; }
        const float4 v = t[0];
        bf16_t* dst = (q < 16 ? dK + b * sdkv + (long long)key * lddk + q * 4 : dV + b * sdkv + (long long)key * lddv + (q - 16) * 4);
        st4<H>(reinterpret_cast<H*>(dst), v);
    }
}

// delta[row] = sum_d O[row][d] dO[row][d]: eight lanes x 16 bytes per 128-byte row (the first version read 2 bytes per lane)
template <typename H>
__global__ __launch_bounds__(256) void delta_rows_kernel(const bf16_t* __restrict__ O, int ldo, const bf16_t* __restrict__ dO, int lddo,
                                                         float* __restrict__ delta, long long rows, int wide) {
    if (wide) {
        const long long row = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
        const int c8 = (threadIdx.x & 7) * 8;
        float s = 0.f;
        if (row < rows) {
            const uint4 a = *reinterpret_cast<const uint4*>(O + row * ldo + c8), g = *reinterpret_cast<const uint4*>(dO + row * lddo + c8);
            const unsigned aw[4] = {a.x, a.y, a.z, a.w}, gw[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float a0, a1, g0, g1;
                unpack2<H>(aw[i], a0, a1);
                unpack2<H>(gw[i], g0, g1);
                s = fmaf(a0, g0, fmaf(a1, g1, s));
            }
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        if (row < rows && (threadIdx.x & 7) == 0) delta[row] = s;
        return;
    }
    const int lane = threadIdx.x & 63;
    for (int w = 0; w < 8; ++w) {
        const long long row = (long long)blockIdx.x * 32 + (threadIdx.x >> 6) * 8 + w;
        if (row >= rows) return;
        const float s = wave_sum(ldf<H>(reinterpret_cast<const H*>(O + row * ldo + lane)) * ldf<H>(reinterpret_cast<const H*>(dO + row * lddo + lane)));
        if (lane == 0) delta[row] = s;
    }
}

bool make_segs(Segs& sg, int B, int nseg, const int* nq, long long& total_rows) {
    if (nseg < 1 || nseg > 4) return false;
    sg.n = nseg;
    int row = 0;
    sg.t32[0] = 0;
    for (int i = 0; i < 4; ++i) {
        const int n = i < nseg ? nq[i] : 0;
        if (i < nseg && n <= 0) return false;
        sg.nq[i] = n;
        sg.row0[i] = row;
        row += B * n;
        sg.t32[i + 1] = sg.t32[i] + (n + 31) / 32;
    }
    total_rows = row;
    return true;
}

}  // namespace

extern "C" int tc_attn_fwd_seg(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, long long skv, void* O, int ldo,
                               float* lse, int B, int nseg, const int* nq, int Nk, float scale, int qscaled, int dtype, void* stream) {
    Segs sg;
    long long rows;
    if (!Q || !K || !V || !O || !lse || !nq || B <= 0 || Nk <= 0 || !make_segs(sg, B, nseg, nq, rows)) return TC_ERR_ARG;
    if (dtype == TC_F32) {                                      // parity path: one fp32 launch per segment
        if (qscaled) return TC_ERR_ARG;
        for (int i = 0; i < nseg; ++i) {
            const long long off = sg.row0[i];
            const int rc = tc_attn_fwd((const float*)Q + off * ldq, ldq, (long long)nq[i] * ldq, K, ldk, V, ldv, skv, (float*)O + off * ldo, ldo,
                                       (long long)nq[i] * ldo, lse + off, B, nq[i], Nk, scale, dtype, stream);
            if (rc != TC_OK) return rc;
        }
        return TC_OK;
    }
    if ((dtype != TC_BF16 && dtype != TC_F16) || ((ldq | ldk | ldv) & 7) || (skv & 7) || (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V) & 15) || (ldo & 3))
        return TC_ERR_ARG;
    const dim3 grid((unsigned)B * ((sg.t32[nseg] + FW_NW - 1) / FW_NW));
    const int wide_o = !(ldo & 7) && !((uintptr_t)O & 15);
    const float qs = qscaled ? 1.0f : scale * LOG2E;
    const char* asm_env = getenv("TC_ATTN_FWD_ASM");               // read per call: the tests run both kernels in one process
    const bool use_asm = !(asm_env && atoi(asm_env) == 0);
    // the stream takes Q already scaled (an in-kernel scale would round Q a second time: 2^-9 relative on every score)
    if (use_asm && qscaled && wide_o && Nk >= 64 && (long long)Nk * (ldk > ldv ? ldk : ldv) * 2 < (1ll << 31)) {   // the hand-scheduled stream (needs >= 2 key sub-tiles)
        static bool as_ok[2] = {false, false};
#define TC_FWD_ASM(HH, IDX)                                                                                                                  \
    {                                                                                                                                       \
        if (!as_ok[IDX]) {                                                                                                                  \
            if (hipFuncSetAttribute((const void*)attn_fwd_asm_kernel<HH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AS_SMEM) != hipSuccess) \
                return TC_ERR_LAUNCH;                                                                                                       \
            as_ok[IDX] = true;                                                                                                              \
        }                                                                                                                                   \
        hipLaunchKernelGGL(attn_fwd_asm_kernel<HH>, dim3((unsigned)B * ((sg.t32[nseg] + AS_NW - 1) / AS_NW)), dim3(AS_NW * 64), AS_SMEM,    \
                           (hipStream_t)stream, (const bf16_t*)Q, ldq, (const bf16_t*)K, ldk, (const bf16_t*)V, ldv, skv, (bf16_t*)O, ldo,  \
                           lse, sg, Nk, qs);                                                                                                \
    }
        if (dtype == TC_BF16) TC_FWD_ASM(bf16_t, 0) else TC_FWD_ASM(f16_t, 1)
#undef TC_FWD_ASM
        return tc_launch_status();
    }
    static bool lds_ok[2] = {false, false};                     // the kernels use 127 KB of dynamic LDS: raise the per-function limit once
#define TC_FWD(HH, IDX)                                                                                                                      \
    {                                                                                                                                       \
        if (!lds_ok[IDX]) {                                                                                                                 \
            if (hipFuncSetAttribute((const void*)attn_fwd_seg_kernel<HH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FW_SMEM) != hipSuccess) \
                return TC_ERR_LAUNCH;                                                                                                       \
            lds_ok[IDX] = true;                                                                                                             \
        }                                                                                                                                   \
        hipLaunchKernelGGL(attn_fwd_seg_kernel<HH>, grid, dim3(FW_NT), FW_SMEM, (hipStream_t)stream, (const bf16_t*)Q, ldq, (const bf16_t*)K, \
                           ldk, (const bf16_t*)V, ldv, skv, (bf16_t*)O, ldo, lse, sg, Nk, qs, wide_o);                                      \
    }
    if (dtype == TC_BF16) TC_FWD(bf16_t, 0) else TC_FWD(f16_t, 1)
#undef TC_FWD
    return tc_launch_status();
}

#ifdef TC_ATTN_ASM_TIMING
extern "C" int tc_attn_dbg_read(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_attn_dbg), sizeof(unsigned long long) * 256 * 12 * 8); }
#endif
#ifdef TC_DKV_TIMING
extern "C" int tc_dkv_dbg_read(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_dkv_dbg), sizeof(unsigned long long) * 512 * 4); }
#endif
extern "C" int tc_attn_bwd_seg(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, long long skv, const void* O, int ldo,
                               const void* dO, int lddo, const float* lse, float* delta, float* dkv32, void* dQ, int lddq, void* dK, int lddk,
                               void* dV, int lddv, long long sdkv, int B, int nseg, const int* nq, int Nk, float scale, int qscaled,
                               int dtype, void* stream) {
    Segs sg;
    long long rows;
    if (!Q || !K || !V || !O || !dO || !lse || !delta || !dQ || !dK || !dV || !nq || B <= 0 || Nk <= 0 || !make_segs(sg, B, nseg, nq, rows))
        return TC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == TC_F32) {
        if (qscaled) return TC_ERR_ARG;
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
    if ((dtype != TC_BF16 && dtype != TC_F16) || !dkv32 || ((ldq | ldk | ldv | lddo) & 7) || (skv & 7) ||
        (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)dO) & 15) || ((ldo | lddq | lddk | lddv) & 3) || (sdkv & 3))
        return TC_ERR_ARG;
    // workgroups of 4 key waves (measured: 4 waves x 2 workgroups per CU beats 5 x 1); query chunks so that ~2 workgroups per CU exist
    const int kt = (Nk + 31) / 32, nw = 4;
    const int kb = (kt + nw - 1) / nw, ntiles = sg.t32[nseg];
    static const int dkv_slots = getenv("TC_DKV_SLOTS") ? atoi(getenv("TC_DKV_SLOTS")) : 512;   // resident workgroups aimed for
    int zs = dkv_slots / (kb * B);
    zs = zs < 1 ? 1 : (zs > (ntiles + 1) / 2 ? (ntiles + 1) / 2 : zs);
    int tpc = (ntiles + zs - 1) / zs;
    constexpr int dkv_nt = DKV_QS / 32;
    tpc = (tpc + dkv_nt - 1) / dkv_nt * dkv_nt;                      // whole stages
    zs = (ntiles + tpc - 1) / tpc;
    if (zs > TC_ATTN_DKV_SPLITS) {                                    // dkv32 holds TC_ATTN_DKV_SPLITS partial buffers
        tpc = ((ntiles + TC_ATTN_DKV_SPLITS - 1) / TC_ATTN_DKV_SPLITS + dkv_nt - 1) / dkv_nt * dkv_nt;
        zs = (ntiles + tpc - 1) / tpc;
    }
    const float qs = qscaled ? 1.0f : scale * LOG2E, kscale = qscaled ? LN2 : scale;    // dK = dS^T Q: ln 2 when Q is stored as q * scale * log2(e)
    const int wide_dq = !(lddq & 7) && !((uintptr_t)dQ & 15);
    const int wide_rows = !((ldo | lddo) & 7) && !(((uintptr_t)O | (uintptr_t)dO) & 15);
    static const int fuse_env = getenv("TC_ATTN_FUSE_DELTA") ? atoi(getenv("TC_ATTN_FUSE_DELTA")) : 1;
    const bool fuse_delta = wide_rows && fuse_env;
    // the hand-scheduled dK/dV stream: Q stored scaled, 16-byte rows, every query chunk at least two tiles long, and room for the per-tile
    // statistics (B x tiles x 64 floats) in the LAST of the TC_ATTN_DKV_SPLITS partial buffers
    const int dkv_asm_env = getenv("TC_ATTN_DKV_ASM") ? atoi(getenv("TC_ATTN_DKV_ASM")) : 1;
    const bool dkv_asm = dkv_asm_env && qscaled && wide_rows && zs < TC_ATTN_DKV_SPLITS && (long long)ntiles * 64 <= (long long)Nk * 128 &&
                         ntiles - (zs - 1) * tpc >= 2 && tpc >= 2 && rows * (long long)(ldq > lddo ? ldq : lddo) * 2 < 0x7fffffffLL;
    float* const nstat = dkv_asm ? dkv32 + (long long)(TC_ATTN_DKV_SPLITS - 1) * B * Nk * 128 : nullptr;
    // paired query chunks (8-wave workgroups): an even number of equally long chunks, the last one overlapping its predecessor by fewer tiles
    // than the kernel stages itself
    const int pair_env = getenv("TC_ATTN_DKV_PAIR") ? atoi(getenv("TC_ATTN_DKV_PAIR")) : 1;     // read per call: the tests run both forms in one process
    const bool dkv_pair = pair_env && dkv_asm && zs >= 2 && !(zs & 1) && tpc <= ntiles && zs * tpc - ntiles < DA_AHEAD - 1 && (zs - 1) * tpc < ntiles;
    const int dq_asm_env = getenv("TC_ATTN_DQ_ASM") ? atoi(getenv("TC_ATTN_DQ_ASM")) : 1;
    const bool dq_asm = dq_asm_env && qscaled && fuse_delta && wide_dq && Nk >= 33 && rows * (long long)(ldq > lddo ? ldq : lddo) * 2 < 0x7fffffffLL;
    static bool lds_ok[2] = {false, false};
    // dQ first: it computes delta = rowsum(O dO) on the way (when O / dO rows are 16-byte accessible) and the dK/dV kernel reads it
#define TC_BWD_DQ(HH, QMV)                                                                                                                 \
        hipLaunchKernelGGL((attn_bwd_dq_seg_kernel<HH, QMV>), dim3((unsigned)B * ((sg.t32[nseg] + FW_NW - 1) / FW_NW)), dim3(FW_NT), FW_SMEM, s, \
                           (const bf16_t*)Q, ldq, (const bf16_t*)K, ldk, (const bf16_t*)V, ldv, skv, fuse_delta ? (const bf16_t*)O : nullptr, ldo, \
                           (const bf16_t*)dO, lddo, lse, delta, (bf16_t*)dQ, lddq, sg, Nk, scale, qs, wide_dq, nstat)
#define TC_BWD(HH, IDX)                                                                                                                     \
    {                                                                                                                                       \
        if (!lds_ok[IDX]) {                                                                                                                 \
            if (hipFuncSetAttribute((const void*)attn_bwd_dq_seg_kernel<HH, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FW_SMEM) != hipSuccess || \
                hipFuncSetAttribute((const void*)attn_bwd_dq_seg_kernel<HH, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FW_SMEM) != hipSuccess || \
                hipFuncSetAttribute((const void*)attn_bwd_dkv_asm_kernel<HH, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, DA_SMEM) != hipSuccess || \
                hipFuncSetAttribute((const void*)attn_bwd_dkv_asm_kernel<HH, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * DA_HALF) != hipSuccess || \
                hipFuncSetAttribute((const void*)attn_bwd_dq_asm_kernel<HH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AS_SMEM) != hipSuccess || \
                hipFuncSetAttribute((const void*)attn_bwd_dkv_seg_kernel<HH, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_SMEM_B) != hipSuccess || \
                hipFuncSetAttribute((const void*)attn_bwd_dkv_seg_kernel<HH, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, DKV_SMEM_B) != hipSuccess) \
                return TC_ERR_LAUNCH;                                                                                                       \
            lds_ok[IDX] = true;                                                                                                             \
        }                                                                                                                                   \
        if (!fuse_delta)                                                                                                                    \
            hipLaunchKernelGGL(delta_rows_kernel<HH>, dim3((unsigned)((rows + 31) / 32)), dim3(256), 0, s, (const bf16_t*)O, ldo,             \
                               (const bf16_t*)dO, lddo, delta, rows, wide_rows);                                                            \
        if (dq_asm)                                                                                                                         \
            hipLaunchKernelGGL((attn_bwd_dq_asm_kernel<HH>), dim3((unsigned)B * ((sg.t32[nseg] + AS_NW - 1) / AS_NW)), dim3(AS_NW * 64), AS_SMEM, s, \
                               (const bf16_t*)Q, ldq, (const bf16_t*)K, ldk, (const bf16_t*)V, ldv, skv, (const bf16_t*)O, ldo, (const bf16_t*)dO, lddo, \
                               lse, delta, (bf16_t*)dQ, lddq, sg, Nk, scale, nstat, rows);                                                     \
        else if (qscaled) TC_BWD_DQ(HH, 1); else TC_BWD_DQ(HH, 0);                                                                          \
        if (dkv_pair)                                                                                                                       \
            hipLaunchKernelGGL((attn_bwd_dkv_asm_kernel<HH, 2>), dim3(kb, B, zs / 2), dim3(512), 2 * DA_HALF, s, (const bf16_t*)Q, ldq,           \
                               (const bf16_t*)K, ldk, (const bf16_t*)V, ldv, skv, (const bf16_t*)dO, lddo, nstat, dkv32, sg, Nk, kscale, tpc, rows); \
        else if (dkv_asm)                                                                                                                   \
            hipLaunchKernelGGL((attn_bwd_dkv_asm_kernel<HH, 1>), dim3(kb, B, zs), dim3(256), DA_SMEM, s, (const bf16_t*)Q, ldq, (const bf16_t*)K, ldk, \
                               (const bf16_t*)V, ldv, skv, (const bf16_t*)dO, lddo, nstat, dkv32, sg, Nk, kscale, tpc, rows);                    \
        else if (qscaled)                                                                                                                   \
            hipLaunchKernelGGL((attn_bwd_dkv_seg_kernel<HH, 4, true>), dim3(kb, B, zs), dim3(256), DKV_SMEM_B, s, (const bf16_t*)Q, ldq, (const bf16_t*)K, \
                               ldk, (const bf16_t*)V, ldv, skv, (const bf16_t*)dO, lddo, lse, delta, dkv32, sg, Nk, kscale, qs, tpc);             \
        else                                                                                                                                \
            hipLaunchKernelGGL((attn_bwd_dkv_seg_kernel<HH, 4, false>), dim3(kb, B, zs), dim3(256), DKV_SMEM_B, s, (const bf16_t*)Q, ldq, (const bf16_t*)K, \
                               ldk, (const bf16_t*)V, ldv, skv, (const bf16_t*)dO, lddo, lse, delta, dkv32, sg, Nk, kscale, qs, tpc);             \
        hipLaunchKernelGGL(attn_dkv_store_kernel<HH>, dim3(tc_blocks((long long)B * Nk * 32, 256, 1024)), dim3(256), 0, s, dkv32,           \
                           (bf16_t*)dK, lddk, (bf16_t*)dV, lddv, sdkv, B, Nk, dkv_pair ? zs / 2 : zs);                                          \
    }
    if (dtype == TC_BF16) TC_BWD(bf16_t, 0) else TC_BWD(f16_t, 1)
#undef TC_BWD
#undef TC_BWD_DQ
    return tc_launch_status();
}
