// Depthwise k x k convolutions (k = 3/5/7, stride 1/2) on NHWC maps: forward, input gradient, weight/bias gradient.
// HBM-bound: every thread owns 4 consecutive channels of a pixel (16 lanes = 256 B contiguous per pixel), the
// filter slice of the block's 64 channels is staged once in LDS in [tap][channel] order (the PyTorch [C,1,k,k]
// layout is transposed on the way in), and the k*k window re-reads are served by L1/L2.
#include "tc_common.h"
#include <cstdlib>
#ifndef DW_FOLD
#define DW_FOLD 4      // workgroups that share an arrival counter in the two-level folds of the weight-gradient kernels (4 measured best of 1-16)
#endif

namespace {
// workgroups a weight-gradient launch aims for (A/B switches; defaults = one tile walker per CU)
inline int tc_dw_wg_target() { static const int v = getenv("TC_DW_WG") ? atoi(getenv("TC_DW_WG")) : 512; return v; }   // 256 while every launch folded at its tail; with the deferred fold: 12.07 / 12.03 / 12.01 / 12.03 ms at 256 / 384 / 512 / 768
inline int tc_mid_wg_target() { static const int v = getenv("TC_MID_WG") ? atoi(getenv("TC_MID_WG")) : 384; return v; }   // (re-swept with the other two: 256 / 384 / 512 -> 12.07 / 12.03 / 12.06 ms)


// STRIDE is a template parameter (the strided form serves the first depthwise convolution of every RIPM stage, stride 2): with a run-time
// stride the input-gradient form spends ~40 instructions per tap on `t % stride` / `t / stride`, and the pixel index is split with
// 32-bit arithmetic (a 64-bit division is ~100 instructions on this machine) -- 15 -> see DESIGN.md section 5 us per launch at [16, 56, 56, 64].
template <typename T, int K, bool BWD, int STRIDE>
__global__ __launch_bounds__(256) void dw_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ w,
                                                 const T* __restrict__ bias, T* __restrict__ y, int ldy, int B, int H, int W,
                                                 int Ho, int Wo, int C, int add_input, int accumulate) {
    constexpr int stride = STRIDE;
    // forward:  x is the [B,H,W] input, y the [B,Ho,Wo] output.
    // BWD    :  x is dy on [B,Ho,Wo], y is dx on [B,H,W]  (transposed convolution with the same taps).
    __shared__ float wsm[K * K][64];
    __shared__ float bsm[64];
    const int c0 = blockIdx.y * 64;
    for (int i = threadIdx.x; i < K * K * 64; i += 256) {
        const int tap = i >> 6, cc = i & 63;
        wsm[tap][cc] = (c0 + cc < C) ? ldf<T>(w + (long long)(c0 + cc) * K * K + tap) : 0.f;
    }
    if (threadIdx.x < 64) bsm[threadIdx.x] = (bias && c0 + threadIdx.x < C) ? ldf<T>(bias + c0 + threadIdx.x) : 0.f;
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = c0 + tx * 4;
    if (c >= C) return;
    constexpr int P = (K - 1) / 2;
    const int OH = BWD ? H : Ho, OW = BWD ? W : Wo;          // extent of the tensor being written
    const int IH = BWD ? Ho : H, IW = BWD ? Wo : W;          // extent of the tensor being read
    const unsigned npix = (unsigned)B * OH * OW;                // (< 2^31: checked by the host)
    for (unsigned pix = blockIdx.x * 16 + ty; pix < npix; pix += gridDim.x * 16) {
        const int ow = (int)(pix % (unsigned)OW);
        const unsigned prow = pix / (unsigned)OW;
        const int oh = (int)(prow % (unsigned)OH);
        const int b = (int)(prow / (unsigned)OH);
        float4 acc = BWD ? make_float4(0.f, 0.f, 0.f, 0.f)
                         : make_float4(bsm[tx * 4], bsm[tx * 4 + 1], bsm[tx * 4 + 2], bsm[tx * 4 + 3]);
        const T* xb = x + (long long)b * IH * IW * ldx + c;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            int ih;
            if (!BWD) ih = oh * stride + ky - P;
            else { const int t = oh + P - ky; if (t % stride) continue; ih = t / stride; if (t < 0) continue; }
            if (ih < 0 || ih >= IH) continue;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                int iw;
                if (!BWD) iw = ow * stride + kx - P;
                else { const int t = ow + P - kx; if (t % stride) continue; iw = t / stride; if (t < 0) continue; }
                if (iw < 0 || iw >= IW) continue;
                const float4 v = ld4<T>(xb + ((long long)ih * IW + iw) * ldx);
                const float* wt = &wsm[ky * K + kx][tx * 4];
                acc.x += v.x * wt[0]; acc.y += v.y * wt[1]; acc.z += v.z * wt[2]; acc.w += v.w * wt[3];
            }
        }
        if (add_input) {   // stride 1: same pixel of the read tensor
            const float4 v = ld4<T>(xb + ((long long)oh * IW + ow) * ldx);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        if (accumulate) { const float4 o = ld4<T>(y + (long long)pix * ldy + c); acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
        st4<T>(y + (long long)pix * ldy + c, acc);
    }
}

// dw[c, tap] += sum_pix dy[pix, c] * x[pix shifted by tap, c] ; db[c] += sum_pix dy[pix, c]
template <typename T, int K>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ x, int ldx,
                                                       float* __restrict__ dw, float* __restrict__ db, int B, int H, int W,
                                                       int Ho, int Wo, int C, int stride) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + tx;
    constexpr int P = (K - 1) / 2;
    float acc[K * K];
    float accb = 0.f;
#pragma unroll
    for (int i = 0; i < K * K; ++i) acc[i] = 0.f;
    const long long npix = (long long)B * Ho * Wo;
    if (c < C) {
        for (long long pix = (long long)blockIdx.x * 4 + ty; pix < npix; pix += (long long)gridDim.x * 4) {
            const int ow = (int)(pix % Wo);
            const int oh = (int)((pix / Wo) % Ho);
            const int b = (int)(pix / ((long long)Wo * Ho));
            const float d = ldf<T>(dy + pix * lddy + c);
            accb += d;
            const T* xb = x + (long long)b * H * W * ldx + c;
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const int ih = oh * stride + ky - P;
                if (ih < 0 || ih >= H) continue;
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const int iw = ow * stride + kx - P;
                    if (iw < 0 || iw >= W) continue;
                    acc[ky * K + kx] += d * ldf<T>(xb + ((long long)ih * W + iw) * ldx);
                }
            }
        }
    }
    constexpr int NT = K * K + 1;
    __shared__ float redall[4][NT][64];
#pragma unroll
    for (int i = 0; i < NT; ++i) redall[ty][i][tx] = (i < K * K) ? acc[i < K * K ? i : 0] : accb;
    __syncthreads();
    for (int f = threadIdx.x; f < NT * 64; f += 256) {
        const int cc = f / NT, t = f - cc * NT, ch = blockIdx.y * 64 + cc;      // taps fastest: coalesced atomics
        if (ch >= C) continue;
        const float s = redall[0][t][cc] + redall[1][t][cc] + redall[2][t][cc] + redall[3][t][cc];
        if (t < K * K) atomicAdd(dw + (long long)ch * K * K + t, s);
        else if (db) atomicAdd(db + ch, s);
    }
}

// The same for 3 x 3 filters with 8- / 16-byte loads: a thread owns 4 consecutive channels and one of 16 pixel lanes (the scalar form
// above moves 2 bytes per lane and load and took 65 us on the 6 MB map of the first RIPM convolution); the 16 lanes of a channel
// quad fold by two shuffles and one pass through LDS, the grid is capped at 64 workgroups per channel chunk so that at most 64
// atomics queue on a word.
template <typename T>
__global__ __launch_bounds__(256) void dw_wgrad3_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ x, int ldx,
                                                        float* __restrict__ dw, float* __restrict__ db, int B, int H, int W,
                                                        int Ho, int Wo, int C, int stride) {
    constexpr int K = 3, NT = K * K + 1;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = blockIdx.y * 64 + tx * 4;
    float4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned npix = (unsigned)B * Ho * Wo;                // (< 2^31: checked by the host; 32-bit index arithmetic, see dw_kernel)
    if (c < C) {
        for (unsigned pix = blockIdx.x * 16 + ty; pix < npix; pix += gridDim.x * 16) {
            const unsigned prow = pix / (unsigned)Wo;
            const int ow = (int)(pix - prow * (unsigned)Wo), oh = (int)(prow % (unsigned)Ho), b = (int)(prow / (unsigned)Ho);
            const float4 d = ld4<T>(dy + (long long)pix * lddy + c);
            const T* xb = x + (long long)b * H * W * ldx + c;
            float4 v[K * K];
#pragma unroll
            for (int t = 0; t < K * K; ++t) {                      // all nine loads issued before the first use
                const int ih = oh * stride + t / K - 1, iw = ow * stride + t % K - 1;
                const bool in = ih >= 0 && ih < H && iw >= 0 && iw < W;
                v[t] = ld4<T>(xb + (in ? ((long long)ih * W + iw) * ldx : 0));
                if (!in) v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int t = 0; t < K * K; ++t) { acc[t].x += d.x * v[t].x; acc[t].y += d.y * v[t].y; acc[t].z += d.z * v[t].z; acc[t].w += d.w * v[t].w; }
            acc[K * K].x += d.x; acc[K * K].y += d.y; acc[K * K].z += d.z; acc[K * K].w += d.w;
        }
    }
    __shared__ float red[4][NT][64];
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float4 a = acc[t];
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {                         // lanes of one channel quad inside a wave are 16 apart
            a.x += __shfl_xor(a.x, o, 64); a.y += __shfl_xor(a.y, o, 64); a.z += __shfl_xor(a.z, o, 64); a.w += __shfl_xor(a.w, o, 64);
        }
        if ((threadIdx.x & 63) < 16) { float* r = &red[wave][t][tx * 4]; r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; }
    }
    __syncthreads();
    for (int f = threadIdx.x; f < NT * 64; f += 256) {
        const int cc = f / NT, t = f - cc * NT, ch = blockIdx.y * 64 + cc;      // taps fastest: coalesced atomics
        if (ch >= C) continue;
        const float s_ = red[0][t][cc] + red[1][t][cc] + red[2][t][cc] + red[3][t][cc];
        if (t < K * K) atomicAdd(dw + (long long)ch * K * K + t, s_);
        else if (db) atomicAdd(db + ch, s_);
    }
}

// ------------------------------------------------------------------------------------------------------------
// Stride-1 "row strip" kernels: a thread owns CPT consecutive channels of one output row (b, oh) and walks along ow with
// the K x K input window held in registers (K new loads per output instead of K*K; column slots rotate at compile time).
//   MODE 0: y = conv(x) (+bias) (+x)          MODE 1: dx = conv^T(dy) (+dy)  [same walk with flipped taps]
//   MODE 2: dw[c,ky,kx] += sum dy * x ;  db[c] += sum dy   (block-level LDS reduction, then one atomic per block and tap)
template <typename T, int CPT> __device__ __forceinline__ void ldv(const T* p, float* o);
template <> __device__ __forceinline__ void ldv<float, 4>(const float* p, float* o) { const float4 v = *reinterpret_cast<const float4*>(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
template <> __device__ __forceinline__ void ldv<float, 2>(const float* p, float* o) { const float2 v = *reinterpret_cast<const float2*>(p); o[0] = v.x; o[1] = v.y; }
template <> __device__ __forceinline__ void ldv<bf16_t, 4>(const bf16_t* p, float* o) { const float4 v = ld4<bf16_t>(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
template <> __device__ __forceinline__ void ldv<bf16_t, 2>(const bf16_t* p, float* o) { const unsigned r = *reinterpret_cast<const unsigned*>(p); o[0] = __uint_as_float(r << 16); o[1] = __uint_as_float(r & 0xffff0000u); }
template <> __device__ __forceinline__ void ldv<f16_t, 4>(const f16_t* p, float* o) { const float4 v = ld4<f16_t>(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
template <> __device__ __forceinline__ void ldv<f16_t, 2>(const f16_t* p, float* o) { unpack2<f16_t>(*reinterpret_cast<const unsigned*>(p), o[0], o[1]); }
template <typename T, int CPT> __device__ __forceinline__ void stv(T* p, const float* o);
template <> __device__ __forceinline__ void stv<float, 4>(float* p, const float* o) { *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]); }
template <> __device__ __forceinline__ void stv<float, 2>(float* p, const float* o) { *reinterpret_cast<float2*>(p) = make_float2(o[0], o[1]); }
template <> __device__ __forceinline__ void stv<bf16_t, 4>(bf16_t* p, const float* o) { st4<bf16_t>(p, make_float4(o[0], o[1], o[2], o[3])); }
template <> __device__ __forceinline__ void stv<bf16_t, 2>(bf16_t* p, const float* o) { *reinterpret_cast<unsigned*>(p) = pack2bf(o[0], o[1]); }

template <> __device__ __forceinline__ void stv<f16_t, 4>(f16_t* p, const float* o) { st4<f16_t>(p, make_float4(o[0], o[1], o[2], o[3])); }
template <> __device__ __forceinline__ void stv<f16_t, 2>(f16_t* p, const float* o) { *reinterpret_cast<unsigned*>(p) = pack2h(o[0], o[1]); }

template <typename T, int K, int CPT, int MODE>
__global__ __launch_bounds__(256) void dw_strip_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ w, const T* __restrict__ bias,
                                                       const T* __restrict__ dy, int lddy, T* __restrict__ y, int ldy,
                                                       float* __restrict__ dw, float* __restrict__ db, int B, int H, int W, int C,
                                                       int add_input, int accumulate, int nseg, long long wstride) {
    constexpr int P = (K - 1) / 2, CGB = 64 / CPT, RI = 256 / CGB;
    {   // group = blockIdx.z: B images each; every parameter / parameter-gradient pointer of group g is wstride elements further on
        const long long g = blockIdx.z;
        const long long img = (long long)B * H * W;
        if (x) x += g * img * ldx;
        if (dy) dy += g * img * lddy;
        if (y) y += g * img * ldy;
        if (w) w += g * wstride;
        if (bias) bias += g * wstride;
        if (dw) dw += g * wstride;
        if (db) db += g * wstride;
    }
    const int cgl = threadIdx.x % CGB, ri = threadIdx.x / CGB;
    const int c = blockIdx.y * 64 + cgl * CPT;
    const bool cok = c < C;
    const T* src = (MODE == 1) ? dy : x;              // tensor the window slides over
    const int lds_ = (MODE == 1) ? lddy : ldx;
    float wr[(MODE == 2) ? 1 : K * K][CPT];
    float bs[CPT];
    float acc[(MODE == 2) ? K * K : 1][CPT];
    float accb[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) { bs[i] = 0.f; accb[i] = 0.f; }
    if (MODE != 2 && cok) {
#pragma unroll
        for (int t = 0; t < K * K; ++t)
#pragma unroll
            for (int i = 0; i < CPT; ++i)
                wr[t][i] = ldf<T>(w + (long long)(c + i) * K * K + (MODE == 1 ? (K * K - 1 - t) : t));
        if (MODE == 0 && bias)
#pragma unroll
            for (int i = 0; i < CPT; ++i) bs[i] = ldf<T>(bias + c + i);
    }
    if (MODE == 2) {
#pragma unroll
        for (int t = 0; t < K * K; ++t)
#pragma unroll
            for (int i = 0; i < CPT; ++i) acc[t][i] = 0.f;
    }
    const int nrows = B * H * nseg;                 // work item = (b, oh, W-segment)
    const int seglen = (W + nseg - 1) / nseg;
    if (cok) {
        for (int item = blockIdx.x * RI + ri; item < nrows; item += gridDim.x * RI) {
            const int seg = item % nseg, row = item / nseg;
            const int oh = row % H, b = row / H;
            const int w0 = seg * seglen, w1 = min(W, w0 + seglen);
            const T* sb = src + (long long)b * H * W * lds_ + c;
            float win[K][K][CPT];                      // win[ky][slot]: slot (j + kx) % K holds column ow - P + kx at step j
            bool rok[K];
#pragma unroll
            for (int ky = 0; ky < K; ++ky) rok[ky] = (oh + ky - P >= 0) && (oh + ky - P < H);
            // prime: columns w0-P .. w0+P-1 go to slots 0 .. K-2
#pragma unroll
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int kx = 0; kx < K - 1; ++kx) {
                    const int iw = w0 + kx - P;
#pragma unroll
                    for (int i = 0; i < CPT; ++i) win[ky][kx][i] = 0.f;
                    if (rok[ky] && iw >= 0 && iw < W) ldv<T, CPT>(sb + ((long long)(oh + ky - P) * W + iw) * lds_, win[ky][kx]);
                }
            for (int ow0 = w0; ow0 < w1; ow0 += K) {
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const int ow = ow0 + j;
                    if (ow < w1) {
                        // newest column ow + P enters slot (j + K - 1) % K
                        const int iw = ow + P;
#pragma unroll
                        for (int ky = 0; ky < K; ++ky) {
#pragma unroll
                            for (int i = 0; i < CPT; ++i) win[ky][(j + K - 1) % K][i] = 0.f;
                            if (rok[ky] && iw < W) ldv<T, CPT>(sb + ((long long)(oh + ky - P) * W + iw) * lds_, win[ky][(j + K - 1) % K]);
                        }
                        const long long pix = ((long long)b * H + oh) * W + ow;
                        if (MODE == 2) {
                            float d[CPT];
                            ldv<T, CPT>(dy + pix * lddy + c, d);
#pragma unroll
                            for (int i = 0; i < CPT; ++i) accb[i] += d[i];
#pragma unroll
                            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                                for (int kx = 0; kx < K; ++kx)
#pragma unroll
                                    for (int i = 0; i < CPT; ++i) acc[ky * K + kx][i] += d[i] * win[ky][(j + kx) % K][i];
                        } else {
                            float o[CPT];
#pragma unroll
                            for (int i = 0; i < CPT; ++i) o[i] = bs[i];
#pragma unroll
                            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                                for (int kx = 0; kx < K; ++kx)
#pragma unroll
                                    for (int i = 0; i < CPT; ++i) o[i] += wr[ky * K + kx][i] * win[ky][(j + kx) % K][i];
                            if (add_input)
#pragma unroll
                                for (int i = 0; i < CPT; ++i) o[i] += win[P][(j + P) % K][i];
                            T* dst = y + pix * ldy + c;
                            if (accumulate) { float q[CPT]; ldv<T, CPT>(dst, q);
#pragma unroll
                                for (int i = 0; i < CPT; ++i) o[i] += q[i]; }
                            stv<T, CPT>(dst, o);
                        }
                    }
                }
            }
        }
    }
    if (MODE == 2) {
        // reduce over the row items of the block: lanes that hold the same channels are CGB apart inside a wave (xor
        // shuffles), the four waves meet in LDS for all taps at once (two barriers instead of two per tap), then one
        // atomic per channel and tap per block.
        constexpr int NT = K * K + 1;
        __shared__ float red[4][NT][64];
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                float v = (t < K * K) ? acc[t < K * K ? t : 0][i] : accb[i];
#pragma unroll
                for (int o = CGB; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
                if (lane < CGB) red[wave][t][cgl * CPT + i] = v;
            }
        }
        __syncthreads();
        for (int f = threadIdx.x; f < NT * 64; f += 256) {
            const int cc = f / NT, t = f - cc * NT, ch = blockIdx.y * 64 + cc;      // taps fastest: neighbouring lanes hit neighbouring dw words
            if (ch >= C) continue;
            const float s_ = red[0][t][cc] + red[1][t][cc] + red[2][t][cc] + red[3][t][cc];
            if (t < K * K) atomicAdd(dw + (long long)ch * K * K + t, s_);
            else if (db) atomicAdd(db + ch, s_);
        }
    }
}

template <typename T, int MODE>
int launch_strip(const void* x, int ldx, const void* w, const void* bias, const void* dy, int lddy, void* y, int ldy, float* dw,
                 float* db, int B, int H, int W, int C, int k, int add_input, int accumulate, int groups, long long wstride,
                 hipStream_t s) {
    const int nrows = B * H;
#define TC_STRIP(KK, CPT)                                                                                                       \
    {                                                                                                                           \
        constexpr int RI = 256 / (64 / CPT);                                                                                    \
        /* split rows into W-segments until ~256k threads exist (few long serial walks leave most of the 256 CUs idle) */      \
        int nseg = (int)(262144LL / ((long long)nrows * groups * ((C + CPT - 1) / CPT)));                                                 \
        const int maxseg = W / (KK + 1) > 0 ? W / (KK + 1) : 1;                                                                 \
        nseg = nseg < 1 ? 1 : (nseg > maxseg ? maxseg : nseg);                                                                  \
        dim3 grid(tc_blocks((long long)nrows * nseg, RI, MODE == 2 ? 512 : 8192), (C + 63) / 64, groups);                                \
        hipLaunchKernelGGL((dw_strip_kernel<T, KK, CPT, MODE>), grid, dim3(256), 0, s, (const T*)x, ldx, (const T*)w, (const T*)bias, \
                           (const T*)dy, lddy, (T*)y, ldy, dw, db, B, H, W, C, add_input, accumulate, nseg, wstride);                    \
    }
    if (k == 3) TC_STRIP(3, 4) else if (k == 5) TC_STRIP(5, 2) else TC_STRIP(7, 2)
#undef TC_STRIP
    return tc_launch_status();
}

// ------------------------------------------------------------------------------------------------------------
// Stride-1 LDS-tile kernels (the default): a workgroup owns TH x 16 output pixels x (CG * VEC) channels.  The input tile with
// its (K-1) halo is fetched with 16-byte loads that are ALL issued before the first use (the row-strip walk above keeps only
// K loads per thread in flight and is latency-bound at ~1/4 of the HBM rate), parked in LDS with a padded pixel pitch, and
// every thread then produces a run of 4 horizontally adjacent pixels for its VEC channels from LDS.
//   VEC = channels per 16 bytes (8 bf16 / 4 fp32); CG = 16-byte lanes per pixel (8/4/2: 64/32/16 bf16 channels per workgroup,
//   picked on the host to waste the fewest lanes on the 16/24/48/80/120-channel maps of the multi-branch stages).
template <typename T> struct Vec16;
template <> struct Vec16<float> { static constexpr int N = 4; };
template <> struct Vec16<bf16_t> { static constexpr int N = 8; };
template <> struct Vec16<f16_t> { static constexpr int N = 8; };
template <typename T> __device__ __forceinline__ void unpack16(const uint4& r, float* o);
template <> __device__ __forceinline__ void unpack16<float>(const uint4& r, float* o) {
    o[0] = __uint_as_float(r.x); o[1] = __uint_as_float(r.y); o[2] = __uint_as_float(r.z); o[3] = __uint_as_float(r.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const uint4& r, float* o) {
    o[0] = __uint_as_float(r.x << 16); o[1] = __uint_as_float(r.x & 0xffff0000u);
    o[2] = __uint_as_float(r.y << 16); o[3] = __uint_as_float(r.y & 0xffff0000u);
    o[4] = __uint_as_float(r.z << 16); o[5] = __uint_as_float(r.z & 0xffff0000u);
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
// the same vectors as pairs for the packed fp32 pipe (v_pk_fma_f32: two multiply-adds per issue slot)
template <typename T> __device__ __forceinline__ void unpack16v(const uint4& r, tc_f32x2* o);
template <> __device__ __forceinline__ void unpack16v<float>(const uint4& r, tc_f32x2* o) {
    o[0] = tc_f32x2{__uint_as_float(r.x), __uint_as_float(r.y)}; o[1] = tc_f32x2{__uint_as_float(r.z), __uint_as_float(r.w)};
}
template <> __device__ __forceinline__ void unpack16v<bf16_t>(const uint4& r, tc_f32x2* o) {
    o[0] = tc_f32x2{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u)};
    o[1] = tc_f32x2{__uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
    o[2] = tc_f32x2{__uint_as_float(r.z << 16), __uint_as_float(r.z & 0xffff0000u)};
    o[3] = tc_f32x2{__uint_as_float(r.w << 16), __uint_as_float(r.w & 0xffff0000u)};
}
template <> __device__ __forceinline__ void unpack16v<f16_t>(const uint4& r, tc_f32x2* o) {
    float a, b;
    unpack2<f16_t>(r.x, a, b); o[0] = tc_f32x2{a, b};
    unpack2<f16_t>(r.y, a, b); o[1] = tc_f32x2{a, b};
    unpack2<f16_t>(r.z, a, b); o[2] = tc_f32x2{a, b};
    unpack2<f16_t>(r.w, a, b); o[3] = tc_f32x2{a, b};
}
template <typename T> __device__ __forceinline__ uint4 pack16v(const tc_f32x2* o);
template <> __device__ __forceinline__ uint4 pack16v<float>(const tc_f32x2* o) {
    return make_uint4(__float_as_uint(o[0].x), __float_as_uint(o[0].y), __float_as_uint(o[1].x), __float_as_uint(o[1].y));
}
template <> __device__ __forceinline__ uint4 pack16v<bf16_t>(const tc_f32x2* o) {
    return make_uint4(pack2bf(o[0].x, o[0].y), pack2bf(o[1].x, o[1].y), pack2bf(o[2].x, o[2].y), pack2bf(o[3].x, o[3].y));
}

template <> __device__ __forceinline__ uint4 pack16v<f16_t>(const tc_f32x2* o) {
    return make_uint4(pack2h(o[0].x, o[0].y), pack2h(o[1].x, o[1].y), pack2h(o[2].x, o[2].y), pack2h(o[3].x, o[3].y));
}

// SV consecutive channels (a whole 16-byte vector or half of one) from LDS / to global memory, as packed pairs
template <typename T, int SV> __device__ __forceinline__ void unpack_sv(const void* p, tc_f32x2* o);
template <> __device__ __forceinline__ void unpack_sv<float, 4>(const void* p, tc_f32x2* o) { unpack16v<float>(*reinterpret_cast<const uint4*>(p), o); }
template <> __device__ __forceinline__ void unpack_sv<float, 2>(const void* p, tc_f32x2* o) { const float2 v = *reinterpret_cast<const float2*>(p); o[0] = tc_f32x2{v.x, v.y}; }
template <> __device__ __forceinline__ void unpack_sv<bf16_t, 8>(const void* p, tc_f32x2* o) { unpack16v<bf16_t>(*reinterpret_cast<const uint4*>(p), o); }
template <> __device__ __forceinline__ void unpack_sv<bf16_t, 4>(const void* p, tc_f32x2* o) {
    const uint2 r = *reinterpret_cast<const uint2*>(p);
    o[0] = tc_f32x2{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u)};
    o[1] = tc_f32x2{__uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
}
template <> __device__ __forceinline__ void unpack_sv<f16_t, 8>(const void* p, tc_f32x2* o) { unpack16v<f16_t>(*reinterpret_cast<const uint4*>(p), o); }
template <> __device__ __forceinline__ void unpack_sv<f16_t, 4>(const void* p, tc_f32x2* o) {
    const uint2 r = *reinterpret_cast<const uint2*>(p);
    float a, b;
    unpack2<f16_t>(r.x, a, b); o[0] = tc_f32x2{a, b};
    unpack2<f16_t>(r.y, a, b); o[1] = tc_f32x2{a, b};
}
template <typename T, int SV> __device__ __forceinline__ void store_sv(T* p, const tc_f32x2* o);
template <> __device__ __forceinline__ void store_sv<float, 4>(float* p, const tc_f32x2* o) { *reinterpret_cast<uint4*>(p) = pack16v<float>(o); }
template <> __device__ __forceinline__ void store_sv<float, 2>(float* p, const tc_f32x2* o) { *reinterpret_cast<float2*>(p) = make_float2(o[0].x, o[0].y); }
template <> __device__ __forceinline__ void store_sv<bf16_t, 8>(bf16_t* p, const tc_f32x2* o) { *reinterpret_cast<uint4*>(p) = pack16v<bf16_t>(o); }
template <> __device__ __forceinline__ void store_sv<bf16_t, 4>(bf16_t* p, const tc_f32x2* o) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack2bf(o[0].x, o[0].y), pack2bf(o[1].x, o[1].y));
}
template <> __device__ __forceinline__ void store_sv<f16_t, 8>(f16_t* p, const tc_f32x2* o) { *reinterpret_cast<uint4*>(p) = pack16v<f16_t>(o); }
template <> __device__ __forceinline__ void store_sv<f16_t, 4>(f16_t* p, const tc_f32x2* o) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack2h(o[0].x, o[0].y), pack2h(o[1].x, o[1].y));
}

template <int K, int CG> struct DwTile {
    static constexpr int PT = 256 / CG, R = 4, TW = 16, TH = PT / (TW / R), P = (K - 1) / 2;
    static constexpr int IW = TW + K - 1, IH = TH + K - 1;
};

// Tile fill, global -> registers -> LDS.  The loads are issued unconditionally (a position outside the map or past channel C reads
// the tile's first valid vector instead) and the zeroing happens when the registers go to LDS: a load inside a per-lane branch, or
// a select right behind it, makes the compiler wait for each load before issuing the next (s_waitcnt vmcnt(0) per element) and the
// fill becomes a chain of memory round trips.
template <typename T, int CG>
__device__ __forceinline__ bool dw_inside(int v, int h0, int w0, int NH, int NW, int H, int W, int crem, int& off_pix, int& cgi) {
    constexpr int VEC = Vec16<T>::N;
    cgi = v % CG;
    const int pix = v / CG, ix = pix % NW, iy = pix / NW;
    const int ih = h0 + iy, iw = w0 + ix;
    off_pix = ih * W + iw;
    return v < NH * NW * CG && ih >= 0 && ih < H && iw >= 0 && iw < W && cgi * VEC < crem;
}
template <typename T, int CG, int NREG, int NTH = 256>
__device__ __forceinline__ void dw_fetch(uint4 (&reg)[NREG], const T* img, int ld, int h0, int w0, int NH, int NW, int H, int W, int crem) {
    constexpr int VEC = Vec16<T>::N;
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
        int op, cgi;
        const bool ok = dw_inside<T, CG>(threadIdx.x + i * NTH, h0, w0, NH, NW, H, W, crem, op, cgi);
        // 32-bit byte offset from the (uniform) image base: one multiply-add per load instead of a 64-bit address per lane
        // (an image's map is far below 4 GB: H * W * ld * sizeof(T))
        const unsigned boff = ok ? (unsigned)(op * ld + cgi * VEC) * (unsigned)sizeof(T) : 0u;
        reg[i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(img) + boff);
    }
}
template <typename T, int CG, int PIXQ, int NREG>
__device__ __forceinline__ void dw_put(uint4* tile, const uint4 (&reg)[NREG], int h0, int w0, int NH, int NW, int H, int W, int crem) {
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
        const int v = threadIdx.x + i * 256;
        int op, cgi;
        const bool ok = dw_inside<T, CG>(v, h0, w0, NH, NW, H, W, crem, op, cgi);
        if (v < NH * NW * CG) tile[(v / CG) * PIXQ + cgi] = ok ? reg[i] : make_uint4(0u, 0u, 0u, 0u);
    }
}
// stage rows [h0, h0+NH) x cols [w0, w0+NW) of image b (zero outside the map / past channel C) into LDS, pixel pitch PIXQ uint4
template <typename T, int CG, int PIXQ, int NH, int NW>
__device__ __forceinline__ void dw_stage(uint4* tile, const T* img, int ld, int h0, int w0, int H, int W, int crem) {
    constexpr int NREG = (NH * NW * CG + 255) / 256;
    uint4 reg[NREG];
    dw_fetch<T, CG, NREG>(reg, img, ld, h0, w0, NH, NW, H, W, crem);
    dw_put<T, CG, PIXQ, NREG>(tile, reg, h0, w0, NH, NW, H, W, crem);
}

// LDS needs of the tile bodies, in uint4 (the multi-segment launches size one dynamic buffer for the largest body they contain)
template <typename T, int K, int CG> constexpr int dw_tile_smem_q() {
    return DwTile<K, CG>::IH * DwTile<K, CG>::IW * (CG + (CG == 8 ? 2 : 1)) + (K * K * CG * Vec16<T>::N + 3) / 4;
}
template <typename T, int K, int CG> constexpr int dw_wgrad_smem_q() {
    return DwTile<K, CG>::IH * DwTile<K, CG>::IW * (CG + 1) + DwTile<K, CG>::TH * DwTile<K, CG>::TW * (CG + 1) +
           ((K * K + 1) * CG * Vec16<T>::N + 3) / 4;
}

// body of one workgroup (bx = tile, by = channel chunk, bz = weight group); `smem` holds dw_tile_smem_q() uint4
// stat (MODE 0 only, C a multiple of the chunk width CG * VEC): the kernel also leaves, per output pixel and channel chunk,
// (sum, sum of squared deviations from the chunk mean) of the values it writes -- float2 stat[(pixel row) * chunks + chunk], pixel
// rows counted through all images of all weight groups -- so that the LayerNorm that follows (MixFFN_skip, MSTr.py:898-900) never
// reads the map for its statistics: the consumer merges the chunk partials (Chan's parallel variance formula).
template <typename T, int K, int CG, int MODE>      // MODE 0: y = conv(x) (+bias) (+x);  MODE 1: dx = conv^T(dy) (+dy) (+dx)
__device__ __forceinline__ void dw_tile_body(const T* __restrict__ src, int lds_, const T* __restrict__ w,
                                             const T* __restrict__ bias, T* __restrict__ y, int ldy, int B, int H, int W,
                                             int C, int add_input, int accumulate, long long wstride, int tilesW,
                                             int tilesH, const int bx, const int by, const int bz, uint4* smem,
                                             float* __restrict__ stat = nullptr, int nchunks = 0) {
    using D = DwTile<K, CG>;
    constexpr int VEC = Vec16<T>::N, CH = CG * VEC, R = D::R, P = D::P;
    constexpr int PIXQ = CG + (CG == 8 ? 2 : 1);
    uint4* tile = smem;
    float (*wsm)[CH] = reinterpret_cast<float (*)[CH]>(smem + D::IH * D::IW * PIXQ);
    {
        const long long g = bz, img = (long long)B * H * W;
        src += g * img * lds_; y += g * img * ldy; w += g * wstride;
        if (bias) bias += g * wstride;
    }
    const int c0 = by * CH;
    // filter taps: requested together and without a per-lane branch (taps of channels past C read tap 0 and are zeroed when they
    // go to LDS), so that they travel with the tile's pixels instead of costing 3-13 memory round trips before the fill starts
    constexpr int NWL = (K * K * CH + 255) / 256;
    float wv[NWL];
#pragma unroll
    for (int j = 0; j < NWL; ++j) {
        const int i = threadIdx.x + j * 256, cc = i / (K * K), t = i - cc * (K * K);
        const bool ok = i < K * K * CH && c0 + cc < C;
        wv[j] = ldf<T>(w + (ok ? (long long)(c0 + cc) * K * K + t : 0));
    }
    const int tix = bx % tilesW, tiy = (bx / tilesW) % tilesH, b = bx / (tilesW * tilesH);
    const int oh0 = tiy * D::TH, ow0 = tix * D::TW;
    dw_stage<T, CG, PIXQ, D::IH, D::IW>(tile, src + (long long)b * H * W * lds_ + c0, lds_, oh0 - P, ow0 - P, H, W, C - c0);
#pragma unroll
    for (int j = 0; j < NWL; ++j) {
        const int i = threadIdx.x + j * 256, cc = i / (K * K), t = i - cc * (K * K);
        if (i < K * K * CH) wsm[MODE == 1 ? K * K - 1 - t : t][cc] = (c0 + cc < C) ? wv[j] : 0.f;
    }
    __syncthreads();
    const int cg = threadIdx.x % CG, pt = threadIdx.x / CG, run = pt % (D::TW / R), row = pt / (D::TW / R);
    const int oh = oh0 + row, owb = ow0 + run * R, c = c0 + cg * VEC;
    const bool live = !(c >= C || oh >= H || owb >= W);
    if (!live && !(MODE == 0 && stat)) return;               // (with statistics every lane stays for the lane-group reductions)
    float acc[R][VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const float bv = (MODE == 0 && bias && live) ? ldf<T>(bias + c + e) : 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r][e] = bv;
    }
    // K = 3: fully unrolled (all 18 LDS reads in flight); K >= 5: one filter row at a time, or the unrolled loads of all K rows
    // are hoisted and the kernel needs > 256 VGPRs
#pragma unroll(K == 3 ? 3 : 1)
    for (int ky = 0; ky < K; ++ky) {
        float in[R + K - 1][VEC];
#pragma unroll
        for (int i = 0; i < R + K - 1; ++i) unpack16<T>(tile[((row + ky) * D::IW + run * R + i) * PIXQ + cg], in[i]);
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            float wr[VEC];
#pragma unroll
            for (int e = 0; e < VEC; e += 4) {
                const float4 t4 = *reinterpret_cast<const float4*>(&wsm[ky * K + kx][cg * VEC + e]);
                wr[e] = t4.x; wr[e + 1] = t4.y; wr[e + 2] = t4.z; wr[e + 3] = t4.w;
            }
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[r][e] += wr[e] * in[r + kx][e];
        }
        if (ky == P && add_input) {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[r][e] += in[r + P][e];
        }
    }
    if (MODE == 0 && stat) {
        // per pixel: sum over this chunk's CH channels (CG consecutive lanes), then the squared deviations from the chunk mean
        float sm[R], m2[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < VEC; ++e) t += acc[r][e];
            t = tc_group_sum<CG>(t);
            sm[r] = t;
            const float mu = t * (1.0f / (float)CH);
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < VEC; ++e) { const float dlt = acc[r][e] - mu; q += dlt * dlt; }
            q = tc_group_sum<CG>(q);
            m2[r] = q;
        }
        if (!live) return;
        if (cg == 0) {
            const long long prow = (((long long)bz * B + b) * H + oh) * W + owb;
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (owb + r < W) *reinterpret_cast<float2*>(stat + ((prow + r) * nchunks + by) * 2) = make_float2(sm[r], m2[r]);
        }
    }
    T* dst0 = y + (((long long)b * H + oh) * W + owb) * ldy + c;
    if (accumulate) {                                         // all R reads in flight at once (columns past W re-read column owb)
        uint4 old[R];
#pragma unroll
        for (int r = 0; r < R; ++r) old[r] = *reinterpret_cast<const uint4*>(dst0 + (owb + r < W ? (long long)r * ldy : 0));
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float q[VEC];
            unpack16<T>(old[r], q);
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[r][e] += q[e];
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (owb + r < W) *reinterpret_cast<uint4*>(dst0 + (long long)r * ldy) = pack16<T>(acc[r]);
}

// One workgroup per tile: tile indices l, l + 8, ... run on one XCD (whatever the grid offset of the chunk / group), so each XCD takes
// a CONTIGUOUS share of the tile list and neighbouring tiles find each other's halo rows in its L2 (any n: a bijection).
__device__ __forceinline__ int dw_xcd_tile(int l, int n) {
    const int r = l & 7, q = n >> 3, rem = n & 7;
    return r * q + min(r, rem) + (l >> 3);
}
template <typename T, int K, int CG, int MODE>
__global__ __launch_bounds__(256) void dw_tile_kernel(const T* __restrict__ src, int lds_, const T* __restrict__ w,
                                                      const T* __restrict__ bias, T* __restrict__ y, int ldy, int B, int H, int W,
                                                      int C, int add_input, int accumulate, long long wstride, int tilesW,
                                                      int tilesH, float* __restrict__ stat) {
    __shared__ uint4 smem[dw_tile_smem_q<T, K, CG>()];
    dw_tile_body<T, K, CG, MODE>(src, lds_, w, bias, y, ldy, B, H, W, C, add_input, accumulate, wstride, tilesW, tilesH,
                                 dw_xcd_tile((int)blockIdx.x, (int)gridDim.x), blockIdx.y, blockIdx.z, smem, stat, (int)gridDim.y);
}

// dw[c,ky,kx] += sum_pix dy[pix] * x[pix + (ky-P, kx-P)] ; db[c] += sum_pix dy[pix].  x tile (+halo) and dy tile in LDS; a thread
// owns (16-byte channel lane, filter row ky) and a share of the tile's (row, 4-pixel run) units, K x VEC sums in registers
// across all the tiles its workgroup visits; LDS float atomics fold the workgroup, then one global atomic per tap and channel.
// body of one workgroup: bx of gx workgroups walk the tiles of channel chunk by (of gy) for weight group bz
template <typename T, int K, int CG>
__device__ __forceinline__ void dw_tile_wgrad_body(const T* __restrict__ x, int ldx, const T* __restrict__ dy, int lddy,
                                                   float* __restrict__ dw, float* __restrict__ db, int B, int H, int W, int C,
                                                   long long wstride, int tilesW, int tilesH, float* __restrict__ ws_part,
                                                   int* __restrict__ ws_cnt, const int bx, const int by, const int bz, const int gx,
                                                   const int gy, uint4* smem) {
    using D = DwTile<K, CG>;
    constexpr int VEC = Vec16<T>::N, CH = CG * VEC, R = D::R, P = D::P, PIXQ = CG + 1, NT = K * K + 1;
    constexpr int RG = D::PT / K, UNITS = D::TH * (D::TW / R);
    uint4* xt = smem;
    uint4* dt = xt + D::IH * D::IW * PIXQ;
    float (*lacc)[CH] = reinterpret_cast<float (*)[CH]>(dt + D::TH * D::TW * PIXQ);
    {
        const long long g = bz, img = (long long)B * H * W;
        x += g * img * ldx; dy += g * img * lddy; dw += g * wstride;
        if (db) db += g * wstride;
    }
    const int c0 = by * CH;
    for (int i = threadIdx.x; i < NT * CH; i += 256) (&lacc[0][0])[i] = 0.f;
    const int cg = threadIdx.x % CG, wk = threadIdx.x / CG, ky = wk / RG, rg = wk % RG;
    const bool active = ky < K && c0 + cg * VEC < C;
    float acc[K][VEC], accb[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        accb[e] = 0.f;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) acc[kx][e] = 0.f;
    }
    const int ntiles = B * tilesH * tilesW;
    // K = 3 / 5: the next tile's pixels are fetched into registers while this one is multiplied (a workgroup visits 2-7 tiles and
    // only 2-3 workgroups fit a CU, so an un-prefetched fill is exposed HBM latency); K = 7 has no registers to spare
    constexpr bool PF = K <= 5;
    constexpr int NX = PF ? (D::IH * D::IW * CG + 255) / 256 : 1, ND = PF ? (D::TH * D::TW * CG + 255) / 256 : 1;
    uint4 xr[NX], dr[ND];
    auto tile_org = [&](int tidx, int& b, int& oh0, int& ow0) {
        const int tix = tidx % tilesW, tiy = (tidx / tilesW) % tilesH;
        b = tidx / (tilesW * tilesH); oh0 = tiy * D::TH; ow0 = tix * D::TW;
    };
    auto fetch = [&](int tidx) {
        int b, oh0, ow0;
        tile_org(tidx, b, oh0, ow0);
        dw_fetch<T, CG, NX>(xr, x + (long long)b * H * W * ldx + c0, ldx, oh0 - P, ow0 - P, D::IH, D::IW, H, W, C - c0);
        dw_fetch<T, CG, ND>(dr, dy + (long long)b * H * W * lddy + c0, lddy, oh0, ow0, D::TH, D::TW, H, W, C - c0);
    };
    if (PF && bx < ntiles) fetch(bx);
    for (int tidx = bx; tidx < ntiles; tidx += gx) {
        __syncthreads();
        {
            int b, oh0, ow0;
            tile_org(tidx, b, oh0, ow0);
            if (PF) {
                dw_put<T, CG, PIXQ, NX>(xt, xr, oh0 - P, ow0 - P, D::IH, D::IW, H, W, C - c0);
                dw_put<T, CG, PIXQ, ND>(dt, dr, oh0, ow0, D::TH, D::TW, H, W, C - c0);
            } else {
                dw_stage<T, CG, PIXQ, D::IH, D::IW>(xt, x + (long long)b * H * W * ldx + c0, ldx, oh0 - P, ow0 - P, H, W, C - c0);
                dw_stage<T, CG, PIXQ, D::TH, D::TW>(dt, dy + (long long)b * H * W * lddy + c0, lddy, oh0, ow0, H, W, C - c0);
            }
        }
        __syncthreads();
        if (PF && tidx + gx < ntiles) fetch(tidx + gx);
        if (active) {
            for (int u = rg; u < UNITS; u += RG) {
                const int row = u / (D::TW / R), run = u % (D::TW / R);
                float d[R][VEC], in[R + K - 1][VEC];
#pragma unroll
                for (int r = 0; r < R; ++r) unpack16<T>(dt[(row * D::TW + run * R + r) * PIXQ + cg], d[r]);
#pragma unroll
                for (int i = 0; i < R + K - 1; ++i) unpack16<T>(xt[((row + ky) * D::IW + run * R + i) * PIXQ + cg], in[i]);
#pragma unroll
                for (int kx = 0; kx < K; ++kx)
#pragma unroll
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int e = 0; e < VEC; ++e) acc[kx][e] += d[r][e] * in[r + kx][e];
                if (ky == 0) {
#pragma unroll
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int e = 0; e < VEC; ++e) accb[e] += d[r][e];
                }
            }
        }
    }
    // Workgroup sums through the (now free) LDS tiles instead of LDS atomics (RG threads per word, K x VEC + VEC atomics per thread:
    // a fifth of the MixFFN mid-backward kernel's cycles before the same change there): every thread parks SLOTS filter columns
    // ([value][thread]), one thread per output word adds its RG contributors.
    {
        constexpr int SLOTS_FIT = ((D::IH * D::IW + D::TH * D::TW) * PIXQ * 4) / (256 * VEC);
        constexpr int SLOTS = SLOTS_FIT > K + 1 ? K + 1 : SLOTS_FIT;
        static_assert(SLOTS >= 1 && K * RG <= 256 / CG, "scratch / lane roles");
        float* scr = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int k0 = 0; k0 < K + 1; k0 += SLOTS) {               // column K = the bias sums
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < SLOTS; ++kk) {
                const int kx = k0 + kk;
                if (kx <= K) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) scr[(kk * VEC + e) * 256 + threadIdx.x] = kx < K ? acc[kx < K ? kx : 0][e] : accb[e];
                }
            }
            __syncthreads();
            for (int f = threadIdx.x; f < SLOTS * K * CH; f += 256) {
                const int kk = f / (K * CH), rem = f - kk * (K * CH), kyo = rem / CH, c = rem - kyo * CH, kx = k0 + kk;
                if (kx > K || (kx == K && kyo > 0)) continue;
                const float* col = scr + (kk * VEC + (c % VEC)) * 256 + (c / VEC) + CG * (kyo * RG);
                float v = 0.f;
#pragma unroll
                for (int g = 0; g < RG; ++g) v += col[g * CG];
                lacc[kx < K ? kyo * K + kx : K * K][c] = v;
            }
        }
    }
    __syncthreads();
#ifdef TC_DBG_DW_NOFOLD                                            // what-if build (scripts/exp): weight gradients dropped, timing only
    return;
#endif
    float* lflat = &lacc[0][0];
    int gm = 1;
    const float* pgroup = nullptr;
    if (ws_part && !ws_cnt) {
        // Deferred (tc_dwconv_bwd with ws_bytes < 0): the sums go, plainly, to a buffer of this launch's own -- [group * chunks + chunk][walker]
        // [tap][channel] -- and tc_dw_fold adds the walkers of every such launch of a backward leg in one launch: no write-through stores, no
        // arrival counter, no last-arriver fold and no contended atomics at the tail of every launch (0.19 ms of a 12.2 ms step).
        float* part = ws_part + ((long long)(bz * gy + by) * gx + bx) * (NT * CH);
        for (int f = threadIdx.x; f < NT * CH; f += 256) part[f] = lflat[f];
        return;
    }
    if (ws_part) {
        // Two-level fold (same protocol as the GEMM split-K fix-up): the workgroups of a (group, channel chunk) park their sums in
        // the workspace, 16 consecutive ones share an arrival counter, the last to arrive adds the 16 and is the only one that
        // touches dw / db atomically -- a contended fp32 atomic costs ~0.13 us and 128-170 workgroups used to queue on every word.
        constexpr int FG = DW_FOLD;
        const int chain = bz * gy + by, grp = bx / FG, ngrp = (gx + FG - 1) / FG;
        gm = min(FG, gx - grp * FG);
        float* part = ws_part + ((long long)chain * gx + bx) * (NT * CH);
        pgroup = ws_part + ((long long)chain * gx + grp * FG) * (NT * CH);
        if (gm > 1) {
            for (int f = threadIdx.x; f < NT * CH; f += 256) __hip_atomic_store(part + f, lflat[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0);                          // vmcnt(0): THIS thread's write-through stores have been acknowledged
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // (the barrier alone does not wait for stores in flight)
            __syncthreads();
            __shared__ int s_last;
            if (threadIdx.x == 0) {
                int* c = ws_cnt + chain * ngrp + grp;
                const int old = atomicAdd(c, 1);
                s_last = (old == gm - 1);
                if (s_last) atomicExch(c, 0);
            }
            __syncthreads();
            if (!s_last) return;
        }
    }
    for (int f = threadIdx.x; f < NT * CH; f += 256) {
        const int t = f / CH, cc = f - t * CH, ch = c0 + cc;                // lacc is [tap][channel]
        if (ch >= C) continue;
        float v;
        if (gm > 1) {
            float tmp[DW_FOLD];                                   // all 16 loads in flight before the first add (they bypass L2: ~2 us each)
#pragma unroll
            for (int m = 0; m < DW_FOLD; ++m)
                tmp[m] = m < gm ? __hip_atomic_load(pgroup + (long long)m * (NT * CH) + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
            v = 0.f;
#pragma unroll
            for (int m = 0; m < DW_FOLD; ++m) v += tmp[m];
        } else {
            v = lflat[f];
        }
        if (t < K * K) atomicAdd(dw + (long long)ch * K * K + t, v);
        else if (db) atomicAdd(db + ch, v);
    }
}

template <typename T, int K, int CG>
__global__ __launch_bounds__(256) void dw_tile_wgrad_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dy, int lddy,
                                                            float* __restrict__ dw, float* __restrict__ db, int B, int H, int W, int C,
                                                            long long wstride, int tilesW, int tilesH, float* __restrict__ ws_part,
                                                            int* __restrict__ ws_cnt) {
    __shared__ uint4 smem[dw_wgrad_smem_q<T, K, CG>()];
    dw_tile_wgrad_body<T, K, CG>(x, ldx, dy, lddy, dw, db, B, H, W, C, wstride, tilesW, tilesH, ws_part, ws_cnt, blockIdx.x, blockIdx.y,
                                 blockIdx.z, gridDim.x, gridDim.y, smem);
}

// Input gradient and weight gradient of one stride-1 convolution in ONE grid: the first n_wg workgroups of a row are the (long-running)
// weight-gradient walkers, the rest the input-gradient tiles -- both read dy, neither reads the other's result, and as two launches each
// paid its own ramp and drain (5 + 11 us at the cpe shapes).
template <typename T, int K, int CG>
__global__ __launch_bounds__(256) void dw_tile_bwd_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ w, T* __restrict__ dx, int lddx,
                                                          const T* __restrict__ x, int ldx, float* __restrict__ dw, float* __restrict__ db,
                                                          int B, int H, int W, int C, int add_input, int accumulate, long long wstride, int tilesW,
                                                          int tilesH, float* __restrict__ ws_part, int* __restrict__ ws_cnt, int n_wg, int n_in) {
    constexpr int SQ = dw_tile_smem_q<T, K, CG>() > dw_wgrad_smem_q<T, K, CG>() ? dw_tile_smem_q<T, K, CG>() : dw_wgrad_smem_q<T, K, CG>();
    __shared__ uint4 smem[SQ];
    if ((int)blockIdx.x < n_wg)
        dw_tile_wgrad_body<T, K, CG>(x, ldx, dy, lddy, dw, db, B, H, W, C, wstride, tilesW, tilesH, ws_part, ws_cnt, blockIdx.x, blockIdx.y, blockIdx.z,
                                     n_wg, gridDim.y, smem);
    else
        dw_tile_body<T, K, CG, 1>(dy, lddy, w, nullptr, dx, lddx, B, H, W, C, add_input, accumulate, wstride, tilesW, tilesH,
                                  dw_xcd_tile((int)blockIdx.x - n_wg, n_in), blockIdx.y, blockIdx.z, smem, nullptr, (int)gridDim.y);
}

// 16-byte lanes per pixel that waste the fewest channels (ties: the widest)
template <typename T> int dw_pick_cg(int C) {
    constexpr int VEC = Vec16<T>::N;
    int best = 8, waste = (C + 8 * VEC - 1) / (8 * VEC) * (8 * VEC) - C;
    for (int cg = 4; cg >= 2; cg >>= 1) {
        const int wst = (C + cg * VEC - 1) / (cg * VEC) * (cg * VEC) - C;
        if (wst < waste) { best = cg; waste = wst; }
    }
    return best;
}
template <typename T> bool dw_tile_ok(const void* a, int lda, const void* b, int ldb, int C) {
    constexpr int VEC = Vec16<T>::N;
    return C % VEC == 0 && lda % VEC == 0 && ldb % VEC == 0 && (uintptr_t)a % 16 == 0 && (uintptr_t)b % 16 == 0;
}

template <typename T, int MODE>
int launch_tile(const void* src, int lds_, const void* w, const void* bias, void* y, int ldy, const void* dy, int lddy, float* dw,
                float* db, int B, int H, int W, int C, int k, int add_input, int accumulate, int groups, long long wstride,
                hipStream_t s, void* ws = nullptr, long long ws_bytes = 0, float* stat = nullptr) {
    constexpr int VEC = Vec16<T>::N;
    const int cg = dw_pick_cg<T>(C);
    if (stat && (MODE != 0 || C % (cg * VEC))) return TC_ERR_ARG;
    const int chunks = (C + cg * VEC - 1) / (cg * VEC);
    const int TH = (256 / cg) / 4, tilesW = (W + 15) / 16, tilesH = (H + TH - 1) / TH;
    const long long ntiles = (long long)B * tilesW * tilesH;
    if (ntiles > 0x7fffffffLL) return TC_ERR_ARG;
#define TC_TILE(KK, CGG)                                                                                                                \
    if (MODE == 2) {                                                                                                                    \
        int gx = tc_dw_wg_target() / (chunks * groups);   /* measured best with the fold: one workgroup per CU, each visiting ~7 tiles */ /* contended fp32 atomics cost ~0.13 us each: few contributors per word */          \
        if (gx > ntiles) gx = (int)ntiles;                                                                                              \
        constexpr int NTC = ((KK) * (KK) + 1) * (CGG) * VEC;                                                                            \
        float* wp = nullptr; int* wc = nullptr;                                                                                         \
        if (ws && (uintptr_t)ws % 16 == 0 && ws_bytes >= 16384 + (long long)chunks * groups * (gx < 1 ? 1 : gx) * NTC * 4 &&             \
            (long long)chunks * groups * (((gx < 1 ? 1 : gx) + DW_FOLD - 1) / DW_FOLD) <= 4096) {                                                      \
            wc = reinterpret_cast<int*>(ws); wp = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + 16384);                         \
        }                                                                                          \
        gx = gx < 1 ? 1 : gx;                                                                                                           \
        dim3 grid((unsigned)(ntiles < gx ? ntiles : gx), chunks, groups);                                                               \
        hipLaunchKernelGGL((dw_tile_wgrad_kernel<T, KK, CGG>), grid, dim3(256), 0, s, (const T*)src, lds_, (const T*)dy, lddy, dw, db,  \
                           B, H, W, C, wstride, tilesW, tilesH, wp, wc);                                                                        \
    } else {                                                                                                                            \
        dim3 grid((unsigned)ntiles, chunks, groups);                                                                                    \
        hipLaunchKernelGGL((dw_tile_kernel<T, KK, CGG, (MODE == 2 ? 0 : MODE)>), grid, dim3(256), 0, s, (const T*)src, lds_,            \
                           (const T*)w, (const T*)bias, (T*)y, ldy, B, H, W, C, add_input, accumulate, wstride, tilesW, tilesH, stat);  \
    }
#define TC_TILE_K(KK) { if (cg == 8) { TC_TILE(KK, 8) } else if (cg == 4) { TC_TILE(KK, 4) } else { TC_TILE(KK, 2) } }
    if (k == 3) TC_TILE_K(3) else if (k == 5) TC_TILE_K(5) else TC_TILE_K(7)
#undef TC_TILE_K
#undef TC_TILE
    return tc_launch_status();
}

struct DwGeom { int cg, ch, chunks, gx, nt; };
template <typename T> DwGeom dw_bwd_geom(int B, int H, int W, int C, int k, int groups) {     // ONE statement of the one-launch backward's geometry: launch_tile_bwd launches it, tc_dwconv_bwd_plan hands it to the fold
    constexpr int VEC = Vec16<T>::N;
    DwGeom g;
    g.cg = dw_pick_cg<T>(C); g.ch = g.cg * VEC; g.chunks = (C + g.ch - 1) / g.ch;
    const int TH = (256 / g.cg) / 4, tilesW = (W + 15) / 16, tilesH = (H + TH - 1) / TH;
    const long long ntiles = (long long)B * tilesW * tilesH;
    long long gx = tc_dw_wg_target() / (g.chunks * groups);
    if (gx > ntiles) gx = ntiles;
    g.gx = (int)(gx < 1 ? 1 : gx); g.nt = k * k + 1;
    return g;
}

template <typename T>
int launch_tile_bwd(const void* dy, int lddy, const void* w, void* dx, int lddx, const void* x, int ldx, float* dw, float* db, int B, int H, int W,
                    int C, int k, int add_input, int accumulate, int groups, long long wstride, hipStream_t s, void* ws, long long ws_bytes) {
    constexpr int VEC = Vec16<T>::N;
    const DwGeom geo = dw_bwd_geom<T>(B, H, W, C, k, groups);      // (the plan the caller sized and described its deferred sums with)
    const int cg = geo.cg, chunks = geo.chunks;
    const int TH = (256 / cg) / 4, tilesW = (W + 15) / 16, tilesH = (H + TH - 1) / TH;
    const long long ntiles = (long long)B * tilesW * tilesH;
    if (ntiles > 0x3fffffffLL) return TC_ERR_ARG;
#define TC_TILE(KK, CGG) {                                                                                                              \
        const int gx = geo.gx;                                                                                                          \
        constexpr int NTC = ((KK) * (KK) + 1) * (CGG) * VEC;                                                                            \
        float* wp = nullptr; int* wc = nullptr;                                                                                         \
        if (ws && ws_bytes < 0) {                                  /* deferred: the caller's own buffer, folded later by tc_dw_fold */   \
            if (-ws_bytes < (long long)chunks * groups * gx * NTC * 4 || (uintptr_t)ws % 16) return TC_ERR_ARG;                          \
            wp = reinterpret_cast<float*>(ws);                                                                                          \
        } else if (ws && (uintptr_t)ws % 16 == 0 && ws_bytes >= 16384 + (long long)chunks * groups * gx * NTC * 4 &&                     \
            (long long)chunks * groups * ((gx + DW_FOLD - 1) / DW_FOLD) <= 4096) {                                                      \
            wc = reinterpret_cast<int*>(ws); wp = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + 16384);                         \
        }                                                                                                                               \
        dim3 grid((unsigned)(gx + ntiles), chunks, groups);                                                                             \
        hipLaunchKernelGGL((dw_tile_bwd_kernel<T, KK, CGG>), grid, dim3(256), 0, s, (const T*)dy, lddy, (const T*)w, (T*)dx, lddx,       \
                           (const T*)x, ldx, dw, db, B, H, W, C, add_input, accumulate, wstride, tilesW, tilesH, wp, wc, gx, (int)ntiles); }
#define TC_TILE_K(KK) { if (cg == 8) TC_TILE(KK, 8) else if (cg == 4) TC_TILE(KK, 4) else TC_TILE(KK, 2) }
    if (k == 3) TC_TILE_K(3) else if (k == 5) TC_TILE_K(5) else TC_TILE_K(7)
#undef TC_TILE_K
#undef TC_TILE
    return tc_launch_status();
}

template <typename T, bool BWD>
int launch_dw(const void* x, int ldx, const void* w, const void* bias, void* y, int ldy, int B, int H, int W, int C, int k,
              int stride, int add_input, int accumulate, hipStream_t s) {
    const int P = (k - 1) / 2;
    const int Ho = (H + 2 * P - k) / stride + 1, Wo = (W + 2 * P - k) / stride + 1;
    const long long npix = (long long)B * (BWD ? H * W : Ho * Wo);
    if (npix >= 0x7fffffffLL || (stride != 1 && stride != 2)) return TC_ERR_ARG;
    dim3 grid(tc_blocks(npix, 16, 4096), (C + 63) / 64), block(256);       // one pixel row of 16 per workgroup pass: maps of 12.5 k pixels fill the chip
#define TC_DW2(KK, SS) hipLaunchKernelGGL((dw_kernel<T, KK, BWD, SS>), grid, block, 0, s, (const T*)x, ldx, (const T*)w, (const T*)bias, \
                                          (T*)y, ldy, B, H, W, Ho, Wo, C, add_input, accumulate)
#define TC_DW(KK) { if (stride == 1) TC_DW2(KK, 1); else TC_DW2(KK, 2); }
    if (k == 3) TC_DW(3) else if (k == 5) TC_DW(5) else TC_DW(7)
#undef TC_DW
#undef TC_DW2
    return tc_launch_status();
}

// ------------------------------------------------------------------------------------------------------------
// Multi-segment launches: up to three stride-1 depthwise convolutions with different k / channel counts on column slices of the
// same maps (ConvRelPosEnc: 3x3 / 5x5 / 7x7 on 2 / 3 / 3 heads' channels, MSTr.py:785-816) in ONE grid.  Each is a 10-30 us
// latency-bound launch on its own; side by side they cost what the slowest costs.
struct DwSegDev {
    const void* src; const void* w; const void* bias; void* y; const void* dy; float* dw; float* db; float* wsp; int* wsc; float* stat;
    int C, k, cg, chunks, tilesW, tilesH, gx, blk0, lds_, ldy, lddy, B, H, W;
    int nt;           // mode 3: input-gradient tiles per chunk row, after the gx weight-gradient walkers
};
constexpr int DW_MULTI_MAX = 4;
struct DwMultiDev { DwSegDev s[DW_MULTI_MAX]; int n, add_input, accumulate; long long wstride; };

#define TC_DW_CASES(X) X(3, 2) X(3, 4) X(3, 8) X(5, 2) X(5, 4) X(5, 8) X(7, 2) X(7, 4) X(7, 8)

template <typename T, int MODE>
__global__ __launch_bounds__(256) void dw_multi_kernel(DwMultiDev a) {
    extern __shared__ uint4 dsm[];
    int lin = blockIdx.x, si = 0;
    if (a.n > 1 && lin >= a.s[1].blk0) si = 1;
    if (a.n > 2 && lin >= a.s[2].blk0) si = 2;
    if (a.n > 3 && lin >= a.s[3].blk0) si = 3;
    const DwSegDev& g = a.s[si];
    lin -= g.blk0;
    const int bx = dw_xcd_tile(lin % g.gx, g.gx), by = lin / g.gx;
#define TC_CASE(KK, CGG)                                                                                                            \
    if (g.k == KK && g.cg == CGG) {                                                                                                 \
        dw_tile_body<T, KK, CGG, MODE>((const T*)g.src, g.lds_, (const T*)g.w, (const T*)g.bias, (T*)g.y, g.ldy, g.B, g.H, g.W, g.C, \
                                       a.add_input, a.accumulate, a.wstride, g.tilesW, g.tilesH, bx, by, blockIdx.y, dsm,           \
                                       MODE == 0 ? g.stat : nullptr, g.chunks);                                                     \
        return;                                                                                                                     \
    }
    TC_DW_CASES(TC_CASE)
#undef TC_CASE
}

template <typename T>
__global__ __launch_bounds__(256) void dw_multi_wgrad_kernel(DwMultiDev a) {
    extern __shared__ uint4 dsm[];
    int lin = blockIdx.x, si = 0;
    if (a.n > 1 && lin >= a.s[1].blk0) si = 1;
    if (a.n > 2 && lin >= a.s[2].blk0) si = 2;
    if (a.n > 3 && lin >= a.s[3].blk0) si = 3;
    const DwSegDev& g = a.s[si];
    lin -= g.blk0;
    const int bx = lin % g.gx, by = lin / g.gx;
#define TC_CASE(KK, CGG)                                                                                                            \
    if (g.k == KK && g.cg == CGG) {                                                                                                 \
        dw_tile_wgrad_body<T, KK, CGG>((const T*)g.src, g.lds_, (const T*)g.dy, g.lddy, g.dw, g.db, g.B, g.H, g.W, g.C, a.wstride,    \
                                       g.tilesW, g.tilesH, g.wsp, g.wsc, bx, by, blockIdx.y, g.gx, g.chunks, dsm);                  \
        return;                                                                                                                     \
    }
    TC_DW_CASES(TC_CASE)
#undef TC_CASE
}

// mode 3: both gradients of every segment in one grid -- per segment the weight-gradient walkers (src = x, dy -> dw, db) first, then the
// input-gradient tiles (dy, w -> y = dx)
template <typename T>
__global__ __launch_bounds__(256) void dw_multi_bwd_kernel(DwMultiDev a) {
    extern __shared__ uint4 dsm[];
    int lin = blockIdx.x, si = 0;
    if (a.n > 1 && lin >= a.s[1].blk0) si = 1;
    if (a.n > 2 && lin >= a.s[2].blk0) si = 2;
    if (a.n > 3 && lin >= a.s[3].blk0) si = 3;
    const DwSegDev& g = a.s[si];
    lin -= g.blk0;
    const int nw = g.gx * g.chunks;
    const bool wg = lin < nw;
    if (!wg) lin -= nw;
    const int per = wg ? g.gx : g.nt;
    const int bx = wg ? lin % per : dw_xcd_tile(lin % per, per), by = lin / per;
#define TC_CASE(KK, CGG)                                                                                                            \
    if (g.k == KK && g.cg == CGG) {                                                                                                 \
        if (wg) dw_tile_wgrad_body<T, KK, CGG>((const T*)g.src, g.lds_, (const T*)g.dy, g.lddy, g.dw, g.db, g.B, g.H, g.W, g.C, a.wstride, \
                                               g.tilesW, g.tilesH, g.wsp, g.wsc, bx, by, blockIdx.y, g.gx, g.chunks, dsm);          \
        else dw_tile_body<T, KK, CGG, 1>((const T*)g.dy, g.lddy, (const T*)g.w, nullptr, (T*)g.y, g.ldy, g.B, g.H, g.W, g.C,        \
                                         a.add_input, a.accumulate, a.wstride, g.tilesW, g.tilesH, bx, by, blockIdx.y, dsm, nullptr,   \
                                         g.chunks);                                                                                 \
        return;                                                                                                                     \
    }
    TC_DW_CASES(TC_CASE)
#undef TC_CASE
}

template <typename T> int dw_smem_q(int k, int cg, bool wgrad) {
#define TC_CASE(KK, CGG) if (k == KK && cg == CGG) return wgrad ? dw_wgrad_smem_q<T, KK, CGG>() : dw_tile_smem_q<T, KK, CGG>();
    TC_DW_CASES(TC_CASE)
#undef TC_CASE
    return 0;
}

// Weight-gradient walkers of segment i of a multi-segment launch: ~tc_dw_wg_target() workgroups in total, shared out in proportion to each
// segment's tiles x chunks x (k + 2) (an even split gave the 56x56 map of a bridge layer 16 workgroups of 28 tiles each next to 1-tile
// workgroups of the 7x7 map).  ONE statement: launch_multi launches it, tc_dwconv_multi_plan describes the deferred sums with it.
template <typename T> long long dw_multi_work(const TcDwSeg* segs, int nseg) {
    constexpr int VEC = Vec16<T>::N;
    long long total_work = 0;
    for (int i = 0; i < nseg; ++i) {
        const int cg = dw_pick_cg<T>(segs[i].C), TH = (256 / cg) / 4;
        total_work += (long long)segs[i].B * ((segs[i].W + 15) / 16) * ((segs[i].H + TH - 1) / TH) * ((segs[i].C + cg * VEC - 1) / (cg * VEC)) *
                      (segs[i].k + 2);                  // cost per tile grows roughly with k (k filter rows per thread, k-wide window)
    }
    return total_work;
}
inline long long dw_multi_gx(long long ntiles, int k, long long total_work, int groups) {
    long long gx = (long long)((double)tc_dw_wg_target() * (double)ntiles * (k + 2) / (double)(total_work > 0 ? total_work : 1) / groups + 0.5);
    return gx < 1 ? 1 : (gx > ntiles ? ntiles : gx);
}

template <typename T>
int launch_multi(const TcDwSeg* segs, int nseg, int mode, int add_input, int accumulate, int groups, long long wstride, void* ws,
                 long long ws_bytes, hipStream_t s) {
    constexpr int VEC = Vec16<T>::N;
    DwMultiDev a;
    a.n = nseg; a.add_input = add_input; a.accumulate = accumulate;
    a.wstride = wstride;
    long long blk = 0, part_floats = 0, cnts = 0;
    int smem_q = 0;
    const bool wgm = mode >= 2;                                  // weight-gradient walkers in the grid (mode 3: followed by the input-gradient tiles)
    const bool defer = wgm && ws && ws_bytes < 0;                // the walkers' sums go to the caller's own buffer (tc_dw_fold adds them later)
    if (defer && (uintptr_t)ws % 16) return TC_ERR_ARG;
    const bool have_ws = !defer && wgm && ws && (uintptr_t)ws % 16 == 0 && ws_bytes > 16384;
    const long long total_work = dw_multi_work<T>(segs, nseg);
    for (int i = 0; i < nseg; ++i) {
        const TcDwSeg& g = segs[i];
        DwSegDev& d = a.s[i];
        d.src = g.x; d.w = g.w; d.bias = g.bias; d.y = g.y; d.dy = g.dy; d.dw = g.dw; d.db = g.db; d.C = g.C; d.k = g.k;
        d.lds_ = g.ldx; d.ldy = g.ldy; d.lddy = g.lddy; d.B = g.B; d.H = g.H; d.W = g.W;
        const int B = g.B, H = g.H, W = g.W;
        d.cg = dw_pick_cg<T>(g.C);
        d.stat = mode == 0 ? g.stat : nullptr;
        if (d.stat && (g.k != 3 || g.C % (d.cg * VEC))) return TC_ERR_ARG;
        d.chunks = (g.C + d.cg * VEC - 1) / (d.cg * VEC);
        const int TH = (256 / d.cg) / 4;
        d.tilesW = (W + 15) / 16; d.tilesH = (H + TH - 1) / TH;
        const long long ntiles = (long long)B * d.tilesW * d.tilesH;
        d.nt = (int)ntiles;
        if (wgm) {
            // ~256 workgroups in total, shared out in proportion to each segment's tiles x chunks (an even split gave the 56x56
            // map of a bridge layer 16 workgroups of 28 tiles each next to 1-tile workgroups of the 7x7 map)
            const long long gx = dw_multi_gx(ntiles, g.k, total_work, groups);
            d.gx = (int)gx;
            const long long nt_ch = (long long)(g.k * g.k + 1) * d.cg * VEC;
            d.wsc = have_ws ? reinterpret_cast<int*>(ws) + cnts : nullptr;
            d.wsp = have_ws ? reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + 16384) + part_floats : nullptr;
            if (defer) d.wsp = reinterpret_cast<float*>(ws) + part_floats;
            cnts += (long long)d.chunks * groups * ((gx + DW_FOLD - 1) / DW_FOLD);
            part_floats += (long long)d.chunks * groups * gx * nt_ch;
        } else {
            d.gx = (int)ntiles; d.wsc = nullptr; d.wsp = nullptr;
        }
        d.blk0 = (int)blk;
        blk += ((long long)d.gx + (mode == 3 ? ntiles : 0)) * d.chunks;
        const int q = dw_smem_q<T>(g.k, d.cg, wgm), q2 = mode == 3 ? dw_smem_q<T>(g.k, d.cg, false) : 0;
        smem_q = q > smem_q ? q : smem_q;
        smem_q = q2 > smem_q ? q2 : smem_q;
    }
    if (blk > 0x7fffffffLL) return TC_ERR_ARG;
    if (defer && part_floats * 4 > -ws_bytes) return TC_ERR_ARG;
    if (wgm && have_ws && (cnts > 4096 || 16384 + part_floats * 4 > ws_bytes))
        for (int i = 0; i < nseg; ++i) { a.s[i].wsc = nullptr; a.s[i].wsp = nullptr; }
    const size_t smem = (size_t)smem_q * 16;
    dim3 grid((unsigned)blk, groups);
    if (mode == 0) {
        if (smem > 64 * 1024) hipFuncSetAttribute((const void*)dw_multi_kernel<T, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((dw_multi_kernel<T, 0>), grid, dim3(256), smem, s, a);
    } else if (mode == 1) {
        if (smem > 64 * 1024) hipFuncSetAttribute((const void*)dw_multi_kernel<T, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((dw_multi_kernel<T, 1>), grid, dim3(256), smem, s, a);
    } else if (mode == 2) {
        if (smem > 64 * 1024) hipFuncSetAttribute((const void*)dw_multi_wgrad_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((dw_multi_wgrad_kernel<T>), grid, dim3(256), smem, s, a);
    } else {
        if (smem > 64 * 1024) hipFuncSetAttribute((const void*)dw_multi_bwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((dw_multi_bwd_kernel<T>), grid, dim3(256), smem, s, a);
    }
    return tc_launch_status();
}

// ------------------------------------------------------------------------------------------------------------
// MixFFN_skip middle, backward (MSTr.py:889-902: out = fc2(GELU(LN(d))), d = dw3x3(h) + h, h = fc1(x)) in ONE kernel:
//   in : gp = dOut W2 (.) GELU'(u)          (the fc2 input-gradient GEMM's epilogue, u = xhat * gamma + beta)
//        S1[row] = sum_c gp * gamma, S2[row] = sum_c gp * gamma * xhat   (per-row partials left by that epilogue)
//   dd = rstd * (gp * gamma - S1 / C - xhat * S2 / C)         LayerNorm backward, finished per pixel of the haloed tile
//   dh = conv3x3^T(dd) + dd                                    depthwise input gradient + the skip
//   dw[c, tap] += sum dd * h(shifted), db[c] += sum dd, dgamma[c] += sum gp * xhat, dbeta[c] += sum gp
// i.e. the LayerNorm backward, the depthwise input gradient and the depthwise weight gradient of the unfused form (three launches,
// seven passes over hidden-width maps) read gp, d and h once and write dh once.  Work decomposition as dw_tile_wgrad_body: a
// workgroup owns one channel chunk (CG * VEC channels) and walks its share of the TH x 16 pixel tiles with every parameter sum in
// registers, LDS atomics fold the workgroup, the 16-way workspace fold and one atomic per word finish.
struct FfnSegDev {
    const void* gp; const void* d; const void* h; void* dh; const float* stat; const float* part2; const void* w; const void* gamma;
    float* dw; float* db; float* dgamma; float* dbeta; float* wsp; int* wsc;
    int C, ldg, ldd, ldh, lddh, B, H, W, nch2, chunks, tilesW, tilesH, gx, blk0;
};
constexpr int FFN_MULTI_MAX = 4;
struct FfnMultiDev { FfnSegDev s[FFN_MULTI_MAX]; int n; int dbg_nofold; long long wstride; };

template <typename T> struct FfnTile {
    static constexpr int K = 3, CG = 8, VEC = Vec16<T>::N, CH = CG * VEC, R = 4, TW = 16, TH = 8, P = 1, IW = TW + 2, IH = TH + 2;
    static constexpr int PIXQ = CG + 1, NT = K * K + 3;            // taps, conv bias, dgamma, dbeta
    static constexpr int smem_q = 2 * IH * IW * PIXQ + IH * IW + (K * K * CH + CH + NT * CH + 3) / 4;
};

// NTH = 256: a thread owns a whole 16-byte channel vector in the convolution / weight-gradient phases (one wave per SIMD).
// NTH = 512: two threads share a vector (SV = VEC / 2 channels each): half the registers per thread, two waves per SIMD, so that one
// wave's LDS reads overlap the other's arithmetic -- with a single resident wave the LDS and VALU phases of a tile ran back to back.
#ifdef TC_MID_TIMING
__device__ long long g_mid_dbg[64 * 16];
#define MSTAMP(k) do { if (tid == 0 && by == 0 && bz == 0 && bx < 64) { const long long t_ = __builtin_readcyclecounter(); g_mid_dbg[bx * 16 + (k)] += t_ - mt_; mt_ = t_; } } while (0)
#define MSTAMP_INIT() long long mt_ = __builtin_readcyclecounter(); if (tid == 0 && by == 0 && bz == 0 && bx < 64) { for (int k_ = 0; k_ < 16; ++k_) g_mid_dbg[bx * 16 + k_] = 0; }
#else
#define MSTAMP(k)
#define MSTAMP_INIT()
#endif
template <typename T, int NTH>
__device__ __forceinline__ void ffn_mid_bwd_body(const FfnSegDev& a, long long wstride, const int bx, const int by, const int bz,
                                                 uint4* smem, const int dbg_nofold = 0) {
    using D = FfnTile<T>;
    constexpr int K = D::K, CG = D::CG, VEC = D::VEC, CH = D::CH, R = D::R, IW = D::IW, IH = D::IH, PIXQ = D::PIXQ, NT = D::NT;
    constexpr int HS = NTH / 256, SV = VEC / HS, S2 = SV / 2, V2 = VEC / 2, SVB = 16 / HS;       // sub-vector: SV channels = SVB bytes
    constexpr int NPIX = IH * IW, NX = (NPIX * CG + NTH - 1) / NTH, RG = (256 / CG) / K, UNITS = D::TH * (D::TW / R);
    uint4* ddt = smem;                                           // dd on the haloed tile, storage type
    uint4* ht = ddt + NPIX * PIXQ;                               // h on the haloed tile
    float4* pst = reinterpret_cast<float4*>(ht + NPIX * PIXQ);   // per haloed pixel: mean, rstd, S1 / C, S2 / C
    float (*wsm)[CH] = reinterpret_cast<float (*)[CH]>(pst + NPIX);      // flipped taps
    float* gsm = &wsm[K * K][0];
    float (*lacc)[CH] = reinterpret_cast<float (*)[CH]>(gsm + CH);
    const char* ddb = reinterpret_cast<const char*>(ddt);
    const char* htb = reinterpret_cast<const char*>(ht);
    const int B = a.B, H = a.H, W = a.W, C = a.C;
    const long long img = (long long)B * H * W, grow = (long long)bz * img;     // first pixel row of this weight group
    const T* gp = (const T*)a.gp + grow * a.ldg;
    const T* dm = (const T*)a.d + grow * a.ldd;
    const T* hm = (const T*)a.h + grow * a.ldh;
    T* dh = (T*)a.dh + grow * a.lddh;
    const float* stat = a.stat + grow * 2;
    const float* part2 = a.part2 + grow * a.nch2 * 2;
    const T* w = (const T*)a.w + bz * wstride;
    const T* gamma = (const T*)a.gamma + bz * wstride;
    const int c0 = by * CH, tid = threadIdx.x;
    MSTAMP_INIT();
    for (int i = tid; i < K * K * CH; i += NTH) {
        const int cc = i / (K * K), t = i - cc * (K * K);
        wsm[K * K - 1 - t][cc] = (c0 + cc < C) ? ldf<T>(w + (long long)(c0 + cc) * K * K + t) : 0.f;
    }
    for (int i = tid; i < CH; i += NTH) gsm[i] = (c0 + i < C) ? ldf<T>(gamma + c0 + i) : 0.f;
    for (int i = tid; i < NT * CH; i += NTH) (&lacc[0][0])[i] = 0.f;
    const int cg = tid % CG, hf = (tid / CG) % HS, wk = tid / (CG * HS);       // channel vector, half of it, one of 32 work lanes
    const int ch0 = cg * VEC + hf * SV;                           // first of this thread's SV channels within the chunk
    const int ky3 = wk / RG, rg = wk % RG;                        // weight-gradient role: filter row, unit lane
    const bool wactive = ky3 < K && c0 + ch0 < C;
    const int run = wk % (D::TW / R), row = wk / (D::TW / R);     // input-gradient role: 4-pixel run of tile row `row`
    tc_f32x2 acc[K][S2], accb[S2], accg[V2], accbt[V2];
#pragma unroll
    for (int e = 0; e < S2; ++e) {
        accb[e] = tc_f32x2{0.f, 0.f};
#pragma unroll
        for (int kx = 0; kx < K; ++kx) acc[kx][e] = tc_f32x2{0.f, 0.f};
    }
#pragma unroll
    for (int e = 0; e < V2; ++e) accg[e] = accbt[e] = tc_f32x2{0.f, 0.f};
    const int ntiles = B * a.tilesH * a.tilesW;
    const float invC = 1.0f / (float)C;
    // One workgroup per CU: the NEXT tile's vectors of the three maps and its per-pixel LayerNorm quantities are requested before this
    // tile's arithmetic starts and land in registers under it.
    uint4 gr[NX], dr[NX], hr[NX];
    constexpr int PQ = 8;
    float2 pq[PQ], pst_raw = make_float2(0.f, 0.f), ps_extra = make_float2(0.f, 0.f);
    bool pin_next = false;
    auto tile_org = [&](int tidx, int& b, int& oh0, int& ow0) __attribute__((always_inline)) {
        const int tix = tidx % a.tilesW, tiy = (tidx / a.tilesW) % a.tilesH;
        b = tidx / (a.tilesW * a.tilesH); oh0 = tiy * D::TH; ow0 = tix * D::TW;
    };
    auto fetch = [&](int tidx) __attribute__((always_inline)) {
        int b, oh0, ow0;
        tile_org(tidx, b, oh0, ow0);
        const long long ibase = (long long)b * H * W;
        const int iy = tid / IW, ix = tid - iy * IW, ih = oh0 - 1 + iy, iw = ow0 - 1 + ix;
        const bool pin = tid < NPIX && ih >= 0 && ih < H && iw >= 0 && iw < W;
        const long long prow = pin ? ibase + (long long)ih * W + iw : 0;
        const float2* pp = reinterpret_cast<const float2*>(part2) + prow * a.nch2;
        pst_raw = *reinterpret_cast<const float2*>(stat + prow * 2);
        pin_next = pin;
        // the first PQ row-sum partials are requested here and added up when the tile is consumed (a load-add-load loop here would
        // be nch2 dependent memory round trips in front of the tile's own loads); wider LayerNorms finish the sum in the loop below.
        // An even number of partials per row (every width of this model) is read as 16-byte pairs: half the requests.
        float s1 = 0.f, s2 = 0.f;
        if (!(a.nch2 & 1)) {
            const float4* pp4 = reinterpret_cast<const float4*>(pp);
            const int n4 = a.nch2 >> 1;
#pragma unroll
            for (int k = 0; k < PQ / 2; ++k) {
                const float4 v = pp4[k < n4 ? k : 0];
                pq[2 * k] = make_float2(v.x, v.y); pq[2 * k + 1] = make_float2(v.z, v.w);
            }
            for (int k0 = PQ / 2; k0 < n4; k0 += 4) {
                float4 q4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) q4[j] = pp4[k0 + j < n4 ? k0 + j : 0];
#pragma unroll
                for (int j = 0; j < 4; ++j) if (k0 + j < n4) { s1 += q4[j].x + q4[j].z; s2 += q4[j].y + q4[j].w; }
            }
        } else {
#pragma unroll
            for (int k = 0; k < PQ; ++k) pq[k] = pp[k < a.nch2 ? k : 0];
            for (int k0 = PQ; k0 < a.nch2; k0 += 4) {
                float2 q4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) q4[j] = pp[k0 + j < a.nch2 ? k0 + j : 0];
#pragma unroll
                for (int j = 0; j < 4; ++j) if (k0 + j < a.nch2) { s1 += q4[j].x; s2 += q4[j].y; }
            }
        }
        ps_extra = make_float2(s1, s2);
        dw_fetch<T, CG, NX, NTH>(gr, gp + ibase * a.ldg + c0, a.ldg, oh0 - 1, ow0 - 1, IH, IW, H, W, C - c0);
        dw_fetch<T, CG, NX, NTH>(dr, dm + ibase * a.ldd + c0, a.ldd, oh0 - 1, ow0 - 1, IH, IW, H, W, C - c0);
        dw_fetch<T, CG, NX, NTH>(hr, hm + ibase * a.ldh + c0, a.ldh, oh0 - 1, ow0 - 1, IH, IW, H, W, C - c0);
    };
    if (bx < ntiles) fetch(bx);
    __syncthreads();                                              // taps and gamma are in LDS
    MSTAMP(0);
    tc_f32x2 gk[V2];                                              // gamma of this thread's channel vector (its vectors all share cg)
#pragma unroll
    for (int e = 0; e < V2; ++e) gk[e] = tc_f32x2{gsm[cg * VEC + 2 * e], gsm[cg * VEC + 2 * e + 1]};
    for (int tidx = bx; tidx < ntiles; tidx += a.gx) {
        int b, oh0, ow0;
        tile_org(tidx, b, oh0, ow0);
        const long long ibase = (long long)b * H * W;
        __syncthreads();                                          // the previous tile's readers are done with the LDS tiles
        MSTAMP(1);
        if (tid < NPIX) {
            float s1 = ps_extra.x, s2 = ps_extra.y;
#pragma unroll
            for (int k = 0; k < PQ; ++k) if (k < a.nch2) { s1 += pq[k].x; s2 += pq[k].y; }
            pst[tid] = pin_next ? make_float4(pst_raw.x, pst_raw.y, s1 * invC, s2 * invC) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        MSTAMP(2);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int v = tid + i * NTH;
            int op, cgi;
            const bool ok = dw_inside<T, CG>(v, oh0 - 1, ow0 - 1, IH, IW, H, W, C - c0, op, cgi);
            if (v < NPIX * CG) {
                const int pix = v / CG, iy = pix / IW, ix = pix - iy * IW;
                const float4 st = pst[pix];
                tc_f32x2 g8[V2], d8[V2], o8[V2];
                unpack16v<T>(gr[i], g8);
                unpack16v<T>(dr[i], d8);
                const bool inner = ok && iy >= 1 && iy <= D::TH && ix >= 1 && ix <= D::TW;
                const float rs = ok ? st.y : 0.f, wi = inner ? 1.0f : 0.f;
#pragma unroll
                for (int e = 0; e < V2; ++e) {
                    const tc_f32x2 xh = (d8[e] - st.x) * st.y;
                    const tc_f32x2 gg = g8[e] * gk[e];
                    o8[e] = (gg - st.z - xh * st.w) * rs;
                    const tc_f32x2 gi = g8[e] * wi;
                    accg[e] += gi * xh; accbt[e] += gi;
                }
                ddt[pix * PIXQ + cgi] = pack16v<T>(o8);
                ht[pix * PIXQ + cgi] = ok ? hr[i] : make_uint4(0u, 0u, 0u, 0u);
            }
        }
        __syncthreads();
        MSTAMP(3);
        if (tidx + a.gx < ntiles) fetch(tidx + a.gx);
        MSTAMP(4);
        {   // dh = conv^T(dd) + dd for this thread's 4-pixel run and SV channels
            const int oh = oh0 + row, owb = ow0 + run * R, c = c0 + ch0;
            if (c < C && oh < H && owb < W) {
                tc_f32x2 o[R][S2];
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int e = 0; e < S2; ++e) o[r][e] = tc_f32x2{0.f, 0.f};
#pragma unroll
                for (int ky = 0; ky < K; ++ky) {
                    tc_f32x2 in[R + K - 1][S2];
#pragma unroll
                    for (int i = 0; i < R + K - 1; ++i)
                        unpack_sv<T, SV>(ddb + (((row + ky) * IW + run * R + i) * PIXQ + cg) * 16 + hf * SVB, in[i]);
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) {
                        tc_f32x2 wr[S2];
#pragma unroll
                        for (int e = 0; e < S2; ++e) wr[e] = *reinterpret_cast<const tc_f32x2*>(&wsm[ky * K + kx][ch0 + 2 * e]);
#pragma unroll
                        for (int r = 0; r < R; ++r)
#pragma unroll
                            for (int e = 0; e < S2; ++e) o[r][e] += wr[e] * in[r + kx][e];
                    }
                    if (ky == 1) {
#pragma unroll
                        for (int r = 0; r < R; ++r)
#pragma unroll
                            for (int e = 0; e < S2; ++e) o[r][e] += in[r + 1][e];
                    }
                }
                T* dst0 = dh + ((ibase + (long long)oh * W + owb)) * a.lddh + c;
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (owb + r < W) store_sv<T, SV>(dst0 + (long long)r * a.lddh, o[r]);
            }
        }
        MSTAMP(5);
        if (wactive) {   // filter-row ky3 of the weight gradient: sum over the tile of dd[p] * h[p + (ky3 - 1, kx - 1)]
            for (int u = rg; u < UNITS; u += RG) {
                const int urow = u / (D::TW / R), urun = u % (D::TW / R);
                tc_f32x2 d[R][S2], in[R + K - 1][S2];
#pragma unroll
                for (int r = 0; r < R; ++r) unpack_sv<T, SV>(ddb + (((urow + 1) * IW + urun * R + r + 1) * PIXQ + cg) * 16 + hf * SVB, d[r]);
#pragma unroll
                for (int i = 0; i < R + K - 1; ++i) unpack_sv<T, SV>(htb + (((urow + ky3) * IW + urun * R + i) * PIXQ + cg) * 16 + hf * SVB, in[i]);
#pragma unroll
                for (int kx = 0; kx < K; ++kx)
#pragma unroll
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int e = 0; e < S2; ++e) acc[kx][e] += d[r][e] * in[r + kx][e];
                if (ky3 == 0) {
#pragma unroll
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int e = 0; e < S2; ++e) accb[e] += d[r][e];
                }
            }
        }
        MSTAMP(10);
    }
    MSTAMP(6);
    __syncthreads();
    MSTAMP(7);
    // Workgroup sums of the per-thread accumulators, through the (now free) LDS tiles: every thread parks its values
    // ([value][thread], conflict-free), then each output word is summed by ONE thread from its RG (taps, bias) or NTH / CG (dgamma,
    // dbeta) contributors.  The first version added them with LDS atomics: 32 threads per word, 60 atomics per thread -- 31 k of
    // the kernel's 160 k cycles (scripts/exp/mid_timing.py).
    float* scr = reinterpret_cast<float*>(smem);
    static_assert((K * SV + SV) * NTH * 4 <= 2 * NPIX * PIXQ * 16 && 2 * VEC * NTH * 4 <= 2 * NPIX * PIXQ * 16, "scratch fits the dd / h tiles");
#pragma unroll
    for (int kx = 0; kx < K; ++kx)
#pragma unroll
        for (int e = 0; e < S2; ++e) {
            scr[(kx * SV + 2 * e) * NTH + tid] = acc[kx][e].x;
            scr[(kx * SV + 2 * e + 1) * NTH + tid] = acc[kx][e].y;
        }
#pragma unroll
    for (int e = 0; e < S2; ++e) {
        scr[(K * SV + 2 * e) * NTH + tid] = accb[e].x;
        scr[(K * SV + 2 * e + 1) * NTH + tid] = accb[e].y;
    }
    __syncthreads();
    for (int f = tid; f < (K * K + 1) * CH; f += NTH) {           // taps (ky3, kx) and the bias row: RG contributors each
        const int t = f / CH, c = f - t * CH;
        const int ky = t < K * K ? t / K : 0, kx = t < K * K ? t - ky * K : K;
        const int src = (c / VEC) + CG * (((c % VEC) / SV) + HS * (ky * RG));      // thread (cg, hf, wk = ky * RG)
        const float* col = scr + (kx * SV + (c % SV)) * NTH + src;
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < RG; ++g) v += col[g * CG * HS];
        lacc[t][c] = v;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < V2; ++e) {
        scr[(2 * e) * NTH + tid] = accg[e].x;
        scr[(2 * e + 1) * NTH + tid] = accg[e].y;
        scr[(VEC + 2 * e) * NTH + tid] = accbt[e].x;
        scr[(VEC + 2 * e + 1) * NTH + tid] = accbt[e].y;
    }
    __syncthreads();
    for (int f = tid; f < 2 * CH; f += NTH) {                     // dgamma and dbeta: every thread of the channel vector contributes
        const int t = f / CH, c = f - t * CH;
        const float* col = scr + (t * VEC + (c % VEC)) * NTH + (c / VEC);
        float v = 0.f;
        for (int g = 0; g < NTH / CG; ++g) v += col[g * CG];
        lacc[K * K + 1 + t][c] = v;
    }
    __syncthreads();
    MSTAMP(8);
    if (dbg_nofold) return;                                       // timing what-if only (TC_DEBUG_FFN_NOFOLD=1): parameter gradients are dropped
    float* lflat = &lacc[0][0];
    int gm = 1;
    const float* pgroup = nullptr;
    if (a.wsp && !a.wsc) {                                        // deferred (ws_bytes < 0): sums parked for tc_dw_fold, as dw_tile_wgrad_body
        float* part = a.wsp + ((long long)(bz * a.chunks + by) * a.gx + bx) * (NT * CH);
        for (int f = tid; f < NT * CH; f += NTH) part[f] = lflat[f];
        return;
    }
    if (a.wsp) {                                                  // two-level fold, as dw_tile_wgrad_body
        constexpr int FG = DW_FOLD;
        const int chain = bz * a.chunks + by, grp = bx / FG, ngrp = (a.gx + FG - 1) / FG;
        gm = min(FG, a.gx - grp * FG);
        float* part = a.wsp + ((long long)chain * a.gx + bx) * (NT * CH);
        pgroup = a.wsp + ((long long)chain * a.gx + grp * FG) * (NT * CH);
        if (gm > 1) {
            for (int f = tid; f < NT * CH; f += NTH) __hip_atomic_store(part + f, lflat[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            __shared__ int s_last;
            if (tid == 0) {
                int* cn = a.wsc + chain * ngrp + grp;
                const int old = atomicAdd(cn, 1);
                s_last = (old == gm - 1);
                if (s_last) atomicExch(cn, 0);
            }
            __syncthreads();
            if (!s_last) return;
        }
    }
    float* dwp = a.dw + bz * wstride;
    float* dbp = a.db ? a.db + bz * wstride : nullptr;
    float* dgp = a.dgamma ? a.dgamma + bz * wstride : nullptr;
    float* dbtp = a.dbeta ? a.dbeta + bz * wstride : nullptr;
    for (int f = tid; f < NT * CH; f += NTH) {
        const int t = f / CH, cc = f - t * CH, ch = c0 + cc;
        if (ch >= C) continue;
        float v;
        if (gm > 1) {
            float tmp[DW_FOLD];
#pragma unroll
            for (int m = 0; m < DW_FOLD; ++m)
                tmp[m] = m < gm ? __hip_atomic_load(pgroup + (long long)m * (NT * CH) + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
            v = 0.f;
#pragma unroll
            for (int m = 0; m < DW_FOLD; ++m) v += tmp[m];
        } else {
            v = lflat[f];
        }
        if (t < K * K) atomicAdd(dwp + (long long)ch * K * K + t, v);
        else if (t == K * K) { if (dbp) atomicAdd(dbp + ch, v); }
        else if (t == K * K + 1) { if (dgp) atomicAdd(dgp + ch, v); }
        else if (dbtp) atomicAdd(dbtp + ch, v);
    }
    MSTAMP(9);
}

template <typename T, int NTH>
__global__ __launch_bounds__(NTH, 1) void ffn_mid_bwd_kernel(FfnMultiDev q) {
    extern __shared__ uint4 dsm[];
    int lin = blockIdx.x, si = 0;
    if (q.n > 1 && lin >= q.s[1].blk0) si = 1;
    if (q.n > 2 && lin >= q.s[2].blk0) si = 2;
    if (q.n > 3 && lin >= q.s[3].blk0) si = 3;
    const FfnSegDev& g = q.s[si];
    lin -= g.blk0;
    ffn_mid_bwd_body<T, NTH>(g, q.wstride, lin % g.gx, lin / g.gx, blockIdx.y, dsm, q.dbg_nofold);
}

// Tile walkers of segment i of a tc_ffn_mid_bwd launch: ~tc_mid_wg_target() workgroups in total, shared out by tiles x chunks.  ONE statement
// for launch_ffn_mid_bwd and tc_ffn_mid_plan (the deferred sums are laid out by it).
template <typename T> long long ffn_mid_work(const TcFfnSeg* segs, int nseg) {
    using D = FfnTile<T>;
    long long total_work = 0;
    for (int i = 0; i < nseg; ++i)
        total_work += (long long)segs[i].B * ((segs[i].W + 15) / 16) * ((segs[i].H + D::TH - 1) / D::TH) * ((segs[i].C + D::CH - 1) / D::CH);
    return total_work;
}
inline long long ffn_mid_gx(long long ntiles, long long total_work, int groups) {
    long long gx = (long long)((double)tc_mid_wg_target() * (double)ntiles / (double)(total_work > 0 ? total_work : 1) / groups + 0.5);
    return gx < 1 ? 1 : (gx > ntiles ? ntiles : gx);
}

template <typename T>
int launch_ffn_mid_bwd(const TcFfnSeg* segs, int nseg, int groups, long long wstride, void* ws, long long ws_bytes, hipStream_t s) {
    using D = FfnTile<T>;
    static_assert(D::smem_q * 16 <= 64 * 1024, "static dynamic-LDS limit");
    static const int nth = getenv("TC_FFN_MID_THREADS") ? atoi(getenv("TC_FFN_MID_THREADS")) : 512;   // A/B switch: 256 = one wave per SIMD (14.18 vs 13.95 ms per step)
    static const int nofold = getenv("TC_DEBUG_FFN_NOFOLD") ? atoi(getenv("TC_DEBUG_FFN_NOFOLD")) : 0;
    FfnMultiDev q;
    q.n = nseg; q.wstride = wstride; q.dbg_nofold = nofold;
    long long blk = 0, part_floats = 0, cnts = 0;
    const bool defer = ws && ws_bytes < 0;                       // the walkers' sums go to the caller's own buffer (tc_dw_fold adds them later)
    if (defer && (uintptr_t)ws % 16) return TC_ERR_ARG;
    const bool have_ws = !defer && ws && (uintptr_t)ws % 16 == 0 && ws_bytes > 16384;
    const long long total_work = ffn_mid_work<T>(segs, nseg);
    for (int i = 0; i < nseg; ++i) {
        const TcFfnSeg& g = segs[i];
        FfnSegDev& d = q.s[i];
        if (!g.gp || !g.d || !g.h || !g.dh || !g.stat || !g.part2 || !g.w || !g.gamma || !g.dw || g.C <= 0 || g.C % D::VEC || g.nch2 < 1 ||
            !dw_tile_ok<T>(g.gp, g.ldg, g.d, g.ldd, g.C) || !dw_tile_ok<T>(g.h, g.ldh, g.dh, g.lddh, g.C) ||
            (!(g.nch2 & 1) && (uintptr_t)g.part2 % 16))              // even partial counts are read as 16-byte pairs
            return TC_ERR_ARG;
        d.gp = g.gp; d.d = g.d; d.h = g.h; d.dh = g.dh; d.stat = g.stat; d.part2 = g.part2; d.w = g.w; d.gamma = g.gamma;
        d.dw = g.dw; d.db = g.db; d.dgamma = g.dgamma; d.dbeta = g.dbeta;
        d.C = g.C; d.ldg = g.ldg; d.ldd = g.ldd; d.ldh = g.ldh; d.lddh = g.lddh; d.B = g.B; d.H = g.H; d.W = g.W; d.nch2 = g.nch2;
        d.chunks = (g.C + D::CH - 1) / D::CH;
        d.tilesW = (g.W + 15) / 16; d.tilesH = (g.H + D::TH - 1) / D::TH;
        const long long ntiles = (long long)g.B * d.tilesW * d.tilesH;
        // ~256 workgroups in total (one per CU), shared out by tiles x chunks; every channel chunk gets gx tile walkers
        const long long gx = ffn_mid_gx(ntiles, total_work, groups);
        d.gx = (int)gx;
        d.wsc = have_ws ? reinterpret_cast<int*>(ws) + cnts : nullptr;
        d.wsp = have_ws ? reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + 16384) + part_floats : nullptr;
        if (defer) d.wsp = reinterpret_cast<float*>(ws) + part_floats;
        cnts += (long long)d.chunks * groups * ((gx + DW_FOLD - 1) / DW_FOLD);
        part_floats += (long long)d.chunks * groups * gx * D::NT * D::CH;
        d.blk0 = (int)blk;
        blk += gx * d.chunks;
    }
    if (blk > 0x7fffffffLL) return TC_ERR_ARG;
    if (defer && part_floats * 4 > -ws_bytes) return TC_ERR_ARG;
    if (have_ws && (cnts > 4096 || 16384 + part_floats * 4 > ws_bytes))
        for (int i = 0; i < nseg; ++i) { q.s[i].wsc = nullptr; q.s[i].wsp = nullptr; }
    const size_t smem = (size_t)D::smem_q * 16;
    if (nth == 256) hipLaunchKernelGGL((ffn_mid_bwd_kernel<T, 256>), dim3((unsigned)blk, groups), dim3(256), smem, s, q);
    else hipLaunchKernelGGL((ffn_mid_bwd_kernel<T, 512>), dim3((unsigned)blk, groups), dim3(512), smem, s, q);
    return tc_launch_status();
}

bool dw_args_ok(int B, int H, int W, int C, int k, int stride, int add_input) {
    return B > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0 && (k == 3 || k == 5 || k == 7) && (stride == 1 || stride == 2) &&
           !(add_input && stride != 1);
}

}  // namespace

extern "C" int tc_dwconv_fwd(const void* x, int ldx, const void* w, const void* bias, void* y, int ldy, int B, int H, int W,
                             int C, int k, int stride, int add_input, int groups, long long wstride, int dtype, void* stream) {
    if (!x || !w || !y || (ldx & 3) || (ldy & 3) || groups < 1 || (groups > 1 && stride != 1) || !dw_args_ok(B, H, W, C, k, stride, add_input))
        return TC_ERR_ARG;
    if (stride == 1)
        TC_DISPATCH_DTYPE(dtype, {
            if (dw_tile_ok<T>(x, ldx, y, ldy, C))
                return (launch_tile<T, 0>(x, ldx, w, bias, y, ldy, nullptr, 0, nullptr, nullptr, B, H, W, C, k, add_input, 0, groups,
                                          wstride, (hipStream_t)stream));
            return (launch_strip<T, 0>(x, ldx, w, bias, nullptr, 0, y, ldy, nullptr, nullptr, B, H, W, C, k, add_input, 0, groups,
                                       wstride, (hipStream_t)stream));
        });
    TC_DISPATCH_DTYPE(dtype, return (launch_dw<T, false>(x, ldx, w, bias, y, ldy, B, H, W, C, k, stride, add_input, 0,
                                                         (hipStream_t)stream)));
    return TC_ERR_ARG;
}

extern "C" int tc_dwconv_bwd_input(const void* dy, int lddy, const void* w, void* dx, int lddx, int B, int H, int W, int C,
                                   int k, int stride, int add_input, int accumulate, int groups, long long wstride, int dtype,
                                   void* stream) {
    if (!dy || !w || !dx || (lddy & 3) || (lddx & 3) || groups < 1 || (groups > 1 && stride != 1) ||
        !dw_args_ok(B, H, W, C, k, stride, add_input))
        return TC_ERR_ARG;
    if (stride == 1)
        TC_DISPATCH_DTYPE(dtype, {
            if (dw_tile_ok<T>(dy, lddy, dx, lddx, C))
                return (launch_tile<T, 1>(dy, lddy, w, nullptr, dx, lddx, nullptr, 0, nullptr, nullptr, B, H, W, C, k, add_input,
                                          accumulate, groups, wstride, (hipStream_t)stream));
            return (launch_strip<T, 1>(nullptr, 0, w, nullptr, dy, lddy, dx, lddx, nullptr, nullptr, B, H, W, C, k, add_input,
                                       accumulate, groups, wstride, (hipStream_t)stream));
        });
    TC_DISPATCH_DTYPE(dtype, return (launch_dw<T, true>(dy, lddy, w, nullptr, dx, lddx, B, H, W, C, k, stride, add_input, accumulate,
                                                        (hipStream_t)stream)));
    return TC_ERR_ARG;
}

extern "C" int tc_dwconv_bwd_weight(const void* dy, int lddy, const void* x, int ldx, float* dw, float* db, int B, int H, int W, int C,
                                    int k, int stride, int groups, long long wstride, void* ws, long long ws_bytes, int dtype, void* stream);
/* see include/transception_hip.h */
// ---- deferred weight-gradient fold of tc_dwconv_bwd (ws_bytes < 0) --------------------------------------------------------------------
namespace {
constexpr int DWF_SITES = 48;                                  // (48 sites x 80 bytes: the argument block stays below 4 KiB)
struct DwFoldSite { const float* part; float* dw; float* db; float* dgamma; float* dbeta; long long wstride; int C, kk, ch, chunks, gx, nt, groups, blk0; };
constexpr int DWF_WALKERS = 32;                                 // walkers per fold workgroup: one batch of eight loads per thread
struct DwFoldDev { DwFoldSite s[DWF_SITES]; int n; };
static_assert(sizeof(DwFoldDev) <= 4096, "kernel argument block");
// workgroup = 64 consecutive (tap, channel) words of one (site, group, channel chunk) x 4 slices of its walkers (a first version gave a whole
// chain to one workgroup, every thread adding all walkers of its words: ~300 workgroups and 33 us per launch)
__global__ __launch_bounds__(256) void dw_fold_kernel(const DwFoldDev q) {
    __shared__ float red[4][64];
    int si = 0;
    for (int i = 1; i < q.n; ++i) if ((int)blockIdx.x >= q.s[i].blk0) si = i;
    const DwFoldSite& t = q.s[si];
    const int ntc = t.nt * t.ch, per = (ntc + 63) / 64, wsplit = (t.gx + DWF_WALKERS - 1) / DWF_WALKERS;
    int lin = blockIdx.x - t.blk0;
    const int ws = lin % wsplit; lin /= wsplit;
    const int chain = lin / per, f = (lin - chain * per) * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
    const int g = chain / t.chunks, c0 = (chain - g * t.chunks) * t.ch;
    const int mend = min(t.gx, (ws + 1) * DWF_WALKERS);
    float v = 0.f;
    if (f < ntc) {
        const float* p = t.part + (long long)chain * t.gx * ntc + f;
        int m = ws * DWF_WALKERS + sl;
        for (; m + 28 < mend; m += 32) {
            float tmp[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) tmp[e] = p[(long long)(m + 4 * e) * ntc];
#pragma unroll
            for (int e = 0; e < 8; ++e) v += tmp[e];
        }
        for (; m < mend; m += 4) v += p[(long long)m * ntc];
    }
    red[sl][threadIdx.x & 63] = v;
    __syncthreads();
    if (sl != 0 || f >= ntc) return;
    v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    const int tap = f / t.ch, ch = c0 + (f - tap * t.ch);
    if (ch >= t.C) return;
    if (tap < t.kk) atomicAdd(t.dw + g * t.wstride + (long long)ch * t.kk + tap, v);      // (shared modules: a weight may have two writers)
    else if (tap == t.kk) { if (t.db) atomicAdd(t.db + g * t.wstride + ch, v); }
    else if (tap == t.kk + 1) { if (t.dgamma) atomicAdd(t.dgamma + g * t.wstride + ch, v); }   // tc_ffn_mid_bwd: the LayerNorm behind the convolution
    else if (t.dbeta) atomicAdd(t.dbeta + g * t.wstride + ch, v);
}
}  // namespace

extern "C" long long tc_dwconv_bwd_plan(int B, int H, int W, int C, int k, int groups, int dtype, TcDwFold* site) {
    if (!site || groups < 1 || !dw_args_ok(B, H, W, C, k, 1, 0)) return 0;
    TC_DISPATCH_DTYPE(dtype, {
        constexpr int VEC = Vec16<T>::N;
        if (C % VEC) return 0;
        const DwGeom g = dw_bwd_geom<T>(B, H, W, C, k, groups);
        site->C = C; site->k = k; site->groups = groups; site->ch = g.ch; site->chunks = g.chunks; site->gx = g.gx; site->nt = g.nt;
        return (long long)g.chunks * groups * g.gx * g.nt * g.ch;
    });
    return 0;
}

// the same for the weight-gradient walkers of a tc_dwconv_multi launch (mode 2 or 3): sites[i] / offs[i] (floats into the one buffer) per segment
extern "C" long long tc_dwconv_multi_plan(const TcDwSeg* segs, int nseg, int groups, int dtype, TcDwFold* sites, long long* offs) {
    if (!segs || !sites || !offs || nseg < 1 || nseg > DW_MULTI_MAX || groups < 1) return 0;
    long long total = 0;
    TC_DISPATCH_DTYPE(dtype, {
        constexpr int VEC = Vec16<T>::N;
        for (int i = 0; i < nseg; ++i)
            if (segs[i].C % VEC || !dw_args_ok(segs[i].B, segs[i].H, segs[i].W, segs[i].C, segs[i].k, 1, 0)) return 0;
        const long long total_work = dw_multi_work<T>(segs, nseg);
        for (int i = 0; i < nseg; ++i) {                         // launch_multi's numbers
            const TcDwSeg& g = segs[i];
            const int cg = dw_pick_cg<T>(g.C), ch = cg * VEC, chunks = (g.C + ch - 1) / ch, TH = (256 / cg) / 4;
            const long long ntiles = (long long)g.B * ((g.W + 15) / 16) * ((g.H + TH - 1) / TH);
            const long long gx = dw_multi_gx(ntiles, g.k, total_work, groups);
            sites[i].C = g.C; sites[i].k = g.k; sites[i].groups = groups; sites[i].ch = ch; sites[i].chunks = chunks; sites[i].gx = (int)gx;
            sites[i].nt = g.k * g.k + 1;
            offs[i] = total;
            total += (long long)chunks * groups * gx * (g.k * g.k + 1) * ch;
        }
        return total;
    });
    return 0;
}

extern "C" int tc_dw_fold(const TcDwFold* sites, int n, void* stream) {
    if (!sites || n < 1 || n > DWF_SITES) return TC_ERR_ARG;
    DwFoldDev q;
    q.n = n;
    int blk = 0;
    for (int i = 0; i < n; ++i) {
        const TcDwFold& t = sites[i];
        if (!t.part || !t.dw || t.groups < 1 || t.C < 1 || (t.k != 3 && t.k != 5 && t.k != 7)) return TC_ERR_ARG;
        if (t.gx < 1 || t.ch < 1 || t.chunks != (t.C + t.ch - 1) / t.ch) return TC_ERR_ARG;
        if (t.nt != t.k * t.k + 1 && t.nt != t.k * t.k + 3) return TC_ERR_ARG;
        q.s[i] = DwFoldSite{t.part, t.dw, t.db, t.dgamma, t.dbeta, t.wstride, t.C, t.k * t.k, t.ch, t.chunks, t.gx, t.nt, t.groups, blk};
        blk += t.chunks * t.groups * ((t.nt * t.ch + 63) / 64) * ((t.gx + DWF_WALKERS - 1) / DWF_WALKERS);
    }
    hipLaunchKernelGGL(dw_fold_kernel, dim3(blk), dim3(256), 0, (hipStream_t)stream, q);
    return tc_launch_status();
}

extern "C" int tc_dwconv_bwd(const void* dy, int lddy, const void* x, int ldx, const void* w, void* dx, int lddx, float* dw, float* db, int B, int H,
                             int W, int C, int k, int add_input, int accumulate, int groups, long long wstride, void* ws, long long ws_bytes,
                             int dtype, void* stream) {
    if (!dy || !x || !w || !dx || !dw || (lddy & 3) || (lddx & 3) || groups < 1 || !dw_args_ok(B, H, W, C, k, 1, add_input)) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, {
        if (dw_tile_ok<T>(dy, lddy, dx, lddx, C) && dw_tile_ok<T>(x, ldx, dy, lddy, C))
            return (launch_tile_bwd<T>(dy, lddy, w, dx, lddx, x, ldx, dw, db, B, H, W, C, k, add_input, accumulate, groups, wstride, (hipStream_t)stream,
                                       ws, ws_bytes));
    });
    if (ws_bytes < 0) return TC_ERR_ARG;                        // deferred sums exist only in the tile kernels (16-byte addressable operands)
    const int rc = tc_dwconv_bwd_input(dy, lddy, w, dx, lddx, B, H, W, C, k, 1, add_input, accumulate, groups, wstride, dtype, stream);
    return rc != TC_OK ? rc : tc_dwconv_bwd_weight(dy, lddy, x, ldx, dw, db, B, H, W, C, k, 1, groups, wstride, ws, ws_bytes, dtype, stream);
}

extern "C" int tc_dwconv_bwd_weight(const void* dy, int lddy, const void* x, int ldx, float* dw, float* db, int B, int H,
                                    int W, int C, int k, int stride, int groups, long long wstride, void* ws, long long ws_bytes,
                                    int dtype, void* stream) {
    if (!dy || !x || !dw || groups < 1 || (groups > 1 && stride != 1) || !dw_args_ok(B, H, W, C, k, stride, 0)) return TC_ERR_ARG;
    if (stride == 1)
        TC_DISPATCH_DTYPE(dtype, {
            if (dw_tile_ok<T>(x, ldx, dy, lddy, C))
                return (launch_tile<T, 2>(x, ldx, nullptr, nullptr, nullptr, 0, dy, lddy, dw, db, B, H, W, C, k, 0, 0, groups, wstride,
                                          (hipStream_t)stream, ws, ws_bytes));
            return (launch_strip<T, 2>(x, ldx, nullptr, nullptr, dy, lddy, nullptr, 0, dw, db, B, H, W, C, k, 0, 0, groups, wstride,
                                       (hipStream_t)stream));
        });
    const int P = (k - 1) / 2;
    const int Ho = (H + 2 * P - k) / stride + 1, Wo = (W + 2 * P - k) / stride + 1;
    const long long npix = (long long)B * Ho * Wo;
    dim3 grid(tc_blocks(npix, 4 * 16, 256), (C + 63) / 64), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (k == 3 && (ldx & 3) == 0 && (lddy & 3) == 0 && npix < 0x7fffffffLL) {
        static const int wg3 = getenv("TC_DW_WGRAD3_WG") ? atoi(getenv("TC_DW_WGRAD3_WG")) : 64;
        const dim3 g3(tc_blocks(npix, 16 * 8, wg3), (C + 63) / 64);
        TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dw_wgrad3_kernel<T>), g3, block, 0, s, (const T*)dy, lddy, (const T*)x, ldx, dw, db, B, H, W,
                                                    Ho, Wo, C, stride));
        return tc_launch_status();
    }
#define TC_DWW(KK) hipLaunchKernelGGL((dw_wgrad_kernel<T, KK>), grid, block, 0, s, (const T*)dy, lddy, (const T*)x, ldx, dw, db, \
                                      B, H, W, Ho, Wo, C, stride)
    TC_DISPATCH_DTYPE(dtype, { if (k == 3) TC_DWW(3); else if (k == 5) TC_DWW(5); else TC_DWW(7); });
#undef TC_DWW
    return tc_launch_status();
}

extern "C" int tc_dwconv_multi(const TcDwSeg* segs, int nseg, int mode, int add_input, int accumulate, int groups, long long wstride,
                               void* ws, long long ws_bytes, int dtype, void* stream) {
    if (!segs || nseg < 1 || nseg > DW_MULTI_MAX || mode < 0 || mode > 3 || groups < 1) return TC_ERR_ARG;
    bool tile_ok = true;
    for (int i = 0; i < nseg; ++i) {
        const TcDwSeg& g = segs[i];
        if (!g.x || !dw_args_ok(g.B, g.H, g.W, g.C, g.k, 1, add_input) || (mode != 2 && (!g.w || !g.y)) || (mode >= 2 && (!g.dy || !g.dw)))
            return TC_ERR_ARG;
        const void* second = mode >= 2 ? g.dy : g.y;
        const int ld2 = mode >= 2 ? g.lddy : g.ldy;
        if (dtype == TC_F32) tile_ok = tile_ok && dw_tile_ok<float>(g.x, g.ldx, second, ld2, g.C);
        else tile_ok = tile_ok && dw_tile_ok<bf16_t>(g.x, g.ldx, second, ld2, g.C);           // (either 16-bit type)
        if (mode == 3) {                                       // mode 3 also writes y = dx from dy
            if (dtype == TC_F32) tile_ok = tile_ok && dw_tile_ok<float>(g.dy, g.lddy, g.y, g.ldy, g.C);
            else tile_ok = tile_ok && dw_tile_ok<bf16_t>(g.dy, g.lddy, g.y, g.ldy, g.C);
        }
    }
    if (mode == 3 && !tile_ok) {                               // both gradients, one segment at a time through the single entry
        for (int i = 0; i < nseg; ++i) {
            const TcDwSeg& g = segs[i];
            const int rc = tc_dwconv_bwd(g.dy, g.lddy, g.x, g.ldx, g.w, g.y, g.ldy, g.dw, g.db, g.B, g.H, g.W, g.C, g.k, add_input, accumulate, groups,
                                         wstride, ws, ws_bytes, dtype, stream);
            if (rc != TC_OK) return rc;
        }
        return TC_OK;
    }
    if (tile_ok) TC_DISPATCH_DTYPE(dtype, return (launch_multi<T>(segs, nseg, mode, add_input, accumulate, groups, wstride, ws, ws_bytes,
                                                                  (hipStream_t)stream)));
    for (int i = 0; i < nseg; ++i) if (mode == 0 && segs[i].stat) return TC_ERR_ARG;      // statistics exist on the tile path only
    for (int i = 0; i < nseg; ++i) {                          // unaligned views: one launch per segment through the single entries
        const TcDwSeg& g = segs[i];
        int rc;
        if (mode == 0) rc = tc_dwconv_fwd(g.x, g.ldx, g.w, g.bias, g.y, g.ldy, g.B, g.H, g.W, g.C, g.k, 1, add_input, groups, wstride, dtype, stream);
        else if (mode == 1) rc = tc_dwconv_bwd_input(g.x, g.ldx, g.w, g.y, g.ldy, g.B, g.H, g.W, g.C, g.k, 1, add_input, accumulate, groups, wstride, dtype, stream);
        else rc = tc_dwconv_bwd_weight(g.dy, g.lddy, g.x, g.ldx, g.dw, g.db, g.B, g.H, g.W, g.C, g.k, 1, groups, wstride, ws, ws_bytes, dtype, stream);
        if (rc != TC_OK) return rc;
    }
    return TC_OK;
}

/* see include/transception_hip.h */
extern "C" int tc_ffn_chunk(int C, int dtype) {
    if (C <= 0) return TC_ERR_ARG;
    if (dtype == TC_F32) return dw_pick_cg<float>(C) * Vec16<float>::N;
    if (dtype == TC_BF16 || dtype == TC_F16) return dw_pick_cg<bf16_t>(C) * Vec16<bf16_t>::N;
    return TC_ERR_ARG;
}

extern "C" int tc_ffn_dw_fwd(const void* x, int ldx, const void* w, const void* bias, void* y, int ldy, float* stat, int B, int H, int W,
                             int C, int groups, long long wstride, int dtype, void* stream) {
    if (!x || !w || !y || (ldx & 3) || (ldy & 3) || groups < 1 || !dw_args_ok(B, H, W, C, 3, 1, 1)) return TC_ERR_ARG;     // stat may be NULL
    TC_DISPATCH_DTYPE(dtype, {
        if (!dw_tile_ok<T>(x, ldx, y, ldy, C)) return TC_ERR_ARG;
        return (launch_tile<T, 0>(x, ldx, w, bias, y, ldy, nullptr, 0, nullptr, nullptr, B, H, W, C, 3, 1, 0, groups, wstride,
                                  (hipStream_t)stream, nullptr, 0, stat));
    });
    return TC_ERR_ARG;
}

#ifdef TC_MID_TIMING
extern "C" int tc_mid_dbg_read(long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_mid_dbg), sizeof(long long) * 64 * 16); }
#endif
// the same for tc_ffn_mid_bwd (ws_bytes < 0): per segment the walkers' geometry (taps: 9 + conv bias + dgamma + dbeta) and its offset
extern "C" long long tc_ffn_mid_plan(const TcFfnSeg* segs, int nseg, int groups, int dtype, TcDwFold* sites, long long* offs) {
    if (!segs || !sites || !offs || nseg < 1 || nseg > FFN_MULTI_MAX || groups < 1) return 0;
    TC_DISPATCH_DTYPE(dtype, {
        using D = FfnTile<T>;
        long long total = 0;
        const long long total_work = ffn_mid_work<T>(segs, nseg);
        for (int i = 0; i < nseg; ++i) {                         // launch_ffn_mid_bwd's numbers
            const TcFfnSeg& g = segs[i];
            if (g.C <= 0 || g.C % D::VEC) return 0;
            const int chunks = (g.C + D::CH - 1) / D::CH;
            const long long ntiles = (long long)g.B * ((g.W + 15) / 16) * ((g.H + D::TH - 1) / D::TH);
            const long long gx = ffn_mid_gx(ntiles, total_work, groups);
            sites[i].C = g.C; sites[i].k = 3; sites[i].groups = groups; sites[i].ch = D::CH; sites[i].chunks = chunks; sites[i].gx = (int)gx;
            sites[i].nt = D::NT;
            offs[i] = total;
            total += (long long)chunks * groups * gx * D::NT * D::CH;
        }
        return total;
    });
    return 0;
}

extern "C" int tc_ffn_mid_bwd(const TcFfnSeg* segs, int nseg, int groups, long long wstride, void* ws, long long ws_bytes, int dtype,
                              void* stream) {
    if (!segs || nseg < 1 || nseg > FFN_MULTI_MAX || groups < 1) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, return (launch_ffn_mid_bwd<T>(segs, nseg, groups, wstride, ws, ws_bytes, (hipStream_t)stream)));
    return TC_ERR_ARG;
}
