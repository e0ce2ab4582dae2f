// CBAMBlock pieces (the concat = "cbam" aggregate of MHCA_stage, MSTr.py:1128-1211, 1400-1401) that the rest of the library does not have:
//   ChannelAttention: global max AND average pooling of the token map per image and channel (:1141-1142)
//   SpatialAttention: per-token maximum and mean over the channels (:1156-1158), Conv2d(2 -> 1, k x k, padding k/2) + sigmoid (:1151,1161-1163)
//   the per-token gate out = x * sa (:1206)
// Maps are [B * N, C] token rows (row stride ld), storage dtype T; small index arrays are int32.  All of it is memory-bound glue on maps of
// a few MB; the kernels follow the pooling / gating kernels of elementwise.hip (16 channel quads x 16 row lanes per workgroup).
#include "tc_common.h"

namespace {

#define TC_S ((hipStream_t)stream)

// pooled rows 0..B-1 = max over the image's N rows (first maximal row in idx), rows B..2B-1 = mean
template <typename T>
__global__ __launch_bounds__(256) void chan_pool2_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ pooled, int* __restrict__ idx, int B, int N, int C) {
    __shared__ float4 rs[16][16], rm[16][16];
    __shared__ int ri[16][16][4];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4, c = blockIdx.y * 64 + tx * 4, b = blockIdx.x;
    float s[4] = {0.f, 0.f, 0.f, 0.f}, m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int mi[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    if (c < C)
        for (int r = ty; r < N; r += 16) {
            const float4 v = ld4<T>(x + ((long long)b * N + r) * ldx + c);
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[e] += vv[e]; if (vv[e] > m[e]) { m[e] = vv[e]; mi[e] = r; } }
        }
    rs[ty][tx] = make_float4(s[0], s[1], s[2], s[3]);
    rm[ty][tx] = make_float4(m[0], m[1], m[2], m[3]);
#pragma unroll
    for (int e = 0; e < 4; ++e) ri[ty][tx][e] = mi[e];
    __syncthreads();
    if (ty == 0 && c < C) {
        float ts[4] = {0.f, 0.f, 0.f, 0.f}, tm[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int ti[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
        for (int k = 0; k < 16; ++k) {
            const float4 a = rs[k][tx], mm = rm[k][tx];
            const float av[4] = {a.x, a.y, a.z, a.w}, mv[4] = {mm.x, mm.y, mm.z, mm.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ts[e] += av[e];
                const int i2 = ri[k][tx][e];
                if (mv[e] > tm[e] || (mv[e] == tm[e] && i2 < ti[e])) { tm[e] = mv[e]; ti[e] = i2; }      // the FIRST maximal row, as torch
            }
        }
        const float inv = 1.f / (float)N;
        st4<T>(pooled + (long long)b * C + c, make_float4(tm[0], tm[1], tm[2], tm[3]));
        st4<T>(pooled + ((long long)B + b) * C + c, make_float4(ts[0] * inv, ts[1] * inv, ts[2] * inv, ts[3] * inv));
#pragma unroll
        for (int e = 0; e < 4; ++e) idx[(long long)b * C + c + e] = ti[e];
    }
}

// dx[b, n, c] (+)= (n == idx[b, c] ? dpooled[b, c] : 0) + dpooled[B + b, c] / N
template <typename T>
__global__ void chan_pool2_bwd_kernel(const T* dp, const int* idx, T* dx, int lddx, int B, int N, int C, int acc) {
    const int cq = C >> 2;
    const float inv = 1.f / (float)N;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)B * N * cq; i += gridDim.x * blockDim.x) {
        const int q = (int)(i % cq) * 4; const unsigned row = i / cq; const int b = (int)(row / (unsigned)N), n = (int)(row - (unsigned)b * N);
        const float4 gm = ld4<T>(dp + (long long)b * C + q), ga = ld4<T>(dp + ((long long)B + b) * C + q);
        const int4 ix = *reinterpret_cast<const int4*>(idx + (long long)b * C + q);
        float4 o = make_float4(ga.x * inv + (ix.x == n ? gm.x : 0.f), ga.y * inv + (ix.y == n ? gm.y : 0.f), ga.z * inv + (ix.z == n ? gm.z : 0.f),
                               ga.w * inv + (ix.w == n ? gm.w : 0.f));
        T* d = dx + (long long)row * lddx + q;
        if (acc) { const float4 v = ld4<T>(d); o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w; }
        st4<T>(d, o);
    }
}

// per token row: st[row] = (max over the C channels, mean over them), idx[row] = the first maximal channel.  16 lanes per row.
template <typename T>
__global__ __launch_bounds__(256) void pix_stats_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ st, int* __restrict__ idx, int rows, int C) {
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    for (int r = blockIdx.x * 16 + ty; r < rows; r += gridDim.x * 16) {
        float s = 0.f, m = -INFINITY;
        int mi = 0x7fffffff;
        for (int c = tx * 4; c < C; c += 64) {
            const float4 v = ld4<T>(x + (long long)r * ldx + c);
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { s += vv[e]; if (vv[e] > m) { m = vv[e]; mi = c + e; } }
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            s += __shfl_xor(s, o, 64);
            const float m2 = __shfl_xor(m, o, 64);
            const int i2 = __shfl_xor(mi, o, 64);
            if (m2 > m || (m2 == m && i2 < mi)) { m = m2; mi = i2; }
        }
        if (tx == 0) { stf<T>(st + (long long)r * 2, m); stf<T>(st + (long long)r * 2 + 1, s / (float)C); idx[r] = mi; }
    }
}

// dx[r, c] (+)= dst[r, 1] / C + (c == idx[r] ? dst[r, 0] : 0)
template <typename T>
__global__ void pix_stats_bwd_kernel(const T* dst, const int* idx, T* dx, int lddx, int rows, int C, int acc) {
    const int cq = C >> 2;
    const float inv = 1.f / (float)C;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)rows * cq; i += gridDim.x * blockDim.x) {
        const int q = (int)(i % cq) * 4; const unsigned r = i / cq;
        const float gm = ldf<T>(dst + (long long)r * 2), ga = ldf<T>(dst + (long long)r * 2 + 1) * inv;
        const int ix = idx[r];
        float4 o = make_float4(ga + (ix == q ? gm : 0.f), ga + (ix == q + 1 ? gm : 0.f), ga + (ix == q + 2 ? gm : 0.f), ga + (ix == q + 3 ? gm : 0.f));
        T* d = dx + (long long)r * lddx + q;
        if (acc) { const float4 v = ld4<T>(d); o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w; }
        st4<T>(d, o);
    }
}

// g[b, h, w] = sigmoid(bias + sum_{ch, ky, kx} w[ch][ky][kx] st[b, h + ky - P, w + kx - P, ch])   (zero padding), one thread per token
template <typename T, int K>
__global__ __launch_bounds__(256) void sa_conv_fwd_kernel(const T* __restrict__ st, const T* __restrict__ w, const T* __restrict__ bias, T* __restrict__ g,
                                                          int B, int H, int W) {
    __shared__ float ws[2 * K * K + 1];
    for (int i = threadIdx.x; i < 2 * K * K; i += 256) ws[i] = ldf<T>(w + i);
    if (threadIdx.x == 0) ws[2 * K * K] = ldf<T>(bias);
    __syncthreads();
    constexpr int P = K / 2;
    const unsigned n = (unsigned)B * H * W;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int x0 = (int)(i % (unsigned)W); const unsigned t = i / (unsigned)W; const int y0 = (int)(t % (unsigned)H), b = (int)(t / (unsigned)H);
        float acc = ws[2 * K * K];
        for (int ky = 0; ky < K; ++ky) {
            const int yy = y0 + ky - P;
            if (yy < 0 || yy >= H) continue;
            for (int kx = 0; kx < K; ++kx) {
                const int xx = x0 + kx - P;
                if (xx < 0 || xx >= W) continue;
                const T* p = st + (((long long)b * H + yy) * W + xx) * 2;
                acc += ws[ky * K + kx] * ldf<T>(p) + ws[K * K + ky * K + kx] * ldf<T>(p + 1);
            }
        }
        stf<T>(g + i, sigmoid_f(acc));
    }
}

// dz = dg * g (1 - g);  dst[b, y, x, ch] = sum_taps w[ch][ky][kx] dz[b, y - ky + P, x - kx + P];  dw[ch][ky][kx] += sum dz[p] st[p + tap];  db += sum dz
template <typename T, int K>
__global__ __launch_bounds__(256) void sa_conv_bwd_kernel(const T* __restrict__ dg, const T* __restrict__ g, const T* __restrict__ st, const T* __restrict__ w,
                                                          T* __restrict__ dst, float* __restrict__ dw, float* __restrict__ db, int B, int H, int W) {
    __shared__ float ws[2 * K * K];
    __shared__ float red[2 * K * K + 1];
    for (int i = threadIdx.x; i < 2 * K * K; i += 256) ws[i] = ldf<T>(w + i);
    for (int i = threadIdx.x; i < 2 * K * K + 1; i += 256) red[i] = 0.f;
    __syncthreads();
    constexpr int P = K / 2;
    const unsigned n = (unsigned)B * H * W;
    auto dz_at = [&](int b, int yy, int xx) -> float {
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) return 0.f;
        const long long p = ((long long)b * H + yy) * W + xx;
        const float s = ldf<T>(g + p);
        return ldf<T>(dg + p) * s * (1.f - s);
    };
    for (unsigned i0 = blockIdx.x * 256; i0 < n; i0 += gridDim.x * 256) {
        const unsigned i = i0 + threadIdx.x;
        const bool in = i < n;
        const int x0 = (int)(i % (unsigned)W); const unsigned t = i / (unsigned)W; const int y0 = (int)(t % (unsigned)H), b = (int)(t / (unsigned)H);
        float a0 = 0.f, a1 = 0.f;
        const float dz = in ? dz_at(b, y0, x0) : 0.f;
        for (int ky = 0; ky < K; ++ky)
            for (int kx = 0; kx < K; ++kx) {
                if (in) {                                           // input gradient: the transposed convolution
                    const float d2 = dz_at(b, y0 - ky + P, x0 - kx + P);
                    a0 += ws[ky * K + kx] * d2; a1 += ws[K * K + ky * K + kx] * d2;
                }
                // weight gradient of this tap: dz[p] * st[p + (ky - P, kx - P)], summed over the workgroup's tokens
                float w0 = 0.f, w1 = 0.f;
                const int yy = y0 + ky - P, xx = x0 + kx - P;
                if (in && yy >= 0 && yy < H && xx >= 0 && xx < W) {
                    const T* p = st + (((long long)b * H + yy) * W + xx) * 2;
                    w0 = dz * ldf<T>(p); w1 = dz * ldf<T>(p + 1);
                }
                w0 = wave_sum(w0); w1 = wave_sum(w1);
                if ((threadIdx.x & 63) == 0) { atomicAdd(&red[ky * K + kx], w0); atomicAdd(&red[K * K + ky * K + kx], w1); }
            }
        const float dzs = wave_sum(dz);
        if ((threadIdx.x & 63) == 0) atomicAdd(&red[2 * K * K], dzs);
        if (in) { stf<T>(dst + (long long)i * 2, a0); stf<T>(dst + (long long)i * 2 + 1, a1); }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * K * K; i += 256) atomicAdd(dw + i, red[i]);
    if (threadIdx.x == 0) atomicAdd(db, red[2 * K * K]);
}

// y[r, c] = x[r, c] * g[r]
template <typename T>
__global__ void pix_gate_fwd_kernel(const T* x, int ldx, const T* g, T* y, int ldy, int rows, int C) {
    const int cq = C >> 2;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)rows * cq; i += gridDim.x * blockDim.x) {
        const int q = (int)(i % cq) * 4; const unsigned r = i / cq;
        const float a = ldf<T>(g + r);
        const float4 v = ld4<T>(x + (long long)r * ldx + q);
        st4<T>(y + (long long)r * ldy + q, make_float4(v.x * a, v.y * a, v.z * a, v.w * a));
    }
}

// dx (+)= dy * g[r];  dg[r] = sum_c dy * x.  16 lanes per row.
template <typename T>
__global__ __launch_bounds__(256) void pix_gate_bwd_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ x, int ldx, const T* __restrict__ g,
                                                           T* __restrict__ dx, int lddx, int acc, T* __restrict__ dg, int rows, int C) {
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    for (int r = blockIdx.x * 16 + ty; r < rows; r += gridDim.x * 16) {
        const float a = ldf<T>(g + r);
        float s = 0.f;
        for (int c = tx * 4; c < C; c += 64) {
            const float4 d = ld4<T>(dy + (long long)r * lddy + c), v = ld4<T>(x + (long long)r * ldx + c);
            s += d.x * v.x + d.y * v.y + d.z * v.z + d.w * v.w;
            float4 o = make_float4(d.x * a, d.y * a, d.z * a, d.w * a);
            T* dst = dx + (long long)r * lddx + c;
            if (acc) { const float4 w = ld4<T>(dst); o.x += w.x; o.y += w.y; o.z += w.z; o.w += w.w; }
            st4<T>(dst, o);
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o, 64);
        if (tx == 0) stf<T>(dg + r, s);
    }
}

// ---------------------------------------------------------------------------------------------------------------- CAM_Module
// (the concat = "cam" aggregate, MSTr.py:464-509).  x is [B * N, 4 * C]: the four branch maps side by side, column p * C + c.  Per image
// and channel: E[p][q] = sum_n x[n,p] x[n,q], A = softmax_q(max_q E[p,:] - E[p,q]), y[n,p] = gamma * sum_q A[p][q] x[n,q] + x[n,p].
// att holds A as [B][C][16] fp32.  Backward (softmax is shift invariant, so the row maximum carries no gradient):
//   dA[p][q] = gamma sum_n dy[n,p] x[n,q];  dEn = A (.) (dA - rowsum(A (.) dA));  dE = -dEn;  G = dE + dE^T;
//   dx[n,p] = dy[n,p] + gamma sum_q A[q][p] dy[n,q] + sum_q G[p][q] x[n,q];   dgamma = sum dy (.) (A x).
template <typename T>
__global__ __launch_bounds__(256) void cam_att_fwd_kernel(const T* __restrict__ x, int ldx, float* __restrict__ att, int N, int C) {
    __shared__ float red[16][64][10];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4, c = blockIdx.y * 64 + tx * 4, b = blockIdx.x;
    float e[4][10];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 10; ++i) e[k][i] = 0.f;
    if (c < C)
        for (int r = ty; r < N; r += 16) {
            const T* row = x + ((long long)b * N + r) * ldx + c;
            float v[4][4];                                              // [path][channel of the quad]
#pragma unroll
            for (int p = 0; p < 4; ++p) { const float4 t = ld4<T>(row + p * C); v[p][0] = t.x; v[p][1] = t.y; v[p][2] = t.z; v[p][3] = t.w; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int i = 0;
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int q = p; q < 4; ++q) e[k][i++] += v[p][k] * v[q][k];
            }
        }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 10; ++i) red[ty][tx * 4 + k][i] = e[k][i];
    __syncthreads();
    if (threadIdx.x < 64 && blockIdx.y * 64 + threadIdx.x < C) {
        const int cc = threadIdx.x;
        float E[4][4];
        int i = 0;
        for (int p = 0; p < 4; ++p)
            for (int q = p; q < 4; ++q, ++i) {
                float t = 0.f;
                for (int w = 0; w < 16; ++w) t += red[w][cc][i];
                E[p][q] = E[q][p] = t;
            }
        float* a = att + ((long long)b * C + blockIdx.y * 64 + cc) * 16;
        for (int p = 0; p < 4; ++p) {
            const float m = fmaxf(fmaxf(E[p][0], E[p][1]), fmaxf(E[p][2], E[p][3]));
            float en[4], mx = -INFINITY, sum = 0.f;
            for (int q = 0; q < 4; ++q) { en[q] = m - E[p][q]; mx = fmaxf(mx, en[q]); }
            for (int q = 0; q < 4; ++q) { en[q] = __expf(en[q] - mx); sum += en[q]; }
            for (int q = 0; q < 4; ++q) a[p * 4 + q] = en[q] / sum;
        }
    }
}

// MODE 0: y = gamma (A x) + x.   MODE 1 (backward): dx (+)= dy + gamma A^T dy + G x   (G in `att2`)
template <typename T, int MODE>
__global__ void cam_apply_kernel(const T* x, int ldx, const T* dy, int lddy, const float* att, const float* att2, const float* gamma, T* y, int ldy, int acc,
                                 int B, int N, int C) {
    const int cq = C >> 2;
    const float gm = *gamma;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)B * N * cq; i += gridDim.x * blockDim.x) {
        const int q4 = (int)(i % cq) * 4; const unsigned row = i / cq; const int b = (int)(row / (unsigned)N);
        float v[4][4], d[4][4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float4 t = ld4<T>(x + (long long)row * ldx + p * C + q4); v[p][0] = t.x; v[p][1] = t.y; v[p][2] = t.z; v[p][3] = t.w;
            if (MODE == 1) { const float4 u = ld4<T>(dy + (long long)row * lddy + p * C + q4); d[p][0] = u.x; d[p][1] = u.y; d[p][2] = u.z; d[p][3] = u.w; }
        }
        float o[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float* a = att + ((long long)b * C + q4 + k) * 16;
            if (MODE == 0) {
#pragma unroll
                for (int p = 0; p < 4; ++p) o[p][k] = gm * (a[p * 4] * v[0][k] + a[p * 4 + 1] * v[1][k] + a[p * 4 + 2] * v[2][k] + a[p * 4 + 3] * v[3][k]) + v[p][k];
            } else {
                const float* g = att2 + ((long long)b * C + q4 + k) * 16;
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    o[p][k] = d[p][k] + gm * (a[p] * d[0][k] + a[4 + p] * d[1][k] + a[8 + p] * d[2][k] + a[12 + p] * d[3][k]) +
                              g[p * 4] * v[0][k] + g[p * 4 + 1] * v[1][k] + g[p * 4 + 2] * v[2][k] + g[p * 4 + 3] * v[3][k];
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            T* dst = y + (long long)row * ldy + p * C + q4;
            float4 w = make_float4(o[p][0], o[p][1], o[p][2], o[p][3]);
            if (acc) { const float4 u = ld4<T>(dst); w.x += u.x; w.y += u.y; w.z += u.z; w.w += u.w; }
            st4<T>(dst, w);
        }
    }
}

// per (image, channel): dA = gamma sum_n dy x^T (16 sums) -> G = dE + dE^T into att2;  dgamma += sum dy (.) (A x)
template <typename T>
__global__ __launch_bounds__(256) void cam_bwd_reduce_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dy, int lddy, const float* __restrict__ att,
                                                             const float* __restrict__ gamma, float* __restrict__ att2, float* __restrict__ dgamma, int N, int C) {
    __shared__ float red[16][64][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4, c = blockIdx.y * 64 + tx * 4, b = blockIdx.x;
    float s[4][17];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 17; ++i) s[k][i] = 0.f;
    if (c < C) {
        float a[4][16];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) a[k][i] = att[((long long)b * C + c + k) * 16 + i];
        for (int r = ty; r < N; r += 16) {
            const long long off = ((long long)b * N + r);
            float v[4][4], d[4][4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float4 t = ld4<T>(x + off * ldx + p * C + c); v[p][0] = t.x; v[p][1] = t.y; v[p][2] = t.z; v[p][3] = t.w;
                const float4 u = ld4<T>(dy + off * lddy + p * C + c); d[p][0] = u.x; d[p][1] = u.y; d[p][2] = u.z; d[p][3] = u.w;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float ax = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { s[k][p * 4 + q] += d[p][k] * v[q][k]; ax += a[k][p * 4 + q] * v[q][k]; }
                    s[k][16] += d[p][k] * ax;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 17; ++i) red[ty][tx * 4 + k][i] = s[k][i];
    __syncthreads();
    float dgl = 0.f;
    if (threadIdx.x < 64 && blockIdx.y * 64 + threadIdx.x < C) {
        const int cc = threadIdx.x;
        const float gm = *gamma;
        float dA[16], dg = 0.f;
        for (int i = 0; i < 16; ++i) { float t = 0.f; for (int w = 0; w < 16; ++w) t += red[w][cc][i]; dA[i] = gm * t; }
        for (int w = 0; w < 16; ++w) dg += red[w][cc][16];
        dgl = dg;
        const float* a = att + ((long long)b * C + blockIdx.y * 64 + cc) * 16;
        float dE[16];
        for (int p = 0; p < 4; ++p) {
            float rs = 0.f;
            for (int q = 0; q < 4; ++q) rs += a[p * 4 + q] * dA[p * 4 + q];
            for (int q = 0; q < 4; ++q) dE[p * 4 + q] = -a[p * 4 + q] * (dA[p * 4 + q] - rs);
        }
        float* g = att2 + ((long long)b * C + blockIdx.y * 64 + cc) * 16;
        for (int p = 0; p < 4; ++p)
            for (int q = 0; q < 4; ++q) g[p * 4 + q] = dE[p * 4 + q] + dE[q * 4 + p];
    }
    if (threadIdx.x < 64) {
        dgl = wave_sum(dgl);
        if (threadIdx.x == 0) atomicAdd(dgamma, dgl);
    }
}

// y = gamma * a + x (the residual scale of the CAM modules, MSTr.py:508, 566);  backward: da = gamma dy, dx (+)= dy, dgamma += sum dy (.) a
template <typename T>
__global__ void gamma_res_fwd_kernel(const T* a, int lda, const T* x, int ldx, const float* gamma, T* y, int ldy, int rows, int C) {
    const int cq = C >> 2;
    const float gm = *gamma;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)rows * cq; i += gridDim.x * blockDim.x) {
        const int q = (int)(i % cq) * 4; const unsigned r = i / cq;
        const float4 u = ld4<T>(a + (long long)r * lda + q), v = ld4<T>(x + (long long)r * ldx + q);
        st4<T>(y + (long long)r * ldy + q, make_float4(gm * u.x + v.x, gm * u.y + v.y, gm * u.z + v.z, gm * u.w + v.w));
    }
}
template <typename T>
__global__ __launch_bounds__(256) void gamma_res_bwd_kernel(const T* dy, int lddy, const T* a, int lda, const float* gamma, T* da, int ldda, T* dx, int lddx,
                                                            int acc, float* dgamma, int rows, int C) {
    const int cq = C >> 2;
    const float gm = *gamma;
    float s = 0.f;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)rows * cq; i += gridDim.x * blockDim.x) {
        const int q = (int)(i % cq) * 4; const unsigned r = i / cq;
        const float4 d = ld4<T>(dy + (long long)r * lddy + q), u = ld4<T>(a + (long long)r * lda + q);
        s += d.x * u.x + d.y * u.y + d.z * u.z + d.w * u.w;
        st4<T>(da + (long long)r * ldda + q, make_float4(gm * d.x, gm * d.y, gm * d.z, gm * d.w));
        T* p = dx + (long long)r * lddx + q;
        float4 o = d;
        if (acc) { const float4 w = ld4<T>(p); o.x += w.x; o.y += w.y; o.z += w.z; o.w += w.w; }
        st4<T>(p, o);
    }
    __shared__ float red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dgamma, red[0] + red[1] + red[2] + red[3]);
}

template <typename T>
__global__ void gelu_fwd_kernel(const T* x, T* y, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) stf<T>(y + i, gelu_fT<T>(ldf<T>(x + i)));
}
template <typename T>
__global__ void gelu_bwd_kernel(const T* dy, const T* x, T* dz, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) stf<T>(dz + i, ldf<T>(dy + i) * gelu_grad_f(ldf<T>(x + i)));
}

inline dim3 gq(long long n) { return dim3(tc_blocks(n, 256, 8192)); }

}  // namespace

extern "C" int tc_cam_att_fwd(const void* x, int ldx, float* att, int B, int N, int C, int dtype, void* stream) {
    if (!x || !att || B <= 0 || N <= 0 || C <= 0 || ((C | ldx) & 3)) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((cam_att_fwd_kernel<T>), dim3(B, (C + 63) / 64), dim3(256), 0, TC_S, (const T*)x, ldx, att, N, C));
    return tc_launch_status();
}
extern "C" int tc_cam_apply_fwd(const void* x, int ldx, const float* att, const float* gamma, void* y, int ldy, int B, int N, int C, int dtype, void* stream) {
    if (!x || !att || !gamma || !y || B <= 0 || N <= 0 || C <= 0 || ((C | ldx | ldy) & 3) || (long long)B * N * (C / 4) >= 0x7fffffffLL) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((cam_apply_kernel<T, 0>), gq((long long)B * N * C / 4), dim3(256), 0, TC_S, (const T*)x, ldx, (const T*)nullptr, 0, att,
                                                (const float*)nullptr, gamma, (T*)y, ldy, 0, B, N, C));
    return tc_launch_status();
}
extern "C" int tc_cam_bwd(const void* x, int ldx, const void* dy, int lddy, const float* att, const float* gamma, float* att2, float* dgamma, void* dx, int lddx,
                          int dx_accumulate, int B, int N, int C, int dtype, void* stream) {
    if (!x || !dy || !att || !gamma || !att2 || !dgamma || !dx || B <= 0 || N <= 0 || C <= 0 || ((C | ldx | lddy | lddx) & 3) ||
        (long long)B * N * (C / 4) >= 0x7fffffffLL)
        return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, {
        hipLaunchKernelGGL((cam_bwd_reduce_kernel<T>), dim3(B, (C + 63) / 64), dim3(256), 0, TC_S, (const T*)x, ldx, (const T*)dy, lddy, att, gamma, att2, dgamma, N, C);
        hipLaunchKernelGGL((cam_apply_kernel<T, 1>), gq((long long)B * N * C / 4), dim3(256), 0, TC_S, (const T*)x, ldx, (const T*)dy, lddy, att, att2, gamma, (T*)dx, lddx,
                           dx_accumulate, B, N, C);
    });
    return tc_launch_status();
}
extern "C" int tc_gamma_res_fwd(const void* a, int lda, const void* x, int ldx, const float* gamma, void* y, int ldy, int rows, int C, int dtype, void* stream) {
    if (!a || !x || !gamma || !y || rows <= 0 || C <= 0 || ((C | lda | ldx | ldy) & 3) || (long long)rows * (C / 4) >= 0x7fffffffLL) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gamma_res_fwd_kernel<T>), gq((long long)rows * C / 4), dim3(256), 0, TC_S, (const T*)a, lda, (const T*)x, ldx, gamma,
                                                (T*)y, ldy, rows, C));
    return tc_launch_status();
}
extern "C" int tc_gamma_res_bwd(const void* dy, int lddy, const void* a, int lda, const float* gamma, void* da, int ldda, void* dx, int lddx, int dx_accumulate,
                                float* dgamma, int rows, int C, int dtype, void* stream) {
    if (!dy || !a || !gamma || !da || !dx || !dgamma || rows <= 0 || C <= 0 || ((C | lddy | lda | ldda | lddx) & 3) || (long long)rows * (C / 4) >= 0x7fffffffLL)
        return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gamma_res_bwd_kernel<T>), dim3(tc_blocks((long long)rows * C / 4, 256, 1024)), dim3(256), 0, TC_S, (const T*)dy, lddy,
                                                (const T*)a, lda, gamma, (T*)da, ldda, (T*)dx, lddx, dx_accumulate, dgamma, rows, C));
    return tc_launch_status();
}
extern "C" int tc_gelu_fwd(const void* x, void* y, long long n, int dtype, void* stream) {
    if (!x || !y || n <= 0) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gelu_fwd_kernel<T>), gq(n), dim3(256), 0, TC_S, (const T*)x, (T*)y, n));
    return tc_launch_status();
}
extern "C" int tc_gelu_bwd(const void* dy, const void* x, void* dz, long long n, int dtype, void* stream) {
    if (!dy || !x || !dz || n <= 0) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((gelu_bwd_kernel<T>), gq(n), dim3(256), 0, TC_S, (const T*)dy, (const T*)x, (T*)dz, n));
    return tc_launch_status();
}

extern "C" int tc_chan_pool2_fwd(const void* x, int ldx, void* pooled, int* idx, int B, int N, int C, int dtype, void* stream) {
    if (!x || !pooled || !idx || B <= 0 || N <= 0 || C <= 0 || ((C | ldx) & 3) || (long long)B * N * (C / 4) >= 0x7fffffffLL) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((chan_pool2_fwd_kernel<T>), dim3(B, (C + 63) / 64), dim3(256), 0, TC_S, (const T*)x, ldx, (T*)pooled, idx, B, N, C));
    return tc_launch_status();
}
extern "C" int tc_chan_pool2_bwd(const void* dpooled, const int* idx, void* dx, int lddx, int B, int N, int C, int accumulate, int dtype, void* stream) {
    if (!dpooled || !idx || !dx || B <= 0 || N <= 0 || C <= 0 || ((C | lddx) & 3) || (long long)B * N * (C / 4) >= 0x7fffffffLL) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((chan_pool2_bwd_kernel<T>), gq((long long)B * N * C / 4), dim3(256), 0, TC_S, (const T*)dpooled, idx, (T*)dx, lddx,
                                                B, N, C, accumulate));
    return tc_launch_status();
}
extern "C" int tc_pix_stats_fwd(const void* x, int ldx, void* st, int* idx, int rows, int C, int dtype, void* stream) {
    if (!x || !st || !idx || rows <= 0 || C <= 0 || ((C | ldx) & 3)) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((pix_stats_fwd_kernel<T>), dim3(tc_blocks(rows, 16, 4096)), dim3(256), 0, TC_S, (const T*)x, ldx, (T*)st, idx, rows, C));
    return tc_launch_status();
}
extern "C" int tc_pix_stats_bwd(const void* dst, const int* idx, void* dx, int lddx, int rows, int C, int accumulate, int dtype, void* stream) {
    if (!dst || !idx || !dx || rows <= 0 || C <= 0 || ((C | lddx) & 3) || (long long)rows * (C / 4) >= 0x7fffffffLL) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((pix_stats_bwd_kernel<T>), gq((long long)rows * C / 4), dim3(256), 0, TC_S, (const T*)dst, idx, (T*)dx, lddx, rows, C,
                                                accumulate));
    return tc_launch_status();
}
extern "C" int tc_sa_conv_fwd(const void* st, const void* w, const void* bias, void* g, int B, int H, int W, int k, int dtype, void* stream) {
    if (!st || !w || !bias || !g || B <= 0 || H <= 0 || W <= 0 || (k != 3 && k != 7) || (long long)B * H * W >= 0x7fffffffLL) return TC_ERR_ARG;
    const dim3 grid(tc_blocks((long long)B * H * W, 256, 4096));
    TC_DISPATCH_DTYPE(dtype, {
        if (k == 7) hipLaunchKernelGGL((sa_conv_fwd_kernel<T, 7>), grid, dim3(256), 0, TC_S, (const T*)st, (const T*)w, (const T*)bias, (T*)g, B, H, W);
        else hipLaunchKernelGGL((sa_conv_fwd_kernel<T, 3>), grid, dim3(256), 0, TC_S, (const T*)st, (const T*)w, (const T*)bias, (T*)g, B, H, W);
    });
    return tc_launch_status();
}
extern "C" int tc_sa_conv_bwd(const void* dg, const void* g, const void* st, const void* w, void* dst, float* dw, float* db, int B, int H, int W, int k, int dtype,
                              void* stream) {
    if (!dg || !g || !st || !w || !dst || !dw || !db || B <= 0 || H <= 0 || W <= 0 || (k != 3 && k != 7) || (long long)B * H * W >= 0x7fffffffLL) return TC_ERR_ARG;
    const dim3 grid(tc_blocks((long long)B * H * W, 256, 256));
    TC_DISPATCH_DTYPE(dtype, {
        if (k == 7) hipLaunchKernelGGL((sa_conv_bwd_kernel<T, 7>), grid, dim3(256), 0, TC_S, (const T*)dg, (const T*)g, (const T*)st, (const T*)w, (T*)dst, dw, db, B, H, W);
        else hipLaunchKernelGGL((sa_conv_bwd_kernel<T, 3>), grid, dim3(256), 0, TC_S, (const T*)dg, (const T*)g, (const T*)st, (const T*)w, (T*)dst, dw, db, B, H, W);
    });
    return tc_launch_status();
}
extern "C" int tc_pix_gate_fwd(const void* x, int ldx, const void* g, void* y, int ldy, int rows, int C, int dtype, void* stream) {
    if (!x || !g || !y || rows <= 0 || C <= 0 || ((C | ldx | ldy) & 3) || (long long)rows * (C / 4) >= 0x7fffffffLL) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((pix_gate_fwd_kernel<T>), gq((long long)rows * C / 4), dim3(256), 0, TC_S, (const T*)x, ldx, (const T*)g, (T*)y, ldy,
                                                rows, C));
    return tc_launch_status();
}
extern "C" int tc_pix_gate_bwd(const void* dy, int lddy, const void* x, int ldx, const void* g, void* dx, int lddx, int dx_accumulate, void* dg, int rows, int C,
                               int dtype, void* stream) {
    if (!dy || !x || !g || !dx || !dg || rows <= 0 || C <= 0 || ((C | lddy | ldx | lddx) & 3)) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((pix_gate_bwd_kernel<T>), dim3(tc_blocks(rows, 16, 4096)), dim3(256), 0, TC_S, (const T*)dy, lddy, (const T*)x, ldx,
                                                (const T*)g, (T*)dx, lddx, dx_accumulate, (T*)dg, rows, C));
    return tc_launch_status();
}
