// Elementwise and layout kernels of the TransCeption path (all HBM-bound, 4 channels = 8/16 B per lane).
#include "tc_common.h"

namespace {

// Element index of a grid-stride loop.  32-bit: the kernels below split it with run-time divisors (row / column, pixel / channel quad),
// and a 64-bit division is ~100 instructions on this ISA -- the layout moves were integer-ALU-bound.  g1() refuses n >= 2^31 (the launch
// then fails and the entry point reports it); the two kernels without a division keep the 64-bit form.
#define TC_GRID_STRIDE(i, n) for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)(n); i += gridDim.x * blockDim.x)
#define TC_GRID_STRIDE64(i, n) for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

template <typename T>
__global__ void fma3_fwd_kernel(const T* a, int lda, const T* b, int ldb, const T* c, int ldc, T* o, int ldo, int rows, int cq,
                                float alpha) {
    TC_GRID_STRIDE(i, (long long)rows * cq) {
        const long long r = i / cq; const int q = (int)(i % cq) * 4;
        const float4 va = ld4<T>(a + r * lda + q), vb = ld4<T>(b + r * ldb + q), vc = ld4<T>(c + r * ldc + q);
        st4<T>(o + r * ldo + q, make_float4(alpha * va.x + vb.x * vc.x, alpha * va.y + vb.y * vc.y, alpha * va.z + vb.z * vc.z,
                                            alpha * va.w + vb.w * vc.w));
    }
}

template <typename T>
__global__ void fma3_bwd_kernel(const T* d, int ldd, const T* b, int ldb, const T* c, int ldc, T* da, int ldda, T* db, int lddb,
                                int db_acc, T* dc, int lddc, int rows, int cq, float alpha) {
    TC_GRID_STRIDE(i, (long long)rows * cq) {
        const long long r = i / cq; const int q = (int)(i % cq) * 4;
        const float4 vd = ld4<T>(d + r * ldd + q), vb = ld4<T>(b + r * ldb + q), vc = ld4<T>(c + r * ldc + q);
        st4<T>(da + r * ldda + q, make_float4(alpha * vd.x, alpha * vd.y, alpha * vd.z, alpha * vd.w));
        float4 g = make_float4(vd.x * vc.x, vd.y * vc.y, vd.z * vc.z, vd.w * vc.w);
        if (db_acc) { const float4 o = ld4<T>(db + r * lddb + q); g.x += o.x; g.y += o.y; g.z += o.z; g.w += o.w; }
        st4<T>(db + r * lddb + q, g);
        st4<T>(dc + r * lddc + q, make_float4(vd.x * vb.x, vd.y * vb.y, vd.z * vb.z, vd.w * vb.w));
    }
}

template <typename T>
__global__ void add_kernel(const T* a, int lda, const T* b, int ldb, T* y, int ldy, int rows, int cq) {
    TC_GRID_STRIDE(i, (long long)rows * cq) {
        const long long r = i / cq; const int q = (int)(i % cq) * 4;
        const float4 va = ld4<T>(a + r * lda + q), vb = ld4<T>(b + r * ldb + q);
        st4<T>(y + r * ldy + q, make_float4(va.x + vb.x, va.y + vb.y, va.z + vb.z, va.w + vb.w));
    }
}

template <typename T>
__global__ void sigmoid_bwd_kernel(const T* dy, const T* s, T* dz, long long n) {
    TC_GRID_STRIDE64(i, n) { const float v = ldf<T>(s + i); stf<T>(dz + i, ldf<T>(dy + i) * v * (1.f - v)); }
}

template <typename T>
__device__ __forceinline__ void copy3d_body(const T* src, long long sbs, int lds, T* dst, long long sbd, int ldd, int nb, int rows, int cols, int acc) {
    constexpr int V = 16 / (int)sizeof(T);                      // elements of a 16-byte piece
    const bool wide = !((cols | lds | ldd) % V) && !(sbs % V) && !(sbd % V) && !(((uintptr_t)src | (uintptr_t)dst) & 15);
    if (wide) {                                                 // whole 16-byte pieces per thread (the residual-gradient copies: 8 elements at a time)
        const int cv = cols / V;
        TC_GRID_STRIDE(i, (long long)nb * rows * cv) {
            const int c = (int)(i % cv) * V; const unsigned t = i / cv; const int r = (int)(t % rows); const long long b = t / rows;
            T* d = dst + b * sbd + (long long)r * ldd + c;
            const uint4 sv = *reinterpret_cast<const uint4*>(src + b * sbs + (long long)r * lds + c);
            if (!acc) { *reinterpret_cast<uint4*>(d) = sv; continue; }
            const uint4 dv = *reinterpret_cast<const uint4*>(d);
            if constexpr (sizeof(T) == 4) {
                const float4 a = __builtin_bit_cast(float4, sv), o = __builtin_bit_cast(float4, dv);
                *reinterpret_cast<float4*>(d) = make_float4(a.x + o.x, a.y + o.y, a.z + o.z, a.w + o.w);
            } else {
                const unsigned sw[4] = {sv.x, sv.y, sv.z, sv.w}, dw[4] = {dv.x, dv.y, dv.z, dv.w};
                unsigned ow[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { float a0, a1, o0, o1; unpack2<T>(sw[e], a0, a1); unpack2<T>(dw[e], o0, o1); ow[e] = pack2<T>(a0 + o0, a1 + o1); }
                *reinterpret_cast<uint4*>(d) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            }
        }
        return;
    }
    TC_GRID_STRIDE(i, (long long)nb * rows * cols) {
        const int c = (int)(i % cols); const unsigned t = i / cols; const int r = (int)(t % rows); const long long b = t / rows;
        T* d = dst + b * sbd + (long long)r * ldd + c;
        float v = ldf<T>(src + b * sbs + (long long)r * lds + c);
        if (acc) v += ldf<T>(d);
        stf<T>(d, v);
    }
}
template <typename T>
__global__ void copy3d_kernel(const T* src, long long sbs, int lds, T* dst, long long sbd, int ldd, int nb, int rows, int cols, int acc) {
    copy3d_body<T>(src, sbs, lds, dst, sbd, ldd, nb, rows, cols, acc);
}

template <typename T>
__global__ void transpose_kernel(const T* src, T* dst, int R, int Cc) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const T* s = src + (long long)b * R * Cc;
    T* d = dst + (long long)b * R * Cc;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    for (int j = ty; j < 32; j += 8) if (r0 + j < R && c0 + tx < Cc) tile[j][tx] = ldf<T>(s + (long long)(r0 + j) * Cc + c0 + tx);
    __syncthreads();
    for (int j = ty; j < 32; j += 8) if (c0 + j < Cc && r0 + tx < R) stf<T>(d + (long long)(c0 + j) * R + r0 + tx, tile[tx][j]);
}

// ---------------------------------------------------------------- SE_Block (concat = "se"): squeeze, gate, ReLU
// One workgroup = (image b, 64 channels): 16 channel quads x 16 row lanes walk the image's N token rows; the 16 row lanes meet in LDS.
template <typename T>
__global__ __launch_bounds__(256) void chan_pool_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ pooled, int N, int C) {
    __shared__ float4 red[16][16];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4, c = blockIdx.y * 64 + tx * 4, b = blockIdx.x;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C)
        for (int r = ty; r < N; r += 16) { const float4 v = ld4<T>(x + ((long long)b * N + r) * ldx + c); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < C) {
        float4 t = red[0][tx];
#pragma unroll
        for (int k = 1; k < 16; ++k) { const float4 v = red[k][tx]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        const float inv = 1.f / (float)N;
        st4<T>(pooled + (long long)b * C + c, make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv));
    }
}

template <typename T>
__global__ void chan_pool_bwd_kernel(const T* dp, T* dx, int lddx, int B, int N, int C, int acc) {
    const int cq = C >> 2;
    const float inv = 1.f / (float)N;
    TC_GRID_STRIDE(i, (long long)B * N * cq) {
        const int q = (int)(i % cq) * 4; const unsigned row = i / cq; const int b = (int)(row / (unsigned)N);
        const float4 g = ld4<T>(dp + (long long)b * C + q);
        float4 o = make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv);
        T* d = dx + (long long)row * lddx + q;
        if (acc) { const float4 v = ld4<T>(d); o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w; }
        st4<T>(d, o);
    }
}

template <typename T>
__global__ void chan_gate_fwd_kernel(const T* x, int ldx, const T* g, T* y, int ldy, int B, int N, int C) {
    const int cq = C >> 2;
    TC_GRID_STRIDE(i, (long long)B * N * cq) {
        const int q = (int)(i % cq) * 4; const unsigned row = i / cq; const int b = (int)(row / (unsigned)N);
        const float4 v = ld4<T>(x + (long long)row * ldx + q), a = ld4<T>(g + (long long)b * C + q);
        st4<T>(y + (long long)row * ldy + q, make_float4(v.x * a.x, v.y * a.y, v.z * a.z, v.w * a.w));
    }
}

// dx (+)= dy * g[b]; dg[b][c] = sum over the image's rows of dy * x
template <typename T>
__global__ __launch_bounds__(256) void chan_gate_bwd_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ x, int ldx,
                                                            const T* __restrict__ g, T* __restrict__ dx, int lddx, int acc,
                                                            T* __restrict__ dg, int N, int C) {
    __shared__ float4 red[16][16];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4, c = blockIdx.y * 64 + tx * 4, b = blockIdx.x;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {
        const float4 a = ld4<T>(g + (long long)b * C + c);
        for (int r = ty; r < N; r += 16) {
            const long long row = (long long)b * N + r;
            const float4 d = ld4<T>(dy + row * lddy + c), v = ld4<T>(x + row * ldx + c);
            s.x += d.x * v.x; s.y += d.y * v.y; s.z += d.z * v.z; s.w += d.w * v.w;
            float4 o = make_float4(d.x * a.x, d.y * a.y, d.z * a.z, d.w * a.w);
            T* dst = dx + row * lddx + c;
            if (acc) { const float4 w = ld4<T>(dst); o.x += w.x; o.y += w.y; o.z += w.z; o.w += w.w; }
            st4<T>(dst, o);
        }
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < C) {
        float4 t = red[0][tx];
#pragma unroll
        for (int k = 1; k < 16; ++k) { const float4 v = red[k][tx]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        st4<T>(dg + (long long)b * C + c, t);
    }
}

template <typename T>
__global__ void relu_fwd_kernel(const T* x, T* y, long long n) {
    TC_GRID_STRIDE64(i, n) { stf<T>(y + i, fmaxf(ldf<T>(x + i), 0.f)); }
}
template <typename T>
__global__ void relu_bwd_kernel(const T* dy, const T* y, T* dz, long long n) {
    TC_GRID_STRIDE64(i, n) { stf<T>(dz + i, ldf<T>(y + i) > 0.f ? ldf<T>(dy + i) : 0.f); }
}

// ---------------------------------------------------------------- CoordAtt (IFF) pooling and gating
template <typename T>
__global__ void coord_pool_fwd_kernel(const T* x, T* pooled, int B, int H, int W, int C) {
    const int cq = C >> 2;
    TC_GRID_STRIDE(i, (long long)B * (H + W) * cq) {
        const int q = (int)(i % cq) * 4; const int j = (int)((i / cq) % (H + W)); const int b = (int)(i / (unsigned)(cq * (H + W)));
        const T* xb = x + (long long)b * H * W * C + q;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < H) { for (int w = 0; w < W; ++w) { const float4 v = ld4<T>(xb + ((long long)j * W + w) * C); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
                     const float inv = 1.f / W; s.x *= inv; s.y *= inv; s.z *= inv; s.w *= inv; }
        else { const int w = j - H; for (int h = 0; h < H; ++h) { const float4 v = ld4<T>(xb + ((long long)h * W + w) * C); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
               const float inv = 1.f / H; s.x *= inv; s.y *= inv; s.z *= inv; s.w *= inv; }
        st4<T>(pooled + (j < H ? ((long long)b * H + j) : ((long long)B * H + (long long)b * W + (j - H))) * C + q, s);
    }
}

template <typename T>
__global__ void coord_pool_bwd_kernel(const T* dp, T* dx, int B, int H, int W, int C, int acc) {
    const int cq = C >> 2;
    TC_GRID_STRIDE(i, (long long)B * H * W * cq) {
        const int q = (int)(i % cq) * 4; const unsigned pix32 = i / cq;
        const int w = (int)(pix32 % W), h = (int)((pix32 / W) % H), b = (int)(pix32 / (unsigned)(W * H));
        const long long pix = pix32;
        const float4 gh = ld4<T>(dp + ((long long)b * H + h) * C + q), gw = ld4<T>(dp + ((long long)B * H + (long long)b * W + w) * C + q);
        const float ih = 1.f / W, iw = 1.f / H;
        float4 g = make_float4(gh.x * ih + gw.x * iw, gh.y * ih + gw.y * iw, gh.z * ih + gw.z * iw, gh.w * ih + gw.w * iw);
        T* o = dx + pix * C + q;
        if (acc) { const float4 v = ld4<T>(o); g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w; }
        st4<T>(o, g);
    }
}

template <typename T>
__global__ void coord_gate_fwd_kernel(const T* x, const T* att, T* y, int B, int H, int W, int C) {
    const int cq = C >> 2;
    TC_GRID_STRIDE(i, (long long)B * H * W * cq) {
        const int q = (int)(i % cq) * 4; const unsigned pix32 = i / cq;
        const int w = (int)(pix32 % W), h = (int)((pix32 / W) % H), b = (int)(pix32 / (unsigned)(W * H));
        const long long pix = pix32;
        const float4 ah = ld4<T>(att + ((long long)b * H + h) * C + q), aw = ld4<T>(att + ((long long)B * H + (long long)b * W + w) * C + q);
        const float4 v = ld4<T>(x + pix * C + q);
        st4<T>(y + pix * C + q, make_float4(v.x * aw.x * ah.x, v.y * aw.y * ah.y, v.z * aw.z * ah.z, v.w * aw.w * ah.w));
    }
}

// dx = dy*aw*ah (optionally accumulated)
template <typename T>
__global__ void coord_gate_bwd_dx_kernel(const T* dy, const T* att, T* dx, int acc, int B, int H, int W, int C) {
    const int cq = C >> 2;
    TC_GRID_STRIDE(i, (long long)B * H * W * cq) {
        const int q = (int)(i % cq) * 4; const unsigned pix32 = i / cq;
        const int w = (int)(pix32 % W), h = (int)((pix32 / W) % H), b = (int)(pix32 / (unsigned)(W * H));
        const long long pix = pix32;
        const float4 ah = ld4<T>(att + ((long long)b * H + h) * C + q), aw = ld4<T>(att + ((long long)B * H + (long long)b * W + w) * C + q);
        const float4 d = ld4<T>(dy + pix * C + q);
        float4 g = make_float4(d.x * aw.x * ah.x, d.y * aw.y * ah.y, d.z * aw.z * ah.z, d.w * aw.w * ah.w);
        T* o = dx + pix * C + q;
        if (acc) { const float4 v = ld4<T>(o); g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w; }
        st4<T>(o, g);
    }
}

// datt[b,h] = sum_w dy*x*aw ; datt[b,H+w] = sum_h dy*x*ah
template <typename T>
__global__ void coord_gate_bwd_att_kernel(const T* dy, const T* x, const T* att, T* datt, int B, int H, int W, int C) {
    const int cq = C >> 2;
    TC_GRID_STRIDE(i, (long long)B * (H + W) * cq) {
        const int q = (int)(i % cq) * 4; const int j = (int)((i / cq) % (H + W)); const int b = (int)(i / (unsigned)(cq * (H + W)));
        const T* ah_b = att + (long long)b * H * C + q;
        const T* aw_b = att + ((long long)B * H + (long long)b * W) * C + q;
        const long long base = (long long)b * H * W;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < H) {
            for (int w = 0; w < W; ++w) {
                const long long pix = base + (long long)j * W + w;
                const float4 d = ld4<T>(dy + pix * C + q), v = ld4<T>(x + pix * C + q), a = ld4<T>(aw_b + (long long)w * C);
                s.x += d.x * v.x * a.x; s.y += d.y * v.y * a.y; s.z += d.z * v.z * a.z; s.w += d.w * v.w * a.w;
            }
        } else {
            const int w = j - H;
            for (int h = 0; h < H; ++h) {
                const long long pix = base + (long long)h * W + w;
                const float4 d = ld4<T>(dy + pix * C + q), v = ld4<T>(x + pix * C + q), a = ld4<T>(ah_b + (long long)h * C);
                s.x += d.x * v.x * a.x; s.y += d.y * v.y * a.y; s.z += d.z * v.z * a.z; s.w += d.w * v.w * a.w;
            }
        }
        st4<T>(datt + (j < H ? ((long long)b * H + j) : ((long long)B * H + (long long)b * W + (j - H))) * C + q, s);
    }
}

// ---------------------------------------------------------------- layout permutations
template <typename T>
__global__ void pixel_shuffle_kernel(const T* in, T* out, int B, int H, int W, int p, int c, int inverse) {
    // expanded pixel (b, h*p+p1, w*p+p2), channel cc  <->  coarse pixel (b,h,w), channel (p1*p+p2)*c + cc
    const int cq = c >> 2;
    const long long n = (long long)B * H * p * W * p * cq;
    TC_GRID_STRIDE(i, n) {
        const int q = (int)(i % cq) * 4; unsigned t = i / cq;
        const int ow = (int)(t % (unsigned)(W * p)); t /= (unsigned)(W * p);
        const int oh = (int)(t % (unsigned)(H * p)); const int b = (int)(t / (unsigned)(H * p));
        const int h = oh / p, p1 = oh % p, w = ow / p, p2 = ow % p;
        const long long fine = (((long long)b * H * p + oh) * W * p + ow) * c + q;
        const long long coarse = (((long long)b * H + h) * W + w) * (long long)(p * p * c) + (p1 * p + p2) * c + q;
        if (!inverse) st4<T>(out + fine, ld4<T>(in + coarse)); else st4<T>(out + coarse, ld4<T>(in + fine));
    }
}

// 16-bit storage, k = 2 / 4 / 8 (Scale_reduce's three convolutions, MSTr.py:2225-2249): a thread owns one filter row ky of one patch and four
// channels -- k pixels x 8 bytes on the map side, four runs of k consecutive columns (c, ky, 0..k-1) on the matrix side -- and moves whole
// 4 / 8 / 16-byte pieces between them with byte permutes, no conversion.  The element-wise walk below wrote (or read) the matrix two bytes at a
// time at a stride of k k elements: 20 us for the 23 MB of a bridge layer's three maps.  Lane order by direction, so that the side being WRITTEN
// is contiguous across adjacent lanes: ky fastest towards the matrix (eight lanes fill a channel's 128-byte run at k = 8), the channel group
// fastest towards the map.  Indices through sdiv (tc_common.h): the caller checks n + divisors < 2^22 and that every offset fits 31 bits.
template <typename T, int K>
__device__ __forceinline__ void patchify16_body(T* m, long long sb, int ld, T* cols, int B, int H, int W, int C, int inverse) {
    static_assert(sizeof(T) == 2 && (K == 2 || K == 4 || K == 8), "16-bit storage, power-of-two patch");
    constexpr int KK = K * K, LK = K == 2 ? 1 : (K == 4 ? 2 : 3), NW = K / 2;
    const int cqn = C >> 2, Ho = H / K, Wo = W / K;
    const unsigned n = (unsigned)B * Ho * Wo * K * cqn;
    const SDiv dcq = sdiv_make(cqn), dWo = sdiv_make(Wo), dHo = sdiv_make(Ho);
    TC_GRID_STRIDE(i, n) {
        int ky, cq, t;
        if (!inverse) { ky = i & (K - 1); const int t0 = (int)(i >> LK); t = sdiv(t0, dcq); cq = smod(t0, t, dcq); }
        else { const int t0 = sdiv((int)i, dcq); cq = smod((int)i, t0, dcq); ky = t0 & (K - 1); t = t0 >> LK; }
        const int t1 = sdiv(t, dWo), wo = smod(t, t1, dWo), b = sdiv(t1, dHo), ho = smod(t1, b, dHo);
        T* mp = m + b * sb + (long long)(((ho * K + ky) * W + wo * K) * ld + cq * 4);
        T* cp = cols + (long long)(((b * Ho + ho) * Wo + wo) * (KK * C) + cq * 4 * KK + ky * K);
        unsigned w[4][NW];                                          // channel j, pixels (2 m, 2 m + 1)
        if (!inverse) {
            uint2 px[K];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) px[kx] = *reinterpret_cast<const uint2*>(mp + kx * ld);
#pragma unroll
            for (int mw = 0; mw < NW; ++mw) {
                w[0][mw] = __byte_perm(px[2 * mw].x, px[2 * mw + 1].x, 0x5410); w[1][mw] = __byte_perm(px[2 * mw].x, px[2 * mw + 1].x, 0x7632);
                w[2][mw] = __byte_perm(px[2 * mw].y, px[2 * mw + 1].y, 0x5410); w[3][mw] = __byte_perm(px[2 * mw].y, px[2 * mw + 1].y, 0x7632);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (K == 8) *reinterpret_cast<uint4*>(cp + j * KK) = make_uint4(w[j][0], w[j][1], w[j][2], w[j][3]);
                else if constexpr (K == 4) *reinterpret_cast<uint2*>(cp + j * KK) = make_uint2(w[j][0], w[j][1]);
                else *reinterpret_cast<unsigned*>(cp + j * KK) = w[j][0];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (K == 8) { const uint4 v = *reinterpret_cast<const uint4*>(cp + j * KK); w[j][0] = v.x; w[j][1] = v.y; w[j][2] = v.z; w[j][3] = v.w; }
                else if constexpr (K == 4) { const uint2 v = *reinterpret_cast<const uint2*>(cp + j * KK); w[j][0] = v.x; w[j][1] = v.y; }
                else w[j][0] = *reinterpret_cast<const unsigned*>(cp + j * KK);
            }
            uint2 old[K];
            if (inverse == 2) {
#pragma unroll
                for (int kx = 0; kx < K; ++kx) old[kx] = *reinterpret_cast<const uint2*>(mp + kx * ld);
            }
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int mw = kx >> 1;
                uint2 v;
                if (kx & 1) { v.x = __byte_perm(w[0][mw], w[1][mw], 0x7632); v.y = __byte_perm(w[2][mw], w[3][mw], 0x7632); }
                else { v.x = __byte_perm(w[0][mw], w[1][mw], 0x5410); v.y = __byte_perm(w[2][mw], w[3][mw], 0x5410); }
                if (inverse == 2) {
                    float a0, a1, a2, a3, o0, o1, o2, o3;
                    unpack2<T>(v.x, a0, a1); unpack2<T>(v.y, a2, a3); unpack2<T>(old[kx].x, o0, o1); unpack2<T>(old[kx].y, o2, o3);
                    v.x = pack2<T>(a0 + o0, a1 + o1); v.y = pack2<T>(a2 + o2, a3 + o3);
                }
                *reinterpret_cast<uint2*>(mp + kx * ld) = v;
            }
        }
    }
}
// whether patchify16_body may run: 16-byte pieces aligned, every index below 2^22 and every element offset below 2^31
template <typename T>
__device__ __forceinline__ bool patchify16_ok(const T* map, long long sb, int ld, const T* cols, int B, int H, int W, int C, int k) {
    if (sizeof(T) != 2 || (k != 2 && k != 4 && k != 8)) return false;
    const long long n = (long long)B * (H / k) * (W / k) * k * (C >> 2);
    return n + H + W + C < (1LL << 22) && (long long)B * sb < (1LL << 31) && (long long)H * W * ld < (1LL << 31) && (long long)B * H * W * C < (1LL << 31) &&
           !(((uintptr_t)map | (uintptr_t)cols) & 15) && !(ld & 3) && !(sb & 3);
}

template <typename T>
__device__ __forceinline__ void patchify_body(const T* map, long long sb, int ld, T* cols, int B, int H, int W, int C, int k, int inverse) {
    const int cq = C >> 2, Ho = H / k, Wo = W / k;
    const long long n = (long long)B * H * W * cq;
    T* m = const_cast<T*>(map);
    if constexpr (sizeof(T) == 2) {
        if (patchify16_ok<T>(map, sb, ld, cols, B, H, W, C, k)) {
            if (k == 8) patchify16_body<T, 8>(m, sb, ld, cols, B, H, W, C, inverse);
            else if (k == 4) patchify16_body<T, 4>(m, sb, ld, cols, B, H, W, C, inverse);
            else patchify16_body<T, 2>(m, sb, ld, cols, B, H, W, C, inverse);
            return;
        }
    }
    TC_GRID_STRIDE(i, n) {
        const int q = (int)(i % cq) * 4; unsigned t = i / cq;
        const int w = (int)(t % W); t /= W; const int h = (int)(t % H); const int b = (int)(t / H);
        const long long src = b * sb + ((long long)h * W + w) * ld + q;
        const long long row = ((long long)b * Ho + h / k) * Wo + w / k;
        const int kk = k * k;
        T* d = cols + row * (long long)(kk * C) + (long long)q * kk + (h % k) * k + (w % k);     // column (c, ky, kx)
        if (!inverse) {
            const float4 v = ld4<T>(m + src);
            stf<T>(d, v.x); stf<T>(d + kk, v.y); stf<T>(d + 2 * kk, v.z); stf<T>(d + 3 * kk, v.w);
        } else {
            float4 v = make_float4(ldf<T>(d), ldf<T>(d + kk), ldf<T>(d + 2 * kk), ldf<T>(d + 3 * kk));
            if (inverse == 2) { const float4 o = ld4<T>(m + src); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            st4<T>(m + src, v);
        }
    }
}
template <typename T>
__global__ void patchify_kernel(const T* map, long long sb, int ld, T* cols, int B, int H, int W, int C, int k, int inverse) {
    patchify_body<T>(map, sb, ld, cols, B, H, W, C, k, inverse);
}

template <typename T>
__device__ __forceinline__ void sr_deinterleave_body(const T* in, T* out, long long sbo, int ldo, int B, int P, int C, int mult, int inverse) {
    // out[b, g*P+pos, c] = in[b, pos, c*mult+g]   (in: [B, P, C*mult] contiguous)
    const long long n = (long long)B * P * C * mult;
    T* o = out; T* ii = const_cast<T*>(in);
    TC_GRID_STRIDE(i, n) {
        const int c = (int)(i % C); unsigned t = i / C;
        const int pos = (int)(t % P); t /= P; const int g = (int)(t % mult); const int b = (int)(t / mult);
        const long long oi = b * sbo + ((long long)g * P + pos) * ldo + c;
        const long long si = ((long long)b * P + pos) * (C * mult) + c * mult + g;
        if (!inverse) o[oi] = ii[si]; else ii[si] = o[oi];
    }
}
template <typename T>
__global__ void sr_deinterleave_kernel(const T* in, T* out, long long sbo, int ldo, int B, int P, int C, int mult, int inverse) {
    sr_deinterleave_body<T>(in, out, sbo, ldo, B, P, C, mult, inverse);
}

// Up to TC_EW_MULTI_MAX independent layout moves (patchify / de-interleave / strided copy) in ONE launch: blockIdx.y picks the segment.
// Scale_reduce (MSTr.py:2225-2249) is three patchifies, three convolutions as GEMMs, three de-interleaves and a copy on maps of a few
// hundred KB: each of them alone sits at the ~4.5 us floor of a launch.
struct EwMultiDev { TcEwSeg s[TC_EW_MULTI_MAX]; };
template <typename T>
__global__ void ew_multi_kernel(EwMultiDev q) {
    const TcEwSeg& g = q.s[blockIdx.y];
    if (g.kind == TC_EW_PATCHIFY) patchify_body<T>((const T*)g.a, g.sa, g.lda, (T*)g.b, g.n0, g.n1, g.n2, g.n3, g.n4, g.flag);
    else if (g.kind == TC_EW_DEINTERLEAVE) sr_deinterleave_body<T>((const T*)g.a, (T*)g.b, g.sb, g.ldb, g.n0, g.n1, g.n2, g.n3, g.flag);
    else copy3d_body<T>((const T*)g.a, g.sa, g.lda, (T*)g.b, g.sb, g.ldb, g.n0, g.n1, g.n2, g.flag);
}

template <typename T>
__global__ void stem_im2col_kernel(const T* img, T* cols, int ldc, int B, int in_ch, int H, int W, int Ho, int Wo) {
    const long long n = (long long)B * Ho * Wo * ldc;
    TC_GRID_STRIDE(i, n) {
        const int col = (int)(i % ldc); unsigned t = i / ldc;
        const int ow = (int)(t % Wo); t /= Wo; const int oh = (int)(t % Ho); const int b = (int)(t / Ho);
        float v = 0.f;
        if (col < 147) {
            const int ci = col / 49, ky = (col % 49) / 7, kx = col % 7;
            const int ih = oh * 4 + ky - 3, iw = ow * 4 + kx - 3;
            if (ih >= 0 && ih < H && iw >= 0 && iw < W)
                v = ldf<T>(img + (((long long)b * in_ch + (in_ch == 1 ? 0 : ci)) * H + ih) * W + iw);
        }
        stf<T>(cols + i, v);
    }
}

// im2col / col2im of a 3x3, stride-2, pad-1 convolution (the Conv2d_BN stem of MSViT_4Stages, MSTr.py:1793-1810): column c * 9 + ky * 3 + kx of
// row (b, oy, ox) is x(b, 2 oy + ky - 1, 2 ox + kx - 1, c), zero outside the map -- the order of a [Cout, Cin, 3, 3] weight's rows.
// nchw: x is an image [B, src_ch, H, W] (src_ch = 1 feeds all Cin channels: the reference's x.repeat(1, 3, 1, 1)); else token-major rows.
template <typename T>
__global__ void im2col3s2_kernel(const T* x, int ldx, int nchw, int src_ch, T* cols, int ldc, int B, int Cin, int H, int W, int Ho, int Wo) {
    const long long n = (long long)B * Ho * Wo * ldc;
    TC_GRID_STRIDE(i, n) {
        const int col = (int)(i % ldc); unsigned t = i / ldc;
        const int ox = (int)(t % Wo); t /= Wo; const int oy = (int)(t % Ho); const int b = (int)(t / Ho);
        float v = 0.f;
        if (col < 9 * Cin) {
            const int c = col / 9, ky = (col % 9) / 3, kx = col % 3;
            const int iy = 2 * oy + ky - 1, ix = 2 * ox + kx - 1;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W)
                v = nchw ? ldf<T>(x + (((long long)b * src_ch + (src_ch == 1 ? 0 : c)) * H + iy) * W + ix)
                         : ldf<T>(x + ((long long)(b * H + iy) * W + ix) * ldx + c);
        }
        stf<T>(cols + i, v);
    }
}
// dx(b, y, x, c) (+)= the sum of the column gradients of every (output pixel, tap) that read it
template <typename T>
__global__ void col2im3s2_kernel(const T* dcols, int ldc, T* dx, int lddx, int B, int Cin, int H, int W, int Ho, int Wo, int accumulate) {
    const long long n = (long long)B * H * W * Cin;
    TC_GRID_STRIDE(i, n) {
        const int c = (int)(i % Cin); unsigned t = i / Cin;
        const int x_ = (int)(t % W); t /= W; const int y = (int)(t % H); const int b = (int)(t / H);
        float a = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int sy = y + 1 - ky;
            if (sy < 0 || (sy & 1) || (sy >> 1) >= Ho) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int sx = x_ + 1 - kx;
                if (sx < 0 || (sx & 1) || (sx >> 1) >= Wo) continue;
                a += ldf<T>(dcols + ((long long)(b * Ho + (sy >> 1)) * Wo + (sx >> 1)) * ldc + c * 9 + ky * 3 + kx);
            }
        }
        T* d = dx + ((long long)(b * H + y) * W + x_) * lddx + c;
        stf<T>(d, accumulate ? ldf<T>(d) + a : a);
    }
}

// Window partition of SpatialAwareTrans (have_bridge = "sp", MSTr.py:2627-2640, 2649-2658): pixel (b, wy ws + iy, wx ws + ix) of a [B, H, W, C]
// token map is token off + iy ws + ix of window (b, wy, wx) in a [B (H/ws) (W/ws), ntw, C] window-token matrix.  dir 0: windows <- map;
// dir 1: map (+)= windows (the reverse step, and each direction's gradient).  Eight channels per thread.
template <typename T>
__global__ void window_rows_kernel(const T* src, int lds, T* dst, int ldd, int B, int H, int W, int ws, int ntw, int off, int C8, int dir, int accumulate) {
    const long long n = (long long)B * H * W * C8;
    const int nbx = W / ws, nby = H / ws;
    TC_GRID_STRIDE(i, n) {
        const int c8 = (int)(i % C8); unsigned t = i / C8;
        const int x_ = (int)(t % W); t /= W; const int y = (int)(t % H); const int b = (int)(t / H);
        const long long rmap = ((long long)b * H + y) * W + x_;
        const long long rwin = (((long long)b * nby + y / ws) * nbx + x_ / ws) * ntw + off + (y % ws) * ws + x_ % ws;
        const T* s_ = src + (dir ? rwin : rmap) * lds + c8 * 8;
        T* d_ = dst + (dir ? rmap : rwin) * ldd + c8 * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) stf<T>(d_ + e, accumulate ? ldf<T>(d_ + e) + ldf<T>(s_ + e) : ldf<T>(s_ + e));
    }
}
// Dropout (MLP_FFN of the "sp" bridge, MSTr.py:70,75-77): y = x * keep / (1 - p), keep from a counter-based generator keyed by (*seed + salt,
// element index) -- the backward calls it on the gradient with the same key.  No reference counterpart for the bits (torch's CPU generator).
template <typename T>
__global__ void dropout_kernel(const T* x, T* y, long long n, float p, const long long* seed, unsigned salt) {
    const unsigned key = (unsigned)(*seed) * 0x9E3779B9u + salt * 0x85EBCA6Bu;
    const float scale = 1.0f / (1.0f - p);
    TC_GRID_STRIDE64(i, n) {
        unsigned v = (unsigned)i ^ ((unsigned)(i >> 32) * 0xC2B2AE35u) ^ key;
        v ^= v >> 16; v *= 0x7FEB352Du; v ^= v >> 15; v *= 0x846CA68Bu; v ^= v >> 16;
        const bool keep = (float)v * 2.3283064365386963e-10f >= p;
        stf<T>(y + i, keep ? ldf<T>(x + i) * scale : 0.f);
    }
}

template <typename TS, typename TD>
__global__ void cast_kernel(const TS* s, TD* d, long long n) { TC_GRID_STRIDE64(i, n) stf<TD>(d + i, ldf<TS>(s + i)); }

inline dim3 g1(long long n) { return n < 0x7fffffffLL ? dim3(tc_blocks(n, 256, 8192)) : dim3(0); }     // (an empty grid is a launch error: reported)
inline dim3 g1_64(long long n) { return dim3(tc_blocks(n, 256, 8192)); }                                // kernels on the 64-bit TC_GRID_STRIDE64 loop (cast, sigmoid')

}  // namespace

#define TC_S ((hipStream_t)stream)

extern "C" int tc_fma3_fwd(const void* a, int lda, const void* b, int ldb, const void* c, int ldc, void* out, int ldo, int rows,
                           int cols, float alpha, int dtype, void* stream) {
    if (!a || !b || !c || !out || rows <= 0 || cols <= 0 || ((cols | lda | ldb | ldc | ldo) & 3)) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((fma3_fwd_kernel<T>), g1((long long)rows * cols / 4), dim3(256), 0, TC_S, (const T*)a,
                                                lda, (const T*)b, ldb, (const T*)c, ldc, (T*)out, ldo, rows, cols / 4, alpha));
    return tc_launch_status();
}
extern "C" int tc_fma3_bwd(const void* dout, int lddo, const void* b, int ldb, const void* c, int ldc, void* da, int ldda, void* db,
                           int lddb, int db_accumulate, void* dc, int lddc, int rows, int cols, float alpha, int dtype, void* stream) {
    if (!dout || !b || !c || !da || !db || !dc || rows <= 0 || cols <= 0 || ((cols | lddo | ldb | ldc | ldda | lddb | lddc) & 3))
        return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((fma3_bwd_kernel<T>), g1((long long)rows * cols / 4), dim3(256), 0, TC_S,
                                                (const T*)dout, lddo, (const T*)b, ldb, (const T*)c, ldc, (T*)da, ldda, (T*)db, lddb,
                                                db_accumulate, (T*)dc, lddc, rows, cols / 4, alpha));
    return tc_launch_status();
}
extern "C" int tc_add(const void* a, int lda, const void* b, int ldb, void* y, int ldy, int rows, int cols, int dtype, void* stream) {
    if (!a || !b || !y || rows <= 0 || cols <= 0 || ((cols | lda | ldb | ldy) & 3)) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((add_kernel<T>), g1((long long)rows * cols / 4), dim3(256), 0, TC_S, (const T*)a, lda,
                                                (const T*)b, ldb, (T*)y, ldy, rows, cols / 4));
    return tc_launch_status();
}
extern "C" int tc_sigmoid_bwd(const void* dy, const void* s, void* dz, long long n, int dtype, void* stream) {
    if (!dy || !s || !dz || n <= 0) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((sigmoid_bwd_kernel<T>), g1_64(n), dim3(256), 0, TC_S, (const T*)dy, (const T*)s, (T*)dz, n));
    return tc_launch_status();
}
extern "C" int tc_copy3d(const void* src, long long sbs, int lds, void* dst, long long sbd, int ldd, int nb, int rows, int cols,
                         int accumulate, int dtype, void* stream) {
    if (!src || !dst || nb <= 0 || rows <= 0 || cols <= 0) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((copy3d_kernel<T>), g1((long long)nb * rows * cols), dim3(256), 0, TC_S, (const T*)src,
                                                sbs, lds, (T*)dst, sbd, ldd, nb, rows, cols, accumulate));
    return tc_launch_status();
}
extern "C" int tc_transpose(const void* src, void* dst, int nb, int R, int Cc, int dtype, void* stream) {
    if (!src || !dst || nb <= 0 || R <= 0 || Cc <= 0 || nb > 65535) return TC_ERR_ARG;
    dim3 grid((Cc + 31) / 32, (R + 31) / 32, nb);
    if (grid.y > 65535) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((transpose_kernel<T>), grid, dim3(256), 0, TC_S, (const T*)src, (T*)dst, R, Cc));
    return tc_launch_status();
}
extern "C" int tc_chan_pool_fwd(const void* x, int ldx, void* pooled, int B, int N, int C, int dtype, void* stream) {
    if (!x || !pooled || B <= 0 || N <= 0 || C <= 0 || ((C | ldx) & 3)) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((chan_pool_fwd_kernel<T>), dim3(B, (C + 63) / 64), dim3(256), 0, TC_S, (const T*)x, ldx, (T*)pooled, N, C));
    return tc_launch_status();
}
extern "C" int tc_chan_pool_bwd(const void* dpooled, void* dx, int lddx, int B, int N, int C, int accumulate, int dtype, void* stream) {
    if (!dpooled || !dx || B <= 0 || N <= 0 || C <= 0 || ((C | lddx) & 3)) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((chan_pool_bwd_kernel<T>), g1((long long)B * N * C / 4), dim3(256), 0, TC_S, (const T*)dpooled,
                                                (T*)dx, lddx, B, N, C, accumulate));
    return tc_launch_status();
}
extern "C" int tc_chan_gate_fwd(const void* x, int ldx, const void* gate, void* y, int ldy, int B, int N, int C, int dtype, void* stream) {
    if (!x || !gate || !y || B <= 0 || N <= 0 || C <= 0 || ((C | ldx | ldy) & 3)) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((chan_gate_fwd_kernel<T>), g1((long long)B * N * C / 4), dim3(256), 0, TC_S, (const T*)x, ldx,
                                                (const T*)gate, (T*)y, ldy, B, N, C));
    return tc_launch_status();
}
extern "C" int tc_chan_gate_bwd(const void* dy, int lddy, const void* x, int ldx, const void* gate, void* dx, int lddx, int dx_accumulate,
                                void* dgate, int B, int N, int C, int dtype, void* stream) {
    if (!dy || !x || !gate || !dx || !dgate || B <= 0 || N <= 0 || C <= 0 || ((C | lddy | ldx | lddx) & 3)) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((chan_gate_bwd_kernel<T>), dim3(B, (C + 63) / 64), dim3(256), 0, TC_S, (const T*)dy, lddy,
                                                (const T*)x, ldx, (const T*)gate, (T*)dx, lddx, dx_accumulate, (T*)dgate, N, C));
    return tc_launch_status();
}
extern "C" int tc_relu_fwd(const void* x, void* y, long long n, int dtype, void* stream) {
    if (!x || !y || n <= 0) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((relu_fwd_kernel<T>), g1_64(n), dim3(256), 0, TC_S, (const T*)x, (T*)y, n));
    return tc_launch_status();
}
extern "C" int tc_relu_bwd(const void* dy, const void* y, void* dz, long long n, int dtype, void* stream) {
    if (!dy || !y || !dz || n <= 0) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((relu_bwd_kernel<T>), g1_64(n), dim3(256), 0, TC_S, (const T*)dy, (const T*)y, (T*)dz, n));
    return tc_launch_status();
}
extern "C" int tc_coord_pool_fwd(const void* x, void* pooled, int B, int H, int W, int C, int dtype, void* stream) {
    if (!x || !pooled || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((coord_pool_fwd_kernel<T>), g1((long long)B * (H + W) * C / 4), dim3(256), 0, TC_S,
                                                (const T*)x, (T*)pooled, B, H, W, C));
    return tc_launch_status();
}
extern "C" int tc_coord_pool_bwd(const void* dpooled, void* dx, int B, int H, int W, int C, int accumulate, int dtype, void* stream) {
    if (!dpooled || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((coord_pool_bwd_kernel<T>), g1((long long)B * H * W * C / 4), dim3(256), 0, TC_S,
                                                (const T*)dpooled, (T*)dx, B, H, W, C, accumulate));
    return tc_launch_status();
}
extern "C" int tc_coord_gate_fwd(const void* x, const void* att, void* y, int B, int H, int W, int C, int dtype, void* stream) {
    if (!x || !att || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((coord_gate_fwd_kernel<T>), g1((long long)B * H * W * C / 4), dim3(256), 0, TC_S,
                                                (const T*)x, (const T*)att, (T*)y, B, H, W, C));
    return tc_launch_status();
}
extern "C" int tc_coord_gate_bwd(const void* dy, const void* x, const void* att, void* dx, int dx_accumulate, void* datt, int B, int H,
                                 int W, int C, int dtype, void* stream) {
    if (!dy || !x || !att || !dx || !datt || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, {
        hipLaunchKernelGGL((coord_gate_bwd_dx_kernel<T>), g1((long long)B * H * W * C / 4), dim3(256), 0, TC_S, (const T*)dy,
                           (const T*)att, (T*)dx, dx_accumulate, B, H, W, C);
        hipLaunchKernelGGL((coord_gate_bwd_att_kernel<T>), g1((long long)B * (H + W) * C / 4), dim3(256), 0, TC_S, (const T*)dy,
                           (const T*)x, (const T*)att, (T*)datt, B, H, W, C);
    });
    return tc_launch_status();
}
extern "C" int tc_pixel_shuffle(const void* in, void* out, int B, int H, int W, int p, int c, int inverse, int dtype, void* stream) {
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || p <= 0 || c <= 0 || (c & 3)) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((pixel_shuffle_kernel<T>), g1((long long)B * H * W * p * p * c / 4), dim3(256), 0, TC_S,
                                                (const T*)in, (T*)out, B, H, W, p, c, inverse));
    return tc_launch_status();
}
extern "C" int tc_patchify(const void* map, long long sb_map, int ld_map, void* cols, int B, int H, int W, int C, int k, int inverse,
                           int dtype, void* stream) {
    if (!map || !cols || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || k <= 0 || H % k || W % k || (ld_map & 3) || (sb_map & 3))
        return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((patchify_kernel<T>), g1((long long)B * H * W * C / 4), dim3(256), 0, TC_S, (const T*)map,
                                                sb_map, ld_map, (T*)cols, B, H, W, C, k, inverse));
    return tc_launch_status();
}
extern "C" int tc_sr_deinterleave(const void* in, void* out, long long sbo, int ldo, int B, int P, int C, int mult, int inverse, int dtype,
                                  void* stream) {
    if (!in || !out || B <= 0 || P <= 0 || C <= 0 || mult <= 0) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((sr_deinterleave_kernel<T>), g1((long long)B * P * C * mult), dim3(256), 0, TC_S,
                                                (const T*)in, (T*)out, sbo, ldo, B, P, C, mult, inverse));
    return tc_launch_status();
}
extern "C" int tc_ew_multi(const TcEwSeg* segs, int nseg, int dtype, void* stream) {
    if (!segs || nseg < 1 || nseg > TC_EW_MULTI_MAX) return TC_ERR_ARG;
    EwMultiDev q;
    long long nmax = 0;
    for (int i = 0; i < nseg; ++i) {
        const TcEwSeg& g = segs[i];
        if (!g.a || !g.b) return TC_ERR_ARG;
        long long n;
        if (g.kind == TC_EW_PATCHIFY) {                          // tc_patchify's conditions
            if (g.n0 <= 0 || g.n1 <= 0 || g.n2 <= 0 || g.n3 <= 0 || (g.n3 & 3) || g.n4 <= 0 || g.n1 % g.n4 || g.n2 % g.n4 || (g.lda & 3) || (g.sa & 3)) return TC_ERR_ARG;
            n = (long long)g.n0 * g.n1 * g.n2 * g.n3 / 4;
        } else if (g.kind == TC_EW_DEINTERLEAVE) {
            if (g.n0 <= 0 || g.n1 <= 0 || g.n2 <= 0 || g.n3 <= 0) return TC_ERR_ARG;
            n = (long long)g.n0 * g.n1 * g.n2 * g.n3;
        } else if (g.kind == TC_EW_COPY) {
            if (g.n0 <= 0 || g.n1 <= 0 || g.n2 <= 0) return TC_ERR_ARG;
            n = (long long)g.n0 * g.n1 * g.n2;
        } else return TC_ERR_ARG;
        nmax = n > nmax ? n : nmax;
        q.s[i] = g;
    }
    dim3 grid = g1(nmax);
    grid.y = nseg;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ew_multi_kernel<T>), grid, dim3(256), 0, TC_S, q));
    return tc_launch_status();
}
extern "C" int tc_stem_im2col(const void* img, void* cols, int ldc, int B, int in_ch, int H, int W, int dtype, void* stream) {
    if (!img || !cols || B <= 0 || (in_ch != 1 && in_ch != 3) || H <= 0 || W <= 0 || ldc < 147) return TC_ERR_ARG;
    const int Ho = (H + 6 - 7) / 4 + 1, Wo = (W + 6 - 7) / 4 + 1;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((stem_im2col_kernel<T>), g1((long long)B * Ho * Wo * ldc), dim3(256), 0, TC_S, (const T*)img,
                                                (T*)cols, ldc, B, in_ch, H, W, Ho, Wo));
    return tc_launch_status();
}
extern "C" int tc_window_rows(const void* src, int lds, void* dst, int ldd, int B, int H, int W, int ws, int ntw, int off, int C, int dir,
                              int accumulate, int dtype, void* stream) {
    if (!src || !dst || B <= 0 || H <= 0 || W <= 0 || ws <= 0 || H % ws || W % ws || C <= 0 || (C & 7) || off < 0 || off + ws * ws > ntw || lds < C || ldd < C)
        return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((window_rows_kernel<T>), g1((long long)B * H * W * (C / 8)), dim3(256), 0, TC_S, (const T*)src, lds,
                                                (T*)dst, ldd, B, H, W, ws, ntw, off, C / 8, dir, accumulate));
    return tc_launch_status();
}
extern "C" int tc_dropout(const void* x, void* y, long long n, float p, const long long* seed_dev, unsigned salt, int dtype, void* stream) {
    if (!x || !y || n <= 0 || !(p >= 0.f && p < 1.f) || !seed_dev) return TC_ERR_ARG;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dropout_kernel<T>), g1(n), dim3(256), 0, TC_S, (const T*)x, (T*)y, n, p, seed_dev, salt));
    return tc_launch_status();
}
extern "C" int tc_im2col3s2(const void* x, int ldx, int nchw, int src_ch, void* cols, int ldc, int B, int Cin, int H, int W, int dtype, void* stream) {
    if (!x || !cols || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || ldc < 9 * Cin || (nchw ? (src_ch != 1 && src_ch != Cin) : ldx < Cin)) return TC_ERR_ARG;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((im2col3s2_kernel<T>), g1((long long)B * Ho * Wo * ldc), dim3(256), 0, TC_S, (const T*)x, ldx, nchw,
                                                src_ch, (T*)cols, ldc, B, Cin, H, W, Ho, Wo));
    return tc_launch_status();
}
extern "C" int tc_col2im3s2(const void* dcols, int ldc, void* dx, int lddx, int B, int Cin, int H, int W, int accumulate, int dtype, void* stream) {
    if (!dcols || !dx || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || ldc < 9 * Cin || lddx < Cin) return TC_ERR_ARG;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((col2im3s2_kernel<T>), g1((long long)B * H * W * Cin), dim3(256), 0, TC_S, (const T*)dcols, ldc,
                                                (T*)dx, lddx, B, Cin, H, W, Ho, Wo, accumulate));
    return tc_launch_status();
}
extern "C" int tc_cast(const void* src, void* dst, long long n, int src_dtype, int dst_dtype, void* stream) {
    if (!src || !dst || n <= 0) return TC_ERR_ARG;
    if (src_dtype == TC_F32 && dst_dtype == TC_BF16)
        hipLaunchKernelGGL((cast_kernel<float, bf16_t>), g1_64(n), dim3(256), 0, TC_S, (const float*)src, (bf16_t*)dst, n);
    else if (src_dtype == TC_BF16 && dst_dtype == TC_F32)
        hipLaunchKernelGGL((cast_kernel<bf16_t, float>), g1_64(n), dim3(256), 0, TC_S, (const bf16_t*)src, (float*)dst, n);
    else if (src_dtype == TC_F32 && dst_dtype == TC_F16)
        hipLaunchKernelGGL((cast_kernel<float, f16_t>), g1_64(n), dim3(256), 0, TC_S, (const float*)src, (f16_t*)dst, n);
    else if (src_dtype == TC_F16 && dst_dtype == TC_F32)
        hipLaunchKernelGGL((cast_kernel<f16_t, float>), g1_64(n), dim3(256), 0, TC_S, (const f16_t*)src, (float*)dst, n);
    else return TC_ERR_ARG;
    return tc_launch_status();
}
// A named cut in a kernel trace: an empty one-wave launch whose only purpose is to show up (with its id as the grid's x extent) between the
// launches of two sections of a step (engine.Graph.segment, scripts/seg_timeline.py).  Never launched unless TC_SEG_MARKS=1.
__global__ void seg_marker_kernel(int id) { (void)id; }
extern "C" int tc_seg_marker(int id, void* stream) {
    if (id < 0) return TC_ERR_ARG;
    hipLaunchKernelGGL(seg_marker_kernel, dim3(id + 1), dim3(64), 0, TC_S, id);
    return tc_launch_status();
}
extern "C" int tc_abi_version(void) { return 15; }
