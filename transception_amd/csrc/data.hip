// Input pipeline on the device: what datasets/dataset_synapse.py:101-112 and the trainer's
// transforms (trainer.py:89-93) do to one training slice on the host -- augment, scipy.ndimage.zoom order 3 / order 0 to the
// network size, Normalize(0.5, 0.5) -- done for a whole batch of raw 512x512 slices resident in HBM:
//
//   slice_augment_kernel        geometric warp (affine + 4x4 control-point displacement, order 1 image / order 0 label, cval 0)
//                               -> Gaussian blur sigma 1 (9 taps, mirror) -> linear contrast -> additive Gaussian noise,
//                               one 32x32 output tile (+4 halo when blurring) per workgroup, staged in LDS
//   spline_prefilter_cols/rows  cubic B-spline coefficients of the slice, mirror boundary, fp64 like scipy (ni_splines.c):
//                               columns in 32-row segments with a 16-row run-in (lanes along x, coalesced), rows one wavefront
//                               per row as shuffle scans; simple one-thread-per-line kernels for ragged sizes
//   zoom_sample_kernel          4x4-tap B-spline evaluation at o*(in-1)/(out-1), nearest label, normalise, int64 labels
//
// All of it is HBM-streaming work (16 slices = 16 MB in, 3 MB out); none of it is GEMM-shaped.  Coordinates are computed in
// fp64 without contraction so that the nearest-neighbour decisions and scipy's "coordinate rounds above n-1 -> cval" quirk
// (the last output row/column at 512 -> 224) come out exactly as on the host.
#include "tc_common.h"

namespace {

constexpr double kPole = -0.26794919243112270647;            // sqrt(3) - 2
constexpr int kBlurR = 4;
constexpr int kTile = 32;
#ifndef TC_AUG_THREADS
#define TC_AUG_THREADS 256
#endif
constexpr int kAugThreads = TC_AUG_THREADS;     // (1024 = a thread per pixel of the tile measured no faster: 163 against 158 us for the four rounds + zoom of a sampled batch)

__device__ __forceinline__ int mirror101(int i, int n) {       // scipy 'mirror' / cv2 BORDER_REFLECT_101
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return min(max(i, 0), n - 1);                              // (only halo positions no output needs get clamped)
}

__device__ __forceinline__ unsigned mix32(unsigned v) {
    v ^= v >> 16; v *= 0x7FEB352Du; v ^= v >> 15; v *= 0x846CA68Bu; v ^= v >> 16;
    return v;
}

#pragma clang fp contract(off)
__device__ __forceinline__ void aug_source(const TcSliceAug& A, int oy, int ox, int H, int W, double& sy, double& sx) {
    double qy = (double)oy, qx = (double)ox;
    if (A.flags & TC_AUG_PIECEWISE) {
        // imgaug PiecewiseAffine = skimage PiecewiseAffineTransform: the control grid spans [0, H] x [0, W] (linspace(0, size, 4)), every cell is
        // two triangles (which diagonal: bit 8 + cell of flags, from the host's Delaunay triangulation), and inside a triangle the map is
        // affine: the pixel moves by the barycentric mix of its three corners' displacements
        const double gy = qy * (3.0 / (double)H), gx = qx * (3.0 / (double)W);
        const int y0 = min((int)floor(gy), 2), x0 = min((int)floor(gx), 2);
        const double fy = gy - (double)y0, fx = gx - (double)x0;
        const float* d = A.disp + (y0 * 4 + x0) * 2;                       // TL d[0..1], TR d[2..3], BL d[8..9], BR d[10..11]
        double l00 = 0.0, l01 = 0.0, l10 = 0.0, l11 = 0.0;                 // weights of TL, TR, BL, BR
        if (!((A.flags >> (8 + y0 * 3 + x0)) & 1)) {                      // diagonal TL - BR
            if (fx >= fy) { l00 = 1.0 - fx; l01 = fx - fy; l11 = fy; }
            else { l00 = 1.0 - fy; l10 = fy - fx; l11 = fx; }
        } else {                                                          // diagonal TR - BL
            if (fx + fy <= 1.0) { l00 = (1.0 - fx) - fy; l01 = fx; l10 = fy; }
            else { l11 = (fx + fy) - 1.0; l01 = 1.0 - fy; l10 = 1.0 - fx; }
        }
        const double dy = ((l00 * (double)d[0] + l01 * (double)d[2]) + l10 * (double)d[8]) + l11 * (double)d[10];
        const double dx = ((l00 * (double)d[1] + l01 * (double)d[3]) + l10 * (double)d[9]) + l11 * (double)d[11];
        qy = qy + dy; qx = qx + dx;
    }
    sy = (A.m[2] + A.m[0] * qy) + A.m[1] * qx;
    sx = (A.m[5] + A.m[3] * qy) + A.m[4] * qx;
}

#pragma clang fp contract(off)
__device__ __forceinline__ float aug_image_at(const float* img, const TcSliceAug& A, int oy, int ox, int H, int W) {
    if (!(A.flags & TC_AUG_WARP)) return img[(long long)oy * W + ox];
    double sy, sx;
    aug_source(A, oy, ox, H, W, sy, sx);
    if (!(sy >= 0.0 && sy <= (double)(H - 1) && sx >= 0.0 && sx <= (double)(W - 1))) return 0.f;
    if (A.flags & TC_AUG_LINEAR) {
        const int y0 = min((int)floor(sy), H - 2), x0 = min((int)floor(sx), W - 2);
        const double fy = sy - (double)y0, fx = sx - (double)x0;
        const float* p = img + (long long)y0 * W + x0;
        const double v = (((1.0 - fy) * (1.0 - fx) * (double)p[0] + (1.0 - fy) * fx * (double)p[1]) + fy * (1.0 - fx) * (double)p[W])
                         + fy * fx * (double)p[W + 1];
        return (float)v;
    }
    const int iy = min(max((int)floor(sy + 0.5), 0), H - 1), ix = min(max((int)floor(sx + 0.5), 0), W - 1);
    return img[(long long)iy * W + ix];
}

#pragma clang fp contract(off)
__device__ __forceinline__ unsigned char aug_label_at(const unsigned char* lab, const TcSliceAug& A, int oy, int ox, int H, int W) {
    if (!(A.flags & TC_AUG_WARP)) return lab[(long long)oy * W + ox];
    double sy, sx;
    aug_source(A, oy, ox, H, W, sy, sx);
    if (!(sy >= 0.0 && sy <= (double)(H - 1) && sx >= 0.0 && sx <= (double)(W - 1))) return 0;
    const int iy = min(max((int)floor(sy + 0.5), 0), H - 1), ix = min(max((int)floor(sx + 0.5), 0), W - 1);
    return lab[(long long)iy * W + ix];
}

__global__ __launch_bounds__(kAugThreads) void slice_augment_kernel(const float* __restrict__ img, const unsigned char* __restrict__ lab,
                                                            const float* __restrict__ raw_img, const unsigned char* __restrict__ raw_lab,
                                                            const TcSliceAug* __restrict__ aug, float* __restrict__ oimg,
                                                            unsigned char* __restrict__ olab, int H, int W) {
    constexpr int TS = kTile + 2 * kBlurR;
    __shared__ float t0[TS * TS];
    __shared__ float t1[kTile * TS];
    __shared__ TcSliceAug A;
    const int b = blockIdx.z, tid = threadIdx.x;
    for (int i = tid; i < (int)(sizeof(TcSliceAug) / 4); i += kAugThreads) reinterpret_cast<unsigned*>(&A)[i] = reinterpret_cast<const unsigned*>(aug + b)[i];
    __syncthreads();
    // rounds of a chain (tc_slice_augment_chain): a slice with nothing to do this round and nothing to deliver leaves at once; the first
    // stage of a slice's chain reads the raw slice, later ones the previous round's buffer
    if (A.flags & TC_AUG_SKIP) return;
    const bool from_raw = A.flags & TC_AUG_FROM_RAW;
    const float* im = (from_raw ? raw_img : img) + (long long)b * H * W;
    const unsigned char* lb = (from_raw ? raw_lab : lab) + (long long)b * H * W;
    const int ty0 = blockIdx.y * kTile, tx0 = blockIdx.x * kTile;
    // An identity record (a slice without stages on its way to the result buffer, or the plain tc_slice_augment entry given one) is copied in
    // 16-byte pieces: the general path below moves a pixel per thread and step and made a copy cost what a warp costs.  (alpha 1 about
    // center 0 and sigma 0 leave every value bit for bit.)
    if (!(A.flags & (TC_AUG_WARP | TC_AUG_BLUR)) && A.alpha == 1.0f && A.center == 0.0f && !(A.noise_sigma > 0.0f) && !(W & 3) && tx0 + kTile <= W) {
        const int ly = tid >> 3, lx = (tid & 7) * 4, oy = ty0 + ly;
        if (tid < 256 && oy < H) {
            const long long o = ((long long)b * H + oy) * W + tx0 + lx, oi = (long long)oy * W + tx0 + lx;
            *reinterpret_cast<float4*>(oimg + o) = *reinterpret_cast<const float4*>(im + oi);
            *reinterpret_cast<uchar4*>(olab + o) = *reinterpret_cast<const uchar4*>(lb + oi);
        }
        return;
    }
    const bool blur = A.flags & TC_AUG_BLUR;
    // Pixel stages run in the order the sampler drew them (A.reserved: up to three 2-bit codes, first stage in the low bits;
    // 1 = blur, 2 = contrast, 3 = noise; 0 = the canonical blur -> contrast -> noise).  Blur is the only non-pointwise stage: the
    // pointwise stages drawn before it are applied while the haloed tile is filled, the ones after it to the blurred value.
    int ord = A.reserved & 63;
    if (ord == 0) ord = 1 | (2 << 2) | (3 << 4);
    int pre = 0, post = 0, npre = 0, npost = 0;
    {
        bool seen_blur = !blur;
        for (int i = 0; i < 3; ++i) {
            const int c = (ord >> (2 * i)) & 3;
            if (c == 1) { seen_blur = true; continue; }
            if (c == 0) continue;
            if (seen_blur) { post |= c << (2 * npost); ++npost; } else { pre |= c << (2 * npre); ++npre; }
        }
    }
    auto pointwise = [&](float v, int oy, int ox, int codes, int n) {
        for (int i = 0; i < n; ++i) {
            const int c = (codes >> (2 * i)) & 3;
            if (c == 2) v = A.center + A.alpha * (v - A.center);
            else if (c == 3 && A.noise_sigma > 0.f) {
                const unsigned idx = (unsigned)(oy * W + ox), base = A.noise_seed * 0x9E3779B9u;
                const unsigned ra = mix32(idx * 2u + base), rb = mix32(idx * 2u + 1u + base + 0x85EBCA6Bu);
                const float u1 = ((float)ra + 1.0f) * 2.3283064365386963e-10f, u2 = (float)rb * 2.3283064365386963e-10f;
                v += A.noise_sigma * (sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2));
            }
        }
        return v;
    };
    float g[kBlurR + 1];
    if (blur) {
        double w[kBlurR + 1], s = 1.0;
        w[0] = 1.0;
        for (int k = 1; k <= kBlurR; ++k) { w[k] = exp(-0.5 * (double)(k * k)); s += 2.0 * w[k]; }
        for (int k = 0; k <= kBlurR; ++k) g[k] = (float)(w[k] / s);
        // warped slice over the tile + halo; positions beyond the slice mirror back into it
        for (int p = tid; p < TS * TS; p += kAugThreads) {
            const int ly = p / TS, lx = p % TS;
            const int my = mirror101(ty0 + ly - kBlurR, H), mx = mirror101(tx0 + lx - kBlurR, W);
            t0[p] = pointwise(aug_image_at(im, A, my, mx, H, W), my, mx, pre, npre);
        }
        __syncthreads();
        for (int p = tid; p < kTile * TS; p += kAugThreads) {             // axis 0 first (scipy.ndimage.gaussian_filter order), fp32 intermediate
            const int ly = p / TS, lx = p % TS;
            double a = (double)g[0] * (double)t0[(ly + kBlurR) * TS + lx];
            for (int k = 1; k <= kBlurR; ++k) a += (double)g[k] * ((double)t0[(ly + kBlurR + k) * TS + lx] + (double)t0[(ly + kBlurR - k) * TS + lx]);
            t1[p] = (float)a;
        }
        __syncthreads();
    }
    for (int p = tid; p < kTile * kTile; p += kAugThreads) {
        const int ly = p / kTile, lx = p % kTile, oy = ty0 + ly, ox = tx0 + lx;
        if (oy >= H || ox >= W) continue;
        float v;
        if (blur) {
            double a = (double)g[0] * (double)t1[ly * TS + lx + kBlurR];
            for (int k = 1; k <= kBlurR; ++k) a += (double)g[k] * ((double)t1[ly * TS + lx + kBlurR + k] + (double)t1[ly * TS + lx + kBlurR - k]);
            v = (float)a;
        } else {
            v = aug_image_at(im, A, oy, ox, H, W);
        }
        v = pointwise(v, oy, ox, post, npost);
        const long long o = ((long long)b * H + oy) * W + ox;
        oimg[o] = v;
        olab[o] = aug_label_at(lb, A, oy, ox, H, W);
    }
}

// ---------------------------------------------------------------------------------------------- cubic B-spline prefilter
// One line of n samples: c *= 6; c[0] = exact mirror sum; causal c[i] += z c[i-1]; c[n-1] = (z c[n-2] + c[n-1]) z / (z^2 - 1);
// anti-causal c[i] = z (c[i+1] - c[i]).
// Columns: one thread per column, lanes along x (coalesced).  The recursion is a chain of dependent FMAs, so what matters is
// that the loads do not sit in that chain: every phase fetches 16 rows ahead into registers, then runs its 16 steps.
__global__ __launch_bounds__(256) void spline_prefilter_cols_kernel(const float* __restrict__ img, double* __restrict__ coef, int B, int H, int W) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)B * W) return;
    const int b = (int)(t / W), x = (int)(t % W);
    const float* src = img + (long long)b * H * W + x;
    double* c = coef + (long long)b * H * W + x;
    if (H == 1) { c[0] = (double)src[0]; return; }
    constexpr int U = 16;
    const double z = kPole, zn1 = pow(z, (double)(H - 1));
    // c0 = 6 s[0] + zn1 6 s[H-1] + sum_{i=1}^{H-2} z^i (6 s[i] + zn1 6 s[H-1-i]); |z|^64 ~ 1e-37: 64 terms are exact in fp64, and
    // the zn1 terms only exist for H <= 64
    const bool shortcol = H <= 64;
    double c0 = 6.0 * (double)src[0] + (shortcol ? zn1 * 6.0 * (double)src[(long long)(H - 1) * W] : 0.0), zi = z;
    const int nterm = min(H - 1, 64);
    for (int i0 = 1; i0 < nterm; i0 += U) {
        float v[U], m[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int i = i0 + k;
            v[k] = i < nterm ? src[(long long)i * W] : 0.f;
            m[k] = (shortcol && i < nterm) ? src[(long long)(H - 1 - i) * W] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            if (i0 + k < nterm) { c0 += zi * (6.0 * (double)v[k] + zn1 * 6.0 * (double)m[k]); zi *= z; }
        }
    }
    double prev = c0 / (1.0 - zn1 * zn1), prev2 = prev;
    c[0] = prev;
    for (int i0 = 1; i0 < H; i0 += U) {                                // causal: c[i] = 6 s[i] + z c[i-1]
        float v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = i0 + k < H ? src[(long long)(i0 + k) * W] : 0.f;
#pragma unroll
        for (int k = 0; k < U; ++k) {
            if (i0 + k < H) { prev2 = prev; prev = 6.0 * (double)v[k] + z * prev; c[(long long)(i0 + k) * W] = prev; }
        }
    }
    double nxt = (z * prev2 + prev) * z / (z * z - 1.0);               // prev2 = c[H-2], prev = c[H-1]
    c[(long long)(H - 1) * W] = nxt;
    for (int i0 = H - 2; i0 >= 0; i0 -= U) {                           // anti-causal: c[i] = z (c[i+1] - c[i])
        double v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = i0 - k >= 0 ? c[(long long)(i0 - k) * W] : 0.0;
#pragma unroll
        for (int k = 0; k < U; ++k) {
            if (i0 - k >= 0) { nxt = z * (nxt - v[k]); c[(long long)(i0 - k) * W] = nxt; }
        }
    }
}

// Columns of at least 128 rows, a multiple of 32: the pole's impulse response is down to |z|^16 = 7e-10 after 16 samples -- a hundredth of
// the float32 resolution of the zoomed slice these coefficients are for -- so a thread can produce 32 consecutive coefficients of a column
// from a 16-row run-in above and below them (with a 28-row run-in, 1e-16, it read 2.75 instead of 2 rows per row written: prefilter + zoom
// of sixteen 512 x 512 slices 98 -> 71 us) -- 16x the threads of the one-thread-per-column form, each with a short dependency chain.  The first segment starts from the exact mirror sum, the
// last one ends with the exact anti-causal initial value.
template <int L, int WU>
__global__ __launch_bounds__(256) void spline_prefilter_cols_seg_kernel(const float* __restrict__ img, double* __restrict__ coef, int B, int H, int W) {
    const int nseg = H / L;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)B * W * nseg) return;
    const int x = (int)(t % W), seg = (int)((t / W) % nseg), b = (int)(t / ((long long)W * nseg));
    const float* src = img + (long long)b * H * W + x;
    double* c = coef + (long long)b * H * W + x;
    const double z = kPole;
    const int s0 = seg * L, lo = s0 - WU;
    const bool first = seg == 0, last = seg == nseg - 1;
    double prev = 0.0;
    if (first) {                                                       // c0 = sum_{i < 64} z^i 6 s[i]
        double zi = 1.0;
#pragma unroll 1
        for (int i0 = 0; i0 < 64; i0 += 16) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = src[(long long)(i0 + i) * W];
#pragma unroll
            for (int i = 0; i < 16; ++i) { prev += zi * 6.0 * (double)v[i]; zi *= z; }
        }
    }
    double cp[L + WU];                                                 // causal values of rows s0 .. s0 + L + WU - 1
    constexpr int U = 16;
#pragma unroll
    for (int r0 = 0; r0 < L + 2 * WU; r0 += U) {
        float v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int row = lo + r0 + k;
            v[k] = (r0 + k < L + 2 * WU && row >= 0 && row < H) ? src[(long long)row * W] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int r = r0 + k, row = lo + r;
            if (r < L + 2 * WU) {
                if (row >= 0 && row < H && !(first && row == 0)) prev = 6.0 * (double)v[k] + z * prev;
                if (r >= WU) cp[r - WU] = prev;                        // (rows beyond H-1 of the last segment are never read back)
            }
        }
    }
    double nxt = 0.0;
#pragma unroll
    for (int r = L + WU - 1; r >= 0; --r) {
        if (last && r >= L) continue;
        if (last && r == L - 1) nxt = (z * cp[L - 2] + cp[L - 1]) * z / (z * z - 1.0);
        else nxt = z * (nxt - cp[r]);
        if (r < L) c[(long long)(s0 + r) * W] = nxt;
    }
}

// Rows whose length is a multiple of 64 (<= 1024): one wavefront per row, lane l holding elements 64 s + l.  A first-order
// recurrence y[i] = a[i] + z y[i-1] over 64 lanes is a 6-step shuffle scan (y += z^d * y[lane - d], d = 1..32); segments chain
// through the last lane's value.  All global accesses are 512-byte row segments and the row never leaves registers.
template <int S>
__global__ __launch_bounds__(256) void spline_prefilter_rows_wave_kernel(double* __restrict__ coef, long long rows) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    constexpr int W = 64 * S;
    double* base = coef + row * W + lane;
    const double z = kPole;
    double zp[6];                                                      // z^1, z^2, z^4, .. z^32
    zp[0] = z;
#pragma unroll
    for (int k = 1; k < 6; ++k) zp[k] = zp[k - 1] * zp[k - 1];
    double zl = 1.0;                                                   // z^lane
    {
        double p = z;
        for (int e = lane; e; e >>= 1) { if (e & 1) zl *= p; p *= p; }
    }
    const double zl1 = zl * z, zr = zp[5] * zp[5] / zl;                // z^(lane+1), z^(64-lane)
    double y[S];
#pragma unroll
    for (int s = 0; s < S; ++s) y[s] = 6.0 * base[64 * s];
    // exact start: c0 = sum_i z^i 6 s[i] (terms beyond the first 64 are below 1e-37 of it)
    double c0 = zl * y[0];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c0 += __shfl_xor(c0, o, 64);
    if (lane == 0) y[0] = c0;
    double carry = 0.0;
#pragma unroll
    for (int s = 0; s < S; ++s) {                                      // causal
        double v = y[s];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double t = __shfl_up(v, 1 << k, 64);
            if (lane >= (1 << k)) v += zp[k] * t;
        }
        v += zl1 * carry;
        carry = __shfl(v, 63, 64);
        y[s] = v;
    }
    const double ym2 = __shfl(y[S - 1], 62, 64), ym1 = carry;
    const double cinit = (z * ym2 + ym1) * z / (z * z - 1.0);
    carry = 0.0;
#pragma unroll
    for (int s = S - 1; s >= 0; --s) {                                 // anti-causal: c[i] = -z y[i] + z c[i+1]
        double v = (s == S - 1 && lane == 63) ? cinit : -z * y[s];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double t = __shfl_down(v, 1 << k, 64);
            if (lane + (1 << k) < 64) v += zp[k] * t;
        }
        v += zr * carry;
        carry = __shfl(v, 0, 64);
        base[64 * s] = v;
    }
}

// 64 rows per workgroup (thread t owns row r0 + t); the row is walked in 64-column chunks staged through LDS so that every
// global access is a 512-byte row segment.
__global__ __launch_bounds__(64) void spline_prefilter_rows_kernel(double* __restrict__ coef, long long rows, int W) {
    __shared__ double tile[64][65];
    const int t = threadIdx.x;
    const long long r0 = (long long)blockIdx.x * 64;
    const int nr = (int)min((long long)64, rows - r0);
    double* base = coef + r0 * W;
    if (W == 1) return;
    const int nch = (W + 63) / 64;
    auto load = [&](int ch) {
        const int x = ch * 64 + t;
        for (int r = 0; r < nr; ++r) tile[r][t] = x < W ? base[(long long)r * W + x] : 0.0;
        __syncthreads();
    };
    auto store = [&](int ch) {
        __syncthreads();
        const int x = ch * 64 + t;
        if (x < W) for (int r = 0; r < nr; ++r) base[(long long)r * W + x] = tile[r][t];
        __syncthreads();
    };
    const double z = kPole, zn1 = pow(z, (double)(W - 1));
    // causal initialisation: c0 = 6 c[0] + zn1 6 c[W-1] + sum_{i=1}^{W-2} z^i 6 (c[i] + zn1 c[W-1-i]).  The zn1 terms only matter
    // for short rows (|z|^(W-1) < 1e-16 beyond W = 29), where the whole row sits in the first chunk.
    double c0 = 0.0, zi = 1.0;
    for (int ch = 0; ch < nch; ++ch) {
        load(ch);
        if (t < nr) {
            const int n = min(64, W - ch * 64);
            for (int j = 0; j < n; ++j) {
                const int i = ch * 64 + j;
                double v = 6.0 * tile[t][j];
                if (W <= 64 && i >= 1 && i <= W - 2) v += zn1 * 6.0 * tile[t][W - 1 - i];
                if (i == 0 && W <= 64) v += zn1 * 6.0 * tile[t][W - 1];
                if (i <= W - 2 || i == 0) c0 += zi * v;
                zi *= z;
            }
        }
        __syncthreads();
        if (ch >= 1) break;                                            // |z|^128 ~ 1e-73: further terms vanish in fp64
    }
    double prev = c0 / (1.0 - zn1 * zn1);
    for (int ch = 0; ch < nch; ++ch) {                                 // causal pass
        load(ch);
        if (t < nr) {
            const int n = min(64, W - ch * 64);
            for (int j = 0; j < n; ++j) {
                if (ch * 64 + j > 0) prev = 6.0 * tile[t][j] + z * prev;
                tile[t][j] = prev;
            }
        }
        store(ch);
    }
    double nxt = 0.0;
    for (int ch = nch - 1; ch >= 0; --ch) {                            // anti-causal pass
        load(ch);
        if (t < nr) {
            const int n = min(64, W - ch * 64);
            for (int j = n - 1; j >= 0; --j) {
                const int i = ch * 64 + j;
                if (i == W - 1) {
                    // c[W-2] is the previous element of this row: in this chunk, or (W-1 a multiple of 64) the last of the one before
                    const double cm2 = j > 0 ? tile[t][j - 1] : base[(long long)t * W + i - 1];
                    nxt = (z * cm2 + tile[t][j]) * z / (z * z - 1.0);
                } else {
                    nxt = z * (nxt - tile[t][j]);
                }
                tile[t][j] = nxt;
            }
        }
        store(ch);
    }
}

// ---------------------------------------------------------------------------------------------- resize + normalise
#pragma clang fp contract(off)
__device__ __forceinline__ void cubic_taps(double c, int& start, double (&w)[4]) {
    const double f = floor(c), y = c - f, z = 1.0 - y;
    w[1] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0;
    w[2] = (z * z * (z - 2.0) * 3.0 + 4.0) / 6.0;
    w[0] = z * z * z / 6.0;
    w[3] = 1.0 - w[0] - w[1] - w[2];
    start = (int)f - 1;
}

__device__ __forceinline__ int mirror_any(int i, int n) {
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    i = (i < 0 ? -i : i) % p;
    return i >= n ? p - i : i;
}

#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void zoom_sample_kernel(const double* __restrict__ coef, const float* __restrict__ img,
                                                          const unsigned char* __restrict__ lab, float* __restrict__ x,
                                                          long long* __restrict__ y, int B, int H, int W, int OH, int OW, float mean, float stdv) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)B * OH * OW) return;
    const int ox = (int)(t % OW), oy = (int)((t / OW) % OH), b = (int)(t / ((long long)OW * OH));
    float v;
    long long l;
    if (!coef) {                                                     // sizes equal: the reference skips the zoom (dataset_synapse.py:109)
        v = img[t];
        l = lab ? (long long)lab[t] : 0;
    } else {
        const double sy = OH > 1 ? (double)(H - 1) / (double)(OH - 1) : 0.0, sx = OW > 1 ? (double)(W - 1) / (double)(OW - 1) : 0.0;
        const double cy = (double)oy * sy, cx = (double)ox * sx;
        if (cy > (double)(H - 1) || cx > (double)(W - 1)) {          // scipy mode='constant': outside -> cval 0
            v = 0.f; l = 0;
        } else {
            int y0, x0;
            double wy[4], wx[4];
            cubic_taps(cy, y0, wy);
            cubic_taps(cx, x0, wx);
            const double* c = coef + (long long)b * H * W;
            int xi[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) xi[j] = mirror_any(x0 + j, W);
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double* row = c + (long long)mirror_any(y0 + i, H) * W;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc += wy[i] * wx[j] * row[xi[j]];
            }
            v = (float)acc;
            const int iy = min((int)floor(cy + 0.5), H - 1), ix = min((int)floor(cx + 0.5), W - 1);
            l = lab ? (long long)lab[((long long)b * H + iy) * W + ix] : 0;
        }
    }
    x[t] = (v - mean) / stdv;
    if (y) y[t] = l;
}

}  // namespace

extern "C" int tc_slice_augment(const float* img, const unsigned char* lab, const TcSliceAug* aug_dev, float* img_out,
                                unsigned char* lab_out, int B, int H, int W, void* stream) {
    if (!img || !lab || !aug_dev || !img_out || !lab_out || B <= 0 || H < 2 * kBlurR + 1 || W < 2 * kBlurR + 1) return TC_ERR_ARG;
    if (img == img_out || lab == lab_out) return TC_ERR_ARG;        // the warp gathers: not in place
    hipLaunchKernelGGL(slice_augment_kernel, dim3((W + kTile - 1) / kTile, (H + kTile - 1) / kTile, B), dim3(kAugThreads), 0, (hipStream_t)stream,
                       img, lab, img, lab, aug_dev, img_out, lab_out, H, W);
    return tc_launch_status();
}

extern "C" int tc_slice_augment_chain(const float* raw_img, const unsigned char* raw_lab, const TcSliceAug* aug_dev, int first_round,
                                      int rounds, float* img_a, unsigned char* lab_a, float* img_b, unsigned char* lab_b, int B, int H,
                                      int W, void* stream) {
    if (!raw_img || !raw_lab || !aug_dev || !img_a || !lab_a || !img_b || !lab_b || B <= 0 || H < 2 * kBlurR + 1 || W < 2 * kBlurR + 1) return TC_ERR_ARG;
    if (first_round < 0 || rounds < first_round || img_a == img_b || lab_a == lab_b || raw_img == img_a || raw_img == img_b || raw_lab == lab_a || raw_lab == lab_b)
        return TC_ERR_ARG;
    for (int r = first_round; r < rounds; ++r) {                    // round r: reads the buffer round r - 1 wrote (or the raw slice), writes the other
        const bool odd = r & 1;
        hipLaunchKernelGGL(slice_augment_kernel, dim3((W + kTile - 1) / kTile, (H + kTile - 1) / kTile, B), dim3(kAugThreads), 0, (hipStream_t)stream,
                           odd ? img_a : img_b, odd ? lab_a : lab_b, raw_img, raw_lab, aug_dev + (long long)r * B, odd ? img_b : img_a,
                           odd ? lab_b : lab_a, H, W);
        if (tc_launch_status() != TC_OK) return TC_ERR_LAUNCH;
    }
    return TC_OK;
}

extern "C" int tc_spline_prefilter(const float* img, double* coef, int B, int H, int W, void* stream) {
    if (!img || !coef || B <= 0 || H <= 0 || W <= 0) return TC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (H >= 128 && H % 32 == 0)
        hipLaunchKernelGGL((spline_prefilter_cols_seg_kernel<32, 16>), dim3((unsigned)(((long long)B * W * (H / 32) + 255) / 256)), dim3(256), 0, s,
                           img, coef, B, H, W);
    else
        hipLaunchKernelGGL(spline_prefilter_cols_kernel, dim3((unsigned)(((long long)B * W + 255) / 256)), dim3(256), 0, s, img, coef, B, H, W);
    const long long rows = (long long)B * H;
    const dim3 wg((unsigned)((rows + 3) / 4));
    switch (W % 64 == 0 ? W / 64 : 0) {
#define TC_ROWS_WAVE(S) case S: hipLaunchKernelGGL((spline_prefilter_rows_wave_kernel<S>), wg, dim3(256), 0, s, coef, rows); break;
        TC_ROWS_WAVE(1) TC_ROWS_WAVE(2) TC_ROWS_WAVE(3) TC_ROWS_WAVE(4) TC_ROWS_WAVE(6) TC_ROWS_WAVE(8) TC_ROWS_WAVE(12) TC_ROWS_WAVE(16)
#undef TC_ROWS_WAVE
        default: hipLaunchKernelGGL(spline_prefilter_rows_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, s, coef, rows, W);
    }
    return tc_launch_status();
}

extern "C" int tc_zoom_normalize(const double* coef, const float* img, const unsigned char* lab, float* x, long long* y, int B, int H,
                                 int W, int OH, int OW, float mean, float stdv, void* stream) {
    if (!x || B <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || stdv == 0.f) return TC_ERR_ARG;
    if (!coef && (!img || H != OH || W != OW)) return TC_ERR_ARG;
    if (y && !lab) return TC_ERR_ARG;
    const long long n = (long long)B * OH * OW;
    hipLaunchKernelGGL(zoom_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, coef, img, lab, x, y, B, H, W,
                       OH, OW, mean, stdv);
    return tc_launch_status();
}
