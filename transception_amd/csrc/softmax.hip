// Softmax along either axis of a batch of row-major [R, Ccols] matrices (forward and backward).
// axis=1: one wavefront per row, shuffle reductions.  axis=0 (softmax over tokens, per channel): a block owns
// 64 columns (16 lanes x 4 channels = 256 B per row) and 16 row lanes, reductions through LDS.
#include "tc_common.h"
#include <initializer_list>

namespace {

template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_fwd(const T* __restrict__ x, T* __restrict__ y, long long sbx, long long sby,
                                                        int R, int Cc, int ldx, int ldy, long long nrows) {
    const int lane = threadIdx.x & 63;
    const long long gr = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gr >= nrows) return;
    const int b = (int)(gr / R), r = (int)(gr % R);
    const T* xr = x + b * sbx + (long long)r * ldx;
    T* yr = y + b * sby + (long long)r * ldy;
    float m = -INFINITY;
    for (int c = lane; c < Cc; c += 64) m = fmaxf(m, ldf<T>(xr + c));
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < Cc; c += 64) s += __expf(ldf<T>(xr + c) - m);
    s = 1.0f / wave_sum(s);
    for (int c = lane; c < Cc; c += 64) stf<T>(yr + c, __expf(ldf<T>(xr + c) - m) * s);
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_bwd(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx,
                                                        long long sbdy, long long sby, long long sbdx, int R, int Cc, int lddy,
                                                        int ldy, int lddx, long long nrows, int accumulate) {
    const int lane = threadIdx.x & 63;
    const long long gr = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gr >= nrows) return;
    const int b = (int)(gr / R), r = (int)(gr % R);
    const T* dyr = dy + b * sbdy + (long long)r * lddy;
    const T* yr = y + b * sby + (long long)r * ldy;
    T* dxr = dx + b * sbdx + (long long)r * lddx;
    float s = 0.f;
    for (int c = lane; c < Cc; c += 64) s += ldf<T>(dyr + c) * ldf<T>(yr + c);
    s = wave_sum(s);
    for (int c = lane; c < Cc; c += 64) {
        float v = ldf<T>(yr + c) * (ldf<T>(dyr + c) - s);
        if (accumulate) v += ldf<T>(dxr + c);
        stf<T>(dxr + c, v);
    }
}

// Long rows (the N = 6076 key softmax of the bridge's channel attention, MSTr.py:2322-2324: 64 rows per image): ONE workgroup per row,
// the row in registers as up to 8 four-element vectors per thread, maxima and sums folded through LDS -- one pass of 8/16-byte
// accesses where softmax_rows_fwd walks the row three times with one element per lane and one wavefront per row (50 / 62 us).
constexpr int ROWWG_NV = 8;
__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red) {
    v = is_max ? wave_max(v) : wave_sum(v);
    __syncthreads();                                             // (red may still be read by the previous reduction)
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
}
template <typename T>
__global__ __launch_bounds__(256) void softmax_rowwg_fwd(const T* __restrict__ x, T* __restrict__ y, long long sbx, long long sby,
                                                         int R, int Cc, int ldx, int ldy) {
    __shared__ float red[4];
    const int b = blockIdx.x / R, r = blockIdx.x - b * R, nv = Cc >> 2;
    const T* xr = x + b * sbx + (long long)r * ldx;
    T* yr = y + b * sby + (long long)r * ldy;
    float4 v[ROWWG_NV];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < ROWWG_NV; ++i) {
        const int q = threadIdx.x + i * 256;
        v[i] = q < nv ? ld4<T>(xr + 4 * q) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        m = fmaxf(fmaxf(m, fmaxf(v[i].x, v[i].y)), fmaxf(v[i].z, v[i].w));
    }
    m = block_reduce(m, true, red);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ROWWG_NV; ++i) {
        v[i].x = __expf(v[i].x - m); v[i].y = __expf(v[i].y - m); v[i].z = __expf(v[i].z - m); v[i].w = __expf(v[i].w - m);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);              // (padding lanes hold exp(-inf) = 0)
    }
    s = 1.0f / block_reduce(s, false, red);
#pragma unroll
    for (int i = 0; i < ROWWG_NV; ++i) {
        const int q = threadIdx.x + i * 256;
        if (q < nv) st4<T>(yr + 4 * q, make_float4(v[i].x * s, v[i].y * s, v[i].z * s, v[i].w * s));
    }
}
template <typename T>
__global__ __launch_bounds__(256) void softmax_rowwg_bwd(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx,
                                                         long long sbdy, long long sby, long long sbdx, int R, int Cc, int lddy,
                                                         int ldy, int lddx, int accumulate) {
    __shared__ float red[4];
    const int b = blockIdx.x / R, r = blockIdx.x - b * R, nv = Cc >> 2;
    const T* dyr = dy + b * sbdy + (long long)r * lddy;
    const T* yr = y + b * sby + (long long)r * ldy;
    T* dxr = dx + b * sbdx + (long long)r * lddx;
    float4 g[ROWWG_NV], p[ROWWG_NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ROWWG_NV; ++i) {
        const int q = threadIdx.x + i * 256;
        g[i] = q < nv ? ld4<T>(dyr + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        p[i] = q < nv ? ld4<T>(yr + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (g[i].x * p[i].x + g[i].y * p[i].y) + (g[i].z * p[i].z + g[i].w * p[i].w);
    }
    s = block_reduce(s, false, red);
#pragma unroll
    for (int i = 0; i < ROWWG_NV; ++i) {
        const int q = threadIdx.x + i * 256;
        if (q < nv) {
            float4 o = make_float4(p[i].x * (g[i].x - s), p[i].y * (g[i].y - s), p[i].z * (g[i].z - s), p[i].w * (g[i].w - s));
            if (accumulate) { const float4 a = ld4<T>(dxr + 4 * q); o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w; }
            st4<T>(dxr + 4 * q, o);
        }
    }
}
// long-row form usable: whole 4-element vectors at 4-element-aligned addresses, row within ROWWG_NV x 1024 elements
template <typename T>
bool rowwg_ok(int Cc, std::initializer_list<long long> strides, std::initializer_list<const void*> ptrs) {
    if ((Cc & 3) || Cc <= 512 || Cc > ROWWG_NV * 1024) return false;
    for (long long v : strides) if (v & 3) return false;
    for (const void* q : ptrs) if ((uintptr_t)q % (4 * sizeof(T))) return false;
    return true;
}

// Rows of up to 512 channels, a multiple of 4 (the channel softmax of the query in EfficientAttention, MSTr.py:124-128): 16 lanes per
// row, 4 rows per wavefront, each lane keeps its NV four-channel vectors in registers -- one pass over the row with 8/16-byte
// accesses instead of three passes of one element per lane and one row per wavefront.
__device__ __forceinline__ float grp16_max(float v) {
    v = tc_group_max<16>(v);
    return v;
}
__device__ __forceinline__ float grp16_sum(float v) {
    v = tc_group_sum<16>(v);
    return v;
}
template <typename T, int NV>
__global__ __launch_bounds__(256) void softmax_rows16_fwd(const T* __restrict__ x, T* __restrict__ y, long long sbx, long long sby,
                                                          int R, int Cc, int ldx, int ldy, long long nrows) {
    const int l = threadIdx.x & 15;
    const long long gr = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (gr >= nrows) return;
    const int b = (int)(gr / R), r = (int)(gr % R), nv = Cc >> 2;
    const T* xr = x + b * sbx + (long long)r * ldx;
    T* yr = y + b * sby + (long long)r * ldy;
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = ld4<T>(xr + min(l + i * 16, nv - 1) * 4);           // all loads first, no per-lane branch
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < NV; ++i) if (l + i * 16 < nv) m = fmaxf(fmaxf(m, fmaxf(v[i].x, v[i].y)), fmaxf(v[i].z, v[i].w));
    m = grp16_max(m);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i].x = __expf(v[i].x - m); v[i].y = __expf(v[i].y - m); v[i].z = __expf(v[i].z - m); v[i].w = __expf(v[i].w - m);
        if (l + i * 16 < nv) s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    s = 1.0f / grp16_sum(s);
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (l + i * 16 < nv) st4<T>(yr + (l + i * 16) * 4, make_float4(v[i].x * s, v[i].y * s, v[i].z * s, v[i].w * s));
}
template <typename T, int NV>
__global__ __launch_bounds__(256) void softmax_rows16_bwd(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx,
                                                          long long sbdy, long long sby, long long sbdx, int R, int Cc, int lddy,
                                                          int ldy, int lddx, long long nrows, int accumulate) {
    const int l = threadIdx.x & 15;
    const long long gr = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (gr >= nrows) return;
    const int b = (int)(gr / R), r = (int)(gr % R), nv = Cc >> 2;
    const T* dyr = dy + b * sbdy + (long long)r * lddy;
    const T* yr = y + b * sby + (long long)r * ldy;
    T* dxr = dx + b * sbdx + (long long)r * lddx;
    float4 yv[NV], dv[NV], ov[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = min(l + i * 16, nv - 1) * 4;
        yv[i] = ld4<T>(yr + q); dv[i] = ld4<T>(dyr + q);
        ov[i] = accumulate ? ld4<T>(dxr + q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) if (l + i * 16 < nv) s += dv[i].x * yv[i].x + dv[i].y * yv[i].y + dv[i].z * yv[i].z + dv[i].w * yv[i].w;
    s = grp16_sum(s);
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (l + i * 16 < nv)
            st4<T>(dxr + (l + i * 16) * 4, make_float4(yv[i].x * (dv[i].x - s) + ov[i].x, yv[i].y * (dv[i].y - s) + ov[i].y,
                                                       yv[i].z * (dv[i].z - s) + ov[i].z, yv[i].w * (dv[i].w - s) + ov[i].w));
}
template <typename T> bool rows16_ok(int Cc, std::initializer_list<long long> strides, std::initializer_list<const void*> ptrs) {
    constexpr int AL = 4 * (int)sizeof(T);
    if ((Cc & 3) || Cc > 512) return false;
    for (long long v : strides) if (v & 3) return false;
    for (const void* q : ptrs) if ((uintptr_t)q % AL) return false;
    return true;
}

template <typename T>
void launch_rows16_fwd(const T* x, T* y, long long sbx, long long sby, int R, int Cc, int ldx, int ldy, long long nrows, hipStream_t s) {
    const dim3 g16((unsigned)((nrows + 15) / 16));
#define TC_SM16(NV) case NV: hipLaunchKernelGGL((softmax_rows16_fwd<T, NV>), g16, dim3(256), 0, s, x, y, sbx, sby, R, Cc, ldx, ldy, nrows); break;
    switch ((Cc / 4 + 15) / 16) { TC_SM16(1) TC_SM16(2) TC_SM16(3) TC_SM16(4) TC_SM16(5) TC_SM16(6) TC_SM16(7) default: TC_SM16(8) }
#undef TC_SM16
}
template <typename T>
void launch_rows16_bwd(const T* dy, const T* y, T* dx, long long sbdy, long long sby, long long sbdx, int R, int Cc, int lddy, int ldy,
                       int lddx, long long nrows, int accumulate, hipStream_t s) {
    const dim3 g16((unsigned)((nrows + 15) / 16));
#define TC_SM16(NV) case NV: hipLaunchKernelGGL((softmax_rows16_bwd<T, NV>), g16, dim3(256), 0, s, dy, y, dx, sbdy, sby, sbdx, R, Cc, lddy, ldy, lddx, nrows, accumulate); break;
    switch ((Cc / 4 + 15) / 16) { TC_SM16(1) TC_SM16(2) TC_SM16(3) TC_SM16(4) TC_SM16(5) TC_SM16(6) TC_SM16(7) default: TC_SM16(8) }
#undef TC_SM16
}

__device__ __forceinline__ float4 f4max(float4 a, float4 b) { return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

template <bool MAX>
__device__ __forceinline__ float4 block_reduce16(float4 v, float4 (*sm)[16], int tx, int ty) {
    sm[ty][tx] = v;
    __syncthreads();
    float4 r = sm[0][tx];
#pragma unroll
    for (int i = 1; i < 16; ++i) r = MAX ? f4max(r, sm[i][tx]) : f4add(r, sm[i][tx]);
    __syncthreads();
    return r;
}

// Column softmax (over the rows of each [R, Ccols] matrix) split over row ranges so that tall-skinny matrices (R = tokens,
// Ccols = 64) fill the chip: pass 1 writes per-range partial statistics, pass 2 folds them (a few values per column) and
// normalises its own rows.  scratch: fp32 [nb][RS][2][Ccols].
//   forward  : partial (max, sum exp(x - max))            backward: partial sum(dy * y)
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void softmax_cols_stats(const T* __restrict__ x, const T* __restrict__ y2, float* __restrict__ scratch,
                                                          long long sbx, long long sby2, int R, int Cc, int ldx, int ldy2, int rper) {
    __shared__ float4 sm[16][16];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + tx * 4;
    const bool ok = c < Cc;
    const int RS = gridDim.z, rs = blockIdx.z;
    const int r0 = rs * rper, r1 = min(R, r0 + rper);
    const T* xb = x + blockIdx.y * sbx + (ok ? c : 0);
    float* sc = scratch + ((long long)blockIdx.y * RS + rs) * 2 * Cc;
    if (!BWD) {
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        if (ok) for (int r = r0 + ty; r < r1; r += 16) m = f4max(m, ld4<T>(xb + (long long)r * ldx));
        m = block_reduce16<true>(m, sm, tx, ty);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) for (int r = r0 + ty; r < r1; r += 16) {
            const float4 v = ld4<T>(xb + (long long)r * ldx);
            s.x += __expf(v.x - m.x); s.y += __expf(v.y - m.y); s.z += __expf(v.z - m.z); s.w += __expf(v.w - m.w);
        }
        s = block_reduce16<false>(s, sm, tx, ty);
        if (ok && ty == 0) { *reinterpret_cast<float4*>(sc + c) = m; *reinterpret_cast<float4*>(sc + Cc + c) = s; }
    } else {
        const T* yb = y2 + blockIdx.y * sby2 + (ok ? c : 0);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) for (int r = r0 + ty; r < r1; r += 16) {
            const float4 a = ld4<T>(xb + (long long)r * ldx), p = ld4<T>(yb + (long long)r * ldy2);
            s.x += a.x * p.x; s.y += a.y * p.y; s.z += a.z * p.z; s.w += a.w * p.w;
        }
        s = block_reduce16<false>(s, sm, tx, ty);
        if (ok && ty == 0) *reinterpret_cast<float4*>(sc + c) = s;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_cols_fwd(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ scratch,
                                                        long long sbx, long long sby, int R, int Cc, int ldx, int ldy, int rper) {
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + tx * 4;
    if (c >= Cc) return;
    const int RS = gridDim.z, rs = blockIdx.z;
    const float* sc = scratch + (long long)blockIdx.y * RS * 2 * Cc;
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int k = 0; k < RS; ++k) m = f4max(m, *reinterpret_cast<const float4*>(sc + (long long)k * 2 * Cc + c));
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < RS; ++k) {
        const float4 mk = *reinterpret_cast<const float4*>(sc + (long long)k * 2 * Cc + c);
        const float4 sk = *reinterpret_cast<const float4*>(sc + (long long)k * 2 * Cc + Cc + c);
        s.x += sk.x * __expf(mk.x - m.x); s.y += sk.y * __expf(mk.y - m.y); s.z += sk.z * __expf(mk.z - m.z); s.w += sk.w * __expf(mk.w - m.w);
    }
    s.x = 1.f / s.x; s.y = 1.f / s.y; s.z = 1.f / s.z; s.w = 1.f / s.w;
    const T* xb = x + blockIdx.y * sbx + c;
    T* yb = y + blockIdx.y * sby + c;
    const int r0 = rs * rper, r1 = min(R, r0 + rper);
    for (int r = r0 + ty; r < r1; r += 16) {
        float4 v = ld4<T>(xb + (long long)r * ldx);
        v.x = __expf(v.x - m.x) * s.x; v.y = __expf(v.y - m.y) * s.y; v.z = __expf(v.z - m.z) * s.z; v.w = __expf(v.w - m.w) * s.w;
        st4<T>(yb + (long long)r * ldy, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_cols_bwd(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx,
                                                        const float* __restrict__ scratch, long long sbdy, long long sby, long long sbdx, int R,
                                                        int Cc, int lddy, int ldy, int lddx, int accumulate, int rper) {
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + tx * 4;
    if (c >= Cc) return;
    const int RS = gridDim.z, rs = blockIdx.z;
    const float* sc = scratch + (long long)blockIdx.y * RS * 2 * Cc;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < RS; ++k) s = f4add(s, *reinterpret_cast<const float4*>(sc + (long long)k * 2 * Cc + c));
    const T* dyb = dy + blockIdx.y * sbdy + c;
    const T* yb = y + blockIdx.y * sby + c;
    T* dxb = dx + blockIdx.y * sbdx + c;
    const int r0 = rs * rper, r1 = min(R, r0 + rper);
    for (int r = r0 + ty; r < r1; r += 16) {
        const float4 a = ld4<T>(dyb + (long long)r * lddy), p = ld4<T>(yb + (long long)r * ldy);
        float4 o = make_float4(p.x * (a.x - s.x), p.y * (a.y - s.y), p.z * (a.z - s.z), p.w * (a.w - s.w));
        if (accumulate) { const float4 q = ld4<T>(dxb + (long long)r * lddx); o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w; }
        st4<T>(dxb + (long long)r * lddx, o);
    }
}

inline int cols_rsplit(int R, int& rper) {
    int RS = (R + 127) / 128;
    if (RS > 64) RS = 64;
    if (RS < 1) RS = 1;
    rper = (R + RS - 1) / RS;
    return (R + rper - 1) / rper;
}

}  // namespace

extern "C" long long tc_softmax_scratch_floats(int nb, int R, int Cc) {
    int rper;
    return (long long)nb * cols_rsplit(R, rper) * 2 * Cc;
}

extern "C" int tc_softmax_fwd(const void* x, void* y, float* scratch, int nb, long long sbx, long long sby, int R, int Cc, int ldx, int ldy,
                              int axis, int dtype, void* stream) {
    if (!x || !y || nb <= 0 || R <= 0 || Cc <= 0 || (axis != 0 && axis != 1)) return TC_ERR_ARG;
    if (axis == 0 && (!scratch || (Cc & 3) || (ldx & 3) || (ldy & 3) || (sbx & 3) || (sby & 3))) return TC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const long long nrows = (long long)nb * R;
    TC_DISPATCH_DTYPE(dtype, {
        if (axis == 1 && rows16_ok<T>(Cc, {sbx, sby, ldx, ldy}, {x, y})) {
            launch_rows16_fwd<T>((const T*)x, (T*)y, sbx, sby, R, Cc, ldx, ldy, nrows, s);
        } else if (axis == 1 && rowwg_ok<T>(Cc, {sbx, sby, ldx, ldy}, {x, y})) {
            hipLaunchKernelGGL((softmax_rowwg_fwd<T>), dim3((unsigned)nrows), dim3(256), 0, s, (const T*)x, (T*)y, sbx, sby, R, Cc, ldx, ldy);
        } else if (axis == 1) hipLaunchKernelGGL((softmax_rows_fwd<T>), dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, s, (const T*)x,
                                          (T*)y, sbx, sby, R, Cc, ldx, ldy, nrows);
        else {
            int rper;
            const int RS = cols_rsplit(R, rper);
            dim3 grid((Cc + 63) / 64, nb, RS);
            hipLaunchKernelGGL((softmax_cols_stats<T, false>), grid, dim3(256), 0, s, (const T*)x, (const T*)nullptr, scratch, sbx, 0LL, R, Cc,
                               ldx, 0, rper);
            hipLaunchKernelGGL((softmax_cols_fwd<T>), grid, dim3(256), 0, s, (const T*)x, (T*)y, scratch, sbx, sby, R, Cc, ldx, ldy, rper);
        }
    });
    return tc_launch_status();
}

extern "C" int tc_softmax_bwd(const void* dy, const void* y, void* dx, float* scratch, int nb, long long sbdy, long long sby, long long sbdx,
                              int R, int Cc, int lddy, int ldy, int lddx, int axis, int accumulate, int dtype, void* stream) {
    if (!dy || !y || !dx || nb <= 0 || R <= 0 || Cc <= 0 || (axis != 0 && axis != 1)) return TC_ERR_ARG;
    if (axis == 0 && (!scratch || (Cc & 3) || (lddy & 3) || (ldy & 3) || (lddx & 3) || (sbdy & 3) || (sby & 3) || (sbdx & 3)))
        return TC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const long long nrows = (long long)nb * R;
    TC_DISPATCH_DTYPE(dtype, {
        if (axis == 1 && rows16_ok<T>(Cc, {sbdy, sby, sbdx, lddy, ldy, lddx}, {dy, y, dx})) {
            launch_rows16_bwd<T>((const T*)dy, (const T*)y, (T*)dx, sbdy, sby, sbdx, R, Cc, lddy, ldy, lddx, nrows, accumulate, s);
        } else if (axis == 1 && rowwg_ok<T>(Cc, {sbdy, sby, sbdx, lddy, ldy, lddx}, {dy, y, dx})) {
            hipLaunchKernelGGL((softmax_rowwg_bwd<T>), dim3((unsigned)nrows), dim3(256), 0, s, (const T*)dy, (const T*)y, (T*)dx, sbdy, sby,
                               sbdx, R, Cc, lddy, ldy, lddx, accumulate);
        } else if (axis == 1) hipLaunchKernelGGL((softmax_rows_bwd<T>), dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, s, (const T*)dy,
                                          (const T*)y, (T*)dx, sbdy, sby, sbdx, R, Cc, lddy, ldy, lddx, nrows, accumulate);
        else {
            int rper;
            const int RS = cols_rsplit(R, rper);
            dim3 grid((Cc + 63) / 64, nb, RS);
            hipLaunchKernelGGL((softmax_cols_stats<T, true>), grid, dim3(256), 0, s, (const T*)dy, (const T*)y, scratch, sbdy, sby, R, Cc, lddy,
                               ldy, rper);
            hipLaunchKernelGGL((softmax_cols_bwd<T>), grid, dim3(256), 0, s, (const T*)dy, (const T*)y, (T*)dx, scratch, sbdy, sby, sbdx, R, Cc,
                               lddy, ldy, lddx, accumulate, rper);
        }
    });
    return tc_launch_status();
}
