// Reductions and training-step kernels: bias-gradient column sums, the fused CE + Dice segmentation loss
// (trainer.py:141-143, utils.py:24-47) and the fused SGD-momentum update (trainer.py:125).
#include "tc_common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int rows, int cols, int ldx, float* __restrict__ out,
                                                     int nb, long long sb) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + tx;
    float s = 0.f;
    if (c < cols)
        for (long long r = (long long)blockIdx.x * 4 + ty; r < (long long)rows * nb; r += (long long)gridDim.x * 4)
            s += ldf<T>(x + (r / rows) * sb + (r % rows) * ldx + c);
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < cols) atomicAdd(out + c, red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]);
}

constexpr int MAXCLS = 16;

// a token row of up to 16 classes whose storage is 16-byte aligned and padded to a multiple of 8 elements: one or two 16-byte pieces
template <typename T> __device__ __forceinline__ void tok_row_load(const T* lp, int ncls, float* v) {
    if constexpr (sizeof(T) == 2) {
        const uint4 a = *reinterpret_cast<const uint4*>(lp);
        unpack2<T>(a.x, v[0], v[1]); unpack2<T>(a.y, v[2], v[3]); unpack2<T>(a.z, v[4], v[5]); unpack2<T>(a.w, v[6], v[7]);
        if (ncls > 8) {
            const uint4 b = *reinterpret_cast<const uint4*>(lp + 8);
            unpack2<T>(b.x, v[8], v[9]); unpack2<T>(b.y, v[10], v[11]); unpack2<T>(b.z, v[12], v[13]); unpack2<T>(b.w, v[14], v[15]);
        }
    }
}
template <typename T> __device__ __forceinline__ void tok_row_store(T* dp, int ncls, const float* v) {
    if constexpr (sizeof(T) == 2) {
        *reinterpret_cast<uint4*>(dp) = make_uint4(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), pack2<T>(v[4], v[5]), pack2<T>(v[6], v[7]));
        if (ncls > 8) *reinterpret_cast<uint4*>(dp + 8) = make_uint4(pack2<T>(v[8], v[9]), pack2<T>(v[10], v[11]), pack2<T>(v[12], v[13]), pack2<T>(v[14], v[15]));
    }
}

// ld = 0: logits / dlogits are [B, ncls, HW] (the module's NCHW output); ld > 0: token-major [B * HW, ld] rows as the last Linear leaves
// them (the captured training step hands them over without the transpose and the fp32 copy); prob is [B, ncls, HW] either way
template <typename T>
__global__ __launch_bounds__(256) void seg_loss_fwd_kernel(const T* __restrict__ logits, const long long* __restrict__ labels,
                                                           float* __restrict__ prob, float* __restrict__ sums, int B, int ncls, int HW, int ld) {
    __shared__ float red[4][1 + 3 * MAXCLS];
    const bool vec16 = ld > 0 && !(ld & 7) && ld >= ((ncls + 7) & ~7) && !((uintptr_t)logits & 15);
    float ce = 0.f, I[MAXCLS], Y[MAXCLS], Z[MAXCLS];
#pragma unroll
    for (int k = 0; k < MAXCLS; ++k) I[k] = Y[k] = Z[k] = 0.f;
    const long long n = (long long)B * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int b = (ld && !prob) ? 0 : (int)((unsigned)i / (unsigned)HW), p = (int)i - b * HW;     // (B * HW < 2^31: checked by the entry points;
                                                                                                    //  token-major without a probability map needs neither)
        const T* lp = ld ? logits + i * ld : logits + (long long)b * ncls * HW + p;
        const long long ks = ld ? 1 : HW;
        float v[MAXCLS], m = -INFINITY;
        if (sizeof(T) == 2 && vec16) {                          // padded token rows (Graph.ln_cls(pad_rows)): whole 16-byte pieces instead of ncls 2-byte loads
            tok_row_load<T>(lp, ncls, v);
#pragma unroll
            for (int k = 0; k < MAXCLS; ++k) if (k < ncls) m = fmaxf(m, v[k]);
        } else {
#pragma unroll
            for (int k = 0; k < MAXCLS; ++k) if (k < ncls) { v[k] = ldf<T>(lp + k * ks); m = fmaxf(m, v[k]); }
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < MAXCLS; ++k) if (k < ncls) { v[k] = expf(v[k] - m); s += v[k]; }
        const float inv = 1.f / s;
        const int lab = (int)labels[i];
        float* pp = prob + (long long)b * ncls * HW + p;
#pragma unroll
        for (int k = 0; k < MAXCLS; ++k) if (k < ncls) {
            const float pk = v[k] * inv;
            if (prob) pp[(long long)k * HW] = pk;                  // (null: the backward recomputes the softmax from the logits)
            Z[k] += pk * pk;
            if (k == lab) { I[k] += pk; Y[k] += 1.f; ce -= logf(fmaxf(pk, 1e-38f)); }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    ce = wave_sum(ce);
    if (lane == 0) red[wave][0] = ce;
#pragma unroll
    for (int k = 0; k < MAXCLS; ++k) if (k < ncls) {
        const float a = wave_sum(I[k]), b = wave_sum(Y[k]), c = wave_sum(Z[k]);
        if (lane == 0) { red[wave][1 + 3 * k] = a; red[wave][2 + 3 * k] = b; red[wave][3 + 3 * k] = c; }
    }
    __syncthreads();
    if (threadIdx.x < 1 + 3 * ncls)
        atomicAdd(sums + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

template <typename T>
__global__ __launch_bounds__(256) void seg_loss_bwd_kernel(const float* __restrict__ prob, const long long* __restrict__ labels,
                                                           const float* __restrict__ sums, T* __restrict__ dlogits, int B, int ncls,
                                                           int HW, float w_ce, float w_dice, float n_pix, float gscale, const float* __restrict__ gscale_dev, int ld,
                                                           const T* __restrict__ logits, int ldl) {
    if (gscale_dev) gscale *= *gscale_dev;
    const bool vin = !prob && !(ldl & 7) && ldl >= ((ncls + 7) & ~7) && !((uintptr_t)logits & 15);
    const bool vout = ld > 0 && !(ld & 7) && ld >= ((ncls + 7) & ~7) && !((uintptr_t)dlogits & 15);
    __shared__ float ca[MAXCLS], cb[MAXCLS];          // dDice/dp_c = ca[c]*onehot_c + cb[c]*p_c
    if (threadIdx.x < ncls) {
        const float I = sums[1 + 3 * threadIdx.x], Y = sums[2 + 3 * threadIdx.x], Z = sums[3 + 3 * threadIdx.x];
        const float den = Z + Y + 1e-5f, num = 2.f * I + 1e-5f;
        ca[threadIdx.x] = -w_dice / (float)ncls * 2.f / den;
        cb[threadIdx.x] = w_dice / (float)ncls * 2.f * num / (den * den);
    }
    __syncthreads();
    const long long n = (long long)B * HW;
    const float cew = w_ce / n_pix;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int b = (ld && !prob) ? 0 : (int)((unsigned)i / (unsigned)HW), p = (int)i - b * HW;
        const float* pp = prob + (long long)b * ncls * HW + p;
        const int lab = (int)labels[i];
        float pk[MAXCLS], g[MAXCLS], dot = 0.f;
        if (prob) {
#pragma unroll
            for (int k = 0; k < MAXCLS; ++k) if (k < ncls) pk[k] = pp[(long long)k * HW];
        } else {                                                // the forward's arithmetic again (same operations, same order: same bits)
            const T* lp = logits + i * ldl;
            float m = -INFINITY, ssum = 0.f;
            if (sizeof(T) == 2 && vin) {
                tok_row_load<T>(lp, ncls, pk);
#pragma unroll
                for (int k = 0; k < MAXCLS; ++k) if (k < ncls) m = fmaxf(m, pk[k]);
            } else {
#pragma unroll
                for (int k = 0; k < MAXCLS; ++k) if (k < ncls) { pk[k] = ldf<T>(lp + k); m = fmaxf(m, pk[k]); }
            }
#pragma unroll
            for (int k = 0; k < MAXCLS; ++k) if (k < ncls) { pk[k] = expf(pk[k] - m); ssum += pk[k]; }
            const float inv = 1.f / ssum;
#pragma unroll
            for (int k = 0; k < MAXCLS; ++k) if (k < ncls) pk[k] *= inv;
        }
#pragma unroll
        for (int k = 0; k < MAXCLS; ++k) if (k < ncls) {
            g[k] = (k == lab ? ca[k] : 0.f) + cb[k] * pk[k];
            dot += g[k] * pk[k];
        }
        T* dp = ld ? dlogits + i * ld : dlogits + (long long)b * ncls * HW + p;
        const long long ks = ld ? 1 : HW;
        if (sizeof(T) == 2 && vout) {                           // padded rows: the pad elements are written as zeros
            float o[MAXCLS];
#pragma unroll
            for (int k = 0; k < MAXCLS; ++k) o[k] = k < ncls ? gscale * (cew * (pk[k] - (k == lab ? 1.f : 0.f)) + pk[k] * (g[k] - dot)) : 0.f;
            tok_row_store<T>(dp, ncls, o);
        } else {
#pragma unroll
            for (int k = 0; k < MAXCLS; ++k) if (k < ncls)
                stf<T>(dp + k * ks, gscale * (cew * (pk[k] - (k == lab ? 1.f : 0.f)) + pk[k] * (g[k] - dot)));
        }
    }
}

__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long long n, float lr,
                           float mom, float wd, float gscale, int first, const float* __restrict__ lr_dev) {
    if (lr_dev) lr = *lr_dev;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float w = p[i];
        const float d = g[i] * gscale + wd * w;
        const float b = first ? d : mom * buf[i] + d;
        buf[i] = b;
        p[i] = w - lr * b;
    }
}

// all contiguous "used" parameter segments in one launch: blockIdx.y = segment, segs[2*i] = offset, segs[2*i+1] = length
// sum of squares of a flat fp32 buffer, accumulated into *out (the total gradient norm of nn.utils.clip_grad_norm_, trainer.py:147-148:
// parameters without a gradient hold zeros in the arena)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long long n4, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

__global__ void sgd_multi_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, const long long* __restrict__ segs,
                                 float lr, float mom, float wd, float gscale, int first, const float* __restrict__ lr_dev,
                                 const float* __restrict__ clip_sumsq, float clip_norm, void* __restrict__ lp, int lp_dtype) {
    if (lr_dev) lr = *lr_dev;
    if (clip_sumsq) {
        const float ss = *clip_sumsq;
        if (!(ss < __builtin_inff())) return;                     // a non-finite gradient (float16 overflow under a static loss scale): skip the
        gscale *= fminf(clip_norm / (sqrtf(ss) + 1e-6f), 1.0f);   // update -- weights, momentum and the 16-bit copy stay as they were.  Otherwise
    }                                                             // clip_grad_norm_'s coefficient, clamped to 1 (clip_norm = inf: no clipping)
    const long long off = segs[2 * blockIdx.y], n = segs[2 * blockIdx.y + 1];
    if (!((off | n) & 3) && !(((uintptr_t)p | (uintptr_t)g | (uintptr_t)buf) & 15) && !((uintptr_t)lp & 7)) {
        // four elements per thread in 16-byte pieces (the arena's segments start and end on multiples of 8 elements): the scalar form
        // ran at 4.1 TB/s for the 22 bytes per weight it moves
        const long long n4 = n >> 2;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
            const long long j = off + 4 * i;
            const float4 w = *reinterpret_cast<const float4*>(p + j), gv = *reinterpret_cast<const float4*>(g + j);
            float4 b;
            if (first) b = make_float4(gv.x * gscale + wd * w.x, gv.y * gscale + wd * w.y, gv.z * gscale + wd * w.z, gv.w * gscale + wd * w.w);
            else {
                const float4 m = *reinterpret_cast<const float4*>(buf + j);
                b = make_float4(mom * m.x + (gv.x * gscale + wd * w.x), mom * m.y + (gv.y * gscale + wd * w.y), mom * m.z + (gv.z * gscale + wd * w.z),
                                mom * m.w + (gv.w * gscale + wd * w.w));
            }
            *reinterpret_cast<float4*>(buf + j) = b;
            const float4 wn = make_float4(w.x - lr * b.x, w.y - lr * b.y, w.z - lr * b.z, w.w - lr * b.w);
            *reinterpret_cast<float4*>(p + j) = wn;
            if (lp_dtype == TC_BF16) *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(lp) + j) = make_uint2(pack2<bf16_t>(wn.x, wn.y), pack2<bf16_t>(wn.z, wn.w));
            else if (lp_dtype == TC_F16) *reinterpret_cast<uint2*>(reinterpret_cast<f16_t*>(lp) + j) = make_uint2(pack2<f16_t>(wn.x, wn.y), pack2<f16_t>(wn.z, wn.w));
        }
        return;
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long j = off + i;
        const float w = p[j];
        const float d = g[j] * gscale + wd * w;
        const float b = first ? d : mom * buf[j] + d;
        buf[j] = b;
        const float wn = w - lr * b;
        p[j] = wn;
        // the 16-bit working copy of the weights that the next forward reads (otherwise a separate cast pass over the whole arena)
        if (lp_dtype == TC_BF16) reinterpret_cast<bf16_t*>(lp)[j] = f2bf(wn);
        else if (lp_dtype == TC_F16) reinterpret_cast<f16_t*>(lp)[j] = f2h(wn);
    }
}

// Evaluation (utils.py:72-76,96-97): pred = argmax_k logits[b,k,p] (softmax is monotone; first maximum wins like torch.argmax) and,
// when labels are given, per-class voxel counts  counts[3k..3k+2] += (|pred==k & gt==k|, |pred==k|, |gt==k|)  for the Dice of
// calculate_metric_percase (utils.py:50-60).  Counts are exact in fp32 up to 2^24 per launch; the host accumulates in float64.
template <typename T>
__global__ __launch_bounds__(256) void argmax_counts_kernel(const T* __restrict__ logits, const long long* __restrict__ labels,
                                                            unsigned char* __restrict__ pred, float* __restrict__ counts, int B, int ncls,
                                                            int HW) {
    __shared__ float red[4][3 * MAXCLS];
    float cnt[3 * MAXCLS];
#pragma unroll
    for (int k = 0; k < 3 * MAXCLS; ++k) cnt[k] = 0.f;
    const long long n = (long long)B * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / HW), p = (int)(i % HW);
        const T* lp = logits + (long long)b * ncls * HW + p;
        float m = ldf<T>(lp);
        int arg = 0;
#pragma unroll
        for (int k = 1; k < MAXCLS; ++k) if (k < ncls) { const float v = ldf<T>(lp + (long long)k * HW); if (v > m) { m = v; arg = k; } }
        pred[i] = (unsigned char)arg;
        if (labels) {
            const int lab = (int)labels[i];
#pragma unroll
            for (int k = 0; k < MAXCLS; ++k) if (k < ncls) {
                cnt[3 * k] += (arg == k && lab == k) ? 1.f : 0.f;
                cnt[3 * k + 1] += (arg == k) ? 1.f : 0.f;
                cnt[3 * k + 2] += (lab == k) ? 1.f : 0.f;
            }
        }
    }
    if (!labels) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3 * MAXCLS; ++k) if (k < 3 * ncls) { const float a = wave_sum(cnt[k]); if (lane == 0) red[wave][k] = a; }
    __syncthreads();
    if (threadIdx.x < 3 * ncls) atomicAdd(counts + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ void zero_floats_kernel(float* __restrict__ p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0.f;
}

}  // namespace

extern "C" int tc_colsum(const void* x, int rows, int cols, int ldx, int nb, long long sb, float* out, int accumulate, int dtype,
                         void* stream) {
    if (!x || !out || rows <= 0 || cols <= 0 || nb <= 0) return TC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    // (a kernel, not hipMemsetAsync: memset NODES of a captured single-stream graph were observed to run out of order on ROCm 7.2)
    if (!accumulate) hipLaunchKernelGGL(zero_floats_kernel, dim3((cols + 255) / 256), dim3(256), 0, s, out, cols);
    dim3 grid(tc_blocks((long long)rows * nb, 4 * 32, 256), (cols + 63) / 64);
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((colsum_kernel<T>), grid, dim3(256), 0, s, (const T*)x, rows, cols, ldx, out, nb, sb));
    return tc_launch_status();
}

extern "C" int tc_seg_loss_fwd(const void* logits, const long long* labels, float* prob, float* sums, int B, int ncls, int HW,
                               int dtype, void* stream) {
    if (!logits || !labels || !prob || !sums || B <= 0 || ncls <= 0 || ncls > MAXCLS || HW <= 0 || (long long)B * HW >= 0x7fffffffLL) return TC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((seg_loss_fwd_kernel<T>), dim3(tc_blocks((long long)B * HW, 256, 1024)), dim3(256), 0, s,
                                                (const T*)logits, labels, prob, sums, B, ncls, HW, 0));
    return tc_launch_status();
}

extern "C" int tc_seg_loss_fwd_tok(const void* logits, int ld, const long long* labels, float* prob, float* sums, int B, int ncls, int HW,
                                   int dtype, void* stream) {
    if (!logits || !labels || !sums || B <= 0 || ncls <= 0 || ncls > MAXCLS || HW <= 0 || ld < ncls || (long long)B * HW >= 0x7fffffffLL) return TC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((seg_loss_fwd_kernel<T>), dim3(tc_blocks((long long)B * HW, 256, 1024)), dim3(256), 0, s,
                                                (const T*)logits, labels, prob, sums, B, ncls, HW, ld));
    return tc_launch_status();
}

// loss, CE and Dice from the (all-reduced) sums, in double like the host expression it replaces (sixteen scalar launches between the
// forward and the backward of every step).  trainer.py:141-143, utils.py:34-47.
__global__ void seg_loss_value_kernel(const float* __restrict__ sums, int ncls, double n_pix, double w_ce, double w_dice,
                                      float* __restrict__ out3) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double ce = (double)sums[0] / n_pix;
    double acc = 0.0;
    for (int c = 0; c < ncls; ++c) {
        const double inter = sums[1 + 3 * c], ysum = sums[2 + 3 * c], zsum = sums[3 + 3 * c];
        acc += 1.0 - (2.0 * inter + 1e-5) / (zsum + ysum + 1e-5);
    }
    const double dice = acc / (double)ncls;
    out3[0] = (float)(w_ce * ce + w_dice * dice);
    out3[1] = (float)ce;
    out3[2] = (float)dice;
}

extern "C" int tc_seg_loss_value(const float* sums, int ncls, double n_pix, double w_ce, double w_dice, float* out3, void* stream) {
    if (!sums || !out3 || ncls <= 0 || ncls > MAXCLS || !(n_pix > 0.0)) return TC_ERR_ARG;
    hipLaunchKernelGGL(seg_loss_value_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, ncls, n_pix, w_ce, w_dice, out3);
    return tc_launch_status();
}

extern "C" int tc_argmax_counts(const void* logits, const long long* labels, unsigned char* pred, float* counts, int B, int ncls, int HW,
                                int dtype, void* stream) {
    if (!logits || !pred || (labels && !counts) || B <= 0 || ncls <= 0 || ncls > MAXCLS || HW <= 0) return TC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((argmax_counts_kernel<T>), dim3(tc_blocks((long long)B * HW, 256, 512)), dim3(256), 0, s,
                                                (const T*)logits, labels, pred, counts, B, ncls, HW));
    return tc_launch_status();
}

extern "C" int tc_seg_loss_bwd(const float* prob, const long long* labels, const float* sums, void* dlogits, int B, int ncls, int HW,
                               float w_ce, float w_dice, float n_pix_global, float gscale, const float* gscale_dev, int dtype, void* stream) {
    if (!prob || !labels || !sums || !dlogits || B <= 0 || ncls <= 0 || ncls > MAXCLS || HW <= 0 || (long long)B * HW >= 0x7fffffffLL) return TC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((seg_loss_bwd_kernel<T>), dim3(tc_blocks((long long)B * HW, 256, 2048)), dim3(256), 0, s,
                                                prob, labels, sums, (T*)dlogits, B, ncls, HW, w_ce, w_dice, n_pix_global, gscale, gscale_dev, 0, (const T*)nullptr, 0));
    return tc_launch_status();
}

extern "C" int tc_seg_loss_bwd_tok(const float* prob, const void* logits, int ldl, const long long* labels, const float* sums, void* dlogits, int ld,
                                   int B, int ncls, int HW, float w_ce, float w_dice, float n_pix_global, float gscale, const float* gscale_dev,
                                   int dtype, void* stream) {
    if ((!prob && (!logits || ldl < ncls)) || !labels || !sums || !dlogits || B <= 0 || ncls <= 0 || ncls > MAXCLS || HW <= 0 || ld < ncls ||
        (long long)B * HW >= 0x7fffffffLL)
        return TC_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    TC_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((seg_loss_bwd_kernel<T>), dim3(tc_blocks((long long)B * HW, 256, 2048)), dim3(256), 0, s,
                                                prob, labels, sums, (T*)dlogits, B, ncls, HW, w_ce, w_dice, n_pix_global, gscale, gscale_dev, ld, (const T*)logits, ldl));
    return tc_launch_status();
}

extern "C" int tc_sgd_step(float* p, const float* grad, float* buf, long long n, float lr, const float* lr_dev, float momentum, float wd,
                           float gscale, int first, void* stream) {
    if (!p || !grad || !buf || n <= 0) return TC_ERR_ARG;
    hipLaunchKernelGGL(sgd_kernel, dim3(tc_blocks(n, 256 * 4, 4096)), dim3(256), 0, (hipStream_t)stream, p, grad, buf, n, lr, momentum, wd,
                       gscale, first, lr_dev);
    return tc_launch_status();
}

extern "C" int tc_sgd_step_multi(float* p, const float* grad, float* buf, const long long* segs_dev, int nseg, long long max_len, float lr,
                                 const float* lr_dev, float momentum, float wd, float gscale, int first, const float* clip_sumsq,
                                 float clip_norm, void* lp, int lp_dtype, void* stream) {
    if (!p || !grad || !buf || !segs_dev || nseg <= 0 || nseg > 65535 || max_len <= 0 || (clip_sumsq && !(clip_norm > 0.f)) ||
        (lp && lp_dtype != TC_BF16 && lp_dtype != TC_F16))
        return TC_ERR_ARG;
    hipLaunchKernelGGL(sgd_multi_kernel, dim3(tc_blocks(max_len, 256 * 8, 256), nseg), dim3(256), 0, (hipStream_t)stream, p, grad, buf, segs_dev,
                       lr, momentum, wd, gscale, first, lr_dev, clip_sumsq, clip_norm, lp, lp ? lp_dtype : -1);
    return tc_launch_status();
}

__global__ void fill_f32_kernel(float* __restrict__ p, long long n, float v) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = v;
}
extern "C" int tc_fill_f32(float* p, long long n, float v, void* stream) {
    if (!p || n <= 0) return TC_ERR_ARG;
    hipLaunchKernelGGL(fill_f32_kernel, dim3(tc_blocks(n, 256 * 4, 2048)), dim3(256), 0, (hipStream_t)stream, p, n, v);
    return tc_launch_status();
}

extern "C" int tc_grad_sumsq(const float* g, long long n, float* out, void* stream) {
    if (!g || !out || n <= 0 || (n & 3) || (uintptr_t)g % 16) return TC_ERR_ARG;
    hipLaunchKernelGGL(sumsq_kernel, dim3(tc_blocks(n / 4, 256 * 8, 1024)), dim3(256), 0, (hipStream_t)stream, g, n / 4, out);
    return tc_launch_status();
}
