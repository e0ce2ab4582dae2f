// ResInception Patch Merging (RIPM), MSTr.py:704-732 / 309-362: Patch_Embed_stage = three DWConv2d_BN steps
//   y = dw3x3(x, stride 2 | 1)   z = pw1x1(y)   x' = Hardswish(BatchNorm(z))
// One step cost three launches (depthwise, 1x1 GEMM with the statistics in its epilogue, BatchNorm apply) on maps of 1.6 - 0.25 MB: every
// one a ~5 us memory round trip.  Here a step is ONE launch, and the BatchNorm + Hardswish of step i is applied by step i + 1 as it
// loads its input ("the consumer normalises on read"), which also writes the normalised map out for the step's other consumers (the MB
// path that starts from it):
//   tc_ripm_fwd:  [x = Hardswish(BN(z_prev)) on the way in, x written out]  ->  dw3x3 (tile with halo in LDS)  ->  y written (the 1x1's
//                 backward reads it)  ->  pw1x1 by MFMA (D^T tiles: lane = pixel)  ->  z written + the per-tile shifted sums of z, in
//                 tc_bn_fwd's scratch layout (shift[C] | S1[T][C] | S2[T][C]; T = tiles), so that tc_bn_fwd / tc_bn_bwd / the next step
//                 consume them unchanged.
// Workgroup = (image, 8 x 8 output pixels, 64 output channels); the input channels go through LDS 64 at a time (the next chunk's rows are
// in flight while this chunk is convolved and multiplied).  16-bit storage; C a multiple of 64.  The backward stays op by op for now:
// every tensor its kernels read (x, y, z, the statistics) is written exactly as the three launches left it.
#include "tc_common.h"
#include <cstdlib>

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct RipmFwdDev {
    const void *xin, *wd, *wp, *gamma, *beta;
    void *xnorm, *y, *z;
    float* part_out;
    const float *shift_out, *part_in;
    float *rmean, *rvar, *save_mean, *save_rstd;
    float eps, momentum;
    int chunks_in, rows_in, training, bn_in;
    int ldx, ldn, ldy, ldz, B, Hi, Wi, Ho, Wo, tilesH, tilesW;
};

template <typename T> __device__ __forceinline__ float rp_cvt16(unsigned r);
template <> __device__ __forceinline__ float rp_cvt16<bf16_t>(unsigned r) { return __uint_as_float(r << 16); }
template <> __device__ __forceinline__ float rp_cvt16<f16_t>(unsigned r) { f16_t h; h.v = (unsigned short)r; return h2f(h); }
template <typename T> __device__ __forceinline__ void rp_unpack(const u32x4& r, float* o) {
    unpack2<T>(r.x, o[0], o[1]); unpack2<T>(r.y, o[2], o[3]); unpack2<T>(r.z, o[4], o[5]); unpack2<T>(r.w, o[6], o[7]);
}
template <typename T> __device__ __forceinline__ u32x4 rp_pack(const float* o) {
    u32x4 r;
    r.x = pack2<T>(o[0], o[1]); r.y = pack2<T>(o[2], o[3]); r.z = pack2<T>(o[4], o[5]); r.w = pack2<T>(o[6], o[7]);
    return r;
}

// workgroup barrier that waits for this wave's LDS traffic only: __syncthreads() also waits for every global load in flight, i.e. for the
// NEXT chunk's rows -- the prefetch then hides nothing and every chunk costs a memory round trip (18.7 us per step at C = 320)
#define RP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <typename T, int C, int S>
__global__ __launch_bounds__(256) void ripm_fwd_kernel(RipmFwdDev p) {
    constexpr int NC = C / 64, TWI = 8 * S + 2, NPX = TWI * TWI, NPJ = (NPX * 8 + 255) / 256, PT = 72;   // LDS row pitch (elements): 64 + 8
    typedef typename TcHalf<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_[];
    float* sc = reinterpret_cast<float*>(smem_);          // [C] BatchNorm scale of the INPUT (bn_in)
    float* sh = sc + C;                                   // [C] shift
    float* wt = sh + C;                                   // [9][64] taps of the chunk's channels
    float* red = wt + 9 * 64;                             // fold / statistics scratch: 2 * 16 * 64... sized below
    T* in_t = reinterpret_cast<T*>(red + 2048);           // [NPX][PT]
    T* a_t = in_t + NPX * PT;                             // [64][PT]  dw output of the chunk (MFMA operand), then the z tile
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int tile = blockIdx.x, js = blockIdx.y;         // js: 64-wide slice of the output channels
    const int tx = tile % p.tilesW, ty = (tile / p.tilesW) % p.tilesH, b = tile / (p.tilesW * p.tilesH);
    const int oy0 = ty * 8, ox0 = tx * 8, iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
    const T* xin = reinterpret_cast<const T*>(p.xin) + (long long)b * p.Hi * p.Wi * p.ldx;
    const T* WD = reinterpret_cast<const T*>(p.wd);
    const T* WP = reinterpret_cast<const T*>(p.wp);

    // ---- chunk loads: the halo tile's 16-byte pieces, the chunk's taps, the 1x1's operand fragments; issued a chunk ahead
    u32x4 raw[NPJ];
    unsigned tapr[3];
    v8 wf[4];
#define RIPM_FETCH(KC)                                                                                                                   \
    {                                                                                                                                    \
        _Pragma("unroll") for (int j = 0; j < NPJ; ++j) {                                                                                \
            const int pc = tid + 256 * j, px = pc >> 3, v = pc & 7;                                                                      \
            const int iy = iy0 + px / TWI, ix = ix0 + px % TWI;                                                                          \
            const bool ok = pc < NPX * 8 && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;                              \
            const T* src = xin + (long long)((ok ? iy : 0) * p.Wi + (ok ? ix : 0)) * p.ldx + (KC) * 64 + v * 8;                          \
            raw[j] = *reinterpret_cast<const u32x4*>(src);                                                                               \
        }                                                                                                                                \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                                                  \
            const int i = tid + 256 * j;                     /* < 576 = 64 channels x 9 taps */                                          \
            tapr[j] = (unsigned)*reinterpret_cast<const unsigned short*>(WD + (long long)(KC) * 64 * 9 + (i < 576 ? i : 0));             \
        }                                                                                                                                \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                                 \
            wf[ks] = *reinterpret_cast<const v8*>(WP + (long long)(js * 64 + (wave >> 1) * 32 + l31) * C + (KC) * 64 + ks * 16 + hh * 8); \
    }
    RIPM_FETCH(0);

    // ---- BatchNorm of the input: fold the producer's per-tile sums (training) or read the running statistics
    if (p.bn_in) {
        const T* gm = reinterpret_cast<const T*>(p.gamma);
        const T* bt = reinterpret_cast<const T*>(p.beta);
        if (p.training) {
            constexpr int NQ = C / 4, L = 256 / NQ;          // channel quads; chunk lanes per quad
            const int q = tid % NQ, ln = tid / NQ;
            // fp64 sums and difference: the producer's shift (its running mean) may lie many standard deviations off the batch mean
            // (norm.hip, bn_fold_partials).  The fold's scratch is 2 * L * C doubles: `red` and the head of in_t, which nobody writes
            // before the barrier behind this block.
            double* redd = reinterpret_cast<double*>(red);
            double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
            if (ln < L) {
                const float* p1 = p.part_in + C + 4 * q;
                const float* p2 = p1 + (long long)p.chunks_in * C;
                for (int k = ln; k < p.chunks_in; k += 8 * L) {
                    float4 a[8], c[8];
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        const int kk = k + m * L, kc_ = kk < p.chunks_in ? kk : ln;
                        a[m] = *reinterpret_cast<const float4*>(p1 + (long long)kc_ * C);
                        c[m] = *reinterpret_cast<const float4*>(p2 + (long long)kc_ * C);
                        if (kk >= p.chunks_in) { a[m] = make_float4(0.f, 0.f, 0.f, 0.f); c[m] = a[m]; }
                    }
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        s1[0] += (double)a[m].x; s1[1] += (double)a[m].y; s1[2] += (double)a[m].z; s1[3] += (double)a[m].w;
                        s2[0] += (double)c[m].x; s2[1] += (double)c[m].y; s2[2] += (double)c[m].z; s2[3] += (double)c[m].w;
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { redd[ln * C + 4 * q + u] = s1[u]; redd[L * C + ln * C + 4 * q + u] = s2[u]; }
            }
            __syncthreads();
            for (int c = tid; c < C; c += 256) {
                double a = 0.0, d = 0.0;
#pragma unroll
                for (int m = 0; m < L; ++m) { a += redd[m * C + c]; d += redd[L * C + m * C + c]; }
                const double nd = (double)p.rows_in, m1d = a / nd;
                const float n = (float)p.rows_in, m1 = (float)m1d, var = (float)fmax(d / nd - m1d * m1d, 0.0);
                const float mean = (float)((double)p.part_in[c] + m1d), rstd = rsqrtf(var + p.eps);
                const float g = ldf<T>(gm + c);
                sc[c] = rstd * g; sh[c] = ldf<T>(bt + c) - mean * rstd * g;
                if (tile == 0 && js == 0) {
                    p.save_mean[c] = mean; p.save_rstd[c] = rstd;
                    const float unb = p.rows_in > 1 ? var * n / (n - 1.f) : var;
                    p.rmean[c] = (1.f - p.momentum) * p.rmean[c] + p.momentum * mean;
                    p.rvar[c] = (1.f - p.momentum) * p.rvar[c] + p.momentum * unb;
                }
            }
        } else {
            for (int c = tid; c < C; c += 256) {
                const float rstd = rsqrtf(p.rvar[c] + p.eps), g = ldf<T>(gm + c);
                sc[c] = rstd * g; sh[c] = ldf<T>(bt + c) - p.rmean[c] * rstd * g;
            }
        }
    }
    __syncthreads();

    tc_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int kc = 0; kc < NC; ++kc) {
        if (kc) RP_BARRIER();                             // the previous chunk's MFMAs have read a_t, its depthwise pass in_t
        // park the chunk: input pieces (normalised + activated when bn_in, zero outside the image), taps
        v8 wcur[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wcur[ks] = wf[ks];
#pragma unroll
        for (int j = 0; j < NPJ; ++j) {
            const int pc = tid + 256 * j, px = pc >> 3, v = pc & 7;
            if (pc >= NPX * 8) continue;
            const int py = px / TWI, pxx = px % TWI, iy = iy0 + py, ix = ix0 + pxx;
            const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            u32x4 r = raw[j];
            if (p.bn_in) {
                float f[8];
                rp_unpack<T>(r, f);
                const int c0 = kc * 64 + v * 8;
#pragma unroll
                for (int u = 0; u < 8; ++u) f[u] = hswish_f(f[u] * sc[c0 + u] + sh[c0 + u]);
                r = rp_pack<T>(f);
                // the normalised map leaves from the workgroup that owns the pixel (the tile's interior; S = 1 on this path) and this chunk
                if (ok && S == 1 && py >= 1 && py <= 8 && pxx >= 1 && pxx <= 8 && (kc % (int)gridDim.y) == js)
                    *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(p.xnorm) + ((long long)(b * p.Hi + iy) * p.Wi + ix) * p.ldn + c0) = r;
            }
            if (!ok) r = u32x4{0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4*>(in_t + px * PT + v * 8) = r;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int i = tid + 256 * j;
            if (i < 576) { const int ch = i / 9, tap = i - ch * 9; wt[tap * 64 + ch] = rp_cvt16<T>(tapr[j]); }
        }
        if (kc + 1 < NC) RIPM_FETCH(kc + 1);
        RP_BARRIER();
        // depthwise 3x3 on the chunk: thread = (output pixel, 8 channels) x 2
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int it = tid + 256 * j, opx = it >> 3, v = it & 7, oy = opx >> 3, ox = opx & 7;
            float a8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a8[u] = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    float x8[8];
                    rp_unpack<T>(*reinterpret_cast<const u32x4*>(in_t + ((oy * S + ky) * TWI + ox * S + kx) * PT + v * 8), x8);
                    const float* w = wt + (ky * 3 + kx) * 64 + v * 8;
                    const float4 w0 = *reinterpret_cast<const float4*>(w), w1 = *reinterpret_cast<const float4*>(w + 4);
                    a8[0] = fmaf(x8[0], w0.x, a8[0]); a8[1] = fmaf(x8[1], w0.y, a8[1]); a8[2] = fmaf(x8[2], w0.z, a8[2]); a8[3] = fmaf(x8[3], w0.w, a8[3]);
                    a8[4] = fmaf(x8[4], w1.x, a8[4]); a8[5] = fmaf(x8[5], w1.y, a8[5]); a8[6] = fmaf(x8[6], w1.z, a8[6]); a8[7] = fmaf(x8[7], w1.w, a8[7]);
                }
            const u32x4 yv = rp_pack<T>(a8);
            *reinterpret_cast<u32x4*>(a_t + opx * PT + v * 8) = yv;
            const int gy = oy0 + oy, gx = ox0 + ox;
            if (gy < p.Ho && gx < p.Wo && (kc % (int)gridDim.y) == js)
                *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(p.y) + ((long long)(b * p.Ho + gy) * p.Wo + gx) * p.ldy + kc * 64 + v * 8) = yv;
        }
        RP_BARRIER();
        // 1x1: D^T[32 output channels x 32 pixels] += W[.., chunk] y^T
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            acc = TcHalf<T>::mfma(wcur[ks], *reinterpret_cast<const v8*>(a_t + ((wave & 1) * 32 + l31) * PT + ks * 16 + hh * 8), acc);
    }
    RP_BARRIER();
    // ---- z tile through LDS (rounded to the storage type first: the statistics are those of the values the next reader sees)
    {
        const int tok = (wave & 1) * 32 + l31;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int ch = (wave >> 1) * 32 + 8 * gq + 4 * hh;
            *reinterpret_cast<uint2*>(a_t + tok * PT + ch) = make_uint2(pack2<T>(acc[4 * gq], acc[4 * gq + 1]), pack2<T>(acc[4 * gq + 2], acc[4 * gq + 3]));
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int it = tid + 256 * j, opx = it >> 3, v = it & 7, gy = oy0 + (opx >> 3), gx = ox0 + (opx & 7);
        if (gy < p.Ho && gx < p.Wo)
            *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(p.z) + ((long long)(b * p.Ho + gy) * p.Wo + gx) * p.ldz + js * 64 + v * 8) =
                *reinterpret_cast<const u32x4*>(a_t + opx * PT + v * 8);
    }
    // shifted sums of the tile per channel: thread = (channel, quarter of the 64 pixels)
    {
        const int ch = tid & 63, part = tid >> 6, cg = js * 64 + ch;
        const float shf = p.shift_out ? p.shift_out[cg] : 0.f;
        float s1 = 0.f, s2 = 0.f;
        for (int k = 0; k < 16; ++k) {
            const int opx = part * 16 + k, gy = oy0 + (opx >> 3), gx = ox0 + (opx & 7);
            if (gy < p.Ho && gx < p.Wo) { const float d = ldf<T>(a_t + opx * PT + ch) - shf; s1 += d; s2 += d * d; }
        }
        red[part * 64 + ch] = s1; red[256 + part * 64 + ch] = s2;
        __syncthreads();
        if (tid < 64) {
            const int T_ = gridDim.x;
            const float a = red[ch] + red[64 + ch] + red[128 + ch] + red[192 + ch], d = red[256 + ch] + red[320 + ch] + red[384 + ch] + red[448 + ch];
            p.part_out[C + (long long)tile * C + cg] = a;
            p.part_out[C + ((long long)T_ + tile) * C + cg] = d;
            if (tile == 0) p.part_out[cg] = shf;
        }
    }
}

template <int C, int S> constexpr size_t ripm_smem() {
    return sizeof(float) * (2 * C + 9 * 64 + 2048) + 2 * ((size_t)(8 * S + 2) * (8 * S + 2) * 72 + 64 * 72);
}

template <typename T, int C, int S> int ripm_launch(const RipmFwdDev& p, hipStream_t s) {
    constexpr size_t smem = ripm_smem<C, S>();
    static_assert(smem <= 160 * 1024, "LDS");
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)ripm_fwd_kernel<T, C, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((ripm_fwd_kernel<T, C, S>), dim3(p.B * p.tilesH * p.tilesW, C / 64), dim3(256), smem, s, p);
    return tc_launch_status();
}

}  // namespace

// C = 320 (stage 4: 7 x 7 maps, 16 tiles x 5 output slices = 80 workgroups, five 64-channel chunks each) is built and tested but measured
// SLOWER than the three launches it replaces (18.9 us per step against 17): the chunk loop is issue-bound for a lone wave per SIMD and the
// depthwise pass is repeated by the five output slices of a tile.  TC_RIPM_C320=1 enables it.
extern "C" int tc_ripm_supported(int C, int dtype) {
    static const bool wide = getenv("TC_RIPM_C320") && atoi(getenv("TC_RIPM_C320")) != 0;
    return (dtype == TC_BF16 || dtype == TC_F16) && (C == 64 || C == 128 || (C == 320 && wide));
}
/* tiles of a step's output map = row chunks of its statistics (tc_bn_fwd's stats_chunks) */
extern "C" int tc_ripm_tiles(int B, int Ho, int Wo) { return B * ((Ho + 7) / 8) * ((Wo + 7) / 8); }

extern "C" int tc_ripm_fwd(const void* xin, int ldx, int bn_in, const float* part_in, int chunks_in, const void* gamma, const void* beta,
                           float* running_mean, float* running_var, float* save_mean, float* save_rstd, float eps, float momentum,
                           int training, void* xnorm, int ldn, const void* wd, const void* wp, void* y, int ldy, void* z, int ldz,
                           float* part_out, const float* shift_out, int B, int Hi, int Wi, int C, int stride, int dtype, void* stream) {
    if (!xin || !wd || !wp || !y || !z || !part_out || B <= 0 || Hi <= 0 || Wi <= 0 || (stride != 1 && stride != 2)) return TC_ERR_ARG;
    if (!tc_ripm_supported(C, dtype)) return TC_ERR_UNSUPPORTED;
    if (bn_in && (stride != 1 || !gamma || !beta || !running_mean || !running_var || !xnorm || (training && (!part_in || chunks_in <= 0 || !save_mean || !save_rstd))))
        return TC_ERR_ARG;
    if (ldx % 8 || ldy % 8 || ldz % 8 || (bn_in && ldn % 8) || (((uintptr_t)xin | (uintptr_t)y | (uintptr_t)z | (uintptr_t)wp | (bn_in ? (uintptr_t)xnorm : 0)) & 15))
        return TC_ERR_ARG;
    RipmFwdDev p;
    p.xin = xin; p.wd = wd; p.wp = wp; p.gamma = gamma; p.beta = beta; p.xnorm = xnorm; p.y = y; p.z = z; p.part_out = part_out;
    p.shift_out = shift_out; p.part_in = part_in; p.rmean = running_mean; p.rvar = running_var; p.save_mean = save_mean; p.save_rstd = save_rstd;
    p.eps = eps; p.momentum = momentum; p.chunks_in = chunks_in; p.rows_in = B * Hi * Wi; p.training = training; p.bn_in = bn_in;
    p.ldx = ldx; p.ldn = ldn; p.ldy = ldy; p.ldz = ldz; p.B = B; p.Hi = Hi; p.Wi = Wi;
    p.Ho = (Hi - 1) / stride + 1; p.Wo = (Wi - 1) / stride + 1; p.tilesH = (p.Ho + 7) / 8; p.tilesW = (p.Wo + 7) / 8;
    hipStream_t s = (hipStream_t)stream;
#define RP(TT) \
    if (stride == 1) { if (C == 64) return ripm_launch<TT, 64, 1>(p, s); if (C == 128) return ripm_launch<TT, 128, 1>(p, s); return ripm_launch<TT, 320, 1>(p, s); } \
    else { if (C == 64) return ripm_launch<TT, 64, 2>(p, s); if (C == 128) return ripm_launch<TT, 128, 2>(p, s); return ripm_launch<TT, 320, 2>(p, s); }
    if (dtype == TC_BF16) { RP(bf16_t) }
    RP(f16_t)
#undef RP
}
