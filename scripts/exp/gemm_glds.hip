// Prototype (round 6): C[M x N] = A[M x K] B[N x K]^T on 16-bit operands with the K loop fed by LDS-DMA (global_load_lds_dwordx4) instead of
// register staging -- four slab buffers, three slabs in flight behind ONE raw barrier per slab, counted vmcnt, XOR-swizzled unpadded tiles
// (the swizzle is applied to the SOURCE address: the DMA writes lane-linear).  Question it answers: how much of the 2.7 k cycles per 64-deep
// slab that csrc/gemm.hip's register-staged loop spends at ~1.5 workgroups per CU is latency that deeper prefetch removes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o scripts/exp/libgemm_glds.so scripts/exp/gemm_glds.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const f2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));
}

constexpr int BM = 64, BN = 64, BK = 64, NBUF = 4, TILE_B = 64 * 128;      // a tile: 64 rows x 128 bytes, unpadded

// one 1-KB piece (8 rows x 128 B) of a K-contiguous operand tile: lane -> (row = 8 piece + lane / 8, LDS chunk = lane % 8), source chunk = LDS chunk ^ (row & 7)
__device__ __forceinline__ void dma_piece(const bf16_t* base, int ld, int row0, int nrows, int k0, unsigned char* tile, int piece, int lane) {
    const int row = 8 * piece + (lane >> 3), cl = lane & 7, cs = cl ^ (row & 7);
    const int gr = min(row0 + row, nrows - 1);
    const bf16_t* src = base + (long long)gr * ld + k0 + 8 * cs;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(tile + piece * 1024), 16, 0, 0);
}

template <int DEPTH>
__global__ __launch_bounds__(256, 2) void gemm_glds_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, bf16_t* __restrict__ C, int ldc,
                                                           int M, int N, int K) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];      // [NBUF][A tile | B tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1, h = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nslab = K / BK;
    auto issue = [&](int s) __attribute__((always_inline)) {
        unsigned char* buf = smem + (s % NBUF) * 2 * TILE_B;
        const int k0 = s * BK;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            dma_piece(A, lda, m0, M, k0, buf, 2 * wave + j, lane);
            dma_piece(B, ldb, n0, N, k0, buf + TILE_B, 2 * wave + j, lane);
        }
    };
    // fragment addresses: lane (row = 32 * w + l31, k-slice 8 h + 16 kk) -> chunk c = h + 2 kk, swizzled by the row
    const int arow = wr * 32 + l31, brow = wc * 32 + l31;
    for (int s = 0; s < DEPTH && s < nslab; ++s) issue(s);
    for (int s = 0; s < nslab; ++s) {
        // this wave's pieces of slab s have landed when at most 4 * (slabs issued after s) of its DMAs are outstanding
        const int ahead = min(nslab - 1 - s, DEPTH - 1);
        if (ahead >= 3) __builtin_amdgcn_s_waitcnt(0x0F70 | (12 & 0xF) | ((12 >> 4) << 14));   // vmcnt(12): lgkm / exp untouched
        else if (ahead == 2) __builtin_amdgcn_s_waitcnt(0x0F70 | 8);
        else if (ahead == 1) __builtin_amdgcn_s_waitcnt(0x0F70 | 4);
        else __builtin_amdgcn_s_waitcnt(0x0F70 | 0);
        __builtin_amdgcn_s_barrier();                                   // every wave's pieces of slab s are in LDS; buffer (s - 1) % NBUF is free
        if (s + DEPTH < nslab) issue(s + DEPTH);                        // (DEPTH <= NBUF - 1: the buffer it fills was read in iteration s - 1)
        const unsigned char* at = smem + (s % NBUF) * 2 * TILE_B;
        const unsigned char* bt = at + TILE_B;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int c = h + 2 * kk;
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(at + arow * 128 + ((c ^ (arow & 7)) << 4));
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(bt + brow * 128 + ((c ^ (brow & 7)) << 4));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc, 0, 0, 0);      // D^T: lane = output row
        }
    }
    // D^T tile: lane holds row m = wr * 32 + l31, columns n = wc * 32 + 8 g + 4 h + (0..3)
    const int row = m0 + wr * 32 + l31;
    if (row < M) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = n0 + wc * 32 + 8 * g + 4 * h;
            if (col < N) *reinterpret_cast<uint2*>(C + (long long)row * ldc + col) = make_uint2(pack2bf(acc[4 * g], acc[4 * g + 1]), pack2bf(acc[4 * g + 2], acc[4 * g + 3]));
        }
    }
}

extern "C" int gemm_glds(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, int depth, void* stream) {
    if (K % BK || N % 4) return 1;
    const dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM);
    const size_t smem = (size_t)NBUF * 2 * TILE_B;
    hipStream_t s = (hipStream_t)stream;
    if (depth >= 3) hipLaunchKernelGGL(gemm_glds_kernel<3>, grid, dim3(256), smem, s, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, (bf16_t*)C, ldc, M, N, K);
    else if (depth == 2) hipLaunchKernelGGL(gemm_glds_kernel<2>, grid, dim3(256), smem, s, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, (bf16_t*)C, ldc, M, N, K);
    else hipLaunchKernelGGL(gemm_glds_kernel<1>, grid, dim3(256), smem, s, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, (bf16_t*)C, ldc, M, N, K);
    return (int)hipGetLastError();
}
