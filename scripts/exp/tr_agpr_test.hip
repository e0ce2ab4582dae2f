// does ds_read_b64_tr_b16 / ds_read_b128 deliver the same data into AGPRs as into VGPRs?  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    __shared__ unsigned short lds[64 * 72];
    for (int i = threadIdx.x; i < 64 * 72; i += 64) lds[i] = (unsigned short)(i * 7 + 3);
    __syncthreads();
    const int lane = threadIdx.x, gi = lane & 15, gq = (lane >> 4) & 1, h = lane >> 5;
    const unsigned addr = (unsigned)(uintptr_t)lds + ((16 * h + 4 * (gi >> 2)) * 144 + 32 * gq + 8 * (gi & 3));
    unsigned v0, v1, a0, a1, b0, b1, b2, b3, c0, c1, c2, c3;
    asm volatile("ds_read_b64_tr_b16 v[10:11], %12 offset:144\n ds_read_b64_tr_b16 a[10:11], %12 offset:144\n"
                 "ds_read_b128 v[12:15], %12 offset:32\n ds_read_b128 a[12:15], %12 offset:32\n s_waitcnt lgkmcnt(0)\n"
                 "v_mov_b32 %0, v10\n v_mov_b32 %1, v11\n v_accvgpr_read_b32 %2, a10\n v_accvgpr_read_b32 %3, a11\n"
                 "v_mov_b32 %4, v12\n v_mov_b32 %5, v13\n v_mov_b32 %6, v14\n v_mov_b32 %7, v15\n"
                 "v_accvgpr_read_b32 %8, a12\n v_accvgpr_read_b32 %9, a13\n v_accvgpr_read_b32 %10, a14\n v_accvgpr_read_b32 %11, a15\n"
                 : "=v"(v0), "=v"(v1), "=v"(a0), "=v"(a1), "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3), "=v"(c0), "=v"(c1), "=v"(c2), "=v"(c3)
                 : "v"(addr) : "v10", "v11", "v12", "v13", "v14", "v15", "a10", "a11", "a12", "a13", "a14", "a15", "memory");
    out[lane] = (v0 != a0) | ((v1 != a1) << 1) | ((b0 != c0 || b1 != c1 || b2 != c2 || b3 != c3) << 2);
    out[64 + lane] = v0; out[128 + lane] = a0;
}
int main() {
    unsigned *d, h[192];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) bad += h[i] != 0;
    printf("lanes with a VGPR/AGPR mismatch: %d (flags of lane 0: %u; v0 %08x a0 %08x; lane 5: v0 %08x a0 %08x)\n", bad, h[0], h[64], h[128], h[69], h[133]);
    return 0;
}
