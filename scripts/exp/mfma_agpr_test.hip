// v_mfma with the A operand, the B operand, or both in AGPRs: same result as with both in VGPRs?  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(64) void k(float* out) {
    const int lane = threadIdx.x;
    unsigned a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = 0x3f803c00u + 0x00010001u * ((lane * 7 + i * 3) & 63); b[i] = 0x3f003e80u + 0x00010001u * ((lane * 5 + i) & 63); }
    f32x16 r0, r1, r2, r3;
    asm volatile(
        "v_mov_b32 v20, %4\n v_mov_b32 v21, %5\n v_mov_b32 v22, %6\n v_mov_b32 v23, %7\n"
        "v_mov_b32 v24, %8\n v_mov_b32 v25, %9\n v_mov_b32 v26, %10\n v_mov_b32 v27, %11\n"
        "v_accvgpr_write_b32 a20, %4\n v_accvgpr_write_b32 a21, %5\n v_accvgpr_write_b32 a22, %6\n v_accvgpr_write_b32 a23, %7\n"
        "v_accvgpr_write_b32 a24, %8\n v_accvgpr_write_b32 a25, %9\n v_accvgpr_write_b32 a26, %10\n v_accvgpr_write_b32 a27, %11\n"
        "s_nop 7\n"
        "v_mfma_f32_32x32x16_bf16 %0, v[20:23], v[24:27], 0\n"
        "v_mfma_f32_32x32x16_bf16 %1, a[20:23], v[24:27], 0\n"
        "v_mfma_f32_32x32x16_bf16 %2, v[20:23], a[24:27], 0\n"
        "v_mfma_f32_32x32x16_bf16 %3, a[20:23], a[24:27], 0\n"
        "s_nop 15\n s_nop 15\n"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3])
        : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27");
    float e1 = 0, e2 = 0, e3 = 0;
    for (int i = 0; i < 16; ++i) { e1 = fmaxf(e1, fabsf(r1[i] - r0[i])); e2 = fmaxf(e2, fabsf(r2[i] - r0[i])); e3 = fmaxf(e3, fabsf(r3[i] - r0[i])); }
    out[lane * 4] = e1; out[lane * 4 + 1] = e2; out[lane * 4 + 2] = e3; out[lane * 4 + 3] = r0[0];
}
int main() {
    float *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    float m1 = 0, m2 = 0, m3 = 0;
    for (int i = 0; i < 64; ++i) { m1 = fmaxf(m1, h[4 * i]); m2 = fmaxf(m2, h[4 * i + 1]); m3 = fmaxf(m3, h[4 * i + 2]); }
    printf("max |diff| vs VGPR/VGPR: A in AGPR %g, B in AGPR %g, both %g (r0[0] lane0 = %g)\n", m1, m2, m3, h[3]);
    return 0;
}
