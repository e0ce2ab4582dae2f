// What does ds_read_b64_tr_b16 return?  lds[i] = i; lane l passes the address of elements [row(l)*PITCH + 4*(l&3) ...]
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    // group g: block rows g*4 + (i>>2) of a [16][100]-pitch matrix, cols 4*(i&3)..+3
    const int addr = (g * 4 + (i >> 2)) * 100 + 4 * (i & 3);
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(lds + addr));
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int e = 0; e < 4; ++e) printf(" %4d(r%d,c%d)", h[l*4+e], h[l*4+e]/100, h[l*4+e]%100); printf("\n"); }
    return 0;
}
