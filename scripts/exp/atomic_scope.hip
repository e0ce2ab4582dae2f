// fp32 atomic adds at device (agent) scope vs workgroup scope on gfx950, and the workgroup -> XCD dispatch order.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o atomic_scope atomic_scope.hip && ./atomic_scope
// Pattern of a split-K weight-gradient epilogue: `splits` workgroups add the same 64 x 64 fp32 tile (4096 words, 16 per thread); tiles x splits
// workgroups per launch.  Device scope: the atomic executes at the memory side (eight L2s).  Workgroup scope: in the issuing XCD's L2 -- coherent
// only if every workgroup that touches a tile runs on the same XCD, which the XCD-pinned numbering below provides when dispatch is round-robin.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int SCOPE, bool PIN>
__global__ __launch_bounds__(256) void add_tiles(float* __restrict__ C, int tiles, int splits, unsigned* __restrict__ xcc_bad) {
    int lin = blockIdx.x, tile, ks;
    if (PIN) {                      // all splits of a tile on XCD (tile % 8): workgroup lin runs on XCD lin % 8
        const int x = lin & 7, slot = lin >> 3, per = tiles / 8;      // tiles % 8 == 0 here
        tile = (slot % per) * 8 + x; ks = slot / per;
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        if ((id & 0xf) != (unsigned)x && threadIdx.x == 0) atomicAdd(xcc_bad, 1u);
    } else { tile = lin % tiles; ks = lin / tiles; }
    (void)ks;
    float* t = C + (long long)tile * 4096;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float* p = t + i * 256 + threadIdx.x;
        if (SCOPE == 0) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
template <int SCOPE, bool PIN> float run(float* C, int tiles, int splits, unsigned* bad, const char* what) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipMemset(C, 0, (size_t)tiles * 4096 * 4);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((add_tiles<SCOPE, PIN>), dim3(tiles * splits), dim3(256), 0, 0, C, tiles, splits, bad);
    hipMemset(C, 0, (size_t)tiles * 4096 * 4); hipMemset(bad, 0, 4);
    hipEventRecord(a);
    const int reps = 20;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((add_tiles<SCOPE, PIN>), dim3(tiles * splits), dim3(256), 0, 0, C, tiles, splits, bad);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<float> h((size_t)tiles * 4096); unsigned hb;
    hipMemcpy(h.data(), C, h.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    size_t wrong = 0; for (float v : h) wrong += v != (float)(reps * splits);
    printf("  %-44s %7.2f us/launch  %6.2f G atomics/s   wrong words %zu   XCC mismatches %u\n", what, ms * 1e3 / reps,
           (double)tiles * splits * 4096 / (ms * 1e-3 / reps) / 1e9, wrong, hb);
    return ms;
}
int main() {
    float* C; unsigned* bad; hipMalloc(&C, 64 << 20); hipMalloc(&bad, 4);
    for (int cfg = 0; cfg < 3; ++cfg) {
        const int tiles = cfg == 0 ? 48 : (cfg == 1 ? 256 : 16), splits = cfg == 0 ? 12 : (cfg == 1 ? 3 : 49);
        printf("tiles %d x splits %d (%.1f M atomics per launch)\n", tiles, splits, tiles * splits * 4096 / 1e6);
        run<0, false>(C, tiles, splits, bad, "device scope, natural numbering");
        run<0, true>(C, tiles, splits, bad, "device scope, XCD-pinned numbering");
        run<1, true>(C, tiles, splits, bad, "workgroup scope, XCD-pinned numbering");
        run<1, false>(C, tiles, splits, bad, "workgroup scope, natural numbering (expected wrong)");
    }
    return 0;
}
