// Feasibility probe for persistent cooperative kernels on MI355X (round 5): cost and correctness of a grid-wide barrier + cross-XCD
// data exchange inside ONE launch.  NWG workgroups (<= 256: one per CU, all resident) run ITER rounds of
//   write a tile (own slot of a global buffer) -> grid barrier -> read the tiles of two other workgroups (other XCDs) and check them.
// Variants of how written data becomes visible to readers on other XCDs (each XCD has its own, non-coherent L2):
//   0: plain stores + __threadfence() (agent-scope release: L2 write-back) ; readers: __threadfence() (acquire: L2 invalidate)
//   1: write-through stores (sc0 sc1) ; readers: loads that bypass L2 (sc0 sc1)
//   2: write-through stores ; readers: one buffer_inv sc1 per workgroup after the barrier, then plain loads
// The barrier: every thread waits for its stores (s_waitcnt vmcnt(0)), __syncthreads, thread 0 adds 1 to a monotonic counter with a
// device-scope atomic and spins (bounded) until the counter reaches epoch * NWG, __syncthreads.
//   hipcc --offload-arch=gfx950 -O3 -o gridbar gridbar.hip && ./gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_wt(float4* p, float4 v) {
    const f32x4 w = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ float4 ld_bypass(const float4* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return make_float4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned nwg, unsigned& epoch, int* err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int ok;
    if (threadIdx.x == 0) {
        epoch += 1;
        const unsigned target = epoch * nwg;
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        ok = 1;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 4000000) { ok = 0; atomicAdd(err, 1); break; }
        }
    }
    __syncthreads();
    return ok != 0;
}

template <int MODE>
__global__ __launch_bounds__(256) void probe(float4* buf, unsigned* counter, int* err, int iters, int tile_f4, long long* cycles) {
    const int wg = blockIdx.x, nwg = gridDim.x, tid = threadIdx.x;
    unsigned epoch = 0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        // parity double buffer: readers of round it read buffer (it & 1) while writers of round it + 1 fill the other one
        float4* mine = buf + ((long long)(it & 1) * nwg + wg) * tile_f4;
        for (int i = tid; i < tile_f4; i += 256) {
            const float4 v = make_float4((float)wg, (float)it, (float)i, 1.0f);
            if (MODE == 0) mine[i] = v; else st_wt(mine + i, v);
        }
        if (MODE == 0) __threadfence();
        if (!grid_barrier(counter, nwg, epoch, err)) return;
        if (MODE == 0) __threadfence();
        if (MODE == 2) { if (tid < 64) asm volatile("buffer_inv sc1" ::: "memory"); __syncthreads(); }
        for (int k = 1; k <= 2; ++k) {
            const int other = (wg + k * 3) % nwg;               // (index l runs on XCD l % 8: +3 / +6 are other XCDs)
            const float4* theirs = buf + ((long long)(it & 1) * nwg + other) * tile_f4;
            for (int i = tid; i < tile_f4; i += 256) {
                const float4 v = MODE == 1 ? ld_bypass(theirs + i) : theirs[i];
                if (v.x != (float)other || v.y != (float)it || v.z != (float)i) atomicAdd(err + 1, 1);
            }
        }
    }
    if (tid == 0 && wg == 0) cycles[0] = clock64() - t0;
}

int main() {
    const int nwg = 196, iters = 200;
    for (int tile_kb : {8, 64}) {
        const int tile_f4 = tile_kb * 1024 / 16;
        float4* buf; unsigned* counter; int* err; long long* cyc;
        CHECK(hipMalloc(&buf, sizeof(float4) * 2ll * nwg * tile_f4));
        CHECK(hipMalloc(&counter, 64)); CHECK(hipMalloc(&err, 64)); CHECK(hipMalloc(&cyc, 64));
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipMemset(counter, 0, 64)); CHECK(hipMemset(err, 0, 64)); CHECK(hipMemset(buf, 0, sizeof(float4) * 2ll * nwg * tile_f4));
                hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
                CHECK(hipEventRecord(a));
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(nwg), dim3(256), 0, 0, buf, counter, err, iters, tile_f4, cyc);
                if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(nwg), dim3(256), 0, 0, buf, counter, err, iters, tile_f4, cyc);
                if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(nwg), dim3(256), 0, 0, buf, counter, err, iters, tile_f4, cyc);
                CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
                float ms; CHECK(hipEventElapsedTime(&ms, a, b));
                int h[2]; CHECK(hipMemcpy(h, err, 8, hipMemcpyDeviceToHost));
                if (rep == 1) printf("tile %2d KB/wg (%5.1f MB per round) mode %d: %7.2f us per round (write + barrier + read 2 tiles), barrier timeouts %d, stale reads %d\n",
                                     tile_kb, nwg * tile_kb / 1024.0, mode, ms * 1e3 / iters, h[0], h[1]);
            }
        }
        // barrier alone
        CHECK(hipMemset(counter, 0, 64)); CHECK(hipMemset(err, 0, 64));
        hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(probe<1>, dim3(nwg), dim3(256), 0, 0, buf, counter, err, iters, 0, cyc);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        printf("barrier alone: %.2f us per round\n", ms * 1e3 / iters);
        hipFree(buf); hipFree(counter); hipFree(err); hipFree(cyc);
    }
    return 0;
}
