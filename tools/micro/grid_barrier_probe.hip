// Cost and correctness of a hand-rolled grid barrier on gfx950 (8 XCDs, one L2 each): NWG resident workgroups run NB rounds of
//   write one partial per workgroup (+ DIRTY bytes of other stores, to load the L2 write-back a release has to do)
//   -> barrier -> read ALL partials and compare with the value every workgroup must have written this round.
// Prints us per round and the number of stale reads (must be 0).  Build: hipcc --offload-arch=gfx950 -O3 -o gbp grid_barrier_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Bar { unsigned count; unsigned gen; };

__device__ __forceinline__ void grid_barrier(Bar* b, unsigned n) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned g = __hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __atomic_thread_fence(__ATOMIC_RELEASE);                       // this workgroup's stores -> visible at agent scope
        if (__hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n - 1) {
            __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&b->gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(2);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
}

// variant B: no agent-scope fences (no L2 write-back / invalidate): the exchanged data itself moves with agent-scope relaxed atomic
// stores / loads (write-through, L2-bypassing reads), ordered against the barrier's atomics by workgroup-scope fences (s_waitcnt)
__device__ __forceinline__ void grid_barrier_nofence(Bar* b, unsigned n) {
    __atomic_thread_fence(__ATOMIC_RELEASE);      // replaced below by a workgroup-scope fence via the builtin
}
__device__ __forceinline__ void grid_barrier_b(Bar* b, unsigned n) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned g = __hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n - 1) {
            __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __hip_atomic_fetch_add(&b->gen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void probe_b(Bar* bar, float* partial, float* dirty, int dirty_f4, int nb, unsigned* bad) {
    const unsigned n = gridDim.x;
    unsigned wrong = 0;
    for (int it = 0; it < nb; ++it) {
        float* p = partial + (it & 1) * n * 64;
        if (threadIdx.x < 64)
            __hip_atomic_store(&p[blockIdx.x * 64 + threadIdx.x], static_cast<float>(it * 7 + (blockIdx.x & 15) + threadIdx.x), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        for (int i = threadIdx.x; i < dirty_f4; i += 256)
            reinterpret_cast<float4*>(dirty)[static_cast<size_t>(blockIdx.x) * dirty_f4 + i] = make_float4(it, it, it, it);
        grid_barrier_b(bar, n);
        const int c = threadIdx.x & 63;
        for (unsigned w = threadIdx.x >> 6; w < n; w += 4)
            wrong += __hip_atomic_load(&p[w * 64 + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != static_cast<float>(it * 7 + (w & 15) + c);
    }
    if (wrong) atomicAdd(bad, wrong);
}

__global__ __launch_bounds__(256) void probe(Bar* bar, float* partial, float* dirty, int dirty_f4, int nb, unsigned* bad) {
    const unsigned n = gridDim.x;
    unsigned wrong = 0;
    for (int it = 0; it < nb; ++it) {
        float* p = partial + (it & 1) * n * 64;
        if (threadIdx.x < 64) p[blockIdx.x * 64 + threadIdx.x] = static_cast<float>(it * 7 + (blockIdx.x & 15) + threadIdx.x);
        for (int i = threadIdx.x; i < dirty_f4; i += 256)
            reinterpret_cast<float4*>(dirty)[static_cast<size_t>(blockIdx.x) * dirty_f4 + i] = make_float4(it, it, it, it);
        grid_barrier(bar, n);
        // every workgroup reads the column threadIdx.x % 64 of all partial rows (4 rows in flight per thread group)
        const int c = threadIdx.x & 63;
        for (unsigned w = threadIdx.x >> 6; w < n; w += 4)
            wrong += p[w * 64 + c] != static_cast<float>(it * 7 + (w & 15) + c);
    }
    if (wrong) atomicAdd(bad, wrong);
}

// the same rounds as NB separate launches (what the barrier replaces)
__global__ __launch_bounds__(256) void probe_launch(float* partial, float* dirty, int dirty_f4, int it, unsigned* bad) {
    const unsigned n = gridDim.x;
    float* p = partial + (it & 1) * n * 64;
    const float* q = partial + ((it + 1) & 1) * n * 64;
    unsigned wrong = 0;
    const int c = threadIdx.x & 63;
    if (it > 0)
        for (unsigned w = threadIdx.x >> 6; w < n; w += 4) wrong += q[w * 64 + c] != static_cast<float>((it - 1) * 7 + (w & 15) + c);
    if (threadIdx.x < 64) p[blockIdx.x * 64 + threadIdx.x] = static_cast<float>(it * 7 + (blockIdx.x & 15) + threadIdx.x);
    for (int i = threadIdx.x; i < dirty_f4; i += 256)
        reinterpret_cast<float4*>(dirty)[static_cast<size_t>(blockIdx.x) * dirty_f4 + i] = make_float4(it, it, it, it);
    if (wrong) atomicAdd(bad, wrong);
}

int main(int argc, char** argv) {
    const int nb = 200;
    for (int nwg : {256, 512}) {
        for (int dirty_kb : {0, 16, 128}) {
            Bar* bar; float *partial, *dirty; unsigned* bad;
            hipMalloc(&bar, sizeof(Bar)); hipMemset(bar, 0, sizeof(Bar));
            hipMalloc(&partial, 2 * nwg * 64 * 4);
            hipMalloc(&dirty, static_cast<size_t>(nwg) * 128 * 1024 + 64);
            hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
            const int df4 = dirty_kb * 1024 / 16;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float ms = 0, ms2 = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(probe, dim3(nwg), dim3(256), 0, 0, bar, partial, dirty, df4, nb, bad);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            unsigned hbad = 0; hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
            hipMemset(bad, 0, 4);
            float ms3 = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(probe_b, dim3(nwg), dim3(256), 0, 0, bar, partial, dirty, df4, nb, bad);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms3, e0, e1);
            }
            unsigned hbad3 = 0; hipMemcpy(&hbad3, bad, 4, hipMemcpyDeviceToHost);
            hipMemset(bad, 0, 4);
            printf("   variant B (atomic data, no agent fences): %.2f us per round (stale reads %u)\n", ms3 * 1e3 / nb, hbad3);
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                for (int it = 0; it < nb; ++it) hipLaunchKernelGGL(probe_launch, dim3(nwg), dim3(256), 0, 0, partial, dirty, df4, it, bad);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms2, e0, e1);
            }
            unsigned hbad2 = 0; hipMemcpy(&hbad2, bad, 4, hipMemcpyDeviceToHost);
            printf("workgroups %3d  dirty %3d KB/wg: barrier round %.2f us (stale reads %u)   separate launches %.2f us per round (stale %u)\n",
                   nwg, dirty_kb, ms * 1e3 / nb, hbad, ms2 * 1e3 / nb, hbad2);
            hipFree(bar); hipFree(partial); hipFree(dirty); hipFree(bad);
        }
    }
    return 0;
}
