// Dependent-accumulator latency of v_mfma_f32_32x32x16_bf16 on gfx950: one wave per SIMD issues ITERS x 6 MFMAs into ONE
// accumulator tile (chain), or alternates between TWO / THREE tiles.  Prints cycles per MFMA per SIMD (2.4 GHz assumed).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define ITERS 2048
template <int CH>
__global__ __launch_bounds__(256) void k(float* out) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = static_cast<__bf16>(0.001f * (threadIdx.x + e)); b[e] = static_cast<__bf16>(0.002f * (threadIdx.x - e)); }
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CH> void run(float* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<CH>, dim3(256), dim3(256), 0, 0, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%d interleaved accumulator tiles, one wave per SIMD: %.1f cycles per MFMA\n", CH, ms * 1e-3 * 2.4e9 / (ITERS * 6.0 * CH));
}
int main() { float* out; (void)hipMalloc(&out, 256 * 256 * 4); run<1>(out); run<2>(out); run<3>(out); return 0; }
