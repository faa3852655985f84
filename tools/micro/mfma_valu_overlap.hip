// Microbenchmark: which VALU instructions execute in the shadow of an MFMA on a gfx950 SIMD?
// Every wave (two per SIMD) runs ITERS x 8 x { 1 MFMA 32x32x16 bf16 ; NV x one VALU opcode (inline asm, independent
// registers) }.  Reported: time with the MFMAs alone, the VALU alone, and both; "hidden" = share of the VALU time that
// disappears behind the MFMAs.
// build: hipcc --offload-arch=gfx950 -O3 -w -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define ITERS 512
#define NV 6

template <int OP>
__device__ __forceinline__ void valu(float& x, f32x2& p, float c) {
    if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
    if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(f32x2{c, c}));
    if (OP == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(f32x2{c, c}));
    if (OP == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(c));
    if (OP == 4) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(x));
    if (OP == 5) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(c));
    if (OP == 6) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(x));
    if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(c));
}

template <int OP, bool DO_M, bool DO_V>
__global__ __launch_bounds__(512) void k(float* out, float seed) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = static_cast<__bf16>(seed + e); b[e] = static_cast<__bf16>(seed - e); }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float v[NV];
    f32x2 pv[NV];
    for (int j = 0; j < NV; ++j) { v[j] = seed + j + threadIdx.x; pv[j] = f32x2{seed + j, seed - j}; }
    float c = seed * 1.0001f;
    asm volatile("" : "+v"(c));
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (DO_M) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            if (DO_V) {
#pragma unroll
                for (int j = 0; j < NV; ++j) valu<OP>(v[j], pv[j], c);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int j = 0; j < NV; ++j) s += v[j] + pv[j][0] + pv[j][1];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int OP, bool DO_M, bool DO_V>
static float run(float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP, DO_M, DO_V>), dim3(256), dim3(512), 0, 0, out, 1.0f);     // 8 waves per CU = 2 per SIMD
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<OP, DO_M, DO_V>), dim3(256), dim3(512), 0, 0, out, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 10 * 1e3f;
}

template <int OP>
static void report(float* out, const char* name, float t_m) {
    const float t_v = run<OP, false, true>(out), t_b = run<OP, true, true>(out);
    printf("%-20s valu alone %6.1f us   with MFMAs %6.1f us   hidden %3.0f %%\n", name, t_v, t_b, 100.f * (1.f - (t_b - t_m) / t_v));
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    const float t_m = run<0, true, false>(out);
    printf("MFMA alone %.1f us (2 waves per SIMD x %d x 8 MFMAs 32x32x16 bf16; %d VALU per MFMA below)\n", t_m, ITERS, NV);
    report<0>(out, "v_fma_f32", t_m);
    report<1>(out, "v_pk_fma_f32", t_m);
    report<2>(out, "v_pk_add_f32", t_m);
    report<3>(out, "v_cvt_pk_bf16_f32", t_m);
    report<4>(out, "v_lshlrev_b32", t_m);
    report<5>(out, "v_sub_f32", t_m);
    report<6>(out, "v_and_b32", t_m);
    report<7>(out, "v_cndmask_b32", t_m);
    return 0;
}
