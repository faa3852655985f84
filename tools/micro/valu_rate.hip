// Issue cost of a few VALU opcodes on gfx950 (cycles per wave64 instruction per SIMD): WAVES waves per SIMD on every CU run
// ITERS x 32 independent instructions of one opcode; cycles = time * clock / (ITERS * 32 * waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 4096

#define BODY8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define KERNEL(NAME, ASM)                                                                  \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out) {                           \
        uint32_t r[8];                                                                     \
        for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 2654435761u + i;                  \
        uint32_t a = threadIdx.x | 1u, b = 0x00010001u, c = 0x3F803F80u;                   \
        for (int it = 0; it < ITERS; ++it) {                                               \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                \
                _Pragma("unroll") for (int i = 0; i < 8; ++i)                              \
                    asm volatile(ASM : "+v"(r[i]) : "v"(a), "v"(b), "v"(c));               \
            }                                                                              \
        }                                                                                  \
        uint32_t s = 0;                                                                    \
        for (int i = 0; i < 8; ++i) s ^= r[i];                                             \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                           \
    }

KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL(k_and, "v_and_b32 %0, %0, %1")
KERNEL(k_pk_sub_u16, "v_pk_sub_u16 %0, %0, %1")
KERNEL(k_pk_min_u16, "v_pk_min_u16 %0, %0, %2")
KERNEL(k_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
KERNEL(k_pk_add_f32x, "v_pk_mul_lo_u16 %0, %0, %1")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_cmp_cnd, "v_cmp_eq_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %2, %1, vcc")
KERNEL(k_cvt_pk_bf16, "v_cvt_pk_bf16_f32 %0, %0, %1")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 16, %1")
KERNEL(k_max3, "v_max3_u32 %0, %0, %1, %2")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 5")
KERNEL(k_sub_f32, "v_sub_f32 %0, %0, %1")

typedef void (*kfn)(uint32_t*);
int main() {
    uint32_t* d;
    hipMalloc(&d, 4096 * 256 * 4);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const double clk = prop.clockRate * 1e3;     // Hz
    struct { const char* name; kfn fn; int per; } ks[] = {
        {"v_fma_f32", k_fma, 1}, {"v_add_u32", k_add_u32, 1}, {"v_and_b32", k_and, 1}, {"v_pk_sub_u16", k_pk_sub_u16, 1},
        {"v_pk_min_u16", k_pk_min_u16, 1}, {"v_pk_mad_u16", k_pk_mad_u16, 1}, {"v_pk_mul_lo_u16", k_pk_add_f32x, 1},
        {"v_cndmask_b32", k_cndmask, 1}, {"v_cmp_eq_u32 + v_cndmask", k_cmp_cnd, 2}, {"v_cvt_pk_bf16_f32", k_cvt_pk_bf16, 1},
        {"v_perm_b32", k_perm, 1}, {"v_lshl_or_b32", k_lshl_or, 1}, {"v_max3_u32", k_max3, 1}, {"v_bfe_u32", k_bfe, 1},
        {"v_sub_f32", k_sub_f32, 1}};
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = prop.multiProcessorCount * wps;     // 256 threads = 4 waves = one wave per SIMD per block
        printf("---- %d wave(s) per SIMD (clock %.2f GHz)\n", wps, clk / 1e9);
        for (auto& k : ks) {
            hipEvent_t a, b;
            hipEventCreate(&a); hipEventCreate(&b);
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, d);
            hipDeviceSynchronize();
            hipEventRecord(a);
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, d);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double cyc = ms * 1e-3 * clk / (double(ITERS) * 32 * wps * k.per);
            printf("%-28s %6.2f cycles per instruction per SIMD\n", k.name, cyc);
        }
    }
    return 0;
}
