// Prototype of the first-layer forward product on ready-made bf16 planes (round 3):
//   Z[M, N] = A' Wf^T,  A' = [148 real columns (+12 pad) | 128 one-hot columns],  N = 148
// A' arrives as "A3": per 32-row tile, per 16-column group, 1-KB blocks in MFMA fragment order (lane (i, s) -> 16 bytes:
// 8 consecutive k of row i), three blocks (h, m, l terms) for real groups, one for the exact one-hot groups.  The
// weights arrive as "B3": the same block shape per (chunk of 2 groups, group, 32-column tile, plane), so a chunk is a
// linear 30-KB copy into LDS and every fragment read is a conflict-free ds_read_b128 at lane * 16.
// The probe packs both layouts from fp32 on the device, times the kernel and checks it against an fp64 host product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define M_ROWS 65536
#define KR 148            // real columns
#define G3 10             // 16-column groups with three planes (real, padded to 160)
#define G1 8              // one-hot groups (128 columns), one plane
#define NG (G3 + G1)
#define NBLK (3 * G3 + G1)    // 1-KB blocks per 32-row tile of A3
#define NCOL 148
#define NT 5
#define KF (KR + 128)     // columns of the fp32 operand the reference kernel multiplies

#define CHUNK_U4 (2 * NT * 3 * 64)          // uint4 per chunk of B3 (30 KB)
#define CHUNK_PAD_U4 (32 * 64)               // chunk pitch in global memory and LDS: 32 blocks of 1 KB (8 DMA pieces per wave)
#define ST_PER_THREAD ((CHUNK_U4 + 255) / 256)

struct Bf3 { __bf16 h, m, l; };
__host__ __device__ inline Bf3 split3(float x) {
    Bf3 r;
    r.h = static_cast<__bf16>(x);
    const float r1 = x - static_cast<float>(r.h);
    r.m = static_cast<__bf16>(r1);
    const float r2 = r1 - static_cast<float>(r.m);
    r.l = static_cast<__bf16>(r2);
    return r;
}

// column c of the padded layout (0 .. 287) -> column of the fp32 operand, or -1 (padding)
__host__ __device__ inline int src_col(int c) { return c < KR ? c : (c < 16 * G3 ? -1 : KR + (c - 16 * G3)); }
__host__ __device__ inline int blk_of(int g, int p) { return g < G3 ? 3 * g + p : 3 * G3 + (g - G3); }

__global__ void pack_a3(const float* __restrict__ A, __bf16* __restrict__ A3) {     // thread = (row, group, s)
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (idx >= static_cast<int64_t>(M_ROWS) * NG * 2) return;
    const int s = idx & 1, g = (idx >> 1) % NG;
    const int64_t row = (idx >> 1) / NG;
    const int64_t T = row >> 5;
    const int i = row & 31;
    __bf16 h[8], m[8], l[8];
    for (int e = 0; e < 8; ++e) {
        const int c = src_col(16 * g + 8 * s + e);
        const Bf3 t = split3(c >= 0 ? A[row * KF + c] : 0.f);
        h[e] = t.h; m[e] = t.m; l[e] = t.l;
    }
    const int np = g < G3 ? 3 : 1;
    for (int p = 0; p < np; ++p) {
        __bf16* d = A3 + ((T * NBLK + blk_of(g, p)) * 64 + (s * 32 + i)) * 8;
        for (int e = 0; e < 8; ++e) d[e] = p == 0 ? h[e] : (p == 1 ? m[e] : l[e]);
    }
}

// B3: [chunk (NG / 2)][group in chunk (2)][tile (NT)][plane (3)][64 lanes][8]
__global__ void pack_b3(const float* __restrict__ W, __bf16* __restrict__ B3) {     // thread = (group, tile, lane)
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= NG * NT * 64) return;
    const int lane = idx & 63, t = (idx >> 6) % NT, g = idx / (64 * NT);
    const int j = lane & 31, s = lane >> 5, n = 32 * t + j;
    for (int p = 0; p < 3; ++p)
        for (int e = 0; e < 8; ++e) {
            const int c = src_col(16 * g + 8 * s + e);
            const Bf3 v = split3((c >= 0 && n < NCOL) ? W[n * KF + c] : 0.f);
            B3[static_cast<size_t>(g >> 1) * (CHUNK_PAD_U4 * 8) + (((g & 1) * NT + t) * 3 + p) * 512 + lane * 8 + e] = p == 0 ? v.h : (p == 1 ? v.m : v.l);
        }
}


template <int EPI>
__global__ __launch_bounds__(256, 2) void rows3_fwd(const uint4* __restrict__ A3, const uint4* __restrict__ B3,
                                                    float* __restrict__ Z, int ldz, float* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];      // [2][CHUNK_U4]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, s = lane >> 5;
    const int64_t T = static_cast<int64_t>(blockIdx.x) * 4 + wave;
    const uint4* __restrict__ a = A3 + T * (NBLK * 64) + lane;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // A ring: the blocks of groups g .. g + DEPTH - 1 in flight (raw 16-byte loads, 1 KB per wave-instruction)
    constexpr int DEPTH = 4;
    uint4 ar[DEPTH][3];
    auto a_load = [&](int g, uint4 (&dst)[3]) {
        if (g < G3) {
#pragma unroll
            for (int p = 0; p < 3; ++p) dst[p] = a[(3 * g + p) * 64];
        } else {
            dst[0] = a[(3 * G3 + min(g, NG - 1) - G3) * 64];
        }
    };
    uint4 st[ST_PER_THREAD];
    auto stage_load = [&](int c) {
        const uint4* src = B3 + static_cast<size_t>(min(c, NG / 2 - 1)) * CHUNK_PAD_U4;
#pragma unroll
        for (int u = 0; u < ST_PER_THREAD; ++u) st[u] = src[min(static_cast<int>(threadIdx.x) + 256 * u, CHUNK_U4 - 1)];
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int u = 0; u < ST_PER_THREAD; ++u) {
            const int q = threadIdx.x + 256 * u;
            if (q < CHUNK_U4) lds[buf * CHUNK_U4 + q] = st[u];
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int p = 0; p < 3; ++p) ar[d][p] = a[(3 * d + p) * 64];
    stage_load(0);
    stage_store(0);
    __syncthreads();

    // one chunk = two groups; everything about it is a compile-time constant (ring slots, LDS buffer, 6- or 3-product body,
    // which A blocks to prefetch): no branches inside, so hipcc keeps counted waits across the whole unrolled K loop
    auto chunk = [&](auto c_c) {
        constexpr int C = decltype(c_c)::value, BUF = C & 1;
        if (C + 1 < NG / 2) stage_load(C + 1);
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
            constexpr int dummy = 0; (void)dummy;
            const int g = 2 * C + gq;
            const bool exact = g >= G3;                              // compile-time after unrolling (C constexpr, gq unrolled)
            const int slot = 2 * BUF + gq;
            const uint4* bp = lds + BUF * CHUNK_U4 + gq * (NT * 3 * 64) + lane;
            const bf16x8 ah = __builtin_bit_cast(bf16x8, ar[slot][0]);
            const bf16x8 am = __builtin_bit_cast(bf16x8, ar[slot][1]);
            const bf16x8 al = __builtin_bit_cast(bf16x8, ar[slot][2]);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bf16x8 b0 = __builtin_bit_cast(bf16x8, bp[(t * 3 + 0) * 64]);
                const bf16x8 b1 = __builtin_bit_cast(bf16x8, bp[(t * 3 + 1) * 64]);
                const bf16x8 b2 = __builtin_bit_cast(bf16x8, bp[(t * 3 + 2) * 64]);
                f32x16 c_ = acc[t];
                if (!exact) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b0, c_, 0, 0, 0);
                c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b2, c_, 0, 0, 0);
                if (!exact) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b1, c_, 0, 0, 0);
                if (!exact) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b0, c_, 0, 0, 0);
                c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1, c_, 0, 0, 0);
                c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0, c_, 0, 0, 0);
                acc[t] = c_;
            }
            const int gn = g + DEPTH;                                 // next group for this slot
            if (gn < NG) {
                if (gn < G3) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) ar[slot][p] = a[(3 * gn + p) * 64];
                } else {
                    ar[slot][0] = a[(3 * G3 + gn - G3) * 64];
                }
            }
        }
        if (C + 1 < NG / 2) stage_store(BUF ^ 1);
        __syncthreads();
    };
    chunk(std::integral_constant<int, 0>{}); chunk(std::integral_constant<int, 1>{}); chunk(std::integral_constant<int, 2>{});
    chunk(std::integral_constant<int, 3>{}); chunk(std::integral_constant<int, 4>{}); chunk(std::integral_constant<int, 5>{});
    chunk(std::integral_constant<int, 6>{}); chunk(std::integral_constant<int, 7>{}); chunk(std::integral_constant<int, 8>{});

    // ---- epilogue: C tile + per-tile BatchNorm partials (mean, M2), as gemm.hip's rows_epilogue
    const int64_t m0 = T * 32;
    if (EPI == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = 32 * t + i;
            float sum = 0.f;
            if (32 * t + 32 <= NCOL) {
                float* base = Z + m0 * ldz + static_cast<uint32_t>(4 * s) * ldz + n;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    base[static_cast<int64_t>((r & 3) + 8 * (r >> 2)) * ldz] = acc[t][r];
                    sum += acc[t][r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * s;
                    if (n < NCOL) { Z[(m0 + row) * ldz + n] = acc[t][r]; sum += acc[t][r]; }
                }
            }
            sum += __shfl_xor(sum, 32);
            const float mean = sum * (1.f / 32.f);
            float m2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = acc[t][r] - mean; m2 = fmaf(d, d, m2); }
            m2 += __shfl_xor(m2, 32);
            if (s == 0 && n < NCOL) { stats[(T * NCOL + n) * 2] = mean; stats[(T * NCOL + n) * 2 + 1] = m2; }
        }
    } else {
        // transposed through LDS (the B buffers are free after the last barrier): each wave owns 32 rows x 160 floats
        // (pitch 164 -> conflict-light), then writes its 32 x 148 block -- contiguous in Z when ldz == 148 -- with 16-byte stores
        float* tile = reinterpret_cast<float*>(lds) + wave * (32 * 164);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = 32 * t + i;
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * s;
                tile[row * 164 + n] = acc[t][r];
                sum += acc[t][r];
            }
            sum += __shfl_xor(sum, 32);
            const float mean = sum * (1.f / 32.f);
            float m2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = acc[t][r] - mean; m2 = fmaf(d, d, m2); }
            m2 += __shfl_xor(m2, 32);
            if (s == 0 && n < NCOL) { stats[(T * NCOL + n) * 2] = mean; stats[(T * NCOL + n) * 2 + 1] = m2; }
        }
        // (wave-private tile: no barrier needed, only the wave's own LDS writes -> reads ordering)
        __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0)
        constexpr int F4_PER_ROW = NCOL / 4;      // 37
        float4* zo = reinterpret_cast<float4*>(Z + m0 * ldz);
        for (int q = lane; q < 32 * F4_PER_ROW; q += 64) {
            const int row = q / F4_PER_ROW, c4 = q - row * F4_PER_ROW;
            const float4 v = *reinterpret_cast<const float4*>(tile + row * 164 + 4 * c4);
            zo[q] = v;                            // ldz == NCOL: the block is contiguous
        }
    }
}


// ---- v3: every global access of the K loop is either an LDS-DMA (B chunks, __builtin_amdgcn_global_load_lds) or an inline-asm
// load (A fragments) with hand-counted s_waitcnt vmcnt(N) and raw s_barrier: hipcc's own scheduling of the v1 kernel
// re-used ONE register for the eight staging loads of a chunk and waited vmcnt(0) after each of them (81 us).
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* glb_ptr_t;

#define ALOAD(dst, base, OFF) asm volatile("global_load_dwordx4 %0, %1, off offset:" #OFF : "=v"(dst) : "v"(base) : "memory")
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// ABL (ablation of the main loop, timing only): 1 = no A loads in the loop, 2 = no B DMA in the loop, 4 = no barriers,
// 8 = no LDS reads (B fragments taken from registers)
// PIPE: the B fragments are read with inline-asm ds_read_b128 one (group, tile) item ahead of the MFMAs that use them, with hand-counted
// s_waitcnt lgkmcnt(3) (hipcc waits lgkmcnt(0) right after every read it emits: the LDS latency of every item is exposed)
#define LDSREAD(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory")
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int OFF> __device__ __forceinline__ void lds_read3(u32x4 (&b)[3], uint32_t addr) {
    LDSREAD(b[0], addr, OFF); LDSREAD(b[1], addr, OFF + 1024); LDSREAD(b[2], addr, OFF + 2048);
}
template <int N> __device__ __forceinline__ void wait_lgkm(u32x4 (&b)[3]) {      // the fragments become usable only after the wait
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]) : "n"(N) : "memory");
}
template <bool IL, int EPI, int ABL = 0, bool PIPE = false>
__global__ __launch_bounds__(256, 2) void rows3_fwd_v3(const uint4* __restrict__ A3, const uint4* __restrict__ B3,
                                                       float* __restrict__ Z, int ldz, float* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];      // [2][CHUNK_PAD_U4]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const int i = lane & 31, s = lane >> 5;
    const int64_t T = static_cast<int64_t>(blockIdx.x) * 4 + wave;
    const uint4* __restrict__ a = A3 + T * (NBLK * 64) + lane;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const uint32_t lds_lane = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_ptr_t)lds)) + lane * 16;
    uint4 ar[4][3];
    // blocks of group g: 3 g + p (g < G3) or 3 G3 + g - G3; each 1 KB = 64 uint4: the immediate offset reaches +-4 KB
    auto a_group = [&](auto g_c, uint4 (&dst)[3]) {
        constexpr int G = decltype(g_c)::value;
        if constexpr (G < G3) {
            const uint4* base = a + (3 * G) * 64;
            ALOAD(dst[0], base, 0); ALOAD(dst[1], base, 1024); ALOAD(dst[2], base, 2048);
        } else if constexpr (G < NG) {
            const uint4* base = a + (3 * G3 + G - G3) * 64;
            ALOAD(dst[0], base, 0);
        }
    };
    auto n_loads = [](int g) constexpr { return g < G3 ? 3 : (g < NG ? 1 : 0); };
    // B chunk c -> LDS buffer: 32 DMA pieces of 1 KB, 8 per wave
    auto dma_chunk = [&](int c, int buf) {
        const uint4* src = B3 + static_cast<size_t>(c) * CHUNK_PAD_U4 + wave * (8 * 64) + lane;
        uint4* dst = lds + buf * CHUNK_PAD_U4 + wave * (8 * 64);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + u * 64), (lds_ptr_t)(dst + u * 64), 16, 0, 0);
    };
    a_group(std::integral_constant<int, 0>{}, ar[0]);
    a_group(std::integral_constant<int, 1>{}, ar[1]);
    a_group(std::integral_constant<int, 2>{}, ar[2]);
    a_group(std::integral_constant<int, 3>{}, ar[3]);
    dma_chunk(0, 0);
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    auto chunk = [&](auto c_c) {
        constexpr int C = decltype(c_c)::value, BUF = C & 1;
        if constexpr (C + 1 < NG / 2 && !(ABL & 2)) dma_chunk(C + 1, BUF ^ 1);
        if constexpr (PIPE) {
            u32x4 bq[2][3];
            constexpr int BASE = BUF * CHUNK_PAD_U4 * 16;
            auto rd = [&](auto j_c) {
                constexpr int J = decltype(j_c)::value, OFF = BASE + ((J / NT) * NT + (J % NT)) * 3 * 1024;
                lds_read3<OFF>(bq[J & 1], lds_lane);
            };
            auto item = [&](auto j_c) {
                constexpr int J = decltype(j_c)::value, GQ = J / NT, TT = J % NT, G = 2 * C + GQ, SLOT = 2 * BUF + GQ;
                constexpr bool exact = G >= G3;
                if constexpr (J + 1 < 2 * NT) {
                    rd(std::integral_constant<int, J + 1>{});
                    wait_lgkm<3>(bq[J & 1]);
                } else {
                    wait_lgkm<0>(bq[J & 1]);
                }
                const bf16x8 ah = __builtin_bit_cast(bf16x8, ar[SLOT][0]);
                const bf16x8 am = __builtin_bit_cast(bf16x8, ar[SLOT][1]);
                const bf16x8 al = __builtin_bit_cast(bf16x8, ar[SLOT][2]);
                const bf16x8 b0 = __builtin_bit_cast(bf16x8, bq[J & 1][0]);
                const bf16x8 b1 = __builtin_bit_cast(bf16x8, bq[J & 1][1]);
                const bf16x8 b2 = __builtin_bit_cast(bf16x8, bq[J & 1][2]);
                f32x16 c_ = acc[TT];
                if (!exact) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b0, c_, 0, 0, 0);
                c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b2, c_, 0, 0, 0);
                if (!exact) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b1, c_, 0, 0, 0);
                if (!exact) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b0, c_, 0, 0, 0);
                c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1, c_, 0, 0, 0);
                c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0, c_, 0, 0, 0);
                acc[TT] = c_;
                if constexpr (TT == NT - 1) {
                    __builtin_amdgcn_sched_barrier(0);   // the slot's MFMAs are issued before its registers are re-loaded
                    if constexpr (!(ABL & 1)) a_group(std::integral_constant<int, 2 * C + 4 + GQ>{}, ar[SLOT]);
                }
            };
            rd(std::integral_constant<int, 0>{});
            item(std::integral_constant<int, 0>{}); item(std::integral_constant<int, 1>{}); item(std::integral_constant<int, 2>{});
            item(std::integral_constant<int, 3>{}); item(std::integral_constant<int, 4>{}); item(std::integral_constant<int, 5>{});
            item(std::integral_constant<int, 6>{}); item(std::integral_constant<int, 7>{}); item(std::integral_constant<int, 8>{});
            item(std::integral_constant<int, 9>{});
        } else
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
            const int g = 2 * C + gq;
            const bool exact = g >= G3;
            const int slot = 2 * BUF + gq;
            const uint4* bp = lds + BUF * CHUNK_PAD_U4 + gq * (NT * 3 * 64) + lane;
            const bf16x8 ah = __builtin_bit_cast(bf16x8, ar[slot][0]);
            const bf16x8 am = __builtin_bit_cast(bf16x8, ar[slot][1]);
            const bf16x8 al = __builtin_bit_cast(bf16x8, ar[slot][2]);
            if (IL) {
                // tile-interleaved order: consecutive MFMAs go to DIFFERENT accumulators (a chain of six dependent MFMAs per
                // tile left the wave stalled at issue 59 % of its cycles); every accumulator still sees its six products in
                // the same order, so the results are bit-identical
                bf16x8 b[NT][3];
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int p = 0; p < 3; ++p) b[t][p] = __builtin_bit_cast(bf16x8, bp[(t * 3 + p) * 64]);
                if (!exact) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b[t][0], acc[t], 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[t][2], acc[t], 0, 0, 0);
                if (!exact) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b[t][1], acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b[t][0], acc[t], 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[t][1], acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[t][0], acc[t], 0, 0, 0);
            } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                bf16x8 b0 = (ABL & 8) ? am : __builtin_bit_cast(bf16x8, bp[(t * 3 + 0) * 64]);
                bf16x8 b1 = (ABL & 8) ? al : __builtin_bit_cast(bf16x8, bp[(t * 3 + 1) * 64]);
                bf16x8 b2 = (ABL & 8) ? ah : __builtin_bit_cast(bf16x8, bp[(t * 3 + 2) * 64]);
                if (ABL & 8) asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2));      // opaque: the five tiles are not merged
                f32x16 c_ = acc[t];
                if (!exact) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b0, c_, 0, 0, 0);
                c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b2, c_, 0, 0, 0);
                if (!exact) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b1, c_, 0, 0, 0);
                if (!exact) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b0, c_, 0, 0, 0);
                c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1, c_, 0, 0, 0);
                c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0, c_, 0, 0, 0);
                acc[t] = c_;
            }
            }
            __builtin_amdgcn_sched_barrier(0);       // the slot's MFMAs are issued before its registers are re-loaded
            if constexpr (!(ABL & 1)) {
                if (gq == 0) a_group(std::integral_constant<int, 2 * C + 4>{}, ar[2 * BUF]);
                else a_group(std::integral_constant<int, 2 * C + 5>{}, ar[2 * BUF + 1]);
            }
        }
        // everything but the A loads issued in THIS chunk has landed: the DMA of chunk C + 1 and the A blocks of chunk C + 1
        wait_vm<(ABL & 1) ? 0 : n_loads(2 * C + 4) + n_loads(2 * C + 5)>();
        if constexpr (!(ABL & 4)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    chunk(std::integral_constant<int, 0>{}); chunk(std::integral_constant<int, 1>{}); chunk(std::integral_constant<int, 2>{});
    chunk(std::integral_constant<int, 3>{}); chunk(std::integral_constant<int, 4>{}); chunk(std::integral_constant<int, 5>{});
    chunk(std::integral_constant<int, 6>{}); chunk(std::integral_constant<int, 7>{}); chunk(std::integral_constant<int, 8>{});

    const int64_t m0 = T * 32;
    if (EPI == 2) {                       // no C stores: the main loop alone (one value per lane keeps the accumulators live)
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[t][r];
        if (sum == 123.456f) Z[m0 * ldz + lane] = sum;
        return;
    }
    if (EPI == 1) {
        // rows through LDS (the B buffers are free after the last barrier; 4 waves x 32 rows x pitch 164 floats = 84 KB does
        // not fit beside another workgroup, so each wave transposes its tile in two halves of 16 rows: 4 x 10.5 KB)
        float* tile = reinterpret_cast<float*>(lds) + wave * (16 * 164);
        constexpr int F4_PER_ROW = NCOL / 4;      // 37
        float4* zo = reinterpret_cast<float4*>(Z + m0 * ldz);
#pragma unroll
        for (int t = 0; t < NT; ++t) {            // statistics first (registers only)
            const int n = 32 * t + i;
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[t][r];
            sum += __shfl_xor(sum, 32);
            const float mean = sum * (1.f / 32.f);
            float m2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = acc[t][r] - mean; m2 = fmaf(d, d, m2); }
            m2 += __shfl_xor(m2, 32);
            if (s == 0 && n < NCOL) { stats[(T * NCOL + n) * 2] = mean; stats[(T * NCOL + n) * 2 + 1] = m2; }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {    // rows 16 half .. 16 half + 15: registers r = 8 half .. 8 half + 7
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = 8 * half + rr;
                    const int row = (r & 3) + 8 * ((r >> 2) & 1) + 4 * s;      // row within the half
                    tile[row * 164 + 32 * t + i] = acc[t][r];
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (int q = lane; q < 16 * F4_PER_ROW; q += 64) {
                const int row = q / F4_PER_ROW, c4 = q - row * F4_PER_ROW;
                zo[half * 16 * F4_PER_ROW + q] = *reinterpret_cast<const float4*>(tile + row * 164 + 4 * c4);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = 32 * t + i;
        float sum = 0.f;
        if (32 * t + 32 <= NCOL) {
            float* base = Z + m0 * ldz + static_cast<uint32_t>(4 * s) * ldz + n;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                base[static_cast<int64_t>((r & 3) + 8 * (r >> 2)) * ldz] = acc[t][r];
                sum += acc[t][r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * s;
                if (n < NCOL) { Z[(m0 + row) * ldz + n] = acc[t][r]; sum += acc[t][r]; }
            }
        }
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.f / 32.f);
        float m2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = acc[t][r] - mean; m2 = fmaf(d, d, m2); }
        m2 += __shfl_xor(m2, 32);
        if (s == 0 && n < NCOL) { stats[(T * NCOL + n) * 2] = mean; stats[(T * NCOL + n) * 2 + 1] = m2; }
    }
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main() {
    const size_t nA = static_cast<size_t>(M_ROWS) * KF, nW = static_cast<size_t>(NCOL) * KF;
    std::vector<float> hA(nA), hW(nW);
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return static_cast<float>((st >> 40) & 0xFFFFFF) / 8388608.f - 1.f; };
    for (size_t r = 0; r < M_ROWS; ++r) {
        for (int c = 0; c < KR; ++c) hA[r * KF + c] = rnd() * 0.3f;
        for (int c = KR; c < KF; ++c) hA[r * KF + c] = 0.f;
        for (int t = 0; t < 24; ++t) hA[r * KF + KR + 5 * t + (static_cast<int>(fabsf(rnd()) * 5) % 5)] = 1.f;     // 24 small tables
    }
    for (auto& w : hW) w = rnd() * 0.1f;
    float *dA, *dW, *dZ, *dS;
    __bf16 *dA3, *dB3;
    CHECK(hipMalloc(&dA, nA * 4)); CHECK(hipMalloc(&dW, nW * 4));
    CHECK(hipMalloc(&dZ, static_cast<size_t>(M_ROWS) * NCOL * 4 * 4));          // four rotated outputs
    CHECK(hipMalloc(&dS, static_cast<size_t>(M_ROWS / 32) * NCOL * 2 * 4));
    const size_t a3_elems = static_cast<size_t>(M_ROWS / 32) * NBLK * 512;
    CHECK(hipMalloc(&dA3, a3_elems * 2 * 4));                                    // four rotated copies (> Infinity Cache)
    CHECK(hipMalloc(&dB3, static_cast<size_t>(NG / 2) * CHUNK_PAD_U4 * 16));
    CHECK(hipMemset(dB3, 0, static_cast<size_t>(NG / 2) * CHUNK_PAD_U4 * 16));
    CHECK(hipMemcpy(dA, hA.data(), nA * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dW, hW.data(), nW * 4, hipMemcpyHostToDevice));
    for (int j = 0; j < 4; ++j)
        hipLaunchKernelGGL(pack_a3, dim3(static_cast<unsigned>((static_cast<int64_t>(M_ROWS) * NG * 2 + 255) / 256)), dim3(256), 0, 0, dA, dA3 + j * a3_elems);
    hipLaunchKernelGGL(pack_b3, dim3((NG * NT * 64 + 255) / 256), dim3(256), 0, 0, dW, dB3);
    CHECK(hipDeviceSynchronize());

    typedef void (*kfn)(const uint4*, const uint4*, float*, int, float*);
    struct { const char* name; kfn fn; bool check; } vs[] = {
        {"v3 chain  / direct stores", rows3_fwd_v3<false, 0>, true}, {"v3 chain  / LDS rows + 16-B stores", rows3_fwd_v3<false, 1>, true},
        {"v3 chain  / no stores", rows3_fwd_v3<false, 2>, false}, {"v3 interleaved / LDS rows", rows3_fwd_v3<true, 1>, true},
        {"v3 pipelined B reads / LDS rows", rows3_fwd_v3<false, 1, 0, true>, true},
        {"v3 pipelined B reads / no stores", rows3_fwd_v3<false, 2, 0, true>, false},
        {"  pipelined, no stores, no loads, no barriers", rows3_fwd_v3<false, 2, 7, true>, false},
        {"  no stores, no A loads", rows3_fwd_v3<false, 2, 1>, false}, {"  no stores, no B DMA", rows3_fwd_v3<false, 2, 2>, false},
        {"  no stores, no A loads, no B DMA", rows3_fwd_v3<false, 2, 3>, false},
        {"  no stores, no loads, no barriers", rows3_fwd_v3<false, 2, 7>, false},
        {"  no stores, MFMA only", rows3_fwd_v3<false, 2, 15>, false},
        {"  no stores, loads but no barriers", rows3_fwd_v3<false, 2, 4>, false}};
    for (auto& v : vs) {
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(v.fn), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        auto run = [&](int j) {
            const uint4* a3 = reinterpret_cast<const uint4*>(dA3 + (j & 3) * a3_elems);
            float* z = dZ + static_cast<size_t>(j & 3) * M_ROWS * NCOL;
            hipLaunchKernelGGL(v.fn, dim3(M_ROWS / 128), dim3(256), 2 * CHUNK_PAD_U4 * 16, 0, a3, reinterpret_cast<const uint4*>(dB3), z, NCOL, dS);
        };
        CHECK(hipMemset(dZ, 0, static_cast<size_t>(M_ROWS) * NCOL * 4));
        for (int j = 0; j < 4; ++j) run(j);
        CHECK(hipDeviceSynchronize());
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        const int n = 40;
        for (int j = 0; j < n; ++j) run(j);
        hipEventRecord(b);
        CHECK(hipEventSynchronize(b));
        float ms;
        hipEventElapsedTime(&ms, a, b);
        // check 256 sampled rows against fp64
        std::vector<float> hZ(static_cast<size_t>(M_ROWS) * NCOL);
        CHECK(hipMemcpy(hZ.data(), dZ, hZ.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int q = 0; q < 256; ++q) {
            const size_t r = (static_cast<size_t>(q) * 2654435761u) % M_ROWS;
            for (int n_ = 0; n_ < NCOL; ++n_) {
                double ref = 0, mag = 0;
                for (int c = 0; c < KF; ++c) { ref += static_cast<double>(hA[r * KF + c]) * hW[n_ * KF + c]; mag += fabs(static_cast<double>(hA[r * KF + c]) * hW[n_ * KF + c]); }
                worst = fmax(worst, fabs(hZ[r * NCOL + n_] - ref) / (mag + 1e-30));
            }
        }
        printf("%-40s %.1f us per launch; max |err| / sum|a b| = %.2e%s\n", v.name, ms / n * 1e3, worst, v.check ? "" : " (not stored)");
    }
    return 0;
}
