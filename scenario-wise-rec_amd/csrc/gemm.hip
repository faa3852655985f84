// K2: fp32 matrix products on the CDNA4 f32 MFMA pipe (v_mfma_f32_32x32x2_f32).
//
// Why f32 MFMA: the parity bar is 1e-4 on fp32 logits with reductions up to K ~ 580
// (SURVEY.md fact 5); the f32-input MFMA is an exact fp32 fma chain in k order, so the products
// match the reference's fp32 addmm to rounding.  It issues one 32x32x2 tile per 64 cycles per SIMD
// (157 TFLOP/s chip peak), i.e. ONE operand register pair feeds 64 cycles of matrix work.  That
// leaves so much load slack that no LDS staging is needed: every wave owns 32 output rows and a
// strip of 32*NT output columns and reads its MFMA operands straight from global memory / L2 in
// fragment layout, 16 bytes per lane where the reduction index is contiguous.  No barriers, no LDS.
//
// Fragment trick: the MFMA sums over two k-slots (lanes 0-31 hold slot 0, lanes 32-63 slot 1).  Any
// assignment of real k indices to (slot, step) is valid as long as A and B agree, so lane (i, s) loads
// the four consecutive k's  kb + 4s .. kb + 4s + 3  with ONE 16-byte load and spends them on four
// consecutive MFMAs: 8 k's per group, one dwordx4 per operand row per group.
//
//   nt:  C[m, n] = sum_k A[m, k] B[n, k]   A rows and B rows k-contiguous: both operands 16-B loads
//   nn:  C[m, n] = sum_k A[m, k] B[k, n]   A as above, B one dword per (k, lane), lanes along n
//   tn:  C[p, q] = sum_m A[m, p] B[m, q]   both operands one dword per (m, lane), lanes along p / q;
//                                          the batch dimension m is split over waves, partial tiles are
//                                          written to a workspace and summed in fixed order
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"

#include "rows_epilogue.h"

#define GEMM_WAVES 4
#define GEMM_THREADS (GEMM_WAVES * 64)

struct GemmK {
    swr_gemm_args a;
    int n_tiles_m;   // ceil(M / 32)
};

template <bool VEC>
__device__ __forceinline__ float4 load_k4(const float* __restrict__ row, int k, int K) {
    // four consecutive reduction elements k..k+3 of one operand row, zero beyond K
    if (VEC) {
        if (k + 3 < K) return *reinterpret_cast<const float4*>(row + k);
    }
    float4 v;
    v.x = k < K ? row[k] : 0.f;
    v.y = k + 1 < K ? row[k + 1] : 0.f;
    v.z = k + 2 < K ? row[k + 2] : 0.f;
    v.w = k + 3 < K ? row[k + 3] : 0.f;
    return v;
}

__device__ __forceinline__ float comp(const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }

// ---- LDS variant: the B operand (weights) of a k-chunk is staged once per workgroup in LDS and shared by the four
// waves (4x less L2 traffic, no B registers); the A operand (activations, streamed from HBM, private to a wave) stays on
// the direct global -> register path with a deep prefetch ring.  One MFMA consumes one ds_read_b32 per lane.
#define LDS_KC 32                       // k per chunk (4 MFMA k-groups)
template <int NT, bool BT, bool PRO>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_rows_lds_kernel(const GemmK kk) {
    constexpr int LDN = NT * 32 + 4;                    // row pitch of the k-major B tile
    constexpr int F4 = NT * 32 * LDS_KC / 4;            // float4 per chunk
    constexpr int F4_PER_THREAD = (F4 + GEMM_THREADS - 1) / GEMM_THREADS;
    extern __shared__ __attribute__((aligned(16))) float Bs[];       // [2][LDS_KC * LDN]
    const swr_gemm_args& a = kk.a;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, s = lane >> 5;
    const int g = blockIdx.z;
    const int64_t tile_m = static_cast<int64_t>(blockIdx.x) * GEMM_WAVES + wave;
    const int64_t m0 = tile_m * 32;
    const int n0 = blockIdx.y * (32 * NT);
    const int K = a.K, N = a.N;
    const float* __restrict__ Ag = a.A + g * a.gsA;
    const float* __restrict__ Bg = a.B + g * a.gsB;
    const int64_t ra = max<int64_t>(0, min(m0 + i, a.M - 1));
    const float* __restrict__ arow = Ag + ra * a.lda;
    const float* __restrict__ psc = PRO ? a.a_scale + g * a.gsScale : nullptr;
    const float* __restrict__ psh = PRO ? a.a_shift + g * a.gsScale : nullptr;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // Everything below is branch-free (clamped addresses + selects) so that hipcc can keep counted
    // s_waitcnt vmcnt / lgkmcnt across the unrolled body.  Requires K % 4 == 0 (and N % 4 == 0 for [K, N] weights).
    auto ld4c = [&](const float* __restrict__ row, int k) -> float4 {      // k % 4 == 0; zero when k >= K
        const float4 v = *reinterpret_cast<const float4*>(row + min(k, K - 4));
        const bool ok = k < K;
        return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    };
    // ---- B staging: global -> registers (issued a chunk ahead) -> LDS
    float4 stage[F4_PER_THREAD];
    auto stage_load = [&](int kc) {
#pragma unroll
        for (int u = 0; u < F4_PER_THREAD; ++u) {
            const int q = min(static_cast<int>(threadIdx.x) + u * GEMM_THREADS, F4 - 1);
            if (BT) {                                   // W[n][k]: 8 float4 along k per output column
                const int n = min(n0 + (q >> 3), N - 1);
                stage[u] = ld4c(Bg + static_cast<int64_t>(n) * a.ldb, kc + 4 * (q & 7));
            } else {                                    // W[k][n]: NT*8 float4 along n per k row
                const int kr = q / (NT * 8), n = n0 + 4 * (q - kr * (NT * 8)), k = kc + kr;
                const float4 v = *reinterpret_cast<const float4*>(Bg + static_cast<int64_t>(min(k, K - 1)) * a.ldb + min(n, N - 4));
                const bool ok = k < K && n < N;
                stage[u] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
            }
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int u = 0; u < F4_PER_THREAD; ++u) {
            const int q = threadIdx.x + u * GEMM_THREADS;
            if (q < F4) {
                if (BT) {
                    const int n = q >> 3, kq = 4 * (q & 7);
                    float* d = Bs + buf * (LDS_KC * LDN) + kq * LDN + n;
                    d[0] = stage[u].x; d[LDN] = stage[u].y; d[2 * LDN] = stage[u].z; d[3 * LDN] = stage[u].w;
                } else {
                    const int kr = q / (NT * 8), nq = 4 * (q - kr * (NT * 8));
                    *reinterpret_cast<float4*>(Bs + buf * (LDS_KC * LDN) + kr * LDN + nq) = stage[u];
                }
            }
        }
    };

    // ---- A ring: one float4 per k-group, DEPTH groups (two chunks) in flight
    constexpr int DEPTH = 8;
    float4 ar[DEPTH];
    auto a_load = [&](int gi) -> float4 {
        const int k = 8 * gi + 4 * s;
        float4 v = ld4c(arow, k);
        if (PRO) {
            const float4 sc = ld4c(psc, k), sh = ld4c(psh, k);         // zero beyond K: the product stays zero
            v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
            if (a.a_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        }
        return v;
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) ar[d] = a_load(d);

    stage_load(0);
    stage_store(0);
    __syncthreads();
    const int n_chunks = (K + LDS_KC - 1) / LDS_KC;
    // B fragments of a k-group: 4 steps x NT tiles, read from LDS one group ahead of the MFMAs that use them
    auto b_read = [&](float (&bf)[4 * NT], int half, int gq) {
        const float* bp = Bs + half * (LDS_KC * LDN) + (8 * gq + 4 * s) * LDN + i;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int t = 0; t < NT; ++t) bf[cc * NT + t] = bp[cc * LDN + 32 * t];
    };
    // ring slots are compile-time constants: DEPTH = 2 chunks, chunk parity selects the half of the ring
    auto do_chunk = [&](int c, auto half) {
        constexpr int HALF = decltype(half)::value;
        stage_load((c + 1) * LDS_KC);                  // beyond K this loads zeros (never stored)
        float bf0[4 * NT], bf1[4 * NT];
        b_read(bf0, HALF, 0);
#pragma unroll
        for (int gq = 0; gq < LDS_KC / 8; ++gq) {
            float (&cur)[4 * NT] = (gq & 1) ? bf1 : bf0;
            float (&nxt)[4 * NT] = (gq & 1) ? bf0 : bf1;
            if (gq + 1 < LDS_KC / 8) b_read(nxt, HALF, gq + 1);
            // pin the order: the LDS reads of the NEXT group are issued before this group's MFMAs, which then wait only
            // for the older reads (counted lgkmcnt) -- otherwise the scheduler sinks every read next to its use
            __builtin_amdgcn_sched_barrier(0);
            const float4 av = ar[gq + HALF * (DEPTH / 2)];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(av, cc), cur[cc * NT + t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            ar[gq + HALF * (DEPTH / 2)] = a_load(c * (LDS_KC / 8) + gq + DEPTH);
        }
        __syncthreads();                       // all waves are done with the other buffer (read one chunk ago)
        if (c + 1 < n_chunks) stage_store(HALF ^ 1);
        __syncthreads();
    };
    for (int c = 0; c < n_chunks; c += 2) {
        do_chunk(c, std::integral_constant<int, 0>{});
        if (c + 1 < n_chunks) do_chunk(c + 1, std::integral_constant<int, 1>{});
    }
    rows_epilogue<NT>(a, kk.n_tiles_m, acc, g, tile_m, m0, n0, i, s);
}

// ---- bf16 x 3 split variant ("x6"): every fp32 operand is split exactly into three bf16 pieces x = h + m + l
// (8 + 8 + 8 significant bits) and the product is rebuilt from the six partial products that matter,
//     a b ~= ah bh + ah bm + am bh + ah bl + al bh + am bm            (dropped terms are <= 2^-24 |a b|),
// each one a bf16 MFMA (v_mfma_f32_32x32x16_bf16, fp32 accumulate, exact bf16 x bf16 products).  Six bf16 MFMAs cover
// 16 k in 6 x 32 cycles where the f32 MFMA needs 8 x 64: 2.7x the matrix rate at fp32-class accuracy, which is what the
// 1e-4 logit bar needs (a plain bf16 product misses it by two orders of magnitude).
// Structure as the LDS kernel above: B (weights, [N, K] only) split once per workgroup while it is staged into LDS as
// three bf16 planes [plane][col][k]; A split in registers after its 16-byte loads.
#include "split3.h"
#define X6_KC 32                        // k per chunk (2 MFMA groups of 16)
#define X6_PITCH 40                     // bf16 per LDS row: 32 + 8 pad -> 80-byte pitch, conflict-free ds_read_b128
#define X6_DOUBLE_BUFFER(NT) 0         // measured: two buffers / one barrier per chunk is no faster than one / two
// BF (SWR_GEMM=bf16, the perf mode of SURVEY.md fact 5 -- never the parity path): operands rounded to bf16, ONE product
// per k-group instead of six; what the matrix pipes can do for this layer once the 1e-4 logit bar is given up.
template <int NT, bool PRO, bool PS, bool BF>
__global__ __launch_bounds__(GEMM_THREADS, (NT > 6 ? 1 : 2)) void gemm_rows_x6_kernel(const GemmK kk) {   // two waves per SIMD (eight accumulator
                                                                                      // tiles: one -- 140-208 bytes per lane of scratch at two)
    constexpr int NCOL = NT * 32;
    constexpr int PLANE = NCOL * X6_PITCH;              // bf16 elements per plane
    constexpr int F4 = NCOL * X6_KC / 4;                // float4 of W per chunk
    constexpr int F4_PER_THREAD = (F4 + GEMM_THREADS - 1) / GEMM_THREADS;
    // [buffers][3 planes][NCOL][X6_PITCH]: two buffers while two workgroups still fit a CU (NT <= 5: 77 KB), so the
    // refill of the next chunk overlaps this chunk's MFMAs and a chunk costs one barrier; one buffer (two barriers) above
    constexpr bool DB = X6_DOUBLE_BUFFER(NT);
    extern __shared__ __attribute__((aligned(16))) __bf16 Bx[];
    const swr_gemm_args& a = kk.a;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, s = lane >> 5;
    const int g = blockIdx.z;
    const int64_t tile_m = static_cast<int64_t>(blockIdx.x) * GEMM_WAVES + wave;
    const int64_t m0 = tile_m * 32;
    const int n0 = blockIdx.y * NCOL;
    const int K = a.K, N = a.N;
    const float* __restrict__ Ag = a.A + g * a.gsA;
    const float* __restrict__ Bg = a.B + g * a.gsB;
    const int64_t ra = max<int64_t>(0, min(m0 + i, a.M - 1));
    const float* __restrict__ arow = Ag + ra * a.lda;
    const float* __restrict__ psc = PRO ? a.a_scale + g * a.gsScale : nullptr;
    const float* __restrict__ psh = PRO ? a.a_shift + g * a.gsScale : nullptr;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // column tiles this workgroup really computes (wave-uniform): tiles past N -- or past n_compute, whose columns are
    // stored as the zeros their accumulators still hold -- are skipped instead of multiplied and masked
    const int n_need = (a.n_compute > 0 && a.n_compute < N) ? a.n_compute : N;
    const int nt_valid = max(0, min(NT, (n_need - n0 + 31) / 32));

    // Prefetched values stay RAW in their registers: any arithmetic on a loaded value at the load site (even the
    // zeroing of k >= K) makes hipcc wait for the load right there (91 x s_waitcnt vmcnt(0) in this kernel before) and
    // the rings prefetch nothing.  Addresses are clamped; the k >= K mask is applied when a value is consumed.
    auto ld4raw = [&](const float* __restrict__ row, int k) -> float4 {      // k % 4 == 0
        return *reinterpret_cast<const float4*>(row + min(k, K - 4));
    };
    auto mask4 = [&](float4 v, int k) -> float4 {
        const bool ok = k < K;
        return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    };
    // ---- B staging: fp32 global -> registers (a chunk ahead) -> split -> three bf16 planes in LDS; or, when the
    // caller pre-split B (swr_split_weights: once per step instead of once per workgroup), three plain 8-byte copies
    constexpr bool presplit = PS;
    const __bf16* __restrict__ Bs = static_cast<const __bf16*>(a.B_split);
    float4 stage[PS ? 1 : F4_PER_THREAD];
    bf16x4 stage_p[PS ? F4_PER_THREAD : 1][3];
    auto stage_load = [&](int kc) {
#pragma unroll
        for (int u = 0; u < F4_PER_THREAD; ++u) {
            const int q = min(static_cast<int>(threadIdx.x) + u * GEMM_THREADS, F4 - 1);
            const int n = min(n0 + (q >> 3), N - 1);
            if (presplit) {
                // planes are zero-padded to a multiple of X6_KC columns: no clamp; chunks past the end are never stored
                const __bf16* src = Bs + static_cast<int64_t>(n) * a.ld_split + min(kc, K - 1) / X6_KC * X6_KC + 4 * (q & 7);
#pragma unroll
                for (int p = 0; p < 3; ++p) stage_p[PS ? u : 0][p] = *reinterpret_cast<const bf16x4*>(src + p * a.plane_stride);
            } else {
                stage[PS ? 0 : u] = ld4raw(Bg + static_cast<int64_t>(n) * a.ldb, kc + 4 * (q & 7));
            }
        }
    };
    auto stage_store = [&](int kc, int buf) {
#pragma unroll
        for (int u = 0; u < F4_PER_THREAD; ++u) {
            const int q = threadIdx.x + u * GEMM_THREADS;
            if (q < F4) {
                const int n = q >> 3, kq = 4 * (q & 7);
                bf16x4 h, m, l;
                if (presplit) {
                    h = stage_p[PS ? u : 0][0]; m = stage_p[PS ? u : 0][1]; l = stage_p[PS ? u : 0][2];
                } else {
                    const float4 sv = mask4(stage[PS ? 0 : u], kc + kq);
                    SPLIT3_PAIR(sv.x, sv.y, h, m, l, 0);
                    SPLIT3_PAIR(sv.z, sv.w, h, m, l, 2);
                }
                __bf16* d = Bx + buf * (3 * PLANE) + n * X6_PITCH + kq;
                *reinterpret_cast<bf16x4*>(d) = h;
                *reinterpret_cast<bf16x4*>(d + PLANE) = m;
                *reinterpret_cast<bf16x4*>(d + 2 * PLANE) = l;
            }
        }
    };

    // ---- A ring: 8 fp32 (two float4) per 16-k group per lane, DEPTH groups (two chunks) in flight
    constexpr int DEPTH = 4;
    float4 ar[DEPTH][2];
    auto a_load = [&](int gi, float4 (&dst)[2]) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) dst[hh] = ld4raw(arow, 16 * gi + 8 * s + 4 * hh);
    };
    auto a_use = [&](int gi, const float4 (&raw)[2], float4 (&v2)[2]) {      // mask (+ the fused BN / ReLU prologue)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int k = 16 * gi + 8 * s + 4 * hh;
            float4 v = raw[hh];
            if (PRO) {
                const float4 sc = ld4raw(psc, k), sh = ld4raw(psh, k);
                v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
                if (a.a_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            }
            v2[hh] = mask4(v, k);
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) a_load(d, ar[d]);

    stage_load(0);
    stage_store(0, 0);
    __syncthreads();
    const int n_chunks = (K + X6_KC - 1) / X6_KC;
    auto b_read = [&](bf16x8 (&bf)[3], int gq, int t, int buf) {
        const __bf16* bp = Bx + buf * (3 * PLANE) + (32 * t + i) * X6_PITCH + 16 * gq + 8 * s;
#pragma unroll
        for (int p = 0; p < 3; ++p) bf[p] = *reinterpret_cast<const bf16x8*>(bp + p * PLANE);
    };
    // EX: the A columns of this chunk are exactly representable in bf16 (a_exact_from: the one-hot block of a folded first
    // layer): their middle / low terms are zero, so three of the six products vanish -- and so does the split
    auto do_chunk = [&](int c, auto half, auto exact) {
        constexpr int HALF = decltype(half)::value;
        constexpr bool EX = decltype(exact)::value;
        stage_load((c + 1) * X6_KC);                   // beyond K this loads zeros (never stored)
#pragma unroll
        for (int gq = 0; gq < X6_KC / 16; ++gq) {
            float4 av[2];
            a_use(c * (X6_KC / 16) + gq, ar[gq + HALF * (DEPTH / 2)], av);
            bf16x8 ah, am, al;
            if (EX || BF) {
                CVT_PAIR(av[0].x, av[0].y, ah, 0); CVT_PAIR(av[0].z, av[0].w, ah, 2);
                CVT_PAIR(av[1].x, av[1].y, ah, 4); CVT_PAIR(av[1].z, av[1].w, ah, 6);
            } else {
                SPLIT3_PAIR(av[0].x, av[0].y, ah, am, al, 0); SPLIT3_PAIR(av[0].z, av[0].w, ah, am, al, 2);
                SPLIT3_PAIR(av[1].x, av[1].y, ah, am, al, 4); SPLIT3_PAIR(av[1].z, av[1].w, ah, am, al, 6);
            }
            bf16x8 b0[3], b1[3];
            b_read(b0, gq, 0, DB ? HALF : 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t >= nt_valid) break;               // (uniform) nothing but empty tiles from here on
                bf16x8 (&cur)[3] = (t & 1) ? b1 : b0;
                bf16x8 (&nxt)[3] = (t & 1) ? b0 : b1;
                if (t + 1 < NT) b_read(nxt, gq, t + 1, DB ? HALF : 0);
                __builtin_amdgcn_sched_barrier(0);      // next tile's LDS reads are issued before this tile's MFMAs
                f32x16 c_ = acc[t];
                if ((!EX && !BF) && !SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, cur[0], c_, 0, 0, 0);     // small terms first
                if ((!BF) && !SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, cur[2], c_, 0, 0, 0);
                if ((!EX && !BF) && !SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, cur[1], c_, 0, 0, 0);
                if (!EX && !BF) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, cur[0], c_, 0, 0, 0);
                if (!BF) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, cur[1], c_, 0, 0, 0);
                c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, cur[0], c_, 0, 0, 0);
                acc[t] = c_;
            }
            a_load(c * (X6_KC / 16) + gq + DEPTH, ar[gq + HALF * (DEPTH / 2)]);
        }
        if (DB) {
            if (c + 1 < n_chunks) stage_store((c + 1) * X6_KC, HALF ^ 1);      // the other buffer: last read in chunk c - 1
            __syncthreads();
        } else {
            __syncthreads();                   // every wave is done reading this chunk's planes
            if (c + 1 < n_chunks) stage_store((c + 1) * X6_KC, 0);
            __syncthreads();
        }
    };
    // chunk PAIRS that lie entirely inside the exact column range run the three-product body (PRO rescales A: never
    // exact).  Two plain loops: a single loop with a per-chunk choice of body made hipcc spill 449 VGPRs, and so did a
    // mixed pair (full chunk + exact chunk) between the two loops (488)
    const int c_exact = (!PRO && a.a_exact_from > 0) ? min(n_chunks, (a.a_exact_from + 2 * X6_KC - 1) / (2 * X6_KC) * 2) : n_chunks;
    int c = 0;
    for (; c < c_exact; c += 2) {
        do_chunk(c, std::integral_constant<int, 0>{}, std::false_type{});
        if (c + 1 < n_chunks) do_chunk(c + 1, std::integral_constant<int, 1>{}, std::false_type{});
    }
    if (!PRO) {
        for (; c < n_chunks; c += 2) {
            do_chunk(c, std::integral_constant<int, 0>{}, std::true_type{});
            if (c + 1 < n_chunks) do_chunk(c + 1, std::integral_constant<int, 1>{}, std::true_type{});
        }
    }
    rows_epilogue<NT>(a, kk.n_tiles_m, acc, g, tile_m, m0, n0, i, s);
}

// NT output tiles (32 columns each) per wave; BT: B is [N, K] (nt) else [K, N] (nn)
template <int NT, bool BT, bool VEC, bool PRO>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_rows_kernel(const GemmK kk) {
    const swr_gemm_args& a = kk.a;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, s = lane >> 5;
    const int g = blockIdx.z;
    const int64_t tile_m = static_cast<int64_t>(blockIdx.x) * GEMM_WAVES + wave;
    const int64_t m0 = tile_m * 32;
    if (m0 >= a.M) return;
    const int n0 = blockIdx.y * (32 * NT);
    const int K = a.K, N = a.N;

    const float* __restrict__ Ag = a.A + g * a.gsA;
    const float* __restrict__ Bg = a.B + g * a.gsB;
    const int64_t ra = min(m0 + i, a.M - 1);
    const float* __restrict__ arow = Ag + ra * a.lda;
    const float* __restrict__ psc = PRO ? a.a_scale + g * a.gsScale : nullptr;
    const float* __restrict__ psh = PRO ? a.a_shift + g * a.gsScale : nullptr;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const float* brow[NT];
    bool bvalid[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n0 + 32 * t + i;
        bvalid[t] = n < N;
        brow[t] = BT ? Bg + static_cast<int64_t>(min(n, N - 1)) * a.ldb : Bg + min(n, N - 1);
    }

    // Software pipeline: the operands of k-group g+1 are in flight while the 4*NT MFMAs of group g issue
    // (one MFMA = 64 cycles on its SIMD; an L2 round trip is several hundred).  hipcc keeps the loads of the
    // next group outstanding behind counted s_waitcnt vmcnt(N).
    struct Frag {
        float4 a;
        float4 b[NT];
    };
    // main loop: no predicates at all.  Rows beyond M and columns beyond N are clamped to valid addresses and
    // produce values that are never stored; only the last partial k-group needs predicated loads.
    auto load4 = [&](const float* __restrict__ p) -> float4 {
        if (VEC) return *reinterpret_cast<const float4*>(p);
        return make_float4(p[0], p[1], p[2], p[3]);
    };
    auto prologue = [&](float4& v, int k, bool tail) {
        if (!PRO) return;
        // previous layer's BatchNorm (+ReLU) applied while the operand is loaded
        const float4 sc = tail ? load_k4<false>(psc, k, K) : load4(psc + k);
        const float4 sh = tail ? load_k4<false>(psh, k, K) : load4(psh + k);
        v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
        v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
        if (a.a_relu) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (tail) {
            if (k >= K) v.x = 0.f;
            if (k + 1 >= K) v.y = 0.f;
            if (k + 2 >= K) v.z = 0.f;
            if (k + 3 >= K) v.w = 0.f;
        }
    };
    auto load_frag = [&](int kb, Frag& f) {              // full group: kb + 8 <= K
        const int k = kb + 4 * s;
        f.a = load4(arow + k);
        prologue(f.a, k, false);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (BT) {
                f.b[t] = load4(brow[t] + k);
            } else {
                const float* p = brow[t] + static_cast<int64_t>(k) * a.ldb;
                f.b[t] = make_float4(p[0], p[a.ldb], p[2 * a.ldb], p[3 * a.ldb]);
            }
        }
    };
    auto load_frag_tail = [&](int kb, Frag& f) {         // last, partial group: zero beyond K
        const int k = kb + 4 * s;
        f.a = load_k4<false>(arow, k, K);
        prologue(f.a, k, true);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (BT) {
                f.b[t] = load_k4<false>(brow[t], k, K);
            } else {
                const float* p = brow[t] + static_cast<int64_t>(k) * a.ldb;
                f.b[t].x = k < K ? p[0] : 0.f;
                f.b[t].y = k + 1 < K ? p[a.ldb] : 0.f;
                f.b[t].z = k + 2 < K ? p[2 * a.ldb] : 0.f;
                f.b[t].w = k + 3 < K ? p[3 * a.ldb] : 0.f;
            }
        }
    };
    auto mma_frag = [&](const Frag& f) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(f.a, c), comp(f.b[t], c), acc[t], 0, 0, 0);
    };
    // ring of ROWS_DEPTH register sets: the operands of the next ROWS_DEPTH-1 k-groups are in flight while one group's
    // MFMAs issue (A streams from HBM: one group of MFMAs is shorter than an HBM round trip)
    constexpr int DEPTH = 3;
    const int n_full = K / 8;
    Frag f[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d)
        if (d < n_full) load_frag(8 * d, f[d]);
    int gi = 0;
    for (; gi + DEPTH <= n_full; gi += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int nxt = gi + d + DEPTH - 1;
            if (nxt < n_full) load_frag(8 * nxt, f[(d + DEPTH - 1) % DEPTH]);
            mma_frag(f[d]);
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d)          // the last (n_full % DEPTH) groups are already loaded
        if (gi + d < n_full) mma_frag(f[d]);
    if (n_full * 8 < K) {
        Frag ft;
        load_frag_tail(n_full * 8, ft);
        mma_frag(ft);
    }

    rows_epilogue<NT>(a, kk.n_tiles_m, acc, g, tile_m, m0, n0, i, s);
}

// ------------------------------------------------------------------------------------ weight pre-split
// thread = 4 consecutive k of one row n of W: the three planes of W ([N][ld]) with 8-byte stores and, element by
// element, of W^T ([K][ld_t]); pad columns are written as zeros (the GEMM stages whole 32-column chunks)
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ W, int64_t ldw, int N, int K, __bf16* P,
                                                            int64_t ld, __bf16* Pt, int64_t ld_t) {
    const int k4n = static_cast<int>(ld / 4);
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t rows = max<int64_t>(N, Pt ? ld_t : N);     // rows of the index space: also covers W^T's pad columns
    if (idx >= rows * k4n) return;
    const int n = static_cast<int>(idx / k4n), k = static_cast<int>(idx - static_cast<int64_t>(n) * k4n) * 4;
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = (n < N && k + c < K) ? W[static_cast<int64_t>(n) * ldw + k + c] : 0.f;
    bf16x4 h, m, l;
    SPLIT3_PAIR(v[0], v[1], h, m, l, 0);
    SPLIT3_PAIR(v[2], v[3], h, m, l, 2);
    if (P && n < N) {
        __bf16* d = P + static_cast<int64_t>(n) * ld + k;
        *reinterpret_cast<bf16x4*>(d) = h;
        *reinterpret_cast<bf16x4*>(d + static_cast<int64_t>(N) * ld) = m;
        *reinterpret_cast<bf16x4*>(d + 2 * static_cast<int64_t>(N) * ld) = l;
    }
    if (Pt && n < ld_t) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (k + c < K) {
                __bf16* d = Pt + static_cast<int64_t>(k + c) * ld_t + n;
                d[0] = h[c];
                d[static_cast<int64_t>(K) * ld_t] = m[c];
                d[2 * static_cast<int64_t>(K) * ld_t] = l[c];
            }
        }
    }
}

extern "C" int64_t swr_split_ld(int64_t cols) { return (cols + X6_KC - 1) / X6_KC * X6_KC; }

extern "C" int swr_split_weights(const float* W, int64_t ldw, int N, int K, void* planes, void* planes_t, void* stream) {
    SWR_REQUIRE(W && N > 0 && K > 0 && ldw >= K && (planes || planes_t), SWR_ERR_ARG);
    const int64_t ld = swr_split_ld(K), ld_t = swr_split_ld(N);
    const int64_t rows = std::max<int64_t>(N, planes_t ? ld_t : N);
    const int64_t items = rows * (ld / 4);
    hipLaunchKernelGGL(split_weights_kernel, dim3(static_cast<unsigned>(swr_ceil_div(items, 256))), dim3(256), 0,
                       static_cast<hipStream_t>(stream), W, ldw, N, K, static_cast<__bf16*>(planes), ld,
                       static_cast<__bf16*>(planes_t), ld_t);
    return swr_launch_status();
}

// SWR_GEMM=f32 forces the f32-MFMA kernels (default: bf16x3-split "x6" kernels where applicable); SWR_GEMM=bf16 selects the
// single-product perf mode of those kernels (operands rounded to bf16: NOT the parity path, its logit error is measured by
// tests/test_perf_mode_gpu.py and reported by bench.py as a separate, labelled line)
static int gemm_mode() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("SWR_GEMM");
        v = (e && (e[0] == 'f' || e[0] == 'F')) ? 0 : ((e && (e[0] == 'b' || e[0] == 'B')) ? 2 : 1);
    }
    return v;
}
// SWR_TN_WIDE=0: the blocked x6 weight-gradient kernels where the wide one would run (A/B measurements)
static bool tn_wide_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("SWR_TN_WIDE"); v = (e && e[0] == '0') ? 0 : 1; }
    return v != 0;
}
static bool use_x6() { return gemm_mode() != 0; }
static bool use_bf16() { return gemm_mode() == 2; }
extern "C" int swr_gemm_precision_mode() { return gemm_mode(); }

template <int NT, bool PRO, bool PS, bool BF = false>
static void launch_x6(dim3 grid, unsigned lds, hipStream_t st, const GemmK& kk) {
    // more than 64 KB of dynamic LDS needs the attribute once per instantiation (not a stream operation; the first
    // call of a shape happens in a warm-up step, never inside a hipGraph capture)
    if (lds > 64 * 1024) swr_raise_lds(reinterpret_cast<const void*>(gemm_rows_x6_kernel<NT, PRO, PS, BF>), 96 * 1024);
    hipLaunchKernelGGL((gemm_rows_x6_kernel<NT, PRO, PS, BF>), grid, dim3(GEMM_THREADS), lds, st, kk);
}

template <bool BT>
static int launch_rows(const swr_gemm_args* args, void* stream) {
    SWR_REQUIRE(args != nullptr, SWR_ERR_ARG);
    const swr_gemm_args& a = *args;
    SWR_REQUIRE(a.M >= 0 && a.N > 0 && a.K > 0 && a.A && a.B && a.C && a.groups >= 1, SWR_ERR_ARG);
    SWR_REQUIRE(a.lda >= a.K && a.ldc >= a.N && a.ldb >= (BT ? a.K : a.N), SWR_ERR_ARG);
    SWR_REQUIRE((a.a_scale == nullptr) == (a.a_shift == nullptr), SWR_ERR_ARG);
    SWR_REQUIRE(a.c_act >= 0 && a.c_act <= 2 && (a.c_act == 0 || (!a.accumulate && !a.stat_partials)), SWR_ERR_ARG);
    if (a.M == 0) return SWR_OK;
    GemmK kk;
    kk.a = a;
    kk.n_tiles_m = static_cast<int>(swr_ceil_div(a.M, 32));
    if (a.B_split) {
        SWR_REQUIRE(BT && a.groups == 1 && a.ld_split >= swr_split_ld(a.K) && a.ld_split % 4 == 0 &&
                        a.plane_stride >= static_cast<int64_t>(a.N) * a.ld_split && swr_aligned16(a.B_split), SWR_ERR_ARG);
    }
    const bool pro = a.a_scale != nullptr;
    bool vec = swr_aligned16(a.A) && a.lda % 4 == 0 && a.gsA % 4 == 0;
    vec = vec && swr_aligned16(a.B) && a.ldb % 4 == 0 && a.gsB % 4 == 0;      // both layouts are staged with 16-byte loads
    if (pro) vec = vec && swr_aligned16(a.a_scale) && swr_aligned16(a.a_shift) && a.gsScale % 4 == 0;
    const bool lds_ok = vec && a.K % 4 == 0 && a.K >= 4 && (BT || (a.N % 4 == 0 && a.N >= 4));
    const int tiles = static_cast<int>(swr_ceil_div(a.N, 32));
    const bool x6_ok = BT && lds_ok && use_x6();
    // tiles per wave: the f32 LDS kernel keeps two waves per SIMD up to 5 tiles (VGPR + AGPR <= 256), the bf16-split
    // kernel up to 7 (252 VGPRs).  Fewer, wider column groups = fewer re-reads / re-splits of A and less padding
    // (N = 516: 3 groups of 6 tiles instead of 4 of 5).
    int nblk = static_cast<int>(swr_ceil_div(tiles, x6_ok ? (a.B_split ? 6 : 7) : (lds_ok ? 5 : 8)));   // pre-split: 6 (VGPRs)
    // short batches: a workgroup is GEMM_WAVES x 32 rows, so M = 8192 is 64 row blocks -- PLE's first-layer dX (N = 96, one column
    // group) ran on 64 of the 256 CUs for 31 us.  While the grid is smaller than the chip, cut the columns into more groups
    // (each output element is still one K-ordered sum: results unchanged).
    const int64_t row_blocks = swr_ceil_div(kk.n_tiles_m, GEMM_WAVES);
    static int fill = -1;
    if (fill < 0) { const char* e = getenv("SWR_GEMM_FILL"); fill = (e && e[0] == '0') ? 0 : 1; }
    if (fill) {
        while (nblk < tiles && row_blocks * nblk * a.groups < 256) ++nblk;
        nblk = static_cast<int>(swr_ceil_div(tiles, swr_ceil_div(tiles, nblk)));      // no empty group
    }
    const int nt = static_cast<int>(swr_ceil_div(tiles, nblk));
    const dim3 grid(static_cast<unsigned>(swr_ceil_div(kk.n_tiles_m, GEMM_WAVES)), static_cast<unsigned>(nblk),
                    static_cast<unsigned>(a.groups));
    hipStream_t st = static_cast<hipStream_t>(stream);
#define X6_BYTES(NTV) static_cast<unsigned>((X6_DOUBLE_BUFFER(NTV) ? 2 : 1) * 3 * (NTV) * 32 * X6_PITCH * 2)
#define LDS_BYTES(NTV) static_cast<unsigned>(2 * LDS_KC * ((NTV) * 32 + 4) * sizeof(float))
#define GO(NTV)                                                                                                   \
    do {                                                                                                          \
        if (x6_ok && use_bf16() && !pro && !a.B_split) launch_x6<NTV, false, false, true>(grid, X6_BYTES(NTV), st, kk);   \
        else if (x6_ok && pro) launch_x6<NTV, true, false>(grid, X6_BYTES(NTV), st, kk);   \
        else if (x6_ok && a.B_split) launch_x6<NTV, false, true>(grid, X6_BYTES(NTV), st, kk);   \
        else if (x6_ok) launch_x6<NTV, false, false>(grid, X6_BYTES(NTV), st, kk);   \
        else if (lds_ok && pro) hipLaunchKernelGGL((gemm_rows_lds_kernel<NTV, BT, true>), grid, dim3(GEMM_THREADS), LDS_BYTES(NTV), st, kk);   \
        else if (lds_ok) hipLaunchKernelGGL((gemm_rows_lds_kernel<NTV, BT, false>), grid, dim3(GEMM_THREADS), LDS_BYTES(NTV), st, kk); \
        else if (vec && pro) hipLaunchKernelGGL((gemm_rows_kernel<NTV, BT, true, true>), grid, dim3(GEMM_THREADS), 0, st, kk);   \
        else if (vec) hipLaunchKernelGGL((gemm_rows_kernel<NTV, BT, true, false>), grid, dim3(GEMM_THREADS), 0, st, kk);    \
        else if (pro) hipLaunchKernelGGL((gemm_rows_kernel<NTV, BT, false, true>), grid, dim3(GEMM_THREADS), 0, st, kk);    \
        else hipLaunchKernelGGL((gemm_rows_kernel<NTV, BT, false, false>), grid, dim3(GEMM_THREADS), 0, st, kk);            \
    } while (0)
    switch (nt) {
        case 1: GO(1); break;
        case 2: GO(2); break;
        case 3: GO(3); break;
        case 4: GO(4); break;
        case 5: GO(5); break;
        case 6: GO(6); break;
        case 7: GO(7); break;
        default: GO(8); break;
    }
#undef GO
#undef LDS_BYTES
#undef X6_BYTES
    return swr_launch_status();
}

// out[g][k][n] = in[g][n][k]: the transposed weights of a (grouped) layer, the operand of its dX product in the [N, K] layout the
// bf16-split kernel stages (replaces an ATen strided copy per layer and step)
__global__ __launch_bounds__(256) void transpose_groups_kernel(const float* __restrict__ in, int N, int K, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int g = blockIdx.z, n0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* src = in + static_cast<int64_t>(g) * N * K;
    float* dst = out + static_cast<int64_t>(g) * N * K;
#pragma unroll
    for (int r = ty; r < 32; r += 8)
        if (n0 + r < N && k0 + tx < K) tile[r][tx] = src[static_cast<int64_t>(n0 + r) * K + k0 + tx];
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8)
        if (k0 + r < K && n0 + tx < N) dst[static_cast<int64_t>(k0 + r) * N + n0 + tx] = tile[tx][r];
}

extern "C" int swr_transpose_groups(const float* in, int groups, int N, int K, float* out, void* stream) {
    SWR_REQUIRE(in && out && groups >= 1 && N >= 1 && K >= 1 && groups <= 65535, SWR_ERR_ARG);
    hipLaunchKernelGGL(transpose_groups_kernel, dim3(static_cast<unsigned>((K + 31) / 32), static_cast<unsigned>((N + 31) / 32),
                                                      static_cast<unsigned>(groups)), dim3(256), 0, static_cast<hipStream_t>(stream), in, N, K, out);
    return swr_launch_status();
}

extern "C" int swr_gemm_nt(const swr_gemm_args* args, void* stream) { return launch_rows<true>(args, stream); }
extern "C" int swr_gemm_nn(const swr_gemm_args* args, void* stream) { return launch_rows<false>(args, stream); }

// ------------------------------------------------------------------------------------------------ tn
#define TN_TA_MAX 5
#define TN_TB 1

struct TnK {
    swr_gemm_tn_args a;
    int splits;
    int64_t rows_per_split;   // even
    float* part;              // [groups][splits][K1][K2]
    float* part_cs;           // [groups][splits][K1]  (colsum) or null
    unsigned qblk, pblk, n_tiles;   // column tiles, row-tile groups, qblk * pblk * zsplit * groups
    // x6 kernel with a ragged last column tile (K2 = 128 j + r, r <= 32): the tile is not a column block of its own;
    // block b of a batch split takes it on the stages with stage % qblk == b and writes its partial sums to replica b
    // (columns tail_q0 + 32 b .. of a partial row of pitch k2p); tn_reduce_kernel adds the replicas.  0 = no tail.
    int tail_q0, tail_rep, k2p;
};

template <int TA>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_tn_kernel(const TnK kk) {   // two waves per SIMD
    // the 4 waves of a workgroup take 4 consecutive row slices of the SAME output tile and are summed in LDS in a
    // fixed tree ((w0 + w1) + (w2 + w3)) before one partial tile per workgroup goes to the workspace
    extern __shared__ __attribute__((aligned(16))) float red[];
    const swr_gemm_tn_args& a = kk.a;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // SGPR
    const int i = lane & 31, s = lane >> 5;
    const int zsplit = (kk.splits + GEMM_WAVES - 1) / GEMM_WAVES;
    // XCD-aware tile order: workgroup ids go round-robin over the 8 XCDs, so id -> (id % 8) * (n / 8) + id / 8 puts
    // the workgroups that share one row slice of A (all column tiles q, then all row tiles p) on ONE XCD's L2
    const unsigned per_xcd = gridDim.x / 8;
    const unsigned lin = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (lin >= kk.n_tiles) return;
    const int bq = lin % kk.qblk, bp = (lin / kk.qblk) % kk.pblk, bz = lin / (kk.qblk * kk.pblk);
    const int part = bz % zsplit;
    const int g = bz / zsplit;
    const int split = part * GEMM_WAVES + wave;
    const int p0 = bp * (32 * TA);
    const int q0 = bq * (32 * TN_TB);
    const int64_t ms = min(static_cast<int64_t>(split) * kk.rows_per_split, a.M);
    const int64_t me = min(ms + kk.rows_per_split, a.M);

    const float* __restrict__ Ag = a.A + g * a.gsA;
    const float* __restrict__ Bg = a.B + g * a.gsB;
    bool pa[TA], pb[TN_TB];
    int ca[TA], cb[TN_TB];
#pragma unroll
    for (int t = 0; t < TA; ++t) { ca[t] = p0 + 32 * t + i; pa[t] = ca[t] < a.K1; ca[t] = min(ca[t], a.K1 - 1); }
#pragma unroll
    for (int t = 0; t < TN_TB; ++t) { cb[t] = q0 + 32 * t + i; pb[t] = cb[t] < a.K2; cb[t] = min(cb[t], a.K2 - 1); }

    f32x16 acc[TA][TN_TB];
#pragma unroll
    for (int ta = 0; ta < TA; ++ta)
#pragma unroll
        for (int tb = 0; tb < TN_TB; ++tb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ta][tb][r] = 0.f;
    float cs[TA];
#pragma unroll
    for (int t = 0; t < TA; ++t) cs[t] = 0.f;

    struct Frag {
        float a[TA];
        float b[TN_TB];
    };
    // main loop without predicates: columns beyond K1 / K2 are clamped (their products are never stored).
    // Addressing = wave-uniform row-pair base (SGPR pair, advanced by scalar adds) + a fixed 32-bit lane offset
    // (slot * ld + column): one global_load per operand with no per-load 64-bit vector arithmetic.
    uint32_t offa[TA], offb[TN_TB];
#pragma unroll
    for (int t = 0; t < TA; ++t) offa[t] = static_cast<uint32_t>(s * a.lda + ca[t]);
#pragma unroll
    for (int t = 0; t < TN_TB; ++t) offb[t] = static_cast<uint32_t>(s * a.ldb + cb[t]);
    auto load_frag = [&](int64_t mb, Frag& f) {               // rows mb, mb + 1 both valid; mb wave-uniform
        const float* __restrict__ ra = Ag + mb * a.lda;
        const float* __restrict__ rb = Bg + mb * a.ldb;
#pragma unroll
        for (int t = 0; t < TA; ++t) f.a[t] = ra[offa[t]];
#pragma unroll
        for (int t = 0; t < TN_TB; ++t) f.b[t] = rb[offb[t]];
    };
    auto mma_frag = [&](const Frag& f) {
#pragma unroll
        for (int ta = 0; ta < TA; ++ta)
#pragma unroll
            for (int tb = 0; tb < TN_TB; ++tb)
                acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[ta], f.b[tb], acc[ta][tb], 0, 0, 0);
        // column sums as opaque scalar adds: left to the SLP vectoriser they become v_pk_add_f32 on register pairs
        // assembled with v_mov from freshly loaded ring slots, which forces a wait on nearly every load in flight
#pragma unroll
        for (int t = 0; t < TA; ++t) asm volatile("v_add_f32 %0, %0, %1" : "+v"(cs[t]) : "v"(f.a[t]));
    };
    const int64_t me2 = ms + ((me - ms) / 2) * 2;              // rows taken in pairs
    // wave-uniform trip count in an SGPR: the loop control is scalar and the loads of the main loop are
    // unconditional (pair index clamped to the last valid pair), so the compiler can keep DEPTH-1 row pairs in flight
    // with exact vmcnt waits instead of draining the queue at every conditional load
    const int n_pairs = __builtin_amdgcn_readfirstlane(static_cast<int>((me2 - ms) / 2));
    if (n_pairs > 0) {
        constexpr int DEPTH = 6;
        const int last = n_pairs - 1;
        Frag f[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d) load_frag(ms + 2 * static_cast<int64_t>(min(d, last)), f[d]);
        int pi = 0;
        for (; pi + DEPTH <= n_pairs; pi += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                // the scheduler must not sink the loads or hoist the products across a ring step (it otherwise
                // shortens the prefetch distance to ~1 step to save registers)
                load_frag(ms + 2 * static_cast<int64_t>(min(pi + d + DEPTH - 1, last)), f[(d + DEPTH - 1) % DEPTH]);
                __builtin_amdgcn_sched_barrier(0);
                mma_frag(f[d]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d)
            if (pi + d < n_pairs) mma_frag(f[d]);
    }
    if (me2 < me) {                                            // odd last row: slot 1 contributes zeros
        Frag f;
#pragma unroll
        for (int t = 0; t < TA; ++t) f.a[t] = s == 0 ? Ag[me2 * a.lda + ca[t]] : 0.f;
#pragma unroll
        for (int t = 0; t < TN_TB; ++t) f.b[t] = s == 0 ? Bg[me2 * a.ldb + cb[t]] : 0.f;
        mma_frag(f);
    }

    // ---- workgroup reduction through LDS: [TA*TN_TB*16][64] floats + column sums
    constexpr int PER_WAVE = TA * TN_TB * 16 * 64;
    auto put = [&](float* dst) {
#pragma unroll
        for (int ta = 0; ta < TA; ++ta)
#pragma unroll
            for (int tb = 0; tb < TN_TB; ++tb)
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[((ta * TN_TB + tb) * 16 + r) * 64 + lane] = acc[ta][tb][r];
    };
    auto take = [&](const float* src) {
#pragma unroll
        for (int ta = 0; ta < TA; ++ta)
#pragma unroll
            for (int tb = 0; tb < TN_TB; ++tb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ta][tb][r] += src[((ta * TN_TB + tb) * 16 + r) * 64 + lane];
    };
    float* csr = red + PER_WAVE;         // [4][TA][64] column-sum partials
#pragma unroll
    for (int t = 0; t < TA; ++t) csr[(wave * TA + t) * 64 + lane] = cs[t];
    // one 40 KB staging buffer, three rounds: ((w0 + w1) + w2) + w3 -- keeps 3 workgroups resident per CU
#pragma unroll
    for (int w = 1; w < GEMM_WAVES; ++w) {
        if (wave == w) put(red);
        __syncthreads();
        if (wave == 0) take(red);
        __syncthreads();
    }
    if (wave != 0) return;

    float* __restrict__ P = kk.part + (static_cast<int64_t>(g) * zsplit + part) * a.K1 * a.K2;
#pragma unroll
    for (int ta = 0; ta < TA; ++ta)
#pragma unroll
        for (int tb = 0; tb < TN_TB; ++tb) {
            const int q = q0 + 32 * tb + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = p0 + 32 * ta + (r & 3) + 8 * (r >> 2) + 4 * s;
                if (p < a.K1 && q < a.K2) P[static_cast<int64_t>(p) * a.K2 + q] = acc[ta][tb][r];
            }
        }
    if (kk.part_cs && bq == 0) {
#pragma unroll
        for (int t = 0; t < TA; ++t) {
            float tot = ((csr[(0 * TA + t) * 64 + lane] + csr[(1 * TA + t) * 64 + lane]) +
                         csr[(2 * TA + t) * 64 + lane]) + csr[(3 * TA + t) * 64 + lane];
            tot += __shfl_xor(tot, 32);
            const int p = p0 + 32 * t + i;
            if (s == 0 && p < a.K1) kk.part_cs[(static_cast<int64_t>(g) * zsplit + part) * a.K1 + p] = tot;
        }
    }
}

// ------------------------------------------------------------------------- tn on the bf16 MFMA (x6 split)
// The weight-gradient product has its reduction index (the batch) as the row index of BOTH operands, so the 8
// consecutive k a bf16 MFMA fragment needs are a column slice: nothing a lane can load directly.  A workgroup of 8
// waves stages TX_ROWS batch rows of A[:, 0:K1] (all of it, <= 160 columns) and of its 128 columns of B with coalesced
// loads (a thread = 8 rows x 2 columns, three stages in flight in registers), splits every value into three bf16
// terms ONCE and writes them transposed into LDS planes [buffer][plane][column][row] with 16-byte stores; fragments
// are then single conflict-free ds_read_b128.  Waves 0-3 / 4-7 own q-tile (w & 3) and the first / second half of the
// p-tiles; the planes are double-buffered: one barrier per 16 batch rows.  Partial tiles per batch split go to the
// workspace and are summed by tn_reduce_kernel in fixed order.
#define TX_ROWS 16
#define TX_PM 24                        // bf16 pitch along the batch rows: 16 + 8 pad (48 bytes: conflict-free b128 access)
#define TX_QCOLS 128                    // B columns per workgroup (4 q-tiles)
#define TX_THREADS 512
#define TX_DEPTH 3                      // stages in flight in registers

template <int PT, bool TAIL, bool BF>
__global__ __launch_bounds__(TX_THREADS) void gemm_tn_x6_kernel(const TnK kk) {
    constexpr int ACOLS = PT * 32, COLS = ACOLS + TX_QCOLS + (TAIL ? 32 : 0);    // slab: A | 128 columns of B | tail tile
    constexpr int UNITS = 2 * (COLS / 2);                              // (row oct, column pair)
    constexpr int PA = (PT + 1) / 2;                                   // p-tiles of the first wave group
    constexpr int PLANE = COLS * TX_PM, BUF = 3 * PLANE;               // bf16 elements
    extern __shared__ __attribute__((aligned(16))) __bf16 Lx[];        // [2][3][COLS][TX_PM]
    const swr_gemm_tn_args& a = kk.a;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, s = lane >> 5;
    const int qt = wave & 3, p_first = (wave >> 2) ? PA : 0;
    const int p_count = (wave >> 2) ? PT - PA : PA;
    // XCD-aware order (workgroup ids go round-robin over the 8 XCDs): the q-blocks of one batch split, which all read the
    // same rows of A, are consecutive ids of ONE XCD -> A comes from HBM once, not once per q-block
    const unsigned per_xcd = gridDim.x / 8;
    const unsigned lin = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (lin >= kk.n_tiles) return;
    const int qb = static_cast<int>(lin % kk.qblk);
    const int q0 = qb * TX_QCOLS;
    const int pb = static_cast<int>((lin / kk.qblk) % kk.pblk);        // A wider than 5 tiles: column blocks of PT tiles
    const int p0 = pb * ACOLS;
    const int split = static_cast<int>(lin / (kk.qblk * kk.pblk));
    const int64_t ms = min(static_cast<int64_t>(split) * kk.rows_per_split, a.M);
    const int64_t me = min(ms + kk.rows_per_split, a.M);

    f32x16 acc[PA];
#pragma unroll
    for (int t = 0; t < PA; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // tail tile x p-tiles (wave - 4) and (wave - 4) + 4, waves 4-7 only (the wave group with the smaller share)
    constexpr int TT = TAIL ? (PT > 4 ? 2 : 1) : 1;
    f32x16 acc_tail[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_tail[t][r] = 0.f;

    // this thread's staging unit: row oct `ro` (rows 8 ro .. 8 ro + 7 of the stage), column pair `cp` of the slab
    // the staging units go to waves 4-7 first (they own the smaller half of the p-tiles: their split work overlaps the
    // other wave group's MFMAs on the same SIMD), the rest to wave 0
    const int uid = (threadIdx.x + TX_THREADS / 2) & (TX_THREADS - 1);
    const bool unit_on = uid < UNITS;
    const int ro = uid & 1, cp = unit_on ? (uid >> 1) : 0;
    const int scol = 2 * cp;                                           // slab column (A columns, then the 128 B columns)
    const bool isA = scol < ACOLS;
    const bool isT = TAIL && scol >= ACOLS + TX_QCOLS;
    const int gcol = isA ? p0 + scol : (isT ? kk.tail_q0 + (scol - ACOLS - TX_QCOLS) : q0 + (scol - ACOLS));
    const bool col_ok = unit_on && (isA ? gcol < a.K1 : gcol < a.K2);  // K1, K2 even
    const float* __restrict__ src = (isA ? a.A : a.B) + (col_ok ? gcol : 0);
    const int64_t ld = isA ? a.lda : a.ldb;
    float cs0 = 0.f, cs1 = 0.f;                                        // column sums of A (bias gradient)

    float2 st[TX_DEPTH][8];
    // raw loads only: any arithmetic on a loaded value here would make the compiler wait for it here, three stages
    // before it is needed; rows past the end / invalid columns are zeroed when the stage is stored.  32-bit element
    // offsets (checked by the launcher), advanced by a scalar per stage and clamped to the last row of the matrix.
    const uint32_t ld32 = static_cast<uint32_t>(ld);
    const uint32_t off_last = static_cast<uint32_t>((a.M - 1 - ms) * ld);                // row M - 1, relative to row ms
    const float* __restrict__ base = src + ms * ld;
    uint32_t roff[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) roff[r] = static_cast<uint32_t>(8 * ro + r) * ld32;
    auto stage_load = [&](int stage, float2 (&dst)[8]) {
        const uint32_t so = static_cast<uint32_t>(stage) * (TX_ROWS * ld32);             // wave-uniform
#pragma unroll
        for (int r = 0; r < 8; ++r) dst[r] = *reinterpret_cast<const float2*>(base + min(so + roff[r], off_last));
    };
    const int rows_total = static_cast<int>(me - ms);
    auto stage_store = [&](int stage, const float2 (&raw)[8], __bf16* buf) {
        if (!col_ok) return;                      // columns past the matrix were zeroed once (prologue)
        const int left = rows_total - stage * TX_ROWS - 8 * ro;                          // valid rows of this unit
        bf16x8 h0, m0_, l0, h1, m1, l1;
        if (left >= 8) {                                                                 // (almost always)
#pragma unroll
            for (int r = 0; r < 8; r += 2) {
                SPLIT3_PAIR(raw[r].x, raw[r + 1].x, h0, m0_, l0, r);
                SPLIT3_PAIR(raw[r].y, raw[r + 1].y, h1, m1, l1, r);
                cs0 += raw[r].x;
                cs1 += raw[r].y;
                cs0 += raw[r + 1].x;
                cs1 += raw[r + 1].y;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const bool ok = r < left;
                const float vx = ok ? raw[r].x : 0.f, vy = ok ? raw[r].y : 0.f;
                SPLIT3_INTO(vx, h0, m0_, l0, r);
                SPLIT3_INTO(vy, h1, m1, l1, r);
                cs0 += vx;
                cs1 += vy;
            }
        }
        __bf16* d = buf + scol * TX_PM + 8 * ro;
        *reinterpret_cast<bf16x8*>(d) = h0;
        *reinterpret_cast<bf16x8*>(d + PLANE) = m0_;
        *reinterpret_cast<bf16x8*>(d + 2 * PLANE) = l0;
        *reinterpret_cast<bf16x8*>(d + TX_PM) = h1;
        *reinterpret_cast<bf16x8*>(d + TX_PM + PLANE) = m1;
        *reinterpret_cast<bf16x8*>(d + TX_PM + 2 * PLANE) = l1;
    };
    auto frag = [&](bf16x8 (&f)[3], const __bf16* buf, int col) {
        const __bf16* p = buf + (col + i) * TX_PM + 8 * s;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) f[pl] = *reinterpret_cast<const bf16x8*>(p + pl * PLANE);
    };

    const int n_stages = static_cast<int>((me - ms + TX_ROWS - 1) / TX_ROWS);
    // prologue: stages 0 .. DEPTH-1 in flight, stage 0 into buffer 0, then stage DEPTH takes its slot
#pragma unroll
    for (int d = 0; d < TX_DEPTH; ++d) stage_load(d, st[d]);           // beyond the end: zeros (never used)
    if (unit_on && !col_ok) {                                          // slab columns past the matrix: zero in both buffers, once
        bf16x8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = static_cast<__bf16>(0.f);
#pragma unroll
        for (int bsel = 0; bsel < 2; ++bsel)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                __bf16* d = Lx + bsel * BUF + pl * PLANE + scol * TX_PM + 8 * ro;
                *reinterpret_cast<bf16x8*>(d) = z;
                *reinterpret_cast<bf16x8*>(d + TX_PM) = z;
            }
    }
    stage_store(0, st[0], Lx);
    stage_load(TX_DEPTH, st[0]);
    __syncthreads();
    // steady state, unrolled by DEPTH so that the register slots are static
    auto step = [&](int sg, auto slot_c) {
        constexpr int NEXT = (decltype(slot_c)::value + 1) % TX_DEPTH;  // slot holding stage sg + 1
        const __bf16* buf = Lx + (sg & 1) * BUF;
        bf16x8 b[3], a0[3], a1[3];
        frag(b, buf, ACOLS + 32 * qt);
        frag(a0, buf, 32 * p_first);
#pragma unroll
        for (int t = 0; t < PA; ++t) {
            if (t < p_count) {
                bf16x8 (&af)[3] = (t & 1) ? a1 : a0;
                bf16x8 (&nx)[3] = (t & 1) ? a0 : a1;
                if (t + 1 < PA && t + 1 < p_count) frag(nx, buf, 32 * (p_first + t + 1));
                __builtin_amdgcn_sched_barrier(0);      // the next tile's LDS reads are issued before this tile's MFMAs
                f32x16 c_ = acc[t];
                if (!BF) {
                    if (!SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], b[0], c_, 0, 0, 0);     // small terms first
                    if (!SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], b[2], c_, 0, 0, 0);
                    if (!SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], b[1], c_, 0, 0, 0);
                    c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], b[0], c_, 0, 0, 0);
                    c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], b[1], c_, 0, 0, 0);
                }
                c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], b[0], c_, 0, 0, 0);
                acc[t] = c_;
            }
        }
        if (TAIL && wave >= 4 && (sg % static_cast<int>(kk.qblk)) == qb) {       // this block's turn at the tail tile
            bf16x8 tb[3], ta[3];
            frag(tb, buf, ACOLS + TX_QCOLS);
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const int pt_ = (wave - 4) + 4 * t;
                if (pt_ < PT) {
                    frag(ta, buf, 32 * pt_);
                    f32x16 c_ = acc_tail[t];
                    if (!BF) {
                        if (!SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[2], tb[0], c_, 0, 0, 0);
                        if (!SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[0], tb[2], c_, 0, 0, 0);
                        if (!SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[1], tb[1], c_, 0, 0, 0);
                        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[1], tb[0], c_, 0, 0, 0);
                        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[0], tb[1], c_, 0, 0, 0);
                    }
                    c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[0], tb[0], c_, 0, 0, 0);
                    acc_tail[t] = c_;
                }
            }
        }
        if (sg + 1 < n_stages) stage_store(sg + 1, st[NEXT], Lx + ((sg + 1) & 1) * BUF);
        stage_load(sg + 1 + TX_DEPTH, st[NEXT]);
        __syncthreads();
    };
    int sg = 0;
    for (; sg + TX_DEPTH <= n_stages; sg += TX_DEPTH) {
        step(sg, std::integral_constant<int, 0>{});
        step(sg + 1, std::integral_constant<int, 1>{});
        step(sg + 2, std::integral_constant<int, 2>{});
    }
    if (sg < n_stages) { step(sg, std::integral_constant<int, 0>{}); ++sg; }
    if (sg < n_stages) { step(sg, std::integral_constant<int, 1>{}); ++sg; }

    // partial tile of this batch split -> workspace [split][K1][k2p]
    const int k2p = kk.k2p;
    float* __restrict__ P = kk.part + static_cast<int64_t>(split) * a.K1 * k2p;
    const int q = q0 + 32 * qt + i;
    const int q_end = TAIL ? kk.tail_q0 : a.K2;
#pragma unroll
    for (int t = 0; t < PA; ++t) {
        if (t < p_count) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = p0 + 32 * (p_first + t) + (r & 3) + 8 * (r >> 2) + 4 * s;
                if (p < a.K1 && q < q_end) P[static_cast<int64_t>(p) * k2p + q] = acc[t][r];
            }
        }
    }
    if (TAIL && wave >= 4) {                                           // replica qb of the tail tile's partial sums
        const int qtl = kk.tail_q0 + 32 * qb + i;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const int pt_ = (wave - 4) + 4 * t;
            if (pt_ < PT) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = p0 + 32 * pt_ + (r & 3) + 8 * (r >> 2) + 4 * s;
                    if (p < a.K1 && kk.tail_q0 + i < a.K2) P[static_cast<int64_t>(p) * k2p + qtl] = acc_tail[t][r];
                }
            }
        }
    }
    // column sums of A: the two row-oct lanes of a column pair are adjacent lanes
    if (kk.part_cs && q0 == 0) {
        cs0 += __shfl_xor(cs0, 1);
        cs1 += __shfl_xor(cs1, 1);
        if (col_ok && isA && ro == 0) {
            kk.part_cs[static_cast<int64_t>(split) * a.K1 + gcol] = cs0;
            kk.part_cs[static_cast<int64_t>(split) * a.K1 + gcol + 1] = cs1;
        }
    }
}

// ---- the same product with the second operand GATHERED (tn_gather.h; csrc/first_layer.hip): the staging thread of a
// column pair of A' reads its 8 rows' keys (two 16-byte loads, one stage ahead of the values), then 8-byte pieces of the
// table rows the keys name -- the access granularity is the one the written block had (8 bytes per (row, column pair)),
// only the rows now come from the tables (L2-resident but for the row-sparse ones) instead of a 72 MB block; one-hot column
// pairs are expanded from two bits of the sample's mask word and, being exact in bf16, take three products instead of six.
#include "tn_gather.h"
#include "dw_tr.h"
template <int PT, bool TAIL>
__global__ __launch_bounds__(TX_THREADS) void gemm_tn_x6g_kernel(const TnK kk, const TnGather g) {
    constexpr int ACOLS = PT * 32, COLS = ACOLS + TX_QCOLS + (TAIL ? 32 : 0);
    constexpr int UNITS = 2 * (COLS / 2);
    constexpr int PA = (PT + 1) / 2;
    constexpr int PLANE = COLS * TX_PM, BUF = 3 * PLANE;
    extern __shared__ __attribute__((aligned(16))) __bf16 Lx[];        // [2][3][COLS][TX_PM]
    const swr_gemm_tn_args& a = kk.a;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, s = lane >> 5;
    const int qt = wave & 3, p_first = (wave >> 2) ? PA : 0;
    const int p_count = (wave >> 2) ? PT - PA : PA;
    const unsigned per_xcd = gridDim.x / 8;
    const unsigned lin = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (lin >= kk.n_tiles) return;
    const int qb = static_cast<int>(lin % kk.qblk);
    const int q0 = qb * TX_QCOLS;
    const int pb = static_cast<int>((lin / kk.qblk) % kk.pblk);
    const int p0 = pb * ACOLS;
    const int split = static_cast<int>(lin / (kk.qblk * kk.pblk));
    const int64_t ms = min(static_cast<int64_t>(split) * kk.rows_per_split, a.M);
    const int64_t me = min(ms + kk.rows_per_split, a.M);
    // exact (one-hot) column tiles: three products instead of six (wave-uniform)
    const bool q_exact = q0 + 32 * qt >= g.kp;
    const bool t_exact = TAIL && kk.tail_q0 >= g.kp;

    f32x16 acc[PA];
#pragma unroll
    for (int t = 0; t < PA; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    constexpr int TT = TAIL ? (PT > 4 ? 2 : 1) : 1;
    f32x16 acc_tail[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_tail[t][r] = 0.f;

    // ---- this thread's staging unit and where its column pair comes from
    const int uid = (threadIdx.x + TX_THREADS / 2) & (TX_THREADS - 1);
    const bool unit_on = uid < UNITS;
    const int ro = uid & 1, cp = unit_on ? (uid >> 1) : 0;
    const int scol = 2 * cp;
    const bool isA = scol < ACOLS;
    const bool isT = TAIL && scol >= ACOLS + TX_QCOLS;
    const int gcol = isA ? p0 + scol : (isT ? kk.tail_q0 + (scol - ACOLS - TX_QCOLS) : q0 + (scol - ACOLS));
    bool col_ok = unit_on && (isA ? gcol < a.K1 : gcol < a.K2);
    const float* __restrict__ vbase = a.A;      // (a harmless address for the units that load nothing)
    uint32_t vstride = 0, kmax = 0;
    // (units without keys read A as "keys": M * 4 bytes into it and never used -- the key loads are unconditional so that every
    // step issues a fixed number of loads and the waits are counted; with `if (keyed)` hipcc waited for vmcnt(0) every stage)
    const uint32_t* __restrict__ kwp = reinterpret_cast<const uint32_t*>(a.A);
    bool keyed = false, is_oh = false;
    int bit = 0;
    if (isA) {
        if (col_ok) { vbase = a.A + gcol; vstride = static_cast<uint32_t>(a.lda); }
    } else if (col_ok) {
        const TnGatherPiece P = g.piece[min(gcol >> 3, g.n_pieces - 1)];
        const int e = gcol & 7;
        if (P.kind == TNG_TABLE) { vbase = P.vbase + e; vstride = P.vstride; kwp = P.kwp; kmax = P.kmax; keyed = true; }
        else if (P.kind == TNG_ROWIDX && e < P.n_valid) { vbase = P.vbase + e; vstride = P.vstride; }
        else if (P.kind == TNG_ONEHOT) { kwp = P.kwp; keyed = true; is_oh = true; bit = P.bit0 + e; }
        else col_ok = false;
    }
    // the middle / low planes of a one-hot column are read only when its 32-column tile also holds real columns
    const bool oh_planes = is_oh && (gcol & ~31) < g.kp;
    float cs0 = 0.f, cs1 = 0.f;

    typedef uint32_t kw_t[8];
    float2 st[TX_DEPTH][8];
    uint32_t ob[TX_DEPTH];                  // one-hot units: two bits per row of the stage
    kw_t kw[2];                             // keys / mask words of the stage whose values are loaded NEXT step
    const uint32_t m_last = static_cast<uint32_t>(a.M - 1);
    const uint32_t row0 = static_cast<uint32_t>(ms) + 8u * ro;
    auto kw_load = [&](int stage, kw_t& dst) {
        // 8 consecutive rows' words (read past row M - 1 at the very end: still inside the workspace, see TnGatherPiece)
        const uint4* p = reinterpret_cast<const uint4*>(kwp + (row0 + static_cast<uint32_t>(stage) * TX_ROWS));
        const uint4 x = p[0], y = p[1];
        dst[0] = x.x; dst[1] = x.y; dst[2] = x.z; dst[3] = x.w; dst[4] = y.x; dst[5] = y.y; dst[6] = y.z; dst[7] = y.w;
    };
    auto val_load = [&](int stage, const kw_t& k8, float2 (&dst)[8], uint32_t& bits) {
        const uint32_t rbase = row0 + static_cast<uint32_t>(stage) * TX_ROWS;
        uint32_t b2 = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint32_t idx = keyed ? min(k8[r], kmax) : min(rbase + r, m_last);
            dst[r] = *reinterpret_cast<const float2*>(vbase + static_cast<uint64_t>(idx) * vstride);
            b2 |= ((k8[r] >> bit) & 3u) << (2 * r);
        }
        bits = b2;
    };
    const int rows_total = static_cast<int>(me - ms);
    auto stage_store = [&](int stage, const float2 (&raw)[8], uint32_t bits, __bf16* buf) {
        if (!col_ok) return;
        const int left = rows_total - stage * TX_ROWS - 8 * ro;
        bf16x8 h0, m0_, l0, h1, m1, l1;
        __bf16* d = buf + scol * TX_PM + 8 * ro;
        if (is_oh) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const bool ok = r < left;
                h0[r] = static_cast<__bf16>((ok && ((bits >> (2 * r)) & 1u)) ? 1.f : 0.f);
                h1[r] = static_cast<__bf16>((ok && ((bits >> (2 * r + 1)) & 1u)) ? 1.f : 0.f);
                m0_[r] = static_cast<__bf16>(0.f); l0[r] = m0_[r]; m1[r] = m0_[r]; l1[r] = m0_[r];
            }
            *reinterpret_cast<bf16x8*>(d) = h0;
            *reinterpret_cast<bf16x8*>(d + TX_PM) = h1;
            if (oh_planes) {
                *reinterpret_cast<bf16x8*>(d + PLANE) = m0_;
                *reinterpret_cast<bf16x8*>(d + 2 * PLANE) = l0;
                *reinterpret_cast<bf16x8*>(d + TX_PM + PLANE) = m1;
                *reinterpret_cast<bf16x8*>(d + TX_PM + 2 * PLANE) = l1;
            }
            return;
        }
        if (left >= 8) {
#pragma unroll
            for (int r = 0; r < 8; r += 2) {
                SPLIT3_PAIR(raw[r].x, raw[r + 1].x, h0, m0_, l0, r);
                SPLIT3_PAIR(raw[r].y, raw[r + 1].y, h1, m1, l1, r);
                cs0 += raw[r].x;
                cs1 += raw[r].y;
                cs0 += raw[r + 1].x;
                cs1 += raw[r + 1].y;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const bool ok = r < left;
                const float vx = ok ? raw[r].x : 0.f, vy = ok ? raw[r].y : 0.f;
                SPLIT3_INTO(vx, h0, m0_, l0, r);
                SPLIT3_INTO(vy, h1, m1, l1, r);
                cs0 += vx;
                cs1 += vy;
            }
        }
        *reinterpret_cast<bf16x8*>(d) = h0;
        *reinterpret_cast<bf16x8*>(d + PLANE) = m0_;
        *reinterpret_cast<bf16x8*>(d + 2 * PLANE) = l0;
        *reinterpret_cast<bf16x8*>(d + TX_PM) = h1;
        *reinterpret_cast<bf16x8*>(d + TX_PM + PLANE) = m1;
        *reinterpret_cast<bf16x8*>(d + TX_PM + 2 * PLANE) = l1;
    };
    auto frag = [&](bf16x8 (&f)[3], const __bf16* buf, int col) {
        const __bf16* p = buf + (col + i) * TX_PM + 8 * s;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) f[pl] = *reinterpret_cast<const bf16x8*>(p + pl * PLANE);
    };
    auto frag1 = [&](bf16x8 (&f)[3], const __bf16* buf, int col) {          // the high plane only (exact columns)
        f[0] = *reinterpret_cast<const bf16x8*>(buf + (col + i) * TX_PM + 8 * s);
    };

    const int n_stages = static_cast<int>((me - ms + TX_ROWS - 1) / TX_ROWS);
    // prologue: keys of stages 0 .. DEPTH, values of stages 0 .. DEPTH - 1 in flight, stage 0 into buffer 0
#pragma unroll
    for (int r = 0; r < 8; ++r) { kw[0][r] = 0u; kw[1][r] = 0u; }
    {
        kw_t k0, k1, k2;
#pragma unroll
        for (int r = 0; r < 8; ++r) { k0[r] = 0u; k1[r] = 0u; k2[r] = 0u; }
        kw_load(0, k0); kw_load(1, k1); kw_load(2, k2);
        kw_load(TX_DEPTH, kw[0]);                                          // values of stage DEPTH are loaded in step 0
        val_load(0, k0, st[0], ob[0]);
        val_load(1, k1, st[1], ob[1]);
        val_load(2, k2, st[2], ob[2]);
    }
    static_assert(TX_DEPTH == 3, "prologue written for three stages in flight");
    if (unit_on && !col_ok) {
        bf16x8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = static_cast<__bf16>(0.f);
#pragma unroll
        for (int bsel = 0; bsel < 2; ++bsel)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                __bf16* d = Lx + bsel * BUF + pl * PLANE + scol * TX_PM + 8 * ro;
                *reinterpret_cast<bf16x8*>(d) = z;
                *reinterpret_cast<bf16x8*>(d + TX_PM) = z;
            }
    }
    stage_store(0, st[0], ob[0], Lx);
    kw_load(TX_DEPTH + 1, kw[1]);
    val_load(TX_DEPTH, kw[0], st[0], ob[0]);
    __syncthreads();
    // steady state, unrolled by 2 * DEPTH: the value slots (3) and the key slots (2) are both static
    auto step = [&](int sg, auto slot_c, auto kslot_c) {
        constexpr int NEXT = (decltype(slot_c)::value + 1) % TX_DEPTH;      // slot holding stage sg + 1
        constexpr int KS = decltype(kslot_c)::value;                       // key slot holding stage sg + 1 + DEPTH
        const __bf16* buf = Lx + (sg & 1) * BUF;
        bf16x8 b[3], a0[3], a1[3];
        if (q_exact) frag1(b, buf, ACOLS + 32 * qt); else frag(b, buf, ACOLS + 32 * qt);
        frag(a0, buf, 32 * p_first);
#pragma unroll
        for (int t = 0; t < PA; ++t) {
            if (t < p_count) {
                bf16x8 (&af)[3] = (t & 1) ? a1 : a0;
                bf16x8 (&nx)[3] = (t & 1) ? a0 : a1;
                if (t + 1 < PA && t + 1 < p_count) frag(nx, buf, 32 * (p_first + t + 1));
                __builtin_amdgcn_sched_barrier(0);
                f32x16 c_ = acc[t];
                if (!q_exact) {
                    if (!SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], b[0], c_, 0, 0, 0);
                    if (!SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], b[2], c_, 0, 0, 0);
                    if (!SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], b[1], c_, 0, 0, 0);
                    c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], b[0], c_, 0, 0, 0);
                    c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], b[1], c_, 0, 0, 0);
                    c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], b[0], c_, 0, 0, 0);
                } else {
                    if (!SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], b[0], c_, 0, 0, 0);
                    c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], b[0], c_, 0, 0, 0);
                    c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], b[0], c_, 0, 0, 0);
                }
                acc[t] = c_;
            }
        }
        if (TAIL && wave >= 4 && (sg % static_cast<int>(kk.qblk)) == qb) {
            bf16x8 tb[3], ta[3];
            if (t_exact) frag1(tb, buf, ACOLS + TX_QCOLS); else frag(tb, buf, ACOLS + TX_QCOLS);
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const int pt_ = (wave - 4) + 4 * t;
                if (pt_ < PT) {
                    frag(ta, buf, 32 * pt_);
                    f32x16 c_ = acc_tail[t];
                    if (!t_exact) {
                        if (!SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[2], tb[0], c_, 0, 0, 0);
                        if (!SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[0], tb[2], c_, 0, 0, 0);
                        if (!SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[1], tb[1], c_, 0, 0, 0);
                        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[1], tb[0], c_, 0, 0, 0);
                        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[0], tb[1], c_, 0, 0, 0);
                        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[0], tb[0], c_, 0, 0, 0);
                    } else {
                        if (!SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[2], tb[0], c_, 0, 0, 0);
                        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[1], tb[0], c_, 0, 0, 0);
                        c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[0], tb[0], c_, 0, 0, 0);
                    }
                    acc_tail[t] = c_;
                }
            }
        }
        if (sg + 1 < n_stages) stage_store(sg + 1, st[NEXT], ob[NEXT], Lx + ((sg + 1) & 1) * BUF);
        // keys of stage sg + 2 + DEPTH first, then the values of stage sg + 1 + DEPTH through the keys loaded a step ago
        kw_load(sg + 2 + TX_DEPTH, kw[KS ^ 1]);
        val_load(sg + 1 + TX_DEPTH, kw[KS], st[NEXT], ob[NEXT]);
        __syncthreads();
    };
    // (key slot of step sg: stage sg + 1 + DEPTH was loaded into kw[(sg + 1) & 1] -- step 0 reads kw[1])
    int sg = 0;
    for (; sg + 2 * TX_DEPTH <= n_stages; sg += 2 * TX_DEPTH) {
        step(sg, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        step(sg + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
        step(sg + 2, std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
        step(sg + 3, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        step(sg + 4, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        step(sg + 5, std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
    }
    if (sg < n_stages) { step(sg, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}); ++sg; }
    if (sg < n_stages) { step(sg, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}); ++sg; }
    if (sg < n_stages) { step(sg, std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}); ++sg; }
    if (sg < n_stages) { step(sg, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}); ++sg; }
    if (sg < n_stages) { step(sg, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}); ++sg; }

    const int k2p = kk.k2p;
    float* __restrict__ P = kk.part + static_cast<int64_t>(split) * a.K1 * k2p;
    const int q = q0 + 32 * qt + i;
    const int q_end = TAIL ? kk.tail_q0 : a.K2;
#pragma unroll
    for (int t = 0; t < PA; ++t) {
        if (t < p_count) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = p0 + 32 * (p_first + t) + (r & 3) + 8 * (r >> 2) + 4 * s;
                if (p < a.K1 && q < q_end) P[static_cast<int64_t>(p) * k2p + q] = acc[t][r];
            }
        }
    }
    if (TAIL && wave >= 4) {
        const int qtl = kk.tail_q0 + 32 * qb + i;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const int pt_ = (wave - 4) + 4 * t;
            if (pt_ < PT) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = p0 + 32 * pt_ + (r & 3) + 8 * (r >> 2) + 4 * s;
                    if (p < a.K1 && kk.tail_q0 + i < a.K2) P[static_cast<int64_t>(p) * k2p + qtl] = acc_tail[t][r];
                }
            }
        }
    }
    if (kk.part_cs && q0 == 0) {
        cs0 += __shfl_xor(cs0, 1);
        cs1 += __shfl_xor(cs1, 1);
        if (col_ok && isA && ro == 0) {
            kk.part_cs[static_cast<int64_t>(split) * a.K1 + gcol] = cs0;
            kk.part_cs[static_cast<int64_t>(split) * a.K1 + gcol + 1] = cs1;
        }
    }
}

// ---- the WIDE form of the gathered product: ONE workgroup per batch split stages A (<= 5 column tiles) and ALL column tiles of
// A' (the slab holds CT <= 14 tiles of 32 columns: 129 KB of LDS), so dZ is read, split and transposed once per batch row
// instead of once per 128-column block of A'.  The two kinds of work have waves of their own: waves 4-7 stage (a thread = up
// to two units of 8 rows x 2 columns, two stages in flight in registers), waves 0-3 multiply -- every SIMD holds one wave of
// each kind, so the VALU work of the split and the matrix pipe overlap by construction instead of alternating in lockstep
// between barriers.  The (p, q) output tiles of a multiplying wave are a COMPILE-TIME table (wide_deal: <= 12 accumulator tiles
// per wave, one code path per wave): with the table in registers hipcc copies the accumulators around every tile's branch and
// waits for lgkmcnt(0) in front of every MFMA block (measured: 64 us, no faster than the blocked form).
// The staging waves load the row keys of every unit unconditionally and BEFORE the values of the stage that uses the keys loaded
// a step earlier: the loads of a step are then a fixed count and the waits are counted (vmcnt(N)); with `if (keyed)` around the
// key loads hipcc waits for vmcnt(0) -- a full memory round trip per 16-row stage, which is what bounds gemm_tn_x6g_kernel
// (removing its MFMAs, splits or loads alone gains 12 / 4 / 7 of 63 us).
#define TXW_NT 12                        // output tiles per multiplying wave
#define TXW_DEPTH 2                      // stages in flight in the staging waves' registers
struct WideDeal {
    int n[4];
    int p[4][TXW_NT + 4], q[4][TXW_NT + 4];
    bool ok;
};
// Each multiplying wave takes a run of the six-product tiles and a run of the three-product tiles (columns that are all
// one-hot: q >= QE), both in q-major order (a wave's tiles share their B fragments); equal shares, the odd tiles of the two
// kinds go to opposite ends.
constexpr WideDeal wide_deal(int PT, int QT, int QE) {
    WideDeal d{};
    int lp[2][64] = {}, lq[2][64] = {}, nk[2] = {0, 0};
    for (int q = 0; q < QT; ++q)
        for (int p = 0; p < PT; ++p) {
            const int k = q >= QE ? 1 : 0;
            lp[k][nk[k]] = p; lq[k][nk[k]] = q; ++nk[k];
        }
    int at[2] = {0, 0};
    d.ok = true;
    for (int w = 0; w < 4; ++w) {
        int n = 0;
        for (int k = 0; k < 2; ++k) {
            const int base = nk[k] / 4, rem = nk[k] % 4;
            const int take = base + ((k == 0 ? w >= 4 - rem : w < rem) ? 1 : 0);
            for (int j = 0; j < take; ++j) {
                if (n < TXW_NT + 4) { d.p[w][n] = lp[k][at[k]]; d.q[w][n] = lq[k][at[k]]; }
                ++at[k]; ++n;
            }
        }
        d.n[w] = n;
        if (n > TXW_NT) d.ok = false;
    }
    return d;
}

template <int CT, int PT, int QT, int QE, int W>
__device__ __forceinline__ void wide_mul(const TnK& kk, const __bf16* Lx, int n_stages, int lane, float* __restrict__ P) {
    constexpr int COLS = CT * 32, PLANE = COLS * TX_PM, BUF = 3 * PLANE, ACOLS = PT * 32;
    constexpr WideDeal D = wide_deal(PT, QT, QE);
    static_assert(D.ok, "more than TXW_NT tiles for one wave");
    constexpr int NW = D.n[W];
    const swr_gemm_tn_args& a = kk.a;
    const int i = lane & 31, s = lane >> 5;
    f32x16 acc[NW > 0 ? NW : 1];
#pragma unroll
    for (int t = 0; t < NW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const __bf16* lane_base = Lx + i * TX_PM + 8 * s;
    __syncthreads();                                                       // stage 0 is in buffer 0
    for (int sg = 0; sg < n_stages; ++sg) {
        const __bf16* buf = lane_base + (sg & 1) * BUF;
        bf16x8 bq[2][3], ap[2][3];
        // fragments of tile 0, then inside the loop those of tile t + 1 before the products of tile t
        if constexpr (NW > 0) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) ap[0][pl] = *reinterpret_cast<const bf16x8*>(buf + (32 * D.p[W][0]) * TX_PM + pl * PLANE);
#pragma unroll
            for (int pl = 0; pl < (D.q[W][0] >= QE ? 1 : 3); ++pl)
                bq[0][pl] = *reinterpret_cast<const bf16x8*>(buf + (ACOLS + 32 * D.q[W][0]) * TX_PM + pl * PLANE);
        }
        int bsel = 0;                                                      // (compile-time after unrolling)
#pragma unroll
        for (int t = 0; t < NW; ++t) {
            const int bcur = bsel;
            if (t + 1 < NW) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    ap[(t + 1) & 1][pl] = *reinterpret_cast<const bf16x8*>(buf + (32 * D.p[W][t + 1]) * TX_PM + pl * PLANE);
                if (D.q[W][t + 1] != D.q[W][t]) {
                    bsel ^= 1;
#pragma unroll
                    for (int pl = 0; pl < (D.q[W][t + 1] >= QE ? 1 : 3); ++pl)
                        bq[bsel][pl] = *reinterpret_cast<const bf16x8*>(buf + (ACOLS + 32 * D.q[W][t + 1]) * TX_PM + pl * PLANE);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 (&af)[3] = ap[t & 1];
            bf16x8 (&b)[3] = bq[bcur];
            f32x16 c_ = acc[t];
            if (D.q[W][t] < QE) {
                if (!SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], b[0], c_, 0, 0, 0);
                if (!SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], b[2], c_, 0, 0, 0);
                if (!SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], b[1], c_, 0, 0, 0);
                c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], b[0], c_, 0, 0, 0);
                c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], b[1], c_, 0, 0, 0);
                c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], b[0], c_, 0, 0, 0);
            } else {
                if (!SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], b[0], c_, 0, 0, 0);
                c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], b[0], c_, 0, 0, 0);
                c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], b[0], c_, 0, 0, 0);
            }
            acc[t] = c_;
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < NW; ++t) {
        const int q = 32 * D.q[W][t] + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p = 32 * D.p[W][t] + (r & 3) + 8 * (r >> 2) + 4 * s;
            if (p < a.K1 && q < a.K2) P[static_cast<int64_t>(p) * kk.k2p + q] = acc[t][r];
        }
    }
}

template <int CT, int PT, int QT, int QE>
__global__ __launch_bounds__(TX_THREADS) void gemm_tn_x6w_kernel(const TnK kk, const TnGather g) {
    constexpr int COLS = CT * 32;
    constexpr int PLANE = COLS * TX_PM, BUF = 3 * PLANE;
    constexpr int ACOLS = PT * 32, UNITS = ACOLS + QT * 32;            // 2 row octs x (columns / 2)
    static_assert(UNITS <= COLS && UNITS <= 512, "slab too narrow / more than two units per staging thread");
    extern __shared__ __attribute__((aligned(16))) __bf16 Lx[];        // [2][3][COLS][TX_PM]
    const swr_gemm_tn_args& a = kk.a;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int split = blockIdx.x;
    const int64_t ms = min(static_cast<int64_t>(split) * kk.rows_per_split, a.M);
    const int64_t me = min(ms + kk.rows_per_split, a.M);
    const int n_stages = static_cast<int>((me - ms + TX_ROWS - 1) / TX_ROWS);
    float* __restrict__ P = kk.part + static_cast<int64_t>(split) * a.K1 * kk.k2p;

    if (wave < 4) {                                                      // multiplying waves: one code path each
        if (wave == 0) wide_mul<CT, PT, QT, QE, 0>(kk, Lx, n_stages, lane, P);
        else if (wave == 1) wide_mul<CT, PT, QT, QE, 1>(kk, Lx, n_stages, lane, P);
        else if (wave == 2) wide_mul<CT, PT, QT, QE, 2>(kk, Lx, n_stages, lane, P);
        else wide_mul<CT, PT, QT, QE, 3>(kk, Lx, n_stages, lane, P);
        return;
    }
    // ====================================================================== staging waves
    const int tid = threadIdx.x - 256;
    struct Unit {
        bool on, col_ok, isA, keyed, is_oh, oh_planes;
        int ro, scol, gcol, bit;
        const float* vbase;
        const uint32_t* kwp;
        uint32_t vstride, kmax;
        float cs0, cs1;
    } un[2];
    // unit 0 of an A column pair when A is recomputed (TnGather.a_z): Z's column pair and the pair's coefficients
    const bool bnA = g.a_z != nullptr;
    const float* zbase = a.A;
    float2 qa = {0.f, 0.f}, qb = {0.f, 0.f}, qc = {0.f, 0.f}, qm = {0.f, 0.f};
    float2 sz[TXW_DEPTH][8];
    typedef uint32_t kw_t[8];
    float2 st[2][TXW_DEPTH][8];
    uint32_t ob[2][TXW_DEPTH];
    kw_t kw[2][2];                                                        // [unit][stage & 1]: keys / mask words, loaded two steps ahead
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        Unit& U = un[u];
        const int uid = tid + 256 * u;
        U.on = uid < UNITS;
        U.ro = uid & 1;
        const int cp = U.on ? (uid >> 1) : 0;
        U.scol = 2 * cp;
        U.isA = U.scol < ACOLS;
        U.gcol = U.isA ? U.scol : U.scol - ACOLS;
        U.col_ok = U.on && (U.isA ? U.gcol < a.K1 : U.gcol < a.K2);
        // (units that load nothing read harmless addresses: A's first row, and A as "keys" -- M * 4 bytes into it and never used)
        U.vbase = a.A; U.vstride = 0; U.kmax = 0; U.kwp = reinterpret_cast<const uint32_t*>(a.A);
        U.keyed = false; U.is_oh = false; U.bit = 0; U.cs0 = 0.f; U.cs1 = 0.f;
        if (U.isA) {
            if (U.col_ok) { U.vbase = a.A + U.gcol; U.vstride = static_cast<uint32_t>(a.lda); }
            if (U.col_ok && bnA && u == 0) {
                zbase = g.a_z + U.gcol;
                qa = *reinterpret_cast<const float2*>(g.a_ca + U.gcol); qb = *reinterpret_cast<const float2*>(g.a_cb + U.gcol);
                qc = *reinterpret_cast<const float2*>(g.a_cc + U.gcol); qm = *reinterpret_cast<const float2*>(g.a_mean + U.gcol);
            }
        } else if (U.col_ok) {
            const TnGatherPiece Pc = g.piece[min(U.gcol >> 3, g.n_pieces - 1)];
            const int e = U.gcol & 7;
            if (Pc.kind == TNG_TABLE) { U.vbase = Pc.vbase + e; U.vstride = Pc.vstride; U.kwp = Pc.kwp; U.kmax = Pc.kmax; U.keyed = true; }
            else if (Pc.kind == TNG_ROWIDX && e < Pc.n_valid) { U.vbase = Pc.vbase + e; U.vstride = Pc.vstride; }
            else if (Pc.kind == TNG_ONEHOT) { U.kwp = Pc.kwp; U.keyed = true; U.is_oh = true; U.bit = Pc.bit0 + e; }
            else U.col_ok = false;
        }
        U.oh_planes = U.is_oh && (U.gcol >> 5) < QE;                     // its tile is multiplied with all three planes
    }
    const uint32_t m_last = static_cast<uint32_t>(a.M - 1);
    const int rows_total = static_cast<int>(me - ms);
    auto kw_load = [&](const Unit& U, int stage, kw_t& dst) {             // unconditional: a fixed number of loads per step
        const uint4* p = reinterpret_cast<const uint4*>(U.kwp + (static_cast<uint32_t>(ms) + 8u * U.ro + static_cast<uint32_t>(stage) * TX_ROWS));
        const uint4 x = p[0], y = p[1];
        dst[0] = x.x; dst[1] = x.y; dst[2] = x.z; dst[3] = x.w; dst[4] = y.x; dst[5] = y.y; dst[6] = y.z; dst[7] = y.w;
    };
    auto val_load = [&](const Unit& U, int stage, const kw_t& k8, float2 (&dst)[8], uint32_t& bits) {
        const uint32_t rbase = static_cast<uint32_t>(ms) + 8u * U.ro + static_cast<uint32_t>(stage) * TX_ROWS;
        uint32_t b2 = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint32_t idx = U.keyed ? min(k8[r], U.kmax) : min(rbase + r, m_last);
            dst[r] = *reinterpret_cast<const float2*>(U.vbase + static_cast<uint64_t>(idx) * U.vstride);
            b2 |= ((k8[r] >> U.bit) & 3u) << (2 * r);
        }
        bits = b2;
    };
    const uint32_t zstride = bnA && un[0].isA && un[0].col_ok ? static_cast<uint32_t>(g.a_ldz) : 0u;
    auto z_load = [&](int stage, float2 (&dst)[8]) {                    // unit 0's rows of Z (a harmless address when A is A)
        const uint32_t rbase = static_cast<uint32_t>(ms) + 8u * un[0].ro + static_cast<uint32_t>(stage) * TX_ROWS;
#pragma unroll
        for (int r = 0; r < 8; ++r) dst[r] = *reinterpret_cast<const float2*>(zbase + static_cast<uint64_t>(min(rbase + r, m_last)) * zstride);
    };
    // dZ of a column pair from dY and Z: the operations of act_bwd_apply_v4_kernel (bn.hip) / fl_dx_kernel in their order
    auto bn_apply = [&](float2 (&raw)[8], const float2 (&z)[8]) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            raw[r].x = fmaf(qb.x, z[r].x - qm.x, raw[r].x * qa.x) + qc.x;
            raw[r].y = fmaf(qb.y, z[r].y - qm.y, raw[r].y * qa.y) + qc.y;
        }
    };
    auto stage_store = [&](Unit& U, int stage, const float2 (&raw)[8], uint32_t bits, __bf16* buf) {
        if (!U.col_ok) return;
        const int left = rows_total - stage * TX_ROWS - 8 * U.ro;
        bf16x8 h0, m0_, l0, h1, m1, l1;
        __bf16* d = buf + U.scol * TX_PM + 8 * U.ro;
        if (U.is_oh) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const bool ok = r < left;
                h0[r] = static_cast<__bf16>((ok && ((bits >> (2 * r)) & 1u)) ? 1.f : 0.f);
                h1[r] = static_cast<__bf16>((ok && ((bits >> (2 * r + 1)) & 1u)) ? 1.f : 0.f);
                m0_[r] = static_cast<__bf16>(0.f); l0[r] = m0_[r]; m1[r] = m0_[r]; l1[r] = m0_[r];
            }
            *reinterpret_cast<bf16x8*>(d) = h0;
            *reinterpret_cast<bf16x8*>(d + TX_PM) = h1;
            if (U.oh_planes) {
                *reinterpret_cast<bf16x8*>(d + PLANE) = m0_;
                *reinterpret_cast<bf16x8*>(d + 2 * PLANE) = l0;
                *reinterpret_cast<bf16x8*>(d + TX_PM + PLANE) = m1;
                *reinterpret_cast<bf16x8*>(d + TX_PM + 2 * PLANE) = l1;
            }
            return;
        }
        if (left >= 8) {
#pragma unroll
            for (int r = 0; r < 8; r += 2) {
                // (scalar splits: the packed-f32 subtractions of SPLIT3_PAIR run on the matrix pipe's side and cost the partner
                // wave's MFMAs more than they save here)
                SPLIT3_INTO(raw[r].x, h0, m0_, l0, r); SPLIT3_INTO(raw[r + 1].x, h0, m0_, l0, r + 1);
                SPLIT3_INTO(raw[r].y, h1, m1, l1, r); SPLIT3_INTO(raw[r + 1].y, h1, m1, l1, r + 1);
                U.cs0 += raw[r].x;
                U.cs1 += raw[r].y;
                U.cs0 += raw[r + 1].x;
                U.cs1 += raw[r + 1].y;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const bool ok = r < left;
                const float vx = ok ? raw[r].x : 0.f, vy = ok ? raw[r].y : 0.f;
                SPLIT3_INTO(vx, h0, m0_, l0, r);
                SPLIT3_INTO(vy, h1, m1, l1, r);
                U.cs0 += vx;
                U.cs1 += vy;
            }
        }
        *reinterpret_cast<bf16x8*>(d) = h0;
        *reinterpret_cast<bf16x8*>(d + PLANE) = m0_;
        *reinterpret_cast<bf16x8*>(d + 2 * PLANE) = l0;
        *reinterpret_cast<bf16x8*>(d + TX_PM) = h1;
        *reinterpret_cast<bf16x8*>(d + TX_PM + PLANE) = m1;
        *reinterpret_cast<bf16x8*>(d + TX_PM + 2 * PLANE) = l1;
    };
    static_assert(TXW_DEPTH == 2 || TXW_DEPTH == 3, "step sequence written for two or three stages in flight");
    // prologue: keys of stages 0 .. DEPTH + 1, values of stages 0 .. DEPTH - 1 in flight, stage 0 into buffer 0, stage DEPTH takes its slot
    {
        kw_t kp[2][TXW_DEPTH];
#pragma unroll
        for (int d = 0; d < TXW_DEPTH; ++d)
#pragma unroll
            for (int u = 0; u < 2; ++u) kw_load(un[u], d, kp[u][d]);
#pragma unroll
        for (int u = 0; u < 2; ++u) { kw_load(un[u], TXW_DEPTH, kw[u][TXW_DEPTH & 1]); kw_load(un[u], TXW_DEPTH + 1, kw[u][(TXW_DEPTH + 1) & 1]); }
#pragma unroll
        for (int d = 0; d < TXW_DEPTH; ++d) {
#pragma unroll
            for (int u = 0; u < 2; ++u) val_load(un[u], d, kp[u][d], st[u][d], ob[u][d]);
            z_load(d, sz[d]);
        }
    }
    const bool applyA = bnA && un[0].isA && un[0].col_ok;
    // slab columns that are never staged (past K1 inside A's tiles, past K2, the tiles past QT): zero in both buffers, once
    for (int c = tid; c < COLS; c += 256) {
        const bool isa = c < ACOLS;
        const int gc = isa ? c : c - ACOLS;
        bool live = c < UNITS && (isa ? gc < a.K1 : gc < a.K2);
        if (live && !isa) {
            const TnGatherPiece Pc = g.piece[min(gc >> 3, g.n_pieces - 1)];
            live = Pc.kind == TNG_TABLE || Pc.kind == TNG_ONEHOT || (Pc.kind == TNG_ROWIDX && (gc & 7) < Pc.n_valid);
        }
        if (!live) {
            bf16x8 z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = static_cast<__bf16>(0.f);
#pragma unroll
            for (int bsel = 0; bsel < 2; ++bsel)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    __bf16* d = Lx + bsel * BUF + pl * PLANE + c * TX_PM;
                    *reinterpret_cast<bf16x8*>(d) = z;
                    *reinterpret_cast<bf16x8*>(d + 8) = z;
                }
        }
    }
    if (applyA) bn_apply(st[0][0], sz[0]);
#pragma unroll
    for (int u = 0; u < 2; ++u) stage_store(un[u], 0, st[u][0], ob[u][0], Lx);
#pragma unroll
    for (int u = 0; u < 2; ++u) val_load(un[u], TXW_DEPTH, kw[u][TXW_DEPTH & 1], st[u][0], ob[u][0]);
    z_load(TXW_DEPTH, sz[0]);
    __syncthreads();
    // step sg: stage sg + 1 (slot (sg + 1) % DEPTH) goes to the other buffer; the keys of stage sg + DEPTH + 2 are requested, THEN
    // the values of stage sg + DEPTH + 1 through the keys requested a step ago
    auto step = [&](int sg, auto slot_c, auto par_c) {
        constexpr int NX = decltype(slot_c)::value;                       // (sg + 1) % DEPTH
        constexpr int KS = decltype(par_c)::value;                        // (sg + DEPTH + 1) & 1
        if (sg + 1 < n_stages) {
            if (applyA) bn_apply(st[0][NX], sz[NX]);
#pragma unroll
            for (int u = 0; u < 2; ++u) stage_store(un[u], sg + 1, st[u][NX], ob[u][NX], Lx + ((sg + 1) & 1) * BUF);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) kw_load(un[u], sg + TXW_DEPTH + 2, kw[u][KS ^ 1]);
#pragma unroll
        for (int u = 0; u < 2; ++u) val_load(un[u], sg + TXW_DEPTH + 1, kw[u][KS], st[u][NX], ob[u][NX]);
        z_load(sg + TXW_DEPTH + 1, sz[NX]);
        __syncthreads();
    };
    int sg = 0;
    // (unrolled by 6 = lcm(value slots, key slots); the key slot is the parity of stage sg + DEPTH + 1)
#define TXW_STEP(OFF) step(sg + OFF, std::integral_constant<int, (OFF + 1) % TXW_DEPTH>{}, std::integral_constant<int, (OFF + TXW_DEPTH + 1) & 1>{})
    for (; sg + 6 <= n_stages; sg += 6) { TXW_STEP(0); TXW_STEP(1); TXW_STEP(2); TXW_STEP(3); TXW_STEP(4); TXW_STEP(5); }
    if (sg < n_stages) { TXW_STEP(0); }
    if (sg + 1 < n_stages) { TXW_STEP(1); }
    if (sg + 2 < n_stages) { TXW_STEP(2); }
    if (sg + 3 < n_stages) { TXW_STEP(3); }
    if (sg + 4 < n_stages) { TXW_STEP(4); }
#undef TXW_STEP
    if (kk.part_cs) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float c0 = un[u].cs0, c1 = un[u].cs1;
            c0 += __shfl_xor(c0, 1);
            c1 += __shfl_xor(c1, 1);
            if (un[u].col_ok && un[u].isA && un[u].ro == 0) {
                kk.part_cs[static_cast<int64_t>(split) * a.K1 + un[u].gcol] = c0;
                kk.part_cs[static_cast<int64_t>(split) * a.K1 + un[u].gcol + 1] = c1;
            }
        }
    }
}

// fixed-order sum of the per-workgroup partial tiles.  A workgroup owns 256 / L consecutive output elements; thread
// (sub, o) adds partials sub, sub + L, ... of element o in order (UNROLL loads in flight), so the 64 lanes of a wave read
// 64 CONSECUTIVE floats of one partial matrix (coalesced; with the L lanes of an element adjacent, as before, every lane
// of a wave touched another partial matrix: 19 us for 24 MB at config 2); the L per-thread sums of an element are then
// added in sub order through LDS -> deterministic.  L grows with the number of partials so that the per-thread chain
// stays a few loads long (tower layers have hundreds of partials).
template <int L>
__global__ __launch_bounds__(256) void tn_reduce_kernel(const TnK kk) {
    constexpr int OUTS = 256 / L;                     // output elements per workgroup
    __shared__ float red[256];
    const swr_gemm_tn_args& a = kk.a;
    const int g = blockIdx.y;
    const int64_t n = static_cast<int64_t>(a.K1) * a.K2;
    const int64_t np = static_cast<int64_t>(a.K1) * kk.k2p;           // elements of one partial matrix (pitch k2p >= K2)
    const int sub = threadIdx.x / OUTS, o = threadIdx.x % OUTS;
    const int64_t j = static_cast<int64_t>(blockIdx.x) * OUTS + o;
    const int nparts = (kk.splits + GEMM_WAVES - 1) / GEMM_WAVES;
    auto lane_sum = [&](const float* __restrict__ p, int64_t stride) {
        float sum = 0.f;
        int sp = sub;
        for (; sp + 3 * L < nparts; sp += 4 * L) {
            const float v0 = p[sp * stride], v1 = p[(sp + L) * stride], v2 = p[(sp + 2 * L) * stride],
                        v3 = p[(sp + 3 * L) * stride];
            sum += v0; sum += v1; sum += v2; sum += v3;
        }
        for (; sp < nparts; sp += L) sum += p[sp * stride];
        return sum;
    };
    float sum = 0.f;
    if (j < n) {
        const int64_t r = j / a.K2, c = j - r * a.K2;
        const float* __restrict__ pj = kk.part + static_cast<int64_t>(g) * nparts * np + r * kk.k2p + c;
        sum = lane_sum(pj, np);
        if (kk.tail_rep > 0 && c >= kk.tail_q0)                          // the other replicas of the tail tile, in order
            for (int b = 1; b < kk.tail_rep; ++b) sum += lane_sum(pj + 32 * b, np);
    }
    red[threadIdx.x] = sum;
    __syncthreads();
    if (sub == 0 && j < n) {
        float tot = red[o];
#pragma unroll
        for (int q = 1; q < L; ++q) tot += red[q * OUTS + o];
        const int64_t r = j / a.K2, c = j - r * a.K2;
        if (a.C2 && c >= a.c2_from) {
            a.C2[r * a.ldc2 + (c - a.c2_from)] = tot;                    // second destination: always overwritten
        } else {
            float* dst = a.C + g * a.gsC + r * a.ldc + c;
            *dst = a.accumulate ? *dst + tot : tot;
        }
    }
    __syncthreads();
    float cs = 0.f;
    if (kk.part_cs && j < a.K1) cs = lane_sum(kk.part_cs + static_cast<int64_t>(g) * nparts * a.K1 + j, a.K1);
    red[threadIdx.x] = cs;
    __syncthreads();
    if (kk.part_cs && sub == 0 && j < a.K1) {
        float tot = red[o];
#pragma unroll
        for (int q = 1; q < L; ++q) tot += red[q * OUTS + o];
        float* dst = a.colsum + g * a.gsColsum + j;
        *dst = a.accumulate ? *dst + tot : tot;
    }
}

// the same fixed-order sum for MANY partial matrices (the wide kernel writes one per CU: 256 x 184 KB at config 2): a thread
// owns four consecutive outputs (16-byte loads), 16 threads share them and take partials sub, sub + 16, ... (8 loads in
// flight), their sums are added in sub order through LDS.  tn_reduce_kernel<32> reads 32-byte runs: 27 us for the 47 MB.
#ifndef TN_RED4_SUBS
#define TN_RED4_SUBS 16
#endif
__global__ __launch_bounds__(256) void tn_reduce4_kernel(const TnK kk) {
    constexpr int SUBS = TN_RED4_SUBS, OUT4 = 256 / SUBS;      // 16 float4 columns = 64 outputs per workgroup
    __shared__ float4 red[256];
    const swr_gemm_tn_args& a = kk.a;
    const int64_t n4 = static_cast<int64_t>(a.K1) * a.K2 / 4;
    const int64_t np = static_cast<int64_t>(a.K1) * kk.k2p;
    const int sub = threadIdx.x / OUT4, o = threadIdx.x % OUT4;
    const int64_t j4 = static_cast<int64_t>(blockIdx.x) * OUT4 + o;
    const int nparts = (kk.splits + GEMM_WAVES - 1) / GEMM_WAVES;
    auto lane_sum = [&](const float* __restrict__ p, int64_t stride) {
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        int sp = sub;
        for (; sp + 7 * SUBS < nparts; sp += 8 * SUBS) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (sp + u * SUBS) * stride);
#pragma unroll
            for (int u = 0; u < 8; ++u) { sum.x += v[u].x; sum.y += v[u].y; sum.z += v[u].z; sum.w += v[u].w; }
        }
        for (; sp < nparts; sp += SUBS) {
            const float4 v = *reinterpret_cast<const float4*>(p + sp * stride);
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
        return sum;
    };
    auto total = [&]() {
        float4 t = red[o];
#pragma unroll
        for (int q = 1; q < SUBS; ++q) { const float4 v = red[q * OUT4 + o]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        return t;
    };
    const int64_t j = 4 * j4;
    const int64_t r = j / a.K2, c = j - r * a.K2;                          // (K2 % 4 == 0: the four outputs share a row)
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j4 < n4) sum = lane_sum(kk.part + r * kk.k2p + c, np);
    red[threadIdx.x] = sum;
    __syncthreads();
    if (sub == 0 && j4 < n4) {
        const float4 t = total();
        const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (a.C2 && c + e >= a.c2_from) {
                a.C2[r * a.ldc2 + (c + e - a.c2_from)] = tv[e];
            } else {
                float* dst = a.C + r * a.ldc + c + e;
                *dst = a.accumulate ? *dst + tv[e] : tv[e];
            }
        }
    }
    __syncthreads();
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool cs_on = kk.part_cs && j < a.K1;                             // (K1 % 4 == 0)
    if (cs_on) cs = lane_sum(kk.part_cs + j, a.K1);
    red[threadIdx.x] = cs;
    __syncthreads();
    if (cs_on && sub == 0) {
        const float4 t = total();
        const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float* dst = a.colsum + j + e;
            *dst = a.accumulate ? *dst + tv[e] : tv[e];
        }
    }
}
static bool tn_reduce4_ok(const TnK& kk) {
    const swr_gemm_tn_args& a = kk.a;
    return a.groups == 1 && kk.tail_rep == 0 && a.K2 % 4 == 0 && kk.k2p % 4 == 0 && a.K1 % 4 == 0 &&
           (reinterpret_cast<uintptr_t>(kk.part) & 15u) == 0 && (!kk.part_cs || (reinterpret_cast<uintptr_t>(kk.part_cs) & 15u) == 0);
}
static void tn_reduce4(const TnK& kk, hipStream_t st) {
    const int64_t n4 = static_cast<int64_t>(kk.a.K1) * kk.a.K2 / 4;
    hipLaunchKernelGGL(tn_reduce4_kernel, dim3(static_cast<unsigned>(swr_ceil_div(n4, 256 / TN_RED4_SUBS))), dim3(256), 0, st, kk);
}

static int tn_plan(const swr_gemm_tn_args& a, int& ta, int& splits, int64_t& rps) {
    const int tiles1 = static_cast<int>(swr_ceil_div(a.K1, 32));
    const int pblk = static_cast<int>(swr_ceil_div(tiles1, TN_TA_MAX));
    ta = static_cast<int>(swr_ceil_div(tiles1, pblk));
    const int qblk = static_cast<int>(swr_ceil_div(a.K2, 32 * TN_TB));
    const int64_t tiles = static_cast<int64_t>(pblk) * qblk * a.groups;
    constexpr int waves_target = 1024;
    int64_t want = std::max<int64_t>(1, waves_target / tiles);         // one wave per SIMD over the chip: measured 145 us vs 157 (2 per SIMD) and
                                                                       // 174 (4): more resident waves only thrash the dword-load path
    want = std::min<int64_t>(want, std::max<int64_t>(1, a.M / 64));    // at least 64 rows per wave
    rps = swr_ceil_div(a.M, want);
    rps += rps & 1;
    splits = static_cast<int>(swr_ceil_div(a.M, rps));
    return pblk;
}

// bf16-split tn kernel: one group, A narrow enough to stage whole (<= 5 column tiles), 16-byte rows, a batch worth it
static bool tn_x6_ok(const swr_gemm_tn_args& a) {
    return use_x6() && a.groups == 1 && a.K1 <= 32 * TN_TA_MAX * 32 && a.K1 % 2 == 0 && a.K2 % 2 == 0 && a.lda % 2 == 0 &&
           a.ldb % 2 == 0 && (reinterpret_cast<uintptr_t>(a.A) & 7u) == 0 && (reinterpret_cast<uintptr_t>(a.B) & 7u) == 0 && a.M >= 4096 &&
           a.M * a.lda < (1ll << 31) && a.M * a.ldb < (1ll << 31);      // 32-bit element offsets inside the kernel
}
// ragged last column tile folded into the full 128-column blocks (see TnK): K2 = 128 j + r with j >= 1, 0 < r <= 32
static bool tn_x6_tail(const swr_gemm_tn_args& a) {
    const int r = a.K2 % TX_QCOLS;
    return a.K2 > TX_QCOLS && r > 0 && r <= 32;
}
// the wide form (gemm_tn_x6w_kernel): every column tile of B in one workgroup's slab
static bool tn_wide_shape(int pt, int qt);
static bool tn_wide_ok(const swr_gemm_tn_args& a) {
    const int pt = static_cast<int>(swr_ceil_div(a.K1, 32)), qt = static_cast<int>(swr_ceil_div(a.K2, 32));
    return !use_bf16() && tn_wide_enabled() && tn_wide_shape(pt, qt) && a.K2 <= 8 * TNG_MAX_PIECES && a.M >= 256 * TX_ROWS;
}
// rows per batch split below which the wide / transpose-read products are not cut (SWR_TN_MIN_ROWS, a multiple of 32; read per call).
// 256 rows (8 stages of 32) amortise a split's two bursts -- operands of its first stages in, its partial tile out -- at long batches;
// at a short batch (the 8 192-row strong-scaling shard) that leaves 32 workgroups on 256 CUs, and the product is all latency: fewer
// rows per split, more CUs
static int64_t tn_min_rows() {
    const char* e = getenv("SWR_TN_MIN_ROWS");
    const int64_t v = e ? atoll(e) : 0;
    return (v >= 32 && v % 32 == 0) ? v : 64;
}
static void tn_x6_plan(const swr_gemm_tn_args& a, int& n_splits, int64_t& rps) {
    constexpr int blocks_target = 256;   // one per CU
    const int qblk = tn_x6_tail(a) ? a.K2 / TX_QCOLS : static_cast<int>(swr_ceil_div(a.K2, TX_QCOLS));
    const int pblk = static_cast<int>(swr_ceil_div(swr_ceil_div(a.K1, 32), TN_TA_MAX));
    int64_t want = tn_wide_ok(a) ? blocks_target : std::max<int64_t>(1, blocks_target / (qblk * pblk));
    want = std::min<int64_t>(want, std::max<int64_t>(1, a.M / (tn_wide_ok(a) ? tn_min_rows() : 8 * TX_ROWS)));
    rps = swr_ceil_div(swr_ceil_div(a.M, want), TX_ROWS) * TX_ROWS;
    n_splits = static_cast<int>(swr_ceil_div(a.M, rps));
}

// the instantiated shapes of the wide kernel: (column tiles of A, of A', first all-one-hot tile).  A gathered product whose
// one-hot columns start later than QE takes the form with the next smaller QE (more six-product tiles than needed: exact).
#define TNW_SHAPES(X) X(14, 5, 9, 5) X(14, 5, 9, 9)
static const void* tn_wide_fn(int pt, int qt, int qe_have, unsigned& lds) {
    const void* best = nullptr;
    int best_qe = -1;
#define X(CT, PT, QT, QE)                                                                                                  \
    if (pt == PT && qt == QT && QE <= qe_have && QE > best_qe) {                                                           \
        best = reinterpret_cast<const void*>(gemm_tn_x6w_kernel<CT, PT, QT, QE>); best_qe = QE;                            \
        lds = static_cast<unsigned>(2 * 3 * CT * 32 * TX_PM * sizeof(__bf16));                                             \
    }
    TNW_SHAPES(X)
#undef X
    return best;
}
static bool tn_wide_shape(int pt, int qt) {
#define X(CT, PT, QT, QE) if (pt == PT && qt == QT) return true;
    TNW_SHAPES(X)
#undef X
    return false;
}

static int tn_wide_launch(const TnK& kk, const TnGather& g, int n_splits, hipStream_t st) {
    unsigned lds = 0;
    const int qt = static_cast<int>(swr_ceil_div(kk.a.K2, 32));
    const void* fn = tn_wide_fn(static_cast<int>(swr_ceil_div(kk.a.K1, 32)), qt, std::min(qt, g.kp / 32), lds);
    if (!fn) return SWR_ERR_UNSUPPORTED;
    if (lds > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 1024) != hipSuccess) return SWR_ERR_LAUNCH;
    TnK k2 = kk;
    TnGather gg = g;
    void* kargs[] = {&k2, &gg};
    if (hipLaunchKernel(fn, dim3(static_cast<unsigned>(n_splits)), dim3(TX_THREADS), kargs, lds, st) != hipSuccess) return SWR_ERR_LAUNCH;
    return SWR_OK;
}

extern "C" size_t swr_gemm_tn_workspace_bytes(const swr_gemm_tn_args* args);
// the gathered form: same plan, same partial layout, same fixed-order reduction as the x6 branch of swr_gemm_tn
bool tn_x6_gather_ok(const swr_gemm_tn_args& a) {
    return use_x6() && !use_bf16() && a.groups == 1 && a.K1 <= 32 * TN_TA_MAX && a.K1 % 2 == 0 && a.K2 % 2 == 0 && a.lda % 2 == 0 &&
           (reinterpret_cast<uintptr_t>(a.A) & 7u) == 0 && a.M >= 4096 && a.M * a.lda < (1ll << 31) && !a.C2 && a.K2 <= 8 * TNG_MAX_PIECES;
}

bool tn_x6_gather_wide(const swr_gemm_tn_args& a, int kp) {
    unsigned lds = 0;
    const int qt = static_cast<int>(swr_ceil_div(a.K2, 32));
    return tn_x6_gather_ok(a) && tn_wide_ok(a) && tn_wide_fn(static_cast<int>(swr_ceil_div(a.K1, 32)), qt, std::min(qt, kp / 32), lds) != nullptr;
}

int tn_x6_gather(const swr_gemm_tn_args& a, const TnGather& g, void* workspace, size_t workspace_bytes, void* stream) {
    SWR_REQUIRE(tn_x6_gather_ok(a) && a.A && a.C && a.lda >= a.K1 && a.ldc >= a.K2 && g.n_pieces * 8 >= a.K2 && g.kp % 16 == 0, SWR_ERR_ARG);
    SWR_REQUIRE(g.a_z == nullptr || (tn_x6_gather_wide(a, g.kp) && g.a_ca && g.a_cb && g.a_cc && g.a_mean && g.a_ldz >= a.K1 && g.a_ldz % 2 == 0 &&
                                     (reinterpret_cast<uintptr_t>(g.a_z) & 7u) == 0 && a.M * g.a_ldz < (1ll << 31)), SWR_ERR_UNSUPPORTED);
    swr_gemm_tn_args sized = a;
    sized.B = a.A; sized.ldb = a.K2;                               // (workspace size: B is only looked at for alignment)
    const size_t need = swr_gemm_tn_workspace_bytes(&sized);
    SWR_REQUIRE(workspace != nullptr && workspace_bytes >= need, SWR_ERR_WORKSPACE);
    hipStream_t st = static_cast<hipStream_t>(stream);
    TnK kk;
    kk.a = a;
    int n_splits;
    tn_x6_plan(a, n_splits, kk.rows_per_split);
    kk.splits = n_splits * GEMM_WAVES;
    const bool wide = tn_wide_ok(a);
    // transpose-read form (dw_tr.hip): A recomputed from dY and Z, A' through the pre-split pieces; splits of a multiple of 32 rows
    const bool tr = wide && g.a_z && g.tr_voff && dw_tr_shape_ok(a.K1, a.K2, g.tr_nr) && a.M < (1ll << 26) && a.M * a.lda * 4 < (1ll << 32) &&
                    a.M * g.a_ldz * 4 < (1ll << 32) && a.lda % 4 == 0 && g.a_ldz % 4 == 0 && (reinterpret_cast<uintptr_t>(a.A) & 15u) == 0 &&
                    (reinterpret_cast<uintptr_t>(g.a_z) & 15u) == 0;
    const bool tail = !wide && tn_x6_tail(a);
    kk.qblk = tail ? static_cast<unsigned>(a.K2 / TX_QCOLS) : static_cast<unsigned>(swr_ceil_div(a.K2, TX_QCOLS));
    kk.tail_q0 = tail ? static_cast<int>(kk.qblk) * TX_QCOLS : 0;
    kk.tail_rep = tail ? static_cast<int>(kk.qblk) : 0;
    kk.k2p = tail ? static_cast<int>(kk.qblk) * (TX_QCOLS + 32) : a.K2;
    kk.part = static_cast<float*>(workspace);
    kk.part_cs = a.colsum ? kk.part + static_cast<size_t>(n_splits) * a.K1 * kk.k2p : nullptr;
    const int pt_all = static_cast<int>(swr_ceil_div(a.K1, 32));
    kk.pblk = 1;
    kk.n_tiles = kk.qblk * static_cast<unsigned>(n_splits);
    if (tr) {
        DwTrArgs t;
        t.dY = reinterpret_cast<const char*>(a.A); t.Z = reinterpret_cast<const char*>(g.a_z);
        t.lddy_b = static_cast<uint32_t>(a.lda * 4); t.ldz_b = static_cast<uint32_t>(g.a_ldz * 4);
        t.ca = g.a_ca; t.cb = g.a_cb; t.cc = g.a_cc; t.mean = g.a_mean;
        t.M = a.M; t.K1 = a.K1; t.K2 = a.K2;
        t.ws = g.tr_ws; t.voff = g.tr_voff; t.mask_t = g.tr_mask_t; t.NR = g.tr_nr;
        t.part = kk.part; t.part_cs = kk.part_cs; t.k2p = kk.k2p;
        t.rows_per_split = std::max<int64_t>(std::min<int64_t>(128, tn_min_rows()), (kk.rows_per_split + 31) / 32 * 32);   // (never more splits than planned)
        {
            static int64_t force = -1;              // SWR_DW_TR_ROWS: rows per split (experiments: fewer, longer splits)
            if (force < 0) { const char* e = getenv("SWR_DW_TR_ROWS"); force = e ? atoll(e) : 0; }
            if (force >= t.rows_per_split && force % 32 == 0) t.rows_per_split = force;
        }
        t.n_splits = static_cast<int>(swr_ceil_div(a.M, t.rows_per_split));
        const int rc = dw_tr_launch(t, st);
        if (rc != SWR_OK) return rc;
        kk.rows_per_split = t.rows_per_split;
        kk.splits = t.n_splits * GEMM_WAVES;
    } else if (wide) {
        const int rc = tn_wide_launch(kk, g, n_splits, st);
        if (rc != SWR_OK) return rc;
    } else {
    const dim3 grid((kk.n_tiles + 7) / 8 * 8);
    const unsigned lds = static_cast<unsigned>(2 * 3 * (pt_all * 32 + TX_QCOLS + (tail ? 32 : 0)) * TX_PM * sizeof(__bf16));
    const void* fn = nullptr;
#define TNG(PTV) (tail ? reinterpret_cast<const void*>(gemm_tn_x6g_kernel<PTV, true>) : reinterpret_cast<const void*>(gemm_tn_x6g_kernel<PTV, false>))
    switch (pt_all) {
        case 1: fn = TNG(1); break;
        case 2: fn = TNG(2); break;
        case 3: fn = TNG(3); break;
        case 4: fn = TNG(4); break;
        default: fn = TNG(5); break;
    }
#undef TNG
    if (lds > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024) != hipSuccess)
        return SWR_ERR_LAUNCH;
    TnGather gg = g;
    void* kargs[] = {&kk, &gg};
    if (hipLaunchKernel(fn, grid, dim3(TX_THREADS), kargs, lds, st) != hipSuccess) return SWR_ERR_LAUNCH;
    }
    const int64_t n = static_cast<int64_t>(a.K1) * a.K2;
    const dim3 rblock(256);
#define TN_RED(LV)                                                                                                      \
    hipLaunchKernelGGL(tn_reduce_kernel<LV>, dim3(static_cast<unsigned>(swr_ceil_div(n * LV, 256)), 1u), rblock, 0, st, kk)
    if (wide && tn_reduce4_ok(kk)) tn_reduce4(kk, st);
    else if (n_splits > 128) TN_RED(32);
    else if (n_splits > 64) TN_RED(8);
    else TN_RED(4);
#undef TN_RED
    return swr_launch_status();
}

// a grouped product whose groups each go through the bf16-split kernel (one launch per group, swr_gemm_tn)
static bool tn_groups_as_x6(const swr_gemm_tn_args& a) {
    if (a.groups <= 1 || a.C2 || !use_x6() || use_bf16()) return false;
    swr_gemm_tn_args one = a;
    one.groups = 1;
    // measured at config 5 (8 groups of [32 768, 128]^T x [32 768, 256]): 144 us + a 40 us reduction per group against 286 us for the
    // grouped f32-MFMA launch -- the blocked bf16-split kernel is built for ONE wide product; off unless SWR_TN_GROUPS_X6=1
    static int on = -1;
    if (on < 0) { const char* e = getenv("SWR_TN_GROUPS_X6"); on = (e && e[0] == '1') ? 1 : 0; }
    if (!on || !tn_x6_ok(one) || 2.0 * a.M * a.K1 * a.K2 < 1.0e9) return false;
    for (int g = 1; g < a.groups; ++g)
        if (((reinterpret_cast<uintptr_t>(a.A + g * a.gsA) | reinterpret_cast<uintptr_t>(a.B + g * a.gsB)) & 7u) != 0) return false;
    return true;
}

extern "C" size_t swr_gemm_tn_workspace_bytes(const swr_gemm_tn_args* args) {
    if (!args || args->M <= 0 || args->K1 <= 0 || args->K2 <= 0 || args->groups < 1) return 0;
    if (tn_groups_as_x6(*args)) {
        swr_gemm_tn_args one = *args;
        one.groups = 1;
        return swr_gemm_tn_workspace_bytes(&one);
    }
    int ta, splits;
    int64_t rps;
    if (tn_x6_ok(*args))
        tn_x6_plan(*args, splits, rps);
    else
        tn_plan(*args, ta, splits, rps);
    size_t k2p = args->K2;
    if (tn_x6_ok(*args) && tn_x6_tail(*args) && !tn_wide_ok(*args)) k2p = static_cast<size_t>(args->K2 / TX_QCOLS) * (TX_QCOLS + 32);
    return static_cast<size_t>(args->groups) * splits * (static_cast<size_t>(args->K1) * k2p + args->K1) * 4 + 256;
}

extern "C" int swr_gemm_tn(const swr_gemm_tn_args* args, void* workspace, size_t workspace_bytes, void* stream) {
    SWR_REQUIRE(args != nullptr, SWR_ERR_ARG);
    const swr_gemm_tn_args& a = *args;
    SWR_REQUIRE(a.M >= 0 && a.K1 > 0 && a.K2 > 0 && a.A && a.B && a.C && a.groups >= 1, SWR_ERR_ARG);
    SWR_REQUIRE(a.lda >= a.K1 && a.ldb >= a.K2, SWR_ERR_ARG);
    if (a.C2) SWR_REQUIRE(a.groups == 1 && a.c2_from > 0 && a.c2_from < a.K2 && a.ldc >= a.c2_from && a.ldc2 >= a.K2 - a.c2_from, SWR_ERR_ARG);
    else SWR_REQUIRE(a.ldc >= a.K2, SWR_ERR_ARG);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (tn_groups_as_x6(a)) {
        // grouped layers (the D per-domain layers of a LayerBank): the bf16-split kernel takes one group per launch -- 2.7 x the
        // matrix rate of the grouped f32-MFMA kernel where a group is worth a launch of its own; same workspace, stream-ordered
        swr_gemm_tn_args one = a;
        one.groups = 1; one.gsA = one.gsB = one.gsC = one.gsColsum = 0;
        for (int g = 0; g < a.groups; ++g) {
            one.A = a.A + g * a.gsA; one.B = a.B + g * a.gsB; one.C = a.C + g * a.gsC;
            one.colsum = a.colsum ? a.colsum + g * a.gsColsum : nullptr;
            const int rc = swr_gemm_tn(&one, workspace, workspace_bytes, stream);
            if (rc != SWR_OK) return rc;
        }
        return SWR_OK;
    }
    TnK kk;
    kk.a = a;
    if (a.M == 0 && a.C2) {
        for (int r = 0; r < a.K1; ++r)
            if (hipMemsetAsync(a.C2 + static_cast<int64_t>(r) * a.ldc2, 0, sizeof(float) * (a.K2 - a.c2_from), st) != hipSuccess)
                return SWR_ERR_LAUNCH;
    }
    if (a.M == 0) {
        // empty batch: the gradient is zero
        if (a.accumulate) return SWR_OK;
        for (int g = 0; g < a.groups; ++g) {
            for (int r = 0; r < a.K1; ++r)
                if (hipMemsetAsync(a.C + g * a.gsC + static_cast<int64_t>(r) * a.ldc, 0, sizeof(float) * (a.C2 ? a.c2_from : a.K2), st) != hipSuccess)
                    return SWR_ERR_LAUNCH;
            if (a.colsum && hipMemsetAsync(a.colsum + g * a.gsColsum, 0, sizeof(float) * a.K1, st) != hipSuccess)
                return SWR_ERR_LAUNCH;
        }
        return SWR_OK;
    }
    const size_t need = swr_gemm_tn_workspace_bytes(args);
    SWR_REQUIRE(workspace != nullptr && workspace_bytes >= need, SWR_ERR_WORKSPACE);
    if (tn_x6_ok(a)) {
        int n_splits;
        tn_x6_plan(a, n_splits, kk.rows_per_split);
        kk.splits = n_splits * GEMM_WAVES;                      // tn_reduce_kernel counts partial tiles as splits / 4
        const bool wide = tn_wide_ok(a);
        const bool tail = !wide && tn_x6_tail(a);
        kk.qblk = tail ? static_cast<unsigned>(a.K2 / TX_QCOLS) : static_cast<unsigned>(swr_ceil_div(a.K2, TX_QCOLS));
        kk.tail_q0 = tail ? static_cast<int>(kk.qblk) * TX_QCOLS : 0;
        kk.tail_rep = tail ? static_cast<int>(kk.qblk) : 0;
        kk.k2p = tail ? static_cast<int>(kk.qblk) * (TX_QCOLS + 32) : a.K2;
        kk.part = static_cast<float*>(workspace);
        kk.part_cs = a.colsum ? kk.part + static_cast<size_t>(n_splits) * a.K1 * kk.k2p : nullptr;
        const int pt_all = static_cast<int>(swr_ceil_div(a.K1, 32));
        kk.pblk = static_cast<unsigned>(swr_ceil_div(pt_all, TN_TA_MAX));          // A column blocks of <= 5 tiles
        kk.n_tiles = kk.qblk * kk.pblk * static_cast<unsigned>(n_splits);
        const int pt = static_cast<int>(swr_ceil_div(pt_all, kk.pblk));
        if (wide) {
            // B as plain 8-column pieces of the gathered form: the same kernel, plan and summation order as the fused lookup's
            // product (tests/test_ops_gpu.py compares the two bit for bit)
            TnGather g;
            g.a_z = nullptr; g.a_ldz = 0; g.a_ca = g.a_cb = g.a_cc = g.a_mean = nullptr;
            g.n_pieces = static_cast<int>(swr_ceil_div(a.K2, 8));
            g.kp = 32 * 16;                                           // no one-hot columns
            for (int c = 0; c < g.n_pieces; ++c) {
                TnGatherPiece& P = g.piece[c];
                P.vbase = a.B + 8 * c; P.kwp = nullptr; P.vstride = static_cast<uint32_t>(a.ldb); P.kmax = 0;
                P.kind = TNG_ROWIDX; P.bit0 = 0; P.n_valid = static_cast<int16_t>(std::min<int64_t>(8, a.K2 - 8 * c)); P.pad = 0;
            }
            const int rc = tn_wide_launch(kk, g, n_splits, st);
            if (rc != SWR_OK) return rc;
        } else {
        const dim3 grid((kk.n_tiles + 7) / 8 * 8);
        const unsigned lds = static_cast<unsigned>(2 * 3 * (pt * 32 + TX_QCOLS + (tail ? 32 : 0)) * TX_PM * sizeof(__bf16));
        const void* fn = nullptr;
#define TNX(PTV) (use_bf16() ? (tail ? reinterpret_cast<const void*>(gemm_tn_x6_kernel<PTV, true, true>) : reinterpret_cast<const void*>(gemm_tn_x6_kernel<PTV, false, true>)) \
                             : (tail ? reinterpret_cast<const void*>(gemm_tn_x6_kernel<PTV, true, false>) : reinterpret_cast<const void*>(gemm_tn_x6_kernel<PTV, false, false>)))
        switch (pt) {
            case 1: fn = TNX(1); break;
            case 2: fn = TNX(2); break;
            case 3: fn = TNX(3); break;
            case 4: fn = TNX(4); break;
            default: fn = TNX(5); break;
        }
#undef TNX
        // > 64 KB of dynamic LDS needs the attribute (idempotent, not a stream operation; first call = a warm-up step)
        if (lds > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024) != hipSuccess)
            return SWR_ERR_LAUNCH;
        void* kargs[] = {&kk};
        if (hipLaunchKernel(fn, grid, dim3(TX_THREADS), kargs, lds, st) != hipSuccess) return SWR_ERR_LAUNCH;
        }
        const int64_t n = static_cast<int64_t>(a.K1) * a.K2;
        const dim3 rblock(256);
#define TN_RED(LV)                                                                                                      \
    hipLaunchKernelGGL(tn_reduce_kernel<LV>, dim3(static_cast<unsigned>(swr_ceil_div(n * LV, 256)), 1u), rblock, 0, st, kk)
        if (wide && tn_reduce4_ok(kk)) tn_reduce4(kk, st);
        else if (n_splits > 128) TN_RED(32);
        else if (n_splits > 64) TN_RED(8);
        else TN_RED(4);
#undef TN_RED
        return swr_launch_status();
    }
    int ta;
    const int pblk = tn_plan(a, ta, kk.splits, kk.rows_per_split);
    kk.tail_q0 = kk.tail_rep = 0;
    kk.k2p = a.K2;
    kk.part = static_cast<float*>(workspace);
    kk.part_cs = a.colsum ? kk.part + static_cast<size_t>(a.groups) * kk.splits * a.K1 * a.K2 : nullptr;
    const int zsplit = (kk.splits + GEMM_WAVES - 1) / GEMM_WAVES;
    kk.qblk = static_cast<unsigned>(swr_ceil_div(a.K2, 32 * TN_TB));
    kk.pblk = static_cast<unsigned>(pblk);
    kk.n_tiles = kk.qblk * kk.pblk * static_cast<unsigned>(zsplit * a.groups);
    const dim3 grid((kk.n_tiles + 7) / 8 * 8);
#define TN_LDS(TAV) static_cast<unsigned>(((TAV) * TN_TB * 16 * 64 + 4 * (TAV) * 64) * sizeof(float))
    switch (ta) {
        case 1: hipLaunchKernelGGL(gemm_tn_kernel<1>, grid, dim3(GEMM_THREADS), TN_LDS(1), st, kk); break;
        case 2: hipLaunchKernelGGL(gemm_tn_kernel<2>, grid, dim3(GEMM_THREADS), TN_LDS(2), st, kk); break;
        case 3: hipLaunchKernelGGL(gemm_tn_kernel<3>, grid, dim3(GEMM_THREADS), TN_LDS(3), st, kk); break;
        case 4: hipLaunchKernelGGL(gemm_tn_kernel<4>, grid, dim3(GEMM_THREADS), TN_LDS(4), st, kk); break;
        default: hipLaunchKernelGGL(gemm_tn_kernel<5>, grid, dim3(GEMM_THREADS), TN_LDS(5), st, kk); break;
    }
#undef TN_LDS
    const int64_t n = static_cast<int64_t>(a.K1) * a.K2;
    const dim3 rblock(256);
#define TN_RED(LV)                                                                                                      \
    hipLaunchKernelGGL(tn_reduce_kernel<LV>, dim3(static_cast<unsigned>(swr_ceil_div(n * LV, 256)), static_cast<unsigned>(a.groups)), \
                       rblock, 0, st, kk)
    if (zsplit > 64) TN_RED(32);
    else if (zsplit > 16) TN_RED(8);
    else TN_RED(4);
#undef TN_RED
    return swr_launch_status();
}
