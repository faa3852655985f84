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

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

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

    for (int kb = 0; kb < K; kb += 8) {
        const int k = kb + 4 * s;
        float4 av = load_k4<VEC>(arow, k, K);
        if (PRO) {
            // previous layer's BatchNorm (+ReLU) applied while the operand is loaded
            const float4 sc = load_k4<VEC>(psc, k, K), sh = load_k4<VEC>(psh, k, K);
            av.x = fmaf(av.x, sc.x, sh.x); av.y = fmaf(av.y, sc.y, sh.y);
            av.z = fmaf(av.z, sc.z, sh.z); av.w = fmaf(av.w, sc.w, sh.w);
            if (a.a_relu) {
                av.x = fmaxf(av.x, 0.f); av.y = fmaxf(av.y, 0.f); av.z = fmaxf(av.z, 0.f); av.w = fmaxf(av.w, 0.f);
            }
            if (k >= K) av.x = 0.f;
            if (k + 1 >= K) av.y = 0.f;
            if (k + 2 >= K) av.z = 0.f;
            if (k + 3 >= K) av.w = 0.f;
        }
        float4 bv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (BT) {
                bv[t] = load_k4<VEC>(brow[t], k, K);
                if (!bvalid[t]) bv[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                const float* p = brow[t] + static_cast<int64_t>(k) * a.ldb;
                bv[t].x = (bvalid[t] && k < K) ? p[0] : 0.f;
                bv[t].y = (bvalid[t] && k + 1 < K) ? p[a.ldb] : 0.f;
                bv[t].z = (bvalid[t] && k + 2 < K) ? p[2 * a.ldb] : 0.f;
                bv[t].w = (bvalid[t] && k + 3 < K) ? p[3 * a.ldb] : 0.f;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(av, c), comp(bv[t], c), acc[t], 0, 0, 0);
    }

    // epilogue: lane (i, s) holds column n0 + 32t + i, rows m0 + (r & 3) + 8 (r >> 2) + 4 s
    float* __restrict__ Cg = a.C + g * a.gsC;
    const float* __restrict__ bias = a.bias ? a.bias + g * a.gsBias : nullptr;
    const int nvalid = static_cast<int>(min<int64_t>(32, a.M - m0));
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n0 + 32 * t + i;
        const float bn = (bias && n < N) ? bias[n] : 0.f;
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * s;
            float v = acc[t][r] + bn;
            const bool ok = row < nvalid && n < N;
            if (ok) {
                float* c = Cg + (m0 + row) * a.ldc + n;
                if (a.accumulate) v += *c;
                *c = v;
                sum += v;
            }
            acc[t][r] = v;
        }
        if (a.stat_partials) {
            // BatchNorm batch statistics of this 32-row tile, two-pass in registers
            sum += __shfl_xor(sum, 32);
            const float mean = sum / static_cast<float>(nvalid);
            float m2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * s;
                const float d = acc[t][r] - mean;
                if (row < nvalid) m2 = fmaf(d, d, m2);
            }
            m2 += __shfl_xor(m2, 32);
            if (s == 0 && n < N) {
                float* sp = a.stat_partials + ((tile_m * a.groups + g) * N + n) * 2;   // [tiles][groups * N][2]
                sp[0] = mean;
                sp[1] = m2;
            }
        }
    }
}

template <bool BT>
static int launch_rows(const swr_gemm_args* args, void* stream) {
    SWR_REQUIRE(args != nullptr, SWR_ERR_ARG);
    const swr_gemm_args& a = *args;
    SWR_REQUIRE(a.M >= 0 && a.N > 0 && a.K > 0 && a.A && a.B && a.C && a.groups >= 1, SWR_ERR_ARG);
    SWR_REQUIRE(a.lda >= a.K && a.ldc >= a.N && a.ldb >= (BT ? a.K : a.N), SWR_ERR_ARG);
    SWR_REQUIRE((a.a_scale == nullptr) == (a.a_shift == nullptr), SWR_ERR_ARG);
    if (a.M == 0) return SWR_OK;
    GemmK kk;
    kk.a = a;
    kk.n_tiles_m = static_cast<int>(swr_ceil_div(a.M, 32));
    const bool pro = a.a_scale != nullptr;
    bool vec = swr_aligned16(a.A) && a.lda % 4 == 0 && a.gsA % 4 == 0;
    if (BT) vec = vec && swr_aligned16(a.B) && a.ldb % 4 == 0 && a.gsB % 4 == 0;
    if (pro) vec = vec && swr_aligned16(a.a_scale) && swr_aligned16(a.a_shift) && a.gsScale % 4 == 0;
    const int tiles = static_cast<int>(swr_ceil_div(a.N, 32));
    const int nblk = static_cast<int>(swr_ceil_div(tiles, 8));
    const int nt = static_cast<int>(swr_ceil_div(tiles, nblk));
    const dim3 grid(static_cast<unsigned>(swr_ceil_div(kk.n_tiles_m, GEMM_WAVES)), static_cast<unsigned>(nblk),
                    static_cast<unsigned>(a.groups));
    hipStream_t st = static_cast<hipStream_t>(stream);
#define GO(NTV)                                                                                                   \
    do {                                                                                                          \
        if (vec && pro) hipLaunchKernelGGL((gemm_rows_kernel<NTV, BT, true, true>), grid, dim3(GEMM_THREADS), 0, st, kk);   \
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
    return swr_launch_status();
}

extern "C" int swr_gemm_nt(const swr_gemm_args* args, void* stream) { return launch_rows<true>(args, stream); }
extern "C" int swr_gemm_nn(const swr_gemm_args* args, void* stream) { return launch_rows<false>(args, stream); }

// ------------------------------------------------------------------------------------------------ tn
#define TN_TA_MAX 5
#define TN_TB 2

struct TnK {
    swr_gemm_tn_args a;
    int splits;
    int64_t rows_per_split;   // even
    float* part;              // [groups][splits][K1][K2]
    float* part_cs;           // [groups][splits][K1]  (colsum) or null
};

template <int TA>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_tn_kernel(const TnK kk) {
    const swr_gemm_tn_args& a = kk.a;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, s = lane >> 5;
    const int split = (blockIdx.z % ((kk.splits + GEMM_WAVES - 1) / GEMM_WAVES)) * GEMM_WAVES + wave;
    const int g = blockIdx.z / ((kk.splits + GEMM_WAVES - 1) / GEMM_WAVES);
    if (split >= kk.splits) return;
    const int p0 = blockIdx.y * (32 * TA);
    const int q0 = blockIdx.x * (32 * TN_TB);
    const int64_t ms = split * kk.rows_per_split;
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

#pragma unroll 2
    for (int64_t mb = ms; mb < me; mb += 2) {
        const int64_t m = mb + s;
        const bool okm = m < me;
        const int64_t mc = okm ? m : me - 1;
        float av[TA], bv[TN_TB];
#pragma unroll
        for (int t = 0; t < TA; ++t) {
            av[t] = Ag[mc * a.lda + ca[t]];
            if (!(okm && pa[t])) av[t] = 0.f;
            cs[t] += av[t];
        }
#pragma unroll
        for (int t = 0; t < TN_TB; ++t) {
            bv[t] = Bg[mc * a.ldb + cb[t]];
            if (!(okm && pb[t])) bv[t] = 0.f;
        }
#pragma unroll
        for (int ta = 0; ta < TA; ++ta)
#pragma unroll
            for (int tb = 0; tb < TN_TB; ++tb)
                acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ta], bv[tb], acc[ta][tb], 0, 0, 0);
    }

    float* __restrict__ P = kk.part + (static_cast<int64_t>(g) * kk.splits + split) * a.K1 * a.K2;
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
    if (kk.part_cs && blockIdx.x == 0) {
#pragma unroll
        for (int t = 0; t < TA; ++t) {
            const float tot = cs[t] + __shfl_xor(cs[t], 32);
            const int p = p0 + 32 * t + i;
            if (s == 0 && p < a.K1) kk.part_cs[(static_cast<int64_t>(g) * kk.splits + split) * a.K1 + p] = tot;
        }
    }
}

__global__ __launch_bounds__(256) void tn_reduce_kernel(const TnK kk) {
    const swr_gemm_tn_args& a = kk.a;
    const int g = blockIdx.y;
    const int64_t n = static_cast<int64_t>(a.K1) * a.K2;
    const int64_t j = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (j < n) {
        const float* p = kk.part + static_cast<int64_t>(g) * kk.splits * n + j;
        float sum = 0.f;
        for (int sp = 0; sp < kk.splits; ++sp) sum += p[sp * n];
        const int64_t r = j / a.K2, c = j - r * a.K2;
        float* dst = a.C + g * a.gsC + r * a.ldc + c;
        *dst = a.accumulate ? *dst + sum : sum;
    }
    if (kk.part_cs && j < a.K1) {
        const float* p = kk.part_cs + static_cast<int64_t>(g) * kk.splits * a.K1 + j;
        float sum = 0.f;
        for (int sp = 0; sp < kk.splits; ++sp) sum += p[sp * a.K1];
        float* dst = a.colsum + g * a.gsColsum + j;
        *dst = a.accumulate ? *dst + sum : sum;
    }
}

static int tn_plan(const swr_gemm_tn_args& a, int& ta, int& splits, int64_t& rps) {
    const int tiles1 = static_cast<int>(swr_ceil_div(a.K1, 32));
    const int pblk = static_cast<int>(swr_ceil_div(tiles1, TN_TA_MAX));
    ta = static_cast<int>(swr_ceil_div(tiles1, pblk));
    const int qblk = static_cast<int>(swr_ceil_div(a.K2, 32 * TN_TB));
    const int64_t tiles = static_cast<int64_t>(pblk) * qblk * a.groups;
    int64_t want = std::max<int64_t>(1, 2048 / tiles);                 // ~2 waves per SIMD over the chip
    want = std::min<int64_t>(want, std::max<int64_t>(1, a.M / 64));    // at least 64 rows per wave
    rps = swr_ceil_div(a.M, want);
    rps += rps & 1;
    splits = static_cast<int>(swr_ceil_div(a.M, rps));
    return pblk;
}

extern "C" size_t swr_gemm_tn_workspace_bytes(const swr_gemm_tn_args* args) {
    if (!args || args->M <= 0 || args->K1 <= 0 || args->K2 <= 0 || args->groups < 1) return 0;
    int ta, splits;
    int64_t rps;
    tn_plan(*args, ta, splits, rps);
    return static_cast<size_t>(args->groups) * splits * (static_cast<size_t>(args->K1) * args->K2 + args->K1) * 4 + 256;
}

extern "C" int swr_gemm_tn(const swr_gemm_tn_args* args, void* workspace, size_t workspace_bytes, void* stream) {
    SWR_REQUIRE(args != nullptr, SWR_ERR_ARG);
    const swr_gemm_tn_args& a = *args;
    SWR_REQUIRE(a.M >= 0 && a.K1 > 0 && a.K2 > 0 && a.A && a.B && a.C && a.groups >= 1, SWR_ERR_ARG);
    SWR_REQUIRE(a.lda >= a.K1 && a.ldb >= a.K2 && a.ldc >= a.K2, SWR_ERR_ARG);
    hipStream_t st = static_cast<hipStream_t>(stream);
    TnK kk;
    kk.a = a;
    if (a.M == 0) {
        // empty batch: the gradient is zero
        if (a.accumulate) return SWR_OK;
        for (int g = 0; g < a.groups; ++g) {
            for (int r = 0; r < a.K1; ++r)
                if (hipMemsetAsync(a.C + g * a.gsC + static_cast<int64_t>(r) * a.ldc, 0, sizeof(float) * a.K2, st) != hipSuccess)
                    return SWR_ERR_LAUNCH;
            if (a.colsum && hipMemsetAsync(a.colsum + g * a.gsColsum, 0, sizeof(float) * a.K1, st) != hipSuccess)
                return SWR_ERR_LAUNCH;
        }
        return SWR_OK;
    }
    int ta;
    const int pblk = tn_plan(a, ta, kk.splits, kk.rows_per_split);
    const size_t need = swr_gemm_tn_workspace_bytes(args);
    SWR_REQUIRE(workspace != nullptr && workspace_bytes >= need, SWR_ERR_WORKSPACE);
    kk.part = static_cast<float*>(workspace);
    kk.part_cs = a.colsum ? kk.part + static_cast<size_t>(a.groups) * kk.splits * a.K1 * a.K2 : nullptr;
    const int zsplit = (kk.splits + GEMM_WAVES - 1) / GEMM_WAVES;
    const dim3 grid(static_cast<unsigned>(swr_ceil_div(a.K2, 32 * TN_TB)), static_cast<unsigned>(pblk),
                    static_cast<unsigned>(zsplit * a.groups));
    switch (ta) {
        case 1: hipLaunchKernelGGL(gemm_tn_kernel<1>, grid, dim3(GEMM_THREADS), 0, st, kk); break;
        case 2: hipLaunchKernelGGL(gemm_tn_kernel<2>, grid, dim3(GEMM_THREADS), 0, st, kk); break;
        case 3: hipLaunchKernelGGL(gemm_tn_kernel<3>, grid, dim3(GEMM_THREADS), 0, st, kk); break;
        case 4: hipLaunchKernelGGL(gemm_tn_kernel<4>, grid, dim3(GEMM_THREADS), 0, st, kk); break;
        default: hipLaunchKernelGGL(gemm_tn_kernel<5>, grid, dim3(GEMM_THREADS), 0, st, kk); break;
    }
    const int64_t n = static_cast<int64_t>(a.K1) * a.K2;
    hipLaunchKernelGGL(tn_reduce_kernel, dim3(static_cast<unsigned>(swr_ceil_div(n, 256)), static_cast<unsigned>(a.groups)),
                       dim3(256), 0, st, kk);
    return swr_launch_status();
}
