// Per-domain tower heads: G independent [Linear(K, H) -> BatchNorm1d(H) -> ReLU -> Linear(H, 1)] evaluated together
// (the `towers` of mmoe.py:38-41,50-51 / ple.py / sharebottom.py with tower_params = {"dims": [H]}).
//
// The products are tiny (K = 32, H = 16 at the KuaiRand config: 0.3 GFLOP per pass) but the activations are not
// ([B, G*K] and [B, G*H] fp32 = 42 + 21 MB at B = 65 536), so the generic GEMM + BN + activation launches (nine in the
// backward pass) are bound by HBM passes and launch latency.  Here every pass over the batch is ONE kernel whose
// thread owns one (row, tower) pair, keeps the H hidden values in registers and reads the weights as wave-uniform
// scalars:
//
//   forward  : tower_linear_fwd  Z1 = X_g W1_g^T + b1, with the per-32-row (mean, M2) tiles of the BatchNorm statistics
//              (swr_bn_finalize)
//              tower_head_fwd    V[:, g] = relu(scale * Z1 + shift)_g . w2_g + b2_g          (A1 is never stored)
//   backward : tower_bwd_stats   per-64-row tiles of (sum dY, sum dY * xhat) and of dW2, db2, A1 recomputed from Z1
//              tower_bwd_finalize   fixed-order fp64 sums -> dgamma, dbeta, dW2, db2 and the BN backward coefficients
//              tower_bwd_apply   dZ1 = ca dY + cb (Z1 - mean) + cc  and  dX_g = dZ1_g W1_g
//              (dW1, db1 = the ordinary grouped weight-gradient product swr_gemm_tn on dZ1, X)
//
// All reductions are per-tile partials summed in a fixed order: deterministic.
#include "common.h"

#ifndef TW_THREADS
#define TW_THREADS 256
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
// shapes the matrix-pipe paths take: whole 16-wide column tiles, k fragments of whole 16-byte pieces
#define TW_MFMA_OK(K, H) ((K) % 16 == 0 && (H) % 16 == 0 && TW_THREADS % 64 == 0)

struct TowerK {
    swr_tower_args a;
    float* bn_partials;     // [tiles of 64 rows][G*H][2]   (sum dY, sum dY * xhat)
    float* head_partials;   // [tiles][G*H + G]             (dW2 columns, then db2 per tower)
};

__device__ __forceinline__ float half_wave_sum(float v) {   // over the 32 lanes of a half-wave (one 32-row tile)
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v += __shfl_xor(v, off);
    return v;
}

// dV[m, g]: read, or -- selected mode (swr.h) -- implied by the fused select + BCE: the gradient of the mean BCE with
// respect to the logit of the row's own tower, zero for the other towers
__device__ __forceinline__ float tower_dv(const swr_tower_args& a, int64_t m, int g) {
    if (a.dV) return a.dV[m * a.lddv + g];
    if (swr_load_index(a.sel_domain, a.sel_dom_dtype, m) != g) return 0.f;
    return swr_bce_logit_grad(a.sel_p[m], swr_load_value(a.sel_y, a.sel_y_dtype, m), a.sel_dloss[0], a.M);
}

template <int N4>
__device__ __forceinline__ void load_row(const float* __restrict__ p, float (&v)[4 * N4]) {
#pragma unroll
    for (int q = 0; q < N4; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(p + 4 * q);
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
}

// Tiles of TW_THREADS rows x W columns cross HBM with consecutive lanes on consecutive 16-byte pieces of a row
// (coalesced) and live in LDS with a row pitch of W + 4 floats (the per-thread 16-byte row accesses are then
// conflict-free).  A thread owns one row of the tile.
template <int W>
__device__ __forceinline__ void tile_load(const float* __restrict__ src, int64_t ld, int rows, float* lds) {
    constexpr int P = W + 4, Q = W / 4;
    for (int idx = threadIdx.x; idx < rows * Q; idx += TW_THREADS) {
        const int r = idx / Q, c = idx - r * Q;
        *reinterpret_cast<float4*>(lds + r * P + 4 * c) = *reinterpret_cast<const float4*>(src + r * ld + 4 * c);
    }
}
template <int W>
__device__ __forceinline__ void tile_store(float* __restrict__ dst, int64_t ld, int rows, const float* lds) {
    constexpr int P = W + 4, Q = W / 4;
    for (int idx = threadIdx.x; idx < rows * Q; idx += TW_THREADS) {
        const int r = idx / Q, c = idx - r * Q;
        *reinterpret_cast<float4*>(dst + r * ld + 4 * c) = *reinterpret_cast<const float4*>(lds + r * P + 4 * c);
    }
}

// The hidden index j is a REAL loop (one weight row per trip) and the H values of a row live in the LDS tile, not in a
// register array: fully unrolled with the weights read through the (wave-uniform) argument pointers, hipcc hoists all
// H*K scalar loads to the top of the kernel and spills ~450 SGPRs through v_writelane / v_readlane (3x slower).

// ---------------------------------------------------------------------------------- forward, first layer
template <int K, int H>
__global__ __launch_bounds__(TW_THREADS) void tower_linear_fwd_kernel(const TowerK kk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PX = K + 4, PZ = H + 4;
    float* lx = lds;                                          // [TW_THREADS][PX]
    float* lz = lds;                                          // [TW_THREADS][PZ], reuses the X tile once it is in registers
    const swr_tower_args& a = kk.a;
    const int g = blockIdx.y;
    const int64_t m0 = static_cast<int64_t>(blockIdx.x) * TW_THREADS;
    const int rows = static_cast<int>(min<int64_t>(TW_THREADS, a.M - m0));
    const bool valid = static_cast<int>(threadIdx.x) < rows;
    if constexpr (TW_MFMA_OK(K, H)) {
        // ---- the product on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation).  The
        // scalar path below stages X through a [256][K + 4] LDS tile (four workgroups per CU for the five a CU owes at the
        // KuaiRand shape: a second, quarter-full round) and reads every weight through an LDS broadcast once per WAVE
        // (H*K/4 ds_read_b128 per wave: 18 us of LDS pipe).  Here a wave keeps the tower's W1 in K/4 * H/16 registers and
        // takes its A fragments straight from HBM: lane group kq holds k = KS*kq .. KS*kq + KS - 1 of row n (k is permuted
        // the same way in both operands), so the four groups of a row read one contiguous 4*K-byte piece.  LDS holds only
        // the Z1 tile (coalesced stores, statistics): every workgroup of the launch is resident at once.
        constexpr int KS = K / 4, CT = H / 16, RT = 4;                     // k steps, column tiles, 16-row tiles per wave (64 rows)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int n = lane & 15, kq = lane >> 4;
        float af[RT][KS];
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int64_t m = min<int64_t>(m0 + wave * 64 + 16 * t + n, a.M - 1);
            load_row<KS / 4>(a.X + m * a.ldx + static_cast<int64_t>(g) * K + KS * kq, af[t]);
        }
        float bf[CT][KS], bias[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)                       // (4-byte loads: a parameter view need not be 16-byte aligned)
                bf[ct][ks] = a.W1[(static_cast<int64_t>(g) * H + 16 * ct + n) * K + KS * kq + ks];
            bias[ct] = a.b1 ? a.b1[g * H + 16 * ct + n] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                f32x4 acc = {bias[ct], bias[ct], bias[ct], bias[ct]};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t][ks], bf[ct][ks], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {                     // D[row 4 kq + r][column n]
                    const int row = wave * 64 + 16 * t + 4 * kq + r;
                    lz[row * PZ + 16 * ct + n] = row < rows ? acc[r] : 0.f;
                }
            }
        }
    } else {
    float* lw = lds + TW_THREADS * (PX > PZ ? PX : PZ);       // [H][K] weights of this tower (+ [H] biases)
    tile_load<K>(a.X + m0 * a.ldx + static_cast<int64_t>(g) * K, a.ldx, rows, lx);
    for (int i = threadIdx.x; i < H * K; i += TW_THREADS) lw[i] = a.W1[static_cast<int64_t>(g) * H * K + i];
    if (static_cast<int>(threadIdx.x) < H) lw[H * K + threadIdx.x] = a.b1 ? a.b1[g * H + threadIdx.x] : 0.f;
    __syncthreads();
    float x[K];
    load_row<K / 4>(lx + threadIdx.x * PX, x);
    __syncthreads();
    // weights as LDS broadcasts (one 16-byte read feeds 4 FMAs of the whole wave): scalar loads from the kernel
    // argument pointers cost ~0.4 us of latency per weight row and a wave has nothing else to do meanwhile
#pragma unroll 2
    for (int j = 0; j < H; ++j) {
        float acc = lw[H * K + j];
#pragma unroll
        for (int k = 0; k < K; ++k) acc = fmaf(x[k], lw[j * K + k], acc);
        lz[threadIdx.x * PZ + j] = valid ? acc : 0.f;
    }
    }
    __syncthreads();
    tile_store<H>(a.Z1 + m0 * a.ldz + g * H, a.ldz, rows, lz);
    if (!a.stat_partials) return;
    // (mean, M2) per 32-row tile and column, the layout swr_bn_finalize merges: TPP threads share one (tile, column),
    // each over H of its rows (two passes over LDS), combined with a fixed butterfly
    constexpr int TPP = 32 / H > 0 ? 32 / H : 1;              // H <= 32
    constexpr int RPT = 32 / TPP;
    const int pair = threadIdx.x / TPP, part = threadIdx.x % TPP;
    const int t8 = pair / H, j = pair % H;
    const int64_t tile = (m0 >> 5) + t8;
    const int tile_rows = static_cast<int>(min<int64_t>(32, a.M - tile * 32));
    const float* col = lz + (t8 * 32) * PZ + j;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < RPT; ++i) sum += col[(i * TPP + part) * PZ];           // rows past the end hold 0
#pragma unroll
    for (int off = 1; off < TPP; off <<= 1) sum += __shfl_xor(sum, off);
    const float mean = sum / static_cast<float>(tile_rows > 0 ? tile_rows : 1);
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = i * TPP + part;
        const float d = r < tile_rows ? col[r * PZ] - mean : 0.f;
        m2 = fmaf(d, d, m2);
    }
#pragma unroll
    for (int off = 1; off < TPP; off <<= 1) m2 += __shfl_xor(m2, off);
    if (part == 0 && tile_rows > 0) {
        float* p = a.stat_partials + (tile * (a.G * H) + g * H + j) * 2;
        p[0] = mean;
        p[1] = m2;
    }
}

// --------------------------------------------------------------------------- forward, BN + ReLU + output layer
template <int H>
__global__ __launch_bounds__(TW_THREADS) void tower_head_fwd_kernel(const TowerK kk) {
    const swr_tower_args& a = kk.a;
    const int64_t m = static_cast<int64_t>(blockIdx.x) * TW_THREADS + threadIdx.x;
    if (m >= a.M) return;
    for (int g = 0; g < a.G; ++g) {
        float z[H];
        load_row<H / 4>(a.Z1 + m * a.ldz + g * H, z);
        float v = a.b2 ? a.b2[g] : 0.f;
#pragma unroll
        for (int j = 0; j < H; ++j) {
            const float act = fmaxf(fmaf(z[j], a.scale[g * H + j], a.shift[g * H + j]), 0.f);
            v = fmaf(act, a.w2[g * H + j], v);
        }
        a.V[m * a.ldv + g] = v;
    }
}

// ----------------------------------------------------------------------------------- backward, statistics
// Z1 tile and dV column staged in LDS; TPP threads share one (64-row tile, column) and walk H of its rows each with the
// column's coefficients in registers: no per-element shuffles.
template <int H>
__global__ __launch_bounds__(TW_THREADS) void tower_bwd_stats_kernel(const TowerK kk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PZ = H + 4;
    float* lz = lds;                                          // [TW_THREADS][PZ]
    float* ldv = lds + TW_THREADS * PZ;                       // [TW_THREADS]
    const swr_tower_args& a = kk.a;
    const int g = blockIdx.y;
    const int64_t m0 = static_cast<int64_t>(blockIdx.x) * TW_THREADS;
    const int rows = static_cast<int>(min<int64_t>(TW_THREADS, a.M - m0));
    tile_load<H>(a.Z1 + m0 * a.ldz + g * H, a.ldz, rows, lz);
    ldv[threadIdx.x] = static_cast<int>(threadIdx.x) < rows ? tower_dv(a, m0 + threadIdx.x, g) : 0.f;
    __syncthreads();
    constexpr int TPP = 64 / H;                               // H <= 32 -> >= 2
    constexpr int RPT = 64 / TPP;
    const int pair = threadIdx.x / TPP, part = threadIdx.x % TPP;
    const int t4 = pair / H, j = pair % H;
    const int n = g * H + j, N = a.G * H;
    const float sc = a.scale[n], sh = a.shift[n], mu = a.mean[n], rs = a.rstd[n], w2 = a.w2[n];
    float s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = t4 * 64 + i * TPP + part;
        const float dv = ldv[r];                              // 0 on rows past the end: they add nothing
        const float zj = r < rows ? lz[r * PZ + j] : 0.f;    // (their LDS rows were never written: could hold NaN bits)
        const float pre = fmaf(zj, sc, sh);
        const bool on = pre > 0.f;
        const float dy = on ? dv * w2 : 0.f;
        s1 += dy;
        s2 = fmaf(dy, (zj - mu) * rs, s2);
        s3 = fmaf(on ? dv : 0.f, pre, s3);
        s4 += dv;
    }
#pragma unroll
    for (int off = 1; off < TPP; off <<= 1) {
        s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); s3 += __shfl_xor(s3, off); s4 += __shfl_xor(s4, off);
    }
    const int64_t tile = (m0 >> 6) + t4;
    if (part == 0 && tile * 64 < a.M) {
        float* p = kk.bn_partials + (tile * N + n) * 2;
        p[0] = s1;
        p[1] = s2;
        kk.head_partials[tile * (N + a.G) + n] = s3;
        if (j == 0) kk.head_partials[tile * (N + a.G) + N + g] = s4;
    }
}

// fixed-order fp64 sums over the 64-row tiles; one workgroup per hidden column (+ one per tower for db2)
__global__ __launch_bounds__(TW_THREADS) void tower_bwd_finalize_kernel(const TowerK kk, int n_tiles) {
    const swr_tower_args& a = kk.a;
    __shared__ double t1[TW_THREADS / 64], t2[TW_THREADS / 64], t3[TW_THREADS / 64];
    const int N = a.G * (a.H);
    const int n = blockIdx.x;                                                  // < N: hidden column, >= N: tower
    double a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const int per = (n_tiles + TW_THREADS - 1) / TW_THREADS;
    const int t0 = threadIdx.x * per;
    for (int t = t0; t < min(t0 + per, n_tiles); ++t) {
        if (n < N) {
            const float2 p = *reinterpret_cast<const float2*>(kk.bn_partials + (static_cast<int64_t>(t) * N + n) * 2);
            a1 += p.x;
            a2 += p.y;
        }
        a3 += kk.head_partials[static_cast<int64_t>(t) * (N + a.G) + n];
    }
    const double S1 = swr_block_sum_f64<TW_THREADS>(a1, t1), S2 = swr_block_sum_f64<TW_THREADS>(a2, t2);
    const double S3 = swr_block_sum_f64<TW_THREADS>(a3, t3);
    if (threadIdx.x != 0) return;
    const float s3 = static_cast<float>(S3);
    if (n >= N) {
        if (a.db2) a.db2[n - N] = (a.accumulate ? a.db2[n - N] : 0.f) + s3;
        return;
    }
    if (a.dw2) a.dw2[n] = (a.accumulate ? a.dw2[n] : 0.f) + s3;
    const double gm = a.gamma ? a.gamma[n] : 1.0, rs = a.rstd[n];
    if (a.dgamma) a.dgamma[n] = (a.accumulate ? a.dgamma[n] : 0.f) + static_cast<float>(S2);
    if (a.dbeta) a.dbeta[n] = (a.accumulate ? a.dbeta[n] : 0.f) + static_cast<float>(S1);
    // dZ = g rs dY - g rs^2 (S2 / M) (Z - mean) - g rs (S1 / M)      (same coefficients as swr_bn_bwd_finalize)
    a.ca[n] = static_cast<float>(gm * rs);
    a.cb[n] = static_cast<float>(-gm * rs * rs * S2 / static_cast<double>(a.M));
    a.cc[n] = static_cast<float>(-gm * rs * S1 / static_cast<double>(a.M));
}

// --------------------------------------------------------------------------------------- backward, apply
template <int K, int H>
__global__ __launch_bounds__(TW_THREADS) void tower_bwd_apply_kernel(const TowerK kk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PX = K + 4, PZ = H + 4;
    const swr_tower_args& a = kk.a;
    const int g = blockIdx.y;
    const int64_t m0 = static_cast<int64_t>(blockIdx.x) * TW_THREADS;
    const int rows = static_cast<int>(min<int64_t>(TW_THREADS, a.M - m0));
    if constexpr (TW_MFMA_OK(H, K)) {
        // ---- matrix-pipe path (see tower_linear_fwd_kernel).  dZ1 is computed directly in the A-fragment layout -- lane
        // (n, kq) owns hidden units KS*kq .. KS*kq + KS - 1 of row n of each 16-row tile, with the same operations in the
        // same order as the scalar path -- from Z1 read straight from HBM (the four lane groups of a row read one
        // contiguous 4*H-byte piece) and stored the same way; dX_g = dZ1_g W1_g leaves through a wave-private 16-row
        // staging tile (16-byte coalesced stores).  9 KB of LDS instead of 39: the launch is resident at once.
        constexpr int KS = H / 4, CT = K / 16, RT = 4;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int n = lane & 15, kq = lane >> 4;
        float sc[KS], sh[KS], w2[KS], cb[KS], mu[KS], ca[KS], cc[KS];
        {
            const int c0 = g * H + KS * kq;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {                     // (4-byte loads: parameter views need not be 16-byte aligned)
                sc[ks] = a.scale[c0 + ks]; sh[ks] = a.shift[c0 + ks]; w2[ks] = a.w2[c0 + ks]; cb[ks] = a.cb[c0 + ks];
                mu[ks] = a.mean[c0 + ks];  ca[ks] = a.ca[c0 + ks];    cc[ks] = a.cc[c0 + ks];
            }
        }
        float bf[CT][KS];                                         // B[k = hidden KS*kq + s][column 16 ct + n] = W1[hidden][column]
        if (a.dX) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    bf[ct][ks] = a.W1[(static_cast<int64_t>(g) * H + KS * kq + ks) * K + 16 * ct + n];
        }
        float dz[RT][KS];
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int row = wave * 64 + 16 * t + n;
            const bool ok = row < rows;
            const int64_t m = min<int64_t>(m0 + row, a.M - 1);
            load_row<KS / 4>(a.Z1 + m * a.ldz + g * H + KS * kq, dz[t]);
            const float dv = ok ? tower_dv(a, m, g) : 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const float zj = dz[t][ks];
                const float pre = fmaf(zj, sc[ks], sh[ks]);
                const float dy = pre > 0.f ? dv * w2[ks] : 0.f;
                dz[t][ks] = fmaf(cb[ks], zj - mu[ks], dy * ca[ks]) + cc[ks];
            }
            if (ok) {
                float* o = a.dZ1 + m * a.lddz + g * H + KS * kq;
#pragma unroll
                for (int q = 0; q < KS / 4; ++q)
                    *reinterpret_cast<float4*>(o + 4 * q) = make_float4(dz[t][4 * q], dz[t][4 * q + 1], dz[t][4 * q + 2], dz[t][4 * q + 3]);
            }
        }
        if (!a.dX) return;
        float* stage = lds + wave * (16 * PX);
#pragma unroll
        for (int t = 0; t < RT; ++t) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dz[t][ks], bf[ct][ks], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) stage[(4 * kq + r) * PX + 16 * ct + n] = acc[r];
            }
            __builtin_amdgcn_wave_barrier();                      // (same wave: the LDS queue is in order; this only pins hipcc)
            constexpr int Q = K / 4;
#pragma unroll
            for (int u = 0; u < 16 * Q / 64; ++u) {
                const int idx = lane + 64 * u, r = idx / Q, c = idx - r * Q;
                const int row = wave * 64 + 16 * t + r;
                const float4 v = *reinterpret_cast<const float4*>(stage + r * PX + 4 * c);
                if (row < rows) *reinterpret_cast<float4*>(a.dX + (m0 + row) * a.lddx + static_cast<int64_t>(g) * K + 4 * c) = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    float* lz = lds;                                          // [TW_THREADS][PZ]: Z1 in, dZ1 out (in place)
    float* lx = lds;                                          // [TW_THREADS][PX]: dX out, after dZ1 has left
    const bool valid = static_cast<int>(threadIdx.x) < rows;
    float* lw = lds + TW_THREADS * (PX > PZ ? PX : PZ);       // [H][K] weights, then 7 x [H] per-column coefficients
    float* lc = lw + H * K;
    tile_load<H>(a.Z1 + m0 * a.ldz + g * H, a.ldz, rows, lz);
    for (int i = threadIdx.x; i < H * K; i += TW_THREADS) lw[i] = a.W1[static_cast<int64_t>(g) * H * K + i];
    if (static_cast<int>(threadIdx.x) < H) {
        const int n = g * H + threadIdx.x;
        lc[threadIdx.x] = a.scale[n];         lc[H + threadIdx.x] = a.shift[n];     lc[2 * H + threadIdx.x] = a.w2[n];
        lc[3 * H + threadIdx.x] = a.cb[n];    lc[4 * H + threadIdx.x] = a.mean[n];  lc[5 * H + threadIdx.x] = a.ca[n];
        lc[6 * H + threadIdx.x] = a.cc[n];
    }
    const float dv = valid ? tower_dv(a, m0 + threadIdx.x, g) : 0.f;
    __syncthreads();
    float* zrow = lz + threadIdx.x * PZ;
    float dx[K];
#pragma unroll
    for (int k = 0; k < K; ++k) dx[k] = 0.f;
#pragma unroll 2
    for (int j = 0; j < H; ++j) {
        const float zj = valid ? zrow[j] : 0.f;
        const float pre = fmaf(zj, lc[j], lc[H + j]);
        const float dy = pre > 0.f ? dv * lc[2 * H + j] : 0.f;
        const float dz = fmaf(lc[3 * H + j], zj - lc[4 * H + j], dy * lc[5 * H + j]) + lc[6 * H + j];
        zrow[j] = dz;
        if (a.dX) {
#pragma unroll
            for (int k = 0; k < K; ++k) dx[k] = fmaf(dz, lw[j * K + k], dx[k]);
        }
    }
    __syncthreads();
    tile_store<H>(a.dZ1 + m0 * a.lddz + g * H, a.lddz, rows, lz);
    if (!a.dX) return;
    __syncthreads();
    float* xrow = lx + threadIdx.x * PX;
#pragma unroll
    for (int q = 0; q < K / 4; ++q)
        *reinterpret_cast<float4*>(xrow + 4 * q) = make_float4(dx[4 * q], dx[4 * q + 1], dx[4 * q + 2], dx[4 * q + 3]);
    __syncthreads();
    tile_store<K>(a.dX + m0 * a.lddx + static_cast<int64_t>(g) * K, a.lddx, rows, lx);
}

// ------------------------------------------------------------------------------------------- dispatch
static bool tower_shape_ok(int K, int H) {
    return (K == 8 || K == 16 || K == 32) && (H == 4 || H == 8 || H == 16 || H == 32);   // K = 64 would need > 64 KB of LDS
}

// ---------------------------------------------------------------------------- backward, first-layer weights
// dW1_g = dZ1_g^T X_g, db1_g = column sums of dZ1_g for all G towers in ONE pass over dZ1 and X (63 MB at the KuaiRand shape): the
// generic grouped weight-gradient product (gemm_tn_kernel<1> + its reduce, 24 + 5 us) multiplies 5 slivers of 16 x 32 outputs
// through a kernel built for wide outputs.  Here a workgroup walks 64-row tiles (coalesced 16-byte loads into LDS, the next
// tile's loads in flight in registers), wave w owns rows 16 w .. 16 w + 15 of every tile and keeps all G * H/16 * K/16 output
// tiles (+ the column sums, as a product with ones) in v_mfma_f32_16x16x4_f32 accumulators: exact fp32 products.  The four
// waves' sums are added in a fixed tree, the workgroups' partials by tower_dw_reduce_kernel in a fixed order: deterministic.
#define TDW_ROWS 64
template <int K, int H, int G>
__global__ __launch_bounds__(TW_THREADS) void tower_dw_kernel(const float* __restrict__ dZ1, int64_t ldz, const float* __restrict__ X,
                                                              int64_t ldx, int64_t M, float* __restrict__ part) {
    constexpr int WX = G * K, WZ = G * H, PX = WX + 4, PZ = WZ + 4;
    constexpr int QX = WX / 4, QZ = WZ / 4;                        // 16-byte pieces per row
    constexpr int NX = (TDW_ROWS * QX + TW_THREADS - 1) / TW_THREADS, NZ = (TDW_ROWS * QZ + TW_THREADS - 1) / TW_THREADS;
    constexpr int CT = H / 16, KT = K / 16;
    constexpr int OUT = G * H * K + G * H;                         // floats of one partial: dW1 [G][H][K], then db1 [G][H]
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lx = lds;                                               // [64][PX]
    float* lz = lds + TDW_ROWS * PX;                               // [64][PZ]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int64_t n_tiles = (M + TDW_ROWS - 1) / TDW_ROWS;
    f32x4 acc[G][CT][KT], accb[G][CT];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            accb[g][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < KT; ++t) acc[g][ct][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    float4 rx[NX], rz[NZ];
    auto tile_fetch = [&](int64_t tile) {                           // rows past the end: zeros (they add nothing)
        const int64_t m0 = tile * TDW_ROWS;
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int idx = threadIdx.x + u * TW_THREADS, r = idx / QX, c = idx - r * QX;
            rx[u] = (idx < TDW_ROWS * QX && m0 + r < M) ? *reinterpret_cast<const float4*>(X + (m0 + r) * ldx + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < NZ; ++u) {
            const int idx = threadIdx.x + u * TW_THREADS, r = idx / QZ, c = idx - r * QZ;
            rz[u] = (idx < TDW_ROWS * QZ && m0 + r < M) ? *reinterpret_cast<const float4*>(dZ1 + (m0 + r) * ldz + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto tile_put = [&]() {
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int idx = threadIdx.x + u * TW_THREADS, r = idx / QX, c = idx - r * QX;
            if (idx < TDW_ROWS * QX) *reinterpret_cast<float4*>(lx + r * PX + 4 * c) = rx[u];
        }
#pragma unroll
        for (int u = 0; u < NZ; ++u) {
            const int idx = threadIdx.x + u * TW_THREADS, r = idx / QZ, c = idx - r * QZ;
            if (idx < TDW_ROWS * QZ) *reinterpret_cast<float4*>(lz + r * PZ + 4 * c) = rz[u];
        }
    };
    int64_t tile = blockIdx.x;
    if (tile < n_tiles) tile_fetch(tile);
    for (; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();                                            // the previous tile's fragments have been read
        tile_put();
        __syncthreads();
        if (tile + gridDim.x < n_tiles) tile_fetch(tile + gridDim.x);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int r = 16 * wave + 4 * ks + kq;                  // this lane's row (k index) of the step
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float bf[KT];
#pragma unroll
                for (int t = 0; t < KT; ++t) bf[t] = lx[r * PX + g * K + 16 * t + n];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const float af = lz[r * PZ + g * H + 16 * ct + n];
#pragma unroll
                    for (int t = 0; t < KT; ++t) acc[g][ct][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[t], acc[g][ct][t], 0, 0, 0);
                    accb[g][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, 1.f, accb[g][ct], 0, 0, 0);
                }
            }
        }
    }
    // ---- the four waves' tiles -> one partial per workgroup: ((w0 + w1) + (w2 + w3)) through LDS
    __syncthreads();
    float* red = lds;                                               // [4][OUT]
    static_assert(4 * OUT <= TDW_ROWS * (PX + PZ), "reduction buffer larger than the tiles");
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)          // D[row 4 kq + r][column n]: dW1[g][16 ct + 4 kq + r][16 t + n]
                    red[wave * OUT + (g * H + 16 * ct + 4 * kq + r) * K + 16 * t + n] = acc[g][ct][t][r];
            if (n == 0)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave * OUT + G * H * K + g * H + 16 * ct + 4 * kq + r] = accb[g][ct][r];
        }
    __syncthreads();
    float* P = part + static_cast<int64_t>(blockIdx.x) * OUT;
    for (int j = threadIdx.x; j < OUT; j += TW_THREADS) P[j] = (red[j] + red[OUT + j]) + (red[2 * OUT + j] + red[3 * OUT + j]);
}

// out[j] (+)= sum over the workgroups' partials, in order: 32 threads share four consecutive outputs and take partials sub, sub + 32,
// ... (8 loads in flight); their sums are added in sub order through LDS.  (16 sharers and 4 loads in flight: 11 us for 5 MB.)
#define TDR_SUBS 32
__global__ __launch_bounds__(256) void tower_dw_reduce_kernel(const float* __restrict__ part, int n_part, int out, int n_w, float* dW1,
                                                              float* db1, int accumulate) {
    constexpr int OUT4 = 256 / TDR_SUBS;
    __shared__ float4 red[256];
    const int sub = threadIdx.x / OUT4, o = threadIdx.x % OUT4;
    const int j = 4 * (blockIdx.x * OUT4 + o);
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < out) {
        int sp = sub;
        for (; sp + 7 * TDR_SUBS < n_part; sp += 8 * TDR_SUBS) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(part + static_cast<int64_t>(sp + TDR_SUBS * u) * out + j);
#pragma unroll
            for (int u = 0; u < 8; ++u) { sum.x += v[u].x; sum.y += v[u].y; sum.z += v[u].z; sum.w += v[u].w; }
        }
        for (; sp < n_part; sp += TDR_SUBS) {
            const float4 v = *reinterpret_cast<const float4*>(part + static_cast<int64_t>(sp) * out + j);
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
    }
    red[threadIdx.x] = sum;
    __syncthreads();
    if (sub != 0 || j >= out) return;
    float4 t = red[o];
#pragma unroll
    for (int q = 1; q < TDR_SUBS; ++q) { const float4 v = red[q * OUT4 + o]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float* dst = j + e < n_w ? dW1 + j + e : db1 + (j + e - n_w);
        if (j + e < n_w || db1) *dst = accumulate ? *dst + tv[e] : tv[e];
    }
}

static int tower_dw_blocks(int64_t M) { return static_cast<int>(std::min<int64_t>(512, (M + TDW_ROWS - 1) / TDW_ROWS)); }
extern "C" int swr_tower_dw_supported(int K, int H, int G) { return (K == 32 && H == 16 && G >= 1 && G <= 6) ? 1 : 0; }
extern "C" size_t swr_tower_dw_workspace_bytes(int64_t M, int K, int H, int G) {
    if (!swr_tower_dw_supported(K, H, G) || M <= 0) return 0;
    return static_cast<size_t>(tower_dw_blocks(M)) * (static_cast<size_t>(G) * H * K + G * H) * sizeof(float) + 256;
}
extern "C" int swr_tower_dw(const float* dZ1, int64_t ldz, const float* X, int64_t ldx, int64_t M, int K, int H, int G, float* dW1,
                            float* db1, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    SWR_REQUIRE(dZ1 && X && dW1 && M > 0 && ldz >= G * H && ldx >= G * K, SWR_ERR_ARG);
    SWR_REQUIRE(swr_tower_dw_supported(K, H, G), SWR_ERR_UNSUPPORTED);
    SWR_REQUIRE(ldz % 4 == 0 && ldx % 4 == 0 && swr_aligned16(dZ1) && swr_aligned16(X), SWR_ERR_ALIGN);
    SWR_REQUIRE(workspace && workspace_bytes >= swr_tower_dw_workspace_bytes(M, K, H, G) && swr_aligned16(workspace), SWR_ERR_WORKSPACE);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nb = tower_dw_blocks(M);
    float* part = static_cast<float*>(workspace);
    const int out = G * H * K + G * H;
#define TDW(GV)                                                                                                            \
    case GV: {                                                                                                             \
        const size_t lds = static_cast<size_t>(TDW_ROWS) * (GV * 32 + 4 + GV * 16 + 4) * sizeof(float);                    \
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(tower_dw_kernel<32, 16, GV>),             \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)   \
            return SWR_ERR_LAUNCH;                                                                                         \
        hipLaunchKernelGGL((tower_dw_kernel<32, 16, GV>), dim3(nb), dim3(TW_THREADS), lds, st, dZ1, ldz, X, ldx, M, part); \
        break;                                                                                                             \
    }
    switch (G) { TDW(1) TDW(2) TDW(3) TDW(4) TDW(5) TDW(6) default: return SWR_ERR_UNSUPPORTED; }
#undef TDW
    hipLaunchKernelGGL(tower_dw_reduce_kernel, dim3((out / 4 + 256 / TDR_SUBS - 1) / (256 / TDR_SUBS)), dim3(256), 0, st, part, nb, out, G * H * K, dW1, db1, accumulate);
    return swr_launch_status();
}

extern "C" int swr_tower_supported(int K, int H) { return tower_shape_ok(K, H) ? 1 : 0; }

#define TW_DISPATCH_KH(FN, K, H, ...)                                   \
    do {                                                                \
        switch ((K) * 100 + (H)) {                                      \
            case 804: FN<8, 4> __VA_ARGS__; break;                      \
            case 808: FN<8, 8> __VA_ARGS__; break;                      \
            case 816: FN<8, 16> __VA_ARGS__; break;                     \
            case 832: FN<8, 32> __VA_ARGS__; break;                     \
            case 1604: FN<16, 4> __VA_ARGS__; break;                    \
            case 1608: FN<16, 8> __VA_ARGS__; break;                    \
            case 1616: FN<16, 16> __VA_ARGS__; break;                   \
            case 1632: FN<16, 32> __VA_ARGS__; break;                   \
            case 3204: FN<32, 4> __VA_ARGS__; break;                    \
            case 3208: FN<32, 8> __VA_ARGS__; break;                    \
            case 3216: FN<32, 16> __VA_ARGS__; break;                   \
            case 3232: FN<32, 32> __VA_ARGS__; break;                   \
            case 6404: FN<64, 4> __VA_ARGS__; break;                    \
            case 6408: FN<64, 8> __VA_ARGS__; break;                    \
            case 6416: FN<64, 16> __VA_ARGS__; break;                   \
            default: FN<64, 32> __VA_ARGS__; break;                     \
        }                                                               \
    } while (0)

#define TW_DISPATCH_H(FN, H, ...)                                       \
    do {                                                                \
        switch (H) {                                                    \
            case 4: FN<4> __VA_ARGS__; break;                           \
            case 8: FN<8> __VA_ARGS__; break;                           \
            case 16: FN<16> __VA_ARGS__; break;                         \
            default: FN<32> __VA_ARGS__; break;                         \
        }                                                               \
    } while (0)

template <int K, int H>
static void launch_linear_fwd(const TowerK& kk, dim3 grid, hipStream_t st) {
    const size_t lds = TW_MFMA_OK(K, H) ? TW_THREADS * (H + 4) * sizeof(float)
                                        : (TW_THREADS * ((K > H ? K : H) + 4) + H * K + H) * sizeof(float);
    hipLaunchKernelGGL((tower_linear_fwd_kernel<K, H>), grid, dim3(TW_THREADS), lds, st, kk);
}
template <int H>
static void launch_head_fwd(const TowerK& kk, dim3 grid, hipStream_t st) {
    hipLaunchKernelGGL((tower_head_fwd_kernel<H>), grid, dim3(TW_THREADS), 0, st, kk);
}
template <int H>
static void launch_bwd_stats(const TowerK& kk, dim3 grid, hipStream_t st) {
    hipLaunchKernelGGL((tower_bwd_stats_kernel<H>), grid, dim3(TW_THREADS), TW_THREADS * (H + 5) * sizeof(float), st, kk);
}
template <int K, int H>
static void launch_bwd_apply(const TowerK& kk, dim3 grid, hipStream_t st) {
    const size_t lds = TW_MFMA_OK(H, K) ? (TW_THREADS / 64) * 16 * (K + 4) * sizeof(float)
                                        : (TW_THREADS * ((K > H ? K : H) + 4) + H * K + 7 * H) * sizeof(float);
    hipLaunchKernelGGL((tower_bwd_apply_kernel<K, H>), grid, dim3(TW_THREADS), lds, st, kk);
}

static int tower_common(const swr_tower_args* args) {
    SWR_REQUIRE(args != nullptr, SWR_ERR_ARG);
    const swr_tower_args& a = *args;
    SWR_REQUIRE(a.M > 0 && a.G > 0 && a.G <= 65535, SWR_ERR_ARG);
    SWR_REQUIRE(tower_shape_ok(a.K, a.H), SWR_ERR_UNSUPPORTED);
    SWR_REQUIRE(a.Z1 && a.ldz >= a.G * a.H && a.ldz % 4 == 0 && swr_aligned16(a.Z1), SWR_ERR_ARG);
    return SWR_OK;
}

extern "C" int swr_tower_fwd_linear(const swr_tower_args* args, void* stream) {
    int rc = tower_common(args);
    if (rc != SWR_OK) return rc;
    const swr_tower_args& a = *args;
    SWR_REQUIRE(a.X && a.W1 && a.ldx >= static_cast<int64_t>(a.G) * a.K, SWR_ERR_ARG);
    SWR_REQUIRE(a.ldx % 4 == 0 && swr_aligned16(a.X), SWR_ERR_ALIGN);
    TowerK kk;
    kk.a = a;
    kk.bn_partials = kk.head_partials = nullptr;
        const dim3 grid(static_cast<unsigned>(swr_ceil_div(a.M, TW_THREADS)), static_cast<unsigned>(a.G));
    TW_DISPATCH_KH(launch_linear_fwd, a.K, a.H, (kk, grid, static_cast<hipStream_t>(stream)));
    return swr_launch_status();
}

extern "C" int swr_tower_fwd_head(const swr_tower_args* args, void* stream) {
    int rc = tower_common(args);
    if (rc != SWR_OK) return rc;
    const swr_tower_args& a = *args;
    SWR_REQUIRE(a.scale && a.shift && a.w2 && a.V && a.ldv >= a.G, SWR_ERR_ARG);
    TowerK kk;
    kk.a = a;
    kk.bn_partials = kk.head_partials = nullptr;
    const dim3 grid(static_cast<unsigned>(swr_ceil_div(a.M, TW_THREADS)));
    TW_DISPATCH_H(launch_head_fwd, a.H, (kk, grid, static_cast<hipStream_t>(stream)));
    return swr_launch_status();
}

extern "C" size_t swr_tower_bwd_workspace_bytes(int64_t M, int G, int H) {
    if (M <= 0 || G <= 0 || H <= 0) return 0;
    const size_t tiles = static_cast<size_t>(swr_ceil_div(M, 64));
    return tiles * (static_cast<size_t>(G) * H * 3 + G) * sizeof(float) + 256;
}

extern "C" int swr_tower_bwd(const swr_tower_args* args, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = tower_common(args);
    if (rc != SWR_OK) return rc;
    swr_tower_args a = *args;
    SWR_REQUIRE(a.scale && a.shift && a.mean && a.rstd && a.w2 && a.W1, SWR_ERR_ARG);
    if (a.dV) {
        SWR_REQUIRE(a.lddv >= a.G, SWR_ERR_ARG);
    } else {                                                   // selected mode: dV implied by (p, y, domain, dloss)
        SWR_REQUIRE(a.sel_domain && a.sel_y && a.sel_p && a.sel_dloss, SWR_ERR_ARG);
        SWR_REQUIRE(swr_is_index_dtype(a.sel_dom_dtype), SWR_ERR_DTYPE);
        SWR_REQUIRE(swr_is_value_dtype(a.sel_y_dtype), SWR_ERR_DTYPE);
    }
    SWR_REQUIRE(a.ca && a.cb && a.cc && a.dZ1 && a.lddz >= a.G * a.H, SWR_ERR_ARG);
    SWR_REQUIRE(a.lddz % 4 == 0 && swr_aligned16(a.dZ1), SWR_ERR_ALIGN);
    SWR_REQUIRE(!a.dX || (a.lddx >= static_cast<int64_t>(a.G) * a.K && a.lddx % 4 == 0 && swr_aligned16(a.dX)), SWR_ERR_ALIGN);
    SWR_REQUIRE(workspace && workspace_bytes >= swr_tower_bwd_workspace_bytes(a.M, a.G, a.H), SWR_ERR_WORKSPACE);
    const int n_tiles = static_cast<int>(swr_ceil_div(a.M, 64));
    const int N = a.G * a.H;
    TowerK kk;
    kk.a = a;
    kk.bn_partials = static_cast<float*>(workspace);
    kk.head_partials = kk.bn_partials + static_cast<size_t>(n_tiles) * N * 2;
    hipStream_t st = static_cast<hipStream_t>(stream);
        const dim3 grid(static_cast<unsigned>(swr_ceil_div(a.M, TW_THREADS)), static_cast<unsigned>(a.G));
    TW_DISPATCH_H(launch_bwd_stats, a.H, (kk, grid, st));
    hipLaunchKernelGGL(tower_bwd_finalize_kernel, dim3(static_cast<unsigned>(N + a.G)), dim3(TW_THREADS), 0, st, kk, n_tiles);
    TW_DISPATCH_KH(launch_bwd_apply, a.K, a.H, (kk, grid, st));
    return swr_launch_status();
}
