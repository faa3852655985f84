// LayerNorm (+ReLU) of G side-by-side column groups, forward and backward, and the domain block select -- the pieces of
// M3oE (reference models/multi_domain/m3oe.py:49-67 `Mlp_N` = [Linear, LayerNorm, ReLU]; :121-128 towers; :141-146
// `emb = where(mask_d, output_d, emb)`) that the BatchNorm-based families do not have.
//
// Layout: X [M, G*N] row-major (ld), group g = columns [g*N, (g+1)*N) with its own gamma / beta (vectors of length G*N,
// the G LayerNorm modules' parameters back to back).  A (row, group) segment is normalised by LPS consecutive lanes
// (LPS = 8..64, a power of two): 16-byte loads, the segment stays in registers, mean / variance by butterfly shuffles
// (two-pass: mean, then sum (x - mean)^2, fp32, biased variance, eps inside the sqrt -- torch.nn.LayerNorm).
// HBM-bound: 4 N bytes read + 4 N written per segment forward; backward reads X, dY and writes dX.
// The parameter gradients are column sums over all rows: every workgroup walks a fixed row range and leaves one
// partial per column; a second launch adds the partials in workgroup order (deterministic, no atomics).
#include <algorithm>

#include "common.h"

#define LN_THREADS 256
#define LN_MAXC 16            // register-resident chunks per lane, scalar kernels; the 16-byte kernels keep LN_MAXC_V chunks of 4
#define LN_MAXC_V 4            // (16 x 4 values x 4 arrays in the backward kernel = 256 registers: 528 bytes per lane went to scratch)

struct LnK {
    int64_t M;
    int G, N, relu, lps, rows_per_block, n_blocks_x, accumulate;
    float eps;
    const float* X; int64_t ldx;
    const float* gamma; const float* beta;
    float* Y; int64_t ldy;
    float* mean; float* rstd;              // [M, G]
    const float* dY; int64_t lddy;
    float* dX; int64_t lddx;
    float* partials;                       // [n_blocks_x][2][G*N]
    float* dgamma; float* dbeta;
};

template <int LPS_MAX>
__device__ __forceinline__ float seg_sum(float v, int lps) {
#pragma unroll
    for (int o = 1; o < LPS_MAX; o <<= 1)
        if (o < lps) v += __shfl_xor(v, o, 64);
    return v;
}

template <int VEC, int MAXC>
__global__ __launch_bounds__(LN_THREADS) void layernorm_fwd_kernel(const LnK k) {
    const int lps = k.lps, slots = LN_THREADS / lps;
    const int slot = threadIdx.x / lps, j0 = threadIdx.x % lps;
    const int g = blockIdx.y;
    const int C = k.N / VEC;
    const float inv_n = 1.f / static_cast<float>(k.N);
    const int64_t r0 = static_cast<int64_t>(blockIdx.x) * k.rows_per_block;
    const int64_t r1 = min(r0 + k.rows_per_block, k.M);
    for (int64_t rb = r0; rb < r1; rb += slots) {
        const int64_t r = rb + slot;
        const bool live = r < r1;
        float x[MAXC][VEC];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int j = j0 + c * lps;
            if (live && j < C) {
                const float* p = k.X + r * k.ldx + static_cast<int64_t>(g) * k.N + j * VEC;
                if (VEC == 4) {
                    const float4 t = *reinterpret_cast<const float4*>(p);
                    x[c][0] = t.x; x[c][1 % VEC] = t.y; x[c][2 % VEC] = t.z; x[c][3 % VEC] = t.w;
                } else {
                    x[c][0] = p[0];
                }
#pragma unroll
                for (int v = 0; v < VEC; ++v) s += x[c][v];
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) x[c][v] = 0.f;
            }
        }
        const float mean = seg_sum<64>(s, lps) * inv_n;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int j = j0 + c * lps;
            if (live && j < C) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) { const float d = x[c][v] - mean; q += d * d; }
            }
        }
        const float rstd = rsqrtf(seg_sum<64>(q, lps) * inv_n + k.eps);
        if (!live) continue;
        if (j0 == 0 && k.mean) { k.mean[r * k.G + g] = mean; k.rstd[r * k.G + g] = rstd; }
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int j = j0 + c * lps;
            if (j < C) {
                const int col = g * k.N + j * VEC;
                float y[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    y[v] = (x[c][v] - mean) * rstd * k.gamma[col + v] + k.beta[col + v];
                    if (k.relu) y[v] = fmaxf(y[v], 0.f);
                }
                float* o = k.Y + r * k.ldy + col;
                if (VEC == 4) *reinterpret_cast<float4*>(o) = make_float4(y[0], y[1 % VEC], y[2 % VEC], y[3 % VEC]);
                else o[0] = y[0];
            }
        }
    }
}

template <int VEC, int MAXC>
__global__ __launch_bounds__(LN_THREADS) void layernorm_bwd_kernel(const LnK k) {
    __shared__ float red[2][LN_THREADS * 4];          // VEC <= 4 values per (thread, chunk); chunk loop outside
    const int lps = k.lps, slots = LN_THREADS / lps;
    const int slot = threadIdx.x / lps, j0 = threadIdx.x % lps;
    const int g = blockIdx.y;
    const int C = k.N / VEC;
    const float inv_n = 1.f / static_cast<float>(k.N);
    const int64_t r0 = static_cast<int64_t>(blockIdx.x) * k.rows_per_block;
    const int64_t r1 = min(r0 + k.rows_per_block, k.M);
    float ag[MAXC][VEC], ab[MAXC][VEC];        // this thread's column sums over the rows of its slot
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
#pragma unroll
        for (int v = 0; v < VEC; ++v) { ag[c][v] = 0.f; ab[c][v] = 0.f; }
    for (int64_t rb = r0; rb < r1; rb += slots) {
        const int64_t r = rb + slot;
        const bool live = r < r1;
        const float mean = live ? k.mean[r * k.G + g] : 0.f;
        const float rstd = live ? k.rstd[r * k.G + g] : 0.f;
        float xh[MAXC][VEC], dh[MAXC][VEC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int j = j0 + c * lps;
#pragma unroll
            for (int v = 0; v < VEC; ++v) { xh[c][v] = 0.f; dh[c][v] = 0.f; }
            if (live && j < C) {
                const int col = g * k.N + j * VEC;
                float xv[VEC], dy[VEC];
                const float* px = k.X + r * k.ldx + col;
                const float* pd = k.dY + r * k.lddy + col;
                if (VEC == 4) {
                    const float4 a = *reinterpret_cast<const float4*>(px), b = *reinterpret_cast<const float4*>(pd);
                    xv[0] = a.x; xv[1 % VEC] = a.y; xv[2 % VEC] = a.z; xv[3 % VEC] = a.w;
                    dy[0] = b.x; dy[1 % VEC] = b.y; dy[2 % VEC] = b.z; dy[3 % VEC] = b.w;
                } else {
                    xv[0] = px[0]; dy[0] = pd[0];
                }
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const float gm = k.gamma[col + v];
                    const float h = (xv[v] - mean) * rstd;
                    float d = dy[v];
                    if (k.relu && h * gm + k.beta[col + v] <= 0.f) d = 0.f;       // ReLU'(y): the forward's y recomputed
                    ag[c][v] += d * h;
                    ab[c][v] += d;
                    xh[c][v] = h;
                    dh[c][v] = d * gm;
                    s1 += dh[c][v];
                    s2 += dh[c][v] * h;
                }
            }
        }
        s1 = seg_sum<64>(s1, lps) * inv_n;
        s2 = seg_sum<64>(s2, lps) * inv_n;
        if (!live || !k.dX) continue;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int j = j0 + c * lps;
            if (j < C) {
                float o[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) o[v] = rstd * (dh[c][v] - s1 - xh[c][v] * s2);
                float* p = k.dX + r * k.lddx + g * k.N + j * VEC;
                if (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1 % VEC], o[2 % VEC], o[3 % VEC]);
                else p[0] = o[0];
            }
        }
    }
    // column partials of this workgroup: slots added in slot order through LDS, one chunk index at a time
    float* part = k.partials + static_cast<int64_t>(blockIdx.x) * 2 * k.G * k.N;
    for (int c = 0; c < MAXC; ++c) {
        if (c * lps >= C) break;                               // uniform
        __syncthreads();
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            red[0][threadIdx.x * VEC + v] = ag[c][v];
            red[1][threadIdx.x * VEC + v] = ab[c][v];
        }
        __syncthreads();
        if (slot == 0) {
            const int j = j0 + c * lps;
            if (j < C) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    float sg = 0.f, sb = 0.f;
                    for (int q = 0; q < slots; ++q) {
                        sg += red[0][(q * lps + j0) * VEC + v];
                        sb += red[1][(q * lps + j0) * VEC + v];
                    }
                    const int col = g * k.N + j * VEC + v;
                    part[col] = sg;
                    part[k.G * k.N + col] = sb;
                }
            }
        }
    }
}

__global__ __launch_bounds__(LN_THREADS) void layernorm_param_reduce_kernel(const LnK k) {
    const int col = blockIdx.x * LN_THREADS + threadIdx.x;
    const int W = k.G * k.N;
    if (col >= W) return;
    float sg = 0.f, sb = 0.f;
    for (int b = 0; b < k.n_blocks_x; ++b) {                   // fixed order
        sg += k.partials[(static_cast<int64_t>(b) * 2) * W + col];
        sb += k.partials[(static_cast<int64_t>(b) * 2 + 1) * W + col];
    }
    if (k.dgamma) k.dgamma[col] = k.accumulate ? k.dgamma[col] + sg : sg;
    if (k.dbeta) k.dbeta[col] = k.accumulate ? k.dbeta[col] + sb : sb;
}

static int ln_plan(LnK& k, bool vec) {
    const int V = vec ? 4 : 1, maxc = vec ? LN_MAXC_V : LN_MAXC;
    const int C = k.N / V;
    int lps = 8;
    while (lps < 64 && lps * maxc < C) lps <<= 1;
    while (lps < 64 && lps * 2 <= C) lps <<= 1;               // short segments: as many lanes as there are chunks
    if (static_cast<int64_t>(lps) * maxc < C) return SWR_ERR_UNSUPPORTED;
    k.lps = lps;
    const int slots = LN_THREADS / lps;
    int64_t blocks = swr_ceil_div(k.M, slots);
    const int64_t cap = std::max<int64_t>(1, 2048 / k.G);
    if (blocks > cap) blocks = cap;
    k.rows_per_block = static_cast<int>(swr_ceil_div(swr_ceil_div(k.M, blocks), slots) * slots);
    k.n_blocks_x = static_cast<int>(swr_ceil_div(k.M, k.rows_per_block));
    return SWR_OK;
}

static bool ln_vec(const swr_layernorm_args* a, bool bwd) {
    bool v = a->N % 4 == 0 && a->ldx % 4 == 0 && swr_aligned16(a->X);
    if (!bwd) v = v && a->ldy % 4 == 0 && swr_aligned16(a->Y);
    else v = v && a->lddy % 4 == 0 && swr_aligned16(a->dY) && (!a->dX || (a->lddx % 4 == 0 && swr_aligned16(a->dX)));
    return v;
}

static void ln_fill(LnK& k, const swr_layernorm_args* a) {
    k.M = a->M; k.G = a->G; k.N = a->N; k.relu = a->relu; k.eps = a->eps; k.accumulate = a->accumulate;
    k.X = a->X; k.ldx = a->ldx; k.gamma = a->gamma; k.beta = a->beta; k.Y = a->Y; k.ldy = a->ldy;
    k.mean = a->mean; k.rstd = a->rstd; k.dY = a->dY; k.lddy = a->lddy; k.dX = a->dX; k.lddx = a->lddx;
    k.dgamma = a->dgamma; k.dbeta = a->dbeta; k.partials = nullptr;
}

extern "C" int swr_layernorm_fwd(const swr_layernorm_args* a, void* stream) {
    SWR_REQUIRE(a && a->M >= 0 && a->G > 0 && a->N > 0 && a->X && a->Y && a->gamma && a->beta, SWR_ERR_ARG);
    SWR_REQUIRE(a->ldx >= static_cast<int64_t>(a->G) * a->N && a->ldy >= static_cast<int64_t>(a->G) * a->N, SWR_ERR_ARG);
    SWR_REQUIRE((a->mean == nullptr) == (a->rstd == nullptr) && a->G <= 65535, SWR_ERR_ARG);
    if (a->M == 0) return SWR_OK;
    LnK k;
    ln_fill(k, a);
    const bool vec = ln_vec(a, false);
    int rc = ln_plan(k, vec);
    if (rc != SWR_OK) return rc;
    const dim3 grid(static_cast<unsigned>(k.n_blocks_x), static_cast<unsigned>(k.G));
    if (vec) hipLaunchKernelGGL((layernorm_fwd_kernel<4, LN_MAXC_V>), grid, dim3(LN_THREADS), 0, static_cast<hipStream_t>(stream), k);
    else hipLaunchKernelGGL((layernorm_fwd_kernel<1, LN_MAXC>), grid, dim3(LN_THREADS), 0, static_cast<hipStream_t>(stream), k);
    return swr_launch_status();
}

extern "C" size_t swr_layernorm_bwd_workspace_bytes(int64_t M, int G, int N) {
    if (M <= 0 || G <= 0 || N <= 0) return 256;
    LnK k;
    k.M = M; k.G = G; k.N = N;
    if (ln_plan(k, N % 4 == 0) != SWR_OK && ln_plan(k, false) != SWR_OK) return 0;
    // the scalar plan never has fewer workgroups than the vector one: size for the larger
    LnK s;
    s.M = M; s.G = G; s.N = N;
    int nb = k.n_blocks_x;
    if (ln_plan(s, false) == SWR_OK && s.n_blocks_x > nb) nb = s.n_blocks_x;
    return static_cast<size_t>(nb) * 2 * G * N * sizeof(float) + 256;
}

extern "C" int swr_layernorm_bwd(const swr_layernorm_args* a, void* workspace, size_t workspace_bytes, void* stream) {
    SWR_REQUIRE(a && a->M >= 0 && a->G > 0 && a->N > 0 && a->X && a->dY && a->gamma && a->beta && a->mean && a->rstd && workspace,
                SWR_ERR_ARG);
    SWR_REQUIRE(a->G <= 65535, SWR_ERR_ARG);
    if (a->M == 0) return SWR_OK;
    LnK k;
    ln_fill(k, a);
    const bool vec = ln_vec(a, true);
    int rc = ln_plan(k, vec);
    if (rc != SWR_OK) return rc;
    SWR_REQUIRE(workspace_bytes >= static_cast<size_t>(k.n_blocks_x) * 2 * k.G * k.N * sizeof(float), SWR_ERR_WORKSPACE);
    k.partials = static_cast<float*>(workspace);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(static_cast<unsigned>(k.n_blocks_x), static_cast<unsigned>(k.G));
    if (vec) hipLaunchKernelGGL((layernorm_bwd_kernel<4, LN_MAXC_V>), grid, dim3(LN_THREADS), 0, st, k);
    else hipLaunchKernelGGL((layernorm_bwd_kernel<1, LN_MAXC>), grid, dim3(LN_THREADS), 0, st, k);
    if (a->dgamma || a->dbeta)
        hipLaunchKernelGGL(layernorm_param_reduce_kernel, dim3(static_cast<unsigned>(swr_ceil_div(static_cast<int64_t>(k.G) * k.N, LN_THREADS))),
                           dim3(LN_THREADS), 0, st, k);
    return swr_launch_status();
}

// ---- domain block select: out[b, h] = V[b, d_b * H + h] for 0 <= d_b < D, else 0 (exact integer compare on the raw id)
__global__ __launch_bounds__(LN_THREADS) void block_select_fwd_kernel(const float* __restrict__ V, int64_t ldv, const void* dom,
                                                                      int dom_dtype, int D, int H, int64_t M,
                                                                      float* __restrict__ out, int64_t ldo) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * LN_THREADS + threadIdx.x;
    if (i >= M * H) return;
    const int64_t b = i / H;
    const int h = static_cast<int>(i - b * H);
    const int64_t d = swr_load_index(dom, dom_dtype, b);
    out[b * ldo + h] = (d >= 0 && d < D) ? V[b * ldv + d * H + h] : 0.f;
}

__global__ __launch_bounds__(LN_THREADS) void block_select_bwd_kernel(const float* __restrict__ dout, int64_t ldo, const void* dom,
                                                                      int dom_dtype, int D, int H, int64_t M,
                                                                      float* __restrict__ dV, int64_t ldv) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * LN_THREADS + threadIdx.x;
    if (i >= M * D * H) return;
    const int64_t b = i / (D * H);
    const int c = static_cast<int>(i - b * D * H);
    const int64_t d = swr_load_index(dom, dom_dtype, b);
    dV[b * ldv + c] = (c / H == d) ? dout[b * ldo + c % H] : 0.f;
}

extern "C" int swr_block_select_fwd(const float* V, int64_t ldv, const void* domain, int dom_dtype, int D, int H, int64_t M,
                                    float* out, int64_t ldo, void* stream) {
    SWR_REQUIRE(V && domain && out && D > 0 && H > 0 && M >= 0 && ldv >= static_cast<int64_t>(D) * H && ldo >= H, SWR_ERR_ARG);
    SWR_REQUIRE(swr_is_index_dtype(dom_dtype), SWR_ERR_DTYPE);
    if (M == 0) return SWR_OK;
    hipLaunchKernelGGL(block_select_fwd_kernel, dim3(static_cast<unsigned>(swr_ceil_div(M * H, LN_THREADS))), dim3(LN_THREADS), 0,
                       static_cast<hipStream_t>(stream), V, ldv, domain, dom_dtype, D, H, M, out, ldo);
    return swr_launch_status();
}

extern "C" int swr_block_select_bwd(const float* dout, int64_t ldo, const void* domain, int dom_dtype, int D, int H, int64_t M,
                                    float* dV, int64_t ldv, void* stream) {
    SWR_REQUIRE(dout && domain && dV && D > 0 && H > 0 && M >= 0 && ldv >= static_cast<int64_t>(D) * H && ldo >= H, SWR_ERR_ARG);
    SWR_REQUIRE(swr_is_index_dtype(dom_dtype), SWR_ERR_DTYPE);
    if (M == 0) return SWR_OK;
    hipLaunchKernelGGL(block_select_bwd_kernel, dim3(static_cast<unsigned>(swr_ceil_div(M * D * H, LN_THREADS))), dim3(LN_THREADS), 0,
                       static_cast<hipStream_t>(stream), dout, ldo, domain, dom_dtype, D, H, M, dV, ldv);
    return swr_launch_status();
}
