// Gate mixing, domain select, BCE and small elementwise helpers.
// All kernels are streaming passes over [M, *] fp32 row-major data (HBM-bound); every reduction that
// crosses workgroups goes through per-block partials summed in a fixed order (deterministic).
#include <algorithm>

#include "common.h"
#include "adam_advance.h"

#define EW_THREADS 256

// ------------------------------------------------------------------------------------- gate mixing
// expert_pooling = sum_j gate_j * expert_j   (mmoe.py:48-49, ple.py:121-126,131-133)
//
// One wave per sample row: the row of activated experts + gate probabilities (and, backward, the row of
// pooled gradients) is staged once in LDS with coalesced loads, every output of the row is then computed
// from LDS and written with coalesced stores.  HBM traffic = each input row read once, each output row
// written once.
#define MIX_WAVES 4
#define MIX_MAX_EXPERTS 64
#define MIX_ROW_FLOATS 2560          // upper bound of LDS floats per wave: Y row + dP row

struct MixK {
    swr_mix_desc d;
    int32_t n_expert;
    int32_t y_lo, y_hi;              // column span of Y that the kernel touches
    int32_t row_floats;              // LDS floats per wave (rounded up to 4)
    uint8_t inv_cnt[MIX_MAX_EXPERTS];            // backward: (o, j) pairs that select expert e
    uint8_t inv[MIX_MAX_EXPERTS][SWR_MIX_MAX_OUT];   // packed o * 16 + j
};

static int make_mix(const swr_mix_desc* desc, MixK& k) {
    SWR_REQUIRE(desc->n_out > 0 && desc->n_out <= SWR_MIX_MAX_OUT && desc->n_sel > 0 && desc->n_sel <= SWR_MIX_MAX_SEL &&
                    desc->H > 0 && desc->x_col >= 0 && desc->g_col >= 0 && desc->g_stride >= desc->n_sel, SWR_ERR_ARG);
    k.d = *desc;
    k.n_expert = 0;
    for (int e = 0; e < MIX_MAX_EXPERTS; ++e) k.inv_cnt[e] = 0;
    for (int o = 0; o < desc->n_out; ++o)
        for (int j = 0; j < desc->n_sel; ++j) {
            const int e = desc->sel[o][j];
            SWR_REQUIRE(e < MIX_MAX_EXPERTS, SWR_ERR_UNSUPPORTED);
            if (e + 1 > k.n_expert) k.n_expert = e + 1;
            SWR_REQUIRE(k.inv_cnt[e] < SWR_MIX_MAX_OUT, SWR_ERR_UNSUPPORTED);
            k.inv[e][k.inv_cnt[e]++] = static_cast<uint8_t>(o * 16 + j);
        }
    const int x_hi = desc->x_col + k.n_expert * desc->H;
    const int g_hi = desc->g_col + (desc->n_out - 1) * desc->g_stride + desc->n_sel;
    k.y_lo = desc->x_col < desc->g_col ? desc->x_col : desc->g_col;
    k.y_hi = x_hi > g_hi ? x_hi : g_hi;
    SWR_REQUIRE((k.y_hi - k.y_lo) + desc->n_out * desc->H <= MIX_ROW_FLOATS, SWR_ERR_UNSUPPORTED);
    k.row_floats = ((k.y_hi - k.y_lo) + desc->n_out * desc->H + 3) / 4 * 4;
    return SWR_OK;
}

__global__ __launch_bounds__(MIX_WAVES * 64) void mix_fwd_kernel(const MixK k, const float* __restrict__ Y, int64_t ldy,
                                                                 float* __restrict__ P, int64_t ldp, int64_t M) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* row = lds + wave * k.row_floats;
    const swr_mix_desc& d = k.d;
    const int wy = k.y_hi - k.y_lo, wp = d.n_out * d.H;
    for (int64_t m = static_cast<int64_t>(blockIdx.x) * MIX_WAVES + wave; m < M; m += static_cast<int64_t>(gridDim.x) * MIX_WAVES) {
        const float* y = Y + m * ldy + k.y_lo;
        for (int c = lane; c < wy; c += 64) row[c] = y[c];
        __builtin_amdgcn_wave_barrier();
        const float* x = row + (d.x_col - k.y_lo);
        const float* g = row + (d.g_col - k.y_lo);
        for (int c = lane; c < wp; c += 64) {
            const int o = c / d.H, h = c - o * d.H;
            float acc = 0.f;
            for (int j = 0; j < d.n_sel; ++j) acc = fmaf(g[o * d.g_stride + j], x[d.sel[o][j] * d.H + h], acc);
            P[m * ldp + c] = acc;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- 16-byte fast path (H, x_col, row strides multiples of 4 floats): thread = 4 pooled outputs of one row; the expert
// and gate rows are read straight from global memory (a row is 0.6 KB: the re-reads by the other outputs of the row
// hit the CU's L1), no LDS, no barriers -> every load of a workgroup is in flight at once.
// (the gate probabilities come from Gt / ldg: Y itself, or a tensor of their own -- PLE's gates stay in the first layer's
// output while the experts run their second layer, ple.py:107-125; no concatenation pass)
__global__ __launch_bounds__(256) void mix_fwd_v4_kernel(const MixK k, const float* __restrict__ Y, int64_t ldy,
                                                         const float* __restrict__ Gt, int64_t ldg,
                                                         float* __restrict__ P, int64_t ldp, int64_t M) {
    const swr_mix_desc& d = k.d;
    const int h4n = d.H >> 2, per_row = d.n_out * h4n;
    const int64_t item = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t m = item / per_row;
    if (m >= M) return;
    const int q = static_cast<int>(item - m * per_row);
    const int o = q / h4n, h = (q - o * h4n) * 4;
    const float* __restrict__ y = Y + m * ldy;
    const float* __restrict__ g = Gt + m * ldg + d.g_col + o * d.g_stride;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < d.n_sel; ++j) {
        const float gj = g[j];
        const float4 x = *reinterpret_cast<const float4*>(y + d.x_col + d.sel[o][j] * d.H + h);
        acc.x = fmaf(gj, x.x, acc.x); acc.y = fmaf(gj, x.y, acc.y);
        acc.z = fmaf(gj, x.z, acc.z); acc.w = fmaf(gj, x.w, acc.w);
    }
    *reinterpret_cast<float4*>(P + m * ldp + o * d.H + h) = acc;
}

extern "C" int swr_moe_mix_fwd(const swr_mix_desc* desc, const float* Y, int64_t ldy, const float* G, int64_t ldg, float* P,
                               int64_t ldp, int64_t M, void* stream) {
    SWR_REQUIRE(desc && Y && P && M >= 0, SWR_ERR_ARG);
    MixK k;
    const int rc = make_mix(desc, k);
    if (rc != SWR_OK) return rc;
    if (M == 0) return SWR_OK;
    if (desc->H % 4 == 0 && desc->x_col % 4 == 0 && ldy % 4 == 0 && ldp % 4 == 0 && swr_aligned16(Y) && swr_aligned16(P)) {
        const int64_t items = M * desc->n_out * (desc->H / 4);
        hipLaunchKernelGGL(mix_fwd_v4_kernel, dim3(static_cast<unsigned>(swr_ceil_div(items, 256))), dim3(256), 0,
                           static_cast<hipStream_t>(stream), k, Y, ldy, G ? G : Y, G ? ldg : ldy, P, ldp, M);
        return swr_launch_status();
    }
    SWR_REQUIRE(G == nullptr, SWR_ERR_UNSUPPORTED);          // gates in a tensor of their own: the 16-byte path only
    const unsigned grid = static_cast<unsigned>(swr_ceil_div(M, MIX_WAVES) < 8192 ? swr_ceil_div(M, MIX_WAVES) : 8192);
    hipLaunchKernelGGL(mix_fwd_kernel, dim3(grid), dim3(MIX_WAVES * 64), MIX_WAVES * k.row_floats * sizeof(float),
                       static_cast<hipStream_t>(stream), k, Y, ldy, P, ldp, M);
    return swr_launch_status();
}

// dX[e, h] = sum over (o, j) with sel[o][j] == e of gate[o][j] * dP[o, h];  dG[o][j] = sum_h dP[o, h] X[sel[o][j], h]
__global__ __launch_bounds__(MIX_WAVES * 64) void mix_bwd_kernel(const MixK k, const float* __restrict__ dP, int64_t lddp,
                                                                 const float* __restrict__ Y, int64_t ldy,
                                                                 float* __restrict__ dY, int64_t lddy, int accumulate, int64_t M) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* row = lds + wave * k.row_floats;
    const swr_mix_desc& d = k.d;
    const int wy = k.y_hi - k.y_lo, wp = d.n_out * d.H;
    float* rdp = row + wy;
    const int wx = k.n_expert * d.H, ng = d.n_out * d.n_sel;
    for (int64_t m = static_cast<int64_t>(blockIdx.x) * MIX_WAVES + wave; m < M; m += static_cast<int64_t>(gridDim.x) * MIX_WAVES) {
        const float* y = Y + m * ldy + k.y_lo;
        const float* dp = dP + m * lddp;
        for (int c = lane; c < wy; c += 64) row[c] = y[c];
        for (int c = lane; c < wp; c += 64) rdp[c] = dp[c];
        __builtin_amdgcn_wave_barrier();
        const float* x = row + (d.x_col - k.y_lo);
        const float* g = row + (d.g_col - k.y_lo);
        float* out = dY + m * lddy;
        for (int c = lane; c < wx; c += 64) {
            const int e = c / d.H, h = c - e * d.H;
            float acc = 0.f;
            for (int q = 0; q < k.inv_cnt[e]; ++q) {
                const int o = k.inv[e][q] >> 4, j = k.inv[e][q] & 15;
                acc = fmaf(g[o * d.g_stride + j], rdp[o * d.H + h], acc);
            }
            float* dst = out + d.x_col + c;
            *dst = accumulate ? *dst + acc : acc;
        }
        // gate gradients: 4 lanes share one (o, j) dot product over h
        for (int q0 = 0; q0 < ng; q0 += 16) {
            const int q = q0 + (lane >> 2), part = lane & 3;
            float acc = 0.f;
            if (q < ng) {
                const int o = q / d.n_sel, j = q - o * d.n_sel;
                const float* xe = x + d.sel[o][j] * d.H;
                for (int h = part; h < d.H; h += 4) acc = fmaf(rdp[o * d.H + h], xe[h], acc);
            }
            acc += __shfl_xor(acc, 1);
            acc += __shfl_xor(acc, 2);
            if (q < ng && part == 0) {
                const int o = q / d.n_sel, j = q - o * d.n_sel;
                float* dst = out + d.g_col + o * d.g_stride + j;
                *dst = accumulate ? *dst + acc : acc;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// 16-byte fast path of the backward: a workgroup takes MIXB_ROWS rows; phase 1: thread = 4 expert-gradient columns of
// one row (sum over the outputs that select the expert), phase 2: thread = one gate gradient (dot product over H).
// Everything is read from global memory / L1; no LDS, no barriers.
#define MIXB_ROWS 32
__global__ __launch_bounds__(256) void mix_bwd_v4_kernel(const MixK k, const float* __restrict__ dP, int64_t lddp,
                                                         const float* __restrict__ Y, int64_t ldy,
                                                         const float* __restrict__ Gt, int64_t ldg,
                                                         float* __restrict__ dY, int64_t lddy, float* __restrict__ dGt, int64_t lddg,
                                                         int accumulate, int64_t M, int rows_per_wg) {
    const swr_mix_desc& d = k.d;
    const int64_t m0 = static_cast<int64_t>(blockIdx.x) * rows_per_wg;
    const int rows = static_cast<int>(min<int64_t>(rows_per_wg, M - m0));
    const int h4n = d.H >> 2;
    const int xper = k.n_expert * h4n;                       // expert-gradient items per row
    for (int it = threadIdx.x; it < rows * xper; it += 256) {
        const int r = it / xper, q = it - r * xper;
        const int e = q / h4n, h = (q - e * h4n) * 4;
        const int64_t m = m0 + r;
        const float* __restrict__ g = Gt + m * ldg + d.g_col;
        const float* __restrict__ dp = dP + m * lddp + h;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int c = 0; c < k.inv_cnt[e]; ++c) {
            const int o = k.inv[e][c] >> 4, j = k.inv[e][c] & 15;
            const float gj = g[o * d.g_stride + j];
            const float4 v = *reinterpret_cast<const float4*>(dp + o * d.H);
            acc.x = fmaf(gj, v.x, acc.x); acc.y = fmaf(gj, v.y, acc.y);
            acc.z = fmaf(gj, v.z, acc.z); acc.w = fmaf(gj, v.w, acc.w);
        }
        float4* dst = reinterpret_cast<float4*>(dY + m * lddy + d.x_col + e * d.H + h);
        if (accumulate) {
            const float4 old = *dst;
            acc.x += old.x; acc.y += old.y; acc.z += old.z; acc.w += old.w;
        }
        *dst = acc;
    }
    const int ng = d.n_out * d.n_sel;
    for (int it = threadIdx.x; it < rows * ng; it += 256) {
        const int r = it / ng, q = it - r * ng;
        const int o = q / d.n_sel, j = q - o * d.n_sel;
        const int64_t m = m0 + r;
        const float* __restrict__ x = Y + m * ldy + d.x_col + d.sel[o][j] * d.H;
        const float* __restrict__ dp = dP + m * lddp + o * d.H;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;       // four interleaved chains, combined in a fixed order
        for (int h = 0; h < d.H; h += 4) {
            const float4 u = *reinterpret_cast<const float4*>(dp + h);
            const float4 v = *reinterpret_cast<const float4*>(x + h);
            a0 = fmaf(u.x, v.x, a0); a1 = fmaf(u.y, v.y, a1);
            a2 = fmaf(u.z, v.z, a2); a3 = fmaf(u.w, v.w, a3);
        }
        const float acc = (a0 + a1) + (a2 + a3);
        float* dst = dGt + m * lddg + d.g_col + o * d.g_stride + j;
        *dst = accumulate ? *dst + acc : acc;
    }
}

// Identity selection (every output mixes experts 0..n_sel-1 in order: MMoE) with H <= 256: thread = 4 columns h of one
// row for ALL outputs and experts.  The expert rows stay in registers, each dP row is read once, the expert gradients
// are complete in-thread, and the gate gradients are finished by a fixed butterfly over the H/4 lanes of the row.
// Every byte of Y and dP is read exactly once.
template <int NS>
__global__ __launch_bounds__(256) void mix_bwd_ident_kernel(const MixK k, const float* __restrict__ dP, int64_t lddp,
                                                            const float* __restrict__ Y, int64_t ldy,
                                                            const float* __restrict__ Gt, int64_t ldg,
                                                            float* __restrict__ dY, int64_t lddy, float* __restrict__ dGt, int64_t lddg,
                                                            int accumulate, int64_t M) {
    const swr_mix_desc& d = k.d;
    const int h4n = d.H >> 2;                                // lanes per row: a power of two <= 64
    const int64_t item = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t m = min(item / h4n, M - 1);                // surplus lanes redo the last row (shuffles need them)
    const bool live = item / h4n < M;
    const int h = static_cast<int>(item % h4n) * 4;
    const float* __restrict__ y = Y + m * ldy;
    float4 x[NS], dx[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        x[j] = j < d.n_sel ? *reinterpret_cast<const float4*>(y + d.x_col + j * d.H + h) : make_float4(0.f, 0.f, 0.f, 0.f);
        dx[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float* __restrict__ out = dY + m * lddy;
    for (int o = 0; o < d.n_out; ++o) {
        const float4 v = *reinterpret_cast<const float4*>(dP + m * lddp + o * d.H + h);
        const float* __restrict__ g = Gt + m * ldg + d.g_col + o * d.g_stride;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            if (j < d.n_sel) {
                const float gj = g[j];
                dx[j].x = fmaf(gj, v.x, dx[j].x); dx[j].y = fmaf(gj, v.y, dx[j].y);
                dx[j].z = fmaf(gj, v.z, dx[j].z); dx[j].w = fmaf(gj, v.w, dx[j].w);
                float pd = (v.x * x[j].x + v.y * x[j].y) + (v.z * x[j].z + v.w * x[j].w);
                for (int off = 1; off < h4n; off <<= 1) pd += __shfl_xor(pd, off);
                if (live && h == 0) {
                    float* dst = dGt + m * lddg + d.g_col + o * d.g_stride + j;
                    *dst = accumulate ? *dst + pd : pd;
                }
            }
        }
    }
    if (!live) return;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        if (j < d.n_sel) {
            float4* dst = reinterpret_cast<float4*>(out + d.x_col + j * d.H + h);
            float4 a = dx[j];
            if (accumulate) {
                const float4 old = *dst;
                a.x += old.x; a.y += old.y; a.z += old.z; a.w += old.w;
            }
            *dst = a;
        }
    }
}

extern "C" int swr_moe_mix_bwd(const swr_mix_desc* desc, const float* dP, int64_t lddp, const float* Y, int64_t ldy,
                               const float* G, int64_t ldg, float* dY, int64_t lddy, float* dG, int64_t lddg, int accumulate,
                               int64_t M, void* stream) {
    SWR_REQUIRE(desc && dP && Y && dY && M >= 0 && (G == nullptr) == (dG == nullptr), SWR_ERR_ARG);
    const float* Gt = G ? G : Y;
    float* dGt = dG ? dG : dY;
    if (!G) { ldg = ldy; lddg = lddy; }
    MixK k;
    const int rc = make_mix(desc, k);
    if (rc != SWR_OK) return rc;
    if (M == 0) return SWR_OK;
    if (desc->H % 4 == 0 && desc->x_col % 4 == 0 && ldy % 4 == 0 && lddp % 4 == 0 && lddy % 4 == 0 && swr_aligned16(Y) &&
        swr_aligned16(dP) && swr_aligned16(dY)) {
        bool ident = desc->n_sel <= 8 && (desc->H & (desc->H - 1)) == 0 && desc->H <= 256 && k.n_expert == desc->n_sel;
        for (int o = 0; o < desc->n_out && ident; ++o)
            for (int j = 0; j < desc->n_sel; ++j) ident = ident && desc->sel[o][j] == j;
        if (ident) {
            const int64_t items = M * (desc->H / 4);
            const dim3 grid(static_cast<unsigned>(swr_ceil_div(items, 256)));
            if (desc->n_sel <= 4)
                hipLaunchKernelGGL(mix_bwd_ident_kernel<4>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), k, dP, lddp,
                                   Y, ldy, Gt, ldg, dY, lddy, dGt, lddg, accumulate, M);
            else
                hipLaunchKernelGGL(mix_bwd_ident_kernel<8>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), k, dP, lddp,
                                   Y, ldy, Gt, ldg, dY, lddy, dGt, lddg, accumulate, M);
            return swr_launch_status();
        }
        // rows per workgroup: 32, fewer for short batches (no state is shared between rows; at M = 8192 256 workgroups walked
        // 9 dependent-load iterations each for 18.5 us -- PLE, config 4)
        int rpw = MIXB_ROWS;
        while (rpw > 4 && M / rpw < 1024) rpw >>= 1;
        hipLaunchKernelGGL(mix_bwd_v4_kernel, dim3(static_cast<unsigned>(swr_ceil_div(M, rpw))), dim3(256), 0,
                           static_cast<hipStream_t>(stream), k, dP, lddp, Y, ldy, Gt, ldg, dY, lddy, dGt, lddg, accumulate, M, rpw);
        return swr_launch_status();
    }
    SWR_REQUIRE(G == nullptr, SWR_ERR_UNSUPPORTED);
    const unsigned grid = static_cast<unsigned>(swr_ceil_div(M, MIX_WAVES) < 8192 ? swr_ceil_div(M, MIX_WAVES) : 8192);
    hipLaunchKernelGGL(mix_bwd_kernel, dim3(grid), dim3(MIX_WAVES * 64), MIX_WAVES * k.row_floats * sizeof(float),
                       static_cast<hipStream_t>(stream), k, dP, lddp, Y, ldy, dY, lddy, accumulate, M);
    return swr_launch_status();
}

// ----------------------------------------------------------------------------------- domain select
// final = 0; for d: final = where(domain_id == d, y_d, final)   (mmoe.py:53-55): exact integer compare
__global__ __launch_bounds__(EW_THREADS) void select_fwd_kernel(const float* __restrict__ V, int64_t ldv, int D,
                                                                const void* __restrict__ domain, int dom_dtype, int apply_sigmoid,
                                                                const float* __restrict__ extra, float* __restrict__ out, int64_t M) {
    const int64_t m = static_cast<int64_t>(blockIdx.x) * EW_THREADS + threadIdx.x;
    if (m >= M) return;
    const int64_t dom = swr_load_index(domain, dom_dtype, m);
    float v = 0.f;
    if (dom >= 0 && dom < D) {
        v = V[m * ldv + dom];
        if (apply_sigmoid) v = swr_sigmoid(v);
    }
    if (extra) v = swr_sigmoid(v + extra[m]);     // STAR: sig(final + aux_out), star.py:117
    out[m] = v;
}

extern "C" int swr_select_fwd(const float* V, int64_t ldv, int D, const void* domain, int dom_dtype, int apply_sigmoid,
                              const float* extra, float* out, int64_t M, void* stream) {
    SWR_REQUIRE(V && domain && out && D > 0 && M >= 0 && ldv >= D, SWR_ERR_ARG);
    SWR_REQUIRE(swr_is_index_dtype(dom_dtype), SWR_ERR_DTYPE);
    if (M == 0) return SWR_OK;
    hipLaunchKernelGGL(select_fwd_kernel, dim3(static_cast<unsigned>(swr_ceil_div(M, EW_THREADS))), dim3(EW_THREADS), 0,
                       static_cast<hipStream_t>(stream), V, ldv, D, domain, dom_dtype, apply_sigmoid, extra, out, M);
    return swr_launch_status();
}

// thread per (m, d): dV[m, d] = (d == dom[m]) ? dout * f'(.) : 0 -- the select makes dL/dy_d zero on
// foreign rows (BatchNorm re-densifies it one layer down)
__global__ __launch_bounds__(EW_THREADS) void select_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                                int D, const void* __restrict__ domain, int dom_dtype,
                                                                int apply_sigmoid, int has_extra, float* __restrict__ dV,
                                                                int64_t lddv, float* __restrict__ dextra, int64_t M) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * EW_THREADS + threadIdx.x;
    const int64_t m = idx / D;
    if (m >= M) return;
    const int d = static_cast<int>(idx - m * D);
    const int64_t dom = swr_load_index(domain, dom_dtype, m);
    float g = dout[m];
    const float o = out[m];
    if (has_extra) {
        g *= o * (1.f - o);                        // through the outer sigmoid
        if (d == 0 && dextra) dextra[m] = g;
    }
    float r = 0.f;
    if (dom == d) r = (apply_sigmoid && !has_extra) ? g * o * (1.f - o) : g;
    dV[m * lddv + d] = r;
}

extern "C" int swr_select_bwd(const float* dout, const float* out, int D, const void* domain, int dom_dtype,
                              int apply_sigmoid, int has_extra, float* dV, int64_t lddv, float* dextra, int64_t M,
                              void* stream) {
    SWR_REQUIRE(dout && out && domain && dV && D > 0 && M >= 0 && lddv >= D, SWR_ERR_ARG);
    SWR_REQUIRE(!(apply_sigmoid && has_extra), SWR_ERR_UNSUPPORTED);
    SWR_REQUIRE(swr_is_index_dtype(dom_dtype), SWR_ERR_DTYPE);
    if (M == 0) return SWR_OK;
    hipLaunchKernelGGL(select_bwd_kernel, dim3(static_cast<unsigned>(swr_ceil_div(M * D, EW_THREADS))), dim3(EW_THREADS), 0,
                       static_cast<hipStream_t>(stream), dout, out, D, domain, dom_dtype, apply_sigmoid, has_extra, dV, lddv,
                       dextra, M);
    return swr_launch_status();
}

// --------------------------------------------------------------------------------------------- BCE
__device__ __forceinline__ float block_sum(float v, float* sm) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0)
        for (int w = 0; w < EW_THREADS / 64; ++w) r += sm[w];
    return r;   // valid in thread 0
}

#define BCE_PER_BLOCK 512     // 2 rows per thread: enough workgroups to cover the latency of a 65 536-row batch
__global__ __launch_bounds__(EW_THREADS) void bce_partial_kernel(const float* __restrict__ p, const void* __restrict__ y,
                                                                 int y_dtype, int64_t M, float* __restrict__ part) {
    __shared__ float sm[EW_THREADS / 64];
    const int64_t base = static_cast<int64_t>(blockIdx.x) * BCE_PER_BLOCK;
    float acc = 0.f;
    for (int k = threadIdx.x; k < BCE_PER_BLOCK; k += EW_THREADS) {
        const int64_t m = base + k;
        if (m < M) {
            const float pi = p[m], yi = swr_load_value(y, y_dtype, m);
            const float lp = fmaxf(logf(pi), -100.f), l1 = fmaxf(logf(1.f - pi), -100.f);   // torch clamps the logs
            acc -= yi * lp + (1.f - yi) * l1;
        }
    }
    const float tot = block_sum(acc, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

__global__ __launch_bounds__(EW_THREADS) void bce_final_kernel(const float* __restrict__ part, int n_part, int64_t M,
                                                               float* __restrict__ loss) {
    __shared__ double sm[EW_THREADS / 64];
    double acc = 0.0;
    const int per = (n_part + EW_THREADS - 1) / EW_THREADS;
    for (int t = threadIdx.x * per; t < min((threadIdx.x + 1) * per, n_part); ++t) acc += part[t];
    const double total = swr_block_sum_f64<EW_THREADS>(acc, sm);
    if (threadIdx.x == 0) *loss = static_cast<float>(total / static_cast<double>(M));
}

extern "C" size_t swr_bce_workspace_bytes(int64_t M) { return static_cast<size_t>(swr_ceil_div(M > 0 ? M : 1, BCE_PER_BLOCK)) * 4 + 256; }

extern "C" int swr_bce_fwd(const float* p, const void* y, int y_dtype, int64_t M, float* loss, void* workspace,
                           size_t workspace_bytes, void* stream) {
    SWR_REQUIRE(p && y && loss && workspace && M > 0, SWR_ERR_ARG);
    SWR_REQUIRE(swr_is_value_dtype(y_dtype), SWR_ERR_DTYPE);
    SWR_REQUIRE(workspace_bytes >= swr_bce_workspace_bytes(M), SWR_ERR_WORKSPACE);
    const int nb = static_cast<int>(swr_ceil_div(M, BCE_PER_BLOCK));
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(bce_partial_kernel, dim3(nb), dim3(EW_THREADS), 0, st, p, y, y_dtype, M, static_cast<float*>(workspace));
    hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(EW_THREADS), 0, st, static_cast<const float*>(workspace), nb, M, loss);
    return swr_launch_status();
}

__global__ __launch_bounds__(EW_THREADS) void bce_bwd_kernel(const float* __restrict__ p, const void* __restrict__ y, int y_dtype,
                                                             int64_t M, const float* __restrict__ dloss, float* __restrict__ dp) {
    const int64_t m = static_cast<int64_t>(blockIdx.x) * EW_THREADS + threadIdx.x;
    if (m >= M) return;
    const float pi = p[m], yi = swr_load_value(y, y_dtype, m);
    dp[m] = dloss[0] * (pi - yi) / fmaxf(pi * (1.f - pi), 1e-12f) / static_cast<float>(M);
}

extern "C" int swr_bce_bwd(const float* p, const void* y, int y_dtype, int64_t M, const float* dloss, float* dp, void* stream) {
    SWR_REQUIRE(p && y && dloss && dp && M > 0, SWR_ERR_ARG);
    SWR_REQUIRE(swr_is_value_dtype(y_dtype), SWR_ERR_DTYPE);
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(static_cast<unsigned>(swr_ceil_div(M, EW_THREADS))), dim3(EW_THREADS), 0,
                       static_cast<hipStream_t>(stream), p, y, y_dtype, M, dloss, dp);
    return swr_launch_status();
}

// per-workgroup loss partial + the final fixed-order fp64 sum by whichever workgroup draws the last ticket (no spinning: a
// workgroup either is last or leaves); the ticket word is zero on entry and zero again on exit
// `adv` (nullable): the optimizer's step bookkeeping rides here -- run by the one thread that finishes the launch (swr.h
// "swr_select_bce_fwd_adv")
struct BceAdv { swr_adam_hyper* hyper; float* hist; int64_t cap; };
__device__ __forceinline__ void bce_finish(float acc, float* sm, double* smd, bool* is_last, float* part, int n_part,
                                           uint32_t* ticket, float* __restrict__ loss, int64_t M, const BceAdv adv = BceAdv{nullptr, nullptr, 0}) {
    const float tot = block_sum(acc, sm);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = tot;
        const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        *is_last = t == gridDim.x - 1;
    }
    __syncthreads();
    if (!*is_last) return;
    double a = 0.0;
    const int per = (n_part + EW_THREADS - 1) / EW_THREADS;
    for (int t = threadIdx.x * per; t < min((threadIdx.x + 1) * per, n_part); ++t)
        a += __hip_atomic_load(part + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double total = swr_block_sum_f64<EW_THREADS>(a, smd);
    if (threadIdx.x == 0) {
        *loss = static_cast<float>(total / static_cast<double>(M));
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (adv.hyper) swr_adam_advance_body(adv.hyper, adv.hist, adv.cap);
    }
}

// ------------------------------------------------------------------- select + BCE in one launch each way
// The model's last op (domain select of the tower sigmoids, mmoe.py:51-55) and the trainer's criterion
// (BCELoss, ctr_trainer.py:56,70) back to back: p and the per-workgroup loss partials in one pass, the final fp64 tree
// by whichever workgroup draws the last ticket (same partials, same tree as swr_bce_fwd -> same bits; no spinning:
// a workgroup either is last or leaves).  The ticket word must be zero on entry and is zero again on exit.
__global__ __launch_bounds__(EW_THREADS) void select_bce_fwd_kernel(const float* __restrict__ V, int64_t ldv, int D,
                                                                    const void* __restrict__ domain, int dom_dtype,
                                                                    const void* __restrict__ y, int y_dtype, int64_t M,
                                                                    float* __restrict__ p_out, float* part, int n_part,
                                                                    uint32_t* ticket, float* __restrict__ loss, const BceAdv adv) {
    __shared__ float sm[EW_THREADS / 64];
    __shared__ double smd[EW_THREADS / 64];
    __shared__ bool is_last;
    const int64_t base = static_cast<int64_t>(blockIdx.x) * BCE_PER_BLOCK;
    float acc = 0.f;
    for (int k = threadIdx.x; k < BCE_PER_BLOCK; k += EW_THREADS) {
        const int64_t m = base + k;
        if (m < M) {
            const int64_t dom = swr_load_index(domain, dom_dtype, m);
            float pi = 0.f;
            if (dom >= 0 && dom < D) pi = swr_sigmoid(V[m * ldv + dom]);
            p_out[m] = pi;
            const float yi = swr_load_value(y, y_dtype, m);
            const float lp = fmaxf(logf(pi), -100.f), l1 = fmaxf(logf(1.f - pi), -100.f);   // torch clamps the logs
            acc -= yi * lp + (1.f - yi) * l1;
        }
    }
    bce_finish(acc, sm, smd, &is_last, part, n_part, ticket, loss, M, adv);
}

extern "C" int swr_select_bce_fwd_adv(const float* V, int64_t ldv, int D, const void* domain, int dom_dtype, const void* y,
                                      int y_dtype, int64_t M, float* p, float* loss, void* workspace, size_t workspace_bytes,
                                      uint32_t* ticket, void* adv_hyper, float* adv_hist, int64_t adv_hist_cap, void* stream);
extern "C" int swr_select_bce_fwd(const float* V, int64_t ldv, int D, const void* domain, int dom_dtype, const void* y,
                                  int y_dtype, int64_t M, float* p, float* loss, void* workspace, size_t workspace_bytes,
                                  uint32_t* ticket, void* stream) {
    return swr_select_bce_fwd_adv(V, ldv, D, domain, dom_dtype, y, y_dtype, M, p, loss, workspace, workspace_bytes, ticket, nullptr,
                                  nullptr, 0, stream);
}
extern "C" int swr_select_bce_fwd_adv(const float* V, int64_t ldv, int D, const void* domain, int dom_dtype, const void* y,
                                      int y_dtype, int64_t M, float* p, float* loss, void* workspace, size_t workspace_bytes,
                                      uint32_t* ticket, void* adv_hyper, float* adv_hist, int64_t adv_hist_cap, void* stream) {
    SWR_REQUIRE(V && domain && y && p && loss && workspace && ticket && D > 0 && M > 0 && ldv >= D, SWR_ERR_ARG);
    SWR_REQUIRE(!adv_hyper || !adv_hist || (adv_hist_cap > 1 && adv_hist_cap <= (1ll << 31) && (adv_hist_cap & (adv_hist_cap - 1)) == 0),
                SWR_ERR_ARG);
    const BceAdv adv = {static_cast<swr_adam_hyper*>(adv_hyper), adv_hyper ? adv_hist : nullptr, adv_hist_cap};
    SWR_REQUIRE(swr_is_index_dtype(dom_dtype), SWR_ERR_DTYPE);
    SWR_REQUIRE(swr_is_value_dtype(y_dtype), SWR_ERR_DTYPE);
    SWR_REQUIRE(workspace_bytes >= swr_bce_workspace_bytes(M), SWR_ERR_WORKSPACE);
    const int nb = static_cast<int>(swr_ceil_div(M, BCE_PER_BLOCK));
    hipLaunchKernelGGL(select_bce_fwd_kernel, dim3(nb), dim3(EW_THREADS), 0, static_cast<hipStream_t>(stream), V, ldv, D, domain,
                       dom_dtype, y, y_dtype, M, p, static_cast<float*>(workspace), nb, ticket, loss, adv);
    return swr_launch_status();
}

// The per-domain tower output layer + sigmoid + domain select + BCE in one launch (towers of mmoe.py:38-41,50-55 /
// ple.py / sharebottom.py, tower_params = {"dims": [H]}, in training mode): a row evaluates ONLY the tower of its own
// domain -- V[m, g] for the other towers is never used by the loss -- with the operation order of tower_head_fwd_kernel
// (tower.hip) and of select_bce_fwd_kernel above, so p and the loss are the bits of the two-launch path.
__global__ __launch_bounds__(EW_THREADS) void tower_head_select_bce_kernel(
    const float* __restrict__ Z1, int64_t ldz, int G, int Hd, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ w2, const float* __restrict__ b2, const void* __restrict__ domain, int dom_dtype,
    const void* __restrict__ y, int y_dtype, int64_t M, float* __restrict__ p_out, float* part, int n_part, uint32_t* ticket,
    float* __restrict__ loss, const BceAdv adv) {
    __shared__ float sm[EW_THREADS / 64];
    __shared__ double smd[EW_THREADS / 64];
    __shared__ bool is_last;
    const int64_t base = static_cast<int64_t>(blockIdx.x) * BCE_PER_BLOCK;
    float acc = 0.f;
    for (int k = threadIdx.x; k < BCE_PER_BLOCK; k += EW_THREADS) {
        const int64_t m = base + k;
        if (m < M) {
            const int64_t dom = swr_load_index(domain, dom_dtype, m);
            float pi = 0.f;
            if (dom >= 0 && dom < G) {
                const int c0 = static_cast<int>(dom) * Hd;
                const float* __restrict__ z = Z1 + m * ldz + c0;
                float v = b2 ? b2[dom] : 0.f;
                for (int j = 0; j < Hd; j += 4) {
                    const float4 zq = *reinterpret_cast<const float4*>(z + j);
                    const float zz[4] = {zq.x, zq.y, zq.z, zq.w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float act = fmaxf(fmaf(zz[u], scale[c0 + j + u], shift[c0 + j + u]), 0.f);
                        v = fmaf(act, w2[c0 + j + u], v);
                    }
                }
                pi = swr_sigmoid(v);
            }
            p_out[m] = pi;
            const float yi = swr_load_value(y, y_dtype, m);
            const float lp = fmaxf(logf(pi), -100.f), l1 = fmaxf(logf(1.f - pi), -100.f);   // torch clamps the logs
            acc -= yi * lp + (1.f - yi) * l1;
        }
    }
    bce_finish(acc, sm, smd, &is_last, part, n_part, ticket, loss, M, adv);
}

extern "C" int swr_tower_head_select_bce_fwd_adv(const float* Z1, int64_t ldz, int G, int H, const float* scale,
                                                 const float* shift, const float* w2, const float* b2, const void* domain,
                                                 int dom_dtype, const void* y, int y_dtype, int64_t M, float* p, float* loss,
                                                 void* workspace, size_t workspace_bytes, uint32_t* ticket, void* adv_hyper,
                                                 float* adv_hist, int64_t adv_hist_cap, void* stream);
extern "C" int swr_tower_head_select_bce_fwd(const float* Z1, int64_t ldz, int G, int H, const float* scale,
                                             const float* shift, const float* w2, const float* b2, const void* domain,
                                             int dom_dtype, const void* y, int y_dtype, int64_t M, float* p, float* loss,
                                             void* workspace, size_t workspace_bytes, uint32_t* ticket, void* stream) {
    return swr_tower_head_select_bce_fwd_adv(Z1, ldz, G, H, scale, shift, w2, b2, domain, dom_dtype, y, y_dtype, M, p, loss, workspace,
                                             workspace_bytes, ticket, nullptr, nullptr, 0, stream);
}
extern "C" int swr_tower_head_select_bce_fwd_adv(const float* Z1, int64_t ldz, int G, int H, const float* scale,
                                                 const float* shift, const float* w2, const float* b2, const void* domain,
                                                 int dom_dtype, const void* y, int y_dtype, int64_t M, float* p, float* loss,
                                                 void* workspace, size_t workspace_bytes, uint32_t* ticket, void* adv_hyper,
                                                 float* adv_hist, int64_t adv_hist_cap, void* stream) {
    SWR_REQUIRE(Z1 && scale && shift && w2 && domain && y && p && loss && workspace && ticket && G > 0 && M > 0, SWR_ERR_ARG);
    SWR_REQUIRE(!adv_hyper || !adv_hist || (adv_hist_cap > 1 && adv_hist_cap <= (1ll << 31) && (adv_hist_cap & (adv_hist_cap - 1)) == 0),
                SWR_ERR_ARG);
    const BceAdv adv = {static_cast<swr_adam_hyper*>(adv_hyper), adv_hyper ? adv_hist : nullptr, adv_hist_cap};
    SWR_REQUIRE(H > 0 && H % 4 == 0 && ldz >= static_cast<int64_t>(G) * H && ldz % 4 == 0 && swr_aligned16(Z1), SWR_ERR_ALIGN);
    SWR_REQUIRE(swr_is_index_dtype(dom_dtype), SWR_ERR_DTYPE);
    SWR_REQUIRE(swr_is_value_dtype(y_dtype), SWR_ERR_DTYPE);
    SWR_REQUIRE(workspace_bytes >= swr_bce_workspace_bytes(M), SWR_ERR_WORKSPACE);
    const int nb = static_cast<int>(swr_ceil_div(M, BCE_PER_BLOCK));
    hipLaunchKernelGGL(tower_head_select_bce_kernel, dim3(nb), dim3(EW_THREADS), 0, static_cast<hipStream_t>(stream), Z1, ldz, G,
                       H, scale, shift, w2, b2, domain, dom_dtype, y, y_dtype, M, p, static_cast<float*>(workspace), nb, ticket,
                       loss, adv);
    return swr_launch_status();
}

// dV[m, d] = (d == dom[m]) ? dBCE/dp * p (1 - p) : 0, the two factors rounded exactly as swr_bce_bwd and
// swr_select_bwd round them
__global__ __launch_bounds__(EW_THREADS) void select_bce_bwd_kernel(const float* __restrict__ p, const void* __restrict__ y,
                                                                    int y_dtype, int D, const void* __restrict__ domain,
                                                                    int dom_dtype, int64_t M, const float* __restrict__ dloss,
                                                                    float* __restrict__ dV, int64_t lddv) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * EW_THREADS + threadIdx.x;
    const int64_t m = idx / D;
    if (m >= M) return;
    const int d = static_cast<int>(idx - m * D);
    float r = 0.f;
    if (swr_load_index(domain, dom_dtype, m) == d) {
        r = swr_bce_logit_grad(p[m], swr_load_value(y, y_dtype, m), dloss[0], M);
    }
    dV[m * lddv + d] = r;
}

extern "C" int swr_select_bce_bwd(const float* p, const void* y, int y_dtype, int D, const void* domain, int dom_dtype,
                                  int64_t M, const float* dloss, float* dV, int64_t lddv, void* stream) {
    SWR_REQUIRE(p && y && domain && dloss && dV && D > 0 && M > 0 && lddv >= D, SWR_ERR_ARG);
    SWR_REQUIRE(swr_is_index_dtype(dom_dtype), SWR_ERR_DTYPE);
    SWR_REQUIRE(swr_is_value_dtype(y_dtype), SWR_ERR_DTYPE);
    hipLaunchKernelGGL(select_bce_bwd_kernel, dim3(static_cast<unsigned>(swr_ceil_div(M * D, EW_THREADS))), dim3(EW_THREADS), 0,
                       static_cast<hipStream_t>(stream), p, y, y_dtype, D, domain, dom_dtype, M, dloss, dV, lddv);
    return swr_launch_status();
}

// ------------------------------------------------------------------ per-sample row-vector x matrix product
// out[b, d, :] = T[b, d, :] @ Hm[b]  with a k x k matrix PER SAMPLE: the middle factor of HAMUR's adapter weights
// W_b = U H_b V applied as ((h U) H_b) V (hamur.py:175-186 materialises U H_b V per sample instead).  HBM-bound on
// the k*k floats of H_b per sample.  One wave per sample: H_b and the D rows staged in LDS, lane j owns column j.
#define RM_WAVES 4
#define RM_DB 8                         // rows of T per pass: one register per row and lane
#define RM_KMAX 64                      // k <= 64: lane j owns column j
// One wave per sample, lane j = column j.  The D rows of T (and of dOut) live in registers, one per row, lane i holding
// element i; the scalar t[d, i] an FMA needs is a v_readlane of that register (uniform i) -- no LDS round trip, no branch
// in the inner loops (rows past D are zero registers).  H_b streams straight from global memory (forward: lane j reads
// h[i, j], consecutive lanes consecutive addresses) or through a wave-private LDS copy (backward, which also walks its
// rows).  The first version kept T and H_b in LDS and gave every (d, j) its own k-trip dependent loop of two LDS reads per
// FMA: LDS-latency-bound, 105 us forward / 232 us backward for HAMUR's [32 768, 8, 35] x [35, 35] (160 MB of H_b).
__device__ __forceinline__ float rm_bcast(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

__global__ __launch_bounds__(RM_WAVES * 64) void rowmat_fwd_kernel(const float* __restrict__ T, const float* __restrict__ Hm,
                                                                   float* __restrict__ out, int64_t B, int D, int k) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kk = k * k, dk = D * k;
    const bool on = lane < k;
    const int j = on ? lane : 0;
    for (int64_t b = static_cast<int64_t>(blockIdx.x) * RM_WAVES + wave; b < B; b += static_cast<int64_t>(gridDim.x) * RM_WAVES) {
        const float* __restrict__ h = Hm + b * kk + j;
        for (int d0 = 0; d0 < D; d0 += RM_DB) {
            float tr[RM_DB], acc[RM_DB];
#pragma unroll
            for (int u = 0; u < RM_DB; ++u) {
                tr[u] = (on && d0 + u < D) ? T[b * dk + (d0 + u) * k + lane] : 0.f;
                acc[u] = 0.f;
            }
#pragma unroll 5
            for (int i = 0; i < k; ++i) {
                const float hv = h[i * k];
#pragma unroll
                for (int u = 0; u < RM_DB; ++u) acc[u] = fmaf(rm_bcast(tr[u], i), hv, acc[u]);
            }
#pragma unroll
            for (int u = 0; u < RM_DB; ++u)
                if (on && d0 + u < D) out[b * dk + (d0 + u) * k + lane] = acc[u];
        }
    }
}

// dT[b, d, i] = sum_j dOut[b, d, j] Hm[b, i, j];   dHm[b, i, j] = sum_d T[b, d, i] dOut[b, d, j]
__global__ __launch_bounds__(RM_WAVES * 64) void rowmat_bwd_kernel(const float* __restrict__ dOut, const float* __restrict__ T,
                                                                   const float* __restrict__ Hm, float* __restrict__ dT,
                                                                   float* __restrict__ dHm, int acc_dhm, int64_t B, int D, int k) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kk = k * k, dk = D * k;
    const bool on = lane < k;
    float* hs = lds + wave * kk;                                  // this wave's H_b (dT only)
    for (int64_t b = static_cast<int64_t>(blockIdx.x) * RM_WAVES + wave; b < B; b += static_cast<int64_t>(gridDim.x) * RM_WAVES) {
        if (dT) {
            const float* __restrict__ src = Hm + b * kk;
#pragma unroll 5
            for (int q = lane; q < kk; q += 64) hs[q] = src[q];
            __builtin_amdgcn_wave_barrier();
        }
        for (int d0 = 0; d0 < D; d0 += RM_DB) {
            float gr[RM_DB], tr[RM_DB];
#pragma unroll
            for (int u = 0; u < RM_DB; ++u) {
                const bool ok = on && d0 + u < D;
                gr[u] = ok ? dOut[b * dk + (d0 + u) * k + lane] : 0.f;
                tr[u] = ok ? T[b * dk + (d0 + u) * k + lane] : 0.f;
            }
            if (dT) {                                             // lane i: row i of H_b against the rows of dOut
                float acc[RM_DB];
#pragma unroll
                for (int u = 0; u < RM_DB; ++u) acc[u] = 0.f;
                const float* hrow = hs + (on ? lane : 0) * k;
#pragma unroll 5
                for (int jj = 0; jj < k; ++jj) {
                    const float hv = hrow[jj];
#pragma unroll
                    for (int u = 0; u < RM_DB; ++u) acc[u] = fmaf(rm_bcast(gr[u], jj), hv, acc[u]);
                }
#pragma unroll
                for (int u = 0; u < RM_DB; ++u)
                    if (on && d0 + u < D) dT[b * dk + (d0 + u) * k + lane] = acc[u];
            }
            if (dHm) {                                            // lane j: column j of dOut (its own registers) against T
                float* __restrict__ dst = dHm + b * kk + (on ? lane : 0);
#pragma unroll 5
                for (int i = 0; i < k; ++i) {
                    float acc = 0.f;
#pragma unroll
                    for (int u = 0; u < RM_DB; ++u) acc = fmaf(rm_bcast(tr[u], i), gr[u], acc);
                    if (on) {
                        if (d0 == 0 && !acc_dhm) dst[i * k] = acc;
                        else dst[i * k] += acc;                   // D > 8: further passes add (same thread, same address)
                    }
                }
            }
        }
        if (dT) __builtin_amdgcn_wave_barrier();
    }
}


// ---- pipelined forward (k <= 35, long batches).  rowmat_fwd_kernel walks a sample with its loads issued as the loop reaches them (35
// loads of one 140-byte row of H_b each, five in flight) and two vector instructions per FMA (v_readlane + v_fmac): 88 us for HAMUR's
// [32 768, 8, 35] x [35, 35] (234 MB).  Here a wave fetches the NEXT sample's H_b (16-byte loads from the 16-byte boundary below it)
// and rows of T into registers before it computes the current one from a wave-private LDS copy, and an FMA is ONE v_fmac with a DPP
// row broadcast of its T operand: 71 us.  Same sums in the same order: identical bits.  (The backward in the same two forms was
// slower than rowmat_bwd_kernel -- 193 us pipelined with read-lanes at 171 VGPRs, 385 us with broadcasts at 256 VGPRs, against
// 150 -- and is not kept; T through the scalar cache instead of broadcasts: 152 us forward, the scalar cache does not stream.)
// acc += t[lane E of the lane's 16-lane row] * h in ONE vector instruction: the DPP form of v_fmac with the row broadcast on its
// first source (v_readlane + v_fmac are two).  `t` must not have been written by the two preceding instructions (DPP hazard):
// the callers keep it loop-invariant.
// GUARD: the chunk's first product carries two wait states in front of it inside the SAME asm statement (hipcc's hazard
// recogniser does not look into inline asm, and nothing else stops it from scheduling the load-to-register move of `t` right in front).
template <int E, bool GUARD = false>
__device__ __forceinline__ void rm_fmac_bcast(float& acc, float t, float h) {
    if constexpr (GUARD)
        asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(t), "v"(h), "n"(E));
    else
        asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(t), "v"(h), "n"(E));
}
// one 16-element chunk starting at i0: acc[u] += tc[u][row element E] * hv[E] for E = 0 .. 15 (hv: the chunk's column values,
// zero past k: those products add +-0 to sums that are never -0 ... so they are skipped instead, by a wave-uniform test)
template <int E>
__device__ __forceinline__ void rm_chunk(float (&acc)[RM_DB], const float (&tc)[RM_DB], const float (&hv)[16], int n) {
    if (E < n) {
#pragma unroll
        for (int u = 0; u < RM_DB; ++u) {
            if (E == 0 && u == 0) rm_fmac_bcast<E, true>(acc[u], tc[u], hv[E]);
            else rm_fmac_bcast<E>(acc[u], tc[u], hv[E]);
        }
    }
    if constexpr (E + 1 < 16) rm_chunk<E + 1>(acc, tc, hv, n);
}
#define RM_CHUNKS 3                     // 16-element chunks of a row held in registers: k <= 48 (the launchers route k <= 35 here)

template <int R>
__global__ __launch_bounds__(RM_WAVES * 64) void rowmat_fwd2_kernel(const float* __restrict__ T, const float* __restrict__ Hm,
                                                                    float* __restrict__ out, int64_t B, int D, int k) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l16 = lane & 15;
    const int kk = k * k, dk = D * k;
    const bool on = lane < k;
    const int j = on ? lane : 0;
    float* hs = lds + wave * kk;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * RM_WAVES;
    int64_t b = static_cast<int64_t>(blockIdx.x) * RM_WAVES + wave;
    float4 hn[R];                        // R = 16-byte loads per lane of one H_b (from the 16-byte boundary at or below its start)
    float tn[RM_CHUNKS][RM_DB];
    int sh_n = 0;
    // a row of T as RM_CHUNKS registers: chunk c holds elements 16 c + (lane & 15), the same 16 values in each of the wave's four
    // 16-lane rows (what the row broadcast of rm_fmac_bcast reads)
    auto fetch_t = [&](int64_t bb, int d0, float (&dst)[RM_CHUNKS][RM_DB]) {
        const float* __restrict__ tb = T + bb * dk;
#pragma unroll
        for (int c = 0; c < RM_CHUNKS; ++c)
#pragma unroll
            for (int u = 0; u < RM_DB; ++u)
                dst[c][u] = tb[min((d0 + u) * k + 16 * c + l16, dk - 1)];   // (elements past k / rows past D: never multiplied in / stored)
    };
    auto fetch = [&](int64_t bb) {
        // 16-byte loads whatever the alignment of the sample (k k floats is rarely a multiple of 4): start at the 16-byte boundary
        // at or below it -- the <= 3 floats in front and behind belong to the neighbouring samples / the same 16-byte granule of the
        // allocation -- and drop them when the registers go to LDS
        const uintptr_t a0 = reinterpret_cast<uintptr_t>(Hm + bb * kk);
        sh_n = static_cast<int>((a0 & 15u) >> 2);
        const float4* __restrict__ src = reinterpret_cast<const float4*>(a0 - 4 * sh_n);
        const int n4 = (kk + sh_n + 3) >> 2;
        // the caller's buffer is [Hm, Hm + B k k): the granule in front of sample 0 and the one behind sample B - 1 may reach
        // outside it -- those two (at most) are read element by element, inside the buffer only
        const float* const buf_lo = Hm;
        const float* const buf_hi = Hm + B * kk;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float* q = reinterpret_cast<const float*>(src + lane + 64 * r);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lane + 64 * r < n4) {
                if (q >= buf_lo && q + 4 <= buf_hi) {
                    v = *reinterpret_cast<const float4*>(q);
                } else {
                    if (q >= buf_lo && q < buf_hi) v.x = q[0];
                    if (q + 1 >= buf_lo && q + 1 < buf_hi) v.y = q[1];
                    if (q + 2 >= buf_lo && q + 2 < buf_hi) v.z = q[2];
                    if (q + 3 >= buf_lo && q + 3 < buf_hi) v.w = q[3];
                }
            }
            hn[r] = v;
        }
        fetch_t(bb, 0, tn);
    };
    if (b < B) fetch(b);
    for (; b < B; b += stride) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int q = 4 * (lane + 64 * r) - sh_n;
            if (q >= 0 && q < kk) hs[q] = hn[r].x;
            if (q + 1 >= 0 && q + 1 < kk) hs[q + 1] = hn[r].y;
            if (q + 2 >= 0 && q + 2 < kk) hs[q + 2] = hn[r].z;
            if (q + 3 >= 0 && q + 3 < kk) hs[q + 3] = hn[r].w;
        }
        float tr[RM_CHUNKS][RM_DB];
#pragma unroll
        for (int c = 0; c < RM_CHUNKS; ++c)
#pragma unroll
            for (int u = 0; u < RM_DB; ++u) tr[c][u] = tn[c][u];
        __builtin_amdgcn_wave_barrier();
        if (b + stride < B) fetch(b + stride);
        for (int d0 = 0; d0 < D; d0 += RM_DB) {
            if (d0 > 0) fetch_t(b, d0, tr);
            float acc[RM_DB];
#pragma unroll
            for (int u = 0; u < RM_DB; ++u) acc[u] = 0.f;
#pragma unroll
            for (int c = 0; c < RM_CHUNKS; ++c) {
                const int n = min(16, k - 16 * c);                // (wave-uniform)
                if (n > 0) {
                    float hv[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) hv[e] = hs[min(16 * c + e, k - 1) * k + j];
                    rm_chunk<0>(acc, tr[c], hv, n);
                }
                __builtin_amdgcn_sched_barrier(0);                // (one chunk's 16 column values live at a time)
            }
#pragma unroll
            for (int u = 0; u < RM_DB; ++u)
                if (on && d0 + u < D) out[b * dk + (d0 + u) * k + lane] = acc[u];
        }
        __builtin_amdgcn_wave_barrier();              // every lane is done with hs: the next sample overwrites it
    }
}

// ---- matrix-pipe forms (round 6).  The kernels above issue ONE multiply-add per lane and instruction on 35 of 64 lanes (88 / 71 us
// forward, 150 us backward for HAMUR's [32 768, 8, 35] x [35, 35]: vector-instruction-bound, 3-6 x the HBM time of their 160 MB of H_b).
// v_mfma_f32_4x4x1_16B_f32 multiplies sixteen independent 4 x 1 by 1 x 4 blocks per instruction -- 256 multiply-adds, each output a
// k-ordered fp32 fmaf chain, i.e. the SAME sums in the same order as the loops above (bit-identical results) -- and the blocks map
// onto this product without padding waste:
//   out / dT: a wave takes 8 samples; block (sample s, row block rb) x column block cb, k steps: lane (blk, q) feeds
//             A = X[s][4 rb + q][kk] (its own row of T / dOut, held in registers) and B = H_s[kk][4 cb + q] (out) or
//             H_s[4 cb + q][kk] (dT); ceil(k / 4) accumulators of 4 registers; 16 / 16 blocks busy.
//   dH:       a wave takes one sample; its ceil(k / 4)^2 output blocks in groups of 16, D <= 8 steps each:
//             A = T[s][d][4 rb + q], B = dOut[s][d][4 cb + q].
// Loads are plain dword loads (rows of 35 floats are 4-byte aligned at best); the B operand of the next k step is requested
// before the current step's products.  D <= 8 and k <= 4 RMM_NCB (36); other shapes keep the kernels above.
typedef float rm_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* rm_lds_ptr;
typedef __attribute__((address_space(1))) const void* rm_glb_ptr;
#define RMM_NCB 9                       // column blocks of 4: k <= 36
#define RMM_WAVES 4

// out / dT.  A workgroup = ONE wave = 8 samples: their H_b (8 k k contiguous floats) are copied into LDS with coalesced loads first --
// straight from global memory a B-operand load touched eight cache lines for 128 useful bytes (16 bytes per sample) and the kernel
// ran at a fifth of the vector kernels' speed -- and the operands are read back with conflict-free 4-byte LDS reads (a sample's
// matrices are k k = 1 225 floats = 9 banks apart).
template <bool TRANS>
__global__ __launch_bounds__(RMM_WAVES * 64) void rowmat_mfma_kernel(const float* __restrict__ X, const float* __restrict__ Hm,
                                                                     float* __restrict__ Y, int64_t B, int D, int k) {
    extern __shared__ __attribute__((aligned(16))) float lds[];            // [8][k k]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = lane >> 2, q = lane & 3;
    const int rb = blk & 1;
    const int64_t s0 = static_cast<int64_t>(blockIdx.x) * 8;
    const int64_t s = s0 + (blk >> 1);
    const bool s_ok = s < B;
    const int r = 4 * rb + q;                                   // the lane's row of X
    const int kk2 = k * k;
    // the four waves of the workgroup share the 8 samples' H and split the column blocks: wave 0 takes blocks 0-2, wave w blocks
    // 2 w + 1, 2 w + 2 (four workgroups per CU in different phases: one's copy runs under another's products)
    const int cb0 = wave == 0 ? 0 : 2 * wave + 1;
    const int ncb_w = wave == 0 ? 3 : 2;
    // ---- H of the workgroup's samples -> LDS (16-byte LDS-DMA copies when the block is aligned, as it is for a 16-byte aligned
    // tensor: 8 k k floats per workgroup; 1 KB per instruction, all in flight at once, no registers)
    {
        const int ns = static_cast<int>(min<int64_t>(8, B - s0));
        const int n = ns * kk2;
        const float* __restrict__ src = Hm + s0 * kk2;
        if ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
            const int n4 = n >> 2;
            int e = 64 * wave;
            for (; e + 64 <= n4; e += 64 * RMM_WAVES)
                __builtin_amdgcn_global_load_lds((rm_glb_ptr)(reinterpret_cast<const float4*>(src) + e + lane),
                                                 (rm_lds_ptr)(reinterpret_cast<float4*>(lds) + e), 16, 0, 0);
            for (int t = 4 * (n4 / 64 * 64) + tid; t < n; t += 64 * RMM_WAVES) lds[t] = src[t];          // (< 1 KB of tail)
        } else {
            for (int e0 = 0; e0 < n; e0 += 16 * 64 * RMM_WAVES) {                  // 16 loads in flight per thread
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) { const int e = e0 + u * 64 * RMM_WAVES + tid; v[u] = e < n ? src[e] : 0.f; }
#pragma unroll
                for (int u = 0; u < 16; ++u) { const int e = e0 + u * 64 * RMM_WAVES + tid; if (e < n) lds[e] = v[u]; }
            }
        }
    }
    // A operand: the lane's row, all k elements (zero past k, for rows past D and samples past B)
    float a[4 * RMM_NCB];
    {
        const float* __restrict__ xr = X + (s_ok ? s : 0) * (static_cast<int64_t>(D) * k) + (r < D ? r : 0) * k;
        const bool ok = s_ok && r < D;
#pragma unroll
        for (int e = 0; e < 4 * RMM_NCB; ++e) a[e] = (ok && e < k) ? xr[e] : 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const float* hs = lds + (blk >> 1) * kk2;
    // B operand of step kk, the wave's column block u: H[kk][4 cb + q] (out) / H[4 cb + q][kk] (dT); zero where 4 cb + q >= k
    auto load_b = [&](int kk, float (&b)[3]) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int c = 4 * (cb0 + u) + q;
            const bool ok = s_ok && u < ncb_w && c < k && kk < k;
            const int idx = TRANS ? c * k + kk : kk * k + c;
            b[u] = ok ? hs[ok ? idx : 0] : 0.f;
        }
    };
    rm_f32x4 acc[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) acc[u] = rm_f32x4{0.f, 0.f, 0.f, 0.f};
    float b0[3], b1[3], b2[3], b3[3];
    load_b(0, b0);
    load_b(1, b1);
    load_b(2, b2);
    // (fully unrolled: every register-array index is a constant; steps past k are skipped by wave-uniform tests, not by `break` --
    // with early exits hipcc kept the loop rolled and indexed `a` through scratch).  Four steps' operands in flight.
#define RMM_STEP(KK, BC, BN)                                                                                      \
    if ((KK) < k) {                                                                                               \
        load_b((KK) + 3, BN);                                                                                     \
        _Pragma("unroll") for (int u = 0; u < 3; ++u)                                                              \
            if (u < ncb_w) acc[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[KK], BC[u], acc[u], 0, 0, 0);            \
    }
#pragma unroll
    for (int kk = 0; kk < 4 * RMM_NCB; kk += 4) {
        RMM_STEP(kk, b0, b3)
        RMM_STEP(kk + 1, b1, b0)
        RMM_STEP(kk + 2, b2, b1)
        RMM_STEP(kk + 3, b3, b2)
    }
#undef RMM_STEP
    // D[i][j] of block blk: register i, lane 4 blk + j  ->  Y[s][4 rb + i][4 cb + q]
    if (s_ok) {
        float* __restrict__ yr = Y + s * (static_cast<int64_t>(D) * k);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int c = 4 * (cb0 + u) + q;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (u < ncb_w && c < k && 4 * rb + i < D) yr[(4 * rb + i) * k + c] = acc[u][i];
        }
    }
}

// dHm[s][i][j] (+)= sum_d T[s][d][i] dOut[s][d][j].  A wave takes one sample at a time: the sample's rows of T and dOut come in with
// coalesced loads (LDS), the k k products leave through an LDS tile and coalesced stores (written straight from the accumulators
// the 160 MB of dHm went out in 16-byte pieces).
__global__ __launch_bounds__(RMM_WAVES * 64) void rowmat_mfma_dh_kernel(const float* __restrict__ T, const float* __restrict__ dOut,
                                                                        float* __restrict__ dHm, int accumulate, int64_t B, int D, int k) {
    extern __shared__ __attribute__((aligned(16))) float lds[];            // per wave: [2][D k] operands | [k k] products
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bq = lane >> 2, q = lane & 3;
    const int ncb = (k + 3) >> 2, nblk = ncb * ncb;
    const int kk2 = k * k, dk = D * k;
    float* tsh = lds + wave * (2 * dk + kk2);
    float* gsh = tsh + dk;
    float* osh = gsh + dk;
    constexpr int NG = (RMM_NCB * RMM_NCB + 15) / 16;           // groups of 16 blocks
    for (int64_t s = static_cast<int64_t>(blockIdx.x) * RMM_WAVES + wave; s < B; s += static_cast<int64_t>(gridDim.x) * RMM_WAVES) {
        const float* __restrict__ ts = T + s * dk;
        const float* __restrict__ gs = dOut + s * dk;
        float* __restrict__ ds = dHm + s * kk2;
        {
            float tv[5], gv[5];                                  // D k <= 8 x 36 = 288 floats: five per lane, all loads in flight
#pragma unroll
            for (int u = 0; u < 5; ++u) { const int e = u * 64 + lane; tv[u] = e < dk ? ts[e] : 0.f; gv[u] = e < dk ? gs[e] : 0.f; }
#pragma unroll
            for (int u = 0; u < 5; ++u) { const int e = u * 64 + lane; if (e < dk) { tsh[e] = tv[u]; gsh[e] = gv[u]; } }
        }
        __builtin_amdgcn_wave_barrier();
        rm_f32x4 acc[NG];
        int rbv[NG], cbv[NG];
        bool okv[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int blk = 16 * g + bq;
            okv[g] = blk < nblk;
            rbv[g] = okv[g] ? blk / ncb : 0;
            cbv[g] = okv[g] ? blk - rbv[g] * ncb : 0;
            acc[g] = rm_f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            if (d < D) {                                         // (wave-uniform)
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int ra = 4 * rbv[g] + q, ca = 4 * cbv[g] + q;
                    const float av = (okv[g] && ra < k) ? tsh[d * k + ra] : 0.f;
                    const float bv = (okv[g] && ca < k) ? gsh[d * k + ca] : 0.f;
                    acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc[g], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int c = 4 * cbv[g] + q;
            if (okv[g] && c < k) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int rr = 4 * rbv[g] + i;
                    if (rr < k) osh[rr * k + c] = acc[g][i];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (accumulate) {
            float dv[21];                                        // k k <= 1 296 floats: 21 per lane, all loads in flight
#pragma unroll
            for (int u = 0; u < 21; ++u) { const int e = u * 64 + lane; dv[u] = e < kk2 ? ds[e] : 0.f; }
#pragma unroll
            for (int u = 0; u < 21; ++u) { const int e = u * 64 + lane; if (e < kk2) ds[e] = dv[u] + osh[e]; }
        } else {
            for (int e = lane; e < kk2; e += 64) ds[e] = osh[e];
        }
        __builtin_amdgcn_wave_barrier();                         // (the tile and the operands are reused by the next sample)
    }
}

// OPT-IN (SWR_ROWMAT_MFMA=1; read per call: the tests compare both forms in one process).  Measured at HAMUR's config-5 shape
// (tools/micro/rowmat_probe.py, rowmat_parts.py): forward 83 us against 109 for the pipelined vector kernel, dT 79 vs 81, dHm 99 vs 80
// (the vector backward computes dT and dHm in ONE pass over its operands: 147 us against 182 for the two launches here), and the
// config-5 step does not move (4.80-4.86 vs 4.84-4.85 ms): every one of these passes, vector or matrix, read-only or write-only,
// moves its 160 MB of per-sample matrices at ~2 TB/s -- the instruction mix is not what bounds them.
static int rowmat_mfma_mode() {
    const char* e = getenv("SWR_ROWMAT_MFMA");
    return (e && e[0] == '1') ? 1 : 0;
}
static bool rowmat_mfma_ok(int64_t B, int D, int k) { return rowmat_mfma_mode() && D <= 8 && k <= 4 * RMM_NCB && k >= 4 && B >= 64; }

static bool rowmat_pipelined() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("SWR_ROWMAT_PIPE"); on = (e && e[0] == '0') ? 0 : 1; }
    return on != 0;
}

extern "C" int swr_rowmat_fwd(const float* T, const float* Hm, float* out, int64_t B, int D, int k, void* stream) {
    SWR_REQUIRE(T && Hm && out && B >= 0 && D > 0 && k > 0, SWR_ERR_ARG);
    SWR_REQUIRE(k <= RM_KMAX, SWR_ERR_UNSUPPORTED);
    if (B == 0) return SWR_OK;
    if (rowmat_mfma_ok(B, D, k)) {
        hipLaunchKernelGGL(rowmat_mfma_kernel<false>, dim3(static_cast<unsigned>(swr_ceil_div(B, 8))), dim3(RMM_WAVES * 64),
                           static_cast<size_t>(8) * k * k * sizeof(float), static_cast<hipStream_t>(stream), T, Hm, out, B, D, k);
        return swr_launch_status();
    }
    if (rowmat_pipelined() && B >= 4096 && k * k <= 64 * 20) {
        // a wave walks ~8 samples: long enough for the prefetch to pay, short enough to fill the chip
        const unsigned grid2 = static_cast<unsigned>(std::min<int64_t>(swr_ceil_div(B, RM_WAVES), 1024));
        const size_t lds2 = static_cast<size_t>(RM_WAVES) * k * k * sizeof(float);
        hipStream_t st = static_cast<hipStream_t>(stream);
        if (k * k + 3 <= 256 * 2) hipLaunchKernelGGL(rowmat_fwd2_kernel<2>, dim3(grid2), dim3(RM_WAVES * 64), lds2, st, T, Hm, out, B, D, k);
        else hipLaunchKernelGGL(rowmat_fwd2_kernel<5>, dim3(grid2), dim3(RM_WAVES * 64), lds2, st, T, Hm, out, B, D, k);
        return swr_launch_status();
    }
    const unsigned grid = static_cast<unsigned>(std::min<int64_t>(swr_ceil_div(B, RM_WAVES), 16384));
    hipLaunchKernelGGL(rowmat_fwd_kernel, dim3(grid), dim3(RM_WAVES * 64), 0, static_cast<hipStream_t>(stream), T, Hm, out, B, D, k);
    return swr_launch_status();
}

extern "C" int swr_rowmat_bwd(const float* dOut, const float* T, const float* Hm, float* dT, float* dHm, int accumulate_dhm,
                              int64_t B, int D, int k, void* stream) {
    SWR_REQUIRE(dOut && T && Hm && (dT || dHm) && B >= 0 && D > 0 && k > 0, SWR_ERR_ARG);
    SWR_REQUIRE(k <= RM_KMAX, SWR_ERR_UNSUPPORTED);
    const size_t lds = dT ? static_cast<size_t>(RM_WAVES) * k * k * sizeof(float) : 0;
    if (B == 0) return SWR_OK;
    if (rowmat_mfma_ok(B, D, k)) {
        hipStream_t st = static_cast<hipStream_t>(stream);
        if (dT)
            hipLaunchKernelGGL(rowmat_mfma_kernel<true>, dim3(static_cast<unsigned>(swr_ceil_div(B, 8))), dim3(RMM_WAVES * 64),
                               static_cast<size_t>(8) * k * k * sizeof(float), st, dOut, Hm, dT, B, D, k);
        if (dHm)
            hipLaunchKernelGGL(rowmat_mfma_dh_kernel, dim3(static_cast<unsigned>(std::min<int64_t>(swr_ceil_div(B, RMM_WAVES), 2048))),
                               dim3(RMM_WAVES * 64), static_cast<size_t>(RMM_WAVES) * (2 * D * k + k * k) * sizeof(float), st, T, dOut,
                               dHm, accumulate_dhm, B, D, k);
        return swr_launch_status();
    }
    const unsigned grid = static_cast<unsigned>(std::min<int64_t>(swr_ceil_div(B, RM_WAVES), 16384));
    hipLaunchKernelGGL(rowmat_bwd_kernel, dim3(grid), dim3(RM_WAVES * 64), lds, static_cast<hipStream_t>(stream), dOut, T, Hm, dT,
                       dHm, accumulate_dhm, B, D, k);
    return swr_launch_status();
}

// ------------------------------------------------------------------------------------- elementwise
__global__ __launch_bounds__(EW_THREADS) void mul_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                         float* __restrict__ C, int64_t n) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * EW_THREADS + threadIdx.x;
    if (i < n) C[i] = A[i] * B[i];
}

// c = a + b (residual connections: HAMUR's adapter `out + h`, hamur.py:197 / 366; M3oE's `star_mlp(emb) + skip`, m3oe.py:147):
// 16 bytes per lane when the three pointers allow it
__global__ __launch_bounds__(EW_THREADS) void add_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                         int64_t n, int vec) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * EW_THREADS + threadIdx.x;
    if (vec) {
        if (4 * i + 3 < n) {
            const float4 a = *reinterpret_cast<const float4*>(A + 4 * i), b = *reinterpret_cast<const float4*>(B + 4 * i);
            *reinterpret_cast<float4*>(C + 4 * i) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        } else {
            for (int64_t j = 4 * i; j < n; ++j) C[j] = A[j] + B[j];
        }
    } else if (i < n) {
        C[i] = A[i] + B[i];
    }
}

extern "C" int swr_add_fwd(const float* A, const float* B, float* C, int64_t n, void* stream) {
    SWR_REQUIRE(A && B && C && n >= 0, SWR_ERR_ARG);
    if (n == 0) return SWR_OK;
    const int vec = swr_aligned16(A) && swr_aligned16(B) && swr_aligned16(C);
    const int64_t items = vec ? swr_ceil_div(n, 4) : n;
    hipLaunchKernelGGL(add_kernel, dim3(static_cast<unsigned>(swr_ceil_div(items, EW_THREADS))), dim3(EW_THREADS), 0,
                       static_cast<hipStream_t>(stream), A, B, C, n, vec);
    return swr_launch_status();
}

__global__ __launch_bounds__(EW_THREADS) void mul_scale_kernel(const float* __restrict__ A, const float* __restrict__ B, float s,
                                                               float* __restrict__ C, int64_t n, int vec) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * EW_THREADS + threadIdx.x;
    if (vec) {
        if (4 * i + 3 < n) {
            const float4 a = *reinterpret_cast<const float4*>(A + 4 * i), b = *reinterpret_cast<const float4*>(B + 4 * i);
            *reinterpret_cast<float4*>(C + 4 * i) = make_float4(a.x * (b.x * s), a.y * (b.y * s), a.z * (b.z * s), a.w * (b.w * s));
        } else {
            for (int64_t j = 4 * i; j < n; ++j) C[j] = A[j] * (B[j] * s);
        }
    } else if (i < n) {
        C[i] = A[i] * (B[i] * s);
    }
}

// C = A * (s * B): PPNet's `hidden * gate`, gate = gamma * sigmoid(...) (ppnet.py:27, layers.py:318-320) with the GateNU's
// gamma folded in -- same rounding as scaling the gate first and multiplying then
extern "C" int swr_mul_scale_fwd(const float* A, const float* B, float scale, float* C, int64_t n, void* stream) {
    SWR_REQUIRE(A && B && C && n >= 0, SWR_ERR_ARG);
    if (n == 0) return SWR_OK;
    const int vec = swr_aligned16(A) && swr_aligned16(B) && swr_aligned16(C);
    const int64_t items = vec ? swr_ceil_div(n, 4) : n;
    hipLaunchKernelGGL(mul_scale_kernel, dim3(static_cast<unsigned>(swr_ceil_div(items, EW_THREADS))), dim3(EW_THREADS), 0,
                       static_cast<hipStream_t>(stream), A, B, scale, C, n, vec);
    return swr_launch_status();
}

// backward of C = A * (s * B) in one pass over dC: dA = dC * (s * B), dB = dC * (s * A) (the roundings of two swr_mul_scale_fwd
// calls; dC is read once, one launch)
__global__ __launch_bounds__(EW_THREADS) void mul_scale_bwd_kernel(const float* __restrict__ dC, const float* __restrict__ A,
                                                                   const float* __restrict__ B, float s, float* __restrict__ dA,
                                                                   float* __restrict__ dB, int64_t n, int vec) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * EW_THREADS + threadIdx.x;
    if (vec && 4 * i + 3 < n) {
        const float4 g = *reinterpret_cast<const float4*>(dC + 4 * i), a = *reinterpret_cast<const float4*>(A + 4 * i),
                     b = *reinterpret_cast<const float4*>(B + 4 * i);
        *reinterpret_cast<float4*>(dA + 4 * i) = make_float4(g.x * (b.x * s), g.y * (b.y * s), g.z * (b.z * s), g.w * (b.w * s));
        *reinterpret_cast<float4*>(dB + 4 * i) = make_float4(g.x * (a.x * s), g.y * (a.y * s), g.z * (a.z * s), g.w * (a.w * s));
    } else {
        const int64_t j0 = vec ? 4 * i : i, j1 = vec ? n : min<int64_t>(i + 1, n);
        for (int64_t j = j0; j < j1; ++j) {
            const float g = dC[j];
            dA[j] = g * (B[j] * s);
            dB[j] = g * (A[j] * s);
        }
    }
}

extern "C" int swr_mul_scale_bwd(const float* dC, const float* A, const float* B, float scale, float* dA, float* dB, int64_t n,
                                 void* stream) {
    SWR_REQUIRE(dC && A && B && dA && dB && n >= 0, SWR_ERR_ARG);
    if (n == 0) return SWR_OK;
    const int vec = swr_aligned16(dC) && swr_aligned16(A) && swr_aligned16(B) && swr_aligned16(dA) && swr_aligned16(dB);
    const int64_t items = vec ? swr_ceil_div(n, 4) : n;
    hipLaunchKernelGGL(mul_scale_bwd_kernel, dim3(static_cast<unsigned>(swr_ceil_div(items, EW_THREADS))), dim3(EW_THREADS), 0,
                       static_cast<hipStream_t>(stream), dC, A, B, scale, dA, dB, n, vec);
    return swr_launch_status();
}

// C = A * (s * sigmoid(Z)): a GateNU's output layer activation and the gating product in one pass (ppnet.py:27 with
// layers.py:318-320: `hidden * (gamma * sigmoid(z))`) -- the sigmoid of swr_affine_act_fwd and the product of
// swr_mul_scale_fwd, same roundings, without the gate tensor in between.  Backward: dA = dC * (s * y),
// dZ = ((dC * (s * A)) * y) * (1 - y), y = sigmoid(Z) (swr_mul_scale_bwd followed by the sigmoid rule of swr_act_bwd_apply).
__global__ __launch_bounds__(EW_THREADS) void mul_sigmoid_fwd_kernel(const float* __restrict__ A, const float* __restrict__ Z, float s,
                                                                     float* __restrict__ C, int64_t n, int vec) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * EW_THREADS + threadIdx.x;
    if (vec && 4 * i + 3 < n) {
        const float4 a = *reinterpret_cast<const float4*>(A + 4 * i), z = *reinterpret_cast<const float4*>(Z + 4 * i);
        *reinterpret_cast<float4*>(C + 4 * i) = make_float4(a.x * (swr_sigmoid(z.x) * s), a.y * (swr_sigmoid(z.y) * s),
                                                            a.z * (swr_sigmoid(z.z) * s), a.w * (swr_sigmoid(z.w) * s));
    } else {
        const int64_t j0 = vec ? 4 * i : i, j1 = vec ? n : min<int64_t>(i + 1, n);
        for (int64_t j = j0; j < j1; ++j) C[j] = A[j] * (swr_sigmoid(Z[j]) * s);
    }
}

__global__ __launch_bounds__(EW_THREADS) void mul_sigmoid_bwd_kernel(const float* __restrict__ dC, const float* __restrict__ A,
                                                                     const float* __restrict__ Z, float s, float* __restrict__ dA,
                                                                     float* __restrict__ dZ, int64_t n, int vec) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * EW_THREADS + threadIdx.x;
    const int64_t j0 = vec ? 4 * i : i, j1 = vec ? min<int64_t>(4 * i + 4, n) : min<int64_t>(i + 1, n);
    if (vec && 4 * i + 3 < n) {
        const float4 g = *reinterpret_cast<const float4*>(dC + 4 * i), a = *reinterpret_cast<const float4*>(A + 4 * i),
                     z = *reinterpret_cast<const float4*>(Z + 4 * i);
        const float y0 = swr_sigmoid(z.x), y1 = swr_sigmoid(z.y), y2 = swr_sigmoid(z.z), y3 = swr_sigmoid(z.w);
        *reinterpret_cast<float4*>(dA + 4 * i) = make_float4(g.x * (y0 * s), g.y * (y1 * s), g.z * (y2 * s), g.w * (y3 * s));
        *reinterpret_cast<float4*>(dZ + 4 * i) = make_float4((g.x * (a.x * s)) * y0 * (1.f - y0), (g.y * (a.y * s)) * y1 * (1.f - y1),
                                                             (g.z * (a.z * s)) * y2 * (1.f - y2), (g.w * (a.w * s)) * y3 * (1.f - y3));
    } else {
        for (int64_t j = j0; j < j1; ++j) {
            const float y = swr_sigmoid(Z[j]), g = dC[j];
            dA[j] = g * (y * s);
            dZ[j] = (g * (A[j] * s)) * y * (1.f - y);
        }
    }
}

extern "C" int swr_mul_sigmoid_fwd(const float* A, const float* Z, float scale, float* C, int64_t n, void* stream) {
    SWR_REQUIRE(A && Z && C && n >= 0, SWR_ERR_ARG);
    if (n == 0) return SWR_OK;
    const int vec = swr_aligned16(A) && swr_aligned16(Z) && swr_aligned16(C);
    const int64_t items = vec ? swr_ceil_div(n, 4) : n;
    hipLaunchKernelGGL(mul_sigmoid_fwd_kernel, dim3(static_cast<unsigned>(swr_ceil_div(items, EW_THREADS))), dim3(EW_THREADS), 0,
                       static_cast<hipStream_t>(stream), A, Z, scale, C, n, vec);
    return swr_launch_status();
}

extern "C" int swr_mul_sigmoid_bwd(const float* dC, const float* A, const float* Z, float scale, float* dA, float* dZ, int64_t n,
                                   void* stream) {
    SWR_REQUIRE(dC && A && Z && dA && dZ && n >= 0, SWR_ERR_ARG);
    if (n == 0) return SWR_OK;
    const int vec = swr_aligned16(dC) && swr_aligned16(A) && swr_aligned16(Z) && swr_aligned16(dA) && swr_aligned16(dZ);
    const int64_t items = vec ? swr_ceil_div(n, 4) : n;
    hipLaunchKernelGGL(mul_sigmoid_bwd_kernel, dim3(static_cast<unsigned>(swr_ceil_div(items, EW_THREADS))), dim3(EW_THREADS), 0,
                       static_cast<hipStream_t>(stream), dC, A, Z, scale, dA, dZ, n, vec);
    return swr_launch_status();
}

extern "C" int swr_mul_fwd(const float* A, const float* B, float* C, int64_t n, void* stream) {
    SWR_REQUIRE(A && B && C && n >= 0, SWR_ERR_ARG);
    if (n == 0) return SWR_OK;
    hipLaunchKernelGGL(mul_kernel, dim3(static_cast<unsigned>(swr_ceil_div(n, EW_THREADS))), dim3(EW_THREADS), 0,
                       static_cast<hipStream_t>(stream), A, B, C, n);
    return swr_launch_status();
}

// column sums over the batch: 64 columns x 4 row phases per block over a 256-row tile, then a fixed-order
// fp64 sum over tiles
#define CS_TILE 256
__global__ __launch_bounds__(EW_THREADS) void colsum_partial_kernel(const float* __restrict__ X, int64_t ldx, int64_t M, int N,
                                                                    float* __restrict__ part) {
    __shared__ float s[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int n = blockIdx.y * 64 + cx;
    const int64_t m0 = static_cast<int64_t>(blockIdx.x) * CS_TILE;
    float a = 0.f;
    if (n < N)
        for (int r = ry; r < CS_TILE && m0 + r < M; r += 4) a += X[(m0 + r) * ldx + n];
    s[ry][cx] = a;
    __syncthreads();
    if (ry == 0 && n < N) part[static_cast<int64_t>(blockIdx.x) * N + n] = (s[0][cx] + s[1][cx]) + (s[2][cx] + s[3][cx]);
}

// 32 lanes per column (lane l takes tiles l, l + 32, ... in order, 4 loads in flight), combined with a fixed butterfly:
// deterministic, and a long batch (HAMUR's per-sample factors: thousands of tiles) is no longer one serial chain per
// column (was 226 us per launch at 4 480 tiles)
#define CSF_LANES 32
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int n_tiles, int N, float* out,
                                                           int accumulate) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const int n = tid / CSF_LANES, l = tid % CSF_LANES;
    double acc = 0.0;
    if (n < N) {
        int t = l;
        for (; t + 3 * CSF_LANES < n_tiles; t += 4 * CSF_LANES) {
            const float v0 = part[static_cast<int64_t>(t) * N + n], v1 = part[static_cast<int64_t>(t + CSF_LANES) * N + n],
                        v2 = part[static_cast<int64_t>(t + 2 * CSF_LANES) * N + n],
                        v3 = part[static_cast<int64_t>(t + 3 * CSF_LANES) * N + n];
            acc += v0; acc += v1; acc += v2; acc += v3;
        }
        for (; t < n_tiles; t += CSF_LANES) acc += part[static_cast<int64_t>(t) * N + n];
    }
#pragma unroll
    for (int off = 1; off < CSF_LANES; off <<= 1) acc += __shfl_xor(acc, off);
    if (n < N && l == 0) out[n] = (accumulate ? out[n] : 0.f) + static_cast<float>(acc);
}

extern "C" size_t swr_colsum_workspace_bytes(int64_t M, int N) {
    return static_cast<size_t>(swr_ceil_div(M > 0 ? M : 1, CS_TILE)) * (N > 0 ? N : 1) * 4 + 256;
}

extern "C" int swr_colsum(const float* X, int64_t ldx, int64_t M, int N, float* out, int accumulate, void* workspace,
                          size_t workspace_bytes, void* stream) {
    SWR_REQUIRE(X && out && workspace && M > 0 && N > 0 && ldx >= N, SWR_ERR_ARG);
    SWR_REQUIRE(workspace_bytes >= swr_colsum_workspace_bytes(M, N), SWR_ERR_WORKSPACE);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nt = static_cast<int>(swr_ceil_div(M, CS_TILE));
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nt, static_cast<unsigned>(swr_ceil_div(N, 64))), dim3(EW_THREADS), 0, st, X,
                       ldx, M, N, static_cast<float*>(workspace));
    hipLaunchKernelGGL(colsum_final_kernel, dim3(static_cast<unsigned>(swr_ceil_div(static_cast<int64_t>(N) * CSF_LANES, 256))),
                       dim3(256), 0, st, static_cast<const float*>(workspace), nt, N,
                       out, accumulate);
    return swr_launch_status();
}

// ------------------------------------------------------------------------------------------- misc
__global__ void swr_zero_kernel(uint4* p, size_t n16, unsigned char* tail, size_t ntail) {
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += stride) p[i] = make_uint4(0, 0, 0, 0);
    if (blockIdx.x == 0 && threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

int swr_zero_async(void* p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return SWR_OK;
    if (!swr_aligned16(p)) return SWR_ERR_ALIGN;
    const size_t n16 = bytes / 16, ntail = bytes % 16;
    size_t blocks = (n16 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(swr_zero_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, static_cast<uint4*>(p), n16,
                       static_cast<unsigned char*>(p) + n16 * 16, ntail);
    return swr_launch_status();
}

// zero_grad of the gradient arena (SwrModule.zero_grad) as a kernel node of this library, like every other zero-fill
extern "C" int swr_zero(void* p, size_t bytes, void* stream) {
    SWR_REQUIRE(p != nullptr || bytes == 0, SWR_ERR_ARG);
    return swr_zero_async(p, bytes, static_cast<hipStream_t>(stream));
}

// Stream-skew harness (tests/test_skew_gpu.py, SWR_SKEW): one wave that occupies its stream for `us` microseconds
// (wall_clock64 ticks at 100 MHz).  Injected at the fork points of the step (ops._skew) it stretches one branch of the
// stream graph against the others: a missing cross-stream edge then shows as a changed bit in the results.
__global__ void swr_spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

extern "C" int swr_spin_us(int us, void* stream) {
    SWR_REQUIRE(us >= 0 && us <= 100000, SWR_ERR_ARG);
    if (us == 0) return SWR_OK;
    hipLaunchKernelGGL(swr_spin_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), static_cast<long long>(us) * 100);
    return swr_launch_status();
}

// Schedule instrument (SWR_STAMPS, ops._stamp): one lane writes the 100 MHz wall clock where its stream has got to.  A kernel
// trace under rocprofv3 intercepts the queues and moves the branches of a replayed graph against each other; this does not.
__global__ void swr_stamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); }

extern "C" int swr_stamp(unsigned long long* slot, void* stream) {
    SWR_REQUIRE(slot != nullptr, SWR_ERR_ARG);
    hipLaunchKernelGGL(swr_stamp_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), slot);
    return swr_launch_status();
}

extern "C" int swr_abi_version(void) { return SWR_ABI_VERSION; }

extern "C" const char* swr_status_str(int status) {
    switch (status) {
        case SWR_OK: return "ok";
        case SWR_ERR_ARG: return "invalid argument";
        case SWR_ERR_DTYPE: return "unsupported dtype";
        case SWR_ERR_ALIGN: return "misaligned pointer or leading dimension";
        case SWR_ERR_LAUNCH: return "HIP launch/runtime error";
        case SWR_ERR_UNSUPPORTED: return "shape outside the implemented envelope";
        case SWR_ERR_WORKSPACE: return "workspace too small";
        default: return "unknown status";
    }
}

extern "C" int swr_device_available(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n > 0 ? 1 : 0;
}
