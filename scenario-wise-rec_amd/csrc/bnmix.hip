// BatchNorm + activation + gate mixing of an MMoE level in one pass each way (mmoe.py:44-49):
//
//   Z[:, 0 : ne*H]            pre-BN outputs of the ne experts' last Linear           (ReLU experts)
//   Z[:, ne*H : ne*H + D*ne]  pre-BN outputs of the D gates' Linear(., ne)            (softmax over the ne experts)
//   P[:, o*H : (o+1)*H] = sum_j softmax(bn(Z_gate_o))[j] * relu(bn(Z_expert_j))
//
// The layer-wise path streams [B, 148] activations five times forward and eleven times backward (affine + activation,
// mix, mix backward, BN statistics, BN apply: 550 MB at B = 65 536).  Here the activated experts and the gate
// probabilities are never stored:
//
//   bnmix_fwd : reads Z, writes P                                                      (81 MB)
//   bnmix_bwd : reads dP and Z, recomputes the activations, writes dY = dL/d(BN output) and the per-64-row-tile
//               (sum dY, sum dY * xhat) pairs for swr_bn_bwd_finalize                  (120 MB)
//   then swr_act_bwd_apply (no activation) turns dY into dZ                            (117 MB)
//
// Thread = 4 hidden columns h of one row for ALL experts and outputs (H/4 lanes per row); the gate gradients are
// completed by a butterfly over those lanes; the gates' softmax is evaluated redundantly by the lanes of a row (20
// values) rather than exchanged.  Identity selection only (every output mixes every expert, in order).
#include "common.h"

#ifndef BM_ROWS
#define BM_ROWS 32            // rows per workgroup = one tile of partial sums for swr_bn_bwd_finalize
#endif
// waves per SIMD the backward kernel is compiled for.  The kernel needs 157 VGPRs; capped at 128 (4 waves per SIMD, what two
// 64-row workgroups per CU would be) hipcc SPILLS 27 registers per lane to scratch: 57 MB written + 57 MB read back per launch
// at batch 65 536, the "2.1 x algorithmic" HBM traffic of profiles/r03_f_pmc_hbm.json.  At 3 waves per SIMD (cap 168) nothing
// spills, and 32-row workgroups (4 waves at H = 32, 39 KB of LDS) let three of them share a CU: 12 waves per CU where a
// 64-row workgroup under the same cap leaves 8.  Measured in one run at batch 65 536, config 2 (tools/micro/bnmix_time.py):
// 64 rows / cap 128: 48.7 us;  64 rows / cap 256: 35.0 us;  32 rows / cap 168: 30.0 us (step 0.4309 -> 0.4140 ms).
#ifndef SWR_BM_BWD_WAVES
#define SWR_BM_BWD_WAVES 3
#endif
#define BM_BWD_WAVES(NE, DD) ((NE) <= 4 && (DD) <= 5 ? SWR_BM_BWD_WAVES : ((NE) <= 4 ? 2 : 1))     // six outputs: > 168 registers; eight experts: > 256 (48 bytes per lane of scratch at two waves)
#define BM_MAX_D 8
#define BM_MAX_G 32           // D * ne

struct BnMixK {
    swr_bnmix_args a;
    int h4n;                  // H / 4: lanes per row
    int n_cols;               // ne*H + D*ne
};

__device__ __forceinline__ float4 ldf4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 affine4(float4 z, float4 sc, float4 sh) {
    return make_float4(fmaf(sc.x, z.x, sh.x), fmaf(sc.y, z.y, sh.y), fmaf(sc.z, z.z, sh.z), fmaf(sc.w, z.w, sh.w));
}
__device__ __forceinline__ float4 relu4(float4 v) {
    return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
}

// sum over the H4N (4 or 8) adjacent lanes of a row, every lane gets it: DPP moves instead of ds_bpermute round trips.
// Same additions as the xor-1, -2, -4 butterfly (the last step pairs lane i with 7 - i, whose quad already holds the other
// quad's sum: own + other either way), so the bits are those of the butterfly.
template <int H4N>
__device__ __forceinline__ float row_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    if (H4N == 8) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    return v;
}

// gate probabilities of one row: g[o*NE + j] = softmax_j(scale * z + shift), same operation order as act_fwd4 (bn.hip)
template <int NE, int DD, bool EXACT>
__device__ __forceinline__ void gate_probs(const swr_bnmix_args& a, const float* __restrict__ zg, const float* __restrict__ gsc,
                                           const float* __restrict__ gsh, float (&g)[DD * NE]) {
    const int ne = EXACT ? NE : a.ne, D = EXACT ? DD : a.D;
#pragma unroll
    for (int o = 0; o < DD; ++o) {
        if (EXACT || o < D) {
            float v[NE];
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                v[j] = (EXACT || j < ne) ? fmaf(gsc[o * ne + j], zg[o * ne + j], gsh[o * ne + j]) : -INFINITY;
                mx = fmaxf(mx, v[j]);
            }
            float den = 0.f;
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                v[j] = (EXACT || j < ne) ? expf(v[j] - mx) : 0.f;
                den += v[j];
            }
#pragma unroll
            for (int j = 0; j < NE; ++j) g[o * NE + j] = v[j] / den;
        }
    }
}

// ... or read back what the forward pass saved (20 floats per row at the KuaiRand config): no exp / divide in the backward
template <int NE, int DD, bool EXACT>
__device__ __forceinline__ void gate_probs_saved(const swr_bnmix_args& a, const float* __restrict__ gs, float (&g)[DD * NE]) {
    const int ne = EXACT ? NE : a.ne, D = EXACT ? DD : a.D;
    if constexpr (EXACT && (DD * NE) % 4 == 0) {                  // rows of 16-byte multiples (swr_bnmix_bwd checks the base)
#pragma unroll
        for (int q = 0; q < DD * NE / 4; ++q) {
            const float4 v = ldf4(gs + 4 * q);
            g[4 * q] = v.x; g[4 * q + 1] = v.y; g[4 * q + 2] = v.z; g[4 * q + 3] = v.w;
        }
        return;
    }
#pragma unroll
    for (int o = 0; o < DD; ++o)
#pragma unroll
        for (int j = 0; j < NE; ++j)
            if (EXACT || (o < D && j < ne)) g[o * NE + j] = gs[o * ne + j];
}

// ------------------------------------------------------------------------------------------- forward
template <int NE, int DD, bool EXACT, int H4N>
__global__ __launch_bounds__(256) void bnmix_fwd_kernel(const BnMixK kk) {
    const swr_bnmix_args& a = kk.a;
    const int ne = EXACT ? NE : a.ne, D = EXACT ? DD : a.D;
    constexpr int h4n = H4N;                                     // compile time: the lane / row arithmetic is shifts
    const int lane = threadIdx.x % h4n;
    const int h = lane * 4;
    const int rows_per_pass = 256 / h4n;
    const int gc = ne * a.H;                                     // first gate column
    float4 sc[NE], sh[NE];
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        if (EXACT || j < ne) {
            sc[j] = ldf4(a.scale + j * a.H + h);
            sh[j] = ldf4(a.shift + j * a.H + h);
        }
    }
    for (int64_t m = static_cast<int64_t>(blockIdx.x) * rows_per_pass + threadIdx.x / h4n; m < a.M;
         m += static_cast<int64_t>(gridDim.x) * rows_per_pass) {
        const float* __restrict__ z = a.Z + m * a.ldz;
        float4 x[NE];
#pragma unroll
        for (int j = 0; j < NE; ++j)
            if (EXACT || j < ne) x[j] = relu4(affine4(ldf4(z + j * a.H + h), sc[j], sh[j]));
        float g[DD * NE];
        gate_probs<NE, DD, EXACT>(a, z + gc, a.scale + gc, a.shift + gc, g);
        if (a.G) {                                               // keep the probabilities for the backward pass
            float* __restrict__ gs = a.G + m * (D * ne);
            if constexpr (EXACT) {
                // lane l keeps columns l, l + H4N, ...: one select chain and ONE store per group of H4N columns (a store
                // per (o, j) under its own lane predicate was 20 exec-masked stores per thread)
                constexpr int NQ = (DD * NE + H4N - 1) / H4N;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    float v = 0.f;
#pragma unroll
                    for (int u = 0; u < H4N; ++u)
                        if (q * H4N + u < DD * NE) v = lane == u ? g[(q * H4N + u < DD * NE) ? q * H4N + u : 0] : v;
                    const int c = q * H4N + lane;
                    if (c < DD * NE) gs[c] = v;
                }
            } else {
#pragma unroll
                for (int o = 0; o < DD; ++o)
#pragma unroll
                    for (int j = 0; j < NE; ++j)
                        if ((o < D && j < ne) && (o * ne + j) % h4n == lane) gs[o * ne + j] = g[o * NE + j];
            }
        }
        float* __restrict__ p = a.P + m * a.ldp + h;
#pragma unroll
        for (int o = 0; o < DD; ++o) {
            if (EXACT || o < D) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < NE; ++j) {
                    if (EXACT || j < ne) {
                        const float gj = g[o * NE + j];
                        acc.x = fmaf(gj, x[j].x, acc.x); acc.y = fmaf(gj, x[j].y, acc.y);
                        acc.z = fmaf(gj, x[j].z, acc.z); acc.w = fmaf(gj, x[j].w, acc.w);
                    }
                }
                *reinterpret_cast<float4*>(p + o * a.H) = acc;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ backward
// workgroup = one BM_ROWS-row tile, BM_ROWS * H/4 threads.  LDS: the tile's dY and dY * xhat, [BM_ROWS][n_cols + 4] floats each.
template <int NE, int DD, bool EXACT, int H4N>
__global__ __launch_bounds__(BM_ROWS * H4N, BM_BWD_WAVES(NE, DD)) void bnmix_bwd_kernel(const BnMixK kk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const swr_bnmix_args& a = kk.a;
    const int ne = EXACT ? NE : a.ne, D = EXACT ? DD : a.D;
    constexpr int h4n = H4N;
    const int N = kk.n_cols, P = N + 4;
    float* t1 = lds;                     // dY
    float* t2 = lds + BM_ROWS * P;       // dY * xhat
    const int lane = threadIdx.x % h4n, r = threadIdx.x / h4n;
    const int h = lane * 4;
    const int gc = ne * a.H;
    const int64_t m0 = static_cast<int64_t>(blockIdx.x) * BM_ROWS;
    const int64_t m = m0 + r;
    const bool valid = m < a.M;
    const int64_t mm = valid ? m : a.M - 1;                      // surplus rows redo the last one (the butterfly needs
    const float* __restrict__ z = a.Z + mm * a.ldz;              // every lane) and contribute nothing
    // ---- recompute the activations
    float4 ze[NE], x[NE];
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        if (EXACT || j < ne) {
            ze[j] = ldf4(z + j * a.H + h);
            x[j] = relu4(affine4(ze[j], ldf4(a.scale + j * a.H + h), ldf4(a.shift + j * a.H + h)));
        }
    }
    float g[DD * NE];
    if (a.G)
        gate_probs_saved<NE, DD, EXACT>(a, a.G + mm * (D * ne), g);
    else
        gate_probs<NE, DD, EXACT>(a, z + gc, a.scale + gc, a.shift + gc, g);
    // ---- dL/dx_j = sum_o g[o][j] dP_o (then the ReLU mask); dL/dg[o][j] = <dP_o, x_j> over all H columns
    float4 dx[NE];
#pragma unroll
    for (int j = 0; j < NE; ++j) dx[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    float dg[DD * NE];
#pragma unroll
    for (int o = 0; o < DD; ++o) {
        if (EXACT || o < D) {
            const float4 v = ldf4(a.dP + mm * a.lddp + o * a.H + h);
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                if (EXACT || j < ne) {
                    const float gj = g[o * NE + j];
                    dx[j].x = fmaf(gj, v.x, dx[j].x); dx[j].y = fmaf(gj, v.y, dx[j].y);
                    dx[j].z = fmaf(gj, v.z, dx[j].z); dx[j].w = fmaf(gj, v.w, dx[j].w);
                    float pd = (v.x * x[j].x + v.y * x[j].y) + (v.z * x[j].z + v.w * x[j].w);
                    dg[o * NE + j] = row_sum<H4N>(pd);
                }
            }
        }
    }
    // ---- the gate columns this lane will finish (q * H4N + lane): their pre-activations and statistics are fetched
    // here, unconditionally and together, and land under the expert columns' section (they used to sit inside 20
    // divergent single-lane branches, each exposing a full load latency)
    constexpr int NQ = (DD * NE + H4N - 1) / H4N;
    float zq[NQ], muq[NQ], rsq[NQ];
    if constexpr (EXACT) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c = gc + min(q * H4N + lane, DD * NE - 1);
            zq[q] = z[c]; muq[q] = a.mean[c]; rsq[q] = a.rstd[c];
        }
    }
    // ---- expert columns: dY = relu'(.) dx; stage dY and dY * xhat, write dY
    float* row1 = t1 + r * P;
    float* row2 = t2 + r * P;
    // The statistics of expert j + 1 are requested BEFORE expert j's dY store: the compiler may not move a load of a.mean
    // above a store to a.dY (they could alias), so in the plain loop every iteration began with an exposed load latency.
    float4 mu = ldf4(a.mean + h), rs = ldf4(a.rstd + h);
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        if (EXACT || j < ne) {
            const int c = j * a.H + h;
            float4 d = make_float4(x[j].x > 0.f ? dx[j].x : 0.f, x[j].y > 0.f ? dx[j].y : 0.f, x[j].z > 0.f ? dx[j].z : 0.f,
                                   x[j].w > 0.f ? dx[j].w : 0.f);
            if (!valid) d = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 mu_c = mu, rs_c = rs;
            if (j + 1 < NE && (EXACT || j + 1 < ne)) {
                mu = ldf4(a.mean + c + a.H);
                rs = ldf4(a.rstd + c + a.H);
            }
            *reinterpret_cast<float4*>(row1 + c) = d;
            *reinterpret_cast<float4*>(row2 + c) =
                make_float4(d.x * ((ze[j].x - mu_c.x) * rs_c.x), d.y * ((ze[j].y - mu_c.y) * rs_c.y),
                            d.z * ((ze[j].z - mu_c.z) * rs_c.z), d.w * ((ze[j].w - mu_c.w) * rs_c.w));
        }
    }
    // ---- gate columns: softmax backward dZg[o][j] = g (dg - <dg_o, g_o>); lane (o*ne + j) % h4n keeps column o*ne + j
    if constexpr (EXACT) {
        float dgz[DD * NE];
#pragma unroll
        for (int o = 0; o < DD; ++o) {
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < NE; ++j) dot = fmaf(dg[o * NE + j], g[o * NE + j], dot);
#pragma unroll
            for (int j = 0; j < NE; ++j) dgz[o * NE + j] = g[o * NE + j] * (dg[o * NE + j] - dot);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float d = 0.f;
#pragma unroll
            for (int u = 0; u < H4N; ++u)
                if (q * H4N + u < DD * NE) d = lane == u ? dgz[(q * H4N + u < DD * NE) ? q * H4N + u : 0] : d;
            if (!valid) d = 0.f;
            const int cq = q * H4N + lane;
            if (cq < DD * NE) {
                const int c = gc + cq;
                row1[c] = d;
                row2[c] = d * ((zq[q] - muq[q]) * rsq[q]);
            }
        }
    } else
#pragma unroll
    for (int o = 0; o < DD; ++o) {
        if (o < D) {
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < NE; ++j)
                if (EXACT || j < ne) dot = fmaf(dg[o * NE + j], g[o * NE + j], dot);
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                if ((EXACT || j < ne) && (o * ne + j) % h4n == lane) {
                    const int c = gc + o * ne + j;
                    const float d = valid ? g[o * NE + j] * (dg[o * NE + j] - dot) : 0.f;
                    row1[c] = d;
                    row2[c] = d * ((z[c] - a.mean[c]) * a.rstd[c]);
                    }
            }
        }
    }
    __syncthreads();
    // ---- dY leaves from the LDS tile, row by row in 16-byte pieces of consecutive lanes: the tile's 64 rows are one
    // contiguous block of dY when lddy == N.  (Written from the registers -- 128 bytes per (row, expert) at a 592-byte row
    // pitch, the gate columns float by float -- the partially written lines left the L2 more than once: 100 MB of HBM
    // writes for a 39 MB tensor, profiles/r04_a_pmc_hbm.txt.)
    {
        const int n4 = N >> 2;                                  // N % 4 == 0 (checked by the launcher)
        const int rows_here = static_cast<int>(min<int64_t>(BM_ROWS, a.M - m0));
        for (int item = threadIdx.x; item < rows_here * n4; item += blockDim.x) {
            const int rr = item / n4, c4 = item - rr * n4;
            *reinterpret_cast<float4*>(a.dY + (m0 + rr) * a.lddy + 4 * c4) = *reinterpret_cast<const float4*>(t1 + rr * P + 4 * c4);
        }
    }
    // ---- column sums over the 64 rows: 4 threads per column (rows i*4 + part), fixed order
    for (int item = threadIdx.x; item < N * 4; item += blockDim.x) {
        const int c = item >> 2, part = item & 3;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < BM_ROWS / 4; ++i) {
            s1 += t1[(i * 4 + part) * P + c];
            s2 += t2[(i * 4 + part) * P + c];
        }
        s1 += __shfl_xor(s1, 1); s2 += __shfl_xor(s2, 1);
        s1 += __shfl_xor(s1, 2); s2 += __shfl_xor(s2, 2);
        if (part == 0) {
            float* p = a.bn_partials + (static_cast<int64_t>(blockIdx.x) * N + c) * 2;
            p[0] = s1;
            p[1] = s2;
        }
    }
}

// ------------------------------------------------------------------------------------------- dispatch
static bool bnmix_ok(int ne, int H, int D) {
    return ne >= 1 && ne <= 8 && D >= 1 && D <= BM_MAX_D && D * ne <= BM_MAX_G && (H == 16 || H == 32);   // H = 64: 1024 threads per tile need <= 128 VGPRs
}

extern "C" int swr_bnmix_supported(int ne, int H, int D) { return bnmix_ok(ne, H, D) ? 1 : 0; }
extern "C" int swr_bnmix_tile_rows(void) { return BM_ROWS; }

static int bnmix_common(const swr_bnmix_args* args, BnMixK& kk) {
    SWR_REQUIRE(args != nullptr, SWR_ERR_ARG);
    const swr_bnmix_args& a = *args;
    SWR_REQUIRE(a.M > 0 && bnmix_ok(a.ne, a.H, a.D), SWR_ERR_UNSUPPORTED);
    kk.a = a;
    kk.h4n = a.H / 4;
    kk.n_cols = a.ne * a.H + a.D * a.ne;
    SWR_REQUIRE(a.Z && a.scale && a.shift && a.ldz >= kk.n_cols, SWR_ERR_ARG);
    SWR_REQUIRE(a.ldz % 4 == 0 && swr_aligned16(a.Z) && swr_aligned16(a.scale) && swr_aligned16(a.shift), SWR_ERR_ALIGN);
    return SWR_OK;
}

// exact (compile-time) expert / output counts for the common small configurations, predicated kernels otherwise
#define BM_EXACT_LIST(X) X(2, 2) X(2, 3) X(3, 2) X(3, 3) X(4, 2) X(4, 3) X(4, 4) X(4, 5) X(4, 6) X(8, 5)

template <int H4N>
static void bnmix_launch_fwd_h(const BnMixK& kk, dim3 grid, hipStream_t st) {
    const int key = kk.a.ne * 16 + kk.a.D;
    switch (key) {
#define X(NEV, DV) case NEV * 16 + DV: hipLaunchKernelGGL((bnmix_fwd_kernel<NEV, DV, true, H4N>), grid, dim3(256), 0, st, kk); return;
        BM_EXACT_LIST(X)
#undef X
        default: break;
    }
    if (kk.a.ne <= 4)
        hipLaunchKernelGGL((bnmix_fwd_kernel<4, BM_MAX_D, false, H4N>), grid, dim3(256), 0, st, kk);
    else
        hipLaunchKernelGGL((bnmix_fwd_kernel<8, BM_MAX_D, false, H4N>), grid, dim3(256), 0, st, kk);
}

static void bnmix_launch_fwd(const BnMixK& kk, dim3 grid, hipStream_t st) {
    if (kk.h4n == 8) bnmix_launch_fwd_h<8>(kk, grid, st);
    else bnmix_launch_fwd_h<4>(kk, grid, st);
}

template <int H4N>
static const void* bnmix_bwd_fn_h(int ne, int D) {
    const int key = ne * 16 + D;
    switch (key) {
#define X(NEV, DV) case NEV * 16 + DV: return reinterpret_cast<const void*>(bnmix_bwd_kernel<NEV, DV, true, H4N>);
        BM_EXACT_LIST(X)
#undef X
        default: break;
    }
    return ne <= 4 ? reinterpret_cast<const void*>(bnmix_bwd_kernel<4, BM_MAX_D, false, H4N>)
                   : reinterpret_cast<const void*>(bnmix_bwd_kernel<8, BM_MAX_D, false, H4N>);
}

static const void* bnmix_bwd_fn(int ne, int D, int h4n) {
    return h4n == 8 ? bnmix_bwd_fn_h<8>(ne, D) : bnmix_bwd_fn_h<4>(ne, D);
}

extern "C" int swr_bnmix_fwd(const swr_bnmix_args* args, void* stream) {
    BnMixK kk;
    const int rc = bnmix_common(args, kk);
    if (rc != SWR_OK) return rc;
    const swr_bnmix_args& a = kk.a;
    SWR_REQUIRE(a.P && a.ldp >= a.D * a.H, SWR_ERR_ARG);
    SWR_REQUIRE(a.ldp % 4 == 0 && swr_aligned16(a.P), SWR_ERR_ALIGN);
    const int rows_per_pass = 256 / kk.h4n;
    const unsigned grid = static_cast<unsigned>(std::min<int64_t>(swr_ceil_div(a.M, rows_per_pass), 4096));
    bnmix_launch_fwd(kk, dim3(grid), static_cast<hipStream_t>(stream));
    return swr_launch_status();
}

extern "C" int swr_bnmix_bwd(const swr_bnmix_args* args, void* stream) {
    BnMixK kk;
    const int rc = bnmix_common(args, kk);
    if (rc != SWR_OK) return rc;
    const swr_bnmix_args& a = kk.a;
    SWR_REQUIRE(a.dP && a.lddp >= a.D * a.H && a.mean && a.rstd && a.dY && a.lddy >= kk.n_cols && a.bn_partials && kk.n_cols % 4 == 0, SWR_ERR_ARG);
    SWR_REQUIRE(a.lddp % 4 == 0 && a.lddy % 4 == 0 && swr_aligned16(a.dP) && swr_aligned16(a.dY) && swr_aligned16(a.mean) &&
                    swr_aligned16(a.rstd) && (a.G == nullptr || swr_aligned16(a.G)), SWR_ERR_ALIGN);
    const unsigned grid = static_cast<unsigned>(swr_ceil_div(a.M, BM_ROWS));
    const unsigned threads = static_cast<unsigned>(BM_ROWS * kk.h4n);
    const size_t lds = 2 * static_cast<size_t>(BM_ROWS) * (kk.n_cols + 4) * sizeof(float);
    SWR_REQUIRE(lds <= 159 * 1024, SWR_ERR_UNSUPPORTED);
    const void* fn = bnmix_bwd_fn(a.ne, a.D, kk.h4n);
    // more than 64 KB of dynamic LDS needs the attribute (idempotent, not a stream operation); the first call of a
    // configuration happens in a warm-up step, before any hipGraph capture
    if (lds > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) != hipSuccess)
        return SWR_ERR_LAUNCH;
    void* kargs[] = {const_cast<BnMixK*>(&kk)};
    if (hipLaunchKernel(fn, dim3(grid), dim3(threads), kargs, lds, static_cast<hipStream_t>(stream)) != hipSuccess)
        return SWR_ERR_LAUNCH;
    return swr_launch_status();
}
