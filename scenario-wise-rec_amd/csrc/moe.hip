// Gate mixing, domain select, BCE and small elementwise helpers.
// All kernels are streaming passes over [M, *] fp32 row-major data (HBM-bound); every reduction that
// crosses workgroups goes through per-block partials summed in a fixed order (deterministic).
#include "common.h"

#define EW_THREADS 256

// ------------------------------------------------------------------------------------- gate mixing
// expert_pooling = sum_j gate_j * expert_j   (mmoe.py:48-49, ple.py:121-126,131-133)
__global__ __launch_bounds__(EW_THREADS) void mix_fwd_kernel(const swr_mix_desc d, const float* __restrict__ Y, int64_t ldy,
                                                             float* __restrict__ P, int64_t ldp, int64_t M) {
    const int width = d.n_out * d.H;
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * EW_THREADS + threadIdx.x;
    const int64_t m = idx / width;
    if (m >= M) return;
    const int c = static_cast<int>(idx - m * width);
    const int o = c / d.H, h = c - o * d.H;
    const float* y = Y + m * ldy;
    float acc = 0.f;
    for (int j = 0; j < d.n_sel; ++j)
        acc = fmaf(y[d.g_col + o * d.g_stride + j], y[d.x_col + d.sel[o][j] * d.H + h], acc);
    P[m * ldp + c] = acc;
}

extern "C" int swr_moe_mix_fwd(const swr_mix_desc* desc, const float* Y, int64_t ldy, float* P, int64_t ldp, int64_t M,
                               void* stream) {
    SWR_REQUIRE(desc && Y && P && M >= 0, SWR_ERR_ARG);
    SWR_REQUIRE(desc->n_out > 0 && desc->n_out <= SWR_MIX_MAX_OUT && desc->n_sel > 0 && desc->n_sel <= SWR_MIX_MAX_SEL &&
                    desc->H > 0, SWR_ERR_ARG);
    if (M == 0) return SWR_OK;
    const int64_t n = M * desc->n_out * desc->H;
    hipLaunchKernelGGL(mix_fwd_kernel, dim3(static_cast<unsigned>(swr_ceil_div(n, EW_THREADS))), dim3(EW_THREADS), 0,
                       static_cast<hipStream_t>(stream), *desc, Y, ldy, P, ldp, M);
    return swr_launch_status();
}

// dX[e, h] = sum over (o, j) with sel[o][j] == e of gate[o][j] * dP[o, h];  dG[o][j] = sum_h dP[o, h] X[sel[o][j], h]
__global__ __launch_bounds__(EW_THREADS) void mix_bwd_kernel(const swr_mix_desc d, int n_expert, const float* __restrict__ dP,
                                                             int64_t lddp, const float* __restrict__ Y, int64_t ldy,
                                                             float* __restrict__ dY, int64_t lddy, int accumulate, int64_t M) {
    const int wx = n_expert * d.H;            // expert columns
    const int wg = d.n_out * d.n_sel;         // gate columns
    const int width = wx + wg;
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * EW_THREADS + threadIdx.x;
    const int64_t m = idx / width;
    if (m >= M) return;
    const int c = static_cast<int>(idx - m * width);
    const float* y = Y + m * ldy;
    const float* dp = dP + m * lddp;
    float acc = 0.f;
    int col;
    if (c < wx) {
        const int e = c / d.H, h = c - e * d.H;
        for (int o = 0; o < d.n_out; ++o)
            for (int j = 0; j < d.n_sel; ++j)
                if (d.sel[o][j] == e) acc = fmaf(y[d.g_col + o * d.g_stride + j], dp[o * d.H + h], acc);
        col = d.x_col + c;
    } else {
        const int gidx = c - wx;
        const int o = gidx / d.n_sel, j = gidx - o * d.n_sel;
        const float* x = y + d.x_col + d.sel[o][j] * d.H;
        for (int h = 0; h < d.H; ++h) acc = fmaf(dp[o * d.H + h], x[h], acc);
        col = d.g_col + o * d.g_stride + j;
    }
    float* dst = dY + m * lddy + col;
    *dst = accumulate ? *dst + acc : acc;
}

extern "C" int swr_moe_mix_bwd(const swr_mix_desc* desc, const float* dP, int64_t lddp, const float* Y, int64_t ldy,
                               float* dY, int64_t lddy, int accumulate, int64_t M, void* stream) {
    SWR_REQUIRE(desc && dP && Y && dY && M >= 0, SWR_ERR_ARG);
    SWR_REQUIRE(desc->n_out > 0 && desc->n_out <= SWR_MIX_MAX_OUT && desc->n_sel > 0 && desc->n_sel <= SWR_MIX_MAX_SEL &&
                    desc->H > 0, SWR_ERR_ARG);
    if (M == 0) return SWR_OK;
    int n_expert = 0;
    for (int o = 0; o < desc->n_out; ++o)
        for (int j = 0; j < desc->n_sel; ++j)
            if (desc->sel[o][j] + 1 > n_expert) n_expert = desc->sel[o][j] + 1;
    const int64_t n = M * (static_cast<int64_t>(n_expert) * desc->H + desc->n_out * desc->n_sel);
    hipLaunchKernelGGL(mix_bwd_kernel, dim3(static_cast<unsigned>(swr_ceil_div(n, EW_THREADS))), dim3(EW_THREADS), 0,
                       static_cast<hipStream_t>(stream), *desc, n_expert, dP, lddp, Y, ldy, dY, lddy, accumulate, M);
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

#define BCE_PER_BLOCK 4096
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
    __shared__ double sm[EW_THREADS];
    double acc = 0.0;
    const int per = (n_part + EW_THREADS - 1) / EW_THREADS;
    for (int t = threadIdx.x * per; t < min((threadIdx.x + 1) * per, n_part); ++t) acc += part[t];
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 1; st < EW_THREADS; st <<= 1) {
        if ((threadIdx.x & (2 * st - 1)) == 0) sm[threadIdx.x] += sm[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = static_cast<float>(sm[0] / static_cast<double>(M));
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

// ------------------------------------------------------------------------------------- elementwise
__global__ __launch_bounds__(EW_THREADS) void mul_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                         float* __restrict__ C, int64_t n) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * EW_THREADS + threadIdx.x;
    if (i < n) C[i] = A[i] * B[i];
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

__global__ void colsum_final_kernel(const float* __restrict__ part, int n_tiles, int N, float* out, int accumulate) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double acc = 0.0;
    for (int t = 0; t < n_tiles; ++t) acc += part[static_cast<int64_t>(t) * N + n];
    out[n] = (accumulate ? out[n] : 0.f) + static_cast<float>(acc);
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
    hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 255) / 256), dim3(256), 0, st, static_cast<const float*>(workspace), nt, N,
                       out, accumulate);
    return swr_launch_status();
}

// ------------------------------------------------------------------------------------------- misc
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
