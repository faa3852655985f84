// BatchNorm1d statistics / affine + activation forward and backward (basic/layers.py:253-258).
//
// Forward statistics arrive as per-32-row-tile (mean, M2) pairs from the producing GEMM's epilogue
// and are merged here with Chan's formula in fp64 in a fixed tree order (deterministic).  Backward
// statistics (sum dA, sum dA * xhat) are reduced per 64-row tile by `bn_act_bwd_stats` and summed in
// fp64.  All elementwise passes are HBM-bound streaming kernels over [M, N] fp32 row-major data.
#include "common.h"

#define BN_THREADS 256

struct ActSpec {
    swr_act_range r[SWR_MAX_ACT_RANGES];
    int n;
};

static int make_acts(const swr_act_range* acts, int n_acts, int N, ActSpec& out) {
    SWR_REQUIRE(n_acts >= 0 && n_acts <= SWR_MAX_ACT_RANGES && (n_acts == 0 || acts), SWR_ERR_ARG);
    out.n = n_acts;
    for (int i = 0; i < n_acts; ++i) {
        out.r[i] = acts[i];
        SWR_REQUIRE(acts[i].col_lo >= 0 && acts[i].col_hi <= N && acts[i].col_lo <= acts[i].col_hi, SWR_ERR_ARG);
        SWR_REQUIRE(acts[i].act >= SWR_ACT_NONE && acts[i].act <= SWR_ACT_SOFTMAX, SWR_ERR_ARG);
        if (acts[i].act == SWR_ACT_SOFTMAX)
            SWR_REQUIRE(acts[i].group > 0 && (acts[i].col_hi - acts[i].col_lo) % acts[i].group == 0, SWR_ERR_ARG);
    }
    return SWR_OK;
}

__device__ __forceinline__ int find_act(const ActSpec& a, int n, int& lo, int& group) {
    for (int i = 0; i < a.n; ++i)
        if (n >= a.r[i].col_lo && n < a.r[i].col_hi) {
            lo = a.r[i].col_lo;
            group = a.r[i].group;
            return a.r[i].act;
        }
    lo = 0;
    group = 1;
    return SWR_ACT_NONE;
}

// ---------------------------------------------------------------------------------- forward stats
__global__ __launch_bounds__(BN_THREADS) void bn_finalize_kernel(
    const float* __restrict__ part, int n_tiles, int64_t M, int N, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float momentum, float* running_mean, float* running_var,
    int64_t* nbt, int n_tracked, float* mean_o, float* rstd_o, float* scale_o, float* shift_o) {
    __shared__ SwrMoments sm[BN_THREADS];
    const int n = blockIdx.x;
    SwrMoments acc = {0.0, 0.0, 0.0};
    // each thread merges a contiguous run of tiles in order, then a fixed binary tree over threads
    const int per = (n_tiles + BN_THREADS - 1) / BN_THREADS;
    const int t0 = threadIdx.x * per;
    for (int t = t0; t < min(t0 + per, n_tiles); ++t) {
        const float* p = part + (static_cast<int64_t>(t) * N + n) * 2;
        SwrMoments b;
        b.n = static_cast<double>(min<int64_t>(32, M - static_cast<int64_t>(t) * 32));
        b.mean = p[0];
        b.m2 = p[1];
        acc = swr_merge(acc, b);
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 1; st < BN_THREADS; st <<= 1) {
        if ((threadIdx.x & (2 * st - 1)) == 0) sm[threadIdx.x] = swr_merge(sm[threadIdx.x], sm[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const SwrMoments tot = sm[0];
        const double var_b = tot.m2 / tot.n;
        const float mean = static_cast<float>(tot.mean);
        const float rstd = static_cast<float>(1.0 / sqrt(var_b + static_cast<double>(eps)));
        const float g = gamma ? gamma[n] : 1.f, b = beta ? beta[n] : 0.f;
        const float scale = g * rstd;
        if (mean_o) mean_o[n] = mean;
        if (rstd_o) rstd_o[n] = rstd;
        if (scale_o) scale_o[n] = scale;
        if (shift_o) shift_o[n] = b - mean * scale;
        if (running_mean) running_mean[n] = (1.f - momentum) * running_mean[n] + momentum * mean;
        if (running_var) {
            const float var_u = static_cast<float>(tot.n > 1.0 ? tot.m2 / (tot.n - 1.0) : var_b);
            running_var[n] = (1.f - momentum) * running_var[n] + momentum * var_u;
        }
        if (nbt && n < n_tracked) nbt[n] += 1;
    }
}

extern "C" int swr_bn_finalize(const float* stat_partials, int n_tiles, int64_t M, int N, const float* gamma,
                               const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                               int64_t* num_batches_tracked, int n_tracked, float* mean, float* rstd, float* scale,
                               float* shift, void* stream) {
    SWR_REQUIRE(stat_partials && n_tiles > 0 && M > 0 && N > 0, SWR_ERR_ARG);
    SWR_REQUIRE(n_tiles == swr_ceil_div(M, 32) && n_tracked >= 0 && n_tracked <= N, SWR_ERR_ARG);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(N), dim3(BN_THREADS), 0, static_cast<hipStream_t>(stream), stat_partials,
                       n_tiles, M, N, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, n_tracked,
                       mean, rstd, scale, shift);
    return swr_launch_status();
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      int N, float* scale, float* shift) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float sc = (gamma ? gamma[n] : 1.f) / sqrtf(rv[n] + eps);
    scale[n] = sc;
    shift[n] = (beta ? beta[n] : 0.f) - rm[n] * sc;
}

extern "C" int swr_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, int N, float* scale, float* shift, void* stream) {
    SWR_REQUIRE(running_mean && running_var && scale && shift && N > 0, SWR_ERR_ARG);
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((N + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), gamma,
                       beta, running_mean, running_var, eps, N, scale, shift);
    return swr_launch_status();
}

// ------------------------------------------------------------------------------ affine + act forward
__global__ __launch_bounds__(BN_THREADS) void affine_act_fwd_kernel(const float* __restrict__ Z, int64_t ldz,
                                                                    const float* __restrict__ scale,
                                                                    const float* __restrict__ shift, const ActSpec acts,
                                                                    float* __restrict__ Y, int64_t ldy, int64_t M, int N) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * BN_THREADS + threadIdx.x;
    const int64_t m = idx / N;
    const int n = static_cast<int>(idx - m * N);
    if (m >= M) return;
    int lo, group;
    const int act = find_act(acts, n, lo, group);
    const float* z = Z + m * ldz;
    const float v = (scale ? scale[n] : 1.f) * z[n] + (shift ? shift[n] : 0.f);
    float y;
    if (act == SWR_ACT_RELU) {
        y = fmaxf(v, 0.f);
    } else if (act == SWR_ACT_SIGMOID) {
        y = swr_sigmoid(v);
    } else if (act == SWR_ACT_SOFTMAX) {
        const int g0 = lo + ((n - lo) / group) * group;
        float mx = -INFINITY;
        for (int j = 0; j < group; ++j)
            mx = fmaxf(mx, (scale ? scale[g0 + j] : 1.f) * z[g0 + j] + (shift ? shift[g0 + j] : 0.f));
        float den = 0.f;
        for (int j = 0; j < group; ++j)
            den += expf((scale ? scale[g0 + j] : 1.f) * z[g0 + j] + (shift ? shift[g0 + j] : 0.f) - mx);
        y = expf(v - mx) / den;
    } else {
        y = v;
    }
    Y[m * ldy + n] = y;
}

extern "C" int swr_affine_act_fwd(const float* Z, int64_t ldz, const float* scale, const float* shift,
                                  const swr_act_range* acts, int n_acts, float* Y, int64_t ldy, int64_t M, int N,
                                  void* stream) {
    SWR_REQUIRE(Z && Y && M >= 0 && N > 0 && ldz >= N && ldy >= N, SWR_ERR_ARG);
    ActSpec as;
    const int rc = make_acts(acts, n_acts, N, as);
    if (rc != SWR_OK) return rc;
    if (M == 0) return SWR_OK;
    hipLaunchKernelGGL(affine_act_fwd_kernel, dim3(static_cast<unsigned>(swr_ceil_div(M * N, BN_THREADS))), dim3(BN_THREADS),
                       0, static_cast<hipStream_t>(stream), Z, ldz, scale, shift, as, Y, ldy, M, N);
    return swr_launch_status();
}

// --------------------------------------------------------------------------------------- backward
// dA = act'(Y) * dY for one element (softmax needs the whole group of the row)
__device__ __forceinline__ float act_grad(const ActSpec& acts, const float* __restrict__ dy, const float* __restrict__ y, int n) {
    int lo, group;
    const int act = find_act(acts, n, lo, group);
    const float g = dy[n];
    if (act == SWR_ACT_RELU) return y[n] > 0.f ? g : 0.f;
    if (act == SWR_ACT_SIGMOID) return g * y[n] * (1.f - y[n]);
    if (act == SWR_ACT_SOFTMAX) {
        const int g0 = lo + ((n - lo) / group) * group;
        float dot = 0.f;
        for (int j = 0; j < group; ++j) dot = fmaf(dy[g0 + j], y[g0 + j], dot);
        return y[n] * (g - dot);
    }
    return g;
}

#define BWD_TILE 64
// block = 64 columns x 4 row phases over a 64-row tile; partials[tile][n] = (sum dA, sum dA * xhat)
__global__ __launch_bounds__(BN_THREADS) void bn_act_bwd_stats_kernel(
    const float* __restrict__ dY, int64_t lddy, const float* __restrict__ Y, int64_t ldy, const float* __restrict__ Z,
    int64_t ldz, const float* __restrict__ mean, const float* __restrict__ rstd, const ActSpec acts,
    float* __restrict__ partials, int64_t M, int N) {
    __shared__ float s1[4][64], s2[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int n = blockIdx.y * 64 + cx;
    const int64_t m0 = static_cast<int64_t>(blockIdx.x) * BWD_TILE;
    float a1 = 0.f, a2 = 0.f;
    if (n < N) {
        const float mu = mean[n], rs = rstd[n];
        for (int r = ry; r < BWD_TILE; r += 4) {
            const int64_t m = m0 + r;
            if (m >= M) break;
            const float da = act_grad(acts, dY + m * lddy, Y + m * ldy, n);
            a1 += da;
            a2 = fmaf(da, (Z[m * ldz + n] - mu) * rs, a2);
        }
    }
    s1[ry][cx] = a1;
    s2[ry][cx] = a2;
    __syncthreads();
    if (ry == 0 && n < N) {
        float* p = partials + (static_cast<int64_t>(blockIdx.x) * N + n) * 2;
        p[0] = (s1[0][cx] + s1[1][cx]) + (s1[2][cx] + s1[3][cx]);
        p[1] = (s2[0][cx] + s2[1][cx]) + (s2[2][cx] + s2[3][cx]);
    }
}

extern "C" int swr_bn_act_bwd_stats(const float* dY, int64_t lddy, const float* Y, int64_t ldy, const float* Z,
                                    int64_t ldz, const float* mean, const float* rstd, const swr_act_range* acts,
                                    int n_acts, float* partials, int64_t M, int N, void* stream) {
    SWR_REQUIRE(dY && Y && Z && mean && rstd && partials && M > 0 && N > 0, SWR_ERR_ARG);
    ActSpec as;
    const int rc = make_acts(acts, n_acts, N, as);
    if (rc != SWR_OK) return rc;
    const dim3 grid(static_cast<unsigned>(swr_ceil_div(M, BWD_TILE)), static_cast<unsigned>(swr_ceil_div(N, 64)));
    hipLaunchKernelGGL(bn_act_bwd_stats_kernel, grid, dim3(BN_THREADS), 0, static_cast<hipStream_t>(stream), dY, lddy, Y,
                       ldy, Z, ldz, mean, rstd, as, partials, M, N);
    return swr_launch_status();
}

__global__ __launch_bounds__(BN_THREADS) void bn_bwd_finalize_kernel(const float* __restrict__ part, int n_tiles, int64_t M,
                                                                     int N, const float* __restrict__ gamma,
                                                                     const float* __restrict__ rstd, float* dgamma,
                                                                     float* dbeta, int accumulate, float* ca, float* cb,
                                                                     float* cc) {
    __shared__ double t1[BN_THREADS], t2[BN_THREADS];
    const int n = blockIdx.x;
    double a1 = 0.0, a2 = 0.0;
    const int per = (n_tiles + BN_THREADS - 1) / BN_THREADS;
    const int t0 = threadIdx.x * per;
    for (int t = t0; t < min(t0 + per, n_tiles); ++t) {
        const float* p = part + (static_cast<int64_t>(t) * N + n) * 2;
        a1 += p[0];
        a2 += p[1];
    }
    t1[threadIdx.x] = a1;
    t2[threadIdx.x] = a2;
    __syncthreads();
    for (int st = 1; st < BN_THREADS; st <<= 1) {
        if ((threadIdx.x & (2 * st - 1)) == 0) {
            t1[threadIdx.x] += t1[threadIdx.x + st];
            t2[threadIdx.x] += t2[threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double S1 = t1[0], S2 = t2[0];
        const double g = gamma ? gamma[n] : 1.0, rs = rstd[n];
        if (dgamma) dgamma[n] = (accumulate ? dgamma[n] : 0.f) + static_cast<float>(S2);
        if (dbeta) dbeta[n] = (accumulate ? dbeta[n] : 0.f) + static_cast<float>(S1);
        // dZ = g rs dA - g rs^2 (S2 / M) (Z - mean) - g rs (S1 / M)
        ca[n] = static_cast<float>(g * rs);
        cb[n] = static_cast<float>(-g * rs * rs * S2 / static_cast<double>(M));
        cc[n] = static_cast<float>(-g * rs * S1 / static_cast<double>(M));
    }
}

extern "C" int swr_bn_bwd_finalize(const float* partials, int n_tiles, int64_t M, int N, const float* gamma,
                                   const float* rstd, float* dgamma, float* dbeta, int accumulate, float* ca, float* cb,
                                   float* cc, void* stream) {
    SWR_REQUIRE(partials && rstd && ca && cb && cc && n_tiles > 0 && M > 0 && N > 0, SWR_ERR_ARG);
    SWR_REQUIRE(n_tiles == swr_ceil_div(M, BWD_TILE), SWR_ERR_ARG);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(N), dim3(BN_THREADS), 0, static_cast<hipStream_t>(stream), partials,
                       n_tiles, M, N, gamma, rstd, dgamma, dbeta, accumulate, ca, cb, cc);
    return swr_launch_status();
}

__global__ __launch_bounds__(BN_THREADS) void act_bwd_apply_kernel(
    const float* __restrict__ dY, int64_t lddy, const float* __restrict__ Y, int64_t ldy, const float* __restrict__ Z,
    int64_t ldz, const float* __restrict__ ca, const float* __restrict__ cb, const float* __restrict__ cc,
    const float* __restrict__ mean, const ActSpec acts, float* __restrict__ dZ, int64_t lddz, int64_t M, int N) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * BN_THREADS + threadIdx.x;
    const int64_t m = idx / N;
    const int n = static_cast<int>(idx - m * N);
    if (m >= M) return;
    float v = act_grad(acts, dY + m * lddy, Y + m * ldy, n);
    if (ca) v *= ca[n];
    if (cb) v = fmaf(cb[n], Z[m * ldz + n] - mean[n], v) + cc[n];
    dZ[m * lddz + n] = v;
}

extern "C" int swr_act_bwd_apply(const float* dY, int64_t lddy, const float* Y, int64_t ldy, const float* Z, int64_t ldz,
                                 const float* ca, const float* cb, const float* cc, const float* mean,
                                 const swr_act_range* acts, int n_acts, float* dZ, int64_t lddz, int64_t M, int N,
                                 void* stream) {
    SWR_REQUIRE(dY && Y && dZ && M >= 0 && N > 0, SWR_ERR_ARG);
    SWR_REQUIRE(cb == nullptr || (Z && cc && mean), SWR_ERR_ARG);
    ActSpec as;
    const int rc = make_acts(acts, n_acts, N, as);
    if (rc != SWR_OK) return rc;
    if (M == 0) return SWR_OK;
    hipLaunchKernelGGL(act_bwd_apply_kernel, dim3(static_cast<unsigned>(swr_ceil_div(M * N, BN_THREADS))), dim3(BN_THREADS),
                       0, static_cast<hipStream_t>(stream), dY, lddy, Y, ldy, Z, ldz, ca, cb, cc, mean, as, dZ, lddz, M, N);
    return swr_launch_status();
}
