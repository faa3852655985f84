// STAR's factorised weights (reference models/multi_domain/star.py:99-107): layer l of domain d
// multiplies by W_s,l (.) W_d,l and adds b_s,l + b_d,l, and the first layer also absorbs the domain affine of the
// partitioned normalisation (gamma_s gamma_d, beta_s + beta_d).  The reference rebuilds these small tensors with a
// dozen elementwise launches per (layer, domain) and autograd replays three times as many backwards; here one launch
// per layer each way produces, for all domains at once, the effective weights in Linear layout [out, in] (what the
// product kernels stage) and, backwards, every parameter gradient.  Latency-bound: the tensors are <= 376 x 256.
//   forward : w0 = Ws (.) Wd[d];  first layer: W_eff[o][i] = a_i w0[i][o], b_eff[o] = bs[o] + bd[d][o] + sum_i c_i w0[i][o]
//             with a = gamma_s gamma_d[d], c = beta_s + beta_d[d];  other layers: W_eff[o][i] = w0[i][o], b_eff = bs + bd[d]
//   backward: dw0[i][o] = G[o][i] a_i + gb[o] c_i  (a = 1, c = 0 after the first layer);  dWd[d] = dw0 (.) Ws,
//             dWs = sum_d dw0 (.) Wd[d];  dbd[d] = gb,  dbs = sum_d gb;  first layer: da_i = sum_o G[o][i] w0[i][o],
//             dc_i = sum_o gb[o] w0[i][o];  dgamma_d[d] = da gamma_s, dgamma_s = sum_d da gamma_d[d],  dbeta_d[d] = dc,
//             dbeta_s = sum_d dc.   All sums over d and o in index order (deterministic).
#include "common.h"

#define STAR_THREADS 256

struct StarK {
    swr_star_layer_args a;
};

__device__ __forceinline__ void star_store(float* p, float v, int accumulate) {
    if (p) *p = accumulate ? *p + v : v;
}

// workgroups [0, w_blocks): effective weights, thread = (i, o) with i fastest (coalesced writes of W_eff[o][i]);
// the rest: effective biases
__device__ __forceinline__ void star_layer_fwd_body(const swr_star_layer_args& a, int block, int w_blocks) {
    const int I = a.in_dim, O = a.out_dim;
    if (block < w_blocks) {
        const int64_t e = static_cast<int64_t>(block) * STAR_THREADS + threadIdx.x;
        if (e >= static_cast<int64_t>(I) * O) return;
        const int o = static_cast<int>(e / I), i = static_cast<int>(e - static_cast<int64_t>(o) * I);
        const float ws = a.Ws[static_cast<int64_t>(i) * O + o];
        for (int d = 0; d < a.D; ++d) {
            float w = ws * a.Wd[d][static_cast<int64_t>(i) * O + o];
            if (a.first) w *= a.gamma_s[i] * a.gamma_d[d][i];
            a.W_eff[d][e] = w;
        }
        return;
    }
    // biases: workgroup = 16 outputs x 16 slices of the input index (the first layer's sum over i is 376 terms long:
    // one thread per output would be one serial chain); slices are added in order
    __shared__ float red[16][17];
    const int ol = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int o = (block - w_blocks) * 16 + ol;
    for (int d = 0; d < a.D; ++d) {
        float s = 0.f;
        if (a.first && o < O)
            for (int i = sl; i < I; i += 16)
                s = fmaf(a.beta_s[i] + a.beta_d[d][i], a.Ws[static_cast<int64_t>(i) * O + o] * a.Wd[d][static_cast<int64_t>(i) * O + o], s);
        red[sl][ol] = s;
        __syncthreads();
        if (sl == 0 && o < O) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += red[q][ol];
            a.b_eff[d][o] = a.bs[o] + a.bd[d][o] + t;
        }
        __syncthreads();
    }
}

// workgroups [0, w_blocks): weight gradients, thread = (i, o) with o fastest (coalesced reads / writes of the [in, out]
// parameters); then b_blocks of bias gradients (thread = o); then (first layer) the affine's gradients (thread = i)
__global__ __launch_bounds__(STAR_THREADS) void star_layer_fwd_kernel(const StarK k, int w_blocks) {
    star_layer_fwd_body(k.a, static_cast<int>(blockIdx.x), w_blocks);
}

__device__ __forceinline__ void star_layer_bwd_body(const swr_star_layer_args& a, int block, int w_blocks, int b_blocks) {
    const int I = a.in_dim, O = a.out_dim;
    const int acc = a.accumulate;
    if (block < w_blocks) {
        const int64_t e = static_cast<int64_t>(block) * STAR_THREADS + threadIdx.x;
        if (e >= static_cast<int64_t>(I) * O) return;
        const int i = static_cast<int>(e / O), o = static_cast<int>(e - static_cast<int64_t>(i) * O);
        const float ws = a.Ws[e];
        float sum_s = 0.f;
        for (int d = 0; d < a.D; ++d) {
            const float G = a.dW_eff[d] ? a.dW_eff[d][static_cast<int64_t>(o) * I + i] : 0.f;
            float dw0 = G;
            if (a.first) {
                const float gb = a.db_eff[d] ? a.db_eff[d][o] : 0.f;
                dw0 = fmaf(G, a.gamma_s[i] * a.gamma_d[d][i], gb * (a.beta_s[i] + a.beta_d[d][i]));
            }
            star_store(a.dWd[d] ? a.dWd[d] + e : nullptr, dw0 * ws, acc);
            sum_s = fmaf(dw0, a.Wd[d][e], sum_s);
        }
        star_store(a.dWs ? a.dWs + e : nullptr, sum_s, acc);
        return;
    }
    const int bb = block - w_blocks;
    if (bb < b_blocks) {
        const int o = bb * STAR_THREADS + threadIdx.x;
        if (o >= O) return;
        float s = 0.f;
        for (int d = 0; d < a.D; ++d) {
            const float gb = a.db_eff[d] ? a.db_eff[d][o] : 0.f;
            star_store(a.dbd[d] ? a.dbd[d] + o : nullptr, gb, acc);
            s += gb;
        }
        star_store(a.dbs ? a.dbs + o : nullptr, s, acc);
        return;
    }
    // first layer only: gradients of the partitioned norm's affine; workgroup = 16 inputs x 16 slices of the output index
    __shared__ float ra[16][17], rc[16][17];
    const int il = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int i = (bb - b_blocks) * 16 + il;
    float dgs = 0.f, dbs_ = 0.f;
    for (int d = 0; d < a.D; ++d) {
        float da = 0.f, dc = 0.f;
        if (i < I)
            for (int o = sl; o < O; o += 16) {
                const float w0 = a.Ws[static_cast<int64_t>(i) * O + o] * a.Wd[d][static_cast<int64_t>(i) * O + o];
                const float G = a.dW_eff[d] ? a.dW_eff[d][static_cast<int64_t>(o) * I + i] : 0.f;
                const float gb = a.db_eff[d] ? a.db_eff[d][o] : 0.f;
                da = fmaf(G, w0, da);
                dc = fmaf(gb, w0, dc);
            }
        ra[sl][il] = da;
        rc[sl][il] = dc;
        __syncthreads();
        if (sl == 0 && i < I) {
            float ta = 0.f, tc = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) { ta += ra[q][il]; tc += rc[q][il]; }
            star_store(a.dgamma_d[d] ? a.dgamma_d[d] + i : nullptr, ta * a.gamma_s[i], acc);
            star_store(a.dbeta_d[d] ? a.dbeta_d[d] + i : nullptr, tc, acc);
            dgs = fmaf(ta, a.gamma_d[d][i], dgs);
            dbs_ += tc;
        }
        __syncthreads();
    }
    if (sl == 0 && i < I) {
        star_store(a.dgamma_s ? a.dgamma_s + i : nullptr, dgs, acc);
        star_store(a.dbeta_s ? a.dbeta_s + i : nullptr, dbs_, acc);
    }
}

__global__ __launch_bounds__(STAR_THREADS) void star_layer_bwd_kernel(const StarK k, int w_blocks, int b_blocks) {
    star_layer_bwd_body(k.a, static_cast<int>(blockIdx.x), w_blocks, b_blocks);
}

// several layers per launch (the tensors are parameter-sized: one launch per layer each way is 14 latency-bound launches
// per step at config 3): the layers' workgroups back to back, descriptors by value (4 x 856 bytes of the 4 KB kernarg segment)
#define STAR_MULTI 4
struct StarMultiK {
    swr_star_layer_args a[STAR_MULTI];
    int first[STAR_MULTI + 1];          // first workgroup of each layer
    int w_blocks[STAR_MULTI], b_blocks[STAR_MULTI];
    int n;
};
__device__ __forceinline__ int star_multi_layer(const StarMultiK& k, int block) {
    int l = 0;
    while (l + 1 < k.n && block >= k.first[l + 1]) ++l;
    return l;
}
__global__ __launch_bounds__(STAR_THREADS) void star_layers_fwd_kernel(const StarMultiK k) {
    const int l = star_multi_layer(k, static_cast<int>(blockIdx.x));
    star_layer_fwd_body(k.a[l], static_cast<int>(blockIdx.x) - k.first[l], k.w_blocks[l]);
}
__global__ __launch_bounds__(STAR_THREADS) void star_layers_bwd_kernel(const StarMultiK k) {
    const int l = star_multi_layer(k, static_cast<int>(blockIdx.x));
    star_layer_bwd_body(k.a[l], static_cast<int>(blockIdx.x) - k.first[l], k.w_blocks[l], k.b_blocks[l]);
}

static int star_check(const swr_star_layer_args* args, bool bwd) {
    SWR_REQUIRE(args != nullptr, SWR_ERR_ARG);
    const swr_star_layer_args& a = *args;
    SWR_REQUIRE(a.D >= 1 && a.D <= SWR_STAR_MAX_DOMAINS && a.in_dim > 0 && a.out_dim > 0 && a.Ws && a.bs, SWR_ERR_ARG);
    for (int d = 0; d < a.D; ++d) {
        SWR_REQUIRE(a.Wd[d] && a.bd[d], SWR_ERR_ARG);
        if (!bwd) SWR_REQUIRE(a.W_eff[d] && a.b_eff[d], SWR_ERR_ARG);
        if (a.first) SWR_REQUIRE(a.gamma_d[d] && a.beta_d[d], SWR_ERR_ARG);
    }
    if (a.first) SWR_REQUIRE(a.gamma_s && a.beta_s, SWR_ERR_ARG);
    return SWR_OK;
}

extern "C" int swr_star_layer_fwd(const swr_star_layer_args* args, void* stream) {
    const int rc = star_check(args, false);
    if (rc != SWR_OK) return rc;
    StarK k;
    k.a = *args;
    const int w_blocks = static_cast<int>(swr_ceil_div(static_cast<int64_t>(k.a.in_dim) * k.a.out_dim, STAR_THREADS));
    const int b_blocks = static_cast<int>(swr_ceil_div(k.a.out_dim, 16));
    hipLaunchKernelGGL(star_layer_fwd_kernel, dim3(w_blocks + b_blocks), dim3(STAR_THREADS), 0, static_cast<hipStream_t>(stream), k,
                       w_blocks);
    return swr_launch_status();
}

extern "C" int swr_star_layer_bwd(const swr_star_layer_args* args, void* stream) {
    const int rc = star_check(args, true);
    if (rc != SWR_OK) return rc;
    StarK k;
    k.a = *args;
    const int w_blocks = static_cast<int>(swr_ceil_div(static_cast<int64_t>(k.a.in_dim) * k.a.out_dim, STAR_THREADS));
    const int b_blocks = static_cast<int>(swr_ceil_div(k.a.out_dim, STAR_THREADS));
    const int a_blocks = k.a.first ? static_cast<int>(swr_ceil_div(k.a.in_dim, 16)) : 0;
    hipLaunchKernelGGL(star_layer_bwd_kernel, dim3(w_blocks + b_blocks + a_blocks), dim3(STAR_THREADS), 0,
                       static_cast<hipStream_t>(stream), k, w_blocks, b_blocks);
    return swr_launch_status();
}

static int star_layers(const swr_star_layer_args* layers, int n_layers, bool bwd, void* stream) {
    SWR_REQUIRE(layers != nullptr && n_layers > 0, SWR_ERR_ARG);
    for (int l0 = 0; l0 < n_layers; l0 += STAR_MULTI) {
        StarMultiK k;
        k.n = n_layers - l0 < STAR_MULTI ? n_layers - l0 : STAR_MULTI;
        int pos = 0;
        for (int l = 0; l < k.n; ++l) {
            const int rc = star_check(layers + l0 + l, bwd);
            if (rc != SWR_OK) return rc;
            k.a[l] = layers[l0 + l];
            const swr_star_layer_args& a = k.a[l];
            k.w_blocks[l] = static_cast<int>(swr_ceil_div(static_cast<int64_t>(a.in_dim) * a.out_dim, STAR_THREADS));
            k.b_blocks[l] = bwd ? static_cast<int>(swr_ceil_div(a.out_dim, STAR_THREADS)) : static_cast<int>(swr_ceil_div(a.out_dim, 16));
            const int a_blocks = (bwd && a.first) ? static_cast<int>(swr_ceil_div(a.in_dim, 16)) : 0;
            k.first[l] = pos;
            pos += k.w_blocks[l] + k.b_blocks[l] + a_blocks;
        }
        k.first[k.n] = pos;
        if (bwd)
            hipLaunchKernelGGL(star_layers_bwd_kernel, dim3(pos), dim3(STAR_THREADS), 0, static_cast<hipStream_t>(stream), k);
        else
            hipLaunchKernelGGL(star_layers_fwd_kernel, dim3(pos), dim3(STAR_THREADS), 0, static_cast<hipStream_t>(stream), k);
    }
    return swr_launch_status();
}

extern "C" int swr_star_layers_fwd(const swr_star_layer_args* layers, int n_layers, void* stream) {
    return star_layers(layers, n_layers, false, stream);
}

extern "C" int swr_star_layers_bwd(const swr_star_layer_args* layers, int n_layers, void* stream) {
    return star_layers(layers, n_layers, true, stream);
}
