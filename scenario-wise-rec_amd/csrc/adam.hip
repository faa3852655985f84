// torch.optim.Adam (L2 weight decay added to the gradient) of the reference trainer
// (trainers/ctr_trainer.py:50-52,73) as HBM-streaming kernels: 4 reads + 3 writes of 4 bytes per
// parameter.  Step count / bias corrections live in device memory so a captured hipGraph replays.
#include "common.h"

#define AD_THREADS 256

__global__ void adam_advance_kernel(swr_adam_hyper* h) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    h->step += 1;
    const double t = static_cast<double>(h->step);
    const double bc1 = 1.0 - pow(h->beta1, t), bc2 = 1.0 - pow(h->beta2, t);
    h->step_size = static_cast<float>(h->lr / bc1);
    h->inv_bc2_sqrt = static_cast<float>(1.0 / sqrt(bc2));
    h->one_minus_b1 = static_cast<float>(1.0 - h->beta1);
    h->b2 = static_cast<float>(h->beta2);
    h->one_minus_b2 = static_cast<float>(1.0 - h->beta2);
    h->eps_f = static_cast<float>(h->eps);
    h->wd_f = static_cast<float>(h->weight_decay);
}

extern "C" int swr_adam_advance(swr_adam_hyper* hyper, void* stream) {
    SWR_REQUIRE(hyper != nullptr, SWR_ERR_ARG);
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), hyper);
    return swr_launch_status();
}

// one element of torch's _single_tensor_adam: grad += wd * p; m.lerp_(grad, 1 - b1);
// v = v * b2 + (1 - b2) grad^2; p -= step_size * m / (sqrt(v) / sqrt(bc2) + eps)
__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const swr_adam_hyper& h) {
    g = fmaf(h.wd_f, p, g);
    m = fmaf(g - m, h.one_minus_b1, m);
    v = fmaf(v, h.b2, h.one_minus_b2 * g * g);
    const float denom = sqrtf(v) * h.inv_bc2_sqrt + h.eps_f;
    p -= h.step_size * (m / denom);
}

__global__ __launch_bounds__(AD_THREADS) void adam_dense_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                                float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                                const swr_adam_hyper* __restrict__ hp) {
    const swr_adam_hyper h = *hp;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * AD_THREADS;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * AD_THREADS + threadIdx.x; i < n; i += stride) {
        float pi = p[i], mi = m[i], vi = v[i];
        adam_elem(pi, g[i], mi, vi, h);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

extern "C" int swr_adam_dense(float* p, const float* g, float* m, float* v, int64_t n, const swr_adam_hyper* hyper,
                              void* stream) {
    SWR_REQUIRE(p && g && m && v && hyper && n >= 0, SWR_ERR_ARG);
    if (n == 0) return SWR_OK;
    const unsigned grid = static_cast<unsigned>(swr_ceil_div(n, AD_THREADS) < 4096 ? swr_ceil_div(n, AD_THREADS) : 4096);
    hipLaunchKernelGGL(adam_dense_kernel, dim3(grid), dim3(AD_THREADS), 0, static_cast<hipStream_t>(stream), p, g, m, v, n, hyper);
    return swr_launch_status();
}

// touched rows of a large table: thread per (entry, column)
__global__ __launch_bounds__(AD_THREADS) void adam_rows_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                               int64_t vocab, int dim, const int32_t* __restrict__ urow,
                                                               const float* __restrict__ ugrad, int64_t n_entries,
                                                               uint32_t* __restrict__ bitmap,
                                                               const swr_adam_hyper* __restrict__ hp) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * AD_THREADS + threadIdx.x;
    const int64_t i = idx / dim;
    if (i >= n_entries) return;
    const int e = static_cast<int>(idx - i * dim);
    const int32_t row = urow[i];
    if (row < 0 || row >= vocab) return;
    const swr_adam_hyper h = *hp;
    const int64_t o = static_cast<int64_t>(row) * dim + e;
    float pi = p[o], mi = m[o], vi = v[o];
    adam_elem(pi, ugrad[i * dim + e], mi, vi, h);
    p[o] = pi; m[o] = mi; v[o] = vi;
    if (e == 0) atomicOr(bitmap + (row >> 5), 1u << (row & 31));
}

extern "C" int swr_adam_rows(float* p, float* m, float* v, int64_t vocab, int dim, const int32_t* urow, const float* ugrad,
                             int64_t n_entries, uint32_t* bitmap, const swr_adam_hyper* hyper, void* stream) {
    SWR_REQUIRE(p && m && v && urow && ugrad && bitmap && hyper && vocab > 0 && dim > 0 && n_entries >= 0, SWR_ERR_ARG);
    if (n_entries == 0) return SWR_OK;
    hipLaunchKernelGGL(adam_rows_kernel, dim3(static_cast<unsigned>(swr_ceil_div(n_entries * dim, AD_THREADS))),
                       dim3(AD_THREADS), 0, static_cast<hipStream_t>(stream), p, m, v, vocab, dim, urow, ugrad, n_entries,
                       bitmap, hyper);
    return swr_launch_status();
}

// every row NOT marked in the bitmap takes g = wd * p (zero data gradient); one wave-slice of 32 rows per
// bitmap word, the word is cleared after use so the bitmap leaves as it came (all zero)
__global__ __launch_bounds__(AD_THREADS) void adam_sweep_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                                int64_t vocab, int dim, uint32_t* __restrict__ bitmap,
                                                                const swr_adam_hyper* __restrict__ hp) {
    const swr_adam_hyper h = *hp;
    const int64_t n = vocab * dim;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * AD_THREADS;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * AD_THREADS + threadIdx.x; i < n; i += stride) {
        const int64_t row = i / dim;
        if ((bitmap[row >> 5] >> (row & 31)) & 1u) continue;
        float pi = p[i], mi = m[i], vi = v[i];
        adam_elem(pi, 0.f, mi, vi, h);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

extern "C" int swr_adam_sweep_untouched(float* p, float* m, float* v, int64_t vocab, int dim, uint32_t* bitmap,
                                        int clear_bitmap, const swr_adam_hyper* hyper, void* stream) {
    SWR_REQUIRE(p && m && v && bitmap && hyper && vocab > 0 && dim > 0, SWR_ERR_ARG);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t n = vocab * dim;
    const unsigned grid = static_cast<unsigned>(swr_ceil_div(n, AD_THREADS) < 8192 ? swr_ceil_div(n, AD_THREADS) : 8192);
    hipLaunchKernelGGL(adam_sweep_kernel, dim3(grid), dim3(AD_THREADS), 0, st, p, m, v, vocab, dim, bitmap, hyper);
    if (!clear_bitmap) return swr_launch_status();
    return swr_zero_async(bitmap, static_cast<size_t>(swr_ceil_div(vocab, 32)) * 4, st);
}
