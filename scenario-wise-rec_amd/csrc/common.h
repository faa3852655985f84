// Shared helpers for the libswr HIP sources (gfx950 / CDNA4 only).
#pragma once
// -DSWR_X3 (variant builds only, tools/build_variant.py): every fp32 product as THREE bf16 products (h h, h m, m h) instead of six --
// the terms of order 2^-16 (l h, h l, m m) are dropped.  A measurement build: the product library never defines it.
#ifdef SWR_X3
#define SWR_X3_ON 1
#else
#define SWR_X3_ON 0
#endif
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "swr.h"

#define SWR_WAVE 64

#define SWR_REQUIRE(cond, code) \
    do {                        \
        if (!(cond)) return (code); \
    } while (0)

static inline int swr_launch_status() {
    return hipGetLastError() == hipSuccess ? SWR_OK : SWR_ERR_LAUNCH;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize), once per (kernel, device): the attribute belongs to the device the call is made on,
// so a process-wide "done" flag leaves the second GPU of a multi-GPU process without it (its launch then fails).  Not a stream
// operation: legal during hipGraph capture.
#include <mutex>
#include <set>
#include <utility>
static inline bool swr_raise_lds(const void* fn, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({fn, dev})) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    done.insert({fn, dev});
    return true;
}

static inline bool swr_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static inline int64_t swr_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- typed loads of caller columns: `.long()` / `.float()` of the reference (layers.py:70,89)
__device__ __forceinline__ int64_t swr_load_index(const void* p, int dtype, int64_t i) {
    switch (dtype) {
        case SWR_I8: return static_cast<const int8_t*>(p)[i];
        case SWR_U8: return static_cast<const uint8_t*>(p)[i];
        case SWR_BOOL: return static_cast<const uint8_t*>(p)[i] != 0;
        case SWR_I16: return static_cast<const int16_t*>(p)[i];
        case SWR_I32: return static_cast<const int32_t*>(p)[i];
        default: return static_cast<const int64_t*>(p)[i];
    }
}

__device__ __forceinline__ float swr_bf16_to_float(uint16_t h) { return __uint_as_float(static_cast<uint32_t>(h) << 16); }

__device__ __forceinline__ float swr_load_value(const void* p, int dtype, int64_t i) {
    switch (dtype) {
        case SWR_F32: return static_cast<const float*>(p)[i];
        case SWR_F16: return __half2float(static_cast<const __half*>(p)[i]);
        case SWR_BF16: return swr_bf16_to_float(static_cast<const uint16_t*>(p)[i]);
        case SWR_F64: return static_cast<float>(static_cast<const double*>(p)[i]);
        case SWR_I8: return static_cast<float>(static_cast<const int8_t*>(p)[i]);
        case SWR_U8: return static_cast<float>(static_cast<const uint8_t*>(p)[i]);
        case SWR_BOOL: return static_cast<const uint8_t*>(p)[i] != 0 ? 1.f : 0.f;
        case SWR_I16: return static_cast<float>(static_cast<const int16_t*>(p)[i]);
        case SWR_I32: return static_cast<float>(static_cast<const int32_t*>(p)[i]);
        default: return static_cast<float>(static_cast<const int64_t*>(p)[i]);
    }
}

static inline bool swr_is_index_dtype(int d) {
    return d == SWR_I8 || d == SWR_I16 || d == SWR_I32 || d == SWR_I64 || d == SWR_U8 || d == SWR_BOOL;
}
static inline bool swr_is_value_dtype(int d) { return d >= SWR_I8 && d <= SWR_BOOL; }

// Chan's parallel merge of (count, mean, M2) in fp64
struct SwrMoments {
    double n, mean, m2;
};
__device__ __forceinline__ SwrMoments swr_merge(SwrMoments a, SwrMoments b) {
    if (b.n == 0) return a;
    if (a.n == 0) return b;
    SwrMoments r;
    r.n = a.n + b.n;
    const double d = b.mean - a.mean;
    r.mean = a.mean + d * (b.n / r.n);
    r.m2 = a.m2 + b.m2 + d * d * (a.n * b.n / r.n);
    return r;
}

__device__ __forceinline__ float swr_sigmoid(float x) {
    // two-sided form: no overflow, monotone, matches torch.sigmoid to 1 ulp
    const float e = __expf(-fabsf(x));
    const float r = 1.f / (1.f + e);
    return x >= 0.f ? r : e * r;
}

// Zero-fill as an ordinary kernel node.  hipMemsetAsync becomes a memset node under hipGraph capture;
// on ROCm 7.2 those were observed to misbehave in back-to-back replays of a captured training step, so the
// library never issues memsets on the hot path.
__global__ void swr_zero_kernel(uint4* p, size_t n16, unsigned char* tail, size_t ntail);
int swr_zero_async(void* p, size_t bytes, hipStream_t st);

// d(mean BCE)/d(logit) of one row whose probability p = sigmoid(logit) was selected: dBCE/dp rounded as swr_bce_bwd rounds it
// (torch's clamp of p (1 - p) at 1e-12), times p (1 - p) as swr_select_bwd rounds it.
__device__ __forceinline__ float swr_bce_logit_grad(float pi, float yi, float dloss, int64_t M) {
    const float g = dloss * (pi - yi) / fmaxf(pi * (1.f - pi), 1e-12f) / static_cast<float>(M);
    return g * pi * (1.f - pi);
}

// Sum of one double per thread over a workgroup of NT threads (NT a multiple of 64, <= 1024), in a FIXED order: butterfly
// inside each wave, then the per-wave sums added in wave order by every thread (result in all threads).  `sm` = one
// double per wave, a different array for every call inside a kernel (no barrier protects its reuse).  Replaces the
// 8-step LDS tree (8 barriers) of the statistics finalisers.
template <int NT>
__device__ __forceinline__ double swr_block_sum_f64(double v, double* sm) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = sm[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) r += sm[w];
    return r;
}
