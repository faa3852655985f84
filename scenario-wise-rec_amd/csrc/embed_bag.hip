// K1 for SequenceFeature columns: pooled lookup of a padded id sequence per sample
// (EmbeddingLayer.forward, reference basic/layers.py:73-87 + InputMask / SumPooling / AveragePooling / ConcatPooling,
// basic/layers.py:117-146,174-228).  The reference gathers [B, L, E], builds a float mask [B, 1, L] and runs a
// batched matmul; here a workgroup stages the ids of its samples in LDS (one coalesced read of the tile's [rows, L] id
// block, any integer width, optional hash), and dim / 4 consecutive lanes per sample walk the L staged rows with 16-byte
// loads, reducing in registers:
//     sum    : out[b] = sum_l m(b,l) W[id(b,l)]
//     mean   : out[b] = (sum_l m(b,l) W[id(b,l)]) / (count_b + 1e-16)           (layers.py:204-206)
//     concat : out[b, l] = W[id(b,l)]                                            (no mask, layers.py:187)
// m(b,l) = id != padding_idx, or id != -1 when the feature has no padding_idx (layers.py:137-140).
// Also emits what the backward needs: the looked-up row of every (sample, position) (masked positions: row 0) and the
// factor its output gradient is scaled with (0 for masked positions) -- swr_embed_bag_bwd_expand turns dOut[B, .] into
// one gradient row per (sample, position), which then takes the ordinary K3 path (swr_embed_bwd with B * L "samples").
// HBM-bound: L * (idx + 4 dim) bytes read + 4 dim written per sample.
#include "common.h"

#define BAG_THREADS 256
#define BAG_MASKED 0xFFFFFFFFu

struct BagArgs {
    const float* weight;
    const void* idx;
    int64_t vocab, B, pad;
    int32_t dim, L, idx_dtype, mode, has_pad, tile_b;
    uint32_t hash_seed;
    float* out;
    int64_t ld;
    int32_t out_col;
    uint32_t* keys;
    float* wts;
    uint32_t* err;
};

__device__ __forceinline__ uint64_t bag_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <int VEC>
__global__ __launch_bounds__(BAG_THREADS) void embed_bag_fwd_kernel(const BagArgs a) {
    extern __shared__ uint32_t s_ids[];                   // [tile_b][L]: row, or BAG_MASKED
    const int tid = threadIdx.x;
    const int64_t b0 = static_cast<int64_t>(blockIdx.x) * a.tile_b;
    const int rows = static_cast<int>(min<int64_t>(a.tile_b, a.B - b0));
    const int n_ids = rows * a.L;
    for (int i = tid; i < n_ids; i += BAG_THREADS) {      // the tile's ids are one contiguous block of idx
        int64_t id = swr_load_index(a.idx, a.idx_dtype, b0 * a.L + i);
        const bool masked = a.mode != 2 && (a.has_pad ? id == a.pad : id == -1);
        uint32_t row = BAG_MASKED;
        if (!masked) {
            if (a.hash_seed != 0u)
                id = static_cast<int64_t>(bag_mix64(static_cast<uint64_t>(id) ^ a.hash_seed) % static_cast<uint64_t>(a.vocab));
            if (id < 0 || id >= a.vocab) {
                if (a.err) atomicOr(a.err, SWR_FLAG_INDEX_OOR);
                id = 0;
            }
            row = static_cast<uint32_t>(id);
        }
        s_ids[i] = row;
        if (a.keys) a.keys[b0 * a.L + i] = masked ? 0u : row;
    }
    __syncthreads();
    const int upr = a.dim / VEC;                          // lanes per sample
    const int spp = BAG_THREADS / upr;                    // samples per pass
    if (tid >= spp * upr) return;
    const int q = tid % upr;
    for (int r = tid / upr; r < rows; r += spp) {
        const uint32_t* ids = s_ids + r * a.L;
        float* o = a.out + (b0 + r) * a.ld + a.out_col + q * VEC;
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        int cnt = 0;
        for (int l = 0; l < a.L; ++l) {
            const uint32_t row = ids[l];
            if (row == BAG_MASKED) continue;
            ++cnt;
            const float* w = a.weight + static_cast<int64_t>(row) * a.dim + q * VEC;
            float x[VEC];
            if (VEC == 4) {
                const float4 t = *reinterpret_cast<const float4*>(w);
                x[0] = t.x; x[1 % VEC] = t.y; x[2 % VEC] = t.z; x[3 % VEC] = t.w;
            } else {
                x[0] = w[0];
            }
            if (a.mode == 2) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) o[l * a.dim + v] = x[v];
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] += x[v];
            }
        }
        const float scale = a.mode == 1 ? 1.f / (static_cast<float>(cnt) + 1e-16f) : 1.f;
        if (a.mode != 2) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) o[v] = a.mode == 1 ? acc[v] / (static_cast<float>(cnt) + 1e-16f) : acc[v];
        }
        if (a.wts && q == 0)
            for (int l = 0; l < a.L; ++l) a.wts[(b0 + r) * a.L + l] = ids[l] == BAG_MASKED ? 0.f : scale;
    }
}

extern "C" int swr_embed_bag_fwd(const float* weight, int64_t vocab, int dim, const void* idx, int idx_dtype, int64_t B,
                                 int L, int mode, int has_pad, int64_t padding_idx, uint32_t hash_seed, float* out,
                                 int64_t ld_out, int out_col, uint32_t* keys_out, float* wts_out, uint32_t* err_flag,
                                 void* stream) {
    SWR_REQUIRE(weight && idx && out && vocab > 0 && dim > 0 && dim <= BAG_THREADS && B >= 0 && L > 0 && L <= 4096 &&
                    mode >= 0 && mode <= 2 && ld_out > 0 && out_col >= 0, SWR_ERR_ARG);
    SWR_REQUIRE(vocab <= 0xFFFFFFFEll, SWR_ERR_UNSUPPORTED);
    SWR_REQUIRE(swr_is_index_dtype(idx_dtype), SWR_ERR_DTYPE);
    if (B == 0) return SWR_OK;
    BagArgs a;
    a.weight = weight; a.idx = idx; a.vocab = vocab; a.B = B; a.pad = padding_idx;
    a.dim = dim; a.L = L; a.idx_dtype = idx_dtype; a.mode = mode; a.has_pad = has_pad;
    a.hash_seed = hash_seed; a.out = out; a.ld = ld_out; a.out_col = out_col; a.keys = keys_out; a.wts = wts_out;
    a.err = err_flag;
    int tile = 64;
    while (tile > 1 && static_cast<int64_t>(tile) * L * 4 > 48 * 1024) tile >>= 1;       // ids of a tile fit 48 KB of LDS
    a.tile_b = tile;
    const size_t lds = static_cast<size_t>(tile) * L * 4;
    const bool vec = dim % 4 == 0 && out_col % 4 == 0 && ld_out % 4 == 0 && swr_aligned16(out) && swr_aligned16(weight);
    const dim3 grid(static_cast<unsigned>(swr_ceil_div(B, tile)));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (vec)
        hipLaunchKernelGGL(embed_bag_fwd_kernel<4>, grid, dim3(BAG_THREADS), lds, st, a);
    else
        hipLaunchKernelGGL(embed_bag_fwd_kernel<1>, grid, dim3(BAG_THREADS), lds, st, a);
    return swr_launch_status();
}

// dEx[(b, l), e] = wts[b, l] * dOut[b, col + (concat ? l * dim : 0) + e]: one gradient row per looked-up position
__global__ __launch_bounds__(BAG_THREADS) void embed_bag_expand_kernel(const float* __restrict__ dOut, int64_t ld, int col,
                                                                       int dim, int L, int concat,
                                                                       const float* __restrict__ wts, int64_t n,
                                                                       float* __restrict__ dEx) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * BAG_THREADS + threadIdx.x;
    if (i >= n * dim) return;
    const int64_t p = i / dim;                             // position (b, l)
    const int e = static_cast<int>(i - p * dim);
    const int64_t b = p / L;
    const int l = static_cast<int>(p - b * L);
    dEx[i] = wts[p] * dOut[b * ld + col + (concat ? l * dim : 0) + e];
}

extern "C" int swr_embed_bag_bwd_expand(const float* d_out, int64_t ld, int in_col, int dim, int L, int concat,
                                        const float* wts, int64_t B, float* d_rows, void* stream) {
    SWR_REQUIRE(d_out && wts && d_rows && ld > 0 && in_col >= 0 && dim > 0 && L > 0 && B >= 0, SWR_ERR_ARG);
    if (B == 0) return SWR_OK;
    const int64_t n = B * L;
    hipLaunchKernelGGL(embed_bag_expand_kernel, dim3(static_cast<unsigned>(swr_ceil_div(n * dim, BAG_THREADS))), dim3(BAG_THREADS),
                       0, static_cast<hipStream_t>(stream), d_out, ld, in_col, dim, L, concat, wts, n, d_rows);
    return swr_launch_status();
}
