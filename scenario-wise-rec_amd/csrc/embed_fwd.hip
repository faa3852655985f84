// K1: fused multi-table embedding gather  (EmbeddingLayer.forward, basic/layers.py:64-105).
//
// One launch for all sparse + dense features of a batch; every workgroup owns a tile of TILE_B
// consecutive samples and writes their [TILE_B, K0] slab of the concat layout directly
// (the reference does F_s index_selects, F_d casts and two torch.cat).
//
// HBM layout / access pattern
//   ids      : one column [B] per feature (any integer width)   -> read once, coalesced, staged in LDS
//   tables   : [V, E] fp32 row-major                            -> one 4*E-byte row read per lookup;
//              E/4 consecutive lanes read one row with 16-B loads
//   out      : [B, ld] fp32                                     -> rows written with 16-B stores, a whole
//              row (K0*4 bytes) by consecutive lanes
// Algorithmic bytes per sample: F_s*(idx_bytes + 4E) + 4F_d + 4K0, HBM-bound.
#include "common.h"

#define TILE_B 64
#define GATHER_THREADS 256
#define MAX_SPARSE 64
#define MAX_DENSE 64
#define FUSED_DENSE 32   // dense columns written by the gather kernel itself (kernarg budget: 4 KiB)

struct GatherArgs {
    swr_sparse_slot sparse[MAX_SPARSE];
    swr_dense_slot dense[FUSED_DENSE];
    int n_sparse;
    int n_dense;
    int units_per_row;   // sum over slots of dim / VEC
    int64_t B;
    float* out;
    int64_t ld;
    uint32_t* keys;
    uint32_t* err;
};

struct DenseArgs {
    swr_dense_slot dense[MAX_DENSE];
    int n_dense;
    int64_t B;
    float* out;
    int64_t ld;
};

// splitmix64 finaliser: the optional hash stage (not in the reference; oracle sees post-hash rows)
__device__ __forceinline__ uint64_t swr_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <int VEC>
__global__ __launch_bounds__(GATHER_THREADS) void embed_gather_kernel(const GatherArgs a) {
    // LDS: staged row ids of the tile, per-slot table base / stride / output column, unit -> (slot, q) map
    __shared__ uint32_t s_row[MAX_SPARSE][TILE_B];
    __shared__ const float* s_w[MAX_SPARSE];
    __shared__ int s_dim[MAX_SPARSE];
    __shared__ int s_col[MAX_SPARSE];
    __shared__ uint16_t s_unit_slot[2048];
    __shared__ uint16_t s_unit_q[2048];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int64_t b0 = static_cast<int64_t>(blockIdx.x) * TILE_B;
    const int rows = static_cast<int>(min<int64_t>(TILE_B, a.B - b0));

    // stage ids: one wave per slot at a time, 64 consecutive ids per wave-load
    for (int s = wave; s < a.n_sparse; s += GATHER_THREADS / 64) {
        const swr_sparse_slot& sl = a.sparse[s];
        uint32_t row = 0;
        if (lane < rows) {
            int64_t id = swr_load_index(sl.idx, sl.idx_dtype, b0 + lane);
            if (sl.hash_seed != 0u)
                id = static_cast<int64_t>(swr_mix64(static_cast<uint64_t>(id) ^ sl.hash_seed) % static_cast<uint64_t>(sl.vocab));
            if (id < 0 || id >= sl.vocab) {
                if (a.err) atomicOr(a.err, SWR_FLAG_INDEX_OOR);
                id = 0;
            }
            row = static_cast<uint32_t>(id);
            if (a.keys) a.keys[static_cast<int64_t>(s) * a.B + b0 + lane] = row;
        }
        s_row[s][lane] = row;
        if (lane == 0) {
            s_w[s] = sl.weight;
            s_dim[s] = sl.dim;
            s_col[s] = sl.out_col;
        }
    }
    // unit map (a unit = VEC consecutive floats of one table row): thread s lays out slot s
    if (tid < a.n_sparse) {
        int u = 0;
        for (int s = 0; s < tid; ++s) u += a.sparse[s].dim / VEC;
        for (int q = 0; q < a.sparse[tid].dim / VEC; ++q) {
            s_unit_slot[u + q] = static_cast<uint16_t>(tid);
            s_unit_q[u + q] = static_cast<uint16_t>(q);
        }
    }
    __syncthreads();

    const int upr = a.units_per_row;
    // U lanes walk along a row (consecutive units -> consecutive addresses of `out`); the other
    // GATHER_THREADS / U row groups take different samples of the tile
    int U = 1;
    while (U < upr && U < GATHER_THREADS) U <<= 1;
    const int nrg = GATHER_THREADS / U;
    const int rg = tid / U;
    for (int j = tid % U; j < upr; j += U) {
        const int s = s_unit_slot[j], q = s_unit_q[j];
        const float* w = s_w[s] + q * VEC;
        const int dim = s_dim[s];
        float* o = a.out + b0 * a.ld + s_col[s] + q * VEC;
#pragma unroll 4
        for (int r = rg; r < rows; r += nrg) {
            const uint32_t row = s_row[s][r];
            if (VEC == 4) {
                const float4 v = *reinterpret_cast<const float4*>(w + static_cast<int64_t>(row) * dim);
                *reinterpret_cast<float4*>(o + r * a.ld) = v;
            } else {
                o[r * a.ld] = w[static_cast<int64_t>(row) * dim];
            }
        }
    }
    // x[name].float() columns (layers.py:88-89), the tail of each row of the slab
    for (int t = tid; t < rows * a.n_dense; t += GATHER_THREADS) {
        const int r = t / a.n_dense, s = t - r * a.n_dense;
        a.out[(b0 + r) * a.ld + a.dense[s].out_col] = swr_load_value(a.dense[s].values, a.dense[s].dtype, b0 + r);
    }
}

__global__ __launch_bounds__(GATHER_THREADS) void dense_cast_kernel(const DenseArgs a) {
    // x[name].float() columns (layers.py:88-89): thread per (sample, dense feature); consecutive lanes
    // take consecutive dense columns of one sample -> contiguous 4*F_d-byte writes
    const int64_t i = static_cast<int64_t>(blockIdx.x) * GATHER_THREADS + threadIdx.x;
    const int64_t b = i / a.n_dense;
    const int s = static_cast<int>(i - b * a.n_dense);
    if (b >= a.B) return;
    a.out[b * a.ld + a.dense[s].out_col] = swr_load_value(a.dense[s].values, a.dense[s].dtype, b);
}

extern "C" int swr_embed_gather_fwd(const swr_sparse_slot* sparse, int n_sparse, const swr_dense_slot* dense,
                                    int n_dense, int64_t B, float* out, int64_t ld_out, uint32_t* keys_out,
                                    uint32_t* err_flag, void* stream) {
    SWR_REQUIRE(B >= 0 && n_sparse >= 0 && n_dense >= 0 && out != nullptr && ld_out > 0, SWR_ERR_ARG);
    SWR_REQUIRE((n_sparse == 0 || sparse) && (n_dense == 0 || dense), SWR_ERR_ARG);
    if (B == 0) return SWR_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);

    for (int s = 0; s < n_dense; ++s) {
        SWR_REQUIRE(dense[s].values && dense[s].out_col >= 0, SWR_ERR_ARG);
        SWR_REQUIRE(swr_is_value_dtype(dense[s].dtype), SWR_ERR_DTYPE);
    }
    int dense_done = 0;
    for (int s0 = 0; s0 < n_sparse; s0 += MAX_SPARSE) {
        GatherArgs a;
        a.n_sparse = n_sparse - s0 < MAX_SPARSE ? n_sparse - s0 : MAX_SPARSE;
        bool vec = swr_aligned16(out) && (ld_out % 4 == 0);
        for (int s = 0; s < a.n_sparse; ++s) {
            const swr_sparse_slot& sl = sparse[s0 + s];
            SWR_REQUIRE(sl.weight && sl.idx && sl.vocab > 0 && sl.dim > 0 && sl.out_col >= 0, SWR_ERR_ARG);
            SWR_REQUIRE(sl.vocab <= 0xFFFFFFFFll, SWR_ERR_UNSUPPORTED);
            SWR_REQUIRE(swr_is_index_dtype(sl.idx_dtype), SWR_ERR_DTYPE);
            vec = vec && (sl.dim % 4 == 0) && (sl.out_col % 4 == 0) && swr_aligned16(sl.weight);
            a.sparse[s] = sl;
        }
        const int V = vec ? 4 : 1;
        int upr = 0;
        for (int s = 0; s < a.n_sparse; ++s) upr += a.sparse[s].dim / V;
        SWR_REQUIRE(upr <= 2048, SWR_ERR_UNSUPPORTED);
        a.units_per_row = upr;
        a.n_dense = 0;
        if (s0 == 0) {
            a.n_dense = n_dense < FUSED_DENSE ? n_dense : FUSED_DENSE;
            for (int s = 0; s < a.n_dense; ++s) a.dense[s] = dense[s];
            dense_done = a.n_dense;
        }
        a.B = B;
        a.out = out;
        a.ld = ld_out;
        a.keys = keys_out ? keys_out + static_cast<int64_t>(s0) * B : nullptr;
        a.err = err_flag;
        const dim3 grid(static_cast<unsigned>(swr_ceil_div(B, TILE_B)));
        if (vec)
            hipLaunchKernelGGL(embed_gather_kernel<4>, grid, dim3(GATHER_THREADS), 0, st, a);
        else
            hipLaunchKernelGGL(embed_gather_kernel<1>, grid, dim3(GATHER_THREADS), 0, st, a);
    }
    for (int s0 = dense_done; s0 < n_dense; s0 += MAX_DENSE) {
        DenseArgs d;
        d.n_dense = n_dense - s0 < MAX_DENSE ? n_dense - s0 : MAX_DENSE;
        for (int s = 0; s < d.n_dense; ++s) d.dense[s] = dense[s0 + s];
        d.B = B;
        d.out = out;
        d.ld = ld_out;
        const int64_t n = B * d.n_dense;
        hipLaunchKernelGGL(dense_cast_kernel, dim3(static_cast<unsigned>(swr_ceil_div(n, GATHER_THREADS))),
                           dim3(GATHER_THREADS), 0, st, d);
    }
    return swr_launch_status();
}
