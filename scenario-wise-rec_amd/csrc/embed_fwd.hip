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

#define OH_MAX 256         // widest one-hot block (columns)
#define OH_NONE 255

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
    // one-hot block behind the concat (see swr_embed_gather_fwd_onehot): columns [oh_col, oh_col + oh_width) of `out`,
    // slot s owns columns oh_col + oh_off[s] .. + vocab (oh_off < 0: not a one-hot slot); [pad_col, oh_col) is zeroed
    int oh_col, oh_width, pad_col;
    int16_t oh_off[MAX_SPARSE];
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
    __shared__ uint8_t s_oh_slot[OH_MAX];
    __shared__ uint16_t s_oh_val[OH_MAX];

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
    if (a.oh_width > 0) {                              // column of the one-hot block -> (slot, row value)
        for (int c = tid; c < a.oh_width; c += GATHER_THREADS) s_oh_slot[c] = OH_NONE;
    }
    __syncthreads();
    if (a.oh_width > 0 && tid < a.n_sparse && a.oh_off[tid] >= 0) {
        const int v = static_cast<int>(a.sparse[tid].vocab);
        for (int j = 0; j < v; ++j) {
            s_oh_slot[a.oh_off[tid] + j] = static_cast<uint8_t>(tid);
            s_oh_val[a.oh_off[tid] + j] = static_cast<uint16_t>(j);
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
    // U = units_per_row itself when it fits (not the next power of two: at 36 units per row -- the folded KuaiRand
    // layout -- a 64-lane group idles 28 lanes; 7 groups of 36 lanes idle 4 of 256)
    const int U = max(1, min(upr, GATHER_THREADS));
    const int nrg = GATHER_THREADS / U;
    const int rg = tid / U;
    if (rg >= nrg) goto dense_part;
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
dense_part:
    // x[name].float() columns (layers.py:88-89), the tail of each row of the slab
    for (int t = tid; t < rows * a.n_dense; t += GATHER_THREADS) {
        const int r = t / a.n_dense, s = t - r * a.n_dense;
        a.out[(b0 + r) * a.ld + a.dense[s].out_col] = swr_load_value(a.dense[s].values, a.dense[s].dtype, b0 + r);
    }
    // one-hot block of the small tables: out[b, oh_col + off_s + v] = (row_s(b) == v); 16-byte stores, 4 columns per lane
    if (a.oh_width > 0) {
        const int upr4 = a.oh_width / 4;
        for (int t = tid; t < rows * upr4; t += GATHER_THREADS) {
            const int r = t / upr4, q = t - r * upr4;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = 4 * q + j;
                const int sl = s_oh_slot[c];
                v[j] = (sl != OH_NONE && s_row[sl][r] == s_oh_val[c]) ? 1.f : 0.f;
            }
            *reinterpret_cast<float4*>(a.out + (b0 + r) * a.ld + a.oh_col + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
        }
        const int npad = a.oh_col - a.pad_col;          // alignment columns between the concat and the block: zeros
        for (int t = tid; t < rows * npad; t += GATHER_THREADS)
            a.out[(b0 + t / npad) * a.ld + a.pad_col + t % npad] = 0.f;
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
    return swr_embed_gather_fwd_onehot(sparse, n_sparse, dense, n_dense, B, out, ld_out, keys_out, nullptr, 0, 0, 0, err_flag,
                                       stream);
}

extern "C" int swr_embed_gather_fwd_onehot(const swr_sparse_slot* sparse, int n_sparse, const swr_dense_slot* dense,
                                           int n_dense, int64_t B, float* out, int64_t ld_out, uint32_t* keys_out,
                                           const int32_t* oh_off, int pad_col, int oh_col, int oh_width,
                                           uint32_t* err_flag, void* stream) {
    SWR_REQUIRE(B >= 0 && n_sparse >= 0 && n_dense >= 0 && out != nullptr && ld_out > 0, SWR_ERR_ARG);
    if (oh_width > 0) {
        SWR_REQUIRE(oh_off && n_sparse <= MAX_SPARSE && n_sparse > 0 && oh_width <= OH_MAX && oh_width % 4 == 0 && oh_col % 4 == 0 &&
                        pad_col >= 0 && pad_col <= oh_col && oh_col - pad_col < 16 && oh_col + oh_width <= ld_out && ld_out % 4 == 0 &&
                        swr_aligned16(out), SWR_ERR_ARG);
        for (int s = 0; s < n_sparse; ++s)
            SWR_REQUIRE(oh_off[s] < 0 || (sparse[s].hash_seed == 0 && sparse[s].vocab <= 0xFFFF &&
                                          oh_off[s] + sparse[s].vocab <= oh_width), SWR_ERR_ARG);
    }
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
            SWR_REQUIRE(sl.weight && sl.idx && sl.vocab > 0 && sl.out_col >= 0, SWR_ERR_ARG);
            // dim == 0: the slot only feeds the one-hot block (its embedding is folded into the consuming layer's weights)
            SWR_REQUIRE(sl.dim > 0 || (sl.dim == 0 && oh_width > 0 && oh_off[s0 + s] >= 0), SWR_ERR_ARG);
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
        a.oh_width = 0; a.oh_col = 0; a.pad_col = 0;
        if (oh_width > 0) {                              // (n_sparse <= MAX_SPARSE: one launch)
            a.oh_width = oh_width; a.oh_col = oh_col; a.pad_col = pad_col;
            for (int s = 0; s < a.n_sparse; ++s) a.oh_off[s] = static_cast<int16_t>(oh_off[s]);
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


// ---- gradients of the small (one-hot) tables from the segment sums of dZ: grad_t[v, e] (+)= sum_n S[n, off_t + v] W[n, col_t + e]
#define OHT_MAX 64
struct OhtK {
    swr_onehot_table tab[OHT_MAX];
    int first[OHT_MAX + 1];          // first output element of each table
    int n_tables, N, accumulate;
    const float* S; int64_t lds;
    const float* W; int64_t ldw;
};

// one output element per 16 consecutive lanes: lane j of the group takes n = j, j + 16, ... (independent loads in flight,
// fused multiply-adds in ascending n), then a fixed butterfly over the 16 partials -- deterministic
#define OHT_LANES 16
__device__ __forceinline__ void onehot_table_grads_body(const OhtK& k, unsigned block) {
    const int gidx = (block * GATHER_THREADS + threadIdx.x) / OHT_LANES;
    const int j0 = threadIdx.x % OHT_LANES;
    const int total = k.first[k.n_tables];
    const bool live = gidx < total;
    const int i = live ? gidx : total - 1;
    int t = 0;
    while (i >= k.first[t + 1]) ++t;
    const swr_onehot_table& T = k.tab[t];
    const int j = i - k.first[t];
    const int v = j / T.dim, e = j - v * T.dim;
    const float* s = k.S + T.oh_off + v;
    const float* w = k.W + T.w_col + e;
    float acc = 0.f;
    // four steps' loads in flight at once, the multiply-adds in the same ascending order (same bits): the plain loop waited for
    // each step's two loads -- ten dependent round trips for N = 148 at the very end of the weight-gradient branch
    for (int n0 = j0; n0 < k.N; n0 += 4 * OHT_LANES) {
        float sv[4], wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int n = n0 + u * OHT_LANES;
            const bool ok = n < k.N;
            sv[u] = ok ? s[n * k.lds] : 0.f;
            wv[u] = ok ? w[n * k.ldw] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (n0 + u * OHT_LANES < k.N) acc = fmaf(sv[u], wv[u], acc);
    }
#pragma unroll
    for (int o = 1; o < OHT_LANES; o <<= 1) acc += __shfl_xor(acc, o, 64);
    if (live && j0 == 0) {
        float* dst = T.grad + static_cast<int64_t>(v) * T.dim + e;
        *dst = k.accumulate ? *dst + acc : acc;
    }
}
__global__ __launch_bounds__(GATHER_THREADS) void onehot_table_grads_kernel(const OhtK k) { onehot_table_grads_body(k, blockIdx.x); }

static int oht_fill(OhtK& k, const float* S, int64_t lds, const float* W, int64_t ldw, int N, const swr_onehot_table* tables, int n_tables,
                    int accumulate) {
    k.n_tables = n_tables; k.N = N; k.accumulate = accumulate; k.S = S; k.lds = lds; k.W = W; k.ldw = ldw;
    int pos = 0;
    for (int t = 0; t < n_tables; ++t) {
        const swr_onehot_table& T = tables[t];
        SWR_REQUIRE(T.grad && T.vocab > 0 && T.dim > 0 && T.oh_off >= 0 && T.w_col >= 0, SWR_ERR_ARG);
        k.tab[t] = T;
        k.first[t] = pos;
        pos += T.vocab * T.dim;
    }
    k.first[n_tables] = pos;
    return SWR_OK;
}

extern "C" int swr_onehot_table_grads(const float* S, int64_t lds, const float* W, int64_t ldw, int N,
                                      const swr_onehot_table* tables, int n_tables, int accumulate, void* stream) {
    SWR_REQUIRE(S && W && tables && N > 0 && n_tables >= 0 && lds > 0 && ldw > 0, SWR_ERR_ARG);
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int t0 = 0; t0 < n_tables; t0 += OHT_MAX) {
        OhtK k;
        k.n_tables = n_tables - t0 < OHT_MAX ? n_tables - t0 : OHT_MAX;
        k.N = N; k.accumulate = accumulate; k.S = S; k.lds = lds; k.W = W; k.ldw = ldw;
        int pos = 0;
        for (int t = 0; t < k.n_tables; ++t) {
            const swr_onehot_table& T = tables[t0 + t];
            SWR_REQUIRE(T.grad && T.vocab > 0 && T.dim > 0 && T.oh_off >= 0 && T.w_col >= 0, SWR_ERR_ARG);
            k.tab[t] = T;
            k.first[t] = pos;
            pos += T.vocab * T.dim;
        }
        k.first[k.n_tables] = pos;
        if (pos == 0) continue;
        hipLaunchKernelGGL(onehot_table_grads_kernel,
                           dim3(static_cast<unsigned>(swr_ceil_div(static_cast<int64_t>(pos) * OHT_LANES, GATHER_THREADS))),
                           dim3(GATHER_THREADS), 0, st, k);
    }
    return swr_launch_status();
}


// ---- first layer folded over the one-hot block (include/swr.h "folded first layer")
__device__ __forceinline__ bool swr_aligned16_dev(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

struct FoldK {
    swr_onehot_table tab[OHT_MAX];      // .grad = the table's weights here (forward) / dEmb is not touched by these kernels
    int n_tables, N, Kp, ohw, K, accumulate;
    const int32_t* src_col;             // [Kp]: column of W behind compact column j, -1 = zero padding
    const int32_t* inv_col;             // [K] : compact column of W's column k, or -1 - t for a column of small table t
    const float* W; int64_t ldw;
    float* Wp; int64_t ldwp;            // forward: folded weights [N, Kp + ohw]
    const int64_t* sel; int n_sel; float* Wt; int64_t ldt;   // forward: rows `sel` of W^T for the backward's dX product (nullable)
    const float* dWp; int64_t lddwp;    // backward: gradient of the folded weights
    const float* dbp;                   // backward: bias gradient (column sums of dZ), nullable
    float* dW; int64_t lddw; float* db;
};

// Wp[n, j] = W[n, src_col[j]] (j < Kp);  Wp[n, Kp + off_t + v] = sum_e emb_t[v, e] W[n, col_t + e]
// thread = one output; the dot product's 2 x dim loads are independent 16-byte loads (the table of a one-hot column comes
// from `oh_table`, not from a search)
__global__ __launch_bounds__(GATHER_THREADS) void fold_fwd_kernel(const FoldK k, const int32_t* __restrict__ oh_table) {
    const int width = k.Kp + k.ohw;
    int64_t i = static_cast<int64_t>(blockIdx.x) * GATHER_THREADS + threadIdx.x;
    if (i >= static_cast<int64_t>(k.N) * width) {
        // the tail of the grid transposes the selected columns of W (thread = one element, consecutive lanes along n)
        i -= static_cast<int64_t>(k.N) * width;
        if (i < static_cast<int64_t>(k.n_sel) * k.N) {
            const int r = static_cast<int>(i / k.N), n = static_cast<int>(i - static_cast<int64_t>(r) * k.N);
            k.Wt[r * k.ldt + n] = k.W[n * k.ldw + k.sel[r]];
        }
        return;
    }
    const int n = static_cast<int>(i / width), j = static_cast<int>(i - static_cast<int64_t>(n) * width);
    float v = 0.f;
    if (j < k.Kp) {
        const int c = k.src_col[j];
        if (c >= 0) v = k.W[n * k.ldw + c];
    } else {
        const int o = j - k.Kp;
        const int t = oh_table[o];
        if (t >= 0) {
            const swr_onehot_table& T = k.tab[t];
            const float* __restrict__ e = T.grad + static_cast<int64_t>(o - T.oh_off) * T.dim;
            const float* __restrict__ w = k.W + n * k.ldw + T.w_col;
            if ((T.dim & 3) == 0 && (T.w_col & 3) == 0 && (k.ldw & 3) == 0 && swr_aligned16_dev(k.W) && swr_aligned16_dev(T.grad)) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                for (int q = 0; q < T.dim; q += 4) {
                    const float4 ev = *reinterpret_cast<const float4*>(e + q), wv = *reinterpret_cast<const float4*>(w + q);
                    a0 = fmaf(ev.x, wv.x, a0); a1 = fmaf(ev.y, wv.y, a1); a2 = fmaf(ev.z, wv.z, a2); a3 = fmaf(ev.w, wv.w, a3);
                }
                v = (a0 + a1) + (a2 + a3);
            } else {
                for (int q = 0; q < T.dim; ++q) v = fmaf(e[q], w[q], v);
            }
        }
    }
    k.Wp[n * k.ldwp + j] = v;
}

// dW[n, c] (+)= dWp[n, inv_col[c]]  or, for a column of small table t,  sum_v dWp[n, Kp + off_t + v] emb_t[v, c - col_t];
// db (+)= dbp
__device__ __forceinline__ void fold_bwd_body(const FoldK& k, unsigned block) {
    const int64_t i = static_cast<int64_t>(block) * GATHER_THREADS + threadIdx.x;
    if (i < k.N && k.db && k.dbp) k.db[i] = k.accumulate ? k.db[i] + k.dbp[i] : k.dbp[i];
    if (i >= static_cast<int64_t>(k.N) * k.K) return;
    const int n = static_cast<int>(i / k.K), c = static_cast<int>(i - static_cast<int64_t>(n) * k.K);
    const int m = k.inv_col[c];
    float v = 0.f;
    if (m >= 0) {
        v = k.dWp[n * k.lddwp + m];
    } else {
        const swr_onehot_table& T = k.tab[-1 - m];
        const int e = c - T.w_col;
        const float* s = k.dWp + n * k.lddwp + k.Kp + T.oh_off;
        // (eight rows' loads in flight, multiply-adds in row order: same bits as the one-row-at-a-time loop, an eighth of its round trips)
        const int vocab = T.vocab, dim = T.dim;
        const float* __restrict__ g = T.grad + e;
        for (int q0 = 0; q0 < vocab; q0 += 8) {
            float sv[8], gv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool ok = q0 + u < vocab;
                sv[u] = ok ? s[q0 + u] : 0.f;
                gv[u] = ok ? g[static_cast<int64_t>(q0 + u) * dim] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (q0 + u < vocab) v = fmaf(sv[u], gv[u], v);
        }
    }
    float* dst = k.dW + n * k.lddw + c;
    *dst = k.accumulate ? *dst + v : v;
}
__global__ __launch_bounds__(GATHER_THREADS) void fold_bwd_kernel(const FoldK k) { fold_bwd_body(k, blockIdx.x); }
// the unfolding and the small tables' gradients read the same reduced dWp and nothing of each other: one launch, the first
// `nb_fold` workgroups unfold
__global__ __launch_bounds__(GATHER_THREADS) void fold_bwd_tables_kernel(const FoldK k, const OhtK o, unsigned nb_fold) {
    if (blockIdx.x < nb_fold) fold_bwd_body(k, blockIdx.x);
    else onehot_table_grads_body(o, blockIdx.x - nb_fold);
}

static int fold_fill(FoldK& k, const swr_onehot_table* tables, int n_tables, int N, int K, int Kp, int ohw, const int32_t* src_col,
                     const int32_t* inv_col, const float* W, int64_t ldw) {
    SWR_REQUIRE(tables && n_tables > 0 && n_tables <= OHT_MAX && N > 0 && K > 0 && Kp >= 0 && ohw > 0 && src_col && inv_col &&
                    (W == nullptr || ldw >= K), SWR_ERR_ARG);
    for (int t = 0; t < n_tables; ++t) {
        SWR_REQUIRE(tables[t].grad && tables[t].vocab > 0 && tables[t].dim > 0 && tables[t].oh_off >= 0 &&
                        tables[t].oh_off + tables[t].vocab <= ohw && tables[t].w_col >= 0 && tables[t].w_col + tables[t].dim <= K,
                    SWR_ERR_ARG);
        k.tab[t] = tables[t];
    }
    k.n_tables = n_tables; k.N = N; k.K = K; k.Kp = Kp; k.ohw = ohw; k.src_col = src_col; k.inv_col = inv_col; k.W = W; k.ldw = ldw;
    return SWR_OK;
}

extern "C" int swr_fold_first_layer_fwd(const float* W, int64_t ldw, int N, int K, int Kp, int ohw, const int32_t* src_col,
                                        const int32_t* inv_col, const int32_t* oh_table, const swr_onehot_table* tables,
                                        int n_tables, float* Wp, int64_t ldwp, const int64_t* sel, int n_sel, float* Wt_sel,
                                        int64_t ldt, void* stream) {
    FoldK k;
    int rc = fold_fill(k, tables, n_tables, N, K, Kp, ohw, src_col, inv_col, W, ldw);
    if (rc != SWR_OK) return rc;
    SWR_REQUIRE(W && Wp && oh_table && ldwp >= Kp + ohw, SWR_ERR_ARG);
    SWR_REQUIRE(n_sel >= 0 && (n_sel == 0 || (sel && Wt_sel && ldt >= N)), SWR_ERR_ARG);
    k.Wp = Wp; k.ldwp = ldwp;
    k.sel = sel; k.n_sel = n_sel; k.Wt = Wt_sel; k.ldt = ldt;
    const int64_t total = static_cast<int64_t>(N) * (Kp + ohw) + static_cast<int64_t>(n_sel) * N;
    hipLaunchKernelGGL(fold_fwd_kernel, dim3(static_cast<unsigned>(swr_ceil_div(total, GATHER_THREADS))), dim3(GATHER_THREADS), 0,
                       static_cast<hipStream_t>(stream), k, oh_table);
    return swr_launch_status();
}

extern "C" int swr_fold_first_layer_bwd(const float* dWp, int64_t lddwp, const float* dbp, int N, int K, int Kp, int ohw,
                                        const int32_t* src_col, const int32_t* inv_col, const swr_onehot_table* tables,
                                        int n_tables, float* dW, int64_t lddw, float* db, int accumulate, void* stream) {
    FoldK k;
    int rc = fold_fill(k, tables, n_tables, N, K, Kp, ohw, src_col, inv_col, nullptr, 0);   // (W is not read here)
    if (rc != SWR_OK) return rc;
    SWR_REQUIRE(dWp && dW && lddwp >= Kp + ohw && lddw >= K, SWR_ERR_ARG);
    k.dWp = dWp; k.lddwp = lddwp; k.dbp = dbp; k.dW = dW; k.lddw = lddw; k.db = db; k.accumulate = accumulate;
    const int64_t total = static_cast<int64_t>(N) * K;
    hipLaunchKernelGGL(fold_bwd_kernel, dim3(static_cast<unsigned>(swr_ceil_div(total, GATHER_THREADS))), dim3(GATHER_THREADS), 0,
                       static_cast<hipStream_t>(stream), k);
    return swr_launch_status();
}

extern "C" int swr_fold_first_layer_bwd_tables(const float* dWp, int64_t lddwp, const float* dbp, int N, int K, int Kp, int ohw,
                                               const int32_t* src_col, const int32_t* inv_col, const swr_onehot_table* tables,
                                               int n_tables, float* dW, int64_t lddw, float* db, int accumulate, const float* W,
                                               int64_t ldw, const swr_onehot_table* grad_tables, int n_grad_tables, void* stream) {
    FoldK k;
    int rc = fold_fill(k, tables, n_tables, N, K, Kp, ohw, src_col, inv_col, nullptr, 0);
    if (rc != SWR_OK) return rc;
    SWR_REQUIRE(dWp && dW && W && grad_tables && lddwp >= Kp + ohw && lddw >= K && ldw > 0 && n_grad_tables > 0 && n_grad_tables <= OHT_MAX,
                SWR_ERR_ARG);
    k.dWp = dWp; k.lddwp = lddwp; k.dbp = dbp; k.dW = dW; k.lddw = lddw; k.db = db; k.accumulate = accumulate;
    OhtK o;
    rc = oht_fill(o, dWp, lddwp, W, ldw, N, grad_tables, n_grad_tables, accumulate);
    if (rc != SWR_OK) return rc;
    const unsigned nb_fold = static_cast<unsigned>(swr_ceil_div(static_cast<int64_t>(N) * K, GATHER_THREADS));
    const unsigned nb_oh = static_cast<unsigned>(swr_ceil_div(static_cast<int64_t>(o.first[n_grad_tables]) * OHT_LANES, GATHER_THREADS));
    hipLaunchKernelGGL(fold_bwd_tables_kernel, dim3(nb_fold + nb_oh), dim3(GATHER_THREADS), 0, static_cast<hipStream_t>(stream), k, o, nb_fold);
    return swr_launch_status();
}
