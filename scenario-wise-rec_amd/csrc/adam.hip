// torch.optim.Adam (L2 weight decay added to the gradient) of the reference trainer
// (trainers/ctr_trainer.py:50-52,73) as HBM-streaming kernels: 4 reads + 3 writes of 4 bytes per
// parameter.  Step count / bias corrections live in device memory so a captured hipGraph replays.
#include "common.h"
#include "adam_advance.h"

#define AD_THREADS 256

__global__ void adam_advance_kernel(swr_adam_hyper* h, float* hist, int64_t cap) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    swr_adam_advance_body(h, hist, cap);
}

extern "C" int swr_adam_advance(swr_adam_hyper* hyper, float* hist, int64_t hist_cap, void* stream) {
    SWR_REQUIRE(hyper != nullptr && (hist == nullptr || (hist_cap > 1 && hist_cap <= (1ll << 31) && (hist_cap & (hist_cap - 1)) == 0)),
                SWR_ERR_ARG);
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), hyper, hist, hist_cap);
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

// clear_grad: the gradient is consumed -- zero it behind the update, so that the next step's zero_grad has nothing to fill
// (its launch sat on a side branch the backward pass had to wait for)
__global__ __launch_bounds__(AD_THREADS) void adam_dense_kernel(float* __restrict__ p, float* __restrict__ g,
                                                                float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                                int clear_grad, const swr_adam_hyper* __restrict__ hp) {
    const swr_adam_hyper h = *hp;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * AD_THREADS;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * AD_THREADS + threadIdx.x; i < n; i += stride) {
        float pi = p[i], mi = m[i], vi = v[i];
        adam_elem(pi, g[i], mi, vi, h);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (clear_grad) g[i] = 0.f;
    }
}

extern "C" int swr_adam_dense(float* p, float* g, float* m, float* v, int64_t n, int clear_grad, const swr_adam_hyper* hyper,
                              void* stream) {
    SWR_REQUIRE(p && g && m && v && hyper && n >= 0, SWR_ERR_ARG);
    if (n == 0) return SWR_OK;
    const unsigned grid = static_cast<unsigned>(swr_ceil_div(n, AD_THREADS) < 4096 ? swr_ceil_div(n, AD_THREADS) : 4096);
    hipLaunchKernelGGL(adam_dense_kernel, dim3(grid), dim3(AD_THREADS), 0, static_cast<hipStream_t>(stream), p, g, m, v, n,
                       clear_grad, hyper);
    return swr_launch_status();
}

// touched rows of a large table: thread per (entry, column)
__global__ __launch_bounds__(AD_THREADS) void adam_rows_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                               int64_t vocab, int dim, const int32_t* __restrict__ urow,
                                                               const float* __restrict__ ugrad, int64_t n_entries,
                                                               uint32_t* __restrict__ bitmap, int32_t* __restrict__ last,
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
    if (e == 0) {
        if (bitmap) atomicOr(bitmap + (row >> 5), 1u << (row & 31));
        if (last) last[row] = static_cast<int32_t>(h.step);        // lazy tables: this row is current as of this step
    }
}

extern "C" int swr_adam_rows(float* p, float* m, float* v, int64_t vocab, int dim, const int32_t* urow, const float* ugrad,
                             int64_t n_entries, uint32_t* bitmap, int32_t* last, const swr_adam_hyper* hyper, void* stream) {
    SWR_REQUIRE(p && m && v && urow && ugrad && (bitmap || last) && hyper && vocab > 0 && dim > 0 && n_entries >= 0, SWR_ERR_ARG);
    if (n_entries == 0) return SWR_OK;
    hipLaunchKernelGGL(adam_rows_kernel, dim3(static_cast<unsigned>(swr_ceil_div(n_entries * dim, AD_THREADS))),
                       dim3(AD_THREADS), 0, static_cast<hipStream_t>(stream), p, m, v, vocab, dim, urow, ugrad, n_entries,
                       bitmap, last, hyper);
    return swr_launch_status();
}

// the two launches above as ONE (the common step: one contiguous parameter arena + one large lazily updated table):
// workgroups [0, dense_blocks) stream the arena, the rest take the row entries -- at config 2 each launch is pure
// latency (~5 us), so one fewer is what this saves
__global__ __launch_bounds__(AD_THREADS) void adam_dense_rows_kernel(float* __restrict__ dp, float* __restrict__ dg,
                                                                     float* __restrict__ dm, float* __restrict__ dv, int64_t dn,
                                                                     int clear_grad, int dense_blocks, float* __restrict__ p, float* __restrict__ m,
                                                                     float* __restrict__ v, int64_t vocab, int dim,
                                                                     const int32_t* __restrict__ urow, const float* __restrict__ ugrad,
                                                                     int64_t n_entries, int32_t* __restrict__ last,
                                                                     const swr_adam_hyper* __restrict__ hp) {
    const swr_adam_hyper h = *hp;
    if (static_cast<int>(blockIdx.x) < dense_blocks) {
        const int64_t stride = static_cast<int64_t>(dense_blocks) * AD_THREADS;
        for (int64_t i = static_cast<int64_t>(blockIdx.x) * AD_THREADS + threadIdx.x; i < dn; i += stride) {
            float pi = dp[i], mi = dm[i], vi = dv[i];
            adam_elem(pi, dg[i], mi, vi, h);
            dp[i] = pi; dm[i] = mi; dv[i] = vi;
            if (clear_grad) dg[i] = 0.f;
        }
        return;
    }
    const int64_t idx = static_cast<int64_t>(blockIdx.x - dense_blocks) * AD_THREADS + threadIdx.x;
    const int64_t i = idx / dim;
    if (i >= n_entries) return;
    const int e = static_cast<int>(idx - i * dim);
    const int32_t row = urow[i];
    if (row < 0 || row >= vocab) return;
    const int64_t o = static_cast<int64_t>(row) * dim + e;
    float pi = p[o], mi = m[o], vi = v[o];
    adam_elem(pi, ugrad[i * dim + e], mi, vi, h);
    p[o] = pi; m[o] = mi; v[o] = vi;
    if (e == 0) last[row] = static_cast<int32_t>(h.step);
}

extern "C" int swr_adam_dense_rows(float* dp, float* dg, float* dm, float* dv, int64_t dn, int clear_grad, float* p, float* m, float* v,
                                   int64_t vocab, int dim, const int32_t* urow, const float* ugrad, int64_t n_entries,
                                   int32_t* last, const swr_adam_hyper* hyper, void* stream) {
    SWR_REQUIRE(dp && dg && dm && dv && dn > 0 && p && m && v && urow && ugrad && last && hyper && vocab > 0 && dim > 0 &&
                    n_entries > 0, SWR_ERR_ARG);
    const int dense_blocks = static_cast<int>(swr_ceil_div(dn, AD_THREADS) < 4096 ? swr_ceil_div(dn, AD_THREADS) : 4096);
    const int64_t row_blocks = swr_ceil_div(n_entries * dim, AD_THREADS);
    hipLaunchKernelGGL(adam_dense_rows_kernel, dim3(static_cast<unsigned>(dense_blocks + row_blocks)), dim3(AD_THREADS), 0,
                       static_cast<hipStream_t>(stream), dp, dg, dm, dv, dn, clear_grad, dense_blocks, p, m, v, vocab, dim, urow, ugrad,
                       n_entries, last, hyper);
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

// ------------------------------------------------------------------------------ lazy (exact) row updates
// A row of a large table that is not looked up in a step still moves under the reference's dense Adam:
// g = weight_decay * p.  That update depends on nothing but the row's own (p, m, v) and the step's two scalars, so it
// can be applied LATER, bit for bit: `last[row]` records the step the row is current for, `hist[s]` the scalars of step
// s.  A row is caught up (i) right before it is looked up (swr_adam_catchup_rows, called by the forward gather) and
// (ii) when the whole table is materialised (swr_adam_flush: checkpoints).  The per-step sweep over the entire table
// (6 x 280 MB of HBM traffic at the KuaiRand config, 25 GB x 6 at the 100 M-row config) disappears; the arithmetic is
// the same sequence of fp32 operations the sweep would have executed.
__device__ __forceinline__ void adam_replay(float& p, float& m, float& v, int from, int to, const float* __restrict__ hist,
                                            swr_adam_hyper h) {
    const uint32_t mask = h.hist_mask;        // ring of mask + 1 steps; the caller guarantees to - from <= mask
    for (int s = from + 1; s <= to; ++s) {
        const uint32_t slot = static_cast<uint32_t>(s) & mask;
        h.step_size = hist[2 * slot];
        h.inv_bc2_sqrt = hist[2 * slot + 1];
        adam_elem(p, 0.f, m, v, h);
    }
}

// phase 1: one thread per looked-up id.  Rows that are behind get a "ticket": claim[row] = i with a PLAIN store -- the
// last writer wins, which elects exactly one of the (possibly thousands of, for a hot row) duplicate lookups without
// a single atomic.  Stale tickets of earlier steps are harmless: a thread only looks at the ticket of a row it found
// behind in this launch, and then at least its own store has overwritten the old value.
__global__ __launch_bounds__(AD_THREADS) void adam_claim_kernel(const void* __restrict__ idx, int idx_dtype, uint32_t hash_seed,
                                                                int64_t n, int64_t vocab, const int32_t* __restrict__ last,
                                                                int32_t* __restrict__ claim, int32_t* __restrict__ from,
                                                                uint32_t* __restrict__ rows,
                                                                const swr_adam_hyper* __restrict__ hp) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * AD_THREADS + threadIdx.x;
    if (i >= n) return;
    int64_t id = swr_load_index(idx, idx_dtype, i);
    if (hash_seed != 0u) {
        uint64_t z = static_cast<uint64_t>(id) ^ hash_seed;
        z += 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        id = static_cast<int64_t>((z ^ (z >> 31)) % static_cast<uint64_t>(vocab));
    }
    const int now = static_cast<int>(hp->step);
    int f = now;                             // out-of-range ids: the gather flags them; nothing to replay
    if (id >= 0 && id < vocab) {
        f = last[id];
        rows[i] = static_cast<uint32_t>(id);
        if (f < now) claim[id] = static_cast<int32_t>(i);
    }
    from[i] = f;
}

// phase 2: one thread per (looked-up id, column); only the ticket holder of a row replays it
__global__ __launch_bounds__(AD_THREADS) void adam_catchup_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                                  int dim, const int32_t* __restrict__ from,
                                                                  const uint32_t* __restrict__ rows,
                                                                  const int32_t* __restrict__ claim, int32_t* __restrict__ last,
                                                                  int64_t n, const float* __restrict__ hist,
                                                                  const swr_adam_hyper* __restrict__ hp) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * AD_THREADS + threadIdx.x;
    const int64_t i = idx / dim;
    if (i >= n) return;
    const swr_adam_hyper h = *hp;
    const int now = static_cast<int>(h.step);
    const int f = from[i];
    if (f >= now) return;
    const uint32_t row = rows[i];
    if (claim[row] != static_cast<int32_t>(i)) return;
    const int e = static_cast<int>(idx - i * dim);
    const int64_t o = static_cast<int64_t>(row) * dim + e;
    float pi = p[o], mi = m[o], vi = v[o];
    adam_replay(pi, mi, vi, f, now, hist, h);
    p[o] = pi; m[o] = mi; v[o] = vi;
    if (e == 0) last[row] = now;             // nobody reads `last` in this launch
}

// phases 1 + 2 as ONE launch when dim is a power of two <= 64 (a row's dim lanes then sit inside one wavefront): the lane of
// column 0 elects the row's replayer with a compare-and-swap on `last[row]` itself (f -> now; exactly one of the duplicate
// lookups of a row that is behind sees its own f come back), and hands (row, f) to the other lanes by shuffle.  Which
// duplicate wins does not matter: the replay reads only the row's own (p, m, v) and the step scalars.  Nobody reads
// p / m / v in this launch, so marking the row current before it is replayed is safe; a second entry of the SAME table in
// one launch (a table shared by two features) simply loses the election.  One ~5 us launch less in front of every lookup.
__device__ __forceinline__ void adam_catchup_elect(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                   int32_t* __restrict__ last, int64_t vocab, int dim, const void* __restrict__ idx,
                                                   int idx_dtype, uint32_t hash_seed, int64_t i, int e,
                                                   const float* __restrict__ hist, const swr_adam_hyper& h) {
    const int now = static_cast<int>(h.step);
    int f = now;
    uint32_t row = 0u;
    if (e == 0) {
        int64_t id = swr_load_index(idx, idx_dtype, i);
        if (hash_seed != 0u) {
            uint64_t z = static_cast<uint64_t>(id) ^ hash_seed;
            z += 0x9E3779B97F4A7C15ull;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            id = static_cast<int64_t>((z ^ (z >> 31)) % static_cast<uint64_t>(vocab));
        }
        if (id >= 0 && id < vocab) {            // out-of-range ids: the gather flags them; nothing to replay
            row = static_cast<uint32_t>(id);
            f = __hip_atomic_load(last + id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (f < now && atomicCAS(last + id, f, now) != f) f = now;      // another lookup of the row replays it
        }
    }
    const int src = static_cast<int>(__lane_id()) - e;
    f = __shfl(f, src);
    row = static_cast<uint32_t>(__shfl(static_cast<int>(row), src));
    if (f >= now) return;
    const int64_t o = static_cast<int64_t>(row) * dim + e;
    float pi = p[o], mi = m[o], vi = v[o];
    adam_replay(pi, mi, vi, f, now, hist, h);
    p[o] = pi; m[o] = mi; v[o] = vi;
}

__global__ __launch_bounds__(AD_THREADS) void adam_catchup_fused_kernel(float* __restrict__ p, float* __restrict__ m,
                                                                        float* __restrict__ v, int32_t* __restrict__ last,
                                                                        int64_t vocab, int dim, const void* __restrict__ idx,
                                                                        int idx_dtype, uint32_t hash_seed, int64_t n,
                                                                        const float* __restrict__ hist,
                                                                        const swr_adam_hyper* __restrict__ hp) {
    const int64_t g = static_cast<int64_t>(blockIdx.x) * AD_THREADS + threadIdx.x;
    const int64_t i = g / dim;
    if (i >= n) return;                           // (whole lane groups: n * dim is a multiple of dim)
    const swr_adam_hyper h = *hp;
    adam_catchup_elect(p, m, v, last, vocab, dim, idx, idx_dtype, hash_seed, i, static_cast<int>(g - i * dim), hist, h);
}

static inline bool adam_dim_in_wave(int dim) { return dim > 0 && dim <= 64 && (dim & (dim - 1)) == 0; }

extern "C" int swr_adam_catchup_rows(float* p, float* m, float* v, int32_t* last, int32_t* claim, int64_t vocab, int dim,
                                     const void* idx, int idx_dtype, uint32_t hash_seed, int64_t n, const float* hist,
                                     const swr_adam_hyper* hyper, void* workspace, size_t workspace_bytes, void* stream) {
    SWR_REQUIRE(p && m && v && last && claim && idx && hist && hyper && workspace && vocab > 0 && dim > 0 && n >= 0, SWR_ERR_ARG);
    SWR_REQUIRE(swr_is_index_dtype(idx_dtype), SWR_ERR_DTYPE);
    SWR_REQUIRE(workspace_bytes >= static_cast<size_t>(n) * 8, SWR_ERR_WORKSPACE);
    if (n == 0) return SWR_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (adam_dim_in_wave(dim)) {
        hipLaunchKernelGGL(adam_catchup_fused_kernel, dim3(static_cast<unsigned>(swr_ceil_div(n * dim, AD_THREADS))), dim3(AD_THREADS),
                           0, st, p, m, v, last, vocab, dim, idx, idx_dtype, hash_seed, n, hist, hyper);
        return swr_launch_status();
    }
    int32_t* from = static_cast<int32_t*>(workspace);
    uint32_t* rows = reinterpret_cast<uint32_t*>(from + n);
    hipLaunchKernelGGL(adam_claim_kernel, dim3(static_cast<unsigned>(swr_ceil_div(n, AD_THREADS))), dim3(AD_THREADS), 0, st, idx,
                       idx_dtype, hash_seed, n, vocab, last, claim, from, rows, hyper);
    hipLaunchKernelGGL(adam_catchup_kernel, dim3(static_cast<unsigned>(swr_ceil_div(n * dim, AD_THREADS))), dim3(AD_THREADS), 0, st,
                       p, m, v, dim, from, rows, claim, last, n, hist, hyper);
    return swr_launch_status();
}

// materialise every row (checkpoint / hand-over to code that reads the table directly)
__global__ __launch_bounds__(AD_THREADS) void adam_flush_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                                int64_t vocab, int dim, const int32_t* __restrict__ last,
                                                                const float* __restrict__ hist,
                                                                const swr_adam_hyper* __restrict__ hp) {
    const swr_adam_hyper h = *hp;
    const int64_t n = vocab * dim;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * AD_THREADS;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * AD_THREADS + threadIdx.x; i < n; i += stride) {
        const int f = last[i / dim];
        if (f >= static_cast<int>(h.step)) continue;
        float pi = p[i], mi = m[i], vi = v[i];
        adam_replay(pi, mi, vi, f, static_cast<int>(h.step), hist, h);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

__global__ __launch_bounds__(AD_THREADS) void adam_mark_current_kernel(int32_t* __restrict__ last, int64_t vocab,
                                                                       const swr_adam_hyper* __restrict__ hp) {
    const int now = static_cast<int>(hp->step);
    const int64_t stride = static_cast<int64_t>(gridDim.x) * AD_THREADS;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * AD_THREADS + threadIdx.x; i < vocab; i += stride) last[i] = now;
}

extern "C" int swr_adam_flush(float* p, float* m, float* v, int32_t* last, int64_t vocab, int dim, const float* hist,
                              const swr_adam_hyper* hyper, void* stream) {
    SWR_REQUIRE(p && m && v && last && hist && hyper && vocab > 0 && dim > 0, SWR_ERR_ARG);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t n = vocab * dim;
    const unsigned grid = static_cast<unsigned>(swr_ceil_div(n, AD_THREADS) < 8192 ? swr_ceil_div(n, AD_THREADS) : 8192);
    hipLaunchKernelGGL(adam_flush_kernel, dim3(grid), dim3(AD_THREADS), 0, st, p, m, v, vocab, dim, last, hist, hyper);
    const unsigned g2 = static_cast<unsigned>(swr_ceil_div(vocab, AD_THREADS) < 4096 ? swr_ceil_div(vocab, AD_THREADS) : 4096);
    hipLaunchKernelGGL(adam_mark_current_kernel, dim3(g2), dim3(AD_THREADS), 0, st, last, vocab, hyper);
    return swr_launch_status();
}


// ------------------------------------------------------------------------------ several large tables, one launch each
// A model with T row-sparse tables (Ali-CCP: 10) paid T x (claim + catch-up) launches in front of every lookup and T
// row-update launches in every optimizer step -- 30 launches of ~5 us with nothing in them.  The kernels below take a
// descriptor per table and walk the concatenation of the tables' entries; per entry they do exactly what the
// single-table kernels do (same arithmetic, same ticket election per table).
struct AdamMultiK {
    swr_adam_table tab[SWR_ADAM_MAX_TABLES];
    int64_t first[SWR_ADAM_MAX_TABLES + 1];     // first work item of each table
    int n_tables;
    const float* hist;
    const swr_adam_hyper* hp;
};

__device__ __forceinline__ int adam_multi_table(const AdamMultiK& k, int64_t i) {
    int t = 0;
    while (t + 1 < k.n_tables && i >= k.first[t + 1]) ++t;
    return t;
}

__global__ __launch_bounds__(AD_THREADS) void adam_claim_multi_kernel(const AdamMultiK k) {
    const int64_t g = static_cast<int64_t>(blockIdx.x) * AD_THREADS + threadIdx.x;
    if (g >= k.first[k.n_tables]) return;
    const int t = adam_multi_table(k, g);
    const swr_adam_table& T = k.tab[t];
    const int64_t i = g - k.first[t];
    int64_t id = swr_load_index(T.idx, T.idx_dtype, i);
    if (T.hash_seed != 0u) {
        uint64_t z = static_cast<uint64_t>(id) ^ T.hash_seed;
        z += 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        id = static_cast<int64_t>((z ^ (z >> 31)) % static_cast<uint64_t>(T.vocab));
    }
    const int now = static_cast<int>(k.hp->step);
    int f = now;
    if (id >= 0 && id < T.vocab) {
        f = T.last[id];
        T.rows[i] = static_cast<uint32_t>(id);
        if (f < now) T.claim[id] = static_cast<int32_t>(i);
    }
    T.from_step[i] = f;
}

__global__ __launch_bounds__(AD_THREADS) void adam_catchup_multi_kernel(const AdamMultiK k) {
    const int64_t g = static_cast<int64_t>(blockIdx.x) * AD_THREADS + threadIdx.x;
    if (g >= k.first[k.n_tables]) return;
    const int t = adam_multi_table(k, g);
    const swr_adam_table& T = k.tab[t];
    const int64_t idx = g - k.first[t];
    const int64_t i = idx / T.dim;
    const swr_adam_hyper h = *k.hp;
    const int now = static_cast<int>(h.step);
    const int f = T.from_step[i];
    if (f >= now) return;
    const uint32_t row = T.rows[i];
    if (T.claim[row] != static_cast<int32_t>(i)) return;
    const int e = static_cast<int>(idx - i * T.dim);
    const int64_t o = static_cast<int64_t>(row) * T.dim + e;
    float pi = T.p[o], mi = T.m[o], vi = T.v[o];
    adam_replay(pi, mi, vi, f, now, k.hist, h);
    T.p[o] = pi; T.m[o] = mi; T.v[o] = vi;
    if (e == 0) T.last[row] = now;
}

// claim + catch-up of every table in one launch (see adam_catchup_fused_kernel); each table's work items start at a
// multiple of 64 so that no row's lanes straddle a wavefront
__global__ __launch_bounds__(AD_THREADS) void adam_catchup_fused_multi_kernel(const AdamMultiK k) {
    const int64_t g = static_cast<int64_t>(blockIdx.x) * AD_THREADS + threadIdx.x;
    if (g >= k.first[k.n_tables]) return;
    const int t = adam_multi_table(k, g);
    const swr_adam_table& T = k.tab[t];
    const int64_t idx = g - k.first[t];
    const int64_t i = idx / T.dim;
    if (i >= T.n) return;                         // alignment padding behind the table's last row
    const swr_adam_hyper h = *k.hp;
    adam_catchup_elect(T.p, T.m, T.v, T.last, T.vocab, T.dim, T.idx, T.idx_dtype, T.hash_seed, i,
                       static_cast<int>(idx - i * T.dim), k.hist, h);
}

__global__ __launch_bounds__(AD_THREADS) void adam_rows_multi_kernel(const AdamMultiK k) {
    const int64_t g = static_cast<int64_t>(blockIdx.x) * AD_THREADS + threadIdx.x;
    if (g >= k.first[k.n_tables]) return;
    const int t = adam_multi_table(k, g);
    const swr_adam_table& T = k.tab[t];
    const int64_t idx = g - k.first[t];
    const int64_t i = idx / T.dim;
    const int e = static_cast<int>(idx - i * T.dim);
    const int32_t row = T.urow[i];
    if (row < 0 || row >= T.vocab) return;
    const swr_adam_hyper h = *k.hp;
    const int64_t o = static_cast<int64_t>(row) * T.dim + e;
    float pi = T.p[o], mi = T.m[o], vi = T.v[o];
    adam_elem(pi, T.ugrad[i * T.dim + e], mi, vi, h);
    T.p[o] = pi; T.m[o] = mi; T.v[o] = vi;
    if (e == 0) T.last[row] = static_cast<int32_t>(h.step);
}

static int adam_multi_fill(AdamMultiK& k, const swr_adam_table* tables, int n_tables, bool per_elem, bool rows_mode,
                           const float* hist, const swr_adam_hyper* hyper) {
    SWR_REQUIRE(tables && n_tables > 0 && n_tables <= SWR_ADAM_MAX_TABLES && hyper, SWR_ERR_ARG);
    int64_t pos = 0;
    for (int t = 0; t < n_tables; ++t) {
        const swr_adam_table& T = tables[t];
        SWR_REQUIRE(T.p && T.m && T.v && T.last && T.vocab > 0 && T.dim > 0 && T.n >= 0, SWR_ERR_ARG);
        if (rows_mode) SWR_REQUIRE(T.urow && T.ugrad, SWR_ERR_ARG);
        else {
            SWR_REQUIRE(T.idx && T.claim && T.from_step && T.rows, SWR_ERR_ARG);
            SWR_REQUIRE(swr_is_index_dtype(T.idx_dtype), SWR_ERR_DTYPE);
        }
        k.tab[t] = T;
        k.first[t] = pos;
        pos += per_elem ? T.n * T.dim : T.n;
    }
    k.first[n_tables] = pos;
    k.n_tables = n_tables; k.hist = hist; k.hp = hyper;
    return SWR_OK;
}

extern "C" int swr_adam_catchup_multi(const swr_adam_table* tables, int n_tables, const float* hist,
                                      const swr_adam_hyper* hyper, void* stream) {
    SWR_REQUIRE(hist != nullptr, SWR_ERR_ARG);
    hipStream_t st = static_cast<hipStream_t>(stream);
    AdamMultiK k;
    int rc = adam_multi_fill(k, tables, n_tables, false, false, hist, hyper);
    if (rc != SWR_OK) return rc;
    if (k.first[n_tables] == 0) return SWR_OK;
    bool in_wave = true;
    for (int t = 0; t < n_tables; ++t) in_wave = in_wave && adam_dim_in_wave(tables[t].dim);
    if (in_wave) {
        int64_t pos = 0;
        for (int t = 0; t < n_tables; ++t) {
            k.first[t] = pos;
            pos += (tables[t].n * tables[t].dim + 63) / 64 * 64;
        }
        k.first[n_tables] = pos;
        hipLaunchKernelGGL(adam_catchup_fused_multi_kernel, dim3(static_cast<unsigned>(swr_ceil_div(pos, AD_THREADS))),
                           dim3(AD_THREADS), 0, st, k);
        return swr_launch_status();
    }
    hipLaunchKernelGGL(adam_claim_multi_kernel, dim3(static_cast<unsigned>(swr_ceil_div(k.first[n_tables], AD_THREADS))),
                       dim3(AD_THREADS), 0, st, k);
    rc = adam_multi_fill(k, tables, n_tables, true, false, hist, hyper);
    if (rc != SWR_OK) return rc;
    hipLaunchKernelGGL(adam_catchup_multi_kernel, dim3(static_cast<unsigned>(swr_ceil_div(k.first[n_tables], AD_THREADS))),
                       dim3(AD_THREADS), 0, st, k);
    return swr_launch_status();
}

extern "C" int swr_adam_rows_multi(const swr_adam_table* tables, int n_tables, const swr_adam_hyper* hyper, void* stream) {
    AdamMultiK k;
    int rc = adam_multi_fill(k, tables, n_tables, true, true, nullptr, hyper);
    if (rc != SWR_OK) return rc;
    if (k.first[n_tables] == 0) return SWR_OK;
    hipLaunchKernelGGL(adam_rows_multi_kernel, dim3(static_cast<unsigned>(swr_ceil_div(k.first[n_tables], AD_THREADS))),
                       dim3(AD_THREADS), 0, static_cast<hipStream_t>(stream), k);
    return swr_launch_status();
}
