// K3: backward of the embedding lookup for ALL features of a batch at once
// (replaces F_s x aten::embedding_dense_backward).
//
//   1. build_keys : composite key (table_id << row_bits | row) and payload (slot << 24 | sample)
//   2. sort       : device-wide LSD radix sort of the (key, payload) pairs on the used key bits only
//                   (rocPRIM's radix_sort_pairs -- the one library primitive on this path)
//   3. reduce     : each group of LPE lanes walks a fixed chunk of CHUNK sorted entries, sums runs of
//                   equal keys in registers and flushes each run with two 64-bit integer atomics
//   4. finalise   : integer accumulators -> fp32 gradients (dense tables) or per-row entries (sparse)
//
// Accumulation is dual-limb fixed point: x * 2^20 = hi + frac, hi in 2^-20 units, frac kept in 2^-60
// units.  Both limbs are integers, so the sum is exact (to 2^-60) and independent of the order in which
// runs, chunks and workgroups meet: bitwise deterministic, identical on every data-parallel rank.
// Representable range |x| < 2^20 (flagged otherwise); sums of up to 2^22 entries per row cannot overflow.
//
// HBM traffic per sample at dim E: keys 4 B + sorted (key, payload) 8 B + dE row 4E, plus 16E bytes of
// accumulator read-modify-write per DISTINCT (table, row) touched.
#include <algorithm>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"

#define MAX_SLOTS 48   // BwdMeta travels by value in the kernarg segment (4 KiB)
#define CHUNK 32
#define RB_THREADS 256

struct TableMeta {
    int64_t vocab;
    int64_t acc_off;     // dense: offset (in elements) into the dense accumulator region
    int64_t sorted_off;  // first sorted position of this table's entries
    float* grad_dense;
    int32_t* urow;
    float* ugrad;
    int32_t dim;
    int32_t mode;
};

struct BwdMeta {
    TableMeta tab[MAX_SLOTS];
    int32_t slot_col[MAX_SLOTS];
    int32_t slot_dim[MAX_SLOTS];
    int32_t slot_table[MAX_SLOTS];
    int32_t n_slots, n_tables;
    int32_t row_bits;
    int32_t dim_max;
    int64_t B, n;
    int64_t sparse_start;   // first sorted position belonging to a sparse-mode table
};

struct HostPlan {
    BwdMeta m;
    int64_t dense_acc_elems;
    int64_t sparse_acc_elems;
    size_t off_ck0, off_ck1, off_v0, off_v1, off_acc_hi, off_acc_lo, off_temp, temp_bytes, total;
    int key_bits;
};

static int bits_for(uint64_t max_value) {   // bits needed to represent values 0..max_value
    int b = 1;
    while ((max_value >> b) != 0) ++b;
    return b;
}
static size_t align_up(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

static int make_plan(const swr_embed_grad_slot* slots, int n_slots, int64_t B, bool query_temp, HostPlan& p) {
    SWR_REQUIRE(slots && n_slots > 0 && n_slots <= MAX_SLOTS && B > 0, SWR_ERR_ARG);
    SWR_REQUIRE(B <= (1 << 24), SWR_ERR_UNSUPPORTED);
    BwdMeta& m = p.m;
    m.n_slots = n_slots;
    m.B = B;
    m.n = static_cast<int64_t>(n_slots) * B;
    int n_tables = 0;
    int64_t max_vocab = 1;
    int dim_max = 1;
    for (int s = 0; s < n_slots; ++s) {
        SWR_REQUIRE(slots[s].table_id >= 0 && slots[s].table_id < MAX_SLOTS && slots[s].dim > 0 && slots[s].vocab > 0,
                    SWR_ERR_ARG);
        if (slots[s].table_id + 1 > n_tables) n_tables = slots[s].table_id + 1;
    }
    bool seen[MAX_SLOTS] = {false};
    int64_t count[MAX_SLOTS] = {0};
    for (int s = 0; s < n_slots; ++s) {
        const swr_embed_grad_slot& sl = slots[s];
        TableMeta& t = m.tab[sl.table_id];
        if (!seen[sl.table_id]) {
            seen[sl.table_id] = true;
            t.vocab = sl.vocab;
            t.dim = sl.dim;
            t.mode = sl.mode;
            t.grad_dense = sl.grad_dense;
            t.urow = sl.urow;
            t.ugrad = sl.ugrad;
            SWR_REQUIRE(sl.mode == 0 ? sl.grad_dense != nullptr : (sl.urow != nullptr && sl.ugrad != nullptr), SWR_ERR_ARG);
        } else {
            SWR_REQUIRE(t.vocab == sl.vocab && t.dim == sl.dim && t.mode == sl.mode, SWR_ERR_ARG);
        }
        count[sl.table_id] += B;
        m.slot_col[s] = sl.in_col;
        m.slot_dim[s] = sl.dim;
        m.slot_table[s] = sl.table_id;
        if (sl.vocab > max_vocab) max_vocab = sl.vocab;
        if (sl.dim > dim_max) dim_max = sl.dim;
    }
    int64_t acc = 0, pos = 0;
    m.sparse_start = -1;
    for (int t = 0; t < n_tables; ++t) {
        SWR_REQUIRE(seen[t], SWR_ERR_ARG);                       // table ids must be dense 0..n_tables-1
        TableMeta& tm = m.tab[t];
        tm.sorted_off = pos;
        if (tm.mode == 0) {
            SWR_REQUIRE(m.sparse_start < 0, SWR_ERR_ARG);        // sparse tables carry the largest ids
            tm.acc_off = acc;
            acc += tm.vocab * tm.dim;
        } else {
            if (m.sparse_start < 0) m.sparse_start = pos;
            tm.acc_off = 0;
        }
        pos += count[t];
    }
    if (m.sparse_start < 0) m.sparse_start = m.n;
    m.n_tables = n_tables;
    m.dim_max = dim_max;
    m.row_bits = bits_for(static_cast<uint64_t>(max_vocab - 1));
    p.key_bits = m.row_bits + (n_tables > 1 ? bits_for(static_cast<uint64_t>(n_tables - 1)) : 0);
    SWR_REQUIRE(p.key_bits <= 32, SWR_ERR_UNSUPPORTED);
    p.dense_acc_elems = acc;
    p.sparse_acc_elems = (m.n - m.sparse_start) * dim_max;

    p.temp_bytes = 0;
    if (query_temp) {
        uint32_t* nul = nullptr;
        if (rocprim::radix_sort_pairs(nullptr, p.temp_bytes, nul, nul, nul, nul, static_cast<size_t>(m.n), 0u,
                                      static_cast<unsigned>(p.key_bits), hipStream_t(0)) != hipSuccess)
            return SWR_ERR_LAUNCH;
    }
    size_t off = 0;
    const size_t kb = align_up(static_cast<size_t>(m.n) * 4);
    p.off_ck0 = off; off += kb;
    p.off_ck1 = off; off += kb;
    p.off_v0 = off; off += kb;
    p.off_v1 = off; off += kb;
    const size_t ab = align_up(static_cast<size_t>(p.dense_acc_elems + p.sparse_acc_elems) * 8);
    p.off_acc_hi = off; off += ab;
    p.off_acc_lo = off; off += ab;
    p.off_temp = off; off += align_up(p.temp_bytes);
    p.total = off;
    return SWR_OK;
}

__global__ __launch_bounds__(RB_THREADS) void build_keys_kernel(const BwdMeta m, const uint32_t* __restrict__ keys,
                                                                uint32_t* __restrict__ ck, uint32_t* __restrict__ val) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * RB_THREADS + threadIdx.x;
    if (i >= m.n) return;
    const int slot = static_cast<int>(i / m.B);
    const uint32_t b = static_cast<uint32_t>(i - static_cast<int64_t>(slot) * m.B);
    ck[i] = (m.row_bits >= 32 ? 0u : (static_cast<uint32_t>(m.slot_table[slot]) << m.row_bits)) | keys[i];
    val[i] = (static_cast<uint32_t>(slot) << 24) | b;
}

__device__ __forceinline__ void to_fixed(float x, long long& hi, long long& lo, uint32_t* err) {
    if (!(fabsf(x) < 1048576.f)) {            // also catches NaN / Inf
        if (err) atomicOr(err, SWR_FLAG_GRAD_RANGE);
        x = x > 0.f ? 1048575.f : (x < 0.f ? -1048575.f : 0.f);
    }
    const double xd = static_cast<double>(x) * 1048576.0;       // exact
    const double fl = floor(xd);
    hi = static_cast<long long>(fl);
    lo = static_cast<long long>(rint((xd - fl) * 1099511627776.0));   // frac * 2^40
}

__device__ __forceinline__ float from_fixed(long long hi, long long lo) {
    return static_cast<float>(static_cast<double>(hi) * (1.0 / 1048576.0) +
                              static_cast<double>(lo) * (1.0 / 1152921504606846976.0));
}

__device__ __forceinline__ int table_of(const BwdMeta& m, uint32_t key) {
    return m.row_bits >= 32 ? 0 : static_cast<int>(key >> m.row_bits);
}

// first position whose key is >= key
__device__ __forceinline__ int64_t lower_bound_key(const uint32_t* ck, int64_t n, uint32_t key) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (ck[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// LPE lanes per entry (one lane per gradient column; dims above 64 are walked in 64-column blocks)
template <int LPE>
__global__ __launch_bounds__(RB_THREADS) void reduce_kernel(const BwdMeta m, const uint32_t* __restrict__ ck,
                                                            const uint32_t* __restrict__ val,
                                                            const float* __restrict__ dE, int64_t ld,
                                                            unsigned long long* acc_hi, unsigned long long* acc_lo,
                                                            int64_t dense_acc_elems, uint32_t* err) {
    const int64_t gid = (static_cast<int64_t>(blockIdx.x) * RB_THREADS + threadIdx.x) / LPE;
    const int e0 = threadIdx.x % LPE;
    const int64_t i0 = gid * CHUNK;
    if (i0 >= m.n) return;
    const int64_t i1 = min(i0 + CHUNK, m.n);
    const uint32_t row_mask = (m.row_bits >= 32) ? 0xFFFFFFFFu : ((1u << m.row_bits) - 1u);

    for (int c0 = 0; c0 < m.dim_max; c0 += LPE) {
        const int e = c0 + e0;
        uint32_t cur = ck[i0];
        int64_t head = i0;
        if (i0 > 0 && ck[i0 - 1] == cur && m.tab[table_of(m, cur)].mode != 0) head = lower_bound_key(ck, m.n, cur);
        long long s_hi = 0, s_lo = 0;
        auto flush = [&](uint32_t key, int64_t head_pos) {
            const TableMeta& t = m.tab[table_of(m, key)];
            if (e >= t.dim) return;
            int64_t dst;
            if (t.mode == 0)
                dst = t.acc_off + static_cast<int64_t>(key & row_mask) * t.dim + e;
            else
                dst = dense_acc_elems + (head_pos - m.sparse_start) * m.dim_max + e;
            atomicAdd(acc_hi + dst, static_cast<unsigned long long>(s_hi));
            atomicAdd(acc_lo + dst, static_cast<unsigned long long>(s_lo));
        };
#pragma unroll 4
        for (int64_t i = i0; i < i1; ++i) {
            const uint32_t k = ck[i];
            const uint32_t v = val[i];
            if (k != cur) {
                flush(cur, head);
                cur = k;
                head = i;
                s_hi = 0;
                s_lo = 0;
            }
            const int slot = static_cast<int>(v >> 24);
            const int64_t b = v & 0xFFFFFFu;
            if (e < m.slot_dim[slot]) {
                long long h, l;
                to_fixed(dE[b * ld + m.slot_col[slot] + e], h, l, err);
                s_hi += h;
                s_lo += l;
            }
        }
        flush(cur, head);
    }
}

__global__ __launch_bounds__(RB_THREADS) void finalize_dense_kernel(const BwdMeta m, const long long* __restrict__ acc_hi,
                                                                    const long long* __restrict__ acc_lo) {
    const TableMeta& t = m.tab[blockIdx.y];
    if (t.mode != 0) return;
    const int64_t n = t.vocab * t.dim;
    for (int64_t j = static_cast<int64_t>(blockIdx.x) * RB_THREADS + threadIdx.x; j < n;
         j += static_cast<int64_t>(gridDim.x) * RB_THREADS)
        t.grad_dense[j] = from_fixed(acc_hi[t.acc_off + j], acc_lo[t.acc_off + j]);
}

__global__ __launch_bounds__(RB_THREADS) void finalize_sparse_kernel(const BwdMeta m, const uint32_t* __restrict__ ck,
                                                                     const long long* __restrict__ acc_hi,
                                                                     const long long* __restrict__ acc_lo,
                                                                     int64_t dense_acc_elems) {
    // one thread per (sorted entry of the sparse region, column)
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * RB_THREADS + threadIdx.x;
    const int64_t i = m.sparse_start + idx / m.dim_max;
    const int e = static_cast<int>(idx % m.dim_max);
    if (i >= m.n) return;
    const uint32_t key = ck[i];
    const TableMeta& t = m.tab[table_of(m, key)];
    if (e >= t.dim) return;
    const bool head = (i == 0) || (ck[i - 1] != key);
    const int64_t local = i - t.sorted_off;
    const uint32_t row_mask = (m.row_bits >= 32) ? 0xFFFFFFFFu : ((1u << m.row_bits) - 1u);
    if (e == 0) t.urow[local] = head ? static_cast<int32_t>(key & row_mask) : -1;
    const int64_t a = dense_acc_elems + (i - m.sparse_start) * m.dim_max + e;
    t.ugrad[local * t.dim + e] = head ? from_fixed(acc_hi[a], acc_lo[a]) : 0.f;
}

extern "C" size_t swr_embed_bwd_workspace_bytes(const swr_embed_grad_slot* slots, int n_slots, int64_t B) {
    HostPlan p;
    if (make_plan(slots, n_slots, B, true, p) != SWR_OK) return 0;
    return p.total;
}

extern "C" int swr_embed_bwd(const swr_embed_grad_slot* slots, int n_slots, const uint32_t* keys, const float* dE,
                             int64_t ld, int64_t B, void* workspace, size_t workspace_bytes, uint32_t* err_flag,
                             void* stream) {
    SWR_REQUIRE(keys && dE && workspace && ld > 0, SWR_ERR_ARG);
    if (B == 0) return SWR_OK;
    HostPlan p;
    int rc = make_plan(slots, n_slots, B, true, p);
    if (rc != SWR_OK) return rc;
    SWR_REQUIRE(workspace_bytes >= p.total, SWR_ERR_WORKSPACE);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    uint32_t* ck0 = reinterpret_cast<uint32_t*>(ws + p.off_ck0);
    uint32_t* ck1 = reinterpret_cast<uint32_t*>(ws + p.off_ck1);
    uint32_t* v0 = reinterpret_cast<uint32_t*>(ws + p.off_v0);
    uint32_t* v1 = reinterpret_cast<uint32_t*>(ws + p.off_v1);
    unsigned long long* acc_hi = reinterpret_cast<unsigned long long*>(ws + p.off_acc_hi);
    unsigned long long* acc_lo = reinterpret_cast<unsigned long long*>(ws + p.off_acc_lo);
    const BwdMeta& m = p.m;
    const int64_t n = m.n;

    hipLaunchKernelGGL(build_keys_kernel, dim3(static_cast<unsigned>(swr_ceil_div(n, RB_THREADS))), dim3(RB_THREADS), 0,
                       st, m, keys, ck0, v0);
    size_t temp = p.temp_bytes;
    if (rocprim::radix_sort_pairs(ws + p.off_temp, temp, ck0, ck1, v0, v1, static_cast<size_t>(n), 0u,
                                  static_cast<unsigned>(p.key_bits), st) != hipSuccess)
        return SWR_ERR_LAUNCH;
    // both accumulator limbs are contiguous: one memset
    if (hipMemsetAsync(acc_hi, 0, (p.off_acc_lo - p.off_acc_hi) * 2, st) != hipSuccess) return SWR_ERR_LAUNCH;

    int lpe = 1;
    while (lpe < m.dim_max && lpe < 64) lpe <<= 1;
    const int64_t groups = swr_ceil_div(n, CHUNK);
    const dim3 grid(static_cast<unsigned>(swr_ceil_div(groups * lpe, RB_THREADS)));
#define LAUNCH_REDUCE(L)                                                                                              \
    hipLaunchKernelGGL(reduce_kernel<L>, grid, dim3(RB_THREADS), 0, st, m, ck1, v1, dE, ld, acc_hi, acc_lo,            \
                       p.dense_acc_elems, err_flag)
    switch (lpe) {
        case 1: LAUNCH_REDUCE(1); break;
        case 2: LAUNCH_REDUCE(2); break;
        case 4: LAUNCH_REDUCE(4); break;
        case 8: LAUNCH_REDUCE(8); break;
        case 16: LAUNCH_REDUCE(16); break;
        case 32: LAUNCH_REDUCE(32); break;
        default: LAUNCH_REDUCE(64); break;
    }
#undef LAUNCH_REDUCE
    if (p.dense_acc_elems > 0) {
        int64_t biggest = 1;
        for (int t = 0; t < m.n_tables; ++t)
            if (m.tab[t].mode == 0 && m.tab[t].vocab * m.tab[t].dim > biggest) biggest = m.tab[t].vocab * m.tab[t].dim;
        const unsigned gx = static_cast<unsigned>(std::min<int64_t>(swr_ceil_div(biggest, RB_THREADS), 1024));
        hipLaunchKernelGGL(finalize_dense_kernel, dim3(gx, static_cast<unsigned>(m.n_tables)), dim3(RB_THREADS), 0, st, m,
                           reinterpret_cast<const long long*>(acc_hi), reinterpret_cast<const long long*>(acc_lo));
    }
    if (m.sparse_start < n) {
        const int64_t work = (n - m.sparse_start) * m.dim_max;
        hipLaunchKernelGGL(finalize_sparse_kernel, dim3(static_cast<unsigned>(swr_ceil_div(work, RB_THREADS))),
                           dim3(RB_THREADS), 0, st, m, ck1, reinterpret_cast<const long long*>(acc_hi),
                           reinterpret_cast<const long long*>(acc_lo), p.dense_acc_elems);
    }
    return swr_launch_status();
}
