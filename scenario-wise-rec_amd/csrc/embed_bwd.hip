// K3: backward of the embedding lookup for ALL features of a batch at once
// (replaces F_s x aten::embedding_dense_backward).
//
//   1. build_keys : entries (row, slot << 24 | sample) written grouped by table (one segment per table)
//   2. sort       : segmented LSD radix sort of every table segment by row, 8 bits per pass, stable; pass p
//                   only does work for tables whose row ids need more than 8p bits (others copy through).
//                   Hand-written (histogram -> per-table scan -> ranked scatter), no atomics on global
//                   memory, no memset nodes, no inter-workgroup spinning: safe to capture and replay.
//   3. reduce     : each group of LPE lanes walks a fixed chunk of CHUNK sorted entries of one table, sums
//                   runs of equal rows in registers and flushes each run (plain stores for runs that lie
//                   inside the chunk, two 64-bit integer atomics for runs cut by a chunk boundary)
//   4. finalise   : integer accumulators -> fp32 gradients (dense tables) or per-row entries (sparse)
//
// Accumulation is dual-limb fixed point: x * 2^20 = hi + frac, hi in 2^-20 units, frac kept in 2^-60
// units.  Both limbs are integers, so the sum is exact (to 2^-60) and independent of the order in which
// runs, chunks and workgroups meet: bitwise deterministic, identical on every data-parallel rank.
// Representable range |x| < 2^20 (flagged otherwise); sums of up to 2^22 entries per row cannot overflow.
//
// HBM traffic per sample at dim E: keys 4 B + (8 B read + 8 B write) per sort pass + dE row 4E, plus 16E
// bytes of accumulator traffic per DISTINCT (table, row) touched.
#include <algorithm>
#include <cstring>

#include "common.h"

#define MAX_SLOTS 40   // BwdMeta travels by value in the kernarg segment (4 KiB)
#define CHUNK 32
#define RB_THREADS 256
#define ACC_STRIPES 16   // copies of the dense accumulators: chunk c adds into stripe c % 16, so a hot row of a tiny
                       // table (V = 2: 1000+ partial runs per row) does not serialise its atomics on one address
#define SORT_THREADS 256
#define SORT_ITEMS 8
#define SORT_TILE (SORT_THREADS * SORT_ITEMS)

struct TableMeta {
    int64_t vocab;
    int64_t acc_off;     // dense: offset (in elements) into the dense accumulator region
    int64_t sorted_off;  // first sorted position of this table's segment
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
    int64_t slot_dst[MAX_SLOTS];     // where this slot's B entries start inside its table segment
    int32_t chunk_off[MAX_SLOTS + 1];   // first reduce chunk of each table
    int32_t n_slots, n_tables;
    int32_t dim_max;
    int32_t pad;
    int64_t B, n;
    int64_t sparse_start;   // first sorted position belonging to a sparse-mode table
};

struct SortMeta {
    int64_t seg_off[MAX_SLOTS + 1];
    int32_t tile_off[MAX_SLOTS + 1];
    int32_t passes[MAX_SLOTS];
    int32_t n_tables;
    int32_t n_tiles;
};

struct HostPlan {
    BwdMeta m;
    SortMeta sm;
    int64_t dense_acc_elems;
    int64_t sparse_acc_elems;
    size_t off_k0, off_k1, off_v0, off_v1, off_hist, off_acc_hi, off_acc_lo, total;
    int n_passes;
    int n_chunks;
};

static int bits_for(uint64_t max_value) {   // bits needed to represent values 0..max_value
    int b = 0;
    while ((max_value >> b) != 0) ++b;
    return b;
}
static size_t align_up(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

static int make_plan(const swr_embed_grad_slot* slots, int n_slots, int64_t B, HostPlan& p) {
    SWR_REQUIRE(slots && n_slots > 0 && n_slots <= MAX_SLOTS && B > 0, SWR_ERR_ARG);
    SWR_REQUIRE(B <= (1 << 24), SWR_ERR_UNSUPPORTED);
    BwdMeta& m = p.m;
    m.n_slots = n_slots;
    m.B = B;
    m.n = static_cast<int64_t>(n_slots) * B;
    int n_tables = 0;
    int dim_max = 1;
    for (int s = 0; s < n_slots; ++s) {
        SWR_REQUIRE(slots[s].table_id >= 0 && slots[s].table_id < MAX_SLOTS && slots[s].dim > 0 && slots[s].vocab > 0,
                    SWR_ERR_ARG);
        SWR_REQUIRE(slots[s].vocab <= 0xFFFFFFFFll + 1, SWR_ERR_UNSUPPORTED);
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
            SWR_REQUIRE(sl.mode >= 0 && sl.mode <= 2, SWR_ERR_ARG);
            SWR_REQUIRE(sl.mode != 1 ? sl.grad_dense != nullptr : (sl.urow != nullptr && sl.ugrad != nullptr), SWR_ERR_ARG);
        } else {
            SWR_REQUIRE(t.vocab == sl.vocab && t.dim == sl.dim && t.mode == sl.mode, SWR_ERR_ARG);
        }
        m.slot_dst[s] = count[sl.table_id];          // offset inside the table segment (segment base added below)
        count[sl.table_id] += B;
        m.slot_col[s] = sl.in_col;
        m.slot_dim[s] = sl.dim;
        if (sl.dim > dim_max) dim_max = sl.dim;
    }
    int64_t acc = 0, pos = 0;
    int tiles = 0, chunks = 0, passes = 0;
    m.sparse_start = -1;
    for (int t = 0; t < n_tables; ++t) {
        SWR_REQUIRE(seen[t], SWR_ERR_ARG);                       // table ids must be dense 0..n_tables-1
        TableMeta& tm = m.tab[t];
        tm.sorted_off = pos;
        if (tm.mode != 1) {
            SWR_REQUIRE(m.sparse_start < 0, SWR_ERR_ARG);        // sparse tables carry the largest ids
            tm.acc_off = acc;
            acc += tm.vocab * tm.dim;
        } else {
            if (m.sparse_start < 0) m.sparse_start = pos;
            tm.acc_off = 0;
        }
        p.sm.seg_off[t] = pos;
        p.sm.tile_off[t] = tiles;
        p.sm.passes[t] = (bits_for(static_cast<uint64_t>(tm.vocab - 1)) + 7) / 8;
        if (p.sm.passes[t] > passes) passes = p.sm.passes[t];
        m.chunk_off[t] = chunks;
        tiles += static_cast<int>(swr_ceil_div(count[t], SORT_TILE));
        chunks += static_cast<int>(swr_ceil_div(count[t], CHUNK));
        pos += count[t];
    }
    for (int s = 0; s < n_slots; ++s) m.slot_dst[s] += m.tab[slots[s].table_id].sorted_off;
    p.sm.seg_off[n_tables] = pos;
    p.sm.tile_off[n_tables] = tiles;
    p.sm.n_tables = n_tables;
    p.sm.n_tiles = tiles;
    m.chunk_off[n_tables] = chunks;
    if (m.sparse_start < 0) m.sparse_start = m.n;
    m.n_tables = n_tables;
    m.dim_max = dim_max;
    p.n_passes = passes;
    p.n_chunks = chunks;
    p.dense_acc_elems = acc;
    p.sparse_acc_elems = (m.n - m.sparse_start) * dim_max;

    size_t off = 0;
    const size_t kb = align_up(static_cast<size_t>(m.n) * 4);
    p.off_k0 = off; off += kb;
    p.off_k1 = off; off += kb;
    p.off_v0 = off; off += kb;
    p.off_v1 = off; off += kb;
    p.off_hist = off; off += align_up(static_cast<size_t>(tiles) * 256 * 4);
    const size_t ab = align_up(static_cast<size_t>(p.dense_acc_elems * ACC_STRIPES + p.sparse_acc_elems) * 8);
    p.off_acc_hi = off; off += ab;
    p.off_acc_lo = off; off += ab;
    p.total = off;
    return SWR_OK;
}

__global__ __launch_bounds__(RB_THREADS) void build_keys_kernel(const BwdMeta m, const uint32_t* __restrict__ keys,
                                                                uint32_t* __restrict__ ck, uint32_t* __restrict__ val) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * RB_THREADS + threadIdx.x;
    if (i >= m.n) return;
    const int slot = static_cast<int>(i / m.B);
    const int64_t b = i - static_cast<int64_t>(slot) * m.B;
    const int64_t dst = m.slot_dst[slot] + b;
    ck[dst] = keys[i];
    val[dst] = (static_cast<uint32_t>(slot) << 24) | static_cast<uint32_t>(b);
}

// ------------------------------------------------------------------------------- segmented radix sort
__device__ __forceinline__ int sort_table_of_tile(const SortMeta& sm, int tile) {
    int t = 0;
    while (t + 1 < sm.n_tables && sm.tile_off[t + 1] <= tile) ++t;
    return t;
}

// per-tile histogram of the pass's digit
__global__ __launch_bounds__(SORT_THREADS) void sort_hist_kernel(const SortMeta sm, int pass, const uint32_t* __restrict__ kin,
                                                                 uint32_t* __restrict__ hist) {
    __shared__ uint32_t lh[256];
    const int tile = blockIdx.x;
    const int t = sort_table_of_tile(sm, tile);
    if (pass >= sm.passes[t]) return;
    const int64_t start = sm.seg_off[t] + static_cast<int64_t>(tile - sm.tile_off[t]) * SORT_TILE;
    const int len = static_cast<int>(min<int64_t>(SORT_TILE, sm.seg_off[t + 1] - start));
    lh[threadIdx.x] = 0;
    __syncthreads();
    for (int e = threadIdx.x; e < len; e += SORT_THREADS) atomicAdd(&lh[(kin[start + e] >> (8 * pass)) & 255u], 1u);
    __syncthreads();
    hist[static_cast<int64_t>(tile) * 256 + threadIdx.x] = lh[threadIdx.x];
}

// one workgroup per table: hist[tile][digit] -> global output offset of (tile, digit).  Thread = digit for the digit
// totals; the running offsets over the table's tiles are produced by 4 tile-strided passes per digit quarter so that
// 1024 threads share the work of long tables (the big table has 32+ tiles, shared small tables 64).
#define SCAN_THREADS 1024
__global__ __launch_bounds__(SCAN_THREADS) void sort_scan_kernel(const SortMeta sm, int pass, uint32_t* __restrict__ hist) {
    __shared__ uint32_t tot[256];
    __shared__ uint32_t partial[4][256];
    const int t = blockIdx.x;
    if (pass >= sm.passes[t]) return;
    const int t0 = sm.tile_off[t], t1 = sm.tile_off[t + 1];
    const int nt = t1 - t0;
    const int d = threadIdx.x & 255, q = threadIdx.x >> 8;          // digit, tile quarter
    const int per = (nt + 3) / 4;
    const int qa = t0 + min(q * per, nt), qb = t0 + min((q + 1) * per, nt);
    // pass 1: per-quarter totals of this digit
    uint32_t run = 0;
    for (int tile = qa; tile < qb; ++tile) run += hist[static_cast<int64_t>(tile) * 256 + d];
    partial[q][d] = run;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < q) before += partial[k][d];
        total += partial[k][d];
    }
    // exclusive scan of the 256 digit totals (Hillis-Steele in LDS), done by quarter 0, read by all
    if (q == 0) tot[d] = total;
    __syncthreads();
    uint32_t v = total;
    for (int off = 1; off < 256; off <<= 1) {
        const uint32_t add = (q == 0 && d >= off) ? tot[d - off] : 0u;
        __syncthreads();
        v += add;
        if (q == 0) tot[d] = v;
        __syncthreads();
    }
    const uint32_t digit_base = static_cast<uint32_t>(sm.seg_off[t]) + (tot[d] - total);
    // pass 2: exclusive running offsets inside the quarter
    uint32_t off = digit_base + before;
    for (int tile = qa; tile < qb; ++tile) {
        const uint32_t c = hist[static_cast<int64_t>(tile) * 256 + d];
        hist[static_cast<int64_t>(tile) * 256 + d] = off;
        off += c;
    }
}

// stable scatter: element order inside a tile is round-major, then wave, then lane (= memory order)
__global__ __launch_bounds__(SORT_THREADS) void sort_scatter_kernel(const SortMeta sm, int pass, const uint32_t* __restrict__ kin,
                                                                    const uint32_t* __restrict__ vin,
                                                                    uint32_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                                    const uint32_t* __restrict__ hist) {
    __shared__ uint32_t off[256];
    __shared__ uint32_t wc[SORT_THREADS / 64][256];
    const int tile = blockIdx.x;
    const int t = sort_table_of_tile(sm, tile);
    const int64_t start = sm.seg_off[t] + static_cast<int64_t>(tile - sm.tile_off[t]) * SORT_TILE;
    const int len = static_cast<int>(min<int64_t>(SORT_TILE, sm.seg_off[t + 1] - start));
    if (pass >= sm.passes[t]) {          // this table is already sorted: carry it to the other buffer
        for (int e = threadIdx.x; e < len; e += SORT_THREADS) {
            kout[start + e] = kin[start + e];
            vout[start + e] = vin[start + e];
        }
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    off[threadIdx.x] = hist[static_cast<int64_t>(tile) * 256 + threadIdx.x];
#pragma unroll
    for (int w = 0; w < SORT_THREADS / 64; ++w) wc[w][threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const int e = r * SORT_THREADS + threadIdx.x;
        const bool valid = e < len;
        uint32_t key = 0, val = 0;
        if (valid) {
            key = kin[start + e];
            val = vin[start + e];
        }
        const uint32_t d = (key >> (8 * pass)) & 255u;
        unsigned long long mask = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const bool b = (d >> bit) & 1u;
            const unsigned long long mb = __ballot(valid && b);
            mask &= b ? mb : ~mb;
        }
        const int rank = __popcll(mask & lt_mask);
        if (valid && rank == 0) wc[wave][d] = static_cast<uint32_t>(__popcll(mask));
        __syncthreads();
        if (valid) {
            uint32_t pre = off[d];
            for (int w = 0; w < wave; ++w) pre += wc[w][d];
            kout[pre + rank] = key;
            vout[pre + rank] = val;
        }
        __syncthreads();
        uint32_t add = 0;
#pragma unroll
        for (int w = 0; w < SORT_THREADS / 64; ++w) {
            add += wc[w][threadIdx.x];
            wc[w][threadIdx.x] = 0;
        }
        off[threadIdx.x] += add;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ reduce
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

// first position in [lo, hi) whose key is >= key
__device__ __forceinline__ int64_t lower_bound_key(const uint32_t* ck, int64_t lo, int64_t hi, uint32_t key) {
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (ck[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// LPE lanes per entry (one lane per gradient column; dims above 64 are walked in 64-column blocks).
// A chunk never crosses a table segment.
template <int LPE>
__global__ __launch_bounds__(RB_THREADS) void reduce_kernel(const BwdMeta m, int n_chunks, const uint32_t* __restrict__ ck,
                                                            const uint32_t* __restrict__ val,
                                                            const float* __restrict__ dE, int64_t ld,
                                                            unsigned long long* acc_hi, unsigned long long* acc_lo,
                                                            int64_t dense_acc_elems, uint32_t* err) {
    const int64_t gid = (static_cast<int64_t>(blockIdx.x) * RB_THREADS + threadIdx.x) / LPE;
    const int e0 = threadIdx.x % LPE;
    if (gid >= n_chunks) return;
    int ti = 0;
    while (ti + 1 < m.n_tables && m.chunk_off[ti + 1] <= gid) ++ti;
    const TableMeta& t = m.tab[ti];
    const int64_t seg0 = t.sorted_off;
    const int64_t seg1 = (ti + 1 < m.n_tables) ? m.tab[ti + 1].sorted_off : m.n;
    const int64_t i0 = seg0 + (gid - m.chunk_off[ti]) * CHUNK;
    const int64_t i1 = min(i0 + CHUNK, seg1);

    // entries are taken in batches of BATCH: the BATCH key / payload loads, then the BATCH gradient loads are
    // independent and in flight together; only then does the run logic walk them (no per-entry load chain)
    constexpr int BATCH = 8;
    const int cnt = static_cast<int>(i1 - i0);
    const uint32_t prev_key = (i0 > seg0) ? ck[i0 - 1] : 0u;
    const uint32_t next_key = (i1 < seg1) ? ck[i1] : 0u;
    for (int c0 = 0; c0 < t.dim; c0 += LPE) {
        const int e = c0 + e0;
        uint32_t cur = ck[i0];
        int64_t head = i0;
        // a run that starts and ends inside this chunk is the only contributor to its row: plain stores;
        // only runs cut by a chunk boundary (at most two per chunk) need atomics
        bool whole_start = (i0 == seg0) || (prev_key != cur);
        if (!whole_start && t.mode == 1) head = lower_bound_key(ck, seg0, i0, cur);
        long long s_hi = 0, s_lo = 0;
        auto flush = [&](uint32_t key, int64_t head_pos, bool whole) {
            if (e >= t.dim) return;
            int64_t dst;
            if (t.mode != 1)
                dst = (gid % ACC_STRIPES) * dense_acc_elems + t.acc_off + static_cast<int64_t>(key) * t.dim + e;
            else
                dst = ACC_STRIPES * dense_acc_elems + (head_pos - m.sparse_start) * m.dim_max + e;
            if (whole) {
                acc_hi[dst] = static_cast<unsigned long long>(s_hi);
                acc_lo[dst] = static_cast<unsigned long long>(s_lo);
            } else {
                atomicAdd(acc_hi + dst, static_cast<unsigned long long>(s_hi));
                atomicAdd(acc_lo + dst, static_cast<unsigned long long>(s_lo));
            }
        };
        for (int j0 = 0; j0 < cnt; j0 += BATCH) {
            uint32_t ks[BATCH], vs[BATCH];
            float x[BATCH];
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                const int64_t i = i0 + min(j0 + j, cnt - 1);
                ks[j] = ck[i];
                vs[j] = val[i];
            }
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                const int slot = static_cast<int>(vs[j] >> 24);
                const int64_t b = vs[j] & 0xFFFFFFu;
                x[j] = e < t.dim ? dE[b * ld + m.slot_col[slot] + e] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                if (j0 + j < cnt) {
                    if (ks[j] != cur) {
                        flush(cur, head, whole_start);
                        cur = ks[j];
                        head = i0 + j0 + j;
                        whole_start = true;
                        s_hi = 0;
                        s_lo = 0;
                    }
                    long long h, l;
                    to_fixed(x[j], h, l, err);
                    s_hi += h;
                    s_lo += l;
                }
            }
        }
        flush(cur, head, whole_start && ((i1 == seg1) || (next_key != cur)));
    }
}

__global__ __launch_bounds__(RB_THREADS) void finalize_dense_kernel(const BwdMeta m, const long long* __restrict__ acc_hi,
                                                                    const long long* __restrict__ acc_lo,
                                                                    int64_t dense_acc_elems) {
    const TableMeta& t = m.tab[blockIdx.y];
    if (t.mode == 1) return;
    const int64_t n = t.vocab * t.dim;
    for (int64_t j = static_cast<int64_t>(blockIdx.x) * RB_THREADS + threadIdx.x; j < n;
         j += static_cast<int64_t>(gridDim.x) * RB_THREADS) {
        long long hi = 0, lo = 0;                                     // integer sums: order-free, exact
#pragma unroll
        for (int st = 0; st < ACC_STRIPES; ++st) {
            hi += acc_hi[st * dense_acc_elems + t.acc_off + j];
            lo += acc_lo[st * dense_acc_elems + t.acc_off + j];
        }
        const float g = from_fixed(hi, lo);
        t.grad_dense[j] = t.mode == 2 ? t.grad_dense[j] + g : g;     // mode 2: add to the caller's gradient arena
    }
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
    int ti = m.n_tables - 1;
    while (ti > 0 && m.tab[ti].sorted_off > i) --ti;
    const TableMeta& t = m.tab[ti];
    if (e >= t.dim) return;
    const uint32_t key = ck[i];
    const bool head = (i == t.sorted_off) || (ck[i - 1] != key);
    const int64_t local = i - t.sorted_off;
    if (e == 0) t.urow[local] = head ? static_cast<int32_t>(key) : -1;
    const int64_t a = ACC_STRIPES * dense_acc_elems + (i - m.sparse_start) * m.dim_max + e;
    t.ugrad[local * t.dim + e] = head ? from_fixed(acc_hi[a], acc_lo[a]) : 0.f;
}

extern "C" size_t swr_embed_bwd_workspace_bytes(const swr_embed_grad_slot* slots, int n_slots, int64_t B) {
    HostPlan p;
    if (make_plan(slots, n_slots, B, p) != SWR_OK) return 0;
    return p.total;
}

extern "C" int swr_embed_bwd(const swr_embed_grad_slot* slots, int n_slots, const uint32_t* keys, const float* dE,
                             int64_t ld, int64_t B, void* workspace, size_t workspace_bytes, uint32_t* err_flag,
                             void* stream) {
    SWR_REQUIRE(keys && dE && workspace && ld > 0, SWR_ERR_ARG);
    if (B == 0) return SWR_OK;
    HostPlan p;
    int rc = make_plan(slots, n_slots, B, p);
    if (rc != SWR_OK) return rc;
    SWR_REQUIRE(workspace_bytes >= p.total, SWR_ERR_WORKSPACE);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    uint32_t* kbuf[2] = {reinterpret_cast<uint32_t*>(ws + p.off_k0), reinterpret_cast<uint32_t*>(ws + p.off_k1)};
    uint32_t* vbuf[2] = {reinterpret_cast<uint32_t*>(ws + p.off_v0), reinterpret_cast<uint32_t*>(ws + p.off_v1)};
    uint32_t* hist = reinterpret_cast<uint32_t*>(ws + p.off_hist);
    unsigned long long* acc_hi = reinterpret_cast<unsigned long long*>(ws + p.off_acc_hi);
    unsigned long long* acc_lo = reinterpret_cast<unsigned long long*>(ws + p.off_acc_lo);
    const BwdMeta& m = p.m;
    const int64_t n = m.n;

    hipLaunchKernelGGL(build_keys_kernel, dim3(static_cast<unsigned>(swr_ceil_div(n, RB_THREADS))), dim3(RB_THREADS), 0,
                       st, m, keys, kbuf[0], vbuf[0]);
    int cur = 0;
    for (int pass = 0; pass < p.n_passes; ++pass) {
        hipLaunchKernelGGL(sort_hist_kernel, dim3(p.sm.n_tiles), dim3(SORT_THREADS), 0, st, p.sm, pass, kbuf[cur], hist);
        hipLaunchKernelGGL(sort_scan_kernel, dim3(p.sm.n_tables), dim3(SCAN_THREADS), 0, st, p.sm, pass, hist);
        hipLaunchKernelGGL(sort_scatter_kernel, dim3(p.sm.n_tiles), dim3(SORT_THREADS), 0, st, p.sm, pass, kbuf[cur], vbuf[cur],
                           kbuf[cur ^ 1], vbuf[cur ^ 1], hist);
        cur ^= 1;
    }
    const uint32_t* ck = kbuf[cur];
    const uint32_t* sv = vbuf[cur];
    // both accumulator limbs are contiguous: one zero-fill (a kernel, not a memset node)
    rc = swr_zero_async(acc_hi, (p.off_acc_lo - p.off_acc_hi) * 2, st);
    if (rc != SWR_OK) return rc;

    int lpe = 1;
    while (lpe < m.dim_max && lpe < 64) lpe <<= 1;
    const dim3 grid(static_cast<unsigned>(swr_ceil_div(static_cast<int64_t>(p.n_chunks) * lpe, RB_THREADS)));
#define LAUNCH_REDUCE(L)                                                                                              \
    hipLaunchKernelGGL(reduce_kernel<L>, grid, dim3(RB_THREADS), 0, st, m, p.n_chunks, ck, sv, dE, ld, acc_hi, acc_lo,   \
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
            if (m.tab[t].mode != 1 && m.tab[t].vocab * m.tab[t].dim > biggest) biggest = m.tab[t].vocab * m.tab[t].dim;
        const unsigned gx = static_cast<unsigned>(std::min<int64_t>(swr_ceil_div(biggest, RB_THREADS), 1024));
        hipLaunchKernelGGL(finalize_dense_kernel, dim3(gx, static_cast<unsigned>(m.n_tables)), dim3(RB_THREADS), 0, st, m,
                           reinterpret_cast<const long long*>(acc_hi), reinterpret_cast<const long long*>(acc_lo),
                           p.dense_acc_elems);
    }
    if (m.sparse_start < n) {
        const int64_t work = (n - m.sparse_start) * m.dim_max;
        hipLaunchKernelGGL(finalize_sparse_kernel, dim3(static_cast<unsigned>(swr_ceil_div(work, RB_THREADS))),
                           dim3(RB_THREADS), 0, st, m, ck, reinterpret_cast<const long long*>(acc_hi),
                           reinterpret_cast<const long long*>(acc_lo), p.dense_acc_elems);
    }
    return swr_launch_status();
}
