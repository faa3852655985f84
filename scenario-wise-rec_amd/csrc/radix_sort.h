// Segmented LSD radix sort of (uint32 key, uint32 value) pairs, 8 bits per pass, stable; shared by the embedding
// backward (entries of the large tables grouped by row, one segment per table) and the evaluation metrics (scores).
// Own kernels, no library primitives: no memset nodes, no look-back spinning -- safe under back-to-back hipGraph
// replays.  Per pass: sort_hist (per-tile digit histogram) -> sort_scan (one workgroup per
// segment: global offsets of every (tile, digit)) -> sort_scatter (stable, wave-ballot ranking).
#pragma once
#include "common.h"

#define MAX_SLOTS 40   // BwdMeta travels by value in the kernarg segment (4 KiB)
#define SORT_THREADS 256
#define SORT_ITEMS_MAX 8   // keys per thread and tile; fewer when there are few keys (more, smaller tiles fill the chip)

struct SortMeta {
    int64_t seg_off[MAX_SLOTS + 1];
    int32_t tile_off[MAX_SLOTS + 1];
    int32_t passes[MAX_SLOTS];
    int32_t n_tables;
    int32_t n_tiles;
    int32_t items;        // keys per thread of a tile
    int32_t tile;         // SORT_THREADS * items
    int32_t fold_scan;    // no table has more than SORT_FOLD_TILES tiles: sort_scatter computes its offsets itself, no sort_scan launch
};
#define SORT_FOLD_TILES 64

// keys per thread and tile, and whether the scan launch folds into the scatter: the smallest tile that leaves the largest table
// <= SORT_FOLD_TILES tiles (a scatter workgroup then sums <= 64 histograms per digit itself: one launch and one kernel boundary
// less per pass -- the sort sits ON the critical path of the single-stream steps, ops.SIDE_MIN_BATCH); tables too long for that
// keep the three-launch pass with the tile size that fills the chip
static inline void sort_choose_tile(SortMeta& sm, int64_t n_total, int64_t max_table_keys) {
    for (int items = 1; items <= SORT_ITEMS_MAX; items <<= 1)
        if ((max_table_keys + SORT_THREADS * items - 1) / (SORT_THREADS * items) <= SORT_FOLD_TILES) {
            sm.items = items; sm.tile = SORT_THREADS * items; sm.fold_scan = 1;
            return;
        }
    int items = SORT_ITEMS_MAX;
    while (items > 1 && n_total / (SORT_THREADS * items) < 512) items >>= 1;
    sm.items = items; sm.tile = SORT_THREADS * items; sm.fold_scan = 0;
}

// ------------------------------------------------------------------------------- segmented radix sort
__device__ __forceinline__ int sort_table_of_tile(const SortMeta& sm, int tile) {
    int t = 0;
    while (t + 1 < sm.n_tables && sm.tile_off[t + 1] <= tile) ++t;
    return t;
}

// per-tile histogram of the pass's digit
static __global__ __launch_bounds__(SORT_THREADS) void sort_hist_kernel(const SortMeta sm, int pass, const uint32_t* __restrict__ kin,
                                                                 uint32_t* __restrict__ hist) {
    __shared__ uint32_t lh[256];
    const int tile = blockIdx.x;
    const int t = sort_table_of_tile(sm, tile);
    if (pass >= sm.passes[t]) return;
    const int64_t start = sm.seg_off[t] + static_cast<int64_t>(tile - sm.tile_off[t]) * sm.tile;
    const int len = static_cast<int>(min<int64_t>(sm.tile, sm.seg_off[t + 1] - start));
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
static __global__ __launch_bounds__(SCAN_THREADS) void sort_scan_kernel(const SortMeta sm, int pass, uint32_t* __restrict__ hist) {
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
    {
        int tile = qa;
        for (; tile + 8 <= qb; tile += 8) {              // 8 independent loads in flight, then the adds
            uint32_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = hist[static_cast<int64_t>(tile + k) * 256 + d];
#pragma unroll
            for (int k = 0; k < 8; ++k) run += v[k];
        }
        for (; tile < qb; ++tile) run += hist[static_cast<int64_t>(tile) * 256 + d];
    }
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
    int tile = qa;
    for (; tile + 8 <= qb; tile += 8) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = hist[static_cast<int64_t>(tile + k) * 256 + d];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            hist[static_cast<int64_t>(tile + k) * 256 + d] = off;
            off += v[k];
        }
    }
    for (; tile < qb; ++tile) {
        const uint32_t c = hist[static_cast<int64_t>(tile) * 256 + d];
        hist[static_cast<int64_t>(tile) * 256 + d] = off;
        off += c;
    }
}

// stable scatter: element order inside a tile is round-major, then wave, then lane (= memory order)
static __global__ __launch_bounds__(SORT_THREADS) void sort_scatter_kernel(const SortMeta sm, int pass, const uint32_t* __restrict__ kin,
                                                                    const uint32_t* __restrict__ vin,
                                                                    uint32_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                                    const uint32_t* __restrict__ hist) {
    __shared__ uint32_t off[256];
    __shared__ uint32_t wc[SORT_THREADS / 64][256];
    const int tile = blockIdx.x;
    const int t = sort_table_of_tile(sm, tile);
    const int64_t start = sm.seg_off[t] + static_cast<int64_t>(tile - sm.tile_off[t]) * sm.tile;
    const int len = static_cast<int>(min<int64_t>(sm.tile, sm.seg_off[t + 1] - start));
    if (pass >= sm.passes[t]) {          // this table is already sorted: carry it to the other buffer
        for (int e = threadIdx.x; e < len; e += SORT_THREADS) {
            kout[start + e] = kin[start + e];
            vout[start + e] = vin[start + e];
        }
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (sm.fold_scan) {
        // hist holds the raw per-tile counts: this digit's output offset = segment base + keys of smaller digits (all tiles of the
        // table) + keys of this digit in earlier tiles -- what sort_scan_kernel would have written
        const int t0 = sm.tile_off[t], t1 = sm.tile_off[t + 1];
        const int d = threadIdx.x;
        uint32_t before = 0, total = 0;
        for (int tt = t0; tt < t1; tt += 8) {
            uint32_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = tt + k < t1 ? hist[static_cast<int64_t>(tt + k) * 256 + d] : 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                total += v[k];
                before += (tt + k < tile) ? v[k] : 0u;
            }
        }
        off[d] = total;
        __syncthreads();
        uint32_t incl = total;
        for (int o = 1; o < 256; o <<= 1) {               // inclusive scan of the digit totals (Hillis-Steele in LDS)
            const uint32_t add = d >= o ? off[d - o] : 0u;
            __syncthreads();
            incl += add;
            off[d] = incl;
            __syncthreads();
        }
        off[d] = static_cast<uint32_t>(sm.seg_off[t]) + (incl - total) + before;
    } else {
        off[threadIdx.x] = hist[static_cast<int64_t>(tile) * 256 + threadIdx.x];
    }
#pragma unroll
    for (int w = 0; w < SORT_THREADS / 64; ++w) wc[w][threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int r = 0; r < sm.items; ++r) {
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


// ------------------------------------------------------------------------------- rank sort (short segments, ONE launch)
// A segment of n <= RANK_SORT_MAX_KEYS keys is sorted by COUNTING: the output position of entry i is the number of entries j with
// (key_j, j) < (key_i, i) -- n^2 comparisons, embarrassingly parallel, one launch instead of two per 8-bit pass (a 3-pass sort of
// the 8 192 keys of a strong-scaling shard was 6 launches of ~5 us on the critical path of a single-stream step; the 67 M
// comparisons spread over the chip take less than two of them).  Stable by construction (ties broken by position), so the
// permutation -- and everything downstream of it -- is bit-identical to the radix passes'.
// Workgroup = RANK_EPB consecutive entries i of one segment (their keys are wave-uniform: scalar registers); thread t owns the
// entries j = t, t + 256, ... of the segment (coalesced loads, all in flight at once) and counts, for each i, how many of them
// come first: `<=` for rows of j wholly in front of the block, `<` behind it, the exact (key, position) order in the one row
// that holds the block.  Counters leave through LDS ([thread][i], summed by columns).
#define RANK_EPB 16
#define RANK_THREADS 256
#define RANK_ROWS_MAX 64                       // rows of 256 entries per segment: n <= 16 384
#define RANK_SORT_MAX_KEYS (RANK_ROWS_MAX * RANK_THREADS)
#define RANK_SORT_MAX_WORK (3ll << 26)         // sum over segments of n^2 up to which the counting sort is taken (2 x 8 192^2 and a bit)

static inline bool rank_sort_enabled() {          // SWR_RANK_SORT=0: radix passes (read per call: the tests compare both in one process)
    const char* e = getenv("SWR_RANK_SORT");
    return !(e && e[0] == '0');
}

// where the entries come from: straight from the lookup's key matrix [slot][B] -- segment t is the concatenation of the B keys of
// each of its slots, entry (slot, b) carries the payload slot << 24 | b -- so the launch that used to write the unsorted entries
// (build_keys_kernel) is not needed; the zero-fill that launch carried rides here
struct RankSrc {
    const uint32_t* keys;
    int64_t B;
    int16_t slot[MAX_SLOTS];         // slots of segment t, in segment order: slot[first[t]] .. slot[first[t + 1] - 1]
    int16_t first[MAX_SLOTS + 1];
    int16_t slot0[MAX_SLOTS];        // = slot[first[t]] and first[t + 1] - first[t]: what a one-slot segment (the common case) needs,
    int16_t n_of[MAX_SLOTS];         //   one load level away from t (through `first` the key loads sat behind two dependent loads)
    uint4* zero;                     // 16-byte words to clear (the caller's accumulators)
    int64_t zero16;
};

// THREADS: threads of the workgroup = entries per row (256 in the launch of its own; 512 where the sort shares a launch with the direct
// sums of the embedding backward, embed_bwd.hip); `tile` / `n_tiles`: this workgroup among the sort's workgroups
template <int ROWS, int THREADS>
static __device__ __forceinline__ void rank_sort_body(const SortMeta& sm, const RankSrc& src, uint32_t* __restrict__ kout,
                                                      uint32_t* __restrict__ vout, const int tile, const int n_tiles) {
        __shared__ uint32_t cnt[THREADS][RANK_EPB + 1];
    const int tid = threadIdx.x;
    const int t = sort_table_of_tile(sm, tile);
    const int64_t seg = sm.seg_off[t];
    const int len = static_cast<int>(sm.seg_off[t + 1] - seg);
    const int i0 = (tile - sm.tile_off[t]) * RANK_EPB;                 // first entry of the block (inside the segment)
    const int s_first = src.first[t];
    const int slot_first = src.slot0[t];
    const bool one_slot = src.n_of[t] == 1;                            // (wave-uniform) the common case: no division per entry
    const int B = static_cast<int>(src.B);
    const uint32_t* __restrict__ k0 = src.keys + static_cast<int64_t>(slot_first) * src.B;
    // entry j of the segment -> (slot, sample); its key
    auto locate = [&](int j, int& slot, int& b) {
        if (one_slot) { slot = slot_first; b = j; return; }
        const int k = j / B;
        slot = src.slot[s_first + k];
        b = j - k * B;
    };
    auto key_at = [&](int j) -> uint32_t {
        if (one_slot) return k0[j];
        int slot, b;
        locate(j, slot, b);
        return src.keys[static_cast<int64_t>(slot) * src.B + b];
    };
    const int row_of_block = i0 / THREADS;                        // (RANK_EPB divides THREADS: the block lies in one row)
    // x = the entry's key as the loop below compares it: every comparison is `x < key_i + 1` with the right-hand sides loop-invariant
    // scalars.  Rows in front of the block count when key_j <= key_i: x = key_j; the block's row and the rows behind it when
    // key_j < key_i: x = key_j + 1 (real keys are < 0xFFFFFFFF, the launcher checks: no wrap); padding never counts: x = 0xFFFFFFFF
    uint32_t x[ROWS];
    uint32_t kb = 0xFFFFFFFFu;                                         // the thread's key in the block's own row
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int j = r * THREADS + tid;
        const uint32_t kj = j < len ? key_at(j) : 0xFFFFFFFFu;
        if (r == row_of_block) kb = kj;
        x[r] = (j < len && r >= row_of_block) ? kj + 1u : kj;
    }
    uint32_t thr[RANK_EPB];
#pragma unroll
    for (int e = 0; e < RANK_EPB; ++e)
        thr[e] = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(key_at(min(i0 + e, len - 1))))) + 1u;
    uint32_t c[RANK_EPB];
    // the block's own row: entries with an EQUAL key come first when they stand in front
    {
        const int j = row_of_block * THREADS + tid;
#pragma unroll
        for (int e = 0; e < RANK_EPB; ++e) c[e] = (j < len && kb + 1u == thr[e] && j < i0 + e) ? 1u : 0u;
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
#pragma unroll
        for (int e = 0; e < RANK_EPB; ++e) c[e] += x[r] < thr[e] ? 1u : 0u;
    }
#pragma unroll
    for (int e = 0; e < RANK_EPB; ++e) cnt[tid][e] = c[e];
    __syncthreads();
    // column sums: thread (e, part) adds 16 of the 256 partial counts of entry e; then the 16 parts
    constexpr int PARTS = THREADS / RANK_EPB;
    const int e = tid & (RANK_EPB - 1), part = tid / RANK_EPB;
    uint32_t sum = 0;
#pragma unroll
    for (int q = 0; q < THREADS / PARTS; ++q) sum += cnt[part * (THREADS / PARTS) + q][e];
    __syncthreads();
    cnt[part][e] = sum;
    __syncthreads();
    if (tid < RANK_EPB && i0 + tid < len) {
        uint32_t pos = 0;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) pos += cnt[q][tid];
        int slot, b;
        locate(i0 + tid, slot, b);
        kout[seg + pos] = key_at(i0 + tid);
        vout[seg + pos] = (static_cast<uint32_t>(slot) << 24) | static_cast<uint32_t>(b);
    }
    // (behind the sort's own loads: in front of them the fill's stores delayed the first round trip of every workgroup)
    const int64_t stride = static_cast<int64_t>(n_tiles) * THREADS;
    for (int64_t z = static_cast<int64_t>(tile) * THREADS + tid; z < src.zero16; z += stride) src.zero[z] = make_uint4(0u, 0u, 0u, 0u);
}

template <int ROWS>
static __global__ __launch_bounds__(RANK_THREADS) void rank_sort_kernel(const SortMeta sm, const RankSrc src, uint32_t* __restrict__ kout,
                                                                         uint32_t* __restrict__ vout) {
    rank_sort_body<ROWS, RANK_THREADS>(sm, src, kout, vout, static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x));
}

// the counting sort of every segment, one launch: sm.tile == RANK_EPB (sm.tile_off counts blocks); result in buffer 1
static inline void rank_sort_launch(const SortMeta& sm, const RankSrc& src, int64_t max_keys, uint32_t* const kbuf[2], uint32_t* const vbuf[2],
                                    hipStream_t st) {
    const dim3 grid(static_cast<unsigned>(sm.n_tiles)), block(RANK_THREADS);
    const int rows = static_cast<int>((max_keys + RANK_THREADS - 1) / RANK_THREADS);
    if (rows <= 8) hipLaunchKernelGGL(rank_sort_kernel<8>, grid, block, 0, st, sm, src, kbuf[1], vbuf[1]);
    else if (rows <= 16) hipLaunchKernelGGL(rank_sort_kernel<16>, grid, block, 0, st, sm, src, kbuf[1], vbuf[1]);
    else if (rows <= 32) hipLaunchKernelGGL(rank_sort_kernel<32>, grid, block, 0, st, sm, src, kbuf[1], vbuf[1]);
    else hipLaunchKernelGGL(rank_sort_kernel<RANK_ROWS_MAX>, grid, block, 0, st, sm, src, kbuf[1], vbuf[1]);
}

// every pass of the sort on `st`: keys / values ping-pong between buffer 0 and 1; the result is in buffer (n_passes & 1)
static inline void radix_sort_launch(const SortMeta& sm, int n_passes, uint32_t* const kbuf[2], uint32_t* const vbuf[2],
                                     uint32_t* hist, hipStream_t st) {
    int cur = 0;
    for (int pass = 0; pass < n_passes; ++pass) {
        hipLaunchKernelGGL(sort_hist_kernel, dim3(sm.n_tiles), dim3(SORT_THREADS), 0, st, sm, pass, kbuf[cur], hist);
        if (!sm.fold_scan) hipLaunchKernelGGL(sort_scan_kernel, dim3(sm.n_tables), dim3(SCAN_THREADS), 0, st, sm, pass, hist);
        hipLaunchKernelGGL(sort_scatter_kernel, dim3(sm.n_tiles), dim3(SORT_THREADS), 0, st, sm, pass, kbuf[cur], vbuf[cur],
                           kbuf[cur ^ 1], vbuf[cur ^ 1], hist);
        cur ^= 1;
    }
}
