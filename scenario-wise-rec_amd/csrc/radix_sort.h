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
