// Evaluation metrics on the device: per-domain and overall log-loss and ROC-AUC of a prediction
// run, replacing `.tolist()` + sklearn.metrics.{log_loss, roc_auc_score} on the host
// (reference trainers/ctr_trainer.py:99-165).  Exact arithmetic where the metric is one:
//   AUC  = U / (P N) with U the Mann-Whitney statistic, ties counted 1/2 -- what sklearn's trapezoid over the distinct
//          thresholds integrates to.  2U = sum over positives of (#negatives with a smaller score + #negatives with a
//          smaller-or-equal score) is an INTEGER: scores are sorted (own radix sort, csrc/radix_sort.h), negatives
//          prefix-summed, and every positive finds the two ends of its tie group by binary search.  Integer atomics:
//          order-free, bitwise reproducible.
//   log-loss = -mean(y log p + (1 - y) log(1 - p)) in fp64 with p clipped to [eps, 1 - eps], eps = 2^-52 (sklearn clips
//          a float64 array); per-block partial sums are added in block order (fixed -> reproducible).
// Per-domain figures come from the same sorted scores after ONE more stable counting pass on the domain id.
// HBM / latency-bound integer work; nothing here is on the training path.
#include "radix_sort.h"

#define MT_THREADS 256
#define MT_SCAN_TILE 1024            // entries per scan workgroup (4 per thread)
#define MT_MAX_DOMAINS 254

struct MetricsPlan {
    SortMeta sm;
    int64_t n;
    int n_blocks, n_scan_tiles;
    size_t off_k0, off_k1, off_v0, off_v1, off_hist, off_prefix, off_tilesum, off_ll, off_seg, total;
};

static size_t mt_align(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

static int mt_plan(int64_t n, int D, MetricsPlan& p) {
    SWR_REQUIRE(n >= 0 && n < (1ll << 31) && D >= 1 && D <= MT_MAX_DOMAINS, SWR_ERR_ARG);
    p.n = n;
    int items = SORT_ITEMS_MAX;
    while (items > 1 && n / (SORT_THREADS * items) < 512) items >>= 1;
    p.sm.items = items;
    p.sm.tile = SORT_THREADS * items;
    p.sm.fold_scan = 0;
    p.sm.n_tables = 1;
    p.sm.seg_off[0] = 0;
    p.sm.seg_off[1] = n;
    p.sm.tile_off[0] = 0;
    p.sm.n_tiles = static_cast<int>(swr_ceil_div(std::max<int64_t>(n, 1), p.sm.tile));
    p.sm.tile_off[1] = p.sm.n_tiles;
    p.sm.passes[0] = 4;
    p.n_blocks = static_cast<int>(swr_ceil_div(std::max<int64_t>(n, 1), MT_THREADS));
    p.n_scan_tiles = static_cast<int>(swr_ceil_div(std::max<int64_t>(n, 1), MT_SCAN_TILE));
    size_t off = 0;
    const size_t kb = mt_align(static_cast<size_t>(std::max<int64_t>(n, 1)) * 4);
    p.off_k0 = off; off += kb;
    p.off_k1 = off; off += kb;
    p.off_v0 = off; off += kb;
    p.off_v1 = off; off += kb;
    p.off_hist = off; off += mt_align(static_cast<size_t>(p.sm.n_tiles) * 256 * 4);
    p.off_prefix = off; off += mt_align((static_cast<size_t>(n) + 1) * 4);
    p.off_tilesum = off; off += mt_align((static_cast<size_t>(p.n_scan_tiles) + 1) * 4);
    p.off_ll = off; off += mt_align(static_cast<size_t>(p.n_blocks) * (D + 1) * 8);
    p.off_seg = off; off += mt_align(static_cast<size_t>(D + 2) * 8);
    p.total = off;
    return SWR_OK;
}

// float bits -> unsigned key with the same total order (negative values and -0 included)
__device__ __forceinline__ uint32_t mt_order_key(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// keys / payloads, per-domain counts, per-block log-loss partial sums (thread d < D + 1 walks the block's 256 terms in order)
__global__ __launch_bounds__(MT_THREADS) void metrics_prepare_kernel(const float* __restrict__ p, const void* __restrict__ y,
                                                                     int y_dtype, const void* __restrict__ dom, int dom_dtype,
                                                                     int64_t n, int D, uint32_t* __restrict__ keys,
                                                                     uint32_t* __restrict__ vals, double* __restrict__ ll_part,
                                                                     unsigned long long* __restrict__ counts) {
    __shared__ double term[MT_THREADS];
    __shared__ int sdom[MT_THREADS];          // D = outside [0, D); -1 = past the end
    __shared__ int spos[MT_THREADS];
    const int64_t i = static_cast<int64_t>(blockIdx.x) * MT_THREADS + threadIdx.x;
    int d = -1, pos = 0;
    double t = 0.0;
    if (i < n) {
        const float pf = p[i];
        pos = swr_load_value(y, y_dtype, i) > 0.5f ? 1 : 0;
        const int64_t dv = swr_load_index(dom, dom_dtype, i);
        d = (dv >= 0 && dv < D) ? static_cast<int>(dv) : D;
        keys[i] = mt_order_key(pf);
        vals[i] = (pos ? 0x80000000u : 0u) | static_cast<uint32_t>(d);
        const double eps = 2.220446049250313e-16;
        const double pc = fmin(fmax(static_cast<double>(pf), eps), 1.0 - eps);
        t = -(pos ? log(pc) : log(1.0 - pc));
    }
    term[threadIdx.x] = t;
    sdom[threadIdx.x] = d;
    spos[threadIdx.x] = pos;
    __syncthreads();
    if (static_cast<int>(threadIdx.x) <= D) {
        const int me = threadIdx.x;             // me == D: every row (the overall figure)
        double s = 0.0;
        unsigned long long cnt = 0, npos = 0;
        for (int j = 0; j < MT_THREADS; ++j) {
            const bool mine = sdom[j] >= 0 && (me == D || sdom[j] == me);
            if (mine) {
                s += term[j];
                cnt += 1;
                npos += spos[j];
            }
        }
        ll_part[static_cast<int64_t>(blockIdx.x) * (D + 1) + me] = s;
        if (cnt) {
            atomicAdd(&counts[me * 3 + 0], cnt);
            atomicAdd(&counts[me * 3 + 1], npos);
        }
    }
}

// log-loss sums: one workgroup per domain, partial sums of the blocks added in block order by 256 lanes (lane l takes
// blocks l, l + 256, ...), then a fixed tree
__global__ __launch_bounds__(MT_THREADS) void metrics_ll_kernel(const double* __restrict__ ll_part, int n_blocks, int D,
                                                                double* __restrict__ ll_sum) {
    __shared__ double sm[MT_THREADS];
    const int d = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < n_blocks; b += MT_THREADS) s += ll_part[static_cast<int64_t>(b) * (D + 1) + d];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int st = MT_THREADS / 2; st > 0; st >>= 1) {
        if (static_cast<int>(threadIdx.x) < st) sm[threadIdx.x] += sm[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) ll_sum[d] = sm[0];
}

// exclusive prefix count of NEGATIVES along the sorted order, three launches: tile totals, scan of the totals, apply
__global__ __launch_bounds__(MT_THREADS) void metrics_neg_tiles_kernel(const uint32_t* __restrict__ payload, int64_t n,
                                                                       uint32_t* __restrict__ tile_sum) {
    __shared__ uint32_t sm[MT_THREADS];
    const int64_t base = static_cast<int64_t>(blockIdx.x) * MT_SCAN_TILE;
    uint32_t c = 0;
    for (int k = 0; k < MT_SCAN_TILE / MT_THREADS; ++k) {
        const int64_t i = base + k * MT_THREADS + threadIdx.x;
        if (i < n) c += (payload[i] >> 31) ^ 1u;
    }
    sm[threadIdx.x] = c;
    __syncthreads();
    for (int st = MT_THREADS / 2; st > 0; st >>= 1) {
        if (static_cast<int>(threadIdx.x) < st) sm[threadIdx.x] += sm[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = sm[0];
}

__global__ __launch_bounds__(1024) void metrics_scan_tiles_kernel(uint32_t* __restrict__ tile_sum, int n_tiles) {
    // one workgroup: thread t owns a contiguous run of tiles; run totals scanned in LDS; in place -> exclusive offsets
    __shared__ uint32_t run[1024];
    const int per = (n_tiles + 1023) / 1024;
    const int a = min(static_cast<int>(threadIdx.x) * per, n_tiles), b = min(a + per, n_tiles);
    uint32_t s = 0;
    for (int t = a; t < b; ++t) s += tile_sum[t];
    run[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t add = static_cast<int>(threadIdx.x) >= off ? run[threadIdx.x - off] : 0u;
        __syncthreads();
        run[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t o = run[threadIdx.x] - s;
    for (int t = a; t < b; ++t) {
        const uint32_t c = tile_sum[t];
        tile_sum[t] = o;
        o += c;
    }
    if (threadIdx.x == 1023) tile_sum[n_tiles] = run[1023];
}

__global__ __launch_bounds__(MT_THREADS) void metrics_neg_apply_kernel(const uint32_t* __restrict__ payload, int64_t n,
                                                                       const uint32_t* __restrict__ tile_off,
                                                                       uint32_t* __restrict__ prefix) {
    // thread = 4 consecutive entries; wave scan + cross-wave offsets
    __shared__ uint32_t wsum[MT_THREADS / 64];
    const int64_t base = static_cast<int64_t>(blockIdx.x) * MT_SCAN_TILE + 4 * threadIdx.x;
    uint32_t f[4], c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f[k] = (base + k < n) ? ((payload[base + k] >> 31) ^ 1u) : 0u;
        c += f[k];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(inc, off);
        if (lane >= off) inc += v;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t before = tile_off[blockIdx.x];
    for (int w = 0; w < wave; ++w) before += wsum[w];
    uint32_t o = before + inc - c;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) prefix[base + k] = o;
        o += f[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) prefix[n] = tile_off[gridDim.x];        // total (written by the tile scan)
}

// segment bounds of the domains in the (domain, score)-sorted order: seg[d] = first entry of domain d, seg[D + 1] = n
__global__ void metrics_segments_kernel(const unsigned long long* __restrict__ counts, int D, int64_t n,
                                        int64_t* __restrict__ seg) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int64_t o = 0;
    for (int d = 0; d <= D; ++d) {            // bucket D = rows whose domain id lies outside [0, D): up to n
        seg[d] = o;
        if (d < D) o += static_cast<int64_t>(counts[d * 3]);
    }
    seg[D + 1] = n;
}

// 2U: every positive adds (#negatives below its tie group) + (#negatives up to the end of its tie group), inside its
// segment [lo, hi).  by_domain: segments = domains (entries sorted by domain, then score); else one segment, slot D.
__global__ __launch_bounds__(MT_THREADS) void metrics_auc_kernel(const uint32_t* __restrict__ score, const uint32_t* __restrict__ payload,
                                                                 const uint32_t* __restrict__ prefix, int64_t n, int D,
                                                                 const int64_t* __restrict__ seg, int by_domain,
                                                                 unsigned long long* __restrict__ counts) {
    __shared__ unsigned long long acc[MT_MAX_DOMAINS + 2];
    for (int j = threadIdx.x; j <= D; j += MT_THREADS) acc[j] = 0ull;
    __syncthreads();
    const int64_t i = static_cast<int64_t>(blockIdx.x) * MT_THREADS + threadIdx.x;
    if (i < n && (payload[i] >> 31)) {
        const int d = static_cast<int>(payload[i] & 0xFFu);
        if (!by_domain || d < D) {
            const int64_t lo0 = by_domain ? seg[d] : 0, hi0 = by_domain ? seg[d + 1] : n;
            const uint32_t key = score[i];
            int64_t lo = lo0, hi = i;                                  // first entry of the segment with score >= key
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (score[mid] < key) lo = mid + 1; else hi = mid;
            }
            const int64_t a = lo;
            lo = i + 1; hi = hi0;                                      // first entry with score > key
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (score[mid] <= key) lo = mid + 1; else hi = mid;
            }
            const int64_t b = lo;
            const unsigned long long base = prefix[lo0];
            const unsigned long long u2 = (static_cast<unsigned long long>(prefix[a]) - base) +
                                          (static_cast<unsigned long long>(prefix[b]) - base);
            atomicAdd(&acc[by_domain ? d : D], u2);
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j <= D; j += MT_THREADS)
        if (acc[j]) atomicAdd(&counts[j * 3 + 2], acc[j]);
}

extern "C" size_t swr_eval_metrics_workspace_bytes(int64_t n, int n_domains) {
    MetricsPlan p;
    if (mt_plan(n, n_domains, p) != SWR_OK) return 0;
    return p.total;
}

extern "C" int swr_eval_metrics(const float* prob, const void* label, int label_dtype, const void* domain, int domain_dtype,
                                int64_t n, int n_domains, unsigned long long* counts, double* logloss_sum,
                                void* workspace, size_t workspace_bytes, void* stream) {
    SWR_REQUIRE(counts && logloss_sum && workspace && (n == 0 || (prob && label && domain)), SWR_ERR_ARG);
    SWR_REQUIRE(swr_is_value_dtype(label_dtype) && swr_is_index_dtype(domain_dtype), SWR_ERR_DTYPE);
    MetricsPlan p;
    int rc = mt_plan(n, n_domains, p);
    if (rc != SWR_OK) return rc;
    SWR_REQUIRE(workspace_bytes >= p.total, SWR_ERR_WORKSPACE);
    const int D = n_domains;
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = swr_zero_async(counts, static_cast<size_t>(D + 1) * 3 * 8, st);
    if (rc != SWR_OK) return rc;
    rc = swr_zero_async(logloss_sum, static_cast<size_t>(D + 1) * 8, st);
    if (rc != SWR_OK || n == 0) return rc;
    char* ws = static_cast<char*>(workspace);
    uint32_t* kbuf[2] = {reinterpret_cast<uint32_t*>(ws + p.off_k0), reinterpret_cast<uint32_t*>(ws + p.off_k1)};
    uint32_t* vbuf[2] = {reinterpret_cast<uint32_t*>(ws + p.off_v0), reinterpret_cast<uint32_t*>(ws + p.off_v1)};
    uint32_t* hist = reinterpret_cast<uint32_t*>(ws + p.off_hist);
    uint32_t* prefix = reinterpret_cast<uint32_t*>(ws + p.off_prefix);
    uint32_t* tilesum = reinterpret_cast<uint32_t*>(ws + p.off_tilesum);
    double* ll_part = reinterpret_cast<double*>(ws + p.off_ll);
    int64_t* seg = reinterpret_cast<int64_t*>(ws + p.off_seg);

    hipLaunchKernelGGL(metrics_prepare_kernel, dim3(p.n_blocks), dim3(MT_THREADS), 0, st, prob, label, label_dtype, domain,
                       domain_dtype, n, D, kbuf[0], vbuf[0], ll_part, counts);
    hipLaunchKernelGGL(metrics_ll_kernel, dim3(D + 1), dim3(MT_THREADS), 0, st, ll_part, p.n_blocks, D, logloss_sum);
    auto neg_prefix = [&](const uint32_t* payload) {
        hipLaunchKernelGGL(metrics_neg_tiles_kernel, dim3(p.n_scan_tiles), dim3(MT_THREADS), 0, st, payload, n, tilesum);
        hipLaunchKernelGGL(metrics_scan_tiles_kernel, dim3(1), dim3(1024), 0, st, tilesum, p.n_scan_tiles);
        hipLaunchKernelGGL(metrics_neg_apply_kernel, dim3(p.n_scan_tiles), dim3(MT_THREADS), 0, st, payload, n, tilesum, prefix);
    };
    // overall: sort by score (4 passes: result back in buffer 0)
    radix_sort_launch(p.sm, 4, kbuf, vbuf, hist, st);
    neg_prefix(vbuf[0]);
    hipLaunchKernelGGL(metrics_auc_kernel, dim3(p.n_blocks), dim3(MT_THREADS), 0, st, kbuf[0], vbuf[0], prefix, n, D, seg, 0, counts);
    // per domain: one more stable pass with the payload (low byte = domain) as the key -> (domain, score) order in buffer 1
    SortMeta sm1 = p.sm;
    sm1.passes[0] = 1;
    uint32_t* k2[2] = {vbuf[0], vbuf[1]};
    uint32_t* v2[2] = {kbuf[0], kbuf[1]};
    radix_sort_launch(sm1, 1, k2, v2, hist, st);
    hipLaunchKernelGGL(metrics_segments_kernel, dim3(1), dim3(64), 0, st, counts, D, n, seg);
    neg_prefix(vbuf[1]);
    hipLaunchKernelGGL(metrics_auc_kernel, dim3(p.n_blocks), dim3(MT_THREADS), 0, st, kbuf[1], vbuf[1], prefix, n, D, seg, 1, counts);
    return swr_launch_status();
}
