// Row permutation of a columnar, device-resident dataset (include/swr.h "input columns"): what
// DataLoader(TorchDataset(x, y), shuffle=True) does one python dict per row on the host (reference utils/data.py:11-22,55)
// is one byte-copy launch per epoch here: dst[c][i] = src[c][perm[i]] for every column.  HBM-bound gather of small
// elements: consecutive lanes take consecutive output rows (coalesced stores; the loads are as random as the permutation).
#include "common.h"

#define TAKE_THREADS 256

struct TakeK {
    swr_take_column col[SWR_TAKE_MAX_COLUMNS];
    int n_columns;
    const int64_t* perm;
    int64_t n_in, n_out;
    uint32_t* err;
};

// grid = (row blocks, columns): a workgroup copies 256 x ROWS_PER_THREAD rows of ONE column -- every load is independent
// of every store (one thread walking all columns of its row serialised ~37 load -> store round trips: 38 us for a
// 65 536-row batch of the KuaiRand schema; this layout: the copy is bandwidth-bound)
#define TAKE_RPT 4
__global__ __launch_bounds__(TAKE_THREADS) void take_rows_kernel(const TakeK k) {
    const swr_take_column& col = k.col[blockIdx.y];
    const int64_t i0 = (static_cast<int64_t>(blockIdx.x) * TAKE_RPT) * TAKE_THREADS + threadIdx.x;
    int64_t r[TAKE_RPT];
#pragma unroll
    for (int u = 0; u < TAKE_RPT; ++u) {
        const int64_t i = i0 + static_cast<int64_t>(u) * TAKE_THREADS;
        int64_t v = i < k.n_out ? (k.perm ? k.perm[i] : i) : 0;
        if (v < 0 || v >= k.n_in) {
            if (k.err && blockIdx.y == 0) atomicOr(k.err, SWR_FLAG_INDEX_OOR);
            v = v < 0 ? 0 : k.n_in - 1;
        }
        r[u] = v;
    }
    switch (col.elem_bytes) {
#define TAKE_CASE(T)                                                                                        \
    {                                                                                                       \
        T v[TAKE_RPT];                                                                                      \
        _Pragma("unroll") for (int u = 0; u < TAKE_RPT; ++u) v[u] = static_cast<const T*>(col.src)[r[u]];   \
        _Pragma("unroll") for (int u = 0; u < TAKE_RPT; ++u) {                                              \
            const int64_t i = i0 + static_cast<int64_t>(u) * TAKE_THREADS;                                  \
            if (i < k.n_out) static_cast<T*>(col.dst)[i] = v[u];                                            \
        }                                                                                                   \
    }
        case 1: TAKE_CASE(uint8_t) break;
        case 2: TAKE_CASE(uint16_t) break;
        case 4: TAKE_CASE(uint32_t) break;
        default: TAKE_CASE(uint64_t) break;
#undef TAKE_CASE
    }
}

extern "C" int swr_take_rows(const swr_take_column* columns, int n_columns, const int64_t* perm, int64_t n_in, int64_t n_out,
                             uint32_t* err_flag, void* stream) {
    SWR_REQUIRE(columns && n_columns > 0 && n_columns <= SWR_TAKE_MAX_COLUMNS && n_in > 0 && n_out >= 0 && (perm || n_out <= n_in),
                SWR_ERR_ARG);
    TakeK k;
    for (int c = 0; c < n_columns; ++c) {
        const int eb = columns[c].elem_bytes;
        SWR_REQUIRE(columns[c].src && columns[c].dst && (eb == 1 || eb == 2 || eb == 4 || eb == 8), SWR_ERR_ARG);
        k.col[c] = columns[c];
    }
    if (n_out == 0) return SWR_OK;
    k.n_columns = n_columns; k.perm = perm; k.n_in = n_in; k.n_out = n_out; k.err = err_flag;
    hipLaunchKernelGGL(take_rows_kernel, dim3(static_cast<unsigned>(swr_ceil_div(n_out, TAKE_THREADS * TAKE_RPT)), static_cast<unsigned>(n_columns)),
                       dim3(TAKE_THREADS), 0, static_cast<hipStream_t>(stream), k);
    return swr_launch_status();
}
