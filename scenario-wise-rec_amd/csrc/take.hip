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

__global__ __launch_bounds__(TAKE_THREADS) void take_rows_kernel(const TakeK k) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * TAKE_THREADS + threadIdx.x;
    if (i >= k.n_out) return;
    int64_t r = k.perm ? k.perm[i] : i;           // no permutation: a straight copy of every column's first n_out rows
    if (r < 0 || r >= k.n_in) {
        if (k.err) atomicOr(k.err, SWR_FLAG_INDEX_OOR);
        r = r < 0 ? 0 : k.n_in - 1;
    }
    for (int c = 0; c < k.n_columns; ++c) {
        const swr_take_column& col = k.col[c];
        switch (col.elem_bytes) {
            case 1: static_cast<uint8_t*>(col.dst)[i] = static_cast<const uint8_t*>(col.src)[r]; break;
            case 2: static_cast<uint16_t*>(col.dst)[i] = static_cast<const uint16_t*>(col.src)[r]; break;
            case 4: static_cast<uint32_t*>(col.dst)[i] = static_cast<const uint32_t*>(col.src)[r]; break;
            default: static_cast<uint64_t*>(col.dst)[i] = static_cast<const uint64_t*>(col.src)[r]; break;
        }
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
    hipLaunchKernelGGL(take_rows_kernel, dim3(static_cast<unsigned>(swr_ceil_div(n_out, TAKE_THREADS))), dim3(TAKE_THREADS), 0,
                       static_cast<hipStream_t>(stream), k);
    return swr_launch_status();
}
