// Device side of the data-parallel exchange step (include/swr.h "exchange"): replaces the replica
// gradient reduction of torch.nn.DataParallel (reference trainers/ctr_trainer.py:45-47) after the step's one
// all-gather.  HBM/L2-bound integer + fp32 work, one launch:
//   workgroups [0, dense_blocks)  : mean of the `world` gradient arenas, ranks added in order;
//   the rest, 256 entries each    : sort-free merge of the `world` row lists of the large tables.
// A rank's list is ordered by row (negative ids = holes that still carry the complement of their row), so an entry
// finds its row in another list with one binary search; the searches of one entry over the other lists run in
// lock step (independent loads in flight), one thread per entry; the 16..64 gradient columns are then summed by
// consecutive lanes, positions handed over through LDS.
#include "common.h"

#define DPF_THREADS 256

struct DpK {
    const float* recv_d;
    const float* recv;                       // the row lists' buffer
    int world;
    int64_t dstride, total, A;
    float* dense_out;
    float scale;
    int n_tables;
    int dense_blocks;
    int block0[SWR_DP_MAX_TABLES + 1];       // first merge workgroup of each table (relative to dense_blocks)
    swr_dp_table tab[SWR_DP_MAX_TABLES];
};

__device__ __forceinline__ int dp_row_of(int32_t r) { return r < 0 ? ~r : r; }

__global__ __launch_bounds__(DPF_THREADS) void dp_finish_kernel(const DpK k) {
    const int tid = threadIdx.x;
    if (static_cast<int>(blockIdx.x) < k.dense_blocks) {
        // 4 words per thread; A and total are multiples of 4 words when the arena is 16-byte aligned (checked on the host)
        const int64_t j = (static_cast<int64_t>(blockIdx.x) * DPF_THREADS + tid) * 4;
        if (j >= k.A) return;
        if (j + 4 <= k.A) {
            float4 s = *reinterpret_cast<const float4*>(k.recv_d + j);
            for (int r = 1; r < k.world; ++r) {
                const float4 v = *reinterpret_cast<const float4*>(k.recv_d + r * k.dstride + j);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            *reinterpret_cast<float4*>(k.dense_out + j) = make_float4(s.x * k.scale, s.y * k.scale, s.z * k.scale, s.w * k.scale);
        } else {
            for (int64_t q = j; q < k.A; ++q) {
                float s = k.recv_d[q];
                for (int r = 1; r < k.world; ++r) s += k.recv_d[r * k.dstride + q];
                k.dense_out[q] = s * k.scale;
            }
        }
        return;
    }
    __shared__ int32_t s_pos[DPF_THREADS][SWR_DP_MAX_WORLD];      // position of the entry's row in list r (-1: absent)
    __shared__ int32_t s_row[DPF_THREADS];                          // row id if this entry owns its row, else -1
    const int mb = static_cast<int>(blockIdx.x) - k.dense_blocks;
    int ti = 0;
    while (ti + 1 < k.n_tables && k.block0[ti + 1] <= mb) ++ti;
    const swr_dp_table& T = k.tab[ti];
    const int64_t n = T.n, n_all = n * k.world;
    const int64_t e0 = static_cast<int64_t>(mb - k.block0[ti]) * DPF_THREADS;
    const int32_t* __restrict__ rows = reinterpret_cast<const int32_t*>(k.recv) + T.row_off;    // + r * total
    const float* __restrict__ grads = k.recv + T.grad_off;

    // ---- phase 1: one thread per entry
    {
        const int64_t e = e0 + tid;
        int32_t own = -1;
        if (e < n_all) {
            const int r = static_cast<int>(e / n);
            const int64_t i = e - r * n;
            const int32_t id = rows[r * k.total + i];
            if (id >= 0) {
                int64_t lo[SWR_DP_MAX_WORLD], hi[SWR_DP_MAX_WORLD];
#pragma unroll
                for (int q = 0; q < SWR_DP_MAX_WORLD; ++q) { lo[q] = 0; hi[q] = (q < k.world && q != r) ? n : 0; }
                bool more = true;
                while (more) {                                   // lower bound of `id` in every other list, in lock step
                    more = false;
#pragma unroll
                    for (int q = 0; q < SWR_DP_MAX_WORLD; ++q) {
                        if (lo[q] < hi[q]) {
                            const int64_t mid = (lo[q] + hi[q]) >> 1;
                            if (dp_row_of(rows[q * k.total + mid]) < id) lo[q] = mid + 1; else hi[q] = mid;
                            more = more || lo[q] < hi[q];
                        }
                    }
                }
                bool owner = true;
#pragma unroll
                for (int q = 0; q < SWR_DP_MAX_WORLD; ++q) {
                    int32_t pos = -1;
                    if (q < k.world && q != r && lo[q] < n && rows[q * k.total + lo[q]] == id) {   // the run's head
                        pos = static_cast<int32_t>(lo[q]);
                        owner = owner && q > r;
                    }
                    if (q == r) pos = static_cast<int32_t>(i);
                    s_pos[tid][q] = pos;
                }
                own = owner ? id : -1;
            }
            T.out_row[e] = own;
        }
        s_row[tid] = own;
    }
    __syncthreads();
    // ---- phase 2: consecutive lanes take the gradient columns
    const int dim = T.dim;
    for (int idx = tid; idx < DPF_THREADS * dim; idx += DPF_THREADS) {
        const int le = idx / dim, c = idx - le * dim;
        const int64_t e = e0 + le;
        if (e >= n_all) break;
        float s = 0.f;
        if (s_row[le] >= 0) {
            bool first = true;
            for (int q = 0; q < k.world; ++q) {                  // rank order; the first holder is this entry itself
                const int32_t pos = s_pos[le][q];
                if (pos >= 0) {
                    const float v = grads[q * k.total + static_cast<int64_t>(pos) * dim + c];
                    s = first ? v : s + v;
                    first = false;
                }
            }
            s *= k.scale;
        }
        T.out_grad[e * dim + c] = s;
    }
}

extern "C" int swr_dp_finish(const float* recv_dense, int64_t dense_stride, int64_t A, float* dense_out,
                             const float* recv_rows, int64_t rows_stride, const swr_dp_table* tables, int n_tables,
                             int world, float scale, void* stream) {
    SWR_REQUIRE(world >= 1 && world <= SWR_DP_MAX_WORLD && A >= 0, SWR_ERR_ARG);
    SWR_REQUIRE(n_tables >= 0 && n_tables <= SWR_DP_MAX_TABLES && (n_tables == 0 || (tables && recv_rows && rows_stride > 0)),
                SWR_ERR_ARG);
    SWR_REQUIRE(A == 0 || (dense_out && recv_dense && dense_stride >= A), SWR_ERR_ARG);
    SWR_REQUIRE(A == 0 || (dense_stride % 4 == 0 && swr_aligned16(recv_dense) && swr_aligned16(dense_out)), SWR_ERR_ALIGN);
    const int64_t total = rows_stride;
    DpK k;
    k.recv_d = recv_dense; k.recv = recv_rows; k.world = world; k.dstride = dense_stride; k.total = total; k.A = A;
    k.dense_out = dense_out; k.scale = scale;
    k.n_tables = n_tables;
    k.dense_blocks = static_cast<int>(swr_ceil_div(A, DPF_THREADS * 4));
    int64_t blocks = 0;
    for (int t = 0; t < n_tables; ++t) {
        const swr_dp_table& T = tables[t];
        SWR_REQUIRE(T.n >= 0 && T.dim > 0 && T.out_row && T.out_grad && T.row_off >= 0 && T.row_off + T.n <= total &&
                    T.grad_off >= 0 && T.grad_off + T.n * T.dim <= total && T.n * world < (1ll << 31), SWR_ERR_ARG);
        k.tab[t] = T;
        k.block0[t] = static_cast<int>(blocks);
        blocks += swr_ceil_div(T.n * world, DPF_THREADS);
    }
    k.block0[n_tables] = static_cast<int>(blocks);
    const int64_t grid = k.dense_blocks + blocks;
    if (grid == 0) return SWR_OK;
    SWR_REQUIRE(grid < (1ll << 31), SWR_ERR_ARG);
    hipLaunchKernelGGL(dp_finish_kernel, dim3(static_cast<unsigned>(grid)), dim3(DPF_THREADS), 0,
                       static_cast<hipStream_t>(stream), k);
    return swr_launch_status();
}
