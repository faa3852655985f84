// BatchNorm1d statistics / affine + activation forward and backward (basic/layers.py:253-258).
//
// Forward statistics arrive as per-32-row-tile (mean, M2) pairs from the producing GEMM's epilogue
// and are merged here with Chan's formula in fp64 in a fixed tree order (deterministic).  Backward
// statistics (sum dA, sum dA * xhat) are reduced per 64-row tile by `bn_act_bwd_stats` and summed in
// fp64.  All elementwise passes are HBM-bound streaming kernels over [M, N] fp32 row-major data.
#include "common.h"

#define BN_THREADS 256

struct ActSpec {
    swr_act_range r[SWR_MAX_ACT_RANGES];
    int n;
};

static int make_acts(const swr_act_range* acts, int n_acts, int N, ActSpec& out) {
    SWR_REQUIRE(n_acts >= 0 && n_acts <= SWR_MAX_ACT_RANGES && (n_acts == 0 || acts), SWR_ERR_ARG);
    out.n = n_acts;
    for (int i = 0; i < n_acts; ++i) {
        out.r[i] = acts[i];
        SWR_REQUIRE(acts[i].col_lo >= 0 && acts[i].col_hi <= N && acts[i].col_lo <= acts[i].col_hi, SWR_ERR_ARG);
        SWR_REQUIRE(acts[i].act >= SWR_ACT_NONE && acts[i].act <= SWR_ACT_SOFTMAX, SWR_ERR_ARG);
        if (acts[i].act == SWR_ACT_SOFTMAX)
            SWR_REQUIRE(acts[i].group > 0 && (acts[i].col_hi - acts[i].col_lo) % acts[i].group == 0, SWR_ERR_ARG);
    }
    return SWR_OK;
}

__device__ __forceinline__ int find_act(const ActSpec& a, int n, int& lo, int& group) {
    for (int i = 0; i < a.n; ++i)
        if (n >= a.r[i].col_lo && n < a.r[i].col_hi) {
            lo = a.r[i].col_lo;
            group = a.r[i].group;
            return a.r[i].act;
        }
    lo = 0;
    group = 1;
    return SWR_ACT_NONE;
}

#define BWD_TILE 64

// =========================================================================================== fast paths
// Rows are 16-byte aligned multiples of 4 floats and activation ranges do not cut a float4 (softmax groups are
// exactly one float4): every thread owns ONE float4 column slot for all the rows it visits, so the per-column
// coefficients are loaded once per thread and every access is a 16-byte load / store.
struct V4Plan {
    int vpr;        // float4 per row
    int rows;       // rows covered by one pass of the 256-thread block
};

// tail: activation ranges the float4 kernels cannot take (softmax groups other than one float4, ranges that cut a float4) are
// allowed at the END of the row -- PLE's gates are softmax groups of n_expert_specific + n_expert_shared = 3 columns behind
// 9 x 64 expert columns (ple.py:89-94), and thread-per-column kernels ran that layer at 30 us per pass at M = 8192, every
// workgroup waiting for its softmax lanes.  Columns [0, n4) go to the float4 threads, the <= TAIL_MAX columns [n4, N) to
// extra workgroups of the same launch, thread = (row group, column), TAIL_U rows in flight.
#define TAIL_MAX 64
#ifndef TAIL_U
#define TAIL_U 4
#endif

static bool v4_split(const ActSpec& as, int N, int64_t ld0, int64_t ld1, int64_t ld2, int64_t ld3, const void* p0, const void* p1,
                     const void* p2, const void* p3, const void* c0, const void* c1, const void* c2, const void* c3, int& n4) {
    n4 = 0;
    if (N % 4 != 0) return false;
    const int64_t lds[4] = {ld0, ld1, ld2, ld3};
    for (int i = 0; i < 4; ++i)
        if (lds[i] % 4 != 0) return false;
    const void* ps[8] = {p0, p1, p2, p3, c0, c1, c2, c3};
    for (int i = 0; i < 8; ++i)
        if (ps[i] && !swr_aligned16(ps[i])) return false;
    int bad = N;
    for (int i = 0; i < as.n; ++i) {
        const bool ok = as.r[i].col_lo % 4 == 0 && as.r[i].col_hi % 4 == 0 && (as.r[i].act != SWR_ACT_SOFTMAX || as.r[i].group == 4);
        if (!ok && as.r[i].col_lo < as.r[i].col_hi) bad = std::min(bad, as.r[i].col_lo & ~3);
    }
    if (bad < 4 || N - bad > TAIL_MAX || bad / 4 > BN_THREADS) return false;
    n4 = bad;
    return true;
}

__device__ __forceinline__ int act_of_col(const ActSpec& a, int n) {
    for (int i = 0; i < a.n; ++i)
        if (n >= a.r[i].col_lo && n < a.r[i].col_hi) return a.r[i].act;
    return SWR_ACT_NONE;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4_or(const float* p, int n, float dflt) { return p ? ld4(p + n) : make_float4(dflt, dflt, dflt, dflt); }

__device__ __forceinline__ float4 act_fwd4(int act, float4 v) {
    if (act == SWR_ACT_RELU) return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    if (act == SWR_ACT_SIGMOID) return make_float4(swr_sigmoid(v.x), swr_sigmoid(v.y), swr_sigmoid(v.z), swr_sigmoid(v.w));
    if (act == SWR_ACT_SOFTMAX) {
        const float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
        const float ex = expf(v.x - mx), ey = expf(v.y - mx), ez = expf(v.z - mx), ew = expf(v.w - mx);
        const float den = ex + ey + ez + ew;        // same left-to-right order as the scalar kernel
        return make_float4(ex / den, ey / den, ez / den, ew / den);
    }
    return v;
}

__device__ __forceinline__ float4 act_bwd4(int act, float4 dy, float4 y) {
    if (act == SWR_ACT_RELU) return make_float4(y.x > 0.f ? dy.x : 0.f, y.y > 0.f ? dy.y : 0.f, y.z > 0.f ? dy.z : 0.f, y.w > 0.f ? dy.w : 0.f);
    if (act == SWR_ACT_SIGMOID)
        return make_float4(dy.x * y.x * (1.f - y.x), dy.y * y.y * (1.f - y.y), dy.z * y.z * (1.f - y.z), dy.w * y.w * (1.f - y.w));
    if (act == SWR_ACT_SOFTMAX) {
        float dot = 0.f;
        dot = fmaf(dy.x, y.x, dot); dot = fmaf(dy.y, y.y, dot); dot = fmaf(dy.z, y.z, dot); dot = fmaf(dy.w, y.w, dot);
        return make_float4(y.x * (dy.x - dot), y.y * (dy.y - dot), y.z * (dy.z - dot), y.w * (dy.w - dot));
    }
    return dy;
}

// ---- thread-per-column row helpers: U rows of column n in flight (rows m0 .. m0 + cnt - 1, cnt <= U; the loads of a missing row
// repeat the last one).  A softmax column walks its group once for the maxima and once for the sums with the loads of all U rows
// issued together -- one row at a time each of the 2 x group loads waited for the previous one.  Operation order per row = the
// float4 kernels' (left to right over the group).
template <int U>
__device__ __forceinline__ void affine_act_fwd_rows(const float* __restrict__ Z, int64_t ldz, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, int act, int lo, int group,
                                                    float* __restrict__ Y, int64_t ldy, int n, int64_t m0, int cnt) {
    const float sc = scale ? scale[n] : 1.f, sh = shift ? shift[n] : 0.f;
    const float* z[U];
#pragma unroll
    for (int u = 0; u < U; ++u) z[u] = Z + (m0 + min(u, cnt - 1)) * ldz;
    float y[U];
    if (act == SWR_ACT_SOFTMAX) {
        const int g0 = lo + ((n - lo) / group) * group;
        float mx[U], den[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { mx[u] = -INFINITY; den[u] = 0.f; }
        for (int j = 0; j < group; ++j) {
            const float sj = scale ? scale[g0 + j] : 1.f, hj = shift ? shift[g0 + j] : 0.f;
#pragma unroll
            for (int u = 0; u < U; ++u) mx[u] = fmaxf(mx[u], sj * z[u][g0 + j] + hj);
        }
        for (int j = 0; j < group; ++j) {
            const float sj = scale ? scale[g0 + j] : 1.f, hj = shift ? shift[g0 + j] : 0.f;
#pragma unroll
            for (int u = 0; u < U; ++u) den[u] += expf(sj * z[u][g0 + j] + hj - mx[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) y[u] = expf(sc * z[u][n] + sh - mx[u]) / den[u];
    } else {
        float zv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) zv[u] = z[u][n];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float v = sc * zv[u] + sh;
            y[u] = act == SWR_ACT_RELU ? fmaxf(v, 0.f) : (act == SWR_ACT_SIGMOID ? swr_sigmoid(v) : v);
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (u < cnt) Y[(m0 + u) * ldy + n] = y[u];
}

// dA = act'(Y) * dY of column n for U rows (row pointers given; a softmax column needs dot(dY, Y) over its group)
template <int U>
__device__ __forceinline__ void act_grad_rows(int act, int lo, int group, const float* const (&dy)[U], const float* const (&y)[U], int n,
                                              float (&da)[U]) {
    float g[U], yv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        g[u] = dy[u][n];
        yv[u] = act == SWR_ACT_NONE ? 0.f : y[u][n];
    }
    if (act == SWR_ACT_SOFTMAX) {
        const int g0 = lo + ((n - lo) / group) * group;
        float dot[U];
#pragma unroll
        for (int u = 0; u < U; ++u) dot[u] = 0.f;
        for (int j = 0; j < group; ++j) {
#pragma unroll
            for (int u = 0; u < U; ++u) dot[u] = fmaf(dy[u][g0 + j], y[u][g0 + j], dot[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) da[u] = yv[u] * (g[u] - dot[u]);
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u)
            da[u] = act == SWR_ACT_RELU ? (yv[u] > 0.f ? g[u] : 0.f) : (act == SWR_ACT_SIGMOID ? g[u] * yv[u] * (1.f - yv[u]) : g[u]);
    }
}

template <int U>
__device__ __forceinline__ void act_bwd_apply_rows(const float* __restrict__ dY, int64_t lddy, const float* __restrict__ Y, int64_t ldy,
                                                   const float* __restrict__ Z, int64_t ldz, const float* __restrict__ ca,
                                                   const float* __restrict__ cb, const float* __restrict__ cc,
                                                   const float* __restrict__ mean, int act, int lo, int group,
                                                   float* __restrict__ dZ, int64_t lddz, int n, int64_t m0, int cnt) {
    const float a_ = ca ? ca[n] : 1.f;
    const float b_ = cb ? cb[n] : 0.f, c_ = cb ? cc[n] : 0.f, mu = cb ? mean[n] : 0.f;
    const float* dy[U];
    const float* y[U];
    float z[U], da[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t m = m0 + min(u, cnt - 1);
        dy[u] = dY + m * lddy;
        y[u] = Y + m * ldy;
        z[u] = cb ? Z[m * ldz + n] : 0.f;
    }
    act_grad_rows<U>(act, lo, group, dy, y, n, da);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        float v = da[u];
        if (ca) v *= a_;
        if (cb) v = fmaf(b_, z[u] - mu, v) + c_;
        if (u < cnt) dZ[(m0 + u) * lddz + n] = v;
    }
}

// the tail columns [n4, N) of a float4 launch: thread = (row group, column), TAIL_U consecutive rows each
struct TailMap {
    int n;           // column, -1: thread idle
    int64_t m0;      // first row
    int cnt;         // rows (<= TAIL_U)
};
__device__ __forceinline__ TailMap tail_map(int tail_block, int n4, int N, int64_t M) {
    const int tc = N - n4, groups = BN_THREADS / tc;
    const int rg = threadIdx.x / tc, c = threadIdx.x - rg * tc;
    TailMap t;
    t.m0 = (static_cast<int64_t>(tail_block) * groups + rg) * TAIL_U;
    t.n = (rg < groups && t.m0 < M) ? n4 + c : -1;
    t.cnt = static_cast<int>(min<int64_t>(TAIL_U, M - t.m0));
    return t;
}
static inline unsigned tail_blocks(int n4, int N, int64_t M) {
    return n4 < N ? static_cast<unsigned>(swr_ceil_div(M, static_cast<int64_t>(BN_THREADS / (N - n4)) * TAIL_U)) : 0u;
}

#define V4_ITERS 8
__global__ __launch_bounds__(BN_THREADS) void affine_act_fwd_v4_kernel(const float* __restrict__ Z, int64_t ldz,
                                                                       const float* __restrict__ scale,
                                                                       const float* __restrict__ shift, const ActSpec acts,
                                                                       float* __restrict__ Y, int64_t ldy, int64_t M, int N,
                                                                       const V4Plan pl, int v4_blocks) {
    if (static_cast<int>(blockIdx.x) >= v4_blocks) {          // tail columns [4 vpr, N)
        const TailMap t = tail_map(blockIdx.x - v4_blocks, 4 * pl.vpr, N, M);
        if (t.n < 0) return;
        int lo, group;
        const int act = find_act(acts, t.n, lo, group);
        affine_act_fwd_rows<TAIL_U>(Z, ldz, scale, shift, act, lo, group, Y, ldy, t.n, t.m0, t.cnt);
        return;
    }
    const int r_in = threadIdx.x / pl.vpr, v = threadIdx.x - r_in * pl.vpr;
    if (r_in >= pl.rows) return;
    const int n = 4 * v;
    const int act = act_of_col(acts, n);
    const float4 sc = ld4_or(scale, n, 1.f), sh = ld4_or(shift, n, 0.f);
    const int64_t m0 = static_cast<int64_t>(blockIdx.x) * pl.rows * V4_ITERS + r_in;
#pragma unroll
    for (int it = 0; it < V4_ITERS; ++it) {
        const int64_t m = m0 + static_cast<int64_t>(it) * pl.rows;
        if (m < M) {
            const float4 z = ld4(Z + m * ldz + n);
            const float4 a = make_float4(sc.x * z.x + sh.x, sc.y * z.y + sh.y, sc.z * z.z + sh.z, sc.w * z.w + sh.w);
            *reinterpret_cast<float4*>(Y + m * ldy + n) = act_fwd4(act, a);
        }
    }
}

__global__ __launch_bounds__(BN_THREADS) void act_bwd_apply_v4_kernel(
    const float* __restrict__ dY, int64_t lddy, const float* __restrict__ Y, int64_t ldy, const float* __restrict__ Z,
    int64_t ldz, const float* __restrict__ ca, const float* __restrict__ cb, const float* __restrict__ cc,
    const float* __restrict__ mean, const ActSpec acts, float* __restrict__ dZ, int64_t lddz, int64_t M, int N, const V4Plan pl,
    int v4_blocks) {
    if (static_cast<int>(blockIdx.x) >= v4_blocks) {          // tail columns [4 vpr, N)
        const TailMap t = tail_map(blockIdx.x - v4_blocks, 4 * pl.vpr, N, M);
        if (t.n < 0) return;
        int lo, group;
        const int act = find_act(acts, t.n, lo, group);
        act_bwd_apply_rows<TAIL_U>(dY, lddy, Y, ldy, Z, ldz, ca, cb, cc, mean, act, lo, group, dZ, lddz, t.n, t.m0, t.cnt);
        return;
    }
    const int r_in = threadIdx.x / pl.vpr, v = threadIdx.x - r_in * pl.vpr;
    if (r_in >= pl.rows) return;
    const int n = 4 * v;
    const int act = act_of_col(acts, n);
    const float4 a4 = ld4_or(ca, n, 1.f), b4 = ld4_or(cb, n, 0.f), c4 = ld4_or(cc, n, 0.f), mu = ld4_or(mean, n, 0.f);
    const bool has_b = cb != nullptr;
    const int64_t m0 = static_cast<int64_t>(blockIdx.x) * pl.rows * V4_ITERS + r_in;
#pragma unroll
    for (int it = 0; it < V4_ITERS; ++it) {
        const int64_t m = m0 + static_cast<int64_t>(it) * pl.rows;
        if (m < M) {
            float4 g = act_bwd4(act, ld4(dY + m * lddy + n), ld4(Y + m * ldy + n));
            g.x *= a4.x; g.y *= a4.y; g.z *= a4.z; g.w *= a4.w;
            if (has_b) {
                const float4 z = ld4(Z + m * ldz + n);
                g.x = fmaf(b4.x, z.x - mu.x, g.x) + c4.x; g.y = fmaf(b4.y, z.y - mu.y, g.y) + c4.y;
                g.z = fmaf(b4.z, z.z - mu.z, g.z) + c4.z; g.w = fmaf(b4.w, z.w - mu.w, g.w) + c4.w;
            }
            *reinterpret_cast<float4*>(dZ + m * lddz + n) = g;
        }
    }
}

// (sum dA, sum dA * xhat) of column n over the rows ry, ry + 4, ... of the 64-row tile at m0: TAIL_U rows in flight
__device__ __forceinline__ void bn_act_bwd_stats_col(const float* __restrict__ dY, int64_t lddy, const float* __restrict__ Y, int64_t ldy,
                                                     const float* __restrict__ Z, int64_t ldz, float mu, float rs, int act, int lo,
                                                     int group, int n, int ry, int64_t m0, int64_t M, float& a1, float& a2) {
    const int rows = static_cast<int>(min<int64_t>(BWD_TILE, M - m0));
    for (int r = ry; r < rows; r += 4 * TAIL_U) {
        const float* dy[TAIL_U];
        const float* y[TAIL_U];
        float z[TAIL_U], da[TAIL_U];
#pragma unroll
        for (int u = 0; u < TAIL_U; ++u) {
            const int64_t m = m0 + min(r + 4 * u, rows - 1);
            dy[u] = dY + m * lddy;
            y[u] = Y + m * ldy;
            z[u] = Z[m * ldz + n];
        }
        act_grad_rows<TAIL_U>(act, lo, group, dy, y, n, da);
#pragma unroll
        for (int u = 0; u < TAIL_U; ++u) {
            if (r + 4 * u < rows) {
                a1 += da[u];
                a2 = fmaf(da[u], (z[u] - mu) * rs, a2);
            }
        }
    }
}

// one workgroup per (64-row tile, block of <= 64 float4 columns) (the tile layout swr_bn_bwd_finalize expects); threads
// with the same column slot are summed through LDS in fixed order.  Wide layers are cut into column blocks: with one
// workgroup per row tile, N = 768 left 192 threads walking 64 rows each from 256 workgroups -- 75 us for 150 MB at
// M = 16 384 (STAR); four rows are loaded before any is used
__global__ __launch_bounds__(BN_THREADS) void bn_act_bwd_stats_v4_kernel(
    const float* __restrict__ dY, int64_t lddy, const float* __restrict__ Y, int64_t ldy, const float* __restrict__ Z,
    int64_t ldz, const float* __restrict__ mean, const float* __restrict__ rstd, const ActSpec acts,
    float* __restrict__ partials, int64_t M, int N, const V4Plan pl, int vpb, int v4_yblocks) {
    __shared__ float4 s1[BN_THREADS], s2[BN_THREADS];
    if (static_cast<int>(blockIdx.y) >= v4_yblocks) {          // tail columns [4 vpr, N): thread = (column, row phase)
        float* t1 = reinterpret_cast<float*>(s1);
        float* t2 = reinterpret_cast<float*>(s2);
        const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
        const int n = 4 * pl.vpr + cx;
        float a1 = 0.f, a2 = 0.f;
        if (n < N) {
            int lo, group;
            const int act = find_act(acts, n, lo, group);
            bn_act_bwd_stats_col(dY, lddy, Y, ldy, Z, ldz, mean[n], rstd[n], act, lo, group, n, ry, static_cast<int64_t>(blockIdx.x) * BWD_TILE,
                                 M, a1, a2);
        }
        t1[ry * 64 + cx] = a1;
        t2[ry * 64 + cx] = a2;
        __syncthreads();
        if (ry == 0 && n < N) {
            float* p = partials + (static_cast<int64_t>(blockIdx.x) * N + n) * 2;
            p[0] = (t1[cx] + t1[64 + cx]) + (t1[128 + cx] + t1[192 + cx]);
            p[1] = (t2[cx] + t2[64 + cx]) + (t2[128 + cx] + t2[192 + cx]);
        }
        return;
    }
    const int r_in = threadIdx.x / vpb, vl = threadIdx.x - r_in * vpb;
    const int rows = BN_THREADS / vpb;
    const int v = blockIdx.y * vpb + vl;
    const bool active = r_in < rows && v < pl.vpr;
    const int n = 4 * v;
    float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a1;
    if (active) {
        const int act = act_of_col(acts, n);
        const float4 mu = ld4(mean + n), rs = ld4(rstd + n);
        const int64_t m0 = static_cast<int64_t>(blockIdx.x) * BWD_TILE;
        const int64_t mend = min<int64_t>(m0 + BWD_TILE, M);
        constexpr int U = 4;
        for (int64_t mb = m0 + r_in; mb < mend; mb += static_cast<int64_t>(U) * rows) {
            float4 d[U], y[U], z[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t m = min<int64_t>(mb + static_cast<int64_t>(u) * rows, mend - 1);
                d[u] = ld4(dY + m * lddy + n); y[u] = ld4(Y + m * ldy + n); z[u] = ld4(Z + m * ldz + n);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (mb + static_cast<int64_t>(u) * rows < mend) {
                    const float4 g = act_bwd4(act, d[u], y[u]);
                    a1.x += g.x; a1.y += g.y; a1.z += g.z; a1.w += g.w;
                    a2.x = fmaf(g.x, (z[u].x - mu.x) * rs.x, a2.x); a2.y = fmaf(g.y, (z[u].y - mu.y) * rs.y, a2.y);
                    a2.z = fmaf(g.z, (z[u].z - mu.z) * rs.z, a2.z); a2.w = fmaf(g.w, (z[u].w - mu.w) * rs.w, a2.w);
                }
            }
        }
    }
    s1[threadIdx.x] = a1;
    s2[threadIdx.x] = a2;
    __syncthreads();
    if (active && r_in == 0) {
        for (int r = 1; r < rows; ++r) {
            const float4 b1 = s1[r * vpb + vl], b2 = s2[r * vpb + vl];
            a1.x += b1.x; a1.y += b1.y; a1.z += b1.z; a1.w += b1.w;
            a2.x += b2.x; a2.y += b2.y; a2.z += b2.z; a2.w += b2.w;
        }
        float* p = partials + (static_cast<int64_t>(blockIdx.x) * N + n) * 2;
        *reinterpret_cast<float4*>(p) = make_float4(a1.x, a2.x, a1.y, a2.y);
        *reinterpret_cast<float4*>(p + 4) = make_float4(a1.z, a2.z, a1.w, a2.w);
    }
}


// ---------------------------------------------------------------------------------- forward stats
// Two fixed-order fp64 passes over the tiles of one column: mean = sum(n_t mean_t) / n, then
// M2 = sum(M2_t + n_t (mean_t - mean)^2) -- the pairwise (Chan) combination written as two plain sums, so the
// block sums are additions only (a tree of fp64 divisions made this small kernel take 10 us); the tile statistics stay
// in registers between the passes and the sums are wave butterflies + one LDS exchange (swr_block_sum_f64).
__global__ __launch_bounds__(BN_THREADS) void bn_finalize_kernel(
    const float* __restrict__ part, int n_tiles, int64_t M, int N, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float momentum, float* running_mean, float* running_var,
    int64_t* nbt, int n_tracked, float* mean_o, float* rstd_o, float* scale_o, float* shift_o) {
    __shared__ double sm1[BN_THREADS / 64], sm2[BN_THREADS / 64];
    constexpr int KEEP = 8;                                   // tiles per thread held in registers (B <= 65 536)
    const int n = blockIdx.x;
    const int per = (n_tiles + BN_THREADS - 1) / BN_THREADS;
    const int t0 = threadIdx.x * per, t1 = min(t0 + per, n_tiles);
    const bool keep = per <= KEEP;
    float2 pv[KEEP];
    if (keep) {                                               // one trip to memory for both passes
#pragma unroll
        for (int u = 0; u < KEEP; ++u)
            pv[u] = t0 + u < t1 ? *reinterpret_cast<const float2*>(part + (static_cast<int64_t>(t0 + u) * N + n) * 2)
                                : make_float2(0.f, 0.f);
    }
    auto tile_rows = [&](int t) { return static_cast<double>(min<int64_t>(32, M - static_cast<int64_t>(t) * 32)); };
    double acc = 0.0;
    if (keep) {
#pragma unroll
        for (int u = 0; u < KEEP; ++u)
            if (t0 + u < t1) acc += tile_rows(t0 + u) * static_cast<double>(pv[u].x);
    } else {
        for (int t = t0; t < t1; ++t) acc += tile_rows(t) * static_cast<double>(part[(static_cast<int64_t>(t) * N + n) * 2]);
    }
    const double mu = swr_block_sum_f64<BN_THREADS>(acc, sm1) / static_cast<double>(M);
    acc = 0.0;
    if (keep) {
#pragma unroll
        for (int u = 0; u < KEEP; ++u) {
            if (t0 + u < t1) {
                const double d = static_cast<double>(pv[u].x) - mu;
                acc += static_cast<double>(pv[u].y) + tile_rows(t0 + u) * d * d;
            }
        }
    } else {
        for (int t = t0; t < t1; ++t) {
            const float* p = part + (static_cast<int64_t>(t) * N + n) * 2;
            const double d = static_cast<double>(p[0]) - mu;
            acc += static_cast<double>(p[1]) + tile_rows(t) * d * d;
        }
    }
    const double m2 = swr_block_sum_f64<BN_THREADS>(acc, sm2);
    if (threadIdx.x == 0) {
        const double cnt = static_cast<double>(M);
        const double var_b = m2 / cnt;
        const float mean = static_cast<float>(mu);
        const float rstd = static_cast<float>(1.0 / sqrt(var_b + static_cast<double>(eps)));
        const float g = gamma ? gamma[n] : 1.f, b = beta ? beta[n] : 0.f;
        const float scale = g * rstd;
        if (mean_o) mean_o[n] = mean;
        if (rstd_o) rstd_o[n] = rstd;
        if (scale_o) scale_o[n] = scale;
        if (shift_o) shift_o[n] = b - mean * scale;
        if (running_mean) running_mean[n] = (1.f - momentum) * running_mean[n] + momentum * mean;
        if (running_var) {
            const float var_u = static_cast<float>(cnt > 1.0 ? m2 / (cnt - 1.0) : var_b);
            running_var[n] = (1.f - momentum) * running_var[n] + momentum * var_u;
        }
        if (nbt && n < n_tracked) nbt[n] += 1;
    }
}

extern "C" int swr_bn_finalize(const float* stat_partials, int n_tiles, int64_t M, int N, const float* gamma,
                               const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                               int64_t* num_batches_tracked, int n_tracked, float* mean, float* rstd, float* scale,
                               float* shift, void* stream) {
    SWR_REQUIRE(stat_partials && n_tiles > 0 && M > 0 && N > 0, SWR_ERR_ARG);
    SWR_REQUIRE(n_tiles == swr_ceil_div(M, 32) && n_tracked >= 0 && n_tracked <= N, SWR_ERR_ARG);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(N), dim3(BN_THREADS), 0, static_cast<hipStream_t>(stream), stat_partials,
                       n_tiles, M, N, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, n_tracked,
                       mean, rstd, scale, shift);
    return swr_launch_status();
}

// (mean, M2) per 32-row tile and column of an arbitrary [M, N] tensor: workgroup = 8 tiles (256 rows) x 64 columns;
// a thread owns one column of one tile half (16 rows, two passes in registers), halves combined by Chan's formula
__global__ __launch_bounds__(BN_THREADS) void col_moments_kernel(const float* __restrict__ X, int64_t ldx, int64_t M, int N,
                                                                 float* __restrict__ partials) {
    __shared__ float s_sum[4][64], s_m2[4][64];
    const int cx = threadIdx.x & 63, part = threadIdx.x >> 6;          // 4 row quarters of a 32-row tile
    const int n = blockIdx.y * 64 + cx;
    for (int tl = 0; tl < 8; ++tl) {
        const int64_t tile = static_cast<int64_t>(blockIdx.x) * 8 + tl;
        const int64_t m0 = tile * 32;
        if (m0 >= M) break;                                             // uniform
        const int rows = static_cast<int>(min<int64_t>(32, M - m0));
        float v[8];
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int row = part * 8 + r;
            v[r] = (n < N && row < rows) ? X[(m0 + row) * ldx + n] : 0.f;
            sum += v[r];
        }
        s_sum[part][cx] = sum;
        __syncthreads();
        const float mean = ((s_sum[0][cx] + s_sum[1][cx]) + (s_sum[2][cx] + s_sum[3][cx])) / static_cast<float>(rows);
        float m2 = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float d = v[r] - mean;
            if (part * 8 + r < rows) m2 = fmaf(d, d, m2);
        }
        s_m2[part][cx] = m2;
        __syncthreads();
        if (part == 0 && n < N) {
            float* p = partials + (tile * N + n) * 2;
            p[0] = mean;
            p[1] = (s_m2[0][cx] + s_m2[1][cx]) + (s_m2[2][cx] + s_m2[3][cx]);
        }
        __syncthreads();
    }
}

extern "C" int swr_col_moments(const float* X, int64_t ldx, int64_t M, int N, float* stat_partials, void* stream) {
    SWR_REQUIRE(X && stat_partials && M > 0 && N > 0 && ldx >= N, SWR_ERR_ARG);
    const dim3 grid(static_cast<unsigned>(swr_ceil_div(swr_ceil_div(M, 32), 8)), static_cast<unsigned>(swr_ceil_div(N, 64)));
    hipLaunchKernelGGL(col_moments_kernel, grid, dim3(BN_THREADS), 0, static_cast<hipStream_t>(stream), X, ldx, M, N,
                       stat_partials);
    return swr_launch_status();
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      int N, float* scale, float* shift) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float sc = (gamma ? gamma[n] : 1.f) / sqrtf(rv[n] + eps);
    scale[n] = sc;
    shift[n] = (beta ? beta[n] : 0.f) - rm[n] * sc;
}

extern "C" int swr_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, int N, float* scale, float* shift, void* stream) {
    SWR_REQUIRE(running_mean && running_var && scale && shift && N > 0, SWR_ERR_ARG);
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((N + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), gamma,
                       beta, running_mean, running_var, eps, N, scale, shift);
    return swr_launch_status();
}

// ------------------------------------------------------------------------------ affine + act forward
// general-width path (N not a multiple of 4, or wider than 1024): thread = one column, EW_ROWS rows of it -- no
// per-element division by N, the column's coefficients and activation looked up once
#define EW_ROWS 16
#define EW_UNROLL 4
__global__ __launch_bounds__(BN_THREADS) void affine_act_fwd_kernel(const float* __restrict__ Z, int64_t ldz,
                                                                    const float* __restrict__ scale,
                                                                    const float* __restrict__ shift, const ActSpec acts,
                                                                    float* __restrict__ Y, int64_t ldy, int64_t M, int N,
                                                                    int rows_per_block) {
    const int n = blockIdx.x * BN_THREADS + threadIdx.x;
    if (n >= N) return;
    int lo, group;
    const int act = find_act(acts, n, lo, group);
    const int64_t m0 = static_cast<int64_t>(blockIdx.y) * rows_per_block, m1 = min<int64_t>(m0 + rows_per_block, M);
    // four rows in flight per thread (a 4-byte load per row and lane needs several outstanding to cover the HBM latency; one
    // row at a time ran at 2.8 TB/s on HAMUR's [32 768, 1 225] hyper-net output)
    for (int64_t m = m0; m < m1; m += EW_UNROLL)
        affine_act_fwd_rows<EW_UNROLL>(Z, ldz, scale, shift, act, lo, group, Y, ldy, n, m, static_cast<int>(min<int64_t>(EW_UNROLL, m1 - m)));
}

extern "C" int swr_affine_act_fwd(const float* Z, int64_t ldz, const float* scale, const float* shift,
                                  const swr_act_range* acts, int n_acts, float* Y, int64_t ldy, int64_t M, int N,
                                  void* stream) {
    SWR_REQUIRE(Z && Y && M >= 0 && N > 0 && ldz >= N && ldy >= N, SWR_ERR_ARG);
    ActSpec as;
    const int rc = make_acts(acts, n_acts, N, as);
    if (rc != SWR_OK) return rc;
    if (M == 0) return SWR_OK;
    int n4;
    if (v4_split(as, N, ldz, ldy, 4, 4, Z, Y, nullptr, nullptr, scale, shift, nullptr, nullptr, n4)) {
        V4Plan pl;
        pl.vpr = n4 / 4;
        pl.rows = BN_THREADS / pl.vpr;
        const unsigned v4_blocks = static_cast<unsigned>(swr_ceil_div(M, pl.rows * V4_ITERS));
        hipLaunchKernelGGL(affine_act_fwd_v4_kernel, dim3(v4_blocks + tail_blocks(n4, N, M)), dim3(BN_THREADS), 0,
                           static_cast<hipStream_t>(stream), Z, ldz, scale, shift, as, Y, ldy, M, N, pl, static_cast<int>(v4_blocks));
        return swr_launch_status();
    }
    const int rpb = static_cast<int>(std::max<int64_t>(EW_ROWS, swr_ceil_div(M, 65535)));       // gridDim.y <= 65535
    hipLaunchKernelGGL(affine_act_fwd_kernel, dim3(static_cast<unsigned>(swr_ceil_div(N, BN_THREADS)), static_cast<unsigned>(swr_ceil_div(M, rpb))),
                       dim3(BN_THREADS), 0, static_cast<hipStream_t>(stream), Z, ldz, scale, shift, as, Y, ldy, M, N, rpb);
    return swr_launch_status();
}

// --------------------------------------------------------------------------------------- backward
// block = 64 columns x 4 row phases over a 64-row tile; partials[tile][n] = (sum dA, sum dA * xhat)
__global__ __launch_bounds__(BN_THREADS) void bn_act_bwd_stats_kernel(
    const float* __restrict__ dY, int64_t lddy, const float* __restrict__ Y, int64_t ldy, const float* __restrict__ Z,
    int64_t ldz, const float* __restrict__ mean, const float* __restrict__ rstd, const ActSpec acts,
    float* __restrict__ partials, int64_t M, int N) {
    __shared__ float s1[4][64], s2[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int n = blockIdx.y * 64 + cx;
    const int64_t m0 = static_cast<int64_t>(blockIdx.x) * BWD_TILE;
    float a1 = 0.f, a2 = 0.f;
    if (n < N) {
        int lo_, group_;
        const int act = find_act(acts, n, lo_, group_);
        bn_act_bwd_stats_col(dY, lddy, Y, ldy, Z, ldz, mean[n], rstd[n], act, lo_, group_, n, ry, m0, M, a1, a2);
    }
    s1[ry][cx] = a1;
    s2[ry][cx] = a2;
    __syncthreads();
    if (ry == 0 && n < N) {
        float* p = partials + (static_cast<int64_t>(blockIdx.x) * N + n) * 2;
        p[0] = (s1[0][cx] + s1[1][cx]) + (s1[2][cx] + s1[3][cx]);
        p[1] = (s2[0][cx] + s2[1][cx]) + (s2[2][cx] + s2[3][cx]);
    }
}

extern "C" int swr_bn_act_bwd_stats(const float* dY, int64_t lddy, const float* Y, int64_t ldy, const float* Z,
                                    int64_t ldz, const float* mean, const float* rstd, const swr_act_range* acts,
                                    int n_acts, float* partials, int64_t M, int N, void* stream) {
    SWR_REQUIRE(dY && Y && Z && mean && rstd && partials && M > 0 && N > 0, SWR_ERR_ARG);
    ActSpec as;
    const int rc = make_acts(acts, n_acts, N, as);
    if (rc != SWR_OK) return rc;
    int n4;
    if (v4_split(as, N, lddy, ldy, ldz, 4, dY, Y, Z, partials, mean, rstd, nullptr, nullptr, n4)) {
        V4Plan pl;
        pl.vpr = n4 / 4;
        pl.rows = BN_THREADS / pl.vpr;
        const int vpb = pl.vpr < 64 ? pl.vpr : 64;
        const unsigned v4_y = static_cast<unsigned>(swr_ceil_div(pl.vpr, vpb));
        hipLaunchKernelGGL(bn_act_bwd_stats_v4_kernel, dim3(static_cast<unsigned>(swr_ceil_div(M, BWD_TILE)), v4_y + (n4 < N ? 1u : 0u)),
                           dim3(BN_THREADS), 0, static_cast<hipStream_t>(stream), dY, lddy, Y, ldy, Z, ldz, mean, rstd, as, partials,
                           M, N, pl, vpb, static_cast<int>(v4_y));
        return swr_launch_status();
    }
    const dim3 grid(static_cast<unsigned>(swr_ceil_div(M, BWD_TILE)), static_cast<unsigned>(swr_ceil_div(N, 64)));
    hipLaunchKernelGGL(bn_act_bwd_stats_kernel, grid, dim3(BN_THREADS), 0, static_cast<hipStream_t>(stream), dY, lddy, Y,
                       ldy, Z, ldz, mean, rstd, as, partials, M, N);
    return swr_launch_status();
}

__global__ __launch_bounds__(BN_THREADS) void bn_bwd_finalize_kernel(const float* __restrict__ part, int n_tiles, int64_t M,
                                                                     int N, const float* __restrict__ gamma,
                                                                     const float* __restrict__ rstd, float* dgamma,
                                                                     float* dbeta, int accumulate, float* ca, float* cb,
                                                                     float* cc) {
    __shared__ double t1[BN_THREADS / 64], t2[BN_THREADS / 64];
    const int n = blockIdx.x;
    double a1 = 0.0, a2 = 0.0;
    const int per = (n_tiles + BN_THREADS - 1) / BN_THREADS;
    const int t0 = threadIdx.x * per;
    for (int t = t0; t < min(t0 + per, n_tiles); ++t) {
        const float2 p = *reinterpret_cast<const float2*>(part + (static_cast<int64_t>(t) * N + n) * 2);
        a1 += p.x;
        a2 += p.y;
    }
    const double S1 = swr_block_sum_f64<BN_THREADS>(a1, t1), S2 = swr_block_sum_f64<BN_THREADS>(a2, t2);
    if (threadIdx.x == 0) {
        const double g = gamma ? gamma[n] : 1.0, rs = rstd[n];
        if (dgamma) dgamma[n] = (accumulate ? dgamma[n] : 0.f) + static_cast<float>(S2);
        if (dbeta) dbeta[n] = (accumulate ? dbeta[n] : 0.f) + static_cast<float>(S1);
        // dZ = g rs dA - g rs^2 (S2 / M) (Z - mean) - g rs (S1 / M)
        ca[n] = static_cast<float>(g * rs);
        cb[n] = static_cast<float>(-g * rs * rs * S2 / static_cast<double>(M));
        cc[n] = static_cast<float>(-g * rs * S1 / static_cast<double>(M));
    }
}

extern "C" int swr_bn_bwd_finalize(const float* partials, int n_tiles, int64_t M, int N, const float* gamma,
                                   const float* rstd, float* dgamma, float* dbeta, int accumulate, float* ca, float* cb,
                                   float* cc, void* stream) {
    SWR_REQUIRE(partials && rstd && ca && cb && cc && n_tiles > 0 && M > 0 && N > 0, SWR_ERR_ARG);
    SWR_REQUIRE(n_tiles <= M, SWR_ERR_ARG);          // any row tiling: the partials are summed in tile order
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(N), dim3(BN_THREADS), 0, static_cast<hipStream_t>(stream), partials,
                       n_tiles, M, N, gamma, rstd, dgamma, dbeta, accumulate, ca, cb, cc);
    return swr_launch_status();
}

__global__ __launch_bounds__(BN_THREADS) void act_bwd_apply_kernel(
    const float* __restrict__ dY, int64_t lddy, const float* __restrict__ Y, int64_t ldy, const float* __restrict__ Z,
    int64_t ldz, const float* __restrict__ ca, const float* __restrict__ cb, const float* __restrict__ cc,
    const float* __restrict__ mean, const ActSpec acts, float* __restrict__ dZ, int64_t lddz, int64_t M, int N,
    int rows_per_block) {
    const int n = blockIdx.x * BN_THREADS + threadIdx.x;           // thread = one column, rows_per_block rows of it
    if (n >= N) return;
    const int64_t m0 = static_cast<int64_t>(blockIdx.y) * rows_per_block, m1 = min<int64_t>(m0 + rows_per_block, M);
    int lo_, group_;
    const int act = find_act(acts, n, lo_, group_);
    for (int64_t m = m0; m < m1; m += EW_UNROLL)               // four rows in flight per thread
        act_bwd_apply_rows<EW_UNROLL>(dY, lddy, Y, ldy, Z, ldz, ca, cb, cc, mean, act, lo_, group_, dZ, lddz, n, m,
                                      static_cast<int>(min<int64_t>(EW_UNROLL, m1 - m)));
}

extern "C" int swr_act_bwd_apply(const float* dY, int64_t lddy, const float* Y, int64_t ldy, const float* Z, int64_t ldz,
                                 const float* ca, const float* cb, const float* cc, const float* mean,
                                 const swr_act_range* acts, int n_acts, float* dZ, int64_t lddz, int64_t M, int N,
                                 void* stream) {
    SWR_REQUIRE(dY && Y && dZ && M >= 0 && N > 0, SWR_ERR_ARG);
    SWR_REQUIRE(cb == nullptr || (Z && cc && mean), SWR_ERR_ARG);
    ActSpec as;
    const int rc = make_acts(acts, n_acts, N, as);
    if (rc != SWR_OK) return rc;
    if (M == 0) return SWR_OK;
    int n4;
    if (v4_split(as, N, lddy, ldy, cb ? ldz : 4, lddz, dY, Y, cb ? Z : nullptr, dZ, ca, cb, cc, mean, n4)) {
        V4Plan pl;
        pl.vpr = n4 / 4;
        pl.rows = BN_THREADS / pl.vpr;
        const unsigned v4_blocks = static_cast<unsigned>(swr_ceil_div(M, pl.rows * V4_ITERS));
        hipLaunchKernelGGL(act_bwd_apply_v4_kernel, dim3(v4_blocks + tail_blocks(n4, N, M)), dim3(BN_THREADS), 0,
                           static_cast<hipStream_t>(stream), dY, lddy, Y, ldy, Z, ldz, ca, cb, cc, mean, as, dZ, lddz, M, N, pl,
                           static_cast<int>(v4_blocks));
        return swr_launch_status();
    }
    const int rpb = static_cast<int>(std::max<int64_t>(EW_ROWS, swr_ceil_div(M, 65535)));       // gridDim.y <= 65535
    hipLaunchKernelGGL(act_bwd_apply_kernel, dim3(static_cast<unsigned>(swr_ceil_div(N, BN_THREADS)), static_cast<unsigned>(swr_ceil_div(M, rpb))),
                       dim3(BN_THREADS), 0, static_cast<hipStream_t>(stream), dY, lddy, Y, ldy, Z, ldz, ca, cb, cc, mean, as, dZ, lddz, M, N, rpb);
    return swr_launch_status();
}
