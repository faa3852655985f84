// Routed inference of an MMoE head: in eval mode BatchNorm is a fixed affine, so a row's output
// depends only on its own domain's gate and tower -- the reference still evaluates every domain's mix and tower on the
// whole batch and selects afterwards (models/multi_domain/mmoe.py:48-55).  Here each row mixes the experts with ITS
// domain's gate probabilities and runs ITS domain's tower [Linear(H, T) -> BN(eval) -> ReLU -> Linear(T, 1)] ->
// sigmoid; rows whose domain id lies outside [0, D) give 0.0, like the select.  One launch, one read of the [B, ne H +
// D ne] activations; all tower weights of all domains sit in LDS (D (T H + 4 T + 1) floats).
// Workgroup = 64 rows x 4 parts: the activation tile is staged through LDS (coalesced), part q of a row computes a
// quarter of the pooled vector, then a quarter of the hidden units, the four partial logits are added in order.
#include "common.h"

#define RT_ROWS 64
#define RT_THREADS 256

struct RoutedK {
    const float* Y; int64_t ldy; int64_t M;
    int ne, H, D, T;
    const float* W1; const float* b1; const float* scale1; const float* shift1; const float* w2; const float* b2;
    const void* dom; int dom_dtype;
    float* out;
};

__global__ __launch_bounds__(RT_THREADS) void routed_mmoe_eval_kernel(const RoutedK k) {
    extern __shared__ float lds[];
    const int ne = k.ne, H = k.H, D = k.D, T = k.T;
    const int W = ne * H + D * ne;                  // activation columns
    const int WP = W + 1;                            // LDS pitch (odd: rows of a wave hit different banks)
    float* sW1 = lds;                                // [D][T][H]
    float* sb1 = sW1 + D * T * H;                    // [D][T]   bias of the first tower layer
    float* ssc = sb1 + D * T;                        // [D][T]   BN(eval) scale
    float* ssh = ssc + D * T;                        // [D][T]   BN(eval) shift
    float* sw2 = ssh + D * T;                        // [D][T]
    float* sb2 = sw2 + D * T;                        // [D]
    float* sY = sb2 + D;                             // [RT_ROWS][WP]
    float* sP = sY + RT_ROWS * WP;                   // [RT_ROWS][H + 1] pooled vectors
    float* sL = sP + RT_ROWS * (H + 1);              // [4][RT_ROWS] partial logits
    const int tid = threadIdx.x;
    for (int j = tid; j < D * T * H; j += RT_THREADS) sW1[j] = k.W1[j];
    for (int j = tid; j < D * T; j += RT_THREADS) {
        sb1[j] = k.b1[j]; ssc[j] = k.scale1[j]; ssh[j] = k.shift1[j]; sw2[j] = k.w2[j];
    }
    for (int j = tid; j < D; j += RT_THREADS) sb2[j] = k.b2[j];
    const int64_t m0 = static_cast<int64_t>(blockIdx.x) * RT_ROWS;
    const int rows = static_cast<int>(min<int64_t>(RT_ROWS, k.M - m0));
    for (int j = tid; j < rows * W; j += RT_THREADS) {
        const int r = j / W, c = j - r * W;
        sY[r * WP + c] = k.Y[(m0 + r) * k.ldy + c];
    }
    __syncthreads();
    const int r = tid & (RT_ROWS - 1), q = tid >> 6;          // row of the tile, part 0..3
    int d = -1;
    if (r < rows) {
        const int64_t dv = swr_load_index(k.dom, k.dom_dtype, m0 + r);
        d = (dv >= 0 && dv < D) ? static_cast<int>(dv) : -1;
    }
    // pooled[c] = sum_j gate[d][j] * expert_j[c], columns c = q, q + 4, ...
    if (d >= 0) {
        const float* y = sY + r * WP;
        const float* g = y + ne * H + d * ne;
        for (int c = q; c < H; c += 4) {
            float p = 0.f;
            for (int j = 0; j < ne; ++j) p = fmaf(g[j], y[j * H + c], p);
            sP[r * (H + 1) + c] = p;
        }
    }
    __syncthreads();
    float part = 0.f;
    if (d >= 0) {
        const float* p = sP + r * (H + 1);
        for (int u = q; u < T; u += 4) {
            const float* w = sW1 + (d * T + u) * H;
            float z = sb1[d * T + u];
            for (int c = 0; c < H; ++c) z = fmaf(w[c], p[c], z);
            const float h = fmaxf(fmaf(z, ssc[d * T + u], ssh[d * T + u]), 0.f);
            part = fmaf(sw2[d * T + u], h, part);
        }
    }
    sL[q * RT_ROWS + r] = part;
    __syncthreads();
    if (q == 0 && r < rows) {
        float o = 0.f;
        if (d >= 0) {
            const float logit = ((sL[r] + sL[RT_ROWS + r]) + (sL[2 * RT_ROWS + r] + sL[3 * RT_ROWS + r])) + sb2[d];
            o = swr_sigmoid(logit);
        }
        k.out[m0 + r] = o;
    }
}

static size_t routed_lds_bytes(int ne, int H, int D, int T) {
    const size_t W = static_cast<size_t>(ne) * H + static_cast<size_t>(D) * ne;
    return (static_cast<size_t>(D) * T * H + 4 * static_cast<size_t>(D) * T + D + RT_ROWS * (W + 1) + RT_ROWS * (H + 1) + 4 * RT_ROWS) * 4;
}

extern "C" int swr_routed_mmoe_eval_supported(int n_expert, int H, int D, int T) {
    return n_expert >= 1 && H >= 1 && D >= 1 && T >= 1 && routed_lds_bytes(n_expert, H, D, T) <= 150 * 1024;
}

extern "C" int swr_routed_mmoe_eval(const float* Y, int64_t ldy, int64_t M, int n_expert, int H, int D, int T,
                                    const float* W1, const float* b1, const float* scale1, const float* shift1,
                                    const float* w2, const float* b2, const void* domain, int domain_dtype, float* out,
                                    void* stream) {
    SWR_REQUIRE(Y && W1 && b1 && scale1 && shift1 && w2 && b2 && domain && out && M >= 0, SWR_ERR_ARG);
    SWR_REQUIRE(swr_is_index_dtype(domain_dtype), SWR_ERR_DTYPE);
    SWR_REQUIRE(swr_routed_mmoe_eval_supported(n_expert, H, D, T), SWR_ERR_UNSUPPORTED);
    SWR_REQUIRE(ldy >= static_cast<int64_t>(n_expert) * H + static_cast<int64_t>(D) * n_expert, SWR_ERR_ARG);
    if (M == 0) return SWR_OK;
    RoutedK k;
    k.Y = Y; k.ldy = ldy; k.M = M; k.ne = n_expert; k.H = H; k.D = D; k.T = T;
    k.W1 = W1; k.b1 = b1; k.scale1 = scale1; k.shift1 = shift1; k.w2 = w2; k.b2 = b2;
    k.dom = domain; k.dom_dtype = domain_dtype; k.out = out;
    const size_t lds = routed_lds_bytes(n_expert, H, D, T);
    if (lds > 64 * 1024 && !swr_raise_lds(reinterpret_cast<const void*>(routed_mmoe_eval_kernel), 150 * 1024)) return SWR_ERR_LAUNCH;
    hipLaunchKernelGGL(routed_mmoe_eval_kernel, dim3(static_cast<unsigned>(swr_ceil_div(M, RT_ROWS))), dim3(RT_THREADS), lds,
                       static_cast<hipStream_t>(stream), k);
    return swr_launch_status();
}
