// Epilogue of the row-parallel products (gemm.hip, first_layer.hip): C tile + bias (+ activation) + the per-32-row-tile
// BatchNorm partials (mean, M2).
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef SWR_EPI_OLD
#define SWR_EPI_OLD 0
#endif

// epilogue shared by the row-parallel kernels: lane (i, s) holds column n0 + 32t + i, rows m0 + (r & 3) + 8 (r >> 2) + 4 s
// c_act (swr.h): the activation of a layer WITHOUT BatchNorm applied while C is stored (1 ReLU, 2 sigmoid)
__device__ __forceinline__ float epi_act(int act, float v) {
    return act == 1 ? fmaxf(v, 0.f) : (act == 2 ? swr_sigmoid(v) : v);
}

template <int NT>
__device__ __forceinline__ void rows_epilogue(const swr_gemm_args& a, int n_tiles_m, f32x16 (&acc)[NT], int g, int64_t tile_m,
                                              int64_t m0, int n0, int i, int s) {
    const int N = a.N;
    float* __restrict__ Cg = a.C + g * a.gsC;
    const float* __restrict__ bias = a.bias ? a.bias + g * a.gsBias : nullptr;
    const int nvalid = static_cast<int>(min<int64_t>(32, a.M - m0));
    if (nvalid <= 0) return;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n0 + 32 * t + i;
#ifdef FLF_EPI_NOBIAS          // (FLF_EPI_*: ablation switches of tools/micro/ab_flf.sh, never defined in the product build)
        float bn = 0.f;
#else
        float bn = (bias && n < N) ? bias[n] : 0.f;
#endif
        if (a.c_act) {
            // (wave-uniform, outside the store loops: with the activation inside them the plain case -- every product of
            // config 2 -- lost 9 us per step to the epilogue)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = epi_act(a.c_act, acc[t][r] + bn);
            bn = 0.f;
        }
        float sum = 0.f;
        if (nvalid == 32 && n0 + 32 * t + 32 <= N && !a.accumulate) {
            // whole tile inside C (wave-uniform test): no per-element predicates; wave-uniform row bases + one 32-bit
            // lane offset, so each store is a scalar-base access instead of a 64-bit multiply-add per element
            const uint32_t lane_off = static_cast<uint32_t>(4 * s) * static_cast<uint32_t>(a.ldc) + static_cast<uint32_t>(n);
            float* __restrict__ tile_base = Cg + m0 * a.ldc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[t][r] + bn;
#ifndef FLF_EPI_NOSTORE
                (tile_base + static_cast<int64_t>((r & 3) + 8 * (r >> 2)) * a.ldc)[lane_off] = v;
#endif
                sum += v;
                acc[t][r] = v;
            }
        } else if (nvalid == 32 && !a.accumulate && !SWR_EPI_OLD) {
            // full rows, the layer's last (ragged) column tile: the same scalar-base stores under ONE lane predicate.  (Through
            // the element-wise branch below -- a row test, a 64-bit address and an accumulate test per element -- this one
            // tile cost the fused first layer's forward 5 of its 34 us at N = 148.)
            const bool okn = n < N;
            const uint32_t lane_off = static_cast<uint32_t>(4 * s) * static_cast<uint32_t>(a.ldc) + static_cast<uint32_t>(n);
            float* __restrict__ tile_base = Cg + m0 * a.ldc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[t][r] + bn;
                if (okn) {
                    (tile_base + static_cast<int64_t>((r & 3) + 8 * (r >> 2)) * a.ldc)[lane_off] = v;
                    sum += v;
                }
                acc[t][r] = v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * s;
                float v = acc[t][r] + bn;
                const bool ok = row < nvalid && n < N;
                if (ok) {
                    float* c = Cg + (m0 + row) * a.ldc + n;
                    if (a.accumulate) v += *c;
                    *c = v;
                    sum += v;
                }
                acc[t][r] = v;
            }
        }
#ifdef FLF_EPI_NOSTATS
        if (sum == 12345.678f) a.C[0] = sum;
        if (false) {
#else
        if (a.stat_partials) {
#endif
            // BatchNorm batch statistics of this 32-row tile, two-pass in registers
            sum += __shfl_xor(sum, 32);
            const float mean = sum / static_cast<float>(nvalid);
            float m2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * s;
                const float d = acc[t][r] - mean;
                if (row < nvalid) m2 = fmaf(d, d, m2);
            }
            m2 += __shfl_xor(m2, 32);
            if (s == 0 && n < N) {
                float* sp = a.stat_partials + ((tile_m * a.groups + g) * N + n) * 2;   // [tiles][groups * N][2]
                sp[0] = mean;
                sp[1] = m2;
            }
        }
    }
}

