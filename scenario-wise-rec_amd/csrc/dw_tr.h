// Weight gradient of the fused first layer, transpose-read form (dw_tr.hip):
//   dWp[n, c] = sum_b dZ[b, n] A'[b, c],  dZ = ca dY + cb (Z - mean) + cc recomputed from dY and Z,
//   A' = the compact layout of first_layer.hip ([real 16-column groups | one-hot columns]) that is never written.
// Reference work: the weight gradient of the stacked expert / gate Linear under autograd, basic/layers.py:253-258, mmoe.py:37-40.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct DwTrArgs {
    const char* dY; const char* Z;          // [M][ld] fp32, byte pointers (32-bit byte offsets inside the kernel)
    uint32_t lddy_b, ldz_b;                 // row pitches in bytes
    const float* ca; const float* cb; const float* cc; const float* mean;      // [K1]
    int64_t M;
    int K1, K2;                             // columns of dZ (the layer's outputs), of A' (16 NR + one-hot width)
    const char* ws;                         // workspace of the keys launch: a piece = 48 bytes (h | m | l of 8 columns) at ws + voff
    const uint32_t* voff;                   // [ceil(M / 32)][NR][64]: lane s * 32 + i -> piece 2 g + s of sample 32 T + i
    const uint32_t* mask_t;                 // [ceil(ohw / 32)][M]: one-hot bits, word w = columns 32 w .. of the one-hot block
    int NR;                                 // real 16-column groups of A'
    float* part;                            // [n_splits][K1][k2p]
    float* part_cs;                         // [n_splits][K1] column sums of dZ, or null
    int k2p;
    int64_t rows_per_split;                 // a multiple of 32
    int n_splits;
};

// shape test: K1 / K2 tiles and the one-hot start (kp = 16 NR) are an instantiated form
bool dw_tr_shape_ok(int K1, int K2, int NR);
int dw_tr_launch(const DwTrArgs& a, hipStream_t st);
