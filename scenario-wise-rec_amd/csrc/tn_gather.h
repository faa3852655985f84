// Operand description of the weight-gradient product whose second operand is GATHERED (first_layer.hip -> gemm.hip):
//   dWp[n, c] = sum_b dZ[b, n] A'[b, c],   A' = the compact layout [table pieces | dense pieces | 0 | one-hot] that is never
// written -- the staging threads of gemm_tn_x6g_kernel fetch table rows through the row keys, dense columns from the fp32
// block of the keys launch, and expand the one-hot columns from the transposed mask words.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "swr.h"

#define TNG_MAX_PIECES 48
enum { TNG_ZERO = 0, TNG_TABLE = 1, TNG_ROWIDX = 2, TNG_ONEHOT = 3 };
struct TnGatherPiece {        // 8 consecutive columns of A'
    const float* vbase;       // TABLE: table + first column inside the row; ROWIDX: fp32 block [M][vstride] + first column
    const uint32_t* kwp;      // TABLE: row keys of the slot [M]; ONEHOT: the 32-bit mask word that holds the piece's bits [M]
    uint32_t vstride;         // floats per row of vbase
    uint32_t kmax;            // TABLE: vocab - 1 (keys read past row M - 1 are garbage: clamped, their rows are zeroed)
    int16_t kind, bit0;       // ONEHOT: bit of the piece's first column in its word
    int16_t n_valid, pad;     // ROWIDX: valid columns of the piece
};
struct TnGather {
    TnGatherPiece piece[TNG_MAX_PIECES];
    int n_pieces;
    int kp;                   // first one-hot column (a multiple of 16)
    // A recomputed while it is staged (wide kernel only): the product's A pointer is dY and
    //   A[b, n] = ca[n] dY[b, n] + cb[n] (Z[b, n] - mean[n]) + cc[n]
    // -- the BatchNorm backward of swr_act_bwd_apply / swr_bn_bwd_dx in its operation order, so dZ is never written.  a_z = null: A is A.
    const float* a_z;
    int64_t a_ldz;
    const float* a_ca; const float* a_cb; const float* a_cc; const float* a_mean;
    // the pre-split pieces of the keys launch (first_layer.hip): with them and a recomputed A the transpose-read kernel (dw_tr.hip)
    // takes the product where its shape is instantiated.  tr_voff = null: not available.
    const char* tr_ws; const uint32_t* tr_voff; const uint32_t* tr_mask_t; int tr_nr;
};

// a = the product's description with B ignored (K2 = columns of A'); same workspace size as swr_gemm_tn_workspace_bytes
int tn_x6_gather(const swr_gemm_tn_args& a, const TnGather& g, void* workspace, size_t workspace_bytes, void* stream);
bool tn_x6_gather_ok(const swr_gemm_tn_args& a);
bool tn_x6_gather_wide(const swr_gemm_tn_args& a, int kp);      // the wide kernel takes this product (a recomputed A needs it)
