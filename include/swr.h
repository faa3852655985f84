/*
 * swr.h -- C ABI of the MI355X-native multi-domain CTR hot path ("libswr").
 *
 * The reference (Xiaopengli1/Scenario-Wise-Rec) has no FFI / plugin layer: the
 * boundary its hot path sits behind is the Python module API
 * (scenario_wise_rec.basic.layers / .models.multi_domain / .trainers), and the
 * arithmetic is dispatched to ATen.  This header is the operator boundary a
 * replacement for that ATen work binds to; every entry point cites the
 * reference call site whose work it replaces (paths relative to the reference
 * root).  INTEGRATION.md shows the ctypes stub a maintainer of the reference
 * would add.
 *
 * Conventions
 *  - plain C symbols, plain pointers and sizes, no C++ / torch types;
 *  - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *  - the caller owns every buffer; the library allocates nothing persistent
 *    (scratch comes in through explicit `workspace` arguments whose size the
 *    matching `*_workspace_bytes` call returns);
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*),
 *    performs no hidden synchronisation and is capturable into a hipGraph;
 *  - return value: SWR_OK (0) or a negative swr_status; nothing throws;
 *  - re-entrant and thread-safe for distinct streams / workspaces;
 *  - matrices are row-major fp32 with an explicit leading dimension (`ld*`,
 *    in elements).
 */
#ifndef SWR_H_
#define SWR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWR_ABI_VERSION 8

typedef enum {
    SWR_OK = 0,
    SWR_ERR_ARG = -1,         /* NULL pointer, negative size, bad enum */
    SWR_ERR_DTYPE = -2,       /* unsupported element type */
    SWR_ERR_ALIGN = -3,       /* pointer / leading dimension not aligned as required */
    SWR_ERR_LAUNCH = -4,      /* hip launch / runtime error */
    SWR_ERR_UNSUPPORTED = -5, /* shape outside the implemented envelope */
    SWR_ERR_WORKSPACE = -6    /* workspace too small */
} swr_status;

/* element types of caller-provided index / value columns: the reference casts
 * with `.long()` / `.float()` (basic/layers.py:70,89) after
 * `reduce_mem_usage` may have narrowed them (utils/data.py:109-124) */
typedef enum {
    SWR_I8 = 1, SWR_I16 = 2, SWR_I32 = 3, SWR_I64 = 4, SWR_U8 = 5,
    SWR_F16 = 6, SWR_BF16 = 7, SWR_F32 = 8, SWR_F64 = 9, SWR_BOOL = 10
} swr_dtype;

/* bits of the device-side error word (`err_flag` arguments, may be NULL) */
#define SWR_FLAG_INDEX_OOR 1u      /* embedding index outside [0, vocab): torch raises IndexError */
#define SWR_FLAG_GRAD_RANGE 2u     /* |embedding grad| >= 2^20: outside the fixed-point accumulator */

int swr_abi_version(void);
/* Zero-fill of `bytes` at a 16-byte aligned address as an ordinary kernel (model.zero_grad() of the gradient arena,
 * `ctr_trainer.py:71`; hipMemsetAsync nodes misbehaved under back-to-back hipGraph replays on ROCm 7.2). */
int swr_zero(void* p, size_t bytes, void* stream);
/* Test instrument (tests/test_skew_gpu.py): keeps `stream` busy for `us` microseconds with one idle-spinning wave.  The
 * Python layer injects it at its stream forks when SWR_SKEW is set, to expose missing cross-stream dependencies. */
int swr_spin_us(int us, void* stream);
/* Measurement instrument (SWR_STAMPS): a one-lane kernel that writes the device's 100 MHz wall clock to *slot (device memory)
 * when `stream` reaches it -- the true placement of the branches of a replayed hipGraph, which a kernel trace perturbs. */
int swr_stamp(unsigned long long* slot, void* stream);
const char* swr_status_str(int status);
/* 1 when a HIP device is visible to the calling process (no compute is done) */
int swr_device_available(void);

/* ------------------------------------------------------------------ K1 ----
 * EmbeddingLayer.forward(x, features, squeeze_dim=True)
 * (basic/layers.py:64-105): F_s row gathers + F_d casts + two cats, fused
 * into ONE launch that writes straight into the [B, K0] concat layout
 * (sparse block first, dense columns last -- the caller assigns `out_col`).
 * Optional hash stage (build-side addition, the reference has none): when
 * `hash_seed != 0` the raw id is mapped to row = mix64(id ^ seed) % vocab.
 * `keys_out` (nullable) receives the looked-up row of every (slot, sample) as
 * uint32 [n_sparse * B]; the backward consumes it. */
typedef struct {
    const float* weight;   /* [vocab, dim] table */
    const void* idx;       /* [B] ids, type idx_dtype */
    int64_t vocab;
    int32_t dim;
    int32_t idx_dtype;     /* swr_dtype, integer kinds */
    int32_t out_col;       /* first output column */
    uint32_t hash_seed;    /* 0 = ids are rows (the reference behaviour) */
} swr_sparse_slot;

typedef struct {
    const void* values;    /* [B] */
    int32_t dtype;         /* swr_dtype */
    int32_t out_col;
} swr_dense_slot;

int swr_embed_gather_fwd(const swr_sparse_slot* sparse_host, int n_sparse,
                         const swr_dense_slot* dense_host, int n_dense,
                         int64_t B, float* out, int64_t ld_out,
                         uint32_t* keys_out, uint32_t* err_flag, void* stream);
/* The same lookup, plus a ONE-HOT block of the small tables behind the concat (training only; no reference
 * counterpart -- it restructures the backward of `embedding -> Linear`, basic/layers.py:64-105 + 253): columns
 * [oh_col, oh_col + oh_width) of `out`, slot s with oh_off[s] >= 0 owns columns oh_col + oh_off[s] + v, v < vocab_s, and
 * out[b, oh_col + oh_off[s] + v] = (row_s(b) == v) ? 1 : 0; unowned columns of the block and the alignment columns
 * [pad_col, oh_col) (fewer than 16) are zeros.  With the block in place, the weight-gradient product of the layer that consumes the
 * concat, dZ^T [out | one-hot] (swr_gemm_tn with a second destination), also yields S[v, :] = sum of dZ over the samples
 * whose row is v -- and the gradient of a small table is S W_t (swr_onehot_table_grads): its columns of dX are
 * never computed and it skips K3.  oh_width, oh_col multiples of 4, oh_width <= 256, n_sparse <= 64. */
int swr_embed_gather_fwd_onehot(const swr_sparse_slot* sparse_host, int n_sparse,
                                const swr_dense_slot* dense_host, int n_dense,
                                int64_t B, float* out, int64_t ld_out, uint32_t* keys_out,
                                const int32_t* oh_off_host /* [n_sparse], -1 = none */, int pad_col, int oh_col, int oh_width,
                                uint32_t* err_flag, void* stream);
/* grad_t[v, e] (+)= sum_n S_t[n, oh_off_t + v] * W[n, w_col_t + e] for every listed small table t: S_t = the second
 * destination of the weight-gradient product ([N, >= oh columns], leading dimension lds), W = the layer's stacked
 * weights [N, K] (ldw).  Fixed summation order (deterministic). */
typedef struct {
    float* grad;           /* [vocab, dim] */
    int32_t vocab, dim;
    int32_t oh_off;        /* first column of the table's block in S */
    int32_t w_col;         /* first column of the table's embedding in W */
} swr_onehot_table;
int swr_onehot_table_grads(const float* S, int64_t lds, const float* W, int64_t ldw, int N,
                           const swr_onehot_table* tables_host, int n_tables, int accumulate, void* stream);

/* K1 for SequenceFeature columns (reference basic/layers.py:73-87 pooled lookup; InputMask 117-146; Sum / Average /
 * ConcatPooling 174-228): `idx` is [B, L] (padded id sequences, row-major, type idx_dtype).
 *   mode 0 sum    : out[b, out_col : +dim] = sum_l m(b,l) W[idx[b,l]]
 *   mode 1 mean   : the same sum / (count_b + 1e-16)
 *   mode 2 concat : out[b, out_col + l*dim : +dim] = W[idx[b,l]]   (no mask)
 * m(b,l) = idx[b,l] != padding_idx when has_pad, else idx[b,l] != -1.  Masked ids are never dereferenced.
 * `keys_out` ([B*L] uint32, nullable) = looked-up row per position (0 where masked), `wts_out` ([B*L] float, nullable) =
 * the factor the position's output gradient is scaled with (0 where masked; 1/(count+1e-16) for mean): the backward is
 * swr_embed_bag_bwd_expand (one gradient row per position: d_rows[(b,l), :] = wts[b,l] * d_out[b, in_col (+ l*dim) : +dim])
 * followed by swr_embed_bwd over B*L "samples" of one slot (in_col 0, ld = dim). */
int swr_embed_bag_fwd(const float* weight, int64_t vocab, int dim, const void* idx, int idx_dtype, int64_t B, int L,
                      int mode, int has_pad, int64_t padding_idx, uint32_t hash_seed, float* out, int64_t ld_out,
                      int out_col, uint32_t* keys_out, float* wts_out, uint32_t* err_flag, void* stream);
int swr_embed_bag_bwd_expand(const float* d_out, int64_t ld, int in_col, int dim, int L, int concat, const float* wts,
                             int64_t B, float* d_rows /* [B*L, dim] */, void* stream);

/* ---- folded first layer (training; no reference counterpart: an algebraic restructuring of `embedding -> Linear`,
 * basic/layers.py:64-105 + 253).  With the one-hot block O of the small tables next to the embeddings E_big of the
 * other tables and the dense features x_d, the layer's product is
 *     [E_small | E_big | x_d] W^T  =  [E_big | x_d | O] Wp^T,   Wp = [W_big | W_d | P],  P_t[:, v] = W_t emb_t[v]  (tiny)
 * so the lookup writes only [E_big | x_d | O] (a slot with dim = 0 contributes nothing but its one-hot columns) and the
 * products of the layer run over Kp + ohw columns instead of K.  Backward: dWp = dZ^T [E_big | x_d | O] (swr_gemm_tn),
 * then swr_fold_first_layer_bwd scatters it back: dW[:, c] (+)= dWp[:, compact(c)] for the other tables' and the dense
 * columns, dW[:, col_t + e] (+)= sum_v S_t[:, v] emb_t[v, e] with S = dWp[:, Kp:], and db (+)= dbp;
 * swr_onehot_table_grads(S, W) gives the small tables' own gradients.
 * src_col [Kp] (device): column of W behind compact column j (-1: zero padding); inv_col [K] (device): compact column
 * of W's column c, or -1 - t for a column of small table t (index into `tables`).  tables[t].grad = emb_t here. */
int swr_fold_first_layer_fwd(const float* W, int64_t ldw, int N, int K, int Kp, int ohw, const int32_t* src_col,
                             const int32_t* inv_col, const int32_t* oh_table /* [ohw] (device): table of one-hot column o, -1 = padding */,
                             const swr_onehot_table* tables_host, int n_tables, float* Wp, int64_t ldwp,
                             const int64_t* sel /* nullable: [n_sel] columns of W (device) */, int n_sel,
                             float* Wt_sel /* [n_sel, ldt]: Wt_sel[r, n] = W[n, sel[r]], the operand of the backward's dX product */,
                             int64_t ldt, void* stream);
int swr_fold_first_layer_bwd(const float* dWp, int64_t lddwp, const float* dbp /* nullable */, int N, int K, int Kp, int ohw,
                             const int32_t* src_col, const int32_t* inv_col, const swr_onehot_table* tables_host,
                             int n_tables, float* dW, int64_t lddw, float* db /* nullable */, int accumulate, void* stream);
/* swr_fold_first_layer_bwd and swr_onehot_table_grads(S = dWp, ...) in ONE launch: both read the reduced dWp and nothing of each
 * other.  `tables` carry the table VALUES (the unfolding multiplies with them), `grad_tables` the gradient destinations with
 * oh_off counted from column 0 of dWp (Kp + the table's offset in the one-hot block); W [N][ldw] is the layer's weight. */
int swr_fold_first_layer_bwd_tables(const float* dWp, int64_t lddwp, const float* dbp, int N, int K, int Kp, int ohw,
                                    const int32_t* src_col, const int32_t* inv_col, const swr_onehot_table* tables, int n_tables,
                                    float* dW, int64_t lddw, float* db, int accumulate, const float* W, int64_t ldw,
                                    const swr_onehot_table* grad_tables, int n_grad_tables, void* stream);

/* ---- fused lookup + first layer ("fl"; training; no reference counterpart: `embedding -> Linear`,
 * basic/layers.py:64-105 + 253-258, with the lookup as the A-operand producer of the layer's products).  The [B, K0]
 * concat is never written.  The layer's k axis is the COMPACT layout of the folded first layer above, in 8-column pieces:
 *     [ pieces of the tables that keep an embedding (K3 tables) | dense-feature pieces | zero pieces ]  = 16 * n_real_groups
 *     [ one-hot columns of the small tables ]                                                            = oh_width (<= 128)
 * and three launches replace  lookup + fold + product:
 *   swr_fl_prep : parameters only.  Wf = [W_big | W_d | 0 | P] split into three bf16 terms and laid out in MFMA fragment
 *                 order (B3 image), bf16-term shadows of the tables of kind SWR_FL_PLANES, and Wt_sel (rows `sel` of W^T,
 *                 the operand of the backward's dX product);
 *   swr_fl_keys : ids only.  Row keys of slots [0, n_keys) ([n_keys][B] uint32: K3 and the weight-gradient product read
 *                 them), the one-hot block as a 128-bit mask per sample (bit c = one-hot column c), the byte offset of every
 *                 (32-row tile, group, lane)'s 48-byte piece (three 16-byte bf16 terms h | m | l of 8 columns), and the
 *                 pieces of kind SWR_FL_ROWS / SWR_FL_DENSE gathered from fp32 and split;
 *   swr_fl_fwd  : Z[B, N] = A' Wf^T + bias, per-32-row-tile (mean, M2) of every column -- fp32 results from six bf16
 *                 MFMA products per real group and three per one-hot group (same arithmetic as swr_gemm_nt on the written
 *                 concat); a lane's A fragment is three 16-byte loads through its offset, or 8 mask bits.
 * Everything the launches exchange lives in ONE caller-owned workspace (swr_fl_workspace_bytes; sections:
 * swr_fl_layout).  n_real_groups in [1, 16], oh_width a multiple of 16 (<= 128), N <= 160, dims multiples of 8. */
enum { SWR_FL_ZERO = 0, SWR_FL_PLANES = 1, SWR_FL_ROWS = 2, SWR_FL_DENSE = 3 };
typedef struct {
    int32_t kind;          /* SWR_FL_*: zero padding / small table read through its bf16-term shadow / fp32 table rows
                              (large, row-sparse tables) / dense-feature columns */
    int32_t slot;          /* PLANES, ROWS: lookup slot; DENSE: first dense feature of the piece */
    int32_t off;           /* PLANES, ROWS: first column inside the table row (multiple of 8) */
    int32_t n_valid;       /* DENSE: features in the piece (<= 8; the rest are zero columns); others: 8 */
    int32_t w_col;         /* first column of W behind the piece (n_valid consecutive columns) */
    int32_t pad;
} swr_fl_piece;
typedef struct {
    const swr_sparse_slot* sparse_host; int32_t n_sparse;   /* as swr_embed_gather_fwd; out_col = the slot's column in W */
    const swr_dense_slot* dense_host; int32_t n_dense;
    int32_t n_keys;                 /* slots [0, n_keys) get their row keys written (the slots that keep an embedding) */
    int32_t n_real_groups;          /* 16-column groups in front of the one-hot block */
    swr_fl_piece piece[32];         /* [2 * n_real_groups] */
    int32_t oh_width;               /* one-hot columns (multiple of 16) */
    const int32_t* oh_off_host;     /* [n_sparse]: first one-hot column of the slot, -1 = none */
    int64_t B;
    int32_t N;                      /* output columns of the layer (swr_fl_layout / swr_fl_keys accept 0 = not known yet) */
    int32_t pad;
} swr_fl_plan;
typedef struct {                    /* byte offsets of the workspace sections (keys: what swr_embed_bwd* take) */
    int64_t zero, planes, a3f, b3, keys, mask, mask_t, voff, densef, b3x, total;
    int32_t nd4;                    /* row pitch of densef (fp32 dense features, [B][nd4]) */
    int32_t n_fpieces;              /* pieces of kind ROWS / DENSE */
} swr_fl_offsets;
int swr_fl_layout(const swr_fl_plan* plan_host, swr_fl_offsets* out_host);
size_t swr_fl_workspace_bytes(const swr_fl_plan* plan_host);
int swr_fl_prep(const swr_fl_plan* plan_host, const float* W, int64_t ldw, int K,
                const int32_t* oh_table /* [oh_width] (device): table of one-hot column o, -1 = padding */,
                const swr_onehot_table* tables_host /* .grad = the table's weights */, int n_tables,
                const int64_t* sel /* nullable */, int n_sel, float* Wt_sel, int64_t ldt,
                void* workspace, void* stream);
int swr_fl_keys(const swr_fl_plan* plan_host, void* workspace, uint32_t* err_flag, void* stream);
int swr_fl_fwd(const swr_fl_plan* plan_host, const void* workspace, const float* bias /* nullable */,
               float* Z, int64_t ldz, float* stat_partials /* nullable, [ceil(B/32)][N][2] */, void* stream);
/* Weight gradient of the layer with the same gathered operand: dWp[N, Kp + ohw] = dZ^T A' and colsum[n] = sum_b dZ[b, n]
 * (nullable) -- what swr_gemm_tn(dZ, written block) gives, bit for bit (same batch splits, same fixed-order reduction; the
 * one-hot column tiles take three products instead of six: their middle / low terms are zero).  The staging threads read 8
 * row keys per column pair and stage, then 8-byte pieces of the table rows.  `workspace`: swr_fl_dw_workspace_bytes.
 * swr_fl_dw_supported = 0: the shape is not the bf16-split kernel's (batch < 4096, N > 160, SWR_GEMM != default) -- the
 * caller writes the block (swr_embed_gather_fwd_onehot) and uses swr_gemm_tn. */
/* BatchNorm backward of the layer + its dX product in one pass (no activation in between: the expert / gate level of
 * swr_bnmix_bwd): dZ = ca * dY + cb * (Z - mean) + cc -- swr_act_bwd_apply's arithmetic, bit for bit -- is computed in the A
 * fragment of the product dX[B, n_out] = dZ W[:, sel] (weights: the B3X image swr_fl_prep wrote for `sel`) and written out
 * for the weight-gradient product.  K = plan->N <= 160, n_out = n_sel <= 160. */
int swr_bn_bwd_dx_supported(int K, int n_out);
/* (dZ may be NULL: nothing is written -- for a weight-gradient product that recomputes it, swr_fl_dw_bn) */
int swr_bn_bwd_dx(const swr_fl_plan* plan_host, const void* fl_workspace, const float* dY, int64_t lddy, const float* Z, int64_t ldz,
                  const float* ca, const float* cb, const float* cc, const float* mean, int n_out,
                  float* dZ, int64_t lddz, float* dX, int64_t lddx, void* stream);
int swr_fl_dw_supported(const swr_fl_plan* plan_host, int64_t lddz);
size_t swr_fl_dw_workspace_bytes(const swr_fl_plan* plan_host);
int swr_fl_dw(const swr_fl_plan* plan_host, const void* fl_workspace, const float* dZ, int64_t lddz, float* dWp, int64_t lddwp,
              float* colsum /* nullable */, void* workspace, size_t workspace_bytes, void* stream);
/* swr_fl_dw with dZ RECOMPUTED while it is staged: dZ[b, n] = ca[n] dY[b, n] + cb[n] (Z[b, n] - mean[n]) + cc[n], the BatchNorm
 * backward of swr_act_bwd_apply / swr_bn_bwd_dx in its operation order (`layers.py:254-256` under autograd): together with
 * swr_bn_bwd_dx(dZ = NULL) the [B, N] gradient of the pre-activations is never written or read back.  Supported where the wide
 * weight-gradient kernel takes the product (swr_fl_dw_bn_supported); same workspace as swr_fl_dw. */
int swr_fl_dw_bn_supported(const swr_fl_plan* plan_host, int64_t lddy, int64_t ldz);
/* Which kernel takes swr_fl_dw_bn where both fit: 1 (default; SWR_DW_TR=0 in the environment starts at 0) = the transpose-read
 * form (csrc/dw_tr.hip: operands stored in LDS as they arrive, read back with ds_read_b64_tr_b16; A' from the pre-split pieces
 * by LDS-DMA; all 8 waves multiply), 0 = the wide register-transposing form (gemm_tn_x6w_kernel), whose sums are bit-identical
 * to swr_gemm_tn on the written block.  set < 0: query only.  Returns the previous value.  Both are deterministic. */
int swr_dw_tr_mode(int set);
int swr_fl_dw_bn(const swr_fl_plan* plan_host, const void* fl_workspace, const float* dY, int64_t lddy, const float* Z, int64_t ldz,
                 const float* ca, const float* cb, const float* cc, const float* mean, float* dWp, int64_t lddwp, float* colsum,
                 void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ K3 ----
 * Backward of the lookup: replaces aten::embedding_dense_backward (32 calls
 * per step at the KuaiRand config, each zero-filling [V, E]).
 * All slots are reduced together: sort (table, row) keys, then a segmented
 * reduction in dual-limb 64-bit fixed point, so the result does not depend on
 * the order in which partial sums meet (bitwise deterministic, and identical
 * on every data-parallel rank).  Slots that share a table (`shared_with`,
 * basic/layers.py:71-72) carry the same `table_id` and sum into one gradient.
 *   mode 0 (dense) : grad_dense[vocab, dim] is fully written (zeros included);
 *   mode 2 (dense+): the table's gradient is ADDED to grad_dense (gradient arena fan-in);
 *   mode 1 (sparse): for large tables; `urow` / `ugrad` have one entry per
 *                    looked-up sample of the table, in sorted-row order:
 *                    urow[i] = row (first entry of each distinct row), otherwise the
 *                    bitwise complement of the row the entry belongs to (negative: "no
 *                    entry here"; consumers skip every negative id, and the list stays
 *                    binary-searchable by row -- swr_dp_finish relies on that),
 *                    ugrad[i, :] = summed gradient of that row (or 0).
 * Sparse-mode tables must have the largest table ids. */
typedef struct {
    int64_t vocab;
    int32_t dim;
    int32_t in_col;        /* first column of this slot in dE */
    int32_t table_id;      /* 0..n_tables-1 */
    int32_t mode;          /* 0 dense, 1 sparse, 2 dense accumulate */
    float* grad_dense;     /* mode 0: [vocab, dim] of the TABLE (same pointer for sharing slots) */
    int32_t* urow;         /* mode 1: [count of samples of this table] */
    float* ugrad;          /* mode 1: [count, dim] */
} swr_embed_grad_slot;

size_t swr_embed_bwd_workspace_bytes(const swr_embed_grad_slot* slots_host, int n_slots, int64_t B);
int swr_embed_bwd(const swr_embed_grad_slot* slots_host, int n_slots,
                  const uint32_t* keys /* [n_slots * B] from the forward */,
                  const float* dE, int64_t ld, int64_t B,
                  void* workspace, size_t workspace_bytes, uint32_t* err_flag, void* stream);
/* The same work in two halves, so that the half that needs only the lookup keys (grouping the entries of the large
 * tables by row: build keys + segmented radix sort) can run on another stream while the rest of the forward and
 * backward pass executes, and only the reduction waits for dE:
 *   swr_embed_bwd_sort   : output pointers of the slots may be null; leaves the sorted entries in `workspace`;
 *   swr_embed_bwd_reduce : same arguments as swr_embed_bwd, same `workspace` (ordered after the sort by the caller).
 * swr_embed_bwd == sort followed by reduce on one stream.  Small dense tables never enter the sort (they are summed
 * in LDS per workgroup), whichever entry point is used.
 *   swr_embed_bwd_reduce_part : the reduce in two pieces again, for the data-parallel step, which sends the large
 *   tables' row lists to the other ranks while the rest still computes -- `part` 1: the sorted entries are reduced and
 *   the mode-1 (urow, ugrad) lists written; 2 (after 1): the small tables' direct sums and every dense gradient;
 *   3 = swr_embed_bwd_reduce. */
int swr_embed_bwd_sort(const swr_embed_grad_slot* slots_host, int n_slots, const uint32_t* keys, int64_t B,
                       void* workspace, size_t workspace_bytes, void* stream);
int swr_embed_bwd_reduce(const swr_embed_grad_slot* slots_host, int n_slots, const uint32_t* keys,
                         const float* dE, int64_t ld, int64_t B,
                         void* workspace, size_t workspace_bytes, uint32_t* err_flag, void* stream);
int swr_embed_bwd_reduce_part(const swr_embed_grad_slot* slots_host, int n_slots, const uint32_t* keys,
                              const float* dE, int64_t ld, int64_t B, int part,
                              void* workspace, size_t workspace_bytes, uint32_t* err_flag, void* stream);

/* ------------------------------------------------------------------ K2 ----
 * fp32 matrix products on the f32 MFMA pipe (v_mfma_f32_32x32x2_f32: exact
 * fp32 fused multiply-add chains in k order).  They replace aten::addmm / mm
 * of nn.Linear forward and backward (basic/layers.py:253-260) and of STAR's
 * `x @ (W_s * W_d)` (models/multi_domain/star.py:103-107).
 * `groups` > 1 runs independent problems (per-domain towers / experts) in one
 * launch; `gs*` are the element strides between consecutive groups.
 *
 *   nt:  C[m, n] = sum_k A[m, k] * B[n, k] (+ bias[n])      Linear forward, dX of [in,out] weights
 *   nn:  C[m, n] = sum_k A[m, k] * B[k, n] (+ bias[n])      dX of Linear, forward of [in,out] weights
 *   tn:  C[k1, k2] = sum_m A[m, k1] * B[m, k2]              weight gradients (reduction over the batch)
 *
 * nt / nn options:
 *   a_scale / a_shift (nullable, [K] per group): the A operand is read as
 *     relu?(a_scale[k] * A[m,k] + a_shift[k]) -- the previous layer's
 *     BatchNorm + ReLU applied on the fly;
 *   stat_partials (nullable): per 32-row tile and output column the pair
 *     (mean, M2 = sum (c - mean)^2) of the tile's rows, laid out
 *     [ceil(M/32)][groups][N][2] (= [tiles][groups*N][2], so the
 *     groups of one launch finalise as ONE BatchNorm over groups*N columns);
 *     swr_bn_finalize merges them (Chan) in fp64;
 *   accumulate != 0: C += result (gradient fan-in).
 * tn: the batch is split over workgroups; partial products go to `workspace`
 *   and are summed in a fixed order (deterministic).  `colsum` (nullable,
 *   [groups][K1]) additionally receives sum_m A[m, k1] (bias gradient). */
typedef struct {
    int64_t M;
    int32_t N, K;
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    const float* bias;
    float* C; int64_t ldc;
    const float* a_scale; const float* a_shift; int32_t a_relu;
    int32_t accumulate;
    float* stat_partials;
    int32_t groups;
    int64_t gsA, gsB, gsC, gsBias, gsScale;
    const void* B_split;   /* nt only, nullable: B already split into three bf16 planes by swr_split_weights
                              ([3][>= N rows][ld_split] bf16, zero-padded to a multiple of 32 columns); the bf16-split
                              kernel then stages it with plain copies instead of re-splitting B in every workgroup */
    int64_t ld_split, plane_stride;   /* in bf16 elements */
    int32_t n_compute;     /* 0 = all: columns n >= n_compute of C are written as ZERO without being computed (whole
                              32-column tiles are skipped; nt bf16-split kernel, one group).  dX of a layer that sits on
                              the embedding concat: the trailing dense-feature columns take no gradient */
    int32_t a_exact_from;  /* 0 = none: columns k >= a_exact_from of A hold values that are exactly representable in bf16
                              (the one-hot block of a folded first layer: 0 / 1).  The bf16-split kernel then issues
                              three products instead of six for those k and skips their split; same bits either way */
    int32_t c_act;         /* nt / nn: 0 none, 1 ReLU, 2 sigmoid applied to (product + bias) as C is stored -- the activation of a
                              layer without BatchNorm (GateNU, layers.py:314-320) without a pass of its own; not with
                              `accumulate` or `stat_partials` */
    int32_t pad1;
} swr_gemm_args;

int swr_gemm_nt(const swr_gemm_args* args_host, void* stream);
int swr_gemm_nn(const swr_gemm_args* args_host, void* stream);
/* W[N, K] fp32 -> the three bf16 planes of W (`planes`, rows = N, columns = K padded to ld = swr_split_ld(K)) and / or
 * of W^T (`planes_t`, rows = K, columns = N padded to ld_t = swr_split_ld(N)); x = h + m + l with h = bf16(x),
 * m = bf16(x - h), l = bf16(x - h - m).  Either output may be null.  Plane p starts at p * rows * ld. */
int64_t swr_split_ld(int64_t cols);
/* Which arithmetic the nt / tn products use (environment SWR_GEMM, read once): 1 = the default, every fp32 product as six
 * bf16 MFMA products (fp32-class accuracy: the parity path); 0 = f32 MFMA; 2 = ONE bf16 product per k-group (operands
 * rounded to bf16) -- the perf mode of the north star's "MFMA bf16 for the dense expert GEMMs", outside the 1e-4 logit
 * tolerance by two orders of magnitude (measured: tests/test_perf_mode_gpu.py) and never used for parity. */
int swr_gemm_precision_mode(void);
int swr_split_weights(const float* W, int64_t ldw, int N, int K, void* planes, void* planes_t, void* stream);

typedef struct {
    int64_t M;
    int32_t K1, K2;
    const float* A; int64_t lda;   /* [M, K1] */
    const float* B; int64_t ldb;   /* [M, K2] */
    float* C; int64_t ldc;         /* [K1, K2] */
    float* colsum;                 /* nullable [K1] */
    int32_t accumulate;
    int32_t groups;
    int64_t gsA, gsB, gsC, gsColsum;
    /* optional second destination (one group only): columns >= c2_from of the product are written to
     * C2[k1, k2 - c2_from] (leading dimension ldc2; always overwritten) instead of C -- the weight-gradient product over
     * an embedding concat that carries one-hot columns behind the embeddings (swr_embed_gather_fwd `oh_*`) yields dW
     * and the per-row segment sums of dZ in one launch.  C2 = NULL: everything goes to C. */
    float* C2;
    int64_t ldc2;
    int64_t c2_from;
} swr_gemm_tn_args;

/* out[g][k][n] = in[g][n][k] for `groups` contiguous [N, K] matrices: the transposed weights of a (grouped) Linear, the operand of its
 * dX product dZ W in the [N, K] layout swr_gemm_nt's bf16-split kernel stages (`basic/layers.py:253` under autograd). */
int swr_transpose_groups(const float* in, int groups, int N, int K, float* out, void* stream);
size_t swr_gemm_tn_workspace_bytes(const swr_gemm_tn_args* args_host);
int swr_gemm_tn(const swr_gemm_tn_args* args_host, void* workspace, size_t workspace_bytes, void* stream);

/* --------------------------------------------------------- BatchNorm1d ----
 * nn.BatchNorm1d(eps, momentum=0.1) of every MLP block
 * (basic/layers.py:253-258).  Training: batch mean and BIASED batch variance
 * normalise; running_var takes the UNBIASED variance.  The statistics come
 * from the producing GEMM's `stat_partials`. */
int swr_bn_finalize(const float* stat_partials, int n_tiles, int64_t M, int N,
                    const float* gamma, const float* beta, float eps, float momentum,
                    float* running_mean, float* running_var,
                    int64_t* num_batches_tracked, int n_tracked /* counters to bump (fused BN modules) */,
                    float* mean, float* rstd, float* scale, float* shift, void* stream);
/* Statistics of a tensor that no GEMM of this library produced (STAR's partitioned normalisation of the embedding,
 * star.py:91-98): per 32-row tile and column the pair (mean, M2) in the layout swr_bn_finalize merges,
 * stat_partials[ceil(M/32)][N][2]. */
int swr_col_moments(const float* X, int64_t ldx, int64_t M, int N, float* stat_partials, void* stream);
/* eval mode: scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale */
int swr_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, int N, float* scale, float* shift, void* stream);

/* activation applied after the affine, per column range (basic/activation.py:27-54) */
typedef enum { SWR_ACT_NONE = 0, SWR_ACT_RELU = 1, SWR_ACT_SIGMOID = 2, SWR_ACT_SOFTMAX = 3 } swr_act;
typedef struct {
    int32_t col_lo, col_hi;   /* [col_lo, col_hi) */
    int32_t act;              /* swr_act */
    int32_t group;            /* softmax: width of each softmax group (Softmax(dim=1) of one gate) */
} swr_act_range;
#define SWR_MAX_ACT_RANGES 4

/* Y[m, n] = act(scale[n] * Z[m, n] + shift[n]); scale/shift NULL = identity affine */
int swr_affine_act_fwd(const float* Z, int64_t ldz, const float* scale, const float* shift,
                       const swr_act_range* acts_host, int n_acts,
                       float* Y, int64_t ldy, int64_t M, int N, void* stream);

/* Backward of Y = act(BN_train(Z)):
 *   pass 1  dA = act'(Y) * dY;  partials of S1[n] = sum dA, S2[n] = sum dA * xhat   ([n_tiles][N][2], 64-row tiles)
 *   finalize dgamma = S2, dbeta = S1, coefficients of dZ = ca*dA + cb*Z + cc
 *   pass 2  dZ
 * (torch native_batch_norm_backward; SURVEY.md fact 1: dZ is dense in every
 * row, other domains' rows included). */
int swr_bn_act_bwd_stats(const float* dY, int64_t lddy, const float* Y, int64_t ldy,
                         const float* Z, int64_t ldz, const float* mean, const float* rstd,
                         const swr_act_range* acts_host, int n_acts,
                         float* partials, int64_t M, int N, void* stream);
int swr_bn_bwd_finalize(const float* partials, int n_tiles, int64_t M, int N,
                        const float* gamma, const float* rstd,
                        float* dgamma, float* dbeta, int accumulate,
                        float* ca, float* cb, float* cc, void* stream);
/* dZ = ca[n] * act'(Y) * dY + cb[n] * (Z - mean[n]) + cc[n].  With ca = scale (eval-mode BN or plain
 * affine; NULL = 1) and cb = cc = mean = NULL this is the backward of swr_affine_act_fwd w.r.t. Z. */
int swr_act_bwd_apply(const float* dY, int64_t lddy, const float* Y, int64_t ldy,
                      const float* Z, int64_t ldz, const float* ca, const float* cb, const float* cc,
                      const float* mean, const swr_act_range* acts_host, int n_acts,
                      float* dZ, int64_t lddz, int64_t M, int N, void* stream);

/* ------------------------------------------------------- gate mixing ------
 * expert_pooling = sum_j gate[:, j] * expert_j  (models/multi_domain/mmoe.py:48-49,
 * ple.py:121-126,131-133).  Y holds the activated experts and gates side by side;
 * output d mixes the `n_sel` experts listed in sel[d] with the n_sel weights at
 * Y[:, g_col + d * g_stride + j].
 *   P[m, d*H + h] = sum_j Y[m, g_col + d*g_stride + j] * Y[m, x_col + sel[d][j]*H + h] */
#define SWR_MIX_MAX_OUT 16
#define SWR_MIX_MAX_SEL 16
typedef struct {
    int32_t n_out, n_sel, H;
    int32_t x_col, g_col, g_stride;
    uint8_t sel[SWR_MIX_MAX_OUT][SWR_MIX_MAX_SEL];
} swr_mix_desc;

/* G (ABI 7): NULL -- the gate probabilities sit in Y at g_col, as above; else a tensor of their own [M, ldg] with g_col
 * counted in IT (PLE: the gates are columns of the first layer's output while the experts run a second layer,
 * ple.py:107-125 -- no concatenation pass).  With G the shapes must take the 16-byte path (H, x_col, row strides multiples
 * of 4 floats, 16-byte aligned Y / P / dP / dY): SWR_ERR_UNSUPPORTED otherwise. */
int swr_moe_mix_fwd(const swr_mix_desc* desc_host, const float* Y, int64_t ldy, const float* G, int64_t ldg,
                    float* P, int64_t ldp, int64_t M, void* stream);
/* writes (accumulate = 0) or adds to (accumulate != 0) dY[:, expert and gate columns]; experts that no
 * output selects get a zero gradient.  G / dG: both NULL, or the gate tensor and the tensor that takes its gradient */
int swr_moe_mix_bwd(const swr_mix_desc* desc_host, const float* dP, int64_t lddp,
                    const float* Y, int64_t ldy, const float* G, int64_t ldg, float* dY, int64_t lddy,
                    float* dG, int64_t lddg, int accumulate, int64_t M, void* stream);

/* --------------------------------------- BatchNorm + activation + gate mix -----
 * One MMoE level (mmoe.py:44-49) without materialising the activated experts / gate probabilities:
 *   Z[:, 0 : ne*H] experts (BN -> ReLU), Z[:, ne*H : ne*H + D*ne] gates (BN -> softmax over ne),
 *   P[:, o*H:(o+1)*H] = sum_j gate_o[j] * expert_j          (identity selection: every output mixes every expert)
 * swr_bnmix_fwd: Z, scale, shift -> P.   swr_bnmix_bwd: dP, Z, scale, shift, mean, rstd -> dY = dL/d(BN output)
 * [M, ne*H + D*ne] and bn_partials [ceil(M/T)][ne*H + D*ne][2], T = swr_bnmix_tile_rows(), for swr_bn_bwd_finalize; swr_act_bwd_apply without
 * activations then gives dZ.  ne <= 8, D <= 8, H in {16, 32} (swr_bnmix_supported); other shapes:
 * swr_affine_act_fwd + swr_moe_mix_* + swr_bn_act_bwd_stats. */
typedef struct {
    int64_t M;
    int32_t ne, H, D, pad;
    const float* Z;  int64_t ldz;
    const float* scale; const float* shift;       /* [ne*H + D*ne], 16-byte aligned */
    float* P;        int64_t ldp;                 /* forward out [M, >= D*H] */
    const float* dP; int64_t lddp;                /* backward in */
    const float* mean; const float* rstd;         /* backward */
    float* dY;       int64_t lddy;                /* backward out */
    float* bn_partials;                           /* backward out */
    float* G;                                     /* [M, D*ne] gate probabilities: written by fwd, read by bwd (nullable:
                                                     the backward then recomputes them) */
} swr_bnmix_args;

int swr_bnmix_supported(int ne, int H, int D);
int swr_bnmix_tile_rows(void);
int swr_bnmix_fwd(const swr_bnmix_args* args_host, void* stream);
int swr_bnmix_bwd(const swr_bnmix_args* args_host, void* stream);

/* ------------------------------------------------------ tower heads -------
 * G independent [Linear(K, H) -> BatchNorm1d(H) (batch statistics) -> ReLU -> Linear(H, 1)] evaluated together: the
 * per-domain `towers` of mmoe.py:38-41,50-51 (ple.py, sharebottom.py alike) with tower_params = {"dims": [H]}, each
 * reading its own K columns of X (column block g).  One kernel per pass over the batch, thread = (row, tower); the
 * post-activation values are never stored (the backward recomputes them from Z1).  K in {8,16,32}, H in
 * {4,8,16,32} (swr_tower_supported); other shapes go through swr_gemm_* + swr_bn_* + swr_affine_act_*.
 *   swr_tower_fwd_linear : Z1[M, G*H] = X_g W1_g^T + b1 (+ stat_partials [ceil(M/32)][G*H][2] for swr_bn_finalize)
 *   swr_tower_fwd_head   : V[M, G]    = relu(scale * Z1 + shift)_g . w2_g + b2_g
 *   swr_tower_head_select_bce_fwd : the output layer of each row's OWN tower + sigmoid + domain select (mmoe.py:51-55) +
 *                          mean BCE (ctr_trainer.py:56,70) in one launch: p[M], loss; the bits of swr_tower_fwd_head +
 *                          swr_select_bce_fwd.  workspace: swr_bce_workspace_bytes(M); ticket as swr_select_bce_fwd
 *   swr_tower_bwd        : from dV[M, G]: dgamma, dbeta, dw2[G*H], db2[G] (written, or added when `accumulate`),
 *                          dZ1[M, G*H] (feed swr_gemm_tn for dW1 / db1) and dX[M, G*K] (skipped when null).
 *                          ca / cb / cc: [G*H] scratch for the BatchNorm backward coefficients. */
typedef struct {
    int64_t M;
    int32_t G, K, H, accumulate;
    const float* X;  int64_t ldx;                 /* [M, >= G*K] */
    const float* W1; const float* b1;             /* [G][H][K], [G][H] (b1 nullable) */
    float* Z1;       int64_t ldz;                 /* [M, >= G*H], ldz % 4 == 0 */
    float* stat_partials;                         /* fwd_linear: nullable */
    const float* scale; const float* shift;       /* [G*H] BatchNorm as an affine map (swr_bn_finalize) */
    const float* mean;  const float* rstd; const float* gamma;   /* backward */
    const float* w2; const float* b2;             /* [G][H], [G] (b2 nullable) */
    float* V;        int64_t ldv;                 /* fwd_head: [M, >= G] */
    const float* dV; int64_t lddv;                /* bwd */
    float* ca; float* cb; float* cc;              /* [G*H] each */
    float* dgamma; float* dbeta; float* dw2; float* db2;        /* nullable */
    float* dZ1;      int64_t lddz;
    float* dX;       int64_t lddx;                /* nullable */
    /* selected mode of swr_tower_bwd (dV == NULL): the towers fed swr_tower_head_select_bce_fwd, so
       dV[m, g] = (domain[m] == g) ? d(mean BCE)/d(logit)(p[m], y[m]) * dloss : 0 is computed on the fly, never stored */
    const void* sel_domain; const void* sel_y;    /* [M] each */
    const float* sel_p;                           /* [M] selected probabilities (forward output) */
    const float* sel_dloss;                       /* device scalar: gradient arriving on the mean loss */
    int32_t sel_dom_dtype, sel_y_dtype;
} swr_tower_args;

int swr_tower_head_select_bce_fwd(const float* Z1, int64_t ldz, int G, int H, const float* scale, const float* shift,
                                  const float* w2, const float* b2, const void* domain, int dom_dtype, const void* y,
                                  int y_dtype, int64_t M, float* p, float* loss, void* workspace, size_t workspace_bytes,
                                  uint32_t* ticket, void* stream);
/* ABI 8: with the optimizer's step bookkeeping as a rider (see swr_select_bce_fwd_adv) */
int swr_tower_head_select_bce_fwd_adv(const float* Z1, int64_t ldz, int G, int H, const float* scale, const float* shift,
                                      const float* w2, const float* b2, const void* domain, int dom_dtype, const void* y,
                                      int y_dtype, int64_t M, float* p, float* loss, void* workspace, size_t workspace_bytes,
                                      uint32_t* ticket, void* adv_hyper, float* adv_hist, int64_t adv_hist_cap,
                                      void* stream);
int swr_tower_supported(int K, int H);
int swr_tower_fwd_linear(const swr_tower_args* args_host, void* stream);
int swr_tower_fwd_head(const swr_tower_args* args_host, void* stream);
size_t swr_tower_bwd_workspace_bytes(int64_t M, int G, int H);
int swr_tower_bwd(const swr_tower_args* args_host, void* workspace, size_t workspace_bytes, void* stream);
/* First-layer weight gradients of the G towers in one pass: dW1 [G][H][K] (+)= dZ1_g^T X_g, db1 [G][H] (+)= column sums of dZ1
 * (the `towers` of mmoe.py:38-41, the autograd of their first nn.Linear) -- replaces the grouped swr_gemm_tn on (dZ1, X).
 * dZ1 [M][ldz >= G*H], X [M][ldx >= G*K], 16-byte aligned rows; db1 may be NULL; workspace: swr_tower_dw_workspace_bytes. */
int swr_tower_dw_supported(int K, int H, int G);
size_t swr_tower_dw_workspace_bytes(int64_t M, int K, int H, int G);
int swr_tower_dw(const float* dZ1, int64_t ldz, const float* X, int64_t ldx, int64_t M, int K, int H, int G, float* dW1, float* db1,
                 int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------- domain select + loss ------
 * final = 0; for d: final = where(domain_id == d, y_d, final)
 * (mmoe.py:53-55, base_example.py:72-74): integer equality on the raw id, ids
 * outside [0, D) give exactly 0.0.  `apply_sigmoid`: y_d = sigmoid(V[:, d])
 * (towers, mmoe.py:51) or V[:, d] as is.  `extra` (nullable, [M]):
 * out = sigmoid(select(V) + extra), STAR's `sig(final + aux_out)` (star.py:117). */
int swr_select_fwd(const float* V, int64_t ldv, int D, const void* domain, int dom_dtype,
                   int apply_sigmoid, const float* extra, float* out, int64_t M, void* stream);
int swr_select_bwd(const float* dout, const float* out, int D, const void* domain, int dom_dtype,
                   int apply_sigmoid, int has_extra, float* dV, int64_t lddv, float* dextra,
                   int64_t M, void* stream);

/* torch.nn.BCELoss(reduction='mean') on probabilities (trainers/ctr_trainer.py:56,70):
 * logs clamped at -100; backward (p - y) / max(p (1 - p), 1e-12) / M * dloss. */
size_t swr_bce_workspace_bytes(int64_t M);
int swr_bce_fwd(const float* p, const void* y, int y_dtype, int64_t M, float* loss,
                void* workspace, size_t workspace_bytes, void* stream);
int swr_bce_bwd(const float* p, const void* y, int y_dtype, int64_t M, const float* dloss,
                float* dp, void* stream);
/* swr_select_fwd(apply_sigmoid = 1, no extra) + swr_bce_fwd in ONE launch, and their two backward kernels in one:
 * the model's final domain select (mmoe.py:51-55) feeding the trainer's criterion (ctr_trainer.py:70) directly.
 * Same arithmetic and the same summation trees as the separate entry points (bit-identical p, loss and dV).
 * `workspace`: swr_bce_workspace_bytes(M).  `ticket`: one device word owned by the caller, zero before the first
 * call; the kernel leaves it zero (the workgroup that finishes last does the final sum; nothing spins). */
int swr_select_bce_fwd(const float* V, int64_t ldv, int D, const void* domain, int dom_dtype,
                       const void* y, int y_dtype, int64_t M, float* p, float* loss,
                       void* workspace, size_t workspace_bytes, uint32_t* ticket, void* stream);
int swr_select_bce_bwd(const float* p, const void* y, int y_dtype, int D, const void* domain,
                       int dom_dtype, int64_t M, const float* dloss, float* dV, int64_t lddv, void* stream);
/* ABI 8: the same launch carrying the optimizer's step bookkeeping as a RIDER (optim.Adam's `state["step"] += 1` and the bias
 * corrections of the step, ctr_trainer.py:73 through torch/optim/adam.py): when `adv_hyper` (a swr_adam_hyper*, declared further
 * down) is not NULL the one thread that finishes
 * the launch also does what swr_adam_advance(adv_hyper, adv_hist, adv_hist_cap) does -- the loss kernel reads none of those fields,
 * every reader of them (the lookup's lazy-row catch-up before, the optimizer's kernels after) sits behind a kernel boundary --
 * and the 1-thread swr_adam_advance launch (~5 us of a launch-bound short-batch step) is not needed.  All three NULL / 0: exactly
 * swr_select_bce_fwd. */
int swr_select_bce_fwd_adv(const float* V, int64_t ldv, int D, const void* domain, int dom_dtype,
                           const void* y, int y_dtype, int64_t M, float* p, float* loss,
                           void* workspace, size_t workspace_bytes, uint32_t* ticket,
                           void* adv_hyper, float* adv_hist, int64_t adv_hist_cap, void* stream);

/* -------------------------------------------------------- elementwise -----
 * small helpers of STAR / PPNet / HAMUR middles */
/* C = A * B (elementwise, same shape) */
int swr_mul_fwd(const float* A, const float* B, float* C, int64_t n, void* stream);
/* C = A * (scale * B), contiguous [n]: `hidden * gate_out` with the GateNU's gamma folded in (ppnet.py:27, layers.py:318-320) */
int swr_mul_scale_fwd(const float* A, const float* B, float scale, float* C, int64_t n, void* stream);
/* its backward in one pass: dA = dC * (scale * B), dB = dC * (scale * A) (the bits of two swr_mul_scale_fwd calls) */
int swr_mul_scale_bwd(const float* dC, const float* A, const float* B, float scale, float* dA, float* dB, int64_t n,
                      void* stream);
/* C = A * (scale * sigmoid(Z)) and its backward (dA = dC * (scale * y), dZ = ((dC * (scale * A)) * y) * (1 - y)): the output
 * activation of a GateNU and the gating product without the gate tensor in between (ppnet.py:27, layers.py:318-320); the
 * bits of swr_affine_act_fwd(sigmoid) + swr_mul_scale_fwd / swr_mul_scale_bwd + swr_act_bwd_apply(sigmoid). */
int swr_mul_sigmoid_fwd(const float* A, const float* Z, float scale, float* C, int64_t n, void* stream);
int swr_mul_sigmoid_bwd(const float* dC, const float* A, const float* Z, float scale, float* dA, float* dZ, int64_t n,
                        void* stream);
/* C = A + B, contiguous [n]: residual connections (hamur.py:197,366 `adapter + h`; m3oe.py:147 `star_mlp(emb) + skip`) */
int swr_add_fwd(const float* A, const float* B, float* C, int64_t n, void* stream);
/* column sums: out[n] (+)= sum_m X[m, n]  (bias gradients of layers without BatchNorm) */
/* out[b, d, :] = T[b, d, :] @ Hm[b] with a k x k matrix per sample (contiguous [B, D, k] / [B, k, k]): the per-sample
 * middle factor of HAMUR's adapter weights U H_b V (hamur.py:175-186, 344-355), applied as ((h U) H_b) V.
 * bwd: dT[b,d,i] = sum_j dOut[b,d,j] Hm[b,i,j]; dHm[b,i,j] (+)= sum_d T[b,d,i] dOut[b,d,j] (either may be null;
 * accumulate_dhm != 0 adds into dHm: H_b feeds both products of an adapter cell, hamur.py:177,186, and the second
 * backward adds onto the first instead of leaving two 160 MB gradients for autograd to sum). */
int swr_rowmat_fwd(const float* T, const float* Hm, float* out, int64_t B, int D, int k, void* stream);
int swr_rowmat_bwd(const float* dOut, const float* T, const float* Hm, float* dT, float* dHm, int accumulate_dhm,
                   int64_t B, int D, int k, void* stream);
size_t swr_colsum_workspace_bytes(int64_t M, int N);
int swr_colsum(const float* X, int64_t ldx, int64_t M, int N, float* out, int accumulate,
               void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------- Adam -------
 * torch.optim.Adam(lr, betas, eps, weight_decay) of the reference trainer
 * (trainers/ctr_trainer.py:50-52,73): L2 decay added to the gradient, bias
 * corrections from the step count.  The step count, learning rate and the
 * derived scalars live in device memory (`swr_adam_hyper`) so that a captured
 * hipGraph replays correctly: swr_adam_advance bumps the step and recomputes
 * them in fp64 on the device. */
typedef struct {
    double lr, beta1, beta2, eps, weight_decay;
    int64_t step;
    float step_size;        /* lr / (1 - beta1^step) */
    float inv_bc2_sqrt;     /* 1 / sqrt(1 - beta2^step) */
    float one_minus_b1, b2, one_minus_b2, eps_f, wd_f;
    uint32_t hist_mask;     /* hist_cap - 1 of the last swr_adam_advance (0: no history) */
} swr_adam_hyper;

/* `hist` (nullable, float [hist_cap][2], device; hist_cap a power of two): step s writes its (step_size,
 * inv_bc2_sqrt) to hist[s % hist_cap] -- a ring; the lazy row updates below replay them.  The caller must flush every
 * lazily updated table (swr_adam_flush) before any of its rows lags hist_cap - 1 steps. */
int swr_adam_advance(swr_adam_hyper* hyper_dev, float* hist, int64_t hist_cap, void* stream);
/* dense update of a flat fp32 arena; clear_grad != 0: g is zeroed behind the update (the optimizer consumed it: the next
 * step's zero_grad -- optimizer.zero_grad(), ctr_trainer.py:71 -- then has nothing to fill) */
int swr_adam_dense(float* p, float* g, float* m, float* v, int64_t n, int clear_grad,
                   const swr_adam_hyper* hyper_dev, void* stream);
/* large tables with row-sparse gradients (mode 1 of swr_embed_bwd): touched rows take the summed
 * gradient, every other row takes g = weight_decay * p -- together exactly the dense update the
 * reference performs.  `bitmap` ([ceil(vocab/32)] words, zero on entry, zero on
 * exit) marks touched rows between the two launches. */
int swr_adam_rows(float* p, float* m, float* v, int64_t vocab, int dim,
                  const int32_t* urow, const float* ugrad, int64_t n_entries,
                  uint32_t* bitmap /* sweep mode, nullable */, int32_t* last /* lazy mode, nullable */,
                  const swr_adam_hyper* hyper_dev, void* stream);
/* swr_adam_dense + swr_adam_rows (lazy mode) in ONE launch: the common step of one parameter arena + one large table */
int swr_adam_dense_rows(float* dense_p, float* dense_g, float* dense_m, float* dense_v, int64_t dense_n, int clear_grad,
                        float* p, float* m, float* v, int64_t vocab, int dim,
                        const int32_t* urow, const float* ugrad, int64_t n_entries, int32_t* last,
                        const swr_adam_hyper* hyper_dev, void* stream);
int swr_adam_sweep_untouched(float* p, float* m, float* v, int64_t vocab, int dim,
                             uint32_t* bitmap, int clear_bitmap /* 1: zero the bitmap afterwards (normal use) */,
                             const swr_adam_hyper* hyper_dev, void* stream);

/* Lazy but EXACT alternative to the sweep: a row that is not looked up still takes g = weight_decay * p every step,
 * an update that depends only on the row's own (p, m, v) and the step's scalars -- so it is replayed, bit for bit,
 * when the row is next looked up (`swr_adam_catchup_rows`, called by the forward lookup before the gather) or when
 * the table is materialised (`swr_adam_flush`, checkpoints).  `last[vocab]` (int32, zero-initialised) holds the step
 * each row is current for.  Workspace of catch-up: 8 * n bytes. */
int swr_adam_catchup_rows(float* p, float* m, float* v, int32_t* last, int32_t* claim /* [vocab] scratch, any contents */,
                          int64_t vocab, int dim,
                          const void* idx, int idx_dtype, uint32_t hash_seed, int64_t n,
                          const float* hist, const swr_adam_hyper* hyper_dev,
                          void* workspace, size_t workspace_bytes, void* stream);
int swr_adam_flush(float* p, float* m, float* v, int32_t* last, int64_t vocab, int dim,
                   const float* hist, const swr_adam_hyper* hyper_dev, void* stream);

/* Several large (lazily updated) tables in one launch each: swr_adam_catchup_multi = swr_adam_catchup_rows for every
 * listed table (one claim launch + one replay launch in all), swr_adam_rows_multi = swr_adam_rows in lazy mode for
 * every listed table.  Per entry the arithmetic is the single-table kernels'.  `from` / `rows` ([n] each) are the
 * per-table workspace of the catch-up. */
#define SWR_ADAM_MAX_TABLES 16
typedef struct {
    float* p; float* m; float* v;      /* [vocab, dim] */
    int32_t* last;                     /* [vocab] */
    int32_t* claim;                    /* [vocab] scratch (catch-up) */
    int64_t vocab;
    int32_t dim;
    int32_t idx_dtype;                 /* catch-up: type of idx */
    const void* idx;                   /* catch-up: [n] looked-up ids */
    uint32_t hash_seed; uint32_t pad;
    int64_t n;                         /* entries: looked-up ids (catch-up) or row-list entries (rows) */
    int32_t* from_step; uint32_t* rows;   /* catch-up workspace, [n] each */
    const int32_t* urow; const float* ugrad;   /* rows: the row list ([n], [n, dim]) */
} swr_adam_table;
int swr_adam_catchup_multi(const swr_adam_table* tables_host, int n_tables, const float* hist,
                           const swr_adam_hyper* hyper_dev, void* stream);
int swr_adam_rows_multi(const swr_adam_table* tables_host, int n_tables, const swr_adam_hyper* hyper_dev, void* stream);

/* ---------------------------------------------------------------- STAR ----
 * STAR's factorised weights (reference models/multi_domain/star.py:99-107): per layer, for ALL domains in one launch
 * each way, the effective weights W_s (.) W_d[d] in Linear layout [out, in] and biases b_s + b_d[d] -- the first layer
 * (`first` = 1) also folds in the domain affine of the partitioned normalisation (gamma_s gamma_d[d], beta_s +
 * beta_d[d]; star.py:91-100) -- and, backwards, every parameter gradient from the gradients of the effective tensors.
 * Parameters are in the reference's layout: weights [in, out], biases [out], affine vectors [in].  Gradient pointers
 * may be null (not needed); `accumulate` = 1 adds into them (gradient arena), 0 overwrites.  dW_eff / db_eff may be
 * null per domain (no gradient arrived from it). */
#define SWR_STAR_MAX_DOMAINS 8
typedef struct {
    int32_t D, in_dim, out_dim, first;
    const float* Ws;
    const float* bs;
    const float* Wd[SWR_STAR_MAX_DOMAINS];
    const float* bd[SWR_STAR_MAX_DOMAINS];
    const float* gamma_s;
    const float* beta_s;
    const float* gamma_d[SWR_STAR_MAX_DOMAINS];
    const float* beta_d[SWR_STAR_MAX_DOMAINS];
    float* W_eff[SWR_STAR_MAX_DOMAINS];       /* forward outputs [out, in] */
    float* b_eff[SWR_STAR_MAX_DOMAINS];       /* [out] */
    const float* dW_eff[SWR_STAR_MAX_DOMAINS];   /* backward inputs */
    const float* db_eff[SWR_STAR_MAX_DOMAINS];
    float* dWs;
    float* dbs;
    float* dWd[SWR_STAR_MAX_DOMAINS];
    float* dbd[SWR_STAR_MAX_DOMAINS];
    float* dgamma_s;
    float* dbeta_s;
    float* dgamma_d[SWR_STAR_MAX_DOMAINS];
    float* dbeta_d[SWR_STAR_MAX_DOMAINS];
    int32_t accumulate;
    int32_t pad;
} swr_star_layer_args;
int swr_star_layer_fwd(const swr_star_layer_args* args, void* stream);
int swr_star_layer_bwd(const swr_star_layer_args* args, void* stream);
/* the same for n_layers layers (host array): their launches merged, four layers per launch -- every effective weight of the
 * FCN stack before its first product, every parameter gradient after its last (star.py:99-110 over all layers) */
int swr_star_layers_fwd(const swr_star_layer_args* layers_host, int n_layers, void* stream);
int swr_star_layers_bwd(const swr_star_layer_args* layers_host, int n_layers, void* stream);

/* ------------------------------------------------------------ routed inference ----
 * MMoE head in eval mode, routed: every row mixes the experts with its OWN domain's gate
 * probabilities and runs its own domain's tower [Linear(H, T) -> BatchNorm(eval) -> ReLU -> Linear(T, 1)] -> sigmoid;
 * the reference evaluates every domain's mix and tower on the whole batch and selects afterwards
 * (models/multi_domain/mmoe.py:48-55) -- identical results in eval mode, D x less tower work, one launch.
 *   Y [M, >= ne H + D ne]: expert outputs (ne blocks of H columns) then the D gates' softmax probabilities (ne each);
 *   W1 [D][T][H], b1 [D][T], scale1 / shift1 [D][T] (BatchNorm eval affine, swr_bn_eval_coeffs), w2 [D][T], b2 [D];
 *   out [M]: probabilities; 0.0 for a domain id outside [0, D). */
int swr_routed_mmoe_eval_supported(int n_expert, int H, int D, int T);
int swr_routed_mmoe_eval(const float* Y, int64_t ldy, int64_t M, int n_expert, int H, int D, int T,
                         const float* W1, const float* b1, const float* scale1, const float* shift1,
                         const float* w2, const float* b2, const void* domain, int domain_dtype, float* out,
                         void* stream);

/* ------------------------------------------------------------ LayerNorm / block select (M3oE) ----
 * torch.nn.LayerNorm(N, eps) (+ ReLU) over G side-by-side column groups of X [M, G*N], each group with its own
 * gamma / beta (vectors of length G*N): the [Linear, LayerNorm, ReLU] blocks and towers of M3oE (reference
 * models/multi_domain/m3oe.py:49-67,121-128).  Biased variance, eps inside the square root.  `mean` / `rstd` ([M, G],
 * nullable in eval) are saved for the backward, which recomputes the ReLU mask from X.  Backward: dX (nullable),
 * dgamma / dbeta (nullable; `accumulate` = 1 adds into them -- gradient arena) as deterministic fixed-order column sums.
 * Workspace of the backward: swr_layernorm_bwd_workspace_bytes. */
typedef struct {
    int64_t M;
    int32_t G, N, relu, accumulate;
    float eps;
    int32_t pad;
    const float* X; int64_t ldx;
    const float* gamma; const float* beta;
    float* Y; int64_t ldy;
    float* mean; float* rstd;
    const float* dY; int64_t lddy;
    float* dX; int64_t lddx;
    float* dgamma; float* dbeta;
} swr_layernorm_args;
int swr_layernorm_fwd(const swr_layernorm_args* a, void* stream);
size_t swr_layernorm_bwd_workspace_bytes(int64_t M, int G, int N);
int swr_layernorm_bwd(const swr_layernorm_args* a, void* workspace, size_t workspace_bytes, void* stream);
/* out[b, 0:H] = V[b, d_b*H : (d_b+1)*H] when 0 <= d_b < D, else zeros (m3oe.py:141-146: the STAR front's
 * `emb = where(mask_d, output_d, emb)`; exact integer compare on the raw domain id); backward scatters dout into the
 * owning block of dV and zero-fills the rest. */
int swr_block_select_fwd(const float* V, int64_t ldv, const void* domain, int dom_dtype, int D, int H, int64_t M,
                         float* out, int64_t ldo, void* stream);
int swr_block_select_bwd(const float* dout, int64_t ldo, const void* domain, int dom_dtype, int D, int H, int64_t M,
                         float* dV, int64_t ldv, void* stream);

/* ------------------------------------------------------------ input columns ----
 * Row permutation of a columnar, device-resident dataset (replaces DataLoader(shuffle=True) over
 * TorchDataset, whose __getitem__ builds one python dict per ROW, utils/data.py:11-22,55): for every column c,
 * dst[c][i] = src[c][perm[i]], i < n_out -- all columns in ONE launch, element sizes 1 / 2 / 4 / 8 bytes.  Byte copy,
 * bit-exact; perm values must lie in [0, n_in) (out-of-range entries are clamped and raise SWR_FLAG_INDEX_OOR).
 * `perm` = NULL copies rows [0, n_out) of every column as they are (a batch into the input buffers of a captured step:
 * one launch for all columns + the label). */
#define SWR_TAKE_MAX_COLUMNS 96
typedef struct {
    const void* src;
    void* dst;
    int32_t elem_bytes;    /* 1, 2, 4 or 8 */
    int32_t pad;
} swr_take_column;
int swr_take_rows(const swr_take_column* columns_host, int n_columns, const int64_t* perm, int64_t n_in, int64_t n_out,
                  uint32_t* err_flag, void* stream);

/* ------------------------------------------------------------- metrics ----
 * Evaluation metrics on the device: replaces `.tolist()` + sklearn.metrics.log_loss /
 * roc_auc_score of CTRTrainer.evaluate / evaluate_multi_domain_loss (trainers/ctr_trainer.py:99-165).
 *   prob [n] fp32 predictions, label [n] (any value dtype; > 0.5 = positive), domain [n] (any index dtype).
 *   counts [(n_domains + 1) * 3] uint64: per domain d (slot n_domains = ALL rows, whatever their domain id):
 *       rows, positives, 2U -- U the Mann-Whitney statistic with ties counted 1/2, so AUC = 2U / (2 P N), exactly
 *       the area sklearn's trapezoid rule gives; integer arithmetic, bitwise reproducible;
 *   logloss_sum [n_domains + 1] fp64: sum of -(y log p + (1 - y) log(1 - p)), p clipped to [2^-52, 1 - 2^-52]
 *       (sklearn's clip of a float64 array), summed in a fixed order.
 * The host divides (and reports None for an empty domain / raises for a single-class one, as the reference's
 * sklearn calls do).  n < 2^31, n_domains <= 254. */
size_t swr_eval_metrics_workspace_bytes(int64_t n, int n_domains);
int swr_eval_metrics(const float* prob, const void* label, int label_dtype, const void* domain, int domain_dtype,
                     int64_t n, int n_domains, unsigned long long* counts, double* logloss_sum,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------ exchange ----
 * Device side of the data-parallel exchange step: what torch.nn.DataParallel's gradient reduction
 * does for the reference (trainers/ctr_trainer.py:45-47 -- replica gradients summed in device order), after the
 * all-gather(s) have put every rank's message side by side.  Two gathered buffers (they may be one and the same):
 *   recv_dense [world][dense_stride] fp32 words -- the gradient arena (A words) at the start of each rank's message;
 *   recv_rows  [world][rows_stride]  fp32 words -- per large table the row ids (n int32) and row gradients (n x dim)
 *                                                  at `row_off` / `grad_off` of each rank's message.
 * One launch:
 *   dense_out[j]   = scale * sum_r recv_dense[r][j], j < A, ranks added in order 0..world-1;
 *   per large table: the `world` row lists (each in swr_embed_bwd's mode-1 format: ordered by row, negative = no
 *   entry) are merged WITHOUT a sort: every entry looks its row up in the other ranks' lists (binary searches); the
 *   lowest rank holding the row owns it and writes row id + scale * (sum of the holders' gradients in rank order) to
 *   its own position of out_row / out_grad ([world * n] entries); all other positions get -1 / 0.
 * Every rank computes bit-identical results from the same gathered buffers.  world <= SWR_DP_MAX_WORLD; strides are
 * multiples of 4 words and the buffers 16-byte aligned. */
#define SWR_DP_MAX_WORLD 8
#define SWR_DP_MAX_TABLES 16
typedef struct {
    int64_t row_off;       /* word offset of the row ids inside one rank's message */
    int64_t grad_off;      /* word offset of the gradients */
    int64_t n;             /* entries per rank */
    int32_t dim;
    int32_t pad;
    int32_t* out_row;      /* [world * n] */
    float* out_grad;       /* [world * n, dim] */
} swr_dp_table;
int swr_dp_finish(const float* recv_dense, int64_t dense_stride, int64_t A, float* dense_out,
                  const float* recv_rows, int64_t rows_stride, const swr_dp_table* tables_host, int n_tables,
                  int world, float scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SWR_H_ */
