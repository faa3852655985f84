// Fused lookup + first layer ("fl", include/swr.h): the lookup is the A-operand producer of the first layer's products.
//
// Reference: EmbeddingLayer.forward -> Linear of the stacked expert / gate (or shared-bottom) layer, basic/layers.py:64-105,
// 253-258, mmoe.py:37-40.  The [B, K0] concat (and the compact [B, Kp + ohw] block A' of the folded first layer) is never
// written: a lane of the product kernel fetches the 48 bytes it needs of its sample's table row through the row key.
//
//   fl_prep_kernel : parameters -> (a) B3, the folded weights Wf = [W_big | W_d | 0 | P] split into three bf16 terms
//                    (x = h + m + l) in MFMA fragment order, one linear chunk per two 16-column groups: a chunk is one
//                    LDS-DMA copy and every fragment read a conflict-free ds_read_b128 at lane * 16;
//                    (b) the bf16-term shadow of every small table, piece-major: [row][dim / 8][h | m | l] x 16 bytes, so the
//                    three terms of a lane's 8 columns are 48 consecutive bytes; (c) rows `sel` of W^T for the dX product.
//   fl_keys_kernel : ids -> row keys [n_keys][B], one-hot mask (128 bits per sample, and transposed [4][B] for the
//                    weight-gradient product), the byte offset of every (tile, group, lane)'s piece, and the pieces that
//                    come from fp32 sources (row-sparse tables, dense features) gathered + split (A3f).
//   fl_fwd_kernel  : a wave owns 32 samples and all N <= 160 columns (NT tiles of 32): per real group three 16-byte loads
//                    per lane (ring of 4 groups in flight, inline-asm loads, one counted s_waitcnt per chunk), one-hot
//                    groups expanded from 8 mask bits; weights streamed through a double-buffered LDS chunk by LDS-DMA;
//                    6 (real) / 3 (one-hot) v_mfma_f32_32x32x16_bf16 per (group, tile); epilogue = gemm.hip's.
// HBM traffic at config 2 (B = 65 536): keys launch ~17 MB of ids in, ~25 MB out; forward ~45 MB in (mostly L2 hits on
// the shadows) + 38.8 MB of Z out -- against 72 MB written + 72 MB re-read for A' before.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "rows_epilogue.h"
#include "split3.h"
#include "tn_gather.h"

#define FL_MAX_GROUPS 16
#define FL_MAX_OH_GROUPS 8
#define FL_MAX_SPARSE 64
#define FL_MAX_DENSE 32
#define FL_MAX_TABLES 32
#define FL_MAX_OHT 64
#define FL_NT_MAX 5
#define FL_THREADS 256
#define FL_PIECE_BYTES 48

// splitmix64 finaliser: the optional hash stage of the lookup (embed_fwd.hip)
__device__ __forceinline__ uint64_t fl_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct FlDevPiece {           // what the keys launch needs of a piece
    int16_t kind, slot, off, n_valid;
    int16_t fp;               // index among the pieces of kind ROWS / DENSE
    int16_t pad;
    uint32_t base;            // PLANES: byte offset of the piece's column block in row 0 of the table's shadow
    uint32_t rowbytes;        // PLANES: bytes per shadow row
};
struct FlPrepPiece {
    int16_t kind, n_valid;
    int32_t w_col;
};
struct FlTable {              // a table of kind PLANES (unique by weight pointer)
    const float* w;
    int32_t vocab, dim;
    uint32_t planes_off;      // byte offset of its shadow in the workspace
    int32_t first_item;       // first (row, piece) item of the table in the prep launch
};

static inline int fl_pitch_blocks(int nt) { return (6 * nt + 3) / 4 * 4; }     // 1-KB blocks per B3 chunk (4 waves x whole pieces)
static inline int64_t fl_align(int64_t v) { return (v + 255) / 256 * 256; }

struct FlHost {
    swr_fl_offsets o;
    int NR, NOg, ncr, nco, NT, n_tiles;
    FlDevPiece piece[2 * FL_MAX_GROUPS];
    FlTable tbl[FL_MAX_TABLES];
    int n_tbl;
    int items_planes;
    int8_t fpiece[2 * FL_MAX_GROUPS];
};

static int fl_build(const swr_fl_plan* p, FlHost& h) {
    SWR_REQUIRE(p != nullptr, SWR_ERR_ARG);
    SWR_REQUIRE(p->n_sparse >= 1 && p->n_sparse <= FL_MAX_SPARSE && p->sparse_host && p->n_dense >= 0 && p->n_dense <= FL_MAX_DENSE &&
                    (p->n_dense == 0 || p->dense_host), SWR_ERR_ARG);
    SWR_REQUIRE(p->n_real_groups >= 1 && p->n_real_groups <= FL_MAX_GROUPS && p->oh_width >= 0 && p->oh_width % 16 == 0 &&
                    p->oh_width <= 16 * FL_MAX_OH_GROUPS && p->N >= 0 && p->N <= 32 * FL_NT_MAX && p->B >= 0 &&
                    p->n_keys >= 0 && p->n_keys <= p->n_sparse && (p->oh_width == 0 || p->oh_off_host), SWR_ERR_ARG);
    SWR_REQUIRE(p->B < (1ll << 31), SWR_ERR_UNSUPPORTED);
    h.NR = p->n_real_groups;
    h.NOg = p->oh_width / 16;
    h.ncr = (h.NR + 1) / 2;
    h.nco = (h.NOg + 1) / 2;
    h.NT = p->N > 0 ? (p->N + 31) / 32 : FL_NT_MAX;        // (N = 0: not known yet -- the layout does not depend on it)
    h.n_tiles = static_cast<int>((p->B + 31) / 32);
    h.n_tbl = 0;
    h.items_planes = 0;
    int nfp = 0;
    int64_t planes = 0;
    for (int q = 0; q < 2 * h.NR; ++q) {
        const swr_fl_piece& pc = p->piece[q];
        FlDevPiece& d = h.piece[q];
        std::memset(&d, 0, sizeof(d));
        d.kind = static_cast<int16_t>(pc.kind);
        d.fp = -1;
        if (pc.kind == SWR_FL_ZERO) continue;
        SWR_REQUIRE(pc.w_col >= 0, SWR_ERR_ARG);
        if (pc.kind == SWR_FL_PLANES || pc.kind == SWR_FL_ROWS) {
            SWR_REQUIRE(pc.slot >= 0 && pc.slot < p->n_sparse, SWR_ERR_ARG);
            const swr_sparse_slot& sl = p->sparse_host[pc.slot];
            SWR_REQUIRE(sl.weight && sl.idx && sl.vocab > 0 && sl.dim > 0 && sl.dim % 8 == 0 && pc.off >= 0 && pc.off % 8 == 0 &&
                            pc.off + 8 <= sl.dim && swr_aligned16(sl.weight) && swr_is_index_dtype(sl.idx_dtype), SWR_ERR_ARG);
            SWR_REQUIRE(sl.vocab <= 0xFFFFFFFFll, SWR_ERR_UNSUPPORTED);
            d.slot = static_cast<int16_t>(pc.slot);
            d.off = static_cast<int16_t>(pc.off);
            d.n_valid = 8;
            if (pc.kind == SWR_FL_PLANES) {
                int t = 0;
                while (t < h.n_tbl && h.tbl[t].w != sl.weight) ++t;
                if (t == h.n_tbl) {
                    SWR_REQUIRE(h.n_tbl < FL_MAX_TABLES, SWR_ERR_UNSUPPORTED);
                    FlTable& T = h.tbl[h.n_tbl++];
                    T.w = sl.weight;
                    T.vocab = static_cast<int32_t>(sl.vocab);
                    T.dim = sl.dim;
                    T.planes_off = static_cast<uint32_t>(planes);      // relative to the planes section (fixed up below)
                    T.first_item = h.items_planes;
                    const int64_t bytes = sl.vocab * (sl.dim / 8) * FL_PIECE_BYTES;
                    SWR_REQUIRE(sl.vocab < (1 << 24) && planes + bytes < (1ll << 30), SWR_ERR_UNSUPPORTED);
                    planes += fl_align(bytes);
                    h.items_planes += static_cast<int>(sl.vocab) * (sl.dim / 8);
                }
                SWR_REQUIRE(h.tbl[t].dim == sl.dim && h.tbl[t].vocab == sl.vocab, SWR_ERR_ARG);
                d.base = h.tbl[t].planes_off + static_cast<uint32_t>(pc.off / 8) * FL_PIECE_BYTES;
                d.rowbytes = static_cast<uint32_t>(sl.dim / 8) * FL_PIECE_BYTES;
            } else {
                h.fpiece[nfp] = static_cast<int8_t>(q);
                d.fp = static_cast<int16_t>(nfp++);
            }
        } else if (pc.kind == SWR_FL_DENSE) {
            SWR_REQUIRE(pc.slot >= 0 && pc.n_valid >= 1 && pc.n_valid <= 8 && pc.slot + pc.n_valid <= p->n_dense, SWR_ERR_ARG);
            for (int e = 0; e < pc.n_valid; ++e)
                SWR_REQUIRE(p->dense_host[pc.slot + e].values && swr_is_value_dtype(p->dense_host[pc.slot + e].dtype), SWR_ERR_DTYPE);
            d.slot = static_cast<int16_t>(pc.slot);
            d.n_valid = static_cast<int16_t>(pc.n_valid);
            h.fpiece[nfp] = static_cast<int8_t>(q);
            d.fp = static_cast<int16_t>(nfp++);
        } else {
            return SWR_ERR_ARG;
        }
    }
    for (int s = 0; s < p->n_sparse; ++s) {
        const swr_sparse_slot& sl = p->sparse_host[s];
        SWR_REQUIRE(sl.idx && sl.vocab > 0 && sl.vocab <= 0xFFFFFFFFll && swr_is_index_dtype(sl.idx_dtype), SWR_ERR_ARG);
        if (p->oh_width > 0 && p->oh_off_host[s] >= 0)
            SWR_REQUIRE(sl.hash_seed == 0 && p->oh_off_host[s] + sl.vocab <= p->oh_width, SWR_ERR_ARG);
    }
    swr_fl_offsets& o = h.o;
    o.n_fpieces = nfp;
    o.nd4 = (p->n_dense + 3) / 4 * 4;
    int64_t at = 0;
    o.zero = at; at += 256;
    o.planes = at; at += planes;
    o.a3f = at; at += fl_align(static_cast<int64_t>(h.n_tiles) * nfp * 32 * FL_PIECE_BYTES);
    SWR_REQUIRE(at < (1ll << 32), SWR_ERR_UNSUPPORTED);                 // everything a lane's 32-bit piece offset can reach
    o.b3 = at; at += fl_align(static_cast<int64_t>(h.ncr + h.nco) * fl_pitch_blocks(FL_NT_MAX) * 1024);     // (sized for the widest layer)
    o.keys = at; at += fl_align(static_cast<int64_t>(p->n_keys) * p->B * 4);
    o.mask = at; at += fl_align(p->B * 16);
    o.mask_t = at; at += fl_align(p->B * 16);
    o.voff = at; at += fl_align(static_cast<int64_t>(h.n_tiles) * h.NR * 64 * 4);
    o.densef = at; at += fl_align(p->B * o.nd4 * 4);
    // B3X: rows `sel` of W^T (the dX product's weights) in the same fragment order: 16-column k groups over the layer's
    // N <= 160 outputs, column tiles over the <= 160 selected columns (sized for the widest)
    o.b3x = at; at += fl_align(static_cast<int64_t>(FL_NT_MAX) * fl_pitch_blocks(FL_NT_MAX) * 1024);
    o.total = at;
    for (int t = 0; t < h.n_tbl; ++t) h.tbl[t].planes_off += static_cast<uint32_t>(o.planes);
    for (int q = 0; q < 2 * h.NR; ++q)
        if (h.piece[q].kind == SWR_FL_PLANES) h.piece[q].base += static_cast<uint32_t>(o.planes);
    return SWR_OK;
}

extern "C" int swr_fl_layout(const swr_fl_plan* plan, swr_fl_offsets* out) {
    SWR_REQUIRE(out != nullptr, SWR_ERR_ARG);
    FlHost h;
    const int rc = fl_build(plan, h);
    if (rc != SWR_OK) return rc;
    *out = h.o;
    return SWR_OK;
}

extern "C" size_t swr_fl_workspace_bytes(const swr_fl_plan* plan) {
    FlHost h;
    if (fl_build(plan, h) != SWR_OK) return 0;
    return static_cast<size_t>(h.o.total);
}

// ------------------------------------------------------------------------------------------------ prep
struct FlPrepK {
    FlPrepPiece piece[2 * FL_MAX_GROUPS];
    swr_onehot_table tab[FL_MAX_OHT];           // .grad = the table's weights
    FlTable tbl[FL_MAX_TABLES];
    int n_tbl, items_planes;
    int NR, NOg, ncr, nco, NT, N, pitch_blocks;
    const float* W; int64_t ldw;
    const int32_t* oh_table;
    const int64_t* sel; int n_sel; float* Wt; int64_t ldt;
    char* ws; int64_t off_b3, off_zero, off_b3x;
    int blocks_a, blocks_b, blocks_c;
    int ntx, pitch_x;               // B3X: column tiles of the dX product, 1-KB blocks per chunk
};

__device__ __forceinline__ void fl_split8(const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
    SPLIT3_PAIR(v[0], v[1], h, m, l, 0);
    SPLIT3_PAIR(v[2], v[3], h, m, l, 2);
    SPLIT3_PAIR(v[4], v[5], h, m, l, 4);
    SPLIT3_PAIR(v[6], v[7], h, m, l, 6);
}

__global__ __launch_bounds__(FL_THREADS) void fl_prep_kernel(const FlPrepK k) {
    const int tid = threadIdx.x;
    int blk = blockIdx.x;
    if (blk < k.blocks_a) {
        // ---- B3: thread = ONE folded weight (chunk group, column tile, lane, element): its three bf16 terms go out as three
        // 2-byte stores, the 8 threads of a 16-byte fragment piece are adjacent lanes (a thread per piece walked its eight
        // dot products one after the other: 13 us for a parameter-sized launch)
        if (blk == 0 && tid < 16) reinterpret_cast<uint4*>(k.ws + k.off_zero)[tid] = make_uint4(0u, 0u, 0u, 0u);
        const int idx = blk * FL_THREADS + tid;
        const int n_cg = 2 * (k.ncr + k.nco);
        if (idx >= n_cg * k.NT * 512) return;
        const int e = idx & 7, lane = (idx >> 3) & 63, t = (idx >> 9) % k.NT, cg = idx / (512 * k.NT);
        const int j = lane & 31, s = lane >> 5, n = 32 * t + j;
        const int chunk = cg >> 1, gq = cg & 1;
        float v = 0.f;
        if (n < k.N) {
            const float* __restrict__ wrow = k.W + static_cast<int64_t>(n) * k.ldw;
            if (chunk < k.ncr) {
                const int g = 2 * chunk + gq;
                if (g < k.NR) {
                    const FlPrepPiece pc = k.piece[2 * g + s];
                    if (pc.kind != SWR_FL_ZERO && e < pc.n_valid) v = wrow[pc.w_col + e];
                }
            } else {
                const int q = 2 * (chunk - k.ncr) + gq;
                const int o = 16 * q + 8 * s + e;
                const int tb = q < k.NOg ? k.oh_table[o] : -1;
                if (tb >= 0) {
                    // P[n, o] = sum_e emb_t[v, e] W[n, col_t + e]: the arithmetic of fold_fwd_kernel (embed_fwd.hip)
                    const swr_onehot_table& T = k.tab[tb];
                    const float* __restrict__ em = T.grad + static_cast<int64_t>(o - T.oh_off) * T.dim;
                    const float* __restrict__ w = wrow + T.w_col;
                    if ((T.dim & 3) == 0 && (T.w_col & 3) == 0 && (k.ldw & 3) == 0 &&
                        (reinterpret_cast<uintptr_t>(k.W) & 15u) == 0 && (reinterpret_cast<uintptr_t>(T.grad) & 15u) == 0) {
                        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                        for (int c = 0; c < T.dim; c += 4) {
                            const float4 ev = *reinterpret_cast<const float4*>(em + c), wv = *reinterpret_cast<const float4*>(w + c);
                            a0 = fmaf(ev.x, wv.x, a0); a1 = fmaf(ev.y, wv.y, a1); a2 = fmaf(ev.z, wv.z, a2); a3 = fmaf(ev.w, wv.w, a3);
                        }
                        v = (a0 + a1) + (a2 + a3);
                    } else {
                        float acc = 0.f;
                        for (int c = 0; c < T.dim; ++c) acc = fmaf(em[c], w[c], acc);
                        v = acc;
                    }
                }
            }
        }
        const Bf3 sp3 = split3(v);
        __bf16* d = reinterpret_cast<__bf16*>(k.ws + k.off_b3 + static_cast<int64_t>(chunk) * k.pitch_blocks * 1024) +
                    (((gq * k.NT + t) * 3) * 64 + lane) * 8 + e;
        d[0] = sp3.h;
        d[512] = sp3.m;
        d[1024] = sp3.l;
        return;
    }
    blk -= k.blocks_a;
    if (blk < k.blocks_b) {
        // ---- table shadows: thread = (row, 8-column piece) of one small table
        const int it = blk * FL_THREADS + tid;
        if (it >= k.items_planes) return;
        int t = 0;
        while (t + 1 < k.n_tbl && k.tbl[t + 1].first_item <= it) ++t;
        const FlTable& T = k.tbl[t];
        const int local = it - T.first_item, pieces = T.dim >> 3;
        const int row = local / pieces, pc = local - row * pieces;
        const float4* src = reinterpret_cast<const float4*>(T.w + static_cast<int64_t>(row) * T.dim + 8 * pc);
        const float4 x0 = src[0], x1 = src[1];
        const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        bf16x8 h, m, l;
        fl_split8(v, h, m, l);
        bf16x8* d = reinterpret_cast<bf16x8*>(k.ws + T.planes_off + static_cast<int64_t>(local) * FL_PIECE_BYTES);
        d[0] = h;
        d[1] = m;
        d[2] = l;
        return;
    }
    blk -= k.blocks_b;
    if (blk < k.blocks_c) {
        // ---- Wt_sel[r, n] = W[n, sel[r]] (thread = one element, consecutive lanes along n)
        const int64_t i = static_cast<int64_t>(blk) * FL_THREADS + tid;
        if (i < static_cast<int64_t>(k.n_sel) * k.N) {
            const int r = static_cast<int>(i / k.N), n = static_cast<int>(i - static_cast<int64_t>(r) * k.N);
            k.Wt[r * k.ldt + n] = k.W[n * k.ldw + k.sel[r]];
        }
        return;
    }
    blk -= k.blocks_c;
    // ---- B3X: the same rows of W^T as the B operand of the dX product (swr_bn_bwd_dx): thread = one weight; k = the layer's
    // output n (groups of 16), column = selected input column c (tiles of 32)
    {
        const int idx = blk * FL_THREADS + tid;
        const int n_g = (k.N + 15) / 16, n_cg = 2 * ((n_g + 1) / 2);
        if (idx >= n_cg * k.ntx * 512) return;
        const int e = idx & 7, lane = (idx >> 3) & 63, t = (idx >> 9) % k.ntx, cg = idx / (512 * k.ntx);
        const int j = lane & 31, s = lane >> 5, c = 32 * t + j, n = 16 * cg + 8 * s + e;
        const float v = (c < k.n_sel && n < k.N) ? k.W[static_cast<int64_t>(n) * k.ldw + k.sel[c]] : 0.f;
        const Bf3 sp3 = split3(v);
        __bf16* d = reinterpret_cast<__bf16*>(k.ws + k.off_b3x + static_cast<int64_t>(cg >> 1) * k.pitch_x * 1024) +
                    ((((cg & 1) * k.ntx + t) * 3) * 64 + lane) * 8 + e;
        d[0] = sp3.h;
        d[512] = sp3.m;
        d[1024] = sp3.l;
    }
}

extern "C" int swr_fl_prep(const swr_fl_plan* plan, const float* W, int64_t ldw, int K, const int32_t* oh_table,
                           const swr_onehot_table* tables, int n_tables, const int64_t* sel, int n_sel, float* Wt_sel, int64_t ldt,
                           void* workspace, void* stream) {
    FlHost h;
    int rc = fl_build(plan, h);
    if (rc != SWR_OK) return rc;
    SWR_REQUIRE(W && workspace && ldw >= K && K > 0 && n_tables >= 0 && n_tables <= FL_MAX_OHT && plan->N >= 1, SWR_ERR_ARG);
    SWR_REQUIRE(plan->oh_width == 0 || (oh_table && tables && n_tables > 0), SWR_ERR_ARG);
    SWR_REQUIRE(n_sel >= 0 && (n_sel == 0 || (sel && Wt_sel && ldt >= plan->N)), SWR_ERR_ARG);
    FlPrepK k;
    for (int q = 0; q < 2 * h.NR; ++q) {
        const swr_fl_piece& pc = plan->piece[q];
        SWR_REQUIRE(pc.kind == SWR_FL_ZERO || pc.w_col + (pc.kind == SWR_FL_DENSE ? pc.n_valid : 8) <= K, SWR_ERR_ARG);
        k.piece[q].kind = static_cast<int16_t>(pc.kind);
        k.piece[q].n_valid = static_cast<int16_t>(pc.kind == SWR_FL_DENSE ? pc.n_valid : 8);
        k.piece[q].w_col = pc.w_col;
    }
    for (int t = 0; t < n_tables; ++t) {
        SWR_REQUIRE(tables[t].grad && tables[t].vocab > 0 && tables[t].dim > 0 && tables[t].oh_off >= 0 &&
                        tables[t].oh_off + tables[t].vocab <= plan->oh_width && tables[t].w_col >= 0 &&
                        tables[t].w_col + tables[t].dim <= K, SWR_ERR_ARG);
        k.tab[t] = tables[t];
    }
    for (int t = 0; t < h.n_tbl; ++t) k.tbl[t] = h.tbl[t];
    k.n_tbl = h.n_tbl; k.items_planes = h.items_planes;
    k.NR = h.NR; k.NOg = h.NOg; k.ncr = h.ncr; k.nco = h.nco; k.NT = h.NT; k.N = plan->N;
    k.pitch_blocks = fl_pitch_blocks(h.NT);
    k.W = W; k.ldw = ldw; k.oh_table = oh_table;
    k.sel = sel; k.n_sel = n_sel; k.Wt = Wt_sel; k.ldt = ldt;
    k.ws = static_cast<char*>(workspace); k.off_b3 = h.o.b3; k.off_zero = h.o.zero;
    k.blocks_a = static_cast<int>(swr_ceil_div(static_cast<int64_t>(2 * (h.ncr + h.nco)) * h.NT * 512, FL_THREADS));
    k.blocks_b = static_cast<int>(swr_ceil_div(h.items_planes, FL_THREADS));
    k.blocks_c = static_cast<int>(swr_ceil_div(static_cast<int64_t>(n_sel) * plan->N, FL_THREADS));
    // the dX product's image only when it fits that kernel (<= 160 selected columns)
    k.off_b3x = h.o.b3x;
    k.ntx = (n_sel > 0 && n_sel <= 32 * FL_NT_MAX) ? (n_sel + 31) / 32 : 0;
    k.pitch_x = fl_pitch_blocks(k.ntx > 0 ? k.ntx : 1);
    const int n_gx = (plan->N + 15) / 16;
    const int blocks_d = k.ntx > 0 ? static_cast<int>(swr_ceil_div(static_cast<int64_t>(2 * ((n_gx + 1) / 2)) * k.ntx * 512, FL_THREADS)) : 0;
    hipLaunchKernelGGL(fl_prep_kernel, dim3(static_cast<unsigned>(k.blocks_a + k.blocks_b + k.blocks_c + blocks_d)), dim3(FL_THREADS), 0,
                       static_cast<hipStream_t>(stream), k);
    return swr_launch_status();
}

// ------------------------------------------------------------------------------------------------ keys
#define FLK_TILE 64
struct FlKeysK {
    swr_sparse_slot sparse[FL_MAX_SPARSE];
    swr_dense_slot dense[FL_MAX_DENSE];
    FlDevPiece piece[2 * FL_MAX_GROUPS];
    int16_t oh_off[FL_MAX_SPARSE];
    int32_t fpiece[2 * FL_MAX_GROUPS];      // (32-bit: read with scalar loads)
    int n_sparse, n_dense, n_keys, NR, ohw, nfp, nd4, n_tiles;
    int64_t B;
    char* ws;
    int64_t off_keys, off_mask, off_mask_t, off_voff, off_a3f, off_densef;
    uint32_t off_zero32, off_a3f32;
    uint32_t* err;
};

// Everything a thread loads from HBM is issued before anything is consumed: the launch is a chain of dependent round trips
// (ids -> rows -> table rows) on a few MB, i.e. latency, and a thread that walks its slots one load at a time (the first
// form: 22 us) pays a round trip per slot.
#define FLK_IDS 16        // ids per thread: slots (tid >> 6) + 4 j of sample tid & 63
#define FLK_FP 4          // fp32-sourced pieces per thread (64 samples x at most 16 pieces)
template <int DDT>
__device__ __forceinline__ float fl_dense(const swr_dense_slot& d, int64_t i) {
    return DDT == SWR_F32 ? static_cast<const float*>(d.values)[i] : swr_load_value(d.values, d.dtype, i);
}
// IDT: the common dtype of all id columns (SWR_I64 / SWR_I32), or 0: per-slot dispatch; DDT: SWR_F32 when every dense
// feature is fp32 (a per-load dtype switch makes every load wait where it stands)
// SG: 64-sample groups per workgroup (4 waves each).  The descriptor copy -- a memory round trip before any work starts -- and
// the launch's fixed costs are paid once per workgroup: four groups per workgroup at large batches.
template <int IDT, int DDT, int SG>
__global__ __launch_bounds__(FL_THREADS * SG) void fl_keys_kernel(const FlKeysK a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_row_all[];        // [SG][n_sparse][FLK_TILE]
    // the descriptor arrays, copied out of the kernel argument once (one coalesced round of vector loads): indexed in place
    // with anything but a compile-time constant, hipcc fetched every field with a vector load from the kernarg segment and
    // waited for it on the spot -- a round trip per field per slot (22 us for this launch)
    __shared__ swr_sparse_slot c_sp[FL_MAX_SPARSE];
    __shared__ swr_dense_slot c_dn[FL_MAX_DENSE];
    __shared__ FlDevPiece c_pc[2 * FL_MAX_GROUPS];
    __shared__ int32_t c_fp[2 * FL_MAX_GROUPS];
    __shared__ int16_t c_oh[FL_MAX_SPARSE];
    const int tid_all = threadIdx.x, lane = tid_all & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid_all >> 6);
    const int grp = wave_all >> 2, wave = wave_all & 3, tid = tid_all & (FL_THREADS - 1);     // the group's own 4 waves / 256 threads
    uint32_t (*s_row)[FLK_TILE] = reinterpret_cast<uint32_t (*)[FLK_TILE]>(s_row_all) + grp * a.n_sparse;
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(a.sparse);
        uint32_t* dst = reinterpret_cast<uint32_t*>(c_sp);
        for (int w = tid_all; w < a.n_sparse * static_cast<int>(sizeof(swr_sparse_slot) / 4); w += FL_THREADS * SG) dst[w] = src[w];
        src = reinterpret_cast<const uint32_t*>(a.dense);
        dst = reinterpret_cast<uint32_t*>(c_dn);
        for (int w = tid_all; w < a.n_dense * static_cast<int>(sizeof(swr_dense_slot) / 4); w += FL_THREADS * SG) dst[w] = src[w];
        src = reinterpret_cast<const uint32_t*>(a.piece);
        dst = reinterpret_cast<uint32_t*>(c_pc);
        for (int w = tid_all; w < 2 * a.NR * static_cast<int>(sizeof(FlDevPiece) / 4); w += FL_THREADS * SG) dst[w] = src[w];
        if (tid_all < 2 * FL_MAX_GROUPS) c_fp[tid_all] = a.fpiece[tid_all];
        if (tid_all < FL_MAX_SPARSE) c_oh[tid_all] = a.oh_off[tid_all];
    }
    __syncthreads();
    const int64_t b0 = (static_cast<int64_t>(blockIdx.x) * SG + grp) * FLK_TILE;
    // (a group past the end of the batch keeps running -- the barriers are the workgroup's -- on the last sample, stores nothing)
    const int rows = static_cast<int>(max<int64_t>(0, min<int64_t>(FLK_TILE, a.B - b0)));
    uint32_t* __restrict__ keys = reinterpret_cast<uint32_t*>(a.ws + a.off_keys);
    const int64_t bl = min<int64_t>(b0 + min(lane, max(rows, 1) - 1), a.B - 1);                 // (lanes past the end re-read the last sample; nothing is stored)

    // 1. ids -> rows (layers.py:70 `.long()` lookup index; optional hash stage; out-of-range -> row 0 + sticky flag)
    int64_t id[FLK_IDS];
#pragma unroll
    for (int j = 0; j < FLK_IDS; ++j) {
        const int s = wave + 4 * j;
        id[j] = 0;
        if (s < a.n_sparse) {
            if (IDT == SWR_I64) id[j] = static_cast<const int64_t*>(c_sp[s].idx)[bl];
            else if (IDT == SWR_I32) id[j] = static_cast<const int32_t*>(c_sp[s].idx)[bl];
            else id[j] = swr_load_index(c_sp[s].idx, c_sp[s].idx_dtype, bl);
        }
    }
    // (two passes: every id is turned into a row before the first store -- with the stores interleaved, hipcc waited for
    // each store to retire before it would look at the next id)
    bool oor = false;
    uint32_t rowj[FLK_IDS];
#pragma unroll
    for (int j = 0; j < FLK_IDS; ++j) {
        const int s = wave + 4 * j;
        rowj[j] = 0u;
        if (s < a.n_sparse) {
            const int64_t v = id[j];
            const bool bad = (v < 0 || v >= c_sp[s].vocab) && c_sp[s].hash_seed == 0u;
            oor = oor || bad;
            rowj[j] = (lane < rows && !bad) ? static_cast<uint32_t>(v) : 0u;
        }
    }
#pragma unroll
    for (int j = 0; j < FLK_IDS; ++j) {
        const int s = wave + 4 * j;
        if (s < a.n_sparse && c_sp[s].hash_seed == 0u) {
            if (s < a.n_keys && lane < rows) keys[static_cast<int64_t>(s) * a.B + b0 + lane] = rowj[j];
            s_row[s][lane] = rowj[j];
        }
    }
    // hashed slots (none in the reference's configurations): a rolled loop -- the 64-bit modulo is ~100 instructions, and
    // sixteen inlined copies of it evicted the rest of the kernel from the instruction cache
#pragma unroll 1
    for (int s = wave; s < a.n_sparse; s += FL_THREADS / 64) {
        const swr_sparse_slot& sl = c_sp[s];
        if (sl.hash_seed == 0u) continue;
        const int64_t raw = swr_load_index(sl.idx, sl.idx_dtype, bl);
        const uint32_t row = lane < rows ? static_cast<uint32_t>(fl_mix64(static_cast<uint64_t>(raw) ^ sl.hash_seed) % static_cast<uint64_t>(sl.vocab)) : 0u;
        if (s < a.n_keys && lane < rows) keys[static_cast<int64_t>(s) * a.B + b0 + lane] = row;
        s_row[s][lane] = row;
    }
    if (oor && lane < rows && a.err) atomicOr(a.err, SWR_FLAG_INDEX_OOR);
    __syncthreads();

    // 2. the fp32-sourced pieces and the dense block: every load now, the splits and stores at the end
    float fv[FLK_FP][8];
#pragma unroll
    for (int u = 0; u < FLK_FP; ++u) {
        const int fp = wave + 4 * u, r = min(lane, max(rows, 1) - 1);      // item tid + 256 u: piece (wave-uniform), sample
#pragma unroll
        for (int e = 0; e < 8; ++e) fv[u][e] = 0.f;
        if (fp < a.nfp) {
            const FlDevPiece& pc = c_pc[c_fp[fp]];
            if (pc.kind == SWR_FL_ROWS) {
                const swr_sparse_slot& sl = c_sp[pc.slot];
                const float4* src = reinterpret_cast<const float4*>(sl.weight + static_cast<int64_t>(s_row[pc.slot][r]) * sl.dim + pc.off);
                const float4 x0 = src[0], x1 = src[1];
                fv[u][0] = x0.x; fv[u][1] = x0.y; fv[u][2] = x0.z; fv[u][3] = x0.w;
                fv[u][4] = x1.x; fv[u][5] = x1.y; fv[u][6] = x1.z; fv[u][7] = x1.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (e < pc.n_valid) fv[u][e] = fl_dense<DDT>(c_dn[pc.slot + e], min<int64_t>(b0 + r, a.B - 1));
            }
        }
    }
    // x[name].float() of the dense features (layers.py:88-89) as an fp32 block [B][nd4]: the weight-gradient product's operand
    float dv[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int it = tid + u * FL_THREADS;
        if (it < rows * a.nd4) {
            const int r = it / a.nd4, c = it - r * a.nd4;
            if (c < a.n_dense) dv[u] = fl_dense<DDT>(c_dn[c], b0 + r);
        }
    }

    // 3. one-hot block as bits: bit (oh_off_s + row_s) per small table
    if (a.ohw > 0 && tid < rows) {
        uint32_t m[4] = {0u, 0u, 0u, 0u};
        for (int s = 0; s < a.n_sparse; ++s) {
            const int off = c_oh[s];
            if (off >= 0) {
                const uint32_t bit = static_cast<uint32_t>(off) + s_row[s][tid];
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    if ((bit >> 5) == static_cast<uint32_t>(w)) m[w] |= 1u << (bit & 31u);
            }
        }
        reinterpret_cast<uint4*>(a.ws + a.off_mask)[b0 + tid] = make_uint4(m[0], m[1], m[2], m[3]);
        uint32_t* mt = reinterpret_cast<uint32_t*>(a.ws + a.off_mask_t);
#pragma unroll
        for (int w = 0; w < 4; ++w) mt[static_cast<int64_t>(w) * a.B + b0 + tid] = m[w];
    }

    // 4. the byte offset of every lane's piece: [tile][group][lane (s * 32 + i)]
    uint32_t* __restrict__ voff = reinterpret_cast<uint32_t*>(a.ws + a.off_voff);
    for (int wg = wave; wg < 2 * a.NR; wg += FL_THREADS / 64) {       // (tile of the block, group): wave-uniform
        const int tt = wg >= a.NR ? 1 : 0, g = wg - tt * a.NR;
        const int i = lane & 31, s = lane >> 5;
        const int64_t T = b0 / 32 + tt;
        if (T >= a.n_tiles) continue;
        const int r = 32 * tt + i;
        const FlDevPiece p0 = c_pc[2 * g], p1 = c_pc[2 * g + 1];
        const int kind = s ? p1.kind : p0.kind, slot = s ? p1.slot : p0.slot, fp = s ? p1.fp : p0.fp;
        const uint32_t base = s ? p1.base : p0.base, rowbytes = s ? p1.rowbytes : p0.rowbytes;
        uint32_t v = a.off_zero32;
        if (r < rows) {
            if (kind == SWR_FL_PLANES) v = base + s_row[slot][r] * rowbytes;
            else if (kind != SWR_FL_ZERO) v = a.off_a3f32 + static_cast<uint32_t>((T * a.nfp + fp) * 32 + i) * FL_PIECE_BYTES;
        }
        voff[(T * a.NR + g) * 64 + lane] = v;
    }

    // 5. split the gathered pieces into the three bf16 terms, 48 bytes per (sample, piece); store the dense block
#pragma unroll
    for (int u = 0; u < FLK_FP; ++u) {
        const int fp = wave + 4 * u, r = lane;
        if (fp < a.nfp && r < rows) {
            bf16x8 h, m, l;
            fl_split8(fv[u], h, m, l);
            const int64_t T = (b0 + r) / 32;
            bf16x8* d = reinterpret_cast<bf16x8*>(a.ws + a.off_a3f + ((T * a.nfp + fp) * 32 + (r & 31)) * FL_PIECE_BYTES);
            d[0] = h;
            d[1] = m;
            d[2] = l;
        }
    }
    float* __restrict__ df = reinterpret_cast<float*>(a.ws + a.off_densef);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int it = tid + u * FL_THREADS;
        if (it < rows * a.nd4) df[b0 * a.nd4 + it] = dv[u];
    }
    // (more than 16 fp32-sourced pieces or more than 8 dense features: the rest, item by item)
    for (int it = tid + FLK_FP * FL_THREADS; it < FLK_TILE * a.nfp; it += FL_THREADS) {
        const int fp = it >> 6, r = it & 63;
        if (r >= rows) continue;
        const FlDevPiece& pc = c_pc[c_fp[fp]];
        float v[8];
        if (pc.kind == SWR_FL_ROWS) {
            const swr_sparse_slot& sl = c_sp[pc.slot];
            const float4* src = reinterpret_cast<const float4*>(sl.weight + static_cast<int64_t>(s_row[pc.slot][r]) * sl.dim + pc.off);
            const float4 x0 = src[0], x1 = src[1];
            v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                v[e] = e < pc.n_valid ? swr_load_value(c_dn[pc.slot + e].values, c_dn[pc.slot + e].dtype, b0 + r) : 0.f;
        }
        bf16x8 h, m, l;
        fl_split8(v, h, m, l);
        const int64_t T = (b0 + r) / 32;
        bf16x8* d = reinterpret_cast<bf16x8*>(a.ws + a.off_a3f + ((T * a.nfp + fp) * 32 + (r & 31)) * FL_PIECE_BYTES);
        d[0] = h;
        d[1] = m;
        d[2] = l;
    }
    for (int it = tid + 2 * FL_THREADS; it < rows * a.nd4; it += FL_THREADS) {
        const int r = it / a.nd4, c = it - r * a.nd4;
        df[b0 * a.nd4 + it] = c < a.n_dense ? swr_load_value(c_dn[c].values, c_dn[c].dtype, b0 + r) : 0.f;
    }
}

extern "C" int swr_fl_keys(const swr_fl_plan* plan, void* workspace, uint32_t* err_flag, void* stream) {
    FlHost h;
    int rc = fl_build(plan, h);
    if (rc != SWR_OK) return rc;
    SWR_REQUIRE(workspace != nullptr, SWR_ERR_ARG);
    if (plan->B == 0) return SWR_OK;
    FlKeysK a;
    for (int s = 0; s < plan->n_sparse; ++s) {
        a.sparse[s] = plan->sparse_host[s];
        a.oh_off[s] = static_cast<int16_t>(plan->oh_width > 0 ? plan->oh_off_host[s] : -1);
    }
    for (int s = 0; s < plan->n_dense; ++s) a.dense[s] = plan->dense_host[s];
    for (int q = 0; q < 2 * h.NR; ++q) a.piece[q] = h.piece[q];
    for (int f = 0; f < h.o.n_fpieces; ++f) a.fpiece[f] = h.fpiece[f];
    a.n_sparse = plan->n_sparse; a.n_dense = plan->n_dense; a.n_keys = plan->n_keys; a.NR = h.NR; a.ohw = plan->oh_width;
    a.nfp = h.o.n_fpieces; a.nd4 = h.o.nd4; a.n_tiles = h.n_tiles; a.B = plan->B;
    a.ws = static_cast<char*>(workspace);
    a.off_keys = h.o.keys; a.off_mask = h.o.mask; a.off_mask_t = h.o.mask_t; a.off_voff = h.o.voff; a.off_a3f = h.o.a3f;
    a.off_densef = h.o.densef;
    a.off_zero32 = static_cast<uint32_t>(h.o.zero); a.off_a3f32 = static_cast<uint32_t>(h.o.a3f);
    a.err = err_flag;
    int common = plan->sparse_host[0].idx_dtype;
    for (int s = 1; s < plan->n_sparse; ++s)
        if (plan->sparse_host[s].idx_dtype != common) common = 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    bool f32 = true;
    for (int s = 0; s < plan->n_dense; ++s) f32 = f32 && plan->dense_host[s].dtype == SWR_F32;
    int sg = plan->B >= 32768 ? 4 : 1;
    {
        static int force = -1;                      // SWR_FLK_SG=1|2|4: sample groups per workgroup (experiments)
        if (force < 0) { const char* e = getenv("SWR_FLK_SG"); force = e ? atoi(e) : 0; }
        if (force == 1 || force == 2 || force == 4) sg = force;
    }
    const dim3 grid(static_cast<unsigned>(swr_ceil_div(plan->B, FLK_TILE * sg)));
    const unsigned lds = static_cast<unsigned>(sg) * plan->n_sparse * FLK_TILE * sizeof(uint32_t);
#define FLK_GO(SGV)                                                                                                              \
    do {                                                                                                                         \
        if (common == SWR_I64 && f32) hipLaunchKernelGGL((fl_keys_kernel<SWR_I64, SWR_F32, SGV>), grid, dim3(FL_THREADS * SGV), lds, st, a);   \
        else if (common == SWR_I32 && f32) hipLaunchKernelGGL((fl_keys_kernel<SWR_I32, SWR_F32, SGV>), grid, dim3(FL_THREADS * SGV), lds, st, a); \
        else hipLaunchKernelGGL((fl_keys_kernel<0, 0, SGV>), grid, dim3(FL_THREADS * SGV), lds, st, a);                           \
    } while (0)
    if (sg == 4) FLK_GO(4);
    else if (sg == 2) FLK_GO(2);
    else FLK_GO(1);
#undef FLK_GO
    return swr_launch_status();
}

// ------------------------------------------------------------------------------------------------ forward product
// Short batches: when a wave per 32-row tile leaves most of the chip idle (fewer than 2 workgroups per CU ... 256 CUs x 4 waves),
// the products spread their column tiles over the grid's y dimension (NA = 1 forms).  SWR_FL_SPLIT=0 / 1 forces the choice
// (read per call: the tests compare both forms in one process).
static bool fl_split_columns(int64_t n_tiles, int nt) {
    if (nt < 2) return false;
    const char* e = getenv("SWR_FL_SPLIT");
    if (e && (e[0] == '0' || e[0] == '1')) return e[0] == '1';
    return n_tiles * nt <= 2560;          // measured at N = 148 (5 tiles): 8 192 rows 0.2259 -> 0.2110 ms per step, 16 384 rows 0.2678 ->
                                          // 0.2637, 32 768 rows 0.2634 -> 0.279, 65 536 rows 0.377 -> 0.417: up to 16 384 rows x 5 tiles
}

typedef __attribute__((address_space(3))) void* fl_lds_ptr;
typedef __attribute__((address_space(1))) const void* fl_glb_ptr;

// a lane's three 16-byte terms: 32-bit byte offset (VGPR) + the workspace base (SGPR pair)
#define FL_ALOAD(dst, voff, sbase, OFF) \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #OFF : "=v"(dst) : "v"(voff), "s"(sbase) : "memory")

typedef unsigned fl_u32x4 __attribute__((ext_vector_type(4)));

struct FlFwdK {
    swr_gemm_args e;              // epilogue fields: M, N, C, ldc, bias, stat_partials (groups = 1)
    const char* ws;
    const uint4* b3;
    const uint32_t* voff;
    const uint4* mask;
    int NR, NOg, ncr, nco, n_tiles;
};

// NA = column tiles a wave accumulates.  NA == NT: the wave owns all N columns (the long-batch form).  NA == 1 (SHORT batches): the
// grid's y dimension walks the NT column tiles -- NT times the workgroups, each fetching only its tile's weight pieces.  At the 8 192
// rows of a strong-scaling shard the long-batch form is 64 workgroups of four 540-MFMA chains on a 256-CU chip.  Every output element
// sees the same products in the same order: identical bits.
template <int NT, int NA = NT>
__global__ __launch_bounds__(FL_THREADS, 2) void fl_fwd_kernel(const FlFwdK k) {
    static_assert(NA == NT || NA == 1, "a wave takes all column tiles or one");
    constexpr bool SPLIT = NA != NT;
    constexpr int PITCH_U4 = ((6 * NT + 3) / 4 * 4) * 64;       // uint4 per chunk in HBM (the long-batch form's LDS buffer size too)
    constexpr int LPITCH_U4 = SPLIT ? 8 * 64 : PITCH_U4;        // uint4 per LDS buffer: split form = 2 groups x 3 terms (+ 2 unused) pieces
    constexpr int GSTRIDE = SPLIT ? 3 * 64 : NT * 3 * 64;       // uint4 from a chunk's first group to its second, in LDS
    constexpr int PPW = PITCH_U4 / 64 / 4;                      // 1-KB DMA pieces per wave per chunk (long-batch form)
    const int t0 = SPLIT ? static_cast<int>(blockIdx.y) : 0;    // first column tile of this workgroup
    extern __shared__ __attribute__((aligned(16))) uint4 lds[]; // [2][LPITCH_U4]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const int i = lane & 31, s = lane >> 5;
    const int64_t tile = static_cast<int64_t>(blockIdx.x) * 4 + wave;
    const bool live = tile < k.n_tiles;                         // (wave-uniform) a wave past the end computes the last tile again, stores nothing
    const int64_t T = live ? tile : k.n_tiles - 1;
    const int NR = k.NR, ncr = k.ncr, nct = k.ncr + k.nco;
    const char* __restrict__ wsb = k.ws;

    f32x16 acc[NA];
#pragma unroll
    for (int t = 0; t < NA; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // piece offsets of all real groups (groups past NR repeat the last one: never used), the sample's one-hot bits
    uint32_t vo[FL_MAX_GROUPS];
    const uint32_t* __restrict__ vp = k.voff + T * NR * 64 + lane;
#pragma unroll
    for (int g = 0; g < FL_MAX_GROUPS; ++g) vo[g] = vp[min(g, NR - 1) * 64];
    fl_u32x4 mk = {0u, 0u, 0u, 0u};
    if (k.nco > 0) mk = reinterpret_cast<const fl_u32x4*>(k.mask)[min<int64_t>(T * 32 + i, k.e.M - 1)];

    fl_u32x4 ar[4][3];     // (native vectors: a HIP uint4 is a struct, which an asm "+v" operand cannot be)
    auto a_issue = [&](uint32_t off, fl_u32x4 (&dst)[3]) {
#ifdef FLF_NO_A           // (ablation switches FLF_*: measurement aids of tools/micro/ab_flf.sh, all off in the product build)
        off = 0;
#endif
        FL_ALOAD(dst[0], off, wsb, 0);
        FL_ALOAD(dst[1], off, wsb, 16);
        FL_ALOAD(dst[2], off, wsb, 32);
    };
    // weights of chunk c -> LDS buffer: PITCH_U4 / 64 DMA pieces of 1 KB, PPW per wave
    auto dma_chunk = [&](int c, int buf) {
        if constexpr (SPLIT) {
            // the six pieces (group, term) of column tile t0: wave w copies pieces w and w + 4
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = wave + 4 * u;                          // (wave-uniform)
                if (q < 6) {
                    const int gq = q / 3, term = q - 3 * gq;
                    const uint4* src = k.b3 + static_cast<size_t>(c) * PITCH_U4 + gq * (NT * 3 * 64) + (t0 * 3 + term) * 64 + lane;
                    uint4* dst = lds + buf * LPITCH_U4 + q * 64;
                    __builtin_amdgcn_global_load_lds((fl_glb_ptr)src, (fl_lds_ptr)dst, 16, 0, 0);
                }
            }
        } else {
        const uint4* src = k.b3 + static_cast<size_t>(c) * PITCH_U4 + wave * (PPW * 64) + lane;
        uint4* dst = lds + buf * PITCH_U4 + wave * (PPW * 64);
#ifndef FLF_NO_DMA
#pragma unroll
        for (int u = 0; u < PPW; ++u)
            __builtin_amdgcn_global_load_lds((fl_glb_ptr)(src + u * 64), (fl_lds_ptr)(dst + u * 64), 16, 0, 0);
#endif
        }
    };
    // six (real group) or three (exact A: one-hot group) products per column tile, small terms first
    auto mma_group = [&](const uint4* bp, bf16x8 ah, bf16x8 am, bf16x8 al, auto exact_c) {
        constexpr bool EX = decltype(exact_c)::value;
#ifdef FLF_NO_MMA
        acc[0][0] += __builtin_bit_cast(float, __builtin_bit_cast(fl_u32x4, ah)[0] ^ __builtin_bit_cast(fl_u32x4, al)[1] ^ bp[0].x);
        return;
#endif
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, bp[(t * 3 + 0) * 64]);
            const bf16x8 b1 = __builtin_bit_cast(bf16x8, bp[(t * 3 + 1) * 64]);
            const bf16x8 b2 = __builtin_bit_cast(bf16x8, bp[(t * 3 + 2) * 64]);
            f32x16 c_ = acc[t];
            if ((!EX) && !SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b0, c_, 0, 0, 0);
            if (!SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b2, c_, 0, 0, 0);
            if ((!EX) && !SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b1, c_, 0, 0, 0);
            if (!EX) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b0, c_, 0, 0, 0);
            c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1, c_, 0, 0, 0);
            c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0, c_, 0, 0, 0);
            acc[t] = c_;
        }
    };

    a_issue(vo[0], ar[0]);
    a_issue(vo[1], ar[1]);
    if (NR > 2) a_issue(vo[2], ar[2]);
    if (NR > 3) a_issue(vo[3], ar[3]);
    dma_chunk(0, 0);
    // everything has landed; the ring registers become usable HERE (the asm ties them: nothing that reads them moves above).
    // The offsets and the mask are tied too: hipcc does not count the asm loads, so a wait it inserted in front of their
    // first use further down would be a vmcnt(0) in the middle of the pipeline
    {
        // (operands are locals: an asm operand reached through the lambdas' by-reference captures is "indirect" to hipcc)
        fl_u32x4 t0 = ar[0][0], t1 = ar[0][1], t2 = ar[0][2], t3 = ar[1][0], t4 = ar[1][1], t5 = ar[1][2];
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "+v"(t4), "+v"(t5),
                       "+v"(vo[0]), "+v"(vo[4]), "+v"(vo[5]), "+v"(vo[6]), "+v"(vo[7]), "+v"(vo[8]), "+v"(vo[9]), "+v"(vo[10]),
                       "+v"(vo[11]), "+v"(vo[12]), "+v"(vo[13]), "+v"(vo[14]), "+v"(vo[15]), "+v"(mk)
                     :: "memory");
        ar[0][0] = t0; ar[0][1] = t1; ar[0][2] = t2; ar[1][0] = t3; ar[1][1] = t4; ar[1][2] = t5;
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // ---- real groups: chunk C = groups 2 C, 2 C + 1 (ring slots 2 (C & 1) + gq); fully unrolled over the maximum, a chunk
    // past the end is skipped by a wave-uniform branch -- every index and every wait count inside is a compile-time constant
    auto real_chunk = [&](auto c_c) {
        constexpr int C = decltype(c_c)::value, BUF = C & 1;
        if (C >= ncr) return;
        if (C + 1 < nct) dma_chunk(C + 1, BUF ^ 1);
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
            constexpr int dummy = 0; (void)dummy;
            const int slot = 2 * BUF + gq;
            const uint4* bp = lds + BUF * LPITCH_U4 + gq * GSTRIDE + lane;
            // (an odd group count leaves the second half of the last chunk empty: its ring slot was never loaded, and
            // whatever bits it holds -- NaN patterns included -- must not meet the zero weights)
            if (2 * C + gq < NR)
                mma_group(bp, __builtin_bit_cast(bf16x8, ar[slot][0]), __builtin_bit_cast(bf16x8, ar[slot][1]),
                          __builtin_bit_cast(bf16x8, ar[slot][2]), std::false_type{});
            __builtin_amdgcn_sched_barrier(0);       // the slot's MFMAs are issued before its registers are re-loaded
            // group 2 C + 4 + gq takes the slot (never a dummy load: the hardware would write the slot's registers after
            // the compiler has given them to something else)
            const int gn = 2 * C + 4 + gq;
            if (gn < FL_MAX_GROUPS && gn < NR) a_issue(vo[gn < FL_MAX_GROUPS ? gn : 0], ar[slot]);
        }
        // all but the loads issued in THIS chunk have landed: the weights of chunk C + 1 and the A pieces of groups
        // 2 C + 2, 2 C + 3 (slots of the other parity).  (wave-uniform choice between three counted waits)
        constexpr int NB = 2 * (BUF ^ 1);
        const int n_issued = __builtin_amdgcn_readfirstlane(min(2, max(0, NR - (2 * C + 4))));      // A groups loaded in this chunk (SGPR)
        fl_u32x4 t0 = ar[NB][0], t1 = ar[NB][1], t2 = ar[NB][2], t3 = ar[NB + 1][0], t4 = ar[NB + 1][1], t5 = ar[NB + 1][2];
        // ONE asm statement (the choice between the three counted waits is made inside it): with one statement per case
        // hipcc resolved the tied operands of the three branches by copying the ring registers IN FRONT of the waits --
        // reading registers whose loads were still in flight
        asm volatile("s_cmp_eq_u32 %6, 2\n\t"
                     "s_cbranch_scc1 1f\n\t"
                     "s_cmp_eq_u32 %6, 1\n\t"
                     "s_cbranch_scc1 2f\n\t"
                     "s_waitcnt vmcnt(0)\n\t"
                     "s_branch 3f\n"
                     "2:\n\t"
                     "s_waitcnt vmcnt(3)\n\t"
                     "s_branch 3f\n"
                     "1:\n\t"
                     "s_waitcnt vmcnt(6)\n"
                     "3:\n\t"
                     : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "+v"(t4), "+v"(t5)
                     : "s"(n_issued)
                     : "memory", "scc");
        ar[NB][0] = t0; ar[NB][1] = t1; ar[NB][2] = t2; ar[NB + 1][0] = t3; ar[NB + 1][1] = t4; ar[NB + 1][2] = t5;
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    real_chunk(std::integral_constant<int, 0>{}); real_chunk(std::integral_constant<int, 1>{});
    real_chunk(std::integral_constant<int, 2>{}); real_chunk(std::integral_constant<int, 3>{});
    real_chunk(std::integral_constant<int, 4>{}); real_chunk(std::integral_constant<int, 5>{});
    real_chunk(std::integral_constant<int, 6>{}); real_chunk(std::integral_constant<int, 7>{});

    // ---- one-hot groups: the A fragment is 8 bits of the sample's mask, expanded to bf16 1.0 / 0.0; exact in bf16, so the
    // middle / low terms of A vanish and three products remain
    auto oh_chunk = [&](auto q_c) {
        constexpr int Q = decltype(q_c)::value;
        if (Q >= k.nco) return;
        const int c = ncr + Q, buf = c & 1;
        if (c + 1 < nct) dma_chunk(c + 1, buf ^ 1);
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
            constexpr int dummy = 0; (void)dummy;
            const int q = 2 * Q + gq;                                        // one-hot group: mask bytes 2 q, 2 q + 1
            const uint32_t word = mk[q >> 1];
            const uint32_t byte = (word >> (16 * (q & 1) + 8 * s)) & 0xFFu;
            fl_u32x4 f;
            f[0] = ((byte >> 0) & 1u) * 0x3F80u | ((byte >> 1) & 1u) * 0x3F800000u;
            f[1] = ((byte >> 2) & 1u) * 0x3F80u | ((byte >> 3) & 1u) * 0x3F800000u;
            f[2] = ((byte >> 4) & 1u) * 0x3F80u | ((byte >> 5) & 1u) * 0x3F800000u;
            f[3] = ((byte >> 6) & 1u) * 0x3F80u | ((byte >> 7) & 1u) * 0x3F800000u;
            const uint4* bp = lds + buf * LPITCH_U4 + gq * GSTRIDE + lane;
            const bf16x8 ah = __builtin_bit_cast(bf16x8, f);
            if (q < k.NOg) mma_group(bp, ah, ah, ah, std::true_type{});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    oh_chunk(std::integral_constant<int, 0>{}); oh_chunk(std::integral_constant<int, 1>{});
    oh_chunk(std::integral_constant<int, 2>{}); oh_chunk(std::integral_constant<int, 3>{});
    // (the ring's last re-loads may still be in flight: their registers are dead)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if (!live) return;
#ifdef FLF_NO_EPI
    {
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NA; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[t][r];
        if (sum == 12345.678f) k.e.C[T * 32 + i] = sum;
        return;
    }
#endif
    rows_epilogue<NA>(k.e, k.n_tiles, acc, 0, T, T * 32, 32 * t0, i, s);
}

extern "C" int swr_fl_fwd(const swr_fl_plan* plan, const void* workspace, const float* bias, float* Z, int64_t ldz,
                          float* stat_partials, void* stream) {
    FlHost h;
    int rc = fl_build(plan, h);
    if (rc != SWR_OK) return rc;
    SWR_REQUIRE(workspace && Z && plan->N >= 1 && ldz >= plan->N, SWR_ERR_ARG);
    if (plan->B == 0) return SWR_OK;
    FlFwdK k;
    std::memset(&k, 0, sizeof(k));
    k.e.M = plan->B; k.e.N = plan->N; k.e.K = 16 * h.NR + plan->oh_width;
    k.e.C = Z; k.e.ldc = ldz; k.e.bias = bias; k.e.stat_partials = stat_partials; k.e.groups = 1;
    const char* ws = static_cast<const char*>(workspace);
    k.ws = ws;
    k.b3 = reinterpret_cast<const uint4*>(ws + h.o.b3);
    k.voff = reinterpret_cast<const uint32_t*>(ws + h.o.voff);
    k.mask = reinterpret_cast<const uint4*>(ws + h.o.mask);
    k.NR = h.NR; k.NOg = h.NOg; k.ncr = h.ncr; k.nco = h.nco; k.n_tiles = h.n_tiles;
    const dim3 grid(static_cast<unsigned>(swr_ceil_div(h.n_tiles, 4)));
    const unsigned lds = static_cast<unsigned>(2 * fl_pitch_blocks(h.NT) * 1024);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (fl_split_columns(h.n_tiles, h.NT)) {
        // short batch: one column tile per workgroup (grid y), 16 KB of LDS
        const dim3 grid2(grid.x, static_cast<unsigned>(h.NT));
        switch (h.NT) {
            case 2: hipLaunchKernelGGL((fl_fwd_kernel<2, 1>), grid2, dim3(FL_THREADS), 16 * 1024, st, k); break;
            case 3: hipLaunchKernelGGL((fl_fwd_kernel<3, 1>), grid2, dim3(FL_THREADS), 16 * 1024, st, k); break;
            case 4: hipLaunchKernelGGL((fl_fwd_kernel<4, 1>), grid2, dim3(FL_THREADS), 16 * 1024, st, k); break;
            default: hipLaunchKernelGGL((fl_fwd_kernel<5, 1>), grid2, dim3(FL_THREADS), 16 * 1024, st, k); break;
        }
        return swr_launch_status();
    }
#define FL_GO(NTV)                                                                                                      \
    do {                                                                                                                \
        if (lds >= 64 * 1024 && !swr_raise_lds(reinterpret_cast<const void*>(fl_fwd_kernel<NTV>), 80 * 1024)) return SWR_ERR_LAUNCH; \
        hipLaunchKernelGGL(fl_fwd_kernel<NTV>, grid, dim3(FL_THREADS), lds, st, k);                                     \
    } while (0)
    switch (h.NT) {
        case 1: FL_GO(1); break;
        case 2: FL_GO(2); break;
        case 3: FL_GO(3); break;
        case 4: FL_GO(4); break;
        default: FL_GO(5); break;
    }
#undef FL_GO
    return swr_launch_status();
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dWp[N, Kp + ohw] = dZ^T A' (+ column sums of dZ) with A' gathered by the staging threads of gemm.hip's gemm_tn_x6g_kernel:
// the product that needed the written block.  Falls to the caller's written-block path when the shape is not the
// bf16-split kernel's (swr_fl_dw_supported).
static int fl_dw_args(const swr_fl_plan* plan, const FlHost& h, const float* dZ, int64_t lddz, float* dWp, int64_t lddwp,
                      float* colsum, swr_gemm_tn_args& a) {
    std::memset(&a, 0, sizeof(a));
    a.M = plan->B; a.K1 = plan->N; a.K2 = 16 * h.NR + plan->oh_width;
    a.A = dZ; a.lda = lddz; a.C = dWp; a.ldc = lddwp; a.colsum = colsum; a.groups = 1;
    return SWR_OK;
}

extern "C" int swr_fl_dw_supported(const swr_fl_plan* plan, int64_t lddz) {
    FlHost h;
    if (fl_build(plan, h) != SWR_OK || plan->N < 1) return 0;
    swr_gemm_tn_args a;
    float* fake = reinterpret_cast<float*>(static_cast<uintptr_t>(256));       // (alignment is checked again at launch)
    fl_dw_args(plan, h, fake, lddz, fake, 16 * h.NR + plan->oh_width, nullptr, a);
    return tn_x6_gather_ok(a) ? 1 : 0;
}

extern "C" size_t swr_fl_dw_workspace_bytes(const swr_fl_plan* plan) {
    FlHost h;
    if (fl_build(plan, h) != SWR_OK || plan->N < 1) return 0;
    swr_gemm_tn_args a;
    float* fake = reinterpret_cast<float*>(static_cast<uintptr_t>(256));       // (only the pointers' alignment is looked at)
    fl_dw_args(plan, h, fake, plan->N + (plan->N & 1), fake, 16 * h.NR + plan->oh_width, fake, a);
    a.B = fake; a.ldb = a.K2;
    return swr_gemm_tn_workspace_bytes(&a);
}

struct FlDwBn { const float* Z; int64_t ldz; const float* ca; const float* cb; const float* cc; const float* mean; };
static int fl_dw_launch(const swr_fl_plan* plan, const void* fl_workspace, const float* dZ, int64_t lddz, const FlDwBn* bn, float* dWp,
                        int64_t lddwp, float* colsum, void* workspace, size_t workspace_bytes, void* stream) {
    FlHost h;
    int rc = fl_build(plan, h);
    if (rc != SWR_OK) return rc;
    SWR_REQUIRE(fl_workspace && dZ && dWp && plan->N >= 1 && lddz >= plan->N && lddwp >= 16 * h.NR + plan->oh_width, SWR_ERR_ARG);
    swr_gemm_tn_args a;
    fl_dw_args(plan, h, dZ, lddz, dWp, lddwp, colsum, a);
    SWR_REQUIRE(tn_x6_gather_ok(a), SWR_ERR_UNSUPPORTED);
    const char* ws = static_cast<const char*>(fl_workspace);
    const uint32_t* keys = reinterpret_cast<const uint32_t*>(ws + h.o.keys);
    const uint32_t* mask_t = reinterpret_cast<const uint32_t*>(ws + h.o.mask_t);
    const float* densef = reinterpret_cast<const float*>(ws + h.o.densef);
    TnGather g;
    std::memset(&g, 0, sizeof(g));
    g.kp = 16 * h.NR;
    g.n_pieces = 2 * h.NR + plan->oh_width / 8;
    SWR_REQUIRE(g.n_pieces <= TNG_MAX_PIECES, SWR_ERR_UNSUPPORTED);
    for (int q = 0; q < 2 * h.NR; ++q) {
        const swr_fl_piece& pc = plan->piece[q];
        TnGatherPiece& P = g.piece[q];
        P.kwp = keys;
        if (pc.kind == SWR_FL_PLANES || pc.kind == SWR_FL_ROWS) {
            SWR_REQUIRE(pc.slot < plan->n_keys, SWR_ERR_ARG);            // its keys were written
            const swr_sparse_slot& sl = plan->sparse_host[pc.slot];
            P.kind = TNG_TABLE;
            P.vbase = sl.weight + pc.off;
            P.vstride = static_cast<uint32_t>(sl.dim);
            P.kmax = static_cast<uint32_t>(sl.vocab - 1);
            P.kwp = keys + static_cast<int64_t>(pc.slot) * plan->B;
        } else if (pc.kind == SWR_FL_DENSE) {
            P.kind = TNG_ROWIDX;
            P.vbase = densef + pc.slot;
            P.vstride = static_cast<uint32_t>(h.o.nd4);
            P.n_valid = static_cast<int16_t>(std::min(8, h.o.nd4 - pc.slot));      // (the block's pad columns hold zeros)
        } else {
            P.kind = TNG_ZERO;
        }
    }
    for (int o8 = 0; o8 < plan->oh_width / 8; ++o8) {
        TnGatherPiece& P = g.piece[2 * h.NR + o8];
        P.kind = TNG_ONEHOT;
        P.kwp = mask_t + static_cast<int64_t>((8 * o8) / 32) * plan->B;
        P.bit0 = static_cast<int16_t>((8 * o8) % 32);
    }
    if (bn) {
        SWR_REQUIRE(bn->Z && bn->ca && bn->cb && bn->cc && bn->mean && tn_x6_gather_wide(a, g.kp), SWR_ERR_UNSUPPORTED);
        g.a_z = bn->Z; g.a_ldz = bn->ldz; g.a_ca = bn->ca; g.a_cb = bn->cb; g.a_cc = bn->cc; g.a_mean = bn->mean;
        g.tr_ws = ws; g.tr_voff = reinterpret_cast<const uint32_t*>(ws + h.o.voff); g.tr_mask_t = mask_t; g.tr_nr = h.NR;
    }
    return tn_x6_gather(a, g, workspace, workspace_bytes, stream);
}

extern "C" int swr_fl_dw(const swr_fl_plan* plan, const void* fl_workspace, const float* dZ, int64_t lddz, float* dWp, int64_t lddwp,
                         float* colsum, void* workspace, size_t workspace_bytes, void* stream) {
    return fl_dw_launch(plan, fl_workspace, dZ, lddz, nullptr, dWp, lddwp, colsum, workspace, workspace_bytes, stream);
}

// the same product with dZ = ca dY + cb (Z - mean) + cc recomputed by the staging threads (the operations of swr_bn_bwd_dx, in
// their order): dZ is never written -- swr_bn_bwd_dx(dZ = NULL) then moves 122 MB instead of 164
extern "C" int swr_fl_dw_bn_supported(const swr_fl_plan* plan, int64_t lddy, int64_t ldz) {
    FlHost h;
    if (!plan || fl_build(plan, h) != SWR_OK || plan->N < 1 || lddy < plan->N || ldz < plan->N || lddy % 2 || ldz % 2) return 0;
    swr_gemm_tn_args a;
    fl_dw_args(plan, h, reinterpret_cast<const float*>(256), lddy, reinterpret_cast<float*>(256), 16 * h.NR + plan->oh_width, nullptr, a);
    return (tn_x6_gather_wide(a, 16 * h.NR) && plan->B * ldz < (1ll << 31)) ? 1 : 0;
}
extern "C" int swr_fl_dw_bn(const swr_fl_plan* plan, const void* fl_workspace, const float* dY, int64_t lddy, const float* Z, int64_t ldz,
                            const float* ca, const float* cb, const float* cc, const float* mean, float* dWp, int64_t lddwp, float* colsum,
                            void* workspace, size_t workspace_bytes, void* stream) {
    const FlDwBn bn = {Z, ldz, ca, cb, cc, mean};
    return fl_dw_launch(plan, fl_workspace, dY, lddy, &bn, dWp, lddwp, colsum, workspace, workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------------ BN backward + dX
// dZ = ca * dY + cb * (Z - mean) + cc   (swr_act_bwd_apply without an activation: the BatchNorm backward of the stacked
// expert / gate layer, mmoe.py:44-49 through layers.py:254-256) and dX = dZ W[:, sel] in ONE pass: a lane computes the 8
// values of its A fragment from dY and Z (four 16-byte loads), writes them out as dZ -- the weight-gradient product reads it
// -- and splits them into the three bf16 terms in registers; the weights arrive as the B3X image of swr_fl_prep.
// Same arithmetic, in the same order, as swr_act_bwd_apply followed by swr_gemm_nt: identical bits.  Saves the dZ read of the
// product, one launch, and the pass's latency: 23 + 31 us -> one HBM-bound pass over dY, Z, dZ, dX.
struct FlDxK {
    const char* dY; const char* Z;        // byte pointers: 32-bit lane offsets
    const float* ca; const float* cb; const float* cc; const float* mean;
    float* dZ; int64_t lddz;
    float* dX; int64_t lddx;
    const uint4* b3x;
    int64_t M;
    int K, n_out, n_groups, n_tiles;
    uint32_t lddy_b, ldz_b;               // row pitches in bytes
};

#define FL_XLOAD(dst, voff, sbase, OFF) \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #OFF : "=v"(dst) : "v"(voff), "s"(sbase) : "memory")

// NA: as in fl_fwd_kernel -- NA == 1 spreads the NT output column tiles over the grid's y dimension (short batches); every
// workgroup column rebuilds the dZ fragment (loads from L2, a few VALU), the first one writes dZ out where it is wanted.
template <int NT, int NA = NT>
__global__ __launch_bounds__(FL_THREADS, 2) void fl_dx_kernel(const FlDxK k) {
    static_assert(NA == NT || NA == 1, "a wave takes all column tiles or one");
    constexpr bool SPLIT = NA != NT;
    constexpr int PITCH_U4 = ((6 * NT + 3) / 4 * 4) * 64;
    constexpr int LPITCH_U4 = SPLIT ? 8 * 64 : PITCH_U4;
    constexpr int GSTRIDE = SPLIT ? 3 * 64 : NT * 3 * 64;
    constexpr int PPW = PITCH_U4 / 64 / 4;
    const int ct0 = SPLIT ? static_cast<int>(blockIdx.y) : 0;            // first output column tile of this workgroup
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];          // [2][LPITCH_U4] weights | 4 x 160 coefficients
    float* coef = reinterpret_cast<float*>(lds + 2 * LPITCH_U4);         // ca | cb | cc | mean, 160 each (zero past K)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const int i = lane & 31, s = lane >> 5;
    const int64_t tile = static_cast<int64_t>(blockIdx.x) * 4 + wave;
    const bool live = tile < k.n_tiles;
    const int64_t T = live ? tile : k.n_tiles - 1;
    const int K = k.K, ng = k.n_groups, nc = (ng + 1) / 2;
    for (int c = threadIdx.x; c < 4 * 160; c += FL_THREADS) {
        const int a = c / 160, n = c - a * 160;
        const float* src = a == 0 ? k.ca : (a == 1 ? k.cb : (a == 2 ? k.cc : k.mean));
        coef[c] = n < K ? src[n] : 0.f;
    }
    const int64_t row = min<int64_t>(T * 32 + i, k.M - 1);
    const bool row_ok = live && T * 32 + i < k.M;
    // byte offset of the lane's 32 bytes in group 0 of its row; group g = + 64 g (an immediate), the last group's loads are
    // clamped to the row's last 16 bytes (K % 4 == 0) and masked
    const uint32_t oy = static_cast<uint32_t>(row) * k.lddy_b + 32u * s;
    const uint32_t oz = static_cast<uint32_t>(row) * k.ldz_b + 32u * s;
    const int kt = 16 * (ng - 1) + 8 * s;                                // first k of the lane in the last group
    const uint32_t t0 = static_cast<uint32_t>(min(kt, K - 4)) * 4u, t1 = static_cast<uint32_t>(min(kt + 4, K - 4)) * 4u;
    const uint32_t oy_t0 = static_cast<uint32_t>(row) * k.lddy_b + t0, oy_t1 = static_cast<uint32_t>(row) * k.lddy_b + t1;
    const uint32_t oz_t0 = static_cast<uint32_t>(row) * k.ldz_b + t0, oz_t1 = static_cast<uint32_t>(row) * k.ldz_b + t1;

    f32x16 acc[NA];
#pragma unroll
    for (int t = 0; t < NA; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    fl_u32x4 ar[4][4];                   // ring of 4 groups: dY (2 x 16 bytes), Z (2 x 16 bytes)
    // (g is a constant wherever this is called -- unrolled loops, constexpr chunk numbers -- so the switch folds away; the
    // byte offset of a group must be a literal of the instruction)
    auto a_issue = [&](int g, fl_u32x4 (&dst)[4]) {
        const char* by = k.dY;
        const char* bz = k.Z;
        if (g + 1 < ng) {
            switch (g) {
#define FL_XG(GV)                                                                              \
                case GV:                                                                       \
                    FL_XLOAD(dst[0], oy, by, GV * 64); FL_XLOAD(dst[1], oy, by, GV * 64 + 16);  \
                    FL_XLOAD(dst[2], oz, bz, GV * 64); FL_XLOAD(dst[3], oz, bz, GV * 64 + 16);  \
                    break;
                FL_XG(0) FL_XG(1) FL_XG(2) FL_XG(3) FL_XG(4) FL_XG(5) FL_XG(6) FL_XG(7) FL_XG(8) FL_XG(9) FL_XG(10) FL_XG(11)
                FL_XG(12) FL_XG(13) FL_XG(14) FL_XG(15)
#undef FL_XG
                default: break;
            }
        } else {
            const uint32_t y0 = oy_t0, y1 = oy_t1, z0 = oz_t0, z1 = oz_t1;
            FL_XLOAD(dst[0], y0, by, 0); FL_XLOAD(dst[1], y1, by, 0);
            FL_XLOAD(dst[2], z0, bz, 0); FL_XLOAD(dst[3], z1, bz, 0);
        }
    };
    auto dma_chunk = [&](int c, int buf) {
        if constexpr (SPLIT) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = wave + 4 * u;                          // piece (group, term) of column tile ct0; (wave-uniform)
                if (q < 6) {
                    const int gq = q / 3, term = q - 3 * gq;
                    const uint4* src = k.b3x + static_cast<size_t>(c) * PITCH_U4 + gq * (NT * 3 * 64) + (ct0 * 3 + term) * 64 + lane;
                    uint4* dst = lds + buf * LPITCH_U4 + q * 64;
                    __builtin_amdgcn_global_load_lds((fl_glb_ptr)src, (fl_lds_ptr)dst, 16, 0, 0);
                }
            }
        } else {
        const uint4* src = k.b3x + static_cast<size_t>(c) * PITCH_U4 + wave * (PPW * 64) + lane;
        uint4* dst = lds + buf * PITCH_U4 + wave * (PPW * 64);
#pragma unroll
        for (int u = 0; u < PPW; ++u)
            __builtin_amdgcn_global_load_lds((fl_glb_ptr)(src + u * 64), (fl_lds_ptr)(dst + u * 64), 16, 0, 0);
        }
    };

    a_issue(0, ar[0]);
    if (ng > 1) a_issue(1, ar[1]);
    if (ng > 2) a_issue(2, ar[2]);
    if (ng > 3) a_issue(3, ar[3]);
    dma_chunk(0, 0);
    {
        fl_u32x4 t0_ = ar[0][0], t1_ = ar[0][1], t2_ = ar[0][2], t3_ = ar[0][3], t4_ = ar[1][0], t5_ = ar[1][1], t6_ = ar[1][2],
                 t7_ = ar[1][3];
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(t0_), "+v"(t1_), "+v"(t2_), "+v"(t3_), "+v"(t4_), "+v"(t5_), "+v"(t6_), "+v"(t7_) :: "memory");
        ar[0][0] = t0_; ar[0][1] = t1_; ar[0][2] = t2_; ar[0][3] = t3_; ar[1][0] = t4_; ar[1][1] = t5_; ar[1][2] = t6_; ar[1][3] = t7_;
    }
    __syncthreads();                     // (the coefficients too)
    __builtin_amdgcn_sched_barrier(0);

    float* __restrict__ dzrow = (k.dZ && ct0 == 0) ? k.dZ + row * k.lddz : nullptr;   // null: the weight-gradient product recomputes dZ (swr_fl_dw_bn)
    // dZ of the lane's 8 columns of group g (the operations of act_bwd_apply_v4_kernel, bn.hip, in its order), written out,
    // and its three bf16 terms
    auto make_frag = [&](int g, const fl_u32x4 (&raw)[4], bf16x8& ah, bf16x8& am, bf16x8& al) {
        const bool last = g + 1 == ng;
        float v[8];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int kk_ = last ? min(16 * g + 8 * s + 4 * hh, K - 4) : 16 * g + 8 * s + 4 * hh;
            const float4 a4 = *reinterpret_cast<const float4*>(coef + kk_), b4 = *reinterpret_cast<const float4*>(coef + 160 + kk_),
                         c4 = *reinterpret_cast<const float4*>(coef + 320 + kk_), mu = *reinterpret_cast<const float4*>(coef + 480 + kk_);
            const float4 dy = __builtin_bit_cast(float4, raw[hh]), z = __builtin_bit_cast(float4, raw[2 + hh]);
            float4 q4 = make_float4(dy.x * a4.x, dy.y * a4.y, dy.z * a4.z, dy.w * a4.w);
            q4.x = fmaf(b4.x, z.x - mu.x, q4.x) + c4.x; q4.y = fmaf(b4.y, z.y - mu.y, q4.y) + c4.y;
            q4.z = fmaf(b4.z, z.z - mu.z, q4.z) + c4.z; q4.w = fmaf(b4.w, z.w - mu.w, q4.w) + c4.w;
            const bool ok = 16 * g + 8 * s + 4 * hh < K;          // (only the last group can be cut)
            if (!ok) q4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok && row_ok && dzrow) *reinterpret_cast<float4*>(dzrow + 16 * g + 8 * s + 4 * hh) = q4;
            v[4 * hh] = q4.x; v[4 * hh + 1] = q4.y; v[4 * hh + 2] = q4.z; v[4 * hh + 3] = q4.w;
        }
        fl_split8(v, ah, am, al);
    };
    auto mma_group = [&](const uint4* bp, bf16x8 ah, bf16x8 am, bf16x8 al) {
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, bp[(t * 3 + 0) * 64]);
            const bf16x8 b1 = __builtin_bit_cast(bf16x8, bp[(t * 3 + 1) * 64]);
            const bf16x8 b2 = __builtin_bit_cast(bf16x8, bp[(t * 3 + 2) * 64]);
            f32x16 c_ = acc[t];
            if (!SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b0, c_, 0, 0, 0);
            if (!SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b2, c_, 0, 0, 0);
            if (!SWR_X3_ON) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b1, c_, 0, 0, 0);
            c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b0, c_, 0, 0, 0);
            c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1, c_, 0, 0, 0);
            c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0, c_, 0, 0, 0);
            acc[t] = c_;
        }
    };
    // A chunk = two groups: both fragments are built first (their ring slots are free at once: the loads of the groups two
    // chunks ahead leave early), then the 2 x 6 NT products run as one block -- the two waves of a SIMD then alternate
    // between a VALU phase and a matrix phase instead of each stalling its own products behind its own arithmetic
    // (53.9 -> 50.5 us stand-alone, tools/micro/fl_probe.py).  (Tried on top: 16 consecutive k per lane half and chunk, so
    // that a lane reads / writes 64 contiguous bytes and a row's 128-byte piece is completed inside one chunk: 54.4 us.)
    auto chunk = [&](auto c_c) {
        constexpr int C = decltype(c_c)::value, BUF = C & 1;
        if (C >= nc) return;
        if (C + 1 < nc) dma_chunk(C + 1, BUF ^ 1);
        bf16x8 h0, m0_, l0, h1, m1, l1;
        const bool g1_on = 2 * C + 1 < ng;
        make_frag(2 * C, ar[2 * BUF], h0, m0_, l0);
        if (g1_on) make_frag(2 * C + 1, ar[2 * BUF + 1], h1, m1, l1);
        __builtin_amdgcn_sched_barrier(0);       // the slots' values are consumed before their registers are re-loaded
        if (2 * C + 4 < ng) a_issue(2 * C + 4, ar[2 * BUF]);
        if (2 * C + 5 < ng) a_issue(2 * C + 5, ar[2 * BUF + 1]);
        const uint4* bp = lds + BUF * LPITCH_U4 + lane;
        mma_group(bp, h0, m0_, l0);
        if (g1_on) mma_group(bp + GSTRIDE, h1, m1, l1);
        constexpr int NB = 2 * (BUF ^ 1);
        const int n_issued = __builtin_amdgcn_readfirstlane(min(2, max(0, ng - (2 * C + 4))));
        fl_u32x4 t0_ = ar[NB][0], t1_ = ar[NB][1], t2_ = ar[NB][2], t3_ = ar[NB][3], t4_ = ar[NB + 1][0], t5_ = ar[NB + 1][1],
                 t6_ = ar[NB + 1][2], t7_ = ar[NB + 1][3];
        // everything but THIS chunk's loads has landed (its dZ stores were issued in front of them; stores and loads retire
        // in order on the one counter): the weights of chunk C + 1 and the operands of groups 2 C + 2, 2 C + 3
        asm volatile("s_cmp_eq_u32 %8, 2\n\t"
                     "s_cbranch_scc1 1f\n\t"
                     "s_cmp_eq_u32 %8, 1\n\t"
                     "s_cbranch_scc1 2f\n\t"
                     "s_waitcnt vmcnt(0)\n\t"
                     "s_branch 3f\n"
                     "2:\n\t"
                     "s_waitcnt vmcnt(4)\n\t"
                     "s_branch 3f\n"
                     "1:\n\t"
                     "s_waitcnt vmcnt(8)\n"
                     "3:\n\t"
                     : "+v"(t0_), "+v"(t1_), "+v"(t2_), "+v"(t3_), "+v"(t4_), "+v"(t5_), "+v"(t6_), "+v"(t7_)
                     : "s"(n_issued)
                     : "memory", "scc");
        ar[NB][0] = t0_; ar[NB][1] = t1_; ar[NB][2] = t2_; ar[NB][3] = t3_; ar[NB + 1][0] = t4_; ar[NB + 1][1] = t5_;
        ar[NB + 1][2] = t6_; ar[NB + 1][3] = t7_;
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    chunk(std::integral_constant<int, 0>{}); chunk(std::integral_constant<int, 1>{}); chunk(std::integral_constant<int, 2>{});
    chunk(std::integral_constant<int, 3>{}); chunk(std::integral_constant<int, 4>{}); chunk(std::integral_constant<int, 5>{});
    chunk(std::integral_constant<int, 6>{}); chunk(std::integral_constant<int, 7>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!live) return;
    swr_gemm_args e;
    e.M = k.M; e.N = k.n_out; e.C = k.dX; e.ldc = k.lddx; e.bias = nullptr; e.stat_partials = nullptr; e.groups = 1; e.gsC = 0;
    e.gsBias = 0; e.c_act = 0; e.accumulate = 0;
    rows_epilogue<NA>(e, k.n_tiles, acc, 0, T, T * 32, 32 * ct0, i, s);
}

extern "C" int swr_bn_bwd_dx_supported(int K, int n_out) {
    return (K >= 16 && K <= 160 && K % 4 == 0 && n_out >= 1 && n_out <= 32 * FL_NT_MAX) ? 1 : 0;
}

extern "C" int swr_bn_bwd_dx(const swr_fl_plan* plan, const void* fl_workspace, const float* dY, int64_t lddy, const float* Z,
                             int64_t ldz, const float* ca, const float* cb, const float* cc, const float* mean, int n_out,
                             float* dZ, int64_t lddz, float* dX, int64_t lddx, void* stream) {
    FlHost h;
    int rc = fl_build(plan, h);
    if (rc != SWR_OK) return rc;
    const int K = plan->N;
    SWR_REQUIRE(fl_workspace && dY && Z && ca && cb && cc && mean && dX, SWR_ERR_ARG);      // dZ may be NULL (swr.h)
    SWR_REQUIRE(swr_bn_bwd_dx_supported(K, n_out) && lddy >= K && ldz >= K && (!dZ || lddz >= K) && lddx >= n_out, SWR_ERR_UNSUPPORTED);
    SWR_REQUIRE(lddy % 4 == 0 && ldz % 4 == 0 && (!dZ || (lddz % 4 == 0 && swr_aligned16(dZ))) && swr_aligned16(dY) && swr_aligned16(Z) &&
                    plan->B * lddy * 4 < (1ll << 32) && plan->B * ldz * 4 < (1ll << 32), SWR_ERR_ALIGN);
    if (plan->B == 0) return SWR_OK;
    FlDxK k;
    k.dY = reinterpret_cast<const char*>(dY); k.Z = reinterpret_cast<const char*>(Z);
    k.ca = ca; k.cb = cb; k.cc = cc; k.mean = mean;
    k.dZ = dZ; k.lddz = lddz; k.dX = dX; k.lddx = lddx;
    k.b3x = reinterpret_cast<const uint4*>(static_cast<const char*>(fl_workspace) + h.o.b3x);
    k.M = plan->B; k.K = K; k.n_out = n_out; k.n_groups = (K + 15) / 16; k.n_tiles = h.n_tiles;
    k.lddy_b = static_cast<uint32_t>(lddy * 4); k.ldz_b = static_cast<uint32_t>(ldz * 4);
    const int nt = (n_out + 31) / 32;
    const dim3 grid(static_cast<unsigned>(swr_ceil_div(h.n_tiles, 4)));
    const unsigned lds = static_cast<unsigned>(2 * fl_pitch_blocks(nt) * 1024 + 4 * 160 * sizeof(float));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (fl_split_columns(h.n_tiles, nt)) {
        const dim3 grid2(grid.x, static_cast<unsigned>(nt));
        const unsigned lds2 = static_cast<unsigned>(16 * 1024 + 4 * 160 * sizeof(float));
        switch (nt) {
            case 2: hipLaunchKernelGGL((fl_dx_kernel<2, 1>), grid2, dim3(FL_THREADS), lds2, st, k); break;
            case 3: hipLaunchKernelGGL((fl_dx_kernel<3, 1>), grid2, dim3(FL_THREADS), lds2, st, k); break;
            case 4: hipLaunchKernelGGL((fl_dx_kernel<4, 1>), grid2, dim3(FL_THREADS), lds2, st, k); break;
            default: hipLaunchKernelGGL((fl_dx_kernel<5, 1>), grid2, dim3(FL_THREADS), lds2, st, k); break;
        }
        return swr_launch_status();
    }
#define FL_GOX(NTV)                                                                                                     \
    do {                                                                                                                \
        if (lds >= 64 * 1024 && !swr_raise_lds(reinterpret_cast<const void*>(fl_dx_kernel<NTV>), 80 * 1024)) return SWR_ERR_LAUNCH; \
        hipLaunchKernelGGL(fl_dx_kernel<NTV>, grid, dim3(FL_THREADS), lds, st, k);                                      \
    } while (0)
    switch (nt) {
        case 1: FL_GOX(1); break;
        case 2: FL_GOX(2); break;
        case 3: FL_GOX(3); break;
        case 4: FL_GOX(4); break;
        default: FL_GOX(5); break;
    }
#undef FL_GOX
    return swr_launch_status();
}
