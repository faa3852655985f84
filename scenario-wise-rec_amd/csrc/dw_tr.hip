// Weight gradient of the fused first layer, transpose-read form (dw_tr.h).
//
//   dWp[n, c] = sum_b dZ[b, n] A'[b, c]        (reference: Linear's weight gradient under autograd, basic/layers.py:253-258
//   colsum[n] = sum_b dZ[b, n]                  for the stacked expert / gate layer of mmoe.py:37-40)
//
// The reduction runs over the BATCH, so both MFMA operands need 8 consecutive samples per lane while HBM holds both of them
// row-major by sample.  gemm_tn_x6w_kernel (gemm.hip) transposes in registers: staging threads load 8-byte column pairs of 8
// rows, split them and write column-major bf16 planes -- ~500 VALU instructions per 16-row stage on four dedicated waves.
// Here the transposition is the LDS's: every operand is stored the way it arrives (a sample's 16 columns = 32 contiguous
// bytes per bf16 term) and read back with ds_read_b64_tr_b16, which hands lane i of a 16-lane group column i of a 4 x 16 block.
//   * dZ: a lane loads 8 columns of one sample of dY and Z (2 x 16 bytes each, rows are contiguous), applies the BatchNorm
//     backward (the operations of fl_dx_kernel, first_layer.hip, in their order), splits into three bf16 terms and writes three
//     16-byte pieces, lane-linear (conflict-free).
//   * A', real groups: the pieces are ALREADY split in HBM (table shadows / the keys launch's a3f section, 48 bytes per
//     sample and 8 columns): three LDS-DMA copies per 16-column group and stage through the lane's piece offset, no VALU at all.
//   * A', one-hot columns: 32 bits per sample and column tile (mask_t), expanded to bf16 1.0 / 0.0 in the B fragment.
// One workgroup per batch split (256 at config 2 = one per CU), 8 waves, ALL of them multiply: the 45 output tiles are dealt
// to the waves at compile time (<= 6 accumulator tiles each, 27 of 210 product units per 16-sample step), and all of them
// stage (one or two 16-column groups of each operand per 32-sample stage).  Two LDS buffers, one barrier per stage.
// Summation order differs from the written-block product (k runs over the samples of a split in tr-read order), so results
// are not bit-identical to gemm_tn_x6_kernel's; they are deterministic (fixed partial order, tn_reduce4_kernel).
#include "dw_tr.h"

#include <cstdlib>

#include "common.h"
#include "rows_epilogue.h"
#include "split3.h"

#define DT_THREADS 512
#define DT_BLK 1152                       // bytes per (16-column group, term) block: 32 samples x 32 bytes + 128: the blocks of an
                                          // even / odd group pair start 128 bytes apart modulo 256 (the two 16-lane groups of a
                                          // half wave read them in the same cycle)
#define DT_MK_BYTES 512                   // one-hot words of a stage: [4][32] u32
typedef __attribute__((address_space(3))) char* dt_lds;
typedef __attribute__((address_space(3))) void* dt_lds_v;
typedef __attribute__((address_space(1))) const void* dt_glb;
typedef __attribute__((address_space(3))) bf16x4* dt_tr_ptr;
typedef unsigned dt_u32x4 __attribute__((ext_vector_type(4)));
typedef float dt_f32x4 __attribute__((ext_vector_type(4)));

template <int PT, int RQ, int OQ>
struct DtCfg {
    static constexpr int NCA = 2 * PT, NCB = 2 * RQ;
    static constexpr int ZT_BYTES = NCA * 3 * DT_BLK, AT_BYTES = NCB * 3 * DT_BLK;
    static constexpr int BUF = ZT_BYTES + AT_BYTES + DT_MK_BYTES;
    static constexpr int NCOEF = 32 * PT;
    static constexpr int LDS = 2 * BUF + 4 * NCOEF * 4;
};

// ---- the deal: (p, q) output tiles of every wave; q < RQ: six products per 16-sample step, q >= RQ (one-hot columns): three
struct DtDeal {
    int n[8];
    int p[8][8], q[8][8];
    bool ok;
};
constexpr DtDeal dt_deal(int PT, int RQ, int OQ) {
    DtDeal d{};
    d.ok = true;
    if (PT == 5 && RQ == 5 && OQ == 4) {
        // config 2 (148 x (160 + 128)): 27 units for waves 0-6, 21 for wave 7; waves w and w + 4 share a SIMD (54 / 54 / 54 / 48),
        // the lighter SIMD's waves take the two extra staging groups.  Tiles of a wave share their A fragments (a p-triple) or
        // their B fragment.
        constexpr int T[8][6][2] = {
            {{0, 0}, {1, 0}, {2, 0}, {0, 5}, {1, 5}, {2, 5}}, {{0, 1}, {1, 1}, {2, 1}, {0, 6}, {1, 6}, {2, 6}},
            {{0, 2}, {1, 2}, {2, 2}, {0, 7}, {1, 7}, {2, 7}}, {{0, 3}, {1, 3}, {2, 3}, {0, 8}, {1, 8}, {2, 8}},
            {{0, 4}, {1, 4}, {2, 4}, {3, 5}, {3, 6}, {3, 7}}, {{4, 0}, {3, 0}, {4, 1}, {3, 1}, {3, 8}, {-1, -1}},
            {{3, 2}, {4, 2}, {3, 3}, {4, 3}, {4, 8}, {-1, -1}}, {{3, 4}, {4, 4}, {4, 5}, {4, 6}, {4, 7}, {-1, -1}}};
        for (int w = 0; w < 8; ++w) {
            int n = 0;
            for (int t = 0; t < 6; ++t)
                if (T[w][t][0] >= 0) { d.p[w][n] = T[w][t][0]; d.q[w][n] = T[w][t][1]; ++n; }
            d.n[w] = n;
        }
        return d;
    }
    // any other shape: longest-processing-time first (six-product tiles, then three-product tiles, each to the least loaded wave)
    int load[8] = {};
    for (int kind = 0; kind < 2; ++kind)
        for (int q = kind ? RQ : 0; q < (kind ? RQ + OQ : RQ); ++q)
            for (int p = 0; p < PT; ++p) {
                int w = 0;
                for (int v = 1; v < 8; ++v)
                    if (load[v] < load[w]) w = v;
                if (d.n[w] >= 8) { d.ok = false; continue; }
                d.p[w][d.n[w]] = p; d.q[w][d.n[w]] = q; ++d.n[w];
                load[w] += kind ? 3 : 6;
            }
    return d;
}
// staging duty j (0, 1) of wave W among n 16-column groups: group W, and groups 8 / 9 to the waves of the SIMD with the fewest products
constexpr int dt_cg(int W, int j, int n) {
    if (j == 0) return W < n ? W : -1;
    if (W == 3 && n > 8) return 8;
    if (W == 7 && n > 9) return 9;
    return -1;
}

#ifdef DT_STAMPS
// timing experiments (tools/micro/dw_tr_harness.hip): shader-clock stamps of one wave of workgroup 0
__device__ uint64_t* dt_stamp_buf;
#ifndef DT_STAMP_WAVE
#define DT_STAMP_WAVE 0
#endif
#define DT_STAMP(idx) do { if (W == DT_STAMP_WAVE && blockIdx.x == 0 && lane == 0) dt_stamp_buf[idx] = __builtin_readcyclecounter(); } while (0)
#else
#define DT_STAMP(idx) do { } while (0)
#endif
#ifdef DT_STAMPS_FINE
#define DT_FINE(k) do { if (st == 3) { __builtin_amdgcn_sched_barrier(0); DT_STAMP(16 + (k)); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define DT_FINE(k) do { } while (0)
#endif

// LDS-DMA copies as inline asm.  Through __builtin_amdgcn_global_load_lds hipcc tracks the copy as a pending LDS store and, unable to
// prove that the fragment reads of the OTHER buffer do not alias it (runtime buffer index, lane-dependent offsets), puts an
// s_waitcnt vmcnt(0) in front of the first LDS read after every copy: a full memory round trip at the top of every stage
// (measured with the stage stamps: 2 300 - 5 300 of a stage's 10 000 cycles).  The asm form is invisible to that tracking; the kernel
// waits for its copies itself (counted vmcnt at the end of the stage).  hipcc's own vmcnt waits for register loads count only the
// loads it knows: with copies in between they wait for MORE than they need, never for less.
__device__ __forceinline__ void dt_dma16(uint32_t lds_addr, uint32_t voff, const char* sbase) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase) : "memory", "m0");
}
__device__ __forceinline__ void dt_dma4(uint32_t lds_addr, uint32_t voff, const void* sbase) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase) : "memory", "m0");
}
__device__ __forceinline__ uint32_t dt_lds_addr(dt_lds p) {
    return __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p)));
}

struct DtRaw {
    float4 y, z;
};
// dZ staging units of a 32-sample stage: unit u = (128-byte column block lc = u % PT, 8-row block ro = u / PT); a wave's load of a
// unit reads 8 rows x 128 contiguous bytes (8 whole cache lines: lane = (row, 16-byte chunk)) of dY and of Z.  The first form --
// lane = (row, 8 columns of a 16-column group) -- touched 32 half lines per load and every line four times, and with the scattered
// LDS-DMA copies the waves spent a third of every stage queueing at the vector-memory issue port.
constexpr int dt_unit(int W, int j, int n_units) { return W + 8 * j < n_units ? W + 8 * j : -1; }

template <int PT, int RQ, int OQ, int W>
__device__ __forceinline__ void dt_wave(const DwTrArgs& a, dt_lds lds, const int lane) {
    using C = DtCfg<PT, RQ, OQ>;
    constexpr DtDeal D = dt_deal(PT, RQ, OQ);
    static_assert(D.ok, "more than 8 tiles for one wave");
    constexpr int NW = D.n[W];
    constexpr int NU = 4 * PT;                                                                        // dZ staging units per stage
    constexpr int NZ = (dt_unit(W, 0, NU) >= 0 ? 1 : 0) + (dt_unit(W, 1, NU) >= 0 ? 1 : 0) + (dt_unit(W, 2, NU) >= 0 ? 1 : 0);
    constexpr int NA = (dt_cg(W, 0, C::NCB) >= 0 ? 1 : 0) + (dt_cg(W, 1, C::NCB) >= 0 ? 1 : 0);      // A' groups (those below NR)
    DT_STAMP(0);
    const int i = lane & 31, s = lane >> 5;
    const int split = blockIdx.x;
    const int64_t ms = static_cast<int64_t>(split) * a.rows_per_split;
    const int64_t me = min(ms + a.rows_per_split, a.M);
    const int n_stages = static_cast<int>((me - ms + 31) / 32);
    const int K1 = a.K1;

    // ---- staging: lane = (sample r of the stage, column half hb of the 16-column group)
    const int r = lane >> 1, hb = lane & 1;
    const int ur = lane >> 3, uc = lane & 7;                               // dZ staging: row of the unit's 8, 16-byte chunk of its 128 bytes
    // (32-bit row / tile counters: M < 2^31 and M * ld * 4 < 2^32 are the launcher's conditions.  Loads past the split's last stage
    // re-read rows of the next split -- or row M - 1 -- and are never used.)
    const uint32_t m_last = static_cast<uint32_t>(a.M - 1), ms32 = static_cast<uint32_t>(ms), me32 = static_cast<uint32_t>(me);
    const uint32_t t_last = static_cast<uint32_t>((a.M + 31) / 32 - 1);
    auto raw_load = [&](int u, int st, DtRaw& d) {
        const int lc = u % PT, ro = u / PT;
        const uint32_t row = min(ms32 + 32u * static_cast<uint32_t>(st) + static_cast<uint32_t>(8 * ro + ur), m_last);
        const uint32_t cb = static_cast<uint32_t>(min(32 * lc + 4 * uc, K1 - 4)) * 4u;
        d.y = *reinterpret_cast<const float4*>(a.dY + (row * a.lddy_b + cb));
        d.z = *reinterpret_cast<const float4*>(a.Z + (row * a.ldz_b + cb));
    };
    auto vo_load = [&](int cg, int st) -> uint32_t {
        const uint32_t T = min((ms32 >> 5) + static_cast<uint32_t>(st), t_last);
        return a.voff[(T * static_cast<uint32_t>(a.NR) + static_cast<uint32_t>(min(cg, a.NR - 1))) * 64u + static_cast<uint32_t>(hb * 32 + r)];
    };
    float cs[NZ > 0 ? NZ : 1][4];
#pragma unroll
    for (int j = 0; j < NZ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) cs[j][e] = 0.f;
    dt_lds coef = lds + 2 * C::BUF;
    // dZ of the lane's 4 columns (fl_dx_kernel's operations in its order), its column sums, its three bf16 terms -> the stage's buffer
    auto split_store = [&](int j, int u, int st, const DtRaw& d, dt_lds zt) {
        const int lc = u % PT, ro = u / PT;
        const bool row_ok = ms32 + 32u * static_cast<uint32_t>(st) + static_cast<uint32_t>(8 * ro + ur) < me32;
        const int c0 = 32 * lc + 4 * uc;
        const int cc_ = min(c0, K1 - 4);
        const dt_f32x4 a4 = *reinterpret_cast<__attribute__((address_space(3))) const dt_f32x4*>(coef + 4 * cc_);
        const dt_f32x4 b4 = *reinterpret_cast<__attribute__((address_space(3))) const dt_f32x4*>(coef + 4 * (C::NCOEF + cc_));
        const dt_f32x4 c4 = *reinterpret_cast<__attribute__((address_space(3))) const dt_f32x4*>(coef + 4 * (2 * C::NCOEF + cc_));
        const dt_f32x4 mu = *reinterpret_cast<__attribute__((address_space(3))) const dt_f32x4*>(coef + 4 * (3 * C::NCOEF + cc_));
        const float4 dy = d.y, z = d.z;
        float v[4];
        v[0] = fmaf(b4.x, z.x - mu.x, dy.x * a4.x) + c4.x; v[1] = fmaf(b4.y, z.y - mu.y, dy.y * a4.y) + c4.y;
        v[2] = fmaf(b4.z, z.z - mu.z, dy.z * a4.z) + c4.z; v[3] = fmaf(b4.w, z.w - mu.w, dy.w * a4.w) + c4.w;
        if (!(row_ok && c0 < K1)) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }
#pragma unroll
        for (int e = 0; e < 4; ++e) cs[j][e] += v[e];
        bf16x4 th, tm, tl;
        SPLIT3_PAIR(v[0], v[1], th, tm, tl, 0);
        SPLIT3_PAIR(v[2], v[3], th, tm, tl, 2);
        dt_lds dst = zt + ((2 * lc + (uc >> 2)) * 3) * DT_BLK + (8 * ro + ur) * 32 + (uc & 3) * 8;
        *reinterpret_cast<__attribute__((address_space(3))) bf16x4*>(dst) = th;
        *reinterpret_cast<__attribute__((address_space(3))) bf16x4*>(dst + DT_BLK) = tm;
        *reinterpret_cast<__attribute__((address_space(3))) bf16x4*>(dst + 2 * DT_BLK) = tl;
    };
    // the three terms of a real group of A' for the stage's 32 samples: three 1-KB LDS-DMA copies through the lane's piece offset
    auto dma_group = [&](int cg, uint32_t vo, dt_lds at) {
        if (cg >= a.NR) return;                                             // (wave-uniform)
#pragma unroll
        for (int t = 0; t < 3; ++t) dt_dma16(dt_lds_addr(at + (cg * 3 + t) * DT_BLK), vo + 16u * t, a.ws);
    };
    // the stage's one-hot words [4][32] (wave 5): lane -> word 2 u + (lane >> 5) of sample lane & 31
    auto dma_mask = [&](int st, dt_lds mk) {
        if (OQ == 0 || W != 5) return;
        const uint32_t row = min(ms32 + 32u * static_cast<uint32_t>(st) + static_cast<uint32_t>(i), m_last);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int o = min(2 * u + s, OQ - 1);
            dt_dma4(dt_lds_addr(mk + u * 256), (static_cast<uint32_t>(o) * static_cast<uint32_t>(a.M) + row) * 4u, a.mask_t);
        }
    };

    f32x16 acc[NW > 0 ? NW : 1];
#pragma unroll
    for (int t = 0; t < NW; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // ---- fragment addresses: 16-lane group gq = lane >> 4 reads rows 8 (gq >> 1) + (li >> 2) (+ 4 for the second read) of the
    // even (gq & 1 = 0) / odd 16-column group of its tile, 8-byte chunk li & 3: lane (i, s) receives samples 8 s + j of the
    // 16-sample step for column i of the tile -- the MFMA operand layout
    const int gq = lane >> 4, li = lane & 15;
    const uint32_t tr_lane = static_cast<uint32_t>((gq & 1) * (3 * DT_BLK) + (8 * (gq >> 1) + (li >> 2)) * 32 + (li & 3) * 8);
    auto frag = [&](dt_lds base, int cg2, int term, int ks) -> bf16x8 {
#ifdef DT_NO_FRAG
        dt_u32x4 fk = {static_cast<unsigned>(lane + cg2), static_cast<unsigned>(term + ks), static_cast<unsigned>(lane), 0x3C003C00u};
        asm volatile("" : "+v"(fk));
        return __builtin_bit_cast(bf16x8, fk);
#endif
        dt_lds p = base + tr_lane + (cg2 * 3 + term) * DT_BLK + ks * 512;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((dt_tr_ptr)(p));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((dt_tr_ptr)(p + 128));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto oh_frag = [&](dt_lds mk, int o, int ks) -> bf16x8 {
#ifdef DT_NO_FRAG
        dt_u32x4 fk = {static_cast<unsigned>(lane + o), static_cast<unsigned>(ks), static_cast<unsigned>(lane), 0x3C003C00u};
        asm volatile("" : "+v"(fk));
        return __builtin_bit_cast(bf16x8, fk);
#endif
        dt_lds p = mk + o * 128 + ks * 64 + s * 32;
        const dt_u32x4 w0 = *reinterpret_cast<__attribute__((address_space(3))) const dt_u32x4*>(p);
        const dt_u32x4 w1 = *reinterpret_cast<__attribute__((address_space(3))) const dt_u32x4*>(p + 16);
        dt_u32x4 f;
        f[0] = (__builtin_amdgcn_ubfe(w0[0], i, 1) | (__builtin_amdgcn_ubfe(w0[1], i, 1) << 16)) * 0x3F80u;
        f[1] = (__builtin_amdgcn_ubfe(w0[2], i, 1) | (__builtin_amdgcn_ubfe(w0[3], i, 1) << 16)) * 0x3F80u;
        f[2] = (__builtin_amdgcn_ubfe(w1[0], i, 1) | (__builtin_amdgcn_ubfe(w1[1], i, 1) << 16)) * 0x3F80u;
        f[3] = (__builtin_amdgcn_ubfe(w1[2], i, 1) | (__builtin_amdgcn_ubfe(w1[3], i, 1) << 16)) * 0x3F80u;
        return __builtin_bit_cast(bf16x8, f);
    };
    // The products of a stage as a software pipeline over PAIRS of tiles: the fragments of the next pair are requested, THEN the
    // products of the current pair are issued back to back, alternating between the two accumulators.  Why pairs: a lone chain of
    // products on ONE accumulator issues every ~43 cycles, two interleaved chains every 32 (the pipe's rate) -- and anything hipcc
    // places between two MFMAs on the same accumulator costs another ~43 cycles (MI355X_MICROARCH.md), so the blocks are fenced
    // with sched_barrier.  Fragments are named by (16-sample step, p) / (16-sample step, q): straight-line code, the register
    // allocator overlaps their lifetimes.  The LAST fragment requested for a pair is an operand of the block's FIRST product: LDS
    // reads return in order, so the one wait in front of the block covers them all.
    constexpr int NP = (NW + 1) / 2;
    auto mma_stage = [&](dt_lds buf, auto&& hook) {
        dt_lds zt = buf, at = buf + C::ZT_BYTES, mk = buf + C::ZT_BYTES + C::AT_BYTES;
        bf16x8 af[2][PT][3];
        bf16x8 bq[2][RQ + OQ][3];
        auto load_tile = [&](int ks, int t, bool b_last) {                  // (compile-time arguments after unrolling)
            const int p = D.p[W][t], q = D.q[W][t];
            bool need_a = true, need_b = true;
#pragma unroll
            for (int v = 0; v < t; ++v) {
                if (D.p[W][v] == p) need_a = false;
                if (D.q[W][v] == q) need_b = false;
            }
            if (need_b && q >= RQ) bq[ks][q][0] = oh_frag(mk, q - RQ, ks);
            if (need_b && q < RQ && !b_last) {
#pragma unroll
                for (int term = 1; term <= 3; ++term) bq[ks][q][term % 3] = frag(at, 2 * q, term % 3, ks);
            }
            if (need_a) {
#pragma unroll
                for (int term = 0; term < 3; ++term) af[ks][p][term] = frag(zt, 2 * p, term, ks);       // (the low term last)
            }
            if (need_b && q < RQ && b_last) {
#pragma unroll
                for (int term = 1; term <= 3; ++term) bq[ks][q][term % 3] = frag(at, 2 * q, term % 3, ks);   // (the high term last)
            }
        };
        auto load_pair = [&](int U) {
            const int ks = U / NP, v = U - ks * NP;
            // the pair's first product multiplies tile 2 v's low A term with its high B term: whichever of the two is requested is
            // requested last (B if both)
            if (2 * v + 1 < NW) load_tile(ks, 2 * v + 1, false);
            load_tile(ks, 2 * v, true);
        };
        auto prod = [&](int ks, int t, int j, f32x16 c_) -> f32x16 {         // product j of tile t
            const int p = D.p[W][t], q = D.q[W][t];
            if (q < RQ) {
                constexpr int ta[6] = {2, 0, 1, 1, 0, 0}, tb[6] = {0, 2, 1, 0, 1, 0};
                return __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][p][ta[j]], bq[ks][q][tb[j]], c_, 0, 0, 0);
            }
            constexpr int to[3] = {2, 1, 0};
            return __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][p][to[j]], bq[ks][q][0], c_, 0, 0, 0);
        };
        load_pair(0);
#pragma unroll
        for (int U = 0; U < 2 * NP; ++U) {
            const int ks = U / NP, v = U - ks * NP;
            const int t0 = 2 * v, t1 = 2 * v + 1;
            const int n0 = D.q[W][t0] < RQ ? 6 : 3;
            const int n1 = t1 < NW ? (D.q[W][t1] < RQ ? 6 : 3) : 0;
            if (U + 1 < 2 * NP) load_pair(U + 1);
            __builtin_amdgcn_sched_barrier(0);
            f32x16 c0 = acc[t0], c1 = acc[t1 < NW ? t1 : t0];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                // (SWR_X3 measurement builds skip the small terms: products 0-2 of a real tile, product 0 of a one-hot tile)
                if (j < n0 && !(SWR_X3_ON && j < (n0 == 6 ? 3 : 1))) c0 = prod(ks, t0, j, c0);
                if (j < n1 && !(SWR_X3_ON && j < (n1 == 6 ? 3 : 1))) c1 = prod(ks, t1, j, c1);
            }
            acc[t0] = c0;
            if (t1 < NW) acc[t1] = c1;
            __builtin_amdgcn_sched_barrier(0);
            hook(U);
        }
    };

    // ---- prologue: the operands of stages 0 AND 1 are requested at once (one memory round trip in front of the first products
    // instead of two: every CU's workgroup is here at the same time, ~110 KB per CU in flight); stage 0 goes into buffer 0
    DtRaw raw[NZ > 0 ? NZ : 1];
    uint32_t vo[NA > 0 ? NA : 1];
    {
        DtRaw raw0[NZ > 0 ? NZ : 1];
        uint32_t vo0[NA > 0 ? NA : 1];
#pragma unroll
        for (int j = 0; j < NA; ++j) vo0[j] = vo_load(dt_cg(W, j, C::NCB), 0);
#pragma unroll
        for (int j = 0; j < NZ; ++j) raw_load(dt_unit(W, j, NU), 0, raw0[j]);
#pragma unroll
        for (int j = 0; j < NA; ++j) vo[j] = vo_load(dt_cg(W, j, C::NCB), 1);
#pragma unroll
        for (int j = 0; j < NZ; ++j) raw_load(dt_unit(W, j, NU), 1, raw[j]);
        __builtin_amdgcn_sched_barrier(0);
        // coefficients (zero past K1) and the A' groups that are never copied (NR < 2 RQ), by all waves of the workgroup
        const int tid = threadIdx.x;
        // (four stores per index: one loop with the source chosen by index put the four pointers into scratch)
        for (int n = tid; n < C::NCOEF; n += DT_THREADS) {
            *reinterpret_cast<__attribute__((address_space(3))) float*>(coef + 4 * n) = n < K1 ? a.ca[n] : 0.f;
            *reinterpret_cast<__attribute__((address_space(3))) float*>(coef + 4 * (C::NCOEF + n)) = n < K1 ? a.cb[n] : 0.f;
            *reinterpret_cast<__attribute__((address_space(3))) float*>(coef + 4 * (2 * C::NCOEF + n)) = n < K1 ? a.cc[n] : 0.f;
            *reinterpret_cast<__attribute__((address_space(3))) float*>(coef + 4 * (3 * C::NCOEF + n)) = n < K1 ? a.mean[n] : 0.f;
        }
        const int dead0 = a.NR * 3 * DT_BLK, dead1 = C::NCB * 3 * DT_BLK;
        for (int b = dead0 + 16 * tid; b < dead1; b += 16 * DT_THREADS) {
            const dt_u32x4 z4 = {0u, 0u, 0u, 0u};
            *reinterpret_cast<__attribute__((address_space(3))) dt_u32x4*>(lds + C::ZT_BYTES + b) = z4;
            *reinterpret_cast<__attribute__((address_space(3))) dt_u32x4*>(lds + C::BUF + C::ZT_BYTES + b) = z4;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NA; ++j) dma_group(dt_cg(W, j, C::NCB), vo0[j], lds + C::ZT_BYTES);
        dma_mask(0, lds + C::ZT_BYTES + C::AT_BYTES);
        __syncthreads();                                                   // (the coefficients)
#pragma unroll
        for (int j = 0; j < NZ; ++j) split_store(j, dt_unit(W, j, NU), 0, raw0[j], lds);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    DT_STAMP(1);

    // Waves w and w + 4 share a SIMD: the first four stage the next stage's dZ (VALU) and then multiply, the other four multiply
    // first -- the matrix pipe has work from one of the two while the other splits.  At the end of a stage everything but the
    // register loads issued LAST (dY / Z / piece offsets of the stage after next) has landed: the LDS-DMA copies of the next stage
    // were issued in front of them, and vmcnt retires in order.
    // A stage: waves w and w + 4 share a SIMD -- the first four stage the next stage's dZ (VALU) and then multiply, the other four
    // multiply first: the matrix pipe has work from one of the two while the other splits.  The LDS-DMA copies of the next stage's
    // A' groups are issued one group per product block, behind the first blocks (all eight waves issuing them at the top of the stage
    // queued at the vector-memory port for ~1 100 cycles per stage).  Memory operations are pinned with sched_barrier: hipcc
    // otherwise sinks the piece-offset loads behind the dY / Z loads, and the wait for an offset then waits for all of those.
    // End of stage: waves 0-3 issued their dY / Z loads before the products (everything has landed: vmcnt(0)); waves 4-7 issue them
    // last and leave them in flight (vmcnt retires in order: "all but the last 2 NZ" = the copies have landed).
#ifndef DT_ROLE
#define DT_ROLE 0            // 0: waves 0-3 split first; 1: waves 4-7 split first; 2: all split first; 3: all multiply first (experiments)
#endif
    constexpr bool SPLIT_FIRST = DT_ROLE == 0 ? (W < 4) : (DT_ROLE == 1 ? (W >= 4) : DT_ROLE == 2);
    constexpr int N_LATE = SPLIT_FIRST ? 0 : 2 * NZ;
    for (int st = 0; st < n_stages; ++st) {
        dt_lds cur = lds + (st & 1) * C::BUF, nxt = lds + ((st + 1) & 1) * C::BUF;
        const bool more = st + 1 < n_stages;
        DT_FINE(0);
        uint32_t vo_next[NA > 0 ? NA : 1];
#ifndef DT_NO_DMA
#pragma unroll
        for (int j = 0; j < NA; ++j) vo_next[j] = vo_load(dt_cg(W, j, C::NCB), st + 2);
        __builtin_amdgcn_sched_barrier(0);
#endif
        auto copies = [&](int U) {                                          // behind product block U: A' group U, the mask words
#ifndef DT_NO_DMA
            if (more) {
                if (U < NA) dma_group(dt_cg(W, U < NA ? U : 0, C::NCB), vo[U < NA ? U : 0], nxt + C::ZT_BYTES);
                if (U == 0) dma_mask(st + 1, nxt + C::ZT_BYTES + C::AT_BYTES);
            }
            __builtin_amdgcn_sched_barrier(0);
#endif
        };
        auto stage_dz = [&]() {
#ifndef DT_NO_SPLIT
            if (more) {
#pragma unroll
                for (int j = 0; j < NZ; ++j) split_store(j, dt_unit(W, j, NU), st + 1, raw[j], nxt);
            }
            __builtin_amdgcn_sched_barrier(0);
#endif
        };
        auto request_dz = [&]() {
#ifndef DT_NO_RAW
#pragma unroll
            for (int j = 0; j < NZ; ++j) raw_load(dt_unit(W, j, NU), st + 2, raw[j]);
            __builtin_amdgcn_sched_barrier(0);
#endif
        };
        if (SPLIT_FIRST) {
            stage_dz();
            request_dz();
            DT_FINE(1);
        }
#ifndef DT_NO_MMA
        mma_stage(cur, copies);
#else
        for (int U = 0; U < 2; ++U) copies(U);
#endif
        __builtin_amdgcn_sched_barrier(0);
        DT_FINE(2);
#ifndef DT_NO_DMA
#pragma unroll
        for (int j = 0; j < NA; ++j) vo[j] = vo_next[j];
#endif
        if (!SPLIT_FIRST) {
            stage_dz();
            request_dz();
        }
        DT_FINE(3);
#if defined(DT_NO_RAW) || defined(DT_NO_DMA)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N_LATE) : "memory");
#endif
        DT_FINE(4);
#ifndef DT_NO_BARRIER
        __builtin_amdgcn_s_barrier();
#endif
        __builtin_amdgcn_sched_barrier(0);
        DT_FINE(5);
        DT_STAMP(2 + st);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if (a.part_cs) {
        // column sums of dZ: a unit's lanes hold 8 rows x 4 columns; rows are added inside the wave, the four 8-row blocks of a
        // column (units of different waves) through LDS in block order
        dt_lds red = lds;                                                   // [4 row blocks][32 PT] floats (the stage buffers are dead)
#pragma unroll
        for (int j = 0; j < NZ; ++j) {
            const int u = dt_unit(W, j, NU), lc = u % PT, ro = u / PT;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = cs[j][e];
#pragma unroll
                for (int off = 8; off < 64; off <<= 1) v += __shfl_xor(v, off);
                if (ur == 0) *reinterpret_cast<__attribute__((address_space(3))) float*>(red + 4 * (ro * 32 * PT + 32 * lc + 4 * uc + e)) = v;
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < K1; c += DT_THREADS) {
            float v = *reinterpret_cast<__attribute__((address_space(3))) const float*>(red + 4 * c);
#pragma unroll
            for (int ro = 1; ro < 4; ++ro) v += *reinterpret_cast<__attribute__((address_space(3))) const float*>(red + 4 * (ro * 32 * PT + c));
            a.part_cs[static_cast<int64_t>(split) * K1 + c] = v;
        }
    }
    // ---- partial tile of this split (tn_reduce4_kernel adds the splits in order)
    float* __restrict__ P = a.part + static_cast<int64_t>(split) * K1 * a.k2p;
#pragma unroll
    for (int t = 0; t < NW; ++t) {
        const int q = 32 * D.q[W][t] + i;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int p = 32 * D.p[W][t] + (e & 3) + 8 * (e >> 2) + 4 * s;
#ifdef DT_NO_STORE
            if (p < K1 && q < a.K2 && acc[t][e] == 123.456f) P[static_cast<int64_t>(p) * a.k2p + q] = acc[t][e];
#else
            if (p < K1 && q < a.K2) P[static_cast<int64_t>(p) * a.k2p + q] = acc[t][e];
#endif
        }
    }
#ifdef DT_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DT_STAMP(2 + n_stages);
#endif
}

template <int PT, int RQ, int OQ>
__global__ __launch_bounds__(DT_THREADS) void dw_tr_kernel(const DwTrArgs a) {
    extern __shared__ __attribute__((aligned(16))) char dt_smem[];
    dt_lds lds = (dt_lds)dt_smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    switch (wave) {
        case 0: dt_wave<PT, RQ, OQ, 0>(a, lds, lane); break;
        case 1: dt_wave<PT, RQ, OQ, 1>(a, lds, lane); break;
        case 2: dt_wave<PT, RQ, OQ, 2>(a, lds, lane); break;
        case 3: dt_wave<PT, RQ, OQ, 3>(a, lds, lane); break;
        case 4: dt_wave<PT, RQ, OQ, 4>(a, lds, lane); break;
        case 5: dt_wave<PT, RQ, OQ, 5>(a, lds, lane); break;
        case 6: dt_wave<PT, RQ, OQ, 6>(a, lds, lane); break;
        default: dt_wave<PT, RQ, OQ, 7>(a, lds, lane); break;
    }
}

// instantiated shapes: (row tiles of dZ^T, real column tiles of A', one-hot column tiles)
#define DT_SHAPES(X) X(5, 5, 4)

static int dt_mode = -1;
extern "C" int swr_dw_tr_mode(int set) {
    if (dt_mode < 0) { const char* e = getenv("SWR_DW_TR"); dt_mode = (e && e[0] == '0') ? 0 : 1; }
    const int prev = dt_mode;
    if (set >= 0) dt_mode = set ? 1 : 0;
    return prev;
}

bool dw_tr_shape_ok(int K1, int K2, int NR) {
    if (!swr_dw_tr_mode(-1) || K1 < 4 || K1 % 4 || NR < 1 || K2 < 16 * NR) return false;
    const int pt = (K1 + 31) / 32, rq = (16 * NR + 31) / 32, oq = (K2 - 16 * NR + 31) / 32;
    if ((16 * NR) % 32 != 0 && K2 > 16 * NR) return false;                 // the one-hot block starts on a 32-column tile
#define X(PTV, RQV, OQV) if (pt == PTV && rq == RQV && oq == OQV) return true;
    DT_SHAPES(X)
#undef X
    return false;
}

int dw_tr_launch(const DwTrArgs& a, hipStream_t st) {
    if (!dw_tr_shape_ok(a.K1, a.K2, a.NR) || a.rows_per_split % 32 || a.n_splits < 1 || !a.dY || !a.Z || !a.ws || !a.voff || !a.part)
        return SWR_ERR_UNSUPPORTED;
    if (a.K2 > 16 * a.NR && !a.mask_t) return SWR_ERR_ARG;
    if (a.M < 1 || a.M >= (1ll << 26) || a.M * a.lddy_b >= (1ll << 32) || a.M * a.ldz_b >= (1ll << 32)) return SWR_ERR_UNSUPPORTED;   // 32-bit offsets
    if ((reinterpret_cast<uintptr_t>(a.dY) & 15u) || (reinterpret_cast<uintptr_t>(a.Z) & 15u) || a.lddy_b % 16 || a.ldz_b % 16)
        return SWR_ERR_ALIGN;
    const int pt = (a.K1 + 31) / 32, rq = (16 * a.NR + 31) / 32, oq = (a.K2 - 16 * a.NR + 31) / 32;
    DwTrArgs k = a;
#define X(PTV, RQV, OQV)                                                                                                    \
    if (pt == PTV && rq == RQV && oq == OQV) {                                                                              \
        const void* fn = reinterpret_cast<const void*>(dw_tr_kernel<PTV, RQV, OQV>);                                        \
        const unsigned lds = static_cast<unsigned>(DtCfg<PTV, RQV, OQV>::LDS);                                              \
        if (!swr_raise_lds(fn, static_cast<int>(lds))) return SWR_ERR_LAUNCH;                                               \
        void* kargs[] = {&k};                                                                                               \
        if (hipLaunchKernel(fn, dim3(static_cast<unsigned>(a.n_splits)), dim3(DT_THREADS), kargs, lds, st) != hipSuccess)   \
            return SWR_ERR_LAUNCH;                                                                                          \
        return SWR_OK;                                                                                                      \
    }
    DT_SHAPES(X)
#undef X
    return SWR_ERR_UNSUPPORTED;
}
