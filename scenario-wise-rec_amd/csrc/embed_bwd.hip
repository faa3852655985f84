// K3: backward of the embedding lookup for ALL features of a batch at once
// (replaces F_s x aten::embedding_dense_backward).
//
//   0. direct     : lookups of SMALL dense-gradient tables (<= DIRECT_MAX_PARTS x 64 KB of accumulators; 16 row ranges while
//                   the slabs stay small) skip the sort: a workgroup owns a span of adjacent lookup columns and a chunk of
//                   samples, adds the fixed-point gradients into LDS accumulators (ds_add_u64) and STORES them into the
//                   slab of its (lookup, chunk); the finalise launch adds a table's slabs
//   0'. segsum    : MID-SIZE dense-gradient tables (17 .. 4096 rows, dim a multiple of 8 up to 64) at batches >= 2048 are
//                   summed on the matrix pipes: dEmb_t = OneHot_t^T dE_t, the one-hot fragment built in registers from
//                   the keys (0 / 1 are exact in bf16), dE split into three bf16 terms (exact), fp32 accumulation by
//                   v_mfma_f32_16x16x32_bf16 in a fixed order, per-batch-split partial tables added in split order by
//                   the finalise launch: deterministic, no atomics, no LDS
//   1. build_keys : the other entries (row, slot << 24 | sample) written grouped by table (one segment per table)
//   2. sort       : segmented LSD radix sort of every table segment by row, 8 bits per pass, stable; pass p
//                   only does work for tables whose row ids need more than 8p bits (others copy through).
//                   Hand-written (histogram -> per-table scan -> ranked scatter), no atomics on global
//                   memory, no memset nodes, no inter-workgroup spinning: safe to capture and replay.
//   3. reduce     : each group of LPE lanes walks a fixed chunk of 8-32 sorted entries of one table, sums
//                   runs of equal rows in registers and flushes each run (plain stores for runs that lie
//                   inside the chunk, two 64-bit integer atomics for runs cut by a chunk boundary)
//   4. finalise   : integer accumulators -> fp32 gradients (dense tables) or per-row entries (sparse), one launch
//
// Accumulation is dual-limb fixed point: x * 2^60 (truncated) = hi * 2^40 + lo, hi in 2^-20 units, lo in 2^-60
// units.  Both limbs are integers, so the sum is exact (to 2^-60) and independent of the order in which
// runs, chunks and workgroups meet: bitwise deterministic, identical on every data-parallel rank.
// Representable range |x| < 2^20 (flagged otherwise); sums of up to 2^22 entries per row cannot overflow.
//
// HBM traffic per sample at dim E: keys 4 B + (8 B read + 8 B write) per sort pass + dE row 4E, plus 16E
// bytes of accumulator traffic per DISTINCT (table, row) touched.
#include <algorithm>
#include <cstring>

#include "common.h"
#include "radix_sort.h"
#include "split3.h"

#define CHUNK_MAX 32     // sorted entries per reduce walker; 8 / 16 when there are few entries
#define RB_THREADS 256
#ifndef ACC_STRIPES
#define ACC_STRIPES 4    // copies of the dense accumulators of SORTED dense tables: walker w adds into stripe w % ACC_STRIPES, so a hot row of a
                       // table does not serialise its atomics on one address (16 -> 4 stripes: -8 us of zero-fill and finalise per step at config 2)
#endif

#ifndef DIRECT_THREADS
#define DIRECT_THREADS 512
#endif
#define DIRECT_LIST_MAX 3072      // samples per workgroup up to which a single-lookup workgroup lists its samples in LDS (12 KB)
#define DIRECT_RESIDENT 512       // workgroups of the direct sums the chip holds at once (2 per CU by LDS)
#define DIRECT_CAP_ELEMS 4096     // (hi, lo) int64 accumulator pairs per workgroup = 64 KB of LDS, 2 workgroups per CU
#define DIRECT_MAX_PARTS 8        // a table larger than the cap is cut into row ranges, one workgroup column each
#define DIRECT_MAX_PARTS_SMALL 16 // ... up to 16 ranges while the table's slabs stay below DIRECT_SMALL_SLAB_BYTES: a small batch
#define DIRECT_SMALL_SLAB_BYTES (4ll << 20)   // over a 6 040-row table (config 1) then needs no sort at all -- no side branch, no
                                  // fork / join edges: 0.255 -> 0.238 ms per step; at batch 32 768 the two 1 000 x 64 tables of
                                  // configs 5 / 6 would write 56 MB of slabs each and stay on the sorted path
#define DIRECT_MAX_MEMBERS 72
#define DIRECT_MAX_GROUPS 48
#ifndef DIRECT_TARGET
#define DIRECT_TARGET 49152       // (sample, column) elements per workgroup: 3 072 samples of a 16-wide lookup = DIRECT_LIST_MAX, the
                                  // largest chunk that still lists its samples in LDS; fewer chunks = fewer slabs for finalize_kernel to
                                  // add (config 2: 32 -> 22 per table, finalise 36 -> 25 MB), and the direct sums now run beside the
                                  // towers' dW, not alone: 32 768 -> 49 152: 0.3905 -> 0.384 ms per step (65 536: 0.388)
#endif

struct DirectMember {
    int64_t acc_off;      // slab element of (row_lo, column 0) in the member's slab of sample chunk 0 (see `pad`)
    int32_t col;          // first column of the lookup in dE
    int32_t row_lo, rows; // row range accumulated by this member
    int32_t lds_off;      // first LDS accumulator element
    int16_t dim, slot;
    int32_t pad;          // slab stride: elements (vocab x dim) from one sample chunk's slab of this lookup to the next
};
struct DirectGroup {
    int32_t col0, width;  // contiguous column span of dE
    int32_t chunk;        // samples per workgroup
    int32_t block0;       // first workgroup of the group
    int16_t member0, n_members;
    int32_t elems;        // LDS accumulator elements
};
struct DirectMeta {
    DirectMember mem[DIRECT_MAX_MEMBERS];
    DirectGroup grp[DIRECT_MAX_GROUPS];
    int32_t n_groups, n_blocks, n_members, pad;
    int64_t B;
};

// ---- MFMA segment sums (mid-size tables)
#define SEG_WAVES 4               // the waves of a workgroup take the four quarters of its batch range; summed through LDS
// 16-row tiles per job: a job has its class's big or small count (tables are padded to the small one), so the main loop
// has no per-tile conditions.  Accumulators: tiles x column blocks x 4 VGPRs (64 for the big jobs of dim <= 16; wider
// tables keep one size: their split of dE already feeds 2 / 4 column blocks per tile)
static inline int seg_tpw_big(int cls) { return cls == 0 ? 16 : (cls == 1 ? 4 : 2); }
static inline int seg_tpw_small(int cls) { return cls == 2 ? 2 : 4; }
static inline int seg_jobs(int64_t vocab, int cls, int& n_big, int& rows_pad) {
    const int small_rows = 16 * seg_tpw_small(cls), big_rows = 16 * seg_tpw_big(cls);
    rows_pad = static_cast<int>(swr_ceil_div(vocab, small_rows)) * small_rows;
    n_big = rows_pad / big_rows;
    return n_big + (rows_pad - n_big * big_rows) / small_rows;
}
#define SEG_MAX_JOBS 96
struct SegJob {
    int64_t part_off;     // float offset of this lookup's partial tables [split][rows_pad][dim]
    int32_t slot, col;    // key column / first column in dE
    int32_t dim, rows_pad;
    int32_t tile0, tpw;   // first 16-row tile of the job, its number of tiles (the class's big or small count)
};
struct SegMeta {
    SegJob job[SEG_MAX_JOBS];
    int32_t n_jobs, n_splits;
    int64_t rows_per_split;   // multiple of 32 * SEG_WAVES
    int64_t B;
};
struct SegFin {               // per table: where the finalise launch finds the partial tables (n == 0: neither kind)
    int64_t off[MAX_SLOTS];   // n > 0: float offset of the MFMA segment sums' partial tables; n < 0: first slab element of the
    int32_t n[MAX_SLOTS];     // direct sums' -n slabs (one per lookup of the table x sample chunk), `stride` elements apart
    int32_t stride[MAX_SLOTS];   // rows_pad * dim
};
static inline int seg_class(int dim) { return dim <= 16 ? 0 : (dim <= 32 ? 1 : 2); }

struct TableMeta {
    int64_t vocab;
    int64_t acc_off;     // dense: offset (in elements) into the dense accumulator region
    int64_t sorted_off;  // first sorted position of this table's segment
    float* grad_dense;
    int32_t* urow;
    float* ugrad;
    int32_t dim;
    int32_t mode;
};

struct BwdMeta {
    TableMeta tab[MAX_SLOTS];
    int32_t slot_col[MAX_SLOTS];
    int32_t slot_dim[MAX_SLOTS];
    int64_t slot_dst[MAX_SLOTS];     // where this slot's B entries start inside its table segment
    int32_t chunk_off[MAX_SLOTS + 1];   // first reduce chunk of each table
    int16_t sorted_slot[MAX_SLOTS];     // the slots that go through the sort (the others take the direct path)
    int32_t n_sorted_slots;
    int32_t n_slots, n_tables;
    int32_t dim_max;
    int32_t chunk;          // sorted entries per reduce walker
    int64_t B, n;           // n = number of SORTED entries (n_sorted_slots * B)
    int64_t sparse_start;   // first sorted position belonging to a sparse-mode table
};


struct HostPlan {
    BwdMeta m;
    SortMeta sm;
    DirectMeta dm;
    SegMeta sg[3];           // column-block classes: dim <= 16, <= 32, <= 64
    SegFin sf;
    int64_t part_elems;      // floats of segsum partial tables
    size_t off_part;
    int64_t slab_elems;      // (hi, lo) pairs of the direct sums' slabs
    size_t off_slab;
    int64_t dense_acc_elems;
    int64_t sparse_acc_elems;
    size_t off_k0, off_k1, off_v0, off_v1, off_hist, off_acc_hi, off_acc_lo, total;
    int n_passes;
    bool rank_sort;       // the segments are short: counting sort, one launch, result in buffer 1
    int64_t max_keys;
    int n_chunks;
};

// SWR_K3_MFMA=1 opts in (read per call: tests/test_ops_gpu.py compares the paths in one process).  OFF by default: built
// for VERDICT round 2 item 2 and measured at config 2 (tools/micro/k3_probe.py, profiles/r03_k3_probe.txt): the eight
// mid-size tables take 46 us stand-alone against 42 us for the fixed-point direct sums, and 64 - 134 us inside the step,
// where they share VALU and matrix pipes with the weight-gradient product that runs beside them (the direct sums are
// bound by LDS atomics, a resource that product leaves free).  Building a one-hot fragment costs 3 packed 16-bit VALU
// per pair of samples at ~4.4 cycles each (tools/micro/valu_rate.hip) = 53 cycles per fragment, its three MFMAs 48: the
// kernel cannot be matrix-bound, and its work grows with rows x samples where a scatter's grows with samples.
static bool seg_enabled() {
    const char* e = getenv("SWR_K3_MFMA");
    return e && e[0] == '1';
}

// Largest table that takes the MFMA segment sums.  Their work grows with rows x samples (every 16-row tile multiplies
// every sample), the fixed-point direct sums' with samples only; measured at batch 65 536, dim 16 (tools/micro/k3_probe.py)
static int64_t seg_max_rows() {
    const char* e = getenv("SWR_K3_MFMA_MAX_ROWS");
    return e ? atoll(e) : 4096;
}

static int bits_for(uint64_t max_value) {   // bits needed to represent values 0..max_value
    int b = 0;
    while ((max_value >> b) != 0) ++b;
    return b;
}
static size_t align_up(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }

static int make_plan(const swr_embed_grad_slot* slots, int n_slots, int64_t B, HostPlan& p, bool need_outputs = true) {
    SWR_REQUIRE(slots && n_slots > 0 && n_slots <= MAX_SLOTS && B > 0, SWR_ERR_ARG);
    SWR_REQUIRE(B <= (1 << 24), SWR_ERR_UNSUPPORTED);
    BwdMeta& m = p.m;
    m.n_slots = n_slots;
    m.B = B;
    int n_tables = 0;
    int dim_max = 1;
    for (int s = 0; s < n_slots; ++s) {
        SWR_REQUIRE(slots[s].table_id >= 0 && slots[s].table_id < MAX_SLOTS && slots[s].dim > 0 && slots[s].vocab > 0,
                    SWR_ERR_ARG);
        SWR_REQUIRE(slots[s].vocab <= 0xFFFFFFFFll + 1, SWR_ERR_UNSUPPORTED);
        if (slots[s].table_id + 1 > n_tables) n_tables = slots[s].table_id + 1;
    }
    bool seen[MAX_SLOTS] = {false};
    int64_t count[MAX_SLOTS] = {0};
    for (int s = 0; s < n_slots; ++s) {
        const swr_embed_grad_slot& sl = slots[s];
        TableMeta& t = m.tab[sl.table_id];
        if (!seen[sl.table_id]) {
            seen[sl.table_id] = true;
            t.vocab = sl.vocab;
            t.dim = sl.dim;
            t.mode = sl.mode;
            t.grad_dense = sl.grad_dense;
            t.urow = sl.urow;
            t.ugrad = sl.ugrad;
            SWR_REQUIRE(sl.mode >= 0 && sl.mode <= 2, SWR_ERR_ARG);
            if (need_outputs)
                SWR_REQUIRE(sl.mode != 1 ? sl.grad_dense != nullptr : (sl.urow != nullptr && sl.ugrad != nullptr), SWR_ERR_ARG);
        } else {
            SWR_REQUIRE(t.vocab == sl.vocab && t.dim == sl.dim && t.mode == sl.mode, SWR_ERR_ARG);
        }
        m.slot_col[s] = sl.in_col;
        m.slot_dim[s] = sl.dim;
        if (sl.dim > dim_max) dim_max = sl.dim;
    }
    // ---- MFMA segment sums: which tables, the (lookup, tile range) jobs and the batch splits
    bool segsum[MAX_SLOTS] = {false};
    for (int c = 0; c < 3; ++c) { p.sg[c].n_jobs = 0; p.sg[c].n_splits = 0; p.sg[c].rows_per_split = 0; p.sg[c].B = B; }
    for (int t = 0; t < MAX_SLOTS; ++t) { p.sf.off[t] = 0; p.sf.n[t] = 0; p.sf.stride[t] = 0; }
    p.part_elems = 0;
    p.slab_elems = 0;
    if (seg_enabled() && B >= 2048) {        // (the kernel's 32-bit element offsets: B * ld < 2^31 is checked at launch)
        int n_jobs_all = 0, count_cls[3] = {0, 0, 0};
        for (int t = 0; t < n_tables; ++t) {
            if (!seen[t] || m.tab[t].mode == 1) continue;
            const TableMeta& tm = m.tab[t];
            if (tm.vocab <= 16 || tm.vocab > seg_max_rows() || tm.dim % 8 != 0 || tm.dim > 64) continue;
            int uses = 0;
            for (int s = 0; s < n_slots; ++s) uses += slots[s].table_id == t;
            const int cls = seg_class(tm.dim);
            int n_big, rows_pad;
            const int jobs = seg_jobs(tm.vocab, cls, n_big, rows_pad);
            if (count_cls[cls] + uses * jobs > SEG_MAX_JOBS) continue;
            count_cls[cls] += uses * jobs;
            n_jobs_all += uses * jobs;
            segsum[t] = true;
        }
        if (n_jobs_all > 0) {
            const char* env = getenv("SWR_K3_MFMA_WGS");
            const int wgs_target = env ? atoi(env) : 768;              // x 4 waves: three waves per SIMD over the chip
            int64_t want = std::max<int64_t>(1, wgs_target / n_jobs_all);
            want = std::min<int64_t>(want, std::max<int64_t>(1, B / (128 * SEG_WAVES)));
            const int64_t rps = swr_ceil_div(swr_ceil_div(B, want), 32 * SEG_WAVES) * (32 * SEG_WAVES);
            const int n_splits = static_cast<int>(swr_ceil_div(B, rps));
            for (int c = 0; c < 3; ++c) { p.sg[c].n_splits = n_splits; p.sg[c].rows_per_split = rps; }
            int64_t off = 0;
            for (int t = 0; t < n_tables; ++t) {
                if (!segsum[t]) continue;
                const TableMeta& tm = m.tab[t];
                const int cls = seg_class(tm.dim);
                int n_big, rows_pad;
                const int jobs = seg_jobs(tm.vocab, cls, n_big, rows_pad);
                p.sf.off[t] = off;
                p.sf.stride[t] = rows_pad * tm.dim;
                for (int s = 0; s < n_slots; ++s) {
                    if (slots[s].table_id != t) continue;
                    for (int j = 0, tile = 0; j < jobs; ++j) {
                        SegJob& J = p.sg[cls].job[p.sg[cls].n_jobs++];
                        J.part_off = off; J.slot = s; J.col = slots[s].in_col; J.dim = tm.dim; J.rows_pad = rows_pad;
                        J.tile0 = tile; J.tpw = j < n_big ? seg_tpw_big(cls) : seg_tpw_small(cls);
                        tile += J.tpw;
                    }
                    off += static_cast<int64_t>(n_splits) * rows_pad * tm.dim;
                    p.sf.n[t] += n_splits;
                }
            }
            p.part_elems = off;
        }
    }
    // ---- direct path: which tables, and the workgroup layout (groups of adjacent lookup columns x sample chunks)
    bool direct[MAX_SLOTS] = {false};
    DirectMeta& dm = p.dm;
    dm.n_groups = dm.n_blocks = dm.n_members = 0;
    dm.B = B;
    {
        int need_members = 0, need_groups = 0;     // worst case: one group per member
        for (int t = 0; t < n_tables; ++t) {
            if (!seen[t] || segsum[t] || m.tab[t].mode == 1 || m.tab[t].dim > DIRECT_THREADS) continue;
            const int64_t elems = m.tab[t].vocab * m.tab[t].dim;
            if (elems > static_cast<int64_t>(DIRECT_CAP_ELEMS) * DIRECT_MAX_PARTS_SMALL || m.tab[t].dim > DIRECT_CAP_ELEMS) continue;
            int uses = 0;
            for (int s = 0; s < n_slots; ++s) uses += slots[s].table_id == t;
            const int parts = static_cast<int>(swr_ceil_div(elems, DIRECT_CAP_ELEMS));
            if (parts > DIRECT_MAX_PARTS) {
                const int64_t chunks = swr_ceil_div(B, std::max<int64_t>(64, DIRECT_TARGET / m.tab[t].dim));
                if (chunks * uses * elems * 16 > DIRECT_SMALL_SLAB_BYTES) continue;
            }
            if (need_members + uses * parts > DIRECT_MAX_MEMBERS || need_groups + uses * parts > DIRECT_MAX_GROUPS) continue;
            need_members += uses * parts;
            need_groups += uses * parts;
            direct[t] = true;
        }
    }
    m.n_sorted_slots = 0;
    for (int s = 0; s < n_slots; ++s) {
        const int t = slots[s].table_id;
        if (direct[t] || segsum[t]) continue;
        m.sorted_slot[m.n_sorted_slots++] = static_cast<int16_t>(s);
        m.slot_dst[s] = count[t];                    // offset inside the table segment (segment base added below)
        count[t] += B;
    }
    m.n = static_cast<int64_t>(m.n_sorted_slots) * B;
    int64_t acc = 0, pos = 0;
    int tiles = 0, chunks = 0, passes = 0;
    // few sorted entries: smaller sort tiles and reduce chunks, so that the launch still covers the chip and the
    // per-walker dependent-load chain stays short
    {
        int64_t max_keys = 1;
        for (int t = 0; t < n_tables; ++t)
            if (!direct[t] && !segsum[t]) max_keys = std::max<int64_t>(max_keys, count[t]);
        sort_choose_tile(p.sm, m.n, max_keys);
        // short segments (a strong-scaling shard, a short batch): ONE counting launch instead of two per radix pass
        double work = 0;
        for (int t = 0; t < n_tables; ++t)
            if (!direct[t] && !segsum[t]) work += static_cast<double>(count[t]) * static_cast<double>(count[t]);
        bool keys_fit = true;                                          // (the counting sort adds 1 to a key: none may be 0xFFFFFFFF)
        for (int t = 0; t < n_tables; ++t)
            if (!direct[t] && !segsum[t] && count[t] > 0 && m.tab[t].vocab > 0xFFFFFFFFll) keys_fit = false;
        p.rank_sort = rank_sort_enabled() && keys_fit && m.n > 0 && max_keys <= RANK_SORT_MAX_KEYS &&
                      work <= static_cast<double>(RANK_SORT_MAX_WORK);
        p.max_keys = max_keys;
        if (p.rank_sort) { p.sm.items = 1; p.sm.tile = RANK_EPB; p.sm.fold_scan = 0; }
    }
    m.chunk = CHUNK_MAX;
    while (m.chunk > 8 && m.n / m.chunk < 16384) m.chunk >>= 1;
    m.sparse_start = -1;
    for (int t = 0; t < n_tables; ++t) {
        SWR_REQUIRE(seen[t], SWR_ERR_ARG);                       // table ids must be dense 0..n_tables-1
        TableMeta& tm = m.tab[t];
        tm.sorted_off = pos;
        if (tm.mode != 1) {
            SWR_REQUIRE(m.sparse_start < 0, SWR_ERR_ARG);        // sparse tables carry the largest ids
            tm.acc_off = acc;
            if (!direct[t]) acc += tm.vocab * tm.dim;      // (the direct sums keep their accumulators in slabs)
        } else {
            if (m.sparse_start < 0) m.sparse_start = pos;
            tm.acc_off = 0;
        }
        p.sm.seg_off[t] = pos;
        p.sm.tile_off[t] = tiles;
        p.sm.passes[t] = (bits_for(static_cast<uint64_t>(tm.vocab - 1)) + 7) / 8;
        if (count[t] > 0 && p.sm.passes[t] > passes) passes = p.sm.passes[t];
        m.chunk_off[t] = chunks;
        tiles += static_cast<int>(swr_ceil_div(count[t], p.sm.tile));
        chunks += static_cast<int>(swr_ceil_div(count[t], m.chunk));
        pos += count[t];
    }
    for (int s = 0; s < n_slots; ++s)
        if (!direct[slots[s].table_id] && !segsum[slots[s].table_id]) m.slot_dst[s] += m.tab[slots[s].table_id].sorted_off;
    // direct groups (needs the accumulator offsets assigned above)
    {
        int open = -1;                               // group still accepting adjacent small lookups
        // SWR_DIRECT_MERGE=0: every lookup a group of its own (the single-lookup workgroups list their samples first and walk the
        // list with one memory latency per round; a merged group walks its whole chunk with two -- key, then row)
        bool direct_merge = true;
        { const char* e = getenv("SWR_DIRECT_MERGE"); direct_merge = !(e && e[0] == '0'); }
        auto new_group = [&](int col0) {
            DirectGroup& g = dm.grp[dm.n_groups];
            g.col0 = col0; g.width = 0; g.member0 = static_cast<int16_t>(dm.n_members); g.n_members = 0; g.elems = 0;
            return dm.n_groups++;
        };
        auto add_member = [&](int gi, int s, int row_lo, int rows) {
            const TableMeta& tm = m.tab[slots[s].table_id];
            DirectGroup& g = dm.grp[gi];
            DirectMember& mb = dm.mem[dm.n_members++];
            mb.acc_off = tm.acc_off + static_cast<int64_t>(row_lo) * tm.dim;
            mb.col = slots[s].in_col; mb.row_lo = row_lo; mb.rows = rows; mb.lds_off = g.elems;
            mb.dim = static_cast<int16_t>(tm.dim); mb.slot = static_cast<int16_t>(s); mb.pad = 0;
            g.elems += rows * tm.dim;
            g.width += tm.dim;
            g.n_members++;
        };
        for (int s = 0; s < n_slots; ++s) {
            const int t = slots[s].table_id;
            if (!direct[t]) { open = -1; continue; }
            const TableMeta& tm = m.tab[t];
            const int64_t elems = tm.vocab * tm.dim;
            if (elems > DIRECT_CAP_ELEMS) {          // row ranges, one single-member group each
                const int parts = static_cast<int>(swr_ceil_div(elems, DIRECT_CAP_ELEMS));
                const int rpp = static_cast<int>(swr_ceil_div(tm.vocab, parts));
                for (int r0 = 0; r0 < tm.vocab; r0 += rpp)
                    add_member(new_group(slots[s].in_col), s, r0, static_cast<int>(std::min<int64_t>(rpp, tm.vocab - r0)));
                open = -1;
                continue;
            }
            if (open >= 0) {
                const DirectGroup& g = dm.grp[open];
                if (g.col0 + g.width != slots[s].in_col || g.elems + elems > DIRECT_CAP_ELEMS ||
                    g.width + tm.dim > DIRECT_THREADS || !direct_merge)
                    open = -1;
            }
            if (open < 0) open = new_group(slots[s].in_col);
            add_member(open, s, 0, static_cast<int>(tm.vocab));
        }
        // workgroups: DIRECT_TARGET (sample, column) elements each -- unless a launch of slightly more than the chip's 512
        // resident workgroups (64 KB of LDS each, 256 CUs) results: its second round costs a full workgroup time for a few
        // stragglers (config 2: 576 workgroups at 32 768 elements, 504 at 38 400: 38.7 -> 34.2 us), so the target grows by
        // up to half until the launch fits one round
        auto count_blocks = [&](int64_t target) {
            int64_t nb = 0;
            for (int gi = 0; gi < dm.n_groups; ++gi)
                nb += swr_ceil_div(B, std::min<int64_t>(B, std::max<int64_t>(64, target / dm.grp[gi].width)));
            return nb;
        };
        int64_t target = DIRECT_TARGET;
        {
            // short batches: at the full target a strong-scaling shard (8 192 rows, 18 row-range parts) is 54 workgroups walking
            // 3 072 samples each on a 256-CU chip -- the launch is one long latency chain.  Halve the chunks until the grid reaches
            // SWR_DIRECT_MIN_BLOCKS (more, smaller slabs for the finalise launch: a few MB at these sizes), not below 512 samples
            const char* e = getenv("SWR_DIRECT_MIN_BLOCKS");
            const int64_t min_blocks = e ? atoll(e) : 192;
            while (min_blocks > 0 && count_blocks(target) < min_blocks && target / 2 >= 512 * 16) target /= 2;
        }
        if (count_blocks(target) > DIRECT_RESIDENT)
            for (int64_t t2 = target + target / 16; t2 <= target + target / 2; t2 += target / 16)
                if (count_blocks(t2) <= DIRECT_RESIDENT) { target = t2; break; }
        // balance: a single-lookup workgroup walks only the samples whose row falls in ITS row range -- chunk / parts of them.  With one
        // chunk size for all, the one-part workgroups of the small tables (35 .. 200 rows) walk 3 072 samples in six dependent rounds
        // while a sixth of the 1 472-row table walks 512 in one, and the launch lasts as long as the former.  SWR_DIRECT_E > 0: chunk
        // = E x parts (expected matches per workgroup = E), within [512, the common chunk]
        int64_t e_target = 0;
        { const char* e = getenv("SWR_DIRECT_E"); e_target = e ? atoll(e) : 0; }
        int blocks = 0;
        for (int gi = 0; gi < dm.n_groups; ++gi) {
            DirectGroup& g = dm.grp[gi];
            g.chunk = static_cast<int32_t>(std::min<int64_t>(B, std::max<int64_t>(64, target / g.width)));
            if (e_target > 0 && g.n_members == 1) {
                int parts = 0;
                for (int gj = 0; gj < dm.n_groups; ++gj)
                    parts += dm.grp[gj].n_members == 1 && dm.mem[dm.grp[gj].member0].slot == dm.mem[g.member0].slot;
                const int64_t c = std::max<int64_t>(512, e_target * parts);
                if (c < g.chunk) g.chunk = static_cast<int32_t>(c);
            }
            g.block0 = blocks;
            blocks += static_cast<int>(swr_ceil_div(B, g.chunk));
        }
        dm.n_blocks = blocks;
        // slabs: every workgroup STORES its accumulators (all of them) into the slab of (lookup, sample chunk) -- plain 16-byte
        // stores instead of two memory-side atomics per element into shared stripes (4 M atomics per step at config 2: they,
        // not the LDS atomics or the reads, set the kernel's time); the finalise launch adds a table's slabs
        int64_t slot_first[MAX_SLOTS], n_slabs[MAX_SLOTS];
        for (int s = 0; s < MAX_SLOTS; ++s) { slot_first[s] = -1; n_slabs[s] = 0; }
        for (int gi = 0; gi < dm.n_groups; ++gi) {
            const DirectGroup& g = dm.grp[gi];
            const int64_t chunks = swr_ceil_div(B, g.chunk);
            for (int q = 0; q < g.n_members; ++q) {
                const int s = dm.mem[g.member0 + q].slot, t = slots[s].table_id;
                if (slot_first[s] < 0) { slot_first[s] = n_slabs[t]; n_slabs[t] += chunks; }   // (row-range parts share the slabs)
            }
        }
        int64_t slab = 0;
        for (int t = 0; t < n_tables; ++t) {
            if (!direct[t] || n_slabs[t] == 0) continue;
            const int64_t te = m.tab[t].vocab * m.tab[t].dim;
            SWR_REQUIRE(n_slabs[t] < (1ll << 30), SWR_ERR_ARG);
            p.sf.off[t] = slab;
            p.sf.n[t] = -static_cast<int32_t>(n_slabs[t]);
            p.sf.stride[t] = static_cast<int32_t>(te);
            slab += n_slabs[t] * te;
        }
        p.slab_elems = slab;
        for (int q = 0; q < dm.n_members; ++q) {
            DirectMember& mb = dm.mem[q];
            const int t = slots[mb.slot].table_id;
            const int64_t te = m.tab[t].vocab * m.tab[t].dim;
            mb.acc_off = p.sf.off[t] + slot_first[mb.slot] * te + static_cast<int64_t>(mb.row_lo) * mb.dim;
            mb.pad = static_cast<int32_t>(te);
        }
    }
    p.sm.seg_off[n_tables] = pos;
    p.sm.tile_off[n_tables] = tiles;
    p.sm.n_tables = n_tables;
    p.sm.n_tiles = tiles;
    m.chunk_off[n_tables] = chunks;
    if (m.sparse_start < 0) m.sparse_start = m.n;
    m.n_tables = n_tables;
    m.dim_max = dim_max;
    p.n_passes = passes;
    p.n_chunks = chunks;
    p.dense_acc_elems = acc;
    p.sparse_acc_elems = (m.n - m.sparse_start) * dim_max;

    size_t off = 0;
    const size_t kb = align_up(static_cast<size_t>(m.n) * 4);
    p.off_k0 = off; off += kb;
    p.off_k1 = off; off += kb;
    p.off_v0 = off; off += kb;
    p.off_v1 = off; off += kb;
    p.off_hist = off; off += align_up(static_cast<size_t>(tiles) * 256 * 4);
    const size_t ab = align_up(static_cast<size_t>(p.dense_acc_elems * ACC_STRIPES + p.sparse_acc_elems) * 8);
    p.off_acc_hi = off; off += ab;
    p.off_acc_lo = off; off += ab;
    p.off_part = off; off += align_up(static_cast<size_t>(p.part_elems) * 4);
    p.off_slab = off; off += align_up(static_cast<size_t>(p.slab_elems) * 16);
    p.total = off;
    return SWR_OK;
}

// (+ the zero-fill of both accumulator limbs, `zero16` 16-byte words from `zero`: it was a launch of its own in front of this one)
__global__ __launch_bounds__(RB_THREADS) void build_keys_kernel(const BwdMeta m, const uint32_t* __restrict__ keys,
                                                                uint32_t* __restrict__ ck, uint32_t* __restrict__ val,
                                                                uint4* __restrict__ zero, int64_t zero16) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * RB_THREADS + threadIdx.x;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * RB_THREADS;
    for (int64_t z = i; z < zero16; z += stride) zero[z] = make_uint4(0u, 0u, 0u, 0u);
    if (i >= m.n) return;
    const int slot = m.sorted_slot[i / m.B];
    const int64_t b = i % m.B;
    const int64_t dst = m.slot_dst[slot] + b;
    ck[dst] = keys[static_cast<int64_t>(slot) * m.B + b];
    val[dst] = (static_cast<uint32_t>(slot) << 24) | static_cast<uint32_t>(b);
}

// ------------------------------------------------------------------------------------------ reduce
// floor(x * 2^60) as two limbs: x * 2^60 ~ hi * 2^40 + lo, 0 <= lo < 2^40 for normal-sized x (tiny x: lo in {-1, 0, ..}).
// Integer-only and BRANCH-FREE (selects on 32-bit halves): the fp64 version made the reduction VALU-bound, a branchy
// one spends more on exec-mask bookkeeping than on arithmetic.  Signed mantissa sm and exponent e of the fp32 word,
// value = sm << (e - 90).  |x| >= 2^20, Inf and NaN contribute nothing and set `bad` (reported once per thread).
__device__ __forceinline__ void to_fixed_general(float x, long long& hi, long long& lo, uint32_t& bad) {
    const uint32_t u = __float_as_uint(x);
    const int e = static_cast<int>((u >> 23) & 0xFFu);
    bad |= e >= 147 ? 1u : 0u;
    const int m = static_cast<uint32_t>(e - 1) < 146u ? static_cast<int>((u & 0x7FFFFFu) | 0x800000u) : 0;   // 0, subnormal, bad -> 0
    const int sm = static_cast<int>(u) < 0 ? -m : m;
    const int sh = e - 90;                                   // A: sh >= 40, B: 0 <= sh < 40, C: sh < 0
    const bool A = sh >= 40, C = sh < 0;
    const int left = A ? sh - 40 : (C ? 0 : sh);
    const long long t = static_cast<long long>(sm) << left; // |t| < 2^63
    const int tl = static_cast<int>(t), th = static_cast<int>(t >> 32);
    const int c = sm >> min(max(-sh, 0), 31);                // C: floor(sm / 2^-sh)
    const int hi_lo = A ? tl : (C ? 0 : th >> 8);
    const int hi_hi = A ? th : (C ? 0 : th >> 31);
    const int lo_lo = A ? 0 : (C ? c : tl);
    const int lo_hi = A ? 0 : (C ? c >> 31 : (th & 0xFF));
    hi = static_cast<long long>((static_cast<unsigned long long>(static_cast<uint32_t>(hi_hi)) << 32) | static_cast<uint32_t>(hi_lo));
    lo = static_cast<long long>((static_cast<unsigned long long>(static_cast<uint32_t>(lo_hi)) << 32) | static_cast<uint32_t>(lo_lo));
}

// Case B only (2^-37 <= |x| < 8, or x == 0): what gradients are in practice.  Half the instructions of the general form.
__device__ __forceinline__ void to_fixed_mid(float x, long long& hi, long long& lo) {
    const uint32_t u = __float_as_uint(x);
    const uint32_t e = (u >> 23) & 0xFFu;
    const int m = e ? static_cast<int>((u & 0x7FFFFFu) | 0x800000u) : 0;
    const int s = static_cast<int>(u) >> 31;
    const int sm = (m ^ s) - s;
    const long long t = static_cast<long long>(sm) << ((e - 90u) & 63u);   // e == 0: sm == 0, any shift will do
    const int tl = static_cast<int>(t), th = static_cast<int>(t >> 32);
    hi = static_cast<long long>(th >> 8);
    lo = static_cast<long long>((static_cast<unsigned long long>(static_cast<uint32_t>(th & 0xFF)) << 32) | static_cast<uint32_t>(tl));
}

// N values at once: the wave takes the short form when every value of every lane is in case B (wave-uniform branch)
template <int N>
__device__ __forceinline__ void to_fixed_n(const float (&x)[N], long long (&hi)[N], long long (&lo)[N], uint32_t& bad) {
    bool mid = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t e = (__float_as_uint(x[i]) >> 23) & 0xFFu;
        mid = mid && (e == 0u || (e - 90u) < 40u);
    }
    if (__all(mid)) {
#pragma unroll
        for (int i = 0; i < N; ++i) to_fixed_mid(x[i], hi[i], lo[i]);
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) to_fixed_general(x[i], hi[i], lo[i], bad);
    }
}

// The direct sums' form: when every value of the wave is zero or 2^-37 <= |x| < 2^-13 (gradients of a mean loss over a large
// batch are), the whole product sm << (e - 90) -- below 2^47 -- goes into the LOW limb and the high limb is not touched: one
// LDS atomic per element instead of two.  A workgroup adds at most 32 768 of them per accumulator (< 2^62), and its flush
// carries bits 40 and up into the high limb, so the totals that reach memory are the pairs the two-limb form produces.
// Returns true when the wave took the one-limb form (wave-uniform).
template <int N>
__device__ __forceinline__ bool to_fixed_wide_n(const float (&x)[N], long long (&hi)[N], long long (&lo)[N], uint32_t& bad) {
    bool small = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t e = (__float_as_uint(x[i]) >> 23) & 0xFFu;
        small = small && (e == 0u || (e - 90u) < 24u);
    }
    if (__all(small)) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const uint32_t u = __float_as_uint(x[i]);
            const uint32_t e = (u >> 23) & 0xFFu;
            const int m = e ? static_cast<int>((u & 0x7FFFFFu) | 0x800000u) : 0;
            const int s = static_cast<int>(u) >> 31;
            const int sm = (m ^ s) - s;
            lo[i] = static_cast<long long>(sm) << ((e - 90u) & 63u);
            hi[i] = 0;
        }
        return true;
    }
    to_fixed_n<N>(x, hi, lo, bad);
    return false;
}

__device__ __forceinline__ float from_fixed(long long hi, long long lo) {
    return static_cast<float>(static_cast<double>(hi) * (1.0 / 1048576.0) +
                              static_cast<double>(lo) * (1.0 / 1152921504606846976.0));
}

// ------------------------------------------------------------------------------------------ direct
// Small tables: no sort.  Thread = V adjacent gradient columns of one lookup of the group's span (V = 4: one 16-byte
// load + one key load feed 4 accumulator elements); the DIRECT_THREADS / (width / V) sample lanes of a workgroup walk
// the chunk with U row loads in flight each.  Every addend is the same fixed-point pair the sorted path would add,
// and integer addition commutes, so the result is bit-identical to the sorted path whatever the order of the
// LDS / memory-side atomics.
template <int V>
__device__ __forceinline__ void direct_body(const DirectMeta& dm, const uint32_t* __restrict__ keys, const float* __restrict__ dE, int64_t ld,
                                            longlong2* __restrict__ slab, uint32_t* err, const int block) {
    extern __shared__ unsigned long long lacc[];              // [elems] hi limbs, then [elems] lo limbs
#ifdef DIRECT_EMPTY            // (ablation builds of tools/build_variant.py only)
    if (dm.B > 0) return;
#endif
    int gi = 0;
    while (gi + 1 < dm.n_groups && dm.grp[gi + 1].block0 <= block) ++gi;
    const DirectGroup& G = dm.grp[gi];
    const int chunk_id = block - G.block0;
    const int64_t b0 = static_cast<int64_t>(chunk_id) * G.chunk;
    const int64_t b1 = min(b0 + G.chunk, dm.B);
    const int elems = G.elems;
    const int tid = threadIdx.x;
    for (int j = tid; j < 2 * elems + 1; j += DIRECT_THREADS) lacc[j] = 0ull;    // (+ the sample list's counter word)
    __syncthreads();
#ifdef DIRECT_ZERO_ONLY
    if (dm.B > 0) return;
#endif

    const int W = G.width / V;                                // threads per sample row
    const int spl = DIRECT_THREADS / W;                       // sample lanes
#ifndef DIRECT_U
#define DIRECT_U 4
#endif
    constexpr int U = DIRECT_U;                               // row loads in flight per thread
    uint32_t bad = 0u;
    // one quad (V = 4) / one element of a gradient row -> the LDS accumulators at `a`; a lane's four elements are taken in an
    // order rotated by its sample lane: rows start at multiples of 128 bytes (dim 16), so with every lane on element v of its
    // quad the 16 lanes of an LDS group hit 4 bank pairs 4 deep
    auto add_quad = [&](const float (&xq)[V], int a, int rot) {
        long long h[V], l[V];
        float xr[V];
        if (V == 4) {
            const bool r1 = rot & 1, r2 = rot & 2;
            const float y0 = r1 ? xq[1 % V] : xq[0], y1 = r1 ? xq[2 % V] : xq[1 % V], y2 = r1 ? xq[3 % V] : xq[2 % V],
                        y3 = r1 ? xq[0] : xq[3 % V];
            xr[0] = r2 ? y2 : y0; xr[1 % V] = r2 ? y3 : y1; xr[2 % V] = r2 ? y0 : y2; xr[3 % V] = r2 ? y1 : y3;
        } else {
            xr[0] = xq[0];
        }
        const bool wide = to_fixed_wide_n<V>(xr, h, l, bad);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int av = a + ((v + rot) & (V - 1));
            if (!wide) atomicAdd(&lacc[av], static_cast<unsigned long long>(h[v]));
            atomicAdd(&lacc[elems + av], static_cast<unsigned long long>(l[v]));
        }
    };
    if (V == 4 && G.n_members == 1 && G.chunk <= DIRECT_LIST_MAX) {
        // ---- one lookup per workgroup (every mid-size table, every row range of a cut table): the chunk's keys are read in
        // ONE round of independent loads and the samples whose row falls in this workgroup's range are listed in LDS
        // ((sample << 16) | row; list order is arbitrary, integer sums do not care).  The walk over the list then has one
        // memory latency per round (the gradient row) instead of two (key, then row), and a row range of a table cut 6 ways
        // walks a sixth of the chunk instead of idling through all of it.
        const DirectMember& M = dm.mem[G.member0];
        uint32_t* list = reinterpret_cast<uint32_t*>(lacc + 2 * elems);      // [0] = count (zeroed above), entries from [1]
        const uint32_t* __restrict__ kp = keys + static_cast<int64_t>(M.slot) * dm.B + b0;
        const int ns = static_cast<int>(b1 - b0);
        // (one counter increment per WAVE: the lanes whose sample belongs here take consecutive places behind the wave's base.  With
        // one returning LDS atomic per lane on the single counter word, the 3 072 increments of a workgroup ran one after the other --
        // ~15 of the launch's 25 us, found by building the kernel without its walk and without its flush: 27.5 of 35 us remained)
        // (ALL of the chunk's keys requested before the first is looked at: the rolled loop made one memory round trip per 512 keys,
        // six in a row for a 3 072-sample chunk -- 14 of the launch's 21 us)
        constexpr int KR = (DIRECT_LIST_MAX + DIRECT_THREADS - 1) / DIRECT_THREADS;
        uint32_t kr[KR];
#pragma unroll
        for (int u = 0; u < KR; ++u) {
            const int s = u * DIRECT_THREADS + tid;
            kr[u] = s < ns ? kp[s] : 0xFFFFFFFFu;
        }
#ifdef DIRECT_KEYS_ONLY        // (ablation builds only)
        if (kr[0] + kr[KR - 1] != 0x12345u) return;
#endif
#pragma unroll
        for (int u = 0; u < KR; ++u) {
            const int s0 = u * DIRECT_THREADS;
            if (s0 >= ns) break;                                  // (workgroup-uniform)
            const int s = s0 + tid;
            const uint32_t r = s < ns ? kr[u] - static_cast<uint32_t>(M.row_lo) : 0xFFFFFFFFu;
            const bool mine = r < static_cast<uint32_t>(M.rows);
            const unsigned long long mask = __ballot(mine);
            uint32_t base = 0u;
            if (mask != 0ull) {
                const int leader = __ffsll(static_cast<long long>(mask)) - 1;
                if ((tid & 63) == leader) base = atomicAdd(&list[0], static_cast<uint32_t>(__popcll(mask)));
                base = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(base), leader));
            }
            if (mine) list[1 + base + __popcll(mask & ((1ull << (tid & 63)) - 1ull))] = (static_cast<uint32_t>(s) << 16) | r;
        }
        __syncthreads();
#ifdef DIRECT_LIST_ONLY
        if (list[0] != 0x7FFFFFFFu) return;
#endif
#ifdef DIRECT_NO_WALK          // (ablation builds of tools/build_variant.py only)
        const int n_mine = 0;
#else
        const int n_mine = static_cast<int>(list[0]);
#endif
        if (tid < spl * W) {
            const int c = (tid % W) * V, sl = tid / W;
            const int rot = sl & 3;
            const int dim = M.dim, base = M.lds_off + c;
            const float* __restrict__ xp = dE + G.col0 + c + b0 * ld;
            for (int e0 = sl; e0 < n_mine; e0 += spl * U) {
                float x[U][V];
                uint32_t ent[U];
#pragma unroll
                for (int u = 0; u < U; ++u) ent[u] = list[1 + min(e0 + u * spl, n_mine - 1)];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float4 v = *reinterpret_cast<const float4*>(xp + static_cast<int64_t>(ent[u] >> 16) * ld);
                    x[u][0] = v.x; x[u][1 % V] = v.y; x[u][2 % V] = v.z; x[u][3 % V] = v.w;
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (e0 + u * spl < n_mine) add_quad(x[u], base + static_cast<int>(ent[u] & 0xFFFFu) * dim, rot);
            }
        }
    } else if (tid < spl * W) {
        const int c = (tid % W) * V, sl = tid / W;
        const int rot = V == 4 ? (sl & 3) : 0;
        int mi = G.member0, cc = c;
        while (cc >= dm.mem[mi].dim) { cc -= dm.mem[mi].dim; ++mi; }
        const int dim = dm.mem[mi].dim, row_lo = dm.mem[mi].row_lo, rows = dm.mem[mi].rows;
        const int base = dm.mem[mi].lds_off + cc;
        const uint32_t* __restrict__ kp = keys + static_cast<int64_t>(dm.mem[mi].slot) * dm.B;
        const float* __restrict__ xp = dE + G.col0 + c;
        for (int64_t s0 = b0 + sl; s0 < b1; s0 += static_cast<int64_t>(spl) * U) {
            float x[U][V];
            uint32_t k[U];
            bool mine[U];
            // keys first: only the samples whose row falls in THIS workgroup's range fetch their gradient row
#pragma unroll
            for (int u = 0; u < U; ++u) k[u] = kp[min(s0 + static_cast<int64_t>(u) * spl, b1 - 1)];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t sidx = s0 + static_cast<int64_t>(u) * spl;
                mine[u] = sidx < b1 && (k[u] - static_cast<uint32_t>(row_lo)) < static_cast<uint32_t>(rows);
                if (V == 4) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (mine[u]) v = *reinterpret_cast<const float4*>(xp + sidx * ld);
                    x[u][0] = v.x; x[u][1 % V] = v.y; x[u][2 % V] = v.z; x[u][3 % V] = v.w;
                } else {
                    x[u][0] = mine[u] ? xp[sidx * ld] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (mine[u]) add_quad(x[u], base + static_cast<int>(k[u] - static_cast<uint32_t>(row_lo)) * dim, rot);
        }
    }
    if (bad && err) atomicOr(err, SWR_FLAG_GRAD_RANGE);
    __syncthreads();
    // every accumulator (touched or not) goes to this workgroup's part of the slab of (lookup, sample chunk): one coalesced
    // 16-byte store per element, no zero-fill needed in front, nothing shared with another workgroup
#ifdef DIRECT_NO_FLUSH
    if (lacc[0] == 0x123456789ull)
#endif
    for (int q = 0; q < G.n_members; ++q) {
        const DirectMember& M = dm.mem[G.member0 + q];
        const int n = M.rows * M.dim;
        longlong2* __restrict__ dst = slab + M.acc_off + static_cast<int64_t>(chunk_id) * M.pad;
        for (int j = tid; j < n; j += DIRECT_THREADS) {
            long long h = static_cast<long long>(lacc[M.lds_off + j]), l = static_cast<long long>(lacc[elems + M.lds_off + j]);
            h += l >> 40;                                        // carry of the one-limb form (floor, like the two-limb split)
            l &= (1ll << 40) - 1;
            dst[j] = make_longlong2(h, l);
        }
    }
}
template <int V>
__global__ __launch_bounds__(DIRECT_THREADS) void direct_kernel(const DirectMeta dm, const uint32_t* __restrict__ keys,
                                                                const float* __restrict__ dE, int64_t ld,
                                                                longlong2* __restrict__ slab, uint32_t* err) {
    direct_body<V>(dm, keys, dE, ld, slab, err, static_cast<int>(blockIdx.x));
}

// Short batches: the counting sort of the large tables' keys (radix_sort.h) and the direct sums of the small tables are independent
// and each leaves most of the chip idle (a few hundred workgroups of one latency chain each): ONE launch runs both -- workgroups
// [0, n_direct) the direct sums, the rest the sort with 512-entry rows -- so the two chains overlap instead of following each other
// on the step's single stream (a second stream would make the replayed graph host-bound, ops.SIDE_MIN_BATCH).  Same bodies: same bits.
template <int ROWS>
__global__ __launch_bounds__(DIRECT_THREADS) void rank_direct_kernel(const DirectMeta dm, const uint32_t* __restrict__ keys,
                                                                     const float* __restrict__ dE, int64_t ld, longlong2* __restrict__ slab,
                                                                     uint32_t* err, int n_direct, const SortMeta sm, const RankSrc src,
                                                                     uint32_t* __restrict__ kout, uint32_t* __restrict__ vout) {
    if (static_cast<int>(blockIdx.x) < n_direct) direct_body<4>(dm, keys, dE, ld, slab, err, static_cast<int>(blockIdx.x));
    else rank_sort_body<ROWS, DIRECT_THREADS>(sm, src, kout, vout, static_cast<int>(blockIdx.x) - n_direct, static_cast<int>(gridDim.x) - n_direct);
}

// ------------------------------------------------------------------------------------------ segsum (MFMA)
// dEmb_t[v, :] = sum over the samples b with key_t(b) == v of dE[b, cols_t]  =  OneHot_t^T dE_t, a product whose reduction
// index is the batch.  A wave owns up to TPW 16-row tiles of one table and the batch range of its split; per 32 samples:
//   * A fragment (16 table rows x 32 samples) of tile r0: lane (i, kq) holds (key[b0 + 8 kq + e] == r0 + i) for e = 0..7,
//     built from the 8 keys of the lane's sample octet -- nothing is read but the keys;
//   * B fragment (32 samples x 16 columns): lane (j, kq) holds dE[b0 + 8 kq + e][col + j], split exactly into three bf16
//     terms; three MFMAs (low term first) accumulate in fp32.
// |dE| >= 2^20, Inf and NaN contribute nothing and raise the sticky error word, as on the fixed-point paths (and a
// non-finite value would otherwise poison the other rows of its tile through 0 * Inf).
// The order of additions is fixed by the code: results are bitwise reproducible and identical on every data-parallel rank.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// one-hot fragment of the tile whose first row is `c` rows behind the lane's reference row: dd[q] holds two 16-bit
// differences (key - reference row) of consecutive samples; a half that equals c becomes bf16 1.0 (0x3F80), any other 0.
// Three packed 16-bit VALU per PAIR of samples, no scalar mask registers:  x = dd - c;  ne = min(x, 1);  w = 0x3F80 - 0x3F80 ne.
// (Inline asm: written as vector arithmetic, hipcc turns the min into compares + selects + byte permutes, 2.5x the work.
// Packed 16-bit operations issue every ~4.4 cycles per SIMD (tools/micro/valu_rate.hip), a 16x16x32 MFMA every 16: the
// four pairs of a fragment cost about as much VALU time as its three MFMAs cost matrix time, so the two are interleaved.)
__device__ __forceinline__ uint32_t seg_onehot2(uint32_t ddq, uint32_t cc, uint32_t one, uint32_t negv, uint32_t posv) {
    uint32_t x;
    asm volatile("v_pk_sub_u16 %0, %1, %2\n\t"
                 "v_pk_min_u16 %0, %0, %3\n\t"
                 "v_pk_mad_u16 %0, %0, %4, %5"
                 : "=&v"(x) : "v"(ddq), "s"(cc), "v"(one), "v"(negv), "v"(posv));
    return x;
}
__device__ __forceinline__ bf16x8 seg_onehot8(const u32x4 dd, uint32_t cc, uint32_t one, uint32_t negv, uint32_t posv) {
    u32x4 w;
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = seg_onehot2(dd[q], cc, one, negv, posv);
    return __builtin_bit_cast(bf16x8, w);
}

template <int CB, int TPW>
__device__ __forceinline__ void seg_body(const SegMeta& sm, const SegJob& J, int split, const uint32_t* __restrict__ keys,
                                         const float* __restrict__ dE, int64_t ld, float* __restrict__ part, uint32_t* err,
                                         float* seg_red) {                // seg_red: [SEG_WAVES - 1][TPW * CB * 4][64]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const int i = lane & 15, kq = lane >> 4;
    // this wave's quarter of the split's batch range, in 32-sample steps
    const int64_t s_begin = static_cast<int64_t>(split) * sm.rows_per_split;
    const int64_t quarter = sm.rows_per_split / SEG_WAVES;             // multiple of 32
    const int64_t b_begin = min(s_begin + wave * quarter, sm.B);
    const int64_t b_end = min(b_begin + quarter, sm.B);
    const int n_full = static_cast<int>((b_end - b_begin) / 32);
    const int rows_tail = static_cast<int>(b_end - b_begin) - 32 * n_full;      // valid rows of the ragged last step (0: none)
    const uint32_t ld32 = static_cast<uint32_t>(ld);
    const uint32_t row_ref = static_cast<uint32_t>(16 * J.tile0 + i);
    // wave-uniform bases (scalar registers) + lane-constant 32-bit offsets: the loads of a step are
    // `global_load_dword v, v_lane_offset, s[base]` with the step / row advance folded into the scalar base
    const uint32_t* __restrict__ kp = keys + static_cast<int64_t>(J.slot) * sm.B + b_begin;
    const float* __restrict__ xp = dE + b_begin * ld + J.col;
    const uint32_t koff = static_cast<uint32_t>(8 * kq);
    bool colv[CB];
    uint32_t coff[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        colv[cb] = 16 * cb + i < J.dim;
        coff[cb] = static_cast<uint32_t>(8 * kq) * ld32 + (colv[cb] ? 16 * cb + i : 0);
    }

    f32x4 acc[TPW][CB];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[t][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint32_t amax = 0u;                                            // largest |x| bit pattern seen (range / NaN / Inf check)
    uint32_t k_one = 0x00010001u, k_neg = 0xC080C080u, k_pos = 0x3F803F80u;      // packed constants of seg_onehot8, in VGPRs
    asm volatile("" : "+v"(k_one), "+v"(k_neg), "+v"(k_pos));

    // raw loads of a full 32-sample step
    auto load_step = [&](int step, uint32_t (&k)[8], float (&x)[CB][8]) {
        const uint32_t* kps = kp + 32 * step;                           // (scalar)
        const float* xps = xp + static_cast<int64_t>(32 * step) * ld;    // (scalar)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            k[e] = (kps + e)[koff];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) x[cb][e] = (xps + static_cast<int64_t>(e) * ld)[coff[cb]];
        }
    };
    // d: (key - reference row) mod 2^16 per sample, two per dword; v: the gradient values (non-finite / out-of-range -> 0)
    auto mma_step = [&](const u32x4 dd, const float (&v)[CB][8]) {
        bf16x8 bh[CB], bm[CB], bl[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int e = 0; e < 8; e += 2) SPLIT3_PAIR(v[cb][e], v[cb][e + 1], bh[cb], bm[cb], bl[cb], e);
        // software pipeline inside the wave: the pairs of the NEXT tile's one-hot fragment are generated between the three
        // (dependent) MFMAs of the current tile, so the matrix pipe works while the VALU builds its next operand
        u32x4 wn;
#pragma unroll
        for (int q = 0; q < 4; ++q) wn[q] = seg_onehot2(dd[q], 0u, k_one, k_neg, k_pos);
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const bf16x8 a = __builtin_bit_cast(bf16x8, wn);
            const uint32_t cn = static_cast<uint32_t>(16 * (t + 1)) * 0x00010001u;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                f32x4 c_ = acc[t][cb];
                c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bl[cb], c_, 0, 0, 0);
                if (cb == 0 && t + 1 < TPW) { __builtin_amdgcn_sched_barrier(0); wn[0] = seg_onehot2(dd[0], cn, k_one, k_neg, k_pos); __builtin_amdgcn_sched_barrier(0); }
                c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bm[cb], c_, 0, 0, 0);
                if (cb == 0 && t + 1 < TPW) {
                    __builtin_amdgcn_sched_barrier(0);
                    wn[1] = seg_onehot2(dd[1], cn, k_one, k_neg, k_pos);
                    wn[2] = seg_onehot2(dd[2], cn, k_one, k_neg, k_pos);
                    __builtin_amdgcn_sched_barrier(0);
                }
                c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bh[cb], c_, 0, 0, 0);
                if (cb == 0 && t + 1 < TPW) { __builtin_amdgcn_sched_barrier(0); wn[3] = seg_onehot2(dd[3], cn, k_one, k_neg, k_pos); __builtin_amdgcn_sched_barrier(0); }
                acc[t][cb] = c_;
            }
        }
    };
    // |x| >= 2^20, Inf, NaN (exponent field >= 147) contribute nothing; the largest magnitude pattern is kept for the flag
    auto clean = [&](float xv, bool live) -> float {
        const uint32_t ab = __float_as_uint(xv) & 0x7FFFFFFFu;
        amax = max(amax, live ? ab : 0u);
        return (ab < 0x49800000u && live) ? xv : 0.f;
    };
    auto full_step = [&](const uint32_t (&k)[8], const float (&x)[CB][8]) {
        u32x4 dd;
#pragma unroll
        for (int q = 0; q < 4; ++q) dd[q] = ((k[2 * q] - row_ref) & 0xFFFFu) | ((k[2 * q + 1] - row_ref) << 16);
        float v[CB][8];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[cb][e] = clean(x[cb][e], colv[cb]);
        mma_step(dd, v);
    };
    if (n_full > 0) {
        uint32_t kr[2][8];
        float xr[2][CB][8];
        load_step(0, kr[0], xr[0]);
        int st = 0;
        for (; st + 2 <= n_full; st += 2) {
            load_step(st + 1, kr[1], xr[1]);
            full_step(kr[0], xr[0]);
            load_step(min(st + 2, n_full - 1), kr[0], xr[0]);          // (the last pair re-loads a valid step; unused)
            full_step(kr[1], xr[1]);
        }
        if (st < n_full) full_step(kr[0], xr[0]);
    }
    if (rows_tail > 0) {                                               // ragged last step of the batch: rows masked one by one
        const uint32_t* kps = kp + 32 * n_full;
        const float* xps = xp + static_cast<int64_t>(32 * n_full) * ld;
        u32x4 dd;
        float v[CB][8];
        uint32_t d[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool live = 8 * kq + e < rows_tail;
            d[e] = live ? ((kps + e)[koff] - row_ref) & 0xFFFFu : 0x8000u;      // 0x8000: the row of no tile
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
                v[cb][e] = clean((live && colv[cb]) ? (xps + static_cast<int64_t>(e) * ld)[coff[cb]] : 0.f, true);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) dd[q] = d[2 * q] | (d[2 * q + 1] << 16);
        mma_step(dd, v);
    }
    if (amax >= 0x49800000u && err) atomicOr(err, SWR_FLAG_GRAD_RANGE);

    // ---- the four quarters of the split, summed in wave order through LDS: ((w0 + w1) + w2) + w3
    constexpr int REGS = TPW * CB * 4;
    if (wave > 0) {
        float* dst = seg_red + static_cast<size_t>(wave - 1) * REGS * 64;
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[((t * CB + cb) * 4 + r) * 64 + lane] = acc[t][cb][r];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 1; w < SEG_WAVES; ++w) {
        const float* src = seg_red + static_cast<size_t>(w - 1) * REGS * 64;
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][cb][r] += src[((t * CB + cb) * 4 + r) * 64 + lane];
    }
    // partial table of this split: lane holds rows 16 t + 4 kq + r, column 16 cb + i
    float* __restrict__ P = part + J.part_off + static_cast<int64_t>(split) * J.rows_pad * J.dim;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            if (colv[cb]) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    P[static_cast<int64_t>(16 * (J.tile0 + t) + 4 * kq + r) * J.dim + 16 * cb + i] = acc[t][cb][r];
            }
        }
    }
}

// jobs come in two sizes per column-block class (SegJob.tpw): BIG tiles per wave where a table has that many left -- the
// per-step work that does not depend on the tile count (key arithmetic, the split of dE: ~130 instructions against 12 per
// tile) is then spread over 16 tiles instead of 4 -- and SMALL for the remainders and the small tables
template <int CB, int BIG, int SMALL>
__global__ __launch_bounds__(SEG_WAVES * 64) void segsum_mfma_kernel(const SegMeta sm, const uint32_t* __restrict__ keys,
                                                                      const float* __restrict__ dE, int64_t ld,
                                                                      float* __restrict__ part, uint32_t* err) {
    extern __shared__ __attribute__((aligned(16))) float seg_red[];
    const int jb = static_cast<int>(blockIdx.x) % sm.n_jobs, split = static_cast<int>(blockIdx.x) / sm.n_jobs;
    const SegJob& J = sm.job[jb];
    if (J.tpw == BIG) seg_body<CB, BIG>(sm, J, split, keys, dE, ld, part, err, seg_red);
    else seg_body<CB, SMALL>(sm, J, split, keys, dE, ld, part, err, seg_red);
}

// first position in [lo, hi) whose key is >= key
__device__ __forceinline__ int64_t lower_bound_key(const uint32_t* ck, int64_t lo, int64_t hi, uint32_t key) {
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (ck[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// LPE lanes per entry (one lane per gradient column; dims above 64 are walked in 64-column blocks).
// A chunk never crosses a table segment.  Runs that lie inside one chunk are the only contributors to their row and
// are stored plainly.  Runs cut by chunk boundaries are first joined ACROSS THE WALKERS OF THE WORKGROUP in LDS (walker
// q's open tail + the "through" chunks after it + the open head that ends the run), so a hot row whose run spans
// thousands of entries costs one atomic pair per workgroup instead of one per chunk (same-address atomics serialise
// at ~0.2 us each on the memory side); only runs that cross a workgroup boundary use atomics.
template <int LPE>
__global__ __launch_bounds__(RB_THREADS) void reduce_kernel(const BwdMeta m, int n_chunks, const uint32_t* __restrict__ ck,
                                                            const uint32_t* __restrict__ val,
                                                            const float* __restrict__ dE, int64_t ld,
                                                            unsigned long long* acc_hi, unsigned long long* acc_lo,
                                                            int64_t dense_acc_elems, uint32_t* err) {
    constexpr int NW = RB_THREADS / LPE;
    __shared__ long long sh_sum[4][RB_THREADS];     // open head (hi, lo), open tail (hi, lo) per walker lane
    __shared__ long long w_head[2][NW];             // sorted position where the open head / tail run starts
    __shared__ uint32_t w_key[2][NW];
    __shared__ int w_info[NW];                      // bit 0 head open, bit 1 tail open, bit 2 through; table << 8
    __shared__ int s_col[MAX_SLOTS];                // m.slot_col: indexed per lane below -- from the kernel arguments that is a
    if (threadIdx.x < MAX_SLOTS) s_col[threadIdx.x] = m.slot_col[threadIdx.x];   // memory load between the payload and its gradient row
    const int w = threadIdx.x / LPE, e0 = threadIdx.x % LPE;
    const int64_t gid = static_cast<int64_t>(blockIdx.x) * NW + w;
    const bool valid = gid < n_chunks;
    int ti = 0;
    if (valid)
        while (ti + 1 < m.n_tables && m.chunk_off[ti + 1] <= gid) ++ti;
    const TableMeta& t = m.tab[ti];
    const int64_t seg0 = t.sorted_off;
    const int64_t seg1 = (ti + 1 < m.n_tables) ? m.tab[ti + 1].sorted_off : m.n;
    const int64_t i0 = valid ? seg0 + (gid - m.chunk_off[ti]) * m.chunk : 0;
    const int64_t i1 = valid ? min(i0 + m.chunk, seg1) : 0;

    auto emit = [&](int tab, uint32_t key, int64_t head_pos, int e, long long hi, long long lo, bool plain, int64_t walker) {
        const TableMeta& tt = m.tab[tab];
        if (e >= tt.dim) return;
        int64_t dst;
        if (tt.mode != 1)
            dst = (walker % ACC_STRIPES) * dense_acc_elems + tt.acc_off + static_cast<int64_t>(key) * tt.dim + e;
        else
            dst = ACC_STRIPES * dense_acc_elems + (head_pos - m.sparse_start) * m.dim_max + e;
        if (plain) {
            acc_hi[dst] = static_cast<unsigned long long>(hi);
            acc_lo[dst] = static_cast<unsigned long long>(lo);
        } else {
            atomicAdd(acc_hi + dst, static_cast<unsigned long long>(hi));
            atomicAdd(acc_lo + dst, static_cast<unsigned long long>(lo));
        }
    };

    // entries are taken in batches of BATCH: the BATCH key / payload loads, then the BATCH gradient loads are
    // independent and in flight together; only then does the run logic walk them (no per-entry load chain)
    constexpr int BATCH = 8;
    const int cnt = static_cast<int>(i1 - i0);
    // the chunk's neighbours and its first batch of entries in ONE round of independent loads (they used to be three dependent
    // round trips -- neighbours, first key, batch -- in front of the gradient rows: at short batches the launch is nothing but that chain)
    uint32_t prev_key = 0u, next_key = 0u;
    uint32_t ks0[BATCH], vs0[BATCH];
#pragma unroll
    for (int j = 0; j < BATCH; ++j) { ks0[j] = 0u; vs0[j] = 0u; }
    if (valid) {
        prev_key = (i0 > seg0) ? ck[i0 - 1] : 0u;
        next_key = (i1 < seg1) ? ck[i1] : 0u;
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const int64_t i = i0 + min(j, cnt - 1);
            ks0[j] = ck[i];
            vs0[j] = val[i];
        }
    }
    uint32_t bad = 0u;
    __syncthreads();                                        // (s_col)
    for (int c0 = 0; c0 < m.dim_max; c0 += LPE) {          // same trip count for every walker (barriers inside)
        const int e = c0 + e0;
        int info = 0;
        long long hh = 0, hl = 0, th = 0, tl = 0;
        int64_t head_h = 0, head_t = 0;
        uint32_t key_h = 0u, key_t = 0u;
        if (valid) {
            uint32_t cur = ks0[0];
            int64_t head = i0;
            bool in_first = !((i0 == seg0) || (prev_key != cur));   // the first run continues the previous chunk's
            long long s_hi = 0, s_lo = 0;
            for (int j0 = 0; j0 < cnt; j0 += BATCH) {
                uint32_t ks[BATCH], vs[BATCH];
                float x[BATCH];
                if (j0 == 0) {
#pragma unroll
                    for (int j = 0; j < BATCH; ++j) { ks[j] = ks0[j]; vs[j] = vs0[j]; }
                } else {
#pragma unroll
                    for (int j = 0; j < BATCH; ++j) {
                        const int64_t i = i0 + min(j0 + j, cnt - 1);
                        ks[j] = ck[i];
                        vs[j] = val[i];
                    }
                }
#pragma unroll
                for (int j = 0; j < BATCH; ++j) {
                    const int slot = static_cast<int>(vs[j] >> 24);
                    const int64_t b = vs[j] & 0xFFFFFFu;
                    x[j] = e < t.dim ? dE[b * ld + s_col[slot] + e] : 0.f;
                }
                long long fh[BATCH], fl[BATCH];
                to_fixed_n<BATCH>(x, fh, fl, bad);
#pragma unroll
                for (int j = 0; j < BATCH; ++j) {
                    if (j0 + j < cnt) {
                        if (ks[j] != cur) {
                            if (in_first) {
                                info |= 1; hh = s_hi; hl = s_lo; key_h = cur;
                                in_first = false;
                            } else {
                                emit(ti, cur, head, e, s_hi, s_lo, true, gid);
                            }
                            cur = ks[j];
                            head = i0 + j0 + j;
                            s_hi = 0;
                            s_lo = 0;
                        }
                        s_hi += fh[j];
                        s_lo += fl[j];
                    }
                }
            }
            const bool last_open = !((i1 == seg1) || (next_key != cur));
            if (in_first) {
                info |= last_open ? 4 : 1; hh = s_hi; hl = s_lo; key_h = cur;
            } else if (last_open) {
                info |= 2; th = s_hi; tl = s_lo; key_t = cur; head_t = head;
            } else {
                emit(ti, cur, head, e, s_hi, s_lo, true, gid);
            }
        }
        // only the first walker can meet a run that started in another workgroup: find where it starts.  The whole first WAVE
        // searches for it, 64 probes per round (three dependent loads for a 65 536-entry segment; the one-lane binary search
        // this replaces made 16 -- under Zipf ids most workgroups open inside a hot row's run, and that chain was most of the launch)
        if (threadIdx.x < 64) {
            const int need = __builtin_amdgcn_readfirstlane((valid && (info & 5) && t.mode == 1) ? 1 : 0);    // (lane 0 = walker 0)
            if (need) {
                const uint32_t skey = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(key_h)));
                int lo = __builtin_amdgcn_readfirstlane(static_cast<int>(seg0));       // (m.n < 2^31: 40 slots x 2^24 samples at most ... checked by the plan)
                int hi = __builtin_amdgcn_readfirstlane(static_cast<int>(i0));
                const int lane = static_cast<int>(threadIdx.x);
                while (lo < hi) {
                    const int step = (hi - lo + 63) / 64;
                    const int p = lo + lane * step;
                    const bool less = p < hi && ck[p] < skey;                   // monotone in the lane: sorted keys
                    const int c = __popcll(__ballot(less));
                    if (c == 0) { hi = lo; break; }
                    const int nlo = lo + (c - 1) * step + 1;
                    hi = c == 64 ? hi : min(hi, lo + c * step);
                    lo = nlo;
                }
                if (w == 0) head_h = hi;
            }
        }
        sh_sum[0][threadIdx.x] = hh; sh_sum[1][threadIdx.x] = hl;
        sh_sum[2][threadIdx.x] = th; sh_sum[3][threadIdx.x] = tl;
        if (e0 == 0) {
            w_info[w] = info | (ti << 8);
            w_key[0][w] = key_h; w_key[1][w] = key_t;
            w_head[0][w] = head_h; w_head[1][w] = head_t;
        }
        __syncthreads();
        if (w == 0) {                                       // lanes e0 of walker 0 join the open runs, in walker order
            bool have = false, outside = false;
            long long c_hi = 0, c_lo = 0;
            int64_t c_head = 0;
            uint32_t c_key = 0u;
            int c_tab = 0;
            const int64_t g0 = static_cast<int64_t>(blockIdx.x) * NW;
            for (int q = 0; q < NW; ++q) {
                const int inf = w_info[q];
                const int idx = q * LPE + e0;
                if (inf & 4) {
                    if (have) {
                        c_hi += sh_sum[0][idx]; c_lo += sh_sum[1][idx];
                    } else {
                        have = true; outside = true;
                        c_hi = sh_sum[0][idx]; c_lo = sh_sum[1][idx];
                        c_key = w_key[0][q]; c_head = w_head[0][q]; c_tab = inf >> 8;
                    }
                    continue;
                }
                if (inf & 1) {
                    if (have) {
                        emit(c_tab, c_key, c_head, e, c_hi + sh_sum[0][idx], c_lo + sh_sum[1][idx], !outside, g0 + q);
                        have = false;
                    } else {
                        emit(inf >> 8, w_key[0][q], w_head[0][q], e, sh_sum[0][idx], sh_sum[1][idx], false, g0 + q);
                    }
                }
                if (inf & 2) {
                    have = true; outside = false;
                    c_hi = sh_sum[2][idx]; c_lo = sh_sum[3][idx];
                    c_key = w_key[1][q]; c_head = w_head[1][q]; c_tab = inf >> 8;
                }
            }
            if (have) emit(c_tab, c_key, c_head, e, c_hi, c_lo, false, g0 + NW - 1);
        }
        __syncthreads();
    }
    if (bad && err) atomicOr(err, SWR_FLAG_GRAD_RANGE);
}

// integer accumulators -> fp32 gradients, ONE launch: workgroups [0, dense_blocks) sweep the dense tables
// (blockIdx.x = table * gx + slice; sums the ACC_STRIPES copies), the rest emit the per-entry rows of the sparse tables
__global__ __launch_bounds__(RB_THREADS) void finalize_kernel(const BwdMeta m, const uint32_t* __restrict__ ck,
                                                              const long long* __restrict__ acc_hi,
                                                              const long long* __restrict__ acc_lo, int64_t dense_acc_elems,
                                                              int gx, int dense_blocks, const SegFin sf,
                                                              const float* __restrict__ part,
                                                              const longlong2* __restrict__ slab) {
    if (static_cast<int>(blockIdx.x) < dense_blocks) {
        const int ti = blockIdx.x / gx;
        const TableMeta& t = m.tab[ti];
        if (t.mode == 1) return;
        const int64_t n = t.vocab * t.dim;
        if (sf.n[ti] > 0) {
            // MFMA segment sums: the partial tables of the table's lookups and batch splits, added in a fixed order
            const float* __restrict__ p0 = part + sf.off[ti];
            const int64_t stride = sf.stride[ti];
            const int np = sf.n[ti];
            for (int64_t j = static_cast<int64_t>(blockIdx.x % gx) * RB_THREADS + threadIdx.x; j < n;
                 j += static_cast<int64_t>(gx) * RB_THREADS) {
                float g = 0.f;
                int q = 0;
                for (; q + 4 <= np; q += 4) {
                    const float v0 = p0[q * stride + j], v1 = p0[(q + 1) * stride + j], v2 = p0[(q + 2) * stride + j],
                                v3 = p0[(q + 3) * stride + j];
                    g += v0; g += v1; g += v2; g += v3;
                }
                for (; q < np; ++q) g += p0[q * stride + j];
                t.grad_dense[j] = t.mode == 2 ? t.grad_dense[j] + g : g;
            }
            return;
        }
        if (sf.n[ti] < 0) {
            // direct sums: the table's slabs (one per lookup x sample chunk), integer sums of (hi, lo) pairs
            const longlong2* __restrict__ s0 = slab + sf.off[ti];
            const int64_t stride = sf.stride[ti];
            const int ns = -sf.n[ti];
            for (int64_t j = static_cast<int64_t>(blockIdx.x % gx) * RB_THREADS + threadIdx.x; j < n;
                 j += static_cast<int64_t>(gx) * RB_THREADS) {
                long long hi = 0, lo = 0;
                int q = 0;
                for (; q + 4 <= ns; q += 4) {
                    const longlong2 v0 = s0[q * stride + j], v1 = s0[(q + 1) * stride + j], v2 = s0[(q + 2) * stride + j],
                                    v3 = s0[(q + 3) * stride + j];
                    hi += (v0.x + v1.x) + (v2.x + v3.x);
                    lo += (v0.y + v1.y) + (v2.y + v3.y);
                }
                for (; q < ns; ++q) {
                    const longlong2 v = s0[q * stride + j];
                    hi += v.x; lo += v.y;
                }
                const float g = from_fixed(hi, lo);
                t.grad_dense[j] = t.mode == 2 ? t.grad_dense[j] + g : g;
            }
            return;
        }
        for (int64_t j = static_cast<int64_t>(blockIdx.x % gx) * RB_THREADS + threadIdx.x; j < n;
             j += static_cast<int64_t>(gx) * RB_THREADS) {
            long long hi = 0, lo = 0;                                     // integer sums: order-free, exact
#pragma unroll
            for (int st = 0; st < ACC_STRIPES; ++st) {
                hi += acc_hi[st * dense_acc_elems + t.acc_off + j];
                lo += acc_lo[st * dense_acc_elems + t.acc_off + j];
            }
            const float g = from_fixed(hi, lo);
            t.grad_dense[j] = t.mode == 2 ? t.grad_dense[j] + g : g;     // mode 2: add to the caller's gradient arena
        }
        return;
    }
    // one thread per (sorted entry of the sparse region, column)
    const int64_t idx = static_cast<int64_t>(blockIdx.x - dense_blocks) * RB_THREADS + threadIdx.x;
    const int64_t i = m.sparse_start + idx / m.dim_max;
    const int e = static_cast<int>(idx % m.dim_max);
    if (i >= m.n) return;
    int ti = m.n_tables - 1;
    while (ti > 0 && m.tab[ti].sorted_off > i) --ti;
    const TableMeta& t = m.tab[ti];
    if (e >= t.dim) return;
    const uint32_t key = ck[i];
    const bool head = (i == t.sorted_off) || (ck[i - 1] != key);
    const int64_t local = i - t.sorted_off;
    if (e == 0) t.urow[local] = head ? static_cast<int32_t>(key) : ~static_cast<int32_t>(key);   // negative = no entry; still ordered by row
    const int64_t a = ACC_STRIPES * dense_acc_elems + (i - m.sparse_start) * m.dim_max + e;
    t.ugrad[local * t.dim + e] = head ? from_fixed(acc_hi[a], acc_lo[a]) : 0.f;
}

extern "C" size_t swr_embed_bwd_workspace_bytes(const swr_embed_grad_slot* slots, int n_slots, int64_t B) {
    HostPlan p;
    if (make_plan(slots, n_slots, B, p, false) != SWR_OK) return 0;
    return p.total;
}

// phases (bits): 1 = build keys + sort (needs only the lookup keys); 2 = reduce of the sorted entries + the sparse
// tables' row lists; 4 = direct sums of the small tables + every dense gradient (2 and 4 need dE; 4 comes after 2)
static int run_embed_bwd(int phases, const swr_embed_grad_slot* slots, int n_slots, const uint32_t* keys, const float* dE,
                         int64_t ld, int64_t B, void* workspace, size_t workspace_bytes, uint32_t* err_flag, void* stream) {
    SWR_REQUIRE(keys && workspace, SWR_ERR_ARG);
    SWR_REQUIRE(!(phases & 6) || (dE && ld > 0), SWR_ERR_ARG);
    if (B == 0) return SWR_OK;
    HostPlan p;
    int rc = make_plan(slots, n_slots, B, p, (phases & 6) != 0);
    if (rc != SWR_OK) return rc;
    SWR_REQUIRE(workspace_bytes >= p.total, SWR_ERR_WORKSPACE);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    uint32_t* kbuf[2] = {reinterpret_cast<uint32_t*>(ws + p.off_k0), reinterpret_cast<uint32_t*>(ws + p.off_k1)};
    uint32_t* vbuf[2] = {reinterpret_cast<uint32_t*>(ws + p.off_v0), reinterpret_cast<uint32_t*>(ws + p.off_v1)};
    uint32_t* hist = reinterpret_cast<uint32_t*>(ws + p.off_hist);
    unsigned long long* acc_hi = reinterpret_cast<unsigned long long*>(ws + p.off_acc_hi);
    unsigned long long* acc_lo = reinterpret_cast<unsigned long long*>(ws + p.off_acc_lo);
    const BwdMeta& m = p.m;
    const int64_t n = m.n;

    const size_t zero_bytes = static_cast<size_t>(p.off_acc_lo - p.off_acc_hi) * 2;
    const bool zero_in_keys = (phases & 1) && n > 0 && zero_bytes % 16 == 0 && swr_aligned16(acc_hi);
    if ((phases & 1) && !zero_in_keys) {
        // both accumulator limbs are contiguous: one zero-fill (a kernel, not a memset node); part of the keys-only half
        rc = swr_zero_async(acc_hi, zero_bytes, st);
        if (rc != SWR_OK) return rc;
    }
    // the direct sums' launch shape (needed up here: at short batches the sort shares their launch)
    bool direct_vec4 = (ld % 4 == 0) && dE && swr_aligned16(dE);
    size_t direct_lds = 0;
    if ((phases & 4) && p.dm.n_blocks > 0) {
        int max_elems = 0;
        for (int g = 0; g < p.dm.n_groups; ++g) max_elems = std::max(max_elems, p.dm.grp[g].elems);
        for (int q = 0; q < p.dm.n_members; ++q) direct_vec4 = direct_vec4 && p.dm.mem[q].dim % 4 == 0 && p.dm.mem[q].col % 4 == 0;
        bool lists = false;
        for (int g = 0; g < p.dm.n_groups; ++g)
            lists = lists || (p.dm.grp[g].n_members == 1 && p.dm.grp[g].chunk <= DIRECT_LIST_MAX);
        direct_lds = 2 * static_cast<size_t>(max_elems) * sizeof(unsigned long long) + 8 +
                     ((direct_vec4 && lists) ? sizeof(uint32_t) * (DIRECT_LIST_MAX + 2) : 0);
        SWR_REQUIRE(direct_lds <= 160 * 1024, SWR_ERR_UNSUPPORTED);
    }
    bool fused_rank_direct = false;
    {
        const char* e = getenv("SWR_K3_FUSE");
        bool segs = false;
        for (int c = 0; c < 3; ++c) segs = segs || p.sg[c].n_jobs > 0;
        // OPT-IN (SWR_K3_FUSE=1): measured at the 8 192-row shards -- config 2 0.2046-0.2059 -> 0.2025-0.2027 ms, config 4 0.3168 ->
        // 0.3203 -- the launch's LDS size is the direct sums' (64 KB of accumulators + 12 KB list + the sort's 35 KB static): one
        // workgroup per CU for BOTH kinds, so the ~700 workgroups run in rounds instead of side by side
        fused_rank_direct = (e && e[0] == '1') && phases == 7 && n > 0 && p.rank_sort && p.dm.n_blocks > 0 && direct_vec4 && !segs &&
                            direct_lds + sizeof(uint32_t) * DIRECT_THREADS * (RANK_EPB + 1) <= 150 * 1024 &&
                            p.max_keys <= 32 * DIRECT_THREADS;
    }
    if ((phases & 1) && n > 0 && p.rank_sort) {
        // short segments: the counting sort reads the lookup's keys itself (no build_keys launch) and carries the zero-fill
        RankSrc src;
        std::memset(&src, 0, sizeof(src));
        src.keys = keys; src.B = B;
        int ns = 0;
        for (int t = 0; t < m.n_tables; ++t) {
            src.first[t] = static_cast<int16_t>(ns);
            // the slots of table t in segment order (slot_dst grows with the slot index inside a table)
            for (int q = 0; q < m.n_sorted_slots; ++q)
                if (slots[m.sorted_slot[q]].table_id == t) src.slot[ns++] = m.sorted_slot[q];
        }
        for (int t = m.n_tables; t <= MAX_SLOTS; ++t) src.first[t] = static_cast<int16_t>(ns);
        for (int t = 0; t < m.n_tables; ++t) {
            src.n_of[t] = static_cast<int16_t>(src.first[t + 1] - src.first[t]);
            src.slot0[t] = src.n_of[t] > 0 ? src.slot[src.first[t]] : 0;
        }
        src.zero = reinterpret_cast<uint4*>(acc_hi);
        src.zero16 = zero_in_keys ? static_cast<int64_t>(zero_bytes / 16) : 0;
        if (fused_rank_direct) {
            // one launch with the direct sums; the sort's blocks are RANK_EPB entries of 512-entry rows: its plan counts the same blocks
            const int rows = static_cast<int>((p.max_keys + DIRECT_THREADS - 1) / DIRECT_THREADS);
            const void* fn = rows <= 4 ? reinterpret_cast<const void*>(rank_direct_kernel<4>)
                           : rows <= 8 ? reinterpret_cast<const void*>(rank_direct_kernel<8>)
                           : rows <= 16 ? reinterpret_cast<const void*>(rank_direct_kernel<16>)
                                        : reinterpret_cast<const void*>(rank_direct_kernel<32>);
            // (static LDS of the sort's counters + the direct sums' dynamic accumulators: past 64 KB in all the attribute is needed, and
            // the dynamic part it allows must leave room for the static one)
            constexpr int rank_static = static_cast<int>(sizeof(uint32_t)) * DIRECT_THREADS * (RANK_EPB + 1);
            if (direct_lds + rank_static > 64 * 1024 && !swr_raise_lds(fn, 160 * 1024 - rank_static)) return SWR_ERR_LAUNCH;
            const dim3 grid(static_cast<unsigned>(p.dm.n_blocks + p.sm.n_tiles));
            longlong2* slab = reinterpret_cast<longlong2*>(ws + p.off_slab);
            if (rows <= 4) hipLaunchKernelGGL(rank_direct_kernel<4>, grid, dim3(DIRECT_THREADS), direct_lds, st, p.dm, keys, dE, ld, slab, err_flag, p.dm.n_blocks, p.sm, src, kbuf[1], vbuf[1]);
            else if (rows <= 8) hipLaunchKernelGGL(rank_direct_kernel<8>, grid, dim3(DIRECT_THREADS), direct_lds, st, p.dm, keys, dE, ld, slab, err_flag, p.dm.n_blocks, p.sm, src, kbuf[1], vbuf[1]);
            else if (rows <= 16) hipLaunchKernelGGL(rank_direct_kernel<16>, grid, dim3(DIRECT_THREADS), direct_lds, st, p.dm, keys, dE, ld, slab, err_flag, p.dm.n_blocks, p.sm, src, kbuf[1], vbuf[1]);
            else hipLaunchKernelGGL(rank_direct_kernel<32>, grid, dim3(DIRECT_THREADS), direct_lds, st, p.dm, keys, dE, ld, slab, err_flag, p.dm.n_blocks, p.sm, src, kbuf[1], vbuf[1]);
        } else {
            rank_sort_launch(p.sm, src, p.max_keys, kbuf, vbuf, st);
        }
    } else if ((phases & 1) && n > 0) {
        hipLaunchKernelGGL(build_keys_kernel, dim3(static_cast<unsigned>(swr_ceil_div(n, RB_THREADS))), dim3(RB_THREADS), 0,
                           st, m, keys, kbuf[0], vbuf[0], reinterpret_cast<uint4*>(acc_hi),
                           zero_in_keys ? static_cast<int64_t>(zero_bytes / 16) : 0);
        radix_sort_launch(p.sm, p.n_passes, kbuf, vbuf, hist, st);
    }
    if (!(phases & 6)) return swr_launch_status();
    const int sorted_buf = p.rank_sort ? 1 : (p.n_passes & 1);
    const uint32_t* ck = kbuf[sorted_buf];                   // where the last pass left the sorted entries
    const uint32_t* sv = vbuf[sorted_buf];

    if (phases & 4) {
        float* part = reinterpret_cast<float*>(ws + p.off_part);
        for (int c = 0; c < 3; ++c) {
            const SegMeta& sg = p.sg[c];
            if (sg.n_jobs == 0) continue;
            const dim3 grid(static_cast<unsigned>(sg.n_jobs * sg.n_splits));
            const int cbk = c == 0 ? 1 : (c == 1 ? 2 : 4);
            const unsigned lds = static_cast<unsigned>((SEG_WAVES - 1) * seg_tpw_big(c) * cbk * 4 * 64 * sizeof(float));
            if (c == 0) hipLaunchKernelGGL((segsum_mfma_kernel<1, 16, 4>), grid, dim3(SEG_WAVES * 64), lds, st, sg, keys, dE, ld, part, err_flag);
            else if (c == 1) hipLaunchKernelGGL((segsum_mfma_kernel<2, 4, 4>), grid, dim3(SEG_WAVES * 64), lds, st, sg, keys, dE, ld, part, err_flag);
            else hipLaunchKernelGGL((segsum_mfma_kernel<4, 2, 2>), grid, dim3(SEG_WAVES * 64), lds, st, sg, keys, dE, ld, part, err_flag);
        }
    }
    if ((phases & 4) && p.dm.n_blocks > 0 && !fused_rank_direct) {
        // LDS sized by the largest group (small groups -> more workgroups per CU); 16-byte loads when every lookup
        // column span is 4-float aligned; the sample list (12 KB) is spent only when some workgroup takes the list branch
        // (direct_kernel: V == 4, one lookup per workgroup, chunk <= DIRECT_LIST_MAX) -- computed above
        const bool vec4 = direct_vec4;
        const size_t lds = direct_lds;
        if (lds > 64 * 1024) {
            // above 64 KB of dynamic LDS the attribute is needed (idempotent, not a stream operation; the first call of a
            // shape happens in a warm-up step, never inside a hipGraph capture -- as bnmix.hip, gemm.hip)
            const void* fn = vec4 ? reinterpret_cast<const void*>(direct_kernel<4>) : reinterpret_cast<const void*>(direct_kernel<1>);
            if (!swr_raise_lds(fn, 160 * 1024)) return SWR_ERR_LAUNCH;
        }
        if (vec4)
            hipLaunchKernelGGL(direct_kernel<4>, dim3(static_cast<unsigned>(p.dm.n_blocks)), dim3(DIRECT_THREADS), lds, st,
                               p.dm, keys, dE, ld, reinterpret_cast<longlong2*>(ws + p.off_slab), err_flag);
        else
            hipLaunchKernelGGL(direct_kernel<1>, dim3(static_cast<unsigned>(p.dm.n_blocks)), dim3(DIRECT_THREADS), lds, st,
                               p.dm, keys, dE, ld, reinterpret_cast<longlong2*>(ws + p.off_slab), err_flag);
    }
    if ((phases & 2) && n > 0) {
        int lpe = 1;
        while (lpe < m.dim_max && lpe < 64) lpe <<= 1;
        const dim3 grid(static_cast<unsigned>(swr_ceil_div(static_cast<int64_t>(p.n_chunks), RB_THREADS / lpe)));
#define LAUNCH_REDUCE(L)                                                                                              \
    hipLaunchKernelGGL(reduce_kernel<L>, grid, dim3(RB_THREADS), 0, st, m, p.n_chunks, ck, sv, dE, ld, acc_hi, acc_lo,   \
                       p.dense_acc_elems, err_flag)
        switch (lpe) {
            case 1: LAUNCH_REDUCE(1); break;
            case 2: LAUNCH_REDUCE(2); break;
            case 4: LAUNCH_REDUCE(4); break;
            case 8: LAUNCH_REDUCE(8); break;
            case 16: LAUNCH_REDUCE(16); break;
            case 32: LAUNCH_REDUCE(32); break;
            default: LAUNCH_REDUCE(64); break;
        }
#undef LAUNCH_REDUCE
    }
    int gx = 0, dense_blocks = 0;
    if ((phases & 4) && (p.dense_acc_elems > 0 || p.slab_elems > 0)) {
        int64_t biggest = 1;
        for (int t = 0; t < m.n_tables; ++t)
            if (m.tab[t].mode != 1 && m.tab[t].vocab * m.tab[t].dim > biggest) biggest = m.tab[t].vocab * m.tab[t].dim;
        gx = static_cast<int>(std::min<int64_t>(swr_ceil_div(biggest, RB_THREADS), 1024));
        dense_blocks = gx * m.n_tables;
    }
    int64_t sparse_blocks = 0;
    if ((phases & 2) && m.sparse_start < n) sparse_blocks = swr_ceil_div((n - m.sparse_start) * m.dim_max, RB_THREADS);
    if (dense_blocks + sparse_blocks > 0)
        hipLaunchKernelGGL(finalize_kernel, dim3(static_cast<unsigned>(dense_blocks + sparse_blocks)), dim3(RB_THREADS), 0, st,
                           m, ck, reinterpret_cast<const long long*>(acc_hi), reinterpret_cast<const long long*>(acc_lo),
                           p.dense_acc_elems, gx > 0 ? gx : 1, dense_blocks, p.sf, reinterpret_cast<const float*>(ws + p.off_part),
                           reinterpret_cast<const longlong2*>(ws + p.off_slab));
    return swr_launch_status();
}

extern "C" int swr_embed_bwd(const swr_embed_grad_slot* slots, int n_slots, const uint32_t* keys, const float* dE,
                             int64_t ld, int64_t B, void* workspace, size_t workspace_bytes, uint32_t* err_flag,
                             void* stream) {
    return run_embed_bwd(7, slots, n_slots, keys, dE, ld, B, workspace, workspace_bytes, err_flag, stream);
}

extern "C" int swr_embed_bwd_sort(const swr_embed_grad_slot* slots, int n_slots, const uint32_t* keys, int64_t B,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    return run_embed_bwd(1, slots, n_slots, keys, nullptr, 0, B, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int swr_embed_bwd_reduce(const swr_embed_grad_slot* slots, int n_slots, const uint32_t* keys, const float* dE,
                                    int64_t ld, int64_t B, void* workspace, size_t workspace_bytes, uint32_t* err_flag,
                                    void* stream) {
    return run_embed_bwd(6, slots, n_slots, keys, dE, ld, B, workspace, workspace_bytes, err_flag, stream);
}

extern "C" int swr_embed_bwd_reduce_part(const swr_embed_grad_slot* slots, int n_slots, const uint32_t* keys, const float* dE,
                                         int64_t ld, int64_t B, int part, void* workspace, size_t workspace_bytes,
                                         uint32_t* err_flag, void* stream) {
    SWR_REQUIRE(part >= 1 && part <= 3, SWR_ERR_ARG);
    return run_embed_bwd(part << 1, slots, n_slots, keys, dE, ld, B, workspace, workspace_bytes, err_flag, stream);
}
