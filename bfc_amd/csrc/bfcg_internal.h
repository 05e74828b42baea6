// bfcg_internal.h -- shared between the kernels (bfcg_kernels.hip) and the host context (bfcg_ctx.hip)
#pragma once
#include <hip/hip_runtime.h>
#define BFCG_TILE1 4096
#define BFCG_TILE2 4096
#define BFCG_SCAN_CH 64
/* records per scatter tile: 20-byte records (k > 47) take 3072 so that two workgroups' stages fit a CU's LDS */
/* dwords per k-mer record: y0 (minus rec_n bucket bits), y1, the quality flag and the 32-bit file index in 96 or 128 bits, else 20 bytes */
static inline int bfcg_rec_dwords(int k, int rec_n) { const int bits = 2 * k - rec_n + 33; return bits <= 96 ? 3 : bits <= 128 ? 4 : 5; }
static inline int bfcg_tile_of_rw(int rw) { return rw == 5 ? 2048 : rw == 4 ? 3072 : 4096; } /* level 2's tile: 48 / 48 / 40 KiB of stage for 12- / 16- / 20-byte records -- with the 2 - 4 KiB of counters three workgroups per CU each (round 6) */
/* positions per tile of stage A (k_hist1 / k_scatter1: bfcg_kernels.hip, S1<RW>) */
static inline int bfcg_tile1_of_rw(int rw) { return rw == 3 ? 4096 : 3072; }
#define BFCG_MAXB 1024   /* most buckets one scatter level fans out to */
#define BFCG_HO_MAX_PAGES 8 /* batches whose seen k-mers may wait in a region's hand-over log for ONE commit pass (k_commit_seg's page arrays, the context's marks) */
/* The measurement switches (KParams.ablate = BFCG_ABLATE: skip the stores / the cursor atomics / the hashing of k_scatter1, phase clocks in
   k_bloom, ...) exist only in a library built with -DBFCG_MEASURE (`python -m bfc_amd.build --measure` -> build/libbfc_gpu_measure.so, loaded
   through BFC_GPU_LIB by the scripts that need them).  In the shipped library BFCG_ABL() is the constant 0: the kernels carry neither the
   tests nor an environment variable that could turn counts wrong. */
#ifdef BFCG_MEASURE
#define BFCG_ABL(P, bits) ((P).ablate & (bits))
#else
#define BFCG_ABL(P, bits) 0
#endif
#include <stdint.h>

namespace bfcg {

// statistics block in device memory (u64 counters)
enum { ST_KMERS = 0, ST_HIGH, ST_SEEN, ST_KEYS, ST_TAB_OVF, ST_ERR_POOL, ST_SLOW_BUCKETS, ST_CROWDED /* regions whose aggregation table was full */, ST_N = 16 };
// counters are replicated ST_SLOTS times (one 128-byte row each) and summed on the host: a single hot
// address costs ~12 ns per atomic chip-wide (MI355X_MICROARCH.md, row fanin)
enum { ST_SLOTS = 256 };

struct KParams {
	int k, q, bf_shift, n_hashes, l_pre, filter_mode;
	int R;          // log2(bloom blocks per LDS region)
	int F, F1, F2;  // fine-bucket bits = bf_shift-9-R, split over two scatter levels (F2 == 0: one level)
	int tab_cshift; // log2(slots per sub-table region)
	uint32_t fs_cap, list_cap; // LDS first-setter table entries (pow2), unresolved-list entries
	uint32_t ag_cap;            // LDS aggregation table entries (pow2)
	uint32_t idx_rank;          // rank << (32 - rank_bits): prefixed to the in-batch position so file order is rank-major
	int track;                  // 1: keep first/last insertion stamps for the byte-identical dump
	int ablate;                 // measurement switches (BFCG_ABLATE): honoured only by a library built with -DBFCG_MEASURE (see BFCG_ABL below)
	int bloom_bt;               // threads per workgroup of the bloom kernel: 512 (three workgroups per CU), 1024 when the LDS footprint allows one only
	int seg;                    // 1: the count table is kept as region-owned segments (seg_tab) and updated through LDS by k_commit_seg
	int seg_shift;              // log2 slots per region's table segment
	int seg_blk;                // a segment is 2^(seg_shift - seg_blk) BLOCKS of 2^seg_blk slots (seg_blk = min(seg_shift, 12): a block fits a third of a CU's LDS with its counter pairs); a key lives in
	                            // block (seg_home(id) >> seg_blk) & (blocks - 1), probing stays inside it: up to 2^12 slots a segment is one block
	int seg_lo, seg_hi;         // bits [seg_lo, seg_hi) of y0 are implied by the region (kmer_dev.h: SegGeom)
	uint32_t f_base;            // global id of this rank's first bloom region
	int no_kstats;              // stage A does not count k-mers / high-quality k-mers (a batch that is replayed was counted the first time)
	int dedupe;                 // this batch goes into a (nearly) empty filter: k_bloom resolves the copies of a k-mer by class first (host's hint, speed only)
	uint32_t ct_cap;            // entries of k_bloom's class table (a power of two; 8-byte entries over the first-setter table and the lists)
	int rec_lo, rec_n;          // bits [rec_lo, rec_lo + rec_n) of y0 are a record's level-1 bucket and are not stored in it (0: everything is stored)
	int b3;                     // the default path's bloom insert runs k_bloom3: a list entry in LDS is 10 bytes (bloom_lds_bytes), and batches without `dedupe` take that kernel
	int b3_warm;                // this batch goes into a warm filter: fs_cap / list_cap are the SHORT list's (four workgroups of k_bloom3 per CU instead of three)
	int b3fm;                   // filter mode (`bfc -1`) on 16-byte records whose bloom address is a bit field of their words: k_bloom3fm (list_cap: its 10-byte entries)
	int l2_big;                 // level 2 of this batch works on tiles of 8192 records with 1024 threads (round 6: 2^10 regions per bucket -- config c4's -b37 --, 12-byte
	                            // records, one-pass partition of a single GPU): a (tile, region) run is 8 records = 96 bytes as at c3's 2^9, not 4 = 48
	int b3_cold;                // this batch goes into a filter that is still filling up: k_bloom3<.., COLD> -- the list ordered by (block, file index), one lane walks a
	                            // block's k-mers as bfc_bf_insert would; list_cap is the cold list's (12-byte entries, no first-setter table beside them)
};

struct BatchBufs {
	uint32_t *rows1, *chunk1;           // level-1 histogram rows [tiles][2^F1] -> offsets; chunk sums
	uint32_t *start1, *row_base;        // level-1 bucket starts (2^F1+1); first level-2 row per bucket (2^F1+1)
	uint32_t *rows2, *start2;           // level-2 histogram rows [rows][2^F2] -> offsets; fine starts (2^F+1)
	uint64_t *recs1, *recs2;            // record buffers, max_kmers * RW words each
	uint64_t max_kmers;
	unsigned long long *bloom, *bloom_hi, *table;
	unsigned long long *stats;
	uint64_t *tab_ovf; uint32_t tab_ovf_cap;
	unsigned long long *pool; uint32_t pool_slices; // slow-path first-setter pool (locked slices, see k_bloom)
	uint8_t *seen_out;
	uint64_t *agg_out; uint32_t *agg_cnt; // aggregated seen k-mers per fine bucket (k_bloom -> k_commit)
	uint32_t *stream_out; int stream;     // STREAM mode: seen k-mers as records (k_bloom -> k_commit_stream); on / off for this batch
	// one-pass level 1 (K1 once per batch): 8 slabs of op_cap records per level-1 bucket in recs1, their cursors (one per 128-byte line), the overflow
	// flags of THIS batch's slot ([0] level 1, [2] level 2) and the run's sticky poison word (stage B's stream only), and the segment arrays level 2
	// reads the slabs through: seg_beg[8 nb1] | seg_end[8 nb1] | row_base[8 nb1 + 1] | bucket_start[nb1 + 1]
	uint32_t *op_cursor, *op_flags, *op_sticky, *op_seg; uint32_t op_cap;
	uint32_t op_own_lo, op_own_n, op_own_delta; // a rank of a multi-GPU group: the slabs of its OWN buckets lie op_own_delta records further on (bfcg_kernels.hip: OnePass)
	uint32_t *cnt2; uint32_t cap2;            // one-pass level 2: a slab of cap2 records per bloom region in recs2 and its cursor (0: two passes, start2 says where)
	uint32_t *cnt_live;                       // two-pass level 2: the regions' record counts (k_scan2) -- start2 alone does not say them when the level-1 slabs hold dead records
	// hand-over log of the region-owned table (bfcg_kernels.hip: BloomArgs): arena, entries per region (0: this batch's entries go to stream_out at
	// its records' offsets and are applied at once), cursors, marks [pages][ho_mark_stride], the page this batch fills, whether stage B ends with
	// the commit of pages 0..ho_page, and the per-page key counters [pages][ST_SLOTS]
	unsigned long long *ho; uint32_t ho_stride; uint32_t *ho_cur, *ho_mark; uint32_t ho_mark_stride, ho_page; int ho_commit; unsigned long long *ho_keys;
	unsigned long long *seg_tab;              // region-owned table segments: [regions][2^seg_shift] slots of id << 14 | high << 8 | count (KParams.seg)
	unsigned long long *tab_first, *sub_last; // order stamps (NULL unless KParams.track)
	unsigned long long batch_hi;              // batch number << 32
};

void run_stage_a(const KParams &P, const BatchBufs &B, const uint8_t *seq, const uint8_t *qual, int64_t n_pos, uint64_t *out1, hipStream_t st, hipEvent_t *ev);
// one-pass stage A: K1 + level-1 scatter into the slabs of B.recs1 + the segment arrays of B.op_seg (no histogram pass); ev as run_stage_a
void run_stage_a_onepass(const KParams &P, const BatchBufs &B, const uint8_t *seq, const uint8_t *qual, int64_t n_pos, uint64_t *out1, hipStream_t st, hipEvent_t *ev);
void run_stage_b(const KParams &P, const BatchBufs &B, const uint64_t *in1, const uint32_t *seg_beg, const uint32_t *seg_end, int n_seg, int segs_per_bucket,
                 const uint32_t *row_base, const uint32_t *bucket_start, uint64_t n_rec_bound, hipStream_t st, hipEvent_t *ev);
// a rank of a group, slab mode without the host's sizes: the fills of stage A's slabs as one row per destination (behind run_stage_a_onepass, same stream);
// the owner's segment arrays (seg_beg | seg_end | row_base | bucket_start, as mg_process_any lays them out) from all sources' rows
void run_pack_rows(const KParams &P, const BatchBufs &B, int n_ranks, uint32_t row_w, uint32_t *rows, hipStream_t st);
void run_seg_setup_mg(const KParams &P, int rw_dwords, const uint32_t *rows, uint32_t row_w, int n_ranks, int s_lo, int s_hi, uint32_t cap, uint32_t *seg, unsigned long long *total, hipStream_t st);
void run_batch(const KParams &P, const BatchBufs &B, const uint8_t *seq, const uint8_t *qual, int64_t n_pos, hipStream_t st, hipEvent_t *ev);
int bloom_lds_bytes(const KParams &P);
bool bloom3fm_geometry_ok(const KParams &P); // k_bloom3fm's conditions (16-byte records, k >= bf_shift + 9: block, h1, h2 are bits of y0; 4 hashes; regions of <= 256 blocks)
bool bloom3_geometry_ok(const KParams &P); // k_bloom3's conditions (12-byte records whose bloom address is a bit field of their words, 4 hashes, regions of <= 256 blocks)
hipError_t set_bloom_lds_attr(const KParams &P);
void run_query(const KParams &P, const uint8_t *seq, int64_t n_pos, const void *bloom, uint8_t *flags, hipStream_t st);
void run_kcov(const KParams &P, const uint8_t *seq, int64_t n_pos, int min_occ, const void *tab, uint8_t *flags, uint16_t *out, hipStream_t st);
void run_streak(int k, float min_frac, const uint8_t *flags, const uint64_t *off, uint64_t n_reads, int32_t *out_start, int32_t *out_end, hipStream_t st);
void run_hash_only(const KParams &P, const uint8_t *seq, const uint8_t *qual, int64_t n_pos, uint64_t *out, hipStream_t st);
void run_table_replay(const KParams &P, unsigned long long *tab, const uint64_t *src, uint64_t n, unsigned long long *stats, uint64_t *ovf, uint32_t ovf_cap,
                      unsigned long long *first, unsigned long long *sub_last, hipStream_t st);
// region-owned segments: grow every segment from 2^old_shift to 2^P.seg_shift slots; replay parked k-mers; convert to the (sub-table, key) layout
void run_seg_rehash(const KParams &P, const unsigned long long *old_tab, int old_shift, int old_blk, unsigned long long *new_tab, uint32_t n_fine, hipStream_t st);
void run_seg_replay(const KParams &P, unsigned long long *seg_tab, const uint64_t *src, uint64_t n, unsigned long long *stats, uint64_t *ovf, uint32_t ovf_cap, hipStream_t st);
void run_seg_to_table(const KParams &P, const unsigned long long *seg_tab, uint32_t n_fine, unsigned long long *tab, unsigned long long *stats, uint64_t *ovf, uint32_t ovf_cap, hipStream_t st);
hipError_t set_seg_lds_attr(void);
hipError_t set_scatter1wc_lds_attr(void); // bfcg_scatter1wc.hip (k_scatter1_wc: level 1 through write-combining buffers in LDS)
// apply the pages 0..pages-1 of the hand-over log to the segments (no bloom pass): before a batch that cannot use the log, and when the pipeline is drained
void run_commit_pages(const KParams &P, const BatchBufs &B, uint32_t n_fine, uint32_t pages, hipStream_t st);
#define BFCG_SEG_MAX_SHIFT 14 /* a segment's BLOCK must fit a CU's LDS: 2^14 slots = 128 KiB */
#define BFCG_SEG_TOTAL_MAX 24 /* slots per region at most (seg_home has 26 bits): 2^10 blocks */
void run_table_rehash(const KParams &P, const unsigned long long *old_tab, int cshift_old, unsigned long long *new_tab,
                      const unsigned long long *old_first, unsigned long long *new_first, hipStream_t st);

} // namespace bfcg
