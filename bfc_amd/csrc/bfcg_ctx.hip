// bfcg_ctx.hip -- host side of the device-level API (include/bfc_gpu.h, PART 2):
// owns the HBM-resident state (bloom filter(s), count table, batch scratch) and drives the
// kernels of bfcg_kernels.hip batch by batch on one HIP stream.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <sys/mman.h>
#include <time.h>
#include <atomic>
#include <mutex>
#include <vector>
#include <thread>
#include <vector>
#include "bfc_gpu.h"
#include "bfcg_internal.h"
#include "bfc_host.h"

using namespace bfcg;

static thread_local char g_err[512] = "";
enum { HO_MAX_PAGES = BFCG_HO_MAX_PAGES };
enum { OP_FLAG_WORDS = 12, OP_STICKY = 8 }; // bfcg_ctx.op_flags: two slots of four words, the sticky poison word
static double dbg_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static inline void set_seg_shift(KParams &P, int s, int blk_max) { P.seg_shift = s; P.seg_blk = s < blk_max ? s : blk_max; }
static int set_err(const char *fmt, ...)
{
	va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
	fprintf(stderr, "[E::bfcg] %s\n", g_err);
	return -1;
}
#define HIPCK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return set_err("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define HIPCKN(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_err("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return NULL; } } while (0)

struct bfcg_ctx {
	bfcg_params_t prm;
	KParams P;
	BatchBufs B;
	hipStream_t st;              // stage B (level 2, bloom regions, table) -- also the stream of everything synchronous
	hipStream_t stA;             // stage A (K1 histogram + level-1 scatter) of the NEXT batch runs here, under stage B of the current one
	hipStream_t stC;             // small D2H copies that must not queue behind kernels
	hipEvent_t evt[2][7];        // timing events per in-flight batch ([6] = start of stage B on its stream)
	hipEvent_t evA[2], evB[2], evCopy; // stage A done / stage B done (buffer set reusable) / host batch copied
	uint32_t *rows1[2], *chunk1[2], *start1[2]; uint64_t *recs1[2]; // double-buffered stage-A outputs
	uint8_t *d_seq2[2], *d_qual2[2]; // staging for host batches (one per in-flight batch)
	uint32_t *d_planes[2];       // the same for batches that arrive as bit planes (4 planes of plane_cap words each; allocated at the first such batch)
	uint64_t plane_cap;
	int cur, pend;               // buffer set of the next batch; 1 if the previous batch is not finalised yet
	int used[2];
	uint8_t *d_seq, *d_qual;     // = d_seq2[0], d_qual2[0]
	unsigned long long *h_stats; // pinned mirror
	unsigned long long *h_snap[2]; // pinned: the counters as they stood when stage B of the batch in this slot ended (copied on stream st right behind it:
	                             // keys and seen are touched by stage B alone, so these two are EXACT per batch without draining the pipeline)
	uint64_t call_no, slot_call[2]; // top-level count calls so far; the call a slot's batch belongs to
	int call_depth;
	uint64_t final_call, call_keys[64]; // last call with a finalised batch; distinct keys after the last finalised batch of call (no % 64)
	uint64_t n_batches;
	float last_ms[6];
	double sum_ms[6]; uint64_t n_timed; // cumulative per-stage GPU time of finalised batches (bfcg_stage_ms)
	int rw;                      // bytes per record: 12 (k <= 31), 16 (k <= 47), 20
	uint64_t bloom_bytes;        // bytes of the bloom slice this rank owns
	int n_ranks, rank, log2n;
	uint32_t *d_seg, *h_seg;     // multi-GPU: seg_beg | seg_end | row_base | bucket_start (device / pinned host), one set per in-flight batch
	size_t seg_words;
	uint64_t recv_cap;           // records the level-2 buffers can take
	uint64_t keys_last, grow[2]; // distinct keys at the last finalised batch; keys added by the last two batches (growth forecast)
	uint64_t crowded_last;       // ST_CROWDED at the last finalised batch
	int stream_mode;             // 1: the batches' k-mers hardly repeat -- seen k-mers are streamed to k_commit_stream instead of aggregated
	uint32_t *stream_out; uint64_t n_stream_batches;
	double seen_per_pos;         // seen k-mers per stream position in the last finalised batch
	int cold;                    // the filter is (nearly) empty: most k-mers of the next batch will find clear bits and go through a region's LDS list
	uint64_t seen_last, pos_final, slot_pos[2]; // seen k-mers at the last finalised batch; positions of the batch(es) finalised since; positions of a slot's batch
	int pipeline;                // stage A of batch t+1 on its own stream under stage B of batch t
	int seg_ok;                  // the geometry allows region-owned table segments (KParams.seg): every reset starts in that layout
	int seg_total_max;           // slots of a segment at most (log2; BFCG_SEG_TOTAL lowers BFCG_SEG_TOTAL_MAX: 14 = round 3's behaviour, larger tables leave for the host's layout)
	int seg_blk_max;             // slots of a segment's block (log2): 14 = what a CU's LDS holds; BFCG_SEG_BLOCK lowers it so that small tests run segments of several blocks
	int b3_ok;                   // ... and the bloom insert of batches without `dedupe` runs k_bloom3 (KParams.b3)
	uint32_t fs_cap_w, list_cap_w; // k_bloom3's LDS tables for batches into a WARM filter (0: none): a footprint of a quarter of a CU's LDS
	uint32_t list_cap_cw;        // ... and for batches into a warm filter where those take the walk too (BFCG_B3_WALK_WARM): a quarter of a CU's LDS
	uint32_t list_cap_c;         // k_bloom3<.., COLD>'s list for batches into a filter that is still filling up (0: none): 12-byte entries, no first-setter table
	int seg_init_shift;          // log2 slots per segment after a reset
	int seg_escaped;             // the segments outgrew LDS (or the table was exported): converted to the (sub-table, key) layout until the next reset
	uint64_t n_seg_grow;         // segment growths since creation
	// one-pass level-1 partition (K1 once per batch; single-rank contexts).  A batch whose slabs overflow (few, often repeated k-mers) poisons itself and
	// the batches behind it on the device; the host then replays them, in order, through the two-pass partition and stays there until the next reset.
	int onepass_ok, onepass;     // the buffers exist / the current run still uses the one-pass partition
	uint32_t *op_cursor[2], *op_seg[2], *op_flags, *h_flags[2];
	int unfinished, unfinished_mg;     // bfcg_mg_process_slabs_dev has enqueued a stage B that bfcg_mg_process_finish has not seen yet (and its entry in the replay queue)
	uint32_t *h_rows[2]; int rows_slot; // a group's slab mode without the host's wait: the host's copy of a stage A's rows (bfcg_mg_scatter_slabs_async), the slot of the last one
	uint32_t op_cap; uint64_t op_min_pos;
	uint32_t *cnt2; uint32_t cap2; uint64_t recs2_n; // one-pass level 2 (region slabs); records recs2 / stream_out hold
	struct opq_t { const uint8_t *seq, *qual; uint64_t n_pos; int slot; const void *recv; int mg; } opq[4]; int n_opq; // batches enqueued one-pass and not yet known to be clean
	int mg_op2_ok, mg_op2, mg_op2_allowed; uint32_t *mg_seg[4];  // a rank of a multi-GPU run: level 2 (its own stage B) in one pass; copies of the queued batches' segment sizes
	int mg_slab_ok; int mg_seg_slab[4]; // ... level 1 in one pass into per-destination slabs (bfcg_mg_scatter_slabs); which queued copies are slab fills
	uint64_t n_replayed;
	int reused;                  // a reset has followed counted batches: this context counts one data set after the other
	int seg_no_grow;             // the next segment size does not fit (memory / LDS): grow only when a segment overflows or the load passes 85 %
	int seg_cap_shift;           // the allocation behind B.seg_tab holds segments of up to 2^seg_cap_shift slots
	// Hand-over log of the region-owned table: the seen k-mers of up to ho_K batches wait per region and are applied in ONE pass over the table
	// segments (k_commit_seg).  A page = one batch's entries; ho_stride = ho_K x (a region's slab at level 2) entries per region.
	unsigned long long *ho, *ho_keys, *h_ho_keys[2]; uint32_t *ho_cur, *ho_mark; uint32_t ho_stride; int ho_K;
	int ho_pending;                     // pages filled since the last commit
	uint64_t ho_page_call[HO_MAX_PAGES]; // the call a page's batch belongs to
	int slot_commit[2]; uint64_t slot_page_call[2][HO_MAX_PAGES]; // pages committed inside the stage B of the batch in this slot, and their calls
	int slot_all_committed[2];          // nothing was left in the log behind that batch: the counters' snapshot is the table's exact state
	uint64_t keys_known;                // distinct keys after the last absorbed commit
	int commit_absorbed;                // a commit's key counts have arrived since the last growth forecast
	uint64_t keys_per_batch; int n_commits, ho_window; // most keys one batch created in the last absorbed commit; commits absorbed since the reset; pages the current window takes
	unsigned long long *seg_spare; int seg_spare_shift; // the buffer the last growth left behind (kept up to 16 GiB): growth rehashes from one into the other,
	                             // so a context that counts one data set after the other stops calling hipMalloc / hipFree (tens of ms per multi-GiB call)
};

extern "C" const char *bfcg_last_error(void) { return g_err; }
extern "C" void bfcg_set_error(const char *msg) { set_err("%s", msg); } // bfcg_mg.hip reports through the same channel

extern "C" int bfcg_device_count(void)
{
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess) { (void)hipGetLastError(); return 0; }
	return ndev;
}

extern "C" void bfcg_params_default(bfcg_params_t *p)
{
	memset(p, 0, sizeof(*p));
	p->k = 33; p->q = 20; p->bf_shift = 33; p->n_hashes = 4; p->l_pre = 20; // bfc.c:17-26
	p->max_batch_pos = 1ULL << 27;
}

static int clamp_lpre(int k, int l_pre) // htab.c:24-26
{
	if (k * 2 - l_pre > BFC_CH_KEYBITS) l_pre = k * 2 - BFC_CH_KEYBITS;
	if (l_pre > BFC_CH_MAXPRE) l_pre = BFC_CH_MAXPRE;
	return l_pre;
}

extern "C" bfcg_ctx_t *bfcg_create(const bfcg_params_t *prm)
{
	int ndev = 0;
	const bool timing = getenv("BFC_GPU_TIMING") != 0; // (phase times on stderr, as bfc_count's)
	const double t_c0 = dbg_now();
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_err("no HIP device available: the counting path has no CPU fallback"); return NULL; }
	const double t_c1 = dbg_now();
	if (prm->device < 0 || prm->device >= ndev) { set_err("device %d out of range (%d devices)", prm->device, ndev); return NULL; }
	if (prm->k < 1 || prm->k > 63) { set_err("k=%d outside [1,63] (bfc.h:8, htab.c:23)", prm->k); return NULL; }
	if (prm->bf_shift < 9 + 0 || prm->bf_shift > 37) { set_err("bf_shift=%d outside [9,37] (bbf.c:9, bfc.h:9)", prm->bf_shift); return NULL; }
	if (prm->n_hashes < 1 || prm->n_hashes > 12) { set_err("n_hashes=%d outside [1,12]", prm->n_hashes); return NULL; }
	if (prm->max_batch_pos == 0 || prm->max_batch_pos >= (1ULL << 32)) { set_err("max_batch_pos must be in [1, 2^32)"); return NULL; }
	const int n_ranks = prm->n_ranks > 0 ? prm->n_ranks : 1;
	int log2n = 0; while ((1 << log2n) < n_ranks) ++log2n;
	if ((1 << log2n) != n_ranks || n_ranks > 64 || prm->rank < 0 || prm->rank >= n_ranks) { set_err("n_ranks=%d must be a power of two <= 64 and 0 <= rank=%d < n_ranks", n_ranks, prm->rank); return NULL; }
	if (n_ranks > 1 && prm->max_batch_pos >= (1ULL << (32 - log2n))) { set_err("with %d ranks a batch holds < 2^%d positions", n_ranks, 32 - log2n); return NULL; }
	HIPCKN(hipSetDevice(prm->device));

	bfcg_ctx_t *c = (bfcg_ctx_t *)calloc(1, sizeof(bfcg_ctx_t));
	c->prm = *prm;
	KParams &P = c->P;
	P.k = prm->k; P.q = prm->q; P.bf_shift = prm->bf_shift; P.n_hashes = prm->n_hashes; P.filter_mode = prm->filter_mode;
	P.l_pre = clamp_lpre(prm->k, prm->l_pre);
	if (!prm->filter_mode && 2 * prm->k < P.l_pre) { // htab.c:49-50 shifts by 2k - l_pre: negative, undefined in the reference itself
		set_err("k=%d is too small for l_pre=%d: the count table needs 2k >= l_pre (htab.c:45-58)", prm->k, P.l_pre); free(c); return NULL;
	}
	// geometry of the bloom kernel; environment overrides are tuning knobs, not semantics
	{
		const char *e;
		P.R = prm->region_shift > 0 ? prm->region_shift : ((e = getenv("BFCG_R")) ? atoi(e) : 8);
		if (P.R > 10) P.R = 10;
		if (P.R < 4) P.R = 4;
		if (P.R > P.bf_shift - 9) P.R = P.bf_shift - 9;
		while (P.bf_shift - 9 - P.R > 20 && P.R < 10) ++P.R; // two scatter levels of at most 1024 buckets each: 16 KiB regions up to -b37
		P.F = P.bf_shift - 9 - P.R;
		if (P.F <= 8) { P.F1 = P.F; P.F2 = 0; }
		else { P.F2 = (P.F + 1) / 2; if (P.F2 > 10) P.F2 = 10; P.F1 = P.F - P.F2; }
		if (P.F2 > 0 && (e = getenv("BFCG_F1"))) { // how the two levels share the F bits (profiles/round3_scatter_probe.txt: a 256-bucket level 1 is the cheaper skeleton)
			const int f1 = atoi(e);
			if (f1 >= 1 && f1 <= 10 && P.F - f1 >= 1 && P.F - f1 <= 10 && f1 >= log2n) { P.F1 = f1; P.F2 = P.F - f1; }
		}
		if (P.F1 > 10) { set_err("bf_shift=%d needs more than two scatter levels at region_shift=%d", P.bf_shift, P.R); free(c); return NULL; }
		if (n_ranks > 1 && P.F2 == 0) { // with several ranks level 2 also gathers a bucket's records from the sources' blocks: it must exist
			if (P.F <= log2n) { set_err("bf_shift=%d gives %d bloom regions: too few for %d ranks", P.bf_shift, 1 << P.F, n_ranks); free(c); return NULL; }
			P.F1 = (P.F + 1) / 2 > log2n ? (P.F + 1) / 2 : log2n; P.F2 = P.F - P.F1;
		}
#ifdef BFCG_MEASURE
		P.ablate = (e = getenv("BFCG_ABLATE")) ? atoi(e) : 0;
#else
		P.ablate = 0; // (the measurement switches are not compiled into this library: bfcg_internal.h)
		if ((e = getenv("BFCG_ABLATE")) && atoi(e)) fprintf(stderr, "[W::bfcg] BFCG_ABLATE=%s ignored: this library was built without -DBFCG_MEASURE\n", e);
#endif
		// records do not store the bits of y0 that their level-1 bucket implies (bfcg_kernels.hip: RecGeom) -- where the bucket IS a bit field of y0:
		// k >= bf_shift - 9 (the bloom block id is the low bf_shift-9 bits of the hash, and those are y0's, kmer.h:87)
		// -- and where it makes the record smaller (c3: k=33, 9 bucket bits: 12 instead of 16 bytes; c2's k=31 records are 12 bytes anyway)
		{ const char *e = getenv("BFCG_REC_DROP"); P.rec_n = (P.k >= P.bf_shift - 9 && bfcg_rec_dwords(P.k, P.F1) < bfcg_rec_dwords(P.k, 0) && !(e && atoi(e) == 0)) ? P.F1 : 0; P.rec_lo = P.R + P.F2; }
		c->rw = 4 * bfcg_rec_dwords(P.k, P.rec_n);
		P.ag_cap = (e = getenv("BFCG_AG")) ? (uint32_t)atoi(e) : 256;
		// LDS budget: a third of a CU (3 workgroups of 512 threads resident = 24 waves), else half, else all of it -- the first tier that
		// leaves 16 KiB for the list and the first-setter table next to the region and the second slice / aggregation table
		const size_t region = (size_t)64 << P.R;
		size_t rwb = 8; // list entry: file-order index + (record index | mask)
		{ // region-owned table segments (bfcg_kernels.hip: k_commit_seg) where a k-mer's identity inside its region fits the 50 key bits of a slot;
		  // decided here because the bloom kernel then needs no aggregation table in LDS: the room goes to the list (larger batches at full speed)
			const char *es = getenv("BFCG_SEG");
			P.seg_lo = P.R < P.k ? P.R : P.k;
			P.seg_hi = P.k < P.bf_shift - 9 ? P.k : P.bf_shift - 9;
			if (P.seg_hi < P.seg_lo) P.seg_hi = P.seg_lo;
			const int id_bits = 2 * P.k - (P.seg_hi - P.seg_lo);
			c->seg_ok = !prm->filter_mode && !prm->track_order && P.R <= 8 && !getenv("BFCG_BT") && id_bits <= BFC_CH_KEYBITS && prm->table_layout != 1 && !(es && atoi(es) == 0);
			P.seg = c->seg_ok;
			// the default path's bloom insert (bfcg_kernels.hip: k_bloom3) keeps 10-byte list entries: bloom address and clear-bit mask instead of
			// the record's index, which gets two bytes of its own
			c->b3_ok = c->seg_ok && prm->n_hashes == 4 && c->rw == 12 && bloom3_geometry_ok(P) && !getenv("BFCG_NO_B3");
			P.b3 = c->b3_ok;
			if (P.b3) rwb = 10;
		}
		const size_t second = prm->filter_mode ? region : P.seg ? 0 : (size_t)P.ag_cap * ((P.k > 32 ? 24 : 16) + (prm->track_order ? 8 : 0)); // second filter's slice, or the aggregation table
		size_t budget = (size_t)53000;
		if (region + second + 16 * 1024 > budget) budget = 80 * 1024 - 1024;
		if (region + second + 16 * 1024 > budget) budget = 160 * 1024 - 1024;
		if ((e = getenv("BFCG_LDS")) != 0 && (size_t)atoi(e) >= region + second + 4096 && (size_t)atoi(e) <= 160 * 1024 - 1024) budget = (size_t)atoi(e);
		size_t left = budget - region - second - 16;
		// split: 8 B per k-mer with clear bits (list) and 4 B per first-setter entry (power of two).  Only contended bits get an entry;
		// the worst realistic case (a genome at ~1x per batch: every new k-mer twice) has ~1.2 per list entry: list <= 3/4 of the table
		uint32_t fs = 512, best_fs = 512, best_list = 0;
		for (; fs <= 32768 && (size_t)fs * 4 + rwb * 64 <= left; fs <<= 1) {
			uint32_t list = (uint32_t)((left - (size_t)fs * 4) / rwb);
			if (list > fs / 4 * 3) list = fs / 4 * 3;
			if (list > best_list) { best_list = list; best_fs = fs; }
		}
		fs = best_fs;
		if ((e = getenv("BFCG_FS")) != 0) { fs = (uint32_t)atoi(e); best_list = (uint32_t)((left - (size_t)fs * 4) / rwb); }
		P.fs_cap = fs;
		P.list_cap = best_list;
		if (P.list_cap > 8191) P.list_cap = 8191; // 13-bit list index inside a first-setter entry
		// Into a warm filter only the unseen k-mers take list entries and the library sizes such batches for 60 % of the list above (split_rule):
		// a list of 3/4 of that size still has a quarter of head room, and with a first-setter table of half the size (contended bits are few once
		// the filter is warm) the footprint drops to a quarter of a CU's LDS -- FOUR workgroups of k_bloom3 per CU instead of three (c3: the same 9
		// batches, bloom stage 65.5 -> see profiles/round4_k_bloom.md).  A region that does overflow takes the exact slow path as ever.
		c->fs_cap_w = c->list_cap_w = 0;
		if (P.b3 && !getenv("BFCG_LDS") && !(getenv("BFCG_B3_WARM") && atoi(getenv("BFCG_B3_WARM")) == 0)) {
			const size_t bw = 40900;
			if (bw > region + 16 + 8192) {
				const size_t lw = bw - region - 16;
				uint32_t fw = fs; while (fw > 1024 && (size_t)fw * 4 + (size_t)(P.list_cap / 4 * 3) * rwb > lw) fw >>= 1;
				uint32_t l = (size_t)fw * 4 < lw ? (uint32_t)((lw - (size_t)fw * 4) / rwb) : 0;
				if (l > P.list_cap) l = P.list_cap;
				if (l >= P.list_cap / 4 * 3 && fw >= 1024) { c->fs_cap_w = fw; c->list_cap_w = l; }
			}
		}
		// Into a filter that is still filling up (round 5): k_bloom3<.., COLD> orders the list by (block, file index) and walks the blocks; its entries
		// are 12 bytes and need no first-setter table beside them -- the same third of a CU's LDS holds 2 879 of them instead of 2 021
		c->list_cap_c = 0; c->list_cap_cw = 0;
		if (P.b3 && !(getenv("BFCG_B3_COLD") && atoi(getenv("BFCG_B3_COLD")) == 0)) {
			const size_t fixed = region + ((size_t)2 << P.R) * 4 + 16 + 16;
			if (budget > fixed + 12 * 256) {
				uint32_t l = (uint32_t)((budget - fixed) / 12);
				if (l > 4096) l = 4096; // (B3_COLD_NR x 512 positions in the rank pass)
				c->list_cap_c = l;
			}
			c->list_cap_cw = 0;
			if ((size_t)40900 > fixed + 12 * 256 && getenv("BFCG_B3_WALK_WARM") && atoi(getenv("BFCG_B3_WALK_WARM"))) c->list_cap_cw = (uint32_t)(((size_t)40900 - fixed) / 12);
		}
		// filter mode on 16-byte records (config c5: k = 51, -b37): k_bloom3fm keeps both slices, the block counters and 10-byte list entries -- no
		// first-setter table; a region whose list overflows is taken in rounds of file-index ranges, so the capacity only sets the speed
		P.b3fm = 0;
		if (prm->filter_mode && c->rw == 16 && bloom3fm_geometry_ok(P) && !getenv("BFCG_NO_B3")) {
			const size_t fixed = 2 * region + ((size_t)2 << P.R) * 4 + 16 + 16;
			if (budget > fixed + 10 * 256) {
				uint32_t l = (uint32_t)((budget - fixed) / 10);
				if ((e = getenv("BFCG_B3FM_LIST")) != 0 && atoi(e) >= 64 && (uint32_t)atoi(e) < l) l = (uint32_t)atoi(e); // (tests: a tiny list forces the rounds)
				if (l > 4096) l = 4096;
				P.b3fm = 1; P.list_cap = l; P.fs_cap = 512;
			}
		}
		{ // the class table of cold batches lies over the first-setter table and the lists
			const size_t room = (size_t)P.fs_cap * 4 + (size_t)P.list_cap * 8; // (k_bloom, which serves those batches, uses 8 of a list entry's bytes)
			uint32_t ct = 1; while ((size_t)ct * 2 * 8 <= room) ct <<= 1;
			P.ct_cap = ct;
		}
	}
	P.tab_cshift = prm->tab_cshift > 0 ? prm->tab_cshift : (P.l_pre <= 20 ? 5 : 3);
	if (prm->tab_cshift <= 0) // a first batch can create up to half as many keys as it has k-mers: start with a quarter of the batch's positions in slots
		while ((1ULL << (P.l_pre + P.tab_cshift)) < prm->max_batch_pos / 4 && P.l_pre + P.tab_cshift < 34) ++P.tab_cshift;
	P.track = (prm->track_order && !prm->filter_mode) ? 1 : 0;
	// one workgroup per CU (regions of 32 KiB and more: -b36, -b37) runs 1024 threads so that the CU still has 16 waves; no such variant with order stamps
	{ const char *e = getenv("BFCG_BT"); P.bloom_bt = e ? atoi(e) : (bloom_lds_bytes(P) > 80 * 1024 && !P.track && P.n_hashes == 4) ? 1024 : 512; if (P.bloom_bt != 1024 || P.track || P.n_hashes != 4) P.bloom_bt = 512; }
	c->n_ranks = n_ranks; c->rank = prm->rank; c->log2n = log2n;
	// Stage A of the next batch runs under stage B of this one (two streams): c2 14.2 vs 15.4 ms per step.  While the count table was updated by
	// random device-scope atomics the overlap HURT config c3 (389 vs 328 ms per step: the scatter kernels crawled beside k_commit_stream); with
	// the table streamed through LDS (k_commit_seg) it is neither (265.9 vs 269.0 ms): batches of half a billion positions keep the chip busy on
	// their own, and under rocprofv3 the overlapped kernels stretch.  So: two streams for batches up to 2^28 positions.  BFCG_PIPELINE=0/1 overrides.
	{ const char *e = getenv("BFCG_PIPELINE"); c->pipeline = e ? atoi(e) != 0 : prm->max_batch_pos <= (1ULL << 28); }

	if (n_ranks > 1 && log2n > P.F1) { set_err("multi-GPU needs a two-level partition with 2^F1=%d >= n_ranks (bf_shift=%d is too small)", 1 << P.F1, P.bf_shift); free(c); return NULL; }
	P.idx_rank = n_ranks > 1 ? (uint32_t)prm->rank << (32 - log2n) : 0u;
	c->bloom_bytes = (1ULL << (P.bf_shift - 3)) >> log2n; // owner computes: this rank keeps 1/n_ranks of the regions

	BatchBufs &B = c->B;
	B.max_kmers = prm->max_batch_pos;
	const int nb1 = 1 << P.F1, nfine = (1 << P.F) >> log2n; // fine buckets owned by this rank
	// records arriving from all ranks for the owned buckets: hashing balances them; 25 % + 1 M head room, checked per batch
	c->recv_cap = n_ranks > 1 ? B.max_kmers + B.max_kmers / 4 + (1u << 20) : B.max_kmers;
	HIPCKN(hipStreamCreate(&c->st)); HIPCKN(hipStreamCreate(&c->stA)); HIPCKN(hipStreamCreate(&c->stC));
	for (int b = 0; b < 2; ++b) {
		for (int i = 0; i < 7; ++i) HIPCKN(hipEventCreate(&c->evt[b][i]));
		HIPCKN(hipEventCreateWithFlags(&c->evA[b], hipEventDisableTiming)); HIPCKN(hipEventCreateWithFlags(&c->evB[b], hipEventDisableTiming));
	}
	HIPCKN(hipEventCreateWithFlags(&c->evCopy, hipEventDisableTiming));
	{
		const uint64_t tile = (uint64_t)bfcg_tile_of_rw(c->rw / 4), tile1 = (uint64_t)bfcg_tile1_of_rw(c->rw / 4);
		const uint64_t tiles1 = (prm->max_batch_pos + tile1 - 1) / tile1, chunks1 = (tiles1 + BFCG_SCAN_CH - 1) / BFCG_SCAN_CH;
		// one ragged row per segment at most (one-pass level 1: 8 segments per bucket); a one-pass level 1 hands over its slabs' FILL, dead records
		// included -- up to the slabs' capacity, 9/8 of the batch (+ rounding) --, and the two-pass level 2 keeps a histogram row for every tile of that
		const uint64_t rows2 = (c->recv_cap + c->recv_cap / 8) / tile + (uint64_t)nb1 * 16 + 8;
		for (int b = 0; b < 2; ++b) {
			HIPCKN(hipMalloc(&c->rows1[b], sizeof(uint32_t) * tiles1 * nb1));
			HIPCKN(hipMalloc(&c->chunk1[b], sizeof(uint32_t) * chunks1 * nb1));
			HIPCKN(hipMalloc(&c->start1[b], sizeof(uint32_t) * (nb1 + 1) * 2));
		}
		B.rows1 = c->rows1[0]; B.chunk1 = c->chunk1[0]; B.start1 = c->start1[0]; B.row_base = B.start1 + nb1 + 1;
		if (P.F2 > 0) {
			HIPCKN(hipMalloc(&B.rows2, sizeof(uint32_t) * rows2 * (1u << P.F2)));
			HIPCKN(hipMalloc(&B.start2, sizeof(uint32_t) * (nfine + 1)));
		}
	}
	{ // one-pass level 1: 8 slabs per bucket, together 9/8 of the batch's positions (a stream holds ~0.8 k-mers per position: a quarter of head room)
		const char *e = getenv("BFCG_ONEPASS");
		c->onepass_ok = n_ranks == 1 && P.F2 > 0 && !(e && atoi(e) == 0); // level 2 gathers a bucket's slabs: filters of 2^26 bits and more
		uint64_t cap = (B.max_kmers + B.max_kmers / 8) / ((uint64_t)nb1 * 8) + 1;
		while (cap * nb1 * 8 > 0xffffffffULL) --cap;
		if (cap >= 1024) cap &= ~31ULL; // (32 records of 12 bytes are three 128-byte lines: k_scatter1_wc's chunks then begin on sector boundaries in every slab;
		else if (cap >= 8) cap &= ~3ULL; //  four records are three 16-byte pieces: what its copy-out needs at least)
		c->op_cap = (uint32_t)cap;
		// a slab should expect ~1000 records or more: below that its fill scatters by more than the head room, and the batch would be replayed
		c->op_min_pos = (e = getenv("BFCG_ONEPASS_MIN_TILES")) ? (uint64_t)atoi(e) * 4096 : (uint64_t)nb1 * 8 * 1024;
	}
	// (one pass: + a tile and a chunk of slack behind the last slab.  A run that finds its slab full is still stored, at the slab's start -- the batch
	// is poisoned and replayed, nobody reads it -- and a run is up to a whole tile where every k-mer of it falls into one bucket (poly-A reads):
	// with slabs smaller than a tile such a run reaches into the following slabs, and behind the last one it must still be inside the buffer)
	const uint64_t recs1_n = c->onepass_ok ? (uint64_t)c->op_cap * nb1 * 8 + (uint64_t)bfcg_tile1_of_rw(c->rw / 4) + 64 : B.max_kmers;
	// (batches above 2^28 positions run on ONE stream -- c->pipeline -- so stage A of a batch never runs beside stage B of the one before: both
	// slots share one level-1 buffer there, which leaves config c4 34 GB more for the hand-over log and the table's growth)
	HIPCKN(hipMalloc(&c->recs1[0], recs1_n * c->rw));
	if (n_ranks == 1 && !c->pipeline) c->recs1[1] = c->recs1[0];
	else HIPCKN(hipMalloc(&c->recs1[1], recs1_n * c->rw));
	{ // a rank of a multi-GPU run partitions what it RECEIVES in one pass (level 2 is the owner's own stage B; level 1 feeds the exchange and stays two-pass)
		const char *e = getenv("BFCG_ONEPASS");
		c->mg_op2_ok = P.F2 > 0 && !(e && atoi(e) == 0); // (also the single rank of a group of one: bench.py with BFC_BENCH_FORCE_DIST, tests)
	}
	{ const char *e = getenv("BFCG_MG_SLABS"); c->mg_slab_ok = P.F2 > 0 && !(e && atoi(e) == 0); } // (a rank's level 1 in one pass: bfcg_mg_scatter_slabs; used by groups only)
	if (c->onepass_ok || c->mg_slab_ok) {
		for (int b = 0; b < 2; ++b) {
			HIPCKN(hipMalloc(&c->op_cursor[b], sizeof(uint32_t) * (8 * nb1 * 32 + 32))); // (+ the tile counter of k_scatter1)
			HIPCKN(hipMalloc(&c->op_seg[b], sizeof(uint32_t) * ((size_t)25 * nb1 + 8)));
		}
	}
	if (c->mg_slab_ok) for (int b = 0; b < 2; ++b) HIPCKN(hipHostMalloc(&c->h_rows[b], sizeof(uint32_t) * ((size_t)8 * nb1 + 2 * (size_t)n_ranks)));
	if (c->onepass_ok || c->mg_op2_ok || c->mg_slab_ok) {
		for (int b = 0; b < 2; ++b) HIPCKN(hipHostMalloc(&c->h_flags[b], 4 * sizeof(uint32_t)));
		// per batch slot b: op_flags[4 b + 0] a level-1 slab overflowed (raised on stage A's stream), [4 b + 2] a region's slab (stage B's stream);
		// op_flags[8]: the run is poisoned (k_seal, stage B's stream only) -- a batch's stage B never reads what another batch's stage A writes
		HIPCKN(hipMalloc(&c->op_flags, OP_FLAG_WORDS * sizeof(uint32_t)));
		HIPCKN(hipMemset(c->op_flags, 0, OP_FLAG_WORDS * sizeof(uint32_t)));
	}
	if (c->mg_op2_ok) for (int i = 0; i < 4; ++i) c->mg_seg[i] = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n_ranks * (size_t)(nb1 >> log2n) * 8); // (slab fills: 8 per source and bucket)
	B.recs1 = c->recs1[0];
	c->recs2_n = c->recv_cap;
	if (c->onepass_ok || c->mg_op2_ok || c->mg_slab_ok) { // level 2 in one pass: a slab per region, 9/8 of the mean of a full batch's positions + 64 records (k-mers are ~0.8 of the positions)
		const char *e = getenv("BFCG_ONEPASS2");
		const uint64_t cap2 = (B.max_kmers + B.max_kmers / 8) / (uint64_t)nfine + 64, n2 = cap2 * (uint64_t)nfine + 8192; // (a tile of slack: 8192 = the largest level-2 tile, KParams.l2_big)
		if (!(e && atoi(e) == 0) && n2 < 0xffffffffULL) {
			c->cap2 = (uint32_t)cap2; c->recs2_n = n2 > c->recv_cap ? n2 : c->recv_cap;
			HIPCKN(hipMalloc(&c->cnt2, sizeof(uint32_t) * (size_t)nfine));
		} else { c->mg_op2_ok = 0; HIPCKN(hipMalloc(&c->cnt2, sizeof(uint32_t) * (size_t)nfine)); } // (the counters also serve the two-pass level 2: BatchBufs.cnt_live)
		B.cnt_live = c->cnt2;
	}
	if (P.F2 > 0) HIPCKN(hipMalloc(&B.recs2, c->recs2_n * c->rw));
	c->seg_words = (size_t)25 * nb1 + 8; // (slab mode: 8 segments per source and bucket)
	{ HIPCKN(hipMalloc(&c->d_seg, sizeof(uint32_t) * 2 * c->seg_words)); HIPCKN(hipHostMalloc(&c->h_seg, sizeof(uint32_t) * 2 * c->seg_words)); }
	HIPCKN(hipMalloc(&B.bloom, c->bloom_bytes));
	P.f_base = (uint32_t)c->rank * (uint32_t)nfine;
	if (c->seg_ok) { // segments start with a quarter of a batch's positions in slots, like the table below
		int sh = prm->tab_cshift > 0 ? prm->tab_cshift + 3 : 6;
		if (prm->tab_cshift <= 0) while (((uint64_t)nfine << sh) < prm->max_batch_pos / 4 && sh < BFCG_SEG_MAX_SHIFT) ++sh;
		if (sh > BFCG_SEG_MAX_SHIFT) sh = BFCG_SEG_MAX_SHIFT;
		c->seg_init_shift = sh;
		// A block of 2^12 slots, not the 2^14 a CU's LDS could hold: blocks up to 2^12 take the counter-pair path at three workgroups of 512 threads per
		// CU -- c4's 2^13-slot segments as two such blocks instead of one of 64 KiB (1024 threads, compare-and-swap): commits 328 -> 273 ms at 12x
		// coverage (2^11: 317; every block's workgroup reads all the region's entries), same table
		{ const char *e = getenv("BFCG_SEG_BLOCK"); c->seg_blk_max = e && atoi(e) >= 3 && atoi(e) <= BFCG_SEG_MAX_SHIFT ? atoi(e) : 12; }
		{ const char *e = getenv("BFCG_SEG_TOTAL"); c->seg_total_max = e && atoi(e) >= c->seg_blk_max && atoi(e) <= BFCG_SEG_TOTAL_MAX ? atoi(e) : BFCG_SEG_TOTAL_MAX; }
		if (c->seg_total_max > c->seg_blk_max + 15) c->seg_total_max = c->seg_blk_max + 15; // (the blocks are a grid's second dimension: at most 65 535)
		P.seg = 1; set_seg_shift(P, sh, c->seg_blk_max); c->seg_cap_shift = sh;
		HIPCKN(set_seg_lds_attr());
		HIPCKN(hipMalloc(&B.seg_tab, ((uint64_t)nfine << sh) * 8));
		{ // the hand-over log: ho_K pages per region where level 2 bounds a region's share of a batch (its slab), else one batch at its records' offsets
			const char *e = getenv("BFCG_COMMIT_K");
			int K = e ? atoi(e) : 4;
			if (K < 1) K = 1;
			if (K > HO_MAX_PAGES) K = HO_MAX_PAGES;
			if (!c->cap2) K = 1;
			size_t free_b = 0, total_b = 0;
			const uint64_t page = (uint64_t)nfine * (c->cap2 ? c->cap2 : 1) * 8;
			if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) while (K > 1 && page * (uint64_t)K > (uint64_t)free_b / 3) --K; // (a third of what is free, at most: the segments grow into the rest)
			c->ho_K = K;
			c->ho_stride = c->cap2 ? (uint32_t)K * c->cap2 : 0u;
			uint64_t entries = (uint64_t)nfine * c->ho_stride;
			if (entries < c->recs2_n) entries = c->recs2_n; // (two-pass batches hand over at their records' offsets)
			HIPCKN(hipMalloc(&c->ho, (entries + 4096) * 8));
			HIPCKN(hipMalloc(&c->ho_cur, sizeof(uint32_t) * (size_t)nfine));
			HIPCKN(hipMalloc(&c->ho_mark, sizeof(uint32_t) * (size_t)nfine * HO_MAX_PAGES));
			HIPCKN(hipMalloc(&c->ho_keys, sizeof(unsigned long long) * ST_SLOTS * HO_MAX_PAGES));
			for (int b = 0; b < 2; ++b) HIPCKN(hipHostMalloc(&c->h_ho_keys[b], sizeof(unsigned long long) * ST_SLOTS * HO_MAX_PAGES));
		}
	}
	if (P.filter_mode) HIPCKN(hipMalloc(&B.bloom_hi, c->bloom_bytes));
	else if (!c->seg_ok) HIPCKN(hipMalloc(&B.table, 8ULL << (P.l_pre + P.tab_cshift)));
	if (P.track) { HIPCKN(hipMalloc(&B.tab_first, 8ULL << (P.l_pre + P.tab_cshift))); HIPCKN(hipMalloc(&B.sub_last, 8ULL << P.l_pre)); }
	HIPCKN(hipMalloc(&B.stats, sizeof(unsigned long long) * ST_N * (ST_SLOTS + 1))); // last row: unslotted words
	// parked k-mers of a batch that outruns the table (the host grows it and replays them): a batch cannot create more keys than
	// half its k-mers, bloom false positives aside
	{ uint64_t cap = prm->max_batch_pos / 2; if (cap < (1u << 20)) cap = 1u << 20; if (cap > (1u << 28)) cap = 1u << 28; B.tab_ovf_cap = (uint32_t)cap; }
	HIPCKN(hipMalloc(&B.tab_ovf, (uint64_t)B.tab_ovf_cap * 40));
	// HBM first-setter pool for regions whose LDS tables overflow: 256 slices of 2 entries per region bit (the most a region can need), each
	// behind a lock word.  A workgroup takes any free slice and waits for one if all are taken (their holders never wait for anything), so no
	// input can exhaust the pool; 512 MiB instead of the 2 GiB that one slice per possibly resident workgroup cost at every context creation
	B.pool_slices = 256;
	{
		const uint64_t pool_words = (uint64_t)B.pool_slices + (uint64_t)B.pool_slices * ((uint64_t)1024 << P.R);
		HIPCKN(hipMalloc(&B.pool, pool_words * 8));
		HIPCKN(hipMemset(B.pool, 0, (size_t)B.pool_slices * 8));
	}
	if (prm->debug_seen) HIPCKN(hipMalloc(&B.seen_out, prm->max_batch_pos));
	if (!P.filter_mode) { // filter mode hands nothing over: both filters' slices are in LDS
		// (the aggregation buffer of the host-layout commit -- 1.6 GB at -b35, 6.4 GB at -b37 -- is allocated when a batch first needs it:
		// a context on region-owned segments never does unless its table escapes to the host's layout: ensure_agg)
		if (!c->seg_ok) HIPCKN(hipMalloc(&B.agg_out, (uint64_t)nfine * P.ag_cap * (P.track ? 32 : 24)));
		HIPCKN(hipMalloc(&B.agg_cnt, sizeof(uint32_t) * nfine));
	}
	for (int b = 0; b < 2; ++b) { HIPCKN(hipMalloc(&c->d_seq2[b], prm->max_batch_pos)); HIPCKN(hipMalloc(&c->d_qual2[b], prm->max_batch_pos + 64)); }
	c->d_seq = c->d_seq2[0]; c->d_qual = c->d_qual2[0];
	HIPCKN(hipHostMalloc(&c->h_stats, sizeof(unsigned long long) * ST_N * (ST_SLOTS + 2)));
	for (int b = 0; b < 2; ++b) HIPCKN(hipHostMalloc(&c->h_snap[b], sizeof(unsigned long long) * ST_N * (ST_SLOTS + 1)));
	const double t_c2 = dbg_now();
	HIPCKN(set_bloom_lds_attr(P));
	HIPCKN(set_scatter1wc_lds_attr());
	const double t_c3 = dbg_now();
	if (bfcg_reset(c) != 0) { bfcg_destroy(c); return NULL; }
	if (timing) fprintf(stderr, "[T::bfcg_create] HIP runtime up %.3f s, device buffers %.3f s, kernel attributes %.3f s, filter and table cleared %.3f s\n", t_c1 - t_c0, t_c2 - t_c1, t_c3 - t_c2, dbg_now() - t_c3);
	return c;
}

extern "C" void bfcg_destroy(bfcg_ctx_t *c)
{
	if (!c) return;
	(void)hipSetDevice(c->prm.device);
	(void)hipDeviceSynchronize();
	for (int b = 0; b < 2; ++b) { (void)hipFree(c->rows1[b]); (void)hipFree(c->chunk1[b]); (void)hipFree(c->start1[b]); if (b == 0 || c->recs1[1] != c->recs1[0]) (void)hipFree(c->recs1[b]); (void)hipFree(c->d_seq2[b]); (void)hipFree(c->d_qual2[b]); (void)hipFree(c->d_planes[b]); }
	(void)hipFree(c->B.rows2); (void)hipFree(c->B.start2); (void)hipFree(c->B.recs2);
	(void)hipFree(c->B.bloom); (void)hipFree(c->B.bloom_hi); (void)hipFree(c->B.table); (void)hipFree(c->B.stats); (void)hipFree(c->B.tab_first); (void)hipFree(c->B.sub_last);
	(void)hipFree(c->stream_out); (void)hipFree(c->B.seg_tab); (void)hipFree(c->seg_spare);
	(void)hipFree(c->ho); (void)hipFree(c->ho_cur); (void)hipFree(c->ho_mark); (void)hipFree(c->ho_keys);
	for (int b = 0; b < 2; ++b) if (c->h_ho_keys[b]) (void)hipHostFree(c->h_ho_keys[b]);
	for (int b = 0; b < 2; ++b) { (void)hipFree(c->op_cursor[b]); (void)hipFree(c->op_seg[b]); if (c->h_flags[b]) (void)hipHostFree(c->h_flags[b]); if (c->h_rows[b]) (void)hipHostFree(c->h_rows[b]); }
	(void)hipFree(c->op_flags); (void)hipFree(c->cnt2); for (int i = 0; i < 4; ++i) free(c->mg_seg[i]);
	(void)hipFree(c->B.tab_ovf); (void)hipFree(c->B.pool); (void)hipFree(c->B.seen_out); (void)hipFree(c->B.agg_out); (void)hipFree(c->B.agg_cnt);
	(void)hipHostFree(c->h_stats); (void)hipHostFree(c->h_snap[0]); (void)hipHostFree(c->h_snap[1]); (void)hipFree(c->d_seg); if (c->h_seg) (void)hipHostFree(c->h_seg);
	for (int b = 0; b < 2; ++b) { for (int i = 0; i < 7; ++i) (void)hipEventDestroy(c->evt[b][i]); (void)hipEventDestroy(c->evA[b]); (void)hipEventDestroy(c->evB[b]); }
	(void)hipEventDestroy(c->evCopy);
	(void)hipStreamDestroy(c->st); (void)hipStreamDestroy(c->stA); (void)hipStreamDestroy(c->stC);
	free(c);
}

static int drain(bfcg_ctx_t *c);
static int replay_poisoned(bfcg_ctx_t *c);
static int mg_process_any(bfcg_ctx_t *c, const void *d_recv, const uint32_t *seg_cnt, uint32_t slab_cap, hipEvent_t *wait, int n_wait);
extern "C" int bfcg_mg_process_ev(bfcg_ctx_t *c, const void *d_recv, const uint32_t *seg_cnt, hipEvent_t *wait, int n_wait);
static int enqueue_batch(bfcg_ctx_t *c, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t n_pos, int wait_copy, int no_kstats);
static int ensure_agg(bfcg_ctx_t *c);
static int same_block_offset(bfcg_ctx_t *c, int b, const uint8_t *d_seq, const uint8_t **d_qual, uint64_t n_pos, hipStream_t s);
static int commit_pending_pages(bfcg_ctx_t *c, int b);
static void absorb_pages(bfcg_ctx_t *c, int b);
static int handover_begin(bfcg_ctx_t *c, BatchBufs &Bt, int b, int slabs, uint64_t call);
static int handover_end(bfcg_ctx_t *c, const BatchBufs &Bt, int b);

// k_bloom resolves the copies of a k-mer by class before the bit-level protocol (KParams.dedupe) where that pays: into an empty filter
// (c3's first launch 14.2 -> 10.3 ms); from the second batch on most k-mers are seen outright and the extra passes cost more than they save
// k_bloom3 with the short list and four workgroups per CU for a batch into a warm filter (see bfcg_create)
static void warm_tables(const bfcg_ctx_t *c, KParams &Pt)
{
	Pt.b3_warm = 0; Pt.b3_cold = 0;
	if (Pt.b3 && c->cold && c->list_cap_c) { Pt.list_cap = c->list_cap_c; Pt.b3_cold = 1; Pt.dedupe = 0; } // (the walk resolves copies by itself)
	else if (Pt.b3 && !Pt.dedupe && !c->cold && c->list_cap_cw && c->list_cap_cw >= c->list_cap_w) { Pt.list_cap = c->list_cap_cw; Pt.b3_warm = 1; Pt.b3_cold = 1; }
	else if (Pt.b3 && !Pt.dedupe && !c->cold && c->list_cap_w) { Pt.fs_cap = c->fs_cap_w; Pt.list_cap = c->list_cap_w; Pt.b3_warm = 1; }
}
static int dedupe_hint(const bfcg_ctx_t *c) { return (c->n_batches == 0 || (c->cold && c->seen_per_pos < 0.15)) && !getenv("BFCG_NO_DEDUPE"); }

extern "C" int bfcg_reset(bfcg_ctx_t *c)
{
	const double t_dbg = dbg_now();
	struct dbg_guard { double t; ~dbg_guard() { if (getenv("BFCG_DEBUG")) fprintf(stderr, "[D::reset] %.3f ms\n", (dbg_now() - t) * 1e3); } } dg{t_dbg};
	HIPCK(hipSetDevice(c->prm.device));
	if (c->pend && drain(c) != 0) return -1;
	// the statistics first: stage A of the next batch (stream stA) adds to them and only has to wait for that small memset; the
	// filters and the table are touched by stage B alone, on this same stream, so zeroing them needs no host synchronisation
	HIPCK(hipMemsetAsync(c->B.stats, 0, sizeof(unsigned long long) * ST_N * (ST_SLOTS + 1), c->st));
	if (c->op_flags) HIPCK(hipMemsetAsync(c->op_flags, 0, OP_FLAG_WORDS * sizeof(uint32_t), c->st)); // stage A raises them: cleared before stage A's stream goes on
	HIPCK(hipEventRecord(c->evCopy, c->st));
	HIPCK(hipStreamWaitEvent(c->stA, c->evCopy, 0));
	HIPCK(hipMemsetAsync(c->B.bloom, 0, c->bloom_bytes, c->st));
	if (c->B.bloom_hi) HIPCK(hipMemsetAsync(c->B.bloom_hi, 0, c->bloom_bytes, c->st));
	if (c->seg_ok) { // back to region-owned segments of the initial size (a run that outgrew them, or an export, left the other layout behind)
		const uint64_t nfine = ((uint64_t)1 << c->P.F) >> c->log2n;
		if (c->seg_escaped || !c->B.seg_tab) {
			HIPCK(hipStreamSynchronize(c->st));
			if (c->B.table) { HIPCK(hipFree(c->B.table)); c->B.table = 0; }
			if (c->B.seg_tab) { HIPCK(hipFree(c->B.seg_tab)); c->B.seg_tab = 0; }
			set_seg_shift(c->P, c->seg_init_shift, c->seg_blk_max); c->seg_cap_shift = c->seg_init_shift;
			HIPCK(hipMalloc(&c->B.seg_tab, (nfine << c->P.seg_shift) * 8));
		} else {
			set_seg_shift(c->P, c->seg_init_shift, c->seg_blk_max); // start small again inside the allocation the last run grew to (seg_cap_shift says how far it goes)
			if (c->n_batches && c->seg_spare && c->seg_spare_shift < c->seg_cap_shift && ((nfine << c->seg_cap_shift) * 8) <= (16ULL << 30)) {
				// second data set on this context: make the spare as large as the segments grew, so that this run's growths find their target ready
				HIPCK(hipStreamSynchronize(c->st));
				HIPCK(hipFree(c->seg_spare)); c->seg_spare = 0;
				if (hipMalloc(&c->seg_spare, (nfine << c->seg_cap_shift) * 8) != hipSuccess) { (void)hipGetLastError(); c->seg_spare = 0; }
				c->seg_spare_shift = c->seg_cap_shift;
			}
		}
		if (c->n_batches) c->reused = 1;
		c->P.seg = 1; c->P.b3 = c->b3_ok; c->seg_escaped = 0;
		HIPCK(hipMemsetAsync(c->ho_cur, 0, sizeof(uint32_t) * (size_t)nfine, c->st));
		c->ho_pending = 0; c->slot_commit[0] = c->slot_commit[1] = 0; c->slot_all_committed[0] = c->slot_all_committed[1] = 0; c->keys_known = 0; c->commit_absorbed = 0;
		c->keys_per_batch = 0; c->n_commits = 0; c->ho_window = 1;
		HIPCK(hipMemsetAsync(c->B.seg_tab, 0, (nfine << c->P.seg_shift) * 8, c->st));
	}
	if (c->B.table) HIPCK(hipMemsetAsync(c->B.table, 0, 8ULL << (c->P.l_pre + c->P.tab_cshift), c->st));
	if (c->B.tab_first) { HIPCK(hipMemsetAsync(c->B.tab_first, 0xff, 8ULL << (c->P.l_pre + c->P.tab_cshift), c->st)); HIPCK(hipMemsetAsync(c->B.sub_last, 0, 8ULL << c->P.l_pre, c->st)); }
	c->n_batches = 0;
	c->keys_last = 0; c->grow[0] = c->grow[1] = 0;
	c->crowded_last = 0; c->stream_mode = c->P.seg ? 1 : 0;
	c->cold = 1; c->seen_last = c->pos_final = 0; c->seg_no_grow = 0;
	c->onepass = c->onepass_ok; c->mg_op2 = c->mg_op2_ok; c->n_opq = 0;
	c->call_no = c->final_call = 0; c->call_depth = 0; memset(c->call_keys, 0, sizeof(c->call_keys));
	return 0;
}

// bring the slotted counters to the host and fold them into h_stats[0..ST_N)
static int fetch_stats_on(bfcg_ctx_t *c, hipStream_t s)
{
	unsigned long long *raw = c->h_stats + ST_N;
	HIPCK(hipMemcpyAsync(raw, c->B.stats, sizeof(unsigned long long) * ST_N * (ST_SLOTS + 1), hipMemcpyDeviceToHost, s));
	HIPCK(hipStreamSynchronize(s));
	for (int i = 0; i < ST_N; ++i) {
		unsigned long long s = 0;
		for (int j = 0; j < ST_SLOTS; ++j) s += raw[(size_t)j * ST_N + i];
		c->h_stats[i] = s;
	}
	c->h_stats[ST_TAB_OVF] = raw[(size_t)ST_SLOTS * ST_N]; // the overflow list index is a single word
	return 0;
}

static int fetch_stats(bfcg_ctx_t *c) { return fetch_stats_on(c, c->st); }
// the snapshot of slot b (complete once evB[b] has fired) folded into h_stats
static void fold_snapshot(bfcg_ctx_t *c, int b)
{
	const unsigned long long *raw = c->h_snap[b];
	for (int i = 0; i < ST_N; ++i) {
		unsigned long long s = 0;
		for (int j = 0; j < ST_SLOTS; ++j) s += raw[(size_t)j * ST_N + i];
		c->h_stats[i] = s;
	}
	c->h_stats[ST_TAB_OVF] = raw[(size_t)ST_SLOTS * ST_N];
	absorb_pages(c, b);
	if (c->slot_all_committed[b]) { // nothing waits in the hand-over log behind this batch: the snapshot is the table's state
		c->final_call = c->slot_call[b]; c->call_keys[c->slot_call[b] & 63] = c->h_stats[ST_KEYS];
		if (c->h_stats[ST_KEYS] > c->keys_known && c->h_stats[ST_KEYS] - c->keys_known > c->keys_per_batch) c->keys_per_batch = c->h_stats[ST_KEYS] - c->keys_known; // (a batch applied at its records' offsets, parked k-mers replayed)
		c->keys_known = c->h_stats[ST_KEYS]; c->commit_absorbed = 1;
	}
	c->pos_final += c->slot_pos[b]; c->slot_pos[b] = 0;
}

static int table_maintain(bfcg_ctx_t *c);
static int seg_maintain(bfcg_ctx_t *c);
static int seg_target_shift(const bfcg_ctx_t *c);
static int seg_to_legacy(bfcg_ctx_t *c, int for_export = 0);
static void note_growth(bfcg_ctx_t *c);
static int finalise_previous(bfcg_ctx_t *c, int b);
static int table_target_cshift(const bfcg_ctx_t *c);
static int use_stream(bfcg_ctx_t *c);

static int batch_times(bfcg_ctx_t *c, int b)
{
	// stage B's stream has been synchronised by the caller; the stage-A events of this slot sit on stream stA and are long complete when stage A
	// did work, but an EMPTY stage A (a rank without a share, or the extra pairs of a batch processed in source groups) only records them,
	// possibly behind that stream's wait for an older stage B: wait for them explicitly
	for (int i = 0; i < 3; ++i) HIPCK(hipEventSynchronize(c->evt[b][i]));
	for (int i = 0; i < 5; ++i) HIPCK(hipEventElapsedTime(&c->last_ms[i], c->evt[b][i == 2 ? 6 : i], c->evt[b][i + 1]));
	c->last_ms[5] = c->last_ms[0] + c->last_ms[1] + c->last_ms[2] + c->last_ms[3] + c->last_ms[4]; // GPU time of the stages (they overlap across batches)
	for (int i = 0; i < 6; ++i) c->sum_ms[i] += c->last_ms[i];
	++c->n_timed;
	return 0;
}
static int check_health(bfcg_ctx_t *c)
{
	if (c->h_stats[ST_ERR_POOL]) return set_err("first-setter pool exhausted in %llu bloom regions: batch too large for max_batch_pos", (unsigned long long)c->h_stats[ST_ERR_POOL]);
	if (c->P.seg) return seg_maintain(c);
	if (c->B.table) return table_maintain(c);
	return 0;
}
// finish every batch in flight: afterwards the device is idle, statistics are final, the table is maintained
static int drain(bfcg_ctx_t *c)
{
	if (!c->pend) return 0;
	HIPCK(hipStreamSynchronize(c->stA));
	HIPCK(hipStreamSynchronize(c->st));
	HIPCK(hipGetLastError());
	c->pend = 0;
	if (batch_times(c, c->cur ^ 1) != 0) return -1;
	absorb_pages(c, c->cur); absorb_pages(c, c->cur ^ 1); // (older slot first)
	if (c->n_opq) { // batches that went through the one-pass partition: were their slabs large enough?
		uint32_t sticky = 0;
		HIPCK(hipMemcpy(&sticky, c->op_flags + OP_STICKY, sizeof(sticky), hipMemcpyDeviceToHost));
		if (sticky) return replay_poisoned(c);
		c->n_opq = 0;
	}
	if (c->P.seg && c->ho_pending) { // what still waits in the hand-over log
		if (commit_pending_pages(c, 0) != 0) return -1;
		HIPCK(hipStreamSynchronize(c->st));
		HIPCK(hipGetLastError());
		absorb_pages(c, 0);
	}
	if (fetch_stats(c) != 0) return -1;
	c->keys_known = c->h_stats[ST_KEYS]; c->commit_absorbed = 1;
	c->final_call = c->call_no; c->call_keys[c->call_no & 63] = c->h_stats[ST_KEYS];
	c->pos_final += c->slot_pos[0] + c->slot_pos[1]; c->slot_pos[0] = c->slot_pos[1] = 0;
	note_growth(c);
	return check_health(c);
}

extern "C" int bfcg_sync(bfcg_ctx_t *c) { return drain(c); }

// The one-pass partition gave up on a batch (a level-1 slab overflowed: few, often repeated k-mers): that batch and every batch enqueued behind it
// changed NOTHING on the device (k_seg_setup handed level 2 empty segments).  The device is idle.  Replay them in order through the two-pass
// partition -- their k-mers were counted the first time -- and stay with it until the next reset: the input is skewed.
static int replay_poisoned(bfcg_ctx_t *c)
{
	const int n = c->n_opq;
	bfcg_ctx::opq_t q[4];
	for (int i = 0; i < n; ++i) q[i] = c->opq[i];
	c->n_opq = 0; c->onepass = 0; c->mg_op2 = 0;
	if (getenv("BFCG_DEBUG")) {
		uint32_t fl[OP_FLAG_WORDS];
		HIPCK(hipMemcpy(fl, c->op_flags, sizeof(fl), hipMemcpyDeviceToHost));
		fprintf(stderr, "[D::replay] %d batches; flags (level-1 slab, -, region slab, -) of slot 0: %u %u %u %u, slot 1: %u %u %u %u, sticky %u\n", n, fl[0], fl[1], fl[2], fl[3], fl[4], fl[5], fl[6], fl[7], fl[OP_STICKY]);
	}
	HIPCK(hipMemset(c->op_flags, 0, OP_FLAG_WORDS * sizeof(uint32_t)));
	if (fetch_stats(c) != 0) return -1;
	c->n_batches -= (uint64_t)n; // the batches keep their places in the count (order stamps carry the batch number)
	std::vector<uint32_t> seg;
	for (int i = 0; i < n; ++i) if (q[i].mg) --c->call_no; // (bfcg_mg_process_ev numbers its calls itself)
	for (int i = 0; i < n; ++i) {
		if (q[i].mg) { // a rank's stage B: what it received is still in its receive buffer (the exchange of the next batch has not begun: this thread starts it)
			const int slab = c->mg_seg_slab[q[i].mg - 1];
			const size_t w = (size_t)c->n_ranks * (size_t)((1 << c->P.F1) >> c->log2n) * (slab ? 8 : 1);
			seg.assign(c->mg_seg[q[i].mg - 1], c->mg_seg[q[i].mg - 1] + w);
			if (mg_process_any(c, q[i].recv, seg.data(), slab ? c->op_cap : 0u, 0, 0) != 0) return -1;
		} else if (enqueue_batch(c, q[i].seq, q[i].qual, q[i].n_pos, 0, 1) != 0) return -1;
		if (drain(c) != 0) return -1;
		++c->n_replayed;
	}
	return 0;
}

// keys the next batch is expected to add: the smaller of the last two batches' additions (one batch alone says nothing:
// the first batch of a high-coverage read set brings nearly all keys, the following ones almost none)
static uint64_t growth_forecast(const bfcg_ctx_t *c) { return c->grow[0] < c->grow[1] ? c->grow[0] : c->grow[1]; }
static void note_growth(bfcg_ctx_t *c)
{
	const uint64_t keys = c->h_stats[ST_KEYS];
	{ // fewer than half of the last batch's k-mers were seen before: the filter is still filling up
		// (the seen counter is exact per batch -- stage B alone moves it -- the k-mer counter is not: stage A of the next batch runs ahead;
		// so the batch's size is taken from its positions, of which ~0.8 are k-mers)
		const uint64_t ds = c->h_stats[ST_SEEN] - c->seen_last, np = c->pos_final;
		if (np) { c->cold = ds * 5 < np * 2; c->seen_per_pos = (double)ds / (double)np; }
		c->seen_last = c->h_stats[ST_SEEN]; c->pos_final = 0;
	}
	if (!c->P.seg || c->commit_absorbed) { // (with the hand-over log the table changes when a commit lands: a forecast per commit, not per batch)
		c->grow[1] = c->grow[0]; c->grow[0] = keys > c->keys_last ? keys - c->keys_last : 0; c->keys_last = keys;
		c->commit_absorbed = 0;
	}
	// more than half of the regions filled their aggregation table in the batch(es) just finalised: aggregation does not pay here
	const uint64_t crowded = c->h_stats[ST_CROWDED], nfine = ((uint64_t)1 << c->P.F) >> c->log2n;
	if (!c->stream_mode && c->B.table && !c->P.track && c->P.n_hashes == 4 && c->P.bloom_bt == 512 && !getenv("BFCG_NO_STREAM") && (crowded - c->crowded_last) * 2 > nfine) c->stream_mode = 1;
	c->crowded_last = crowded;
}
// sub-table size the table should have now: load <= 1/2 counting parked k-mers and the forecast
static int table_target_cshift(const bfcg_ctx_t *c)
{
	const KParams &P = c->P;
	const uint64_t need = c->h_stats[ST_KEYS] + c->h_stats[ST_TAB_OVF], g = growth_forecast(c);
	int t = P.tab_cshift;
	if ((need + g) * 2 > (1ULL << (P.l_pre + t)))
		while ((need + 2 * g) * 2 > (1ULL << (P.l_pre + t)) && P.l_pre + t < 36) ++t;
	if (c->h_stats[ST_TAB_OVF] && t == P.tab_cshift) ++t; // a full sub-table under a low overall load
	if (t > P.tab_cshift) { // the old table lives until the new one is filled: grow only as far as memory allows, and run fuller instead (<= 80 %)
		size_t free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
			const uint64_t margin = 2ULL << 30;
			while (t > P.tab_cshift && (8ULL << (P.l_pre + t)) + margin > (uint64_t)free_b) --t;
			if (t == P.tab_cshift && !c->h_stats[ST_TAB_OVF] && need * 5 <= (4ULL << (P.l_pre + t))) return t; // cannot grow, need not yet
			if (t == P.tab_cshift) return t + 1; // must grow and cannot: table_maintain reports it
		}
	}
	return t;
}

// grow the table (any number of doublings in one rehash) and replay parked k-mers until none is left
static int table_maintain(bfcg_ctx_t *c)
{
	KParams &P = c->P; BatchBufs &B = c->B;
	for (;;) {
		const uint64_t ovf = c->h_stats[ST_TAB_OVF];
		const int target = table_target_cshift(c);
		if (ovf == 0 && target == P.tab_cshift) return 0;
		if (ovf > B.tab_ovf_cap) return set_err("count table overflow list exhausted (%llu parked k-mers)", (unsigned long long)ovf);
		if (P.l_pre + target > 36) return set_err("count table cannot grow beyond 2^36 slots");
		{
			size_t free_b = 0, total_b = 0;
			if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (8ULL << (P.l_pre + target)) + (1ULL << 30) > (uint64_t)free_b)
				return set_err("count table of %llu keys cannot grow to 2^%d slots: %.1f GiB of device memory free", (unsigned long long)c->h_stats[ST_KEYS], P.l_pre + target, free_b / 1073741824.0);
		}
		int old_cshift = P.tab_cshift;
		unsigned long long *nt = 0;
		P.tab_cshift = target;
		unsigned long long *nf = 0;
		HIPCK(hipMalloc(&nt, 8ULL << (P.l_pre + P.tab_cshift)));
		HIPCK(hipMemsetAsync(nt, 0, 8ULL << (P.l_pre + P.tab_cshift), c->st));
		if (B.tab_first) { HIPCK(hipMalloc(&nf, 8ULL << (P.l_pre + P.tab_cshift))); HIPCK(hipMemsetAsync(nf, 0xff, 8ULL << (P.l_pre + P.tab_cshift), c->st)); }
		run_table_rehash(P, B.table, old_cshift, nt, B.tab_first, nf, c->st);
		HIPCK(hipStreamSynchronize(c->st));
		HIPCK(hipFree(B.table));
		B.table = nt;
		if (B.tab_first) { HIPCK(hipFree(B.tab_first)); B.tab_first = nf; }
		if (ovf) {
			uint64_t *tmp = 0;
			HIPCK(hipMalloc(&tmp, ovf * 40));
			HIPCK(hipMemcpyAsync(tmp, B.tab_ovf, ovf * 40, hipMemcpyDeviceToDevice, c->st));
			HIPCK(hipMemsetAsync(&B.stats[(size_t)ST_SLOTS * ST_N], 0, 8, c->st));
			run_table_replay(P, B.table, tmp, ovf, B.stats, B.tab_ovf, B.tab_ovf_cap, B.tab_first, B.sub_last, c->st);
			if (fetch_stats(c) != 0) return -1;
			HIPCK(hipFree(tmp));
		} else c->h_stats[ST_TAB_OVF] = 0;
	}
}

// ---- region-owned table segments (KParams.seg)

// segment size the table should have now.  Upserts probe in LDS, so the segments run fuller than the table in the host's layout: growth
// at 62 % mean load (a region's load is Poisson around the mean: +5 sigma of 1000 keys is 16 %); a segment that fills up parks its k-mers.
static int seg_target_shift(const bfcg_ctx_t *c)
{
	const KParams &P = c->P;
	const uint64_t nfine = ((uint64_t)1 << P.F) >> c->log2n;
	// growing is one coalesced pass over the segments (k_seg_rehash: c3's 4 GiB in 0.9 ms), a batch into segments that are too full is not:
	// forecast with the larger of the last two batches' additions
	// (with the hand-over log the table changes commit by commit: what ONE more batch may add -- the largest page of the last commit, and what a
	// batch adds only shrinks -- must fit below the growth mark; how many batches a commit then takes is the window's business: handover_begin)
	const uint64_t need = c->h_stats[ST_KEYS] + c->h_stats[ST_TAB_OVF], g = c->ho_K > 1 ? c->keys_per_batch : c->grow[0] > c->grow[1] ? c->grow[0] : c->grow[1];
	int t = P.seg_shift;
	const double full = c->seg_no_grow ? 0.85 : 0.62;
	while ((double)(need + (c->seg_no_grow ? 0 : g)) > full * (double)(nfine << t)) ++t;
	if (c->h_stats[ST_TAB_OVF] && t == P.seg_shift) ++t; // one full segment under a low overall load
	if (t != P.seg_shift && getenv("BFCG_DEBUG")) fprintf(stderr, "[D::seg_target] shift %d -> %d: keys %llu parked %llu forecast %llu slots %llu\n", P.seg_shift, t,
		(unsigned long long)c->h_stats[ST_KEYS], (unsigned long long)c->h_stats[ST_TAB_OVF], (unsigned long long)g, (unsigned long long)(nfine << P.seg_shift));
	return t;
}

// grow the segments (k_seg_rehash streams every segment through LDS once) and replay parked k-mers until none is left; segments that
// would no longer fit a CU's LDS send the table to the host's layout for the rest of the run (seg_to_legacy)
static int seg_maintain(bfcg_ctx_t *c)
{
	KParams &P = c->P; BatchBufs &B = c->B;
	const uint64_t nfine = ((uint64_t)1 << P.F) >> c->log2n;
	const double t_dbg = dbg_now();
	struct dbg_guard { double t; int s0; const int *s1; ~dbg_guard() { if (getenv("BFCG_DEBUG") && *s1 != s0) fprintf(stderr, "[D::seg_maintain] shift %d -> %d: %.3f ms\n", s0, *s1, (dbg_now() - t) * 1e3); } } dg{t_dbg, P.seg_shift, &P.seg_shift};
	for (;;) {
		const uint64_t ovf = c->h_stats[ST_TAB_OVF];
		const int target = seg_target_shift(c);
		if (ovf == 0 && target == P.seg_shift) return 0;
		if (ovf > B.tab_ovf_cap) return set_err("count table overflow list exhausted (%llu parked k-mers)", (unsigned long long)ovf);
		{ // the old segments live until the new ones are filled.  No room (or no LDS: s > 14) for the next size: run fuller instead while nothing is
		  // parked and the mean load stays below 85 % (probing happens in LDS: a fuller segment costs probes, not HBM traffic) ...
			size_t free_b = 0, total_b = 0;
			const bool no_room = hipMemGetInfo(&free_b, &total_b) == hipSuccess && ((nfine << target) * 8) + (1ULL << 30) > (uint64_t)free_b + (c->seg_spare && c->seg_spare_shift >= target ? (nfine << c->seg_spare_shift) * 8 : 0);
			if (no_room || target > c->seg_total_max) { // (beyond 2^14 slots a segment is several blocks, one workgroup each: KParams.seg_blk)
				if (ovf == 0 && (double)c->h_stats[ST_KEYS] < 0.85 * (double)(nfine << P.seg_shift)) { c->seg_no_grow = 1; return 0; }
				if (target > c->seg_total_max) return seg_to_legacy(c); // ... else the host's layout takes over (random CAS upserts, any size)
				// no room for the next segment size, and too full (or k-mers parked) to run on as it is: the host's layout sizes its table to the memory
				// that IS free (seg_to_legacy) and may well succeed where doubling every segment cannot -- try it before giving up
				if (seg_to_legacy(c) == 0) return 0;
				return set_err("count table of %llu keys cannot grow to 2^%d slots per region: %.1f GiB of device memory free", (unsigned long long)c->h_stats[ST_KEYS], target, free_b / 1073741824.0);
			}
		}
		const int old_shift = P.seg_shift, old_blk = P.seg_blk;
		unsigned long long *nt = 0;
		int nt_shift = target;
		set_seg_shift(P, target, c->seg_blk_max);
		if (c->seg_spare && c->seg_spare_shift >= target) { nt = c->seg_spare; nt_shift = c->seg_spare_shift; c->seg_spare = 0; }
		else {
			if (c->seg_spare) { HIPCK(hipFree(c->seg_spare)); c->seg_spare = 0; }
			HIPCK(hipMalloc(&nt, (nfine << target) * 8));
		}
		run_seg_rehash(P, B.seg_tab, old_shift, old_blk, nt, (uint32_t)nfine, c->st);
		HIPCK(hipStreamSynchronize(c->st));
		HIPCK(hipGetLastError());
		if (((nfine << nt_shift) * 8) <= (16ULL << 30)) {
			// the buffer left behind becomes the spare the NEXT growth rehashes into; one that is smaller than its successor could not
			// take that growth after a reset: replace it now, so that from the second data set on a context never allocates
			// (a fresh multi-GiB hipMalloc costs ~35 ms per GiB on this box: only contexts that are being reused pay for the second buffer)
			if (c->seg_cap_shift < nt_shift && c->reused) { HIPCK(hipFree(B.seg_tab)); c->seg_spare = 0; if (hipMalloc(&c->seg_spare, (nfine << nt_shift) * 8) != hipSuccess) { (void)hipGetLastError(); c->seg_spare = 0; } c->seg_spare_shift = nt_shift; }
			else { c->seg_spare = B.seg_tab; c->seg_spare_shift = c->seg_cap_shift; }
		} else HIPCK(hipFree(B.seg_tab));
		B.seg_tab = nt; c->seg_cap_shift = nt_shift;
		++c->n_seg_grow;
		if (ovf) {
			uint64_t *tmp = 0;
			HIPCK(hipMalloc(&tmp, ovf * 40));
			HIPCK(hipMemcpyAsync(tmp, B.tab_ovf, ovf * 40, hipMemcpyDeviceToDevice, c->st));
			HIPCK(hipMemsetAsync(&B.stats[(size_t)ST_SLOTS * ST_N], 0, 8, c->st));
			run_seg_replay(P, B.seg_tab, tmp, ovf, B.stats, B.tab_ovf, B.tab_ovf_cap, c->st);
			if (fetch_stats(c) != 0) return -1;
			HIPCK(hipFree(tmp));
		} else c->h_stats[ST_TAB_OVF] = 0;
	}
}

// The table in the host's layout (2^l_pre sub-tables, htab.c:45-58) from the segments: for export, for the k-mer coverage kernels, and
// for runs whose segments outgrow LDS.  The device must be idle.  Afterwards the context counts on in that layout until bfcg_reset.
// for_export: the table is about to leave for the host (bfcg_export_table: normally the end of the count).  Every byte of it is allocated,
// cleared, copied over PCIe into fresh host pages and freed again -- 10 ms per GB each in the driver alone -- so it is sized for a load of
// at most 4/7 there, not for another batch's keys (a context that does count on grows it like any table: table_maintain); the export of a
// chr1-sized genome's 249 M keys moved 8 GiB before, 4 GiB now.
static int seg_to_legacy(bfcg_ctx_t *c, int for_export)
{
	KParams &P = c->P; BatchBufs &B = c->B;
	const uint64_t nfine = ((uint64_t)1 << P.F) >> c->log2n;
	const uint64_t keys = c->h_stats[ST_KEYS];
	if (c->seg_spare) { HIPCK(hipFree(c->seg_spare)); c->seg_spare = 0; }
	// a context that counts on in the host's layout needs k_bloom's aggregation buffer (1.6 GB at -b35, 6.4 GB at -b37): taken BEFORE the table is
	// sized to the memory that is free, not at the next batch when nothing may be left
	if (!for_export && !P.filter_mode && !B.agg_out) HIPCK(hipMalloc(&B.agg_out, nfine * P.ag_cap * (P.track ? 32 : 24)));
	int cs = c->prm.tab_cshift > 0 ? c->prm.tab_cshift : 2;
	const uint64_t want = for_export ? keys + keys * 3 / 4 : 2 * keys + (c->prm.max_batch_pos / 4);
	while ((1ULL << (P.l_pre + cs)) < want && P.l_pre + cs < 36) ++cs;
	{ // as much as memory allows (the segments live until the table is filled)
		size_t free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
			while (cs > 1 && (8ULL << (P.l_pre + cs)) + (2ULL << 30) > (uint64_t)free_b) --cs;
	}
	P.tab_cshift = cs;
	HIPCK(hipMalloc(&B.table, 8ULL << (P.l_pre + cs)));
	HIPCK(hipMemsetAsync(B.table, 0, 8ULL << (P.l_pre + cs), c->st));
	// the keys are counted afresh: k-mers the reference's lossy key cannot tell apart (k >= 38) become one key here
	HIPCK(hipMemset2DAsync(B.stats + ST_KEYS, sizeof(unsigned long long) * ST_N, 0, sizeof(unsigned long long), ST_SLOTS, c->st));
	run_seg_to_table(P, B.seg_tab, (uint32_t)nfine, B.table, B.stats, B.tab_ovf, B.tab_ovf_cap, c->st);
	HIPCK(hipStreamSynchronize(c->st));
	HIPCK(hipGetLastError());
	HIPCK(hipFree(B.seg_tab));
	B.seg_tab = 0;
	P.seg = 0; P.b3 = 0; c->seg_escaped = 1;
	HIPCK(set_bloom_lds_attr(P)); // the aggregation table is back in the bloom kernel's LDS footprint
	if (fetch_stats(c) != 0) return -1;
	c->keys_last = c->h_stats[ST_KEYS];
	if (for_export && c->h_stats[ST_TAB_OVF] == 0) return 0; // (no head room for a forecast batch: the next batch's check_health grows the table if there is one)
	return table_maintain(c);
}

// partition of the current run: out[0] 1 = one-pass level 1 (K1 once per batch) still in use, out[1] batches replayed through the two-pass partition
// since the context was created (a slab overflowed: few, often repeated k-mers)
extern "C" int bfcg_partition_info(bfcg_ctx_t *c, uint64_t out[2]) { out[0] = (uint64_t)((c->onepass ? 1 : 0) | ((c->onepass || (c->mg_op2 && c->mg_op2_allowed)) && c->cap2 ? 2 : 0)); out[1] = c->n_replayed; return 0; }
// A rank's stage B may partition what it received in one pass only if the caller keeps every receive buffer unchanged until the call after the
// next has returned (a slab overflow is found one call later and replayed from the buffer): bfcg_group_* alternates its buffers and says so here.
extern "C" void bfcg_mg_allow_onepass(bfcg_ctx_t *c, int on) { c->mg_op2_allowed = on != 0; }

extern "C" int bfcg_table_info(bfcg_ctx_t *c, int out[4])
{
	out[0] = c->P.seg; out[1] = c->P.seg ? c->P.seg_shift : 0; out[2] = c->P.seg ? 0 : c->P.tab_cshift; out[3] = (int)c->n_seg_grow;
	return 0;
}

// ---- multi-GPU (owner computes): stage A on every rank, exchange by the caller, stage B on the owner

extern "C" int bfcg_mg_info(bfcg_ctx_t *c, int out[4])
{
	out[0] = 1 << c->P.F1; out[1] = (1 << c->P.F1) >> c->log2n; out[2] = c->rw; out[3] = c->n_ranks;
	return 0;
}

// Stage A of a global batch on stream stA: it runs UNDER stage B of the previous batch (stream st), which bfcg_mg_process left
// running.  Returns when the records are in d_send and their per-bucket counts on the host; the caller may start the exchange.
static int mg_scatter2(bfcg_ctx_t *c, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t n_pos, void *d_send, uint32_t *counts, int no_kstats);
extern "C" int bfcg_mg_scatter(bfcg_ctx_t *c, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t n_pos, void *d_send, uint32_t *counts) { return mg_scatter2(c, d_seq, d_qual, n_pos, d_send, counts, 0); }
// (a batch whose one-pass stage A overflowed a slab, repeated: its k-mers were counted the first time)
extern "C" int bfcg_mg_scatter_again(bfcg_ctx_t *c, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t n_pos, void *d_send, uint32_t *counts) { return mg_scatter2(c, d_seq, d_qual, n_pos, d_send, counts, 1); }
// A group of ONE rank has no exchange for stage A of the next batch to run beside: its kernels follow stage B on the same stream wherever a
// single GPU's do (c->pipeline: kernels sized to fill the chip fight for its CUs -- DESIGN.md section 6b, c3 on two streams).
static inline hipStream_t mg_stage_a_stream(bfcg_ctx_t *c) { return c->n_ranks == 1 && !c->pipeline ? c->st : c->stA; }
static int mg_scatter2(bfcg_ctx_t *c, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t n_pos, void *d_send, uint32_t *counts, int no_kstats)
{
	const int nb1 = 1 << c->P.F1, b = c->cur;
	const hipStream_t sA = mg_stage_a_stream(c);
	if (n_pos > c->prm.max_batch_pos) return set_err("batch of %llu positions exceeds max_batch_pos=%llu", (unsigned long long)n_pos, (unsigned long long)c->prm.max_batch_pos);
	HIPCK(hipSetDevice(c->prm.device));
	if (n_pos == 0) { // nothing to contribute to this global batch: keep the timing events defined
		for (int i = 0; i < 3; ++i) HIPCK(hipEventRecord(c->evt[b][i], sA));
		memset(counts, 0, sizeof(uint32_t) * nb1); return 0;
	}
	BatchBufs Bt = c->B;
	Bt.rows1 = c->rows1[b]; Bt.chunk1 = c->chunk1[b]; Bt.start1 = c->start1[b]; Bt.row_base = Bt.start1 + nb1 + 1;
	if (same_block_offset(c, b, d_seq, &d_qual, n_pos, sA) != 0) return -1;
	KParams Pa = c->P;
	Pa.no_kstats = no_kstats;
	run_stage_a(Pa, Bt, d_seq, d_qual, (int64_t)n_pos, (uint64_t *)d_send, sA, c->evt[b]);
	HIPCK(hipGetLastError());
	uint32_t *tmp = (uint32_t *)malloc(sizeof(uint32_t) * (nb1 + 1));
	hipError_t e = hipMemcpyAsync(tmp, Bt.start1, sizeof(uint32_t) * (nb1 + 1), hipMemcpyDeviceToHost, sA);
	if (e == hipSuccess) e = hipStreamSynchronize(sA);
	if (e != hipSuccess) { free(tmp); return set_err("reading the level-1 bucket starts failed: %s", hipGetErrorString(e)); }
	for (int i = 0; i < nb1; ++i) counts[i] = tmp[i + 1] - tmp[i];
	free(tmp);
	return 0;
}

// Stage A of a global batch in ONE pass (round 4; bfcg_mg.hip's slab mode): K1 once, the records into 8 slabs of `slab capacity` records per
// level-1 bucket in d_send (bucket-major: a destination's buckets are one contiguous range of slabs, sent as it is -- the unfilled ends of the
// slabs travel too), except the rank's OWN buckets, whose slabs lie own_delta records further on: in the receive buffer the group put behind the
// send buffer, where this rank's block belongs -- no self-copy.  fills[8 * 2^F1]: records in every slab (bucket-major, XCD-minor), on the host
// when the call returns; *overflow: a slab was too small (skewed input) -- nothing of this batch may be used, the group falls back to
// bfcg_mg_scatter for it.  count_kmers = 0: a repeated stage A of a batch whose k-mers were counted already.
extern "C" int bfcg_mg_slab_info(bfcg_ctx_t *c, uint32_t out[2]) { out[0] = c->op_cap; out[1] = (uint32_t)(c->mg_slab_ok && c->op_cursor[0] != 0); return 0; }
extern "C" int bfcg_mg_scatter_slabs(bfcg_ctx_t *c, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t n_pos, void *d_send, uint32_t own_delta, uint32_t *fills, int *overflow)
{
	const int nb1 = 1 << c->P.F1, nb_loc = nb1 >> c->log2n, b = c->cur;
	const hipStream_t sA = mg_stage_a_stream(c);
	*overflow = 0;
	if (!c->mg_slab_ok || !c->op_cursor[b]) return set_err("this context has no slab mode");
	if (n_pos > c->prm.max_batch_pos) return set_err("batch of %llu positions exceeds max_batch_pos=%llu", (unsigned long long)n_pos, (unsigned long long)c->prm.max_batch_pos);
	HIPCK(hipSetDevice(c->prm.device));
	if (n_pos == 0) { // nothing to contribute to this global batch: keep the timing events defined
		for (int i = 0; i < 3; ++i) HIPCK(hipEventRecord(c->evt[b][i], sA));
		memset(fills, 0, sizeof(uint32_t) * (size_t)nb1 * 8); return 0;
	}
	BatchBufs Bt = c->B;
	Bt.op_cursor = c->op_cursor[b]; Bt.op_seg = c->op_seg[b]; Bt.op_flags = c->op_flags + 4 * b; Bt.op_cap = c->op_cap; Bt.cap2 = c->cap2;
	Bt.op_own_lo = (uint32_t)c->rank * (uint32_t)nb_loc; Bt.op_own_n = (uint32_t)nb_loc; Bt.op_own_delta = own_delta;
	if (same_block_offset(c, b, d_seq, &d_qual, n_pos, sA) != 0) return -1;
	run_stage_a_onepass(c->P, Bt, d_seq, d_qual, (int64_t)n_pos, (uint64_t *)d_send, sA, c->evt[b]);
	HIPCK(hipGetLastError());
	std::vector<uint32_t> tmp((size_t)8 * nb1);
	hipError_t e = hipMemcpyAsync(tmp.data(), Bt.op_seg + (size_t)8 * nb1, sizeof(uint32_t) * (size_t)8 * nb1, hipMemcpyDeviceToHost, sA); // the segments' ends (k_seg_setup)
	if (e == hipSuccess) e = hipMemcpyAsync(c->h_flags[b], Bt.op_flags, sizeof(uint32_t), hipMemcpyDeviceToHost, sA);
	if (e == hipSuccess) e = hipStreamSynchronize(sA);
	if (e != hipSuccess) return set_err("reading the slabs' fill failed: %s", hipGetErrorString(e));
	*overflow = c->h_flags[b][0] != 0;
	for (int seg = 0; seg < 8 * nb1; ++seg) fills[seg] = *overflow ? 0u : tmp[seg] - (uint32_t)seg * c->op_cap; // (k_seg_setup: the slab of segment seg starts at seg x capacity)
	return 0;
}

// The same stage A WITHOUT the host's wait (round 5: a group whose ranks are all in one process keeps the host out of the batch's loop).  The fills
// stay on the device: k_pack_rows lays them out as one row per destination in d_rows_out (n_ranks rows of row_w = nb_loc x 8 + 2 words: the fills of
// the destination's slabs, then this stage A's overflow flag), which the caller sends beside the blocks; a copy of the rows is on its way to the
// host.  bfcg_mg_scatter_slabs_wait -- called when the exchange and the owner's stage B are already enqueued -- waits for that copy and hands
// out what bfcg_mg_scatter_slabs would have.  *done: recorded on stage A's stream behind all of it (the exchange waits for it).
extern "C" int bfcg_mg_row_words(bfcg_ctx_t *c) { return (((1 << c->P.F1) >> c->log2n) * 8) + 2; }
extern "C" int bfcg_mg_scatter_slabs_async(bfcg_ctx_t *c, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t n_pos, void *d_send, uint32_t own_delta, uint32_t *d_rows_out, hipEvent_t *done)
{
	const int nb1 = 1 << c->P.F1, nb_loc = nb1 >> c->log2n, b = c->cur, row_w = nb_loc * 8 + 2;
	const hipStream_t sA = mg_stage_a_stream(c);
	if (!c->mg_slab_ok || !c->op_cursor[b] || !c->h_rows[b]) return set_err("this context has no slab mode");
	if (n_pos > c->prm.max_batch_pos) return set_err("batch of %llu positions exceeds max_batch_pos=%llu", (unsigned long long)n_pos, (unsigned long long)c->prm.max_batch_pos);
	HIPCK(hipSetDevice(c->prm.device));
	const size_t row_bytes = sizeof(uint32_t) * (size_t)c->n_ranks * row_w;
	if (n_pos == 0) { // nothing to contribute to this global batch: rows of zeros; keep the timing events defined
		for (int i = 0; i < 3; ++i) HIPCK(hipEventRecord(c->evt[b][i], sA));
		HIPCK(hipMemsetAsync(d_rows_out, 0, row_bytes, sA));
	} else {
		BatchBufs Bt = c->B;
		Bt.op_cursor = c->op_cursor[b]; Bt.op_seg = c->op_seg[b]; Bt.op_flags = c->op_flags + 4 * b; Bt.op_cap = c->op_cap; Bt.cap2 = c->cap2;
		Bt.op_own_lo = (uint32_t)c->rank * (uint32_t)nb_loc; Bt.op_own_n = (uint32_t)nb_loc; Bt.op_own_delta = own_delta;
		if (same_block_offset(c, b, d_seq, &d_qual, n_pos, sA) != 0) return -1;
		run_stage_a_onepass(c->P, Bt, d_seq, d_qual, (int64_t)n_pos, (uint64_t *)d_send, sA, c->evt[b]);
		run_pack_rows(c->P, Bt, c->n_ranks, (uint32_t)row_w, d_rows_out, sA);
		HIPCK(hipGetLastError());
	}
	HIPCK(hipMemcpyAsync(c->h_rows[b], d_rows_out, row_bytes, hipMemcpyDeviceToHost, sA));
	HIPCK(hipEventRecord(c->evA[b], sA));
	c->rows_slot = b;
	*done = c->evA[b];
	return 0;
}
// fills[8 * 2^F1] (bucket-major, XCD-minor, as bfcg_mg_scatter_slabs) and *overflow of the stage A enqueued last
extern "C" int bfcg_mg_scatter_slabs_wait(bfcg_ctx_t *c, uint32_t *fills, int *overflow)
{
	const int nb1 = 1 << c->P.F1, nb_loc = nb1 >> c->log2n, b = c->rows_slot, row_w = nb_loc * 8 + 2;
	HIPCK(hipSetDevice(c->prm.device));
	HIPCK(hipEventSynchronize(c->evA[b]));
	int ovf = 0;
	for (int p = 0; p < c->n_ranks; ++p) ovf |= c->h_rows[b][(size_t)p * row_w + (size_t)nb_loc * 8] != 0;
	for (int p = 0; p < c->n_ranks; ++p)
		for (int i = 0; i < nb_loc * 8; ++i) fills[(size_t)p * nb_loc * 8 + i] = ovf ? 0u : c->h_rows[b][(size_t)p * row_w + i];
	*overflow = ovf;
	return 0;
}

// d_recv: records for the owned level-1 buckets, source-major (rank 0's block, rank 1's block, ...), inside each block
// grouped by bucket; seg_cnt[s * nb_loc + b] = records from source s for owned bucket b.
// Stage B is ENQUEUED (stream st) and left running: the call returns once the PREVIOUS batch is finalised (statistics read,
// table maintained), so the caller's next bfcg_mg_scatter and exchange overlap with it.  d_recv must stay untouched until the
// next bfcg_mg_process (or bfcg_sync) returns: callers alternate between two receive buffers.
// slab_cap != 0 (bfcg_mg_process_slabs): d_recv holds, source-major, every source's SLABS for the owned buckets (8 per bucket, slab_cap records
// each, at fixed places) and seg_cnt[(s * nb_loc + b) * 8 + x] says how far each is filled.
static int mg_process_any(bfcg_ctx_t *c, const void *d_recv, const uint32_t *seg_cnt, uint32_t slab_cap, hipEvent_t *wait, int n_wait);
extern "C" int bfcg_mg_process_ev(bfcg_ctx_t *c, const void *d_recv, const uint32_t *seg_cnt, hipEvent_t *wait, int n_wait) { return mg_process_any(c, d_recv, seg_cnt, 0u, wait, n_wait); }
extern "C" int bfcg_mg_process(bfcg_ctx_t *c, const void *d_recv, const uint32_t *seg_cnt) { return mg_process_any(c, d_recv, seg_cnt, 0u, 0, 0); }
extern "C" int bfcg_mg_process_slabs(bfcg_ctx_t *c, const void *d_recv, const uint32_t *fills, uint32_t slab_cap, hipEvent_t *wait, int n_wait) { return mg_process_any(c, d_recv, fills, slab_cap, wait, n_wait); }
// (stage B ordered behind `wait[0..n_wait)`: events of the exchange that fills d_recv, on other streams / devices -- no host wait)
static int mg_process_any(bfcg_ctx_t *c, const void *d_recv, const uint32_t *seg_cnt, uint32_t slab_cap, hipEvent_t *wait, int n_wait)
{
	const int N = c->n_ranks, nb_loc = (1 << c->P.F1) >> c->log2n, spb = slab_cap ? N * 8 : N, n_seg = nb_loc * spb, b = c->cur;
	HIPCK(hipSetDevice(c->prm.device));
	const size_t words = (size_t)3 * n_seg + 1 + nb_loc + 1;
	if (words > c->seg_words) return set_err("internal: %zu segment words, room for %zu", words, c->seg_words);
	uint32_t *seg_beg = c->h_seg + (size_t)b * c->seg_words, *seg_end = seg_beg + n_seg, *row_base = seg_end + n_seg, *bucket_start = row_base + n_seg + 1;
	uint64_t off = 0, rows = 0, tot = 0;
	if (slab_cap) {
		for (int s = 0; s < N; ++s)
			for (int k = 0; k < nb_loc; ++k)
				for (int x = 0; x < 8; ++x) { // slab (source s, bucket k, XCD x) sits at its fixed place
					const int seg = (k * N + s) * 8 + x;
					const uint64_t at = ((uint64_t)((size_t)s * nb_loc + k) * 8 + x) * slab_cap;
					const uint32_t fill = seg_cnt[((size_t)s * nb_loc + k) * 8 + x];
					if (fill > slab_cap || at + fill > 0xffffffffULL) return set_err("slab of %u records holds %u", slab_cap, fill);
					seg_beg[seg] = (uint32_t)at; seg_end[seg] = (uint32_t)at + fill; off += fill;
				}
	} else
	for (int s = 0; s < N; ++s)
		for (int k = 0; k < nb_loc; ++k) { // position of (source s, bucket k) in the receive buffer
			int seg = k * N + s;
			seg_beg[seg] = (uint32_t)off; off += seg_cnt[s * nb_loc + k]; seg_end[seg] = (uint32_t)off;
		}
	if (off > c->recv_cap) return set_err("received %llu records for this rank's buckets, capacity %llu", (unsigned long long)off, (unsigned long long)c->recv_cap);
	const uint32_t tile2 = (uint32_t)bfcg_tile_of_rw(c->rw / 4);
	for (int seg = 0; seg < n_seg; ++seg) { row_base[seg] = (uint32_t)rows; rows += (seg_end[seg] - seg_beg[seg] + tile2 - 1) / tile2; }
	row_base[n_seg] = (uint32_t)rows;
	for (int k = 0; k < nb_loc; ++k) {
		bucket_start[k] = (uint32_t)tot;
		for (int s = 0; s < spb; ++s) tot += seg_end[k * spb + s] - seg_beg[k * spb + s];
	}
	bucket_start[nb_loc] = (uint32_t)tot;
	uint32_t *d = c->d_seg + (size_t)b * c->seg_words;
	for (int i = 0; i < n_wait; ++i) HIPCK(hipStreamWaitEvent(c->st, wait[i], 0));
	HIPCK(hipEventRecord(c->evt[b][6], c->st));
	HIPCK(hipMemcpyAsync(d, seg_beg, sizeof(uint32_t) * words, hipMemcpyHostToDevice, c->st));
	if (c->B.seen_out) HIPCK(hipMemsetAsync(c->B.seen_out, 0, c->prm.max_batch_pos, c->st));
	c->B.batch_hi = (unsigned long long)(c->n_batches + 1) << 32;
	if (use_stream(c) != 0 || ensure_agg(c) != 0) return -1;
	c->B.stream = c->stream_mode; c->B.stream_out = c->stream_out;
	BatchBufs Bt = c->B;
	const int op2_run = c->mg_op2 && c->mg_op2_allowed && c->cap2; // (as enqueue_batch: every batch of a one-pass run tests the sticky word and is queued)
	const int op2 = op2_run && off >= (uint64_t)8 << c->P.F >> c->log2n; // (a handful of records per region: two passes)
	if (op2_run) Bt.op_sticky = c->op_flags + OP_STICKY;
	if (op2) {
		Bt.cnt2 = c->cnt2; Bt.cap2 = c->cap2; Bt.op_flags = c->op_flags + 4 * b;
		HIPCK(hipMemsetAsync(Bt.op_flags, 0, 4 * sizeof(uint32_t), c->st)); // (no one-pass stage A on a rank: stage B clears its slot's flags itself)
	}
	if (handover_begin(c, Bt, b, op2, c->call_no + 1) != 0) return -1; // (this call's number: assigned below)
	KParams Pm = c->P;
	Pm.dedupe = dedupe_hint(c);
	warm_tables(c, Pm);
	run_stage_b(Pm, Bt, (const uint64_t *)d_recv, d, d + n_seg, n_seg, spb, d + 2 * n_seg, d + 3 * n_seg + 1, off, c->st, c->evt[b]);
	if (handover_end(c, Bt, b) != 0) return -1;
	if (op2_run) { // was every region's slab large enough?  (as for a single GPU: the flag is read with the batch's snapshot, an overflow is replayed -- stage B only)
		HIPCK(hipMemcpyAsync(c->h_flags[b] + 1, c->op_flags + OP_STICKY, sizeof(uint32_t), hipMemcpyDeviceToHost, c->st));
		if (c->n_opq == 4) return set_err("internal: one-pass queue overflow");
		int free_i = 0;
		for (;; ++free_i) { int used = 0; for (int i = 0; i < c->n_opq; ++i) used |= c->opq[i].mg == free_i + 1; if (!used) break; }
		memcpy(c->mg_seg[free_i], seg_cnt, sizeof(uint32_t) * (size_t)N * nb_loc * (slab_cap ? 8 : 1)); c->mg_seg_slab[free_i] = slab_cap != 0;
		bfcg_ctx::opq_t &q = c->opq[c->n_opq++];
		q.seq = q.qual = nullptr; q.n_pos = off; q.slot = b; q.recv = d_recv; q.mg = free_i + 1;
	}
	HIPCK(hipMemcpyAsync(c->h_snap[b], c->B.stats, sizeof(unsigned long long) * ST_N * (ST_SLOTS + 1), hipMemcpyDeviceToHost, c->st));
	c->slot_call[b] = ++c->call_no; c->slot_pos[b] = off;
	HIPCK(hipEventRecord(c->evB[b], c->st));
	HIPCK(hipGetLastError());
	c->used[b] = 1;
	++c->n_batches;
	return finalise_previous(c, b);
}

// Stage B of a global batch whose sizes the host does not have yet (slab mode, all ranks in one process): d_rows_in holds every source's row for
// this rank (n_ranks rows of bfcg_mg_row_words words, written by the exchange that `wait` orders this behind); k_seg_setup_mg turns them into the
// segment arrays on the device.  rec_bound: what the batch can hold at most (sizes the level-2 grid; surplus workgroups exit).  The batch is
// enqueued and NOT finalised: bfcg_mg_process_finish does that once the caller has the sizes (the same call of the group, a moment later).
// may this rank take batches that way?  A property of the context's parameters, not of its state (no order stamps, no per-position debug output):
// the processes of a multi-process group must all answer alike.  (Level 2 in one pass or two: the slabs a source can fill -- nb1 x 8 x capacity
// records -- are fewer than the level-2 buffers hold, so the bound sizes the grids and nothing can be overrun whatever the sizes turn out to be.)
extern "C" int bfcg_mg_async_ok(bfcg_ctx_t *c) { return c->mg_slab_ok && !c->B.seen_out && !c->P.track && c->h_rows[0] != 0 && (c->n_ranks == 1 || (uint64_t)8 * (1u << c->P.F1) * c->op_cap <= c->recv_cap); } // (one rank: what it "receives" is its own batch, <= max_batch_pos = its capacity)
// (only the sources [s_lo, s_hi) of the receive buffer: an overloaded owner applies the others in further passes, bfcg_mg_process_slabs with their sizes)
extern "C" int bfcg_mg_process_slabs_dev(bfcg_ctx_t *c, const void *d_recv, const uint32_t *d_rows_in, uint32_t slab_cap, uint64_t rec_bound, int s_lo, int s_hi, hipEvent_t *wait, int n_wait)
{
	const int N = c->n_ranks, nb_loc = (1 << c->P.F1) >> c->log2n, spb = N * 8, n_seg = nb_loc * spb, b = c->cur;
	HIPCK(hipSetDevice(c->prm.device));
	const size_t words = (size_t)3 * n_seg + 1 + nb_loc + 1;
	if (words > c->seg_words) return set_err("internal: %zu segment words, room for %zu", words, c->seg_words);
	if (!slab_cap || !bfcg_mg_async_ok(c)) return set_err("internal: bfcg_mg_process_slabs_dev needs slabs and no debug_seen");
	if (rec_bound > c->recv_cap) rec_bound = c->recv_cap;
	uint32_t *d = c->d_seg + (size_t)b * c->seg_words;
	for (int i = 0; i < n_wait; ++i) HIPCK(hipStreamWaitEvent(c->st, wait[i], 0));
	HIPCK(hipEventRecord(c->evt[b][6], c->st));
	run_seg_setup_mg(c->P, c->rw / 4, d_rows_in, (uint32_t)(nb_loc * 8 + 2), N, s_lo, s_hi, slab_cap, d, nullptr, c->st);
	c->B.batch_hi = (unsigned long long)(c->n_batches + 1) << 32;
	if (use_stream(c) != 0 || ensure_agg(c) != 0) return -1;
	c->B.stream = c->stream_mode; c->B.stream_out = c->stream_out;
	BatchBufs Bt = c->B;
	const int op2_run = c->mg_op2 && c->mg_op2_allowed && c->cap2;
	const int op2 = op2_run && rec_bound >= (uint64_t)8 << c->P.F >> c->log2n;
	if (op2_run) Bt.op_sticky = c->op_flags + OP_STICKY;
	if (op2) {
		Bt.cnt2 = c->cnt2; Bt.cap2 = c->cap2; Bt.op_flags = c->op_flags + 4 * b;
		HIPCK(hipMemsetAsync(Bt.op_flags, 0, 4 * sizeof(uint32_t), c->st));
	}
	if (handover_begin(c, Bt, b, op2, c->call_no + 1) != 0) return -1;
	KParams Pm = c->P;
	Pm.dedupe = dedupe_hint(c);
	warm_tables(c, Pm);
	run_stage_b(Pm, Bt, (const uint64_t *)d_recv, d, d + n_seg, n_seg, spb, d + 2 * n_seg, d + 3 * n_seg + 1, rec_bound, c->st, c->evt[b]);
	if (handover_end(c, Bt, b) != 0) return -1;
	if (op2_run) {
		HIPCK(hipMemcpyAsync(c->h_flags[b] + 1, c->op_flags + OP_STICKY, sizeof(uint32_t), hipMemcpyDeviceToHost, c->st));
		if (c->n_opq == 4) return set_err("internal: one-pass queue overflow");
		int free_i = 0;
		for (;; ++free_i) { int used = 0; for (int i = 0; i < c->n_opq; ++i) used |= c->opq[i].mg == free_i + 1; if (!used) break; }
		memset(c->mg_seg[free_i], 0, sizeof(uint32_t) * (size_t)N * nb_loc * 8); c->mg_seg_slab[free_i] = 1; // (the sizes follow: bfcg_mg_process_finish)
		bfcg_ctx::opq_t &q = c->opq[c->n_opq++];
		q.seq = q.qual = nullptr; q.n_pos = rec_bound; q.slot = b; q.recv = d_recv; q.mg = free_i + 1;
		c->unfinished_mg = free_i + 1;
	} else c->unfinished_mg = 0;
	HIPCK(hipMemcpyAsync(c->h_snap[b], c->B.stats, sizeof(unsigned long long) * ST_N * (ST_SLOTS + 1), hipMemcpyDeviceToHost, c->st));
	c->slot_call[b] = ++c->call_no; c->slot_pos[b] = rec_bound;
	HIPCK(hipEventRecord(c->evB[b], c->st));
	HIPCK(hipGetLastError());
	c->used[b] = 1;
	++c->n_batches;
	c->unfinished = 1;
	return 0;
}
// ... and the sizes have arrived on the host: fills[(s * nb_loc + k) * 8 + x] = records from source s in the slab of (owned bucket k, XCD x) -- kept
// for a replay of this stage B (a region's slab may turn out too small one call later) --, then the batch BEFORE this one is finalised as
// bfcg_mg_process_slabs does.
extern "C" int bfcg_mg_process_finish(bfcg_ctx_t *c, const uint32_t *fills)
{
	if (!c->unfinished) return set_err("internal: no stage B waits for its sizes");
	const int N = c->n_ranks, nb_loc = (1 << c->P.F1) >> c->log2n, b = c->cur;
	c->unfinished = 0;
	uint64_t tot = 0;
	for (size_t i = 0; i < (size_t)N * nb_loc * 8; ++i) tot += fills[i];
	c->slot_pos[b] = tot;
	if (c->unfinished_mg) {
		memcpy(c->mg_seg[c->unfinished_mg - 1], fills, sizeof(uint32_t) * (size_t)N * nb_loc * 8);
		for (int i = 0; i < c->n_opq; ++i) if (c->opq[i].mg == c->unfinished_mg) c->opq[i].n_pos = tot;
		c->unfinished_mg = 0;
	}
	return finalise_previous(c, b);
}

// the hand-over buffer of the STREAM mode is allocated when the mode is first used
static int use_stream(bfcg_ctx_t *c)
{
	if (c->stream_mode) {
		if (!c->P.seg && !c->stream_out) HIPCK(hipMalloc(&c->stream_out, c->recs2_n * (uint64_t)c->rw)); // (the segment layout hands over through its log)
		++c->n_stream_batches;
	}
	return 0;
}

// batch b has just been enqueued: finalise the one before it while b runs
static int finalise_previous(bfcg_ctx_t *c, int b)
{
	if (c->pend) {
		const int pb = b ^ 1;
		HIPCK(hipEventSynchronize(c->evB[pb]));
		if (batch_times(c, pb) != 0) return -1;
		if (c->n_opq && c->opq[0].slot == pb) { // the batch just finished went through the one-pass partition: clean?
			if (c->h_flags[pb][1]) { c->pend = 1; c->cur = b ^ 1; return drain(c); } // a slab overflowed: drain() replays it and what was enqueued behind it
			for (int i = 1; i < c->n_opq; ++i) c->opq[i - 1] = c->opq[i];
			--c->n_opq;
		}
		fold_snapshot(c, pb); // the counters as they stood at the end of that batch's stage B (k-mer / high counts may include the next batch's stage A)
		note_growth(c);
		if (c->h_stats[ST_ERR_POOL] || c->h_stats[ST_TAB_OVF] || (c->P.seg ? seg_target_shift(c) != c->P.seg_shift : (c->B.table && table_target_cshift(c) != c->P.tab_cshift))) {
			c->pend = 1; c->cur = b ^ 1; // make drain() see the batch just enqueued
			return drain(c);
		}
	}
	c->pend = 1; c->cur = b ^ 1;
	return 0;
}

// ---- hand-over log (region-owned table segments)

// the pages filled so far -> the segments, as a launch of its own on stream st (before a batch that cannot use the log; when the pipeline is
// drained); the pages' key counts go to the pinned buffer of slot b
static int commit_pending_pages(bfcg_ctx_t *c, int b)
{
	if (!c->ho_pending) return 0;
	const uint32_t nfine = (uint32_t)(((uint64_t)1 << c->P.F) >> c->log2n);
	BatchBufs Bt = c->B;
	Bt.ho = c->ho; Bt.ho_stride = c->ho_stride; Bt.ho_cur = c->ho_cur; Bt.ho_mark = c->ho_mark; Bt.ho_mark_stride = nfine; Bt.ho_keys = c->ho_keys;
	HIPCK(hipMemsetAsync(c->ho_keys, 0, sizeof(unsigned long long) * ST_SLOTS * (size_t)c->ho_pending, c->st));
	run_commit_pages(c->P, Bt, nfine, (uint32_t)c->ho_pending, c->st);
	HIPCK(hipMemcpyAsync(c->h_ho_keys[b], c->ho_keys, sizeof(unsigned long long) * ST_SLOTS * (size_t)c->ho_pending, hipMemcpyDeviceToHost, c->st));
	c->slot_commit[b] = c->ho_pending;
	memcpy(c->slot_page_call[b], c->ho_page_call, sizeof(uint64_t) * (size_t)c->ho_pending);
	c->ho_pending = 0;
	return 0;
}
// Stage B of the batch in slot b is about to be enqueued: where do its seen k-mers go?  `slabs`: its level 2 runs in one pass, so a region's
// share is bounded by its slab and the batch takes a page of the log; else it hands over at its records' offsets and is applied at once
// (what waits in the log goes first: the two share the arena).
static int handover_begin(bfcg_ctx_t *c, BatchBufs &Bt, int b, int slabs, uint64_t call)
{
	c->slot_commit[b] = 0; c->slot_all_committed[b] = 1;
	if (!c->P.seg) return 0;
	const uint32_t nfine = (uint32_t)(((uint64_t)1 << c->P.F) >> c->log2n);
	Bt.stream_out = (uint32_t *)c->ho;
	Bt.ho = c->ho; Bt.ho_cur = c->ho_cur; Bt.ho_mark = c->ho_mark; Bt.ho_mark_stride = nfine; Bt.ho_keys = c->ho_keys;
	if (!slabs || c->ho_stride == 0 || Bt.seen_out) { // (debug_seen contexts run one batch at a time anyway)
		if (commit_pending_pages(c, b) != 0) return -1;
		Bt.ho_stride = 0; Bt.ho_page = 0; Bt.ho_commit = 1; Bt.ho_keys = nullptr;
		return 0;
	}
	if (c->ho_pending == 0) { // a new window: as many batches as the segments have room for, judged by what the last commit's batches created
		// (a young table grows by most of a batch's k-mers: applied batch by batch, and grown in between, until the keys level off)
		int w = 1;
		if (c->ho_K > 1 && c->n_commits > 0) {
			// (upserts probe in LDS: a window may fill the segments to 85 % on average; what a batch adds only shrinks as the keys level off, so
			// the last commit's largest page bounds the next batches'; a segment that fills up all the same parks its k-mers, exactly)
			const double room = 0.85 * (double)((uint64_t)nfine << c->P.seg_shift) - (double)c->keys_known, g = 1.1 * (double)(c->keys_per_batch ? c->keys_per_batch : 1);
			w = room <= g ? 1 : room / g >= (double)c->ho_K ? c->ho_K : (int)(room / g);
		}
		c->ho_window = w;
		if (getenv("BFCG_DEBUG")) fprintf(stderr, "[D::window] %d batches: keys %llu, largest page of the last commit %llu, slots %llu, commits so far %d\n", w,
			(unsigned long long)c->keys_known, (unsigned long long)c->keys_per_batch, (unsigned long long)((uint64_t)nfine << c->P.seg_shift), c->n_commits);
	}
	Bt.ho_stride = c->ho_stride; Bt.ho_page = (uint32_t)c->ho_pending;
	c->ho_page_call[c->ho_pending++] = call;
	Bt.ho_commit = c->ho_pending >= c->ho_window;
	if (Bt.ho_commit) HIPCK(hipMemsetAsync(c->ho_keys, 0, sizeof(unsigned long long) * ST_SLOTS * (size_t)c->ho_pending, c->st));
	else c->slot_all_committed[b] = 0;
	return 0;
}
// ... and it has been enqueued: the key counts of a commit it carried follow it to the host
static int handover_end(bfcg_ctx_t *c, const BatchBufs &Bt, int b)
{
	if (!c->P.seg || Bt.ho_stride == 0 || !Bt.ho_commit) return 0;
	HIPCK(hipMemcpyAsync(c->h_ho_keys[b], c->ho_keys, sizeof(unsigned long long) * ST_SLOTS * (size_t)c->ho_pending, hipMemcpyDeviceToHost, c->st));
	c->slot_commit[b] = c->ho_pending;
	memcpy(c->slot_page_call[b], c->ho_page_call, sizeof(uint64_t) * (size_t)c->ho_pending);
	c->ho_pending = 0;
	return 0;
}
// the key counts of the pages committed with slot b's batch have arrived (its stage B is complete): distinct keys after every page's batch
static void absorb_pages(bfcg_ctx_t *c, int b)
{
	const int J = c->slot_commit[b];
	if (!J) return;
	uint64_t run = c->keys_known, most = 0;
	for (int j = 0; j < J; ++j) {
		const unsigned long long *row = c->h_ho_keys[b] + (size_t)j * ST_SLOTS;
		uint64_t d = 0;
		for (int i = 0; i < ST_SLOTS; ++i) d += row[i];
		run += d;
		if (d > most) most = d;
		c->call_keys[c->slot_page_call[b][j] & 63] = run;
	}
	c->keys_per_batch = most; ++c->n_commits;
	c->final_call = c->slot_page_call[b][J - 1];
	c->keys_known = run;
	c->slot_commit[b] = 0;
	c->commit_absorbed = 1;
}

// k_scatter1 reads both streams as aligned 16-byte blocks and wants them at the SAME offset inside a block (any offset: sub-batches start
// where a read ends).  Streams that differ there -- pointers a caller cut differently -- get their qualities copied to the slot's staging
// buffer at the sequence's offset.
static int same_block_offset(bfcg_ctx_t *c, int b, const uint8_t *d_seq, const uint8_t **d_qual, uint64_t n_pos, hipStream_t s)
{
	if (!*d_qual || ((((uintptr_t)d_seq) ^ ((uintptr_t)*d_qual)) & 15) == 0) return 0;
	uint8_t *dst = c->d_qual2[b] + (((uintptr_t)d_seq) & 15);
	HIPCK(hipMemcpyAsync(dst, *d_qual, n_pos, hipMemcpyDeviceToDevice, s));
	*d_qual = dst;
	return 0;
}

// One batch, software-pipelined over two streams: stage A of this batch (ALU-bound K1) is enqueued on stA and runs
// under stage B of the previous batch (LDS/latency-bound) on st.  The call returns once the PREVIOUS batch is
// finalised (statistics read, table maintained); bfcg_sync / bfcg_stats / exports drain the pipeline.
static int ensure_agg(bfcg_ctx_t *c)
{
	if (c->P.filter_mode || c->P.seg || c->B.agg_out) return 0;
	const uint64_t nfine = ((uint64_t)1 << c->P.F) >> c->log2n;
	HIPCK(hipMalloc(&c->B.agg_out, nfine * c->P.ag_cap * (c->P.track ? 32 : 24)));
	return 0;
}

static int enqueue_batch(bfcg_ctx_t *c, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t n_pos, int wait_copy, int no_kstats)
{
	const int b = c->cur, nb1 = 1 << c->P.F1;
	if (ensure_agg(c) != 0) return -1;
	BatchBufs Bt = c->B;
	Bt.batch_hi = (unsigned long long)(c->n_batches + 1) << 32;
	Bt.rows1 = c->rows1[b]; Bt.chunk1 = c->chunk1[b]; Bt.start1 = c->start1[b]; Bt.row_base = Bt.start1 + nb1 + 1; Bt.recs1 = c->recs1[b];
	hipStream_t sA = c->pipeline ? c->stA : c->st; // one stream: the batches' kernels simply follow each other
	if (c->used[b]) HIPCK(hipStreamWaitEvent(sA, c->evB[b], 0)); // stage B two batches ago has released this buffer set
	if (!c->pipeline && wait_copy) HIPCK(hipStreamWaitEvent(sA, c->evCopy, 0)); // the host batch is being copied on stream stA
	if (use_stream(c) != 0) return -1;
	Bt.stream = c->stream_mode; Bt.stream_out = c->stream_out;
	KParams Pt = c->P;
	Pt.no_kstats = no_kstats;
	Pt.dedupe = dedupe_hint(c);
	warm_tables(c, Pt);
	const uint8_t *const d_qual_given = d_qual; // (what a replay starts from again)
	if (same_block_offset(c, b, d_seq, &d_qual, n_pos, sA) != 0) return -1;
	const int op_run = c->onepass && !no_kstats;  // the run still uses the one-pass partition: an earlier batch may turn out to have overflowed a slab
	const int op = op_run && n_pos >= c->op_min_pos; // (a handful of tiles cannot fill 8 slabs per bucket evenly: such a batch takes two passes)
	// every batch of such a run -- one-pass or not -- tests the sticky word in its stage B and waits in the queue until it is known to be clean:
	// a small two-pass batch enqueued behind a poisoned one must not be applied before the replay of that one
	if (op_run) Bt.op_sticky = c->op_flags + OP_STICKY;
	if (op) {
		Bt.op_cursor = c->op_cursor[b]; Bt.op_seg = c->op_seg[b]; Bt.op_flags = c->op_flags + 4 * b; Bt.op_cap = c->op_cap;
		Bt.cnt2 = c->cnt2; Bt.cap2 = c->cap2;
		{ // level 2 on tiles of 8192 records where a tile of 4096 would leave as runs of four (2^10 regions per bucket): BFCG_L2_BIG=2; =1: from 2^9 on.  OFF by default:
		  // measured level with the small tile on c4e (64.9 against 62.3 ms per step: DESIGN 6b) -- the kernel's rate there is not set by the run length
			const char *e = getenv("BFCG_L2_BIG");
			const int want = e ? atoi(e) : 0;
			Pt.l2_big = c->rw == 12 && c->cap2 && want > 0 && Pt.F2 >= (want == 1 ? 9 : 10);
		}
		run_stage_a_onepass(Pt, Bt, d_seq, d_qual, (int64_t)n_pos, Bt.recs1, sA, c->evt[b]);
	} else run_stage_a(Pt, Bt, d_seq, d_qual, (int64_t)n_pos, Bt.recs1, sA, c->evt[b]);
	HIPCK(hipEventRecord(c->evA[b], sA));
	HIPCK(hipStreamWaitEvent(c->st, c->evA[b], 0));
	HIPCK(hipEventRecord(c->evt[b][6], c->st));
	if (c->B.seen_out) HIPCK(hipMemsetAsync(c->B.seen_out, 0, n_pos, c->st));
	if (handover_begin(c, Bt, b, op && Bt.cap2, c->call_no) != 0) return -1;
	if (op) {
		uint32_t *sg = c->op_seg[b];
		// the bound on level 2's rows is the slabs' CAPACITY, not the batch's positions: a segment is a slab's fill, and that holds the dead records of
		// everything level 1 reserved and did not use -- per (workgroup, bucket) a group and a half and a padded buffer with k_scatter1_wc, whatever
		// the batch's size.  With n_pos here, a batch of a few million positions in a context sized for 2^28 had more rows than workgroups, and the
		// k-mers of the rows beyond the grid were lost without a word (tests/test_gpu_parity.py::test_small_batch_in_a_large_context).
		// Surplus workgroups leave at once.
		run_stage_b(Pt, Bt, Bt.recs1, sg, sg + 8 * nb1, 8 * nb1, 8, sg + 16 * nb1, sg + 24 * nb1 + 1, (uint64_t)8 * nb1 * c->op_cap, c->st, c->evt[b]);
	} else
	run_stage_b(Pt, Bt, Bt.recs1, Bt.start1, Bt.start1 + 1, nb1, 1, Bt.row_base, Bt.start1, n_pos, c->st, c->evt[b]);
	if (handover_end(c, Bt, b) != 0) return -1;
	if (op_run) {
		HIPCK(hipMemcpyAsync(c->h_flags[b] + 1, c->op_flags + OP_STICKY, sizeof(uint32_t), hipMemcpyDeviceToHost, c->st)); // behind k_seal: is the run poisoned up to and including this batch?
		if (c->n_opq == 4) return set_err("internal: one-pass queue overflow");
		c->opq[c->n_opq].seq = d_seq; c->opq[c->n_opq].qual = d_qual_given; c->opq[c->n_opq].n_pos = n_pos; c->opq[c->n_opq].slot = b; c->opq[c->n_opq].recv = nullptr; c->opq[c->n_opq].mg = 0; ++c->n_opq;
	}
	HIPCK(hipMemcpyAsync(c->h_snap[b], c->B.stats, sizeof(unsigned long long) * ST_N * (ST_SLOTS + 1), hipMemcpyDeviceToHost, c->st));
	c->slot_call[b] = c->call_no; c->slot_pos[b] = n_pos;
	HIPCK(hipEventRecord(c->evB[b], c->st));
	HIPCK(hipGetLastError());
	c->used[b] = 1;
	++c->n_batches;
	return finalise_previous(c, b);
}

// A batch much larger than the filter's regions can take at full speed (list_cap k-mers with clear bits per region) is cut into
// sub-batches here.  Any byte that is not ACGTacgt is a cut point -- no k-mer spans it (count.c:83,88) -- and batch boundaries never
// change results, so this is invisible except in speed (every region of an oversized batch takes the exact but slow HBM path).
// Into a COLD filter every k-mer of a batch brings clear bits, and a region's load spreads far wider than sqrt(mean): there the cut sits
// at 0.85 of the list capacity and applies from the limit itself on (a batch just above the limit went whole into the empty filter and put
// every region on the slow path: c3 at 4 M reads per batch 0.66 s instead of 0.45 s).  Into a warm filter only new k-mers take list entries:
// callers split from 7/6 of the limit on -- just above it a few hundred slow regions cost less than a second pass over the bitmap.
static uint64_t split_limit(const bfcg_ctx_t *c);
static void split_rule(const bfcg_ctx_t *c, uint64_t *from, uint64_t *target_max)
{
	const uint64_t lim = split_limit(c);
	if (c->cold) { const char *e = getenv("BFCG_COLD_FRAC"); *target_max = e ? (uint64_t)((double)lim * atof(e)) : lim - lim / 10; *from = *target_max; }
	else { // warm: only the k-mers that still find clear bits take list entries -- about (0.79 - seen per position) of the positions, as the last batch
	       // showed; aim at 60 % of the list capacity, at most 3x the cold limit
		double unseen = 0.79 - c->seen_per_pos;
		if (unseen < 0.2) unseen = 0.2;
		double w = getenv("BFCG_NO_WARM_BATCHES") ? 1.0 : 0.6 / 0.95 / unseen;
		if (w < 1.0) w = 1.0;
		*target_max = (uint64_t)((double)lim * w); *from = *target_max + *target_max / 6;
	}
}
static uint64_t split_limit(const bfcg_ctx_t *c)
{
	const uint64_t nfine = ((uint64_t)1 << c->P.F) >> c->log2n;
	// positions; ~0.8 k-mers per position.  In the first batch of an empty filter every k-mer has clear bits, and a region's load varies by far
	// more than sqrt(mean) -- its few dozen distinct genome k-mers come at coverage/4 copies each -- so the cut sits below list_cap on average:
	// c2 at 1 048 576 reads per batch ran at 12.2 instead of 25 G k-mers/s with the cut at 1.15 (5 % of the regions on the slow path).
	// Callers split from 7/6 of this on: just above the limit a few hundred slow regions cost less (5 %) than a second pass over the filter (12 %).
	// (into a filter that is still filling up k_bloom3's COLD mode holds a longer list)
	const uint32_t cap = c->P.b3 && c->cold && c->list_cap_c ? c->list_cap_c : c->P.list_cap;
	// Filter mode through k_bloom3fm: a batch costs a sweep over BOTH filters (c5: 2 x 2 x 16 GiB each way -- 17 ms at the rate the kernel draws, as
	// much as the rest of its work on 8 M reads), while a region whose list overflows only takes its records in two rounds instead of one: the limit
	// sits where a region's k-mers (0.66 per position at k = 51) about fill the list, not a third below it
	if (c->P.b3fm) return (uint64_t)((double)nfine * (double)cap * 0.95 * 1.6);
	return (uint64_t)((double)nfine * (double)cap * 0.95);
}
static inline int is_acgt(uint8_t ch) { ch &= 0xDF; return ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T'; }
// last cut point in (lo, hi]: index just behind a non-ACGT byte, searched backwards from hi over at most 1 MiB; 0 = none
static uint64_t find_cut(bfcg_ctx_t *c, const uint8_t *h_seq, const uint8_t *d_seq, uint64_t lo, uint64_t hi)
{
	const uint64_t win = hi - lo < (1u << 20) ? hi - lo : (1u << 20);
	const uint8_t *p = h_seq ? h_seq + hi - win : 0;
	uint8_t *tmp = 0;
	if (!h_seq) {
		tmp = (uint8_t *)malloc(win);
		if (hipMemcpy(tmp, d_seq + hi - win, win, hipMemcpyDeviceToHost) != hipSuccess) { free(tmp); return 0; }
		p = tmp;
	}
	uint64_t cut = 0;
	for (uint64_t i = win; i > 0; --i) if (!is_acgt(p[i - 1])) { cut = hi - win + i; break; }
	free(tmp);
	return cut > lo ? cut : 0;
}

// positions of a batch this context's regions take at full speed (with several ranks: of the global batch, divided by the ranks)
extern "C" uint64_t bfcg_batch_limit(bfcg_ctx_t *c) { return split_limit(c); }
// A rank of a multi-GPU run: how many times the cold limit its regions take at full speed right now.  Into a filter that is still filling up
// every received k-mer takes an entry of its region's LDS list (1.0); later only the unseen ones do -- the share the last stage B showed
// (of RECORDS here: bfcg_mg_process_ev counts what it received) -- and the aim is 60 % of the list capacity, at most 3 x the cold limit.
extern "C" double bfcg_mg_warm_factor(bfcg_ctx_t *c)
{
	if (c->cold || getenv("BFCG_NO_WARM_BATCHES")) return 1.0;
	double unseen = 1.0 - c->seen_per_pos;
	if (unseen < 0.2) unseen = 0.2;
	const double w = 0.6 / unseen;
	return w < 1.0 ? 1.0 : w;
}

extern "C" int bfcg_count_batch_dev(bfcg_ctx_t *c, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t n_pos)
{
	if (c->n_ranks > 1) return set_err("this context is one of %d ranks: use bfcg_mg_scatter / bfcg_mg_process", c->n_ranks);
	if (n_pos == 0) return 0;
	if (n_pos > c->prm.max_batch_pos) return set_err("batch of %llu positions exceeds max_batch_pos=%llu", (unsigned long long)n_pos, (unsigned long long)c->prm.max_batch_pos);
	HIPCK(hipSetDevice(c->prm.device));
	uint64_t from, lim;
	split_rule(c, &from, &lim);
	struct depth_guard { bfcg_ctx_t *c; depth_guard(bfcg_ctx_t *c_) : c(c_) { if (c->call_depth++ == 0) ++c->call_no; } ~depth_guard() { --c->call_depth; } } guard(c);
	if (!c->B.seen_out && n_pos > from) { // oversized for this filter: equal sub-batches of at most `lim` positions
		const uint64_t target = n_pos / ((n_pos + lim - 1) / lim) + 1;
		uint64_t o = 0;
		while (n_pos - o > target + target / 8) {
			const uint64_t cut = find_cut(c, 0, d_seq, o, o + target);
			if (cut == 0) break; // a megabase without a cut point: take the rest as it is
			if (bfcg_count_batch_dev(c, d_seq + o, d_qual ? d_qual + o : 0, cut - o) != 0) return -1;
			o = cut;
		}
		if (o) return o < n_pos ? bfcg_count_batch_dev(c, d_seq + o, d_qual ? d_qual + o : 0, n_pos - o) : 0;
	}
	int rc = enqueue_batch(c, d_seq, d_qual, n_pos, 0, 0);
	if (rc == 0 && (c->B.seen_out || getenv("BFCG_SYNC_BATCHES"))) rc = drain(c); // debug aids want one batch at a time
	return rc;
}

extern "C" int bfcg_count_batch_host(bfcg_ctx_t *c, const uint8_t *h_seq, const uint8_t *h_qual, uint64_t n_pos)
{
	if (c->n_ranks > 1) return set_err("this context is one of %d ranks: use bfcg_mg_scatter / bfcg_mg_process", c->n_ranks);
	if (n_pos == 0) return 0;
	if (n_pos > c->prm.max_batch_pos) return set_err("batch of %llu positions exceeds max_batch_pos=%llu", (unsigned long long)n_pos, (unsigned long long)c->prm.max_batch_pos);
	HIPCK(hipSetDevice(c->prm.device));
	uint64_t from, lim;
	split_rule(c, &from, &lim);
	struct depth_guard { bfcg_ctx_t *c; depth_guard(bfcg_ctx_t *c_) : c(c_) { if (c->call_depth++ == 0) ++c->call_no; } ~depth_guard() { --c->call_depth; } } guard(c);
	if (!c->B.seen_out && n_pos > from) { // oversized for this filter: equal sub-batches of at most `lim` positions (see bfcg_count_batch_dev)
		const uint64_t target = n_pos / ((n_pos + lim - 1) / lim) + 1;
		uint64_t o = 0;
		while (n_pos - o > target + target / 8) {
			const uint64_t cut = find_cut(c, h_seq, 0, o, o + target);
			if (cut == 0) break;
			if (bfcg_count_batch_host(c, h_seq + o, h_qual ? h_qual + o : 0, cut - o) != 0) return -1;
			o = cut;
		}
		if (o) return o < n_pos ? bfcg_count_batch_host(c, h_seq + o, h_qual ? h_qual + o : 0, n_pos - o) : 0;
	}
	const int b = c->cur;
	HIPCK(hipMemcpyAsync(c->d_seq2[b], h_seq, n_pos, hipMemcpyHostToDevice, c->stA)); // ordered behind stage A of two batches ago (same stream)
	if (h_qual) HIPCK(hipMemcpyAsync(c->d_qual2[b], h_qual, n_pos, hipMemcpyHostToDevice, c->stA));
	HIPCK(hipEventRecord(c->evCopy, c->stA));
	int rc = enqueue_batch(c, c->d_seq2[b], h_qual ? c->d_qual2[b] : NULL, n_pos, 1, 0);
	HIPCK(hipEventSynchronize(c->evCopy)); // the caller may reuse its host buffers now
	if (rc == 0 && (c->B.seen_out || getenv("BFCG_SYNC_BATCHES"))) rc = drain(c);
	return rc;
}


// ---- batches that arrive as bit planes (include/bfc_gpu.h: bfcg_count_batch_planes) --------------------------------------------------
// 4 bits per position cross PCIe instead of 16; on the device the planes are expanded into the byte streams the stage-A kernels read (one
// streaming pass: 0.5 B read + 2 B written per position, ~0.4 ms per 10^9 positions at the copy rate) with bytes that reproduce exactly what
// count.c:72-89 would see: 'A' 'C' 'G' 'T' by code, '\n' where the position is not a base, and for the quality 0x7f (a signed char of 127:
// 127 - 33 >= q for every q <= 94) or 0x80 (-128: never >= q for q > -161).  For thresholds outside (-161, 94] no real byte can (q > 94) or
// every byte does (q <= -161) pass, so the plane is constant and the two bytes give that constant too.
__global__ __launch_bounds__(256) void k_expand_planes(const uint32_t *__restrict__ pl, uint64_t pw, uint32_t bit_off, uint64_t n_pos, int has_qual,
                                                        uint8_t *__restrict__ seq, uint8_t *__restrict__ qual)
{
	const uint64_t i0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
	if (i0 >= n_pos) return;
	const uint64_t s = i0 + bit_off, k = s >> 5;
	const uint32_t sh = (uint32_t)s & 31u;
	auto take = [&](int p) -> uint32_t { // 16 bits of plane p from bit s on
		const uint32_t *w = pl + (uint64_t)p * pw + k;
		const unsigned long long v = (unsigned long long)w[0] | ((unsigned long long)w[1] << 32);
		return (uint32_t)(v >> sh) & 0xffffu;
	};
	const uint32_t m0 = take(0), m1 = take(1), mn = take(2), mq = has_qual ? take(3) : 0u;
	auto spread = [](uint32_t nib) -> uint32_t { return ((nib & 0xFu) * 0x00204081u) & 0x01010101u; }; // bit j of the nibble -> bit 0 of byte j
	uint32_t sw[4], qw[4];
#pragma unroll
	for (int d = 0; d < 4; ++d) {
		const uint32_t c0 = spread(m0 >> (4 * d)), c1 = spread(m1 >> (4 * d)), nn = spread(mn >> (4 * d)) * 0xFFu, hq = spread(mq >> (4 * d));
		const uint32_t base = 0x41414141u + c0 * 2u + c1 * 6u + (c0 & c1) * 0x0bu; // A 0x41, C 0x43, G 0x47, T 0x54 (no carries between bytes)
		sw[d] = (base & ~nn) | (0x0a0a0a0au & nn);
		qw[d] = 0x80808080u - hq;
	}
	if (i0 + 16 <= n_pos) {
		*reinterpret_cast<uint4 *>(seq + i0) = make_uint4(sw[0], sw[1], sw[2], sw[3]);
		if (has_qual) *reinterpret_cast<uint4 *>(qual + i0) = make_uint4(qw[0], qw[1], qw[2], qw[3]);
	} else {
		for (uint64_t i = i0; i < n_pos; ++i) {
			const int j = (int)(i - i0);
			seq[i] = (uint8_t)(sw[j >> 2] >> (8 * (j & 3)));
			if (has_qual) qual[i] = (uint8_t)(qw[j >> 2] >> (8 * (j & 3)));
		}
	}
}

// last cut point in (lo, hi] of a plane set: the position just behind a set bit of the not-ACGT plane, searched backwards over at most 2^20 positions; 0 = none
static uint64_t find_cut_planes(const uint32_t *np, uint64_t lo, uint64_t hi)
{
	const uint64_t stop = hi - lo < (1u << 20) ? lo : hi - (1u << 20);
	for (uint64_t i = hi; i > stop; --i) {
		const uint64_t p = i - 1;
		if ((p & 31) == 31 && p >= stop + 32 && np[p >> 5] == 0) { i -= 31; continue; } // (a word without a separator)
		if ((np[p >> 5] >> (p & 31)) & 1u) return p + 1 > lo ? p + 1 : 0;
	}
	return 0;
}

extern "C" int bfcg_count_batch_planes(bfcg_ctx_t *c, const uint32_t *h_planes, uint64_t plane_words, uint64_t first_pos, uint64_t n_pos, int has_qual)
{
	if (c->n_ranks > 1) return set_err("this context is one of %d ranks: use bfcg_mg_scatter / bfcg_mg_process", c->n_ranks);
	if (n_pos == 0) return 0;
	if (n_pos > c->prm.max_batch_pos) return set_err("batch of %llu positions exceeds max_batch_pos=%llu", (unsigned long long)n_pos, (unsigned long long)c->prm.max_batch_pos);
	if ((first_pos + n_pos + 31) / 32 > plane_words) return set_err("positions [%llu, %llu) lie outside planes of %llu words", (unsigned long long)first_pos, (unsigned long long)(first_pos + n_pos), (unsigned long long)plane_words);
	HIPCK(hipSetDevice(c->prm.device));
	if (!c->d_planes[0]) { // (first batch of this kind)
		c->plane_cap = c->prm.max_batch_pos / 32 + 4;
		for (int b = 0; b < 2; ++b) HIPCK(hipMalloc(&c->d_planes[b], c->plane_cap * 4 * sizeof(uint32_t)));
	}
	uint64_t from, lim;
	split_rule(c, &from, &lim);
	struct depth_guard { bfcg_ctx_t *c; depth_guard(bfcg_ctx_t *c_) : c(c_) { if (c->call_depth++ == 0) ++c->call_no; } ~depth_guard() { --c->call_depth; } } guard(c);
	if (!c->B.seen_out && n_pos > from) { // oversized for this filter: equal sub-batches of at most `lim` positions (see bfcg_count_batch_dev)
		const uint64_t target = n_pos / ((n_pos + lim - 1) / lim) + 1;
		uint64_t o = 0;
		while (n_pos - o > target + target / 8) {
			const uint64_t cut = find_cut_planes(h_planes + 2 * plane_words, first_pos + o, first_pos + o + target);
			if (cut == 0) break;
			if (bfcg_count_batch_planes(c, h_planes, plane_words, first_pos + o, cut - (first_pos + o), has_qual) != 0) return -1;
			o = cut - first_pos;
		}
		if (o) return o < n_pos ? bfcg_count_batch_planes(c, h_planes, plane_words, first_pos + o, n_pos - o, has_qual) : 0;
	}
	const int b = c->cur;
	const uint64_t w0 = first_pos >> 5, nw = ((first_pos + n_pos + 31) >> 5) - w0 + 1; // (+ the word the expansion reads ahead: a spare one at the planes' end)
	const uint64_t nw_c = w0 + nw <= plane_words ? nw : plane_words - w0;
	for (int p = 0; p < (has_qual ? 4 : 3); ++p) // ordered behind stage A of two batches ago (same stream), as the byte streams of bfcg_count_batch_host are
		HIPCK(hipMemcpyAsync(c->d_planes[b] + (uint64_t)p * c->plane_cap, h_planes + (uint64_t)p * plane_words + w0, nw_c * sizeof(uint32_t), hipMemcpyHostToDevice, c->stA));
	hipLaunchKernelGGL(k_expand_planes, dim3((unsigned)((n_pos + 4095) / 4096)), dim3(256), 0, c->stA, c->d_planes[b], c->plane_cap, (uint32_t)(first_pos & 31), n_pos, has_qual,
	                   c->d_seq2[b], c->d_qual2[b]);
	HIPCK(hipEventRecord(c->evCopy, c->stA)); // (behind the expansion: where the batches' kernels run on the other stream they wait for this)
	int rc = enqueue_batch(c, c->d_seq2[b], has_qual ? c->d_qual2[b] : NULL, n_pos, 1, 0);
	HIPCK(hipEventSynchronize(c->evCopy)); // the caller may reuse its planes now
	if (rc == 0 && (c->B.seen_out || getenv("BFCG_SYNC_BATCHES"))) rc = drain(c);
	return rc;
}

extern "C" void *bfcg_dev_alloc(bfcg_ctx_t *c, uint64_t bytes)
{
	void *p = 0;
	if (hipSetDevice(c->prm.device) != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) { set_err("hipMalloc(%llu) failed", (unsigned long long)bytes); return NULL; }
	return p;
}
extern "C" void bfcg_dev_free(bfcg_ctx_t *c, void *p) { (void)c; (void)hipFree(p); }
extern "C" int bfcg_h2d(bfcg_ctx_t *c, void *dst, const void *src, uint64_t bytes)
{ HIPCK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stC)); HIPCK(hipStreamSynchronize(c->stC)); return 0; }
extern "C" int bfcg_d2h(bfcg_ctx_t *c, void *dst, const void *src, uint64_t bytes)
{ HIPCK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stC)); HIPCK(hipStreamSynchronize(c->stC)); return 0; }

// Progress WITHOUT draining the pipeline: calls = bfcg_count_batch_* calls made so far (numbered from 1 after a reset), final = the last
// call all of whose batches are known to be complete, keys_of[final - i], i < n: distinct keys after call final - i (exact: stage B alone
// creates keys, and its counters are copied out right behind it).  bfc_count prints the reference's progress lines from this.
extern "C" int bfcg_progress(bfcg_ctx_t *c, uint64_t *calls, uint64_t *final, uint64_t *keys_of, int n)
{
	// a call is complete when a batch of a LATER call has been finalised, or when nothing is in flight
	const uint64_t fin = c->pend ? (c->final_call ? c->final_call - 1 : 0) : c->call_no;
	if (calls) *calls = c->call_no;
	if (final) *final = fin;
	for (int i = 0; i < n && (uint64_t)i < fin && i < 63; ++i) keys_of[i] = c->call_keys[(fin - i) & 63];
	return 0;
}

extern "C" int bfcg_stats(bfcg_ctx_t *c, uint64_t out[BFCG_ST_N])
{
	if (drain(c) != 0) return -1;
	if (fetch_stats(c) != 0) return -1;
	for (int i = 0; i < BFCG_ST_N; ++i) out[i] = c->h_stats[i];
	out[BFCG_ST_TAB_CSHIFT] = (uint64_t)c->P.tab_cshift;
	out[BFCG_ST_BATCHES] = c->n_batches;
	return 0;
}

// cumulative stage times over all batches finalised since the last call with reset != 0 (drains the pipeline first)
extern "C" int bfcg_stage_ms(bfcg_ctx_t *c, double out[6], uint64_t *n_batches, int reset)
{
	if (drain(c) != 0) return -1;
	for (int i = 0; i < 6; ++i) out[i] = c->sum_ms[i];
	if (n_batches) *n_batches = c->n_timed;
	if (reset) { for (int i = 0; i < 6; ++i) c->sum_ms[i] = 0; c->n_timed = 0; }
	return 0;
}

extern "C" uint64_t bfcg_stream_batches(bfcg_ctx_t *c) { return c->n_stream_batches; }
extern "C" int bfcg_last_batch_ms(bfcg_ctx_t *c, float out[6]) { for (int i = 0; i < 6; ++i) out[i] = c->last_ms[i]; return 0; }

// Large device -> pageable host copies (a 16 GiB filter, a 64 GiB table).  hipMemcpy into pageable memory stages through one
// pinned buffer and one host thread, whose memcpy into never-touched pages (page faults) sets the pace at about a third of the PCIe
// rate.  Here T threads each own every T-th 32 MiB chunk: D2H into their own pinned buffer on their own stream, then memcpy to the
// destination -- the copies of some threads run under the page faults of the others.  The device must be idle on `src`.
static int d2h_parallel(int device, void *dst, const void *src, uint64_t bytes)
{
	const uint64_t CH = 32ull << 20;
	const char *env = getenv("BFC_GPU_D2H_THREADS");
	int T = env ? atoi(env) : 8;
	if (T > 32) T = 32;
	if (T < 2 || bytes < (2ull << 30)) { // below 2 GiB the threads' set-up (pinned buffers, streams: 0.1 s) costs more than it saves
		HIPCK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
		return 0;
	}
	if (!getenv("BFC_GPU_NO_THP")) { // fresh destination pages: ask for 2 MiB ones (512x fewer page faults); advisory, ignored where unsupported
		const uintptr_t a = ((uintptr_t)dst + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1), e = ((uintptr_t)dst + bytes) & ~(uintptr_t)((2u << 20) - 1);
		if (e > a) (void)madvise((void *)a, e - a, MADV_HUGEPAGE);
	}
	std::vector<hipError_t> err((size_t)T, hipSuccess);
	std::vector<std::thread> th;
	for (int t = 0; t < T; ++t)
		th.emplace_back([&, t]() {
			void *stage = 0; hipStream_t st = 0; hipError_t e;
			if ((e = hipSetDevice(device)) != hipSuccess || (e = hipHostMalloc(&stage, CH, hipHostMallocDefault)) != hipSuccess) { err[t] = e; return; }
			if ((e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) != hipSuccess) { err[t] = e; (void)hipHostFree(stage); return; }
			for (uint64_t o = (uint64_t)t * CH; o < bytes; o += (uint64_t)T * CH) {
				const uint64_t n = bytes - o < CH ? bytes - o : CH;
				if ((e = hipMemcpyAsync(stage, (const char *)src + o, n, hipMemcpyDeviceToHost, st)) != hipSuccess || (e = hipStreamSynchronize(st)) != hipSuccess) { err[t] = e; break; }
				memcpy((char *)dst + o, stage, n);
			}
			(void)hipStreamDestroy(st); (void)hipHostFree(stage);
		});
	for (auto &x : th) x.join();
	for (int t = 0; t < T; ++t) if (err[t] != hipSuccess) return set_err("device-to-host copy failed: %s", hipGetErrorString(err[t]));
	return 0;
}

extern "C" int bfcg_bloom_to_host(bfcg_ctx_t *c, int which, uint8_t *dst)
{
	unsigned long long *src = which ? c->B.bloom_hi : c->B.bloom;
	if (!src) return set_err("bloom filter %d does not exist in this mode", which);
	if (drain(c) != 0) return -1;
	HIPCK(hipStreamSynchronize(c->st));
	return d2h_parallel(c->prm.device, dst, src, c->bloom_bytes);
}

extern "C" bfc_bf_t *bfcg_export_bloom(bfcg_ctx_t *c, int which)
{
	if (c->n_ranks > 1) { set_err("this context owns 1/%d of the filter: bfcg_bloom_to_host copies the slice, a whole bfc_bf_t cannot come from one rank", c->n_ranks); return NULL; }
	bfc_bf_t *b = bfc_bf_alloc_raw(c->P.bf_shift, c->P.n_hashes); // every byte is overwritten below: no 2^(b-3)-byte memset on the host
	if (!b) { set_err("host allocation of the bloom filter failed"); return NULL; }
	if (bfcg_bloom_to_host(c, which, b->b) != 0) { bfc_bf_destroy(b); return NULL; }
	return b;
}

// ---- filters that stay in HBM behind their host object.  `bfc -1` counts into bf_high and then queries it for every k-mer again
// (correct.c:556): bfc_count hands the host copy the reference's API promises (bfc_bf_t.b is public) AND leaves a device copy here,
// which bfcg_trim_create adopts instead of uploading 2^(b-3) bytes again.  The copy is dropped when the host object is destroyed or
// written to through this library (bfc_bf_destroy / bfc_bf_insert call bfcg_resident_drop).
struct resident_t { const void *bf; void *dev; int device, n_shift; };
static resident_t g_res[16];
static std::atomic<int> g_res_n{0};
static std::mutex g_res_mu;

extern "C" void bfcg_resident_drop(const void *bf)
{
	if (g_res_n.load(std::memory_order_relaxed) == 0) return;
	for (;;) { // a filter counted on several GPUs has a copy on each of them
		void *dev = 0; int device = 0;
		{
			std::lock_guard<std::mutex> lk(g_res_mu);
			for (int i = 0; i < 16; ++i) if (g_res[i].dev && g_res[i].bf == bf) { dev = g_res[i].dev; device = g_res[i].device; g_res[i].dev = 0; g_res_n.fetch_sub(1); break; }
		}
		if (!dev) return;
		int cur = 0; (void)hipGetDevice(&cur); (void)hipSetDevice(device); (void)hipFree(dev); (void)hipSetDevice(cur);
	}
}
// a full copy of host filter `bf` that sits at `dev` on `device` (bfcg_mg.hip: gathered from the ranks' slices); 0, or -1 if the registry is full
extern "C" int bfcg_resident_register(const void *bf, void *dev, int device, int n_shift)
{
	std::lock_guard<std::mutex> lk(g_res_mu);
	for (int i = 0; i < 16; ++i) if (!g_res[i].dev) { g_res[i] = resident_t{bf, dev, device, n_shift}; g_res_n.fetch_add(1); return 0; }
	return -1;
}
// the slice of filter `which` this context owns, in place (device memory), and its size in bytes
extern "C" void *bfcg_bloom_slice(bfcg_ctx_t *c, int which, uint64_t *bytes)
{
	if (bytes) *bytes = c->bloom_bytes;
	return which ? (void *)c->B.bloom_hi : (void *)c->B.bloom;
}
static void *resident_take(const bfc_bf_t *bf, int device)
{
	if (g_res_n.load(std::memory_order_relaxed) == 0) return 0;
	std::lock_guard<std::mutex> lk(g_res_mu);
	for (int i = 0; i < 16; ++i)
		if (g_res[i].dev && g_res[i].bf == (const void *)bf && g_res[i].device == device && g_res[i].n_shift == bf->n_shift) {
			void *dev = g_res[i].dev; g_res[i].dev = 0; g_res_n.fetch_sub(1); return dev;
		}
	return 0;
}

extern "C" bfc_bf_t *bfcg_export_bloom_resident(bfcg_ctx_t *c, int which)
{
	bfc_bf_t *b = bfcg_export_bloom(c, which);
	if (!b) return NULL;
	void *dev = 0;
	if (hipSetDevice(c->prm.device) != hipSuccess || hipMalloc(&dev, c->bloom_bytes) != hipSuccess) { (void)hipGetLastError(); return b; } // no room: the host copy alone is a complete answer
	if (hipMemcpyAsync(dev, which ? c->B.bloom_hi : c->B.bloom, c->bloom_bytes, hipMemcpyDeviceToDevice, c->st) != hipSuccess ||
	    hipStreamSynchronize(c->st) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(dev); return b; }
	std::lock_guard<std::mutex> lk(g_res_mu);
	for (int i = 0; i < 16; ++i) if (!g_res[i].dev) { g_res[i] = resident_t{b, dev, c->prm.device, c->P.bf_shift}; g_res_n.fetch_add(1); return b; }
	(void)hipFree(dev); // registry full
	return b;
}

extern "C" bfc_ch_t *bfcg_export_table(bfcg_ctx_t *c)
{
	if (c->P.filter_mode) { set_err("no count table in filter mode"); return NULL; }
	const double t_e0 = dbg_now();
	if (drain(c) != 0) return NULL;
	const double t_e1 = dbg_now();
	if (c->P.seg && seg_to_legacy(c, 1) != 0) return NULL; // the host's (sub-table, key) layout is made now
	bfc_ch_t *ch = bfc_ch_alloc_raw(c->P.k, c->P.l_pre, c->P.tab_cshift);
	if (!ch) { set_err("host allocation of the count table failed"); return NULL; }
	if (hipStreamSynchronize(c->st) != hipSuccess) { set_err("the conversion of the table segments failed"); bfc_ch_destroy(ch); return NULL; }
	const double t_e2 = dbg_now();
	if (d2h_parallel(c->prm.device, bfc_ch_raw_slots(ch), c->B.table, 8ULL << (c->P.l_pre + c->P.tab_cshift)) != 0) {
		set_err("D2H copy of the count table failed"); bfc_ch_destroy(ch); return NULL;
	}
	if (getenv("BFC_GPU_TIMING")) fprintf(stderr, "[T::bfcg_export_table] batches in flight finished %.3f s, segments -> host layout (2^%d slots) %.3f s, device -> host %.3f s\n",
	                                      t_e1 - t_e0, c->P.l_pre + c->P.tab_cshift, t_e2 - t_e1, dbg_now() - t_e2);
	if (c->B.tab_first) { // order stamps travel with the table (with several ranks: into bfc_ch_union): bfc_ch_dump can then reproduce khash's layout byte for byte
		uint64_t *hf = 0, *hl = 0;
		if (bfc_ch_raw_order(ch, &hf, &hl) != 0 ||
		    d2h_parallel(c->prm.device, hf, c->B.tab_first, 8ULL << (c->P.l_pre + c->P.tab_cshift)) != 0 ||
		    hipMemcpyAsync(hl, c->B.sub_last, 8ULL << c->P.l_pre, hipMemcpyDeviceToHost, c->st) != hipSuccess ||
		    hipStreamSynchronize(c->st) != hipSuccess) { set_err("D2H copy of the order stamps failed"); bfc_ch_destroy(ch); return NULL; }
	}
	// the number of keys is the device's own statistic (exact: every test compares it with the oracle's distinct count, and at c4's full
	// size with a host recount); a recount of a 64 GiB table costs 1.4 s even on 16 threads.  BFC_GPU_RECOUNT=1 recounts and cross-checks.
	bfc_ch_raw_set_count(ch, c->h_stats[ST_KEYS]);
	if (getenv("BFC_GPU_RECOUNT")) {
		bfc_ch_raw_recount(ch);
		if (bfc_ch_count(ch) != c->h_stats[ST_KEYS]) { set_err("exported table holds %llu keys, the device counted %llu", (unsigned long long)bfc_ch_count(ch), (unsigned long long)c->h_stats[ST_KEYS]); bfc_ch_destroy(ch); return NULL; }
	}
	return ch;
}

extern "C" int bfcg_hash_positions(bfcg_ctx_t *c, const uint8_t *h_seq, const uint8_t *h_qual, uint64_t n_pos, uint64_t *out)
{
	if (n_pos > c->prm.max_batch_pos) return set_err("too many positions");
	if (drain(c) != 0) return -1;
	uint64_t *d_out = 0;
	HIPCK(hipSetDevice(c->prm.device));
	HIPCK(hipMalloc(&d_out, n_pos * 24));
	HIPCK(hipMemcpyAsync(c->d_seq, h_seq, n_pos, hipMemcpyHostToDevice, c->st));
	if (h_qual) HIPCK(hipMemcpyAsync(c->d_qual, h_qual, n_pos, hipMemcpyHostToDevice, c->st));
	run_hash_only(c->P, c->d_seq, h_qual ? c->d_qual : NULL, (int64_t)n_pos, d_out, c->st);
	HIPCK(hipGetLastError());
	HIPCK(hipMemcpyAsync(out, d_out, n_pos * 24, hipMemcpyDeviceToHost, c->st));
	HIPCK(hipStreamSynchronize(c->st));
	HIPCK(hipFree(d_out));
	return 0;
}

extern "C" int bfcg_seen_flags(bfcg_ctx_t *c, uint8_t *dst, uint64_t n_pos)
{
	if (!c->B.seen_out) return set_err("context was created without debug_seen");
	if (drain(c) != 0) return -1;
	HIPCK(hipMemcpyAsync(dst, c->B.seen_out, n_pos, hipMemcpyDeviceToHost, c->st));
	HIPCK(hipStreamSynchronize(c->st));
	return 0;
}

// pinned host memory for the ingest double buffers (bfc_count.c)
extern "C" void *bfcg_host_alloc(uint64_t bytes)
{
	void *p = 0;
	if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { set_err("hipHostMalloc(%llu) failed", (unsigned long long)bytes); return NULL; }
	return p;
}
extern "C" void bfcg_host_free(void *p) { if (p) (void)hipHostFree(p); }


// ---------------------------------------------------------------------------------------------------------------
// trim pass of `bfc -1` on the GPU (config c5): the bloom filter of k-mers seen twice is resident in HBM, every read
// of a batch gets its longest streak of bloom hits and the keep / trim decision of correct.c:557-569

struct bfcg_trim {
	KParams P;
	int device;
	hipStream_t st;
	unsigned int *bloom;
	int adopted;    // the filter was already in HBM (left there by bfc_count), not uploaded
	uint8_t *d_seq, *d_flags;
	uint64_t *d_off;
	int32_t *d_start, *d_end;
	uint64_t max_pos, max_reads;
	hipEvent_t e0, e1;
	float last_ms;
};

extern "C" bfcg_trim_t *bfcg_trim_create(int k, const bfc_bf_t *bf, int device, uint64_t max_pos, uint64_t max_reads)
{
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_err("no HIP device available: the trim pass has no CPU fallback here"); return NULL; }
	if (!bf || k < 1 || k > 63 || bf->n_shift < 9 || bf->n_shift > 37 || bf->n_hashes < 1 || bf->n_hashes > 12) { set_err("bad arguments to bfcg_trim_create"); return NULL; }
	HIPCKN(hipSetDevice(device));
	bfcg_trim_t *t = (bfcg_trim_t *)calloc(1, sizeof(bfcg_trim_t));
	memset(&t->P, 0, sizeof(t->P));
	t->P.k = k; t->P.bf_shift = bf->n_shift; t->P.n_hashes = bf->n_hashes; t->P.q = 0;
	t->device = device; t->max_pos = max_pos; t->max_reads = max_reads;
	HIPCKN(hipStreamCreate(&t->st));
	HIPCKN(hipEventCreate(&t->e0)); HIPCKN(hipEventCreate(&t->e1));
	t->bloom = (unsigned int *)resident_take(bf, device); // left in HBM by bfc_count (bfcg_export_bloom_resident)?
	t->adopted = t->bloom != 0;
	if (!t->bloom) {
		HIPCKN(hipMalloc(&t->bloom, 1ULL << (bf->n_shift - 3)));
		HIPCKN(hipMemcpy(t->bloom, bf->b, 1ULL << (bf->n_shift - 3), hipMemcpyHostToDevice));
	}
	HIPCKN(hipMalloc(&t->d_seq, max_pos)); HIPCKN(hipMalloc(&t->d_flags, max_pos));
	HIPCKN(hipMalloc(&t->d_off, (max_reads + 1) * 8));
	HIPCKN(hipMalloc(&t->d_start, max_reads * 4)); HIPCKN(hipMalloc(&t->d_end, max_reads * 4));
	return t;
}

extern "C" void bfcg_trim_destroy(bfcg_trim_t *t)
{
	if (!t) return;
	(void)hipSetDevice(t->device);
	(void)hipStreamSynchronize(t->st);
	(void)hipFree(t->bloom); (void)hipFree(t->d_seq); (void)hipFree(t->d_flags); (void)hipFree(t->d_off); (void)hipFree(t->d_start); (void)hipFree(t->d_end);
	(void)hipEventDestroy(t->e0); (void)hipEventDestroy(t->e1);
	(void)hipStreamDestroy(t->st);
	free(t);
}

// device-resident stream (d_seq may be NULL: then h_seq is copied in).  off[n_reads+1] are stream offsets: read r is
// [off[r], off[r+1]-1), byte off[r+1]-1 its separator.  start[r] = -1 if the read is dropped, else keep [start, end).
extern "C" int bfcg_trim_batch(bfcg_trim_t *t, const uint8_t *h_seq, const uint8_t *d_seq, uint64_t n_pos, const uint64_t *h_off, uint64_t n_reads,
                               float min_frac, int32_t *start, int32_t *end)
{
	if (n_pos > t->max_pos || n_reads > t->max_reads) return set_err("trim batch exceeds the capacity given to bfcg_trim_create");
	if (n_reads == 0) return 0;
	HIPCK(hipSetDevice(t->device));
	if (!d_seq) { HIPCK(hipMemcpyAsync(t->d_seq, h_seq, n_pos, hipMemcpyHostToDevice, t->st)); d_seq = t->d_seq; }
	HIPCK(hipMemcpyAsync(t->d_off, h_off, (n_reads + 1) * 8, hipMemcpyHostToDevice, t->st));
	HIPCK(hipEventRecord(t->e0, t->st));
	run_query(t->P, d_seq, (int64_t)n_pos, t->bloom, t->d_flags, t->st);
	run_streak(t->P.k, min_frac, t->d_flags, t->d_off, n_reads, t->d_start, t->d_end, t->st);
	HIPCK(hipEventRecord(t->e1, t->st));
	HIPCK(hipGetLastError());
	HIPCK(hipMemcpyAsync(start, t->d_start, n_reads * 4, hipMemcpyDeviceToHost, t->st));
	HIPCK(hipMemcpyAsync(end, t->d_end, n_reads * 4, hipMemcpyDeviceToHost, t->st));
	HIPCK(hipStreamSynchronize(t->st));
	HIPCK(hipEventElapsedTime(&t->last_ms, t->e0, t->e1));
	return 0;
}
extern "C" float bfcg_trim_last_ms(bfcg_trim_t *t) { return t->last_ms; }
extern "C" int bfcg_trim_adopted(bfcg_trim_t *t) { return t->adopted; }
extern "C" void *bfcg_trim_dev_seq(bfcg_trim_t *t) { return t->d_seq; }

// ---------------------------------------------------------------------------------------------------------------
// k-mer coverage for the corrector: bfc_ec_kcov (correct.c:96-117) for a whole batch of reads against the count
// table resident in HBM -- either uploaded from a host bfc_ch_t or borrowed from a counting context that still holds it

struct bfcg_kcov {
	KParams P;
	int device, owns_table;
	hipStream_t st;
	unsigned long long *table;
	uint8_t *d_seq, *d_flags;
	uint16_t *d_out;
	uint64_t max_pos;
	hipEvent_t e0, e1;
	float last_ms;
};

static bfcg_kcov_t *kcov_new(int k, int l_pre, int cshift, int device, uint64_t max_pos)
{
	HIPCKN(hipSetDevice(device));
	bfcg_kcov_t *t = (bfcg_kcov_t *)calloc(1, sizeof(bfcg_kcov_t));
	memset(&t->P, 0, sizeof(t->P));
	t->P.k = k; t->P.l_pre = l_pre; t->P.tab_cshift = cshift; t->P.q = 0;
	t->device = device; t->max_pos = max_pos;
	HIPCKN(hipStreamCreate(&t->st));
	HIPCKN(hipEventCreate(&t->e0)); HIPCKN(hipEventCreate(&t->e1));
	HIPCKN(hipMalloc(&t->d_seq, max_pos)); HIPCKN(hipMalloc(&t->d_flags, max_pos)); HIPCKN(hipMalloc(&t->d_out, max_pos * 2));
	return t;
}

extern "C" bfcg_kcov_t *bfcg_kcov_create(const bfc_ch_t *ch, int device, uint64_t max_pos)
{
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_err("no HIP device available: the k-mer coverage pass has no CPU fallback here"); return NULL; }
	if (!ch || max_pos == 0) { set_err("bad arguments to bfcg_kcov_create"); return NULL; }
	bfcg_kcov_t *t = kcov_new(bfc_ch_get_k(ch), bfc_ch_get_lpre(ch), bfc_ch_raw_cshift(ch), device, max_pos);
	if (!t) return NULL;
	const uint64_t bytes = 8ULL << (t->P.l_pre + t->P.tab_cshift);
	t->owns_table = 1;
	HIPCKN(hipMalloc(&t->table, bytes));
	HIPCKN(hipMemcpy(t->table, bfc_ch_raw_slots((bfc_ch_t *)ch), bytes, hipMemcpyHostToDevice));
	return t;
}

// the table stays where the count kernels built it; the context must outlive the returned object and must not count meanwhile
extern "C" bfcg_kcov_t *bfcg_kcov_attach(bfcg_ctx_t *c, uint64_t max_pos)
{
	if (!c || c->P.filter_mode || max_pos == 0) { set_err("bfcg_kcov_attach needs a table-mode context"); return NULL; }
	if (drain(c) != 0) return NULL;
	if (c->P.seg && seg_to_legacy(c) != 0) return NULL; // bfc_ch_kmer_occ probes the host's layout
	bfcg_kcov_t *t = kcov_new(c->P.k, c->P.l_pre, c->P.tab_cshift, c->prm.device, max_pos);
	if (!t) return NULL;
	t->table = c->B.table;
	return t;
}

extern "C" void bfcg_kcov_destroy(bfcg_kcov_t *t)
{
	if (!t) return;
	(void)hipSetDevice(t->device);
	(void)hipStreamSynchronize(t->st);
	if (t->owns_table) (void)hipFree(t->table);
	(void)hipFree(t->d_seq); (void)hipFree(t->d_flags); (void)hipFree(t->d_out);
	(void)hipEventDestroy(t->e0); (void)hipEventDestroy(t->e1);
	(void)hipStreamDestroy(t->st);
	free(t);
}

// stream = batch format of PART 2; out[p] (host, may be NULL) / the device buffer of bfcg_kcov_dev_out() get one packed u16 per position
extern "C" int bfcg_kcov_batch(bfcg_kcov_t *t, const uint8_t *h_seq, const uint8_t *d_seq, uint64_t n_pos, int min_occ, uint16_t *out)
{
	if (n_pos > t->max_pos) return set_err("k-mer coverage batch exceeds the capacity given at creation");
	if (n_pos == 0) return 0;
	HIPCK(hipSetDevice(t->device));
	if (!d_seq) { HIPCK(hipMemcpyAsync(t->d_seq, h_seq, n_pos, hipMemcpyHostToDevice, t->st)); d_seq = t->d_seq; }
	HIPCK(hipEventRecord(t->e0, t->st));
	run_kcov(t->P, d_seq, (int64_t)n_pos, min_occ, t->table, t->d_flags, t->d_out, t->st);
	HIPCK(hipEventRecord(t->e1, t->st));
	HIPCK(hipGetLastError());
	if (out) HIPCK(hipMemcpyAsync(out, t->d_out, n_pos * 2, hipMemcpyDeviceToHost, t->st));
	HIPCK(hipStreamSynchronize(t->st));
	HIPCK(hipEventElapsedTime(&t->last_ms, t->e0, t->e1));
	return 0;
}
extern "C" float bfcg_kcov_last_ms(bfcg_kcov_t *t) { return t->last_ms; }
extern "C" void *bfcg_kcov_dev_seq(bfcg_kcov_t *t) { return t->d_seq; }
extern "C" void *bfcg_kcov_dev_out(bfcg_kcov_t *t) { return t->d_out; }
