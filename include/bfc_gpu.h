/* bfc_gpu.h -- C ABI of libbfc_gpu.so: MI355X-native k-mer counting for BFC.
 *
 * PART 1 is the drop-in boundary: exactly the symbols the reference's count.o + bbf.o + htab.o
 * export, with the reference's argument meaning and error behaviour, so that an unmodified
 * correct.c / bfc.c link against this library instead.  Each declaration cites
 * the reference interface it replaces (paths relative to lh3/bfc r181).
 *
 * PART 2 is the device-level interface underneath (context, batches already resident in HBM,
 * export), used by bench.py, the parity tests and multi-GPU orchestration.
 *
 * Plain C, plain pointers and sizes.  All counting work runs in HIP kernels on gfx950; there is
 * no CPU fallback: without a usable GPU every entry point that has to count fails loudly
 * ("[E::bfcg] ..." on stderr, then abort() for the reference-shaped entry points that have no
 * error channel, or a negative return code for the bfcg_* ones).
 */
#ifndef BFC_GPU_H
#define BFC_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ============================================================ PART 1: reference-shaped API */

/* k-mer as four k-bit planes -- replaces kmer.h:6-8 */
typedef struct { uint64_t x[4]; } bfc_kmer_t;

/* blocked bloom filter -- replaces bbf.h:9-12.  The layout is public: correct.c:490 reads
 * bf->n_hashes directly.  `b` is host memory (2^(n_shift-3) bytes, 64-byte aligned). */
#define BFC_BLK_SHIFT 9
#define BFC_BLK_MASK  ((1 << BFC_BLK_SHIFT) - 1)
typedef struct { int n_shift, n_hashes; uint8_t *b; } bfc_bf_t;

bfc_bf_t *bfc_bf_init(int n_shift, int n_hashes);      /* bbf.h:14, bbf.c:5   NULL if n_shift outside [9,55] */
void      bfc_bf_destroy(bfc_bf_t *b);                 /* bbf.h:15, bbf.c:19 */
int       bfc_bf_insert(bfc_bf_t *b, uint64_t hash);   /* bbf.h:16, bbf.c:25  returns # of bits already set */
int       bfc_bf_get(const bfc_bf_t *b, uint64_t hash);/* bbf.h:17, bbf.c:47  thread-safe, read-only */

/* k-mer count table -- replaces htab.h:10-23 (opaque) */
#define BFC_CH_KEYBITS 50
#define BFC_CH_MAXPRE  24
struct bfc_ch_s;
typedef struct bfc_ch_s bfc_ch_t;

bfc_ch_t *bfc_ch_init(int k, int l_pre);                                               /* htab.c:19  */
void      bfc_ch_destroy(bfc_ch_t *ch);                                                /* htab.c:36  */
int       bfc_ch_insert(bfc_ch_t *ch, const uint64_t x[2], int is_high, int forced);   /* htab.c:60  0, or -1 if !forced and busy */
int       bfc_ch_get(const bfc_ch_t *ch, const uint64_t x[2]);                         /* htab.c:84  -1 | high<<8|count; thread-safe */
uint64_t  bfc_ch_count(const bfc_ch_t *ch);                                            /* htab.c:101 */
int       bfc_ch_hist(const bfc_ch_t *ch, uint64_t cnt[256], uint64_t high[64]);       /* htab.c:110 returns the mode (i>=3) or -1 */
int       bfc_ch_dump(const bfc_ch_t *ch, const char *fn);                             /* htab.c:129 0 | -1 */
bfc_ch_t *bfc_ch_restore(const char *fn);                                              /* htab.c:151 NULL on failure */
int       bfc_ch_get_k(const bfc_ch_t *ch);                                            /* htab.c:178 */
int       bfc_ch_kmer_occ(const bfc_ch_t *ch, const bfc_kmer_t *z);                    /* htab.c:94  */

/* options -- replaces bfc.h:15-33; field order and types are the ABI */
typedef struct {
	int chunk_size;
	int n_threads, no_mt_io;
	int q, k;
	int filter_mode, refine_ec, no_qual;
	float min_frac;
	int l_pre, bf_shift, n_hashes;
	int discard;
	int max_end_ext;
	int win_multi_ec;
	int min_cov;
	int w_ec, w_ec_high, w_absent, w_absent_high;
	int max_path_diff, max_heap;
} bfc_opt_t;

/* the count phase -- replaces bfc.h:39 / count.c:127.  Returns a bfc_ch_t* (table mode) or the
 * bfc_bf_t* holding k-mers seen at least twice (opt->filter_mode); the caller frees it with
 * bfc_ch_destroy / bfc_bf_destroy (bfc.c:146,149).  fn may be "-" (stdin) or gzip'd. */
void *bfc_count(const char *fn, const bfc_opt_t *opt);

/* the second phase's entry point -- bfc.h:40 / correct.c:620.  Provided ONLY for the trim pass of `bfc -1`
 * (opt->filter_mode, ptr = the bfc_bf_t* bfc_count returned): bloom queries (bbf.c:47-63), longest streak
 * (correct.c:478-497) and the keep/trim rule (correct.c:557-569) run on the GPU, output as correct.c:595-611.
 * With filter_mode off it forwards to bfc_correct_cpu(), i.e. the reference's correct.c compiled with
 * -Dbfc_correct=bfc_correct_cpu; error correction itself is not part of this library. */
void bfc_correct(const char *fn, const bfc_opt_t *opt, const void *ptr);

/* ============================================================ PART 2: device-level API */

typedef struct bfcg_ctx bfcg_ctx_t;

typedef struct {
	int k, q, bf_shift, n_hashes, l_pre, filter_mode;  /* as bfc_opt_t */
	int device;                 /* HIP device ordinal */
	uint64_t max_batch_pos;     /* capacity: positions (bases + separators) per batch, < 2^32 */
	int region_shift;           /* log2 bloom blocks per LDS region; 0 = default */
	int tab_cshift;             /* initial log2 slots per sub-table; 0 = default */
	int debug_seen;             /* allocate the per-position seen-flag buffer (tests) */
	int track_order;            /* keep per-key first-insert / per-sub-table last-call stamps: bfc_ch_dump becomes byte-identical to `bfc -t1 -d` */
	int rank, n_ranks;          /* multi-GPU, owner computes: this process is rank of n_ranks (0/0 or 0/1: single GPU) */
	int table_layout;           /* 0: region-owned table segments updated through LDS wherever the geometry allows (2k minus the bits a bloom region
	                             * implies <= 50), converted to the host's layout at export; 1: the host's (sub-table, key) layout from the start */
} bfcg_params_t;

void bfcg_params_default(bfcg_params_t *p);
/* NULL on failure (message on stderr); never falls back to the CPU */
bfcg_ctx_t *bfcg_create(const bfcg_params_t *p);
void bfcg_destroy(bfcg_ctx_t *c);
const char *bfcg_last_error(void);
/* "src:<sha256/16 of the library's sources> git:<HEAD when it was built>": ties a travelling libbfc_gpu.so to the sources it claims
 * (bfc_amd/build.py::source_hash recomputes the first part from the tree; __graft_entry__.smoke() and bench.py compare). */
const char *bfcg_build_id(void);
/* HIP devices this process sees (0 if none, or if the runtime cannot start): callers that spread a run over several GPUs (bfc_count with
 * BFC_GPU_DEVICES, bench.py --gpus N) check their device list against it and fail loudly rather than run on fewer */
int bfcg_device_count(void);
/* clear both bloom filters and the table */
int bfcg_reset(bfcg_ctx_t *c);

/* A batch is a byte stream: reads concatenated, each followed by ONE separator byte (any byte
 * that is not ACGTacgt, e.g. '\n'); `qual` is the aligned Phred+33 stream or NULL (FASTA:
 * every base counts as high quality, count.c:85).  Batches must be submitted in file order.
 * _dev: the streams are already resident in HBM.  _host: pageable/pinned host memory, copied
 * asynchronously.  Both return 0 or a negative error. */
int bfcg_count_batch_dev(bfcg_ctx_t *c, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t n_pos);
int bfcg_count_batch_host(bfcg_ctx_t *c, const uint8_t *h_seq, const uint8_t *h_qual, uint64_t n_pos);
/* The same batch as FOUR BIT PLANES (bit i of a plane = stream position i, 32 positions per word): all that count.c:72-89 reads of a position
 * is its base code (seq_nt6_table minus one: A C G T = 0 1 2 3, bseq.c:9-26, count.c:82), whether it is a base at all (count.c:83,88) and
 * whether `qual - 33 >= q` (count.c:85, a signed char) -- 4 bits instead of the 16 that cross PCIe with bfcg_count_batch_host, whose rate is
 * the link's (17 G k-mers/s at 45 GB/s).  Plane p starts at planes + p * plane_words: [0] low code bit, [1] high code bit, [2] not ACGTacgt
 * (separators included; positions beyond the batch's end in the last word too), [3] quality - 33 >= q.
 *   bfcg_plane_words(n)     words per plane for n positions
 *   bfcg_pack_planes        positions [lo, hi) of a byte-stream batch into its planes; lo a multiple of 32 (threads pack disjoint word ranges),
 *                           hi a multiple of 32 or the batch's end n_pos; qual NULL: no quality plane is written
 *   bfcg_count_batch_planes counts positions [first_pos, first_pos + n_pos) of the plane set; has_qual = 0: the records carry no qualities
 *                           (every base is high quality, count.c:85) and plane [3] is not read.  Same results as the byte streams, bit for bit. */
uint64_t bfcg_plane_words(uint64_t n_pos);
void bfcg_pack_planes(const uint8_t *seq, const uint8_t *qual, uint64_t lo, uint64_t hi, uint64_t n_pos, int q, uint32_t *planes, uint64_t plane_words);
int bfcg_count_batch_planes(bfcg_ctx_t *c, const uint32_t *h_planes, uint64_t plane_words, uint64_t first_pos, uint64_t n_pos, int has_qual);
int bfcg_sync(bfcg_ctx_t *c);

/* Multi-GPU (one process per GPU; SURVEY 8e partitioning B, "owner computes"): rank r owns 1/n_ranks of the bloom
 * regions and every k-mer that falls into them.  Per global batch every rank calls bfcg_mg_scatter on ITS contiguous
 * share of the batch (rank-major file order): records are written to d_send grouped by level-1 bucket and
 * counts[2^F1] (host) receives the bucket sizes; bucket b belongs to rank b / nb_loc.  The caller exchanges records
 * (all-to-all over RCCL) so that d_recv holds, source-major, the records of the owned buckets, and calls
 * bfcg_mg_process with seg_cnt[source][owned bucket].  info: {2^F1, owned buckets nb_loc, bytes per record, n_ranks}. */
int bfcg_mg_info(bfcg_ctx_t *c, int out[4]);
/* Positions per batch the filter's regions take at full speed (a region holds about list-capacity k-mers with clear bits in LDS; beyond
 * that it takes an exact but far slower path).  bfcg_count_batch_* cut larger batches themselves.  With several ranks the GLOBAL batch
 * (all ranks' shares together) should stay below n_ranks times this: size the filter, or the shares, accordingly -- bench.py keeps
 * 2^33 bits of filter per rank. */
uint64_t bfcg_batch_limit(bfcg_ctx_t *c);
int bfcg_mg_scatter(bfcg_ctx_t *c, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t n_pos, void *d_send, uint32_t *counts);
int bfcg_mg_process(bfcg_ctx_t *c, const void *d_recv, const uint32_t *seg_cnt);
/* stage B may partition what it received in one pass (region slabs; an overflow is found one call later and that stage B replayed from d_recv)
 * only if the caller keeps every d_recv unchanged until the call after the next has returned -- bfcg_group_* alternates its receive buffers and
 * turns this on; callers of bfcg_mg_process that reuse one buffer leave it off (the default) */
void bfcg_mg_allow_onepass(bfcg_ctx_t *c, int on);

enum { BFCG_ST_KMERS = 0, BFCG_ST_HIGH, BFCG_ST_SEEN, BFCG_ST_KEYS, BFCG_ST_TAB_OVF, BFCG_ST_ERR_POOL, BFCG_ST_SLOW_BUCKETS, BFCG_ST_CROWDED,
       BFCG_ST_TAB_CSHIFT = 8, BFCG_ST_BATCHES, BFCG_ST_N = 16 };

/* Multi-GPU in C.  A GROUP drives the local ranks of an owner-computes run of n_ranks GPUs from inside the library -- stage A, the
 * exchange of the k-mer records (grouped ncclSend / ncclRecv over RCCL, or direct peer copies between the devices of one process) and
 * stage B, one host thread per local rank, no Python and no torch in the data path.  Either every rank is local (one process, n_ranks
 * devices: what bfc_count does with BFC_GPU_DEVICES=0,1,...; a device may be named several times to emulate ranks on one GPU), or
 * exactly one is (one process per GPU, as torch.distributed.run launches bench.py): then `uid` is the run's RCCL unique id, made by
 * bfcg_group_unique_id on one process and handed to the others out of band.  `prm` as for bfcg_create (device / rank / n_ranks are set
 * per rank; max_batch_pos = positions of ONE RANK's share of a global batch).  transport: 0 auto, 1 RCCL, 2 peer copies.
 * A global batch is the ranks' shares in rank order (rank-major file order): results are those of `bfc -t1` on that order.
 * Stage A of a rank is ONE pass since round 4 (K1 once): its records land in 8 slabs per level-1 bucket, a destination's buckets are one
 * contiguous range of slabs and travel as they are, the rank's own share is written straight into its receive buffer; input that overflows
 * a slab (few, often repeated k-mers) sends that batch and the rest of the run through the two-pass stage A with exact bucket sizes.
 * reference: count.c:106 (kt_for over reads), count.c:143 (kt_pipeline) -- the fan-out lives inside bfc_count. */
typedef struct bfcg_group bfcg_group_t;
#define BFCG_UID_BYTES 128
int bfcg_group_unique_id(uint8_t uid[BFCG_UID_BYTES]);
bfcg_group_t *bfcg_group_create(const bfcg_params_t *prm, int n_ranks, int first_rank, int n_local, const int *devices, const uint8_t *uid, int transport);
void bfcg_group_destroy(bfcg_group_t *g);
int bfcg_group_info(bfcg_group_t *g, int out[6]);     /* n_ranks, n_local, transport in use (1 RCCL, 2 peer copies, 3 peer copies with the PUSH kernel for exact sizes), bytes per record, 2^F1, first rank */
/* out[0] bytes the local ranks put on the links since creation / the last reset (blocks and messages as sent), out[1] the bytes of the live records
 * among them (what an exact exchange moves; transport 3 moves exactly these + the rows), out[2] global batches, out[3] transport.  Drains the exchange. */
int bfcg_group_exchange_bytes(bfcg_group_t *g, uint64_t out[4]);
bfcg_ctx_t *bfcg_group_ctx(bfcg_group_t *g, int i);  /* local rank i's context (statistics, exports of its slice); owned by the group */
int bfcg_group_slab_mode(bfcg_group_t *g);           /* 1: stage A runs in one pass into slabs; 0: two passes (never possible, switched off, or a slab overflowed in this run) */
uint64_t bfcg_group_lazy_batches(bfcg_group_t *g);   /* global batches since creation whose exchange and stage B were enqueued before the host saw any size (slab mode; in-process and multi-process groups alike -- the processes agree on it through the set-up's all-gather, bit 1 of its first word) */
int bfcg_group_reset(bfcg_group_t *g);
/* one global batch: local rank i contributes the stream d_seq[i] / d_qual[i] (on ITS device) of n_pos[i] positions (0 = nothing) */
int bfcg_group_count_batch_dev(bfcg_group_t *g, const uint8_t *const *d_seq, const uint8_t *const *d_qual, const uint64_t *n_pos);
/* one global batch from host memory, cut by the library into n_ranks contiguous shares at read boundaries (all ranks local) */
int bfcg_group_count_batch_host(bfcg_group_t *g, const uint8_t *h_seq, const uint8_t *h_qual, uint64_t n_pos);
/* Drains every local rank.  COLLECTIVE when the ranks live in several processes: it ends with a one-word all-gather of the ranks' status, so
 * every process of the run must call it (also one that has met an error -- it then reports -1 everywhere), or its peers block in that
 * all-gather.  The same holds for bfcg_group_count_batch_*: every process submits every global batch (an empty share is fine). */
int bfcg_group_sync(bfcg_group_t *g);
int bfcg_group_stats(bfcg_group_t *g, uint64_t out[BFCG_ST_N]);   /* sums over the local ranks */
/* progress without draining (as bfcg_progress, below): *batches = global batches submitted, *final = the last one complete on every local rank,
 * keys_of[j], j < n: distinct keys (local ranks summed) after global batch *final - j; returns the number of valid entries */
int bfcg_group_progress(bfcg_group_t *g, uint64_t *batches, uint64_t *final, uint64_t *keys_of, int n);
bfc_ch_t *bfcg_group_export_table(bfcg_group_t *g);              /* all ranks local: THE table (union of the ranks' disjoint tables) */
bfc_bf_t *bfcg_group_export_bloom(bfcg_group_t *g, int which);   /* all ranks local: THE filter (the ranks' slices in rank order) */
/* the same, and every local device keeps a full copy of the filter in HBM (all-gathered from the slices by peer copies) for the sharded
 * trim pass of `bfc -1` to adopt (bfcg_trim_create on that device), until bfc_bf_destroy */
bfc_bf_t *bfcg_group_export_bloom_resident(bfcg_group_t *g, int which);

/* device memory helpers so that callers (bench.py) can stage inputs without another runtime */
void *bfcg_dev_alloc(bfcg_ctx_t *c, uint64_t bytes);
void  bfcg_dev_free(bfcg_ctx_t *c, void *p);
int   bfcg_h2d(bfcg_ctx_t *c, void *dst, const void *src, uint64_t bytes);
int   bfcg_d2h(bfcg_ctx_t *c, void *dst, const void *src, uint64_t bytes);
void *bfcg_host_alloc(uint64_t bytes);   /* pinned host memory */
void  bfcg_host_free(void *p);

int bfcg_stats(bfcg_ctx_t *c, uint64_t out[BFCG_ST_N]);   /* drains the pipeline first */
/* progress without draining: *calls = bfcg_count_batch_* calls since the last reset, *final = the last of them whose batches are all
 * complete, keys_of[i] (i < n) = distinct keys in the table after call *final - i.  Exact, because the table's counters are copied out
 * right behind every batch's table stage (count.c:113 prints the count after every chunk; bfc_count prints it from here). */
int bfcg_progress(bfcg_ctx_t *c, uint64_t *calls, uint64_t *final, uint64_t *keys_of, int n);
/* how the count table is held right now: out[0] 1 = region-owned segments, 0 = the host's layout; out[1] log2 slots per segment;
 * out[2] log2 slots per sub-table; out[3] segment growths so far */
int bfcg_table_info(bfcg_ctx_t *c, int out[4]);
/* how the k-mers are partitioned: out[0] bit 0 = the one-pass level-1 partition (the k-mer hash is computed once per batch, no histogram pass) is in
 * use, bit 1 = level 2 runs in one pass too (a slab per bloom region, no k_hist2); out[1] batches that had to be replayed through the two-pass
 * partition so far (input with few, often repeated k-mers overflows a one-pass slab) */
int bfcg_partition_info(bfcg_ctx_t *c, uint64_t out[2]);
/* launches of k_scatter1_wc (level 1 of the one-pass partition through write-combining buffers in LDS: bfcg_scatter1wc.hip) by this process so far */
uint64_t bfcg_s1wc_launches(void);
/* batches handled without aggregation (k-mers that hardly repeat inside a batch: seen k-mers are streamed to the table kernel) */
uint64_t bfcg_stream_batches(bfcg_ctx_t *c);

/* per-stage GPU time of the last batch, HIP events on the context's stream (ms):
 * [0] hist1+scans [1] scatter1 [2] hist2+scan2+scatter2 [3] bloom regions [4] table commit [5] total */
int bfcg_last_batch_ms(bfcg_ctx_t *c, float out[6]);
/* the same, summed over every batch finalised since the last reset != 0 call; drains the pipeline first */
int bfcg_stage_ms(bfcg_ctx_t *c, double out[6], uint64_t *n_batches, int reset);

/* results */
int bfcg_bloom_to_host(bfcg_ctx_t *c, int which /*0: bf, 1: bf_high*/, uint8_t *dst);   /* 2^(bf_shift-3) / n_ranks bytes: the owned slice */
bfc_bf_t *bfcg_export_bloom(bfcg_ctx_t *c, int which);   /* host bfc_bf_t (caller: bfc_bf_destroy) */
/* the same, and a copy of the filter stays in HBM behind the returned object until bfc_bf_destroy / a host bfc_bf_insert on it:
 * bfcg_trim_create on the same device adopts that copy instead of uploading 2^(n_shift-3) bytes (`bfc -1`: count.c:153 -> correct.c:556) */
bfc_bf_t *bfcg_export_bloom_resident(bfcg_ctx_t *c, int which);
void bfcg_resident_drop(const void *bf);                   /* drop the HBM copy behind a host filter, if there is one */
bfc_ch_t *bfcg_export_table(bfcg_ctx_t *c);              /* host bfc_ch_t (caller: bfc_ch_destroy) */

/* Ingest only, no GPU: parses `fn` (FASTA/FASTQ, plain or gzip) into the batches bfc_count would submit -- kseq's grammar
 * (kseq.h:185-224) and bseq_read's batch boundary (bseq.c:52-76) -- and digests them: out[0] batches, [1] reads, [2] stream positions,
 * [3]/[4] FNV-1a of all sequence / quality streams, [5] FNV-1a of the per-batch read counts, [6] batches parsed by the multi-threaded
 * fast path (uncompressed strict 4-line FASTQ; n_threads = 0 forces the serial parser).  Used to test that both parsers agree. */
int bfc_ingest_digest(const char *fn, uint64_t chunk_size, uint64_t cap, int n_threads, uint64_t out[7]);
/* the same batches as bit planes: direct = 1 written by the FASTQ fast path straight from the mapped file, 0 packed from the byte streams (they
 * must agree word for word: tests/test_ingest.py).  out[0] batches, out[1] positions, out[2..5] FNV-1a of planes 0..3, out[6] batches packed directly */
int bfc_ingest_planes_digest(const char *fn, uint64_t chunk_size, uint64_t cap, int n_threads, int q, int direct, uint64_t out[7]);
/* gzip input (bseq.c:33-50 reads it through one gzread stream): bfc_count inflates a regular .gz file with n_threads threads -- guessed
 * block starts, 16-bit symbols with markers for the unknown 32 KiB window, pieces chained only where one started exactly where its
 * predecessor stopped, CRC-32 / ISIZE of every member checked (bfc_pgz.h); anything it cannot decode goes through gzread.  This runs that
 * inflate alone over `fn` in windows of `window` bytes with compressed chunks of `chunk` bytes: out[0] text bytes, [1] their CRC-32,
 * [2] pieces taken as guessed, [3] pieces decoded again from the chain position, [4] rounds.  0 / -1 not a mappable gzip file / -2 damaged. */
int bfc_pgz_digest(const char *fn, int n_threads, uint64_t chunk, uint64_t window, uint64_t out[5]);

/* Union of the per-GPU tables of an owner-computes run (disjoint key sets: the union is the reference's table); order stamps travel
 * along, so that bfc_ch_dump of the union is byte-identical to `bfc -t1 -d` across GPUs too.  NULL if k / l_pre differ. */
bfc_ch_t *bfc_ch_union(const bfc_ch_t *const *tabs, int n);

/* L1 form of a host table: sizes[2^l_pre]; slots (may be NULL) = per sub-table sorted */
int      bfc_ch_get_lpre(const bfc_ch_t *ch);
uint64_t bfc_ch_export_sorted(const bfc_ch_t *ch, uint32_t *sizes, uint64_t *slots);

/* Trim pass of `bfc -1` (BASELINE config c5: bloom query-only kernel): replaces, for a whole batch of reads, the per-read
 * max_streak (correct.c:478-497: one bfc_bf_get per k-mer, bbf.c:47-63) and the keep/trim rule (correct.c:557-569).
 * `bf` is the filter bfc_count returned in filter mode; it is uploaded once.  The stream is the batch format of PART 2
 * (one separator byte after each read); off[n_reads+1] are the reads' stream offsets.  start[r] = -1: read dropped;
 * otherwise keep bases [start[r], end[r]) -- exactly correct.c's memmove window. */
typedef struct bfcg_trim bfcg_trim_t;
bfcg_trim_t *bfcg_trim_create(int k, const bfc_bf_t *bf, int device, uint64_t max_pos, uint64_t max_reads);
void bfcg_trim_destroy(bfcg_trim_t *t);
int bfcg_trim_batch(bfcg_trim_t *t, const uint8_t *h_seq, const uint8_t *d_seq, uint64_t n_pos, const uint64_t *h_off, uint64_t n_reads,
                    float min_frac, int32_t *start, int32_t *end);
float bfcg_trim_last_ms(bfcg_trim_t *t);   /* GPU time of the last batch's query + streak kernels (HIP events) */
int bfcg_trim_adopted(bfcg_trim_t *t);     /* 1 if the filter was found resident in HBM (no upload) */
void *bfcg_trim_dev_seq(bfcg_trim_t *t);   /* the context's device staging buffer (max_pos bytes) */

/* k-mer coverage of the corrector: replaces, for a whole batch of reads, bfc_ec_kcov (correct.c:96-117) as
 * bfc_ec1 first calls it on the unmodified read (correct.c:403) -- one bfc_ch_kmer_occ (htab.c:94-99) per k-mer.  The table
 * is uploaded once from the host bfc_ch_t that bfc_count / bfc_ch_restore returned (bfcg_kcov_create), or borrowed in place
 * from a counting context (bfcg_kcov_attach: no export).  Output: one u16 per stream position, packed like ecbase_t's
 * bit-fields (correct.c:17): lcov (bits 0-5) | hcov (6-11) | solid_end (12) | high_end (13); 0 for separators.
 * min_occ is opt->min_cov (bfc.h:20, correct.c:403). */
typedef struct bfcg_kcov bfcg_kcov_t;
bfcg_kcov_t *bfcg_kcov_create(const bfc_ch_t *ch, int device, uint64_t max_pos);
bfcg_kcov_t *bfcg_kcov_attach(bfcg_ctx_t *ctx, uint64_t max_pos);
void bfcg_kcov_destroy(bfcg_kcov_t *t);
int bfcg_kcov_batch(bfcg_kcov_t *t, const uint8_t *h_seq, const uint8_t *d_seq, uint64_t n_pos, int min_occ, uint16_t *out);
float bfcg_kcov_last_ms(bfcg_kcov_t *t);   /* GPU time of the last batch's two kernels (HIP events) */
void *bfcg_kcov_dev_seq(bfcg_kcov_t *t);   /* device staging buffer for the stream (max_pos bytes) */
void *bfcg_kcov_dev_out(bfcg_kcov_t *t);   /* device result of the last batch (max_pos u16) */

/* unit-test hooks: K1 only.  out = 3 u64 per position: y0, y1, flags (bit0 k-mer ends here, bit1 high) */
int bfcg_hash_positions(bfcg_ctx_t *c, const uint8_t *h_seq, const uint8_t *h_qual, uint64_t n_pos, uint64_t *out);
/* per-position seen flags of the last batch (debug_seen): 0 none, 1 not seen, 2 seen */
int bfcg_seen_flags(bfcg_ctx_t *c, uint8_t *dst, uint64_t n_pos);

#ifdef __cplusplus
}
#endif
#endif
