/* bfc_count.c -- the count phase behind the reference's own entry point:
 *     void *bfc_count(const char *fn, const bfc_opt_t *opt)          (bfc.h:39, count.c:127-157)
 * Host work here is ingest only: FASTA/FASTQ (plain or gzip, "-" = stdin) is parsed into the
 * separator-delimited byte streams the kernels take, in pinned double buffers, by a reader
 * thread that runs one batch ahead of the GPU (the reference overlaps I/O and counting the same
 * way with kt_pipeline, count.c:143).  Every k-mer is hashed, filtered and counted on the GPU
 * (bfcg_count_batch_host); there is no CPU counting path.
 *
 * Record grammar follows kseq.h:185-224 as bseq_read (bseq.c:52-76) uses it: '>' or '@' header,
 * sequence lines until a line starting with '+', '>' or '@'; after '+', quality lines until at
 * least as many characters as bases; a length mismatch ends the input (kseq returns -2 and
 * bseq_read stops).  A FASTA record has no qualities: all its bases count as high quality
 * (count.c:85): a batch without any qualities is submitted without a quality stream, a mixed batch as its
 * homogeneous runs (bfc_ingest.h: kind_cut).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <pthread.h>
#include <sys/time.h>
#include <sys/resource.h>
#include <zlib.h>
#include "bfc_gpu.h"

/* globals owned by the reference's bfc.c:13-15 when this library is linked into `bfc`;
 * weak so that the library also loads on its own (ctypes) */
extern double bfc_real_time __attribute__((weak));
extern int bfc_verbose __attribute__((weak));

void *bfcg_host_alloc(uint64_t bytes);
void bfcg_host_free(void *p);

static double now_real(void) { struct timeval tp; gettimeofday(&tp, 0); return tp.tv_sec + tp.tv_usec * 1e-6; }
static double now_cpu(void)
{
	struct rusage r; getrusage(RUSAGE_SELF, &r);
	return r.ru_utime.tv_sec + r.ru_stime.tv_sec + 1e-6 * (r.ru_utime.tv_usec + r.ru_stime.tv_usec);
}

#include "bfc_ingest.h"

/* Plain (unpinned) batch buffers: 2 MiB-aligned and advised to use huge pages -- 64 parser threads touching 1.6 GB of fresh 4 KiB pages take the
 * address space's lock 400 000 times beside the context's hipMalloc calls (device buffers 0.4 -> 1.0 s when they did), and giving such pages back
 * costs as much as unpinning them.  Advisory: where transparent huge pages are off this is a plain aligned allocation. */
#include <sys/mman.h>
static void *big_alloc(uint64_t bytes)
{
	void *p = 0;
	const uint64_t rounded = (bytes + (2u << 20) - 1) & ~(uint64_t)((2u << 20) - 1);
	if (posix_memalign(&p, 2u << 20, rounded) != 0) return 0;
#ifdef MADV_HUGEPAGE
	(void)madvise(p, rounded, MADV_HUGEPAGE);
#endif
	return p;
}

/* ------------------------------------------------------------------ reader thread, one batch ahead */

typedef struct {
	ingest_t *ps;
	batch_t b[2];
	int ready[2];   /* filled and not yet consumed */
	int have[2];    /* the slot's buffers exist (round 6: the reader starts on slot 0 while slot 1 is still being pinned) */
	int done;
	pthread_mutex_t mtx; pthread_cond_t cv;
	/* one GPU: a filled batch is packed into its four bit planes (bfcg_pack_planes: 4 bits per position cross PCIe instead of 16) by the
	 * parser's own threads, and only the planes' buffers are pinned -- an eighth of the bytes the two streams take */
	uint32_t *planes[2]; uint64_t plane_words; int q, pack_threads;
	bfc_pool_t *pack_pool; /* the submitting thread packs batch t with workers of its own while the reader's threads parse batch t + 1 */
} pipe_t;

typedef struct { const batch_t *b; uint32_t *planes; uint64_t plane_words, lo, hi; int q; } pack_job_t;
static void *pack_main(void *arg)
{
	pack_job_t *j = (pack_job_t*)arg;
	bfcg_pack_planes(j->b->seq, j->b->has_qual ? j->b->qual : 0, j->lo, j->hi, j->b->n_pos, j->q, j->planes, j->plane_words);
	return 0;
}
static void pack_batch(pipe_t *pp, int i)
{
	pack_job_t job[64];
	const batch_t *b = &pp->b[i];
	int t, T = pp->pack_threads < 1 ? 1 : pp->pack_threads > 64 ? 64 : pp->pack_threads;
	uint64_t per;
	if (b->n_pos == 0) return;
	if (b->n_pos < ((uint64_t)T << 16)) T = 1;
	per = ((b->n_pos + T - 1) / T + 31) & ~(uint64_t)31; /* threads pack disjoint words */
	for (t = 0; t < T; ++t) {
		job[t].b = b; job[t].planes = pp->planes[i]; job[t].plane_words = pp->plane_words; job[t].q = pp->q;
		job[t].lo = per * t < b->n_pos ? per * t : b->n_pos; job[t].hi = per * (t + 1) < b->n_pos ? per * (t + 1) : b->n_pos;
	}
	bfc_pool_run(pp->pack_pool, pack_main, job, sizeof(pack_job_t), T);
}

static void *reader_main(void *arg)
{
	pipe_t *pp = (pipe_t*)arg;
	int i = 0;
	for (;;) {
		pthread_mutex_lock(&pp->mtx);
		while (pp->ready[i] || !pp->have[i]) pthread_cond_wait(&pp->cv, &pp->mtx);
		pthread_mutex_unlock(&pp->mtx);
		ingest_fill(pp->ps, &pp->b[i]);
		pthread_mutex_lock(&pp->mtx);
		pp->ready[i] = 1;
		pthread_cond_broadcast(&pp->cv);
		pthread_mutex_unlock(&pp->mtx);
		if (pp->b[i].last) break;
		i ^= 1;
	}
	return 0;
}

/* BFC_GPU_DEVICES="0,1,2,3" (or a count, "4" = devices 0..3): the GPUs bfc_count and the trim pass of bfc_correct spread a file over.
 * A device may be named several times (ranks emulated on one GPU).  Returns the number of devices (0: variable not set). */
int bfcg_env_devices(int *dev, int max)
{
	const char *e = getenv("BFC_GPU_DEVICES");
	int n = 0;
	if (!e || !*e) return 0;
	if (!strchr(e, ',')) { int c = atoi(e), i; if (c > max) c = max; if (c < 0) c = 0; for (i = 0; i < c; ++i) dev[i] = i; return c; } /* a count */
	while (*e && n < max) { dev[n++] = (int)strtol(e, (char**)&e, 10); while (*e == ',' || *e == ' ') ++e; }
	return n;
}

/* context creation (device buffers, zeroed filter and table: 0.1-0.3 s) runs on its own thread while the input is opened, the host buffers
 * are pinned and the first batch is parsed */
typedef struct { bfcg_params_t prm; int n_dev; int *devs; bfcg_ctx_t *ctx; bfcg_group_t *grp; char err[512]; } create_job_t;
static void *create_main(void *arg)
{
	create_job_t *j = (create_job_t*)arg;
	const char *env;
	if (j->n_dev > 1) j->grp = bfcg_group_create(&j->prm, j->n_dev, 0, j->n_dev, j->devs, 0, (env = getenv("BFC_GPU_TRANSPORT")) ? atoi(env) : 0);
	else j->ctx = bfcg_create(&j->prm);
	if (!j->grp && !j->ctx) { strncpy(j->err, bfcg_last_error(), sizeof(j->err) - 1); j->err[sizeof(j->err) - 1] = 0; } /* the message is thread-local */
	return 0;
}

/* ------------------------------------------------------------------ bfc_count */

/* count.c:110-114, for every submitted reader batch that is complete on the GPU by now (in order) */
static void print_progress(bfcg_ctx_t *ctx, bfcg_group_t *grp, const bfc_opt_t *opt, double t0, const uint64_t *pend_call, const int *pend_seqs, unsigned *lo, unsigned hi)
{
	uint64_t final = 0, keys[64];
	int n_keys = 63;
	if (grp) n_keys = bfcg_group_progress(grp, 0, &final, keys, 63); /* several GPUs: a "call" is a global batch, the keys are the ranks' sums */
	else bfcg_progress(ctx, 0, &final, keys, 63);
	while (*lo != hi && pend_call[*lo & 63] <= final && n_keys > 0) {
		const double rt = now_real() - t0, eff = 100. * now_cpu() / (rt + 1e-6);
		/* keys[a] = distinct keys after the call `a` calls before the last complete one.  A batch that has fallen out of that window (the
		 * library cut it into many calls, or nobody asked for a while) is printed with the oldest count still known -- never left waiting,
		 * which would stall every line behind it and let the 64-entry ring of pending batches wrap */
		uint64_t age = final - pend_call[*lo & 63];
		if (age >= (uint64_t)n_keys) age = (uint64_t)n_keys - 1;
		if (!opt->filter_mode)
			fprintf(stderr, "[M::%s @%.1f*%.1f%%] processed %d sequences; # distinct k-mers: %ld\n", "bfc_count_cb", rt, eff, pend_seqs[*lo & 63], (long)keys[age]);
		else
			fprintf(stderr, "[M::%s @%.1f*%.1f%%] processed %d sequences\n", "bfc_count_cb", rt, eff, pend_seqs[*lo & 63]);
		++*lo;
	}
}

void *bfc_count(const char *fn, const bfc_opt_t *opt)
{
	bfcg_params_t prm;
	bfcg_ctx_t *ctx = 0;
	bfcg_group_t *grp = 0; /* several GPUs (BFC_GPU_DEVICES): the library's rank threads do stage A, the exchange over RCCL and stage B */
	int devs[64], n_dev;
	ingest_t ps;
	pipe_t pp;
	pthread_t tid, ctid;
	create_job_t cj;
	void *ret;
	const char *env;
	double t0 = (&bfc_real_time && bfc_real_time > 0.) ? bfc_real_time : now_real();
	uint64_t cap, bases;
	int i, cur = 0, io_threads, timing, small_input = 0, pin, use_planes;
	double tt, t_wait = 0, t_submit = 0, t_pack = 0;
	uint64_t pend_call[64]; int pend_seqs[64]; unsigned n_pend_lo = 0, n_pend_hi = 0; /* reader batches submitted, their progress line not printed yet */

	bfcg_params_default(&prm);
	prm.k = opt->k; prm.q = opt->q; prm.bf_shift = opt->bf_shift; prm.n_hashes = opt->n_hashes;
	prm.l_pre = opt->l_pre; prm.filter_mode = opt->filter_mode;
	prm.device = (env = getenv("BFC_GPU_DEVICE")) ? atoi(env) : 0;
	prm.track_order = (env = getenv("BFC_GPU_EXACT_DUMP")) ? atoi(env) : 0; /* byte-identical -d dump (costs one u64 per table slot) */
	/* a batch holds one reference chunk (opt->chunk_size bases, bfc.c:20,-L) plus separators;
	 * BFC_GPU_BATCH overrides the number of positions per GPU batch */
	cap = (uint64_t)(opt->chunk_size > 0 ? opt->chunk_size : 100000000);
	/* a batch streams the whole bitmap through the CUs once, and a 16 KiB region handles ~1000 k-mers per batch at full speed:
	 * batches grow with the filter (x4 for -b35, x16 for -b37 as `-s 3g` sets it); batch boundaries never change results */
	if (opt->bf_shift > 33) cap <<= opt->bf_shift - 33;
	if ((env = getenv("BFC_GPU_BATCH")) != 0) cap = strtoull(env, 0, 10);
	if (cap > (1ULL << 32) - (1ULL << 27)) cap = (1ULL << 32) - (1ULL << 27);
	bases = cap;
	{ /* an uncompressed regular file cannot hold more positions than bytes: no 100 GB of buffers for a small input behind a big filter */
		struct stat sb;
		FILE *fp;
		if (fn && strcmp(fn, "-") && stat(fn, &sb) == 0 && S_ISREG(sb.st_mode) && (fp = fopen(fn, "rb")) != 0) {
			int c0 = fgetc(fp), c1 = fgetc(fp);
			fclose(fp);
			if (!(c0 == 0x1f && c1 == 0x8b) && (uint64_t)sb.st_size + 4096 < cap) cap = (uint64_t)sb.st_size + 4096;
			if ((uint64_t)sb.st_size < (3ULL << 30)) small_input = 1;
		}
	}                 /* the batch boundary is bseq_read's: the read that brings a batch to `bases` is its last */
	if (cap < (1u << 16)) cap = 1u << 16;
	cap += cap / 64 + (1u << 20); /* separators, and the read that crosses the boundary */
	if (cap >= (1ULL << 32)) cap = (1ULL << 32) - 1;
	prm.max_batch_pos = cap;
	timing = getenv("BFC_GPU_TIMING") != 0; /* phase times on stderr */
	tt = now_real();
	if (timing) fprintf(stderr, "[T::bfc_count] entered %.3f s after the process started\n", tt - t0);
	n_dev = bfcg_env_devices(devs, 64);
	if (n_dev > 1 && (n_dev & (n_dev - 1))) { fprintf(stderr, "[E::%s] BFC_GPU_DEVICES names %d devices: the bloom regions are dealt to a power of two of GPUs\n", __func__, n_dev); abort(); }
	if (n_dev > 1) prm.max_batch_pos = cap / (uint64_t)n_dev + cap / 64 + (1u << 16); /* every rank takes 1/n of each batch (cut at read boundaries: shares differ by a read or two) */
	else if (n_dev == 1) prm.device = devs[0];
	memset(&cj, 0, sizeof(cj));
	cj.prm = prm; cj.n_dev = n_dev; cj.devs = devs;
	pthread_create(&ctid, 0, create_main, &cj);

	/* parser threads: -t (the count itself needs no host threads), BFC_GPU_IO_THREADS overrides; 0 = serial parser only */
	io_threads = (env = getenv("BFC_GPU_IO_THREADS")) ? atoi(env) : opt->n_threads > 1 ? opt->n_threads : 0;
	if (ingest_open(&ps, fn, bases, io_threads, opt->no_mt_io ? 1 : 2) != 0) { fprintf(stderr, "[E::%s] cannot open '%s'\n", __func__, fn ? fn : "-"); abort(); }

	pin = (env = getenv("BFC_GPU_PIN")) ? atoi(env) != 0 : !small_input;
	/* one GPU: batches cross PCIe as bit planes (BFC_GPU_PLANES=0: as byte streams); several GPUs: every rank takes its share of the streams */
	use_planes = n_dev <= 1 && !((env = getenv("BFC_GPU_PLANES")) && atoi(env) == 0);
	memset(&pp, 0, sizeof(pp));
	pp.ps = &ps;
	pp.q = opt->q; pp.pack_threads = io_threads > 32 ? 32 : io_threads > 1 ? io_threads : 1; pp.plane_words = bfcg_plane_words(cap);
	if (use_planes && pp.pack_threads > 1) pp.pack_pool = bfc_pool_create(pp.pack_threads);
	pthread_mutex_init(&pp.mtx, 0); pthread_cond_init(&pp.cv, 0);
	for (i = 0; i < 2; ++i) {
		pp.b[i].cap = cap;
		/* pinning costs ~0.35 ms per MB: it pays from a few GB of input on; below, plain memory and staged copies.  With planes only THEY are
		 * copied to the device: a quarter of a stream's bytes each */
		if (use_planes) { /* the FASTQ fast path writes the planes straight from the mapped file; byte streams exist only if the serial parser is ever needed (bfc_ingest.h: batch_need_streams) */
			pp.planes[i] = (uint32_t*)(pin ? bfcg_host_alloc(pp.plane_words * 16) : malloc(pp.plane_words * 16));
			pp.b[i].planes = pp.planes[i]; pp.b[i].plane_words = pp.plane_words; pp.b[i].q = opt->q;
		} else if (pin) { pp.b[i].seq = (uint8_t*)bfcg_host_alloc(cap); pp.b[i].qual = (uint8_t*)bfcg_host_alloc(cap); }
		else { pp.b[i].seq = (uint8_t*)big_alloc(cap); pp.b[i].qual = (uint8_t*)big_alloc(cap); }
		if ((!use_planes && (!pp.b[i].seq || !pp.b[i].qual)) || (use_planes && !pp.planes[i])) { fprintf(stderr, "[E::%s] cannot pin %llu bytes of host memory\n", __func__, (unsigned long long)cap); abort(); }
		/* pinning a slot's buffers is 70 ms for c3's batches: the reader starts parsing into slot 0 while slot 1 is pinned (it asks for slot 1 a batch later) */
		pthread_mutex_lock(&pp.mtx); pp.have[i] = 1; pthread_cond_broadcast(&pp.cv); pthread_mutex_unlock(&pp.mtx);
		if (i == 0 && !opt->no_mt_io) pthread_create(&tid, 0, reader_main, &pp);
	}
	if (timing) fprintf(stderr, "[T::bfc_count] input opened (%s), pinned buffers: %.3f s\n", ps.fast.pipe ? "a pipe: reader thread into a ring, multi-threaded fast path" : ps.fast.active ? "mapped, multi-threaded fast path" : "serial parser", now_real() - tt);
	pthread_join(ctid, 0);
	ctx = cj.ctx; grp = cj.grp;
	if (!ctx && !grp) { fprintf(stderr, "[E::%s] cannot set up the GPU count path: %s\n", __func__, cj.err); abort(); }
	if (timing) fprintf(stderr, "[T::bfc_count] GPU context%s ready (buffers for %llu positions per batch): %.3f s\n", grp ? "s" : "", (unsigned long long)cap, now_real() - tt);

	for (;;) {
		batch_t *b = &pp.b[cur];
		tt = now_real();
		if (opt->no_mt_io) ingest_fill(&ps, b);
		else {
			pthread_mutex_lock(&pp.mtx);
			while (!pp.ready[cur]) pthread_cond_wait(&pp.cv, &pp.mtx);
			pthread_mutex_unlock(&pp.mtx);
		}
		t_wait += now_real() - tt; tt = now_real();
		if (use_planes && !b->packed) { pack_batch(&pp, cur); t_pack += now_real() - tt; tt = now_real(); } /* (a batch of the serial parser: byte streams) */
		fprintf(stderr, "[M::%s] read %d sequences\n", "bfc_count_cb", b->n_seqs); /* count.c:99, once per bseq_read call */
		if (b->n_seqs) {
			int rc = 0;
			if (b->has_qual && b->n_noq) { /* mixed batch: its runs of records with / without qualities, one after the other (a record without
			                                * qualities is all high quality whatever -q says, count.c:85) -- on one GPU and on several alike */
				uint64_t o = 0;
				int j, kind = (b->n_cut & 1) ? !b->last_kind : b->last_kind; /* kind of the first run: the kinds alternate at every cut */
				for (j = 0; j <= b->n_cut && rc == 0; ++j, kind = !kind) {
					const uint64_t e = j < b->n_cut ? b->kind_cut[j] : b->n_pos;
					if (e > o) rc = grp ? bfcg_group_count_batch_host(grp, b->seq + o, kind ? b->qual + o : 0, e - o)
					              : use_planes ? bfcg_count_batch_planes(ctx, pp.planes[cur], pp.plane_words, o, e - o, kind)
					                    : bfcg_count_batch_host(ctx, b->seq + o, kind ? b->qual + o : 0, e - o);
					o = e;
				}
			} else rc = grp ? bfcg_group_count_batch_host(grp, b->seq, b->has_qual ? b->qual : 0, b->n_pos)
			          : use_planes ? bfcg_count_batch_planes(ctx, pp.planes[cur], pp.plane_words, 0, b->n_pos, b->has_qual)
			                : bfcg_count_batch_host(ctx, b->seq, b->has_qual ? b->qual : 0, b->n_pos);
			if (rc != 0) { fprintf(stderr, "[E::%s] GPU counting failed: %s\n", __func__, bfcg_last_error()); abort(); }
			{ /* the line of count.c:110-114 is printed when the batch is COMPLETE on the GPU(s) -- without waiting for it here: the kernels of
			   * this batch run under the parsing and the copies of the next (the reference's two pipeline steps interleave their lines too) */
				uint64_t calls = 0;
				if (grp) bfcg_group_progress(grp, &calls, 0, 0, 0); else bfcg_progress(ctx, &calls, 0, 0, 0);
				pend_call[n_pend_hi & 63] = calls; pend_seqs[n_pend_hi & 63] = b->n_seqs; ++n_pend_hi;
				print_progress(ctx, grp, opt, t0, pend_call, pend_seqs, &n_pend_lo, n_pend_hi);
			}
		}
		t_submit += now_real() - tt;
		if (b->last) break; /* the last of the pipeline workers has got an empty batch */
		if (!opt->no_mt_io) {
			pthread_mutex_lock(&pp.mtx);
			pp.ready[cur] = 0;
			pthread_cond_broadcast(&pp.cv);
			pthread_mutex_unlock(&pp.mtx);
		}
		cur ^= 1;
	}
	if (!opt->no_mt_io) pthread_join(tid, 0);
	/* the batches still in flight */
	if ((grp ? bfcg_group_sync(grp) : bfcg_sync(ctx)) != 0) { fprintf(stderr, "[E::%s] GPU counting failed: %s\n", __func__, bfcg_last_error()); abort(); }
	print_progress(ctx, grp, opt, t0, pend_call, pend_seqs, &n_pend_lo, n_pend_hi);

	tt = now_real();
	if (grp) ret = opt->filter_mode ? (void*)(getenv("BFC_GPU_NO_RESIDENT") ? bfcg_group_export_bloom(grp, 1) : bfcg_group_export_bloom_resident(grp, 1)) /* all-gathered onto every device for the sharded trim pass */
	                                : (void*)bfcg_group_export_table(grp);
	else ret = opt->filter_mode ? (void*)(getenv("BFC_GPU_NO_RESIDENT") ? bfcg_export_bloom(ctx, 1) : bfcg_export_bloom_resident(ctx, 1)) /* bf_high also stays in HBM for the trim pass (bfc_trim.c) */
	                            : (void*)bfcg_export_table(ctx);
	if (timing) fprintf(stderr, "[T::bfc_count] waited for the parser %.3f s, packed bit planes %.3f s, submitted batches %.3f s, result to the host %.3f s (%d fast / %d serial batches)\n", t_wait, t_pack, t_submit, now_real() - tt, ps.fast_batches, ps.serial_batches);
	if (ret == 0) { fprintf(stderr, "[E::%s] cannot bring the result to the host: %s\n", __func__, bfcg_last_error()); abort(); }
	/* (Round 4 tried to take the clean-up off the path -- buffers pinned by four threads, buffers and input released by a thread of its own under
	 * the export, large tables freed by a detached thread: every one of these contends with the export's own allocations and page faults for the
	 * same locks; the process' wall time did not move: profiles/round4_e2e.md) */
	tt = now_real();
	for (i = 0; i < 2; ++i) {
		if (pin && !use_planes) { bfcg_host_free(pp.b[i].seq); bfcg_host_free(pp.b[i].qual); } else { free(pp.b[i].seq); free(pp.b[i].qual); } /* (NULL where no batch ever needed them) */
		if (pp.planes[i]) { if (pin) bfcg_host_free(pp.planes[i]); else free(pp.planes[i]); }
		free(pp.b[i].kind_cut);
	}
	bfc_pool_destroy(pp.pack_pool);
	pthread_mutex_destroy(&pp.mtx); pthread_cond_destroy(&pp.cv);
	t_wait = now_real() - tt; tt = now_real();
	ingest_close(&ps);
	t_submit = now_real() - tt; tt = now_real();
	if (grp) bfcg_group_destroy(grp); else bfcg_destroy(ctx); /* the first bloom filter dies here, as in count.c:155 */
	if (timing) fprintf(stderr, "[T::bfc_count] clean-up: host buffers %.3f s, input closed %.3f s, GPU context%s destroyed %.3f s; left %.3f s after the process started\n", t_wait, t_submit, grp ? "s" : "", now_real() - tt, now_real() - t0);
	return ret;
}

/* Ingest only (no GPU): parses `fn` into batches exactly as bfc_count does and digests them.  out[0] batches, out[1] reads,
 * out[2] stream positions, out[3] FNV-1a over all sequence streams, out[4] over all quality streams (FASTA: '~'), out[5] over the
 * per-batch read counts (the batch boundaries), out[6] batches that came from the parallel fast path.  n_threads = 0: serial parser. */
int bfc_ingest_digest(const char *fn, uint64_t chunk_size, uint64_t cap, int n_threads, uint64_t out[7])
{
	ingest_t in;
	batch_t b;
	uint64_t hs = 0xcbf29ce484222325ULL, hq = hs, hb = hs, i;
	const int no_hash = getenv("BFC_INGEST_NOHASH") != 0; /* timing the parsers alone (scripts/ingest_rate.py) */
	memset(out, 0, 7 * sizeof(uint64_t));
	if (ingest_open(&in, fn, chunk_size, n_threads, 1) != 0) return -1; /* one worker: the first empty batch ends the input, as a plain bseq_read loop does */
	memset(&b, 0, sizeof(b));
	b.cap = cap; b.seq = (uint8_t*)malloc(cap); b.qual = (uint8_t*)malloc(cap);
	for (;;) {
		ingest_fill(&in, &b);
		if (b.n_seqs) {
			++out[0]; out[1] += (uint64_t)b.n_seqs; out[2] += b.n_pos;
			if (!no_hash) for (i = 0; i < b.n_pos; ++i) { hs = (hs ^ b.seq[i]) * 0x100000001b3ULL; hq = (hq ^ b.qual[i]) * 0x100000001b3ULL; }
			for (i = 0; i < 4; ++i) hb = (hb ^ (((uint64_t)b.n_seqs >> (8 * i)) & 0xff)) * 0x100000001b3ULL;
		}
		if (b.last) break;
	}
	out[3] = hs; out[4] = hq; out[5] = hb; out[6] = (uint64_t)in.fast_batches;
	free(b.seq); free(b.qual); free(b.kind_cut);
	ingest_close(&in);
	return 0;
}

/* The batches' BIT PLANES (no GPU): `direct` = 1 lets the FASTQ fast path write them straight from the mapped file (batch_t.planes), 0 fills byte
 * streams and packs them with bfcg_pack_planes -- the two must agree word for word.  out[0] batches, out[1] positions, out[2..5] FNV-1a over the
 * words [0, ceil(n_pos / 32)) of planes 0..3 of every batch, out[6] batches that were packed directly. */
int bfc_ingest_planes_digest(const char *fn, uint64_t chunk_size, uint64_t cap, int n_threads, int q, int direct, uint64_t out[7])
{
	ingest_t in;
	batch_t b;
	uint64_t h[4], pw = bfcg_plane_words(cap), w;
	uint32_t *pl = (uint32_t*)malloc(pw * 16);
	int p;
	memset(out, 0, 7 * sizeof(uint64_t));
	for (p = 0; p < 4; ++p) h[p] = 0xcbf29ce484222325ULL;
	if (!pl || ingest_open(&in, fn, chunk_size, n_threads, 1) != 0) { free(pl); return -1; }
	memset(&b, 0, sizeof(b));
	b.cap = cap;
	memset(pl, 0xa5, pw * 16); /* (stale words of an earlier batch must not show) */
	if (direct) { b.planes = pl; b.plane_words = pw; b.q = q; }
	for (;;) {
		ingest_fill(&in, &b);
		if (b.n_seqs) {
			const uint64_t nw = (b.n_pos + 31) / 32;
			++out[0]; out[1] += b.n_pos; out[6] += (uint64_t)b.packed;
			if (!b.packed) bfcg_pack_planes(b.seq, b.has_qual ? b.qual : 0, 0, b.n_pos, b.n_pos, q, pl, pw);
			if (!getenv("BFC_INGEST_NOHASH")) for (p = 0; p < (b.has_qual ? 4 : 3); ++p) for (w = 0; w < nw; ++w) {
				const uint32_t v = pl[(uint64_t)p * pw + w];
				int k;
				for (k = 0; k < 4; ++k) h[p] = (h[p] ^ ((v >> (8 * k)) & 0xff)) * 0x100000001b3ULL;
			}
		}
		if (b.last) break;
	}
	for (p = 0; p < 4; ++p) out[2 + p] = h[p];
	free(b.seq); free(b.qual); free(b.kind_cut); free(pl);
	ingest_close(&in);
	return 0;
}

/* The parallel inflate alone (bfc_pgz.h; no GPU, no parser): the whole of `fn` in windows of `window` bytes.  out[0] text bytes,
 * out[1] their CRC-32, out[2] pieces taken as the threads guessed them, out[3] pieces decoded again from the known position,
 * out[4] rounds.  Returns 0, -1 if the file cannot be mapped or is not gzip, -2 if the stream cannot be decoded (bfc_count then reads
 * it through gzread). */
int bfc_pgz_digest(const char *fn, int n_threads, uint64_t chunk, uint64_t window, uint64_t out[5])
{
	struct stat st;
	int fd = open(fn, O_RDONLY), rc = 0;
	void *m;
	pgz_t *g;
	uint64_t pos = 0;
	uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
	memset(out, 0, 5 * sizeof(uint64_t));
	if (fd < 0) return -1;
	if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 18) { close(fd); return -1; }
	m = mmap(0, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
	close(fd);
	if (m == MAP_FAILED) return -1;
	if (((const uint8_t*)m)[0] != 0x1f || ((const uint8_t*)m)[1] != 0x8b) { munmap(m, (size_t)st.st_size); return -1; }
	g = pgz_open((const uint8_t*)m, (size_t)st.st_size, n_threads, (size_t)chunk);
	for (;;) {
		const uint8_t *p; uint64_t avail, o; int eof;
		if (pgz_ensure(g, pos, window, &p, &avail, &eof) != 0) { rc = -2; break; }
		if (!getenv("BFC_INGEST_NOHASH")) for (o = pos; o < avail;) { const uint64_t step = avail - o < (1u << 30) ? avail - o : (1u << 30); crc = (uint32_t)crc32(crc, p + o, (uInt)step); o += step; }
		pos = avail;
		if (eof) break;
	}
	out[0] = pos; out[1] = crc; out[2] = g->n_spec; out[3] = g->n_redo; out[4] = g->n_rounds;
	pgz_close(g);
	munmap(m, (size_t)st.st_size);
	return rc;
}
