/* bfc_trim.c -- `bfc -1` trim pass behind the reference's own entry point
 *     void bfc_correct(const char *fn, const bfc_opt_t *opt, const void *ptr)        (bfc.h:40, correct.c:620)
 * for opt->filter_mode: per read, one bloom query per k-mer (max_streak, correct.c:478-497), keep the longest streak if
 * (streak + k) / length > min_frac (correct.c:557-569), print kept reads (correct.c:605-611).  The queries and the streak
 * scan run on the GPU (bfcg_trim_batch); the host parses and prints.
 *
 * An unmodified correct.c can never call a batched kernel, so this object provides `bfc_correct` itself; the reference's
 * corrector is linked under another name (compile correct.c with -Dbfc_correct=bfc_correct_cpu, source untouched, see
 * INTEGRATION.md) and is what this function forwards to when filter_mode is off.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <unistd.h>
#include <sys/time.h>
#include <sys/resource.h>
#include "bfc_gpu.h"

extern double bfc_real_time __attribute__((weak));
extern int bfc_verbose __attribute__((weak));
void bfc_correct_cpu(const char *fn, const bfc_opt_t *opt, const void *ptr) __attribute__((weak));
void *bfcg_host_alloc(uint64_t bytes);
void bfcg_host_free(void *p);

static double t_real(void) { struct timeval tp; gettimeofday(&tp, 0); return tp.tv_sec + tp.tv_usec * 1e-6; }
static double t_cpu(void)
{
	struct rusage r; getrusage(RUSAGE_SELF, &r);
	return r.ru_utime.tv_sec + r.ru_stime.tv_sec + 1e-6 * (r.ru_utime.tv_usec + r.ru_stime.tv_usec);
}

#include "bfc_ingest.h"
#include <pthread.h>

int bfcg_env_devices(int *dev, int max); /* bfc_count.c: BFC_GPU_DEVICES */

/* the trim pass is embarrassingly parallel over reads: with several GPUs every batch's reads are dealt to them in
 * contiguous ranges, one host thread per device; each device holds the whole of bf_high (left there by bfc_count, or uploaded) */
typedef struct { bfcg_trim_t *tr; const uint8_t *seq; uint64_t n_pos; uint64_t *off; uint64_t n; float min_frac; int32_t *st, *en; int rc; char err[256]; } trim_job_t;
static void *trim_worker(void *p)
{
	trim_job_t *j = (trim_job_t*)p;
	j->rc = j->n ? bfcg_trim_batch(j->tr, j->seq, 0, j->n_pos, j->off, j->n, j->min_frac, j->st, j->en) : 0;
	if (j->rc != 0) { strncpy(j->err, bfcg_last_error(), sizeof(j->err) - 1); j->err[sizeof(j->err) - 1] = 0; } /* the message is thread-local: it dies with this thread */
	return 0;
}

typedef struct { uint64_t off_hdr, off_cmt; int has_comment, has_qual; } rinfo_t; /* name and comment (as bseq_read copied them) in hdrs[] */

void bfc_correct(const char *fn, const bfc_opt_t *opt, const void *ptr)
{
	const bfc_bf_t *bf = (const bfc_bf_t*)ptr;
	parser_t ps;
	batch_t b;
	bfcg_trim_t *tr, *trs[64];
	int devs[64], n_dev, d;
	uint64_t cap, max_reads, *off, *off2 = 0;
	int32_t *st, *en;
	rinfo_t *ri;
	char *hdrs = 0; size_t l_hdrs, m_hdrs = 0;
	const char *env;
	double t0 = (&bfc_real_time && bfc_real_time > 0.) ? bfc_real_time : t_real();

	if (!opt->filter_mode) {
		if (bfc_correct_cpu) { bfc_correct_cpu(fn, opt, ptr); return; }
		fprintf(stderr, "[E::%s] error correction is the reference's correct.c: link it as bfc_correct_cpu (INTEGRATION.md)\n", __func__);
		abort();
	}
	if (!(&bfc_verbose) || bfc_verbose >= 3)
		fprintf(stderr, "[M::%s @%.1f*%.1f%%] Starting...\n", __func__, t_real() - t0, 100. * t_cpu() / (t_real() - t0 + 1e-6));
	cap = (uint64_t)(opt->chunk_size > 0 ? opt->chunk_size : 100000000);
	if ((env = getenv("BFC_GPU_BATCH")) != 0) cap = strtoull(env, 0, 10);
	if (cap < (1u << 16)) cap = 1u << 16;
	cap += cap / 64 + (1u << 20);
	max_reads = cap / 16 + 1024;
	n_dev = bfcg_env_devices(devs, 64);
	if (n_dev == 0) { n_dev = 1; devs[0] = (env = getenv("BFC_GPU_DEVICE")) ? atoi(env) : 0; }
	for (d = 0; d < n_dev; ++d) {
		int dup = 0, j;
		for (j = 0; j < d; ++j) if (devs[j] == devs[d]) dup = 1; /* a device named twice: its first context serves both shares' turns */
		(void)dup;
		/* a device's share of a batch: 1/N of its positions (cut at the nearest read boundary) -- and up to all of its reads, if they are short there */
		trs[d] = bfcg_trim_create(opt->k, bf, devs[d], n_dev > 1 ? cap / (uint64_t)n_dev + cap / 64 + (1u << 16) : cap, max_reads);
		if (!trs[d]) { fprintf(stderr, "[E::%s] cannot set up the GPU trim pass: %s\n", __func__, bfcg_last_error()); abort(); }
	}
	tr = trs[0];
	if (n_dev > 1) off2 = (uint64_t*)malloc((max_reads + 1 + (uint64_t)n_dev) * 8);

	memset(&ps, 0, sizeof(ps));
	ps.keep_hdr = 1;
	ps.chunk_size = (uint64_t)(opt->chunk_size > 0 ? opt->chunk_size : 100000000);
	if (ps.chunk_size > cap - cap / 32) ps.chunk_size = cap - cap / 32;
	ps.rd.fp = fn && strcmp(fn, "-") ? gzopen(fn, "r") : gzdopen(fileno(stdin), "r");
	if (ps.rd.fp == 0) { fprintf(stderr, "[E::%s] cannot open '%s'\n", __func__, fn ? fn : "-"); abort(); }
	/* (no gzbuffer: zlib is asked for kseq's own 16 KiB pieces through its default buffers, so that a damaged gzip file ends where it ends for
	 * the reference -- rd_fill in bfc_ingest.h) */
	ps.rd.buf = (uint8_t*)malloc(RD_BUF);
	memset(&b, 0, sizeof(b));
	b.cap = cap;
	b.seq = (uint8_t*)bfcg_host_alloc(cap); b.qual = (uint8_t*)malloc(cap);
	off = (uint64_t*)malloc((max_reads + 1) * 8); st = (int32_t*)malloc(max_reads * 4); en = (int32_t*)malloc(max_reads * 4);
	ri = (rinfo_t*)malloc(max_reads * sizeof(rinfo_t));
	if (!b.seq || !b.qual || !off || !st || !en || !ri) { fprintf(stderr, "[E::%s] out of memory\n", __func__); abort(); }

	int empties = 0;
	for (;;) { /* one batch: parse (keeping headers), trim on the GPU, print */
		uint64_t bases = 0, r, n = 0;
		int last = 0;
		batch_clear(&b); l_hdrs = 0;
		off[0] = 0;
		for (;;) {
			if (!ps.have_rec) { /* bseq_read (bseq.c:52-76): the batch ends at the end of the input or at a malformed record; an empty batch is the last */
				int rc = next_record(&ps);
				if (rc <= 0) break;
				ps.have_rec = 1;
			}
			if (ps.l_seq + 1 > b.cap) { fprintf(stderr, "[E::%s] a read of %zu bases does not fit a GPU batch\n", __func__, ps.l_seq); abort(); }
			if (n == max_reads || !batch_put(&b, ps.seq, ps.rec_has_qual ? ps.qual : 0, ps.l_seq)) break;
			ps.have_rec = 0;
			if (l_hdrs + ps.l_hdr + ps.l_cmt + 2 > m_hdrs) { m_hdrs = (l_hdrs + ps.l_hdr + ps.l_cmt + 2) * 2; hdrs = (char*)realloc(hdrs, m_hdrs); }
			memcpy(hdrs + l_hdrs, ps.hdr, ps.l_hdr + 1);
			ri[n].off_hdr = l_hdrs; l_hdrs += ps.l_hdr + 1;
			ri[n].has_comment = ps.have_cmt; ri[n].has_qual = ps.rec_has_qual; ri[n].off_cmt = l_hdrs;
			if (ps.have_cmt) { memcpy(hdrs + l_hdrs, ps.cmt, ps.l_cmt + 1); l_hdrs += ps.l_cmt + 1; } /* bseq.c:64: whatever kseq's comment buffer holds now */
			off[++n] = b.n_pos;
			bases += ps.l_seq;
			if (bases >= ps.chunk_size) break;
		}
		fprintf(stderr, "[M::%s] read %d sequences\n", "bfc_ec_cb", (int)n); /* correct.c:582, once per bseq_read call */
		if (n == 0 && ++empties >= (opt->no_mt_io ? 1 : 2)) last = 1; /* each of the pipeline's workers ends on its own empty batch (kthread.c:88-106, correct.c:644) */
		if (n) {
			if (n_dev == 1) {
				if (bfcg_trim_batch(tr, b.seq, 0, b.n_pos, off, n, opt->min_frac, st, en) != 0) {
					fprintf(stderr, "[E::%s] GPU trim pass failed: %s\n", __func__, bfcg_last_error()); abort();
				}
			} else { /* device d takes the reads up to the boundary nearest to d+1 N-ths of the batch's POSITIONS (the contexts are sized by positions:
			          * dealt by read count, a batch of reads sorted by length would overflow one of them): their part of the stream, offsets rebased */
				trim_job_t job[64];
				pthread_t th[64];
				uint64_t o2 = 0, r1 = 0;
				for (d = 0; d < n_dev; ++d) {
					const uint64_t r0 = r1, want = d + 1 == n_dev ? b.n_pos : b.n_pos / (uint64_t)n_dev * (uint64_t)(d + 1);
					uint64_t q;
					if (d + 1 == n_dev) r1 = n;
					else { /* first boundary at or behind `want` (off[] ascends), or the one before it if that is nearer */
						uint64_t lo = r0, hi = n;
						while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (off[mid] < want) lo = mid + 1; else hi = mid; }
						r1 = lo;
						if (r1 > r0 && off[r1] - want > want - off[r1 - 1]) --r1;
					}
					job[d].tr = trs[d]; job[d].seq = b.seq + off[r0]; job[d].n_pos = off[r1] - off[r0]; job[d].n = r1 - r0;
					job[d].off = off2 + o2; job[d].min_frac = opt->min_frac; job[d].st = st + r0; job[d].en = en + r0; job[d].rc = 0;
					for (q = r0; q <= r1; ++q) off2[o2++] = off[q] - off[r0];
					pthread_create(&th[d], 0, trim_worker, &job[d]);
				}
				for (d = 0; d < n_dev; ++d) {
					pthread_join(th[d], 0);
					if (job[d].rc != 0) { fprintf(stderr, "[E::%s] GPU trim pass failed on device %d: %s\n", __func__, devs[d], job[d].err); abort(); }
				}
			}
			fprintf(stderr, "[M::%s @%.1f*%.1f%%] processed %d sequences\n", "bfc_ec_cb", t_real() - t0, 100. * t_cpu() / (t_real() - t0 + 1e-6), (int)n);
			for (r = 0; r < n; ++r) { /* correct.c:595-611 */
				char *h = hdrs + ri[r].off_hdr;
				int is_fq = ri[r].has_qual && !opt->no_qual;
				if (st[r] < 0) continue;
				putchar(is_fq ? '@' : '>');
				fputs(h, stdout);
				if (ri[r].has_comment) { putchar('\t'); fputs(hdrs + ri[r].off_cmt, stdout); }
				putchar('\n');
				fwrite(b.seq + off[r] + st[r], 1, (size_t)(en[r] - st[r]), stdout); putchar('\n');
				if (is_fq) { puts("+"); fwrite(b.qual + off[r] + st[r], 1, (size_t)(en[r] - st[r]), stdout); putchar('\n'); }
			}
		}
		if (last) break;
	}
	for (d = 0; d < n_dev; ++d) bfcg_trim_destroy(trs[d]);
	free(off2);
	gzclose(ps.rd.fp);
	free(ps.rd.buf); free(ps.rd.line); free(ps.seq); free(ps.qual); free(ps.hdr); free(ps.cmt);
	bfcg_host_free(b.seq); free(b.qual); free(b.kind_cut); free(off); free(st); free(en); free(ri); free(hdrs);
}
