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
 * (count.c:85), expressed here by writing '~' into the quality stream.
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

/* ------------------------------------------------------------------ line reader over zlib */

typedef struct {
	gzFile fp;
	uint8_t *buf; int begin, end, eof;
	uint8_t *line; size_t l_line, m_line;
	int pending;     /* a header line already read into `line` */
	int failed;
} reader_t;

#define RD_BUF (1 << 20)

static int rd_fill(reader_t *r)
{
	if (r->eof) return 0;
	r->begin = 0;
	r->end = gzread(r->fp, r->buf, RD_BUF);
	if (r->end < RD_BUF) r->eof = 1;
	if (r->end < 0) r->end = 0;
	return r->end;
}
/* next line without its '\n' into r->line; returns 0 at end of input */
static int rd_line(reader_t *r)
{
	int got = 0;
	r->l_line = 0;
	for (;;) {
		uint8_t *p, *q;
		size_t n;
		if (r->begin >= r->end && rd_fill(r) == 0) return got;
		got = 1;
		p = r->buf + r->begin;
		q = (uint8_t*)memchr(p, '\n', (size_t)(r->end - r->begin));
		n = q ? (size_t)(q - p) : (size_t)(r->end - r->begin);
		if (r->l_line + n + 1 > r->m_line) { r->m_line = (r->l_line + n + 1) * 2; r->line = (uint8_t*)realloc(r->line, r->m_line); }
		memcpy(r->line + r->l_line, p, n);
		r->l_line += n;
		r->begin += (int)n + (q ? 1 : 0);
		if (q) return 1;
	}
}

/* ------------------------------------------------------------------ batches */

typedef struct {
	uint8_t *seq, *qual;   /* pinned */
	uint64_t n_pos, cap;
	int n_seqs, has_qual, last;
} batch_t;

/* append one record to the batch; returns 0 if it does not fit */
static int batch_put(batch_t *b, const uint8_t *s, const uint8_t *q, size_t l)
{
	if (b->n_pos + l + 1 > b->cap) return 0;
	memcpy(b->seq + b->n_pos, s, l);
	if (q) { memcpy(b->qual + b->n_pos, q, l); b->has_qual = 1; }
	else memset(b->qual + b->n_pos, '~', l);
	b->seq[b->n_pos + l] = '\n'; b->qual[b->n_pos + l] = '!';
	b->n_pos += l + 1; ++b->n_seqs;
	return 1;
}

typedef struct {
	reader_t rd;
	uint8_t *seq, *qual; size_t l_seq, m_seq, l_qual, m_qual; /* record being assembled */
	int have_rec, rec_has_qual;
	uint64_t chunk_size;
} parser_t;

static void app(uint8_t **s, size_t *l, size_t *m, const uint8_t *p, size_t n)
{
	if (*l + n + 1 > *m) { *m = (*l + n + 1) * 2; *s = (uint8_t*)realloc(*s, *m); }
	memcpy(*s + *l, p, n); *l += n;
}

/* parse the next record into ps->seq/qual; 1 = record, 0 = end of input */
static int next_record(parser_t *ps)
{
	reader_t *r = &ps->rd;
	if (r->failed) return 0;
	if (!r->pending) { /* jump to the next header line */
		for (;;) {
			if (!rd_line(r)) return 0;
			if (r->l_line && (r->line[0] == '>' || r->line[0] == '@')) break;
		}
	}
	r->pending = 0;
	ps->l_seq = ps->l_qual = 0; ps->rec_has_qual = 0;
	for (;;) { /* sequence lines */
		if (!rd_line(r)) return 1; /* FASTA record ended by EOF */
		if (r->l_line == 0) continue;
		if (r->line[0] == '>' || r->line[0] == '@') { r->pending = 1; return 1; }
		if (r->line[0] == '+') break;
		app(&ps->seq, &ps->l_seq, &ps->m_seq, r->line, r->l_line);
	}
	ps->rec_has_qual = 1;
	while (ps->l_qual < ps->l_seq) { /* quality lines (the '+' line itself is already consumed) */
		if (!rd_line(r)) break;
		app(&ps->qual, &ps->l_qual, &ps->m_qual, r->line, r->l_line);
	}
	if (ps->l_qual != ps->l_seq) { r->failed = 1; return 0; } /* kseq: -2, bseq_read stops */
	return 1;
}

/* fill one batch: reads until at least chunk_size bases (bseq.c:52-76) or the buffer is full */
static void fill_batch(parser_t *ps, batch_t *b)
{
	uint64_t bases = 0;
	b->n_pos = 0; b->n_seqs = 0; b->has_qual = 0; b->last = 0;
	for (;;) {
		if (!ps->have_rec) {
			if (!next_record(ps)) { b->last = 1; return; }
			ps->have_rec = 1;
		}
		if (ps->l_seq + 1 > b->cap) {
			fprintf(stderr, "[E::bfc_count] a read of %zu bases does not fit a GPU batch of %llu positions\n", ps->l_seq, (unsigned long long)b->cap);
			abort();
		}
		if (!batch_put(b, ps->seq, ps->rec_has_qual ? ps->qual : 0, ps->l_seq)) return; /* keep the record for the next batch */
		ps->have_rec = 0;
		bases += ps->l_seq;
		if (bases >= ps->chunk_size) return;
	}
}

/* ------------------------------------------------------------------ reader thread, one batch ahead */

typedef struct {
	parser_t *ps;
	batch_t b[2];
	int ready[2];   /* filled and not yet consumed */
	int done;
	pthread_mutex_t mtx; pthread_cond_t cv;
} pipe_t;

static void *reader_main(void *arg)
{
	pipe_t *pp = (pipe_t*)arg;
	int i = 0;
	for (;;) {
		pthread_mutex_lock(&pp->mtx);
		while (pp->ready[i]) pthread_cond_wait(&pp->cv, &pp->mtx);
		pthread_mutex_unlock(&pp->mtx);
		fill_batch(pp->ps, &pp->b[i]);
		pthread_mutex_lock(&pp->mtx);
		pp->ready[i] = 1;
		pthread_cond_broadcast(&pp->cv);
		pthread_mutex_unlock(&pp->mtx);
		if (pp->b[i].last) break;
		i ^= 1;
	}
	return 0;
}

/* ------------------------------------------------------------------ bfc_count */

void *bfc_count(const char *fn, const bfc_opt_t *opt)
{
	bfcg_params_t prm;
	bfcg_ctx_t *ctx;
	parser_t ps;
	pipe_t pp;
	pthread_t tid;
	void *ret;
	const char *env;
	double t0 = (&bfc_real_time && bfc_real_time > 0.) ? bfc_real_time : now_real();
	uint64_t cap;
	int i, cur = 0;
	uint64_t st[BFCG_ST_N];

	bfcg_params_default(&prm);
	prm.k = opt->k; prm.q = opt->q; prm.bf_shift = opt->bf_shift; prm.n_hashes = opt->n_hashes;
	prm.l_pre = opt->l_pre; prm.filter_mode = opt->filter_mode;
	prm.device = (env = getenv("BFC_GPU_DEVICE")) ? atoi(env) : 0;
	/* a batch holds one reference chunk (opt->chunk_size bases, bfc.c:20,-L) plus separators;
	 * BFC_GPU_BATCH overrides the number of positions per GPU batch */
	cap = (uint64_t)(opt->chunk_size > 0 ? opt->chunk_size : 100000000);
	if ((env = getenv("BFC_GPU_BATCH")) != 0) cap = strtoull(env, 0, 10);
	if (cap < (1u << 16)) cap = 1u << 16;
	cap += cap / 64 + (1u << 20);
	if (cap >= (1ULL << 32)) cap = (1ULL << 32) - 1;
	prm.max_batch_pos = cap;
	ctx = bfcg_create(&prm);
	if (ctx == 0) { fprintf(stderr, "[E::%s] cannot set up the GPU count path: %s\n", __func__, bfcg_last_error()); abort(); }

	memset(&ps, 0, sizeof(ps));
	ps.chunk_size = (uint64_t)(opt->chunk_size > 0 ? opt->chunk_size : 100000000);
	if (ps.chunk_size > cap - cap / 32) ps.chunk_size = cap - cap / 32;
	ps.rd.fp = fn && strcmp(fn, "-") ? gzopen(fn, "r") : gzdopen(fileno(stdin), "r");
	if (ps.rd.fp == 0) { fprintf(stderr, "[E::%s] cannot open '%s'\n", __func__, fn ? fn : "-"); abort(); }
	gzbuffer(ps.rd.fp, 1 << 18);
	ps.rd.buf = (uint8_t*)malloc(RD_BUF);

	memset(&pp, 0, sizeof(pp));
	pp.ps = &ps;
	pthread_mutex_init(&pp.mtx, 0); pthread_cond_init(&pp.cv, 0);
	for (i = 0; i < 2; ++i) {
		pp.b[i].cap = cap;
		pp.b[i].seq = (uint8_t*)bfcg_host_alloc(cap); pp.b[i].qual = (uint8_t*)bfcg_host_alloc(cap);
		if (!pp.b[i].seq || !pp.b[i].qual) { fprintf(stderr, "[E::%s] cannot pin %llu bytes of host memory\n", __func__, (unsigned long long)cap); abort(); }
	}
	if (!opt->no_mt_io) pthread_create(&tid, 0, reader_main, &pp);

	for (;;) {
		batch_t *b = &pp.b[cur];
		if (opt->no_mt_io) fill_batch(&ps, b);
		else {
			pthread_mutex_lock(&pp.mtx);
			while (!pp.ready[cur]) pthread_cond_wait(&pp.cv, &pp.mtx);
			pthread_mutex_unlock(&pp.mtx);
		}
		if (b->n_seqs) {
			double rt, eff;
			fprintf(stderr, "[M::%s] read %d sequences\n", "bfc_count_cb", b->n_seqs);
			if (bfcg_count_batch_host(ctx, b->seq, b->has_qual ? b->qual : 0, b->n_pos) != 0) {
				fprintf(stderr, "[E::%s] GPU counting failed: %s\n", __func__, bfcg_last_error()); abort();
			}
			bfcg_stats(ctx, st);
			rt = now_real() - t0; eff = 100. * now_cpu() / (rt + 1e-6);
			if (!opt->filter_mode)
				fprintf(stderr, "[M::%s @%.1f*%.1f%%] processed %d sequences; # distinct k-mers: %ld\n", "bfc_count_cb", rt, eff, b->n_seqs, (long)st[BFCG_ST_KEYS]);
			else
				fprintf(stderr, "[M::%s @%.1f*%.1f%%] processed %d sequences\n", "bfc_count_cb", rt, eff, b->n_seqs);
		}
		if (b->last) break;
		if (!opt->no_mt_io) {
			pthread_mutex_lock(&pp.mtx);
			pp.ready[cur] = 0;
			pthread_cond_broadcast(&pp.cv);
			pthread_mutex_unlock(&pp.mtx);
		}
		cur ^= 1;
	}
	if (!opt->no_mt_io) pthread_join(tid, 0);

	ret = opt->filter_mode ? (void*)bfcg_export_bloom(ctx, 1) : (void*)bfcg_export_table(ctx);
	if (ret == 0) { fprintf(stderr, "[E::%s] cannot bring the result to the host: %s\n", __func__, bfcg_last_error()); abort(); }
	for (i = 0; i < 2; ++i) { bfcg_host_free(pp.b[i].seq); bfcg_host_free(pp.b[i].qual); }
	pthread_mutex_destroy(&pp.mtx); pthread_cond_destroy(&pp.cv);
	gzclose(ps.rd.fp);
	free(ps.rd.buf); free(ps.rd.line); free(ps.seq); free(ps.qual);
	bfcg_destroy(ctx); /* the first bloom filter dies here, as in count.c:155 */
	return ret;
}
