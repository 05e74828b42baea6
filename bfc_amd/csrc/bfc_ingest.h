/* bfc_ingest.h -- FASTA/FASTQ ingest shared by bfc_count.c (count phase) and bfc_trim.c (trim pass).
 * Record grammar follows kseq.h:185-224 as bseq_read (bseq.c:52-76) uses it: '>' or '@' header, sequence lines until
 * a line starting with '+', '>' or '@'; after '+', quality lines until at least as many characters as bases; a length
 * mismatch ends the input (kseq returns -2 and bseq_read stops). */
#ifndef BFC_INGEST_H
#define BFC_INGEST_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* ------------------------------------------------------------------ line reader over zlib */

typedef struct {
	gzFile fp;
	uint8_t *buf; int begin, end, eof;
	uint8_t *line; size_t l_line, m_line;
	int pending;     /* a header line already read into `line` */
	int failed;
} reader_t;

#define RD_BUF (1 << 20)

static inline int rd_fill(reader_t *r)
{
	if (r->eof) return 0;
	r->begin = 0;
	r->end = gzread(r->fp, r->buf, RD_BUF);
	if (r->end < RD_BUF) r->eof = 1;
	if (r->end < 0) r->end = 0;
	return r->end;
}
/* next line without its '\n' into r->line; returns 0 at end of input */
static inline int rd_line(reader_t *r)
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
static inline int batch_put(batch_t *b, const uint8_t *s, const uint8_t *q, size_t l)
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
	uint8_t *hdr; size_t l_hdr, m_hdr;  /* header line of the record without its '>' / '@' (name [whitespace comment]) */
	int keep_hdr;
	int have_rec, rec_has_qual;
	uint64_t chunk_size;
} parser_t;

static inline void app(uint8_t **s, size_t *l, size_t *m, const uint8_t *p, size_t n)
{
	if (*l + n + 1 > *m) { *m = (*l + n + 1) * 2; *s = (uint8_t*)realloc(*s, *m); }
	memcpy(*s + *l, p, n); *l += n;
}

/* parse the next record into ps->seq/qual; 1 = record, 0 = end of input */
static inline int next_record(parser_t *ps)
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
	if (ps->keep_hdr) { ps->l_hdr = 0; app(&ps->hdr, &ps->l_hdr, &ps->m_hdr, r->line + 1, r->l_line - 1); ps->hdr[ps->l_hdr] = 0; }
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
static inline void fill_batch(parser_t *ps, batch_t *b)
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


#endif
