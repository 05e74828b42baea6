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
#include <time.h>

/* ------------------------------------------------------------------ line reader over zlib */

struct psrc_s;
typedef struct {
	gzFile fp;
	struct psrc_s *mem; uint64_t mem_pos; /* pipe input (psrc_t below): the bytes come from the pipe reader's ring, from stream offset mem_pos on, not from gzread */
	uint8_t *buf; int begin, end, eof;
	uint64_t total;  /* bytes delivered by gzread so far */
	int first_piece; /* size of the next gzread if it is not RD_PIECE */
	uint8_t *line; size_t l_line, m_line;
	int line_nl;     /* the line just read ended with '\n' (0: the input ended first) */
	int pending;     /* a header line already read into `line`; hdr_at = index of its '>' / '@' */
	size_t hdr_at;
} reader_t;

#define RD_BUF (1 << 20)
#define RD_PIECE 16384

static inline int psrc_take(struct psrc_s *ps, uint64_t pos, uint8_t *dst, int want);
static inline int rd_fill(reader_t *r)
{
	if (r->eof) return 0;
	r->begin = 0; r->end = 0;
	if (r->mem) { /* what gzread's loop below delivers for plain text: RD_BUF bytes, fewer only at the end of the input */
		r->end = psrc_take(r->mem, r->mem_pos, r->buf, RD_BUF - RD_BUF % RD_PIECE);
		r->mem_pos += (uint64_t)r->end;
		if (r->end < RD_BUF - RD_BUF % RD_PIECE) r->eof = 1;
		r->total += (uint64_t)r->end;
		return r->end;
	}
	/* gzread in kseq's own pieces (kseq.h:72-74,104-106: 16384 bytes a call, a short or failed read ends the input): what zlib hands out
	 * before it reports a damaged gzip stream depends on the sizes it is asked for */
	while (r->end + RD_PIECE <= RD_BUF) {
		const int want = r->first_piece ? r->first_piece : RD_PIECE; /* behind a seek: up to kseq's next buffer boundary */
		const int n = gzread(r->fp, r->buf + r->end, (unsigned)want);
		r->first_piece = 0;
		if (n > 0) r->end += n;
		if (n < want) { r->eof = 1; break; }
	}
	r->total += (uint64_t)r->end;
	return r->end;
}
/* next line without its '\n' into r->line; returns 0 at end of input */
static inline int rd_line(reader_t *r)
{
	int got = 0;
	r->l_line = 0; r->line_nl = 0;
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
		if (q) { r->line_nl = 1; return 1; }
	}
}

#include "bfc_planes.h"
#include <sys/mman.h>

/* ------------------------------------------------------------------ batches */

typedef struct {
	uint8_t *seq, *qual;   /* pinned */
	uint64_t n_pos, cap;
	int n_seqs, has_qual, last;
	/* a batch that mixes records with and without qualities (FASTA records in a FASTQ file): stream offsets at which the kind changes.
	 * Records without qualities are always high quality (count.c:85: qual == NULL), whatever -q says -- no in-band quality byte can
	 * say that for every q, so bfc_count submits such a batch as its homogeneous runs (batch boundaries never change results). */
	uint64_t *kind_cut; int n_cut, m_cut, n_noq, last_kind; /* last_kind: -1 none yet, 0 no qualities, 1 qualities */
	/* planes != NULL: the caller wants the batch as bit planes (bfc_planes.h; 4 planes of plane_words words, threshold q).  The FASTQ fast path
	 * then writes them straight from the mapped file -- packed = 1, seq / qual are not touched and need not exist --; a batch the serial
	 * parser fills comes as byte streams (packed = 0; allocated here on first need), which the caller packs itself. */
	uint32_t *planes; uint64_t plane_words; int q, packed;
} batch_t;

static inline int batch_need_streams(batch_t *b)
{
	if (b->seq && b->qual) return 1;
	{
		void *p = 0, *r = 0;
		const uint64_t rounded = (b->cap + (2u << 20) - 1) & ~(uint64_t)((2u << 20) - 1);
		if (posix_memalign(&p, 2u << 20, rounded) != 0 || posix_memalign(&r, 2u << 20, rounded) != 0) { free(p); return 0; }
#ifdef MADV_HUGEPAGE
		(void)madvise(p, rounded, MADV_HUGEPAGE); (void)madvise(r, rounded, MADV_HUGEPAGE);
#endif
		b->seq = (uint8_t*)p; b->qual = (uint8_t*)r;
	}
	return 1;
}

static inline void batch_clear(batch_t *b) { b->n_pos = 0; b->n_seqs = 0; b->has_qual = 0; b->last = 0; b->n_cut = 0; b->n_noq = 0; b->last_kind = -1; b->packed = 0; }

/* append one record to the batch; returns 0 if it does not fit */
static inline int batch_put(batch_t *b, const uint8_t *s, const uint8_t *q, size_t l)
{
	if (b->n_pos + l + 1 > b->cap) return 0;
	if (b->last_kind >= 0 && b->last_kind != (q != 0)) {
		if (b->n_cut == b->m_cut) { b->m_cut = b->m_cut ? b->m_cut * 2 : 16; b->kind_cut = (uint64_t*)realloc(b->kind_cut, sizeof(uint64_t) * (size_t)b->m_cut); }
		b->kind_cut[b->n_cut++] = b->n_pos;
	}
	b->last_kind = q != 0; b->n_noq += q == 0;
	if (l) memcpy(b->seq + b->n_pos, s, l); /* an empty record may come with s == NULL (no sequence line was ever buffered) */
	if (q) { if (l) memcpy(b->qual + b->n_pos, q, l); b->has_qual = 1; }
	else memset(b->qual + b->n_pos, '~', l);
	b->seq[b->n_pos + l] = '\n'; b->qual[b->n_pos + l] = '!';
	b->n_pos += l + 1; ++b->n_seqs;
	return 1;
}

typedef struct {
	reader_t rd;
	uint8_t *seq, *qual; size_t l_seq, m_seq, l_qual, m_qual; /* record being assembled */
	uint8_t *hdr; size_t l_hdr, m_hdr;  /* name of the record (kseq_t.name): the header up to the first white space */
	uint8_t *cmt; size_t l_cmt, m_cmt;  /* kseq_t.comment: the rest of the LAST header line that had one -- kseq leaves it untouched when a
	                                     * header has no comment, and bseq_read copies whatever is there (bseq.c:64) */
	int have_cmt;                       /* comment.s != NULL */
	int keep_hdr;
	int have_rec, rec_has_qual;
	uint64_t chunk_size;
} parser_t;

static inline void app(uint8_t **s, size_t *l, size_t *m, const uint8_t *p, size_t n)
{
	if (*l + n + 1 > *m) { *m = (*l + n + 1) * 2; *s = (uint8_t*)realloc(*s, *m); }
	memcpy(*s + *l, p, n); *l += n;
}

/* kseq_read (kseq.h:185-224) line by line.  Returns 1 = a record in ps->seq/qual (possibly of length 0), 0 = end of input (kseq: -1),
 * -2 = a FASTQ record whose quality string is missing or of another length (kseq: -2; the caller ends its batch there, the next call
 * goes on scanning after it, exactly as repeated bseq_read calls do). */
static inline int next_record(parser_t *ps)
{
	reader_t *r = &ps->rd;
	if (!r->pending) { /* kseq.h:190-194: jump to the next '>' or '@' -- anywhere, not only at the start of a line */
		for (;;) {
			uint8_t *a, *b;
			if (!rd_line(r)) return 0;
			a = (uint8_t*)memchr(r->line, '>', r->l_line); b = (uint8_t*)memchr(r->line, '@', r->l_line);
			if (a == 0 || (b != 0 && b < a)) a = b;
			if (a) { r->hdr_at = (size_t)(a - r->line); break; }
		}
	}
	r->pending = 0;
	ps->l_seq = ps->l_qual = 0; ps->rec_has_qual = 0;
	if (ps->keep_hdr) { /* kseq.h:195-196: name = up to the first isspace() character; unless that was the line end, comment = the rest */
		const uint8_t *h = r->line + r->hdr_at + 1;
		const size_t lh = r->l_line - r->hdr_at - 1;
		size_t j = 0;
		while (j < lh && !(h[j] == ' ' || (h[j] >= '\t' && h[j] <= '\r'))) ++j;
		ps->l_hdr = 0; app(&ps->hdr, &ps->l_hdr, &ps->m_hdr, h, j); ps->hdr[ps->l_hdr] = 0;
		if (j < lh) { /* a delimiter other than '\n' (a '\r' of a "\r\n" line end counts: the comment is then the empty string) */
			ps->l_cmt = 0; app(&ps->cmt, &ps->l_cmt, &ps->m_cmt, h + j + 1, lh - j - 1);
			if (ps->l_cmt > 1 && ps->cmt[ps->l_cmt - 1] == '\r') --ps->l_cmt;
			ps->cmt[ps->l_cmt] = 0; ps->have_cmt = 1;
		}
	}
	if (r->hdr_at + 1 >= r->l_line && !r->line_nl) return 0; /* kseq.h:195: the input ends right behind the header character */
	for (;;) { /* sequence lines (kseq.h:201-205): decided by the first character of each line */
		if (!rd_line(r)) return 1; /* FASTA record ended by the end of the input */
		if (r->l_line == 0) continue;
		if (r->line[0] == '>' || r->line[0] == '@') { r->pending = 1; r->hdr_at = 0; return 1; }
		if (r->line[0] == '+') break;
		app(&ps->seq, &ps->l_seq, &ps->m_seq, r->line, r->l_line);
		/* kseq.h:138 ("\r\n" line ends), on the accumulated string -- except where kseq returns before it gets there: the line's first
		 * character was the last byte of the input and its 16 KiB stream buffer already knew (kseq.h:97; it knows unless the input's
		 * length is a multiple of the buffer size) */
		if (ps->l_seq > 1 && ps->seq[ps->l_seq - 1] == '\r' && !(r->l_line == 1 && !r->line_nl && r->total % 16384 != 0)) --ps->l_seq;
	}
	ps->rec_has_qual = 1;
	if (!r->line_nl) return -2; /* kseq.h:218-219: the input ends inside the '+' line */
	do { /* quality lines (kseq.h:220): at least one, then until as long as the sequence */
		if (!rd_line(r)) break;
		app(&ps->qual, &ps->l_qual, &ps->m_qual, r->line, r->l_line);
		if (ps->l_qual > 1 && ps->qual[ps->l_qual - 1] == '\r') --ps->l_qual;
	} while (ps->l_qual < ps->l_seq);
	if (ps->l_qual != ps->l_seq) return -2;
	return 1;
}

/* ------------------------------------------------------------------ fast path: uncompressed 4-line FASTQ in memory, several threads
 *
 * A plain (not gzip) regular file is mapped; a batch is cut out of a byte window that starts on a record boundary.  Every thread takes
 * a slice of the window, finds the first line in it that begins a STRICT record -- '@' line, one non-empty sequence line, '+' line, one
 * quality line of the same length: the shape on which kseq's grammar (above) and a line walk cannot differ -- and walks strict records
 * from there.  The walks are then chained: the first thread starts on the window start (a true boundary by induction) and every thread
 * must have started exactly where its predecessor stopped, so the window is ONE strict walk from a true boundary, i.e. what kseq_read
 * returns.  Anything else (multi-line records, FASTA, an empty sequence, a length mismatch, junk between records, a guessed start that
 * does not chain) sends this batch and the rest of the file through the serial parser above, from the window start.  The threads then
 * copy their records into the batch at offsets known from the per-thread totals. */
#include <pthread.h>
#include "bfc_pgz.h"

typedef struct { uint64_t hdr; uint32_t len, seq_delta; } fq_rec_t; /* '@' position in the file; bases (CR stripped); sequence line - '@' */

enum { FQ_OK = 0, FQ_CUT = 1, FQ_BAD = 2 };
#define FQ_MAX_THREADS 256

static inline const uint8_t *fq_eol(const uint8_t *p, const uint8_t *e) { const uint8_t *q = (const uint8_t*)memchr(p, '\n', (size_t)(e - p)); return q ? q : e; }

/* one strict record at s inside [.., e); at_eof: e is the end of the file, so the last line may lack its '\n'.
 * FQ_OK: *nx = position after the record; FQ_CUT: e comes before the record is complete; FQ_BAD: not a strict record */
static inline int fq_strict(const uint8_t *s, const uint8_t *e, int at_eof, fq_rec_t *rec, const uint8_t *base, const uint8_t **nx)
{
	const uint8_t *h, *q, *r, *t;
	size_t ls, lq;
	if (s >= e) return FQ_CUT;
	if (*s != '@') return FQ_BAD;
	h = fq_eol(s, e); if (h >= e) return at_eof ? FQ_BAD : FQ_CUT;
	q = fq_eol(h + 1, e); if (q >= e) return at_eof ? FQ_BAD : FQ_CUT;
	ls = (size_t)(q - (h + 1));
	if (ls > 1 && q[-1] == '\r') --ls;
	if (ls == 0 || h[1] == '>' || h[1] == '@' || h[1] == '+' || ls > 0x7fffffffu) return FQ_BAD;
	if (q + 1 >= e) return at_eof ? FQ_BAD : FQ_CUT;
	if (q[1] != '+') return FQ_BAD;
	r = fq_eol(q + 1, e); if (r >= e) return at_eof ? FQ_BAD : FQ_CUT;
	t = fq_eol(r + 1, e);
	if (t >= e && !at_eof) return FQ_CUT;
	lq = (size_t)(t - (r + 1));
	if (lq > 1 && t[-1] == '\r') --lq;
	if (lq != ls) return FQ_BAD;
	if (rec) { rec->hdr = (uint64_t)(s - base); rec->len = (uint32_t)ls; rec->seq_delta = (uint32_t)(h + 1 - s); }
	*nx = t < e ? t + 1 : e;
	return FQ_OK;
}

typedef struct {
	const uint8_t *base; uint64_t lo, hi, win_end; int first, at_eof; /* slice [lo,hi) of the window that ends at win_end */
	fq_rec_t *rec; uint64_t n_rec, m_rec, bases; /* strict records whose '@' lies in the slice */
	uint64_t start, end; int found, status;      /* where the walk began / stopped, and why (FQ_OK: ran into the slice end) */
	uint8_t *oseq, *oqual; uint64_t n_copy;      /* copy phase */
	uint32_t *pl; uint64_t pw, opos; int q;      /* ... or straight into bit planes: the job's records begin at stream position opos */
} fq_job_t;

static void *fq_scan(void *arg)
{
	fq_job_t *j = (fq_job_t*)arg;
	const uint8_t *base = j->base, *e = base + j->win_end, *s = base + j->lo, *lim = base + j->hi, *nx;
	j->n_rec = 0; j->bases = 0; j->found = 0; j->status = FQ_OK;
	if (!j->first) { /* first line start in the slice that begins a strict (or window-cut) record */
		if (s[-1] != '\n') { s = fq_eol(s, e); if (s < e) ++s; }
		while (s < lim && !(*s == '@' && fq_strict(s, e, j->at_eof, 0, base, &nx) != FQ_BAD)) { s = fq_eol(s, e); if (s < e) ++s; }
		if (s >= lim) return 0; /* no record starts in this slice */
	}
	j->found = 1; j->start = (uint64_t)(s - base);
	while (s < lim) {
		fq_rec_t rc;
		int st = fq_strict(s, e, j->at_eof, &rc, base, &nx);
		if (st != FQ_OK) {
			if (st == FQ_BAD && j->at_eof) { /* only line ends left before the end of the file? (kseq skips them looking for a header) */
				const uint8_t *t = s;
				while (t < e && (*t == '\n' || *t == '\r')) ++t;
				if (t == e) { s = e; st = FQ_CUT; }
			}
			j->status = st; break;
		}
		if (j->n_rec == j->m_rec) { j->m_rec = j->m_rec ? j->m_rec * 2 : 1 << 16; j->rec = (fq_rec_t*)realloc(j->rec, j->m_rec * sizeof(fq_rec_t)); }
		j->rec[j->n_rec++] = rc; j->bases += rc.len;
		s = nx;
	}
	j->end = (uint64_t)(s - base);
	return 0;
}

static void *fq_copy(void *arg)
{
	fq_job_t *j = (fq_job_t*)arg;
	uint8_t *os = j->oseq, *oq = j->oqual;
	uint64_t i;
	for (i = 0; i < j->n_copy; ++i) {
		const uint8_t *sq = j->base + j->rec[i].hdr + j->rec[i].seq_delta;
		const uint32_t l = j->rec[i].len;
		const uint8_t *pl = sq + l + (sq[l] == '\r' ? 2 : 1);                               /* the '+' line */
		const uint8_t *ql = fq_eol(pl, j->base + j->win_end) + 1;                            /* fq_strict saw its '\n' */
		memcpy(os, sq, l); os[l] = '\n'; os += l + 1;
		memcpy(oq, ql, l); oq[l] = '!'; oq += l + 1;
	}
	return 0;
}

/* The same records as bit planes (bfc_planes.h), without the byte streams in between: a job appends its records' positions -- bases, then one
 * separator per record -- at bit opos of the batch's four planes.  Words that a job shares with its neighbours (its first and its last) are
 * OR-ed in atomically (the caller cleared them), the others are stored. */
typedef struct { uint32_t *pl; uint64_t pw, w, w_first, w_last; int off; uint64_t a[4]; } bitw_t;
static inline void bw_flush(bitw_t *s)
{
	int p;
	if (s->w == s->w_first || s->w == s->w_last) { for (p = 0; p < 4; ++p) __atomic_fetch_or(&s->pl[(uint64_t)p * s->pw + s->w], (uint32_t)s->a[p], __ATOMIC_RELAXED); }
	else for (p = 0; p < 4; ++p) s->pl[(uint64_t)p * s->pw + s->w] = (uint32_t)s->a[p];
	for (p = 0; p < 4; ++p) s->a[p] >>= 32;
	++s->w; s->off -= 32;
}
static inline void bw_put(bitw_t *s, const uint32_t m[4], int n)
{
	s->a[0] |= (uint64_t)m[0] << s->off; s->a[1] |= (uint64_t)m[1] << s->off; s->a[2] |= (uint64_t)m[2] << s->off; s->a[3] |= (uint64_t)m[3] << s->off;
	s->off += n;
	if (s->off >= 32) bw_flush(s);
}
#if BFC_PLANES_HAVE_AVX2
/* fq_pack's record loop, 32 positions a step (bfc_planes32_avx2); a read's tail is one masked step where 32 bytes are readable behind it */
__attribute__((target("avx2"))) static void fq_pack_records_avx2(fq_job_t *j, bitw_t *s, bfc_qthr_t t, const uint32_t sep[4])
{
	const uint8_t *const wend = j->base + j->win_end;
	uint64_t i;
	for (i = 0; i < j->n_copy; ++i) {
		const uint8_t *sq = j->base + j->rec[i].hdr + j->rec[i].seq_delta;
		const uint32_t l = j->rec[i].len;
		const uint8_t *pl = sq + l + (sq[l] == '\r' ? 2 : 1);
		const uint8_t *ql = fq_eol(pl, wend) + 1;
		uint32_t k = 0, m[4];
		for (; k + 32 <= l; k += 32) { bfc_planes32_avx2(sq + k, ql + k, t, 32, m); bw_put(s, m, 32); }
		if (k < l) {
			if (ql + k + 32 <= wend) { bfc_planes32_avx2(sq + k, ql + k, t, (int)(l - k), m); bw_put(s, m, (int)(l - k)); }
			else {
				for (; k + 8 <= l; k += 8) { bfc_planes8(sq + k, ql + k, t, m); bw_put(s, m, 8); }
				for (; k < l; ++k) { bfc_planes1(sq[k], ql + k, t, m); bw_put(s, m, 1); }
			}
		}
		bw_put(s, sep, 1);
	}
}
#endif
static void *fq_pack(void *arg)
{
	fq_job_t *j = (fq_job_t*)arg;
	const bfc_qthr_t t = bfc_qthr(j->q);
	const uint32_t sep[4] = { 0, 0, 1, (uint32_t)((int)'!' >= t.T) }; /* (the byte streams carry '\n' and '!' there: the same bits) */
	bitw_t s;
	uint64_t i, n_pos = 0;
	if (j->n_copy == 0) return 0;
	for (i = 0; i < j->n_copy; ++i) n_pos += j->rec[i].len + 1;
	s.pl = j->pl; s.pw = j->pw; s.w = j->opos >> 5; s.w_first = s.w; s.w_last = (j->opos + n_pos - 1) >> 5; s.off = (int)(j->opos & 31);
	s.a[0] = s.a[1] = s.a[2] = s.a[3] = 0;
#if BFC_PLANES_HAVE_AVX2
	if (bfc_planes_avx2_ok()) { fq_pack_records_avx2(j, &s, t, sep); i = j->n_copy; }
	else
#endif
	for (i = 0; i < j->n_copy; ++i) {
		const uint8_t *sq = j->base + j->rec[i].hdr + j->rec[i].seq_delta;
		const uint32_t l = j->rec[i].len;
		const uint8_t *pl = sq + l + (sq[l] == '\r' ? 2 : 1);
		const uint8_t *ql = fq_eol(pl, j->base + j->win_end) + 1;
		uint32_t k = 0, m[4];
		for (; k + 8 <= l; k += 8) { bfc_planes8(sq + k, ql + k, t, m); bw_put(&s, m, 8); }
		for (; k < l; ++k) { bfc_planes1(sq[k], ql + k, t, m); bw_put(&s, m, 1); }
		bw_put(&s, sep, 1);
	}
	if (s.off > 0) { s.off += 32; bw_flush(&s); } /* (the last, partial word) */
	return 0;
}

/* ------------------------------------------------------------------ pipe input (round 6): stdin, a FIFO, `<(seqtk mergepe ...)` -- what the reference's
 * published command line feeds it (tex/README.md:26; bseq.c:33-50 opens whatever it is given with gzdopen / gzopen).  A pipe cannot be mapped, but the
 * chained walks above need memory, not a mapping: a reader thread drains the pipe into a RING whose cap bytes are mapped twice back to back (one
 * memfd, two mappings), so any window of up to cap bytes is contiguous in the address space wherever it starts, nothing is ever moved, and the
 * reader's read() lands directly behind the last byte.  The parser's threads walk the window while the reader fills the ring behind it.  The
 * serial parser takes its bytes from the same ring (reader_t.mem) when the input is not strict 4-line FASTQ: a pipe cannot be re-read. */
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <fcntl.h>
#include <unistd.h>
#include <errno.h>
typedef struct psrc_s {
	int fd, own_fd; uint8_t *ring; uint64_t cap;
	uint64_t head, tail;   /* stream offsets: [head, tail) is in the ring, byte o at ring[o % cap] (and again cap further on) */
	int eof, quit, on;
	pthread_t th; pthread_mutex_t mu; pthread_cond_t cv_data, cv_space;
} psrc_t;

static void *psrc_reader(void *arg)
{
	psrc_t *p = (psrc_t*)arg;
	int old_state;
	pthread_setcancelstate(PTHREAD_CANCEL_DISABLE, &old_state); /* (psrc_close cancels a reader blocked in read() -- and only there: never with the mutex held) */
	for (;;) {
		uint64_t t, space;
		ssize_t n;
		pthread_mutex_lock(&p->mu);
		while (p->tail - p->head == p->cap && !p->quit) pthread_cond_wait(&p->cv_space, &p->mu);
		if (p->quit) { pthread_mutex_unlock(&p->mu); return 0; }
		t = p->tail; space = p->cap - (p->tail - p->head);
		pthread_mutex_unlock(&p->mu);
		if (space > ((uint64_t)8 << 20)) space = (uint64_t)8 << 20;
		pthread_setcancelstate(PTHREAD_CANCEL_ENABLE, &old_state);
		do n = read(p->fd, p->ring + t % p->cap, (size_t)space); while (n < 0 && errno == EINTR);
		pthread_setcancelstate(PTHREAD_CANCEL_DISABLE, &old_state);
		pthread_mutex_lock(&p->mu);
		if (n <= 0) { p->eof = 1; pthread_cond_broadcast(&p->cv_data); pthread_mutex_unlock(&p->mu); return 0; } /* (a read error ends the input, as gzread's does) */
		p->tail += (uint64_t)n;
		pthread_cond_broadcast(&p->cv_data);
		pthread_mutex_unlock(&p->mu);
	}
}
static inline psrc_t *psrc_open(int fd, int own_fd, uint64_t cap)
{
	psrc_t *p = (psrc_t*)calloc(1, sizeof(psrc_t));
	int mfd;
	void *base;
	cap = (cap + ((2u << 20) - 1)) & ~(uint64_t)((2u << 20) - 1);
	mfd = (int)syscall(SYS_memfd_create, "bfc_pipe_ring", 0u);
	if (mfd < 0 || ftruncate(mfd, (off_t)cap) != 0) { if (mfd >= 0) close(mfd); free(p); return 0; }
	base = mmap(0, (size_t)(2 * cap), PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
	if (base == MAP_FAILED) { close(mfd); free(p); return 0; }
	if (mmap(base, (size_t)cap, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED, mfd, 0) == MAP_FAILED ||
	    mmap((uint8_t*)base + cap, (size_t)cap, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED, mfd, 0) == MAP_FAILED) { munmap(base, (size_t)(2 * cap)); close(mfd); free(p); return 0; }
	close(mfd);
	p->fd = fd; p->own_fd = own_fd; p->ring = (uint8_t*)base; p->cap = cap;
#ifdef F_SETPIPE_SZ
	(void)fcntl(fd, F_SETPIPE_SZ, 1 << 20); /* fewer, larger reads (the default pipe holds 64 KiB); refused above /proc/sys/fs/pipe-max-size: then as it was */
#endif
	pthread_mutex_init(&p->mu, 0); pthread_cond_init(&p->cv_data, 0); pthread_cond_init(&p->cv_space, 0);
	if (pthread_create(&p->th, 0, psrc_reader, p) != 0) { munmap(base, (size_t)(2 * cap)); free(p); return 0; }
	p->on = 1;
	return p;
}
static inline void psrc_close(psrc_t *p)
{
	if (!p) return;
	if (p->on) {
		pthread_mutex_lock(&p->mu); p->quit = 1; pthread_cond_broadcast(&p->cv_space); pthread_mutex_unlock(&p->mu);
		if (!p->eof) pthread_cancel(p->th); /* (blocked in read() on a pipe nobody writes to any more) */
		pthread_join(p->th, 0);
	}
	pthread_mutex_destroy(&p->mu); pthread_cond_destroy(&p->cv_data); pthread_cond_destroy(&p->cv_space);
	munmap(p->ring, (size_t)(2 * p->cap));
	if (p->own_fd) close(p->fd);
	free(p);
}
/* pgz_ensure's contract on the ring: bytes [pos, pos + want) made available (fewer at the end of the input), everything before pos given back to the
 * reader.  0 and a pointer q with q[off] = byte `off` of the stream for pos <= off < *avail_end; -1: the window does not fit the ring. */
static inline int psrc_ensure(psrc_t *p, uint64_t pos, uint64_t want, const uint8_t **q, uint64_t *avail_end, int *eof)
{
	if (want > p->cap) return -1;
	pthread_mutex_lock(&p->mu);
	if (pos < p->head) { pthread_mutex_unlock(&p->mu); return -1; }
	while (!p->eof && p->tail < pos) { p->head = p->tail; pthread_cond_signal(&p->cv_space); pthread_cond_wait(&p->cv_data, &p->mu); } /* (never: callers consume in order) */
	if (pos > p->head) { p->head = pos < p->tail ? pos : p->tail; pthread_cond_signal(&p->cv_space); }
	while (!p->eof && p->tail < pos + want) pthread_cond_wait(&p->cv_data, &p->mu);
	*avail_end = p->tail; *eof = p->eof;
	pthread_mutex_unlock(&p->mu);
	*q = p->ring + pos % p->cap - pos;
	return 0;
}
static inline int psrc_take(struct psrc_s *p, uint64_t pos, uint8_t *dst, int want)
{
	const uint8_t *q; uint64_t avail; int eof, n;
	if (psrc_ensure(p, pos, (uint64_t)want, &q, &avail, &eof) != 0) return 0;
	n = avail - pos < (uint64_t)want ? (int)(avail - pos) : want;
	if (n > 0) memcpy(dst, q + pos, (size_t)n);
	return n;
}
/* the first two bytes of a pipe WITHOUT consuming them (tee(2) duplicates what is there into a scratch pipe): -1 = not a pipe / cannot tell,
 * else how many were seen (0: the input is empty) */
static inline int pipe_peek2(int fd, uint8_t out[2])
{
	int sc[2], n = -1, tries;
	if (pipe(sc) != 0) return -1;
	for (tries = 0; tries < 200000; ++tries) {
		ssize_t r = tee(fd, sc[1], 2, 0); /* blocks until the writer has sent something or closed */
		if (r < 0) { if (errno == EINTR) continue; n = -1; break; }
		if (r > 0) { uint8_t tmp[2]; ssize_t g = read(sc[0], tmp, (size_t)r); if (g > 0) memcpy(out, tmp, (size_t)g); n = (int)g; }
		else n = 0;
		if (r == 0 || r >= 2) break;
		usleep(50); /* one byte so far */
	}
	close(sc[0]); close(sc[1]);
	return n;
}

typedef struct {
	const uint8_t *map; uint64_t size, pos; /* pos: next record boundary */
	psrc_t *pipe;          /* pipe input: `map` is re-based on the ring for every window, `size` unknown until the pipe's end */
	pgz_t *gz;             /* gzip input: `map` is the text the parallel inflate (bfc_pgz.h) holds from `pos` on, `size` unknown until its end */
	const uint8_t *zmap; uint64_t zsize; /* the mapped .gz file */
	int n_threads, active;
	/* The mapping is given back BEHIND the parser (round 5): a thread of its own unmaps what the batches have consumed, 2 MiB-aligned, while the
	 * parser's threads fault the next window in -- tearing down the page tables of c3's 15.6 GB in one munmap at the end cost 0.24-0.38 s of
	 * bfc_count's 1.6.  (Reading the windows into a reused buffer instead of mapping the file was measured too: no teardown at all, but the copy
	 * cost the parser more than that -- 2.0 against 1.8 s on the whole file.) */
	pthread_t um_th; int um_on, um_quit; pthread_mutex_t um_mu; pthread_cond_t um_cv; uint64_t um_to, um_done;
	bfc_pool_t *pool;      /* n_threads - 1 workers, kept for the whole input */
	uint64_t min_slice;    /* bytes a thread's slice has at least (65536; tests lower it to chain walks inside small files) */
	double bytes_per_base; /* of the batches so far: sizes the next window */
	fq_job_t *job;
} fq_fast_t;

static void *fq_unmapper(void *arg)
{
	fq_fast_t *f = (fq_fast_t*)arg;
	for (;;) {
		uint64_t to; int quit;
		pthread_mutex_lock(&f->um_mu);
		while (f->um_to == f->um_done && !f->um_quit) pthread_cond_wait(&f->um_cv, &f->um_mu);
		to = f->um_to; quit = f->um_quit;
		pthread_mutex_unlock(&f->um_mu);
		if (to > f->um_done) { munmap((void*)(f->map + f->um_done), (size_t)(to - f->um_done)); f->um_done = to; }
		else if (quit) return 0;
	}
}
/* everything before file offset `pos` has been parsed into batches: the mapping up to there may go */
static inline void fq_consumed(fq_fast_t *f, uint64_t pos)
{
	const uint64_t to = pos & ~(uint64_t)((2u << 20) - 1);
	if (!f->um_on || to <= f->um_to) return;
	pthread_mutex_lock(&f->um_mu);
	f->um_to = to;
	pthread_cond_signal(&f->um_cv);
	pthread_mutex_unlock(&f->um_mu);
}

static inline void fq_run(fq_fast_t *f, void *(*fn)(void*), int n)
{
	bfc_pool_run(f->pool, fn, f->job, sizeof(fq_job_t), n);
}

/* one batch out of the mapped file: 1 = done (b filled, f->pos advanced), 0 = not strict here: the caller falls back to the serial
 * parser from f->pos.  Batch boundary as in bseq_read (bseq.c:52-76): the read that brings the batch to >= chunk_size bases is its last. */
static inline int fq_fill_batch(fq_fast_t *f, batch_t *b, uint64_t chunk_size)
{
	uint64_t win = (uint64_t)((double)chunk_size * (f->bytes_per_base > 0 ? f->bytes_per_base * 1.02 : 3.0)) + (1u << 18);
	const uint64_t pos0 = f->pos;
	batch_clear(b);
	for (;;) {
		uint64_t wend = f->pos + win < f->size ? f->pos + win : f->size;
		int at_eof = wend == f->size;
		int T = f->n_threads, i, n_used = 0, cut = 0, bad = 0;
		uint64_t slice, cur = f->pos, bases = 0, npos = 0, nseq = 0;
		if (f->gz) { /* inflate until the window is there (or the input ends) */
			uint64_t avail; int eof;
			if (pgz_ensure(f->gz, f->pos, win, &f->map, &avail, &eof) != 0) return 0; /* damaged gzip: gzread decides what the reference would see */
			if (eof) f->size = avail;
			wend = f->pos + win < avail ? f->pos + win : avail; at_eof = eof && wend == avail;
			if (f->pos == 0 && wend > 0 && f->map[0] != '@') return 0;
		}
		if (f->pipe) { /* wait until the reader has the window in the ring (or the pipe's end) */
			uint64_t avail; int eof;
			if (psrc_ensure(f->pipe, f->pos, win, &f->map, &avail, &eof) != 0) return 0; /* a window larger than the ring: the serial parser goes on from here */
			if (eof) f->size = avail;
			wend = f->pos + win < avail ? f->pos + win : avail; at_eof = eof && wend == avail;
			if (f->pos == 0 && wend > 0 && f->map[0] != '@') return 0;
		}
		if (wend - f->pos < (uint64_t)T * f->min_slice) T = 1;
		slice = (wend - f->pos + T - 1) / T;
		for (i = 0; i < T; ++i) {
			fq_job_t *j = &f->job[i];
			j->base = f->map; j->lo = f->pos + i * slice; j->hi = j->lo + slice < wend ? j->lo + slice : wend; j->win_end = wend;
			j->first = i == 0; j->at_eof = at_eof;
			if (j->lo >= wend) { T = i; break; }
		}
		if (T == 0) return 1; /* nothing left: an empty batch */
		{ struct timespec t0_, t1_; static int tm_ = -1; if (tm_ < 0) tm_ = getenv("BFC_INGEST_TIMING") != 0; if (tm_) clock_gettime(CLOCK_MONOTONIC, &t0_);
		fq_run(f, fq_scan, T);
		if (tm_) { clock_gettime(CLOCK_MONOTONIC, &t1_); fprintf(stderr, "[T::fq] scan of %.1f MB by %d threads: %.1f ms\n", (double)(wend - f->pos) / 1e6, T, (t1_.tv_sec - t0_.tv_sec) * 1e3 + (t1_.tv_nsec - t0_.tv_nsec) / 1e6); } }
		/* chain the walks */
		for (i = 0; i < T && !cut; ++i) { /* records before the first thing that is not strict are still one chained walk */
			fq_job_t *j = &f->job[i];
			if (!j->found) {
				if (cur >= j->hi) continue; /* the previous walk's last record covers this slice */
				if (at_eof) { /* only line ends left before the end of the file? then the walk simply ended */
					const uint8_t *t = f->map + cur, *e = f->map + wend;
					while (t < e && (*t == '\n' || *t == '\r')) ++t;
					if (t == e) { cur = wend; cut = 1; break; }
				}
				bad = 1; break; /* a record should have started in this slice */
			}
			if (j->start != cur) { bad = 1; break; }
			cur = j->end; n_used = i + 1;
			if (j->status == FQ_CUT) cut = 1;
			if (j->status == FQ_BAD) { bad = 1; cut = 1; }
		}
		/* where does the batch end? */
		{
			int done = 0;
			for (i = 0; i < n_used && !done; ++i) {
				fq_job_t *j = &f->job[i];
				j->n_copy = 0;
				if (bases + j->bases < chunk_size && npos + j->bases + j->n_rec <= b->cap) { /* the whole slice goes in */
					j->n_copy = j->n_rec; bases += j->bases; npos += j->bases + j->n_rec; nseq += j->n_rec;
				} else {
					uint64_t k;
					for (k = 0; k < j->n_rec; ++k) {
						if (npos + j->rec[k].len + 1 > b->cap) { done = 1; break; }
						bases += j->rec[k].len; npos += j->rec[k].len + 1; ++nseq; ++j->n_copy;
						if (bases >= chunk_size) { done = 1; break; }
					}
				}
			}
			for (; i < n_used; ++i) f->job[i].n_copy = 0;
			if (!done && bad) return 0; /* the batch would have to run through something that is not strict 4-line FASTQ */
			if (!done && !at_eof && bases < chunk_size) { win *= 2; continue; } /* the window held less than one chunk: look further */
		}
		if (nseq == 0 && f->job[0].n_rec > 0) return 0; /* the first read does not fit the batch: the serial path reports it */
		if (nseq == 0 && cur < f->size && !at_eof) { /* not even one record inside the window */
			if (win < f->size) { win *= 2; continue; }
			return 0;
		}
		if (nseq > 0x7fffffffu) return 0;
		/* copy */
		{
			uint64_t o = 0, last_hdr = 0; const fq_job_t *lastj = 0;
			const int to_planes = b->planes != 0;
			if (!to_planes && !batch_need_streams(b)) return 0;
			for (i = 0; i < n_used; ++i) {
				fq_job_t *j = &f->job[i];
				uint64_t k, bsum = 0;
				if (to_planes) { j->pl = b->planes; j->pw = b->plane_words; j->opos = o; j->q = b->q; }
				else { j->oseq = b->seq + o; j->oqual = b->qual + o; }
				if (j->n_copy == j->n_rec) bsum = j->bases; else for (k = 0; k < j->n_copy; ++k) bsum += j->rec[k].len;
				if (to_planes && j->n_copy) { /* the words this job shares with its neighbours start out clear */
					int p;
					for (p = 0; p < 4; ++p) { b->planes[(uint64_t)p * b->plane_words + (o >> 5)] = 0; b->planes[(uint64_t)p * b->plane_words + ((o + bsum + j->n_copy - 1) >> 5)] = 0; }
				}
				o += bsum + j->n_copy;
				if (j->n_copy) { lastj = j; last_hdr = j->rec[j->n_copy - 1].hdr; }
			}
			if (n_used == 0) f->job[0].n_copy = 0;
			{ struct timespec t0_, t1_; const int tm_ = getenv("BFC_INGEST_TIMING") != 0; if (tm_) clock_gettime(CLOCK_MONOTONIC, &t0_);
			fq_run(f, to_planes ? fq_pack : fq_copy, n_used > 0 ? n_used : 1);
			if (tm_) { clock_gettime(CLOCK_MONOTONIC, &t1_); fprintf(stderr, "[T::fq] %s of %llu positions by %d threads: %.1f ms\n", to_planes ? "pack" : "copy", (unsigned long long)o, n_used, (t1_.tv_sec - t0_.tv_sec) * 1e3 + (t1_.tv_nsec - t0_.tv_nsec) / 1e6); } }
			if (to_planes) {
				if (o & 31) b->planes[2 * b->plane_words + (o >> 5)] |= ~0u << (o & 31); /* beyond the batch's end: separators */
				b->packed = 1;
			}
			b->n_pos = o; b->n_seqs = (int)nseq; b->has_qual = nseq > 0; b->last_kind = nseq > 0 ? 1 : -1;
			if (lastj) { /* the next batch starts after the last record taken */
				const uint8_t *nx = 0;
				int all = 1;
				for (i = 0; i < n_used; ++i) if (f->job[i].n_copy != f->job[i].n_rec) all = 0;
				if (all) f->pos = cur;
				else { fq_strict(f->map + last_hdr, f->map + wend, at_eof, 0, f->map, &nx); f->pos = (uint64_t)(nx - f->map); }
			} else f->pos = cur;
		}
		if (bases > 0) f->bytes_per_base = (double)(f->pos - pos0) / (double)bases;
		if (!f->gz && !f->pipe) fq_consumed(f, f->pos);
		return 1;
	}
}

/* fill one batch as bseq_read does (bseq.c:52-76): records until at least chunk_size bases, the end of the input or a malformed record
 * (kseq_read < 0) */
static inline void fill_batch(parser_t *ps, batch_t *b)
{
	uint64_t bases = 0;
	batch_clear(b);
	for (;;) {
		if (!ps->have_rec) {
			int rc = next_record(ps);
			if (rc <= 0) return; /* the caller counts empty batches (ingest_fill) */
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

/* ------------------------------------------------------------------ one input: fast path when possible, serial parser otherwise */
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>

typedef struct {
	parser_t ps;
	fq_fast_t fast;
	int fast_batches, serial_batches;
	int workers, empties; /* the reference's pipeline has 2 workers (1 with -J), each goes on until ITS bseq_read returns nothing
	                       * (kthread.c:88-106, count.c:97-101,143): the input ends with the workers-th empty batch, not the first */
} ingest_t;

/* 0 on success; the gz stream is always opened (it is the fallback and the only way for gzip / stdin / FASTA input) */
static inline int ingest_open(ingest_t *in, const char *fn, uint64_t chunk_size, int n_threads, int workers)
{
	memset(in, 0, sizeof(*in));
	in->ps.chunk_size = chunk_size;
	in->workers = workers < 1 ? 1 : workers;
	if (n_threads > 0 && !getenv("BFC_INGEST_NO_PIPE")) { /* a pipe (stdin, a FIFO, /dev/fd/N of a process substitution) with plain text in it: the ring */
		struct stat st;
		const int is_stdin = !(fn && strcmp(fn, "-"));
		int fd = -1;
		if (is_stdin) { if (fstat(fileno(stdin), &st) == 0 && S_ISFIFO(st.st_mode)) fd = fileno(stdin); }
		else if (stat(fn, &st) == 0 && S_ISFIFO(st.st_mode)) fd = open(fn, O_RDONLY); /* (waits for the pipe's writer, as gzopen would) */
		if (fd >= 0) {
			uint8_t two[2] = {0, 0};
			const int n = pipe_peek2(fd, two);
			if (n >= 1 && !(n >= 2 && two[0] == 0x1f && two[1] == 0x8b)) { /* not gzip (zlib would copy it through: gzread's transparent mode) */
				/* the ring holds the largest window a batch may need: the fast path starts with 3 bytes per base and doubles once at most before it gives up */
				uint64_t cap = getenv("BFC_INGEST_RING") ? strtoull(getenv("BFC_INGEST_RING"), 0, 10) : chunk_size * 6 + ((uint64_t)64 << 20);
				psrc_t *p = psrc_open(fd, !is_stdin, cap);
				if (p) {
					in->fast.pipe = p; in->fast.map = 0; in->fast.size = ~(uint64_t)0; in->fast.pos = 0; in->fast.active = 1;
					in->fast.n_threads = n_threads > FQ_MAX_THREADS ? FQ_MAX_THREADS : n_threads;
					in->fast.min_slice = getenv("BFC_INGEST_MIN_SLICE") ? strtoull(getenv("BFC_INGEST_MIN_SLICE"), 0, 10) : 65536;
					if (in->fast.min_slice < 16) in->fast.min_slice = 16;
					in->fast.job = (fq_job_t*)calloc((size_t)in->fast.n_threads, sizeof(fq_job_t));
					in->fast.pool = bfc_pool_create(in->fast.n_threads);
					in->ps.rd.buf = (uint8_t*)malloc(RD_BUF);
					in->ps.rd.mem = p; in->ps.rd.mem_pos = 0; /* (the serial parser, should it take over, reads the ring too: rd_fill) */
					return 0;
				}
			}
			/* gzip data, an empty pipe, or no ring: gzread on THIS descriptor (nothing was consumed; opening a FIFO a second time would wait for a second writer) */
			in->ps.rd.fp = gzdopen(fd, "r");
			if (in->ps.rd.fp == 0) return -1;
			in->ps.rd.buf = (uint8_t*)malloc(RD_BUF);
			return 0;
		}
	}
	in->ps.rd.fp = fn && strcmp(fn, "-") ? gzopen(fn, "r") : gzdopen(fileno(stdin), "r");
	if (in->ps.rd.fp == 0) return -1;
	in->ps.rd.buf = (uint8_t*)malloc(RD_BUF);
	if (n_threads > 0 && fn && strcmp(fn, "-")) { /* a regular, uncompressed file that starts like a FASTQ: map it */
		struct stat st;
		int fd = open(fn, O_RDONLY);
		if (fd >= 0 && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 4) {
			void *m = mmap(0, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
			if (m != MAP_FAILED) {
				const uint8_t *p = (const uint8_t*)m;
				if (p[0] == 0x1f && p[1] == 0x8b && (uint64_t)st.st_size >= (getenv("BFC_INGEST_GZ_MIN") ? strtoull(getenv("BFC_INGEST_GZ_MIN"), 0, 10) : (uint64_t)1 << 20)) {
					/* gzip: the same chained walks over text that several threads inflate (bfc_pgz.h) */
					(void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
					in->fast.zmap = p; in->fast.zsize = (uint64_t)st.st_size;
					in->fast.n_threads = n_threads > FQ_MAX_THREADS ? FQ_MAX_THREADS : n_threads;
					in->fast.gz = pgz_open(p, (size_t)st.st_size, in->fast.n_threads, getenv("BFC_INGEST_GZ_CHUNK") ? strtoull(getenv("BFC_INGEST_GZ_CHUNK"), 0, 10) : (size_t)2 << 20);
					in->fast.map = 0; in->fast.size = ~(uint64_t)0; in->fast.pos = 0; in->fast.active = 1;
					in->fast.min_slice = getenv("BFC_INGEST_MIN_SLICE") ? strtoull(getenv("BFC_INGEST_MIN_SLICE"), 0, 10) : 65536;
					if (in->fast.min_slice < 16) in->fast.min_slice = 16;
					in->fast.job = (fq_job_t*)calloc((size_t)in->fast.n_threads, sizeof(fq_job_t));
					in->fast.pool = bfc_pool_create(in->fast.n_threads); in->fast.gz->pool = in->fast.pool;
				} else if (p[0] == '@') {
					(void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
					in->fast.map = p; in->fast.size = (uint64_t)st.st_size; in->fast.pos = 0; in->fast.active = 1;
					in->fast.n_threads = n_threads > FQ_MAX_THREADS ? FQ_MAX_THREADS : n_threads;
					in->fast.min_slice = getenv("BFC_INGEST_MIN_SLICE") ? strtoull(getenv("BFC_INGEST_MIN_SLICE"), 0, 10) : 65536;
					if (in->fast.min_slice < 16) in->fast.min_slice = 16;
					in->fast.job = (fq_job_t*)calloc((size_t)in->fast.n_threads, sizeof(fq_job_t));
					in->fast.pool = bfc_pool_create(in->fast.n_threads);
					if ((uint64_t)st.st_size >= (getenv("BFC_INGEST_UNMAP_MIN") ? strtoull(getenv("BFC_INGEST_UNMAP_MIN"), 0, 10) : (uint64_t)256 << 20)) { /* (small files: one munmap at the end) */
						pthread_mutex_init(&in->fast.um_mu, 0); pthread_cond_init(&in->fast.um_cv, 0);
						if (pthread_create(&in->fast.um_th, 0, fq_unmapper, &in->fast) == 0) in->fast.um_on = 1;
					}
				} else munmap(m, (size_t)st.st_size);
			}
		}
		if (fd >= 0) close(fd);
	}
	return 0;
}

static inline void ingest_fill(ingest_t *in, batch_t *b)
{
	int done = 0;
	if (in->fast.active) {
		if (fq_fill_batch(&in->fast, b, in->ps.chunk_size)) { if (b->n_seqs) ++in->fast_batches; done = 1; }
		else {
			in->fast.active = 0; /* not strict 4-line FASTQ from here on: the serial parser takes over at the last record boundary */
			if (in->fast.gz) { pgz_close(in->fast.gz); in->fast.gz = 0; in->fast.map = 0; } /* (for gzip input zlib inflates up to there again) */
			if (in->fast.pipe) { in->ps.rd.mem_pos = in->fast.pos; in->ps.rd.total = in->fast.pos; } /* a pipe is not read again: the ring still holds everything from the boundary on */
			else {
			gzseek(in->ps.rd.fp, (z_off_t)in->fast.pos, SEEK_SET);
			in->ps.rd.total = in->fast.pos; in->ps.rd.first_piece = RD_PIECE - (int)(in->fast.pos % RD_PIECE);
			}
		}
	}
	if (!done) {
		if (!batch_need_streams(b)) { fprintf(stderr, "[E::bfc_count] cannot allocate %llu bytes of batch buffers\n", (unsigned long long)b->cap); abort(); }
		fill_batch(&in->ps, b); if (b->n_seqs) ++in->serial_batches;
	}
	b->last = 0;
	if (b->n_seqs == 0 && ++in->empties >= in->workers) b->last = 1;
}

static inline void ingest_close(ingest_t *in)
{
	int i;
	if (in->fast.gz) pgz_close(in->fast.gz);
	bfc_pool_destroy(in->fast.pool);
	if (in->fast.um_on) {
		pthread_mutex_lock(&in->fast.um_mu); in->fast.um_quit = 1; pthread_cond_signal(&in->fast.um_cv); pthread_mutex_unlock(&in->fast.um_mu);
		pthread_join(in->fast.um_th, 0);
		pthread_mutex_destroy(&in->fast.um_mu); pthread_cond_destroy(&in->fast.um_cv);
	}
	if (in->fast.zmap) munmap((void*)in->fast.zmap, (size_t)in->fast.zsize);
	else if (in->fast.pipe) ;
	else if (in->fast.map && in->fast.size > in->fast.um_done) munmap((void*)(in->fast.map + in->fast.um_done), (size_t)(in->fast.size - in->fast.um_done));
	if (in->fast.job) { for (i = 0; i < in->fast.n_threads; ++i) free(in->fast.job[i].rec); free(in->fast.job); }
	if (in->fast.pipe) psrc_close(in->fast.pipe);
	if (in->ps.rd.fp) gzclose(in->ps.rd.fp);
	free(in->ps.rd.buf); free(in->ps.rd.line); free(in->ps.seq); free(in->ps.qual); free(in->ps.hdr); free(in->ps.cmt);
}

#endif
