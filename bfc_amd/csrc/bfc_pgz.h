/* bfc_pgz.h -- gzip input for the ingest fast path: ONE deflate stream inflated by several threads.
 *
 * The reference reads its input through zlib's gzread (bseq.c:33-50, kseq.h:185-224 over gzFile): one inflate stream, ~0.3 GB/s of
 * text, far below what the GPU path consumes.  A deflate stream has no index, but its blocks are self-delimiting once a block start
 * is known, and everything a block needs from before it is the last 32 KiB of text.  So, per round:
 *   1. the compressed bytes ahead are cut into T chunks.  Thread 0 starts at the exact position the previous round stopped at;
 *      every other thread SEARCHES its chunk for the first bit offset at which a dynamic-Huffman block header parses (complete
 *      code-length code, complete literal/length and distance codes, end-of-block symbol present), or the first gzip member
 *      header, decodes from there and stops at the first block boundary at or behind the next chunk's start;
 *   2. a thread that started in the middle of the stream does not know the 32 KiB before it: it decodes into 16-bit symbols and
 *      writes a back reference into the unknown window as a MARKER (0x8000 | window offset); markers are copied like literals;
 *   3. the pieces are chained in stream order: a piece is taken only if it STARTED exactly where its predecessor STOPPED (bit
 *      position and kind of boundary), so the result is one valid decode from the first byte -- never a guess; its markers are
 *      resolved with the window its predecessor left.  A piece that does not chain (a guessed start that was not a block start, a
 *      stored / fixed-Huffman block at the boundary, a block longer than a chunk) is decoded again from the known position;
 *   4. all threads narrow their pieces into the text buffer and take the CRC-32 of them; CRC-32 and ISIZE of every gzip member are
 *      checked from the pieces' sums (crc32_combine).
 * Any error (damaged or truncated stream, CRC mismatch, unknown header flags) makes pgz_ensure() fail: the caller then falls back
 * to gzread from its last record boundary, i.e. to exactly what the reference does with such a file.
 * (Technique: pugz / rapidgzip; written from the format, RFC 1951 / RFC 1952.) */
#ifndef BFC_PGZ_H
#define BFC_PGZ_H
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <zlib.h> /* crc32, crc32_combine */

#if defined(__x86_64__)
#include <immintrin.h>
#define PGZ_X86 1
#else
#define PGZ_X86 0
#endif

#include <sys/mman.h>

/* the large buffers (symbols, text) on 2 MiB pages where the kernel gives them: a fresh piece buffer is otherwise a page fault per 4 KiB,
 * taken by all threads at once */
static inline void *pgz_big_realloc(void *old, size_t old_bytes, size_t new_bytes)
{
	void *p = 0;
	const size_t al = (size_t)2 << 20;
	if (new_bytes < al) return realloc(old, new_bytes);
	if (posix_memalign(&p, al, (new_bytes + al - 1) & ~(al - 1)) != 0) return 0;
#ifdef MADV_HUGEPAGE
	if (!getenv("BFC_GPU_NO_THP")) (void)madvise(p, (new_bytes + al - 1) & ~(al - 1), MADV_HUGEPAGE);
#endif
	if (old) { memcpy(p, old, old_bytes < new_bytes ? old_bytes : new_bytes); free(old); }
	return p;
}

/* A small pool of worker threads for the ingest (the chained FASTQ walks, their copy phase, the inflate's two phases): a batch of 100 M bases
 * is ~20 ms of work for 64 threads, and creating + joining 63 threads twice per batch was a third of that.  run(pool, fn, jobs, stride, n)
 * runs fn(jobs + i * stride) for i < n, job 0 on the caller; one run at a time. */
typedef struct bfc_pool_s {
	pthread_t *th; int n_th, n_alloc;
	pthread_mutex_t mu; pthread_cond_t cv_go, cv_done;
	uint64_t gen; int quit, pending;
	void *(*fn)(void*); char *jobs; size_t stride; int n_jobs;
	struct bfc_pool_arg_s { struct bfc_pool_s *p; int idx; } *arg;
} bfc_pool_t;

static void *bfc_pool_worker(void *a)
{
	struct bfc_pool_arg_s *arg = (struct bfc_pool_arg_s*)a;
	bfc_pool_t *p = arg->p;
	const int idx = arg->idx;
	uint64_t seen = 0;
	for (;;) {
		void *(*fn)(void*); char *job;
		pthread_mutex_lock(&p->mu);
		while (p->gen == seen && !p->quit) pthread_cond_wait(&p->cv_go, &p->mu);
		if (p->quit) { pthread_mutex_unlock(&p->mu); return 0; }
		seen = p->gen;
		if (idx >= p->n_jobs) { pthread_mutex_unlock(&p->mu); continue; }
		fn = p->fn; job = p->jobs + (size_t)idx * p->stride;
		pthread_mutex_unlock(&p->mu);
		fn(job);
		pthread_mutex_lock(&p->mu);
		if (--p->pending == 0) pthread_cond_signal(&p->cv_done);
		pthread_mutex_unlock(&p->mu);
	}
}

static inline bfc_pool_t *bfc_pool_create(int n_threads) /* n_threads - 1 workers; NULL for one thread or on failure (callers then run serially created threads) */
{
	bfc_pool_t *p;
	int i;
	if (n_threads < 2) return 0;
	p = (bfc_pool_t*)calloc(1, sizeof(bfc_pool_t));
	p->n_alloc = n_threads - 1;
	p->th = (pthread_t*)calloc((size_t)p->n_alloc, sizeof(pthread_t));
	p->arg = (struct bfc_pool_arg_s*)calloc((size_t)p->n_alloc, sizeof(*p->arg));
	pthread_mutex_init(&p->mu, 0); pthread_cond_init(&p->cv_go, 0); pthread_cond_init(&p->cv_done, 0);
	for (i = 0; i < p->n_alloc; ++i) {
		p->arg[i].p = p; p->arg[i].idx = i + 1;
		if (pthread_create(&p->th[i], 0, bfc_pool_worker, &p->arg[i]) != 0) break;
		++p->n_th;
	}
	return p;
}

static inline void bfc_pool_destroy(bfc_pool_t *p)
{
	int i;
	if (p == 0) return;
	pthread_mutex_lock(&p->mu); p->quit = 1; pthread_cond_broadcast(&p->cv_go); pthread_mutex_unlock(&p->mu);
	for (i = 0; i < p->n_th; ++i) pthread_join(p->th[i], 0);
	pthread_mutex_destroy(&p->mu); pthread_cond_destroy(&p->cv_go); pthread_cond_destroy(&p->cv_done);
	free(p->th); free(p->arg); free(p);
}

static inline void bfc_pool_run(bfc_pool_t *p, void *(*fn)(void*), void *jobs, size_t stride, int n)
{
	int i, n_par = n;
	if (p == 0 || n < 2) { for (i = 0; i < n; ++i) fn((char*)jobs + (size_t)i * stride); return; }
	if (n_par > p->n_th + 1) n_par = p->n_th + 1; /* (fewer workers than jobs: the caller takes the rest) */
	pthread_mutex_lock(&p->mu);
	p->fn = fn; p->jobs = (char*)jobs; p->stride = stride; p->n_jobs = n_par; p->pending = n_par - 1; ++p->gen;
	pthread_cond_broadcast(&p->cv_go);
	pthread_mutex_unlock(&p->mu);
	fn(jobs);
	for (i = n_par; i < n; ++i) fn((char*)jobs + (size_t)i * stride);
	pthread_mutex_lock(&p->mu);
	while (p->pending) pthread_cond_wait(&p->cv_done, &p->mu);
	pthread_mutex_unlock(&p->mu);
}

#define PGZ_WIN 32768
#define PGZ_LB 11 /* bits of the literal/length root table */
#define PGZ_DB 9  /* bits of the distance root table */
#define PGZ_LT_CAP (2048 + 288 * 16)
#define PGZ_DT_CAP (512 + 32 * 64)
#define PGZ_MAX_THREADS 64

/* table entry: val | bits << 16 | op << 24.  op: 0 literal, 1 end of block, 2 length / distance (val = base, xb = extra bits),
 * 3 link to a sub-table (val = its offset, xb = its index bits), 4 invalid code */
typedef struct { uint16_t val; uint8_t bits; uint8_t op : 3, xb : 5; } pgz_ent_t;
enum { PGZ_LIT = 0, PGZ_EOB = 1, PGZ_BASE = 2, PGZ_LINK = 3, PGZ_BAD = 4 };

enum { PGZ_AT_BLOCK = 0, PGZ_AT_MEMBER = 1 }; /* what begins at a chain position: a deflate block inside a member / a gzip member (or the end) */

typedef struct { const uint8_t *p, *end, *base; uint64_t bb; int bc; int over; } pgz_br_t;

static inline void pgz_br_init(pgz_br_t *b, const uint8_t *base, size_t len, uint64_t bitpos)
{
	b->base = base; b->end = base + len; b->p = base + (bitpos >> 3); b->bb = 0; b->bc = 0; b->over = 0;
	if (b->p > b->end) b->p = b->end;
	if (bitpos & 7) { if (b->p < b->end) { b->bb = (uint64_t)(*b->p++) >> (bitpos & 7); b->bc = 8 - (int)(bitpos & 7); } else b->over = 1; }
}
static inline uint64_t pgz_br_pos(const pgz_br_t *b) { return (uint64_t)(b->p - b->base) * 8 - (uint64_t)b->bc; }
static inline void pgz_br_refill(pgz_br_t *b)
{
	if (b->p + 8 <= b->end) {
		uint64_t w; memcpy(&w, b->p, 8);
		b->bb |= w << b->bc; b->p += (63 - b->bc) >> 3; b->bc |= 56;
	} else {
		b->bb &= b->bc >= 64 ? ~0ULL : (1ULL << b->bc) - 1; /* the fast path leaves look-ahead bits above bc */
		while (b->bc <= 56 && b->p < b->end) { b->bb |= (uint64_t)(*b->p++) << b->bc; b->bc += 8; }
	}
}
/* n <= 32 bits; past the end of the input the bits read as zero and `over` is set */
static inline uint32_t pgz_br_get(pgz_br_t *b, int n)
{
	uint32_t v;
	if (b->bc < n) { pgz_br_refill(b); if (b->bc < n) { b->over = 1; b->bb &= b->bc > 0 ? (1ULL << b->bc) - 1 : 0; v = (uint32_t)(b->bb & ((1ULL << n) - 1)); b->bb = 0; b->bc = 0; return v; } }
	v = (uint32_t)(b->bb & ((1ULL << n) - 1)); b->bb >>= n; b->bc -= n;
	return v;
}

static const uint16_t pgz_len_base[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
static const uint8_t pgz_len_xb[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
static const uint16_t pgz_dist_base[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
static const uint8_t pgz_dist_xb[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };

enum { PGZ_K_CODES = 0, PGZ_K_LENS = 1, PGZ_K_DISTS = 2 };

static inline pgz_ent_t pgz_mk(int kind, int sym, int bits)
{
	pgz_ent_t e; e.bits = (uint8_t)bits; e.val = (uint16_t)sym; e.op = PGZ_LIT; e.xb = 0;
	if (kind == PGZ_K_LENS) {
		if (sym == 256) e.op = PGZ_EOB;
		else if (sym > 256) { if (sym > 285) e.op = PGZ_BAD; else { e.op = PGZ_BASE; e.val = pgz_len_base[sym - 257]; e.xb = pgz_len_xb[sym - 257]; } }
	} else if (kind == PGZ_K_DISTS) {
		if (sym > 29) e.op = PGZ_BAD; else { e.op = PGZ_BASE; e.val = pgz_dist_base[sym]; e.xb = pgz_dist_xb[sym]; }
	}
	return e;
}

/* canonical Huffman code of RFC 1951 3.2.2 -> root table of rb bits + sub-tables; the rules on which sets of lengths are valid are
 * zlib's (inftrees.c): over-subscribed is an error, incomplete is an error unless it is a single one-bit code (not for the
 * code-length code), no code at all is accepted (every index then decodes as invalid).  Returns 0, or -1 for an invalid set. */
static inline int pgz_build(int kind, const uint8_t *len, int n, int rb, pgz_ent_t *tab, int cap)
{
	int count[16], next[16], i, l, max = 0, left = 1, used = 1 << rb;
	uint8_t sub_bits[1 << PGZ_LB];
	pgz_ent_t bad; bad.val = 0; bad.bits = 1; bad.op = PGZ_BAD; bad.xb = 0;
	memset(count, 0, sizeof(count));
	for (i = 0; i < n; ++i) ++count[len[i]];
	for (l = 15; l >= 1; --l) if (count[l]) { max = l; break; }
	for (i = 0; i < (1 << rb); ++i) tab[i] = bad;
	if (max == 0) return 0;
	for (l = 1; l <= 15; ++l) { left <<= 1; left -= count[l]; if (left < 0) return -1; }
	if (left > 0 && (kind == PGZ_K_CODES || max != 1)) return -1;
	next[1] = 0;
	for (l = 1; l < 15; ++l) next[l + 1] = (next[l] + count[l]) << 1;
	if (max > rb) { /* first the sizes of the sub-tables: longest code under each root index */
		int nx[16];
		memset(sub_bits, 0, (size_t)1 << rb);
		memcpy(nx, next, sizeof(nx));
		for (i = 0; i < n; ++i) {
			int c, r = 0, k;
			if ((l = len[i]) == 0) continue;
			c = nx[l]++;
			if (l <= rb) continue;
			for (k = 0; k < l; ++k) r |= ((c >> k) & 1) << (l - 1 - k);
			r &= (1 << rb) - 1;
			if (l - rb > sub_bits[r]) sub_bits[r] = (uint8_t)(l - rb);
		}
		for (i = 0; i < (1 << rb); ++i) if (sub_bits[i]) {
			int k, sz = 1 << sub_bits[i];
			if (used + sz > cap) return -1;
			tab[i].op = PGZ_LINK; tab[i].val = (uint16_t)used; tab[i].bits = (uint8_t)rb; tab[i].xb = sub_bits[i];
			for (k = 0; k < sz; ++k) { tab[used + k] = bad; tab[used + k].bits = 1; }
			used += sz;
		}
	}
	for (i = 0; i < n; ++i) {
		int c, r = 0, k;
		if ((l = len[i]) == 0) continue;
		c = next[l]++;
		for (k = 0; k < l; ++k) r |= ((c >> k) & 1) << (l - 1 - k); /* deflate packs Huffman codes starting from their most significant bit */
		if (l <= rb) {
			pgz_ent_t e = pgz_mk(kind, i, l);
			for (k = r; k < (1 << rb); k += 1 << l) tab[k] = e;
		} else {
			const pgz_ent_t lk = tab[r & ((1 << rb) - 1)];
			pgz_ent_t e = pgz_mk(kind, i, l - rb);
			for (k = r >> rb; k < (1 << lk.xb); k += 1 << (l - rb)) tab[lk.val + k] = e;
		}
	}
	return 0;
}

typedef struct { pgz_ent_t lt[PGZ_LT_CAP], dt[PGZ_DT_CAP]; } pgz_tabs_t;

/* header of a dynamic block behind its 3 type bits (RFC 1951 3.2.7), checks as inflate.c makes them */
static inline int pgz_dyn_header(pgz_br_t *b, pgz_tabs_t *t)
{
	static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
	uint8_t cl[19], lens[320];
	pgz_ent_t ct[128];
	int nlen, ndist, ncode, i, n;
	nlen = (int)pgz_br_get(b, 5) + 257; ndist = (int)pgz_br_get(b, 5) + 1; ncode = (int)pgz_br_get(b, 4) + 4;
	if (nlen > 286 || ndist > 30) return -1;
	memset(cl, 0, sizeof(cl));
	for (i = 0; i < ncode; ++i) cl[order[i]] = (uint8_t)pgz_br_get(b, 3);
	if (b->over) return -1;
	if (pgz_build(PGZ_K_CODES, cl, 19, 7, ct, 128) != 0) return -1;
	for (n = 0; n < nlen + ndist;) {
		pgz_ent_t e;
		if (b->bc < 16) pgz_br_refill(b);
		e = ct[b->bb & 127];
		if (e.op == PGZ_BAD) return -1;
		if (b->bc < e.bits) return -1;
		b->bb >>= e.bits; b->bc -= e.bits;
		if (e.val < 16) lens[n++] = (uint8_t)e.val;
		else {
			int rep, v = 0;
			if (e.val == 16) { if (n == 0) return -1; v = lens[n - 1]; rep = 3 + (int)pgz_br_get(b, 2); }
			else if (e.val == 17) rep = 3 + (int)pgz_br_get(b, 3);
			else rep = 11 + (int)pgz_br_get(b, 7);
			if (n + rep > nlen + ndist) return -1;
			while (rep--) lens[n++] = (uint8_t)v;
		}
		if (b->over) return -1;
	}
	if (lens[256] == 0) return -1; /* no end-of-block code */
	if (pgz_build(PGZ_K_LENS, lens, nlen, PGZ_LB, t->lt, PGZ_LT_CAP) != 0) return -1;
	if (pgz_build(PGZ_K_DISTS, lens + nlen, ndist, PGZ_DB, t->dt, PGZ_DT_CAP) != 0) return -1;
	return 0;
}

static inline void pgz_fixed_tabs(pgz_tabs_t *t)
{
	uint8_t lens[288], d[32];
	int i;
	for (i = 0; i < 144; ++i) lens[i] = 8;
	for (; i < 256; ++i) lens[i] = 9;
	for (; i < 280; ++i) lens[i] = 7;
	for (; i < 288; ++i) lens[i] = 8;
	for (i = 0; i < 32; ++i) d[i] = 5;
	pgz_build(PGZ_K_LENS, lens, 288, PGZ_LB, t->lt, PGZ_LT_CAP);
	pgz_build(PGZ_K_DISTS, d, 32, PGZ_DB, t->dt, PGZ_DT_CAP);
}

/* a piece of decoded stream: 16-bit symbols behind a 32 KiB prefix that stands for the window before the piece (real bytes when the
 * window is known, markers 0x8000|i when it is not) */
typedef struct { uint64_t at; uint32_t crc, isize; } pgz_mend_t; /* a gzip member ended after `at` symbols of the piece */
typedef struct {
	uint16_t *sym; size_t n, cap, cap0; /* n counts the prefix; cap0: first allocation */
	size_t floor;                 /* symbols before this index do not exist for back references (a gzip member started there) */
	pgz_mend_t *mend; int n_mend, m_mend;
	uint64_t start, end; int start_kind, end_kind; /* chain positions in bits */
	int ok;                       /* decoded to `end` without an error */
	int at_eof;                   /* `end` is the end of the input (nothing but trailing garbage follows) */
} pgz_piece_t;

static inline int pgz_piece_room(pgz_piece_t *pc, size_t more)
{
	if (pc->n + more > pc->cap) {
		size_t nc = pc->cap ? pc->cap * 2 : pc->cap0 > ((size_t)1 << 20) ? pc->cap0 : (size_t)1 << 20;
		uint16_t *ns;
		while (nc < pc->n + more) nc *= 2;
		if (nc > ((size_t)1 << 31)) return -1; /* a piece of more than 2 G symbols: gzread's business */
		if ((ns = (uint16_t*)pgz_big_realloc(pc->sym, pc->n * sizeof(uint16_t), nc * sizeof(uint16_t))) == 0) return -1;
		pc->sym = ns; pc->cap = nc;
	}
	return 0;
}

/* the symbols of one Huffman-coded block; 0 at its end-of-block code, -1 on an invalid code / distance or the end of the input.
 * This version checks for the end of the input after every symbol; pgz_codes() below runs ahead of it while 16 bytes of input are left. */
static inline int pgz_codes_careful(pgz_br_t *b, const pgz_tabs_t *t, pgz_piece_t *pc)
{
	for (;;) {
		uint16_t *out;
		size_t n = pc->n, lim;
		if (pgz_piece_room(pc, 65536 + 258) != 0) return -1;
		out = pc->sym; lim = pc->cap - 258 - 1;
		while (n < lim) {
			pgz_ent_t e;
			pgz_br_refill(b); /* >= 56 bits unless the input ends: a whole length / distance pair needs at most 48 */
			e = t->lt[b->bb & ((1u << PGZ_LB) - 1)];
			if (e.op == PGZ_LINK) e = t->lt[e.val + ((b->bb >> PGZ_LB) & ((1u << e.xb) - 1))], b->bb >>= PGZ_LB, b->bc -= PGZ_LB;
			b->bb >>= e.bits; b->bc -= e.bits;
			if (e.op == PGZ_LIT) {
				out[n++] = e.val;
				/* a second literal from the same refill (most of a FASTQ's symbols are literals) */
				e = t->lt[b->bb & ((1u << PGZ_LB) - 1)];
				if (e.op == PGZ_LIT && b->bc >= 32) { b->bb >>= e.bits; b->bc -= e.bits; out[n++] = e.val; }
				if (b->bc < 0) { pc->n = n - 1; b->over = 1; return -1; }
				continue;
			}
			if (e.op == PGZ_BASE) {
				uint32_t len = e.val + (uint32_t)(b->bb & ((1u << e.xb) - 1)), dist;
				pgz_ent_t d;
				size_t src;
				b->bb >>= e.xb; b->bc -= e.xb;
				d = t->dt[b->bb & ((1u << PGZ_DB) - 1)];
				if (d.op == PGZ_LINK) d = t->dt[d.val + ((b->bb >> PGZ_DB) & ((1u << d.xb) - 1))], b->bb >>= PGZ_DB, b->bc -= PGZ_DB;
				b->bb >>= d.bits; b->bc -= d.bits;
				if (d.op != PGZ_BASE) { pc->n = n; return -1; }
				dist = d.val + (uint32_t)(b->bb & ((1u << d.xb) - 1));
				b->bb >>= d.xb; b->bc -= d.xb;
				if (b->bc < 0) { pc->n = n; b->over = 1; return -1; }
				if (dist > n - pc->floor) { pc->n = n; return -1; } /* before the start of the member, or further back than a window */
				src = n - dist;
				if (dist >= len) memcpy(out + n, out + src, (size_t)len * 2);
				else { uint32_t k; for (k = 0; k < len; ++k) out[n + k] = out[src + k]; }
				n += len;
				continue;
			}
			pc->n = n;
			if (b->bc < 0) { b->over = 1; return -1; }
			return e.op == PGZ_EOB ? 0 : -1;
		}
		pc->n = n;
	}
}

/* the same while at least 16 bytes of input lie ahead (a refill then always delivers 56 valid bits: no end-of-input checks), bit buffer and
 * cursors in registers, up to three literals per refill, matches copied 16 bytes at a time; the careful loop finishes the block */
static inline int pgz_codes(pgz_br_t *b, const pgz_tabs_t *t, pgz_piece_t *pc)
{
	const pgz_ent_t *lt = t->lt, *dt = t->dt;
	const uint32_t LM = (1u << PGZ_LB) - 1, DM = (1u << PGZ_DB) - 1;
	while (b->end - b->p >= 32) {
		uint64_t bb = b->bb; int bc = b->bc;
		const uint8_t *p = b->p, *fast_end = b->end - 16;
		uint16_t *out;
		size_t n = pc->n, lim;
		const size_t floor = pc->floor;
		int done = 0; /* 1 end of block, -1 error */
		if (pgz_piece_room(pc, 65536 + 300) != 0) return -1;
		out = pc->sym; lim = pc->cap - 258 - 16;
		while (n < lim && p <= fast_end) {
			pgz_ent_t e;
			uint64_t w;
			memcpy(&w, p, 8); bb |= w << bc; p += (63 - bc) >> 3; bc |= 56;
			e = lt[bb & LM];
			if (e.op == PGZ_LIT) {
				bb >>= e.bits; bc -= e.bits; out[n++] = e.val;
				e = lt[bb & LM];
				if (e.op == PGZ_LIT) {
					bb >>= e.bits; bc -= e.bits; out[n++] = e.val;
					e = lt[bb & LM];
					if (e.op == PGZ_LIT) { bb >>= e.bits; bc -= e.bits; out[n++] = e.val; }
				}
				continue;
			}
			if (e.op == PGZ_LINK) { e = lt[e.val + ((bb >> PGZ_LB) & ((1u << e.xb) - 1))]; bb >>= PGZ_LB; bc -= PGZ_LB; }
			bb >>= e.bits; bc -= e.bits;
			if (e.op == PGZ_LIT) { out[n++] = e.val; continue; }
			if (e.op == PGZ_BASE) {
				const uint32_t len = e.val + (uint32_t)(bb & ((1u << e.xb) - 1));
				uint32_t dist;
				pgz_ent_t d;
				bb >>= e.xb; bc -= e.xb;
				d = dt[bb & DM];
				if (d.op == PGZ_LINK) { d = dt[d.val + ((bb >> PGZ_DB) & ((1u << d.xb) - 1))]; bb >>= PGZ_DB; bc -= PGZ_DB; }
				bb >>= d.bits; bc -= d.bits;
				if (d.op != PGZ_BASE) { done = -1; break; }
				dist = d.val + (uint32_t)(bb & ((1u << d.xb) - 1));
				bb >>= d.xb; bc -= d.xb;
				if (dist > n - floor) { done = -1; break; }
				if (dist >= 8) { /* 16-byte pieces never overlap their own source; up to 7 symbols of slack behind the match are overwritten later */
					const uint16_t *sp = out + (n - dist);
					uint16_t *dp = out + n;
					uint32_t k = 0;
					do { memcpy(dp + k, sp + k, 16); k += 8; } while (k < len);
				} else { const size_t src = n - dist; uint32_t k; for (k = 0; k < len; ++k) out[n + k] = out[src + k]; }
				n += len;
				continue;
			}
			done = e.op == PGZ_EOB ? 1 : -1;
			break;
		}
		b->bb = bb; b->bc = bc; b->p = p; pc->n = n;
		if (done) return done > 0 ? 0 : -1;
		if (p > fast_end) break;
	}
	return pgz_codes_careful(b, t, pc);
}

/* one deflate block at b (its 3 header bits first); *final = BFINAL.  0 / -1 */
static inline int pgz_block(pgz_br_t *b, pgz_tabs_t *t, pgz_piece_t *pc, int *final)
{
	uint32_t hdr = pgz_br_get(b, 3);
	int type = (int)(hdr >> 1);
	*final = (int)(hdr & 1);
	if (b->over || type == 3) return -1;
	if (type == 0) {
		uint32_t len, nlen;
		pgz_br_get(b, b->bc & 7); /* to the byte boundary */
		len = pgz_br_get(b, 16); nlen = pgz_br_get(b, 16);
		if (b->over || (len ^ 0xffffu) != nlen) return -1;
		if (pgz_piece_room(pc, len) != 0) return -1;
		b->p -= b->bc >> 3; b->bb = 0; b->bc = 0; /* whole bytes were buffered: hand them back */
		if ((size_t)(b->end - b->p) < len) { size_t k, m = (size_t)(b->end - b->p); for (k = 0; k < m; ++k) pc->sym[pc->n++] = b->p[k]; b->p = b->end; b->over = 1; return -1; }
		{ uint32_t k; for (k = 0; k < len; ++k) pc->sym[pc->n + k] = b->p[k]; }
		pc->n += len; b->p += len;
		return 0;
	}
	if (type == 1) pgz_fixed_tabs(t);
	else if (pgz_dyn_header(b, t) != 0) return -1;
	return pgz_codes(b, t, pc);
}

/* gzip member header at byte p (RFC 1952): bytes it takes, 0 if there is none (end of input / trailing garbage), -1 if it is damaged */
static inline int64_t pgz_member_header(const uint8_t *p, const uint8_t *end)
{
	const uint8_t *q = p;
	int flg;
	if (end - p < 2 || p[0] != 0x1f || p[1] != 0x8b) return 0;
	if (end - p < 10 || p[2] != 8 || (p[3] & 0xe0)) return -1;
	flg = p[3]; q = p + 10;
	if (flg & 4) { size_t xl; if (end - q < 2) return -1; xl = (size_t)q[0] | (size_t)q[1] << 8; q += 2; if ((size_t)(end - q) < xl) return -1; q += xl; }
	if (flg & 8) { while (q < end && *q) ++q; if (q >= end) return -1; ++q; }
	if (flg & 16) { while (q < end && *q) ++q; if (q >= end) return -1; ++q; }
	if (flg & 2) { if (end - q < 2) return -1; q += 2; }
	return q - p;
}

/* decode from (pos, kind) until the first chain position >= stop_bit (or the end of the input).  Fills pc->end / end_kind / ok. */
static inline void pgz_run(const uint8_t *z, size_t zlen, uint64_t pos, int kind, uint64_t stop_bit, pgz_piece_t *pc, pgz_tabs_t *t)
{
	pgz_br_t b;
	pc->ok = 0; pc->at_eof = 0;
	for (;;) {
		int final;
		if (kind == PGZ_AT_MEMBER) {
			int64_t h;
			if (pos >= stop_bit || (pos >> 3) >= zlen) { pc->end = pos; pc->end_kind = kind; pc->ok = 1; pc->at_eof = (pos >> 3) >= zlen; return; }
			h = pgz_member_header(z + (pos >> 3), z + zlen);
			if (h < 0) return;
			if (h == 0) { pc->end = pos; pc->end_kind = kind; pc->ok = 1; pc->at_eof = 1; return; } /* zlib ignores what follows the last member (gzlib: trailing garbage) */
			pos += (uint64_t)h * 8; kind = PGZ_AT_BLOCK; pc->floor = pc->n;
			continue; /* a block may not start at or behind stop_bit either: checked below */
		}
		if (pos >= stop_bit) { pc->end = pos; pc->end_kind = kind; pc->ok = 1; return; }
		pgz_br_init(&b, z, zlen, pos);
		if (pgz_block(&b, t, pc, &final) != 0 || b.over) { pc->end = pgz_br_pos(&b); pc->end_kind = PGZ_AT_BLOCK; return; }
		pos = pgz_br_pos(&b);
		if (final) {
			const uint8_t *q;
			pos = (pos + 7) & ~7ULL;
			q = z + (pos >> 3);
			if ((size_t)(z + zlen - q) < 8) return; /* no trailer */
			if (pc->n_mend == pc->m_mend) { pc->m_mend = pc->m_mend ? pc->m_mend * 2 : 4; pc->mend = (pgz_mend_t*)realloc(pc->mend, sizeof(pgz_mend_t) * (size_t)pc->m_mend); }
			pc->mend[pc->n_mend].at = pc->n - PGZ_WIN;
			pc->mend[pc->n_mend].crc = (uint32_t)q[0] | (uint32_t)q[1] << 8 | (uint32_t)q[2] << 16 | (uint32_t)q[3] << 24;
			pc->mend[pc->n_mend].isize = (uint32_t)q[4] | (uint32_t)q[5] << 8 | (uint32_t)q[6] << 16 | (uint32_t)q[7] << 24;
			++pc->n_mend;
			pos += 64; kind = PGZ_AT_MEMBER;
		}
	}
}

static inline int pgz_piece_reset(pgz_piece_t *pc, const uint8_t *win, int win_len)
{
	int i;
	pc->n = 0; pc->n_mend = 0; pc->ok = 0; pc->at_eof = 0;
	if (pgz_piece_room(pc, PGZ_WIN) != 0) return -1;
	if (win) { /* the last win_len bytes before the piece are known and nothing before them exists */
		for (i = 0; i < win_len; ++i) pc->sym[PGZ_WIN - win_len + i] = win[PGZ_WIN - win_len + i];
		pc->floor = (size_t)(PGZ_WIN - win_len);
	} else {
		for (i = 0; i < PGZ_WIN; ++i) pc->sym[i] = (uint16_t)(0x8000 | i);
		pc->floor = 0;
	}
	pc->n = PGZ_WIN;
	return 0;
}

/* does a dynamic block with BFINAL = 0 start at bit `pos`?  Cheap rejections first. */
static inline int pgz_try_block(const uint8_t *z, size_t zlen, uint64_t pos, pgz_tabs_t *t)
{
	pgz_br_t b;
	pgz_br_init(&b, z, zlen, pos);
	pgz_br_refill(&b);
	if ((b.bb & 7) != 4) return 0;                                   /* BFINAL 0, BTYPE 10b */
	if (((b.bb >> 3) & 31) > 29 || ((b.bb >> 8) & 31) > 29) return 0; /* HLIT, HDIST */
	pgz_br_get(&b, 3);
	return pgz_dyn_header(&b, t) == 0;
}

#if PGZ_X86
/* CRC-32 (the gzip polynomial) by carry-less multiplication: folds 64 bytes per step (Gopal et al., "Fast CRC Computation for Generic
 * Polynomials Using PCLMULQDQ Instruction", Intel 2009; the constants are x^n mod P for the fold distances).  len: a multiple of 16, >= 64;
 * crc in and out as zlib's crc32() passes it. */
__attribute__((target("pclmul,sse4.1")))
static uint32_t pgz_crc32_clmul(uint32_t crc0, const uint8_t *buf, size_t len)
{
	const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596LL, 0x0154442bd4LL);
	const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009eLL, 0x01751997d0LL);
	const __m128i k5k0 = _mm_set_epi64x(0, 0x0163cd6124LL);
	const __m128i poly = _mm_set_epi64x(0x01f7011641LL, 0x01db710641LL);
	const __m128i m32 = _mm_setr_epi32(~0, 0, ~0, 0);
	__m128i x0, x1, x2, x3, x4, x5, x6, x7, x8;
	x1 = _mm_loadu_si128((const __m128i*)(buf + 0)); x2 = _mm_loadu_si128((const __m128i*)(buf + 16));
	x3 = _mm_loadu_si128((const __m128i*)(buf + 32)); x4 = _mm_loadu_si128((const __m128i*)(buf + 48));
	x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)~crc0));
	x0 = k1k2; buf += 64; len -= 64;
	while (len >= 64) {
		x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00); x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
		x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11); x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
		x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), _mm_loadu_si128((const __m128i*)(buf + 0)));
		x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), _mm_loadu_si128((const __m128i*)(buf + 16)));
		x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), _mm_loadu_si128((const __m128i*)(buf + 32)));
		x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), _mm_loadu_si128((const __m128i*)(buf + 48)));
		buf += 64; len -= 64;
	}
	x0 = k3k4;
	x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
	x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
	x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
	while (len >= 16) {
		x2 = _mm_loadu_si128((const __m128i*)buf);
		x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
		buf += 16; len -= 16;
	}
	x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
	x1 = _mm_srli_si128(x1, 8); x1 = _mm_xor_si128(x1, x2);
	x0 = k5k0;
	x2 = _mm_srli_si128(x1, 4); x1 = _mm_and_si128(x1, m32); x1 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_xor_si128(x1, x2);
	x0 = poly;
	x2 = _mm_and_si128(x1, m32); x2 = _mm_clmulepi64_si128(x2, x0, 0x10); x2 = _mm_and_si128(x2, m32); x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
	x1 = _mm_xor_si128(x1, x2);
	return ~(uint32_t)_mm_extract_epi32(x1, 1);
}
#endif
/* crc32() of zlib, 6x faster where the CPU multiplies carry-less (checked against zlib's own in tests/test_pgz.py) */
static inline int pgz_have_clmul(void) /* asked once per stream, before its threads run */
{
#if PGZ_X86
	return __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1") && !getenv("BFC_PGZ_NO_CLMUL");
#else
	return 0;
#endif
}
static inline uint32_t pgz_crc32(uint32_t crc, const uint8_t *buf, size_t len, int have)
{
#if PGZ_X86
	if (have && len >= 64) { const size_t n = len & ~(size_t)15; crc = pgz_crc32_clmul(crc, buf, n); buf += n; len -= n; }
#endif
	while (len) { const size_t step = len < ((size_t)1 << 30) ? len : (size_t)1 << 30; crc = (uint32_t)crc32(crc, buf, (uInt)step); buf += step; len -= step; }
	return crc;
}

typedef struct pgz_s pgz_t;
typedef struct {
	pgz_t *g; int idx;
	uint64_t lo, hi;        /* search from byte lo, stop at the first chain position >= byte hi */
	int exact;              /* start = (g->cur, g->cur_kind) with the known window */
	pgz_piece_t pc;
	pgz_tabs_t tabs;
	/* narrow phase */
	uint8_t win[PGZ_WIN]; uint8_t *dst; int take;
	uint32_t *seg_crc; uint64_t *seg_len; int m_seg; /* CRC-32 / length of the text between member ends (n_mend + 1 of them) */
} pgz_job_t;

struct pgz_s {
	const uint8_t *z; size_t zlen;
	int T; size_t chunk;
	uint64_t cur; int cur_kind;            /* chain position */
	uint8_t win[PGZ_WIN]; int win_len;     /* text before it (right-aligned) */
	uint32_t run_crc; uint64_t run_len;    /* of the current member so far */
	int eof, err;
	uint8_t *text; uint64_t text_off, text_len, text_cap, text_head; /* the text held is text[text_head, text_head + text_len), stream offset text_off */
	pgz_job_t *job; pgz_job_t *redo;       /* redo: the one that decodes again what did not chain */
	uint64_t n_spec, n_redo, n_rounds;     /* pieces taken as guessed / decoded again */
	bfc_pool_t *pool;                      /* the caller's worker threads (NULL: threads are created per phase) */
	int clmul;                             /* CRC-32 by carry-less multiplication */
};

static void *pgz_decode_job(void *arg)
{
	pgz_job_t *j = (pgz_job_t*)arg;
	pgz_t *g = j->g;
	const uint64_t stop = j->hi * 8;
	if (j->exact) {
		if (pgz_piece_reset(&j->pc, g->win, g->win_len) != 0) return 0;
		j->pc.start = g->cur; j->pc.start_kind = g->cur_kind;
		pgz_run(g->z, g->zlen, g->cur, g->cur_kind, stop, &j->pc, &j->tabs);
		return 0;
	}
	{
		uint64_t pos = j->lo * 8;
		const uint64_t lim = stop < (uint64_t)g->zlen * 8 ? stop : (uint64_t)g->zlen * 8;
		j->pc.ok = 0; j->pc.start = ~0ULL;
		for (; pos < lim; ++pos) {
			int kind = -1;
			if ((pos & 7) == 0 && g->z[pos >> 3] == 0x1f && (pos >> 3) + 18 <= g->zlen && g->z[(pos >> 3) + 1] == 0x8b && g->z[(pos >> 3) + 2] == 8 && pgz_member_header(g->z + (pos >> 3), g->z + g->zlen) > 0) kind = PGZ_AT_MEMBER;
			else if (((g->z[pos >> 3] >> (pos & 7)) & 1) == 0 && pgz_try_block(g->z, g->zlen, pos, &j->tabs)) kind = PGZ_AT_BLOCK;
			if (kind < 0) continue;
			if (pgz_piece_reset(&j->pc, 0, 0) != 0) return 0;
			if (kind == PGZ_AT_MEMBER) j->pc.floor = PGZ_WIN;
			j->pc.start = pos; j->pc.start_kind = kind;
			pgz_run(g->z, g->zlen, pos, kind, stop, &j->pc, &j->tabs);
			if (j->pc.ok || j->pc.n > PGZ_WIN + (1u << 16)) return 0; /* an error far into the piece is not a bad guess: give up, the chain decodes it again */
		}
		j->pc.ok = 0; j->pc.start = ~0ULL;
	}
	return 0;
}

static void *pgz_narrow_job(void *arg)
{
	pgz_job_t *j = (pgz_job_t*)arg;
	const uint16_t *s = j->pc.sym + PGZ_WIN;
	const size_t n = j->pc.n - PGZ_WIN;
	size_t i, a = 0;
	int m;
	if (!j->take) return 0;
	i = 0;
#if PGZ_X86
	for (; i + 16 <= n; i += 16) { /* 16 symbols at a time; markers are rare behind the first 32 KiB of most pieces, frequent in FASTQ headers */
		const __m128i a = _mm_loadu_si128((const __m128i*)(s + i)), b = _mm_loadu_si128((const __m128i*)(s + i + 8));
		if (_mm_movemask_epi8(_mm_or_si128(a, b)) & 0xaaaa) { size_t k; for (k = i; k < i + 16; ++k) { const uint16_t v = s[k]; j->dst[k] = (v & 0x8000) ? j->win[v & 0x7fff] : (uint8_t)v; } }
		else _mm_storeu_si128((__m128i*)(j->dst + i), _mm_packus_epi16(a, b));
	}
#endif
	for (; i < n; ++i) { const uint16_t v = s[i]; j->dst[i] = (v & 0x8000) ? j->win[v & 0x7fff] : (uint8_t)v; }
	if (j->m_seg < j->pc.n_mend + 1) { j->m_seg = j->pc.n_mend + 8; j->seg_crc = (uint32_t*)realloc(j->seg_crc, sizeof(uint32_t) * (size_t)j->m_seg); j->seg_len = (uint64_t*)realloc(j->seg_len, sizeof(uint64_t) * (size_t)j->m_seg); }
	for (m = 0; m <= j->pc.n_mend; ++m) {
		const size_t e = m < j->pc.n_mend ? (size_t)j->pc.mend[m].at : n;
		j->seg_crc[m] = pgz_crc32((uint32_t)crc32(0L, Z_NULL, 0), j->dst + a, e - a, j->g->clmul); j->seg_len[m] = e - a; a = e;
	}
	return 0;
}

static inline void pgz_par(pgz_t *g, void *(*fn)(void*), int n)
{
	pthread_t tid[PGZ_MAX_THREADS];
	int i;
	if (g->pool) { bfc_pool_run(g->pool, fn, g->job, sizeof(pgz_job_t), n); return; }
	for (i = 1; i < n; ++i) pthread_create(&tid[i], 0, fn, &g->job[i]);
	fn(&g->job[0]);
	for (i = 1; i < n; ++i) pthread_join(tid[i], 0);
}

static inline pgz_t *pgz_open(const uint8_t *z, size_t zlen, int n_threads, size_t chunk)
{
	pgz_t *g = (pgz_t*)calloc(1, sizeof(pgz_t));
	int i;
	g->z = z; g->zlen = zlen;
	g->T = n_threads < 1 ? 1 : n_threads > PGZ_MAX_THREADS ? PGZ_MAX_THREADS : n_threads;
	g->chunk = chunk < 64 ? 64 : chunk;
	g->cur = 0; g->cur_kind = PGZ_AT_MEMBER; g->win_len = 0;
	g->run_crc = (uint32_t)crc32(0L, Z_NULL, 0);
	g->clmul = pgz_have_clmul();
	g->job = (pgz_job_t*)calloc((size_t)g->T + 1, sizeof(pgz_job_t));
	g->redo = &g->job[g->T];
	for (i = 0; i <= g->T; ++i) { g->job[i].g = g; g->job[i].idx = i; g->job[i].pc.cap0 = PGZ_WIN + g->chunk * 6; }
	return g;
}

static inline void pgz_close(pgz_t *g)
{
	int i;
	if (g == 0) return;
	for (i = 0; i <= g->T; ++i) { free(g->job[i].pc.sym); free(g->job[i].pc.mend); free(g->job[i].seg_crc); free(g->job[i].seg_len); }
	free(g->job); free(g->text); free(g);
}

/* window behind a piece: the last 32 KiB of (window before it, its text) */
static inline void pgz_next_window(const pgz_piece_t *pc, const uint8_t *win, int win_len, uint8_t *out, int *out_len)
{
	const size_t n = pc->n - PGZ_WIN;
	const size_t have = (size_t)win_len + n;
	const int keep = have < PGZ_WIN ? (int)have : PGZ_WIN;
	uint8_t tmp[PGZ_WIN];
	int i;
	/* a member that ended inside the piece cut the history, but bytes before a member start are never referenced (floor): keeping them is harmless */
	for (i = 0; i < keep; ++i) {
		const size_t back = (size_t)(keep - i); /* this many bytes before the end */
		if (back <= n) { const uint16_t v = pc->sym[pc->n - back]; tmp[PGZ_WIN - keep + i] = (v & 0x8000) ? win[v & 0x7fff] : (uint8_t)v; }
		else tmp[PGZ_WIN - keep + i] = win[PGZ_WIN - (back - n)];
	}
	memcpy(out + PGZ_WIN - keep, tmp + PGZ_WIN - keep, (size_t)keep);
	*out_len = keep;
}

static inline int pgz_text_room(pgz_t *g, uint64_t more)
{
	if (g->text_head + g->text_len + more > g->text_cap) {
		if (g->text_head) { memmove(g->text, g->text + g->text_head, (size_t)g->text_len); g->text_head = 0; } /* what the parser has taken goes first */
		if (g->text_len + more > g->text_cap) {
			uint64_t nc = g->text_cap ? g->text_cap : (uint64_t)1 << 24;
			uint8_t *nt;
			while (nc < g->text_len + more) nc += nc < ((uint64_t)1 << 30) ? nc : (uint64_t)1 << 30;
			nc += nc / 2; /* so that a window's worth of taken text can usually stay where it is */
			if ((nt = (uint8_t*)pgz_big_realloc(g->text, (size_t)g->text_len, (size_t)nc)) == 0) return -1; /* (text_head is 0 here) */
			g->text = nt; g->text_cap = nc;
		}
	}
	return 0;
}

/* append one accepted piece's bookkeeping: member ends against the running CRC.  The piece's text is at g->text + at. */
static inline int pgz_account(pgz_t *g, const pgz_job_t *j)
{
	int m;
	for (m = 0; m <= j->pc.n_mend; ++m) {
		g->run_crc = (uint32_t)crc32_combine(g->run_crc, j->seg_crc[m], (z_off_t)j->seg_len[m]);
		g->run_len += j->seg_len[m];
		if (m < j->pc.n_mend) {
			if (g->run_crc != j->pc.mend[m].crc || (uint32_t)g->run_len != j->pc.mend[m].isize) return -1;
			g->run_crc = (uint32_t)crc32(0L, Z_NULL, 0); g->run_len = 0;
		}
	}
	return 0;
}

/* one round: up to T chunks ahead of the chain position.  0, or -1 with g->err set */
static inline int pgz_round(pgz_t *g)
{
	int T = g->T, i, n_take = 0;
	const uint64_t c0 = g->cur >> 3;
	uint64_t total = 0, at;
	uint8_t win[PGZ_WIN]; int win_len = g->win_len;
	uint64_t cur = g->cur; int cur_kind = g->cur_kind;
	if (g->eof || g->err) return g->err ? -1 : 0;
	for (i = 0; i < T; ++i) {
		pgz_job_t *j = &g->job[i];
		j->lo = c0 + (uint64_t)i * g->chunk; j->hi = j->lo + g->chunk; j->exact = i == 0; j->take = 0;
		if (j->hi > g->zlen) j->hi = g->zlen;
		if (j->lo >= g->zlen && i > 0) { T = i; break; }
	}
	if (g->job[T - 1].hi >= g->zlen) g->job[T - 1].hi = g->zlen + 1; /* the last chunk of the file runs to the end of the input */
	pgz_par(g, pgz_decode_job, T);
	++g->n_rounds;
	/* chain */
	memcpy(win, g->win, PGZ_WIN);
	for (i = 0; i < T; ++i) {
		pgz_job_t *j = &g->job[i];
		const uint64_t stop = j->hi * 8;
		if (j->pc.ok && j->pc.start == cur && j->pc.start_kind == cur_kind) { if (i) ++g->n_spec; }
		else if (cur >= stop) continue; /* an earlier piece already covers this chunk (never the file's last: its stop lies behind the input) */
		else { /* decode this chunk again from the known position; the piece replaces the job's own */
			if (i == 0) { g->err = 1; return -1; } /* the exact piece failed: the stream is damaged */
			++g->n_redo;
			if (pgz_piece_reset(&j->pc, win, win_len) != 0) { g->err = 1; return -1; }
			if (cur_kind == PGZ_AT_MEMBER) j->pc.floor = PGZ_WIN;
			j->pc.start = cur; j->pc.start_kind = cur_kind;
			pgz_run(g->z, g->zlen, cur, cur_kind, stop, &j->pc, &j->tabs);
			if (!j->pc.ok) { g->err = 1; return -1; }
		}
		j->take = 1; ++n_take;
		memcpy(j->win, win, PGZ_WIN);
		pgz_next_window(&j->pc, j->win, win_len, win, &win_len);
		cur = j->pc.end; cur_kind = j->pc.end_kind;
		total += j->pc.n - PGZ_WIN;
		if (j->pc.at_eof) { g->eof = 1; T = i + 1; break; }
	}
	if (pgz_text_room(g, total) != 0) { g->err = 1; return -1; }
	at = g->text_len;
	for (i = 0; i < T; ++i) if (g->job[i].take) { g->job[i].dst = g->text + g->text_head + at; at += g->job[i].pc.n - PGZ_WIN; }
	pgz_par(g, pgz_narrow_job, T);
	for (i = 0; i < T; ++i) if (g->job[i].take && pgz_account(g, &g->job[i]) != 0) { g->err = 1; return -1; }
	g->text_len = at;
	memcpy(g->win, win, PGZ_WIN); g->win_len = win_len;
	g->cur = cur; g->cur_kind = cur_kind;
	if (g->eof && g->run_len != 0) { g->err = 1; return -1; } /* the input ended inside a member */
	(void)n_take;
	return 0;
}

/* make text [pos, pos + want) available (less at the end of the input); text before pos is dropped.  Returns 0 and a pointer p with
 * p[off] = byte `off` of the uncompressed stream for pos <= off < *avail_end; -1 if the stream cannot be decoded. */
static inline int pgz_ensure(pgz_t *g, uint64_t pos, uint64_t want, const uint8_t **p, uint64_t *avail_end, int *eof)
{
	if (pos < g->text_off) return -1;
	if (pos > g->text_off) {
		const uint64_t d = pos - g->text_off;
		if (d > g->text_len) return -1;
		g->text_head += d; g->text_len -= d; g->text_off = pos;
		if (g->text_len == 0) g->text_head = 0;
	}
	while (!g->eof && g->text_len < want) if (pgz_round(g) != 0) return -1;
	if (g->err) return -1;
	*p = g->text + g->text_head - g->text_off; *avail_end = g->text_off + g->text_len; *eof = g->eof;
	return 0;
}

#endif
