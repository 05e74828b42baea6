/* bfc_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.
 *
 * A plain-C, single-threaded restatement of the k-mer counting path of lh3/bfc r181
 * (count.c + bbf.c + htab.c + kmer.h + the khash behaviour that shows in the -d dump),
 * written from the bit-exact specification in SURVEY.md App. A.  Every function names
 * the reference file:line it follows.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library -- as the checker, never as the
 * thing measured or shipped.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this file against
 *   (i)   the single-k-mer known answers captured from the compiled reference,
 *   (ii)  the checksum goldens of `bfc -t1` on fixtures g1/g42: k-mer / is_high /
 *         seen totals, bloom popcount + FNV-1a, distinct keys, histogram mode, `-d` dump md5,
 *         L1 digests,
 *   (iii) live calls into oracle/_ref/libbfcref.so (the reference compiled in place from
 *         /root/reference by oracle/Makefile) when that library is present.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_BLK_SHIFT 9   /* bbf.h:6 : 512-bit (64-byte) bloom blocks */
#define ORC_KEYBITS   50  /* htab.h:7 */
#define ORC_MAXPRE    24  /* htab.h:8 */

/* ------------------------------------------------------------------ k-mer math */

static inline uint64_t orc_mask(int k) { return (1ULL << k) - 1; } /* k <= 63 (bfc.h:8) */

/* base code: A/a 0, C/c 1, G/g 2, T/t 3, anything else 4   (bseq.c:9-26 used as tbl-1, count.c:82) */
int orc_base_code(int ch)
{
	switch (ch) {
	case 'A': case 'a': return 0;
	case 'C': case 'c': return 1;
	case 'G': case 'g': return 2;
	case 'T': case 't': return 3;
	default: return 4;
	}
}

/* kmer.h:10-17 : four k-bit planes; p[0]/p[1] forward low/high bit (newest base at bit 0),
 * p[2]/p[3] reverse complement (newest base's complement enters at bit k-1). */
void orc_kmer_push(int k, uint64_t p[4], int c)
{
	uint64_t m = orc_mask(k);
	uint64_t lo = (uint64_t)(c & 1), hi = (uint64_t)(c >> 1);
	p[0] = ((p[0] << 1) | lo) & m;
	p[1] = ((p[1] << 1) | hi) & m;
	p[2] = (p[2] >> 1) | ((lo ^ 1) << (k - 1));
	p[3] = (p[3] >> 1) | ((hi ^ 1) << (k - 1));
}

/* kmer.h:30-40 : Thomas Wang 64-bit mix with every addition reduced to k bits */
uint64_t orc_mix64(uint64_t v, uint64_t m)
{
	v = (~v + (v << 21)) & m;
	v ^= v >> 24;
	v = (v + (v << 3) + (v << 8)) & m;
	v ^= v >> 14;
	v = (v + (v << 2) + (v << 4)) & m;
	v ^= v >> 28;
	v = (v + (v << 31)) & m;
	return v;
}

/* kmer.h:79-88 : strand-canonical two-word hash.  Returns the bloom hash; y[0]=(h0+h1)&m, y[1]=h1. */
uint64_t orc_kmer_hash(int k, const uint64_t p[4], uint64_t y[2])
{
	int t = k >> 1;
	int rev = ((p[1] >> t) & 1) > ((p[3] >> t) & 1);
	uint64_t m = orc_mask(k);
	uint64_t a = p[rev ? 2 : 0], b = p[rev ? 3 : 1];
	uint64_t h0 = orc_mix64((a + b) & m, m);
	uint64_t h1 = orc_mix64(h0 ^ b, m);
	y[0] = (h0 + h1) & m;
	y[1] = h1;
	return ((h0 ^ h1) << k) | y[0];
}

/* the bloom hash is a function of y alone (cf. kmer.h:85-86): h0 = (y0 - y1) & m */
uint64_t orc_hash_from_y(int k, const uint64_t y[2])
{
	uint64_t m = orc_mask(k), h0 = (y[0] - y[1]) & m;
	return ((h0 ^ y[1]) << k) | y[0];
}

/* ------------------------------------------------------------------ blocked bloom filter */

typedef struct { int n_shift, n_hashes; uint8_t *b; } orc_bf_t; /* bbf.h:9-12 */

orc_bf_t *orc_bf_new(int n_shift, int n_hashes) /* bbf.c:5-17 */
{
	orc_bf_t *f;
	if (n_shift + ORC_BLK_SHIFT > 64 || n_shift < ORC_BLK_SHIFT) return 0;
	f = (orc_bf_t*)calloc(1, sizeof(*f));
	f->n_shift = n_shift; f->n_hashes = n_hashes;
	f->b = (uint8_t*)calloc(1ULL << (n_shift - 3), 1);
	return f;
}
void orc_bf_free(orc_bf_t *f) { if (f) { free(f->b); free(f); } }
uint8_t *orc_bf_bits(orc_bf_t *f) { return f->b; }
uint64_t orc_bf_nbytes(const orc_bf_t *f) { return 1ULL << (f->n_shift - 3); }

/* enumerate the bit positions one hash touches (bbf.c:27-41): block, then n_hashes positions in
 * [8,512) walking z=h1, h1+h2, ... mod 512 and skipping z<8 (the lock byte). */
static inline uint64_t orc_bf_addr(int n_shift, int n_hashes, uint64_t hash, int pos[/*n_hashes*/])
{
	int x = n_shift - ORC_BLK_SHIFT, i, z;
	uint64_t blk = hash & ((1ULL << x) - 1);
	int h1 = (int)((hash >> x) & 511);
	int h2 = (int)((hash >> n_shift) & 511);
	if ((h2 & 31) == 0) h2 = (h2 + 1) & 511;
	for (i = 0, z = h1; i < n_hashes; z = (z + h2) & 511) {
		if (z < 8) continue;
		pos[i++] = z;
	}
	return blk;
}
/* exported for the KAT test: returns block, fills pos[] */
uint64_t orc_bf_positions(int n_shift, int n_hashes, uint64_t hash, int *pos)
{ return orc_bf_addr(n_shift, n_hashes, hash, pos); }

int orc_bf_insert(orc_bf_t *f, uint64_t hash) /* bbf.c:25-45 ; returns # bits already set */
{
	int pos[64], i, cnt = 0;
	uint64_t blk = orc_bf_addr(f->n_shift, f->n_hashes, hash, pos);
	uint8_t *p = f->b + (blk << (ORC_BLK_SHIFT - 3));
	for (i = 0; i < f->n_hashes; ++i) {
		uint8_t u = (uint8_t)(1u << (pos[i] & 7));
		cnt += (p[pos[i] >> 3] & u) != 0;
		p[pos[i] >> 3] |= u;
	}
	return cnt;
}
int orc_bf_get(const orc_bf_t *f, uint64_t hash) /* bbf.c:47-63 */
{
	int pos[64], i, cnt = 0;
	uint64_t blk = orc_bf_addr(f->n_shift, f->n_hashes, hash, pos);
	const uint8_t *p = f->b + (blk << (ORC_BLK_SHIFT - 3));
	for (i = 0; i < f->n_hashes; ++i)
		cnt += (p[pos[i] >> 3] >> (pos[i] & 7)) & 1;
	return cnt;
}

uint64_t orc_popcount_bytes(const uint8_t *b, uint64_t n)
{
	uint64_t i, c = 0;
	const uint64_t *w = (const uint64_t*)b;
	for (i = 0; i < n / 8; ++i) c += (uint64_t)__builtin_popcountll(w[i]);
	for (i = n & ~7ULL; i < n; ++i) c += (uint64_t)__builtin_popcount(b[i]);
	return c;
}
uint64_t orc_fnv1a64(const uint8_t *b, uint64_t n) /* checksum used by the SURVEY B.3 goldens */
{
	uint64_t h = 0xcbf29ce484222325ULL, i;
	for (i = 0; i < n; ++i) { h ^= b[i]; h *= 0x100000001b3ULL; }
	return h;
}

/* ------------------------------------------------------------------ k-mer count table
 * 2^l_pre open-addressing sets of u64 slots  key(50)<<14 | high(6)<<8 | count(8)   (htab.c:7-17)
 * with the khash behaviour of SURVEY A.7 (khash.h:219-336) so that the dump is byte-comparable. */

typedef struct { uint32_t nb, size; uint32_t *used; uint64_t *slot; uint8_t late_call; } orc_sub_t;
typedef struct { int k, l_pre; orc_sub_t *sub; } orc_ch_t;

static inline int sub_used(const orc_sub_t *s, uint32_t i) { return (s->used[i >> 5] >> (i & 31)) & 1; }
static inline void sub_mark(uint32_t *used, uint32_t i) { used[i >> 5] |= 1u << (i & 31); }

orc_ch_t *orc_ch_new(int k, int l_pre) /* htab.c:19-34 */
{
	orc_ch_t *ch;
	if (k > 63) return 0;
	if (k * 2 - l_pre > ORC_KEYBITS) l_pre = k * 2 - ORC_KEYBITS;
	if (l_pre > ORC_MAXPRE) l_pre = ORC_MAXPRE;
	ch = (orc_ch_t*)calloc(1, sizeof(*ch));
	ch->k = k; ch->l_pre = l_pre;
	ch->sub = (orc_sub_t*)calloc((size_t)1 << l_pre, sizeof(orc_sub_t));
	return ch;
}
void orc_ch_free(orc_ch_t *ch)
{
	size_t i;
	if (!ch) return;
	for (i = 0; i < (size_t)1 << ch->l_pre; ++i) { free(ch->sub[i].used); free(ch->sub[i].slot); }
	free(ch->sub); free(ch);
}
int orc_ch_k(const orc_ch_t *ch) { return ch->k; }
int orc_ch_lpre(const orc_ch_t *ch) { return ch->l_pre; }

/* htab.c:45-58 : (y0,y1) -> sub-table index and slot key (count field preset to 1) */
uint64_t orc_ch_subkey(int k, int l_pre, const uint64_t y[2], uint64_t *key)
{
	if (k <= 32) {
		int t = k * 2 - l_pre;
		uint64_t z = (y[0] << k) | y[1];
		*key = ((z & ((1ULL << t) - 1)) << 14) | 1;
		return z >> t;
	} else {
		int t = k - l_pre;
		int sh = (t + k < ORC_KEYBITS) ? k : ORC_KEYBITS - t;
		*key = ((((y[0] & ((1ULL << t) - 1)) << sh) ^ y[1]) << 14) | 1;
		return y[0] >> t;
	}
}
int orc_ch_clamp_lpre(int k, int l_pre)
{
	if (k * 2 - l_pre > ORC_KEYBITS) l_pre = k * 2 - ORC_KEYBITS;
	if (l_pre > ORC_MAXPRE) l_pre = ORC_MAXPRE;
	return l_pre;
}

/* grow/rehash to nb_new buckets (khash.h:233-293).  The reference rehashes in place with a
 * kick-out chain; the resulting layout equals inserting into a fresh array in this order:
 * walk old buckets upward; an element not yet moved is placed by triangular probing in the new
 * array; if it lands on an index that still holds an unmoved old element, that element is the
 * next one placed. */
static void sub_rehash(orc_sub_t *s, uint32_t nb_new)
{
	uint32_t nm = nb_new - 1, j;
	uint32_t *nused = (uint32_t*)calloc((nb_new + 31) / 32, 4);
	uint64_t *nslot = (uint64_t*)calloc(nb_new, 8);
	uint32_t *moved = s->nb ? (uint32_t*)calloc((s->nb + 31) / 32, 4) : 0;
	for (j = 0; j < s->nb; ++j) {
		uint64_t cur;
		if (!sub_used(s, j) || ((moved[j >> 5] >> (j & 31)) & 1)) continue;
		cur = s->slot[j]; sub_mark(moved, j);
		for (;;) {
			uint32_t i = (uint32_t)(cur >> 14) & nm, step = 0;
			while ((nused[i >> 5] >> (i & 31)) & 1) i = (i + (++step)) & nm;
			sub_mark(nused, i);
			if (i < s->nb && sub_used(s, i) && !((moved[i >> 5] >> (i & 31)) & 1)) {
				uint64_t ev = s->slot[i]; /* evict the not-yet-moved resident of index i */
				sub_mark(moved, i);
				nslot[i] = cur; cur = ev;
			} else { nslot[i] = cur; break; }
		}
	}
	free(moved); free(s->used); free(s->slot);
	s->used = nused; s->slot = nslot; s->nb = nb_new;
}

/* khash.h:295-336 restricted to the no-deletion case: the 0.75 check runs on EVERY call */
static uint32_t sub_put(orc_sub_t *s, uint64_t key, int *absent)
{
	uint32_t mask, i, step = 0, last;
	if (s->size >= (s->nb >> 2) + (s->nb >> 1)) sub_rehash(s, s->nb ? s->nb << 1 : 4);
	mask = s->nb - 1;
	i = (uint32_t)(key >> 14) & mask; last = i;
	while (sub_used(s, i) && (s->slot[i] >> 14) != (key >> 14)) {
		i = (i + (++step)) & mask;
		if (i == last) break; /* cannot happen below 75 % load */
	}
	if (!sub_used(s, i)) { s->slot[i] = key; sub_mark(s->used, i); ++s->size; *absent = 1; }
	else *absent = 0;
	return i;
}

int orc_ch_insert(orc_ch_t *ch, const uint64_t y[2], int is_high) /* htab.c:60-82 */
{
	uint64_t key, sub = orc_ch_subkey(ch->k, ch->l_pre, y, &key);
	orc_sub_t *s = &ch->sub[sub];
	int absent;
	uint32_t i = sub_put(s, key, &absent);
	if (absent) { if (is_high) s->slot[i] |= 1 << 8; }
	else {
		if ((s->slot[i] & 0xff) != 0xff) ++s->slot[i];
		if (is_high && ((s->slot[i] >> 8) & 0x3f) != 0x3f) s->slot[i] += 1 << 8;
	}
	return 0;
}
int orc_ch_get(const orc_ch_t *ch, const uint64_t y[2]) /* htab.c:84-92, khash.h:219-232 */
{
	uint64_t key, sub = orc_ch_subkey(ch->k, ch->l_pre, y, &key);
	const orc_sub_t *s = &ch->sub[sub];
	uint32_t mask, i, step = 0, last;
	if (s->nb == 0) return -1;
	mask = s->nb - 1; i = (uint32_t)(key >> 14) & mask; last = i;
	while (sub_used(s, i) && (s->slot[i] >> 14) != (key >> 14)) {
		i = (i + (++step)) & mask;
		if (i == last) return -1;
	}
	return sub_used(s, i) ? (int)(s->slot[i] & 0x3fff) : -1;
}
uint64_t orc_ch_count(const orc_ch_t *ch) /* htab.c:101-108 */
{
	uint64_t n = 0; size_t i;
	for (i = 0; i < (size_t)1 << ch->l_pre; ++i) n += ch->sub[i].size;
	return n;
}
int orc_ch_hist(const orc_ch_t *ch, uint64_t cnt[256], uint64_t high[64]) /* htab.c:110-127 */
{
	size_t i; uint32_t j; int best = -1; uint64_t max = 0;
	memset(cnt, 0, 256 * 8); memset(high, 0, 64 * 8);
	for (i = 0; i < (size_t)1 << ch->l_pre; ++i) {
		const orc_sub_t *s = &ch->sub[i];
		for (j = 0; j < s->nb; ++j)
			if (sub_used(s, j)) { ++cnt[s->slot[j] & 0xff]; ++high[(s->slot[j] >> 8) & 0x3f]; }
	}
	for (j = 3; j < 256; ++j) if (cnt[j] > max) { max = cnt[j]; best = (int)j; }
	return best;
}
int orc_ch_dump(const orc_ch_t *ch, const char *fn) /* htab.c:129-149, format SURVEY A.7 */
{
	FILE *fp = fopen(fn, "wb");
	uint32_t t[2]; size_t i; uint32_t j;
	if (!fp) return -1;
	t[0] = (uint32_t)ch->k; t[1] = (uint32_t)ch->l_pre; fwrite(t, 4, 2, fp);
	for (i = 0; i < (size_t)1 << ch->l_pre; ++i) {
		const orc_sub_t *s = &ch->sub[i];
		t[0] = s->nb; t[1] = s->size; fwrite(t, 4, 2, fp);
		for (j = 0; j < s->nb; ++j) if (sub_used(s, j)) fwrite(&s->slot[j], 8, 1, fp);
	}
	fclose(fp);
	return 0;
}
static int cmp_u64(const void *a, const void *b)
{ uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? -1 : x > y; }

/* L1 export: sizes[2^l_pre] and, concatenated in sub-table order, each sub-table's
 * slot values sorted ascending.  slots may be NULL to query sizes only. */
uint64_t orc_ch_export(const orc_ch_t *ch, uint32_t *sizes, uint64_t *slots)
{
	size_t i; uint32_t j; uint64_t n = 0;
	for (i = 0; i < (size_t)1 << ch->l_pre; ++i) {
		const orc_sub_t *s = &ch->sub[i];
		uint64_t n0 = n;
		if (sizes) sizes[i] = s->size;
		if (slots) {
			for (j = 0; j < s->nb; ++j) if (sub_used(s, j)) slots[n++] = s->slot[j];
			qsort(slots + n0, n - n0, 8, cmp_u64);
		} else n += s->size;
	}
	return n;
}

/* ------------------------------------------------------------------ count driver */

typedef struct {
	int k, q, n_hashes, bf_shift, l_pre, filter_mode;
	orc_bf_t *bf, *bf_high;
	orc_ch_t *ch;
	uint64_t n_kmers, n_high, n_seen, hash_xor;
} orc_state_t;

orc_state_t *orc_state_new(int k, int q, int bf_shift, int n_hashes, int l_pre, int filter_mode) /* count.c:127-141,148-149 */
{
	orc_state_t *st = (orc_state_t*)calloc(1, sizeof(*st));
	st->k = k; st->q = q; st->n_hashes = n_hashes; st->bf_shift = bf_shift; st->filter_mode = filter_mode;
	st->bf = orc_bf_new(bf_shift, n_hashes);
	if (!filter_mode) { st->ch = orc_ch_new(k, l_pre); st->l_pre = st->ch->l_pre; }
	else st->bf_high = orc_bf_new(bf_shift, n_hashes);
	return st;
}
void orc_state_free(orc_state_t *st)
{ if (st) { orc_bf_free(st->bf); orc_bf_free(st->bf_high); orc_ch_free(st->ch); free(st); } }
orc_bf_t *orc_state_bf(orc_state_t *st) { return st->bf; }
orc_bf_t *orc_state_bf_high(orc_state_t *st) { return st->bf_high; }
orc_ch_t *orc_state_ch(orc_state_t *st) { return st->ch; }
void orc_state_stats(const orc_state_t *st, uint64_t out[4])
{ out[0] = st->n_kmers; out[1] = st->n_high; out[2] = st->n_seen; out[3] = st->hash_xor; }

/* One read through worker_count (count.c:72-89) and bfc_kmer_insert (count.c:54-70), strictly in
 * order.  qual may be NULL (FASTA: every base counts as high quality, count.c:85).
 * If trace != NULL it receives 4 u64 per k-mer: hash, y0, y1, flags(bit0 is_high, bit1 seen).
 * Returns the number of k-mers of this read. */
int orc_count_read(orc_state_t *st, const uint8_t *seq, const uint8_t *qual, int len, uint64_t *trace)
{
	int k = st->k, i, l = 0, n = 0;
	uint64_t p[4] = {0, 0, 0, 0}, qmer = 0, m = orc_mask(k);
	for (i = 0; i < len; ++i) {
		int c = orc_base_code(seq[i]);
		if (c < 4) {
			orc_kmer_push(k, p, c);
			qmer = ((qmer << 1) | (uint64_t)(qual == 0 || (int)(signed char)qual[i] - 33 >= st->q)) & m; /* count.c:85: s->qual is a char*, signed here */
			if (++l >= k) {
				uint64_t y[2], hash = orc_kmer_hash(k, p, y);
				int is_high = (qmer == m);
				int seen = (orc_bf_insert(st->bf, hash) == st->n_hashes);
				++st->n_kmers; st->n_high += (uint64_t)is_high; st->n_seen += (uint64_t)seen;
				st->hash_xor ^= hash * (st->n_kmers | 1);
				if (seen) {
					if (st->ch) orc_ch_insert(st->ch, y, is_high);
					else if (st->bf_high) orc_bf_insert(st->bf_high, hash);
				}
				if (trace) { trace[4*n] = hash; trace[4*n+1] = y[0]; trace[4*n+2] = y[1]; trace[4*n+3] = (uint64_t)(is_high | seen << 1); }
				++n;
			}
		} else { l = 0; qmer = 0; p[0] = p[1] = p[2] = p[3] = 0; }
	}
	return n;
}

/* A batch in the SoA form the GPU path takes: reads concatenated in seq[]/qual[] (no separators),
 * read r occupying [off[r], off[r+1]).  qual == NULL means FASTA. */
uint64_t orc_count_batch(orc_state_t *st, const uint8_t *seq, const uint8_t *qual, const uint64_t *off, uint64_t n_reads, uint64_t *trace)
{
	uint64_t r, n = 0;
	for (r = 0; r < n_reads; ++r)
		n += (uint64_t)orc_count_read(st, seq + off[r], qual ? qual + off[r] : 0, (int)(off[r+1] - off[r]), trace ? trace + 4 * n : 0);
	return n;
}

/* number of k-mers (calls of bfc_kmer_insert) a read contributes: sum over ACGT runs of max(0,len-k+1) */
uint64_t orc_kmers_in_read(const uint8_t *seq, int len, int k)
{
	int i, l = 0; uint64_t n = 0;
	for (i = 0; i < len; ++i) { if (orc_base_code(seq[i]) < 4) { if (++l >= k) ++n; } else l = 0; }
	return n;
}

/* ------------------------------------------------------------------ trim pass (filter mode)
 * max_streak (correct.c:478-497): hi32 = longest run of bloom-hit k-mers, lo32 = index+1 ... as in
 * the reference: t accumulates 1<<32 per hit, resets to i+1 otherwise; returns the max t. */
uint64_t orc_max_streak(int k, const orc_bf_t *bf, const uint8_t *seq, int len)
{
	int i, l = 0;
	uint64_t max = 0, t = 0, p[4] = {0, 0, 0, 0};
	for (i = 0; i < len; ++i) {
		int c = orc_base_code(seq[i]);
		if (c < 4) {
			orc_kmer_push(k, p, c);
			if (++l >= k) {
				uint64_t y[2], hash = orc_kmer_hash(k, p, y);
				if (orc_bf_get(bf, hash) == bf->n_hashes) t += 1ULL << 32;
				else t = (uint64_t)i + 1;
			} else t = (uint64_t)i + 1;
		} else { l = 0; p[0] = p[1] = p[2] = p[3] = 0; t = (uint64_t)i + 1; }
		if (t > max) max = t;
	}
	return max;
}
/* keep/trim decision of correct.c:557-569.  Returns 1 and [*start,*end) if the read is kept. */
int orc_trim_decide(uint64_t max, int k, int len, float min_frac, int *start, int *end) /* min_frac is a float in bfc_opt_t (bfc.h:21): 0.9f < 0.9 */
{
	if ((max >> 32) && (double)((max >> 32) + (uint64_t)k) / len > min_frac) {
		int s = (int)(uint32_t)max, e = s + (int)(max >> 32);
		*start = s - (k - 1); *end = e;
		return 1;
	}
	return 0;
}

/* ------------------------------------------------------------------ k-mer coverage of the corrector
 * bfc_ec_kcov (correct.c:96-117) on one read: for the k-mer ending at base i, r = bfc_ch_kmer_occ (htab.c:94-99);
 * high_end if the high count (r>>8&0x3f) >= min_occ+1, solid_end if the count (r&0xff) >= min_occ, and every base of a
 * solid k-mer gets ++lcov, hcov += high_end.  lcov/hcov are 6-bit fields of ecbase_t (correct.c:17) -- k <= 63 cannot
 * wrap them.  out[i] = lcov | hcov<<6 | solid_end<<12 | high_end<<13. */
void orc_kcov(const orc_ch_t *ch, int min_occ, const uint8_t *seq, int len, uint16_t *out)
{
	int i, j, l = 0, k = ch->k;
	uint64_t p[4] = {0, 0, 0, 0};
	uint8_t *lc = (uint8_t*)calloc(len + 1, 1), *hc = (uint8_t*)calloc(len + 1, 1);
	for (i = 0; i < len; ++i) out[i] = 0;
	for (i = 0; i < len; ++i) {
		int c = orc_base_code(seq[i]);
		if (c < 4) {
			orc_kmer_push(k, p, c);
			if (++l >= k) {
				uint64_t y[2]; int r;
				orc_kmer_hash(k, p, y);
				if ((r = orc_ch_get(ch, y)) >= 0) {
					int high_end = (r >> 8 & 0x3f) >= min_occ + 1;
					if (high_end) out[i] |= 1u << 13;
					if ((r & 0xff) >= min_occ) {
						out[i] |= 1u << 12;
						for (j = i - k + 1; j <= i; ++j) lc[j] = (lc[j] + 1) & 63, hc[j] = (hc[j] + high_end) & 63;
					}
				}
			}
		} else { l = 0; p[0] = p[1] = p[2] = p[3] = 0; }
	}
	for (i = 0; i < len; ++i) out[i] |= lc[i] | hc[i] << 6;
	free(lc); free(hc);
}
