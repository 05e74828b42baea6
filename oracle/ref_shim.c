/* ref_shim.c -- TEST INFRASTRUCTURE ONLY.
 * Thin exports around the *reference itself*, compiled in place from /root/reference by
 * oracle/Makefile into oracle/_ref/libbfcref.so (never committed; no reference source is copied).
 * It (1) defines the three globals bfc.c:13-15 owns, because bfc.c's main() is not linked into
 * the library, (2) gives the static-inline k-mer functions of kmer.h external names, and
 * (3) provides the sequential counting harness of SURVEY.md App. D.2: the loop of
 * worker_count (count.c:72-89) calling the reference's own bfc_kmer_append / bfc_kmer_hash /
 * bfc_bf_insert / bfc_ch_insert in file order -- proven byte-identical to `bfc -t1 -E -d`
 * (tests/test_oracle.py re-checks that through the dump md5 goldens).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "bfc.h"
#include "kmer.h"

int bfc_verbose = 3;
double bfc_real_time;
bfc_kmer_t bfc_kmer_null = {{0,0,0,0}};

void ref_kmer_append(int k, uint64_t x[4], int c) { bfc_kmer_append(k, x, c); }
uint64_t ref_hash_64(uint64_t key, uint64_t mask) { return bfc_hash_64(key, mask); }
uint64_t ref_kmer_hash(int k, const uint64_t x[4], uint64_t h[2]) { return bfc_kmer_hash(k, x, h); }
void ref_kmer_hash_inv(int k, const uint64_t h[2], uint64_t y[2]) { bfc_kmer_hash_inv(k, h, y); }
int ref_nt6(int c) { return seq_nt6_table[(uint8_t)c]; }

typedef struct { bfc_bf_t *bf, *bf_high; bfc_ch_t *ch; uint64_t n_kmers, n_high, n_seen, hash_xor; } ref_state_t;

ref_state_t *ref_state_new(int k, int bf_shift, int n_hashes, int l_pre, int filter_mode)
{
	ref_state_t *st = (ref_state_t*)calloc(1, sizeof(*st));
	st->bf = bfc_bf_init(bf_shift, n_hashes);
	if (filter_mode) st->bf_high = bfc_bf_init(bf_shift, n_hashes);
	else st->ch = bfc_ch_init(k, l_pre);
	return st;
}
void ref_state_free(ref_state_t *st) { bfc_bf_destroy(st->bf); bfc_bf_destroy(st->bf_high); bfc_ch_destroy(st->ch); free(st); }
bfc_bf_t *ref_state_bf(ref_state_t *st) { return st->bf; }
bfc_bf_t *ref_state_bf_high(ref_state_t *st) { return st->bf_high; }
bfc_ch_t *ref_state_ch(ref_state_t *st) { return st->ch; }
uint8_t *ref_bf_bits(bfc_bf_t *b) { return b->b; }
void ref_state_stats(const ref_state_t *st, uint64_t out[4]) { out[0] = st->n_kmers; out[1] = st->n_high; out[2] = st->n_seen; out[3] = st->hash_xor; }

/* one read, strictly sequential; trace (optional) gets hash,y0,y1,flags per k-mer */
int ref_count_read(ref_state_t *st, int k, int q, int n_hashes, const char *seq, const char *qual, int len, uint64_t *trace)
{
	int i, l = 0, n = 0;
	bfc_kmer_t x = bfc_kmer_null;
	uint64_t qmer = 0, mask = (1ULL << k) - 1;
	for (i = 0; i < len; ++i) {
		int c = seq_nt6_table[(uint8_t)seq[i]] - 1;
		if (c < 4) {
			bfc_kmer_append(k, x.x, c);
			qmer = (qmer << 1 | (qual == 0 || qual[i] - 33 >= q)) & mask;
			if (++l >= k) {
				uint64_t y[2], hash = bfc_kmer_hash(k, x.x, y);
				int is_high = (qmer == mask), seen = (bfc_bf_insert(st->bf, hash) == n_hashes);
				++st->n_kmers; st->n_high += is_high; st->n_seen += seen; st->hash_xor ^= hash * (st->n_kmers | 1);
				if (seen) {
					if (st->ch) bfc_ch_insert(st->ch, y, is_high, 1);
					else if (st->bf_high) bfc_bf_insert(st->bf_high, hash);
				}
				if (trace) { trace[4*n] = hash; trace[4*n+1] = y[0]; trace[4*n+2] = y[1]; trace[4*n+3] = (uint64_t)(is_high | seen << 1); }
				++n;
			}
		} else l = 0, qmer = 0, x = bfc_kmer_null;
	}
	return n;
}
uint64_t ref_count_batch(ref_state_t *st, int k, int q, int n_hashes, const uint8_t *seq, const uint8_t *qual, const uint64_t *off, uint64_t n_reads, uint64_t *trace)
{
	uint64_t r, n = 0;
	for (r = 0; r < n_reads; ++r)
		n += ref_count_read(st, k, q, n_hashes, (const char*)seq + off[r], qual ? (const char*)qual + off[r] : 0, (int)(off[r+1] - off[r]), trace ? trace + 4 * n : 0);
	return n;
}

/* The reference's own ingest (bseq_open / bseq_read = kseq, bseq.c:52-76) digested like bfc_ingest_digest of the product library:
 * out[0] batches, [1] reads, [2] stream positions (bases + one separator per read), [3]/[4] FNV-1a of the sequence / quality streams
 * (read + '\n'; quality + '!', '~' for records without qualities), [5] FNV-1a of the per-batch read counts. */
#include "bseq.h"
int ref_ingest_digest(const char *fn, int chunk_size, uint64_t out[7])
{
	bseq_file_t *fp = bseq_open(fn);
	uint64_t hs = 0xcbf29ce484222325ULL, hq = hs, hb = hs;
	int n, i, j;
	bseq1_t *s;
	memset(out, 0, 7 * sizeof(uint64_t));
	if (fp == 0) return -1;
	while ((s = bseq_read(fp, chunk_size, 0, &n)) != 0 && n > 0) {
		++out[0]; out[1] += (uint64_t)n;
		for (i = 0; i < n; ++i) {
			for (j = 0; j < s[i].l_seq; ++j) { hs = (hs ^ (uint8_t)s[i].seq[j]) * 0x100000001b3ULL; hq = (hq ^ (uint8_t)(s[i].qual ? s[i].qual[j] : '~')) * 0x100000001b3ULL; }
			hs = (hs ^ '\n') * 0x100000001b3ULL; hq = (hq ^ '!') * 0x100000001b3ULL;
			out[2] += (uint64_t)s[i].l_seq + 1;
			free(s[i].name); free(s[i].comment); free(s[i].seq); free(s[i].qual);
		}
		for (i = 0; i < 4; ++i) hb = (hb ^ (((uint64_t)n >> (8 * i)) & 0xff)) * 0x100000001b3ULL;
		free(s);
	}
	if (s) free(s);
	out[3] = hs; out[4] = hq; out[5] = hb;
	bseq_close(fp);
	return 0;
}

/* The same harness spread over threads WITHOUT changing its answers (round 6: the goldens of the eighth-of-human read sets took 2-2.5 h
 * of one core each).  What a k-mer does depends only on the earlier k-mers of ITS OWN 64-byte bloom block (bbf.c:27-31: every bit it
 * tests and sets lies in block `hash & (2^(n_shift-9) - 1)`), the second filter's block is the same one (same hash, count.c:67-68), and
 * table counts commute (saturating adds of htab.c:74-79 under the sub-table's lock).  So part p of n_parts walks ALL reads in file order,
 * hashes every k-mer, and calls the reference's bfc_bf_insert / bfc_ch_insert for the k-mers whose block id is p modulo n_parts only:
 * each block sees exactly its k-mers in file order, i.e. what `bfc -t1` does to it.  Filters, totals and the layout-free table
 * digest (L1) are therefore those of the sequential run -- tests/golden/make_baseline_goldens.py re-derives the committed sequential
 * c5e entry (bf / bf_high checksums, totals) through this function before it adds anything to it.  The table's LAYOUT (the -d dump
 * bytes) is insertion-order dependent and not reproduced by this variant. */
void kt_for(int n_threads, void (*func)(void*, long, int), void *data, long n);
typedef struct {
	ref_state_t *st; int k, q, n_hashes, n_parts; const uint8_t *seq, *qual; const uint64_t *off; uint64_t n_reads, base;
	uint64_t (*acc)[4];
} ref_mt_t;
static void ref_mt_worker(void *data, long part, int tid)
{
	ref_mt_t *m = (ref_mt_t*)data;
	ref_state_t *st = m->st;
	const int k = m->k;
	const uint64_t mask = (1ULL << k) - 1, bmask = (1ULL << (st->bf->n_shift - 9)) - 1;
	uint64_t r, g = m->base, nk = 0, nh = 0, ns = 0, hx = 0;
	for (r = 0; r < m->n_reads; ++r) {
		const char *seq = (const char*)m->seq + m->off[r], *qual = m->qual ? (const char*)m->qual + m->off[r] : 0;
		int i, l = 0, len = (int)(m->off[r+1] - m->off[r]);
		bfc_kmer_t x = bfc_kmer_null;
		uint64_t qmer = 0;
		for (i = 0; i < len; ++i) {
			int c = seq_nt6_table[(uint8_t)seq[i]] - 1;
			if (c < 4) {
				bfc_kmer_append(k, x.x, c);
				qmer = (qmer << 1 | (qual == 0 || qual[i] - 33 >= m->q)) & mask;
				if (++l >= k) {
					uint64_t y[2], hash = bfc_kmer_hash(k, x.x, y);
					++g;
					if ((long)((hash & bmask) % (uint64_t)m->n_parts) == part) {
						int is_high = (qmer == mask), seen = (bfc_bf_insert(st->bf, hash) == m->n_hashes);
						++nk; nh += is_high; ns += seen; hx ^= hash * (g | 1);
						if (seen) {
							if (st->ch) bfc_ch_insert(st->ch, y, is_high, 1);
							else if (st->bf_high) bfc_bf_insert(st->bf_high, hash);
						}
					}
				}
			} else l = 0, qmer = 0, x = bfc_kmer_null;
		}
	}
	m->acc[part][0] = nk; m->acc[part][1] = nh; m->acc[part][2] = ns; m->acc[part][3] = hx;
}
uint64_t ref_count_batch_blocks(ref_state_t *st, int k, int q, int n_hashes, const uint8_t *seq, const uint8_t *qual, const uint64_t *off, uint64_t n_reads, int n_parts)
{
	ref_mt_t m;
	uint64_t n = 0;
	int p;
	m.st = st; m.k = k; m.q = q; m.n_hashes = n_hashes; m.n_parts = n_parts; m.seq = seq; m.qual = qual; m.off = off; m.n_reads = n_reads; m.base = st->n_kmers;
	m.acc = (uint64_t(*)[4])calloc((size_t)n_parts, sizeof(*m.acc));
	kt_for(n_parts, ref_mt_worker, &m, n_parts);
	for (p = 0; p < n_parts; ++p) { n += m.acc[p][0]; st->n_high += m.acc[p][1]; st->n_seen += m.acc[p][2]; st->hash_xor ^= m.acc[p][3]; }
	st->n_kmers += n;
	free(m.acc);
	return n;
}
