/* bfc_host.c -- host side of the reference-shaped API (include/bfc_gpu.h, PART 1):
 * the query functions correct.c calls per k-mer from its worker threads (bfc_bf_get,
 * bfc_ch_get, bfc_ch_kmer_occ, bfc_ch_hist) plus the single-element insert / dump / restore
 * entry points of bbf.h and htab.h.  These operate on HOST copies of what the GPU built; the
 * counting itself (bfc_count) never runs here.
 *
 * The host table keeps the GPU layout: 2^l_pre regions of 2^cshift u64 slots, slot value
 * key(50)<<14 | high(6)<<8 | count(8) exactly as htab.c:7-17, 0 = empty, home slot = low bits of
 * key>>14, linear probing inside the region.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include <pthread.h>
#include <sys/stat.h>
#include <unistd.h>
#include <sys/mman.h>
#include "bfc_gpu.h"
#include "bfc_host.h"

/* ------------------------------------------------------------------ k-mer hash (kmer.h:30-40,79-88) */

static inline uint64_t mix_k(uint64_t v, uint64_t m)
{
	v = (~v + (v << 21)) & m; v ^= v >> 24;
	v = (v + (v << 3) + (v << 8)) & m; v ^= v >> 14;
	v = (v + (v << 2) + (v << 4)) & m; v ^= v >> 28;
	v = (v + (v << 31)) & m;
	return v;
}
static inline void kmer_y(int k, const uint64_t x[4], uint64_t y[2])
{
	int t = k >> 1, rev = ((x[1] >> t) & 1) > ((x[3] >> t) & 1);
	uint64_t m = (1ULL << k) - 1, a = x[rev << 1], b = x[rev << 1 | 1];
	uint64_t h0 = mix_k((a + b) & m, m), h1 = mix_k(h0 ^ b, m);
	y[0] = (h0 + h1) & m; y[1] = h1;
}

/* ------------------------------------------------------------------ bloom filter (bbf.c) */

/* A filter exported by bfc_count may still have its copy in HBM (bfcg_export_bloom_resident), for the trim pass to adopt.  That copy
 * dies with the host object and as soon as the host object is written to.  Weak: this file also links without the GPU objects. */
void bfcg_resident_drop(const void *bf) __attribute__((weak));

bfc_bf_t *bfc_bf_alloc_raw(int n_shift, int n_hashes)
{
	bfc_bf_t *b; void *p = 0;
	if (n_shift + BFC_BLK_SHIFT > 64 || n_shift < BFC_BLK_SHIFT) return 0;
	b = (bfc_bf_t*)calloc(1, sizeof(bfc_bf_t));
	if (!b) return 0;
	if (bfcg_resident_drop) bfcg_resident_drop(b); /* an address can come back: whatever was registered for it belongs to a dead object */
	b->n_shift = n_shift; b->n_hashes = n_hashes;
	if (posix_memalign(&p, 64, 1ULL << (n_shift - 3)) != 0) { free(b); return 0; }
	b->b = (uint8_t*)p;
	return b;
}
bfc_bf_t *bfc_bf_init(int n_shift, int n_hashes)
{
	bfc_bf_t *b = bfc_bf_alloc_raw(n_shift, n_hashes);
	if (b) memset(b->b, 0, 1ULL << (n_shift - 3));
	return b;
}
void bfc_bf_destroy(bfc_bf_t *b)
{
	if (!b) return;
	if (bfcg_resident_drop) bfcg_resident_drop(b);
	free(b->b); free(b);
}

static inline uint8_t *bf_walk(const bfc_bf_t *b, uint64_t hash, int *h1, int *h2)
{
	int x = b->n_shift - BFC_BLK_SHIFT;
	*h1 = (int)(hash >> x) & BFC_BLK_MASK;
	*h2 = (int)(hash >> b->n_shift) & BFC_BLK_MASK;
	if ((*h2 & 31) == 0) *h2 = (*h2 + 1) & BFC_BLK_MASK;
	return b->b + ((hash & ((1ULL << x) - 1)) << (BFC_BLK_SHIFT - 3));
}
int bfc_bf_insert(bfc_bf_t *b, uint64_t hash) /* single-k-mer host insert; atomic OR instead of the block spin lock */
{
	int h1, h2, i, z, cnt = 0;
	uint8_t *p = bf_walk(b, hash, &h1, &h2);
	if (bfcg_resident_drop) bfcg_resident_drop(b); /* the copy in HBM, if any, is stale from here on (one relaxed load when there is none) */
	for (i = 0, z = h1; i < b->n_hashes; z = (z + h2) & BFC_BLK_MASK) {
		uint8_t u;
		if (z < 8) continue;
		u = (uint8_t)(1u << (z & 7));
		cnt += (__sync_fetch_and_or(&p[z >> 3], u) & u) != 0;
		++i;
	}
	return cnt;
}
int bfc_bf_get(const bfc_bf_t *b, uint64_t hash)
{
	int h1, h2, i, z, cnt = 0;
	const uint8_t *p = bf_walk(b, hash, &h1, &h2);
	for (i = 0, z = h1; i < b->n_hashes; z = (z + h2) & BFC_BLK_MASK) {
		if (z < 8) continue;
		cnt += (p[z >> 3] >> (z & 7)) & 1;
		++i;
	}
	return cnt;
}

/* ------------------------------------------------------------------ count table (htab.c) */

struct bfc_ch_s {
	int k, l_pre, cshift;
	uint64_t *slots;
	uint64_t n_keys;
	pthread_rwlock_t grow_lock; /* inserts share it, growth owns it */
	/* optional order stamps from the GPU build (bfcg_params_t.track_order): first[slot] = (batch << 32 | file index) of the
	 * bfc_ch_insert call that created the key, sub_last[sub] = stamp of the last call of any kind on that sub-table.  They let
	 * bfc_ch_dump replay khash and write `bfc -t1 -d`'s bytes.  Dropped as soon as the host inserts or restores. */
	uint64_t *first, *sub_last;
};

static int clamp_lpre(int k, int l_pre)
{
	if (k * 2 - l_pre > BFC_CH_KEYBITS) l_pre = k * 2 - BFC_CH_KEYBITS;
	if (l_pre > BFC_CH_MAXPRE) l_pre = BFC_CH_MAXPRE;
	return l_pre;
}
/* A large table (c3: 4 GiB, human: 64 GiB) on 2 MiB pages where the kernel gives them: the table arrives from the device through threads that touch
 * every page once (0.2 s of c3's whole-file wall time went into 4 KiB faults), and main()'s bfc_ch_destroy tears the mapping down again.  calloc's
 * block is a fresh mapping for such sizes; the advice covers its 2 MiB-aligned inside. */
static void huge_advice(void *p, size_t bytes)
{
#ifdef MADV_HUGEPAGE
	const uintptr_t al = (uintptr_t)2 << 20, a = ((uintptr_t)p + al - 1) & ~(al - 1), e = ((uintptr_t)p + bytes) & ~(al - 1);
	if (bytes >= ((size_t)64 << 20) && e > a && !getenv("BFC_GPU_NO_THP")) (void)madvise((void*)a, (size_t)(e - a), MADV_HUGEPAGE);
#endif
}
bfc_ch_t *bfc_ch_alloc_raw(int k, int l_pre, int cshift)
{
	bfc_ch_t *ch = (bfc_ch_t*)calloc(1, sizeof(bfc_ch_t));
	if (!ch) return 0;
	ch->k = k; ch->l_pre = l_pre; ch->cshift = cshift;
	ch->slots = (uint64_t*)calloc((size_t)1 << (l_pre + cshift), 8);
	if (!ch->slots) { free(ch); return 0; }
	huge_advice(ch->slots, (size_t)8 << (l_pre + cshift));
	pthread_rwlock_init(&ch->grow_lock, 0);
	return ch;
}
uint64_t *bfc_ch_raw_slots(bfc_ch_t *ch) { return ch->slots; }
int bfc_ch_raw_cshift(const bfc_ch_t *ch) { return ch->cshift; }
int bfc_ch_raw_order(bfc_ch_t *ch, uint64_t **first, uint64_t **sub_last)
{
	ch->first = (uint64_t*)malloc((size_t)8 << (ch->l_pre + ch->cshift));
	ch->sub_last = (uint64_t*)malloc((size_t)8 << ch->l_pre);
	if (!ch->first || !ch->sub_last) { free(ch->first); free(ch->sub_last); ch->first = ch->sub_last = 0; return -1; }
	*first = ch->first; *sub_last = ch->sub_last;
	return 0;
}
static void drop_order(bfc_ch_t *ch) { free(ch->first); free(ch->sub_last); ch->first = ch->sub_last = 0; }
/* one pass over all slots: keys, count histogram, high-count histogram.  A human-sized table is 2^33 slots (64 GiB): the pass is split
 * over up to 16 threads from 2^24 slots on (read-only, so callers' concurrent reads stay safe). */
typedef struct { const uint64_t *slots; uint64_t beg, end, keys, cnt[256], high[64]; } scan_t;
static void *scan_worker(void *p)
{
	scan_t *w = (scan_t*)p;
	uint64_t i;
	for (i = w->beg; i < w->end; ++i) {
		uint64_t v = w->slots[i];
		if (v) { ++w->keys; ++w->cnt[v & 0xff]; ++w->high[v >> 8 & 0x3f]; }
	}
	return 0;
}
static uint64_t scan_slots(const bfc_ch_t *ch, uint64_t cnt[256], uint64_t high[64])
{
	uint64_t n = (uint64_t)1 << (ch->l_pre + ch->cshift), keys = 0;
	int T = 1, t, j;
	scan_t *w;
	pthread_t tid[16];
	if (n >= (1ULL << 24)) { long nc = sysconf(_SC_NPROCESSORS_ONLN); T = nc >= 16 ? 16 : nc > 1 ? (int)nc : 1; }
	w = (scan_t*)calloc((size_t)T, sizeof(scan_t));
	for (t = 0; t < T; ++t) { w[t].slots = ch->slots; w[t].beg = n / T * t; w[t].end = t == T - 1 ? n : n / T * (t + 1); }
	for (t = 1; t < T; ++t) if (pthread_create(&tid[t], 0, scan_worker, &w[t]) != 0) { scan_worker(&w[t]); tid[t] = 0; }
	scan_worker(&w[0]);
	for (t = 1; t < T; ++t) if (tid[t]) pthread_join(tid[t], 0);
	if (cnt) memset(cnt, 0, 256 * 8);
	if (high) memset(high, 0, 64 * 8);
	for (t = 0; t < T; ++t) {
		keys += w[t].keys;
		if (cnt) for (j = 0; j < 256; ++j) cnt[j] += w[t].cnt[j];
		if (high) for (j = 0; j < 64; ++j) high[j] += w[t].high[j];
	}
	free(w);
	return keys;
}
void bfc_ch_raw_recount(bfc_ch_t *ch) { ch->n_keys = scan_slots(ch, 0, 0); }
void bfc_ch_raw_set_count(bfc_ch_t *ch, uint64_t n_keys) { ch->n_keys = n_keys; } /* the builder already knows it */
/* Union of tables whose key sets are disjoint -- the per-GPU tables of an owner-computes run: every key lives on
 * exactly one rank, so the union IS the reference's table.  Order stamps travel along (first[] per key; sub_last[] = the latest of the
 * ranks'), which keeps bfc_ch_dump byte-identical to `bfc -t1 -d` across GPUs: the stamps are (batch << 32 | rank-major file index),
 * one global order.  A key present in several inputs is merged with saturating counts (cannot happen between ranks). */
bfc_ch_t *bfc_ch_union(const bfc_ch_t *const *tabs, int n)
{
	bfc_ch_t *u;
	uint64_t keys = 0, i;
	int t, cshift = 2, ordered = 1;
	if (n <= 0 || !tabs || !tabs[0]) return 0;
	for (t = 0; t < n; ++t) {
		if (!tabs[t] || tabs[t]->k != tabs[0]->k || tabs[t]->l_pre != tabs[0]->l_pre) return 0;
		keys += tabs[t]->n_keys;
		if (!tabs[t]->first) ordered = 0;
	}
	while (((uint64_t)1 << (tabs[0]->l_pre + cshift)) < keys * 2) ++cshift;
	for (;;) { /* a sub-table that overflows its region sends us round again with twice the room */
		const uint64_t cmask = ((uint64_t)1 << cshift) - 1;
		int full = 0;
		u = bfc_ch_alloc_raw(tabs[0]->k, tabs[0]->l_pre, cshift);
		if (!u) return 0;
		if (ordered) {
			uint64_t *f, *sl;
			if (bfc_ch_raw_order(u, &f, &sl) != 0) { bfc_ch_destroy(u); return 0; }
			memset(f, 0xff, (size_t)8 << (u->l_pre + cshift)); memset(sl, 0, (size_t)8 << u->l_pre);
		}
		for (t = 0; t < n && !full; ++t) {
			const bfc_ch_t *a = tabs[t];
			const uint64_t na = (uint64_t)1 << (a->l_pre + a->cshift);
			for (i = 0; i < na && !full; ++i) {
				const uint64_t v = a->slots[i];
				uint64_t sub, pos, probe, *reg;
				if (!v) continue;
				sub = i >> a->cshift; reg = u->slots + (sub << cshift); pos = (v >> 14) & cmask;
				for (probe = 0; probe <= cmask; ++probe, pos = (pos + 1) & cmask) {
					if (reg[pos] == 0) { reg[pos] = v; ++u->n_keys; if (ordered) u->first[(sub << cshift) + pos] = a->first[i]; break; }
					if ((reg[pos] >> 14) == (v >> 14)) { /* the same key twice: counts add up, saturating (htab.c:74-79) */
						uint64_t c = (reg[pos] & 0xff) + (v & 0xff), h = (reg[pos] >> 8 & 0x3f) + (v >> 8 & 0x3f);
						reg[pos] = (reg[pos] & ~0x3fffULL) | (c < 255 ? c : 255) | ((h < 63 ? h : 63) << 8);
						if (ordered && a->first[i] < u->first[(sub << cshift) + pos]) u->first[(sub << cshift) + pos] = a->first[i];
						break;
					}
				}
				if (probe > cmask) full = 1;
			}
			if (ordered && !full) for (i = 0; i < (uint64_t)1 << a->l_pre; ++i) if (a->sub_last[i] > u->sub_last[i]) u->sub_last[i] = a->sub_last[i];
		}
		if (!full) return u;
		bfc_ch_destroy(u); ++cshift;
	}
}

bfc_ch_t *bfc_ch_init(int k, int l_pre)
{
	assert(k <= 63);
	l_pre = clamp_lpre(k, l_pre);
	assert(k - l_pre < BFC_CH_KEYBITS);
	return bfc_ch_alloc_raw(k, l_pre, 2);
}
void bfc_ch_destroy(bfc_ch_t *ch)
{
	if (!ch) return;
	pthread_rwlock_destroy(&ch->grow_lock);
	free(ch->first); free(ch->sub_last);
	free(ch->slots); free(ch);
}
int bfc_ch_get_k(const bfc_ch_t *ch) { return ch->k; }
int bfc_ch_get_lpre(const bfc_ch_t *ch) { return ch->l_pre; }

static inline uint32_t subkey(const bfc_ch_t *ch, const uint64_t x[2], uint64_t *key) /* htab.c:45-58 */
{
	if (ch->k <= 32) {
		int t = ch->k * 2 - ch->l_pre;
		uint64_t z = x[0] << ch->k | x[1];
		*key = (z & ((1ULL << t) - 1)) << 14 | 1;
		return (uint32_t)(z >> t);
	} else {
		int t = ch->k - ch->l_pre;
		int shift = t + ch->k < BFC_CH_KEYBITS ? ch->k : BFC_CH_KEYBITS - t;
		*key = ((x[0] & ((1ULL << t) - 1)) << shift ^ x[1]) << 14 | 1;
		return (uint32_t)(x[0] >> t);
	}
}

static void grow(bfc_ch_t *ch) /* caller holds the write lock */
{
	drop_order(ch); /* host-side growth does not carry the GPU's stamps */
	int nc = ch->cshift + 1;
	uint64_t n = (uint64_t)1 << (ch->l_pre + ch->cshift), i;
	uint64_t *ns = (uint64_t*)calloc((size_t)1 << (ch->l_pre + nc), 8);
	uint32_t cmask = (1u << nc) - 1;
	if (!ns) { fprintf(stderr, "[E::%s] out of memory growing the count table to %llu slots\n", __func__, 1ULL << (ch->l_pre + nc)); abort(); } /* no error codes on this path */
	for (i = 0; i < n; ++i) {
		uint64_t v = ch->slots[i], *reg;
		uint32_t pos;
		if (!v) continue;
		reg = ns + ((i >> ch->cshift) << nc);
		for (pos = (uint32_t)(v >> 14) & cmask; reg[pos]; pos = (pos + 1) & cmask);
		reg[pos] = v;
	}
	free(ch->slots);
	ch->slots = ns; ch->cshift = nc;
}

/* returns 1 new, 0 updated, -1 region full */
static int upsert(bfc_ch_t *ch, uint32_t sub, uint64_t key, int is_high)
{
	uint32_t cmask = (1u << ch->cshift) - 1, pos = (uint32_t)(key >> 14) & cmask, probe;
	uint64_t *reg = ch->slots + ((uint64_t)sub << ch->cshift);
	uint64_t fresh = key | (uint64_t)(is_high != 0) << 8;
	for (probe = 0; probe <= cmask; ++probe, pos = (pos + 1) & cmask) {
		uint64_t cur = __atomic_load_n(&reg[pos], __ATOMIC_RELAXED);
		if (cur == 0) {
			if (__atomic_compare_exchange_n(&reg[pos], &cur, fresh, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return 1;
		}
		if (cur >> 14 == key >> 14) {
			for (;;) {
				uint64_t nv = cur;
				if ((nv & 0xff) != 0xff) ++nv;                                   /* htab.c:77 */
				if (is_high && (nv >> 8 & 0x3f) != 0x3f) nv += 1 << 8;             /* htab.c:78 */
				if (nv == cur) return 0;
				if (__atomic_compare_exchange_n(&reg[pos], &cur, nv, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return 0;
			}
		}
	}
	return -1;
}

int bfc_ch_insert(bfc_ch_t *ch, const uint64_t x[2], int is_high, int forced)
{
	uint64_t key;
	uint32_t sub = subkey(ch, x, &key);
	(void)forced; /* lock-free upsert never has to give up (htab.c:67-72 returns -1 only on lock contention) */
	if (ch->first) { pthread_rwlock_wrlock(&ch->grow_lock); drop_order(ch); pthread_rwlock_unlock(&ch->grow_lock); }
	for (;;) {
		int r, seen_cshift;
		pthread_rwlock_rdlock(&ch->grow_lock);
		r = upsert(ch, sub, key, is_high);
		seen_cshift = ch->cshift;
		pthread_rwlock_unlock(&ch->grow_lock);
		if (r >= 0) { if (r) __sync_fetch_and_add(&ch->n_keys, 1); return 0; }
		pthread_rwlock_wrlock(&ch->grow_lock);
		if (ch->cshift == seen_cshift) grow(ch); /* else another thread that found the same sub-table full has grown it meanwhile: just retry */
		pthread_rwlock_unlock(&ch->grow_lock);
	}
}

int bfc_ch_get(const bfc_ch_t *ch, const uint64_t x[2])
{
	uint64_t key;
	uint32_t sub = subkey(ch, x, &key);
	uint32_t cmask = (1u << ch->cshift) - 1, pos = (uint32_t)(key >> 14) & cmask, probe;
	const uint64_t *reg = ch->slots + ((uint64_t)sub << ch->cshift);
	for (probe = 0; probe <= cmask; ++probe, pos = (pos + 1) & cmask) {
		uint64_t cur = reg[pos];
		if (cur == 0) return -1;
		if (cur >> 14 == key >> 14) return (int)(cur & 0x3fff);
	}
	return -1;
}
int bfc_ch_kmer_occ(const bfc_ch_t *ch, const bfc_kmer_t *z)
{
	uint64_t y[2];
	kmer_y(ch->k, z->x, y);
	return bfc_ch_get(ch, y);
}
uint64_t bfc_ch_count(const bfc_ch_t *ch) { return ch->n_keys; }

int bfc_ch_hist(const bfc_ch_t *ch, uint64_t cnt[256], uint64_t high[64])
{
	uint64_t i, max = 0;
	int max_i = -1;
	scan_slots(ch, cnt, high);
	for (i = 3; i < 256; ++i) if (cnt[i] > max) { max = cnt[i]; max_i = (int)i; }
	return max_i;
}

static int cmp_u64(const void *a, const void *b) { uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? -1 : x > y; }

uint64_t bfc_ch_export_sorted(const bfc_ch_t *ch, uint32_t *sizes, uint64_t *slots)
{
	uint64_t s, n_sub = (uint64_t)1 << ch->l_pre, n = 0;
	uint32_t c = 1u << ch->cshift, j;
	for (s = 0; s < n_sub; ++s) {
		const uint64_t *reg = ch->slots + (s << ch->cshift);
		uint64_t n0 = n;
		for (j = 0; j < c; ++j) if (reg[j]) { if (slots) slots[n] = reg[j]; ++n; }
		if (sizes) sizes[s] = (uint32_t)(n - n0);
		if (slots && n - n0 > 1) qsort(slots + n0, n - n0, 8, cmp_u64);
	}
	return n;
}

/* ---- khash layout replay (khash.h:219-336 restricted to the no-deletion case) --------------------
 * Inside a sub-table the reference's bucket layout depends only on (i) the order in which distinct keys first arrive and
 * (ii) when growth fires: before EVERY put -- new key or duplicate -- the table doubles if n_occupied >= 0.75 n_buckets
 * (khash.h:298), starting from 4 buckets (khash.h:239); keys sit at (uint32)(key>>14) & mask with triangular probing
 * i += ++step (khash.h:227,315); growth re-places elements in place with a kick-out chain (khash.h:257-283). */
typedef struct { uint32_t nb, size; uint64_t *slot; uint8_t *used; } ksub_t;

static void ksub_grow(ksub_t *t, uint32_t nb_new)
{
	uint32_t nm = nb_new - 1, j;
	uint64_t *ns = (uint64_t*)calloc(nb_new, 8);
	uint8_t *nu = (uint8_t*)calloc(nb_new, 1), *moved = (uint8_t*)calloc(t->nb ? t->nb : 1, 1);
	for (j = 0; j < t->nb; ++j) { /* same order as the in-place rehash: walk old buckets, follow kick-outs */
		uint64_t cur;
		if (!t->used[j] || moved[j]) continue;
		cur = t->slot[j]; moved[j] = 1;
		for (;;) {
			uint32_t i = (uint32_t)(cur >> 14) & nm, step = 0;
			while (nu[i]) i = (i + (++step)) & nm;
			nu[i] = 1;
			if (i < t->nb && t->used[i] && !moved[i]) { uint64_t ev = t->slot[i]; moved[i] = 1; ns[i] = cur; cur = ev; }
			else { ns[i] = cur; break; }
		}
	}
	free(moved); free(t->slot); free(t->used);
	t->slot = ns; t->used = nu; t->nb = nb_new;
}
static void ksub_put_new(ksub_t *t, uint64_t v) /* v's key is known to be absent */
{
	uint32_t mask, i, step = 0;
	if (t->size >= (t->nb >> 2) + (t->nb >> 1)) ksub_grow(t, t->nb ? t->nb << 1 : 4);
	mask = t->nb - 1;
	for (i = (uint32_t)(v >> 14) & mask; t->used[i]; i = (i + (++step)) & mask);
	t->slot[i] = v; t->used[i] = 1; ++t->size;
}

typedef struct { uint64_t stamp, v; } ord_t;
static int cmp_ord(const void *a, const void *b) { uint64_t x = ((const ord_t*)a)->stamp, y = ((const ord_t*)b)->stamp; return x < y ? -1 : x > y; }

/* dump in the reference's format (htab.c:129-149): u32 k, u32 l_pre, then per sub-table u32 n_buckets, u32 size and the
 * occupied slots in bucket order.  With order stamps (GPU build with track_order) every sub-table is replayed through the
 * khash rules above: the file is byte-identical to `bfc -t1 -d` (parity level L2).  Without them n_buckets follows khash's
 * growth rule and the slot order is this implementation's (level L1; the reference's -r restores either). */
int bfc_ch_dump(const bfc_ch_t *ch, const char *fn)
{
	FILE *fp;
	uint32_t t[2], c = 1u << ch->cshift, j;
	uint64_t s, n_sub = (uint64_t)1 << ch->l_pre;
	ord_t *ord = ch->first ? (ord_t*)malloc((size_t)c * sizeof(ord_t)) : 0;
	uint64_t *pack = 0;
	if ((fp = strcmp(fn, "-") ? fopen(fn, "wb") : stdout) == 0) { free(ord); return -1; }
	t[0] = (uint32_t)ch->k; t[1] = (uint32_t)ch->l_pre;
	fwrite(t, 4, 2, fp);
	for (s = 0; s < n_sub; ++s) {
		const uint64_t *reg = ch->slots + (s << ch->cshift);
		uint32_t size = 0, nb = 0;
		for (j = 0; j < c; ++j) size += reg[j] != 0;
		if (ord && size) {
			ksub_t kt = {0, 0, 0, 0};
			uint32_t n = 0;
			uint64_t last_new = 0;
			for (j = 0; j < c; ++j) if (reg[j]) { ord[n].stamp = ch->first[(s << ch->cshift) + j]; ord[n].v = reg[j]; ++n; }
			qsort(ord, n, sizeof(ord_t), cmp_ord);
			for (j = 0; j < n; ++j) ksub_put_new(&kt, ord[j].v);
			last_new = ord[n - 1].stamp;
			/* a call after the last new key still runs the 0.75 check (khash.h:298 is evaluated on every put) */
			if (ch->sub_last[s] > last_new && kt.size >= (kt.nb >> 2) + (kt.nb >> 1)) ksub_grow(&kt, kt.nb << 1);
			t[0] = kt.nb; t[1] = kt.size;
			fwrite(t, 4, 2, fp);
			for (j = 0; j < kt.nb; ++j) if (kt.used[j]) fwrite(&kt.slot[j], 8, 1, fp);
			free(kt.slot); free(kt.used);
			continue;
		}
		if (size) for (nb = 4; size >= (nb >> 2) + (nb >> 1); nb <<= 1);
		t[0] = nb; t[1] = size;
		fwrite(t, 4, 2, fp);
		if (size) { /* one write per sub-table, not one per key */
			uint32_t n = 0;
			if (!pack) pack = (uint64_t*)malloc((size_t)c * 8);
			for (j = 0; j < c; ++j) if (reg[j]) pack[n++] = reg[j];
			fwrite(pack, 8, n, fp);
		}
	}
	free(ord); free(pack);
	fprintf(stderr, "[M::%s] dumpped the hash table to file '%s'.\n", __func__, fn);
	if (fp != stdout) fclose(fp);
	return 0;
}

bfc_ch_t *bfc_ch_restore(const char *fn)
{
	FILE *fp;
	uint32_t t[2];
	uint64_t s, n_sub, left = UINT64_MAX;
	struct stat st;
	bfc_ch_t *ch;
	if ((fp = fopen(fn, "rb")) == 0) return 0;
	if (fread(t, 4, 2, fp) != 2) { fclose(fp); return 0; }
	if (fstat(fileno(fp), &st) == 0 && S_ISREG(st.st_mode)) left = (uint64_t)st.st_size - 8;
	ch = bfc_ch_init((int)t[0], (int)t[1]);
	assert((int)t[1] == ch->l_pre);
	n_sub = (uint64_t)1 << ch->l_pre;
	for (s = 0; s < n_sub; ++s) {
		uint32_t j;
		if (fread(t, 4, 2, fp) != 2) { fclose(fp); bfc_ch_destroy(ch); return 0; }
		if (left != UINT64_MAX) { /* a truncated or damaged file: stop before a bogus size makes the table grow */
			if (left < 8 || (uint64_t)t[1] * 8 > left - 8) { fclose(fp); bfc_ch_destroy(ch); return 0; }
			left -= 8 + (uint64_t)t[1] * 8;
		}
		for (j = 0; j < t[1]; ++j) {
			uint64_t v;
			if (fread(&v, 8, 1, fp) != 1) { fclose(fp); bfc_ch_destroy(ch); return 0; }
			for (;;) { /* place the stored slot value (key + counters) verbatim */
				uint32_t cmask = (1u << ch->cshift) - 1, pos = (uint32_t)(v >> 14) & cmask, probe;
				uint64_t *reg = ch->slots + (s << ch->cshift);
				for (probe = 0; probe <= cmask && reg[pos]; ++probe, pos = (pos + 1) & cmask);
				if (probe <= cmask) { reg[pos] = v; ++ch->n_keys; break; }
				grow(ch);
			}
		}
	}
	fclose(fp);
	fprintf(stderr, "[M::%s] restored the hash table from file '%s'.\n", __func__, fn);
	return ch;
}

/* ---- bit planes of a byte-stream batch (include/bfc_gpu.h: bfcg_count_batch_planes) -- what count.c:72-89 reads of a position ---- */
#include "bfc_planes.h"
uint64_t bfcg_plane_words(uint64_t n_pos) { return (n_pos + 31) / 32 + 2; } /* (two spare words: the device reads a word ahead) */
#if BFC_PLANES_HAVE_AVX2
__attribute__((target("avx2"))) static uint64_t pack_words_avx2(const uint8_t *seq, const uint8_t *qual, uint64_t w, uint64_t hi, bfc_qthr_t t, uint32_t *planes, uint64_t plane_words)
{
	uint32_t m[4];
	for (; ((w + 1) << 5) <= hi; ++w) { /* whole words: 32 positions a step, one store per plane */
		bfc_planes32_avx2(seq + (w << 5), qual ? qual + (w << 5) : 0, t, 32, m);
		planes[w] = m[0]; planes[plane_words + w] = m[1]; planes[2 * plane_words + w] = m[2];
		if (qual) planes[3 * plane_words + w] = m[3];
	}
	return w;
}
#endif
void bfcg_pack_planes(const uint8_t *seq, const uint8_t *qual, uint64_t lo, uint64_t hi, uint64_t n_pos, int q, uint32_t *planes, uint64_t plane_words)
{
	const bfc_qthr_t t = bfc_qthr(q);
	uint64_t w;
	if (hi > n_pos) hi = n_pos;
	w = lo >> 5;
#if BFC_PLANES_HAVE_AVX2
	if (bfc_planes_avx2_ok()) w = pack_words_avx2(seq, qual, w, hi, t, planes, plane_words);
#endif
	for (; (w << 5) < hi; ++w) {
		const uint64_t p0 = w << 5;
		const int n = hi - p0 < 32 ? (int)(hi - p0) : 32;
		uint32_t m0 = 0, m1 = 0, mn = n < 32 && hi == n_pos ? ~0u << n : 0u, mq = 0, m[4]; /* beyond the batch's end: separators */
		int b;
		if (n == 32) {
			for (b = 0; b < 4; ++b) { bfc_planes8(seq + p0 + 8 * b, qual ? qual + p0 + 8 * b : 0, t, m); m0 |= m[0] << (8 * b); m1 |= m[1] << (8 * b); mn |= m[2] << (8 * b); mq |= m[3] << (8 * b); }
		} else {
			for (b = 0; b < n; ++b) { bfc_planes1(seq[p0 + b], qual ? qual + p0 + b : 0, t, m); m0 |= m[0] << b; m1 |= m[1] << b; mn |= m[2] << b; mq |= m[3] << b; }
		}
		planes[w] = m0; planes[plane_words + w] = m1; planes[2 * plane_words + w] = mn;
		if (qual) planes[3 * plane_words + w] = mq;
	}
}
