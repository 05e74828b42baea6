/* bfc_host.c (host side of the reference's bbf.h / htab.h API) under AddressSanitizer/UBSan: random inserts with growth, get, hist,
 * dump -> restore -> dump (bytes equal), union of two tables, restore of truncated / garbage files.
 *   gcc -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude -o build/asan_host scripts/asan/host_main.c bfc_amd/csrc/bfc_host.c -lpthread */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bfc_gpu.h"
#include "bfc_host.h"

static uint64_t rs = 88172645463325252ULL;
static uint64_t rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }

static long fsize(const char *fn, uint8_t **buf)
{
	FILE *f = fopen(fn, "rb"); long n;
	if (!f) return -1;
	fseek(f, 0, SEEK_END); n = ftell(f); fseek(f, 0, SEEK_SET);
	*buf = (uint8_t*)malloc(n + 1);
	if (fread(*buf, 1, n, f) != (size_t)n) n = -1;
	fclose(f);
	return n;
}

int main(int argc, char **argv)
{
	const char *dir = argc > 1 ? argv[1] : "/tmp";
	char f1[512], f2[512], f3[512];
	int ks[] = {15, 21, 31, 32, 33, 37, 51, 63}, ki, bad = 0;
	snprintf(f1, 512, "%s/a.hash", dir); snprintf(f2, 512, "%s/b.hash", dir); snprintf(f3, 512, "%s/c.hash", dir);
	for (ki = 0; ki < 8; ++ki) {
		int k = ks[ki], l_pre = ki & 1 ? 20 : 8, i;
		uint64_t m = k == 64 ? ~0ULL : (1ULL << k) - 1, cnt[256], high[64];
		bfc_ch_t *a = bfc_ch_init(k, l_pre), *b = bfc_ch_init(k, l_pre), *r, *u;
		const bfc_ch_t *two[2];
		uint64_t (*xs)[2] = malloc(40000 * sizeof(*xs));
		for (i = 0; i < 40000; ++i) {
			uint64_t x[2];
			if (i && (rnd() & 3) == 0) { int j = rnd() % i; x[0] = xs[j][0]; x[1] = xs[j][1]; } else { x[0] = rnd() & m; x[1] = rnd() & m; }
			xs[i][0] = x[0]; xs[i][1] = x[1];
			if (bfc_ch_insert((i & 1) ? a : b, x, rnd() & 1, 1) < 0) { fprintf(stderr, "insert failed\n"); ++bad; }
		}
		for (i = 0; i < 40000; ++i) {
			int ga = bfc_ch_get(a, xs[i]), gb = bfc_ch_get(b, xs[i]);
			if (((i & 1) ? ga : gb) < 0) { fprintf(stderr, "k=%d: inserted key not found\n", k); ++bad; break; }
		}
		bfc_ch_hist(a, cnt, high);
		if (bfc_ch_dump(a, f1) != 0) { fprintf(stderr, "dump failed\n"); ++bad; }
		r = bfc_ch_restore(f1);
		if (!r || bfc_ch_count(r) != bfc_ch_count(a) || bfc_ch_dump(r, f2) != 0) { fprintf(stderr, "k=%d: restore/dump failed\n", k); ++bad; }
		else {
			uint8_t *b1 = 0, *b2 = 0; long n1 = fsize(f1, &b1), n2 = fsize(f2, &b2), cut;
			if (n1 != n2 || n1 < 0) { fprintf(stderr, "k=%d: dump -> restore -> dump changes the size\n", k); ++bad; } /* slot order inside a sub-table is free without order stamps */
			for (i = 1; i < 40000; i += 2) if (bfc_ch_get(r, xs[i]) != bfc_ch_get(a, xs[i])) { fprintf(stderr, "k=%d: restored table differs\n", k); ++bad; break; }
			for (cut = 0; cut < 40 && n1 > 0 && bfc_ch_get_lpre(a) <= 8; ++cut) { /* truncated and damaged dumps must be rejected or loaded, never crash */
				long len = cut < 20 ? (long)(rnd() % (uint64_t)n1) : n1;
				FILE *f = fopen(f3, "wb"); bfc_ch_t *t;
				if (cut >= 20 && n1 > 8) { int z; for (z = 0; z < 8; ++z) b1[8 + rnd() % (uint64_t)((n1 < 4096 ? n1 : 4096) - 8)] ^= (uint8_t)rnd(); } /* behind the (k, l_pre) header: a bad header asserts, as htab.c:162 does */
				if (len < 8) len = 8;
				fwrite(b1, 1, len, f); fclose(f);
				t = bfc_ch_restore(f3);
				if (t) { bfc_ch_hist(t, cnt, high); bfc_ch_destroy(t); }
			}
			free(b1); free(b2);
		}
		two[0] = a; two[1] = b;
		u = bfc_ch_union(two, 2);
		if (!u) { fprintf(stderr, "k=%d: union failed\n", k); ++bad; }
		else {
			for (i = 0; i < 40000; i += 7) if (bfc_ch_get(u, xs[i]) < 0) { fprintf(stderr, "k=%d: key missing from the union\n", k); ++bad; break; }
			bfc_ch_dump(u, f2);
			bfc_ch_destroy(u);
		}
		if (r) bfc_ch_destroy(r);
		bfc_ch_destroy(a); bfc_ch_destroy(b); free(xs);
	}
	{ /* the threaded slot scan (tables of 2^24 slots and more) against the single-threaded one on the same inserts */
		bfc_ch_t *big = bfc_ch_alloc_raw(31, 20, 5), *small = bfc_ch_init(31, 20);
		uint64_t c1[256], h1[64], c2[256], h2[64], m = (1ULL << 31) - 1; int i, m1, m2;
		for (i = 0; i < 300000; ++i) {
			uint64_t x[2]; int hi = rnd() & 1;
			x[0] = rnd() & m & ~0xffULL; x[1] = (rnd() & m) >> (rnd() % 24); /* few distinct values: counts of 3 and more appear */
			bfc_ch_insert(big, x, hi, 1); bfc_ch_insert(small, x, hi, 1);
		}
		m1 = bfc_ch_hist(big, c1, h1); m2 = bfc_ch_hist(small, c2, h2);
		if (m1 != m2 || memcmp(c1, c2, sizeof(c1)) || memcmp(h1, h2, sizeof(h1)) || bfc_ch_count(big) != bfc_ch_count(small)) { fprintf(stderr, "threaded histogram differs\n"); ++bad; }
		bfc_ch_destroy(big); bfc_ch_destroy(small);
	}
	{ /* bloom filter host calls */
		bfc_bf_t *bf = bfc_bf_init(20, 4); int i, seen = 0;
		if (bfc_bf_init(8, 4) != 0 || bfc_bf_init(56, 4) != 0) { fprintf(stderr, "bad shifts accepted\n"); ++bad; }
		for (i = 0; i < 100000; ++i) { uint64_t h = rnd(); int c = bfc_bf_insert(bf, h); seen += c == 4; if (bfc_bf_get(bf, h) != 4) { ++bad; break; } }
		bfc_bf_destroy(bf);
	}
	remove(f1); remove(f2); remove(f3);
	printf("done, problems: %d\n", bad);
	return bad != 0;
}
