/* bfcgen.c -- deterministic synthetic read generator.
 *
 * `bfcgen seed G cov L err` : uniform random genome of G bases, reads of length L from uniform
 * positions and random strand, substitution errors at rate err (errors get Q2-14, the rest Q25-40),
 * 1 % of reads get one 'N'.  The random stream is splitmix64, which is counter based: the n-th draw
 * is mix(seed + (n+1)*gamma), so reads can be generated in parallel and in any order while staying
 * byte-identical to the sequential definition (draw order: G genome draws, then per read
 * pos, strand, L per-base draws, one 'N' draw).
 *
 * Library entry points are used by tests/ and bench.py to fill SoA batches directly; the CLI
 * (compiled with -DBFCGEN_MAIN) prints FASTQ.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define GAMMA 0x9E3779B97F4A7C15ULL

static inline uint64_t draw(uint64_t seed, uint64_t n) /* n-th draw, n = 0,1,... */
{
	uint64_t z = seed + (n + 1) * GAMMA;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}

uint64_t bfcgen_n_reads(uint64_t G, double cov, int L) { return (uint64_t)((double)G * cov / L); }

/* genome as codes 0..3 */
void bfcgen_genome(uint64_t seed, uint64_t G, uint8_t *g)
{
	int64_t i;
#pragma omp parallel for schedule(static)
	for (i = 0; i < (int64_t)G; ++i) g[i] = (uint8_t)(draw(seed, (uint64_t)i) >> 62);
}

/* reads r0 <= r < r1 into seq/qual (L bytes per read, no separators, ASCII; qual Phred+33) */
void bfcgen_reads(uint64_t seed, uint64_t G, const uint8_t *g, int L, double err,
                  uint64_t r0, uint64_t r1, uint8_t *seq, uint8_t *qual)
{
	uint64_t thr = (uint64_t)(err * 16777216.0);
	int64_t r;
#pragma omp parallel for schedule(static)
	for (r = (int64_t)r0; r < (int64_t)r1; ++r) {
		uint64_t base = G + (uint64_t)r * (uint64_t)(L + 3);
		uint64_t pos = draw(seed, base) % (G - (uint64_t)L + 1);
		int strand = (int)(draw(seed, base + 1) >> 63), j;
		uint8_t *sq = seq + ((uint64_t)r - r0) * (uint64_t)L, *ql = qual + ((uint64_t)r - r0) * (uint64_t)L;
		uint64_t v;
		for (j = 0; j < L; ++j) {
			int b = strand ? 3 - g[pos + (uint64_t)L - 1 - (uint64_t)j] : g[pos + (uint64_t)j];
			uint64_t u = draw(seed, base + 2 + (uint64_t)j);
			if ((u >> 40) < thr) { b = (b + 1 + (int)(u % 3)) & 3; ql[j] = (uint8_t)(33 + 2 + (int)((u >> 8) % 13)); }
			else ql[j] = (uint8_t)(33 + 25 + (int)((u >> 8) % 16));
			sq[j] = (uint8_t)"ACGT"[b];
		}
		v = draw(seed, base + 2 + (uint64_t)L);
		if (v % 100 == 0) sq[(v >> 32) % (uint64_t)L] = 'N';
	}
}

/* write reads r0..r1 as FASTQ (names @r<r>) */
int bfcgen_fastq(uint64_t seed, uint64_t G, const uint8_t *g, int L, double err, uint64_t r0, uint64_t r1, const char *fn)
{
	FILE *fp = (fn && strcmp(fn, "-")) ? fopen(fn, "wb") : stdout;
	uint64_t r, chunk = 65536;
	uint8_t *sq, *ql;
	if (!fp) return -1;
	sq = (uint8_t*)malloc(chunk * (uint64_t)L); ql = (uint8_t*)malloc(chunk * (uint64_t)L);
	for (r = r0; r < r1; r += chunk) {
		uint64_t e = r + chunk < r1 ? r + chunk : r1, i;
		bfcgen_reads(seed, G, g, L, err, r, e, sq, ql);
		for (i = 0; i < e - r; ++i) {
			fprintf(fp, "@r%llu\n", (unsigned long long)(r + i));
			fwrite(sq + i * (uint64_t)L, 1, (size_t)L, fp); fputs("\n+\n", fp);
			fwrite(ql + i * (uint64_t)L, 1, (size_t)L, fp); fputc('\n', fp);
		}
	}
	free(sq); free(ql);
	if (fp != stdout) fclose(fp);
	return 0;
}

/* ---- helpers of the benchmark / test tooling (no part of the counting path) ---- */

/* number of bfc_kmer_insert calls on n reads of length L (count.c:83-88: one per position with >= k ACGT in a row) */
uint64_t bfcgen_count_kmers(const uint8_t *seq, uint64_t n, int L, int k)
{
	uint64_t total = 0;
	int64_t r;
#pragma omp parallel for schedule(static) reduction(+:total)
	for (r = 0; r < (int64_t)n; ++r) {
		const uint8_t *s = seq + (uint64_t)r * (uint64_t)L;
		int j, run = 0;
		for (j = 0; j < L; ++j) {
			const uint8_t c = s[j] & 0xDF;
			run = (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? run + 1 : 0;
			total += run >= k;
		}
	}
	return total;
}

/* n reads of length L -> the separator-delimited stream form of include/bfc_gpu.h PART 2: read, then one byte `sep` */
void bfcgen_to_stream(const uint8_t *src, uint64_t n, int L, uint8_t sep, uint8_t *dst)
{
	int64_t r;
#pragma omp parallel for schedule(static)
	for (r = 0; r < (int64_t)n; ++r) {
		memcpy(dst + (uint64_t)r * (uint64_t)(L + 1), src + (uint64_t)r * (uint64_t)L, (size_t)L);
		dst[(uint64_t)r * (uint64_t)(L + 1) + (uint64_t)L] = sep;
	}
}

/* checksums of a bitmap as SURVEY App. B.3 defines them: popcount, and FNV-1a/64 over all bytes in order */
uint64_t bfcgen_popcount(const uint8_t *p, uint64_t n)
{
	uint64_t total = 0;
	int64_t i, nw = (int64_t)(n / 8);
	const uint64_t *w = (const uint64_t*)p;
#pragma omp parallel for schedule(static) reduction(+:total)
	for (i = 0; i < nw; ++i) total += (uint64_t)__builtin_popcountll(w[i]);
	for (i = nw * 8; i < (int64_t)n; ++i) total += (uint64_t)__builtin_popcount(p[i]);
	return total;
}
uint64_t bfcgen_fnv1a64(const uint8_t *p, uint64_t n)
{
	uint64_t h = 0xcbf29ce484222325ULL, i;
	for (i = 0; i < n; ++i) h = (h ^ p[i]) * 0x100000001b3ULL;
	return h;
}

/* the same, continued from h (a stream hashed piece by piece; start with 0xcbf29ce484222325) */
uint64_t bfcgen_fnv1a64_from(uint64_t h, const uint8_t *p, uint64_t n)
{
	uint64_t i;
	for (i = 0; i < n; ++i) h = (h ^ p[i]) * 0x100000001b3ULL;
	return h;
}

/* A digest of a bitmap that many threads can take (FNV-1a is a serial chain: 17 s for a 16 GiB filter, seven times in the GPU suite): the sum over the
 * 64-bit little-endian words w_i of w_i * (i * 0x9E3779B97F4A7C15 | 1), modulo 2^64.  n must be a multiple of 8. */
uint64_t bfcgen_mix64(const uint8_t *p, uint64_t n)
{
	const uint64_t nw = n / 8;
	uint64_t sum = 0;
	int64_t i;
#pragma omp parallel for reduction(+:sum) schedule(static)
	for (i = 0; i < (int64_t)nw; ++i) {
		uint64_t w;
		memcpy(&w, p + 8 * (uint64_t)i, 8);
		sum += w * (((uint64_t)i * 0x9E3779B97F4A7C15ULL) | 1ULL);
	}
	return sum;
}

#ifdef BFCGEN_MAIN
int main(int argc, char **argv)
{
	uint64_t seed, G, n; double cov, err; int L; uint8_t *g;
	if (argc < 6) { fprintf(stderr, "Usage: bfcgen <seed> <G> <cov> <L> <err> [out.fq]\n"); return 1; }
	seed = strtoull(argv[1], 0, 10); G = strtoull(argv[2], 0, 10); cov = atof(argv[3]); L = atoi(argv[4]); err = atof(argv[5]);
	g = (uint8_t*)malloc(G);
	bfcgen_genome(seed, G, g);
	n = bfcgen_n_reads(G, cov, L);
	bfcgen_fastq(seed, G, g, L, err, 0, n, argc > 6 ? argv[6] : "-");
	free(g);
	return 0;
}
#endif
