/* bfc_planes.h -- the four bit planes of stream positions (include/bfc_gpu.h: bfcg_count_batch_planes), shared by bfcg_pack_planes (a byte-stream
 * batch into planes) and the FASTQ fast path of the ingest, whose threads write a batch's planes straight from the mapped file.
 * What count.c:72-89 reads of a position: base code (bseq.c:9-26 minus one, count.c:82), not-a-base (count.c:83,88), qual - 33 >= q (count.c:85). */
#ifndef BFC_PLANES_H
#define BFC_PLANES_H
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

static const uint8_t bfc_plane_code[256] = { /* A C G T (either case) = 0 1 2 3; 4 = not a base */
	4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,
	4,0,4,1,4,4,4,2,4,4,4,4,4,4,4,4, 4,4,4,4,3,4,4,4,4,4,4,4,4,4,4,4, 4,0,4,1,4,4,4,2,4,4,4,4,4,4,4,4, 4,4,4,4,3,4,4,4,4,4,4,4,4,4,4,4,
	4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,
	4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4 };

typedef struct { int T, swar; uint64_t addq; } bfc_qthr_t; /* qual - 33 >= q  <=>  (signed char)qual >= T = q + 33 */
static inline bfc_qthr_t bfc_qthr(int q)
{
	bfc_qthr_t t; t.T = q + 33; t.swar = t.T >= 1 && t.T <= 127; t.addq = (uint64_t)(128 - (t.swar ? t.T : 1)) * 0x0101010101010101ULL;
	return t;
}

/* Eight positions at once, bytes side by side in one register (the device's bases4x / quals4x on 64 bits): after folding case a byte is A C G T iff
 * bit 7 = 0, bit 6 = 1, bit 3 = 0 and (bit 4, bits 2..0) is (0,001) (0,011) (0,111) or (1,100); the code's low bit is bit 1 ^ bit 2, its high bit
 * is bit 2; a multiplication gathers bit 0 of the eight bytes into one byte.  m[0..3]: 8 bits each (code bits of non-bases zero); qual may be NULL
 * (m[3] = 0). */
static inline void bfc_planes8(const uint8_t *seq, const uint8_t *qual, bfc_qthr_t t, uint32_t m[4])
{
	const uint64_t one = 0x0101010101010101ULL, G = 0x0102040810204080ULL;
	uint64_t x, u, s1, s2, s3, s4, s6, s7, t1, t2, ok, bad;
	memcpy(&x, seq, 8);
	u = x & 0xDFDFDFDFDFDFDFDFULL; s1 = u >> 1; s2 = u >> 2; s3 = u >> 3; s4 = u >> 4; s6 = u >> 6; s7 = u >> 7;
	t1 = u & (s1 | ~s2); t2 = s2 & ~s1 & ~u;
	ok = (s4 & t2) | (~s4 & t1);
	bad = (s7 | ~s6 | s3 | ~ok) & one;
	m[2] = (uint32_t)((bad * G) >> 56);
	m[0] = (uint32_t)((((s1 ^ s2) & one) * G) >> 56) & ~m[2];
	m[1] = (uint32_t)(((s2 & one) * G) >> 56) & ~m[2];
	m[3] = 0;
	if (qual) {
		if (t.swar) { /* (b & 0x7f) + (128 - T) carries into bit 7 iff (b & 0x7f) >= T; bytes above 0x7f are negative: never >= T */
			uint64_t y;
			memcpy(&y, qual, 8);
			m[3] = (uint32_t)((((((y & 0x7F7F7F7F7F7F7F7FULL) + t.addq) & ~y) >> 7 & one) * G) >> 56);
		} else { int b; for (b = 0; b < 8; ++b) m[3] |= (uint32_t)((int)(int8_t)qual[b] >= t.T) << b; }
	}
}
/* Thirty-two positions at once (x86 with AVX2, chosen at run time; round 6): a byte compare per letter after folding case, and ONE instruction -- the
 * byte mask -- delivers 32 bits of a plane, where the 64-bit code above spends a multiplication per plane on eight.  The same planes word for word
 * (tests/test_planes.py, tests/test_ingest.py compare them with the serial parser's).  `n` < 32 keeps the low n positions: the load still reads 32
 * bytes, so the caller guarantees them readable (a read's tail reads on into its '+' line; the ingest checks against the end of its window). */
#if defined(__x86_64__) && !defined(BFC_PLANES_NO_AVX2)
#include <immintrin.h>
#define BFC_PLANES_HAVE_AVX2 1
__attribute__((target("avx2"))) static inline void bfc_planes32_avx2(const uint8_t *seq, const uint8_t *qual, bfc_qthr_t t, int n, uint32_t m[4])
{
	const __m256i x = _mm256_loadu_si256((const __m256i*)seq), u = _mm256_and_si256(x, _mm256_set1_epi8((char)0xDF));
	const __m256i a = _mm256_cmpeq_epi8(u, _mm256_set1_epi8('A')), c = _mm256_cmpeq_epi8(u, _mm256_set1_epi8('C'));
	const __m256i g = _mm256_cmpeq_epi8(u, _mm256_set1_epi8('G')), tt = _mm256_cmpeq_epi8(u, _mm256_set1_epi8('T'));
	const uint32_t keep = n >= 32 ? 0xffffffffu : ((1u << n) - 1u);
	const uint32_t lo = (uint32_t)_mm256_movemask_epi8(_mm256_or_si256(c, tt)), hi = (uint32_t)_mm256_movemask_epi8(_mm256_or_si256(g, tt));
	const uint32_t ok = (uint32_t)_mm256_movemask_epi8(a) | lo | hi;
	m[0] = lo & keep; m[1] = hi & keep; m[2] = ~ok & keep; m[3] = 0;
	if (qual) {
		const __m256i y = _mm256_loadu_si256((const __m256i*)qual);
		/* (signed char)qual >= T  <=>  qual > T - 1 as signed bytes; T - 1 < -128: always, T - 1 >= 127: never */
		if (t.T - 1 < -128) m[3] = keep;
		else if (t.T - 1 >= 127) m[3] = 0;
		else m[3] = (uint32_t)_mm256_movemask_epi8(_mm256_cmpgt_epi8(y, _mm256_set1_epi8((char)(t.T - 1)))) & keep;
	}
}
static inline int bfc_planes_avx2_ok(void)
{
	static int ok = -1;
	if (ok < 0) ok = __builtin_cpu_supports("avx2") && !getenv("BFC_INGEST_NO_AVX2");
	return ok;
}
#else
#define BFC_PLANES_HAVE_AVX2 0
#endif

/* one position */
static inline void bfc_planes1(uint8_t s, const uint8_t *qual, bfc_qthr_t t, uint32_t m[4])
{
	const uint32_t c = bfc_plane_code[s];
	m[0] = c & 1u; m[1] = (c >> 1) & 1u; m[2] = c >> 2; m[3] = qual ? (uint32_t)((int)(int8_t)*qual >= t.T) : 0u;
}
#endif
