/* bfc_planes.h -- the four bit planes of stream positions (include/bfc_gpu.h: bfcg_count_batch_planes), shared by bfcg_pack_planes (a byte-stream
 * batch into planes) and the FASTQ fast path of the ingest, whose threads write a batch's planes straight from the mapped file.
 * What count.c:72-89 reads of a position: base code (bseq.c:9-26 minus one, count.c:82), not-a-base (count.c:83,88), qual - 33 >= q (count.c:85). */
#ifndef BFC_PLANES_H
#define BFC_PLANES_H
#include <stdint.h>
#include <string.h>

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
/* one position */
static inline void bfc_planes1(uint8_t s, const uint8_t *qual, bfc_qthr_t t, uint32_t m[4])
{
	const uint32_t c = bfc_plane_code[s];
	m[0] = c & 1u; m[1] = (c >> 1) & 1u; m[2] = c >> 2; m[3] = qual ? (uint32_t)((int)(int8_t)*qual >= t.T) : 0u;
}
#endif
