/* The host ingest (bfc_ingest.h: serial kseq-grammar parser + threaded fast path) under AddressSanitizer/UBSan, without a GPU.
 *   gcc -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -Ibfc_amd/csrc -Iinclude -o build/asan_ingest scripts/asan/ingest_main.c -lz -lpthread
 *   build/asan_ingest file chunk cap threads   -> prints the digest fields of bfc_ingest_digest (same loop as bfc_count.c) */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
void *bfcg_host_alloc(uint64_t bytes) { return malloc(bytes); }
void bfcg_host_free(void *p) { free(p); }
#include "bfc_ingest.h"

int main(int argc, char **argv)
{
	ingest_t in;
	batch_t b;
	uint64_t hs = 0xcbf29ce484222325ULL, hq = hs, hb = hs, i, out[3] = {0, 0, 0}, cap;
	if (argc < 5) return 2;
	if (argc > 5) { /* the trim pass's use of the serial parser (bfc_trim.c): names and comments kept */
		parser_t ps; uint64_t h = 0xcbf29ce484222325ULL, n = 0; size_t j; int rc, calls = 0;
		memset(&ps, 0, sizeof(ps)); ps.keep_hdr = 1; ps.chunk_size = 1 << 30;
		ps.rd.fp = gzopen(argv[1], "r"); if (!ps.rd.fp) return 3;
		ps.rd.buf = (uint8_t*)malloc(RD_BUF);
		while (calls < 4) {
			rc = next_record(&ps);
			if (rc <= 0) { ++calls; continue; }
			++n;
			for (j = 0; j < ps.l_hdr; ++j) h = (h ^ ps.hdr[j]) * 0x100000001b3ULL;
			if (ps.have_cmt) for (j = 0; j < ps.l_cmt; ++j) h = (h ^ ps.cmt[j]) * 0x100000001b3ULL;
			for (j = 0; j < ps.l_seq; ++j) h = (h ^ ps.seq[j]) * 0x100000001b3ULL;
			if (ps.rec_has_qual) for (j = 0; j < ps.l_seq; ++j) h = (h ^ ps.qual[j]) * 0x100000001b3ULL;
		}
		printf("%llu %llx\n", (unsigned long long)n, (unsigned long long)h);
		gzclose(ps.rd.fp); free(ps.rd.buf); free(ps.rd.line); free(ps.seq); free(ps.qual); free(ps.hdr); free(ps.cmt);
		return 0;
	}
	cap = strtoull(argv[3], 0, 10);
	if (ingest_open(&in, argv[1], strtoull(argv[2], 0, 10), atoi(argv[4]), 1) != 0) return 3;
	memset(&b, 0, sizeof(b));
	b.cap = cap; b.seq = (uint8_t*)malloc(cap); b.qual = (uint8_t*)malloc(cap);
	if (getenv("ASAN_INGEST_PLANES")) { /* round 6: the batch as bit planes -- the fast path packs them straight from the text (AVX2, 32 positions a step, masked tails) */
		const uint64_t pw = (cap + 31) / 32 + 2;
		b.planes = (uint32_t*)malloc(pw * 16); b.plane_words = pw; b.q = 20;
		memset(b.planes, 0xa5, pw * 16);
	}
	for (;;) {
		ingest_fill(&in, &b);
		if (b.n_seqs && b.planes) { /* digest of the planes: packed directly, or (serial batches) from the byte streams by the same per-position rule */
			const uint64_t nw = (b.n_pos + 31) / 32, pw = b.plane_words;
			uint64_t w; int pl;
			++out[0]; out[1] += (uint64_t)b.n_seqs; out[2] += b.n_pos;
			if (!b.packed) {
				const bfc_qthr_t t = bfc_qthr(b.q);
				for (w = 0; w < nw; ++w) { int k; uint32_t m[4], a[4] = {0, 0, 0, 0}; for (k = 0; k < 32 && w * 32 + k < b.n_pos; ++k) { bfc_planes1(b.seq[w * 32 + k], b.has_qual ? b.qual + w * 32 + k : 0, t, m); a[0] |= m[0] << k; a[1] |= m[1] << k; a[2] |= m[2] << k; a[3] |= m[3] << k; }
					if (b.n_pos - w * 32 < 32) a[2] |= ~0u << (b.n_pos - w * 32);
					for (pl = 0; pl < 4; ++pl) b.planes[(uint64_t)pl * pw + w] = a[pl]; }
			}
			for (pl = 0; pl < (b.has_qual ? 4 : 3); ++pl) for (w = 0; w < nw; ++w) { const uint32_t v = b.planes[(uint64_t)pl * pw + w]; int k; for (k = 0; k < 4; ++k) hs = (hs ^ ((v >> (8 * k)) & 0xff)) * 0x100000001b3ULL; }
		} else
		if (b.n_seqs) {
			++out[0]; out[1] += (uint64_t)b.n_seqs; out[2] += b.n_pos;
			for (i = 0; i < b.n_pos; ++i) { hs = (hs ^ b.seq[i]) * 0x100000001b3ULL; hq = (hq ^ b.qual[i]) * 0x100000001b3ULL; }
			for (i = 0; i < 4; ++i) hb = (hb ^ (((uint64_t)b.n_seqs >> (8 * i)) & 0xff)) * 0x100000001b3ULL;
		}
		if (b.last) break;
	}
	printf("%llu %llu %llu %llx %llx %llx\n", (unsigned long long)out[0], (unsigned long long)out[1], (unsigned long long)out[2], (unsigned long long)hs, (unsigned long long)hq, (unsigned long long)hb);
	free(b.seq); free(b.qual); free(b.kind_cut); free(b.planes);
	ingest_close(&in);
	return 0;
}
