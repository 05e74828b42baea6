// bfcg_k1.h -- the pieces of K1 (bases -> bit planes -> k-mer hash -> 12-byte record) that the level-1 scatter kernels of two translation
// units share: bfcg_kernels.hip (k_hist1, k_scatter1, k_query, ...) and bfcg_scatter1wc.hip (k_scatter1_wc).  Inline device code and plain structs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kmer_dev.h"
#include "bfcg_internal.h"
#include "bfcg_dev.h"

using namespace bfcg;

#ifndef WAVE
#define WAVE 64
#endif

// Workgroups are dealt to the 8 XCDs round-robin (block b -> XCD b % 8, MI355X_MICROARCH.md).  Neighbouring tiles
// write neighbouring runs of the same bucket, sharing a cache line at the seam; mapping blocks so that each
// XCD walks a CONTIGUOUS range of tiles lets one L2 merge both halves of those lines.  Speed only.
__device__ __forceinline__ int64_t xcd_tile(int64_t bid, int64_t n_tiles)
{
	const int64_t per = (n_tiles + 7) / 8;
	if ((bid >> 3) >= per) return n_tiles; // surplus block of an over-sized grid: nothing to do
	return (bid & 7) * per + (bid >> 3);
}

// planes: [0] low base bit, [1] high base bit, [2] not-ACGT, [3] quality >= q
// Covers positions [t0-64, t0+TILE); PLANE_WORDS = (TILE+64)/32 + 2 spare words per plane.
// Fast path (16-byte aligned streams): every lane loads 16 bases + 16 qualities with one
// dwordx4 each and writes four 16-bit plane pieces -- one load round per tile instead of a
// latency-bound byte loop.  A=0 C=1 G=2 T=3 (bseq.c:9-26 minus one, count.c:82): with
// u = ch & 0xDF, x = (u>>1)&3 gives A0 C1 G3 T2 and x^(x>>1) the code.
__device__ __forceinline__ void bases16(uint32_t w, int sh, uint32_t &m0, uint32_t &m1, uint32_t &mn)
{
#pragma unroll
	for (int b = 0; b < 4; ++b) {
		uint32_t u = (w >> (8 * b)) & 0xDFu;
		uint32_t x = (u >> 1) & 3u, code = x ^ (x >> 1);
		bool ok = (u == 'A') | (u == 'C') | (u == 'G') | (u == 'T');
		m0 |= (code & 1u) << (sh + b); m1 |= (code >> 1) << (sh + b); mn |= (ok ? 0u : 1u) << (sh + b);
	}
}
__device__ __forceinline__ void quals16(uint32_t w, int sh, int q, uint32_t &mq)
{
#pragma unroll
	// count.c:85 compares a (signed) char: bytes above 0x7f are negative there, never high quality for a sane -q
	for (int b = 0; b < 4; ++b) mq |= (uint32_t)((int)(int8_t)((w >> (8 * b)) & 0xffu) - 33 >= q) << (sh + b);
}

// The same for four bases at a time, bytes side by side in one register (no extraction, no compares): after folding case (& 0xDF) a byte
// is A C G T iff bit 7 = 0, bit 6 = 1, bit 3 = 0 and (bit 4, bits 2..0) is one of (0,001) (0,011) (0,111) (1,100); the code's low bit is
// bit 1 ^ bit 2, its high bit is bit 2.  Every quantity is computed in bit 0 of each byte; a multiplication gathers the four bits into a nibble.
__device__ __forceinline__ uint32_t gather4(uint32_t m) { return (m * 0x01020408u) >> 24; } // m has bits 0, 8, 16, 24 only -> bits 0..3
__device__ __forceinline__ void bases4x(uint32_t w, int sh, uint32_t &m0, uint32_t &m1, uint32_t &mn)
{
	const uint32_t u = w & 0xDFDFDFDFu, s1 = u >> 1, s2 = u >> 2, s3 = u >> 3, s4 = u >> 4, s6 = u >> 6, s7 = u >> 7, one = 0x01010101u;
	const uint32_t t1 = u & (s1 | ~s2);          // bits 2..0 in {001, 011, 111}
	const uint32_t t2 = s2 & ~s1 & ~u;           // bits 2..0 = 100
	const uint32_t ok = (s4 & t2) | (~s4 & t1);
	const uint32_t bad = (s7 | ~s6 | s3 | ~ok) & one;
	m0 |= gather4((s1 ^ s2) & one) << sh; m1 |= gather4(s2 & one) << sh; mn |= gather4(bad) << sh;
}
// count.c:85 on four signed chars at once, for a threshold T = q + 33 in 1..127 (bytes above 0x7f are negative: never high quality):
// (b & 0x7f) + (128 - T) carries into bit 7 iff (b & 0x7f) >= T
__device__ __forceinline__ void quals4x(uint32_t w, int sh, uint32_t add, uint32_t &mq)
{ mq |= gather4(((((w & 0x7F7F7F7Fu) + add) & ~w) >> 7) & 0x01010101u) << sh; }

// k-mer ending at tile-relative position r (0 <= r < TILE), 32 < k < 64, on 32-bit halves (kmer_dev.h); KC > 0: k at compile time.
// Returns false if there is none.
template <int TILE, int KC>
__device__ __forceinline__ bool kmer_at2(const uint32_t *planes, int r, int k_, U2 &y0, U2 &y1, bool &is_high)
{
	constexpr int PW = (TILE + 64) / 32 + 2;
	const int k = KC ? KC : k_, bit = r + 65 - k, wi = bit >> 5, s = bit & 31;
	const uint32_t mh = (1u << (k - 32)) - 1u;
	const uint32_t *p = planes + wi;
	{
		const uint32_t a = p[2 * PW], b = p[2 * PW + 1], c = p[2 * PW + 2];
		if ((__builtin_amdgcn_alignbit(b, a, s) | (__builtin_amdgcn_alignbit(c, b, s) & mh)) != 0) return false; // a base that is not ACGT in the window (count.c:83,86-87)
	}
	const uint32_t l0 = p[0], l1 = p[1], l2 = p[2], h0 = p[PW], h1 = p[PW + 1], h2 = p[PW + 2], q0 = p[3 * PW], q1 = p[3 * PW + 1], q2 = p[3 * PW + 2];
	is_high = (__builtin_amdgcn_alignbit(q1, q0, s) & (__builtin_amdgcn_alignbit(q2, q1, s) | ~mh)) == 0xffffffffu; // count.c:85-86
	kmer_hash_from_windows2<KC>(k, __builtin_amdgcn_alignbit(l1, l0, s), __builtin_amdgcn_alignbit(l2, l1, s),
	                            __builtin_amdgcn_alignbit(h1, h0, s), __builtin_amdgcn_alignbit(h2, h1, s), y0, y1);
	return true;
}

// k-mer ending at tile-relative position r (0 <= r < TILE).  Returns false if there is none.
template <typename W, int TILE>
__device__ __forceinline__ bool kmer_at(const uint32_t *planes, int r, int k, W m, W &y0, W &y1, bool &is_high)
{
	if constexpr (sizeof(W) == 8) {
		if (k > 32) { // (W = 64 bits serves k > 32 only; the generic code below remains for completeness)
			U2 a, b;
			if (!kmer_at2<TILE, 0>(planes, r, k, a, b, is_high)) return false;
			y0 = u2_join(a); y1 = u2_join(b);
			return true;
		}
	}
	constexpr int PW = (TILE + 64) / 32 + 2;
	int bit = r + 65 - k;
	if (window<W>(planes + 2 * PW, bit, m) != 0) return false;
	W wl = window<W>(planes, bit, m), wh = window<W>(planes + PW, bit, m);
	is_high = window<W>(planes + 3 * PW, bit, m) == m;
	kmer_hash_from_windows<W>(k, wl, wh, m, y0, y1);
	return true;
}

// ---- records

// A k-mer record is RD dwords: y0, y1 (the two words of bfc_kmer_hash, kmer.h:79-88), the high-quality flag and the file-order index
// (position in the batch, 32 bit).  After the level-1 scatter a record sits in the bucket its bloom block id selects, and for k >= bf_shift-9
// that id is a bit field of y0 (kmer_dev.h: the low bf_shift-9 bits of the hash are y0's): bits [rec_lo, rec_lo + rec_n) of y0 ARE the
// level-1 bucket.  They are not stored (RecGeom): config c3's records (k=33) take 12 instead of 16 bytes, c5's (k=51) 16 instead of 20.
//   RD=3: u64 A = y0' | y1 << a | hi << (a+k)   (a = k - rec_n kept bits of y0; a + k + 1 <= 64), u32 index          -- 12 bytes
//   RD=4: one 128-bit word  y0' | y1 << a | hi << (a+k) | index << (a+k+1)                       (a + k + 33 <= 128)  -- 16 bytes
//   RD=5: u64 y0 | is_high<<63, u64 y1, u32 index (nothing dropped)                                                    -- 20 bytes (4-byte aligned)
struct RecGeom { int k, a, lo, n; };
__device__ __forceinline__ RecGeom rec_geom(const KParams &P) { RecGeom g; g.k = P.k; g.n = P.rec_n; g.a = P.k - P.rec_n; g.lo = P.rec_lo; return g; }
__device__ __forceinline__ uint64_t y0_drop(const RecGeom g, uint64_t y0) { return g.n ? (y0 & ((1ULL << g.lo) - 1)) | ((y0 >> (g.lo + g.n)) << g.lo) : y0; }
__device__ __forceinline__ uint64_t y0_join(const RecGeom g, uint64_t y0c, uint32_t imp)
{ return g.n ? (y0c & ((1ULL << g.lo) - 1)) | ((uint64_t)imp << g.lo) | ((y0c >> g.lo) << (g.lo + g.n)) : y0c; }

// A DEAD record (all ones) fills what a level-1 workgroup left unused of its last chunk of a slab (k_scatter1, OnePass): level 2 skips it.  For
// 12- and 20-byte records the last dword is the file index, which is never 2^32 - 1 (a batch has fewer positions); a 16-byte record's last
// dword mixes index, y1 and the quality flag, so there every dword is tested (a live record with y0' and y1 all ones AND the last index does not exist).
template <int RD> __device__ __forceinline__ bool rec_dead(const RecW<RD> &w)
{
	if (RD == 4) return (w.d[0] & w.d[1] & w.d[2] & w.d[3]) == 0xffffffffu;
	return w.d[RD - 1] == 0xffffffffu;
}

// pack: y0 is the FULL word (the bucket's bits are dropped here); unpack: imp = the record's (global) level-1 bucket
template <int RD> struct Rec;
template <> struct Rec<3> {
	static __device__ __forceinline__ void pack(RecW<3> &r, const RecGeom g, uint64_t y0, uint64_t y1, uint32_t idx, bool hi)
	{
		const uint64_t A = y0_drop(g, y0) | (y1 << g.a) | ((uint64_t)hi << (g.a + g.k));
		r.d[0] = (uint32_t)A; r.d[1] = (uint32_t)(A >> 32); r.d[2] = idx;
	}
	static __device__ __forceinline__ void unpack(const RecW<3> &r, const RecGeom g, uint32_t imp, uint64_t &y0, uint64_t &y1, uint32_t &idx, bool &hi)
	{
		const uint64_t A = r.d[0] | ((uint64_t)r.d[1] << 32);
		y0 = y0_join(g, A & ((1ULL << g.a) - 1), imp);
		y1 = (A >> g.a) & ((1ULL << g.k) - 1);
		hi = (A >> (g.a + g.k)) & 1; idx = r.d[2];
	}
};
template <> struct Rec<4> {
	// On two 64-bit halves (round 5; the 128-bit arithmetic it replaces compiled into chains of selects around variable funnel shifts): records are
	// 16 bytes iff 96 < a + k + 33 <= 128, i.e. 64 <= a + k <= 95 with 1 <= a <= 63 -- y1 straddles the halves, the quality flag (bit a + k) and
	// the index (from bit a + k + 1) lie in the upper one.
	static __device__ __forceinline__ void pack(RecW<4> &r, const RecGeom g, uint64_t y0, uint64_t y1, uint32_t idx, bool hi)
	{
		const uint64_t lo = y0_drop(g, y0) | (y1 << g.a);
		const uint64_t hi64 = (y1 >> (64 - g.a)) | ((uint64_t)hi << (g.a + g.k - 64)) | ((uint64_t)idx << (g.a + g.k - 63));
		r.d[0] = (uint32_t)lo; r.d[1] = (uint32_t)(lo >> 32); r.d[2] = (uint32_t)hi64; r.d[3] = (uint32_t)(hi64 >> 32);
	}
	static __device__ __forceinline__ void unpack(const RecW<4> &r, const RecGeom g, uint32_t imp, uint64_t &y0, uint64_t &y1, uint32_t &idx, bool &hi)
	{
		const uint64_t lo = r.d[0] | ((uint64_t)r.d[1] << 32), hi64 = r.d[2] | ((uint64_t)r.d[3] << 32);
		y0 = y0_join(g, lo & ((1ULL << g.a) - 1), imp);
		y1 = ((lo >> g.a) | (hi64 << (64 - g.a))) & ((1ULL << g.k) - 1);
		hi = (hi64 >> (g.a + g.k - 64)) & 1;
		idx = (uint32_t)(hi64 >> (g.a + g.k - 63));
	}
};
template <> struct Rec<5> {
	static __device__ __forceinline__ void pack(RecW<5> &r, const RecGeom, uint64_t y0, uint64_t y1, uint32_t idx, bool hi)
	{
		const uint64_t a = y0 | ((uint64_t)hi << 63);
		r.d[0] = (uint32_t)a; r.d[1] = (uint32_t)(a >> 32); r.d[2] = (uint32_t)y1; r.d[3] = (uint32_t)(y1 >> 32); r.d[4] = idx;
	}
	static __device__ __forceinline__ void unpack(const RecW<5> &r, const RecGeom, uint32_t, uint64_t &y0, uint64_t &y1, uint32_t &idx, bool &hi)
	{
		const uint64_t a = r.d[0] | ((uint64_t)r.d[1] << 32);
		y0 = a & ~(1ULL << 63); hi = a >> 63; y1 = r.d[2] | ((uint64_t)r.d[3] << 32); idx = r.d[4];
	}
};

// ONEPASS: no histogram pass at all (K1 runs ONCE per batch).  The output is not one contiguous run per bucket but 8 SLABS per bucket, one per
// XCD (workgroups are dealt to the XCDs round-robin: blockIdx & 7), each of `cap` records: a tile reserves room for its bucket runs with one
// returning atomicAdd per bucket on the slab's cursor -- 8 x 2^F1 cursors on cache lines of their own, so that the chains of same-address
// atomics (~12 ns each) are 8 x 2^F1 wide -- and level 2 reads a bucket as its 8 segments (the machinery multi-GPU runs use for the sources'
// blocks).  Uniform hashing fills a slab to its mean +- a fraction of a per cent; a batch of few, often repeated k-mers overflows one:
// the kernel then raises `flags[0]` of ITS batch slot, stage B of that batch changes nothing and seals the run (k_seal: the sticky word,
// which only stage B's stream ever touches, turns every batch behind it into a no-op as well) and the host replays those batches through the
// two-pass partition (bfcg_ctx.hip: replay_poisoned).  The flags are per slot because stage A of batch t+1 runs beside stage B of batch t:
// an overflow of t+1 must not be seen by the kernels of the clean batch t.
// Round 3: room is reserved in CHUNKS.  Device-scope atomics on this chip are executed at the memory side, not in the XCD's L2 (TCC_EA0_ATOMIC
// = TCC_ATOMIC), and they share that path with the records' stores: with 2^10 buckets -- runs of ~3 records, one atomic each -- config c4's
// level 1 ran at 20 ps per position against 5 without the stores or without the atomics.  A workgroup therefore keeps, per bucket, what is left
// of the chunk it reserved last (thread-private: a thread owns its buckets for the kernel's life); a run goes there first and the rest into
// a new chunk of max(chunk, rest) records, so a run is at most two pieces and a slab has no holes except the workgroups' last chunks, which
// are filled with DEAD records (all ones: no file index is 2^32-1) that level 2 skips.  chunk <= 1: every run reserves exactly its size.
// own_n > 0 (a rank of a multi-GPU group): the slabs of buckets [own_lo, own_lo + own_n) -- the rank's OWN share of the exchange -- lie own_delta
// records further on: in the rank's receive buffer, which the group allocates behind the send buffer, at the place its block has there.
struct OnePass { uint32_t *cursor; uint32_t cap; uint32_t *flags; unsigned long long *stats; uint32_t chunk; uint32_t own_lo, own_n, own_delta; };

// A 12-byte record from halves without 64-bit shifts (same bits as Rec<3>::pack): possible when the kept part of y0 fits one word
// (0 < a = k - rec_n < 32), the dropped field ends below bit 32, and y1 reaches into the second word (a + k >= 32)
struct Pack3 { int ok, a, lo, n, sh_flag; uint32_t lowmask; };
__device__ __forceinline__ Pack3 pack3_geom(const KParams &P)
{
	Pack3 g; g.a = P.k - P.rec_n; g.lo = P.rec_lo; g.n = P.rec_n; g.sh_flag = g.a + P.k - 32; g.lowmask = P.rec_n ? (1u << (P.rec_lo & 31)) - 1u : 0xffffffffu;
	g.ok = g.a >= 1 && g.a <= 31 && g.a + P.k >= 32 && (P.rec_n == 0 || P.rec_lo + P.rec_n <= 31);
	return g;
}
__device__ __forceinline__ void pack3_fast(RecW<3> &r, const Pack3 g, const U2 y0, const U2 y1, uint32_t idx, bool hi)
{
	const uint32_t y0d = g.n ? (y0.lo & g.lowmask) | (__builtin_amdgcn_alignbit(y0.hi, y0.lo, g.lo + g.n) << g.lo) : y0.lo;
	r.d[0] = y0d | (y1.lo << g.a);
	r.d[1] = __builtin_amdgcn_alignbit(y1.hi, y1.lo, 32 - g.a) | ((uint32_t)hi << g.sh_flag);
	r.d[2] = idx;
}

namespace bfcg {
// bfcg_scatter1wc.hip: level 1 through write-combining buffers in LDS (k_scatter1_wc) -- whether this one-pass stage A can take it and how
// (threads per workgroup, chunks per reservation, persistent workgroups), and its launch
struct WcPlan { int rw, bt, spt; uint32_t G; unsigned grid; }; // record dwords, threads, positions per thread and round, chunks per reservation, workgroups
bool scatter1_wc_plan(const KParams &P, const OnePass &OP, int rw, int64_t n_pos, WcPlan *pl);
void run_scatter1_wc(const KParams &P, const uint8_t *seq, const uint8_t *qual, int64_t n_pos, uint32_t *out, const OnePass &OP, const WcPlan &pl, hipStream_t st);
}
