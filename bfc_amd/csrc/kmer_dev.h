// kmer_dev.h -- device-side k-mer math for gfx950 (wave64).
//
// The reference rolls four k-bit planes per read, one base at a time (kmer.h:10-17,
// count.c:81-88).  On the GPU every lane owns ONE end position of the batch's byte stream
// instead: a workgroup turns its tile of ASCII bases into four packed bit-planes in LDS with
// wave ballots (bit i of a word = position i, LSB first), and a lane obtains the planes of the
// k-mer ending at position e as the k-bit WINDOW [e-k+1, e] of those bit streams:
//     x2 = ~W(low-bit plane)  & m      (kmer.h:15: oldest base at bit 0, complemented)
//     x3 = ~W(high-bit plane) & m      (kmer.h:16)
//     x0 = bitreverse_k(W(low))        (kmer.h:13: newest base at bit 0)
//     x1 = bitreverse_k(W(high))       (kmer.h:14)
// A k-mer exists at e iff the window of the "not ACGT" plane is zero (count.c:83,86-87: l >= k),
// and it is high quality iff the window of the "qual-33 >= q" plane is all ones (count.c:85-86).
// No rolling state, no warm-up, perfectly coalesced lane->position mapping.
//
// W = uint32_t serves k <= 32 (all arithmetic of bfc_hash_64 is mod 2^k, so 32-bit registers
// are exact), W = uint64_t serves k <= 63.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bfcg {

template <typename W> __device__ __forceinline__ W kmask(int k)
{
	return k >= (int)(8 * sizeof(W)) ? ~W(0) : (W(1) << k) - 1;
}

// kmer.h:30-40 -- Thomas Wang mix, every sum reduced to k bits
template <typename W> __device__ __forceinline__ W mix_k(W v, W m)
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

__device__ __forceinline__ uint32_t brev_w(uint32_t v) { return __brev(v); }
__device__ __forceinline__ uint64_t brev_w(uint64_t v) { return __brevll(v); }

// k-bit window starting at bit `bit` of a packed LSB-first plane (plane must have 2 spare words)
template <typename W> __device__ __forceinline__ W window(const uint32_t *plane, int bit, W m);
template <> __device__ __forceinline__ uint32_t window<uint32_t>(const uint32_t *plane, int bit, uint32_t m)
{
	int wi = bit >> 5, s = bit & 31;
	return __builtin_amdgcn_alignbit(plane[wi + 1], plane[wi], s) & m;
}
template <> __device__ __forceinline__ uint64_t window<uint64_t>(const uint32_t *plane, int bit, uint64_t m)
{
	int wi = bit >> 5, s = bit & 31;
	uint32_t a = plane[wi], b = plane[wi + 1], c = plane[wi + 2];
	uint32_t lo = __builtin_amdgcn_alignbit(b, a, s), hi = __builtin_amdgcn_alignbit(c, b, s);
	return (((uint64_t)hi << 32) | lo) & m;
}

// kmer.h:79-88 from the two forward windows.  Outputs y0=(h0+h1)&m, y1=h1.
template <typename W> __device__ __forceinline__ void kmer_hash_from_windows(int k, W w_lo, W w_hi, W m, W &y0, W &y1)
{
	const int nb = 8 * (int)sizeof(W);
	W x0 = brev_w(w_lo) >> (nb - k), x1 = brev_w(w_hi) >> (nb - k);
	W x2 = ~w_lo & m, x3 = ~w_hi & m;
	int t = k >> 1;
	// bit t of x1 above bit t of x3 (kmer.h:82-83; the middle base always differs between strands for odd k): bit t of x1 is bit k-1-t of the
	// window, bit t of x3 the complement of the window's bit t -- as a lane mask, so that the selection is bitwise (no compare, no v_cndmask)
	const W sel = W(0) - ((w_hi >> t) & (w_hi >> (k - 1 - t)) & W(1));
	W a = (x0 & ~sel) | (x2 & sel), b = (x1 & ~sel) | (x3 & sel);
	W h0 = mix_k<W>((a + b) & m, m);
	W h1 = mix_k<W>(h0 ^ b, m);
	y0 = (h0 + h1) & m;
	y1 = h1;
}

// ---- k > 32 on 32-bit halves -------------------------------------------------------------------------------------------
// gfx950 issues every 64-bit integer instruction (v_lshl_add_u64, v_mad_u64_u32, the b64 shifts) at the 4-cycle rate of a 32-bit
// VOP3 one (scripts/probes/valu_rate.hip), so what counts is the NUMBER of instructions.  A k-bit value, 32 < k < 64, is kept as
// (lo, hi) with hi holding k - 32 bits.  The steps of bfc_hash_64 (kmer.h:30-40) then are
//     v * C (+ c)   : one v_mad_u64_u32 for lo * C, and hi' = (carry word + hi * C) & mh      (C = 2^21 - 1, 265, 21, 2^31 + 1:
//                     ~v + (v << 21) = v (2^21 - 1) - 1;  v + (v << 3) + (v << 8) = 265 v;  v + (v << 2) + (v << 4) = 21 v;  v + (v << 31))
//     v ^= v >> s   : lo ^= alignbit(hi, lo, s); hi ^= hi >> s   (the latter vanishes for k - 32 <= s)
// instead of the compiler's generic 64-bit code with two masks per step.  KC > 0: k is known at compile time (the masks fold, for
// k = 33 the carry into the single high bit becomes one v_bitop3); KC == 0: k at run time.
struct U2 { uint32_t lo, hi; };
__device__ __forceinline__ uint64_t u2_join(const U2 v) { return ((uint64_t)v.hi << 32) | v.lo; }
__device__ __forceinline__ U2 u2_split(uint64_t v) { U2 r; r.lo = (uint32_t)v; r.hi = (uint32_t)(v >> 32); return r; }

template <int KC> __device__ __forceinline__ U2 mix2(U2 v, int k) // v.hi may carry garbage above bit k - 32: the first step masks
{
	const int kh = (KC ? KC : k) - 32;
	const uint32_t mh = (1u << kh) - 1u;
	uint64_t d = (uint64_t)v.lo * 0x1FFFFFu + ~0ULL;
	v.hi = ((uint32_t)(d >> 32) + v.hi * 0x1FFFFFu) & mh; v.lo = (uint32_t)d;
	v.lo ^= __builtin_amdgcn_alignbit(v.hi, v.lo, 24); if (!KC || kh > 24) v.hi ^= v.hi >> 24;
	d = (uint64_t)v.lo * 265u;
	v.hi = ((uint32_t)(d >> 32) + v.hi * 265u) & mh; v.lo = (uint32_t)d;
	v.lo ^= __builtin_amdgcn_alignbit(v.hi, v.lo, 14); if (!KC || kh > 14) v.hi ^= v.hi >> 14;
	d = (uint64_t)v.lo * 21u;
	v.hi = ((uint32_t)(d >> 32) + v.hi * 21u) & mh; v.lo = (uint32_t)d;
	v.lo ^= __builtin_amdgcn_alignbit(v.hi, v.lo, 28); if (!KC || kh > 28) v.hi ^= v.hi >> 28;
	d = (uint64_t)v.lo * 0x80000001u;
	v.hi = ((uint32_t)(d >> 32) + v.hi) & mh; v.lo = (uint32_t)d; // hi * (2^31 + 1) = hi + (hi << 31): bit 31 lies above mh
	return v;
}

// kmer.h:79-88 from the two forward windows given as halves (w?_hi may carry the stream's next bits above bit k - 32).
// Strand: bit t = k >> 1 of x1 is bit k-1-t of the window, bit t of x3 the complement of its bit t; both lie below bit 32.
template <int KC> __device__ __forceinline__ void kmer_hash_from_windows2(int k_, uint32_t wl_lo, uint32_t wl_hi, uint32_t wh_lo, uint32_t wh_hi, U2 &y0, U2 &y1)
{
	const int k = KC ? KC : k_, kh = k - 32, s = 32 - kh, t = k >> 1;
	const uint32_t mh = (1u << kh) - 1u;
	const uint32_t sel = 0u - ((wh_lo >> t) & (wh_lo >> (k - 1 - t)) & 1u); // all ones: the reverse strand is the canonical one
	const uint32_t rl = __brev(wl_lo), rh = __brev(wh_lo);
	// forward planes x0, x1 = bit reversal of the window (kmer.h:13-14); reverse planes x2, x3 = its complement (kmer.h:15-16)
	U2 a, b;
	a.lo = (__builtin_amdgcn_alignbit(rl, __brev(wl_hi), s) & ~sel) | (~wl_lo & sel);
	a.hi = ((rl >> s) & ~sel) | (~wl_hi & sel);
	b.lo = (__builtin_amdgcn_alignbit(rh, __brev(wh_hi), s) & ~sel) | (~wh_lo & sel);
	b.hi = ((rh >> s) & ~sel) | (~wh_hi & sel); // (garbage above bit kh in a.hi, b.hi: masked by the first step of mix2)
	const U2 h0 = mix2<KC>(u2_split(u2_join(a) + u2_join(b)), k);
	U2 x; x.lo = h0.lo ^ b.lo; x.hi = h0.hi ^ b.hi;
	const U2 h1 = mix2<KC>(x, k);
	y0 = u2_split(u2_join(h0) + u2_join(h1)); y0.hi &= mh;
	y1 = h1;
}

// the bloom hash as a function of y (kmer.h:85-86 solved for h0): hash = (h0^h1)<<k | (h0+h1)&m
template <typename W> __device__ __forceinline__ uint64_t bloom_hash(int k, W y0, W y1, W m)
{
	W h0 = (y0 - y1) & m;
	return ((uint64_t)(h0 ^ y1) << k) | (uint64_t)y0;
}

// htab.c:45-58 -- sub-table index and slot key (count field preset to 1)
__device__ __forceinline__ uint32_t ch_subkey(int k, int l_pre, uint64_t y0, uint64_t y1, uint64_t &key)
{
	if (k <= 32) {
		int t = 2 * k - l_pre;
		uint64_t z = (y0 << k) | y1;
		key = ((z & ((1ULL << t) - 1)) << 14) | 1;
		return (uint32_t)(z >> t);
	} else {
		int t = k - l_pre;
		int sh = (t + k < 50) ? k : 50 - t;
		key = ((((y0 & ((1ULL << t) - 1)) << sh) ^ y1) << 14) | 1;
		return (uint32_t)(y0 >> t);
	}
}

// ---- region-owned table segments ----
// Every occurrence of a k-mer falls into the same bloom region f = block id >> R, and block id = the low bf_shift-9 bits of
// hash = (h0^h1)<<k | y0 (kmer.h:87): bits [lo, hi) of y0, lo = min(R, k), hi = max(lo, min(k, bf_shift-9)), ARE the low hi-lo bits of f.
// Inside region f a k-mer is therefore identified by y minus those bits: 2k - (hi - lo) bits, which together with the 14 count bits
// of htab.c:7-17 fit one 64-bit word for every configuration BASELINE.json names (c2: 46, c3: 48, c4: 46 identity bits).
struct SegGeom { int k, lo, hi; }; // hi - lo = implied bits
__host__ __device__ __forceinline__ uint64_t seg_id(const SegGeom g, uint64_t y0, uint64_t y1)
{
	const uint64_t low = y0 & ((1ULL << g.lo) - 1), up = g.hi < g.k ? y0 >> g.hi : 0;
	return low | up << g.lo | y1 << (g.k - (g.hi - g.lo));
}
__host__ __device__ __forceinline__ void seg_unpack(const SegGeom g, uint64_t f_global, uint64_t id, uint64_t &y0, uint64_t &y1)
{
	const int nimp = g.hi - g.lo, kb = g.k - nimp; // bits of y0 kept in the id
	const uint64_t imp = f_global & ((1ULL << nimp) - 1);
	const uint64_t kept = id & ((1ULL << kb) - 1);
	y0 = (kept & ((1ULL << g.lo) - 1)) | imp << g.lo | (g.hi < g.k ? (kept >> g.lo) << g.hi : 0);
	y1 = id >> kb;
}
__host__ __device__ __forceinline__ uint32_t seg_home(uint64_t id) { return (uint32_t)((id * 0x9E3779B97F4A7C15ULL) >> 38); }

// bbf.c:27-41 -- block id and the n_hashes bit positions (8..511) inside the 512-bit block
struct BloomAddr {
	uint64_t blk;
	uint32_t h1, h2;
};
__device__ __forceinline__ BloomAddr bloom_addr(uint64_t hash, int bf_shift)
{
	BloomAddr a;
	int x = bf_shift - 9;
	a.blk = hash & ((1ULL << x) - 1);
	a.h1 = (uint32_t)(hash >> x) & 511u;
	a.h2 = (uint32_t)(hash >> bf_shift) & 511u;
	if ((a.h2 & 31u) == 0) a.h2 = (a.h2 + 1) & 511u;
	return a;
}
// next position of the walk z = h1, h1+h2, ... (mod 512) that is not in the lock byte (z >= 8)
__device__ __forceinline__ uint32_t bloom_next(uint32_t &z, uint32_t h2)
{
	while (z < 8) z = (z + h2) & 511u;
	uint32_t r = z;
	z = (z + h2) & 511u;
	return r;
}

} // namespace bfcg
