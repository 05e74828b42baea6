// bfcg_dev.h -- device-side pieces shared by the kernel translation units (bfcg_kernels.hip, bfcg_bloom3.hip): record words, the
// first-setter tables of the bloom insert, the arguments of the bloom / commit kernels, the 32-bit decode geometry of
// 12-byte records.  Everything here is inline device code or plain structs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kmer_dev.h"
#include "bfcg_internal.h"

using namespace bfcg;

template <int RD> struct RecW { uint32_t d[RD]; };

template <int RD> __device__ __forceinline__ RecW<RD> rec_load(const uint32_t *p);
template <> __device__ __forceinline__ RecW<3> rec_load<3>(const uint32_t *p)
{ const uint3 v = *reinterpret_cast<const uint3 *>(p); RecW<3> r; r.d[0] = v.x; r.d[1] = v.y; r.d[2] = v.z; return r; }
template <> __device__ __forceinline__ RecW<4> rec_load<4>(const uint32_t *p)
{ const uint4 v = *reinterpret_cast<const uint4 *>(p); RecW<4> r; r.d[0] = v.x; r.d[1] = v.y; r.d[2] = v.z; r.d[3] = v.w; return r; }
template <> __device__ __forceinline__ RecW<5> rec_load<5>(const uint32_t *p) // records are only dword-aligned
{ RecW<5> r; r.d[0] = p[0]; r.d[1] = p[1]; r.d[2] = p[2]; r.d[3] = p[3]; r.d[4] = p[4]; return r; }
template <int RD> __device__ __forceinline__ void rec_store(uint32_t *p, const RecW<RD> &r);
template <> __device__ __forceinline__ void rec_store<3>(uint32_t *p, const RecW<3> &r)
{ *reinterpret_cast<uint3 *>(p) = make_uint3(r.d[0], r.d[1], r.d[2]); }
template <> __device__ __forceinline__ void rec_store<4>(uint32_t *p, const RecW<4> &r)
{ *reinterpret_cast<uint4 *>(p) = make_uint4(r.d[0], r.d[1], r.d[2], r.d[3]); }
template <> __device__ __forceinline__ void rec_store<5>(uint32_t *p, const RecW<5> &r)
{ p[0] = r.d[0]; p[1] = r.d[1]; p[2] = r.d[2]; p[3] = r.d[3]; p[4] = r.d[4]; }


// Optional order bookkeeping for the byte-identical `-d` dump: per slot the stamp (batch << 32 | file index)
// of the FIRST bfc_ch_insert call that created the key, per sub-table the stamp of the LAST call of any kind.  The host
// replays khash's growth from them (bfc_host.c).  Both are order-independent (min / max), so parking and replay keep them exact.
struct TabOrder {
	unsigned long long *first, *sub_last; // NULL: not tracked
	__device__ __forceinline__ void note(uint32_t sub, uint64_t slot, unsigned long long sf, unsigned long long sl) const
	{
		if (first) { atomicMin(&first[slot], sf); atomicMax(&sub_last[sub], sl); }
	}
};


#define FS_EMPTY 0xffffffffffffffffULL

// first-setter table: entry = bit offset inside the region (high 32) | k-mer index (low 32);
// atomicMin keeps, per bit, the earliest k-mer (file order) that finds the bit clear.
template <bool GLOBAL>
__device__ __forceinline__ bool fs_insert(unsigned long long *tab, uint32_t cap_mask, uint32_t bitoff, uint32_t idx, uint32_t max_probe)
{
	const unsigned long long e = ((unsigned long long)bitoff << 32) | idx;
	uint32_t p = ((bitoff * 0x9E3779B1u) >> 12 ^ bitoff) & cap_mask; // low bits of a multiplicative hash are weak: fold the high half in
	for (uint32_t probe = 0; probe <= max_probe; ++probe, p = (p + 1) & cap_mask) {
		unsigned long long cur = GLOBAL ? __hip_atomic_load(&tab[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tab[p];
		if (cur == FS_EMPTY) {
			cur = atomicCAS(&tab[p], FS_EMPTY, e);
			if (cur == FS_EMPTY) return true;
		}
		if ((uint32_t)(cur >> 32) == bitoff) { if (e < cur) atomicMin(&tab[p], e); return true; }
	}
	return false;
}
// true and the first setter's index if the bit has an entry (<=> it was clear before the batch)
template <bool GLOBAL>
__device__ __forceinline__ bool fs_lookup(const unsigned long long *tab, uint32_t cap_mask, uint32_t bitoff, uint32_t &first)
{
	uint32_t p = ((bitoff * 0x9E3779B1u) >> 12 ^ bitoff) & cap_mask;
	for (uint32_t probe = 0; probe <= cap_mask; ++probe, p = (p + 1) & cap_mask) {
		unsigned long long cur = GLOBAL ? __hip_atomic_load((unsigned long long *)&tab[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tab[p];
		if (cur == FS_EMPTY) return false;
		if ((uint32_t)(cur >> 32) == bitoff) { first = (uint32_t)cur; return true; }
	}
	return false;
}

struct BloomArgs {
	const uint32_t *recs;          // fine-bucketed records (RD dwords each)
	const uint32_t *start;         // fine bucket starts (n_fine+1)
	unsigned long long *bloom;     // first bloom filter (device)
	unsigned long long *bloom_hi;  // second bloom filter (filter mode) or NULL
	unsigned long long *table;     // count table or NULL
	unsigned long long *stats;
	uint64_t *tab_ovf; uint32_t tab_ovf_cap; unsigned long long *ovf_cnt;
	unsigned long long *pool; uint32_t pool_slices;        // slow-path first-setter pool: pool_slices lock words, then pool_slices slices of 2^(R+10) entries
	uint8_t *seen_out;             // optional debug: seen flag (1/2) per batch position
	uint64_t *agg_out;             // aggregated seen k-mers: three planes [y0 | y1 | count|high<<16] of [n_fine][ag_cap], or NULL = commit inline
	uint32_t *agg_cnt;             // entries per fine bucket
	uint32_t *stream_out;          // STREAM mode: seen k-mers as records, region f's at [start[f], start[f] + agg_cnt[f])
	unsigned long long *seg_tab;   // region-owned table segments (KParams.seg) or NULL
	uint32_t n_fine;               // fine buckets (= bloom regions) this launch owns
	TabOrder ord;                  // optional first/last stamps (byte-identical dump)
	unsigned long long batch_hi;   // batch number << 32: high half of a stamp
	const uint32_t *cnt2; uint32_t cap2; // one-pass level 2: region f's records are recs[f * cap2 .. + cnt2[f]) (cap2 = 0: start[] says where)
	// Hand-over log of the region-owned table: region f owns ho[f * ho_stride .. + ho_stride); k_bloom appends its seen k-mers
	// behind ho_cur[f] -- which lives on from batch to batch -- and notes where the batch ended (ho_mark: this batch's page, one word per
	// region); k_commit_seg applies the pages of several batches in one pass over the segment and clears the cursor.  ho_stride == 0:
	// the batch's entries sit at its records' offsets in stream_out instead (two-pass level 2: a region's share has no bound) and are applied at once.
	unsigned long long *ho; uint32_t ho_stride; uint32_t *ho_cur; uint32_t *ho_mark;
	uint32_t ho_pages, ho_mark_stride;  // k_commit_seg: pages to apply (page j's marks at ho_mark + j * ho_mark_stride)
	unsigned long long *ho_keys;        // k_commit_seg: keys created by page j, slotted: ho_keys[j * ST_SLOTS + (f & (ST_SLOTS - 1))]
	const uint32_t *flags;         // one-pass partition, this batch's slot: [0] a level-1 slab overflowed, [2] a region's slab (NULL: two-pass batch)
	const uint32_t *sticky;        // an earlier batch of the run overflowed (written by k_seal on stage B's stream only)
};

// where region f's records are
__device__ __forceinline__ void region_list(const BloomArgs &A, uint32_t f, uint32_t &rs, uint32_t &n)
{
	if (A.cap2) { const uint32_t c = A.cnt2[f]; rs = f * A.cap2; n = c < A.cap2 ? c : A.cap2; }
	else if (A.cnt2) { rs = A.start[f]; n = A.cnt2[f]; } // (two-pass level 2 over slabs with dead records: a bucket's last region is followed by a gap)
	else { rs = A.start[f]; n = A.start[f + 1] - rs; }
}
// A batch the one-pass partition gave up on must change nothing: the host replays it (and every batch behind it) through the two-pass one.
__device__ __forceinline__ bool batch_poisoned(const BloomArgs &A) { return (A.sticky && *A.sticky) || (A.flags && (A.flags[0] | A.flags[2])); }


// Bloom address and hand-over entry of a 12-byte record on 32-bit words (what decode_rec / seg_id compute through 64-bit y0, y1).  The record
// holds y0' = y0 without its level-1 bucket bits [lo, lo+n) in bits [0, a) and y1 in bits [a, a+k); for k >= bf_shift-9 the block id is the low
// bf_shift-9 bits of y0 (kmer.h:87), so
//     block inside the region = y0' & (2^R - 1)
//     h1 | h2 << 9 = bits [bf_shift-9, bf_shift+9) of the hash (h0^h1) << k | y0  =  (y0' >> up) | ((y0 - y1) ^ y1) << (k - (bf_shift-9))
// (up = where the part of y0 above the block id starts inside y0'; only the low bits of y0 - y1 are needed), and the k-mer's identity inside the
// region -- y0 without the bits the region implies, then y1 (kmer_dev.h: seg_id) -- is  (y0' & (2^R-1)) | (y0' >> up) << R | y1 << (k - F).
struct Dec3 { int ok, a, lo, n, up, sh_x, R, sh_flag, sh_y1; uint32_t lowmask, rmask, mk32; };
__device__ __forceinline__ Dec3 dec3_geom(const KParams &P)
{
	Dec3 g;
	g.a = P.k - P.rec_n; g.lo = P.rec_lo; g.n = P.rec_n; g.up = P.rec_n ? P.rec_lo : P.bf_shift - 9; g.sh_x = P.k - (P.bf_shift - 9);
	g.R = P.R; g.sh_flag = g.a + P.k - 32; g.sh_y1 = P.k - P.F;
	g.lowmask = P.rec_n ? (1u << (P.rec_lo & 31)) - 1u : 0xffffffffu; g.rmask = (1u << P.R) - 1u; g.mk32 = P.k >= 32 ? 0xffffffffu : (1u << P.k) - 1u;
	g.ok = P.k >= P.bf_shift - 9 && P.bf_shift + 9 <= 2 * P.k && g.a >= 1 && g.a <= 31 && g.a + P.k >= 32 && (P.rec_n == 0 || P.rec_lo + P.rec_n <= 31)
	       && g.up <= g.a && g.sh_x >= 0 && g.sh_x <= 31 && g.sh_y1 >= 0 && g.sh_y1 <= 32 && (P.rec_n == 0 || P.rec_lo + P.rec_n == P.bf_shift - 9) && P.R <= g.up;
	return g;
}

// first-setter table in LDS, 4 bytes per entry: bit offset in the region << 13 | index into the LDS list of
// k-mers with clear bits.  The earliest k-mer (file order = record idx, read through the list) wins a bit.
#define FS32_EMPTY 0xffffffffu
__device__ __forceinline__ uint32_t fs32_slot(uint32_t bitoff, uint32_t mask) { return ((bitoff * 0x9E3779B1u) >> 12 ^ bitoff) & mask; }

__device__ __forceinline__ bool fs32_insert(unsigned int *fs, uint32_t mask, uint32_t bitoff, uint32_t li, uint32_t idx, const unsigned int *list_idx)
{
	const uint32_t e = (bitoff << 13) | li;
	uint32_t p = fs32_slot(bitoff, mask);
	for (int probe = 0; probe < 1024; ++probe, p = (p + 1) & mask) {
		uint32_t cur = fs[p];
		if (cur == FS32_EMPTY) {
			cur = atomicCAS(&fs[p], FS32_EMPTY, e);
			if (cur == FS32_EMPTY) return true;
		}
		if ((cur >> 13) == bitoff) {
			while (list_idx[cur & 0x1fffu] > idx) { // the holder is later in file order: take the bit over
				uint32_t old = atomicCAS(&fs[p], cur, e);
				if (old == cur) break;
				cur = old;
			}
			return true;
		}
	}
	return false;
}
// a k-mer that touches a bit which HAS an entry competes for it (no entry is created: the bit is uncontended)
// (returns whether the bit has an entry)
__device__ __forceinline__ bool fs32_compete(unsigned int *fs, uint32_t mask, uint32_t bitoff, uint32_t li, uint32_t idx, const unsigned int *list_idx)
{
	const uint32_t e = (bitoff << 13) | li;
	uint32_t p = fs32_slot(bitoff, mask);
	for (uint32_t probe = 0; probe <= mask; ++probe, p = (p + 1) & mask) {
		uint32_t cur = fs[p];
		if (cur == FS32_EMPTY) return false;
		if ((cur >> 13) == bitoff) {
			while (list_idx[cur & 0x1fffu] > idx) {
				uint32_t old = atomicCAS(&fs[p], cur, e);
				if (old == cur) break;
				cur = old;
			}
			return true;
		}
	}
	return false;
}
// list index of the first setter of a bit, or FS32_EMPTY if the bit has no entry
__device__ __forceinline__ uint32_t fs32_lookup(const unsigned int *fs, uint32_t mask, uint32_t bitoff)
{
	uint32_t p = fs32_slot(bitoff, mask);
	for (uint32_t probe = 0; probe <= mask; ++probe, p = (p + 1) & mask) {
		uint32_t cur = fs[p];
		if (cur == FS32_EMPTY) return FS32_EMPTY;
		if ((cur >> 13) == bitoff) return cur & 0x1fffu;
	}
	return FS32_EMPTY;
}


// ---- the four bit positions of a k-mer inside its 512-bit block without a loop (k_bloom3, k_bloom3fm, k_query4)
struct B3Pos { uint32_t b0, b1, b2, b3; };
// bbf.c:33-41: positions z = h1, h1 + h2, ... (mod 512), those below 8 (the lock byte) skipped.  The first five candidates of the walk; a
// skipped one shifts the rest by one (the tests combine as lane masks on the scalar unit); a second skip shows as a position below 8 and
// takes the reference's loop (h2 < 8 or > 504 AND a step into the lock byte: one k-mer in a thousand).
__device__ __forceinline__ B3Pos b3_positions(uint32_t h1, uint32_t h2)
{
	const uint32_t u1 = h1 + h2, u2 = u1 + h2, u3 = u2 + h2, u4 = u3 + h2;
	const uint32_t c0 = h1, c1 = u1 & 511u, c2 = u2 & 511u, c3 = u3 & 511u, c4 = u4 & 511u;
	const bool s0 = c0 < 8u, s1 = s0 | (c1 < 8u), s2 = s1 | (c2 < 8u), s3 = s2 | (c3 < 8u);
	B3Pos p;
	p.b0 = s0 ? c1 : c0; p.b1 = s1 ? c2 : c1; p.b2 = s2 ? c3 : c2; p.b3 = s3 ? c4 : c3;
	if (__builtin_expect(min(min(p.b0, p.b1), min(p.b2, p.b3)) < 8u, 0)) {
		uint32_t z = h1;
		p.b0 = bloom_next(z, h2); p.b1 = bloom_next(z, h2); p.b2 = bloom_next(z, h2); p.b3 = bloom_next(z, h2);
	}
	return p;
}
// the dword of bit b of block bl inside the LDS region, by its byte offset; bit b of that dword
__device__ __forceinline__ unsigned int *b3_wordp(unsigned int *region, uint32_t bl64, uint32_t b)
{ return reinterpret_cast<unsigned int *>(reinterpret_cast<unsigned char *>(region) + (bl64 | ((b >> 3) & 0x3cu))); }
__device__ __forceinline__ uint32_t b3_bit(uint32_t w, uint32_t b) { return __builtin_amdgcn_ubfe(w, b, 1u); } // (the field offset is b's low five bits)


namespace bfcg {
// bfcg_bloom3.hip
hipError_t set_bloom3_lds_attr(int lds);
void run_bloom3(const KParams &P, const BloomArgs &A, int nfine, size_t lds, hipStream_t st);
hipError_t set_bloom3fm_lds_attr(int lds);
void run_bloom3fm(const KParams &P, const BloomArgs &A, int nfine, size_t lds, hipStream_t st);
}
