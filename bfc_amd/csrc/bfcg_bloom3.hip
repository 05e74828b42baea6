// bfcg_bloom3.hip -- k_bloom3, the bloom insert of the default path (a translation unit of its own: the kernel is iterated on most)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "kmer_dev.h"
#include "bfcg_internal.h"
#include "bfcg_dev.h"

using namespace bfcg;

// ------------------------------------------------------------------------------------------
// k_bloom3 (round 4): the bloom insert of the DEFAULT path -- 12-byte records on 32-bit words (dec3_geom), n_hashes = 4, seen k-mers into the
// hand-over log / stream of the region-owned table -- written for INSTRUCTION COUNT.  k_bloom on config c3 issued 5.0 VALU and 4.5 SALU wave
// instructions per k-mer at 62 % of the chip's VALU issue rate (profiles/round4_k_bloom.md: SQ counters); it was bound by what it executed, not
// by memory: divergent loops (bloom_next's skip of the lock byte four times per k-mer and pass, probe loops inlined per bit), one ballot + one
// LDS atomic round trip per record and output, the record decoded again in every pass over the list, and a pass C that looked every bit up again.
// Same protocol (the comment above k_bloom), restated:
//   pass 1  every record: decode, the four bit positions WITHOUT a loop (a conditional step over the lock byte per position; the one-in-4000
//           k-mer that would need a second step in a row takes the loop), four LDS reads back to back, then ONE LDS atomic per wave and round for
//           the seen k-mers' slots in the hand-over log and ONE for the list slots of the k-mers with clear bits.  A list entry is 10 bytes:
//           file index | block, h1, h2, clear-bit mask (30 bits: no pass decodes a record again) | record index (16 bits: only an emit needs it);
//   pass A  the returning ORs of a k-mer's clear bits are issued back to back; contended bits (rare once the filter is warm: 2 % of the touches)
//           enter the first-setter table in a loop over the lane's contended bits, not in four inlined copies;
//   pass B  the home slots of a k-mer's clear bits in the first-setter table are read back to back; an empty home slot means no entry, i.e. the
//           k-mer alone touched that bit: it is a first setter, NOT seen, and needs no pass C (bit 30 of its list word stays clear).  Bits with a
//           non-empty home slot compete as before;
//   pass C  only k-mers ALL of whose clear bits have entries look them up; those that won none are seen and emitted (record re-read by index).
// The records of round t+1 are requested before round t is processed.  Regions whose list, first-setter table or record index overflow take the
// same exact HBM-pool path as in k_bloom.  Cold batches with copies resolved by class (KParams.dedupe) stay with k_bloom<..., F3>.
#define B3_UND 0x40000000u /* list word: every clear bit of this k-mer has a first-setter entry -- pass C decides */
// One workgroup per region.  (Tried in round 4 and not kept, profiles/round4_k_bloom.md: workgroups that WALK their regions with the next
// region's records and filter slice requested into registers under the list passes -- the registers that must live through the passes spill
// (86 - 225 VGPRs to scratch at the 80 the occupancy allows) and c3's bloom stage went 66 -> 120 - 125 ms; touching the cache lines of the region
// a later workgroup of the same XCD will take, so that its loads hit in L2: 66.4 -> 70.1 ms.)
//
// COLD (round 5): a batch into a filter that is still filling up.  There nearly every k-mer brings clear bits and half of them are copies of an
// earlier k-mer of the same batch (c3's first call holds every genome k-mer twice): the first-setter protocol -- returning ORs, an entry per
// contended bit, competitors, look-ups, all by divergent probe loops on LDS compare-and-swaps -- took 28 ps per k-mer for the launch into the
// empty filter and 13 for the next against 6.7 once the filter is warm (profiles/round4_k_bloom.md, section 3).  A k-mer touches ONE 64-byte
// block (bbf.c:27-31), so the reference's order only matters inside a block, and a region has 2^R of them with a handful of listed k-mers each:
// the list is ordered by (block, file index) -- one LDS counter per block, a scan of 2^R counters, a scatter, ranks by comparing a block's few
// indices -- and ONE LANE PER BLOCK walks its k-mers in file order doing literally what bfc_bf_insert does (bbf.c:33-44): test the four bits,
// seen iff all are set, set them.  The lane owns the block: plain LDS reads, fire-and-forget ORs, no table, no probing, no retries, and the
// result is the sequential one by construction.  A list entry is 12 bytes (file index, packed address + clear mask, record index, position in
// block order); with no first-setter table beside it the list holds 2 879 entries where the protocol's held 2 021.
#define B3_SEEN 0x40000000u /* COLD: list word: the walk found every bit of this k-mer set */
#define B3_COLD_NR 8       /* COLD: list positions per thread in the rank pass (8 x 512 = 4096 >= the cold list's capacity) */
template <int BT, int PF, bool COLD>
__global__ __launch_bounds__(BT, PF <= 2 ? 8 : 6) void k_bloom3(KParams P, BloomArgs A)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	__shared__ uint32_t s_list_n, s_ovf, s_pool_off, s_emit_n, s_fs_used, s_wb_n, s_pad3[2];
	const uint32_t f = blockIdx.x;
	const bool ho_log = A.ho_stride != 0;
	if (batch_poisoned(A)) { if (ho_log && threadIdx.x == 0) A.ho_mark[f] = A.ho_cur[f]; return; } // (an empty page: the batch will be replayed)
	uint32_t rs, n;
	region_list(A, f, rs, n);
	if (n == 0) { if (threadIdx.x == 0) { if (ho_log) A.ho_mark[f] = A.ho_cur[f]; else if (A.agg_cnt) A.agg_cnt[f] = 0; } return; }
	uint32_t ho_cur0 = 0;
	unsigned long long *ho_base;
	if (ho_log) {
		ho_cur0 = A.ho_cur[f];
		if (ho_cur0 + n > A.ho_stride) { // (the host commits before a log can fill up: a bug if it ever happens -- loudly, not silently)
			if (threadIdx.x == 0) { atomicAdd(&A.stats[(size_t)(f & (ST_SLOTS - 1)) * ST_N + ST_ERR_POOL], 1ULL); A.ho_mark[f] = ho_cur0; }
			return;
		}
		ho_base = A.ho + (uint64_t)f * A.ho_stride + ho_cur0;
	} else ho_base = reinterpret_cast<unsigned long long *>(A.stream_out) + rs;
	A.stats += (size_t)(f & (ST_SLOTS - 1)) * ST_N;
	const uint32_t region_dw = 16u << P.R;
	unsigned char *sp = smem;
	unsigned int *region = reinterpret_cast<unsigned int *>(sp); sp += (size_t)region_dw * 4;
	const uint32_t nblk = 1u << P.R;
	// COLD: per block of the region its listed k-mers' count, then (boff) its first position in block order / (bfill) the scatter's cursor
	unsigned int *bcnt = reinterpret_cast<unsigned int *>(sp), *boff = bcnt + nblk; // boff has nblk + 1 entries
	if (COLD) sp += (size_t)(2 * nblk + 4) * 4;
	unsigned int *fs = reinterpret_cast<unsigned int *>(sp); if (!COLD) sp += (size_t)P.fs_cap * 4;
	unsigned int *la = reinterpret_cast<unsigned int *>(sp); sp += (size_t)P.list_cap * 4;   // file-order index
	unsigned int *lb = reinterpret_cast<unsigned int *>(sp); sp += (size_t)P.list_cap * 4;   // block | h1 << 8 | h2 << 17 | clear-bit mask << 26 (| B3_UND)
	unsigned short *lc = reinterpret_cast<unsigned short *>(sp); sp += (size_t)P.list_cap * 2; // record index inside the region's slab
	unsigned short *ord = reinterpret_cast<unsigned short *>(sp);                            // COLD: the list's entries in (block, file index) order
	const uint32_t fs_mask = P.fs_cap - 1;
	const uint32_t *recs = A.recs + (uint64_t)rs * 3;
	unsigned int *g_region = reinterpret_cast<unsigned int *>(A.bloom) + (uint64_t)f * region_dw;
	const uint32_t imp = (P.f_base + f) >> P.F2; // the region's level-1 bucket: the bits of y0 its records do not store
	const Dec3 D = dec3_geom(P);
	const int tid = threadIdx.x, lane = tid & 63;
	const uint32_t mask_a = (1u << D.a) - 1u, imp_sh = D.n ? imp << D.lo : 0u, sh_up = D.lo + D.n;
#ifdef BFCG_MEASURE
	const bool timing = BFCG_ABL(P, 64) && tid == 0;
	long long tq[7] = {0, 0, 0, 0, 0, 0, 0};
	if (timing) tq[0] = clock64();
#endif
	// block (8 bits at R = 8) | h1 << 8 | h2 << 17 of a record (decode_fast3, packed)
	auto addr3 = [&](uint32_t d0, uint32_t d1) -> uint32_t {
		const uint32_t y0d = d0 & mask_a, upper = y0d >> D.up;
		const uint32_t y0lo = D.n ? (y0d & D.lowmask) | imp_sh | (upper << sh_up) : y0d;
		const uint32_t y1lo = __builtin_amdgcn_alignbit(d1, d0, D.a) & D.mk32;
		const uint32_t hh = upper | (((y0lo - y1lo) ^ y1lo) << D.sh_x);
		uint32_t h2 = (hh >> 9) & 511u;
		h2 |= (uint32_t)((h2 & 31u) == 0); // bbf.c:33 (the low five bits are zero: + 1 is | 1)
		return (y0d & D.rmask) | ((hh & 511u) << 8) | (h2 << 17);
	};
	auto entry3 = [&](uint32_t d0, uint32_t d1) -> unsigned long long { // the 8-byte hand-over entry: identity inside the region << 1 | high-quality flag
		const unsigned long long AA = d0 | ((unsigned long long)d1 << 32);
		const unsigned long long y1 = (AA >> D.a) & ((1ULL << P.k) - 1);
		const uint32_t y0d = d0 & mask_a;
		const unsigned long long id = (unsigned long long)((y0d & D.rmask) | ((y0d >> D.up) << D.R)) | (y1 << D.sh_y1);
		return (id << 1) | (unsigned long long)((d1 >> D.sh_flag) & 1u);
	};
	RecW<3> cur[PF], nxt[PF];
#pragma unroll
	for (int u = 0; u < PF; ++u) {
		const uint32_t i = tid + u * BT;
		cur[u].d[0] = cur[u].d[1] = cur[u].d[2] = 0; nxt[u] = cur[u]; // (lanes beyond the region's records compute on zeros)
		if (i < n) cur[u] = rec_load<3>(recs + (uint64_t)i * 3);
		if (i + BT * PF < n) nxt[u] = rec_load<3>(recs + (uint64_t)(i + BT * PF) * 3);
	}
	{ // stage the region (16-byte loads), clear the first-setter table
		const uint4 *src = reinterpret_cast<const uint4 *>(g_region);
		uint4 *dst = reinterpret_cast<uint4 *>(region);
		for (uint32_t i = tid; i < region_dw / 4; i += BT) dst[i] = src[i];
		if constexpr (COLD) { for (uint32_t i = tid; i < 2 * nblk + 4; i += BT) bcnt[i] = 0; }
		else {
			uint4 *f4 = reinterpret_cast<uint4 *>(fs);
			for (uint32_t i = tid; i < P.fs_cap / 4; i += BT) f4[i] = make_uint4(FS32_EMPTY, FS32_EMPTY, FS32_EMPTY, FS32_EMPTY);
		}
		if (tid == 0) { s_list_n = 0; s_emit_n = 0; s_ovf = 0; s_fs_used = 0; s_wb_n = 0; }
	}
	__syncthreads();
#ifdef BFCG_MEASURE
	if (timing) tq[1] = clock64();
#endif
	volatile uint32_t *v_ovf = &s_ovf;

	// ---- pass 1: classify against the pre-batch region; seen -> hand-over log, clear bits -> list
	for (uint32_t base = 0; base < n; base += BT * PF) {
		uint32_t pk[PF], um[PF];
#pragma unroll
		for (int u = 0; u < PF; ++u) {
			pk[u] = 0; um[u] = 0;
			if (base + u * BT + (uint32_t)(tid & ~63) < n) { // (wave-uniform: a wave whose 64 slots all lie beyond the region's records skips the work)
				pk[u] = addr3(cur[u].d[0], cur[u].d[1]);
				const B3Pos b = b3_positions((pk[u] >> 8) & 511u, pk[u] >> 17);
				const uint32_t bl64 = (pk[u] & 255u) << 6;
				const uint32_t w0 = *b3_wordp(region, bl64, b.b0), w1 = *b3_wordp(region, bl64, b.b1), w2 = *b3_wordp(region, bl64, b.b2), w3 = *b3_wordp(region, bl64, b.b3);
				um[u] = (b3_bit(w0, b.b0) | (b3_bit(w1, b.b1) << 1) | (b3_bit(w2, b.b2) << 2) | (b3_bit(w3, b.b3) << 3)) ^ 15u;
			}
		}
		unsigned long long ms[PF], ml[PF];
		uint32_t tot_s = 0, tot_l = 0;
#pragma unroll
		for (int u = 0; u < PF; ++u) {
			const bool act = base + tid + u * BT < n;
			ms[u] = __ballot(act && um[u] == 0); ml[u] = __ballot(act && um[u] != 0);
			tot_s += (uint32_t)__popcll(ms[u]); tot_l += (uint32_t)__popcll(ml[u]);
		}
		uint32_t o_s = 0, o_l = 0;
		if (lane == 0) { if (tot_s) o_s = atomicAdd(&s_emit_n, tot_s); if (tot_l) o_l = atomicAdd(&s_list_n, tot_l); }
		o_s = __builtin_amdgcn_readfirstlane(o_s); o_l = __builtin_amdgcn_readfirstlane(o_l);
#pragma unroll
		for (int u = 0; u < PF; ++u) {
			const uint32_t below_lo = (uint32_t)__builtin_amdgcn_mbcnt_lo((uint32_t)ms[u], 0u), below_s = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(ms[u] >> 32), below_lo);
			const uint32_t blow_lo = (uint32_t)__builtin_amdgcn_mbcnt_lo((uint32_t)ml[u], 0u), below_l = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(ml[u] >> 32), blow_lo);
			const bool act = base + tid + u * BT < n;
			if (act && um[u] == 0) { // every bit was set before this batch: seen, whatever the order inside the batch
				ho_base[o_s + below_s] = entry3(cur[u].d[0], cur[u].d[1]);
				if (A.seen_out) A.seen_out[cur[u].d[2]] = 2;
			} else if (act) {
				const uint32_t li = o_l + below_l;
				if (li < P.list_cap) {
					la[li] = cur[u].d[2]; lb[li] = pk[u] | (um[u] << 26); lc[li] = (unsigned short)(base + tid + u * BT);
					if constexpr (COLD) __hip_atomic_fetch_add(&bcnt[pk[u] & 255u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				}
			}
			o_s += (uint32_t)__popcll(ms[u]); o_l += (uint32_t)__popcll(ml[u]);
		}
		if (base + BT * PF < n) {
#pragma unroll
			for (int u = 0; u < PF; ++u) {
				cur[u] = nxt[u];
				const uint32_t i = base + 2 * BT * PF + tid + u * BT;
				if (i < n) nxt[u] = rec_load<3>(recs + (uint64_t)i * 3);
			}
		}
	}
	__syncthreads();
#ifdef BFCG_MEASURE
	if (timing) tq[2] = clock64();
#endif
	const uint32_t ln = s_list_n;
	const bool ovf_list = ln > P.list_cap || ln > (COLD ? (uint32_t)(B3_COLD_NR * BT) : 8191u) || n > 65535u; // (13-bit list index in a first-setter entry, 16-bit record index in the list; COLD: 16 positions per thread in the rank pass)
	bool dirty = true;
	if constexpr (COLD) {
		if (!ovf_list) {
			// ---- block offsets: exclusive scan of the 2^R counters by the first wave (2^R <= 256: four blocks per lane)
			if (tid < 64) {
				const uint32_t per = (nblk + 63u) >> 6, b0 = (uint32_t)tid * per;
				uint32_t v[4] = {0, 0, 0, 0}, sum = 0;
#pragma unroll
				for (int t = 0; t < 4; ++t) if ((uint32_t)t < per && b0 + t < nblk) { v[t] = bcnt[b0 + t]; sum += v[t]; }
				uint32_t inc = sum;
#pragma unroll
				for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(inc, o); if (lane >= o) inc += x; }
				uint32_t run = inc - sum;
#pragma unroll
				for (int t = 0; t < 4; ++t) if ((uint32_t)t < per && b0 + t < nblk) { boff[b0 + t] = run; bcnt[b0 + t] = run; run += v[t]; } // (bcnt becomes the scatter's cursor)
				if (tid == 63) boff[nblk] = inc;
			}
			__syncthreads();
			// ---- scatter: the list's entries grouped by block (any order inside a block)
			for (uint32_t li = tid; li < ln; li += BT) ord[atomicAdd(&bcnt[lb[li] & 255u], 1u)] = (unsigned short)li;
			__syncthreads();
			// ---- ranks: an entry's place among its block's entries = how many of them come earlier in the file (a block holds a handful).  Every
			// position is read before any is rewritten: the new places wait in registers across the barrier.
			uint32_t mv[B3_COLD_NR]; // place << 16 | entry, per position this thread owns (B3_COLD_NR x BT >= the cold list's capacity)
#pragma unroll
			for (int t = 0; t < B3_COLD_NR; ++t) {
				const uint32_t pos = (uint32_t)tid + (uint32_t)t * BT;
				mv[t] = 0;
				if (pos < ln) {
					const uint32_t li = ord[pos], x = la[li], bl = lb[li] & 255u, s0 = boff[bl], s1 = boff[bl + 1];
					uint32_t r = 0;
					for (uint32_t j = s0; j < s1; ++j) r += (uint32_t)(la[ord[j]] < x);
					mv[t] = ((s0 + r) << 16) | li;
				}
			}
			__syncthreads();
#pragma unroll
			for (int t = 0; t < B3_COLD_NR; ++t) { const uint32_t pos = (uint32_t)tid + (uint32_t)t * BT; if (pos < ln) ord[mv[t] >> 16] = (unsigned short)mv[t]; }
			__syncthreads();
			// ---- the walk: lane b takes block b's k-mers in file order -- bbf.c:33-44 as it stands: seen iff every bit is set, then set them
			if ((uint32_t)tid < nblk) {
				const uint32_t bl = (uint32_t)tid, bl64 = bl << 6, s0 = boff[bl], s1 = boff[bl + 1];
				uint32_t li_n = s0 < s1 ? ord[s0] : 0u, w_n = s0 < s1 ? lb[li_n] : 0u;
				for (uint32_t j = s0; j < s1; ++j) {
					const uint32_t li = li_n, w = w_n;
					if (j + 1 < s1) { li_n = ord[j + 1]; w_n = lb[li_n]; } // (the next entry is on its way while this one is decided)
					const B3Pos b = b3_positions((w >> 8) & 511u, (w >> 17) & 511u);
					unsigned int *p0 = b3_wordp(region, bl64, b.b0), *p1 = b3_wordp(region, bl64, b.b1), *p2 = b3_wordp(region, bl64, b.b2), *p3 = b3_wordp(region, bl64, b.b3);
					const uint32_t w0 = *p0, w1 = *p1, w2 = *p2, w3 = *p3;
					const uint32_t clr = (b3_bit(w0, b.b0) | (b3_bit(w1, b.b1) << 1) | (b3_bit(w2, b.b2) << 2) | (b3_bit(w3, b.b3) << 3)) ^ 15u; // (bits set before the batch are still set)
					if (clr == 0) lb[li] = w | B3_SEEN;
					else { // (two positions may share a word: ORs, not stores; this lane's later reads follow them in order)
						if (clr & 1u) __hip_atomic_fetch_or(p0, 1u << (b.b0 & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
						if (clr & 2u) __hip_atomic_fetch_or(p1, 1u << (b.b1 & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
						if (clr & 4u) __hip_atomic_fetch_or(p2, 1u << (b.b2 & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
						if (clr & 8u) __hip_atomic_fetch_or(p3, 1u << (b.b3 & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					}
				}
			}
			__syncthreads();
			// ---- emit the listed k-mers the walk found seen (record re-read by its index), behind what pass 1 emitted
			for (uint32_t t0 = 0; t0 < ln; t0 += BT) {
				const uint32_t li = t0 + tid;
				bool seen = false;
				if (li < ln) {
					seen = (lb[li] & B3_SEEN) != 0;
					if (A.seen_out) A.seen_out[la[li]] = seen ? 2 : 1;
				}
				const unsigned long long vote = __ballot(seen);
				if (vote) {
					uint32_t o = 0;
					if (lane == 0) o = atomicAdd(&s_emit_n, (uint32_t)__popcll(vote));
					o = __builtin_amdgcn_readfirstlane(o);
					if (seen) {
						const RecW<3> r = rec_load<3>(recs + (uint64_t)lc[li] * 3);
						ho_base[o + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(vote >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vote, 0u))] = entry3(r.d[0], r.d[1]);
					}
				}
			}
			dirty = ln != 0;
			__syncthreads();
		}
	} else
	if (!ovf_list) {
		// ---- pass A (dense over the list): set every clear bit; the returning OR tells whether another k-mer of this batch got there first
		for (uint32_t li = tid; li < ln; li += BT) {
			const uint32_t w = lb[li], bl = w & 255u, bl64 = bl << 6, um = (w >> 26) & 15u;
			const B3Pos b = b3_positions((w >> 8) & 511u, (w >> 17) & 511u);
			uint32_t o0 = 0, o1 = 0, o2 = 0, o3 = 0;
			if (um & 1u) o0 = atomicOr(b3_wordp(region, bl64, b.b0), 1u << (b.b0 & 31u));
			if (um & 2u) o1 = atomicOr(b3_wordp(region, bl64, b.b1), 1u << (b.b1 & 31u));
			if (um & 4u) o2 = atomicOr(b3_wordp(region, bl64, b.b2), 1u << (b.b2 & 31u));
			if (um & 8u) o3 = atomicOr(b3_wordp(region, bl64, b.b3), 1u << (b.b3 & 31u));
			uint32_t cm = (b3_bit(o0, b.b0) | (b3_bit(o1, b.b1) << 1) | (b3_bit(o2, b.b2) << 2) | (b3_bit(o3, b.b3) << 3)) & um;
			if (cm) { // contended bits: file order decides them
				*(volatile uint32_t *)&s_fs_used = 1;
				__hip_atomic_fetch_or(&region[bl * 16u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // (the block's mark for pass B)
				const uint32_t idx = la[li];
				const uint32_t p01 = b.b0 | (b.b1 << 16), p23 = b.b2 | (b.b3 << 16);
				while (cm) {
					const int j = __ffs((int)cm) - 1;
					cm &= cm - 1;
					const uint32_t bj = ((j & 2 ? p23 : p01) >> ((j & 1) << 4)) & 0xffffu;
					if (!fs32_insert(fs, fs_mask, bl * 512u + bj, li, idx, la)) *v_ovf = 1;
				}
			}
		}
	}
	__syncthreads();
#ifdef BFCG_MEASURE
	if (timing) tq[6] = clock64();
#endif
	const bool ovf = ovf_list || s_ovf; // (the same for every thread: nothing writes s_ovf behind this barrier)
	if (COLD && !ovf) ; // (decided and emitted above)
	else if (!ovf) {
		if (s_fs_used) {
			// ---- pass B: every toucher of a bit that has an entry competes for it (this brings in the k-mer that set the bit first in EXECUTION
			// order).  A clear bit without an entry is this k-mer's alone: it is a first setter and not seen -- nothing left to decide.
			// Only k-mers of blocks in which pass A met a contended bit can have such a bit (a bit and its touchers share the block): pass A marks
			// those blocks in their lock byte (bit 0 of the block's first word: positions below 8 are never a k-mer's, bbf.c:37; cleared again before the
			// region goes back), the k-mers of marked blocks -- a third of the list once the filter is warm -- are gathered in a worklist (the free
			// tail of the record-index array), and the probing runs DENSE over that worklist instead of in every wave for a few of its lanes.
			auto compete_entry = [&](uint32_t li) {
				const uint32_t w = lb[li], bl = w & 255u, um = (w >> 26) & 15u;
				const B3Pos b = b3_positions((w >> 8) & 511u, (w >> 17) & 511u);
				const uint32_t q0 = bl * 512u + b.b0, q1 = bl * 512u + b.b1, q2 = bl * 512u + b.b2, q3 = bl * 512u + b.b3;
				const uint32_t c0 = fs[fs32_slot(q0, fs_mask)], c1 = fs[fs32_slot(q1, fs_mask)], c2 = fs[fs32_slot(q2, fs_mask)], c3 = fs[fs32_slot(q3, fs_mask)];
				uint32_t todo = ((uint32_t)(c0 != FS32_EMPTY) | ((uint32_t)(c1 != FS32_EMPTY) << 1) | ((uint32_t)(c2 != FS32_EMPTY) << 2) | ((uint32_t)(c3 != FS32_EMPTY) << 3)) & um;
				bool und = todo == um; // (a home slot that is empty: no entry for that bit)
				if (todo) {
					const uint32_t idx = la[li];
					const uint32_t p01 = b.b0 | (b.b1 << 16), p23 = b.b2 | (b.b3 << 16);
					while (todo) {
						const int j = __ffs((int)todo) - 1;
						todo &= todo - 1;
						const uint32_t bj = ((j & 2 ? p23 : p01) >> ((j & 1) << 4)) & 0xffffu;
						und &= fs32_compete(fs, fs_mask, bl * 512u + bj, li, idx, la);
					}
				}
				if (und) lb[li] = w | B3_UND;
			};
			unsigned short *const wb = lc + ln;
			const uint32_t wb_cap = P.list_cap - ln;
			for (uint32_t li0 = 0; li0 < ln; li0 += BT) {
				const uint32_t li = li0 + tid;
				bool want = false;
				if (li < ln) {
					want = (region[(lb[li] & 255u) * 16u] & 1u) != 0;
					if (!want && A.seen_out) A.seen_out[la[li]] = 1;
				}
				const unsigned long long vote = __ballot(want);
				if (vote) {
					uint32_t o = 0;
					if (lane == 0) o = atomicAdd(&s_wb_n, (uint32_t)__popcll(vote));
					o = __builtin_amdgcn_readfirstlane(o);
					if (want) {
						const uint32_t slot = o + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(vote >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vote, 0u));
						if (slot < wb_cap) wb[slot] = (unsigned short)li;
						else compete_entry(li); // (no room in the worklist: at once; pass C then walks the whole list)
					}
				}
			}
			__syncthreads();
			const uint32_t wb_all = s_wb_n, wb_n = wb_all < wb_cap ? wb_all : wb_cap;
			for (uint32_t t = tid; t < wb_n; t += BT) compete_entry(wb[t]);
			__syncthreads();
			// ---- pass C: seen iff an earlier k-mer of the batch is the first setter of each of its clear bits
			const bool wl_ok = wb_all <= wb_cap;
			const uint32_t cn = wl_ok ? wb_n : ln;
			for (uint32_t t0 = 0; t0 < cn; t0 += BT) {
				const uint32_t t = t0 + tid;
				uint32_t li = 0;
				bool seen = false;
				if (t < cn) {
					li = wl_ok ? (uint32_t)wb[t] : t;
					const uint32_t w = lb[li];
					if (w & B3_UND) {
						const uint32_t bl = w & 255u, um = (w >> 26) & 15u;
						const B3Pos b = b3_positions((w >> 8) & 511u, (w >> 17) & 511u);
						bool first = false;
						if (um & 1u) first |= fs32_lookup(fs, fs_mask, bl * 512u + b.b0) == li;
						if (um & 2u) first |= fs32_lookup(fs, fs_mask, bl * 512u + b.b1) == li;
						if (um & 4u) first |= fs32_lookup(fs, fs_mask, bl * 512u + b.b2) == li;
						if (um & 8u) first |= fs32_lookup(fs, fs_mask, bl * 512u + b.b3) == li;
						seen = !first;
					}
					if (A.seen_out) A.seen_out[la[li]] = seen ? 2 : 1;
				}
				const unsigned long long vote = __ballot(seen);
				if (vote) { // (wave-uniform) the seen k-mers of this wave take consecutive slots behind what pass 1 emitted
					uint32_t o = 0;
					if (lane == 0) o = atomicAdd(&s_emit_n, (uint32_t)__popcll(vote));
					o = __builtin_amdgcn_readfirstlane(o);
					if (seen) {
						const RecW<3> r = rec_load<3>(recs + (uint64_t)lc[li] * 3);
						ho_base[o + (uint32_t)__popcll(vote & ((1ULL << lane) - 1))] = entry3(r.d[0], r.d[1]);
					}
				}
			}
			for (uint32_t i = tid; i < (1u << P.R); i += BT) region[i * 16u] &= ~0xffu; // the blocks' marks: the lock byte goes back as it came, zero
		} else if (A.seen_out) {
			for (uint32_t li = tid; li < ln; li += BT) A.seen_out[la[li]] = 1; // nobody shares a clear bit: every listed k-mer is a first setter
		}
		dirty = ln != 0;
		__syncthreads();
	} else {
		// ---- slow path (as in k_bloom): first-setter table in HBM, one locked slice of the pool
		const int nh = 4;
		uint64_t want = (uint64_t)n * nh * 2;
		const uint64_t lim = (uint64_t)(1u << P.R) * 512 * 2;
		if (want > lim) want = lim;
		uint32_t cap = 1024; while (cap < want) cap <<= 1;
		// (pass 1's emits are discarded: the region starts again from the pre-batch state, every seen k-mer is emitted below)
		if (tid == 0) {
			uint32_t sl = f & (A.pool_slices - 1);
			while (atomicCAS(&A.pool[sl], 0ULL, 1ULL) != 0ULL) sl = (sl + 1) & (A.pool_slices - 1);
			s_pool_off = sl; s_emit_n = 0;
			atomicAdd(&A.stats[ST_SLOW_BUCKETS], 1ULL);
		}
		{
			const uint4 *src = reinterpret_cast<const uint4 *>(g_region);
			uint4 *dst = reinterpret_cast<uint4 *>(region);
			for (uint32_t i = tid; i < region_dw / 4; i += BT) dst[i] = src[i];
		}
		__syncthreads();
		unsigned long long *gfs = A.pool + A.pool_slices + (uint64_t)s_pool_off * lim;
		for (uint32_t i = tid; i < cap; i += BT) gfs[i] = FS_EMPTY;
		__threadfence();
		__syncthreads();
		const uint32_t gmask = cap - 1;
		for (uint32_t i = tid; i < n; i += BT) {
			const RecW<3> r = rec_load<3>(recs + (uint64_t)i * 3);
			const uint32_t w = addr3(r.d[0], r.d[1]), bl = w & 255u;
			uint32_t z = (w >> 8) & 511u;
			for (int j = 0; j < nh; ++j) {
				const uint32_t b = bloom_next(z, w >> 17);
				if (!b3_bit(*b3_wordp(region, bl << 6, b), b)) fs_insert<true>(gfs, gmask, bl * 512 + b, r.d[2], gmask);
			}
		}
		__threadfence();
		__syncthreads();
		for (uint32_t i0 = 0; i0 < n; i0 += BT) {
			const uint32_t i = i0 + tid;
			bool seen = false;
			RecW<3> r; r.d[0] = r.d[1] = r.d[2] = 0;
			if (i < n) {
				r = rec_load<3>(recs + (uint64_t)i * 3);
				const uint32_t w = addr3(r.d[0], r.d[1]), bl = w & 255u;
				uint32_t z = (w >> 8) & 511u; bool first = false, unresolved = false;
				for (int j = 0; j < nh; ++j) {
					uint32_t b = bloom_next(z, w >> 17), fi;
					if (fs_lookup<true>(gfs, gmask, bl * 512 + b, fi)) { // has an entry <=> was clear before the batch
						unresolved = true; first |= (fi == r.d[2]);
						atomicOr(b3_wordp(region, bl << 6, b), 1u << (b & 31));
					}
				}
				seen = !unresolved || !first;
				if (A.seen_out) A.seen_out[r.d[2]] = seen ? 2 : 1;
			}
			const unsigned long long vote = __ballot(seen);
			if (vote) {
				uint32_t o = 0;
				if (lane == 0) o = atomicAdd(&s_emit_n, (uint32_t)__popcll(vote));
				o = __builtin_amdgcn_readfirstlane(o);
				if (seen) ho_base[o + (uint32_t)__popcll(vote & ((1ULL << lane) - 1))] = entry3(r.d[0], r.d[1]);
			}
		}
		__syncthreads();
		if (tid == 0) { __threadfence(); atomicExch(&A.pool[s_pool_off], 0ULL); } // release the slice
	}
#ifdef BFCG_MEASURE
	if (timing) tq[3] = clock64();
#endif
	if (dirty) { // write the region back
		uint4 *dst = reinterpret_cast<uint4 *>(g_region);
		const uint4 *src = reinterpret_cast<const uint4 *>(region);
		for (uint32_t i = tid; i < region_dw / 4; i += BT) dst[i] = src[i];
	}
#ifdef BFCG_MEASURE
	if (timing) tq[4] = clock64();
#endif
	if (tid == 0) { // every seen k-mer was emitted exactly once: the log's fill is the count
		const uint32_t ns = s_emit_n;
		if (ns) atomicAdd(&A.stats[ST_SEEN], (unsigned long long)ns);
		if (ho_log) { A.ho_cur[f] = ho_cur0 + ns; A.ho_mark[f] = ho_cur0 + ns; }
		else if (A.agg_cnt) A.agg_cnt[f] = ns;
	}
#ifdef BFCG_MEASURE
	if (timing) {
		tq[5] = clock64();
		for (int t = 0; t < 5; ++t) atomicAdd(&A.stats[10 + t], (unsigned long long)(tq[t + 1] - tq[t]));
		atomicAdd(&A.stats[15], (unsigned long long)(tq[6] - tq[2])); // pass A alone (part of slot 12)
	}
#endif
	(void)s_pad3;
}

namespace bfcg {

hipError_t set_bloom3_lds_attr(int lds)
{
	hipError_t e = hipFuncSetAttribute((const void *)k_bloom3<512, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
	if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_bloom3<512, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
	if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_bloom3<512, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
	if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_bloom3<512, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
	return e;
}

// PF = 4: 80 registers, three workgroups per CU at the 53 KB of LDS a full-size list takes; PF = 2: 47 registers, four per CU where the LDS
// footprint allows it (a shorter list: bfcg_ctx.hip decides per batch)
void run_bloom3(const KParams &P, const BloomArgs &A, int nfine, size_t lds, hipStream_t st)
{
	static int pf = 0;
	if (!pf) { const char *e = getenv("BFCG_B3_PF"); pf = e && atoi(e) == 2 ? 2 : e && atoi(e) == 4 ? 4 : 1; } // (1: by the batch)
	if (P.b3_cold && P.b3_warm) hipLaunchKernelGGL((k_bloom3<512, 2, true>), dim3(nfine), dim3(512), lds, st, P, A); // (the walk for a warm batch: the short list, four workgroups per CU)
	else if (P.b3_cold) hipLaunchKernelGGL((k_bloom3<512, 4, true>), dim3(nfine), dim3(512), lds, st, P, A);
	else if (pf == 2 || (pf == 1 && P.b3_warm)) hipLaunchKernelGGL((k_bloom3<512, 2, false>), dim3(nfine), dim3(512), lds, st, P, A);
	else hipLaunchKernelGGL((k_bloom3<512, 4, false>), dim3(nfine), dim3(512), lds, st, P, A);
}

} // namespace bfcg

// ------------------------------------------------------------------------------------------
// k_bloom3fm (round 5): the bloom insert of `bfc -1`'s count pass (filter mode, count.c:67-68: a k-mer seen before goes into the SECOND filter) for
// 16-byte records whose bloom address is a bit field of their first two words -- k >= bf_shift + 9, so block, h1 and h2 are all bits of y0
// (kmer.h:87; BASELINE config c5: k = 51, -b37) -- in k_bloom3's structure and with the block walk for EVERY batch:
//   pass 1   every record: address, four bit positions, four LDS reads; all set before the batch => seen whatever the order: its bits are ORed
//            into the second filter's slice (same hash, same block: it sits in LDS beside the first, count.c:67-68); else a list entry
//            (file index | block, h1, h2) and the block's counter;
//   walk     the list ordered by (block, file index), one lane per block doing what bfc_bf_insert does (bbf.c:33-44): seen iff every bit is set
//            -- then the second filter's bits --, else set them.
// Nothing is handed over.  A region whose list does not fit is not sent to a slower path: the walk is exact over any prefix of the file order,
// so the region's records are taken in ROUNDS of file-index ranges (halved until a range's list fits) -- no HBM pool, no first-setter table.
// Round 1-4's k_bloom<.., FM> served this mode before (c5's count pass: 1.52 of 3.10 s in it, 24.6 ps per k-mer against 10 for k_bloom3 on c4).
struct Dec4 { int up, iw, is; uint32_t rmask; };
__device__ __forceinline__ Dec4 dec4_geom(const KParams &P)
{
	Dec4 g;
	g.up = P.rec_n ? P.rec_lo : P.bf_shift - 9;
	const int io = (P.k - P.rec_n) + P.k + 1; // first bit of the file index inside the 128-bit record (Rec<4>::pack)
	g.iw = io >> 5; g.is = io & 31; g.rmask = (1u << P.R) - 1u;
	return g;
}
template <int BT, int PF>
__global__ __launch_bounds__(BT, 6) void k_bloom3fm(KParams P, BloomArgs A)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	__shared__ uint32_t s_list_n, s_seen;
	const uint32_t f = blockIdx.x;
	if (batch_poisoned(A)) return;
	uint32_t rs, n;
	region_list(A, f, rs, n);
	if (n == 0) return;
	A.stats += (size_t)(f & (ST_SLOTS - 1)) * ST_N;
	const uint32_t region_dw = 16u << P.R, nblk = 1u << P.R;
	unsigned char *sp = smem;
	unsigned int *region = reinterpret_cast<unsigned int *>(sp); sp += (size_t)region_dw * 4;
	unsigned int *region_hi = reinterpret_cast<unsigned int *>(sp); sp += (size_t)region_dw * 4;
	unsigned int *bcnt = reinterpret_cast<unsigned int *>(sp), *boff = bcnt + nblk; sp += (size_t)(2 * nblk + 4) * 4;
	unsigned int *la = reinterpret_cast<unsigned int *>(sp); sp += (size_t)P.list_cap * 4;   // file-order index
	unsigned int *lb = reinterpret_cast<unsigned int *>(sp); sp += (size_t)P.list_cap * 4;   // block | h1 << 8 | h2 << 17 (| B3_SEEN)
	unsigned short *ord = reinterpret_cast<unsigned short *>(sp);                            // the list's entries in (block, file index) order
	const uint32_t *recs = A.recs + (uint64_t)rs * 4;
	unsigned int *g_region = reinterpret_cast<unsigned int *>(A.bloom) + (uint64_t)f * region_dw;
	unsigned int *g_region_hi = reinterpret_cast<unsigned int *>(A.bloom_hi) + (uint64_t)f * region_dw;
	const Dec4 D = dec4_geom(P);
	const int tid = threadIdx.x, lane = tid & 63;
	// block (R <= 8 bits) | h1 << 8 | h2 << 17 from the record's first two words: y0' bits [0, R) and [up, up + 18)
	auto addr4 = [&](uint32_t d0, uint32_t d1) -> uint32_t {
		const uint32_t hh = __builtin_amdgcn_alignbit(d1, d0, D.up);
		uint32_t h2 = (hh >> 9) & 511u;
		h2 |= (uint32_t)((h2 & 31u) == 0); // bbf.c:33
		return (d0 & D.rmask) | ((hh & 511u) << 8) | (h2 << 17);
	};
	auto idx4 = [&](const RecW<4> &r) -> uint32_t { // bits [io, io + 32) of the 128-bit record
		const uint32_t a = D.iw == 0 ? r.d[0] : D.iw == 1 ? r.d[1] : D.iw == 2 ? r.d[2] : r.d[3], b = D.iw == 0 ? r.d[1] : D.iw == 1 ? r.d[2] : r.d[3];
		return D.is ? __builtin_amdgcn_alignbit(b, a, D.is) : a;
	};
	auto set_hi = [&](uint32_t bl64, const B3Pos &b) { // count.c:68 on the slice in LDS (two positions may share a word: ORs)
		__hip_atomic_fetch_or(b3_wordp(region_hi, bl64, b.b0), 1u << (b.b0 & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		__hip_atomic_fetch_or(b3_wordp(region_hi, bl64, b.b1), 1u << (b.b1 & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		__hip_atomic_fetch_or(b3_wordp(region_hi, bl64, b.b2), 1u << (b.b2 & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		__hip_atomic_fetch_or(b3_wordp(region_hi, bl64, b.b3), 1u << (b.b3 & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	};
	{ // stage both slices (16-byte loads)
		const uint4 *src = reinterpret_cast<const uint4 *>(g_region), *src2 = reinterpret_cast<const uint4 *>(g_region_hi);
		uint4 *dst = reinterpret_cast<uint4 *>(region), *dst2 = reinterpret_cast<uint4 *>(region_hi);
		for (uint32_t i = tid; i < region_dw / 4; i += BT) { dst[i] = src[i]; dst2[i] = src2[i]; }
	}
	// rounds of file-index ranges: range r of 2^(32 - sh) = [r << sh, (r + 1) << sh); one round takes everything
	uint32_t r = 0, sh = 32, seen_total = 0;
	bool dirty = false;
	const uint32_t cap = P.list_cap < (uint32_t)(B3_COLD_NR * BT) ? P.list_cap : (uint32_t)(B3_COLD_NR * BT);
	for (;;) {
		for (uint32_t i = tid; i < 2 * nblk + 4; i += BT) bcnt[i] = 0;
		if (tid == 0) { s_list_n = 0; s_seen = 0; }
		__syncthreads(); // (first round: the slices are staged)
		// ---- pass 1 over the range's records: classify against the region as the earlier ranges left it
		uint32_t my_seen = 0;
		RecW<4> rec[PF], nxt[PF]; // (the records of the round after this one are on their way while this one is classified)
#pragma unroll
		for (int u = 0; u < PF; ++u) {
			const uint32_t i = tid + u * BT;
			rec[u].d[0] = rec[u].d[1] = rec[u].d[2] = rec[u].d[3] = 0; nxt[u] = rec[u];
			if (i < n) rec[u] = rec_load<4>(recs + (uint64_t)i * 4);
			if (i + BT * PF < n) nxt[u] = rec_load<4>(recs + (uint64_t)(i + BT * PF) * 4);
		}
		for (uint32_t base = 0; base < n; base += BT * PF) {
			uint32_t pk[PF], um[PF], ix[PF];
			bool act[PF];
#pragma unroll
			for (int u = 0; u < PF; ++u) {
				ix[u] = idx4(rec[u]);
				act[u] = base + tid + u * BT < n && (sh == 32 || (ix[u] >> sh) == r);
				pk[u] = addr4(rec[u].d[0], rec[u].d[1]);
				const B3Pos b = b3_positions((pk[u] >> 8) & 511u, pk[u] >> 17);
				const uint32_t bl64 = (pk[u] & 255u) << 6;
				const uint32_t w0 = *b3_wordp(region, bl64, b.b0), w1 = *b3_wordp(region, bl64, b.b1), w2 = *b3_wordp(region, bl64, b.b2), w3 = *b3_wordp(region, bl64, b.b3);
				um[u] = (b3_bit(w0, b.b0) | (b3_bit(w1, b.b1) << 1) | (b3_bit(w2, b.b2) << 2) | (b3_bit(w3, b.b3) << 3)) ^ 15u;
				if (act[u] && um[u] == 0) { // every bit was set before this range: seen, whatever the order inside it
					set_hi(bl64, b); ++my_seen;
					if (A.seen_out) A.seen_out[ix[u]] = 2;
				}
			}
			unsigned long long ml[PF];
			uint32_t tot_l = 0;
#pragma unroll
			for (int u = 0; u < PF; ++u) { ml[u] = __ballot(act[u] && um[u] != 0); tot_l += (uint32_t)__popcll(ml[u]); }
			uint32_t o_l = 0;
			if (lane == 0 && tot_l) o_l = atomicAdd(&s_list_n, tot_l);
			o_l = __builtin_amdgcn_readfirstlane(o_l);
#pragma unroll
			for (int u = 0; u < PF; ++u) {
				if (act[u] && um[u] != 0) {
					const uint32_t li = o_l + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(ml[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ml[u], 0u));
					if (li < cap) { la[li] = ix[u]; lb[li] = pk[u]; __hip_atomic_fetch_add(&bcnt[pk[u] & 255u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
				}
				o_l += (uint32_t)__popcll(ml[u]);
			}
			if (base + BT * PF < n) {
#pragma unroll
				for (int u = 0; u < PF; ++u) {
					rec[u] = nxt[u];
					const uint32_t i = base + 2 * BT * PF + tid + u * BT;
					if (i < n) nxt[u] = rec_load<4>(recs + (uint64_t)i * 4);
				}
			}
		}
		for (int o = 32; o; o >>= 1) my_seen += __shfl_down(my_seen, o);
		if (lane == 0 && my_seen) atomicAdd(&s_seen, my_seen);
		__syncthreads();
		const uint32_t ln = s_list_n;
		if (ln > cap) { // too many for one walk: the lower half of the range first (the region itself is as it was; the second filter's ORs were right and stay)
			__syncthreads(); // (everybody has read the counters before the next round clears them)
			--sh; r <<= 1;   // (a range of ONE index holds one k-mer: this ends)
			continue;
		}
		seen_total += s_seen;
		if (ln) {
			if (tid < 64) { // block offsets: exclusive scan of the 2^R counters (four blocks per lane at most)
				const uint32_t per = (nblk + 63u) >> 6, b0 = (uint32_t)tid * per;
				uint32_t v[4] = {0, 0, 0, 0}, sum = 0;
#pragma unroll
				for (int t = 0; t < 4; ++t) if ((uint32_t)t < per && b0 + t < nblk) { v[t] = bcnt[b0 + t]; sum += v[t]; }
				uint32_t inc = sum;
#pragma unroll
				for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(inc, o); if (lane >= o) inc += x; }
				uint32_t run = inc - sum;
#pragma unroll
				for (int t = 0; t < 4; ++t) if ((uint32_t)t < per && b0 + t < nblk) { boff[b0 + t] = run; bcnt[b0 + t] = run; run += v[t]; }
				if (tid == 63) boff[nblk] = inc;
			}
			__syncthreads();
			for (uint32_t li = tid; li < ln; li += BT) ord[atomicAdd(&bcnt[lb[li] & 255u], 1u)] = (unsigned short)li;
			__syncthreads();
			uint32_t mv[B3_COLD_NR];
#pragma unroll
			for (int t = 0; t < B3_COLD_NR; ++t) {
				const uint32_t pos = (uint32_t)tid + (uint32_t)t * BT;
				mv[t] = 0;
				if (pos < ln) {
					const uint32_t li = ord[pos], x = la[li], bl = lb[li] & 255u, s0 = boff[bl], s1 = boff[bl + 1];
					uint32_t rk = 0;
					for (uint32_t j = s0; j < s1; ++j) rk += (uint32_t)(la[ord[j]] < x);
					mv[t] = ((s0 + rk) << 16) | li;
				}
			}
			__syncthreads();
#pragma unroll
			for (int t = 0; t < B3_COLD_NR; ++t) { const uint32_t pos = (uint32_t)tid + (uint32_t)t * BT; if (pos < ln) ord[mv[t] >> 16] = (unsigned short)mv[t]; }
			__syncthreads();
			uint32_t w_seen = 0;
			if ((uint32_t)tid < nblk) { // the walk: lane b takes block b's k-mers in file order (bbf.c:33-44, count.c:59-68)
				const uint32_t bl64 = (uint32_t)tid << 6, s0 = boff[tid], s1 = boff[tid + 1];
				uint32_t li_n = s0 < s1 ? ord[s0] : 0u, w_n = s0 < s1 ? lb[li_n] : 0u;
				for (uint32_t j = s0; j < s1; ++j) {
					const uint32_t li = li_n, w = w_n;
					if (j + 1 < s1) { li_n = ord[j + 1]; w_n = lb[li_n]; }
					const B3Pos b = b3_positions((w >> 8) & 511u, (w >> 17) & 511u);
					unsigned int *p0 = b3_wordp(region, bl64, b.b0), *p1 = b3_wordp(region, bl64, b.b1), *p2 = b3_wordp(region, bl64, b.b2), *p3 = b3_wordp(region, bl64, b.b3);
					const uint32_t w0 = *p0, w1 = *p1, w2 = *p2, w3 = *p3;
					const uint32_t clr = (b3_bit(w0, b.b0) | (b3_bit(w1, b.b1) << 1) | (b3_bit(w2, b.b2) << 2) | (b3_bit(w3, b.b3) << 3)) ^ 15u;
					if (clr == 0) { set_hi(bl64, b); ++w_seen; }
					else {
						if (clr & 1u) __hip_atomic_fetch_or(p0, 1u << (b.b0 & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
						if (clr & 2u) __hip_atomic_fetch_or(p1, 1u << (b.b1 & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
						if (clr & 4u) __hip_atomic_fetch_or(p2, 1u << (b.b2 & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
						if (clr & 8u) __hip_atomic_fetch_or(p3, 1u << (b.b3 & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					}
					if (A.seen_out) A.seen_out[la[li]] = clr == 0 ? 2 : 1;
				}
			}
			if (tid == 0) s_seen = 0;
			__syncthreads();
			for (int o = 32; o; o >>= 1) w_seen += __shfl_down(w_seen, o);
			if (lane == 0 && w_seen) atomicAdd(&s_seen, w_seen);
			__syncthreads();
			seen_total += s_seen;
			dirty = true;
		}
		// the next range: the sibling if this was a lower half, else up until a range that is one, and its sibling
		if (sh == 32) break;
		while (sh < 32 && (r & 1u)) { r >>= 1; ++sh; }
		if (sh == 32) break;
		r |= 1u;
		__syncthreads();
	}
	if (dirty) {
		uint4 *dst = reinterpret_cast<uint4 *>(g_region);
		const uint4 *src = reinterpret_cast<const uint4 *>(region);
		for (uint32_t i = tid; i < region_dw / 4; i += BT) dst[i] = src[i];
	}
	if (seen_total) {
		uint4 *dst = reinterpret_cast<uint4 *>(g_region_hi);
		const uint4 *src = reinterpret_cast<const uint4 *>(region_hi);
		for (uint32_t i = tid; i < region_dw / 4; i += BT) dst[i] = src[i];
		if (tid == 0) atomicAdd(&A.stats[ST_SEEN], (unsigned long long)seen_total);
	}
}

namespace bfcg {
hipError_t set_bloom3fm_lds_attr(int lds) { return hipFuncSetAttribute((const void *)k_bloom3fm<512, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); }
void run_bloom3fm(const KParams &P, const BloomArgs &A, int nfine, size_t lds, hipStream_t st)
{
	hipLaunchKernelGGL((k_bloom3fm<512, 2>), dim3(nfine), dim3(512), lds, st, P, A);
}
} // namespace bfcg
