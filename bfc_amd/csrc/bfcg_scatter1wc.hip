// bfcg_scatter1wc.hip -- k_scatter1_wc (round 5): level 1 of the partition through WRITE-COMBINING BUFFERS in LDS.
//
// What it replaces on the default path: k_scatter1's tile skeleton (bfcg_kernels.hip) -- rank every record of a tile by a returning LDS add,
// scan the bucket counters, reserve a place per (tile, bucket) run, stage the tile in bucket order, copy it out run by run: five barriers per
// tile, ~61 of the kernel's 175 lane-instructions per k-mer, and runs of ~6 records = 77 bytes that leave as partial lines (WRITE_SIZE 1.54 x the
// records; profiles/round5_c3.md).  Semantics are the same: count.c:72-89 (rolling k-mers, quality mask) and kmer.h:79-88 (the canonical hash)
// per position -> a 12-byte record (y0 minus its bucket bits | y1 | quality flag | file index; 16 bytes for 37 <= k <= 52 at 2^10 buckets: config
// c5's k = 51) in the slab of its level-1 bucket; the order inside a slab is irrelevant because every record carries its file index.
//
// Here a persistent workgroup owns, per bucket, ONE buffer of CAP records in LDS: 2^12 records (48 KiB) with 512 threads and two workgroups per
// CU -- buffers of 8 records at 2^9 buckets, config c3 --, 2^13 (96 KiB; 128 KiB of 16-byte records) with 1024 threads and one (2^10 buckets,
// configs c4's and c5's -b37; BFCG_S1_WC_BT=1024).
// A record takes the slot a returning LDS add on the bucket's fill hands out and is written straight into the buffer; a full buffer leaves as one
// CHUNK of CAP x 12 (16) contiguous bytes (whole 32-byte sectors) into room reserved from the slab's cursor a GROUP of four chunks ahead.  A round
// (one tile of four positions per thread; three with 16-byte records) has THREE barriers:
//
//   P1  every thread: the 16-byte pieces it read in the last P3 are stored; the puts that found their buffer full in the previous round (slot -
//       CAP: the buffer was flushed meanwhile); then the k-mers of its four positions: hash, record, slot = fill[bucket]++; slot < CAP: put; slot <
//       2 CAP: keep it in registers for the next P1; beyond that (one bucket drew more than two buffers' worth in one round: 3 in 10 000 (bucket,
//       round) pairs on hashed k-mers) the lane reserves a chunk of its own and stores the record with CAP - 1 dead records behind it
//   --- barrier A
//   P2  roles by lane (with 1024 threads and 2^9 buckets they fall on different waves; with 512 threads, or 2^10 buckets, every thread owns a
//       bucket and has its other roles beside that):
//       OWNERS      thread OWN0 + i owns bucket i: fill >= CAP -> (bucket, destination) into the round's list (one LDS atomic per wave), fill -=
//                   CAP, the chunk pointer moves on -- into the next group, taken from LDS, when the current one is used up; the next but one is
//                   asked for as soon as a group is begun
//       RESERVERS   waves RSV0.., one per round in turn: a bucket whose owner asked for its next group (a word in LDS) gets it by a returning
//                   atomic on the slab's cursor; the answer is published in LDS at the wave's next turn (what it waits for then is a turn old).
//                   An owner that reserved for itself waited for the answer BEHIND its wave's fresh stores -- loads, returning atomics and
//                   stores share one in-order counter (vmcnt) on this chip --: 3 000 - 4 000 cycles of a round in the first cut
//       LOADERS     threads 0 .. NCH - 1: the NEXT tile's three base planes from the bases requested a round ago, then the request for the tile
//                   after it; threads QL0 .. QL0 + NCH - 1 the same for the quality plane; thread DRAW_T settles the tile draw of a round ago
//   --- barrier B
//   P3  every thread: up to three 16-byte pieces of the listed chunks from the buffers into registers (nothing but LDS reads)
//   --- barrier C: the flushed buffers are free
//
// The pieces are stored at the top of the next P1, where a wave that the memory pipeline holds up at the issue of its stores leaves its SIMD to
// the other waves' hashing (stored from inside P2 by the owner waves, everyone else at the barrier: 2 400 of a round's 14 800 cycles).  The slabs,
// their cursors, the dead records (all ones) in what was reserved and not filled, the overflow flag and the statistics are exactly k_scatter1's
// (OnePass, bfcg_k1.h): k_seg_setup and level 2 read this kernel's output as they read that one's.  A tile belongs to an XCD for the DRAW (own
// counter first, then the others'), but every record of a workgroup goes to its home XCD's slabs.  profiles/round5_k_scatter1_wc.md has the
// phase clocks, the ablations and what the compiler did on the way.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include "kmer_dev.h"
#include "bfcg_internal.h"
#include "bfcg_dev.h"
#include "bfcg_k1.h"

using namespace bfcg;

constexpr uint32_t WC_NONE = 0xffffffffu;

// W: the k-mer word (one 32-bit word up to k = 32, else halves); RW: dwords per record (3, or 4: k up to 52 at 2^10 buckets -- config c5); CAPL, NBL:
// log2 records per buffer, log2 buckets; BT: threads (512: two workgroups per CU, 1024: one); S: positions per thread and round (a round is a
// tile of S BT positions); KC: k at compile time or 0.
template <typename W, int RW, int CAPL, int NBL, int BT, int S, int KC>
__global__ __launch_bounds__(BT, 4) void k_scatter1_wc(KParams P, const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual, int64_t n_pos,
                                                                       uint32_t *__restrict__ out, OnePass OP, uint32_t G)
{
	constexpr int TILE = BT * S;
	constexpr uint32_t CAP = 1u << CAPL, NB = 1u << NBL, PIECES = CAP * RW / 4; // records per buffer, buckets, 16-byte pieces of a chunk
	static_assert(RW == 3 || RW == 4, "12- or 16-byte records");
	constexpr int PW = (TILE + 64) / 32 + 2, NC16 = (TILE + 64) / 16, NCH = NC16 + 1;
	// Roles.  Thread OWN0 + i owns bucket i; threads 0 .. NCH - 1 build the base planes, QL0 .. QL0 + NCH - 1 the quality plane; waves RSV0 .. RSV0 +
	// NRW - 1 reserve.  With 1024 threads and 512 buckets the loaders, the reservers and the owners are different waves (the owners' first
	// ones also build the quality plane); with 512 threads every thread owns a bucket and has its other roles beside that.
	constexpr int OWN0 = BT - (int)NB, QL0 = BT / 2, RSV0 = BT == 512 ? 7 : 5, NRW = BT == 512 ? 1 : 3;
	constexpr int DRAW_T = BT == 512 ? 3 * WAVE : 4 * WAVE + 40; // the thread that draws the tiles: a lane of a wave without a loader's chain (512 threads: wave 3; 1024: a spare lane of wave 4)
	static_assert(OWN0 >= 0 && NCH <= QL0 && QL0 + NCH <= BT && (NCH + WAVE - 1) / WAVE <= RSV0 && NB % (2 * WAVE) == 0 && CAP >= 8, "roles by wave");
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_wc[];
	uint32_t *buf = reinterpret_cast<uint32_t *>(smem_wc); // NB buffers of CAP records of 3 dwords
	__shared__ uint32_t fill[NB];
	__shared__ uint2 wl[NB];                        // this round's flushes (bucket, destination in 16-byte units), nq[round & 1] of them
	__shared__ uint32_t nq[2];
	__shared__ uint32_t planes[2 * 4 * PW];
	__shared__ uint32_t s_draw[4];
	__shared__ uint32_t s_draw_a;                   // (thread 0) XCDs whose tile counters this workgroup has found exhausted, counted from its own
	// owner -> reservers: rq[b] = 1: bucket b's next group, please (the reserver that takes the request up clears it); reservers -> owner: res[b] = the
	// group's base in the slab, then rdy[b] = 1 (the owner clears it when it takes the group).  An owner has at most one request open.
	__shared__ uint32_t rq[NB], rdy[NB], res[NB];
	const int tid = threadIdx.x, lane = tid & (WAVE - 1);
	const uint32_t RES = G << CAPL;                  // records per reservation
	const uint32_t home = blockIdx.x & 7u;
	const int64_t n_tiles = (n_pos + TILE - 1) / TILE;
	const int b1_shift = P.R + P.F2;
	const Pack3 PK = pack3_geom(P);
	const W m = kmask<W>(P.k);

	// ---- tile draws (k_scatter1's: one counter per XCD behind the cursors; own XCD first)
	uint32_t *const tile_ctr = OP.cursor + (size_t)8 * NB * 32;
#define draw_a s_draw_a
	if (tid == DRAW_T) s_draw_a = 0;
	auto draw_issue = [&]() -> uint32_t { return draw_a < 8u ? atomicAdd(&tile_ctr[((blockIdx.x + draw_a) & 7u) * 4u], 1u) : 0u; };
	auto draw_settle = [&](uint32_t t) -> uint32_t {
		while (draw_a < 8u) {
			const uint32_t x = (blockIdx.x + draw_a) & 7u;
			if ((int64_t)t * 8 + x < n_tiles) return t * 8u + x;
			if (++draw_a < 8u) {
				t = atomicAdd(&tile_ctr[((blockIdx.x + draw_a) & 7u) * 4u], 1u);
				__builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0), here: otherwise the compiler's wait for this rare answer lands in front of the first reuse of its register -- inside P1's hashing, every round
			}
		}
		return WC_NONE;
	};
	uint32_t draw = 0;
	if (tid == DRAW_T) {
		s_draw[0] = draw_settle(draw_issue()); s_draw[1] = draw_settle(draw_issue()); s_draw[2] = draw_settle(draw_issue());
		draw = draw_issue();
	}
	for (int i = tid; i < (int)NB; i += BT) { fill[i] = 0; rq[i] = 0; rdy[i] = 0; }
	if (tid < 2) nq[tid] = 0;
	__syncthreads();
	uint32_t t_cur = s_draw[0], t_next = s_draw[1], t_pf = s_draw[2];
	if (t_cur == WC_NONE) return; // (nothing was reserved yet)

	// ---- the input side (k_scatter1's): aligned 16-byte blocks at any offset, planes in block-stream coordinates
	const int mis = (int)((uintptr_t)seq & 15);
	const uint8_t *const sb = seq - mis, *const qb = qual ? qual - mis : nullptr;
	const int64_t v_end = n_pos + mis, v_last = (v_end - 1) & ~(int64_t)15;
	// A 16-byte block of the tile is the work of TWO lanes: thread c < NCH turns its bases into three plane pieces, thread QL0 + c its qualities
	// into the fourth (one lane for both was a chain of ~300 instructions in five waves: 2 300 cycles of every round's P2).
	const bool ld_b = tid < NCH, ld_q = qual != nullptr && tid >= QL0 && tid < QL0 + NCH;
	uint4 pf = make_uint4(0, 0, 0, 0);
	auto prefetch = [&](uint32_t t) { // (threads with ld_b or ld_q)
		const int64_t v = (int64_t)t * TILE - 64 + (int64_t)(ld_b ? tid : tid - QL0) * 16, at = v < 0 ? 0 : v > v_last ? v_last : v;
		pf = *reinterpret_cast<const uint4 *>((ld_b ? sb : qb) + at);
	};
	auto ragged = [&](int64_t v, uint32_t &k0, uint32_t &k1, uint32_t &k2, uint32_t &k3) -> bool { // a block at the ragged ends of the batch: byte masks of what belongs to it
		if (!(v < mis || v + 16 > v_end)) return false;
		const int lo = v >= mis ? 0 : mis - v >= 16 ? 16 : (int)(mis - v), hi = v_end - v >= 16 ? 16 : v_end - v <= 0 ? 0 : (int)(v_end - v);
		const uint32_t bm = hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
		auto keep = [&](int d) { return (((bm >> (4 * d)) & 0xFu) * 0x00204081u & 0x01010101u) * 0xFFu; };
		k0 = keep(0); k1 = keep(1); k2 = keep(2); k3 = keep(3);
		return true;
	};
	auto planes_b = [&](uint32_t t, uint32_t *pl) { // threads with ld_b: low bit, high bit, not-ACGT (and an all-ones quality piece for FASTA)
		const int c = tid;
		const int64_t v = (int64_t)t * TILE - 64 + (int64_t)c * 16;
		uint4 s4 = pf;
		uint32_t k0, k1, k2, k3;
		if (ragged(v, k0, k1, k2, k3)) { // bytes outside the batch read as separators
			s4.x = (s4.x & k0) | (0x0a0a0a0au & ~k0); s4.y = (s4.y & k1) | (0x0a0a0a0au & ~k1); s4.z = (s4.z & k2) | (0x0a0a0a0au & ~k2); s4.w = (s4.w & k3) | (0x0a0a0a0au & ~k3);
		}
		uint32_t m0 = 0, m1 = 0, mn = 0;
		bases4x(s4.x, 0, m0, m1, mn); bases4x(s4.y, 4, m0, m1, mn); bases4x(s4.z, 8, m0, m1, mn); bases4x(s4.w, 12, m0, m1, mn);
		unsigned short *p16 = reinterpret_cast<unsigned short *>(pl);
		if (c < NC16) {
			p16[0 * PW * 2 + c] = (unsigned short)m0; p16[1 * PW * 2 + c] = (unsigned short)m1; p16[2 * PW * 2 + c] = (unsigned short)mn;
			if (!qual) p16[3 * PW * 2 + c] = (unsigned short)0xffffu;
		} else { // the last block's piece shares its word with the first spare piece
			pl[0 * PW + PW - 2] = m0; pl[1 * PW + PW - 2] = m1; pl[2 * PW + PW - 2] = mn;
			if (!qual) pl[3 * PW + PW - 2] = 0xffffu;
		}
		if (c < 4) pl[c * PW + PW - 1] = 0;
	};
	auto planes_q = [&](uint32_t t, uint32_t *pl) { // threads with ld_q: quality >= q (count.c:85's signed compare)
		const int c = tid - QL0;
		const int64_t v = (int64_t)t * TILE - 64 + (int64_t)c * 16;
		uint4 q4 = pf;
		uint32_t k0, k1, k2, k3;
		if (ragged(v, k0, k1, k2, k3)) { q4.x &= k0; q4.y &= k1; q4.z &= k2; q4.w &= k3; } // (qualities outside the batch: 0)
		uint32_t mq = 0;
		const int T = P.q + 33;
		if (T >= 1 && T <= 127) {
			const uint32_t add = (uint32_t)(128 - T) * 0x01010101u;
			quals4x(q4.x, 0, add, mq); quals4x(q4.y, 4, add, mq); quals4x(q4.z, 8, add, mq); quals4x(q4.w, 12, add, mq);
		} else { quals16(q4.x, 0, P.q, mq); quals16(q4.y, 4, P.q, mq); quals16(q4.z, 8, P.q, mq); quals16(q4.w, 12, P.q, mq); }
		if (c < NC16) reinterpret_cast<unsigned short *>(pl)[3 * PW * 2 + c] = (unsigned short)mq;
		else pl[3 * PW + PW - 2] = mq;
	};

	// ---- the buckets' owners: lane i of waves 8.. owns bucket i -- its slab's base, the chunk it hands out next, chunks left in the group, the next group
	const bool owner = tid >= OWN0;
	// (recomputed where they are used -- the kernel runs at the register limit of its 1024 threads: ob = the bucket an owner thread owns, its slab's base)
#define ob ((uint32_t)(tid - OWN0))
	auto to16 = [](uint32_t r) -> uint32_t { return RW == 3 ? (r >> 2) * 3u : r; }; // a record index (a multiple of four where records are 12 bytes) in 16-byte units
	auto slab_of = [&](uint32_t b) -> uint32_t { return (b * 8u + home) * OP.cap + (b - OP.own_lo < OP.own_n ? OP.own_delta : 0u); };
	// An owner's state in one register (the kernel runs at the register limit of its 1024 threads): the chunk it hands out next (counted in chunks) << 4
	// | a request for the next group is on its way (rq, a reserver's register, or res / rdy) << 3 | chunks left in the group (G <= 4)
	uint32_t ost = G;
	auto ost_pos = [&]() -> uint32_t { return (ost >> 4) << CAPL; }; // (the position counts chunks there: a buffer may be 8 records)
	const uint32_t sig = G > 1u ? G - 1u : 1u; // ask for the next group when this many chunks of the current one are left (as soon as it is begun)
	auto claim = [&](uint32_t base, uint32_t n) -> uint32_t { // a reservation's answer: a full slab poisons the batch (it is replayed), write where it does no harm
		if (base + n > OP.cap) { OP.flags[0] = 1; return 0u; }
		return base;
	};
	if (owner) ost |= (claim(atomicAdd(&OP.cursor[((size_t)home * NB + ob) * 32], RES), RES) >> CAPL) << 4;
	// ---- the reservers: waves RSV0 .. RSV0 + NRW - 1, wave r active in rounds = r (mod NRW); lane l serves buckets l, l + 64, ...
	// (a lane serves NRS PAIRS of buckets l + 128 s, l + 128 s + 64 of its bank and takes up one request per pair and turn; 2^10 buckets are two
	// banks of 2^9, the second one's reservers eight waves further on: a lane's answers are registers, and four is what the kernel has)
	constexpr int NBANK = NB > 512 ? NB / 512 : 1, NRS = NB / NBANK / WAVE / 2;
	static_assert(NRS <= 4 && (NBANK == 1 || RSV0 + 8 * (NBANK - 1) + NRW <= BT / WAVE), "reservers");
	const int rsv_w = tid / WAVE - RSV0, rsv = rsv_w >= 0 && (rsv_w & 7) < NRW && (rsv_w >> 3) < NBANK ? (rsv_w & 7) : -1;
	const uint32_t rsv_b0 = rsv >= 0 ? (uint32_t)(rsv_w >> 3) * 512u + (uint32_t)lane : 0u; // its bank's first bucket of this lane
	uint32_t rv[NRS], infl = 0; // answers on their way: bit s of infl = rv[s] is one, bit 8 + s = it is the pair's second bucket's
#pragma unroll
	for (int u = 0; u < NRS; ++u) rv[u] = 0;
	uint32_t round = 0;

	uint4 *const out16 = reinterpret_cast<uint4 *>(out);
	const uint4 *const buf16 = reinterpret_cast<const uint4 *>(smem_wc);
	// The flushed chunks leave in 16-byte pieces, up to NPC per thread and round: read from the buffers into registers between barriers B and C
	// (nothing but LDS reads there), stored at the top of the next P1 -- where a wave that the memory pipeline holds up at the issue of its
	// stores (2 400 cycles per round when the owner waves stored inside P2, everyone else waiting at the barrier) leaves its SIMD to the
	// three other waves' hashing.  More chunks than NPC x BT pieces in one round (half the buckets due at once): the rest is stored from P3.
	constexpr int NPC = 3; // (a round flushes a fifth of its 4 BT positions' k-mers' worth of chunks on average: 2.4 pieces per thread)
	uint4 pv[NPC]; uint32_t pa[NPC];
#pragma unroll
	for (int i = 0; i < NPC; ++i) { pv[i] = make_uint4(0, 0, 0, 0); pa[i] = WC_NONE; }
	auto take_pieces = [&](uint32_t n) { // P3
		const uint32_t np = n * PIECES;
#pragma unroll
		for (int i = 0; i < NPC; ++i) {
			const uint32_t x = (uint32_t)tid + (uint32_t)i * BT;
			pa[i] = WC_NONE;
			if (x < np) {
				const uint32_t j = x / PIECES, p = x - j * PIECES;
				const uint2 e = wl[j];
				// (24-bit multiplies: the compiler's v_mad_u64_u32 for a 32-bit product + offset takes ANY register as the undefined high half of
				// its addend -- here the one the tile draw's atomic returns to, and with it an s_waitcnt vmcnt(0) for this thread's fresh stores)
				pv[i] = *reinterpret_cast<const uint4 *>(smem_wc + (__umul24(e.x, PIECES * 16u) + p * 16u)); pa[i] = e.y + p;
			}
		}
		for (uint32_t x = (uint32_t)tid + (uint32_t)NPC * BT; x < np; x += BT) { // (rare)
			const uint32_t j = x / PIECES, p = x - j * PIECES;
			const uint2 e = wl[j];
			out16[(size_t)e.y + p] = buf16[e.x * PIECES + p];
		}
	};
	auto store_pieces = [&]() { // top of P1 (and once behind the last round)
#pragma unroll
		for (int i = 0; i < NPC; ++i) if (pa[i] != WC_NONE && !BFCG_ABL(P, 256)) out16[(size_t)pa[i]] = pv[i];
	};

	if (ld_b | ld_q) {
		prefetch(t_cur);
		if (ld_b) planes_b(t_cur, planes); else planes_q(t_cur, planes);
		if (t_next != WC_NONE) prefetch(t_next);
	}
	__syncthreads();

	RecW<RW> w[S];
	uint32_t pend[S]; // a record that found its buffer full: where it goes once the buffer has been flushed (record index in buf), else WC_NONE
#pragma unroll
	for (int j = 0; j < S; ++j) pend[j] = WC_NONE;
	uint32_t n_k = 0, n_h = 0;
	int cur = 0;
	auto put = [&](uint32_t o, const RecW<RW> &r) {
		if constexpr (RW == 4) *reinterpret_cast<uint4 *>(smem_wc + __umul24(o, 16u)) = make_uint4(r.d[0], r.d[1], r.d[2], r.d[3]);
		else { uint32_t *p = reinterpret_cast<uint32_t *>(smem_wc + __umul24(o, 12u)); p[0] = r.d[0]; p[1] = r.d[1]; p[2] = r.d[2]; }
	};
	const RecGeom RG = rec_geom(P);

#ifdef BFCG_MEASURE // phase clocks (scripts/s1wc_phases.py): thread 0 and the first owner lane, summed per workgroup into the statistics' spare words
	unsigned long long tm_p1 = 0, tm_p2 = 0, tm_ld = 0, tm_ow = 0, tm_w8 = 0, tm_n = 0, tm_t = __builtin_readcyclecounter();
#endif
	for (;;) {
		// ---------------- P1
		store_pieces();
#pragma unroll
		for (int j = 0; j < S; ++j) if (pend[j] != WC_NONE) { put(pend[j], w[j]); pend[j] = WC_NONE; }
		const uint32_t *pl = planes + cur * 4 * PW;
		uint32_t sl[S], bo[S]; // slot; bucket << CAPL
#pragma unroll
		for (int j = 0; j < S; ++j) {
			int r = j * BT + tid;
			asm volatile("" : "+v"(r)); // (opaque: nothing of a body is hoisted out of the round loop)
			bool hi, have;
			U2 y0, y1;
			if constexpr (sizeof(W) == 8) have = kmer_at2<TILE, KC>(pl, r + mis, P.k, y0, y1, hi);
			else {
				W a0, a1;
				have = kmer_at<W, TILE>(pl, r + mis, P.k, m, a0, a1, hi);
				y0.lo = (uint32_t)a0; y0.hi = 0; y1.lo = (uint32_t)a1; y1.hi = 0;
			}
			sl[j] = WC_NONE; bo[j] = 0;
			if (have) {
				const uint32_t b = (y0.lo >> b1_shift) & (NB - 1u);
				if constexpr (RW == 3) pack3_fast(w[j], PK, y0, y1, P.idx_rank | (t_cur * (uint32_t)TILE + (uint32_t)r), hi);
				else Rec<RW>::pack(w[j], RG, u2_join(y0), u2_join(y1), P.idx_rank | (t_cur * (uint32_t)TILE + (uint32_t)r), hi);
				sl[j] = atomicAdd(&fill[b], 1u); bo[j] = b << CAPL;
				++n_k; n_h += hi;
			}
		}
#pragma unroll
		for (int j = 0; j < S; ++j) {
			if (sl[j] < CAP) put(bo[j] + sl[j], w[j]);
			else if (sl[j] < 2u * CAP) pend[j] = bo[j] + sl[j] - CAP;
			else if (sl[j] != WC_NONE) { // more than two buffers' worth for one bucket in one round: a chunk of its own, the record and CAP - 1 dead ones
				const uint32_t b = bo[j] >> CAPL;
				const uint32_t base = claim(atomicAdd(&OP.cursor[((size_t)home * NB + b) * 32], CAP), CAP);
				uint32_t *d = out + (size_t)(slab_of(b) + base) * RW;
#pragma unroll
				for (int z = 0; z < RW; ++z) d[z] = w[j].d[z];
				for (uint32_t z = RW; z < CAP * RW; ++z) d[z] = 0xffffffffu;
			}
		}
#ifdef BFCG_MEASURE
		if (tid == OWN0) tm_w8 += __builtin_readcyclecounter() - tm_t;
#endif
		__syncthreads(); // ---------------- A: every slot of this round is drawn, every put below CAP is in its buffer
#ifdef BFCG_MEASURE
		{ const unsigned long long now = __builtin_readcyclecounter(); tm_p1 += now - tm_t; tm_t = now; }
#endif
		// ---------------- P2
		if (tid == DRAW_T) { s_draw[3] = draw_settle(draw); draw = draw_issue(); } // (the draw issued a round ago; the next one)
		if (ld_b) {
			if (t_next != WC_NONE) {
				planes_b(t_next, planes + (cur ^ 1) * 4 * PW); // (its bases arrived while this tile was hashed)
				if (t_pf != WC_NONE) prefetch(t_pf);
			}
		}
		if (rsv >= 0) {
			if ((int)(round % NRW) == rsv) {
				__builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0), once and for what is NRW rounds old (said here: the compiler's own waits would sit between the new requests below)
#pragma unroll
				for (int u = 0; u < NRS; ++u) { // first every answer asked for at this wave's last turn, NRW rounds ago (one wait, for what is long there) ...
					if (infl & (1u << u)) {
						const uint32_t b = rsv_b0 + (uint32_t)u * 2u * WAVE + (infl & (0x100u << u) ? WAVE : 0u);
						res[b] = rv[u];
						__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
						rdy[b] = 1u;
					}
				}
				infl = 0;
#pragma unroll
				for (int u = 0; u < NRS; ++u) { // ... then the new requests (nothing below waits for them)
					const uint32_t b0 = rsv_b0 + (uint32_t)u * 2u * WAVE, b1 = b0 + WAVE;
					const bool n0 = rq[b0] != 0u, n1 = rq[b1] != 0u;
					if (n0 | n1) {
						const uint32_t b = n0 ? b0 : b1;
						rv[u] = atomicAdd(&OP.cursor[((size_t)home * NB + b) * 32], RES);
						rq[b] = 0u; infl |= (n0 ? 0x1u : 0x101u) << u;
					}
				}
			}
		}
		if (owner) {
			const uint32_t f = fill[ob];
			const bool due = f >= CAP;
			const unsigned long long dm = __ballot(due);
			uint32_t qb = 0;
			if (lane == 0) qb = atomicAdd(&nq[round & 1u], (uint32_t)__popcll(dm)); // one LDS atomic per wave for its places in the round's list
			qb = __builtin_amdgcn_readfirstlane(qb);
			if (due) {
				const uint32_t my = qb + __builtin_amdgcn_mbcnt_hi((uint32_t)(dm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dm, 0u));
				wl[my] = make_uint2(ob, to16(slab_of(ob) + ost_pos()));
				fill[ob] = (f < 2u * CAP ? f : 2u * CAP) - CAP;
				ost += 16u - 1u; // (the next chunk, one less left)
				if ((ost & 7u) == 0u) { // the next group
					uint32_t bb = ob, base;
					asm volatile("" : "+v"(bb)); // (opaque: the addresses of this seldom-taken path are made here, not hoisted out of the round loop into registers the kernel does not have)
					uint32_t op = ost & 8u;
					if (op && rdy[bb] != 0u) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); base = res[bb]; rdy[bb] = 0u; op = 0u; }
					else { // not there yet (or never asked for: the first groups of G = 1): this lane reserves for itself and waits, behind its wave's stores
						base = atomicAdd(&OP.cursor[((size_t)home * NB + bb) * 32], RES);
						__builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0) here, not in front of the register's next use
					}
					ost = (claim(base, RES) >> CAPL) << 4 | op | G;
				}
			}
			if ((ost & 7u) <= sig && !(ost & 8u)) { rq[ob] = 1u; ost |= 8u; }
			if (tid == OWN0) nq[(round & 1u) ^ 1u] = 0; // (the next round's counter: nobody reads or bumps it before barrier A of that round)
		}
		if (ld_q && t_next != WC_NONE) { // (threads of the owners' first waves)
			planes_q(t_next, planes + (cur ^ 1) * 4 * PW);
			if (t_pf != WC_NONE) prefetch(t_pf);
		}
#ifdef BFCG_MEASURE
		if (tid == 0) tm_ld += __builtin_readcyclecounter() - tm_t;
		if (tid == OWN0) tm_ow += __builtin_readcyclecounter() - tm_t;
#endif
		__syncthreads(); // ---------------- B: the round's list of flushes stands, the next tile's planes too
		if (!BFCG_ABL(P, 512)) take_pieces(nq[round & 1u]); // ---------------- P3
		__syncthreads(); // ---------------- C: the flushed buffers are free
#ifdef BFCG_MEASURE
		{ const unsigned long long now = __builtin_readcyclecounter(); tm_p2 += now - tm_t; tm_t = now; ++tm_n; }
#endif
		t_cur = t_next; t_next = t_pf; t_pf = s_draw[3]; cur ^= 1; ++round;
		if (t_cur == WC_NONE) break;
	}

	// ---- the end: what is left in the buffers leaves padded with dead records; what was reserved and not used is dead
	store_pieces();
#pragma unroll
	for (int j = 0; j < S; ++j) if (pend[j] != WC_NONE) put(pend[j], w[j]);
	__syncthreads();
	const uint4 dead = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
	auto dead_group = [&](uint32_t b_slab, uint32_t base) { // a group that was reserved and never begun
		if (base + RES > OP.cap) { OP.flags[0] = 1; return; } // (k_seg_setup reads the slab up to min(cursor, capacity): a piece of this group would lie inside it, unwritten)
		uint4 *d = out16 + (size_t)to16(b_slab + base);
		for (uint32_t z = 0; z < G * PIECES; ++z) d[z] = dead;
	};
	if (owner) {
		const uint32_t f = fill[ob]; // <= CAP
		for (uint32_t z = f * RW; z < CAP * RW; ++z) buf[(ob << CAPL) * RW + z] = 0xffffffffu;
		{ // the (padded) buffer itself: lane by lane, once per kernel
			const uint4 *sp = buf16 + ob * PIECES;
			uint4 *dp = out16 + (size_t)to16(slab_of(ob) + ost_pos());
			for (uint32_t z = 0; z < PIECES; ++z) dp[z] = sp[z];
		}
		uint4 *d = out16 + (size_t)to16(slab_of(ob) + ost_pos() + CAP);
		for (uint32_t z = 0; z < ((ost & 7u) - 1u) * PIECES; ++z) d[z] = dead; // the rest of the current group
		if ((ost & 8u) && rdy[ob] != 0u) dead_group(slab_of(ob), res[ob]); // a group that was published and never taken
	}
	if (rsv >= 0) { // groups whose answers never were published
#pragma unroll
		for (int u = 0; u < NRS; ++u) if (infl & (1u << u)) {
			const uint32_t b = rsv_b0 + (uint32_t)u * 2u * WAVE + (infl & (0x100u << u) ? WAVE : 0u);
			dead_group(slab_of(b), rv[u]);
		}
	}
#ifdef BFCG_MEASURE
	if (tid == 0 || tid == OWN0) {
		unsigned long long *st = OP.stats + (size_t)(blockIdx.x & (ST_SLOTS - 1)) * ST_N;
		if (tid == 0) { atomicAdd(&st[10], tm_p1); atomicAdd(&st[11], tm_p2); atomicAdd(&st[12], tm_ld); atomicAdd(&st[15], tm_n); }
		else { atomicAdd(&st[13], tm_ow); atomicAdd(&st[14], tm_w8); }
	}
#endif
	{ // the statistics k_hist1 keeps in the two-pass partition: k-mers, high-quality k-mers (as k_scatter1's one-pass variant does)
		for (int o = 32; o; o >>= 1) { n_k += __shfl_down(n_k, o); n_h += __shfl_down(n_h, o); }
		if (lane == 0 && n_k) {
			unsigned long long *st = OP.stats + (size_t)(blockIdx.x & (ST_SLOTS - 1)) * ST_N;
			atomicAdd(&st[ST_KMERS], (unsigned long long)n_k); atomicAdd(&st[ST_HIGH], (unsigned long long)n_h);
		}
	}
}

#undef draw_a
#undef ob

namespace {

template <typename W, int RW, int CAPL, int NBL, int BT, int S, int KC>
void launch_wc(const KParams &P, const uint8_t *seq, const uint8_t *qual, int64_t n_pos, uint32_t *out, const OnePass &OP, uint32_t G, unsigned grid, hipStream_t st)
{
	hipLaunchKernelGGL((k_scatter1_wc<W, RW, CAPL, NBL, BT, S, KC>), dim3(grid), dim3(BT), (size_t)(RW * 4) << (CAPL + NBL), st, P, seq, qual, n_pos, out, OP, G);
}
template <typename W, int RW, int CAPL, int NBL, int BT, int S, int KC>
hipError_t attr_wc() { return hipFuncSetAttribute((const void *)k_scatter1_wc<W, RW, CAPL, NBL, BT, S, KC>, hipFuncAttributeMaxDynamicSharedMemorySize, (RW * 4) << (CAPL + NBL)); }

} // namespace

namespace bfcg {

// The variants: <word, log2 records per buffer, log2 buckets, threads, k at compile time>.  512 threads: 2^12 records of buffers (48 KiB), two
// workgroups per CU -- one's flush (P2, P3: chains of dependent instructions and LDS round trips in a few waves) runs under the other's hashing;
// 1024 threads: 2^13 records (96 KiB), one workgroup per CU (BFCG_S1_WC_BT=1024, and what 2^10 buckets -- config c4's -b37 -- always take).
#define WC_VARIANTS(X) \
	X(uint64_t, 3, 3, 9, 512, 4, 33) X(uint64_t, 3, 3, 9, 512, 4, 0) X(uint64_t, 3, 4, 8, 512, 4, 0) X(uint32_t, 3, 3, 9, 512, 4, 0) X(uint32_t, 3, 4, 8, 512, 4, 0) \
	X(uint64_t, 3, 4, 9, 1024, 4, 33) X(uint64_t, 3, 4, 9, 1024, 4, 0) X(uint64_t, 3, 5, 8, 1024, 4, 0) X(uint32_t, 3, 4, 9, 1024, 4, 0) X(uint32_t, 3, 5, 8, 1024, 4, 0) \
	X(uint64_t, 3, 3, 10, 1024, 4, 33) X(uint64_t, 3, 3, 10, 1024, 4, 0) X(uint32_t, 3, 3, 10, 1024, 4, 0) \
	X(uint64_t, 4, 3, 10, 1024, 3, 0) /* 16-byte records (config c5: k = 51, -b37): 2^10 buffers of 8 x 16 bytes = 128 KiB, three positions per thread and round */

// A device or driver that refuses the kernels' dynamic LDS (up to 128 KiB + ~27 KiB static) loses k_scatter1_wc, not the library: scatter1_wc_plan
// then says no and level 1 runs the tile kernel (this used to fail every bfcg_create, whatever the geometry).
static int g_wc_refused = 0;
hipError_t set_scatter1wc_lds_attr(void)
{
	hipError_t e = hipSuccess;
#define X(W, R, C, N, B, S, K) if (e == hipSuccess) e = attr_wc<W, R, C, N, B, S, K>();
	WC_VARIANTS(X)
#undef X
	if (e != hipSuccess) {
		(void)hipGetLastError();
		if (!__atomic_exchange_n(&g_wc_refused, 1, __ATOMIC_RELAXED))
			fprintf(stderr, "[W::bfcg] k_scatter1_wc unavailable on this device (%s for its LDS buffers): level 1 takes the tile kernel\n", hipGetErrorString(e));
	}
	return hipSuccess;
}

// Whether a one-pass stage A of this geometry can run k_scatter1_wc, and how: 12-byte records packed from halves with the level-1 bucket a bit
// field of y0's low word (scatter1_fast, checked by the caller), 2^8 .. 2^10 level-1 buckets, slabs that begin on 16-byte boundaries, and slabs
// large enough for what the workgroups leave unused (up to a group and a half and a padded buffer per workgroup and bucket at the end: dead
// records).  BFCG_S1_WC=0: never; =2: whenever the geometry allows (tests: tiny slabs overflow and are replayed).
bool scatter1_wc_plan(const KParams &P, const OnePass &OP, int rw, int64_t n_pos, WcPlan *pl)
{
	const char *e = getenv("BFCG_S1_WC");
	const int mode = e ? atoi(e) : 1;
	if (mode == 0 || __atomic_load_n(&g_wc_refused, __ATOMIC_RELAXED)) return false;
	if (P.F1 < 8 || P.F1 > 10) return false;
	if ((OP.cap & 3u) || (OP.own_delta & 3u)) return false;
	// 16-byte records: the one geometry that is instantiated (config c5's: k > 32, 2^10 buckets); the bucket must be a bit field of y0's low word
	if (rw == 4 && !(P.F1 == 10 && P.k > 32 && P.k >= P.bf_shift - 9 && P.R + P.F2 + P.F1 <= 32)) return false;
	if (rw != 3 && rw != 4) return false;
	e = getenv("BFCG_S1_WC_BT");
	const int bt = P.F1 == 10 || (e && atoi(e) == 1024) ? 1024 : 512, spt = rw == 4 ? 3 : 4;
	const uint32_t capl = (bt == 512 ? 12 : 13) - P.F1, cap_rec = 1u << capl;
	static int n_cu = 0;
	if (!n_cu) { hipDeviceProp_t pr; int dev = 0; n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256; }
	unsigned g = (unsigned)(n_cu * (bt == 512 ? 2 : 1)) & ~7u; if (g < 8) g = 8; // persistent workgroups: as many as are resident at once, a multiple of 8 (XCDs)
	e = getenv("BFCG_S1_WC_WGS"); if (e && atoi(e) >= 8) g = (unsigned)atoi(e) & ~7u; // (tests: fewer workgroups on small draws)
	const int64_t tiles = (n_pos + spt * bt - 1) / (spt * bt);
	const unsigned gt = (unsigned)(((tiles + 7) / 8) * 8);
	if (gt < g) g = gt;
	// Chunks per reservation: four -- an owner asks for its next group when it begins one, the answer is published one to four rounds later, and a
	// bucket fills a chunk in two and a half (with two, measured: nearly every group found its successor missing and its owner reserved for
	// itself, 3 000 - 4 000 cycles of a round behind the wave's fresh stores).  Fewer where the slabs are small: all of an XCD's workgroups share a slab.
	uint32_t G = 4;
	const char *ce = getenv("BFCG_S1_CHUNK"); // (tests force chunk sizes on small draws)
	auto waste = [&](uint32_t gg) { return (uint64_t)(g / 8 + 1) * (gg * cap_rec * 3 / 2 + cap_rec) * 8; }; // (an eighth of a slab at most: a third of its head room)
	if (ce && atoi(ce) > 0) { G = (uint32_t)atoi(ce) >> capl; if (G < 1) G = 1; if (G > 4) G = 4; }
	else while (G > 1 && waste(G) > OP.cap) G >>= 1;
	if (mode != 2 && waste(G) > OP.cap) return false;
	// ... and against the BATCH (round 6): what a workgroup leaves unused does not shrink with the batch, so a batch of a few million
	// positions in a context sized for 2^28 filled its slabs with two or three dead records per live one for level 2 to read and skip.  A slab's
	// dead records stay below a quarter of the positions it expects: smaller groups first, then fewer workgroups (each then fills several chunks
	// per bucket); a batch too small even for eight workgroups of single chunks takes the tile kernel.
	if (mode != 2 && !(ce && atoi(ce) > 0)) {
		const uint64_t live = (uint64_t)n_pos / ((uint64_t)8 << P.F1);
		auto dead = [&](uint32_t gg, unsigned wg) { return (uint64_t)(wg / 8 + 1) * (gg * cap_rec * 3 / 2 + cap_rec); };
		while (dead(G, g) * 4 > live) {
			if (G > 1) G >>= 1;
			else if (g > 8) g = (g / 2 + 7) & ~7u;
			else return false;
		}
	}
	pl->rw = rw; pl->bt = bt; pl->spt = spt; pl->G = G; pl->grid = g;
	return true;
}

static unsigned long long g_wc_launches = 0; // (process-wide, for the tests: did a run take this kernel at all)

void run_scatter1_wc(const KParams &P, const uint8_t *seq, const uint8_t *qual, int64_t n_pos, uint32_t *out, const OnePass &OP, const WcPlan &pl, hipStream_t st)
{
	__atomic_fetch_add(&g_wc_launches, 1ull, __ATOMIC_RELAXED);
	const int kc = pl.rw == 3 && P.k == 33 && P.F1 >= 9 ? 33 : 0, w64 = P.k > 32;
	const int capl = (pl.bt == 512 ? 12 : 13) - P.F1;
#define X(W, R, C, N, B, S, K) if ((sizeof(W) == 8) == (w64 != 0) && R == pl.rw && C == capl && N == P.F1 && B == pl.bt && S == pl.spt && K == kc) { launch_wc<W, R, C, N, B, S, K>(P, seq, qual, n_pos, out, OP, pl.G, pl.grid, st); return; }
	WC_VARIANTS(X)
#undef X
	fprintf(stderr, "[bfcg] k_scatter1_wc: no variant for k=%d F1=%d threads=%d record dwords=%d\n", P.k, P.F1, pl.bt, pl.rw); abort(); // (scatter1_wc_plan admits only what is instantiated)
}

} // namespace bfcg

extern "C" uint64_t bfcg_s1wc_launches(void) { return __atomic_load_n(&bfcg::g_wc_launches, __ATOMIC_RELAXED); }
