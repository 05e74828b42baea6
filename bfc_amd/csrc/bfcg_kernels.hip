// bfcg_kernels.hip -- hand-written gfx950 kernels for the k-mer counting path of bfc
// (count.c + bbf.c + htab.c).  DESIGN.md section 2 has the pipeline; per batch of reads:
//
//   k_scatter1_wc (bfcg_scatter1wc.hip, round 5: 12-byte records, and 16-byte ones at 2^10 buckets) / k_scatter1 (every other geometry)
//                bases -> k-mers ONCE (K1, kmer_dev.h: windows of four bit planes, no rolling state) -> 12/16/20-byte records into the
//                level-1 buckets' slabs: through write-combining buffers in LDS that leave as whole chunks, or (k_scatter1) ordered by
//                bucket in LDS tile by tile and appended run by run (one cursor atomic per run)
//   k_seg_setup  the slabs' fill -> the segment list level 2 reads
//   k_scatter2   level-1 slabs -> one slab per bloom REGION (2^R blocks of 64 bytes), again one pass
//   k_bloom      one workgroup per region: the region of the bitmap in LDS, exact sequential `seen` flags by a first-setter
//                protocol in LDS, bits set, region back to HBM; seen k-mers leave as 8-byte table entries
//   k_commit_seg one workgroup per region: the region-owned segment of the count table through LDS, upserts by LDS atomics
//                (or, in filter mode, the second filter's slice sits in k_bloom's LDS and nothing is handed over)
//
// The two-pass partition of round 1 (k_hist1 / k_colsum / k_scan_top / k_apply; k_hist2 / k_scan2: histogram rows, scans, exact
// offsets, no atomics) remains for what the slabs cannot take: replays of batches that overflowed a slab, batches too small for
// their slabs, and level 1 of a multi-GPU rank.  k_commit / k_commit_stream + table_upsert (device-scope CAS) serve the table in the
// host's (sub-table, key) layout: order stamps, geometries whose identity does not fit a slot, segments that outgrew LDS, export.
//
// A region holds the k-mers whose bloom block id has the region's number as its top bits, so everything that can interact under the
// reference's sequential semantics (bbf.c:27-31: one 64-byte block per k-mer) meets inside one workgroup.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "kmer_dev.h"
#include "bfcg_internal.h"
#include "bfcg_dev.h"
#include "bfcg_k1.h"

using namespace bfcg;

#define WAVE 64


// ------------------------------------------------------------------------------------------
// K1: tile of positions -> bit planes in LDS


// 16 positions from `pos` on at the ragged ends of a batch (positions outside it read as separators), byte by byte
// (a real call, results by value: met by two tiles of a batch, and inlined its 16 byte loads would cost every tile's path registers)
__device__ __noinline__ uint4 ragged16(const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual, int64_t n_pos, int64_t pos, int q)
{
	uint32_t m0 = 0, m1 = 0, mn = 0, mq = 0;
#pragma unroll 1
	for (int b = 0; b < 16; ++b) {
		const int64_t pb = pos + b;
		const bool in = pb >= 0 && pb < n_pos;
		uint32_t w = in ? seq[pb] : (uint32_t)'\n', t0m = 0, t1m = 0, tnm = 0;
		bases16(w | 0x0a0a0a00u, 0, t0m, t1m, tnm);
		m0 |= (t0m & 1u) << b; m1 |= (t1m & 1u) << b; mn |= (tnm & 1u) << b;
		mq |= (uint32_t)(qual ? (in && ((int)(int8_t)qual[pb] - 33 >= q)) : 1) << b;
	}
	return make_uint4(m0, m1, mn, mq);
}

template <int TILE, int BT>
__device__ __forceinline__ void build_planes(const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual,
                                             int64_t n_pos, int64_t t0, int q, uint32_t *planes)
{
	constexpr int PW = (TILE + 64) / 32 + 2;
	const bool aligned = ((((uintptr_t)seq) | ((uintptr_t)qual)) & 15) == 0;
	if (aligned) {
		constexpr int NC16 = (TILE + 64) / 16;
		unsigned short *p16 = reinterpret_cast<unsigned short *>(planes);
		for (int c = threadIdx.x; c < NC16; c += BT) {
			const int64_t pos = t0 - 64 + (int64_t)c * 16;
			uint32_t m0 = 0, m1 = 0, mn = 0, mq = 0;
			if (pos >= 0 && pos + 16 <= n_pos) {
				uint4 s = *reinterpret_cast<const uint4 *>(seq + pos);
				bases4x(s.x, 0, m0, m1, mn); bases4x(s.y, 4, m0, m1, mn); bases4x(s.z, 8, m0, m1, mn); bases4x(s.w, 12, m0, m1, mn);
				if (qual) {
					uint4 v = *reinterpret_cast<const uint4 *>(qual + pos);
					const int T = q + 33;
					if (T >= 1 && T <= 127) {
						const uint32_t add = (uint32_t)(128 - T) * 0x01010101u;
						quals4x(v.x, 0, add, mq); quals4x(v.y, 4, add, mq); quals4x(v.z, 8, add, mq); quals4x(v.w, 12, add, mq);
					} else { quals16(v.x, 0, q, mq); quals16(v.y, 4, q, mq); quals16(v.z, 8, q, mq); quals16(v.w, 12, q, mq); }
				} else mq = 0xffffu;
			} else { const uint4 r = ragged16(seq, qual, n_pos, pos, q); m0 = r.x; m1 = r.y; mn = r.z; mq = r.w; } // ragged ends of the batch
			p16[0 * PW * 2 + c] = (unsigned short)m0; p16[1 * PW * 2 + c] = (unsigned short)m1;
			p16[2 * PW * 2 + c] = (unsigned short)mn; p16[3 * PW * 2 + c] = (unsigned short)mq;
		}
	} else {
		const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
		constexpr int NCH = (TILE + 64) / 64;
		for (int c = wave; c < NCH; c += BT / WAVE) {
			int64_t pos = t0 - 64 + (int64_t)c * 64 + lane;
			bool in = pos >= 0 && pos < n_pos;
			uint32_t ch = in ? seq[pos] : (uint32_t)'\n';
			uint32_t u = ch & 0xDFu; // fold case
			uint32_t code = (u == 'A') ? 0u : (u == 'C') ? 1u : (u == 'G') ? 2u : (u == 'T') ? 3u : 4u;
			bool hq = qual ? (in && ((int)(int8_t)qual[pos] - 33 >= q)) : true; // count.c:85 (signed char)
			uint64_t b0 = __ballot(code & 1u), b1 = __ballot((code >> 1) & 1u), bn = __ballot(code >> 2), bq = __ballot(hq);
			if (lane == 0) {
				planes[0 * PW + 2 * c] = (uint32_t)b0; planes[0 * PW + 2 * c + 1] = (uint32_t)(b0 >> 32);
				planes[1 * PW + 2 * c] = (uint32_t)b1; planes[1 * PW + 2 * c + 1] = (uint32_t)(b1 >> 32);
				planes[2 * PW + 2 * c] = (uint32_t)bn; planes[2 * PW + 2 * c + 1] = (uint32_t)(bn >> 32);
				planes[3 * PW + 2 * c] = (uint32_t)bq; planes[3 * PW + 2 * c + 1] = (uint32_t)(bq >> 32);
			}
		}
	}
	if (threadIdx.x < 8) planes[(threadIdx.x >> 1) * PW + PW - 2 + (threadIdx.x & 1)] = 0;
}


// ------------------------------------------------------------------------------------------
// records

// (the records -- RecGeom, Rec<3 / 4 / 5>, rec_dead -- are in bfcg_k1.h: bfcg_scatter1wc.hip packs them too)

template <typename W> __device__ __forceinline__ uint32_t fine_id(const KParams &P, uint64_t y0, uint64_t y1)
{
	W m = kmask<W>(P.k);
	uint64_t hash = bloom_hash<W>(P.k, (W)y0, (W)y1, m);
	uint64_t blk = hash & ((1ULL << (P.bf_shift - 9)) - 1);
	return (uint32_t)(blk >> P.R);
}

// ------------------------------------------------------------------------------------------
// The partition is free of global atomics: every tile writes its histogram row, a small scan
// turns the (tiles x buckets) matrix into absolute output offsets, and the scatter pass reads its
// row back.  (A first version reserved space with one atomicAdd per tile and bucket on 128 cursor
// words: 6.6 of 9.4 ms per step went into those atomics.)  Ranks inside a (tile, bucket) cell
// still come from LDS atomics -- the order inside a bucket is irrelevant because every record
// carries its file-order index.

// pass A: level-1 histogram rows straight from the bases
template <typename W, int TILE, int BT>
__global__ __launch_bounds__(BT) void k_hist1(KParams P, const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual,
                                              int64_t n_pos, uint32_t *__restrict__ rows1, unsigned long long *__restrict__ stats)
{
	constexpr int PW = (TILE + 64) / 32 + 2;
	__shared__ uint32_t planes[4 * PW];
	__shared__ uint32_t hist[BFCG_MAXB];
	const int nb1 = 1 << P.F1;
	const W m = kmask<W>(P.k);
	const int shift2 = P.F2;
	uint32_t n_k = 0, n_h = 0;
	const int64_t n_tiles = (n_pos + TILE - 1) / TILE;
	for (int64_t tile = xcd_tile(blockIdx.x, n_tiles); tile < n_tiles; tile += n_tiles) {
		__syncthreads();
		for (int i = threadIdx.x; i < nb1; i += BT) hist[i] = 0;
		build_planes<TILE, BT>(seq, qual, n_pos, tile * TILE, P.q, planes);
		__syncthreads();
#pragma unroll 2
		for (int r = threadIdx.x; r < TILE; r += BT) {
			W y0, y1; bool hi;
			if (kmer_at<W, TILE>(planes, r, P.k, m, y0, y1, hi)) {
				uint32_t f = fine_id<W>(P, y0, y1);
				atomicAdd(&hist[f >> shift2], 1u);
				++n_k; n_h += hi;
			}
		}
		__syncthreads();
		for (int i = threadIdx.x; i < nb1; i += BT) rows1[tile * nb1 + i] = hist[i];
	}
	// statistics: k-mers, high-quality k-mers
	for (int o = 32; o; o >>= 1) { n_k += __shfl_down(n_k, o); n_h += __shfl_down(n_h, o); }
	if ((threadIdx.x & 63) == 0 && !P.no_kstats) {
		unsigned long long *sl = stats + (size_t)(blockIdx.x & (ST_SLOTS - 1)) * ST_N;
		atomicAdd(&sl[ST_KMERS], (unsigned long long)n_k); atomicAdd(&sl[ST_HIGH], (unsigned long long)n_h);
	}
}

// column scan of the level-1 matrix rows1[T][NB] in three small steps (chunks of SCAN_CH rows)
#define SCAN_CH BFCG_SCAN_CH
__global__ __launch_bounds__(256) void k_colsum(const uint32_t *__restrict__ rows, int T, int NB, uint32_t *__restrict__ chunk)
{
	const int r0 = blockIdx.x * SCAN_CH, r1 = min(T, r0 + SCAN_CH);
	for (int b = threadIdx.x; b < NB; b += 256) {
		uint32_t s = 0;
#pragma unroll 16
		for (int r = r0; r < r1; ++r) s += rows[(size_t)r * NB + b];
		chunk[(size_t)blockIdx.x * NB + b] = s;
	}
}
// one workgroup: bucket totals -> start[NB+1]; chunk sums -> chunk offsets (in place);
// row_base[NB+1] = first level-2 histogram row of each bucket (ceil(total/tile2) rows per bucket)
__global__ __launch_bounds__(BFCG_MAXB) void k_scan_top(uint32_t *__restrict__ chunk, int n_chunks, int NB, uint32_t *__restrict__ start,
                                                  uint32_t *__restrict__ row_base, int tile2)
{
	__shared__ uint32_t tot[BFCG_MAXB], rws[BFCG_MAXB];
	const int b = threadIdx.x;
	uint32_t total = 0;
	if (b < NB) {
#pragma unroll 16
		for (int c = 0; c < n_chunks; ++c) total += chunk[(size_t)c * NB + b];
	}
	tot[b] = b < NB ? total : 0;
	rws[b] = b < NB ? (total + tile2 - 1) / tile2 : 0;
	__syncthreads();
	for (int o = 1; o < NB; o <<= 1) {
		uint32_t v = b >= o ? tot[b - o] : 0, w = b >= o ? rws[b - o] : 0;
		__syncthreads();
		tot[b] += v; rws[b] += w;
		__syncthreads();
	}
	if (b < NB) {
		uint32_t run = tot[b] - total; // exclusive
		start[b] = run;
		row_base[b] = rws[b] - (total + tile2 - 1) / tile2;
#pragma unroll 16
		for (int c = 0; c < n_chunks; ++c) { uint32_t v = chunk[(size_t)c * NB + b]; chunk[(size_t)c * NB + b] = run; run += v; }
	}
	if (b == NB - 1) { start[NB] = tot[b]; row_base[NB] = rws[b]; }
}
__global__ __launch_bounds__(256) void k_apply(uint32_t *__restrict__ rows, int T, int NB, const uint32_t *__restrict__ chunk)
{
	const int r0 = blockIdx.x * SCAN_CH, r1 = min(T, r0 + SCAN_CH);
	for (int b = threadIdx.x; b < NB; b += 256) {
		uint32_t run = chunk[(size_t)blockIdx.x * NB + b];
#pragma unroll 16
		for (int r = r0; r < r1; ++r) { uint32_t v = rows[(size_t)r * NB + b]; rows[(size_t)r * NB + b] = run; run += v; }
	}
}

// Exclusive scan of nb <= 2*BT bucket counters in LDS (in place) by the whole workgroup: every thread takes one or two consecutive
// entries, waves scan with shuffles, the waves' totals meet in wsum[BT/64].  Two barriers (the Hillis-Steele loop it replaces took
// 2 log2(nb)); returns the grand total to every thread.  The caller synchronises before using cnt[].
template <int BT>
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t *cnt, int nb, uint32_t *wsum)
{
	constexpr int NW = BT / WAVE;
	const int E = (nb + BT - 1) / BT, i0 = threadIdx.x * E, lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
	uint32_t v0 = i0 < nb ? cnt[i0] : 0, v1 = (E > 1 && i0 + 1 < nb) ? cnt[i0 + 1] : 0;
	const uint32_t s = v0 + v1;
	uint32_t inc = s;
#pragma unroll
	for (int o = 1; o < WAVE; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if (lane >= o) inc += t; }
	if (lane == WAVE - 1) wsum[wave] = inc;
	__syncthreads();
	uint32_t woff = 0, tot = 0;
#pragma unroll
	for (int w = 0; w < NW; ++w) { const uint32_t x = wsum[w]; if (w < wave) woff += x; tot += x; }
	const uint32_t ex = woff + inc - s;
	if (i0 < nb) cnt[i0] = ex;
	if (E > 1 && i0 + 1 < nb) cnt[i0 + 1] = ex + v0;
	return tot;
}

// pass B: K1 again; the tile's records are ordered by level-1 bucket in LDS and copied out run by run with
// neighbouring lanes (coalesced stores), at rows1[tile][bucket] (absolute offsets after the scan).


// KC > 0: k is known at compile time (instantiated for the reference's default k = 33: bfc.c:17, and what `-s 3g` sets).
// FAST: the caller has checked that the bucket is a bit field of y0's low word (k >= bf_shift - 9) and that 12-byte records can be packed
// from halves (scatter1_fast): the generic 64-bit code is not even compiled in.
//
// A workgroup walks its tiles (the grid is as many workgroups as the chip holds at once, dealt to the XCDs like the tiles: xcd_tile) and is
// software-pipelined against the latencies a tile meets (a workgroup that lives for one tile spends a quarter of its life waiting for its
// bases): the NEXT tile's bases and qualities are requested a round ahead, and the returning atomics that reserve the runs' places in the
// slabs are waited for only after the records are staged.  Loads, atomics and stores share one in-order counter (vmcnt) on this chip, so the
// order inside a round matters: whatever is waited for was issued BEFORE the previous round's stores or long after them.
// What the kernel costs on config c3 (scripts/s1_ablate.py, per step): ~26 ms the skeleton (planes, LDS ranks, scan, staging, five
// barriers), + ~12 ms hashing, + ~7 ms the cursors' atomics, + ~11 ms the stores (runs of ~6 records: partial lines).
template <typename W, int RW, int TILE, int BT, bool ONEPASS = false, int KC = 0, bool FAST = false>
__global__ __launch_bounds__(BT, 4) void k_scatter1(KParams P, const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual,
                                                 int64_t n_pos, const uint32_t *__restrict__ rows1, uint32_t *__restrict__ out, OnePass OP)
{
	constexpr int PW = (TILE + 64) / 32 + 2;
	constexpr int S = TILE / BT;
	static_assert(S * BT == TILE, "TILE = threads x k-mers per thread");
	constexpr int NC16 = (TILE + 64) / 16, NCH = NC16 + 1; // 16-byte blocks of a tile's planes (one more when the streams start inside a block)
	static_assert(NCH <= BT, "one 16-byte block per thread");
	extern __shared__ __attribute__((aligned(16))) unsigned char smem1[];
	uint32_t *stage = reinterpret_cast<uint32_t *>(smem1);                                   // TILE * RW dwords
	// The bucket of a staged record.  Where the file index is a dword of its own (12- and 20-byte records) the stage holds  bucket << 13 | r
	// in its place -- r = the position inside the tile; the copy-out puts the index back -- ; 16-byte records keep the bucket in 2 more bytes
	// where their y0 no longer says it (RecGeom), else it is recomputed.
	constexpr bool IDX_BK = RW != 4;
	constexpr int IDX_DW = RW - 1;
	static_assert(TILE <= 8192, "13 bits for the position inside a tile");
	const bool KEEP_BK = !IDX_BK && P.rec_n > 0;
	const RecGeom RG = rec_geom(P);
	unsigned short *sbk = reinterpret_cast<unsigned short *>(smem1 + (size_t)TILE * RW * 4);
	__shared__ uint32_t planes[2 * 4 * PW]; // two sets: the next tile's planes are made while this tile's records wait in the stage
	__shared__ uint32_t s_total;
	const int nb1 = 1 << P.F1;
	uint32_t *cnt = reinterpret_cast<uint32_t *>(smem1 + (size_t)TILE * (RW * 4 + (KEEP_BK ? 2 : 0))), *gdelta = cnt + nb1; // 2 x nb1 counters behind the stage
	uint32_t *gdelta_b = gdelta + nb1, *splitp = gdelta_b + nb1; // (one pass: a run's second piece, and the staged position where it begins)
	__shared__ uint32_t wsum1[BT / WAVE];
	const W m = kmask<W>(P.k);
	const int64_t n_tiles = (n_pos + TILE - 1) / TILE;
	// the bloom block id is the low bf_shift-9 bits of the hash, and for k >= bf_shift-9 those are y0's (kmer.h:87): the level-1 bucket is a
	// bit field of y0's low word then
	const bool blk_in_y0 = FAST || P.k >= P.bf_shift - 9;
	const int b1_shift = P.R + P.F2;
	Pack3 PK = pack3_geom(P);
	if (FAST) PK.ok = 1;
	uint32_t n_k = 0, n_h = 0;

	// The streams are read as aligned 16-byte blocks whatever their own alignment (the library cuts sub-batches where a read ends): both
	// pointers sit at the same offset `mis` inside a block (the host sees to it), position p is byte p + mis of the block stream that starts
	// at seq - mis, and the planes are built in THOSE coordinates: plane bit j = block-stream byte tile * TILE - 64 + j, a window starts
	// `mis` bits later.  Bytes of a block that lie outside the batch are turned into separators in the registers: no byte loops.
	const int mis = (int)((uintptr_t)seq & 15);
	const uint8_t *const sb = seq - mis, *const qb = qual ? qual - mis : nullptr;
	const int64_t v_end = n_pos + mis, v_last = (v_end - 1) & ~(int64_t)15; // valid block-stream bytes [mis, v_end); the last block that holds one
	uint4 pf_s = make_uint4(0, 0, 0, 0), pf_q = make_uint4(0, 0, 0, 0);
	// (the loads are unconditional, from an address clamped into the batch: their targets are the very registers the next round reads,
	// with no copy in between that would have to wait for them)
	const int pf_c = (int)threadIdx.x < NCH ? (int)threadIdx.x : NCH - 1;
	auto prefetch = [&](int64_t t) {
		const int64_t v = t * TILE - 64 + (int64_t)pf_c * 16, at = v < 0 ? 0 : v > v_last ? v_last : v;
		pf_s = *reinterpret_cast<const uint4 *>(sb + at);
		if (qual) pf_q = *reinterpret_cast<const uint4 *>(qb + at);
	};
	// bit planes of block-stream bytes [t * TILE - 64, (t + 1) * TILE + 16) into `pl`, from the prefetched 16 bases + 16 qualities of every thread
	auto make_planes = [&](int64_t t, uint32_t *pl) {
		const int c = threadIdx.x;
		if (c < NCH) {
			const int64_t v = t * TILE - 64 + (int64_t)c * 16;
			uint4 s4 = pf_s, q4 = pf_q;
			if (v < mis || v + 16 > v_end) { // a block at the ragged ends of the batch: bytes outside it read as separators (qualities: 0)
				const int lo = v >= mis ? 0 : mis - v >= 16 ? 16 : (int)(mis - v), hi = v_end - v >= 16 ? 16 : v_end - v <= 0 ? 0 : (int)(v_end - v);
				const uint32_t bm = hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u; // bit i: byte i belongs to the batch
				auto keep = [&](int d) { return (((bm >> (4 * d)) & 0xFu) * 0x00204081u & 0x01010101u) * 0xFFu; };
				const uint32_t k0 = keep(0), k1 = keep(1), k2 = keep(2), k3 = keep(3);
				s4.x = (s4.x & k0) | (0x0a0a0a0au & ~k0); s4.y = (s4.y & k1) | (0x0a0a0a0au & ~k1); s4.z = (s4.z & k2) | (0x0a0a0a0au & ~k2); s4.w = (s4.w & k3) | (0x0a0a0a0au & ~k3);
				q4.x &= k0; q4.y &= k1; q4.z &= k2; q4.w &= k3;
			}
			uint32_t m0 = 0, m1 = 0, mn = 0, mq = 0;
			bases4x(s4.x, 0, m0, m1, mn); bases4x(s4.y, 4, m0, m1, mn); bases4x(s4.z, 8, m0, m1, mn); bases4x(s4.w, 12, m0, m1, mn);
			if (qual) {
				const int T = P.q + 33;
				if (T >= 1 && T <= 127) {
					const uint32_t add = (uint32_t)(128 - T) * 0x01010101u;
					quals4x(q4.x, 0, add, mq); quals4x(q4.y, 4, add, mq); quals4x(q4.z, 8, add, mq); quals4x(q4.w, 12, add, mq);
				} else { quals16(q4.x, 0, P.q, mq); quals16(q4.y, 4, P.q, mq); quals16(q4.z, 8, P.q, mq); quals16(q4.w, 12, P.q, mq); }
			} else mq = 0xffffu;
			unsigned short *p16 = reinterpret_cast<unsigned short *>(pl);
			if (c < NC16) {
				p16[0 * PW * 2 + c] = (unsigned short)m0; p16[1 * PW * 2 + c] = (unsigned short)m1;
				p16[2 * PW * 2 + c] = (unsigned short)mn; p16[3 * PW * 2 + c] = (unsigned short)mq;
			} else { // the last block's piece shares its word with the first spare piece
				pl[0 * PW + PW - 2] = m0; pl[1 * PW + PW - 2] = m1; pl[2 * PW + PW - 2] = mn; pl[3 * PW + PW - 2] = mq;
			}
		}
		if (threadIdx.x < 4) pl[threadIdx.x * PW + PW - 1] = 0;
	};
	// Which tile next.  Two-pass partition: workgroup per tile, dealt XCD-contiguously (rows1 is indexed by the tile).  One pass: the workgroups
	// of the persistent grid draw tiles from a counter (behind the cursors) -- a workgroup that becomes resident late, because the previous
	// kernel still held its CU, then simply draws fewer; with a fixed deal such stragglers cost 50 -> 70 ms per c3 step.  The draw for the
	// tile after the next rides with the cursors' atomics, so that it is known when its bases are to be requested.
	// A tile BELONGS to an XCD (tile & 7: the slabs its records go to), one counter per XCD; a workgroup draws from its own XCD's counter and,
	// once that has run out, from the others' -- a slab's load is then the eighth of the batch it was sized for whatever the XCDs' speeds
	// (one counter for all, slabs by the drawing workgroup's XCD: with 64 KiB table segments being committed on the other stream the XCDs'
	// shares of c4's batches differed by more than the slabs' margin, and a batch that overflows a slab sends the whole run to two passes).
	__shared__ uint32_t s_draw[3];
	uint32_t *const tile_ctr = ONEPASS ? OP.cursor + (size_t)8 * nb1 * 32 : nullptr;
	uint32_t draw_a = 0; // (thread 0) XCDs whose counters this workgroup has found exhausted, counted from its own
	auto draw_issue = [&]() -> uint32_t { return draw_a < 8u ? atomicAdd(&tile_ctr[((blockIdx.x + draw_a) & 7u) * 4u], 1u) : 0u; };
	auto draw_settle = [&](uint32_t t) -> uint32_t {
		while (draw_a < 8u) {
			const uint32_t x = (blockIdx.x + draw_a) & 7u;
			if ((int64_t)t * 8 + x < n_tiles) return t * 8u + x;
			if (++draw_a < 8u) t = atomicAdd(&tile_ctr[((blockIdx.x + draw_a) & 7u) * 4u], 1u);
		}
		return 0xffffffffu;
	};
	int64_t it = blockIdx.x;
	int64_t tile, next_tile, next_it = 0;
	if (ONEPASS) {
		if (threadIdx.x == 0) { s_draw[0] = draw_settle(draw_issue()); s_draw[1] = draw_settle(draw_issue()); }
		__syncthreads();
		tile = s_draw[0]; next_tile = s_draw[1];
	} else { tile = xcd_tile(it, n_tiles); next_it = it + gridDim.x; next_tile = xcd_tile(next_it, n_tiles); }
	if (tile >= n_tiles) return;
	int cur = 0;
	constexpr int NBT = (BFCG_MAXB + BT - 1) / BT; // buckets per thread
	const uint32_t home = blockIdx.x & 7u;         // this workgroup's own slabs (its XCD's)
	uint32_t c_ptr[NBT], c_rem[NBT];               // what is left of the chunks this thread's buckets reserved last in the home slabs
#pragma unroll
	for (int u = 0; u < NBT; ++u) { c_ptr[u] = 0; c_rem[u] = 0; }
	for (int i = threadIdx.x; i < nb1; i += BT) cnt[i] = 0;
	prefetch(tile);
	make_planes(tile, planes);
	prefetch(next_tile);
	__syncthreads();
	for (;;) {
		const uint32_t *pl = planes + cur * 4 * PW;
		RecW<RW> w[S];
		uint32_t br[S]; // bucket << 16 | rank inside (tile, bucket)
#pragma unroll
		for (int j = 0; j < S; ++j) {
			int r = j * BT + threadIdx.x;
			asm volatile("" : "+v"(r)); // (opaque, so that nothing of a body is hoisted out of the tile loop: S sets of window addresses would live across it)
			bool hi;
			br[j] = 0xffffffffu;
			U2 y0, y1;
			bool have;
			if (BFCG_ABL(P, 2048)) { // (debug: no hashing, pseudo-random buckets -- what the scatter alone costs)
				const uint32_t z = ((uint32_t)(tile * TILE + r) * 2654435761u) ^ 0x9e3779b9u;
				y0.lo = z * 0x85ebca6bu; y0.hi = 0; y1.lo = z; y1.hi = 0; hi = true; have = (r % 151) >= 33;
			} else
			if constexpr (sizeof(W) == 8) have = kmer_at2<TILE, KC>(pl, r + mis, P.k, y0, y1, hi);
			else {
				W a0, a1;
				have = kmer_at<W, TILE>(pl, r + mis, P.k, m, a0, a1, hi);
				y0.lo = (uint32_t)a0; y0.hi = 0; y1.lo = (uint32_t)a1; y1.hi = 0;
			}
			if (have) {
				const uint32_t b = blk_in_y0 ? (y0.lo >> b1_shift) & (uint32_t)(nb1 - 1) : fine_id<W>(P, u2_join(y0), u2_join(y1)) >> P.F2;
				const uint32_t idx = P.idx_rank | (uint32_t)(tile * TILE + r); // end position = file order (rank-major across GPUs)
				if (RW == 3 && (FAST || PK.ok)) pack3_fast(*reinterpret_cast<RecW<3> *>(&w[j]), PK, y0, y1, idx, hi);
				else Rec<RW>::pack(w[j], RG, u2_join(y0), u2_join(y1), idx, hi);
				br[j] = (b << 16) | atomicAdd(&cnt[b], 1u); // (with IDX_BK the record's last dword is made at staging time: bucket << 13 | r)
				if (ONEPASS) { ++n_k; n_h += hi; }
			}
		}
		__syncthreads();
		// bucket counters -> exclusive offsets inside the stage; the runs' places in the output are requested now and used after the staging
		const uint32_t tot = block_scan_excl<BT>(cnt, nb1, wsum1);
		if (threadIdx.x == 0) s_total = tot;
		__syncthreads();
		uint32_t gd[NBT], g_ex[NBT], g_c[NBT]; // (one-pass: gd holds the cursor's answer until the records are staged)
		const uint32_t xcd = (uint32_t)tile & 7u; // (the XCD this tile belongs to -- this workgroup's own but for the last few: its slabs)
		const bool own = xcd == home;             // (a tile taken from another XCD's share reserves exactly, in that XCD's slabs)
		uint32_t draw = 0;
		if (ONEPASS && threadIdx.x == 0) draw = draw_issue();
#pragma unroll
		for (int u = 0; u < NBT; ++u) {
			const int i = threadIdx.x + u * BT;
			gd[u] = 0; g_ex[u] = 0; g_c[u] = 0;
			if (i < nb1) {
				g_ex[u] = cnt[i];
				if (!ONEPASS) gd[u] = rows1[tile * nb1 + i]; // global record index = staged position + gdelta[bucket] (u32 modular)
				else {
					g_c[u] = (i + 1 < nb1 ? cnt[i + 1] : tot) - g_ex[u];
					const uint32_t left = own ? c_rem[u] : 0u;
					if (g_c[u] > left && !BFCG_ABL(P, 1024)) {
						const uint32_t need = g_c[u] - left;
						gd[u] = atomicAdd(&OP.cursor[((size_t)xcd * nb1 + i) * 32], own && OP.chunk > need ? OP.chunk : need);
					}
				}
			}
		}
#pragma unroll
		for (int j = 0; j < S; ++j) {
			if (br[j] != 0xffffffffu) {
				const uint32_t b = br[j] >> 16, pos = cnt[b] + (br[j] & 0xffffu);
#pragma unroll
				for (int t = 0; t < (IDX_BK ? RW - 1 : RW); ++t) stage[(size_t)pos * RW + t] = w[j].d[t];
				if (IDX_BK) stage[(size_t)pos * RW + IDX_DW] = (b << 13) | (uint32_t)(j * BT + threadIdx.x);
				if (KEEP_BK) sbk[pos] = (unsigned short)b;
			}
		}
#pragma unroll
		for (int u = 0; u < NBT; ++u) {
			const int i = threadIdx.x + u * BT;
			if (i < nb1) {
				if (!ONEPASS) gdelta[i] = gd[u] - g_ex[u];
				else {
					const uint32_t slab = ((uint32_t)i * 8u + xcd) * OP.cap + ((uint32_t)i - OP.own_lo < OP.own_n ? OP.own_delta : 0u), left = own ? c_rem[u] : 0u;
					gdelta[i] = slab + c_ptr[u] - g_ex[u]; // (first piece: what the last chunk still holds; unused when left == 0)
					if (g_c[u] <= left) { splitp[i] = 0xffffffffu; c_ptr[u] += g_c[u]; c_rem[u] -= g_c[u]; }
					else {
						const uint32_t need = g_c[u] - left, sz = own && OP.chunk > need ? OP.chunk : need;
						uint32_t base = gd[u];
						if (base + sz > OP.cap) { OP.flags[0] = 1; base = 0; } // the slab is full: this batch will be replayed; meanwhile write where it does no harm
						splitp[i] = g_ex[u] + left;
						gdelta_b[i] = slab + base - (g_ex[u] + left);
						if (own) { c_ptr[u] = base + need; c_rem[u] = sz - need; }
					}
				}
			}
		}
		if (ONEPASS && threadIdx.x == 0) s_draw[2] = draw_settle(draw);
		if (next_tile < n_tiles) make_planes(next_tile, planes + (cur ^ 1) * 4 * PW); // (its bases arrived while this tile was hashed)
		__syncthreads();
		for (int i = threadIdx.x; i < nb1; i += BT) cnt[i] = 0; // (the offsets have served: counters of the next tile)
		const uint32_t n_in = s_total;
		for (uint32_t pos = threadIdx.x; pos < (BFCG_ABL(P, 512) ? 0u : n_in); pos += BT) {
			RecW<RW> rec;
#pragma unroll
			for (int t = 0; t < RW; ++t) rec.d[t] = stage[(size_t)pos * RW + t];
			uint32_t b;
			if (IDX_BK) { b = rec.d[IDX_DW] >> 13; rec.d[IDX_DW] = P.idx_rank | ((uint32_t)(tile * TILE) + (rec.d[IDX_DW] & 0x1fffu)); }
			else if (KEEP_BK) b = sbk[pos];
			else { uint64_t y0, y1; uint32_t idx; bool hi; Rec<RW>::unpack(rec, RG, 0u, y0, y1, idx, hi); b = fine_id<W>(P, y0, y1) >> P.F2; } // (nothing dropped here)
			const uint64_t dst = (uint32_t)(pos + (ONEPASS && pos >= splitp[b] ? gdelta_b[b] : gdelta[b]));
			if (!BFCG_ABL(P, 256)) rec_store<RW>(out + dst * RW, rec);
		}
		tile = next_tile; cur ^= 1;
		if (tile >= n_tiles) break;
		if (ONEPASS) next_tile = s_draw[2];
		else { it = next_it; next_it = it + gridDim.x; next_tile = xcd_tile(next_it, n_tiles); }
		prefetch(next_tile); // requested behind this tile's stores, used a whole round later
		__syncthreads(); // the stage, the counters and gdelta are free again only when every wave has copied its records out
	}
	if (ONEPASS) { // what is left of the last chunks: dead records
		RecW<RW> dead;
#pragma unroll
		for (int t = 0; t < RW; ++t) dead.d[t] = 0xffffffffu;
#pragma unroll
		for (int u = 0; u < NBT; ++u) {
			const int i = threadIdx.x + u * BT;
			if (i < nb1) for (uint32_t r = 0; r < c_rem[u]; ++r) rec_store<RW>(out + ((uint64_t)(((uint32_t)i * 8u + home) * OP.cap + ((uint32_t)i - OP.own_lo < OP.own_n ? OP.own_delta : 0u)) + c_ptr[u] + r) * RW, dead);
		}
	}
	if (ONEPASS) { // the statistics k_hist1 keeps in the two-pass partition: k-mers, high-quality k-mers
		for (int o = 32; o; o >>= 1) { n_k += __shfl_down(n_k, o); n_h += __shfl_down(n_h, o); }
		if ((threadIdx.x & 63) == 0 && n_k) {
			unsigned long long *sl = OP.stats + (size_t)(blockIdx.x & (ST_SLOTS - 1)) * ST_N;
			atomicAdd(&sl[ST_KMERS], (unsigned long long)n_k); atomicAdd(&sl[ST_HIGH], (unsigned long long)n_h);
		}
	}
}

// ------------------------------------------------------------------------------------------
// level 2: one level-1 bucket (blockIdx.y) into its 2^F2 fine buckets, tiles of TILE records.
// Histogram rows of bucket b1 live at rows2[row_base[b1] + tile][2^F2].

// level-1 bucket that owns histogram row `row` (row_base is ascending, nb1+1 entries)
__device__ __forceinline__ int row_bucket(const uint32_t *__restrict__ row_base, int nb1, uint32_t row)
{
	int lo = 0, hi = nb1; // invariant: row_base[lo] <= row < row_base[hi]
	while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (row_base[mid] <= row) lo = mid; else hi = mid; }
	return lo;
}

template <typename W, int RW, int TILE, int BT>
__global__ __launch_bounds__(BT) void k_hist2(KParams P, const uint32_t *__restrict__ in, const uint32_t *__restrict__ seg_beg,
                                              const uint32_t *__restrict__ seg_end, int n_seg, int segs_per_bucket,
                                              const uint32_t *__restrict__ row_base, uint32_t *__restrict__ rows2)
{
	const RecGeom RG = rec_geom(P);
	__shared__ uint32_t hist[BFCG_MAXB];
	const int nb2 = 1 << P.F2;
	const uint32_t n_rows = row_base[n_seg];
	const int64_t row = xcd_tile(blockIdx.x, n_rows);
	if (row >= n_rows) return;
	const int b1 = row_bucket(row_base, n_seg, (uint32_t)row); // segment = one level-1 bucket (from one source rank)
	const uint32_t tile = (uint32_t)row - row_base[b1];
	const uint32_t s = seg_beg[b1], e = seg_end[b1];
	const uint32_t imp = (P.f_base >> P.F2) + (uint32_t)(b1 / segs_per_bucket); // the segment's level-1 bucket (global): the bits its records do not store
	for (int i = threadIdx.x; i < nb2; i += BT) hist[i] = 0;
	__syncthreads();
#pragma unroll
	for (int j = 0; j < TILE / BT; ++j) {
		uint64_t i = (uint64_t)s + (uint64_t)tile * TILE + j * BT + threadIdx.x;
		if (i < e) {
			uint64_t y0, y1; uint32_t idx; bool hi;
			const RecW<RW> w = rec_load<RW>(in + i * RW);
			if (!rec_dead<RW>(w)) { // (dead records: slabs of a one-pass level 1 read by the two-pass level 2 -- a rank of a multi-GPU run, replays)
				Rec<RW>::unpack(w, RG, imp, y0, y1, idx, hi);
				atomicAdd(&hist[fine_id<W>(P, y0, y1) & (nb2 - 1)], 1u);
			}
		}
	}
	__syncthreads();
	uint32_t *rowp = rows2 + (size_t)row * nb2;
	for (int i = threadIdx.x; i < nb2; i += BT) rowp[i] = hist[i];
}

// one workgroup per level-1 bucket: column totals -> fine starts; rows -> absolute offsets in place
__global__ __launch_bounds__(BFCG_MAXB) void k_scan2(KParams P, const uint32_t *__restrict__ bucket_start, int segs_per_bucket,
                                               const uint32_t *__restrict__ row_base, uint32_t *__restrict__ rows2, uint32_t *__restrict__ start2, uint32_t *__restrict__ cnt_live)
{
	__shared__ uint32_t tot[BFCG_MAXB];
	const int nb2 = 1 << P.F2, b1 = blockIdx.x, c = threadIdx.x;
	const uint32_t r0 = row_base[b1 * segs_per_bucket], r1 = row_base[(b1 + 1) * segs_per_bucket];
	uint32_t total = 0;
	if (c < nb2) {
#pragma unroll 16
		for (uint32_t r = r0; r < r1; ++r) total += rows2[(size_t)r * nb2 + c];
	}
	tot[c] = c < nb2 ? total : 0;
	__syncthreads();
	for (int o = 1; o < nb2; o <<= 1) {
		uint32_t v = c >= o ? tot[c - o] : 0;
		__syncthreads();
		tot[c] += v;
		__syncthreads();
	}
	if (c < nb2) {
		uint32_t run = bucket_start[b1] + tot[c] - total;
		start2[((size_t)b1 << P.F2) + c] = run;
		// (the level-1 bucket's records were counted WITH the dead ones -- bucket_start -- so its last region is followed by a gap: the regions'
		// own counts say where they end)
		if (cnt_live) cnt_live[((size_t)b1 << P.F2) + c] = total;
#pragma unroll 16
		for (uint32_t r = r0; r < r1; ++r) { uint32_t v = rows2[(size_t)r * nb2 + c]; rows2[(size_t)r * nb2 + c] = run; run += v; }
	}
	if (b1 == (int)gridDim.x - 1 && c == 0) start2[(size_t)gridDim.x << P.F2] = bucket_start[gridDim.x];
}

// The tile is first ordered by fine bucket in LDS, then copied out: neighbouring lanes store neighbouring
// records of one run, so the 16-byte stores coalesce into line-sized requests (registers->HBM scatter of single
// records measured 2x WRITE_SIZE inflation and ~1.1 TB/s).
// ONEPASS2 (no k_hist2, no k_scan2): region f owns the slab out[f * cap2 .. (f + 1) * cap2); a tile's run for region f goes where an atomic on
// cnt2[f] says.  Rows are dealt XCD-contiguously, so a region's counter is (nearly always) touched from one XCD only.  A slab that cannot take a
// run raises flags[2] (of the batch's slot): k_bloom and the commit kernels of this batch then do nothing, k_seal makes it sticky and the host
// replays the batch.
struct OnePass2 { uint32_t *cnt2; uint32_t cap2; uint32_t *flags; };

// last kernel of a one-pass batch's stage B: a batch that overflowed a slab (level 1: flags[0], level 2: flags[2]) poisons the run
__global__ void k_seal(const uint32_t *flags, uint32_t *sticky) { if (flags[0] | flags[2]) *sticky = 1; }

// FAST2 (12-byte records whose region is a bit field of their first word -- scatter2_fast: k >= bf_shift - 9, so the block id is y0's low bits,
// and the bits below rec_lo = R + F2 are stored as they are): the level-2 bucket is (d[0] >> R) & (2^F2 - 1), no record is unpacked, no hash
// recomputed -- twice per record in the generic code (ranking and copy-out).
template <typename W, int RW, int TILE, int BT, bool ONEPASS2 = false, bool FAST2 = false>
__global__ __launch_bounds__(BT) void k_scatter2(KParams P, const uint32_t *__restrict__ in, const uint32_t *__restrict__ seg_beg,
                                                 const uint32_t *__restrict__ seg_end, int n_seg, int segs_per_bucket,
                                                 const uint32_t *__restrict__ row_base, const uint32_t *__restrict__ rows2,
                                                 uint32_t *__restrict__ out, OnePass2 O2)
{
	const RecGeom RG = rec_geom(P);
	constexpr int S = TILE / BT;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
	uint32_t *stage = reinterpret_cast<uint32_t *>(smem2);                              // TILE * RW dwords
	constexpr bool KEEP_BK = false;
	unsigned short *sbk = reinterpret_cast<unsigned short *>(smem2 + (size_t)TILE * RW * 4); // bucket of each staged record
	const int nb2 = 1 << P.F2;
	// nb2 counters behind the stage -- ONE array (round 6): the runs' output offsets (gdelta) are written over the counters once the tile is staged.  With a second array the
	// stage of 4096 12-byte records and 2 x 2^10 words was 56 KiB: TWO workgroups per CU where c3's 2^9 regions per bucket run three (52 KiB) -- config c4's level 2 at
	// 7.0 ps per record against c3's 5.2 was an occupancy step, not its shorter runs (tiles of 8192 records, which doubled the runs, changed nothing: DESIGN 6b).
	uint32_t *cnt = reinterpret_cast<uint32_t *>(smem2 + (size_t)TILE * (RW * 4 + (KEEP_BK ? 2 : 0))), *gdelta = cnt;
	__shared__ uint32_t wsum2[BT / WAVE];
	const uint32_t n_rows = row_base[n_seg];
	const int64_t row = xcd_tile(blockIdx.x, n_rows);
	if (row >= n_rows) return;
	const int b1 = row_bucket(row_base, n_seg, (uint32_t)row);
	const uint32_t tile = (uint32_t)row - row_base[b1];
	const uint32_t s = seg_beg[b1], e = seg_end[b1];
	const uint32_t imp = (P.f_base >> P.F2) + (uint32_t)(b1 / segs_per_bucket);
	const uint32_t *rowp = ONEPASS2 ? nullptr : rows2 + (size_t)row * nb2;
	for (int i = threadIdx.x; i < nb2; i += BT) cnt[i] = 0;
	__syncthreads();
	RecW<RW> w[S];
	uint32_t br[S]; // bucket << 16 | rank inside (tile, bucket)
#pragma unroll
	for (int j = 0; j < S; ++j) {
		uint64_t i = (uint64_t)s + (uint64_t)tile * TILE + j * BT + threadIdx.x;
		br[j] = 0xffffffffu;
		if (i < e) {
			uint64_t y0, y1; uint32_t idx; bool hi;
			w[j] = rec_load<RW>(in + i * RW);
			if (!rec_dead<RW>(w[j])) { // (what a level-1 workgroup left unused of its last chunk -- OnePass)
				uint32_t b;
				if constexpr (FAST2) b = (w[j].d[0] >> P.R) & (uint32_t)(nb2 - 1);
				else { Rec<RW>::unpack(w[j], RG, imp, y0, y1, idx, hi); b = fine_id<W>(P, y0, y1) & (nb2 - 1); }
				br[j] = (b << 16) | atomicAdd(&cnt[b], 1u);
			}
		}
	}
	__syncthreads();
	uint32_t n_live;
	{ // bucket counters -> exclusive offsets inside the stage; gdelta = where the bucket's run goes in the output
		const uint32_t tot = block_scan_excl<BT>(cnt, nb2, wsum2);
		n_live = tot;
		__syncthreads();
	}
	// One pass: a (tile, region) run reserves its place with a RETURNING atomic on the region's cursor -- executed at the memory side, a round trip of
	// microseconds --, and nothing needs the answer before the copy-out: the atomics are issued here, the tile is put in region order in LDS
	// meanwhile, and the answers are turned into gdelta behind that (round 6; they were awaited before the staging began).
	constexpr int NQ = (BFCG_MAXB + BT - 1) / BT;
	uint32_t q_base[NQ], q_ex[NQ], q_c[NQ];
	if (!ONEPASS2) { // two passes: the run's place is its histogram row's; global record index = staged position + gdelta[bucket] (wraps are fine: u32 modular)
#pragma unroll
		for (int u = 0; u < NQ; ++u) {
			const int i = threadIdx.x + u * BT;
			q_base[u] = 0; q_ex[u] = 0; q_c[u] = 0;
			if (i < nb2) q_base[u] = rowp[i] - cnt[i];
		}
	}
	if (ONEPASS2) {
		const uint32_t f0 = (uint32_t)(b1 / segs_per_bucket) << P.F2;
#pragma unroll
		for (int u = 0; u < NQ; ++u) {
			const int i = threadIdx.x + u * BT;
			q_base[u] = 0; q_ex[u] = 0; q_c[u] = 0;
			if (i < nb2) {
				q_ex[u] = cnt[i]; q_c[u] = (i + 1 < nb2 ? cnt[i + 1] : n_live) - q_ex[u];
				if (q_c[u]) q_base[u] = atomicAdd(&O2.cnt2[f0 + i], q_c[u]);
			}
		}
	}
#pragma unroll
	for (int j = 0; j < S; ++j) {
		if (br[j] != 0xffffffffu) {
			const uint32_t b = br[j] >> 16, pos = cnt[b] + (br[j] & 0xffffu);
#pragma unroll
			for (int t = 0; t < RW; ++t) stage[(size_t)pos * RW + t] = w[j].d[t];
			if (KEEP_BK) sbk[pos] = (unsigned short)b;
		}
	}
	__syncthreads(); // (every record is staged: nobody reads the counters as offsets any more -- they become the runs' output offsets)
	{
		const uint32_t f0 = (uint32_t)(b1 / segs_per_bucket) << P.F2;
#pragma unroll
		for (int u = 0; u < NQ; ++u) {
			const int i = threadIdx.x + u * BT;
			if (i < nb2) {
				uint32_t base = q_base[u];
				if (ONEPASS2) {
					if (q_c[u] && base + q_c[u] > O2.cap2) { O2.flags[2] = 1; base = 0; } // (the run then lands on records nobody will read: the buffer ends with a tile of slack)
					base = (f0 + (uint32_t)i) * O2.cap2 + base - q_ex[u];
				}
				gdelta[i] = base;
			}
		}
	}
	__syncthreads();
	const uint32_t n_in = n_live;
	for (uint32_t pos = threadIdx.x; pos < n_in; pos += BT) {
		RecW<RW> rec;
#pragma unroll
		for (int t = 0; t < RW; ++t) rec.d[t] = stage[(size_t)pos * RW + t];
		uint32_t b;
		if (KEEP_BK) b = sbk[pos];
		else if constexpr (FAST2) b = (rec.d[0] >> P.R) & (uint32_t)(nb2 - 1);
		else { uint64_t y0, y1; uint32_t idx; bool hi; Rec<RW>::unpack(rec, RG, imp, y0, y1, idx, hi); b = fine_id<W>(P, y0, y1) & (uint32_t)(nb2 - 1); }
		const uint64_t dst = (uint32_t)(pos + gdelta[b]);
		rec_store<RW>(out + dst * RW, rec);
	}
}

// ONEPASS: the level-1 output as segments for level 2.  One workgroup.  seg (bucket b, XCD x) = slab (b * 8 + x) of `cap` records, filled
// up to its cursor; row_base = first level-2 histogram row of every segment; bucket_start = the buckets' starts in the level-2 output.
// A batch whose slabs overflowed (flags[0] of its slot) gets empty segments -- level 2 has nothing to move; k_bloom and the table stage test the
// flags themselves -- and the host replays it and the batches behind it in order (two-pass partition).
__global__ __launch_bounds__(1024) void k_seg_setup(KParams P, const uint32_t *__restrict__ cursor, uint32_t cap, const uint32_t *flags, int tile2,
                                                    uint32_t *__restrict__ seg_beg, uint32_t *__restrict__ seg_end, uint32_t *__restrict__ row_base,
                                                    uint32_t *__restrict__ bucket_start)
{
	__shared__ uint32_t s_rows[1024], s_recs[1024];
	__shared__ uint32_t s_poison;
	const int nb1 = 1 << P.F1, t = threadIdx.x;
	if (t == 0) s_poison = flags[0];
	__syncthreads();
	const bool poison = s_poison != 0;
	// thread t owns bucket t (nb1 <= 1024): its 8 segments
	uint32_t len[8], rows = 0, recs = 0;
#pragma unroll
	for (int x = 0; x < 8; ++x) {
		uint32_t l = 0;
		if (t < nb1 && !poison) { l = cursor[((size_t)x * nb1 + t) * 32]; if (l > cap) l = cap; }
		len[x] = l; rows += (l + tile2 - 1) / tile2; recs += l;
	}
	s_rows[t] = rows; s_recs[t] = recs;
	__syncthreads();
	for (int o = 1; o < 1024; o <<= 1) { // inclusive scans over the buckets
		const uint32_t a = t >= o ? s_rows[t - o] : 0, b = t >= o ? s_recs[t - o] : 0;
		__syncthreads();
		s_rows[t] += a; s_recs[t] += b;
		__syncthreads();
	}
	if (t < nb1) {
		uint32_t r = s_rows[t] - rows;
		bucket_start[t] = s_recs[t] - recs;
#pragma unroll
		for (int x = 0; x < 8; ++x) {
			const uint32_t seg = (uint32_t)t * 8u + x, beg = seg * cap;
			seg_beg[seg] = beg; seg_end[seg] = beg + len[x]; row_base[seg] = r;
			r += (len[x] + tile2 - 1) / tile2;
		}
	}
	if (t == nb1 - 1) { row_base[(size_t)nb1 * 8] = s_rows[t]; bucket_start[nb1] = s_recs[t]; }
}

// A rank of a group, slab mode without the host in the batch's loop (bfcg_mg.hip, round 5).
// k_pack_rows (source, behind k_seg_setup): the fills of this rank's slabs as one ROW per destination -- rows[p * row_w + k * 8 + x] = records in
// the slab of (bucket p * nb_loc + k, XCD x), rows[p * row_w + nb_loc * 8] = a slab of this stage A overflowed (nothing of it may be used) -- which
// travels to rank p beside the block of slabs itself.
__global__ __launch_bounds__(256) void k_pack_rows(const uint32_t *__restrict__ seg_end, const uint32_t *__restrict__ flags, int nb_loc, uint32_t cap, uint32_t row_w,
                                                  uint32_t *__restrict__ rows)
{
	const uint32_t p = blockIdx.x, n = (uint32_t)nb_loc * 8u;
	const uint32_t poison = flags[0];
	for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
		const uint32_t seg = p * n + i; // (global bucket * 8 + XCD: k_seg_setup's segment number; its slab starts at seg x capacity)
		rows[(size_t)p * row_w + i] = poison ? 0u : seg_end[seg] - seg * cap;
	}
	if (threadIdx.x == 0) { rows[(size_t)p * row_w + n] = poison; rows[(size_t)p * row_w + n + 1] = 0u; }
}
// k_seg_setup_mg (owner): the rows of all N sources -> the segment arrays of this rank's level 2, as mg_process_any builds them on the host from
// the sizes: segment ((k * N + s) * 8 + x) = the slab of (source s, owned bucket k, XCD x) at its fixed place ((s * nb_loc + k) * 8 + x) x cap of the
// receive buffer.  One workgroup; thread t = k * N + s owns 8 segments (nb_loc x N = 2^F1 <= 1024 threads).  A source whose stage A overflowed
// empties the whole batch (every rank sees the same rows, so every rank's stage B of this batch moves nothing; the host finds the flag when it
// reads its own copy of the rows and repeats the batch through the two passes).  Only the sources [s_lo, s_hi) contribute: an owner that
// receives more than its regions take at full speed applies the sources in consecutive groups (bfcg_mg.hip), the first of them this way.
__global__ __launch_bounds__(1024) void k_seg_setup_mg(const uint32_t *__restrict__ rows, uint32_t row_w, int N, int s_lo, int s_hi, int nb_loc, uint32_t cap, int tile2,
                                                       uint32_t *__restrict__ seg_beg, uint32_t *__restrict__ seg_end, uint32_t *__restrict__ row_base,
                                                       uint32_t *__restrict__ bucket_start, unsigned long long *__restrict__ total)
{
	__shared__ uint32_t s_rows[1024], s_recs[1024];
	__shared__ uint32_t s_poison;
	const int t = threadIdx.x, nt = nb_loc * N, k = t / N, s = t - k * N;
	if (t == 0) s_poison = 0;
	__syncthreads();
	if (t < N && rows[(size_t)t * row_w + (size_t)nb_loc * 8] != 0) s_poison = 1; // (benign race: every writer stores 1)
	__syncthreads();
	const bool poison = s_poison != 0;
	uint32_t len[8], nrow = 0, recs = 0;
#pragma unroll
	for (int x = 0; x < 8; ++x) {
		uint32_t l = 0;
		if (t < nt && !poison && s >= s_lo && s < s_hi) { l = rows[(size_t)s * row_w + (size_t)k * 8 + x]; if (l > cap) l = cap; } // (sources outside [s_lo, s_hi): another pass takes them)
		len[x] = l; nrow += (l + tile2 - 1) / tile2; recs += l;
	}
	s_rows[t] = nrow; s_recs[t] = recs;
	__syncthreads();
	for (int o = 1; o < 1024; o <<= 1) {
		const uint32_t a = t >= o ? s_rows[t - o] : 0, b = t >= o ? s_recs[t - o] : 0;
		__syncthreads();
		s_rows[t] += a; s_recs[t] += b;
		__syncthreads();
	}
	if (t < nt) {
		uint32_t r = s_rows[t] - nrow;
		if (s == 0) bucket_start[k] = s_recs[t] - recs;
#pragma unroll
		for (int x = 0; x < 8; ++x) {
			const uint32_t seg = (uint32_t)t * 8u + x, beg = (((uint32_t)s * (uint32_t)nb_loc + (uint32_t)k) * 8u + x) * cap;
			seg_beg[seg] = beg; seg_end[seg] = beg + len[x]; row_base[seg] = r;
			r += (len[x] + tile2 - 1) / tile2;
		}
	}
	if (t == nt - 1) { row_base[(size_t)nt * 8] = s_rows[t]; bucket_start[nb_loc] = s_recs[t]; if (total) *total = s_recs[t]; }
}

__device__ __forceinline__ SegGeom seg_geom(const KParams &P) { SegGeom g; g.k = P.k; g.lo = P.seg_lo; g.hi = P.seg_hi; return g; }

// ------------------------------------------------------------------------------------------
// count table in HBM: 2^l_pre regions of 2^tab_cshift u64 slots, slot = key(50)<<14|high(6)<<8|count(8)
// exactly as htab.c:7-17 stores it; empty = 0.  Home slot = low bits of key>>14 (as khash does),
// linear probing confined to the region.  Saturating counters by CAS (htab.c:74-79).
// An upsert carries (c, h) = number of bfc_ch_insert calls and how many of them were high quality:
// count = min(255, #calls), high = min(63, #high calls) whatever the order (the first call stores
// count 1 and high = is_high, htab.c:73-75), so pre-aggregated increments are exact.

template <bool TRACK>
__device__ __forceinline__ void table_upsert(const KParams &P, unsigned long long *__restrict__ tab, uint64_t y0, uint64_t y1,
                                             uint32_t c, uint32_t h, unsigned long long *__restrict__ stats,
                                             uint64_t *__restrict__ ovf, uint32_t ovf_cap, unsigned long long *__restrict__ ovf_cnt,
                                             const TabOrder &O, unsigned long long sf, unsigned long long sl)
{
	uint64_t key;
	uint32_t sub = ch_subkey(P.k, P.l_pre, y0, y1, key);
	const uint32_t cmask = (1u << P.tab_cshift) - 1;
	unsigned long long *reg = tab + ((uint64_t)sub << P.tab_cshift);
	uint32_t pos = (uint32_t)(key >> 14) & cmask;
	const unsigned long long fresh = (key & ~0x3fffULL) | (c < 255 ? c : 255) | ((uint64_t)(h < 63 ? h : 63) << 8);
	for (uint32_t probe = 0; probe <= cmask; ++probe, pos = (pos + 1) & cmask) {
		unsigned long long cur = __hip_atomic_load(&reg[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (cur == 0) {
			cur = atomicCAS(&reg[pos], 0ULL, fresh);
			if (cur == 0) { atomicAdd(&stats[ST_KEYS], 1ULL); if (TRACK) O.note(sub, (uint64_t)(reg - tab) + pos, sf, sl); return; } // stats already points at this workgroup's slot
		}
		if ((cur >> 14) == (key >> 14)) {
			for (;;) {
				uint32_t nc = (uint32_t)(cur & 0xff) + c, nh = (uint32_t)((cur >> 8) & 0x3f) + h;
				unsigned long long nv = (cur & ~0x3fffULL) | (nc < 255 ? nc : 255) | ((uint64_t)(nh < 63 ? nh : 63) << 8);
				if (nv == cur) { if (TRACK) O.note(sub, (uint64_t)(reg - tab) + pos, sf, sl); return; }
				unsigned long long old = atomicCAS(&reg[pos], cur, nv);
				if (old == cur) { if (TRACK) O.note(sub, (uint64_t)(reg - tab) + pos, sf, sl); return; }
				cur = old;
			}
		}
	}
	// region full: park the k-mer; the host grows the table and replays (counts commute)
	unsigned long long o = atomicAdd(ovf_cnt, 1ULL); // one chip-wide list index (rare path)
	if (o < ovf_cap) { ovf[5 * o] = y0; ovf[5 * o + 1] = y1; ovf[5 * o + 2] = (uint64_t)c | ((uint64_t)h << 32); ovf[5 * o + 3] = sf; ovf[5 * o + 4] = sl; }
}

__global__ void k_table_replay(KParams P, unsigned long long *tab, const uint64_t *__restrict__ src, uint64_t n,
                               unsigned long long *stats, uint64_t *ovf, uint32_t ovf_cap, unsigned long long *ovf_cnt, TabOrder O)
{
	stats += (size_t)(blockIdx.x & (ST_SLOTS - 1)) * ST_N;
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
		table_upsert<true>(P, tab, src[5 * i], src[5 * i + 1], (uint32_t)src[5 * i + 2], (uint32_t)(src[5 * i + 2] >> 32), stats, ovf, ovf_cap, ovf_cnt,
		             O, src[5 * i + 3], src[5 * i + 4]);
}

// grow: re-insert every occupied slot of the old table (cshift_old) into the new one (P.tab_cshift)
__global__ void k_table_rehash(KParams P, const unsigned long long *__restrict__ old_tab, int cshift_old, unsigned long long *new_tab,
                               const unsigned long long *__restrict__ old_first, unsigned long long *__restrict__ new_first)
{
	const uint64_t n = (uint64_t)1 << (P.l_pre + cshift_old);
	const uint32_t cmask = (1u << P.tab_cshift) - 1;
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		unsigned long long v = old_tab[i];
		if (!v) continue;
		unsigned long long *reg = new_tab + ((i >> cshift_old) << P.tab_cshift);
		uint32_t pos = (uint32_t)(v >> 14) & cmask;
		for (;;) { // the new region is at most half full: always terminates
			if (atomicCAS(&reg[pos], 0ULL, v) == 0) break;
			pos = (pos + 1) & cmask;
		}
		if (old_first) new_first[(uint64_t)(reg - new_tab) + pos] = old_first[i];
	}
}

// ------------------------------------------------------------------------------------------
// bloom region kernel

// what finally happens to a k-mer that was seen c times (h of them high quality) in this batch
template <typename W, bool TRACK>
__device__ __forceinline__ void commit_seen(const KParams &P, const BloomArgs &A, uint64_t y0, uint64_t y1, uint32_t c, uint32_t h,
                                            uint32_t idx_first, uint32_t idx_last)
{
	if (BFCG_ABL(P, 1)) return;
	if (A.table) table_upsert<TRACK>(P, A.table, y0, y1, c, h, A.stats, A.tab_ovf, A.tab_ovf_cap, A.ovf_cnt, A.ord, A.batch_hi | idx_first, A.batch_hi | idx_last);
	else if (A.bloom_hi) { // count.c:67-68: second filter keeps k-mers seen at least twice (order independent)
		uint64_t hash = bloom_hash<W>(P.k, (W)y0, (W)y1, kmask<W>(P.k));
		BloomAddr a = bloom_addr(hash, P.bf_shift);
		unsigned int *blk = reinterpret_cast<unsigned int *>(A.bloom_hi) + a.blk * 16;
		uint32_t z = a.h1;
		for (int j = 0; j < P.n_hashes; ++j) { uint32_t b = bloom_next(z, a.h2); atomicOr(&blk[b >> 5], 1u << (b & 31)); }
	}
}

// LDS aggregation of seen k-mers: every occurrence of a k-mer lands in the same fine bucket, so a
// small LDS table keyed by the k-mer folds the batch's occurrences into ONE table update.
// id0 claims the slot by CAS; for k > 32 a second word id1 completes the identity.  A reader that
// finds id0 equal but id1 not yet published cannot decide and simply takes the direct path
// (always exact: updates commute).  cnt[2p] = occurrences, cnt[2p+1] = high-quality occurrences
// (plain non-returning LDS adds; a batch has < 2^32 k-mers, so they cannot wrap).
struct AggView { unsigned long long *id0, *id1; unsigned int *cnt; unsigned int *imin, *imax; uint32_t mask; }; // imin/imax: first / last seen file index (NULL: not tracked)

template <typename W, bool TRACK>
__device__ __forceinline__ bool agg_add(const KParams &P, const AggView &G, uint64_t y0, uint64_t y1, bool hi, uint32_t idx)
{
	const bool two = sizeof(W) == 8;
	const unsigned long long a = two ? y0 : ((y0 << P.k) | y1);
	if (a == FS_EMPTY) return false;
	uint32_t p = (uint32_t)((a * 0x9E3779B97F4A7C15ULL) >> 40) & G.mask;
	for (int probe = 0; probe < 12; ++probe, p = (p + 1) & G.mask) {
		unsigned long long cur = G.id0[p];
		if (cur == FS_EMPTY) {
			cur = atomicCAS(&G.id0[p], FS_EMPTY, a);
			if (cur == FS_EMPTY) { // claimed
				if (two) atomicExch(&G.id1[p], (unsigned long long)y1);
				cur = a;
			}
		}
		if (cur == a) {
			if (two) {
				unsigned long long b = *(volatile unsigned long long *)&G.id1[p];
				if (b == FS_EMPTY) return false; // identity not published yet
				if (b != y1) continue;           // another k-mer sharing y0
			}
			__hip_atomic_fetch_add(&G.cnt[2 * p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			if (hi) __hip_atomic_fetch_add(&G.cnt[2 * p + 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			if (TRACK) { atomicMin(&G.imin[p], idx); atomicMax(&G.imax[p], idx); }
			return true;
		}
	}
	return false;
}

template <typename W, bool TRACK>
__device__ __forceinline__ void emit_seen(const KParams &P, const BloomArgs &A, const AggView &G, uint64_t y0, uint64_t y1, bool hi, uint32_t idx)
{
	if (BFCG_ABL(P, 2)) return;
	if (!agg_add<W, TRACK>(P, G, y0, y1, hi, idx)) commit_seen<W, TRACK>(P, A, y0, y1, 1u, (uint32_t)hi, TRACK ? idx : 0u, TRACK ? idx : 0u);
}

// one k-mer record of the bloom kernel, decoded
struct KRec { uint64_t y0, y1; uint32_t idx, bl, h1, h2; bool hi; uint32_t d0, d1; }; // (d0, d1: the 12-byte record's first words, for the fast paths below)

__device__ __forceinline__ KRec decode_fast3(const Dec3 g, const RecW<3> &w, uint32_t imp)
{
	KRec r;
	r.d0 = w.d[0]; r.d1 = w.d[1]; r.idx = w.d[2];
	const uint32_t y0d = w.d[0] & ((1u << g.a) - 1u), upper = y0d >> g.up;
	const uint32_t y0lo = g.n ? (y0d & g.lowmask) | (imp << g.lo) | (upper << (g.lo + g.n)) : y0d; // low word of y0
	const uint32_t y1lo = __builtin_amdgcn_alignbit(w.d[1], w.d[0], g.a) & g.mk32;
	const uint32_t x = (y0lo - y1lo) ^ y1lo;            // low bits of h0 ^ h1 (h0 = y0 - y1 mod 2^k, kmer.h:85-86)
	const uint32_t hh = upper | (x << g.sh_x);
	r.bl = y0d & g.rmask; r.h1 = hh & 511u; r.h2 = (hh >> 9) & 511u;
	if ((r.h2 & 31u) == 0) r.h2 = (r.h2 + 1) & 511u;   // bbf.c:33
	r.hi = (w.d[1] >> g.sh_flag) & 1u;
	r.y0 = r.y1 = 0;
	return r;
}
// the 8-byte hand-over entry: identity inside the region << 1 | high-quality flag
__device__ __forceinline__ unsigned long long entry_fast3(const Dec3 g, const KRec &r, int k)
{
	const unsigned long long A = r.d0 | ((unsigned long long)r.d1 << 32);
	const unsigned long long y1 = (A >> g.a) & (k >= 64 ? ~0ULL : (1ULL << k) - 1);
	const uint32_t y0d = r.d0 & ((1u << g.a) - 1u);
	const unsigned long long id = (unsigned long long)((y0d & g.rmask) | ((y0d >> g.up) << g.R)) | (y1 << g.sh_y1);
	return (id << 1) | (unsigned long long)r.hi;
}

template <typename W, int RW>
__device__ __forceinline__ KRec decode_rec(const KParams &P, const RecW<RW> &w, W m, uint32_t rmask, uint32_t imp)
{
	KRec r;
	Rec<RW>::unpack(w, rec_geom(P), imp, r.y0, r.y1, r.idx, r.hi);
	BloomAddr a = bloom_addr(bloom_hash<W>(P.k, (W)r.y0, (W)r.y1, m), P.bf_shift);
	r.bl = (uint32_t)a.blk & rmask; r.h1 = a.h1; r.h2 = a.h2;
	r.d0 = r.d1 = 0;
	return r;
}

// Class table of the cold batches (KParams.dedupe): k-mers with the same bloom block, h1 and h2 touch the same bits, so among the copies of
// such a class inside one batch only the EARLIEST can be unseen -- the others find every bit set by it.  Entry = class << 32 | file index,
// the minimum wins; 8-byte entries over the memory the first-setter table and the lists use afterwards.  In an empty filter a batch
// holds ~1.3 copies of every genome k-mer: without this, every second k-mer contends for all its bits with its own copy (round 2's
// cold launches: 13.8 / 7.6 / 5.6 ms against 5.5 warm).
__device__ __forceinline__ uint32_t ct_home(uint32_t cls, uint32_t mask) { return ((cls * 0x9E3779B1u) >> 9 ^ cls) & mask; }
__device__ __forceinline__ void ct_insert(unsigned long long *ct, uint32_t mask, uint32_t cls, uint32_t idx)
{
	const unsigned long long e = ((unsigned long long)cls << 32) | idx;
	for (uint32_t p = ct_home(cls, mask);; p = (p + 1) & mask) { // (the table has twice the entries a workgroup can bring: always terminates)
		unsigned long long cur = ct[p];
		if (cur == FS_EMPTY) { cur = atomicCAS(&ct[p], FS_EMPTY, e); if (cur == FS_EMPTY) return; }
		if ((uint32_t)(cur >> 32) == cls) { if (e < cur) atomicMin(&ct[p], e); return; }
	}
}
__device__ __forceinline__ uint32_t ct_first(const unsigned long long *ct, uint32_t mask, uint32_t cls)
{
	for (uint32_t p = ct_home(cls, mask);; p = (p + 1) & mask) {
		const unsigned long long cur = ct[p];
		if ((uint32_t)(cur >> 32) == cls) return (uint32_t)cur;
	}
}

// LDS layout (dynamic): region 2^R*64 B | agg id0 ag*8 [| id1 ag*8] | agg cnt ag*8 | fs fs_cap*4 B | list idx list_cap*4 | list (record index | mask<<20) list_cap*4
//
// Structure, driven by measurements (SQ_WAIT_ANY 63 % of a workgroup's life, SALU instructions 1.6x VALU):
//  * pass 1 classifies every k-mer against the pre-batch region: the NH bit tests of the PF records a thread holds are
//    plain LDS reads issued back to back.  Seen k-mers (all bits set) go to the aggregation table; the others are
//    COMPACTED into an LDS list, so that the expensive part runs with full lanes instead of 26 % of them;
//  * pass 1.5 (dense over the list) enters each clear bit into the first-setter table;
//  * pass 2 (dense) decides seen <=> not the first setter of any of its bits, sets the bits, aggregates;
//  * the region goes back to HBM, the aggregated k-mers are streamed to k_commit (no returning atomics here).
// Buckets whose list or first-setter table overflow take the HBM-pool path (exact, slow).
// 512-thread workgroups run three per CU (LDS), i.e. 6 waves per SIMD: allow the registers that go with it (no scratch)
// FM (filter mode, `bfc -1`): the second bloom filter takes the k-mers seen before (count.c:67-68).  It is addressed by the same hash,
// so its blocks for this region's k-mers are the same 2^R blocks: that slice sits in LDS next to the first filter's (instead of the
// aggregation table) and a seen k-mer costs n_hashes LDS ORs -- no global atomics, no hand-over to k_commit.
// STREAM: no aggregation; every seen k-mer is appended, as the record it came in, to the region's slice of A.stream_out and k_commit_stream
// applies them.  For batches in which k-mers hardly repeat (a large genome at ~1x per batch) the aggregation table only costs: it fills
// with singletons and the rest updates the count table from inside this kernel, a returning atomic in a workgroup that lives microseconds.
// F3 (12-byte records into the hand-over log, geometry checked by the host: bloom_fast3): bloom address and hand-over entry on 32-bit words
// (decode_fast3 / entry_fast3) -- a variant of its own, so that the generic 64-bit decode does not cost it registers.
template <typename W, int RW, int BT, int PF, int NH, bool TRACK, bool FM = false, bool STREAM = false, bool SEGOUT = false, bool F3 = false>
__global__ __launch_bounds__(BT, BT == 512 ? 6 : 8) void k_bloom(KParams P, BloomArgs A)
{
	if (BFCG_ABL(P, 8)) return;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	__shared__ uint32_t s_list_n, s_ovf, s_pool_off, s_seen, s_agg_n, s_fs_used, s_pad[2];
	const uint32_t f = blockIdx.x;
	const bool ho_log = SEGOUT && A.ho_stride != 0;
	if (batch_poisoned(A)) { if (ho_log && threadIdx.x == 0) A.ho_mark[f] = A.ho_cur[f]; return; } // (an empty page: the batch will be replayed)
	uint32_t rs, n;
	region_list(A, f, rs, n);
	if (n == 0) { if (threadIdx.x == 0) { if (ho_log) A.ho_mark[f] = A.ho_cur[f]; else if (A.agg_cnt) A.agg_cnt[f] = 0; } return; }
	// where this batch's seen k-mers go: behind what earlier batches left in the region's log, or at the records' own offsets
	uint32_t ho_cur0 = 0;
	unsigned long long *ho_base = nullptr;
	if (SEGOUT) {
		if (ho_log) {
			ho_cur0 = A.ho_cur[f];
			if (ho_cur0 + n > A.ho_stride) { // (the host commits before a log can fill up: a bug if it ever happens -- loudly, not silently)
				if (threadIdx.x == 0) { atomicAdd(&A.stats[(size_t)(f & (ST_SLOTS - 1)) * ST_N + ST_ERR_POOL], 1ULL); A.ho_mark[f] = ho_cur0; }
				return;
			}
			ho_base = A.ho + (uint64_t)f * A.ho_stride + ho_cur0;
		} else ho_base = reinterpret_cast<unsigned long long *>(A.stream_out) + rs;
	}
	A.stats += (size_t)(f & (ST_SLOTS - 1)) * ST_N; // statistics are slotted: no chip-wide single-address atomics
	const int region_blocks = 1 << P.R;                 // P.R already clamped to bf_shift-9
	const uint32_t region_dw = region_blocks * 16;
	constexpr bool two = sizeof(W) == 8;
	unsigned char *sp = smem;
	unsigned int *region = reinterpret_cast<unsigned int *>(sp); sp += (size_t)region_dw * 4;
	AggView G;
	unsigned int *region_hi = nullptr;
	if (FM) { region_hi = reinterpret_cast<unsigned int *>(sp); sp += (size_t)region_dw * 4; G.id0 = G.id1 = nullptr; G.cnt = G.imin = G.imax = nullptr; G.mask = 0; }
	else if (STREAM) { G.id0 = G.id1 = nullptr; G.cnt = G.imin = G.imax = nullptr; G.mask = 0; }
	else {
		G.id0 = reinterpret_cast<unsigned long long *>(sp); sp += (size_t)P.ag_cap * 8;
		G.id1 = G.id0;
		if (two) { G.id1 = reinterpret_cast<unsigned long long *>(sp); sp += (size_t)P.ag_cap * 8; }
		G.cnt = reinterpret_cast<unsigned int *>(sp); sp += (size_t)P.ag_cap * 8;
		G.imin = G.imax = nullptr;
		if (TRACK) { G.imin = reinterpret_cast<unsigned int *>(sp); sp += (size_t)P.ag_cap * 4; G.imax = reinterpret_cast<unsigned int *>(sp); sp += (size_t)P.ag_cap * 4; }
		G.mask = P.ag_cap - 1;
	}
	unsigned int *fs = reinterpret_cast<unsigned int *>(sp); sp += (size_t)P.fs_cap * 4;
	unsigned int *list_a = reinterpret_cast<unsigned int *>(sp); sp += (size_t)P.list_cap * 4; // file-order index of the k-mer
	unsigned int *list_b = reinterpret_cast<unsigned int *>(sp);                                // record index in the bucket | clear-bit mask << 20
	const uint32_t fs_mask = P.fs_cap - 1;
	const uint32_t *recs = A.recs + (uint64_t)rs * RW;
	unsigned int *g_region = reinterpret_cast<unsigned int *>(A.bloom) + (uint64_t)f * region_dw;
	unsigned int *g_region_hi = FM ? reinterpret_cast<unsigned int *>(A.bloom_hi) + (uint64_t)f * region_dw : nullptr;
	const W m = kmask<W>(P.k);
	const uint32_t rmask = region_blocks - 1;
	const uint32_t imp = (P.f_base + f) >> P.F2; // the region's level-1 bucket: the bits of y0 its records do not store
	const int nh = NH ? NH : P.n_hashes;
	static_assert(!F3 || (RW == 3 && SEGOUT), "the fast decode is for 12-byte records into the hand-over log");
	const Dec3 D3 = dec3_geom(P);
	constexpr bool fast3 = F3;
	auto dec = [&](const RecW<RW> &w) -> KRec {
		if constexpr (F3) return decode_fast3(D3, *reinterpret_cast<const RecW<3> *>(&w), imp);
		else return decode_rec<W, RW>(P, w, m, rmask, imp);
	};
	// cold batch (host's hint) and every record of the region fits the threads' registers: copies of a k-mer are resolved by class first, and the
	// passes over the k-mers with clear bits run on the decoded records in registers (no list of record indices, no second look at HBM)
	const bool dd = P.dedupe && P.ct_cap >= 2u * BT * PF && n <= (uint32_t)(BT * PF) && n <= P.list_cap && P.R <= 10; // (cls_of packs the block into 10 bits)

	const bool timing = BFCG_ABL(P, 64) && threadIdx.x == 0;
	long long tq[7] = {0, 0, 0, 0, 0, 0, 0};
	if (timing) tq[0] = clock64();

	RecW<RW> rw[PF];
#pragma unroll
	for (int u = 0; u < PF; ++u) {
		uint32_t i = threadIdx.x + u * BT;
		if (i < n) {
			rw[u] = rec_load<RW>(recs + (uint64_t)i * RW);
		}
	}
	{ // stage the region (16-byte loads), clear the LDS tables
		const uint4 *src = reinterpret_cast<const uint4 *>(g_region);
		uint4 *dst = reinterpret_cast<uint4 *>(region);
		for (uint32_t i = threadIdx.x; i < region_dw / 4; i += BT) dst[i] = src[i];
		for (uint32_t i = threadIdx.x; i < (dd ? 2u * P.ct_cap : P.fs_cap); i += BT) fs[i] = FS32_EMPTY; // (dd: the class table lies over the first-setter table and the lists)
		if (FM) {
			const uint4 *src2 = reinterpret_cast<const uint4 *>(g_region_hi);
			uint4 *dst2 = reinterpret_cast<uint4 *>(region_hi);
			for (uint32_t i = threadIdx.x; i < region_dw / 4; i += BT) dst2[i] = src2[i];
		} else if (!STREAM) {
			for (uint32_t i = threadIdx.x; i < P.ag_cap; i += BT) {
				G.id0[i] = FS_EMPTY; if (two) G.id1[i] = FS_EMPTY; G.cnt[2 * i] = 0; G.cnt[2 * i + 1] = 0;
				if (TRACK) { G.imin[i] = 0xffffffffu; G.imax[i] = 0; }
			}
		}
		if (threadIdx.x == 0) { s_list_n = 0; s_seen = 0; s_agg_n = 0; s_ovf = 0; s_fs_used = 0; }
	}
	__syncthreads();
	if (timing) tq[1] = clock64();

	uint32_t n_seen = 0;
	volatile uint32_t *v_ovf = &s_ovf; // set when the LDS list or the LDS first-setter table cannot take a k-mer
	const int lane = threadIdx.x & 63;

	// mask of the bits of r that are clear in the (pre-batch) region
	auto clear_mask = [&](const KRec &r) -> uint32_t {
		uint32_t z = r.h1, um = 0;
#pragma unroll
		for (int j = 0; j < (NH ? NH : 12); ++j) {
			if (j >= nh) break;
			uint32_t b = bloom_next(z, r.h2);
			um |= (((region[r.bl * 16 + (b >> 5)] >> (b & 31)) & 1u) ^ 1u) << j;
		}
		return um;
	};
	// what happens to a seen k-mer: into the second filter's LDS slice (FM) or the aggregation table / count table
	auto emit = [&](const KRec &r) {
		if constexpr (FM) {
			uint32_t z = r.h1;
#pragma unroll
			for (int j = 0; j < (NH ? NH : 12); ++j) {
				if (j >= nh) break;
				uint32_t b = bloom_next(z, r.h2);
				__hip_atomic_fetch_or(&region_hi[r.bl * 16 + (b >> 5)], 1u << (b & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		} else if constexpr (STREAM) { // the lanes that emit right now take consecutive slots: one LDS atomic per wave, coalesced stores
			const unsigned long long vote = __ballot(1);
			const int leader = __ffsll((long long)vote) - 1;
			uint32_t o0 = 0;
			if (lane == leader) o0 = atomicAdd(&s_agg_n, (uint32_t)__popcll(vote));
			o0 = __shfl(o0, leader);
			const uint64_t at = (uint64_t)rs + o0 + (uint32_t)__popcll(vote & ((1ULL << lane) - 1));
			if constexpr (SEGOUT) { // region-owned table segments: all k_commit_seg needs is the k-mer's identity inside this region and its quality flag
				if constexpr (fast3) ho_base[at - rs] = entry_fast3(D3, r, P.k);
				else ho_base[at - rs] = (seg_id(seg_geom(P), r.y0, r.y1) << 1) | (unsigned long long)r.hi;
			} else {
				RecW<RW> w;
				Rec<RW>::pack(w, rec_geom(P), r.y0, r.y1, r.idx, r.hi);
				rec_store<RW>(A.stream_out + at * RW, w);
			}
		} else emit_seen<W, TRACK>(P, A, G, r.y0, r.y1, r.hi, r.idx);
	};
	// append a k-mer with clear bits to the LDS list: one LDS atomic per wave
	auto list_push = [&](bool want, uint32_t idx, uint32_t i, uint32_t um) {
		const unsigned long long vote = __ballot(want);
		if (!vote) return;
		const int leader = __ffsll((long long)vote) - 1;
		uint32_t li0 = 0;
		if (lane == leader) li0 = atomicAdd(&s_list_n, (uint32_t)__popcll(vote));
		li0 = __shfl(li0, leader);
		if (want) {
			const uint32_t li = li0 + (uint32_t)__popcll(vote & ((1ULL << lane) - 1));
			if (li < P.list_cap) { list_a[li] = idx; list_b[li] = i | (um << 20); }
			else *v_ovf = 1;
		}
	};

	// ---- cold batch, copies resolved by class (dd): the thread's records stay in registers (rw[], decoded again in every pass: arithmetic,
	// not memory) through all passes
	uint32_t dum[PF], dli[PF]; // clear-bit mask (0: nothing left to decide; bit 31: a later copy of its class -- seen, emitted in pass C); list index
	constexpr uint32_t DCOPY = 0x80000000u;
	auto cls_of = [](const KRec &r) { return r.bl | (r.h1 << 10) | (r.h2 << 19); };
	if (dd) {
		unsigned long long *ct = reinterpret_cast<unsigned long long *>(fs);
		const uint32_t ct_mask = P.ct_cap - 1;
#pragma unroll
		for (int u = 0; u < PF; ++u) {
			dum[u] = 0; dli[u] = 0;
			if (threadIdx.x + u * BT < n) {
				const KRec r = dec(rw[u]);
				dum[u] = clear_mask(r);
				if (dum[u] == 0) { // every bit was set before this batch: seen, whatever the order inside the batch
					++n_seen;
					if (A.seen_out) A.seen_out[r.idx] = 2;
					emit(r);
				} else ct_insert(ct, ct_mask, cls_of(r), r.idx);
			}
		}
		__syncthreads();
#pragma unroll
		for (int u = 0; u < PF; ++u)
			if (dum[u]) {
				const KRec r = dec(rw[u]);
				if (ct_first(ct, ct_mask, cls_of(r)) != r.idx) dum[u] = DCOPY; // an earlier copy sets every bit before this one comes
			}
		__syncthreads(); // the class table has served: its memory becomes the first-setter table and the list of file indices
		for (uint32_t i = threadIdx.x; i < P.fs_cap; i += BT) fs[i] = FS32_EMPTY;
		__syncthreads();
#pragma unroll
		for (int u = 0; u < PF; ++u) { // a list index for every k-mer still to decide: the first-setter entries name their k-mer by it
			const bool want = dum[u] != 0 && dum[u] != DCOPY;
			const unsigned long long vote = __ballot(want);
			if (vote) {
				const int leader = __ffsll((long long)vote) - 1;
				uint32_t li0 = 0;
				if (lane == leader) li0 = atomicAdd(&s_list_n, (uint32_t)__popcll(vote));
				li0 = __shfl(li0, leader);
				if (want) { dli[u] = li0 + (uint32_t)__popcll(vote & ((1ULL << lane) - 1)); list_a[dli[u]] = dec(rw[u]).idx; } // (<= n <= list_cap entries)
			}
		}
		__syncthreads();
		// pass A: set every clear bit; a bit somebody of this batch set before is contended -> first-setter entry (as below, on the list)
#pragma unroll
		for (int u = 0; u < PF; ++u) {
			if (!dum[u] || dum[u] == DCOPY) continue;
			const KRec r = dec(rw[u]);
			uint32_t z = r.h1;
#pragma unroll
			for (int j = 0; j < (NH ? NH : 12); ++j) {
				if (j >= nh) break;
				const uint32_t b = bloom_next(z, r.h2);
				if ((dum[u] >> j) & 1u) {
					const uint32_t bit = 1u << (b & 31);
					const uint32_t old = atomicOr(&region[r.bl * 16 + (b >> 5)], bit);
					if (old & bit) {
						*(volatile uint32_t *)&s_fs_used = 1;
						if (!fs32_insert(fs, fs_mask, r.bl * 512 + b, dli[u], r.idx, list_a)) *v_ovf = 1;
					}
				}
			}
		}
	}
	// ---- pass 1: classify; seen -> aggregate, clear bits -> list
	for (uint32_t base = 0; !dd && base < n; base += BT * PF) {
		KRec r[PF]; uint32_t um[PF]; bool act[PF];
#pragma unroll
		for (int u = 0; u < PF; ++u) {
			act[u] = base + threadIdx.x + u * BT < n;
			um[u] = 0;
			if (act[u]) { r[u] = dec(rw[u]); um[u] = clear_mask(r[u]); }
		}
#pragma unroll
		for (int u = 0; u < PF; ++u) {
			if (act[u] && um[u] == 0) { // every bit was set before this batch: seen, whatever the order inside the batch
				++n_seen;
				if (A.seen_out) A.seen_out[r[u].idx] = 2;
				emit(r[u]);
			}
			list_push(act[u] && um[u] != 0, r[u].idx, base + threadIdx.x + u * BT, um[u]);
		}
		if (base + BT * PF < n) { // next round (buckets larger than BT*PF)
#pragma unroll
			for (int u = 0; u < PF; ++u) {
				uint32_t i = base + BT * PF + threadIdx.x + u * BT;
				if (i < n) rw[u] = rec_load<RW>(recs + (uint64_t)i * RW);
			}
		}
	}
	__syncthreads();
	if (timing) tq[2] = clock64();
	const uint32_t ln = s_list_n;
	if (ln > P.list_cap || ln > 8191 || n >= (1u << 20)) s_ovf = 1; // benign race: every writer stores 1
	// ---- pass A (dense over the list): set every clear bit.  The returning atomicOr tells whether another k-mer of this
	// batch got there first: only such CONTENDED bits need a first-setter entry (file order decides them); a bit only one k-mer
	// touches is trivially set first by that k-mer.  A thread keeps the records of its first LK list entries in registers.
	constexpr int LK = 2;
	RecW<RW> lw[LK];
	if (!dd && !*v_ovf) {
#pragma unroll
		for (int q = 0; q < LK; ++q) {
			const uint32_t li = threadIdx.x + q * BT;
			if (li < ln) {
				const uint32_t i = list_b[li] & 0xfffffu;
				lw[q] = rec_load<RW>(recs + (uint64_t)i * RW);
			}
		}
		for (uint32_t li = threadIdx.x, q = 0; li < ln; li += BT, ++q) {
			RecW<RW> w;
			if (q < LK) w = q == 0 ? lw[0] : lw[LK - 1];
			else w = rec_load<RW>(recs + (uint64_t)(list_b[li] & 0xfffffu) * RW);
			KRec r = dec(w);
			const uint32_t um = list_b[li] >> 20;
			uint32_t z = r.h1;
#pragma unroll
			for (int j = 0; j < (NH ? NH : 12); ++j) {
				if (j >= nh) break;
				uint32_t b = bloom_next(z, r.h2);
				if ((um >> j) & 1u) {
					const uint32_t bit = 1u << (b & 31);
					const uint32_t old = atomicOr(&region[r.bl * 16 + (b >> 5)], bit);
					if (old & bit) { // contended
						*(volatile uint32_t *)&s_fs_used = 1;
						if (!fs32_insert(fs, fs_mask, r.bl * 512 + b, li, r.idx, list_a)) *v_ovf = 1;
					}
				}
			}
		}
	}
	__syncthreads();
	if (timing) tq[6] = clock64();

	bool dirty = true;
	if (!s_ovf) {
		// ---- pass B: the k-mer that set a contended bit first IN EXECUTION ORDER has not entered the competition yet:
		// every toucher of a bit that has an entry competes for it (earliest in file order wins)
		if (dd) { // the same two passes on the records in registers
			if (s_fs_used) {
#pragma unroll
				for (int u = 0; u < PF; ++u) {
					if (!dum[u] || dum[u] == DCOPY) continue;
					const KRec r = dec(rw[u]);
					uint32_t z = r.h1;
#pragma unroll
					for (int j = 0; j < (NH ? NH : 12); ++j) {
						if (j >= nh) break;
						const uint32_t b = bloom_next(z, r.h2);
						if ((dum[u] >> j) & 1u) fs32_compete(fs, fs_mask, r.bl * 512 + b, dli[u], r.idx, list_a);
					}
				}
				__syncthreads();
			}
#pragma unroll
			for (int u = 0; u < PF; ++u) {
				if (!dum[u]) continue;
				const KRec r = dec(rw[u]);
				bool first = dum[u] != DCOPY && !s_fs_used;
				if (dum[u] != DCOPY && s_fs_used) {
					uint32_t z = r.h1;
#pragma unroll
					for (int j = 0; j < (NH ? NH : 12); ++j) {
						if (j >= nh) break;
						const uint32_t b = bloom_next(z, r.h2);
						if ((dum[u] >> j) & 1u) {
							const uint32_t h = fs32_lookup(fs, fs_mask, r.bl * 512 + b);
							first |= (h == FS32_EMPTY) | (h == dli[u]); // uncontended, or this k-mer won
						}
					}
				}
				if (A.seen_out) A.seen_out[r.idx] = first ? 1 : 2;
				if (!first) { ++n_seen; emit(r); }
			}
		} else {
		if (s_fs_used) {
			for (uint32_t li = threadIdx.x, q = 0; li < ln; li += BT, ++q) {
				RecW<RW> w;
				if (q < LK) w = q == 0 ? lw[0] : lw[LK - 1];
				else w = rec_load<RW>(recs + (uint64_t)(list_b[li] & 0xfffffu) * RW);
				KRec r = dec(w);
				const uint32_t um = list_b[li] >> 20;
				uint32_t z = r.h1;
#pragma unroll
				for (int j = 0; j < (NH ? NH : 12); ++j) {
					if (j >= nh) break;
					uint32_t b = bloom_next(z, r.h2);
					if ((um >> j) & 1u) fs32_compete(fs, fs_mask, r.bl * 512 + b, li, r.idx, list_a);
				}
			}
			__syncthreads();
		}
		// ---- pass C: seen iff an earlier k-mer of the batch is the first setter of each of its clear bits
		for (uint32_t li = threadIdx.x, q = 0; li < ln; li += BT, ++q) {
			RecW<RW> w;
			if (q < LK) w = q == 0 ? lw[0] : lw[LK - 1];
			else w = rec_load<RW>(recs + (uint64_t)(list_b[li] & 0xfffffu) * RW);
			KRec r = dec(w);
			const uint32_t um = list_b[li] >> 20;
			uint32_t z = r.h1; bool first = !s_fs_used;
			if (!first) {
#pragma unroll
				for (int j = 0; j < (NH ? NH : 12); ++j) {
					if (j >= nh) break;
					uint32_t b = bloom_next(z, r.h2);
					if ((um >> j) & 1u) {
						const uint32_t h = fs32_lookup(fs, fs_mask, r.bl * 512 + b);
						first |= (h == FS32_EMPTY) | (h == li); // uncontended, or this k-mer won
					}
				}
			}
			if (A.seen_out) A.seen_out[r.idx] = first ? 1 : 2;
			if (!first) { ++n_seen; emit(r); }
		}
		}
		dirty = ln != 0;
		__syncthreads();
	} else {
		// ---- slow path: first-setter table in HBM.  The pool is A.pool_slices slices (a power of two, more than the workgroups
		// that can be resident at once) of 2 entries per bit of a region -- the most a region can ever need -- each guarded by a
		// lock word: a workgroup takes any free slice, so the path cannot run out of memory whatever the input.
		uint64_t want = (uint64_t)n * nh * 2;
		const uint64_t lim = (uint64_t)region_blocks * 512 * 2;
		if (want > lim) want = lim;
		uint32_t cap = 1024; while (cap < want) cap <<= 1;
		if (threadIdx.x == 0) {
			uint32_t sl = f & (A.pool_slices - 1);
			while (atomicCAS(&A.pool[sl], 0ULL, 1ULL) != 0ULL) sl = (sl + 1) & (A.pool_slices - 1);
			s_pool_off = sl;
			atomicAdd(&A.stats[ST_SLOW_BUCKETS], 1ULL);
		}
		{ // pass A may have set bits already: start again from the pre-batch region
			const uint4 *src = reinterpret_cast<const uint4 *>(g_region);
			uint4 *dst = reinterpret_cast<uint4 *>(region);
			for (uint32_t i = threadIdx.x; i < region_dw / 4; i += BT) dst[i] = src[i];
		}
		__syncthreads();
		unsigned long long *gfs = A.pool + A.pool_slices + (uint64_t)s_pool_off * lim;
		for (uint32_t i = threadIdx.x; i < cap; i += BT) gfs[i] = FS_EMPTY;
		__threadfence();
		__syncthreads();
		const uint32_t gmask = cap - 1;
		for (uint32_t i = threadIdx.x; i < n; i += BT) {
			KRec r = dec(rec_load<RW>(recs + (uint64_t)i * RW));
			uint32_t z = r.h1;
			for (int j = 0; j < nh; ++j) {
				uint32_t b = bloom_next(z, r.h2);
				if (!((region[r.bl * 16 + (b >> 5)] >> (b & 31)) & 1u)) fs_insert<true>(gfs, gmask, r.bl * 512 + b, r.idx, gmask);
			}
		}
		__threadfence();
		__syncthreads();
		for (uint32_t i = threadIdx.x; i < n; i += BT) {
			KRec r = dec(rec_load<RW>(recs + (uint64_t)i * RW));
			uint32_t z = r.h1; bool first = false, unresolved = false;
			for (int j = 0; j < nh; ++j) {
				uint32_t b = bloom_next(z, r.h2), fi;
				if (fs_lookup<true>(gfs, gmask, r.bl * 512 + b, fi)) { // has an entry <=> was clear before the batch
					unresolved = true; first |= (fi == r.idx);
					atomicOr(&region[r.bl * 16 + (b >> 5)], 1u << (b & 31));
				}
			}
			if (unresolved) {
				if (A.seen_out) A.seen_out[r.idx] = first ? 1 : 2;
				if (!first) { ++n_seen; emit(r); }
			}
		}
		__syncthreads();
		if (threadIdx.x == 0) { __threadfence(); atomicExch(&A.pool[s_pool_off], 0ULL); } // release the slice
	}
	if (timing) tq[3] = clock64();
	if (dirty) { // write the region back
		uint4 *dst = reinterpret_cast<uint4 *>(g_region);
		const uint4 *src = reinterpret_cast<const uint4 *>(region);
		for (uint32_t i = threadIdx.x; i < region_dw / 4; i += BT) dst[i] = src[i];
	}
	if (timing) tq[4] = clock64();
	if (FM) { // the second filter's slice goes back whole (every workgroup of a batch sees some k-mer again, except in the first batches)
		uint4 *dst = reinterpret_cast<uint4 *>(g_region_hi);
		const uint4 *src = reinterpret_cast<const uint4 *>(region_hi);
		if (__syncthreads_or(n_seen != 0))
			for (uint32_t i = threadIdx.x; i < region_dw / 4; i += BT) dst[i] = src[i];
	}
	// ---- hand the aggregated k-mers over: compacted into this bucket's slice of agg_out (k_commit applies them)
	for (uint32_t p0 = 0; p0 < ((FM || STREAM) ? 0u : P.ag_cap); p0 += BT) {
		const uint32_t p = p0 + threadIdx.x;
		unsigned long long a = p < P.ag_cap ? G.id0[p] : FS_EMPTY;
		const bool used = a != FS_EMPTY;
		const unsigned long long vote = __ballot(used);
		uint32_t o0 = 0;
		if (vote) {
			const int leader = __ffsll((long long)vote) - 1;
			if (lane == leader) o0 = atomicAdd(&s_agg_n, (uint32_t)__popcll(vote));
			o0 = __shfl(o0, leader);
		}
		if (used) {
			const uint32_t o = o0 + (uint32_t)__popcll(vote & ((1ULL << lane) - 1));
			uint32_t c = G.cnt[2 * p], h = G.cnt[2 * p + 1];
			if (c > 0xffffu) c = 0xffffu;
			if (h > 0xffffu) h = 0xffffu;
			uint64_t y0, y1;
			if (two) { y0 = a; y1 = G.id1[p]; } else { y0 = a >> P.k; y1 = a & (uint64_t)m; }
			if (A.agg_out) { // three planes (y0 | y1 | counts): every store instruction writes whole lines
				const uint64_t slot = (uint64_t)f * P.ag_cap + o, plane = (uint64_t)A.n_fine * P.ag_cap;
				A.agg_out[slot] = y0; A.agg_out[plane + slot] = y1; A.agg_out[2 * plane + slot] = c | (h << 16);
				if (TRACK) A.agg_out[3 * plane + slot] = (uint64_t)G.imin[p] | ((uint64_t)G.imax[p] << 32);
			} else commit_seen<W, TRACK>(P, A, y0, y1, c, h, TRACK ? G.imin[p] : 0u, TRACK ? G.imax[p] : 0u);
		}
	}
	for (int o = 32; o; o >>= 1) n_seen += __shfl_down(n_seen, o);
	if ((threadIdx.x & 63) == 0 && n_seen) atomicAdd(&s_seen, n_seen);
	__syncthreads();
	if (threadIdx.x == 0) {
		if (s_seen) atomicAdd(&A.stats[ST_SEEN], (unsigned long long)s_seen);
		if (ho_log) { A.ho_cur[f] = ho_cur0 + s_agg_n; A.ho_mark[f] = ho_cur0 + s_agg_n; }
		else if (A.agg_cnt) A.agg_cnt[f] = s_agg_n;
	}
	if (timing) {
		tq[5] = clock64();
		for (int t = 0; t < 5; ++t) atomicAdd(&A.stats[10 + t], (unsigned long long)(tq[t + 1] - tq[t]));
		atomicAdd(&A.stats[15], (unsigned long long)(tq[6] - tq[2])); // pass 1.5 alone (part of slot 12)
	}
	(void)s_pad;
}

// apply the aggregated k-mers of every bucket.  WALK = false: one thread per slot of agg_out (full occupancy; best up to ~2^18 regions);
// WALK = true: a wave walks COMMIT_RPW regions, lane j takes entries j, j+64, ... (at 2^20 regions a thread per slot launches a million
// nearly empty waves: 12.5 instead of 9 ms per batch on config c4)
#define COMMIT_RPW 4
template <typename W, bool TRACK, bool WALK>
__global__ __launch_bounds__(256) void k_commit(KParams P, BloomArgs A)
{
	A.stats += (size_t)(blockIdx.x & (ST_SLOTS - 1)) * ST_N;
	if (batch_poisoned(A)) return;
	const uint64_t plane = (uint64_t)A.n_fine * P.ag_cap;
	if (!WALK) {
		const uint64_t gid = blockIdx.x * 256ull + threadIdx.x;
		const uint32_t f = (uint32_t)(gid / P.ag_cap), j = (uint32_t)(gid % P.ag_cap);
		if (f >= A.n_fine) return;
		const uint32_t nf = A.agg_cnt[f];
		if (j == 0 && nf * 10 >= P.ag_cap * 9) atomicAdd(&A.stats[ST_CROWDED], 1ULL); // a full aggregation table: this batch's k-mers hardly repeat
		if (j >= nf) return;
		const uint32_t c = (uint32_t)A.agg_out[2 * plane + gid];
		const uint64_t fl = TRACK ? A.agg_out[3 * plane + gid] : 0;
		commit_seen<W, TRACK>(P, A, A.agg_out[gid], A.agg_out[plane + gid], c & 0xffffu, c >> 16, (uint32_t)fl, (uint32_t)(fl >> 32));
	} else {
		const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63;
		for (uint32_t f = wave * COMMIT_RPW; f < wave * COMMIT_RPW + COMMIT_RPW && f < A.n_fine; ++f) {
			const uint32_t n = A.agg_cnt[f];
			if (lane == 0 && n * 10 >= P.ag_cap * 9) atomicAdd(&A.stats[ST_CROWDED], 1ULL);
			for (uint32_t j = lane; j < n; j += 64) {
				const uint64_t gid = (uint64_t)f * P.ag_cap + j;
				const uint32_t c = (uint32_t)A.agg_out[2 * plane + gid];
				const uint64_t fl = TRACK ? A.agg_out[3 * plane + gid] : 0;
				commit_seen<W, TRACK>(P, A, A.agg_out[gid], A.agg_out[plane + gid], c & 0xffffu, c >> 16, (uint32_t)fl, (uint32_t)(fl >> 32));
			}
		}
	}
}

// STREAM mode: the seen k-mers of region f are the records stream_out[start[f] .. start[f] + agg_cnt[f]); one table update each
template <typename W, int RW>
__global__ __launch_bounds__(256) void k_commit_stream(KParams P, BloomArgs A)
{
	A.stats += (size_t)(blockIdx.x & (ST_SLOTS - 1)) * ST_N;
	if (batch_poisoned(A)) return;
	const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	for (uint32_t f = wave * COMMIT_RPW; f < wave * COMMIT_RPW + COMMIT_RPW && f < A.n_fine; ++f) {
		const uint32_t n = A.agg_cnt[f];
		uint32_t rs0, n0;
		region_list(A, f, rs0, n0);
		const uint64_t base = rs0;
		for (uint32_t j = lane; j < n; j += 64) {
			uint64_t y0, y1; uint32_t idx; bool hi;
			Rec<RW>::unpack(rec_load<RW>(A.stream_out + (base + j) * RW), rec_geom(P), (P.f_base + f) >> P.F2, y0, y1, idx, hi);
			commit_seen<W, false>(P, A, y0, y1, 1u, (uint32_t)hi, 0u, 0u);
		}
	}
}

// ------------------------------------------------------------------------------------------
// Region-owned table segments (KParams.seg).  Random device-scope CAS on a multi-GiB table ran at the chip's atomic rate (~20 G/s,
// 218 B of HBM traffic per upsert on config c3: profiles/round1_c3.md).  All k-mers of a bloom region already meet in one workgroup, so the
// table is kept per region instead: segment f = 2^seg_shift slots of  id << 14 | high << 8 | count  (the 14 low bits exactly as
// htab.c:7-17; id = the k-mer's y without the bits the region implies, kmer_dev.h), 0 = empty, linear probing inside the segment.
// k_commit_seg streams segment f through LDS -- coalesced load, upserts by LDS atomics (ds_cmpst_b64), coalesced store -- no global
// atomics at all.  The host's (sub-table, key) layout (htab.c:45-58) is produced once, at export (k_seg_to_table).


// returns 1: key created, 0: counts updated, -1: segment full.  (c, h) as in table_upsert: saturating, order independent.
template <bool LDS>
__device__ __forceinline__ int seg_upsert(unsigned long long *seg, uint32_t mask, uint64_t id, uint32_t c, uint32_t h)
{
	const unsigned long long fresh = (id << 14) | (c < 255 ? c : 255) | ((uint64_t)(h < 63 ? h : 63) << 8);
	uint32_t p = seg_home(id) & mask;
	// (linear probing: among 5 G keys in segments half full the longest run of occupied slots passes 256 -- measured on config c4 -- so the
	// probe sequence is bounded by the segment, not by a constant; a segment that is really full parks the k-mer and the host grows)
	for (uint32_t probe = 0; probe <= mask; ++probe, p = (p + 1) & mask) {
		unsigned long long cur = LDS ? __hip_atomic_load(&seg[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : __hip_atomic_load(&seg[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (cur == 0) {
			cur = atomicCAS(&seg[p], 0ULL, fresh);
			if (cur == 0) return 1;
		}
		if ((cur >> 14) == id) {
			for (;;) {
				const uint32_t nc = (uint32_t)(cur & 0xff) + c, nh = (uint32_t)((cur >> 8) & 0x3f) + h;
				const unsigned long long nv = (cur & ~0x3fffULL) | (nc < 255 ? nc : 255) | ((uint64_t)(nh < 63 ? nh : 63) << 8);
				if (nv == cur) return 0;
				const unsigned long long old = atomicCAS(&seg[p], cur, nv);
				if (old == cur) return 0;
				cur = old;
			}
		}
	}
	return -1;
}

__device__ __forceinline__ void seg_park(const BloomArgs &A, uint64_t y0, uint64_t y1, uint32_t c, uint32_t h)
{
	const unsigned long long o = atomicAdd(A.ovf_cnt, 1ULL); // segment full: the host grows the segments and replays (rare)
	if (o < A.tab_ovf_cap) { A.tab_ovf[5 * o] = y0; A.tab_ovf[5 * o + 1] = y1; A.tab_ovf[5 * o + 2] = (uint64_t)c | ((uint64_t)h << 32); A.tab_ovf[5 * o + 3] = 0; A.tab_ovf[5 * o + 4] = 0; }
}

// One workgroup per region: its seen k-mers are 8-byte entries -- identity inside the region << 1 | high-quality flag -- either the pages
// that the batches since the last commit appended to the region's hand-over log (A.ho_stride != 0: page j = log[mark[j-1][f], mark[j][f])),
// or this one batch's entries at stream_out[start[f] .. + agg_cnt[f]).  The segment goes through LDS ONCE for all of them: counts are
// saturating and order-free (htab.c:73-79), so nothing forces a pass over the table per pass over the filter (round 2 made one: 32 of 200 ms
// on c3, 38 % of c4).  Pages are applied one after the other with a barrier between them, so that the keys a page creates are exactly the
// keys that batch would have created: bfc_count's `# distinct k-mers` lines (count.c:113) stay exact per chunk.
template <int BT>
__global__ __launch_bounds__(BT) void k_commit_seg(KParams P, BloomArgs A)
{
	extern __shared__ __attribute__((aligned(16))) unsigned long long lseg[];
	__shared__ uint32_t s_new[BFCG_HO_MAX_PAGES], s_mark[BFCG_HO_MAX_PAGES];
	// Round 4: a segment of more than 2^14 slots (a genome that is large for its filter: 10 000 keys per region and more) is 2^ub BLOCKS of
	// 2^14 slots, one workgroup per (region, block): a key lives in block (seg_home >> seg_blk) & (blocks - 1) and is probed inside it, so a
	// block is what a segment was -- staged in LDS, applied, stored -- and its workgroup takes those entries of the region's pages that are
	// its own (every block's workgroup reads all of them: 8 bytes per entry against the 128 KiB of the block).  Before, such a table left
	// the region-owned layout for the host's and random device-scope CAS (218 B of HBM traffic per upsert).
	const uint32_t ub = (uint32_t)(P.seg_shift - P.seg_blk), blk_mask = (1u << ub) - 1u;
	// (region, block) of this workgroup.  One-dimensional launch `f * blocks + blk` wherever it fits (HIP bounds a dimension's THREADS by 2^32): a
	// region's blocks are then dispatched back to back and the second finds the region's log entries in L2 (c4: two blocks per region; with the
	// two-dimensional grid of round 4's end -- blockIdx.y = block -- they ran a whole sweep apart and the commits cost 1.28 instead of 1.09 s)
	const uint32_t f = gridDim.y == 1 ? blockIdx.x >> ub : blockIdx.x, blk = gridDim.y == 1 ? blockIdx.x & blk_mask : blockIdx.y;
	const bool log = A.ho_stride != 0;
	if (log && A.ho_pages > (uint32_t)BFCG_HO_MAX_PAGES) { // (the host never asks for more: a bug if it ever does -- loudly, not by dropping pages)
		if (threadIdx.x == 0) atomicAdd(&A.stats[(size_t)(f & (ST_SLOTS - 1)) * ST_N + ST_ERR_POOL], 1ULL);
		return;
	}
	const uint32_t pages = log ? A.ho_pages : 1u;
	uint32_t n;
	const unsigned long long *recs;
	if (log) { n = A.ho_mark[(size_t)(pages - 1) * A.ho_mark_stride + f]; recs = A.ho + (uint64_t)f * A.ho_stride; }
	else {
		if (batch_poisoned(A)) return;
		n = A.agg_cnt[f];
		uint32_t rs0, n0;
		region_list(A, f, rs0, n0);
		recs = reinterpret_cast<const unsigned long long *>(A.stream_out) + rs0;
	}
	if (n == 0) return;
	const uint32_t slots = 1u << P.seg_blk, mask = slots - 1;
	unsigned long long *gseg = A.seg_tab + ((uint64_t)f << P.seg_shift) + ((uint64_t)blk << P.seg_blk);
	const int blk_sh = P.seg_blk;
	auto mine = [&](unsigned long long v) -> bool { return ((seg_home(v >> 1) >> blk_sh) & blk_mask) == blk; }; // (one block: always)
	const SegGeom G = seg_geom(P);
	// Round 4: everything a page needs is requested BEFORE the segment is staged -- every page's end mark (one lane each, not one dependent
	// load per page in front of its barrier) and a thread's first two entries -- and a page costs one barrier, not two (a counter of new keys
	// per page instead of one that is cleared in between).  With 64 KiB segments (config c4: 1024 threads, ~1.2 entries per thread and page)
	// a page was two exposed memory latencies and two barriers for a microsecond of upserts: 22 ms per batch beside 23 ms for the stream.
	if (threadIdx.x < BFCG_HO_MAX_PAGES) {
		s_new[threadIdx.x] = 0;
		s_mark[threadIdx.x] = threadIdx.x < pages ? (log ? A.ho_mark[(size_t)threadIdx.x * A.ho_mark_stride + f] : n) : n;
	}
	uint32_t j = threadIdx.x;
	const unsigned long long pre0 = j < n ? recs[j] : 0ULL, pre1 = j + BT < n ? recs[j + BT] : 0ULL;
	// few k-mers for a large segment: touch their lines only (this workgroup alone owns the segment, the atomics order its own lanes)
	const bool direct = (uint64_t)n * 16 < slots;
	// Segments up to 2^12 slots keep a 32-bit counter pair per slot behind the segment in LDS: an occurrence of a key that is already there is ONE
	// non-returning LDS add (calls | high-quality calls << 16) instead of a compare-and-swap loop on the slot -- hot keys no longer make their
	// lanes retry -- and the counters are folded into the slots, saturating, before the segment goes back (htab.c:73-79 is order-free).
	const bool use_cnt = !direct && P.seg_blk <= 12 && n < 65536u && !BFCG_ABL(P, 16);
	unsigned int *lcnt = reinterpret_cast<unsigned int *>(lseg + slots);
	if (!direct) {
		const uint4 *src = reinterpret_cast<const uint4 *>(gseg);
		uint4 *dst = reinterpret_cast<uint4 *>(lseg);
		for (uint32_t i = threadIdx.x; i < slots / 2; i += BT) dst[i] = src[i];
		if (use_cnt) for (uint32_t i = threadIdx.x; i < slots; i += BT) lcnt[i] = 0;
	}
	__syncthreads(); // (the segment is staged, the marks are there)
	if (use_cnt) {
		// Entries and probes in ONE loop (round 4): a lane whose entry is done takes its next one while its neighbours still probe, so a wave runs
		// as many probe steps as its busiest LANE needs for all its entries, not the sum over entries of the longest probe among 64 lanes (at 60 %
		// load a key sits 2 slots from home on average, but the longest of 64 probes is 8 - 10: the counter passes showed 130 lane-instructions
		// per upsert).  The next entry is already requested (v1).
		unsigned long long v0 = pre0, v1 = pre1;
		uint32_t p = seg_home(v0 >> 1) & mask, probes = 0;
		for (uint32_t pg = 0; pg < pages; ++pg) {
			const uint32_t end = s_mark[pg];
			uint32_t n_new = 0;
			while (__any(j < end)) {
				if (j < end) {
					const uint64_t id = v0 >> 1;
					const uint32_t hi = (uint32_t)(v0 & 1);
					const unsigned long long fresh = (id << 14) | 1ULL | ((unsigned long long)hi << 8); // (the creating call is the slot's count 1)
					bool done = ub != 0 && !mine(v0); // (another block's entry)
					// An even slot is read together with its neighbour (one 16-byte LDS read): the probe sequence is the same, a step that would only
					// have found another key in slot p goes on to p + 1 at once (scripts/probes/commit_probe.hip on c3's shape: 12.5 -> 11.6 ms per pass).
					// A zero that went stale meanwhile is examined again by the compare-and-swap; a slot that holds a key keeps it.
					unsigned long long cur, nxt = 0;
					const bool pair = !(p & 1u);
					if (pair) { const ulonglong2 pr = *reinterpret_cast<const ulonglong2 *>(&lseg[p]); cur = pr.x; nxt = pr.y; }
					else cur = lseg[p];
					if (!done && cur == 0) {
						cur = atomicCAS(&lseg[p], 0ULL, fresh);
						if (cur == 0) { ++n_new; done = true; }
					}
					if (!done && (cur >> 14) == id) { __hip_atomic_fetch_add(&lcnt[p], 1u | (hi << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); done = true; }
					if (!done && pair) { // (p + 1 <= mask: p is even)
						++probes;
						if (nxt == 0) {
							nxt = atomicCAS(&lseg[p + 1], 0ULL, fresh);
							if (nxt == 0) { ++n_new; done = true; }
						}
						if (!done && (nxt >> 14) == id) { __hip_atomic_fetch_add(&lcnt[p + 1], 1u | (hi << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); done = true; }
						if (!done) p += 1;
					}
					if (!done) {
						p = (p + 1) & mask;
						if (++probes > mask) { uint64_t y0, y1; seg_unpack(G, (uint64_t)P.f_base + f, id, y0, y1); seg_park(A, y0, y1, 1u, hi); done = true; } // (the segment is full)
					}
					if (done) { // this lane's next entry
						j += BT; v0 = v1;
						v1 = j + BT < n ? recs[j + BT] : 0ULL; // (one entry ahead: two ahead cost more register copies per step than the load's latency, which the CU's other waves cover: 11.7 -> 10.9 ms per pass in the probe)
						p = seg_home(v0 >> 1) & mask; probes = 0;
					}
				}
			}
			for (int o = 32; o; o >>= 1) n_new += __shfl_down(n_new, o);
			if ((threadIdx.x & 63) == 0 && n_new) atomicAdd(&s_new[pg], n_new);
			__syncthreads(); // (the page is applied: the keys it created are its own -- bfc_count's `# distinct k-mers` lines stay exact per chunk)
		}
	} else {
	uint32_t k = 0; // entries this thread has applied
	for (uint32_t pg = 0; pg < pages; ++pg) {
		const uint32_t end = s_mark[pg];
		uint32_t n_new = 0;
		for (; j < end; j += BT, ++k) {
			const unsigned long long v = k == 0 ? pre0 : k == 1 ? pre1 : recs[j];
			if (ub != 0 && !mine(v)) continue; // (another block's entry)
			const uint64_t id = v >> 1;
			const uint32_t hi = (uint32_t)(v & 1);
			const int r = direct ? seg_upsert<false>(gseg, mask, id, 1u, hi) : seg_upsert<true>(lseg, mask, id, 1u, hi);
			if (r > 0) ++n_new;
			else if (r < 0) { uint64_t y0, y1; seg_unpack(G, (uint64_t)P.f_base + f, id, y0, y1); seg_park(A, y0, y1, 1u, hi); }
		}
		for (int o = 32; o; o >>= 1) n_new += __shfl_down(n_new, o);
		if ((threadIdx.x & 63) == 0 && n_new) atomicAdd(&s_new[pg], n_new);
		__syncthreads(); // (the page is applied: the keys it created are its own -- bfc_count's `# distinct k-mers` lines stay exact per chunk)
	}
	}
	if (use_cnt) { // the counters into their slots (behind the last page's barrier)
		for (uint32_t i = threadIdx.x; i < slots; i += BT) {
			const uint32_t c = lcnt[i];
			if (c) {
				const unsigned long long v = lseg[i];
				const uint32_t nc = (uint32_t)(v & 0xff) + (c & 0xffffu), nh = (uint32_t)((v >> 8) & 0x3f) + (c >> 16);
				lseg[i] = (v & ~0x3fffULL) | (nc < 255 ? nc : 255) | ((unsigned long long)(nh < 63 ? nh : 63) << 8);
			}
		}
		__syncthreads();
	}
	if (!direct) {
		uint4 *dst = reinterpret_cast<uint4 *>(gseg);
		const uint4 *src = reinterpret_cast<const uint4 *>(lseg);
		for (uint32_t i = threadIdx.x; i < slots / 2; i += BT) dst[i] = src[i];
	}
	if (threadIdx.x < pages) { // the keys page j created, slotted
		const uint32_t pn = s_new[threadIdx.x];
		if (pn) {
			if (A.ho_keys) atomicAdd(&A.ho_keys[(size_t)threadIdx.x * ST_SLOTS + (f & (ST_SLOTS - 1))], (unsigned long long)pn);
			atomicAdd(&A.stats[(size_t)(f & (ST_SLOTS - 1)) * ST_N + ST_KEYS], (unsigned long long)pn);
		}
	}
	if (threadIdx.x == 0 && log && blk == 0) A.ho_cur[f] = 0; // the log is empty again (the other blocks' workgroups read the pages' marks, not the cursor)
}

// grow: segment f of 2^old_shift slots -> 2^P.seg_shift slots, rebuilt in LDS BLOCK by block (all keys are distinct, a new block is at most half
// full).  Workgroup (f, blk) builds block blk of the new segment from the old block its keys can come from -- block blk & (old blocks - 1): the
// block of a key is a bit field of its home, which only gains high bits when the segment grows -- and takes the keys whose new block it is.
template <int BT>
__global__ __launch_bounds__(BT) void k_seg_rehash(KParams P, const unsigned long long *__restrict__ old_tab, int old_shift, int old_blk, unsigned long long *__restrict__ new_tab)
{
	extern __shared__ __attribute__((aligned(16))) unsigned long long lseg[];
	const uint32_t ub = (uint32_t)(P.seg_shift - P.seg_blk), blk_mask = (1u << ub) - 1u, ub_old = (uint32_t)(old_shift - old_blk);
	const uint32_t f = gridDim.y == 1 ? blockIdx.x >> ub : blockIdx.x, blk = gridDim.y == 1 ? blockIdx.x & blk_mask : blockIdx.y; // (as k_commit_seg)
	const uint32_t slots = 1u << P.seg_blk, mask = slots - 1, old_slots = 1u << old_blk;
	const unsigned long long *src = old_tab + ((uint64_t)f << old_shift) + ((uint64_t)(blk & ((1u << ub_old) - 1u)) << old_blk);
	for (uint32_t i = threadIdx.x; i < slots; i += BT) lseg[i] = 0;
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < old_slots; i += BT) {
		const unsigned long long v = src[i];
		if (!v) continue;
		const uint32_t home = seg_home(v >> 14);
		if (((home >> P.seg_blk) & blk_mask) != blk) continue;
		uint32_t p = home & mask;
		while (atomicCAS(&lseg[p], 0ULL, v) != 0ULL) p = (p + 1) & mask;
	}
	__syncthreads();
	uint4 *dst = reinterpret_cast<uint4 *>(new_tab + ((uint64_t)f << P.seg_shift) + ((uint64_t)blk << P.seg_blk));
	const uint4 *s4 = reinterpret_cast<const uint4 *>(lseg);
	for (uint32_t i = threadIdx.x; i < slots / 2; i += BT) dst[i] = s4[i];
}

// parked k-mers (y0, y1, c | h << 32, -, -) into the grown segments, straight in HBM (a handful per batch at most)
template <typename W>
__global__ void k_seg_replay(KParams P, unsigned long long *seg_tab, const uint64_t *__restrict__ src, uint64_t n, unsigned long long *stats,
                             uint64_t *ovf, uint32_t ovf_cap, unsigned long long *ovf_cnt)
{
	BloomArgs A; A.tab_ovf = ovf; A.tab_ovf_cap = ovf_cap; A.ovf_cnt = ovf_cnt;
	const SegGeom G = seg_geom(P);
	const uint32_t mask = (1u << P.seg_blk) - 1, blk_mask = (1u << (P.seg_shift - P.seg_blk)) - 1u;
	stats += (size_t)(blockIdx.x & (ST_SLOTS - 1)) * ST_N;
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t y0 = src[5 * i], y1 = src[5 * i + 1];
		const uint32_t c = (uint32_t)src[5 * i + 2], h = (uint32_t)(src[5 * i + 2] >> 32);
		const uint32_t f = fine_id<W>(P, y0, y1) - P.f_base;
		const uint64_t id = seg_id(G, y0, y1);
		const uint32_t blk = (seg_home(id) >> P.seg_blk) & blk_mask; // (the key's block of the segment; probing stays inside it)
		const int r = seg_upsert<false>(seg_tab + ((uint64_t)f << P.seg_shift) + ((uint64_t)blk << P.seg_blk), mask, id, c, h);
		if (r > 0) atomicAdd(&stats[ST_KEYS], 1ULL);
		else if (r < 0) seg_park(A, y0, y1, c, h);
	}
}

// export: every occupied slot of every segment becomes one bfc_ch_insert-equivalent upsert (with its counts) into the table in the
// host's layout.  Keys that the reference itself cannot tell apart (get_subhash is lossy for k >= 38, htab.c:53-56) meet here and
// their saturated counts add up, saturating -- what one shared counter would have reached (htab.c:74-79).
__global__ void k_seg_to_table(KParams P, const unsigned long long *__restrict__ seg_tab, uint32_t n_fine, unsigned long long *tab,
                               unsigned long long *stats, uint64_t *ovf, uint32_t ovf_cap, unsigned long long *ovf_cnt)
{
	const SegGeom G = seg_geom(P);
	const uint64_t n = (uint64_t)n_fine << P.seg_shift;
	TabOrder O; O.first = nullptr; O.sub_last = nullptr;
	stats += (size_t)(blockIdx.x & (ST_SLOTS - 1)) * ST_N;
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		const unsigned long long v = seg_tab[i];
		if (!v) continue;
		uint64_t y0, y1;
		seg_unpack(G, (uint64_t)P.f_base + (i >> P.seg_shift), v >> 14, y0, y1);
		table_upsert<false>(P, tab, y0, y1, (uint32_t)(v & 0xff), (uint32_t)((v >> 8) & 0x3f), stats, ovf, ovf_cap, ovf_cnt, O, 0ULL, 0ULL);
	}
}

// ------------------------------------------------------------------------------------------
// debug / unit-test kernel: K1 only, one output row per position (y0,y1,flags) ; flags bit0 valid, bit1 high
template <typename W, int TILE, int BT>
__global__ __launch_bounds__(BT) void k_hash_only(KParams P, const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual,
                                                  int64_t n_pos, uint64_t *__restrict__ out)
{
	constexpr int PW = (TILE + 64) / 32 + 2;
	__shared__ uint32_t planes[4 * PW];
	const W m = kmask<W>(P.k);
	const int64_t n_tiles = (n_pos + TILE - 1) / TILE;
	for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		__syncthreads();
		build_planes<TILE, BT>(seq, qual, n_pos, tile * TILE, P.q, planes);
		__syncthreads();
		for (int j = 0; j < TILE / BT; ++j) {
			int r = j * BT + threadIdx.x;
			int64_t e = tile * TILE + r;
			if (e >= n_pos) continue;
			W y0 = 0, y1 = 0; bool hi = false;
			bool ok = kmer_at<W, TILE>(planes, r, P.k, m, y0, y1, hi);
			out[3 * e] = ok ? (uint64_t)y0 : 0; out[3 * e + 1] = ok ? (uint64_t)y1 : 0; out[3 * e + 2] = (uint64_t)ok | ((uint64_t)(ok && hi) << 1);
		}
	}
}

// ------------------------------------------------------------------------------------------
// trim pass of `bfc -1` (config c5): bloom QUERY kernel + per-read longest streak
//   k_query : K1 + bfc_bf_get (bbf.c:47-63) for the k-mer ending at every position -> flag byte 0 none / 1 miss / 2 hit
//   k_streak: max_streak (correct.c:478-497) and the keep/trim rule (correct.c:557-569), one lane per read

// COOP: for filters far larger than the caches (see below); otherwise every lane gathers for itself, first bit first
// The query kernels' result is ONE BIT per stream position -- the k-mer ending there is in the filter -- written as the wave's ballot: 64 positions,
// one 8-byte store (a tile starts at a multiple of 64 and a wave takes 64 consecutive positions, so word e / 64 is the wave's own).  Round 1-4 wrote a
// byte per position (none / miss / hit) and k_streak, one lane per read, fetched them back byte by byte at a stride of a read's length: 83 of the
// 174 ms of c5's trim pass per 41 M reads went into that kernel (profiles/round5_trim.md).  All a read's longest streak needs is the hit bit
// (correct.c:483-495: anything that is not a hit -- a miss, a position without a k-mer -- restarts the run).
__device__ __forceinline__ void query_emit(unsigned long long *__restrict__ bits, int64_t e, int64_t n_pos, bool hit)
{
	const unsigned long long v = __ballot(hit);
	if ((threadIdx.x & 63) == 0 && e < n_pos) bits[e >> 6] = v; // (positions at and beyond n_pos vote 0)
}
template <typename W, int TILE, int BT, bool COOP>
__global__ __launch_bounds__(BT) void k_query(KParams P, const uint8_t *__restrict__ seq, int64_t n_pos,
                                              const unsigned int *__restrict__ bloom, unsigned long long *__restrict__ flags)
{
	constexpr int PW = (TILE + 64) / 32 + 2;
	__shared__ uint32_t planes[4 * PW];
	const W m = kmask<W>(P.k);
	const int64_t n_tiles = (n_pos + TILE - 1) / TILE;
	const int64_t tile = xcd_tile(blockIdx.x, n_tiles);
	if (tile >= n_tiles) return;
	build_planes<TILE, BT>(seq, nullptr, n_pos, tile * TILE, P.q, planes);
	__syncthreads();
	// One 64-byte block per query, fetched ONCE: four adjacent lanes load its four 16-byte quarters with one instruction (a single
	// line-sized request), each tests the bits that fall into its quarter, and the partial counts are added across the four lanes.
	// (A lane gathering its own four dwords pays nearly a full miss for each of them: so many lines are in flight per CU that the block
	// has left the caches before the next dword is asked for -- 22.1 ms per 306 M queries on a 16 GiB filter, 12.4 ms with one gather.)
	if (!COOP) {
#pragma unroll 4
		for (int j = 0; j < TILE / BT; ++j) {
			const int r = j * BT + threadIdx.x;
			const int64_t e = tile * TILE + r;
			W y0, y1; bool hi;
			uint8_t fl = 0;
			if (e < n_pos && kmer_at<W, TILE>(planes, r, P.k, m, y0, y1, hi)) {
				BloomAddr a = bloom_addr(bloom_hash<W>(P.k, y0, y1, m), P.bf_shift);
				const unsigned int *blk = bloom + a.blk * 16;
				uint32_t z = a.h1;
				// the first bit alone, the others only where it is set: a k-mer that is not in the filter mostly fails here
				uint32_t b = bloom_next(z, a.h2);
				uint32_t cnt = (blk[b >> 5] >> (b & 31)) & 1u;
				if (cnt) for (int t = 1; t < P.n_hashes; ++t) { b = bloom_next(z, a.h2); cnt += (blk[b >> 5] >> (b & 31)) & 1u; }
				fl = cnt == (uint32_t)P.n_hashes ? 2 : 1;
			}
			query_emit(flags, e, n_pos, fl == 2);
		}
		return;
	}
	const int lane = threadIdx.x & 63, member = lane & 3, grp_base = lane & ~3;
	for (int j = 0; j < TILE / BT; ++j) {
		const int r = j * BT + threadIdx.x;
		const int64_t e = tile * TILE + r;
		W y0, y1; bool hi;
		uint64_t my_blk = 0; uint32_t my_h = 0; // h1 | h2 << 9 | valid << 18
		if (e < n_pos && kmer_at<W, TILE>(planes, r, P.k, m, y0, y1, hi)) {
			BloomAddr a = bloom_addr(bloom_hash<W>(P.k, y0, y1, m), P.bf_shift);
			my_blk = a.blk; my_h = a.h1 | (a.h2 << 9) | (1u << 18);
		}
		uint32_t my_cnt = 0;
#pragma unroll
		for (int p = 0; p < 4; ++p) { // the query of lane grp_base + p
			const int src = grp_base + p;
			const uint32_t h = __shfl(my_h, src);
			const uint64_t blk = ((uint64_t)(uint32_t)__shfl((int)(my_blk >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)my_blk, src);
			uint32_t part = 0;
			if (h >> 18) {
				const uint4 v = *reinterpret_cast<const uint4 *>(bloom + blk * 16 + member * 4);
				uint32_t z = h & 511u; const uint32_t h2 = (h >> 9) & 511u;
				for (int t = 0; t < P.n_hashes; ++t) {
					const uint32_t b = bloom_next(z, h2);
					if ((int)(b >> 7) == member) { // this lane's quarter
						const uint32_t w = (b >> 5) & 3u;
						const uint32_t word = w == 0 ? v.x : w == 1 ? v.y : w == 2 ? v.z : v.w;
						part += (word >> (b & 31)) & 1u;
					}
				}
			}
			part += __shfl_xor(part, 1); part += __shfl_xor(part, 2);
			if (member == p) my_cnt = part;
		}
		query_emit(flags, e, n_pos, (my_h >> 18) && my_cnt == (uint32_t)P.n_hashes);
	}
}

// reads r: positions [off[r], off[r+1]-1) of the stream (the last byte of the span is the separator)
// k_query4 (round 5): the bloom query for n_hashes = 4 with the memory system's REQUEST RATE in mind.  Random 64-byte gathers into a 16 GiB
// array reach 48.8 G per second on this chip whatever their size between 16 and 128 bytes (scripts/probes/gather_probe.hip,
// profiles/round5_gather_probe.txt: the rate of HBM row activations, not bytes, is the limit); k_query's co-operative gather -- four lanes, four
// 16-byte quarters, one query after the other with the k-mer arithmetic of the next in between -- reached 22 G queries/s: it kept ONE gather in
// flight per wave most of the time.  Here a group of four lanes still serves its four queries together, but lane m fetches only the DWORD that
// holds position m of the query (the four dwords of a block leave as one request), tests one bit, and the group ANDs the four bits (quad
// permutes on the vector unit: no LDS crossbar); the loads of query j + 1 are issued before the bits of query j are looked at, so a wave
// has two queries' gathers in flight beside its arithmetic.  bbf.c:47-63 as it stands: all n_hashes bits set <=> hit.
template <int CTRL> __device__ __forceinline__ uint32_t dpp_quad(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); }
template <typename W, int TILE, int BT>
__global__ __launch_bounds__(BT) void k_query4(KParams P, const uint8_t *__restrict__ seq, int64_t n_pos,
                                               const unsigned int *__restrict__ bloom, unsigned long long *__restrict__ flags)
{
	constexpr int PW = (TILE + 64) / 32 + 2, S = TILE / BT;
	static_assert(S % 2 == 0, "queries are taken in pairs");
	__shared__ uint32_t planes[4 * PW];
	const W m = kmask<W>(P.k);
	const int64_t n_tiles = (n_pos + TILE - 1) / TILE;
	const int64_t tile = xcd_tile(blockIdx.x, n_tiles);
	if (tile >= n_tiles) return;
	build_planes<TILE, BT>(seq, nullptr, n_pos, tile * TILE, P.q, planes);
	__syncthreads();
	const uint32_t member = threadIdx.x & 3u, sh = (member & 1u) << 4;
	const bool upper = (member & 2u) != 0;
	// the owner's part: block and the four positions (two per word; bit 31 of the second: there is a k-mer) of the query at tile position j * BT + thread
	auto own = [&](int j, uint32_t &blk, uint32_t &pa, uint32_t &pb) {
		const int r = j * BT + (int)threadIdx.x;
		W y0, y1; bool hi;
		blk = 0; pa = 0; pb = 0; // (no k-mer here: the group fetches block 0's first dword for it, and nobody looks)
		if (tile * TILE + r < n_pos && kmer_at<W, TILE>(planes, r, P.k, m, y0, y1, hi)) {
			const BloomAddr a = bloom_addr(bloom_hash<W>(P.k, y0, y1, m), P.bf_shift);
			const B3Pos b = b3_positions(a.h1, a.h2);
			blk = (uint32_t)a.blk; pa = b.b0 | (b.b1 << 16); pb = b.b2 | (b.b3 << 16) | 0x80000000u;
		}
	};
	// the group's part: lane m fetches the dword of position m of each of the group's four queries
	auto fetch1 = [&](uint32_t qb, uint32_t qa, uint32_t qc, uint32_t &w, uint32_t &bit) {
		const uint32_t b = ((upper ? qc : qa) >> sh) & 511u;
		w = bloom[(uint64_t)qb * 16u + (b >> 5)]; bit = b & 31u;
	};
	auto fetch = [&](uint32_t blk, uint32_t pa, uint32_t pb, uint32_t (&w)[4], uint32_t (&bit)[4]) {
		fetch1(dpp_quad<0x00>(blk), dpp_quad<0x00>(pa), dpp_quad<0x00>(pb), w[0], bit[0]);
		fetch1(dpp_quad<0x55>(blk), dpp_quad<0x55>(pa), dpp_quad<0x55>(pb), w[1], bit[1]);
		fetch1(dpp_quad<0xAA>(blk), dpp_quad<0xAA>(pa), dpp_quad<0xAA>(pb), w[2], bit[2]);
		fetch1(dpp_quad<0xFF>(blk), dpp_quad<0xFF>(pa), dpp_quad<0xFF>(pb), w[3], bit[3]);
	};
	auto settle = [&](int j, uint32_t pb, const uint32_t (&w)[4], const uint32_t (&bit)[4]) {
		uint32_t mine = 0;
#pragma unroll
		for (uint32_t p = 0; p < 4; ++p) {
			uint32_t t = (w[p] >> bit[p]) & 1u;
			t &= dpp_quad<0xB1>(t); t &= dpp_quad<0x4E>(t); // all four bits of query p (bbf.c:60: every one of the n_hashes bits)
			if (member == p) mine = t;
		}
		const int64_t e = tile * TILE + j * BT + (int)threadIdx.x;
		query_emit(flags, e, n_pos, (pb >> 31) && mine);
	};
	uint32_t blkA, paA, pbA, wA[4], bitA[4], blkB, paB, pbB, wB[4], bitB[4];
	own(0, blkA, paA, pbA);
	fetch(blkA, paA, pbA, wA, bitA);
#pragma unroll 1
	for (int j = 0; j < S; j += 2) {
		own(j + 1, blkB, paB, pbB);
		fetch(blkB, paB, pbB, wB, bitB);
		settle(j, pbA, wA, bitA);
		if (j + 2 < S) { own(j + 2, blkA, paA, pbA); fetch(blkA, paA, pbA, wA, bitA); }
		settle(j + 1, pbB, wB, bitB);
	}
}

__global__ __launch_bounds__(256) void k_streak(int k, float min_frac, const unsigned long long *__restrict__ bits, const uint64_t *__restrict__ off,
                                                uint64_t n_reads, int32_t *__restrict__ out_start, int32_t *__restrict__ out_end)
{
	const uint64_t r = blockIdx.x * 256ull + threadIdx.x;
	if (r >= n_reads) return;
	const uint64_t p0 = off[r];
	const int len = (int)(off[r + 1] - p0) - 1;
	// correct.c:483-495 on the hit bits: a hit extends the run, anything else restarts it behind itself; t = run length << 32 | run start, the maximum
	// keeps the longest run and of equal ones the later.  The read's bits are taken 64 at a time and walked RUN by run (count-trailing-zeros on the
	// word and on its complement): a handful of steps per read instead of one per position, no memory access inside.
	unsigned long long mx = 0, t = 0;
	for (int i = 0; i < len; i += 64) {
		const uint64_t bp = p0 + (uint64_t)i;
		const unsigned long long lo = bits[bp >> 6], hi = (bp & 63) ? bits[(bp >> 6) + 1] : 0ULL; // (the word behind the read's last may be stale: masked by nb)
		const int sft = (int)(bp & 63), nb = len - i < 64 ? len - i : 64;
		unsigned long long w = sft ? (lo >> sft) | (hi << (64 - sft)) : lo;
		if (nb < 64) w &= (1ULL << nb) - 1ULL;
		int pos = 0;
		while (pos < nb) {
			const unsigned long long x = w >> pos;
			if (x & 1ULL) { // a run of hits
				int ones = ~x ? __builtin_ctzll(~x) : 64;
				if (ones > nb - pos) ones = nb - pos;
				t += (unsigned long long)ones << 32;
				mx = mx > t ? mx : t;
				pos += ones;
			} else { // everything up to the next hit restarts the run behind it
				int zeros = x ? __builtin_ctzll(x) : 64;
				if (zeros > nb - pos) zeros = nb - pos;
				pos += zeros;
				t = (unsigned long long)(i + pos);
			}
		}
	}
	int st = -1, en = -1;
	if ((mx >> 32) && (double)((mx >> 32) + (unsigned long long)k) / len > min_frac) { // correct.c:557 (min_frac is a float, bfc.h:21)
		st = (int)(uint32_t)mx; en = st + (int)(mx >> 32); st -= k - 1;
	}
	out_start[r] = st; out_end[r] = en;
}

// ------------------------------------------------------------------------------------------
// k-mer coverage of the corrector, bfc_ec_kcov (correct.c:96-117), for every read of a batch against the count table in HBM
//   k_occ : K1 + bfc_ch_kmer_occ (htab.c:85-99) for the k-mer ENDING at every position -> flag byte: bit0 solid_end
//           (count >= min_occ), bit1 high_end (high count >= min_occ+1) ; absent k-mers and non-k-mer positions give 0
//   k_cov : lcov / hcov of every base = number of solid (solid and high) k-mers covering it = the k flags that follow it;
//           packed like ecbase_t's bit-fields (correct.c:14-19): lcov | hcov<<6 | solid_end<<12 | high_end<<13

// bfc_ch_get on the device layout: probe the sub-table's region from the key's home slot until the key or an empty slot
__device__ __forceinline__ int table_get(const KParams &P, const unsigned long long *__restrict__ tab, uint64_t y0, uint64_t y1)
{
	uint64_t key;
	const uint32_t sub = ch_subkey(P.k, P.l_pre, y0, y1, key);
	const uint32_t cmask = (1u << P.tab_cshift) - 1;
	const unsigned long long *reg = tab + ((uint64_t)sub << P.tab_cshift);
	uint32_t pos = (uint32_t)(key >> 14) & cmask;
	for (uint32_t probe = 0; probe <= cmask; ++probe, pos = (pos + 1) & cmask) {
		const unsigned long long cur = reg[pos];
		if (cur == 0) return -1;
		if ((cur >> 14) == (key >> 14)) return (int)(cur & 0x3fff);
	}
	return -1;
}

template <typename W, int TILE, int BT>
__global__ __launch_bounds__(BT) void k_occ(KParams P, const uint8_t *__restrict__ seq, int64_t n_pos, int min_occ,
                                            const unsigned long long *__restrict__ tab, uint8_t *__restrict__ flags)
{
	constexpr int PW = (TILE + 64) / 32 + 2;
	__shared__ uint32_t planes[4 * PW];
	const W m = kmask<W>(P.k);
	const int64_t n_tiles = (n_pos + TILE - 1) / TILE;
	const int64_t tile = xcd_tile(blockIdx.x, n_tiles);
	if (tile >= n_tiles) return;
	build_planes<TILE, BT>(seq, nullptr, n_pos, tile * TILE, P.q, planes);
	__syncthreads();
#pragma unroll 4
	for (int j = 0; j < TILE / BT; ++j) {
		const int r = j * BT + threadIdx.x;
		const int64_t e = tile * TILE + r;
		if (e >= n_pos) continue;
		W y0, y1; bool hi;
		uint8_t fl = 0;
		if (kmer_at<W, TILE>(planes, r, P.k, m, y0, y1, hi)) {
			const int occ = table_get(P, tab, (uint64_t)y0, (uint64_t)y1);
			if (occ >= 0) fl = (uint8_t)(((occ & 0xff) >= min_occ ? 1 : 0) | ((occ >> 8 & 0x3f) >= min_occ + 1 ? 2 : 0));
		}
		flags[e] = fl;
	}
}

// TILE positions per workgroup; the flags of the TILE+64 positions from the tile's start become two bit rows in LDS
template <int TILE, int BT>
__global__ __launch_bounds__(BT) void k_cov(int k, const uint8_t *__restrict__ flags, int64_t n_pos, uint16_t *__restrict__ out)
{
	constexpr int NW = TILE / 64 + 1;
	__shared__ unsigned long long solid[NW + 1], high[NW + 1];
	const int64_t n_tiles = (n_pos + TILE - 1) / TILE;
	const int64_t tile = xcd_tile(blockIdx.x, n_tiles);
	if (tile >= n_tiles) return;
	const int64_t t0 = tile * TILE;
	for (int r = threadIdx.x; r < NW * 64; r += BT) { // a wave's 64 flags -> one word of each row
		const int64_t e = t0 + r;
		const uint8_t fl = e < n_pos ? flags[e] : 0;
		const unsigned long long bs = __ballot(fl & 1), bh = __ballot((fl & 3) == 3);
		if ((threadIdx.x & 63) == 0) { solid[r >> 6] = bs; high[r >> 6] = bh; }
	}
	if (threadIdx.x == 0) { solid[NW] = 0; high[NW] = 0; }
	__syncthreads();
	const unsigned long long wm = k >= 64 ? ~0ULL : (1ULL << k) - 1;
	for (int r = threadIdx.x; r < TILE; r += BT) {
		const int64_t e = t0 + r;
		if (e >= n_pos) break;
		const int w = r >> 6, sh = r & 63;
		unsigned long long a = solid[w] >> sh, b = high[w] >> sh;
		if (sh) { a |= solid[w + 1] << (64 - sh); b |= high[w + 1] << (64 - sh); }
		const uint8_t fl = flags[e];
		out[e] = (uint16_t)(__popcll(a & wm) | __popcll(b & wm) << 6 | (fl & 1) << 12 | (fl >> 1 & 1) << 13);
	}
}

// ------------------------------------------------------------------------------------------
// host-callable launchers (C++ linkage, used by bfcg_ctx.hip)

namespace bfcg {

static inline int grid_for(int64_t n_tiles, int cap) { return (int)(n_tiles < cap ? (n_tiles > 0 ? n_tiles : 1) : cap); }

#define TILE1 BFCG_TILE1
#define BT1 256
// k_scatter1's tile by record size: 4096 positions, 3072 for 20-byte records -- two workgroups of 512 threads per CU (56 / 73 / 68 KiB of LDS).
// (Three workgroups of 3584 positions, 768 threads on 4608, 1024 threads, one memory-in wave beside seven hashing waves: all measured, none
// faster -- the kernel's floor is its skeleton of LDS ranks, scan, staging and barriers, not the hashing: DESIGN.md section 6b.)
// Round 5: 16-byte records take 3072 too -- with 2^10 level-1 buckets (config c5: k = 51, -b37) the stage of 4096 records, their bucket bytes and
// the counters were 92 KiB: ONE workgroup per CU, 17 ps per k-mer against 8 for c3's 12-byte records; 3072 positions: 75 KiB, two per CU.
template <int RW> struct S1 { static constexpr int BT = 512, TILE = RW == 3 ? 4096 : 3072; };
static_assert(S1<3>::TILE == 4096 && S1<4>::TILE == 3072 && S1<5>::TILE == 3072, "bfcg_tile1_of_rw (bfcg_internal.h) sizes the host's buffers");
#define TILE2 BFCG_TILE2
#define BT2 512
// Round 6: level 2 at 2^10 regions per bucket (config c4's -b37).  A tile of 4096 records spread over 1024 regions leaves as runs of FOUR records --
// 48 bytes, a sector and a half -- and the kernel that streams at 4.8-5.0 TB/s at c3's 2^9 regions (runs of eight = 96 bytes = three whole sectors) moved
// c4e's records at 3.5 (profiles/round5_c4e.md: WRITE_SIZE 1.09 x the records, so the bytes were right and the rate was not).  Tiles of 8192 records
// with 1024 threads restore the run length: 96 KiB of stage + 8 KiB of counters, one workgroup per CU.
#define TILE2_BIG 8192
#define BT2_BIG 1024

// BFCG_DEBUG_SYNC=1: wait behind every stage's launches and say which one the device failed in (a memory fault names no kernel)
static void dbg_sync(hipStream_t st, const char *what)
{
	static int on = -1;
	if (on < 0) on = getenv("BFCG_DEBUG_SYNC") != 0;
	if (!on) return;
	fprintf(stderr, "[D::sync] %s ...\n", what); fflush(stderr);
	const hipError_t e = hipStreamSynchronize(st);
	fprintf(stderr, "[D::sync] %s: %s\n", what, hipGetErrorString(e)); fflush(stderr);
}

// k_scatter1<..., FAST>: the level-1 bucket is a bit field of y0's low word and a 12-byte record packs from halves (pack3_geom's conditions)
static inline bool scatter1_fast(const KParams &P)
{
	const int a = P.k - P.rec_n;
	return P.k >= P.bf_shift - 9 && a >= 1 && a <= 31 && a + P.k >= 32 && (P.rec_n == 0 || P.rec_lo + P.rec_n <= 31) && !getenv("BFCG_NO_FAST_K1");
}

// stage A: bases -> records grouped by (global) level-1 bucket in `out1`; B.start1[2^F1+1] = bucket starts
template <typename W, int RW>
static void run_stage_a_t(const KParams &P, const BatchBufs &B, const uint8_t *seq, const uint8_t *qual, int64_t n_pos, uint32_t *out1, hipStream_t st, hipEvent_t *ev)
{
	constexpr int T1 = S1<RW>::TILE, BTS1 = S1<RW>::BT, T2 = RW == 5 ? 2048 : RW == 4 ? 3072 : TILE2;
	const int nb1 = 1 << P.F1;
	const int64_t tiles1 = (n_pos + T1 - 1) / T1;
	const int n_chunks = (int)((tiles1 + SCAN_CH - 1) / SCAN_CH);
	if (ev) hipEventRecord(ev[0], st);
	const unsigned g1 = (unsigned)(((tiles1 + 7) / 8) * 8); // one block per tile, dealt XCD-contiguously
	hipLaunchKernelGGL((k_hist1<W, T1, BT1>), dim3(g1), dim3(BT1), 0, st, P, seq, qual, n_pos, B.rows1, B.stats);
	hipLaunchKernelGGL(k_colsum, dim3(n_chunks), dim3(256), 0, st, B.rows1, (int)tiles1, nb1, B.chunk1);
	hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(nb1 < 64 ? 64 : nb1), 0, st, B.chunk1, n_chunks, nb1, B.start1, B.row_base, T2);
	hipLaunchKernelGGL(k_apply, dim3(n_chunks), dim3(256), 0, st, B.rows1, (int)tiles1, nb1, B.chunk1);
	if (ev) hipEventRecord(ev[1], st);
	hipLaunchKernelGGL((k_scatter1<W, RW, T1, BTS1>), dim3(g1), dim3(BTS1), (size_t)T1 * (RW * 4 + ((P.rec_n > 0 && RW == 4) ? 2 : 0)) + (size_t)8 * nb1, st, P, seq, qual, n_pos, B.rows1, out1, OnePass{nullptr, 0u, nullptr, nullptr, 0u, 0u, 0u, 0u});
	if (ev) hipEventRecord(ev[2], st);
}

// one-pass stage A (K1 once): cursors cleared, K1 + level-1 scatter into the slabs, the slabs described as segments for level 2
template <typename W, int RW>
static void run_stage_a_onepass_t(const KParams &P, const BatchBufs &B, const uint8_t *seq, const uint8_t *qual, int64_t n_pos, uint32_t *out1, hipStream_t st, hipEvent_t *ev)
{
	constexpr int T1 = S1<RW>::TILE, BTS1 = S1<RW>::BT, T2 = RW == 5 ? 2048 : RW == 4 ? 3072 : TILE2;
	const int nb1 = 1 << P.F1;
	const int64_t tiles1 = (n_pos + T1 - 1) / T1;
	if (ev) hipEventRecord(ev[0], st);
	hipMemsetAsync(B.op_cursor, 0, ((size_t)8 * nb1 * 32 + 32) * sizeof(uint32_t), st); // the slabs' cursors and, behind them, k_scatter1's tile counter
	hipMemsetAsync(B.op_flags, 0, 4 * sizeof(uint32_t), st); // this slot's overflow flags (its previous batch's stage B and flag copy are complete: the caller waited)
	if (ev) hipEventRecord(ev[1], st);
	unsigned g1 = (unsigned)(((tiles1 + 7) / 8) * 8);
	const size_t lds1 = (size_t)T1 * (RW * 4 + ((P.rec_n > 0 && RW == 4) ? 2 : 0)) + (size_t)16 * nb1;
	OnePass OP{B.op_cursor, B.op_cap, B.op_flags, B.stats, 1u, B.op_own_lo, B.op_own_n, B.op_own_delta};
	{ // as many workgroups as are resident at once (a CU's 160 KiB of LDS, at most 2048 threads), each walking its tiles; a multiple of 8 (XCDs)
		static int n_cu = 0;
		if (!n_cu) { hipDeviceProp_t pr; int dev = 0; n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256; }
		int per_cu = (int)((size_t)160 * 1024 / (lds1 + 4400)); if (per_cu > 2) per_cu = 2; if (per_cu < 1) per_cu = 1; // (the kernel is built for 4 waves per SIMD)
		const char *e = getenv("BFCG_S1_WGS"); if (e && atoi(e) > 0) per_cu = atoi(e);
		const unsigned gp = (unsigned)(n_cu * per_cu) & ~7u;
		if (gp >= 8 && gp < g1) g1 = gp;
	}
	if (B.cnt_live) { // chunked reservations leave dead records behind, which level 2 skips (both its variants since round 4; the two-pass one needs the
		// regions' counts for it: cnt_live).  What the g1 / 8 workgroups of an XCD
		// leave unused in a slab -- half a chunk each on average -- stays below a quarter of the slab's head room (a ninth of its capacity).
		const char *e = getenv("BFCG_S1_CHUNK"); // (tests force chunk sizes on small draws)
		const int forced = e ? atoi(e) : 0;
		uint32_t ch = 32;
		while (ch > 1 && (uint64_t)(g1 / 8) * (ch / 2) * 36 > B.op_cap) ch >>= 1;
		OP.chunk = forced > 0 ? (uint32_t)forced : ch;
	}
	if constexpr (RW == 3) {
		WcPlan wc;
		if (B.cnt_live && scatter1_fast(P) && scatter1_wc_plan(P, OP, 3, n_pos, &wc)) run_scatter1_wc(P, seq, qual, n_pos, out1, OP, wc, st); // (round 5: write-combining buffers in LDS)
		else if (scatter1_fast(P)) {
			if (sizeof(W) == 8 && P.k == 33) hipLaunchKernelGGL((k_scatter1<W, RW, T1, BTS1, true, sizeof(W) == 8 ? 33 : 0, true>), dim3(g1), dim3(BTS1), lds1, st, P, seq, qual, n_pos, (const uint32_t *)nullptr, out1, OP);
			else hipLaunchKernelGGL((k_scatter1<W, RW, T1, BTS1, true, 0, true>), dim3(g1), dim3(BTS1), lds1, st, P, seq, qual, n_pos, (const uint32_t *)nullptr, out1, OP);
		} else hipLaunchKernelGGL((k_scatter1<W, RW, T1, BTS1, true>), dim3(g1), dim3(BTS1), lds1, st, P, seq, qual, n_pos, (const uint32_t *)nullptr, out1, OP);
	} else {
		WcPlan wc;
		if (RW == 4 && B.cnt_live && scatter1_wc_plan(P, OP, 4, n_pos, &wc)) run_scatter1_wc(P, seq, qual, n_pos, out1, OP, wc, st); // (16-byte records at 2^10 buckets: config c5)
		else hipLaunchKernelGGL((k_scatter1<W, RW, T1, BTS1, true>), dim3(g1), dim3(BTS1), lds1, st, P, seq, qual, n_pos, (const uint32_t *)nullptr, out1, OP);
	}
	dbg_sync(st, "k_scatter1 (one pass)");
	uint32_t *sg = B.op_seg;
	hipLaunchKernelGGL(k_seg_setup, dim3(1), dim3(1024), 0, st, P, B.op_cursor, B.op_cap, B.op_flags, (RW == 3 && P.l2_big) ? TILE2_BIG : T2, sg, sg + 8 * nb1, sg + 16 * nb1, sg + 24 * nb1 + 1);
	dbg_sync(st, "k_seg_setup");
	if (ev) hipEventRecord(ev[2], st);
}

// k_scatter2<..., FAST2>: the region's low F2 bits are bits [R, R + F2) of a 12-byte record's first word
static inline bool scatter2_fast(const KParams &P)
{
	const int a = P.k - P.rec_n;
	return P.k >= P.bf_shift - 9 && P.F2 > 0 && P.R + P.F2 <= 31 && a >= P.R + P.F2 && P.rec_lo == P.R + P.F2 && !getenv("BFCG_NO_FAST_S2");
}

// k_bloom<..., F3>: dec3_geom's conditions, on the host
static inline bool bloom_fast3(const KParams &P)
{
	const int a = P.k - P.rec_n, up = P.rec_n ? P.rec_lo : P.bf_shift - 9, sh_x = P.k - (P.bf_shift - 9), sh_y1 = P.k - P.F;
	return P.k >= P.bf_shift - 9 && P.bf_shift + 9 <= 2 * P.k && a >= 1 && a <= 31 && a + P.k >= 32 && (P.rec_n == 0 || P.rec_lo + P.rec_n <= 31)
	       && up <= a && sh_x >= 0 && sh_x <= 31 && sh_y1 >= 0 && sh_y1 <= 32 && (P.rec_n == 0 || P.rec_lo + P.rec_n == P.bf_shift - 9) && P.R <= up && !getenv("BFCG_NO_FAST_BLOOM");
}

// a CU's LDS holds 160 KB / segment size workgroups: keep its 2048 lanes busy whatever that number is (c4's 64 KiB segments at 256
// threads per workgroup: commit 2.89 s, at 1024: 1.44 s)
bool bloom3fm_geometry_ok(const KParams &P)
{
	const int up = P.rec_n ? P.rec_lo : P.bf_shift - 9, a = P.k - P.rec_n, io = a + P.k + 1;
	return bfcg_rec_dwords(P.k, P.rec_n) == 4 && P.n_hashes == 4 && P.R <= 8 && P.k >= P.bf_shift + 9 && (P.rec_n == 0 || P.rec_lo + P.rec_n == P.bf_shift - 9)
	       && P.R <= up && up >= 1 && up <= 31 && a >= up + 18 && io + 32 <= 128;
}
bool bloom3_geometry_ok(const KParams &P) { return bfcg_rec_dwords(P.k, P.rec_n) == 3 && P.n_hashes == 4 && P.R <= 8 && bloom_fast3(P); }

// one workgroup per (region, block of its segment): one-dimensional `f * blocks + blk` -- a region's blocks back to back -- unless that
// exceeds HIP's 2^32 threads per grid dimension (tiny test blocks on large filters), then regions x blocks
static dim3 seg_grid(const KParams &P, uint32_t n_fine, uint32_t bt)
{
	const uint32_t ub = (uint32_t)(P.seg_shift - P.seg_blk);
	const uint64_t wgs = (uint64_t)n_fine << ub;
	static const bool force2d = getenv("BFCG_SEG_GRID2D") != nullptr; // (A/B only)
	if (wgs * bt < (1ULL << 32) && !force2d) return dim3((unsigned)wgs, 1u);
	return dim3(n_fine, 1u << ub);
}
static void launch_commit_seg(const KParams &P, const BloomArgs &A, int nfine, hipStream_t st)
{
	const dim3 grid = seg_grid(P, (uint32_t)nfine, P.seg_blk >= 13 ? 1024u : P.seg_blk == 12 ? 512u : 256u); // one workgroup per (region, block of its segment)
	if (P.seg_blk >= 13) hipLaunchKernelGGL((k_commit_seg<1024>), grid, dim3(1024), (size_t)8 << P.seg_blk, st, P, A);
	else if (P.seg_blk == 12) hipLaunchKernelGGL((k_commit_seg<512>), grid, dim3(512), (size_t)12 << P.seg_blk, st, P, A); // (+ 4 bytes of counters per slot)
	else hipLaunchKernelGGL((k_commit_seg<256>), grid, dim3(256), (size_t)12 << P.seg_blk, st, P, A);
	dbg_sync(st, "k_commit_seg");
}
void run_commit_pages(const KParams &P, const BatchBufs &B, uint32_t n_fine, uint32_t pages, hipStream_t st)
{
	BloomArgs A;
	memset(&A, 0, sizeof(A));
	A.stats = B.stats; A.tab_ovf = B.tab_ovf; A.tab_ovf_cap = B.tab_ovf_cap; A.ovf_cnt = B.stats + (size_t)ST_SLOTS * ST_N; A.seg_tab = B.seg_tab; A.n_fine = n_fine;
	A.ho = B.ho; A.ho_stride = B.ho_stride; A.ho_cur = B.ho_cur; A.ho_mark = B.ho_mark; A.ho_mark_stride = B.ho_mark_stride; A.ho_pages = pages; A.ho_keys = B.ho_keys;
	launch_commit_seg(P, A, (int)n_fine, st);
}

// stage B: records in `in1` as n_seg segments (seg_beg/seg_end, row_base over segments, bucket_start over the
// nb_loc = n_seg/segs_per_bucket owned level-1 buckets) -> fine buckets -> bloom regions -> table
template <typename W, int RW>
static void run_stage_b_t(const KParams &P, const BatchBufs &B, const uint32_t *in1, const uint32_t *seg_beg, const uint32_t *seg_end, int n_seg,
                          int segs_per_bucket, const uint32_t *row_base, const uint32_t *bucket_start, uint64_t n_rec_bound, hipStream_t st, hipEvent_t *ev)
{
	const int nb_loc = n_seg / segs_per_bucket, nfine = nb_loc << P.F2;
	const uint32_t *fine_recs = in1; const uint32_t *fine_start = bucket_start;
	if (P.F2 > 0) {
		constexpr int T2 = RW == 5 ? 2048 : RW == 4 ? 3072 : TILE2; // = bfcg_tile_of(k)
		// rows of level 2 <= records/T2 + one ragged row per segment; surplus blocks exit at once
		const unsigned g2 = (unsigned)(((n_rec_bound / T2 + n_seg + 1 + 7) / 8) * 8);
		bool big_done = false;
		if constexpr (RW == 3) {
			if (B.cap2 && P.l2_big && scatter2_fast(P)) { // (the segments' rows were counted in tiles of TILE2_BIG by this batch's k_seg_setup)
				const unsigned g2b = (unsigned)(((n_rec_bound / TILE2_BIG + n_seg + 1 + 7) / 8) * 8);
				hipMemsetAsync(B.cnt2, 0, sizeof(uint32_t) * (size_t)nfine, st);
				hipLaunchKernelGGL((k_scatter2<W, RW, TILE2_BIG, BT2_BIG, true, true>), dim3(g2b), dim3(BT2_BIG), (size_t)TILE2_BIG * (RW * 4) + ((size_t)4 << P.F2), st, P, in1, seg_beg, seg_end, n_seg, segs_per_bucket, row_base,
				                   (const uint32_t *)nullptr, (uint32_t *)B.recs2, OnePass2{B.cnt2, B.cap2, B.op_flags});
				big_done = true;
			}
		}
		if (big_done) ;
		else if (B.cap2) { // one pass: region slabs and cursors
			hipMemsetAsync(B.cnt2, 0, sizeof(uint32_t) * (size_t)nfine, st);
			if ((RW == 3 || RW == 4) && scatter2_fast(P)) // (round 5: 16-byte records too -- their first word holds the same low bits of y0)
				hipLaunchKernelGGL((k_scatter2<W, RW, T2, BT2, true, RW == 3 || RW == 4>), dim3(g2), dim3(BT2), (size_t)T2 * (RW * 4) + ((size_t)4 << P.F2), st, P, in1, seg_beg, seg_end, n_seg, segs_per_bucket, row_base,
				                   (const uint32_t *)nullptr, (uint32_t *)B.recs2, OnePass2{B.cnt2, B.cap2, B.op_flags});
			else
			hipLaunchKernelGGL((k_scatter2<W, RW, T2, BT2, true>), dim3(g2), dim3(BT2), (size_t)T2 * (RW * 4) + ((size_t)4 << P.F2), st, P, in1, seg_beg, seg_end, n_seg, segs_per_bucket, row_base,
			                   (const uint32_t *)nullptr, (uint32_t *)B.recs2, OnePass2{B.cnt2, B.cap2, B.op_flags});
		} else {
			hipLaunchKernelGGL((k_hist2<W, RW, T2, BT2>), dim3(g2), dim3(BT2), 0, st, P, in1, seg_beg, seg_end, n_seg, segs_per_bucket, row_base, B.rows2);
			hipLaunchKernelGGL(k_scan2, dim3(nb_loc), dim3((1 << P.F2) < 64 ? 64 : (1 << P.F2)), 0, st, P, bucket_start, segs_per_bucket, row_base, B.rows2, B.start2, B.cnt_live);
			hipLaunchKernelGGL((k_scatter2<W, RW, T2, BT2>), dim3(g2), dim3(BT2), (size_t)T2 * (RW * 4) + ((size_t)4 << P.F2), st, P, in1, seg_beg, seg_end, n_seg, segs_per_bucket, row_base, B.rows2,
			                   (uint32_t *)B.recs2, OnePass2{nullptr, 0u, nullptr});
		}
		fine_recs = (const uint32_t *)B.recs2; fine_start = B.start2;
		dbg_sync(st, "level 2");
	}
	if (ev) hipEventRecord(ev[3], st);
	BloomArgs A;
	memset(&A, 0, sizeof(A));
	A.recs = fine_recs; A.start = fine_start; A.bloom = B.bloom; A.bloom_hi = B.bloom_hi; A.table = B.table; A.stats = B.stats;
	A.tab_ovf = B.tab_ovf; A.tab_ovf_cap = B.tab_ovf_cap; A.ovf_cnt = B.stats + (size_t)ST_SLOTS * ST_N; A.pool = B.pool; A.pool_slices = B.pool_slices; A.seen_out = B.seen_out;
	A.agg_out = B.agg_out; A.agg_cnt = B.agg_cnt; A.n_fine = (uint32_t)nfine; A.stream_out = nullptr; A.seg_tab = nullptr;
	A.ord.first = B.tab_first; A.ord.sub_last = B.sub_last; A.batch_hi = B.batch_hi;
	A.cnt2 = nullptr; A.cap2 = 0; A.flags = B.op_flags; A.sticky = B.op_sticky; // (op_flags: NULL unless this batch went through the one-pass partition)
	if (P.F2 > 0 && B.cap2) { A.cnt2 = B.cnt2; A.cap2 = B.cap2; }
	else if (P.F2 > 0 && B.cnt_live) A.cnt2 = B.cnt_live; // two passes: the regions' counts beside their starts (region_list)
	size_t lds = (size_t)bloom_lds_bytes(P);
	if (P.filter_mode && B.bloom_hi) { // both filters' slices in LDS, nothing to hand over
		if (RW == 4 && P.b3fm) run_bloom3fm(P, A, nfine, lds, st); // (bfcg_bloom3.hip)
		else if (P.n_hashes == 4 && P.bloom_bt == 1024) hipLaunchKernelGGL((k_bloom<W, RW, 1024, 2, 4, false, true>), dim3(nfine), dim3(1024), lds, st, P, A);
		else if (P.n_hashes == 4) hipLaunchKernelGGL((k_bloom<W, RW, 512, 4, 4, false, true>), dim3(nfine), dim3(512), lds, st, P, A);
		else hipLaunchKernelGGL((k_bloom<W, RW, 512, 4, 0, false, true>), dim3(nfine), dim3(512), lds, st, P, A);
	} else if (P.seg && B.seg_tab && (B.stream_out || B.ho)) { // region-owned table segments: seen k-mers are handed to k_commit_seg, one workgroup per region
		A.stream_out = B.stream_out; A.seg_tab = B.seg_tab; A.table = nullptr; A.agg_out = nullptr;
		A.ho = B.ho; A.ho_stride = B.ho_stride; A.ho_cur = B.ho_cur; A.ho_mark = B.ho_stride ? B.ho_mark + (size_t)B.ho_page * B.ho_mark_stride : nullptr;
		bool f3 = false;
		if constexpr (RW == 3) {
			if (P.b3 && (P.b3_cold || !P.dedupe)) { f3 = true; run_bloom3(P, A, nfine, lds, st); } // (bfcg_bloom3.hip; without its COLD mode, cold batches with copies resolved by class stay with k_bloom)
			else if (P.n_hashes == 4 && bloom_fast3(P)) { f3 = true; hipLaunchKernelGGL((k_bloom<W, RW, 512, 4, 4, false, false, true, true, true>), dim3(nfine), dim3(512), lds, st, P, A); }
		}
		if (f3) ;
		else if (P.n_hashes == 4) hipLaunchKernelGGL((k_bloom<W, RW, 512, 4, 4, false, false, true, true>), dim3(nfine), dim3(512), lds, st, P, A);
		else hipLaunchKernelGGL((k_bloom<W, RW, 512, 4, 0, false, false, true, true>), dim3(nfine), dim3(512), lds, st, P, A);
		if (ev) hipEventRecord(ev[4], st);
		dbg_sync(st, "k_bloom");
		if (B.ho_stride == 0 || B.ho_commit) { // the log's pages so far (or this batch's entries at its records' offsets) into the segments
			A.ho_mark = B.ho_mark; A.ho_mark_stride = B.ho_mark_stride; A.ho_pages = B.ho_page + 1; A.ho_keys = B.ho_keys;
			launch_commit_seg(P, A, nfine, st);
		}
		if (A.flags) hipLaunchKernelGGL(k_seal, dim3(1), dim3(1), 0, st, B.op_flags, B.op_sticky);
		if (ev) hipEventRecord(ev[5], st);
		return;
	} else if (B.stream && B.stream_out && !P.track && P.n_hashes == 4) { // low-multiplicity batches: no aggregation (ctx decides, see bfcg_ctx.hip)
		A.stream_out = B.stream_out;
		hipLaunchKernelGGL((k_bloom<W, RW, 512, 4, 4, false, false, true>), dim3(nfine), dim3(512), lds, st, P, A);
		if (ev) hipEventRecord(ev[4], st);
		dbg_sync(st, "k_bloom");
		hipLaunchKernelGGL((k_commit_stream<W, RW>), dim3((unsigned)((nfine + 4 * COMMIT_RPW - 1) / (4 * COMMIT_RPW))), dim3(256), 0, st, P, A);
		if (A.flags) hipLaunchKernelGGL(k_seal, dim3(1), dim3(1), 0, st, B.op_flags, B.op_sticky);
		if (ev) hipEventRecord(ev[5], st);
		return;
	} else if (P.track) { // order stamps for the byte-identical dump: its own instantiation, so that the default path pays nothing for it
		if (P.n_hashes == 4) hipLaunchKernelGGL((k_bloom<W, RW, 512, 4, 4, true>), dim3(nfine), dim3(512), lds, st, P, A);
		else hipLaunchKernelGGL((k_bloom<W, RW, 512, 4, 0, true>), dim3(nfine), dim3(512), lds, st, P, A);
	} else if (P.n_hashes == 4) {
		if (P.bloom_bt == 1024) hipLaunchKernelGGL((k_bloom<W, RW, 1024, 2, 4, false>), dim3(nfine), dim3(1024), lds, st, P, A);
		else hipLaunchKernelGGL((k_bloom<W, RW, 512, 4, 4, false>), dim3(nfine), dim3(512), lds, st, P, A);
	} else hipLaunchKernelGGL((k_bloom<W, RW, 512, 4, 0, false>), dim3(nfine), dim3(512), lds, st, P, A);
	if (ev) hipEventRecord(ev[4], st);
	if (B.agg_out) {
		const uint64_t slots = (uint64_t)nfine * P.ag_cap;
		const unsigned gs = (unsigned)((slots + 255) / 256), gw = (unsigned)((nfine + 4 * COMMIT_RPW - 1) / (4 * COMMIT_RPW)); // 4 waves per workgroup
		if (nfine > (1 << 18)) {
			if (P.track) hipLaunchKernelGGL((k_commit<W, true, true>), dim3(gw), dim3(256), 0, st, P, A);
			else hipLaunchKernelGGL((k_commit<W, false, true>), dim3(gw), dim3(256), 0, st, P, A);
		} else {
			if (P.track) hipLaunchKernelGGL((k_commit<W, true, false>), dim3(gs), dim3(256), 0, st, P, A);
			else hipLaunchKernelGGL((k_commit<W, false, false>), dim3(gs), dim3(256), 0, st, P, A);
		}
	}
	if (A.flags) hipLaunchKernelGGL(k_seal, dim3(1), dim3(1), 0, st, B.op_flags, B.op_sticky);
	if (ev) hipEventRecord(ev[5], st);
}

// W: 32-bit k-mer arithmetic up to k = 32; RW: dwords per record (bfcg_rec_dwords: 12 bytes up to 2k - rec_n + 33 <= 96 bits, 16 up to 128, else 20)
#define DISPATCH_W(fn, ...) do { const int rw_ = bfcg_rec_dwords(P.k, P.rec_n); \
	if (P.k <= 32) { if (rw_ == 3) fn<uint32_t, 3>(__VA_ARGS__); else fn<uint32_t, 4>(__VA_ARGS__); } \
	else if (rw_ == 3) fn<uint64_t, 3>(__VA_ARGS__); else if (rw_ == 4) fn<uint64_t, 4>(__VA_ARGS__); else fn<uint64_t, 5>(__VA_ARGS__); } while (0)

void run_stage_a(const KParams &P, const BatchBufs &B, const uint8_t *seq, const uint8_t *qual, int64_t n_pos, uint64_t *out1, hipStream_t st, hipEvent_t *ev)
{ DISPATCH_W(run_stage_a_t, P, B, seq, qual, n_pos, (uint32_t *)out1, st, ev); }

void run_stage_a_onepass(const KParams &P, const BatchBufs &B, const uint8_t *seq, const uint8_t *qual, int64_t n_pos, uint64_t *out1, hipStream_t st, hipEvent_t *ev)
{ DISPATCH_W(run_stage_a_onepass_t, P, B, seq, qual, n_pos, (uint32_t *)out1, st, ev); }

// a group's slab mode without host sizes (kernels above): the source's rows behind its one-pass stage A; the owner's segment arrays from all sources' rows
void run_pack_rows(const KParams &P, const BatchBufs &B, int n_ranks, uint32_t row_w, uint32_t *rows, hipStream_t st)
{
	const int nb1 = 1 << P.F1;
	hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)n_ranks), dim3(256), 0, st, B.op_seg + (size_t)8 * nb1, B.op_flags, nb1 / n_ranks, B.op_cap, row_w, rows);
}
void run_seg_setup_mg(const KParams &P, int rw_dwords, const uint32_t *rows, uint32_t row_w, int n_ranks, int s_lo, int s_hi, uint32_t cap, uint32_t *seg, unsigned long long *total, hipStream_t st)
{
	const int nb1 = 1 << P.F1, nb_loc = nb1 / n_ranks, n_seg = nb1 * 8;
	hipLaunchKernelGGL(k_seg_setup_mg, dim3(1), dim3(1024), 0, st, rows, row_w, n_ranks, s_lo, s_hi, nb_loc, cap, rw_dwords == 5 ? 2048 : rw_dwords == 4 ? 3072 : TILE2, seg, seg + n_seg, seg + 2 * n_seg, seg + 3 * n_seg + 1, total);
}

void run_stage_b(const KParams &P, const BatchBufs &B, const uint64_t *in1, const uint32_t *seg_beg, const uint32_t *seg_end, int n_seg, int segs_per_bucket,
                 const uint32_t *row_base, const uint32_t *bucket_start, uint64_t n_rec_bound, hipStream_t st, hipEvent_t *ev)
{ DISPATCH_W(run_stage_b_t, P, B, (const uint32_t *)in1, seg_beg, seg_end, n_seg, segs_per_bucket, row_base, bucket_start, n_rec_bound, st, ev); }

// single GPU: both stages back to back, segment = level-1 bucket, everything stays on the device
void run_batch(const KParams &P, const BatchBufs &B, const uint8_t *seq, const uint8_t *qual, int64_t n_pos, hipStream_t st, hipEvent_t *ev)
{
	run_stage_a(P, B, seq, qual, n_pos, B.recs1, st, ev);
	run_stage_b(P, B, B.recs1, B.start1, B.start1 + 1, 1 << P.F1, 1, B.row_base, B.start1, (uint64_t)n_pos, st, ev);
}

int bloom_lds_bytes(const KParams &P)
{
	const size_t second = P.filter_mode ? ((size_t)64 << P.R) : P.seg ? 0 : (size_t)P.ag_cap * ((P.k > 32 ? 24 : 16) + (P.track ? 8 : 0)); // second filter's slice or aggregation table
	if (P.b3fm) return (int)(((size_t)128 << P.R) + ((size_t)2 << P.R) * 4 + 16 + (size_t)P.list_cap * 10 + 16); // k_bloom3fm: both slices, block counters + offsets, 10-byte entries
	if (P.b3 && P.b3_cold) return (int)(((size_t)64 << P.R) + ((size_t)2 << P.R) * 4 + 16 + (size_t)P.list_cap * 12 + 16); // k_bloom3<.., COLD>: region, block counters + offsets, 12-byte entries
	return (int)(((size_t)64 << P.R) + (size_t)P.fs_cap * 4 + second + (size_t)P.list_cap * (P.b3 ? 10 : 8) + 16);
}

template <typename W, int RW> static hipError_t set_attr_t(int lds)
{
	hipError_t e;
	constexpr int T1 = S1<RW>::TILE, BTS1 = S1<RW>::BT, T2 = RW == 5 ? 2048 : RW == 4 ? 3072 : TILE2;
	e = hipFuncSetAttribute((const void *)k_scatter1<W, RW, T1, BTS1>, hipFuncAttributeMaxDynamicSharedMemorySize, T1 * (RW * 4 + 2) + 8 * BFCG_MAXB); if (e != hipSuccess) return e;
	e = hipFuncSetAttribute((const void *)k_scatter1<W, RW, T1, BTS1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, T1 * (RW * 4 + 2) + 16 * BFCG_MAXB); if (e != hipSuccess) return e;
	if constexpr (RW == 3) {
		e = hipFuncSetAttribute((const void *)k_scatter1<W, RW, T1, BTS1, true, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, T1 * (RW * 4 + 2) + 16 * BFCG_MAXB); if (e != hipSuccess) return e;
		e = hipFuncSetAttribute((const void *)k_scatter1<W, RW, T1, BTS1, true, sizeof(W) == 8 ? 33 : 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, T1 * (RW * 4 + 2) + 16 * BFCG_MAXB); if (e != hipSuccess) return e;
	}
	e = hipFuncSetAttribute((const void *)k_scatter2<W, RW, T2, BT2>, hipFuncAttributeMaxDynamicSharedMemorySize, T2 * (RW * 4) + 8 * BFCG_MAXB); if (e != hipSuccess) return e;
	e = hipFuncSetAttribute((const void *)k_scatter2<W, RW, T2, BT2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, T2 * (RW * 4) + 8 * BFCG_MAXB); if (e != hipSuccess) return e;
	if (RW == 3 || RW == 4) { e = hipFuncSetAttribute((const void *)k_scatter2<W, RW, T2, BT2, true, RW == 3 || RW == 4>, hipFuncAttributeMaxDynamicSharedMemorySize, T2 * (RW * 4) + 8 * BFCG_MAXB); if (e != hipSuccess) return e; }
	if constexpr (RW == 3) { e = hipFuncSetAttribute((const void *)k_scatter2<W, RW, TILE2_BIG, BT2_BIG, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, TILE2_BIG * (RW * 4) + 8 * BFCG_MAXB); if (e != hipSuccess) return e; }
	e = hipFuncSetAttribute((const void *)k_bloom<W, RW, 1024, 2, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); if (e != hipSuccess) return e;
	e = hipFuncSetAttribute((const void *)k_bloom<W, RW, 512, 4, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); if (e != hipSuccess) return e;
	e = hipFuncSetAttribute((const void *)k_bloom<W, RW, 512, 4, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); if (e != hipSuccess) return e;
	e = hipFuncSetAttribute((const void *)k_bloom<W, RW, 512, 4, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); if (e != hipSuccess) return e;
	e = hipFuncSetAttribute((const void *)k_bloom<W, RW, 512, 4, 4, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); if (e != hipSuccess) return e;
	e = hipFuncSetAttribute((const void *)k_bloom<W, RW, 512, 4, 4, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); if (e != hipSuccess) return e;
	e = hipFuncSetAttribute((const void *)k_bloom<W, RW, 512, 4, 4, false, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); if (e != hipSuccess) return e;
	if constexpr (RW == 3) { e = hipFuncSetAttribute((const void *)k_bloom<W, RW, 512, 4, 4, false, false, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); if (e != hipSuccess) return e; }
	if constexpr (RW == 3) { e = set_bloom3_lds_attr(lds > 53008 ? lds : 53008); if (e != hipSuccess) return e; }
	if constexpr (RW == 4) { e = set_bloom3fm_lds_attr(lds > 53008 ? lds : 53008); if (e != hipSuccess) return e; } // (its COLD mode lays the same third of a CU's LDS out differently)
	e = hipFuncSetAttribute((const void *)k_bloom<W, RW, 512, 4, 0, false, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); if (e != hipSuccess) return e;
	e = hipFuncSetAttribute((const void *)k_commit_seg<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 << BFCG_SEG_MAX_SHIFT); if (e != hipSuccess) return e;
	e = hipFuncSetAttribute((const void *)k_commit_seg<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 << BFCG_SEG_MAX_SHIFT); if (e != hipSuccess) return e;
	e = hipFuncSetAttribute((const void *)k_commit_seg<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 << BFCG_SEG_MAX_SHIFT); if (e != hipSuccess) return e;
	e = hipFuncSetAttribute((const void *)k_bloom<W, RW, 1024, 2, 4, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); if (e != hipSuccess) return e;
	e = hipFuncSetAttribute((const void *)k_bloom<W, RW, 512, 4, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); if (e != hipSuccess) return e;
	e = hipFuncSetAttribute((const void *)k_bloom<W, RW, 512, 4, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); if (e != hipSuccess) return e;
	return hipSuccess;
}
hipError_t set_bloom_lds_attr(const KParams &P)
{
	int lds = bloom_lds_bytes(P);
	const int rw = bfcg_rec_dwords(P.k, P.rec_n);
	if (P.k <= 32) return rw == 3 ? set_attr_t<uint32_t, 3>(lds) : set_attr_t<uint32_t, 4>(lds);
	if (rw == 3) return set_attr_t<uint64_t, 3>(lds);
	if (rw == 4) return set_attr_t<uint64_t, 4>(lds);
	return set_attr_t<uint64_t, 5>(lds);
}

void run_query(const KParams &P, const uint8_t *seq, int64_t n_pos, const void *bloom, uint8_t *flags, hipStream_t st)
{
	const int64_t tiles = (n_pos + TILE1 - 1) / TILE1;
	const unsigned g = (unsigned)(((tiles + 7) / 8) * 8);
	const char *eq = getenv("BFCG_QUERY4"); // (1: k_query4 whatever the filter's size -- tests; 0: never -- A/B; read per call: tests switch it)
	const int q4 = eq ? atoi(eq) : -1;
	if (P.n_hashes == 4 && P.bf_shift <= 41 && (q4 == 1 || (q4 != 0 && P.bf_shift >= 35))) {
		if (P.k <= 32) hipLaunchKernelGGL((k_query4<uint32_t, TILE1, BT1>), dim3(g), dim3(BT1), 0, st, P, seq, n_pos, (const unsigned int *)bloom, reinterpret_cast<unsigned long long *>(flags));
		else hipLaunchKernelGGL((k_query4<uint64_t, TILE1, BT1>), dim3(g), dim3(BT1), 0, st, P, seq, n_pos, (const unsigned int *)bloom, reinterpret_cast<unsigned long long *>(flags));
		return;
	}
	const bool coop = P.bf_shift >= 35; // 4 GiB and more: measured 13.8 -> 21.2 G queries/s on 16 GiB, but 23.1 -> 21.9 on 1 GiB
	if (P.k <= 32) {
		if (coop) hipLaunchKernelGGL((k_query<uint32_t, TILE1, BT1, true>), dim3(g), dim3(BT1), 0, st, P, seq, n_pos, (const unsigned int *)bloom, reinterpret_cast<unsigned long long *>(flags));
		else hipLaunchKernelGGL((k_query<uint32_t, TILE1, BT1, false>), dim3(g), dim3(BT1), 0, st, P, seq, n_pos, (const unsigned int *)bloom, reinterpret_cast<unsigned long long *>(flags));
	} else {
		if (coop) hipLaunchKernelGGL((k_query<uint64_t, TILE1, BT1, true>), dim3(g), dim3(BT1), 0, st, P, seq, n_pos, (const unsigned int *)bloom, reinterpret_cast<unsigned long long *>(flags));
		else hipLaunchKernelGGL((k_query<uint64_t, TILE1, BT1, false>), dim3(g), dim3(BT1), 0, st, P, seq, n_pos, (const unsigned int *)bloom, reinterpret_cast<unsigned long long *>(flags));
	}
}
void run_streak(int k, float min_frac, const uint8_t *flags, const uint64_t *off, uint64_t n_reads, int32_t *out_start, int32_t *out_end, hipStream_t st)
{
	if (n_reads == 0) return;
	hipLaunchKernelGGL(k_streak, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, st, k, min_frac, reinterpret_cast<const unsigned long long *>(flags), off, n_reads, out_start, out_end);
}

void run_kcov(const KParams &P, const uint8_t *seq, int64_t n_pos, int min_occ, const void *tab, uint8_t *flags, uint16_t *out, hipStream_t st)
{
	const int64_t tiles = (n_pos + TILE1 - 1) / TILE1;
	const unsigned g = (unsigned)(((tiles + 7) / 8) * 8);
	if (P.k <= 32) hipLaunchKernelGGL((k_occ<uint32_t, TILE1, BT1>), dim3(g), dim3(BT1), 0, st, P, seq, n_pos, min_occ, (const unsigned long long *)tab, flags);
	else hipLaunchKernelGGL((k_occ<uint64_t, TILE1, BT1>), dim3(g), dim3(BT1), 0, st, P, seq, n_pos, min_occ, (const unsigned long long *)tab, flags);
	hipLaunchKernelGGL((k_cov<TILE1, BT1>), dim3(g), dim3(BT1), 0, st, P.k, flags, n_pos, out);
}

void run_hash_only(const KParams &P, const uint8_t *seq, const uint8_t *qual, int64_t n_pos, uint64_t *out, hipStream_t st)
{
	const int64_t tiles = (n_pos + TILE1 - 1) / TILE1;
	if (P.k <= 32) hipLaunchKernelGGL((k_hash_only<uint32_t, TILE1, BT1>), dim3(grid_for(tiles, 4096)), dim3(BT1), 0, st, P, seq, qual, n_pos, out);
	else hipLaunchKernelGGL((k_hash_only<uint64_t, TILE1, BT1>), dim3(grid_for(tiles, 4096)), dim3(BT1), 0, st, P, seq, qual, n_pos, out);
}

void run_table_replay(const KParams &P, unsigned long long *tab, const uint64_t *src, uint64_t n, unsigned long long *stats, uint64_t *ovf, uint32_t ovf_cap,
                      unsigned long long *first, unsigned long long *sub_last, hipStream_t st)
{
	int g = (int)((n + 255) / 256); if (g > 4096) g = 4096; if (g < 1) g = 1;
	TabOrder O; O.first = first; O.sub_last = sub_last;
	hipLaunchKernelGGL(k_table_replay, dim3(g), dim3(256), 0, st, P, tab, src, n, stats, ovf, ovf_cap, stats + (size_t)ST_SLOTS * ST_N, O);
}
hipError_t set_seg_lds_attr(void) { return hipFuncSetAttribute((const void *)k_seg_rehash<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 << BFCG_SEG_MAX_SHIFT); }
void run_seg_rehash(const KParams &P, const unsigned long long *old_tab, int old_shift, int old_blk, unsigned long long *new_tab, uint32_t n_fine, hipStream_t st)
{
	hipLaunchKernelGGL((k_seg_rehash<512>), seg_grid(P, n_fine, 512u), dim3(512), (size_t)8 << P.seg_blk, st, P, old_tab, old_shift, old_blk, new_tab);
}
void run_seg_replay(const KParams &P, unsigned long long *seg_tab, const uint64_t *src, uint64_t n, unsigned long long *stats, uint64_t *ovf, uint32_t ovf_cap, hipStream_t st)
{
	int g = (int)((n + 255) / 256); if (g > 4096) g = 4096; if (g < 1) g = 1;
	if (P.k <= 32) hipLaunchKernelGGL((k_seg_replay<uint32_t>), dim3(g), dim3(256), 0, st, P, seg_tab, src, n, stats, ovf, ovf_cap, stats + (size_t)ST_SLOTS * ST_N);
	else hipLaunchKernelGGL((k_seg_replay<uint64_t>), dim3(g), dim3(256), 0, st, P, seg_tab, src, n, stats, ovf, ovf_cap, stats + (size_t)ST_SLOTS * ST_N);
}
void run_seg_to_table(const KParams &P, const unsigned long long *seg_tab, uint32_t n_fine, unsigned long long *tab, unsigned long long *stats, uint64_t *ovf, uint32_t ovf_cap, hipStream_t st)
{
	hipLaunchKernelGGL(k_seg_to_table, dim3(4096), dim3(256), 0, st, P, seg_tab, n_fine, tab, stats, ovf, ovf_cap, stats + (size_t)ST_SLOTS * ST_N);
}
void run_table_rehash(const KParams &P, const unsigned long long *old_tab, int cshift_old, unsigned long long *new_tab,
                      const unsigned long long *old_first, unsigned long long *new_first, hipStream_t st)
{
	hipLaunchKernelGGL(k_table_rehash, dim3(4096), dim3(256), 0, st, P, old_tab, cshift_old, new_tab, old_first, new_first);
}

} // namespace bfcg
