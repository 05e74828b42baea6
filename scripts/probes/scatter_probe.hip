// DESIGN.md section 7 item 2b, as a stand-alone experiment: what does the level-1 partition's SKELETON cost when 12-byte records
// leave through write-combining buffers in LDS (whole 128-byte lines, one reservation per CAP records) instead of the tile skeleton
// k_scatter1 has now (LDS ranks -> scan -> staging -> runs of ~6 records per (tile, bucket))?
//
// No k-mer arithmetic here: a record is a hash of its index (profiles/round3_k_scatter1.md: the hashing hides completely behind the
// skeleton), its bucket a bit field of it -- the probe prices ranks, barriers, LDS traffic, reservations and stores, nothing else.
//
//   tile  <NB, BT, S>       the present skeleton: tile of BT*S records, rank by a returning LDS add, scan, one cursor atomic per
//                           (tile, bucket), records staged in bucket order, copied out run by run
//   wc    <NB, CAP, BT, R>  persistent workgroups; a buffer of CAP records per bucket in LDS; a record takes the slot a returning LDS
//                           add hands out; a full buffer leaves as CAP*12 contiguous bytes to a chunk its owner thread reserved a
//                           round AHEAD (the memory-side atomic's latency is off the path); what found its buffer full waits for
//                           the flush and takes slot - CAP; the last partial buffers leave padded with dead records
//
// Every run is checked: per bucket, the live records of its slab must be the bucket's own (count and a 64-bit sum over the records).
// Build: hipcc --offload-arch=gfx950 -O3 -o build/scatter_probe scripts/probes/scatter_probe.hip ; run: build/scatter_probe [n_records]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define DEAD 0xffffffffu

// X extra rounds of a 64-bit mix (~9 VALU each) stand in for K1's ~114 VALU per k-mer: does the arithmetic hide behind the skeleton?
template <int X> __device__ __forceinline__ void gen(uint64_t i, uint32_t &w0, uint32_t &w1, uint32_t &w2)
{
	uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
#pragma unroll
	for (int x = 0; x < X; ++x) { z ^= z >> 31; z *= 0x94D049BB133111EBull; }
	if (X) z ^= z >> 32;
	w0 = (uint32_t)z; w1 = (uint32_t)(z >> 32); w2 = (uint32_t)i; // i < 2^32 - 1: never DEAD
}
template <int NB> __device__ __forceinline__ int bucket_of(uint32_t w1) { return (int)((w1 >> 9) & (NB - 1)); }
__device__ __forceinline__ uint64_t rec_sum(uint32_t w0, uint32_t w1, uint32_t w2) { return (uint64_t)w0 * 0x9E3779B1u + w1 + ((uint64_t)w2 << 20); }

// what every bucket must hold
template <int NB, int X> __global__ __launch_bounds__(256) void k_expect(uint64_t n, unsigned long long *cnt, unsigned long long *sum)
{
	__shared__ unsigned long long c[NB], s[NB];
	for (int b = threadIdx.x; b < NB; b += 256) c[b] = s[b] = 0;
	__syncthreads();
	for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
		uint32_t w0, w1, w2; gen<X>(i, w0, w1, w2);
		const int b = bucket_of<NB>(w1);
		atomicAdd(&c[b], 1ull); atomicAdd(&s[b], (unsigned long long)rec_sum(w0, w1, w2));
	}
	__syncthreads();
	for (int b = threadIdx.x; b < NB; b += 256) if (c[b]) { atomicAdd(&cnt[b], c[b]); atomicAdd(&sum[b], s[b]); }
}
// what every slab holds: blockIdx.y = bucket
template <int NB> __global__ __launch_bounds__(256) void k_check(const uint32_t *out, uint64_t slab, const uint32_t *cursor, uint32_t cs, unsigned long long *cnt, unsigned long long *sum, uint32_t *bad)
{
	const int b = blockIdx.y;
	const uint64_t m = cursor[(size_t)b * cs];
	unsigned long long c = 0, s = 0;
	for (uint64_t j = blockIdx.x * 256ull + threadIdx.x; j < m; j += (uint64_t)gridDim.x * 256) {
		const uint32_t *r = out + ((uint64_t)b * slab + j) * 3;
		const uint32_t w0 = r[0], w1 = r[1], w2 = r[2];
		if (w2 == DEAD) continue;
		if (bucket_of<NB>(w1) != b) atomicAdd(bad, 1u);
		++c; s += rec_sum(w0, w1, w2);
	}
	if (c) { atomicAdd(&cnt[b], c); atomicAdd(&sum[b], s); }
}

// ---------------------------------------------------------------- the present skeleton
template <int NB, int BT, int S, int X>
__global__ __launch_bounds__(BT) void k_tile(uint64_t n, uint32_t *__restrict__ out, uint64_t slab, uint32_t *__restrict__ cursor, uint32_t cs)
{
	constexpr int TILE = BT * S;
	static_assert(NB <= BT, "one bucket per thread in the scan");
	extern __shared__ uint4 smem[];
	uint32_t *stage = reinterpret_cast<uint32_t *>(smem);                          // TILE * 3
	unsigned short *sbk = reinterpret_cast<unsigned short *>(stage + TILE * 3);    // TILE
	uint32_t *cnt = reinterpret_cast<uint32_t *>(sbk + TILE), *offs = cnt + NB, *gbase = offs + NB; // 3 x NB
	__shared__ uint32_t wtot[BT / 64];
	const int tid = threadIdx.x;
	const uint64_t n_tiles = (n + TILE - 1) / TILE;
	for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
		if (tid < NB) cnt[tid] = 0;
		__syncthreads();
		uint32_t w0[S], w1[S], w2[S], rk[S]; int bk[S];
#pragma unroll
		for (int s = 0; s < S; ++s) {
			const uint64_t i = t * TILE + (uint64_t)s * BT + tid;
			gen<X>(i, w0[s], w1[s], w2[s]); bk[s] = i < n ? bucket_of<NB>(w1[s]) : -1;
			rk[s] = bk[s] >= 0 ? atomicAdd(&cnt[bk[s]], 1u) : 0;
		}
		__syncthreads();
		uint32_t v = tid < NB ? cnt[tid] : 0, inc = v;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(inc, d, 64); if ((tid & 63) >= d) inc += u; }
		if ((tid & 63) == 63) wtot[tid >> 6] = inc;
		const uint32_t g = (tid < NB && v) ? atomicAdd(&cursor[(size_t)tid * cs], v) : 0; // one cursor atomic per (tile, bucket)
		__syncthreads();
		uint32_t pre = 0;
		for (int w = 0; w < (tid >> 6); ++w) pre += wtot[w];
		if (tid < NB) { offs[tid] = pre + inc - v; gbase[tid] = g; }
		__syncthreads();
#pragma unroll
		for (int s = 0; s < S; ++s) if (bk[s] >= 0) {
			const uint32_t p = offs[bk[s]] + rk[s];
			stage[p * 3] = w0[s]; stage[p * 3 + 1] = w1[s]; stage[p * 3 + 2] = w2[s]; sbk[p] = (unsigned short)bk[s];
		}
		__syncthreads();
		const uint32_t live = (uint32_t)((t + 1) * TILE <= n ? TILE : n - t * TILE);
#pragma unroll
		for (int s = 0; s < S; ++s) {
			const uint32_t p = (uint32_t)s * BT + tid;
			if (p < live) {
				const int b = sbk[p];
				uint32_t *d = out + ((uint64_t)b * slab + gbase[b] + (p - offs[b])) * 3;
				d[0] = stage[p * 3]; d[1] = stage[p * 3 + 1]; d[2] = stage[p * 3 + 2];
			}
		}
		__syncthreads();
	}
}

// ---------------------------------------------------------------- write-combining buffers
// ABL (measurement switches; 1 and 2 give wrong slabs and are not checked): 1 = no reservation atomics (positions from a private counter),
// 2 = no copy-out stores, 4 = the flush list by ballot + one LDS atomic per wave instead of one same-address LDS atomic per flush
template <int NB, int CAP, int BT, int R, int X, int ABL = 0>
__global__ __launch_bounds__(BT) void k_wc(uint64_t n, uint32_t *__restrict__ out, uint64_t slab, uint32_t *__restrict__ cursor, uint32_t cs)
{
	static_assert(NB <= BT, "a bucket's owner is the thread of its number");
	static_assert(CAP % 4 == 0, "a buffer is a whole number of 16-byte pieces");
	constexpr int PIECES = CAP * 12 / 16;
	extern __shared__ uint4 smem[];
	uint32_t *buf = reinterpret_cast<uint32_t *>(smem);  // NB buffers of CAP records
	uint32_t *fill = buf + NB * CAP * 3;                 // slots handed out per bucket (>= CAP: the buffer is due)
	uint32_t *jobs = fill + NB, *jpos = jobs + NB;       // this round's flushes: bucket, chunk position in its slab
	__shared__ uint32_t njobs;
	const int tid = threadIdx.x;
	if (tid < NB) fill[tid] = 0;
	if (tid == 0) njobs = 0;
	uint32_t nextpos = tid < NB ? atomicAdd(&cursor[(size_t)tid * cs], (uint32_t)CAP) : 0; // always one chunk ahead
	__syncthreads();
	uint32_t w0[R], w1[R], w2[R], sl[R]; int bk[R];
	// x * 3 by shift and add, opaque to the compiler: its v_mad_u64_u32 (x * 12 + base) took the reservation atomic's result register as the
	// undefined high half of the addend, and with it an s_waitcnt vmcnt(0) -- the previous round's stores -- in front of every put
	auto put = [&](int r) { uint32_t o = (uint32_t)bk[r] * CAP + sl[r]; o += o << 1; asm volatile("" : "+v"(o)); buf[o] = w0[r]; buf[o + 1] = w1[r]; buf[o + 2] = w2[r]; };
	auto flush = [&]() {
		if (tid < NB) { // whole waves: NB is a multiple of 64
			const uint32_t f = fill[tid];
			const bool due = f >= CAP;
			uint32_t j = 0;
			if (ABL & 4) {
				const unsigned long long m = __ballot(due);
				uint32_t base = 0;
				if ((tid & 63) == 0 && m) base = atomicAdd(&njobs, (uint32_t)__popcll(m));
				base = __shfl(base, 0, 64);
				j = base + __popcll(m & ((1ull << (tid & 63)) - 1));
			}
			if (due) {
				if (!(ABL & 4)) j = atomicAdd(&njobs, 1u);
				jobs[j] = tid; jpos[j] = nextpos; fill[tid] = f - CAP;
				if (ABL & 1) nextpos = (nextpos + (uint32_t)CAP * 997u) % (uint32_t)(slab - CAP) / CAP * CAP;
				else nextpos = atomicAdd(&cursor[(size_t)tid * cs], (uint32_t)CAP);
			}
		}
		__syncthreads();
		const uint32_t np = (ABL & 2) ? 0 : njobs * PIECES;
		for (uint32_t x = tid; x < np; x += BT) {
			const uint32_t j = x / PIECES, p = x - j * PIECES, b = jobs[j];
			reinterpret_cast<uint4 *>(out)[((uint64_t)b * slab + jpos[j]) / 4 * 3 + p] = reinterpret_cast<const uint4 *>(buf)[b * PIECES + p];
		}
		__syncthreads();
		if (tid == 0) njobs = 0;
		bool still = false;
#pragma unroll
		for (int r = 0; r < R; ++r) if (bk[r] >= 0 && sl[r] >= CAP) { sl[r] -= CAP; if (sl[r] < CAP) { put(r); bk[r] = -1; } else still = true; }
		return still;
	};
	const uint64_t per = (uint64_t)BT * R, n_tiles = (n + per - 1) / per;
	for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
		bool far = false;
#pragma unroll
		for (int r = 0; r < R; ++r) {
			const uint64_t i = t * per + (uint64_t)r * BT + tid;
			gen<X>(i, w0[r], w1[r], w2[r]); bk[r] = i < n ? bucket_of<NB>(w1[r]) : -1;
			sl[r] = bk[r] >= 0 ? atomicAdd(&fill[bk[r]], 1u) : 0;
		}
#pragma unroll
		for (int r = 0; r < R; ++r) if (bk[r] >= 0) { if (sl[r] < CAP) { put(r); bk[r] = -1; } else far |= sl[r] >= 2 * CAP; }
		const int any_far = __syncthreads_or(far);
		bool still = flush();
		if (any_far) while (__syncthreads_or(still)) still = flush(); // a bucket drew more than two buffers' worth in one round
	}
	__syncthreads();
	for (uint32_t x = tid; x < NB * CAP; x += BT) if (x % CAP >= fill[x / CAP]) buf[x * 3 + 2] = DEAD;
	if (tid < NB) { jobs[tid] = tid; jpos[tid] = nextpos; }
	__syncthreads();
	for (uint32_t x = tid; x < NB * PIECES; x += BT) {
		const uint32_t j = x / PIECES, p = x - j * PIECES, b = jobs[j];
		reinterpret_cast<uint4 *>(out)[((uint64_t)b * slab + jpos[j]) / 4 * 3 + p] = reinterpret_cast<const uint4 *>(buf)[b * PIECES + p];
	}
}

// ---------------------------------------------------------------- write-combining buffers, roles by wave
// k_wc's ISA (first GPU run): every put() carried an s_waitcnt vmcnt(0) -- a false dependence on the register the reservation atomic
// returns to -- so every round waited for the previous round's STORES to be acknowledged (~5 us per round whatever its size).  Here
// wave 0 is the buckets' OWNER (fill tests, flush list, reservations: the only wave with returning memory operations, G chunks per
// atomic, the next group requested one flush before the current one runs out) and waves 1..NW are WORKERS (slots, puts, copy-out:
// stores only, nothing they ever wait for).
template <int NB, int CAP, int NW, int R, int G, int X>
__global__ __launch_bounds__(64 * (NW + 1)) void k_wc2(uint64_t n, uint32_t *__restrict__ out, uint64_t slab, uint32_t *__restrict__ cursor, uint32_t cs)
{
	static_assert(NB % 64 == 0 && CAP % 4 == 0, "buckets per owner lane, 16-byte pieces");
	constexpr int BT = 64 * (NW + 1), WT = 64 * NW, Q = NB / 64, PIECES = CAP * 12 / 16;
	extern __shared__ uint4 smem[];
	uint32_t *buf = reinterpret_cast<uint32_t *>(smem);
	uint32_t *fill = buf + NB * CAP * 3, *jobs = fill + NB, *jpos = jobs + NB;
	__shared__ uint32_t njobs;
	const int tid = threadIdx.x, wt = tid - 64; // wt < 0: the owner wave
	const bool owner = tid < 64;
	for (int b = tid; b < NB; b += BT) fill[b] = 0;
	if (tid == 0) njobs = 0;
	uint32_t pos[Q], left[Q], npos[Q];
	if (owner) {
#pragma unroll
		for (int q = 0; q < Q; ++q) { pos[q] = atomicAdd(&cursor[(size_t)(tid + 64 * q) * cs], (uint32_t)(G * CAP)); left[q] = G; npos[q] = 0; }
	}
	__syncthreads();
	uint32_t w0[R], w1[R], w2[R], sl[R]; int bk[R];
#pragma unroll
	for (int r = 0; r < R; ++r) bk[r] = -1;
	auto put = [&](int r) { const uint32_t o = ((uint32_t)bk[r] * CAP + sl[r]) * 3u; buf[o] = w0[r]; buf[o + 1] = w1[r]; buf[o + 2] = w2[r]; };
	auto flush = [&]() {
		if (owner) {
#pragma unroll
			for (int q = 0; q < Q; ++q) {
				const int b = tid + 64 * q;
				const uint32_t f = fill[b];
				if (f >= CAP) {
					const uint32_t j = atomicAdd(&njobs, 1u);
					jobs[j] = b; jpos[j] = pos[q]; fill[b] = f - CAP;
					if (G == 1) pos[q] = atomicAdd(&cursor[(size_t)b * cs], (uint32_t)CAP);
					else {
						pos[q] += CAP; --left[q];
						if (left[q] == 1) npos[q] = atomicAdd(&cursor[(size_t)b * cs], (uint32_t)(G * CAP));
						else if (left[q] == 0) { pos[q] = npos[q]; left[q] = G; }
					}
				}
			}
		}
		__syncthreads();
		if (!owner) {
			const uint32_t np = njobs * PIECES;
			for (uint32_t x = wt; x < np; x += WT) {
				const uint32_t j = x / PIECES, p = x - j * PIECES, b = jobs[j];
				reinterpret_cast<uint4 *>(out)[((uint64_t)b * slab + jpos[j]) / 4 * 3 + p] = reinterpret_cast<const uint4 *>(buf)[b * PIECES + p];
			}
		}
		__syncthreads();
		if (tid == 0) njobs = 0;
		bool still = false;
#pragma unroll
		for (int r = 0; r < R; ++r) if (bk[r] >= 0 && sl[r] >= CAP) { sl[r] -= CAP; if (sl[r] < CAP) { put(r); bk[r] = -1; } else still = true; }
		return still;
	};
	const uint64_t per = (uint64_t)WT * R, n_tiles = (n + per - 1) / per;
	for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
		bool far = false;
		if (!owner) {
#pragma unroll
			for (int r = 0; r < R; ++r) {
				const uint64_t i = t * per + (uint64_t)r * WT + wt;
				gen<X>(i, w0[r], w1[r], w2[r]); bk[r] = i < n ? bucket_of<NB>(w1[r]) : -1;
				sl[r] = bk[r] >= 0 ? atomicAdd(&fill[bk[r]], 1u) : 0;
			}
#pragma unroll
			for (int r = 0; r < R; ++r) if (bk[r] >= 0) { if (sl[r] < CAP) { put(r); bk[r] = -1; } else far |= sl[r] >= 2 * CAP; }
		}
		const int any_far = __syncthreads_or(far);
		bool still = flush();
		if (any_far) while (__syncthreads_or(still)) still = flush();
	}
	__syncthreads();
	for (uint32_t x = tid; x < NB * CAP; x += BT) if (x % CAP >= fill[x / CAP]) buf[x * 3 + 2] = DEAD;
	if (owner) {
#pragma unroll
		for (int q = 0; q < Q; ++q) {
			const int b = tid + 64 * q;
			jobs[b] = b; jpos[b] = pos[q];
			uint32_t *d = out + ((uint64_t)b * slab) * 3 + 2; // what was reserved and never filled: dead records
			for (uint32_t c = pos[q] + CAP; c < pos[q] + left[q] * CAP; ++c) d[(uint64_t)c * 3] = DEAD;
			if (G > 1 && left[q] == 1) for (uint32_t c = npos[q]; c < npos[q] + G * CAP; ++c) d[(uint64_t)c * 3] = DEAD;
		}
	}
	__syncthreads();
	for (uint32_t x = tid; x < NB * PIECES; x += BT) {
		const uint32_t j = x / PIECES, p = x - j * PIECES, b = jobs[j];
		reinterpret_cast<uint4 *>(out)[((uint64_t)b * slab + jpos[j]) / 4 * 3 + p] = reinterpret_cast<const uint4 *>(buf)[b * PIECES + p];
	}
}

static int g_cs = 1; // the cursors' stride in 32-bit words: 1 = dense (32 cursors share a 128-byte line), 32 = a line each, ...
struct Bufs { uint32_t *out, *cursor, *bad; unsigned long long *exp_c, *exp_s, *got_c, *got_s; uint64_t out_bytes; };

template <int NB, int X> static int verify(const Bufs &B, uint64_t n, uint64_t slab, double *dead_frac)
{
	CK(hipMemset(B.exp_c, 0, 8 * NB)); CK(hipMemset(B.exp_s, 0, 8 * NB)); CK(hipMemset(B.got_c, 0, 8 * NB)); CK(hipMemset(B.got_s, 0, 8 * NB)); CK(hipMemset(B.bad, 0, 4));
	hipLaunchKernelGGL((k_expect<NB, X>), dim3(4096), dim3(256), 0, 0, n, B.exp_c, B.exp_s);
	hipLaunchKernelGGL((k_check<NB>), dim3(64, NB), dim3(256), 0, 0, B.out, slab, B.cursor, (uint32_t)g_cs, B.got_c, B.got_s, B.bad);
	CK(hipDeviceSynchronize());
	static unsigned long long ec[1024], es[1024], gc[1024], gs[1024]; static uint32_t cur[1024]; uint32_t bad;
	CK(hipMemcpy2D(cur, 4, B.cursor, (size_t)4 * g_cs, 4, NB, hipMemcpyDeviceToHost));
	CK(hipMemcpy(ec, B.exp_c, 8 * NB, hipMemcpyDeviceToHost)); CK(hipMemcpy(es, B.exp_s, 8 * NB, hipMemcpyDeviceToHost));
	CK(hipMemcpy(gc, B.got_c, 8 * NB, hipMemcpyDeviceToHost)); CK(hipMemcpy(gs, B.got_s, 8 * NB, hipMemcpyDeviceToHost));
	CK(hipMemcpy(&bad, B.bad, 4, hipMemcpyDeviceToHost));
	int wrong = bad != 0; uint64_t tot = 0, used = 0;
	for (int b = 0; b < NB; ++b) { wrong += ec[b] != gc[b] || es[b] != gs[b]; wrong += cur[b] > slab; tot += gc[b]; used += cur[b]; }
	wrong += tot != n;
	*dead_frac = used ? (double)(used - tot) / used : 0;
	return wrong;
}

template <typename L> static double timed(const Bufs &B, int NB, L launch)
{
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	float best = 1e30f;
	for (int rep = 0; rep < 4; ++rep) { // the first is the warm-up
		CK(hipMemsetAsync(B.cursor, 0, (size_t)4 * NB * g_cs, 0));
		CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
		float ms; CK(hipEventElapsedTime(&ms, e0, e1));
		if (rep && ms < best) best = ms;
	}
	return best;
}

static int n_cu;
static void report(const char *what, int wgs_per_cu, size_t lds, double ms, uint64_t n, int wrong, double dead)
{
	printf("%-44s %d wg/CU %6.1f KiB LDS : %7.3f ms = %5.2f ps/record = %5.0f GB/s of records written, %4.2f %% dead  %s\n", what, wgs_per_cu, lds / 1024.0, ms, ms * 1e9 / n,
	       n * 12.0 / ms / 1e6, dead * 100, wrong ? "*** WRONG ***" : "ok");
	fflush(stdout);
}

template <int NB, int BT, int S, int X = 0> static void run_tile(const Bufs &B, uint64_t n)
{
	const size_t lds = (size_t)BT * S * 14 + 12 * NB;
	auto k = k_tile<NB, BT, S, X>;
	CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	int per = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k, BT, lds));
	const uint64_t slab = ((uint64_t)(n / NB * 1.05) + 65536) / 64 * 64;
	if (slab * NB * 12 > B.out_bytes) { printf("tile: slabs do not fit\n"); return; }
	const double ms = timed(B, NB, [&] { hipLaunchKernelGGL(k, dim3(n_cu * per), dim3(BT), lds, 0, n, B.out, slab, B.cursor, (uint32_t)g_cs); });
	double dead; const int wrong = verify<NB, X>(B, n, slab, &dead);
	char what[96]; snprintf(what, sizeof what, "tile  NB=%d tile=%d (%d thr x %d)%s", NB, BT * S, BT, S, X ? " +ALU" : "");
	report(what, per, lds, ms, n, wrong, dead);
}
template <int NB, int CAP, int BT, int R, int X = 0, int ABL = 0> static void run_wc(const Bufs &B, uint64_t n, int max_per = 8)
{
	const size_t lds = (size_t)NB * CAP * 12 + 12 * NB;
	auto k = k_wc<NB, CAP, BT, R, X, ABL>;
	CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	int per = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k, BT, lds));
	if (per > max_per) per = max_per;
	const uint64_t slab = ((uint64_t)(n / NB * 1.05) + 65536 + (uint64_t)2 * CAP * n_cu * per) / 64 * 64;
	if (slab * NB * 12 > B.out_bytes) { printf("wc: slabs do not fit\n"); return; }
	const double ms = timed(B, NB, [&] { hipLaunchKernelGGL(k, dim3(n_cu * per), dim3(BT), lds, 0, n, B.out, slab, B.cursor, (uint32_t)g_cs); });
	double dead = 0; const int wrong = (ABL & 3) ? 0 : verify<NB, X>(B, n, slab, &dead);
	char what[96]; snprintf(what, sizeof what, "wc    NB=%d CAP=%d (%d B) %d thr x %d%s%s%s%s", NB, CAP, CAP * 12, BT, R, X ? " +ALU" : "", (ABL & 4) ? " ballot" : "", (ABL & 1) ? " NO-ATOMICS" : "", (ABL & 2) ? " NO-STORES" : "");
	report(what, per, lds, ms, n, wrong, dead);
	if (ABL & 3) printf("    (switched off: not a partition, not checked)\n");
}

template <int NB, int CAP, int NW, int R, int G, int X = 0> static void run_wc2(const Bufs &B, uint64_t n, int max_per = 8)
{
	const size_t lds = (size_t)NB * CAP * 12 + 12 * NB;
	constexpr int BT = 64 * (NW + 1);
	auto k = k_wc2<NB, CAP, NW, R, G, X>;
	CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	int per = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k, BT, lds));
	if (per > max_per) per = max_per;
	const uint64_t slab = ((uint64_t)(n / NB * 1.05) + 65536 + (uint64_t)(2 * G + 1) * CAP * n_cu * per) / 64 * 64;
	if (slab * NB * 12 > B.out_bytes) { printf("wc2: slabs do not fit\n"); return; }
	const double ms = timed(B, NB, [&] { hipLaunchKernelGGL(k, dim3(n_cu * per), dim3(BT), lds, 0, n, B.out, slab, B.cursor, (uint32_t)g_cs); });
	double dead; const int wrong = verify<NB, X>(B, n, slab, &dead);
	char what[96]; snprintf(what, sizeof what, "wc2   NB=%d CAP=%d 1+%d waves x %d, G=%d%s", NB, CAP, NW, R, G, X ? " +ALU" : "");
	report(what, per, lds, ms, n, wrong, dead);
}

int main(int argc, char **argv)
{
	const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 0) : 554000000ull; // one c3 batch: 3.67 M reads = 554 M positions
	if (n >= 0xffffffffull) { fprintf(stderr, "n < 2^32 - 1\n"); return 1; }
	hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); n_cu = pr.multiProcessorCount;
	printf("%s, %d CUs; %llu records of 12 bytes (%.2f GB), buckets = a bit field of the record\n", pr.name, n_cu, (unsigned long long)n, n * 12e-9);
	Bufs B; B.out_bytes = (uint64_t)(n * 12 * 1.08) + (1ull << 30);
	CK(hipMalloc(&B.out, B.out_bytes)); CK(hipMalloc(&B.cursor, (size_t)4 * 1024 * 1024)); CK(hipMalloc(&B.bad, 4));
	CK(hipMalloc(&B.exp_c, 8192)); CK(hipMalloc(&B.exp_s, 8192)); CK(hipMalloc(&B.got_c, 8192)); CK(hipMalloc(&B.got_s, 8192));
	CK(hipMemset(B.out, 0, B.out_bytes));
	const bool all = argc > 2;        // any second argument: the longer list of shapes
	const int strides[3] = {1, 32, 1024};
	for (int si = 0; si < 3; ++si) {
		g_cs = strides[si];
		printf("---- the buckets' cursors %d bytes apart%s\n", 4 * g_cs, g_cs == 1 ? " (32 share a line)" : "");
		run_tile<512, 512, 8>(B, n);      // k_scatter1's shape on c3
		run_tile<256, 512, 8>(B, n);
		run_wc<256, 32, 1024, 4>(B, n);   // whole 384-byte chunks (three lines), one workgroup of 16 waves per CU
		run_wc<512, 16, 1024, 4>(B, n);   // 192-byte chunks: every other one starts in mid-line
		run_wc<256, 32, 1024, 8>(B, n);
		run_wc<256, 16, 1024, 2>(B, n);   // 51 KiB: two workgroups of 16 waves per CU; a round = a quarter of the buffers
		run_wc<512, 8, 1024, 2>(B, n);
		run_wc<256, 32, 1024, 4, 0, 1>(B, n);   // where the time goes: switches
		run_wc<256, 32, 1024, 4, 0, 2>(B, n);
		run_wc<256, 32, 1024, 4, 0, 3>(B, n);
		printf("with 12 extra rounds of a 64-bit mix per record (~110 VALU, K1's weight):\n");
		run_tile<512, 512, 8, 12>(B, n);
		run_wc<256, 32, 1024, 4, 12>(B, n);
		run_wc<512, 16, 1024, 4, 12>(B, n);
		run_wc<256, 32, 1024, 8, 12>(B, n);
		run_wc<256, 16, 1024, 2, 12>(B, n);
	}
	g_cs = 32;
	if (all) {
		printf("---- more shapes, cursors 128 bytes apart\n");
		run_tile<512, 1024, 4>(B, n);
		run_wc<256, 32, 1024, 4, 0, 4>(B, n); run_wc<512, 16, 1024, 4, 0, 4>(B, n);
		run_wc<256, 16, 1024, 4>(B, n); run_wc<512, 8, 1024, 4>(B, n); run_wc<256, 16, 512, 4>(B, n); run_wc<256, 16, 512, 8>(B, n); run_wc<512, 16, 1024, 8>(B, n);
		run_wc<256, 32, 1024, 2>(B, n); run_wc<256, 32, 512, 4>(B, n);
		run_wc<512, 16, 512, 4>(B, n); run_wc<128, 32, 512, 4>(B, n); run_wc<128, 64, 512, 4>(B, n); run_wc<128, 32, 256, 4>(B, n);
		run_wc<256, 8, 512, 4>(B, n);
		run_wc2<256, 32, 15, 4, 1>(B, n); run_wc2<256, 32, 15, 4, 4>(B, n); run_wc2<256, 32, 15, 8, 4>(B, n); run_wc2<256, 16, 7, 8, 4>(B, n);
		run_wc2<512, 16, 15, 4, 4>(B, n); run_wc2<512, 8, 7, 4, 8>(B, n);
		run_wc2<256, 32, 15, 4, 4, 12>(B, n); run_wc2<512, 16, 15, 4, 4, 12>(B, n);
		run_wc<256, 16, 512, 4, 12>(B, n); run_wc<256, 16, 1024, 4, 12>(B, n); run_wc<128, 32, 512, 4, 12>(B, n); run_wc<128, 32, 256, 4, 12>(B, n); run_tile<256, 512, 8, 12>(B, n);
	}
	return 0;
}
