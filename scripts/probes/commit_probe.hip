// Microbenchmark behind profiles/round4_c4_c5.md: what does a pass of k_commit_seg over config c4's 64 KiB table segments cost, and what would a
// persistent, double-buffered workgroup save?  Synthetic segments (2^S slots, half full) and hand-over pages (E entries per region and page,
// a tenth of them new keys), NF regions -- far more than the caches hold.
//   V0   the shipped structure: one workgroup of 1024 threads per region, segment -> LDS, pages applied with a barrier each, LDS -> segment
//   V0s  V0 without the upserts (what streaming the segments through LDS costs in that structure)
//   V0u  V0 without the segment's load and store (what the upserts cost)
//   V1   persistent workgroups (one per CU, two 64 KiB buffers): the NEXT region's segment is requested into registers before the pages of the
//        current one are applied, and the current one's stores are left in flight behind it
// Build: hipcc --offload-arch=gfx950 -O3 -o build/commit_probe scripts/probes/commit_probe.hip ; run: build/commit_probe [log2 regions] [seg_shift] [entries] [pages]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__host__ __device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
__device__ __forceinline__ uint32_t seg_home(uint64_t id) { return (uint32_t)((id * 0x9E3779B97F4A7C15ULL) >> 38); }
__device__ __forceinline__ uint64_t key_of(uint32_t f, uint32_t r) { return (mix(((uint64_t)f << 20) | r) & ((1ULL << 46) - 1)) | 1ULL; }

struct Args { unsigned long long *seg; const unsigned long long *log; const uint32_t *mark; uint32_t log_stride, mark_stride, pages, n_fine; int seg_shift; unsigned long long *stats; };

// entries of the log: build = every key of the region's pool once; test = `pages` pages of E draws from [0, pool * 11 / 10)
__global__ void k_make_log(unsigned long long *log, uint32_t *mark, uint32_t log_stride, uint32_t mark_stride, uint32_t n_fine, uint32_t pool, uint32_t E, uint32_t pages, int build, uint64_t seed)
{
	const uint32_t f = blockIdx.x;
	const uint32_t n = build ? pool : E * pages;
	for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) {
		const uint32_t r = build ? j : (uint32_t)(mix(seed + ((uint64_t)f << 24) + j) % (pool + pool / 10));
		log[(uint64_t)f * log_stride + j] = (key_of(f, r) << 1) | (mix(r + seed) & 1);
	}
	if (threadIdx.x < pages) mark[(size_t)threadIdx.x * mark_stride + f] = build ? pool : E * (threadIdx.x + 1);
}

__device__ __forceinline__ int seg_upsert(unsigned long long *seg, uint32_t mask, uint64_t id, uint32_t c, uint32_t h)
{
	const unsigned long long fresh = (id << 14) | c | ((uint64_t)h << 8);
	uint32_t p = seg_home(id) & mask;
	for (uint32_t probe = 0; probe <= mask; ++probe, p = (p + 1) & mask) {
		unsigned long long cur = __hip_atomic_load(&seg[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		if (cur == 0) { cur = atomicCAS(&seg[p], 0ULL, fresh); if (cur == 0) return 1; }
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

// V0 and its ablations (MODE 0: all, 1: no upserts, 2: no stream, 3: all, six entries per thread requested before the segment)
template <int BT, int MODE>
__global__ __launch_bounds__(BT) void k_v0(Args A)
{
	extern __shared__ __attribute__((aligned(16))) unsigned long long lseg[];
	__shared__ uint32_t s_new[8], s_mark[8];
	const uint32_t f = blockIdx.x, pages = A.pages;
	if (MODE >= 4 && blockIdx.x < 512u && (blockIdx.x & 1u)) { // the first workgroups to become resident: every second one waits (MODE - 3) x 3 us
		const long long t0 = wall_clock64();
		while (wall_clock64() - t0 < (long long)(MODE - 3) * 300) __builtin_amdgcn_s_sleep(8); // (100 MHz counter)
	}
	const uint32_t n = A.mark[(size_t)(pages - 1) * A.mark_stride + f];
	const unsigned long long *recs = A.log + (uint64_t)f * A.log_stride;
	const uint32_t slots = 1u << A.seg_shift, mask = slots - 1;
	unsigned long long *gseg = A.seg + ((uint64_t)f << A.seg_shift);
	if (threadIdx.x < 8) { s_new[threadIdx.x] = 0; s_mark[threadIdx.x] = threadIdx.x < pages ? A.mark[(size_t)threadIdx.x * A.mark_stride + f] : n; }
	uint32_t j = threadIdx.x;
	const unsigned long long pre0 = j < n ? recs[j] : 0ULL, pre1 = j + BT < n ? recs[j + BT] : 0ULL;
	unsigned long long pre2 = 0, pre3 = 0, pre4 = 0, pre5 = 0;
	if (MODE == 3) { pre2 = j + 2 * BT < n ? recs[j + 2 * BT] : 0ULL; pre3 = j + 3 * BT < n ? recs[j + 3 * BT] : 0ULL; pre4 = j + 4 * BT < n ? recs[j + 4 * BT] : 0ULL; pre5 = j + 5 * BT < n ? recs[j + 5 * BT] : 0ULL; }
	if (MODE != 2) {
		const uint4 *src = reinterpret_cast<const uint4 *>(gseg);
		uint4 *dst = reinterpret_cast<uint4 *>(lseg);
		for (uint32_t i = threadIdx.x; i < slots / 2; i += BT) dst[i] = src[i];
	}
	__syncthreads();
	uint32_t k = 0;
	for (uint32_t pg = 0; pg < pages; ++pg) {
		const uint32_t end = s_mark[pg];
		uint32_t n_new = 0;
		if (MODE != 1) for (; j < end; j += BT, ++k) {
			const unsigned long long v = k == 0 ? pre0 : k == 1 ? pre1 : MODE == 3 && k == 2 ? pre2 : MODE == 3 && k == 3 ? pre3 : MODE == 3 && k == 4 ? pre4 : MODE == 3 && k == 5 ? pre5 : recs[j];
			const int r = seg_upsert(lseg, mask, v >> 1, 1u, (uint32_t)(v & 1));
			if (r > 0) ++n_new;
		}
		for (int o = 32; o; o >>= 1) n_new += __shfl_down(n_new, o);
		if ((threadIdx.x & 63) == 0 && n_new) atomicAdd(&s_new[pg], n_new);
		__syncthreads();
	}
	if (MODE != 2) {
		uint4 *dst = reinterpret_cast<uint4 *>(gseg);
		const uint4 *src = reinterpret_cast<const uint4 *>(lseg);
		for (uint32_t i = threadIdx.x; i < slots / 2; i += BT) dst[i] = src[i];
	}
	if (threadIdx.x < pages && s_new[threadIdx.x]) atomicAdd(&A.stats[(size_t)(f & 255) * 8 + threadIdx.x], (unsigned long long)s_new[threadIdx.x]);
}

// V1: persistent workgroups, two LDS buffers; PER = 16-byte pieces of a segment per thread
template <int BT, int PER>
__global__ __launch_bounds__(BT) void k_v1(Args A)
{
	extern __shared__ __attribute__((aligned(16))) unsigned long long lbuf[];
	__shared__ uint32_t s_new[2][8], s_mark[2][8];
	const uint32_t pages = A.pages, slots = 1u << A.seg_shift, mask = slots - 1;
	uint32_t f = blockIdx.x;
	if (f >= A.n_fine) return;
	uint4 rg[PER];
	auto request = [&](uint32_t ff) {
		const uint4 *src = reinterpret_cast<const uint4 *>(A.seg + ((uint64_t)ff << A.seg_shift));
#pragma unroll
		for (int u = 0; u < PER; ++u) rg[u] = src[threadIdx.x + u * BT];
	};
	auto land = [&](int b) {
		uint4 *dst = reinterpret_cast<uint4 *>(lbuf + (size_t)b * slots);
#pragma unroll
		for (int u = 0; u < PER; ++u) dst[threadIdx.x + u * BT] = rg[u];
	};
	int cur = 0;
	request(f);
	unsigned long long pre0, pre1, pre2;
	uint32_t n;
	auto entries = [&](uint32_t ff, int b) {
		const unsigned long long *recs = A.log + (uint64_t)ff * A.log_stride;
		n = A.mark[(size_t)(pages - 1) * A.mark_stride + ff];
		if (threadIdx.x < 8) { s_new[b][threadIdx.x] = 0; s_mark[b][threadIdx.x] = threadIdx.x < pages ? A.mark[(size_t)threadIdx.x * A.mark_stride + ff] : n; }
		pre0 = threadIdx.x < n ? recs[threadIdx.x] : 0ULL; pre1 = threadIdx.x + BT < n ? recs[threadIdx.x + BT] : 0ULL; pre2 = threadIdx.x + 2 * BT < n ? recs[threadIdx.x + 2 * BT] : 0ULL;
	};
	entries(f, 0);
	land(0);
	for (;;) {
		const uint32_t fn = f + gridDim.x;
		const bool more = fn < A.n_fine;
		const unsigned long long e0 = pre0, e1 = pre1, e2 = pre2;
		const uint32_t n_cur = n;
		if (more) { request(fn); entries(fn, cur ^ 1); } // in flight under the pages of the current region
		__syncthreads();       // buffer `cur` has landed (and the previous round's reads of the other buffer are done)
		unsigned long long *lseg = lbuf + (size_t)cur * slots;
		const unsigned long long *recs = A.log + (uint64_t)f * A.log_stride;
		uint32_t j = threadIdx.x, k = 0;
		for (uint32_t pg = 0; pg < pages; ++pg) {
			const uint32_t end = s_mark[cur][pg];
			uint32_t n_new = 0;
			for (; j < end; j += BT, ++k) {
				const unsigned long long v = k == 0 ? e0 : k == 1 ? e1 : k == 2 ? e2 : recs[j];
				if (seg_upsert(lseg, mask, v >> 1, 1u, (uint32_t)(v & 1)) > 0) ++n_new;
			}
			for (int o = 32; o; o >>= 1) n_new += __shfl_down(n_new, o);
			if ((threadIdx.x & 63) == 0 && n_new) atomicAdd(&s_new[cur][pg], n_new);
			__syncthreads();
		}
		(void)n_cur;
		if (more) land(cur ^ 1); // (waits for the requested pieces; the other buffer was stored a round ago)
		{
			uint4 *dst = reinterpret_cast<uint4 *>(A.seg + ((uint64_t)f << A.seg_shift));
			const uint4 *src = reinterpret_cast<const uint4 *>(lseg);
#pragma unroll
			for (int u = 0; u < PER; ++u) dst[threadIdx.x + u * BT] = src[threadIdx.x + u * BT];
		}
		if (threadIdx.x < pages && s_new[cur][threadIdx.x]) atomicAdd(&A.stats[(size_t)(f & 255) * 8 + threadIdx.x], (unsigned long long)s_new[cur][threadIdx.x]);
		if (!more) break;
		f = fn; cur ^= 1;
	}
}


// V2: persistent workgroups of 16 waves, two 64 KiB buffers, roles by wave: the lower half moves segments (store the previous region, load the
// next one: global -> registers -> LDS at once, nothing held across a phase), the upper half applies the current region's pages.  One barrier
// per page and round.
template <int BT>
__global__ __launch_bounds__(BT) void k_v2(Args A)
{
	extern __shared__ __attribute__((aligned(16))) unsigned long long lbuf[];
	__shared__ uint32_t s_new[2][8], s_mark[2][8];
	constexpr int HT = BT / 2;
	const uint32_t pages = A.pages, slots = 1u << A.seg_shift, mask = slots - 1;
	const bool mover = threadIdx.x < HT;
	const uint32_t t = mover ? threadIdx.x : threadIdx.x - HT;
	uint32_t f = blockIdx.x;
	if (f >= A.n_fine) return;
	auto load_seg = [&](uint32_t ff, int b) {
		const uint4 *src = reinterpret_cast<const uint4 *>(A.seg + ((uint64_t)ff << A.seg_shift));
		uint4 *dst = reinterpret_cast<uint4 *>(lbuf + (size_t)b * slots);
		for (uint32_t i = t; i < slots / 2; i += HT) dst[i] = src[i];
	};
	auto store_seg = [&](uint32_t ff, int b) {
		uint4 *dst = reinterpret_cast<uint4 *>(A.seg + ((uint64_t)ff << A.seg_shift));
		const uint4 *src = reinterpret_cast<const uint4 *>(lbuf + (size_t)b * slots);
		for (uint32_t i = t; i < slots / 2; i += HT) dst[i] = src[i];
	};
	auto marks = [&](uint32_t ff, int b) { // (by the movers' first lanes)
		if (t < 8) { const uint32_t n = A.mark[(size_t)(pages - 1) * A.mark_stride + ff]; s_new[b][t] = 0; s_mark[b][t] = t < pages ? A.mark[(size_t)t * A.mark_stride + ff] : n; }
	};
	if (mover) { load_seg(f, 0); marks(f, 0); }
	int cur = 0;
	uint32_t prev = 0xffffffffu;
	for (;;) {
		__syncthreads(); // buffer cur holds region f; buffer cur ^ 1 holds region prev, its pages applied
		const uint32_t fn = f + gridDim.x;
		const bool more = fn < A.n_fine;
		if (mover) {
			if (prev != 0xffffffffu) {
				store_seg(prev, cur ^ 1);
				if (t < pages && s_new[cur ^ 1][t]) atomicAdd(&A.stats[(size_t)(prev & 255) * 8 + t], (unsigned long long)s_new[cur ^ 1][t]);
			}
		}
		if (mover && more) { __builtin_amdgcn_s_waitcnt(0); /* (the LDS reads of the store are done before the buffer is overwritten) */ }
		if (!mover) {
			unsigned long long *lseg = lbuf + (size_t)cur * slots;
			const unsigned long long *recs = A.log + (uint64_t)f * A.log_stride;
			uint32_t j = t;
			for (uint32_t pg = 0; pg < pages; ++pg) {
				const uint32_t end = s_mark[cur][pg];
				uint32_t n_new = 0;
				for (; j < end; j += HT) { const unsigned long long v = recs[j]; if (seg_upsert(lseg, mask, v >> 1, 1u, (uint32_t)(v & 1)) > 0) ++n_new; }
				for (int o = 32; o; o >>= 1) n_new += __shfl_down(n_new, o);
				if ((t & 63) == 0 && n_new) atomicAdd(&s_new[cur][pg], n_new);
				// (pages of one region are applied by the upper half alone: its waves meet at the round's barrier only -- per-page key counts
				// would need a barrier among them; the probe counts per round)
			}
		} else if (more) { load_seg(fn, cur ^ 1); marks(fn, cur ^ 1); }
		prev = f;
		if (!more) break;
		f = fn; cur ^= 1;
	}
	__syncthreads();
	if (mover) {
		store_seg(prev, cur);
		if (t < pages && s_new[cur][t]) atomicAdd(&A.stats[(size_t)(prev & 255) * 8 + t], (unsigned long long)s_new[cur][t]);
	}
}

// ---- c3's shape: segments of 2^11 slots, a 32-bit counter pair per slot behind the segment in LDS (k_commit_seg's use_cnt path), 256 threads ----
// VAR 0: the shipped loop (one entry per lane at a time, entries and probes in one loop); VAR 1: two independent entry streams per lane (two LDS
// reads in flight); VAR 2: slots read in aligned PAIRS (one 16-byte LDS read covers the home slot and its neighbour when the home is even)
template <int BT, int VAR>
__global__ __launch_bounds__(BT) void k_c3(Args A)
{
	extern __shared__ __attribute__((aligned(16))) unsigned long long lseg[];
	__shared__ uint32_t s_new[8], s_mark[8];
	const uint32_t f = blockIdx.x, pages = A.pages;
	const uint32_t n = A.mark[(size_t)(pages - 1) * A.mark_stride + f];
	const unsigned long long *recs = A.log + (uint64_t)f * A.log_stride;
	const uint32_t slots = 1u << A.seg_shift, mask = slots - 1;
	unsigned long long *gseg = A.seg + ((uint64_t)f << A.seg_shift);
	unsigned int *lcnt = reinterpret_cast<unsigned int *>(lseg + slots);
	if (threadIdx.x < 8) { s_new[threadIdx.x] = 0; s_mark[threadIdx.x] = threadIdx.x < pages ? A.mark[(size_t)threadIdx.x * A.mark_stride + f] : n; }
	{
		const uint4 *src = reinterpret_cast<const uint4 *>(gseg);
		uint4 *dst = reinterpret_cast<uint4 *>(lseg);
		for (uint32_t i = threadIdx.x; i < slots / 2; i += BT) dst[i] = src[i];
		for (uint32_t i = threadIdx.x; i < slots; i += BT) lcnt[i] = 0;
	}
	__syncthreads();
	// VAR 4: a page goes through LDS in tiles of TE entries (staged by all threads, home slots computed there: every lane busy), and the lanes of a
	// wave DRAW entries from the wave's share of the tile as they finish (ballot + prefix count: no atomics) instead of owning every BT-th entry:
	// a wave runs (its probes / 64) steps, not its busiest lane's; the probe loop carries no loads, no hash, no entry rotation
	constexpr int TE = 1024;
	__shared__ unsigned long long t_v[VAR == 4 ? TE : 1];
	__shared__ unsigned short t_h[VAR == 4 ? TE : 1];
	if (VAR == 4) {
		const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
		constexpr int NW = BT / 64;
		uint32_t beg = 0;
		for (uint32_t pg = 0; pg < pages; ++pg) {
			const uint32_t end = s_mark[pg];
			uint32_t n_new = 0;
			for (uint32_t t0 = beg; t0 < end; t0 += TE) {
				const uint32_t nt = end - t0 < TE ? end - t0 : TE;
				for (uint32_t i = threadIdx.x; i < nt; i += BT) { const unsigned long long v = recs[t0 + i]; t_v[i] = v; t_h[i] = (unsigned short)(seg_home(v >> 1) & mask); }
				__syncthreads();
				const uint32_t per = (nt + NW - 1) / NW, lo = per * wave < nt ? per * wave : nt, hi = lo + per < nt ? lo + per : nt;
				uint32_t cursor = lo + 64; // (wave-uniform) the next entry nobody has drawn
				uint32_t idx = lo + lane;
				bool have = idx < hi;
				unsigned long long v = have ? t_v[idx] : 0ULL;
				uint32_t p = have ? t_h[idx] : 0u, probes = 0;
				while (__any(have)) {
					bool done = false;
					if (have) {
						const uint64_t id = v >> 1;
						const uint32_t hq = (uint32_t)(v & 1);
						unsigned long long cur = lseg[p];
						if (cur == 0) { cur = atomicCAS(&lseg[p], 0ULL, (id << 14) | 1ULL | ((unsigned long long)hq << 8)); if (cur == 0) { ++n_new; done = true; } }
						if (!done && (cur >> 14) == id) { __hip_atomic_fetch_add(&lcnt[p], 1u | (hq << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); done = true; }
						if (!done) { p = (p + 1) & mask; if (++probes > mask) done = true; }
					}
					const unsigned long long b = __ballot(done);
					if (b) {
						const uint32_t mine = cursor + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
						cursor += (uint32_t)__popcll(b);
						if (done) { have = mine < hi; if (have) { v = t_v[mine]; p = t_h[mine]; probes = 0; } }
					}
				}
				__syncthreads(); // (the tile's buffer is free again; at a page's last tile: the page is applied)
			}
			for (int o = 32; o; o >>= 1) n_new += __shfl_down(n_new, o);
			if ((threadIdx.x & 63) == 0 && n_new) atomicAdd(&s_new[pg], n_new);
			beg = end;
		}
		__syncthreads();
	}
	__shared__ unsigned long long s_ent[BT * 3];
	if (VAR == 3) { for (uint32_t i = threadIdx.x; i < BT * 3; i += BT) s_ent[i] = i < n ? recs[i] : 0ULL; __syncthreads(); }
	if (VAR == 5 || VAR == 6) { // pair reads; the next entry is loaded where the lane advances (5) or one entry ahead (6): no three-deep rotation
		uint32_t j = threadIdx.x;
		unsigned long long v0 = j < n ? recs[j] : 0ULL, v1 = VAR == 6 && j + BT < n ? recs[j + BT] : 0ULL;
		uint32_t p = seg_home(v0 >> 1) & mask, probes = 0;
		for (uint32_t pg = 0; pg < pages; ++pg) {
			const uint32_t end = s_mark[pg];
			uint32_t n_new = 0;
			while (__any(j < end)) {
				if (j < end) {
					const uint64_t id = v0 >> 1;
					const uint32_t hi = (uint32_t)(v0 & 1);
					const unsigned long long fresh = (id << 14) | 1ULL | ((unsigned long long)hi << 8);
					bool done = false;
					unsigned long long cur, nxt = 0;
					const bool pair = !(p & 1u);
					if (pair) { const ulonglong2 pr = *reinterpret_cast<const ulonglong2 *>(&lseg[p]); cur = pr.x; nxt = pr.y; } else cur = lseg[p];
					if (cur == 0) { cur = atomicCAS(&lseg[p], 0ULL, fresh); if (cur == 0) { ++n_new; done = true; } }
					if (!done && (cur >> 14) == id) { __hip_atomic_fetch_add(&lcnt[p], 1u | (hi << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); done = true; }
					if (!done && pair) {
						++probes;
						if (nxt == 0) { nxt = atomicCAS(&lseg[p + 1], 0ULL, fresh); if (nxt == 0) { ++n_new; done = true; } }
						if (!done && (nxt >> 14) == id) { __hip_atomic_fetch_add(&lcnt[p + 1], 1u | (hi << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); done = true; }
						if (!done) p += 1;
					}
					if (!done) { p = (p + 1) & mask; if (++probes > mask) done = true; }
					if (done) {
						j += BT;
						if (VAR == 6) { v0 = v1; v1 = j + BT < n ? recs[j + BT] : 0ULL; } else v0 = j < n ? recs[j] : 0ULL;
						p = seg_home(v0 >> 1) & mask; probes = 0;
					}
				}
			}
			for (int o = 32; o; o >>= 1) n_new += __shfl_down(n_new, o);
			if ((threadIdx.x & 63) == 0 && n_new) atomicAdd(&s_new[pg], n_new);
			__syncthreads();
		}
	}
	if (VAR == 0 || VAR == 2 || VAR == 3) {
		uint32_t j = threadIdx.x;
		unsigned long long v0 = j < n ? recs[j] : 0ULL, v1 = j + BT < n ? recs[j + BT] : 0ULL, v2 = j + 2 * BT < n ? recs[j + 2 * BT] : 0ULL;
		uint32_t p = seg_home(v0 >> 1) & mask, probes = 0;
		for (uint32_t pg = 0; pg < pages; ++pg) {
			const uint32_t end = s_mark[pg];
			uint32_t n_new = 0;
			while (__any(j < end)) {
				if (j < end) {
					const uint64_t id = v0 >> 1;
					const uint32_t hi = (uint32_t)(v0 & 1);
					bool done = false;
					if (VAR == 2 && !(p & 1u)) { // the aligned pair (p, p + 1) in one read
						const ulonglong2 pr = *reinterpret_cast<const ulonglong2 *>(&lseg[p]);
						unsigned long long cur = pr.x;
						if (cur == 0) { cur = atomicCAS(&lseg[p], 0ULL, (id << 14) | 1ULL | ((unsigned long long)hi << 8)); if (cur == 0) { ++n_new; done = true; } }
						if (!done && (cur >> 14) == id) { __hip_atomic_fetch_add(&lcnt[p], 1u | (hi << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); done = true; }
						if (!done) {
							cur = pr.y; // (a stale zero is re-examined by the CAS; a non-zero slot never changes its key)
							if (cur == 0) { cur = atomicCAS(&lseg[p + 1], 0ULL, (id << 14) | 1ULL | ((unsigned long long)hi << 8)); if (cur == 0) { ++n_new; done = true; } }
							if (!done && (cur >> 14) == id) { __hip_atomic_fetch_add(&lcnt[p + 1], 1u | (hi << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); done = true; }
							if (!done) { p = (p + 2) & mask; probes += 2; }
						}
					} else {
						unsigned long long cur = lseg[p];
						if (cur == 0) { cur = atomicCAS(&lseg[p], 0ULL, (id << 14) | 1ULL | ((unsigned long long)hi << 8)); if (cur == 0) { ++n_new; done = true; } }
						if (!done && (cur >> 14) == id) { __hip_atomic_fetch_add(&lcnt[p], 1u | (hi << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); done = true; }
						if (!done) { p = (p + 1) & mask; ++probes; }
					}
					if (!done && probes > mask) done = true;
					if (done) { j += BT; v0 = v1; v1 = v2; v2 = VAR == 3 ? s_ent[(j + 2 * BT) % (BT * 3)] : j + 2 * BT < n ? recs[j + 2 * BT] : 0ULL; p = seg_home(v0 >> 1) & mask; probes = 0; }
				}
			}
			for (int o = 32; o; o >>= 1) n_new += __shfl_down(n_new, o);
			if ((threadIdx.x & 63) == 0 && n_new) atomicAdd(&s_new[pg], n_new);
			__syncthreads();
		}
	} else if (VAR == 1) { // two streams per lane: entries j, j + 2 BT, ... and j + BT, j + 3 BT, ...
		uint32_t ja = threadIdx.x, jb = threadIdx.x + BT;
		unsigned long long a0 = ja < n ? recs[ja] : 0ULL, a1 = ja + 2 * BT < n ? recs[ja + 2 * BT] : 0ULL;
		unsigned long long b0 = jb < n ? recs[jb] : 0ULL, b1 = jb + 2 * BT < n ? recs[jb + 2 * BT] : 0ULL;
		uint32_t pa = seg_home(a0 >> 1) & mask, pb = seg_home(b0 >> 1) & mask, qa = 0, qb = 0;
		for (uint32_t pg = 0; pg < pages; ++pg) {
			const uint32_t end = s_mark[pg];
			uint32_t n_new = 0;
			while (__any(ja < end || jb < end)) {
				const bool ona = ja < end, onb = jb < end;
				unsigned long long ca = 0, cb = 0;
				if (ona) ca = lseg[pa];
				if (onb) cb = lseg[pb];
				if (ona) {
					const uint64_t id = a0 >> 1; const uint32_t hi = (uint32_t)(a0 & 1);
					bool done = false;
					if (ca == 0) { ca = atomicCAS(&lseg[pa], 0ULL, (id << 14) | 1ULL | ((unsigned long long)hi << 8)); if (ca == 0) { ++n_new; done = true; } }
					if (!done && (ca >> 14) == id) { __hip_atomic_fetch_add(&lcnt[pa], 1u | (hi << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); done = true; }
					if (!done) { pa = (pa + 1) & mask; if (++qa > mask) done = true; }
					if (done) { ja += 2 * BT; a0 = a1; a1 = ja + 2 * BT < n ? recs[ja + 2 * BT] : 0ULL; pa = seg_home(a0 >> 1) & mask; qa = 0; }
				}
				if (onb) {
					const uint64_t id = b0 >> 1; const uint32_t hi = (uint32_t)(b0 & 1);
					bool done = false;
					if (cb == 0) { cb = atomicCAS(&lseg[pb], 0ULL, (id << 14) | 1ULL | ((unsigned long long)hi << 8)); if (cb == 0) { ++n_new; done = true; } }
					if (!done && (cb >> 14) == id) { __hip_atomic_fetch_add(&lcnt[pb], 1u | (hi << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); done = true; }
					if (!done) { pb = (pb + 1) & mask; if (++qb > mask) done = true; }
					if (done) { jb += 2 * BT; b0 = b1; b1 = jb + 2 * BT < n ? recs[jb + 2 * BT] : 0ULL; pb = seg_home(b0 >> 1) & mask; qb = 0; }
				}
			}
			for (int o = 32; o; o >>= 1) n_new += __shfl_down(n_new, o);
			if ((threadIdx.x & 63) == 0 && n_new) atomicAdd(&s_new[pg], n_new);
			__syncthreads();
		}
	}
	for (uint32_t i = threadIdx.x; i < slots; i += BT) {
		const uint32_t c = lcnt[i];
		if (c) {
			const unsigned long long v = lseg[i];
			const uint32_t nc = (uint32_t)(v & 0xff) + (c & 0xffffu), nh = (uint32_t)((v >> 8) & 0x3f) + (c >> 16);
			lseg[i] = (v & ~0x3fffULL) | (nc < 255 ? nc : 255) | ((unsigned long long)(nh < 63 ? nh : 63) << 8);
		}
	}
	__syncthreads();
	{
		uint4 *dst = reinterpret_cast<uint4 *>(gseg);
		const uint4 *src = reinterpret_cast<const uint4 *>(lseg);
		for (uint32_t i = threadIdx.x; i < slots / 2; i += BT) dst[i] = src[i];
	}
	if (threadIdx.x < pages && s_new[threadIdx.x]) atomicAdd(&A.stats[(size_t)(f & 255) * 8 + threadIdx.x], (unsigned long long)s_new[threadIdx.x]);
}

__global__ void k_digest(const unsigned long long *seg, uint64_t n, unsigned long long *out)
{
	unsigned long long s = 0, c = 0;
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { const unsigned long long v = seg[i]; if (v) { s += mix(v); ++c; } }
	atomicAdd(&out[0], s); atomicAdd(&out[1], c);
}

int main(int argc, char **argv)
{
	const int lf = argc > 1 ? atoi(argv[1]) : 17, S = argc > 2 ? atoi(argv[2]) : 13;
	const uint32_t E = argc > 3 ? (uint32_t)atoi(argv[3]) : 1335, pages = argc > 4 ? (uint32_t)atoi(argv[4]) : 2;
	const uint32_t NF = 1u << lf, slots = 1u << S, pool = slots / 2;
	const uint32_t log_stride = pool > E * pages ? pool : E * pages;
	unsigned long long *seg, *log, *stats, *dig; uint32_t *mark;
	CK(hipMalloc(&seg, (size_t)NF * slots * 8)); CK(hipMalloc(&log, (size_t)NF * log_stride * 8)); CK(hipMalloc(&mark, (size_t)NF * 8 * 4));
	CK(hipMalloc(&stats, 256 * 8 * 8)); CK(hipMalloc(&dig, 16));
	hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
	const int n_cu = pr.multiProcessorCount;
	printf("commit_probe: %u regions x 2^%d slots (%.1f GiB), %u entries x %u pages per region, %d CUs\n", NF, S, (double)NF * slots * 8 / (1 << 30), E, pages, n_cu);
	Args A{seg, log, mark, log_stride, NF, pages, NF, S, stats};
	const size_t lds = (size_t)slots * 8;
	CK(hipFuncSetAttribute((const void *)k_v0<1024, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	CK(hipFuncSetAttribute((const void *)k_v0<1024, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	CK(hipFuncSetAttribute((const void *)k_v0<1024, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	CK(hipFuncSetAttribute((const void *)k_v0<1024, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	CK(hipFuncSetAttribute((const void *)k_v0<1024, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	CK(hipFuncSetAttribute((const void *)k_v0<1024, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	CK(hipFuncSetAttribute((const void *)k_v0<1024, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	CK(hipFuncSetAttribute((const void *)k_v2<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds)));
	CK(hipFuncSetAttribute((const void *)k_v0<512, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	CK(hipFuncSetAttribute((const void *)k_v1<1024, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds)));
	CK(hipFuncSetAttribute((const void *)k_v1<512, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds)));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	auto rebuild = [&]() {
		CK(hipMemset(seg, 0, (size_t)NF * slots * 8)); CK(hipMemset(stats, 0, 256 * 8 * 8));
		Args B = A; B.pages = 1;
		hipLaunchKernelGGL(k_make_log, dim3(NF), dim3(256), 0, 0, log, mark, log_stride, NF, NF, pool, E, 1u, 1, 1ULL);
		hipLaunchKernelGGL((k_v0<1024, 0>), dim3(NF), dim3(1024), lds, 0, B);
		hipLaunchKernelGGL(k_make_log, dim3(NF), dim3(256), 0, 0, log, mark, log_stride, NF, NF, pool, E, pages, 0, 77ULL);
		CK(hipDeviceSynchronize());
	};
	auto report = [&](const char *name, float ms) {
		CK(hipMemset(dig, 0, 16));
		hipLaunchKernelGGL(k_digest, dim3(4096), dim3(256), 0, 0, seg, (uint64_t)NF * slots, dig);
		unsigned long long h[2]; CK(hipMemcpy(h, dig, 16, hipMemcpyDeviceToHost));
		const double gb = 2.0 * NF * slots * 8 / 1e9, ups = (double)NF * E * pages;
		printf("%-34s %8.3f ms  segments %.2f TB/s  %.1f ps per upsert  (x 2^%d regions: %.1f ms)  keys %llu digest %016llx\n", name, ms, gb / ms, ms * 1e9 / ups, 20 - lf, ms * (1 << (20 - lf)), h[1], h[0]);
	};
#define RUN(name, launch) do { rebuild(); CK(hipEventRecord(e0, 0)); launch; CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float ms_; CK(hipEventElapsedTime(&ms_, e0, e1)); report(name, ms_); } while (0)
	if (S <= 12) { // config c3's shape: the counter-pair path
		const size_t l3 = (size_t)slots * 12;
		CK(hipFuncSetAttribute((const void *)k_c3<256, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3));
		CK(hipFuncSetAttribute((const void *)k_c3<256, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3));
		CK(hipFuncSetAttribute((const void *)k_c3<256, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3));
		CK(hipFuncSetAttribute((const void *)k_c3<256, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3));
		CK(hipFuncSetAttribute((const void *)k_c3<256, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3));
		CK(hipFuncSetAttribute((const void *)k_c3<256, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3));
		CK(hipFuncSetAttribute((const void *)k_c3<256, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3));
		CK(hipFuncSetAttribute((const void *)k_c3<512, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3));
		CK(hipFuncSetAttribute((const void *)k_c3<512, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3));
		CK(hipFuncSetAttribute((const void *)k_c3<512, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3));
		for (int rep = 0; rep < 2; ++rep) {
			RUN("c3 shipped loop, 256 thr", hipLaunchKernelGGL((k_c3<256, 0>), dim3(NF), dim3(256), l3, 0, A));
			RUN("c3 two streams per lane, 256 thr", hipLaunchKernelGGL((k_c3<256, 1>), dim3(NF), dim3(256), l3, 0, A));
			RUN("c3 slots read in pairs, 256 thr", hipLaunchKernelGGL((k_c3<256, 2>), dim3(NF), dim3(256), l3, 0, A));
			RUN("c3 pairs, entry loaded at the advance", hipLaunchKernelGGL((k_c3<256, 5>), dim3(NF), dim3(256), l3, 0, A));
			RUN("c3 pairs, one entry ahead", hipLaunchKernelGGL((k_c3<256, 6>), dim3(NF), dim3(256), l3, 0, A));
			RUN("c3 entries drawn from LDS tiles, 256", hipLaunchKernelGGL((k_c3<256, 4>), dim3(NF), dim3(256), l3, 0, A));
			RUN("c3 entries drawn from LDS tiles, 512", hipLaunchKernelGGL((k_c3<512, 4>), dim3(NF), dim3(512), l3, 0, A));
			RUN("c3 TIMING ONLY: no loads in the loop", hipLaunchKernelGGL((k_c3<256, 3>), dim3(NF), dim3(256), l3, 0, A));
			RUN("c3 shipped loop, 512 thr", hipLaunchKernelGGL((k_c3<512, 0>), dim3(NF), dim3(512), l3, 0, A));
			RUN("c3 two streams per lane, 512 thr", hipLaunchKernelGGL((k_c3<512, 1>), dim3(NF), dim3(512), l3, 0, A));
		}
		return 0;
	}
	for (int rep = 0; rep < 2; ++rep) {
		RUN("V0 shipped structure, 1024 thr", hipLaunchKernelGGL((k_v0<1024, 0>), dim3(NF), dim3(1024), lds, 0, A));
		RUN("V0p six entries ahead, 1024 thr", hipLaunchKernelGGL((k_v0<1024, 3>), dim3(NF), dim3(1024), lds, 0, A));
		RUN("V0 staggered by 3 us", hipLaunchKernelGGL((k_v0<1024, 4>), dim3(NF), dim3(1024), lds, 0, A));
		RUN("V0 staggered by 6 us", hipLaunchKernelGGL((k_v0<1024, 5>), dim3(NF), dim3(1024), lds, 0, A));
		RUN("V0 staggered by 9 us", hipLaunchKernelGGL((k_v0<1024, 6>), dim3(NF), dim3(1024), lds, 0, A));
		if (2 * lds <= 160 * 1024 - 2048) RUN("V2 loader waves + upsert waves", hipLaunchKernelGGL((k_v2<1024>), dim3(n_cu), dim3(1024), 2 * lds, 0, A));
		RUN("V0s stream only", hipLaunchKernelGGL((k_v0<1024, 1>), dim3(NF), dim3(1024), lds, 0, A));
		RUN("V0 512 thr", hipLaunchKernelGGL((k_v0<512, 0>), dim3(NF), dim3(512), lds, 0, A));
		if (2 * lds <= 160 * 1024 - 2048) {
			RUN("V1 persistent 2 buffers, 1024 thr", hipLaunchKernelGGL((k_v1<1024, 4>), dim3(n_cu), dim3(1024), 2 * lds, 0, A));
			RUN("V1 persistent 2 buffers, 512 thr", hipLaunchKernelGGL((k_v1<512, 8>), dim3(n_cu), dim3(512), 2 * lds, 0, A));
		}
	}
	return 0;
}
