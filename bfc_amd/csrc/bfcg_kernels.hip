// bfcg_kernels.hip -- hand-written gfx950 kernels for the k-mer counting path of bfc
// (count.c + bbf.c + htab.c).  See DESIGN.md for the pipeline; in short, per batch of reads:
//
//   k_hist1    bases -> k-mers (K1, kmer_dev.h) -> histogram of level-1 bucket ids
//   k_scatter  K1 again -> 16/24-byte k-mer records scattered into level-1 buckets
//   k_hist2    level-1 buckets -> histogram of fine bucket ids          (two-level only)
//   k_scatter2 level-1 buckets -> fine buckets                          (two-level only)
//   k_bloom    one workgroup per fine bucket = one contiguous REGION of 2^R bloom blocks,
//              staged in LDS; exact sequential `seen` flags by a first-setter table in LDS
//              (SURVEY App. C.1); seen k-mers upserted into the HBM-resident count table
//              by atomicCAS probing (or OR-ed into the second bloom filter in filter mode).
//
// A "fine bucket" f holds the k-mers whose bloom block id has f as its top bits, so everything
// that can interact under the reference's sequential semantics (bbf.c:27-31: one 64-byte block
// per k-mer) meets inside one workgroup.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kmer_dev.h"
#include "bfcg_internal.h"

using namespace bfcg;

#define WAVE 64

// ------------------------------------------------------------------------------------------
// K1: tile of positions -> bit planes in LDS

// planes: [0] low base bit, [1] high base bit, [2] not-ACGT, [3] quality >= q
// Covers positions [t0-64, t0+TILE); PLANE_WORDS = (TILE+64)/32 + 2 spare words per plane.
template <int TILE, int BT>
__device__ __forceinline__ void build_planes(const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual,
                                             int64_t n_pos, int64_t t0, int q, uint32_t *planes)
{
	constexpr int PW = (TILE + 64) / 32 + 2;
	const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
	constexpr int NCH = (TILE + 64) / 64;
	for (int c = wave; c < NCH; c += BT / WAVE) {
		int64_t pos = t0 - 64 + (int64_t)c * 64 + lane;
		bool in = pos >= 0 && pos < n_pos;
		uint32_t ch = in ? seq[pos] : (uint32_t)'\n';
		uint32_t u = ch & 0xDFu; // fold case
		// A=0 C=1 G=2 T=3 (bseq.c:9-26 minus one, count.c:82); anything else is a break
		uint32_t code = (u == 'A') ? 0u : (u == 'C') ? 1u : (u == 'G') ? 2u : (u == 'T') ? 3u : 4u;
		bool hq = qual ? (in && ((int)qual[pos] - 33 >= q)) : true; // count.c:85
		uint64_t b0 = __ballot(code & 1u), b1 = __ballot((code >> 1) & 1u), bn = __ballot(code >> 2), bq = __ballot(hq);
		if (lane == 0) {
			planes[0 * PW + 2 * c] = (uint32_t)b0; planes[0 * PW + 2 * c + 1] = (uint32_t)(b0 >> 32);
			planes[1 * PW + 2 * c] = (uint32_t)b1; planes[1 * PW + 2 * c + 1] = (uint32_t)(b1 >> 32);
			planes[2 * PW + 2 * c] = (uint32_t)bn; planes[2 * PW + 2 * c + 1] = (uint32_t)(bn >> 32);
			planes[3 * PW + 2 * c] = (uint32_t)bq; planes[3 * PW + 2 * c + 1] = (uint32_t)(bq >> 32);
		}
	}
	if (threadIdx.x < 8) planes[(threadIdx.x >> 1) * PW + PW - 2 + (threadIdx.x & 1)] = 0;
}

// k-mer ending at tile-relative position r (0 <= r < TILE).  Returns false if there is none.
template <typename W, int TILE>
__device__ __forceinline__ bool kmer_at(const uint32_t *planes, int r, int k, W m, W &y0, W &y1, bool &is_high)
{
	constexpr int PW = (TILE + 64) / 32 + 2;
	int bit = r + 65 - k;
	if (window<W>(planes + 2 * PW, bit, m) != 0) return false;
	W wl = window<W>(planes, bit, m), wh = window<W>(planes + PW, bit, m);
	is_high = window<W>(planes + 3 * PW, bit, m) == m;
	kmer_hash_from_windows<W>(k, wl, wh, m, y0, y1);
	return true;
}

// ------------------------------------------------------------------------------------------
// records

template <int RW> struct Rec;
template <> struct Rec<2> { // k <= 47: y0,y1 < 2^47; flag at bit 47 of w0; idx split over the top 16 bits
	static __device__ __forceinline__ void pack(uint64_t *dst, uint64_t y0, uint64_t y1, uint32_t idx, bool hi)
	{
		ulonglong2 v;
		v.x = y0 | ((uint64_t)hi << 47) | ((uint64_t)(idx & 0xffffu) << 48);
		v.y = y1 | ((uint64_t)(idx >> 16) << 48);
		*reinterpret_cast<ulonglong2 *>(dst) = v;
	}
	static __device__ __forceinline__ void unpack(const uint64_t *src, uint64_t &y0, uint64_t &y1, uint32_t &idx, bool &hi)
	{
		ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(src);
		y0 = v.x & ((1ULL << 47) - 1); hi = (v.x >> 47) & 1;
		y1 = v.y & ((1ULL << 48) - 1);
		idx = (uint32_t)(v.x >> 48) | ((uint32_t)(v.y >> 48) << 16);
	}
};
template <> struct Rec<3> { // k <= 63
	static __device__ __forceinline__ void pack(uint64_t *dst, uint64_t y0, uint64_t y1, uint32_t idx, bool hi)
	{ dst[0] = y0 | ((uint64_t)hi << 63); dst[1] = y1; dst[2] = idx; }
	static __device__ __forceinline__ void unpack(const uint64_t *src, uint64_t &y0, uint64_t &y1, uint32_t &idx, bool &hi)
	{ y0 = src[0] & ~(1ULL << 63); hi = src[0] >> 63; y1 = src[1]; idx = (uint32_t)src[2]; }
};

template <typename W> __device__ __forceinline__ uint32_t fine_id(const KParams &P, uint64_t y0, uint64_t y1)
{
	W m = kmask<W>(P.k);
	uint64_t hash = bloom_hash<W>(P.k, (W)y0, (W)y1, m);
	uint64_t blk = hash & ((1ULL << (P.bf_shift - 9)) - 1);
	return (uint32_t)(blk >> P.R);
}

// ------------------------------------------------------------------------------------------
// pass A: level-1 histogram straight from the bases.  Persistent workgroups, LDS counters,
// one global atomicAdd per (workgroup, non-empty bucket).

template <typename W, int TILE, int BT>
__global__ __launch_bounds__(BT) void k_hist1(KParams P, const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual,
                                              int64_t n_pos, uint32_t *__restrict__ cnt1, unsigned long long *__restrict__ stats)
{
	constexpr int PW = (TILE + 64) / 32 + 2;
	__shared__ uint32_t planes[4 * PW];
	__shared__ uint32_t hist[512];
	const int nb1 = 1 << P.F1;
	for (int i = threadIdx.x; i < nb1; i += BT) hist[i] = 0;
	const W m = kmask<W>(P.k);
	const int shift2 = P.F2;
	uint32_t n_k = 0, n_h = 0;
	const int64_t n_tiles = (n_pos + TILE - 1) / TILE;
	for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		__syncthreads();
		build_planes<TILE, BT>(seq, qual, n_pos, tile * TILE, P.q, planes);
		__syncthreads();
#pragma unroll 4
		for (int j = 0; j < TILE / BT; ++j) {
			int r = j * BT + threadIdx.x;
			W y0, y1; bool hi;
			if (kmer_at<W, TILE>(planes, r, P.k, m, y0, y1, hi)) {
				uint32_t f = fine_id<W>(P, y0, y1);
				atomicAdd(&hist[f >> shift2], 1u);
				++n_k; n_h += hi;
			}
		}
	}
	__syncthreads();
	for (int i = threadIdx.x; i < nb1; i += BT)
		if (hist[i]) atomicAdd(&cnt1[i], hist[i]);
	// statistics: k-mers, high-quality k-mers
	for (int o = 32; o; o >>= 1) { n_k += __shfl_down(n_k, o); n_h += __shfl_down(n_h, o); }
	if ((threadIdx.x & 63) == 0) { atomicAdd(&stats[ST_KMERS], (unsigned long long)n_k); atomicAdd(&stats[ST_HIGH], (unsigned long long)n_h); }
}

// exclusive prefix sum of n counts (n <= 2^18) by one workgroup; also zeroes the cursors
__global__ __launch_bounds__(1024) void k_scan(const uint32_t *__restrict__ cnt, uint32_t *__restrict__ start, uint32_t *__restrict__ cursor, int n)
{
	__shared__ uint32_t part[1024];
	const int per = (n + 1023) / 1024;
	uint32_t s = 0;
	for (int i = 0; i < per; ++i) { int j = threadIdx.x * per + i; if (j < n) s += cnt[j]; }
	part[threadIdx.x] = s;
	__syncthreads();
	for (int o = 1; o < 1024; o <<= 1) {
		uint32_t v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
		__syncthreads();
		part[threadIdx.x] += v;
		__syncthreads();
	}
	uint32_t run = part[threadIdx.x] - s;
	for (int i = 0; i < per; ++i) {
		int j = threadIdx.x * per + i;
		if (j < n) { start[j] = run; run += cnt[j]; if (cursor) cursor[j] = 0; }
	}
	if (threadIdx.x == 1023) start[n] = part[1023];
}

// ------------------------------------------------------------------------------------------
// pass B: K1 again, records scattered to level-1 buckets.
// Per tile: LDS counters give every record its rank inside (tile, bucket); one global atomicAdd
// per (tile, non-empty bucket) reserves a contiguous run in the bucket; records are then stored
// straight from registers (runs are contiguous, so the L2 merges the 16-byte stores).

template <typename W, int RW, int TILE, int BT>
__global__ __launch_bounds__(BT) void k_scatter1(KParams P, const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual,
                                                 int64_t n_pos, const uint32_t *__restrict__ start1, uint32_t *__restrict__ cursor1,
                                                 uint64_t *__restrict__ out)
{
	constexpr int PW = (TILE + 64) / 32 + 2;
	constexpr int S = TILE / BT;
	__shared__ uint32_t planes[4 * PW];
	__shared__ uint32_t cnt[512];
	const int nb1 = 1 << P.F1;
	const W m = kmask<W>(P.k);
	const int64_t n_tiles = (n_pos + TILE - 1) / TILE;
	for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		__syncthreads();
		for (int i = threadIdx.x; i < nb1; i += BT) cnt[i] = 0;
		build_planes<TILE, BT>(seq, qual, n_pos, tile * TILE, P.q, planes);
		__syncthreads();
		W ry0[S], ry1[S];
		uint32_t rbr[S]; // bucket<<20 | rank<<1 | is_high ; 0xffffffff = no k-mer
#pragma unroll
		for (int j = 0; j < S; ++j) {
			int r = j * BT + threadIdx.x;
			bool hi;
			rbr[j] = 0xffffffffu;
			if (kmer_at<W, TILE>(planes, r, P.k, m, ry0[j], ry1[j], hi)) {
				uint32_t b = fine_id<W>(P, ry0[j], ry1[j]) >> P.F2;
				uint32_t rank = atomicAdd(&cnt[b], 1u);
				rbr[j] = (b << 20) | (rank << 1) | (uint32_t)hi;
			}
		}
		__syncthreads();
		for (int i = threadIdx.x; i < nb1; i += BT) {
			uint32_t c = cnt[i];
			cnt[i] = c ? start1[i] + atomicAdd(&cursor1[i], c) : 0;
		}
		__syncthreads();
#pragma unroll
		for (int j = 0; j < S; ++j) {
			if (rbr[j] != 0xffffffffu) {
				uint32_t b = rbr[j] >> 20, rank = (rbr[j] >> 1) & 0x7ffffu;
				uint64_t dst = (uint64_t)cnt[b] + rank;
				uint32_t idx = (uint32_t)(tile * TILE + j * BT + threadIdx.x); // end position = file order
				Rec<RW>::pack(out + dst * RW, (uint64_t)ry0[j], (uint64_t)ry1[j], idx, rbr[j] & 1u);
			}
		}
	}
}

// ------------------------------------------------------------------------------------------
// level 2: histogram / scatter of one level-1 bucket (blockIdx.y) into its 2^F2 fine buckets

template <typename W, int RW, int BT>
__global__ __launch_bounds__(BT) void k_hist2(KParams P, const uint64_t *__restrict__ in, const uint32_t *__restrict__ start1,
                                              uint32_t *__restrict__ cnt2)
{
	__shared__ uint32_t hist[512];
	const int nb2 = 1 << P.F2, b1 = blockIdx.y;
	const uint32_t s = start1[b1], e = start1[b1 + 1];
	if (s == e) return;
	for (int i = threadIdx.x; i < nb2; i += BT) hist[i] = 0;
	__syncthreads();
	for (uint64_t i = (uint64_t)s + blockIdx.x * BT + threadIdx.x; i < e; i += (uint64_t)gridDim.x * BT) {
		uint64_t y0, y1; uint32_t idx; bool hi;
		Rec<RW>::unpack(in + i * RW, y0, y1, idx, hi);
		atomicAdd(&hist[fine_id<W>(P, y0, y1) & (nb2 - 1)], 1u);
	}
	__syncthreads();
	for (int i = threadIdx.x; i < nb2; i += BT)
		if (hist[i]) atomicAdd(&cnt2[((uint32_t)b1 << P.F2) + i], hist[i]);
}

template <typename W, int RW, int TILE, int BT>
__global__ __launch_bounds__(BT) void k_scatter2(KParams P, const uint64_t *__restrict__ in, const uint32_t *__restrict__ start1,
                                                 const uint32_t *__restrict__ start2, uint32_t *__restrict__ cursor2,
                                                 uint64_t *__restrict__ out)
{
	constexpr int S = TILE / BT;
	__shared__ uint32_t cnt[512];
	const int nb2 = 1 << P.F2, b1 = blockIdx.y;
	const uint32_t s = start1[b1], e = start1[b1 + 1];
	const uint32_t n_tiles = (e - s + TILE - 1) / TILE;
	for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		__syncthreads();
		for (int i = threadIdx.x; i < nb2; i += BT) cnt[i] = 0;
		__syncthreads();
		uint64_t w[S][RW];
		uint32_t rbr[S];
#pragma unroll
		for (int j = 0; j < S; ++j) {
			uint64_t i = (uint64_t)s + (uint64_t)tile * TILE + j * BT + threadIdx.x;
			rbr[j] = 0xffffffffu;
			if (i < e) {
				uint64_t y0, y1; uint32_t idx; bool hi;
#pragma unroll
				for (int t = 0; t < RW; ++t) w[j][t] = in[i * RW + t];
				Rec<RW>::unpack(w[j], y0, y1, idx, hi);
				uint32_t b = fine_id<W>(P, y0, y1) & (nb2 - 1);
				rbr[j] = (b << 20) | atomicAdd(&cnt[b], 1u);
			}
		}
		__syncthreads();
		for (int i = threadIdx.x; i < nb2; i += BT) {
			uint32_t c = cnt[i], f = ((uint32_t)b1 << P.F2) + i;
			cnt[i] = c ? start2[f] + atomicAdd(&cursor2[f], c) : 0;
		}
		__syncthreads();
#pragma unroll
		for (int j = 0; j < S; ++j) {
			if (rbr[j] != 0xffffffffu) {
				uint64_t dst = (uint64_t)cnt[rbr[j] >> 20] + (rbr[j] & 0xfffffu);
#pragma unroll
				for (int t = 0; t < RW; ++t) out[dst * RW + t] = w[j][t];
			}
		}
	}
}

// ------------------------------------------------------------------------------------------
// count table in HBM: 2^l_pre regions of 2^tab_cshift u64 slots, slot = key(50)<<14|high(6)<<8|count(8)
// exactly as htab.c:7-17 stores it; empty = 0.  Home slot = low bits of key>>14 (as khash does),
// linear probing confined to the region.  Saturating counters by CAS (htab.c:74-79).

__device__ __forceinline__ void table_upsert(const KParams &P, unsigned long long *__restrict__ tab, uint64_t y0, uint64_t y1, bool hi,
                                             unsigned long long *__restrict__ stats, uint64_t *__restrict__ ovf, uint32_t ovf_cap)
{
	uint64_t key;
	uint32_t sub = ch_subkey(P.k, P.l_pre, y0, y1, key);
	const uint32_t cmask = (1u << P.tab_cshift) - 1;
	unsigned long long *reg = tab + ((uint64_t)sub << P.tab_cshift);
	uint32_t pos = (uint32_t)(key >> 14) & cmask;
	const unsigned long long fresh = key | ((uint64_t)hi << 8);
	for (uint32_t probe = 0; probe <= cmask; ++probe, pos = (pos + 1) & cmask) {
		unsigned long long cur = __hip_atomic_load(&reg[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (cur == 0) {
			cur = atomicCAS(&reg[pos], 0ULL, fresh);
			if (cur == 0) { atomicAdd(&stats[ST_KEYS], 1ULL); return; }
		}
		if ((cur >> 14) == (key >> 14)) {
			for (;;) {
				unsigned long long nv = cur;
				if ((nv & 0xff) != 0xff) ++nv;
				if (hi && ((nv >> 8) & 0x3f) != 0x3f) nv += 1 << 8;
				if (nv == cur) return;
				unsigned long long old = atomicCAS(&reg[pos], cur, nv);
				if (old == cur) return;
				cur = old;
			}
		}
	}
	// region full: park the k-mer; the host grows the table and replays (counts commute)
	unsigned long long o = atomicAdd(&stats[ST_TAB_OVF], 1ULL);
	if (o < ovf_cap) { ovf[3 * o] = y0; ovf[3 * o + 1] = y1; ovf[3 * o + 2] = hi; }
}

__global__ void k_table_replay(KParams P, unsigned long long *tab, const uint64_t *__restrict__ src, uint64_t n,
                               unsigned long long *stats, uint64_t *ovf, uint32_t ovf_cap)
{
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
		table_upsert(P, tab, src[3 * i], src[3 * i + 1], src[3 * i + 2] != 0, stats, ovf, ovf_cap);
}

// grow: re-insert every occupied slot of the old table (cshift_old) into the new one (P.tab_cshift)
__global__ void k_table_rehash(KParams P, const unsigned long long *__restrict__ old_tab, int cshift_old, unsigned long long *new_tab)
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
	}
}

// ------------------------------------------------------------------------------------------
// bloom region kernel

#define FS_EMPTY 0xffffffffffffffffULL

// first-setter table: entry = bit offset inside the region (high 32) | k-mer index (low 32);
// atomicMin keeps, per bit, the earliest k-mer (file order) that finds the bit clear.
template <bool GLOBAL>
__device__ __forceinline__ bool fs_insert(unsigned long long *tab, uint32_t cap_mask, uint32_t bitoff, uint32_t idx, uint32_t *n_used)
{
	const unsigned long long e = ((unsigned long long)bitoff << 32) | idx;
	uint32_t p = ((bitoff * 0x9E3779B1u) >> 12 ^ bitoff) & cap_mask; // low bits of a multiplicative hash are weak: fold the high half in
	for (uint32_t probe = 0; probe <= cap_mask; ++probe, p = (p + 1) & cap_mask) {
		unsigned long long cur = GLOBAL ? __hip_atomic_load(&tab[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tab[p];
		if (cur == FS_EMPTY) {
			cur = atomicCAS(&tab[p], FS_EMPTY, e);
			if (cur == FS_EMPTY) { if (n_used) atomicAdd(n_used, 1u); return true; }
		}
		if ((uint32_t)(cur >> 32) == bitoff) { if (e < cur) atomicMin(&tab[p], e); return true; }
	}
	return false;
}
// returns the first setter's index, or 0xffffffff... if the bit has no entry (it was set before the batch)
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
	const uint64_t *recs;          // fine-bucketed records
	const uint32_t *start;         // fine bucket starts (n_fine+1)
	unsigned long long *bloom;     // first bloom filter (device)
	unsigned long long *bloom_hi;  // second bloom filter (filter mode) or NULL
	unsigned long long *table;     // count table or NULL
	unsigned long long *stats;
	uint64_t *tab_ovf; uint32_t tab_ovf_cap;
	unsigned long long *pool; unsigned long long pool_cap; // global first-setter pool (entries)
	uint8_t *seen_out;             // optional debug: seen flag (1/2) per batch position
};

template <typename W, int RW>
__device__ __forceinline__ void emit_seen(const KParams &P, const BloomArgs &A, uint64_t y0, uint64_t y1, bool hi, uint64_t hash)
{
	if (A.table) table_upsert(P, A.table, y0, y1, hi, A.stats, A.tab_ovf, A.tab_ovf_cap);
	else if (A.bloom_hi) { // count.c:67-68: second filter keeps k-mers seen at least twice (order independent)
		BloomAddr a = bloom_addr(hash, P.bf_shift);
		unsigned int *blk = reinterpret_cast<unsigned int *>(A.bloom_hi) + a.blk * 16;
		uint32_t z = a.h1;
		for (int j = 0; j < P.n_hashes; ++j) { uint32_t b = bloom_next(z, a.h2); atomicOr(&blk[b >> 5], 1u << (b & 31)); }
	}
}

// LDS layout (dynamic): region 2^R*64 B | fs table FS_CAP*8 B | list LIST_CAP*4 B
template <typename W, int RW, int BT>
__global__ __launch_bounds__(BT) void k_bloom(KParams P, BloomArgs A)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	__shared__ uint32_t s_list_n, s_fs_n, s_ovf, s_pool_off;
	const uint32_t f = blockIdx.x;
	const uint32_t rs = A.start[f], n = A.start[f + 1] - rs;
	if (n == 0) return;
	const int region_blocks = 1 << P.R;                 // P.R already clamped to bf_shift-9
	const uint32_t region_dw = region_blocks * 16;
	unsigned int *region = reinterpret_cast<unsigned int *>(smem);
	unsigned long long *fs = reinterpret_cast<unsigned long long *>(smem + (size_t)region_dw * 4);
	uint32_t *list = reinterpret_cast<uint32_t *>(smem + (size_t)region_dw * 4 + (size_t)P.fs_cap * 8);
	const uint32_t fs_mask = P.fs_cap - 1;
	const uint64_t *recs = A.recs + (uint64_t)rs * RW;
	unsigned int *g_region = reinterpret_cast<unsigned int *>(A.bloom) + (uint64_t)f * region_dw;
	const W m = kmask<W>(P.k);
	const uint32_t rmask = region_blocks - 1;
	const int nh = P.n_hashes;

	{ // stage the region (16-byte loads), clear the first-setter table
		const uint4 *src = reinterpret_cast<const uint4 *>(g_region);
		uint4 *dst = reinterpret_cast<uint4 *>(region);
		for (uint32_t i = threadIdx.x; i < region_dw / 4; i += BT) dst[i] = src[i];
		for (uint32_t i = threadIdx.x; i < P.fs_cap; i += BT) fs[i] = FS_EMPTY;
		if (threadIdx.x == 0) { s_list_n = 0; s_fs_n = 0; s_ovf = (n >= (1u << 20)) ? 1u : 0u; }
	}
	__syncthreads();

	uint32_t n_seen = 0;
	const uint32_t fs_limit = (P.fs_cap >> 1) + (P.fs_cap >> 2); // keep probing short; racing inserts overshoot by < BT
	volatile uint32_t *v_ovf = &s_ovf, *v_fs_n = &s_fs_n;
	// ---- pass 1: classify every k-mer against the pre-batch region
	for (uint32_t i = threadIdx.x; i < n; i += BT) {
		uint64_t y0, y1; uint32_t idx; bool hi;
		Rec<RW>::unpack(recs + (uint64_t)i * RW, y0, y1, idx, hi);
		uint64_t hash = bloom_hash<W>(P.k, (W)y0, (W)y1, m);
		BloomAddr a = bloom_addr(hash, P.bf_shift);
		const uint32_t bl = (uint32_t)a.blk & rmask;
		uint32_t z = a.h1, um = 0;
		for (int j = 0; j < nh; ++j) {
			uint32_t b = bloom_next(z, a.h2);
			if (!((region[bl * 16 + (b >> 5)] >> (b & 31)) & 1u)) um |= 1u << j;
		}
		if (um == 0) { // every bit was set before this batch: seen, whatever the order inside the batch
			++n_seen;
			if (A.seen_out) A.seen_out[idx] = 2;
			emit_seen<W, RW>(P, A, y0, y1, hi, hash);
		} else if (!*v_ovf) {
			uint32_t li = atomicAdd(&s_list_n, 1u);
			if (li < P.list_cap) list[li] = i | (um << 20); else *v_ovf = 1;
			z = a.h1;
			for (int j = 0; j < nh; ++j) {
				uint32_t b = bloom_next(z, a.h2);
				if ((um >> j) & 1u) {
					if (*v_fs_n >= fs_limit || !fs_insert<false>(fs, fs_mask, bl * 512 + b, idx, &s_fs_n)) { *v_ovf = 1; break; }
				}
			}
		}
	}
	__syncthreads();

	bool dirty = true;
	if (!s_ovf) {
		dirty = s_list_n != 0;
		// ---- pass 2 (fast): k-mers with clear bits; seen iff an earlier k-mer of the batch sets each of them
		const uint32_t ln = s_list_n;
		for (uint32_t li = threadIdx.x; li < ln; li += BT) {
			uint32_t i = list[li] & 0xfffffu, um = list[li] >> 20;
			uint64_t y0, y1; uint32_t idx; bool hi;
			Rec<RW>::unpack(recs + (uint64_t)i * RW, y0, y1, idx, hi);
			uint64_t hash = bloom_hash<W>(P.k, (W)y0, (W)y1, m);
			BloomAddr a = bloom_addr(hash, P.bf_shift);
			const uint32_t bl = (uint32_t)a.blk & rmask;
			uint32_t z = a.h1; bool first = false;
			for (int j = 0; j < nh; ++j) {
				uint32_t b = bloom_next(z, a.h2);
				if ((um >> j) & 1u) {
					uint32_t fi = 0xffffffffu;
					fs_lookup<false>(fs, fs_mask, bl * 512 + b, fi);
					first |= (fi == idx);
					atomicOr(&region[bl * 16 + (b >> 5)], 1u << (b & 31));
				}
			}
			if (A.seen_out) A.seen_out[idx] = first ? 1 : 2;
			if (!first) { ++n_seen; emit_seen<W, RW>(P, A, y0, y1, hi, hash); }
		}
	} else {
		// ---- slow path: first-setter table in HBM (slice of the pool), sized by the region's bit count
		uint64_t want = (uint64_t)n * nh * 2;
		uint64_t lim = (uint64_t)region_blocks * 512 * 2;
		if (want > lim) want = lim;
		uint32_t cap = 1024; while (cap < want) cap <<= 1;
		if (threadIdx.x == 0) {
			unsigned long long off = atomicAdd(&A.pool[0], (unsigned long long)cap); // pool[0] is the bump cursor; entries start at pool[1]
			if (off + cap > A.pool_cap) { atomicAdd(&A.stats[ST_ERR_POOL], 1ULL); s_pool_off = 0xffffffffu; }
			else s_pool_off = (uint32_t)(off >> 10);
			atomicAdd(&A.stats[ST_SLOW_BUCKETS], 1ULL);
		}
		__syncthreads();
		if (s_pool_off == 0xffffffffu) return; // batch abandoned: the host sees ST_ERR_POOL and aborts
		unsigned long long *gfs = A.pool + 1 + ((uint64_t)s_pool_off << 10);
		for (uint32_t i = threadIdx.x; i < cap; i += BT) gfs[i] = FS_EMPTY;
		__threadfence();
		__syncthreads();
		const uint32_t gmask = cap - 1;
		for (uint32_t i = threadIdx.x; i < n; i += BT) {
			uint64_t y0, y1; uint32_t idx; bool hi;
			Rec<RW>::unpack(recs + (uint64_t)i * RW, y0, y1, idx, hi);
			uint64_t hash = bloom_hash<W>(P.k, (W)y0, (W)y1, m);
			BloomAddr a = bloom_addr(hash, P.bf_shift);
			const uint32_t bl = (uint32_t)a.blk & rmask;
			uint32_t z = a.h1;
			for (int j = 0; j < nh; ++j) {
				uint32_t b = bloom_next(z, a.h2);
				if (!((region[bl * 16 + (b >> 5)] >> (b & 31)) & 1u)) fs_insert<true>(gfs, gmask, bl * 512 + b, idx, nullptr);
			}
		}
		__threadfence();
		__syncthreads();
		for (uint32_t i = threadIdx.x; i < n; i += BT) {
			uint64_t y0, y1; uint32_t idx; bool hi;
			Rec<RW>::unpack(recs + (uint64_t)i * RW, y0, y1, idx, hi);
			uint64_t hash = bloom_hash<W>(P.k, (W)y0, (W)y1, m);
			BloomAddr a = bloom_addr(hash, P.bf_shift);
			const uint32_t bl = (uint32_t)a.blk & rmask;
			uint32_t z = a.h1; bool first = false, unresolved = false;
			for (int j = 0; j < nh; ++j) {
				uint32_t b = bloom_next(z, a.h2), fi;
				if (fs_lookup<true>(gfs, gmask, bl * 512 + b, fi)) { // has an entry <=> was clear before the batch
					unresolved = true; first |= (fi == idx);
					atomicOr(&region[bl * 16 + (b >> 5)], 1u << (b & 31));
				}
			}
			if (unresolved) {
				if (A.seen_out) A.seen_out[idx] = first ? 1 : 2;
				if (!first) { ++n_seen; emit_seen<W, RW>(P, A, y0, y1, hi, hash); }
			}
		}
	}
	__syncthreads();
	if (dirty) { // write the region back
		uint4 *dst = reinterpret_cast<uint4 *>(g_region);
		const uint4 *src = reinterpret_cast<const uint4 *>(region);
		for (uint32_t i = threadIdx.x; i < region_dw / 4; i += BT) dst[i] = src[i];
	}
	for (int o = 32; o; o >>= 1) n_seen += __shfl_down(n_seen, o);
	if ((threadIdx.x & 63) == 0 && n_seen) atomicAdd(&A.stats[ST_SEEN], (unsigned long long)n_seen);
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
// host-callable launchers (C++ linkage, used by bfcg_ctx.hip)

namespace bfcg {

static inline int grid_for(int64_t n_tiles, int cap) { return (int)(n_tiles < cap ? (n_tiles > 0 ? n_tiles : 1) : cap); }

#define TILE1 4096
#define BT1 256
#define TILE2 2048
#define BT2 256
#define BTB 1024

template <typename W, int RW>
static void run_batch_t(const KParams &P, const BatchBufs &B, const uint8_t *seq, const uint8_t *qual, int64_t n_pos, hipStream_t st, hipEvent_t *ev)
{
	const int nb1 = 1 << P.F1, nfine = 1 << P.F;
	const int64_t tiles1 = (n_pos + TILE1 - 1) / TILE1;
	hipMemsetAsync(B.cnt1, 0, sizeof(uint32_t) * (nb1 + 1), st);
	if (ev) hipEventRecord(ev[0], st);
	hipLaunchKernelGGL((k_hist1<W, TILE1, BT1>), dim3(grid_for(tiles1, 2048)), dim3(BT1), 0, st, P, seq, qual, n_pos, B.cnt1, B.stats);
	hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, B.cnt1, B.start1, B.cursor1, nb1);
	if (ev) hipEventRecord(ev[1], st);
	hipLaunchKernelGGL((k_scatter1<W, RW, TILE1, BT1>), dim3(grid_for(tiles1, 8192)), dim3(BT1), 0, st, P, seq, qual, n_pos, B.start1, B.cursor1, B.recs1);
	if (ev) hipEventRecord(ev[2], st);
	const uint64_t *fine_recs = B.recs1; const uint32_t *fine_start = B.start1;
	if (P.F2 > 0) {
		hipMemsetAsync(B.cnt2, 0, sizeof(uint32_t) * (nfine + 1), st);
		int gx = (int)((B.max_kmers / nb1) / (BT2 * 8) + 1); if (gx > 64) gx = 64;
		hipLaunchKernelGGL((k_hist2<W, RW, BT2>), dim3(gx, nb1), dim3(BT2), 0, st, P, B.recs1, B.start1, B.cnt2);
		hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, B.cnt2, B.start2, B.cursor2, nfine);
		int gx2 = (int)((B.max_kmers / nb1) / TILE2 + 1); if (gx2 > 256) gx2 = 256;
		hipLaunchKernelGGL((k_scatter2<W, RW, TILE2, BT2>), dim3(gx2, nb1), dim3(BT2), 0, st, P, B.recs1, B.start1, B.start2, B.cursor2, B.recs2);
		fine_recs = B.recs2; fine_start = B.start2;
	}
	if (ev) hipEventRecord(ev[3], st);
	BloomArgs A;
	A.recs = fine_recs; A.start = fine_start; A.bloom = B.bloom; A.bloom_hi = B.bloom_hi; A.table = B.table; A.stats = B.stats;
	A.tab_ovf = B.tab_ovf; A.tab_ovf_cap = B.tab_ovf_cap; A.pool = B.pool; A.pool_cap = B.pool_cap; A.seen_out = B.seen_out;
	size_t lds = ((size_t)64 << P.R) + (size_t)P.fs_cap * 8 + (size_t)P.list_cap * 4;
	hipLaunchKernelGGL((k_bloom<W, RW, BTB>), dim3(nfine), dim3(BTB), lds, st, P, A);
	if (ev) hipEventRecord(ev[4], st);
}

void run_batch(const KParams &P, const BatchBufs &B, const uint8_t *seq, const uint8_t *qual, int64_t n_pos, hipStream_t st, hipEvent_t *ev)
{
	if (P.k <= 32) run_batch_t<uint32_t, 2>(P, B, seq, qual, n_pos, st, ev);
	else if (P.k <= 47) run_batch_t<uint64_t, 2>(P, B, seq, qual, n_pos, st, ev);
	else run_batch_t<uint64_t, 3>(P, B, seq, qual, n_pos, st, ev);
}

int bloom_lds_bytes(const KParams &P) { return (int)(((size_t)64 << P.R) + (size_t)P.fs_cap * 8 + (size_t)P.list_cap * 4); }

hipError_t set_bloom_lds_attr(const KParams &P)
{
	int lds = bloom_lds_bytes(P);
	hipError_t e;
	e = hipFuncSetAttribute((const void *)k_bloom<uint32_t, 2, BTB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); if (e != hipSuccess) return e;
	e = hipFuncSetAttribute((const void *)k_bloom<uint64_t, 2, BTB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); if (e != hipSuccess) return e;
	e = hipFuncSetAttribute((const void *)k_bloom<uint64_t, 3, BTB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
	return e;
}

void run_hash_only(const KParams &P, const uint8_t *seq, const uint8_t *qual, int64_t n_pos, uint64_t *out, hipStream_t st)
{
	const int64_t tiles = (n_pos + TILE1 - 1) / TILE1;
	if (P.k <= 32) hipLaunchKernelGGL((k_hash_only<uint32_t, TILE1, BT1>), dim3(grid_for(tiles, 4096)), dim3(BT1), 0, st, P, seq, qual, n_pos, out);
	else hipLaunchKernelGGL((k_hash_only<uint64_t, TILE1, BT1>), dim3(grid_for(tiles, 4096)), dim3(BT1), 0, st, P, seq, qual, n_pos, out);
}

void run_table_replay(const KParams &P, unsigned long long *tab, const uint64_t *src, uint64_t n, unsigned long long *stats, uint64_t *ovf, uint32_t ovf_cap, hipStream_t st)
{
	int g = (int)((n + 255) / 256); if (g > 4096) g = 4096; if (g < 1) g = 1;
	hipLaunchKernelGGL(k_table_replay, dim3(g), dim3(256), 0, st, P, tab, src, n, stats, ovf, ovf_cap);
}
void run_table_rehash(const KParams &P, const unsigned long long *old_tab, int cshift_old, unsigned long long *new_tab, hipStream_t st)
{
	hipLaunchKernelGGL(k_table_rehash, dim3(4096), dim3(256), 0, st, P, old_tab, cshift_old, new_tab);
}

} // namespace bfcg
