// bfcg_internal.h -- shared between the kernels (bfcg_kernels.hip) and the host context (bfcg_ctx.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bfcg {

// statistics block in device memory (u64 counters)
enum { ST_KMERS = 0, ST_HIGH, ST_SEEN, ST_KEYS, ST_TAB_OVF, ST_ERR_POOL, ST_SLOW_BUCKETS, ST_N = 16 };

struct KParams {
	int k, q, bf_shift, n_hashes, l_pre, filter_mode;
	int R;          // log2(bloom blocks per LDS region)
	int F, F1, F2;  // fine-bucket bits = bf_shift-9-R, split over two scatter levels (F2 == 0: one level)
	int tab_cshift; // log2(slots per sub-table region)
	uint32_t fs_cap, list_cap; // LDS first-setter table entries (pow2), unresolved-list entries
};

struct BatchBufs {
	uint32_t *cnt1, *start1, *cursor1;  // level-1 histogram / starts / cursors (2^F1 + 1)
	uint32_t *cnt2, *start2, *cursor2;  // fine ...                             (2^F + 1)
	uint64_t *recs1, *recs2;            // record buffers, max_kmers * RW words each
	uint64_t max_kmers;
	unsigned long long *bloom, *bloom_hi, *table;
	unsigned long long *stats;
	uint64_t *tab_ovf; uint32_t tab_ovf_cap;
	unsigned long long *pool; unsigned long long pool_cap;
	uint8_t *seen_out;
};

void run_batch(const KParams &P, const BatchBufs &B, const uint8_t *seq, const uint8_t *qual, int64_t n_pos, hipStream_t st, hipEvent_t *ev);
int bloom_lds_bytes(const KParams &P);
hipError_t set_bloom_lds_attr(const KParams &P);
void run_hash_only(const KParams &P, const uint8_t *seq, const uint8_t *qual, int64_t n_pos, uint64_t *out, hipStream_t st);
void run_table_replay(const KParams &P, unsigned long long *tab, const uint64_t *src, uint64_t n, unsigned long long *stats, uint64_t *ovf, uint32_t ovf_cap, hipStream_t st);
void run_table_rehash(const KParams &P, const unsigned long long *old_tab, int cshift_old, unsigned long long *new_tab, hipStream_t st);

} // namespace bfcg
