// bfcg_mg.hip -- multi-GPU counting in C (include/bfc_gpu.h, PART 2: bfcg_group_*): the owner-computes partition of DESIGN.md section 5
// driven from inside the library, so that bfc_count() itself fans a file out over the GPUs of a node (the reference fans reads out over
// threads inside bfc_count: count.c:106 kt_for, count.c:143 kt_pipeline).
//
// A *group* owns the LOCAL ranks of a run of n_ranks: all of them in one process (bfc_count with BFC_GPU_DEVICES=0,1,...), or one per
// process (bench.py under torch.distributed.run; the RCCL unique id travels out of band).  Every local rank has its own device, counting
// context (bfcg_ctx_t with rank / n_ranks), exchange buffers, exchange stream and host thread.  Per global batch, on every rank:
//     stage A    K1 + level-1 scatter of the rank's share into its send buffer, grouped by level-1 bucket          (bfcg_mg_scatter)
//     sizes      every rank learns every rank's bucket sizes: shared memory between local ranks, ncclAllGather between processes
//     records    bucket b belongs to rank b / nb_loc: grouped ncclSend / ncclRecv, ring-shifted peer order, <= 256 MiB per message
//                (RCCL over xGMI), or -- local ranks only -- direct peer copies (hipMemcpyPeerAsync, the xGMI DMA path without RCCL)
//     stage B    level 2 + bloom regions + table on the owner, ordered behind the exchange by events, left running        (bfcg_mg_process)
// No bitmap reduce and no table merge arithmetic: slices and key sets are disjoint by construction (the union of the ranks' tables IS the
// reference's table, bfc_ch_union).  File order is rank-major inside a global batch.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <vector>
#include "bfc_gpu.h"
#include "bfcg_internal.h"
#include "bfc_host.h"

extern "C" int bfcg_mg_process_ev(bfcg_ctx_t *c, const void *d_recv, const uint32_t *seg_cnt, hipEvent_t *wait, int n_wait);
extern "C" int bfcg_mg_process_slabs(bfcg_ctx_t *c, const void *d_recv, const uint32_t *fills, uint32_t slab_cap, hipEvent_t *wait, int n_wait);
extern "C" int bfcg_mg_slab_info(bfcg_ctx_t *c, uint32_t out[2]);
extern "C" int bfcg_mg_scatter_slabs(bfcg_ctx_t *c, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t n_pos, void *d_send, uint32_t own_delta, uint32_t *fills, int *overflow);
extern "C" int bfcg_mg_scatter_again(bfcg_ctx_t *c, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t n_pos, void *d_send, uint32_t *counts);
extern "C" int bfcg_mg_row_words(bfcg_ctx_t *c);
extern "C" int bfcg_mg_async_ok(bfcg_ctx_t *c);
extern "C" int bfcg_mg_scatter_slabs_async(bfcg_ctx_t *c, const uint8_t *d_seq, const uint8_t *d_qual, uint64_t n_pos, void *d_send, uint32_t own_delta, uint32_t *d_rows_out, hipEvent_t *done);
extern "C" int bfcg_mg_scatter_slabs_wait(bfcg_ctx_t *c, uint32_t *fills, int *overflow);
extern "C" int bfcg_mg_process_slabs_dev(bfcg_ctx_t *c, const void *d_recv, const uint32_t *d_rows_in, uint32_t slab_cap, uint64_t rec_bound, int s_lo, int s_hi, hipEvent_t *wait, int n_wait);
extern "C" int bfcg_mg_process_finish(bfcg_ctx_t *c, const uint32_t *fills);
extern "C" void bfcg_set_error(const char *msg);
extern "C" double bfcg_mg_warm_factor(bfcg_ctx_t *c);
extern "C" void bfcg_mg_allow_onepass(bfcg_ctx_t *c, int on);
extern "C" int bfcg_resident_register(const void *bf, void *dev, int device, int n_shift);
extern "C" void *bfcg_bloom_slice(bfcg_ctx_t *c, int which, uint64_t *bytes);

namespace {

enum { XP_RCCL = 1, XP_PEER = 2, XP_PUSH = 3 };

// PUSH transport (round 6; ranks of one process whose devices reach each other's memory): ONE kernel per rank and global batch writes the FILLED
// part of every slab of this rank's stage A straight into its owner's receive buffer -- peer-mapped memory, stores over xGMI, the mechanism RCCL's own
// send / receive kernels use -- and the rank's row of fills beside it.  The fills are read on the device (the rows k_pack_rows made), so the bytes on
// the links are the records, exactly, and the host still knows no size when it enqueues this (the lazy protocol of round 5 sent whole slabs with
// their unfilled ends for that reason: +40 % bytes).  A work item = one slab of one destination; persistent workgroups stride over them.
#define PUSH_MAX_RANKS 16
struct PushDst { uint8_t *recv[PUSH_MAX_RANKS]; uint32_t *rows[PUSH_MAX_RANKS]; };
__global__ __launch_bounds__(256) void k_push_slabs(const uint8_t *__restrict__ send, const uint32_t *__restrict__ rows_out, PushDst D, int N, int me, uint32_t per /* slabs per destination */,
                                                   uint32_t cap, uint32_t rb, uint32_t row_w, unsigned long long *__restrict__ moved)
{
	const uint32_t n_items = (uint32_t)(N - 1) * per;
	unsigned long long mine = 0;
	for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
		const int to = (me + 1 + (int)(item / per)) % N;
		const uint32_t i = item % per;
		const uint32_t *row = rows_out + (size_t)to * row_w;
		uint32_t fill = row[per] ? 0u : row[i]; // (row[per] != 0: a slab of this stage A overflowed, or the stage failed -- nothing of it may be used; the row says so to the owner)
		if (fill > cap) fill = cap;
		const uint64_t blk = (uint64_t)per * cap;
		const uint8_t *src = send + ((uint64_t)to * blk + (uint64_t)i * cap) * rb;
		uint8_t *dst = D.recv[to] + ((uint64_t)me * blk + (uint64_t)i * cap) * rb;
		const uint64_t nbytes = (uint64_t)fill * rb; // (a multiple of 4: records are 12, 16 or 20 bytes)
		if (((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0) {
			const uint64_t n16 = nbytes >> 4;
			for (uint64_t j = threadIdx.x; j < n16; j += blockDim.x) reinterpret_cast<uint4 *>(dst)[j] = reinterpret_cast<const uint4 *>(src)[j];
			for (uint64_t j = (n16 << 2) + threadIdx.x; j < (nbytes >> 2); j += blockDim.x) reinterpret_cast<uint32_t *>(dst)[j] = reinterpret_cast<const uint32_t *>(src)[j];
		} else
			for (uint64_t j = threadIdx.x; j < (nbytes >> 2); j += blockDim.x) reinterpret_cast<uint32_t *>(dst)[j] = reinterpret_cast<const uint32_t *>(src)[j];
		if (threadIdx.x == 0) mine += nbytes;
	}
	if (blockIdx.x < (unsigned)(N - 1)) { // the rows: workgroup b carries the one for the b-th peer
		const int to = (me + 1 + (int)blockIdx.x) % N;
		for (uint32_t j = threadIdx.x; j < row_w; j += blockDim.x) D.rows[to][(size_t)me * row_w + j] = rows_out[(size_t)to * row_w + j];
		if (threadIdx.x == 0) mine += (unsigned long long)row_w * 4u;
	}
	if (threadIdx.x == 0 && mine) atomicAdd(moved, mine);
	__threadfence_system(); // the stores are another device's memory: visible there before this kernel is seen as complete
}
const uint64_t MSG_BYTES = 256ull << 20; // this image's RCCL truncates single messages above 1 GiB (measured): stay far below

struct rank_t {
	int rank, device;
	bfcg_ctx_t *ctx;
	hipStream_t xs, cs;        // exchange stream; staging copies of host batches
	hipEvent_t ev_x;           // this rank's part of the exchange is done (its receives with RCCL; its outgoing copies with peer copies)
	hipEvent_t ev_sent[2]; int sent_pending[2]; // the exchange that reads send buffer [t & 1] has drained (waited for before stage A writes it again)
	uint8_t *send2[2], *recv[2]; // send buffers alternate, so that stage A of the next batch runs beside this batch's exchange
	int combined;              // recv[b] lies behind send2[b] in ONE allocation (slab mode: stage A writes the rank's own share straight into it)
	uint64_t send_cap, recv_cap; // bytes
	uint32_t *counts;          // this rank's level-1 bucket sizes of the current batch (host) + one word: this rank's group has failed
	uint32_t *d_counts;        // multi-process: all ranks' rows, device side of the all-gather
	uint32_t *d_rows_out[2], *d_rows_in[2]; // slab mode, all ranks in this process: the fills of stage A's slabs as one row per destination (device), and every
	                           // source's row for this rank -- they travel beside the blocks, the owner's stage B reads them on the device (rank_batch: `lazy`)
	unsigned long long *d_moved; // PUSH: bytes k_push_slabs has written to peers (device counter, read at bfcg_group_exchange_bytes)
	uint64_t batch_call[64];   // the context's call number after stage B of global batch t (t & 63): bfcg_group_progress
	ncclComm_t comm;
	uint8_t *d_seq, *d_qual; uint64_t in_cap; // staging of host batches
	// the current batch's share
	const uint8_t *in_seq, *in_qual; uint64_t in_pos; int in_host;
	pthread_t th;
	int rc;
};

} // namespace

struct bfcg_group {
	bfcg_params_t prm;
	int n_ranks, first, n_local, xp, rec_bytes, nb1, nb_loc;
	int mp;                     // one rank per process: sizes travel by ncclAllGather, records by ncclSend / ncclRecv between the processes
	// Slab mode (round 4): stage A is ONE pass (K1 once) into 8 slabs per level-1 bucket of the send buffer; a destination's buckets are one
	// contiguous range of slabs and travel as they are (fill: 8 words per bucket in the sizes), the rank's own share is written into its
	// receive buffer by the kernel itself (no self-copy).  A slab that overflows anywhere (skewed input) sends the batch -- and the rest of the
	// run -- back to the two-pass stage A with exact, contiguous buckets.
	int slabs, slabs_ok;        // in use now; possible at all (every context, the buffers' record indices fit 32 bits)
	uint32_t slab_cap; uint64_t blk; // records per slab; per block of nb_loc x 8 slabs (what one rank sends to one destination)
	size_t row_words;           // a rank's row of sizes: up to 8 x nb1 words, then its failure word, then its overflow word
	uint64_t kmer_limit;        // k-mers of a global batch one rank's regions take at full speed
	int lazy_ok, lazy;          // the sizes reach the host AFTER exchange and stage B are enqueued: possible at all (slab mode, every context able to take its stage B from rows on the device -- in-process AND multi-process groups); for the current batch
	uint64_t n_lazy;            // global batches taken that way
	int push;                   // XP_PEER with the PUSH kernel for lazy batches (k_push_slabs: exact bytes); everything else of the peer-copy transport as it was
	unsigned push_wgs;          // its persistent workgroups per rank
	unsigned long long x_links, x_exact; // bytes the local ranks put on the links so far (whole blocks / messages as sent), and the bytes of the live records among them
	std::vector<rank_t> r;
	uint32_t *all_counts;       // [n_ranks][nb1 + 1], host (pinned): every rank's bucket sizes of the current batch and its failure word
	pthread_barrier_t bar;      // local ranks
	pthread_mutex_t mu; pthread_cond_t cv;
	uint64_t job, done_job; int n_done, quit, failed, go;
	uint64_t t;                 // global batches so far
	char err[512];
};

static int drain_exchange(bfcg_group_t *g);
static void grp_err(bfcg_group_t *g, const char *fmt, ...)
{
	pthread_mutex_lock(&g->mu);
	if (!g->failed) {
		va_list ap; va_start(ap, fmt); vsnprintf(g->err, sizeof(g->err), fmt, ap); va_end(ap);
		g->failed = 1;
	}
	pthread_mutex_unlock(&g->mu);
}
// a failing call marks the whole group as failed; the rank threads never leave a batch early -- they all walk through the same barriers
#define GHIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) grp_err(g, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define GNCCL(call) do { ncclResult_t e_ = (call); if (e_ != ncclSuccess) grp_err(g, "%s failed: %s (%s:%d)", #call, ncclGetErrorString(e_), __FILE__, __LINE__); } while (0)

// stage B of one global batch on the owner.  When the rank received more k-mers than its bloom regions take at full speed the sources are
// processed in consecutive groups, each its own stage B: the receive buffer is source-major and the global order rank-major, so a group's
// records all precede the next group's -- the result is that of one big batch without every region overflowing its LDS list.  Not with
// order stamps (the batch number is part of a stamp and must agree across ranks).
static int process_in_groups(bfcg_group_t *g, rank_t &R, const uint8_t *recv, const uint32_t *seg_cnt, hipEvent_t *wait, int n_wait)
{
	const int N = g->n_ranks, nb_loc = g->nb_loc;
	std::vector<uint64_t> per_src((size_t)N, 0);
	uint64_t total = 0;
	for (int s = 0; s < N; ++s) { for (int k = 0; k < nb_loc; ++k) per_src[s] += seg_cnt[(size_t)s * nb_loc + k]; total += per_src[s]; }
	// (the limit follows the filter's fill, as a single GPU's batch cuts do: once most k-mers are seen again a region takes ~3 x as many per pass,
	// and every pass streams the rank's share of the filter and of the table segments)
	const uint64_t limit = (uint64_t)((double)g->kmer_limit * bfcg_mg_warm_factor(R.ctx));
	if (g->prm.track_order || total <= limit || N == 1) return bfcg_mg_process_ev(R.ctx, recv, seg_cnt, wait, n_wait);
	std::vector<uint32_t> seg((size_t)N * nb_loc);
	uint64_t off = 0;
	int s0 = 0, launched = 0;
	while (s0 < N) {
		uint64_t acc = per_src[s0]; int s1 = s0 + 1;
		while (s1 < N && acc + per_src[s1] <= limit) acc += per_src[s1++];
		if (acc) {
			memset(seg.data(), 0, seg.size() * sizeof(uint32_t));
			memcpy(&seg[(size_t)s0 * nb_loc], &seg_cnt[(size_t)s0 * nb_loc], sizeof(uint32_t) * (size_t)(s1 - s0) * nb_loc);
			if (launched) { // stage A and stage B come in pairs (buffer sets, timing events): an empty stage A opens the next pair
				std::vector<uint32_t> dummy((size_t)g->nb1);
				if (bfcg_mg_scatter(R.ctx, 0, 0, 0, 0, dummy.data()) != 0) return -1;
			}
			if (bfcg_mg_process_ev(R.ctx, recv + off * (uint64_t)g->rec_bytes, seg.data(), launched ? 0 : wait, launched ? 0 : n_wait) != 0) return -1;
			++launched;
		}
		for (int s = s0; s < s1; ++s) off += per_src[s];
		s0 = s1;
	}
	if (!launched) return bfcg_mg_process_ev(R.ctx, recv, seg_cnt, wait, n_wait);
	return 0;
}

// the same in slab mode: every source's slabs sit at fixed places, so a group of sources is simply the fills of the others set to zero
// (s_begin, launched0: the sources before s_begin went through launched0 passes already -- the lazy protocol's first group, rank_batch)
static int process_in_groups_slabs(bfcg_group_t *g, rank_t &R, const uint8_t *recv, const uint32_t *fills, hipEvent_t *wait, int n_wait, int s_begin = 0, int launched0 = 0)
{
	const int N = g->n_ranks, nb_loc = g->nb_loc;
	const size_t per = (size_t)nb_loc * 8;
	std::vector<uint64_t> per_src((size_t)N, 0);
	uint64_t total = 0;
	for (int s = s_begin; s < N; ++s) { for (size_t k = 0; k < per; ++k) per_src[s] += fills[(size_t)s * per + k]; total += per_src[s]; }
	const uint64_t limit = (uint64_t)((double)g->kmer_limit * bfcg_mg_warm_factor(R.ctx));
	if (!launched0 && (g->prm.track_order || total <= limit || N == 1)) return bfcg_mg_process_slabs(R.ctx, recv, fills, g->slab_cap, wait, n_wait);
	std::vector<uint32_t> seg((size_t)N * per);
	int s0 = s_begin, launched = launched0;
	while (s0 < N) {
		uint64_t acc = per_src[s0]; int s1 = s0 + 1;
		while (s1 < N && acc + per_src[s1] <= limit) acc += per_src[s1++];
		if (acc) {
			memset(seg.data(), 0, seg.size() * sizeof(uint32_t));
			memcpy(&seg[(size_t)s0 * per], &fills[(size_t)s0 * per], sizeof(uint32_t) * (size_t)(s1 - s0) * per);
			if (launched) { // stage A and stage B come in pairs (buffer sets, timing events): an empty stage A opens the next pair
				std::vector<uint32_t> dummy((size_t)g->nb1);
				if (bfcg_mg_scatter(R.ctx, 0, 0, 0, 0, dummy.data()) != 0) return -1;
			}
			if (bfcg_mg_process_slabs(R.ctx, recv, seg.data(), g->slab_cap, launched ? 0 : wait, launched ? 0 : n_wait) != 0) return -1;
			++launched;
		}
		s0 = s1;
	}
	if (!launched) return bfcg_mg_process_slabs(R.ctx, recv, fills, g->slab_cap, wait, n_wait);
	return 0;
}
// The lazy protocol's first group of sources: [0, s1) with s1 the most sources whose BOUND fits what this rank's regions take at full speed -- a source
// sends an owner at most its share's positions / N (k-mers <= positions; the hash spreads them evenly: + 5 %).  `pos[s]`: the shares' positions
// where this process knows them (all ranks local), else every share is taken at the contexts' capacity.
// A RANK-LOCAL decision: s1 only says in how many launches THIS owner applies what it received -- every source's records are applied
// either way, in file order by their indices -- so the processes of a multi-process group need not agree on it (the warm factor is per context
// and may differ between them); what they must decide alike is whether the batch is lazy at all: lazy_possible, which reads nothing rank-local.
static int lazy_first_group(bfcg_group_t *g, rank_t &R)
{
	const int N = g->n_ranks;
	if (N == 1 || g->prm.track_order) return N;
	const uint64_t limit = (uint64_t)((double)g->kmer_limit * bfcg_mg_warm_factor(R.ctx));
	uint64_t acc = 0;
	int s1 = 0;
	for (int s = 0; s < N; ++s) {
		const uint64_t p = g->mp ? g->prm.max_batch_pos : g->r[s - g->first].in_pos, est = p / (uint64_t)N + p / (uint64_t)N / 20;
		if (s1 > 0 && acc + est > limit) break;
		acc += est; s1 = s + 1;
	}
	return s1;
}

// every rank's row of sizes: shared memory between the local ranks, an all-gather over RCCL between processes
static void publish_sizes(bfcg_group_t *g, rank_t &R)
{
	const int N = g->n_ranks, me = R.rank;
	const size_t cs = g->row_words;
	memcpy(g->all_counts + (size_t)me * cs, R.counts, sizeof(uint32_t) * cs);
	if (g->mp) { // between processes: all-gather over RCCL (the local ranks of a multi-process group are one per process)
		ncclResult_t ne = ncclSuccess;
		hipError_t he = hipMemcpyAsync(R.d_counts + (size_t)me * cs, R.counts, sizeof(uint32_t) * cs, hipMemcpyHostToDevice, R.xs);
		if (he == hipSuccess) ne = ncclAllGather(R.d_counts + (size_t)me * cs, R.d_counts, cs, ncclUint32, R.comm, R.xs);
		if (he == hipSuccess && ne == ncclSuccess) he = hipMemcpyAsync(g->all_counts, R.d_counts, sizeof(uint32_t) * (size_t)N * cs, hipMemcpyDeviceToHost, R.xs);
		if (he == hipSuccess && ne == ncclSuccess) he = hipStreamSynchronize(R.xs);
		if (he != hipSuccess || ne != ncclSuccess) grp_err(g, "all-gather of the bucket sizes failed: %s", he != hipSuccess ? hipGetErrorString(he) : ncclGetErrorString(ne));
		else for (int p = 0; p < N; ++p) if (p != me && g->all_counts[(size_t)p * cs + cs - 2]) grp_err(g, "rank %d of the run has failed: this batch is not exchanged", p);
	}
}

// one global batch on local rank i (runs on the rank's own host thread)
static int rank_batch(bfcg_group_t *g, int i)
{
	rank_t &R = g->r[i];
	const int N = g->n_ranks, nb1 = g->nb1, nb_loc = g->nb_loc, me = R.rank;
	const size_t cs = g->row_words; // a rank's row: its sizes, then (cs - 2) its failure word, (cs - 1) a slab of its stage A overflowed
	const uint64_t rb = (uint64_t)g->rec_bytes;
	int ok = !g->failed;
	GHIP(hipSetDevice(R.device));
	// ---- stage A into send buffer [t & 1]: the exchange two batches ago read it last
	const int sb = (int)(g->t & 1);
	uint8_t *const send = R.send2[sb];
	uint8_t *recv = R.recv[g->t & 1];
	if (R.sent_pending[sb]) { GHIP(hipEventSynchronize(R.ev_sent[sb])); R.sent_pending[sb] = 0; }
	const uint8_t *ds = R.in_seq, *dq = R.in_qual;
	int slab = g->slabs; // (the same on every rank: decided between batches)
	const int lazy = slab && g->lazy; // (decided by run_job for this batch)
	int redo = 0;        // lazy: a slab overflowed somewhere -- found after this batch's (empty) stage B was enqueued
	memset(R.counts, 0, sizeof(uint32_t) * cs);
	if (ok) {
		if (R.in_host && R.in_pos) {
			if (R.in_pos > R.in_cap) { grp_err(g, "share of %llu positions exceeds the staging buffer", (unsigned long long)R.in_pos); ok = 0; }
			else {
				GHIP(hipMemcpyAsync(R.d_seq, R.in_seq, R.in_pos, hipMemcpyHostToDevice, R.cs));
				if (R.in_qual) GHIP(hipMemcpyAsync(R.d_qual, R.in_qual, R.in_pos, hipMemcpyHostToDevice, R.cs));
				GHIP(hipStreamSynchronize(R.cs));
				ds = R.d_seq; dq = R.in_qual ? R.d_qual : 0;
			}
		}
	}
	if (lazy) {
		// ---- Slab mode without the host in the batch's loop (round 5).  The exchange never depended on the sizes (whole blocks travel); only the
		// owner's stage B did, and only through the host.  Now every source's fills travel beside its block as a ROW in device memory
		// (k_pack_rows), the owner builds its segment arrays from the rows on the device (k_seg_setup_mg), and stage A, exchange and stage B are
		// all enqueued before this thread looks at anything: it reads its own copy of the rows -- the overflow decision, the sizes a replay of
		// this stage B would need -- while the device still has the whole of stage B before it.  A slab that overflowed anywhere marks its row;
		// every owner sees every row, so stage B of the batch moves nothing on every rank, and the ranks repeat it through the two passes
		// below exactly as before -- the decision still falls inside this call, the caller's buffers are still his.
		const size_t rw_ = (size_t)bfcg_mg_row_words(R.ctx);
		hipEvent_t ev_a = 0;
		int a_ok = ok;
		if (ok && bfcg_mg_scatter_slabs_async(R.ctx, ds, dq, R.in_pos, send, (uint32_t)(R.send_cap / rb), R.d_rows_out[sb], &ev_a) != 0) { grp_err(g, "rank %d: %s", me, bfcg_last_error()); ok = a_ok = 0; }
		pthread_barrier_wait(&g->bar); // every local rank has enqueued its stage A, or failed to
		if (!g->mp) ok = !g->failed;
		// Between processes a rank cannot tell its peers that it has failed before they post their sends and receives for it: it takes part in
		// the exchange all the same, with rows that say "nothing of this may be used" (every owner's stage B of the batch then moves nothing),
		// and the failure word of the sizes' all-gather below ends the run on every process in this same batch.
		const int post = g->mp ? 1 : ok;
		if (post) {
			if (a_ok) GHIP(hipStreamWaitEvent(R.xs, ev_a, 0));
			else GHIP(hipMemsetAsync(R.d_rows_out[sb], 0xff, sizeof(uint32_t) * (size_t)N * rw_, R.xs));
			uint32_t *const rin = R.d_rows_in[g->t & 1];
			GHIP(hipMemcpyAsync(rin + (size_t)me * rw_, R.d_rows_out[sb] + (size_t)me * rw_, sizeof(uint32_t) * rw_, hipMemcpyDeviceToDevice, R.xs)); // (the own block was written in place)
			if (g->xp == XP_RCCL) {
				if (N > 1) GNCCL(ncclGroupStart());
				for (int step = 1; step < N; ++step) {
					const int to = (me + step) % N, from = (me - step + N) % N;
					const uint64_t nbytes = g->blk * rb;
					for (uint64_t c0 = 0; c0 < nbytes; c0 += MSG_BYTES) {
						GNCCL(ncclSend(send + (uint64_t)to * nbytes + c0, (size_t)(nbytes - c0 < MSG_BYTES ? nbytes - c0 : MSG_BYTES), ncclUint8, to, R.comm, R.xs));
						GNCCL(ncclRecv(recv + (uint64_t)from * nbytes + c0, (size_t)(nbytes - c0 < MSG_BYTES ? nbytes - c0 : MSG_BYTES), ncclUint8, from, R.comm, R.xs));
					}
					GNCCL(ncclSend(R.d_rows_out[sb] + (size_t)to * rw_, rw_, ncclUint32, to, R.comm, R.xs));
					GNCCL(ncclRecv(rin + (size_t)from * rw_, rw_, ncclUint32, from, R.comm, R.xs));
				}
				if (N > 1) GNCCL(ncclGroupEnd());
			} else if (g->push && N > 1) {
				PushDst D;
				memset(&D, 0, sizeof(D));
				for (int p = 0; p < N; ++p) { const rank_t &T = g->r[p - g->first]; D.recv[p] = T.recv[g->t & 1]; D.rows[p] = T.d_rows_in[g->t & 1]; }
				const uint32_t per = (uint32_t)nb_loc * 8u;
				unsigned wgs = g->push_wgs;
				if (wgs > (unsigned)(N - 1) * per) wgs = (unsigned)(N - 1) * per;
				if (wgs < (unsigned)(N - 1)) wgs = (unsigned)(N - 1);
				hipLaunchKernelGGL(k_push_slabs, dim3(wgs), dim3(256), 0, R.xs, (const uint8_t *)send, (const uint32_t *)R.d_rows_out[sb], D, N, me, per, g->slab_cap, (uint32_t)rb, (uint32_t)rw_, R.d_moved);
				GHIP(hipGetLastError());
			} else {
				for (int step = 1; step < N; ++step) {
					const int to = (me + step) % N;
					const rank_t &T = g->r[to - g->first];
					GHIP(hipMemcpyPeerAsync(T.recv[g->t & 1] + (uint64_t)me * g->blk * rb, T.device, send + (uint64_t)to * g->blk * rb, R.device, g->blk * rb, R.xs));
					GHIP(hipMemcpyPeerAsync(T.d_rows_in[g->t & 1] + (size_t)me * rw_, T.device, R.d_rows_out[sb] + (size_t)to * rw_, R.device, sizeof(uint32_t) * rw_, R.xs));
				}
			}
			if (!(g->push && g->xp == XP_PEER)) __atomic_fetch_add(&g->x_links, (unsigned long long)(N - 1) * (g->blk * rb + sizeof(uint32_t) * rw_), __ATOMIC_RELAXED);
			GHIP(hipEventRecord(R.ev_x, R.xs));
		}
		pthread_barrier_wait(&g->bar); // every sender's event is recorded: the owners may wait for them
		// the owner's stage B, from the rows on the device: all sources, or -- where that is more than this rank's regions take at full speed --
		// a first group of them chosen by what they can send at most; the others follow below, grouped by their sizes
		const int s1 = lazy_first_group(g, R);
		int enq = 0;
		if (ok && !g->failed) {
			std::vector<hipEvent_t> ev;
			if (g->xp == XP_RCCL) ev.push_back(R.ev_x);
			else for (int j = 0; j < g->n_local; ++j) ev.push_back(g->r[j].ev_x);
			if (bfcg_mg_process_slabs_dev(R.ctx, recv, R.d_rows_in[g->t & 1], g->slab_cap, (uint64_t)N * g->blk, 0, s1, ev.data(), (int)ev.size()) != 0) grp_err(g, "rank %d: %s", me, bfcg_last_error());
			else enq = 1;
		}
		// ---- only now the host's copy of the sizes
		int ovf = 0;
		if (a_ok && bfcg_mg_scatter_slabs_wait(R.ctx, R.counts, &ovf) != 0) { grp_err(g, "rank %d: %s", me, bfcg_last_error()); memset(R.counts, 0, sizeof(uint32_t) * cs); }
		if (!a_ok) memset(R.counts, 0, sizeof(uint32_t) * cs);
		if (post && N > 1) { // what of this rank's stage A belonged on the links: the live records of the other ranks' buckets
			unsigned long long live = 0;
			const size_t per = (size_t)nb_loc * 8;
			for (int p = 0; p < N; ++p) if (p != me) for (size_t i2 = 0; i2 < per; ++i2) live += R.counts[(size_t)p * per + i2];
			__atomic_fetch_add(&g->x_exact, live * rb, __ATOMIC_RELAXED);
		}
		R.counts[cs - 1] = ovf ? 1u : 0u;
		R.counts[cs - 2] = g->failed ? 1u : 0u;
		publish_sizes(g, R); // (between processes: the all-gather, behind this batch's exchange on the same stream)
		pthread_barrier_wait(&g->bar); // all local ranks have published their sizes
		for (int p = 0; p < N; ++p) redo |= g->all_counts[(size_t)p * cs + cs - 1] != 0;
		{
			const size_t per = (size_t)nb_loc * 8;
			std::vector<uint32_t> seg((size_t)N * per, 0u), first((size_t)N * per, 0u);
			if (!redo && !g->failed) {
				for (int s2 = 0; s2 < N; ++s2) memcpy(&seg[(size_t)s2 * per], &g->all_counts[(size_t)s2 * cs + (size_t)me * per], sizeof(uint32_t) * per);
				memcpy(first.data(), seg.data(), sizeof(uint32_t) * (size_t)s1 * per);
			}
			if (enq && bfcg_mg_process_finish(R.ctx, first.data()) != 0) grp_err(g, "rank %d: %s", me, bfcg_last_error());
			if (enq && !redo && !g->failed && s1 < N && process_in_groups_slabs(g, R, recv, seg.data(), 0, 0, s1, 1) != 0) grp_err(g, "rank %d: %s", me, bfcg_last_error());
		}
		pthread_barrier_wait(&g->bar); // (everybody has read the rows before anybody writes the next ones)
		ok = !g->failed;
		if (redo) { // the empty stage B and the exchange that carried the overflowed slabs: out of the way before the send buffer is written again
			if (bfcg_sync(R.ctx) != 0) grp_err(g, "rank %d: %s", me, bfcg_last_error());
			GHIP(hipStreamSynchronize(R.xs));
			ok = !g->failed;
		} else if (i == 0) ++g->n_lazy;
	} else
	if (ok) {
		if (ok && slab) {
			int ovf = 0;
			// (the rank's own slabs: the same place inside its block of the receive buffer, which lies send_cap bytes behind the send buffer)
			if (bfcg_mg_scatter_slabs(R.ctx, ds, dq, R.in_pos, send, (uint32_t)(R.send_cap / rb), R.counts, &ovf) != 0) { grp_err(g, "rank %d: %s", me, bfcg_last_error()); ok = 0; }
			R.counts[cs - 1] = ovf ? 1u : 0u;
		} else if (ok && bfcg_mg_scatter(R.ctx, ds, dq, R.in_pos, send, R.counts) != 0) { grp_err(g, "rank %d: %s", me, bfcg_last_error()); ok = 0; }
	}
	if (!lazy) {
	if (!ok) memset(R.counts, 0, sizeof(uint32_t) * cs);
	// The failure word travels with the sizes: g->failed is local to a process, and a rank that skipped the exchange while its peers posted
	// ncclSend / ncclRecv for it would leave them blocked for good.  A group that has failed keeps taking part in this all-gather (and only in it),
	// so that every process of the run takes the same decision in the same batch.
	R.counts[cs - 2] = g->failed ? 1u : 0u;
	publish_sizes(g, R);
	pthread_barrier_wait(&g->bar); // all local ranks have published their sizes
	}
	if (slab) { // a slab overflowed somewhere: every rank repeats its stage A through the two passes (its k-mers are counted) and the run stays with them
		int any = redo;
		if (!lazy) {
			for (int p = 0; p < N; ++p) any |= g->all_counts[(size_t)p * cs + cs - 1] != 0;
			pthread_barrier_wait(&g->bar); // (everybody has read the rows before anybody writes the next ones)
		}
		if (any) {
			if (i == 0 && getenv("BFCG_DEBUG_MG")) fprintf(stderr, "[D::group] batch %llu: a level-1 slab overflowed on some rank: two-pass stage A from here on\n", (unsigned long long)g->t);
			if (i == 0) g->slabs = 0;
			slab = 0;
			memset(R.counts, 0, sizeof(uint32_t) * cs);
			if (ok && bfcg_mg_scatter_again(R.ctx, ds, dq, R.in_pos, send, R.counts) != 0) { grp_err(g, "rank %d: %s", me, bfcg_last_error()); ok = 0; }
			if (!ok) memset(R.counts, 0, sizeof(uint32_t) * cs);
			R.counts[cs - 2] = g->failed ? 1u : 0u;
			publish_sizes(g, R);
			pthread_barrier_wait(&g->bar);
		}
	}
	if (!lazy || redo) { // (a lazy batch that went well is complete: its exchange and stage B were enqueued above)
	if (i == 0) g->go = !g->failed; // one decision for all local ranks -- and, the failure words being all-gathered, for all processes
	pthread_barrier_wait(&g->bar);
	const uint32_t *C = g->all_counts;
	ok = g->go;
	std::vector<uint64_t> s_off((size_t)N + 1, 0), r_off((size_t)N + 1, 0);
	if (slab) { // fixed places: block p of my send buffer goes to rank p, block s of my receive buffer comes from rank s
		for (int p = 0; p <= N; ++p) s_off[p] = r_off[p] = (uint64_t)p * g->blk;
	} else {
		// what I send to rank p: my buckets [p*nb_loc, (p+1)*nb_loc); what I get from rank p: its buckets [me*nb_loc, ...), stored source-major
		for (int p = 0; p < N; ++p) {
			uint64_t a = 0, q = 0;
			for (int k = 0; k < nb_loc; ++k) { a += C[(size_t)me * cs + (size_t)p * nb_loc + k]; q += C[(size_t)p * cs + (size_t)me * nb_loc + k]; }
			s_off[p + 1] = s_off[p] + a; r_off[p + 1] = r_off[p] + q;
		}
		// every rank can compute every rank's receive size: an overflow anywhere stops the exchange everywhere
		for (int p = 0; p < N && ok; ++p) {
			uint64_t q = 0;
			for (int s2 = 0; s2 < N; ++s2) for (int k = 0; k < nb_loc; ++k) q += C[(size_t)s2 * cs + (size_t)p * nb_loc + k];
			if (q * rb > R.recv_cap) { grp_err(g, "rank %d receives %llu records of one global batch, its buffer holds %llu: smaller shares or a larger filter", p, (unsigned long long)q, (unsigned long long)(R.recv_cap / rb)); ok = 0; }
		}
	}
	if (ok && N > 1) { // (the books: bytes as they are sent, and the live records among them)
		unsigned long long links = 0, live = 0;
		for (int p = 0; p < N; ++p) if (p != me) {
			links += (s_off[p + 1] - s_off[p]) * rb;
			if (slab) for (int i2 = 0; i2 < nb_loc * 8; ++i2) live += (unsigned long long)C[(size_t)me * cs + (size_t)p * nb_loc * 8 + i2] * rb;
		}
		__atomic_fetch_add(&g->x_links, links, __ATOMIC_RELAXED);
		__atomic_fetch_add(&g->x_exact, slab ? live : links, __ATOMIC_RELAXED);
	}
	// ---- records
	if (g->xp == XP_RCCL) {
		if (ok) {
			if (!slab && s_off[me + 1] > s_off[me]) GHIP(hipMemcpyAsync(recv + r_off[me] * rb, send + s_off[me] * rb, (s_off[me + 1] - s_off[me]) * rb, hipMemcpyDeviceToDevice, R.xs));
			if (N > 1) GNCCL(ncclGroupStart());
			for (int step = 1; step < N; ++step) { // ring-shifted peer order: every rank talks to a different peer at any time
				const int to = (me + step) % N, from = (me - step + N) % N;
				const uint64_t n_to = (s_off[to + 1] - s_off[to]) * rb, n_from = (r_off[from + 1] - r_off[from]) * rb;
				for (uint64_t c0 = 0; c0 < (n_to > n_from ? n_to : n_from); c0 += MSG_BYTES) {
					if (c0 < n_to) GNCCL(ncclSend(send + s_off[to] * rb + c0, (size_t)(n_to - c0 < MSG_BYTES ? n_to - c0 : MSG_BYTES), ncclUint8, to, R.comm, R.xs));
					if (c0 < n_from) GNCCL(ncclRecv(recv + r_off[from] * rb + c0, (size_t)(n_from - c0 < MSG_BYTES ? n_from - c0 : MSG_BYTES), ncclUint8, from, R.comm, R.xs));
				}
			}
			if (N > 1) GNCCL(ncclGroupEnd());
			GHIP(hipEventRecord(R.ev_x, R.xs));
		}
	} else { // peer copies: I push my records into every owner's receive buffer (all ranks are local)
		if (ok) {
			for (int step = slab ? 1 : 0; step < N; ++step) {
				const int to = (me + step) % N;
				const rank_t &T = g->r[to - g->first];
				const uint64_t n_to = (s_off[to + 1] - s_off[to]) * rb;
				// my block in `to`'s buffer starts behind the blocks of the sources before me
				uint64_t at = 0;
				if (slab) at = (uint64_t)me * g->blk;
				else for (int p = 0; p < me; ++p) for (int k = 0; k < nb_loc; ++k) at += C[(size_t)p * cs + (size_t)to * nb_loc + k];
				if (n_to) GHIP(hipMemcpyPeerAsync(T.recv[g->t & 1] + at * rb, T.device, send + s_off[to] * rb, R.device, n_to, R.xs));
			}
			GHIP(hipEventRecord(R.ev_x, R.xs));
		}
		pthread_barrier_wait(&g->bar); // every sender's event is recorded: the owners may wait for them
	}
	// ---- stage B behind the exchange
	if (ok && !g->failed) {
		std::vector<hipEvent_t> ev;
		if (g->xp == XP_RCCL) ev.push_back(R.ev_x);
		else for (int j = 0; j < g->n_local; ++j) ev.push_back(g->r[j].ev_x);
		if (slab) {
			const size_t per = (size_t)nb_loc * 8;
			std::vector<uint32_t> seg((size_t)N * per);
			for (int s = 0; s < N; ++s) memcpy(&seg[(size_t)s * per], &C[(size_t)s * cs + (size_t)me * per], sizeof(uint32_t) * per);
			if (process_in_groups_slabs(g, R, recv, seg.data(), ev.data(), (int)ev.size()) != 0) grp_err(g, "rank %d: %s", me, bfcg_last_error());
		} else {
			std::vector<uint32_t> seg((size_t)N * nb_loc);
			for (int s = 0; s < N; ++s) memcpy(&seg[(size_t)s * nb_loc], &C[(size_t)s * cs + (size_t)me * nb_loc], sizeof(uint32_t) * (size_t)nb_loc);
			if (process_in_groups(g, R, recv, seg.data(), ev.data(), (int)ev.size()) != 0) grp_err(g, "rank %d: %s", me, bfcg_last_error());
		}
	}
	}
	{ uint64_t calls = 0; bfcg_progress(R.ctx, &calls, 0, 0, 0); R.batch_call[g->t & 63] = calls; } // this global batch is complete on this rank once that call is
	// The exchange is left running: the next batch's stage A (other send buffer) proceeds beside it.  What the next batch may not do before
	// this one is through is ordered elsewhere: its exchange follows this one on the stream xs; a receive buffer is written again two batches
	// on, behind this barrier of the batch in between, which every rank reaches only after its bfcg_mg_process_ev has waited for THIS
	// batch's stage B (finalise_previous); the other ranks have read all_counts before they come here.  (Slab mode: stage A of batch t + 2 writes
	// the rank's own share into receive buffer [t & 1] again -- behind the same wait for this batch's stage B.)
	if (ok) { GHIP(hipEventRecord(R.ev_sent[sb], R.xs)); R.sent_pending[sb] = 1; }
	pthread_barrier_wait(&g->bar);
	return g->failed ? -1 : 0;
}

static void *rank_main(void *arg)
{
	bfcg_group_t *g = (bfcg_group_t *)((void **)arg)[0];
	const int i = (int)(intptr_t)((void **)arg)[1];
	free(arg);
	uint64_t seen = 0;
	for (;;) {
		pthread_mutex_lock(&g->mu);
		while (g->job == seen && !g->quit) pthread_cond_wait(&g->cv, &g->mu);
		if (g->quit) { pthread_mutex_unlock(&g->mu); return 0; }
		seen = g->job;
		pthread_mutex_unlock(&g->mu);
		g->r[i].rc = rank_batch(g, i);
		pthread_mutex_lock(&g->mu);
		if (++g->n_done == g->n_local) { g->done_job = seen; pthread_cond_broadcast(&g->cv); }
		pthread_mutex_unlock(&g->mu);
	}
}

// May the next global batch keep the host out of its loop (rank_batch: `lazy`)?  Slab mode, and every context able to take its stage B from rows
// on the device (checked when the group is created).
static int lazy_possible(bfcg_group_t *g)
{
	// (nothing here may depend on a rank's state: the processes of a multi-process group decide alike, each for itself.  g->slabs changes only by
	// decisions all ranks take together; a group that has failed walks through the protocol like the others and reports the failure in its row)
	return g->lazy_ok && g->slabs;
}

static int run_job(bfcg_group_t *g)
{
	g->lazy = lazy_possible(g);
	pthread_mutex_lock(&g->mu);
	g->n_done = 0; ++g->job;
	pthread_cond_broadcast(&g->cv);
	while (g->done_job != g->job) pthread_cond_wait(&g->cv, &g->mu);
	pthread_mutex_unlock(&g->mu);
	++g->t;
	if (g->failed) { bfcg_set_error(g->err); return -1; }
	return 0;
}

extern "C" int bfcg_group_unique_id(uint8_t uid[BFCG_UID_BYTES])
{
	ncclUniqueId id;
	if (ncclGetUniqueId(&id) != ncclSuccess) { bfcg_set_error("ncclGetUniqueId failed"); return -1; }
	memcpy(uid, &id, BFCG_UID_BYTES);
	return 0;
}

extern "C" void bfcg_group_destroy(bfcg_group_t *g)
{
	if (!g) return;
	pthread_mutex_lock(&g->mu); g->quit = 1; pthread_cond_broadcast(&g->cv); pthread_mutex_unlock(&g->mu);
	for (auto &R : g->r) if (R.th) pthread_join(R.th, 0);
	for (auto &R : g->r) { (void)hipSetDevice(R.device); if (R.xs) (void)hipStreamSynchronize(R.xs); } // every rank's exchange first: it writes into the others' buffers
	for (auto &R : g->r) {
		(void)hipSetDevice(R.device);
		if (R.ctx) (void)bfcg_sync(R.ctx);
		if (R.comm) { if (g->failed && g->mp) (void)ncclCommAbort(R.comm); else (void)ncclCommDestroy(R.comm); } // (peers of a failed run may never post what a clean destroy waits for)
		(void)hipFree(R.send2[0]); (void)hipFree(R.send2[1]); if (!R.combined) { (void)hipFree(R.recv[0]); (void)hipFree(R.recv[1]); } (void)hipFree(R.d_counts); (void)hipFree(R.d_moved); (void)hipFree(R.d_seq); (void)hipFree(R.d_qual);
		for (int b = 0; b < 2; ++b) { (void)hipFree(R.d_rows_out[b]); (void)hipFree(R.d_rows_in[b]); }
		if (R.ev_x) (void)hipEventDestroy(R.ev_x);
		for (int b = 0; b < 2; ++b) if (R.ev_sent[b]) (void)hipEventDestroy(R.ev_sent[b]);
		if (R.xs) (void)hipStreamDestroy(R.xs);
		if (R.cs) (void)hipStreamDestroy(R.cs);
		free(R.counts);
		if (R.ctx) bfcg_destroy(R.ctx);
	}
	if (g->all_counts) (void)hipHostFree(g->all_counts);
	pthread_barrier_destroy(&g->bar); pthread_mutex_destroy(&g->mu); pthread_cond_destroy(&g->cv);
	delete g;
}

extern "C" bfcg_group_t *bfcg_group_create(const bfcg_params_t *prm, int n_ranks, int first_rank, int n_local, const int *devices, const uint8_t *uid, int transport)
{
	if (n_ranks < 1 || n_local < 1 || first_rank < 0 || first_rank + n_local > n_ranks || (n_local < n_ranks && n_local != 1) || (n_local < n_ranks && !uid) || (uid && n_local != 1)) {
		bfcg_set_error("bfcg_group_create: the local ranks are either all n_ranks of the run, or one per process with the RCCL unique id of the run"); return NULL;
	}
	bfcg_group_t *g = new bfcg_group();
	g->prm = *prm; g->n_ranks = n_ranks; g->first = first_rank; g->n_local = n_local;
	g->job = g->done_job = 0; g->n_done = 0; g->quit = 0; g->failed = 0; g->t = 0; g->err[0] = 0; g->all_counts = 0;
	pthread_barrier_init(&g->bar, 0, (unsigned)n_local); pthread_mutex_init(&g->mu, 0); pthread_cond_init(&g->cv, 0);
	g->mp = uid != NULL;
	g->xp = transport ? transport : XP_RCCL;
	if (g->mp) g->xp = XP_RCCL;
	if (n_local == n_ranks && g->xp == XP_RCCL) // RCCL refuses two ranks on one device: repeated devices (emulation on a single GPU) take the peer-copy path
		for (int i = 0; i < n_local; ++i) for (int j = 0; j < i; ++j) if (devices[i] == devices[j]) g->xp = XP_PEER;
	// transport 3, or by default wherever ranks share a device (RCCL is out there anyway): the peer-copy transport with lazy batches PUSHED by k_push_slabs
	// (exact bytes).  Distinct devices keep RCCL unless 3 is asked for: the push kernel has never run over xGMI (no two-GPU box in six rounds).
	g->push = 0; g->push_wgs = 32; g->x_links = g->x_exact = 0;
	if (transport == XP_PUSH) { g->xp = XP_PEER; g->push = 1; }
	else if (g->xp == XP_PEER && transport == 0) { const char *e = getenv("BFCG_MG_PUSH"); g->push = !(e && atoi(e) == 0); }
	if (n_ranks > PUSH_MAX_RANKS) g->push = 0;
	{ const char *e = getenv("BFCG_MG_PUSH_WGS"); if (e && atoi(e) > 0) g->push_wgs = (unsigned)atoi(e); }
	if (g->xp == XP_PEER && n_local < n_ranks) { bfcg_set_error("peer copies need every rank in this process"); bfcg_group_destroy(g); return NULL; }
	g->r.resize((size_t)n_local);
	for (auto &R : g->r) memset(&R, 0, sizeof(R));
	for (int i = 0; i < n_local; ++i) {
		rank_t &R = g->r[i];
		R.rank = first_rank + i; R.device = devices[i];
		bfcg_params_t p = *prm;
		p.device = R.device; p.rank = R.rank; p.n_ranks = n_ranks;
		R.ctx = bfcg_create(&p);
		if (!R.ctx) { bfcg_group_destroy(g); return NULL; }
		bfcg_mg_allow_onepass(R.ctx, 1); // the receive buffers alternate (rank_batch): a stage B can be replayed from the one it read
	}
	{
		int info[4];
		bfcg_mg_info(g->r[0].ctx, info);
		g->nb1 = info[0]; g->nb_loc = info[1]; g->rec_bytes = info[2];
		g->kmer_limit = (uint64_t)((double)bfcg_batch_limit(g->r[0].ctx) / 0.95);
	}
	const uint64_t cap = prm->max_batch_pos, rcap = n_ranks > 1 ? cap + cap / 4 + (1u << 20) : cap; // = the contexts' level-2 capacity
	{ // slab mode: every context must offer it, and a record index that spans send AND receive buffer must fit 32 bits
		const char *e = getenv("BFCG_MG_SLABS");
		g->slabs_ok = !(e && atoi(e) == 0) && !prm->track_order;
		g->slab_cap = 0;
		for (auto &R : g->r) { uint32_t si[2]; bfcg_mg_slab_info(R.ctx, si); if (!si[1]) g->slabs_ok = 0; g->slab_cap = si[0]; }
		g->blk = (uint64_t)g->nb_loc * 8 * g->slab_cap;
		const uint64_t S = (uint64_t)g->nb1 * 8 * g->slab_cap;
		if (S + (rcap > S ? rcap : S) + 8192 >= 0xffffffffULL) g->slabs_ok = 0;
		g->slabs = g->slabs_ok;
		g->row_words = (size_t)g->nb1 * (g->slabs_ok ? 8 : 1) + 2;
		{ const char *e2 = getenv("BFCG_MG_LAZY"); g->lazy_ok = g->slabs_ok && !(e2 && atoi(e2) == 0); for (auto &R : g->r) if (!bfcg_mg_async_ok(R.ctx)) g->lazy_ok = 0; }
		g->lazy = 0; g->n_lazy = 0;
	}
	if (hipHostMalloc(&g->all_counts, sizeof(uint32_t) * (size_t)n_ranks * g->row_words, hipHostMallocDefault) != hipSuccess) { bfcg_set_error("hipHostMalloc failed"); bfcg_group_destroy(g); return NULL; }
	std::vector<ncclComm_t> comms((size_t)n_local, (ncclComm_t)0);
	if (g->xp == XP_RCCL && !g->mp && n_ranks > 1) {
		const ncclResult_t nr = ncclCommInitAll(comms.data(), n_local, devices);
		if (nr != ncclSuccess) {
			char msg[384];
			snprintf(msg, sizeof(msg), "ncclCommInitAll failed: %s (%s)", ncclGetErrorString(nr), ncclGetLastError(NULL));
			bfcg_set_error(msg); bfcg_group_destroy(g); return NULL;
		}
	}
	for (int i = 0; i < n_local; ++i) {
		rank_t &R = g->r[i];
		hipError_t e = hipSetDevice(R.device);
		if (e == hipSuccess) e = hipStreamCreateWithFlags(&R.xs, hipStreamNonBlocking);
		if (e == hipSuccess) e = hipEventCreateWithFlags(&R.ev_x, hipEventDisableTiming);
		if (e == hipSuccess) e = hipStreamCreateWithFlags(&R.cs, hipStreamNonBlocking);
		for (int b = 0; b < 2; ++b) if (e == hipSuccess) e = hipEventCreateWithFlags(&R.ev_sent[b], hipEventDisableTiming);
		R.send_cap = cap * (uint64_t)g->rec_bytes;
		R.recv_cap = rcap * (uint64_t)g->rec_bytes;
		if (g->slabs_ok) { // all slabs of a batch (+ a tile of slack: a run that finds its slab full is still stored), the receive buffer right behind
			const uint64_t S = (uint64_t)g->nb1 * 8 * g->slab_cap;
			R.send_cap = S * (uint64_t)g->rec_bytes;
			if (R.recv_cap < (S + 8192) * (uint64_t)g->rec_bytes) R.recv_cap = (S + 8192) * (uint64_t)g->rec_bytes;
			R.combined = 1;
			for (int b = 0; b < 2; ++b) { if (e == hipSuccess) e = hipMalloc(&R.send2[b], R.send_cap + R.recv_cap); if (e == hipSuccess) R.recv[b] = R.send2[b] + R.send_cap; }
		} else {
			for (int b = 0; b < 2; ++b) if (e == hipSuccess) e = hipMalloc(&R.send2[b], R.send_cap);
			if (e == hipSuccess) e = hipMalloc(&R.recv[0], R.recv_cap);
			if (e == hipSuccess) e = hipMalloc(&R.recv[1], R.recv_cap);
		}
		if (e == hipSuccess) e = hipMalloc(&R.d_counts, sizeof(uint32_t) * (size_t)n_ranks * g->row_words);
		if (g->lazy_ok) for (int b = 0; b < 2; ++b) {
			const size_t w = (size_t)n_ranks * (size_t)bfcg_mg_row_words(R.ctx);
			if (e == hipSuccess) e = hipMalloc(&R.d_rows_out[b], sizeof(uint32_t) * w);
			if (e == hipSuccess) e = hipMalloc(&R.d_rows_in[b], sizeof(uint32_t) * w);
			if (e == hipSuccess) e = hipMemset(R.d_rows_in[b], 0, sizeof(uint32_t) * w);
		}
		if (g->push) { if (e == hipSuccess) e = hipMalloc(&R.d_moved, sizeof(unsigned long long)); if (e == hipSuccess) e = hipMemset(R.d_moved, 0, sizeof(unsigned long long)); }
		R.in_cap = cap;
		if (e == hipSuccess) e = hipMalloc(&R.d_seq, cap);
		if (e == hipSuccess) e = hipMalloc(&R.d_qual, cap);
		R.counts = (uint32_t *)calloc(g->row_words, sizeof(uint32_t));
		if (e != hipSuccess) { bfcg_set_error(hipGetErrorString(e)); bfcg_group_destroy(g); return NULL; }
		if (g->xp == XP_PEER) for (int j = 0; j < n_local; ++j) if (devices[j] != R.device) (void)hipDeviceEnablePeerAccess(devices[j], 0);
		if (g->xp == XP_RCCL && (n_ranks > 1 || g->mp)) {
			if (!g->mp) R.comm = comms[i];
			else {
				ncclUniqueId id; memcpy(&id, uid, BFCG_UID_BYTES);
				const ncclResult_t nr = ncclCommInitRank(&R.comm, n_ranks, id, R.rank);
				if (nr != ncclSuccess) { // (say why: two ranks of one communicator on the same device are refused, for one -- scripts/probes/rccl_same_device.py)
					char msg[384];
					snprintf(msg, sizeof(msg), "ncclCommInitRank failed: %s (%s)", ncclGetErrorString(nr), ncclGetLastError(NULL));
					R.comm = 0; bfcg_set_error(msg); bfcg_group_destroy(g); return NULL;
				}
			}
		}
	}
	(void)hipGetLastError(); // hipDeviceEnablePeerAccess on an already enabled pair
	if (g->mp) { // one rank per process: every process must have decided the same exchange layout (slab mode, slab capacity, words per row of sizes) --
	             // each decides it from its own environment and context; an all-gather with different counts would hang or corrupt memory
		rank_t &R = g->r[0];
		const uint32_t mine[4] = {(uint32_t)(g->slabs_ok | (g->lazy_ok << 1)), g->slab_cap, (uint32_t)g->row_words, (uint32_t)g->rec_bytes}; // (bit 1: the sizes stay on the device -- BFCG_MG_LAZY)
		std::vector<uint32_t> all((size_t)4 * n_ranks, 0);
		hipError_t he = hipSetDevice(R.device);
		ncclResult_t ne = ncclSuccess;
		if (he == hipSuccess) he = hipMemcpyAsync(R.d_counts + (size_t)4 * R.rank, mine, sizeof(mine), hipMemcpyHostToDevice, R.xs); // (d_counts holds n_ranks x row_words >= 4 n_ranks words)
		if (he == hipSuccess) ne = ncclAllGather(R.d_counts + (size_t)4 * R.rank, R.d_counts, 4, ncclUint32, R.comm, R.xs);
		if (he == hipSuccess && ne == ncclSuccess) he = hipMemcpyAsync(all.data(), R.d_counts, sizeof(uint32_t) * all.size(), hipMemcpyDeviceToHost, R.xs);
		if (he == hipSuccess && ne == ncclSuccess) he = hipStreamSynchronize(R.xs);
		if (he != hipSuccess || ne != ncclSuccess) { bfcg_set_error("all-gather of the ranks' exchange layout failed"); g->failed = 1; bfcg_group_destroy(g); return NULL; }
		for (int p = 0; p < n_ranks; ++p)
			if (memcmp(&all[(size_t)4 * p], mine, sizeof(mine)) != 0) {
				char msg[256];
				snprintf(msg, sizeof(msg), "rank %d decided another exchange layout (slab mode %u / %u, slab capacity %u / %u, row words %u / %u): same BFCG_MG_SLABS and parameters on every process?",
				         p, all[(size_t)4 * p], mine[0], all[(size_t)4 * p + 1], mine[1], all[(size_t)4 * p + 2], mine[2]);
				bfcg_set_error(msg); g->failed = 1; bfcg_group_destroy(g); return NULL;
			}
	}
	for (int i = 0; i < n_local; ++i) {
		void **arg = (void **)malloc(2 * sizeof(void *));
		arg[0] = g; arg[1] = (void *)(intptr_t)i;
		if (pthread_create(&g->r[i].th, 0, rank_main, arg) != 0) { g->r[i].th = 0; free(arg); bfcg_set_error("pthread_create failed"); bfcg_group_destroy(g); return NULL; }
	}
	return g;
}

// out[0] bytes the local ranks have put on the links since creation / the last reset (whole blocks and messages as sent; the PUSH kernel's own count of
// what it wrote), out[1] the bytes of the LIVE records among them (what an exact exchange moves), out[2] global batches, out[3] transport (1 RCCL, 2 peer
// copies, 3 peer copies + push kernel).  Drains the exchange streams.
extern "C" int bfcg_group_exchange_bytes(bfcg_group_t *g, uint64_t out[4])
{
	unsigned long long pushed = 0;
	for (auto &R : g->r) if (R.d_moved) {
		unsigned long long v = 0;
		if (hipSetDevice(R.device) != hipSuccess || hipStreamSynchronize(R.xs) != hipSuccess || hipMemcpy(&v, R.d_moved, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) { bfcg_set_error("reading the push counter failed"); return -1; }
		pushed += v;
	}
	out[0] = __atomic_load_n(&g->x_links, __ATOMIC_RELAXED) + pushed; out[1] = __atomic_load_n(&g->x_exact, __ATOMIC_RELAXED); out[2] = g->t; out[3] = g->push ? XP_PUSH : g->xp;
	return 0;
}
extern "C" int bfcg_group_info(bfcg_group_t *g, int out[6])
{
	out[0] = g->n_ranks; out[1] = g->n_local; out[2] = g->push ? XP_PUSH : g->xp; out[3] = g->rec_bytes; out[4] = g->nb1; out[5] = g->first;
	return 0;
}
extern "C" int bfcg_group_slab_mode(bfcg_group_t *g) { return g->slabs; }
extern "C" uint64_t bfcg_group_lazy_batches(bfcg_group_t *g) { return g->n_lazy; }
extern "C" bfcg_ctx_t *bfcg_group_ctx(bfcg_group_t *g, int i) { return i >= 0 && i < g->n_local ? g->r[i].ctx : NULL; }

static int drain_exchange(bfcg_group_t *g)
{
	for (auto &R : g->r) {
		if (hipSetDevice(R.device) != hipSuccess || hipStreamSynchronize(R.xs) != hipSuccess) { bfcg_set_error("exchange stream failed"); return -1; }
		R.sent_pending[0] = R.sent_pending[1] = 0;
	}
	return 0;
}
extern "C" int bfcg_group_reset(bfcg_group_t *g)
{
	if (drain_exchange(g) != 0) return -1;
	for (auto &R : g->r) if (bfcg_reset(R.ctx) != 0) return -1;
	// the contexts count their calls from zero again: so do the global batches (bfcg_group_progress compares the two), and a run that fell back
	// to the two-pass stage A starts the next data set in slab mode again
	g->t = 0;
	for (auto &R : g->r) memset(R.batch_call, 0, sizeof(R.batch_call));
	g->slabs = g->slabs_ok;
	g->x_links = g->x_exact = 0;
	for (auto &R : g->r) if (R.d_moved && (hipSetDevice(R.device) != hipSuccess || hipMemset(R.d_moved, 0, sizeof(unsigned long long)) != hipSuccess)) { bfcg_set_error("clearing the push counter failed"); return -1; }
	return 0;
}
// one rank per process: every process learns whether any of them has failed (one word per rank, all-gathered; the rank threads are idle)
static int mp_agree(bfcg_group_t *g, int local_rc)
{
	if (!g->mp) return local_rc;
	rank_t &R = g->r[0];
	uint32_t w = (local_rc != 0 || g->failed) ? 1u : 0u;
	std::vector<uint32_t> all((size_t)g->n_ranks, 0);
	hipError_t he = hipSetDevice(R.device);
	ncclResult_t ne = ncclSuccess;
	// (d_counts is free between batches; in place: word `rank` is this rank's contribution)
	if (he == hipSuccess) he = hipMemcpyAsync(R.d_counts + R.rank, &w, sizeof(w), hipMemcpyHostToDevice, R.xs);
	if (he == hipSuccess) ne = ncclAllGather(R.d_counts + R.rank, R.d_counts, 1, ncclUint32, R.comm, R.xs);
	if (he == hipSuccess && ne == ncclSuccess) he = hipMemcpyAsync(all.data(), R.d_counts, sizeof(uint32_t) * (size_t)g->n_ranks, hipMemcpyDeviceToHost, R.xs);
	if (he == hipSuccess && ne == ncclSuccess) he = hipStreamSynchronize(R.xs);
	if (he != hipSuccess || ne != ncclSuccess) { bfcg_set_error("all-gather of the ranks' status failed"); return -1; }
	for (int p = 0; p < g->n_ranks; ++p) if (all[(size_t)p] && local_rc == 0 && !g->failed) { grp_err(g, "rank %d of the run has failed", p); bfcg_set_error(g->err); return -1; }
	return local_rc != 0 || g->failed ? -1 : 0;
}
extern "C" int bfcg_group_sync(bfcg_group_t *g)
{
	int rc = 0;
	if (drain_exchange(g) != 0) rc = -1;
	for (auto &R : g->r) if (rc == 0 && bfcg_sync(R.ctx) != 0) rc = -1;
	return mp_agree(g, rc);
}

// Progress without draining (as bfcg_progress): batches = global batches submitted so far, final = the last one that is complete on every local
// rank, keys_of[j], j < n: distinct keys (summed over the local ranks) after global batch final - j.  Returns how many entries of keys_of are
// valid.  bfc_count prints the lines of count.c:110-114 from this.
extern "C" int bfcg_group_progress(bfcg_group_t *g, uint64_t *batches, uint64_t *final, uint64_t *keys_of, int n)
{
	const int NL = g->n_local;
	std::vector<uint64_t> fin((size_t)NL, 0), keys((size_t)NL * 63, 0);
	for (int i = 0; i < NL; ++i) bfcg_progress(g->r[i].ctx, 0, &fin[(size_t)i], &keys[(size_t)i * 63], 63);
	const uint64_t lo = g->t > 63 ? g->t - 63 : 0; // batch_call[] remembers the last 64 batches (a rank runs at most two batches ahead of its stage B)
	uint64_t T = g->t;                               // batches are numbered from 1: batch T sits at index (T - 1) & 63
	for (; T > lo; --T) {
		bool all = true;
		for (int i = 0; i < NL && all; ++i) all = g->r[i].batch_call[(T - 1) & 63] <= fin[(size_t)i];
		if (all) break;
	}
	if (batches) *batches = g->t;
	if (final) *final = T;
	int j = 0;
	for (; j < n && T - (uint64_t)j > lo; ++j) {
		uint64_t sum = 0;
		bool have = true;
		for (int i = 0; i < NL && have; ++i) {
			const uint64_t call = g->r[i].batch_call[(T - (uint64_t)j - 1) & 63], back = fin[(size_t)i] - call;
			have = call >= 1 && back < 63 && back < fin[(size_t)i];
			if (have) sum += keys[(size_t)i * 63 + back];
		}
		if (!have) break;
		keys_of[j] = sum;
	}
	return j;
}

static inline int is_acgt(uint8_t ch) { ch &= 0xDF; return ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T'; }

// A group of ONE rank counts a batch far larger than the filter's regions take at full speed in ROUNDS, as a single GPU cuts an oversized call
// into sub-batches (a region handles only so many k-mers with clear bits per pass -- list_cap -- before it falls onto the exact but slow HBM
// path).  With several ranks the owner splits what it received by SOURCES instead (process_in_groups): the file order of a global batch is
// rank-major, so pieces of every rank's share taken round by round would not be in file order -- consecutive sources are.
static int rounds_for(bfcg_group_t *g, uint64_t local_pos, int *rounds)
{
	*rounds = 1;
	if (g->n_ranks > 1) return 0;
	uint64_t total = local_pos;
	if (g->mp) {
		rank_t &R = g->r[0];
		std::vector<unsigned long long> all((size_t)g->n_ranks, 0);
		unsigned long long mine = local_pos;
		unsigned long long *d = reinterpret_cast<unsigned long long *>(R.d_counts); // (free between batches; N * (nb1 + 1) words >= 2 N)
		hipError_t he = hipSetDevice(R.device);
		ncclResult_t ne = ncclSuccess;
		if (he == hipSuccess) he = hipMemcpyAsync(d + R.rank, &mine, sizeof(mine), hipMemcpyHostToDevice, R.xs);
		if (he == hipSuccess) ne = ncclAllGather(d + R.rank, d, 1, ncclUint64, R.comm, R.xs);
		if (he == hipSuccess && ne == ncclSuccess) he = hipMemcpyAsync(all.data(), d, sizeof(unsigned long long) * (size_t)g->n_ranks, hipMemcpyDeviceToHost, R.xs);
		if (he == hipSuccess && ne == ncclSuccess) he = hipStreamSynchronize(R.xs);
		if (he != hipSuccess || ne != ncclSuccess) { bfcg_set_error("all-gather of the ranks' batch sizes failed"); return -1; }
		total = 0;
		for (auto v : all) total += v;
	}
	double warm = 1e30;
	for (auto &R : g->r) { const double w = bfcg_mg_warm_factor(R.ctx); if (w < warm) warm = w; }
	const double per_rank = 0.8 * (double)total / (double)g->n_ranks, limit = (double)g->kmer_limit * warm * 0.9;
	int r = per_rank <= limit * 7.0 / 6.0 ? 1 : (int)(per_rank / limit) + 1; // (just above the limit a few slow regions cost less than another pass over filter and table)
	if (getenv("BFCG_MG_ROUNDS")) r = atoi(getenv("BFCG_MG_ROUNDS"));
	*rounds = r < 1 ? 1 : r > 64 ? 64 : r;
	return 0;
}

// first position behind a byte that ends every k-mer (not ACGT), at or behind `at`, in a device-resident stream of n bytes
static uint64_t dev_cut(rank_t &R, const uint8_t *d_seq, uint64_t at, uint64_t n)
{
	uint8_t buf[4096];
	while (at < n) {
		const uint64_t w = n - at < sizeof(buf) ? n - at : sizeof(buf);
		if (hipSetDevice(R.device) != hipSuccess || hipMemcpyAsync(buf, d_seq + at, w, hipMemcpyDeviceToHost, R.cs) != hipSuccess || hipStreamSynchronize(R.cs) != hipSuccess) return n;
		for (uint64_t i = 0; i < w; ++i) if (!is_acgt(buf[i])) return at + i + 1;
		at += w;
	}
	return n;
}

extern "C" int bfcg_group_count_batch_dev(bfcg_group_t *g, const uint8_t *const *d_seq, const uint8_t *const *d_qual, const uint64_t *n_pos)
{
	const int NL = g->n_local;
	uint64_t local = 0;
	for (int i = 0; i < NL; ++i) local += n_pos[i];
	int rounds = 1;
	if (rounds_for(g, local, &rounds) != 0) return -1;
	std::vector<uint64_t> cut((size_t)NL * (size_t)(rounds + 1), 0);
	for (int i = 0; i < NL; ++i) { // every rank's share in `rounds` pieces, cut where a read ends
		uint64_t *c = &cut[(size_t)i * (size_t)(rounds + 1)];
		c[rounds] = n_pos[i];
		for (int r = 1; r < rounds; ++r) { c[r] = n_pos[i] ? dev_cut(g->r[i], d_seq[i], n_pos[i] / (uint64_t)rounds * (uint64_t)r, n_pos[i]) : 0; if (c[r] < c[r - 1]) c[r] = c[r - 1]; }
	}
	for (int r = 0; r < rounds; ++r) {
		for (int i = 0; i < NL; ++i) {
			const uint64_t *c = &cut[(size_t)i * (size_t)(rounds + 1)];
			g->r[i].in_seq = d_seq[i] + c[r]; g->r[i].in_qual = d_qual && d_qual[i] ? d_qual[i] + c[r] : 0; g->r[i].in_pos = c[r + 1] - c[r]; g->r[i].in_host = 0;
		}
		if (run_job(g) != 0) return -1;
	}
	return 0;
}

// One global batch from host memory (all ranks local): the stream is cut into n_ranks contiguous shares at separator bytes (no k-mer spans
// one) -- rank r takes the r-th share, which is the rank-major file order the exchange assumes.
static int group_host_piece(bfcg_group_t *g, const uint8_t *h_seq, const uint8_t *h_qual, uint64_t n_pos)
{
	const int N = g->n_ranks;
	uint64_t o = 0;
	for (int i = 0; i < N; ++i) {
		uint64_t e = i + 1 == N ? n_pos : n_pos * (uint64_t)(i + 1) / (uint64_t)N;
		if (e < o) e = o;
		while (e < n_pos && e > o && is_acgt(h_seq[e - 1])) ++e; // forward to just behind the next byte that ends a k-mer
		if (i + 1 == N) e = n_pos;
		g->r[i].in_seq = h_seq + o; g->r[i].in_qual = h_qual ? h_qual + o : 0; g->r[i].in_pos = e - o; g->r[i].in_host = 1;
		o = e;
	}
	return run_job(g);
}
extern "C" int bfcg_group_count_batch_host(bfcg_group_t *g, const uint8_t *h_seq, const uint8_t *h_qual, uint64_t n_pos)
{
	if (g->n_local != g->n_ranks) { bfcg_set_error("bfcg_group_count_batch_host needs every rank in this process"); return -1; }
	int rounds = 1; // an oversized batch: in rounds (rounds_for), each a piece of the stream cut where a read ends
	if (rounds_for(g, n_pos, &rounds) != 0) return -1;
	uint64_t a = 0;
	for (int r = 0; r < rounds && a < n_pos; ++r) {
		uint64_t e = r + 1 == rounds ? n_pos : n_pos / (uint64_t)rounds * (uint64_t)(r + 1);
		if (e < a) e = a;
		while (e < n_pos && e > a && is_acgt(h_seq[e - 1])) ++e;
		if (e > a && group_host_piece(g, h_seq + a, h_qual ? h_qual + a : 0, e - a) != 0) return -1;
		a = e;
	}
	return 0;
}

// sums over the local ranks (a multi-process caller adds the processes' sums up); table geometry of rank `first`
extern "C" int bfcg_group_stats(bfcg_group_t *g, uint64_t out[BFCG_ST_N])
{
	if (drain_exchange(g) != 0) return -1;
	memset(out, 0, sizeof(uint64_t) * BFCG_ST_N);
	for (auto &R : g->r) {
		uint64_t st[BFCG_ST_N];
		if (bfcg_stats(R.ctx, st) != 0) return -1;
		for (int i = 0; i < BFCG_ST_N; ++i) if (i != BFCG_ST_TAB_CSHIFT && i != BFCG_ST_BATCHES) out[i] += st[i];
		out[BFCG_ST_TAB_CSHIFT] = st[BFCG_ST_TAB_CSHIFT]; out[BFCG_ST_BATCHES] = st[BFCG_ST_BATCHES];
	}
	return 0;
}

// THE count table of the run (all ranks local): the union of the ranks' disjoint tables
extern "C" bfc_ch_t *bfcg_group_export_table(bfcg_group_t *g)
{
	if (drain_exchange(g) != 0) return NULL;
	if (g->n_local != g->n_ranks) { bfcg_set_error("bfcg_group_export_table needs every rank in this process (else: export per rank, bfc_ch_union)"); return NULL; }
	std::vector<bfc_ch_t *> t;
	bfc_ch_t *u = 0;
	char why[384] = "";
	for (auto &R : g->r) {
		bfc_ch_t *x = bfcg_export_table(R.ctx);
		if (!x) { snprintf(why, sizeof(why), "rank %d's table could not be exported: %s", R.rank, bfcg_last_error()); break; } // (say which step failed: the union below used to hide it)
		t.push_back(x);
	}
	if ((int)t.size() == g->n_local) u = g->n_local == 1 ? t[0] : bfc_ch_union((const bfc_ch_t *const *)t.data(), g->n_local);
	if (!u) bfcg_set_error(why[0] ? why : "union of the ranks' tables failed (tables of different geometry, or no host memory for the union)");
	if (g->n_local > 1 || !u) for (auto x : t) bfc_ch_destroy(x);
	return u;
}

// THE bloom filter of the run (all ranks local): rank r owns the r-th 1/n_ranks of the bitmap
extern "C" bfc_bf_t *bfcg_group_export_bloom(bfcg_group_t *g, int which)
{
	if (drain_exchange(g) != 0) return NULL;
	if (g->n_local != g->n_ranks) { bfcg_set_error("bfcg_group_export_bloom needs every rank in this process"); return NULL; }
	bfc_bf_t *b = bfc_bf_alloc_raw(g->prm.bf_shift, g->prm.n_hashes);
	if (!b) { bfcg_set_error("host allocation of the bloom filter failed"); return NULL; }
	const uint64_t slice = (1ULL << (g->prm.bf_shift - 3)) / (uint64_t)g->n_ranks;
	for (int i = 0; i < g->n_local; ++i)
		if (bfcg_bloom_to_host(g->r[i].ctx, which, b->b + (uint64_t)i * slice) != 0) { bfc_bf_destroy(b); return NULL; }
	return b;
}

// The same, and every local device keeps a FULL copy of the filter in HBM behind the returned host object, all-gathered from the ranks'
// slices by peer copies over xGMI: `bfc -1` queries bf_high for every k-mer of every read next (correct.c:556), and the trim pass shards
// the reads over the same devices (bfc_trim.c) -- each of its contexts adopts the copy on its device instead of uploading 2^(b-3) bytes.
extern "C" bfc_bf_t *bfcg_group_export_bloom_resident(bfcg_group_t *g, int which)
{
	if (drain_exchange(g) != 0) return NULL;
	bfc_bf_t *b = bfcg_group_export_bloom(g, which);
	if (!b) return NULL;
	const uint64_t full = 1ULL << (g->prm.bf_shift - 3);
	for (int i = 0; i < g->n_local; ++i) {
		const int dev = g->r[i].device;
		int dup = 0;
		for (int j = 0; j < i; ++j) if (g->r[j].device == dev) dup = 1;
		if (dup) continue;
		void *d = 0;
		if (hipSetDevice(dev) != hipSuccess || hipMalloc(&d, full) != hipSuccess) { (void)hipGetLastError(); continue; } // no room here: the host copy is a complete answer
		hipError_t e = hipSuccess;
		for (int j = 0; j < g->n_local && e == hipSuccess; ++j) {
			uint64_t bytes = 0;
			void *src = bfcg_bloom_slice(g->r[j].ctx, which, &bytes);
			e = hipMemcpyPeerAsync((char *)d + (uint64_t)j * bytes, dev, src, g->r[j].device, bytes, g->r[i].xs);
		}
		if (e == hipSuccess) e = hipStreamSynchronize(g->r[i].xs);
		if (e != hipSuccess || bfcg_resident_register(b, d, dev, g->prm.bf_shift) != 0) { (void)hipGetLastError(); (void)hipFree(d); }
	}
	return b;
}
