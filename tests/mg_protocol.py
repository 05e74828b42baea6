"""TEST INFRASTRUCTURE (not part of the product: libbfc_gpu.so exchanges its records itself, bfc_amd/csrc/bfcg_mg.hip).
A Python restatement of the multi-GPU PROTOCOL -- the owner-computes partition of the k-mer counting path -- over
torch.distributed, so that the protocol (who sends what to whom, in which order the owner applies it) can be pinned with a real all-to-all
on CPUs (gloo, world size 2, an oracle-based engine: tests/test_multi_rank.py) and with ranks emulated on one device (LocalCluster).

One process per GPU.  Rank r owns 1/N of the bloom regions (contiguous level-1 buckets) and every k-mer that falls
into them, hence also a disjoint part of the count table.  Per global batch:

    stage A (every rank)   bases -> records grouped by level-1 bucket          bfcg_mg_scatter
    exchange               all-to-all of bucket sizes, then of records         torch.distributed (RCCL on GPUs, gloo in CPU tests)
    stage B (the owner)    level 2 -> bloom regions -> table                    bfcg_mg_process

File order across ranks is rank-major inside a batch (rank 0's share, then rank 1's, ...): stage A prefixes the rank to
every record's in-batch position, which is what the bloom kernel orders first setters by -- so results are those of
`bfc -t1` on the batches concatenated in that order.

The exchange code below is backend-agnostic: an *engine* supplies `scatter(...) -> counts` / `process(seg_cnt)` and
owns `send` / `recv` tensors.  `GpuEngine` drives libbfc_gpu.so; tests/ supply a CPU engine built on the oracle to pin
the protocol under gloo with world_size 2.
"""
import numpy as np

CHUNK = 1 << 26  # elements (int32 on the GPU path: 256 MiB) per point-to-point message


def exchange(engine, counts, group=None):
    """All-to-all of one batch. counts: uint32[nb1] sizes of this rank's level-1 buckets (records in engine.send,
    grouped by bucket). Returns seg_cnt uint32[world][nb_loc] for the owned buckets; engine.recv holds the records
    source-major."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    nb1 = len(counts)
    if nb1 % world:
        raise ValueError("level-1 buckets (%d) not divisible by world size %d" % (nb1, world))
    nb_loc = nb1 // world
    rw = engine.rec_words
    dev = engine.send.device
    mine = torch.from_numpy(np.ascontiguousarray(counts, dtype=np.int64)).view(world, nb_loc).to(dev)  # [destination][bucket]
    theirs = torch.empty_like(mine)                                                                       # [source][bucket]
    dist.all_to_all_single(theirs, mine, group=group)
    theirs_h = theirs.cpu()
    in_splits = (mine.cpu().sum(1) * rw).tolist()
    out_splits = (theirs_h.sum(1) * rw).tolist()
    n_out = int(sum(out_splits))
    if n_out > engine.recv.numel():
        raise RuntimeError("rank receives %d words, receive buffer holds %d" % (n_out, engine.recv.numel()))
    # Records: grouped point-to-point sends/receives, at most CHUNK elements per message, own block by a local copy.
    # (all_to_all_single is not used here: this image's RCCL delivers only the first half of a message larger than 1 GiB --
    # measured with a single-rank all_to_all of 1.1 GB -- and chunked P2P is also what lets every xGMI link work at once.)
    me = dist.get_rank(group)
    in_off = np.concatenate([[0], np.cumsum(in_splits)]).astype(np.int64)
    out_off = np.concatenate([[0], np.cumsum(out_splits)]).astype(np.int64)
    if in_splits[me]:
        engine.recv[int(out_off[me]):int(out_off[me]) + int(in_splits[me])].copy_(engine.send[int(in_off[me]):int(in_off[me]) + int(in_splits[me])])
    ops = []
    for step in range(1, world):  # ring-shifted peer order: every rank talks to a different peer at any time
        to, frm = (me + step) % world, (me - step) % world
        n_to, n_from = int(in_splits[to]), int(out_splits[frm])
        for c0 in range(0, max(n_to, n_from), CHUNK):
            if c0 < n_to:
                ops.append(dist.P2POp(dist.isend, engine.send[int(in_off[to]) + c0:int(in_off[to]) + min(n_to, c0 + CHUNK)], to, group))
            if c0 < n_from:
                ops.append(dist.P2POp(dist.irecv, engine.recv[int(out_off[frm]) + c0:int(out_off[frm]) + min(n_from, c0 + CHUNK)], frm, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return theirs_h.numpy().astype(np.uint32)


def process_in_groups(ctx, recv_ptr, seg_cnt, rec_bytes, kmer_limit):
    """Stage B of one global batch on the owner.  When the rank received more k-mers than its bloom regions take at full speed
    (bfcg_batch_limit) the sources are processed in consecutive groups, each its own stage B: the receive buffer is source-major and the
    global order is rank-major, so a group's records all precede the next group's -- results are those of one big batch, without every
    region overflowing its LDS list.  Not with order stamps (track_order): the batch number is part of a stamp and must agree across ranks.
    Returns the number of stage-B launches."""
    seg_cnt = np.ascontiguousarray(seg_cnt, dtype=np.uint32)
    per_src = seg_cnt.astype(np.int64).sum(1)
    if not kmer_limit or int(per_src.sum()) <= kmer_limit or len(per_src) == 1:
        ctx.mg_process(recv_ptr, seg_cnt)
        return 1
    starts = np.concatenate([[0], np.cumsum(per_src)])
    groups, g0, acc = [], 0, 0
    for s_, n in enumerate(per_src):
        if s_ > g0 and acc + int(n) > kmer_limit:
            groups.append((g0, s_)); g0, acc = s_, 0
        acc += int(n)
    groups.append((g0, len(per_src)))
    done = 0
    for g0, g1 in groups:
        if starts[g1] == starts[g0]:
            continue
        seg = np.zeros_like(seg_cnt); seg[g0:g1] = seg_cnt[g0:g1]
        if done:  # stage A and stage B come in pairs (buffer sets, timing events): an empty stage A opens the next pair
            ctx.mg_scatter(None, None, 0, None)
        ctx.mg_process(recv_ptr + int(starts[g0]) * rec_bytes, seg)
        done += 1
    return done


class GpuEngine:
    """Stage A / stage B on libbfc_gpu.so with torch-owned exchange buffers (torch is plumbing: device memory + RCCL)."""

    def __init__(self, counter):
        import torch
        self.g = counter
        info = counter.mg_info()
        self.rec_words = info["rec_bytes"] // 4  # exchange buffers are int32 tensors
        cap = int(counter.params.max_batch_pos)
        dev = torch.device("cuda", counter.params.device)
        self.send = torch.empty(cap * self.rec_words, dtype=torch.int32, device=dev)
        # two receive buffers: stage B of batch t is left running on buffer t % 2 while batch t + 1 is scattered and exchanged
        self._recv = [torch.empty((cap + cap // 4 + (1 << 20)) * self.rec_words, dtype=torch.int32, device=dev) for _ in range(2)]
        self._cur = 0
        self.track = bool(getattr(counter.params, "track_order", 0))
        self.kmer_limit = int(counter.batch_limit() / 0.95)  # k-mers of a global batch this rank's regions take without the slow path

    @property
    def recv(self):
        return self._recv[self._cur]

    def scatter(self, d_seq, d_qual, n_pos):
        return self.g.mg_scatter(d_seq, d_qual, n_pos, self.send.data_ptr())  # synchronises the library's stream

    def process(self, seg_cnt):
        import torch
        torch.cuda.current_stream(self.send.device).synchronize()  # the exchange ran on torch's stream; stage B of the previous batch keeps running
        # returns once the previous stage B is finalised: its receive buffer is free again
        process_in_groups(self.g, self.recv.data_ptr(), seg_cnt, self.rec_words * 4, None if self.track else self.kmer_limit)
        self._cur ^= 1


def count_batch(engine, d_seq, d_qual, n_pos, group=None):
    """One global batch on this rank: stage A, exchange, stage B.  Stage B is left running (bfcg_mg_process): the next call's
    stage A and exchange overlap with it; engine.g.sync() / stats() / exports drain the pipeline."""
    counts = engine.scatter(d_seq, d_qual, n_pos)
    seg_cnt = exchange(engine, counts, group)
    lim = getattr(engine, "kmer_limit", None)
    if lim and getattr(engine, "track", False) and int(seg_cnt.sum()) > lim and not getattr(engine, "_warned", False):
        import warnings
        engine._warned = True
        warnings.warn("this rank received %d k-mers of one global batch, its bloom regions take about %d at full speed: "
                      "use smaller shares or a larger filter (bfcg_batch_limit)" % (int(seg_cnt.sum()), lim))
    engine.process(seg_cnt)
    return seg_cnt


class LocalCluster:
    """N ranks emulated on ONE device by N contexts and a host-mediated exchange: exercises stage A / stage B, the
    segment bookkeeping and the rank-major order on real kernels where only one GPU is available (tests)."""

    def __init__(self, gpu_lib, n_ranks, k, bf_shift, max_batch_pos, kmer_limit=None, **kw):
        self.n = n_ranks
        self.kmer_limit = kmer_limit  # None: from the contexts (bfcg_batch_limit); tests force small values to exercise the source groups
        self.track = bool(kw.get("track_order"))
        self.ctx = [gpu_lib.GpuCounter(k, bf_shift, max_batch_pos=max_batch_pos, rank=r, n_ranks=n_ranks, **kw) for r in range(n_ranks)]
        info = self.ctx[0].mg_info()
        self.rw, self.nb1, self.nb_loc = info["rec_bytes"] // 4, info["nb1"], info["nb_loc"]
        self.cap = max_batch_pos
        self.d_send = [c.dev_alloc(self.cap * self.rw * 4) for c in self.ctx]
        self.d_recv = [[c.dev_alloc((self.cap * 2 + 4096) * self.rw * 4) for _ in range(2)] for c in self.ctx]  # alternate: stage B runs asynchronously
        self.t = 0

    def batch(self, shares):
        """shares[r] = (seq_stream, qual_stream or None) of rank r for this global batch."""
        import ctypes as C
        sends, counts = [], []
        for r, (s, q) in enumerate(shares):
            c = self.ctx[r]
            s = np.ascontiguousarray(s, dtype=np.uint8)
            d_s = c.dev_alloc(max(len(s), 16)); c.h2d(d_s, s)
            d_q = None
            if q is not None:
                q = np.ascontiguousarray(q, dtype=np.uint8)
                d_q = c.dev_alloc(max(len(q), 16)); c.h2d(d_q, q)
            cnt = c.mg_scatter(d_s, d_q, len(s), self.d_send[r])
            host = np.empty(int(cnt.sum()) * self.rw, dtype=np.uint32)
            if len(host):
                c._ck(c.L.bfcg_d2h(c.ctx, host.ctypes.data, self.d_send[r], host.nbytes))
            sends.append(host); counts.append(cnt)
            c.dev_free(d_s)
            if d_q:
                c.dev_free(d_q)
        for o in range(self.n):  # owner o receives, source-major, its bucket range from every source
            parts, seg = [], np.zeros((self.n, self.nb_loc), dtype=np.uint32)
            for s_ in range(self.n):
                starts = np.concatenate([[0], np.cumsum(counts[s_].astype(np.int64))])
                lo, hi = starts[o * self.nb_loc], starts[(o + 1) * self.nb_loc]
                parts.append(sends[s_][lo * self.rw:hi * self.rw])
                seg[s_] = counts[s_][o * self.nb_loc:(o + 1) * self.nb_loc]
            recv = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint32)
            buf = self.d_recv[o][self.t & 1]
            if len(recv):
                self.ctx[o].h2d(buf, recv)
            lim = self.kmer_limit if self.kmer_limit is not None else int(self.ctx[o].batch_limit() / 0.95)
            self.launches = getattr(self, "launches", 0) + process_in_groups(self.ctx[o], buf, seg, self.rw * 4, None if self.track else lim)
        self.t += 1

    def bloom_bytes(self, which=0):
        return np.concatenate([c.bloom_bytes(which) for c in self.ctx])

    def stats(self):
        out = {}
        for c in self.ctx:
            for k_, v in c.stats().items():
                if isinstance(v, int):
                    out[k_] = out.get(k_, 0) + v
        return out

    def export_sorted(self):
        """Union of the per-rank tables (disjoint key sets) in L1 form."""
        parts = [c.export_table().export_sorted() for c in self.ctx]
        sizes = sum(p[0].astype(np.int64) for p in parts)
        n_sub = len(sizes)
        starts = [np.concatenate([[0], np.cumsum(p[0].astype(np.int64))]) for p in parts]
        out = np.empty(int(sizes.sum()), dtype=np.uint64)
        pos = 0
        nz = np.nonzero(sizes)[0]
        for sub in nz:
            chunk = np.concatenate([p[1][st[sub]:st[sub + 1]] for p, st in zip(parts, starts)])
            chunk.sort()
            out[pos:pos + len(chunk)] = chunk
            pos += len(chunk)
        assert pos == len(out) and n_sub == len(parts[0][0])
        return sizes.astype(np.uint32), out

    def export_table(self):
        """The ranks' tables as ONE host table (bfc_ch_union); with track_order its dump is byte-identical to `bfc -t1 -d`."""
        import ctypes as C
        from bfc_amd import api
        parts = [c.export_table() for c in self.ctx]
        arr = (C.c_void_p * len(parts))(*[p.ptr for p in parts])
        u = api._lib.load().bfc_ch_union(arr, len(parts))
        for p in parts:
            p.close()
        return api.HostTable(u)

    def close(self):
        for c, a, b in zip(self.ctx, self.d_send, self.d_recv):
            c.sync(); c.dev_free(a); c.dev_free(b[0]); c.dev_free(b[1]); c.close()
