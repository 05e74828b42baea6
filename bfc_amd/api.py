"""Python mirror of the reference's count-phase interface on top of the C ABI (libbfc_gpu.so).

Names follow the reference: ``bfc_opt_init`` (bfc.c:17-40), ``bfc_opt_by_size`` (bfc.c:42-53),
``bfc_count`` (count.c:127), and the query surface of ``bfc_ch_t`` / ``bfc_bf_t`` (htab.h, bbf.h).
``GpuCounter`` is the device-level interface (batches of reads already in memory).
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import BfcOpt, BfcgParams, BfcKmer, u64p, u32p, f32p

STAT_NAMES = {0: "n_kmers", 1: "n_high", 2: "n_seen", 3: "n_keys", 4: "tab_ovf", 5: "err_pool", 6: "slow_buckets", 7: "crowded_regions", 8: "tab_cshift", 9: "n_batches"}


class BfcGpuError(RuntimeError):
    pass


def bfc_opt_init():
    """Defaults of bfc.c:17-40."""
    o = BfcOpt()
    o.chunk_size = 100000000
    o.n_threads = 1
    o.q = 20
    o.k = 33
    o.l_pre = 20
    o.bf_shift = 33
    o.n_hashes = 4
    o.min_frac = 0.9
    o.min_cov = 3
    o.win_multi_ec = 10
    o.max_end_ext = 5
    o.w_ec, o.w_ec_high, o.w_absent, o.w_absent_high = 1, 7, 3, 1
    o.max_path_diff, o.max_heap = 15, 100
    return o


def bfc_opt_by_size(opt, size):
    """`-s`: bfc.c:42-53."""
    bits = math.log(size) / math.log(2)
    opt.k = int(bits + 1.0)
    if opt.k & 1 == 0:
        opt.k += 1
    opt.k = min(opt.k, 63)
    opt.bf_shift = min(int(bits + 8.0), 37)
    return opt


def to_stream(seq, off):
    """(concatenated reads, offsets) -> separator-delimited stream (one '\\n' after each read)."""
    off = np.asarray(off, dtype=np.int64)
    n = len(off) - 1
    out = np.empty(len(seq) + n, dtype=np.uint8)
    lens = np.diff(off)
    if n and np.all(lens == lens[0]):
        L = int(lens[0])
        v = out.reshape(n, L + 1)
        v[:, :L] = np.asarray(seq).reshape(n, L)
        v[:, L] = 10
    else:
        pos = off[:-1] + np.arange(n)
        mask = np.ones(len(out), dtype=bool)
        mask[off[1:] + np.arange(n)] = False
        out[mask] = seq
        out[~mask] = 10
        del pos
    return out


class HostTable:
    """A host-resident ``bfc_ch_t`` (opaque, htab.h:10-23)."""

    def __init__(self, ptr):
        if not ptr:
            raise BfcGpuError("NULL bfc_ch_t")
        self.L = _lib.load()
        self.ptr = ptr

    def close(self):
        if self.ptr:
            self.L.bfc_ch_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def k(self):
        return self.L.bfc_ch_get_k(self.ptr)

    @property
    def l_pre(self):
        return self.L.bfc_ch_get_lpre(self.ptr)

    def get(self, y0, y1):
        return self.L.bfc_ch_get(self.ptr, (C.c_uint64 * 2)(y0, y1))

    def kmer_occ(self, x):
        z = BfcKmer()
        for i in range(4):
            z.x[i] = int(x[i])
        return self.L.bfc_ch_kmer_occ(self.ptr, C.byref(z))

    def insert(self, y0, y1, is_high, forced=1):
        return self.L.bfc_ch_insert(self.ptr, (C.c_uint64 * 2)(y0, y1), int(is_high), forced)

    def count(self):
        return int(self.L.bfc_ch_count(self.ptr))

    def hist(self):
        cnt = np.zeros(256, dtype=np.uint64)
        high = np.zeros(64, dtype=np.uint64)
        mode = self.L.bfc_ch_hist(self.ptr, cnt.ctypes.data_as(u64p), high.ctypes.data_as(u64p))
        return mode, cnt, high

    def dump(self, fn):
        return self.L.bfc_ch_dump(self.ptr, fn.encode())

    def export_sorted(self):
        sizes = np.zeros(1 << self.l_pre, dtype=np.uint32)
        n = self.L.bfc_ch_export_sorted(self.ptr, sizes.ctypes.data_as(u32p), None)
        slots = np.zeros(int(n), dtype=np.uint64)
        self.L.bfc_ch_export_sorted(self.ptr, sizes.ctypes.data_as(u32p), slots.ctypes.data_as(u64p))
        return sizes, slots

    @staticmethod
    def init(k, l_pre):
        return HostTable(_lib.load().bfc_ch_init(k, l_pre))

    @staticmethod
    def restore(fn):
        p = _lib.load().bfc_ch_restore(fn.encode())
        return HostTable(p) if p else None


class HostBloom:
    """A host-resident ``bfc_bf_t`` (bbf.h:9-12)."""

    def __init__(self, ptr):
        if not ptr:
            raise BfcGpuError("NULL bfc_bf_t")
        self.L = _lib.load()
        self.ptr = ptr

    def close(self):
        if self.ptr:
            self.L.bfc_bf_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_shift(self):
        return self.ptr.contents.n_shift

    @property
    def n_hashes(self):
        return self.ptr.contents.n_hashes

    def bytes(self):
        return np.ctypeslib.as_array(self.ptr.contents.b, shape=(1 << (self.n_shift - 3),))

    def get(self, h):
        return self.L.bfc_bf_get(self.ptr, h)

    def insert(self, h):
        return self.L.bfc_bf_insert(self.ptr, h)

    @staticmethod
    def init(n_shift, n_hashes):
        p = _lib.load().bfc_bf_init(n_shift, n_hashes)
        return HostBloom(p) if p else None


def bfc_count(fn, opt):
    """count.c:127: count the k-mers of file `fn` on the GPU; returns HostTable or (filter mode) HostBloom."""
    L = _lib.load()
    p = L.bfc_count(fn.encode(), C.byref(opt))
    if opt.filter_mode:
        return HostBloom(C.cast(p, C.POINTER(_lib.BfcBf)))
    return HostTable(p)


def pack_planes(seq_stream, qual_stream, q, n_chunks=1):
    """The four bit planes of a byte-stream batch (bfcg_pack_planes; no GPU involved): uint32 array [4, bfcg_plane_words(n)].
    n_chunks > 1 packs the stream as that many word ranges one after the other (what disjoint threads would each take: tests of the seams)."""
    L = _lib.load()
    seq_stream = np.ascontiguousarray(seq_stream, dtype=np.uint8)
    qs = np.ascontiguousarray(qual_stream, dtype=np.uint8) if qual_stream is not None else None
    n = len(seq_stream)
    pw = int(L.bfcg_plane_words(n))
    planes = np.zeros((4, pw), dtype=np.uint32)
    if qs is None:
        planes[3, :] = 0xffffffff
    step = ((n + n_chunks - 1) // n_chunks + 31) // 32 * 32 if n_chunks > 1 else max(n, 32)
    for lo in range(0, n, max(step, 32)):
        L.bfcg_pack_planes(seq_stream.ctypes.data, qs.ctypes.data if qs is not None else None, lo, min(n, lo + max(step, 32)), n, q, planes.ctypes.data, pw)
    return planes


class GpuCounter:
    """Device-level counting context (bfcg_ctx_t)."""

    def __init__(self, k, bf_shift, q=20, n_hashes=4, l_pre=20, filter_mode=0, device=0, max_batch_pos=1 << 24,
                 region_shift=0, tab_cshift=0, debug_seen=False, rank=0, n_ranks=1, track_order=False, table_layout=0):
        self.L = _lib.load()
        p = BfcgParams()
        self.L.bfcg_params_default(C.byref(p))
        p.k, p.q, p.bf_shift, p.n_hashes, p.l_pre, p.filter_mode = k, q, bf_shift, n_hashes, l_pre, filter_mode
        p.device, p.max_batch_pos, p.region_shift, p.tab_cshift, p.debug_seen = device, int(max_batch_pos), region_shift, tab_cshift, int(debug_seen)
        p.rank, p.n_ranks, p.track_order, p.table_layout = rank, n_ranks, int(track_order), int(table_layout)
        self.rank, self.n_ranks = rank, n_ranks
        self.params = p
        self.bf_shift, self.k = bf_shift, k
        self.ctx = self.L.bfcg_create(C.byref(p))
        if not self.ctx:
            raise BfcGpuError("bfcg_create failed: " + self.L.bfcg_last_error().decode())

    @classmethod
    def _view(cls, ctx, params, n_ranks):
        """A GpuCounter over a context somebody else owns (a group's rank)."""
        self = cls.__new__(cls)
        self.L = _lib.load()
        self.ctx, self.params, self._borrowed = ctx, params, True
        self.rank, self.n_ranks, self.bf_shift, self.k = 0, n_ranks, params.bf_shift, params.k
        return self

    def _ck(self, rc):
        if rc != 0:
            raise BfcGpuError(self.L.bfcg_last_error().decode())

    def close(self):
        if self.ctx and not getattr(self, "_borrowed", False):
            self.L.bfcg_destroy(self.ctx)
        self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self._ck(self.L.bfcg_reset(self.ctx))

    def count_host(self, seq_stream, qual_stream=None):
        seq_stream = np.ascontiguousarray(seq_stream, dtype=np.uint8)
        q = np.ascontiguousarray(qual_stream, dtype=np.uint8) if qual_stream is not None else None
        self._ck(self.L.bfcg_count_batch_host(self.ctx, seq_stream.ctypes.data, q.ctypes.data if q is not None else None, len(seq_stream)))

    def count_planes(self, planes, first_pos, n_pos, has_qual=True):
        """positions [first_pos, first_pos + n_pos) of a plane set made by pack_planes (4 bits per position over PCIe instead of 16)"""
        planes = np.ascontiguousarray(planes, dtype=np.uint32)
        assert planes.ndim == 2 and planes.shape[0] == 4
        self._ck(self.L.bfcg_count_batch_planes(self.ctx, planes.ctypes.data, planes.shape[1], first_pos, n_pos, 1 if has_qual else 0))

    def count_dev(self, d_seq, d_qual, n_pos):
        self._ck(self.L.bfcg_count_batch_dev(self.ctx, d_seq, d_qual, n_pos))

    # ---- multi-GPU stages (owner computes): what bfcg_group_* drives from inside the library; exposed for tests (tests/mg_protocol.py)
    def mg_info(self):
        out = (C.c_int * 4)()
        self.L.bfcg_mg_info(self.ctx, out)
        return dict(nb1=out[0], nb_loc=out[1], rec_bytes=out[2], n_ranks=out[3])

    def batch_limit(self):
        """positions per batch (per rank: of the global batch / n_ranks) the filter's regions take at full speed"""
        return int(self.L.bfcg_batch_limit(self.ctx))

    def mg_scatter(self, d_seq, d_qual, n_pos, d_send):
        counts = np.zeros(self.mg_info()["nb1"], dtype=np.uint32)
        self._ck(self.L.bfcg_mg_scatter(self.ctx, d_seq, d_qual, n_pos, d_send, counts.ctypes.data_as(u32p)))
        return counts

    def mg_process(self, d_recv, seg_cnt):
        seg_cnt = np.ascontiguousarray(seg_cnt, dtype=np.uint32)
        self._ck(self.L.bfcg_mg_process(self.ctx, d_recv, seg_cnt.ctypes.data_as(u32p)))

    def dev_alloc(self, nbytes):
        p = self.L.bfcg_dev_alloc(self.ctx, nbytes)
        if not p:
            raise BfcGpuError(self.L.bfcg_last_error().decode())
        return p

    def dev_free(self, p):
        self.L.bfcg_dev_free(self.ctx, p)

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        self._ck(self.L.bfcg_h2d(self.ctx, dptr, arr.ctypes.data, arr.nbytes))

    def sync(self):
        self._ck(self.L.bfcg_sync(self.ctx))

    def stats(self):
        out = np.zeros(16, dtype=np.uint64)
        self._ck(self.L.bfcg_stats(self.ctx, out.ctypes.data_as(u64p)))
        d = {STAT_NAMES[i]: int(out[i]) for i in STAT_NAMES}
        d["stream_batches"] = int(self.L.bfcg_stream_batches(self.ctx))
        d["phase_cycles"] = [int(out[i]) for i in range(10, 16)]  # BFCG_ABLATE&64: k_bloom stage/pass1/pass2/writeback/handover
        return d

    def table_info(self):
        """How the count table is held right now: region-owned segments (updated through LDS) or the host's (sub-table, key) layout."""
        out = (C.c_int * 4)()
        self.L.bfcg_table_info(self.ctx, out)
        return dict(segments=bool(out[0]), seg_shift=out[1], tab_cshift=out[2], seg_growths=out[3])

    def partition_info(self):
        out = (C.c_uint64 * 2)()
        self.L.bfcg_partition_info(self.ctx, out)
        return dict(one_pass=bool(out[0] & 1), level2_one_pass=bool(out[0] & 2), replayed_batches=int(out[1]))

    def s1wc_launches(self):
        """Launches of k_scatter1_wc (level 1 through write-combining buffers in LDS) by this process so far."""
        return int(self.L.bfcg_s1wc_launches())

    def last_batch_ms(self):
        out = np.zeros(6, dtype=np.float32)
        self.L.bfcg_last_batch_ms(self.ctx, out.ctypes.data_as(f32p))
        return dict(hist1=float(out[0]), scatter1=float(out[1]), level2=float(out[2]), bloom=float(out[3]), commit=float(out[4]), total=float(out[5]))

    def stage_ms(self, reset=False):
        """Cumulative per-stage GPU ms over all batches since the last reset, and their number (drains the pipeline)."""
        out = (C.c_double * 6)()
        n = C.c_uint64()
        self._ck(self.L.bfcg_stage_ms(self.ctx, out, C.byref(n), int(reset)))
        return dict(hist1=out[0], scatter1=out[1], level2=out[2], bloom=out[3], commit=out[4], total=out[5]), int(n.value)

    def bloom_bytes(self, which=0):
        out = np.empty((1 << (self.bf_shift - 3)) // self.n_ranks, dtype=np.uint8)  # the slice this rank owns
        self._ck(self.L.bfcg_bloom_to_host(self.ctx, which, out.ctypes.data))
        return out

    def export_bloom(self, which=0, resident=False):
        """Host bfc_bf_t of filter `which`; resident=True also leaves a copy in HBM for a GpuTrimmer to adopt (what bfc_count does)."""
        p = (self.L.bfcg_export_bloom_resident if resident else self.L.bfcg_export_bloom)(self.ctx, which)
        if not p:
            raise BfcGpuError(self.L.bfcg_last_error().decode())
        return HostBloom(p)

    def export_table(self):
        p = self.L.bfcg_export_table(self.ctx)
        if not p:
            raise BfcGpuError(self.L.bfcg_last_error().decode())
        return HostTable(p)

    def hash_positions(self, seq_stream, qual_stream=None):
        seq_stream = np.ascontiguousarray(seq_stream, dtype=np.uint8)
        q = np.ascontiguousarray(qual_stream, dtype=np.uint8) if qual_stream is not None else None
        out = np.zeros((len(seq_stream), 3), dtype=np.uint64)
        self._ck(self.L.bfcg_hash_positions(self.ctx, seq_stream.ctypes.data, q.ctypes.data if q is not None else None, len(seq_stream), out.ctypes.data_as(u64p)))
        return out

    def seen_flags(self, n_pos):
        out = np.zeros(n_pos, dtype=np.uint8)
        self._ck(self.L.bfcg_seen_flags(self.ctx, out.ctypes.data, n_pos))
        return out


class GpuGroup:
    """The local ranks of a multi-GPU run, driven inside libbfc_gpu.so (bfcg_group_t): stage A, the exchange of the k-mer records over RCCL
    (or peer copies between the devices of one process) and stage B, one host thread per rank.  `devices` lists the local ranks' devices --
    all ranks of the run (one process), or one of them together with `uid` (one process per GPU).  A device may be repeated: ranks emulated
    on one GPU.  max_batch_pos = positions of ONE rank's share of a global batch."""

    def __init__(self, k, bf_shift, devices, max_batch_pos, n_ranks=None, first_rank=0, uid=None, transport=0, q=20, n_hashes=4, l_pre=20,
                 filter_mode=0, track_order=False, table_layout=0, region_shift=0, tab_cshift=0):
        self.L = _lib.load()
        p = BfcgParams()
        self.L.bfcg_params_default(C.byref(p))
        p.k, p.q, p.bf_shift, p.n_hashes, p.l_pre, p.filter_mode = k, q, bf_shift, n_hashes, l_pre, filter_mode
        p.max_batch_pos, p.region_shift, p.tab_cshift, p.track_order, p.table_layout = int(max_batch_pos), region_shift, tab_cshift, int(track_order), int(table_layout)
        self.params, self.k, self.bf_shift = p, k, bf_shift
        self.devices = list(devices)
        self.n_local = len(self.devices)
        self.n_ranks = n_ranks if n_ranks is not None else self.n_local
        dv = (C.c_int * self.n_local)(*self.devices)
        self._uid = (C.c_uint8 * 128).from_buffer_copy(bytes(uid)) if uid is not None else None
        self.g = self.L.bfcg_group_create(C.byref(p), self.n_ranks, first_rank, self.n_local, dv, self._uid, transport)
        if not self.g:
            raise BfcGpuError("bfcg_group_create failed: " + self.L.bfcg_last_error().decode())

    @staticmethod
    def unique_id():
        L = _lib.load()
        buf = (C.c_uint8 * 128)()
        if L.bfcg_group_unique_id(buf) != 0:
            raise BfcGpuError(L.bfcg_last_error().decode())
        return bytes(buf)

    def _ck(self, rc):
        if rc != 0:
            raise BfcGpuError(self.L.bfcg_last_error().decode())

    def close(self):
        if self.g:
            self.L.bfcg_group_destroy(self.g)
            self.g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        out = (C.c_int * 6)()
        self.L.bfcg_group_info(self.g, out)
        return dict(n_ranks=out[0], n_local=out[1], transport={1: "rccl", 2: "peer", 3: "push"}.get(out[2], out[2]), rec_bytes=out[3], nb1=out[4], first_rank=out[5],
                    slab_mode=bool(self.L.bfcg_group_slab_mode(self.g)), lazy_batches=int(self.L.bfcg_group_lazy_batches(self.g)))

    def exchange_bytes(self):
        """Bytes the local ranks put on the links since creation / reset, and the bytes of the live records among them (bfcg_group_exchange_bytes)."""
        out = (C.c_uint64 * 4)()
        self._ck(self.L.bfcg_group_exchange_bytes(self.g, out))
        return dict(links=int(out[0]), exact=int(out[1]), batches=int(out[2]), transport={1: "rccl", 2: "peer", 3: "push"}.get(int(out[3]), int(out[3])))

    def ctx(self, i):
        """Local rank i's counting context as a GpuCounter view (owned by the group)."""
        return GpuCounter._view(self.L.bfcg_group_ctx(self.g, i), self.params, self.n_ranks)

    def reset(self):
        self._ck(self.L.bfcg_group_reset(self.g))

    def sync(self):
        self._ck(self.L.bfcg_group_sync(self.g))

    def count_host(self, seq_stream, qual_stream=None):
        """One global batch from host memory; the library cuts it into the ranks' shares (every rank local)."""
        seq_stream = np.ascontiguousarray(seq_stream, dtype=np.uint8)
        q = np.ascontiguousarray(qual_stream, dtype=np.uint8) if qual_stream is not None else None
        self._ck(self.L.bfcg_group_count_batch_host(self.g, seq_stream.ctypes.data, q.ctypes.data if q is not None else None, len(seq_stream)))

    def count_dev(self, d_seq, d_qual, n_pos):
        """One global batch: lists (one entry per local rank) of device pointers on the rank's own device and stream lengths."""
        n = self.n_local
        ds = (C.c_void_p * n)(*[int(v) if v else None for v in d_seq])
        dq = (C.c_void_p * n)(*[int(v) if v else None for v in d_qual]) if d_qual is not None else None
        npos = (C.c_uint64 * n)(*[int(v) for v in n_pos])
        self._ck(self.L.bfcg_group_count_batch_dev(self.g, ds, dq, npos))

    def stats(self):
        out = np.zeros(16, dtype=np.uint64)
        self._ck(self.L.bfcg_group_stats(self.g, out.ctypes.data_as(u64p)))
        return {STAT_NAMES[i]: int(out[i]) for i in STAT_NAMES}

    def export_table(self):
        p = self.L.bfcg_group_export_table(self.g)
        if not p:
            raise BfcGpuError(self.L.bfcg_last_error().decode())
        return HostTable(p)

    def export_bloom(self, which=0):
        p = self.L.bfcg_group_export_bloom(self.g, which)
        if not p:
            raise BfcGpuError(self.L.bfcg_last_error().decode())
        return HostBloom(p)


class GpuTrimmer:
    """Trim pass of `bfc -1` on the GPU (bfcg_trim_*): bloom query kernel + longest streak per read."""

    def __init__(self, k, bloom, device=0, max_pos=1 << 24, max_reads=1 << 18):
        self.L = _lib.load()
        self.k = k
        self.t = self.L.bfcg_trim_create(k, bloom.ptr, device, int(max_pos), int(max_reads))
        if not self.t:
            raise BfcGpuError("bfcg_trim_create failed: " + self.L.bfcg_last_error().decode())

    def close(self):
        if self.t:
            self.L.bfcg_trim_destroy(self.t)
            self.t = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def trim(self, seq_stream, off, min_frac=0.9, d_seq=None):
        """off: stream offsets (n_reads+1). Returns (start int32[n], end int32[n]); start -1 = read dropped."""
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = len(off) - 1
        start = np.empty(n, dtype=np.int32); end = np.empty(n, dtype=np.int32)
        s = np.ascontiguousarray(seq_stream, dtype=np.uint8) if seq_stream is not None else None
        rc = self.L.bfcg_trim_batch(self.t, s.ctypes.data if s is not None else None, d_seq, int(off[-1]), off.ctypes.data_as(u64p), n,
                                    C.c_float(min_frac), start.ctypes.data_as(C.POINTER(C.c_int32)), end.ctypes.data_as(C.POINTER(C.c_int32)))
        if rc != 0:
            raise BfcGpuError(self.L.bfcg_last_error().decode())
        return start, end

    def last_ms(self):
        return float(self.L.bfcg_trim_last_ms(self.t))

    @property
    def adopted(self):
        """True if the filter was found resident in HBM (left there by the count pass) instead of being uploaded."""
        return bool(self.L.bfcg_trim_adopted(self.t))


class GpuKcov:
    """bfc_ec_kcov (correct.c:96-117) for whole batches on the GPU (bfcg_kcov_*): table probe per k-mer + coverage sums.
    `table` is a HostTable (uploaded once) or a GpuCounter whose device table is used in place."""

    LCOV, HCOV, SOLID_END, HIGH_END = 0x3f, 0x3f << 6, 1 << 12, 1 << 13

    def __init__(self, table, device=0, max_pos=1 << 24):
        self.L = _lib.load()
        self._keep = table
        if isinstance(table, GpuCounter):
            self.t = self.L.bfcg_kcov_attach(table.ctx, int(max_pos))
        else:
            self.t = self.L.bfcg_kcov_create(table.ptr, device, int(max_pos))
        if not self.t:
            raise BfcGpuError("bfcg_kcov_create failed: " + self.L.bfcg_last_error().decode())

    def close(self):
        if self.t:
            self.L.bfcg_kcov_destroy(self.t)
            self.t = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def kcov(self, seq_stream, min_occ=3, d_seq=None, n_pos=None, fetch=True):
        """One packed u16 per stream position: lcov | hcov<<6 | solid_end<<12 | high_end<<13."""
        s = np.ascontiguousarray(seq_stream, dtype=np.uint8) if seq_stream is not None else None
        n = len(s) if s is not None else int(n_pos)
        out = np.empty(n, dtype=np.uint16) if fetch else None
        rc = self.L.bfcg_kcov_batch(self.t, s.ctypes.data if s is not None else None, d_seq, n, int(min_occ), out.ctypes.data if fetch else None)
        if rc != 0:
            raise BfcGpuError(self.L.bfcg_last_error().decode())
        return out

    def last_ms(self):
        return float(self.L.bfcg_kcov_last_ms(self.t))

    def dev_seq(self):
        return self.L.bfcg_kcov_dev_seq(self.t)
