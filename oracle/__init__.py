"""oracle -- TEST INFRASTRUCTURE ONLY (never imported by the product package ``bfc_amd``).

ctypes bindings for
  * ``oracle/liboracle.so``      the CPU restatement (``bfc_oracle.c``), and
  * ``oracle/_ref/libbfcref.so`` the reference compiled in place from /root/reference (optional;
                                 built by ``oracle/Makefile`` where the reference is present,
                                 travels prebuilt to the GPU box).
Allowed importers: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` -- as the checker, never as the thing measured or shipped.
"""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")
u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)


def build(quiet=True):
    """(Re)build liboracle.so and, when /root/reference exists, oracle/_ref/*."""
    out = subprocess.run(["make", "-C", HERE], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if not quiet:
        print(out.stdout)


def _ptr(a, ty):
    return a.ctypes.data_as(ty) if a is not None else None


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_mix64.restype = C.c_uint64
        L.orc_mix64.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_kmer_push.argtypes = [C.c_int, u64p, C.c_int]
        L.orc_kmer_hash.restype = C.c_uint64
        L.orc_kmer_hash.argtypes = [C.c_int, u64p, u64p]
        L.orc_hash_from_y.restype = C.c_uint64
        L.orc_hash_from_y.argtypes = [C.c_int, u64p]
        L.orc_bf_positions.restype = C.c_uint64
        L.orc_bf_positions.argtypes = [C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_int)]
        L.orc_bf_new.restype = C.c_void_p
        L.orc_bf_new.argtypes = [C.c_int, C.c_int]
        L.orc_bf_free.argtypes = [C.c_void_p]
        L.orc_bf_bits.restype = C.c_void_p
        L.orc_bf_bits.argtypes = [C.c_void_p]
        L.orc_bf_nbytes.restype = C.c_uint64
        L.orc_bf_nbytes.argtypes = [C.c_void_p]
        L.orc_bf_insert.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_bf_get.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_popcount_bytes.restype = C.c_uint64
        L.orc_popcount_bytes.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_fnv1a64.restype = C.c_uint64
        L.orc_fnv1a64.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_ch_new.restype = C.c_void_p
        L.orc_ch_new.argtypes = [C.c_int, C.c_int]
        L.orc_ch_free.argtypes = [C.c_void_p]
        L.orc_ch_k.argtypes = [C.c_void_p]
        L.orc_ch_lpre.argtypes = [C.c_void_p]
        L.orc_ch_subkey.restype = C.c_uint64
        L.orc_ch_subkey.argtypes = [C.c_int, C.c_int, u64p, u64p]
        L.orc_ch_clamp_lpre.argtypes = [C.c_int, C.c_int]
        L.orc_ch_insert.argtypes = [C.c_void_p, u64p, C.c_int]
        L.orc_ch_get.argtypes = [C.c_void_p, u64p]
        L.orc_ch_count.restype = C.c_uint64
        L.orc_ch_count.argtypes = [C.c_void_p]
        L.orc_ch_hist.argtypes = [C.c_void_p, u64p, u64p]
        L.orc_ch_dump.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_ch_export.restype = C.c_uint64
        L.orc_ch_export.argtypes = [C.c_void_p, u32p, u64p]
        L.orc_state_new.restype = C.c_void_p
        L.orc_state_new.argtypes = [C.c_int] * 6
        L.orc_state_free.argtypes = [C.c_void_p]
        for f in ("orc_state_bf", "orc_state_bf_high", "orc_state_ch"):
            getattr(L, f).restype = C.c_void_p
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_state_stats.argtypes = [C.c_void_p, u64p]
        L.orc_count_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, u64p]
        L.orc_count_batch.restype = C.c_uint64
        L.orc_count_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, u64p, C.c_uint64, u64p]
        L.orc_kmers_in_read.restype = C.c_uint64
        L.orc_kmers_in_read.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_max_streak.restype = C.c_uint64
        L.orc_max_streak.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_trim_decide.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_kcov.restype = None
        L.orc_kcov.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def have_ref():
    return os.path.exists(os.path.join(REF_DIR, "libbfcref.so"))


_ref = None
_ref_ec = None


def have_ref_ec():
    return os.path.exists(os.path.join(REF_DIR, "libbfcref_ec.so"))


def ref_ec():
    """The reference with its corrector's private bfc_ec_kcov reachable (libbfcref_ec.so, ref_shim_ec.c)."""
    global _ref_ec
    if _ref_ec is None:
        R = C.CDLL(os.path.join(REF_DIR, "libbfcref_ec.so"))
        R.bfc_ch_restore.restype = C.c_void_p
        R.bfc_ch_restore.argtypes = [C.c_char_p]
        R.bfc_ch_destroy.argtypes = [C.c_void_p]
        R.ref_kcov.restype = None
        R.ref_kcov.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_void_p]
        R.ref_trim_batch.restype = None
        R.ref_trim_batch.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p, u64p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
        _ref_ec = R
    return _ref_ec


def ref():
    """The reference itself (libbfcref.so). Raises if it was never built."""
    global _ref
    if _ref is None:
        R = C.CDLL(os.path.join(REF_DIR, "libbfcref.so"))
        R.ref_hash_64.restype = C.c_uint64
        R.ref_hash_64.argtypes = [C.c_uint64, C.c_uint64]
        R.ref_kmer_append.argtypes = [C.c_int, u64p, C.c_int]
        R.ref_kmer_hash.restype = C.c_uint64
        R.ref_kmer_hash.argtypes = [C.c_int, u64p, u64p]
        R.ref_kmer_hash_inv.argtypes = [C.c_int, u64p, u64p]
        R.ref_state_new.restype = C.c_void_p
        R.ref_state_new.argtypes = [C.c_int] * 5
        R.ref_state_free.argtypes = [C.c_void_p]
        for f in ("ref_state_bf", "ref_state_bf_high", "ref_state_ch"):
            getattr(R, f).restype = C.c_void_p
            getattr(R, f).argtypes = [C.c_void_p]
        R.ref_bf_bits.restype = C.c_void_p
        R.ref_bf_bits.argtypes = [C.c_void_p]
        R.ref_state_stats.argtypes = [C.c_void_p, u64p]
        R.ref_count_batch.restype = C.c_uint64
        R.ref_count_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, u64p, C.c_uint64, u64p]
        # the reference's own API (bbf.h:14-17, htab.h:13-23)
        R.bfc_bf_insert.argtypes = [C.c_void_p, C.c_uint64]
        R.bfc_bf_get.argtypes = [C.c_void_p, C.c_uint64]
        R.bfc_ch_get.argtypes = [C.c_void_p, u64p]
        R.bfc_ch_count.restype = C.c_uint64
        R.bfc_ch_count.argtypes = [C.c_void_p]
        R.bfc_ch_hist.argtypes = [C.c_void_p, u64p, u64p]
        R.ref_ingest_digest.argtypes = [C.c_char_p, C.c_int, u64p]
        R.ref_count_batch_blocks.restype = C.c_uint64
        R.ref_count_batch_blocks.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, u64p, C.c_uint64, C.c_int]
        R.bfc_ch_dump.argtypes = [C.c_void_p, C.c_char_p]
        R.bfc_ch_restore.restype = C.c_void_p
        R.bfc_ch_restore.argtypes = [C.c_char_p]
        R.bfc_ch_destroy.argtypes = [C.c_void_p]
        _ref = R
    return _ref


def ref_trim(bf_ptr, k, seq, off, min_frac=0.9, n_threads=1):
    """The reference's own trim pass (worker_ec -> max_streak + keep rule, correct.c:478-497,557-567, reached through ref_shim_ec.c) on a
    bfc_bf_t* of either reference library.  Returns (start int32[n], end int32[n]); start -1 = read dropped."""
    off = np.ascontiguousarray(off, dtype=np.uint64)
    n = len(off) - 1
    st = np.empty(n, dtype=np.int32); en = np.empty(n, dtype=np.int32)
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    ref_ec().ref_trim_batch(bf_ptr, k, C.c_float(min_frac), seq.ctypes.data, _ptr(off, u64p), n, int(n_threads), st.ctypes.data, en.ctypes.data)
    return st, en


# ----------------------------------------------------------------------------- convenience


class Counter:
    """Sequential count state on top of either the restatement ('oracle') or the reference ('ref')."""

    def __init__(self, k, bf_shift, q=20, n_hashes=4, l_pre=20, filter_mode=0, impl="oracle"):
        self.k, self.q, self.bf_shift, self.n_hashes, self.filter_mode, self.impl = k, q, bf_shift, n_hashes, filter_mode, impl
        if impl == "oracle":
            self.L = lib()
            self.st = self.L.orc_state_new(k, q, bf_shift, n_hashes, l_pre, filter_mode)
        else:
            self.L = ref()
            self.st = self.L.ref_state_new(k, bf_shift, n_hashes, l_pre, filter_mode)
        self.l_pre = lib().orc_ch_clamp_lpre(k, l_pre)

    def close(self):
        if self.st:
            (self.L.orc_state_free if self.impl == "oracle" else self.L.ref_state_free)(self.st)
            self.st = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def count(self, seq, qual, off, trace=False):
        """seq/qual: uint8 arrays (concatenated reads); off: uint64 offsets (n_reads+1). Returns trace or n."""
        n_reads = len(off) - 1
        tr = None
        if trace:
            nk = sum(int(lib().orc_kmers_in_read(seq[off[r]:].ctypes.data, int(off[r + 1] - off[r]), self.k)) for r in range(n_reads))
            tr = np.zeros((nk, 4), dtype=np.uint64)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        if self.impl == "oracle":
            n = self.L.orc_count_batch(self.st, seq.ctypes.data, qual.ctypes.data if qual is not None else None,
                                       _ptr(off, u64p), n_reads, _ptr(tr, u64p))
        else:
            n = self.L.ref_count_batch(self.st, self.k, self.q, self.n_hashes, seq.ctypes.data,
                                       qual.ctypes.data if qual is not None else None, _ptr(off, u64p), n_reads, _ptr(tr, u64p))
        return tr if trace else n

    def count_blocks(self, seq, qual, off, n_parts):
        """impl='ref' only: the same answers as count() from n_parts threads, each owning the bloom blocks of one residue class
        (ref_shim.c: ref_count_batch_blocks -- every block still sees its k-mers in file order).  The table's layout is not `bfc -t1`'s."""
        assert self.impl == "ref"
        off = np.ascontiguousarray(off, dtype=np.uint64)
        return self.L.ref_count_batch_blocks(self.st, self.k, self.q, self.n_hashes, seq.ctypes.data,
                                             qual.ctypes.data if qual is not None else None, _ptr(off, u64p), len(off) - 1, int(n_parts))

    def stats(self):
        out = np.zeros(4, dtype=np.uint64)
        (self.L.orc_state_stats if self.impl == "oracle" else self.L.ref_state_stats)(self.st, _ptr(out, u64p))
        return dict(n_kmers=int(out[0]), n_high=int(out[1]), n_seen=int(out[2]), hash_xor=int(out[3]))

    def _bf(self, high=False):
        if self.impl == "oracle":
            return (self.L.orc_state_bf_high if high else self.L.orc_state_bf)(self.st)
        return (self.L.ref_state_bf_high if high else self.L.ref_state_bf)(self.st)

    def bloom_bytes(self, high=False):
        """numpy view (copy) of the bloom bitmap."""
        n = 1 << (self.bf_shift - 3)
        p = self.L.orc_bf_bits(self._bf(high)) if self.impl == "oracle" else self.L.ref_bf_bits(self._bf(high))
        return np.ctypeslib.as_array(C.cast(p, u8p), shape=(n,)).copy()

    def bloom_view(self, high=False):
        """the bitmap in place (no copy; valid until close())"""
        n = 1 << (self.bf_shift - 3)
        p = self.L.orc_bf_bits(self._bf(high)) if self.impl == "oracle" else self.L.ref_bf_bits(self._bf(high))
        return np.ctypeslib.as_array(C.cast(p, u8p), shape=(n,))

    def bloom_checksums(self, high=False):
        n = 1 << (self.bf_shift - 3)
        p = self.L.orc_bf_bits(self._bf(high)) if self.impl == "oracle" else self.L.ref_bf_bits(self._bf(high))
        return int(lib().orc_popcount_bytes(p, n)), int(lib().orc_fnv1a64(p, n))

    def ch(self):
        return (self.L.orc_state_ch if self.impl == "oracle" else self.L.ref_state_ch)(self.st)

    def table_count(self):
        return int(self.L.orc_ch_count(self.ch()) if self.impl == "oracle" else self.L.bfc_ch_count(self.ch()))

    def table_hist(self):
        cnt = np.zeros(256, dtype=np.uint64)
        high = np.zeros(64, dtype=np.uint64)
        f = self.L.orc_ch_hist if self.impl == "oracle" else self.L.bfc_ch_hist
        mode = f(self.ch(), _ptr(cnt, u64p), _ptr(high, u64p))
        return mode, cnt, high

    def table_get(self, y0, y1):
        y = (C.c_uint64 * 2)(y0, y1)
        return (self.L.orc_ch_get if self.impl == "oracle" else self.L.bfc_ch_get)(self.ch(), y)

    def dump(self, fn):
        f = self.L.orc_ch_dump if self.impl == "oracle" else self.L.bfc_ch_dump
        return f(self.ch(), fn.encode())

    def export(self):
        """L1 form: (sizes[2^l_pre] u32, slots u64 sorted per sub-table)."""
        if self.impl != "oracle":
            import tempfile
            with tempfile.NamedTemporaryFile(suffix=".hash") as tf:
                self.dump(tf.name)
                return parse_dump(tf.name)[2:]
        sizes = np.zeros(1 << self.l_pre, dtype=np.uint32)
        n = self.L.orc_ch_export(self.ch(), _ptr(sizes, u32p), None)
        slots = np.zeros(n, dtype=np.uint64)
        self.L.orc_ch_export(self.ch(), _ptr(sizes, u32p), _ptr(slots, u64p))
        return sizes, slots


def parse_dump(fn):
    """Parse a `bfc -d` dump (htab.c:129-149). Returns k, l_pre, sizes(u32), slots sorted per sub-table (u64)."""
    raw = np.fromfile(fn, dtype=np.uint32)
    k, l_pre = int(raw[0]), int(raw[1])
    n_sub = 1 << l_pre
    sizes = np.zeros(n_sub, dtype=np.uint32)
    chunks = []
    p = 2
    for i in range(n_sub):
        sz = int(raw[p + 1])
        sizes[i] = sz
        p += 2
        if sz:
            chunks.append(np.sort(raw[p:p + 2 * sz].copy().view(np.uint64)))
            p += 2 * sz
    slots = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint64)
    return k, l_pre, sizes, slots


def l1_digest(sizes, slots):
    """Canonical layout-free digest of a count table: md5 over, per sub-table i in order,
    <u32 i><u32 size> followed by its slot values sorted ascending as little-endian u64."""
    n_sub = len(sizes)
    sizes = np.asarray(sizes, dtype=np.uint32)
    hdr = np.empty((n_sub, 2), dtype=np.uint32)
    hdr[:, 0] = np.arange(n_sub, dtype=np.uint32)
    hdr[:, 1] = sizes
    # interleave headers and slot runs without a python loop over 2^20 tables:
    total = 2 * n_sub + 2 * int(sizes.sum())
    out = np.empty(total, dtype=np.uint32)
    starts = np.zeros(n_sub + 1, dtype=np.int64)
    np.cumsum(sizes.astype(np.int64), out=starts[1:])
    hpos = 2 * np.arange(n_sub, dtype=np.int64) + 2 * starts[:-1]
    out[hpos] = hdr[:, 0]
    out[hpos + 1] = hdr[:, 1]
    mask = np.ones(total, dtype=bool)
    mask[hpos] = False
    mask[hpos + 1] = False
    out[mask] = np.ascontiguousarray(slots, dtype=np.uint64).view(np.uint32)
    return hashlib.md5(out.tobytes()).hexdigest()


def md5_file(fn):
    h = hashlib.md5()
    with open(fn, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()
