"""Deterministic synthetic read generator (SURVEY.md App. B.2 / BASELINE.md section 2).

ctypes front end of ``bfc_amd/csrc/bfcgen.c``.  Produces the SoA batch form the GPU path takes
(reads concatenated, one offset per read) and FASTQ files for the command-line / reference runs.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbfcgen.so")
_lib = None


def build():
    src = os.path.join(_HERE, "csrc", "bfcgen.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", _SO, src])


def _L():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.bfcgen_n_reads.restype = C.c_uint64
        L.bfcgen_n_reads.argtypes = [C.c_uint64, C.c_double, C.c_int]
        L.bfcgen_genome.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p]
        L.bfcgen_reads.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_int, C.c_double, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
        L.bfcgen_fastq.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_int, C.c_double, C.c_uint64, C.c_uint64, C.c_char_p]
        L.bfcgen_count_kmers.restype = C.c_uint64
        L.bfcgen_count_kmers.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int]
        L.bfcgen_to_stream.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_uint8, C.c_void_p]
        L.bfcgen_popcount.restype = C.c_uint64
        L.bfcgen_popcount.argtypes = [C.c_void_p, C.c_uint64]
        L.bfcgen_fnv1a64.restype = C.c_uint64
        L.bfcgen_fnv1a64.argtypes = [C.c_void_p, C.c_uint64]
        L.bfcgen_mix64.restype = C.c_uint64
        L.bfcgen_mix64.argtypes = [C.c_void_p, C.c_uint64]
        L.bfcgen_fnv1a64_from.restype = C.c_uint64
        L.bfcgen_fnv1a64_from.argtypes = [C.c_uint64, C.c_void_p, C.c_uint64]
        _lib = L
    return _lib


class ReadSet:
    """`bfcgen seed G cov L err`: genome + random-access reads."""

    def __init__(self, seed, G, cov, L=150, err=0.01):
        self.seed, self.G, self.cov, self.L, self.err = int(seed), int(G), float(cov), int(L), float(err)
        self.n_reads = int(_L().bfcgen_n_reads(self.G, self.cov, self.L))
        self.genome = np.empty(self.G, dtype=np.uint8)
        _L().bfcgen_genome(self.seed, self.G, self.genome.ctypes.data)

    def reads(self, r0=0, r1=None, out_seq=None, out_qual=None):
        """Reads r0..r1 as (seq u8[n*L], qual u8[n*L], off u64[n+1])."""
        r1 = self.n_reads if r1 is None else min(int(r1), self.n_reads)
        n = r1 - r0
        seq = out_seq if out_seq is not None else np.empty(n * self.L, dtype=np.uint8)
        qual = out_qual if out_qual is not None else np.empty(n * self.L, dtype=np.uint8)
        _L().bfcgen_reads(self.seed, self.G, self.genome.ctypes.data, self.L, self.err, r0, r1, seq.ctypes.data, qual.ctypes.data)
        off = np.arange(n + 1, dtype=np.uint64) * np.uint64(self.L)
        return seq, qual, off

    def fastq(self, fn, r0=0, r1=None):
        r1 = self.n_reads if r1 is None else min(int(r1), self.n_reads)
        if _L().bfcgen_fastq(self.seed, self.G, self.genome.ctypes.data, self.L, self.err, r0, r1, fn.encode()) != 0:
            raise OSError("cannot write " + fn)
        return fn

    def fastq_parallel(self, fn, r0=0, r1=None, threads=16):
        """The same file as fastq(), written by `threads` threads (ranges of reads into part files, appended in order): the single-threaded writer
        makes ~160 MB/s, 100 s for config c3's 15.6 GB."""
        import shutil
        from concurrent.futures import ThreadPoolExecutor
        r1 = self.n_reads if r1 is None else min(int(r1), self.n_reads)
        threads = max(1, min(int(threads), (r1 - r0 + 999_999) // 1_000_000))
        if threads == 1:
            return self.fastq(fn, r0, r1)
        per = (r1 - r0 + threads - 1) // threads
        parts = [(r0 + i * per, min(r1, r0 + (i + 1) * per), "%s.part%d" % (fn, i)) for i in range(threads) if r0 + i * per < r1]
        with ThreadPoolExecutor(len(parts)) as ex:  # (ctypes releases the GIL around the C writer)
            list(ex.map(lambda t: self.fastq(t[2], t[0], t[1]), parts))
        os.replace(parts[0][2], fn)
        with open(fn, "ab") as out:
            for _, _, pf in parts[1:]:
                with open(pf, "rb") as f:
                    shutil.copyfileobj(f, out, 64 << 20)
                os.unlink(pf)
        return fn


# named fixtures / configs
FIXTURES = {
    "g1": dict(seed=1, G=100_000, cov=10),
    "g42": dict(seed=42, G=1_000_000, cov=30),
    "c1": dict(seed=1, G=4_600_000, cov=1),
    "c2": dict(seed=2, G=4_600_000, cov=100),
    "c3": dict(seed=3, G=248_000_000, cov=30),
    "c4": dict(seed=4, G=3_100_000_000, cov=30),
}


def fixture(name):
    return ReadSet(**FIXTURES[name])


def count_kmers(seq, L, k):
    """Number of bfc_kmer_insert calls on fixed-length reads: positions with >= k ACGT in a row (count.c:83-88)."""
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    return int(_L().bfcgen_count_kmers(seq.ctypes.data, len(seq) // L, L, k))


def to_stream(arr, L, sep=10, out=None):
    """Fixed-length reads -> the stream form of the device API (every read followed by one separator byte)."""
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    n = len(arr) // L
    out = out if out is not None else np.empty(n * (L + 1), dtype=np.uint8)
    _L().bfcgen_to_stream(arr.ctypes.data, n, L, sep, out.ctypes.data)
    return out


def bitmap_checksums(bits):
    """(popcount, FNV-1a/64) of a bitmap, as SURVEY App. B.3 defines the bloom goldens."""
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    return int(_L().bfcgen_popcount(bits.ctypes.data, len(bits))), int(_L().bfcgen_fnv1a64(bits.ctypes.data, len(bits)))


def bitmap_mix64(bits):
    """The parallel digest of a bitmap (bfcgen_mix64): sum of its 64-bit words times position-dependent odd constants, modulo 2^64."""
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    return int(_L().bfcgen_mix64(bits.ctypes.data, len(bits)))
