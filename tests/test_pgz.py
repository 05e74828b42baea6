"""Parallel inflate of gzip input (bfc_amd/csrc/bfc_pgz.h; SURVEY 8f1, bseq.c:33-50 reads gzip through one gzread stream), CPU only.

The decoder guesses block starts, so the tests are about the two things that make it safe: (1) whatever the file -- compression
level, stored / fixed / dynamic blocks, flush points, many members, optional header fields, trailing garbage, chunks smaller than a
block -- the text is zlib's, byte for byte; (2) a damaged file is refused (bfc_count then reads it through gzread), and the batches
cut from a gzip file -- intact or damaged -- are the ones the reference's own bseq_read cuts from it."""
import ctypes as C
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

import oracle
from test_ingest import _digest, _fastq, _ref_digest


def _pgz(fn, threads, chunk, window=1 << 30):
    from bfc_amd import _lib
    out = (C.c_uint64 * 5)()
    rc = _lib.load().bfc_pgz_digest(fn.encode(), threads, chunk, window, out)
    return rc, [int(v) for v in out]


def _gz(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0, flush_kind=zlib.Z_SYNC_FLUSH):
    c = zlib.compressobj(level, zlib.DEFLATED, 31, 8, strategy)
    if not flush_every:
        return c.compress(data) + c.flush()
    parts = []
    for i in range(0, len(data), flush_every):
        parts.append(c.compress(data[i:i + flush_every])); parts.append(c.flush(flush_kind))
    return b"".join(parts) + c.flush()


def _member(data, extra=b"", name=b"", comment=b"", hcrc=False, level=6):
    """one gzip member with the optional header fields of RFC 1952"""
    flg = (4 if extra else 0) | (8 if name else 0) | (16 if comment else 0) | (2 if hcrc else 0)
    h = b"\x1f\x8b\x08" + bytes([flg]) + b"\0\0\0\0\0\x03"
    if extra:
        h += struct.pack("<H", len(extra)) + extra
    if name:
        h += name + b"\0"
    if comment:
        h += comment + b"\0"
    if hcrc:
        h += struct.pack("<H", zlib.crc32(h) & 0xffff)
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    return h + c.compress(data) + c.flush() + struct.pack("<II", zlib.crc32(data), len(data) & 0xffffffff)


def _zlib_all(z):
    """all members through zlib, as gzread walks them: (text, intact)"""
    out = []
    while z[:2] == b"\x1f\x8b":
        d = zlib.decompressobj(31)
        try:
            out.append(d.decompress(z))
        except zlib.error:
            return b"".join(out), False
        if not d.eof:
            return b"".join(out), False
        z = d.unused_data
    return b"".join(out), True


def _texts(rng):
    fq = _fastq(rng, 12000, 50, 150)
    return {
        "fastq": fq,
        "random": rng.integers(0, 256, 700_000).astype(np.uint8).tobytes(),
        "runs": b"".join(bytes([int(rng.integers(65, 70))]) * int(rng.integers(1, 3000)) for _ in range(800)),   # distance-1 copies, long matches
        "periodic": (b"ACGTTGCATTAGGCAT" * 40 + b"\n") * 3000,
        "short": b"@r\nACGT\n+\nIIII\n",
        "empty": b"",
    }


FILES = {
    "l1": lambda t: _gz(t, 1), "l6": lambda t: _gz(t, 6), "l9": lambda t: _gz(t, 9),
    "stored": lambda t: _gz(t, 0),
    "fixed": lambda t: _gz(t, 6, zlib.Z_FIXED),
    "huffman_only": lambda t: _gz(t, 6, zlib.Z_HUFFMAN_ONLY),
    "rle": lambda t: _gz(t, 6, zlib.Z_RLE),
    "sync_flush": lambda t: _gz(t, 6, flush_every=50_000),                               # empty stored blocks between the blocks (pigz writes these)
    "full_flush": lambda t: _gz(t, 6, flush_every=30_011, flush_kind=zlib.Z_FULL_FLUSH),
    "members": lambda t: b"".join(gzip.compress(t[i:i + 65280], 6) for i in range(0, max(len(t), 1), 65280)),     # bgzf-like: one block per member
    "members_hdr": lambda t: b"".join(_member(t[i:i + 200_000], extra=b"BC\x02\x00\x12\x34" if j % 2 else b"", name=b"x.fq" if j % 3 else b"", comment=b"c" if j % 5 == 0 else b"", hcrc=j % 4 == 1)
                                       for j, i in enumerate(range(0, max(len(t), 1), 200_000))),
    "empty_members": lambda t: gzip.compress(b"") + gzip.compress(t[:len(t) // 2]) + gzip.compress(b"") + gzip.compress(t[len(t) // 2:]) + gzip.compress(b""),
    "garbage_tail": lambda t: _gz(t, 6) + b"\0" * 37 + b"not gzip",
}


@pytest.mark.parametrize("kind", sorted(FILES))
def test_text_is_zlibs(gpu_lib, tmp_path, kind):
    rng = np.random.default_rng(zlib.crc32(kind.encode()))
    for name, text in _texts(rng).items():
        z = FILES[kind](text)
        want = zlib.decompressobj(31)
        fn = str(tmp_path / (name + ".gz")); open(fn, "wb").write(z)
        if len(z) < 18:
            continue
        for threads, chunk, window in ((1, 1 << 20, 1 << 30), (4, 1 << 16, 1 << 30), (8, 4096, 70_000), (7, 1500, 1 << 30), (16, 64, 999)):
            rc, o = _pgz(fn, threads, chunk, window)
            assert rc == 0 and (o[0], o[1]) == (len(text), zlib.crc32(text)), (kind, name, threads, chunk, rc, o)
    del want


def test_guessed_pieces_are_used(gpu_lib, tmp_path):
    """on an ordinary gzip'ed FASTQ nearly every chunk behind the first of a round comes from a guessed start"""
    rng = np.random.default_rng(11)
    text = _fastq(rng, 60000, 100, 150)
    fn = str(tmp_path / "big.gz"); open(fn, "wb").write(_gz(text, 6))
    rc, o = _pgz(fn, 8, 256 << 10)
    assert rc == 0 and (o[0], o[1]) == (len(text), zlib.crc32(text))
    n_chunks = (os.path.getsize(fn) + (256 << 10) - 1) // (256 << 10)
    assert o[2] >= n_chunks - o[4] - 1 and o[3] <= 1, o   # one exact piece per round, the rest guessed; at most one decoded again


@pytest.mark.parametrize("seed", range(12))
def test_damaged_gzip_is_refused_or_right(gpu_lib, tmp_path, seed):
    """truncation, flipped bytes in the deflate data or the trailer: never a wrong text -- either refused (-2) or, if the damage hit
    nothing the format checks (e.g. the MTIME field), zlib's text"""
    rng = np.random.default_rng(seed)
    text = _fastq(rng, 20000, 80, 150)
    z = bytearray(_gz(text, int(rng.choice([1, 6, 9]))) if seed % 3 else b"".join(gzip.compress(text[i:i + 300_000]) for i in range(0, len(text), 300_000)))
    what = seed % 4
    if what == 0:
        z = z[:int(rng.integers(20, len(z) - 1))]
    elif what == 1:
        z[int(rng.integers(10, len(z) - 8))] ^= 1 << int(rng.integers(0, 8))
    elif what == 2:
        z[len(z) - int(rng.integers(1, 9))] ^= 0x40   # CRC-32 / ISIZE of the last member
    else:
        p = int(rng.integers(10, len(z) - 100)); z[p:p + 50] = rng.integers(0, 256, 50).astype(np.uint8).tobytes()
    fn = str(tmp_path / "bad.gz"); open(fn, "wb").write(bytes(z))
    good, good_ok = _zlib_all(bytes(z))
    for threads, chunk in ((1, 1 << 20), (6, 50_000), (8, 3000)):
        rc, o = _pgz(fn, threads, chunk, 1 << 20)
        if rc == 0:
            assert good_ok and (o[0], o[1]) == (len(good), zlib.crc32(good)), (seed, threads, chunk)
        else:
            assert rc == -2 and not good_ok, (seed, threads, chunk, rc)


def _with_env(**kw):
    class E:
        def __enter__(self):
            self.old = {k: os.environ.get(k) for k in kw}
            os.environ.update({k: str(v) for k, v in kw.items()})
        def __exit__(self, *a):
            for k, v in self.old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    return E()


@pytest.mark.parametrize("kind", ["l6", "l1", "stored", "sync_flush", "members", "members_hdr", "garbage_tail"])
def test_batches_from_gzip_are_the_plain_files(gpu_lib, tmp_path, kind):
    rng = np.random.default_rng(zlib.crc32(kind.encode()) + 1)
    text = _fastq(rng, 9000, 30, 200)
    fn = str(tmp_path / "x.fq"); open(fn, "wb").write(text)
    gz = str(tmp_path / "x.fq.gz"); open(gz, "wb").write(FILES[kind](text))
    for chunk in (40_000, 1 << 30):
        plain = _digest(gpu_lib, fn, chunk, 0)
        with _with_env(BFC_INGEST_GZ_MIN=0, BFC_INGEST_GZ_CHUNK=20_000):
            for threads in (1, 4, 8):
                got = _digest(gpu_lib, gz, chunk, threads)
                assert got[:6] == plain[:6], (kind, chunk, threads)
                assert got[6] == plain[0], "every batch of a gzip'ed strict FASTQ comes from the parallel inflate + the fast path"
        assert _digest(gpu_lib, gz, chunk, 0)[:6] == plain[:6]   # and through gzread
        if oracle.have_ref():
            assert _ref_digest(gz, chunk)[:6] == plain[:6]


def test_gzip_that_stops_being_strict_fastq(gpu_lib, tmp_path):
    """a wrapped record in the middle of a gzip'ed FASTQ: fast path up to it, then gzread (which inflates up to there again)"""
    rng = np.random.default_rng(3)
    l = 130
    s = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), l).tobytes(); q = rng.integers(34, 74, l).astype(np.uint8).tobytes()
    text = _fastq(rng, 5000, 80, 120) + b"@wrapped\n" + s[:70] + b"\n" + s[70:] + b"\n+\n" + q[:70] + b"\n" + q[70:] + b"\n" + _fastq(rng, 2000, 80, 120) + b">fa\nACGT\n"
    fn = str(tmp_path / "w.fq"); open(fn, "wb").write(text)
    gz = str(tmp_path / "w.fq.gz"); open(gz, "wb").write(_gz(text))
    plain = _digest(gpu_lib, fn, 30_000, 0)
    with _with_env(BFC_INGEST_GZ_MIN=0, BFC_INGEST_GZ_CHUNK=10_000):
        got = _digest(gpu_lib, gz, 30_000, 5)
    assert got[:6] == plain[:6] and 0 < got[6] < plain[0]


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref/libbfcref.so not built (needs /root/reference)")
@pytest.mark.parametrize("seed", range(16))
def test_damaged_gzip_parses_like_the_reference(gpu_lib, tmp_path, seed):
    """what bseq_read gets out of a truncated or corrupted .gz (zlib hands out text until it meets the damage) is what bfc_count gets:
    the parallel inflate refuses the file at the damage and gzread takes over from the last batch boundary"""
    rng = np.random.default_rng(1000 + seed)
    text = _fastq(rng, 6000, 60, 150)
    z = bytearray(_gz(text, 6) if seed % 2 else b"".join(gzip.compress(text[i:i + 150_000]) for i in range(0, len(text), 150_000)))
    what = seed % 4
    if what == 0:
        z = z[:int(rng.integers(len(z) // 4, len(z) - 1))]
    elif what == 1:
        z[int(rng.integers(len(z) // 3, len(z) - 8))] ^= 1 << int(rng.integers(0, 8))
    elif what == 2:
        z[len(z) - 6] ^= 0x10
    else:
        z = z[:len(z) - 8]   # no trailer
    gz = str(tmp_path / "d.fq.gz"); open(gz, "wb").write(bytes(z))
    for chunk in (25_000, 1 << 30):
        want = _ref_digest(gz, chunk)[:6]
        assert _digest(gpu_lib, gz, chunk, 0)[:6] == want, (seed, chunk, "gzread path")
        with _with_env(BFC_INGEST_GZ_MIN=0, BFC_INGEST_GZ_CHUNK=15_000):
            for threads in (2, 6):
                assert _digest(gpu_lib, gz, chunk, threads)[:6] == want, (seed, chunk, threads)
