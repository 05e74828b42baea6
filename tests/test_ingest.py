"""Host ingest, CPU only: the library's two parsers -- the serial kseq-grammar parser and the multi-threaded fast path for
uncompressed strict 4-line FASTQ -- must cut the same batches out of any input, and both must agree with the reference's own
bseq_read/kseq (through oracle/_ref/libbfcref.so when it is built).  bfc_ingest_digest parses without touching a GPU."""
import ctypes as C
import gzip
import os
import zlib

import numpy as np
import pytest

import oracle

u64p = C.POINTER(C.c_uint64)


def _digest(gpu_lib, fn, chunk, threads, cap=1 << 24):
    from bfc_amd import _lib
    out = (C.c_uint64 * 7)()
    assert _lib.load().bfc_ingest_digest(fn.encode(), chunk, cap, threads, out) == 0
    return [int(v) for v in out]


def _ref_digest(fn, chunk):
    out = (C.c_uint64 * 7)()
    assert oracle.ref().ref_ingest_digest(fn.encode(), chunk, out) == 0
    return [int(v) for v in out]


def _fastq(rng, n, lmin, lmax, crlf=False, qual_at=True, names=True):
    eol = b"\r\n" if crlf else b"\n"
    parts = []
    for r in range(n):
        l = int(rng.integers(lmin, lmax + 1))
        s = rng.choice(np.frombuffer(b"ACGTNacgt", dtype=np.uint8), l).tobytes()
        q = rng.integers(33, 74, l).astype(np.uint8)
        if qual_at and l and r % 3 == 0:
            q[0] = ord("@") if r % 2 else ord("+")   # quality lines that look like headers / separators
        name = b"@r%d some comment/1" % r if names else b"@%d" % r
        parts.append(name + eol + s + eol + b"+" + (name[1:] if r % 5 == 0 else b"") + eol + q.tobytes() + eol)
    return b"".join(parts)


CASES = ["plain", "crlf", "no_final_newline", "trailing_blank_lines", "one_read", "long_reads", "tiny_reads", "multiline", "strict_then_multiline",
         "fasta", "truncated_quality", "empty", "blank_only", "garbage_tail"]


def _make(case, path, rng):
    if case == "plain":
        data = _fastq(rng, 5000, 30, 250)
    elif case == "crlf":
        data = _fastq(rng, 3000, 1, 200, crlf=True)
    elif case == "no_final_newline":
        data = _fastq(rng, 2000, 50, 150)[:-1]
    elif case == "trailing_blank_lines":
        data = _fastq(rng, 2000, 50, 150) + b"\n\n\r\n\n"
    elif case == "one_read":
        data = _fastq(rng, 1, 100, 100)
    elif case == "long_reads":
        data = _fastq(rng, 40, 1, 300000)
    elif case == "tiny_reads":
        data = _fastq(rng, 20000, 1, 3, names=False)
    elif case == "multiline":
        recs = []
        for r in range(1500):
            l = int(rng.integers(61, 300))
            s = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), l).tobytes(); q = rng.integers(34, 74, l).astype(np.uint8).tobytes()
            recs.append(b"@m%d\n" % r + b"\n".join(s[i:i + 60] for i in range(0, l, 60)) + b"\n+\n" + b"\n".join(q[i:i + 60] for i in range(0, l, 60)) + b"\n")
        data = b"".join(recs)
    elif case == "strict_then_multiline":
        l = 130
        s = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), l).tobytes(); q = rng.integers(34, 74, l).astype(np.uint8).tobytes()
        data = _fastq(rng, 4000, 80, 120) + b"@wrapped\n" + s[:70] + b"\n" + s[70:] + b"\n+\n" + q[:70] + b"\n" + q[70:] + b"\n" + _fastq(rng, 1500, 80, 120)
    elif case == "fasta":
        data = b"".join(b">f%d\n" % r + rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), int(rng.integers(1, 400))).tobytes() + b"\n" for r in range(2000))
    elif case == "truncated_quality":
        data = _fastq(rng, 3000, 100, 100)
        data = data[:len(data) * 2 // 3]
        data = data[:data.rindex(b"\n+")] + b"\n+\nIIII\n"   # a quality line shorter than its sequence: kseq returns -2 and the input ends there
    elif case == "empty":
        data = b""
    elif case == "blank_only":
        data = b"\n\n\n"
    elif case == "garbage_tail":
        data = _fastq(rng, 2500, 60, 90) + b"this is not a record\nnor is this\n"
    open(path, "wb").write(data)
    return len(data)


@pytest.mark.parametrize("case", CASES)
def test_parsers_agree(gpu_lib, tmp_path, case):
    rng = np.random.default_rng(zlib.crc32(case.encode()))  # not hash(): that one changes from run to run
    fn = str(tmp_path / (case + ".fq"))
    size = _make(case, fn, rng)
    for chunk in (20000, 1 << 30) if size < (1 << 22) else (500000,):
        serial = _digest(gpu_lib, fn, chunk, 0)
        assert serial[6] == 0
        for threads in (1, 3, 8):
            fast = _digest(gpu_lib, fn, chunk, threads)
            assert fast[:6] == serial[:6], (case, chunk, threads)
        if case in ("plain", "crlf", "no_final_newline", "trailing_blank_lines", "long_reads", "tiny_reads") and serial[0]:
            assert fast[6] == serial[0], "every batch of a strict FASTQ comes from the fast path"
            os.environ["BFC_INGEST_MIN_SLICE"] = "64"  # 8 walks inside every batch, however small
            try:
                tiny = _digest(gpu_lib, fn, chunk, 8)
            finally:
                os.environ.pop("BFC_INGEST_MIN_SLICE", None)
            assert tiny[:6] == serial[:6] and tiny[6] == serial[0], (case, chunk)
        if case == "strict_then_multiline" and chunk == 20000:
            assert 0 < fast[6] < serial[0], "fast path until the wrapped record, serial parser from there"
        if case in ("multiline", "fasta"):
            assert fast[6] == 0
        if oracle.have_ref():
            assert _ref_digest(fn, chunk)[:6] == serial[:6], "the reference's bseq_read cuts other batches"


def test_gzip_and_small_batches(gpu_lib, tmp_path):
    rng = np.random.default_rng(5)
    data = _fastq(rng, 3000, 20, 180)
    fn = str(tmp_path / "x.fq"); open(fn, "wb").write(data)
    gz = str(tmp_path / "x.fq.gz"); gzip.open(gz, "wb").write(data)
    a = _digest(gpu_lib, fn, 7777, 4)
    b = _digest(gpu_lib, gz, 7777, 4)
    assert a[:6] == b[:6] and a[6] == a[0] and b[6] == 0
    # a batch buffer smaller than a chunk: the capacity cuts the batches, identically for both parsers
    c0, c4 = _digest(gpu_lib, fn, 1 << 30, 0, cap=50000), _digest(gpu_lib, fn, 1 << 30, 4, cap=50000)
    assert c0[:6] == c4[:6] and c0[0] > 5


def test_mapping_given_back_behind_the_parser(gpu_lib, tmp_path, monkeypatch):
    """Round 5: a thread of its own unmaps what the batches have consumed (2 MiB-aligned) while the parser goes on -- large inputs only, unless
    BFC_INGEST_UNMAP_MIN says otherwise.  12 MB of strict FASTQ in batches of ~0.5 MB: the same digests as the serial parser and as the fast
    path with the one munmap at the end; then a wrapped record at 3/4 of the file -- the serial parser takes over behind unmapped pages."""
    rng = np.random.default_rng(77)
    data = _fastq(rng, 60000, 60, 120)
    fn = str(tmp_path / "big.fq"); open(fn, "wb").write(data)
    assert len(data) > (10 << 20)
    serial = _digest(gpu_lib, fn, 500000, 0)
    plain = _digest(gpu_lib, fn, 500000, 8)
    monkeypatch.setenv("BFC_INGEST_UNMAP_MIN", "1")
    for threads in (1, 4, 8):
        got = _digest(gpu_lib, fn, 500000, threads)
        assert got[:6] == serial[:6] and got[6] == serial[0], threads
    assert plain[:6] == serial[:6]
    cut = data.index(b"\n@", len(data) * 3 // 4) + 1
    s = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 130).tobytes(); q = rng.integers(34, 74, 130).astype(np.uint8).tobytes()
    mixed = data[:cut] + b"@wrapped\n" + s[:70] + b"\n" + s[70:] + b"\n+\n" + q[:70] + b"\n" + q[70:] + b"\n" + data[cut:]
    fn2 = str(tmp_path / "mixed.fq"); open(fn2, "wb").write(mixed)
    monkeypatch.delenv("BFC_INGEST_UNMAP_MIN")
    serial2 = _digest(gpu_lib, fn2, 500000, 0)
    monkeypatch.setenv("BFC_INGEST_UNMAP_MIN", "1")
    got2 = _digest(gpu_lib, fn2, 500000, 8)
    assert got2[:6] == serial2[:6] and 0 < got2[6] < serial2[0]
    planes_a = _planes_digest(fn, 500000, 8, 20, 1)  # ... and the planes written straight from the (partly unmapped) file
    monkeypatch.delenv("BFC_INGEST_UNMAP_MIN")
    assert planes_a == _planes_digest(fn, 500000, 8, 20, 1)


def _mutate(rng, data):
    """random damage to a FASTA/FASTQ text: lines dropped, doubled, split, emptied, junk with '@' '>' '+' inside, CRLF, truncation"""
    lines = data.split(b"\n")
    for _ in range(int(rng.integers(1, 12))):
        if len(lines) < 3:
            break
        i = int(rng.integers(0, len(lines)))
        op = int(rng.integers(0, 10))
        if op == 0:
            del lines[i]
        elif op == 1:
            lines.insert(i, lines[i])
        elif op == 2:
            lines.insert(i, b"")
        elif op == 3:
            lines.insert(i, bytes(rng.choice(np.frombuffer(b"ACGT@>+ xyz\t", dtype=np.uint8), int(rng.integers(1, 30)))))
        elif op == 4 and len(lines[i]) > 2:
            c = int(rng.integers(1, len(lines[i])))
            lines[i:i + 1] = [lines[i][:c], lines[i][c:]]
        elif op == 5:
            lines[i] = lines[i] + b"\r"
        elif op == 6 and len(lines[i]) > 1:
            lines[i] = lines[i][:int(rng.integers(0, len(lines[i])))]
        elif op == 7:
            lines[i] = lines[i] + bytes(rng.choice(np.frombuffer(b"ACGTI#@", dtype=np.uint8), int(rng.integers(1, 9))))
        elif op == 8:
            lines[i] = (b">" if rng.random() < 0.5 else b"@") + lines[i]
        else:
            lines[i] = b"+" + lines[i]
    out = b"\n".join(lines)
    if rng.random() < 0.3:
        out = out[:int(rng.integers(0, len(out) + 1))]
    return out


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref/libbfcref.so not built (needs /root/reference)")
@pytest.mark.parametrize("seed", range(240))
def test_damaged_inputs_parse_like_the_reference(gpu_lib, tmp_path, seed):
    """Whatever the text, both parsers cut the batches bseq_read cuts (kseq's grammar incl. its error returns: a record with a bad
    quality string ends the batch, the next call goes on behind it; an empty batch ends the input)."""
    rng = np.random.default_rng(seed + 100003 * int(os.environ.get("BFC_FUZZ_SEED_BASE", "0")))  # other bases: other damaged texts
    kind = seed % 3
    if kind == 0:
        data = _fastq(rng, int(rng.integers(1, 400)), 1, 120, crlf=rng.random() < 0.2)
    elif kind == 1:
        data = b"".join(b">f%d x\n" % r + rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), int(rng.integers(0, 200))).tobytes() + b"\n" for r in range(int(rng.integers(1, 300))))
    else:
        data = _fastq(rng, 150, 10, 80) + b">fa\nACGTTGCA\nAC\n" + _fastq(rng, 150, 10, 80)
    fn = str(tmp_path / "d.fq")
    for rep in range(6):
        text = _mutate(rng, data)
        open(fn, "wb").write(text)
        for chunk in (300, 5000, 1 << 30):
            want = _ref_digest(fn, chunk)[:6]
            for threads, min_slice in ((0, None), (3, None), (5, "64")):  # the last: walks of a few records each, chained across junk
                if min_slice:
                    os.environ["BFC_INGEST_MIN_SLICE"] = min_slice
                try:
                    got = _digest(gpu_lib, fn, chunk, threads)[:6]
                finally:
                    os.environ.pop("BFC_INGEST_MIN_SLICE", None)
                if got != want:
                    open("/tmp/ingest_fail.fq", "wb").write(text)
                assert got == want, (seed, rep, chunk, threads, min_slice)


def _planes_digest(fn, chunk, threads, q, direct, cap=1 << 24):
    from bfc_amd import _lib
    out = (C.c_uint64 * 7)()
    assert _lib.load().bfc_ingest_planes_digest(fn.encode(), chunk, cap, threads, q, direct, out) == 0
    return [int(v) for v in out]


@pytest.mark.parametrize("case", CASES)
def test_bit_planes_straight_from_the_file(gpu_lib, tmp_path, case):
    """bfc_count on one GPU hands its batches over as bit planes; the FASTQ fast path writes them straight from the mapped file (no byte streams
    in between): word for word what bfcg_pack_planes makes of the serial parser's byte streams -- any read lengths (a record may begin at any
    bit of a word, threads share words at their seams), CRLF, thresholds a signed char cannot reach, inputs that fall back to the serial parser."""
    rng = np.random.default_rng(zlib.crc32(case.encode()) + 17)
    fn = str(tmp_path / (case + ".fq"))
    size = _make(case, fn, rng)
    for chunk in (20000, 1 << 30) if size < (1 << 22) else (500000,):
        for q in (20, 0, 41, 95, -200):
            want = _planes_digest(fn, chunk, 0, q, 0)
            assert want[6] == 0
            for threads in (1, 3, 8):
                got = _planes_digest(fn, chunk, threads, q, 1)
                assert got[:6] == want[:6], (case, chunk, q, threads)
            if case in ("plain", "crlf", "tiny_reads", "long_reads") and want[0]:
                assert got[6] == want[0], "every batch of a strict FASTQ is packed directly"
                os.environ["BFC_INGEST_MIN_SLICE"] = "64"  # 8 walks inside every batch, however small: seams inside words
                try:
                    tiny = _planes_digest(fn, chunk, 8, q, 1)
                finally:
                    os.environ.pop("BFC_INGEST_MIN_SLICE", None)
                assert tiny[:6] == want[:6] and tiny[6] == want[0], (case, chunk, q)
            if q != 20:
                continue


# ------------------------------------------------------------------ pipe input (round 6): a FIFO drained into a ring, the same chained walks


def _digest_fifo(gpu_lib, tmp_path, data, chunk, threads, piece=None, cap=1 << 24):
    """the digest of `data` arriving through a named pipe, written in pieces of `piece` bytes (None: at once) by a thread of this process"""
    import threading
    fifo = str(tmp_path / "in.fifo")
    if os.path.exists(fifo):
        os.unlink(fifo)
    os.mkfifo(fifo)

    def feed():
        with open(fifo, "wb", buffering=0) as f:
            try:
                if piece is None:
                    f.write(data)
                else:
                    for i in range(0, len(data), piece):
                        f.write(data[i:i + piece])
            except BrokenPipeError:
                pass
    th = threading.Thread(target=feed)
    th.start()
    try:
        return _digest(gpu_lib, fifo, chunk, threads, cap=cap)
    finally:
        th.join()
        os.unlink(fifo)


@pytest.mark.parametrize("case", CASES)
def test_pipe_input_parses_like_the_file(gpu_lib, tmp_path, case):
    """VERDICT r5 missing 6: a pipe (the published command feeds `<(seqtk mergepe ...)`, tex/README.md:26; bseq.c:33-50) takes the multi-threaded FASTQ
    path too -- a reader thread drains it into a ring mapped twice, the chained walks run on windows of the ring -- and where the text is not strict
    4-line FASTQ the serial parser goes on FROM THE RING (a pipe cannot be read again).  Same batches as the file's, by every parser."""
    rng = np.random.default_rng(zlib.crc32(case.encode()))
    fn = str(tmp_path / (case + ".fq"))
    size = _make(case, fn, rng)
    data = open(fn, "rb").read()
    for chunk in (20000, 1 << 30) if size < (1 << 22) else (500000,):
        want = _digest(gpu_lib, fn, chunk, 0)
        for threads, piece in ((1, None), (4, 4099), (8, 65536)):
            got = _digest_fifo(gpu_lib, tmp_path, data, chunk, threads, piece)
            assert got[:6] == want[:6], (case, chunk, threads)
            if case in ("plain", "crlf", "no_final_newline", "trailing_blank_lines", "long_reads", "tiny_reads") and want[0] and chunk < (1 << 30):
                assert got[6] == want[0], "every batch of a strict FASTQ on a pipe comes from the fast path"
            if case in ("multiline", "fasta"):
                assert got[6] == 0
        if case == "strict_then_multiline" and chunk == 20000:
            assert 0 < got[6] < want[0], "fast path until the wrapped record, serial parser from the ring behind it"


def test_pipe_ring_wraps_and_gzip_on_a_pipe(gpu_lib, tmp_path, monkeypatch):
    """12 MB of strict FASTQ through a ring of 4 MiB (the windows wrap around its end three times; the reader waits for the parser), then through a
    ring too small for one window (the serial parser takes everything, from the ring), then gzip data on a pipe (left to gzread: nothing consumed
    by the look at its first two bytes), and BFC_INGEST_NO_PIPE=1 (the old way)."""
    rng = np.random.default_rng(78)
    data = _fastq(rng, 60000, 60, 120)
    fn = str(tmp_path / "big.fq"); open(fn, "wb").write(data)
    want = _digest(gpu_lib, fn, 500000, 0)
    monkeypatch.setenv("BFC_INGEST_RING", str(4 << 20))
    for threads in (1, 8):
        got = _digest_fifo(gpu_lib, tmp_path, data, 500000, threads, 1 << 20)
        assert got[:6] == want[:6] and got[6] == want[0], threads
    monkeypatch.setenv("BFC_INGEST_RING", str(1 << 20))  # (rounded up to 2 MiB: a first window of 3 x 500 000 + 256 Ki bytes still fits; 1.5 M bases do not)
    big = _digest(gpu_lib, fn, 1500000, 0)
    got = _digest_fifo(gpu_lib, tmp_path, data, 1500000, 4, 1 << 20)
    assert got[:6] == big[:6] and got[6] == 0
    monkeypatch.delenv("BFC_INGEST_RING")
    gz = gzip.compress(data[:3_000_000], 1)
    small = str(tmp_path / "small.fq"); open(small, "wb").write(data[:3_000_000])
    want_gz = _digest(gpu_lib, small, 500000, 0)
    got = _digest_fifo(gpu_lib, tmp_path, gz, 500000, 4, 70001)
    assert got[:6] == want_gz[:6] and got[6] == 0
    monkeypatch.setenv("BFC_INGEST_NO_PIPE", "1")
    got = _digest_fifo(gpu_lib, tmp_path, data, 500000, 4, 1 << 20)
    assert got[:6] == want[:6] and got[6] == 0


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref/libbfcref.so not built (needs /root/reference)")
@pytest.mark.parametrize("seed", range(40))
def test_damaged_inputs_on_a_pipe_parse_like_the_reference(gpu_lib, tmp_path, seed):
    """The damaged texts of the test above, through a pipe: the reference's bseq_read on the FILE says what the batches are."""
    rng = np.random.default_rng(7000 + seed)
    kind = seed % 3
    if kind == 0:
        data = _fastq(rng, int(rng.integers(1, 400)), 1, 120, crlf=rng.random() < 0.2)
    elif kind == 1:
        data = b"".join(b">f%d x\n" % r + rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), int(rng.integers(0, 200))).tobytes() + b"\n" for r in range(int(rng.integers(1, 300))))
    else:
        data = _fastq(rng, 150, 10, 80) + b">fa\nACGTTGCA\nAC\n" + _fastq(rng, 150, 10, 80)
    fn = str(tmp_path / "d.fq")
    for rep in range(4):
        text = _mutate(rng, data)
        open(fn, "wb").write(text)
        for chunk in (300, 5000, 1 << 30):
            want = _ref_digest(fn, chunk)[:6]
            for threads, min_slice in ((3, None), (5, "64")):
                if min_slice:
                    os.environ["BFC_INGEST_MIN_SLICE"] = min_slice
                try:
                    got = _digest_fifo(gpu_lib, tmp_path, text, chunk, threads, int(rng.integers(1, 5000)))[:6]
                finally:
                    os.environ.pop("BFC_INGEST_MIN_SLICE", None)
                if got != want:
                    open("/tmp/ingest_pipe_fail.fq", "wb").write(text)
                assert got == want, (seed, rep, chunk, threads, min_slice)
