"""GPU tests (-m gpu) of the drop-in boundary: bfc_count(fn, opt) fed from FASTA/FASTQ files, and the reference's
own UNMODIFIED main() + corrector linked against libbfc_gpu.so (oracle/_ref/bfc-dropin, built in place by
oracle/Makefile where /root/reference exists; it travels prebuilt to the GPU box)."""
import gzip
import hashlib
import os
import subprocess

import numpy as np
import pytest

import oracle
from bfc_amd import gen

pytestmark = pytest.mark.gpu

DROPIN = os.path.join(oracle.REF_DIR, "bfc-dropin")


def _missing(*paths):
    """These tests only run where a GPU is (-m gpu).  The reference-built binaries under oracle/_ref/ are made in the build container
    (oracle/Makefile, from /root/reference in place) and must TRAVEL to the GPU box: a missing one is a failure, never a skip."""
    gone = [p for p in paths if not os.path.exists(p)]
    return "did not travel to the GPU box (built by `make -C oracle` where /root/reference exists): " + ", ".join(gone) if gone else None


@pytest.fixture
def dropin_bin():
    m = _missing(DROPIN)
    if m:
        pytest.fail(m)
    return DROPIN


needs_dropin = pytest.mark.usefixtures("dropin_bin")


@pytest.fixture
def gputrim_bin():
    m = _missing(os.path.join(oracle.REF_DIR, "bfc-dropin-gputrim"))
    if m:
        pytest.fail(m)


@pytest.fixture
def ref_bin():
    m = _missing(os.path.join(oracle.REF_DIR, "bfc-ref"))
    if m:
        pytest.fail(m)


@pytest.fixture(scope="module")
def g1_fq(tmp_path_factory):
    d = tmp_path_factory.mktemp("fq")
    fn = str(d / "g1.fq")
    gen.fixture("g1").fastq(fn)
    assert oracle.md5_file(fn) == "7e17d87fc623a6f7171a607d68dde51a"
    return fn


def _opt(gpu_lib, **kw):
    o = gpu_lib.bfc_opt_init()
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def test_bfc_count_fastq(gpu_lib, g1_fq):
    """bfc_count on a FASTQ file (count.c:127): table identical (L1) to `bfc -t1`; chunking (-L) is invisible."""
    for chunk in (100000000, 100000):
        t = gpu_lib.bfc_count(g1_fq, _opt(gpu_lib, k=31, bf_shift=26, chunk_size=chunk))
        assert t.count() == 99561
        assert oracle.l1_digest(*t.export_sorted()) == "237be10261b07ef0677f8136b0a327b6"
        assert t.hist()[0] == 4
        t.close()


def test_bfc_count_gz_fasta_and_no_mt_io(gpu_lib, g1_fq, tmp_path):
    """gzip input, FASTA input (every k-mer high quality, count.c:85; SURVEY B.3 FASTA golden) and -J (no reader thread)."""
    gz = str(tmp_path / "g1.fq.gz")
    with open(g1_fq, "rb") as f, gzip.open(gz, "wb", compresslevel=1) as g:
        g.write(f.read())
    t = gpu_lib.bfc_count(gz, _opt(gpu_lib, k=31, bf_shift=26, no_mt_io=1))
    assert oracle.l1_digest(*t.export_sorted()) == "237be10261b07ef0677f8136b0a327b6"
    t.close()
    fa = str(tmp_path / "g1.fa")
    with open(g1_fq) as f, open(fa, "w") as g:
        for i, line in enumerate(f):
            if i % 4 == 0:
                g.write(">" + line[1:])
            elif i % 4 == 1:
                g.write(line[:70] + "\n" + line[70:])  # multi-line FASTA
    t = gpu_lib.bfc_count(fa, _opt(gpu_lib, k=31, bf_shift=26))
    sizes, slots = t.export_sorted()
    assert t.count() == 99561 and oracle.l1_digest(sizes, slots) == "4e4705f65c9bcc880b6e28a3453cc4fc"
    assert np.array_equal((slots >> np.uint64(8)) & np.uint64(0x3f), np.minimum(slots & np.uint64(0xff), np.uint64(63)))
    t.close()


def test_bfc_count_gz_inflated_by_several_threads(gpu_lib, g1_fq, tmp_path, monkeypatch):
    """-t 4 on a gzip file: the ingest threads inflate the ONE deflate stream together (bfc_pgz.h; forced here for a file below 1 MiB and
    with chunks of 20 KB, so that most pieces come from guessed block starts) and parse the text into batches; the table is `bfc -t1`'s."""
    monkeypatch.setenv("BFC_INGEST_GZ_MIN", "0")
    monkeypatch.setenv("BFC_INGEST_GZ_CHUNK", "20000")
    gz = str(tmp_path / "g1.fq.gz")
    with open(g1_fq, "rb") as f, gzip.open(gz, "wb", compresslevel=6) as g:
        g.write(f.read())
    for chunk in (100000000, 150000):
        t = gpu_lib.bfc_count(gz, _opt(gpu_lib, k=31, bf_shift=26, n_threads=4, chunk_size=chunk))
        assert t.count() == 99561 and oracle.l1_digest(*t.export_sorted()) == "237be10261b07ef0677f8136b0a327b6"
        t.close()


def test_bfc_count_filter_mode(gpu_lib, g1_fq):
    """-1 mode returns the bloom filter of k-mers seen twice (count.c:148-154) as a host bfc_bf_t."""
    b = gpu_lib.bfc_count(g1_fq, _opt(gpu_lib, k=51, bf_shift=26, filter_mode=1))
    bits = b.bytes()
    L = oracle.lib()
    assert (b.n_shift, b.n_hashes) == (26, 4)
    assert int(L.orc_popcount_bytes(bits.ctypes.data, len(bits))) == 364980
    assert int(L.orc_fnv1a64(bits.ctypes.data, len(bits))) == 0xafeb3dee4fc349c5
    b.close()


def _run(args, stdin=None):
    r = subprocess.run([DROPIN] + args, stdin=stdin, capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r


@needs_dropin
def test_dropin_full_pipeline(g1_fq):
    """Reference main() + correct.c, unmodified, consuming the GPU-built table: corrected reads byte-identical
    to reference `bfc -k31 -b26 -t1 g1.fq` (golden md5 from SURVEY B.3)."""
    r = _run(["-k", "31", "-b", "26", "-t", "4", g1_fq])
    assert hashlib.md5(r.stdout).hexdigest() == "06a4284e5010e34a1645d1d8015d6da9"
    assert r.stdout.startswith(b"@r0\tec:Z:0_0:2_0_1:0_0")


@needs_dropin
def test_dropin_trim_mode_and_stdin(g1_fq):
    """`bfc -1` (count + bloom-query trim pass in correct.c:478-497,556-569): stdout md5 golden; reads from stdin too."""
    r = _run(["-1", "-k", "51", "-b", "26", "-t", "2", g1_fq])
    assert hashlib.md5(r.stdout).hexdigest() == "f751f7b1aa28fd74b23194bc7f70158c"
    with open(g1_fq, "rb") as f:
        r2 = _run(["-E", "-k", "31", "-b", "26", "-d", "/dev/null", "-"], stdin=f)
    assert b"# distinct k-mers: 99561" in r2.stderr


@needs_dropin
@pytest.mark.parametrize("how", ["stdin_pipe", "fifo", "fifo_serial"])
def test_dropin_reads_a_pipe(g1_fq, tmp_path, how):
    """The published command line feeds bfc `<(seqtk mergepe ...)` (tex/README.md:26): a PIPE.  Through the reference's unmodified main(): `cat g1.fq | bfc-dropin ... -` and a named
    FIFO take the ring + multi-threaded FASTQ path (round 6; `-t 4`), `fifo_serial` the old serial parser (BFC_INGEST_NO_PIPE=1) -- the same dump as from the file, L1-identical
    to `bfc -t1 -d`, the same `# distinct k-mers` line."""
    import threading
    dump = str(tmp_path / "p.hash")
    args = ["-E", "-k", "31", "-b", "26", "-t", "4", "-d", dump]
    env = dict(os.environ, BFC_GPU_TIMING="1")
    if how == "stdin_pipe":
        cat = subprocess.Popen(["cat", g1_fq], stdout=subprocess.PIPE)
        r = subprocess.run([DROPIN] + args + ["-"], stdin=cat.stdout, capture_output=True, timeout=600, env=env)
        cat.wait()
    else:
        if how == "fifo_serial":
            env["BFC_INGEST_NO_PIPE"] = "1"
        fifo = str(tmp_path / "in.fifo")
        os.mkfifo(fifo)
        data = open(g1_fq, "rb").read()

        def feed():
            with open(fifo, "wb") as f:
                for i in range(0, len(data), 50_000):
                    f.write(data[i:i + 50_000])
        th = threading.Thread(target=feed)
        th.start()
        r = subprocess.run([DROPIN] + args + [fifo], capture_output=True, timeout=600, env=env)
        th.join()
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert b"# distinct k-mers: 99561" in r.stderr
    assert (b"a pipe: reader thread into a ring" in r.stderr) == (how != "fifo_serial"), r.stderr.decode()[-1500:]
    k, l_pre, sizes, slots = oracle.parse_dump(dump)
    assert (k, l_pre) == (31, 20) and oracle.l1_digest(sizes, slots) == "237be10261b07ef0677f8136b0a327b6"


@needs_dropin
def test_dropin_dump_restores_in_reference(g1_fq, tmp_path):
    """-d dump written by this library: L1-identical to `bfc -t1 -d`, and the *reference* restores it (-r) and
    corrects with it to the same bytes."""
    dump = str(tmp_path / "g1.hash")
    _run(["-E", "-k", "31", "-b", "26", "-d", dump, g1_fq])
    k, l_pre, sizes, slots = oracle.parse_dump(dump)
    assert (k, l_pre) == (31, 20) and oracle.l1_digest(sizes, slots) == "237be10261b07ef0677f8136b0a327b6"
    ref = os.path.join(oracle.REF_DIR, "bfc-ref")
    r = subprocess.run([ref, "-r", dump, "-t", "2", g1_fq], capture_output=True, timeout=600)
    assert r.returncode == 0 and hashlib.md5(r.stdout).hexdigest() == "06a4284e5010e34a1645d1d8015d6da9"


def test_gpu_trim_pass_matches_reference(gpu_lib, g1):
    """Config c5's query kernel: count in filter mode on the GPU, then the GPU trim pass (bloom query + max_streak + keep rule)
    per read equals the oracle's correct.c:478-497,557-569 restatement, and the formatted output has the md5 of
    `bfc -1 -k51 -b26 -t1 g1.fq` (golden from the reference binary)."""
    import ctypes as C
    rs, (seq, qual, off) = g1
    k, b = 51, 26
    g = gpu_lib.GpuCounter(k, b, filter_mode=1, max_batch_pos=len(seq) + rs.n_reads + 64)
    s_seq, s_qual = gpu_lib.to_stream(seq, off), gpu_lib.to_stream(qual, off)
    g.count_host(s_seq, s_qual)
    bf = g.export_bloom(1)
    g.close()
    soff = off + np.arange(rs.n_reads + 1, dtype=np.uint64)  # stream offsets (one separator per read)
    tr = gpu_lib.GpuTrimmer(k, bf, max_pos=len(s_seq) + 64, max_reads=rs.n_reads)
    start, end = tr.trim(s_seq, soff, 0.9)
    L = oracle.lib()
    oc = oracle.Counter(k, b, filter_mode=1)
    oc.count(seq, qual, off)
    obf = L.orc_state_bf_high(oc.st)
    out = []
    for r in range(rs.n_reads):
        s = seq[int(off[r]):int(off[r + 1])]
        mx = L.orc_max_streak(k, obf, s.ctypes.data, len(s))
        a, e = C.c_int(), C.c_int()
        if L.orc_trim_decide(mx, k, len(s), 0.9, C.byref(a), C.byref(e)):
            assert (int(start[r]), int(end[r])) == (a.value, e.value), r
            q = qual[int(off[r]):int(off[r + 1])]
            out.append(b"@r%d\n%s\n+\n%s\n" % (r, s[a.value:e.value].tobytes(), q[a.value:e.value].tobytes()))
        else:
            assert start[r] == -1, r
    assert hashlib.md5(b"".join(out)).hexdigest() == "f751f7b1aa28fd74b23194bc7f70158c"
    tr.close(); bf.close(); oc.close()


def test_filter_left_in_hbm_for_the_trim_pass(gpu_lib, g1):
    """bfc_count leaves bf_high in HBM behind the host object it returns (bfcg_export_bloom_resident); the trim context adopts that
    copy (no upload) and trims exactly as from an uploaded filter.  The copy goes away with the host object, and a host-side
    bfc_bf_insert makes it stale: the next trim context uploads the modified host bytes."""
    rs, (seq, qual, off) = g1
    k, b = 31, 24
    g = gpu_lib.GpuCounter(k, b, filter_mode=1, max_batch_pos=len(seq) + rs.n_reads + 64)
    s_seq, s_qual = gpu_lib.to_stream(seq, off), gpu_lib.to_stream(qual, off)
    g.count_host(s_seq, s_qual)
    plain, res, res2 = g.export_bloom(1), g.export_bloom(1, resident=True), g.export_bloom(1, resident=True)
    g.close()
    assert np.array_equal(plain.bytes(), res.bytes())
    soff = off + np.arange(rs.n_reads + 1, dtype=np.uint64)
    kw = dict(max_pos=len(s_seq) + 64, max_reads=rs.n_reads)
    t0 = gpu_lib.GpuTrimmer(k, plain, **kw); t1 = gpu_lib.GpuTrimmer(k, res, **kw)
    assert not t0.adopted and t1.adopted
    a0, a1 = t0.trim(s_seq, soff, 0.9), t1.trim(s_seq, soff, 0.9)
    assert np.array_equal(a0[0], a1[0]) and np.array_equal(a0[1], a1[1]) and (a0[0] >= 0).sum() > 100
    t2 = gpu_lib.GpuTrimmer(k, res, **kw)  # the copy was handed over to t1: a second context uploads
    assert not t2.adopted
    t0.close(); t1.close(); t2.close()
    # a host write through the library drops the stale copy: the next context sees the host's bytes
    from bfc_amd import _lib
    L = _lib.load()
    before = res2.bytes().copy()
    rng = np.random.default_rng(3)
    for h in rng.integers(0, 1 << 62, 200000, dtype=np.uint64):
        L.bfc_bf_insert(res2.ptr, int(h))
    assert not np.array_equal(before, res2.bytes())
    t3 = gpu_lib.GpuTrimmer(k, res2, **kw)
    assert not t3.adopted
    a3 = t3.trim(s_seq, soff, 0.9)
    assert (a3[0] >= 0).sum() >= (a0[0] >= 0).sum()  # more bits set: never fewer reads kept
    t3.close()
    res3_probe = gpu_lib.GpuTrimmer(k, plain, **kw); assert not res3_probe.adopted; res3_probe.close()
    plain.close(); res.close(); res2.close()


GPUTRIM = os.path.join(oracle.REF_DIR, "bfc-dropin-gputrim")


@pytest.mark.usefixtures("gputrim_bin")
def test_dropin_gpu_trim_binary(g1_fq, tmp_path):
    """`bfc -1` with BOTH phases on the GPU (bfc_count + bfc_correct from libbfc_gpu.so, reference main() unmodified):
    stdout byte-identical to the reference; in table mode the same binary forwards to the reference's corrector."""
    r = subprocess.run([GPUTRIM, "-1", "-k", "51", "-b", "26", "-t", "2", g1_fq], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    assert hashlib.md5(r.stdout).hexdigest() == "f751f7b1aa28fd74b23194bc7f70158c"
    r = subprocess.run([GPUTRIM, "-1", "-k", "51", "-b", "26", "-L", "200000", g1_fq], capture_output=True, timeout=600)  # several batches
    assert hashlib.md5(r.stdout).hexdigest() == "f751f7b1aa28fd74b23194bc7f70158c"
    r = subprocess.run([GPUTRIM, "-k", "31", "-b", "26", "-t", "4", g1_fq], capture_output=True, timeout=600)
    assert r.returncode == 0 and hashlib.md5(r.stdout).hexdigest() == "06a4284e5010e34a1645d1d8015d6da9"
    # headers with comments and FASTA input: compare with the reference binary run on the same file
    ref = os.path.join(oracle.REF_DIR, "bfc-ref")
    fa = str(tmp_path / "c.fa")
    with open(g1_fq) as f, open(fa, "w") as g:
        for i, line in enumerate(f):
            if i >= 8000:
                break
            if i % 4 == 0:
                g.write(">" + line[1:].rstrip("\n") + (" comment %d x\n" % i if i % 8 == 0 else "\n"))
            elif i % 4 == 1:
                g.write(line)
    a = subprocess.run([ref, "-1", "-k", "31", "-b", "22", fa], capture_output=True, timeout=600)
    b = subprocess.run([GPUTRIM, "-1", "-k", "31", "-b", "22", fa], capture_output=True, timeout=600)
    assert a.returncode == 0 and b.returncode == 0 and len(a.stdout) > 0 and a.stdout == b.stdout


@needs_dropin
@pytest.mark.parametrize("planes", ["1", "0"])
@pytest.mark.parametrize("threads", ["1", "8"])
def test_dropin_exact_dump_md5(g1_fq, tmp_path, planes, threads):
    """`bfc -E -d` through the unmodified main(): with BFC_GPU_EXACT_DUMP=1 the dump file is byte-identical to the reference's -- with the batches
    handed to the GPU as bit planes (the default on one GPU: bfcg_count_batch_planes) and as byte streams (BFC_GPU_PLANES=0)."""
    dump = str(tmp_path / "g1.hash")
    r = subprocess.run([DROPIN, "-E", "-k", "31", "-b", "26", "-L", "300000", "-t", threads, "-d", dump, g1_fq], capture_output=True, timeout=600,
                       env=dict(os.environ, BFC_GPU_EXACT_DUMP="1", BFC_GPU_PLANES=planes))
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    assert oracle.md5_file(dump) == "d686549d10dd4c71243269013119784a"


REFBIN = os.path.join(oracle.REF_DIR, "bfc-ref")


@needs_dropin
@pytest.mark.usefixtures("ref_bin")
@pytest.mark.parametrize("seed", range(8))
def test_dropin_equals_reference_binary_on_damaged_files(tmp_path, seed):
    """End to end through the reference's own main(): damaged FASTQ / FASTA text (dropped, doubled, split, junk lines, CRLF, truncation)
    -> `bfc -E -d` with the GPU count path (serial parser with -t1, multi-threaded one with -t4) writes the very bytes the reference
    binary writes with -t1 (BFC_GPU_EXACT_DUMP=1), for two chunk sizes."""
    from test_ingest import _fastq, _mutate
    rng = np.random.default_rng(100 + seed + 100003 * int(os.environ.get("BFC_FUZZ_SEED_BASE", "0")))
    data = _fastq(rng, 3000, 20, 160, crlf=seed == 3) if seed % 2 == 0 else _fastq(rng, 1500, 20, 160) + b">fa x\nACGTTGCAACGTTTGACCA\nACGGGT\n" + _fastq(rng, 1500, 20, 160)
    fn = str(tmp_path / "d.fq")
    open(fn, "wb").write(_mutate(rng, data))
    env = dict(os.environ, BFC_GPU_EXACT_DUMP="1")
    for chunk in ("100000000", "50000"):
        ref_dump = str(tmp_path / "ref.hash")
        r = subprocess.run([REFBIN, "-E", "-k", "21", "-b", "24", "-t", "1", "-L", chunk, "-d", ref_dump, fn], capture_output=True, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-800:]
        for t in ("1", "4"):
            gpu_dump = str(tmp_path / ("gpu%s.hash" % t))
            g = subprocess.run([DROPIN, "-E", "-k", "21", "-b", "24", "-t", t, "-L", chunk, "-d", gpu_dump, fn], capture_output=True, timeout=600, env=env)
            assert g.returncode == 0, g.stderr.decode()[-800:]
            assert open(gpu_dump, "rb").read() == open(ref_dump, "rb").read(), (seed, chunk, t)
            # the progress lines (reads per batch) are the reference's too
            want = [l for l in r.stderr.decode().splitlines() if l.startswith("[M::bfc_count_cb] read")]
            got = [l for l in g.stderr.decode().splitlines() if l.startswith("[M::bfc_count_cb] read")]
            assert got == want, (seed, chunk, t)
            # ... and so are the `processed N sequences; # distinct k-mers: K` lines (count.c:113; time stamps aside): K after every chunk is
            # exact although the GPU path prints them without waiting for the chunk (bfcg_progress)
            import re
            pk = lambda txt: re.findall(r"processed (\d+) sequences; # distinct k-mers: (\d+)", txt)  # noqa: E731
            assert pk(g.stderr.decode()) == pk(r.stderr.decode()), (seed, chunk, t)
            assert len(pk(r.stderr.decode())) > 1 or chunk != "50000" or os.environ.get("BFC_FUZZ_SEED_BASE", "0") != "0", (seed, chunk, t)  # (other bases may truncate the text early)


@pytest.mark.usefixtures("dropin_bin", "gputrim_bin", "ref_bin")
@pytest.mark.parametrize("seed", range(6))
def test_binaries_equal_reference_on_damaged_gzip(tmp_path, seed):
    """A truncated or corrupted .gz: zlib hands the reference text until it meets the damage, and where exactly that is depends on how it is asked
    (kseq: 16 KiB a call).  The count pass (`bfc -E -d`, serial ingest and 4 threads with the parallel inflate forced on) dumps the reference's
    bytes, and `bfc -1` (both passes on the GPU, the second reads the file again) prints the reference's trimmed reads."""
    import gzip as gz_
    rng = np.random.default_rng(300 + seed)
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 30000)
    recs = []
    for r in range(4000):
        l = int(rng.integers(60, 151)); p = int(rng.integers(0, 30000 - l))
        q = rng.integers(33, 74, l).astype(np.uint8)
        recs.append(b"@r%d\n" % r + genome[p:p + l].tobytes() + b"\n+\n" + q.tobytes() + b"\n")
    z = bytearray(gz_.compress(b"".join(recs), 6))
    what = seed % 3
    if what == 0:
        z = z[:int(rng.integers(len(z) // 2, len(z) - 1))]
    elif what == 1:
        z[int(rng.integers(len(z) // 2, len(z) - 8))] ^= 1 << int(rng.integers(0, 8))
    else:
        z[len(z) - 6] ^= 0x20   # the member's CRC-32
    fn = str(tmp_path / "d.fq.gz")
    open(fn, "wb").write(bytes(z))
    env = dict(os.environ, BFC_GPU_EXACT_DUMP="1", BFC_INGEST_GZ_MIN="0", BFC_INGEST_GZ_CHUNK="15000")
    ref_dump = str(tmp_path / "ref.hash")
    r = subprocess.run([REFBIN, "-E", "-k", "21", "-b", "24", "-t", "1", "-L", "60000", "-d", ref_dump, fn], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    for t in ("1", "4"):
        gpu_dump = str(tmp_path / ("gpu%s.hash" % t))
        g = subprocess.run([DROPIN, "-E", "-k", "21", "-b", "24", "-t", t, "-L", "60000", "-d", gpu_dump, fn], capture_output=True, timeout=600, env=env)
        assert g.returncode == 0, g.stderr.decode()[-800:]
        assert open(gpu_dump, "rb").read() == open(ref_dump, "rb").read(), (seed, t)
    args = ["-1", "-k", "21", "-b", "24", "-t", "2", "-L", "60000", fn]
    r = subprocess.run([REFBIN] + args, capture_output=True, timeout=600)
    g = subprocess.run([GPUTRIM] + args, capture_output=True, timeout=600, env=env)
    assert r.returncode == 0 and g.returncode == 0, (r.stderr.decode()[-500:], g.stderr.decode()[-500:])
    assert g.stdout == r.stdout and len(r.stdout) > 50000, (seed, len(g.stdout), len(r.stdout))


@pytest.mark.usefixtures("gputrim_bin", "ref_bin")
@pytest.mark.parametrize("seed", range(8))
def test_gpu_trim_equals_reference_binary_on_damaged_files(tmp_path, seed):
    """`bfc -1` with both phases on the GPU vs the reference binary on damaged FASTQ / FASTA text: stdout (names, inherited comments,
    FASTA / FASTQ form, trimmed windows) byte for byte, and the `read N sequences` lines of both passes."""
    from test_ingest import _fastq, _mutate
    rng = np.random.default_rng(200 + seed + 100003 * int(os.environ.get("BFC_FUZZ_SEED_BASE", "0")))
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 20000)
    eol = b"\r\n" if seed == 5 else b"\n"
    recs = []
    for r in range(2500):  # reads of a small genome (8x): most k-mers are seen again, most reads survive the trim
        l = int(rng.integers(60, 161)); p = int(rng.integers(0, 20000 - l))
        s_ = genome[p:p + l].copy()
        if r % 17 == 0:
            s_[int(rng.integers(0, l))] = ord("N")
        q = rng.integers(33, 74, l).astype(np.uint8)
        if r % 3 == 0:
            q[0] = ord("@")
        recs.append(b"@r%d%s" % (r, b" cm:%d" % r if r % 4 else b"") + eol + s_.tobytes() + eol + b"+" + eol + q.tobytes() + eol)
    base = b"".join(recs)
    # reads drawn twice so that k-mers are seen and some reads survive the trim
    data = base if seed % 2 == 0 else base[:len(base) // 2] + b">fa only\nACGTTGCAACGTTTGACCAGGTACCATTGACGGTACCAGT\n" + base[len(base) // 2:]
    fn = str(tmp_path / "d.fq")
    open(fn, "wb").write(_mutate(rng, data))
    for extra in ([], ["-L", "40000"], ["-J"]):
        args = ["-1", "-k", "21", "-b", "24", "-t", "2"] + extra + [fn]
        r = subprocess.run([REFBIN] + args, capture_output=True, timeout=600)
        g = subprocess.run([GPUTRIM] + args, capture_output=True, timeout=600)
        assert r.returncode == 0 and g.returncode == 0, (r.stderr.decode()[-500:], g.stderr.decode()[-500:])
        assert g.stdout == r.stdout, (seed, extra, len(g.stdout), len(r.stdout))
        want = [l for l in r.stderr.decode().splitlines() if "] read " in l]
        got = [l for l in g.stderr.decode().splitlines() if "] read " in l]
        assert got == want, (seed, extra)
        assert len(r.stdout) > 100000 or os.environ.get("BFC_FUZZ_SEED_BASE", "0") != "0"  # the stock seeds keep most of the text; other bases may truncate early


def test_gpu_trim_pass_on_a_large_filter(gpu_lib, g1):
    """Filters of 4 GiB and more take the query kernel in which four lanes fetch a block together: same windows as the oracle (-b35)."""
    import ctypes as C
    rs, (seq, qual, off) = g1
    n = 3000
    seq, qual, off = seq[:n * rs.L], qual[:n * rs.L], off[:n + 1]
    k, b = 33, 35
    g = gpu_lib.GpuCounter(k, b, filter_mode=1, max_batch_pos=len(seq) + n + 64)
    s_seq, s_qual = gpu_lib.to_stream(seq, off), gpu_lib.to_stream(qual, off)
    g.count_host(s_seq, s_qual)
    bf = g.export_bloom(1)
    g.close()
    tr = gpu_lib.GpuTrimmer(k, bf, max_pos=len(s_seq) + 64, max_reads=n)
    start, end = tr.trim(s_seq, off + np.arange(n + 1, dtype=np.uint64), 0.9)
    L = oracle.lib()
    oc = oracle.Counter(k, b, filter_mode=1)
    oc.count(seq, qual, off)
    obf = L.orc_state_bf_high(oc.st)
    kept = 0
    for r in range(n):
        s = seq[int(off[r]):int(off[r + 1])]
        mx = L.orc_max_streak(k, obf, s.ctypes.data, len(s))
        a, e = C.c_int(), C.c_int()
        if L.orc_trim_decide(mx, k, len(s), 0.9, C.byref(a), C.byref(e)):
            assert (int(start[r]), int(end[r])) == (a.value, e.value), r
            kept += 1
        else:
            assert start[r] == -1, r
    assert kept > 500
    tr.close(); bf.close(); oc.close()


@needs_dropin
@pytest.mark.usefixtures("ref_bin")
def test_dropin_with_the_largest_filter(g1_fq, tmp_path):
    """`bfc -s 3g` (k=33, -b37: a 16 GiB filter, batches of 16 reference chunks) on a small file: the dump equals the reference binary's,
    and the library sizes its buffers by the file, not by the 1.6 G-position batch the filter would allow."""
    env = dict(os.environ, BFC_GPU_EXACT_DUMP="1", BFC_GPU_TIMING="1")
    ref_dump, gpu_dump = str(tmp_path / "ref.hash"), str(tmp_path / "gpu.hash")
    r = subprocess.run([REFBIN, "-E", "-s", "3g", "-t", "1", "-d", ref_dump, g1_fq], capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-500:]
    g = subprocess.run([DROPIN, "-E", "-s", "3g", "-t", "4", "-d", gpu_dump, g1_fq], capture_output=True, timeout=900, env=env)
    assert g.returncode == 0, g.stderr.decode()[-800:]
    assert open(gpu_dump, "rb").read() == open(ref_dump, "rb").read()
    import re
    m = re.search(rb"buffers for (\d+) positions per batch", g.stderr)
    assert m and int(m.group(1)) < 4 * os.path.getsize(g1_fq)


@needs_dropin
@pytest.mark.usefixtures("gputrim_bin")
@pytest.mark.parametrize("devices", ["0,0", "0,0,0,0"])
def test_dropin_on_several_gpus(g1_fq, tmp_path, devices):
    """BFC_GPU_DEVICES: bfc_count itself spreads the file over the GPUs (bfcg_group_*: stage A, exchange, stage B in C) and the trim
    pass of `bfc -1` shards every batch's reads over them; ranks emulated on the one device here.  The reference's unmodified main()
    writes the reference's bytes: byte-identical dump with order stamps, corrected reads, trimmed reads."""
    env = dict(os.environ, BFC_GPU_DEVICES=devices, BFC_GPU_EXACT_DUMP="1")
    dump = str(tmp_path / "g1.hash")
    r = subprocess.run([DROPIN, "-E", "-k", "31", "-b", "26", "-L", "300000", "-d", dump, g1_fq], capture_output=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    assert oracle.md5_file(dump) == "d686549d10dd4c71243269013119784a"            # `bfc -t1 -E -d`
    env.pop("BFC_GPU_EXACT_DUMP")
    r = subprocess.run([GPUTRIM, "-k", "31", "-b", "26", "-t", "4", g1_fq], capture_output=True, timeout=600, env=env)
    assert r.returncode == 0 and hashlib.md5(r.stdout).hexdigest() == "06a4284e5010e34a1645d1d8015d6da9"   # full pipeline
    for extra in ([], ["-L", "200000"]):
        r = subprocess.run([GPUTRIM, "-1", "-k", "51", "-b", "26", "-t", "2"] + extra + [g1_fq], capture_output=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr.decode()[-1500:]
        assert hashlib.md5(r.stdout).hexdigest() == "f751f7b1aa28fd74b23194bc7f70158c"                     # `bfc -1`


@needs_dropin
@pytest.mark.usefixtures("ref_bin")
def test_config_c1_dump_equals_reference_binary(tmp_path):
    """BASELINE.json configs[0] end to end: bfcgen's c1 FASTQ (E. coli-sized genome at 1x: 30 666 reads) through the reference's unmodified
    main() on this library, `-E -k31 -d` with the default 1 GiB filter -- the dump is byte-identical to `bfc-ref -t1 -E -k31 -d` run here AND to
    the md5 the build container recorded from the reference (tests/golden/baseline.json[c1]; count.c:127-157, htab.c:129-149)."""
    import json
    e = {x["name"]: x for x in json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "baseline.json")))}["c1"]
    fq = str(tmp_path / "c1.fq")
    gen.ReadSet(**e["gen"]).fastq(fq)
    assert oracle.md5_file(fq) == e["fastq_md5"]
    ref_dump, gpu_dump = str(tmp_path / "ref.hash"), str(tmp_path / "gpu.hash")
    r = subprocess.run([REFBIN, "-t", "1", "-E", "-k", "31", "-d", ref_dump, fq], capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-500:]
    assert oracle.md5_file(ref_dump) == e["ref_dump_md5"]
    for t in ("1", "8"):
        g = subprocess.run([DROPIN, "-t", t, "-E", "-k", "31", "-d", gpu_dump, fq], capture_output=True, timeout=900, env=dict(os.environ, BFC_GPU_EXACT_DUMP="1"))
        assert g.returncode == 0, g.stderr.decode()[-800:]
        assert oracle.md5_file(gpu_dump) == e["ref_dump_md5"], t
        assert b"# distinct k-mers: %d" % e["distinct"] in g.stderr
    # and without order stamps: the same table (L1), laid out by this library
    g = subprocess.run([DROPIN, "-t", "8", "-E", "-k", "31", "-d", gpu_dump, fq], capture_output=True, timeout=900)
    assert g.returncode == 0
    k, l_pre, sizes, slots = oracle.parse_dump(gpu_dump)
    assert (k, l_pre) == (31, 20) and oracle.l1_digest(sizes, slots) == e["l1_digest"]


@needs_dropin
@pytest.mark.usefixtures("ref_bin")
@pytest.mark.parametrize("devices", ["", "0,0", "0,0,0,0"])
def test_mixed_fasta_fastq_at_q94(tmp_path, devices):
    """A file that mixes FASTQ and FASTA records, counted with -q 94: no FASTQ base is high quality any more (33 + 94 = 127 > '~'), every base of
    a FASTA record still is (qual == NULL, count.c:85).  On one GPU and on several (ranks emulated on the one device) a mixed batch is submitted as
    its homogeneous runs, so the dump equals the reference binary's byte for byte."""
    from test_ingest import _fastq
    rng = np.random.default_rng(94)
    fa = b"".join(b">fa%d\n%s\n" % (i, bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 120))) for i in range(400))
    data = _fastq(rng, 1500, 60, 160) + fa + _fastq(rng, 800, 60, 160) + fa[: len(fa) // 2] + _fastq(rng, 700, 60, 160)
    fn = str(tmp_path / "mixed.fq")
    open(fn, "wb").write(data)
    env = dict(os.environ, BFC_GPU_EXACT_DUMP="1")
    if devices:
        env["BFC_GPU_DEVICES"] = devices
    for chunk in ("100000000", "60000", "60000:bytes"):
        if chunk.endswith(":bytes"):  # (one GPU hands its batches over as bit planes by default; once more as byte streams)
            if devices:
                continue
            chunk = chunk.split(":")[0]; env["BFC_GPU_PLANES"] = "0"
        ref_dump, gpu_dump = str(tmp_path / "ref.hash"), str(tmp_path / "gpu.hash")
        r = subprocess.run([REFBIN, "-E", "-k", "21", "-b", "26", "-q", "94", "-t", "1", "-L", chunk, "-d", ref_dump, fn], capture_output=True, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-800:]
        g = subprocess.run([DROPIN, "-E", "-k", "21", "-b", "26", "-q", "94", "-t", "4", "-L", chunk, "-d", gpu_dump, fn], capture_output=True, timeout=600, env=env)
        assert g.returncode == 0, g.stderr.decode()[-800:]
        assert open(gpu_dump, "rb").read() == open(ref_dump, "rb").read(), (devices, chunk)
        import re
        pk = lambda txt: re.findall(r"processed (\d+) sequences; # distinct k-mers: (\d+)", txt)  # noqa: E731
        assert pk(g.stderr.decode()) == pk(r.stderr.decode()), (devices, chunk)  # exact per chunk, printed without draining (bfcg_progress / bfcg_group_progress)
        k, l_pre, sizes, slots = oracle.parse_dump(ref_dump)
        high = (slots >> np.uint64(8)) & np.uint64(0x3f)
        assert 0 < int((high > 0).sum()) < len(slots)  # only the FASTA records' k-mers have high-quality occurrences


@pytest.mark.usefixtures("gputrim_bin", "ref_bin")
def test_trim_pass_on_several_gpus_with_reads_sorted_by_length(tmp_path):
    """`bfc -1` on two devices, reads sorted by length (2 000 down to 60 bases): the trim pass deals a batch to the devices by POSITIONS (cut at
    the nearest read boundary) -- dealt by read count, the first device's share would exceed the context it was sized for -- and prints the
    reference's bytes (correct.c:478-497,557-569,595-611)."""
    rng = np.random.default_rng(7)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    genome = rng.choice(acgt, 60_000)
    lens = np.linspace(2000, 60, 1100).astype(int)
    recs = []
    for i, l in enumerate(lens):
        p = int(rng.integers(0, len(genome) - l))
        s = genome[p:p + l].copy()
        e = rng.random(l) < 0.01
        s[e] = acgt[rng.integers(0, 4, int(e.sum()))]
        recs.append(b"@r%d\n%s\n+\n%s\n" % (i, s.tobytes(), rng.integers(40, 74, l).astype(np.uint8).tobytes()))
    fn = str(tmp_path / "sorted.fq")
    open(fn, "wb").write(b"".join(recs))
    r = subprocess.run([REFBIN, "-1", "-k", "21", "-b", "24", "-t", "1", fn], capture_output=True, timeout=600)
    assert r.returncode == 0 and len(r.stdout) > 10_000, r.stderr.decode()[-500:]
    for devices in ("0", "0,0", "0,0,0,0"):
        g = subprocess.run([GPUTRIM, "-1", "-k", "21", "-b", "24", "-t", "2", fn], capture_output=True, timeout=600,
                           env=dict(os.environ, BFC_GPU_DEVICES=devices, BFC_GPU_BATCH="200000"))
        assert g.returncode == 0, (devices, g.stderr.decode()[-800:])
        assert g.stdout == r.stdout, devices
