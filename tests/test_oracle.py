"""CPU tests (-m "not gpu"): the oracle (oracle/bfc_oracle.c, a restatement of count.c + bbf.c + htab.c + kmer.h + khash
behaviour) is pinned against golden vectors generated from the reference itself (tests/golden/*.json, made by
tests/golden/make_goldens.py from oracle/_ref) and, when oracle/_ref/libbfcref.so is present, against live calls
into the reference."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import oracle
from bfc_amd import gen

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat.json")))
FIX = json.load(open(os.path.join(HERE, "golden", "fixtures.json")))
u64p = C.POINTER(C.c_uint64)


@pytest.mark.parametrize("e", KAT, ids=lambda e: "k%d_b%d_%s" % (e["k"], e["b"], e["kmer"][:6]))
def test_kat_single_kmer(e):
    """k-mer planes (kmer.h:10-17), hash (kmer.h:30-40,79-88), bloom addressing (bbf.c:27-41), key packing and the
    saturating slot update (htab.c:45-58,73-79) for single k-mers, incl. k=63, even k, the k>32 and lossy k>=38 branches."""
    L = oracle.lib()
    k = e["k"]
    x = (C.c_uint64 * 4)(0, 0, 0, 0)
    for ch in e["kmer"]:
        L.orc_kmer_push(k, x, "ACGT".index(ch))
    assert [int(v) for v in x] == e["x"]
    y = (C.c_uint64 * 2)()
    assert int(L.orc_kmer_hash(k, x, y)) == e["hash"]
    assert (int(y[0]), int(y[1])) == (e["y0"], e["y1"])
    assert int(L.orc_hash_from_y(k, y)) == e["hash"]
    for bb, bits in e["bloom_bits"].items():
        bf = L.orc_bf_new(int(bb), 4)
        assert L.orc_bf_insert(bf, e["hash"]) == 0 and L.orc_bf_insert(bf, e["hash"]) == 4 and L.orc_bf_get(bf, e["hash"]) == 4
        raw = np.ctypeslib.as_array(C.cast(L.orc_bf_bits(bf), C.POINTER(C.c_uint8)), shape=(1 << (int(bb) - 3),))
        nz = np.nonzero(raw)[0]
        assert sorted(int(i) * 8 + j for i in nz for j in range(8) if raw[i] >> j & 1) == bits
        assert all((p & 511) >= 8 for p in bits)  # the lock byte of a block is never used (bbf.c:37)
        L.orc_bf_free(bf)
    ch = L.orc_ch_new(k, 20)
    assert L.orc_ch_lpre(ch) == e["l_pre"]
    L.orc_ch_insert(ch, y, 1)
    L.orc_ch_insert(ch, y, 0)
    sizes = np.zeros(1 << e["l_pre"], dtype=np.uint32)
    slots = np.zeros(1, dtype=np.uint64)
    assert L.orc_ch_export(ch, sizes.ctypes.data_as(C.POINTER(C.c_uint32)), slots.ctypes.data_as(u64p)) == 1
    assert int(np.nonzero(sizes)[0][0]) == e["sub"] and int(slots[0]) == e["slot_after_high_then_low"]
    assert L.orc_ch_get(ch, y) == (1 << 8 | 2)
    L.orc_ch_free(ch)


def test_generator_matches_fixture_definition(tmp_path):
    """bfcgen: byte-identical FASTQ (md5) and the SoA form agree."""
    rs = gen.fixture("g1")
    fn = str(tmp_path / "g1.fq")
    rs.fastq(fn)
    assert oracle.md5_file(fn) == FIX["binary_runs"]["fastq_md5"] == "7e17d87fc623a6f7171a607d68dde51a"
    seq, qual, off = rs.reads()
    lines = open(fn, "rb").read().split(b"\n")
    assert lines[1] == seq[:150].tobytes() and lines[3] == qual[:150].tobytes()
    assert lines[4 * 777 + 1] == seq[777 * 150:778 * 150].tobytes()
    s2, q2, _ = rs.reads(700, 800)  # random access
    assert np.array_equal(s2, seq[700 * 150:800 * 150]) and np.array_equal(q2, qual[700 * 150:800 * 150])


def _check_fixture(e, seq, qual, off):
    c = oracle.Counter(e["k"], e["b"], filter_mode=e["filter_mode"])
    c.count(seq, qual, off)
    st = c.stats()
    for key in ("n_kmers", "n_high", "n_seen", "hash_xor"):
        assert st[key] == e[key], key
    assert c.bloom_checksums() == (e["bf_popcount"], e["bf_fnv1a64"])
    if e["filter_mode"]:
        assert c.bloom_checksums(high=True) == (e["bf_high_popcount"], e["bf_high_fnv1a64"])
    else:
        mode, cnt, high = c.table_hist()
        assert c.table_count() == e["distinct"] and mode == e["hist_mode"]
        assert [int(v) for v in cnt[1:5]] == e["cnt_1_4"] and [int(v) for v in high[0:3]] == e["high_0_2"]
        sizes, slots = c.export()
        assert oracle.l1_digest(sizes, slots) == e["l1_digest"]
    return c


@pytest.mark.parametrize("e", [e for e in FIX["fixtures"] if e["fixture"] == "g1"], ids=lambda e: "k%d_b%d_f%d" % (e["k"], e["b"], e["filter_mode"]))
def test_fixture_g1(e, g1, tmp_path):
    """`bfc -t1` checksums on g1: totals, bloom popcount/FNV (L0), table contents (L1) and the byte-exact -d dump (L2:
    the khash growth / probing / kick-out rehash behaviour of SURVEY A.7)."""
    rs, (seq, qual, off) = g1
    c = _check_fixture(e, seq, qual, off)
    if not e["filter_mode"]:
        fn = str(tmp_path / "d.hash")
        assert c.dump(fn) == 0 and oracle.md5_file(fn) == e["dump_md5"]
        k, l_pre, sizes, slots = oracle.parse_dump(fn)
        assert (k, l_pre) == (e["k"], e["l_pre"]) and oracle.l1_digest(sizes, slots) == e["l1_digest"]
    c.close()


@pytest.mark.parametrize("e", [e for e in FIX["fixtures"] if e["fixture"] == "g42"], ids=lambda e: "k%d_b%d_f%d" % (e["k"], e["b"], e["filter_mode"]))
def test_fixture_g42(e, g42):
    rs, (seq, qual, off) = g42
    _check_fixture(e, seq, qual, off).close()


def test_batching_and_fasta_invariants(g1):
    """Splitting the input into chunks never changes results (SURVEY B.3 input-path invariants); FASTA input makes
    every k-mer high quality (count.c:85), golden L1 digest from the reference."""
    rs, (seq, qual, off) = g1
    c = oracle.Counter(31, 26)
    step = 997
    for i in range(0, rs.n_reads, step):
        j = min(rs.n_reads, i + step)
        c.count(seq[int(off[i]):int(off[j])], qual[int(off[i]):int(off[j])], off[i:j + 1] - off[i])
    assert oracle.l1_digest(*c.export()) == "237be10261b07ef0677f8136b0a327b6"
    c.close()
    c = oracle.Counter(31, 26)
    c.count(seq, None, off)
    sizes, slots = c.export()
    assert oracle.l1_digest(sizes, slots) == "4e4705f65c9bcc880b6e28a3453cc4fc"
    assert np.array_equal((slots >> np.uint64(8)) & np.uint64(0x3f), np.minimum(slots & np.uint64(0xff), np.uint64(63)))
    c.close()


def test_trim_pass_matches_reference_binary(g1):
    """max_streak + keep/trim rule (correct.c:478-497,557-569) on the second bloom filter reproduces the stdout of
    `bfc -1 -k51 -b26 -t1 g1.fq` (md5 golden from the reference binary)."""
    rs, (seq, qual, off) = g1
    L = oracle.lib()
    c = oracle.Counter(51, 26, filter_mode=1)
    c.count(seq, qual, off)
    bf = L.orc_state_bf_high(c.st)
    out = []
    kept = bases = 0
    for r in range(rs.n_reads):
        s = seq[int(off[r]):int(off[r + 1])]
        q = qual[int(off[r]):int(off[r + 1])]
        mx = L.orc_max_streak(51, bf, s.ctypes.data, len(s))
        a, b = C.c_int(), C.c_int()
        if L.orc_trim_decide(mx, 51, len(s), 0.9, C.byref(a), C.byref(b)):
            out.append(b"@r%d\n%s\n+\n%s\n" % (r, s[a.value:b.value].tobytes(), q[a.value:b.value].tobytes()))  # correct.c:605-611
            kept += 1
            bases += b.value - a.value
    assert (kept, bases) == (1763, 259433)
    assert hashlib.md5(b"".join(out)).hexdigest() == FIX["binary_runs"]["trim_k51_b26"]["stdout_md5"] == "f751f7b1aa28fd74b23194bc7f70158c"
    c.close()


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref/libbfcref.so not built (needs /root/reference)")
@pytest.mark.parametrize("k,b,nh,fm", [(31, 20, 4, 0), (33, 22, 3, 0), (47, 20, 5, 0), (63, 24, 4, 0), (22, 18, 4, 0), (51, 20, 4, 1), (37, 16, 6, 0)])
def test_oracle_vs_live_reference(k, b, nh, fm):
    """Random reads with Ns, lower case and odd lengths: per-k-mer trace (hash, y, is_high, seen), bitmaps, table and dump."""
    rng = np.random.default_rng(k * 1000 + b)
    n = 400
    lens = rng.integers(1, 260, n)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    genome = rng.integers(0, 4, 3000)
    seq = np.empty(int(off[-1]), dtype=np.uint8)
    for r in range(n):
        p = rng.integers(0, 3000 - 260)
        seq[int(off[r]):int(off[r + 1])] = np.frombuffer(b"ACGT", dtype=np.uint8)[genome[p:p + lens[r]]]
    seq[rng.integers(0, len(seq), 60)] = ord("N")
    seq[rng.integers(0, len(seq), 60)] |= 0x20
    qual = rng.integers(33, 74, len(seq)).astype(np.uint8)
    o = oracle.Counter(k, b, n_hashes=nh, filter_mode=fm)
    r = oracle.Counter(k, b, n_hashes=nh, filter_mode=fm, impl="ref")
    to, tr = o.count(seq, qual, off, trace=True), r.count(seq, qual, off, trace=True)
    assert np.array_equal(to, tr)
    assert o.stats() == r.stats()
    assert np.array_equal(o.bloom_bytes(), r.bloom_bytes())
    if fm:
        assert np.array_equal(o.bloom_bytes(True), r.bloom_bytes(True))
    else:
        so, sr = o.export(), r.export()
        assert np.array_equal(so[0], sr[0]) and np.array_equal(so[1], sr[1])
        for row in to[:50]:
            assert o.table_get(int(row[1]), int(row[2])) == r.table_get(int(row[1]), int(row[2]))
    o.close(); r.close()


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref/libbfcref.so not built (needs /root/reference)")
@pytest.mark.parametrize("seed", range(12))
def test_oracle_vs_live_reference_random(seed):
    """The oracle pinned on the draws the GPU fuzz uses (tests/test_gpu_fuzz.py::_draw: random k, filter size, hashes, l_pre, quality
    threshold, filter mode, read lengths, arbitrary bytes in sequence and quality; BFC_FUZZ_SEED_BASE for other draws): statistics, both
    filters and the table equal the reference's own functions called in file order."""
    from test_gpu_fuzz import _draw
    prm, seq, qual, off, cuts, kw = _draw(30000 + seed)
    o = oracle.Counter(prm["k"], prm["b"], q=prm["q"], n_hashes=prm["nh"], l_pre=prm["l_pre"], filter_mode=prm["fm"])
    r = oracle.Counter(prm["k"], prm["b"], q=prm["q"], n_hashes=prm["nh"], l_pre=prm["l_pre"], filter_mode=prm["fm"], impl="ref")
    o.count(seq, qual, off); r.count(seq, qual, off)
    assert o.stats() == r.stats(), prm
    assert np.array_equal(o.bloom_bytes(), r.bloom_bytes()), prm
    if prm["fm"]:
        assert np.array_equal(o.bloom_bytes(True), r.bloom_bytes(True)), prm
    else:
        so, sr = o.export(), r.export()
        assert np.array_equal(so[0], sr[0]) and np.array_equal(so[1], sr[1]), prm
    o.close(); r.close()


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref/libbfcref.so not built (needs /root/reference)")
@pytest.mark.parametrize("q", [-200, -40, 0, 20, 93, 94, 95, 127])
def test_quality_threshold_on_every_byte_value(q):
    """count.c:85 compares `s->qual[i] - 33 >= q` on a (signed) char: quality bytes from 0 to 255 and thresholds from far below to far
    above the Phred range give the same high-quality flags in the oracle and in the reference."""
    rng = np.random.default_rng(q + 1000)
    n, L = 300, 80
    genome = rng.integers(0, 4, 4000)
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    seq = np.concatenate([np.frombuffer(b"ACGT", dtype=np.uint8)[genome[p:p + L]] for p in rng.integers(0, 4000 - L, n)])
    qual = rng.integers(33, 75, n * L).astype(np.uint8)
    odd = rng.random(n * L) < 0.03
    qual[odd] = rng.integers(0, 256, int(odd.sum())).astype(np.uint8)  # a few arbitrary bytes among plausible qualities
    o = oracle.Counter(21, 20, q=q)
    r = oracle.Counter(21, 20, q=q, impl="ref")
    to, tr = o.count(seq, qual, off, trace=True), r.count(seq, qual, off, trace=True)
    assert np.array_equal(to, tr)
    assert o.stats() == r.stats()
    if -40 <= q <= 0:
        assert 0 < o.stats()["n_high"] < o.stats()["n_kmers"]
    o.close(); r.close()


def test_config_c1_oracle_equals_reference_golden(tmp_path):
    """BASELINE.json configs[0] (E. coli-sized genome at 1x, k=31, the default -b33: the reference's own CPU-runnable plumbing case): the C
    restatement over the c1 read set equals what the reference's functions gave for it (tests/golden/baseline.json[c1], generated by
    make_baseline_goldens.py from count.c:127-157 at -t1) -- totals, filter checksums, both histograms, L1 digest -- and its byte-exact dump has
    the md5 of the reference BINARY's `bfc -t1 -E -d` on bfcgen's FASTQ of the same reads (htab.c:129-149)."""
    base = {e["name"]: e for e in json.load(open(os.path.join(HERE, "golden", "baseline.json")))}
    e = base["c1"]
    rs = gen.ReadSet(**e["gen"])
    assert rs.n_reads == e["n_reads"]
    fq = str(tmp_path / "c1.fq")
    rs.fastq(fq)
    assert oracle.md5_file(fq) == e["fastq_md5"]
    seq, qual, off = rs.reads()
    c = oracle.Counter(e["k"], e["b"])
    c.count(seq, qual, off)
    st = c.stats()
    assert (st["n_kmers"], st["n_high"], st["n_seen"]) == (e["n_kmers"], e["n_high"], e["n_seen"])
    pop, fnv = c.bloom_checksums()
    assert (pop, "%016x" % fnv) == (e["bf_popcount"], e["bf_fnv1a64"])
    mode, cnt, high = c.table_hist()
    assert c.table_count() == e["distinct"] and int(mode) == e["hist_mode"]
    assert [int(v) for v in cnt] == e["cnt"] and [int(v) for v in high] == e["high"]
    dump = str(tmp_path / "c1.hash")
    c.dump(dump)
    assert oracle.md5_file(dump) == e["ref_dump_md5"]
    c.close()


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref/libbfcref.so not built")
@pytest.mark.parametrize("k,b,fm", [(31, 24, 0), (51, 26, 1), (55, 24, 0)])
def test_block_partitioned_harness_equals_the_sequential_one(g1, k, b, fm):
    """ref_count_batch_blocks (oracle/ref_shim.c: n threads, thread p owns the bloom blocks of residue class p and walks all reads in file
    order) gives what the sequential harness (= `bfc -t1`) gives: totals, both filters byte for byte, the table as sorted slot lists.  The
    eighth-of-human goldens of round 6 (c5e_trim, c4e_k55) are made with it."""
    rs, (seq, qual, off) = g1
    a = oracle.Counter(k, b, filter_mode=fm, impl="ref")
    a.count(seq, qual, off)
    for parts in (3, 8):
        m = oracle.Counter(k, b, filter_mode=fm, impl="ref")
        h = rs.n_reads // 3  # two calls: the running k-mer index of hash_xor carries over
        m.count_blocks(seq[:int(off[h])], qual[:int(off[h])], off[:h + 1], parts)
        m.count_blocks(seq[int(off[h]):], qual[int(off[h]):], off[h:] - off[h], parts)
        assert m.stats() == a.stats()
        assert np.array_equal(m.bloom_view(), a.bloom_view())
        if fm:
            assert np.array_equal(m.bloom_view(True), a.bloom_view(True))
        else:
            (sa, la), (sm, lm) = a.export(), m.export()
            assert oracle.l1_digest(sa, la) == oracle.l1_digest(sm, lm)
        m.close()
    a.close()


@pytest.mark.skipif(not (oracle.have_ref() and oracle.have_ref_ec()), reason="oracle/_ref not built")
def test_reference_trim_pass_through_worker_ec(g1):
    """oracle.ref_trim = the reference's worker_ec (max_streak + keep rule, correct.c:478-497,557-567) driven by its own kt_for: the windows
    equal the restatement's, their count and bases the reference binary's (`bfc -1 -k51 -b26 -t1`), on one thread and on four; a read of
    300 bases takes the path that asks max_streak for the start."""
    rs, (seq, qual, off) = g1
    L = oracle.lib()
    r = oracle.Counter(51, 26, filter_mode=1, impl="ref")
    r.count(seq, qual, off)
    o = oracle.Counter(51, 26, filter_mode=1)
    o.count(seq, qual, off)
    obf = L.orc_state_bf_high(o.st)
    long_seq = np.concatenate([seq[:150], seq[:150]])
    seq2 = np.concatenate([seq, long_seq]); off2 = np.concatenate([off, [off[-1] + 300]]).astype(np.uint64)
    exp_s, exp_e = [], []
    for i in range(len(off2) - 1):
        s = seq2[int(off2[i]):int(off2[i + 1])]
        mx = L.orc_max_streak(51, obf, s.ctypes.data, len(s))
        a, b = C.c_int(), C.c_int()
        kept = L.orc_trim_decide(mx, 51, len(s), 0.9, C.byref(a), C.byref(b))
        exp_s.append(a.value if kept else -1); exp_e.append(b.value if kept else -1)
    for nt in (1, 4):
        st, en = oracle.ref_trim(r._bf(True), 51, seq2, off2, 0.9, nt)
        assert st.tolist() == exp_s and en.tolist() == exp_e
    assert (int((st[:-1] >= 0).sum()), int((en[:-1] - st[:-1])[st[:-1] >= 0].sum())) == (1763, 259433)
    r.close(); o.close()
