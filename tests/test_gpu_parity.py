"""GPU parity tests (-m gpu): the HIP path, called through the C ABI (libbfc_gpu.so), against the
CPU oracle on the same seeded inputs.  Bit-exact is the bar: this is integer/byte work.

Parity levels:  L0 bloom bitmap memcmp-equal;  L1 per sub-table sorted slot lists
equal (=> every bfc_ch_get / bfc_ch_hist / bfc_ch_count answer equal).
"""
import os

import numpy as np
import pytest

import oracle
from bfc_amd import gen

pytestmark = pytest.mark.gpu

GOLD_G1 = {  # SURVEY B.3, captured from `bfc -t1`
    (31, 26): dict(n_kmers=798319, n_high=584697, n_seen=486855, pop=1234155, fnv=0x3d61f9259600f794, distinct=99561, mode=4,
                   l1="237be10261b07ef0677f8136b0a327b6"),
    (33, 30): dict(n_kmers=784910, n_high=563486, n_seen=465683, pop=1276151, fnv=0x9df9ca3dbf6002bb, distinct=99119, mode=4,
                   l1="896ce4092ccc51498b446e7d7775f10c"),
}


def _trace_to_positions(trace, seq, off, k, L_plus_sep=None):
    """Oracle trace (one row per k-mer, file order) -> per-position arrays on the separator-delimited stream."""
    lib = oracle.lib()
    n_reads = len(off) - 1
    n_pos = len(seq) + n_reads
    y0 = np.zeros(n_pos, dtype=np.uint64); y1 = np.zeros(n_pos, dtype=np.uint64); fl = np.zeros(n_pos, dtype=np.uint8)
    # end positions of k-mers in the stream: recompute validity per read on the CPU side
    codes = np.full(256, 4, dtype=np.uint8)
    for ch, c in zip(b"ACGTacgt", [0, 1, 2, 3, 0, 1, 2, 3]):
        codes[ch] = c
    row = 0
    for r in range(n_reads):
        s = seq[int(off[r]):int(off[r + 1])]
        valid = codes[s] < 4
        run = 0
        base = int(off[r]) + r
        for i in range(len(s)):
            run = run + 1 if valid[i] else 0
            if run >= k:
                y0[base + i] = trace[row, 1]; y1[base + i] = trace[row, 2]; fl[base + i] = 1 | (int(trace[row, 3]) << 1)
                row += 1
    assert row == len(trace)
    return y0, y1, fl  # fl: bit0 k-mer, bit1 is_high, bit2 seen


@pytest.mark.parametrize("k", [21, 31, 32, 33, 47, 51, 63])
def test_k1_hash_positions(gpu_lib, g1, k):
    """K1 (window k-mer extraction + strand-canonical hash + quality flag) vs count.c:72-89 / kmer.h:79-88."""
    rs, (seq, qual, off) = g1
    n = 300
    seq, qual, off = seq[:n * rs.L].copy(), qual[:n * rs.L].copy(), off[:n + 1]
    seq[5] = ord("n"); seq[rs.L * 3 + 40] = ord("a"); seq[rs.L * 3 + 41] = ord("t")  # lower case and a second break
    oc = oracle.Counter(k, 26)
    tr = oc.count(seq, qual, off, trace=True)
    y0, y1, fl = _trace_to_positions(tr, seq, off, k)
    g = gpu_lib.GpuCounter(k, 26, max_batch_pos=1 << 20)
    s_seq, s_qual = gpu_lib.to_stream(seq, off), gpu_lib.to_stream(qual, off)
    out = g.hash_positions(s_seq, s_qual)
    assert np.array_equal(out[:, 2] & 1, fl & 1), "k-mer positions differ"
    assert np.array_equal(out[:, 0], y0) and np.array_equal(out[:, 1], y1), "hash words differ"
    assert np.array_equal((out[:, 2] >> 1) & 1, (fl >> 1) & 1), "is_high differs"
    # FASTA: no qualities => every k-mer is high quality (count.c:85)
    out2 = g.hash_positions(s_seq, None)
    assert np.array_equal(out2[:, 2] & 1, out2[:, 2] >> 1)
    g.close()


def _gpu_count(gpu_lib, k, b, seq, qual, off, n_batches=1, **kw):
    n_reads = len(off) - 1
    L = int(off[1] - off[0])
    per = (n_reads + n_batches - 1) // n_batches
    g = gpu_lib.GpuCounter(k, b, max_batch_pos=per * (L + 1) + 64, **kw)
    for i in range(0, n_reads, per):
        j = min(n_reads, i + per)
        o = off[i:j + 1] - off[i]
        s = gpu_lib.to_stream(seq[int(off[i]):int(off[j])], o)
        q = gpu_lib.to_stream(qual[int(off[i]):int(off[j])], o) if qual is not None else None
        g.count_host(s, q)
    return g


@pytest.mark.parametrize("k,b", [(31, 26), (33, 30)])
@pytest.mark.parametrize("n_batches", [1, 7])
def test_g1_goldens(gpu_lib, g1, k, b, n_batches):
    """Fixture g1 against the goldens captured from the compiled reference (`bfc -t1`), any batching."""
    rs, (seq, qual, off) = g1
    gold = GOLD_G1[(k, b)]
    g = _gpu_count(gpu_lib, k, b, seq, qual, off, n_batches)
    st = g.stats()
    assert st["n_kmers"] == gold["n_kmers"] and st["n_high"] == gold["n_high"]
    bits = g.bloom_bytes()
    assert int(oracle.lib().orc_popcount_bytes(bits.ctypes.data, len(bits))) == gold["pop"]
    assert int(oracle.lib().orc_fnv1a64(bits.ctypes.data, len(bits))) == gold["fnv"]   # L0
    assert st["n_seen"] == gold["n_seen"]
    t = g.export_table()
    assert t.count() == gold["distinct"] == st["n_keys"]
    mode, cnt, high = t.hist()
    assert mode == gold["mode"]
    sizes, slots = t.export_sorted()
    assert oracle.l1_digest(sizes, slots) == gold["l1"]                                    # L1
    g.close()


@pytest.mark.parametrize("k,b,nh", [(31, 26, 4), (21, 22, 4), (33, 24, 3), (47, 26, 5), (51, 26, 4), (63, 28, 4), (32, 25, 4)])
def test_vs_oracle_full(gpu_lib, g1, k, b, nh):
    """Bloom bitmap (L0), per-k-mer seen flags, and the whole table (L1) vs the oracle, incl. the k>32 key
    branch, the lossy k>=38 keys, l_pre clamping (k>=37 -> 24), even k, other n_hashes."""
    rs, (seq, qual, off) = g1
    n = 3000
    seq, qual, off = seq[:n * rs.L], qual[:n * rs.L], off[:n + 1]
    oc = oracle.Counter(k, b, n_hashes=nh)
    tr = oc.count(seq, qual, off, trace=True)
    g = _gpu_count(gpu_lib, k, b, seq, qual, off, 1, n_hashes=nh, debug_seen=True)
    assert np.array_equal(g.bloom_bytes(), oc.bloom_bytes())
    fl = g.seen_flags(len(seq) + n)
    assert np.array_equal(fl[fl > 0] == 2, (tr[:, 3] >> 1) & 1 == 1), "seen flags differ from the sequential semantics"
    ost = oc.stats(); st = g.stats()
    assert (st["n_kmers"], st["n_high"], st["n_seen"]) == (ost["n_kmers"], ost["n_high"], ost["n_seen"])
    t = g.export_table()
    sizes, slots = t.export_sorted()
    osz, osl = oc.export()
    assert np.array_equal(sizes, osz) and np.array_equal(slots, osl)
    omode, ocnt, ohigh = oc.table_hist()
    mode, cnt, high = t.hist()
    assert mode == omode and np.array_equal(cnt, ocnt) and np.array_equal(high, ohigh)
    # point queries through bfc_ch_get (htab.c:84-92): present and absent keys
    for row in tr[:200]:
        assert t.get(int(row[1]), int(row[2])) == oc.table_get(int(row[1]), int(row[2]))
    assert t.get(1, 2) == oc.table_get(1, 2)
    g.close()


def test_tiny_bloom_forces_slow_path(gpu_lib, g1):
    """A 2^14-bit bloom filter: one region, thousands of k-mers per block => LDS first-setter table overflows
    and the HBM-pool path runs; results must not change."""
    rs, (seq, qual, off) = g1
    n = 2000
    seq, qual, off = seq[:n * rs.L], qual[:n * rs.L], off[:n + 1]
    for b in (14, 18):
        oc = oracle.Counter(31, b)
        oc.count(seq, qual, off)
        g = _gpu_count(gpu_lib, 31, b, seq, qual, off, 1, debug_seen=True)  # debug_seen keeps the batch whole (otherwise an oversized batch is cut into sub-batches)
        assert np.array_equal(g.bloom_bytes(), oc.bloom_bytes())
        assert g.stats()["n_seen"] == oc.stats()["n_seen"]
        sizes, slots = g.export_table().export_sorted()
        osz, osl = oc.export()
        assert np.array_equal(sizes, osz) and np.array_equal(slots, osl)
        if b == 14:
            assert g.stats()["slow_buckets"] > 0
        g.close()


def test_table_growth(gpu_lib, g1):
    """Start with 2 slots per sub-table: overflow parking, growth and replay must keep L1."""
    rs, (seq, qual, off) = g1
    oc = oracle.Counter(31, 26, l_pre=10)
    oc.count(seq, qual, off)
    g = _gpu_count(gpu_lib, 31, 26, seq, qual, off, 3, l_pre=10, tab_cshift=1)
    sizes, slots = g.export_table().export_sorted()
    osz, osl = oc.export()
    assert np.array_equal(sizes, osz) and np.array_equal(slots, osl)
    assert g.stats()["tab_cshift"] > 1
    g.close()


def test_filter_mode(gpu_lib, g1):
    """-1 mode (count.c:148-154,67-68): both bloom filters vs the goldens of SURVEY B.3 (g1, k=51, b=26)."""
    rs, (seq, qual, off) = g1
    g = _gpu_count(gpu_lib, 51, 26, seq, qual, off, 2, filter_mode=1)
    st = g.stats()
    assert st["n_kmers"] == 664405 and st["n_seen"] == 301877
    L = oracle.lib()
    for which, (pop, fnv) in enumerate([(1434180, 0xd25fc72cd1ffffda), (364980, 0xafeb3dee4fc349c5)]):
        bits = g.bloom_bytes(which)
        assert int(L.orc_popcount_bytes(bits.ctypes.data, len(bits))) == pop
        assert int(L.orc_fnv1a64(bits.ctypes.data, len(bits))) == fnv
    g.close()


def test_massive_duplicates_saturate_exactly(gpu_lib):
    """Skew: the same reads thousands of times (a poly-A read and one ordinary read).  Counters saturate at 255 / 63
    (htab.c:77-78), the LDS aggregation must not wrap, and one bloom region receives almost every k-mer."""
    L = 150
    polyA = np.frombuffer(b"A" * L, dtype=np.uint8)
    rng = np.random.default_rng(3)
    other = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, L)]
    n = 6000
    seq = np.concatenate([polyA if i % 3 else other for i in range(n)])
    qual = np.full(len(seq), ord("I"), dtype=np.uint8)
    qual[10::75] = ord("#")  # some low-quality bases: most 31-mers stay high quality
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    oc = oracle.Counter(31, 24)
    oc.count(seq, qual, off)
    for nb in (1, 3):
        g = _gpu_count(gpu_lib, 31, 24, seq, qual, off, nb)
        assert np.array_equal(g.bloom_bytes(), oc.bloom_bytes())
        assert g.stats()["n_seen"] == oc.stats()["n_seen"]
        sizes, slots = g.export_table().export_sorted()
        osz, osl = oc.export()
        assert np.array_equal(sizes, osz) and np.array_equal(slots, osl)
        assert int((slots & np.uint64(0xff)).max()) == 255 and int(((slots >> np.uint64(8)) & np.uint64(0x3f)).max()) == 63
        g.close()


def test_unaligned_device_streams_take_the_byte_path(gpu_lib, g1):
    """Batches whose device pointers are not 16-byte aligned (e.g. a slice of a resident read set) use the ballot path of
    the plane builder; same results."""
    rs, (seq, qual, off) = g1
    n = 2000
    s = gpu_lib.to_stream(seq[:n * rs.L], off[:n + 1]); q = gpu_lib.to_stream(qual[:n * rs.L], off[:n + 1])
    oc = oracle.Counter(31, 24)
    oc.count(seq[:n * rs.L], qual[:n * rs.L], off[:n + 1])
    g = gpu_lib.GpuCounter(31, 24, max_batch_pos=len(s) + 64)
    d_s = g.dev_alloc(len(s) + 64); d_q = g.dev_alloc(len(q) + 64)
    pad = np.zeros(5, dtype=np.uint8)
    g.h2d(d_s, np.concatenate([pad, s])); g.h2d(d_q, np.concatenate([pad[:3], q]))  # seq at +5, qual at +3
    g.count_dev(d_s + 5, d_q + 3, len(s))
    assert np.array_equal(g.bloom_bytes(), oc.bloom_bytes())
    sizes, slots = g.export_table().export_sorted()
    osz, osl = oc.export()
    assert np.array_equal(sizes, osz) and np.array_equal(slots, osl)
    g.dev_free(d_s); g.dev_free(d_q); g.close()


@pytest.mark.parametrize("k,b,md5", [(31, 26, "d686549d10dd4c71243269013119784a"), (33, 30, "61b259cfa3b666884c9f4993b5d5677c"),
                                     (51, 26, "28e866f0bcd07c21e3f698827dd51c3a")])
@pytest.mark.parametrize("n_batches,cshift", [(1, 0), (5, 0), (3, 1)])
def test_exact_dump_is_byte_identical(gpu_lib, g1, tmp_path, k, b, md5, n_batches, cshift):
    """Parity level L2: with order stamps the -d dump has the md5 of `bfc -E -t1 -d` (goldens captured from the
    reference binary), whatever the batching, also across table growth and replay of parked k-mers (cshift=1)."""
    rs, (seq, qual, off) = g1
    g = _gpu_count(gpu_lib, k, b, seq, qual, off, n_batches, track_order=True, tab_cshift=cshift)
    t = g.export_table()
    fn = str(tmp_path / "d.hash")
    assert t.dump(fn) == 0
    assert oracle.md5_file(fn) == md5
    g.close()


@pytest.mark.parametrize("k", [31, 63])
def test_long_and_degenerate_reads(gpu_lib, k):
    """Reads far longer than a kernel tile (4096 positions: every tile boundary cuts k-mers), empty reads (two separators in a row),
    reads shorter than k, all-N reads, lower case, a read ending exactly on a tile boundary -- in several batches, vs the oracle."""
    rng = np.random.default_rng(7 + k)
    genome = rng.integers(0, 4, 60000)
    lens = [0, 1, k - 1, k, 4096 - 1, 4096, 4097, 20011, 0, 0, 12288, 33333, 5, 0, 8191, 50000, 2 * k, 1, 0]
    lens = lens + [int(v) for v in rng.integers(0, 9000, 30)]
    off = np.zeros(len(lens) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    seq = np.empty(int(off[-1]), dtype=np.uint8)
    for r, n in enumerate(lens):
        p = int(rng.integers(0, 60000 - n)) if n < 60000 else 0
        seq[int(off[r]):int(off[r + 1])] = np.frombuffer(b"ACGT", dtype=np.uint8)[genome[p:p + n]]
    seq[int(off[4]):int(off[5])][100:400] = ord("N")             # an N run inside a read
    seq[int(off[10]):int(off[11])] = ord("N")                    # an all-N read of 12288 bases
    seq[rng.integers(0, len(seq), 200)] = ord("N")
    seq[rng.integers(0, len(seq), 500)] |= 0x20
    qual = rng.integers(33, 74, len(seq)).astype(np.uint8)
    # every read twice, so that k-mers are seen and the table fills
    seq2, qual2 = np.concatenate([seq, seq]), np.concatenate([qual, qual])
    off2 = np.concatenate([off, off[1:] + off[-1]])
    oc = oracle.Counter(k, 24)
    oc.count(seq2, qual2, off2)
    n = len(off2) - 1
    g = gpu_lib.GpuCounter(k, 24, max_batch_pos=len(seq2) + n + 64)
    cuts = [0, 7, 8, 9, 25, n // 2, n]  # batches cut at read boundaries, some of them tiny
    for a, e in zip(cuts[:-1], cuts[1:]):
        o = off2[a:e + 1] - off2[a]
        s = gpu_lib.to_stream(seq2[int(off2[a]):int(off2[e])], o)
        q = gpu_lib.to_stream(qual2[int(off2[a]):int(off2[e])], o)
        g.count_host(s, q)
    ost, st = oc.stats(), g.stats()
    assert (st["n_kmers"], st["n_high"], st["n_seen"]) == (ost["n_kmers"], ost["n_high"], ost["n_seen"])
    assert st["n_kmers"] > 300000 and st["n_seen"] > 100000
    assert np.array_equal(g.bloom_bytes(), oc.bloom_bytes())
    sizes, slots = g.export_table().export_sorted()
    osz, osl = oc.export()
    assert np.array_equal(sizes, osz) and np.array_equal(slots, osl)
    g.close(); oc.close()


@pytest.mark.parametrize("fm,l2_big", [(0, 0), (0, 1)] + ([(1, 0)] if os.environ.get("BFC_TEST_MORE") else []))  # (filter mode at -b37: the c5 shape tests run it on c5's own read set; BFC_TEST_MORE=1 adds it here)
def test_largest_filter_b37(gpu_lib, g1, g42, fm, l2_big, monkeypatch):
    """`-s 3g` (bfc.c:42-53) gives -b37: a 16 GiB filter, 64 KiB regions, one 1024-thread workgroup per CU -- table mode and filter mode.
    Compared with the oracle through the set bits' positions (popcount + every word the oracle has set), statistics and the table."""
    rs, (seq, qual, off) = g42 if l2_big else g1
    n = 60_000 if l2_big else 4000  # (l2_big: one batch of 9 M positions -- above the one-pass partition's minimum at 2^10 level-1 buckets, where the large tile applies)
    seq, qual, off = seq[:n * rs.L], qual[:n * rs.L], off[:n + 1]
    k, b = 33, 37
    if l2_big:  # level 2 on tiles of 8192 records (KParams.l2_big: 12-byte records, 2^10 regions per bucket, one-pass partition; off by default since its A/B on c4e: DESIGN 6b)
        monkeypatch.setenv("BFCG_L2_BIG", "2")
    oc = oracle.Counter(k, b, filter_mode=fm)
    oc.count(seq, qual, off)
    g = _gpu_count(gpu_lib, k, b, seq, qual, off, 1 if l2_big else 2, filter_mode=fm)
    if l2_big:
        assert g.partition_info() == dict(one_pass=True, level2_one_pass=True, replayed_batches=0)
    ost, st = oc.stats(), g.stats()
    assert (st["n_kmers"], st["n_high"], st["n_seen"]) == (ost["n_kmers"], ost["n_high"], ost["n_seen"])
    for which in ([0, 1] if fm else [0]):
        ob = oc.bloom_view(bool(which)).view(np.uint64)
        gb = g.bloom_bytes(which).view(np.uint64)
        nz = np.flatnonzero(ob)
        assert len(nz) > 1000 and np.array_equal(gb[nz], ob[nz])
        assert int(oracle.lib().orc_popcount_bytes(gb.ctypes.data, gb.nbytes)) == int(np.bitwise_count(ob[nz]).sum())
        del ob, gb
    if not fm:
        sizes, slots = g.export_table().export_sorted()
        osz, osl = oc.export()
        assert np.array_equal(sizes, osz) and np.array_equal(slots, osl)
    g.close(); oc.close()


def test_oversized_batches_are_cut_into_sub_batches(gpu_lib, g1):
    """A batch far too large for the filter (here 1 M positions for 2^19 bits) is cut at non-ACGT bytes into sub-batches that the regions
    take at full speed: same results as the oracle, no region on the slow path -- through the host and the device entry point
    (device pointers of the parts are not 16-byte aligned)."""
    rs, (seq, qual, off) = g1
    oc = oracle.Counter(31, 19)
    oc.count(seq, qual, off)
    s, q = gpu_lib.to_stream(seq, off), gpu_lib.to_stream(qual, off)
    for dev in (False, True):
        g = gpu_lib.GpuCounter(31, 19, max_batch_pos=len(s) + 64)
        if dev:
            d_s = g.dev_alloc(len(s) + 64); d_q = g.dev_alloc(len(q) + 64)
            g.h2d(d_s, s); g.h2d(d_q, q)
            g.count_dev(d_s, d_q, len(s))
        else:
            g.count_host(s, q)
        st = g.stats()
        assert st["n_batches"] > 3 and st["slow_buckets"] == 0
        assert (st["n_kmers"], st["n_seen"]) == (oc.stats()["n_kmers"], oc.stats()["n_seen"])
        assert np.array_equal(g.bloom_bytes(), oc.bloom_bytes())
        sizes, slots = g.export_table().export_sorted()
        osz, osl = oc.export()
        assert np.array_equal(sizes, osz) and np.array_equal(slots, osl)
        if dev:
            g.dev_free(d_s); g.dev_free(d_q)
        g.close()
    oc.close()


@pytest.mark.parametrize("k", [31, 33])
def test_batches_whose_kmers_hardly_repeat_switch_to_stream_mode(gpu_lib, k):
    """A large genome at ~1.5x per batch: every region's aggregation table fills with singletons, the context notices (crowded regions)
    and hands the seen k-mers of later batches over as a plain stream (k_bloom STREAM + k_commit_stream).  Same results as the oracle."""
    rs = gen.ReadSet(seed=11, G=2_000_000, cov=9)
    seq, qual, off = rs.reads()
    oc = oracle.Counter(k, 28)
    oc.count(seq, qual, off)
    g = _gpu_count(gpu_lib, k, 28, seq, qual, off, 6)
    st, ost = g.stats(), oc.stats()
    assert st["crowded_regions"] > 2048 and st["stream_batches"] >= 3
    assert (st["n_kmers"], st["n_high"], st["n_seen"]) == (ost["n_kmers"], ost["n_high"], ost["n_seen"])
    assert np.array_equal(g.bloom_bytes(), oc.bloom_bytes())
    sizes, slots = g.export_table().export_sorted()
    osz, osl = oc.export()
    assert np.array_equal(sizes, osz) and np.array_equal(slots, osl)
    g.reset()  # back to aggregation
    assert g.stats()["crowded_regions"] == 0
    g.close(); oc.close()


# ---- k_bloom3fm (round 5): filter mode (`bfc -1`, count.c:67-68) on 16-byte records whose bloom address is a bit field of their words
@pytest.mark.parametrize("k,b", [(51, 31), (47, 28), (45, 30), (51, 33), (49, 26)])
@pytest.mark.parametrize("n_batches,list_cap", [(1, 0), (3, 0), (2, 64)])
def test_filter_mode_block_walk(gpu_lib, g1, monkeypatch, k, b, n_batches, list_cap):
    """Both filters (first: every k-mer's bits; second: the k-mers seen before, count.c:67-68) and the per-k-mer seen flags against the sequential
    oracle, for geometries that select k_bloom3fm (k >= bf_shift + 9, 16-byte records), over several batches, and with a list of 64 entries per
    region so that regions are taken in rounds of file-index ranges."""
    if list_cap:
        monkeypatch.setenv("BFCG_B3FM_LIST", str(list_cap))
    rs, (seq, qual, off) = g1
    n = 6000 if b >= 30 else 3000
    seq, qual, off = seq[:n * rs.L], qual[:n * rs.L], off[:n + 1]
    oc = oracle.Counter(k, b, filter_mode=1)
    tr = oc.count(seq, qual, off, trace=True)
    g = _gpu_count(gpu_lib, k, b, seq, qual, off, n_batches, filter_mode=1, debug_seen=(n_batches == 1))
    assert g.mg_info()["rec_bytes"] == 16
    assert np.array_equal(g.bloom_bytes(0), oc.bloom_bytes()), "first filter differs"
    assert np.array_equal(g.bloom_bytes(1), oc.bloom_bytes(True)), "second filter differs"
    ost, st = oc.stats(), g.stats()
    assert (st["n_kmers"], st["n_high"], st["n_seen"]) == (ost["n_kmers"], ost["n_high"], ost["n_seen"])
    assert st["slow_buckets"] == 0  # (no pool path in this kernel: overflowing regions take rounds)
    if n_batches == 1:
        fl = g.seen_flags(len(seq) + n)
        assert np.array_equal(fl[fl > 0] == 2, (tr[:, 3] >> 1) & 1 == 1), "seen flags differ from the sequential semantics"
    g.close(); oc.close()


def test_filter_mode_block_walk_heavy_regions(gpu_lib, monkeypatch):
    """A filter of ONE region's worth per 2^17 bits and a genome with many copies: thousands of k-mers per region and block, lists that overflow
    several times over (rounds halve the file-index range until a range fits), copies of a k-mer inside one batch (the second copy is seen)."""
    rng = np.random.default_rng(515)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    L, G, n = 150, 3000, 4000
    genome = rng.choice(acgt, G + L)
    pos = rng.integers(0, G, n)
    seq = genome[(pos[:, None] + np.arange(L)[None, :])].astype(np.uint8).reshape(-1)
    qual = rng.integers(33, 74, len(seq)).astype(np.uint8)
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    for k, b in ((47, 20), (45, 19)):  # (16-byte records: 2k + 33 <= 128 without dropped bits)
        oc = oracle.Counter(k, b, filter_mode=1)
        oc.count(seq, qual, off)
        g = _gpu_count(gpu_lib, k, b, seq, qual, off, 2, filter_mode=1)
        assert g.mg_info()["rec_bytes"] == 16
        assert np.array_equal(g.bloom_bytes(0), oc.bloom_bytes()) and np.array_equal(g.bloom_bytes(1), oc.bloom_bytes(True))
        assert g.stats()["n_seen"] == oc.stats()["n_seen"]
        g.close(); oc.close()


@pytest.mark.parametrize("wc,n_reads", [("forced", 15_000), ("default", 15_000), ("forced", 40_000)])
def test_small_batch_in_a_large_context(gpu_lib, g42, monkeypatch, wc, n_reads):
    """a batch of a few million positions (just above the one-pass partition's minimum, nb1 * 8 * 1024) in a context sized for
    2^28, on the FULL persistent grid of k_scatter1_wc.  What the workgroups reserve and do not fill is dead records in the slabs -- more of them
    than live ones here --, level 2's rows follow the slabs' fill, and a grid sized by the batch's positions left the last rows' k-mers uncounted.
    `forced` (BFCG_S1_WC=2) keeps the full grid and groups of four chunks whatever the batch's size (the plan's own scaling by the batch is off):
    the case as it was found; `default`: the plan shrinks groups and grid for such a batch.  Counts, bitmap and table equal the oracle's, and the
    two-pass partition's (count.c:72-89, bbf.c:25-45, htab.c:60-82)."""
    rs, (seq, qual, off) = g42
    k, b = 33, 33  # (2^8 level-1 buckets: the one-pass partition from 2^8 x 8 x 1024 = 2.1 M positions on; 15 000 reads are 2.27 M)
    seq, qual, off = seq[:n_reads * rs.L], qual[:n_reads * rs.L], off[:n_reads + 1]
    if wc == "forced":
        monkeypatch.setenv("BFCG_S1_WC", "2")
    s, q = gpu_lib.to_stream(seq, off), gpu_lib.to_stream(qual, off)
    g = gpu_lib.GpuCounter(k, b, max_batch_pos=1 << 28)
    w0 = g.s1wc_launches()
    g.count_host(s, q)
    st = g.stats()
    assert g.partition_info()["one_pass"] and g.partition_info()["replayed_batches"] == 0
    assert g.s1wc_launches() > w0, "the batch did not take k_scatter1_wc"
    oc = oracle.Counter(k, b)
    oc.count(seq, qual, off)
    ost = oc.stats()
    assert (st["n_kmers"], st["n_high"], st["n_seen"]) == (ost["n_kmers"], ost["n_high"], ost["n_seen"])
    bits = g.bloom_bytes()
    assert gen.bitmap_checksums(bits) == oc.bloom_checksums()
    del bits
    sizes, slots = g.export_table().export_sorted()
    assert oracle.l1_digest(sizes, slots) == oracle.l1_digest(*oc.export())
    g.close(); oc.close()
    monkeypatch.setenv("BFCG_ONEPASS", "0")
    g2 = gpu_lib.GpuCounter(k, b, max_batch_pos=1 << 28)
    g2.count_host(s, q)
    st2 = g2.stats()
    assert (st2["n_kmers"], st2["n_seen"], st2["n_keys"]) == (st["n_kmers"], st["n_seen"], st["n_keys"])
    g2.close()
