"""GPU parity under random parameters (-m gpu): k, filter size, number of hashes, l_pre, quality threshold, read lengths, coverage, Ns,
FASTA/FASTQ, batch cuts, region size and initial table size drawn from a seeded generator; every draw is checked bit for bit against the
oracle (bloom bitmaps L0, statistics, table L1).  Small inputs, many shapes: one-level and two-level partitions (bf_shift 10..27),
single-block regions, tables that grow several times, k from 5 to 63; a second family draws bf_shift 28..33."""
import os
import sys

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


SEED_BASE = int(os.environ.get("BFC_FUZZ_SEED_BASE", "0"))  # other values draw other configurations (scripts/more_fuzz.sh)


def _draw(seed, scale=1, b_range=(10, 28)):
    rng = np.random.default_rng(seed + 100003 * SEED_BASE)
    k = int(rng.choice([5, 9, 13, 17, 21, 25, 27, 29, 31, 32, 33, 35, 39, 47, 48, 55, 63]))
    b = int(rng.integers(*b_range))
    nh = int(rng.choice([1, 2, 3, 4, 4, 4, 5, 7, 12]))
    l_pre = int(rng.choice([v for v in (4, 8, 12, 16, 20) if v <= 2 * k - 2]))  # htab.c:49-50 shifts by 2k - l_pre: the reference needs it positive
    q = int(rng.choice([-50, 0, 10, 20, 30, 41, 94, 100]))
    fm = int(rng.random() < 0.25)
    n = int(rng.integers(1, 1500 * scale))
    cov = float(rng.choice([0.5, 2, 8, 40]))
    lmax = int(rng.choice([3, 40, 151, 600]))
    lens = rng.integers(0, lmax + 1, n)
    G = max(64, int(lens.sum() / cov))
    genome = rng.integers(0, 4, G + lmax + 1)
    off = np.zeros(n + 1, dtype=np.uint64); off[1:] = np.cumsum(lens)
    seq = np.empty(int(off[-1]), dtype=np.uint8)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for r in range(n):
        p = int(rng.integers(0, G))
        seq[int(off[r]):int(off[r + 1])] = acgt[genome[p:p + lens[r]]]
    if len(seq):
        err = rng.random(len(seq)) < 0.01
        seq[err] = acgt[rng.integers(0, 4, int(err.sum()))]
        seq[rng.random(len(seq)) < 0.002] = ord("N")
        low = rng.random(len(seq)) < 0.05
        seq[low] |= 0x20
        odd = rng.random(len(seq)) < 0.003
        seq[odd] = rng.integers(0, 256, int(odd.sum())).astype(np.uint8)  # any byte at all: everything but ACGTacgt ends a k-mer (bseq.c:9-26)
    qual = None if rng.random() < 0.2 else (rng.integers(0, 256, len(seq)) if rng.random() < 0.25 else rng.integers(33, 75, len(seq))).astype(np.uint8)  # sometimes every byte value
    cuts = sorted(set([0, n] + [int(v) for v in rng.integers(0, n + 1, int(rng.integers(0, 5)))]))
    kw = {}
    if rng.random() < 0.3:
        kw["region_shift"] = int(rng.integers(4, 11))
    if rng.random() < 0.3:
        kw["tab_cshift"] = int(rng.integers(1, 4))
    if seed % 3 == 0:
        kw["table_layout"] = 1  # the host's (sub-table, key) layout from the start; otherwise region-owned segments where the geometry allows
    return dict(k=k, b=b, nh=nh, l_pre=l_pre, q=q, fm=fm), seq, qual, off, cuts, kw


def _check(gpu_lib, prm, seq, qual, off, cuts, kw, planes=False):
    n = len(off) - 1
    oc = oracle.Counter(prm["k"], prm["b"], q=prm["q"], n_hashes=prm["nh"], l_pre=prm["l_pre"], filter_mode=prm["fm"])
    oc.count(seq, qual, off)
    cap = len(seq) + n + 64
    g = gpu_lib.GpuCounter(prm["k"], prm["b"], q=prm["q"], n_hashes=prm["nh"], l_pre=prm["l_pre"], filter_mode=prm["fm"], max_batch_pos=cap, **kw)
    if planes:  # the whole input as ONE set of bit planes (bfcg_pack_planes, three packing threads' ranges), its pieces counted at their bit offsets
        pl = gpu_lib.pack_planes(gpu_lib.to_stream(seq, off), gpu_lib.to_stream(qual, off) if qual is not None else None, prm["q"], n_chunks=3)
        for a, e in zip(cuts[:-1], cuts[1:]):
            g.count_planes(pl, int(off[a]) + a, int(off[e]) + e - int(off[a]) - a, has_qual=qual is not None)
    for a, e in zip(cuts[:-1], cuts[1:]):
        if planes:
            break
        o = off[a:e + 1] - off[a]
        s = gpu_lib.to_stream(seq[int(off[a]):int(off[e])], o)
        qq = gpu_lib.to_stream(qual[int(off[a]):int(off[e])], o) if qual is not None else None
        g.count_host(s, qq)
    ost, st = oc.stats(), g.stats()
    tag = "%r cuts=%r %r reads=%d bases=%d" % (prm, cuts, kw, n, len(seq))
    assert (st["n_kmers"], st["n_high"], st["n_seen"]) == (ost["n_kmers"], ost["n_high"], ost["n_seen"]), tag
    assert np.array_equal(g.bloom_bytes(), oc.bloom_bytes()), tag
    if prm["fm"]:
        assert np.array_equal(g.bloom_bytes(1), oc.bloom_bytes(True)), tag
    else:
        sizes, slots = g.export_table().export_sorted()
        osz, osl = oc.export()
        assert np.array_equal(sizes, osz) and np.array_equal(slots, osl), tag
    info = g.table_info() if not prm["fm"] else None
    g.close(); oc.close()
    return info


@pytest.mark.parametrize("seed", range(90))
def test_random_configuration(gpu_lib, seed):
    _check(gpu_lib, *_draw(1000 + seed))


@pytest.mark.parametrize("seed", range(60))
def test_random_configuration_through_bit_planes(gpu_lib, seed):
    """the same draws handed over as bit planes (bfcg_count_batch_planes: 4 bits per position over PCIe): any byte value in sequence and quality,
    thresholds from -50 to 100 (beyond what a signed char can reach on either side), records without qualities, pieces that begin at any bit"""
    _check(gpu_lib, *_draw(1000 + seed), planes=True)


@pytest.mark.parametrize("seed", range(6))
def test_bit_planes_oversized_batches_are_cut(gpu_lib, seed):
    """a plane set far larger than its filter takes at full speed is cut at separators found in the not-ACGT plane"""
    prm, seq, qual, off, cuts, kw = _draw(52000 + seed, scale=6, b_range=(14, 18))
    kw.pop("region_shift", None)
    _check(gpu_lib, prm, seq, qual, off, [0, len(off) - 1], kw, planes=True)


@pytest.mark.parametrize("blk", ["3", "5", "7"])
@pytest.mark.parametrize("seed", range(20))
def test_random_configuration_segments_of_several_blocks(gpu_lib, seed, blk, monkeypatch):
    """table segments beyond what a CU's LDS holds are several BLOCKS, one workgroup each (KParams.seg_blk): with BFCG_SEG_BLOCK = 3 / 5 / 7 a block
    is 8 / 32 / 128 slots instead of 2^14, so these small draws grow their segments across the block boundary several times (rehash block by
    block, commits that take only their block's entries of a region's pages, parked k-mers replayed into their block, export) -- in every
    commit path (counter pairs, compare-and-swap, in place), with hand-over windows and without"""
    monkeypatch.setenv("BFCG_SEG_BLOCK", blk)
    prm, seq, qual, off, cuts, kw = _draw(61000 + seed, scale=2, b_range=(10, 20))
    kw.pop("table_layout", None)
    prm["fm"] = 0
    info = _check(gpu_lib, prm, seq, qual, off, cuts, kw)
    _BLOCK_RUNS.append((seed, blk, info))  # (the export converted the segments: what the family exercised is checked once, below)


_BLOCK_RUNS = []


def test_segments_of_several_blocks_were_exercised(gpu_lib, monkeypatch):
    """one fixed draw of the family above, looked at BEFORE its export: region-owned segments, grown past the block size"""
    monkeypatch.setenv("BFCG_SEG_BLOCK", "4")
    rng = np.random.default_rng(5)
    n, L, G = 3000, 100, 40000
    genome = rng.integers(0, 4, G + L)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    off = np.arange(n + 1, dtype=np.uint64) * L
    seq = np.concatenate([acgt[genome[p:p + L]] for p in rng.integers(0, G, n)]).astype(np.uint8)
    qual = rng.integers(33, 75, len(seq)).astype(np.uint8)
    g = gpu_lib.GpuCounter(21, 16, max_batch_pos=len(seq) + n + 64)   # 2^16 bits: one region; 2k - 0 = 42 identity bits
    for a in range(0, n, 500):
        o = off[a:a + 501] - off[a]
        g.count_host(gpu_lib.to_stream(seq[int(off[a]):int(off[a + 500])], o), gpu_lib.to_stream(qual[int(off[a]):int(off[a + 500])], o))
    g.sync()
    ti = g.table_info()
    assert ti["segments"] and ti["seg_shift"] >= 12 and ti["seg_growths"] >= 3, ti   # blocks of 2^4 slots: 2^8 blocks and more per region
    oc = oracle.Counter(21, 16)
    oc.count(seq, qual, off)
    sizes, slots = g.export_table().export_sorted()
    osz, osl = oc.export()
    assert np.array_equal(sizes, osz) and np.array_equal(slots, osl)
    g.close(); oc.close()


@pytest.mark.parametrize("seed", range(16))
def test_random_configuration_large_filters(gpu_lib, seed):
    """the same kind of draws with filters of 2^28 .. 2^33 bits (up to the default 1 GiB): two full scatter levels, tens of thousands of
    bloom regions most of which see no k-mer at all, and the table geometries that go with them"""
    _check(gpu_lib, *_draw(30000 + seed, scale=4, b_range=(28, 34)))


@pytest.mark.parametrize("seed", range(30))
def test_random_configuration_one_pass_partition(gpu_lib, seed, monkeypatch):
    """the one-pass partition (bucket slabs + cursors) on draws it would normally leave to the two-pass one: with
    BFCG_ONEPASS_MIN_TILES=1 every batch of a filter of 2^26 bits and more goes through it, however few records a slab expects -- slabs
    overflow at level 1 or 2 in the middle of a run (low-complexity draws: coverage 40 of a tiny genome), the poisoned batch and the ones
    enqueued behind it are replayed, the rest of the run falls back; any k, any number of hashes, both table layouts, filter mode, batch
    cuts, several region sizes.  Bit for bit the oracle's filter(s), statistics and table."""
    monkeypatch.setenv("BFCG_ONEPASS_MIN_TILES", "1")
    _check(gpu_lib, *_draw(40000 + seed, scale=12, b_range=(26, 32)))


@pytest.mark.parametrize("seed", range(15))
@pytest.mark.parametrize("chunk", [3, 32])
def test_random_configuration_chunked_reservations(gpu_lib, seed, chunk, monkeypatch):
    """level 1 of the one-pass partition reserves room in its slabs in CHUNKS (round 3: a run goes into what is left of the workgroup's last
    chunk for that bucket and then into a new one; what the workgroups leave unused is filled with dead records that level 2 skips).  Forced
    here on draws whose slabs expect a handful of records -- runs split at every chunk size, dead records in most slabs, slabs that overflow
    because of them (replayed) -- against the oracle bit for bit, like every other draw."""
    monkeypatch.setenv("BFCG_ONEPASS_MIN_TILES", "1")
    monkeypatch.setenv("BFCG_S1_CHUNK", str(chunk))
    _check(gpu_lib, *_draw(45000 + seed, scale=12, b_range=(26, 32)))


@pytest.mark.parametrize("seed", range(24))
def test_random_configuration_k33_on_tiny_slabs(gpu_lib, seed, monkeypatch):
    """The default path's own geometry -- k > 32 with 12-byte records (k = 33 and 35 put onto the draws; where 35 needs 16-byte records the
    generic kernels run): K1 on 32-bit halves, the 32-bit decode of k_bloom3 with its block marks and worklists, the hand-over log -- forced
    onto tiny slabs (overflows are replayed through two passes): bit for bit the oracle's filter, statistics and table.  (Round 3 ran this
    family through the write-combining level-1 variant k_scatter1_wc, which round 4 removed: level with the tile kernel, never the default.)"""
    monkeypatch.setenv("BFCG_ONEPASS_MIN_TILES", "1")
    prm, seq, qual, off, cuts, kw = _draw(48000 + seed, scale=12, b_range=(26, 33))
    prm = dict(prm, k=35 if seed % 5 == 4 else 33)
    _check(gpu_lib, prm, seq, qual, off, cuts, kw)


@pytest.mark.parametrize("seed", range(40))
def test_random_configuration_write_combining_level1(gpu_lib, seed, monkeypatch):
    """k_scatter1_wc (round 5, bfcg_scatter1wc.hip): level 1 of the one-pass partition through write-combining buffers in LDS -- a buffer of 16 or 32
    records per bucket (8 or 16 with two workgroups of 512 threads per CU), full buffers leave as whole chunks into room reserved a group ahead, what finds its buffer full waits a round in
    registers, what finds it full twice over takes a chunk of its own, dead records in everything reserved and not filled.  Forced (BFCG_S1_WC=2)
    onto draws whose slabs expect a handful of records, with 8 / 16 / 64 workgroups sharing them and 2^8 / 2^9 / 2^10 level-1 buckets: every path of the
    kernel runs -- spills on the low-complexity draws, slabs that overflow because of the padding (replayed through two passes), tiles stolen
    from other XCDs' counters, batches of less than one tile -- for k = 33 at compile time, k = 35 at run time and k <= 32 on one word.  Bit for
    bit the oracle's filter(s), statistics and table, like every other draw."""
    monkeypatch.setenv("BFCG_ONEPASS_MIN_TILES", "1")
    monkeypatch.setenv("BFCG_S1_WC", "2")
    monkeypatch.setenv("BFCG_S1_WC_WGS", str([8, 16, 64][seed % 3]))
    monkeypatch.setenv("BFCG_F1", str(8 + seed % 3))  # (2^10 buckets: config c4's geometry, 1024 threads with two banks of reservers)
    monkeypatch.setenv("BFCG_S1_WC_BT", "1024" if seed % 4 >= 2 else "512")  # (one workgroup of 1024 threads per CU, or two of 512 with buffers half the size)
    if seed % 4 == 3:
        monkeypatch.setenv("BFCG_S1_CHUNK", "64")  # (four chunks of 16 per reservation)
    prm, seq, qual, off, cuts, kw = _draw(52000 + seed, scale=12, b_range=(28, 34))
    prm = dict(prm, k=[33, 33, 35, 31, 27][seed % 5])
    prm["l_pre"] = min(prm["l_pre"], 2 * prm["k"] - 2)
    kw.pop("region_shift", None)  # (regions of 2^8 blocks: the level split asked for above exists for every filter drawn)
    _check(gpu_lib, prm, seq, qual, off, cuts, kw)


@pytest.mark.parametrize("seed", range(24))
def test_random_configuration_write_combining_level1_16_byte_records(gpu_lib, seed, monkeypatch):
    """k_scatter1_wc on 16-byte records (config c5's geometry: k = 51, 2^10 level-1 buckets; 1024 threads, buffers of 8 x 16 bytes, three
    positions per thread and round): k = 39 / 47 / 51 / 52 forced onto small draws with 8 / 16 / 64 workgroups, table and filter mode, FASTA and
    FASTQ -- bit for bit the oracle's filter(s), statistics and table."""
    monkeypatch.setenv("BFCG_ONEPASS_MIN_TILES", "1")
    monkeypatch.setenv("BFCG_S1_WC", "2")
    monkeypatch.setenv("BFCG_S1_WC_WGS", str([8, 16, 64][seed % 3]))
    monkeypatch.setenv("BFCG_F1", "10")
    prm, seq, qual, off, cuts, kw = _draw(53000 + seed, scale=12, b_range=(29, 34))
    prm = dict(prm, k=[39, 47, 51, 52][seed % 4])
    prm["l_pre"] = min(prm["l_pre"], 2 * prm["k"] - 2)
    kw.pop("region_shift", None)
    _check(gpu_lib, prm, seq, qual, off, cuts, kw)


def test_write_combining_level1_was_exercised(gpu_lib, monkeypatch):
    """the families above are only worth their name if the kernel runs: one input that does not depend on the fuzz seed base, the process-wide
    launch counter before and after -- 12-byte records (k = 33, 2^9 buckets), then 16-byte ones (k = 51, 2^10 buckets, filter mode)"""
    monkeypatch.setenv("BFCG_ONEPASS_MIN_TILES", "1")
    monkeypatch.setenv("BFCG_S1_WC", "2")
    monkeypatch.setenv("BFCG_S1_WC_WGS", "16")
    rng = np.random.default_rng(52001)
    n, L, G = 4000, 120, 60000
    genome = rng.integers(0, 4, G + L)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    off = np.arange(n + 1, dtype=np.uint64) * L
    seq = np.concatenate([acgt[genome[p:p + L]] for p in rng.integers(0, G, n)]).astype(np.uint8)
    qual = rng.integers(33, 75, len(seq)).astype(np.uint8)

    def launches():
        g = gpu_lib.GpuCounter(33, 30, max_batch_pos=64)
        v = g.s1wc_launches(); g.close()
        return v

    before = launches()
    monkeypatch.setenv("BFCG_F1", "9")
    _check(gpu_lib, dict(k=33, b=30, nh=4, l_pre=20, q=20, fm=0), seq, qual, off, [0, n // 3, n], {})
    mid = launches()
    assert mid > before, (before, mid)
    monkeypatch.setenv("BFCG_F1", "10")
    _check(gpu_lib, dict(k=51, b=31, nh=4, l_pre=20, q=20, fm=1), seq, qual, off, [0, n], {})
    after = launches()
    assert after > mid, (mid, after)


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("f1", [2, 7, 10])
def test_random_configuration_uneven_level_split(gpu_lib, seed, f1, monkeypatch):
    """BFCG_F1 moves bits between the two scatter levels (2^F1 level-1 buckets x 2^(F-F1) regions per bucket; the library's own choice is
    the even split, which the c3 A/B in profiles/round3_f1_split.txt found fastest).  Few fat buckets, many thin ones, through the one-pass
    and the two-pass partition: the split is a tuning knob and must not change a bit of the filter, the statistics or the table."""
    monkeypatch.setenv("BFCG_F1", str(f1))
    if seed & 1:
        monkeypatch.setenv("BFCG_ONEPASS_MIN_TILES", "1")
    _check(gpu_lib, *_draw(47000 + seed, scale=8, b_range=(28, 34)))


@pytest.mark.parametrize("seed", range(8))
def test_random_medium_configuration(gpu_lib, seed):
    """the same draws at 40x the size (up to 60 000 reads, tens of millions of positions): many tiles per bucket, multi-chunk scans, full
    aggregation tables, table growth inside a batch, STREAM mode decisions -- still bit for bit the oracle's filter(s), statistics and table"""
    _check(gpu_lib, *_draw(20000 + seed, scale=int(os.environ.get("BFC_FUZZ_SCALE", "40"))))


@pytest.mark.parametrize("seed", range(14))
@pytest.mark.parametrize("base", [0, 21])
def test_random_configuration_on_emulated_ranks(gpu_lib, seed, base, monkeypatch):
    """the same draws through the multi-GPU stages: 2 / 4 / 8 ranks emulated on one device (LocalCluster), ragged rank shares, ranks
    with nothing to contribute; global batches in rank-major order must equal the sequential oracle.  Seed base 21 is part of the suite: its
    draws found round 2's last bug (a one-pass level 2 replayed from a receive buffer the harness had reused)."""
    import mg_protocol as bdist
    if base:
        monkeypatch.setattr(sys.modules[__name__], "SEED_BASE", SEED_BASE + base)
    prm, seq, qual, off, cuts, kw = _draw(5000 + seed, scale=int(os.environ.get("BFC_FUZZ_RANK_SCALE", "1")))  # 40: the medium-size draws through the rank stages
    rng = np.random.default_rng(seed)
    world = int(rng.choice([2, 4, 8]))
    if prm["b"] < 21:
        prm["b"] = int(rng.integers(21, 27))  # more bloom regions (2^(b-17)) than ranks
    if qual is None:
        qual = np.full(len(seq), ord("I"), dtype=np.uint8)
    n = len(off) - 1
    oc = oracle.Counter(prm["k"], prm["b"], q=prm["q"], n_hashes=prm["nh"], l_pre=prm["l_pre"], filter_mode=prm["fm"])
    oc.count(seq, qual, off)
    cl = bdist.LocalCluster(gpu_lib, world, prm["k"], prm["b"], max_batch_pos=len(seq) + n + 64, q=prm["q"], n_hashes=prm["nh"], l_pre=prm["l_pre"], filter_mode=prm["fm"])
    for a, e in zip(cuts[:-1], cuts[1:]):  # one global batch per cut; its reads are dealt to the ranks in contiguous, uneven shares
        marks = sorted([a, e] + [int(v) for v in rng.integers(a, e + 1, world - 1)])
        shares = []
        for r in range(world):
            lo, hi = marks[r], marks[r + 1]
            o = off[lo:hi + 1] - off[lo]
            shares.append((gpu_lib.to_stream(seq[int(off[lo]):int(off[hi])], o), gpu_lib.to_stream(qual[int(off[lo]):int(off[hi])], o)))
        cl.batch(shares)
    st, ost = cl.stats(), oc.stats()
    tag = "%r world=%d cuts=%r" % (prm, world, cuts)
    assert (st["n_kmers"], st["n_high"], st["n_seen"]) == (ost["n_kmers"], ost["n_high"], ost["n_seen"]), tag
    assert np.array_equal(cl.bloom_bytes(), oc.bloom_bytes()), tag
    if prm["fm"]:
        assert np.array_equal(cl.bloom_bytes(1), oc.bloom_bytes(True)), tag
    else:
        sizes, slots = cl.export_sorted()
        osz, osl = oc.export()
        assert np.array_equal(sizes, osz) and np.array_equal(slots, osl), tag
    cl.close(); oc.close()


@pytest.mark.parametrize("seed", range(24))
def test_random_exact_dump(gpu_lib, seed, tmp_path):
    """Parity level L2 under random parameters: with order stamps (track_order) the dump file is byte for byte the oracle's, whose khash
    emulation is pinned to `bfc -E -t1 -d` by the md5 goldens (tests/test_oracle.py) -- any k, l_pre, batching, initial table size."""
    prm, seq, qual, off, cuts, kw = _draw(9000 + seed)
    if prm["fm"]:
        prm["fm"] = 0
    n = len(off) - 1
    oc = oracle.Counter(prm["k"], prm["b"], q=prm["q"], n_hashes=prm["nh"], l_pre=prm["l_pre"])
    oc.count(seq, qual, off)
    g = gpu_lib.GpuCounter(prm["k"], prm["b"], q=prm["q"], n_hashes=prm["nh"], l_pre=prm["l_pre"], max_batch_pos=len(seq) + n + 64, track_order=True, **kw)
    for a, e in zip(cuts[:-1], cuts[1:]):
        o = off[a:e + 1] - off[a]
        g.count_host(gpu_lib.to_stream(seq[int(off[a]):int(off[e])], o), gpu_lib.to_stream(qual[int(off[a]):int(off[e])], o) if qual is not None else None)
    f_gpu, f_orc = str(tmp_path / "gpu.hash"), str(tmp_path / "orc.hash")
    t = g.export_table()
    assert t.dump(f_gpu) == 0
    oc.dump(f_orc)
    assert open(f_gpu, "rb").read() == open(f_orc, "rb").read(), "%r cuts=%r %r" % (prm, cuts, kw)
    t.close(); g.close(); oc.close()


@pytest.mark.parametrize("q4", ["0", "1"])
@pytest.mark.parametrize("seed", range(14))
def test_random_trim_pass(gpu_lib, seed, q4, monkeypatch):
    """`bfc -1` under random parameters: count in filter mode on the GPU, then the GPU trim pass; start / end of every read equal the
    oracle's max_streak + keep rule (correct.c:478-497, 557-569) on the oracle's second filter.  q4 = 1: through k_query4 (four lanes fetch the
    four dwords of a query's positions, two queries in flight -- the kernel filters of 4 GiB and more take) whatever the filter's size."""
    import ctypes as C
    monkeypatch.setenv("BFCG_QUERY4", q4)
    prm, seq, qual, off, cuts, kw = _draw(13000 + seed)
    rng = np.random.default_rng(seed)
    n = len(off) - 1
    min_frac = float(rng.choice([0.5, 0.9, 0.99]))
    oc = oracle.Counter(prm["k"], prm["b"], q=prm["q"], n_hashes=prm["nh"], filter_mode=1)
    oc.count(seq, qual, off)
    g = gpu_lib.GpuCounter(prm["k"], prm["b"], q=prm["q"], n_hashes=prm["nh"], filter_mode=1, max_batch_pos=len(seq) + n + 64)
    s_seq = gpu_lib.to_stream(seq, off)
    g.count_host(s_seq, gpu_lib.to_stream(qual, off) if qual is not None else None)
    bf = g.export_bloom(1)
    g.close()
    L = oracle.lib()
    obf = L.orc_state_bf_high(oc.st)
    soff = off + np.arange(n + 1, dtype=np.uint64)
    tr = gpu_lib.GpuTrimmer(prm["k"], bf, max_pos=len(s_seq) + 64, max_reads=n)
    start, end = tr.trim(s_seq, soff, min_frac)
    kept = 0
    for r in range(n):
        s = seq[int(off[r]):int(off[r + 1])]
        if len(s) == 0:
            assert start[r] == -1
            continue
        mx = L.orc_max_streak(prm["k"], obf, s.ctypes.data, len(s))
        a, e = C.c_int(), C.c_int()
        if L.orc_trim_decide(mx, prm["k"], len(s), min_frac, C.byref(a), C.byref(e)):
            assert (int(start[r]), int(end[r])) == (a.value, e.value), (r, prm)
            kept += 1
        else:
            assert start[r] == -1, (r, prm)
    tr.close(); bf.close(); oc.close()


@pytest.mark.parametrize("seed", range(16))
def test_random_kcov(gpu_lib, seed):
    """bfc_ec_kcov for whole batches under random parameters: table counted on the GPU and kept in HBM, coverage of every base vs orc_kcov
    (pinned to the reference's own function in tests/test_kcov.py)"""
    prm, seq, qual, off, cuts, kw = _draw(17000 + seed)
    rng = np.random.default_rng(seed)
    min_occ = int(rng.choice([1, 2, 3, 5]))
    n = len(off) - 1
    oc = oracle.Counter(prm["k"], prm["b"], q=prm["q"], n_hashes=prm["nh"], l_pre=prm["l_pre"])
    oc.count(seq, qual, off)
    L = oracle.lib()
    want = np.zeros(len(seq), dtype=np.uint16)
    for r in range(n):
        a, e = int(off[r]), int(off[r + 1])
        if e > a:
            L.orc_kcov(L.orc_state_ch(oc.st), min_occ, seq[a:e].ctypes.data, e - a, want[a:e].ctypes.data)
    s = gpu_lib.to_stream(seq, off)
    g = gpu_lib.GpuCounter(prm["k"], prm["b"], q=prm["q"], n_hashes=prm["nh"], l_pre=prm["l_pre"], max_batch_pos=len(s) + 64, **kw)
    g.count_host(s, gpu_lib.to_stream(qual, off) if qual is not None else None)
    kc = gpu_lib.GpuKcov(g, max_pos=len(s) + 64)
    got = kc.kcov(s, min_occ) if len(s) else np.zeros(0, dtype=np.uint16)
    sep = np.asarray(off[1:], dtype=np.int64) + np.arange(n)
    assert not got[sep].any()
    keep = np.ones(len(got), dtype=bool); keep[sep] = False
    assert np.array_equal(got[keep], want), prm
    kc.close(); g.close(); oc.close()
