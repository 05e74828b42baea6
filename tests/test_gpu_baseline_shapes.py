"""GPU parity at BASELINE.json's own shapes (-m gpu).

The fixtures of the other GPU tests stop at 100 kb genomes and filters of 2^27 bits; the configurations the metric is quoted
on run other kernel geometries: c2 (k=31, -b33: 65 536 bloom regions, 8+8 scatter levels, aggregation mode, 12-byte records),
c3 (k=33, -b35: 262 144 regions, 9+9 levels, 16-byte records, STREAM mode, a table that grows several times), c4 / c5
(-b37: 2^20 regions, 10+10 levels; k=51 `-1`: two 16 GiB filters, 20-byte records).  Here the GPU path runs those shapes
and must reproduce what THE REFERENCE ITSELF computed for the same read sets:
  * tests/golden/fixtures.json  (g42 at k=31/-b30 and k=33/-b33), and
  * tests/golden/baseline.json  (the full c2 and c3 read sets, and c4's / c5's parameters on a 20 Mbp genome;
    generator tests/golden/make_baseline_goldens.py: the reference's functions in file order = `bfc -t1`).
Compared: k-mer / high / seen totals, distinct keys, bloom popcount + FNV-1a (L0 through its digest), both histograms,
and the layout-free L1 digest of the whole table.
"""
import json
import os

import numpy as np
import pytest

import oracle
from bfc_amd import gen

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "fixtures.json")))
BASE = {e["name"]: e for e in json.load(open(os.path.join(HERE, "golden", "baseline.json")))}


def _count_fixed(gpu_lib, rs, k, b, batch_reads, filter_mode=0, **kw):
    """Whole read set of `rs` (fixed-length reads) through the device API in batches of `batch_reads`, generated chunk by chunk."""
    stride = rs.L + 1
    g = gpu_lib.GpuCounter(k, b, filter_mode=filter_mode, max_batch_pos=min(batch_reads, rs.n_reads) * stride + 64, **kw)
    for r0 in range(0, rs.n_reads, batch_reads):
        r1 = min(rs.n_reads, r0 + batch_reads)
        seq, qual, _ = rs.reads(r0, r1)
        g.count_host(gen.to_stream(seq, rs.L, 10), gen.to_stream(qual, rs.L, 33))
    return g


def _check_against(g, e, l1=True):
    st = g.stats()
    assert (st["n_kmers"], st["n_high"], st["n_seen"]) == (e["n_kmers"], e["n_high"], e["n_seen"]), (st, e["name"] if "name" in e else e["fixture"])
    def same_filter(bits, pfx):
        # an entry that carries the parallel digest of the reference's filter (bf_mix64, round 6: bfcgen_mix64 -- every 64-bit word times a constant of its
        # position, summed) is compared through it and the popcount: FNV-1a is one serial chain, 17 s of a core for a 16 GiB filter, and bench.py still
        # takes it for the same read sets after every run
        if pfx + "_mix64" in e:
            return (gen._L().bfcgen_popcount(bits.ctypes.data, len(bits)), "%016x" % gen.bitmap_mix64(bits)) == (e[pfx + "_popcount"], e[pfx + "_mix64"])
        want = e[pfx + "_fnv1a64"] if isinstance(e[pfx + "_fnv1a64"], int) else int(e[pfx + "_fnv1a64"], 16)
        return gen.bitmap_checksums(bits) == (e[pfx + "_popcount"], want)
    assert same_filter(g.bloom_bytes(), "bf"), "first bloom filter differs from the reference's (L0)"
    if e["filter_mode"]:
        assert same_filter(g.bloom_bytes(1), "bf_high"), "second bloom filter (bfc -1) differs from the reference's"
        return st
    t = g.export_table()
    assert t.count() == e["distinct"] == st["n_keys"]
    mode, cnt, high = t.hist()
    assert int(mode) == e["hist_mode"]
    if "cnt" in e:
        assert np.array_equal(cnt, np.array(e["cnt"], dtype=np.uint64)) and np.array_equal(high, np.array(e["high"], dtype=np.uint64))
    else:
        assert [int(v) for v in cnt[1:5]] == e["cnt_1_4"] and [int(v) for v in high[0:3]] == e["high_0_2"]
    if l1:
        sizes, slots = t.export_sorted()
        assert oracle.l1_digest(sizes, slots) == e["l1_digest"], "count table differs from the reference's (L1)"
    t.close()
    return st


@pytest.mark.parametrize("k,b", [(31, 30), (33, 33)])
@pytest.mark.parametrize("n_batches", [1, 3])
def test_g42_reference_goldens(gpu_lib, g42, k, b, n_batches):
    """g42 (1 Mbp x 30) at k=31/-b30 (L1 10c69c17...) and k=33/-b33 (37e9a864...): the default 1 GiB filter geometry."""
    rs, _ = g42
    e = [x for x in FIX["fixtures"] if x["fixture"] == "g42" and (x["k"], x["b"], x["filter_mode"]) == (k, b, 0)][0]
    assert e["l1_digest"] == {(31, 30): "10c69c17d5625df5b9fc7530c94b2e66", (33, 33): "37e9a864777226c76bff184bf03ef779"}[(k, b)]
    g = _count_fixed(gpu_lib, rs, k, b, (rs.n_reads + n_batches - 1) // n_batches)
    _check_against(g, e)
    g.close()


def _wc_launches(gpu_lib):
    """launches of k_scatter1_wc by this process so far (a process-wide counter behind any context)"""
    g = gpu_lib.GpuCounter(33, 30, max_batch_pos=64)
    v = g.s1wc_launches()
    g.close()
    return v


@pytest.mark.parametrize("batch_reads,layout", [(786432, 0), (1572864, 0), (786432, 1)])
def test_c2_full_read_set(gpu_lib, batch_reads, layout):
    """Config c2 as bench.py runs it (k=31, -b33, 3.07 M reads at 100x, 4 or 2 batches): equals the reference on the same reads."""
    e = BASE["c2"]
    rs = gen.ReadSet(**e["gen"])
    g = _count_fixed(gpu_lib, rs, e["k"], e["b"], batch_reads, table_layout=layout)
    assert g.table_info()["segments"] == (layout == 0)  # c2's geometry takes the region-owned table segments (46 identity bits)
    _check_against(g, e)
    assert g.partition_info() == dict(one_pass=True, level2_one_pass=True, replayed_batches=0)  # K1 once per batch, no histogram passes, no slab overflow on uniformly hashed k-mers
    g.close()


def test_c3_full_read_set(gpu_lib):
    """Config c3 as bench.py runs it (k=33, -b35 table mode, 49.6 M reads: 9+9 scatter levels, 16-byte records, the batches'
    k-mers hardly repeat so the context switches to STREAM mode, the table grows to 300 M keys): equals the reference."""
    e = BASE["c3"]
    rs = gen.ReadSet(**e["gen"])
    wc0 = _wc_launches(gpu_lib)
    g = _count_fixed(gpu_lib, rs, e["k"], e["b"], 2_883_584)
    assert g.s1wc_launches() > wc0, "c3's level 1 is expected to run k_scatter1_wc (round 5: write-combining buffers in LDS), not the tile kernel"
    ti = g.table_info()
    assert ti["segments"] and ti["seg_growths"] >= 1, ti  # 48 identity bits; the segments grow with the 300 M keys
    st = _check_against(g, e)
    assert st["stream_batches"] > 0, "c3 is expected to run (mostly) without in-LDS aggregation"
    assert g.partition_info() == dict(one_pass=True, level2_one_pass=True, replayed_batches=0)
    g.close()


def test_c3_shape_host_layout(gpu_lib):
    """The same geometry with the table in the host's layout from the start (random-CAS upserts, STREAM decisions, table growth by
    rehash) on the first 3 M reads; compared with the region-owned layout run, which the test above pins to the reference."""
    e = dict(gen=dict(seed=3, G=248_000_000, cov=30.0), k=33, b=35)
    rs = gen.ReadSet(**e["gen"])
    rs.n_reads = 3_000_000  # (12 M until round 5: 43 s of the suite; two batches and a table growth are what the comparison needs)
    res = []
    for layout in (0, 1):
        g = _count_fixed(gpu_lib, rs, e["k"], e["b"], 2_883_584, table_layout=layout)
        st = g.stats()
        t = g.export_table()
        sizes, slots = t.export_sorted()
        res.append((st["n_kmers"], st["n_high"], st["n_seen"], st["n_keys"], gen.bitmap_checksums(g.bloom_bytes()), oracle.l1_digest(sizes, slots)))
        t.close(); g.close()
    assert res[0] == res[1]


@pytest.mark.skipif("c4e" in BASE, reason="c4's geometry is held to the reference on an eighth of c4 itself (test_c4_eighth_full_geometry); this smaller read set only where that golden is absent")
def test_c4_parameters(gpu_lib):
    """`-s 3g`: k=33, -b37 in table mode (16 GiB filter, 2^20 regions, 10+10 scatter levels) on a 20 Mbp genome at 30x."""
    e = BASE["c4s"]
    rs = gen.ReadSet(**e["gen"])
    g = _count_fixed(gpu_lib, rs, e["k"], e["b"], 2_000_000)
    _check_against(g, e)
    g.close()


@pytest.mark.skipif("c4e" not in BASE, reason="tests/golden/baseline.json has no c4e entry (make_baseline_goldens.py c4e: 1.7 h of one core)")
def test_c4_eighth_full_geometry(gpu_lib):
    """An EIGHTH of config c4 itself: 77.5 M reads of a 387.5 Mbp genome at 30x into c4's own geometry (`-s 3g`: k=33, -b37 --
    16 GiB filter, 2^20 regions, 10+10 scatter levels, table segments that grow to 64 KiB), in calls of 16 M reads as scripts/c4_run.py and
    bench.py's secondary `c4e` submit them: totals, distinct keys, both histograms and the filter's popcount + FNV-1a equal the reference's
    (tests/golden/baseline.json[c4e]; the L1 digest of 470 M slots is left to the smaller shapes)."""
    e = BASE["c4e"]
    rs = gen.ReadSet(**e["gen"])
    wc0 = _wc_launches(gpu_lib)
    g = _count_fixed(gpu_lib, rs, e["k"], e["b"], 16_777_216)
    assert g.s1wc_launches() > wc0, "c4's 2^10 level-1 buckets are expected to take k_scatter1_wc"
    _check_against(g, e, l1=False)
    assert g.partition_info() == dict(one_pass=True, level2_one_pass=True, replayed_batches=0)
    g.close()


@pytest.mark.skipif("c4e_k55" not in BASE, reason="tests/golden/baseline.json has no c4e_k55 entry (make_baseline_goldens.py c4e_k55)")
def test_c4_eighth_at_the_published_k55(gpu_lib):
    """The reference's one published command line, `bfc -s 3g -k55` (tex/README.md:26, tex/bfc.tex:189), on the eighth of 30x human: k=55, -b37, TABLE
    mode -- 20-byte records, 2^24 sub-tables after the clamp (htab.c:19-34), the lossy key of k >= 38 (htab.c:45-58: different k-mers may share a
    slot, their counts add).  Totals, distinct keys, both histograms and the filter equal the reference's (tests/golden/baseline.json[c4e_k55])."""
    e = BASE["c4e_k55"]
    rs = gen.ReadSet(**e["gen"])
    g = _count_fixed(gpu_lib, rs, e["k"], e["b"], 16_777_216)
    assert g.mg_info()["rec_bytes"] == 20
    _check_against(g, e, l1=False)
    g.close()


@pytest.mark.skipif("c5e" in BASE, reason="c5's geometry is held to the reference on an eighth of c5 itself (test_c5_eighth_full_geometry_count_and_trim); this smaller read set only where that golden is absent")
def test_c5_parameters(gpu_lib):
    """`-s 3g -k51 -1`: k=51, -b37, filter mode (two 16 GiB filters with both slices of a region in LDS, 20-byte records)."""
    e = BASE["c5s"]
    rs = gen.ReadSet(**e["gen"])
    g = _count_fixed(gpu_lib, rs, e["k"], e["b"], 2_000_000, filter_mode=1)
    _check_against(g, e)
    g.close()


@pytest.mark.skipif("c5e" not in BASE, reason="tests/golden/baseline.json has no c5e entry (make_baseline_goldens.py c5e: ~1.2 h of one core)")
def test_c5_eighth_full_geometry_count_and_trim(gpu_lib):
    """An EIGHTH of config c5 itself: the 77.5 M reads of c4e at c5's parameters -- `-s 3g -k51 -1`: k=51, -b37, two 16 GiB filters,
    2^20 regions, 10+10 scatter levels, 16-byte records, k_bloom3fm -- in calls of 16 M reads as scripts/c4_run.py and bench.py's secondary `c5e`
    submit them: k-mer / high / seen totals and BOTH filters' popcount + FNV-1a equal the reference's (tests/golden/baseline.json[c5e])."""
    e = dict(BASE["c5e"])
    t = BASE.get("c5e_trim")
    if t is not None and t.get("reproduces") == "c5e":  # the re-derivation of this entry carries the filters' parallel digests (same filters: it reproduced their FNV-1a)
        assert (t["bf_fnv1a64"], t["bf_high_fnv1a64"]) == (e["bf_fnv1a64"], e["bf_high_fnv1a64"])
        e.update({f: t[f] for f in ("bf_mix64", "bf_high_mix64") if f in t})
    rs = gen.ReadSet(**e["gen"])
    wc0 = _wc_launches(gpu_lib)
    g = _count_fixed(gpu_lib, rs, e["k"], e["b"], 16_777_216, filter_mode=1)
    assert g.mg_info()["rec_bytes"] == 16
    assert g.s1wc_launches() > wc0, "c5's 16-byte records at 2^10 buckets are expected to take k_scatter1_wc"

    _check_against(g, e)
    assert g.stats()["slow_buckets"] == 0
    # ---- config c5's QUERY pass on c5's own workload and geometry: the second filter stays in HBM as bfc_count leaves it for
    # bfc_correct, the trim context adopts it, and the windows of all 77.5 M reads -- k_query4 (one bfc_bf_get per k-mer, bbf.c:47-63) + k_streak
    # (max_streak + keep rule, correct.c:478-497,557-567) -- are those the reference's own worker_ec gave against the reference's bf_high
    # (tests/golden/baseline.json[c5e_trim], tests/golden/make_baseline_goldens.py)
    if t is None:
        g.close()
        pytest.skip("tests/golden/baseline.json has no c5e_trim entry (make_baseline_goldens.py c5e_trim)")
    assert t["reproduces"] == "c5e" and (t["bf_high_popcount"], t["bf_high_fnv1a64"]) == (e["bf_high_popcount"], e["bf_high_fnv1a64"])
    bf = g.export_bloom(1, resident=True)
    g.close()
    br, stride = 8_388_608, rs.L + 1
    tr = gpu_lib.GpuTrimmer(e["k"], bf, max_pos=br * stride, max_reads=br)
    assert tr.adopted, "the trim context is expected to adopt the filter the count pass left in HBM"
    kept = bases = 0
    h = 0xcbf29ce484222325
    G = gen._L()
    for r0 in range(0, rs.n_reads, br):
        r1 = min(rs.n_reads, r0 + br)
        seq, _, _ = rs.reads(r0, r1)
        st_, en_ = tr.trim(gen.to_stream(seq, rs.L, 10), np.arange(r1 - r0 + 1, dtype=np.uint64) * np.uint64(stride), t["min_frac"])
        m = st_ >= 0
        kept += int(m.sum()); bases += int((en_[m] - st_[m]).sum())
        pairs = np.ascontiguousarray(np.stack([st_, en_], axis=1), dtype="<i4")
        h = int(G.bfcgen_fnv1a64_from(h, pairs.ctypes.data, pairs.nbytes))
    tr.close(); bf.close()
    assert (kept, bases) == (t["trim_reads_kept"], t["trim_bases_kept"])
    assert "%016x" % h == t["trim_windows_fnv1a64"], "the (start, end) windows differ from the reference's"


@pytest.mark.parametrize("fm", [0, 1])
@pytest.mark.parametrize("level", [1, 2])
@pytest.mark.parametrize("pipeline", ["1", "0"])  # two streams (batches up to 2^28 positions) / one stream (larger batches: c3, c4)
def test_one_pass_partition_replays_skewed_batches(gpu_lib, fm, level, pipeline, monkeypatch):
    """Input the one-pass partition cannot take, at -b30 (128 level-1 buckets x 8 slabs, 8192 regions):
      level 1: 300 000 reads of a 400-base genome -- few, often repeated k-mers overflow the level-1 slabs;
      level 2: 80 000 reads of a 50 Mbp genome plus 2 000 copies of one read -- every level-1 slab has room (a repeated k-mer adds 250 records
               to slabs of 13 600), but the regions of the repeated k-mers get 2 000 records more than their slab of ~1 760 holds.
    The batch and the one behind it change nothing on the device, the library replays both through the two-pass partition -- results are the
    oracle's, statistics counted once -- and the rest of the run stays two-pass."""
    monkeypatch.setenv("BFCG_PIPELINE", pipeline)
    rng = np.random.default_rng(77 + fm + 10 * level)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    L = 150
    if level == 1:
        G, n = 400, 300_000
        genome = rng.choice(acgt, G + L)
        pos = rng.integers(0, G, n)
    else:
        G, n = 50_000_000, 82_000  # (one library batch: below the cold batch limit of 8192 regions x the LDS list, whichever bloom kernel sizes it)
        genome = rng.choice(acgt, G + L)
        pos = rng.integers(0, G, n)
        pos[rng.choice(n, 2000, replace=False)] = 12345
    seq = genome[(pos[:, None] + np.arange(L)[None, :])].astype(np.uint8)
    if level == 1:
        err = rng.random(seq.shape) < 0.01
        seq[err] = acgt[rng.integers(0, 4, int(err.sum()))]
    seq = seq.reshape(-1)
    qual = rng.integers(33, 74, len(seq)).astype(np.uint8)
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    k, b = 31, 30
    oc = oracle.Counter(k, b, filter_mode=fm)
    oc.count(seq, qual, off)
    nb = 3 if level == 1 else 1
    per = n // nb + 1
    g = gpu_lib.GpuCounter(k, b, filter_mode=fm, max_batch_pos=per * (L + 1) + 64)
    assert g.partition_info() == dict(one_pass=True, level2_one_pass=True, replayed_batches=0)
    for a in range(0, n, per):
        e = min(n, a + per)
        g.count_host(gen.to_stream(seq[a * L:e * L], L, 10), gen.to_stream(qual[a * L:e * L], L, 33))
    st, ost = g.stats(), oc.stats()
    pi = g.partition_info()
    assert pi["replayed_batches"] >= 1 and not pi["one_pass"], pi
    assert (st["n_kmers"], st["n_high"], st["n_seen"]) == (ost["n_kmers"], ost["n_high"], ost["n_seen"])
    assert np.array_equal(g.bloom_bytes(), oc.bloom_bytes())
    if fm:
        assert np.array_equal(g.bloom_bytes(1), oc.bloom_bytes(True))
    else:
        sizes, slots = g.export_table().export_sorted()
        osz, osl = oc.export()
        assert np.array_equal(sizes, osz) and np.array_equal(slots, osl)
    g.reset()  # the next data set starts with the one-pass partition again
    assert g.partition_info()["one_pass"]
    g.close(); oc.close()


@pytest.mark.parametrize("chunk", ["32", "1"])
def test_level1_run_longer_than_its_slab(gpu_lib, monkeypatch, chunk):
    """Low-complexity reads (homopolymers, dinucleotide repeats) put EVERY k-mer of a tile into one level-1 bucket: a run of up to 4096 records
    against slabs of ~1150 (-b30 with batches of 520 K positions).  The run finds its slab full, the batch is poisoned and replayed through the
    two-pass partition -- and meanwhile the run is still stored from the slab's start on, across the following slabs; behind the LAST slab
    that needs the slack recs1 is allocated with (without it the stray stores left the buffer).  Results are the oracle's."""
    monkeypatch.setenv("BFCG_ONEPASS_MIN_TILES", "1")
    monkeypatch.setenv("BFCG_S1_CHUNK", chunk)
    rng = np.random.default_rng(4242)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    L, k, b = 150, 31, 30
    units = [b"A", b"C", b"G", b"T", b"AC", b"AG", b"AT", b"CG", b"CT", b"GT", b"ACG", b"AAT", b"ACGT"]
    reads = []
    for u in units:
        r = np.frombuffer((u * (L // len(u) + 1))[:L], dtype=np.uint8)
        reads += [r] * 250
    reads += [rng.choice(acgt, L) for _ in range(200)]
    order = rng.permutation(len(reads))
    seq = np.concatenate([reads[i] for i in order]).astype(np.uint8)
    n = len(reads)
    qual = rng.integers(33, 74, len(seq)).astype(np.uint8)
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    oc = oracle.Counter(k, b)
    oc.count(seq, qual, off)
    g = gpu_lib.GpuCounter(k, b, max_batch_pos=n * (L + 1) + 64)
    assert g.partition_info()["one_pass"]
    g.count_host(gen.to_stream(seq, L, 10), gen.to_stream(qual, L, 33))
    st, ost = g.stats(), oc.stats()
    assert g.partition_info()["replayed_batches"] >= 1
    assert (st["n_kmers"], st["n_high"], st["n_seen"]) == (ost["n_kmers"], ost["n_high"], ost["n_seen"])
    assert np.array_equal(g.bloom_bytes(), oc.bloom_bytes())
    sizes, slots = g.export_table().export_sorted()
    osz, osl = oc.export()
    assert np.array_equal(sizes, osz) and np.array_equal(slots, osl)
    g.close(); oc.close()


def test_c3_shape_has_no_batch_size_cliff(gpu_lib):
    """The first 8.4 M reads of the c3 read set into an empty -b35 filter (k=33), handed over in calls of 1 / 2 / 4 / 8 M reads: the library cuts
    what a region's LDS list cannot take, so no region ever falls onto the slow path (round 1 at 4 M: every region, 0.66 instead of 0.45 s), the
    results do not depend on the call size, and a call size costs what its passes over filter and table segments cost: 8 / 4 / 4 / 3 library
    batches, 70 / 51-88 / 48 / 48 ms of stage time measured (the reason small calls are slower is bytes, not overflow)."""
    e = BASE["c3"]
    rs = gen.ReadSet(**e["gen"])
    n = 8_388_608
    seq, qual, _ = rs.reads(0, n)
    s, q = gen.to_stream(seq, rs.L, 10), gen.to_stream(qual, rs.L, 33)
    del seq, qual
    stride = rs.L + 1
    res = {}
    for call in (1_048_576, 2_097_152, 4_194_304, 8_388_608):
        g = gpu_lib.GpuCounter(e["k"], e["b"], max_batch_pos=call * stride + 64)
        for a in range(0, n, call):
            g.count_host(s[a * stride:(a + call) * stride], q[a * stride:(a + call) * stride])
        st = g.stats()
        ms, launches = g.stage_ms()
        assert st["slow_buckets"] == 0, (call, st["slow_buckets"])
        res[call] = (st["n_kmers"], st["n_high"], st["n_seen"], st["n_keys"], ms["total"], launches)
        g.close()
    ref = res[8_388_608]
    for call, r in res.items():
        assert r[:4] == ref[:4], (call, r, ref)   # batch boundaries never change results
    t = {c: r[4] for c, r in res.items()}
    print("GPU ms for 8.4 M reads by call size:", {c: round(v, 1) for c, v in t.items()}, "library batches:", {c: r[5] for c, r in res.items()})
    # the passes over filter and table segments are what a call size costs: no more of them than calls, or than the list capacity asks for
    for call, r in res.items():
        assert r[5] <= max(n // call, 3) + 1, (call, r[5])
    assert t[1_048_576] <= 5 * t[8_388_608], t   # (a sanity bound only: stage times between events include whatever the host delays)


def test_c1_read_set_device_api(gpu_lib):
    """Config c1 (BASELINE.json configs[0]: 1x coverage, k=31, the default -b33) through the device API: equals the reference's answers."""
    e = BASE["c1"]
    rs = gen.ReadSet(**e["gen"])
    g = _count_fixed(gpu_lib, rs, e["k"], e["b"], rs.n_reads)
    _check_against(g, e)
    g.close()


@pytest.mark.parametrize("fm", [0, 1])
def test_clean_batch_followed_by_an_overflowing_one(gpu_lib, monkeypatch, fm):
    """Stage A of batch t+1 runs beside stage B of batch t (two streams).  A level-1 slab overflow in t+1 must not touch the clean batch t: the
    overflow flags are per batch slot, and only stage B's own stream makes the poison sticky (round 2 kept ONE flag word: k_bloom / k_commit_seg of
    batch t saw the flag raised by t+1's stage A, returned early, and the replay then applied t twice).  Behind the poisoned batch comes a batch too
    small for the one-pass partition: it is enqueued before the host knows about the overflow and must wait for the replay all the same.
    Device-resident inputs, no host synchronisation between the calls."""
    monkeypatch.setenv("BFCG_PIPELINE", "1")
    rng = np.random.default_rng(4242 + fm)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    L, k, b = 150, 31, 30
    n1, n2, n3 = 300_000, 150_000, 1_500
    genome = rng.choice(acgt, 40_000_000 + L)
    small = rng.choice(acgt, 400 + L)
    p1 = rng.integers(0, 40_000_000, n1); p2 = rng.integers(0, 400, n2); p3 = rng.integers(0, 40_000_000, n3)
    seq = np.concatenate([genome[p1[:, None] + np.arange(L)[None, :]], small[p2[:, None] + np.arange(L)[None, :]], genome[p3[:, None] + np.arange(L)[None, :]]]).astype(np.uint8).reshape(-1)
    qual = rng.integers(33, 74, len(seq)).astype(np.uint8)
    n = n1 + n2 + n3
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    oc = oracle.Counter(k, b, filter_mode=fm)
    oc.count(seq, qual, off)
    stride = L + 1
    g = gpu_lib.GpuCounter(k, b, filter_mode=fm, max_batch_pos=n1 * stride + 64)
    s, q = gen.to_stream(seq, L, 10), gen.to_stream(qual, L, 33)
    d_s, d_q = g.dev_alloc(len(s)), g.dev_alloc(len(q))
    g.h2d(d_s, s); g.h2d(d_q, q)
    for rep in range(2):  # (the second data set starts with the one-pass partition again)
        assert g.partition_info()["one_pass"]
        o = 0
        for cnt in (n1, n2, n3):
            g.count_dev(d_s + o * stride, d_q + o * stride, cnt * stride)
            o += cnt
        st, ost = g.stats(), oc.stats()
        pi = g.partition_info()
        # the overflowing batch -- and, where nothing drains the pipeline in between (no table to grow in filter mode), the small batch enqueued behind it
        assert pi["replayed_batches"] >= (2 if fm else 1) * (rep + 1) and not pi["one_pass"], pi
        assert (st["n_kmers"], st["n_high"], st["n_seen"]) == (ost["n_kmers"], ost["n_high"], ost["n_seen"])
        assert np.array_equal(g.bloom_bytes(), oc.bloom_bytes())
        if fm:
            assert np.array_equal(g.bloom_bytes(1), oc.bloom_bytes(True))
        else:
            sizes, slots = g.export_table().export_sorted()
            osz, osl = oc.export()
            assert np.array_equal(sizes, osz) and np.array_equal(slots, osl)
        g.reset()
    g.dev_free(d_s); g.dev_free(d_q)
    g.close(); oc.close()


def test_segments_larger_than_lds_stay_region_owned(gpu_lib):
    """A genome that is LARGE for its filter (-b22: 32 bloom regions; 1.2 Mbp at 6x: ~40 000 keys per region): the table's segments pass 2^14
    slots -- what a CU's LDS holds -- and become several blocks, one workgroup each, instead of leaving the region-owned layout for the host's
    and random device-scope CAS as before round 4 (SURVEY 8a htab.c:60-82 through LDS at any table size).  Bit-exact against the oracle."""
    rs = gen.ReadSet(seed=77, G=1_200_000, cov=6)
    seq, qual, off = rs.reads()
    k, b = 27, 22  # (2k - 5 region bits = 49 identity bits: within the 50 a slot's key field holds)
    oc = oracle.Counter(k, b)
    oc.count(seq, qual, off)
    g = gpu_lib.GpuCounter(k, b, max_batch_pos=(len(seq) + rs.n_reads) // 3 + 4096)
    per = (rs.n_reads + 2) // 3
    for i in range(0, rs.n_reads, per):
        j = min(rs.n_reads, i + per)
        o = off[i:j + 1] - off[i]
        g.count_host(gpu_lib.to_stream(seq[int(off[i]):int(off[j])], o), gpu_lib.to_stream(qual[int(off[i]):int(off[j])], o))
    st, ost = g.stats(), oc.stats()
    ti = g.table_info()
    assert ti["segments"] and ti["seg_shift"] > 14, ti   # (eight blocks of 2^12 slots and more per region)
    assert (st["n_kmers"], st["n_high"], st["n_seen"]) == (ost["n_kmers"], ost["n_high"], ost["n_seen"])
    assert np.array_equal(g.bloom_bytes(), oc.bloom_bytes())
    sizes, slots = g.export_table().export_sorted()
    osz, osl = oc.export()
    assert st["n_keys"] == len(osl) > 14 * 32 * 1024   # (more than 2^14 slots' worth of keys per region)
    assert np.array_equal(sizes, osz) and np.array_equal(slots, osl)
    g.close(); oc.close()
