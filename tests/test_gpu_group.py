"""GPU tests (-m gpu) of the multi-GPU path IN C (bfcg_group_*, bfc_amd/csrc/bfcg_mg.hip): stage A, the exchange and stage B driven by
the library's own rank threads.  Only one GPU is here, so the ranks of a group are emulated on device 0 (a device may be named several
times; RCCL refuses that, the records then travel by peer copies through the very same bookkeeping), and the RCCL calls are exercised
with a one-rank run that is set up like one process of a multi-process run (unique id, ncclCommInitRank, ncclAllGather of the sizes).
The bar is the usual one: bloom bitmap, statistics and table bit for bit the sequential oracle's for the batches in rank-major order."""
import os

import numpy as np
import pytest

import oracle
from bfc_amd import gen

pytestmark = pytest.mark.gpu


def _oracle(k, b, seq, qual, off, **kw):
    oc = oracle.Counter(k, b, **kw)
    oc.count(seq, qual, off)
    return oc


def _compare(grp, oc, fm=0):
    st, ost = grp.stats(), oc.stats()
    assert (st["n_kmers"], st["n_high"], st["n_seen"]) == (ost["n_kmers"], ost["n_high"], ost["n_seen"]), (st, ost)
    bf = grp.export_bloom(0)
    assert np.array_equal(bf.bytes(), oc.bloom_bytes()), "first filter differs (L0)"
    bf.close()
    if fm:
        bf = grp.export_bloom(1)
        assert np.array_equal(bf.bytes(), oc.bloom_bytes(True)), "second filter differs"
        bf.close()
    else:
        t = grp.export_table()
        sizes, slots = t.export_sorted()
        osz, osl = oc.export()
        assert np.array_equal(sizes, osz) and np.array_equal(slots, osl), "table differs (L1)"
        assert t.count() == st["n_keys"]
        t.close()


@pytest.mark.parametrize("n_ranks", [1, 2, 4, 8])
@pytest.mark.parametrize("k,b,fm", [(31, 26, 0), (33, 24, 0), (51, 25, 1)])
def test_group_host_batches(gpu_lib, g1, n_ranks, k, b, fm):
    """bfcg_group_count_batch_host: the library cuts every global batch into the ranks' shares; 3 global batches of fixture g1."""
    rs, (seq, qual, off) = g1
    oc = _oracle(k, b, seq, qual, off, filter_mode=fm)
    n = rs.n_reads
    grp = gpu_lib.GpuGroup(k, b, [0] * n_ranks, max_batch_pos=(n // 3 + 2) * (rs.L + 1) // n_ranks + 4096, filter_mode=fm)
    assert grp.info()["transport"] == ("push" if n_ranks > 1 else "rccl")  # (ranks that share a device: peer copies, lazy batches by the push kernel)
    for a in range(0, n, n // 3 + 1):
        e = min(n, a + n // 3 + 1)
        grp.count_host(gen.to_stream(seq[a * rs.L:e * rs.L], rs.L, 10), gen.to_stream(qual[a * rs.L:e * rs.L], rs.L, 33))
    _compare(grp, oc, fm)
    grp.close(); oc.close()


@pytest.mark.parametrize("n_ranks", [2, 8] + ([4] if os.environ.get("BFC_TEST_MORE") else []))
def test_group_push_kernel_moves_exactly_the_records(gpu_lib, n_ranks, monkeypatch):
    """Round 6: with the PUSH transport one kernel per rank and global batch writes the filled part of every slab into its
    owner's receive buffer (k_push_slabs: the fills are read on the device, the host still knows no size) -- the bytes on the links are the live
    records' + the rows, where the whole-block copies of round 5 (transport 2) carry the slabs' unfilled ends.  Same filter, statistics and table
    as the oracle's either way; the books (bfcg_group_exchange_bytes) say what travelled: (N - 1) / N of the k-mers x the record size, exactly."""
    rng = np.random.default_rng(4300 + n_ranks)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    L, n = 150, 120_000
    genome = rng.choice(acgt, 30_000_000 + L)
    seq = genome[(rng.integers(0, 30_000_000, n)[:, None] + np.arange(L)[None, :])].astype(np.uint8).reshape(-1)
    seq[rng.integers(0, len(seq), 300)] = ord("N")
    qual = rng.integers(33, 74, len(seq)).astype(np.uint8)
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    k, b = 33, 30
    oc = _oracle(k, b, seq, qual, off)
    books = {}
    for transport in (3, 2):
        grp = gpu_lib.GpuGroup(k, b, [0] * n_ranks, max_batch_pos=60_000 * (L + 1) // n_ranks + 4096, transport=transport)
        assert grp.info()["slab_mode"] and grp.info()["transport"] == {3: "push", 2: "peer"}[transport]
        nb1 = grp.info()["nb1"]
        for t in range(2):
            a, e = t * 60_000, (t + 1) * 60_000
            grp.count_host(gen.to_stream(seq[a * L:e * L], L, 10), gen.to_stream(qual[a * L:e * L], L, 33))
        grp.sync()
        assert grp.info()["lazy_batches"] == 2
        books[transport] = grp.exchange_bytes()
        _compare(grp, oc, 0)
        grp.reset()
        assert grp.exchange_bytes()["links"] == 0
        grp.close()
    rb = 12
    n_kmers = oc.stats()["n_kmers"]
    ex = books[3]["exact"]
    assert ex == books[2]["exact"] and ex % rb == 0
    # the hash deals the k-mers evenly: (N - 1) / N of them leave their rank
    assert abs(ex / rb - n_kmers * (n_ranks - 1) / n_ranks) < 0.02 * n_kmers
    rows = 2 * n_ranks * (n_ranks - 1) * 4 * (nb1 // n_ranks * 8 + 2)  # (2 batches; a row = the fills of a destination's nb1 / N x 8 slabs + 2 words)
    assert books[3]["links"] == ex + rows, (books[3], rows)
    assert books[2]["links"] > 1.2 * ex, "whole blocks carry the slabs' unfilled ends"
    oc.close()


@pytest.mark.parametrize("n_ranks", [2, 4] + ([1] if os.environ.get("BFC_TEST_MORE") else []))  # (a group of one: bench.py through the group path, below)
@pytest.mark.parametrize("lazy", [1, 0])
def test_group_sizes_stay_on_the_device(gpu_lib, n_ranks, lazy, monkeypatch):
    """Round 5: with every rank in one process the slabs' fills travel beside the blocks as rows in device memory and the owner builds its
    segment arrays from them on the device (k_pack_rows / k_seg_setup_mg): exchange and stage B of a global batch are enqueued before the host
    has seen a size (`lazy_batches` counts them).  BFCG_MG_LAZY=0 is round 4's protocol (the host waits for stage A's sizes first).  Either way:
    the oracle's filter, statistics and table -- 3 global batches of 60 000 reads of a 50 Mbp genome at -b30 (slabs fill evenly: no overflow),
    the middle one FASTA (no qualities), one rank's share of the last one empty; then the same again after a reset."""
    monkeypatch.setenv("BFCG_MG_LAZY", str(lazy))
    rng = np.random.default_rng(4100 + n_ranks)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    L, n = 150, 180_000
    genome = rng.choice(acgt, 50_000_000 + L)
    seq = genome[(rng.integers(0, 50_000_000, n)[:, None] + np.arange(L)[None, :])].astype(np.uint8).reshape(-1)
    qual = rng.integers(33, 74, len(seq)).astype(np.uint8)
    qual[60_000 * L:120_000 * L] = 126  # (what the oracle sees for a record without qualities: every base high quality)
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    k, b = 33, 30
    oc = _oracle(k, b, seq, qual, off)
    grp = gpu_lib.GpuGroup(k, b, [0] * n_ranks, max_batch_pos=60_000 * (L + 1) // n_ranks + 4096)
    assert grp.info()["slab_mode"]
    for rnd in range(2):
        for t in range(3):
            a, e = t * 60_000, (t + 1) * 60_000
            grp.count_host(gen.to_stream(seq[a * L:e * L], L, 10), None if t == 1 else gen.to_stream(qual[a * L:e * L], L, 33))
        grp.sync()
        assert grp.info()["slab_mode"]
        assert grp.info()["lazy_batches"] == (3 * (rnd + 1) if lazy else 0), grp.info()
        _compare(grp, oc)
        grp.reset()
    grp.close(); oc.close()


def test_group_lazy_sizes_with_overloaded_owners(gpu_lib):
    """The lazy protocol where an owner receives more than its regions take at full speed (-b30 over 4 ranks: 2048 regions per rank, a global
    batch of 150 000 reads brings each ~4.4 M k-mers): the FIRST group of sources -- chosen by what they can send at most -- is applied from the
    rows on the device before the host has a size, the others in further passes grouped by their sizes.  Same result as one batch."""
    rng = np.random.default_rng(4242)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    L, per, nb, N = 150, 150_000, 3, 4
    n = per * nb
    genome = rng.choice(acgt, 60_000_000 + L)
    seq = genome[(rng.integers(0, 60_000_000, n)[:, None] + np.arange(L)[None, :])].astype(np.uint8).reshape(-1)
    qual = rng.integers(33, 74, len(seq)).astype(np.uint8)
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    k, b = 33, 30
    oc = _oracle(k, b, seq, qual, off)
    grp = gpu_lib.GpuGroup(k, b, [0] * N, max_batch_pos=per * (L + 1) // N + 4096)
    launches = []
    for t in range(nb):
        a, e = t * per, (t + 1) * per
        grp.count_host(gen.to_stream(seq[a * L:e * L], L, 10), gen.to_stream(qual[a * L:e * L], L, 33))
        grp.sync()
        launches.append([grp.ctx(i).stage_ms()[1] for i in range(N)])
    assert grp.info()["slab_mode"] and grp.info()["lazy_batches"] == nb, grp.info()
    assert all(v >= 2 for v in launches[0]), launches  # the cold batch: several passes per rank
    assert grp.stats()["slow_buckets"] == 0
    _compare(grp, oc)
    grp.close(); oc.close()


def test_group_device_shares_uneven(gpu_lib, g1):
    """bfcg_group_count_batch_dev with ragged shares, a rank that contributes nothing, FASTA (no qualities) and a second pass after reset"""
    rs, (seq, qual, off) = g1
    k, b, N = 27, 25, 4
    n = rs.n_reads
    grp = gpu_lib.GpuGroup(k, b, [0] * N, max_batch_pos=n * (rs.L + 1) + 64)
    ctx = [grp.ctx(i) for i in range(N)]
    for rep in range(2):
        oc = _oracle(k, b, seq, None, off)
        cuts = [0, n // 7, n // 7, n // 2, n]  # rank 1 gets nothing
        ptrs, lens = [], []
        for i in range(N):
            s = gen.to_stream(seq[cuts[i] * rs.L:cuts[i + 1] * rs.L], rs.L, 10)
            d = ctx[i].dev_alloc(max(len(s), 16))
            if len(s):
                ctx[i].h2d(d, s)
            ptrs.append(d); lens.append(len(s))
        grp.count_dev(ptrs, None, lens)
        grp.sync()
        _compare(grp, oc)
        for i in range(N):
            ctx[i].dev_free(ptrs[i])
        oc.close()
        grp.reset()
    grp.close()


def test_group_rccl_single_rank_as_one_process_of_many(gpu_lib, g1):
    """The multi-process set-up with a world of one: unique id -> ncclCommInitRank, the sizes through ncclAllGather, stage B behind the
    exchange stream's event.  (ncclSend / ncclRecv need a second device: the driver's multi-GPU bench is their first run.)"""
    rs, (seq, qual, off) = g1
    k, b = 31, 26
    oc = _oracle(k, b, seq, qual, off)
    uid = gpu_lib.GpuGroup.unique_id()
    assert len(uid) == 128 and any(uid)
    grp = gpu_lib.GpuGroup(k, b, [0], max_batch_pos=rs.n_reads * (rs.L + 1) + 64, n_ranks=1, first_rank=0, uid=uid)
    assert grp.info()["transport"] == "rccl"
    c = grp.ctx(0)
    s, q = gen.to_stream(seq, rs.L, 10), gen.to_stream(qual, rs.L, 33)
    ds, dq = c.dev_alloc(len(s)), c.dev_alloc(len(q))
    c.h2d(ds, s); c.h2d(dq, q)
    half = (rs.n_reads // 2) * (rs.L + 1)
    grp.count_dev([ds], [dq], [half])
    grp.count_dev([ds + half], [dq + half], [len(s) - half])
    st, ost = grp.stats(), oc.stats()
    assert (st["n_kmers"], st["n_high"], st["n_seen"]) == (ost["n_kmers"], ost["n_high"], ost["n_seen"])
    assert np.array_equal(c.bloom_bytes(), oc.bloom_bytes())
    t = c.export_table()
    sizes, slots = t.export_sorted()
    osz, osl = oc.export()
    assert np.array_equal(sizes, osz) and np.array_equal(slots, osl)
    t.close(); c.dev_free(ds); c.dev_free(dq); grp.close(); oc.close()


@pytest.mark.parametrize("seed", range(12))
def test_group_random_configurations(gpu_lib, seed):
    """the GPU fuzz's draws through groups of 2 / 4 / 8 emulated ranks, every cut of the draw one global batch from host memory"""
    from test_gpu_fuzz import _draw
    prm, seq, qual, off, cuts, kw = _draw(41000 + seed, scale=3)
    rng = np.random.default_rng(seed)
    N = int(rng.choice([2, 4, 8]))
    if prm["b"] < 21:
        prm["b"] = int(rng.integers(21, 27))  # more bloom regions (2^(b-17)) than ranks
    kw.pop("region_shift", None)
    oc = oracle.Counter(prm["k"], prm["b"], q=prm["q"], n_hashes=prm["nh"], l_pre=prm["l_pre"], filter_mode=prm["fm"])
    oc.count(seq, qual, off)
    n = len(off) - 1
    grp = gpu_lib.GpuGroup(prm["k"], prm["b"], [0] * N, max_batch_pos=len(seq) + n + 64, q=prm["q"], n_hashes=prm["nh"], l_pre=prm["l_pre"],
                           filter_mode=prm["fm"], **kw)
    for a, e in zip(cuts[:-1], cuts[1:]):
        if e > a:
            o = off[a:e + 1] - off[a]
            grp.count_host(gpu_lib.to_stream(seq[int(off[a]):int(off[e])], o), gpu_lib.to_stream(qual[int(off[a]):int(off[e])], o) if qual is not None else None)
    _compare(grp, oc, prm["fm"])
    grp.close(); oc.close()


@pytest.mark.parametrize("n_ranks", [2, 4])
@pytest.mark.parametrize("fm", [0, 1])
def test_group_level2_slab_overflow_is_replayed(gpu_lib, n_ranks, fm):
    """A rank of a multi-GPU run partitions what it receives in one pass (a slab per bloom region).  100 000 reads of a 50 Mbp genome plus
    3 000 copies of one read, -b30: the regions of the repeated k-mers get 3 000 records more than their slab holds on whichever rank owns
    them.  That rank replays its own stage B from its receive buffer through the two-pass kernels -- no second exchange -- and the group's
    result is the oracle's."""
    rng = np.random.default_rng(500 + n_ranks + fm)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    L, G, n = 150, 50_000_000, 103_000
    genome = rng.choice(acgt, G + L)
    pos = rng.integers(0, G, n)
    pos[rng.choice(n, 3000, replace=False)] = 4242
    seq = genome[(pos[:, None] + np.arange(L)[None, :])].astype(np.uint8).reshape(-1)
    qual = rng.integers(33, 74, len(seq)).astype(np.uint8)
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    k, b = 31, 30
    oc = _oracle(k, b, seq, qual, off, filter_mode=fm)
    half = n // 2 + 1
    grp = gpu_lib.GpuGroup(k, b, [0] * n_ranks, max_batch_pos=half * (L + 1) // n_ranks + 4096, filter_mode=fm)
    before = [grp.ctx(i).partition_info() for i in range(n_ranks)]
    assert all(p["level2_one_pass"] and not p["one_pass"] for p in before), before
    for a in range(0, n, half):
        e = min(n, a + half)
        grp.count_host(gen.to_stream(seq[a * L:e * L], L, 10), gen.to_stream(qual[a * L:e * L], L, 33))
    _compare(grp, oc, fm)
    after = [grp.ctx(i).partition_info() for i in range(n_ranks)]
    assert sum(p["replayed_batches"] for p in after) >= 1, after
    grp.close(); oc.close()


@pytest.mark.parametrize("n_ranks,fm", [(2, 0), (4, 1)] + ([(4, 0), (2, 1)] if os.environ.get("BFC_TEST_MORE") else []))
def test_group_level1_slab_overflow_falls_back_to_two_passes(gpu_lib, n_ranks, fm):
    """Stage A of a rank is one pass into slabs (round 4).  First a clean global batch (100 000 reads of a 50 Mbp genome: slab mode stays on), then
    200 000 reads of a 400-base genome -- few, often repeated k-mers overflow a level-1 slab on some rank: EVERY rank repeats its stage A of
    that batch through the two-pass partition (its k-mers are not counted twice), the exchange carries exact bucket sizes again, the run stays
    with two passes until the reset -- and a third, clean batch behind it: the oracle's filter(s), statistics and table throughout."""
    rng = np.random.default_rng(900 + n_ranks + fm)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    L = 150
    def reads(G, n, err):
        genome = rng.choice(acgt, G + L)
        s = genome[(rng.integers(0, G, n)[:, None] + np.arange(L)[None, :])].astype(np.uint8)
        if err:
            m = rng.random(s.shape) < err
            s[m] = acgt[rng.integers(0, 4, int(m.sum()))]
        return s.reshape(-1)
    parts = [reads(50_000_000, 100_000, 0), reads(400, 200_000, 0.01), reads(50_000_000, 100_000, 0)]
    seq = np.concatenate(parts)
    n = len(seq) // L
    qual = rng.integers(33, 74, len(seq)).astype(np.uint8)
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    k, b = 31, 30
    oc = _oracle(k, b, seq, qual, off, filter_mode=fm)
    grp = gpu_lib.GpuGroup(k, b, [0] * n_ranks, max_batch_pos=200_000 * (L + 1) // n_ranks + 4096, filter_mode=fm)
    assert grp.info()["slab_mode"]
    o = 0
    for i, p in enumerate(parts):
        e = o + len(p)
        grp.count_host(gen.to_stream(seq[o:e], L, 10), gen.to_stream(qual[o:e], L, 33))
        assert grp.info()["slab_mode"] == (i == 0), (i, grp.info())
        o = e
    _compare(grp, oc, fm)
    grp.reset()
    assert grp.info()["slab_mode"]  # the next data set starts in one pass again
    grp.close(); oc.close()


@pytest.mark.parametrize("fm", [0, 1])
def test_group_overloaded_ranks_process_sources_in_groups(gpu_lib, g1, fm):
    """-b24 over 4 ranks: 32 bloom regions per rank take ~80 000 k-mers per pass at full speed, a global batch brings each rank 200 000.  The
    first batch (empty filter) is applied in groups of sources -- several stage B passes per rank, the result that of one batch because the
    receive buffer is source-major and file order rank-major; the later ones (most k-mers seen again: a region takes three times as many) in
    fewer passes.  No region on the slow path, and the oracle's filter(s), statistics and table."""
    rs, (seq, qual, off) = g1
    k, b, N = 31, 24, 4
    oc = _oracle(k, b, seq, qual, off, filter_mode=fm)
    n = rs.n_reads
    per = n // 3 + 1
    grp = gpu_lib.GpuGroup(k, b, [0] * N, max_batch_pos=per * (rs.L + 1) // N + 4096, filter_mode=fm)
    launches = []
    for a in range(0, n, per):
        e = min(n, a + per)
        grp.count_host(gen.to_stream(seq[a * rs.L:e * rs.L], rs.L, 10), gen.to_stream(qual[a * rs.L:e * rs.L], rs.L, 33))
        grp.sync()
        launches.append([grp.ctx(i).stage_ms()[1] for i in range(N)])
    first = launches[0]
    later = [x - y for x, y in zip(launches[-1], launches[-2])]
    assert all(v >= 2 for v in first), launches                     # the cold batch: several passes, each over some of the sources
    assert all(l < f for l, f in zip(later, first)), launches       # a warm batch of the same size: fewer passes
    assert grp.stats()["slow_buckets"] == 0
    _compare(grp, oc, fm)
    grp.close(); oc.close()


@pytest.mark.parametrize("n_ranks", [2, 4])
def test_group_c2_full_read_set(gpu_lib, n_ranks):
    """The whole c2 read set (3.07 M reads, 367 M k-mers, k=31, -b33) through 2 / 4 emulated ranks in 4 global batches: the statistics, the bloom
    filter (popcount + FNV-1a) and the table (distinct keys, both histograms, L1 digest) THE REFERENCE computed for these reads
    (tests/golden/baseline.json) -- the multi-GPU path at the volumes of a real batch (tens of millions of records per rank and exchange)."""
    import json
    here = os.path.dirname(os.path.abspath(__file__))
    e = {x["name"]: x for x in json.load(open(os.path.join(here, "golden", "baseline.json")))}["c2"]
    rs = gen.ReadSet(**e["gen"])
    per = 786_432
    grp = gpu_lib.GpuGroup(e["k"], e["b"], [0] * n_ranks, max_batch_pos=per * (rs.L + 1) // n_ranks + 4096)
    for r0 in range(0, rs.n_reads, per):
        seq, qual, _ = rs.reads(r0, min(rs.n_reads, r0 + per))
        grp.count_host(gen.to_stream(seq, rs.L, 10), gen.to_stream(qual, rs.L, 33))
    st = grp.stats()
    assert (st["n_kmers"], st["n_high"], st["n_seen"], st["n_keys"]) == (e["n_kmers"], e["n_high"], e["n_seen"], e["distinct"]), st
    bf = grp.export_bloom(0)
    assert gen.bitmap_checksums(bf.bytes()) == (e["bf_popcount"], int(e["bf_fnv1a64"], 16))
    bf.close()
    t = grp.export_table()
    mode, cnt, high = t.hist()
    assert int(mode) == e["hist_mode"] and np.array_equal(cnt, np.array(e["cnt"], dtype=np.uint64)) and np.array_equal(high, np.array(e["high"], dtype=np.uint64))
    assert oracle.l1_digest(*t.export_sorted()) == e["l1_digest"]
    t.close()
    assert all(grp.ctx(i).partition_info()["level2_one_pass"] for i in range(n_ranks))
    grp.close()


def test_group_c4_geometry_on_4_ranks(gpu_lib):
    """c4's parameters (`-s 3g`: k=33, -b37 -- a 16 GiB filter, 2^20 bloom regions, 10 + 10 scatter levels, 12-byte records) on 4 emulated ranks,
    each owning 4 GiB of the filter and a quarter of the table segments: the 20 Mbp x 30 read set (4 M reads, 472 M k-mers) in 3 global batches
    must give what THE REFERENCE computed for it with these parameters (tests/golden/baseline.json[c4s])."""
    import json
    here = os.path.dirname(os.path.abspath(__file__))
    e = {x["name"]: x for x in json.load(open(os.path.join(here, "golden", "baseline.json")))}["c4s"]
    rs = gen.ReadSet(**e["gen"])
    n_ranks, per = 4, 1_400_000
    grp = gpu_lib.GpuGroup(e["k"], e["b"], [0] * n_ranks, max_batch_pos=per * (rs.L + 1) // n_ranks + 65536)
    for r0 in range(0, rs.n_reads, per):
        seq, qual, _ = rs.reads(r0, min(rs.n_reads, r0 + per))
        grp.count_host(gen.to_stream(seq, rs.L, 10), gen.to_stream(qual, rs.L, 33))
    st = grp.stats()
    assert (st["n_kmers"], st["n_high"], st["n_seen"], st["n_keys"]) == (e["n_kmers"], e["n_high"], e["n_seen"], e["distinct"]), st
    bf = grp.export_bloom(0)
    assert gen.bitmap_checksums(bf.bytes()) == (e["bf_popcount"], int(e["bf_fnv1a64"], 16))
    bf.close()
    t = grp.export_table()
    mode, cnt, high = t.hist()
    assert int(mode) == e["hist_mode"] and np.array_equal(cnt, np.array(e["cnt"], dtype=np.uint64)) and np.array_equal(high, np.array(e["high"], dtype=np.uint64))
    assert oracle.l1_digest(*t.export_sorted()) == e["l1_digest"]
    t.close()
    grp.close()


def test_bench_through_the_group_path_is_verified(tmp_path):
    """bench.py as the driver launches it for N > 1 -- under torch.distributed.run, one process per GPU, the RCCL unique id over gloo -- with a
    world of ONE and BFC_BENCH_FORCE_DIST=1: the records take the group path of libbfc_gpu.so (bfcg_group_*: stage A, sizes all-gather and
    exchange on the RCCL communicator, stage B behind the exchange stream's events) and the state the last step leaves behind must be the
    reference's for the workload's read set (tests/golden/baseline.json: "verified")."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BFC_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29533",
                        os.path.join(root, "bench.py"), "--gpus", "1", "--workload", "c2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--no-boundary"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-1500:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(line) == 1, r.stdout[-500:]
    d = json.loads(line[0])
    assert d["verified"] is True, d.get("verification")
    assert "RCCL" in d["config"]["parallelism"] and d["n_gpus"] == 1
    assert d["roofline"]["kernel"].startswith("bloom-insert path") and 0 < d["roofline"]["frac"] < 1


def _bench_inproc(n, devices, workload, extra=()):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k_ in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BFC_BENCH_FORCE_DIST"):
        env.pop(k_, None)
    if devices is not None:
        env["BFC_BENCH_DEVICES"] = devices
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--workload", workload, "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-secondary", "--no-boundary"] + list(extra), capture_output=True, text=True, timeout=1500, env=env, cwd=root)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, (r.stdout[-500:], r.stderr[-1500:])
    return r.returncode, json.loads(lines[0]), r.stderr


@pytest.mark.parametrize("n", [2, 4])
def test_bench_gpus_n_runs_n_ranks_in_process(n):
    """`python bench.py --gpus N` exactly as the driver types it (no launcher): all N ranks in this process through bfcg_group_create(n_local = N),
    one host thread per rank -- here emulated on device 0 (BFC_BENCH_DEVICES, which only this test sets: peer copies instead of RCCL, the same
    bookkeeping) --, strong scaling on ONE read set, and the sums of what the ranks hold are the reference's answers for it ("verified")."""
    rc, d, err = _bench_inproc(n, ",".join(["0"] * n), "c2")
    assert rc == 0, err[-1500:]
    assert d["n_gpus"] == n and d["scaling"] == "strong" and d["verified"] is True, d.get("verification")
    assert "owner computes" in d["config"]["parallelism"] and "NO bloom reduce" in d["config"]["parallelism"] and "peer copies" in d["config"]["parallelism"]
    assert d["value"] > 0 and d["roofline"]["clock"].startswith("hip_events")


def test_bench_gpus_2_on_a_one_gpu_box_fails_loudly():
    """the plain invocation (devices 0 and 1) on a box with ONE device: an "error" line and exit code 2, not a 1-GPU number"""
    from bfc_amd import _lib
    if _lib.load().bfcg_device_count() >= 2:
        pytest.skip("this box has two devices: the plain invocation is a real 2-GPU run here")
    rc, d, err = _bench_inproc(2, None, "c2")
    assert rc == 2 and d["value"] is None and d["n_gpus"] == 2 and "not running on fewer GPUs" in d["error"]
