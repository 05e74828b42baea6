"""Multi-rank PROTOCOL (tests/mg_protocol.py: a Python restatement of the owner-computes exchange, test infrastructure -- the product's
exchange is C, bfc_amd/csrc/bfcg_mg.hip, covered by tests/test_gpu_group.py).  CPU (-m "not gpu"): a world_size-2 gloo run of that
exchange with an oracle-based engine pins WHAT is exchanged and in which order the owner applies it.  GPU (-m gpu): N ranks emulated
on one device (LocalCluster) through the library's real stage A / stage B kernels."""
import os
import sys

import numpy as np
import pytest

import oracle
from bfc_amd import gen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K, B = 31, 22


def _shares(seq, qual, off, n_reads, n_batches, world):
    """global batch t -> [rank 0's contiguous share, rank 1's, ...] as (seq, qual, off) triples"""
    per_b = (n_reads + n_batches - 1) // n_batches
    out = []
    for t in range(n_batches):
        lo, hi = t * per_b, min(n_reads, (t + 1) * per_b)
        per_r = (hi - lo + world - 1) // world
        row = []
        for r in range(world):
            a, b = min(hi, lo + r * per_r), min(hi, lo + (r + 1) * per_r)
            row.append((seq[int(off[a]):int(off[b])].copy(), qual[int(off[a]):int(off[b])].copy(), (off[a:b + 1] - off[a]).copy()))
        out.append(row)
    return out


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import mg_protocol as bdist
    from mg_cpu_engine import CpuEngine
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    rs = gen.ReadSet(seed=11, G=20000, cov=6)
    seq, qual, off = rs.reads()
    eng = CpuEngine(rank, world, K, B)
    bdist.CHUNK = 4099  # force several point-to-point messages per peer
    for row in _shares(seq, qual, off, rs.n_reads, 3, world):
        seg = bdist.count_batch(eng, row[rank], None, 0)
        assert seg.shape == (world, eng.nb1 // world)
    bits, sizes, slots, n_seen = eng.result()
    np.savez(os.path.join(tmp, "r%d.npz" % rank), bits=bits, sizes=sizes, slots=slots, n_seen=n_seen)
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_two_ranks_equal_sequential(tmp_path):
    """Owner-computes over a real all-to-all (gloo, 2 processes): OR of the owners' bitmaps and union of their tables equal
    the sequential oracle on the same reads (rank-major shares of each batch = plain file order)."""
    import torch.multiprocessing as mp
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    rs = gen.ReadSet(seed=11, G=20000, cov=6)
    seq, qual, off = rs.reads()
    oc = oracle.Counter(K, B)
    oc.count(seq, qual, off)
    parts = [np.load(str(tmp_path / ("r%d.npz" % r))) for r in range(2)]
    assert not np.any(parts[0]["bits"] & parts[1]["bits"]), "owners must touch disjoint blocks"
    assert np.array_equal(parts[0]["bits"] | parts[1]["bits"], oc.bloom_bytes())
    assert int(parts[0]["n_seen"]) + int(parts[1]["n_seen"]) == oc.stats()["n_seen"]
    osz, osl = oc.export()
    assert np.array_equal(parts[0]["sizes"].astype(np.int64) + parts[1]["sizes"], osz)
    assert np.array_equal(np.sort(np.concatenate([parts[0]["slots"], parts[1]["slots"]])), np.sort(osl))
    # no (sub-table, key) pair is held by both owners: a key lives with the rank that owns its bloom region (keys alone may coincide, across sub-tables)
    pairs = [np.repeat(np.arange(len(p["sizes"]), dtype=np.uint64), p["sizes"].astype(np.int64)) << np.uint64(50) | (p["slots"] >> np.uint64(14)) for p in parts]
    assert len(pairs[0]) == len(parts[0]["slots"]) and len(np.intersect1d(pairs[0], pairs[1])) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("n_ranks,k,b", [(2, 33, 30), (4, 31, 28), (8, 51, 30)])
def test_local_cluster_matches_reference_goldens(gpu_lib, g1, n_ranks, k, b):
    """N ranks on one device through the real kernels: bitmap slices concatenate to the sequential bitmap (L0) and the
    union of the per-rank tables is L1-identical to `bfc -t1` (goldens / oracle)."""
    import mg_protocol as bdist
    rs, (seq, qual, off) = g1
    n = 3000 if k != 33 else rs.n_reads
    seq, qual, off = seq[:n * rs.L], qual[:n * rs.L], off[:n + 1]
    cl = bdist.LocalCluster(gpu_lib, n_ranks, k, b, max_batch_pos=(n // 2 + 64) * (rs.L + 1))
    for row in _shares(seq, qual, off, n, 3, n_ranks):
        cl.batch([(gpu_lib.to_stream(s, o), gpu_lib.to_stream(q, o)) for s, q, o in row])
    oc = oracle.Counter(k, b)
    oc.count(seq, qual, off)
    assert np.array_equal(cl.bloom_bytes(), oc.bloom_bytes())
    st, ost = cl.stats(), oc.stats()
    assert (st["n_kmers"], st["n_high"], st["n_seen"]) == (ost["n_kmers"], ost["n_high"], ost["n_seen"])
    sizes, slots = cl.export_sorted()
    osz, osl = oc.export()
    assert np.array_equal(sizes, osz) and np.array_equal(slots, osl)
    if k == 33:
        assert oracle.l1_digest(sizes, slots) == "896ce4092ccc51498b446e7d7775f10c"  # SURVEY C.5 golden, g1/k33/b30
    cl.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_ranks,limit", [(4, 1), (8, 20000), (2, 1)])
def test_local_cluster_source_groups(gpu_lib, g1, n_ranks, limit):
    """A rank that receives more k-mers of a global batch than its regions take at full speed runs stage B once per group of sources
    (dist.process_in_groups; limit 1 = one group per source): rank-major order is file order, so filter, statistics and table are still
    those of the sequential oracle."""
    import mg_protocol as bdist
    rs, (seq, qual, off) = g1
    n, k, b = rs.n_reads, 31, 28
    cl = bdist.LocalCluster(gpu_lib, n_ranks, k, b, max_batch_pos=(n // 2 + 64) * (rs.L + 1), kmer_limit=limit)
    for row in _shares(seq, qual, off, n, 2, n_ranks):
        cl.batch([(gpu_lib.to_stream(s, o), gpu_lib.to_stream(q, o)) for s, q, o in row])
    assert cl.launches > 2 * n_ranks, cl.launches  # more stage-B launches than (batches x owners)
    oc = oracle.Counter(k, b)
    oc.count(seq, qual, off)
    assert np.array_equal(cl.bloom_bytes(), oc.bloom_bytes())
    st, ost = cl.stats(), oc.stats()
    assert (st["n_kmers"], st["n_high"], st["n_seen"]) == (ost["n_kmers"], ost["n_high"], ost["n_seen"])
    sizes, slots = cl.export_sorted()
    osz, osl = oc.export()
    assert np.array_equal(sizes, osz) and np.array_equal(slots, osl)
    cl.close(); oc.close()


@pytest.mark.gpu
def test_local_cluster_stream_mode(gpu_lib):
    """Emulated ranks on batches whose k-mers hardly repeat: every rank's context switches to the STREAM hand-over; still the oracle's result."""
    import mg_protocol as bdist
    rs = gen.ReadSet(seed=11, G=2_000_000, cov=9)
    seq, qual, off = rs.reads()
    n, k, b, world = rs.n_reads, 33, 28, 4
    cl = bdist.LocalCluster(gpu_lib, world, k, b, max_batch_pos=(n // 6 + 64) * (rs.L + 1))
    for row in _shares(seq, qual, off, n, 6, world):
        cl.batch([(gpu_lib.to_stream(s, o), gpu_lib.to_stream(q, o)) for s, q, o in row])
    oc = oracle.Counter(k, b)
    oc.count(seq, qual, off)
    assert all(c.stats()["stream_batches"] >= 2 for c in cl.ctx)
    assert np.array_equal(cl.bloom_bytes(), oc.bloom_bytes())
    st, ost = cl.stats(), oc.stats()
    assert (st["n_kmers"], st["n_high"], st["n_seen"]) == (ost["n_kmers"], ost["n_high"], ost["n_seen"])
    sizes, slots = cl.export_sorted()
    osz, osl = oc.export()
    assert np.array_equal(sizes, osz) and np.array_equal(slots, osl)
    cl.close(); oc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_ranks", [2, 4])
def test_local_cluster_exact_dump(gpu_lib, g1, n_ranks, tmp_path):
    """Parity level L2 across GPUs: with order stamps the union of the ranks' tables (bfc_ch_union) dumps to the very bytes of
    `bfc -E -t1 -d` (md5 golden from the reference binary, g1 / k=31 / -b26)."""
    import mg_protocol as bdist
    rs, (seq, qual, off) = g1
    n = rs.n_reads
    cl = bdist.LocalCluster(gpu_lib, n_ranks, 31, 26, max_batch_pos=(n // 3 + 64) * (rs.L + 1), track_order=True)
    for row in _shares(seq, qual, off, n, 3, n_ranks):
        cl.batch([(gpu_lib.to_stream(s, o), gpu_lib.to_stream(q, o)) for s, q, o in row])
    t = cl.export_table()
    fn = str(tmp_path / "u.hash")
    assert t.dump(fn) == 0
    assert oracle.md5_file(fn) == "d686549d10dd4c71243269013119784a"
    t.close(); cl.close()
