"""k-mer coverage of the corrector: bfc_ec_kcov, correct.c:96-117.

CPU: the oracle's restatement (orc_kcov) against goldens taken from the reference's own bfc_ec_kcov (fixtures.json "kcov",
made by tests/golden/make_goldens.py through oracle/ref_shim_ec.c) and, when oracle/_ref is present, against live calls.
GPU (-m gpu): the HIP path (k_occ + k_cov through bfcg_kcov_*) against the oracle and the same goldens, bit for bit.
"""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
KCOV = json.load(open(os.path.join(HERE, "golden", "fixtures.json")))["kcov"]
IDS = lambda e: "k%d_b%d_m%d" % (e["k"], e["b"], e["min_occ"])  # noqa: E731


def _oracle_kcov(ch, min_occ, seq, off):
    """orc_kcov per read -> values on the concatenated reads (no separators)"""
    L = oracle.lib()
    out = np.zeros(len(seq), dtype=np.uint16)
    for r in range(len(off) - 1):
        a, e = int(off[r]), int(off[r + 1])
        if e > a:
            L.orc_kcov(ch, min_occ, seq[a:e].ctypes.data, e - a, out[a:e].ctypes.data)
    return out


def _summary(vals):
    return dict(md5=hashlib.md5(vals.tobytes()).hexdigest(), sum_lcov=int((vals & 0x3f).sum()), sum_hcov=int((vals >> 6 & 0x3f).sum()),
                n_solid_end=int((vals >> 12 & 1).sum()), n_high_end=int((vals >> 13 & 1).sum()))


def _unstream(vals, off):
    """values on the separator-delimited stream -> values on the concatenated reads; separators must be 0"""
    n = len(off) - 1
    sep = np.asarray(off[1:], dtype=np.int64) + np.arange(n)
    assert not vals[sep].any()
    keep = np.ones(len(vals), dtype=bool)
    keep[sep] = False
    return vals[keep]


@pytest.mark.parametrize("e", KCOV, ids=IDS)
def test_oracle_kcov_matches_reference_goldens(e, g1):
    rs, (seq, qual, off) = g1
    c = oracle.Counter(e["k"], e["b"])
    c.count(seq, qual, off)
    vals = _oracle_kcov(oracle.lib().orc_state_ch(c.st), e["min_occ"], seq, off)
    assert [int(v) for v in vals[:int(off[1])][:48]] == e["read0_head"]
    got = _summary(vals)
    assert got == {k_: e[k_] for k_ in got}
    c.close()


@pytest.mark.skipif(not oracle.have_ref_ec(), reason="oracle/_ref/libbfcref_ec.so not built (needs /root/reference)")
@pytest.mark.parametrize("k,min_occ", [(22, 1), (31, 3), (33, 2), (47, 3), (63, 1)])
def test_oracle_kcov_vs_live_reference(k, min_occ, tmp_path):
    """Ragged reads with Ns and lower case (lengths 1..260, also shorter than k): table from the oracle's dump, restored by
    the reference; its bfc_ec_kcov vs orc_kcov read by read."""
    rng = np.random.default_rng(k)
    n = 300
    lens = rng.integers(1, 260, n)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    genome = rng.integers(0, 4, 2500)
    seq = np.empty(int(off[-1]), dtype=np.uint8)
    for r in range(n):
        p = rng.integers(0, 2500 - 260)
        seq[int(off[r]):int(off[r + 1])] = np.frombuffer(b"ACGT", dtype=np.uint8)[genome[p:p + lens[r]]]
    seq[rng.integers(0, len(seq), 50)] = ord("N")
    seq[rng.integers(0, len(seq), 50)] |= 0x20
    qual = rng.integers(33, 74, len(seq)).astype(np.uint8)
    c = oracle.Counter(k, 22)
    c.count(seq, qual, off)
    fn = str(tmp_path / "t.hash")
    c.dump(fn)
    R = oracle.ref_ec()
    ch = R.bfc_ch_restore(fn.encode())
    vals = _oracle_kcov(oracle.lib().orc_state_ch(c.st), min_occ, seq, off)
    assert (vals >> 12 & 1).sum() > 100
    for r in range(n):
        a, e = int(off[r]), int(off[r + 1])
        buf = np.zeros(e - a, dtype=np.uint16)
        R.ref_kcov(ch, k, min_occ, 20, seq[a:e].tobytes(), None, buf.ctypes.data)
        assert np.array_equal(buf, vals[a:e]), "read %d" % r
    R.bfc_ch_destroy(ch)
    c.close()


# ------------------------------------------------------------------------------------------------ GPU

@pytest.mark.gpu
@pytest.mark.parametrize("e", KCOV, ids=IDS)
def test_gpu_kcov_matches_reference_goldens(gpu_lib, g1, e):
    """count on the GPU, keep the table in HBM (bfcg_kcov_attach), coverage of every read of g1: the reference's digest"""
    rs, (seq, qual, off) = g1
    s, q = gpu_lib.to_stream(seq, off), gpu_lib.to_stream(qual, off)
    g = gpu_lib.GpuCounter(e["k"], e["b"], max_batch_pos=len(s) + 64)
    g.count_host(s, q)
    kc = gpu_lib.GpuKcov(g, max_pos=len(s))
    vals = _unstream(kc.kcov(s, e["min_occ"]), off)
    assert [int(v) for v in vals[:int(off[1])][:48]] == e["read0_head"]
    got = _summary(vals)
    assert got == {k_: e[k_] for k_ in got}
    kc.close()
    # the same through a host table uploaded again (bfcg_kcov_create), in three batches cut at read boundaries
    t = g.export_table()
    g.close()
    kc = gpu_lib.GpuKcov(t, max_pos=len(s))
    cuts = [0, rs.n_reads // 3, rs.n_reads // 2, rs.n_reads]
    parts = [kc.kcov(s[cuts[i] * (rs.L + 1):cuts[i + 1] * (rs.L + 1)], e["min_occ"]) for i in range(3)]
    assert _summary(_unstream(np.concatenate(parts), off))["md5"] == e["md5"]
    kc.close()
    t.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k,min_occ", [(22, 1), (32, 3), (33, 2), (47, 3), (63, 1)])
def test_gpu_kcov_ragged_vs_oracle(gpu_lib, k, min_occ):
    """ragged reads (1..260 bases, some shorter than k), Ns, lower case, tile-straddling reads: vs orc_kcov on the same table"""
    rng = np.random.default_rng(100 + k)
    n = 2000
    lens = rng.integers(1, 260, n)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    genome = rng.integers(0, 4, 20000)
    seq = np.empty(int(off[-1]), dtype=np.uint8)
    for r in range(n):
        p = rng.integers(0, 20000 - 260)
        seq[int(off[r]):int(off[r + 1])] = np.frombuffer(b"ACGT", dtype=np.uint8)[genome[p:p + lens[r]]]
    seq[rng.integers(0, len(seq), 300)] = ord("N")
    seq[rng.integers(0, len(seq), 300)] |= 0x20
    qual = rng.integers(33, 74, len(seq)).astype(np.uint8)
    c = oracle.Counter(k, 24)
    c.count(seq, qual, off)
    want = _oracle_kcov(oracle.lib().orc_state_ch(c.st), min_occ, seq, off)
    s, q = gpu_lib.to_stream(seq, off), gpu_lib.to_stream(qual, off)
    g = gpu_lib.GpuCounter(k, 24, max_batch_pos=len(s) + 64)
    g.count_host(s, q)
    kc = gpu_lib.GpuKcov(g, max_pos=len(s))
    got = _unstream(kc.kcov(s, min_occ), off)
    assert (want >> 12 & 1).sum() > 1000
    assert np.array_equal(got, want)
    assert kc.kcov(np.zeros(0, dtype=np.uint8), min_occ).size == 0  # empty batch
    kc.close(); g.close(); c.close()
