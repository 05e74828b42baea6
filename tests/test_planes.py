"""Bit planes of a batch (bfcg_pack_planes, include/bfc_gpu.h) -- CPU only: what count.c:72-89 reads of a position, for EVERY byte value and
thresholds on both sides of what a signed char can reach; and the expansion the device makes of them (k_expand_planes, restated in numpy)
must give back bytes with exactly those properties."""
import numpy as np
import pytest

import bfc_amd


def _props(seq, qual, q):
    code = np.full(256, 4, dtype=np.uint8)
    for ch, c in zip(b"ACGTacgt", [0, 1, 2, 3, 0, 1, 2, 3]):  # bseq.c:9-26 minus one (count.c:82)
        code[ch] = c
    c = code[seq]
    hi = (qual.astype(np.int8).astype(np.int32) - 33 >= q) if qual is not None else np.ones(len(seq), dtype=bool)  # count.c:85: a signed char
    return c & 1, (c >> 1) & 1, c >> 2, hi.astype(np.uint8)


def _bits(plane, n):
    return np.unpackbits(plane.view(np.uint8), bitorder="little")[:n]


def _expand(planes, n, has_qual):
    """k_expand_planes (bfcg_ctx.hip) in numpy"""
    m0, m1, mn, mq = (_bits(planes[p], n).astype(np.uint32) for p in range(4))
    base = 0x41 + m0 * 2 + m1 * 6 + (m0 & m1) * 0x0b
    seq = np.where(mn == 1, 0x0a, base).astype(np.uint8)
    qual = (0x80 - mq).astype(np.uint8) if has_qual else None
    return seq, qual


@pytest.mark.parametrize("q", [-200, -162, -161, -160, -34, -33, 0, 1, 20, 41, 93, 94, 95, 127, 200])
@pytest.mark.parametrize("threads", [1, 3])
def test_planes_of_every_byte_value(q, threads):
    rng = np.random.default_rng(q + 1000)
    n = 256 * 256 + 77  # every (sequence byte, quality byte) pair, and a ragged end
    seq = np.concatenate([np.repeat(np.arange(256, dtype=np.uint8), 256), rng.integers(0, 256, 77).astype(np.uint8)])
    qual = np.concatenate([np.tile(np.arange(256, dtype=np.uint8), 256), rng.integers(0, 256, 77).astype(np.uint8)])
    pl = bfc_amd.pack_planes(seq, qual, q, n_chunks=threads)
    want = _props(seq, qual, q)
    for p in range(4):
        assert np.array_equal(_bits(pl[p], n), want[p]), "plane %d" % p
    assert _bits(pl[2], n + 19)[n:].all(), "positions beyond the batch's end in its last word are separators"
    # the bytes the device makes of the planes have the same properties under the same threshold
    eseq, equal = _expand(pl, n, True)
    got = _props(eseq, equal, q)
    for p in (2, 3):
        assert np.array_equal(got[p], want[p]), "expanded plane %d" % p
    ok = want[2] == 0
    assert np.array_equal(got[0][ok], want[0][ok]) and np.array_equal(got[1][ok], want[1][ok])


def test_planes_without_qualities():
    seq = np.frombuffer(b"ACGTNacgt\nAAAC", dtype=np.uint8)
    pl = bfc_amd.pack_planes(seq, None, 20)
    assert _bits(pl[3], len(seq)).all()
    eseq, _ = _expand(pl, len(seq), False)
    assert bytes(eseq) == b"ACGT\nACGT\nAAAC"
