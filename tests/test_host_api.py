"""CPU tests (-m "not gpu") of libbfc_gpu.so's host side: the library loads and exports every symbol include/bfc_gpu.h
declares; the reference-shaped single-element / query / dump / restore entry points (bbf.h, htab.h) behave like the
reference; and counting without a GPU fails loudly instead of falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported(gpu_lib):
    from bfc_amd import _lib
    L = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "bfc_gpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(bfcg?_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 38
    for n in sorted(names):
        assert hasattr(L, n), "symbol %s declared in include/bfc_gpu.h but not exported" % n
    assert names == set(_lib.SYMBOLS), names ^ set(_lib.SYMBOLS)


def test_struct_layouts_match_reference_abi(gpu_lib):
    from bfc_amd._lib import BfcOpt, BfcBf
    assert C.sizeof(BfcOpt) == 22 * 4 and BfcOpt.min_frac.offset == 32 and BfcOpt.l_pre.offset == 36  # bfc.h:15-33
    assert (BfcBf.n_shift.offset, BfcBf.n_hashes.offset, BfcBf.b.offset) == (0, 4, 8)              # bbf.h:9-12, read by correct.c:490


def test_no_gpu_means_loud_failure(gpu_lib):
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import bfc_amd\n"
            "try:\n    bfc_amd.GpuCounter(31, 26)\nexcept bfc_amd.BfcGpuError as e:\n    print('LOUD', e)\n" % ROOT)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert "LOUD" in r.stdout and "no CPU fallback" in (r.stdout + r.stderr)


def test_no_gpu_means_loud_failure_for_groups_too(gpu_lib):
    """the multi-GPU entry (bfcg_group_create) has no CPU path either; the RCCL unique id needs no device"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import bfc_amd\n"
            "try:\n    bfc_amd.GpuGroup(31, 26, [0, 0], max_batch_pos=1 << 20)\nexcept bfc_amd.BfcGpuError as e:\n    print('LOUD', e)\n" % ROOT)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert "LOUD" in r.stdout and "no CPU fallback" in (r.stdout + r.stderr)


def test_bench_gpus_n_without_n_devices_fails_loudly(gpu_lib):
    """`python bench.py --gpus 2` with no launcher runs both ranks in one process (bfcg_group_create, n_local = 2); on a box with fewer devices it
    must say so in its ONE JSON line and exit non-zero -- never a silent 1-GPU run labelled n_gpus 2."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    env.pop("WORLD_SIZE", None)
    for n, word in ((2, "not running on fewer GPUs"), (3, "power of two")):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
        assert r.returncode == 2, (r.returncode, r.stderr[-500:])
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(lines) == 1
        d = json.loads(lines[0])
        assert d["n_gpus"] == n and d["value"] is None and word in d["error"]


def test_host_bloom_matches_oracle(gpu_lib):
    """bfc_bf_init/insert/get (bbf.c): same bits, same return values; bad shifts give NULL (bbf.c:9)."""
    L = oracle.lib()
    rng = np.random.default_rng(5)
    hb = gpu_lib.HostBloom.init(20, 4)
    ob = L.orc_bf_new(20, 4)
    for h in rng.integers(0, 2 ** 63, 5000, dtype=np.int64).astype(np.uint64):
        h = int(h)
        assert hb.get(h) == L.orc_bf_get(ob, h)
        assert hb.insert(h) == L.orc_bf_insert(ob, h)
        assert hb.get(h) == 4
    ref = np.ctypeslib.as_array(C.cast(L.orc_bf_bits(ob), C.POINTER(C.c_uint8)), shape=(1 << 17,))
    assert np.array_equal(hb.bytes(), ref)
    assert gpu_lib.HostBloom.init(8, 4) is None and gpu_lib.HostBloom.init(56, 4) is None
    hb.close(); L.orc_bf_free(ob)


@pytest.mark.parametrize("k,l_pre", [(31, 20), (33, 20), (51, 20), (63, 20), (21, 10)])
def test_host_table_matches_oracle(gpu_lib, k, l_pre, tmp_path):
    """bfc_ch_insert/get/count/hist/dump/restore (htab.c) on the host table vs the oracle: saturation at 255/63,
    growth, l_pre clamping, and a dump the oracle's parser and our restore both read back (L1)."""
    L = oracle.lib()
    rng = np.random.default_rng(k)
    m = (1 << k) - 1
    keys = [(int(a) & m, int(b) & m) for a, b in zip(rng.integers(0, 2 ** 63, 3000, dtype=np.int64), rng.integers(0, 2 ** 63, 3000, dtype=np.int64))]
    t = gpu_lib.HostTable.init(k, l_pre)
    oc = L.orc_ch_new(k, l_pre)
    assert t.k == k and t.l_pre == L.orc_ch_lpre(oc)
    for rep in range(3):
        for i, (a, b) in enumerate(keys):
            hi = (i + rep) & 1
            y = (C.c_uint64 * 2)(a, b)
            assert t.insert(a, b, hi) == 0
            L.orc_ch_insert(oc, y, hi)
    for _ in range(300):  # saturation of one key
        t.insert(*keys[0], 1)
        L.orc_ch_insert(oc, (C.c_uint64 * 2)(*keys[0]), 1)
    assert t.get(*keys[0]) == L.orc_ch_get(oc, (C.c_uint64 * 2)(*keys[0])) == (63 << 8 | 255)
    for a, b in keys[:500] + [(1, 2), (m, m)]:
        assert t.get(a, b) == L.orc_ch_get(oc, (C.c_uint64 * 2)(a, b))
    assert t.count() == L.orc_ch_count(oc)
    cnt = np.zeros(256, dtype=np.uint64); high = np.zeros(64, dtype=np.uint64)
    omode = L.orc_ch_hist(oc, cnt.ctypes.data_as(C.POINTER(C.c_uint64)), high.ctypes.data_as(C.POINTER(C.c_uint64)))
    mode, c2, h2 = t.hist()
    assert mode == omode and np.array_equal(cnt, c2) and np.array_equal(high, h2)
    fn = str(tmp_path / "t.hash")
    assert t.dump(fn) == 0
    kk, lp, sizes, slots = oracle.parse_dump(fn)
    osz = np.zeros(1 << lp, dtype=np.uint32)
    n = L.orc_ch_export(oc, osz.ctypes.data_as(C.POINTER(C.c_uint32)), None)
    osl = np.zeros(n, dtype=np.uint64)
    L.orc_ch_export(oc, osz.ctypes.data_as(C.POINTER(C.c_uint32)), osl.ctypes.data_as(C.POINTER(C.c_uint64)))
    assert (kk, lp) == (k, t.l_pre) and np.array_equal(sizes, osz) and np.array_equal(slots, osl)
    t2 = gpu_lib.HostTable.restore(fn)
    s2 = t2.export_sorted()
    assert t2.count() == t.count() and np.array_equal(s2[0], osz) and np.array_equal(s2[1], osl)
    assert gpu_lib.HostTable.restore(str(tmp_path / "missing")) is None  # htab.c:157
    assert t.dump(str(tmp_path / "no_such_dir" / "x")) == -1                # htab.c:134
    t.close(); t2.close(); L.orc_ch_free(oc)


def test_kmer_occ_uses_the_strand_canonical_hash(gpu_lib):
    """bfc_ch_kmer_occ (htab.c:94-99): a k-mer and its reverse complement hit the same slot."""
    L = oracle.lib()
    k = 31
    s = "ACGTTGCATGCCGATTACAGGCTAGCTTAGG"
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rc = "".join(comp[c] for c in reversed(s))
    t = gpu_lib.HostTable.init(k, 20)
    xs = []
    for st in (s, rc):
        x = (C.c_uint64 * 4)(0, 0, 0, 0)
        for ch in st:
            L.orc_kmer_push(k, x, "ACGT".index(ch))
        xs.append([int(v) for v in x])
    y = (C.c_uint64 * 2)()
    L.orc_kmer_hash(k, (C.c_uint64 * 4)(*xs[0]), y)
    assert t.kmer_occ(xs[0]) == -1
    t.insert(int(y[0]), int(y[1]), 1)
    assert t.kmer_occ(xs[0]) == t.kmer_occ(xs[1]) == (1 << 8 | 1)
    t.close()


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref/libbfcref.so not built (needs /root/reference)")
@pytest.mark.parametrize("seed", range(20))
def test_host_table_interoperates_with_the_reference(gpu_lib, seed, tmp_path):
    """Random k / l_pre / key sets: a table built by the REFERENCE's own bfc_ch_insert and dumped by its bfc_ch_dump is restored by
    this library and answers every bfc_ch_get like the reference; the library's dump of it is restored by the reference and answers
    alike again (htab.c:60-92, 129-176 in both directions)."""
    R = oracle.ref()
    R.bfc_ch_init.restype = C.c_void_p; R.bfc_ch_init.argtypes = [C.c_int, C.c_int]
    R.bfc_ch_insert.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int, C.c_int]
    R.bfc_ch_dump.argtypes = [C.c_void_p, C.c_char_p]
    R.bfc_ch_restore.restype = C.c_void_p; R.bfc_ch_restore.argtypes = [C.c_char_p]
    R.bfc_ch_destroy.argtypes = [C.c_void_p]
    rng = np.random.default_rng(seed)
    k = int(rng.integers(37, 64)) if seed == 0 else int(rng.integers(11, 37))  # k >= 37 clamps l_pre to 24 (htab.c:24-26): 16 M khash tables, once
    l_pre = int(rng.choice([v for v in (4, 8, 12, 16, 16, 20) if v <= 2 * k - 2]))
    m = (1 << k) - 1
    n = int(rng.integers(1, 4000))
    ys = [(int(a) & m, int(b) & m) for a, b in zip(rng.integers(0, 2 ** 63, n, dtype=np.int64), rng.integers(0, 2 ** 63, n, dtype=np.int64))]
    rt = R.bfc_ch_init(k, l_pre)
    for i, (a, b) in enumerate(ys):
        for _ in range(int(rng.integers(1, 4)) if i % 50 else 300):  # a few keys saturate
            R.bfc_ch_insert(rt, (C.c_uint64 * 2)(a, b), int(rng.integers(0, 2)), 0)
    f1, f2 = str(tmp_path / "ref.hash").encode(), str(tmp_path / "mine.hash").encode()
    assert R.bfc_ch_dump(rt, f1) == 0
    t = gpu_lib.HostTable.restore(f1.decode())
    probes = ys + [(int(a) & m, int(b) & m) for a, b in zip(rng.integers(0, 2 ** 63, 500, dtype=np.int64), rng.integers(0, 2 ** 63, 500, dtype=np.int64))]
    for a, b in probes:
        assert t.get(a, b) == R.bfc_ch_get(rt, (C.c_uint64 * 2)(a, b)), (k, l_pre, a, b)
    assert t.count() == R.bfc_ch_count(rt)
    assert t.dump(f2.decode()) == 0
    rt2 = R.bfc_ch_restore(f2)
    for a, b in probes:
        assert R.bfc_ch_get(rt2, (C.c_uint64 * 2)(a, b)) == R.bfc_ch_get(rt, (C.c_uint64 * 2)(a, b))
    assert R.bfc_ch_count(rt2) == R.bfc_ch_count(rt)
    R.bfc_ch_destroy(rt); R.bfc_ch_destroy(rt2); t.close()


def test_union_of_disjoint_tables(gpu_lib):
    """bfc_ch_union: the per-GPU tables of an owner-computes run hold disjoint key sets; their union answers bfc_ch_get / count / hist like
    one table that received all inserts; a key in two inputs gets saturating sums."""
    from bfc_amd import _lib
    L = oracle.lib()
    rng = np.random.default_rng(3)
    k, l_pre = 33, 20
    m = (1 << k) - 1
    keys = [(int(a) & m, int(b) & m) for a, b in zip(rng.integers(0, 2 ** 63, 6000, dtype=np.int64), rng.integers(0, 2 ** 63, 6000, dtype=np.int64))]
    parts = [gpu_lib.HostTable.init(k, l_pre) for _ in range(4)]
    oc = L.orc_ch_new(k, l_pre)
    for i, (a, b) in enumerate(keys):
        for rep in range(1 + i % 3):
            parts[i % 4].insert(a, b, (i + rep) & 1)
            L.orc_ch_insert(oc, (C.c_uint64 * 2)(a, b), (i + rep) & 1)
    for _ in range(200):  # one key on two "ranks", close to saturation on both
        parts[0].insert(*keys[0], 1); parts[1].insert(*keys[0], 1)
    for _ in range(400):
        L.orc_ch_insert(oc, (C.c_uint64 * 2)(*keys[0]), 1)
    arr = (C.c_void_p * 4)(*[p.ptr for p in parts])
    u = gpu_lib.HostTable(_lib.load().bfc_ch_union(arr, 4))
    assert u.count() == L.orc_ch_count(oc)
    for a, b in keys[:1500] + [(5, 6)]:
        assert u.get(a, b) == L.orc_ch_get(oc, (C.c_uint64 * 2)(a, b))
    cnt = np.zeros(256, dtype=np.uint64); high = np.zeros(64, dtype=np.uint64)
    omode = L.orc_ch_hist(oc, cnt.ctypes.data_as(C.POINTER(C.c_uint64)), high.ctypes.data_as(C.POINTER(C.c_uint64)))
    mode, c2, h2 = u.hist()
    assert mode == omode and np.array_equal(cnt, c2) and np.array_equal(high, h2)
    other = gpu_lib.HostTable.init(31, 20)
    arr2 = (C.c_void_p * 2)(parts[0].ptr, other.ptr)
    assert not _lib.load().bfc_ch_union(arr2, 2)  # different k
    for p in parts + [u, other]:
        p.close()
    L.orc_ch_free(oc)


def test_concurrent_inserts_grow_once_per_need(gpu_lib):
    """bfc_ch_insert from many threads (htab.c:60-82 is thread-safe through per-sub-table locks): every thread that finds a
    sub-table full asks for growth, but the table must double only if nobody has grown it meanwhile -- N contenders used to
    double it N times (2^33 slots for a few million keys).  2^10 sub-tables, ~230 keys each: eight doublings under contention."""
    import threading
    k, l_pre, T, N = 21, 10, 8, 30000
    t = gpu_lib.HostTable.init(k, l_pre)
    L = t.L
    L.bfc_ch_raw_cshift.restype = C.c_int
    L.bfc_ch_raw_cshift.argtypes = [C.c_void_p]
    m = (1 << k) - 1
    rng = np.random.default_rng(11)
    y0 = rng.integers(0, m, size=(T, N), dtype=np.int64)
    y1 = rng.integers(0, m, size=(T, N), dtype=np.int64)

    def work(i):
        for a, b in zip(y0[i], y1[i]):
            L.bfc_ch_insert(t.ptr, (C.c_uint64 * 2)(int(a), int(b)), 1, 1)

    th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    keys = {(int(a), int(b)) for a, b in zip(y0.ravel(), y1.ravel())}
    assert t.count() == len(keys)
    # ~234 keys per sub-table (max ~300): 2^9 slots hold them; one spare doubling tolerated
    assert L.bfc_ch_raw_cshift(t.ptr) <= 10, L.bfc_ch_raw_cshift(t.ptr)
    for a, b in list(keys)[:200]:
        assert t.get(a, b) >= 1
    t.close()
