"""c2 through the host-buffer entry point (bfcg_count_batch_host): the PCIe-inclusive rate, from pageable numpy arrays and from pinned buffers
(bfcg_host_alloc, what bfc_count uses).  Not the benchmark's `value` (inputs resident in HBM)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bfc_amd
from bfc_amd import gen, _lib
rs = gen.ReadSet(seed=2, G=4_600_000, cov=100)
seq, qual, off = rs.reads()
s_seq, s_qual = bfc_amd.to_stream(seq, off), bfc_amd.to_stream(qual, off)
stride, br = rs.L + 1, 786432
g = bfc_amd.GpuCounter(31, 33, max_batch_pos=br * stride)
L = _lib.load()
L.bfcg_host_alloc.restype = C.c_void_p; L.bfcg_host_alloc.argtypes = [C.c_uint64]
def pinned(a):
    p = L.bfcg_host_alloc(len(a)); v = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(len(a),)); v[:] = a; return v
p_seq, p_qual = pinned(s_seq), pinned(s_qual)
for name, a, b in (("pageable", s_seq, s_qual), ("pinned", p_seq, p_qual)):
    best = 1e9
    for rep in range(4):
        g.reset(); g.sync()
        t0 = time.perf_counter()
        for r0 in range(0, rs.n_reads, br):
            r1 = min(rs.n_reads, r0 + br)
            g.count_host(a[r0 * stride:r1 * stride], b[r0 * stride:r1 * stride])
        g.sync()
        best = min(best, time.perf_counter() - t0)
    st = g.stats()
    print("%s host buffers: %.1f ms per pass = %.2f G k-mers/s (%.1f GB/s of input over PCIe), %d k-mers, %d distinct" % (name, best * 1e3, st["n_kmers"] / best / 1e9, 2 * len(a) / best / 1e9, st["n_kmers"], st["n_keys"]))
