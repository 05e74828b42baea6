"""Quick rate check of other configurations on the c2-sized read set (k=33 takes the 64-bit arithmetic / 16-byte record path)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bfc_amd
from bfc_amd import gen
k, b = int(sys.argv[1]), int(sys.argv[2])
rs = gen.ReadSet(seed=2, G=4_600_000, cov=100)
seq, qual, off = rs.reads()
s_seq, s_qual = bfc_amd.to_stream(seq, off), bfc_amd.to_stream(qual, off)
stride = rs.L + 1
br = int(sys.argv[3]) if len(sys.argv) > 3 else 786432
g = bfc_amd.GpuCounter(k, b, max_batch_pos=br * stride)
d_seq = g.dev_alloc(len(s_seq)); d_qual = g.dev_alloc(len(s_qual)); g.h2d(d_seq, s_seq); g.h2d(d_qual, s_qual)
for rep in range(3):
    g.reset(); g.sync()
    t0 = time.perf_counter()
    for r0 in range(0, rs.n_reads, br):
        r1 = min(rs.n_reads, r0 + br)
        g.count_dev(d_seq + r0 * stride, d_qual + r0 * stride, (r1 - r0) * stride)
    g.sync()
    dt = time.perf_counter() - t0
st = g.stats()
print("k=%d b=%d batch_reads=%d: %.2f ms, %.1f G k-mers/s; kmers %d seen %d keys %d slow %d cshift %d" % (k, b, br, dt * 1e3, st["n_kmers"] / dt / 1e9, st["n_kmers"], st["n_seen"], st["n_keys"], st["slow_buckets"], st["tab_cshift"]))
