"""Throughput of the k-mer coverage pass (bfc_ec_kcov for whole batches) on the c2 read set, table resident in HBM
(not the headline bench; numbers quoted in DESIGN.md)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bfc_amd
from bfc_amd import gen
k, b = int(sys.argv[1]) if len(sys.argv) > 1 else 31, int(sys.argv[2]) if len(sys.argv) > 2 else 33
rs = gen.ReadSet(seed=2, G=4_600_000, cov=100)
seq, qual, off = rs.reads()
s_seq, s_qual = bfc_amd.to_stream(seq, off), bfc_amd.to_stream(qual, off)
stride = rs.L + 1
br = 786432
g = bfc_amd.GpuCounter(k, b, max_batch_pos=br * stride)
for r0 in range(0, rs.n_reads, br):
    r1 = min(rs.n_reads, r0 + br)
    g.count_host(s_seq[r0 * stride:r1 * stride], s_qual[r0 * stride:r1 * stride])
st = g.stats()
print("counted: k-mers %d seen %d distinct %d" % (st["n_kmers"], st["n_seen"], st["n_keys"]))
kc = bfc_amd.GpuKcov(g, max_pos=br * stride)
for rep in range(2):
    tot_ms = 0.0; solid = 0
    for r0 in range(0, rs.n_reads, br):
        r1 = min(rs.n_reads, r0 + br)
        v = kc.kcov(s_seq[r0 * stride:r1 * stride], 3)
        tot_ms += kc.last_ms(); solid += int((v >> 12 & 1).sum())
nk = st["n_kmers"]
print("kcov pass: %d solid k-mer ends of %d; GPU %.3f ms for %d table probes = %.1f G k-mers/s (%.0f GB/s at 8 B per probe + 3 B per base)" %
      (solid, nk, tot_ms, nk, nk / tot_ms / 1e6, (nk * 8 + len(s_seq) * 3) / tot_ms / 1e6))
