"""Throughput of the bloom-query (trim) pass on a c2-sized read set (not the headline bench; numbers quoted in DESIGN.md)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bfc_amd
from bfc_amd import gen
k, b = int(sys.argv[1]) if len(sys.argv) > 1 else 51, int(sys.argv[2]) if len(sys.argv) > 2 else 33
rs = gen.ReadSet(seed=2, G=4_600_000, cov=100)
seq, qual, off = rs.reads()
s_seq, s_qual = bfc_amd.to_stream(seq, off), bfc_amd.to_stream(qual, off)
stride = rs.L + 1
br = 655360  # two bloom slices per region in LDS: ~1100 k-mers per region and batch
g = bfc_amd.GpuCounter(k, b, filter_mode=1, max_batch_pos=br * stride)
t0 = time.time()
for r0 in range(0, rs.n_reads, br):
    r1 = min(rs.n_reads, r0 + br)
    g.count_host(s_seq[r0 * stride:r1 * stride], s_qual[r0 * stride:r1 * stride])
st = g.stats()
print("count (filter mode, host batches): %.2fs, k-mers %d seen %d" % (time.time() - t0, st["n_kmers"], st["n_seen"]))
bf = g.export_bloom(1)
g.close()
tr = bfc_amd.GpuTrimmer(k, bf, max_pos=br * stride, max_reads=br)
tot_ms = 0.0; kept = 0
for rep in range(2):
    tot_ms = 0.0; kept = 0
    for r0 in range(0, rs.n_reads, br):
        r1 = min(rs.n_reads, r0 + br)
        so = np.arange(r1 - r0 + 1, dtype=np.uint64) * np.uint64(stride)
        st_, en_ = tr.trim(s_seq[r0 * stride:r1 * stride], so, 0.9)
        tot_ms += tr.last_ms(); kept += int((st_ >= 0).sum())
nk = st["n_kmers"]
print("trim pass: %d reads kept of %d; GPU %.3f ms for %d queries = %.1f G queries/s = %.0f GB/s at 64 B per query (%.1f %% of 8 TB/s)" %
      (kept, rs.n_reads, tot_ms, nk, nk / tot_ms / 1e6, nk * 64 / tot_ms / 1e6, nk * 64 / tot_ms / 1e6 / 80))
