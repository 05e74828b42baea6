"""Hand-over of bf_high between the two passes of `bfc -1` at c5's filter size (-b37, 16 GiB): export to the host (with / without
the host-side zeroing the plain bfc_bf_init would do), and the trim context's set-up when the filter is adopted from HBM vs uploaded."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bfc_amd
from bfc_amd import gen

b = int(sys.argv[1]) if len(sys.argv) > 1 else 37
rs = gen.ReadSet(seed=2, G=1_000_000, cov=20)
seq, qual, off = rs.reads()
s_seq, s_qual = bfc_amd.to_stream(seq, off), bfc_amd.to_stream(qual, off)
g = bfc_amd.GpuCounter(51, b, filter_mode=1, max_batch_pos=len(s_seq) + 64)
g.count_host(s_seq, s_qual); g.sync()
for name, kw in (("plain export", {}), ("resident export", dict(resident=True))):
    t0 = time.time(); bf = g.export_bloom(1, **kw); t1 = time.time()
    tr = bfc_amd.GpuTrimmer(51, bf, max_pos=1 << 24, max_reads=1 << 18); t2 = time.time()
    print("%-16s: export %.2f s, trim context set-up %.2f s (adopted from HBM: %s)" % (name, t1 - t0, t2 - t1, tr.adopted), flush=True)
    tr.close(); bf.close()
g.close()
