"""Per-region load in the owner-computes multi-GPU mode, emulated on ONE device (LocalCluster): N ranks each bring a c2-sized batch.
With the filter size fixed (-b33) every region receives N times a single-GPU batch's k-mers and overflows its LDS capacity (exact slow
path); with 1 GiB of filter PER RANK (-b 33+log2 N, what bench.py uses for N > 1) the per-region load is the single-GPU one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bfc_amd
from bfc_amd import gen
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import mg_protocol as bdist

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
br = 786432
for b in (33, 33 + int(np.log2(N))):
    cl = bdist.LocalCluster(bfc_amd, N, 31, b, br * 151)
    sets = [gen.ReadSet(seed=2 + r, G=4_600_000, cov=100 * (2 * br + 8) / 3066666.0) for r in range(N)]
    data = []
    for rs in sets:
        seq, qual, off = rs.reads()
        data.append((bfc_amd.to_stream(seq, off), bfc_amd.to_stream(qual, off)))
    for t in range(2):
        cl.batch([(s[t * br * 151:(t + 1) * br * 151], q[t * br * 151:(t + 1) * br * 151]) for s, q in data])
    for c in cl.ctx:
        c.sync()
    ms = [c.stage_ms()[0] for c in cl.ctx]
    st = cl.stats()
    print("N=%d -b%d: k_bloom %.1f ms per rank for 2 global batches, slow regions %d of %d, %d k-mers" %
          (N, b, np.mean([m["bloom"] for m in ms]), st["slow_buckets"], 2 * (1 << (b - 17)), st["n_kmers"]), flush=True)
    cl.close()
