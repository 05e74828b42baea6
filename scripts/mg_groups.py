import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, bfc_amd
from bfc_amd import gen
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import mg_protocol as bdist
# the overloaded case of scripts/mg_load.py (N ranks, filter fixed at -b33) with stage B in source groups
N, br = int(sys.argv[1]), 786432
cl = bdist.LocalCluster(bfc_amd, N, 31, 33, br * 151)
sets = [gen.ReadSet(seed=2 + r, G=4_600_000, cov=100 * (2 * br + 8) / 3066666.0) for r in range(N)]
data = []
for rs in sets:
    seq, qual, off = rs.reads(); data.append((bfc_amd.to_stream(seq, off), bfc_amd.to_stream(qual, off)))
for t in range(2):
    cl.batch([(s[t * br * 151:(t + 1) * br * 151], q[t * br * 151:(t + 1) * br * 151]) for s, q in data])
for c in cl.ctx: c.sync()
ms = [c.stage_ms()[0] for c in cl.ctx]; st = cl.stats()
print("N=%d -b33 with source groups: k_bloom %.1f ms per rank for 2 global batches (%d stage-B launches), slow regions %d" % (N, np.mean([m["bloom"] for m in ms]), cl.launches, st["slow_buckets"]))
cl.close()
