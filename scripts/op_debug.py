import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bfc_amd, oracle
from bfc_amd import gen
rs = gen.fixture("g1"); seq, qual, off = rs.reads()
for (k, b, fm, n) in [(33, 35, 1, 3000), (33, 35, 0, 3000), (31, 33, 0, 6000), (33, 30, 1, 3000)]:
    s, q, o = seq[:n * rs.L], qual[:n * rs.L], off[:n + 1]
    oc = oracle.Counter(k, b, filter_mode=fm); oc.count(s, q, o)
    g = bfc_amd.GpuCounter(k, b, filter_mode=fm, max_batch_pos=len(s) + n + 64)
    g.count_host(bfc_amd.to_stream(s, o), bfc_amd.to_stream(q, o))
    st = g.stats(); ost = oc.stats()
    ok0 = np.array_equal(g.bloom_bytes(), oc.bloom_bytes())
    ok1 = np.array_equal(g.bloom_bytes(1), oc.bloom_bytes(True)) if fm else None
    print(k, b, fm, g.partition_info(), "stats", (st["n_kmers"], st["n_seen"]), (ost["n_kmers"], ost["n_seen"]), "bloom", ok0, ok1, flush=True)
    g.close(); oc.close()
