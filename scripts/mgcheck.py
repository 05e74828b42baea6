import os
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bfc_amd
from bfc_amd import gen
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import mg_protocol as bdist
rs = gen.fixture("g42"); seq, qual, off = rs.reads()
n = rs.n_reads
for N in (1, 2):
    per = (n // 2 + 15) // 16 * 16
    cl = bdist.LocalCluster(bfc_amd, N, 31, 30, max_batch_pos=(per + 64) * 151)
    for lo in range(0, n, per):
        hi = min(n, lo + per)
        pr = (hi - lo + N - 1) // N
        row = []
        for r in range(N):
            a, b = min(hi, lo + r * pr), min(hi, lo + (r + 1) * pr)
            o = off[a:b + 1] - off[a]
            row.append((bfc_amd.to_stream(seq[int(off[a]):int(off[b])], o), bfc_amd.to_stream(qual[int(off[a]):int(off[b])], o)))
        cl.batch(row)
    st = cl.stats()
    print("N=%d" % N, st["n_kmers"], st["n_seen"], st["n_keys"], "expected 23950926 16708108 1156389", "slow", st["slow_buckets"])
    cl.close()
