import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, bfc_amd
from bfc_amd import gen
rs = gen.ReadSet(seed=2, G=4_600_000, cov=100)
seq, qual, off = rs.reads()
s, q = bfc_amd.to_stream(seq, off), bfc_amd.to_stream(qual, off)
for fm, k in ((1, 31), (1, 51), (0, 31)):
    for reads in (662251, 529801, 786432):
        n = reads * 151
        g = bfc_amd.GpuCounter(k, 33, filter_mode=fm, max_batch_pos=n)
        ds, dq = g.dev_alloc(len(s)), g.dev_alloc(len(q)); g.h2d(ds, s); g.h2d(dq, q)
        for rep in range(2):
            g.reset(); g.sync(); g.stage_ms(reset=True); t0 = time.perf_counter()
            for o in range(0, len(s), n):
                g.count_dev(ds + o, dq + o, min(n, len(s) - o))
            g.sync(); dt = time.perf_counter() - t0
        st = g.stats()
        print("filter_mode %d k=%d, %d reads per batch (%.1f M positions, limit %.1f M): %.2f ms, slow regions %d, bloom %.2f ms" % (fm, k, reads, n / 1e6, g.batch_limit() / 1e6, dt * 1e3, st["slow_buckets"], g.stage_ms()[0]["bloom"]), flush=True)
        g.dev_free(ds); g.dev_free(dq); g.close()
