"""k_bloom phase cycles (BFCG_ABLATE=64: clock64 of thread 0 of every workgroup, summed) for the first, cold batches of config c3 and for warm ones.
    BFCG_ABLATE=64 python scripts/bloom_phases.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BFCG_ABLATE", "64")
# the measurement switches exist only in the -DBFCG_MEASURE build (python -m bfc_amd.build --measure, built before the GPU call)
os.environ.setdefault("BFC_GPU_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "libbfc_gpu_measure.so"))
import numpy as np
import bfc_amd
from bfc_amd import gen
rs = gen.ReadSet(seed=3, G=248_000_000, cov=30.0)
BR = int(os.environ.get("BR", 3670016))
g = bfc_amd.GpuCounter(33, 35, max_batch_pos=BR * 151)
prev = np.zeros(6)
names = ["stage", "pass1", "listA+B+C", "writeback", "handover", "passA"]
for t in range(int(os.environ.get("NB", 6))):
    seq, qual, _ = rs.reads(t * BR, (t + 1) * BR)
    g.count_host(gen.to_stream(seq, rs.L, 10), gen.to_stream(qual, rs.L, 33))
    st = g.stats()
    cur = np.array(st["phase_cycles"], dtype=np.float64)
    d = cur - prev; prev = cur
    ms = g.last_batch_ms()
    print("batch %d: bloom %.2f ms; slow regions so far %d; cycles per region (thread 0): %s" % (t, ms["bloom"], st["slow_buckets"], {n: int(v / 262144) for n, v in zip(names, d)}), flush=True)
g.close()
