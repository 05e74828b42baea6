"""Stage times of consecutive c3 steps in one process (is a kernel's time stable from step to step?).   python scripts/s1_steps.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bfc_amd
from bfc_amd import gen
rs = gen.ReadSet(seed=3, G=248_000_000, cov=float(os.environ.get("COV", 30.0)))
BR = 8_388_608
g = bfc_amd.GpuCounter(33, 35, max_batch_pos=BR * 151, tab_cshift=int(os.environ.get("TABC", 0)))
stride = 151
d_s, d_q = g.dev_alloc(rs.n_reads * stride), g.dev_alloc(rs.n_reads * stride)
for r0 in range(0, rs.n_reads, 2_000_000):
    r1 = min(rs.n_reads, r0 + 2_000_000)
    seq, qual, _ = rs.reads(r0, r1)
    g.h2d(d_s + r0 * stride, gen.to_stream(seq, rs.L, 10)); g.h2d(d_q + r0 * stride, gen.to_stream(qual, rs.L, 33))
for step in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    g.reset()
    g.stage_ms(reset=True)
    import time; t0 = time.perf_counter()
    for r0 in range(0, rs.n_reads, BR):
        r1 = min(rs.n_reads, r0 + BR)
        g.count_dev(d_s + r0 * stride, d_q + r0 * stride, (r1 - r0) * stride)
    g.sync(); wall = (time.perf_counter() - t0) * 1e3
    ms, n = g.stage_ms()
    print("step %2d: wall %.1f ms" % (step, wall), {k: round(v, 1) for k, v in ms.items()}, n, flush=True)
print("table:", g.table_info(), "partition:", g.partition_info())
