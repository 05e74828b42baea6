"""per-call stage times of the first 8.4 M c3 reads handed over in calls of a given size (see tests/test_gpu_baseline_shapes.py::test_c3_shape_has_no_batch_size_cliff)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bfc_amd
from bfc_amd import gen
rs = gen.ReadSet(seed=3, G=248_000_000, cov=30.0)
n = 8_388_608
seq, qual, _ = rs.reads(0, n)
s, q = gen.to_stream(seq, rs.L, 10), gen.to_stream(qual, rs.L, 33)
stride = rs.L + 1
for call in [int(v) for v in sys.argv[1:]] or [2_097_152, 4_194_304]:
    g = bfc_amd.GpuCounter(33, 35, max_batch_pos=call * stride + 64)
    prev = None
    for a in range(0, n, call):
        g.count_host(s[a * stride:(a + call) * stride], q[a * stride:(a + call) * stride])
        ms, nl = g.stage_ms()
        d = {k: round(ms[k] - (prev[0][k] if prev else 0), 2) for k in ms}
        print(call, "call at read", a, "batches", nl - (prev[1] if prev else 0), d, g.partition_info(), flush=True)
        prev = (ms, nl)
    st = g.stats()
    print(call, "slow", st["slow_buckets"], "table", g.table_info(), flush=True)
    g.close()
