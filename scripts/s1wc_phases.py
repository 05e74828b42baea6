"""k_scatter1_wc's phase clocks (the -DBFCG_MEASURE library): one batch of c3 reads, cycles per round of thread 0 and of the first owner lane.
    python scripts/s1wc_phases.py        (BR = reads in the batch, B = filter bits)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BFC_GPU_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "libbfc_gpu_measure.so"))
import bfc_amd
from bfc_amd import gen
rs = gen.ReadSet(seed=3, G=248_000_000, cov=30.0)
BR = int(os.environ.get("BR", 3670016))
B = int(os.environ.get("B", 35))
g = bfc_amd.GpuCounter(33, B, max_batch_pos=BR * 151)
seq, qual, _ = rs.reads(0, BR)
s, q = gen.to_stream(seq, rs.L, 10), gen.to_stream(qual, rs.L, 33)
d_s, d_q = g.dev_alloc(len(s)), g.dev_alloc(len(q))
g.h2d(d_s, s); g.h2d(d_q, q)
for rep in range(3):
    g.reset()
    try:
        g.count_dev(d_s, d_q, len(s)); g.sync()
    except Exception as e:  # noqa: BLE001  (ablated runs leave garbage behind: only stage A's clocks are of interest)
        print("(", str(e)[:80], ")")
    pc = g.stats()["phase_cycles"]
    n = max(pc[5], 1)
    print("tag", os.environ.get("TAG", ""), "scatter1 ms: %.3f" % g.last_batch_ms()["scatter1"], "rounds (all workgroups)", pc[5],
          "cycles per round: P1 %.0f  P2 %.0f | in P2: loaders' work %.0f, owners' work %.0f | wave 8's P1 work %.0f" % (pc[0] / n, pc[1] / n, pc[2] / n, pc[3] / n, pc[4] / n), flush=True)
