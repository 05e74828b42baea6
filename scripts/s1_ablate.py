"""k_scatter1 alone under the debug switches of BFCG_ABLATE (256 no stores, 512 no copy-out, 1024 no cursor atomics, 2048 no hashing): ms for
one batch of c3 reads.    BFCG_ABLATE=2048 python scripts/s1_ablate.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the measurement switches exist only in the -DBFCG_MEASURE build (python -m bfc_amd.build --measure, built before the GPU call)
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
    try:
        g.count_dev(d_s, d_q, len(s)); g.sync()
    except Exception as e:  # noqa: BLE001  (ablated runs leave garbage behind: only stage A's time is of interest)
        print("(", str(e)[:80], ")")
    print("b", B, "ablate", os.environ.get("BFCG_ABLATE", "0"), "scatter1 ms:", round(g.last_batch_ms()["scatter1"], 3), flush=True)
    try:
        g.reset()
    except Exception:  # noqa: BLE001
        break
