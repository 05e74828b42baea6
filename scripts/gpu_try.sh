cd $GRAFT_REPO_ROOT
BFCG_DEBUG=0 timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x -k one_pass 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6
python - <<'PY'
# how often the forced one-pass draws were replayed
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
os.environ["BFCG_ONEPASS_MIN_TILES"] = "1"
import numpy as np, bfc_amd
import test_gpu_fuzz as T
rep = 0; op = 0
for seed in range(40):
    prm, seq, qual, off, cuts, kw = T._draw(40000 + seed, scale=12, b_range=(26, 32))
    n = len(off) - 1
    g = bfc_amd.GpuCounter(prm["k"], prm["b"], q=prm["q"], n_hashes=prm["nh"], l_pre=prm["l_pre"], filter_mode=prm["fm"], max_batch_pos=len(seq) + n + 64, **kw)
    p0 = g.partition_info()
    for a, e in zip(cuts[:-1], cuts[1:]):
        o = off[a:e + 1] - off[a]
        g.count_host(bfc_amd.to_stream(seq[int(off[a]):int(off[e])], o), bfc_amd.to_stream(qual[int(off[a]):int(off[e])], o) if qual is not None else None)
    g.stats(); p1 = g.partition_info()
    op += p0["one_pass"]; rep += p1["replayed_batches"] > 0
    g.close()
print("draws with the one-pass partition:", op, "of 40; with replayed batches:", rep)
PY
bash scripts/more_fuzz.sh 7 8 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8
