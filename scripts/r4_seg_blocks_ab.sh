# table segments beyond 2^14 slots: the c3 genome counted with a filter 16 x too small for it (-b33 for a 1 Gbp genome: 65 536 regions, ~16 000 keys each) --
# blocks of region-owned segments (round 4) against the host's layout with random CAS (BFCG_SEG_TOTAL=14: rounds 2-3)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/exp12
export TMPDIR=/tmp
O=gpurun_out/exp12
for t in 24 14; do
  echo "== BFCG_SEG_TOTAL=$t"
  BFCG_SEG_TOTAL=$t timeout 900 python scripts/c3_run.py --b 33 --G 1000000000 --cov 8 --batch-reads 8388608 --digest 1 > $O/b31_total$t.log 2>&1
  grep "^{" $O/b31_total$t.log | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print({k:d[k] for k in d if k in ('batch_reads','gpu_ms','ms','stage_ms','n_seen','n_keys','bloom_popcount','table','G_kmers_per_s','gpu_stage_ms','wall_s','seg','table')} , list(d.keys())[:30])"
done
