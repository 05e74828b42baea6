cd $GRAFT_REPO_ROOT
for cfg in "BFCG_LDS=53000" "BFCG_LDS=80000" "BFCG_LDS=80000 BFCG_SYNC_BATCHES=1" "BFCG_LDS=66000"; do
echo "== $cfg"
env $cfg python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['roofline']['frac'])
"; done
