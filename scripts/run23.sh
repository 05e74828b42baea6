cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export BFCG_SYNC_BATCHES=1
: > gpurun_out/run23.log
for ab in 0 1 2 64; do
  echo "== BFCG_ABLATE=$ab" >> gpurun_out/run23.log
  BFCG_ABLATE=$ab timeout 600 python scripts/c3_run.py --b 35 --batch-reads 2097152 --digest 0 --cov ${COV:-10} 2>&1 | grep -v "^\[c3\]" | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()); continue
    print(d['G_kmers_per_s'], d['batches'], d['stage_ms'], d['n_seen'], d['n_keys'])
" >> gpurun_out/run23.log 2>&1
done
cat gpurun_out/run23.log
