cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
{
for cfg in "BFC_BENCH_NO_RAMP=1" "BFCG_ABLATE=0"; do
    echo "== $cfg"
    env $cfg BFC_BENCH_VERBOSE=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "batch [0-9]|metric" | tail -10 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['roofline']['frac'])
    else: print(l.strip()[:200])"
done
} > gpurun_out/run13.log 2>&1
cat gpurun_out/run13.log
