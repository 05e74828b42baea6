cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
{
for cfg in "BFCG_PF=4" "BFCG_PF=3" "BFCG_PF=2"; do
echo "== $cfg"; env $cfg python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['roofline']['frac'])"
done
} > gpurun_out/run19.log 2>&1; cat gpurun_out/run19.log
