cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
{
for cfg in "BFCG_PRIO=1" "BFCG_PRIO=0" "BFCG_PRIO=-1"; do
    echo "== $cfg"
    for i in 1 2; do env $cfg python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'])"; done
done
} > gpurun_out/run11.log 2>&1
cat gpurun_out/run11.log
