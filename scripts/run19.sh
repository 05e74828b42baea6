cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
{
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['roofline']['frac'])"
done
} > gpurun_out/run19.log 2>&1; cat gpurun_out/run19.log
