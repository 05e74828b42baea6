cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
{
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py --steps 3 --warmup 1 --cov 20 --no-cpu-baseline
python bench.py --steps 3 --warmup 1
} > gpurun_out/run2.log 2>&1
tail -30 gpurun_out/run2.log
