cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_baseline_shapes.py -x -q -k "g42 or c2" 2>&1 | tail -5
python -m pytest tests/test_gpu_fuzz.py -x -q -k "large_filters" 2>&1 | tail -3
python bench.py --steps 3 --warmup 1 > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.log; echo bench rc=$?
tail -3 gpurun_out/r2_bench1.log; cut -c1-3000 gpurun_out/r2_bench1.json
bash scripts/prof_round2.sh r2a
