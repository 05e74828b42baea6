# all GPU tests on the box gpurun provides:  gpurun --timeout 900 -- 'bash scripts/gpu_tests.sh'
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/gpu_tests.log; cat gpurun_out/gpu_tests.log
