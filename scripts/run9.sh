cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/run9.log; cat gpurun_out/run9.log
