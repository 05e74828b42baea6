cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
python -m pytest tests/test_kcov.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/run20.log
python scripts/kcov_rate.py >> gpurun_out/run20.log 2>&1
cat gpurun_out/run20.log
