cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
: > gpurun_out/run27.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | tail -12 >> gpurun_out/run27.log
python scripts/trim_rate.py 51 33 2>&1 | tail -2 >> gpurun_out/run27.log
python scripts/trim_rate.py 51 37 2>&1 | tail -2 >> gpurun_out/run27.log
cat gpurun_out/run27.log
