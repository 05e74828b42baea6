cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/exp11
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_baseline_shapes.py -q -x -m gpu -k "larger_than_lds" 2>&1 | tail -15
timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -x -m gpu -k "several_blocks" 2>&1 | tail -15
