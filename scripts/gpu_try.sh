cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_baseline_shapes.py -q -m gpu -x -k "cliff" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -12
