cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_dropin.py -q -m gpu -x -k "damaged" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8
