cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_dropin.py -q -m gpu -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -5
timeout 900 python scripts/gz_rate.py 50 6 2>&1 | tee gpurun_out/gz_rate.txt
