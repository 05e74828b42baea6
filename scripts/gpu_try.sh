cd $GRAFT_REPO_ROOT
timeout 600 python scripts/ingest_rate.py 100 2>&1 | tail -5
timeout 900 python scripts/gz_rate.py 50 6 4 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_dropin.py -q -m gpu -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
