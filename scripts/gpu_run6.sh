cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_group.py tests/test_gpu_dropin.py tests/test_multi_rank.py -q -m gpu -x 2>&1 | tail -15
