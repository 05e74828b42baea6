cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/exp16
export TMPDIR=/tmp
# the shape / parity / group / drop-in tests once more with table blocks of 32 slots: every table that takes region-owned segments runs them as many blocks
BFCG_SEG_BLOCK=5 timeout 1200 python -m pytest tests/test_gpu_baseline_shapes.py tests/test_gpu_parity.py tests/test_gpu_group.py tests/test_gpu_dropin.py -q -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > gpurun_out/exp16/gpu_tests_blk5.log; tail -6 gpurun_out/exp16/gpu_tests_blk5.log
PARTS="tests bench dist prof" bash scripts/r4_final.sh
