cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/exp15
export TMPDIR=/tmp
PARTS="bench" bash scripts/r4_final.sh
# the whole GPU suite once more with table blocks of 32 slots: every test that takes region-owned segments runs them as many blocks
BFCG_SEG_BLOCK=5 timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > gpurun_out/exp15/gpu_tests_blk5.log; tail -4 gpurun_out/exp15/gpu_tests_blk5.log
