cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4i
export TMPDIR=/tmp
for rep in 1 2; do timeout 900 python -m pytest tests/test_gpu_group.py -q -m gpu -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4; done
timeout 900 python -m pytest tests/test_gpu_dropin.py -q -m gpu -x -k "several" 2>&1 | tail -3
PARTS="c4 c5" bash scripts/r4_final.sh
