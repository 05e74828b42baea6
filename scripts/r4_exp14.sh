cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/exp14
export TMPDIR=/tmp
timeout 900 python scripts/c4_run.py --batch-reads 16777216 > gpurun_out/exp14/c4_full.log 2>&1; tail -1 gpurun_out/exp14/c4_full.log | cut -c1-600
bash scripts/r4_exp12.sh 2>&1 | tail -4 | cut -c1-420
