cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/exp2
export TMPDIR=/tmp
O=gpurun_out/exp2
timeout 300 build/commit_probe 17 13 1335 2 > $O/commit_probe_2p.txt 2>&1; tail -6 $O/commit_probe_2p.txt | cut -c1-200
bash scripts/e2e_c3.sh > $O/e2e_c3.txt 2>&1; tail -30 $O/e2e_c3.txt | cut -c1-250
timeout 900 python scripts/c4_run.py --cov 12 --batch-reads 16777216 > $O/c4_cov12.log 2>&1; tail -2 $O/c4_cov12.log | cut -c1-900
