# rocprofv3 kernel trace of c5's trim pass at low coverage: which of k_query / k_query4 / k_streak takes the time
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/prof_trim; export TMPDIR=/tmp
for q in 1 0; do
  BFCG_QUERY4=$q timeout -k 5 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_trim/q$q -o p -- python scripts/c4_run.py --batch-reads 16777216 --filter-mode 1 --k 51 --cov ${COV:-2} --trim 1 > gpurun_out/prof_trim/q$q.log 2>&1
  echo "== BFCG_QUERY4=$q"; python tools/rocpd_stats.py gpurun_out/prof_trim/q$q/p_results.db | cut -c1-170 | head -12
done
