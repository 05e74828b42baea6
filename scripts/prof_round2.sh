# rocprofv3 evidence for the numbers bench.py prints (round 2: the headline workload is c3).  Kernel trace of the SAME command
# (pipelined, as benchmarked), a trace with BFCG_SYNC_BATCHES=1 (one batch at a time, no kernel overlap) so that durations can be
# attributed to single kernels, and -- with PMC=1 -- counter passes of that serial run (counters in their own runs, never combined
# with other trace domains).  Always under `timeout`: a rocprofv3 run once hung after finishing.
#   bash scripts/prof_round2.sh [tag] ; summaries: python tools/make_round_md.py gpurun_out/prof_<tag>
cd $GRAFT_REPO_ROOT
TAG=${1:-r2}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline --no-verify --no-secondary --no-boundary ${BENCH_ARGS:-}"
run() { name=$1; shift; timeout -k 5 ${PROF_TIMEOUT:-400} rocprofv3 --kernel-trace "$@" -d $OUT/$name -o p -- $CMD > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run trace --stats
export BFCG_SYNC_BATCHES=1
run trace_sync --stats
if [ -n "$PMC" ]; then
  run pmc_fetch --pmc FETCH_SIZE GRBM_GUI_ACTIVE
  run pmc_write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
  [ "$PMC" = "2" ] && run pmc_sq --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
fi
grep -h '"metric"' $OUT/trace.log $OUT/trace_sync.log | cut -c1-300
for n in trace trace_sync; do echo "== $n"; python tools/rocpd_stats.py $OUT/$n/p_results.db | cut -c1-170 | head -24; done
