# rocprofv3 kernel trace of config c3 (scripts/c3_run.py, one batch size, no host digests): pipelined as run, and one batch at a time
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/prof_c3
export TMPDIR=/tmp
BR=${1:-2097152}
CMD="python scripts/c3_run.py --b 35 --batch-reads $BR --digest 0"
run() { name=$1; shift; timeout -k 5 400 rocprofv3 --kernel-trace "$@" -d gpurun_out/prof_c3/$name -o p -- $CMD > gpurun_out/prof_c3/$name.log 2>&1; echo "$name rc=$?"; }
run trace
export BFCG_SYNC_BATCHES=1
run trace_sync
run pmc_fetch --pmc FETCH_SIZE GRBM_GUI_ACTIVE
run pmc_write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
grep -h 'batch_reads' gpurun_out/prof_c3/trace.log gpurun_out/prof_c3/trace_sync.log | cut -c1-600
for n in trace trace_sync; do echo "== $n"; python tools/rocpd_stats.py gpurun_out/prof_c3/$n/p_results.db | cut -c1-160; done
