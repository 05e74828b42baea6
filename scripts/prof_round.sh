# rocprofv3 evidence for the numbers bench.py prints: kernel trace of the SAME command (pipelined, as benchmarked), plus a
# trace and PMC passes with BFCG_SYNC_BATCHES=1 (one batch at a time, no kernel overlap) so that counters and durations
# can be attributed to single kernels.  Always under `timeout`: a rocprofv3 run once hung after finishing.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/prof_r1
export TMPDIR=/tmp
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
run() { name=$1; shift; timeout -k 5 240 rocprofv3 --kernel-trace "$@" -d gpurun_out/prof_r1/$name -o p -- $CMD > gpurun_out/prof_r1/$name.log 2>&1; echo "$name rc=$?"; }
run trace
export BFCG_SYNC_BATCHES=1
run trace_sync
run pmc_fetch --pmc FETCH_SIZE GRBM_GUI_ACTIVE
run pmc_write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run pmc_sq --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
grep -h '"metric"' gpurun_out/prof_r1/trace.log gpurun_out/prof_r1/trace_sync.log | cut -c1-200
