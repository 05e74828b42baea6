# SQ counters of the c3 kernels (two PMC passes, serial batches), summarised per kernel: bash scripts/prof_sq.sh [tag]
cd $GRAFT_REPO_ROOT
TAG=${1:-sq}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp BFCG_SYNC_BATCHES=1
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-secondary --no-boundary ${BENCH_ARGS:-}"
run() { name=$1; shift; timeout -k 5 ${PROF_TIMEOUT:-300} rocprofv3 --kernel-trace "$@" -d $OUT/$name -o p -- $CMD > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
run sq2 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_VMEM_WR
for n in sq1 sq2; do python tools/rocpd_pmc.py $OUT/$n/p_results.db; done > $OUT/summary.txt
grep -A9 "k_scatter1\|k_bloom\|k_scatter2\|k_commit_seg" $OUT/summary.txt | cut -c1-110
