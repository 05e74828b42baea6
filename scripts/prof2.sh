cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/prof2
export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 5 200 rocprofv3 --kernel-trace "$@" -d gpurun_out/prof2/$name -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof2/$name.log 2>&1; echo "$name rc=$?"; }
run trace
run pmc1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS
run pmc2 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
run pmc3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE
run pmc4 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
ls gpurun_out/prof2/*
