# L2 / memory-side counters of k_scatter1 alone (scripts/s1_ablate.py), one-pass against two-pass partition: bash scripts/prof_s1_tcc.sh
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_s1tcc
mkdir -p $OUT
export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 5 200 rocprofv3 --kernel-trace "$@" -d $OUT/$name -o p -- python scripts/s1_ablate.py > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
for mode in 1 0; do
  export BFCG_ONEPASS=$mode B=${B:-37}
  run m${mode}_a --pmc FETCH_SIZE WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
  run m${mode}_b --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_WRITEBACK_sum
  run m${mode}_c --pmc TCC_REQ_sum TCC_WRITE_sum TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum
  for n in a b c; do python tools/rocpd_pmc.py $OUT/m${mode}_$n/p_results.db; done > $OUT/summary_m$mode.txt
  grep -A7 "k_scatter1\|k_hist1" $OUT/summary_m$mode.txt | cut -c1-140
  tail -2 $OUT/m${mode}_a.log
done
