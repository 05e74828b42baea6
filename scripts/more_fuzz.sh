#!/bin/bash
# tests/test_gpu_fuzz.py again with other seeds: every base draws 210 other random configurations (GPU vs oracle, bit for bit)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for base in "$@"; do
  echo "== seed base $base"; BFC_FUZZ_SEED_BASE=$base timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -3
done 2>&1 | tee gpurun_out/more_fuzz.log
