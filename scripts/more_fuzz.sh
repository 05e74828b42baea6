#!/bin/bash
# More seeds for the randomised GPU parity tests.  bash scripts/more_fuzz.sh BASE...
#   default: tests/test_gpu_fuzz.py, 210 other random configurations per base (GPU vs oracle, bit for bit)
#   MORE_DROPIN=1: instead the drop-in binaries against the reference binary on damaged files (count dump + trim output), 16 per base
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for base in "$@"; do
  if [ -n "$MORE_DROPIN" ]; then
    echo "== drop-in, seed base $base"; BFC_FUZZ_SEED_BASE=$base timeout 900 python -m pytest tests/test_gpu_dropin.py -q -m gpu -x -k damaged 2>&1 | tail -2
  else
    echo "== seed base $base"; BFC_FUZZ_SEED_BASE=$base timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -3
  fi
done 2>&1 | tee gpurun_out/more_fuzz.log
