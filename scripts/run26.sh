cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
: > gpurun_out/run26.log
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 >> gpurun_out/run26.log
for lib in build/old/libbfc_gpu.so bfc_amd/libbfc_gpu.so; do
  echo "== $lib : c2 read set, filter mode k=31 b=33; then c3-like 10x k=51 b=35" >> gpurun_out/run26.log
  BFC_GPU_LIB=$PWD/$lib timeout 600 python scripts/c3_run.py --filter-mode 1 --k 31 --b 33 --G 4600000 --cov 100 --seed 2 --batch-reads 786432,524288 2>&1 | grep -v "^\[c3\] [0-9]" | cut -c1-700 >> gpurun_out/run26.log
  BFC_GPU_LIB=$PWD/$lib timeout 600 python scripts/c3_run.py --filter-mode 1 --k 51 --b 35 --cov 10 --batch-reads 2097152 2>&1 | grep -v "^\[c3\] [0-9]" | cut -c1-700 >> gpurun_out/run26.log
done
python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-300 >> gpurun_out/run26.log
cat gpurun_out/run26.log
