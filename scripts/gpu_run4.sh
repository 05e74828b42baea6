cd $GRAFT_REPO_ROOT
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-verify --no-secondary"
show() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], d['value'], d['ms_per_step'], d['config']['batch_reads'], d['config']['library_batches_per_step'], d['config']['slow_buckets'], d['config']['stage_ms_per_step'])
PY
}
$B > gpurun_out/x_a.json 2>/dev/null; show gpurun_out/x_a.json
BFCG_LDS=80000 $B --batch-reads 6291456 > gpurun_out/x_b.json 2>/dev/null; show gpurun_out/x_b.json
BFCG_LDS=80000 $B --batch-reads 4194304 > gpurun_out/x_c.json 2>/dev/null; show gpurun_out/x_c.json
BFCG_LDS=65000 $B --batch-reads 4718592 > gpurun_out/x_d.json 2>/dev/null; show gpurun_out/x_d.json
BFCG_PIPELINE=1 $B > gpurun_out/x_e.json 2>/dev/null; show gpurun_out/x_e.json
