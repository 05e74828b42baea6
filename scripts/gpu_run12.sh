cd $GRAFT_REPO_ROOT
B="python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-verify --no-secondary"
show() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], d['value'], d['ms_per_step'], d['config']['batch_reads'], d['config']['library_batches_per_step'], d['config']['slow_buckets'], d['config']['host_ms_per_step_enqueue'], d['config']['stage_ms_per_step'])
PY
}
$B > gpurun_out/x_c3.json 2>/dev/null; show gpurun_out/x_c3.json
for br in 786432 917504 1572864 3145728; do $B --workload c2 --batch-reads $br > gpurun_out/x_c2_$br.json 2>/dev/null; show gpurun_out/x_c2_$br.json; done
