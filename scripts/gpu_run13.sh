cd $GRAFT_REPO_ROOT
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-verify --no-secondary"
show() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], d['value'], d['ms_per_step'], d['config']['host_ms_per_step_enqueue'], d['config']['stage_ms_per_step']['total'])
PY
}
BFCG_DEBUG=1 $B > gpurun_out/x_a.json 2>gpurun_out/x_a.log; show gpurun_out/x_a.json; grep "D::" gpurun_out/x_a.log | head -40
$B > gpurun_out/x_b.json 2>/dev/null; show gpurun_out/x_b.json
