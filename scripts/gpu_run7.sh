cd $GRAFT_REPO_ROOT
show() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], d['value'], d['ms_per_step'], d['n_gpus'], d['scaling'], d['config']['parallelism'][:150], d['config']['stage_ms_per_step'])
PY
}
BFC_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-secondary > gpurun_out/x_f.json 2> gpurun_out/x_f.log; echo rc=$?; tail -3 gpurun_out/x_f.log; show gpurun_out/x_f.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-secondary --workload c2 > gpurun_out/x_g.json 2> gpurun_out/x_g.log; echo rc=$?; tail -3 gpurun_out/x_g.log; show gpurun_out/x_g.json
