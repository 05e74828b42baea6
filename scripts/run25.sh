cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
: > gpurun_out/run25.log
BFC_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/run25.out 2>gpurun_out/run25.err
grep '"metric"' gpurun_out/run25.out | python -c '
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d["value"], d["ms_per_step"], d["config"]["stage_ms_per_step"], d["roofline"]["frac"], d["config"]["n_seen"], d["config"]["n_distinct"], d["config"]["parallelism"][:40], d["config"]["exchange_plus_stages_s_per_step"])
' >> gpurun_out/run25.log 2>&1
grep -v '"metric"' gpurun_out/run25.out | head -5 >> gpurun_out/run25.log
tail -3 gpurun_out/run25.err >> gpurun_out/run25.log
cat gpurun_out/run25.log
