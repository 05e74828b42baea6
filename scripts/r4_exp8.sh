cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/exp8
export TMPDIR=/tmp
O=gpurun_out/exp8
PLANES="1 1 0 1" bash scripts/e2e_c3.sh > $O/e2e_c3.txt 2>&1; grep -E "==|Real time|^real|device buffers|clean-up|waited" $O/e2e_c3.txt | cut -c1-260
timeout 900 python -m pytest tests/test_gpu_dropin.py -q -x -m gpu 2>&1 | tail -3
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/bench.log; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d.get('verified')); print(json.dumps(d.get('pcie'))[:400]); print(json.dumps(d.get('pcie_planes'))[:600]); print(json.dumps(d.get('e2e'))[:900])"
