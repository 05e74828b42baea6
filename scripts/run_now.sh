cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -6
for i in 1 2; do python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['value'], d['ms_per_step'], d['config']['stage_ms_per_step'], d['roofline']['frac'])
"; done
for e in "BFCG_NO_STREAM=1" "X=1"; do echo "== $e"; env $e timeout 600 python scripts/c3_run.py --b 35 --batch-reads 2097152,2097152 --digest 0 2>&1 | grep -v "^\[c3\]" | cut -c1-420; done
