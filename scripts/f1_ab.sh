# A/B of the level-1 fan-out on c3 (BFCG_F1: 2^F1 level-1 buckets x 2^(F-F1) regions per bucket), same box, self-checked against the golden
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/f1_ab; mkdir -p $OUT
for f1 in ${F1_LIST:-default 8 default 8}; do
  if [ "$f1" != "default" ]; then export BFCG_F1=$f1; else unset BFCG_F1; fi
  timeout -k 5 170 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-boundary > $OUT/f1_${f1}.json 2> $OUT/f1_${f1}.err; echo "F1=${f1} rc=$?"
  python - "$OUT/f1_${f1}.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(" ", d["ms_per_step"], "ms/step", d["value"], d["unit"], "verified", d.get("verified"), d["config"].get("stage_ms_per_step"), d["config"].get("partition"))
PY
done
