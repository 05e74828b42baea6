# A/B of level 1 with write-combining buffers (BFCG_S1_WC=1: k_scatter1_wc) against the default k_scatter1 on c3, same box, alternating, verified
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/wc_ab; mkdir -p $OUT
i=0
for v in ${WC_LIST:-0 1 0 1}; do
  i=$((i+1))
  if [ "$v" != "0" ]; then export BFCG_S1_WC=$v; else unset BFCG_S1_WC; fi
  timeout -k 5 170 python bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-secondary --no-boundary > $OUT/wc_${v}_$i.json 2> $OUT/wc_${v}_$i.err; echo "BFCG_S1_WC=${v} rc=$?"
  python - "$OUT/wc_${v}_$i.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(" ", d["ms_per_step"], "ms/step", d["value"], d["unit"], "verified", d.get("verified"), d["config"].get("stage_ms_per_step"), d["config"].get("partition"), "whole_job_frac", d["roofline"].get("whole_job_frac"), "path frac", d["roofline"].get("frac"))
PY
done
