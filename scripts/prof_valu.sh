# one more counter pass of the c3 bench (its own run, kernel trace only beside it): where the SIMDs' issue cycles go, per kernel
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_valu; mkdir -p $OUT; export TMPDIR=/tmp BFCG_SYNC_BATCHES=1
CMD="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-secondary"
timeout -k 5 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD -d $OUT/a -o p -- $CMD > $OUT/a.log 2>&1; echo "rc=$?"
python - <<'PY'
import sqlite3, glob, collections
db = sqlite3.connect(glob.glob('gpurun_out/prof_valu/a/*.db')[0]); cur = db.cursor()
rows = list(cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"))
d = collections.defaultdict(dict)
for k, c, v, n in rows:
    d[k.split('<')[0].split('(')[0].replace('void ', '')][c] = (v, n)
print("| kernel | launches | VALU instr / wave-cycle | ACTIVE_INST_VALU / WAVE_CYCLES | ACTIVE_INST_LDS / WAVE_CYCLES | VALU : SALU : LDS : VMEM_RD instr |")
print("|---|---|---|---|---|---|")
for k, c in sorted(d.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', (0, 0))[0]):
    wc = c.get('SQ_WAVE_CYCLES', (0, 0))[0]
    if wc < 1e9: continue
    g = lambda n: c.get(n, (0, 0))[0]
    print("| %s | %d | %.3f | %.3f | %.3f | %.0f : %.0f : %.0f : %.0f (x1e6) |" % (k, c['SQ_WAVE_CYCLES'][1], g('SQ_INSTS_VALU') / wc, g('SQ_ACTIVE_INST_VALU') / wc, g('SQ_ACTIVE_INST_LDS') / wc,
          g('SQ_INSTS_VALU') / 1e6, g('SQ_INSTS_SALU') / 1e6, g('SQ_INSTS_LDS') / 1e6, g('SQ_INSTS_VMEM_RD') / 1e6))
PY
tail -3 $OUT/a.log | cut -c1-200
