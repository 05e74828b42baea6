# scripts/probes/scatter_probe.hip on the GPU box: times + self-check, then one counter pass (WRITE_SIZE per kernel: the write amplification
# of runs of ~6 records against whole chunks).  gpurun --timeout 300 -- 'bash scripts/scatter_probe.sh'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/scatter_probe; mkdir -p $OUT build
[ -x build/scatter_probe ] || hipcc --offload-arch=gfx950 -O3 -o build/scatter_probe scripts/probes/scatter_probe.hip
timeout -k 5 200 build/scatter_probe 554000000 all > $OUT/run.txt 2>&1; echo "rc=$?" >> $OUT/run.txt
cat $OUT/run.txt
timeout -k 5 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc -o p -- build/scatter_probe > $OUT/pmc.log 2>&1; echo "pmc rc=$?"
python - <<'PY' 2>&1 | tee gpurun_out/scatter_probe/write_size.txt
import sqlite3, glob, re
dbs = glob.glob('gpurun_out/scatter_probe/pmc/**/*.db', recursive=True)
if not dbs: raise SystemExit("no rocprofv3 database")
cur = sqlite3.connect(dbs[0]).cursor()
n = 554000000
print("WRITE_SIZE x 1024 per launch against the records' %d x 12 bytes (last 3 of 4 launches averaged by sum/count over all 4):" % n)
for k, v, c in cur.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name = 'WRITE_SIZE' group by kernel_name order by min(dispatch_id)"):
    if "k_tile" not in k and "k_wc" not in k: continue
    name = re.sub(r'^void ', '', k).split('(')[0]
    print("%-44s launches %d  %.3f GB per launch = %.2f x the records" % (name, c, v * 1024 / c / 1e9, v * 1024 / c / (n * 12.0)))
PY
tail -2 $OUT/pmc.log | cut -c1-200
