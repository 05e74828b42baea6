# c4 at 12x coverage: the size of a table segment's block (BFCG_SEG_BLOCK): 2^13-slot segments as ONE block (1024 threads, compare-and-swap path) against
# two blocks of 2^12 (512 threads, counter pairs) or four of 2^11 (256 threads, counter pairs; every block's workgroup reads all the region's entries)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/exp13
export TMPDIR=/tmp
O=gpurun_out/exp13
for blk in 14 12 11; do
  echo "== BFCG_SEG_BLOCK=$blk"
  BFCG_SEG_BLOCK=$blk timeout 900 python scripts/c4_run.py --cov 12 --batch-reads 16777216 > $O/c4_cov12_blk$blk.log 2>&1
  tail -1 $O/c4_cov12_blk$blk.log | python -c "
import sys, json; d=json.loads(sys.stdin.read()); print(d['gpu_stage_ms'], d['gpu_s'], d['n_keys'], d['n_seen'])"
done
