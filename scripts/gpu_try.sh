cd $GRAFT_REPO_ROOT
echo "== faults"; timeout 600 python scripts/ingest_rate.py 100 2>&1 | tail -4
echo "== MADV_POPULATE_READ per slice"; BFC_INGEST_POPULATE=1 timeout 600 python scripts/ingest_rate.py 100 2>&1 | tail -4
uname -r
