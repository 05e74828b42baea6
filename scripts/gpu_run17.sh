cd $GRAFT_REPO_ROOT
BFCG_DEBUG=1 timeout 600 python scripts/c3_run.py --b 33 --batch-reads 1048576 --digest 0 2>&1 | grep -v "D::reset\|note_growth" | tail -25 | cut -c1-300
