cd $GRAFT_REPO_ROOT
STEPS=1 bash scripts/prof_round2.sh r2d 2>&1 | grep -v "k_colsum\|k_apply\|k_scan\|__amd" | tail -22
