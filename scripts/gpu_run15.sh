cd $GRAFT_REPO_ROOT
python scripts/bloom_phases.py 2>&1 | grep batch
