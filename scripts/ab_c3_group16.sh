cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do python scripts/bench_configs.py --only c3 --debug-set 12=$v 2>/dev/null | cut -c1-220; done
