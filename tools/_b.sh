cd $GRAFT_REPO_ROOT
for w in graph48 eager_sweep graph_sweep; do DBG=$w python tools/_dbg.py 2>/dev/null; done
