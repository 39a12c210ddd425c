cd $GRAFT_REPO_ROOT
python scripts/warm_chain_trace.py 260 346 30 bytes=8
python scripts/warm_chain_trace.py 480 640 24 bytes=12
