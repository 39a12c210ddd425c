cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -k "incremental_warp or warp_state or local_window or config4" 2>&1 | tail -4
