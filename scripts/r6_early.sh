cd $GRAFT_REPO_ROOT
for i in 1 2; do python scripts/config3_stream.py | python -c "
import json,sys; d=json.load(sys.stdin); print({k:(round(v['steady_ms_per_slice'],4),round(v['steady_mevents_per_s'])) for k,v in d['modes'].items()}, d['same_result_all_modes'])"; done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
