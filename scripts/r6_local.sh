cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_config4.py -x -q -k "local_window" -s 2>&1 | tail -12
python scripts/config4_tiles.py --many 0 --grids 4 --reps 2 | python -c "
import json,sys; d=json.load(sys.stdin); print(json.dumps(d['local_windows'])); print(d['flow_median_px_s'], d['injected_px_s'])"
