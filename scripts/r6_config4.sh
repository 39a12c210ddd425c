#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for M in "32 2" "32 1" "16 2" "64 1"; do set -- $M; python scripts/config4_tiles.py --many $1 --many-lanes $2 --grids 16 --reps 1 | python -c "
import json,sys; d=json.load(sys.stdin); m=d['many_slices_per_launch']; print('$M', m['one_batch_run_tiles_many_ms'], m['mevents_per_s'], m['tile_iterations_per_s'])"; done
python scripts/config4_tiles.py --many 0 --grids 16 --reps 6 --hw-queues 16 | python -c "
import json,sys; d=json.load(sys.stdin); print('16 grids, 16 queues', json.dumps(d['sustained']))"
timeout 600 python -m pytest tests/test_gpu_config4.py -x -q -k many 2>&1 | tail -2
