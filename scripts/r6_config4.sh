#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_config4; mkdir -p $O; cd $R
for M in "16 2" "16 1" "32 2" "64 1"; do set -- $M; python scripts/config4_tiles.py --many $1 --many-lanes $2 --grids 16 --reps 1 > $O/many_$1_$2.json 2>$O/err_$1_$2.txt; python -c "
import json,sys; d=json.load(open('$O/many_$1_$2.json')); print('$M', json.dumps(d['many_slices_per_launch']))"; done
