#!/bin/bash
# Everything profiles/r6_* is refreshed from, on one box (run through gpurun; then `python scripts/collect_r6.py` here and
# copy gpurun_out/final6/*.json|txt into profiles/).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; mkdir -p gpurun_out/final6
(time python -m pytest tests -m gpu -x -q) > gpurun_out/final6/gputest.log 2>&1
timeout 600 python bench.py > gpurun_out/final6/bench.json 2> gpurun_out/final6/bench.err
timeout 1800 bash scripts/profile_r6.sh > gpurun_out/final6/profile.log 2>&1
cd $R
timeout 300 python scripts/config3_stream.py > gpurun_out/final6/config3.json 2>&1
timeout 300 python scripts/config4_tiles.py > gpurun_out/final6/config4.json 2>&1
timeout 600 python bench.py --config 5 --farm-slices 16 --no-cpu-baseline > gpurun_out/final6/config5.json 2>&1
timeout 600 python scripts/sweep_geometry.py > gpurun_out/final6/sweep.txt 2>&1
timeout 300 python scripts/front_end_bench.py > gpurun_out/final6/front_end.json 2>&1
tail -3 gpurun_out/final6/gputest.log
tail -c 300 gpurun_out/final6/bench.json
