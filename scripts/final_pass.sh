#!/bin/bash
# Everything profiles/r2_* is made from, on one box (run through gpurun; then `python scripts/collect_r2.py` here and copy
# gpurun_out/final/*.json|txt into profiles/).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; mkdir -p gpurun_out/final
timeout 600 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
timeout 1500 bash scripts/profile_r2.sh > gpurun_out/final/profile.log 2>&1
cd $R
timeout 300 python scripts/config3_stream.py > gpurun_out/final/config3.json 2>&1
timeout 300 python scripts/config4_tiles.py > gpurun_out/final/config4.json 2>&1
timeout 600 python bench.py --config 5 --farm-slices 16 --no-cpu-baseline > gpurun_out/final/config5.json 2>&1
timeout 600 python scripts/sweep_geometry.py > gpurun_out/final/sweep.txt 2>&1
BF_RUN_H=720 BF_RUN_W=1280 python scripts/run_once.py 2 >> gpurun_out/final/sweep.txt 2>&1
python scripts/warm_profile.py > gpurun_out/final/warm_profile.txt 2>&1
bash scripts/warm_trace.sh > gpurun_out/final/warm_trace.txt 2>&1
tail -c 300 gpurun_out/final/bench.json
