#!/bin/bash
# Everything profiles/r6_* is refreshed from, on one box (run through gpurun; then `python scripts/collect_r6.py` here and
# copy gpurun_out/final6/*.json|txt into profiles/).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; mkdir -p gpurun_out/final6; O=$R/gpurun_out/final6
(time python -m pytest tests -m gpu -x -q) > $O/gputest.log 2>&1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 1800 bash scripts/profile_r6.sh > $O/profile.log 2>&1
cd $R
timeout 300 python scripts/config3_stream.py > $O/config3.json 2>&1
timeout 300 python scripts/warm_chain_trace.py 480 640 24 bytes=12 defer_uploads=1 ahead=2 > $O/config3_chain.txt 2>&1
for i in 1 2; do python scripts/h2h_warm.py 1 ahead=2 defer=1; python scripts/h2h_warm.py 1 ahead=1 defer=0; python scripts/h2h_warm.py 4 ahead=2 defer=1; python scripts/h2h_warm.py 4 ahead=1 defer=0; python scripts/h2h_warm.py 1 ahead=2 defer=1 bytes=12; done > $O/h2h_warm.txt 2>&1
timeout 300 python scripts/config4_tiles.py --many 32 --many-lanes 2 --grids 4 --reps 4 > $O/config4.json 2>&1
timeout 300 python scripts/config4_tiles.py --many 0 --grids 16 --reps 6 --hw-queues 16 > $O/config4_16queues.json 2>&1
timeout 600 python bench.py --config 5 --farm-slices 16 --no-cpu-baseline > $O/config5.json 2>&1
timeout 600 python scripts/sweep_geometry.py > $O/sweep.txt 2>&1
timeout 300 python scripts/front_end_bench.py > $O/front_end.json 2>&1
timeout 300 python scripts/density_probe.py 720 1280 > $O/density_720p.txt 2>&1
tail -3 $O/gputest.log
tail -c 300 $O/bench.json
