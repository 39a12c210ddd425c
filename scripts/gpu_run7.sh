cd /root/repo
bash scripts/profile_r3.sh > gpurun_out/prof_r3.log 2>&1
tail -2 gpurun_out/prof_r3.log
python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err
python scripts/config4_tiles.py --grids 4 --reps 6 > gpurun_out/r3_config4.json
python scripts/config3_stream.py > gpurun_out/r3_config3.json 2>/dev/null
python scripts/front_end_bench.py --slices 20 > gpurun_out/r3_front_end_346.json
python scripts/front_end_bench.py --slices 10 --height 480 --width 640 > gpurun_out/r3_front_end_640.json
python scripts/front_end_bench.py --slices 10 -o > gpurun_out/r3_front_end_flow.json
python scripts/sweep_geometry.py > gpurun_out/r3_geometry_sweep.txt 2>&1
python bench.py --config 5 --farm-slices 16 > gpurun_out/r3_config5_1gpu.json 2>/dev/null
tail -c 400 gpurun_out/r3_config5_1gpu.json
