set -x
cd /root/repo
mkdir -p gpurun_out/r3a
python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r3a/pytest.txt
for extra in "" "--sync" "--threads=1" "--threads=4"; do
  python scripts/front_end_bench.py --slices 20 --extra="$extra" >> gpurun_out/r3a/front_end.txt 2>&1
done
python scripts/front_end_bench.py --slices 10 -o >> gpurun_out/r3a/front_end.txt 2>&1
python scripts/front_end_bench.py --slices 10 --height 480 --width 640 >> gpurun_out/r3a/front_end.txt 2>&1
cat gpurun_out/r3a/pytest.txt; cat gpurun_out/r3a/front_end.txt
