cd /root/repo
mkdir -p gpurun_out/r3b
python -m pytest tests/test_gpu_config4.py tests/test_host_cli.py -m gpu -q -s 2>&1 | tail -25
for extra in "" "--sync"; do python scripts/front_end_bench.py --slices 20 --extra="$extra"; done
python scripts/front_end_bench.py --slices 10 -o
python - <<'PY'
import sys; sys.path.insert(0,'.')
from better_flow_amd import synth
print(synth.write_stream_bin('/tmp/s480.bin', 8, 1000000, 480, 640))
PY
F="--quiet --timing --res-x=480 --res-y=640 --max-events=1100000 --span=0.03 --refresh-time=0.03 --refresh-event-count=1000000000"
BF_FARM_TIMING=1 better_flow_amd/host/bf_motion_compensator $F /tmp/s480.bin 2>&1 | tail -12
