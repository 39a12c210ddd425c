cd /root/repo
(time python -m pytest tests -m gpu -q 2>&1 | tail -5) 2>&1 | tail -9
python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_bench.json'))
r=d['roofline']
print('value',d['value'],'h2h',d['value_host_to_host'],'frac',r['frac'],r['avg_launch_us'],'shared',r['co_scheduled_shape'],'iter_frac',r['iteration_frac'],'copy',r['measured_copy_ceiling_gbps'])
print(d['front_end']['steady_state_warm'], d['front_end']['file_to_last_model']['mevents_per_s'])
PY
