cd /root/repo
for i in 1 2; do python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-front-end 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('value %.1f h2h %.1f warm4 %.0f one-ctx cold %.1f warm1 %.0f  K1 %.2f K3 %.2f' % (d['value'], d['value_host_to_host'], d['regimes']['warm_stm']['mevents_per_s'], d['regimes']['one_context']['cold']['mevents_per_s'], d['regimes']['one_context']['warm_stm']['mevents_per_s'], d['roofline']['per_kernel_us']['warp_scatter'], d['roofline']['per_kernel_us']['stencil_moments_update']))"; done
python scripts/config3_stream.py 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:(round(v['steady_ms_per_slice'],3), round(v['steady_mevents_per_s'])) for k,v in d['modes'].items()})"
