# A/B of two K3 variants: $1 = env assignment selecting the OLD variant (e.g. BF_K3_POSMAP=1)
cd /root/repo
OLD="$1"
echo "== bits"; env $OLD python scripts/bits_check.py | tail -1; python scripts/bits_check.py | tail -1
for v in old new; do
  if [ $v = old ]; then E="$OLD"; else E="BF_DUMMY=1"; fi
  env $E python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-front-end > /tmp/b_$v.json 2>/tmp/b_$v.err || tail -5 /tmp/b_$v.err
  python - <<PY
import json
d=json.load(open('/tmp/b_$v.json'))
r=d['roofline']; oc=d['regimes']['one_context']
print('$v', 'value %.1f' % d['value'], 'one ctx cold %.1f Mev/s %.2f ms' % (oc['cold']['mevents_per_s'], oc['cold']['ms_per_slice']), 'K1 %.2f us K3 %.2f us (tail mode)' % (r['per_kernel_us']['warp_scatter'], r['per_kernel_us']['stencil_moments_update']), 'warm4 %.0f' % d['regimes']['warm_stm']['mevents_per_s'], 'iters', d['config']['iterations_per_slice'])
PY
done
