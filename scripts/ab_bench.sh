#!/bin/bash
# A/B of two builds of the library on one box: bench.py's headline regime, alternating.
# usage: ab_bench.sh <libA> <libB> [rounds] [extra bench.py arguments]
R=${GRAFT_REPO_ROOT:-/root/repo}
A=$1; B=$2; N=${3:-3}; shift 3
for i in $(seq $N); do
  for L in $A $B; do
    BF_ACCEL_LIB=$L timeout 300 python $R/bench.py --no-cpu-baseline --no-front-end --steps 10 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d.get('roofline', {})
oc = d.get('regimes', {}).get('one_context', {})
print('$L'.split('/')[-1], 'value %.1f  one context cold %.1f warm %.0f  it/slice %.1f  K1 %.2f us  K3 %.2f us  chip-full K1 %.2f K3 %.2f' % (
    d['value'], oc.get('cold', {}).get('mevents_per_s', 0), oc.get('warm_stm', {}).get('mevents_per_s', 0), d['config']['iterations_per_slice'],
    r.get('per_kernel_us', {}).get('warp_scatter', 0), r.get('per_kernel_us', {}).get('stencil_moments_update', 0),
    r.get('chip_full', {}).get('warp_scatter_us_per_1M_events', 0), r.get('chip_full', {}).get('stencil_us_per_config2_image', 0)))
"
  done
done
