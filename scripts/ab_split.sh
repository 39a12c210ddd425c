# A/B of the interior + margin format (bin_split) against the dense slabs: bits, per-kernel times, bench
cd /root/repo
mkdir -p gpurun_out/split
echo "== bits (must be equal)"
python scripts/bits_check.py | tail -1
BF_ACCEL_OPTIONS=bin_split=2 python scripts/bits_check.py | tail -1
echo "== kernel times"
for g in "260 346" "480 640"; do
  for co in 0 1; do for sp in 0 2; do python scripts/kernel_time.py $g bin_split=$sp co_schedule=$co iters=200; done; done
done
if [ "$1" = "tests" ]; then
echo "== parity tests with bin_split=2"
BF_ACCEL_OPTIONS=bin_split=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_geometries.py -m gpu -x -q 2>&1 | tail -5
fi
echo "== bench"
for sp in 0 2; do
  BF_ACCEL_OPTIONS=bin_split=$sp python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-front-end > gpurun_out/split/b_$sp.json 2>gpurun_out/split/b_$sp.err || tail -5 gpurun_out/split/b_$sp.err
  python - <<PY
import json
d=json.load(open('gpurun_out/split/b_$sp.json'))
r=d['roofline']; oc=d['regimes']['one_context']
print('split=$sp', 'value %.1f' % d['value'], 'h2h %.1f' % d['value_host_to_host'], 'one ctx cold %.1f Mev/s %.2f ms' % (oc['cold']['mevents_per_s'], oc['cold']['ms_per_slice']), 'K1 %.2f us K3 %.2f us (tail mode)' % (r['per_kernel_us']['warp_scatter'], r['per_kernel_us']['stencil_moments_update']), 'warm4 %.0f' % d['regimes']['warm_stm']['mevents_per_s'], 'iters', d['config']['iterations_per_slice'])
PY
done
python scripts/config3_stream.py 2>&1 | tail -3
BF_ACCEL_OPTIONS=bin_split=2 python scripts/config3_stream.py 2>&1 | tail -3
