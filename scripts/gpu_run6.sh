cd /root/repo
export BF_ACCEL_LIB=/root/repo/better_flow_amd/libbf_accel_exp.so
for geo in "260 346" "480 640" "720 1280 bin_compact=0"; do
  python scripts/kernel_time.py $geo
  BF_K3_ROWMAP=1 python scripts/kernel_time.py $geo
done
unset BF_ACCEL_LIB
for v in 0 1; do
  if [ $v = 1 ]; then export BF_K3_ROWMAP=1; fi
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-front-end --height 480 --width 640 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('rowmap $v 640x480: value %.1f Mev/s, iters %.0f' % (d['value'], d['config']['iterations_per_slice']))"
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-front-end 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('rowmap $v 346x260: value %.1f Mev/s' % d['value'])"
done
