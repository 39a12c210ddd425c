cd /root/repo
for o in "--opt bin_tile_rows=64" "--opt bin_tile_rows=96" "--opt bin_tile_rows=128" "--opt bin_tile_rows=32 --opt bin_threads=256" "--opt bin_tile=128" "--opt bin_tile=128 --opt bin_tile_rows=32"; do
  python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-front-end $o 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$o: value %.1f h2h %.1f' % (d['value'], d['value_host_to_host']))"
done
