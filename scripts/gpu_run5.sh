cd /root/repo
for g in 1 2 4 8; do python scripts/config4_tiles.py --grids $g --reps 4 | python -c "
import json,sys; d=json.load(sys.stdin); print('grids', d['sustained']['grids_in_flight'], 'single %.2f ms' % d['single_grid']['ms'], 'sustained %.1f Mev/s  %.2f ms/slice  %.3g tile-it/s' % (d['sustained']['mevents_per_s'], d['sustained']['ms_per_slice'], d['sustained']['tile_iterations_per_s']), d['single_grid']['floor'])"; done
