#!/bin/bash
# round 6, first look: how sparse the 1280x720 time image is along a cold run (what a wave-level skip could save), the loop
# kernels per geometry, and the stencil kernel's phases at 1280x720 in the tail-update (co-scheduled) form.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_probe1; mkdir -p $O
cd $R
python scripts/density_probe.py 720 1280 > $O/density_720p.txt 2>&1
for G in "260 346" "480 640" "720 1280"; do
  python scripts/kernel_time.py $G iters=200 | tail -1
  python scripts/kernel_time.py $G iters=200 co_schedule=1 | tail -1
done > $O/kernel_time.txt 2>&1
BF_RUN_H=720 BF_RUN_W=1280 BF_CO=1 TL_LAUNCH=30 python scripts/timeline_k3.py > $O/timeline_720p_co.txt 2>&1
BF_RUN_H=720 BF_RUN_W=1280 TL_LAUNCH=30 python scripts/timeline_k3.py > $O/timeline_720p.txt 2>&1
tail -n 20 $O/*.txt
