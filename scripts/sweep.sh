#!/bin/bash
# rocprofv3 kernel-trace sweep over tile-binned scatter configurations (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  rm -rf /tmp/sweep_prof
  timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/sweep_prof -- python /root/repo/scripts/run_once.py 1 $cfg > /tmp/sweep.log 2>&1
  echo "=== $cfg : $(grep iters /tmp/sweep.log | head -1)"
  timeout 60 python /root/repo/scripts/analyze_trace.py /tmp/sweep_prof 2>&1 | grep -E "k_bin_warp_scatter<true|k_stencil|k_update|iteration period|k_iter|k_warp_scatter"
done
