#!/bin/bash
# A/B of two builds on one box: every bit (bits_check.py), then the two loop kernels per geometry (kernel_time.py: one context alone,
# the kernels' own timestamps), libraries alternating.  usage: ab_k3.sh <libA> <libB> [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}
A=$1; B=$2; N=${3:-2}
for L in $A $B; do echo "$(basename $L): $(BF_ACCEL_LIB=$L python $R/scripts/bits_check.py | tail -1)"; done
for i in $(seq $N); do
  for L in $A $B; do
    for G in "260 346" "480 640" "720 1280"; do
      echo "$(basename $L) $(BF_ACCEL_LIB=$L python $R/scripts/kernel_time.py $G iters=200 | tail -1)"
      echo "$(basename $L) co $(BF_ACCEL_LIB=$L python $R/scripts/kernel_time.py $G iters=200 co_schedule=1 | tail -1)"
    done
  done
done
