#!/bin/bash
# round 6: event lists sorted by (column zone, row) against row only -- libbf_accel_alt.so = the same tree with `c->grid.zw` left 0 in
# bf_set_cloud (bf_operators.cpp; a one-line switch that is not kept in the source): bits, both loop
# kernels per geometry, the 16-slice config-5 batch; then the GPU suite on the new build.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_ab_zones; mkdir -p $O
cd $R
A=$R/better_flow_amd/libbf_accel_alt.so; B=$R/better_flow_amd/libbf_accel.so
for L in $A $B; do echo "$(basename $L): $(BF_ACCEL_LIB=$L python scripts/bits_check.py | tail -1)"; done > $O/bits.txt 2>&1
for i in 1 2; do for L in $A $B; do
  echo "$(basename $L) $(BF_ACCEL_LIB=$L python scripts/kernel_time.py 720 1280 iters=200 | tail -1)"
  echo "$(basename $L) co $(BF_ACCEL_LIB=$L python scripts/kernel_time.py 720 1280 iters=200 co_schedule=1 | tail -1)"
done; done > $O/kernel_time.txt 2>&1
for L in $A $B; do echo "$(basename $L) $(BF_ACCEL_LIB=$L python bench.py --config 5 --farm-slices 16 2>/dev/null | tail -1 | cut -c1-400)"; done > $O/config5_16.txt 2>&1
cat $O/bits.txt $O/kernel_time.txt $O/config5_16.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.txt; cat $O/pytest.txt
