#!/bin/bash
# round 5, task 1: stamped prototype of the persistent loop at config 2 (1M events, 346x260, scale 3)
cd "$(dirname "$0")/.."
export BF_RUN_N=1000000 BF_RUN_H=260 BF_RUN_W=346 BF_RUN_S=3
out=gpurun_out/proto_r5.txt
: > $out
echo "== baseline: one context cold, two-kernel loop ==" >> $out
python scripts/run_once.py 3 >> $out 2>&1
# libbf_accel_tl_{u4,u8,u8own}.so: `make -C better_flow_amd/csrc tl` with TLX="", "-DBF_LOOP_U=8", "-DBF_LOOP_U=8 -DBF_PROTO_OWNONLY"
# (profiles/r5_resident_loop_prototype.txt also holds the 64-row tiles, measured while "fused_rows" was still an option)
for lib in u4 u8 u8own; do
  echo "== lib $lib persist=2 ==" >> $out
  BF_TL_LIB=$PWD/better_flow_amd/libbf_accel_tl_$lib.so timeout 300 python scripts/timeline_loop.py persist=2 >> $out 2>&1
done
cat $out
