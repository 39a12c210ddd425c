#!/bin/bash
# Round-3 rocprofv3 passes (run on the GPU box via gpurun; outputs under gpurun_out/prof_r3, then
# `python scripts/collect_r3.py` here copies the summaries into profiles/):
#   solo_head / solo_tail / solo_tail_1024 : --kernel-trace --stats over ONE slice context (update at the scatter head / in the
#                           stencil tail with the co-scheduled 512-thread scatter shape / the same with 1024-thread groups)
#   bench                 : --kernel-trace --stats over the default bench.py command (4 contexts in flight)
#   fetch_* / write_*     : --pmc FETCH_SIZE / WRITE_SIZE, separate passes, per geometry (346x260, 640x480, 1280x720)
#   sq_720                : SQ issue / wait counters at 1280x720
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_r3
rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/solo_head -o s --output-format csv -- python $R/scripts/run_once.py 3 > $O/solo_head.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/solo_tail -o s --output-format csv -- python $R/scripts/run_once.py 3 co_schedule=1 > $O/solo_tail.log 2>&1
# the shape bench.py's roofline.frac is measured in: update in the stencil tail, 1024-thread scatter work-groups, alone on the GPU
timeout 300 rocprofv3 --kernel-trace --stats -d $O/solo_tail_1024 -o s --output-format csv -- python $R/scripts/run_once.py 3 co_schedule=1 bin_threads=1024 > $O/solo_tail_1024.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/bench -o b --output-format csv -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-front-end > $O/bench.log 2>&1
for G in "260 346 -1" "480 640 300" "720 1280 300"; do
  set -- $G
  for C in FETCH_SIZE WRITE_SIZE; do
    BF_RUN_H=$1 BF_RUN_W=$2 BF_RUN_MAXITER=$3 timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/${C}_$2x$1 -o p --output-format csv -- python $R/scripts/run_once.py 1 co_schedule=1 > $O/${C}_$2x$1.log 2>&1
  done
done
# 640x480, ONE context with the update at the scatter head: the regime "auto" takes the interior + margin format in (bin_split),
# against the dense slabs in the same regime -- kernel stats and the two traffic counters
for SP in 1 0; do
  BF_RUN_H=480 BF_RUN_W=640 BF_RUN_MAXITER=300 timeout 300 rocprofv3 --kernel-trace --stats -d $O/head640_split$SP -o s --output-format csv -- python $R/scripts/run_once.py 2 bin_split=$SP > $O/head640_split$SP.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    BF_RUN_H=480 BF_RUN_W=640 BF_RUN_MAXITER=300 timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/${C}_head640_split$SP -o p --output-format csv -- python $R/scripts/run_once.py 1 bin_split=$SP > $O/${C}_head640_split$SP.log 2>&1
  done
done
BF_RUN_H=720 BF_RUN_W=1280 BF_RUN_MAXITER=300 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES -d $O/sq_720 -o sq --output-format csv -- python $R/scripts/run_once.py 1 co_schedule=1 > $O/sq_720.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES -d $O/sq_346 -o sq --output-format csv -- python $R/scripts/run_once.py 1 co_schedule=1 > $O/sq_346.log 2>&1
grep '^{"metric"' $O/bench.log | cut -c1-200
find $O -name "*.db" -delete < /dev/null
find $O -name "*kernel_trace.csv" -size +8M -delete < /dev/null
find $O -name "*counter_collection.csv" -size +20M -exec sh -c 'head -200000 "$1" > "$1.head" && rm "$1"' _ {} \;
du -sh $O < /dev/null
