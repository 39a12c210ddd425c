#!/bin/bash
# Round-6 rocprofv3 passes (run on the GPU box via gpurun; outputs under gpurun_out/prof_r6, then
# `python scripts/collect_r6.py` here copies the summaries into profiles/r6_*):
#   solo_head / solo_tail : --kernel-trace --stats over ONE slice context (update at the scatter head, 1024-thread scatter
#                           work-groups / lean 512-thread scatter kernel + update in the stencil tail: the headline regime's variants)
#   bench                 : --kernel-trace --stats over the default bench.py command (4 contexts in flight)
#   chip_full             : --kernel-trace --stats over scripts/batch_proxy.py (eight slices side by side: the kernels with the chip full)
#   ring                  : --kernel-trace --stats over the reference's compiled-in ring through the command line (persistent loop kernel)
#   FETCH_* / WRITE_*     : --pmc FETCH_SIZE / WRITE_SIZE, separate passes: config 2 in the two variants bench.py's roofline names
#                           (lean 512-thread scatter + tail update; 1024-thread head update), 640x480 and 1280x720 co-scheduled shape
#   sq_346 / sq_720       : SQ issue / wait counters
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_r6
rm -rf $O; mkdir -p $O
KS="--kernel-trace --stats --output-format csv"
timeout 300 rocprofv3 $KS -d $O/solo_head -o s -- python $R/scripts/run_once.py 3 > $O/solo_head.log 2>&1
timeout 300 rocprofv3 $KS -d $O/solo_tail -o s -- python $R/scripts/run_once.py 3 co_schedule=1 > $O/solo_tail.log 2>&1
timeout 900 rocprofv3 $KS -d $O/bench -o b -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-front-end > $O/bench.log 2>&1
timeout 600 rocprofv3 $KS -d $O/chip_full -o s -- python $R/scripts/batch_proxy.py > $O/chip_full.log 2>&1
timeout 600 rocprofv3 $KS -d $O/ring -o s -- python $R/scripts/default_ring_bench.py --events 2500000 > $O/ring.log 2>&1
for V in "lean512 co_schedule=1" "head1024 co_schedule=0"; do
  set -- $V; T=$1; shift
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/${C}_346x260_$T -o p --output-format csv -- python $R/scripts/run_once.py 1 "$@" > $O/${C}_346x260_$T.log 2>&1
  done
done
for G in "480 640 300" "720 1280 300"; do
  set -- $G
  for C in FETCH_SIZE WRITE_SIZE; do
    BF_RUN_H=$1 BF_RUN_W=$2 BF_RUN_MAXITER=$3 timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/${C}_$2x$1 -o p --output-format csv -- python $R/scripts/run_once.py 1 co_schedule=1 > $O/${C}_$2x$1.log 2>&1
  done
done
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES"
BF_RUN_H=720 BF_RUN_W=1280 BF_RUN_MAXITER=300 timeout 300 rocprofv3 --kernel-trace --pmc $SQ -d $O/sq_720 -o sq --output-format csv -- python $R/scripts/run_once.py 1 co_schedule=1 > $O/sq_720.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $SQ -d $O/sq_346 -o sq --output-format csv -- python $R/scripts/run_once.py 1 co_schedule=1 > $O/sq_346.log 2>&1
grep '^{"metric"' $O/bench.log | cut -c1-200
find $O -name "*.db" -delete < /dev/null
find $O -name "*kernel_trace.csv" -size +8M -delete < /dev/null
find $O -name "*counter_collection.csv" -size +20M -exec sh -c 'head -200000 "$1" > "$1.head" && rm "$1"' _ {} \;
du -sh $O < /dev/null
