R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_r2
mkdir -p $O
for G in "720 1280 300"; do
  set -- $G
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/${C}_$2x$1
    BF_RUN_H=$1 BF_RUN_W=$2 BF_RUN_MAXITER=$3 timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/${C}_$2x$1 -o p --output-format csv -- python $R/scripts/run_once.py 1 co_schedule=1 > $O/${C}_$2x$1.log 2>&1
  done
done
rm -rf $O/sq_720
BF_RUN_H=720 BF_RUN_W=1280 BF_RUN_MAXITER=300 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES -d $O/sq_720 -o sq --output-format csv -- python $R/scripts/run_once.py 1 co_schedule=1 > $O/sq_720.log 2>&1
find $O -name "*.db" -delete < /dev/null
find $O -name "*kernel_trace.csv" -size +8M -delete < /dev/null
find $O -name "*counter_collection.csv" -size +20M -exec sh -c 'head -200000 "$1" > "$1.head" && rm "$1"' _ {} \;
cd $R; mkdir -p gpurun_out/final
timeout 600 python bench.py --config 5 --farm-slices 16 --no-cpu-baseline > gpurun_out/final/config5.json 2>&1
timeout 600 python scripts/sweep_geometry.py > gpurun_out/final/sweep.txt 2>&1
BF_RUN_H=720 BF_RUN_W=1280 python scripts/run_once.py 2 > gpurun_out/final/one720.txt 2>&1; tail -2 gpurun_out/final/one720.txt
BF_RUN_H=720 BF_RUN_W=1280 BF_RUN_MAXITER=600 bash scripts/iter_trace.sh final720 2>&1 | grep -E "k_stencil|k_bin_warp|period"
BF_RUN_H=720 BF_RUN_W=1280 BF_RUN_MAXITER=600 bash scripts/iter_trace.sh final720co co_schedule=1 2>&1 | grep -E "k_stencil|k_bin_warp|period"
