#!/bin/bash
# One slice context, cold config-2 runs (or BF_RUN_H / BF_RUN_W / BF_RUN_MAXITER): un-profiled time per iteration, then
# a rocprofv3 kernel trace of the same command summarised per kernel and per gap.  Usage: iter_trace.sh [tag] [opt=val ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-it}; shift
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/trace_$TAG
rm -rf $O; mkdir -p $O
python $R/scripts/run_once.py 3 "$@" | tee $O/unprofiled.txt
timeout 300 rocprofv3 --kernel-trace -d $O -o t --output-format csv -- python $R/scripts/run_once.py 2 "$@" > $O/profiled.txt 2>&1
python $R/scripts/analyze_trace.py $O -v | tee $O/summary.txt
find $O -name "*.db" -delete < /dev/null
find $O -name "*kernel_trace.csv" -size +20M -delete < /dev/null
