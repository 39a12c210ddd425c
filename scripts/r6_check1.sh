#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_check1; mkdir -p $O; cd $R
(time timeout 2400 python -m pytest tests -m gpu -x -q) > $O/gputest.log 2>&1; tail -15 $O/gputest.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/solo_tail -o s -- python $R/scripts/run_once.py 3 co_schedule=1 > $O/solo_tail.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/solo_head -o s -- python $R/scripts/run_once.py 3 > $O/solo_head.log 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
cd $R; timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json; tail -5 $O/bench.err
