#!/bin/bash
# All rocprofv3 passes behind profiles/ (run on the GPU box via gpurun; outputs under gpurun_out/):
#   prof_bench : --kernel-trace --stats over the default bench.py command
#   prof_fetch / prof_write : --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel trace only)
#   pmc_sq     : SQ issue / wait counters (own pass)
# Then: python scripts/collect_profiles.py <tag>   (here, after gpurun merged gpurun_out/)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
rm -rf $O/prof_bench $O/prof_fetch $O/prof_write $O/pmc_sq
mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench --output-format csv -- python $R/bench.py --steps 8 --warmup 2 > $O/prof_bench.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof_fetch -o f --output-format csv -- python $R/scripts/run_once.py 1 > $O/prof_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof_write -o w --output-format csv -- python $R/scripts/run_once.py 1 > $O/prof_write.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES -d $O/pmc_sq -o sq --output-format csv -- python $R/scripts/run_once.py 1 > $O/pmc_sq.log 2>&1
grep '^{"metric"' $O/prof_bench.log | cut -c1-300
find $O -name "*.db" -delete < /dev/null
du -sh $O < /dev/null
