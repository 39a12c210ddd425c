#!/bin/bash
# SQ counters per kernel of one cold config-2 run (own pass, kernel trace only).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES -d $OUT -o sq --output-format csv -- python $GRAFT_REPO_ROOT/scripts/run_once.py 1 "$@" > $OUT/log.txt 2>&1
tail -3 $OUT/log.txt
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
print(f)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); cnt[k] += 1
for k in sorted(agg, key=lambda k: -agg[k]["SQ_BUSY_CYCLES"])[:8]:
    a = agg[k]; n = cnt[k]
    print(k, "n=%d" % n, " ".join("%s=%.3g" % (c.replace("SQ_", ""), a[c] / n) for c in sorted(a)))
PY
