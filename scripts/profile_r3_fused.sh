#!/bin/bash
# Round-3 rocprofv3 passes of the one-kernel iteration (run on the GPU box via gpurun; summaries are copied to
# gpurun_out/prof_r3_fused/*.csv|txt, from where the judged ones go to profiles/r3_fused_*):
#   small      : --kernel-trace --stats, cold 50 000-event slices at 346x260 (auto takes the one-kernel loop)
#   small_two  : the same slices with fused=0 (two-kernel loop), for the per-kernel comparison
#   ring       : --kernel-trace --stats over the command line on the reference's compiled-in ring (5M events, 240x180)
#   fetch/write: --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on the small slices
#   sq         : SQ issue / wait counters of the pass
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_r3_fused
rm -rf $O; mkdir -p $O
export BF_RUN_N=50000
timeout 300 rocprofv3 --kernel-trace --stats -d $O/small -o s --output-format csv -- python $R/scripts/run_once.py 5 > $O/small.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/small_two -o s --output-format csv -- python $R/scripts/run_once.py 5 fused=0 binned=2 > $O/small_two.log 2>&1
python - <<PY
import sys
sys.path.insert(0, "$R")
from better_flow_amd import synth
synth.write_stream_bin("/tmp/ring5m.bin", 20, 250000, 180, 240, duration_s=0.033)
PY
timeout 300 rocprofv3 --kernel-trace --stats -d $O/ring -o s --output-format csv -- $R/better_flow_amd/host/bf_motion_compensator --quiet --timing --res-x=180 --res-y=240 /tmp/ring5m.bin > $O/ring.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/$C -o p --output-format csv -- python $R/scripts/run_once.py 1 > $O/$C.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES -d $O/sq -o sq --output-format csv -- python $R/scripts/run_once.py 1 > $O/sq.log 2>&1
python - <<PY
import csv, glob, collections, statistics, re, os
O = "$O"
def kname(n):
    m = re.search(r"(k_[a-z_0-9]+)(<[^>(]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:40]
with open(os.path.join(O, "pmc.txt"), "w") as out:
    out.write("rocprofv3 --pmc (separate passes, --kernel-trace only) over one cold 50 000-event 346x260 slice on the one-kernel loop.\n"
              "FETCH_SIZE / WRITE_SIZE in KB per dispatch ('live': dispatches above 64 KB; on gfx950 FETCH_SIZE under-reports wide\n"
              "coalesced reads by 2x -- MI355X_MICROARCH.md, HBM section); SQ counters summed over the dispatches of the kernel.\n\n")
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob(os.path.join(O, cname, "**/*counter_collection.csv"), recursive=True)
        if not fs: continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if r["Counter_Name"] == cname: acc[kname(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            live = [x for x in v if x > 64] or [0.0]
            out.write("%-11s %-40s dispatches %5d  live %5d  median %10.1f KB  mean %10.1f KB\n" % (cname, k, len(v), len(live), statistics.median(live), sum(live) / len(live)))
    fs = glob.glob(os.path.join(O, "sq", "**/*counter_collection.csv"), recursive=True)
    if fs:
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(fs[0])):
            acc[kname(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, d in sorted(acc.items()):
            if "fused" not in k and "stencil" not in k and "warp_scatter" not in k: continue
            out.write("\n%s\n" % k)
            for c, v in sorted(d.items()): out.write("   %-22s %16.0f\n" % (c, v))
            if d.get("SQ_WAVE_CYCLES"):
                out.write("   waves wait %.0f %% of their cycles (SQ_WAIT_ANY / SQ_WAVE_CYCLES), issue VALU in %.0f %%\n" %
                          (100 * d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 100 * d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"]))
PY
for d in small small_two ring; do cp $(find $O/$d -name "*kernel_stats.csv" | head -1) $O/${d}_kernel_stats.csv; done
tail -2 $O/small.log; tail -2 $O/small_two.log; grep mevents $O/ring.log | cut -c1-260
find $O -name "*.db" -delete < /dev/null
find $O -name "*kernel_trace.csv" -delete < /dev/null
find $O -name "*counter_collection.csv" -delete < /dev/null
cat $O/pmc.txt | head -40
