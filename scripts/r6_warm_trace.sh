#!/bin/bash
# Kernel + copy timeline of steady warm slices of ONE chain, host to host: 346x260 (8 B / event) and 640x480 (config 3, 12 B / event)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_warm; mkdir -p $O
for G in "260 346 bytes=8 defer_uploads=1 ahead=2"; do
  set -- $G; T=$2x$1; X="$3 $4 $5"
  for i in 1 2; do python $R/scripts/warm_chain_trace.py $1 $2 24 $X; done > $O/host_$T.txt 2>&1
  rm -rf /tmp/wt; timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/wt -o w --output-format csv -- python $R/scripts/warm_chain_trace.py $1 $2 24 $X > /tmp/wt.log 2>&1
  python - > $O/timeline_$T.txt <<'PY'
import csv, glob
f = glob.glob("/tmp/wt/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void bf::","").replace("bf::","")[:48]) for r in csv.DictReader(open(f))]
m = glob.glob("/tmp/wt/**/*memory_copy_trace.csv", recursive=True)
if m:
    for r in csv.DictReader(open(m[0])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[:24] + " " + r.get("Bytes", "")))
rows.sort()
idx = [i for i, r in enumerate(rows) if r[2].startswith("k_run_init")]
for which in (-5, -4):
    i0 = idx[which]; t0 = rows[i0][0]
    print("--- slice starting at k_run_init #%d" % (len(idx) + which))
    busy = 0
    for s, e, n in rows[i0:i0 + 80]:
        if n.startswith("k_run_init") and s != t0:
            print("next k_run_init at %.1f us; kernels busy %.1f us" % ((s - t0) / 1e3, busy / 1e3)); break
        if not n.startswith("COPY"): busy += e - s
        print("%8.1f %8.1f  %6.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
PY
  cat $O/host_$T.txt; cat $O/timeline_$T.txt
done
