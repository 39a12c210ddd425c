# Kernel timeline of steady-state warm slices of config 3 (640x480, 1M events, STM chain, pinned + overlapped uploads)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/wt; timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/wt -o w --output-format csv -- python $R/scripts/config3_stream.py > /tmp/wt.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/wt/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void bf::","").replace("bf::","")[:44]) for r in csv.DictReader(open(f))]
m = glob.glob("/tmp/wt/**/*memory_copy_trace.csv", recursive=True)
if m:
    for r in csv.DictReader(open(m[0])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[:24] + " " + r.get("Bytes", "")))
rows.sort()
idx = [i for i, r in enumerate(rows) if r[2].startswith("k_prepare")]
for which in (-4, -3):
    i0 = idx[which]; t0 = rows[i0][0]
    print("--- slice starting at k_prepare #%d" % (len(idx) + which))
    busy = 0
    for s, e, n in rows[i0:i0 + 80]:
        if n.startswith("k_prepare") and s != t0:
            print("next k_prepare at %.1f us; kernels busy %.1f us" % ((s - t0) / 1e3, busy / 1e3)); break
        if not n.startswith("COPY"): busy += e - s
        print("%8.1f %8.1f  %6.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
PY
