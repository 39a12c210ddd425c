cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/wt; timeout 200 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/wt -o w --output-format csv -- python $GRAFT_REPO_ROOT/scripts/warm_profile.py > /tmp/wt.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/wt/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void bf::","").replace("bf::","")[:40]) for r in csv.DictReader(open(f))]
m = glob.glob("/tmp/wt/**/*memory_copy_trace.csv", recursive=True)
if m:
    for r in csv.DictReader(open(m[0])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[:20] + " " + r.get("Bytes", "")))
rows.sort()
# find the last k_prepare and print from there for ~one slice
idx = [i for i, r in enumerate(rows) if r[2].startswith("k_prepare")]
i0 = idx[-3]
t0 = rows[i0][0]
for s, e, n in rows[i0:i0 + 40]:
    print("%8.1f %8.1f  %6.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
    if n.startswith("k_prepare") and s != t0: break
PY
