#!/bin/bash
# Kernel trace of bench.py with 4 slice contexts in flight: how busy is the GPU, how long are the gaps inside a stream?
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ct; timeout 300 rocprofv3 --kernel-trace -d /tmp/ct -o c --output-format csv -- python $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 "$@" > /tmp/ct.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/ct/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void bf::", "")[:28], r.get("Queue_Id", r.get("Stream_Id", "0"))) for r in csv.DictReader(open(f))]
rows.sort()
# take the middle 40 % of the trace (the timed cold steps)
t0, t1 = rows[0][0], rows[-1][1]
lo, hi = t0 + 0.30 * (t1 - t0), t0 + 0.55 * (t1 - t0)
sel = [r for r in rows if r[0] >= lo and r[1] <= hi]
ev = sorted([(r[0], 1) for r in sel] + [(r[1], -1) for r in sel])
busy = 0; depth = 0; last = None; area = 0
for t, d in ev:
    if depth > 0: busy += t - last; area += depth * (t - last)
    depth += d; last = t
wall = sel[-1][1] - sel[0][0]
print("window %.1f ms: %d kernels, GPU busy %.1f %%, mean kernels in flight while busy %.2f, launches/s %.0f" % (wall / 1e6, len(sel), 100 * busy / wall, area / busy, len(sel) / (wall / 1e9)))
byq = collections.defaultdict(list)
for r in sel: byq[r[3]].append(r)
for q, rs in byq.items():
    gaps = [rs[i + 1][0] - rs[i][1] for i in range(len(rs) - 1)]
    dur = collections.defaultdict(list)
    for r in rs: dur[r[2]].append(r[1] - r[0])
    gaps.sort()
    print("queue", q, "kernels", len(rs), "gap us: median %.1f mean %.1f p90 %.1f" % (gaps[len(gaps) // 2] / 1e3, sum(gaps) / len(gaps) / 1e3, gaps[int(len(gaps) * .9)] / 1e3),
          " ".join("%s %.1f" % (k, sum(v) / len(v) / 1e3) for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:2]))
PY
