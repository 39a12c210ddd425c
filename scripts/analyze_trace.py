"""Summarise a rocprofv3 kernel_trace.csv: per-kernel durations, gaps, iteration period."""
import csv, collections, glob, statistics, sys
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
NAMES = ["k_bin_warp_scatter<true>", "k_bin_warp_scatter<false>", "k_stencil_binned", "k_stencil<", "k_update",
         "k_bin_scatter", "k_bin_count", "k_bin_scan", "k_warp_scatter", "k_prepare", "k_compute_uv",
         "k_set_state", "k_init_stats", "copyBuffer", "fillBuffer", "k_iter_head", "k_iter_tail"]
import re
def short(n):
    m = re.search(r"(k_[a-z_0-9]+)(<[^>(]*>)?", n)
    if m:
        return m.group(1) + (m.group(2) or "")
    for k in ("copyBuffer", "fillBuffer"):
        if k in n:
            return k
    return n[:40]
seq = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
dur = collections.defaultdict(list)
for n, s, e in seq:
    dur[n].append((e - s) / 1e3)
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    live = [x for x in v if x > 2.0] or v
    print("%-28s calls %5d  total %9.1f us  median(live) %7.2f  mean(live) %7.2f  max %7.1f" %
          (n, len(v), sum(v), statistics.median(live), sum(live) / len(live), max(v)))
gaps = collections.defaultdict(list)
for (n0, s0, e0), (n1, s1, e1) in zip(seq, seq[1:]):
    gaps[(n0, n1)].append((s1 - e0) / 1e3)
for k, v in gaps.items():
    if len(v) > 20 and "-v" in sys.argv:
        print("gap %-52s median %6.2f mean %6.2f max %7.1f n=%d" % (k, statistics.median(v), sum(v) / len(v), max(v), len(v)))
first = [n for n in dur if n.startswith("k_bin_warp_scatter<true") or n.startswith("k_iter_head")]
first = first[0] if first else [n for n in dur if n.startswith("k_warp_scatter")][0]
starts = [s for n, s, e in seq if n == first]
per = [(b - a) / 1e3 for a, b in zip(starts, starts[1:]) if (b - a) < 5e5]
print("iteration period: median %.1f us, mean %.1f us (n=%d)" % (statistics.median(per), sum(per) / len(per), len(per)))
