"""Per-work-group entry / events-done / end stamps of one K1 launch (needs `make tl`)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BF_TIMELINE"] = "/tmp/bf_tl.txt"
os.environ["BF_ACCEL_LIB"] = os.path.join(ROOT, "better_flow_amd", "libbf_accel_tl.so")
from better_flow_amd import accel, synth
N, H, W, s = 1000000, 260, 346, 3
sl = synth.make_slice(N, H, W, 0.030, seed=1)
acc = accel.Accel(max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
for k, v in [a.split("=") for a in sys.argv[1:]]:
    acc.set_option(k, int(v))
opts = acc.default_opts(); opts.res_x, opts.res_y = H, W
opts.max_iter = 40
acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"]); acc.set_cloud(s, H, W)
acc.run(opts)
acc.close()
g = {}
for ln in open("/tmp/bf_tl.txt"):
    kern, L, grp, slot, t = [int(x) for x in ln.split()]
    if kern == 2:
        off = (L * 2 + grp) * 16 + slot
        g.setdefault(off // 4, {})[off % 4] = t
t0 = min(v[0] for v in g.values())
rows = sorted((v[0] - t0, v.get(1, 0) - t0, v.get(2, 0) - t0, v.get(3, 0), b) for b, v in g.items())
print("groups", len(rows))
print("entry   us: min %.2f max %.2f" % (rows[0][0] / 100, rows[-1][0] / 100))
ends = sorted(r[2] for r in rows)
print("end     us: min %.2f median %.2f p90 %.2f max %.2f" % (ends[0] / 100, ends[len(ends) // 2] / 100, ends[int(len(ends) * .9)] / 100, ends[-1] / 100))
for r in sorted(rows, key=lambda r: -r[2])[:8]:
    print("  slow: wg %3d events %5d entry %.2f evdone %.2f end %.2f" % (r[4], r[3], r[0] / 100, r[1] / 100, r[2] / 100))
for r in sorted(rows, key=lambda r: r[2])[:4]:
    print("  fast: wg %3d events %5d entry %.2f evdone %.2f end %.2f" % (r[4], r[3], r[0] / 100, r[1] / 100, r[2] / 100))
