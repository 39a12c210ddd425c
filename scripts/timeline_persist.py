"""Phase timeline of the single-launch loop (needs `make tl`): per iteration, work-groups 0 and mid."""
import sys, os, collections
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
rc, m, info = acc.run(opts)
print("iters", info.iterations, "rebins", info.rebins)
acc.close()
d = collections.defaultdict(dict)
for ln in open("/tmp/bf_tl.txt"):
    kern, L, g, slot, t = [int(x) for x in ln.split()]
    d[(kern, L, g)][slot] = t
names = {0: "start", 1: "scattered", 2: "ring out", 3: "barrier1", 4: "merged", 5: "time img", 6: "stencil+partials",
         7: "barrier2", 8: "gathered", 9: "updated"}
print(len(d), sorted(d)[:6])
for L in (5, 20, 21, 22):
    base = d[(0, L, 0)].get(0)
    for g in (0, 1):
        st = d.get((0, L, g), {})
        if st and base:
            print("iter", L, "wg", "0  " if g == 0 else "mid", " | ".join("%s=%.2f" % (names.get(k, k), (st[k] - base) / 100.0) for k in sorted(st)))
