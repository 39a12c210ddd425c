"""In-kernel phase stamps of the persistent loop kernel (k_fused_loop; needs `make -C better_flow_amd/csrc tl`): work-groups 0
and nb/2, passes 8 .. 63 of a cold run, 100 MHz ticks -> us since the pass's start, and the period between passes.
BF_RUN_N / BF_RUN_H / BF_RUN_W / BF_RUN_S select the slice."""
import sys, os, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BF_TIMELINE"] = "/tmp/bf_tl.txt"
os.environ["BF_ACCEL_LIB"] = os.environ.get("BF_TL_LIB", os.path.join(ROOT, "better_flow_amd", "libbf_accel_tl.so"))
from better_flow_amd import accel, synth
N, H, W = int(os.environ.get("BF_RUN_N", "50000")), int(os.environ.get("BF_RUN_H", "180")), int(os.environ.get("BF_RUN_W", "240"))
s = int(os.environ.get("BF_RUN_S", "3"))
sl = synth.make_slice(N, H, W, 0.030, seed=5)
acc = accel.Accel(max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
acc.set_option("fused", 2)
for k, v in [a.split("=") for a in sys.argv[1:]]:
    acc.set_option(k, int(v))
opts = acc.default_opts(); opts.res_x, opts.res_y = H, W
opts.max_iter = int(os.environ.get("BF_RUN_MAXITER", "200"))
acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"]); acc.set_cloud(s, H, W)
rc, m, info = acc.run(opts)
print("iterations", info.iterations, "launches", info.launches, "re-bins", info.rebins)
acc.close()
NAMES = ["pass start", "tile zeroed + barrier", "events done", "barrier", "time image done", "barrier", "Scharr + sums done", "partials published",
         "record stored", "reduced records in", "update done", "end barrier"]
tl = {}
for ln in open("/tmp/bf_tl.txt"):
    kern, L, grp, slot, t = [int(x) for x in ln.split()]
    if kern == 0:
        tl.setdefault((L, grp), {})[slot] = t
for grp in (0, 1):
    print("work-group", "0" if grp == 0 else "nb/2")
    for slot in range(1, 12):
        v = [(tl[(L, grp)][slot] - tl[(L, grp)][0]) / 100 for L in range(8, 64) if (L, grp) in tl and slot in tl[(L, grp)] and 11 in tl[(L, grp)]]
        if v:
            print("   %-24s median %6.2f us  (min %5.2f max %5.2f, %d passes)" % (NAMES[slot], statistics.median(v), min(v), max(v), len(v)))
    per = [(tl[(L + 1, grp)][0] - tl[(L, grp)][0]) / 100 for L in range(8, 62) if (L, grp) in tl and (L + 1, grp) in tl]
    if per:
        print("   pass start -> next pass start: median %.2f us (min %.2f)" % (statistics.median(per), min(per)))
