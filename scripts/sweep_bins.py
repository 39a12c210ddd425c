"""Per-iteration time of the loop over tile shapes / work-group sizes of the binned scatter (one context).
usage: sweep_bins.py H W max_iter [opt=val ...]"""
import sys, os, time, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from better_flow_amd import accel, synth
H, W, MI = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
fixed = dict(a.split("=") for a in sys.argv[4:])
s = 3
sl = synth.make_slice(1000000, H, W, 0.030, seed=1)
print("%6s %6s %6s | %9s %6s %6s" % ("thr", "rows", "cols", "us/iter", "iters", "ovf"))
for thr, rows, cols in itertools.product((256, 512, 1024), (0, 32, 48, 64, 96), (0,)):
    a = accel.Accel(max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
    for k, v in fixed.items():
        a.set_option(k, int(v))
    a.set_option("bin_threads", thr); a.set_option("bin_tile_rows", rows); a.set_option("bin_tile", cols)
    o = a.default_opts(); o.res_x, o.res_y, o.max_iter = H, W, MI
    best = 1e9
    for rep in range(2):
        a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"]); a.set_cloud(s, H, W); a.synchronize()
        t0 = time.perf_counter(); rc, m, info = a.run(o); a.synchronize(); dt = time.perf_counter() - t0
        best = min(best, dt)
    print("%6d %6d %6d | %9.2f %6d %6d" % (thr, rows, cols, 1e6 * best / max(info.iterations, 1), info.iterations, info.overflow_events), flush=True)
    a.close()
