"""Per-iteration cost of the loop over sensor sizes / scales / slice sizes (cold start, capped at 40 iterations):
a quick way to spot a configuration that falls off the fast path."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from better_flow_amd import accel, synth
cases = [(1000000, 260, 346, 3), (1000000, 260, 346, 1), (1000000, 260, 346, 5), (1000000, 260, 346, 7), (1000000, 260, 346, 9),
         (1000000, 480, 640, 3), (1000000, 180, 240, 3), (1000000, 720, 1280, 3), (1000000, 720, 1280, 1),
         (200000, 260, 346, 3), (50000, 180, 240, 3), (20000, 180, 240, 3), (3000000, 480, 640, 3), (1000000, 128, 128, 3)]
print("%9s %10s %2s | %9s %6s %5s %6s %8s | %s" % ("events", "sensor", "s", "us/iter", "iters", "rebin", "ovf", "image", "path"))
for n, H, W, s in cases:
    sl = synth.make_slice(n, H, W, 0.030, seed=1)
    a = accel.Accel(max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
    a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
    w = a.set_cloud(s, H, W)
    o = a.default_opts(); o.res_x, o.res_y, o.max_iter = H, W, 40
    a.run(o)                      # warm-up (allocations)
    a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"]); a.set_cloud(s, H, W); a.synchronize()
    t0 = time.perf_counter(); rc, m, info = a.run(o); dt = time.perf_counter() - t0
    dense = w.scale_img_x * w.scale_img_y < 12 * len(sl["t"])
    print("%9d %5dx%-4d %2d | %9.1f %6d %5d %6d %4dx%-4d | %s" % (len(sl["t"]), W, H, s, dt * 1e6 / max(info.iterations, 1), info.iterations,
          info.rebins, info.overflow_events, w.scale_img_x, w.scale_img_y, "binned" if info.rebins else "atomics"))
    a.close()
