"""Per-kernel time of the cold loop's first iterations, one context alone (bf_profile: the kernels' own timestamps).
    python scripts/kernel_time.py H W [KEY=VALUE ...]   e.g.  kernel_time.py 480 640 bin_compact=0"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from better_flow_amd import accel, synth
H, W = int(sys.argv[1]), int(sys.argv[2])
opts = dict(kv.split("=") for kv in sys.argv[3:])
s, iters = int(opts.pop("scale", 3)), int(opts.pop("iters", 80))
sl = synth.make_slice(int(opts.pop("events", 1000000)), H, W, 0.030, seed=1)
a = accel.Accel(max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
for k, v in opts.items():
    a.set_option(k, int(v))
o = a.default_opts(); o.res_x, o.res_y, o.max_iter = H, W, iters
for rep in range(2):
    a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"]); a.set_cloud(s, H, W)
    if rep == 1:
        a.profile_enable(1); a.profile_reset()
    rc, m, info = a.run(o)
p = a.profile_get()
n = max(1, info.iterations)
print("%dx%d s=%d %s: K1 %.2f us  K3 %.2f us  sum %.2f us  (%d iterations, %d re-bins, %d overflow events)" %
      (W, H, s, opts, 1e3 * p.warp_scatter_ms / n, 1e3 * p.stencil_ms / n, 1e3 * (p.warp_scatter_ms + p.stencil_ms) / n, info.iterations,
       info.rebins, info.overflow_events))
a.close()
