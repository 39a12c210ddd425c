"""A few un-instrumented cold runs (for rocprofv3 kernel traces)."""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from better_flow_amd import accel, synth
N, H, W, s = int(os.environ.get("BF_RUN_N", "1000000")), int(os.environ.get("BF_RUN_H", "260")), int(os.environ.get("BF_RUN_W", "346")), 3
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sl = synth.make_slice(N, H, W, 0.030, seed=1)
acc = accel.Accel(max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
for k, v in [a.split("=") for a in sys.argv[2:]]:
    acc.set_option(k, int(v))
opts = acc.default_opts()
opts.res_x, opts.res_y, opts.want_uv = H, W, 1
opts.max_iter = int(os.environ.get("BF_RUN_MAXITER", "-1"))   # (BF_RUN_H / BF_RUN_W / BF_RUN_MAXITER: other geometries)
import time
for r in range(reps):
    acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
    acc.set_cloud(s, H, W)
    acc.synchronize()
    t0 = time.perf_counter()
    rc, m, info = acc.run(opts)
    acc.synchronize()
    dt = time.perf_counter() - t0
    print("iters", info.iterations, "rebins", info.rebins, "ovf", info.overflow_events,
          "%.3f ms = %.2f us / iteration" % (1e3 * dt, 1e6 * dt / max(1, info.iterations)))
