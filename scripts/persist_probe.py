"""Cold config-2 runs with the single-launch loop on / off: time, iterations, bit-identity."""
import sys
import os
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from better_flow_amd import accel, synth
N, H, W, s = 1000000, 260, 346, 3
sl = synth.make_slice(N, H, W, 0.030, seed=1)
res = {}
for name, kv in (("persist", {}), ("multi", {"persist": 0})) + tuple(
        (a, dict(x.split("=") for x in a.split(","))) for a in sys.argv[1:]):
    acc = accel.Accel(max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
    for k, v in kv.items():
        acc.set_option(k, int(v))
    opts = acc.default_opts()
    opts.res_x, opts.res_y, opts.want_uv = H, W, 1
    best = 1e9
    for r in range(4):
        acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
        acc.set_cloud(s, H, W)
        acc.synchronize()
        t0 = time.perf_counter()
        rc, m, info = acc.run(opts)
        best = min(best, time.perf_counter() - t0)
    res[name] = m.as_dict()
    print(name, "rc", rc, "iters", info.iterations, "rebins", info.rebins, "ovf", info.overflow_events,
          "launches", info.launches, "polls", info.polls, "best ms %.3f" % (best * 1e3),
          "us/iter %.2f" % (best * 1e6 / max(info.iterations, 1)))
    acc.close()
print("bit-identical:", res["persist"] == res["multi"])
