"""Where a warm (STM) slice spends its time: wall time per C-ABI call of one sequential chain
(config-2 slices resident in HBM), each call followed by a synchronize so that it can be attributed."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from better_flow_amd import accel, synth
N, H, W, s = 1000000, 260, 346, 3
K = 8
acc = accel.Accel(max_events=N, max_rows=s * H + s, max_cols=s * W + s)
for k, v in [a.split("=") for a in sys.argv[1:]]:
    acc.set_option(k, int(v))
res = []
for i in range(K):
    sl = synth.make_slice(N, H, W, 0.030, seed=1 + i)
    res.append((acc.to_device(sl["fr_x"]), acc.to_device(sl["fr_y"]), acc.to_device(sl["t"].astype(np.int32)), len(sl["t"])))
opts = acc.default_opts(); opts.res_x, opts.res_y, opts.want_uv = H, W, 1
tot = {}
def timed(name, fn, sync=True):
    t0 = time.perf_counter(); r = fn()
    if sync: acc.synchronize()
    tot.setdefault(name, []).append(time.perf_counter() - t0); return r
model = None
for rep in range(3):
    for i in range(K):
        dx, dy, dt, n = res[i]
        timed("upload_device", lambda: acc.upload_events_device(dx, dy, dt, n))
        timed("set_cloud", lambda: acc.set_cloud(s, H, W))
        if model is not None:
            timed("set_model", lambda: acc.set_model(model))
        rc, m, info = timed("run", lambda: acc.run(opts))
        tot.setdefault("iters", []).append(info.iterations); tot.setdefault("launches", []).append(info.launches)
        tot.setdefault("polls", []).append(info.polls)
        model = m
for k, v in tot.items():
    v = v[K:] if len(v) > K else v    # drop the first pass (cold first slice, allocations)
    if k in ("iters", "launches", "polls"):
        print("%-14s mean %.1f" % (k, sum(v) / len(v)))
    else:
        print("%-14s mean %7.1f us  min %7.1f us" % (k, 1e6 * sum(v) / len(v), 1e6 * min(v)))
# the same chain without the per-call synchronizes
t0 = time.perf_counter()
for rep in range(3):
    for i in range(K):
        dx, dy, dt, n = res[i]
        acc.upload_events_device(dx, dy, dt, n); acc.set_cloud(s, H, W); acc.set_model(model); rc, model, info = acc.run(opts)
acc.synchronize()
dt_ = (time.perf_counter() - t0) / (3 * K)
print("chain: %.1f us per slice = %.2f Gev/s" % (dt_ * 1e6, N / dt_ / 1e9))
