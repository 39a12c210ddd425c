"""Timing probe: one cold + one warm fused run on a synthetic slice (GPU box)."""
import sys, time, json
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from better_flow_amd import accel, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
H = int(sys.argv[2]) if len(sys.argv) > 2 else 260
W = int(sys.argv[3]) if len(sys.argv) > 3 else 346
s = int(sys.argv[4]) if len(sys.argv) > 4 else 3
poll = int(sys.argv[5]) if len(sys.argv) > 5 else 8
binned = int(sys.argv[6]) if len(sys.argv) > 6 else 1
tile = int(sys.argv[7]) if len(sys.argv) > 7 else 64
margin = int(sys.argv[8]) if len(sys.argv) > 8 else 8
sl = synth.make_slice(N, H, W, 0.030, seed=1)
sl2 = synth.make_slice(N, H, W, 0.030, seed=2)
n = len(sl["t"])
acc = accel.Accel(max_events=max(n, len(sl2["t"])), max_rows=s * H + s, max_cols=s * W + s)
acc.set_option("binned", binned); acc.set_option("bin_tile", tile); acc.set_option("bin_margin", margin)
print("version", acc.L.bf_version().decode(), "events", n, "binned", binned, "tile", tile, "margin", margin)
print("copy GB/s", acc.copy_bandwidth(1 << 30, 5))
opts = acc.default_opts()
opts.res_x, opts.res_y, opts.poll_interval, opts.want_uv = H, W, poll, 1
for rep in range(3):
    acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
    acc.synchronize()
    t0 = time.perf_counter()
    acc.set_cloud(s, H, W)
    rc, m, info = acc.run(opts)
    acc.synchronize()
    dt = time.perf_counter() - t0
    print("cold rep", rep, "rc", rc, "iters", info.iterations, "ms", dt * 1e3, "us/iter", dt * 1e6 / info.iterations,
          "Mev/s", n / dt / 1e6, "polls", info.polls, "rebins", info.rebins, "ovf", info.overflow_events)
    # warm: slice 2 from slice 1's model
    acc.upload_events(sl2["fr_x"], sl2["fr_y"], sl2["t"])
    acc.synchronize()
    t0 = time.perf_counter()
    acc.set_cloud(s, H, W)
    acc.set_model(m)
    rc, m2, info2 = acc.run(opts)
    acc.synchronize()
    dt = time.perf_counter() - t0
    print("warm rep", rep, "rc", rc, "iters", info2.iterations, "ms", dt * 1e3, "Mev/s", len(sl2["t"]) / dt / 1e6, "rebins", info2.rebins, "ovf", info2.overflow_events)
u, v = acc.compute_uv()
print("flow", u.mean(), v.mean(), "truth", sl["velocity"])
print("model", m.as_dict())
acc.profile_enable(1)
acc.profile_reset()
acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
acc.set_cloud(s, H, W)
rc, m, info = acc.run(opts)
p = acc.profile_get()
print("profiled: iters", info.iterations)
print(" K1 warp_scatter us/launch", 1e3 * p.warp_scatter_ms / max(1, p.warp_scatter_launches), "launches", p.warp_scatter_launches)
print(" K3 stencil      us/launch", 1e3 * p.stencil_ms / max(1, p.stencil_launches), "launches", p.stencil_launches)
print(" K4 update       us/launch", 1e3 * p.update_ms / max(1, p.update_launches), "launches", p.update_launches)
print(" other ms", p.other_ms, p.other_launches)
k1 = p.warp_scatter_ms / max(1, p.warp_scatter_launches) * 1e-3
print(" K1 algorithmic GB/s (28 B/event-iter)", 28.0 * n / k1 / 1e9)
