"""Config 3's warm-start chain, slice by slice (inputs resident): wall time, iterations, re-bins, overflow events, launches and
polls of every warm slice -- what a 640x480 stream's 0.55 ms per slice is made of.  usage: warm_chain_640.py [key=value ...]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from better_flow_amd import accel, synth
OPTS = [a.split('=') for a in sys.argv[1:] if '=' in a]
N, H, W, s = 1000000, 480, 640, 3
NS = 10
slices = [synth.make_slice(N, H, W, 0.030, seed=100 + i) for i in range(NS)]
nmax = max(len(sl["t"]) for sl in slices)
acc = accel.Accel(max_events=nmax, max_rows=s * H + s, max_cols=s * W + s)
for k_, v_ in OPTS: acc.set_option(k_, int(v_))
opts = acc.default_opts(); opts.res_x, opts.res_y, opts.want_uv = H, W, 1
res = [(acc.to_device(sl["fr_x"]), acc.to_device(sl["fr_y"]), acc.to_device(sl["t"].astype(np.int32)), len(sl["t"])) for sl in slices]
for rep in range(2):
    prev = None; rows = []
    for i in range(NS):
        dx, dy, dt, n = res[i]
        acc.synchronize(); t0 = time.perf_counter()
        acc.upload_events_device(dx, dy, dt, n); acc.set_cloud(s, H, W)
        if prev is not None: acc.set_model(prev)
        rc, prev, info = acc.run(opts); acc.synchronize()
        rows.append((1e3 * (time.perf_counter() - t0), info.iterations, info.rebins, info.overflow_events, info.launches, info.polls))
print(OPTS, "warm mean %.3f ms" % (sum(r[0] for r in rows[1:]) / (NS - 1)))
for r in rows[1:]: print("   %.3f ms  iters %3d rebins %d ovf %6d launches %3d polls %d" % r)
