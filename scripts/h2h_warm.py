"""Warm-started chains, host to host (bench.py's regimes.host_to_host.warm_stm), with the knobs of round 6's front-end work:
    h2h_warm.py LANES [ahead=1|2] [defer=0|1] [bytes=8|12] [reps=N]
ahead: uploads kept in flight ahead of the slice being solved; defer: "defer_uploads"."""
import sys, os, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from better_flow_amd import accel, synth
B = int(sys.argv[1])
kv = dict(a.split("=") for a in sys.argv[2:])
AHEAD, DEFER, B8, NREP = int(kv.get("ahead", 2)), int(kv.get("defer", 1)), int(kv.get("bytes", 8)) == 8, int(kv.get("reps", 40))
H, W, s = 260, 346, 3
slices = [synth.make_slice(1000000, H, W, 0.030, seed=1 + i) for i in range(6)]
nmax = max(len(sl["t"]) for sl in slices)
accs = [accel.Accel(max_events=nmax, max_rows=s * H + s, max_cols=s * W + s) for _ in range(B)]
pinned = []
for sl in slices:
    n = len(sl["t"])
    trip = [accs[0].pinned_array(n, np.uint16), accs[0].pinned_array(n, np.uint16), accs[0].pinned_int32(n)] if B8 else [accs[0].pinned_int32(n) for _ in range(3)]
    trip[0][:], trip[1][:], trip[2][:] = sl["fr_x"], sl["fr_y"], sl["t"]
    pinned.append((trip, n))
for a in accs:
    a.set_option("co_schedule", 1 if B > 1 else 0)
    a.set_option("defer_uploads", DEFER)
res = []
for rep in range(3):
    tot = [[0, 0] for _ in range(B)]
    def lane_loop(lane):
        a = accs[lane]
        o = a.default_opts(); o.res_x, o.res_y, o.want_uv = H, W, 0
        def put(k):
            trip, n_ = pinned[(k + lane) % len(pinned)]
            a.upload_events_async(trip[0], trip[1], trip[2], n_)
        for j in range(AHEAD):
            put(j)
        prev = None
        for k in range(NREP):
            a.commit_upload()
            if k + AHEAD < NREP:
                put(k + AHEAD)
            a.set_cloud(s, H, W)
            if prev is not None:
                a.set_model(prev)
            rc, prev, info = a.run(o)
            if k > 0:
                tot[lane][0] += a.n; tot[lane][1] += info.iterations
            else:
                a.synchronize(); t_start[lane] = time.perf_counter()
        a.synchronize()
        t_end[lane] = time.perf_counter()
    t_start, t_end = [0.0] * B, [0.0] * B
    th = [threading.Thread(target=lane_loop, args=(l,)) for l in range(B)]
    for t in th: t.start()
    for t in th: t.join()
    dt = max(t_end) - min(t_start)
    res.append(sum(x[0] for x in tot) / dt / 1e9)
print("%d lane(s) ahead=%d defer=%d %d B/event: %s Gevents/s (%.1f us per slice per chain, %.1f iterations)" %
      (B, AHEAD, DEFER, 8 if B8 else 12, " ".join("%.2f" % r for r in res), 1e6 * dt / (NREP - 1), sum(x[1] for x in tot) / (B * (NREP - 1))))
