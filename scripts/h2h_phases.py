"""Where a host-to-host cold slice's time goes with several contexts in flight: wall time of every C-ABI call per lane,
resident inputs (bf_upload_events_device) against pinned host inputs (bf_upload_events_async + bf_commit_upload).
usage: h2h_phases.py [lanes] [slices per lane]"""
import sys, os, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from better_flow_amd import accel, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
NREP = int(sys.argv[2]) if len(sys.argv) > 2 else 8
H, W, s = 260, 346, 3
slices = [synth.make_slice(1000000, H, W, 0.030, seed=1 + i) for i in range(6)]
nmax = max(len(sl["t"]) for sl in slices)
accs = [accel.Accel(max_events=nmax, max_rows=s * H + s, max_cols=s * W + s) for _ in range(B)]
for a in accs:
    a.set_option("co_schedule", 1 if B > 1 else 0)
resident = [(accs[0].to_device(sl["fr_x"]), accs[0].to_device(sl["fr_y"]), accs[0].to_device(sl["t"].astype(np.int32)), len(sl["t"])) for sl in slices]
pinned = []
for sl in slices:
    trip = []
    for key in ("fr_x", "fr_y", "t"):
        p = accs[0].pinned_int32(len(sl["t"]))
        p[:] = sl[key]
        trip.append(p)
    pinned.append((trip, len(sl["t"])))

def run(mode):
    ph = [dict() for _ in range(B)]
    def add(lane, k, dt):
        ph[lane][k] = ph[lane].get(k, 0.0) + dt
    def lane_loop(lane):
        a = accs[lane]
        o = a.default_opts(); o.res_x, o.res_y, o.want_uv = H, W, 0
        def timed(name, fn, *args):
            t0 = time.perf_counter(); r = fn(*args); add(lane, name, time.perf_counter() - t0); return r
        if mode == "h2h":
            trip, n_ = pinned[lane % len(pinned)]
            timed("put", a.upload_events_async, trip[0], trip[1], trip[2], n_)
        for k in range(NREP):
            if mode == "h2h":
                timed("commit", a.commit_upload)
                if k + 1 < NREP:
                    trip, n_ = pinned[(1 + k + lane) % len(pinned)]
                    timed("put", a.upload_events_async, trip[0], trip[1], trip[2], n_)
            else:
                dx, dy, dt, n = resident[(k + lane) % len(resident)]
                timed("upload_device", a.upload_events_device, dx, dy, dt, n)
            timed("set_cloud", a.set_cloud, s, H, W)
            rc, m, info = timed("run", a.run, o)
            add(lane, "iters", info.iterations)
    th = [threading.Thread(target=lane_loop, args=(l,)) for l in range(B)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    for a in accs: a.synchronize()
    dt = time.perf_counter() - t0
    keys = sorted(set(k for p in ph for k in p))
    print("%-8s %d lanes x %d slices: %.1f ms per step, %.1f Mev/s;  per slice and lane (ms): " % (mode, B, NREP, 1e3 * dt / NREP, B * NREP * 1.0 / dt) +
          "  ".join("%s %.3f" % (k, 1e3 * sum(p.get(k, 0) for p in ph) / (B * NREP)) for k in keys if k != "iters") +
          "  iterations %.0f" % (sum(p.get("iters", 0) for p in ph) / (B * NREP)))
for mode in ("resident", "h2h", "resident", "h2h"):
    run(mode)
