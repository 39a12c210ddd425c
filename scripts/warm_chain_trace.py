"""One warm-start chain, host to host (pinned slices -> overlapped upload -> set_cloud -> set_model -> run -> model), the
reference's operating mode (dvs_flow.h:218-224): wall time per slice and per C-ABI call.  Profiled by scripts/r6_warm_trace.sh
for the kernel / copy timeline of one steady slice.   usage: warm_chain_trace.py H W [slices] [key=value ...] [bytes=8|12]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from better_flow_amd import accel, synth
H, W = int(sys.argv[1]), int(sys.argv[2])
NS = int(sys.argv[3]) if len(sys.argv) > 3 and "=" not in sys.argv[3] else 24
OPTS = dict(a.split("=") for a in sys.argv[3:] if "=" in a)
B8 = int(OPTS.pop("bytes", 12)) == 8
_UP = int(OPTS.pop("uploader", 0))
AHEAD = int(OPTS.pop("ahead", 1))   # uploads kept in flight ahead of the slice being solved (the context has two staging slots)
N, s, D = 1000000, 3, 6
slices = [synth.make_slice(N, H, W, 0.030, seed=100 + i) for i in range(D)]
nmax = max(len(sl["t"]) for sl in slices)
acc = accel.Accel(max_events=nmax, max_rows=s * H + s, max_cols=s * W + s)
for k_, v_ in OPTS.items():
    acc.set_option(k_, int(v_))
o = acc.default_opts(); o.res_x, o.res_y, o.want_uv = H, W, 0
pin = []
for sl in slices:
    n = len(sl["t"])
    if B8:
        bx, by, bt = acc.pinned_array(nmax, np.uint16), acc.pinned_array(nmax, np.uint16), acc.pinned_int32(nmax)
    else:
        bx, by, bt = acc.pinned_int32(nmax), acc.pinned_int32(nmax), acc.pinned_int32(nmax)
    bx[:n], by[:n], bt[:n] = sl["fr_x"], sl["fr_y"], sl["t"]
    pin.append(((bx, by, bt), n))
put = lambda i: acc.upload_events_async(*pin[i % D][0], pin[i % D][1])   # (uint16 addresses take bf_upload_events16_async)
# uploader=1: the next slice's upload is issued by a SECOND host thread (the C-ABI's threading contract allows exactly that)
# while this one goes on to set_cloud / run -- the three hipMemcpyAsync calls cost ~15 us of host time during which the
# compute stream would hold nothing but the staging kernels
import threading, queue
UPLOADER = _UP != 0
ph = {}
_q, _done = queue.Queue(), queue.Queue()
def _uploader():
    while True:
        i = _q.get()
        if i is None: return
        put(i); _done.put(i)
if UPLOADER:
    threading.Thread(target=_uploader, daemon=True).start()
def timed(name, fn, *a):
    t0 = time.perf_counter(); r = fn(*a); ph[name] = ph.get(name, 0.0) + time.perf_counter() - t0; return r
for rep in range(2):
    ph.clear()
    prev, its, launches, polls = None, [], 0, 0
    rows = []
    for j in range(AHEAD):
        put(j)
    acc.synchronize()
    t_all = time.perf_counter()
    for i in range(NS):
        if UPLOADER and i > 0:
            timed("wait_uploader", _done.get)
        timed("commit_upload", acc.commit_upload)
        if i + AHEAD < NS:
            if UPLOADER: _q.put(i + AHEAD)
            else: timed("upload_async", put, i + AHEAD)
        timed("set_cloud", acc.set_cloud, s, H, W)
        if prev is not None:
            timed("set_model", acc.set_model, prev)
        rc, prev, info = timed("run", acc.run, o)
        its.append(info.iterations); launches += info.launches; polls += info.polls
        rows.append((info.iterations, info.launches, info.polls, info.rebins))
        if i == 0:
            acc.synchronize(); t_all = time.perf_counter(); ph.clear()   # (slice 0 is the cold one)
    acc.synchronize()
    dt = time.perf_counter() - t_all
n_warm = NS - 1
print("%dx%d %s B/event %s: %.1f us per warm slice = %.2f Gevents/s; iterations %.1f, launches %.1f, polls %.2f per slice" %
      (W, H, 8 if B8 else 12, OPTS, 1e6 * dt / n_warm, np.mean([p[1] for p in pin]) * n_warm / dt / 1e9, np.mean(its[1:]), launches / NS, polls / NS))
print("   host time per slice (us): " + "  ".join("%s %.1f" % (k, 1e6 * v / n_warm) for k, v in ph.items()))
print("   per slice (iterations, launches, polls, re-bins): " + " ".join("%d/%d/%d/%d" % r for r in rows[1:]))
acc.close()
