"""The one-kernel iteration (k_fused_pass) against the two-kernel tile-binned loop: every bit of what a cold run, a warm
start and the per-event outputs return, on a few geometries -- with the default margin, with margins so small that events
outrun their bins (lost -> re-bin -> repeated pass), with the unpacked LDS planes forced, and with 64-row tiles.
usage: fused_check.py [quick]"""
import sys, os, hashlib, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from better_flow_amd import accel, synth


def canon(m):
    return tuple(np.float64(getattr(m, f)).tobytes() for f, _ in m._fields_)


def run(sl, H, W, s, opts, max_iter=-1):
    h = hashlib.sha256()
    def feed(*xs):
        for x in xs:
            h.update(np.ascontiguousarray(x).tobytes() if isinstance(x, np.ndarray) else repr(x).encode())
    a = accel.Accel(max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
    for k, v in opts.items():
        a.set_option(k, v)
    a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
    a.set_cloud(s, H, W)
    o = a.default_opts(); o.res_x, o.res_y, o.want_uv, o.trace_cap = H, W, 1, 64
    o.max_iter = max_iter
    t0 = time.perf_counter()
    rc, m, info = a.run(o)
    dt = time.perf_counter() - t0
    feed(rc, info.iterations, canon(m), [canon(t.model) for t in a.get_trace(64)])
    feed(*a.compute_uv())
    a.set_model(m); rc2, m2, info2 = a.run(o)
    feed(rc2, info2.iterations, canon(m2))
    feed(*a.compute_uv())
    a.close()
    return h.hexdigest()[:16], info.iterations, info2.iterations, info.rebins, info.launches, dt


quick = len(sys.argv) > 1
cases = [(1000000, 260, 346, 3, 1, -1), (300000, 480, 640, 3, 2, -1), (200000, 180, 240, 5, 3, -1), (50000, 180, 240, 1, 4, -1),
         (50000, 260, 346, 3, 6, -1), (3000000, 480, 640, 3, 8, 60)]
if quick: cases = cases[:2]
bad = 0
for (n, H, W, s, seed, mi) in cases:
    sl = synth.make_slice(n, H, W, 0.03, seed=seed)
    ref = run(sl, H, W, s, {"binned": 2, "fused": 0}, mi)
    print("%8d %dx%d s%d  two-kernel      %s it %d/%d rebins %d launches %d  %.2f ms" % ((n, W, H, s) + ref[:5] + (1e3 * ref[5],)))
    for name, o in (("fused", {}), ("fused D=2", {"fused_margin": 2}), ("fused D=1", {"fused_margin": 1}), ("fused 64 rows", {"fused_rows": 64}),
                    ("fused unpacked", {"bin_pack_limit": 20}), ("fused no predict", {"bin_predict": 0, "fused_margin": 3})):
        oo = {"binned": 2, "fused": 2}; oo.update(o)
        r = run(sl, H, W, s, oo, mi)
        ok = r[0] == ref[0]
        bad += 0 if ok else 1
        print("%8s %-22s %s it %d/%d rebins %d launches %d  %.2f ms  %s" % ("", name, r[0], r[1], r[2], r[3], r[4], 1e3 * r[5], "same bits" if ok else "DIFFERENT"))
print("FAILED" if bad else "all the same")
sys.exit(1 if bad else 0)
