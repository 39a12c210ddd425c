"""Prints a digest of everything a few cold / warm / tile runs return (models, iteration counts, traces, images,
per-event outputs) -- run it with BF_ACCEL_LIB pointing at two builds to check that a change keeps every bit.
usage: BF_ACCEL_LIB=<lib> bits_check.py"""
import sys, os, hashlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from better_flow_amd import accel, synth

h = hashlib.sha256()
def feed(*xs):
    for x in xs:
        h.update(np.ascontiguousarray(x).tobytes() if isinstance(x, np.ndarray) else repr(x).encode())
def canon(m):
    return tuple(np.float64(getattr(m, f)).tobytes() for f, _ in m._fields_)
for (n, H, W, s, seed) in ((1000000, 260, 346, 3, 1), (300000, 480, 640, 3, 2), (200000, 180, 240, 5, 3), (50000, 180, 240, 1, 4),
                           (1000000, 720, 1280, 3, 5)):
    sl = synth.make_slice(n, H, W, 0.03, seed=seed)
    for binned in (2, 0):
        a = accel.Accel(max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
        a.set_option("binned", binned)
        a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
        a.set_cloud(s, H, W)
        o = a.default_opts(); o.res_x, o.res_y, o.want_uv, o.trace_cap = H, W, 1, 64
        if (H, W) == (720, 1280): o.max_iter = 40
        rc, m, info = a.run(o)
        feed(rc, info.iterations, canon(m), [canon(t.model) for t in a.get_trace(64)])
        feed(*a.compute_uv()); feed(*a.get_time_img())
        a.set_model(m); rc, m2, info2 = a.run(o)
        feed(rc, info2.iterations, canon(m2))
        print(n, H, W, s, "binned", binned, "iterations", info.iterations, info2.iterations, h.hexdigest()[:16])
        a.close()
sl = synth.make_slice(400000, 260, 346, 0.03, seed=7)
a = accel.Accel(max_events=len(sl["t"]), max_rows=3 * 260 + 3, max_cols=3 * 346 + 3)
a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
models, infos = a.run_tiles(32, 32, 3, (260, 346), (260 // 32, 346 // 32), 64, max_iter=-1)
feed([canon(m) for m in models], [(i.rc, i.iterations) for i in infos])
a.close()
print("digest", h.hexdigest())
