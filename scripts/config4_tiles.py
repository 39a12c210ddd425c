"""BASELINE config 4: 32x32 grid of independent local optimizers over a 1M-event 346x260 slice."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from better_flow_amd import accel, synth
N, H, W, s, G = 1000000, 260, 346, 3, 32
min_events = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sl = synth.make_slice(N, H, W, 0.030, seed=1)
n = len(sl["t"])
acc = accel.Accel(max_events=n, max_rows=s * H + s, max_cols=s * W + s)
guard = (max(1, H // G), max(1, W // G))
best = None
for rep in range(4):
    acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
    acc.synchronize()
    t0 = time.perf_counter()
    models, infos = acc.run_tiles(G, G, s, (H, W), guard, min_events=min_events, hard_iter_cap=20000)
    acc.synchronize()
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
rc = np.array([i.rc for i in infos]); it = np.array([i.iterations for i in infos])
u, v = acc.compute_uv()
ran = rc == 0
out = {"config": "4: %dx%d tiles over a %d-event %dx%d slice, scale %d, guards: min_events=%d, RES=%dx%d" % (G, G, n, W, H, s, min_events, guard[1], guard[0]),
       "ms": best * 1e3, "mevents_per_s": n / best / 1e6, "tiles_optimised": int(ran.sum()), "tiles_skipped": int((rc == 1).sum()),
       "tiles_failed": int((rc < 0).sum()), "iterations_mean": float(it[ran].mean()) if ran.any() else 0, "iterations_max": int(it.max()),
       "tile_iterations_per_s": float(it.sum() / best),
       "flow_median_px_s": [float(np.median(u[np.abs(u) > 0])) if (np.abs(u) > 0).any() else 0.0, float(np.median(v[np.abs(v) > 0])) if (np.abs(v) > 0).any() else 0.0],
       "injected_px_s": list(sl["velocity"])}
print(json.dumps(out))
