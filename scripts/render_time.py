"""Times the per-slice frame tiles (bf_projection_img, bf_color_time_img) on a config-2 slice (1M events, 346x260,
scale 3), and reports how many pixels of the colour image differ from the oracle's (f32 running sums).
usage: render_time.py [n_events]"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from better_flow_amd import accel, synth
import oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
H, W = 260, 346
sl = synth.make_slice(n, H, W, 0.03, seed=1)
a = accel.Accel(max_events=len(sl["t"]), max_rows=3 * H + 3, max_cols=3 * W + 3)
a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
a.set_cloud(3, H, W)
o = a.default_opts(); o.res_x, o.res_y = H, W
a.run(o)
for name, fn in (("projection_img", lambda f: a.projection_img(3, H, W, show_final=f)),
                 ("color_time_img", lambda f: a.color_time_img(3, H, W, show_final=f))):
    fn(False)
    t0 = time.perf_counter()
    for _ in range(10):
        fn(False); fn(True)
    print("%s: %.3f ms per image (incl. the device-to-host copy)" % (name, (time.perf_counter() - t0) / 20 * 1e3))
px, py, nx, ny = a.writeout_events()
oc = oracle.Cloud(sl["fr_x"], sl["fr_y"], sl["t"].astype(np.int64))
oc.pr_x[:], oc.pr_y[:] = px, py
for f in (True, False):
    g, r = a.color_time_img(3, H, W, show_final=f), oc.color_time_img(3, H, W, show_final=f)
    d = np.abs(g.astype(int) - r.astype(int)).max(axis=2)
    lit = r.any(axis=2)
    print("show_final=%d: lit %d, mask equal %s, pixels differing %d (%.4f %%), max diff %d" % (
        f, lit.sum(), np.array_equal(lit, g.any(axis=2)), (d > 0).sum(), 100.0 * (d > 0).sum() / lit.sum(), d.max()))
