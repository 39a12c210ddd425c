"""Where does the GPU's cold 1280x720 trajectory leave the oracle's (tests/golden/config5_720p_seed*.npz, model every 25
iterations, forward and reversed event order)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from better_flow_amd import accel, synth
H, W, s = 720, 1280, 3
for seed in (1, 0):
    z = np.load(os.path.join(ROOT, "tests", "golden", "config5_720p_seed%d.npz" % seed))
    fields = [str(f) for f in z["fields"]]
    sl = synth.make_slice(1000000, H, W, 0.030, seed=seed)
    for opts in ({}, {"binned": 0}):
        a = accel.Accel(max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
        for k, v in opts.items():
            a.set_option(k, v)
        a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
        a.set_cloud(s, H, W)
        o = a.default_opts()
        o.res_x, o.res_y, o.trace_cap = H, W, 9000
        rc, m, info = a.run(o)
        tr = a.get_trace(9000)
        a.close()
        print("seed", seed, opts, "GPU iterations", info.iterations, "oracle", int(z["fwd_iterations"]), int(z["rev_iterations"]),
              "dividers", info.x_divider, info.y_divider, info.rot_divider, info.div_divider, "oracle", z["fwd_dividers"])
        fe, re_ = z["fwd_every"], z["rev_every"]
        for row in range(0, min(len(fe), len(re_), (len(tr) + 24) // 25), 8):
            k = row * 25
            if k >= len(tr):
                break
            g = tr[k].model
            line = "  it %5d" % k
            for f in ("total_dx", "total_dy", "total_rot", "total_div", "cnt"):
                j = fields.index(f)
                line += "  %s g-f %+.3e f-r %+.3e" % (f, getattr(g, f) - fe[row][j], fe[row][j] - re_[row][j])
            line += "  xdiv %g rdiv %g" % (tr[k].x_divider, tr[k].rot_divider)
            print(line)
