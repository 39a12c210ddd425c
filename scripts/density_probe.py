"""How sparse is the time image along a cold run?  After K iterations: share of valid pixels, of 64-pixel row segments, of
4 x 16 and 8 x 8 blocks (aligned to the 16 x 64 stencil tiles) that hold at least one valid pixel -- what a wave-level skip
in the stencil kernel could save.  usage: density_probe.py [H W]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from better_flow_amd import accel, synth
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (720, 1280)
N, s = 1000000, 3
sl = synth.make_slice(N, H, W, 0.030, seed=1)
acc = accel.Accel(max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
for K in (1, 20, 100, 300, 1000, 2000, 4000, -1):
    acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"]); acc.set_cloud(s, H, W)
    o = acc.default_opts(); o.res_x, o.res_y, o.max_iter = H, W, K
    rc, m, info = acc.run(o)
    tim, cnt = acc.get_time_img()
    v = tim > 1e-6
    R, C = v.shape
    Rp, Cp = (R + 15) // 16 * 16, (C + 63) // 64 * 64
    vp = np.zeros((Rp, Cp), bool); vp[:R, :C] = v
    def share(br, bc):
        return vp.reshape(Rp // br, br, Cp // bc, bc).any(axis=(1, 3)).mean()
    print("after %5d iterations: valid px %.3f | 1x64 %.3f | 4x16 %.3f | 8x8 %.3f | 2x32 %.3f | 16x64 tile %.3f" %
          (info.iterations, v.mean(), share(1, 64), share(4, 16), share(8, 8), share(2, 32), share(16, 64)), flush=True)
acc.close()
