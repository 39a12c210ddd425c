"""Per-operator GPU timing through the C-ABI (hipEvent-bracketed launches)."""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from better_flow_amd import accel, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
H, W, s = 260, 346, 3
sl = synth.make_slice(N, H, W, 0.030, seed=1)
n = len(sl["t"])
acc = accel.Accel(max_events=n, max_rows=s * H + s, max_cols=s * W + s)
for split in (0, 1):
    acc.set_option("force_split", split)
    acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
    acc.set_cloud(s, H, W)
    acc.profile_enable(1)
    for label, prm in (("zero-flow", (0, 0, 0, 0, 0, 0)), ("converged", (0.2756, -0.5494, 129, 172, 2e-4, 2.5e-5))):
        acc.profile_reset()
        for _ in range(30):
            acc.project_4param_reinit(*prm)
        p = acc.profile_get()
        print("split", split, label, "warp-only(+write n) us", 1e3 * p.warp_scatter_ms / p.warp_scatter_launches)
        acc.profile_reset()
        for _ in range(30):
            acc.get_time_img(False, False)
        p = acc.profile_get()
        print("split", split, label, "scatter-only us", 1e3 * p.warp_scatter_ms / p.warp_scatter_launches,
              " stencil(+time,count out) us", 1e3 * p.stencil_ms / p.stencil_launches)
    acc.profile_enable(0)
