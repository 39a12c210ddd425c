"""What would ONE launch per iteration for B staged slices cost?  (a lock-step batched bf_run_many.)

A proxy that needs no new kernel: one context solving a slice that is B config-2 slices side by side -- a sensor B times
the area with B x 1M events at the same density, bins of the same kind (chosen per slice by bf_set_cloud) and stencil tiles (B x 752) -- runs
exactly the work-groups a batched launch of B slices would run, minus the per-slice early exits.  Its time per iteration
divided by B is the price of a slice-iteration in a batched launch, to be held against the four-context fan-out's
(bench.py: ms_per_step / iterations per slice).

    python scripts/batch_proxy.py
"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from better_flow_amd import accel, synth

s, iters = 3, 160
for (B, H, W) in ((1, 260, 346), (4, 520, 692), (8, 520, 1384)):
    sl = synth.make_slice(1000000 * B, H, W, 0.030, seed=1)
    for name, opts in (("update in the stencil tail, 512-thread bins", dict(co_schedule=1)),
                       ("update at the scatter head, 1024-thread bins", dict(co_schedule=0))):
        a = accel.Accel(max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
        for k, v in dict(dict(binned=2, fused=0, bin_compact=0, bin_split=0), **opts).items():
            a.set_option(k, v)
        o = a.default_opts()
        o.res_x, o.res_y, o.max_iter = H, W, iters
        best = None
        for rep in range(3):
            a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
            a.set_cloud(s, H, W)
            if rep == 2:
                a.profile_enable(1); a.profile_reset()
            a.synchronize()
            t0 = time.perf_counter()
            rc, m, info = a.run(o)
            a.synchronize()
            dt = time.perf_counter() - t0
            if rep < 2:
                best = dt if best is None else min(best, dt)
        p = a.profile_get()
        n = max(1, info.iterations)
        print("B = %d (%dx%d, %d events), %s: %.2f us per iteration = %.2f us per slice-iteration; kernels alone: K1 %.2f + K3 %.2f = %.2f us per slice-iteration (%d re-bins, %d overflow events)"
              % (B, W, H, len(sl["t"]), name, 1e6 * best / n, 1e6 * best / n / B, 1e3 * p.warp_scatter_ms / n / B, 1e3 * p.stencil_ms / n / B,
                 1e3 * (p.warp_scatter_ms + p.stencil_ms) / n / B, info.rebins, info.overflow_events))
        a.close()
