"""The persistent loop kernel (k_fused_loop, bf_loop.hip) against the launch-per-iteration one-kernel loop: same bits
(model, iteration count, every trace record, per-event flow, the warm start that follows), and the time per iteration.

    python scripts/persist_check.py [events] [height] [width] [scale]
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from better_flow_amd import accel, synth


def run(sl, H, W, s, opts, max_iter=-1, reps=1):
    a = accel.Accel(max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
    for k, v in opts.items():
        a.set_option(k, v)
    best = None
    for rep in range(reps):
        a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
        a.set_cloud(s, H, W)
        o = a.default_opts()
        o.res_x, o.res_y, o.want_uv, o.trace_cap, o.max_iter = H, W, 1, 4096, max_iter
        a.synchronize()
        t0 = time.perf_counter()
        rc, m, info = a.run(o)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    trace = [t.model.as_dict() for t in a.get_trace(4096)]
    u, v = a.compute_uv()
    a.set_model(m)
    rc2, m2, info2 = a.run(o)
    trace2 = [t.model.as_dict() for t in a.get_trace(4096)]
    u2, v2 = a.compute_uv()
    a.close()
    return dict(rc=(rc, rc2), it=(info.iterations, info2.iterations), model=(m.as_dict(), m2.as_dict()), trace=(trace, trace2),
                flow=(u.tobytes(), v.tobytes(), u2.tobytes(), v2.tobytes())), info, best


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 180
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 240
    s = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    extra = dict(kv.split("=") for kv in sys.argv[5:])
    extra = {k: int(v) for k, v in extra.items()}
    sl = synth.make_slice(n, H, W, 0.03, seed=5)
    ref, iref, tref = run(sl, H, W, s, dict({"binned": 2, "fused": 2, "persist": 0}, **extra), reps=3)
    got, igot, tgot = run(sl, H, W, s, dict({"binned": 2, "fused": 2, "persist": 2}, **extra), reps=3)
    print("%d events %dx%d s%d: launch per iteration: %d iterations, %d launches, %d re-bins, %.2f us / iteration; "
          "persistent: %d iterations, %d launches, %d polls, %d re-bins, %.2f us / iteration" %
          (len(sl["t"]), W, H, s, iref.iterations, iref.launches, iref.rebins, 1e6 * tref / max(1, iref.iterations),
           igot.iterations, igot.launches, igot.polls, igot.rebins, 1e6 * tgot / max(1, igot.iterations)))
    bad = [k for k in ("rc", "it", "model", "trace", "flow") if got[k] != ref[k]]
    print("BITS EQUAL" if not bad else "DIFFERENT: %s" % bad)
    if bad:
        print(ref["it"], got["it"], ref["model"][0], got["model"][0])
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
