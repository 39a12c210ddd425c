"""Events on and next to the image borders, on sensors whose scaled size is no multiple of any tile size.

The stencil kernel of the tile-binned loop does not mask the lanes of its slab loads (bf_stencil.hip): a lane whose column lies
outside the image reads the always-zero cell of its row (local column 0 of bin column 0 -- image column -D), and rows outside the
image are skipped on the scalar unit.  That rests on (i) the scatter kernel never adding to a cell outside the image and flushing
its whole tile, zeros included, in every launch, (ii) nothing being read for rows outside the image -- whatever the margin: with
the test hook's margin of 2 the stencil halo (scale / 2 + 1 pixels, up to 4 here) reaches beyond the slabs.
Here most events sit within two sensor pixels of a border or in a corner, so that the time image's halo, the Scharr gate at
rows / columns 0 and R - 1 / C - 1 and the bins' margins all carry weight; every device loop must give the bits of the
global-atomics loop (which has no tiles, no slabs and another stencil kernel), and the first iterations must match the oracle.
"""
import numpy as np
import pytest

from helpers import make_accel

pytestmark = pytest.mark.gpu

K = 12


def border_slice(H, W, n, seed):
    rng = np.random.default_rng(seed)
    T = 0.03
    t = np.sort(rng.uniform(0, T, n))
    kind = rng.integers(0, 6, n)
    r = rng.uniform(0, H, n)
    c = rng.uniform(0, W, n)
    depth = rng.integers(0, 3, n)                      # 0, 1 or 2 pixels from the border
    r = np.where(kind == 0, depth, r)                  # top rows
    r = np.where(kind == 1, H - 1 - depth, r)          # bottom rows
    c = np.where(kind == 2, depth, c)                  # left columns
    c = np.where(kind == 3, W - 1 - depth, c)          # right columns
    corner = kind == 4                                 # the four corners, 3 x 3 sensor pixels each
    r = np.where(corner, np.where(rng.random(n) < 0.5, depth, H - 1 - depth), r)
    c = np.where(corner, np.where(rng.random(n) < 0.5, rng.integers(0, 3, n), W - 1 - rng.integers(0, 3, n)), c)
    v = np.array([40.0, -55.0]) * (H / 180.0)          # px/s: a slow drift, the border events stay at the border
    r = r + v[0] * t
    c = c + v[1] * t
    fr_x = np.clip(np.floor(r), 0, H - 1).astype(np.int32)
    fr_y = np.clip(np.floor(c), 0, W - 1).astype(np.int32)
    return dict(fr_x=fr_x, fr_y=fr_y, t=(t * 1e9).astype(np.int32))


def solve(a, sl, H, W, s):
    a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
    a.set_cloud(s, H, W)
    fmt = a.get_stat("scatter_format")
    o = a.default_opts()
    o.res_x, o.res_y, o.want_uv, o.trace_cap, o.max_iter, o.min_events = H, W, 1, K + 2, K, 10
    rc, m, info = a.run(o)
    trace = a.get_trace(K + 2)
    u, v = a.compute_uv()
    timg = a.get_time_img()
    return dict(rc=rc, it=info.iterations, model=m.as_dict(), trace=[t.model.as_dict() for t in trace],
                flow=(u.tobytes(), v.tobytes()), timg=tuple(np.ascontiguousarray(x).tobytes() for x in timg)), fmt, trace


FORMS = [("global atomics", dict(binned=0, fused=0), -1),
         ("dense slabs, head update", dict(binned=2, fused=0, bin_compact=0, bin_split=0), 0),
         ("dense slabs, tail update", dict(binned=2, fused=0, bin_compact=0, bin_split=0, co_schedule=1), 0),
         ("dense slabs, margin 2", dict(binned=2, fused=0, bin_compact=0, bin_split=0, debug_margin=2), 0),
         ("own pixels + margin plane", dict(binned=2, fused=0, bin_compact=0, bin_split=2), 3),
         ("event lists", dict(binned=2, fused=0, bin_compact=2), 2),
         ("one-kernel iteration", dict(binned=2, fused=2, persist=0), None)]


@pytest.mark.parametrize("geo", [(181, 243, 3, 1), (97, 130, 5, 2), (260, 346, 1, 3), (65, 67, 7, 4), (129, 193, 3, 5)],
                         ids=lambda g: "%dx%d_s%d" % (g[1], g[0], g[2]))
def test_border_events_same_bits_in_every_loop_and_oracle(accel_mod, oracle_lib, geo):
    H, W, s, seed = geo
    sl = border_slice(H, W, 120000, seed)
    res = {}
    for name, opts, want_fmt in FORMS:
        a = make_accel(accel_mod, opts, max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
        try:
            got, fmt, trace = solve(a, sl, H, W, s)
        finally:
            a.close()
        if want_fmt is not None:
            assert fmt == want_fmt, (name, fmt)
        res[name] = (got, trace)
    ref = res["global atomics"][0]
    assert ref["it"] >= 2
    for name, (got, _) in res.items():
        for key in ("rc", "it", "model", "trace", "flow", "timg"):
            assert got[key] == ref[key], (name, key)
    # ... and the oracle on the same slice: the first two updates (before any pixel-boundary crossing can separate the paths)
    o = oracle_lib.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
    w_ = o.set_cloud(s, H, W)
    rc_, lp_, otr = o.run(w_, oracle_lib.Model(), max_iter=K, res_x=H, res_y=W, trace_cap=K + 2, min_events=10)
    gtr = res["global atomics"][1]
    for k in range(2):
        om, gm = otr[k].model, gtr[k].model
        assert om.cnt == gm.cnt, (k, om.cnt, gm.cnt)
        for f in ("cx", "cy", "dx", "dy", "rot", "div"):
            a_, b_ = getattr(om, f), getattr(gm, f)
            assert abs(a_ - b_) <= 3e-4 * max(1.0, abs(a_)), (k, f, a_, b_)
