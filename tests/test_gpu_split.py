"""The interior + margin format of the tile-binned scatter ("bin_split", bf_scatter.hip: flush_split).

A dense slice's bin writes its own pixels into a tiled image and adds the words its events left in the margin of its LDS
tile to a double-buffered margin plane, which it clears again -- from a per-bin list -- at its next executed launch.  All
sums are integers, so the format must return the BITS of the dense-slab loop: model, iteration count, every trace record,
per-event flow, time image -- cold, warm-started, with the update at the scatter head and in the stencil tail, with tight
margins (events outrun the tiles: overflow path + re-bins), on a context that is reused across geometries, formats and
loops (the margin planes and lists must be left consistent by every run), and at the geometry `auto` takes it for
(640x480: also held to the oracle there by test_gpu_geometries.py, which runs with the defaults).
"""
import numpy as np
import pytest

from better_flow_amd import synth
from helpers import make_accel

pytestmark = pytest.mark.gpu


def solve(a, sl, H, W, s, max_iter=-1, warm_from=None):
    a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
    a.set_cloud(s, H, W)
    fmt = a.get_stat("scatter_format")
    if warm_from is not None:
        a.set_model(warm_from)
    o = a.default_opts()
    o.res_x, o.res_y, o.want_uv, o.trace_cap, o.max_iter = H, W, 1, 4096, max_iter
    rc, m, info = a.run(o)
    trace = [t.model.as_dict() for t in a.get_trace(4096)]
    u, v = a.compute_uv()
    timg = a.get_time_img()
    return dict(rc=rc, it=info.iterations, model=m.as_dict(), trace=trace, flow=(u.tobytes(), v.tobytes()),
                timg=tuple(np.ascontiguousarray(x).tobytes() for x in timg)), m, fmt, info


def ctx(accel_mod, n, H, W, s, opts):
    return make_accel(accel_mod, opts, max_events=n, max_rows=s * H + s, max_cols=s * W + s)   # ("debug_margin": tests/helpers.py)


@pytest.mark.parametrize("case", [(1000000, 260, 346, 3, 1, -1), (600000, 480, 640, 3, 2, 120), (400000, 180, 240, 5, 3, -1),
                                  (1000000, 720, 1280, 1, 4, 60)],
                         ids=lambda c: "%dev_%dx%d_s%d" % (c[0], c[2], c[1], c[3]))
@pytest.mark.parametrize("co", [0, 1], ids=["head_update", "tail_update"])
def test_same_bits_as_dense_slabs(accel_mod, case, co):
    n, H, W, s, seed, max_iter = case
    sl = synth.make_slice(n, H, W, 0.03, seed=seed)
    base = {"binned": 2, "fused": 0, "bin_compact": 0, "co_schedule": co}
    res = {}
    for name, o in (("dense", {"bin_split": 0}), ("split", {"bin_split": 2}), ("split, margin 4", {"bin_split": 2, "debug_margin": 4}),
                    ("split, margin 2, no prediction", {"bin_split": 2, "debug_margin": 2, "bin_predict": 0})):
        a = ctx(accel_mod, len(sl["t"]), H, W, s, dict(base, **o))
        cold, m, fmt, info = solve(a, sl, H, W, s, max_iter)
        assert fmt == (0 if name == "dense" else 3), (name, fmt)
        warm, m2, _, _ = solve(a, sl, H, W, s, max_iter, warm_from=m)      # same context: the planes / lists of the cold run
        warm2, _, _, _ = solve(a, sl, H, W, s, max_iter, warm_from=m2)
        a.close()
        res[name] = (cold, warm, warm2)
        if name == "split, margin 2, no prediction":
            assert info.overflow_events > 0, "a 2-pixel margin without prediction must exercise the overflow path"
    assert res["dense"][0]["rc"] == 0 and res["dense"][0]["it"] > 20
    for name in res:
        for i, leg in enumerate(("cold", "warm", "warm again")):
            for key in ("rc", "it", "model", "trace", "flow", "timg"):
                assert res[name][i][key] == res["dense"][i][key], (name, leg, key)


def test_one_context_through_geometries_formats_and_loops(accel_mod):
    """A context reused the way the slice farm reuses it: every slice's bits equal a fresh dense-slab context's."""
    seq = [(700000, 480, 640, 3, 11, 40, {}),                       # auto: split (1.5 M pixels and more, GPU to itself)
           (700000, 480, 640, 3, 12, 41, {}),                       # same grid, odd iteration count before it
           (300000, 260, 346, 3, 13, 30, {"bin_split": 2}),         # another grid: the old lists are cleared first
           (40000, 180, 240, 3, 14, 25, {"bin_split": 1}),          # one-kernel loop in between (touches neither plane)
           (300000, 260, 346, 3, 15, 33, {"bin_split": 2}),
           (900000, 720, 1280, 3, 16, 20, {"bin_split": 2}),        # event lists (sparse): not a dense slice, split does not apply
           (700000, 260, 346, 5, 17, 27, {"bin_split": 2, "co_schedule": 1}),
           (700000, 260, 346, 5, 18, 28, {"bin_split": 2, "co_schedule": 0}),
           (700000, 480, 640, 3, 11, 40, {"bin_split": 1})]
    big = ctx(accel_mod, 1000000, 720, 1280, 5, {})
    want_fmt = [3, 3, 3, None, 3, 2, 3, 3, 3]
    for i, (n, H, W, s, seed, max_iter, o) in enumerate(seq):
        sl = synth.make_slice(n, H, W, 0.03, seed=seed)
        for k, v in o.items():
            big.set_option(k, v)
        got, m, fmt, _ = solve(big, sl, H, W, s, max_iter)
        if want_fmt[i] is not None:
            assert fmt == want_fmt[i], (i, fmt)
        else:
            assert big.get_stat("one_kernel") == 1, i
        fresh = ctx(accel_mod, len(sl["t"]), H, W, s, {"bin_split": 0, "fused": 0})
        ref, _, fmt0, _ = solve(fresh, sl, H, W, s, max_iter)
        fresh.close()
        assert fmt0 in (-1, 0, 1, 2), fmt0
        for key in ("rc", "it", "model", "trace", "flow", "timg"):
            assert got[key] == ref[key], (i, key)
    big.close()


def test_auto_rule(accel_mod):
    for (n, H, W, s, co, want) in ((1000000, 480, 640, 3, 0, 3), (1000000, 480, 640, 3, 1, 0), (1000000, 260, 346, 3, 0, 0),
                                   (1000000, 260, 346, 5, 0, 3), (1000000, 720, 1280, 3, 0, 2)):
        sl = synth.make_slice(n, H, W, 0.03, seed=5)
        a = ctx(accel_mod, len(sl["t"]), H, W, s, {"co_schedule": co})
        a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
        a.set_cloud(s, H, W)
        assert a.get_stat("scatter_format") == want, (H, W, s, co)
        a.close()
