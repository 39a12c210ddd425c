"""Parity of the HIP path (through the C-ABI) against the CPU oracle on the same inputs.

Bars: window / warp state / event-count image / Scharr planes bit-exact;
time image <= 1e-6 relative (the reference's own f32 accumulation-order noise) and
bit-exact where a single event hit the pixel; moments <= 1e-9 relative; converged
(u, v) <= 1e-4 relative or 0.02 px/s, iteration count within +-1.
"""
import os

import numpy as np
import pytest

from better_flow_amd import synth
from helpers import make_accel

pytestmark = pytest.mark.gpu

PARAMS = [  # dnx, dny, cx, cy, div, crl
    (0.0, 0.0, 0.0, 0.0, 0.0, 0.0),
    (0.35, -0.7, 0.0, 0.0, 0.0, 0.0),
    (0.2, 0.4, 88.5, 121.25, 3.0e-4, -2.0e-4),
    (-0.15, 0.05, 90.0, 119.0, -1.0e-3, 1.5e-3),
]


def small_slice(n=20000, H=180, W=240, seed=7, dur=0.05):
    return synth.make_slice(n, H, W, dur, seed=seed)


def make_pair(oracle_lib, accel_mod, sl, scale, split=False, noise=None):
    H, W = sl["height"], sl["width"]
    oc = oracle_lib.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
    if noise is not None:
        oc.noise[:] = noise
    ow = oc.set_cloud(scale, H, W)
    acc = accel_mod.Accel(max_events=max(len(sl["t"]), 8192), max_rows=scale * H + scale,
                          max_cols=scale * W + scale)
    acc.set_option("force_split", 1 if split else 0)
    acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"], noise)
    gw = acc.set_cloud(scale, H, W)
    return oc, ow, acc, gw


@pytest.mark.parametrize("scale", [1, 3, 5])
def test_window_matches(oracle_lib, accel_mod, scale):
    sl = small_slice()
    oc, ow, acc, gw = make_pair(oracle_lib, accel_mod, sl, scale)
    for k in ("scale", "x_min", "y_min", "x_max", "y_max", "metric_wsizex", "metric_wsizey",
              "scale_img_x", "scale_img_y", "x_shift", "y_shift"):
        assert getattr(ow, k) == getattr(gw, k), k
    acc.close()


def test_warp_state_bit_exact(oracle_lib, accel_mod):
    sl = small_slice()
    oc, ow, acc, gw = make_pair(oracle_lib, accel_mod, sl, 3)
    pr_x, pr_y, nx, ny = acc.writeout_events()     # after reset: pr == fr, n == 0
    assert np.array_equal(pr_x, oc.pr_x) and np.array_equal(pr_y, oc.pr_y)
    assert not nx.any() and not ny.any()
    for prm in PARAMS:                              # chained: each warp reads the previous pr
        oc.project_4param_reinit(*prm)
        acc.project_4param_reinit(*prm)
        pr_x, pr_y, nx, ny = acc.writeout_events()
        assert np.array_equal(pr_x, oc.pr_x)
        assert np.array_equal(pr_y, oc.pr_y)
        assert np.array_equal(nx, oc.nx)
        assert np.array_equal(ny, oc.ny)
    u, v = acc.compute_uv()
    ou, ov = oc.compute_uv()
    np.testing.assert_allclose(u, ou, rtol=1e-14, atol=0)
    np.testing.assert_allclose(v, ov, rtol=1e-14, atol=0)
    acc.close()


def test_incremental_warp_bit_exact(oracle_lib, accel_mod):
    """AccelLib::project_4param (accel_lib.h:275-281; Event::project_4param, event.h:88-96): the incremental form -- the reinit
    form's dn ADDED to the event's (nx, ny).  Dead in the reference (its only call is commented out) and exported for signature
    completeness: pr / nx / ny bit for bit as the oracle's, from reset, chained, mixed with the reinit form, after a run (whose
    per-event outputs are tile-sorted on the device) and followed by the count image."""
    sl = small_slice()
    oc, ow, acc, gw = make_pair(oracle_lib, accel_mod, sl, 3)

    def same():
        pr_x, pr_y, nx, ny = acc.writeout_events()
        assert np.array_equal(pr_x, oc.pr_x) and np.array_equal(pr_y, oc.pr_y)
        assert np.array_equal(nx, oc.nx) and np.array_equal(ny, oc.ny)
    for prm in PARAMS:                              # from reset: n == 0, then chained
        oc.project_4param(*prm); acc.project_4param(*prm)
        same()
    oc.project_4param_reinit(*PARAMS[1]); acc.project_4param_reinit(*PARAMS[1])
    oc.project_4param(*PARAMS[0]); acc.project_4param(*PARAMS[0])
    same()
    _, ocnt = oc.get_time_img(ow)
    _, gcnt = acc.get_time_img(want_time=False)
    assert np.array_equal(gcnt, ocnt.astype(np.uint32))
    # after a run the device holds (nx, ny) in tile-sorted order: the incremental form must find each event's own
    acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"]); acc.set_cloud(3, sl["height"], sl["width"])
    o = acc.default_opts(); o.res_x, o.res_y, o.max_iter = sl["height"], sl["width"], 5
    acc.run(o)
    pr_x, pr_y, nx, ny = acc.writeout_events()
    acc.project_4param(*PARAMS[2])
    oc2 = oracle_lib.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
    oc2.pr_x[:], oc2.pr_y[:], oc2.nx[:], oc2.ny[:] = pr_x, pr_y, nx, ny
    oc2.project_4param(*PARAMS[2])
    pr_x, pr_y, nx, ny = acc.writeout_events()
    assert np.array_equal(pr_x, oc2.pr_x) and np.array_equal(pr_y, oc2.pr_y) and np.array_equal(nx, oc2.nx) and np.array_equal(ny, oc2.ny)
    acc.close()


@pytest.mark.parametrize("scale,split", [(1, False), (3, False), (3, True), (5, False)])
def test_count_and_time_image(oracle_lib, accel_mod, scale, split):
    sl = small_slice()
    oc, ow, acc, gw = make_pair(oracle_lib, accel_mod, sl, scale, split=split)
    for prm in PARAMS:
        oc.project_4param_reinit(*prm)
        acc.project_4param_reinit(*prm)
        otime, ocnt = oc.get_time_img(ow)
        gtime, gcnt = acc.get_time_img()
        assert np.array_equal(gcnt, ocnt.astype(np.uint32)), "event-count image must be bit-exact"
        assert gcnt.sum() > 0
        one = ocnt == 1.0
        assert np.array_equal(gtime[one], otime[one]), "single-event pixels must be bit-exact"
        np.testing.assert_allclose(gtime, otime, rtol=1e-6, atol=0)
    acc.close()


def test_time_image_idempotent_and_deterministic(oracle_lib, accel_mod):
    sl = small_slice()
    oc, ow, acc, gw = make_pair(oracle_lib, accel_mod, sl, 3)
    acc.project_4param_reinit(*PARAMS[2])
    t1, c1 = acc.get_time_img()
    t2, c2 = acc.get_time_img()
    assert np.array_equal(t1, t2) and np.array_equal(c1, c2)
    acc.close()


def test_noise_mask(oracle_lib, accel_mod):
    sl = small_slice()
    noise = (np.arange(len(sl["t"])) % 3 == 0).astype(np.uint8)
    oc, ow, acc, gw = make_pair(oracle_lib, accel_mod, sl, 3, noise=noise)
    oc.project_4param_reinit(*PARAMS[1])
    acc.project_4param_reinit(*PARAMS[1])
    otime, ocnt = oc.get_time_img(ow)
    gtime, gcnt = acc.get_time_img()
    assert np.array_equal(gcnt, ocnt.astype(np.uint32))
    np.testing.assert_allclose(gtime, otime, rtol=1e-6, atol=0)
    acc.close()


def test_sobel_bit_exact(oracle_lib, accel_mod):
    sl = small_slice()
    oc, ow, acc, gw = make_pair(oracle_lib, accel_mod, sl, 3)
    oc.project_4param_reinit(*PARAMS[1])
    otime, _ = oc.get_time_img(ow)
    rng = np.random.default_rng(3)
    dense = rng.uniform(2e-6, 0.05, size=(97, 211)).astype(np.float32)
    holes = dense.copy()
    holes[rng.uniform(size=holes.shape) < 0.1] = 0.0
    holes[5, 5] = 1e-6            # exactly on the validity threshold (invalid: not > 1e-6)
    holes[6, 9] = np.float32(1.0000001e-6)
    for img in (otime, dense, holes, np.zeros((3, 3), np.float32), dense[:2, :5], dense[:1, :1]):
        ogx, ogy = oracle_lib.sobel(img)
        ggx, ggy = acc.sobel(img)
        assert np.array_equal(ggx, ogx)
        assert np.array_equal(ggy, ogy)
    acc.close()


def test_fast_model(oracle_lib, accel_mod):
    sl = small_slice()
    oc, ow, acc, gw = make_pair(oracle_lib, accel_mod, sl, 3)
    oc.project_4param_reinit(*PARAMS[2])
    acc.project_4param_reinit(*PARAMS[2])
    otime, _ = oc.get_time_img(ow)
    gtime, _ = acc.get_time_img()
    # host image: same input as the oracle.  Resident image: the GPU's own time image, so
    # the oracle is evaluated on that image (it differs from otime by f32 summation order).
    for gm, om in ((acc.fast_model(otime), oracle_lib.fast_model(otime)),
                   (acc.fast_model(), oracle_lib.fast_model(gtime))):
        assert gm.cnt == om.cnt
        assert gm.cx == om.cx and gm.cy == om.cy           # integer sums: exact
        for k in ("dx", "dy", "rot", "div"):
            a, b = getattr(gm, k), getattr(om, k)
            assert abs(a - b) <= 1e-9 * max(abs(b), 1e-12) + 1e-15, (k, a, b)
    acc.close()


def _flow_close(u, ou, rel=1e-4, abs_=0.02):
    tol = np.maximum(rel * np.abs(ou), abs_)
    ok = np.all(np.abs(u - ou) <= tol)
    if not ok:
        print("flow deviation: max abs %.3e px/s, max rel %.3e" %
              (np.abs(u - ou).max(), (np.abs(u - ou) / np.maximum(np.abs(ou), 1e-9)).max()))
    return ok


@pytest.mark.parametrize("split", [False, True])
def test_run_cold_200k(oracle_lib, accel_mod, split):
    H, W = 260, 346
    sl = synth.make_slice(200000, H, W, 0.030, seed=1)
    oc, ow, acc, gw = make_pair(oracle_lib, accel_mod, sl, 3, split=split)
    om = oracle_lib.Model()
    orc, oloop, otrace = oc.run(ow, om, res_x=H, res_y=W, trace_cap=64)
    opts = acc.default_opts()
    opts.res_x, opts.res_y, opts.trace_cap = H, W, 64
    grc, gm, info = acc.run(opts)
    assert grc == orc == 0
    assert abs(info.iterations - oloop.itercount) <= 1, (info.iterations, oloop.itercount)
    gtrace = acc.get_trace(64)
    # first iterations: same trajectory to rounding (only the time image differs, by 1e-7)
    for k in range(min(16, len(gtrace), len(otrace))):
        g, o = gtrace[k].model, otrace[k].model
        assert g.cnt == o.cnt, k
        for f in ("dx", "dy", "total_dx", "total_dy"):
            assert abs(getattr(g, f) - getattr(o, f)) <= 2e-5 * max(abs(getattr(o, f)), 1e-3), (k, f)
    u, v = acc.compute_uv()
    ou, ov = oc.compute_uv()
    assert _flow_close(u, ou) and _flow_close(v, ov)
    # known answer: the injected flow is recovered
    vr, vc = sl["velocity"]
    assert abs(u.mean() - vr) < 0.01 * abs(vr) and abs(v.mean() - vc) < 0.01 * abs(vc)
    # a second run on the same ctx / slice is bit-identical (deterministic integer scatter)
    acc.set_cloud(3, H, W)
    grc2, gm2, info2 = acc.run(opts)
    assert info2.iterations == info.iterations
    assert gm2.as_dict() == gm.as_dict()
    acc.close()


def test_run_warm_start(oracle_lib, accel_mod):
    H, W = 260, 346
    a = synth.make_slice(100000, H, W, 0.030, seed=11)
    b = synth.make_slice(100000, H, W, 0.030, seed=12)
    oc, ow, acc, gw = make_pair(oracle_lib, accel_mod, a, 3)
    om = oracle_lib.Model()
    _, oloop_a, _ = oc.run(ow, om, res_x=H, res_y=W)
    opts = acc.default_opts()
    opts.res_x, opts.res_y = H, W
    _, gm, info_a = acc.run(opts)
    assert abs(info_a.iterations - oloop_a.itercount) <= 1
    # Slice b warm-started ("STM", dvs_flow.h:218-219) from the SAME model on both sides, so
    # that the comparison is of the warm path itself: the cold models agree only to ~1e-5
    # after 100+ gradient steps, which moves the (threshold-crossing) stop by a few steps.
    oc2 = oracle_lib.Cloud(b["fr_x"], b["fr_y"], b["t"])
    ow2 = oc2.set_cloud(3, H, W)
    om2 = oc2.set_model(om)
    orc, oloop, _ = oc2.run(ow2, om2, res_x=H, res_y=W)
    acc.upload_events(b["fr_x"], b["fr_y"], b["t"])
    acc.set_cloud(3, H, W)
    acc.set_model(accel_mod.Model(**om.as_dict()))
    grc, gm2, info_b = acc.run(opts)
    assert grc == orc == 0
    assert abs(info_b.iterations - oloop.itercount) <= 1
    assert info_b.iterations < info_a.iterations        # warm start converges faster
    u, v = acc.compute_uv()
    ou, ov = oc2.compute_uv()
    assert _flow_close(u, ou) and _flow_close(v, ov)
    # The chained estimate (own cold model -> warm start, the way dvs_flow.h:218-224 runs) against the oracle's own
    # chain.  Yardstick: the oracle chain run on the same two slices in reversed event order (the reference's f32 time
    # sums depend on the order, accel_lib.h:162); the GPU chain must agree with the oracle chain within north_star's
    # bar (1e-4 / 0.02 px/s) or 4 x the oracle's own forward / reversed spread, whichever is larger.
    def oracle_chain(rev):
        sel = slice(None, None, -1) if rev else slice(None)
        c1 = oracle_lib.Cloud(a["fr_x"][sel].copy(), a["fr_y"][sel].copy(), a["t"][sel].copy())
        m1 = oracle_lib.Model()
        c1.run(c1.set_cloud(3, H, W), m1, res_x=H, res_y=W)
        c2 = oracle_lib.Cloud(b["fr_x"][sel].copy(), b["fr_y"][sel].copy(), b["t"][sel].copy())
        w2 = c2.set_cloud(3, H, W)
        m2 = c2.set_model(m1)
        _, lp2, _ = c2.run(w2, m2, res_x=H, res_y=W)
        uu, vv = c2.compute_uv()
        return lp2.itercount, uu[sel], vv[sel]
    it_f, uf, vf = oracle_chain(False)
    it_r, ur, vr = oracle_chain(True)
    acc.set_cloud(3, H, W)
    acc.set_model(gm)
    _, _, info_c = acc.run(opts)
    u2, v2 = acc.compute_uv()
    yard = max(np.abs(uf - ur).max(), np.abs(vf - vr).max())
    dev = max(np.abs(u2 - uf).max(), np.abs(v2 - vf).max())
    print("chained warm start: GPU vs oracle chain %.3e px/s (iterations %d vs %d), oracle forward vs reversed %.3e px/s "
          "(iterations %d vs %d)" % (dev, info_c.iterations, it_f, yard, it_f, it_r))
    # (this pair of slices: the oracle itself needs 8 iterations forward and 11 reversed, and lands 0.29 px/s apart)
    assert abs(info_c.iterations - it_f) <= abs(it_f - it_r) + 1
    for g_, o_ in ((u2, uf), (v2, vf)):
        assert np.all(np.abs(g_ - o_) <= np.maximum(np.maximum(1e-4 * np.abs(o_), 0.02), 4.0 * yard))
    acc.close()


def test_run_max_iter(oracle_lib, accel_mod):
    H, W = 180, 240
    sl = small_slice(30000, H, W, seed=5)
    oc, ow, acc, gw = make_pair(oracle_lib, accel_mod, sl, 3)
    om = oracle_lib.Model()
    orc, oloop, _ = oc.run(ow, om, max_iter=10, res_x=H, res_y=W)
    opts = acc.default_opts()
    opts.max_iter = 10
    grc, gm, info = acc.run(opts)
    assert info.iterations == oloop.itercount == 11      # "itercount > max" breaks after 11 steps
    for f in ("total_dx", "total_dy", "total_rot", "total_div"):
        assert abs(getattr(gm, f) - getattr(om, f)) <= 1e-5 * max(abs(getattr(om, f)), 1e-6), f
    acc.close()


def test_guards_and_edges(oracle_lib, accel_mod):
    H, W = 180, 240
    # fewer than 1000 events: run() returns 1 (optimizer_rolling.h:57-58)
    sl = small_slice(600, H, W, seed=9)
    oc, ow, acc, gw = make_pair(oracle_lib, accel_mod, sl, 3)
    om = oracle_lib.Model()
    assert oc.run(ow, om, res_x=H, res_y=W)[0] == 1
    rc, gm, info = acc.run()
    assert rc == accel_mod.BF_SKIPPED and info.iterations == 0
    u, v = acc.compute_uv()
    assert not u.any() and not v.any()
    # window too small: every event becomes noise (optimizer_rolling.h:49-55)
    tiny = {"fr_x": np.full(2000, 50, np.int32) + (np.arange(2000) % 3).astype(np.int32),
            "fr_y": np.full(2000, 60, np.int32) + (np.arange(2000) % 4).astype(np.int32),
            "t": np.arange(2000, dtype=np.int64) * 1000, "height": H, "width": W}
    acc.upload_events(tiny["fr_x"], tiny["fr_y"], tiny["t"])
    acc.set_cloud(3, H, W)
    rc, _, _ = acc.run()
    assert rc == accel_mod.BF_SKIPPED
    _, cnt = acc.get_time_img()
    assert cnt.sum() == 0
    # empty slice
    acc.upload_events(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32))
    w = acc.set_cloud(3, H, W)
    assert (w.x_min, w.x_max, w.y_min, w.y_max) == (H, 0, W, 0)
    acc.close()
    # capacity errors are reported, not ignored
    small = accel_mod.Accel(max_events=1024, max_rows=64, max_cols=64)
    with pytest.raises(accel_mod.BfError):
        small.upload_events(np.zeros(5000, np.int32), np.zeros(5000, np.int32), np.zeros(5000, np.int32))
    small.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
    with pytest.raises(accel_mod.BfError):
        small.set_cloud(3, H, W)
    small.close()


def test_golden_vectors_gpu(accel_mod):
    """The committed oracle-generated vectors (tests/golden/), no oracle build needed."""
    import json
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    man = json.load(open(os.path.join(gold, "manifest.json")))
    z = np.load(os.path.join(gold, man["file"]))
    H, W, s = man["height"], man["width"], man["scale"]
    acc = accel_mod.Accel(max_events=len(z["t"]), max_rows=s * H + s, max_cols=s * W + s)
    acc.upload_events(z["fr_x"], z["fr_y"], z["t"])
    w = acc.set_cloud(s, H, W)
    assert [w.x_min, w.x_max, w.y_min, w.y_max, w.scale_img_x, w.scale_img_y] == man["window"]
    for k, prm in enumerate(man["warps"]):
        acc.project_4param_reinit(*prm)
        pr_x, pr_y, nx, ny = acc.writeout_events()
        assert np.array_equal(pr_x, z["pr_x_%d" % k]) and np.array_equal(pr_y, z["pr_y_%d" % k])
        assert np.array_equal(nx, z["nx_%d" % k]) and np.array_equal(ny, z["ny_%d" % k])
        gtime, gcnt = acc.get_time_img()
        assert np.array_equal(gcnt, z["cnt_%d" % k].astype(np.uint32))
        np.testing.assert_allclose(gtime, z["time_%d" % k], rtol=1e-6, atol=0)
    ggx, ggy = acc.sobel(z["time_%d" % (len(man["warps"]) - 1)])
    assert np.array_equal(ggx, z["gx"]) and np.array_equal(ggy, z["gy"])
    acc.upload_events(z["fr_x"], z["fr_y"], z["t"])
    acc.set_cloud(s, H, W)
    opts = acc.default_opts()
    opts.res_x, opts.res_y = H, W
    rc, m, info = acc.run(opts)
    assert rc == man["rc"] and abs(info.iterations - man["iterations"]) <= 1
    u, v = acc.compute_uv()
    assert _flow_close(u, z["u"]) and _flow_close(v, z["v"])
    acc.close()


def test_golden_images_gpu(accel_mod):
    """Second fixture (no oracle build needed): contrast-score images, scores and descent result, projection images
    -- all integer work, bit-exact."""
    import json
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    man = json.load(open(os.path.join(gold, "manifest.json")))
    z, zi = np.load(os.path.join(gold, man["file"])), np.load(os.path.join(gold, man["images_file"]))
    H, W, s, loc = man["height"], man["width"], man["scale"], man["local"]
    acc = accel_mod.Accel(max_events=len(z["t"]), max_rows=s * H + s, max_cols=s * W + s)
    acc.upload_events(z["fr_x"], z["fr_y"], z["t"])
    for tag, center, wsz in (("cloud", None, 0), ("window", tuple(loc["center"]), loc["wsz"])):
        acc.local_set_window(s, center=center, wsz=wsz)
        for k, (nx, ny) in enumerate(loc["candidates"]):
            sc, img = acc.local_iteration_step(nx, ny, want_img=True)
            assert sc == loc["scores_" + tag][k] and np.array_equal(img, zi["local_%s_%d" % (tag, k)])
    acc.local_set_window(s)
    rc, st = acc.local_run(H, W)
    assert [rc, st.nx, st.ny, st.last_score, st.evaluations] == [loc["run"][k] for k in ("rc", "nx", "ny", "last_score", "evaluations")]
    acc.set_cloud(s, H, W)
    assert np.array_equal(acc.projection_img(s, H, W, show_final=True), zi["proj_raw"])
    acc.project_4param_reinit(*man["warps"][2])
    assert np.array_equal(acc.projection_img(s, H, W), zi["proj_warp2"])
    acc.close()


def test_golden_stream_gpu(accel_mod):
    """Third fixture (no oracle build needed): the warm-started (STM) run of the next slice from the golden model of
    the previous one -- same iteration count within one, the first iterations' accumulators to 1e-6, the final flow to
    the stated tolerance -- and the colour-coded time images (covered pixels exact, at most 1 % of them off by one hue /
    saturation step)."""
    import json
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    man = json.load(open(os.path.join(gold, "manifest.json")))
    z, zs = np.load(os.path.join(gold, man["file"])), np.load(os.path.join(gold, man["stream"]["file"]))
    H, W, s, st = man["height"], man["width"], man["scale"], man["stream"]
    acc = accel_mod.Accel(max_events=8192, max_rows=s * H + s, max_cols=s * W + s)
    acc.upload_events(zs["b_fr_x"], zs["b_fr_y"], zs["b_t"])
    acc.set_cloud(s, H, W)
    acc.set_model(accel_mod.Model(**man["final_model"]))
    o = acc.default_opts()
    o.res_x, o.res_y, o.trace_cap = H, W, 64
    rc, m, info = acc.run(o)
    assert rc == st["warm_rc"] and abs(info.iterations - st["warm_iterations"]) <= 1
    tr = acc.get_trace(64)
    gt = zs["warm_trajectory"]
    for k in range(min(3, len(tr), len(gt))):
        got = np.array([tr[k].model.total_dx, tr[k].model.total_dy, tr[k].model.total_rot, tr[k].model.total_div])
        np.testing.assert_allclose(got, gt[k, :4], rtol=1e-6, atol=1e-9)
    fm = st["warm_final_model"]
    for f in ("total_dx", "total_dy"):
        assert abs(getattr(m, f) - fm[f]) <= max(1e-4 * abs(fm[f]), 2e-5), (f, getattr(m, f), fm[f])
    acc.upload_events(z["fr_x"], z["fr_y"], z["t"])
    acc.set_cloud(s, H, W)
    for img, want in ((acc.color_time_img(s, H, W, show_final=True), zs["color_raw"]),):
        assert np.array_equal(img.any(axis=2), want.any(axis=2))
        d = np.abs(img.astype(np.int64) - want.astype(np.int64)).max(axis=2)
        assert (d[want.any(axis=2)] > 0).mean() < 0.01 and d.max() <= 9
    acc.project_4param_reinit(*man["warps"][2])
    img, want = acc.color_time_img(s, H, W), zs["color_warp2"]
    assert np.array_equal(img.any(axis=2), want.any(axis=2))
    d = np.abs(img.astype(np.int64) - want.astype(np.int64)).max(axis=2)
    assert (d[want.any(axis=2)] > 0).mean() < 0.01 and d.max() <= 9
    acc.close()


def _run_mode(accel_mod, sl, H, W, scale, trace_cap=0, warm=None, **options):
    if options.get("binned") == 2 and "fused" not in options:
        options = dict(options, fused=0)   # "binned = 2" names the two-kernel tile-binned loop; the one-kernel loop is asked for by name
    acc = make_accel(accel_mod, options, max_events=max(len(sl["t"]), 8192), max_rows=scale * H + scale,
                     max_cols=scale * W + scale)   # ("debug_margin": tests/helpers.py)
    acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
    acc.set_cloud(scale, H, W)
    if warm is not None:
        acc.set_model(warm)
    opts = acc.default_opts()
    opts.res_x, opts.res_y, opts.trace_cap, opts.want_uv = H, W, trace_cap, 1
    rc, m, info = acc.run(opts)
    tr = acc.get_trace(trace_cap) if trace_cap else []
    out = acc.writeout_events() + acc.compute_uv()
    tim, cnt = acc.get_time_img()
    acc.close()
    return rc, m, info, tr, out, tim, cnt


@pytest.mark.parametrize("scale", [1, 3])
def test_binned_scatter_is_bit_identical_to_global_atomics(accel_mod, scale):
    """The tile-binned LDS scatter and the one-global-atomic-per-event scatter accumulate the
    same integers, so whole trajectories must agree bit for bit -- also when a tiny margin
    forces the overflow path and several re-bins."""
    H, W = 180, 240
    sl = synth.make_slice(60000, H, W, 0.05, seed=17)
    ref = _run_mode(accel_mod, sl, H, W, scale, trace_cap=256, binned=0)
    assert ref[2].rebins == 0
    # (fused = 2: the one-kernel iteration, k_fused_pass -- default margin, margins so small that events outrun their
    # bins and passes are repeated on fresh ones, the unpacked LDS planes)
    for opts in (dict(binned=2), dict(binned=2, debug_margin=2), dict(binned=2, debug_margin=4), dict(binned=2, debug_margin=6),
                 dict(binned=2, co_schedule=1, debug_margin=2),
                 dict(fused=2), dict(fused=2, debug_margin=1), dict(fused=2, debug_margin=2, bin_predict=0),
                 dict(fused=2, bin_pack_limit=20)):
        got = _run_mode(accel_mod, sl, H, W, scale, trace_cap=256, **opts)
        assert got[0] == ref[0] and got[2].iterations == ref[2].iterations, opts
        assert got[2].rebins >= 1
        assert got[1].as_dict() == ref[1].as_dict(), opts
        for a, b in zip(got[3], ref[3]):
            assert a.model.as_dict() == b.model.as_dict(), opts
        for a, b in zip(got[4], ref[4]):          # pr_x, pr_y, nx, ny, u, v in upload order
            assert np.array_equal(a, b), opts
        assert np.array_equal(got[5], ref[5]) and np.array_equal(got[6], ref[6]), opts
    # without the drift prediction the re-bin is triggered by observed overflow only: the exact
    # global-atomic overflow path must give the same bits
    tiny = _run_mode(accel_mod, sl, H, W, scale, trace_cap=256, binned=2, debug_margin=2, bin_predict=0)
    assert tiny[2].overflow_events > 0 and tiny[2].rebins > 1, "margin 2 must exercise overflow + re-bin"
    assert tiny[1].as_dict() == ref[1].as_dict() and tiny[2].iterations == ref[2].iterations
    for a, b in zip(tiny[4], ref[4]):
        assert np.array_equal(a, b)


def test_overflow_path_flags_and_sparse_clearing(accel_mod):
    """The overflow path of the tile-binned loop on a large image: every pixel an overflow event touches is flagged in a
    bitmap, the stencil kernel reads the overflow planes only around flagged pixels and clears only those afterwards.  With
    a 2-pixel margin and no prediction thousands of events overflow in every iteration, for dozens of iterations in a row
    (both plane buffers in turn), with the update at the scatter head and in the stencil tail, dense slabs / own pixels +
    margin plane / event lists; the bits must be those of the global-atomic loop.  The context is used for a stand-alone
    time image first (an operator that leaves an overflow plane buffer dirty WITHOUT flags: the first launch clears it whole),
    and for a second run afterwards (what the first run left flagged)."""
    H, W, s = 480, 640, 3
    sl = synth.make_slice(300000, H, W, 0.03, seed=41)
    other = synth.make_slice(200000, H, W, 0.03, seed=42)

    def chain(opts):
        a = make_accel(accel_mod, opts, max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
        out = []
        a.upload_events(other["fr_x"], other["fr_y"], other["t"])
        a.set_cloud(s, H, W)
        out.append(tuple(np.ascontiguousarray(x).tobytes() for x in a.get_time_img()))   # dirties a plane buffer, no flags
        infos = []
        for data, max_iter in ((sl, 45), (other, 30), (sl, 31)):
            a.upload_events(data["fr_x"], data["fr_y"], data["t"])
            a.set_cloud(s, H, W)
            o = a.default_opts()
            o.res_x, o.res_y, o.want_uv, o.trace_cap, o.max_iter = H, W, 1, 64, max_iter
            rc, m, info = a.run(o)
            u, v = a.compute_uv()
            out.append((rc, info.iterations, m.as_dict(), [t.model.as_dict() for t in a.get_trace(64)], u.tobytes(), v.tobytes(),
                        tuple(np.ascontiguousarray(x).tobytes() for x in a.get_time_img())))
            infos.append(info)
        a.close()
        return out, infos

    ref, _ = chain({"binned": 0, "fused": 0})
    for opts in ({"bin_compact": 0, "bin_split": 0}, {"bin_compact": 0, "bin_split": 2}, {"bin_compact": 2},
                 {"bin_compact": 0, "bin_split": 0, "co_schedule": 1}, {"bin_compact": 0, "bin_split": 2, "co_schedule": 1}):
        got, infos = chain(dict({"binned": 2, "fused": 0, "debug_margin": 2, "bin_predict": 0}, **opts))
        assert all(i.overflow_events > 20 * i.iterations for i in infos), (opts, [i.overflow_events for i in infos])
        assert got == ref, opts


def test_binned_warm_start_bit_identical(accel_mod):
    H, W = 180, 240
    a = synth.make_slice(40000, H, W, 0.05, seed=31)
    b = synth.make_slice(40000, H, W, 0.05, seed=32)
    cold = _run_mode(accel_mod, a, H, W, 3, binned=0)
    w0 = _run_mode(accel_mod, b, H, W, 3, warm=cold[1], binned=0)
    for opts in (dict(binned=2), dict(fused=2), dict(fused=2, debug_margin=1)):
        w1 = _run_mode(accel_mod, b, H, W, 3, warm=cold[1], **opts)
        assert w0[2].iterations == w1[2].iterations and w0[1].as_dict() == w1[1].as_dict(), opts
        for x, y in zip(w0[4], w1[4]):
            assert np.array_equal(x, y), opts


def test_tile_grid_matches_per_tile_oracle(oracle_lib, accel_mod):
    """BASELINE config 4 in small: an 8 x 8 grid of independent optimizers over one slice; every
    tile must behave like its own OptimizerRolling (oracle run on the tile's events)."""
    H, W, s, G = 128, 160, 3, 8
    sl = synth.make_slice(120000, H, W, 0.020, seed=41, velocity=(-60.0, 110.0))
    acc = accel_mod.Accel(max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
    acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
    guard = (H // G, W // G)
    models, infos = acc.run_tiles(G, G, s, (H, W), guard, min_events=200, hard_iter_cap=5000)
    u, v = acc.compute_uv()
    tr = np.minimum(sl["fr_x"].astype(np.int64) * G // H, G - 1)
    tc = np.minimum(sl["fr_y"].astype(np.int64) * G // W, G - 1)
    tid = tr * G + tc
    ran = skipped = 0
    worst = [0.0, 0.0]
    for k in range(G * G):
        sel = np.nonzero(tid == k)[0]
        oc = oracle_lib.Cloud(sl["fr_x"][sel], sl["fr_y"][sel], sl["t"][sel])
        ow = oc.set_cloud(s, H, W)
        om = oracle_lib.Model()
        orc, oloop, _ = oc.run(ow, om, res_x=guard[0], res_y=guard[1], min_events=200, hard_cap=5000)
        orc = accel_mod.BF_ERR_NOCONV if orc < 0 else orc
        assert infos[k].rc == orc, (k, infos[k].rc, orc)
        if orc == 1:
            skipped += 1
            assert not u[sel].any() and not v[sel].any()
            continue
        if orc != 0:
            continue
        ran += 1
        assert abs(infos[k].iterations - oloop.itercount) <= 1, (k, infos[k].iterations, oloop.itercount)
        ou, ov = oc.compute_uv()
        worst[0 if infos[k].iterations == oloop.itercount else 1] = max(
            worst[0 if infos[k].iterations == oloop.itercount else 1], np.abs(u[sel] - ou).max(), np.abs(v[sel] - ov).max())
        assert _flow_close(u[sel], ou) and _flow_close(v[sel], ov), (k, infos[k].iterations, oloop.itercount)
    print("tile grid: worst per-event flow deviation %.3e px/s (same iteration count) / %.3e px/s (count off by one)" % tuple(worst))
    assert ran >= G * G // 2, (ran, skipped)
    # the run is repeatable bit for bit (integer accumulators, fixed reduction order)
    acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
    models2, infos2 = acc.run_tiles(G, G, s, (H, W), guard, min_events=200, hard_iter_cap=5000)
    assert [m.as_dict() for m in models] == [m.as_dict() for m in models2]
    acc.close()


def test_streaming_upload_matches_blocking(accel_mod):
    """Config 3 plumbing: the copy-stream upload (two staging slots) must give exactly the
    results of the blocking upload over an STM chain of slices."""
    H, W, s = 180, 240, 3
    sls = [synth.make_slice(30000, H, W, 0.03, seed=60 + i) for i in range(4)]
    nmax = max(len(sl["t"]) for sl in sls)
    acc = accel_mod.Accel(max_events=nmax, max_rows=s * H + s, max_cols=s * W + s)

    def chain(upload):
        prev, out = None, []
        for i, sl in enumerate(sls):
            upload(i, sl)
            acc.set_cloud(s, H, W)
            if prev is not None:
                acc.set_model(prev)
            rc, prev, info = acc.run()
            out.append((rc, info.iterations, prev.as_dict()))
        return out

    ref = chain(lambda i, sl: acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"]))
    pin = [[acc.pinned_int32(nmax) for _ in range(3)] for _ in range(2)]

    def put(i):
        sl, k = sls[i], i & 1
        n = len(sl["t"])
        pin[k][0][:n], pin[k][1][:n], pin[k][2][:n] = sl["fr_x"], sl["fr_y"], sl["t"]
        acc.upload_events_async(pin[k][0], pin[k][1], pin[k][2], n)

    put(0)

    def up(i, sl):
        acc.commit_upload()
        if i + 1 < len(sls):
            put(i + 1)

    got = chain(up)
    assert got == ref
    with pytest.raises(accel_mod.BfError):
        acc.commit_upload()                       # nothing pending
    # ... TWO uploads ahead of the slice being solved (the staging kernels of slice i + 1 run on the copy stream, into the slot's own
    # event arrays, while slice i is solved; the commit swaps pointers), plain and with "defer_uploads" (the uploads' HIP calls are
    # issued by bf_run behind its first batch, or by the commit that needs the slot), 12 and 8 bytes per event, and a per-event
    # read-back (which follows the swapped arrays) per slice
    pin4 = [[acc.pinned_int32(nmax) for _ in range(3)] for _ in range(4)]
    pin16 = [[acc.pinned_array(nmax, np.uint16), acc.pinned_array(nmax, np.uint16), acc.pinned_int32(nmax)] for _ in range(4)]
    want_uv = []
    for sl in sls:
        acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
        acc.set_cloud(s, H, W)
        acc.run()
        want_uv.append(tuple(a.tobytes() for a in acc.compute_uv()))
    for defer in (0, 1):
        for pins in (pin4, pin16):
            acc.set_option("defer_uploads", defer)

            def put2(i):
                sl, n = sls[i], len(sls[i]["t"])
                pins[i][0][:n], pins[i][1][:n], pins[i][2][:n] = sl["fr_x"], sl["fr_y"], sl["t"]
                acc.upload_events_async(pins[i][0], pins[i][1], pins[i][2], n)
            put2(0); put2(1)

            def up2(i, sl):
                acc.commit_upload()
                if i + 2 < len(sls):
                    put2(i + 2)
            assert chain(up2) == ref, (defer, pins is pin16)
            # cold slices with a per-event read-back each, same pattern
            put2(0); put2(1)
            for i in range(len(sls)):
                up2(i, sls[i])
                acc.set_cloud(s, H, W)
                acc.run()
                assert tuple(a.tobytes() for a in acc.compute_uv()) == want_uv[i], (defer, i)
    acc.set_option("defer_uploads", 1)
    put2(0)                                        # recorded only ...
    acc.set_option("defer_uploads", 0)             # ... goes out when the mode ends
    acc.wait_uploads()
    acc.commit_upload()
    acc.set_cloud(s, H, W)
    rc, m0, info0 = acc.run()
    assert (rc, info0.iterations) == ref[0][:2]
    acc.close()


def test_staging_slots_are_not_overwritten_early(accel_mod):
    """Two uploads in flight, then a third into the slot of the first as soon as that one is committed (legal: "at most
    two pending"): the third copy must wait until the staging kernel of the first has read the slot.  Large slices and a
    busy compute stream make the window wide.  A blocking upload while asynchronous ones are pending is refused."""
    H, W, s = 260, 346, 3
    sls = [synth.make_slice(600000, H, W, 0.03, seed=90 + i) for i in range(4)]
    nmax = max(len(sl["t"]) for sl in sls)
    acc = accel_mod.Accel(max_events=nmax, max_rows=s * H + s, max_cols=s * W + s)
    o = acc.default_opts()
    o.res_x, o.res_y, o.max_iter = H, W, 12

    def solve():
        acc.set_cloud(s, H, W)
        rc, m, info = acc.run(o)
        return rc, info.iterations, m.as_dict()

    ref = []
    for sl in sls:
        acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
        ref.append(solve())
    pin = [[acc.pinned_int32(nmax) for _ in range(3)] for _ in range(3)]

    def put(i, k):
        sl = sls[i]
        n = len(sl["t"])
        pin[k][0][:n], pin[k][1][:n], pin[k][2][:n] = sl["fr_x"], sl["fr_y"], sl["t"]
        acc.upload_events_async(pin[k][0], pin[k][1], pin[k][2], n)

    got = []
    put(0, 0); put(1, 1)
    with pytest.raises(accel_mod.BfError):
        acc.upload_events(sls[0]["fr_x"], sls[0]["fr_y"], sls[0]["t"])
    acc.commit_upload()            # slice 0 (slot 0): its staging kernel is only enqueued
    put(2, 2)                      # slot 0 again, straight away
    got.append(solve())
    acc.commit_upload()            # slice 1
    put(3, 0)
    got.append(solve())
    acc.commit_upload(); got.append(solve())
    acc.commit_upload(); got.append(solve())
    assert got == ref
    acc.close()


def test_full_size_config2(oracle_lib, accel_mod):
    """BASELINE config 2 at full size (1M events, 346x260, scale 3): event-count image bit-exact,
    time image within 1e-6, first iterations of the loop on the oracle's trajectory, binned and
    global-atomic scatters bit-identical, runs repeatable."""
    H, W, s = 260, 346, 3
    sl = synth.make_slice(1000000, H, W, 0.030, seed=1)
    oc, ow, acc, gw = make_pair(oracle_lib, accel_mod, sl, s)
    for prm in ((0.0, 0.0, 0.0, 0.0, 0.0, 0.0), (0.27, -0.55, 129.0, 172.0, 2.0e-4, 2.5e-5)):
        oc.project_4param_reinit(*prm)
        acc.project_4param_reinit(*prm)
        otime, ocnt = oc.get_time_img(ow)
        gtime, gcnt = acc.get_time_img()
        assert np.array_equal(gcnt, ocnt.astype(np.uint32))
        one = ocnt == 1.0
        assert np.array_equal(gtime[one], otime[one])
        np.testing.assert_allclose(gtime, otime, rtol=1e-6, atol=0)
        assert int(gcnt.max()) > 20 and int(gcnt.sum()) > 8000000
    acc.close()
    K = 12
    oc2 = oracle_lib.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
    ow2 = oc2.set_cloud(s, H, W)
    om = oracle_lib.Model()
    orc, oloop, otr = oc2.run(ow2, om, max_iter=K, res_x=H, res_y=W, trace_cap=K + 1)
    runs = {}
    for name, opts in (("binned", dict(binned=2, fused=0)), ("atomics", dict(binned=0)), ("binned2", dict(binned=2, fused=0)), ("fused", dict(fused=2)),
                       ("tail_update", dict(binned=2, co_schedule=1)), ("compact", dict(binned=2, bin_compact=2)),
                       ("sep_update", dict(binned=2, co_schedule=1, sep_update=2)), ("compact_co", dict(binned=2, bin_compact=2, co_schedule=1)),
                       ("dense", dict(binned=2, bin_compact=0))):
        a2 = accel_mod.Accel(max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
        for k, v in opts.items():
            a2.set_option(k, v)
        a2.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
        a2.set_cloud(s, H, W)
        o = a2.default_opts()
        o.res_x, o.res_y, o.max_iter, o.trace_cap = H, W, K, K + 1
        rc, m, info = a2.run(o)
        runs[name] = (rc, info.iterations, m.as_dict(), [t_.model.as_dict() for t_ in a2.get_trace(K + 1)], a2.compute_uv())
        a2.close()
    assert runs["binned"][:4] == runs["atomics"][:4] == runs["binned2"][:4] == runs["tail_update"][:4] == runs["compact"][:4] == runs["dense"][:4] == runs["fused"][:4]
    assert np.array_equal(runs["binned"][4][0], runs["atomics"][4][0])
    assert np.array_equal(runs["binned"][4][0], runs["fused"][4][0]) and np.array_equal(runs["binned"][4][1], runs["fused"][4][1])
    assert runs["binned"][1] == oloop.itercount == K + 1
    for k in range(K + 1):
        g, o_ = runs["binned"][3][k], otr[k].model
        assert g["cnt"] == o_.cnt, k
        for f in ("dx", "dy", "rot", "div", "total_dx", "total_dy", "total_rot", "total_div"):
            assert abs(g[f] - getattr(o_, f)) <= 1e-6 * max(abs(getattr(o_, f)), 1e-4), (k, f, g[f], getattr(o_, f))


def test_full_size_properties(accel_mod):
    """Size-independent properties at BASELINE config 2's full size (1M events, cold run to convergence, no oracle):
    * order invariance -- any permutation of the events gives the same model, iteration count and per-event flow BIT
      FOR BIT (the accumulators are integers; the reference itself is order-dependent at the 1e-8 level);
    * sensor translation -- the same scene shifted by whole pixels converges to the same flow within the stated
      tolerance, with the centre shifted by the same amount;
    * the flow that was injected into the synthetic scene comes back (median over the events within 0.5 %)."""
    H, W, s = 260, 346, 3
    sl = synth.make_slice(1000000, H, W, 0.030, seed=1)
    n = len(sl["t"])

    def run(fx, fy, t, Hs=H, Ws=W):
        a = accel_mod.Accel(max_events=n, max_rows=s * Hs + s, max_cols=s * Ws + s)
        a.upload_events(fx, fy, t)
        a.set_cloud(s, Hs, Ws)
        o = a.default_opts()
        o.res_x, o.res_y, o.want_uv = Hs, Ws, 1
        rc, m, info = a.run(o)
        u, v = a.compute_uv()
        a.close()
        return rc, info.iterations, m.as_dict(), u, v

    rc0, it0, m0, u0, v0 = run(sl["fr_x"], sl["fr_y"], sl["t"])
    assert rc0 == 0
    perm = np.random.default_rng(7).permutation(n)
    rc1, it1, m1, u1, v1 = run(sl["fr_x"][perm].copy(), sl["fr_y"][perm].copy(), sl["t"][perm].copy())
    assert (rc1, it1) == (rc0, it0) and m1 == m0
    assert np.array_equal(u1, u0[perm]) and np.array_equal(v1, v0[perm])
    # translation by (7, 12) sensor pixels on a correspondingly larger sensor
    rc2, it2, m2, u2, v2 = run(sl["fr_x"] + 7, sl["fr_y"] + 12, sl["t"], H + 7, W + 12)
    assert rc2 == 0
    assert _flow_close(u2, u0) and _flow_close(v2, v0)
    # (the converged model keeps its centre in sensor coordinates, optimizer_rolling.h:345-346)
    assert abs(m2["cx"] - (m0["cx"] + 7)) < 1e-3 and abs(m2["cy"] - (m0["cy"] + 12)) < 1e-3
    # injected flow: (-150 H / 180, 300 W / 240) px/s
    assert abs(np.median(u0) - (-150.0 * H / 180.0)) < 0.005 * 150.0 * H / 180.0
    assert abs(np.median(v0) - (300.0 * W / 240.0)) < 0.005 * 300.0 * W / 240.0


def test_concurrent_contexts_match_sequential(accel_mod):
    """Several slice contexts in flight on one GPU (bench.py --concurrent): same results as one
    after the other."""
    import threading
    H, W, s = 180, 240, 3
    sls = [synth.make_slice(50000, H, W, 0.04, seed=80 + i) for i in range(4)]

    def solve(sl, out, k):
        a = accel_mod.Accel(max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
        a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
        a.set_cloud(s, H, W)
        rc, m, info = a.run()
        out[k] = (rc, info.iterations, m.as_dict(), a.compute_uv()[0].sum())
        a.close()

    seq, par = {}, {}
    for k, sl in enumerate(sls):
        solve(sl, seq, k)
    th = [threading.Thread(target=solve, args=(sl, par, k)) for k, sl in enumerate(sls)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    assert par == seq


# ---- OptimizerLocal: the contrast-score optimiser (optimizer_sampler.cpp) ----

def test_local_score_bit_exact(oracle_lib, accel_mod):
    """Saturating count image, this build's 8-bit Gaussian and the non-zero mean: integer work, bit-exact
    against the oracle for every supported scale, both constructors, a saturating cloud and shifted (nx, ny)."""
    H, W = 180, 240
    sl = synth.make_slice(60000, H, W, 0.05, seed=31)
    for s in (1, 3, 5, 7):
        oc = oracle_lib.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
        acc = accel_mod.Accel(max_events=len(sl["t"]), max_rows=7 * H + 7, max_cols=7 * W + 7)
        acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
        for center, wsz in ((None, 0), ((70, 100, 20000000), 60)):
            ow = oc.local_window(s, center=center, wsz=wsz)
            gw = acc.local_set_window(s, center=center, wsz=wsz)
            for f in ("metric_wsizex", "metric_wsizey", "scale_img_x", "scale_img_y", "c_fr_x", "c_fr_y", "c_t"):
                assert getattr(ow, f) == getattr(gw, f), f
            for nx, ny in ((0.0, 0.0), (-0.19, 0.38), (0.7, -1.3), (0.0, 0.0)):
                osc, oimg = oc.local_iteration_step(ow, nx, ny)
                gsc, gimg = acc.local_iteration_step(nx, ny, want_img=True)
                assert np.array_equal(gimg, oimg), (s, center, nx, ny)
                assert gsc == osc
                assert acc.local_iteration_step(nx, ny) == osc   # without the image copy, planes alternate
        acc.close()
    # saturation at 255
    n = 4000
    fx, fy, t = np.full(n, 40, np.int32), np.full(n, 50, np.int32), np.arange(n, dtype=np.int64) * 1000
    oc = oracle_lib.Cloud(fx, fy, t)
    acc = accel_mod.Accel(max_events=n, max_rows=200, max_cols=200)
    acc.upload_events(fx, fy, t)
    ow = oc.local_window(3, center=(40, 50, 0), wsz=20)
    acc.local_set_window(3, center=(40, 50, 0), wsz=20)
    osc, oimg = oc.local_iteration_step(ow, 0.0, 0.0)
    gsc, gimg = acc.local_iteration_step(0.0, 0.0, want_img=True)
    assert oimg.max() > 200 and np.array_equal(gimg, oimg) and gsc == osc
    acc.close()


def test_local_run_same_trajectory(oracle_lib, accel_mod):
    """The scores are exact, so the coordinate descent takes the same decisions: identical (nx, ny), steps,
    final score and evaluation count; guard and state errors as specified."""
    H, W, s = 180, 240, 3
    sl = synth.make_slice(40000, H, W, 0.05, seed=32)
    oc = oracle_lib.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
    ow = oc.local_window(s)
    orc, ost, _ = oc.local_run(ow, res_x=H, res_y=W)
    acc = accel_mod.Accel(max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
    with pytest.raises(accel_mod.BfError):
        acc.local_set_window(s)                      # nothing uploaded
    acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
    with pytest.raises(accel_mod.BfError):
        acc.local_run(H, W)                          # no window yet
    acc.local_set_window(s)
    grc, gst = acc.local_run(H, W)
    assert (grc, gst.evaluations) == (orc, ost.evaluations) and gst.evaluations > 10
    for f in ("nx", "ny", "last_score", "dnx", "dny", "dn_th"):
        assert getattr(gst, f) == getattr(ost, f), f
    # the rolling optimizer is undisturbed by local evaluations in between
    acc.set_cloud(s, H, W)
    rc1, m1, i1 = acc.run()
    acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
    acc.set_cloud(s, H, W)
    acc.local_set_window(s)
    acc.local_iteration_step(0.1, 0.1)
    rc2, m2, i2 = acc.run()
    assert (rc1, i1.iterations, m1.as_dict()) == (rc2, i2.iterations, m2.as_dict())
    # window guard (optimizer_sampler.cpp:9-13)
    acc.upload_events(np.array([5, 6], np.int32), np.array([5, 7], np.int32), np.array([0, 10], np.int64))
    acc.local_set_window(3)
    rc, st = acc.local_run(H, W)
    assert rc == accel_mod.BF_SKIPPED and st.evaluations == 0
    with pytest.raises(accel_mod.BfError):
        acc.local_set_window(9)                      # the Gaussian is stated up to 7
    acc.close()


def test_randomised_differential_fuzz():
    """scripts/fuzz_parity.py, a short deterministic run: random sensors / scales / clouds (empty, tiny, hot spots,
    negative times, rotation + divergence) -- event-count image bit-exact against the oracle, all scatter and loop modes
    (global atomics, tile-binned with other tiles / margins, forced overflow path, single-launch loop) bit-identical."""
    import subprocess
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([_sys.executable, os.path.join(root, "scripts", "fuzz_parity.py"), "24", "5"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode()
    assert r.returncode == 0 and "fuzz: 24 cases, 0 problems" in out, out[-3000:]


def test_projection_img_bit_exact(oracle_lib, accel_mod):
    """EventFile::projection_img on the device: raw and motion-compensated 8-bit event image, bit-exact against the
    oracle after identical warps (scales 1-7, with a noise mask), and sharper after compensation."""
    H, W = 180, 240
    sl = synth.make_slice(50000, H, W, 0.05, seed=42)
    noise = (np.arange(len(sl["t"])) % 11 == 0).astype(np.uint8)
    for s in (1, 3, 5, 7):
        for nz in (None, noise):
            oc, ow, acc, gw = make_pair(oracle_lib, accel_mod, sl, 3, noise=nz)
            if s > 3:   # the ctx of make_pair is sized for scale 3: a bigger image needs a bigger ctx
                acc.close()
                acc = accel_mod.Accel(max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
                acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"], nz)
                acc.set_cloud(3, H, W)
            assert np.array_equal(acc.projection_img(s, H, W, show_final=True), oc.projection_img(s, H, W, show_final=True))
            for prm in ((0.15, -0.3, 90.0, 120.0, 1e-4, 2e-5), (0.19, -0.38, 90.0, 120.0, 0.0, 0.0)):
                oc.project_4param_reinit(*prm)
                acc.project_4param_reinit(*prm)
                assert np.array_equal(acc.projection_img(s, H, W), oc.projection_img(s, H, W)), (s, prm)
            acc.close()
    # after a converged run the compensated image is sharper: fewer lit pixels than the raw one
    acc = accel_mod.Accel(max_events=len(sl["t"]), max_rows=3 * H + 3, max_cols=3 * W + 3)
    acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
    acc.set_cloud(3, H, W)
    o = acc.default_opts()
    o.res_x, o.res_y = H, W
    acc.run(o)
    raw, comp = acc.projection_img(3, H, W, show_final=True), acc.projection_img(3, H, W)
    assert (comp > 0).sum() < 0.6 * (raw > 0).sum()
    acc.close()


def test_color_time_img(oracle_lib, accel_mod):
    """EventFile::color_time_img on the device against the oracle (raw and after identical warps, with a noise mask,
    odd and even scales): the covered-pixel mask is exact; hue / saturation come from order-free fixed-point sums on the
    device and f32 running sums in the oracle, so only pixels on an 8-bit truncation boundary may differ -- less than
    1 % of them, by at most one hue step (<= 9 grey levels per channel).  The device image does not depend on the
    order of the events (bit-identical for the reversed slice)."""
    H, W = 180, 240
    sl = synth.make_slice(50000, H, W, 0.05, seed=44)
    noise = (np.arange(len(sl["t"])) % 13 == 0).astype(np.uint8)

    def check(a, b, tag):
        assert a.shape == b.shape, tag
        assert np.array_equal(a.any(axis=2), b.any(axis=2)), tag
        d = np.abs(a.astype(np.int64) - b.astype(np.int64)).max(axis=2)
        lit = a.any(axis=2)
        assert (d[lit] > 0).mean() < 0.01 and d.max() <= 9, (tag, float((d[lit] > 0).mean()), int(d.max()))

    for s in (1, 3, 4):
        for nz in (None, noise):
            oc, ow, acc, gw = make_pair(oracle_lib, accel_mod, sl, 3, noise=nz)
            if s > 3:
                acc.close()
                acc = accel_mod.Accel(max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
                acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"], nz)
                acc.set_cloud(3, H, W)
            check(acc.color_time_img(s, H, W, show_final=True), oc.color_time_img(s, H, W, show_final=True), (s, "raw"))
            for prm in ((0.15, -0.3, 90.0, 120.0, 1e-4, 2e-5), (0.19, -0.38, 90.0, 120.0, 0.0, 0.0)):
                oc.project_4param_reinit(*prm)
                acc.project_4param_reinit(*prm)
                check(acc.color_time_img(s, H, W), oc.color_time_img(s, H, W), (s, prm))
            acc.close()
    # order independence
    acc = accel_mod.Accel(max_events=len(sl["t"]), max_rows=3 * H + 3, max_cols=3 * W + 3)
    acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
    fwd = acc.color_time_img(3, H, W, show_final=True)
    acc.upload_events(sl["fr_x"][::-1].copy(), sl["fr_y"][::-1].copy(), sl["t"][::-1].copy())
    assert np.array_equal(acc.color_time_img(3, H, W, show_final=True), fwd)
    # scale 0 is the reference's default of 11 (needs a larger image than this context holds)
    with pytest.raises(Exception):
        acc.color_time_img(0, H, W)
    acc.close()
    # all events at one instant: phase 0 everywhere (the reference divides 0 by 0)
    acc = accel_mod.Accel(max_events=1000, max_rows=3 * 20 + 3, max_cols=3 * 20 + 3)
    fx = np.arange(5, 15, dtype=np.int32)
    acc.upload_events(fx, fx, np.full(10, 777, dtype=np.int32))
    img = acc.color_time_img(3, 20, 20, show_final=True)
    oc = oracle_lib.Cloud(fx, fx, np.full(10, 777, dtype=np.int64))
    assert np.array_equal(img, oc.color_time_img(3, 20, 20, show_final=True))
    acc.close()


def test_context_reuse_fuzz():
    """scripts/fuzz_reuse.py, a short deterministic run: a long-lived context executing a random mix of uploads (plain,
    asynchronous, ring), windows, warps, images, cold / warm runs in every scatter / loop mode, tile grids, contrast-score
    evaluations and projection images gives, after every observable operation, the bits a fresh context gives for the
    operations since the last upload (no stale state).  Seed 5 is the sequence that exposed a single-launch loop running
    on stale bins after bf_run_tiles."""
    import subprocess
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([_sys.executable, os.path.join(root, "scripts", "fuzz_reuse.py"), "80", "5"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode()
    assert r.returncode == 0 and "0 mismatches" in out.splitlines()[-1], out[-3000:]


def test_event_lists_full_bins_and_second_pass(accel_mod):
    """The event-list form of the tile-binned scatter at its edges: a bin with more events than its list holds sends the
    surplus down the exact overflow path; a bin with more events than one pass of the work-group covers recomputes its
    entries from the stored products; empty bins write empty lists.  Every case must give the bits of the global-atomic
    scatter."""
    for (n, H, W, scale, opts, want_overflow) in (
            (60000, 60, 80, 1, dict(), True),      # a 61 x 81 image: a dozen bins of thousands of events, lists of <= 48 x 32 entries
            (60000, 120, 160, 3, dict(), False),   # several passes per bin
            (1500, 180, 240, 5, dict(), False)):                                                        # mostly empty bins
        sl = synth.make_slice(n, H, W, 0.05, seed=23)
        ref = _run_mode(accel_mod, sl, H, W, scale, trace_cap=64, binned=0)
        got = _run_mode(accel_mod, sl, H, W, scale, trace_cap=64, binned=2, bin_compact=2, **opts)
        tail = _run_mode(accel_mod, sl, H, W, scale, trace_cap=64, binned=2, bin_compact=2, co_schedule=1, sep_update=0, **opts)
        sep = _run_mode(accel_mod, sl, H, W, scale, trace_cap=64, binned=2, bin_compact=2, co_schedule=1, **opts)   # (auto: the update as its own kernel)
        for r in (got, tail, sep):
            assert r[0] == ref[0] and r[2].iterations == ref[2].iterations and r[2].rebins >= 1, (n, H, W)
            assert r[1].as_dict() == ref[1].as_dict(), (n, H, W)
            for a, b in zip(r[3], ref[3]):
                assert a.model.as_dict() == b.model.as_dict(), (n, H, W)
            for a, b in zip(r[4], ref[4]):
                assert np.array_equal(a, b), (n, H, W)
            assert np.array_equal(r[5], ref[5]) and np.array_equal(r[6], ref[6]), (n, H, W)
        if want_overflow:
            assert got[2].overflow_events > 0, "the lists must have overflowed"


def test_tile_grid_first_then_a_run_on_a_fresh_context(accel_mod):
    """bf_run_tiles as the FIRST operation after an upload on a fresh context leaves the slice's events in the second
    event set; the window / run that follow must find them there (scripts/fuzz_reuse.py seed 99: the tile-binned set-up
    allocated that set a second time and the run saw 2833 events at pixel (0, 0))."""
    H, W = 180, 240
    sl = small_slice()

    def go(tiles):
        a = accel_mod.Accel(max_events=32768, max_rows=3 * H + 3, max_cols=3 * W + 3)
        a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
        if tiles:
            a.run_tiles(8, 8, 3, (H, W), (H // 8, W // 8), 64, max_iter=5)
        a.set_cloud(1, H, W)
        tim, cnt = a.get_time_img()
        o = a.default_opts()
        o.res_x, o.res_y, o.max_iter, o.min_events = H, W, 3, 50
        rc, m, info = a.run(o)
        out = (rc, info.iterations, m.as_dict(), tim.tobytes(), cnt.tobytes())
        a.close()
        return out

    clean, after_tiles = go(False), go(True)
    assert clean[0] == 0 and clean[1] == 4
    assert clean == after_tiles


def test_moment_accumulators_are_clean_for_the_next_user(accel_mod):
    """The head-update loop consumes, but does not clear, the moment sums of its last iteration.  Whoever uses the
    accumulators next through the stencil kernel's last-work-group form -- a sparse slice on the global-atomic path,
    bf_fast_model -- must not add onto them (found by scripts/fuzz_reuse.py seed 22: a tile-binned run that ends after an
    odd number of iterations, then a 179-event slice on the same context)."""
    A = synth.make_slice(60000, 180, 240, 0.05, seed=17)
    B = synth.make_slice(180, 104, 204, 0.03, seed=42)
    img = np.random.default_rng(3).uniform(0, 0.03, (90, 130)).astype(np.float32)

    def on(acc, sl, scale, max_iter, binned):
        acc.set_option("binned", binned)
        acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
        acc.set_cloud(scale, sl["height"], sl["width"])
        o = acc.default_opts()
        o.res_x, o.res_y, o.max_iter, o.min_events = sl["height"], sl["width"], max_iter, 50
        rc, m, info = acc.run(o)
        return rc, info.iterations, m.as_dict()

    def ctx():
        return accel_mod.Accel(max_events=65536, max_rows=7 * 180 + 7, max_cols=7 * 240 + 7)

    fresh = ctx()
    want_run = on(fresh, B, 7, 10, 0)
    want_fm = fresh.fast_model(img).as_dict()
    fresh.close()
    parities = set()
    for iters in (3, 4):          # the last iteration's sums sit in either parity
        acc = ctx()
        got_a = on(acc, A, 3, iters, 2)
        parities.add(got_a[1] & 1)
        assert on(acc, B, 7, 10, 0) == want_run, iters
        acc.close()
        acc = ctx()
        on(acc, A, 3, iters, 2)
        assert acc.fast_model(img).as_dict() == want_fm, iters
        acc.close()
    assert parities == {0, 1}


def test_readback_order_and_buffer_reuse(oracle_lib, accel_mod):
    """Per-event outputs after a tile-binned run are produced in tile-sorted order and un-permuted on read-back; the
    position read-back borrows the flow buffer.  Every read-back order must give the same, correctly ordered, data."""
    H, W, s = 180, 240, 3
    sl = synth.make_slice(40000, H, W, 0.04, seed=77)
    oc, ow, acc, gw = make_pair(oracle_lib, accel_mod, sl, s)
    o = acc.default_opts()
    o.res_x, o.res_y, o.want_uv = H, W, 1
    acc.run(o)
    u1, v1 = acc.compute_uv()
    pr_x, pr_y, nx, ny = acc.writeout_events()          # uses the flow buffer as staging for pr
    u2, v2 = acc.compute_uv()
    assert np.array_equal(u1, u2) and np.array_equal(v1, v2)
    # (u, v) == n * 1e5 / 127 element by element (event.h:135-142), i.e. flow and n are in the same (upload) order
    ou, ov = oracle_lib.lib(), None
    uu = np.empty_like(nx); vv = np.empty_like(ny)
    import ctypes as C
    oracle_lib.lib().bfo_compute_uv(nx.ctypes.data_as(C.POINTER(C.c_double)), ny.ctypes.data_as(C.POINTER(C.c_double)),
                                    C.c_int64(len(nx)), uu.ctypes.data_as(C.POINTER(C.c_double)), vv.ctypes.data_as(C.POINTER(C.c_double)))
    assert np.allclose(u1, uu, rtol=1e-12, atol=1e-12) and np.allclose(v1, vv, rtol=1e-12, atol=1e-12)
    # pr belongs to the same event as fr: the compensated position moved by n / 127 * t / 1e4 from the sensor position
    exp_x = sl["fr_x"] - (nx / 127.0) * sl["t"] / 1e4
    assert np.max(np.abs(pr_x - exp_x)) < 1e-3
    # same data when the order of read-backs is different, and without the fused flow
    acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"]); acc.set_cloud(s, H, W)
    o.want_uv = 0
    acc.run(o)
    pr_x3, pr_y3, nx3, ny3 = acc.writeout_events()
    u3, v3 = acc.compute_uv()
    assert np.array_equal(nx3, nx) and np.array_equal(pr_x3, pr_x) and np.array_equal(u3, u1) and np.array_equal(v3, v1)
    acc.close()


def test_large_slices_stay_on_the_binned_path(accel_mod):
    """Slices beyond ~1.05M events x 30 ms no longer fit a slice-wide packed accumulator; the tile-binned scatter packs
    per bin (fields sized from the fullest bin) and must stay bit-identical to the global-atomic split path -- also
    when the packing is forced not to fit and every event takes the overflow path."""
    H, W, s = 260, 346, 3
    sl = synth.make_slice(1600000, H, W, 0.030, seed=3)
    runs = {}
    for name, opts in (("binned", {}), ("atomics", {"binned": 0}), ("fallback", {"bin_pack_limit": 8})):
        a = accel_mod.Accel(max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
        for k, v in opts.items():
            a.set_option(k, v)
        a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
        a.set_cloud(s, H, W)
        o = a.default_opts()
        o.res_x, o.res_y, o.max_iter, o.trace_cap = H, W, 6, 8
        rc, m, info = a.run(o)
        runs[name] = (rc, info.iterations, m.as_dict(), [t_.model.as_dict() for t_ in a.get_trace(8)], a.compute_uv()[0].tobytes())
        if name == "binned":
            assert info.overflow_events == 0 and info.rebins >= 1
        if name == "fallback":
            assert info.overflow_events > len(sl["t"])      # every event, every iteration
        a.close()
    assert runs["binned"] == runs["atomics"] == runs["fallback"]
    assert runs["binned"][2]["cnt"] > 500000


def test_run_to_convergence_with_rotation_and_divergence(oracle_lib, accel_mod):
    """Scenes with rotation and divergence about the sensor centre, cold start to the loop's own termination: same
    iteration count as the oracle (+-1) and per-event flow within the SURVEY 8(d) bar (1e-4 relative or 0.02 px/s)."""
    rng = np.random.default_rng(7)
    H, W, s = 180, 240, 3

    def scene(n, T, v, rot, div):
        npts = n // 16
        pr, pc = rng.uniform(10, H - 10, npts), rng.uniform(10, W - 10, npts)
        t = np.sort(rng.uniform(0, T, n))
        pick = rng.integers(0, npts, n)
        r0, c0 = pr[pick] - H / 2, pc[pick] - W / 2
        row = pr[pick] + (v[0] + div * r0 - rot * c0) * t
        col = pc[pick] + (v[1] + div * c0 + rot * r0) * t
        keep = (row >= 0) & (row < H) & (col >= 0) & (col < W)
        return dict(fr_x=np.floor(row[keep]).astype(np.int32), fr_y=np.floor(col[keep]).astype(np.int32),
                    t=(t[keep] * 1e9).astype(np.int64), height=H, width=W)

    for v, rot, div in (((-80.0, 120.0), 3.0, 0.0), ((60.0, -40.0), 0.0, 2.5), ((-50.0, 90.0), -2.0, 1.5)):
        sl = scene(100000, 0.03, v, rot, div)
        oc, ow, acc, gw = make_pair(oracle_lib, accel_mod, sl, s)
        om = oracle_lib.Model()
        orc, oloop, _ = oc.run(ow, om, res_x=H, res_y=W)
        o = acc.default_opts()
        o.res_x, o.res_y = H, W
        rc, m, info = acc.run(o)
        assert rc == orc == 0 and abs(info.iterations - oloop.itercount) <= 1
        u, vv = acc.compute_uv()
        ou, ov = oc.compute_uv()
        assert _flow_close(u, ou) and _flow_close(vv, ov), (v, rot, div)
        if rot:
            assert abs(m.total_rot - om.total_rot) <= 1e-6 * abs(om.total_rot) + 1e-9 and abs(m.total_rot) > 1e-4
        if div:
            assert abs(m.total_div - om.total_div) <= 1e-6 * abs(om.total_div) + 1e-9 and abs(m.total_div) > 1e-4
        acc.close()


def test_ring16_hand_off_with_noise_and_flow_ring(oracle_lib, accel_mod):
    """bf_upload_ring16_async (16-bit addresses, absolute 64-bit timestamps, an Event::noise ring, a slice that wraps
    around the end of the ring) + bf_compute_uv_ring: event-count image bit-exact against the oracle with the same noise
    mask (accel_lib.h:152), run and per-event flow identical to the blocking int32 upload of the same slice, and the
    (u, v) pairs land at their ring positions."""
    H, W, s = 180, 240, 3
    sl = synth.make_slice(30000, H, W, 0.04, seed=123)
    n = len(sl["t"])
    cap, first, t0 = n + 5000, n + 5000 - 7000, 5_000_000_000      # the slice occupies [first, cap) and [0, n - 7000)
    idx = (first + np.arange(n)) % cap
    ring_row, ring_col = np.zeros(cap, np.uint16), np.zeros(cap, np.uint16)
    ring_ts, ring_noise = np.zeros(cap, np.uint64), np.zeros(cap, np.uint8)
    noise = (np.arange(n) % 7 == 0).astype(np.uint8)
    ring_row[idx], ring_col[idx] = sl["fr_x"], sl["fr_y"]
    ring_ts[idx] = sl["t"].astype(np.uint64) + np.uint64(t0)
    ring_noise[idx] = noise
    ring_noise[(first - 100) % cap] = 1                            # outside the slice: must not matter
    oc = oracle_lib.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
    oc.noise[:] = noise
    ow = oc.set_cloud(s, H, W)
    _, ocnt = oc.get_time_img(ow)

    def solve(acc):
        acc.set_cloud(s, H, W)
        _, cnt = acc.get_time_img(want_time=False)
        o = acc.default_opts()
        o.res_x, o.res_y, o.want_uv = H, W, 1
        rc, m, info = acc.run(o)
        return cnt, rc, m.as_dict(), info.iterations

    acc = accel_mod.Accel(max_events=n, max_rows=s * H + s, max_cols=s * W + s)
    acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"], noise)
    want = solve(acc)
    wu, wv = acc.compute_uv()
    assert np.array_equal(want[0], ocnt.astype(np.uint32)) and want[1] == 0
    for with_noise in (True, False):
        acc.upload_ring_async(ring_row, ring_col, ring_ts, first, n, t0, ring_noise if with_noise else None)
        acc.commit_upload()
        got = solve(acc)
        if with_noise:
            assert np.array_equal(got[0], want[0]) and got[1:] == want[1:]
            uv = np.full(2 * cap, np.nan)
            acc.compute_uv_ring(uv, first)
            assert np.array_equal(uv[2 * idx], wu) and np.array_equal(uv[2 * idx + 1], wv)
            rest = np.ones(cap, bool)
            rest[idx] = False
            assert np.isnan(uv[0::2][rest]).all()                  # nothing outside the slice's ring positions was written
        else:
            assert int(got[0].sum()) > int(want[0].sum())          # without the mask the flagged events count again
    # the int32 ring form carries the same noise ring
    acc.upload_ring_async(ring_row.astype(np.int32), ring_col.astype(np.int32), ring_ts, first, n, t0, ring_noise)
    acc.commit_upload()
    got = solve(acc)
    assert np.array_equal(got[0], want[0]) and got[1:] == want[1:]
    # 8 bytes per event: the LOW 32 bits of the timestamps (bf_upload_ring16t32_async) -- with slice starts whose low halves sit
    # just below a multiple of 2^32, so that the low halves of the slice's timestamps wrap inside the slice, and one whose
    # events lie partly BEFORE t0 (negative local times exist in the reference too, event.h:61-63)
    for t0b in (t0, (7 << 32) - 12_345_678, (1 << 32) - 1, 3 * (1 << 32) + 5):
        ts_b = np.zeros(cap, np.uint64)
        ts_b[idx] = sl["t"].astype(np.uint64) + np.uint64(t0b)
        acc.upload_ring_async(ring_row, ring_col, (ts_b & np.uint64(0xffffffff)).astype(np.uint32), first, n, t0b, ring_noise,
                              span_ns=int(sl["t"].max()))
        acc.commit_upload()
        got = solve(acc)
        assert np.array_equal(got[0], want[0]) and got[1:] == want[1:], t0b
    shift = 9_000_000                                               # the slice start in the middle of the slice: t in [-9 ms, 31 ms)
    acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"] - shift, noise)
    want_neg = solve(acc)
    acc.upload_ring_async(ring_row, ring_col, (ring_ts & np.uint64(0xffffffff)).astype(np.uint32), first, n, t0 + shift, ring_noise,
                          span_ns=max(shift, int(sl["t"].max()) - shift))
    acc.commit_upload()
    # the contract of the 32-bit form is the caller's to state: no span, or a span of 2^31 ns and more, is refused
    with pytest.raises(ValueError):
        acc.upload_ring_async(ring_row, ring_col, (ring_ts & np.uint64(0xffffffff)).astype(np.uint32), first, n, t0, ring_noise)
    with pytest.raises(accel_mod.BfError):
        acc.upload_ring_async(ring_row, ring_col, (ring_ts & np.uint64(0xffffffff)).astype(np.uint32), first, n, t0, ring_noise, span_ns=1 << 31)
    got = solve(acc)
    assert np.array_equal(got[0], want_neg[0]) and got[1:] == want_neg[1:]
    # ... and a linear slice with 16-bit addresses and its own int32 times (bf_upload_events16_async), both staging slots
    acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
    want_lin = solve(acc)
    x16, y16, t32 = acc.pinned_array(n, np.uint16), acc.pinned_array(n, np.uint16), acc.pinned_array(n, np.int32)
    x16[:], y16[:], t32[:] = sl["fr_x"], sl["fr_y"], sl["t"]
    for _ in range(3):
        acc.upload_events_async(x16, y16, t32, n)
        acc.commit_upload()
        got = solve(acc)
        assert np.array_equal(got[0], want_lin[0]) and got[1:] == want_lin[1:]
    acc.close()


def test_long_batches_are_not_declared_hung(accel_mod):
    """A poll interval of 256 iterations: the watchdog of the progress poll is a wall-clock deadline since the device's last
    progress, so a healthy run whose batches take milliseconds is not declared hung -- and gives the bits of the default
    run, in every loop (one-kernel passes that wait for a re-bin count their progress in launches)."""
    H, W, s = 260, 346, 3
    sl = synth.make_slice(400000, H, W, 0.030, seed=2)

    def go(**opt):
        poll = opt.pop("poll", None)
        acc = make_accel(accel_mod, opt, max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
        acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
        acc.set_cloud(s, H, W)
        o = acc.default_opts()
        o.res_x, o.res_y = H, W
        if poll:
            o.poll_interval = poll
        try:
            rc, m, info = acc.run(o)
            return rc, m.as_dict(), info.iterations
        finally:
            acc.close()

    want = go(binned=2)
    assert want[0] == 0 and want[2] > 100
    assert go(fused=2) == want
    assert go(fused=2, poll=256) == want
    assert go(fused=2, debug_margin=1, poll=64) == want   # (passes that wait for a re-bin: progress is counted in launches)
    assert go(binned=2, poll=256) == want
    assert go(binned=2, poll=256, co_schedule=1) == want
    assert go(binned=2, poll=256, watchdog_ms=2000) == want


def test_run_many_equals_one_by_one(accel_mod):
    """bf_run_many: three independent slices (different sizes, one of them skipped by the 1000-event guard) solved
    together give, slice by slice, the bits of bf_run on a context alone -- return code, iterations, model, flow."""
    H, W, s = 180, 240, 3
    sls = [synth.make_slice(n, H, W, 0.04, seed=70 + k) for k, n in enumerate((60000, 25000, 700))]

    def stage(sl, co):
        a = accel_mod.Accel(max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
        a.set_option("co_schedule", co)
        a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
        a.set_cloud(s, H, W)
        return a

    o = None
    alone = []
    for sl in sls:
        a = stage(sl, 0)
        oo = a.default_opts()
        oo.res_x, oo.res_y, oo.want_uv = H, W, 1
        rc, m, info = a.run(oo)
        alone.append((rc, m.as_dict(), info.iterations) + (a.compute_uv() if rc == 0 else ()))
        o = oo
        a.close()
    accs = [stage(sl, 1) for sl in sls]
    got = accel_mod.run_many(accs, o)
    for (rc, m, info), a, want in zip(got, accs, alone):
        assert (rc, m.as_dict(), info.iterations) == want[:3]
        if rc == 0:
            u, v = a.compute_uv()
            assert np.array_equal(u, want[3]) and np.array_equal(v, want[4])
        a.close()
    assert [w[0] for w in alone] == [0, 0, 1]


def test_device_sincos_against_libm(oracle_lib, accel_mod):
    """The device loops evaluate the sine and cosine of the warp's rotation angle themselves (bf_device_fns.h: sincos_small,
    a polynomial for |x| <= 0.25, the device library beyond; sincos_small_tab in the persistent kernel), where the reference
    calls std::cos / std::sin (event.h:102-103) -- the stand-alone operator bf_project_4param_reinit takes them from the
    host, so the bit-exact warp test above never sees the polynomial.  Here it is evaluated on the device (bf_eval_sincos) for
    1.3 million angles -- dense where a 30 ms slice's rotation lives (|x| <= 2e-3), log-spaced down to 1e-12, sparse up to 0.3
    with the 0.25 switch-over bracketed, a few large ones -- and held against this host's libm (through the oracle library) and
    against the 80-bit long-double value: error in ulps of the true result, and the share of results equal to libm's bits.
    The measured figures are printed and stated in DESIGN.md, "Oracle"."""
    rng = np.random.default_rng(17)
    x = np.concatenate([
        rng.uniform(-2e-3, 2e-3, 1000000),
        np.sign(rng.uniform(-1, 1, 100000)) * 10.0 ** rng.uniform(-12, -2.5, 100000),
        rng.uniform(-0.3, 0.3, 150000),
        0.25 + np.arange(-2000, 2001) * np.spacing(0.25), -0.25 + np.arange(-2000, 2001) * np.spacing(0.25),
        np.array([0.0, -0.0, 0.25, -0.25, 0.2500000000000001, 1.0, -1.0, 3.0, 10.0, 1e3, 1e6, np.pi, np.pi / 2, 5e-324, 1e-300]),
    ])
    assert x.size >= 1000000
    lsn, lcs = oracle_lib.sincos(x)
    xl = x.astype(np.longdouble)
    tsn, tcs = np.sin(xl), np.cos(xl)

    def ulp_err(got, true):
        ulp = np.spacing(np.abs(true.astype(np.float64))).astype(np.longdouble)
        return np.abs(got.astype(np.longdouble) - true) / ulp
    small, poly = np.abs(x) <= 2e-3, np.abs(x) <= 0.25
    acc = accel_mod.Accel(device=0, max_events=4096, max_rows=64, max_cols=64)
    try:
        out = [acc.eval_sincos(x, table=t) for t in (False, True)]
    finally:
        acc.close()
    for table, (dsn, dcs) in zip((False, True), out):
        es, ec = ulp_err(dsn, tsn), ulp_err(dcs, tcs)
        eq_s, eq_c = dsn == lsn, dcs == lcs
        ds, dc = np.abs(dsn.view(np.int64) - lsn.view(np.int64)), np.abs(dcs.view(np.int64) - lcs.view(np.int64))
        print("device sin / cos (%s), %d angles: max error %.3f / %.3f ulp at x = %.17g / %.17g (polynomial range |x| <= 0.25: %.3f / %.3f; "
              "this host's libm: %.3f / %.3f); results equal to libm's bits: %.4f %% / %.4f %% (|x| <= 2e-3: %.4f %% / %.4f %%); "
              "largest distance to libm %d / %d ulp" %
              ("LDS table" if table else "immediates", x.size, es.max(), ec.max(), x[es.argmax()], x[ec.argmax()],
               es[poly].max(), ec[poly].max(), ulp_err(lsn, tsn).max(), ulp_err(lcs, tcs).max(),
               100.0 * eq_s.mean(), 100.0 * eq_c.mean(), 100.0 * eq_s[small].mean(), 100.0 * eq_c[small].mean(), int(ds.max()), int(dc.max())))
        # the polynomial range (every real slice: a 30 ms slice rotates by ~1e-3): within 0.6 ulp of the true value, at most one
        # ulp from libm, and libm's own bits for >= 99 % of the angles a slice can produce
        assert es[poly].max() <= 0.6 and ec[poly].max() <= 0.6, (es[poly].max(), ec[poly].max())
        assert ds[poly].max() <= 1 and dc[poly].max() <= 1
        assert eq_s[small].mean() >= 0.99 and eq_c[small].mean() >= 0.99, (eq_s[small].mean(), eq_c[small].mean())
        # beyond it (a diverged model): the device library's sincos, <= 2 ulp
        assert es.max() <= 2.0 and ec.max() <= 2.0 and ds.max() <= 2 and dc.max() <= 2
        if table:   # the persistent kernel's variant: the same operations in the same order
            assert np.array_equal(dsn, out[0][0]) and np.array_equal(dcs, out[0][1])
