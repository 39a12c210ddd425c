"""CPU tests of the oracle (oracle/bf_oracle.c): cross-check against an independent numpy
restatement, analytic known answers, and the committed golden vectors.

The reference itself cannot be built here (OpenCV + TBB missing) and has no tests, so
these pins are the oracle's own -- "parity unpinned" (oracle/bf_oracle.h)."""
import json
import os

import numpy as np
import pytest

import np_ref
from better_flow_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

PARAMS = [
    (0.0, 0.0, 0.0, 0.0, 0.0, 0.0),
    (0.35, -0.7, 0.0, 0.0, 0.0, 0.0),
    (0.2, 0.4, 20.5, 31.25, 3.0e-4, -2.0e-4),
    (-0.15, 0.05, 22.0, 29.0, -1.0e-3, 1.5e-3),
]


def tiny_slice(n=1500, H=48, W=64, seed=4):
    return synth.make_slice(n, H, W, 0.04, seed=seed, velocity=(-120.0, 260.0))


def test_window_hand_computed(oracle_lib):
    # rows 10..20, cols 5..8, scale 3: wsx = 30, wsy = 9, R = 33, C = 12
    c = oracle_lib.Cloud([10, 20, 15], [5, 8, 6], [0, 1, 2])
    w = c.set_cloud(3, 180, 240)
    assert (w.x_min, w.x_max, w.y_min, w.y_max) == (10, 20, 5, 8)
    assert (w.metric_wsizex, w.metric_wsizey, w.scale_img_x, w.scale_img_y) == (30, 9, 33, 12)
    # x_shift = -((20-10)/2 + 10)*3 + 30/2 + 1 = -45 + 15 + 1 ; y: -((3)/2 + 5)*3 + 4.5 + 1
    assert w.x_shift == -29.0 and w.y_shift == -12.5
    assert np.array_equal(c.pr_x, [10.0, 20.0, 15.0]) and not c.nx.any()
    # empty cloud: min stays at RES, max at 0 (optimizer_rolling.h:252-253)
    e = oracle_lib.Cloud([], [], [])
    we = e.set_cloud(3, 180, 240)
    assert (we.x_min, we.x_max, we.y_min, we.y_max) == (180, 0, 240, 0)


def test_set_local_time(oracle_lib):
    ts = np.array([100, 50, 75, 2**40], dtype=np.uint64)
    assert list(oracle_lib.set_local_time(ts, 75)) == [25, -25, 0, 2**40 - 75]


@pytest.mark.parametrize("scale", [1, 3, 5])
def test_oracle_matches_numpy_restatement(oracle_lib, scale):
    sl = tiny_slice()
    H, W = sl["height"], sl["width"]
    c = oracle_lib.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
    w = c.set_cloud(scale, H, W)
    nw = np_ref.window(sl["fr_x"], sl["fr_y"], scale, H, W)
    assert (w.x_min, w.x_max, w.y_min, w.y_max) == (nw["x_min"], nw["x_max"], nw["y_min"], nw["y_max"])
    assert (w.scale_img_x, w.scale_img_y, w.x_shift, w.y_shift) == (nw["R"], nw["C"], nw["x_shift"], nw["y_shift"])
    pr_x, pr_y = c.pr_x.copy(), c.pr_y.copy()
    for prm in PARAMS:
        c.project_4param_reinit(*prm)
        pr_x, pr_y, nx, ny = np_ref.warp(sl["fr_x"], sl["fr_y"], sl["t"], pr_x, pr_y, *prm)
        assert np.array_equal(pr_x, c.pr_x) and np.array_equal(pr_y, c.pr_y)
        assert np.array_equal(nx, c.nx) and np.array_equal(ny, c.ny)
        timg, cimg = c.get_time_img(w)
        ntime, ncnt = np_ref.time_img(pr_x, pr_y, sl["t"], nw, scale)
        assert np.array_equal(cimg, ncnt) and np.array_equal(timg, ntime)
    gx, gy = oracle_lib.sobel(timg)
    ngx, ngy = np_ref.scharr(timg)
    assert np.array_equal(gx, ngx) and np.array_equal(gy, ngy)
    m = oracle_lib.fast_model(timg)
    nm = np_ref.model(timg)
    assert m.cnt == nm["cnt"] and m.cx == nm["cx"] and m.cy == nm["cy"]
    for k in ("dx", "dy", "rot", "div"):
        assert getattr(m, k) == nm[k], k


def test_warp_analytic(oracle_lib):
    """Pure translation: pr = fr - (n/127) * t / 1e4 and flow u = n * 1e5 / 127 px/s."""
    c = oracle_lib.Cloud([100, 50], [30, 200], [10_000_000, 20_000_000])
    c.set_cloud(3, 180, 240)
    c.project_4param_reinit(1.27, -2.54, 0, 0, 0, 0)
    np.testing.assert_allclose(c.pr_x, [100 - 0.01 * 1000, 50 - 0.01 * 2000], rtol=1e-6)
    np.testing.assert_allclose(c.pr_y, [30 + 0.02 * 1000, 200 + 0.02 * 2000], rtol=1e-6)
    u, v = c.compute_uv()
    np.testing.assert_allclose(u, 1000.0, rtol=1e-12)
    np.testing.assert_allclose(v, -2000.0, rtol=1e-12)
    # zero parameters are the identity on the reset state
    c.set_cloud(3, 180, 240)
    c.project_4param_reinit(0, 0, 0, 0, 0, 0)
    assert np.array_equal(c.pr_x, [100.0, 50.0]) and not c.nx.any()


def test_count_image_properties(oracle_lib):
    sl = tiny_slice(3000)
    H, W = sl["height"], sl["width"]
    for scale in (1, 3):
        c = oracle_lib.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
        w = c.set_cloud(scale, H, W)
        timg, cimg = c.get_time_img(w)
        # zero warp: an event is kept iff hs <= fr*s + (int)shift < wsize + hs (accel_lib.h:154-158)
        hs = scale // 2
        X = sl["fr_x"] * scale + int(w.x_shift)
        Y = sl["fr_y"] * scale + int(w.y_shift)
        keep = (X >= hs) & (X < w.metric_wsizex + hs) & (Y >= hs) & (Y < w.metric_wsizey + hs)
        assert 0.9 * len(X) < keep.sum() < len(X)      # the bbox-max row / column is always lost
        assert not keep[(sl["fr_x"] == w.x_max) | (sl["fr_y"] == w.y_max)].any()
        assert cimg.sum() == keep.sum() * scale * scale
        assert np.all(cimg == np.round(cimg))
        assert timg.max() <= sl["t"].max() / 1e9 * (1 + 1e-6) and timg.min() >= 0
        # noise events are skipped (accel_lib.h:152)
        c.noise[::2] = 1
        _, chalf = c.get_time_img(w)
        assert chalf.sum() == keep[1::2].sum() * scale * scale


def test_scharr_orientation_and_gating(oracle_lib):
    # time increasing along rows: grad_x = -(d/d row) * 32 * pitch, grad_y = 0
    i = np.arange(12, dtype=np.float32)[:, None]
    img = (0.001 + 0.0005 * i) * np.ones((1, 9), np.float32)
    gx, gy = oracle_lib.sobel(img)
    np.testing.assert_allclose(gx[1:-1, 1:-1], -2 * 0.0005 * 16, rtol=1e-4)
    assert np.all(np.abs(gy[1:-1, 1:-1]) < 1e-8)
    assert not gx[0].any() and not gx[-1].any() and not gx[:, 0].any() and not gx[:, -1].any()
    # one invalid tap gates all nine neighbours' gradients to zero
    img2 = img.copy()
    img2[5, 4] = 0.0
    gx2, _ = oracle_lib.sobel(img2)
    assert not gx2[4:7, 3:6].any() and gx2[3, 4] != 0


def test_known_answer_flow_recovery(oracle_lib):
    H, W = 180, 240
    sl = synth.make_slice(20000, H, W, 0.05, seed=3)
    c = oracle_lib.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
    w = c.set_cloud(3, H, W)
    m = oracle_lib.Model()
    rc, loop, _ = c.run(w, m)
    assert rc == 0 and 10 < loop.itercount < 2000
    u, v = c.compute_uv()
    vr, vc = sl["velocity"]
    assert abs(u.mean() - vr) < 0.02 * abs(vr) and abs(v.mean() - vc) < 0.02 * abs(vc)


def test_survey_recorded_reference_behaviour(oracle_lib):
    """Soft pins to the REAL reference.  SURVEY.md section 6 recorded, from the reference's own sources compiled in the
    survey container (its probe driver, section 8(d)'s generator): 200 k events, 346x260, scale 3, cold start -> 128
    iterations, row flow u = -216.9855 px/s (the generator's v_row is -216.67); the following slices of the stream,
    warm-started from the previous model (dvs_flow.h:218-219) -> 6 and 5 iterations.  The survey's PRNG is not given bit
    for bit, so this build's generator draws other points from the same distribution: the oracle has to land in the same
    place, not on the same bits -- iteration count within 20 % of 128, row flow within 0.3 % of the recorded value, warm
    slices within 3 .. 10 iterations.  (Everything else the oracle is held to is its own: see the module docstring.)"""
    H, W, s = 260, 346, 3
    sl = synth.make_slice(200000, H, W, 0.030, seed=1)
    c = oracle_lib.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
    w = c.set_cloud(s, H, W)
    m = oracle_lib.Model()
    rc, loop, _ = c.run(w, m, res_x=H, res_y=W)
    assert rc == 0 and abs(loop.itercount - 128) <= 26, loop.itercount
    u, v = c.compute_uv()
    assert abs(u.mean() - (-216.9855)) < 0.003 * 216.9855, u.mean()
    assert abs(v.mean() - 300.0 * W / 240.0) < 0.003 * 300.0 * W / 240.0, v.mean()
    for seed in (2, 3):   # slices 2 and 3 of the stream: the same motion, warm start
        sl2 = synth.make_slice(200000, H, W, 0.030, seed=seed)
        c2 = oracle_lib.Cloud(sl2["fr_x"], sl2["fr_y"], sl2["t"])
        w2 = c2.set_cloud(s, H, W)
        m = c2.set_model(m)
        rc2, loop2, _ = c2.run(w2, m, res_x=H, res_y=W)
        assert rc2 == 0 and 3 <= loop2.itercount <= 10, loop2.itercount


def test_guards(oracle_lib):
    sl = tiny_slice(600)
    c = oracle_lib.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
    w = c.set_cloud(3, 48, 64)
    assert c.run(w, oracle_lib.Model(), res_x=48, res_y=64)[0] == 1       # < 1000 events
    sl = tiny_slice(2000)
    c = oracle_lib.Cloud(sl["fr_x"] % 3 + 20, sl["fr_y"] % 4 + 20, sl["t"])
    w = c.set_cloud(3, 180, 240)
    assert c.run(w, oracle_lib.Model())[0] == 1 and c.noise.all()           # window too small


def test_max_iter_semantics(oracle_lib):
    sl = synth.make_slice(5000, 90, 120, 0.05, seed=8)
    c = oracle_lib.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
    w = c.set_cloud(3, 90, 120)
    rc, loop, tr = c.run(w, oracle_lib.Model(), max_iter=10, res_x=90, res_y=120, trace_cap=32)
    assert rc == 0 and loop.itercount == 11 and len(tr) == 11   # breaks when itercount > max


def test_golden_vectors(oracle_lib):
    """Committed vectors (tests/golden/make_golden.py): the oracle must keep reproducing them."""
    man = json.load(open(os.path.join(GOLD, "manifest.json")))
    z = np.load(os.path.join(GOLD, man["file"]))
    H, W, s = man["height"], man["width"], man["scale"]
    c = oracle_lib.Cloud(z["fr_x"], z["fr_y"], z["t"])
    w = c.set_cloud(s, H, W)
    assert [w.x_min, w.x_max, w.y_min, w.y_max, w.scale_img_x, w.scale_img_y] == man["window"]
    for k, prm in enumerate(man["warps"]):
        c.project_4param_reinit(*prm)
        timg, cimg = c.get_time_img(w)
        assert np.array_equal(c.pr_x, z["pr_x_%d" % k]) and np.array_equal(c.nx, z["nx_%d" % k])
        assert np.array_equal(cimg.astype(np.uint16), z["cnt_%d" % k])
        assert np.array_equal(timg, z["time_%d" % k])
    gx, gy = oracle_lib.sobel(timg)
    assert np.array_equal(gx, z["gx"]) and np.array_equal(gy, z["gy"])
    m = oracle_lib.fast_model(timg)
    assert [m.cx, m.cy, m.dx, m.dy, m.rot, m.div, m.cnt] == man["model"]
    c2 = oracle_lib.Cloud(z["fr_x"], z["fr_y"], z["t"])
    w2 = c2.set_cloud(s, H, W)
    m2 = oracle_lib.Model()
    rc, loop, tr = c2.run(w2, m2, res_x=H, res_y=W, trace_cap=4096)
    assert loop.itercount == man["iterations"]
    got = np.array([[r.model.total_dx, r.model.total_dy, r.model.total_rot, r.model.total_div,
                     r.loop.x_divider, r.loop.y_divider, r.loop.rot_divider, r.loop.div_divider]
                    for r in tr])
    assert np.array_equal(got, z["trajectory"])


def test_golden_images(oracle_lib):
    """Second fixture: contrast-score images / scores / descent result and the projection images."""
    man = json.load(open(os.path.join(GOLD, "manifest.json")))
    z, zi = np.load(os.path.join(GOLD, man["file"])), np.load(os.path.join(GOLD, man["images_file"]))
    H, W, s, loc = man["height"], man["width"], man["scale"], man["local"]
    c = oracle_lib.Cloud(z["fr_x"], z["fr_y"], z["t"])
    c.set_cloud(s, H, W)
    lw, lw2 = c.local_window(s), c.local_window(s, center=tuple(loc["center"]), wsz=loc["wsz"])
    for k, (nx, ny) in enumerate(loc["candidates"]):
        sc, img = c.local_iteration_step(lw, nx, ny)
        assert sc == loc["scores_cloud"][k] and np.array_equal(img, zi["local_cloud_%d" % k])
        sc2, img2 = c.local_iteration_step(lw2, nx, ny)
        assert sc2 == loc["scores_window"][k] and np.array_equal(img2, zi["local_window_%d" % k])
    rc, st, _ = c.local_run(lw, res_x=H, res_y=W)
    assert [rc, st.nx, st.ny, st.last_score, st.evaluations] == [loc["run"][k] for k in ("rc", "nx", "ny", "last_score", "evaluations")]
    c2 = oracle_lib.Cloud(z["fr_x"], z["fr_y"], z["t"])
    c2.set_cloud(s, H, W)
    assert np.array_equal(c2.projection_img(s, H, W, show_final=True), zi["proj_raw"])
    c2.project_4param_reinit(*man["warps"][2])
    assert np.array_equal(c2.projection_img(s, H, W), zi["proj_warp2"])


def test_golden_stream(oracle_lib):
    """Third fixture: a warm-started (STM) run -- the next slice of the scene from the previous slice's model -- and the
    colour-coded time images."""
    man = json.load(open(os.path.join(GOLD, "manifest.json")))
    z, zs = np.load(os.path.join(GOLD, man["file"])), np.load(os.path.join(GOLD, man["stream"]["file"]))
    H, W, s, st = man["height"], man["width"], man["scale"], man["stream"]
    last = oracle_lib.Model()
    for k, v in man["final_model"].items():
        setattr(last, k, v)
    c = oracle_lib.Cloud(zs["b_fr_x"], zs["b_fr_y"], zs["b_t"])
    w = c.set_cloud(s, H, W)
    m = c.set_model(last)
    rc, loop, tr = c.run(w, m, res_x=H, res_y=W, trace_cap=4096)
    assert rc == st["warm_rc"] and loop.itercount == st["warm_iterations"] and m.as_dict() == st["warm_final_model"]
    assert loop.itercount < man["iterations"]          # the warm start needs fewer iterations than the cold one
    got = np.array([[r.model.total_dx, r.model.total_dy, r.model.total_rot, r.model.total_div,
                     r.loop.x_divider, r.loop.y_divider, r.loop.rot_divider, r.loop.div_divider] for r in tr])
    assert np.array_equal(got, zs["warm_trajectory"])
    c2 = oracle_lib.Cloud(z["fr_x"], z["fr_y"], z["t"])
    c2.set_cloud(s, H, W)
    assert np.array_equal(c2.color_time_img(s, H, W, show_final=True), zs["color_raw"])
    c2.project_4param_reinit(*man["warps"][2])
    assert np.array_equal(c2.color_time_img(s, H, W), zs["color_warp2"])


# ---- OptimizerLocal: the contrast-score optimiser (optimizer_sampler.cpp) ----

def _np_gauss(img, k):
    taps = {1: [1], 3: [1, 2, 1], 5: [1, 4, 6, 4, 1], 7: [2, 7, 14, 18, 14, 7, 2]}[k]
    h = k // 2
    pad = np.pad(img.astype(np.int64), h, mode="reflect") if h else img.astype(np.int64)   # == BORDER_REFLECT_101
    acc = np.zeros(img.shape, dtype=np.int64)
    for a in range(k):
        for b in range(k):
            acc += taps[a] * taps[b] * pad[a:a + img.shape[0], b:b + img.shape[1]]
    n2 = sum(taps) ** 2
    return ((acc + n2 // 2) // n2).astype(np.uint8)


def test_local_gauss_matches_independent_numpy(oracle_lib):
    rng = np.random.default_rng(5)
    for shape in ((1, 1), (2, 9), (9, 2), (37, 53), (64, 64)):
        img = rng.integers(0, 256, size=shape, dtype=np.uint8)
        img[rng.random(shape) < 0.4] = 0
        for k in (1, 3, 5, 7):
            if min(shape) > k // 2 or shape == (1, 1) and k == 1:   # np.pad cannot reflect past the array
                assert np.array_equal(oracle_lib.gauss_u8(img, k), _np_gauss(img, k)), (shape, k)
    # constant images stay constant; a saturated image stays 255
    assert np.all(oracle_lib.gauss_u8(np.full((5, 7), 255, np.uint8), 7) == 255)
    with pytest.raises(ValueError):
        oracle_lib.gauss_u8(np.zeros((4, 4), np.uint8), 9)


def test_local_count_image_and_score(oracle_lib):
    H, W, s = 90, 120, 3
    sl = synth.make_slice(6000, H, W, 0.05, seed=21)
    c = oracle_lib.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
    w = c.local_window(s)
    assert w.scale_img_x == w.metric_wsizex + s and w.c_t == 0
    # independent numpy restatement of the splat at (nx, ny) = 0: pr == fr
    img = c.local_count_img(w, 0.0, 0.0)
    x = sl["fr_x"].astype(np.int64) * s + (-w.c_fr_x * s + w.metric_wsizex / 2.0)
    y = sl["fr_y"].astype(np.int64) * s + (-w.c_fr_y * s + w.metric_wsizey / 2.0)
    X, Y = np.trunc(x).astype(np.int64), np.trunc(y).astype(np.int64)
    ok = (X >= 0) & (X < w.metric_wsizex) & (Y >= 0) & (Y < w.metric_wsizey)
    ref = np.zeros(img.shape, dtype=np.int64)
    for da in range(s):
        for db in range(s):
            np.add.at(ref, (X[ok] + da, Y[ok] + db), 1)
    assert np.array_equal(img, np.minimum(ref, 255).astype(np.uint8))
    # saturation really happens on a dense cloud
    dense = oracle_lib.Cloud(np.full(5000, 40, np.int32), np.full(5000, 50, np.int32), np.arange(5000, dtype=np.int64))
    wd = dense.local_window(3, center=(40, 50, 0), wsz=20)
    assert dense.local_count_img(wd, 0.0, 0.0).max() == 255
    # score == mean of the non-zero pixels of the blurred image
    sc, blurred = c.local_iteration_step(w, 0.05, -0.02)
    nz = blurred[blurred > 0].astype(np.float64)
    assert sc == nz.sum() / len(nz) == oracle_lib.nonzero_average(blurred)
    assert np.array_equal(blurred, _np_gauss(c.local_count_img(w, 0.05, -0.02), s))


def test_local_run_sharpens_the_image(oracle_lib):
    """Coordinate descent of optimizer_sampler.cpp:4-38 on a translating scene: terminates on the step
    threshold and ends at a higher contrast score than it started with."""
    H, W, s = 90, 120, 3
    sl = synth.make_slice(8000, H, W, 0.05, seed=22)
    c = oracle_lib.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
    w = c.local_window(s)
    s0, _ = c.local_iteration_step(w, 0.0, 0.0)
    rc, st, img = c.local_run(w, res_x=H, res_y=W)
    assert rc == 0 and st.evaluations % 2 == 1 and 10 < st.evaluations < 400
    assert np.hypot(st.dnx, st.dny) <= st.dn_th == 127000.0 / (10 * s * 1e8)
    assert st.last_score > s0
    # guard: a window smaller than scale * RES / 15 in both directions is skipped (:9-13)
    tiny = oracle_lib.Cloud(np.array([5, 6], np.int32), np.array([5, 7], np.int32), np.array([0, 10], np.int64))
    rc2, st2, _ = tiny.local_run(tiny.local_window(3), res_x=H, res_y=W)
    assert rc2 == 1 and st2.evaluations == 0


def test_projection_img_against_numpy(oracle_lib):
    """EventFile::projection_img at show_final (sensor positions): saturating splat, Gaussian, brightness scaling --
    restated independently in numpy."""
    H, W = 60, 80
    sl = synth.make_slice(9000, H, W, 0.05, seed=41)
    for s in (1, 3, 5):
        c = oracle_lib.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
        c.noise[::7] = 1
        img = c.projection_img(s, H, W, show_final=True)
        keep = c.noise == 0
        X, Y = sl["fr_x"][keep].astype(np.int64) * s, sl["fr_y"][keep].astype(np.int64) * s
        ok = (X < s * (H - 1)) & (Y < s * (W - 1))
        cnt = np.zeros((H * s, W * s), dtype=np.int64)
        for da in range(s):
            for db in range(s):
                np.add.at(cnt, (X[ok] + da, Y[ok] + db), 1)
        ref = _np_gauss(np.minimum(cnt, 255).astype(np.uint8), s)
        nz = ref[ref > 0].astype(np.float64)
        a = np.float32(127.0 / (nz.sum() / len(nz)))
        want = np.minimum(np.rint(ref.astype(np.float32) * a), 255).astype(np.uint8)
        assert np.array_equal(img, want), s
        assert abs(float(img[img > 0].mean()) - 127.0) < 8.0


def _np_hsv_to_bgr(H, S, V):
    """The HSV -> BGR convention stated in include/bf_accel.h, vectorised in float32."""
    h = H.astype(np.float32) * np.float32(6.0 / 180.0)
    sec = np.floor(h).astype(np.int64)
    f = h - sec.astype(np.float32)
    sec %= 6
    s = S.astype(np.float32) * np.float32(1.0 / 255.0)
    v = V.astype(np.float32) * np.float32(1.0 / 255.0)
    one = np.float32(1.0)
    tab = np.stack([v, v * (one - s), v * (one - s * f), v * (one - s * (one - f))])
    m = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])
    out = np.empty(H.shape + (3,), dtype=np.uint8)
    idx = np.arange(H.size).reshape(H.shape)
    flat = tab.reshape(4, -1)
    for ch in range(3):
        x = flat[m[sec, ch].ravel(), idx.ravel()].reshape(H.shape) * np.float32(255.0)
        out[..., ch] = np.clip(np.rint(x), 0, 255).astype(np.uint8)
    return out


def test_hsv_to_bgr_convention(oracle_lib):
    """8-bit HSV -> BGR of the colour time image: primaries, white at zero saturation, and the whole (H, S) table against
    an independent numpy restatement of the stated formula."""
    import ctypes as C
    L = oracle_lib.lib()

    def conv(H, S, V):
        o = (C.c_uint8 * 3)()
        L.bfo_hsv_to_bgr_u8(C.c_int32(H), C.c_int32(S), C.c_int32(V), o)
        return tuple(o)
    assert conv(0, 255, 255) == (0, 0, 255)      # red
    assert conv(30, 255, 255) == (0, 255, 255)   # yellow
    assert conv(60, 255, 255) == (0, 255, 0)     # green
    assert conv(120, 255, 255) == (255, 0, 0)    # blue
    assert conv(77, 0, 255) == (255, 255, 255)
    assert conv(0, 0, 0) == (0, 0, 0)
    H, S = np.meshgrid(np.arange(180), np.arange(256), indexing="ij")
    want = _np_hsv_to_bgr(H, S, np.full_like(H, 255))
    got = np.array([[conv(int(h), int(s), 255) for s in range(256)] for h in range(0, 180, 7)], dtype=np.uint8)
    assert np.array_equal(got, want[::7])


def test_color_time_img_structure(oracle_lib):
    """EventFile::color_time_img: a lone event paints its scale x scale block with the hue of its phase at full
    saturation; hue follows (t - t_min) / (t_max - t_min); and the whole image agrees with an order-free float64
    numpy restatement up to the 8-bit quantisation of hue / saturation."""
    H, W, s = 40, 50, 3
    # five isolated events; the first and the last only pin t_min / t_max (their phases sit on a hue boundary)
    fr_x = np.array([2, 5, 15, 25, 35], dtype=np.int32)
    fr_y = np.array([3, 7, 17, 27, 37], dtype=np.int32)
    t = np.array([0, 100000, 400000, 800000, 1000000], dtype=np.int64)
    c = oracle_lib.Cloud(fr_x, fr_y, t)
    img = c.color_time_img(s, H, W, show_final=True)
    assert img.shape == (H * s + s, W * s + s, 3)
    lit = img.any(axis=2)
    assert lit.sum() == 5 * s * s
    xs = -(H // 2) * s + H * s / 2.0
    ys = -(W // 2) * s + W * s / 2.0
    for k in (1, 2, 3):
        x0, y0 = int(fr_x[k] * s + xs), int(fr_y[k] * s + ys)
        blk = img[x0:x0 + s, y0:y0 + s].reshape(-1, 3)
        assert (blk == blk[0]).all() and lit[x0:x0 + s, y0:y0 + s].all()
        a = np.float32(2 * 3.14 * (t[k] / 1e6))
        cs, sn = float(np.float32(np.cos(float(a)))), float(np.float32(np.sin(float(a))))
        hue = int(((np.arctan2(sn, cs) + 3.1416) * 180 / 3.1416) / 2)
        sat = int(np.hypot(cs, sn) * 255)
        assert tuple(blk[0]) == tuple(_np_hsv_to_bgr(np.array([hue]), np.array([sat]), np.array([255]))[0])

    sl = synth.make_slice(20000, H, W, 0.05, seed=43)
    c = oracle_lib.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
    c.noise[::9] = 1
    for sc in (1, 3, 4):
        img = c.color_time_img(sc, H, W, show_final=True)
        keep = c.noise == 0
        tt = sl["t"].astype(np.int64)
        tmin, tmax = tt.min(), max(tt.max(), 0)
        ang = (2 * 3.14 * ((tt - tmin) / float(tmax - tmin))).astype(np.float32).astype(np.float64)[keep]
        xs = -(H // 2) * sc + H * sc / 2.0
        ys = -(W // 2) * sc + W * sc / 2.0
        X = (sl["fr_x"][keep] * sc + xs).astype(np.int64)
        Y = (sl["fr_y"][keep] * sc + ys).astype(np.int64)
        ok = (X < H * sc) & (Y < W * sc) & (X >= 0) & (Y >= 0)
        R, Cc = H * sc + sc, W * sc + sc
        cnt = np.zeros((R, Cc)); sc_ = np.zeros((R, Cc)); ss_ = np.zeros((R, Cc))
        bw = 2 * (sc // 2) + 1
        for da in range(bw):
            for db in range(bw):
                np.add.at(cnt, (X[ok] + da, Y[ok] + db), 1)
                np.add.at(sc_, (X[ok] + da, Y[ok] + db), np.cos(ang[ok]))
                np.add.at(ss_, (X[ok] + da, Y[ok] + db), np.sin(ang[ok]))
        assert np.array_equal(img.any(axis=2), cnt > 0)      # value = 255: every covered pixel is lit
        m = cnt > 0
        vx = np.where(m, sc_ / np.maximum(cnt, 1), 0).astype(np.float32)
        vy = np.where(m, ss_ / np.maximum(cnt, 1), 0).astype(np.float32)
        speed = np.hypot(vx.astype(np.float64), vy.astype(np.float64))
        angle = np.where(speed != 0, (np.arctan2(vy.astype(np.float64), vx.astype(np.float64)) + 3.1416) * 180 / 3.1416, 0)
        want = _np_hsv_to_bgr((angle / 2).astype(np.int64), (speed * 255).astype(np.int64), np.full(cnt.shape, 255))
        want[~m] = 0
        diff = np.abs(img.astype(np.int64) - want.astype(np.int64)).max(axis=2)
        # f32 running sums vs float64 sums: only pixels sitting on a hue / saturation truncation boundary may move
        assert (diff > 0).mean() < 0.01 and diff.max() <= 10, (sc, (diff > 0).mean(), diff.max())


def test_division_by_event_count_as_table_product():
    """The device's stencil kernel replaces the reference's `sum / count` (f32 / f32, accel_lib.h:172) by
    (float)((double)sum * RN64(1.0 / count)) for counts below 256 (bf_device_fns.h: time_from_sums).  tests/exhaustive_div.c (d)
    proves the identity for EVERY finite f32 sum and every count in [1, 255] (run once per change of that code: 10 minutes);
    this is the sampled form that runs with the suite: 4 M random dividends of all magnitudes a time sum can have, every
    count, plus dividends adjacent to exact multiples (where a quotient comes closest to a rounding boundary)."""
    rng = np.random.default_rng(7)
    a = np.concatenate([
        (10.0 ** rng.uniform(-9, 1.5, 2000000)).astype(np.float32),                       # 1 ns .. 30 s
        rng.uniform(0, 8, 1000000).astype(np.float32),
        -(10.0 ** rng.uniform(-9, 0, 500000)).astype(np.float32),                          # events before the slice start
        np.array([0.0, 1e-9, 3e-2, 7.65], np.float32),
    ])
    # neighbours of exact products q * b: quotients next to representable values
    q = rng.uniform(1e-6, 1.0, 500000).astype(np.float32)
    b_ = rng.integers(1, 256, 500000).astype(np.float32)
    prod = (q * b_).astype(np.float32)
    a = np.concatenate([a, prod, np.nextafter(prod, np.float32(np.inf)), np.nextafter(prod, np.float32(-np.inf))])
    rcp = 1.0 / np.arange(1, 256, dtype=np.float64)
    bad = 0
    for b in range(1, 256):
        ref = a / np.float32(b)
        got = (a.astype(np.float64) * rcp[b - 1]).astype(np.float32)
        bad += int(np.count_nonzero(ref.view(np.uint32) != got.view(np.uint32)))
    assert bad == 0, bad
