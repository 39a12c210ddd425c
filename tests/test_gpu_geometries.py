"""Parity at the geometries of BASELINE configs 3 and 5 (640x480, 1280x720) and config 2 cold to convergence.

At these image sizes a few thousand events leave their bin's LDS tile between two re-sorts in EVERY iteration, so the
exact overflow path, the overflow planes and the stencil's overflow branch are all live -- unlike config 2, where the
overflow count is 0.  Everything goes through the C-ABI and is compared with the CPU oracle on the same inputs.
Bars: event-count image bit-exact, time image <= 1e-6, trajectory <= 1e-6 up to the first
pixel-boundary crossing and within the oracle's own event-order sensitivity after it (derived in the test), converged
per-event flow <= 1e-4 relative or 0.02 px/s, iteration count within +-1.
"""
import numpy as np
import pytest

from better_flow_amd import synth
from helpers import make_accel

pytestmark = pytest.mark.gpu

# Trajectory bar: 1e-6 relative, with a floor per field of <= 1 % of the loop's own stopping threshold for that field
# (|dx / x_div| < 1e-5, |rot / rot_div| < 1e-4 with rot_div >= 1e4, |div / div_div| < 1e-1 with div_div >= 1e4,
# optimizer_rolling.h:81-84): rot and div are near-cancelling sums over ~1e6 pixels of terms of magnitude 1e2 * 1e-2,
# so the 1e-7 order noise of the reference's f32 time sums (accel_lib.h:162) shows up as ~1e-9 ABSOLUTE on them.
FIELDS = {"dx": 1e-4, "dy": 1e-4, "rot": 1e-2, "div": 1.0,
          "total_dx": 1e-4, "total_dy": 1e-4, "total_rot": 1e-6, "total_div": 1e-4}


def _flow_close(u, ou, rel=1e-4, abs_=0.02):
    tol = np.maximum(rel * np.abs(ou), abs_)
    bad = np.abs(u - ou) > tol
    if bad.any():
        print("flow deviation: max abs %.3e px/s on %d events" % (np.abs(u - ou).max(), int(bad.sum())))
    return not bad.any()


def _gpu_run(accel_mod, sl, H, W, s, max_iter, trace_cap, warm=None, **options):
    a = make_accel(accel_mod, options, max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)   # ("debug_margin": tests/helpers.py)
    a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
    a.set_cloud(s, H, W)
    if warm is not None:
        a.set_model(warm)
    o = a.default_opts()
    o.res_x, o.res_y, o.max_iter, o.trace_cap, o.want_uv = H, W, max_iter, trace_cap, 1
    rc, m, info = a.run(o)
    tr = [t_.model.as_dict() for t_ in a.get_trace(trace_cap)] if trace_cap else []
    u, v = a.compute_uv()
    a.close()
    return rc, m, info, tr, u, v


@pytest.mark.parametrize("H,W,K", [(480, 640, 40), (720, 1280, 40)])
def test_large_geometry_against_oracle(oracle_lib, accel_mod, H, W, K):
    """1M events at 640x480 / 1280x720, scale 3 (accel_lib.h:147-178, optimizer_rolling.h:48-125)."""
    s = 3
    sl = synth.make_slice(1000000, H, W, 0.030, seed=1)
    n = len(sl["t"])
    # -- operator parity: event-count image bit-exact, time image <= 1e-6, before and after a warp
    oc = oracle_lib.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
    ow = oc.set_cloud(s, H, W)
    acc = accel_mod.Accel(max_events=n, max_rows=s * H + s, max_cols=s * W + s)
    acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
    gw = acc.set_cloud(s, H, W)
    assert (gw.scale_img_x, gw.scale_img_y, gw.x_shift, gw.y_shift) == (ow.scale_img_x, ow.scale_img_y, ow.x_shift, ow.y_shift)
    for prm in ((0.0, 0.0, 0.0, 0.0, 0.0, 0.0), (0.4, -0.8, H / 2.0, W / 2.0, 1.5e-4, -3.0e-5)):
        oc.project_4param_reinit(*prm)
        acc.project_4param_reinit(*prm)
        otime, ocnt = oc.get_time_img(ow)
        gtime, gcnt = acc.get_time_img()
        assert np.array_equal(gcnt, ocnt.astype(np.uint32)), "event-count image must be bit-exact"
        one = ocnt == 1.0
        assert np.array_equal(gtime[one], otime[one])
        np.testing.assert_allclose(gtime, otime, rtol=1e-6, atol=0)
        assert int(gcnt.sum()) > 8 * n
    acc.close()
    # -- the first K + 1 iterations of the cold loop against the oracle's trajectory.
    # The reference loop is sensitive to the ORDER of the events: accel_lib.h:162 accumulates the time image in f32 in
    # container order, so two orders of the same slice give time images that differ by ~1e-7 relative, trajectories
    # that differ by ~1e-11 -- until that difference moves the first event across a pixel boundary (iteration 33 at
    # 640x480, 15 at 1280x720 for this slice).  From there on the valid-pixel counts differ and the two runs random-walk
    # apart (1280x720: 8.8e-5 on total_dy = -1.3e-2 after 40 iterations).  So the oracle is run twice, events forward
    # and reversed; the GPU (exact integer time sums, order-free) must follow the forward run
    #   * to rounding -- valid-pixel count equal, every field <= 1e-6 -- until its first pixel-boundary crossing, which
    #     must not come before iteration 9, and
    #   * within 4 x the oracle's own forward / reversed spread afterwards.
    def oracle_run(order):
        o = oracle_lib.Cloud(sl["fr_x"][order], sl["fr_y"][order], sl["t"][order])
        w_ = o.set_cloud(s, H, W)
        m_ = oracle_lib.Model()
        rc_, lp_, tr_ = o.run(w_, m_, max_iter=K, res_x=H, res_y=W, trace_cap=K + 1)
        assert rc_ == 0 and lp_.itercount == K + 1
        u_, v_ = o.compute_uv()
        inv = np.empty(n, np.int64)
        inv[order] = np.arange(n)
        return tr_, u_[inv], v_[inv]
    otr, ou, ov = oracle_run(np.arange(n))
    otr_r, ou_r, ov_r = oracle_run(np.arange(n)[::-1].copy())
    spread = {f: max(abs(getattr(otr[k].model, f) - getattr(otr_r[k].model, f)) for k in range(K + 1)) for f in FIELDS}
    k_o = next((k for k in range(K + 1) if otr[k].model.cnt != otr_r[k].model.cnt), None)
    assert k_o is not None, "the window must be long enough for the oracle's own order sensitivity to show"
    runs = {}
    for name, opts in (("binned", dict(binned=2)), ("atomics", dict(binned=0)), ("auto", dict()),
                       ("compact", dict(binned=2, bin_compact=2)), ("dense", dict(binned=2, bin_compact=0)),
                       ("tail_update", dict(binned=2, co_schedule=1)),
                       ("no_predict", dict(binned=2, debug_margin=4, bin_predict=0))):
        runs[name] = _gpu_run(accel_mod, sl, H, W, s, K, K + 1, **opts)
    b = runs["binned"]
    assert b[0] == 0 and b[2].iterations == K + 1
    assert b[2].rebins >= 1, "the binned path must be the one that ran"
    # The overflow path + the stencil's overflow branch must be live at this geometry.  With the drift prediction on, how many
    # events take it depends on how far the host's re-bin lags the device's request -- usually 299 here, NONE when a slow
    # host (a cold box) lets the queue run dry and re-bins exactly at the request: seen once in ~20 suite runs.  Without the
    # prediction a re-bin is only asked for once events HAVE overflowed: live whatever the timing.
    assert runs["no_predict"][2].overflow_events > 0, "overflow path + the stencil's overflow branch must be live at this geometry"
    assert runs["auto"][2].rebins >= 1, "1M events at this geometry are dense enough for the binned path by default"
    for name in ("atomics", "auto", "compact", "dense", "tail_update", "no_predict"):   # integer accumulators: every scatter mode gives the same bits
        r = runs[name]
        assert (r[0], r[2].iterations, r[1].as_dict(), r[3]) == (b[0], b[2].iterations, b[1].as_dict(), b[3]), name
        assert np.array_equal(r[4], b[4]) and np.array_equal(r[5], b[5]), name
    k_g = next((k for k in range(K + 1) if b[3][k]["cnt"] != otr[k].model.cnt), K + 1)
    assert k_g >= 9, k_g
    worst = [0.0, 0.0]
    for k in range(K + 1):
        g, o_ = b[3][k], otr[k].model
        for f, floor in FIELDS.items():
            ov_ = getattr(o_, f)
            if k < k_g:
                dev = abs(g[f] - ov_) / max(abs(ov_), floor)
                worst[0] = max(worst[0], dev)
                assert dev <= 1e-6, (k, f, g[f], ov_)
            else:
                worst[1] = max(worst[1], abs(g[f] - ov_) / max(spread[f], 1e-300))
                assert abs(g[f] - ov_) <= 4.0 * spread[f] + 1e-6 * max(abs(ov_), floor), (k, f, g[f], ov_, spread[f])
        if k >= k_g:
            assert abs(g["cnt"] - o_.cnt) <= 4 * max(abs(otr[j].model.cnt - otr_r[j].model.cnt) for j in range(K + 1))
    print("%dx%d: first pixel-boundary crossing at iteration %d (GPU vs oracle) / %d (oracle forward vs reversed); "
          "worst relative deviation before it %.2e, worst deviation after it %.2f x the oracle's own spread; "
          "%d overflow events, %d re-bins" % (W, H, k_g + 1, k_o + 1, worst[0], worst[1], b[2].overflow_events, b[2].rebins))
    # the flow after the capped run (same iteration count on all sides), against the same yardstick
    for g_, o_, r_ in ((b[4], ou, ou_r), (b[5], ov, ov_r)):
        yard = np.abs(o_ - r_).max()
        assert np.all(np.abs(g_ - o_) <= 4.0 * yard + 1e-6 * np.abs(o_) + 1e-3), (np.abs(g_ - o_).max(), yard)


def test_config3_rolling_slices_with_stm(oracle_lib, accel_mod):
    """BASELINE config 3 as specified: rolling 30 ms slices of ~1M events at 640x480 through the copy stream
    (bf_upload_events_async / bf_commit_upload, two staging slots), each slice warm-started from the previous model
    (STM, dvs_flow.h:218-224).  The streamed chain equals the blocking chain bit for bit, and every warm slice is
    checked against the oracle started from the SAME model (optimizer_rolling.h:289-299)."""
    H, W, s, NS = 480, 640, 3, 3
    sls = [synth.make_slice(1000000, H, W, 0.030, seed=300 + i) for i in range(NS)]
    nmax = max(len(sl["t"]) for sl in sls)
    acc = accel_mod.Accel(max_events=nmax, max_rows=s * H + s, max_cols=s * W + s)
    opts = acc.default_opts()
    opts.res_x, opts.res_y, opts.want_uv = H, W, 1

    def chain(upload):
        prev, out = None, []
        for i, sl in enumerate(sls):
            upload(i, sl)
            acc.set_cloud(s, H, W)
            if prev is not None:
                acc.set_model(prev)
            start = prev
            rc, prev, info = acc.run(opts)
            u, v = acc.compute_uv()
            out.append((rc, info.iterations, info.overflow_events, start, prev, u, v))
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
        if i + 1 < NS:
            put(i + 1)      # the next slice's DMA overlaps this slice's solve

    got = chain(up)
    acc.close()
    for r, g in zip(ref, got):
        assert (r[0], r[1]) == (g[0], g[1]) and r[4].as_dict() == g[4].as_dict()
        assert np.array_equal(r[5], g[5]) and np.array_equal(r[6], g[6])
    assert got[0][0] == 0 and got[0][1] > 100          # the cold head of the chain
    for i in range(1, NS):                              # warm slices against the oracle from the same model
        rc, its, ovf, start, model, u, v = got[i]
        oc = oracle_lib.Cloud(sls[i]["fr_x"], sls[i]["fr_y"], sls[i]["t"])
        ow = oc.set_cloud(s, H, W)
        om = oc.set_model(oracle_lib.Model(**start.as_dict()))
        orc, oloop, _ = oc.run(ow, om, res_x=H, res_y=W)
        assert rc == orc == 0
        assert abs(its - oloop.itercount) <= 1, (i, its, oloop.itercount)
        assert its < 60, "a warm start converges in a handful of iterations"
        ou, ov = oc.compute_uv()
        if its == oloop.itercount:
            assert _flow_close(u, ou, rel=1e-6, abs_=1e-3) and _flow_close(v, ov, rel=1e-6, abs_=1e-3), i
        else:
            assert _flow_close(u, ou) and _flow_close(v, ov), i


def test_config2_cold_to_convergence_against_oracle(oracle_lib, accel_mod):
    """BASELINE config 2 at full size, cold start run to the reference loop's own termination on both sides
    (optimizer_rolling.h:73-101): iteration count within +-1, per-event flow within 1e-4 / 0.02 px/s."""
    H, W, s = 260, 346, 3
    sl = synth.make_slice(1000000, H, W, 0.030, seed=1)
    oc = oracle_lib.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
    ow = oc.set_cloud(s, H, W)
    om = oracle_lib.Model()
    orc, oloop, _ = oc.run(ow, om, res_x=H, res_y=W)
    ou, ov = oc.compute_uv()
    rc, m, info, _, u, v = _gpu_run(accel_mod, sl, H, W, s, -1, 0)
    assert rc == orc == 0
    assert abs(info.iterations - oloop.itercount) <= 1, (info.iterations, oloop.itercount)
    assert info.iterations > 300
    # the real reference on this configuration (its probe driver on the survey's draw of the same
    # generator): 532 iterations -- same place, not same bits
    assert abs(oloop.itercount - 532) <= 27, oloop.itercount
    assert (info.x_divider, info.y_divider, info.rot_divider, info.div_divider) == \
        (oloop.x_divider, oloop.y_divider, oloop.rot_divider, oloop.div_divider)
    assert _flow_close(u, ou) and _flow_close(v, ov)
    for f in ("total_dx", "total_dy", "total_rot", "total_div"):
        assert abs(getattr(m, f) - getattr(om, f)) <= 1e-4 * max(abs(getattr(om, f)), 1e-6), f


def test_config2_scale1_cold_to_convergence_against_oracle(oracle_lib, accel_mod):
    """BASELINE config 2's other scale ("also s = 1"): 1M events at 346x260 on the 263 x 349 image, eleven
    events per pixel; cold start to the loop's own termination on both sides.  The real reference needed 268 iterations
    there."""
    H, W, s = 260, 346, 1
    sl = synth.make_slice(1000000, H, W, 0.030, seed=1)
    oc = oracle_lib.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
    ow = oc.set_cloud(s, H, W)
    om = oracle_lib.Model()
    orc, oloop, _ = oc.run(ow, om, res_x=H, res_y=W)
    ou, ov = oc.compute_uv()
    rc, m, info, _, u, v = _gpu_run(accel_mod, sl, H, W, s, -1, 0)
    assert rc == orc == 0
    assert abs(info.iterations - oloop.itercount) <= 1, (info.iterations, oloop.itercount)
    assert abs(oloop.itercount - 268) <= 27, oloop.itercount
    assert _flow_close(u, ou) and _flow_close(v, ov)


@pytest.mark.parametrize("seed", [1, 0, 2, 4, 3, 5])
def test_config5_to_termination_against_golden(accel_mod, seed):
    """BASELINE config 5's slice (1M events, 1280x720, scale 3) cold to the loop's OWN termination (optimizer_rolling.h:
    76-101): thousands of iterations, where test_large_geometry_against_oracle compares the first 41.  The oracle needs
    ~25 minutes per run, so its results are committed fixtures (tests/golden/config5_720p_seed<S>.npz and
    ..._ensemble.npz, written by tests/golden/make_config5_golden.py): the oracle on SIX orders of the slice's events --
    upload order, reversed, four seeded random permutations.  All six are "the reference's answer": accel_lib.h:162 adds
    the time image in f32 in container order, and the loop amplifies the last bits -- a divider doubles whenever its
    gradient component changes sign between two iterations (optimizer_rolling.h:98-101), which near the optimum is decided
    by rounding noise.
      seed 1: 4891 .. 4958 iterations, per-event flow within 0.5 / 1.3 px/s across the six (on ~1500 px/s);
      seed 0: 5389 .. 7355 iterations, and the row flow differs by 590 px/s between members -- on four of the six orders
              the x divider has doubled away before total_dx got anywhere near the injected -600 px/s.  The reference
              loop has no unique answer on this slice; it is kept because config 5's batch contains such slices.
      seeds 2, 4: families 23 and 9 px/s wide;  seeds 3, 5: the six orders scatter into four branches each (see below).
    The GPU (order-free integer sums) is one more run of the same loop: it must belong to ONE FAMILY of the oracle's members
    (seed 0 has two: upload order / reversed, and the four permutations, 590 px/s apart).  Bars, all against that family:
    return code 0; iteration count inside the family's range widened by a quarter of its width (+ 1 %); per-event flow at
    4096 sampled events, its percentiles and the model's totals inside the family's envelope widened by the family's own
    (largest) width plus north_star's 1e-4 relative / 0.02 px/s."""
    import hashlib
    import os
    H, W, s = 720, 1280, 3
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z = np.load(os.path.join(gold, "config5_720p_seed%d.npz" % seed))
    e = np.load(os.path.join(gold, "config5_720p_seed%d_ensemble.npz" % seed))
    assert tuple(z["geometry"]) == (H, W, s) and str(e["input_sha256"]) == str(z["input_sha256"])
    sl = synth.make_slice(1000000, H, W, 0.030, seed=seed)
    h = hashlib.sha256()
    for k in ("fr_x", "fr_y", "t"):
        h.update(np.ascontiguousarray(sl[k]).tobytes())
    assert h.hexdigest() == str(z["input_sha256"]), "the generator no longer produces the fixture's slice"
    rc, m, info, _, u, v = _gpu_run(accel_mod, sl, H, W, s, -1, 0)
    its = np.array([int(z["fwd_iterations"]), int(z["rev_iterations"])] + e["iterations"].tolist())
    assert rc == 0 and int(z["fwd_rc"]) == int(z["rev_rc"]) == 0 and not e["rc"].any()
    idx = z["sample_idx"]
    assert np.array_equal(idx, e["sample_idx"])
    U, V = np.vstack([z["fwd_u"], z["rev_u"], e["u"]]), np.vstack([z["fwd_v"], z["rev_v"], e["v"]])
    UP, VP = np.vstack([z["fwd_u_pct"], z["rev_u_pct"], e["u_pct"]]), np.vstack([z["fwd_v_pct"], z["rev_v_pct"], e["v_pct"]])
    fields = [str(f) for f in z["fields"]]
    models = np.vstack([z["fwd_model"], z["rev_model"], e["model"]])
    # The members fall into FAMILIES -- branches of the loop's divider doublings: single-link clusters of the sampled flow
    # (two members are linked when they differ by less than a tenth of the ensemble's width, or by less than 2 px/s).  Seed
    # 1: one family, 1.3 px/s wide.  Seed 0: {upload order, reversed} 4.6 px/s wide at u = -600 px/s, and the four
    # permutations 20 px/s wide at u = -10 .. -29 px/s, 590 px/s away.  The GPU must belong to ONE family: every bar below
    # is taken against that family's members and widened by that family's own width, not by the ensemble's.
    nmem = len(U)
    dist = np.array([[max(np.abs(U[i] - U[j]).max(), np.abs(V[i] - V[j]).max()) for j in range(nmem)] for i in range(nmem)])
    link = dist < max(2.0, 0.1 * dist.max())
    fam = list(range(nmem))
    for i in range(nmem):
        for j in range(nmem):
            if link[i, j]:
                fi, fj = fam[i], fam[j]
                fam = [fi if f == fj else f for f in fam]
    families = [np.array([k for k in range(nmem) if fam[k] == f]) for f in sorted(set(fam))]

    def outside(g, members):
        """How far (px/s) g lies outside the members' envelope widened by their own largest width plus north_star's
        1e-4 relative / 0.02 px/s (<= 0: inside); and how far outside the raw envelope."""
        lo, hi = members.min(axis=0), members.max(axis=0)
        w = (hi - lo).max() + np.maximum(1e-4 * np.maximum(np.abs(lo), np.abs(hi)), 0.02)
        return float(np.max(np.maximum(lo - w - g, g - hi - w))), float(np.max(np.maximum(lo - g, g - hi)))

    verdicts = []
    for F in families:
        fits = its[F]
        slack = 0.25 * (fits.max() - fits.min()) + 0.01 * fits.mean()
        ok = fits.min() - slack <= info.iterations <= fits.max() + slack
        ou_, raw_u = outside(u[idx], U[F])
        ov_, raw_v = outside(v[idx], V[F])
        ok = ok and ou_ <= 0 and ov_ <= 0
        ok = ok and outside(np.percentile(u, z["percentiles"]), UP[F])[0] <= 0 and outside(np.percentile(v, z["percentiles"]), VP[F])[0] <= 0
        for f in ("total_dx", "total_dy", "total_rot", "total_div"):
            k = fields.index(f)
            lo, hi = models[F, k].min(), models[F, k].max()
            w = 2.0 * (hi - lo) + 1e-4 * max(abs(lo), abs(hi)) + 1e-9   # (four scalars of a few members: two widths)
            ok = ok and lo - w <= getattr(m, f) <= hi + w
        width = max(float((U[F].max(0) - U[F].min(0)).max()), float((V[F].max(0) - V[F].min(0)).max()))
        verdicts.append((bool(ok), F.tolist(), fits.tolist(), width, max(raw_u, raw_v)))
    if len(families) >= 4 and not any(v_[0] for v_ in verdicts):
        # Seeds 3 and 5: the six orders scatter into four branches -- most members are alone in theirs (seed 3: 5203 ..
        # 22603 iterations, 1672 / 2924 px/s apart; seed 5: 3635 .. 17707 iterations, 44 / 95 px/s apart), so six samples do
        # not cover the branches the reference loop can take on this slice and "belongs to one family" cannot be asked of a
        # seventh run.  What can: the GPU run is one more sample of the same scatter -- return code 0, an iteration count
        # inside the members' range, the sampled flow inside the members' envelope widened by a tenth of its width, and no
        # further from its nearest member than the members are from theirs.
        assert its.min() <= info.iterations <= its.max(), (info.iterations, its.tolist())
        for g_, M in ((u[idx], U), (v[idx], V)):
            lo, hi = M.min(axis=0), M.max(axis=0)
            w = 0.1 * (hi - lo).max()
            assert np.all(g_ >= lo - w) and np.all(g_ <= hi + w), (seed, float(np.max(np.maximum(lo - g_, g_ - hi))))
        nn = np.sort(dist + np.diag([np.inf] * nmem), axis=1)[:, 0]
        mine = min(max(np.abs(u[idx] - U[k]).max(), np.abs(v[idx] - V[k]).max()) for k in range(nmem))
        assert mine <= nn.max(), (mine, nn.tolist())
        print("config 5 seed %d to termination: GPU %d iterations; the oracle's six event orders %s scatter into %d branches (nearest-"
              "neighbour distances %s px/s); the GPU run is %.2f px/s from its nearest member and inside the six orders' envelope" %
              (seed, info.iterations, its.tolist(), len(families), np.round(nn, 2).tolist(), mine))
        return
    assert any(v_[0] for v_ in verdicts), (info.iterations, verdicts)
    home = next(v_ for v_ in verdicts if v_[0])
    print("config 5 seed %d to termination: GPU %d iterations; the oracle's six event orders %s form %d famil%s; the GPU belongs to members "
          "%s (iterations %s, %.2f px/s wide): sampled flow %s that family's raw envelope" %
          (seed, info.iterations, its.tolist(), len(families), "y" if len(families) == 1 else "ies", home[1], home[2], home[3],
           "inside" if home[4] <= 0 else "at most %.3f px/s outside" % home[4]))
