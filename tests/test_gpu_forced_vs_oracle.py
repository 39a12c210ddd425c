"""Every loop form of the library, FORCED, against the CPU oracle (not against another device loop).

tests/test_gpu_fused.py and tests/test_gpu_split.py show that the one-kernel iteration (`fused`) and the own-pixels +
margin-plane format (`bin_split`) return the bits of the two-kernel dense-slab loop, which the other suites hold to the
oracle -- a transitive argument.  Here each form is selected explicitly (and the test checks that it is the one that ran)
and compared with `oracle/bf_oracle.c` directly on the same slice:

* the valid-pixel count of the time image -- an integer, the support of the event-count image seen through the form's own
  scatter + stencil kernels -- equal at EVERY iteration up to the first pixel-boundary crossing (accel_lib.h:147-178,
  object_model.cpp:103-126);
* every field of the 41-record trajectory <= 1e-6 relative (floors as in test_gpu_geometries.py) up to that crossing,
  which must not come before iteration 9, and within 4 x the oracle's own forward / reversed spread afterwards
  (optimizer_rolling.h:48-125,305-347);
* the per-event flow after the capped run within the same yardstick (event.h:135-142).

The loops that have a warm-start path of their own -- the one-kernel iteration and the persistent loop kernel fuse the warp of
OptimizerRolling::set_model (optimizer_rolling.h:289-299) into their first counting sort -- are also forced on a WARM-STARTED
slice: the next slice of the stream, started from the oracle's model of this one, against the oracle doing the same.
"""
import numpy as np
import pytest

from better_flow_amd import synth

pytestmark = pytest.mark.gpu

FIELDS = {"dx": 1e-4, "dy": 1e-4, "rot": 1e-2, "div": 1.0,
          "total_dx": 1e-4, "total_dy": 1e-4, "total_rot": 1e-6, "total_div": 1e-4}
K = 40

# name -> (options, scatter_format the context must report, one-kernel loop expected[, persistent loop kernel expected])
FORMS = {
    "one-kernel iteration (fused=2)": (dict(binned=2, fused=2, persist=0), None, 1),
    "persistent loop kernel (fused=2, persist=2)": (dict(binned=2, fused=2, persist=2), None, 1, 1),
    "own pixels + margin plane, update at the scatter head (bin_split=2)":
        (dict(binned=2, fused=0, bin_compact=0, bin_split=2), 3, 0),
    "own pixels + margin plane, update in the stencil tail (bin_split=2, co_schedule)":
        (dict(binned=2, fused=0, bin_compact=0, bin_split=2, co_schedule=1), 3, 0),
    "dense slabs, update in the stencil tail": (dict(binned=2, fused=0, bin_compact=0, bin_split=0, co_schedule=1), 0, 0),
    "own pixels + margin plane, co-scheduled, the update as a kernel of its own (sep_update=2)":
        (dict(binned=2, fused=0, bin_compact=0, bin_split=2, co_schedule=1, sep_update=2), 3, 0),
    "event lists (bin_compact=2)": (dict(binned=2, fused=0, bin_compact=2), 2, 0),
    "event lists, co-scheduled: the update as a kernel of its own (sep_update auto)": (dict(binned=2, fused=0, bin_compact=2, co_schedule=1), 2, 0),
    "event lists, co-scheduled, update in the stencil tail (sep_update=0)": (dict(binned=2, fused=0, bin_compact=2, co_schedule=1, sep_update=0), 2, 0),
    "dense slabs, co-scheduled, the update as a kernel of its own (sep_update=2)":
        (dict(binned=2, fused=0, bin_compact=0, bin_split=0, co_schedule=1, sep_update=2), 0, 0),
    "global atomics (binned=0)": (dict(binned=0, fused=0), -1, 0),
}


@pytest.fixture(scope="module")
def case(oracle_lib):
    H, W, s = 260, 346, 3
    sl = synth.make_slice(300000, H, W, 0.030, seed=41)
    n = len(sl["t"])

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
    cnt_spread = max(abs(otr[k].model.cnt - otr_r[k].model.cnt) for k in range(K + 1))
    return dict(H=H, W=W, s=s, sl=sl, otr=otr, ou=ou, ov=ov, ou_r=ou_r, ov_r=ov_r, spread=spread, cnt_spread=cnt_spread)


@pytest.fixture(scope="module")
def warm_case(oracle_lib, case):
    """The next slice of the stream (another draw of the same scene), warm-started from the oracle's model of `case`'s slice
    after its K + 1 iterations (dvs_flow.h:218-219: set_model(last) before run())."""
    H, W, s = case["H"], case["W"], case["s"]
    sl = synth.make_slice(300000, H, W, 0.030, seed=42)
    n = len(sl["t"])
    start = oracle_lib.Model(**case["otr"][K].model.as_dict())

    def oracle_run(order):
        o = oracle_lib.Cloud(sl["fr_x"][order], sl["fr_y"][order], sl["t"][order])
        w_ = o.set_cloud(s, H, W)
        m_ = o.set_model(start)
        rc_, lp_, tr_ = o.run(w_, m_, max_iter=K, res_x=H, res_y=W, trace_cap=K + 1)
        assert rc_ == 0
        u_, v_ = o.compute_uv()
        inv = np.empty(n, np.int64)
        inv[order] = np.arange(n)
        return tr_[:lp_.itercount], u_[inv], v_[inv]
    otr, ou, ov = oracle_run(np.arange(n))
    otr_r, ou_r, ov_r = oracle_run(np.arange(n)[::-1].copy())
    L = min(len(otr), len(otr_r))
    spread = {f: max(abs(getattr(otr[k].model, f) - getattr(otr_r[k].model, f)) for k in range(L)) for f in FIELDS}
    cnt_spread = max(abs(otr[k].model.cnt - otr_r[k].model.cnt) for k in range(L))
    return dict(H=H, W=W, s=s, sl=sl, start=start, otr=otr, n_r=len(otr_r), ou=ou, ov=ov, ou_r=ou_r, ov_r=ov_r, spread=spread,
                cnt_spread=cnt_spread)


@pytest.mark.parametrize("form", list(FORMS), ids=lambda f: f.split(" (")[0].replace(" ", "_").replace(",", ""))
def test_forced_form_against_the_oracle(accel_mod, case, form):
    options, want_fmt, want_one_kernel = FORMS[form][:3]
    want_persistent = FORMS[form][3] if len(FORMS[form]) > 3 else 0
    H, W, s, sl = case["H"], case["W"], case["s"], case["sl"]
    a = accel_mod.Accel(max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
    for k, v in options.items():
        a.set_option(k, v)
    a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
    a.set_cloud(s, H, W)
    assert a.get_stat("one_kernel") == want_one_kernel, form
    assert a.get_stat("persistent") == want_persistent, form
    if want_fmt is not None:
        assert a.get_stat("scatter_format") == want_fmt, (form, a.get_stat("scatter_format"))
    o = a.default_opts()
    o.res_x, o.res_y, o.max_iter, o.trace_cap, o.want_uv = H, W, K, K + 1, 1
    rc, m, info = a.run(o)
    tr = [t_.model.as_dict() for t_ in a.get_trace(K + 1)]
    u, v = a.compute_uv()
    giveups = a.get_stat("persist_giveups")
    a.close()
    assert rc == 0 and info.iterations == K + 1 and len(tr) == K + 1
    if want_one_kernel:
        assert info.launches < 1.5 * info.iterations + 3 * info.rebins + 8, "the one-kernel loop was not the one that ran"
    if want_persistent:   # rounds of [re-bin trio, loop kernel, gated final warp], one per re-bin: far fewer launches than iterations
        assert giveups == 0 and info.launches <= 5 * (info.rebins + 2) and info.launches < info.iterations, (form, info.launches, info.rebins)
    otr, spread = case["otr"], case["spread"]
    k_g = next((k for k in range(K + 1) if tr[k]["cnt"] != otr[k].model.cnt), K + 1)
    assert k_g >= 9, (form, k_g)
    worst = [0.0, 0.0]
    for k in range(K + 1):
        g, o_ = tr[k], otr[k].model
        for f, floor in FIELDS.items():
            ov_ = getattr(o_, f)
            if k < k_g:
                dev = abs(g[f] - ov_) / max(abs(ov_), floor)
                worst[0] = max(worst[0], dev)
                assert dev <= 1e-6, (form, k, f, g[f], ov_)
            else:
                worst[1] = max(worst[1], abs(g[f] - ov_) / max(spread[f], 1e-300))
                assert abs(g[f] - ov_) <= 4.0 * spread[f] + 1e-6 * max(abs(ov_), floor), (form, k, f, g[f], ov_, spread[f])
        if k >= k_g:
            assert abs(g["cnt"] - o_.cnt) <= 4 * max(case["cnt_spread"], 1), (form, k)
    for g_, o_, r_ in ((u, case["ou"], case["ou_r"]), (v, case["ov"], case["ov_r"])):
        yard = np.abs(o_ - r_).max()
        assert np.all(np.abs(g_ - o_) <= 4.0 * yard + 1e-6 * np.abs(o_) + 1e-3), (form, np.abs(g_ - o_).max(), yard)
    print("%s: valid-pixel counts equal for the first %d iterations; worst relative deviation before the first crossing "
          "%.2e, after it %.2f x the oracle's own spread" % (form, k_g, worst[0], worst[1]))


WARM_FORMS = {
    "one-kernel iteration": dict(binned=2, fused=2, persist=0),
    "persistent loop kernel": dict(binned=2, fused=2, persist=2),
    "persistent loop kernel as `auto` takes it": dict(),   # (persist = 1: warm-started runs of the one-kernel loop on a context alone)
    "dense slabs, update at the scatter head": dict(binned=2, fused=0, bin_compact=0, bin_split=0),
}


@pytest.mark.parametrize("form", list(WARM_FORMS), ids=lambda f: f.replace(" ", "_").replace(",", "").replace("`", ""))
def test_forced_form_warm_started_against_the_oracle(accel_mod, warm_case, form):
    """A warm-started run (bf_set_model before bf_run) of each loop that treats warm starts specially, against the oracle
    warm-started from the same model: same bars as the cold test."""
    c = warm_case
    H, W, s, sl = c["H"], c["W"], c["s"], c["sl"]
    a = accel_mod.Accel(max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
    for k, v in WARM_FORMS[form].items():
        a.set_option(k, v)
    a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
    a.set_cloud(s, H, W)
    a.set_model(accel_mod.Model(**c["start"].as_dict()))
    persistent = a.get_stat("persistent")
    if "persistent" in form:
        assert persistent == 1, form
    o = a.default_opts()
    o.res_x, o.res_y, o.max_iter, o.trace_cap, o.want_uv = H, W, K, K + 1, 1
    rc, m, info = a.run(o)
    tr = [t_.model.as_dict() for t_ in a.get_trace(K + 1)]
    u, v = a.compute_uv()
    giveups = a.get_stat("persist_giveups")
    a.close()
    otr, spread = c["otr"], c["spread"]
    assert rc == 0 and giveups == 0
    assert abs(info.iterations - len(otr)) <= 1 and abs(info.iterations - c["n_r"]) <= 1, (form, info.iterations, len(otr), c["n_r"])
    if persistent:
        assert info.launches <= 5 * (info.rebins + 2), (form, info.launches, info.rebins)
    L = min(len(tr), len(otr))
    k_g = next((k for k in range(L) if tr[k]["cnt"] != otr[k].model.cnt), L)
    assert k_g >= min(3, L), (form, k_g)
    worst = [0.0, 0.0]
    for k in range(L):
        g, o_ = tr[k], otr[k].model
        for f, floor in FIELDS.items():
            ov_ = getattr(o_, f)
            if k < k_g:
                dev = abs(g[f] - ov_) / max(abs(ov_), floor)
                worst[0] = max(worst[0], dev)
                assert dev <= 1e-6, (form, k, f, g[f], ov_)
            else:
                worst[1] = max(worst[1], abs(g[f] - ov_) / max(spread[f], 1e-300))
                assert abs(g[f] - ov_) <= 4.0 * spread[f] + 1e-6 * max(abs(ov_), floor), (form, k, f, g[f], ov_, spread[f])
        if k >= k_g:
            assert abs(g["cnt"] - o_.cnt) <= 4 * max(c["cnt_spread"], 1), (form, k)
    for g_, o_, r_ in ((u, c["ou"], c["ou_r"]), (v, c["ov"], c["ov_r"])):
        yard = np.abs(o_ - r_).max()
        assert np.all(np.abs(g_ - o_) <= 4.0 * yard + 1e-6 * np.abs(o_) + 1e-3), (form, np.abs(g_ - o_).max(), yard)
    print("%s, warm start: %d iterations (oracle %d / %d); valid-pixel counts equal for the first %d; worst relative deviation "
          "before the first crossing %.2e, after it %.2f x the oracle's own spread" %
          (form, info.iterations, len(otr), c["n_r"], k_g, worst[0], worst[1]))
