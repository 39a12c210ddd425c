"""The one-kernel iteration (k_fused_pass, bf_fused.hip): warp + scatter + stencil + moments of one image tile per
work-group, events of a tile's edge strips read by the neighbouring tiles' work-groups as well.

It must return the bits of the two-kernel tile-binned loop (same integer accumulators, same per-sub-tile f64 partials) --
model, iteration count, every trace record, per-event flow, and the warm start that follows -- with the default margin,
with margins so small that events outrun their bins (the `lost` flag, a re-bin, the pass repeated before its update),
with 64-row tiles (the 640x480 case: too many 32-row tiles for the counting sort), with the unpacked LDS planes, without the predictive re-bin -- and so must its persistent form
(k_fused_loop, bf_loop.hip: many iterations per launch, sums exchanged through tagged records), forced for the cold run
as well ("persist" = 2; by default only the warm start takes it).  And `auto` must take the one-kernel loop exactly for
the slices it was measured to be faster on (bf_set_cloud).
"""
import numpy as np
import pytest

from better_flow_amd import synth
from helpers import make_accel

pytestmark = pytest.mark.gpu


def run(accel_mod, sl, H, W, s, opts, max_iter=-1):
    a = make_accel(accel_mod, opts, max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)   # ("debug_margin": tests/helpers.py)
    a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
    a.set_cloud(s, H, W)
    persistent = a.get_stat("persistent")
    o = a.default_opts()
    o.res_x, o.res_y, o.want_uv, o.trace_cap, o.max_iter = H, W, 1, 4096, max_iter
    rc, m, info = a.run(o)
    trace = [t.model.as_dict() for t in a.get_trace(4096)]
    u, v = a.compute_uv()
    a.set_model(m)
    rc2, m2, info2 = a.run(o)
    trace2 = [t.model.as_dict() for t in a.get_trace(4096)]
    u2, v2 = a.compute_uv()
    giveups = a.get_stat("persist_giveups")
    a.close()
    return dict(rc=(rc, rc2), it=(info.iterations, info2.iterations), model=(m.as_dict(), m2.as_dict()), trace=(trace, trace2),
                flow=(u.tobytes(), v.tobytes(), u2.tobytes(), v2.tobytes()), rebins=info.rebins, launches=info.launches, persistent=persistent,
                giveups=giveups)


VARIANTS = (("default margin", {}), ("persistent kernel", {"persist": 2}), ("persistent kernel, margin 2", {"persist": 2, "debug_margin": 2}),
            ("persistent kernel, unpacked planes", {"persist": 2, "bin_pack_limit": 20}), ("margin 2", {"debug_margin": 2}), ("margin 1", {"debug_margin": 1}),
            ("unpacked planes", {"bin_pack_limit": 20}), ("no predictive re-bin", {"bin_predict": 0}),
            ("no predictive re-bin, margin 3", {"bin_predict": 0, "debug_margin": 3}))


@pytest.mark.parametrize("case", [(1000000, 260, 346, 3, 1, -1), (300000, 480, 640, 3, 2, -1), (200000, 180, 240, 5, 3, -1),
                                  (50000, 180, 240, 1, 4, -1), (50000, 260, 346, 3, 6, -1), (3000000, 480, 640, 3, 8, 60)],
                         ids=lambda c: "%dev_%dx%d_s%d" % (c[0], c[2], c[1], c[3]))
def test_same_bits_as_the_two_kernel_loop(accel_mod, case):
    n, H, W, s, seed, max_iter = case
    sl = synth.make_slice(n, H, W, 0.03, seed=seed)
    ref = run(accel_mod, sl, H, W, s, {"binned": 2, "fused": 0}, max_iter)
    assert ref["rc"][0] == 0 and ref["it"][0] > 20
    for name, o in VARIANTS:
        got = run(accel_mod, sl, H, W, s, dict({"binned": 2, "fused": 2, "persist": 0}, **o), max_iter)
        if name == "default margin":   # (tight margins spend launches waiting for re-bins)
            assert got["launches"] < 0.75 * ref["launches"], "the one-kernel loop was not the one that ran"
        if name == "persistent kernel" and (s * H + s + 31) // 32 * ((s * W + s + 63) // 64) <= 256:
            assert got["persistent"] == 1   # (all tiles resident at once; much larger images fall back to one launch per iteration)
        if name == "persistent kernel" and got["persistent"]:   # (five launches per round -- re-bin trio, loop kernel, gated final warp --, a round per re-bin)
            assert got["launches"] <= 5 * (got["rebins"] + 2) + 8, "the persistent kernel was not the one that ran"
        for key in ("rc", "it", "model", "trace", "flow"):
            assert got[key] == ref[key], (name, key)
        if name == "margin 1":
            assert got["rebins"] > ref["rebins"], "margin 1 must exercise the lost -> re-bin -> repeat path"


def test_auto_takes_it_where_it_was_measured_faster(accel_mod):
    """One launch per iteration (plus re-bins) tells which loop ran."""
    for (n, H, W, want_fused) in ((50000, 180, 240, True), (50000, 260, 346, True), (200000, 260, 346, True), (1000000, 260, 346, False),
                                  (100000, 480, 640, False)):
        sl = synth.make_slice(n, H, W, 0.03, seed=9)
        a = accel_mod.Accel(max_events=len(sl["t"]), max_rows=3 * H + 3, max_cols=3 * W + 3)
        a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
        a.set_cloud(3, H, W)
        o = a.default_opts()
        o.res_x, o.res_y, o.max_iter = H, W, 40
        rc, m, info = a.run(o)
        one_kernel = info.launches < 1.5 * info.iterations + 3 * info.rebins + 8
        assert one_kernel == want_fused, (n, H, W, info.launches, info.iterations, info.rebins)
        # ... and a context that shares the GPU with others takes it for sparser slices only (one event per eight pixels)
        a.set_option("co_schedule", 1)
        a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
        w = a.set_cloud(3, H, W)
        rc2, m2, info2 = a.run(o)
        assert m2.as_dict() == m.as_dict() and info2.iterations == info.iterations
        shared = want_fused and 8 * len(sl["t"]) <= w.scale_img_x * w.scale_img_y
        assert (info2.launches < 1.5 * info2.iterations + 3 * info2.rebins + 8) == shared, (n, H, W, info2.launches, info2.iterations)
        a.close()


def test_persistent_kernel_that_gives_up_undoes_itself(accel_mod):
    """A work-group of the persistent kernel that waits 0.2 s for records that never come (another process's kernels hold part
    of the CUs) makes the launch give up: nobody stores products or state, and bf_run carries on with one launch per
    iteration from where that launch started.  BF_DEBUG_PERSIST_ABORT=<pass> makes every work-group "time out" at that pass:
    the results must be the bits of the other loops -- for single-pass lists (events in registers) and multi-pass lists
    (private product arrays), cold and warm, giving up in the first launch and in a later one."""
    for (n, H, W, s, seed) in ((50000, 180, 240, 3, 4), (300000, 260, 346, 3, 6)):
        sl = synth.make_slice(n, H, W, 0.03, seed=seed)
        ref = run(accel_mod, sl, H, W, s, {"binned": 2, "fused": 2, "persist": 0})
        for at in (0, 3, 17):
            got = run(accel_mod, sl, H, W, s, {"binned": 2, "fused": 2, "persist": 2, "BF_DEBUG_PERSIST_ABORT": str(at)})   # (test build: tests/helpers.py)
            assert got["persistent"] == 1
            assert got["launches"] > ref["it"][0] // 2, "the fall-back (one launch per iteration) must have run"
            for key in ("rc", "it", "model", "trace", "flow"):
                assert got[key] == ref[key], (n, at, key)
        # ... and the real thing: ONE work-group goes silent (BF_DEBUG_PERSIST_MUTE=<pass>: the last work-group stops publishing
        # its records from that pass on, as if it had lost its CU).  Only the reducer in charge of its records times out; it
        # must not publish its partial total -- the others would take it for the sum -- and the whole launch gives up.
        for at in (0, 5):
            got = run(accel_mod, sl, H, W, s, {"binned": 2, "fused": 2, "persist": 2, "BF_DEBUG_PERSIST_MUTE": str(at)})   # (test build: tests/helpers.py)
            assert got["launches"] > ref["it"][0] // 2, "the fall-back (one launch per iteration) must have run"
            assert got["giveups"] >= 1
            for key in ("rc", "it", "model", "trace", "flow"):
                assert got[key] == ref[key], (n, "mute", at, key)
        # ... and the split decision: ONE work-group alone times out on a launch's LAST pass (BF_DEBUG_PERSIST_SPLIT=<pass>: every
        # launch ends after that many passes), the others do not.  Keeping or undoing the launch is one compare-and-swap on
        # the verdict word (bf_loop.hip: verdict_decide).  At once: its ABORT lands before the others' COMMIT, everybody
        # undoes, the fall-back runs.  Late (~100 us): the others have committed, the straggler reads the reduced records
        # again and leaves with them -- no give-up, and no work-group that silently kept its pre-launch products.
        for spec, undone in (("7", True), ("7,late", False), ("0,late", False)):
            got = run(accel_mod, sl, H, W, s, {"binned": 2, "fused": 2, "persist": 2, "BF_DEBUG_PERSIST_SPLIT": spec})
            assert got["persistent"] == 1
            if undone:
                assert got["giveups"] >= 1 and got["launches"] > ref["it"][0] // 2, (spec, got["giveups"], got["launches"])
            else:
                assert got["giveups"] == 0, (spec, got["giveups"])
            for key in ("rc", "it", "model", "trace", "flow"):
                assert got[key] == ref[key], (n, "split", spec, key)
