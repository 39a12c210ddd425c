"""BASELINE config 4 at its full size: a 32 x 32 grid of independent optimizers (bf_run_tiles) over the 1M-event
346x260 slice of config 2, every tile against its own oracle run.

The reference models this as a queue of (events, model) tasks, one OptimizerRolling per task (dvs_flow.h:200-231); its
guards (optimizer_rolling.h:49-58: window >= scale * RES / 15, >= 1000 events) would skip every 8 x 10-pixel tile, so they
are overridden -- min_events 256, RES = the tile size -- exactly as scripts/config4_tiles.py and bench.py --config 4 do.
This is another regime than the 8 x 8 grid of tests/test_gpu_parity.py: ~1000 events on ~720 image pixels per tile, loops
of 20 .. 3000 iterations, one straggler tile far above the mean.

Bars.  Return code (0 / 1 skipped / NOCONV) equal for every tile.  Iteration count within +-1 and per-event flow within
1e-4 relative / 0.02 px/s against the oracle on the tile's events in upload order -- or, for a tile whose
ORACLE run is itself sensitive to the order of its events (the reference's f32 time sums are order dependent,
accel_lib.h:162; a loop on ~1000 events amplifies that into different iteration counts), the same bar against the
oracle on SOME permutation of the tile's events (reversed, or one of 120 seeded random ones).  The number of tiles that
need a permutation is printed and bounded.
"""
import numpy as np
import pytest

from better_flow_amd import synth

pytestmark = pytest.mark.gpu

H, W, S, G = 260, 346, 3, 32
MIN_EVENTS, HARD_CAP = 256, 20000
GUARD = (H // G, W // G)


def tile_ids(sl):
    tr = np.minimum(sl["fr_x"].astype(np.int64) * G // H, G - 1)
    tc = np.minimum(sl["fr_y"].astype(np.int64) * G // W, G - 1)
    return tr * G + tc


def oracle_tile(oracle_lib, sl, sel, max_iter=-1):
    oc = oracle_lib.Cloud(sl["fr_x"][sel], sl["fr_y"][sel], sl["t"][sel])
    ow = oc.set_cloud(S, H, W)
    om = oracle_lib.Model()
    rc, loop, _ = oc.run(ow, om, max_iter=max_iter, res_x=GUARD[0], res_y=GUARD[1], min_events=MIN_EVENTS, hard_cap=HARD_CAP)
    u, v = oc.compute_uv()
    return rc, int(loop.itercount), u, v


def agrees(oracle_lib, sl, order, gu, gv, git):
    """The bar of SURVEY 8(d) for one tile against the oracle on its events in `order` (gu, gv in that order): iteration
    count within +-1; per-event flow within 1e-4 relative / 0.02 px/s -- widened, when the counts differ by one, by the
    size of the oracle's own last step (flow after its last iteration minus flow one iteration earlier): "the worst case
    if the GPU run terminates one iteration earlier / later than the CPU run".  On a 8 x 10-pixel tile that step is not
    the ~0.008 px/s of a full-sensor slice: the rotation / divergence terms of a terminal step reach 0.03 px/s here."""
    rc, it, u, v = oracle_tile(oracle_lib, sl, order)
    if rc != 0 or abs(git - it) > 1:
        return False, it, np.inf
    tol_u, tol_v = np.maximum(1e-4 * np.abs(u), 0.02), np.maximum(1e-4 * np.abs(v), 0.02)
    if git != it and it >= 3:
        _, it1, u1, v1 = oracle_tile(oracle_lib, sl, order, max_iter=it - 2)   # max_iter = K runs K + 1 iterations
        assert it1 == it - 1
        tol_u, tol_v = tol_u + np.abs(u - u1).max(), tol_v + np.abs(v - v1).max()   # (the largest step on the tile: a scalar)
    dev = max(np.abs(gu - u).max(), np.abs(gv - v).max())
    return bool(np.all(np.abs(gu - u) <= tol_u) and np.all(np.abs(gv - v) <= tol_v)), it, dev


def test_config4_full_size_every_tile_against_its_oracle(oracle_lib, accel_mod):
    sl = synth.make_slice(1000000, H, W, 0.030, seed=1)
    n = len(sl["t"])
    acc = accel_mod.Accel(max_events=n, max_rows=S * H + S, max_cols=S * W + S)
    acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
    models, infos = acc.run_tiles(G, G, S, (H, W), GUARD, min_events=MIN_EVENTS, hard_iter_cap=HARD_CAP)
    u, v = acc.compute_uv()
    tid = tile_ids(sl)
    order = np.argsort(tid, kind="stable")
    bounds = np.searchsorted(tid[order], np.arange(G * G + 1))
    ran = skipped = noconv = second_bar = 0
    worst = 0.0
    it_g, it_o = [], []
    for k in range(G * G):
        sel = order[bounds[k]:bounds[k + 1]]
        orc, oit, ou, ov = oracle_tile(oracle_lib, sl, sel)
        orc = accel_mod.BF_ERR_NOCONV if orc < 0 else orc
        assert infos[k].rc == orc, (k, infos[k].rc, orc, len(sel))
        if orc == 1:
            skipped += 1
            assert not u[sel].any() and not v[sel].any(), k
            continue
        if orc != 0:
            noconv += 1
            continue
        ran += 1
        git = infos[k].iterations
        it_g.append(git)
        it_o.append(oit)
        ok, _, dev = agrees(oracle_lib, sl, sel, u[sel], v[sel], git)
        if ok:
            worst = max(worst, dev)
            continue
        # The reference's result depends on the ORDER of a tile's events (f32 running time sums, accel_lib.h:162), and a
        # loop on ~1000 events amplifies that: the oracle itself ends after 56, 119 or 32 iterations on tile 291, depending
        # on the permutation.  The GPU's integer sums are order free, so it must reproduce the reference's answer for SOME
        # order: the same bar against the reversed order or one of 120 seeded random permutations (a branch the oracle
        # takes on 3 of 40 permutations -- tile 649 -- is missed by 120 draws once in 10 000 runs).
        rng = np.random.default_rng(1000 + k)
        hit = None
        for trial in range(121):
            perm = np.arange(len(sel))[::-1].copy() if trial == 0 else rng.permutation(len(sel))
            if agrees(oracle_lib, sl, sel[perm], u[sel][perm], v[sel][perm], git)[0]:
                hit = trial
                break
        assert hit is not None, \
            "tile %d: GPU %d iterations, oracle %d (upload order), flow off by %.3e px/s -- and no permutation of the tile's events makes the oracle agree" % (k, git, oit, dev)
        second_bar += 1
    acc.close()
    it_g, it_o = np.array(it_g), np.array(it_o)
    print("config 4 full size: %d tiles optimised, %d skipped, %d not converged; iterations mean %.1f max %d (oracle %.1f / %d); "
          "%d tiles equal in iteration count, %d off by one, %d matched by the oracle on a permutation of their events; "
          "worst flow deviation under the first bar %.3e px/s" %
          (ran, skipped, noconv, it_g.mean(), it_g.max(), it_o.mean(), it_o.max(), int((it_g == it_o).sum()),
           int((np.abs(it_g - it_o) == 1).sum()), second_bar, worst))
    assert ran >= 900 and skipped >= 1, (ran, skipped)
    assert it_g.max() > 1000, "the straggler regime (one tile far above the mean) must be part of the test"
    # (13 of 957 need the second bar today; bounded at 2 %, not at a comfortable multiple)
    assert second_bar <= ran // 50, second_bar


def test_config4_many_slices_in_one_launch_same_bits(accel_mod):
    """bf_run_tiles_many: K = 4 slices' 32 x 32 grids in ONE launch (work-groups claim (slice, tile) pairs from a device
    counter).  Every tile of every slice must be the bits of bf_run_tiles on that slice alone -- return code, iteration
    count, dividers, model, per-event flow -- and slice 0 is the slice the test above holds to the oracle tile by tile.
    Twice in a row on the same contexts (the claim counter and the tiles' states are re-armed), and with the contexts in
    another order."""
    K = 4
    slices = [synth.make_slice(1000000, H, W, 0.030, seed=1 + k) for k in range(K)]
    nmax = max(len(sl["t"]) for sl in slices)

    def canon(models, infos):
        return [(i.rc, i.iterations, i.x_divider, i.y_divider, i.rot_divider, i.div_divider,
                 tuple(np.float64(getattr(m, f)).tobytes() for f, _ in m._fields_ if f != "_pad")) for m, i in zip(models, infos)]
    single = []
    for sl in slices:
        acc = accel_mod.Accel(max_events=nmax, max_rows=S * H + S, max_cols=S * W + S)
        acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
        models, infos = acc.run_tiles(G, G, S, (H, W), GUARD, min_events=MIN_EVENTS, hard_iter_cap=HARD_CAP)
        u, v = acc.compute_uv()
        single.append((canon(models, infos), u.tobytes(), v.tobytes()))
        acc.close()
    assert max(i[1] for i in single[0][0]) > 1000, "the straggler regime must be part of the test"
    accs = [accel_mod.Accel(max_events=nmax, max_rows=S * H + S, max_cols=S * W + S) for _ in range(K)]
    for order in (list(range(K)), list(range(K)), [2, 0, 3, 1]):
        for k in order:
            sl = slices[k]
            accs[k].upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
        out = accel_mod.run_tiles_many([accs[k] for k in order], G, G, S, (H, W), GUARD, min_events=MIN_EVENTS, hard_iter_cap=HARD_CAP)
        for pos, k in enumerate(order):
            models, infos = out[pos]
            assert canon(models, infos) == single[k][0], (order, k)
            u, v = accs[k].compute_uv()
            assert u.tobytes() == single[k][1] and v.tobytes() == single[k][2], (order, k)
    # errors are loud: a context twice, an empty list
    with pytest.raises(accel_mod.BfError):
        accel_mod.run_tiles_many([accs[0], accs[0]], G, G, S, (H, W), GUARD, min_events=MIN_EVENTS)
    for a in accs:
        a.close()


def _tile_bounds(res, grid, k):
    lo = (k * res + grid - 1) // grid
    hi = ((k + 1) * res + grid - 1) // grid - 1
    return lo, hi


@pytest.mark.parametrize("case", [(200000, 180, 240, 3, 8, 8, 30, 31), (1000000, 260, 346, 3, 32, 32, 11, 1), (60000, 180, 240, 5, 4, 6, 24, 7),
                                  (50000, 180, 240, 1, 6, 6, 40, 9)],
                         ids=["8x8_s3", "config4_32x32_s3", "4x6_s5", "6x6_s1"])
def test_local_window_grid_every_window_equals_its_oracle_run(oracle_lib, accel_mod, case):
    """bf_local_run_tiles -- SURVEY f1's formulation of config 4: a grid of OptimizerLocal windows (optimizer_sampler.h:31-34), each
    on its tile's events, the whole coordinate descent of run() (optimizer_sampler.cpp:4-38) on chip.  Every window's final state
    -- nx, ny, last score, both steps, the evaluation count -- and return code must EQUAL the oracle's run on the tile's events
    with the window the entry point documents (centre = middle of the tile, t = 0): integers and single IEEE operations only."""
    n, Hh, Ww, s, gr, gc, wsz, seed = case
    sl = synth.make_slice(n, Hh, Ww, 0.030, seed=seed)
    acc = accel_mod.Accel(max_events=len(sl["t"]), max_rows=s * Hh + s, max_cols=s * Ww + s)
    acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
    guard = (max(1, Hh // gr), max(1, Ww // gc))
    states, rcs = acc.local_run_tiles(gr, gc, s, wsz, (Hh, Ww), guard, max_evaluations=4000)
    tr = np.minimum(sl["fr_x"].astype(np.int64) * gr // Hh, gr - 1)
    tc = np.minimum(sl["fr_y"].astype(np.int64) * gc // Ww, gc - 1)
    tid = tr * gc + tc
    order = np.argsort(tid, kind="stable")
    bounds = np.searchsorted(tid[order], np.arange(gr * gc + 1))
    evals = []
    # (every window of the small grids; of the 32 x 32 grid every third one -- the oracle is the slow side)
    for k in range(0, gr * gc, 3 if gr * gc > 500 else 1):
        sel = order[bounds[k]:bounds[k + 1]]
        xl, xh = _tile_bounds(Hh, gr, k // gc)
        yl, yh = _tile_bounds(Ww, gc, k % gc)
        oc = oracle_lib.Cloud(sl["fr_x"][sel], sl["fr_y"][sel], sl["t"][sel])
        ow = oc.local_window(s, center=((xl + xh) // 2, (yl + yh) // 2, 0), wsz=wsz)
        orc, ost, _ = oc.local_run(ow, res_x=guard[0], res_y=guard[1], max_evaluations=4000)
        orc = accel_mod.BF_ERR_NOCONV if orc < 0 else orc
        g = states[k]
        assert rcs[k] == orc, (k, rcs[k], orc)
        for f in ("nx", "ny", "last_score", "dnx", "dny", "dn_th", "evaluations"):
            assert getattr(g, f) == getattr(ost, f), (k, f, getattr(g, f), getattr(ost, f), len(sel))
        evals.append(g.evaluations)
    nx = np.array([st.nx for st, r in zip(states, rcs) if r == 0])
    ny = np.array([st.ny for st, r in zip(states, rcs) if r == 0])
    # Event::project: pr = fr - (n / 127) t / 10000 with t in ns -- a flow of v px/s is compensated by n = 127e-5 v
    print("%s: %d windows, evaluations mean %.1f max %d; median (nx, ny) = (%.3f, %.3f), the injected flow corresponds to (%.3f, %.3f)"
          % (case, len(states), np.mean(evals), max(evals), np.median(nx), np.median(ny), 127e-5 * sl["velocity"][0], 127e-5 * sl["velocity"][1]))
    if s == 3:
        assert max(evals) > 19                               # (some window does more than halve its steps from the first evaluation on)
    # the per-event state of the rolling optimizer is untouched territory afterwards: a fresh slice works as before
    acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
    acc.set_cloud(s, Hh, Ww)
    rc, m, info = acc.run()
    assert rc == 0 and info.iterations > 10
    acc.close()


def test_tile_grids_on_empty_and_tiny_slices(oracle_lib, accel_mod):
    """Edge cases of the grid entry points: an EMPTY slice and a slice of a handful of events among the slices of one
    bf_run_tiles_many launch (every tile of the empty one is skipped by the guard, the others are untouched by its presence),
    and bf_local_run_tiles on them -- an OptimizerLocal on an empty cloud is not an error in the reference: every score is 0, every
    step is halved until the threshold, and the oracle does exactly that."""
    g = 8
    full = synth.make_slice(120000, 180, 240, 0.030, seed=77)
    tiny = {k: (v[:5].copy() if hasattr(v, "__len__") and not isinstance(v, tuple) else v) for k, v in full.items()}
    empty = {"fr_x": np.zeros(0, np.int32), "fr_y": np.zeros(0, np.int32), "t": np.zeros(0, np.int64)}
    guard = (180 // g, 240 // g)
    ref = accel_mod.Accel(max_events=len(full["t"]), max_rows=3 * 180 + 3, max_cols=3 * 240 + 3)
    ref.upload_events(full["fr_x"], full["fr_y"], full["t"])
    rm, ri = ref.run_tiles(g, g, 3, (180, 240), guard, min_events=256, hard_iter_cap=20000)
    ref.close()
    accs = [accel_mod.Accel(max_events=len(full["t"]), max_rows=3 * 180 + 3, max_cols=3 * 240 + 3) for _ in range(3)]
    for a, sl in zip(accs, (empty, full, tiny)):
        a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
    out = accel_mod.run_tiles_many(accs, g, g, 3, (180, 240), guard, min_events=256, hard_iter_cap=20000)
    assert all(i.rc == 1 and i.iterations == 0 for i in out[0][1]) and all(i.rc == 1 for i in out[2][1])
    assert [(i.rc, i.iterations) for i in out[1][1]] == [(i.rc, i.iterations) for i in ri]
    assert [m.as_dict() for m in out[1][0]] == [m.as_dict() for m in rm]
    assert any(i.rc == 0 for i in ri)
    for a, sl in zip(accs, (empty, full, tiny)):
        a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
        states, rcs = a.local_run_tiles(g, g, 3, 30, (180, 240), guard, max_evaluations=4000)
        tr = np.minimum(sl["fr_x"].astype(np.int64) * g // 180, g - 1)
        tc = np.minimum(sl["fr_y"].astype(np.int64) * g // 240, g - 1)
        tid = tr * g + tc
        for k in (0, 9, 27, 63):
            sel = np.nonzero(tid == k)[0]
            xl, xh = _tile_bounds(180, g, k // g)
            yl, yh = _tile_bounds(240, g, k % g)
            oc = oracle_lib.Cloud(sl["fr_x"][sel], sl["fr_y"][sel], sl["t"][sel])
            ow = oc.local_window(3, center=((xl + xh) // 2, (yl + yh) // 2, 0), wsz=30)
            orc, ost, _ = oc.local_run(ow, res_x=guard[0], res_y=guard[1], max_evaluations=4000)
            assert rcs[k] == orc
            for f in ("nx", "ny", "last_score", "dnx", "dny", "evaluations"):
                assert getattr(states[k], f) == getattr(ost, f), (len(sl["t"]), k, f)
        a.close()
