"""BASELINE config 4: a 32x32 grid of independent local optimizers (bf_run_tiles: one work-group per sensor tile runs its
whole gradient-descent loop on chip) over 1M-event 346x260 slices.

    python scripts/config4_tiles.py [min_events] [--grids G] [--reps R]

Reports the latency of ONE grid (the slice is done when its slowest tile is: one 638-event tile needs ~3000 iterations
while the mean is ~90) and the SUSTAINED rate with G grids in flight -- G slice contexts (host thread + bf_ctx + HIP
stream each) working through a queue of slices, so that one grid's straggler tile runs under the other grids' bulk; the
reference would queue these (events, model) tasks one after the other (dvs_flow.h:200-231).

A grid's launch lasts as long as its slowest tile (~21 ms) while its bulk is done in under a millisecond, so the sustained rate
is "grids in flight / 21 ms" until the GPU's slots are full -- and the HIP runtime gives a process FOUR hardware queues
(GPU_MAX_HW_QUEUES, default 4): the streams of a fifth grid queue behind another grid's straggler.  With the variable at 16
(--hw-queues 16: set here before the first HIP call) and 16 grids in flight: 189 -> 510 Mevents/s (24 / 32: 450 /
382).  The opposite of the iteration loop's dependent 10 us kernels, where more than four queues lose (EXPERIMENTS.md).

--many K (round 6): bf_run_tiles_many -- K slices' grids in ONE launch, work-groups claiming (slice, tile) pairs from a device
counter, batches issued from --many-lanes host threads (2: the next batch's sort and bulk run under the previous batch's
stragglers).  Needs no environment variable: the default leaves GPU_MAX_HW_QUEUES alone."""
import argparse
import json
import os
import sys
import threading
import time

if "--hw-queues" in sys.argv:   # (before the HIP runtime starts: it reads the variable once)
    os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[sys.argv.index("--hw-queues") + 1]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from better_flow_amd import accel, synth  # noqa: E402

N, H, W, s, G = 1000000, 260, 346, 3, 32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("min_events", nargs="?", type=int, default=256)
    ap.add_argument("--grids", type=int, default=16, help="tile grids (slice contexts) in flight for the sustained figure")
    ap.add_argument("--reps", type=int, default=6, help="slices per context in the sustained run")
    ap.add_argument("--slices", type=int, default=4, help="distinct slices")
    ap.add_argument("--hw-queues", type=int, default=0, help="GPU_MAX_HW_QUEUES for this process (0: leave the environment alone)")
    ap.add_argument("--many", type=int, default=32, help="slices per bf_run_tiles_many call (0: skip that measurement)")
    ap.add_argument("--many-lanes", type=int, default=2, help="host threads issuing bf_run_tiles_many batches")
    ap.add_argument("--many-reps", type=int, default=4, help="batches per lane")
    a_ = ap.parse_args()
    guard = (max(1, H // G), max(1, W // G))
    slices = [synth.make_slice(N, H, W, 0.030, seed=1 + k) for k in range(a_.slices)]
    nmax = max(len(sl["t"]) for sl in slices)

    def run_grid(acc, sl):
        acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
        return acc.run_tiles(G, G, s, (H, W), guard, min_events=a_.min_events, hard_iter_cap=20000)

    # ---- one grid alone: latency
    acc = accel.Accel(max_events=nmax, max_rows=s * H + s, max_cols=s * W + s)
    sl = slices[0]
    best = None
    for rep in range(4):
        acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
        acc.synchronize()
        t0 = time.perf_counter()
        models, infos = acc.run_tiles(G, G, s, (H, W), guard, min_events=a_.min_events, hard_iter_cap=20000)
        acc.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    rc = np.array([i.rc for i in infos])
    it = np.array([i.iterations for i in infos])
    u, v = acc.compute_uv()
    acc.close()
    ran = rc == 0
    n = len(sl["t"])
    # ---- several grids in flight: sustained throughput (upload of the slice included, as in the single-grid figure's caller)
    accs = [accel.Accel(max_events=nmax, max_rows=s * H + s, max_cols=s * W + s) for _ in range(a_.grids)]
    for k, c in enumerate(accs):          # warm every context (allocations, code objects)
        run_grid(c, slices[k % len(slices)])
    tot = [[0, 0] for _ in accs]

    def lane(k):
        for r in range(a_.reps):
            sl_ = slices[(k + r) % len(slices)]
            _, inf = run_grid(accs[k], sl_)
            tot[k][0] += len(sl_["t"])
            tot[k][1] += int(sum(i.iterations for i in inf))
        accs[k].synchronize()
    th = [threading.Thread(target=lane, args=(k,)) for k in range(a_.grids)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dts = time.perf_counter() - t0
    for c in accs:
        c.close()
    ev_s, it_s = sum(x[0] for x in tot), sum(x[1] for x in tot)
    many = None
    if a_.many > 0:
        K, LN = a_.many, max(1, a_.many_lanes)
        sets = [[accel.Accel(max_events=nmax, max_rows=s * H + s, max_cols=s * W + s) for _ in range(K)] for _ in range(LN)]

        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=a_.grids)   # the uploads of a batch side by side, as the grids-in-flight form has them
        t_up, t_run = [0.0], [0.0]

        def batch(ln, r):
            def up(k):
                sl_ = slices[(k + r + ln) % len(slices)]
                sets[ln][k].upload_events(sl_["fr_x"], sl_["fr_y"], sl_["t"])
                return len(sl_["t"])
            t0_ = time.perf_counter()
            ev = sum(pool.map(up, range(K)))
            t1_ = time.perf_counter()
            out_ = accel.run_tiles_many(sets[ln], G, G, s, (H, W), guard, min_events=a_.min_events, hard_iter_cap=20000)
            t_up[0], t_run[0] = t1_ - t0_, time.perf_counter() - t1_
            return ev, int(sum(i.iterations for _, inf in out_ for i in inf))
        for ln in range(LN):
            batch(ln, 0)       # warm: allocations, code objects
        t0 = time.perf_counter()
        one_ev, _ = batch(0, 1)
        one_dt = time.perf_counter() - t0
        up_ms, run_ms = t_up[0], t_run[0]
        totm = [[0, 0] for _ in range(LN)]

        def mlane(ln):
            for r in range(a_.many_reps):
                e_, i_ = batch(ln, r)
                totm[ln][0] += e_; totm[ln][1] += i_
        th = [threading.Thread(target=mlane, args=(ln,)) for ln in range(LN)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dtm = time.perf_counter() - t0
        for st_ in sets:
            for c in st_:
                c.close()
        many = {"slices_per_launch": K, "host_lanes": LN, "batches": LN * a_.many_reps, "hw_queues_env": os.environ.get("GPU_MAX_HW_QUEUES", "unset"),
                "one_batch_ms": 1e3 * one_dt, "one_batch_upload_ms": 1e3 * up_ms, "one_batch_run_tiles_many_ms": 1e3 * run_ms,
                "one_batch_mevents_per_s": one_ev / one_dt / 1e6,
                "seconds": dtm, "mevents_per_s": sum(x[0] for x in totm) / dtm / 1e6, "tile_iterations_per_s": sum(x[1] for x in totm) / dtm,
                "ms_per_slice": 1e3 * dtm / (LN * a_.many_reps * K), "uploads": "included (blocking bf_upload_events per slice)"}
    # ---- the same grid as OptimizerLocal windows (bf_local_run_tiles: SURVEY f1's formulation of this config) ----
    accl = accel.Accel(max_events=nmax, max_rows=s * H + s, max_cols=s * W + s)
    wsz = max((H + G - 1) // G, (W + G - 1) // G)
    bestl = None
    for rep in range(4):
        accl.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
        accl.synchronize()
        t0 = time.perf_counter()
        lst, lrc = accl.local_run_tiles(G, G, s, wsz, (H, W), guard, max_evaluations=4000)
        dtl = time.perf_counter() - t0
        bestl = dtl if bestl is None else min(bestl, dtl)
    accl.close()
    lnx = np.array([t_.nx for t_, r_ in zip(lst, lrc) if r_ == 0]); lny = np.array([t_.ny for t_, r_ in zip(lst, lrc) if r_ == 0])
    lev = np.array([t_.evaluations for t_, r_ in zip(lst, lrc) if r_ == 0])
    local = {"what": "bf_local_run_tiles: %dx%d OptimizerLocal windows of %d sensor pixels (scale %d), each on its tile's events, the whole "
                     "coordinate descent on chip" % (G, G, wsz, s),
             "ms": 1e3 * bestl, "mevents_per_s": n / bestl / 1e6, "windows_run": int(len(lev)), "evaluations_mean": float(lev.mean()),
             "evaluations_max": int(lev.max()), "window_evaluations_per_s": float(lev.sum() / bestl),
             # Event::project: pr = fr - (n / 127) t / 10000, t in ns: a flow of v px/s is compensated by n = 127e-5 v
             "flow_median_px_s": [float(np.median(lnx)) / 127e-5, float(np.median(lny)) / 127e-5],
             "flow_within_20pct_of_injected": float(np.mean((np.abs(lnx / 127e-5 - sl["velocity"][0]) < 0.2 * abs(sl["velocity"][0])) &
                                                            (np.abs(lny / 127e-5 - sl["velocity"][1]) < 0.2 * abs(sl["velocity"][1]))))}
    per_iter_us = 1e6 * best / max(1, it.max())
    out = {"config": "4: %dx%d tiles over %d-event %dx%d slices, scale %d, guards: min_events=%d, RES=%dx%d" %
                     (G, G, n, W, H, s, a_.min_events, guard[1], guard[0]),
           "single_grid": {"ms": best * 1e3, "mevents_per_s": n / best / 1e6, "tiles_optimised": int(ran.sum()), "tiles_skipped": int((rc == 1).sum()),
                           "tiles_failed": int((rc < 0).sum()), "iterations_mean": float(it[ran].mean()) if ran.any() else 0,
                           "iterations_max": int(it.max()), "tile_iterations_per_s": float(it.sum() / best),
                           "floor": "the slowest tile's %d iterations x %.2f us per iteration of one work-group = %.1f ms: a grid cannot "
                                    "finish before its slowest tile" % (int(it.max()), per_iter_us, 1e-3 * it.max() * per_iter_us)},
           "local_windows": local,
           "many_slices_per_launch": many,
           "sustained": {"grids_in_flight": a_.grids, "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "unset (runtime default: 4)"), "slices": a_.grids * a_.reps, "seconds": dts, "mevents_per_s": ev_s / dts / 1e6,
                         "tile_iterations_per_s": it_s / dts, "ms_per_slice": 1e3 * dts / (a_.grids * a_.reps)},
           "flow_median_px_s": [float(np.median(u[np.abs(u) > 0])) if (np.abs(u) > 0).any() else 0.0,
                                float(np.median(v[np.abs(v) > 0])) if (np.abs(v) > 0).any() else 0.0],
           "injected_px_s": list(sl["velocity"])}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
