#!/usr/bin/env python3
"""bench.py -- motion-compensation throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one synthetic slice that is already resident
in HBM: staging (AccelLib::init_gpu) -> set_cloud -> the fused OptimizerRolling::run
gradient-descent loop to the reference loop's own termination (cold start, STM off) ->
final warp + per-event (u, v).  The workload is BASELINE.json configs[1]: 1M-event 30 ms
slice, 346x260, scale 3.

    python bench.py --gpus N --steps K --warmup W

For N > 1 the driver launches one rank per GPU with torch.distributed.run; slices are
independent, so ranks shard slices with NO data-path collective and the
job is weak-scaled.  The rendezvous is only used for the timing barrier / max-over-ranks.

Prints ONE JSON line (rank 0).  `roofline` is the warp+scatter kernel's algorithmic bytes
(28 B per event-iteration) over its hipEvent-measured duration;
`cpu_baseline` is the CPU oracle (oracle/, a port of the reference path) timed on a
bounded sample on this host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

METRIC = "Mevents/s motion-compensated (warp→converged score), 1M-ev slice, 1/2/4/8 GPU"
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
K1_BYTES_PER_EVENT_ITER = 28.0  # fr_x, fr_y, t (12 B) + previous pr (16 B)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--events", type=int, default=1000000)
    ap.add_argument("--height", type=int, default=260)
    ap.add_argument("--width", type=int, default=346)
    ap.add_argument("--scale", type=int, default=3)
    ap.add_argument("--slices", type=int, default=4, help="distinct resident slices per rank")
    ap.add_argument("--poll", type=int, default=8)
    ap.add_argument("--config", type=int, default=2, choices=(2, 5),
                    help="BASELINE.json config: 2 = 1M-event 346x260 slice (default, the metric's config); "
                         "5 = 1M-event 1280x720 slices (the 8-GPU farm geometry)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=60)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        # torch first: its bundled libamdhip64.so.7 is then the one HIP runtime of the process
        import torch  # noqa: F401
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # No data-path collective exists on this path; the process group only carries the
        # timing barrier and two scalar reductions, so the CPU (gloo) backend is enough.
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    import numpy as np
    from better_flow_amd import accel, synth

    if args.config == 5:
        args.height, args.width = 720, 1280
    H, W, s = args.height, args.width, args.scale
    ndev = accel.device_count()
    if ndev <= 0:
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    device = local_rank % ndev
    slices = [synth.make_slice(args.events, H, W, 0.030, seed=1 + rank * 1000 + i)
              for i in range(args.slices)]
    nmax = max(len(sl["t"]) for sl in slices)
    acc = accel.Accel(device=device, max_events=nmax, max_rows=s * H + s, max_cols=s * W + s)
    resident = []
    for sl in slices:
        resident.append((acc.to_device(sl["fr_x"]), acc.to_device(sl["fr_y"]),
                         acc.to_device(sl["t"].astype(np.int32)), len(sl["t"])))
    opts = acc.default_opts()
    opts.res_x, opts.res_y, opts.poll_interval, opts.want_uv = H, W, args.poll, 1

    def step(i, warm_model=None, max_iter=-1):
        dx, dy, dt, n = resident[i % len(resident)]
        acc.upload_events_device(dx, dy, dt, n)
        acc.set_cloud(s, H, W)
        if warm_model is not None:
            acc.set_model(warm_model)
        opts.max_iter = max_iter
        rc, m, info = acc.run(opts)
        return n, m, info

    def barrier():
        acc.synchronize()
        if dist is not None:
            dist.barrier()

    # ---- timed region: exactly K cold-start steps -------------------------------------
    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    events = 0
    iters = 0
    for i in range(args.steps):
        n, m, info = step(i)
        events += n
        iters += info.iterations
    acc.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ee = torch.tensor([float(events), float(iters)], dtype=torch.float64)
        dist.all_reduce(ee, op=dist.ReduceOp.SUM)
        dist.barrier()
        elapsed = float(tt[0])
        events_all, iters_all = float(ee[0]), float(ee[1])
    else:
        events_all, iters_all = float(events), float(iters)

    # ---- other regimes (untimed extras; N = 1 semantics per rank) ----------------------
    regimes = {}
    if rank == 0:
        # warm: consecutive slices of one stream, each started from the previous model (STM)
        _, m_prev, _ = step(0)
        acc.synchronize()
        t1 = time.perf_counter()
        wev = wit = 0
        reps = max(4, min(args.steps, 16))
        for i in range(1, 1 + reps):
            n, m_prev, info = step(i, warm_model=m_prev)
            wev += n
            wit += info.iterations
        acc.synchronize()
        dtw = time.perf_counter() - t1
        regimes["warm_stm"] = {"mevents_per_s": wev / dtw / 1e6, "iterations_per_slice": wit / reps,
                               "ms_per_slice": 1e3 * dtw / reps}
        # capped: the reference's real-time setting max_iter = 10 (ros bf_visualizer.cpp:103)
        acc.synchronize()
        t1 = time.perf_counter()
        cev = cit = 0
        for i in range(reps):
            n, _, info = step(i, max_iter=10)
            cev += n
            cit += info.iterations
        acc.synchronize()
        dtc = time.perf_counter() - t1
        regimes["capped_max_iter_10"] = {"mevents_per_s": cev / dtc / 1e6,
                                         "iterations_per_slice": cit / reps,
                                         "ms_per_slice": 1e3 * dtc / reps}

    # ---- roofline of the dominant kernel (warp+scatter), hipEvent-bracketed launches -----
    roofline = None
    if rank == 0:
        acc.profile_enable(1)
        acc.profile_reset()
        psteps = min(args.steps, 4)
        for i in range(psteps):
            step(i)
        p = acc.profile_get()
        acc.profile_enable(0)
        k1_s = p.warp_scatter_ms * 1e-3 / max(1, p.warp_scatter_launches)
        ev_per_launch = p.warp_scatter_events / max(1, p.warp_scatter_launches)
        achieved = K1_BYTES_PER_EVENT_ITER * ev_per_launch / k1_s / 1e9
        try:
            copy_gbps = acc.copy_bandwidth(1 << 30, 5)
        except Exception:
            copy_gbps = None
        # HBM traffic per launch from the PMC passes (profiles/k1_traffic.json is written by
        # scripts/collect_profiles.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs;
        # FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction).  null if not collected
        # or not for this workload.
        traffic = None
        tj = os.path.join(ROOT, "profiles", "k1_traffic.json")
        if os.path.exists(tj) and (H, W, s, args.events) == (260, 346, 3, 1000000):
            t_ = json.load(open(tj))
            traffic = (2.0 * t_["fetch_kb"] + t_["write_kb"]) * 1024.0
        roofline = {
            "bound": "hbm", "kernel": "k_bin_warp_scatter (warp + tile-binned LDS scatter)", "achieved": achieved,
            "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
            "avg_launch_us": k1_s * 1e6, "launches": int(p.warp_scatter_launches),
            "algorithmic_bytes_per_launch": K1_BYTES_PER_EVENT_ITER * ev_per_launch,
            "measured_copy_ceiling_gbps": copy_gbps,
            "per_kernel_us": {
                "warp_scatter": 1e3 * p.warp_scatter_ms / max(1, p.warp_scatter_launches),
                "stencil_moments_update": 1e3 * p.stencil_ms / max(1, p.stencil_launches),
            },
            "note": "durations are hipEvent-bracketed launches on the ctx stream (adds ~1.5 us per launch over "
                    "rocprofv3's kernel time, see profiles/)",
        }

    # ---- CPU baseline: the oracle (port of the reference path), rank 0 at N = 1 only -------
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        sl = slices[0]
        oc = oracle.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
        ow = oc.set_cloud(s, H, W)
        om = oracle.Model()
        tc = time.perf_counter()
        _, oloop, _ = oc.run(ow, om, max_iter=args.cpu_iters - 1, res_x=H, res_y=W)
        dtc = time.perf_counter() - tc
        per_iter = dtc / max(1, oloop.itercount)
        full_iters = iters / max(1, args.steps)          # the GPU run's iterations per slice
        cpu_baseline = {
            "value": len(sl["t"]) / (per_iter * full_iters) / 1e6, "unit": "Mevents/s",
            "cores": 1, "kind": "port",
            "sample": "first %d iteration_steps of the same %d-event cold run (%.1f s, %.1f ms/iteration), "
                      "extrapolated to the %.0f iterations the full run takes" %
                      (oloop.itercount, len(sl["t"]), dtc, 1e3 * per_iter, full_iters),
            "ms_per_iteration": 1e3 * per_iter,
        }

    if rank == 0:
        out = {
            "metric": METRIC,
            "value": events_all / elapsed / 1e6,
            "unit": "Mevents/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%d-event 30 ms slice, %dx%d, scale %d, global-flow gradient descent, "
                            "cold start (STM off) to the reference loop's own termination; one slice "
                            "per step per GPU, slices resident in HBM" % (args.events, W, H, s),
                "events_per_slice": events_all / (args.steps * world),
                "iterations_per_slice": iters_all / (args.steps * world),
                "event_iterations_per_s": events_all / (args.steps * world) * iters_all / elapsed,
                "parallelism": "slice-parallel x%d, no collectives" % world,
            },
            "regimes": regimes,
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    acc.close()


if __name__ == "__main__":
    main()
