#!/usr/bin/env python3
"""bench.py -- motion-compensation throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one synthetic slice that is already resident
in HBM: staging (AccelLib::init_gpu) -> set_cloud -> the fused OptimizerRolling::run
gradient-descent loop to the reference loop's own termination (cold start, STM off) ->
final warp + per-event (u, v).  The workload is BASELINE.json configs[1]: 1M-event 30 ms
slice, 346x260, scale 3.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU (rank r on device r).  Either the caller launches the ranks
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: WORLD_SIZE is set and
must equal --gpus), or plain `python bench.py --gpus N` spawns them itself (spawn_ranks below:
N copies of this script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, a gloo rendezvous on
127.0.0.1).  Fewer visible devices than ranks is an error unless --oversubscribe (tests: two
ranks on one GPU).  Slices are independent, so ranks shard slices with NO
data-path collective and the job is weak-scaled; the rendezvous only carries the timing
barrier / max-over-ranks.

Prints ONE JSON line (rank 0).  `roofline` prices both loop kernels by SURVEY.md 8(d)'s algorithmic
bytes (28 B per event-iteration for the warp + scatter, 24 B per image pixel for the stencil /
moments) over their own measured launch durations; its top-level achieved / frac / traffic are
the DOMINANT kernel's (the longer average launch), the other one is listed beside it;
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


def host_cores():
    """Host cores this job may use: the affinity mask capped by the container's CPU quota (cgroup v2)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: N ranks of this script, rank r on device r (better_flow_amd/farm.py:
    spawn_local_ranks -- a rank that fails takes the job down with its own exit code, nothing is left running)."""
    from better_flow_amd import farm
    return farm.spawn_local_ranks(n, os.path.abspath(__file__), argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--events", type=int, default=1000000)
    ap.add_argument("--height", type=int, default=260)
    ap.add_argument("--width", type=int, default=346)
    ap.add_argument("--scale", type=int, default=3)
    ap.add_argument("--slices", type=int, default=6, help="distinct resident slices per rank")
    ap.add_argument("--poll", type=int, default=8)
    ap.add_argument("--config", type=int, default=2, choices=(2, 5),
                    help="BASELINE.json config: 2 = 1M-event 346x260 slice (default, the metric's config); "
                         "5 = 1M-event 1280x720 slices (the 8-GPU farm geometry)")
    ap.add_argument("--concurrent", type=int, default=4,
                    help="independent slices in flight per GPU: one host thread + bf_ctx + HIP stream each "
                         "(the slice farm of SURVEY 8(e) applied inside one GPU; a step = this many slices)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-front-end", action="store_true", help="skip the command-line front-end measurement")
    ap.add_argument("--front-end-slices", type=int, default=20, help="rolling slices in the front-end measurement's event file")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="bf_set_option knob for every context (experiments), e.g. --opt bin_split=2")
    ap.add_argument("--cpu-iters", type=int, default=0,
                    help="iteration_steps of the CPU baseline's sample; 0 (default): the whole cold run, to the loop's own termination "
                         "(~530 iterations, 10-40 s on one host core)")
    ap.add_argument("--cpu-cores", type=int, default=0, help="cap on the host cores of the slice-parallel CPU figure")
    ap.add_argument("--farm-slices", type=int, default=512,
                    help="--config 5: independent slices (seeds 0 .. n-1) farmed over the ranks: every slice context of every rank "
                         "claims its next slice from one shared queue")
    ap.add_argument("--farm-static", action="store_true", help="--config 5: the round robin slice i -> rank i %% N instead (A/B)")
    ap.add_argument("--farm-costs", default=None, metavar="JSON",
                    help="--config 5: a previous run's JSON line (its config.per_slice.iterations): slices are handed out longest first")
    ap.add_argument("--farm-record", default=None, metavar="PATH", help="--config 5: also write the JSON line to this file")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="allow more ranks than visible devices (rank r on device r %% devices): tests only")
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-worker-seed", type=int, default=1, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:   # one process of the slice-parallel CPU baseline (no GPU, no torch)
        import oracle
        from better_flow_amd import synth
        sl = synth.make_slice(args.events, args.height, args.width, 0.030, seed=args.cpu_worker_seed)
        oc = oracle.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
        ow = oc.set_cloud(args.scale, args.height, args.width)
        print("ready", flush=True)
        sys.stdin.readline()
        t0 = time.perf_counter()
        _, lp, _ = oc.run(ow, oracle.Model(), max_iter=args.cpu_iters - 1, res_x=args.height, res_y=args.width)
        print(len(sl["t"]), lp.itercount, time.perf_counter() - t0, flush=True)
        return

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: this process only spawns the N ranks and relays the outcome
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: launched with WORLD_SIZE=%d but --gpus %d: one rank per GPU, the two must agree"
                         % (world, args.gpus))
    dist = None
    if world > 1:
        # torch first: its bundled libamdhip64.so.7 is then the one HIP runtime of the process
        import torch  # noqa: F401
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # No data-path collective exists on this path; the process group only carries the
        # timing barrier and two scalar reductions, so the CPU (gloo) backend is enough.
        # (gloo announces its connections on STDOUT, from C++: keep stdout to the one JSON line -- the library's chatter goes
        # to stderr for the duration of the rendezvous)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    import numpy as np
    from better_flow_amd import accel, synth

    if args.config == 5:
        args.height, args.width = 720, 1280
    H, W, s = args.height, args.width, args.scale
    ndev = accel.device_count()
    if ndev <= 0:
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    if world > ndev and not args.oversubscribe:
        raise SystemExit("bench.py: %d ranks but only %d HIP device(s) visible (one rank per GPU; --oversubscribe "
                         "shares devices, for tests)" % (world, ndev))
    device = local_rank % ndev
    # Host-core budget (SURVEY 8(e)'s caveat: feeding must not serialise).  A rank's slice contexts poll their loops from host
    # threads: measured 0.7 - 0.85 busy cores per rank with four contexts in flight (config.host_cores_busy_per_rank), less
    # with fewer.  Far more ranks than the cores this job may use would time the host, not the GPUs: refuse, loudly, unless the
    # caller says the ranks share on purpose (--oversubscribe: tests); a job somewhat over its budget runs and says so.
    cores_avail = host_cores()
    cores_needed = world * min(1.0, 0.25 + 0.15 * max(1, args.concurrent))
    host_budget = {"ranks": world, "slice_contexts_per_rank": max(1, args.concurrent), "polling_cores_needed": round(cores_needed, 2),
                   "host_cores_available": cores_avail, "ok": cores_needed <= cores_avail}
    if rank == 0:
        print("bench.py: host-core budget: %d rank(s) x %d slice context(s) need ~%.1f polling cores, %d available%s"
              % (world, max(1, args.concurrent), cores_needed, cores_avail, "" if host_budget["ok"] else " -- OVER BUDGET"), file=sys.stderr)
    # (refused: fewer than 0.6 x the cores the ranks need -- every rank's polling threads would share a core with another rank's;
    # between that and the full budget the run goes ahead with the line above and "ok": false in the JSON)
    if cores_avail < 0.6 * cores_needed and not args.oversubscribe:
        raise SystemExit("bench.py: %d ranks need ~%.1f host cores for polling but this job may use %d (cpu affinity / cgroup quota): "
                         "the result would time the host; give the job more cores, lower --concurrent, or pass --oversubscribe"
                         % (world, cores_needed, cores_avail))
    # this rank next to its GPU: the CPUs (and preferred memory) of the device's host NUMA node -- every thread started from here
    # inherits the binding, every pinned buffer allocated from here on is first touched there (SURVEY 8(e)'s caveat)
    affinity_at_start = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    numa_node = accel.bind_thread_to_device_numa(device)
    if args.config == 5:
        # BASELINE config 5 as specified: a batch of independent cold slices (seeds 0 .. n-1) at 1280x720 farmed over the
        # ranks (slice i -> rank i % N, --concurrent slice contexts per rank, models gathered on every rank; no data-path
        # collective).  One timed pass over the whole batch; --steps / --warmup do not apply.
        from better_flow_amd import farm
        specs = [farm.SliceSpec(i, H, W, events=args.events, seed=i) for i in range(args.farm_slices)]
        if args.farm_costs:   # iteration counts of an earlier run of the same batch: longest first
            with open(args.farm_costs) as f:
                prev_its = json.loads(f.read().strip().splitlines()[-1])["config"]["per_slice"]["iterations"]
            for sp_, it_ in zip(specs, prev_its):
                sp_.cost = it_
        warm = [farm.SliceSpec(-1 - rank, H, W, events=args.events, seed=100000 + rank)]   # allocations, code objects
        farm.run_farm(warm, rank=0, world=1, device=device, concurrent=1, scale=s, max_iter=3)
        # the slices' arrays exist before the clock starts; the lanes move + solve.  Several ranks: any rank may claim any slice,
        # so each rank leaves the slices it generated where the others can map them (16 B per event under /dev/shm)
        share_dir = None
        if world > 1 and not args.farm_static:
            # rank 0 makes a fresh directory with room for the batch (16 B per event; /dev/shm, else the temporary directory)
            # and tells the others; removed in the `finally` below whatever happens
            box = [None]
            if rank == 0:
                try:
                    box[0] = farm.make_share_dir(16 * args.events * args.farm_slices, tag="farm_%s" % os.environ.get("MASTER_PORT", "0"))
                except RuntimeError as e:
                    box[0] = e
            dist.broadcast_object_list(box, src=0)
            if isinstance(box[0], Exception):
                raise SystemExit("bench.py: %s" % box[0])
            share_dir = box[0]
        try:
            farm.prepare(specs, rank=rank, world=world, share_dir=share_dir)
            if dist is not None:
                dist.barrier()
            t0 = time.perf_counter()
            merged = farm.run_farm(specs, rank=rank, world=world, device=device, concurrent=max(1, args.concurrent), scale=s,
                                   dist=dist, options={k_: int(v_) for k_, v_ in (o_.split("=") for o_ in args.opt)},
                                   static=args.farm_static)
            if dist is not None:
                dist.barrier()
            elapsed = time.perf_counter() - t0
            if dist is not None:
                import torch
                tt = torch.tensor([elapsed], dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                elapsed = float(tt[0])
        finally:
            if share_dir is not None and rank == 0:   # (the slices were loaded by their claimers before run_farm returned or raised)
                import shutil
                shutil.rmtree(share_dir, ignore_errors=True)
        if rank == 0:
            assert sorted(merged) == list(range(args.farm_slices))
            ev = sum(r["events"] for r in merged.values())
            its = [merged[i]["iterations"] for i in range(args.farm_slices)]
            bal = farm.balance(merged, world)
            line = json.dumps({
                "metric": METRIC, "value": ev / elapsed / 1e6, "value_definition": "host_to_host", "unit": "Mevents/s", "n_gpus": world, "steps": 1, "warmup": 0,
                "ms_per_step": 1e3 * elapsed, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f64", "data": "synthetic",
                "config": {"workload": "BASELINE config 5: batch of %d independent %d-event 30 ms slices at %dx%d, scale %d, "
                                       "cold start to the reference loop's own termination, %s"
                                       % (args.farm_slices, args.events, W, H, s,
                                          ("farmed slice i -> rank i %% %d" % world) if args.farm_static else
                                          ("every slice context of the %d rank(s) claims its next slice from one shared queue%s"
                                           % (world, ", longest first by a previous run's iteration counts" if args.farm_costs else ""))),
                           "slices": args.farm_slices, "slices_failed": sum(1 for r in merged.values() if r["rc"] != 0),
                           "iterations_per_slice_mean": sum(its) / len(its), "iterations_per_slice_max": max(its),
                           "ms_per_slice_mean": sum(r["ms"] for r in merged.values()) / len(merged),
                           "parallelism": "slice-parallel: %d GPU(s) x %d slice contexts, no collectives" % (world, args.concurrent),
                           # per rank: seconds from the start to its last slice, slices taken; imbalance = slowest rank / mean
                           "ranks": {"busy_s": bal["busy_s"], "slices": bal["slices"], "imbalance": bal["imbalance"],
                                     "numa_node_of_rank0": numa_node},
                           "host_budget": host_budget,
                           # ms: upload issue (under the lane's previous solve) -> model; solve_ms: the lane's own time for the slice
                           "per_slice": {"iterations": its, "ms": [round(merged[i]["ms"], 3) for i in range(args.farm_slices)],
                                         "solve_ms": [round(merged[i]["solve_ms"], 3) for i in range(args.farm_slices)],
                                         "rank": [merged[i]["rank"] for i in range(args.farm_slices)],
                                         "t_done_s": [round(merged[i].get("t1", 0.0), 4) for i in range(args.farm_slices)]}},
            })
            print(line)
            if args.farm_record:
                with open(args.farm_record, "w") as f:
                    f.write(line + "\n")
        if dist is not None:
            dist.destroy_process_group()
        return
    slices = [synth.make_slice(args.events, H, W, 0.030, seed=1 + rank * 1000 + i)
              for i in range(args.slices)]
    nmax = max(len(sl["t"]) for sl in slices)
    import threading
    B = max(1, args.concurrent)
    accs = [accel.Accel(device=device, max_events=nmax, max_rows=s * H + s, max_cols=s * W + s) for _ in range(B)]
    for a_ in accs:
        if B > 1:
            a_.set_option("co_schedule", 1)   # several slice contexts share this GPU (include/bf_accel.h)
        for kv in args.opt:
            k_, v_ = kv.split("=")
            a_.set_option(k_, int(v_))
    acc = accs[0]
    resident = []
    for sl in slices:
        resident.append((acc.to_device(sl["fr_x"]), acc.to_device(sl["fr_y"]),
                         acc.to_device(sl["t"].astype(np.int32)), len(sl["t"])))

    def make_opts(a):
        o = a.default_opts()
        o.res_x, o.res_y, o.poll_interval, o.want_uv = H, W, args.poll, 1
        return o
    all_opts = [make_opts(a) for a in accs]
    opts = all_opts[0]

    def step(i, warm_model=None, max_iter=-1, lane=0):
        a, o = accs[lane], all_opts[lane]
        dx, dy, dt, n = resident[i % len(resident)]
        a.upload_events_device(dx, dy, dt, n)
        a.set_cloud(s, H, W)
        if warm_model is not None:
            a.set_model(warm_model)
        o.max_iter = max_iter
        rc, m, info = a.run(o)
        return n, m, info

    def run_steps(first, count):
        """`count` steps; a step = B independent cold slices, one per lane, in flight together
        (ctypes releases the GIL inside the C-ABI calls).  Returns (events, iterations)."""
        tot = [[0, 0] for _ in range(B)]

        def lane_loop(lane):
            for k in range(count):
                n, _, info = step(first + k + lane, lane=lane)   # consecutive steps of a lane: different slices
                tot[lane][0] += n
                tot[lane][1] += info.iterations
        if B == 1:
            lane_loop(0)
        else:
            th = [threading.Thread(target=lane_loop, args=(l,)) for l in range(B)]
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
        for a in accs:
            a.synchronize()
        return sum(x[0] for x in tot), sum(x[1] for x in tot)

    def barrier():
        for a in accs:
            a.synchronize()
        if dist is not None:
            dist.barrier()

    # ---- timed region: exactly K cold-start steps (each step = B slices in flight) ----------
    run_steps(0, args.warmup)
    barrier()
    import resource
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    events, iters = run_steps(args.warmup, args.steps)
    elapsed = time.perf_counter() - t0
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    host_cores_busy = ((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)) / max(elapsed, 1e-9)
    # The CPU baseline (one host core, ~10 s to the loop's own termination) starts NOW, in a side process, and runs under the
    # GPU legs that follow (regimes, roofline, other geometries: several seconds of kernels) instead of after them: the
    # headline above was timed without it, and whoever samples the GPU's activity during this run sees it busy.
    cpu_side = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import subprocess
        cpu_side = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", "--events", str(args.events),
                                     "--height", str(H), "--width", str(W), "--scale", str(s), "--cpu-iters", str(args.cpu_iters),
                                     "--cpu-worker-seed", "1"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
        if affinity_at_start is not None:   # (any core the job was given: not pinned to this GPU's NUMA node)
            try:
                os.sched_setaffinity(cpu_side.pid, affinity_at_start)
            except OSError:
                pass
        assert cpu_side.stdout.readline().strip() == "ready"
        cpu_side.stdin.write("go\n"); cpu_side.stdin.flush()
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

    # ---- other regimes (untimed extras, rank 0): B independent chains in flight, like the steps ----
    regimes = {}
    if rank == 0:
        reps = max(4, min(args.steps, 16))

        def run_regime(warm, max_iter):
            tot = [[0, 0] for _ in range(B)]

            def lane_loop(lane):
                prev = None
                if warm:   # consecutive slices of one stream, each started from the previous model (STM)
                    _, prev, _ = step(lane, lane=lane)
                for k in range(reps):
                    n, m, info = step(1 + k + lane, warm_model=prev if warm else None, max_iter=max_iter, lane=lane)
                    if warm:
                        prev = m
                    tot[lane][0] += n
                    tot[lane][1] += info.iterations
            for a in accs:
                a.synchronize()
            th = [threading.Thread(target=lane_loop, args=(l,)) for l in range(B)]
            t1 = time.perf_counter()
            if warm:   # the cold first slice of every chain is not part of the warm regime
                pass
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            for a in accs:
                a.synchronize()
            dt = time.perf_counter() - t1
            return sum(x[0] for x in tot), sum(x[1] for x in tot), dt

        # warm: time only the warm slices -> run the cold heads first, outside the clock
        heads = []
        for lane in range(B):
            _, m0, _ = step(lane, lane=lane)
            heads.append(m0)
        for a in accs:
            a.synchronize()
        tot = [[0, 0] for _ in range(B)]

        def warm_loop(lane):
            prev = heads[lane]
            for k in range(reps):
                n, prev, info = step(1 + k + lane, warm_model=prev, lane=lane)   # slice k+1 of this chain
                tot[lane][0] += n
                tot[lane][1] += info.iterations
        th = [threading.Thread(target=warm_loop, args=(l,)) for l in range(B)]
        t1 = time.perf_counter()
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        for a in accs:
            a.synchronize()
        dtw = time.perf_counter() - t1
        wev, wit = sum(x[0] for x in tot), sum(x[1] for x in tot)
        regimes["warm_stm"] = {"mevents_per_s": wev / dtw / 1e6, "iterations_per_slice": wit / (reps * B),
                               "ms_per_slice_per_chain": 1e3 * dtw / reps, "chains_in_flight": B}
        # capped: the reference's real-time setting max_iter = 10 (ros bf_visualizer.cpp:103)
        cev, cit, dtc = run_regime(False, 10)
        regimes["capped_max_iter_10"] = {"mevents_per_s": cev / dtc / 1e6, "iterations_per_slice": cit / (reps * B),
                                         "ms_per_slice_per_chain": 1e3 * dtc / reps, "chains_in_flight": B}

        # one slice / one chain at a time: the latency view of the same three regimes
        accs[0].set_option("co_schedule", 0)   # one context alone: the latency build of the stencil kernel

        def single(nrep, warm, max_iter):
            prev = step(0)[1] if warm else None
            accs[0].synchronize()
            t1 = time.perf_counter()
            ev = it = 0
            for k in range(nrep):
                n, m, info = step(1 + k, warm_model=prev, max_iter=max_iter)
                if warm:
                    prev = m
                ev += n
                it += info.iterations
            accs[0].synchronize()
            dt1 = time.perf_counter() - t1
            return {"mevents_per_s": ev / dt1 / 1e6, "ms_per_slice": 1e3 * dt1 / nrep, "iterations_per_slice": it / nrep}
        regimes["one_context"] = {"cold": single(3, False, -1), "warm_stm": single(reps, True, -1),
                                  "capped_max_iter_10": single(reps, False, 10)}

        # SURVEY 8(d) as written: from the slice's arrays in (pinned) host memory to the model back on the host -- the
        # H2D copy included, on the copy stream, overlapping the previous slice's solve
        # (bf_upload_events_async / bf_commit_upload).  `value` above keeps the inputs resident in HBM, as the bench
        # contract asks; these are the same regimes with the PCIe leg in.
        # Two host layouts: the reference's device layout (int fr_x, fr_y, t: 12 B / event, accel_lib.h:83-85) and the same with
        # the two sensor addresses as 16-bit values (8 B / event, bf_upload_events16_async) -- a warm-started stream is bound by
        # the link (1M-event slices: 0.25 ms of copy per slice at ~48 GB/s against 0.1 ms of solve with four chains in flight).
        pinned12, pinned8 = [], []
        for sl in slices:
            n_ = len(sl["t"])
            trip = [acc.pinned_int32(n_) for _ in range(3)]
            trip[0][:], trip[1][:], trip[2][:] = sl["fr_x"], sl["fr_y"], sl["t"].astype(np.int32)
            pinned12.append((trip, n_))
            trip8 = [acc.pinned_array(n_, np.uint16), acc.pinned_array(n_, np.uint16), acc.pinned_int32(n_)]
            trip8[0][:], trip8[1][:], trip8[2][:] = sl["fr_x"], sl["fr_y"], sl["t"].astype(np.int32)
            pinned8.append((trip8, n_))
        pinned = pinned8

        def host_to_host(nlanes, warm, max_iter, nrep):
            tot = [[0, 0] for _ in range(nlanes)]
            heads = [None] * nlanes
            for lane in range(nlanes):
                accs[lane].set_option("co_schedule", 1 if nlanes > 1 else 0)
                if warm:
                    heads[lane] = step(lane, lane=lane)[1]
            # untimed warm-up of the streaming path itself (once per context): staging slots, copy stream and events are
            # allocated at the first asynchronous upload -- 8 ms that belong to no slice
            for lane in range(nlanes):
                a = accs[lane]
                # (once per staging form: a context alone stages on the copy stream -- its first kernels there create a hardware
                # queue --, co-scheduled ones on the compute stream; and per host layout: the widening kernel's code object)
                key_ = (nlanes > 1, pinned is pinned8)
                if key_ not in getattr(a, "_h2h_warm", set()):
                    trip, n_ = pinned[lane % len(pinned)]
                    for _ in range(2):   # (both staging slots)
                        a.upload_events_async(trip[0], trip[1], trip[2], n_)
                        a.commit_upload()
                    a.set_cloud(s, H, W)
                    a._h2h_warm = getattr(a, "_h2h_warm", set()) | {key_}
            for a in accs:
                a.synchronize()
                a.set_option("defer_uploads", 1)

            def lane_loop(lane):
                a, o = accs[lane], all_opts[lane]
                prev = heads[lane]

                def put(k):
                    trip, n_ = pinned[(1 + k + lane) % len(pinned)]
                    a.upload_events_async(trip[0], trip[1], trip[2], n_)
                # two uploads ahead of the slice being solved (the context's two staging slots: copies AND staging kernels of
                # slice k + 1 run on the copy stream under slice k's solve), their HIP calls issued by bf_run once its first batch
                # is queued ("defer_uploads": one host thread per chain)
                put(0)
                if nrep > 1:
                    put(1)
                for k in range(nrep):
                    a.commit_upload()
                    if k + 2 < nrep:
                        put(k + 2)
                    a.set_cloud(s, H, W)
                    if warm:
                        a.set_model(prev)
                    o.max_iter = max_iter
                    rc, m, info = a.run(o)
                    if warm:
                        prev = m
                    tot[lane][0] += a.n
                    tot[lane][1] += info.iterations
            th = [threading.Thread(target=lane_loop, args=(l,)) for l in range(nlanes)]
            t1 = time.perf_counter()
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            for a in accs[:nlanes]:
                a.synchronize()
            dt1 = time.perf_counter() - t1
            for a in accs:
                a.set_option("defer_uploads", 0)
            return {"mevents_per_s": sum(x[0] for x in tot) / dt1 / 1e6, "ms_per_slice_per_chain": 1e3 * dt1 / nrep,
                    "iterations_per_slice": sum(x[1] for x in tot) / (nrep * nlanes), "chains_in_flight": nlanes}
        for o_ in all_opts:
            o_.want_uv = 0   # the model comes back; per-event flow stays on the device unless asked for
        regimes["host_to_host"] = {
            "note": "pinned host arrays (u16 row, u16 column, i32 t: 8 B / event) -> H2D on the copy stream (overlapped with the "
                    "previous slice) -> solve -> model on the host; per-event flow not read back",
            "cold": host_to_host(B, False, -1, 10), "warm_stm": host_to_host(B, True, -1, reps),
            "capped_max_iter_10": host_to_host(B, False, 10, reps),
            # (one chain: 3 x the slices of the other regimes -- the chain's first slice has no solve to hide its 160 us copy under,
            # which at 16 slices is 10 us of every slice's 220)
            "one_context": {"cold": host_to_host(1, False, -1, 6), "warm_stm": host_to_host(1, True, -1, 3 * reps),
                            "capped_max_iter_10": host_to_host(1, False, 10, reps)},
        }
        pinned = pinned12
        regimes["host_to_host"]["int32_addresses_12B_per_event"] = {
            "note": "the same with the reference's device layout on the host (int fr_x, fr_y, t)",
            "cold": host_to_host(B, False, -1, 6), "warm_stm": host_to_host(B, True, -1, reps),
            "one_context": {"warm_stm": host_to_host(1, True, -1, reps)}}
        for o_ in all_opts:
            o_.want_uv = 1

    # ---- roofline of the dominant kernel (warp+scatter): HIP events carrying the kernel's own timestamps -----
    roofline = None
    if rank == 0:
        # one slice context ALONE on the GPU (everything above has finished), in the mode the headline regime runs the
        # kernel in: with several contexts per GPU the update sits in the stencil kernel's tail and the warp+scatter
        # kernel is the lean one; a single context runs the update at the head of the warp+scatter kernel instead.
        acc.set_option("co_schedule", 1 if B > 1 else 0)

        def k_profile():
            acc.profile_enable(1)
            acc.profile_reset()
            live_ev_iters_ = live_ = 0
            for i in range(min(args.steps, 4)):
                n_, _, info_ = step(i)
                live_ += info_.iterations             # launches that really warped + scattered the slice
                live_ev_iters_ += n_ * info_.iterations
            p_ = acc.profile_get()
            acc.profile_enable(0)
            return p_, live_, live_ev_iters_
        # The kernel's launch shape follows the regime (bf_run): contexts that share the GPU run the lean kernel with 512-thread
        # work-groups (a 1024-thread group waits for half a CU's wave slots under contention), a context alone the head-update
        # kernel with 1024-thread ones.  The roofline below is the headline regime's own variant, measured with the GPU to
        # itself; the other variant is `context_alone_form`.
        p, live, live_ev_iters = k_profile()
        # ... and the form the same context takes when it really is alone (no co_schedule): update at the head of the
        # scatter kernel, no serial tail in the stencil kernel
        alone_form = None
        if B > 1:
            acc.set_option("co_schedule", 0)
            pa_, la_, lea_ = k_profile()
            acc.set_option("co_schedule", 1)
            alone_form = (pa_, la_, lea_)
        dx_, dy_, dt_, n__ = resident[0]
        acc.upload_events_device(dx_, dy_, dt_, n__)
        win_ = acc.set_cloud(s, H, W)
        img_px = float(win_.scale_img_x) * float(win_.scale_img_y)   # P of SURVEY 8(d): the slice's image
        # total time of ALL loop launches of K1 (the few early-exit launches after convergence included) over the
        # launches that did the work: a slightly pessimistic per-launch duration
        k1_s = p.warp_scatter_ms * 1e-3 / max(1, live)
        ev_per_launch = live_ev_iters / max(1, live)
        achieved = K1_BYTES_PER_EVENT_ITER * ev_per_launch / k1_s / 1e9
        try:
            copy_gbps = acc.copy_bandwidth(1 << 30, 5)
        except Exception:
            copy_gbps = None
        # HBM traffic per launch from the PMC passes (profiles/k1_traffic.json is written by
        # scripts/collect_r4.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs;
        # FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction).  null if not collected
        # or not for this workload.
        # (keyed by geometry AND kernel variant: the line names the 512-thread lean kernel when several contexts share the
        # GPU, the 1024-thread head-update kernel for one context -- the counters must be that variant's)
        traffic = traffic_kernel = stencil_traffic = None
        tj = os.path.join(ROOT, "profiles", "k1_traffic.json")
        if os.path.exists(tj) and args.events == 1000000:
            tjd = json.load(open(tj))
            key = "%dx%dx%d" % (W, H, s)
            t_ = tjd.get(key + ("_lean512" if B > 1 else "_head1024"))
            if t_:
                traffic = (2.0 * t_["fetch_kb"] + t_["write_kb"]) * 1024.0
                traffic_kernel = t_.get("kernel")
            t3_ = tjd.get(key + ("_stencil_tail" if B > 1 else "_stencil_head"))
            if t3_:
                stencil_traffic = (2.0 * t3_["fetch_kb"] + t3_["write_kb"]) * 1024.0
        # The same kernel at BASELINE's other two geometries (configs 3 and 5), one context alone as it runs there by default
        # (a stream's chain / a farm lane with the GPU to itself: update at the scatter head), first 120 iterations of a cold
        # 1M-event slice each -- a second or so, not part of the timed region.
        other_geo = []
        if args.events == 1000000 and (H, W) == (260, 346) and not args.opt:
            for (H2, W2, cfg) in ((480, 640, 3), (720, 1280, 5)):
                try:
                    sl2 = synth.make_slice(1000000, H2, W2, 0.030, seed=1)
                    a2 = accel.Accel(max_events=len(sl2["t"]), max_rows=s * H2 + s, max_cols=s * W2 + s)
                    o2 = a2.default_opts()
                    o2.res_x, o2.res_y, o2.max_iter = H2, W2, 120
                    for rep in range(2):
                        a2.upload_events(sl2["fr_x"], sl2["fr_y"], sl2["t"])
                        w2 = a2.set_cloud(s, H2, W2)
                        if rep == 1:
                            a2.profile_enable(1); a2.profile_reset()
                        _, _, i2 = a2.run(o2)
                    p2 = a2.profile_get()
                    fmt2 = a2.get_stat("scatter_format")
                    a2.close()
                    n2, it2 = len(sl2["t"]), max(1, i2.iterations)
                    k1 = p2.warp_scatter_ms * 1e-3 / it2
                    px2 = float(w2.scale_img_x) * float(w2.scale_img_y)
                    other_geo.append({
                        "geometry": "%dx%d, scale %d (BASELINE config %d), %d events" % (W2, H2, s, cfg, n2),
                        "scatter_format": {0: "dense slabs", 2: "event lists", 3: "own pixels + margin plane"}.get(fmt2, str(fmt2)),
                        "warp_scatter_us": k1 * 1e6, "stencil_us": 1e3 * p2.stencil_ms / it2,
                        "frac": K1_BYTES_PER_EVENT_ITER * n2 / k1 / 1e9 / HBM_PEAK_GBPS,
                        "iteration_frac": (K1_BYTES_PER_EVENT_ITER * n2 + 24.0 * px2) / ((p2.warp_scatter_ms + p2.stencil_ms) * 1e-3 / it2) / 1e9 / HBM_PEAK_GBPS,
                        "iterations": int(i2.iterations)})
                    # The event-list stencil kernel moves ~0.2 x of its 24 B / pixel (the lists follow the events, not the
                    # area): a byte fraction says nothing about it.  Its bound is the VECTOR UNIT: waves x vector instructions
                    # per wave (rocprofv3 SQ_INSTS_VALU / SQ_WAVES of this build, profiles/r6_valu.json) over the chip's issue
                    # rate -- 256 CUs x 4 SIMDs, one wave64 instruction per 4 cycles at 2.4 GHz = 614 G wave-instructions/s.
                    vj_ = os.path.join(ROOT, "profiles", "r6_valu.json")
                    if fmt2 == 2 and os.path.exists(vj_):
                        vk_ = [v_ for k_, v_ in json.load(open(vj_)).get("%dx%d" % (W2, H2), {}).items() if k_.startswith("k_stencil_binned")]
                        if vk_:
                            vpw_ = max(vk_, key=lambda v_: v_["launches"])["valu_per_wave"]
                            waves_ = ((w2.scale_img_x + 15) // 16) * ((w2.scale_img_y + 63) // 64) * 4
                            issue_ = 256 * 4 * 2.4e9 / 4
                            bound_us_ = waves_ * vpw_ / issue_ * 1e6
                            other_geo[-1]["stencil_compute_bound"] = {
                                "bound": "valu", "valu_per_wave": vpw_, "waves_per_launch": waves_, "issue_rate_wave_insts_per_s": issue_,
                                "bound_us": bound_us_, "achieved_frac": bound_us_ / (1e3 * p2.stencil_ms / it2),
                                "source": "profiles/r6_valu.json (the co-scheduled build of the kernel; this run is the head-update form)"}
                except Exception as e:   # (a measurement beside the contract's: never the reason a bench line is missing)
                    other_geo.append({"geometry": "%dx%d" % (W2, H2), "error": str(e)[:200]})
        # The two loop kernels with the CHIP FULL of their own work-groups: one context solving eight config-2 slices side by
        # side (8 x the sensor area, 8 x 1M events at the same density: 2048 bins of 48 x 64 pixels, 6016 stencil tiles) --
        # what a kernel achieves when it is not bound by one slice's latency chain (a single config-2 slice has one
        # work-group of the scatter kernel per CU).  Per-launch times are the kernels' own; fractions on algorithmic bytes.
        chip_full = None
        if args.events == 1000000 and (H, W, s) == (260, 346, 3) and not args.opt:
            try:
                H8, W8 = 2 * H, 4 * W
                sl8 = synth.make_slice(8000000, H8, W8, 0.030, seed=1)
                a8 = accel.Accel(max_events=len(sl8["t"]), max_rows=s * H8 + s, max_cols=s * W8 + s)
                for k8, v8 in (("binned", 2), ("fused", 0), ("bin_compact", 0), ("bin_split", 0), ("co_schedule", 1)):
                    a8.set_option(k8, v8)
                o8 = a8.default_opts()
                o8.res_x, o8.res_y, o8.max_iter = H8, W8, 80
                for rep in range(2):
                    a8.upload_events(sl8["fr_x"], sl8["fr_y"], sl8["t"])
                    w8 = a8.set_cloud(s, H8, W8)
                    if rep == 1:
                        a8.profile_enable(1); a8.profile_reset()
                    _, _, i8 = a8.run(o8)
                p8 = a8.profile_get()
                a8.close()
                n8, it8 = len(sl8["t"]), max(1, i8.iterations)
                px8 = float(w8.scale_img_x) * float(w8.scale_img_y)
                k1_8, k3_8 = p8.warp_scatter_ms * 1e-3 / it8, p8.stencil_ms * 1e-3 / it8
                chip_full = {
                    "what": "one context, %d events on a %dx%d sensor (eight config-2 slices side by side), dense slabs, 512-thread scatter "
                            "work-groups, update in the stencil tail; first %d iterations of a cold run" % (n8, W8, H8, it8),
                    "warp_scatter_us": k1_8 * 1e6, "stencil_us": k3_8 * 1e6,
                    "warp_scatter_us_per_1M_events": k1_8 * 1e6 * 1e6 / n8, "stencil_us_per_config2_image": k3_8 * 1e6 * img_px / px8,
                    "warp_scatter_frac": K1_BYTES_PER_EVENT_ITER * n8 / k1_8 / 1e9 / HBM_PEAK_GBPS,
                    "stencil_frac": 24.0 * px8 / k3_8 / 1e9 / HBM_PEAK_GBPS,
                    "iteration_frac": (K1_BYTES_PER_EVENT_ITER * n8 + 24.0 * px8) / (k1_8 + k3_8) / 1e9 / HBM_PEAK_GBPS,
                }
                del sl8
            except Exception as e:   # noqa: BLE001 -- a measurement beside the contract's
                chip_full = {"error": str(e)[:200]}
        # Both loop kernels by the same rule (28 B per event-iteration, 24 B per image pixel and iteration over
        # the kernel's own average launch time).  The TOP-LEVEL achieved / frac / traffic are those of the kernel that
        # takes more time per launch in this run -- the dominant one, as the contract asks; both are always listed.
        k3_s = p.stencil_ms * 1e-3 / max(1, live)
        k1_obj = {
            "kernel": "k_bin_warp_scatter%s (warp + tile-binned LDS scatter)" % ("_lean" if B > 1 else ""),
            "algorithmic_bytes_per_launch": K1_BYTES_PER_EVENT_ITER * ev_per_launch,
            "avg_launch_us": k1_s * 1e6, "achieved": achieved, "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic, "traffic_kernel": traffic_kernel,
        }
        k3_obj = {
            "kernel": "k_stencil_binned (slab merge + box sum + time image + Scharr + moments + fused update)",
            "algorithmic_bytes_per_launch": 24.0 * img_px,
            "avg_launch_us": k3_s * 1e6 if p.stencil_ms > 0 else None,
            # (no stencil launches at all when --opt fused=2 forces the one-kernel iteration: null then)
            "achieved": (24.0 * img_px / k3_s / 1e9) if p.stencil_ms > 0 else None,
            "frac": (24.0 * img_px / k3_s / 1e9 / HBM_PEAK_GBPS) if p.stencil_ms > 0 else None,
            "traffic": stencil_traffic,
            "note": "bound by dependent latency and instruction issue (its waves wait ~55 % of their cycles), not by bandwidth: see DESIGN.md section 4 and profiles/*pmc_sq_issue.txt",
        }
        # The same two fractions from the COMMITTED rocprofv3 summary of the same solo run (profiles/r6_solo_tail_kernel_stats.csv,
        # `rocprofv3 --kernel-trace --stats -- python scripts/run_once.py 3 co_schedule=1`: scripts/profile_r6.sh): the average
        # duration rocprofv3 reports for the kernel, early-exit launches included, so that the line can be re-derived from
        # profiles/ alone.  null when the file is absent, or when this is not the workload it was taken on.
        def rocprof_of(prefix, alg_bytes):
            path = os.path.join(ROOT, "profiles", "r6_solo_tail_kernel_stats.csv" if B > 1 else "r6_solo_kernel_stats.csv")
            if not (os.path.exists(path) and args.events == 1000000 and (H, W, s) == (260, 346, 3) and not args.opt):
                return None
            import csv
            best = None
            for r_ in csv.DictReader(open(path)):
                nm = r_["Name"].replace("void ", "")
                if nm.startswith("bf::" + prefix) and (best is None or float(r_["TotalDurationNs"]) > float(best["TotalDurationNs"])):
                    best = r_
            if best is None:
                return None
            us = float(best["AverageNs"]) * 1e-3
            return {"source": "profiles/" + os.path.basename(path), "kernel": best["Name"].split("(")[0].replace("void ", ""),
                    "calls": int(best["Calls"]), "avg_launch_us": us, "achieved": alg_bytes / us / 1e3,
                    "frac": alg_bytes / us / 1e3 / HBM_PEAK_GBPS}
        k1_obj["rocprof"] = rocprof_of("k_bin_warp_scatter", K1_BYTES_PER_EVENT_ITER * ev_per_launch)
        k3_obj["rocprof"] = rocprof_of("k_stencil_binned", 24.0 * img_px)
        for o_ in (k1_obj, k3_obj):
            o_["frac_rocprof"] = o_["rocprof"]["frac"] if o_["rocprof"] else None
        dom_is_k3 = p.stencil_ms > p.warp_scatter_ms
        dom = k3_obj if dom_is_k3 else k1_obj
        roofline = {
            "bound": "hbm", "kernel": dom["kernel"], "dominant": "stencil_kernel" if dom_is_k3 else "warp_scatter_kernel",
            "regime": "one slice context alone on the GPU running the kernel variants of the headline regime (%s); the dominant "
                      "kernel is the one with the longer average launch (per_kernel_us)" %
                      ("lean scatter kernel with 512-thread work-groups, update in the stencil kernel's tail, as with %d contexts per GPU" % B
                       if B > 1 else "update at the scatter kernel's head, 1024-thread work-groups"),
            "other_geometries": other_geo,
            "chip_full": chip_full,
            "context_alone_form": None if not alone_form else {
                "what": "the same slice with co_schedule off -- what one context alone on the GPU runs: model / loop update at the "
                        "head of the scatter kernel (k_bin_warp_scatter), no serial tail in the stencil kernel",
                "warp_scatter_us": 1e3 * alone_form[0].warp_scatter_ms / max(1, alone_form[1]),
                "stencil_us": 1e3 * alone_form[0].stencil_ms / max(1, alone_form[1]),
                "warp_scatter_frac": K1_BYTES_PER_EVENT_ITER * (alone_form[2] / max(1, alone_form[1])) /
                                     (alone_form[0].warp_scatter_ms * 1e-3 / max(1, alone_form[1])) / 1e9 / HBM_PEAK_GBPS,
                "stencil_frac": (24.0 * img_px / (alone_form[0].stencil_ms * 1e-3 / max(1, alone_form[1])) / 1e9 / HBM_PEAK_GBPS)
                                if alone_form[0].stencil_ms > 0 else None,
                "iteration_frac": (K1_BYTES_PER_EVENT_ITER * (alone_form[2] / max(1, alone_form[1])) + 24.0 * img_px) /
                                  ((alone_form[0].warp_scatter_ms + alone_form[0].stencil_ms) * 1e-3 / max(1, alone_form[1])) / 1e9 / HBM_PEAK_GBPS,
            },
            "achieved": dom["achieved"],
            "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": dom["frac"], "frac_rocprof": dom["frac_rocprof"], "traffic": dom["traffic"],
            "avg_launch_us": dom["avg_launch_us"], "launches": int(live), "launches_incl_early_exit": int(p.warp_scatter_launches),
            "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
            "measured_copy_ceiling_gbps": copy_gbps,
            # the whole iteration by SURVEY 8(d): 28 B per event + 24 B per image pixel, over the two loop kernels' time
            "iteration_algorithmic_bytes": K1_BYTES_PER_EVENT_ITER * ev_per_launch + 24.0 * img_px,
            "iteration_frac": (K1_BYTES_PER_EVENT_ITER * ev_per_launch + 24.0 * img_px) /
                              ((p.warp_scatter_ms + p.stencil_ms) * 1e-3 / max(1, live)) / 1e9 / HBM_PEAK_GBPS,
            # the other loop kernel, by the same rule: SURVEY 8(d) prices the image side of an iteration at 24 B / pixel
            "stencil_kernel": k3_obj,
            "warp_scatter_kernel": k1_obj,
            "per_kernel_us": {
                "warp_scatter": 1e3 * p.warp_scatter_ms / max(1, live),
                "stencil_moments_update": 1e3 * p.stencil_ms / max(1, live),
            },
            "note": "durations are the kernels' own begin/end timestamps (hipExtLaunchKernelGGL start/stop events on the "
                    "ctx stream), summed over every loop launch and divided by the launches that did work; "
                    "rocprofv3's view of the same solo runs: profiles/r6_solo_tail_kernel_stats.csv (these variants: lean 512-thread "
                    "scatter kernel, update in the stencil tail; `frac_rocprof` is computed from it) and r6_solo_kernel_stats.csv "
                    "(update at the head, 1024 threads)",
        }

    # ---- CPU baseline: the oracle (port of the reference path), rank 0 at N = 1 only -------
    cpu_baseline = None
    if cpu_side is not None:
        import collections
        sl = slices[0]
        n_cpu, it_cpu, dtc = (float(x) for x in cpu_side.stdout.readline().split())   # (the side process started above)
        cpu_side.wait()
        assert int(n_cpu) == len(sl["t"])
        oloop = collections.namedtuple("Loop", "itercount")(int(it_cpu))
        per_iter = dtc / max(1, oloop.itercount)
        full_iters = iters / max(1, args.steps * B)      # the GPU run's iterations per slice
        if args.cpu_iters > 0:
            cpu_value = len(sl["t"]) / (per_iter * full_iters) / 1e6
            sample = ("first %d iteration_steps of the same %d-event cold run (%.1f s, %.1f ms/iteration), extrapolated to the %.0f "
                      "iterations the full run takes" % (oloop.itercount, len(sl["t"]), dtc, 1e3 * per_iter, full_iters))
        else:   # the whole job, nothing extrapolated
            cpu_value = len(sl["t"]) / dtc / 1e6
            sample = ("the same %d-event slice from a cold start to the reference loop's own termination: %d iteration_steps in %.1f s "
                      "(%.1f ms/iteration; the GPU run: %.0f iterations)" % (len(sl["t"]), oloop.itercount, dtc, 1e3 * per_iter, full_iters))
        cpu_baseline = {
            "value": cpu_value, "unit": "Mevents/s",
            "cores": 1, "kind": "port",
            "sample": sample,
            "iterations": int(oloop.itercount), "to_termination": args.cpu_iters <= 0,
            "ms_per_iteration": 1e3 * per_iter,
        }
        # (the same run in the build container, for reference: scripts/cpu_to_termination.py)
        tt = os.path.join(ROOT, "profiles", "cpu_to_termination.json")
        if os.path.exists(tt) and args.events == 1000000 and (H, W, s) == (260, 346, 3):
            cpu_baseline["build_container"] = json.load(open(tt))
        # the fair multi-core figure -- one slice per host core, all cores busy at once (the
        # reference's O(N) loops are serial, so slice-parallel is the only way it uses a multi-core host)
        if affinity_at_start is not None:   # the CPU baseline may use every core the job was given, not only this GPU's NUMA node
            os.sched_setaffinity(0, affinity_at_start)
        ncore = host_cores()   # (a container's CPU quota, not the host's core count, is what this job may use)
        ncore = max(1, min(ncore, args.cpu_cores if args.cpu_cores > 0 else ncore))
        if ncore > 1:
            # one PROCESS per core (threads of one process serialise on page faults of the per-iteration images);
            # every worker builds its slice, reports ready, and all start together
            import subprocess
            short = max(4, (args.cpu_iters if args.cpu_iters > 0 else 240) // 4)
            cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", "--events", str(args.events),
                   "--height", str(H), "--width", str(W), "--scale", str(s), "--cpu-iters", str(short)]
            procs = [subprocess.Popen(cmd + ["--cpu-worker-seed", str(1 + k)], stdin=subprocess.PIPE,
                                      stdout=subprocess.PIPE, text=True) for k in range(ncore)]
            for p_ in procs:
                assert p_.stdout.readline().strip() == "ready"
            tc = time.perf_counter()
            for p_ in procs:
                p_.stdin.write("go\n"); p_.stdin.flush()
            res = [tuple(float(x) for x in p_.stdout.readline().split()) for p_ in procs]
            dta = time.perf_counter() - tc
            for p_ in procs:
                p_.wait()
            # every worker ran under the load of all the others: the job rate is the sum of the workers' own rates
            ev_iter_rate = sum(r[0] * r[1] / r[2] for r in res)
            cpu_baseline["all_cores"] = {
                "value": ev_iter_rate / full_iters / 1e6, "unit": "Mevents/s", "cores": ncore,
                "sample": "%d slices at once, one process per core, first %d iteration_steps each (%.1f s), "
                          "extrapolated as above" % (ncore, short, dta),
            }

    # ---- the drop-in front end: events/s from a binary event FILE to the last model, through bf_motion_compensator ----
    front_end = None
    if rank == 0 and world == 1 and not args.no_front_end and args.config == 2:
        for a in accs:
            a.synchronize()
        try:
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            import front_end_bench
            fe = front_end_bench.run(slices=args.front_end_slices, events=args.events, height=H, width=W, reps=3)
            fo = front_end_bench.run(slices=max(2, args.front_end_slices // 4), events=args.events, height=H, width=W,
                                     with_output=True, reps=1)
            front_end = {
                "what": "bf_motion_compensator (stream engine: pinned SoA ring, slice farm worker, STM chain) on a binary "
                        "event file in the page cache: %d rolling 30 ms slices of ~%d events; wall clock of the tool's own "
                        "phases (--timing)" % (args.front_end_slices, args.events),
                "file_to_last_model": {"events": fe["events"], "slices": fe["slices"], "seconds": fe["stream_s"],
                                       "mevents_per_s": fe["mevents_per_s"],
                                       "note": "includes the cold first slice (%d iterations in all)" % fe["iterations"]},
                "steady_state_warm": {"seconds": fe["steady_s"], "mevents_per_s": fe["steady_mevents_per_s"],
                                      "note": "from the delivery of the first slice's model to the last: warm-started slices"},
                "init_s": fe["init_s"], "process_wall_s": fe["process_wall_s"],
                "with_flow_output": {"events": fo["events"], "stream_s": fo["stream_s"], "output_s": fo["output_s"],
                                     "stream_mevents_per_s": fo["mevents_per_s"],
                                     "output_mlines_per_s": fo["events"] / fo["output_s"] / 1e6 if fo["output_s"] > 0 else None,
                                     "note": "-o: per-event flow read back per slice, de-duplicated table, text written on %s"
                                             % ("several threads")},
            }
            import default_ring_bench
            rr = default_ring_bench.run(events=5000000, reps=2)
            front_end["reference_ring"] = {
                "what": "the same tool with the reference's compiled-in ring (50 000 events, a slice every 20 000 events / 33 ms, "
                        "warm-start chain) on a 240x180 stream of 250 000 events per 33 ms: small slices, sequential chain -- the "
                        "loop's latency (one-kernel iteration), not its throughput",
                "events": rr["events"], "slices": rr["slices"], "iterations": rr["iterations"],
                "mevents_per_s": rr["steady_mevents_per_s"], "us_per_iteration": 1e6 * rr["steady_s"] / max(1, rr["iterations"]),
            }
        except Exception as e:   # noqa: BLE001 -- the front end is an extra; the metric does not depend on it
            front_end = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0:
        h2h = regimes.get("host_to_host", {})
        targets = {
            "north_star": ">= 1 Gevents/s end-to-end motion compensation on 1M-event 346x260 slices at 1 x MI355X",
            "met_by": {
                "regime": "warm_stm (the reference's operating mode: every slice of a stream warm-started from the previous "
                          "model, dvs_flow.h:218-224), host arrays -> model on the host, H2D copy included",
                "mevents_per_s": h2h.get("warm_stm", {}).get("mevents_per_s"),
                "one_chain_mevents_per_s": h2h.get("one_context", {}).get("warm_stm", {}).get("mevents_per_s"),
            },
            "not_met_by": {
                "regime": "cold (STM off, the reference loop to its own termination: ~530 iterations per slice at >= 5.9 us of "
                          "HBM traffic each cannot reach 1 Gevents/s on any hardware)",
                "host_to_host_mevents_per_s": h2h.get("cold", {}).get("mevents_per_s"),
            },
        }
        out = {
            "metric": METRIC,
            "value": events_all / elapsed / 1e6,
            # which definition `value` uses, machine readable: inputs resident in HBM when the timed region starts (the bench
            # contract of this build's task statement); SURVEY 8(d)'s host-to-host form is value_host_to_host
            "value_definition": "hbm_resident",
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
                            "cold start (STM off) to the reference loop's own termination; a step = %d "
                            "independent slices per GPU in flight together, slices resident in HBM"
                            % (args.events, W, H, s, B),
                "slices_per_step_per_gpu": B,
                "events_per_slice": events_all / (args.steps * world * B),
                "iterations_per_slice": iters_all / (args.steps * world * B),
                "event_iterations_per_s": events_all / (args.steps * world * B) * iters_all / elapsed,
                "parallelism": "slice-parallel: %d GPU(s) x %d concurrent slice contexts (HIP streams) per GPU, "
                               "no collectives" % (world, B),
                "host_cores_busy_per_rank": host_cores_busy,
                "host_cores_available": host_cores(),
                "host_budget": host_budget,
            },
            # SURVEY 8(d)'s own definition of the metric (host arrays -> model on the host, H2D included), cold regime:
            # the number next to `value`, which keeps the inputs resident in HBM as the bench contract prescribes
            "value_host_to_host": h2h.get("cold", {}).get("mevents_per_s"),
            "targets": targets,
            "regimes": regimes,
            "front_end": front_end,
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
        }
        if roofline is not None and "iteration_algorithmic_bytes" in roofline and roofline.get("launches"):
            # the timed region itself (all slice contexts of one GPU together) by the same SURVEY 8(d) price per iteration:
            # what fraction of the HBM peak the headline number corresponds to
            its_per_s = iters_all / elapsed / world
            roofline["headline_regime"] = {
                "what": "algorithmic bytes per second of the timed region (%d slice contexts per GPU in flight): iterations per "
                        "second x iteration_algorithmic_bytes, over the HBM peak" % B,
                "iterations_per_s_per_gpu": its_per_s,
                "achieved": its_per_s * roofline["iteration_algorithmic_bytes"] / 1e9,
                "frac": its_per_s * roofline["iteration_algorithmic_bytes"] / 1e9 / HBM_PEAK_GBPS,
                "frac_of_measured_copy_ceiling": (its_per_s * roofline["iteration_algorithmic_bytes"] / 1e9 / roofline["measured_copy_ceiling_gbps"])
                                                 if roofline.get("measured_copy_ceiling_gbps") else None,
            }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    for a in accs:
        a.close()


if __name__ == "__main__":
    main()
