"""Slice farm: independent time slices farmed over ranks / GPUs.

A slice is a complete optimisation problem -- the reference already models work as a QUEUE of (events, model) tasks
(dvs_flow.h:200-231) --, so ranks never exchange data: every lane (slice context) of every rank claims its next slice from
one shared counter (`SliceQueue`: a lock in one process, `TCPStore.add` of the gloo process group across ranks -- an
8-byte control message per slice, no data-path collective), every rank runs its own bf_ctx per lane, and the host gathers
11 doubles per slice.  Slices differ widely (config 5 on one GPU: 6 437 iterations per slice on average, up to 33 410), so
the static round robin `i % world` this replaces left ranks idle behind one long slice; with known costs the queue hands
slices out longest first.  No RCCL collective is on the data path."""
import threading


def shard(n_slices, rank, world):
    """Indices of the slices rank `rank` of `world` processes owns (round robin)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, n_slices, world))


class SliceQueue:
    """The farm's task queue: `claim()` returns the next unclaimed slice index, or None when all are taken.

    order: index order, or -- when `costs` (any monotone estimate of a slice's work: event count x expected iterations, a
    previous run's milliseconds) are given -- longest first, ties by index; every rank computes the same order.  The counter is
    a lock-protected integer in one process and one key of the process group's TCPStore across ranks (`dist`: an initialised
    torch.distributed; `store.add` is atomic and returns the new value).

    Which key: the ranks agree on it THROUGH the store -- every store-backed construction of a queue called `name` takes a
    ticket from "bf_farm/<name>/ctor", and tickets world*g .. world*g + world - 1 are generation g -- so a rank that built
    other queues on the side (process-local ones, or ones under another name) still lands on the same counter as its
    peers.  What the ranks must share is only the number of store-backed queues of that NAME they have built (they build
    them collectively: run_farm does); `world` ranks that disagree about n_slices or the order are caught by `gather`
    (a slice processed twice raises)."""

    def __init__(self, n_slices, costs=None, dist=None, name=None):
        if costs is not None:
            if len(costs) != n_slices:
                raise ValueError("one cost per slice")
            self.order = sorted(range(n_slices), key=lambda i: (-float(costs[i]), i))
        else:
            self.order = list(range(n_slices))
        self._lock = threading.Lock()
        self._next = 0
        self._store = None
        self.world = 1
        if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
            from torch.distributed import distributed_c10d
            self._store = distributed_c10d._get_default_store()
            self.world = dist.get_world_size()
            name = name or "queue"
            gen = (int(self._store.add("bf_farm/%s/ctor" % name, 1)) - 1) // self.world
            self._key = "bf_farm/%s/%d" % (name, gen)

    def claim(self):
        if self._store is not None:
            k = int(self._store.add(self._key, 1)) - 1
        else:
            with self._lock:
                k = self._next
                self._next += 1
        return self.order[k] if k < len(self.order) else None

    def remaining(self):
        """Slices nobody has claimed yet (a snapshot: other lanes and ranks keep claiming)."""
        if self._store is not None:
            k = int(self._store.add(self._key, 0))
        else:
            with self._lock:
                k = self._next
        return max(0, len(self.order) - k)


def run_queue(queue, process_slice, lanes=1, dist=None, rank=0):
    """`lanes` threads of this rank claim slices from `queue` and run process_slice(i) -> dict; returns {i: result} with the
    lane and the claim / completion times (seconds since this call) added as "lane", "t0", "t1".

    With `dist` (an initialised torch.distributed) the records of ALL ranks are gathered and returned on every rank, and an
    exception in any lane of any rank is raised on EVERY rank after that gather -- a failing rank must not leave its peers
    waiting in a collective it never joins (the same contract as run_farm).  Without `dist` the first error is raised here."""
    import time
    results, errors = {}, []
    lock = threading.Lock()
    start = time.perf_counter()

    def lane(k):
        try:
            while True:
                i = queue.claim()
                if i is None:
                    return
                t0 = time.perf_counter() - start
                r = dict(process_slice(i))
                r.update(lane=k, t0=t0, t1=time.perf_counter() - start)
                with lock:
                    results[i] = r
        except Exception as e:   # noqa: BLE001
            with lock:
                errors.append(e)
    th = [threading.Thread(target=lane, args=(k,)) for k in range(max(1, lanes))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        if errors:
            raise errors[0]
        return results
    for r in results.values():
        r.setdefault("rank", rank)
    for n, e in enumerate(errors):   # (travels with the records)
        results[("error", rank, n)] = {"error": "%s: %s" % (type(e).__name__, e), "rank": rank, "slice": None}
    merged = gather(results, dist)
    _raise_lane_errors(merged)
    return merged


def _raise_lane_errors(merged):
    errors = [v for k, v in merged.items() if isinstance(k, tuple)]
    if errors:
        raise RuntimeError("slice farm: %d lane(s) failed; first: rank %s, slice %s: %s" %
                           (len(errors), errors[0]["rank"], errors[0]["slice"], errors[0]["error"]))


def balance(merged, world):
    """Per-rank seconds from the farm's start to the rank's last completed slice ("busy_s"), slices and summed slice
    milliseconds per rank, and imbalance = slowest rank / mean rank (1.0 = perfectly even)."""
    busy = [0.0] * world
    count = [0] * world
    ms = [0.0] * world
    for r in merged.values():
        k = r.get("rank", 0)
        busy[k] = max(busy[k], r.get("t1", 0.0))
        count[k] += 1
        ms[k] += r.get("ms", 0.0)
    mean = sum(busy) / max(1, world)
    return {"busy_s": busy, "slices": count, "slice_ms_sum": ms, "imbalance": (max(busy) / mean) if mean > 0 else 1.0}


def simulate_makespan(durations, world, lanes=1, costs=None, static=False):
    """Makespan (same unit as `durations`) of farming slices with the given durations over `world` ranks x `lanes` lanes:
    static=True: slice i belongs to rank i % world, whose lanes take its slices in index order (the round robin this module
    used to have); else every lane of every rank claims from one queue -- index order, or longest-`costs`-first.  A lane takes
    its next slice the moment it is free (list scheduling)."""
    import heapq
    n = len(durations)

    def schedule(ids, nlanes):
        free = [0.0] * nlanes
        heapq.heapify(free)
        end = 0.0
        for i in ids:
            t = heapq.heappop(free) + durations[i]
            end = max(end, t)
            heapq.heappush(free, t)
        return end
    if static:
        return max([schedule(range(r, n, world), lanes) for r in range(world)] or [0.0])
    order = sorted(range(n), key=lambda i: (-float(costs[i]), i)) if costs is not None else range(n)
    return schedule(order, world * lanes)


def run_shard(slice_ids, process_slice):
    """Run process_slice(i) -> dict for every owned slice; returns {i: result}."""
    return {i: process_slice(i) for i in slice_ids}


def gather(results, dist=None):
    """Merge per-rank {slice: result} dicts on every rank (all_gather_object when
    torch.distributed is initialised, identity otherwise)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return dict(results)
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, results)
    merged = {}
    for p in parts:
        for k, v in p.items():
            if k in merged:
                raise RuntimeError("slice %r processed twice" % (k,))
            merged[k] = v
    return merged


class SliceSpec:
    """One independent slice of the farm: a synthetic slice description (seed, geometry) or ready event arrays."""

    def __init__(self, index, height, width, events=1000000, duration_s=0.030, seed=None, arrays=None):
        self.index, self.height, self.width = index, height, width
        self.events, self.duration_s = events, duration_s
        self.seed = index if seed is None else seed
        self.arrays = arrays
        self.path = None    # prepare(share_dir=...): where the rank that generated the slice left it for the others
        self.cost = None    # optional estimate of the slice's work (SliceQueue hands out longest first)

    def load(self):
        if self.arrays is None:   # generated once, outside the farm's lanes (and outside any timed region: prepare())
            import os
            if self.path is not None and os.path.exists(self.path):
                import numpy as np
                with np.load(self.path) as z:
                    self.arrays = {k: z[k] for k in ("fr_x", "fr_y", "t")}
            else:
                from . import synth
                self.arrays = synth.make_slice(self.events, self.height, self.width, self.duration_s, seed=self.seed)
        return self.arrays

    def drop(self):
        """Forget the arrays of a slice that can be loaded again (shared file): a lane keeps two slices in memory, not 512."""
        if self.path is not None:
            self.arrays = None

    def count(self):
        return len(self.arrays["t"]) if self.arrays is not None else self.events


def prepare(specs, rank=0, world=1, share_dir=None):
    """Generate the event arrays up front -- the farm's lanes only move and solve them.  One rank: all of them, in memory.
    Several ranks: rank r generates the slices i % world == r and, since any rank may claim any slice from the shared queue,
    leaves each as an uncompressed .npz under `share_dir` (a directory all ranks see, e.g. under /dev/shm: 16 bytes per
    event); the caller puts a barrier between prepare() and run_farm().  Without a share_dir a rank generates a foreign
    slice when it claims it (correct, slow: inside the timed region)."""
    import os
    import numpy as np
    for s in specs:
        if share_dir is not None and world > 1:
            s.path = os.path.join(share_dir, "bf_farm_slice_%d.npz" % s.index)
        if s.index % world != rank:
            continue
        sl = s.load()
        if s.path is not None:
            tmp = s.path + ".tmp.%d.npz" % os.getpid()
            np.savez(tmp, fr_x=sl["fr_x"], fr_y=sl["fr_y"], t=sl["t"])
            os.replace(tmp, s.path)
            s.arrays = None
    return specs


def run_farm(specs, rank=0, world=1, device=0, concurrent=4, scale=3, max_iter=-1, want_flow_digest=False, dist=None,
             options=None, static=False, bind_numa=True):
    """The slice farm on the HIP path (dvs_flow.h:200-231's task queue, over ranks and slice contexts).

    Every rank runs `concurrent` slice contexts (lanes: one host thread + bf_ctx + HIP stream + copy stream each); every
    lane of every rank claims its next slice from ONE shared queue (`SliceQueue`: longest first when the specs carry a
    `cost`), so neither a slow slice nor a slow rank holds the others up; `static=True` is the round robin slice i -> rank
    i % world (each rank's lanes then share the rank's own queue), kept for A/B runs.  Every slice is a cold start (STM off:
    independent slices).  A lane stages its next slice in pinned memory and uploads it on its copy stream
    (bf_upload_events_async) while the current one is being solved.  `bind_numa`: every lane first binds itself to the CPUs
    of its GPU's NUMA node (bf_bind_thread_to_device_numa), so that its pinned staging buffers and its polling live next to
    the device (SURVEY 8(e)'s caveat).  The per-slice records -- return code, iterations, the 88-byte model, events, milliseconds, optionally a digest
    of the per-event flow, or the text of an error -- are merged on every rank with all_gather_object when `dist` is an
    initialised torch.distributed (gloo: no data-path collective exists on this path).  An error in any lane of any rank
    is raised on EVERY rank, after the gather (no rank is left waiting in a collective).  Returns {slice index: record}.
    The native form of the same farm, for C++ callers and the command line, is better_flow/slice_farm.h."""
    import hashlib
    import time
    import numpy as np
    from . import accel
    by_index = {s.index: s for s in specs}
    ids = sorted(by_index)
    if not static and world > 1:
        # The shared queue lives in the process group's store.  A caller that says world > 1 without an initialised group of
        # that size would get a process-local counter on every rank -- every rank solving EVERY slice, unmerged results, a
        # throughput figure `world` times too high -- so it gets the round robin instead (what this function did before it
        # had a queue), and is told.
        ok = dist is not None and dist.is_initialized() and dist.get_world_size() == world
        if not ok:
            import warnings
            warnings.warn("run_farm: world=%d but no initialised torch.distributed group of that size: static sharding "
                          "(slice i -> rank i %% world), results NOT gathered across ranks" % world, RuntimeWarning, stacklevel=2)
            static = True
    if static:
        mine = [i for i in ids if i % world == rank]
        work = SliceQueue(len(mine), costs=None, dist=None)
        claim = lambda: (lambda k: None if k is None else by_index[mine[k]])(work.claim())   # noqa: E731
        n_mine = len(mine)
    else:
        costs = [by_index[i].cost for i in ids]
        work = SliceQueue(len(ids), costs=costs if all(c is not None for c in costs) else None, dist=dist, name="run_farm")
        claim = lambda: (lambda k: None if k is None else by_index[ids[k]])(work.claim())   # noqa: E731
        n_mine = len(ids)
    results = {}
    if n_mine:
        hmax = max(s.height for s in specs)
        wmax = max(s.width for s in specs)
        nmax = max(s.count() for s in specs)
        lock = threading.Lock()
        t_start = time.perf_counter()

        def lane(lane_no):
            a = None
            cur = None
            try:
                if bind_numa:
                    accel.bind_thread_to_device_numa(device)   # before the first pinned allocation of this lane
                a = accel.Accel(device=device, max_events=nmax, max_rows=scale * hmax + scale, max_cols=scale * wmax + scale)
                if concurrent > 1:
                    a.set_option("co_schedule", 1)
                a.set_option("stream_prealloc", 1)
                for k, v in (options or {}).items():
                    a.set_option(k, v)
                o = a.default_opts()
                pin = [[a.pinned_int32(nmax) for _ in range(3)] for _ in range(2)]   # two staging slots of pinned memory

                def put(slot):
                    """Take the next slice off the queue and start its upload; None when the queue is empty."""
                    s = claim()
                    if s is None:
                        return None
                    sl = s.load()
                    n = len(sl["t"])
                    pin[slot][0][:n], pin[slot][1][:n], pin[slot][2][:n] = sl["fr_x"], sl["fr_y"], sl["t"]
                    t_issue = time.perf_counter()
                    if n > 0:
                        a.upload_events_async(pin[slot][0], pin[slot][1], pin[slot][2], n)
                    s.drop()
                    return s, n, t_issue

                k = 0
                nxt = put(0)
                # Near the end of the batch a lane stops claiming AHEAD: a slice prefetched by a lane that sits in a long solve
                # (33 000 iterations against a mean of 6 400 at config 5) could not be taken by the lanes and ranks that have
                # gone idle meanwhile.  Once fewer slices are left than there are lanes in the job, the next one is claimed
                # when this lane is free for it (its upload then is not hidden: 0.3 ms against seconds of solve).
                drain_at = max(1, concurrent) * (1 if static else max(1, world))
                while nxt is not None:
                    cur, n, t0 = nxt
                    t_solve = time.perf_counter()
                    if n > 0:
                        a.commit_upload()
                    else:
                        a.upload_events(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32))
                    k ^= 1
                    ahead = work.remaining() >= drain_at
                    nxt = put(k) if ahead else None          # the next slice's DMA overlaps this slice's solve
                    o.res_x, o.res_y, o.max_iter, o.want_uv = cur.height, cur.width, max_iter, 1 if want_flow_digest else 0
                    a.set_cloud(scale, cur.height, cur.width)
                    rc, m, info = a.run(o)
                    rec = {"rc": int(rc), "iterations": int(info.iterations), "model": m.as_dict(), "events": n, "rank": rank,
                           "lane": lane_no}
                    if want_flow_digest:
                        u, v = a.compute_uv()
                        rec["flow_sha1"] = hashlib.sha1(u.tobytes() + v.tobytes()).hexdigest()
                    a.synchronize()
                    now = time.perf_counter()
                    rec["ms"] = 1e3 * (now - t0)       # from the issue of the slice's upload (under the previous solve) to its model
                    rec["solve_ms"] = 1e3 * (now - t_solve)   # the lane's own time for this slice: commit, set_cloud, run, read-back
                    rec["t1"] = now - t_start          # seconds since this rank's farm started

                    with lock:
                        results[cur.index] = rec
                    cur = None
                    if not ahead:
                        nxt = put(k)
            except Exception as e:   # noqa: BLE001 -- travels with the records, raised after the gather
                with lock:
                    results[("error", rank, threading.get_ident())] = {"error": "%s: %s" % (type(e).__name__, e),
                                                                       "slice": None if cur is None else cur.index, "rank": rank}
            finally:
                if a is not None:
                    try:
                        a.close()
                    except Exception:   # noqa: BLE001
                        pass

        threads = [threading.Thread(target=lane, args=(k,)) for k in range(max(1, min(concurrent, n_mine)))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    merged = gather(results, None if static and not (dist is not None and dist.is_initialized()) else dist)
    _raise_lane_errors(merged)
    return merged


def spawn_local_ranks(n, script, argv, env_extra=None, poll_s=0.05, pids_out=None):
    """One process per rank on this node without a launcher: run `n` copies of `python script argv...`, rank r with RANK /
    LOCAL_RANK = r, WORLD_SIZE = n and a fresh MASTER_ADDR=127.0.0.1 / MASTER_PORT in its environment -- exactly what the
    ranks find under `python -m torch.distributed.run` -- and wait for all of them.  Rank 0 writes to the inherited stdout.

    A rank that fails takes the job down: the moment any rank exits non-zero the others (which may be waiting for it in a
    barrier or a gather) are terminated by PID -- killed if they ignore that for 5 s -- and the return value is the FIRST
    failing rank's exit code (a signal's negative code as 128 + signal).  Returns 0 only if every rank did.  No process of
    the job outlives this call."""
    import os
    import socket
    import subprocess
    import sys
    import time
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), BF_BENCH_SPAWNED="1")
        env.update(env_extra or {})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env))
        if pids_out is not None:
            pids_out.append(procs[-1].pid)
    rc = 0
    live = list(procs)
    deadline = None
    try:
        while live:
            time.sleep(poll_s)
            for p_ in list(live):
                code = p_.poll()
                if code is None:
                    continue
                live.remove(p_)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 128 - code
                    for q_ in live:   # (exact PIDs of our own children)
                        q_.terminate()
                    deadline = time.monotonic() + 5.0
            if deadline is not None and live and time.monotonic() > deadline:
                for q_ in live:
                    q_.kill()
                deadline = None
    finally:
        for p_ in procs:   # (an exception in here -- KeyboardInterrupt -- must not leave ranks behind either)
            if p_.poll() is None:
                p_.kill()
                p_.wait()
    return rc


def make_share_dir(n_bytes, tag="farm"):
    """A fresh directory every rank of this node can reach for the slices the ranks exchange (prepare()): under /dev/shm if
    it has `n_bytes` + 10 % free, else under the temporary directory, else an error that says how much was needed -- a
    512-slice batch of 1M events is 8 GB, more than a container's default tmpfs.  The caller removes it (try / finally)."""
    import os
    import shutil
    import tempfile
    need = int(n_bytes * 1.1) + (1 << 20)
    for base in ("/dev/shm", tempfile.gettempdir()):
        if os.path.isdir(base) and os.access(base, os.W_OK) and shutil.disk_usage(base).free >= need:
            return tempfile.mkdtemp(prefix="bf_%s_" % tag, dir=base)
    raise RuntimeError("slice exchange needs %.1f GB; neither /dev/shm nor %s has that much free" %
                       (need / 1e9, tempfile.gettempdir()))


def exit_with(main):
    """Run a rank's main() and leave the process with an exit code that means something: 0, the code of a SystemExit, or 1
    after any other exception (traceback on stderr) -- through os._exit, because a rank that unwinds normally with its
    process group still alive is aborted by the group's watchdog threads ("terminate called without an active exception",
    exit code 134), which would hide the failing rank's own code from the launcher."""
    import os
    import sys
    import traceback
    code = 0
    try:
        main()
    except SystemExit as e:
        code = e.code if isinstance(e.code, int) else (0 if e.code is None else 1)
        if code and not isinstance(e.code, int):
            print(e.code, file=sys.stderr)
    except BaseException:   # noqa: BLE001
        traceback.print_exc()
        code = 1
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(code)
