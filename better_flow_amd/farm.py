"""Slice farm: independent time slices sharded over ranks / GPUs.

A slice is a complete optimisation problem (the reference already models work as a queue
of (events, model) tasks, dvs_flow.h:200-202), so ranks never exchange data: slice i goes
to rank i mod world, every rank runs its own bf_ctx, and the host gathers 11 doubles per
slice.  No RCCL collective is on the data path."""


def shard(n_slices, rank, world):
    """Indices of the slices rank `rank` of `world` processes owns (round robin)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, n_slices, world))


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

    def load(self):
        if self.arrays is None:   # generated once, outside the farm's lanes (and outside any timed region: prepare())
            from . import synth
            self.arrays = synth.make_slice(self.events, self.height, self.width, self.duration_s, seed=self.seed)
        return self.arrays

    def count(self):
        return len(self.arrays["t"]) if self.arrays is not None else self.events


def prepare(specs, rank=0, world=1):
    """Load (generate) the event arrays of this rank's slices up front -- the farm's lanes only move and solve them."""
    for s in specs:
        if s.index % world == rank:
            s.load()
    return specs


def run_farm(specs, rank=0, world=1, device=0, concurrent=4, scale=3, max_iter=-1, want_flow_digest=False, dist=None,
             options=None):
    """The slice farm on the HIP path (dvs_flow.h:200-231's task queue, over ranks and slice contexts).

    Rank `rank` takes the slices i with i % world == rank and runs them with `concurrent` slice contexts (one host
    thread + bf_ctx + HIP stream + copy stream each: contexts pull the rank's slices from a shared queue, so a slow slice
    does not hold the others up); every slice is a cold start (STM off: independent slices).  A lane stages its next
    slice in pinned memory and uploads it on its copy stream (bf_upload_events_async) while the current one is being
    solved.  The per-slice records -- return code, iterations, the 88-byte model, events, milliseconds, optionally a digest
    of the per-event flow, or the text of an error -- are merged on every rank with all_gather_object when `dist` is an
    initialised torch.distributed (gloo: no data-path collective exists on this path).  An error in any lane of any rank
    is raised on EVERY rank, after the gather (no rank is left waiting in a collective).  Returns {slice index: record}.
    The native form of the same farm, for C++ callers and the command line, is better_flow/slice_farm.h."""
    import hashlib
    import queue
    import threading
    import time
    import numpy as np
    from . import accel
    mine = [s for s in specs if s.index % world == rank]
    results = {}
    if mine:
        for s in mine:
            s.load()
        hmax = max(s.height for s in mine)
        wmax = max(s.width for s in mine)
        nmax = max(s.count() for s in mine)
        work = queue.Queue()
        for s in mine:
            work.put(s)
        lock = threading.Lock()

        def lane():
            a = None
            cur = None
            try:
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
                    try:
                        s = work.get_nowait()
                    except queue.Empty:
                        return None
                    sl = s.load()
                    n = len(sl["t"])
                    pin[slot][0][:n], pin[slot][1][:n], pin[slot][2][:n] = sl["fr_x"], sl["fr_y"], sl["t"]
                    t_issue = time.perf_counter()
                    if n > 0:
                        a.upload_events_async(pin[slot][0], pin[slot][1], pin[slot][2], n)
                    return s, n, t_issue

                k = 0
                nxt = put(0)
                while nxt is not None:
                    cur, n, t0 = nxt
                    if n > 0:
                        a.commit_upload()
                    else:
                        a.upload_events(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32))
                    k ^= 1
                    nxt = put(k)          # the next slice's DMA overlaps this slice's solve
                    o.res_x, o.res_y, o.max_iter, o.want_uv = cur.height, cur.width, max_iter, 1 if want_flow_digest else 0
                    a.set_cloud(scale, cur.height, cur.width)
                    rc, m, info = a.run(o)
                    rec = {"rc": int(rc), "iterations": int(info.iterations), "model": m.as_dict(), "events": n, "rank": rank}
                    if want_flow_digest:
                        u, v = a.compute_uv()
                        rec["flow_sha1"] = hashlib.sha1(u.tobytes() + v.tobytes()).hexdigest()
                    a.synchronize()
                    rec["ms"] = 1e3 * (time.perf_counter() - t0)
                    with lock:
                        results[cur.index] = rec
                    cur = None
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

        threads = [threading.Thread(target=lane) for _ in range(max(1, min(concurrent, len(mine))))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    merged = gather(results, dist)
    errors = [v for k, v in merged.items() if isinstance(k, tuple)]
    if errors:
        raise RuntimeError("slice farm: %d lane(s) failed; first: rank %s, slice %s: %s" %
                           (len(errors), errors[0]["rank"], errors[0]["slice"], errors[0]["error"]))
    return merged
