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
        if self.arrays is not None:
            return self.arrays
        from . import synth
        return synth.make_slice(self.events, self.height, self.width, self.duration_s, seed=self.seed)


def run_farm(specs, rank=0, world=1, device=0, concurrent=4, scale=3, max_iter=-1, want_flow_digest=False, dist=None,
             options=None):
    """The slice farm on the HIP path (dvs_flow.h:200-231's task queue, over ranks and slice contexts).

    Rank `rank` takes the slices i with i % world == rank and runs them with `concurrent` slice contexts (one host
    thread + bf_ctx + HIP stream each: contexts pull the rank's slices from a shared queue, so a slow slice does not
    hold the others up); every slice is a cold start (STM off: independent slices).  The per-slice records --
    return code, iterations, the 88-byte model, events, milliseconds, optionally a digest of the per-event flow -- are
    merged on every rank with all_gather_object when `dist` is an initialised torch.distributed (gloo: no data-path
    collective exists on this path).  Returns {slice index: record}."""
    import hashlib
    import queue
    import threading
    import time
    from . import accel
    mine = [s for s in specs if s.index % world == rank]
    results = {}
    if mine:
        loaded = {}
        hmax = max(s.height for s in mine)
        wmax = max(s.width for s in mine)
        nmax = max(s.events for s in mine)
        work = queue.Queue()
        for s in mine:
            work.put(s)
        lock = threading.Lock()
        errors = []

        def lane():
            try:
                a = accel.Accel(device=device, max_events=nmax, max_rows=scale * hmax + scale, max_cols=scale * wmax + scale)
                if concurrent > 1:
                    a.set_option("co_schedule", 1)
                for k, v in (options or {}).items():
                    a.set_option(k, v)
                o = a.default_opts()
                while True:
                    try:
                        s = work.get_nowait()
                    except queue.Empty:
                        break
                    sl = s.load()
                    o.res_x, o.res_y, o.max_iter, o.want_uv = s.height, s.width, max_iter, 1 if want_flow_digest else 0
                    t0 = time.perf_counter()
                    a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
                    a.set_cloud(scale, s.height, s.width)
                    rc, m, info = a.run(o)
                    rec = {"rc": int(rc), "iterations": int(info.iterations), "model": m.as_dict(), "events": len(sl["t"]),
                           "rank": rank}
                    if want_flow_digest:
                        u, v = a.compute_uv()
                        rec["flow_sha1"] = hashlib.sha1(u.tobytes() + v.tobytes()).hexdigest()
                    a.synchronize()
                    rec["ms"] = 1e3 * (time.perf_counter() - t0)
                    with lock:
                        results[s.index] = rec
                a.close()
            except Exception as e:   # noqa: BLE001 -- reported to the caller below
                errors.append(e)

        threads = [threading.Thread(target=lane) for _ in range(max(1, min(concurrent, len(mine))))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        del loaded
    return gather(results, dist)
