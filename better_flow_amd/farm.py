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
