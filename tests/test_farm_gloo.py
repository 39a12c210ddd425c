"""N > 1 path on CPU: 2 gloo ranks shard independent slices with no data-path collective;
results must equal the single-process run slice by slice.  (On the GPU box the per-slice
work is bf_run; here the CPU oracle stands in as the per-slice worker -- allowed in tests.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _process(i):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from better_flow_amd import synth
    sl = synth.make_slice(3000, 60, 80, 0.04, seed=100 + i)
    c = oracle.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
    w = c.set_cloud(3, 60, 80)
    m = oracle.Model()
    rc, loop, _ = c.run(w, m, res_x=60, res_y=80)
    return {"rc": rc, "iters": int(loop.itercount), "model": m.as_dict(), "events": len(sl["t"])}


def _worker(rank, world, port, n_slices, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from better_flow_amd import farm
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    mine = farm.shard(n_slices, rank, world)
    res = farm.run_shard(mine, _process)
    dist.barrier()
    merged = farm.gather(res, dist)
    if rank == 0:
        q.put(merged)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_farm_matches_single_process():
    sys.path.insert(0, ROOT)
    from better_flow_amd import farm
    n_slices = 5
    assert farm.shard(n_slices, 0, 2) == [0, 2, 4] and farm.shard(n_slices, 1, 2) == [1, 3]
    single = farm.gather(farm.run_shard(farm.shard(n_slices, 0, 1), _process))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_slices, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(merged) == list(range(n_slices))
    for i in range(n_slices):
        assert merged[i] == single[i], i      # order independent, bit identical
    with pytest.raises(ValueError):
        farm.shard(4, 2, 2)
