"""The slice farm on the HIP path (better_flow_amd/farm.py: run_farm): two gloo ranks -- both on GPU 0 of the test box --
with two slice contexts each claim independent slices from ONE shared queue (TCPStore.add of the gloo group; and, for the
A/B form, the round robin i -> rank i % 2); every slice's return code, iteration count, model and per-event flow must be
bit-identical to the single-rank run (multi-GPU level).  No collective is on the data path: the process group
carries the claims and gathers the 88-byte models."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

N_SLICES, H, W, EVENTS = 7, 180, 240, 60000


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _specs():
    from better_flow_amd import farm
    return [farm.SliceSpec(i, H, W, events=EVENTS, duration_s=0.04, seed=500 + i) for i in range(N_SLICES)]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from better_flow_amd import farm
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    merged = farm.run_farm(_specs(), rank=rank, world=world, device=0, concurrent=2, want_flow_digest=True, dist=dist)
    static = farm.run_farm(_specs(), rank=rank, world=world, device=0, concurrent=2, want_flow_digest=True, dist=dist, static=True)
    if rank == 0:
        q.put((merged, static))
    dist.barrier()
    dist.destroy_process_group()


def _strip(rec):
    return {k: v for k, v in rec.items() if k not in ("ms", "solve_ms", "rank", "lane", "t1")}


def test_two_rank_farm_on_the_gpu_matches_single_rank():
    sys.path.insert(0, ROOT)
    from better_flow_amd import farm
    single = farm.run_farm(_specs(), rank=0, world=1, device=0, concurrent=1, want_flow_digest=True)
    assert sorted(single) == list(range(N_SLICES))
    assert all(r["rc"] == 0 and r["iterations"] > 20 for r in single.values())
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged, static = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(merged) == list(range(N_SLICES)) == sorted(static)
    for i in range(N_SLICES):
        assert static[i]["rank"] == i % 2                      # the round robin, kept for A/B runs
        assert _strip(merged[i]) == _strip(single[i]), i       # the shared queue: whoever took the slice, the same bits
        assert _strip(static[i]) == _strip(single[i]), i
    assert {merged[i]["rank"] for i in range(N_SLICES)} == {0, 1}, "both ranks must have claimed slices"
    bal = farm.balance(merged, 2)
    assert sum(bal["slices"]) == N_SLICES and bal["imbalance"] < 2.0, bal
