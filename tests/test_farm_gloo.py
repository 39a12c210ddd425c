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


# ---- the shared task queue (dvs_flow.h:200-231 models work as a queue of (events, model) tasks) ----------------------------

def _process_uneven(i):
    """Deliberately uneven slices: the even ones -- ALL of rank 0's under the round robin i % 2 -- cost ~10 x the odd ones
    (more events, and the same optimisation solved several times over)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from better_flow_amd import synth
    heavy = i % 2 == 0
    sl = synth.make_slice(6000 if heavy else 2000, 60, 80, 0.04, seed=300 + i)
    for _ in range(12 if heavy else 1):
        c = oracle.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
        w = c.set_cloud(3, 60, 80)
        m = oracle.Model()
        rc, loop, _ = c.run(w, m, max_iter=60, res_x=60, res_y=80)
    return {"rc": rc, "iters": int(loop.itercount), "model": m.as_dict(), "events": len(sl["t"])}


def _queue_worker(rank, world, port, n_slices, use_costs, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from better_flow_amd import farm
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    costs = [24000 if i % 2 == 0 else 2000 for i in range(n_slices)] if use_costs else None
    work = farm.SliceQueue(n_slices, costs=costs, dist=dist, name="test")
    dist.barrier()
    res = farm.run_queue(work, _process_uneven, lanes=1)
    for r in res.values():
        r["rank"] = rank
        r["ms"] = 1e3 * (r["t1"] - r["t0"])
    merged = farm.gather(res, dist)
    if rank == 0:
        q.put(merged)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("use_costs", [False, True], ids=["index_order", "longest_first"])
def test_two_ranks_share_one_queue_and_finish_balanced(use_costs):
    """Every lane of every rank claims its next slice from ONE counter (TCPStore.add of the gloo group: control messages only).
    With slices whose cost alternates 10 : 1, the round robin i % 2 gives rank 0 every long one; the queue must hand every
    slice out exactly once, return the bits of a single process, and end well before the round robin would have."""
    sys.path.insert(0, ROOT)
    from better_flow_amd import farm
    n_slices = 12
    single = {i: _process_uneven(i) for i in range(n_slices)}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_queue_worker, args=(r, 2, port, n_slices, use_costs, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(merged) == list(range(n_slices))   # every slice once (gather raises on a slice processed twice)
    for i in range(n_slices):
        for k in ("rc", "iters", "model", "events"):
            assert merged[i][k] == single[i][k], (i, k)
    dur = [merged[i]["ms"] for i in range(n_slices)]
    assert min(dur[0::2]) > 3 * max(dur[1::2]), "the test's slices are not uneven enough to show anything"
    bal = farm.balance(merged, 2)
    static = farm.simulate_makespan(dur, 2, 1, static=True)
    dynamic = 1e3 * max(bal["busy_s"])
    heavy_per_rank = [sum(1 for i in range(0, n_slices, 2) if merged[i]["rank"] == r) for r in range(2)]
    print("slices per rank %s (long ones %s), busy %.0f / %.0f ms, imbalance %.2f; the round robin would have taken %.0f ms, "
          "the queue took %.0f ms" % (bal["slices"], heavy_per_rank, 1e3 * bal["busy_s"][0], 1e3 * bal["busy_s"][1], bal["imbalance"],
                                      static, dynamic))
    assert min(heavy_per_rank) >= 2, heavy_per_rank          # both ranks took long slices
    assert dynamic <= 0.8 * static, (dynamic, static)        # (ideal: 0.55)
    assert bal["imbalance"] <= 1.25, bal
    if use_costs:   # longest first: the six long slices are the first six claims
        first = sorted(range(n_slices), key=lambda i: merged[i]["t0"])[:6]
        assert all(i % 2 == 0 for i in first), first


def test_makespan_simulation_and_queue_order():
    sys.path.insert(0, ROOT)
    from better_flow_amd import farm
    d = [10, 1, 10, 1, 10, 1, 10, 1, 1, 1, 1, 1]
    assert farm.simulate_makespan(d, 2, 1, static=True) == 42      # rank 0: 10 + 10 + 10 + 10 + 1 + 1
    assert farm.simulate_makespan(d, 2, 1) == 24                   # one queue, index order
    assert farm.simulate_makespan(d, 2, 1, costs=d) == 24          # longest first
    assert farm.simulate_makespan(d, 1, 1) == sum(d) == farm.simulate_makespan(d, 1, 1, static=True)
    assert farm.simulate_makespan(d, 2, 2) <= 13
    q = farm.SliceQueue(5, costs=[1, 5, 3, 5, 2])
    assert [q.claim() for _ in range(7)] == [1, 3, 2, 4, 0, None, None]
    q = farm.SliceQueue(3)
    assert [q.claim() for _ in range(4)] == [0, 1, 2, None]
    with pytest.raises(ValueError):
        farm.SliceQueue(3, costs=[1, 2])


def test_numa_binding_is_safe_everywhere():
    """bf_bind_thread_to_numa_node (include/bf_accel.h): binds the calling thread to the node's CPUs the process may use and
    never fails on a machine without that node, without NUMA, or inside a cpuset -- it then does nothing."""
    sys.path.insert(0, ROOT)
    import threading
    from better_flow_amd import accel
    before = os.sched_getaffinity(0)
    out = {}

    def t():
        out["none"] = accel.bind_thread_to_numa_node(-1)
        out["far"] = accel.bind_thread_to_numa_node(4095)
        out["aff_untouched"] = os.sched_getaffinity(0) == before
        out["node0"] = accel.bind_thread_to_numa_node(0)
        out["aff"] = os.sched_getaffinity(0)
    th = threading.Thread(target=t)
    th.start()
    th.join()
    assert out["none"] == 0 and out["far"] == 0 and out["aff_untouched"]
    path = "/sys/devices/system/node/node0/cpulist"
    if os.path.exists(path):
        cpus = set()
        for part in open(path).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        want = cpus & before
        if want:
            assert out["node0"] == len(want) and out["aff"] == want
        else:
            assert out["node0"] == 0 and out["aff"] == before
    else:
        assert out["node0"] == 0 and out["aff"] == before
    assert os.sched_getaffinity(0) == before   # (the binding is the calling thread's, not the process's)


# ---- eight ranks: the target world size of BASELINE config 5, on CPU (no 8-GPU node has ever run this job) -------------------

def _run_eight(tmp_path, n_slices, extra=()):
    sys.path.insert(0, ROOT)
    import json
    import time
    from better_flow_amd import farm
    share = farm.make_share_dir(n_slices * 6000 * 16, tag="test8")
    out = str(tmp_path / "merged.json")
    try:
        t0 = time.time()
        pids = []
        rc = farm.spawn_local_ranks(8, os.path.join(ROOT, "tests", "farm_rank_main.py"), [out, str(n_slices), share] + list(extra),
                                    pids_out=pids)
        wall = time.time() - t0
        assert len(pids) == 8
        for pid in pids:   # no rank outlives the launcher's return
            with pytest.raises(ProcessLookupError):
                os.kill(pid, 0)
        merged = json.load(open(out)) if os.path.exists(out) else None
    finally:
        import shutil
        shutil.rmtree(share, ignore_errors=True)
    return rc, merged, wall


@pytest.mark.timeout(600)
def test_eight_ranks_one_queue_shared_slices(tmp_path):
    """World size 8 through the launcher `bench.py --gpus 8` uses (farm.spawn_local_ranks): slices generated by their round-
    robin owners and exchanged through a shared directory, claimed from ONE queue longest first, gathered on every rank.
    Every slice exactly once, the bits of a single process, every rank took work, nobody holds the job up."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from better_flow_amd import farm
    import farm_rank_main as frm
    n = 40
    rc, merged, wall = _run_eight(tmp_path, n)
    assert rc == 0
    assert sorted(int(k) for k in merged) == list(range(n))      # (gather raises on a slice processed twice)
    for i in range(n):
        single = frm.solve(frm.spec_of(farm, i), 1)
        for k in ("rc", "iters", "model", "events"):
            assert merged[str(i)][k] == single[k], (i, k)
    ranks = [merged[str(i)]["rank"] for i in range(n)]
    assert sorted(set(ranks)) == list(range(8)), ranks           # every rank claimed something
    first = sorted(range(n), key=lambda i: merged[str(i)]["t0"])[:8]
    assert all(i % 5 == 0 for i in first), first                 # longest first: the eight heavy slices are the first claims
    bal = farm.balance({i: merged[str(i)] for i in range(n)}, 8)
    print("8 ranks: slices per rank %s, busy %s s, imbalance %.2f, wall %.1f s" %
          (bal["slices"], ["%.2f" % b for b in bal["busy_s"]], bal["imbalance"], wall))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode,code", [("raise", 1), ("exit:7", 7)], ids=["exception", "rank_dies"])
def test_eight_ranks_a_failing_rank_takes_the_job_down(tmp_path, mode, code):
    """The rank that claims slice 20 (a light one: the eight heavy slices are out first, everybody is busy) fails on it -- by an exception inside its worker (then EVERY rank raises after the gather: nobody
    waits in a collective the failing rank never joins), or by dying outright with exit code 7 (then the launcher terminates
    the ranks that wait for it).  The job ends promptly, with the failing rank's exit code, and leaves no process behind."""
    rc, merged, wall = _run_eight(tmp_path, 40, ["--fail-slice", "20", "--fail-mode", mode])
    assert rc == code, rc
    assert merged is None                                          # rank 0 never got to write a result
    assert wall < 120, wall                                        # (no gloo timeout was waited for: that is 30 minutes)


def test_farm_helpers_on_cpu(tmp_path):
    """The pieces of the farm that need neither a GPU nor a process group: the launcher's exit codes (first failing rank wins, a
    signal becomes 128 + signal, nothing is left running), the slice-exchange directory (fresh, with room, or a clear error), and
    run_farm's refusal to hand the whole batch to every rank when it is told `world > 1` without a process group of that size."""
    sys.path.insert(0, ROOT)
    import warnings
    from better_flow_amd import farm
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time, signal\n"
                      "r = int(os.environ['RANK']); mode = sys.argv[1]\n"
                      "assert os.environ['WORLD_SIZE'] == '3' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
                      "if mode == 'ok': sys.exit(0)\n"
                      "if mode == 'code' and r == 1: time.sleep(0.2); sys.exit(5)\n"
                      "if mode == 'signal' and r == 2: time.sleep(0.2); os.kill(os.getpid(), signal.SIGKILL)\n"
                      "time.sleep(60)\n")
    pids = []
    assert farm.spawn_local_ranks(3, str(script), ["ok"], pids_out=pids) == 0 and len(pids) == 3
    assert farm.spawn_local_ranks(3, str(script), ["code"]) == 5          # ranks 0 and 2 sat in their sleep: terminated
    assert farm.spawn_local_ranks(3, str(script), ["signal"]) == 128 + 9
    d = farm.make_share_dir(1 << 20, tag="unit")
    try:
        assert os.path.isdir(d) and os.listdir(d) == [] and d != farm.make_share_dir.__name__
        d2 = farm.make_share_dir(1 << 20, tag="unit")
        assert d2 != d                                                      # a fresh directory every time
        os.rmdir(d2)
    finally:
        os.rmdir(d)
    with pytest.raises(RuntimeError, match="GB"):
        farm.make_share_dir(1 << 60)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert farm.run_farm([], rank=1, world=2, dist=None) == {}
    assert any("static sharding" in str(x.message) for x in w)
