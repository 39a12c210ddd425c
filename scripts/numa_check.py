"""Where do a feeder thread's pinned buffers live?  (NUMA-local pinned buffers.)

A thread allocates 64 MB of pinned host memory through the C-ABI (bf_host_alloc), touches it, and /proc/self/numa_maps says on
which host NUMA node the pages are -- once from an unbound thread, once from a thread bound with bf_bind_thread_to_device_numa,
and once from a thread deliberately bound to the OTHER node's CPUs (what the runtime does on its own, whoever calls)."""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from better_flow_amd import accel  # noqa: E402


def pages_by_node(addr):
    for ln in open("/proc/self/numa_maps"):
        parts = ln.split()
        if int(parts[0], 16) <= addr < int(parts[0], 16) + (1 << 40):
            lo = int(parts[0], 16)
            if lo == addr or (lo < addr and any(p.startswith("N") for p in parts)):
                if lo == addr:
                    return {p.split("=")[0]: int(p.split("=")[1]) for p in parts if p[0] == "N" and "=" in p and p[1].isdigit()}
    return {}


def probe(label, bind):
    out = {}

    def run():
        bind()
        a = accel.Accel(device=0, max_events=4096, max_rows=64, max_cols=64)
        arr = a.pinned_int32(16 << 20)   # 64 MB
        arr[:] = 1
        out["cpus"] = sorted(os.sched_getaffinity(0))
        out["pages"] = pages_by_node(arr.ctypes.data)
        a.close()
    t = threading.Thread(target=run)
    t.start()
    t.join()
    c = out["cpus"]
    print("%-46s thread on CPUs %d..%d (%d), 64 MB of pinned memory: pages per node %s" % (label, c[0], c[-1], len(c), out["pages"]))


node = accel.device_numa_node(0)
print("device 0 sits on host NUMA node", node)
probe("unbound thread", lambda: None)
probe("bf_bind_thread_to_device_numa(0)", lambda: accel.bind_thread_to_device_numa(0))
other = 1 - node if node in (0, 1) else 0
probe("thread bound to the OTHER node (%d)" % other, lambda: accel.bind_thread_to_numa_node(other))
