"""The CPU oracle (oracle/bf_oracle.c, a port of the reference path -- parity unpinned) on BASELINE config 2's slice from a cold
start to the reference loop's OWN termination, one core: the figure bench.py's `cpu_baseline` extrapolates to from a bounded
sample, measured once in full.  Writes profiles/cpu_to_termination.json (committed; bench.py quotes it beside its own sample).

    python scripts/cpu_to_termination.py
"""
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from better_flow_amd import synth  # noqa: E402

H, W, s = 260, 346, 3
sl = synth.make_slice(1000000, H, W, 0.030, seed=1)
oc = oracle.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
ow = oc.set_cloud(s, H, W)
om = oracle.Model()
t0 = time.perf_counter()
rc, loop, _ = oc.run(ow, om, res_x=H, res_y=W)
dt = time.perf_counter() - t0
cpu = ""
try:
    cpu = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
except Exception:  # noqa: BLE001
    cpu = platform.processor()
rec = {"what": "oracle/bf_oracle.c, cold start to the loop's own termination, 1 core, the build container (not the GPU box)",
       "workload": "%d-event 30 ms slice, %dx%d, scale %d, seed 1" % (len(sl["t"]), W, H, s), "rc": int(rc),
       "iterations": int(loop.itercount), "seconds": dt, "ms_per_iteration": 1e3 * dt / max(1, loop.itercount),
       "mevents_per_s": len(sl["t"]) / dt / 1e6, "cpu": cpu, "cores": 1}
json.dump(rec, open(os.path.join(ROOT, "profiles", "cpu_to_termination.json"), "w"), indent=1)
print(json.dumps(rec))
