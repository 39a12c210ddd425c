"""Front-end throughput: events/s from a binary event FILE to the last slice's model, through the product command line
(bf_motion_compensator on the stream engine) -- BASELINE config-2 / config-3 style rolling 30 ms slices of ~1M events.

    python scripts/front_end_bench.py [--slices 20] [--height 260 --width 346] [--extra "--sync"] [-o]

Prints one JSON line: the CLI's own --timing record (init = device contexts + pinned ring; stream = file open -> last
model; output = -o table + text) plus the whole-process wall clock."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from better_flow_amd import synth  # noqa: E402


def run(slices=20, events=1000000, height=260, width=346, extra=(), with_output=False, warm_file_cache=True, reps=3, keep=None):
    d = keep or tempfile.mkdtemp(prefix="bf_fe_")
    path = os.path.join(d, "stream_%dx%d_%d.bin" % (width, height, slices))
    if not os.path.exists(path):
        n = synth.write_stream_bin(path, slices, events, height, width)
    else:
        import numpy as np
        n = int(np.fromfile(path, dtype="<u8", count=2)[1])
    cli = os.path.join(ROOT, "better_flow_amd", "host", "bf_motion_compensator")
    args = [cli, "--quiet", "--timing", "--res-x=%d" % height, "--res-y=%d" % width, "--max-events=%d" % int(events * 1.1),
            "--span=0.03", "--refresh-time=0.03", "--refresh-event-count=1000000000"] + list(extra)
    out = os.path.join(d, "flow.txt")
    if with_output:
        args += ["-o", out]
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        r = subprocess.run(args + [path], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            raise RuntimeError(r.stderr.decode()[-2000:])
        rec = json.loads([ln for ln in r.stderr.decode().splitlines() if ln.startswith("{")][-1])
        rec["process_wall_s"] = wall
        if best is None or rec["stream_s"] < best["stream_s"]:
            best = rec
    best.update({"file_events": n, "file_mb": os.path.getsize(path) / 1e6, "geometry": "%dx%d" % (width, height),
                 "flags": " ".join(extra), "with_output": with_output})
    if with_output and os.path.exists(out):
        best["output_mb"] = os.path.getsize(out) / 1e6
        os.remove(out)
    if not keep:
        os.remove(path)
        os.rmdir(d)
    return best


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--slices", type=int, default=20)
    ap.add_argument("--events", type=int, default=1000000)
    ap.add_argument("--height", type=int, default=260)
    ap.add_argument("--width", type=int, default=346)
    ap.add_argument("--extra", default="")
    ap.add_argument("-o", dest="with_output", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    print(json.dumps(run(a.slices, a.events, a.height, a.width, a.extra.split(), a.with_output, reps=a.reps)))
