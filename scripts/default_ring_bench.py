"""The reference's own configuration through the product command line: its compiled-in ring (50 000 events, a slice every
20 000 events / 33 ms, warm-start chain from slice to slice) on a 240x180 (or --height / --width) stream of 250 000 events
per 33 ms.  Every slice is small and the chain is sequential, so this measures the LATENCY of the loop -- launches per
iteration -- not its throughput.

    python scripts/default_ring_bench.py [--events 10000000] [--options fused=0]

Prints one JSON line per option set: the CLI's --timing record."""
import argparse
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from better_flow_amd import synth  # noqa: E402


def run(events=10000000, height=180, width=240, options="", extra=(), reps=2, keep=None):
    d = keep or tempfile.mkdtemp(prefix="bf_ring_")
    slices = max(1, events // 250000)
    path = os.path.join(d, "ring_%dx%d_%d.bin" % (width, height, slices))
    if not os.path.exists(path):
        synth.write_stream_bin(path, slices, 250000, height, width, duration_s=0.033)
    cli = os.path.join(ROOT, "better_flow_amd", "host", "bf_motion_compensator")
    env = dict(os.environ)
    if options:
        env["BF_ACCEL_OPTIONS"] = options
    best = None
    for _ in range(reps):
        r = subprocess.run([cli, "--quiet", "--timing", "--res-x=%d" % height, "--res-y=%d" % width] + list(extra) + [path],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        if r.returncode != 0:
            raise RuntimeError(r.stderr.decode()[-2000:])
        rec = json.loads([ln for ln in r.stderr.decode().splitlines() if ln.startswith("{")][-1])
        if best is None or rec["stream_s"] < best["stream_s"]:
            best = rec
    best.update({"geometry": "%dx%d" % (width, height), "options": options, "flags": " ".join(extra)})
    if not keep:
        os.remove(path)
        os.rmdir(d)
    return best


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--events", type=int, default=10000000)
    ap.add_argument("--height", type=int, default=180)
    ap.add_argument("--width", type=int, default=240)
    ap.add_argument("--options", action="append", default=None, help="BF_ACCEL_OPTIONS value; repeatable")
    ap.add_argument("--extra", default="")
    a = ap.parse_args()
    keep = tempfile.mkdtemp(prefix="bf_ring_")
    for opt in (a.options or [""]):
        print(json.dumps(run(a.events, a.height, a.width, opt, tuple(a.extra.split()), keep=keep)))
