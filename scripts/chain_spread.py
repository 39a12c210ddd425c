"""Config-1 STM chain through the command line: the oracle's own spread (events forward vs reversed inside every slice,
tests/shim BF_SHIM_EVENT_ORDER) next to the GPU-vs-oracle difference, per flag set."""
import os, re, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "shim"))
import build as shim_build
from better_flow_amd import synth
exe = shim_build.build()
gpu = os.path.join(ROOT, "better_flow_amd", "host", "bf_motion_compensator")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
sl = synth.make_slice(n, 180, 240, 0.1, seed=5)
synth.write_txt("/tmp/ev.txt", sl)

def run(e, extra, env=None):
    en = dict(os.environ); en.update(env or {})
    r = subprocess.run([e] + extra + ["-o", "/tmp/o.txt", "/tmp/ev.txt"], env=en, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    m = re.search(r"slices: (\d+) \(skipped (\d+)\), minimizer iterations: (\d+)", r.stdout.decode())
    return m.groups(), np.loadtxt("/tmp/o.txt")

for extra in ([], ["--stm-disable"], ["--max-iter=10"]):
    sa, a = run(exe, extra)
    sb, b = run(exe, extra, {"BF_SHIM_EVENT_ORDER": "reversed"})
    sg, g = run(gpu, extra)
    print(extra, "oracle", sa, "reversed", sb, "gpu", sg)
    for col, nm in ((4, "v"), (5, "u")):
        print("   %s: oracle fwd-rev max %.3e | gpu-oracle max %.3e mean %.3e rel max %.3e" % (
            nm, np.abs(a[:, col] - b[:, col]).max(), np.abs(a[:, col] - g[:, col]).max(),
            abs(a[:, col].mean() - g[:, col].mean()), (np.abs(a[:, col] - g[:, col]) / np.maximum(np.abs(a[:, col]), 1e-9)).max()))
