"""How many stencil work-groups a CU holds at once (debug build: make -C better_flow_amd/csrc tl TLX=-DBF_CENSUS): every
work-group counts itself in and out on its CU (HW_ID / XCC_ID); prints the histogram of the per-CU maxima over a 40-iteration run.
    BF_RUN_N / BF_RUN_H / BF_RUN_W, BF_OPTS=k=v,k=v as in timeline_k3.py"""
import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BF_TIMELINE"] = "/tmp/bf_tl.txt"
os.environ["BF_ACCEL_LIB"] = os.path.join(ROOT, "better_flow_amd", "libbf_accel_tl.so")
from better_flow_amd import accel, synth
N, H, W, s = int(os.environ.get("BF_RUN_N", "1000000")), int(os.environ.get("BF_RUN_H", "260")), int(os.environ.get("BF_RUN_W", "346")), 3
sl = synth.make_slice(N, H, W, 0.030, seed=1)
acc = accel.Accel(max_events=len(sl["t"]), max_rows=s * H + s, max_cols=s * W + s)
opts = acc.default_opts(); opts.res_x, opts.res_y = H, W
for kv in os.environ.get('BF_OPTS', '').split(','):
    if kv: acc.set_option(kv.split('=')[0], int(kv.split('=')[1]))
opts.max_iter = int(os.environ.get('MI','40'))
acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"]); acc.set_cloud(s, H, W)
rc, m, info = acc.run(opts)
acc.close()
mx = {}
for ln in open("/tmp/bf_tl.txt"):
    kern, L, g, slot, t = [int(x) for x in ln.split()]
    if kern == 2:
        idx = L * 32 + g * 16 + slot
        if idx >= 1024: mx[idx - 1024] = t
        else: print("nonzero count left", idx, t)
h = collections.Counter(mx.values())
print("CUs seen", len(mx), "max resident work-groups per CU: histogram", sorted(h.items()))
