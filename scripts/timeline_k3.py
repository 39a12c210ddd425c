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
if os.environ.get('BF_CO'): acc.set_option('co_schedule', int(os.environ['BF_CO']))
for kv in os.environ.get('BF_OPTS', '').split(','):
    if kv: acc.set_option(kv.split('=')[0], int(kv.split('=')[1]))
opts.max_iter = int(os.environ.get("BF_RUN_MAXITER", "40"))
acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"]); acc.set_cloud(s, H, W)
rc, m, info = acc.run(opts)
acc.close()
d = collections.defaultdict(dict)
for ln in open("/tmp/bf_tl.txt"):
    kern, L, g, slot, t = [int(x) for x in ln.split()]
    d[(kern, L, g)][slot] = t
names = {0: "entry", 1: "state", 2: "slabs loaded+LDS", 3: "sync", 4: "time img+sync", 5: "tail pixels", 6: "wave reduce+sync", 7: "published+ticket",
         10: "LAST:partials loaded", 11: "LAST:reduced", 12: "LAST:update done"}
k1names = {0: "entry", 5: "loads issued", 6: "totals", 7: "updated", 1: "barrier", 2: "events done", 3: "sync", 4: "flushed"}
L0 = int(os.environ.get("TL_LAUNCH", "20"))
for L in (L0, L0 + 1):
    base = min(d[(1, L, 0)].values()) if d.get((1, L, 0)) else (min(d[(0, L, 0)].values()) if d.get((0, L, 0)) else None)
    for g in (0, 1):
        st = d.get((1, L, g), {})
        if st and base:
            print("K1b launch", L, "group", "0" if g == 0 else "mid", " | ".join("%s=%.2f" % (k1names.get(k, k), (st[k] - base) / 100.0) for k in sorted(st, key=lambda k_: st[k_])))
    for g in (0, 1):
        st = d.get((0, L, g), {})
        if st and base:
            print("K3  launch", L, "group", "0" if g == 0 else "mid", " | ".join("%s=%.2f" % (names.get(k, k), (st[k] - base) / 100.0) for k in sorted(st, key=lambda k_: st[k_])))
