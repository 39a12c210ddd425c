import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from better_flow_amd import accel, synth
N, H, W, s, G = 1000000, 260, 346, 3, 32
sl = synth.make_slice(N, H, W, 0.030, seed=1)
acc = accel.Accel(max_events=len(sl["t"]), max_rows=s*H+s, max_cols=s*W+s)
acc.upload_events(sl["fr_x"], sl["fr_y"], sl["t"])
models, infos = acc.run_tiles(G, G, s, (H, W), (H//G, W//G), min_events=256, hard_iter_cap=20000)
it = np.array([i.iterations for i in infos])
tr = np.minimum(sl["fr_x"].astype(np.int64) * G // H, G - 1); tc = np.minimum(sl["fr_y"].astype(np.int64) * G // W, G - 1)
cnt = np.bincount(tr * G + tc, minlength=G*G)
o = np.argsort(-it)[:8]
print("top iterations:", [(int(it[k]), int(cnt[k])) for k in o], "max events per tile", cnt.max())
