import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, "tests")
import numpy as np
import oracle
from better_flow_amd import accel, synth
H, W = 260, 346
a = synth.make_slice(100000, H, W, 0.030, seed=11)
b = synth.make_slice(100000, H, W, 0.030, seed=12)
oc = oracle.Cloud(a["fr_x"], a["fr_y"], a["t"]); ow = oc.set_cloud(3, H, W); om = oracle.Model()
rc, ol, otr = oc.run(ow, om, res_x=H, res_y=W, trace_cap=400)
acc = accel.Accel(max_events=100000, max_rows=3*H+3, max_cols=3*W+3)
acc.upload_events(a["fr_x"], a["fr_y"], a["t"]); acc.set_cloud(3, H, W)
opts = acc.default_opts(); opts.res_x, opts.res_y, opts.trace_cap = H, W, 400
_, gm, ia = acc.run(opts)
gtr = acc.get_trace(400)
print("cold iters", ia.iterations, ol.itercount)
for k in (0, 1, 2, 10, 50, min(ia.iterations, ol.itercount) - 1):
    g, o = gtr[k].model, otr[k].model
    print(k, g.cnt, o.cnt, g.total_dx - o.total_dx, g.total_dy - o.total_dy, g.total_rot - o.total_rot, g.total_div - o.total_div)
print("final model diff", {k: gm.as_dict()[k] - om.as_dict()[k] for k in gm.as_dict()})
oc2 = oracle.Cloud(b["fr_x"], b["fr_y"], b["t"]); ow2 = oc2.set_cloud(3, H, W)
# warm both from the ORACLE's model so the inputs are identical
gm_in = accel.Model(**{k: v for k, v in om.as_dict().items()})
om2 = oc2.set_model(om)
rc, ol2, otr2 = oc2.run(ow2, om2, res_x=H, res_y=W, trace_cap=100)
acc.upload_events(b["fr_x"], b["fr_y"], b["t"]); acc.set_cloud(3, H, W); acc.set_model(gm_in)
_, gm2, ib = acc.run(opts)
gtr2 = acc.get_trace(100)
print("warm iters", ib.iterations, ol2.itercount)
for k in range(max(ib.iterations, ol2.itercount)):
    if k < len(gtr2): g = gtr2[k]; print("G", k, g.model.cnt, g.model.dx, g.model.dy, g.model.rot, g.model.div, g.x_divider, g.y_divider, g.rot_divider, g.div_divider)
    if k < len(otr2): o = otr2[k]; print("O", k, o.model.cnt, o.model.dx, o.model.dy, o.model.rot, o.model.div, o.loop.x_divider, o.loop.y_divider, o.loop.rot_divider, o.loop.div_divider)
