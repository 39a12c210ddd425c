"""Generates tests/golden/*.npz + manifest.json from the CPU oracle.

These vectors are produced by the repo's own oracle (oracle/bf_oracle.c), NOT by the
reference: the reference has no tests / fixtures and cannot be built in this image (needs
OpenCV + TBB), so parity stays "unpinned".  The vectors pin the oracle against regressions
and give the GPU tests a box-independent target.  Run: python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402
from better_flow_amd import synth  # noqa: E402

H, W, S = 90, 120, 3
WARPS = [
    [0.0, 0.0, 0.0, 0.0, 0.0, 0.0],
    [0.3, -0.6, 0.0, 0.0, 0.0, 0.0],
    [0.2, 0.4, 44.5, 61.25, 3.0e-4, -2.0e-4],
    [-0.15, 0.05, 45.0, 59.0, -1.0e-3, 1.5e-3],
]


LOCAL_N = [[0.0, 0.0], [0.19, -0.38], [-0.4, 0.7]]
LOCAL_CENTER, LOCAL_WSZ = (40, 70, 20000000), 30


def main():
    sl = synth.make_slice(6000, H, W, 0.05, seed=21)
    c = oracle.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
    w = c.set_cloud(S, H, W)
    arrays = {"fr_x": sl["fr_x"], "fr_y": sl["fr_y"], "t": sl["t"]}
    for k, prm in enumerate(WARPS):
        c.project_4param_reinit(*prm)
        timg, cimg = c.get_time_img(w)
        arrays["pr_x_%d" % k] = c.pr_x.copy()
        arrays["pr_y_%d" % k] = c.pr_y.copy()
        arrays["nx_%d" % k] = c.nx.copy()
        arrays["ny_%d" % k] = c.ny.copy()
        arrays["cnt_%d" % k] = cimg.astype(np.uint16)
        arrays["time_%d" % k] = timg
    gx, gy = oracle.sobel(timg)
    arrays["gx"], arrays["gy"] = gx, gy
    m = oracle.fast_model(timg)
    c2 = oracle.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
    w2 = c2.set_cloud(S, H, W)
    m2 = oracle.Model()
    rc, loop, tr = c2.run(w2, m2, res_x=H, res_y=W, trace_cap=4096)
    arrays["trajectory"] = np.array(
        [[r.model.total_dx, r.model.total_dy, r.model.total_rot, r.model.total_div,
          r.loop.x_divider, r.loop.y_divider, r.loop.rot_divider, r.loop.div_divider] for r in tr])
    u, v = c2.compute_uv()
    arrays["u"], arrays["v"] = u, v
    np.savez_compressed(os.path.join(HERE, "slice_6k_120x90.npz"), **arrays)
    man = {
        "file": "slice_6k_120x90.npz", "generator": "tests/golden/make_golden.py (CPU oracle)",
        "height": H, "width": W, "scale": S, "seed": 21, "warps": WARPS,
        "window": [w.x_min, w.x_max, w.y_min, w.y_max, w.scale_img_x, w.scale_img_y],
        "model": [m.cx, m.cy, m.dx, m.dy, m.rot, m.div, m.cnt],
        "iterations": int(loop.itercount), "rc": rc,
        "final_model": m2.as_dict(),
    }
    # ---- second fixture: contrast-score path (OptimizerLocal) and EventFile::projection_img, same slice ----
    c3 = oracle.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
    c3.set_cloud(S, H, W)
    extra, local = {}, {"candidates": LOCAL_N, "scores_cloud": [], "scores_window": []}
    lw = c3.local_window(S)
    lw2 = c3.local_window(S, center=LOCAL_CENTER, wsz=LOCAL_WSZ)
    for k, (nx, ny) in enumerate(LOCAL_N):
        sc, img = c3.local_iteration_step(lw, nx, ny)
        extra["local_cloud_%d" % k] = img
        local["scores_cloud"].append(sc)
        sc2, img2 = c3.local_iteration_step(lw2, nx, ny)
        extra["local_window_%d" % k] = img2
        local["scores_window"].append(sc2)
    rcl, st, _ = c3.local_run(lw, res_x=H, res_y=W)
    local["run"] = {"rc": rcl, "nx": st.nx, "ny": st.ny, "last_score": st.last_score, "evaluations": int(st.evaluations)}
    local["center"], local["wsz"] = list(LOCAL_CENTER), LOCAL_WSZ
    c4 = oracle.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
    c4.set_cloud(S, H, W)
    extra["proj_raw"] = c4.projection_img(S, H, W, show_final=True)
    c4.project_4param_reinit(*WARPS[2])
    extra["proj_warp2"] = c4.projection_img(S, H, W)
    np.savez_compressed(os.path.join(HERE, "slice_6k_120x90_images.npz"), **extra)
    man["images_file"] = "slice_6k_120x90_images.npz"
    man["local"] = local
    json.dump(man, open(os.path.join(HERE, "manifest.json"), "w"), indent=1)
    print("wrote", man["file"], man["images_file"], "iterations", loop.itercount, "events", len(sl["t"]))


if __name__ == "__main__":
    main()
