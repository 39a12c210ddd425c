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
    # ---- third fixture: a warm-started (STM) run, the colour-coded time images, the command line ----
    stream = {}
    sl_b = synth.make_slice(6000, H, W, 0.05, seed=22)           # the next slice of the same scene
    c5 = oracle.Cloud(sl_b["fr_x"], sl_b["fr_y"], sl_b["t"])
    w5 = c5.set_cloud(S, H, W)
    m5 = c5.set_model(m2)                                         # OptimizerRolling::set_model(last slice's model)
    rc5, loop5, tr5 = c5.run(w5, m5, res_x=H, res_y=W, trace_cap=4096)
    stream["b_fr_x"], stream["b_fr_y"], stream["b_t"] = sl_b["fr_x"], sl_b["fr_y"], sl_b["t"]
    stream["warm_trajectory"] = np.array(
        [[r.model.total_dx, r.model.total_dy, r.model.total_rot, r.model.total_div,
          r.loop.x_divider, r.loop.y_divider, r.loop.rot_divider, r.loop.div_divider] for r in tr5])
    c6 = oracle.Cloud(sl["fr_x"], sl["fr_y"], sl["t"])
    c6.set_cloud(S, H, W)
    stream["color_raw"] = c6.color_time_img(S, H, W, show_final=True)
    c6.project_4param_reinit(*WARPS[2])
    stream["color_warp2"] = c6.color_time_img(S, H, W)
    # bf_motion_compensator -o on the config-1 input (10k events, 240x180), through the oracle-backed build of the
    # host front end (tests/shim): slice count, skip decisions, and the per-event output file
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "shim"))
    import build as shim_build
    exe = shim_build.build()
    with tempfile.TemporaryDirectory() as d:
        sl_c = synth.make_slice(10000, 180, 240, 0.1, seed=5)
        synth.write_txt(os.path.join(d, "ev.txt"), sl_c)
        so = subprocess.run([exe, "-o", os.path.join(d, "out.txt"), os.path.join(d, "ev.txt")], cwd=d,
                            stdout=subprocess.PIPE, check=True).stdout.decode()
        out = np.loadtxt(os.path.join(d, "out.txt"))
    import re
    summ = [int(x) for x in re.search(r"slices: (\d+) \(skipped (\d+)\), minimizer iterations: (\d+)", so).groups()]
    stream["cli_t"], stream["cli_x"], stream["cli_y"] = out[:, 0], out[:, 1].astype(np.int32), out[:, 2].astype(np.int32)
    stream["cli_v"], stream["cli_u"] = out[:, 4], out[:, 5]
    np.savez_compressed(os.path.join(HERE, "slice_6k_120x90_stream.npz"), **stream)
    man["stream"] = {"file": "slice_6k_120x90_stream.npz", "warm_rc": rc5, "warm_iterations": int(loop5.itercount),
                     "warm_final_model": m5.as_dict(), "cli": {"events": 10000, "height": 180, "width": 240, "seed": 5,
                                                                "slices": summ[0], "skipped": summ[1], "iterations": summ[2]}}
    json.dump(man, open(os.path.join(HERE, "manifest.json"), "w"), indent=1)
    print("wrote", man["file"], man["images_file"], "iterations", loop.itercount, "events", len(sl["t"]))


if __name__ == "__main__":
    main()
