"""Generator of tests/golden/config5_720p_seed<S>.npz -- BASELINE config 5's slice run to the loop's OWN termination.

    python tests/golden/make_config5_golden.py [seed ...]                 (default: seed 1; ~25 min of CPU per seed)
    python tests/golden/make_config5_golden.py --ensemble K [seed ...]    (K more event orders per seed, see ensemble())

The oracle (oracle/bf_oracle.c, the CPU restatement of optimizer_rolling.h:48-125 -- "parity unpinned", see its header)
is run twice on the 1M-event 1280x720 slice `synth.make_slice(1000000, 720, 1280, 0.030, seed)`: events in upload order
and in reversed order.  The reference accumulates its time image in f32 in container order (accel_lib.h:162), so the two
runs are both "the reference's answer"; their difference is the yardstick the GPU run is held to
(tests/test_gpu_geometries.py::test_config5_to_termination_against_golden).  Thousands of iterations at 0.27 s each do not
fit the GPU suite's budget, hence a committed fixture: per order the return code, iteration count, dividers, final model,
the model every 25 iterations, per-event flow at 4096 evenly spaced events and its percentiles, plus the largest
forward / reversed flow difference over ALL events, and a digest of the input arrays.
Oracle-generated: a regression pin of the restatement, not a pin to the reference.
"""
import hashlib
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

H, W, S, N, T = 720, 1280, 3, 1000000, 0.030
STRIDE = 25
SAMPLES = 4096
FIELDS = ("cx", "cy", "dx", "dy", "rot", "div", "cnt", "total_dx", "total_dy", "total_rot", "total_div")
PCT = (0, 1, 5, 25, 50, 75, 95, 99, 100)


def slice_digest(sl):
    h = hashlib.sha256()
    for k in ("fr_x", "fr_y", "t"):
        h.update(np.ascontiguousarray(sl[k]).tobytes())
    return h.hexdigest()


def event_order(n, member):
    """Member 0: upload order; 1: reversed; k >= 2: the permutation numpy's default_rng(k) draws."""
    if member == 0:
        return np.arange(n)
    if member == 1:
        return np.arange(n)[::-1].copy()
    return np.random.default_rng(member).permutation(n)


def one_run(args):
    seed, member = args
    import oracle
    from better_flow_amd import synth
    sl = synth.make_slice(N, H, W, T, seed=seed)
    n = len(sl["t"])
    order = event_order(n, member)
    oc = oracle.Cloud(sl["fr_x"][order], sl["fr_y"][order], sl["t"][order])
    ow = oc.set_cloud(S, H, W)
    om = oracle.Model()
    cap = 60000
    rc, loop, trace = oc.run(ow, om, res_x=H, res_y=W, hard_cap=cap, trace_cap=cap)
    u, v = oc.compute_uv()
    inv = np.empty(n, np.int64)
    inv[order] = np.arange(n)
    u, v = u[inv], v[inv]
    every = np.array([[getattr(trace[k].model, f) for f in FIELDS] for k in range(0, len(trace), STRIDE)], dtype=np.float64)
    return dict(rc=rc, iterations=int(loop.itercount),
                dividers=np.array([loop.x_divider, loop.y_divider, loop.rot_divider, loop.div_divider], np.float32),
                model=np.array([getattr(om, f) for f in FIELDS], np.float64), every=every, u=u, v=v,
                digest=slice_digest(sl), n=n)


def assemble(outdir, seeds, members):
    """python make_config5_golden.py --assemble OUTDIR K seed ...: the fixtures of main() and ensemble(K) from the per-run files
    tests/golden/config5_job.py left in OUTDIR (the same runs, made one by one so that they can be resumed)."""
    def load(seed, member):
        d = np.load(os.path.join(outdir, "job_%d_%d.npz" % (seed, member)))
        return {k: (d[k] if d[k].ndim else d[k].item()) for k in d.files}
    for seed in seeds:
        f, r = load(seed, 0), load(seed, 1)
        assert f["digest"] == r["digest"]
        idx = np.linspace(0, f["n"] - 1, SAMPLES).astype(np.int64)
        out = dict(seed=seed, geometry=np.array([H, W, S]), n=f["n"], input_sha256=f["digest"], fields=np.array(FIELDS),
                   trace_stride=STRIDE, sample_idx=idx, percentiles=np.array(PCT),
                   spread_u=np.abs(f["u"] - r["u"]).max(), spread_v=np.abs(f["v"] - r["v"]).max())
        for tag, d in (("fwd", f), ("rev", r)):
            out[tag + "_rc"] = d["rc"]
            out[tag + "_iterations"] = d["iterations"]
            out[tag + "_dividers"] = d["dividers"]
            out[tag + "_model"] = d["model"]
            out[tag + "_every"] = d["every"]
            out[tag + "_u"] = d["u"][idx]
            out[tag + "_v"] = d["v"][idx]
            out[tag + "_u_pct"] = np.percentile(d["u"], PCT)
            out[tag + "_v_pct"] = np.percentile(d["v"], PCT)
        np.savez_compressed(os.path.join(HERE, "config5_720p_seed%d.npz" % seed), **out)
        rs = [load(seed, m) for m in range(2, 2 + members)]
        eo = dict(seed=seed, members=np.arange(2, 2 + members), input_sha256=rs[0]["digest"], sample_idx=idx, fields=np.array(FIELDS),
                  percentiles=np.array(PCT), rc=np.array([q["rc"] for q in rs]), iterations=np.array([q["iterations"] for q in rs]),
                  dividers=np.array([q["dividers"] for q in rs]), model=np.array([q["model"] for q in rs]),
                  u=np.array([q["u"][idx] for q in rs]), v=np.array([q["v"][idx] for q in rs]),
                  u_pct=np.array([np.percentile(q["u"], PCT) for q in rs]), v_pct=np.array([np.percentile(q["v"], PCT) for q in rs]))
        np.savez_compressed(os.path.join(HERE, "config5_720p_seed%d_ensemble.npz" % seed), **eo)
        allu = np.array([f["u"], r["u"]] + [q["u"] for q in rs]); allv = np.array([f["v"], r["v"]] + [q["v"] for q in rs])
        print("seed %d: iterations fwd %d rev %d ensemble %s; width of the six orders' flow: %.3f / %.3f px/s" %
              (seed, f["iterations"], r["iterations"], eo["iterations"].tolist(), (allu.max(0) - allu.min(0)).max(), (allv.max(0) - allv.min(0)).max()))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--assemble":
        return assemble(sys.argv[2], [int(a) for a in sys.argv[4:]], int(sys.argv[3]))
    if len(sys.argv) > 1 and sys.argv[1] == "--ensemble":
        return ensemble(int(sys.argv[2]), [int(a) for a in sys.argv[3:]] or [1])
    seeds = [int(a) for a in sys.argv[1:]] or [1]
    jobs = [(s, r) for s in seeds for r in (0, 1)]
    with mp.get_context("spawn").Pool(len(jobs)) as pool:
        res = pool.map(one_run, jobs)
    for i, seed in enumerate(seeds):
        f, r = res[2 * i], res[2 * i + 1]
        assert f["digest"] == r["digest"]
        idx = np.linspace(0, f["n"] - 1, SAMPLES).astype(np.int64)
        out = dict(seed=seed, geometry=np.array([H, W, S]), n=f["n"], input_sha256=f["digest"], fields=np.array(FIELDS),
                   trace_stride=STRIDE, sample_idx=idx, percentiles=np.array(PCT),
                   spread_u=np.abs(f["u"] - r["u"]).max(), spread_v=np.abs(f["v"] - r["v"]).max())
        for tag, d in (("fwd", f), ("rev", r)):
            out[tag + "_rc"] = d["rc"]
            out[tag + "_iterations"] = d["iterations"]
            out[tag + "_dividers"] = d["dividers"]
            out[tag + "_model"] = d["model"]
            out[tag + "_every"] = d["every"]
            out[tag + "_u"] = d["u"][idx]
            out[tag + "_v"] = d["v"][idx]
            out[tag + "_u_pct"] = np.percentile(d["u"], PCT)
            out[tag + "_v_pct"] = np.percentile(d["v"], PCT)
        path = os.path.join(HERE, "config5_720p_seed%d.npz" % seed)
        np.savez_compressed(path, **out)
        print("%s: forward %d iterations (rc %d), reversed %d (rc %d); flow spread %.3e / %.3e px/s" %
              (path, f["iterations"], f["rc"], r["iterations"], r["rc"], out["spread_u"], out["spread_v"]))


def ensemble(members, seeds):
    """python make_config5_golden.py --ensemble K [seed ...]: K more members per seed -- the oracle on K seeded random
    permutations of the slice's events (event_order members 2 .. K + 1) -- into config5_720p_seed<S>_ensemble.npz.
    Forward and reversed stay together on seed 0 (7355 / 7344 iterations) while a third order of the same events can
    take another branch of the loop's divider doublings (optimizer_rolling.h:98-101: a divider doubles whenever its
    gradient component changes sign between two iterations -- near the optimum that is decided by the last bits);
    the ensemble shows how wide the reference's own answer is, and is the yardstick of the GPU test."""
    jobs = [(s, m) for s in seeds for m in range(2, 2 + members)]
    with mp.get_context("spawn").Pool(min(len(jobs), os.cpu_count() or 1)) as pool:
        res = pool.map(one_run, jobs, chunksize=1)
    for i, seed in enumerate(seeds):
        rs = res[i * members:(i + 1) * members]
        idx = np.linspace(0, rs[0]["n"] - 1, SAMPLES).astype(np.int64)
        out = dict(seed=seed, members=np.arange(2, 2 + members), input_sha256=rs[0]["digest"], sample_idx=idx, fields=np.array(FIELDS),
                   percentiles=np.array(PCT), rc=np.array([r["rc"] for r in rs]), iterations=np.array([r["iterations"] for r in rs]),
                   dividers=np.array([r["dividers"] for r in rs]), model=np.array([r["model"] for r in rs]),
                   u=np.array([r["u"][idx] for r in rs]), v=np.array([r["v"][idx] for r in rs]),
                   u_pct=np.array([np.percentile(r["u"], PCT) for r in rs]), v_pct=np.array([np.percentile(r["v"], PCT) for r in rs]))
        path = os.path.join(HERE, "config5_720p_seed%d_ensemble.npz" % seed)
        np.savez_compressed(path, **out)
        print("%s: iterations %s" % (path, out["iterations"].tolist()))


if __name__ == "__main__":
    main()
