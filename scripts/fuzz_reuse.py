"""Context-reuse fuzzer: one long-lived bf_ctx runs a random sequence of operations over many slices (uploads of
different sizes / sensors, windows at different scales, warps, images, cold and warm runs with different scatter
modes, contrast-score evaluations, projection images, per-event read-backs).  After every observable operation the
operations since the last upload are replayed on a FRESH context; both must give the same bits.  Catches stale state
carried from one slice / operation to the next (plane buffers, cached flags, lazily allocated buffers).
usage: fuzz_reuse.py [steps] [seed]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from better_flow_amd import accel

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
HMAX, WMAX, SMAX, NMAX = 200, 260, 7, 60000


def new_ctx():
    return accel.Accel(max_events=NMAX, max_rows=SMAX * HMAX + SMAX, max_cols=SMAX * WMAX + SMAX)


def make_slice():
    H, W = int(rng.integers(40, HMAX)), int(rng.integers(40, WMAX))
    n = int(rng.choice([200, 3000, 20000, 55000]))
    npts = max(1, n // 16)
    pr, pc = rng.uniform(0, H, npts), rng.uniform(0, W, npts)
    v = rng.normal(0, 150, 2)
    t = np.sort(rng.uniform(0, 0.03, n))
    pick = rng.integers(0, npts, n)
    row, col = pr[pick] + v[0] * t, pc[pick] + v[1] * t
    keep = (row >= 0) & (row < H) & (col >= 0) & (col < W)
    return dict(H=H, W=W, fr_x=np.floor(row[keep]).astype(np.int32), fr_y=np.floor(col[keep]).astype(np.int32),
                t=(t[keep] * 1e9).astype(np.int64))


def canon(m):
    d = m.as_dict()
    return b"".join(np.float64(d[k]).tobytes() for k in sorted(d))


def apply(a, op, sl):
    """Executes one op on ctx a; returns an observable (bytes-comparable tuple) or None."""
    k = op[0]
    if k == "opt":
        a.set_option(op[1], op[2])
        if a is long_ctx:
            long_opts[op[1]] = op[2]
        return None
    if k == "upload":
        how = op[2] if (a is long_ctx or mimic_long) else "plain"     # the long-lived context also exercises the other hand-overs
        n = len(sl["t"])
        if how == "ring16" and n > 0:   # 16-bit addresses, absolute timestamps, the noise flags as a ring of their own
            cap, first, t0 = n + int(op[3] % 977), int(op[3] % (n + 1)), 5000000000
            idx = (first + np.arange(n)) % cap
            rr, rc_, rts = np.zeros(cap, np.uint16), np.zeros(cap, np.uint16), np.zeros(cap, np.uint64)
            rr[idx], rc_[idx], rts[idx] = sl["fr_x"], sl["fr_y"], (sl["t"] + t0).astype(np.uint64)
            rn = None
            if op[1] is not None:
                rn = np.ones(cap, np.uint8)          # (flags outside the slice must not matter)
                rn[idx] = op[1]
            a.upload_ring_async(rr, rc_, rts, first, n, t0, rn)
            a.commit_upload()
            a.synchronize()
        elif how == "plain" or op[1] is not None or n == 0:
            a.upload_events(sl["fr_x"], sl["fr_y"], sl["t"], op[1])
        elif how == "async":
            px, py, pt = a.pinned_int32(n), a.pinned_int32(n), a.pinned_int32(n)
            px[:], py[:], pt[:] = sl["fr_x"], sl["fr_y"], sl["t"]
            a.upload_events_async(px, py, pt, n)
            a.commit_upload()
            a.synchronize()
        else:   # "ring": absolute 64-bit timestamps in a ring that wraps, local time on the device
            cap, first, t0 = n + int(op[3] % 977), int(op[3] % (n + 1)), 5000000000
            idx = (first + np.arange(n)) % cap
            rx, ry, rts = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.uint64)
            rx[idx], ry[idx], rts[idx] = sl["fr_x"], sl["fr_y"], (sl["t"] + t0).astype(np.uint64)
            a.upload_ring_async(rx, ry, rts, first, n, t0)
            a.commit_upload()
            a.synchronize()
        return None
    if k == "cloud":
        w = a.set_cloud(op[1], sl["H"], sl["W"]); return (w.scale_img_x, w.scale_img_y, w.x_shift, w.y_shift)
    if k == "model":
        a.set_model(accel.Model(**op[1])); return None
    if k == "project":
        a.project_4param_reinit(*op[1]); return None
    if k == "img":
        t_, c_ = a.get_time_img(); return (t_.tobytes(), c_.tobytes())
    if k == "run":
        o = a.default_opts()
        o.res_x, o.res_y, o.max_iter, o.trace_cap, o.min_events, o.want_uv = sl["H"], sl["W"], op[1], 16, 50, op[2]
        rc, m, info = a.run(o)
        return (rc, info.iterations, canon(m), tuple(canon(t_.model) for t_ in a.get_trace(16)))
    if k == "uv":
        u, v = a.compute_uv()
        if a is long_ctx and len(u):   # the ring form of the same read-back must agree, at a random ring position
            n = len(u)
            cap, first = n + int(rng.integers(0, 50)), int(rng.integers(0, n))
            ring = np.full(2 * cap, np.nan)
            a.compute_uv_ring(ring, first)
            idx = (first + np.arange(n)) % cap
            assert np.array_equal(ring[2 * idx], u) and np.array_equal(ring[2 * idx + 1], v), "bf_compute_uv_ring != bf_compute_uv"
        return (u.tobytes(), v.tobytes())
    if k == "writeout":
        return tuple(x.tobytes() for x in a.writeout_events())
    if k == "lwin":
        w = a.local_set_window(op[1], center=op[2], wsz=op[3]); return (w.scale_img_x, w.scale_img_y, w.c_fr_x, w.c_fr_y)
    if k == "lstep":
        sc, img = a.local_iteration_step(op[1], op[2], want_img=True); return (np.float64(sc).tobytes(), img.tobytes())
    if k == "proj":
        return (a.projection_img(op[1], sl["H"], sl["W"], show_final=op[2]).tobytes(),)
    if k == "color":
        return (a.color_time_img(op[1], sl["H"], sl["W"], show_final=op[2]).tobytes(),)
    if k == "tiles":
        models, infos = a.run_tiles(op[1], op[1], 3, (sl["H"], sl["W"]), (sl["H"] // op[1], sl["W"] // op[1]), 64, max_iter=op[2])
        return tuple(canon(m) for m in models) + tuple((i.rc, i.iterations) for i in infos)
    raise ValueError(k)


long_ctx = new_ctx()
mimic_long = False
history = []        # (options in force at the upload, slice, ops) per slice of the long-lived context: bisection of a mismatch
long_opts = {}      # options in force on the long-lived context (a replay only sees those set since the upload)
script, sl, bad, checks = [], None, 0, 0
state = dict(cloud=False, lwin=None, ran=False, scale=3)
last_model = None
for step in range(steps):
    if sl is None or rng.random() < 0.12:
        sl = make_slice()
        noise = (rng.random(len(sl["t"])) < 0.1).astype(np.uint8) if rng.random() < 0.2 else None
        script = [("upload", noise, str(rng.choice(["plain", "async", "ring", "ring16"])), int(rng.integers(0, 1 << 30)))]
        apply(long_ctx, script[0], sl)
        history.append((dict(long_opts), sl, script))
        state = dict(cloud=False, lwin=None, ran=False, scale=3, noise=noise is not None)
        continue
    choices = ["cloud", "opt"]
    if state["cloud"]:
        choices += ["project", "img", "run", "run", "model", "uv", "writeout", "proj", "tiles"]
    choices += ["lwin", "proj", "color"]
    if state["lwin"]:
        choices += ["lstep", "lstep"]
    k = str(rng.choice(choices))
    if k == "cloud":
        state["scale"] = int(rng.choice([1, 3, 3, 5, 7]))
        op = ("cloud", state["scale"]); state["cloud"] = True
    elif k == "opt":
        name = str(rng.choice(["binned", "co_schedule", "co_schedule", "bin_predict", "bin_pack_limit", "bin_compact", "bin_split", "fused", "persist", "sep_update"]))
        val = {"persist": int(rng.integers(0, 3)), "fused": int(rng.integers(0, 3)), "binned": int(rng.integers(0, 3)),
               "co_schedule": int(rng.integers(0, 2)), "bin_compact": int(rng.integers(0, 3)), "bin_split": int(rng.integers(0, 3)), "bin_predict": int(rng.integers(0, 2)),
               "bin_pack_limit": int(rng.choice([64, 64, 30, 1])), "sep_update": int(rng.integers(0, 3))}[name]
        op = ("opt", name, val)
    elif k == "project":
        op = ("project", (rng.normal(0, .3), rng.normal(0, .3), rng.uniform(0, 100), rng.uniform(0, 100), rng.normal(0, 1e-4), rng.normal(0, 3e-5)))
    elif k == "run":
        op = ("run", int(rng.choice([-1, 3, 10, 25])), int(rng.integers(0, 2))); state["ran"] = True
    elif k == "model":
        if last_model is None:
            continue
        op = ("model", last_model)
    elif k == "lwin":
        s_ = int(rng.choice([1, 3, 5, 7]))
        if rng.random() < 0.5:
            op = ("lwin", s_, None, 0)
        else:
            op = ("lwin", s_, (int(rng.integers(0, sl["H"])), int(rng.integers(0, sl["W"])), int(rng.integers(0, 3e7))), int(rng.integers(4, 30)))
        state["lwin"] = True
    elif k == "lstep":
        op = ("lstep", float(rng.normal(0, .4)), float(rng.normal(0, .4)))
    elif k == "proj":
        op = ("proj", int(rng.choice([1, 3, 5])), bool(rng.integers(0, 2)))
    elif k == "color":
        op = ("color", int(rng.choice([1, 2, 3])), bool(rng.integers(0, 2)))
    elif k == "tiles":
        op = ("tiles", int(rng.choice([2, 4])), int(rng.choice([5, 20])))
    else:
        op = (k,)
    script.append(op)
    try:
        got = apply(long_ctx, op, sl)
    except accel.BfError as e:
        got = ("error", e.code)   # a legal refusal (e.g. an image operator on a degenerate window)
    if got is None:
        continue
    fresh = new_ctx()
    want = None
    try:
        for o_ in script:
            want = apply(fresh, o_, sl)
    except accel.BfError as e:
        want = ("error", e.code) if o_ is script[-1] else ("error in replay at", o_[0], e.code)
    fresh.close()
    checks += 1
    if got != want:
        bad += 1
        print("step %d: MISMATCH after %s; ops since upload (%d): %s" % (step, op[0], len(script), [(o_[0],) + tuple(o_[1:3]) if o_[0] in ("opt", "cloud", "run", "proj", "color", "lwin") else o_[0] for o_ in script]))
        if isinstance(got, tuple) and isinstance(want, tuple) and len(got) == len(want):
            print("    differing fields:", [i for i in range(len(got)) if got[i] != want[i]], "of", len(got))
        else:
            print("    got", got if len(str(got)) < 200 else "...", "want", want if len(str(want)) < 200 else "...")
        # is the script itself deterministic?  replay it on two more fresh contexts
        reps = []
        for _ in range(2):
            f2 = new_ctx()
            try:
                for o_ in script:
                    w2 = apply(f2, o_, sl)
            except accel.BfError as e:
                w2 = ("error", e.code)
            f2.close()
            reps.append(w2)
        # ... and with the long-lived context's options in force from the start?
        f3 = new_ctx()
        try:
            for k_, v_ in long_opts.items():
                f3.set_option(k_, v_)
            for o_ in script:
                w3 = apply(f3, o_, sl)
        except accel.BfError as e:
            w3 = ("error", e.code)
        f3.close()
        print("    long-lived options", long_opts, "| fresh context with them equal to long-lived:", w3 == got)
        # how much of the long-lived context's past does it take?  replay its last k slices on a fresh context, then
        # drop slices and operations one by one while the replay still gives the long-lived context's result
        mimic_long = True

        def replay(opts0, hist):
            f4 = new_ctx()
            w4 = None
            try:
                for k_, v_ in opts0.items():
                    f4.set_option(k_, v_)
                for (sl_h, ops_h) in hist:
                    for o_ in ops_h:
                        try:
                            w4 = apply(f4, o_, sl_h)
                        except accel.BfError as e:
                            w4 = ("error", e.code)
                        except (AttributeError, TypeError, ValueError):
                            return None      # the trial removed something a later operation needs: not a reproduction
            finally:
                f4.close()
            return w4

        for k_back in (1, 2, 3, 5, 8, 13, 21, 34, len(history)):
            k_back = min(k_back, len(history))
            opts0 = history[-k_back][0]
            hist = [(h_[1], list(h_[2])) for h_ in history[-k_back:]]
            if replay(opts0, hist) == got:
                print("    replay of the last %d slice(s) reproduces the long-lived result; reducing" % k_back)
                changed = True
                while changed:
                    changed = False
                    for i in range(len(hist) - 1):          # whole slices
                        trial = hist[:i] + hist[i + 1:]
                        if replay(opts0, trial) == got:
                            hist, changed = trial, True
                            break
                    if changed:
                        continue
                    for i in range(len(hist)):              # single operations (never an upload, never the last one)
                        for q in range(1, len(hist[i][1]) - (1 if i == len(hist) - 1 else 0)):
                            trial = [(h_[0], list(h_[1])) for h_ in hist]
                            del trial[i][1][q]
                            if replay(opts0, trial) == got:
                                hist, changed = trial, True
                                break
                        if changed:
                            break
                    if changed:
                        continue
                    for k_ in list(opts0):                  # initial options
                        trial_o = {a_: b_ for a_, b_ in opts0.items() if a_ != k_}
                        if replay(trial_o, hist) == got:
                            opts0, changed = trial_o, True
                            break
                print("    reduced: options at the start", opts0)
                try:      # the slices of the reduced case, for a stand-alone reproduction
                    od = os.path.join(ROOT, "gpurun_out"); os.makedirs(od, exist_ok=True)
                    np.savez(os.path.join(od, "fuzz_reuse_case_%d.npz" % step), ops=repr([h_[1] for h_ in hist]), opts=repr(opts0),
                             **{"s%d_%s" % (q_, k_): np.asarray(v_) for q_, h_ in enumerate(hist) for k_, v_ in h_[0].items()})
                except Exception as e_:
                    print("    (could not save the case:", e_, ")")
                for (sl_h, ops_h) in hist:
                    print("        slice %dx%d n=%d:" % (sl_h["H"], sl_h["W"], len(sl_h["t"])), [tuple(o_[:1]) + tuple(x for x in o_[1:4] if not isinstance(x, np.ndarray)) for o_ in ops_h])
                break
        mimic_long = False
        # ... and the other way round: which operations of the script does the FRESH context need to arrive at its result?
        try:
            keep = list(script)
            changed = True
            while changed:
                changed = False
                for q in range(1, len(keep) - 1):
                    trial = keep[:q] + keep[q + 1:]
                    if replay({}, [(sl, trial)]) == want:
                        keep, changed = trial, True
                        break
            print("    the fresh context's result needs:", [tuple(o_[:1]) + tuple(x for x in o_[1:4] if not isinstance(x, np.ndarray)) for o_ in keep])
        except Exception as e_:
            print("    (second reduction failed:", e_, ")")
        print("    fresh replays equal to each other:", reps[0] == reps[1], "| equal to first fresh:", reps[0] == want, "| equal to long-lived:", reps[0] == got)
        if bad >= 3:
            break
    if isinstance(got, tuple) and got and got[0] == "error":
        script.pop()               # a refused operation leaves no trace
    if op[0] == "run" and got[0] == 0:
        # remember a model for later warm starts (decoded from the canonical bytes of the long-lived run)
        keys = sorted(accel.Model().as_dict())
        vals = np.frombuffer(got[2], dtype=np.float64)
        last_model = {k_: (int(v_) if k_ == "cnt" else float(v_)) for k_, v_ in zip(keys, vals) if k_ != "_pad"}
    if step % 25 == 24:
        print("... %d steps, %d checks, %d mismatches" % (step + 1, checks, bad), flush=True)
long_ctx.close()
print("fuzz_reuse: %d steps, %d checks, %d mismatches" % (steps, checks, bad))
sys.exit(1 if bad else 0)
