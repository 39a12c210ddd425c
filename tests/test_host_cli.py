"""Host front end (better_flow_amd/host): slice manager, CLI, text formats.

CPU part: the CLI linked against the oracle-backed test shim (tests/shim) on BASELINE config 1
(10k-event synthetic .txt, 240x180).  GPU part: the product CLI (libbf_accel.so) on the same
file must make the same slice decisions and agree on the per-event flow."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from better_flow_amd import synth  # noqa: E402


@pytest.fixture(scope="module")
def oracle_cli():
    sys.path.insert(0, os.path.join(ROOT, "tests", "shim"))
    import build as shim_build
    return shim_build.build()


@pytest.fixture(scope="module")
def events_txt(tmp_path_factory):
    d = tmp_path_factory.mktemp("cli")
    sl = synth.make_slice(10000, 180, 240, 0.1, seed=5)
    path = str(d / "ev10k.txt")
    synth.write_txt(path, sl)
    return path, sl


def run_cli(exe, args, cwd):
    r = subprocess.run([exe] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r.stdout.decode()


def parse_summary(stdout):
    m = re.search(r"slices: (\d+) \(skipped (\d+)\), minimizer iterations: (\d+)", stdout)
    return tuple(int(x) for x in m.groups())


def test_ring_buffer_semantics(tmp_path):
    exe = str(tmp_path / "test_ring")
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-I" + os.path.join(ROOT, "better_flow_amd", "host"),
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_ring.cpp"),
                           "-o", exe])
    out = subprocess.check_output([exe]).decode().splitlines()
    assert out[0] == "fill5 size=5 iter=50,40,30,20,10 n_iter=5 newest=50 oldest=10"
    # newest = 200: 140 and 200 are within 100 ns; 90 is 110 away and is trimmed
    assert out[1] == "span100 size=2 iter=200,140 n_iter=2 newest=200 oldest=140"
    # full ring (datastructures.h:71-76): size 4, iteration stops one short
    assert out[2] == "full4 size=4 iter=6,5,4 n_iter=3 newest=6 oldest=3"
    assert out[3] == "eq 1 0 0"
    assert out[4] == "local 600 -99001"
    assert out[5] == "from_sec 33000000 200000000"


def test_cli_config1_oracle(oracle_cli, events_txt, tmp_path):
    path, sl = events_txt
    out_file = str(tmp_path / "out.txt")
    stdout = run_cli(oracle_cli, ["-o", out_file, path], str(tmp_path))
    slices, skipped, iters = parse_summary(stdout)
    assert slices == 4 and skipped == 0           # 3 triggered (33 ms) + the final flush
    a = np.loadtxt(out_file)
    n = len(sl["t"])
    assert a.shape == (n, 6)                      # every event once (overlap de-duplicated)
    # output columns: t x(col) y(row) 1 v(col flow) u(row flow)  (event_file.h:272-276)
    assert np.array_equal(a[:, 1].astype(int), sl["fr_y"]) and np.array_equal(a[:, 2].astype(int), sl["fr_x"])
    # the reader makes every timestamp relative to the first line (bf_motion_compensator.cpp:188-199)
    np.testing.assert_allclose(a[:, 0], (sl["t"] - sl["t"][0]) * 1e-9, rtol=0, atol=3e-9)
    # the committed golden run of this input (tests/golden/make_golden.py): same slices, same per-event output
    import json
    gold = os.path.join(ROOT, "tests", "golden")
    man = json.load(open(os.path.join(gold, "manifest.json")))["stream"]
    zs = np.load(os.path.join(gold, man["file"]))
    assert (slices, skipped, iters) == (man["cli"]["slices"], man["cli"]["skipped"], man["cli"]["iterations"])
    assert np.array_equal(a[:, 0], zs["cli_t"]) and np.array_equal(a[:, 4], zs["cli_v"]) and np.array_equal(a[:, 5], zs["cli_u"])
    vr, vc = sl["velocity"]
    assert abs(a[:, 4].mean() - vc) < 0.01 * abs(vc) and abs(a[:, 5].mean() - vr) < 0.01 * abs(vr)
    # --stm-disable: every slice cold-started -> more iterations in total
    stdout2 = run_cli(oracle_cli, ["--stm-disable", "--quiet", "-o", out_file, path], str(tmp_path))
    assert "slices:" not in stdout2               # --quiet is honoured
    stdout3 = run_cli(oracle_cli, ["--stm-disable", path], str(tmp_path))
    assert parse_summary(stdout3)[2] > iters


def test_oracle_chain_order_spread(oracle_cli, events_txt, tmp_path):
    """How far the ORACLE moves when nothing but the event order inside every slice changes (the reference's time
    image is an f32 running sum in container order, accel_lib.h:162), over the whole config-1 STM chain: the yardstick
    for any bar put on a GPU-vs-oracle comparison of this chain.  It is four orders of magnitude below north_star's
    1e-4 / 0.02 px/s, so that bar is used as it stands (test_cli_gpu_matches_oracle_cli)."""
    path, sl = events_txt
    for extra in ([], ["--stm-disable"], ["--max-iter=10"]):
        outs = []
        for order in ("forward", "reversed"):
            out = str(tmp_path / ("o_%s.txt" % order))
            env = dict(os.environ, BF_SHIM_EVENT_ORDER=order)
            r = subprocess.run([oracle_cli] + extra + ["-o", out, path], cwd=str(tmp_path), env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            outs.append((parse_summary(r.stdout.decode()), np.loadtxt(out)))
        (sa, a), (sb, b) = outs
        assert sa == sb                                    # same slices, same iteration counts
        assert np.array_equal(a[:, :4], b[:, :4])
        spread = max(np.abs(a[:, 4] - b[:, 4]).max(), np.abs(a[:, 5] - b[:, 5]).max())
        assert 0.0 < spread < 1e-4, (extra, spread)        # order matters, at the 1e-6 px/s level


@pytest.mark.gpu
def test_cli_gpu_matches_oracle_cli(oracle_cli, events_txt, tmp_path):
    path, sl = events_txt
    gpu_cli = os.path.join(ROOT, "better_flow_amd", "host", "bf_motion_compensator")
    assert os.path.exists(gpu_cli), "build() must have produced the product CLI"
    for extra in ([], ["--stm-disable"], ["--max-iter=10"]):
        o_out, g_out = str(tmp_path / "o.txt"), str(tmp_path / "g.txt")
        so = run_cli(oracle_cli, extra + ["-o", o_out, path], str(tmp_path))
        sg = run_cli(gpu_cli, extra + ["-o", g_out, path], str(tmp_path))
        os_, og_ = parse_summary(so), parse_summary(sg)
        assert os_[:2] == og_[:2], (extra, os_, og_)              # same slices, same skip decisions
        assert abs(os_[2] - og_[2]) <= os_[0], (extra, os_, og_)   # iteration counts within +-1 per slice
        a, b = np.loadtxt(o_out), np.loadtxt(g_out)
        assert a.shape == b.shape
        assert np.array_equal(a[:, :4], b[:, :4])
        # The whole STM chain (4 slices, each warm-started from the previous model) at north_star's bar: 1e-4 relative
        # or 0.02 px/s.  Measured (scripts/chain_spread.py): GPU vs oracle <= 7e-6 px/s on every flag set, the same
        # size as the oracle's own forward / reversed event-order spread (4e-6 px/s, test_oracle_chain_order_spread).
        for col in (4, 5):
            d = np.abs(a[:, col] - b[:, col])
            assert np.all(d <= np.maximum(1e-4 * np.abs(a[:, col]), 0.02)), (extra, col, d.max())
    # ... and against the committed golden run (no oracle involved): default flags
    import json
    gold = os.path.join(ROOT, "tests", "golden")
    man = json.load(open(os.path.join(gold, "manifest.json")))["stream"]
    zs = np.load(os.path.join(gold, man["file"]))
    g_out = str(tmp_path / "g2.txt")
    sg = parse_summary(run_cli(gpu_cli, ["-o", g_out, path], str(tmp_path)))
    assert sg[:2] == (man["cli"]["slices"], man["cli"]["skipped"]) and abs(sg[2] - man["cli"]["iterations"]) <= sg[0]
    b = np.loadtxt(g_out)
    assert np.array_equal(b[:, 0], zs["cli_t"]) and np.array_equal(b[:, 1].astype(np.int32), zs["cli_x"])
    for col, key in ((4, "cli_v"), (5, "cli_u")):
        assert np.all(np.abs(b[:, col] - zs[key]) <= np.maximum(1e-4 * np.abs(zs[key]), 0.02))
    # the library identifies itself as the HIP build, not the test shim
    ver = subprocess.check_output([gpu_cli, "--version"]).decode()
    assert "gfx950" in ver and "SHIM" not in ver


# ---- OptimizerLocal through the host class (better_flow/optimizer_sampler.h) ----

def _build_test_local(out_dir, against_gpu):
    host = os.path.join(ROOT, "better_flow_amd", "host")
    src = os.path.join(ROOT, "tests", "cpp", "test_local.cpp")
    exe = os.path.join(out_dir, "test_local_gpu" if against_gpu else "test_local_oracle")
    base = ["g++", "-O2", "-std=c++14", "-ffp-contract=off", "-I" + host, "-I" + os.path.join(ROOT, "include"), src]
    if against_gpu:
        subprocess.check_call(base + ["-L" + os.path.join(ROOT, "better_flow_amd"), "-lbf_accel",
                                      "-Wl,-rpath," + os.path.join(ROOT, "better_flow_amd"), "-Wl,-rpath,/opt/rocm/lib",
                                      "-o", exe])
    else:
        obj = os.path.join(out_dir, "bf_oracle_local.o")
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-c",
                               os.path.join(ROOT, "oracle", "bf_oracle.c"), "-o", obj])
        subprocess.check_call(base + [os.path.join(ROOT, "tests", "shim", "bf_accel_oracle_shim.cpp"), obj, "-lm",
                                      "-o", exe])
    return exe


def _local_lines(exe, path, cwd):
    return [ln for ln in run_cli(exe, [path], cwd).splitlines() if ln.startswith(("cloud", "window"))]


def test_optimizer_local_host_class_oracle(events_txt, tmp_path):
    path, _ = events_txt
    lines = _local_lines(_build_test_local(str(tmp_path), False), path, str(tmp_path))
    assert len(lines) == 3
    m = re.match(r"cloud rc=0 nx=(\S+) ny=(\S+) score=(\S+) evals=(\d+)", lines[0])
    assert m and int(m.group(4)) > 10 and float(m.group(3)) > 1.0
    s0 = float(re.match(r"cloud score0=(\S+) img=(\d+)x(\d+)", lines[1]).group(1))
    assert float(m.group(3)) >= s0          # the descent did not end below the score at (0, 0)
    assert lines[2].startswith("window score=")


@pytest.mark.gpu
def test_optimizer_local_host_class_gpu_matches_oracle(events_txt, tmp_path):
    """Same C++ caller, oracle shim vs libbf_accel.so: identical text (scores are exact integers ratios,
    so the descent takes the same path)."""
    path, _ = events_txt
    want = _local_lines(_build_test_local(str(tmp_path), False), path, str(tmp_path))
    got = _local_lines(_build_test_local(str(tmp_path), True), path, str(tmp_path))
    assert got == want


# ---- event input: fast text parser and the binary structure-of-arrays format ----

def _build_cpp(tmp_path, name):
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-O2", "-std=c++14", "-I" + os.path.join(ROOT, "better_flow_amd", "host"),
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", name + ".cpp"),
                           "-o", exe])
    return exe


def test_event_reader_matches_iostream(events_txt, tmp_path):
    exe = _build_cpp(tmp_path, "test_reader")
    path, sl = events_txt
    out = subprocess.check_output([exe, path]).decode()
    assert out.strip() == "records %d mismatches 0" % len(sl["t"])
    rng = np.random.default_rng(3)
    # awkward but legal numbers, mixed whitespace, then a malformed record where both must stop
    lines = []
    for k in range(4000):
        digits = int(rng.integers(1, 25))
        t = "".join(str(d) for d in rng.integers(0, 10, digits))
        cut = int(rng.integers(0, digits + 1))
        t = (t[:cut] or "0") + "." + t[cut:] if rng.random() < 0.8 else t
        if rng.random() < 0.1:
            t += "e%d" % rng.integers(-30, 30)
        if rng.random() < 0.1:
            t = "-" + t
        sep = rng.choice([" ", "\t", "  ", " \t "])
        lines.append(sep.join([t, str(rng.integers(0, 70000)), str(rng.integers(0, 400)), str(rng.integers(0, 2))]))
    weird = str(tmp_path / "weird.txt")
    open(weird, "w").write("\n".join(lines[:3000]) + "\r\n\n" + "\n".join(lines[3000:]) + "\n1.5 10 20 2\n3.0 1 1 1\n")
    out = subprocess.check_output([exe, weird]).decode()
    assert out.strip() == "records 4000 mismatches 0"      # "... 2" is not a bool: both stop there
    for content, n in (("", 0), ("abc 1 2 1\n", 0), ("1.0 2 3 1\n4.0 x 3 1\n", 1), ("1 2 3 1 5", 1)):
        open(weird, "w").write(content)
        assert subprocess.check_output([exe, weird]).decode().strip() == "records %d mismatches 0" % n


def test_cli_binary_input_equals_text_input(oracle_cli, events_txt, tmp_path):
    """--to-bin converts; the binary file drives the same slices and writes the same output file."""
    path, sl = events_txt
    binp = str(tmp_path / "ev.bin")
    run_cli(oracle_cli, ["--to-bin=" + binp, path], str(tmp_path))
    raw = open(binp, "rb").read()
    n = len(sl["t"])
    assert raw[:8] == b"BFEVSOA1" and int.from_bytes(raw[8:16], "little") == n and len(raw) == 16 + 13 * n
    cols = np.frombuffer(raw, dtype="<u2", count=n, offset=16 + 8 * n)
    assert np.array_equal(cols, sl["fr_y"].astype(np.uint16))        # x is the column
    outs = []
    for inp in (path, binp):
        o = str(tmp_path / ("o_%d.txt" % len(outs)))
        so = run_cli(oracle_cli, ["-o", o, inp], str(tmp_path))
        outs.append((parse_summary(so), open(o).read()))
    assert outs[0] == outs[1]
    # synth.write_bin writes the same bytes as the converter
    py = str(tmp_path / "py.bin")
    synth.write_bin(py, sl)
    assert open(py, "rb").read() == raw


# ---- StreamFlow: the structure-of-arrays slice former (better_flow/stream_flow.h) ----

def _build_test_stream(out_dir, against_gpu):
    host = os.path.join(ROOT, "better_flow_amd", "host")
    src = os.path.join(ROOT, "tests", "cpp", "test_stream.cpp")
    exe = os.path.join(out_dir, "test_stream_gpu" if against_gpu else "test_stream_oracle")
    base = ["g++", "-O2", "-std=c++14", "-ffp-contract=off", "-I" + host, "-I" + os.path.join(ROOT, "include"), src]
    if against_gpu:
        subprocess.check_call(base + ["-L" + os.path.join(ROOT, "better_flow_amd"), "-lbf_accel",
                                      "-Wl,-rpath," + os.path.join(ROOT, "better_flow_amd"), "-Wl,-rpath,/opt/rocm/lib",
                                      "-o", exe])
    else:
        obj = os.path.join(out_dir, "bf_oracle_stream.o")
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-c",
                               os.path.join(ROOT, "oracle", "bf_oracle.c"), "-o", obj])
        subprocess.check_call(base + [os.path.join(ROOT, "tests", "shim", "bf_accel_oracle_shim.cpp"), obj, "-lm",
                                      "-o", exe])
    return exe


def _check_stream_output(out):
    verdicts = [ln for ln in out.splitlines() if ln.startswith(("OK", "FAIL"))]
    assert len(verdicts) == 3 and all(v.startswith("OK") for v in verdicts), out[-3000:]
    # the small ring really filled (full-ring quirk: one element fewer iterated) and the short span really trimmed
    assert "ring 3000 / 3000, iterated 2999" in out
    assert any("ring 3000" not in ln and "(ring" not in ln and "/ " in ln and int(ln.split("ring ")[1].split(" /")[0]) < 3000
               for ln in out.splitlines() if ln.startswith("slice") )


def test_stream_flow_equals_dvs_flow_oracle(events_txt, tmp_path):
    """Same stream through DVS_flow (AoS ring, repack per slice) and StreamFlow (pinned SoA ring, ring hand-off): same
    triggers, slices, models and per-event flow -- here on the oracle shim (which holds ring slices in the reference's
    newest -> oldest order, so even the f32 accumulation order is the same)."""
    path, _ = events_txt
    out = run_cli(_build_test_stream(str(tmp_path), False), [path], str(tmp_path))
    _check_stream_output(out)


@pytest.mark.gpu
def test_stream_flow_equals_dvs_flow_gpu(events_txt, tmp_path):
    path, _ = events_txt
    out = run_cli(_build_test_stream(str(tmp_path), True), [path], str(tmp_path))
    _check_stream_output(out)


# ---- --img / --video: one 2 x 2 frame per slice (raw grey | raw colour-time / compensated grey | compensated colour-time) ----

def _read_ppm(path):
    raw = open(path, "rb").read()
    magic, dims, maxv, rest = raw.split(b"\n", 3)
    assert magic == b"P6" and maxv == b"255"
    w, h = (int(x) for x in dims.split())
    return np.frombuffer(rest, dtype=np.uint8, count=w * h * 3).reshape(h, w, 3)   # R, G, B


def _read_avi(path):
    """Minimal reader for the uncompressed AVI the host writes: returns (fps, [frames as rows x cols x 3 RGB])."""
    import struct
    b = open(path, "rb").read()
    assert b[:4] == b"RIFF" and b[8:12] == b"AVI " and struct.unpack("<I", b[4:8])[0] == len(b) - 8
    us_per_frame, = struct.unpack("<I", b[32:36])
    n_frames, = struct.unpack("<I", b[48:52])
    w, h = struct.unpack("<II", b[64:72])
    assert b[112:116] == b"DIB "
    movi = b.index(b"movi")
    stride = (3 * w + 3) & ~3
    frames, pos = [], movi + 4
    for _ in range(n_frames):
        assert b[pos:pos + 4] == b"00db" and struct.unpack("<I", b[pos + 4:pos + 8])[0] == stride * h
        f = np.frombuffer(b, dtype=np.uint8, count=stride * h, offset=pos + 8).reshape(h, stride)[:, :3 * w].reshape(h, w, 3)
        frames.append(f[::-1, :, ::-1])   # bottom-up B G R -> top-down R G B
        pos += 8 + stride * h
    assert b[pos:pos + 4] == b"idx1" and struct.unpack("<I", b[pos + 4:pos + 8])[0] == 16 * n_frames
    return round(1e6 / us_per_frame), frames


def _frames(exe, path, tmp_path, tag, video=False):
    d = tmp_path / ("frames_" + tag)
    d.mkdir()
    extra = ["--video", "--video-name", str(d / "out.avi"), "--video-fps=25"] if video else []
    so = run_cli(exe, ["--img", "--img-prefix", str(d)] + extra + [path], str(tmp_path))
    n = parse_summary(so)[0]
    frames = [_read_ppm(str(d / ("frame_%d.ppm" % k))) for k in range(n)]
    texts = [open(str(d / ("frame_%d.txt" % k))).read() for k in range(n)]
    return n, frames, texts, (_read_avi(str(d / "out.avi")) if video else None)


def _tiles(f):
    R, C = 3 * 180, 3 * 240
    return f[:R, :C], f[:R, C:], f[R:, :C], f[R:, C:]   # raw grey, raw colour, compensated grey, compensated colour


def test_cli_img_frames_oracle(oracle_cli, events_txt, tmp_path):
    path, _ = events_txt
    n, frames, texts, avi = _frames(oracle_cli, path, tmp_path, "o", video=True)
    assert n == 4 and all(f.shape == (2 * 3 * 180, 2 * 3 * 240, 3) for f in frames)
    for f in frames:
        raw, raw_c, comp, comp_c = _tiles(f)
        assert (raw[..., 0] == raw[..., 1]).all() and (raw[..., 1] == raw[..., 2]).all()      # grey tiles
        raw, comp = raw[..., 0], comp[..., 0]
        assert (comp > 0).sum() < 0.75 * (raw > 0).sum()        # compensation sharpens the image
        assert abs(float(raw[raw > 0].mean()) - 127) < 12       # brightness normalised to a non-zero mean of 127
        # colour tiles: value 255 wherever there are events, black elsewhere; the lit area tracks the grey tile's (the
        # 543 x 723 -> 540 x 720 bilinear resize adds a fringe of partly lit pixels around every block)
        for grey, col in ((raw, raw_c), (comp, comp_c)):
            lit = col.any(axis=2)
            assert 0.9 * (grey > 0).sum() < lit.sum() < 1.6 * (grey > 0).sum()
            assert (col[lit].max(axis=1) >= 250).mean() > 0.2
    for t in texts:
        lines = t.splitlines()
        assert lines[0].startswith("timestamp: ") and lines[3].startswith("Events: ") and lines[5] == "Model:"
        assert lines[6].startswith("C: (") and lines[-1].startswith("Div: ")
    fps, vf = avi
    assert fps == 25 and len(vf) == n
    for a, b in zip(vf, frames):
        assert np.array_equal(a, b)                              # the video holds the same frames


@pytest.mark.gpu
def test_cli_img_frames_gpu_match_oracle(oracle_cli, events_txt, tmp_path):
    path, _ = events_txt
    gpu_cli = os.path.join(ROOT, "better_flow_amd", "host", "bf_motion_compensator")
    no, fo, to, _ = _frames(oracle_cli, path, tmp_path, "o")
    ng, fg, tg, avi = _frames(gpu_cli, path, tmp_path, "g", video=True)
    assert no == ng and len(avi[1]) == ng
    for a, b, v in zip(fo, fg, avi[1]):
        ra, ca, pa, qa = _tiles(a)
        rb, cb, pb, qb = _tiles(b)
        assert np.array_equal(ra, rb)                            # raw grey tiles: identical
        # compensated grey: the two models agree to ~1e-6, a handful of events may cross a pixel boundary
        assert (pa != pb).mean() < 0.01
        # colour tiles: same lit pixels (raw), hue / saturation on an 8-bit boundary may differ (f32 vs fixed-point sums)
        assert np.array_equal(ca.any(axis=2), cb.any(axis=2))
        assert (np.abs(ca.astype(int) - cb.astype(int)).max(axis=2) > 1).mean() < 0.01
        assert (np.abs(qa.astype(int) - qb.astype(int)).max(axis=2) > 1).mean() < 0.02
        assert np.array_equal(v, b)
    assert [t.splitlines()[3:5] for t in to] == [t.splitlines()[3:5] for t in tg]   # Events / New events lines
