"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU side ("race detection / sanitizers"): the
oracle (oracle/bf_oracle.c), the oracle-backed C-ABI shim and the WHOLE host front end -- event reader (text and binary),
slice ring, DVS_flow with the overlap de-duplication, -o writer, frame writer -- built into one instrumented binary and
run on BASELINE config 1, plus the ring / reader unit programs.  Any report aborts the program (-fno-sanitize-recover)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "shim"))
from better_flow_amd import synth  # noqa: E402

ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


def _run(cmd, cwd):
    r = subprocess.run(cmd, cwd=cwd, env=ENV, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert b"runtime error" not in r.stderr and b"AddressSanitizer" not in r.stderr, r.stderr.decode()[-3000:]
    return r.stdout.decode()


def test_host_front_end_and_oracle_under_sanitizers(tmp_path):
    import build as shim_build
    exe = shim_build.build(sanitize=True)
    plain = shim_build.build()
    sl = synth.make_slice(10000, 180, 240, 0.1, seed=5)
    txt = str(tmp_path / "ev.txt")
    synth.write_txt(txt, sl)
    out_s, out_p = str(tmp_path / "san.txt"), str(tmp_path / "plain.txt")
    so = _run([exe, "-o", out_s, "--slice-log=" + str(tmp_path / "slices.csv"), txt], str(tmp_path))
    _run([plain, "-o", out_p, txt], str(tmp_path))
    assert "slices: 4 (skipped 0)" in so
    assert np.array_equal(np.loadtxt(out_s), np.loadtxt(out_p))          # the instrumented build computes the same
    log = open(str(tmp_path / "slices.csv")).read().strip().splitlines()
    assert log[0] == "slice,events,new_events,rc,iterations,ms,mevents_per_s" and len(log) == 5
    # binary input, frames, capped iterations, STM off: the other branches of the front end
    binf = str(tmp_path / "ev.bin")
    _run([exe, "--to-bin=" + binf, txt], str(tmp_path))
    os.makedirs(str(tmp_path / "frames"), exist_ok=True)
    _run([exe, "--quiet", "--stm-disable", "--max-iter=5", "--img", "--img-prefix", str(tmp_path / "frames"), "--video",
          "--video-name", str(tmp_path / "v.avi"), "-o", out_s, binf], str(tmp_path))
    assert os.path.exists(str(tmp_path / "frames" / "frame_0.ppm"))


def test_unit_programs_under_sanitizers(tmp_path):
    inc = ["-I" + os.path.join(ROOT, "better_flow_amd", "host"), "-I" + os.path.join(ROOT, "include")]
    import build as shim_build
    for name in ("test_ring", "test_reader"):
        exe = str(tmp_path / name)
        subprocess.check_call(["g++", "-O1", "-std=c++14"] + shim_build.SAN + inc +
                              [os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", exe])
        args = []
        if name == "test_reader":
            sl = synth.make_slice(3000, 180, 240, 0.05, seed=9)
            txt = str(tmp_path / "r.txt")
            synth.write_txt(txt, sl)
            args = [txt]
        _run([exe] + args, str(tmp_path))


def test_stream_engine_and_slice_farm_under_thread_sanitizer(tmp_path):
    """ThreadSanitizer over the threaded half of the host front end: StreamEngine (producer thread) + SliceFarm (worker
    threads, submit-side uploads, in-order delivery, noise re-staging) on the oracle shim -- tests/cpp/test_stream.cpp, the
    program that also holds the engine to DVS_flow: pipelined, bulk, three workers, the noise chain.  Any data-race
    report fails the run (halt_on_error)."""
    host = os.path.join(ROOT, "better_flow_amd", "host")
    exe = str(tmp_path / "test_stream_tsan")
    obj = str(tmp_path / "bf_oracle_tsan.o")
    tsan = ["-fsanitize=thread", "-g", "-fno-omit-frame-pointer"]
    subprocess.check_call(["gcc", "-O1", "-std=c11", "-ffp-contract=off"] + tsan + ["-c", os.path.join(ROOT, "oracle", "bf_oracle.c"), "-o", obj])
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-pthread", "-ffp-contract=off"] + tsan +
                          ["-I" + host, "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_stream.cpp"),
                           os.path.join(ROOT, "tests", "shim", "bf_accel_oracle_shim.cpp"), obj, "-lm", "-o", exe])
    sl = synth.make_slice(10000, 180, 240, 0.1, seed=5)
    txt = str(tmp_path / "ev.txt")
    synth.write_txt(txt, sl)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1")
    r = subprocess.run([exe, txt], cwd=str(tmp_path), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1800)
    assert r.returncode == 0 and b"ThreadSanitizer" not in r.stderr, r.stderr.decode()[-4000:]
    assert r.stdout.decode().count("\nOK ") + r.stdout.decode().startswith("OK ") == 8
