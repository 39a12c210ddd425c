"""The C-ABI library loads and exports every symbol include/bf_accel.h declares.
No compute calls (no GPU here); without a device the product must fail loudly."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "bf_accel.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bf_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    from better_flow_amd import accel
    assert sorted(accel.EXPORTS) == header_functions()


def test_library_exports_every_symbol():
    from better_flow_amd import accel
    assert os.path.exists(accel.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(accel.LIB_PATH)
    for name in header_functions():
        assert hasattr(lib, name), name
    lib.bf_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.bf_version()


def test_structs_match_header_sizes():
    """ctypes mirrors vs sizeof() as compiled into the library."""
    from better_flow_amd import accel
    lib = ctypes.CDLL(accel.LIB_PATH)
    out = (ctypes.c_int32 * 8)()
    assert lib.bf_abi_struct_sizes(out, 8) == 8
    mirrors = [accel.Model, accel.Window, accel.RunOpts, accel.RunInfo, accel.TraceRec, accel.Profile,
               accel.LocalWindow, accel.LocalState]
    assert [ctypes.sizeof(m) for m in mirrors] == list(out)
    assert ctypes.sizeof(accel.Model) == 88 and accel.Model.total_dx.offset == 56


def test_no_cpu_fallback_without_device():
    from better_flow_amd import accel
    if accel.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(accel.BfError) as e:
        accel.Accel()
    assert e.value.code == accel.BF_ERR_NODEVICE


def test_product_never_imports_oracle():
    """Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may touch oracle/."""
    pkg = os.path.join(ROOT, "better_flow_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "bf_oracle" not in txt and "import oracle" not in txt, os.path.join(dirpath, f)


def test_release_library_reads_no_debug_hooks():
    """The BF_DEBUG_* / BF_HOST_TIMING / BF_TIMELINE environment hooks exist only in the test build
    (better_flow_amd/debug/libbf_accel.so, `make debug`) and the timeline build: a stray variable in a host's environment
    must not be able to change the release library's margins or make its persistent kernel give up."""
    from better_flow_amd import accel
    blob = open(accel.LIB_PATH, "rb").read()
    for name in (b"BF_DEBUG_", b"BF_HOST_TIMING", b"BF_TIMELINE", b"BF_TILE_THREADS", b"BF_EXP_"):
        assert name not in blob, name
    assert os.path.exists(accel.DEBUG_LIB_PATH), "build first (make -C better_flow_amd/csrc debug)"
    assert b"BF_DEBUG_MARGIN" in open(accel.DEBUG_LIB_PATH, "rb").read()
    # ... and no experiment switch is left in the product's sources
    src = os.path.join(ROOT, "better_flow_amd", "csrc")
    for f in os.listdir(src):
        if f.endswith((".cpp", ".hip", ".h")):
            txt = open(os.path.join(src, f)).read()
            assert "BF_PROTO_" not in txt and "BF_CENSUS" not in txt, f
