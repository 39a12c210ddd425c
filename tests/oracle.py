"""ctypes binding of oracle/libbf_oracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference path (oracle/bf_oracle.h;
"parity unpinned": the reference cannot be built in this image).  It is imported
only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATH = os.path.join(_ORACLE_DIR, "libbf_oracle.so")


class Window(C.Structure):
    _fields_ = [
        ("scale", C.c_int32),
        ("x_min", C.c_int32), ("y_min", C.c_int32), ("x_max", C.c_int32), ("y_max", C.c_int32),
        ("metric_wsizex", C.c_int32), ("metric_wsizey", C.c_int32),
        ("scale_img_x", C.c_int32), ("scale_img_y", C.c_int32),
        ("x_shift", C.c_double), ("y_shift", C.c_double),
    ]


class Model(C.Structure):
    _fields_ = [
        ("cx", C.c_double), ("cy", C.c_double), ("dx", C.c_double), ("dy", C.c_double),
        ("rot", C.c_double), ("div", C.c_double), ("cnt", C.c_uint32),
        ("total_dx", C.c_double), ("total_dy", C.c_double),
        ("total_rot", C.c_double), ("total_div", C.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Loop(C.Structure):
    _fields_ = [
        ("x_divider", C.c_float), ("y_divider", C.c_float),
        ("rot_divider", C.c_float), ("div_divider", C.c_float),
        ("itercount", C.c_int64),
    ]


class TraceRec(C.Structure):
    _fields_ = [("model", Model), ("loop", Loop)]


class LocalWindow(C.Structure):
    _fields_ = [
        ("scale", C.c_int32), ("metric_wsizex", C.c_int32), ("metric_wsizey", C.c_int32),
        ("scale_img_x", C.c_int32), ("scale_img_y", C.c_int32),
        ("c_fr_x", C.c_int32), ("c_fr_y", C.c_int32), ("c_t", C.c_int64),
    ]


class LocalState(C.Structure):
    _fields_ = [
        ("nx", C.c_double), ("ny", C.c_double), ("last_score", C.c_double),
        ("dnx", C.c_double), ("dny", C.c_double), ("dn_th", C.c_double), ("evaluations", C.c_int64),
    ]


class _Cloud(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("fr_x", C.POINTER(C.c_int32)), ("fr_y", C.POINTER(C.c_int32)),
        ("t", C.POINTER(C.c_int64)), ("noise", C.POINTER(C.c_uint8)),
        ("pr_x", C.POINTER(C.c_double)), ("pr_y", C.POINTER(C.c_double)),
        ("nx", C.POINTER(C.c_double)), ("ny", C.POINTER(C.c_double)),
    ]


def build():
    """(Re)build the oracle shared library with gcc; cheap and idempotent."""
    subprocess.check_call(["make", "-s", "-C", _ORACLE_DIR])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.bfo_run.restype = C.c_int
        L.bfo_local_run.restype = C.c_int
        L.bfo_gauss_u8.restype = C.c_int
        L.bfo_nonzero_average.restype = C.c_double
        L.bfo_local_iteration_step.restype = C.c_double
        _lib = L
    return _lib


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


class Cloud:
    """Event cloud with the live Event fields as numpy arrays."""

    def __init__(self, fr_x, fr_y, t):
        self.fr_x = np.ascontiguousarray(fr_x, dtype=np.int32)
        self.fr_y = np.ascontiguousarray(fr_y, dtype=np.int32)
        self.t = np.ascontiguousarray(t, dtype=np.int64)
        n = self.n = len(self.fr_x)
        self.noise = np.zeros(n, dtype=np.uint8)
        self.pr_x = np.full(n, np.nan)
        self.pr_y = np.full(n, np.nan)
        self.nx = np.zeros(n)
        self.ny = np.zeros(n)
        self.c = _Cloud(
            n, _p(self.fr_x, C.c_int32), _p(self.fr_y, C.c_int32), _p(self.t, C.c_int64),
            _p(self.noise, C.c_uint8), _p(self.pr_x, C.c_double), _p(self.pr_y, C.c_double),
            _p(self.nx, C.c_double), _p(self.ny, C.c_double),
        )

    # optimizer_rolling.h:248-283
    def set_cloud(self, scale, res_x, res_y):
        w = Window()
        lib().bfo_set_cloud(C.byref(self.c), C.c_int32(scale), C.c_int32(res_x), C.c_int32(res_y),
                            C.byref(w))
        return w

    # accel_lib.h:263-267
    def project_4param(self, dnx, dny, cx, cy, div, crl):
        lib().bfo_project_4param(C.byref(self.c), C.c_double(dnx), C.c_double(dny), C.c_double(cx), C.c_double(cy),
                                 C.c_double(div), C.c_double(crl))

    def project_4param_reinit(self, dnx, dny, cx, cy, div, crl):
        lib().bfo_project_4param_reinit(C.byref(self.c), C.c_double(dnx), C.c_double(dny),
                                        C.c_double(cx), C.c_double(cy), C.c_double(div),
                                        C.c_double(crl))

    # accel_lib.h:147-178
    def get_time_img(self, w):
        R, Cc = w.scale_img_x, w.scale_img_y
        timg = np.empty((R, Cc), dtype=np.float32)
        cimg = np.empty((R, Cc), dtype=np.float32)
        lib().bfo_get_time_img(C.byref(self.c), C.c_int32(w.metric_wsizex),
                               C.c_int32(w.metric_wsizey), C.c_int32(w.scale),
                               C.c_int32(int(w.x_shift)), C.c_int32(int(w.y_shift)),
                               _p(timg, C.c_float), _p(cimg, C.c_float))
        return timg, cimg

    def set_model(self, last):
        m = Model()
        lib().bfo_set_model(C.byref(self.c), C.byref(m), C.byref(last))
        return m

    def iteration_step(self, w, model, loop):
        lib().bfo_iteration_step(C.byref(self.c), C.byref(w), C.byref(model), C.byref(loop), None)

    # optimizer_rolling.h:48-125
    def run(self, w, model, max_iter=-1, res_x=180, res_y=240, min_events=1000, hard_cap=200000,
            trace_cap=0):
        loop = Loop()
        trace = (TraceRec * max(1, trace_cap))()
        rc = lib().bfo_run(C.byref(self.c), C.byref(w), C.byref(model), C.c_int32(max_iter),
                           C.c_int32(res_x), C.c_int32(res_y), C.c_int32(min_events),
                           C.c_int64(hard_cap), C.byref(loop), trace if trace_cap else None,
                           C.c_int64(trace_cap))
        return rc, loop, (list(trace)[: min(trace_cap, loop.itercount)] if trace_cap else [])

    # ---- OptimizerLocal (optimizer_sampler.h / .cpp) ----
    def local_window(self, scale, center=None, wsz=None):
        w = LocalWindow()
        if center is None:
            lib().bfo_local_window_cloud(C.byref(self.c), C.c_int32(scale), C.byref(w))
        else:
            lib().bfo_local_window_at(C.c_int32(scale), C.c_int32(wsz), C.c_int32(center[0]), C.c_int32(center[1]),
                                      C.c_int64(center[2]), C.byref(w))
        return w

    def local_count_img(self, w, nx, ny):
        img = np.empty((w.scale_img_x, w.scale_img_y), dtype=np.uint8)
        lib().bfo_local_count_img(C.byref(self.c), C.byref(w), C.c_double(nx), C.c_double(ny), _p(img, C.c_uint8))
        return img

    def local_iteration_step(self, w, nx, ny):
        img = np.empty((w.scale_img_x, w.scale_img_y), dtype=np.uint8)
        scratch = np.empty_like(img)
        sc = lib().bfo_local_iteration_step(C.byref(self.c), C.byref(w), C.c_double(nx), C.c_double(ny),
                                            _p(img, C.c_uint8), _p(scratch, C.c_uint8))
        return sc, img

    def local_run(self, w, res_x=180, res_y=240, max_evaluations=100000):
        st = LocalState()
        img = np.empty((max(w.scale_img_x, 1), max(w.scale_img_y, 1)), dtype=np.uint8)
        scratch = np.empty_like(img)
        rc = lib().bfo_local_run(C.byref(self.c), C.byref(w), C.c_int32(res_x), C.c_int32(res_y),
                                 C.c_int64(max_evaluations), C.byref(st), _p(img, C.c_uint8), _p(scratch, C.c_uint8))
        return rc, st, img

    # EventFile::projection_img (event_file.h:460-515)
    def projection_img(self, scale, res_x, res_y, show_final=False):
        img = np.empty((res_x * scale, res_y * scale), dtype=np.uint8)
        scratch = np.empty_like(img)
        lib().bfo_projection_img(C.byref(self.c), C.c_int32(scale), C.c_int32(res_x), C.c_int32(res_y),
                                 C.c_int32(1 if show_final else 0), _p(img, C.c_uint8), _p(scratch, C.c_uint8))
        return img

    # EventFile::color_time_img (event_file.h:649-747)
    def color_time_img(self, scale, res_x, res_y, show_final=False):
        sc = scale if scale else 11
        img = np.empty((res_x * sc + sc, res_y * sc + sc, 3), dtype=np.uint8)
        scratch = np.empty(img.shape, dtype=np.float32)
        lib().bfo_color_time_img(C.byref(self.c), C.c_int32(scale), C.c_int32(res_x), C.c_int32(res_y),
                                 C.c_int32(1 if show_final else 0), _p(img, C.c_uint8), _p(scratch, C.c_float))
        return img

    def compute_uv(self):
        u = np.empty(self.n)
        v = np.empty(self.n)
        lib().bfo_compute_uv(_p(self.nx, C.c_double), _p(self.ny, C.c_double), C.c_int64(self.n),
                             _p(u, C.c_double), _p(v, C.c_double))
        return u, v


def sincos(x):
    """std::sin / std::cos (event.h:102-103) of the float64 array x on this host's libm -> (sin, cos)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    sn, cs = np.empty_like(x), np.empty_like(x)
    lib().bfo_sincos(_p(x, C.c_double), C.c_int64(x.size), _p(sn, C.c_double), _p(cs, C.c_double))
    return sn, cs


def center_of_mass(img):
    img = np.ascontiguousarray(img, dtype=np.float32)
    m = Model()
    lib().bfo_center_of_mass(_p(img, C.c_float), C.c_int32(img.shape[0]), C.c_int32(img.shape[1]),
                             C.byref(m))
    return m


def sobel(img):
    img = np.ascontiguousarray(img, dtype=np.float32)
    gx = np.empty_like(img)
    gy = np.empty_like(img)
    lib().bfo_sobel(_p(img, C.c_float), C.c_int32(img.shape[0]), C.c_int32(img.shape[1]),
                    _p(gx, C.c_float), _p(gy, C.c_float))
    return gx, gy


def fast_model(img):
    img = np.ascontiguousarray(img, dtype=np.float32)
    m = Model()
    lib().bfo_fast_model(_p(img, C.c_float), C.c_int32(img.shape[0]), C.c_int32(img.shape[1]),
                         C.byref(m))
    return m


def set_local_time(timestamp, t0):
    ts = np.ascontiguousarray(timestamp, dtype=np.uint64)
    out = np.empty(len(ts), dtype=np.int64)
    lib().bfo_set_local_time(_p(ts, C.c_uint64), C.c_int64(len(ts)), C.c_uint64(t0),
                             _p(out, C.c_int64))
    return out


def gauss_u8(img, ksize):
    out = np.ascontiguousarray(img, dtype=np.uint8).copy()
    scratch = np.empty_like(out)
    rc = lib().bfo_gauss_u8(_p(out, C.c_uint8), C.c_int32(out.shape[0]), C.c_int32(out.shape[1]), C.c_int32(ksize),
                            _p(scratch, C.c_uint8))
    if rc != 0:
        raise ValueError("unsupported ksize %d" % ksize)
    return out


def nonzero_average(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    return lib().bfo_nonzero_average(_p(img, C.c_uint8), C.c_int64(img.size))
