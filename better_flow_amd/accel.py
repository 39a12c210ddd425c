"""ctypes binding of libbf_accel.so (include/bf_accel.h) -- plumbing for tests and bench.py.

The product is the C-ABI library and the C++ host code on top of it
(better_flow_amd/host/); this module only lets Python drive the same entry points.
There is no CPU path here: if the library is missing, or no HIP device is present,
construction raises.
"""
import ctypes as C
import os

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BF_ACCEL_LIB") or os.path.join(_DIR, "libbf_accel.so")
# The test build (`make -C better_flow_amd/csrc debug`): the same objects plus the BF_DEBUG_* environment hooks the release
# library does not contain.  Only tests load it (tests/helpers.py), by passing lib=DEBUG_LIB_PATH.
DEBUG_LIB_PATH = os.path.join(_DIR, "debug", "libbf_accel.so")

BF_OK, BF_SKIPPED = 0, 1
BF_ERR_ARG, BF_ERR_HIP, BF_ERR_STATE, BF_ERR_NOCONV, BF_ERR_NODEVICE, BF_ERR_CAPACITY = (
    -1, -2, -3, -4, -5, -6)


class Model(C.Structure):
    """bf_model == ObjectModel (object_model.h:10-13)."""
    _fields_ = [
        ("cx", C.c_double), ("cy", C.c_double), ("dx", C.c_double), ("dy", C.c_double),
        ("rot", C.c_double), ("div", C.c_double), ("cnt", C.c_uint32), ("_pad", C.c_uint32),
        ("total_dx", C.c_double), ("total_dy", C.c_double),
        ("total_rot", C.c_double), ("total_div", C.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "_pad"}


class Window(C.Structure):
    _fields_ = [
        ("scale", C.c_int32),
        ("x_min", C.c_int32), ("y_min", C.c_int32), ("x_max", C.c_int32), ("y_max", C.c_int32),
        ("metric_wsizex", C.c_int32), ("metric_wsizey", C.c_int32),
        ("scale_img_x", C.c_int32), ("scale_img_y", C.c_int32), ("_pad", C.c_int32),
        ("x_shift", C.c_double), ("y_shift", C.c_double),
    ]


class RunOpts(C.Structure):
    _fields_ = [
        ("max_iter", C.c_int32), ("min_events", C.c_int32), ("res_x", C.c_int32),
        ("res_y", C.c_int32), ("hard_iter_cap", C.c_int32), ("poll_interval", C.c_int32),
        ("trace_cap", C.c_int32), ("want_uv", C.c_int32),
    ]


class RunInfo(C.Structure):
    _fields_ = [
        ("rc", C.c_int32), ("iterations", C.c_int32),
        ("x_divider", C.c_float), ("y_divider", C.c_float),
        ("rot_divider", C.c_float), ("div_divider", C.c_float),
        ("launches", C.c_int32), ("polls", C.c_int32),
        ("rebins", C.c_int32), ("overflow_events", C.c_int32),
    ]


class TileOpts(C.Structure):
    _fields_ = [
        ("grid_rows", C.c_int32), ("grid_cols", C.c_int32), ("scale", C.c_int32),
        ("sensor_res_x", C.c_int32), ("sensor_res_y", C.c_int32),
        ("guard_res_x", C.c_int32), ("guard_res_y", C.c_int32),
        ("min_events", C.c_int32), ("max_iter", C.c_int32), ("hard_iter_cap", C.c_int32),
    ]


class TraceRec(C.Structure):
    _fields_ = [
        ("model", Model),
        ("x_divider", C.c_float), ("y_divider", C.c_float),
        ("rot_divider", C.c_float), ("div_divider", C.c_float),
        ("iteration", C.c_int32), ("_pad", C.c_int32),
    ]


class Profile(C.Structure):
    _fields_ = [
        ("warp_scatter_ms", C.c_double), ("warp_scatter_launches", C.c_uint64),
        ("stencil_ms", C.c_double), ("stencil_launches", C.c_uint64),
        ("update_ms", C.c_double), ("update_launches", C.c_uint64),
        ("other_ms", C.c_double), ("other_launches", C.c_uint64),
        ("warp_scatter_events", C.c_uint64),
    ]


class LocalWindow(C.Structure):
    _fields_ = [
        ("scale", C.c_int32), ("metric_wsizex", C.c_int32), ("metric_wsizey", C.c_int32),
        ("scale_img_x", C.c_int32), ("scale_img_y", C.c_int32),
        ("c_fr_x", C.c_int32), ("c_fr_y", C.c_int32), ("_pad", C.c_int32), ("c_t", C.c_int64),
    ]


class LocalState(C.Structure):
    _fields_ = [
        ("nx", C.c_double), ("ny", C.c_double), ("last_score", C.c_double),
        ("dnx", C.c_double), ("dny", C.c_double), ("dn_th", C.c_double), ("evaluations", C.c_int64),
    ]


class LocalTileOpts(C.Structure):
    _fields_ = [
        ("grid_rows", C.c_int32), ("grid_cols", C.c_int32), ("scale", C.c_int32), ("wsz", C.c_int32),
        ("sensor_res_x", C.c_int32), ("sensor_res_y", C.c_int32), ("guard_res_x", C.c_int32), ("guard_res_y", C.c_int32),
        ("max_evaluations", C.c_int64),
    ]


# every symbol include/bf_accel.h declares
EXPORTS = [
    "bf_device_count", "bf_create", "bf_destroy", "bf_last_error", "bf_version",
    "bf_run_opts_default", "bf_abi_struct_sizes", "bf_set_option", "bf_get_stat", "bf_upload_events", "bf_upload_events_device",
    "bf_set_cloud", "bf_project_4param", "bf_project_4param_reinit", "bf_get_time_img", "bf_sobel", "bf_fast_model",
    "bf_writeout_events", "bf_compute_uv", "bf_set_model", "bf_run", "bf_run_many", "bf_run_tiles", "bf_run_tiles_many", "bf_get_trace",
    "bf_profile_enable", "bf_profile_reset", "bf_profile_get", "bf_synchronize",
    "bf_copy_bandwidth", "bf_device_malloc", "bf_device_free", "bf_memcpy_h2d",
    "bf_host_alloc", "bf_host_free", "bf_upload_events_async", "bf_commit_upload",
    "bf_local_set_window", "bf_local_iteration_step", "bf_local_run", "bf_local_run_tiles",
    "bf_upload_ring_async", "bf_upload_ring16_async", "bf_upload_ring16t32_async", "bf_upload_events16_async", "bf_compute_uv_ring", "bf_wait_uploads", "bf_projection_img",
    "bf_color_time_img", "bf_eval_sincos", "bf_device_numa_node", "bf_bind_thread_to_numa_node", "bf_bind_thread_to_device_numa",
]

_lib = None


def device_numa_node(device=0):
    """Host NUMA node of HIP device `device` (-1: the platform does not say)."""
    node = C.c_int32(-1)
    load().bf_device_numa_node(device, C.byref(node))
    return node.value


def bind_thread_to_numa_node(node):
    """Bind the CALLING thread to the CPUs of host NUMA node `node` (that the process may use); returns the number of CPUs it
    is now bound to, 0 when nothing was done (unknown node, not a NUMA system, CPUs outside the container's cpuset)."""
    n = C.c_int32(0)
    rc = load().bf_bind_thread_to_numa_node(node, C.byref(n))
    if rc < 0:
        raise BfError(rc, "bf_bind_thread_to_numa_node(%d)" % node)
    return n.value


def bind_thread_to_device_numa(device=0):
    """bf_bind_thread_to_device_numa: the calling thread next to its GPU; returns the node (-1: unknown, nothing done)."""
    node = C.c_int32(-1)
    load().bf_bind_thread_to_device_numa(device, C.byref(node))
    return node.value


def run_many(accels, opts=None):
    """bf_run_many: the slices staged on `accels` (Accel objects, each after upload + set_cloud) solved together.
    Returns [(rc, Model, RunInfo)]."""
    n = len(accels)
    L = load()
    hs = (C.c_void_p * n)(*[a.h for a in accels])
    models = (Model * n)()
    infos = (RunInfo * n)()
    rc = L.bf_run_many(hs, n, C.byref(opts) if opts is not None else None, models, infos)
    if rc < 0:
        bad = next((a for a, i in zip(accels, infos) if i.rc < 0), accels[0])
        raise BfError(rc, L.bf_last_error(bad.h).decode())
    return [(infos[i].rc, models[i], infos[i]) for i in range(n)]


def run_tiles_many(accels, grid_rows, grid_cols, scale, sensor_res, guard_res, min_events, max_iter=-1, hard_iter_cap=20000):
    """bf_run_tiles_many: the tile grids of the slices uploaded on `accels` in one launch.  Returns [(models, infos)] per slice."""
    n = len(accels)
    L = accels[0].L
    o = TileOpts(grid_rows, grid_cols, scale, sensor_res[0], sensor_res[1], guard_res[0], guard_res[1], min_events, max_iter,
                 hard_iter_cap)
    nt = grid_rows * grid_cols
    hs = (C.c_void_p * n)(*[a.h for a in accels])
    models = (Model * (n * nt))()
    infos = (RunInfo * (n * nt))()
    rc = L.bf_run_tiles_many(hs, n, C.byref(o), models, infos)
    if rc < 0:
        raise BfError(rc, L.bf_last_error(accels[0].h).decode())
    return [(list(models[i * nt:(i + 1) * nt]), list(infos[i * nt:(i + 1) * nt])) for i in range(n)]


class BfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("bf_accel error %d: %s" % (code, msg))
        self.code = code


_libs = {}


def load(path=None):
    """Load libbf_accel.so (or the build at `path`); raises (never falls back) when it has not been built."""
    global _lib
    path = path or LIB_PATH
    if path not in _libs:
        if not os.path.exists(path):
            raise RuntimeError(
                "libbf_accel.so is missing (%s): build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C better_flow_amd/csrc`" % path)
        L = C.CDLL(path)
        L.bf_last_error.restype = C.c_char_p
        L.bf_version.restype = C.c_char_p
        L.bf_create.argtypes = [C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                                C.POINTER(C.c_void_p)]
        L.bf_destroy.argtypes = [C.c_void_p]
        L.bf_destroy.restype = None
        L.bf_last_error.argtypes = [C.c_void_p]
        L.bf_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        L.bf_get_stat.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]
        L.bf_upload_events.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.c_int64]
        L.bf_upload_events_device.argtypes = [C.c_void_p] + [C.c_void_p] * 3 + [C.c_int64]
        L.bf_set_cloud.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Window)]
        L.bf_local_set_window.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64,
                                          C.POINTER(LocalWindow)]
        L.bf_local_iteration_step.argtypes = [C.c_void_p, C.c_double, C.c_double, C.POINTER(C.c_double), C.c_void_p]
        L.bf_local_run.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.POINTER(LocalState)]
        L.bf_local_run_tiles.argtypes = [C.c_void_p, C.POINTER(LocalTileOpts), C.c_void_p, C.c_void_p]
        L.bf_projection_img.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        L.bf_color_time_img.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        L.bf_upload_ring_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                           C.c_int64, C.c_uint64]
        L.bf_upload_ring16_async.argtypes = L.bf_upload_ring_async.argtypes
        L.bf_upload_ring16t32_async.argtypes = L.bf_upload_ring_async.argtypes
        L.bf_upload_events16_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.bf_compute_uv_ring.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
        L.bf_wait_uploads.argtypes = [C.c_void_p]
        L.bf_project_4param_reinit.argtypes = [C.c_void_p] + [C.c_double] * 6
        L.bf_project_4param.argtypes = [C.c_void_p] + [C.c_double] * 6
        L.bf_get_time_img.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.bf_sobel.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.bf_fast_model.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(Model)]
        L.bf_writeout_events.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.bf_compute_uv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.bf_set_model.argtypes = [C.c_void_p, C.POINTER(Model)]
        L.bf_run.argtypes = [C.c_void_p, C.POINTER(RunOpts), C.POINTER(Model), C.POINTER(RunInfo)]
        L.bf_run_tiles.argtypes = [C.c_void_p, C.POINTER(TileOpts), C.c_void_p, C.c_void_p]
        L.bf_run_tiles_many.argtypes = [C.c_void_p, C.c_int32, C.POINTER(TileOpts), C.c_void_p, C.c_void_p]
        L.bf_run_many.argtypes = [C.c_void_p, C.c_int32, C.POINTER(RunOpts), C.c_void_p, C.c_void_p]
        L.bf_get_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        L.bf_profile_enable.argtypes = [C.c_void_p, C.c_int32]
        L.bf_profile_reset.argtypes = [C.c_void_p]
        L.bf_profile_get.argtypes = [C.c_void_p, C.POINTER(Profile)]
        L.bf_synchronize.argtypes = [C.c_void_p]
        L.bf_copy_bandwidth.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_double)]
        L.bf_eval_sincos.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
        L.bf_device_numa_node.argtypes = [C.c_int32, C.POINTER(C.c_int32)]
        L.bf_bind_thread_to_numa_node.argtypes = [C.c_int32, C.POINTER(C.c_int32)]
        L.bf_bind_thread_to_device_numa.argtypes = [C.c_int32, C.POINTER(C.c_int32)]
        L.bf_run_opts_default.argtypes = [C.POINTER(RunOpts)]
        L.bf_device_count.argtypes = [C.POINTER(C.c_int32)]
        L.bf_device_malloc.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
        L.bf_device_free.argtypes = [C.c_void_p, C.c_void_p]
        L.bf_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.bf_host_alloc.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
        L.bf_host_free.argtypes = [C.c_void_p, C.c_void_p]
        L.bf_upload_events_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.bf_commit_upload.argtypes = [C.c_void_p]
        _libs[path] = L
        if path == LIB_PATH:
            _lib = L
    return _libs[path]


def device_count():
    n = C.c_int32(0)
    load().bf_device_count(C.byref(n))
    return n.value


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Accel:
    """One bf_ctx: the AccelLib-equivalent plus the fused OptimizerRolling::run."""

    def __init__(self, device=0, max_events=1 << 20, max_rows=1024, max_cols=1280, stream=None, lib=None):
        self.L = load(lib)
        h = C.c_void_p()
        rc = self.L.bf_create(device, max_events, max_rows, max_cols, stream, C.byref(h))
        if rc != BF_OK:
            raise BfError(rc, "bf_create failed (no HIP device?)" if rc == BF_ERR_NODEVICE
                          else "bf_create failed")
        self.h = h
        self.n = 0
        self.window = None

    def close(self):
        if getattr(self, "h", None):
            for p in getattr(self, "_pinned", []):
                self.L.bf_host_free(self.h, p)
            self._pinned = []
            self.L.bf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, ok=(BF_OK,)):
        if rc not in ok:
            raise BfError(rc, self.L.bf_last_error(self.h).decode())
        return rc

    def set_option(self, key, value):
        self._chk(self.L.bf_set_option(self.h, key.encode(), int(value)))

    def get_stat(self, key):
        v = C.c_int64(0)
        self._chk(self.L.bf_get_stat(self.h, key.encode(), C.byref(v)))
        return v.value

    def upload_events(self, fr_x, fr_y, t_ns, noise=None):
        fr_x = np.ascontiguousarray(fr_x, dtype=np.int32)
        fr_y = np.ascontiguousarray(fr_y, dtype=np.int32)
        t_ns = np.ascontiguousarray(t_ns, dtype=np.int32)
        if noise is not None:
            noise = np.ascontiguousarray(noise, dtype=np.uint8)
        self.n = len(fr_x)
        self._chk(self.L.bf_upload_events(self.h, _ptr(fr_x), _ptr(fr_y), _ptr(t_ns), _ptr(noise),
                                          self.n))

    def upload_events_device(self, d_fr_x, d_fr_y, d_t, n):
        self.n = int(n)
        self._chk(self.L.bf_upload_events_device(self.h, d_fr_x, d_fr_y, d_t, self.n))

    def set_cloud(self, scale=3, res_x=180, res_y=240):
        w = Window()
        self._chk(self.L.bf_set_cloud(self.h, scale, res_x, res_y, C.byref(w)))
        self.window = w
        return w

    def project_4param_reinit(self, dnx, dny, cx, cy, div, crl):
        self._chk(self.L.bf_project_4param_reinit(self.h, dnx, dny, cx, cy, div, crl))

    def project_4param(self, dnx, dny, cx, cy, div, crl):
        """The incremental warp (AccelLib::project_4param, accel_lib.h:275-281): dn added to the events' (nx, ny)."""
        self._chk(self.L.bf_project_4param(self.h, dnx, dny, cx, cy, div, crl))

    def get_time_img(self, want_time=True, want_count=True):
        R, Cc = self.window.scale_img_x, self.window.scale_img_y
        t = np.empty((R, Cc), dtype=np.float32) if want_time else None
        c = np.empty((R, Cc), dtype=np.uint32) if want_count else None
        self._chk(self.L.bf_get_time_img(self.h, _ptr(t), _ptr(c)))
        return t, c

    def sobel(self, img):
        img = np.ascontiguousarray(img, dtype=np.float32)
        gx = np.empty_like(img)
        gy = np.empty_like(img)
        self._chk(self.L.bf_sobel(self.h, _ptr(img), img.shape[0], img.shape[1], _ptr(gx), _ptr(gy)))
        return gx, gy

    def fast_model(self, img=None):
        m = Model()
        if img is None:
            self._chk(self.L.bf_fast_model(self.h, None, 0, 0, C.byref(m)))
        else:
            img = np.ascontiguousarray(img, dtype=np.float32)
            self._chk(self.L.bf_fast_model(self.h, _ptr(img), img.shape[0], img.shape[1], C.byref(m)))
        return m

    def writeout_events(self):
        out = [np.empty(self.n) for _ in range(4)]
        self._chk(self.L.bf_writeout_events(self.h, *[_ptr(a) for a in out]))
        return tuple(out)   # pr_x, pr_y, nx, ny

    def compute_uv(self):
        u, v = np.empty(self.n), np.empty(self.n)
        self._chk(self.L.bf_compute_uv(self.h, _ptr(u), _ptr(v)))
        return u, v

    def set_model(self, model):
        self._chk(self.L.bf_set_model(self.h, C.byref(model)))

    # ---- OptimizerLocal (optimizer_sampler.h:12-68): the contrast-score optimiser ----
    def local_set_window(self, scale, center=None, wsz=0):
        """center=None: window = bounding box of the cloud; else center = (fr_x, fr_y, t_ns) and wsz."""
        w = LocalWindow()
        cx, cy, ct = center if center is not None else (0, 0, 0)
        self._chk(self.L.bf_local_set_window(self.h, scale, wsz if center is not None else 0, cx, cy, ct,
                                             C.byref(w)))
        self._lwin = w
        return w

    def local_iteration_step(self, nx, ny, want_img=False):
        sc = C.c_double()
        img = np.empty((self._lwin.scale_img_x, self._lwin.scale_img_y), dtype=np.uint8) if want_img else None
        self._chk(self.L.bf_local_iteration_step(self.h, nx, ny, C.byref(sc), _ptr(img) if want_img else None))
        return (sc.value, img) if want_img else sc.value

    def projection_img(self, scale, res_x, res_y, show_final=False):
        """EventFile::projection_img (event_file.h:460-515): the (motion-compensated) 8-bit event image."""
        img = np.empty((res_x * scale, res_y * scale), dtype=np.uint8)
        self._chk(self.L.bf_projection_img(self.h, scale, res_x, res_y, 1 if show_final else 0, _ptr(img)))
        return img

    def color_time_img(self, scale, res_x, res_y, show_final=False):
        """EventFile::color_time_img (event_file.h:649-747): B, G, R colour-coded time image of the slice."""
        sc = scale if scale else 11
        img = np.empty((res_x * sc + sc, res_y * sc + sc, 3), dtype=np.uint8)
        self._chk(self.L.bf_color_time_img(self.h, scale, res_x, res_y, 1 if show_final else 0, _ptr(img)))
        return img

    def local_run(self, res_x=180, res_y=240, max_evaluations=100000):
        st = LocalState()
        rc = self.L.bf_local_run(self.h, res_x, res_y, max_evaluations, C.byref(st))
        if rc < 0:
            self._chk(rc)
        return rc, st

    def local_run_tiles(self, grid_rows, grid_cols, scale, wsz, sensor_res, guard_res, max_evaluations=100000):
        """A grid of OptimizerLocal windows, one per sensor tile on the tile's own events (bf_local_run_tiles);
        returns ([LocalState], [rc])."""
        o = LocalTileOpts(grid_rows, grid_cols, scale, wsz, sensor_res[0], sensor_res[1], guard_res[0], guard_res[1], max_evaluations)
        nt = grid_rows * grid_cols
        st = (LocalState * nt)()
        rcs = (C.c_int32 * nt)()
        self._chk(self.L.bf_local_run_tiles(self.h, C.byref(o), st, rcs))
        return list(st), list(rcs)

    def default_opts(self):
        o = RunOpts()
        self.L.bf_run_opts_default(C.byref(o))
        return o

    def run(self, opts=None):
        m, info = Model(), RunInfo()
        rc = self.L.bf_run(self.h, C.byref(opts) if opts is not None else None, C.byref(m),
                           C.byref(info))
        self._chk(rc, ok=(BF_OK, BF_SKIPPED))
        return rc, m, info

    def run_tiles(self, grid_rows, grid_cols, scale, sensor_res, guard_res, min_events, max_iter=-1,
                  hard_iter_cap=20000):
        """One independent optimizer per sensor tile (bf_run_tiles); returns (models, infos)."""
        o = TileOpts(grid_rows, grid_cols, scale, sensor_res[0], sensor_res[1], guard_res[0], guard_res[1],
                     min_events, max_iter, hard_iter_cap)
        nt = grid_rows * grid_cols
        models = (Model * nt)()
        infos = (RunInfo * nt)()
        self._chk(self.L.bf_run_tiles(self.h, C.byref(o), models, infos))
        return list(models), list(infos)

    def get_trace(self, cap):
        buf = (TraceRec * max(cap, 1))()
        n = C.c_int32(0)
        self._chk(self.L.bf_get_trace(self.h, buf, cap, C.byref(n)))
        return list(buf)[: n.value]

    def profile_enable(self, mode=1):
        self._chk(self.L.bf_profile_enable(self.h, mode))

    def profile_reset(self):
        self._chk(self.L.bf_profile_reset(self.h))

    def profile_get(self):
        p = Profile()
        self._chk(self.L.bf_profile_get(self.h, C.byref(p)))
        return p

    def synchronize(self):
        self._chk(self.L.bf_synchronize(self.h))

    def pinned_int32(self, n):
        """A pinned host int32 array of n elements (bf_host_alloc) viewed through numpy."""
        p = C.c_void_p()
        self._chk(self.L.bf_host_alloc(self.h, max(4 * n, 16), C.byref(p)))
        arr = np.ctypeslib.as_array((C.c_int32 * n).from_address(p.value))
        self._pinned = getattr(self, "_pinned", []) + [p]   # released in close()
        return arr

    def pinned_array(self, n, dtype):
        """A pinned host array of n elements of `dtype` (bf_host_alloc) viewed through numpy."""
        dt = np.dtype(dtype)
        p = C.c_void_p()
        self._chk(self.L.bf_host_alloc(self.h, max(dt.itemsize * n, 16), C.byref(p)))
        arr = np.frombuffer((C.c_uint8 * (dt.itemsize * n)).from_address(p.value), dtype=dt)
        self._pinned = getattr(self, "_pinned", []) + [p]   # released in close()
        return arr

    def upload_events_async(self, fr_x, fr_y, t_ns, n):
        """fr_x / fr_y / t_ns: pinned int32 arrays (pinned_int32), or uint16 addresses (pinned_array) with int32 times --
        8 instead of 12 bytes per event over the link (bf_upload_events16_async); returns immediately."""
        assert fr_x.dtype == fr_y.dtype and t_ns.dtype == np.int32
        fn = self.L.bf_upload_events16_async if fr_x.dtype == np.uint16 else self.L.bf_upload_events_async
        self._chk(fn(self.h, _ptr(fr_x), _ptr(fr_y), _ptr(t_ns), int(n)))
        self._pending_n = getattr(self, "_pending_n", []) + [int(n)]

    def upload_ring_async(self, ring_x, ring_y, ring_ts, first, n, t0, ring_noise=None, span_ns=None):
        """Slice = n events from ring index `first` (wrapping) of row / column / uint64 timestamp ring arrays (pinned for
        a true DMA; pageable arrays work too); times become ts - t0 on the device.  int32 addresses go through
        bf_upload_ring_async, uint16 addresses through bf_upload_ring16_async; ring_noise: optional uint8 Event::noise ring.

        A uint32 `ring_ts` holds the LOW 32 bits of the timestamps (bf_upload_ring16t32_async, 8 bytes per event): the device
        can only form (int32)(ts32 - (uint32)t0), which is the true difference while |ts - t0| < 2^31 ns (2.1 s) and a
        plausible-looking WRONG time beyond -- the 64-bit forms mark such an event and bf_set_cloud reports it, this form
        cannot.  So the caller, who has the full timestamps, must state the slice's span: `span_ns` = max |ts - t0| over the
        slice (required with uint32 timestamps); 2^31 or more is refused here."""
        assert ring_x.dtype == ring_y.dtype and ring_x.dtype in (np.int32, np.uint16) and ring_ts.dtype in (np.uint64, np.uint32)
        assert ring_noise is None or ring_noise.dtype == np.uint8
        if ring_ts.dtype == np.uint32:   # the low 32 bits of the timestamps: 8 bytes per event (bf_upload_ring16t32_async)
            assert ring_x.dtype == np.uint16
            if span_ns is None:
                raise ValueError("uint32 timestamps: pass span_ns = max |timestamp - t0| of the slice (from the full timestamps)")
            if not (0 <= int(span_ns) < (1 << 31)):
                raise BfError(BF_ERR_ARG, "a slice that reaches %d ns from its start does not fit 32-bit local times (limit 2^31 ns)" % int(span_ns))
            fn = self.L.bf_upload_ring16t32_async
        else:
            fn = self.L.bf_upload_ring_async if ring_x.dtype == np.int32 else self.L.bf_upload_ring16_async
        self._chk(fn(self.h, _ptr(ring_x), _ptr(ring_y), _ptr(ring_ts), None if ring_noise is None else _ptr(ring_noise),
                     int(len(ring_ts)), int(first), int(n), int(t0)))
        self._pending_n = getattr(self, "_pending_n", []) + [int(n)]

    def compute_uv_ring(self, uv_ring, first):
        """Per-event (u, v) of the current slice as interleaved pairs into uv_ring (float64, 2 * cap), event i at ring
        index (first + i) % cap (bf_compute_uv_ring)."""
        assert uv_ring.dtype == np.float64 and uv_ring.size % 2 == 0
        self._chk(self.L.bf_compute_uv_ring(self.h, _ptr(uv_ring), int(uv_ring.size // 2), int(first)))

    def wait_uploads(self):
        """Block until every asynchronous upload issued so far has been copied (bf_wait_uploads)."""
        self._chk(self.L.bf_wait_uploads(self.h))

    def commit_upload(self):
        self._chk(self.L.bf_commit_upload(self.h))
        self.n = self._pending_n.pop(0)

    def to_device(self, arr):
        """Copy a numpy array into a fresh device buffer; returns the device pointer."""
        arr = np.ascontiguousarray(arr)
        d = C.c_void_p()
        self._chk(self.L.bf_device_malloc(self.h, max(arr.nbytes, 16), C.byref(d)))
        if arr.nbytes:
            self._chk(self.L.bf_memcpy_h2d(self.h, d, _ptr(arr), arr.nbytes))
        return d

    def device_free(self, d):
        self._chk(self.L.bf_device_free(self.h, d))

    def eval_sincos(self, x, table=False):
        """The device loops' sin / cos (bf_eval_sincos) of the float64 array x -> (sin, cos)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        sn, cs = np.empty_like(x), np.empty_like(x)
        self._chk(self.L.bf_eval_sincos(self.h, x.ctypes.data, x.size, 1 if table else 0, sn.ctypes.data, cs.ctypes.data))
        return sn, cs

    def copy_bandwidth(self, nbytes=1 << 30, reps=5):
        g = C.c_double(0)
        self._chk(self.L.bf_copy_bandwidth(self.h, nbytes, reps, C.byref(g)))
        return g.value
