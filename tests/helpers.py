"""Shared by the GPU tests: contexts built from an options dict.

Besides the keys of bf_set_option the dict may hold "debug_margin": the library's test hook BF_DEBUG_MARGIN (read once, at
bf_create), which shrinks the margin of the tile-binned loops from its constant 8 scaled pixels so that events outrun their bins
-- the overflow path, the `lost` flag, re-bins and repeated passes -- within a few iterations.  (It was an option until round 5.)"""
import contextlib
import os


@contextlib.contextmanager
def debug_env(**kv):
    """Environment variables for the duration of the block (None: leave the variable alone)."""
    old = {}
    try:
        for k, v in kv.items():
            if v is None:
                continue
            old[k] = os.environ.get(k)
            os.environ[k] = str(v)
        yield
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def make_accel(accel_mod, options=None, **kw):
    """accel.Accel(**kw) with `options` applied; "debug_margin" goes through the environment of the creation."""
    options = dict(options or {})
    margin = options.pop("debug_margin", None)
    with debug_env(BF_DEBUG_MARGIN=margin):
        a = accel_mod.Accel(**kw)
    for k, v in options.items():
        a.set_option(k, v)
    return a
