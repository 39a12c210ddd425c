"""Shared by the GPU tests: contexts built from an options dict.

Besides the keys of bf_set_option the dict may hold "debug_margin": the test hook BF_DEBUG_MARGIN (read once, at
bf_create), which shrinks the margin of the tile-binned loops from its constant 8 scaled pixels so that events outrun their bins
-- the overflow path, the `lost` flag, re-bins and repeated passes -- within a few iterations.  (It was an option until round 5.)
The hooks exist only in the TEST build of the library (better_flow_amd/debug/libbf_accel.so, `make debug`: the release objects
with bf_context.cpp / bf_run.cpp compiled -DBF_DEBUG_HOOKS); the release library reads no BF_DEBUG_* variable.  A context that
needs a hook is created from that build (`debug_accel`); both builds can live in one process."""
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
    env = {k: options.pop(k) for k in list(options) if k.startswith("BF_DEBUG_")}   # (other hooks, by their variable's name)
    margin = options.pop("debug_margin", None)
    if margin is not None:
        env["BF_DEBUG_MARGIN"] = margin
    a = debug_accel(accel_mod, env, **kw) if env else accel_mod.Accel(**kw)
    for k, v in options.items():
        a.set_option(k, v)
    return a


def debug_accel(accel_mod, _env=None, **kw):
    """accel.Accel(**kw) from the TEST build of the library, created under the given BF_DEBUG_* variables
    (upper-case keyword arguments, or the dict `_env`)."""
    env = dict(_env or {})
    env.update({k: kw.pop(k) for k in list(kw) if k.startswith("BF_")})
    with debug_env(**env):
        return accel_mod.Accel(lib=accel_mod.DEBUG_LIB_PATH, **kw)


def debug_cli_env(env):
    """`env` for a subprocess of the command line that is to load the TEST build: its directory ahead of the binary's RUNPATH."""
    import os as _os
    env = dict(env)
    d = _os.path.dirname(_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "better_flow_amd", "debug", "x"))
    env["LD_LIBRARY_PATH"] = d + (":" + env["LD_LIBRARY_PATH"] if env.get("LD_LIBRARY_PATH") else "")
    return env
