import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    oracle.build()
    return oracle


def _ensure_product_built():
    """A checkout without build artefacts (they are git-ignored): build the product once -- hipcc for the kernels,
    g++ for the command line -- exactly what __graft_entry__.build() does.  Building is not a fallback."""
    from better_flow_amd import accel
    cli = os.path.join(ROOT, "better_flow_amd", "host", "bf_motion_compensator")
    if not (os.path.exists(accel.LIB_PATH) and os.path.exists(cli)):
        import __graft_entry__
        __graft_entry__.build()


def pytest_collection_finish(session):
    if any(item.get_closest_marker("gpu") for item in session.items):
        _ensure_product_built()


@pytest.fixture(scope="session")
def accel_mod():
    """The HIP path.  No skip and no fallback: a missing library / device is a failure."""
    from better_flow_amd import accel
    _ensure_product_built()
    accel.load()
    assert accel.device_count() > 0, "gpu-marked test without a HIP device"
    return accel
