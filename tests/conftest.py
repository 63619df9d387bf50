import os
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "sycl-ray-tracer_amd"))
# The HOST's HIP configuration, made before anything in the test process touches HIP: eight hardware queues, so that the per-bounce
# schedules get the three stream lanes they ask for (the library never sets this itself; test_library_loaded_after_hip_... covers a host
# that did not)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def rtlib():
    """The product library. Built in-tree; tests never fall back to anything else."""
    from rtamd import abi
    if not abi.LIB_PATH.exists():
        import __graft_entry__ as g
        g.build()
    return abi.load_library()


@pytest.fixture(scope="session")
def devlib(rtlib):
    """librt_mi355x_dev.so: the -DRT_DEVELOPER_KNOBS build (csrc/rt_knobs.h), the only one whose sweep knobs and test hooks read the
    environment. Tests of those knobs and hooks load it beside the product library (Scene(..., lib=devlib))."""
    from rtamd import abi
    if not abi.DEV_LIB_PATH.exists():
        import __graft_entry__ as g
        g.build()
    return abi.load_developer_library()


@pytest.fixture(scope="session")
def scene_cache():
    from rtamd import scenes
    cache = {}

    def get(name, **kw):
        key = (name, tuple(sorted(kw.items())))
        if key not in cache:
            cache[key] = scenes.get_scene(name, **kw)
        return cache[key]

    return get
