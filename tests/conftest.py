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
def built():
    """Make sure the product library and the CPU oracle exist (both build without a GPU)."""
    import subprocess
    from opencl_pathtracer_amd import backend
    if not os.path.exists(backend.library_path()):
        subprocess.run(["make", "-s", "-C", ROOT, "lib"], check=True)
    import oracle_ffi
    if not os.path.exists(oracle_ffi.ORACLE_LIB):
        oracle_ffi.build_oracle()
    return True


_scene_cache = {}


@pytest.fixture(scope="session")
def scene_factory(built):
    """scene_factory(name, w, h) -> Scene with its BVH built by the product builder (cached)."""
    from opencl_pathtracer_amd import scenes, bvh_create

    def make(name, w, h):
        key = (name, w, h)
        if key not in _scene_cache:
            _scene_cache[key] = bvh_create(scenes.build(name, w, h))
        return _scene_cache[key]

    return make
