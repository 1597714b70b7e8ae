import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _hip_device_present():
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    """Under `-m gpu` on a box with a GPU the reference-derived checkers (oracle/_ref: the reference kernel's code objects,
    its builder, the launcher) are part of the test suite: without them every HIP-vs-reference comparison would vanish and
    the run would still be green.  So their absence ends the session with an error, before any test runs."""
    if not any("gpu" in item.keywords for item in items) or not _hip_device_present():
        return
    # torch's HIP runtime first: a few tests ask torch for device properties / free memory, and its runtime does not find the
    # device when the library's (ctypes-loaded) one has initialised before it - which made those tests depend on which tests
    # ran earlier in the session
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:  # noqa: BLE001  (no torch: those tests say so themselves)
        pass
    import oracle_ffi
    if oracle_ffi.ALLOW_MISSING_REFERENCE:
        return
    missing = [os.path.relpath(p, ROOT) for p in oracle_ffi.configured_ref_objects() if not os.path.exists(p)]
    if missing:
        pytest.exit("reference-derived checkers are missing on a GPU box: " + ", ".join(missing[:6]) +
                    (f" (+{len(missing) - 6} more)" if len(missing) > 6 else "") +
                    " - build them where the reference tree exists (make -C oracle ref) and ship oracle/_ref/ with the "
                    "repository, or set PTMI_ALLOW_MISSING_REFERENCE=1", returncode=3)


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
