"""ptmi_validate_scene (host-only) and, through it, the validation + re-layout every upload runs (csrc/scene_layout.cpp):
structural corruptions of valid scenes end in an error code - in the product library, and in a build of the same sources
under AddressSanitizer / UBSan, where a read outside the caller's arrays would abort."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from opencl_pathtracer_amd import PtmiError, backend, bvh_create, scenes, validate_scene, structs as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "sanitize"))
import scene_validation as V  # noqa: E402

pytestmark = pytest.mark.filterwarnings("ignore::RuntimeWarning")  # (the hostile scenes divide 0 by 0 on purpose, as the importer would)
MUST_FAIL = {"son_out_of_range", "son_cycle", "leaf_range", "leaf_count_huge", "cut_axis", "material_index", "texture_id", "sky_texture",
             "uv_huge", "uv_nan", "lights_mismatch", "texture_zero_size"}


def test_valid_scenes_pass_and_messages_name_the_record(built):
    sc = bvh_create(scenes.build("matmix", 64, 64))
    validate_scene(sc, 64, 64, 8)
    sc.triangulation["materialWithPositiveNormalIndex"][5] = 77
    with pytest.raises(PtmiError, match="triangle 5 references a material out of range") as e:
        validate_scene(sc, 64, 64, 8)
    assert e.value.code == -5
    hostile = bvh_create(scenes.build("fuzz5h_l1", 64, 64))
    validate_scene(hostile, 64, 64, 8)  # NaN-distance records: the wavefront kernel's NANSAFE instantiation - still a valid scene,
    validate_scene(hostile, 64, 64, 8, super_sampling=True)  # with SUPER_SAMPLING too (round 4) ...
    with pytest.raises(PtmiError, match="SUPER_SAMPLING") as e:  # ... unless the sampler is RANDOM (nothing staged: one-path-per-lane kernel)
        validate_scene(hostile, 64, 64, 8, sampler=1, super_sampling=True)
    assert e.value.code == -7


def test_structural_corruptions_are_error_codes(built):
    out = V.run(backend.load_library(), seeds=range(0, 10))
    for kind in V.CORRUPTIONS:
        assert kind in out, kind
    for kind in MUST_FAIL:
        assert 0 not in out[kind], (kind, out[kind])  # every instance of these is refused
    assert all(rc <= 0 for rcs in out.values() for rc in rcs)


def test_validation_under_sanitizers(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not installed")
    asan = subprocess.run([gxx, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan = subprocess.run([gxx, "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not (os.path.isabs(asan) and os.path.exists(asan) and os.path.isabs(ubsan) and os.path.exists(ubsan)):
        pytest.skip("libasan / libubsan not installed")
    csrc = os.path.join(ROOT, "opencl_pathtracer_amd", "csrc")
    stub = tmp_path / "stub.cpp"
    stub.write_text('#include <string>\nnamespace ptmi_internal { void set_global_error(const std::string&) {} }\n')
    so = tmp_path / "libscene_asan.so"
    r = subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                        "-fno-sanitize-recover=undefined", "-I" + os.path.join(ROOT, "include"), "-I" + csrc, os.path.join(csrc, "scene_layout.cpp"),
                        os.path.join(csrc, "bvh_build.cpp"), str(stub), "-o", str(so)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    env = {**os.environ, "LD_PRELOAD": asan + ":" + ubsan, "ASAN_OPTIONS": "detect_leaks=0"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sanitize", "scene_validation.py"), str(so)], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "clean" in r.stdout, (r.stdout[-800:], r.stderr[-4000:])
