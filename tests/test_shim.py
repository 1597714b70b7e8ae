"""The C++ host shim (PathTracerNS::OpenCL_SetupContext / OpenCL_InitializeMemory / OpenCL_RunKernel / BVH_Create
on top of the C ABI) driven the way PathTracer_Main drives the reference backend."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle_ffi as O
from opencl_pathtracer_amd import scenes, structs as S, bvh_create

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "opencl_pathtracer_amd", "lib")
DRIVER = os.path.join(LIB, "shim_driver")
REF = "/root/reference/Controleur"
CLANGXX = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module")
def driver(built):
    if not os.path.exists(DRIVER):
        subprocess.run(["make", "-s", "-C", ROOT, "shim"], check=True)
    return DRIVER


def dump_scene(path, sc, w, h, depth, sampler, n_images):
    with open(path, "wb") as f:
        f.write(struct.pack("<10I", w, h, depth, sampler, n_images, len(sc.triangulation), len(sc.lights),
                            len(sc.materiaux), len(sc.textures), len(sc.texturesData)))
        for v in (sc.cameraPosition, sc.cameraDirection, sc.cameraRight, sc.cameraUp):
            f.write(np.asarray(v, np.float32).tobytes())
        f.write(np.ascontiguousarray(sc.sky).tobytes())
        for a in (sc.triangulation, sc.lights, sc.materiaux, sc.textures, sc.texturesData):
            f.write(np.ascontiguousarray(a).tobytes())


def read_result(path, w, h, depth):
    raw = open(path, "rb").read()
    meta = struct.unpack_from("<3I", raw, 0)
    off = 12
    color = np.frombuffer(raw, np.float32, w * h * 4, off).reshape(h, w, 4); off += w * h * 16
    count = np.frombuffer(raw, np.float32, w * h, off).reshape(h, w); off += w * h * 4
    dep = np.frombuffer(raw, np.uint32, depth + 1, off); off += (depth + 1) * 4
    bbx = np.frombuffer(raw, np.uint32, 5000, off); off += 20000
    tri = np.frombuffer(raw, np.uint32, 5000, off)
    return meta, color, count, dep, bbx, tri


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
def test_mirror_header_layout_equals_reference(tmp_path):
    """include/pathtracer_backend.hpp must be binary-identical to the reference's GlobalVars and scene structs,
    and PathTracer_HIP.cpp must compile against the reference's OWN headers (the maintainer's build)."""
    fields = ["window", "importer", "cameraDirection", "cameraRight", "cameraUp", "cameraPosition", "bvh", "triangulation",
              "lights", "materiaux", "textures", "texturesData", "bvhSize", "triangulationSize", "lightsSize",
              "materiauxSize", "texturesSize", "texturesDataSize", "bvhMaxDepth", "sampler", "printLogInfos",
              "superSampling", "sky", "imageWidth", "imageHeight", "imageSize", "rayMaxDepth", "imageColor", "imageRayNb",
              "rayDepths", "rayIntersectedBBx", "rayIntersectedTri"]
    checks = "\n".join(f'static_assert(offsetof(PathTracerNS::GlobalVars, {f}) == offsetof(PtmiMirrorNS::GlobalVars, {f}), "{f}");'
                       for f in fields)
    src = tmp_path / "layout.cpp"
    src.write_text('#include <cstddef>\n#include <climits>\n#include <cstring>\n#include "PathTracer_Structs.h"\n'
                   '#define PathTracerNS PtmiMirrorNS\n#include "pathtracer_backend.hpp"\n#undef PathTracerNS\n'
                   'static_assert(sizeof(PathTracerNS::GlobalVars) == sizeof(PtmiMirrorNS::GlobalVars), "GlobalVars size");\n'
                   'static_assert(sizeof(PathTracerNS::Triangle) == sizeof(ptmi_triangle) && sizeof(PathTracerNS::Node) == sizeof(ptmi_node), "scene structs");\n'
                   'static_assert(sizeof(PathTracerNS::Material) == sizeof(ptmi_material) && sizeof(PathTracerNS::Light) == sizeof(ptmi_light), "scene structs");\n'
                   'static_assert(sizeof(PathTracerNS::Sky) == sizeof(ptmi_sky) && sizeof(PathTracerNS::Texture) == sizeof(ptmi_texture), "scene structs");\n'
                   + checks + "\nint main(){return 0;}\n")
    subprocess.run([CLANGXX, "-std=c++14", "-fms-extensions", "-fsyntax-only", "-I", REF, "-I", os.path.join(ROOT, "include"),
                    str(src)], check=True)
    subprocess.run([CLANGXX, "-std=c++14", "-fms-extensions", "-fsyntax-only", "-DPTMI_USE_REFERENCE_HEADERS", "-include", "climits",
                    "-include", "cstring", "-I", REF, "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "opencl_pathtracer_amd", "csrc", "PathTracer_HIP.cpp")], check=True)


def test_shim_reports_errors_as_exceptions(driver, tmp_path):
    """No device (here) or a broken scene: std::runtime_error, caught by the caller like PathTracer.cpp:99-107."""
    from opencl_pathtracer_amd import backend
    sc = scenes.cornell_box(16, 12)
    scene_file, out_file = str(tmp_path / "s.bin"), str(tmp_path / "o.bin")
    if backend.load_library().ptmi_device_count() == 0:
        dump_scene(scene_file, sc, 16, 12, 2, S.JITTERED, 1)
        r = subprocess.run([driver, scene_file, out_file], capture_output=True, text=True)
        assert r.returncode == 1 and "no HIP device" in r.stderr and not os.path.exists(out_file)
    sc.triangulation["materialWithPositiveNormalIndex"][0] = 77  # out of range: must be an error on any box
    dump_scene(scene_file, sc, 16, 12, 2, S.JITTERED, 1)
    r = subprocess.run([driver, scene_file, out_file], capture_output=True, text=True)
    assert r.returncode == 1 and "exception:" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,h,d,sampler,n", [("cornell", 64, 48, 4, S.JITTERED, 5), ("matmix", 96, 96, 8, S.UNIFORM, 3)])
def test_shim_end_to_end_matches_oracle(name, w, h, d, sampler, n, driver, tmp_path):
    sc = scenes.build(name, w, h)  # no BVH: the driver calls BVH_Create itself
    scene_file, out_file = str(tmp_path / "s.bin"), str(tmp_path / "o.bin")
    dump_scene(scene_file, sc, w, h, d, sampler, n)
    r = subprocess.run([driver, scene_file, out_file], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    (callbacks, bvh_size, bvh_depth), color, count, dep, bbx, tri = read_result(out_file, w, h, d)
    ref = bvh_create(scenes.build(name, w, h))
    assert callbacks == n and bvh_size == len(ref.bvh) and bvh_depth == ref.bvhMaxDepth
    o_color, o_count, (o_dep, o_bbx, o_tri), _ = O.oracle_render(ref, w, h, d, n, sampler=sampler)
    assert np.array_equal(color.view(np.uint32), o_color.view(np.uint32)) and np.array_equal(count, o_count)
    assert np.array_equal(dep, o_dep) and np.array_equal(bbx, o_bbx) and np.array_equal(tri, o_tri)
