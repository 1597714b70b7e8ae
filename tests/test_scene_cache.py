"""Scene-cache files in the reference's format (PathTracer_FileImporter.cpp:16-147): byte layout and round trip."""
import os
import struct

import numpy as np
import pytest

import oracle_ffi as O
from opencl_pathtracer_amd import scenes, scene_cache, bvh_create, structs as S


def test_file_layout_is_the_reference_format(tmp_path, built):
    sc = scenes.material_mix(32, 32)
    scene_cache.export_scene(sc, str(tmp_path))
    sizes = open(tmp_path / "sizes.pth", "rb").read()
    assert len(sizes) == 4 * 16 + 5 * 4
    # order in the file: direction, right, up, position (FileImporter.cpp:27-30), then the five counts (:32-36)
    assert np.array_equal(np.frombuffer(sizes, np.float32, 4, 0), sc.cameraDirection)
    assert np.array_equal(np.frombuffer(sizes, np.float32, 4, 48), sc.cameraPosition)
    assert struct.unpack_from("<5I", sizes, 64) == (len(sc.triangulation), 3, len(sc.materiaux), 2, len(sc.texturesData))
    ptr = os.path.getsize(tmp_path / "pointers.pth")
    assert ptr == len(sc.triangulation) * 336 + 3 * 64 + len(sc.materiaux) * 48 + 2 * 12 + 92
    assert os.path.getsize(tmp_path / "textureData.pth") == 4 * len(sc.texturesData)
    first = np.frombuffer(open(tmp_path / "pointers.pth", "rb").read(), S.Triangle, 1)[0]
    assert first.tobytes() == sc.triangulation[0].tobytes()


def test_round_trip_renders_identically(tmp_path, built):
    sc = scenes.material_mix(48, 48)
    scene_cache.export_scene(sc, str(tmp_path / "ExportedScene"))
    back = scene_cache.import_scene(str(tmp_path / "ExportedScene"))
    ref = bvh_create(scenes.material_mix(48, 48))
    for name in ("triangulation", "lights", "materiaux", "textures", "texturesData", "bvh"):
        assert np.ascontiguousarray(getattr(back, name)).tobytes() == np.ascontiguousarray(getattr(ref, name)).tobytes(), name
    assert back.sky.tobytes() == ref.sky.tobytes() and back.bvhMaxDepth == ref.bvhMaxDepth
    a = O.oracle_render(back, 48, 48, 5, 2)
    b = O.oracle_render(ref, 48, 48, 5, 2)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2][0], b[2][0])


@pytest.mark.parametrize("name", ["fuzz2_l1", "fuzz7h_l3", "fuzz41hr_l1", "fuzz42r_l3"])
def test_round_trip_of_fuzzed_scenes(name, tmp_path, built):
    """Records no importer writes and NaN normals survive the files byte for byte, and render to the same bits."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sc = scenes.build(name, 40, 40)
        scene_cache.export_scene(sc, str(tmp_path))
        back = scene_cache.import_scene(str(tmp_path))
        ref = bvh_create(scenes.build(name, 40, 40))
    for field in ("triangulation", "lights", "materiaux", "textures", "texturesData", "bvh"):
        assert np.ascontiguousarray(getattr(back, field)).tobytes() == np.ascontiguousarray(getattr(ref, field)).tobytes(), field
    assert back.sky.tobytes() == ref.sky.tobytes()
    for cam in ("cameraPosition", "cameraDirection", "cameraRight", "cameraUp"):
        assert np.asarray(getattr(back, cam), np.float32).tobytes() == np.asarray(getattr(ref, cam), np.float32).tobytes()
    a, b = O.oracle_render(back, 40, 40, 6, 2, default_arithmetic=True), O.oracle_render(ref, 40, 40, 6, 2, default_arithmetic=True)
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and all(np.array_equal(x, y) for x, y in zip(a[2], b[2]))


def test_load_sky_false_and_errors(tmp_path, built):
    sc = scenes.cornell_box(16, 16)
    scene_cache.export_scene(sc, str(tmp_path))
    nosky = scene_cache.import_scene(str(tmp_path), load_sky=False, build_bvh=False)
    assert len(nosky.texturesData) == 1 and not nosky.texturesData.any() and nosky.sky["cosRotationAngle"] == 1
    assert all(tuple(t) == (1, 1, 0) for t in nosky.sky["skyTextures"])
    os.remove(tmp_path / "textureData.pth")
    with pytest.raises(RuntimeError, match="Fail to read the files to import"):
        scene_cache.import_scene(str(tmp_path))
    scene_cache.export_scene(sc, str(tmp_path))
    with open(tmp_path / "pointers.pth", "r+b") as f:
        f.truncate(1000)
    with pytest.raises(RuntimeError, match="truncated"):
        scene_cache.import_scene(str(tmp_path))


def test_sky_from_cross_unpacks_the_six_faces():
    """MayaImporter.cpp:720-817: top face above, ground face below the second of the four side faces; b,g,r -> r,g,b,255."""
    from opencl_pathtracer_amd import scenes
    fw, fh = 3, 2
    img = np.zeros((3 * fh, 4 * fw, 3), np.uint8)
    img[0:fh, fw:2 * fw] = (10, 11, 12)                      # top
    for i in range(4):
        img[fh:2 * fh, i * fw:(i + 1) * fw] = (20 + i, 30 + i, 40 + i)
    img[2 * fh:, fw:2 * fw] = (50, 51, 52)                   # ground
    img[0, 0] = (99, 99, 99)                                 # a corner of the cross that belongs to no face
    sky, texels = scenes.sky_from_cross(img, first_texel=7)
    assert texels.shape == (6 * fw * fh, 4) and (texels[:, 3] == 255).all()
    want = [(12, 11, 10)] + [(40 + i, 30 + i, 20 + i) for i in range(4)] + [(52, 51, 50)]
    for i in range(6):
        assert tuple(sky["skyTextures"][i]) == (fw, fh, 7 + i * fw * fh)
        assert (texels[i * fw * fh:(i + 1) * fw * fh, :3] == want[i]).all()
    assert sky["cosRotationAngle"] == 1 and sky["sinRotationAngle"] == 0 and sky["groundScale"] == 1
    assert not (texels[:, :3] == 99).any()


def test_camera_from_film_scales_and_permutes():
    """SetCam (MayaImporter.cpp:59-101): Right *= 25.4*apertureX/focal, Up *= 25.4*apertureY/focal (doubles), then zxy."""
    from opencl_pathtracer_amd import scenes
    pos, d, r, u = scenes.camera_from_film((1, 2, 3), (0, 0, -1), (0, 1, 0), (1, 0, 0), 35.0, 1.417, 0.945, maya_axes=True)
    assert pos.tolist() == [3, 1, 2, 1] and d.tolist() == [-1, 0, 0, 0]
    assert np.array_equal(r, np.array([0, np.float32(1.417 * 25.4 / 35.0), 0, 0], np.float32))
    assert np.array_equal(u, np.array([0, 0, np.float32(0.945 * 25.4 / 35.0), 0], np.float32))
    pos, d, r, u = scenes.camera_from_film((1, 2, 3), (0, 0, -1), (0, 1, 0), (1, 0, 0), 50.0, 1.0, 1.0)
    assert pos.tolist() == [1, 2, 3, 1] and r[0] == np.float32(25.4 / 50.0) and u[1] == np.float32(25.4 / 50.0)
