"""The reference's scene-cache format (``Controleur/PathTracer_FileImporter.cpp``): three raw files.

``PathTracerFileImporter::Export`` (:16-64) dumps the importer's arrays with ``fwrite`` and ``Import`` (:68-147)
reads them back; there is no header or version, the byte format IS the in-memory struct layout of
``PathTracer_Structs.h`` (= ``structs.py``):

* ``sizes.pth``       cameraDirection, cameraRight, cameraUp, cameraPosition (4 x Float4, in THAT order), then
                      triangulationSize, lightsSize, materiauxSize, texturesSize, texturesDataSize (5 x uint32)
* ``pointers.pth``    Triangle[] , Light[] , Material[] , Texture[] , Sky   (empty arrays are simply absent)
* ``textureData.pth`` Uchar4[texturesDataSize]

The BVH is not part of the cache: the reference rebuilds it after every import (``PathTracer.cpp:51``), and so does
``import_scene(..., build_bvh=True)``.  ``Material.textureName`` is a host pointer in the dump; it is meaningless after a
reload and never dereferenced by the integrator.
"""
import os

import numpy as np

from . import structs as S
from .scenes import Scene

SIZES, POINTERS, TEXTURE_DATA = "sizes.pth", "pointers.pth", "textureData.pth"


def export_scene(scene, folder):
    """``PathTracerFileImporter::Export`` (:16-64)."""
    os.makedirs(folder, exist_ok=True)
    with open(os.path.join(folder, SIZES), "wb") as f:
        for v in (scene.cameraDirection, scene.cameraRight, scene.cameraUp, scene.cameraPosition):
            f.write(np.asarray(v, np.float32).tobytes())
        f.write(np.array([len(scene.triangulation), len(scene.lights), len(scene.materiaux), len(scene.textures),
                          len(scene.texturesData)], np.uint32).tobytes())
    with open(os.path.join(folder, POINTERS), "wb") as f:
        for a, dt in ((scene.triangulation, S.Triangle), (scene.lights, S.Light), (scene.materiaux, S.Material),
                      (scene.textures, S.Texture)):
            a = np.ascontiguousarray(a)
            assert a.dtype == dt or len(a) == 0
            f.write(a.tobytes())
        f.write(np.ascontiguousarray(scene.sky).tobytes())
    with open(os.path.join(folder, TEXTURE_DATA), "wb") as f:
        f.write(np.ascontiguousarray(scene.texturesData, dtype=np.uint8).tobytes())


def import_scene(folder, load_sky=True, build_bvh=True):
    """``PathTracerFileImporter::Import`` (:68-147); raises like the reference when a file is missing or short."""
    paths = [os.path.join(folder, n) for n in (SIZES, POINTERS, TEXTURE_DATA)]
    missing = [p for p in paths if not os.path.exists(p)]
    if missing:
        raise RuntimeError("Fail to read the files to import : " + ", ".join(missing))
    raw = open(paths[0], "rb").read()
    if len(raw) < 84:
        raise RuntimeError(f"{paths[0]}: truncated ({len(raw)} bytes, 84 expected)")
    cam = np.frombuffer(raw, np.float32, 16).reshape(4, 4).copy()
    n_tri, n_light, n_mat, n_tex, n_texel = (int(x) for x in np.frombuffer(raw, np.uint32, 5, 64))

    raw = open(paths[1], "rb").read()
    need = n_tri * 336 + n_light * 64 + n_mat * 48 + n_tex * 12 + 92
    if len(raw) < need:
        raise RuntimeError(f"{paths[1]}: truncated ({len(raw)} bytes, {need} expected)")
    off = 0

    def take(dtype, n):
        nonlocal off
        # byte-level copy: numpy's .copy() of a padded struct dtype skips the padding bytes
        a = np.frombuffer(bytearray(raw[off:off + n * dtype.itemsize]), dtype, n)
        off += n * dtype.itemsize
        return a

    tris, lights, mats, texs = take(S.Triangle, n_tri), take(S.Light, n_light), take(S.Material, n_mat), take(S.Texture, n_tex)
    sky = take(S.Sky, 1).reshape(())
    texels = np.fromfile(paths[2], np.uint8)
    if len(texels) < 4 * n_texel:
        raise RuntimeError(f"{paths[2]}: truncated")
    texels = texels[:4 * n_texel].reshape(n_texel, 4).copy()

    if not load_sky:  # :131-146
        sky["cosRotationAngle"], sky["sinRotationAngle"] = 1, 0
        sky["exposantFactorX"] = sky["exposantFactorY"] = 0
        sky["groundScale"] = 1
        for i in range(6):
            sky["skyTextures"][i] = (1, 1, 0)
        # the reference shrinks texturesDataSize to 1 and zeroes texel 0; material textures die with it
        texels = np.zeros((1, 4), np.uint8)

    sc = Scene(tris, lights, mats, texs, texels, sky, cameraPosition=cam[3], cameraDirection=cam[0], cameraRight=cam[1],
               cameraUp=cam[2], name=os.path.basename(os.path.normpath(folder)))
    if build_bvh:
        from .backend import bvh_create
        bvh_create(sc)
    return sc
