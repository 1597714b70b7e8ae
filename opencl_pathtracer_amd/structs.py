"""numpy views of the scene structs the integrator consumes.

Byte-for-byte the layout of ``include/ptmi_scene.h`` (= the reference's
``Controleur/PathTracer_Structs.h:16-128`` under MSVC x64 / the device mirror in
``Kernel/PathTracer_FullKernel_header.cl:89-223``).  Field names are the
reference's own so scene code reads like the reference's importers.
"""
import numpy as np

FLOAT4 = np.dtype((np.float32, 4))
FLOAT2 = np.dtype((np.float32, 2))

BoundingBox = np.dtype({
    "names": ["pMin", "pMax", "centroid", "isEmpty"],
    "formats": [FLOAT4, FLOAT4, FLOAT4, np.int8],
    "offsets": [0, 16, 32, 48],
    "itemsize": 64,
})

Light = np.dtype({
    "names": ["position", "direction", "color", "power", "cosOfInnerFallOffAngle", "cosOfOuterFallOffAngle", "type"],
    "formats": [FLOAT4, FLOAT4, FLOAT4, np.float32, np.float32, np.float32, np.int32],
    "offsets": [0, 16, 32, 48, 52, 56, 60],
    "itemsize": 64,
})

Material = np.dtype({
    "names": ["simpleColor", "textureName", "opacity", "textureId", "type", "isSimpleColor", "hasAlphaMap"],
    "formats": [FLOAT4, np.uint64, np.float32, np.int32, np.int32, np.uint8, np.uint8],
    "offsets": [0, 16, 24, 28, 32, 36, 37],
    "itemsize": 48,
})

Node = np.dtype({
    "names": ["trianglesAABB", "centroidsAABB", "cutAxis", "triangleStartIndex", "nbTriangles", "son1Id", "son2Id",
              "comments", "isLeaf"],
    "formats": [BoundingBox, BoundingBox, np.uint32, np.uint32, np.uint32, np.uint32, np.uint32, np.int32, np.int8],
    "offsets": [0, 64, 128, 132, 136, 140, 144, 148, 152],
    "itemsize": 160,
})

Texture = np.dtype({
    "names": ["width", "height", "offset"],
    "formats": [np.uint32, np.uint32, np.uint32],
    "offsets": [0, 4, 8],
    "itemsize": 12,
})

Triangle = np.dtype({
    "names": ["S1", "S2", "S3", "N1", "N2", "N3", "T1", "T2", "T3", "BT1", "BT2", "BT3", "N",
              "UVP1", "UVP2", "UVP3", "UVN1", "UVN2", "UVN3", "AABB",
              "materialWithPositiveNormalIndex", "materialWithNegativeNormalIndex", "id"],
    "formats": [FLOAT4] * 13 + [FLOAT2] * 6 + [BoundingBox, np.uint32, np.uint32, np.uint32],
    "offsets": [0, 16, 32, 48, 64, 80, 96, 112, 128, 144, 160, 176, 192,
                208, 216, 224, 232, 240, 248, 256, 320, 324, 328],
    "itemsize": 336,
})

Sky = np.dtype({
    "names": ["skyTextures", "groundScale", "exposantFactorX", "exposantFactorY", "cosRotationAngle",
              "sinRotationAngle"],
    "formats": [(Texture, 6), np.float32, np.float32, np.float32, np.float32, np.float32],
    "offsets": [0, 72, 76, 80, 84, 88],
    "itemsize": 92,
})

Uchar4 = np.dtype((np.uint8, 4))

# enums (PathTracer_Structs.h:24-30, 43-51, 66-71, 130-135)
LIGHT_DIRECTIONNAL, LIGHT_POINT, LIGHT_SPOT, LIGHT_UNKNOWN = 0, 1, 2, 3
MAT_STANDART, MAT_WATER, MAT_GLASS, MAT_VARNHISHED, MAT_METAL, MAT_UNKNOWN = 0, 1, 2, 3, 4, 5
NODE_BAD_SAH, NODE_LEAF_MAX_SIZE, NODE_LEAF_MIN_DIAG = 0, 1, 2
JITTERED, RANDOM, UNIFORM = 0, 1, 2
SAMPLER_NAMES = {JITTERED: "JITTERED", RANDOM: "RANDOM", UNIFORM: "UNIFORM"}

MAX_INTERSETCION_NUMBER = 5000  # PathTracer_PreProc.h:18
BVH_MAX_DEPTH = 30              # PathTracer_PreProc.h:19
MAX_LIGHT_SIZE = 30             # PathTracer_PreProc.h:20

assert BoundingBox.itemsize == 64 and Light.itemsize == 64 and Material.itemsize == 48
assert Node.itemsize == 160 and Texture.itemsize == 12 and Triangle.itemsize == 336 and Sky.itemsize == 92
