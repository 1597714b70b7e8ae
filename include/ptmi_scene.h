/*
 * ptmi_scene.h - byte layout of the scene arrays the integrator consumes.
 *
 * This is the INPUT CONTRACT of the hot path: the arrays a caller of the
 * reference hands to OpenCL_InitializeMemory (Controleur/PathTracer_OpenCL.cpp:149-198)
 * are raw dumps of the structs in Controleur/PathTracer_Structs.h; the device
 * mirrors are in Kernel/PathTracer_FullKernel_header.cl:89-223.  The structs
 * below restate that layout (MSVC x64, __declspec(align(16))) in portable C so
 * that a pointer to a reference array can be passed to ptmi_* unchanged.
 * Field comments give the reference field name.  Every size and offset is
 * pinned by a static assertion (values measured from both the .h and the .cl
 * compiled with clang).
 *
 * Plain C99 / C++11, no dependencies.
 */
#ifndef PTMI_SCENE_H
#define PTMI_SCENE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#define PTMI_STATIC_ASSERT(c, m) static_assert(c, m)
#define PTMI_ALIGNAS(n) alignas(n)
extern "C" {
#else
#define PTMI_STATIC_ASSERT(c, m) _Static_assert(c, m)
#define PTMI_ALIGNAS(n) _Alignas(n)
#endif

/* Float4 / RGBAColor  (PathTracer_Utils.h:69-105; OpenCL float4) */
typedef struct ptmi_float4 { float x, y, z, w; } ptmi_float4;
/* Float2 (PathTracer_Utils.h:40-65; OpenCL float2) */
typedef struct ptmi_float2 { float x, y; } ptmi_float2;
/* Uchar4 texel (PathTracer_Utils.h:118-125) */
typedef struct ptmi_uchar4 { uint8_t x, y, z, w; } ptmi_uchar4;

/* BoundingBox (PathTracer_Structs.h:16-22 / header.cl:109-115) : 64 bytes */
typedef struct ptmi_bounding_box {
    PTMI_ALIGNAS(16) ptmi_float4 p_min; /* pMin     @0  */
    ptmi_float4 p_max;                  /* pMax     @16 */
    ptmi_float4 centroid;               /* centroid @32 */
    int8_t is_empty;                    /* isEmpty  @48 (char) */
} ptmi_bounding_box;

/* LightType (PathTracer_Structs.h:24-30) */
enum { PTMI_LIGHT_DIRECTIONNAL = 0, PTMI_LIGHT_POINT = 1, PTMI_LIGHT_SPOT = 2, PTMI_LIGHT_UNKNOWN = 3 };

/* Light (PathTracer_Structs.h:32-41 / header.cl:148-157) : 64 bytes */
typedef struct ptmi_light {
    PTMI_ALIGNAS(16) ptmi_float4 position; /* @0  */
    ptmi_float4 direction;                 /* @16 */
    ptmi_float4 color;                     /* @32 */
    float power;                           /* @48 */
    float cos_inner;                       /* cosOfInnerFallOffAngle @52 */
    float cos_outer;                       /* cosOfOuterFallOffAngle @56 */
    int32_t type;                          /* LightType @60 */
} ptmi_light;

/* MaterialType (PathTracer_Structs.h:43-51) */
enum { PTMI_MAT_STANDART = 0, PTMI_MAT_WATER = 1, PTMI_MAT_GLASS = 2, PTMI_MAT_VARNHISHED = 3, PTMI_MAT_METAL = 4, PTMI_MAT_UNKNOWN = 5 };

/* Material (PathTracer_Structs.h:54-63 / header.cl:188-197) : 48 bytes.
 * texture_name is a HOST pointer in the reference; it is never dereferenced
 * by the integrator and is kept only so the layout agrees. */
typedef struct ptmi_material {
    PTMI_ALIGNAS(16) ptmi_float4 simple_color; /* simpleColor @0 */
    uint64_t texture_name;                     /* char const* textureName @16 */
    float opacity;                             /* @24 */
    int32_t texture_id;                        /* textureId @28 */
    int32_t type;                              /* MaterialType @32 */
    uint8_t is_simple_color;                   /* bool isSimpleColor @36 */
    uint8_t has_alpha_map;                     /* bool hasAlphaMap @37 */
} ptmi_material;

/* NodeStopType (PathTracer_Structs.h:66-71) */
enum { PTMI_NODE_BAD_SAH = 0, PTMI_NODE_LEAF_MAX_SIZE = 1, PTMI_NODE_LEAF_MIN_DIAG = 2 };

/* Node (PathTracer_Structs.h:74-85 / header.cl:125-136) : 160 bytes */
typedef struct ptmi_node {
    ptmi_bounding_box triangles_aabb; /* trianglesAABB @0  */
    ptmi_bounding_box centroids_aabb; /* centroidsAABB @64 */
    uint32_t cut_axis;                /* cutAxis @128 */
    uint32_t triangle_start_index;    /* triangleStartIndex @132 */
    uint32_t nb_triangles;            /* nbTriangles @136 */
    uint32_t son1_id;                 /* son1Id @140 */
    uint32_t son2_id;                 /* son2Id @144 */
    int32_t comments;                 /* NodeStopType @148 */
    int8_t is_leaf;                   /* isLeaf @152 (char) */
} ptmi_node;

/* Texture (PathTracer_Structs.h:96-101 / header.cl:159-164) : 12 bytes */
typedef struct ptmi_texture {
    uint32_t width, height;
    uint32_t offset; /* in texels into texturesData */
} ptmi_texture;

/* Triangle (PathTracer_Structs.h:103-117 / header.cl:210-223) : 336 bytes */
typedef struct ptmi_triangle {
    PTMI_ALIGNAS(16) ptmi_float4 s1; /* S1 @0  */
    ptmi_float4 s2, s3;              /* @16 @32 */
    ptmi_float4 n1, n2, n3;          /* vertex normals @48 @64 @80 */
    ptmi_float4 t1, t2, t3;          /* tangents @96.. (unused by the integrator) */
    ptmi_float4 bt1, bt2, bt3;       /* bitangents @144.. (unused) */
    ptmi_float4 n;                   /* N geometric normal @192 */
    ptmi_float2 uvp1, uvp2, uvp3;    /* UVP1..3 @208 @216 @224 */
    ptmi_float2 uvn1, uvn2, uvn3;    /* UVN1..3 @232 @240 @248 */
    ptmi_bounding_box aabb;          /* AABB @256 */
    uint32_t mat_pos;                /* materialWithPositiveNormalIndex @320 */
    uint32_t mat_neg;                /* materialWithNegativeNormalIndex @324 */
    uint32_t id;                     /* @328 */
} ptmi_triangle;

/* Sky (PathTracer_Structs.h:120-128 / header.cl:200-208) : 92 bytes */
typedef struct ptmi_sky {
    ptmi_texture sky_textures[6]; /* @0 */
    float ground_scale;           /* @72 */
    float exposant_factor_x;      /* @76 */
    float exposant_factor_y;      /* @80 */
    float cos_rotation_angle;     /* @84 */
    float sin_rotation_angle;     /* @88 */
} ptmi_sky;

/* Sampler (PathTracer_Structs.h:130-135) */
enum { PTMI_SAMPLER_JITTERED = 0, PTMI_SAMPLER_RANDOM = 1, PTMI_SAMPLER_UNIFORM = 2 };

/* limits the reference enforces / bakes in */
#define PTMI_MAX_INTERSECTION_NUMBER 5000 /* MAX_INTERSETCION_NUMBER, PathTracer_PreProc.h:18 / header.cl:14 */
#define PTMI_BVH_MAX_DEPTH 30             /* PathTracer_PreProc.h:19 / header.cl:16 (traversal stack size) */
#define PTMI_MAX_LIGHT_SIZE 30            /* PathTracer_PreProc.h:20 */

PTMI_STATIC_ASSERT(sizeof(ptmi_float4) == 16, "float4");
PTMI_STATIC_ASSERT(sizeof(ptmi_bounding_box) == 64, "BoundingBox size");
PTMI_STATIC_ASSERT(offsetof(ptmi_bounding_box, p_max) == 16 && offsetof(ptmi_bounding_box, centroid) == 32 &&
                   offsetof(ptmi_bounding_box, is_empty) == 48, "BoundingBox offsets");
PTMI_STATIC_ASSERT(sizeof(ptmi_light) == 64 && offsetof(ptmi_light, power) == 48 && offsetof(ptmi_light, type) == 60, "Light");
PTMI_STATIC_ASSERT(sizeof(ptmi_material) == 48 && offsetof(ptmi_material, texture_name) == 16 &&
                   offsetof(ptmi_material, opacity) == 24 && offsetof(ptmi_material, texture_id) == 28 &&
                   offsetof(ptmi_material, type) == 32 && offsetof(ptmi_material, is_simple_color) == 36 &&
                   offsetof(ptmi_material, has_alpha_map) == 37, "Material");
PTMI_STATIC_ASSERT(sizeof(ptmi_node) == 160 && offsetof(ptmi_node, centroids_aabb) == 64 &&
                   offsetof(ptmi_node, cut_axis) == 128 && offsetof(ptmi_node, triangle_start_index) == 132 &&
                   offsetof(ptmi_node, nb_triangles) == 136 && offsetof(ptmi_node, son1_id) == 140 &&
                   offsetof(ptmi_node, son2_id) == 144 && offsetof(ptmi_node, comments) == 148 &&
                   offsetof(ptmi_node, is_leaf) == 152, "Node");
PTMI_STATIC_ASSERT(sizeof(ptmi_texture) == 12, "Texture");
PTMI_STATIC_ASSERT(sizeof(ptmi_triangle) == 336 && offsetof(ptmi_triangle, n1) == 48 && offsetof(ptmi_triangle, t1) == 96 &&
                   offsetof(ptmi_triangle, bt1) == 144 && offsetof(ptmi_triangle, n) == 192 &&
                   offsetof(ptmi_triangle, uvp1) == 208 && offsetof(ptmi_triangle, uvn1) == 232 &&
                   offsetof(ptmi_triangle, aabb) == 256 && offsetof(ptmi_triangle, mat_pos) == 320 &&
                   offsetof(ptmi_triangle, mat_neg) == 324 && offsetof(ptmi_triangle, id) == 328, "Triangle");
PTMI_STATIC_ASSERT(sizeof(ptmi_sky) == 92 && offsetof(ptmi_sky, ground_scale) == 72 &&
                   offsetof(ptmi_sky, cos_rotation_angle) == 84 && offsetof(ptmi_sky, sin_rotation_angle) == 88, "Sky");

#ifdef __cplusplus
}
#endif
#endif /* PTMI_SCENE_H */
