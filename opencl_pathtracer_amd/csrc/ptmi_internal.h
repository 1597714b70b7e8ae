// ptmi_internal.h - shared between the host side of libptmi.so and its HIP kernels.
// Not part of the ABI.
#pragma once

#include <cstdint>
#include <string>

#include "ptmi.h"

namespace ptmi_internal {

void set_global_error(const std::string& msg);

// ---- device-side scene layout (see DESIGN.md "Data layout in HBM") ----------
//
// The reference walks 160-byte Node records and copies 336-byte Triangles into
// private memory for every test (FullKernel.cl:640,671-672).  At upload the
// scene is re-laid out so that one traversal step touches one aligned 64-byte
// record and nothing it does not need:
//
//  DNode (64 B, one per INNER node): both children's boxes + child references.
//  DTri / DTriPre (64 B, one per triangle of a leaf): what the intersection test reads.
//  DShade (112 B, one per input triangle): what only a confirmed surface hit reads.
//  DMat  (32 B): material without the host pointer.
//
// DNode and DTri records share ONE array in depth-first order (a node, the triangles of its leaf children, son1's
// subtree, son2's subtree), so DScene::nodes == DScene::tris and every index below is an index of 64-byte records;
// tri_ids[] maps a triangle record back to its input triangle.
//
// A child reference ("ref") packs what the traversal needs to know about the
// child without touching it:
//   bit 31      : child is a leaf
//   bit 30      : child's trianglesAABB.isEmpty (box test returns false, FullKernel.cl:68)
//   inner child : bits 29..0 = record index of its DNode (< 2^27)
//   leaf child  : bits 29..27 = triangle count 0..6, or 7 = "big leaf";
//                 bits 26..0  = record index of its first triangle (count<=6) or index into big_leaves[]
constexpr uint32_t REF_LEAF = 0x80000000u;
constexpr uint32_t REF_EMPTY = 0x40000000u;
constexpr uint32_t REF_COUNT_SHIFT = 27;
constexpr uint32_t REF_COUNT_BIG = 7;
constexpr uint32_t REF_INDEX_MASK_INNER = 0x3FFFFFFFu;
constexpr uint32_t REF_INDEX_MASK_LEAF = 0x07FFFFFFu;
constexpr uint32_t REF_NONE = 0xFFFFFFFFu;  // traversal finished (never a valid ref: leaf+empty+count 7 + all ones)
constexpr uint32_t REF_IDLE = 0xFFFFFFFEu;  // wavefront kernel: the lane has no path in flight
constexpr uint32_t REF_DEAD = 0xFFFFFFFDu;  // wavefront kernel: the job queue is empty, the lane is done

struct DNode {
    float lo1[3], hi1[3];  // son1Id's trianglesAABB pMin/pMax xyz
    float lo2[3], hi2[3];  // son2Id's
    uint32_t ref1, ref2;
    uint32_t axis;  // cutAxis
    uint32_t pad;
};
static_assert(sizeof(DNode) == 64, "DNode");

struct DTri {
    float s1[4], s2[4], s3[4], n[4];
};
static_assert(sizeof(DTri) == 64, "DTri");

// Same 64 bytes, ray-independent part of Triangle_Intersects (FullKernel.cl:528-556) done once at upload:
//   n = N;  s1d = (S1.xyz, d = dot(N,S1));  u_den = ((S2-S1).xyz, 1/(uv*uv-uu*vv));  v_s1w = ((S3-S1).xyz, S1.w)
// Valid when S1.w == S2.w == S3.w (every importer writes w = 1), so that (S2-S1).w = (S3-S1).w = +0 exactly.
// The host evaluates these with the same correctly rounded operations the kernel would (fmaf, no contraction),
// so the test's results are bit-identical to the generic DTri form.
struct DTriPre {
    float n[4], s1d[4], u_den[4], v_s1w[4];
};
static_assert(sizeof(DTriPre) == sizeof(DTri), "DTriPre");

struct DShade {
    float n1[4], n2[4], n3[4];
    float uvp[6];  // UVP1.xy UVP2.xy UVP3.xy
    float uvn[6];
    uint32_t mat_pos, mat_neg;
    uint32_t pad[2];
};
static_assert(sizeof(DShade) == 112, "DShade");

struct DMat {
    float color[4];
    float opacity;
    int32_t texture_id;
    int32_t type;
    uint32_t is_simple_color;
};
static_assert(sizeof(DMat) == 32, "DMat");

struct DBigLeaf {
    uint32_t start, count;
};

// calls one launch may render AHEAD for (ptmi_api.cpp: render_on_device): the launch keeps the totals of ptmi_get_counters per
// call, in as many blocks of C_COUNT words
#define PTMI_COUNTER_SPLITS 4

enum CounterSlot {
    C_PATHS = 0, C_SEGMENTS, C_HITS, C_SHADOW, C_BBX, C_TRI,
    // wave scheduler of the wavefront kernel: loop trips and active lanes per step kind
    C_TRIPS_I, C_LANES_I, C_TRIPS_T, C_LANES_T, C_TRIPS_P, C_LANES_P,
    // ... and where its waves spend their life: shader clock cycles inside path-logic passes / in the whole main loop
    C_CYCLES_P, C_CYCLES_LOOP,
    // leaf passes (STATS builds): items a lane read with an owner >= 64 or a record index >= n_records - the invariant of the
    // item protocol (every item below the pass's count is written by its owner before any lane reads it) would be broken
    C_ITEM_VIOLATIONS,
    // the reference's device-side checks (-D LOG_INFO: ASSERT / WARNING of header.cl:21-48), counted instead of printed, in STATS builds:
    C_CHK_SAMPLE,      // cl:1217  the sample position lies in [-0.5, 0.5]^2
    C_CHK_NORMALS,     // cl:1275  geometric and shading normal both face the arriving ray
    C_CHK_RADIANCE,    // cl:951   the light gathered at a hit is non-negative
    C_CHK_HEMISPHERE,  // h:243    the scattered direction lies in the hemisphere of its normal after Vector_PutInSameHemisphereAs
    C_CHK_STATS_RANGE, // cl:1325,1330  a path's box / triangle test count fits the 5000-bin histograms
    C_UNDEF_REFRACTION, // cl:836-843 after :237  a totally reflected ray refracted (random() == 1.0): undefined in the reference's source
    // surface hits on a material with a file texture (STATS builds): the texel term of SURVEY 8d's algorithmic bytes
    C_TEXTURED_HITS,
    // paths a wavefront launch gave up (a ray that is not a number) and the literal loops traced again (every build)
    C_RETRACED,
    C_COUNT
};

// Everything a launch needs, passed by value (lives in SGPRs / kernarg).
struct DScene {
    const DNode* nodes;
    const DTri* tris;
    const DShade* shade;
    const DMat* mats;
    const ptmi_light* lights;
    const ptmi_texture* textures;
    const ptmi_uchar4* texels;
    const DBigLeaf* big_leaves;
    const uint32_t* tri_ids;  // record index -> triangle index (shade[], triangulation[])
    float* image_color;   // float4[W*H]
    float* image_ray_nb;  // float[W*H]
    uint32_t* hist_depths;  // [D+1]   (nullptr = histograms off)
    uint32_t* hist_bbx;     // [5000]
    uint32_t* hist_tri;     // [5000]
    unsigned long long* counters;  // [C_COUNT]; [PTMI_COUNTER_SPLITS][C_COUNT] where split_paths != 0
    uint32_t split_paths;          // 0, or: the launch renders for several calls, staging slots [k * split_paths, (k + 1) * split_paths) are call k's
    float* image_v;                // float4[W*H] global__imageV, only with SUPER_SAMPLING
    const float* x2inv;            // 1001-entry table, only with SUPER_SAMPLING
    float* stage_flag;             // float[W*H] per launch: 1 = path traced, 0 = skipped by the stop criterion
    ptmi_sky sky;
    float cam_pos[4], cam_dir[4], cam_right[4], cam_up[4];
    uint32_t root_ref;
    uint32_t width, height;
    uint32_t max_depth;  // MAX_REFLECTION_NUMBER
    uint32_t n_lights;   // LIGHTS_SIZE
    uint32_t sampler;
    uint32_t super_sampling;  // -D SUPER_SAMPLING
    uint32_t tris_precomputed; // tris[] holds DTriPre records
    uint32_t plain_shading;    // every material a plain-colour MAT_STANDART and every light a LIGHT_POINT
    uint32_t nan_safe;         // the records can yield NaN distances (scene_needs_literal_kernel): the wavefront kernel's NANSAFE instantiation
    // the walk of a ray whose direction is NaN in every component (scene_layout.h: Relayout::nan_walk_*; 0xFFFFFFFF tests: walk it)
    uint32_t nan_walk_box_tests, nan_walk_tri_tests, nan_walk_last_tri;
    uint32_t n_records;        // nodes + leaf triangles in the one record array (nodes == tris)
    uint32_t wide_records;     // that array is 4 GB or more: byte offsets need 64 bits
    uint32_t boxes_ordered;    // every non-empty child box is finite with pMin <= pMax (see box_hit_ordered)
    uint32_t russian_roulette; // PTMI_FLAG_RUSSIAN_ROULETTE
    uint32_t source_seed;      // PTMI_FLAG_SOURCE_SEED
};

// The integrator's device code exists once per ARITHMETIC MODE (ptmi_device.hpp: strict / the reference's default OpenCL
// build, PTMI_FLAG_DEFAULT_ARITHMETIC): kernels.hip and kernel_wavefront.hip are compiled twice and their entry points
// carry the suffix `_da` in the default-arithmetic build.  A context picks one set (ptmi_api.cpp: kernels_of).
#define PTMI_DECLARE_INTEGRATOR_ENTRY_POINTS(SUFFIX)                                                                             \
    /* kernels.hip: one path per lane (kept for A/B and as a second implementation in the parity tests)                         \
       iteration ids of a launch: first_iteration + k * iteration_stride, k < n_iterations */                                   \
    int launch_render##SUFFIX(const DScene& sc, uint32_t first_iteration, uint32_t n_iterations, uint32_t iteration_stride,      \
                              void* stream, std::string* err);                                                                  \
    /* kernel_wavefront.hip: persistent wavefront state machine (default)                                                       \
       stack_levels = LDS traversal-stack entries per lane = depth of the uploaded tree (>= 1, <= 30) */                        \
    /* scene_in_device_memory = a device copy of `sc` (the kernel takes only the hot fields by value)                           \
       stage_stats: one word per staged path for the histograms (nullptr: the kernel issues the reference's atomics itself) */  \
    int launch_render_wavefront##SUFFIX(const DScene& sc, const DScene* scene_in_device_memory, uint32_t first_iteration,        \
                                        uint32_t n_iterations, uint32_t iteration_stride, uint32_t* job_counter,                \
                                        uint32_t stack_levels, bool scheduler_stats, float* stage,                              \
                                        uint32_t* stage_stats, void* stream, std::string* err);                                 \
    void last_wavefront_grid##SUFFIX(int device, uint32_t* lanes_per_workgroup, uint32_t* resident_workgroups);                  \
    /* ... and what must follow it, in launch order: staged radiances -> accumulators, staged statistics -> histograms */       \
    int launch_accumulate_staged##SUFFIX(const DScene& sc, uint32_t first_iteration, uint32_t n_iterations, const float* stage,  \
                                         const uint32_t* stage_stats, bool with_histograms, void* stream, std::string* err);    \
    int launch_histogram_staged##SUFFIX(const DScene& sc, uint32_t n_iterations, const uint32_t* stage_stats, void* stream,     \
                                        std::string* err);
PTMI_DECLARE_INTEGRATOR_ENTRY_POINTS()
PTMI_DECLARE_INTEGRATOR_ENTRY_POINTS(_da)
#undef PTMI_DECLARE_INTEGRATOR_ENTRY_POINTS
// kernel_wavefront.hip, default-arithmetic build only: the reciprocal determinant of every DTriPre record (u_den[3]) as the
// reference's default build computes it (FullKernel.cl:556 through v_rcp_f32: not reproducible on the host), written on
// the device after the upload.  `tri_ids[i]` == 0xFFFFFFFF marks a node record.
int launch_precompute_denominators_da(DTri* records, const uint32_t* tri_ids, uint32_t n_records, void* stream, std::string* err);

// display.hip: accumulators -> padded B,G,R scanlines (the reference's ConvertRGBAToBMPBuffer), on the device
int launch_display_bgr(const float* image_color, const float* image_ray_nb, uint8_t* out, uint32_t width, uint32_t height,
                       uint32_t row_stride, void* stream, std::string* err);

// display.hip: out[i] = ((parts[0][i] + parts[1][i]) + parts[2][i]) + ...  in this fixed order (the accumulators of the devices
// that shared a render, summed on the first one: ptmi_read_image / ptmi_read_snapshot); n_parts <= PTMI_MAX_DEVICES
int launch_sum_images(float* out, const float* const* parts, uint32_t n_parts, size_t n_floats, void* stream, std::string* err);
// dst[i] += src[i], i < n <= 64 (display.hip): a stage set's counter block into the context's
int launch_add_counters(unsigned long long* dst, const unsigned long long* src, uint32_t n, void* stream, std::string* err);

// iterations one wavefront launch may cover (bounds the staging array: 16 B x pixels x this)
#ifndef PTMI_MAX_ITERATIONS_PER_LAUNCH
// Measured on MI355X, 1M triangles 1080p (Msamples/s at 1 / 2 / 4 / 8 / 16 / 32 / 64 iterations per launch):
// 614 / 668 / 697 / 722 / 763 / 778 / 782 - a persistent launch ramps up and ends ragged; 32 keeps the staging arrays at
// 2 x 1.3 GB for a 1080p image (the cap falls with the image size, ptmi_setup_context)
#define PTMI_MAX_ITERATIONS_PER_LAUNCH 32
#endif
constexpr uint32_t kMaxIterationsPerLaunch = PTMI_MAX_ITERATIONS_PER_LAUNCH;

}  // namespace ptmi_internal
