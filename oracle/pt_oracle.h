/*
 * pt_oracle.h - CPU oracle for the path-tracing hot path.  TEST INFRASTRUCTURE.
 *
 * A scalar, plain-C restatement of the reference integrator
 * (Kernel/PathTracer_FullKernel.cl + PathTracer_FullKernel_header.cl), one
 * function per reference function, each citing the file:line it follows.  It
 * is the checker the HIP path is compared against and the timed "cpu_baseline"
 * of bench.py.  Nothing in the product (libptmi.so, the package, the C++ shim)
 * may include, link or call anything in this directory; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Pinning (see DESIGN.md "Oracle"): the reference ships no tests or golden
 * vectors.  The oracle is pinned against outputs of the reference ITSELF:
 *   - Kernel_Main compiled UNMODIFIED for gfx950 with the image's own
 *     clang -x cl + ROCm OpenCL device libraries (oracle/Makefile target
 *     `ref-kernels` -> oracle/_ref/ *.hsaco), run on the MI355X by
 *     oracle/ref_gpu_runner.cpp; its images/histograms are committed under
 *     tests/golden/ with the script that made them.
 *   - BVH_Create compiled UNMODIFIED for x86-64 (target `ref-bvh` ->
 *     oracle/_ref/libref_bvh.so) pins the product's BVH builder.
 * Arithmetic that OpenCL leaves implementation-defined follows the platform
 * the reference runs on here (ROCm OpenCL device library on gfx950): dot()
 * and cross() as its fma chains, normalize() = v * rsqrt(dot) with its range
 * scaling and the hardware's v_rsq_f32 (deviation table measured on the
 * MI355X, tests/golden/rsq_gfx950.npz, see pto_set_rsq_table), sin/cos as its
 * argument reduction + polynomials (include/ptmi_detmath.h); every other
 * operation is one correctly rounded IEEE operation in the written order.
 * That is the reference's STRICT build (-ffp-contract=off
 * -cl-fp32-correctly-rounded-divide-sqrt), which the oracle equals bit for
 * bit (tests/test_oracle_golden.py); its DEFAULT build is other legal
 * arithmetic and is compared statistically.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>
#include "ptmi_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pto_scene {
    const ptmi_node* bvh;
    const ptmi_triangle* triangulation;
    const ptmi_light* lights;
    const ptmi_material* materiaux;
    const ptmi_texture* textures;
    const ptmi_uchar4* textures_data;
    const ptmi_sky* sky;
    ptmi_float4 camera_position, camera_direction, camera_right, camera_up;
    /* the -D specialisation values (PathTracer_OpenCL.cpp:292-314) */
    uint32_t image_width, image_height;
    uint32_t ray_max_depth; /* MAX_REFLECTION_NUMBER */
    uint32_t lights_size;   /* LIGHTS_SIZE */
    uint32_t sampler;       /* PTMI_SAMPLER_* */
    uint32_t super_sampling;
    const float* x2inv;     /* 1001-entry table (Kernel/X2inv.cl), only if super_sampling */
    uint32_t russian_roulette; /* the block the reference keeps commented out (cl:1306-1314, RUSSIAN_ROULETTE false): non-parity mode */
    uint32_t source_seed;      /* non-parity mode: InitializeRandomSeed's zero test on the SQUARE, as the source reads (h:255-264) */
    uint32_t first_sample_guard; /* SUPER_SAMPLING, the product's one guard: a pixel's FIRST sample has no deviation whatever the
                                  * iteration id (the reference divides 0 by 0 when a render does not start at iteration 0, cl:1349,
                                  * and keeps a NaN variance for good); with it the oracle models a render that starts anywhere */
} pto_scene;

/* Output buffers, caller-allocated; accumulated into (not zeroed here). */
typedef struct pto_buffers {
    float* image_color;            /* [4*W*H] global__imageColor */
    float* image_ray_nb;           /* [W*H]   global__imageRayNb */
    float* image_v;                /* [4*W*H] global__imageV     */
    uint32_t* ray_depths;          /* [ray_max_depth+1] */
    uint32_t* ray_intersected_bbx; /* [5000] */
    uint32_t* ray_intersected_tri; /* [5000] */
} pto_buffers;

typedef struct pto_totals {
    uint64_t paths, segments, surface_hits, shadow_rays, box_tests, triangle_tests;
} pto_totals;

/* Runs Kernel_Main for every pixel and every iteration in
 * [first_iteration, first_iteration+n_iterations), iteration-major like the
 * reference's launch loop (PathTracer_OpenCL.cpp:76-107).  n_threads > 1 splits
 * rows over threads (pixel-major inside a thread; identical results for the
 * JITTERED and UNIFORM samplers where a work-item owns its pixel; the RANDOM
 * sampler always runs on one thread).  totals may be NULL. */
void pto_render(const pto_scene* scene, uint32_t first_iteration, uint32_t n_iterations,
                const pto_buffers* out, int n_threads, pto_totals* totals);

/* Same for image rows [row0, row1) only (bounded CPU-baseline samples of a large image). */
void pto_render_rows(const pto_scene* scene, uint32_t first_iteration, uint32_t n_iterations, uint32_t row0,
                     uint32_t row1, const pto_buffers* out, int n_threads, pto_totals* totals);

/* One work-item: returns 0 if the super-sampling criterion skipped it. */
int pto_kernel_main(const pto_scene* scene, uint32_t gid_x, uint32_t gid_y, uint32_t iteration,
                    const pto_buffers* out, pto_totals* totals);

/* Per-bounce trace of one path (debug aid for localising divergences). */
typedef struct pto_bounce {
    uint32_t triangle_id, material_id;
    float s, t;
    float point[4], ns[4], out_dir[4], transfer[4], radiance[4];
    int32_t seed_after;
    uint32_t n_bbx, n_tri;
} pto_bounce;
int pto_trace_path(const pto_scene* scene, uint32_t gid_x, uint32_t gid_y, uint32_t iteration,
                   pto_bounce* bounces, int max_bounces, float radiance_out[4]);

/* Unit-level entry points (known-answer tests). */
int32_t pto_initialize_random_seed(uint32_t gid_x, uint32_t gid_y, uint32_t w, uint32_t h, uint32_t iteration);
float pto_random(int32_t* seed);
void pto_sampler(uint32_t kind, uint32_t gid_x, uint32_t gid_y, uint32_t w, uint32_t h, uint32_t iteration,
                 int32_t* seed, float sample[2]);
int pto_bounding_box_intersects(const ptmi_bounding_box* bb, const float origin[4], const float direction[4],
                                float squared_distance);
int pto_triangle_intersects(const ptmi_triangle* tri, const float origin[4], const float direction[4],
                            float* squared_distance, float* s, float* t, float point[4]);
void pto_concentric_sample_disk(int32_t* seed, float* dx, float* dy);
void pto_cosine_sample_hemisphere(int32_t* seed, const float n[4], float out[4]);
float pto_fresnel_glass(const float incident[4], const float n[4]);
float pto_fresnel_varnish(const float incident[4], const float n[4]);
void pto_sky_color(const ptmi_sky* sky, const ptmi_uchar4* textures_data, const float direction[4], float rgba[4]);
void pto_sincos(float x, float* s, float* c);
/* normalize()'s reciprocal square root is the hardware instruction of the platform the reference runs on: its deviations
 * from the correctly rounded value come from a table measured on that hardware (2^24 entries x 2 bits, see pt_oracle.c).
 * Must be set before anything that normalises a vector is called. */
void pto_set_rsq_table(const uint8_t* packed);
float pto_hardware_rsq(float x);
/* the default-arithmetic build's division and square root go through v_rcp_f32 / v_sqrt_f32 (pt_oracle.c header):
 * tests/golden/rcp_gfx950.npz (2^23 entries), sqrt_gfx950.npz (2^24) */
void pto_set_rcp_table(const uint8_t* packed);
void pto_set_sqrt_table(const uint8_t* packed);
float pto_hardware_rcp(float mantissa);
float pto_hardware_sqrt(float x);
int pto_default_arithmetic(void); /* which of the two builds this library is */
void pto_arith_probe(const float* a, const float* b, uint32_t n, float* out); /* the ten operators of arith_probe.cl */

#ifdef __cplusplus
}
#endif
#endif
