/*
 * ptmi.h - C ABI of the MI355X path-tracing integrator (libptmi.so).
 *
 * Drop-in boundary: the three functions the reference's orchestration calls on
 * its device backend, Controleur/PathTracer_OpenCL.h:17-19
 *     OpenCL_SetupContext(GlobalVars&, Sampler)           (PathTracer_OpenCL.cpp:316-402)
 *     OpenCL_InitializeMemory(GlobalVars&)                (PathTracer_OpenCL.cpp:149-198)
 *     OpenCL_RunKernel(GlobalVars&, cb, nImages, t1,t2,t3)(PathTracer_OpenCL.cpp:56-140)
 * called once each, in that order, from PathTracer_Main (PathTracer.cpp:74,76,82),
 * plus the host BVH build that produces one of their inputs
 *     BVH_Create(GlobalVars&)                             (PathTracer_BVH.cpp:12-37).
 * Each entry point below names the reference code it replaces.  The C++ shim
 * with the reference's own signatures on top of this ABI is
 * opencl_pathtracer_amd/csrc/PathTracer_HIP.cpp.
 *
 * Conventions: plain pointers and sizes, caller-owned host memory,
 * context-owned device memory, no exceptions; every call returns PTMI_OK (0)
 * or a negative ptmi_status and leaves a message for ptmi_last_error().
 * There is NO CPU fallback: without a HIP device every compute entry point
 * fails with PTMI_ERR_NO_DEVICE.
 */
#ifndef PTMI_H
#define PTMI_H

#include <stddef.h>
#include <stdint.h>
#include "ptmi_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PTMI_ABI_VERSION 4 /* 3: PTMI_FLAG_DEFAULT_ARITHMETIC, ptmi_scheduler_stats::leaf_item_violations; 4: ::textured_hits */
#define PTMI_MAX_DEVICES 16       /* devices that can share one render */
#define PTMI_MAX_SNAPSHOT_SLOTS 65 /* ptmi_snapshot ring: slots 0..63 are the caller's, the last one the library's own */

typedef enum ptmi_status {
    PTMI_OK = 0,
    PTMI_ERR_INVALID_ARGUMENT = -1, /* null pointer, zero image, bad sampler ... */
    PTMI_ERR_NO_DEVICE = -2,        /* no HIP device / bad ordinal (reference: "Wrong context.", OpenCL.cpp:359-360) */
    PTMI_ERR_HIP = -3,              /* a HIP runtime call failed (reference: OpenCL_ErrorHandling, OpenCL.cpp:407-486) */
    PTMI_ERR_LIMIT = -4,            /* bvh depth >= 30 or lights >= 30 (reference guard, PathTracer.cpp:54-65) */
    PTMI_ERR_BAD_SCENE = -5,        /* scene arrays inconsistent (index out of range, cyclic bvh ...) */
    PTMI_ERR_STATE = -6,            /* call order violated (e.g. render before initialize_memory) */
    PTMI_ERR_UNSUPPORTED = -7       /* feature compiled out / not available in this build */
} ptmi_status;

typedef struct ptmi_ctx ptmi_ctx;

/* What OpenCL_BuildOptions bakes into the kernel with -D (OpenCL.cpp:292-314),
 * plus the device choice made in OpenCL_SetupContext (OpenCL.cpp:356-366). */
typedef struct ptmi_config {
    uint32_t struct_size;    /* = sizeof(ptmi_config), ABI guard */
    int32_t device;          /* HIP device ordinal (reference: devices[0]) */
    uint32_t image_width;    /* -D IMAGE_WIDTH  (globalVars.imageWidth)  */
    uint32_t image_height;   /* -D IMAGE_HEIGHT (globalVars.imageHeight) */
    uint32_t ray_max_depth;  /* -D MAX_REFLECTION_NUMBER (globalVars.rayMaxDepth) */
    uint32_t lights_size;    /* -D LIGHTS_SIZE (globalVars.lightsSize) */
    uint32_t sampler;        /* PTMI_SAMPLER_* : -D SAMPLE_JITTERED / _RANDOM / _UNIFORM */
    uint32_t super_sampling; /* -D SUPER_SAMPLING (globalVars.superSampling): adaptive sampling, FullKernel.cl:1152-1172,1219-1222.
                                With the RANDOM sampler the reference races on the count / variance of the pixel a sample lands on;
                                here those updates are atomic (results statistically equal, not bit-equal, to a serial evaluation) */
    uint32_t flags;          /* PTMI_FLAG_* */
    /* Multi-GPU render (the reference drives devices[0] only, OpenCL.cpp:363-366): n_devices > 1 replicates the scene on
     * devices[0..n_devices) and spreads the iteration ids of every ptmi_render call over them (device k takes the ids
     * congruent to k modulo n_devices: the same id set as a single-device render, so the same samples); devices[0] is
     * the one the partial accumulators are summed on, by peer copies over xGMI + one add kernel IN DEVICE ORDER (the float
     * sums of an image are a function of the device list alone); the per-image loop (ptmi_read_snapshot), where one device's
     * share changes per image, re-sends only that share.  Environment PTMI_REDUCE=rccl: the sum is an ncclReduce instead
     * (librccl, loaded at run time, NCCL 2 API checked by ncclGetVersion) - opt-in: its order of additions for more than two
     * devices is the collective algorithm's, so the last bits of an image depend on it, and any failure (library absent,
     * device list refused, a run-time error) falls back to the peer path for the rest of the context's life.
     * n_devices 0 or 1 = single device `device`.  A device may be listed more than once (used by the tests on a one-GPU box;
     * RCCL refuses that, the peer path runs). */
    uint32_t n_devices;
    int32_t devices[PTMI_MAX_DEVICES];
} ptmi_config;

#define PTMI_FLAG_NO_HISTOGRAMS 1u /* skip the three per-path histogram atomics (FullKernel.cl:1319-1331); totals are still kept */
#define PTMI_FLAG_SCHEDULER_STATS 4u /* collect ptmi_scheduler_stats (a few scalar ops per loop trip; off by default) */
#define PTMI_FLAG_MEGAKERNEL 2u    /* one path per lane (kernels.hip) instead of the persistent wavefront kernel; same results */
#define PTMI_FLAG_RUSSIAN_ROULETTE 8u /* NON-PARITY mode: the termination block the reference ships commented out (FullKernel.cl:1306-1314,
                                         RUSSIAN_ROULETTE false in header.cl:12), as it is written there: from the 7th bounce on a path
                                         whose largest transfer component / (bounce - 5) is below 1 draws a random number, ends unless it
                                         exceeds that coefficient, and has its transfer divided by it.  Images differ from the reference's. */

#define PTMI_FLAG_SOURCE_SEED 32u /* NON-PARITY mode: InitializeRandomSeed as its SOURCE reads under wrapping arithmetic (header.cl:255-264:
                                    `seed *= 2011; seed *= seed; if(seed == 0) seed = 1;`).  The compiled reference tests the un-squared index
                                    instead (the square overflows a signed int: undefined, and LLVM folds the test), so a path whose index is a
                                    non-zero multiple of 2^16 keeps seed 0 and draws 0 for every random number - one path in 65536, the same
                                    pixels every 2^16 / gcd(2^16, W*H) iterations.  Parity modes reproduce that; this flag gives those paths
                                    seed 1, as the source intends.  Images differ from the reference's at those pixels only. */
#define PTMI_FLAG_DEFAULT_ARITHMETIC 16u /* The arithmetic of the build the reference's own build line produces (OpenCL_BuildOptions,
                                         OpenCL.cpp:292-314, passes no floating-point option): a*b+c written in one expression is one
                                         fused multiply-add, a/b goes through v_rcp_f32 of the divisor's mantissa (2.5 ulp), sqrt is
                                         v_sqrt_f32 - bit for bit what that build computes on this GPU.  Without the flag: the STRICT
                                         arithmetic (every operation of the source correctly rounded, the build the same source gives
                                         with -ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt), also bit for bit.  Same cost. */

/* What OpenCL_InitializeMemory copies with CL_MEM_COPY_HOST_PTR and passes as
 * kernel arguments 1..17 (OpenCL.cpp:165-197).  Arrays are raw dumps of the
 * ptmi_scene.h structs; a zero-length array may be NULL. */
typedef struct ptmi_scene {
    uint32_t struct_size; /* = sizeof(ptmi_scene) */
    const ptmi_node* bvh;                 uint32_t bvh_size;
    const ptmi_triangle* triangulation;   uint32_t triangulation_size;
    const ptmi_light* lights;             uint32_t lights_size;
    const ptmi_material* materiaux;       uint32_t materiaux_size;
    const ptmi_texture* textures;         uint32_t textures_size;
    const ptmi_uchar4* textures_data;     uint32_t textures_data_size;
    const ptmi_sky* sky;
    ptmi_float4 camera_position;  /* kernel arg 1 */
    ptmi_float4 camera_direction; /* kernel arg 2 */
    ptmi_float4 camera_right;     /* kernel arg 3 */
    ptmi_float4 camera_up;        /* kernel arg 4 */
} ptmi_scene;

/* Totals the roofline model is priced from: exact sums of the
 * reference's per-path counters over everything rendered since the last
 * ptmi_initialize_memory / ptmi_clear. */
typedef struct ptmi_counters {
    uint64_t paths;          /* Kernel_Main invocations that reached the epilogue */
    uint64_t segments;       /* BVH_IntersectRay calls ("samples": paths x bounces) */
    uint64_t surface_hits;   /* sum of r.reflectionId = segments that hit geometry */
    uint64_t shadow_rays;    /* BVH_IntersectShadowRay calls */
    uint64_t box_tests;      /* sum of numIntersectedBBx incl. shadow rays */
    uint64_t triangle_tests; /* sum of numIntersectedTri incl. shadow rays */
} ptmi_counters;

/* Wave-scheduler statistics of the persistent wavefront kernel since the last clear: how many loop trips a
 * wave spent on each step kind and how many of its 64 lanes were active in them (lanes / (64 * trips) =
 * SIMD utilisation of that kind).  Collected only with PTMI_FLAG_SCHEDULER_STATS; all zero otherwise and for the
 * one-path-per-lane kernel. */
typedef struct ptmi_scheduler_stats {
    uint64_t trips_node, lanes_node;         /* node trips: lanes that took an inner-node step */
    uint64_t trips_triangle, lanes_triangle; /* leaf passes: triangles tested, one per lane */
    uint64_t trips_path, lanes_path;         /* path logic (shade / shadow set-up / scatter / regenerate) */
    uint64_t cycles_path, cycles_loop;       /* shader-clock cycles, summed over waves: inside path-logic passes / in the main loop */
    uint64_t leaf_item_violations;           /* leaf passes: work items whose owner lane or triangle record index was out of range when a
                                                lane read them (an item read before its writer: must be 0; checked only while collecting) */
    uint64_t paths_retraced;                 /* paths a launch gave up because one of their rays was not a number, traced again by the
                                                reference's literal loops (ptmi_literal_kernel_reason, below); counted ALWAYS, flag or not */
    uint64_t textured_hits;                  /* surface hits whose material has a file texture (one texel fetch each at least: the
                                                4 * N_texel term of the algorithmic-bytes model); counted only while collecting */
    uint64_t workgroup_lanes, resident_workgroups; /* the grid of the most recent wavefront-kernel launch on devices[0] (this context's
                                                or another's): lanes per workgroup, and how many workgroups the device holds at once
                                                - the persistent grid; 0 before the first launch; filled ALWAYS */
} ptmi_scheduler_stats;

/* The reference's device-side consistency checks: with -D LOG_INFO (OpenCL.cpp:310, globalVars.printLogInfos) its kernel
 * prints a line whenever one of the ASSERT / WARNING conditions of PathTracer_FullKernel_header.cl:21-48 fails.  Here the
 * failures of the checks on the live path are COUNTED, in the same debug instantiation of the kernel that collects the
 * scheduler statistics (PTMI_FLAG_SCHEDULER_STATS); all zero otherwise.  A clean render has zeros everywhere except
 * statistics_out_of_range, which counts what the reference's histograms silently drop. */
typedef struct ptmi_invariant_checks {
    uint64_t sample_out_of_range;      /* FullKernel.cl:1217 "SAMPLER - invalid pixel" */
    uint64_t normal_not_facing_ray;    /* :1275 "Kernel_Main incorrect normals": dot(dir, Ns) < 0 && dot(dir, Ng) < 0 */
    uint64_t negative_direct_radiance; /* :951 "Scene_ComputeDirectIllumination incorrect radiance L" */
    uint64_t scattered_below_surface;  /* header.cl:243 "Vector_PutInSameHemisphereAs": dot(out, N) > 0 */
    uint64_t statistics_out_of_range;  /* :1325,1330 "global__rayIntersectionBBx / Tri to large": >= 5000 tests on one path */
    /* NOT a check of the reference: bounces whose result the reference's SOURCE leaves undefined.  Its water material refracts
     * when random() >= the Fresnel fraction (FullKernel.cl:836-843); on total internal reflection the fraction is 1 and
     * Material_FresnelWaterReflectionFraction has returned (:237) BEFORE writing refractionDirection and refractionMultCoeff
     * (:249-251) - and random() returns exactly 1.0 for the 64 seeds nearest 2^31 (header.cl:246-253), so about one interior
     * water hit in 10^8 refracts along an uninitialised direction with an uninitialised factor (the compiled kernel reads
     * whatever its registers hold: stale values of other variables).  The integrator takes a zero direction and the factor
     * n2^2/n1^2 there.  Such a path is the one thing that may differ from the reference kernel's image; counted so that a
     * comparison can tell (tools/north_star_full_size.py). */
    uint64_t refraction_undefined_in_reference;
} ptmi_invariant_checks;

/* ---- lifecycle ---------------------------------------------------------- */

/* Replaces OpenCL_SetupContext (OpenCL.cpp:316-402): picks the device, creates
 * the stream and selects the kernel specialisation.  *ctx is NULL on failure. */
int ptmi_setup_context(ptmi_ctx** ctx, const ptmi_config* config);

/* Replaces OpenCL_InitializeMemory (OpenCL.cpp:149-198): validates the scene,
 * re-lays it out for the device, uploads it and allocates + ZEROES the
 * accumulators (the reference leaves them uninitialised, OpenCL.cpp:159-164).
 * May be called again on the same context with a new scene. */
int ptmi_initialize_memory(ptmi_ctx* ctx, const ptmi_scene* scene);

/* Replaces the clSetKernelArg(0, imageId) + clEnqueueNDRangeKernel(W x H) pair
 * of OpenCL_RunKernel's loop (OpenCL.cpp:85-89), generalised to a range:
 * renders iterations [first_iteration, first_iteration + n_iterations) for
 * every pixel and adds them into the accumulators in iteration order (the
 * float sums equal a launch-per-iteration loop's bit for bit).  Internally
 * at most 32 iterations per kernel launch (1 with super_sampling, whose stop
 * criterion reads the accumulators of the previous iteration).  Asynchronous
 * on the context's stream.
 * A caller that renders one short call (fewer than 4 iterations) after the other, in order, and waits for each - the
 * reference's loop - is RENDERED AHEAD OF: once the pattern has been seen the library keeps launches for the next calls in
 * flight (two per device, each for up to four calls), and a call that finds its iterations rendered adopts them.  Nothing a caller can
 * read differs from rendering on demand (image, sample counts, histograms, counters after every call); a call that leaves
 * the pattern drops what ran ahead.  Costs per device: up to 8 iterations of device time nobody asked for when such a caller
 * stops, and staging memory for 16 iterations (20 bytes per pixel each).  Environment: PTMI_RENDER_AHEAD=0 switches it off,
 * PTMI_RENDER_AHEAD_CALLS=1 keeps it to one call per launch. */
int ptmi_render(ptmi_ctx* ctx, uint32_t first_iteration, uint32_t n_iterations);

/* clFinish (OpenCL.cpp:89). */
int ptmi_synchronize(ptmi_ctx* ctx);

/* Replaces the two blocking clEnqueueReadBuffer of every image (OpenCL.cpp:97-98):
 * image_color = float[4*W*H] sum of radiance, image_ray_nb = float[W*H] sample
 * count.  Either pointer may be NULL.  Waits for everything queued so far.  The bytes cross the bus into pinned
 * memory: straight into the destination if the caller has page-locked it (ptmi_pin_host_buffer), through a
 * pinned staging buffer of the context and a host copy otherwise. */
int ptmi_read_image(ptmi_ctx* ctx, float* image_color, float* image_ray_nb);

/* Page-lock a host buffer the caller will hand to ptmi_read_image / ptmi_read_snapshot again and again (the viewer's
 * imageColor / imageRayNb, which the reference reads into after every image, OpenCL.cpp:97-98): readbacks then DMA
 * into it without an intermediate copy.  The buffer must stay allocated until ptmi_unpin_host_buffer or ptmi_release. */
int ptmi_pin_host_buffer(ptmi_ctx* ctx, void* buffer, size_t bytes);
int ptmi_unpin_host_buffer(ptmi_ctx* ctx, void* buffer);

/* The same readback split in two so that the launches of the NEXT images run while image k crosses the bus
 * (the reference blocks on every image: launch, clFinish, read, callback - OpenCL.cpp:85-103):
 *   ptmi_snapshot(slot)       queued behind the launches issued so far: a device-side copy of the accumulators
 *                             (every device of a multi-GPU render) into ring slot `slot` < PTMI_MAX_SNAPSHOT_SLOTS;
 *   ptmi_read_snapshot(slot)  waits for that copy only (not for launches queued after it), sums the devices'
 *                             partial images on devices[0], copies the result to the host buffers; with both
 *                             pointers NULL it only waits until the snapshot has been taken (clFinish of that image).
 * Loop of a viewer: render(k+1); read_snapshot(k); show; snapshot(k+1) ...  (csrc/PathTracer_HIP.cpp).
 *   ptmi_render_snapshots(first, n, first_slot)   ptmi_render(first, n) that ALSO leaves a snapshot after every one of
 *                             its n iterations - iteration first + k in slot (first_slot + k) % (PTMI_MAX_SNAPSHOT_SLOTS - 1)
 *                             - although the iterations share kernel launches: a viewer still gets every image of the
 *                             reference's launch-per-image loop, at the throughput of n iterations per launch.
 *                             n < PTMI_MAX_SNAPSHOT_SLOTS; JITTERED / UNIFORM sampler, wavefront kernel, no super_sampling. */
int ptmi_snapshot(ptmi_ctx* ctx, uint32_t slot);
int ptmi_render_snapshots(ptmi_ctx* ctx, uint32_t first_iteration, uint32_t n_iterations, uint32_t first_slot);
int ptmi_read_snapshot(ptmi_ctx* ctx, uint32_t slot, float* image_color, float* image_ray_nb);

/* The inverse of ptmi_read_image: load the accumulators (e.g. to resume a render saved earlier, or to accumulate on top of
 * another device's partial result).  Either pointer may be NULL (left as is).  Synchronises first. */
int ptmi_write_image(ptmi_ctx* ctx, const float* image_color, const float* image_ray_nb);

/* What the reference's viewer does with those two buffers after every image - ConvertRGBAToBMPBuffer,
 * Alone/PathTracer_bitmap.cpp:237-286, called from Alone/PathTracer_Dialog.cpp:161-185 - done on the device:
 * 24-bit B,G,R scanlines, image row 0 first, each `row_stride` bytes long ((3*W + 3) & ~3, the BMP padding, zero
 * filled), pixel = (int) min(sum * 255.f / n, 255.f); 3 bytes per pixel cross the bus instead of 20.
 * `bgr` holds H * row_stride bytes.  Synchronises first. */
int ptmi_read_display(ptmi_ctx* ctx, uint8_t* bgr, uint32_t row_stride);

/* Replaces the three statistic reads after the loop (OpenCL.cpp:110-112):
 * ray_depths[ray_max_depth+1], ray_intersected_bbx[5000], ray_intersected_tri[5000].
 * Any pointer may be NULL. */
int ptmi_read_statistics(ptmi_ctx* ctx, uint32_t* ray_depths, uint32_t* ray_intersected_bbx,
                         uint32_t* ray_intersected_tri);

/* Zero accumulators, histograms and counters (new render of the same scene). */
int ptmi_clear(ptmi_ctx* ctx);

/* Replaces the release block at the end of OpenCL_RunKernel (OpenCL.cpp:120-139). */
void ptmi_release(ptmi_ctx* ctx);

/* ---- measurement / plumbing -------------------------------------------- */

int ptmi_get_counters(ptmi_ctx* ctx, ptmi_counters* out);
int ptmi_get_scheduler_stats(ptmi_ctx* ctx, ptmi_scheduler_stats* out);
int ptmi_get_invariant_checks(ptmi_ctx* ctx, ptmi_invariant_checks* out);
/* NaN distances.  The reference's triangle test (FullKernel.cl:519-589) rejects with comparisons only, so a triangle on which
 * it computes NaNs is ACCEPTED, with a NaN distance, and from then on the LAST triangle that passes wins, not the nearest.
 * The integrator reproduces that bit for bit.  Where the RAY is not a number (a refraction at |cos| = 1 + 1 ulp, cl:235) the
 * wavefront kernel gives the path up and re-traces it with the reference's literal loops (every sampler; RANDOM: a give-up list).
 * Where the scene's RECORDS are the source - zero-area triangles, whose normal the importer computes as 0/0; non-finite or
 * astronomically large coordinates - this call returns why (else NULL), and the scene is rendered by an instantiation of the
 * wavefront kernel that looks at every accepted triangle of a closest-hit query: only the paths that REACH such a record are
 * given up and traced again by the literal loops (ptmi_scheduler_stats.paths_retraced), every other path runs as in a clean
 * scene.  With the RANDOM sampler (nothing is staged there, so nothing can be traced again) such a scene is rendered by the
 * one-path-per-lane kernel as a whole (as with PTMI_FLAG_MEGAKERNEL: same results, slower; PTMI_ERR_UNSUPPORTED with
 * super_sampling).  The string lives until the next ptmi_initialize_memory / ptmi_release. */
const char* ptmi_literal_kernel_reason(const ptmi_ctx* ctx);

/* How a multi-device context sums its devices' partial images (for records and scaling logs): *rccl_state = 1 when the last sum
 * went through ncclReduce (PTMI_REDUCE=rccl), 0 when the collective has not been tried, -1 when it is unavailable or was refused
 * (peer copies + the add kernel in device order: the default); *n_communicators = the communicators ncclCommInitAll returned for
 * this context (= the device count when the collective is in use, else 0); *nccl_version = ncclGetVersion's code (0: library not
 * loaded).  Any pointer may be NULL. */
int ptmi_reduce_path(const ptmi_ctx* ctx, int* rccl_state, int* n_communicators, int* nccl_version);

/* Device time of the integrator kernel launches issued by ptmi_render since the
 * last call, measured with HIP events on the context's stream.  Synchronises. */
int ptmi_kernel_time(ptmi_ctx* ctx, double* total_ms, uint32_t* n_launches);

/* Run on a caller-owned hipStream_t (passed as void*; NULL = context's own).  Single-device contexts only, like the
 * three accumulator entry points below (PTMI_ERR_UNSUPPORTED with n_devices > 1: each device has partial sums). */
int ptmi_set_stream(ptmi_ctx* ctx, void* hip_stream);

/* Device pointers of the accumulators (float[4*W*H], float[W*H]) so a caller
 * that owns a collective library can reduce them in place (multi-GPU spp shards). */
int ptmi_device_accumulators(ptmi_ctx* ctx, void** d_image_color, void** d_image_ray_nb);

/* SUPER_SAMPLING only: the per-pixel variance accumulator global__imageV (float4[W*H]; the sum of squared
 * deviations the stop criterion reads, FullKernel.cl:1152-1172,1346-1349).  On a multi-device context every device
 * samples adaptively on its own accumulators (its stop decisions are its own) and ptmi_read_variance returns the moments
 * of all devices merged pairwise (Chan's update; the same arithmetic as opencl_pathtracer_amd/distributed.py:
 * merge_moments, which does it across processes).  PTMI_ERR_STATE without super_sampling. */
int ptmi_read_variance(ptmi_ctx* ctx, float* image_v);
int ptmi_write_variance(ptmi_ctx* ctx, const float* image_v); /* with ptmi_write_image: resume a SUPER_SAMPLING render */
int ptmi_device_variance(ptmi_ctx* ctx, void** d_image_v);

/* Adopt caller-allocated device buffers as accumulators (e.g. torch tensors);
 * they are NOT zeroed and NOT freed by the context.  Pass NULLs to go back. */
int ptmi_bind_accumulators(ptmi_ctx* ctx, void* d_image_color, void* d_image_ray_nb);

/* Message of the last failure on this context (ctx may be NULL for failures of
 * ptmi_setup_context / ptmi_bvh_create).  Never NULL. */
const char* ptmi_last_error(const ptmi_ctx* ctx);

int ptmi_abi_version(void);
int ptmi_device_count(void);

/* Which iteration ids of [first_iteration, first_iteration + n_iterations) device `k` of `n_devices` renders: the ids
 * congruent to k modulo n_devices = *first_k, *first_k + n_devices, ... (*n_k of them).  Pure arithmetic, no device. */
void ptmi_device_share(uint32_t first_iteration, uint32_t n_iterations, uint32_t k, uint32_t n_devices, uint32_t* first_k,
                       uint32_t* n_k);

/* ---- host-side producer of the bvh input -------------------------------- */

/* Replaces BVH_Create (PathTracer_BVH.cpp:12-37 -> BVH_BuildStructure :109-356):
 * binned-SAH build, bit-compatible with the reference (same node order, same
 * in-place reordering of `triangulation`).  `bvh` must hold 2*n-1 nodes.
 * Host only; needs no device. */
/* Host-only (no device, no context): would ptmi_initialize_memory accept `scene` on a context set up with `config`?  The same
 * checks - every index the kernel will follow, texture extents, texture coordinates, the tree's structure and depth - and the
 * same error codes, with the message in ptmi_last_error(NULL).  For importers and asset pipelines on machines without a GPU. */
int ptmi_validate_scene(const ptmi_config* config, const ptmi_scene* scene);

int ptmi_bvh_create(ptmi_triangle* triangulation, uint32_t triangulation_size, ptmi_node* bvh,
                    uint32_t* bvh_size, uint32_t* bvh_max_depth);

#ifdef __cplusplus
}
#endif
#endif /* PTMI_H */
