// PathTracer_HIP.cpp - drop-in replacement for the reference's Controleur/PathTracer_OpenCL.cpp.
//
// Same three entry points, same order of use, same ownership and error convention as the reference backend
// (PathTracer_OpenCL.h:17-19; called from PathTracer_Main, PathTracer.cpp:74,76,82), implemented on the C ABI
// of libptmi.so:
//
//   OpenCL_SetupContext(globalVars, sampler)   -> ptmi_setup_context     (device, stream, specialisation values
//                                                  the reference bakes in with -D, OpenCL.cpp:292-314)
//   OpenCL_InitializeMemory(globalVars)        -> ptmi_initialize_memory (upload of the raw struct arrays,
//                                                  OpenCL.cpp:165-171, camera args :180-183)
//   OpenCL_RunKernel(globalVars, cb, n, t1..3) -> per image: ptmi_render + ptmi_synchronize, ptmi_read_image into
//                                                  globalVars.imageColor / imageRayNb, callback (OpenCL.cpp:76-107);
//                                                  after the loop ptmi_read_statistics (:110-112) and ptmi_release
//                                                  (:120-139).  The three timers accumulate clock() ticks like
//                                                  the reference (:66-104).
//   BVH_Create(globalVars)                     -> ptmi_bvh_create (PathTracer_BVH.cpp:12-37): `new Node[2n-1]`,
//                                                  triangulation reordered in place, bvhSize / bvhMaxDepth set.
//   BVH_GetCharacteristics(...)                -> the tree statistics walk of PathTracer_BVH.cpp:362-401, which the
//                                                  orchestration's printer links against (PathTracer.cpp).
// With these two the file also stands in for Controleur/PathTracer_BVH.cpp; a host that keeps the reference's own
// builder defines PTMI_SHIM_WITHOUT_BVH.
//
// Errors: any failure throws std::runtime_error with the library's message, as every cl error does in the
// reference (OpenCL_ErrorHandling, OpenCL.cpp:407-486); PathTracer_Main catches std::exception (PathTracer.cpp:99).
// State: one context per process in a file-scope variable, like the reference's file-scope cl objects
// (OpenCL.cpp:19-47): one render at a time, released at the end of OpenCL_RunKernel.
//
// Where the reference drives devices[0] of its platform (OpenCL.cpp:363-366), this backend spreads the images of a
// render over EVERY HIP device of the node (iteration ids modulo the device count, inside libptmi) and sums the
// partial images on the first one before each readback: the caller changes nothing.
//
// The render loop keeps the reference's contract - after image k the host buffers hold the sum of images 0..k and the
// callback runs - but does not idle the GPU while image k crosses the bus and the viewer paints it: the launches of the
// next PTMI_LOOKAHEAD steps are already queued, and what is read back is a device-side snapshot taken right behind
// image k's launch (ptmi_snapshot / ptmi_read_snapshot), DMA'd straight into the caller's buffers (page-locked for the
// duration of OpenCL_RunKernel).
//
// Environment knobs (all optional):
//   PTMI_DEVICES = "all" (default) or a comma-separated list of HIP ordinals;  PTMI_DEVICE = one ordinal (wins)
//   PTMI_BURST = images that share one kernel launch while EACH still gets its readback and callback, in order (default 16:
//                the viewer sees every image of the reference's loop, in bursts, at the throughput of 16 images per launch;
//                1 = one launch per image)
//   PTMI_IMAGES_PER_LAUNCH = images rendered per step AND per callback (default 1 = the reference's count of callbacks)
//   PTMI_LOOKAHEAD = steps queued ahead of the one being read back when PTMI_BURST is 1 (default max(2, devices); 0 = the
//                    reference's launch / wait / read / callback sequence)
//   PTMI_STRICT_ARITHMETIC = 1: the strict arithmetic instead of the reference's default-build arithmetic (ptmi.h, PTMI_FLAG_DEFAULT_ARITHMETIC)
//   PTMI_LOG = 1 (or globalVars.printLogInfos, the reference's -D LOG_INFO switch, OpenCL.cpp:310): one line per step
//              on stderr with the iteration range and the three timers
#ifdef PTMI_USE_REFERENCE_HEADERS
#include "PathTracer_OpenCL.h"  // the reference's own header (needs CL/cl.h and -fms-extensions for ALIGN)
#include "PathTracer_BVH.h"
#else
#include "pathtracer_backend.hpp"
#endif

#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <stdexcept>
#include <string>
#include <vector>

#include "ptmi.h"

namespace PathTracerNS {

namespace {

ptmi_ctx* g_ctx = nullptr;
unsigned g_devices = 1;
bool g_log = false;

[[noreturn]] void fail(const char* where, int code)
{
    std::string msg = std::string(where) + " failed (" + std::to_string(code) + "): " + ptmi_last_error(g_ctx);
    if (g_ctx) {
        ptmi_release(g_ctx);
        g_ctx = nullptr;
    }
    throw std::runtime_error(msg);
}

// the samplers whose samples land on the work-item's own pixel (FullKernel.cl:1119-1150): launches of several iterations are
// staged per iteration and can be snapshotted one by one; also no adaptive sampling (its launches are one iteration anyway)
bool g_staged = true;
bool sampler_owns_pixels(const GlobalVars&) { return g_staged; }

unsigned env_uint(const char* name, unsigned fallback)
{
    const char* v = std::getenv(name);
    return v && *v ? (unsigned)std::strtoul(v, nullptr, 10) : fallback;
}

}  // namespace

void OpenCL_SetupContext(GlobalVars& globalVars, Sampler sampler)
{
    if (g_ctx) {  // a previous render that never reached OpenCL_RunKernel
        ptmi_release(g_ctx);
        g_ctx = nullptr;
    }
    ptmi_config cfg{};
    cfg.struct_size = sizeof cfg;
    // devices: PTMI_DEVICE = one ordinal; else PTMI_DEVICES = list or "all" (default: every device of the node)
    std::vector<int> devices;
    const char* one = std::getenv("PTMI_DEVICE");
    const char* list = std::getenv("PTMI_DEVICES");
    if (one && *one) {
        devices.push_back((int)std::strtol(one, nullptr, 10));
    } else if (list && *list && std::string(list) != "all") {
        for (const char* p = list; *p;) {
            char* end = nullptr;
            const long v = std::strtol(p, &end, 10);
            if (end == p) break;
            devices.push_back((int)v);
            p = (*end == ',') ? end + 1 : end;
        }
    } else {
        const int n = ptmi_device_count();
        for (int i = 0; i < n && i < PTMI_MAX_DEVICES; i++) devices.push_back(i);
    }
    if (devices.empty()) devices.push_back(0);  // no device at all: ptmi_setup_context reports it
    if (devices.size() > PTMI_MAX_DEVICES) devices.resize(PTMI_MAX_DEVICES);
    cfg.device = devices[0];
    cfg.n_devices = (uint32_t)devices.size();
    for (size_t i = 0; i < devices.size(); i++) cfg.devices[i] = devices[i];
    g_devices = (unsigned)devices.size();
    g_log = globalVars.printLogInfos || env_uint("PTMI_LOG", 0) != 0;
    cfg.image_width = globalVars.imageWidth;
    cfg.image_height = globalVars.imageHeight;
    cfg.ray_max_depth = globalVars.rayMaxDepth;
    cfg.lights_size = globalVars.lightsSize;
    cfg.sampler = sampler == RANDOM ? PTMI_SAMPLER_RANDOM : (sampler == UNIFORM ? PTMI_SAMPLER_UNIFORM : PTMI_SAMPLER_JITTERED);
    cfg.super_sampling = globalVars.superSampling ? 1u : 0u;
    g_staged = sampler != RANDOM && !globalVars.superSampling;
    // The arithmetic of the kernel OpenCL_BuildOptions would have produced (it passes no floating-point option: OpenCL default
    // arithmetic), so that a caller of the reference API gets the reference's images bit for bit; PTMI_STRICT_ARITHMETIC=1
    // selects the other bit-exact mode (the build the same source gives with correctly rounded operations).
    cfg.flags = env_uint("PTMI_STRICT_ARITHMETIC", 0) ? 0u : PTMI_FLAG_DEFAULT_ARITHMETIC;
    // globalVars.printLogInfos is the reference's -D LOG_INFO (OpenCL.cpp:310): its kernel then checks its invariants on the device
    // (header.cl:21-48).  Here: the kernel instantiation that counts the failures of those checks (report_invariant_checks).
    if (globalVars.printLogInfos) cfg.flags |= PTMI_FLAG_SCHEDULER_STATS;
    const int rc = ptmi_setup_context(&g_ctx, &cfg);
    if (rc) fail("OpenCL_SetupContext", rc);
}

void OpenCL_InitializeMemory(GlobalVars& globalVars)
{
    if (!g_ctx) throw std::runtime_error("OpenCL_InitializeMemory before OpenCL_SetupContext");
    ptmi_scene sc{};
    sc.struct_size = sizeof sc;
    sc.bvh = reinterpret_cast<const ptmi_node*>(globalVars.bvh);
    sc.bvh_size = globalVars.bvhSize;
    sc.triangulation = reinterpret_cast<const ptmi_triangle*>(globalVars.triangulation);
    sc.triangulation_size = globalVars.triangulationSize;
    sc.lights = reinterpret_cast<const ptmi_light*>(globalVars.lights);
    sc.lights_size = globalVars.lightsSize;
    sc.materiaux = reinterpret_cast<const ptmi_material*>(globalVars.materiaux);
    sc.materiaux_size = globalVars.materiauxSize;
    sc.textures = reinterpret_cast<const ptmi_texture*>(globalVars.textures);
    sc.textures_size = globalVars.texturesSize;
    sc.textures_data = reinterpret_cast<const ptmi_uchar4*>(globalVars.texturesData);
    sc.textures_data_size = globalVars.texturesDataSize;
    sc.sky = reinterpret_cast<const ptmi_sky*>(&globalVars.sky);
    static_assert(sizeof(globalVars.cameraPosition) == sizeof(ptmi_float4), "Float4 layout");
    sc.camera_position = *reinterpret_cast<const ptmi_float4*>(&globalVars.cameraPosition);
    sc.camera_direction = *reinterpret_cast<const ptmi_float4*>(&globalVars.cameraDirection);
    sc.camera_right = *reinterpret_cast<const ptmi_float4*>(&globalVars.cameraRight);
    sc.camera_up = *reinterpret_cast<const ptmi_float4*>(&globalVars.cameraUp);
    const int rc = ptmi_initialize_memory(g_ctx, &sc);
    if (rc) fail("OpenCL_InitializeMemory", rc);
    if (const char* why = ptmi_literal_kernel_reason(g_ctx))
        std::fprintf(stderr, "[ptmi] note: %s: the reference's triangle test yields NaN distances there, so the scene is rendered by the "
                             "one-path-per-lane kernel (the reference's images bit for bit, but slower than the wavefront kernel)\n", why);
}

// What the reference prints line by line from its kernel with -D LOG_INFO, as totals after the render (only in that mode)
static void report_invariant_checks(const GlobalVars& globalVars)
{
    if (!globalVars.printLogInfos || !g_ctx) return;
    ptmi_invariant_checks c{};
    if (ptmi_get_invariant_checks(g_ctx, &c) != PTMI_OK) return;
    std::fprintf(stderr, "[ptmi] device-side checks (FullKernel_header.cl:21-48): SAMPLER - invalid pixel %llu, Kernel_Main incorrect normals %llu, "
                         "incorrect radiance L %llu, Vector_PutInSameHemisphereAs %llu, rayIntersection histogram overflow %llu\n",
                 (unsigned long long)c.sample_out_of_range, (unsigned long long)c.normal_not_facing_ray, (unsigned long long)c.negative_direct_radiance,
                 (unsigned long long)c.scattered_below_surface, (unsigned long long)c.statistics_out_of_range);
    if (c.refraction_undefined_in_reference)
        std::fprintf(stderr, "[ptmi] %llu bounce(s) the reference's source leaves undefined (FullKernel.cl:836-843 after :237: a totally reflected ray "
                             "refracted, random() == 1.0): rendered with a zero refraction direction\n",
                     (unsigned long long)c.refraction_undefined_in_reference);
}

void OpenCL_RunKernel(GlobalVars& globalVars, bool (*UpdateWindowFunc)(void), uint numImagesToRender,
                      double* pathTracingTime, double* memoryTime, double* displayTime)
{
    if (!g_ctx) throw std::runtime_error("OpenCL_RunKernel before OpenCL_SetupContext");
    *pathTracingTime = 0;
    *memoryTime = 0;
    *displayTime = 0;
    const unsigned batch = env_uint("PTMI_IMAGES_PER_LAUNCH", 1) ? env_uint("PTMI_IMAGES_PER_LAUNCH", 1) : 1;
    unsigned lookahead = env_uint("PTMI_LOOKAHEAD", g_devices > 2 ? g_devices : 2);
    if (lookahead > PTMI_MAX_SNAPSHOT_SLOTS - 2) lookahead = PTMI_MAX_SNAPSHOT_SLOTS - 2;  // one slot is the library's own
    const unsigned slots = lookahead + 1;
    // the viewer's two buffers are read into after every image (OpenCL.cpp:97-98): page-locked once, so that each readback
    // is one DMA (best effort: a buffer that cannot be locked goes through the library's staging buffer)
    (void)ptmi_pin_host_buffer(g_ctx, globalVars.imageColor, sizeof(RGBAColor) * (size_t)globalVars.imageWidth * globalVars.imageHeight);
    (void)ptmi_pin_host_buffer(g_ctx, globalVars.imageRayNb, sizeof(float) * (size_t)globalVars.imageWidth * globalVars.imageHeight);
    auto show = [&](uint slot, uint first_image, uint last_image, double t_path) {  // read image `slot` back, call the viewer
        std::clock_t start = std::clock();
        int rc = ptmi_read_snapshot(g_ctx, slot, reinterpret_cast<float*>(globalVars.imageColor), globalVars.imageRayNb);
        if (rc) fail("OpenCL_RunKernel (readback)", rc);
        const double t_mem = (double)(std::clock() - start);
        *memoryTime += t_mem;
        start = std::clock();
        if (UpdateWindowFunc) (*UpdateWindowFunc)();  // return value ignored, as in OpenCL.cpp:103
        const double t_disp = (double)(std::clock() - start);
        *displayTime += t_disp;
        if (g_log)
            std::fprintf(stderr, "[ptmi] images %u..%u of %u on %u device(s): wait %.0f, readback %.0f, display %.0f clock ticks\n", first_image,
                         last_image, numImagesToRender, g_devices, t_path, t_mem, t_disp);
    };
    // ---- bursts: B images per launch, a snapshot behind every one of them (ptmi_render_snapshots), one burst queued ahead
    // (with G devices a burst of B images is B / G iterations per device and launch: 16 per device, up to half the ring, so
    // that the launches of a multi-GPU render stay as long as a single GPU's)
    const unsigned ring = PTMI_MAX_SNAPSHOT_SLOTS - 1;
    unsigned burst = env_uint("PTMI_BURST", 16u * (g_devices > 1 ? g_devices : 1u));
    if (burst > ring / 2) burst = ring / 2;
    if (batch == 1 && burst > 1 && sampler_owns_pixels(globalVars)) {
        const uint bursts = (numImagesToRender + burst - 1) / burst;
        auto enqueue_burst = [&](uint b) {
            const uint first = b * burst;
            const uint n = numImagesToRender - first < burst ? numImagesToRender - first : burst;
            const int rc = ptmi_render_snapshots(g_ctx, first, n, first % ring);
            if (rc) fail("OpenCL_RunKernel (launch)", rc);
        };
        uint queued = 0;
        for (uint b = 0; b < bursts; b++) {
            std::clock_t start = std::clock();
            while (queued < bursts && queued <= b + 1) enqueue_burst(queued++);
            const uint first = b * burst;
            const uint n = numImagesToRender - first < burst ? numImagesToRender - first : burst;
            for (uint k = 0; k < n; k++) {
                const int rc = ptmi_read_snapshot(g_ctx, (first + k) % ring, nullptr, nullptr);  // clFinish of that image
                if (rc) fail("OpenCL_RunKernel (wait)", rc);
                const double t_path = (double)(std::clock() - start);
                *pathTracingTime += t_path;
                show((first + k) % ring, first + k, first + k, t_path);
                start = std::clock();
            }
        }
        const int rc = ptmi_read_statistics(g_ctx, globalVars.rayDepths, globalVars.rayIntersectedBBx, globalVars.rayIntersectedTri);
        if (rc) fail("OpenCL_RunKernel (statistics)", rc);
        report_invariant_checks(globalVars);
        ptmi_release(g_ctx);
        g_ctx = nullptr;
        return;
    }
    const uint steps = (numImagesToRender + batch - 1) / batch;
    // step s = images [s * batch, min((s + 1) * batch, numImagesToRender)): launch(es) + a snapshot behind them
    auto enqueue = [&](uint s) {
        const uint first = s * batch;
        const uint n = numImagesToRender - first < batch ? numImagesToRender - first : batch;
        int rc = ptmi_render(g_ctx, first, n);
        if (!rc) rc = ptmi_snapshot(g_ctx, s % slots);
        if (rc) fail("OpenCL_RunKernel (launch)", rc);
    };
    uint queued = 0;
    for (uint s = 0; s < steps; s++) {
        std::clock_t start = std::clock();
        while (queued < steps && queued <= s + lookahead) enqueue(queued++);
        int rc = ptmi_read_snapshot(g_ctx, s % slots, nullptr, nullptr);  // clFinish of image s (OpenCL.cpp:89): later steps keep running
        if (rc) fail("OpenCL_RunKernel (wait)", rc);
        const double t_path = (double)(std::clock() - start);
        *pathTracingTime += t_path;

        show(s % slots, s * batch, (s + 1) * batch < numImagesToRender ? (s + 1) * batch - 1 : numImagesToRender - 1, t_path);
    }
    const int rc = ptmi_read_statistics(g_ctx, globalVars.rayDepths, globalVars.rayIntersectedBBx, globalVars.rayIntersectedTri);
    if (rc) fail("OpenCL_RunKernel (statistics)", rc);
    report_invariant_checks(globalVars);
    ptmi_release(g_ctx);
    g_ctx = nullptr;
}

#ifndef PTMI_SHIM_WITHOUT_BVH
void BVH_Create(GlobalVars& globalVars)
{
    const uint n = globalVars.triangulationSize;
    if (n == 0) throw std::runtime_error("BVH_Create: empty triangulation");
    globalVars.bvhMaxDepth = 0;
    globalVars.bvhSize = 0;
    globalVars.bvh = new Node[2 * (size_t)n - 1];
    uint32_t size = 0, depth = 0;
    const int rc = ptmi_bvh_create(reinterpret_cast<ptmi_triangle*>(globalVars.triangulation), n,
                                   reinterpret_cast<ptmi_node*>(globalVars.bvh), &size, &depth);
    if (rc) {  // host-only entry point: its message is the library's global one, and a live render context is left alone
        delete[] globalVars.bvh;
        globalVars.bvh = nullptr;
        throw std::runtime_error(std::string("BVH_Create failed (") + std::to_string(rc) + "): " + ptmi_last_error(nullptr));
    }
    globalVars.bvhSize = size;
    globalVars.bvhMaxDepth = depth;
}

// PathTracer_BVH.cpp:362-401.  Kept literally: a leaf that lowers a minimum is not compared against the maximum (else-if),
// inner nodes count two per visit, and the "comments" out-parameter receives the leaf's stop reason as one character
// (the reference assigns the enum to a std::string).  Walked with an explicit stack, children in son1, son2 order.
void BVH_GetCharacteristics(Node* global__bvh, uint currentNodeId, uint depth, uint& BVHMaxLeafSize, uint& BVHMinLeafSize,
                            uint& BVHMaxDepth, uint& BVHMinDepth, uint& nNodes, uint& nLeafs, std::string& BVHMaxLeafSizeComments)
{
    struct Visit { uint id, depth; };
    std::vector<Visit> todo{{currentNodeId, depth}};
    while (!todo.empty()) {
        const Visit v = todo.back();
        todo.pop_back();
        const ptmi_node& node = reinterpret_cast<const ptmi_node*>(global__bvh)[v.id];
        if (node.is_leaf) {
            nLeafs++;
            if (v.depth < BVHMinDepth) BVHMinDepth = v.depth;
            else if (v.depth > BVHMaxDepth) BVHMaxDepth = v.depth;
            if (node.nb_triangles < BVHMinLeafSize) {
                BVHMinLeafSize = node.nb_triangles;
            } else if (node.nb_triangles > BVHMaxLeafSize) {
                BVHMaxLeafSize = node.nb_triangles;
                BVHMaxLeafSizeComments = (char)node.comments;
            }
            continue;
        }
        nNodes += 2;
        todo.push_back({node.son2_id, v.depth + 1});
        todo.push_back({node.son1_id, v.depth + 1});
    }
}
#endif  // PTMI_SHIM_WITHOUT_BVH

}  // namespace PathTracerNS
