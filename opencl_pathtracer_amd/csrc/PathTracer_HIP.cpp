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
//
// Errors: any failure throws std::runtime_error with the library's message, as every cl error does in the
// reference (OpenCL_ErrorHandling, OpenCL.cpp:407-486); PathTracer_Main catches std::exception (PathTracer.cpp:99).
// State: one context per process in a file-scope variable, like the reference's file-scope cl objects
// (OpenCL.cpp:19-47): one render at a time, released at the end of OpenCL_RunKernel.
//
// Environment knobs (optional): PTMI_DEVICE = HIP device ordinal (default 0);
// PTMI_IMAGES_PER_LAUNCH = iterations rendered per launch and per callback (default 1 = reference behaviour).
#ifdef PTMI_USE_REFERENCE_HEADERS
#include "PathTracer_OpenCL.h"  // the reference's own header (needs CL/cl.h and -fms-extensions for ALIGN)
#include "PathTracer_BVH.h"
#else
#include "pathtracer_backend.hpp"
#endif

#include <cstdlib>
#include <ctime>
#include <stdexcept>
#include <string>

#include "ptmi.h"

namespace PathTracerNS {

namespace {

ptmi_ctx* g_ctx = nullptr;

[[noreturn]] void fail(const char* where, int code)
{
    std::string msg = std::string(where) + " failed (" + std::to_string(code) + "): " + ptmi_last_error(g_ctx);
    if (g_ctx) {
        ptmi_release(g_ctx);
        g_ctx = nullptr;
    }
    throw std::runtime_error(msg);
}

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
    cfg.device = (int)env_uint("PTMI_DEVICE", 0);
    cfg.image_width = globalVars.imageWidth;
    cfg.image_height = globalVars.imageHeight;
    cfg.ray_max_depth = globalVars.rayMaxDepth;
    cfg.lights_size = globalVars.lightsSize;
    cfg.sampler = sampler == RANDOM ? PTMI_SAMPLER_RANDOM : (sampler == UNIFORM ? PTMI_SAMPLER_UNIFORM : PTMI_SAMPLER_JITTERED);
    cfg.super_sampling = globalVars.superSampling ? 1u : 0u;
    cfg.flags = 0;
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
}

void OpenCL_RunKernel(GlobalVars& globalVars, bool (*UpdateWindowFunc)(void), uint numImagesToRender,
                      double* pathTracingTime, double* memoryTime, double* displayTime)
{
    if (!g_ctx) throw std::runtime_error("OpenCL_RunKernel before OpenCL_SetupContext");
    *pathTracingTime = 0;
    *memoryTime = 0;
    *displayTime = 0;
    const unsigned batch = env_uint("PTMI_IMAGES_PER_LAUNCH", 1) ? env_uint("PTMI_IMAGES_PER_LAUNCH", 1) : 1;
    uint imageId = 0;
    while (imageId < numImagesToRender) {
        const uint n = numImagesToRender - imageId < batch ? numImagesToRender - imageId : batch;
        std::clock_t start = std::clock();
        int rc = ptmi_render(g_ctx, imageId, n);
        if (!rc) rc = ptmi_synchronize(g_ctx);
        if (rc) fail("OpenCL_RunKernel (launch)", rc);
        *pathTracingTime += std::clock() - start;

        start = std::clock();
        rc = ptmi_read_image(g_ctx, reinterpret_cast<float*>(globalVars.imageColor), globalVars.imageRayNb);
        if (rc) fail("OpenCL_RunKernel (readback)", rc);
        *memoryTime += std::clock() - start;

        start = std::clock();
        if (UpdateWindowFunc) (*UpdateWindowFunc)();  // return value ignored, as in OpenCL.cpp:103
        *displayTime += std::clock() - start;
        imageId += n;
    }
    const int rc = ptmi_read_statistics(g_ctx, globalVars.rayDepths, globalVars.rayIntersectedBBx, globalVars.rayIntersectedTri);
    if (rc) fail("OpenCL_RunKernel (statistics)", rc);
    ptmi_release(g_ctx);
    g_ctx = nullptr;
}

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
    if (rc) {
        delete[] globalVars.bvh;
        globalVars.bvh = nullptr;
        fail("BVH_Create", rc);
    }
    globalVars.bvhSize = size;
    globalVars.bvhMaxDepth = depth;
}

}  // namespace PathTracerNS
