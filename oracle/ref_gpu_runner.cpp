// ref_gpu_runner.cpp - TEST INFRASTRUCTURE: launches the REFERENCE kernel on the GPU.
//
// oracle/_ref/ref_kernel_<cfg>.hsaco is Kernel/PathTracer_FullKernel.cl of the
// reference compiled UNMODIFIED for gfx950 by the image's own clang -x cl with
// the ROCm OpenCL device libraries (oracle/Makefile, target ref-kernels).  This
// launcher plays the part of OpenCL_InitializeMemory + OpenCL_RunKernel
// (Controleur/PathTracer_OpenCL.cpp:149-198, 56-140) for that code object using
// the HIP module API: upload the raw struct arrays, set the 18 kernel arguments
// in the order of header.cl:524-546, launch once per iteration over a W x H
// NDRange, read the accumulators back.  It is how the oracle and the HIP
// integrator are pinned to the reference itself, and how "the reference's
// OpenCL kernel on MI355X" is timed.
//
// Deviations from the reference host code, both needed on a GPU and neither
// changing a work-item's result: the accumulators are zeroed (the reference
// relies on fresh memory being zero), and the work-group size is a parameter
// (the reference uses 1x1, which would idle 63 of 64 lanes).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "ptmi_scene.h"

namespace {
std::string g_err;
#define RT(expr)                                                                          \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess) {                                                          \
            g_err = std::string(#expr) + ": " + hipGetErrorString(e__);                   \
            return -1;                                                                    \
        }                                                                                 \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc_copy(const void* host, size_t bytes)
    {
        RT(hipMalloc(&p, bytes ? bytes : 1));  // reference: std::max(size, 1), OpenCL.cpp:165
        if (bytes && host) RT(hipMemcpy(p, host, bytes, hipMemcpyHostToDevice));
        return 0;
    }
    int alloc_zero(size_t bytes)
    {
        RT(hipMalloc(&p, bytes ? bytes : 1));
        RT(hipMemset(p, 0, bytes ? bytes : 1));
        return 0;
    }
};
}  // namespace

extern "C" const char* ref_gpu_last_error(void) { return g_err.c_str(); }

struct ref_gpu_job {
    const char* hsaco_path;
    uint32_t width, height, ray_max_depth;  // must match the -D values the code object was built with
    uint32_t local_x, local_y;              // work-group size; must divide width / height
    uint32_t first_iteration, n_iterations;
    float camera_position[4], camera_direction[4], camera_right[4], camera_up[4];
    const void* bvh;            uint64_t bvh_bytes;
    const void* triangulation;  uint64_t triangulation_bytes;
    const void* lights;         uint64_t lights_bytes;
    const void* materiaux;      uint64_t materiaux_bytes;
    const void* textures;       uint64_t textures_bytes;
    const void* textures_data;  uint64_t textures_data_bytes;
    const void* sky;
    float* image_color;    // out, float[4*W*H]
    float* image_ray_nb;   // out, float[W*H]
    uint32_t* ray_depths;  // out, [depth+1]
    uint32_t* ray_bbx;     // out, [5000]
    uint32_t* ray_tri;     // out, [5000]
    double kernel_ms;      // out: sum of launch durations (HIP events)
};

extern "C" int ref_gpu_run(ref_gpu_job* job)
{
    const uint32_t W = job->width, H = job->height;
    if (!W || !H || !job->local_x || !job->local_y || W % job->local_x || H % job->local_y) {
        g_err = "work-group size must divide the image (the kernel has no bounds check)";
        return -1;
    }
    hipModule_t mod;
    hipFunction_t fn;
    RT(hipModuleLoad(&mod, job->hsaco_path));
    RT(hipModuleGetFunction(&fn, mod, "Kernel_Main"));

    const size_t npix = (size_t)W * H;
    DevBuf color, count, imgv, depths, bbx, tri, bvh, tris, lights, mats, texs, texels, sky;
    if (color.alloc_zero(npix * 16) || count.alloc_zero(npix * 4) || imgv.alloc_zero(npix * 16) ||
        depths.alloc_zero((job->ray_max_depth + 1) * 4) || bbx.alloc_zero(PTMI_MAX_INTERSECTION_NUMBER * 4) ||
        tri.alloc_zero(PTMI_MAX_INTERSECTION_NUMBER * 4) || bvh.alloc_copy(job->bvh, job->bvh_bytes) ||
        tris.alloc_copy(job->triangulation, job->triangulation_bytes) ||
        lights.alloc_copy(job->lights, job->lights_bytes) || mats.alloc_copy(job->materiaux, job->materiaux_bytes) ||
        texs.alloc_copy(job->textures, job->textures_bytes) ||
        texels.alloc_copy(job->textures_data, job->textures_data_bytes) || sky.alloc_copy(job->sky, sizeof(ptmi_sky)))
        return -1;

    hipEvent_t e0, e1;
    RT(hipEventCreate(&e0));
    RT(hipEventCreate(&e1));
    job->kernel_ms = 0;
    struct alignas(16) F4 { float v[4]; };
    F4 cp, cd, cr, cu;
    std::memcpy(cp.v, job->camera_position, 16);
    std::memcpy(cd.v, job->camera_direction, 16);
    std::memcpy(cr.v, job->camera_right, 16);
    std::memcpy(cu.v, job->camera_up, 16);
    for (uint32_t it = job->first_iteration; it < job->first_iteration + job->n_iterations; it++) {
        uint32_t iter = it;
        void* args[18] = {&iter,    &cp,     &cd,      &cr,     &cu,     &color.p, &count.p, &imgv.p, &depths.p,
                          &bbx.p,   &tri.p,  &bvh.p,   &tris.p, &lights.p, &mats.p, &texs.p, &texels.p, &sky.p};
        RT(hipEventRecord(e0, nullptr));
        RT(hipModuleLaunchKernel(fn, W / job->local_x, H / job->local_y, 1, job->local_x, job->local_y, 1, 0, nullptr,
                                 args, nullptr));
        RT(hipEventRecord(e1, nullptr));
        RT(hipEventSynchronize(e1));  // clFinish after every launch, OpenCL.cpp:89
        float ms = 0;
        RT(hipEventElapsedTime(&ms, e0, e1));
        job->kernel_ms += ms;
    }
    if (job->image_color) RT(hipMemcpy(job->image_color, color.p, npix * 16, hipMemcpyDeviceToHost));
    if (job->image_ray_nb) RT(hipMemcpy(job->image_ray_nb, count.p, npix * 4, hipMemcpyDeviceToHost));
    if (job->ray_depths) RT(hipMemcpy(job->ray_depths, depths.p, (job->ray_max_depth + 1) * 4, hipMemcpyDeviceToHost));
    if (job->ray_bbx) RT(hipMemcpy(job->ray_bbx, bbx.p, PTMI_MAX_INTERSECTION_NUMBER * 4, hipMemcpyDeviceToHost));
    if (job->ray_tri) RT(hipMemcpy(job->ray_tri, tri.p, PTMI_MAX_INTERSECTION_NUMBER * 4, hipMemcpyDeviceToHost));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    RT(hipModuleUnload(mod));
    return 0;
}
