// kernels.hip - the integrator kernel for gfx950 (MI355X).
//
// One launch renders a RANGE of iterations (samples per pixel) for every pixel:
// a lane owns one pixel for the whole launch, runs its iterations back to back
// with the radiance sum kept in registers and touches the framebuffer once
// (the reference launches once per iteration and read-modify-writes 40 bytes per
// path, PathTracer_OpenCL.cpp:76-89 / FullKernel.cl:1339-1345).  A wavefront
// covers an 8x8 pixel tile so that camera rays of a wave walk the same nodes.
// The traversal stack (30 child references per lane, FullKernel.cl:627) lives in
// LDS, lane-interleaved so every push/pop is conflict free.
//
// Result contract: per pixel and iteration the SAME radiance bits as the
// reference algorithm evaluated with the numerics of ptmi_device.hpp, added to
// the accumulator in iteration order => images equal the CPU checker's bit for
// bit for the JITTERED/UNIFORM samplers.
#include <hip/hip_runtime.h>

#include "ptmi_literal_path.hpp"

namespace PTMI_DEV_NS {

template <bool PRE>
__global__ void __launch_bounds__(kBlock) render_kernel(const DScene sc, const uint32_t first_iteration,
                                                        const uint32_t n_iterations, const uint32_t iteration_stride)
{
    __shared__ uint32_t stack_mem[kStackDepth * kBlock];
    __shared__ unsigned long long block_counters[C_COUNT];

    const uint32_t tid = threadIdx.x;
    if (tid < C_COUNT) block_counters[tid] = 0;
    __syncthreads();

    // wave -> 8x8 pixel tile; block (4 waves) -> 16x16
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    const uint32_t gx = blockIdx.x * 16u + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t gy = blockIdx.y * 16u + (wave >> 1) * 8u + (lane >> 3);
    const bool valid = gx < sc.width && gy < sc.height;
    uint32_t* stack = &stack_mem[tid];

    if (valid) {
        unsigned long long n_bbx = 0, n_tri = 0;
        uint32_t n_seg = 0, n_shadow = 0, n_hits = 0;
        const bool owns_pixel = sc.sampler != PTMI_SAMPLER_RANDOM;
        const uint32_t own_offset = gy * sc.width + gx;
        V4 sum = v4(0, 0, 0, 0);
        float count = 0;
        if (owns_pixel) {
            sum = v4(*reinterpret_cast<const float4*>(&sc.image_color[4 * own_offset]));
            count = sc.image_ray_nb[own_offset];
        }
        for (uint32_t k = 0; k < n_iterations; k++) {
            const uint32_t it = first_iteration + k * iteration_stride;
            float sx, sy;
            uint32_t depth;
            PathCounters pc;
            const V4 radiance = trace_path<PRE>(sc, gx, gy, it, stack, sx, sy, depth, n_seg, n_shadow, pc);
            n_bbx += pc.bbx;
            n_tri += pc.tri;
            n_hits += depth;
            if (sc.hist_depths) {  // FullKernel.cl:1319-1331
                atomicAdd(&sc.hist_depths[depth], 1u);
                if (pc.bbx < PTMI_MAX_INTERSECTION_NUMBER) atomicAdd(&sc.hist_bbx[pc.bbx], 1u);
                if (pc.tri < PTMI_MAX_INTERSECTION_NUMBER) atomicAdd(&sc.hist_tri[pc.tri], 1u);
            }
            if (owns_pixel) {
                // JITTERED / UNIFORM: the sample always lands on the work-item's own pixel
                sum = sum + radiance;  // :1340
                count = count + 1.f;   // :1342
            } else {
                // RANDOM: samples land anywhere; the reference races here (:1339-1345), we add atomically
                const uint32_t off = sample_pixel(sc, sx, sy);
                atomicAdd(&sc.image_color[4 * off + 0], radiance.x);
                atomicAdd(&sc.image_color[4 * off + 1], radiance.y);
                atomicAdd(&sc.image_color[4 * off + 2], radiance.z);
                atomicAdd(&sc.image_color[4 * off + 3], radiance.w);
                atomicAdd(&sc.image_ray_nb[off], 1.f);
            }
        }
        if (owns_pixel) {
            *reinterpret_cast<float4*>(&sc.image_color[4 * own_offset]) = make_float4(sum.x, sum.y, sum.z, sum.w);
            sc.image_ray_nb[own_offset] = count;
        }
        atomicAdd(&block_counters[C_PATHS], (unsigned long long)n_iterations);
        atomicAdd(&block_counters[C_SEGMENTS], (unsigned long long)n_seg);
        atomicAdd(&block_counters[C_HITS], (unsigned long long)n_hits);
        atomicAdd(&block_counters[C_SHADOW], (unsigned long long)n_shadow);
        atomicAdd(&block_counters[C_BBX], n_bbx);
        atomicAdd(&block_counters[C_TRI], n_tri);
    }
    __syncthreads();
    if (tid <= C_TRI) atomicAdd(&sc.counters[tid], block_counters[tid]);
}

}  // namespace PTMI_DEV_NS

namespace ptmi_internal {

int PTMI_ARITH(launch_render)(const DScene& sc, uint32_t first_iteration, uint32_t n_iterations, uint32_t iteration_stride, void* stream,
                  std::string* err)
{
    if (n_iterations == 0) return PTMI_OK;
    const dim3 grid((sc.width + 15u) / 16u, (sc.height + 15u) / 16u);
    if (sc.tris_precomputed)
        hipLaunchKernelGGL(PTMI_DEV_NS::render_kernel<true>, grid, dim3(PTMI_DEV_NS::kBlock), 0, (hipStream_t)stream, sc,
                           first_iteration, n_iterations, iteration_stride);
    else
        hipLaunchKernelGGL(PTMI_DEV_NS::render_kernel<false>, grid, dim3(PTMI_DEV_NS::kBlock), 0, (hipStream_t)stream, sc,
                           first_iteration, n_iterations, iteration_stride);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (err) *err = std::string("render_kernel launch: ") + hipGetErrorString(e);
        return PTMI_ERR_HIP;
    }
    return PTMI_OK;
}

}  // namespace ptmi_internal
