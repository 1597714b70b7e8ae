// display.hip - accumulators -> displayed pixels on the device.
//
// What the reference shows and saves after every image is produced on the host from the two float buffers it has just
// read back (41.5 MB at 1080p, PathTracer_OpenCL.cpp:97-98): ConvertRGBAToBMPBuffer, Alone/PathTracer_bitmap.cpp:237-286.
// The same quantisation here, on the device, so that a viewer only has to fetch 3 bytes per pixel (6.2 MB at 1080p):
//   pixel = (int) min(sum * 255.f / n, 255.f) per channel, `min` being the Windows macro a < b ? a : b (a NaN from 0/0
//   on a never-sampled pixel shows as 255); a negative red sum marks the pixel pure red (:262 tests .x three times);
//   bytes in B, G, R order, rows padded to a multiple of 4 and zero-filled, image row 0 first.
#include <hip/hip_runtime.h>

#include "ptmi_internal.h"

namespace ptmi_dev {

__device__ __forceinline__ uint32_t display_channel(float sum, float n)
{
    const float v = sum * 255.f / n;  // IEEE multiply, then IEEE divide (no contraction, no reciprocal)
    const float m = v < 255.f ? v : 255.f;
    return (uint32_t)(int)m & 0xFFu;  // (BYTE)(int)
}

__global__ void __launch_bounds__(256) display_bgr_kernel(const float* __restrict__ image_color, const float* __restrict__ image_ray_nb,
                                                          uint8_t* __restrict__ out, const uint32_t width, const uint32_t height,
                                                          const uint32_t row_stride)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (y >= height) return;
    uint8_t* row = out + (size_t)y * row_stride;
    if (x < width) {
        const size_t p = (size_t)y * width + x;
        const float4 c = reinterpret_cast<const float4*>(image_color)[p];
        const float n = image_ray_nb[p];
        uint32_t r = 255u, g = 0u, b = 0u;
        if (!(c.x < 0)) {
            r = display_channel(c.x, n);
            g = display_channel(c.y, n);
            b = display_channel(c.z, n);
        }
        row[3 * x + 0] = (uint8_t)b;
        row[3 * x + 1] = (uint8_t)g;
        row[3 * x + 2] = (uint8_t)r;
    }
    // padding bytes of the scanline (memset 0 in the reference)
    if (x < row_stride - 3u * width) row[3u * width + x] = 0;
}

// Sum of the partial accumulators of the devices that shared a render, in device order (deterministic; the single-device
// image differs from it only by the order of these float additions).
struct ImageParts {
    const float* p[PTMI_MAX_DEVICES];
};
__global__ void __launch_bounds__(256) sum_images_kernel(float* __restrict__ out, const ImageParts parts, const uint32_t n_parts,
                                                         const size_t n_quads)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_quads) return;
    float4 s = reinterpret_cast<const float4*>(parts.p[0])[i];
    for (uint32_t k = 1; k < n_parts; k++) {
        const float4 v = reinterpret_cast<const float4*>(parts.p[k])[i];
        s.x = s.x + v.x; s.y = s.y + v.y; s.z = s.z + v.z; s.w = s.w + v.w;
    }
    reinterpret_cast<float4*>(out)[i] = s;
}

}  // namespace ptmi_dev

namespace ptmi_dev {
// dst[i] += src[i]: the counter block of a launch that ran on a stage set of its own, added when the main stream adopts it
__global__ void add_counters_kernel(unsigned long long* __restrict__ dst, const unsigned long long* __restrict__ src, uint32_t n)
{
    const uint32_t i = threadIdx.x;
    if (i < n) dst[i] += src[i];
}
}  // namespace ptmi_dev

namespace ptmi_internal {

int launch_add_counters(unsigned long long* dst, const unsigned long long* src, uint32_t n, void* stream, std::string* err)
{
    hipLaunchKernelGGL(ptmi_dev::add_counters_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dst, src, n);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (err) *err = std::string("add_counters_kernel launch: ") + hipGetErrorString(e);
        return PTMI_ERR_HIP;
    }
    return PTMI_OK;
}

int launch_sum_images(float* out, const float* const* parts, uint32_t n_parts, size_t n_floats, void* stream, std::string* err)
{
    if (n_parts == 0 || n_parts > PTMI_MAX_DEVICES || (n_floats & 3u)) {
        if (err) *err = "launch_sum_images: bad part count or a float count that is not a multiple of 4";
        return PTMI_ERR_INVALID_ARGUMENT;
    }
    ptmi_dev::ImageParts ip{};
    for (uint32_t k = 0; k < n_parts; k++) ip.p[k] = parts[k];
    const size_t n_quads = n_floats / 4;
    hipLaunchKernelGGL(ptmi_dev::sum_images_kernel, dim3((unsigned)((n_quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, ip,
                       n_parts, n_quads);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (err) *err = std::string("sum_images_kernel launch: ") + hipGetErrorString(e);
        return PTMI_ERR_HIP;
    }
    return PTMI_OK;
}

int launch_display_bgr(const float* image_color, const float* image_ray_nb, uint8_t* out, uint32_t width, uint32_t height,
                       uint32_t row_stride, void* stream, std::string* err)
{
    const dim3 grid((width + 255u) / 256u, height), block(256);
    hipLaunchKernelGGL(ptmi_dev::display_bgr_kernel, grid, block, 0, (hipStream_t)stream, image_color, image_ray_nb, out, width,
                       height, row_stride);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (err) *err = std::string("display_bgr_kernel launch: ") + hipGetErrorString(e);
        return PTMI_ERR_HIP;
    }
    return PTMI_OK;
}

}  // namespace ptmi_internal
