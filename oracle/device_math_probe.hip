// device_math_probe.hip - TEST INFRASTRUCTURE: the platform's math on the GPU, for the numerics contract of DESIGN.md.
//
// The reference kernel compiled for gfx950 calls the ROCm device library: OpenCL sin()/cos() are __ocml_sin_f32 /
// __ocml_cos_f32, normalize() multiplies by the hardware instruction v_rsq_f32.  The integrator and the CPU oracle restate
// the first (include/ptmi_detmath.h) and reproduce the second from a measured table (tests/golden/rsq_gfx950.npz).  This
// probe evaluates, for an array of inputs: the device library's sinf/cosf (the same __ocml functions), the restatement
// compiled as device code, and v_rsq_f32 - so that a GPU test can require all of them to agree bit for bit.
// Built by oracle/Makefile (target probe) into oracle/build/libdevice_math_probe.so; never linked by the product.
#include <hip/hip_runtime.h>

#include "ptmi_detmath.h"

__global__ void probe_kernel(const float* __restrict__ x, float* __restrict__ lib_sin, float* __restrict__ lib_cos,
                             float* __restrict__ port_sin, float* __restrict__ port_cos, float* __restrict__ hw_rsq, unsigned n)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    lib_sin[i] = sinf(v);
    lib_cos[i] = cosf(v);
    float s, c;
    ptmi_sincosf(v, &s, &c);
    port_sin[i] = s;
    port_cos[i] = c;
    hw_rsq[i] = __builtin_amdgcn_rsqf(v);
}

extern "C" int device_math_probe(const float* x, unsigned n, float* lib_sin, float* lib_cos, float* port_sin, float* port_cos, float* hw_rsq)
{
    float* d = nullptr;
    if (hipMalloc(&d, (size_t)n * 6 * sizeof(float)) != hipSuccess) return -1;
    if (hipMemcpy(d, x, (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess) return -2;
    hipLaunchKernelGGL(probe_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, d, d + n, d + 2 * (size_t)n, d + 3 * (size_t)n, d + 4 * (size_t)n,
                       d + 5 * (size_t)n, n);
    float* outs[5] = {lib_sin, lib_cos, port_sin, port_cos, hw_rsq};
    for (int k = 0; k < 5; k++)
        if (hipMemcpy(outs[k], d + (size_t)(k + 1) * n, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) return -3;
    (void)hipFree(d);
    return 0;
}
