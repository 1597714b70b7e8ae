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
// the product's restatement of the reference's default arithmetic (round 3): compiled here in that mode
#define PTMI_DEFAULT_ARITHMETIC 1
#include "ptmi_device.hpp"

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

// ---- round 3: the reference's DEFAULT arithmetic ------------------------------------------------------------------------
// out_product[k * n + i]: the product's operators (ptmi_device.hpp, default-arithmetic mode) on (a[i], b[i]), in the order of
// arith_probe.cl; out_compiler: arith_probe.cl itself, compiled by the image's OpenCL compiler with the reference's flags
// and loaded from `hsaco_path`; out_hw[0..2]: v_rcp_f32(frexp_mant(a)), v_sqrt_f32(|a|) for the oracle's tables.
__global__ void arith_product_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                     float* __restrict__ hw, unsigned n)
{
    using namespace ptmi_dev_da;
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = a[i], y = b[i];
    constexpr DivC k255 = make_divc(255.f), k3 = make_divc(3.f), k155 = make_divc(1.55f);
    const DivC k1920 = make_divc((float)1920), k90 = make_divc((float)90);
    out[0 * (size_t)n + i] = fdiv(x, y);
    out[1 * (size_t)n + i] = frcp(x);
    out[2 * (size_t)n + i] = fdiv(x, k255);
    out[3 * (size_t)n + i] = fdiv(x, k3);
    out[4 * (size_t)n + i] = fdiv(x, k155);
    out[5 * (size_t)n + i] = fdiv(x, k1920);
    out[6 * (size_t)n + i] = fdiv(x, k90);
    out[7 * (size_t)n + i] = fsqrt(x);
    out[8 * (size_t)n + i] = length(v4(x, y, x * 0.5f, 0.0f));
    out[9 * (size_t)n + i] = mad(x, y, 1.0f);
    hw[0 * (size_t)n + i] = __builtin_amdgcn_rcpf(__builtin_amdgcn_frexp_mantf(x));
    hw[1 * (size_t)n + i] = __builtin_amdgcn_sqrtf(__builtin_fabsf(x));
}

extern "C" int device_arith_probe(const char* hsaco_path, const float* a, const float* b, unsigned n, float* out_product,
                                  float* out_compiler, float* out_hw)
{
    constexpr int kOps = 10;
    float *da = nullptr, *db = nullptr, *dp = nullptr, *dc = nullptr, *dh = nullptr;
    if (hipMalloc(&da, (size_t)n * 4) != hipSuccess || hipMalloc(&db, (size_t)n * 4) != hipSuccess ||
        hipMalloc(&dp, (size_t)n * 4 * kOps) != hipSuccess || hipMalloc(&dc, (size_t)n * 4 * kOps) != hipSuccess ||
        hipMalloc(&dh, (size_t)n * 4 * 2) != hipSuccess)
        return -1;
    if (hipMemcpy(da, a, (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess) return -2;
    if (hipMemcpy(db, b, (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess) return -2;
    hipLaunchKernelGGL(arith_product_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, dp, dh, n);
    hipModule_t mod;
    hipFunction_t fn;
    if (hipModuleLoad(&mod, hsaco_path) != hipSuccess) return -4;
    if (hipModuleGetFunction(&fn, mod, "arith_probe") != hipSuccess) return -5;
    // (kernelParams, not a raw argument buffer: the runtime then fills the OpenCL kernel's hidden arguments - global offsets -
    // from the code object's metadata, as in ref_gpu_runner.cpp)
    void* args[4] = {&da, &db, &dc, &n};
    if (hipModuleLaunchKernel(fn, (n + 255) / 256, 1, 1, 256, 1, 1, 0, nullptr, args, nullptr) != hipSuccess) return -6;
    if (hipDeviceSynchronize() != hipSuccess) return -7;
    if (hipMemcpy(out_product, dp, (size_t)n * 4 * kOps, hipMemcpyDeviceToHost) != hipSuccess) return -3;
    if (hipMemcpy(out_compiler, dc, (size_t)n * 4 * kOps, hipMemcpyDeviceToHost) != hipSuccess) return -3;
    if (hipMemcpy(out_hw, dh, (size_t)n * 4 * 2, hipMemcpyDeviceToHost) != hipSuccess) return -3;
    (void)hipModuleUnload(mod);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dp); (void)hipFree(dc); (void)hipFree(dh);
    return 0;
}
