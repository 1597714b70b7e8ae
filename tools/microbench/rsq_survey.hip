// rsq_survey.hip - how far is the hardware reciprocal square root (v_rsq_f32, what the ROCm OpenCL library's normalize()
// multiplies by) from the correctly rounded 1/sqrt(x)?  Every mantissa at both exponent parities ([1,2) and [2,4)): 2^24
// inputs.  Prints the histogram of (hardware - correctly rounded) in ulps and writes the deviations as int8 to argv[1].
// Diagnostic for DESIGN.md "Numerics"; build: hipcc --offload-arch=gfx950 -O2 rsq_survey.hip -o rsq_survey
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void survey(uint32_t* out, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t bits = (i & 0x800000u ? 0x40000000u : 0x3F800000u) | (i & 0x7FFFFFu);
    out[i] = __float_as_uint(__builtin_amdgcn_rsqf(__uint_as_float(bits)));
}

int main(int argc, char** argv)
{
    const uint32_t n = 1u << 24;
    uint32_t* d = nullptr;
    if (hipMalloc(&d, n * 4) != hipSuccess) return 1;
    hipLaunchKernelGGL(survey, dim3(n / 256), dim3(256), 0, 0, d, n);
    std::vector<uint32_t> h(n);
    if (hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    std::vector<int8_t> dev(n);
    long hist[9] = {0};
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t bits = (i & 0x800000u ? 0x40000000u : 0x3F800000u) | (i & 0x7FFFFFu);
        float x;
        std::memcpy(&x, &bits, 4);
        const float exact = (float)(1.0 / std::sqrt((double)x));  // double result rounded once more: correct but for rare ties
        uint32_t eb;
        std::memcpy(&eb, &exact, 4);
        int dlt = (int)((long)h[i] - (long)eb);
        if (dlt < -4) dlt = -4;
        if (dlt > 4) dlt = 4;
        dev[i] = (int8_t)dlt;
        hist[dlt + 4]++;
    }
    for (int k = 0; k < 9; k++) std::printf("hardware - correctly rounded = %+d ulp: %ld (%.3f %%)\n", k - 4, hist[k], 100.0 * hist[k] / n);
    if (argc > 1) {
        FILE* f = std::fopen(argv[1], "wb");
        if (f) { std::fwrite(dev.data(), 1, n, f); std::fclose(f); }
    }
    return 0;
}
