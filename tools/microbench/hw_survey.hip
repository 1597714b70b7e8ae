// hw_survey.hip - how far are the hardware approximations the reference's DEFAULT OpenCL build computes with from the
// correctly rounded functions?  The default build (no -cl-fp32-correctly-rounded-divide-sqrt) divides through
// v_rcp_f32 of the divisor's mantissa and takes square roots with v_sqrt_f32; neither is correctly
// rounded, and the CPU oracle reproduces both from the tables this tool measures.
//   hw_survey rcp  out.bin   every mantissa of [0.5, 1)  (what v_frexp_mant_f32 hands to v_rcp_f32): 2^23 inputs
//   hw_survey sqrt out.bin   every mantissa at both exponent parities, [1, 2) and [2, 4): 2^24 inputs
//   hw_survey rsq  out.bin   the same inputs through v_rsq_f32 (what rsq_survey.hip measured in round 2)
// Prints the histogram of (hardware - reference formula) in ulps and writes the deviations as int8.
// Reference formulas (the oracle evaluates the very same expressions): (float)(1.0 / (double)x), (float)sqrt((double)x),
// (float)(1.0 / sqrt((double)x)).
// build: hipcc --offload-arch=gfx950 -O2 hw_survey.hip -o hw_survey
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

enum Mode { RCP = 0, SQRT = 1, RSQ = 2 };

__host__ __device__ static inline uint32_t input_bits(int mode, uint32_t i)
{
    if (mode == RCP) return 0x3F000000u | (i & 0x7FFFFFu);
    return (i & 0x800000u ? 0x40000000u : 0x3F800000u) | (i & 0x7FFFFFu);
}

__global__ void survey(uint32_t* out, uint32_t n, int mode)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = __uint_as_float(input_bits(mode, i));
    float r;
    if (mode == RCP) r = __builtin_amdgcn_rcpf(x);
    else if (mode == SQRT) r = __builtin_amdgcn_sqrtf(x);
    else r = __builtin_amdgcn_rsqf(x);
    out[i] = __float_as_uint(r);
}

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    const int mode = !std::strcmp(argv[1], "rcp") ? RCP : !std::strcmp(argv[1], "sqrt") ? SQRT : RSQ;
    const uint32_t n = mode == RCP ? 1u << 23 : 1u << 24;
    uint32_t* d = nullptr;
    if (hipMalloc(&d, (size_t)n * 4) != hipSuccess) return 1;
    hipLaunchKernelGGL(survey, dim3(n / 256), dim3(256), 0, 0, d, n, mode);
    std::vector<uint32_t> h(n);
    if (hipMemcpy(h.data(), d, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    std::vector<int8_t> dev(n);
    long hist[9] = {0};
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t bits = input_bits(mode, i);
        float x;
        std::memcpy(&x, &bits, 4);
        const float ref = mode == RCP ? (float)(1.0 / (double)x) : mode == SQRT ? (float)std::sqrt((double)x) : (float)(1.0 / std::sqrt((double)x));
        uint32_t eb;
        std::memcpy(&eb, &ref, 4);
        int dlt = (int)((long)h[i] - (long)eb);
        if (dlt < -4) dlt = -4;
        if (dlt > 4) dlt = 4;
        dev[i] = (int8_t)dlt;
        hist[dlt + 4]++;
    }
    for (int k = 0; k < 9; k++) std::printf("%s: hardware - reference = %+d ulp: %ld (%.3f %%)\n", argv[1], k - 4, hist[k], 100.0 * hist[k] / n);
    if (argc > 2) {
        FILE* f = std::fopen(argv[2], "wb");
        if (f) { std::fwrite(dev.data(), 1, n, f); std::fclose(f); }
    }
    return 0;
}
