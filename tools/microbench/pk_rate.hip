// pk_rate.hip - issue cost of the VALU instructions the integrator's hot loops are made of, on gfx950: cycles a wave64
// instruction holds its SIMD (8 waves per SIMD, 8 independent chains per wave, so the port is the only limit).
// Finding (MI355X): plain 32-bit operations (v_fma_f32, v_mul_f32, v_add_u32, v_cndmask_b32, v_cmp, v_mov_b32, v_max3_f32)
// take ~2 cycles; packed fp32 (v_pk_fma_f32, v_pk_mul_f32) and 64-bit integer forms ~4; v_rcp_f32 / v_mul_lo_u32 ~8.
// So packing two fp32 operations into one v_pk instruction saves nothing, and "4 cycles x SQ_INSTS_VALU" overstates how busy
// the VALUs are: SQ_ACTIVE_INST_VALU is the honest counter.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define EIGHT(INS, C, ...) \
    asm volatile(INS : "+" C(a0) : __VA_ARGS__); asm volatile(INS : "+" C(a1) : __VA_ARGS__); asm volatile(INS : "+" C(a2) : __VA_ARGS__); \
    asm volatile(INS : "+" C(a3) : __VA_ARGS__); asm volatile(INS : "+" C(b0) : __VA_ARGS__); asm volatile(INS : "+" C(b1) : __VA_ARGS__); \
    asm volatile(INS : "+" C(b2) : __VA_ARGS__); asm volatile(INS : "+" C(b3) : __VA_ARGS__);

static const char* const kNames[] = {"v_pk_fma_f32", "v_fma_f32", "v_mul_f32", "v_pk_mul_f32", "v_add_u32", "v_cndmask_b32", "v_mov_b32",
                                     "v_max3_f32", "v_rcp_f32", "v_mul_lo_u32", "v_lshl_add_u64", "v_cmp_gt_f32 (e64)", "v_mbcnt_lo_u32_b32"};
constexpr int kModes = 13;

template <int MODE>
__global__ void __launch_bounds__(256) rate(float* out, int iters, float seed)
{
    const float m = 0.999f, c = 1e-3f;
    const f2 m2 = {0.999f, 1.001f}, c2 = {1e-3f, -1e-3f};
    float r = 0;
    if (MODE == 0 || MODE == 3 || MODE == 10) {
        f2 a0 = {seed, seed + 1}, a1 = a0 * 2.f, a2 = a0 * 3.f, a3 = a0 * 4.f, b0 = a0 * 0.5f, b1 = a1 * 0.5f, b2 = a2 * 0.5f, b3 = a3 * 0.5f;
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (MODE == 0) { EIGHT("v_pk_fma_f32 %0, %0, %1, %2", "v", "v"(m2), "v"(c2)) }
                if (MODE == 3) { EIGHT("v_pk_mul_f32 %0, %0, %1", "v", "v"(m2)) }
                if (MODE == 10) { EIGHT("v_lshl_add_u64 %0, %0, 0, %1", "v", "v"(c2)) }
            }
        }
        const f2 s = a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3;
        r = s.x + s.y;
    } else {
        float a0 = seed, a1 = seed * 2, a2 = seed * 3, a3 = seed * 4, b0 = seed * 5, b1 = seed * 6, b2 = seed * 7, b3 = seed * 8;
        unsigned long long mask = 0;
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (MODE == 1) { EIGHT("v_fma_f32 %0, %0, %1, %2", "v", "v"(m), "v"(c)) }
                if (MODE == 2) { EIGHT("v_mul_f32 %0, %0, %1", "v", "v"(m)) }
                if (MODE == 4) { EIGHT("v_add_u32 %0, %0, %1", "v", "v"(c)) }
                if (MODE == 5) { EIGHT("v_cndmask_b32 %0, %0, %1, vcc", "v", "v"(c)) }
                if (MODE == 6) { EIGHT("v_mov_b32 %0, %1", "v", "v"(c)) }
                if (MODE == 7) { EIGHT("v_max3_f32 %0, %0, %1, %2", "v", "v"(m), "v"(c)) }
                if (MODE == 8) { EIGHT("v_rcp_f32 %0, %0", "v", "v"(c)) }
                if (MODE == 9) { EIGHT("v_mul_lo_u32 %0, %0, %1", "v", "v"(c)) }
                if (MODE == 11) {
                    asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(mask) : "v"(a0), "v"(a1)); asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(mask) : "v"(a1), "v"(a2));
                    asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(mask) : "v"(a2), "v"(a3)); asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(mask) : "v"(a3), "v"(b0));
                    asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(mask) : "v"(b0), "v"(b1)); asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(mask) : "v"(b1), "v"(b2));
                    asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(mask) : "v"(b2), "v"(b3)); asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(mask) : "v"(b3), "v"(a0));
                }
                if (MODE == 12) { EIGHT("v_mbcnt_lo_u32_b32 %0, -1, %0", "v", "v"(c)) }
            }
        }
        r = a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3 + (float)(mask & 1ull);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
static float run(int blocks, float* out, int iters, hipEvent_t e0, hipEvent_t e1)
{
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 8, iters = 10000;
    float* out;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms[kModes] = {run<0>(blocks, out, iters, e0, e1), run<1>(blocks, out, iters, e0, e1), run<2>(blocks, out, iters, e0, e1),
                        run<3>(blocks, out, iters, e0, e1), run<4>(blocks, out, iters, e0, e1), run<5>(blocks, out, iters, e0, e1),
                        run<6>(blocks, out, iters, e0, e1), run<7>(blocks, out, iters, e0, e1), run<8>(blocks, out, iters, e0, e1),
                        run<9>(blocks, out, iters, e0, e1), run<10>(blocks, out, iters, e0, e1), run<11>(blocks, out, iters, e0, e1),
                        run<12>(blocks, out, iters, e0, e1)};
    const double per_simd = (double)blocks * 4 * iters * 64 / (prop.multiProcessorCount * 4);  // wave instructions per SIMD
    for (int m = 0; m < kModes; m++)
        printf("%-22s %.2f ms, %.2f cycles per wave instruction per SIMD at 2.4 GHz\n", kNames[m], ms[m], ms[m] * 1e-3 * 2.4e9 / per_simd);
    return 0;
}
