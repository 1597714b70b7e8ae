// record_size.hip - random record fetch rate vs record size (16..256 B per lane per step), per-lane dwordx4 loads,
// dependent chain like a traversal.  Working set given in bytes: 2 MB = L2-resident, 107 MB = Infinity-Cache-resident.
// Tells whether the cache hierarchy limits REQUESTS or BYTES: if records/s is flat in the record size, wider
// records (two tree levels per fetch) would pay; if bytes/s is flat, they would not.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NQ>
__global__ void __launch_bounds__(256, 8) fetch(const float4* __restrict__ recs, uint32_t n_recs, int iters, float* out)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = tid * 2654435761u + 12345u;
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
        s = s * 1664525u + 1013904223u;
        const uint32_t idx = ((s >> 4) + (__float_as_uint(acc) & 1u)) % n_recs;
        const float4* p = recs + (size_t)idx * NQ;
        float4 v[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) v[q] = p[q];
#pragma unroll
        for (int q = 0; q < NQ; q++) acc += v[q].x + v[q].w;
    }
    out[tid] = acc;
}

template <int NQ>
int run(const float4* recs, size_t bytes, int iters, int blocks, float* out, int w)
{
    const uint32_t n_recs = (uint32_t)(bytes / (NQ * 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(fetch<NQ>, dim3(blocks), dim3(256), 0, 0, recs, n_recs, iters, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double n = (double)blocks * 256 * iters;
    printf("set %6.1f MB  waves/SIMD %d  record %3d B: %7.1f G records/s  %6.2f TB/s\n", bytes / 1e6, w, NQ * 16, n / best / 1e6,
           n * NQ * 16 / best / 1e9);
    return 0;
}

int main(int argc, char** argv)
{
    const size_t bytes = (size_t)(argc > 1 ? atof(argv[1]) : 107.0) * 1000000;
    const int iters = argc > 2 ? atoi(argv[2]) : 2000;
    const int w = argc > 3 ? atoi(argv[3]) : 5;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * w;
    float4* recs; float* out;
    CHECK(hipMalloc(&recs, bytes + 4096));
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CHECK(hipMemset(recs, 0, bytes + 4096));
    if (run<1>(recs, bytes, iters, blocks, out, w)) return 1;
    if (run<2>(recs, bytes, iters, blocks, out, w)) return 1;
    if (run<4>(recs, bytes, iters, blocks, out, w)) return 1;
    if (run<8>(recs, bytes, iters, blocks, out, w)) return 1;
    if (run<16>(recs, bytes, iters, blocks, out, w)) return 1;
    return 0;
}
