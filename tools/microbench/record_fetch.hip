// record_fetch.hip - how fast can a CU fetch random 64-byte records, and does it matter how the lanes ask?
//
// The wavefront integrator's traversal trip loads one 64-byte record per lane with four global_load_dwordx4
// (every lane a different cache line).  rocprofv3 shows the L1 (TCP) doing ~0.96 tag accesses per clock in that
// kernel - one per lane per instruction.  This microbenchmark times the alternatives on the same access pattern
// (random records in a 107 MB array, 5 waves per SIMD, persistent lanes):
//   mode 0: per-lane        - lane L reads its own record with 4 x dwordx4                (4 accesses / record)
//   mode 1: quad-cooperative - the 4 lanes of a quad read ONE record per instruction, 16 B each (contiguous 64 B)
//   mode 2: pair-cooperative - 2 lanes read 32 contiguous bytes of one record per instruction
//   mode 3: quad-cooperative with non-temporal loads (no L1 allocation), mode 4: per-lane with non-temporal loads
// Every mode reads the same 64 bytes per lane per iteration; a checksum keeps the loads alive.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256, 8) fetch(const float4* __restrict__ recs, uint32_t n_recs, int iters, float* out)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = tid * 2654435761u + 12345u;
    float acc = 0.f;
    const uint32_t lane = threadIdx.x & 63u;
    for (int it = 0; it < iters; it++) {
        s = s * 1664525u + 1013904223u;
        // next record depends on the data just loaded (like a traversal): fold acc's low bit in
        const uint32_t idx = ((s >> 4) + (__float_as_uint(acc) & 1u)) % n_recs;
        if (MODE == 0) {
            const float4* p = recs + (size_t)idx * 4;
            const float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
        } else if (MODE == 1) {
            const uint32_t q = lane & 3u;
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t idx_k = __shfl(idx, (int)((lane & ~3u) + k));
                const float4 v = recs[(size_t)idx_k * 4 + q];
                // what lane k of the quad would have summed: a.x + b.y + c.z + d.w -> element q of quad q
                const float e = q == 0 ? v.x : (q == 1 ? v.y : (q == 2 ? v.z : v.w));
                // give it to lane k (sum over the quad's four pieces)
                float sum = e;
                sum += __shfl_xor(sum, 1);
                sum += __shfl_xor(sum, 2);
                if ((int)q == k) t = sum;
            }
            acc += t;
        } else if (MODE == 3 || MODE == 4) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f* base = reinterpret_cast<const v4f*>(recs);
            if (MODE == 4) {
                const v4f* p = base + (size_t)idx * 4;
                const v4f a = __builtin_nontemporal_load(p), b = __builtin_nontemporal_load(p + 1),
                          c = __builtin_nontemporal_load(p + 2), d = __builtin_nontemporal_load(p + 3);
                acc += a.x + b.y + c.z + d.w;
            } else {
                const uint32_t q = lane & 3u;
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t idx_k = __shfl(idx, (int)((lane & ~3u) + k));
                    const v4f v = __builtin_nontemporal_load(base + (size_t)idx_k * 4 + q);
                    const float e = q == 0 ? v.x : (q == 1 ? v.y : (q == 2 ? v.z : v.w));
                    float sum = e;
                    sum += __shfl_xor(sum, 1);
                    sum += __shfl_xor(sum, 2);
                    if ((int)q == k) t = sum;
                }
                acc += t;
            }
        } else {
            const uint32_t h = lane & 1u;
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const uint32_t idx_k = __shfl(idx, (int)((lane & ~1u) + k));
                const float4 v0 = recs[(size_t)idx_k * 4 + 2 * h], v1 = recs[(size_t)idx_k * 4 + 2 * h + 1];
                float sum = h == 0 ? v0.x + v1.y : v0.z + v1.w;
                sum += __shfl_xor(sum, 1);
                if ((int)h == k) t = sum;
            }
            acc += t;
        }
    }
    out[tid] = acc;
}

int main(int argc, char** argv)
{
    const uint32_t n_recs = argc > 1 ? (uint32_t)atoi(argv[1]) : 1665533u;  // 1M triangles + 665533 nodes
    const int iters = argc > 2 ? atoi(argv[2]) : 4000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int waves_per_simd = argc > 3 ? atoi(argv[3]) : 5;
    const int blocks = prop.multiProcessorCount * waves_per_simd;
    float4* recs; float* out;
    CHECK(hipMalloc(&recs, (size_t)n_recs * 64));
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    std::vector<float> h((size_t)n_recs * 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)(i % 7);
    CHECK(hipMemcpy(recs, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode < 5; mode++) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(fetch<0>, dim3(blocks), dim3(256), 0, 0, recs, n_recs, iters, out);
            if (mode == 1) hipLaunchKernelGGL(fetch<1>, dim3(blocks), dim3(256), 0, 0, recs, n_recs, iters, out);
            if (mode == 2) hipLaunchKernelGGL(fetch<2>, dim3(blocks), dim3(256), 0, 0, recs, n_recs, iters, out);
            if (mode == 3) hipLaunchKernelGGL(fetch<3>, dim3(blocks), dim3(256), 0, 0, recs, n_recs, iters, out);
            if (mode == 4) hipLaunchKernelGGL(fetch<4>, dim3(blocks), dim3(256), 0, 0, recs, n_recs, iters, out);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        std::vector<float> o((size_t)blocks * 256);
        CHECK(hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost));
        double cs = 0; for (float v : o) cs += v;
        const double recs_total = (double)blocks * 256 * iters;
        printf("waves/SIMD %d mode %d: %.2f ms  %.1f G records/s  %.2f TB/s  (%.2f records/clk/CU at 2.4 GHz)  checksum %.6g\n", waves_per_simd, mode, best,
               recs_total / best / 1e6, recs_total * 64 / best / 1e9, recs_total / (best * 1e-3) / 2.4e9 / prop.multiProcessorCount, cs);
    }
    return 0;
}
