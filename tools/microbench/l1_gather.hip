// l1_gather.hip - the ceiling bench.py's `roofline` divides by: how many lane accesses per second the 256 L1s (TCP) of the
// chip serve for the integrator's kind of load - every lane of a wave reads ONE 64-byte record of its own with dwordx4 loads,
// the records of a wave scattered over the array (render_wavefront_kernel: node_step / leaf_pass).
//
// Round 3's constant (858 G accesses/s) came from a run whose every record was a NEW line: that run was limited by line
// FILLS, not by the access rate it was quoted for.  This bench separates the two and reproduces the kernel's own regime:
//   * `hot` records live in a set small enough to stay in every L1 (default 64 records = 4 KB): an access to them is an L1 hit;
//   * `cold` records are drawn from a big array (default 107 MB = the 1M-triangle scene's records: L2 / Infinity Cache) and
//     cost a line fill each;
//   * --miss-permille M: M of 1000 records are cold.  M = 0: the pure access-rate limit.  The kernel's mix (PMC:
//     TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES = 0.19 at 3.14 accesses per record) is M = 600;
//   * --late-permille Q: Q of 1000 records read all four quads, the others only the first two (a triangle rejected at the
//     distance tests reads half of its record: 3.14 accesses per record in the kernel = Q 570);
//   * U records in flight per lane (1 = the dependent chain of a traversal: the next index needs the data just loaded;
//     2, 4 = independent chains per lane: what the L1 delivers when latency is hidden);
//   * 5 waves per SIMD by default, like the kernel.
// Output: one JSON object per (U) with records/s, lane accesses/s (= records x quads read) and the parameters; run under
// `rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum` the counters show which of the two rates a run was
// limited by (tools/l1_ceiling.sh puts both into profiles/r04_l1_gather_microbench.json).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int U>
__global__ void __launch_bounds__(256, 5) gather(const float4* __restrict__ recs, uint32_t n_cold, uint32_t n_hot, uint32_t miss_permille,
                                                 uint32_t late_permille, int iters, float* out, unsigned long long* quads_read)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s[U];
    float acc[U];
#pragma unroll
    for (int u = 0; u < U; u++) { s[u] = (tid * U + u) * 2654435761u + 12345u; acc[u] = 0.f; }
    uint32_t quads = 0;
    for (int it = 0; it < iters; it++) {
        float4 a[U], b[U], c[U], d[U];
        bool late[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            s[u] = s[u] * 1664525u + 1013904223u;
            const uint32_t r = (s[u] >> 8) + (__float_as_uint(acc[u]) & 1u);  // (depends on the data just loaded)
            const bool cold = (r % 1000u) < miss_permille;
            late[u] = ((r / 1000u) % 1000u) < late_permille;
            // hot records: the first n_hot records of the array; cold ones anywhere behind them
            const uint32_t idx = cold ? n_hot + (r / 7u) % n_cold : (r / 7u) % n_hot;
            const float4* p = recs + (size_t)idx * 4;
            a[u] = p[0]; b[u] = p[1];
            c[u] = make_float4(0, 0, 0, 0); d[u] = c[u];
            if (late[u]) { c[u] = p[2]; d[u] = p[3]; }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            acc[u] += a[u].x + b[u].y + c[u].z + d[u].w;
            quads += late[u] ? 4u : 2u;
        }
    }
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < U; u++) t += acc[u];
    out[tid] = t;
    // quads read by the whole grid (one atomic per wave)
    unsigned long long q = quads;
    for (int o = 32; o > 0; o >>= 1) q += __shfl_down(q, o);
    if ((threadIdx.x & 63u) == 0) atomicAdd(quads_read, q);
}

// --cooperative 1: the same 64 records per wave and round, but FOUR LANES SHARE A RECORD - lane l reads quad (l & 3) of the record of
// its group of four, in four instructions for four records: every dwordx4 instruction of the wave touches 16 lines instead of 64.
// (Does the L1 charge per lane or per line?  If per line, a traversal that fetched its records this way and handed the quads round
// through the LDS would quadruple the ceiling.)
__global__ void __launch_bounds__(256, 5) gather_cooperative(const float4* __restrict__ recs, uint32_t n_cold, uint32_t n_hot, uint32_t miss_permille,
                                                             int iters, float* out, unsigned long long* quads_read)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t quad = tid & 3u;
    uint32_t s = tid * 2654435761u + 12345u;
    float acc = 0.f;
    uint32_t quads = 0;
    for (int it = 0; it < iters; it++) {
        // every lane draws ONE record, as in the other form (same index arithmetic per record) ...
        s = s * 1664525u + 1013904223u;
        const uint32_t r = (s >> 8) + (__float_as_uint(acc) & 1u);
        const bool cold = (r % 1000u) < miss_permille;
        const uint32_t idx = cold ? n_hot + (r / 7u) % n_cold : (r / 7u) % n_hot;
        // ... and the four lanes of a group fetch their four records together: instruction u reads the record of the group's lane u
        float4 a[4];
        a[0] = recs[(size_t)__builtin_amdgcn_update_dpp(0u, idx, 0x00, 0xf, 0xf, false) * 4 + quad];
        a[1] = recs[(size_t)__builtin_amdgcn_update_dpp(0u, idx, 0x55, 0xf, 0xf, false) * 4 + quad];
        a[2] = recs[(size_t)__builtin_amdgcn_update_dpp(0u, idx, 0xaa, 0xf, 0xf, false) * 4 + quad];
        a[3] = recs[(size_t)__builtin_amdgcn_update_dpp(0u, idx, 0xff, 0xf, 0xf, false) * 4 + quad];
#pragma unroll
        for (int u = 0; u < 4; u++) acc += a[u].x + a[u].w;
        quads += 4u;
    }
    out[tid] = acc;
    unsigned long long q = quads;
    for (int o = 32; o > 0; o >>= 1) q += __shfl_down(q, o);
    if ((threadIdx.x & 63u) == 0) atomicAdd(quads_read, q);
}

int main(int argc, char** argv)
{
    uint32_t n_cold = 1665533u, n_hot = 64u, miss = 600u, late = 570u;
    int iters = 2000, waves = 5, only_u = 0, cooperative = 0;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--cold-records")) n_cold = (uint32_t)atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--hot-records")) n_hot = (uint32_t)atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--miss-permille")) miss = (uint32_t)atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--late-permille")) late = (uint32_t)atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--iters")) iters = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--waves-per-simd")) waves = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--in-flight")) only_u = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--cooperative")) cooperative = atoi(argv[i + 1]);
        else { printf("unknown option %s\n", argv[i]); return 2; }
    }
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * waves;  // 256 threads = 4 waves = one per SIMD: `waves` blocks per CU
    float4* recs; float* out; unsigned long long* quads;
    const size_t n_all = (size_t)n_cold + n_hot;
    CHECK(hipMalloc(&recs, n_all * 64));
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CHECK(hipMalloc(&quads, 8));
    std::vector<float> h(n_all * 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)(i % 7);
    CHECK(hipMemcpy(recs, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    if (cooperative) {
        float best = 1e30f;
        unsigned long long q = 0;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipMemset(quads, 0, 8));
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(gather_cooperative, dim3(blocks), dim3(256), 0, 0, recs, n_cold, n_hot, miss, iters, out, quads);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            CHECK(hipMemcpy(&q, quads, 8, hipMemcpyDeviceToHost));
        }
        const double n_rec = (double)blocks * 256 * iters;  // 64 records per wave and round, as in the other form
        printf("{\"cooperative\": 1, \"in_flight_per_lane\": 1, \"waves_per_simd\": %d, \"hot_records\": %u, \"cold_records\": %u, \"miss_permille\": %u, "
               "\"late_permille\": 1000, \"ms\": %.3f, \"G_records_per_s\": %.2f, \"quads_per_record\": %.3f, \"G_lane_accesses_per_s\": %.2f, "
               "\"lane_accesses_per_clk_per_CU_at_2.4GHz\": %.3f, \"CUs\": %d}\n",
               waves, n_hot, n_cold, miss, best, n_rec / best / 1e6, (double)q / n_rec, (double)q / best / 1e6,
               (double)q / (best * 1e-3) / 2.4e9 / prop.multiProcessorCount, prop.multiProcessorCount);
        return 0;
    }
    for (int u : {1, 2, 4}) {
        if (only_u && u != only_u) continue;
        float best = 1e30f;
        unsigned long long q = 0;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipMemset(quads, 0, 8));
            CHECK(hipEventRecord(e0));
            if (u == 1) hipLaunchKernelGGL(gather<1>, dim3(blocks), dim3(256), 0, 0, recs, n_cold, n_hot, miss, late, iters, out, quads);
            if (u == 2) hipLaunchKernelGGL(gather<2>, dim3(blocks), dim3(256), 0, 0, recs, n_cold, n_hot, miss, late, iters / 2, out, quads);
            if (u == 4) hipLaunchKernelGGL(gather<4>, dim3(blocks), dim3(256), 0, 0, recs, n_cold, n_hot, miss, late, iters / 4, out, quads);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            CHECK(hipMemcpy(&q, quads, 8, hipMemcpyDeviceToHost));
        }
        const double n_rec = (double)blocks * 256 * (iters / u) * u;
        printf("{\"in_flight_per_lane\": %d, \"waves_per_simd\": %d, \"hot_records\": %u, \"cold_records\": %u, \"miss_permille\": %u, "
               "\"late_permille\": %u, \"ms\": %.3f, \"G_records_per_s\": %.2f, \"quads_per_record\": %.3f, \"G_lane_accesses_per_s\": %.2f, "
               "\"lane_accesses_per_clk_per_CU_at_2.4GHz\": %.3f, \"CUs\": %d}\n",
               u, waves, n_hot, n_cold, miss, late, best, n_rec / best / 1e6, (double)q / n_rec, (double)q / best / 1e6,
               (double)q / (best * 1e-3) / 2.4e9 / prop.multiProcessorCount, prop.multiProcessorCount);
    }
    return 0;
}
