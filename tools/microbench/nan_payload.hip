#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__global__ void k(const float* in, uint32_t* out)
{
    const float m = in[0], a = in[1], b = in[2];
    out[0] = __float_as_uint(__builtin_fmaf(a, b, m));
    out[1] = __float_as_uint(m + a);
    out[2] = __float_as_uint(a * b + m);
    out[3] = __float_as_uint(__builtin_fmaf(m, b, a));
    out[4] = __float_as_uint(__builtin_fmaf(in[3], b, m));  // inf * 0 + m
    out[5] = __float_as_uint(m * 1.0f);
    out[6] = __float_as_uint(__builtin_fmaf(in[4], b, m));  // nan(other payload) * b + m
}
int main()
{
    uint32_t mb = 0x7FC0DEADu, ob = 0x7FC12345u; float m, o; memcpy(&m, &mb, 4); memcpy(&o, &ob, 4);
    float h[5] = {m, 0.37f, 2.5f, INFINITY, o};
    float* d; uint32_t* r; hipMalloc(&d, 20); hipMalloc(&r, 28);
    hipMemcpy(d, h, 20, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d, r);
    uint32_t out[7]; hipMemcpy(out, r, 28, hipMemcpyDeviceToHost);
    for (int i = 0; i < 7; i++) printf("%d: %08x\n", i, out[i]);
    return 0;
}
