/* arith_probe.cl - TEST INFRASTRUCTURE (our own code, no reference source): what the image's OpenCL compiler makes of the
 * floating-point operators the reference's kernel uses, when it is invoked the way the reference invokes it (no
 * floating-point option: OpenCL default arithmetic, PathTracer_OpenCL.cpp:292-314).  Compiled by oracle/Makefile (target
 * probe) with the flags of the reference kernels into build/arith_probe.hsaco and launched by device_math_probe.hip beside
 * the product's restatement of the same operators (csrc/ptmi_device.hpp: fdiv, frcp, fdiv by a constant, fsqrt, length),
 * so that a GPU test can require both to agree bit for bit on millions of operands, special values included.
 * out[k * n + i], k = 0..9 */
__kernel void arith_probe(__global const float* a, __global const float* b, __global float* out, uint n)
{
    const uint i = get_global_id(0);
    if (i >= n) return;
    const float x = a[i], y = b[i];
    out[0 * n + i] = x / y;
    out[1 * n + i] = 1.0f / x;
    out[2 * n + i] = x / 255.f;
    out[3 * n + i] = x / 3.f;
    out[4 * n + i] = x / 1.55f;
    out[5 * n + i] = x / ((float)1920);
    out[6 * n + i] = x / ((float)90);
    out[7 * n + i] = sqrt(x);
    out[8 * n + i] = length((float4)(x, y, x * 0.5f, 0.0f));
    out[9 * n + i] = x * y + 1.0f; /* one expression: contracted */
}
