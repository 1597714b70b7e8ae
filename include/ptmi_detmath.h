/*
 * ptmi_detmath.h - the one transcendental on the hot path, fixed to the platform's definition.
 *
 * The reference calls the OpenCL builtins cos()/sin() exactly once per diffuse
 * bounce (Kernel/PathTracer_FullKernel.cl:412-413, theta in [0, 2*pi]).  OpenCL
 * only bounds their error (4 ulp), so the reference's own result depends on the
 * OpenCL library it meets.  On this platform that is the ROCm device library:
 * sin() / cos() = __ocml_sin_f32 / __ocml_cos_f32 (ocml.bc), which for |x| < 2^17
 * on gfx9+ are the fixed sequence of fp32 operations restated below
 * (__ocmlpriv_trigredsmall_f32: reduction by pi/2 in three fma steps;
 * __ocmlpriv_sincosred_f32: two odd/even polynomials evaluated with fma).  Every
 * operation is one IEEE-754 binary32 multiply or fused multiply-add, so the HIP
 * kernels, the CPU checker and the reference kernel compiled for gfx950 evaluate
 * bit-identical values (tests/test_parity_gpu.py: the device library's sinf/cosf
 * against this restatement; the integrator against the reference's strict build).
 *
 * C99 / C++ / HIP.  Do not "simplify" the expression trees; compile with
 * -ffp-contract=off so that only the fma() calls written here are fused.
 */
#ifndef PTMI_DETMATH_H
#define PTMI_DETMATH_H

#if defined(__HIPCC__)
#define PTMI_HD __host__ __device__ static inline
#else
#include <math.h>
#define PTMI_HD static inline
#endif

/* sin and cos of x for |x| < 131072 (the reference's argument is theta in [0, 2*pi]). */
PTMI_HD void ptmi_sincosf(float x, float* sin_out, float* cos_out)
{
    const float ax = x < 0.0f ? -x : x;
    /* __ocmlpriv_trigredsmall_f32: k = rint(ax * 2/pi), r = ax - k * pi/2 with pi/2 split in three */
    const float k = __builtin_rintf(ax * 6.3661975e-01f /* 0x1.45f306p-1 */);
    float r = __builtin_fmaf(k, -1.5707963e+00f /* 0x1.921fb4p+0 */, ax);
    r = __builtin_fmaf(k, -7.5497894e-08f /* 0x1.4442dp-24 */, r);
    r = __builtin_fmaf(k, -5.3903025e-15f /* 0x1.846988p-48 */, r);
    const int q = (int)k & 3;
    /* __ocmlpriv_sincosred_f32 */
    const float z = r * r;
    float ps = __builtin_fmaf(z, -1.9464458e-04f /* 0x1.983304p-13 */, 8.33172e-03f /* 0x1.110388p-7 */);
    ps = __builtin_fmaf(z, ps, -1.6666646e-01f /* 0x1.55553ap-3 */);
    const float s = __builtin_fmaf(r, z * ps, r);
    float pc = __builtin_fmaf(z, 2.5668742e-05f /* 0x1.aea668p-16 */, -1.390911e-03f /* 0x1.6c9e76p-10 */);
    pc = __builtin_fmaf(z, pc, 4.1667905e-02f /* 0x1.5557eep-5 */);
    pc = __builtin_fmaf(z, pc, -5.0000024e-01f /* 0x1.000008p-1 */);
    const float c = __builtin_fmaf(z, pc, 1.0f);
    /* __ocml_sin_f32 / __ocml_cos_f32: quadrant selection and signs */
    float sv = (q & 1) ? c : s;
    float cv = (q & 1) ? -s : c;
    if (q > 1) { sv = -sv; cv = -cv; }
    *sin_out = x < 0.0f ? -sv : sv;
    *cos_out = cv;
}

#endif /* PTMI_DETMATH_H */
