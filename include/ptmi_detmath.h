/*
 * ptmi_detmath.h - the one transcendental on the hot path, made deterministic.
 *
 * The reference calls the OpenCL builtins cos()/sin() exactly once per diffuse
 * bounce (Kernel/PathTracer_FullKernel.cl:412-413, theta in [0, 2*pi]).  OpenCL
 * only bounds their error (4 ulp), so the reference's own result depends on the
 * OpenCL runtime it runs on.  This integrator fixes ONE plain-fp32 algorithm
 * (Cody-Waite reduction by pi/4 + degree-7/8 minimax polynomials, no FMA, no
 * table) so that the HIP kernels and any CPU checker evaluate bit-identical
 * values when compiled with -ffp-contract=off.  Max error ~1 ulp on [0, 2*pi],
 * i.e. inside what OpenCL allows the reference.
 *
 * C99 / C++ / HIP.  Every operation is an IEEE-754 binary32 +,-,* written as an
 * explicit expression tree; do not "simplify" it.
 */
#ifndef PTMI_DETMATH_H
#define PTMI_DETMATH_H

#if defined(__HIPCC__)
#define PTMI_HD __host__ __device__ static inline
#else
#define PTMI_HD static inline
#endif

PTMI_HD void ptmi_sincosf(float x, float* sin_out, float* cos_out)
{
    const float ax = x < 0.0f ? -x : x;
    /* octant index, rounded up to even => r in [-pi/4, pi/4] */
    int j = (int)(ax * 1.27323954473516f); /* 4/pi */
    j = (j + 1) & ~1;
    const float y = (float)j;
    /* pi/4 split in three parts; y*DP1 is exact (DP1 has 8 significant bits) */
    float r = ax - y * 0.78515625f;
    r = r - y * 2.4187564849853515625e-4f;
    r = r - y * 3.77489497744594108e-8f;
    const float z = r * r;

    float ps = -1.9515295891e-4f;
    ps = ps * z + 8.3321608736e-3f;
    ps = ps * z - 1.6666654611e-1f;
    const float sp = r + r * (z * ps);

    float pc = 2.443315711809948e-5f;
    pc = pc * z - 1.388731625493765e-3f;
    pc = pc * z + 4.166664568298827e-2f;
    const float cp = (1.0f - 0.5f * z) + (z * z) * pc;

    float s, c;
    switch ((j >> 1) & 3) {
    case 0:  s = sp;  c = cp;  break;
    case 1:  s = cp;  c = -sp; break;
    case 2:  s = -sp; c = -cp; break;
    default: s = -cp; c = sp;  break;
    }
    *sin_out = x < 0.0f ? -s : s;
    *cos_out = c;
}

#endif /* PTMI_DETMATH_H */
