/*
 * pt_oracle.c - CPU oracle (TEST INFRASTRUCTURE, see pt_oracle.h).
 *
 * Scalar restatement of Kernel/PathTracer_FullKernel.cl ("cl:") and
 * Kernel/PathTracer_FullKernel_header.cl ("h:") of the reference.  OpenCL C
 * semantics are spelled out where C differs: float4 arithmetic is on all four
 * components, dot/length/normalize are 4-component, unsuffixed literals are
 * double, size_t/uint/int promotions follow C.
 *
 * Build with -O2 -ffp-contract=off and WITHOUT -march=native / -ffast-math:
 * every + - * / sqrt below must be one correctly rounded binary32 operation.
 *
 * TWO BUILDS of this file (oracle/Makefile), one per arithmetic the reference's source can be compiled to on this platform:
 *   libpt_oracle.so      PTO_DEFAULT_ARITHMETIC 0: the STRICT build of the reference kernel (-ffp-contract=off
 *                        -cl-fp32-correctly-rounded-divide-sqrt): every operation of the source correctly rounded;
 *   libpt_oracle_da.so   PTO_DEFAULT_ARITHMETIC 1: the build the reference's OWN build line produces (OpenCL_BuildOptions,
 *                        OpenCL.cpp:292-314: no floating-point option = OpenCL default arithmetic), as clang's OpenCL front
 *                        end and the gfx950 back end compile it (established from the LLVM IR and the ISA of
 *                        oracle/_ref/ref_kernel_*.hsaco):
 *                          - `a * b + c` inside one expression (also `x += a * b`) is ONE fused multiply-add: mad() below,
 *                            at exactly the sites where the front end emits llvm.fmuladd (left operand tried first);
 *                          - a / b = ldexp(frexp_mant(a) * v_rcp_f32(frexp_mant(b)), frexp_exp(a) - frexp_exp(b)); a
 *                            constant divisor's reciprocal mantissa is folded at compile time (correctly rounded): fdivc();
 *                          - sqrt(x) = v_sqrt_f32 (behind a 2^32 scaling of denormal inputs).
 *                        v_rcp_f32 and v_sqrt_f32 are not correctly rounded; like v_rsq_f32 they are reproduced from
 *                        tables measured on an MI355X (tests/golden/rcp_gfx950.npz, sqrt_gfx950.npz).
 */
#ifndef PTO_DEFAULT_ARITHMETIC
#define PTO_DEFAULT_ARITHMETIC 0
#endif
#include "pt_oracle.h"
#include "ptmi_detmath.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* float4 helpers: OpenCL vector semantics                                    */
/* ------------------------------------------------------------------------- */

typedef struct { float x, y, z, w; } f4;

static inline f4 mk4(float x, float y, float z, float w) { f4 r = { x, y, z, w }; return r; }
static inline f4 ld4(const ptmi_float4* p) { return mk4(p->x, p->y, p->z, p->w); }
static inline f4 add4(f4 a, f4 b) { return mk4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline f4 sub4(f4 a, f4 b) { return mk4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
static inline f4 mul4(f4 a, f4 b) { return mk4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
static inline f4 scale4(f4 a, float s) { return mk4(a.x * s, a.y * s, a.z * s, a.w * s); }
static inline float fdiv(float a, float b);
static inline f4 div4s(f4 a, float s) { return mk4(fdiv(a.x, s), fdiv(a.y, s), fdiv(a.z, s), fdiv(a.w, s)); }
static inline f4 neg4(f4 a) { return mk4(-a.x, -a.y, -a.z, -a.w); }
static inline float mad(float a, float b, float c);
static inline f4 mad4(f4 a, f4 b, f4 c) { return mk4(mad(a.x, b.x, c.x), mad(a.y, b.y, c.y), mad(a.z, b.z, c.z), mad(a.w, b.w, c.w)); }
static inline f4 mad4s(f4 a, float s, f4 c) { return mk4(mad(a.x, s, c.x), mad(a.y, s, c.y), mad(a.z, s, c.z), mad(a.w, s, c.w)); }

/* The OpenCL geometric builtins are implementation-defined in their last bits.  They are fixed here to
 * the definitions of the OpenCL library the reference meets on this hardware (ROCm device libs, opencl.bc:
 * _Z3dotDv4_fS_, _Z5crossDv4_fS_, _Z9normalizeDv4_f), evaluated exactly:
 *   dot(a,b)     = fma(a.w,b.w, fma(a.z,b.z, fma(a.y,b.y, a.x*b.x)))           -- exact restatement
 *   cross(a,b).x = fma(a.y,b.z, b.y*(-a.z)) (y, z cyclic), w = 0                -- exact restatement
 *   length(a)    = sqrt(dot(a,a))                (the library's sqrt is correctly rounded in the strict build)
 *   normalize(a) = a * rsqrt(dot(a,a)), rsqrt = __ocml_rsqrt_f32 = the hardware instruction v_rsq_f32 (plus range
 *                  scaling): not a correctly rounded function - on gfx950 it is the correctly rounded 1/sqrt for 89 % of
 *                  the inputs, one ulp below for 9.8 % and one above for 1.3 % (tools/microbench/rsq_survey.hip).  The
 *                  deviation depends only on the mantissa and the parity of the exponent, so it is reproduced here from
 *                  the table that survey wrote on an MI355X: 2^24 entries of 2 bits (tests/golden/rsq_gfx950.npz,
 *                  handed over with pto_set_rsq_table).
 * fmaf() is the correctly rounded fused multiply-add (one rounding), whatever -ffp-contract says. */
static inline float dot4(f4 a, f4 b) { return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x))); }

/* ---- the arithmetic of the two builds (see the header) ---- */
static const uint8_t *g_rcp_table, *g_sqrt_table; /* 2 bits per entry, as g_rsq_table below */
void pto_set_rcp_table(const uint8_t* packed) { g_rcp_table = packed; }
void pto_set_sqrt_table(const uint8_t* packed) { g_sqrt_table = packed; }
int pto_default_arithmetic(void) { return PTO_DEFAULT_ARITHMETIC; }

static inline int table_deviation(const uint8_t* table, uint32_t idx) { return (int)((table[idx >> 2] >> ((idx & 3u) * 2u)) & 3u) - 1; }
static inline float nudge(float r, int ulps)
{
    uint32_t rb;
    memcpy(&rb, &r, 4);
    rb = (uint32_t)((int32_t)rb + ulps);
    memcpy(&r, &rb, 4);
    return r;
}

/* v_frexp_mant_f32 / v_frexp_exp_i32_f32: mantissa in +-[0.5, 1) and its exponent; zero, infinities and NaN pass through
 * with exponent 0 */
static inline float hw_frexp_mant(float x, int* e)
{
    *e = 0;
    if (x == 0.0f || isinf(x) || x != x) return x;
    return frexpf(x, e);
}
/* v_rcp_f32 on gfx950 for what v_frexp_mant_f32 can hand it */
float pto_hardware_rcp(float m)
{
    uint32_t bits;
    float r;
    if (m != m) return m;
    if (m == 0.0f) return copysignf(INFINITY, m);
    if (isinf(m)) return copysignf(0.0f, m);
    memcpy(&bits, &m, 4);
    if (!g_rcp_table || (bits & 0x7F800000u) != 0x3F000000u) abort(); /* no table handed over / not a mantissa in [0.5, 1) */
    r = (float)(1.0 / (double)fabsf(m));
    r = nudge(r, table_deviation(g_rcp_table, bits & 0x7FFFFFu));
    return copysignf(r, m);
}
/* v_sqrt_f32 on gfx950 for a normal, zero, infinite or negative input */
float pto_hardware_sqrt(float x)
{
    uint32_t bits, rb;
    int e, parity;
    float red, r;
    if (x != x) return x;
    if (x == 0.0f) return x;
    if (x < 0.0f) return NAN;
    if (isinf(x)) return x;
    memcpy(&bits, &x, 4);
    e = (int)(bits >> 23) - 127;
    if (!g_sqrt_table || e == -127) abort(); /* no table handed over / denormal input (fsqrt scales first) */
    parity = e & 1; /* x = red * 2^(e - parity), red in [1,2) or [2,4), e - parity even */
    rb = (parity ? 0x40000000u : 0x3F800000u) | (bits & 0x7FFFFFu);
    memcpy(&red, &rb, 4);
    r = (float)sqrt((double)red);
    r = nudge(r, table_deviation(g_sqrt_table, ((uint32_t)parity << 23) | (bits & 0x7FFFFFu)));
    return ldexpf(r, (e - parity) / 2); /* exact */
}

/* a * b + c written in one expression of the reference */
static inline float mad(float a, float b, float c)
{
#if PTO_DEFAULT_ARITHMETIC
    return fmaf(a, b, c);
#else
    return a * b + c;
#endif
}
/* a / b, b a run-time value */
static inline float fdiv(float a, float b)
{
#if PTO_DEFAULT_ARITHMETIC
    int ea, eb;
    const float ma = hw_frexp_mant(a, &ea), mb = hw_frexp_mant(b, &eb);
    return ldexpf(ma * pto_hardware_rcp(mb), ea - eb);
#else
    return a / b;
#endif
}
/* a / c, c a constant of the reference's program (a literal, IMAGE_WIDTH, IMAGE_HEIGHT): positive and normal */
static inline float fdivc(float a, float c)
{
#if PTO_DEFAULT_ARITHMETIC
    int ea, ec;
    const float ma = hw_frexp_mant(a, &ea), mc = frexpf(c, &ec);
    return ldexpf(ma * (1.0f / mc), ea - ec);
#else
    return a / c;
#endif
}
/* sqrt(x) where the compiler is not asked for the correctly rounded one: v_sqrt_f32 behind a 2^32 scaling of denormal
 * inputs - the default build's sqrt, and the square root inside the library's length() in BOTH builds */
static inline float approx_sqrt(float x)
{
    if (x < 0x1p-126f) return ldexpf(pto_hardware_sqrt(ldexpf(x, 32)), -16);
    return pto_hardware_sqrt(x);
}
static inline float fsqrt(float x)
{
#if PTO_DEFAULT_ARITHMETIC
    return approx_sqrt(x);
#else
    return sqrtf(x);
#endif
}

static const uint8_t* g_rsq_table; /* 2 bits per entry: 0 = one ulp below, 1 = equal, 2 = one ulp above the correctly rounded value */
void pto_set_rsq_table(const uint8_t* packed) { g_rsq_table = packed; }

/* v_rsq_f32 on gfx950 for a positive normal x (what normalize() feeds it after its own range scaling) */
float pto_hardware_rsq(float x)
{
    uint32_t bits, m, idx, rb;
    int e, parity, dev;
    float red, r;
    memcpy(&bits, &x, 4);
    if (x != x || x < 0.0f) return NAN;
    if (bits == 0u) return INFINITY;
    if (bits == 0x7F800000u) return 0.0f;
    e = (int)(bits >> 23) - 127;
    m = bits & 0x7FFFFFu;
    if (!g_rsq_table || e == -127) {
        /* no table handed over / denormal input: only reachable through the unit-test export (normalize scales first) */
        abort();
    }
    parity = e & 1; /* x = red * 2^(e - parity), red in [1,2) or [2,4), e - parity even */
    rb = (parity ? 0x40000000u : 0x3F800000u) | m;
    memcpy(&red, &rb, 4);
    r = (float)(1.0 / sqrt((double)red));
    idx = ((uint32_t)parity << 23) | m;
    dev = (int)((g_rsq_table[idx >> 2] >> ((idx & 3u) * 2u)) & 3u) - 1;
    memcpy(&rb, &r, 4);
    rb = (uint32_t)((int32_t)rb + dev);
    memcpy(&r, &rb, 4);
    return ldexpf(r, -(e - parity) / 2); /* exact: a power of two, results stay normal */
}

/* __ocml_rsqrt_f32 with denormals enabled: inputs below 2^-126 are scaled into the normal range first */
static inline float cl_rsqrt(float x)
{
    const int tiny = x < 0x1p-126f;
    const float r = pto_hardware_rsq(tiny ? x * 0x1p+24f : x);
    return tiny ? r * 4096.0f : r;
}

/* _Z9normalizeDv4_f of opencl.bc */
static inline f4 normalize4(f4 a)
{
    float d;
    if (a.x == 0 && a.y == 0 && a.z == 0 && a.w == 0) return a;
    d = dot4(a, a);
    if (d < 0x1p-126f) {
        a = scale4(a, 0x1p+86f);
        d = dot4(a, a);
    } else if (d == INFINITY) {
        a = scale4(a, 0x1p-66f);
        d = dot4(a, a);
        if (d == INFINITY) {
            a = mk4(copysignf(isinf(a.x) ? 1.0f : 0.0f, a.x), copysignf(isinf(a.y) ? 1.0f : 0.0f, a.y),
                    copysignf(isinf(a.z) ? 1.0f : 0.0f, a.z), copysignf(isinf(a.w) ? 1.0f : 0.0f, a.w));
            d = dot4(a, a);
        }
    }
    return scale4(a, cl_rsqrt(d));
}
/* _Z6lengthDv4_f of opencl.bc: sqrt(dot) with a rescaling for squared lengths outside the normal range.  The library's code,
 * hence the same in both builds of the reference: -cl-fp32-correctly-rounded-divide-sqrt reaches the kernel's own sqrt() calls,
 * not the library's - its square root stays the bare v_sqrt_f32 (ISA of both builds).  Its one use is the limit of a shadow
 * ray (cl:938), whose last bit only matters when the light sits on a box face or a vertex (the fuzzed scenes of
 * tests/test_reference_default_gpu.py found it). */
static inline float length4(f4 a)
{
    const float d = dot4(a, a);
    if (d < 0x1p-126f) {
        a = scale4(a, 0x1p+86f);
        return approx_sqrt(dot4(a, a)) * 0x1p-86f;
    }
    if (d == INFINITY) {
        a = scale4(a, 0x1p-66f);
        return approx_sqrt(dot4(a, a)) * 0x1p+66f;
    }
    return pto_hardware_sqrt(d);
}
static inline f4 cross4(f4 a, f4 b)
{
    return mk4(fmaf(a.y, b.z, b.y * -a.z), fmaf(a.z, b.x, b.z * -a.x), fmaf(a.x, b.y, b.x * -a.y), 0.0f);
}
/* OpenCL max(float,float): "y if x < y, otherwise x" */
static inline float cl_max(float x, float y) { return x < y ? y : x; }

/* h:10-16 */
#define PATH_PI 3.14159265f
#define PATH_PI_INVERSE 0.31830988618f
#define MIN_REFLECTION_NUMBER 5
#define MIN_CONTRIBUTION_VALUE 0.001f
/* h:166-169 */
#define MATERIAL_N_WATER 1.333f
#define MATERIAL_N_GLASS 1.55f
#define MATERIAL_N_VARNISH 3.f
#define MATERIAL_KSCHLICK 0.8f

/* device Ray3D, h:89-107 */
typedef struct {
    f4 origin, direction, inverse;
    float sample_x, sample_y;
    uint32_t num_bbx, num_tri, reflection_id;
    int is_in_water;
    f4 point;  /* intersectionPoint */
    f4 color;  /* intersectionColor */
    uint32_t triangle_id, material_id;
    float s, t;
} ray_t;

/* ------------------------------------------------------------------------- */
/* RNG and samplers                                                           */
/* ------------------------------------------------------------------------- */

/* random(), h:246-253: seed = (16807 * (ulong)seed) & 0x7FFFFFFF (the int is
 * sign-extended to ulong), result = (float)seed / (float)0x7FFFFFFF (=2^31). */
float pto_random(int32_t* seed)
{
    const uint64_t a = 16807u;
    const uint64_t m = 0x7FFFFFFFu;
    *seed = (int32_t)((a * (uint64_t)(int64_t)*seed) & m);
    return (float)*seed / (float)m;
}

/* InitializeRandomSeed(), h:255-264: everything ends up modulo 2^32.
 * The reference squares a signed int (`seed *= 2011; seed *= seed; if(seed == 0) seed = 1;`): the overflow is undefined in C,
 * and LLVM-based OpenCL compilers (verified on the ROCm one by disassembling oracle/_ref/ref_kernel_*.hsaco:
 * v_mul_lo_u32, v_mul_lo_u32, then v_cmp_ne_u32 0, <the UN-squared index>) fold the zero test onto the pixel/iteration
 * index: the seed becomes 1 only for index 0.  When the square wraps to 0 for another index (index a multiple of 2^16) the
 * seed stays 0 and random() returns 0 for the whole path.  Pinned by the reference fixtures: such a path recurs every
 * 2^16 / gcd(2^16, W*H) iterations at the same pixels and never averages out. */
int32_t pto_initialize_random_seed(uint32_t gx, uint32_t gy, uint32_t w, uint32_t h, uint32_t iteration)
{
    const uint32_t index = gx + gy * w + iteration * w * h;
    uint32_t s = index * 2011u;
    s *= s;
    if (index == 0u) s = 1u;
    return (int32_t)s;
}

/* sampler(), cl:1119-1150 */
void pto_sampler(uint32_t kind, uint32_t gx, uint32_t gy, uint32_t w, uint32_t h, uint32_t iteration,
                 int32_t* seed, float sample[2])
{
    if (kind == PTMI_SAMPLER_UNIFORM) { /* cl:1123-1135 */
        const int sample_id = (int)(iteration % 9u);
        float sx = (float)gx, sy = (float)gy;
        float ox = (float)(sample_id % 3), oy = (float)(sample_id / 3);
        ox += 0.5f; oy += 0.5f;
        ox = fdivc(ox, 3.f); oy = fdivc(oy, 3.f);
        sx += ox;   sy += oy;
        sx = fdivc(sx, (float)w); sy = fdivc(sy, (float)h);
        sx -= 0.5f; sy -= 0.5f;
        sample[0] = sx; sample[1] = sy;
    } else if (kind == PTMI_SAMPLER_RANDOM) { /* cl:1137-1141 */
        float sx = pto_random(seed);
        float sy = pto_random(seed);
        sx *= 0.9f;  sy *= 0.9f;
        sx += 0.05f; sy += 0.05f;
        sx -= 0.5f;  sy -= 0.5f;
        sample[0] = sx; sample[1] = sy;
    } else { /* JITTERED, cl:1145-1146 */
        sample[0] = fdivc(mad(0.9f, pto_random(seed), (float)gx) + 0.05f, (float)w) - 0.5f;
        sample[1] = fdivc(mad(0.9f, pto_random(seed), (float)gy) + 0.05f, (float)h) - 0.5f;
    }
}

/* ------------------------------------------------------------------------- */
/* Ray helpers, h:276-295                                                     */
/* ------------------------------------------------------------------------- */

static void ray_set_direction(ray_t* r, f4 d)
{
    r->direction = normalize4(d);
    r->inverse.x = fdiv(1.0f, r->direction.x);
    r->inverse.y = fdiv(1.0f, r->direction.y);
    r->inverse.z = fdiv(1.0f, r->direction.z);
    r->inverse.w = 0.0f;
}

static void ray_create(ray_t* r, f4 o, f4 d, int in_water)
{
    r->origin = o;
    r->is_in_water = in_water;
    r->num_bbx = 0;
    r->num_tri = 0;
    r->reflection_id = 0;
    r->sample_x = 0.0f;
    r->sample_y = 0.0f;
    ray_set_direction(r, d);
}

/* Vector_PutInSameHemisphereAs, h:237-244 */
static f4 put_in_same_hemisphere(f4 v, f4 n)
{
    const float d = dot4(v, n);
    if (d < 0.001f) v = mad4s(n, 0.01f - d, v); /* h:241 */
    return v;
}

/* ------------------------------------------------------------------------- */
/* BoundingBox_Intersects, cl:64-139                                          */
/* ------------------------------------------------------------------------- */

static int bbox_intersects(const ptmi_bounding_box* bb, ray_t* r, float squared_distance)
{
    float t_min, t_max, ty_min, ty_max, tz_min, tz_max;

    r->num_bbx++; /* cl:66, counted before the isEmpty test */
    if (bb->is_empty) return 0;

    if (r->direction.x > 0) {
        t_min = (bb->p_min.x - r->origin.x) * r->inverse.x;
        t_max = (bb->p_max.x - r->origin.x) * r->inverse.x;
    } else {
        t_min = (bb->p_max.x - r->origin.x) * r->inverse.x;
        t_max = (bb->p_min.x - r->origin.x) * r->inverse.x;
    }
    if (t_min < 0 && t_max < 0) return 0;

    if (r->direction.y > 0) {
        ty_min = (bb->p_min.y - r->origin.y) * r->inverse.y;
        ty_max = (bb->p_max.y - r->origin.y) * r->inverse.y;
    } else {
        ty_min = (bb->p_max.y - r->origin.y) * r->inverse.y;
        ty_max = (bb->p_min.y - r->origin.y) * r->inverse.y;
    }
    if (ty_min < 0 && ty_max < 0) return 0;
    if (t_min > ty_max || ty_min > t_max) return 0;
    if (ty_min > t_min) t_min = ty_min;
    if (ty_max < t_max) t_max = ty_max;

    if (r->direction.z > 0) {
        tz_min = (bb->p_min.z - r->origin.z) * r->inverse.z;
        tz_max = (bb->p_max.z - r->origin.z) * r->inverse.z;
    } else {
        tz_min = (bb->p_max.z - r->origin.z) * r->inverse.z;
        tz_max = (bb->p_min.z - r->origin.z) * r->inverse.z;
    }
    if (tz_min < 0 && tz_max < 0) return 0;
    if (t_min > tz_max || tz_min > t_max) return 0;
    if (tz_min > t_min) t_min = tz_min;
    if (tz_max < t_max) t_max = tz_max;

    if (t_min < 0) return 1;               /* origin inside the box, cl:132 */
    if (t_min > squared_distance) return 0; /* linear t vs "squared" distance, cl:135 (quirk kept) */
    return 1;
}

/* ------------------------------------------------------------------------- */
/* Textures, h:430-459; sky, cl:438-512                                       */
/* ------------------------------------------------------------------------- */

/* float -> integer as the GPU converts (v_cvt_i32_f32 / v_cvt_u32_f32: NaN gives 0, out-of-range values saturate): what the
 * compiled reference and the integrator do where C leaves the conversion undefined (a hit whose barycentrics are NaN reaches
 * the texture lookup with NaN coordinates: tests/sanitize/oracle_asan.py runs this file under -fsanitize=float-cast-overflow) */
static inline int32_t gpu_f2i(float x)
{
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int32_t)x;
}
static inline uint32_t gpu_f2u(float x)
{
    if (!(x > 0.0f)) return 0u; /* NaN, zero, negative */
    if (x >= 4294967296.0f) return UINT32_MAX;
    return (uint32_t)x;
}

static f4 texture_pixel(const ptmi_texture* tex, const ptmi_uchar4* data, float u, float v)
{
    uint32_t x, y;
    u = u - (float)gpu_f2i(u) + (float)(u < 0 ? 1 : 0);
    v = v - (float)gpu_f2i(v) + (float)(v < 0 ? 1 : 0);
    x = gpu_f2u(u * (float)(tex->width - 1u));
    y = gpu_f2u(v * (float)(tex->height - 1u));
    {
        const uint32_t index = tex->offset + y * tex->width + x;
        const ptmi_uchar4 p = data[index];
        f4 c = mk4(fdivc((float)p.x, 255.f), fdivc((float)p.y, 255.f), fdivc((float)p.z, 255.f), fdivc((float)p.w, 255.f));
        c.w = 1.f - c.w;
        return c;
    }
}

static f4 sky_color(const ptmi_sky* sky, const ptmi_uchar4* data, f4 d)
{
    const float x = mad(sky->cos_rotation_angle, d.x, -(sky->sin_rotation_angle * d.y)); /* cl:441 */
    const float y = mad(sky->sin_rotation_angle, d.x, sky->cos_rotation_angle * d.y);    /* cl:442 */
    const float z = d.z;
    int face = 0;
    float u = 0, v = 0;

    if (fabsf(z) > fabsf(x) && fabsf(z) > fabsf(y)) {
        if (z > 0) { face = 5; u = (1 - fdiv(x, z)) / 2; v = (1 + fdiv(y, z)) / 2; }
        else       { face = 0; u = (1 + fdiv(x, z)) / 2; v = (1 + fdiv(y, z)) / 2; }
    } else if (fabsf(x) > fabsf(y) && fabsf(x) > fabsf(z)) {
        if (x > 0) { face = 1; u = (1 - fdiv(y, x)) / 2; v = (1 + fdiv(z, x)) / 2; }
        else       { face = 3; u = (1 - fdiv(y, x)) / 2; v = (1 - fdiv(z, x)) / 2; }
    } else if (fabsf(y) > fabsf(x) && fabsf(y) > fabsf(z)) {
        if (y > 0) { face = 4; u = (1 + fdiv(x, y)) / 2; v = (1 + fdiv(z, y)) / 2; }
        else       { face = 2; u = (1 + fdiv(x, y)) / 2; v = (1 - fdiv(z, y)) / 2; }
    }
    return texture_pixel(&sky->sky_textures[face], data, u, v); /* Sky_GetFaceColorValue, cl:497-512 */
}

/* ------------------------------------------------------------------------- */
/* Triangle, cl:519-610                                                       */
/* ------------------------------------------------------------------------- */

/* Triangle_GetColorValueAt, cl:591-602 */
static f4 triangle_color_at(const pto_scene* sc, const ptmi_triangle* tri, int positive_normal, float s, float t)
{
    const ptmi_material* mat = &sc->materiaux[positive_normal ? tri->mat_pos : tri->mat_neg];
    float u, v;
    const float b = (1 - s) - t;
    if (mat->is_simple_color) return ld4(&mat->simple_color);
    if (positive_normal) {
        u = mad(tri->uvp3.x, t, mad(tri->uvp1.x, b, tri->uvp2.x * s)); /* cl:600 */
        v = mad(tri->uvp3.y, t, mad(tri->uvp1.y, b, tri->uvp2.y * s));
    } else {
        u = mad(tri->uvn3.x, t, mad(tri->uvn1.x, b, tri->uvn2.x * s));
        v = mad(tri->uvn3.y, t, mad(tri->uvn1.y, b, tri->uvn2.y * s));
    }
    return texture_pixel(&sc->textures[mat->texture_id], sc->textures_data, u, v);
}

/* Triangle_Intersects, cl:519-589.  `shade` = fetch the colour as the
 * reference does on every accepted hit (cl:568-570); the shadow traversal
 * never reads it so it passes 0 there. */
static int triangle_intersects(const pto_scene* sc, const ptmi_triangle* tri, ray_t* r, float* squared_distance,
                               int shade)
{
    const f4 S1 = ld4(&tri->s1), S2 = ld4(&tri->s2), S3 = ld4(&tri->s3), N = ld4(&tri->n);
    r->num_tri++;
    {
        const f4 u = sub4(S2, S1);
        const f4 v = sub4(S3, S1);
        const float d = dot4(N, S1);
        const float nd = dot4(N, r->direction);
        f4 q, full_ray, w;
        float nsd, uv, wv, wu, uu, vv, denom, s, t;

        if ((nd > -0.00001f) && (nd < 0.00001f)) return 0;

        q = mad4s(r->direction, fdiv(d - dot4(N, r->origin), nd), r->origin); /* cl:538 */
        full_ray = sub4(q, r->origin);
        nsd = dot4(full_ray, full_ray);
        if (nsd > *squared_distance) return 0;
        if (nsd < 0.00001f) return 0;

        w = sub4(q, S1);
        uv = dot4(u, v); wv = dot4(w, v); wu = dot4(w, u); uu = dot4(u, u); vv = dot4(v, v);
        denom = fdiv(1, mad(uv, uv, -(uu * vv))); /* cl:556 */
        s = mad(uv, wv, -(vv * wu)) * denom;      /* cl:558 */
        t = mad(uv, wu, -(uu * wv)) * denom;      /* cl:559 */
        if (s < 0 || t < 0 || s + t > 1) return 0;
        if (dot4(full_ray, r->direction) < 0) return 0;

        r->material_id = nd < 0 ? tri->mat_pos : tri->mat_neg;
        if (shade) r->color = triangle_color_at(sc, tri, nd < 0, s, t);
        *squared_distance = nsd;
        r->s = s;
        r->t = t;
        r->point = q;
        return 1;
    }
}

/* Triangle_GetSmoothNormal, cl:604-610 */
static f4 triangle_smooth_normal(const ptmi_triangle* tri, int positive_normal, float s, float t)
{
    /* cl:606: (N2 * s) + (N3 * t) + (N1 * (1 - s - t)) */
    const f4 n = normalize4(mad4s(ld4(&tri->n1), (1 - s) - t, mad4s(ld4(&tri->n2), s, scale4(ld4(&tri->n3), t))));
    return positive_normal ? n : neg4(n);
}

/* ------------------------------------------------------------------------- */
/* BVH traversal, cl:620-783                                                  */
/* ------------------------------------------------------------------------- */

static float dir_component(const f4* d, uint32_t axis) { return ((const float*)d)[axis]; }

/* BVH_IntersectRay, cl:620-702: closest hit, near child first, far pushed. */
static int bvh_intersect_ray(const pto_scene* sc, ray_t* r)
{
    float squared_distance = INFINITY;
    int has_intersection = 0;
    int top = -1;
    const ptmi_node* current = &sc->bvh[0];
    const ptmi_node* stack[PTMI_BVH_MAX_DEPTH];

    for (;;) {
        if (current->is_leaf) {
            uint32_t i;
            for (i = current->triangle_start_index; i < current->triangle_start_index + current->nb_triangles; i++) {
                if (triangle_intersects(sc, &sc->triangulation[i], r, &squared_distance, 1)) {
                    r->triangle_id = i;
                    has_intersection = 1;
                }
            }
            if (top < 0) break;
            current = stack[top--];
        } else {
            const ptmi_node *son1, *son2;
            int b1, b2;
            if (dir_component(&r->direction, current->cut_axis) > 0) {
                son1 = &sc->bvh[current->son1_id];
                son2 = &sc->bvh[current->son2_id];
            } else {
                son2 = &sc->bvh[current->son1_id];
                son1 = &sc->bvh[current->son2_id];
            }
            b1 = bbox_intersects(&son1->triangles_aabb, r, squared_distance);
            b2 = bbox_intersects(&son2->triangles_aabb, r, squared_distance);
            if (b1) {
                if (b2) {
                    if (top + 1 >= PTMI_BVH_MAX_DEPTH) abort(); /* reference would overrun stack[30] */
                    stack[++top] = son2;
                }
                current = son1;
            } else if (b2) {
                current = son2;
            } else {
                if (top < 0) break;
                current = stack[top--];
            }
        }
    }
    return has_intersection;
}

/* BVH_IntersectShadowRay, cl:705-783: any hit; `squared_distance` receives the
 * LINEAR light distance from the caller (cl:938-944, quirk kept). */
static int bvh_intersect_shadow_ray(const pto_scene* sc, ray_t* r, float squared_distance)
{
    int top = -1;
    const ptmi_node* current = &sc->bvh[0];
    const ptmi_node* stack[PTMI_BVH_MAX_DEPTH];

    for (;;) {
        if (current->is_leaf) {
            uint32_t i;
            for (i = current->triangle_start_index; i < current->triangle_start_index + current->nb_triangles; i++)
                if (triangle_intersects(sc, &sc->triangulation[i], r, &squared_distance, 0)) return 1;
            if (top < 0) break;
            current = stack[top--];
        } else {
            const ptmi_node *son1, *son2;
            int b1, b2;
            if (dir_component(&r->direction, current->cut_axis) > 0) {
                son1 = &sc->bvh[current->son1_id];
                son2 = &sc->bvh[current->son2_id];
            } else {
                son2 = &sc->bvh[current->son1_id];
                son1 = &sc->bvh[current->son2_id];
            }
            b1 = bbox_intersects(&son1->triangles_aabb, r, squared_distance);
            b2 = bbox_intersects(&son2->triangles_aabb, r, squared_distance);
            if (b1) {
                if (b2) {
                    if (top + 1 >= PTMI_BVH_MAX_DEPTH) abort();
                    stack[++top] = son2;
                }
                current = son1;
            } else if (b2) {
                current = son2;
            } else {
                if (top < 0) break;
                current = stack[top--];
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Materials, cl:166-416                                                      */
/* ------------------------------------------------------------------------- */

/* shared body of the three Fresnel functions (cl:192-292) */
/* n2_is_literal: the glass and varnish functions divide by a literal (1.55f, 3.f), the water function by a value selected at
 * run time (cl:223-232) - the default build divides differently by the two (fdivc / fdiv) */
static float fresnel_fraction(float n1, float n2, int n2_is_literal, float cos1, f4 incident, f4 N, f4* refraction_dir)
{
    const float sin1 = fsqrt(mad(-cos1, cos1, 1));
    const float sin2 = n2_is_literal ? fdivc(n1 * sin1, n2) : fdiv(n1 * sin1, n2);
    float cos2, r_para, r_perp;
    if (sin2 >= 1) return 1;
    cos2 = fsqrt(mad(-sin2, sin2, 1));
    r_para = fdiv(mad(n2, cos1, -(n1 * cos2)), mad(n2, cos1, n1 * cos2));
    r_perp = fdiv(mad(n1, cos1, -(n2 * cos2)), mad(n1, cos1, n2 * cos2));
    if (refraction_dir) { /* cl:249 (live for water only: n1 / n2 of run-time values) */
        const float ratio = fdiv(n1, n2);
        *refraction_dir = mad4s(incident, ratio, scale4(N, mad(ratio, cos1, -cos2)));
    }
    return mad(r_para, r_para, r_perp * r_perp) / 2.0f;
}

/* Material_FresnelGlassReflectionFraction, cl:192-217 */
static float fresnel_glass(f4 incident, f4 N)
{
    return fresnel_fraction(1, MATERIAL_N_GLASS, 1, -dot4(incident, N), incident, N, NULL);
}

/* Material_FresnelWaterReflectionFraction, cl:219-254.  On total reflection the
 * reference returns (cl:237) before writing its outputs (cl:249-251); its caller
 * reads them in the refraction branch (cl:836-843), which is taken when
 * random() >= 1 - and random() returns exactly 1.0 for the 64 seeds nearest 2^31.
 * So about one interior water hit in 10^8 consumes UNINITIALISED values: undefined
 * in the reference's source (its compiled kernel reads stale registers).  This
 * checker and the integrator agree on a zero direction and the factor n2^2/n1^2
 * there (the caller's initial values / fresnel_water's own computation); round 4
 * found the one such path in 531 M of the configs[4] stand-in. */
static float fresnel_water(f4 incident, f4 N, int already_in_water, f4* refraction_dir, float* mult)
{
    float n1, n2, f;
    if (already_in_water) { n1 = MATERIAL_N_WATER; n2 = 1; } else { n1 = 1; n2 = MATERIAL_N_WATER; }
    f = fresnel_fraction(n1, n2, 0, -dot4(incident, N), incident, N, refraction_dir);
    if (mult) *mult = fdiv(n2 * n2, n1 * n1);
    return f;
}

/* Material_FresnelVarnishReflectionFraction, cl:256-292 (isInVarnish is always false on the live path) */
static float fresnel_varnish(f4 incident, f4 N, f4* refraction_dir)
{
    const float cos1 = fmaxf(0.f, fminf(1.f, -dot4(incident, N)));
    (void)refraction_dir; /* computed by the reference, never read (cl:856) */
    return fresnel_fraction(1.0f, MATERIAL_N_VARNISH, 1, cos1, incident, N, NULL);
}

/* Material_FresnelReflection, cl:294-300 */
static f4 fresnel_reflection(f4 v, f4 N) { return mad4s(neg4(N), 2 * dot4(v, N), v); } /* cl:298 */

/* Material_BRDF, cl:166-190 */
static float material_brdf(const ptmi_material* mat, f4 incident, f4 N, f4 reflected)
{
    if (mat->type == PTMI_MAT_STANDART) return PATH_PI_INVERSE;
    if (mat->type == PTMI_MAT_GLASS) return 1;
    if (mat->type == PTMI_MAT_WATER) {
        const float denom = mad(MATERIAL_KSCHLICK, dot4(incident, reflected), 1);
        return fdiv(1 - MATERIAL_KSCHLICK * MATERIAL_KSCHLICK, 4 * PATH_PI * denom * denom);
    }
    if (mat->type == PTMI_MAT_VARNHISHED) return (1 - fresnel_varnish(incident, N, NULL)) * PATH_PI_INVERSE;
    return 1;
}

/* Material_ConcentricSampleDisk, cl:339-416 */
void pto_concentric_sample_disk(int32_t* seed, float* dx, float* dy)
{
    const float u1 = pto_random(seed);
    const float u2 = pto_random(seed);
    float r, theta, sn, cs;
    const float sx = mad(2, u1, -1); /* cl:350 */
    const float sy = mad(2, u2, -1);

    if (fabsf(sx) < 0.0001f) { r = sy; theta = 0; }
    else if (fabsf(sy) < 0.0001f) { r = sx; theta = 2; }
    else if (sx > -sy) {
        if (sx > sy) { r = sx; if (sy > 0) theta = fdiv(sy, sx); else theta = 8.f + fdiv(sy, sx); }
        else { r = sy; theta = 2.f - fdiv(sx, sy); }
    } else {
        if (sx < sy) { r = -sx; theta = 4.f + fdiv(sy, sx); }
        else { r = -sy; theta = 6.f - fdiv(sy, sx); }
    }
    theta *= PATH_PI / 4.f;
    r = (float)((double)r * 0.999); /* cl:408: unsuffixed literal => double multiply */
    ptmi_sincosf(theta, &sn, &cs);
    *dx = r * cs;
    *dy = r * sn;
}

/* Material_CosineSampleHemisphere, cl:303-336 */
static f4 cosine_sample_hemisphere(int32_t* seed, f4 N)
{
    float x, y, z;
    f4 v, sn, tn;
    pto_concentric_sample_disk(seed, &x, &y);
    z = mad(-y, y, mad(-x, x, 1)); /* cl:311 */
    /* correctly rounded in BOTH builds: the optimizer turns `(z<0) ? 0 : sqrt(z)` (cl:312) into a select and the speculated
     * call loses the !fpmath annotation that makes every other sqrt of the default build v_sqrt_f32 (seen in the optimized
     * IR of the reference kernel: llvm.sqrt without !fpmath at exactly this site) */
    z = (z < 0) ? 0 : sqrtf(z);
    v = mk4(x, y, z, 0);
    if (N.z > 0.9999f) return v;
    if (N.z < -0.9999f) return neg4(v);
    sn = normalize4(mk4(-N.y, N.x, 0, 0));
    tn = normalize4(cross4(N, sn));
    return normalize4(mk4(dot4(mk4(sn.x, tn.x, N.x, 0), v),
                          dot4(mk4(sn.y, tn.y, N.y, 0), v),
                          dot4(mk4(sn.z, tn.z, N.z, 0), v), 0));
}

/* Light_PowerToward, h:403-421 */
static float light_power_toward(const ptmi_light* l, f4 p, f4 N)
{
    const f4 pos = ld4(&l->position), dir = ld4(&l->direction);
    if (l->type == PTMI_LIGHT_DIRECTIONNAL) return l->power * fmaxf(dot4(neg4(dir), N), 0.f);
    if (l->type == PTMI_LIGHT_POINT) {
        const f4 d = sub4(p, pos); /* Vector_SquaredDistanceTo(&position, p): temp = p - position, h:231 */
        return fdiv(l->power, dot4(d, d)) * fmaxf(dot4(normalize4(sub4(pos, p)), N), 0.f);
    }
    if (l->type == PTMI_LIGHT_SPOT) {
        const f4 lrd = normalize4(sub4(p, pos));
        const float cos_angle = dot4(lrd, dir);
        if (cos_angle > l->cos_inner) return l->power * fmaxf(-dot4(lrd, N), 0.f);
        if (cos_angle < l->cos_outer) return 0.0f;
        return fdiv(l->power * (cos_angle - l->cos_outer), l->cos_inner - l->cos_outer) * fmaxf(-dot4(lrd, N), 0);
    }
    return 0.f;
}

/* ------------------------------------------------------------------------- */
/* Scene, cl:791-954                                                          */
/* ------------------------------------------------------------------------- */

/* Scene_ComputeDirectIllumination, cl:901-954 */
static f4 compute_direct_illumination(const pto_scene* sc, ray_t* cam, const ptmi_material* mat, f4 N,
                                      pto_totals* totals)
{
    f4 L = mk4(0, 0, 0, 0);
    const f4 tint = mk4(1, 1, 1, 1);
    uint32_t i;
    for (i = 0; i < sc->lights_size; i++) {
        const ptmi_light* light = &sc->lights[i];
        ray_t lr;
        const f4 full_ray = light->type == PTMI_LIGHT_DIRECTIONNAL ? neg4(ld4(&light->direction))
                                                                   : sub4(ld4(&light->position), cam->point);
        float light_distance, brdf;
        ray_create(&lr, cam->point, full_ray, cam->is_in_water);
        light_distance = light->type == PTMI_LIGHT_DIRECTIONNAL ? INFINITY : length4(full_ray);
        brdf = material_brdf(mat, neg4(lr.direction), N, cam->direction);
        if (totals) totals->shadow_rays++;
        if (!bvh_intersect_shadow_ray(sc, &lr, light_distance))
            L = mad4(scale4(tint, light_power_toward(light, cam->point, N) * brdf), ld4(&light->color), L); /* cl:945 */
        cam->num_bbx += lr.num_bbx;
        cam->num_tri += lr.num_tri;
    }
    return L;
}

/* Scene_ComputeRadiance, cl:791-891 */
static f4 compute_radiance(ray_t* r, int32_t* seed, const ptmi_material* mat, f4 direct, f4* transfer, f4 Ng, f4 Ns,
                           f4* out_dir_dbg)
{
    f4 N = r->direction;
    f4 radiance = mk4(0, 0, 0, 0);
    f4 out = r->direction;

    if (mat->type == PTMI_MAT_STANDART) {
        *transfer = mul4(*transfer, r->color);
        radiance = mul4(direct, *transfer);
        out = cosine_sample_hemisphere(seed, Ns);
        N = Ns;
    } else if (mat->type == PTMI_MAT_GLASS) {
        const float f = fresnel_glass(r->direction, Ns);
        if (pto_random(seed) < f) {
            out = fresnel_reflection(r->direction, Ns);
            N = Ng;
        } else {
            *transfer = mul4(*transfer, scale4(r->color, 1 - mat->opacity));
            N = r->direction;
        }
    } else if (mat->type == PTMI_MAT_WATER) {
        f4 refracted = mk4(0, 0, 0, 0);
        float mult = 0;
        const float f = fresnel_water(r->direction, Ns, r->is_in_water, &refracted, &mult);
        if (pto_random(seed) < f) {
            out = fresnel_reflection(r->direction, Ns);
            N = Ng;
        } else {
            r->is_in_water = !r->is_in_water;
            out = refracted;
            N = neg4(Ng);
            *transfer = scale4(*transfer, mult);
        }
    } else if (mat->type == PTMI_MAT_VARNHISHED) {
        f4 refracted;
        float f1;
        radiance = mad4(mul4(direct, r->color), *transfer, radiance); /* cl:849 */
        f1 = fresnel_varnish(r->direction, Ns, &refracted);
        if (pto_random(seed) < f1) {
            out = fresnel_reflection(r->direction, Ns);
        } else {
            out = cosine_sample_hemisphere(seed, Ns);
            *transfer = mul4(*transfer, r->color);
        }
    }

    out = put_in_same_hemisphere(out, N);
    ray_set_direction(r, out);
    r->origin = mad4s(out, 0.001f, r->point); /* cl:880: the un-normalised out direction */
    if (out_dir_dbg) *out_dir_dbg = out;
    return radiance;
}

/* ------------------------------------------------------------------------- */
/* superSamplingStopCriteria, cl:1152-1172                                    */
/* ------------------------------------------------------------------------- */

static void sample_to_pixel(const pto_scene* sc, float sx, float sy, int* px, int* py)
{
    /* cl:1333-1334 / cl:1159-1160: (sample + 0.5) is a DOUBLE add and multiply */
    int x = (int)(((double)sx + 0.5) * (int)sc->image_width);
    int y = (int)(((double)sy + 0.5) * (int)sc->image_height);
    if (x > (int)sc->image_width - 1) x = (int)sc->image_width - 1;
    if (y > (int)sc->image_height - 1) y = (int)sc->image_height - 1;
    *px = x; *py = y;
}

static int super_sampling_stop(const pto_scene* sc, const pto_buffers* out, const ray_t* r, int32_t* seed)
{
    int px, py, off;
    float n, sx, sy, sz, sigma2_n;
    sample_to_pixel(sc, r->sample_x, r->sample_y, &px, &py);
    off = py * (int)sc->image_width + px;
    n = out->image_ray_nb[off];
    sx = fdiv(out->image_v[4 * off + 0], n);
    sy = fdiv(out->image_v[4 * off + 1], n);
    sz = fdiv(out->image_v[4 * off + 2], n);
    sigma2_n = fmaxf(fmaxf(sx, sy), sz);
    {
        uint32_t idx = (uint32_t)n;
        if (idx > 1000u) idx = 1000u; /* the reference reads past its 1001-entry table for n > 1000 (no clamp) */
        return (double)pto_random(seed) > (double)fdiv(100 * sigma2_n, sc->x2inv[idx]) + 0.05;
    }
}

/* ------------------------------------------------------------------------- */
/* Kernel_Main, cl:1180-1350                                                  */
/* ------------------------------------------------------------------------- */

static int kernel_main_impl(const pto_scene* sc, uint32_t gx, uint32_t gy, uint32_t iteration,
                            const pto_buffers* out, pto_totals* totals, pto_bounce* trace, int max_trace,
                            float radiance_out[4])
{
    ray_t r;
    int32_t seed = pto_initialize_random_seed(gx, gy, sc->image_width, sc->image_height, iteration);
    float sample[2];
    if (sc->source_seed && seed == 0) seed = 1; /* non-parity mode: the zero test where the source text has it, on the square */
    f4 shot, radiance = mk4(0, 0, 0, 0), transfer = mk4(1, 1, 1, 1);
    int active = 1, n_trace = 0;

    pto_sampler(sc->sampler, gx, gy, sc->image_width, sc->image_height, iteration, &seed, sample);
    shot = mad4s(ld4(&sc->camera_up), sample[1], mad4s(ld4(&sc->camera_right), sample[0], ld4(&sc->camera_direction))); /* cl:1213 */
    ray_create(&r, ld4(&sc->camera_position), shot, 0);
    r.sample_x = sample[0];
    r.sample_y = sample[1];
    r.point = mk4(0, 0, 0, 0); r.color = mk4(0, 0, 0, 0);
    r.triangle_id = 0; r.material_id = 0; r.s = 0; r.t = 0;

    if (sc->super_sampling && out && iteration > MIN_REFLECTION_NUMBER && super_sampling_stop(sc, out, &r, &seed))
        return 0; /* cl:1219-1222 */

    while (active && r.reflection_id < sc->ray_max_depth) {
        if (totals) totals->segments++;
        if (bvh_intersect_ray(sc, &r)) {
            const ptmi_triangle* tri = &sc->triangulation[r.triangle_id];
            const ptmi_material* mat = &sc->materiaux[r.material_id];
            const int same_dir = dot4(r.direction, ld4(&tri->n)) > 0;
            const f4 Ng = !same_dir ? ld4(&tri->n) : neg4(ld4(&tri->n)); /* Triangle_GetNormal, h:500 */
            f4 Ns = triangle_smooth_normal(tri, !same_dir, r.s, r.t);
            f4 direct, out_dir, rad;
            Ns = put_in_same_hemisphere(Ns, neg4(r.direction));
            Ns = normalize4(Ns);
            direct = compute_direct_illumination(sc, &r, mat, Ns, totals);
            rad = compute_radiance(&r, &seed, mat, direct, &transfer, Ng, Ns, &out_dir);
            radiance = add4(radiance, rad);
            r.reflection_id++;
            if (trace && n_trace < max_trace) {
                pto_bounce* b = &trace[n_trace++];
                b->triangle_id = r.triangle_id; b->material_id = r.material_id;
                b->s = r.s; b->t = r.t;
                memcpy(b->point, &r.point, 16); memcpy(b->ns, &Ns, 16); memcpy(b->out_dir, &out_dir, 16);
                memcpy(b->transfer, &transfer, 16); memcpy(b->radiance, &radiance, 16);
                b->seed_after = seed; b->n_bbx = r.num_bbx; b->n_tri = r.num_tri;
            }
        } else {
            active = 0;
            radiance = mad4(sky_color(sc->sky, sc->textures_data, r.direction), transfer, radiance); /* cl:1287 */
        }
        if (active) { /* cl:1296-1304 */
            const float max_contribution = cl_max(transfer.x, cl_max(transfer.y, transfer.z));
            if (max_contribution <= MIN_CONTRIBUTION_VALUE) active = 0;
            /* cl:1306-1314, commented out in the reference (RUSSIAN_ROULETTE false, h:12): as written there */
            if (sc->russian_roulette && active && r.reflection_id > MIN_REFLECTION_NUMBER) {
                const float coeff = fdiv(max_contribution, (float)(r.reflection_id - MIN_REFLECTION_NUMBER));
                if (coeff < 1) {
                    active = active && (pto_random(&seed) > coeff);
                    transfer = div4s(transfer, coeff);
                }
            }
        }
    }

    if (radiance_out) memcpy(radiance_out, &radiance, 16);
    if (totals) {
        totals->paths++;
        totals->surface_hits += r.reflection_id;
        totals->box_tests += r.num_bbx;
        totals->triangle_tests += r.num_tri;
    }
    if (!out) return n_trace > 0 ? n_trace : 1;

    /* statistics, cl:1319-1331 */
    if (out->ray_depths) out->ray_depths[r.reflection_id]++;
    if (out->ray_intersected_bbx && r.num_bbx < PTMI_MAX_INTERSECTION_NUMBER) out->ray_intersected_bbx[r.num_bbx]++;
    if (out->ray_intersected_tri && r.num_tri < PTMI_MAX_INTERSECTION_NUMBER) out->ray_intersected_tri[r.num_tri]++;

    { /* framebuffer read-modify-write, cl:1333-1349 */
        int px, py, off;
        f4 before, after;
        float n_before, n_after;
        sample_to_pixel(sc, r.sample_x, r.sample_y, &px, &py);
        off = py * (int)sc->image_width + px;
        memcpy(&before, &out->image_color[4 * off], 16);
        after = add4(before, radiance);
        n_before = out->image_ray_nb[off];
        n_after = n_before + 1.f;
        out->image_ray_nb[off] = n_after;
        memcpy(&out->image_color[4 * off], &after, 16);
        if (out->image_v) {
            if (iteration == 0 || (sc->first_sample_guard && n_before == 0.f)) {
                memset(&out->image_v[4 * off], 0, 16);
            } else {
                f4 v;
                memcpy(&v, &out->image_v[4 * off], 16);
                v = mad4(sub4(radiance, div4s(before, n_before)), sub4(radiance, div4s(after, n_after)), v); /* cl:1349 */
                memcpy(&out->image_v[4 * off], &v, 16);
            }
        }
    }
    return 1;
}

int pto_kernel_main(const pto_scene* sc, uint32_t gx, uint32_t gy, uint32_t iteration, const pto_buffers* out,
                    pto_totals* totals)
{
    return kernel_main_impl(sc, gx, gy, iteration, out, totals, NULL, 0, NULL);
}

int pto_trace_path(const pto_scene* sc, uint32_t gx, uint32_t gy, uint32_t iteration, pto_bounce* bounces,
                   int max_bounces, float radiance_out[4])
{
    pto_scene s = *sc;
    s.super_sampling = 0;
    return kernel_main_impl(&s, gx, gy, iteration, NULL, NULL, bounces, max_bounces, radiance_out);
}

/* ------------------------------------------------------------------------- */
/* driver                                                                     */
/* ------------------------------------------------------------------------- */

typedef struct {
    const pto_scene* sc;
    uint32_t first, n, y0, y1;
    pto_buffers out; /* histograms are thread-private */
    pto_totals totals;
} worker_t;

static void* worker_main(void* p)
{
    worker_t* w = (worker_t*)p;
    uint32_t x, y, it;
    for (y = w->y0; y < w->y1; y++)
        for (x = 0; x < w->sc->image_width; x++)
            for (it = w->first; it < w->first + w->n; it++)
                pto_kernel_main(w->sc, x, y, it, &w->out, &w->totals);
    return NULL;
}

void pto_render(const pto_scene* sc, uint32_t first, uint32_t n, const pto_buffers* out, int n_threads,
                pto_totals* totals)
{
    pto_render_rows(sc, first, n, 0, sc->image_height, out, n_threads, totals);
}

void pto_render_rows(const pto_scene* sc, uint32_t first, uint32_t n, uint32_t row0, uint32_t row1,
                     const pto_buffers* out, int n_threads, pto_totals* totals)
{
    const uint32_t H = row1 - row0, W = sc->image_width, D = sc->ray_max_depth;
    pto_totals tt;
    memset(&tt, 0, sizeof tt);

    if (n_threads <= 1 || sc->sampler == PTMI_SAMPLER_RANDOM || sc->super_sampling) {
        /* the reference's order: one full image per iteration (OpenCL.cpp:76-107) */
        uint32_t it, x, y;
        for (it = first; it < first + n; it++)
            for (y = row0; y < row1; y++)
                for (x = 0; x < W; x++) pto_kernel_main(sc, x, y, it, out, &tt);
    } else {
        worker_t* ws;
        pthread_t* th;
        int i;
        if ((uint32_t)n_threads > H) n_threads = (int)H;
        ws = (worker_t*)calloc((size_t)n_threads, sizeof *ws);
        th = (pthread_t*)calloc((size_t)n_threads, sizeof *th);
        for (i = 0; i < n_threads; i++) {
            worker_t* w = &ws[i];
            w->sc = sc; w->first = first; w->n = n;
            w->y0 = row0 + (uint32_t)((uint64_t)H * (uint64_t)i / (uint64_t)n_threads);
            w->y1 = row0 + (uint32_t)((uint64_t)H * (uint64_t)(i + 1) / (uint64_t)n_threads);
            w->out = *out;
            w->out.ray_depths = (uint32_t*)calloc(D + 1, 4);
            w->out.ray_intersected_bbx = (uint32_t*)calloc(PTMI_MAX_INTERSECTION_NUMBER, 4);
            w->out.ray_intersected_tri = (uint32_t*)calloc(PTMI_MAX_INTERSECTION_NUMBER, 4);
            pthread_create(&th[i], NULL, worker_main, w);
        }
        for (i = 0; i < n_threads; i++) {
            uint32_t k;
            worker_t* w = &ws[i];
            pthread_join(th[i], NULL);
            for (k = 0; k <= D; k++) if (out->ray_depths) out->ray_depths[k] += w->out.ray_depths[k];
            for (k = 0; k < PTMI_MAX_INTERSECTION_NUMBER; k++) {
                if (out->ray_intersected_bbx) out->ray_intersected_bbx[k] += w->out.ray_intersected_bbx[k];
                if (out->ray_intersected_tri) out->ray_intersected_tri[k] += w->out.ray_intersected_tri[k];
            }
            free(w->out.ray_depths); free(w->out.ray_intersected_bbx); free(w->out.ray_intersected_tri);
            tt.paths += w->totals.paths; tt.segments += w->totals.segments;
            tt.surface_hits += w->totals.surface_hits; tt.shadow_rays += w->totals.shadow_rays;
            tt.box_tests += w->totals.box_tests; tt.triangle_tests += w->totals.triangle_tests;
        }
        free(ws); free(th);
    }
    if (totals) *totals = tt;
}

/* ------------------------------------------------------------------------- */
/* unit-level wrappers                                                        */
/* ------------------------------------------------------------------------- */

static f4 arr4(const float v[4]) { return mk4(v[0], v[1], v[2], v[3]); }

int pto_bounding_box_intersects(const ptmi_bounding_box* bb, const float origin[4], const float direction[4],
                                float squared_distance)
{
    ray_t r;
    ray_create(&r, arr4(origin), arr4(direction), 0);
    return bbox_intersects(bb, &r, squared_distance);
}

int pto_triangle_intersects(const ptmi_triangle* tri, const float origin[4], const float direction[4],
                            float* squared_distance, float* s, float* t, float point[4])
{
    ray_t r;
    int hit;
    ray_create(&r, arr4(origin), arr4(direction), 0);
    hit = triangle_intersects(NULL, tri, &r, squared_distance, 0);
    if (hit) { *s = r.s; *t = r.t; memcpy(point, &r.point, 16); }
    return hit;
}

void pto_cosine_sample_hemisphere(int32_t* seed, const float n[4], float out[4])
{
    const f4 v = cosine_sample_hemisphere(seed, arr4(n));
    memcpy(out, &v, 16);
}

float pto_fresnel_glass(const float incident[4], const float n[4]) { return fresnel_glass(arr4(incident), arr4(n)); }
float pto_fresnel_varnish(const float incident[4], const float n[4]) { return fresnel_varnish(arr4(incident), arr4(n), NULL); }

void pto_sky_color(const ptmi_sky* sky, const ptmi_uchar4* data, const float direction[4], float rgba[4])
{
    const f4 c = sky_color(sky, data, arr4(direction));
    memcpy(rgba, &c, 16);
}

void pto_sincos(float x, float* s, float* c) { ptmi_sincosf(x, s, c); }

/* the operators of oracle/arith_probe.cl in this build's arithmetic: out[k * n + i], k = 0..9 (GPU test: the default build of
 * this file == the OpenCL compiler's output == the product's default-arithmetic mode) */
void pto_arith_probe(const float* a, const float* b, uint32_t n, float* out)
{
    uint32_t i;
    for (i = 0; i < n; i++) {
        const float x = a[i], y = b[i];
        out[0 * (size_t)n + i] = fdiv(x, y);
        out[1 * (size_t)n + i] = fdiv(1.0f, x);
        out[2 * (size_t)n + i] = fdivc(x, 255.f);
        out[3 * (size_t)n + i] = fdivc(x, 3.f);
        out[4 * (size_t)n + i] = fdivc(x, 1.55f);
        out[5 * (size_t)n + i] = fdivc(x, (float)1920);
        out[6 * (size_t)n + i] = fdivc(x, (float)90);
        out[7 * (size_t)n + i] = fsqrt(x);
        out[8 * (size_t)n + i] = length4(mk4(x, y, x * 0.5f, 0.0f));
        out[9 * (size_t)n + i] = mad(x, y, 1.0f);
    }
}
