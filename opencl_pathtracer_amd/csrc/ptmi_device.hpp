// ptmi_device.hpp - device-side building blocks of the integrator (gfx950).
//
// Numerics contract (DESIGN.md "Numerics"): every fp32 operation here is one
// correctly rounded IEEE-754 +,-,*,/, sqrt or explicit fma in a fixed order,
// compiled with -ffp-contract=off, so that a CPU evaluation of the same formulas
// gives the same bits.  The formulas are those of the reference kernel; each
// function cites the reference lines whose result it must reproduce.  OpenCL
// float4 semantics apply: dot/length/normalize run over all FOUR components
// (Appendix A.1 of SURVEY.md: the w lanes matter after a GLASS/WATER reflection).
#pragma once

#include <hip/hip_runtime.h>

#include "ptmi_detmath.h"
#include "ptmi_internal.h"

// ---- the two arithmetic modes ---------------------------------------------------------------------------
// The reference compiles its kernel at run time with no floating-point option (PathTracer_OpenCL.cpp:292-314), so what it
// computes is its OpenCL compiler's DEFAULT arithmetic; with -ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt the
// same source gives the STRICT arithmetic.  Both are restated here bit for bit; the device sources are compiled once per
// mode (Makefile) into two namespaces, and a context picks one (PTMI_FLAG_DEFAULT_ARITHMETIC).
//   strict  (PTMI_DEFAULT_ARITHMETIC 0): every + - * / sqrt of the source is one correctly rounded operation.
//   default (PTMI_DEFAULT_ARITHMETIC 1): what clang's OpenCL front end and the gfx950 back end make of the same source:
//     * `a * b + c` written inside ONE expression (also `x += a * b`) is one fused multiply-add (-ffp-contract=on is a
//       source-level rule: llvm.fmuladd where the multiplication is a direct operand of the addition; the left operand is
//       tried first).  Each such site below is a mad() and cites the reference line;
//     * a / b   = ldexp(frexp_mant(a) * v_rcp_f32(frexp_mant(b)), frexp_exp(a) - frexp_exp(b))  (the 2.5-ulp division of
//       AMDGPUCodeGenPrepare with denormals enabled); with a CONSTANT divisor the reciprocal of its mantissa is folded at
//       compile time, i.e. correctly rounded instead of the instruction's value: fdiv(a, DivC);
//     * sqrt(x) = v_sqrt_f32 behind a 2^32 scaling of denormal inputs.
//   The platform library (dot, cross, normalize, length - bare v_sqrt_f32 included -, sin, cos) is the same code in both modes.
#ifndef PTMI_DEFAULT_ARITHMETIC
#define PTMI_DEFAULT_ARITHMETIC 0
#endif
#if PTMI_DEFAULT_ARITHMETIC
#define PTMI_DEV_NS ptmi_dev_da
#define PTMI_ARITH(name) name##_da
#else
#define PTMI_DEV_NS ptmi_dev
#define PTMI_ARITH(name) name
#endif

namespace PTMI_DEV_NS {

using namespace ptmi_internal;

constexpr bool kDefaultArithmetic = PTMI_DEFAULT_ARITHMETIC != 0;

struct V4 {
    float x, y, z, w;
};

__device__ __forceinline__ V4 v4(float x, float y, float z, float w) { return V4{x, y, z, w}; }
__device__ __forceinline__ V4 v4(const float* p) { return V4{p[0], p[1], p[2], p[3]}; }
__device__ __forceinline__ V4 v4(const ptmi_float4& p) { return V4{p.x, p.y, p.z, p.w}; }
__device__ __forceinline__ V4 v4(const float4& p) { return V4{p.x, p.y, p.z, p.w}; }
__device__ __forceinline__ V4 operator+(V4 a, V4 b) { return V4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
__device__ __forceinline__ V4 operator-(V4 a, V4 b) { return V4{a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
__device__ __forceinline__ V4 operator*(V4 a, V4 b) { return V4{a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
__device__ __forceinline__ V4 operator*(V4 a, float s) { return V4{a.x * s, a.y * s, a.z * s, a.w * s}; }
__device__ __forceinline__ V4 operator/(V4 a, float s) { return V4{a.x / s, a.y / s, a.z / s, a.w / s}; }
__device__ __forceinline__ V4 operator-(V4 a) { return V4{-a.x, -a.y, -a.z, -a.w}; }

// a * b + c where the reference writes it in one expression: two roundings (strict) or one (default)
__device__ __forceinline__ float mad(float a, float b, float c)
{
    if (kDefaultArithmetic) return __builtin_fmaf(a, b, c);
    return a * b + c;
}
__device__ __forceinline__ V4 mad(V4 a, V4 b, V4 c) { return V4{mad(a.x, b.x, c.x), mad(a.y, b.y, c.y), mad(a.z, b.z, c.z), mad(a.w, b.w, c.w)}; }
__device__ __forceinline__ V4 mad(V4 a, float s, V4 c) { return V4{mad(a.x, s, c.x), mad(a.y, s, c.y), mad(a.z, s, c.z), mad(a.w, s, c.w)}; }
// a / b, b a run-time value
__device__ __forceinline__ float fdiv(float a, float b)
{
    if (!kDefaultArithmetic) return a / b;
    return __builtin_ldexpf(__builtin_amdgcn_frexp_mantf(a) * __builtin_amdgcn_rcpf(__builtin_amdgcn_frexp_mantf(b)),
                            __builtin_amdgcn_frexp_expf(a) - __builtin_amdgcn_frexp_expf(b));
}
// 1.0f / b (the compiler's form for a numerator of one: the same value as fdiv(1, b), two instructions fewer)
__device__ __forceinline__ float frcp(float b)
{
    if (!kDefaultArithmetic) return 1.0f / b;
    return __builtin_ldexpf(__builtin_amdgcn_rcpf(__builtin_amdgcn_frexp_mantf(b)), -__builtin_amdgcn_frexp_expf(b));
}
// a / c, c a constant of the reference's program (a literal, or IMAGE_WIDTH / IMAGE_HEIGHT which it bakes in with -D):
// c = m * 2^e, m in [0.5, 1); inv_mant = 1 / m correctly rounded
struct DivC {
    float c, inv_mant;
    int exp;
};
constexpr DivC make_divc(float c)  // (c positive and normal)
{
    const uint32_t bits = __builtin_bit_cast(uint32_t, c);
    const float m = __builtin_bit_cast(float, (bits & 0x007FFFFFu) | 0x3F000000u);
    return DivC{c, 1.0f / m, (int)((bits >> 23) & 0xFFu) - 126};
}
__device__ __forceinline__ float fdiv(float a, const DivC& d)
{
    if (!kDefaultArithmetic) return a / d.c;
    return __builtin_ldexpf(__builtin_amdgcn_frexp_mantf(a) * d.inv_mant, __builtin_amdgcn_frexp_expf(a) - d.exp);
}
// sqrt(x) where the compiler is NOT asked for the correctly rounded one: v_sqrt_f32 behind a 2^32 scaling of denormal inputs.
// The default build's sqrt - and the square root inside the platform library's length() in BOTH builds (below).
__device__ __forceinline__ float approx_sqrt(float x)
{
    const bool tiny = x < 0x1p-126f;
    const float r = __builtin_amdgcn_sqrtf(tiny ? __builtin_ldexpf(x, 32) : x);
    return tiny ? __builtin_ldexpf(r, -16) : r;
}
__device__ __forceinline__ float fsqrt(float x)
{
    if (!kDefaultArithmetic) return sqrtf(x);
    return approx_sqrt(x);
}

// The OpenCL geometric builtins, fixed to the definitions of the OpenCL library the reference meets on
// this hardware (ROCm device libs, opencl.bc): dot and cross are its FMA chains verbatim, normalize below.
// DESIGN.md "Numerics".
__device__ __forceinline__ float dot(V4 a, V4 b)
{
    return __builtin_fmaf(a.w, b.w, __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)));
}
// length(float4) of the library (_Z6lengthDv4_f): sqrt(dot) with a rescaling for squared lengths outside the normal range.
// The same code in both builds of the reference: -cl-fp32-correctly-rounded-divide-sqrt reaches the kernel's own sqrt()
// calls, not the library's - its square root stays the bare v_sqrt_f32 (ISA of both builds; found by the fuzzed scenes of
// tests/test_reference_default_gpu.py, where a light on a vertex makes the shadow ray's limit - this length, cl:938 - decide
// box tests by its last bit).
__device__ __forceinline__ float length(V4 a)
{
    const float d = dot(a, a);
    if (d < 0x1p-126f) {
        a = a * 0x1p+86f;
        return approx_sqrt(dot(a, a)) * 0x1p-86f;
    }
    if (d == INFINITY) {
        a = a * 0x1p-66f;
        return approx_sqrt(dot(a, a)) * 0x1p+66f;
    }
    return __builtin_amdgcn_sqrtf(d);
}
// normalize(float4) of the ROCm OpenCL library (opencl.bc, _Z9normalizeDv4_f): the vector times rsqrt(dot), where rsqrt is
// __ocml_rsqrt_f32 = the hardware's v_rsq_f32 (one ulp off the correctly rounded value for 11 % of the inputs) behind range
// scaling for tiny / infinite squared lengths; the zero vector is returned as it is.  Written out in full so that the
// integrator and the reference kernel compiled for this GPU agree to the last bit; the CPU checker reproduces v_rsq_f32
// from a table measured on the hardware (oracle/pt_oracle.c).
__device__ __forceinline__ float cl_rsqrt(float x)
{
    const bool tiny = x < 0x1p-126f;
    const float r = __builtin_amdgcn_rsqf(tiny ? x * 0x1p+24f : x);
    return tiny ? r * 4096.0f : r;
}
__device__ __forceinline__ V4 normalize(V4 a)
{
    if ((a.x == 0) & (a.y == 0) & (a.z == 0) & (a.w == 0)) return a;
    float d = dot(a, a);
    if (d < 0x1p-126f) {
        a = a * 0x1p+86f;
        d = dot(a, a);
    } else if (d == INFINITY) {
        a = a * 0x1p-66f;
        d = dot(a, a);
        if (d == INFINITY) {
            a = V4{__builtin_copysignf(__builtin_isinf(a.x) ? 1.0f : 0.0f, a.x), __builtin_copysignf(__builtin_isinf(a.y) ? 1.0f : 0.0f, a.y),
                   __builtin_copysignf(__builtin_isinf(a.z) ? 1.0f : 0.0f, a.z), __builtin_copysignf(__builtin_isinf(a.w) ? 1.0f : 0.0f, a.w)};
            d = dot(a, a);
        }
    }
    return a * cl_rsqrt(d);
}
__device__ __forceinline__ V4 cross(V4 a, V4 b)
{
    return V4{__builtin_fmaf(a.y, b.z, b.y * -a.z), __builtin_fmaf(a.z, b.x, b.z * -a.x),
              __builtin_fmaf(a.x, b.y, b.x * -a.y), 0.0f};
}

constexpr float kPi = 3.14159265f;              // PATH_PI, header.cl:10
constexpr float kPiInverse = 0.31830988618f;    // PATH_PI_INVERSE, header.cl:11
constexpr float kMinContribution = 0.001f;      // MIN_CONTRIBUTION_VALUE, header.cl:15
constexpr float kNWater = 1.333f, kNGlass = 1.55f, kNVarnish = 3.f, kSchlick = 0.8f;  // header.cl:166-169

struct Ray {
    V4 o, d;
    float ix, iy, iz;  // inverse.xyz (inverse.w is never read)
};

// Ray3D_SetDirection, header.cl:276-284
__device__ __forceinline__ void ray_set_direction(Ray& r, V4 d)
{
    r.d = normalize(d);
    r.ix = frcp(r.d.x);
    r.iy = frcp(r.d.y);
    r.iz = frcp(r.d.z);
}

// random(), header.cl:246-253.  Only the low 31 bits of the 64-bit product
// survive the mask, so a 32-bit multiply gives the same seed.
__device__ __forceinline__ float lcg_random(int& seed)
{
    seed = (int)((16807u * (uint32_t)seed) & 0x7FFFFFFFu);
    return (float)seed / 2147483648.0f;
}

// InitializeRandomSeed, header.cl:255-264 (all arithmetic is modulo 2^32).  The reference's `if(seed == 0) seed = 1` follows a
// signed square whose overflow is undefined; the OpenCL compilers it meets (LLVM: verified in both gfx950 code objects of the
// reference kernel built by this image's clang 22 / ROCm 7.2, tests/test_reference_strict_gpu.py) test the un-squared index
// instead, so a square that wraps to 0 leaves the seed 0 - and every random number of that path 0.  Bit parity with the
// compiled reference is the contract, so this follows the compiler, not the source text.
// `source_rule` (PTMI_FLAG_SOURCE_SEED, non-parity): the test as the source text reads under wrapping arithmetic - on the square.
__device__ __forceinline__ int lcg_seed(uint32_t gx, uint32_t gy, uint32_t w, uint32_t h, uint32_t iteration, bool source_rule = false)
{
    const uint32_t index = gx + gy * w + iteration * w * h;
    uint32_t s = index * 2011u;
    s *= s;
    return (int)((source_rule ? s == 0u : index == 0u) ? 1u : s);
}

// Vector_PutInSameHemisphereAs, header.cl:237-244
__device__ __forceinline__ V4 put_in_same_hemisphere(V4 v, V4 n)
{
    const float d = dot(v, n);
    if (d < 0.001f) v = mad(n, 0.01f - d, v);  // h:241 `(*This) += (*N) * (0.01f - dotProd)`
    return v;
}

// BoundingBox_Intersects, FullKernel.cl:64-139, evaluated without branches.
// Every comparison of the reference is kept with its operand order (NaNs from
// 0*inf compare false exactly where they do there); the early returns become
// one conjunction because the function has no side effects besides the counter,
// which the caller bumps.
__device__ __forceinline__ bool box_hit(const float lo[3], const float hi[3], bool is_empty, const Ray& r, float limit)
{
    // bitwise & | on the comparison results: they become scalar ops on lane masks; && || would be compiled
    // into a chain of ~30 tiny divergent blocks (measured: 5 % slower)
    const bool px = r.d.x > 0, py = r.d.y > 0, pz = r.d.z > 0;
    float tmin = ((px ? lo[0] : hi[0]) - r.o.x) * r.ix;
    float tmax = ((px ? hi[0] : lo[0]) - r.o.x) * r.ix;
    bool miss = (tmin < 0) & (tmax < 0);
    const float tymin = ((py ? lo[1] : hi[1]) - r.o.y) * r.iy;
    const float tymax = ((py ? hi[1] : lo[1]) - r.o.y) * r.iy;
    miss |= (tymin < 0) & (tymax < 0);
    miss |= (tmin > tymax) | (tymin > tmax);
    tmin = tymin > tmin ? tymin : tmin;
    tmax = tymax < tmax ? tymax : tmax;
    const float tzmin = ((pz ? lo[2] : hi[2]) - r.o.z) * r.iz;
    const float tzmax = ((pz ? hi[2] : lo[2]) - r.o.z) * r.iz;
    miss |= (tzmin < 0) & (tzmax < 0);
    miss |= (tmin > tzmax) | (tzmin > tmax);
    tmin = tzmin > tmin ? tzmin : tmin;
    miss |= !(tmin < 0) & (tmin > limit);
    return !(miss | is_empty);
}

// The same decision in 5 compare/select instructions per box instead of 16, valid when nothing in it can be a NaN
// and the box is ordered: box coordinates finite with lo <= hi (checked at upload), ray origin not NaN and the three
// reciprocals finite (checked per ray, `ray_slabs_are_ordered`).  Then every slab has tmin_a <= tmax_a (rounding is
// monotonic), so
//   "both ends negative on some axis"  (FullKernel.cl:85-86,99-100,119-120)  ==  min_a tmax_a < 0
//   every cross test  tmin_a > tmax_b  (:101-102,121-122)                    ==  max_a tmin_a > min_a tmax_a
//   the distance test  !(tmin < 0) && tmin > limit  (:136)                   ==  tmin > limit   (limit is never negative)
// Rays with a zero direction component (reciprocal +-inf: 0 * inf = NaN at a box plane through the origin, and the
// reference's "+0 fails every box" quirk) never get here: they keep the literal form above.
// An EMPTY box (isEmpty: never hit, FullKernel.cl:68) is stored for this form as lo = +inf, hi = -inf, which gives
// tmin = +inf > tmax = -inf for either direction sign, so no flag has to be tested.
typedef float ptmi_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bool box_hit_ordered(const float lo[3], const float hi[3], const Ray& r, float limit)
{
    // two-wide arithmetic spelled out (v_pk_add_f32 / v_pk_mul_f32: same IEEE operations, half the instructions):
    // (x, y) of the near planes, (x, y) of the far planes, and (near, far) of z
    const bool px = r.d.x > 0, py = r.d.y > 0, pz = r.d.z > 0;
    const ptmi_f2 o_xy = {r.o.x, r.o.y}, i_xy = {r.ix, r.iy}, o_zz = {r.o.z, r.o.z}, i_zz = {r.iz, r.iz};
    const ptmi_f2 near_xy = {px ? lo[0] : hi[0], py ? lo[1] : hi[1]}, far_xy = {px ? hi[0] : lo[0], py ? hi[1] : lo[1]};
    const ptmi_f2 z_nf = {pz ? lo[2] : hi[2], pz ? hi[2] : lo[2]};
    const ptmi_f2 tn = (near_xy - o_xy) * i_xy, tf = (far_xy - o_xy) * i_xy, tz = (z_nf - o_zz) * i_zz;
    const float tmin = __builtin_fmaxf(__builtin_fmaxf(tn.x, tn.y), tz.x);
    const float tmax = __builtin_fminf(__builtin_fminf(tf.x, tf.y), tz.y);
    return !((tmax < 0) | (tmin > tmax) | (tmin > limit));
}
__device__ __forceinline__ bool ray_slabs_are_ordered(const Ray& r)
{
    return (__builtin_fabsf(r.ix) < INFINITY) & (__builtin_fabsf(r.iy) < INFINITY) & (__builtin_fabsf(r.iz) < INFINITY) &
           (r.o.x == r.o.x) & (r.o.y == r.o.y) & (r.o.z == r.o.z);
}

struct Hit {
    V4 point;       // intersectionPoint
    float s, t;
    uint32_t tri;   // intersectedTriangleId
    bool front;     // N.dir < 0  => materialWithPositiveNormalIndex (FullKernel.cl:568)
};

// Triangle_Intersects, FullKernel.cl:519-589, without the colour fetch (done
// once for the final hit: the fetch is a pure function of triangle, side, s, t).
// Early exits kept on purpose: measured on MI355X, the straight-line form (all rejections OR-ed at the end)
// was 29 % slower on the 1M-triangle workload -- whole groups of lanes do leave at the distance test.
__device__ __forceinline__ bool tri_hit(const V4 S1, const V4 S2, const V4 S3, const V4 N, const Ray& r, float& limit,
                                        Hit& h)
{
    const V4 u = S2 - S1;
    const V4 v = S3 - S1;
    const float d = dot(N, S1);
    const float nd = dot(N, r.d);
    if ((nd > -0.00001f) && (nd < 0.00001f)) return false;
    const V4 q = mad(r.d, fdiv(d - dot(N, r.o), nd), r.o);  // cl:538
    const V4 full = q - r.o;
    const float nsd = dot(full, full);
    if (nsd > limit) return false;
    if (nsd < 0.00001f) return false;
    const V4 w = q - S1;
    const float uv = dot(u, v), wv = dot(w, v), wu = dot(w, u), uu = dot(u, u), vv = dot(v, v);
    const float denom = frcp(mad(uv, uv, -(uu * vv)));  // cl:556
    const float s = mad(uv, wv, -(vv * wu)) * denom;    // cl:558
    const float t = mad(uv, wu, -(uu * wv)) * denom;    // cl:559
    if (s < 0 || t < 0 || s + t > 1) return false;
    if (dot(full, r.d) < 0) return false;
    limit = nsd;
    h.point = q;
    h.s = s;
    h.t = t;
    h.front = nd < 0;
    return true;
}

// The same test on a DTriPre record: d, the edge vectors and the reciprocal determinant come from the upload.
// uv, uu, vv are re-evaluated (12 fma) with the w lanes the generic form would see (+0 * +0), and w.w = q.w - S1.w
// still enters wv / wu multiplied by that +0, so every intermediate equals the generic form's bit for bit.
__device__ __forceinline__ bool tri_hit_pre(const V4 N, const V4 s1d, const V4 u_den, const V4 v_s1w, const Ray& r,
                                            float& limit, Hit& h)
{
    const V4 S1 = v4(s1d.x, s1d.y, s1d.z, v_s1w.w);
    const V4 u = v4(u_den.x, u_den.y, u_den.z, 0.0f);
    const V4 v = v4(v_s1w.x, v_s1w.y, v_s1w.z, 0.0f);
    const float d = s1d.w;
    const float nd = dot(N, r.d);
    if ((nd > -0.00001f) && (nd < 0.00001f)) return false;
    const V4 q = mad(r.d, fdiv(d - dot(N, r.o), nd), r.o);  // cl:538
    const V4 full = q - r.o;
    const float nsd = dot(full, full);
    if (nsd > limit) return false;
    if (nsd < 0.00001f) return false;
    const V4 w = q - S1;
    const float uv = dot(u, v), wv = dot(w, v), wu = dot(w, u), uu = dot(u, u), vv = dot(v, v);
    const float denom = u_den.w;
    const float s = mad(uv, wv, -(vv * wu)) * denom;  // cl:558
    const float t = mad(uv, wu, -(uu * wv)) * denom;  // cl:559
    if (s < 0 || t < 0 || s + t > 1) return false;
    if (dot(full, r.d) < 0) return false;
    limit = nsd;
    h.point = q;
    h.s = s;
    h.t = t;
    h.front = nd < 0;
    return true;
}

template <bool PRE>
__device__ __forceinline__ bool tri_hit_record(const float4 a, const float4 b, const float4 c, const float4 d,
                                               const Ray& r, float& limit, Hit& h)
{
    if (PRE) return tri_hit_pre(v4(a), v4(b), v4(c), v4(d), r, limit, h);
    return tri_hit(v4(a), v4(b), v4(c), v4(d), r, limit, h);
}

// a / b correctly rounded, as the compiler's expansion computes it (reciprocal, one Newton step, quotient, two residual
// corrections) minus its range scaling and special-case fix-up (v_div_scale x 2, v_div_fixup): equal to the IEEE quotient
// whenever a, b and a / b are normal numbers.  The triangle test only USES the quotient in that case: a denominator below
// 1e-5 in magnitude rejects the triangle (FullKernel.cl:533), and a quotient too small to be normal puts the hit within
// 1e-5 of the origin, which rejects it too (:543) - whatever bits the division produced.
__device__ __forceinline__ float div_unscaled(float a, float b)
{
    float r = __builtin_amdgcn_rcpf(b);
    r = __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r);
    float q = a * r;
    q = __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
    return __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
}

// The same test shaped for the wavefront kernel's instruction budget (it is VALU-issue bound): two nesting levels
// instead of five, and the accepted hit is consumed by `on_accept(q, ray parameter, s, t, front, nsd)` INSIDE the innermost
// block, so no value has to be merged back through the early exits (each merge level cost a v_mov per live value).
// The rejections are the reference's (FullKernel.cl:519-589) with unchanged operands; only their order differs -
// the behind-the-ray test (:566) moves up next to the distance tests - which cannot change the outcome because the
// function has no side effects before it accepts.
// (Tried in round 2 and dropped, both slower although they issue fewer arithmetic instructions - the compiler's pairing
// of the dot products into packed fma breaks and moves / spills come back: 3-component dot products for the five
// barycentric dots of a precomputed record, whose edge vectors have w = +0 (778 -> 772 Msamples/s); a second copy of the
// traversal loop for waves whose rays all have direction.w = 0, which drops six more instructions per test (778 -> 763).)
template <bool PRE, class LateQuads, class OnAccept>
__device__ __forceinline__ void tri_test(const float4 e0, const float4 e1, LateQuads&& late_quads, const Ray& r,
                                         const float limit, OnAccept&& on_accept)
{
    // Only two of the record's four quads are needed to reject a triangle at the distance tests (most tests end there):
    //   DTriPre: quad 0 = n, quad 1 = s1 + d          DTri: quad 0 = s1, quad 3 = n
    // the other two (u + 1/det, v + S1.w  /  s2, s3) are fetched by `late_quads` inside the block that needs them -
    // the line is in the L1 by then - which saves a third of the L1 accesses of a triangle step.
    const V4 N = PRE ? v4(e0) : v4(e1);
    const float d = PRE ? e1.w : dot(v4(e1), v4(e0));
    const float nd = dot(N, r.d);
    // (d - N.o) / nd.  strict: measured +0.6 % over the compiler's expansion; default: the reference's own 2.5-ulp form
    const float ray_t = kDefaultArithmetic ? fdiv(d - dot(N, r.o), nd) : div_unscaled(d - dot(N, r.o), nd);
    const V4 q = mad(r.d, ray_t, r.o);  // cl:538
    const V4 full = q - r.o;
    const float nsd = dot(full, full);
    const float fd = dot(full, r.d);
    const bool reject = ((nd > -0.00001f) & (nd < 0.00001f)) | (nsd > limit) | (nsd < 0.00001f) | (fd < 0);
    if (!reject) {
        float4 l0, l1;
        late_quads(l0, l1);
        V4 S1, u, v;
        if (PRE) {
            S1 = v4(e1.x, e1.y, e1.z, l1.w);
            u = v4(l0.x, l0.y, l0.z, 0.0f);
            v = v4(l1.x, l1.y, l1.z, 0.0f);
        } else {
            S1 = v4(e0);
            u = v4(l0) - S1;
            v = v4(l1) - S1;
        }
        const V4 w = q - S1;
        const float uv = dot(u, v), wv = dot(w, v), wu = dot(w, u), uu = dot(u, u), vv = dot(v, v);
        const float denom = PRE ? l0.w : frcp(mad(uv, uv, -(uu * vv)));  // cl:556
        const float s = mad(uv, wv, -(vv * wu)) * denom;                   // cl:558
        const float t = mad(uv, wu, -(uu * wv)) * denom;                   // cl:559
        if (!((s < 0) | (t < 0) | (s + t > 1))) on_accept(q, ray_t, s, t, nd < 0, nsd);
    }
}

__device__ __forceinline__ bool tri_hit(const DTri* __restrict__ tp, const Ray& r, float& limit, Hit& h)
{
    const float4* q4 = reinterpret_cast<const float4*>(tp);
    return tri_hit(v4(q4[0]), v4(q4[1]), v4(q4[2]), v4(q4[3]), r, limit, h);
}

// Texture_GetPixelColorValue, header.cl:430-459
__device__ __forceinline__ V4 texture_pixel(const ptmi_texture& tex, const ptmi_uchar4* __restrict__ texels, float u, float v)
{
    u = u - (float)((int)u) + (float)(u < 0 ? 1 : 0);
    v = v - (float)((int)v) + (float)(v < 0 ? 1 : 0);
    const uint32_t x = (uint32_t)(u * (float)(tex.width - 1u));
    const uint32_t y = (uint32_t)(v * (float)(tex.height - 1u));
    const ptmi_uchar4 p = texels[tex.offset + y * tex.width + x];
    constexpr DivC k255 = make_divc(255.f);
    V4 c = V4{fdiv((float)p.x, k255), fdiv((float)p.y, k255), fdiv((float)p.z, k255), fdiv((float)p.w, k255)};
    c.w = 1.f - c.w;
    return c;
}

// Sky_GetColorValue + Sky_GetFaceColorValue, FullKernel.cl:438-512
__device__ __forceinline__ V4 sky_color(const ptmi_sky& sky, const ptmi_uchar4* __restrict__ texels, V4 d)
{
    const float x = mad(sky.cos_rotation_angle, d.x, -(sky.sin_rotation_angle * d.y));  // cl:441
    const float y = mad(sky.sin_rotation_angle, d.x, sky.cos_rotation_angle * d.y);     // cl:442
    const float z = d.z;
    const float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
    int face = 0;
    float u = 0, v = 0;
    if (az > ax && az > ay) {
        if (z > 0) { face = 5; u = (1 - fdiv(x, z)) / 2; v = (1 + fdiv(y, z)) / 2; }
        else       { face = 0; u = (1 + fdiv(x, z)) / 2; v = (1 + fdiv(y, z)) / 2; }
    } else if (ax > ay && ax > az) {
        if (x > 0) { face = 1; u = (1 - fdiv(y, x)) / 2; v = (1 + fdiv(z, x)) / 2; }
        else       { face = 3; u = (1 - fdiv(y, x)) / 2; v = (1 - fdiv(z, x)) / 2; }
    } else if (ay > ax && ay > az) {
        if (y > 0) { face = 4; u = (1 + fdiv(x, y)) / 2; v = (1 + fdiv(z, y)) / 2; }
        else       { face = 2; u = (1 + fdiv(x, y)) / 2; v = (1 - fdiv(z, y)) / 2; }
    }
    return texture_pixel(sky.sky_textures[face], texels, u, v);
}

// Fresnel core shared by FullKernel.cl:192-217 (glass), :219-254 (water), :256-292 (varnish)
// `n2` is a run-time value in the water function (selected by isInWater, cl:223-232) and a literal in the glass and varnish
// functions: the default arithmetic divides differently by the two (see fdiv), hence the divisor type.
// `total` (optional): set when the function returns on total reflection, i.e. BEFORE the reference writes its refraction outputs
// (cl:237 against :249-251) - see scatter_direction.
template <class Divisor>
__device__ __forceinline__ float fresnel_fraction(float n1, float n2, const Divisor& by_n2, float cos1, V4 incident, V4 N, V4* refr,
                                                  bool* total = nullptr)
{
    const float sin1 = fsqrt(mad(-cos1, cos1, 1));  // cl:202,235,275 `1 - cos1 * cos1`
    const float sin2 = fdiv(n1 * sin1, by_n2);
    if (sin2 >= 1) {
        if (total) *total = true;
        return 1;
    }
    const float cos2 = fsqrt(mad(-sin2, sin2, 1));
    const float r_para = fdiv(mad(n2, cos1, -(n1 * cos2)), mad(n2, cos1, n1 * cos2));  // cl:208
    const float r_perp = fdiv(mad(n1, cos1, -(n2 * cos2)), mad(n1, cos1, n2 * cos2));  // cl:209
    if (refr) {  // cl:249 (water only: a run-time n1 / n2)
        const float ratio = fdiv(n1, n2);
        *refr = mad(incident, ratio, N * mad(ratio, cos1, -cos2));
    }
    return mad(r_para, r_para, r_perp * r_perp) / 2.0f;  // cl:211
}

__device__ __forceinline__ float fresnel_varnish(V4 incident, V4 N)
{
    const float cos1 = fmaxf(0.f, fminf(1.f, -dot(incident, N)));
    constexpr DivC by_n2 = make_divc(kNVarnish);
    return fresnel_fraction(1.0f, kNVarnish, by_n2, cos1, incident, N, nullptr);
}

// Material_FresnelReflection, FullKernel.cl:294-300
__device__ __forceinline__ V4 reflect_about(V4 v, V4 N) { return mad(-N, 2 * dot(v, N), v); }  // cl:298

// Material_BRDF, FullKernel.cl:166-190
__device__ __forceinline__ float material_brdf(int type, V4 incident, V4 N, V4 reflected)
{
    if (type == PTMI_MAT_STANDART) return kPiInverse;
    if (type == PTMI_MAT_GLASS) return 1;
    if (type == PTMI_MAT_WATER) {
        const float denom = mad(kSchlick, dot(incident, reflected), 1);  // cl:177
        return fdiv(1 - kSchlick * kSchlick, 4 * kPi * denom * denom);
    }
    if (type == PTMI_MAT_VARNHISHED) return (1 - fresnel_varnish(incident, N)) * kPiInverse;
    return 1;
}

// Material_ConcentricSampleDisk (FullKernel.cl:339-416) + Material_CosineSampleHemisphere (:303-336)
__device__ __forceinline__ V4 cosine_sample_hemisphere(int& seed, V4 N)
{
    const float u1 = lcg_random(seed);
    const float u2 = lcg_random(seed);
    const float sx = mad(2, u1, -1);  // cl:350
    const float sy = mad(2, u2, -1);
    // The reference's four octant pairs (:356-397), each with a division of its own:
    //   sx > -sy:  sx > sy ? (r = sx, theta = sy > 0 ? sy/sx : 8 + sy/sx) : (r = sy, theta = 2 - sx/sy)
    //   else:      sx < sy ? (r = -sx, theta = 4 + sy/sx)                 : (r = -sy, theta = 6 - sy/sx)
    // written with ONE division whose operands are selected (the lanes of a wave fall into all four branches, so the
    // branchy form executes every copy of the ~10-instruction division): same operations per lane, same bits.
    const bool upper = sx > -sy;
    const bool pair_a = upper & (sx > sy), pair_b = upper & !(sx > sy), pair_c = !upper & (sx < sy);
    const float q = fdiv(pair_b ? sx : sy, pair_b ? sy : sx);
    float r = pair_a ? sx : (pair_b ? sy : (pair_c ? -sx : -sy));
    float theta = pair_a ? (sy > 0 ? q : 8.f + q) : (pair_b ? 2.f - q : (pair_c ? 4.f + q : 6.f - q));
    if (fabsf(sy) < 0.0001f) { r = sx; theta = 2; }
    if (fabsf(sx) < 0.0001f) { r = sy; theta = 0; }
    theta *= kPi / 4.f;
    r = (float)((double)r * 0.999);  // FullKernel.cl:408: the literal is a double
    float sn, cs;
    ptmi_sincosf(theta, &sn, &cs);
    const float x = r * cs, y = r * sn;
    float z = mad(-y, y, mad(-x, x, 1));  // cl:311 `1 - x*x - y*y`
    // This square root is CORRECTLY ROUNDED in both arithmetics: `z = (z<0) ? 0 : sqrt(z)` (cl:312) becomes a select, and the
    // call the optimizer speculates for it loses the !fpmath annotation that lets every other sqrt of the default build be
    // v_sqrt_f32 (the optimized IR of the reference kernel shows llvm.sqrt without it exactly here).
    z = (z < 0) ? 0 : sqrtf(z);
    const V4 v = v4(x, y, z, 0);
    if (N.z > 0.9999f) return v;
    if (N.z < -0.9999f) return -v;
    const V4 sn4 = normalize(v4(-N.y, N.x, 0, 0));
    const V4 tn4 = normalize(cross(N, sn4));
    return normalize(v4(dot(v4(sn4.x, tn4.x, N.x, 0), v), dot(v4(sn4.y, tn4.y, N.y, 0), v),
                        dot(v4(sn4.z, tn4.z, N.z, 0), v), 0));
}

// Light_PowerToward, header.cl:403-421
__device__ __forceinline__ float light_power_toward(const ptmi_light& l, V4 p, V4 N)
{
    const V4 pos = v4(l.position), dir = v4(l.direction);
    if (l.type == PTMI_LIGHT_DIRECTIONNAL) return l.power * fmaxf(dot(-dir, N), 0.f);
    if (l.type == PTMI_LIGHT_POINT) {
        const V4 d = p - pos;
        return fdiv(l.power, dot(d, d)) * fmaxf(dot(normalize(pos - p), N), 0.f);
    }
    if (l.type == PTMI_LIGHT_SPOT) {
        const V4 lrd = normalize(p - pos);
        const float cos_angle = dot(lrd, dir);
        if (cos_angle > l.cos_inner) return l.power * fmaxf(-dot(lrd, N), 0.f);
        if (cos_angle < l.cos_outer) return 0.0f;
        return fdiv(l.power * (cos_angle - l.cos_outer), l.cos_inner - l.cos_outer) * fmaxf(-dot(lrd, N), 0.f);
    }
    return 0.f;
}

}  // namespace PTMI_DEV_NS
