// ptmi_shading.hpp - what happens at a surface hit / path end; shared by both integrator kernels.
#pragma once

#include "ptmi_device.hpp"

namespace PTMI_DEV_NS {

struct Surface {
    V4 Ng, Ns, color;
    DMat mat;
};

// What Kernel_Main does between BVH_IntersectRay and the light loop
// (FullKernel.cl:1254-1274) plus the deferred Triangle_GetColorValueAt (:591-602).
// PLAIN: the scene holds only plain-colour MAT_STANDART materials (checked at upload): no texture fetch, and the type is a
// constant for everything inlined behind this call.
template <bool PLAIN = false, class SceneT>
__device__ __forceinline__ void load_surface(const SceneT& sc, const Ray& r, const Hit& hit, Surface& sf)
{
    const DShade* sh = &sc.shade[sc.tri_ids[hit.tri]];  // hit.tri is a record index
    const float4* s4 = reinterpret_cast<const float4*>(sh);
    // geometric normal: first float4 of a DTriPre record, last of a DTri record
    const V4 N = v4(reinterpret_cast<const float4*>(&sc.tris[hit.tri])[sc.tris_precomputed ? 0 : 3]);
    const V4 N1 = v4(s4[0]), N2 = v4(s4[1]), N3 = v4(s4[2]);
    const uint32_t mat_id = hit.front ? sh->mat_pos : sh->mat_neg;
    sf.mat = sc.mats[mat_id];
    if (PLAIN) sf.mat.type = PTMI_MAT_STANDART;

    const float b = (1 - hit.s) - hit.t;
    if (PLAIN || sf.mat.is_simple_color) {
        sf.color = v4(sf.mat.color);
    } else {
        const float* uv = hit.front ? sh->uvp : sh->uvn;
        const float tu = mad(uv[4], hit.t, mad(uv[0], b, uv[2] * hit.s));  // cl:600 (UV1*(1-s-t)) + (UV2*s) + (UV3*t)
        const float tv = mad(uv[5], hit.t, mad(uv[1], b, uv[3] * hit.s));
        sf.color = texture_pixel(sc.textures[sf.mat.texture_id], sc.texels, tu, tv);
    }

    const bool same_dir = dot(r.d, N) > 0;                       // :1267
    sf.Ng = same_dir ? -N : N;                                   // Triangle_GetNormal, header.cl:500
    V4 Ns = normalize(mad(N1, b, mad(N2, hit.s, N3 * hit.t)));   // Triangle_GetSmoothNormal, :604-610
    if (same_dir) Ns = -Ns;
    Ns = put_in_same_hemisphere(Ns, -r.d);                       // :1273
    sf.Ns = normalize(Ns);                                       // :1274
}

// Scene_ComputeRadiance, FullKernel.cl:791-891: updates transfer, the ray and
// isInWater; returns the radiance gathered at this bounce.
// (split in two so that a caller with several sources of new rays can share ONE copy of the ray set-up, which holds
// four divisions: scatter_direction = everything up to the outgoing direction `out`, un-normalised as :880 uses it)
// `undefined_in_reference` (statistics builds): set when the bounce is one the reference's SOURCE leaves undefined - the water
// material refracting a totally reflected ray, which needs random() == 1.0 exactly: Material_FresnelWaterReflectionFraction has
// then returned before writing the refraction direction and factor that cl:836-843 go on to read.  Here: a zero direction (which
// Vector_PutInSameHemisphereAs turns into 0.01 * N) and the factor n2^2 / n1^2.
__device__ __forceinline__ V4 scatter_direction(const Ray& r, int& seed, bool& in_water, const Surface& sf, V4 direct,
                                                V4& transfer, V4& out_direction, V4* hemisphere_normal = nullptr,
                                                bool* undefined_in_reference = nullptr)
{
    V4 N = r.d;
    V4 radiance = v4(0, 0, 0, 0);
    V4 out = r.d;
    const int type = sf.mat.type;
    bool diffuse = false;  // the outgoing direction is a cosine sample about Ns (drawn once, below, for both users)
    if (type == PTMI_MAT_STANDART) {
        transfer = transfer * sf.color;
        radiance = direct * transfer;
        diffuse = true;
        N = sf.Ns;
    } else if (type == PTMI_MAT_GLASS) {
        constexpr DivC by_n_glass = make_divc(kNGlass);
        const float f = fresnel_fraction(1, kNGlass, by_n_glass, -dot(r.d, sf.Ns), r.d, sf.Ns, nullptr);
        if (lcg_random(seed) < f) {
            out = reflect_about(r.d, sf.Ns);
            N = sf.Ng;
        } else {
            transfer = transfer * (sf.color * (1 - sf.mat.opacity));
            N = r.d;
        }
    } else if (type == PTMI_MAT_WATER) {
        V4 refracted = v4(0, 0, 0, 0);
        const float n1 = in_water ? kNWater : 1.f, n2 = in_water ? 1.f : kNWater;
        bool total = false;
        const float f = fresnel_fraction(n1, n2, n2, -dot(r.d, sf.Ns), r.d, sf.Ns, &refracted, undefined_in_reference ? &total : nullptr);
        if (lcg_random(seed) < f) {
            out = reflect_about(r.d, sf.Ns);
            N = sf.Ng;
        } else {
            if (undefined_in_reference && total) *undefined_in_reference = true;
            in_water = !in_water;
            out = refracted;
            N = -sf.Ng;
            transfer = transfer * fdiv(n2 * n2, n1 * n1);  // cl:251
        }
    } else if (type == PTMI_MAT_VARNHISHED) {
        radiance = mad(direct * sf.color, transfer, radiance);  // cl:849
        const float f1 = fresnel_varnish(r.d, sf.Ns);
        if (lcg_random(seed) < f1) {
            out = reflect_about(r.d, sf.Ns);
        } else {
            diffuse = true;  // (the Fresnel draw above comes first, as in cl:858-870)
            transfer = transfer * sf.color;
        }
    }
    if (diffuse) out = cosine_sample_hemisphere(seed, sf.Ns);
    out_direction = put_in_same_hemisphere(out, N);
    if (hemisphere_normal) *hemisphere_normal = N;  // (for the statistics build's check of header.cl:243)
    return radiance;
}

__device__ __forceinline__ V4 scatter(Ray& r, int& seed, bool& in_water, const Hit& hit, const Surface& sf, V4 direct,
                                      V4& transfer)
{
    V4 out;
    const V4 radiance = scatter_direction(r, seed, in_water, sf, direct, transfer, out);
    ray_set_direction(r, out);
    r.o = mad(out, 0.001f, hit.point);  // :880 uses the un-normalised direction
    return radiance;
}

// The end-of-bounce tests of Kernel_Main's loop (FullKernel.cl:1296-1314): false = the path ends here.  `russian_roulette`
// enables the block the reference ships commented out (:1306-1314), evaluated as it is written there.
__device__ __forceinline__ bool path_continues(V4& transfer, uint32_t reflection, int& seed, bool russian_roulette)
{
    const float m_yz = transfer.y < transfer.z ? transfer.z : transfer.y;  // OpenCL max(x, y) = x < y ? y : x
    const float m = transfer.x < m_yz ? m_yz : transfer.x;
    if (m <= kMinContribution) return false;
    bool active = true;
    if (russian_roulette && reflection > 5u) {  // MIN_REFLECTION_NUMBER, header.cl:13
        const float coeff = fdiv(m, (float)(reflection - 5u));
        if (coeff < 1) {
            active = lcg_random(seed) > coeff;
            transfer = V4{fdiv(transfer.x, coeff), fdiv(transfer.y, coeff), fdiv(transfer.z, coeff), fdiv(transfer.w, coeff)};
        }
    }
    return active;
}

// sampler(), FullKernel.cl:1119-1150
template <class SceneT>
__device__ __forceinline__ void draw_sample(const SceneT& sc, uint32_t gx, uint32_t gy, uint32_t iteration, int& seed,
                                            float& sx, float& sy)
{
    // IMAGE_WIDTH / IMAGE_HEIGHT are constants of the reference's program (-D, OpenCL.cpp:296-299): constant divisors
    const DivC by_w = make_divc((float)sc.width), by_h = make_divc((float)sc.height);
    if (sc.sampler == PTMI_SAMPLER_UNIFORM) {
        constexpr DivC by_3 = make_divc(3.f);
        const int sample_id = (int)(iteration % 9u);
        sx = (float)gx; sy = (float)gy;
        float ox = (float)(sample_id % 3), oy = (float)(sample_id / 3);
        ox += 0.5f; oy += 0.5f;
        ox = fdiv(ox, by_3); oy = fdiv(oy, by_3);
        sx += ox; sy += oy;
        sx = fdiv(sx, by_w); sy = fdiv(sy, by_h);
        sx -= 0.5f; sy -= 0.5f;
    } else if (sc.sampler == PTMI_SAMPLER_RANDOM) {
        sx = lcg_random(seed);
        sy = lcg_random(seed);
        sx *= 0.9f; sy *= 0.9f;
        sx += 0.05f; sy += 0.05f;
        sx -= 0.5f; sy -= 0.5f;
    } else {
        sx = fdiv(mad(0.9f, lcg_random(seed), (float)gx) + 0.05f, by_w) - 0.5f;  // cl:1145
        sy = fdiv(mad(0.9f, lcg_random(seed), (float)gy) + 0.05f, by_h) - 0.5f;
    }
}

// pixel a sample lands on, FullKernel.cl:1333-1336 (double arithmetic)
template <class SceneT>
__device__ __forceinline__ uint32_t sample_pixel(const SceneT& sc, float sx, float sy)
{
    int px = (int)(((double)sx + 0.5) * (int)sc.width);
    int py = (int)(((double)sy + 0.5) * (int)sc.height);
    px = min(px, (int)sc.width - 1);
    py = min(py, (int)sc.height - 1);
    return (uint32_t)py * sc.width + (uint32_t)px;
}

}  // namespace PTMI_DEV_NS
