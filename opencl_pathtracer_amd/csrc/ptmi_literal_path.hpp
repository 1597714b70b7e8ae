// ptmi_literal_path.hpp - one path of the reference, its loops as they are written (Kernel_Main, FullKernel.cl:1180-1331):
// the body of the one-path-per-lane kernel (kernels.hip), and what the wavefront kernel's launches fall back on for the rare
// path whose closest-hit query accepts a NaN distance (kernel_wavefront.hip: redo_poisoned_kernel).
#pragma once

#include "ptmi_device.hpp"
#include "ptmi_shading.hpp"

namespace PTMI_DEV_NS {

constexpr int kBlock = 256;
constexpr int kStackDepth = PTMI_BVH_MAX_DEPTH;

struct PathCounters {
    uint32_t bbx, tri;  // numIntersectedBBx / numIntersectedTri of the current path
};

// BVH_IntersectRay (FullKernel.cl:620-702) when ANY_HIT == false,
// BVH_IntersectShadowRay (:705-783) when true.  Same visit order as the
// reference: at an inner node the child on the side the ray comes from
// (dir[cutAxis] > 0 ? son1 : son2) is tested first and descended first, the
// other is pushed; leaf triangles in ascending index with the distance limit
// updated between tests.
template <bool ANY_HIT, bool PRE>
__device__ __forceinline__ bool traverse(const DScene& sc, const Ray& r, float limit, Hit& hit, PathCounters& pc,
                                         uint32_t* __restrict__ stack)
{
    // A closest-hit query of a ray whose direction is NaN in every component (the scattered ray of a hit on a zero-area
    // triangle, whose normal is 0/0): every comparison of the box test and of the triangle test is false, so every non-empty
    // box is "hit" and every triangle accepted - the query walks the WHOLE tree in one fixed order and keeps its last triangle
    // (seconds for one path on a million triangles, ten times per path).  The upload has walked it once
    // (scene_layout.cpp: nan_walk_*): the counts are added and the LAST triangle's test is made for real, which leaves the
    // very record, bit for bit, that the walk would leave.
    if (!ANY_HIT && sc.nan_walk_box_tests != 0xFFFFFFFFu && (r.d.x != r.d.x) & (r.d.y != r.d.y) & (r.d.z != r.d.z)) {
        pc.bbx += sc.nan_walk_box_tests;
        pc.tri += sc.nan_walk_tri_tests;
        if (sc.nan_walk_last_tri == 0xFFFFFFFFu) return false;
        const float4* q4 = reinterpret_cast<const float4*>(&sc.tris[sc.nan_walk_last_tri]);
        const bool accepted = tri_hit_record<PRE>(q4[0], q4[1], q4[2], q4[3], r, limit, hit);
        hit.tri = sc.nan_walk_last_tri;
        return accepted;
    }
    bool found = false;
    int top = 0;
    uint32_t cur = sc.root_ref;
    for (;;) {
        if (cur & REF_LEAF) {
            uint32_t count = (cur >> REF_COUNT_SHIFT) & 7u;
            uint32_t start = cur & REF_INDEX_MASK_LEAF;
            if (count == REF_COUNT_BIG) {
                const DBigLeaf bl = sc.big_leaves[start];
                start = bl.start;
                count = bl.count;
            }
            for (uint32_t i = start; i < start + count; i++) {
                pc.tri++;
                const float4* q4 = reinterpret_cast<const float4*>(&sc.tris[i]);
                if (tri_hit_record<PRE>(q4[0], q4[1], q4[2], q4[3], r, limit, hit)) {
                    if (ANY_HIT) return true;
                    hit.tri = i;
                    found = true;
                }
            }
            if (top == 0) break;
            cur = stack[(--top) * kBlock];
        } else {
            const float4* np = reinterpret_cast<const float4*>(&sc.nodes[cur & REF_INDEX_MASK_INNER]);
            const float4 a = np[0], b = np[1], c = np[2], d = np[3];
            const float lo1[3] = {a.x, a.y, a.z}, hi1[3] = {a.w, b.x, b.y};
            const float lo2[3] = {b.z, b.w, c.x}, hi2[3] = {c.y, c.z, c.w};
            const uint32_t ref1 = __float_as_uint(d.x), ref2 = __float_as_uint(d.y), axis = __float_as_uint(d.z);
            const float da = axis == 0 ? r.d.x : (axis == 1 ? r.d.y : r.d.z);
            const bool fwd = da > 0;
            const bool h1 = box_hit(lo1, hi1, (ref1 & REF_EMPTY) != 0, r, limit);
            const bool h2 = box_hit(lo2, hi2, (ref2 & REF_EMPTY) != 0, r, limit);
            pc.bbx += 2;
            const uint32_t near_ref = fwd ? ref1 : ref2, far_ref = fwd ? ref2 : ref1;
            const bool near_hit = fwd ? h1 : h2, far_hit = fwd ? h2 : h1;
            if (near_hit) {
                if (far_hit) stack[(top++) * kBlock] = far_ref;
                cur = near_ref;
            } else if (far_hit) {
                cur = far_ref;
            } else {
                if (top == 0) break;
                cur = stack[(--top) * kBlock];
            }
        }
    }
    return found;
}

// One path = one Kernel_Main work-item (FullKernel.cl:1180-1331) up to the
// statistics; returns the radiance and the sample position.
// `stop_criterion_draw`: SUPER_SAMPLING draws one random number for its stop criterion before the path starts (cl:1219-1222,
// from iteration 6 on); a caller that re-traces a path of such a render passes true so that the path's numbers are the same.
template <bool PRE>
__device__ __forceinline__ V4 trace_path(const DScene& sc, uint32_t gx, uint32_t gy, uint32_t iteration,
                                         uint32_t* __restrict__ stack, float& sample_x, float& sample_y,
                                         uint32_t& depth, uint32_t& segments, uint32_t& shadows, PathCounters& pc,
                                         bool stop_criterion_draw = false)
{
    int seed = lcg_seed(gx, gy, sc.width, sc.height, iteration, sc.source_seed != 0);
    draw_sample(sc, gx, gy, iteration, seed, sample_x, sample_y);
    if (stop_criterion_draw) (void)lcg_random(seed);

    Ray r;
    r.o = v4(sc.cam_pos);
    ray_set_direction(r, mad(v4(sc.cam_up), sample_y, mad(v4(sc.cam_right), sample_x, v4(sc.cam_dir))));  // cl:1213

    V4 radiance = v4(0, 0, 0, 0), transfer = v4(1, 1, 1, 1);
    bool active = true, in_water = false;
    uint32_t reflection = 0;
    pc.bbx = 0;
    pc.tri = 0;

    while (active && reflection < sc.max_depth) {
        Hit hit;
        hit.tri = 0; hit.s = 0; hit.t = 0; hit.front = false; hit.point = v4(0, 0, 0, 0);
        segments++;
        if (traverse<false, PRE>(sc, r, INFINITY, hit, pc, stack)) {
            Surface sf;
            load_surface(sc, r, hit, sf);

            // Scene_ComputeDirectIllumination, FullKernel.cl:901-954: every light, every bounce
            V4 direct = v4(0, 0, 0, 0);
            for (uint32_t li = 0; li < sc.n_lights; li++) {
                const ptmi_light light = sc.lights[li];
                const bool directional = light.type == PTMI_LIGHT_DIRECTIONNAL;
                const V4 full = directional ? -v4(light.direction) : v4(light.position) - hit.point;
                Ray lr;
                lr.o = hit.point;
                ray_set_direction(lr, full);
                const float light_distance = directional ? INFINITY : length(full);  // linear, :938
                const float brdf = material_brdf(sf.mat.type, -lr.d, sf.Ns, r.d);
                Hit dummy;
                shadows++;
                if (!traverse<true, PRE>(sc, lr, light_distance, dummy, pc, stack))
                    direct = mad(v4(1, 1, 1, 1) * (light_power_toward(light, hit.point, sf.Ns) * brdf), v4(light.color), direct);  // cl:945
            }

            radiance = radiance + scatter(r, seed, in_water, hit, sf, direct, transfer);
            reflection++;
        } else {
            active = false;
            radiance = mad(sky_color(sc.sky, sc.texels, r.d), transfer, radiance);  // :1281-1288
        }
        if (active) active = path_continues(transfer, reflection, seed, sc.russian_roulette != 0);  // :1296-1314
    }
    depth = reflection;
    return radiance;
}

}  // namespace PTMI_DEV_NS
