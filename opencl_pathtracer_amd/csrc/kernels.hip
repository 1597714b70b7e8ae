// kernels.hip - the integrator kernel for gfx950 (MI355X).
//
// One launch renders a RANGE of iterations (samples per pixel) for every pixel:
// a lane owns one pixel for the whole launch, runs its iterations back to back
// with the radiance sum kept in registers and touches the framebuffer once
// (the reference launches once per iteration and read-modify-writes 40 bytes per
// path, PathTracer_OpenCL.cpp:76-89 / FullKernel.cl:1339-1345).  A wavefront
// covers an 8x8 pixel tile so that camera rays of a wave walk the same nodes.
// The traversal stack (30 child references per lane, FullKernel.cl:627) lives in
// LDS, lane-interleaved so every push/pop is conflict free.
//
// Result contract: per pixel and iteration the SAME radiance bits as the
// reference algorithm evaluated with the numerics of ptmi_device.hpp, added to
// the accumulator in iteration order => images equal the CPU checker's bit for
// bit for the JITTERED/UNIFORM samplers.
#include <hip/hip_runtime.h>

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
template <bool PRE>
__device__ __forceinline__ V4 trace_path(const DScene& sc, uint32_t gx, uint32_t gy, uint32_t iteration,
                                         uint32_t* __restrict__ stack, float& sample_x, float& sample_y,
                                         uint32_t& depth, uint32_t& segments, uint32_t& shadows, PathCounters& pc)
{
    int seed = lcg_seed(gx, gy, sc.width, sc.height, iteration, sc.source_seed != 0);
    draw_sample(sc, gx, gy, iteration, seed, sample_x, sample_y);

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

template <bool PRE>
__global__ void __launch_bounds__(kBlock) render_kernel(const DScene sc, const uint32_t first_iteration,
                                                        const uint32_t n_iterations, const uint32_t iteration_stride)
{
    __shared__ uint32_t stack_mem[kStackDepth * kBlock];
    __shared__ unsigned long long block_counters[C_COUNT];

    const uint32_t tid = threadIdx.x;
    if (tid < C_COUNT) block_counters[tid] = 0;
    __syncthreads();

    // wave -> 8x8 pixel tile; block (4 waves) -> 16x16
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    const uint32_t gx = blockIdx.x * 16u + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t gy = blockIdx.y * 16u + (wave >> 1) * 8u + (lane >> 3);
    const bool valid = gx < sc.width && gy < sc.height;
    uint32_t* stack = &stack_mem[tid];

    if (valid) {
        unsigned long long n_bbx = 0, n_tri = 0;
        uint32_t n_seg = 0, n_shadow = 0, n_hits = 0;
        const bool owns_pixel = sc.sampler != PTMI_SAMPLER_RANDOM;
        const uint32_t own_offset = gy * sc.width + gx;
        V4 sum = v4(0, 0, 0, 0);
        float count = 0;
        if (owns_pixel) {
            sum = v4(*reinterpret_cast<const float4*>(&sc.image_color[4 * own_offset]));
            count = sc.image_ray_nb[own_offset];
        }
        for (uint32_t k = 0; k < n_iterations; k++) {
            const uint32_t it = first_iteration + k * iteration_stride;
            float sx, sy;
            uint32_t depth;
            PathCounters pc;
            const V4 radiance = trace_path<PRE>(sc, gx, gy, it, stack, sx, sy, depth, n_seg, n_shadow, pc);
            n_bbx += pc.bbx;
            n_tri += pc.tri;
            n_hits += depth;
            if (sc.hist_depths) {  // FullKernel.cl:1319-1331
                atomicAdd(&sc.hist_depths[depth], 1u);
                if (pc.bbx < PTMI_MAX_INTERSECTION_NUMBER) atomicAdd(&sc.hist_bbx[pc.bbx], 1u);
                if (pc.tri < PTMI_MAX_INTERSECTION_NUMBER) atomicAdd(&sc.hist_tri[pc.tri], 1u);
            }
            if (owns_pixel) {
                // JITTERED / UNIFORM: the sample always lands on the work-item's own pixel
                sum = sum + radiance;  // :1340
                count = count + 1.f;   // :1342
            } else {
                // RANDOM: samples land anywhere; the reference races here (:1339-1345), we add atomically
                const uint32_t off = sample_pixel(sc, sx, sy);
                atomicAdd(&sc.image_color[4 * off + 0], radiance.x);
                atomicAdd(&sc.image_color[4 * off + 1], radiance.y);
                atomicAdd(&sc.image_color[4 * off + 2], radiance.z);
                atomicAdd(&sc.image_color[4 * off + 3], radiance.w);
                atomicAdd(&sc.image_ray_nb[off], 1.f);
            }
        }
        if (owns_pixel) {
            *reinterpret_cast<float4*>(&sc.image_color[4 * own_offset]) = make_float4(sum.x, sum.y, sum.z, sum.w);
            sc.image_ray_nb[own_offset] = count;
        }
        atomicAdd(&block_counters[C_PATHS], (unsigned long long)n_iterations);
        atomicAdd(&block_counters[C_SEGMENTS], (unsigned long long)n_seg);
        atomicAdd(&block_counters[C_HITS], (unsigned long long)n_hits);
        atomicAdd(&block_counters[C_SHADOW], (unsigned long long)n_shadow);
        atomicAdd(&block_counters[C_BBX], n_bbx);
        atomicAdd(&block_counters[C_TRI], n_tri);
    }
    __syncthreads();
    if (tid <= C_TRI) atomicAdd(&sc.counters[tid], block_counters[tid]);
}

}  // namespace PTMI_DEV_NS

namespace ptmi_internal {

int PTMI_ARITH(launch_render)(const DScene& sc, uint32_t first_iteration, uint32_t n_iterations, uint32_t iteration_stride, void* stream,
                  std::string* err)
{
    if (n_iterations == 0) return PTMI_OK;
    const dim3 grid((sc.width + 15u) / 16u, (sc.height + 15u) / 16u);
    if (sc.tris_precomputed)
        hipLaunchKernelGGL(PTMI_DEV_NS::render_kernel<true>, grid, dim3(PTMI_DEV_NS::kBlock), 0, (hipStream_t)stream, sc,
                           first_iteration, n_iterations, iteration_stride);
    else
        hipLaunchKernelGGL(PTMI_DEV_NS::render_kernel<false>, grid, dim3(PTMI_DEV_NS::kBlock), 0, (hipStream_t)stream, sc,
                           first_iteration, n_iterations, iteration_stride);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (err) *err = std::string("render_kernel launch: ") + hipGetErrorString(e);
        return PTMI_ERR_HIP;
    }
    return PTMI_OK;
}

}  // namespace ptmi_internal
