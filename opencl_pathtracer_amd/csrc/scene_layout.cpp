// scene_layout.cpp - validation and re-layout of a scene for the device (see scene_layout.h).  No device call in this file.
#include "scene_layout.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace ptmi_internal {

namespace {

int fail(std::string& err, int code, const std::string& msg)
{
    err = msg;
    return code;
}

// Scene validation + re-layout.  Everything the kernel will index is checked
// here so that a malformed scene is an error code, not a GPU fault.


bool texture_ok(const ptmi_texture& t, uint32_t data_size)
{
    if (t.width == 0 || t.height == 0) return false;
    const uint64_t end = (uint64_t)t.offset + (uint64_t)t.width * t.height;
    return end <= data_size;
}

// Can the RECORDS of this scene make a triangle test compute a NaN distance?  The reference's test rejects with comparisons
// only (FullKernel.cl:533-567), so a triangle on which it computes NaNs - a zero-area triangle as the importer emits it: N =
// normalize(0) = 0/0 (Utils.h:144) - is ACCEPTED by every ray that reaches it, with a NaN squared distance: from then on
// nothing is "too far" (:543, :92) and the LAST triangle that passes the remaining tests wins, whatever its distance.  The
// one-path-per-lane kernel runs the reference's loops literally and reproduces this bit for bit.  The wavefront kernel's leaf
// passes keep the minimum of (distance, order) keys - the sequential loop's result only while distances are ordered, i.e.
// numbers.  Two sources of NaN distances, two remedies:
//   * the RAY is not a number (a refraction at |cos| = 1 + 1 ulp, cl:235; an origin that overflowed): the wavefront kernel
//     checks every ray it sets up, gives such a path up and the literal loops trace it again into its staging slot
//     (kernel_wavefront.hip: redo_poisoned_kernel) - costs nothing in the traversal loop;
//   * the RECORDS are: then the scene runs the wavefront kernel's NANSAFE instantiation (round 4; before: the one-path-per-lane
//     kernel for the whole scene) - a test per accepted triangle in the leaf pass, which hands only the paths that REACH such a
//     record to the literal loops (0.9 % when every scene pays it: so only these scenes do).  This function finds those scenes;
//     the one-path-per-lane kernel still renders them as a whole with the RANDOM sampler (nothing staged to patch).
// With finite rays from origins below 2^40 a distance is a number when
//   * every triangle's vertices, normals and vertex normals, the lights' positions and directions and the camera are finite
//     and at most 2^21 (normals: 16) in magnitude: then the plane terms are finite, the ray parameter is (|N.d| >= 1e-5 or
//     the triangle is rejected, cl:533), and so are the hit point and its distance (an overflow gives +inf: ordered), and
//   * the barycentric determinant uv^2 - uu.vv of every triangle is non-zero with a finite reciprocal, in both arithmetics
//     (s and t may still overflow for a hit on a corrupted record's plane far outside its triangle; they only gate the
//     acceptance, they do not enter the order).
// Returns the reason, or an empty string.
std::string scene_needs_literal_kernel(const ptmi_scene* sc)
{
    constexpr float kCoord = 2097152.0f, kNormal = 16.0f;
    auto ok = [](const ptmi_float4& v, float bound) {
        return std::fabs(v.x) <= bound && std::fabs(v.y) <= bound && std::fabs(v.z) <= bound && std::fabs(v.w) <= bound;  // (false for NaN)
    };
    if (!ok(sc->camera_position, kCoord) || !ok(sc->camera_direction, kCoord) || !ok(sc->camera_right, kCoord) || !ok(sc->camera_up, kCoord))
        return "the camera is not finite (or beyond 2^21)";
    for (uint32_t i = 0; i < sc->lights_size; i++)
        if (!ok(sc->lights[i].position, kCoord) || !ok(sc->lights[i].direction, kCoord))
            return "light " + std::to_string(i) + " is not finite (or beyond 2^21)";
    auto dot4 = [](const float a[4], const float b[4]) {
        return std::fmaf(a[3], b[3], std::fmaf(a[2], b[2], std::fmaf(a[1], b[1], a[0] * b[0])));
    };
    for (uint32_t i = 0; i < sc->triangulation_size; i++) {
        const ptmi_triangle& t = sc->triangulation[i];
        if (!ok(t.s1, kCoord) || !ok(t.s2, kCoord) || !ok(t.s3, kCoord))
            return "triangle " + std::to_string(i) + " has a vertex that is not finite (or beyond 2^21)";
        if (!ok(t.n, kNormal) || !ok(t.n1, kNormal) || !ok(t.n2, kNormal) || !ok(t.n3, kNormal))
            return "triangle " + std::to_string(i) + " has a normal that is not finite (a zero-area triangle of the importer: N = 0/0)";
        const float u[4] = {t.s2.x - t.s1.x, t.s2.y - t.s1.y, t.s2.z - t.s1.z, t.s2.w - t.s1.w};
        const float v[4] = {t.s3.x - t.s1.x, t.s3.y - t.s1.y, t.s3.z - t.s1.z, t.s3.w - t.s1.w};
        const float uv = dot4(u, v), uu = dot4(u, u), vv = dot4(v, v);
        const float det[2] = {uv * uv - uu * vv, std::fmaf(uv, uv, -(uu * vv))};  // cl:556, strict and default arithmetic
        for (float d : det)
            if (!(d != 0.0f) || !std::isfinite(d) || !std::isfinite(1.0f / d))
                return "triangle " + std::to_string(i) + " has no area (its barycentric determinant is zero or not finite)";
    }
    return std::string();
}

}  // namespace

int build_layout(const ptmi_config& cfg, const ptmi_scene* sc, Relayout& out, std::string& err)
{
    const uint32_t nn = sc->bvh_size, nt = sc->triangulation_size;
    if (nn == 0 || !sc->bvh) return fail(err, PTMI_ERR_BAD_SCENE, "bvh is empty (the kernel always reads bvh[0])");
    if (nt && !sc->triangulation) return fail(err, PTMI_ERR_INVALID_ARGUMENT, "triangulation is NULL");
    if (nt > REF_INDEX_MASK_LEAF) return fail(err, PTMI_ERR_LIMIT, "more than 2^27 triangles");
    if (sc->lights_size && !sc->lights) return fail(err, PTMI_ERR_INVALID_ARGUMENT, "lights is NULL");
    if (sc->materiaux_size && !sc->materiaux) return fail(err, PTMI_ERR_INVALID_ARGUMENT, "materiaux is NULL");
    if (!sc->sky) return fail(err, PTMI_ERR_INVALID_ARGUMENT, "sky is NULL");
    if (sc->textures_data_size && !sc->textures_data) return fail(err, PTMI_ERR_INVALID_ARGUMENT, "textures_data is NULL");
    if (sc->textures_size && !sc->textures) return fail(err, PTMI_ERR_INVALID_ARGUMENT, "textures is NULL");
    if (sc->lights_size != cfg.lights_size)
        return fail(err, PTMI_ERR_INVALID_ARGUMENT, "scene.lights_size differs from config.lights_size (LIGHTS_SIZE is baked at setup)");

    for (int f = 0; f < 6; f++)
        if (!texture_ok(sc->sky->sky_textures[f], sc->textures_data_size))
            return fail(err, PTMI_ERR_BAD_SCENE, "sky texture " + std::to_string(f) + " outside textures_data");
    for (uint32_t i = 0; i < sc->textures_size; i++)
        if (!texture_ok(sc->textures[i], sc->textures_data_size))
            return fail(err, PTMI_ERR_BAD_SCENE, "texture " + std::to_string(i) + " outside textures_data");

    out.mats.resize(sc->materiaux_size);
    for (uint32_t i = 0; i < sc->materiaux_size; i++) {
        const ptmi_material& m = sc->materiaux[i];
        if (!m.is_simple_color && (m.texture_id < 0 || (uint32_t)m.texture_id >= sc->textures_size))
            return fail(err, PTMI_ERR_BAD_SCENE, "material " + std::to_string(i) + " has an invalid textureId");
        DMat& d = out.mats[i];
        std::memcpy(d.color, &m.simple_color, 16);
        d.opacity = m.opacity;
        d.texture_id = m.texture_id;
        d.type = m.type;
        d.is_simple_color = m.is_simple_color ? 1u : 0u;
    }
    // The common untextured Lambert scene (BASELINE's Cornell box and 1M-triangle scene): the wavefront kernel has a
    // specialisation whose path logic holds neither the other four material types nor textures nor the other light types.
    out.plain_shading = std::getenv("PTMI_GENERIC_SHADING") == nullptr;  // developer switch for A/B runs and tests
    for (uint32_t i = 0; i < sc->materiaux_size && out.plain_shading; i++)
        if (sc->materiaux[i].type != PTMI_MAT_STANDART || !sc->materiaux[i].is_simple_color) out.plain_shading = false;
    for (uint32_t i = 0; i < sc->lights_size && out.plain_shading; i++)
        if (sc->lights[i].type != PTMI_LIGHT_POINT) out.plain_shading = false;

    out.tris.resize(nt);
    out.shade.resize(nt);
    for (uint32_t i = 0; i < nt; i++) {
        const ptmi_triangle& t = sc->triangulation[i];
        if (t.mat_pos >= sc->materiaux_size || t.mat_neg >= sc->materiaux_size)
            return fail(err, PTMI_ERR_BAD_SCENE, "triangle " + std::to_string(i) + " references a material out of range");
        if (!sc->materiaux[t.mat_pos].is_simple_color || !sc->materiaux[t.mat_neg].is_simple_color) {
            // texture coordinates index texels (header.cl:430-459: u - (int)u, then (uint)(u * (width - 1))): beyond the int range the
            // wrap does nothing and the index leaves the texture - in the reference as well, which reads whatever is there
            const float* uv = reinterpret_cast<const float*>(&t.uvp1);
            for (int k = 0; k < 12; k++)
                if (!(std::fabs(uv[k]) <= 0x1p+30f))
                    return fail(err, PTMI_ERR_BAD_SCENE, "triangle " + std::to_string(i) + " has texture coordinates that are not finite (or beyond 2^30)");
        }
        DTri& d = out.tris[i];
        std::memcpy(d.s1, &t.s1, 16); std::memcpy(d.s2, &t.s2, 16); std::memcpy(d.s3, &t.s3, 16);
        std::memcpy(d.n, &t.n, 16);
        DShade& s = out.shade[i];
        std::memset(&s, 0, sizeof s);
        std::memcpy(s.n1, &t.n1, 16); std::memcpy(s.n2, &t.n2, 16); std::memcpy(s.n3, &t.n3, 16);
        std::memcpy(s.uvp, &t.uvp1, 24);
        std::memcpy(s.uvn, &t.uvn1, 24);
        s.mat_pos = t.mat_pos;
        s.mat_neg = t.mat_neg;
    }

    out.literal_kernel_reason = scene_needs_literal_kernel(sc);

    // Ray-independent part of the triangle test, if every triangle keeps the importers' convention of equal w
    // on its three vertices (then the edge vectors have w = +0 exactly).  Same operations, same order, same
    // rounding as the kernel's generic form: dot() = fma chain over four components (ptmi_device.hpp).
    out.tris_precomputed = std::getenv("PTMI_GENERIC_TRIANGLES") == nullptr;  // developer switch for A/B runs
    for (uint32_t i = 0; i < nt && out.tris_precomputed; i++) {
        const ptmi_triangle& t = sc->triangulation[i];
        if (!(t.s1.w == t.s2.w && t.s1.w == t.s3.w && std::isfinite(t.s1.w))) out.tris_precomputed = false;  // edge vectors need w = +0 exactly
    }
    if (out.tris_precomputed) {
        auto dot4 = [](const float a[4], const float b[4]) {
            return std::fmaf(a[3], b[3], std::fmaf(a[2], b[2], std::fmaf(a[1], b[1], a[0] * b[0])));
        };
        for (uint32_t i = 0; i < nt; i++) {
            const ptmi_triangle& t = sc->triangulation[i];
            const float S1[4] = {t.s1.x, t.s1.y, t.s1.z, t.s1.w}, N[4] = {t.n.x, t.n.y, t.n.z, t.n.w};
            const float u[4] = {t.s2.x - t.s1.x, t.s2.y - t.s1.y, t.s2.z - t.s1.z, t.s2.w - t.s1.w};
            const float v[4] = {t.s3.x - t.s1.x, t.s3.y - t.s1.y, t.s3.z - t.s1.z, t.s3.w - t.s1.w};
            const float uv = dot4(u, v), uu = dot4(u, u), vv = dot4(v, v);
            const float denom = 1 / (uv * uv - uu * vv);
            DTriPre p;
            std::memcpy(p.n, N, 16);
            p.s1d[0] = S1[0]; p.s1d[1] = S1[1]; p.s1d[2] = S1[2]; p.s1d[3] = dot4(N, S1);
            p.u_den[0] = u[0]; p.u_den[1] = u[1]; p.u_den[2] = u[2]; p.u_den[3] = denom;
            p.v_s1w[0] = v[0]; p.v_s1w[1] = v[1]; p.v_s1w[2] = v[2]; p.v_s1w[3] = S1[3];
            std::memcpy(&out.tris[i], &p, sizeof p);
        }
    }

    // Walk the tree from bvh[0] exactly as the traversal could and emit ONE array of 64-byte records in depth-first
    // order: an inner node's record, then the triangles of its leaf children, then son1's subtree, then son2's.
    // What a ray reads next is then usually the neighbour of what it has just read: the caches fetch 128-byte lines,
    // so a bottom-level node brings its first triangle along and an even-numbered node its first inner child
    // (the integrator is bound by cache misses in flight).  `seen` rejects cycles and shared subtrees.
    std::vector<uint8_t> seen(nn, 0);
    std::string why;
    auto check_node = [&](uint32_t id) -> bool {
        if (id >= nn) { why = "child index out of range"; return false; }
        if (seen[id]) { why = "node " + std::to_string(id) + " reached twice (cycle or shared subtree)"; return false; }
        seen[id] = 1;
        const ptmi_node& n = sc->bvh[id];
        if (n.is_leaf) {
            if ((uint64_t)n.triangle_start_index + n.nb_triangles > nt) { why = "leaf triangle range out of bounds"; return false; }
        } else if (n.cut_axis > 2) { why = "cutAxis > 2"; return false; }
        return true;
    };
    // appends the triangles of leaf `id` and returns the reference to them
    auto emit_leaf = [&](uint32_t id, uint32_t* ref) -> bool {
        const ptmi_node& n = sc->bvh[id];
        const size_t start = out.recs.size();
        if (start + n.nb_triangles > REF_INDEX_MASK_LEAF) { why = "more than 2^27 records"; return false; }
        for (uint32_t k = 0; k < n.nb_triangles; k++) {
            out.recs.push_back(out.tris[n.triangle_start_index + k]);
            out.tri_ids.push_back(n.triangle_start_index + k);
        }
        // A leaf without triangles is stored as an EMPTY child: never descended, which gives the reference's results and
        // counters (its box is still tested and counted; visiting it would test nothing) and keeps the wavefront
        // kernel's invariant that a decoded leaf leaves a non-empty triangle range behind.
        uint32_t r = ((n.triangles_aabb.is_empty || n.nb_triangles == 0) ? REF_EMPTY : 0u) | REF_LEAF;
        if (n.nb_triangles < REF_COUNT_BIG) {
            r |= (n.nb_triangles << REF_COUNT_SHIFT) | (uint32_t)start;
        } else {
            r |= (REF_COUNT_BIG << REF_COUNT_SHIFT) | (uint32_t)out.big_leaves.size();
            out.big_leaves.push_back(DBigLeaf{(uint32_t)start, n.nb_triangles});
        }
        *ref = r;
        return true;
    };
    auto node_at = [&](size_t rec) -> DNode& { return *reinterpret_cast<DNode*>(&out.recs[rec]); };

    constexpr uint32_t NO_PARENT = 0xFFFFFFFFu;
    struct Item { uint32_t id, parent_rec, slot, depth; };
    std::vector<Item> todo;
    if (!check_node(0)) return fail(err, PTMI_ERR_BAD_SCENE, "bvh[0]: " + why);
    if (sc->bvh[0].is_leaf) {
        if (!emit_leaf(0, &out.root_ref)) return fail(err, PTMI_ERR_LIMIT, "bvh[0]: " + why);
    } else {
        out.root_ref = (sc->bvh[0].triangles_aabb.is_empty ? REF_EMPTY : 0u);  // index 0, patched like any inner child
        todo.push_back({0, NO_PARENT, 0, 0});
    }
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        const ptmi_node& n = sc->bvh[it.id];
        const size_t self = out.recs.size();
        if (self > REF_INDEX_MASK_LEAF) return fail(err, PTMI_ERR_LIMIT, "more than 2^27 records");
        out.recs.emplace_back();
        out.tri_ids.push_back(0xFFFFFFFFu);
        if (it.parent_rec == NO_PARENT) out.root_ref |= (uint32_t)self;
        else (it.slot == 0 ? node_at(it.parent_rec).ref1 : node_at(it.parent_rec).ref2) |= (uint32_t)self;
        if (!check_node(n.son1_id) || !check_node(n.son2_id))
            return fail(err, PTMI_ERR_BAD_SCENE, "bvh[" + std::to_string(it.id) + "]: " + why);
        const ptmi_node& c1 = sc->bvh[n.son1_id];
        const ptmi_node& c2 = sc->bvh[n.son2_id];
        uint32_t r1 = c1.triangles_aabb.is_empty ? REF_EMPTY : 0u, r2 = c2.triangles_aabb.is_empty ? REF_EMPTY : 0u;
        if (c1.is_leaf && !emit_leaf(n.son1_id, &r1)) return fail(err, PTMI_ERR_LIMIT, "bvh[" + std::to_string(it.id) + "]: " + why);
        if (c2.is_leaf && !emit_leaf(n.son2_id, &r2)) return fail(err, PTMI_ERR_LIMIT, "bvh[" + std::to_string(it.id) + "]: " + why);
        DNode& d = node_at(self);
        const ptmi_bounding_box& b1 = c1.triangles_aabb;
        const ptmi_bounding_box& b2 = c2.triangles_aabb;
        d.lo1[0] = b1.p_min.x; d.lo1[1] = b1.p_min.y; d.lo1[2] = b1.p_min.z;
        d.hi1[0] = b1.p_max.x; d.hi1[1] = b1.p_max.y; d.hi1[2] = b1.p_max.z;
        d.lo2[0] = b2.p_min.x; d.lo2[1] = b2.p_min.y; d.lo2[2] = b2.p_min.z;
        d.hi2[0] = b2.p_max.x; d.hi2[1] = b2.p_max.y; d.hi2[2] = b2.p_max.z;
        d.ref1 = r1; d.ref2 = r2; d.axis = n.cut_axis; d.pad = 0;  // inner children: index patched in when they are emitted
        // the short slab test (box_hit_ordered) needs finite, ordered boxes; anything else keeps the literal form
        for (int k = 0; k < 3; k++) {
            if (!(r1 & REF_EMPTY) && !(std::isfinite(d.lo1[k]) && std::isfinite(d.hi1[k]) && d.lo1[k] <= d.hi1[k])) out.boxes_ordered = false;
            if (!(r2 & REF_EMPTY) && !(std::isfinite(d.lo2[k]) && std::isfinite(d.hi2[k]) && d.lo2[k] <= d.hi2[k])) out.boxes_ordered = false;
        }
        const uint32_t child_depth = it.depth + 1;
        if (child_depth > out.max_depth) out.max_depth = child_depth;
        // push son2 first so that son1's subtree follows this node's own records
        if (!c2.is_leaf) todo.push_back({n.son2_id, (uint32_t)self, 1, child_depth});
        if (!c1.is_leaf) todo.push_back({n.son1_id, (uint32_t)self, 0, child_depth});
    }
    if (out.recs.empty()) { out.recs.emplace_back(); out.tri_ids.push_back(0xFFFFFFFFu); }  // a root leaf without triangles
    // box_hit_ordered tests no isEmpty flag: an empty child is stored as an inverted infinite box (never hit)
    if (out.boxes_ordered)
        for (size_t i = 0; i < out.recs.size(); i++) {
            if (out.tri_ids[i] != 0xFFFFFFFFu) continue;
            DNode& d = node_at(i);
            const float inf = std::numeric_limits<float>::infinity();
            if (d.ref1 & REF_EMPTY) for (int k = 0; k < 3; k++) { d.lo1[k] = inf; d.hi1[k] = -inf; }
            if (d.ref2 & REF_EMPTY) for (int k = 0; k < 3; k++) { d.lo2[k] = inf; d.hi2[k] = -inf; }
        }
    // the traversal stack has 30 entries (FullKernel.cl:627); the reference refuses deeper trees (PathTracer.cpp:54-58)
    if (out.max_depth >= PTMI_BVH_MAX_DEPTH)
        return fail(err, PTMI_ERR_LIMIT, "bvh depth " + std::to_string(out.max_depth) + " >= 30");
    // The walk of a ray whose direction is NaN in every component (Relayout::nan_walk_*): BVH_IntersectRay (FullKernel.cl:620-702)
    // with every box test answering "not empty" and dir[cutAxis] > 0 false - son2 first, son1 pushed (:660-697).
    {
        uint64_t boxes = 0, tris = 0;
        uint32_t last = 0xFFFFFFFFu;
        std::vector<uint32_t> stack;
        uint32_t cur = out.root_ref;
        for (;;) {
            if (cur & REF_LEAF) {
                uint32_t count = (cur >> REF_COUNT_SHIFT) & 7u, start = cur & REF_INDEX_MASK_LEAF;
                if (count == REF_COUNT_BIG) { const DBigLeaf& bl = out.big_leaves[start]; start = bl.start; count = bl.count; }
                tris += count;
                if (count) last = start + count - 1;
                if (stack.empty()) break;
                cur = stack.back(); stack.pop_back();
            } else {
                const DNode& d = node_at(cur & REF_INDEX_MASK_INNER);
                boxes += 2;
                const bool near_hit = !(d.ref2 & REF_EMPTY), far_hit = !(d.ref1 & REF_EMPTY);
                if (near_hit) {
                    if (far_hit) stack.push_back(d.ref1);
                    cur = d.ref2;
                } else if (far_hit) {
                    cur = d.ref1;
                } else {
                    if (stack.empty()) break;
                    cur = stack.back(); stack.pop_back();
                }
            }
        }
        // (per-path counters are 32 bits wide in the kernels: a walk that does not fit is simply walked)
        if (boxes < 0x10000000ull && tris < 0x10000000ull) {
            out.nan_walk_box_tests = (uint32_t)boxes; out.nan_walk_tri_tests = (uint32_t)tris; out.nan_walk_last_tri = last;
        } else {
            out.nan_walk_box_tests = out.nan_walk_tri_tests = 0xFFFFFFFFu;
        }
    }
    return PTMI_OK;
}

}  // namespace ptmi_internal

// Host-only entry point (no device, no context): would ptmi_initialize_memory accept this scene under this configuration?
extern "C" int ptmi_validate_scene(const ptmi_config* config, const ptmi_scene* scene)
{
    if (!config || config->struct_size != sizeof(ptmi_config) || !scene || scene->struct_size != sizeof(ptmi_scene)) {
        ptmi_internal::set_global_error("ptmi_validate_scene: config / scene is NULL or struct_size mismatch (ABI)");
        return PTMI_ERR_INVALID_ARGUMENT;
    }
    ptmi_internal::Relayout lay;
    std::string err;
    const int rc = ptmi_internal::build_layout(*config, scene, lay, err);
    if (rc == PTMI_OK && !lay.literal_kernel_reason.empty() && config->super_sampling && config->sampler == PTMI_SAMPLER_RANDOM) {
        ptmi_internal::set_global_error("SUPER_SAMPLING needs the wavefront kernel, which cannot reproduce the reference on this scene with "
                                        "the RANDOM sampler: " + lay.literal_kernel_reason);
        return PTMI_ERR_UNSUPPORTED;
    }
    if (rc != PTMI_OK) ptmi_internal::set_global_error(err);
    return rc;
}
