// bvh_build.cpp - host-side binned-SAH BVH build behind ptmi_bvh_create().
//
// Produces the Node[] the integrator consumes, bit-compatible with the
// reference's BVH_Create (Controleur/PathTracer_BVH.cpp:12-37) and its recursive
// BVH_BuildStructure (:109-356): 64 centroid bins on each of the three axes,
// SAH = Nl*Al + Nr*Ar with the half-area of PathTracer_Structs.h:264-271, leaf
// when <= 4 triangles / centroid diagonal^2 < 1e-3 / SAH not worth it, Hoare
// style in-place partition of the triangle array, depth-first pre-order node
// numbering (left child = parent + 1).  The float/double mix of the reference
// (float products, double sums of int*area, k1 rounded through double) is kept
// operation for operation, because the traversal order and therefore the image
// depend on the exact tree.  tests/test_bvh.py checks equality against the
// reference builder compiled unmodified (oracle/_ref/libref_bvh.so) and
// against committed tree digests.
//
// Host only: no device needed.  Compile with -ffp-contract=off.

#include <algorithm>
#include <climits>
#include <cmath>
#include <string>
#include <cstring>
#include <vector>

#include "ptmi.h"
#include "ptmi_internal.h"

namespace {

constexpr int kBins = 64;               // const__K, BVH.cpp:118
constexpr uint32_t kLeafMaxSize = 4;    // const__leafMaxSize, :116
constexpr float kLeafMinDiag = 0.001f;  // const__leafMinDiagLength, :117
constexpr float kKI = 1.0f;             // const__KI, :114
constexpr float kKT = 0.01f;            // const__KT, :115

inline ptmi_float4 min4(const ptmi_float4& a, const ptmi_float4& b)
{
    // ppmin = std::min<float> per component, all four (Utils.h:132)
    return { std::min<float>(a.x, b.x), std::min<float>(a.y, b.y), std::min<float>(a.z, b.z), std::min<float>(a.w, b.w) };
}
inline ptmi_float4 max4(const ptmi_float4& a, const ptmi_float4& b)
{
    return { std::max<float>(a.x, b.x), std::max<float>(a.y, b.y), std::max<float>(a.z, b.z), std::max<float>(a.w, b.w) };
}
inline ptmi_float4 mid4(const ptmi_float4& a, const ptmi_float4& b)
{
    return { (a.x + b.x) / 2, (a.y + b.y) / 2, (a.z + b.z) / 2, (a.w + b.w) / 2 };
}

// BoundingBox_UniteWith, Structs.h:218-236
inline void unite(ptmi_bounding_box& self, const ptmi_bounding_box& bb)
{
    if (bb.is_empty) return;
    if (self.is_empty) {
        self.p_min = bb.p_min; self.p_max = bb.p_max; self.centroid = bb.centroid; self.is_empty = 0;
        return;
    }
    self.p_min = min4(self.p_min, bb.p_min);
    self.p_max = max4(self.p_max, bb.p_max);
    self.centroid = mid4(self.p_min, self.p_max);
}

// BoundingBox_AddPoint, Structs.h:245-261
inline void add_point(ptmi_bounding_box& self, const ptmi_float4& v)
{
    if (self.is_empty) {
        self.is_empty = 0; self.p_min = v; self.p_max = v; self.centroid = v;
        return;
    }
    self.p_min = min4(self.p_min, v);
    self.p_max = max4(self.p_max, v);
    self.centroid = mid4(self.p_min, self.p_max);
}

// BoundingBox_Area, Structs.h:263-271: float products and sums, widened on return
inline double half_area(const ptmi_bounding_box& bb)
{
    if (bb.is_empty) return 0;
    const float dx = bb.p_max.x - bb.p_min.x, dy = bb.p_max.y - bb.p_min.y, dz = bb.p_max.z - bb.p_min.z;
    const float a = dx * dy + dy * dz + dz * dx;
    return a;
}

inline float axis_of(const ptmi_float4& v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }

struct Builder {
    ptmi_triangle* tris;
    ptmi_node* nodes;
    uint32_t size = 0, max_depth = 0, depth = 0;

    // The reference keeps these as function-level statics shared by every
    // recursion level (BVH.cpp:154-161); k1 of an axis that is skipped at some
    // node keeps the value of an earlier node.  Same lifetime here.
    ptmi_bounding_box bin_box[3][kBins];
    int bin_count[3][kBins];
    float sah[3][kBins - 1];
    ptmi_bounding_box l2r_box[3][kBins], r2l_box[3][kBins];
    int l2r_count[3][kBins], r2l_count[3][kBins];
    float k1[3] = { 0, 0, 0 };
    bool failed = false;    // a centroid fell outside its node's bins, or a split left a side empty
    uint32_t capacity = 0;  // nodes the caller's array holds: 2n - 1

    void make_node(uint32_t idx, uint32_t start, uint32_t count, const ptmi_bounding_box& tri_box,
                   const ptmi_bounding_box& cen_box)
    {
        // BVH_CreateNode, BVH.cpp:42-53 (fields it leaves untouched are zero here)
        ptmi_node& n = nodes[idx];
        std::memset(&n, 0, sizeof n);
        n.triangle_start_index = start;
        n.nb_triangles = count;
        n.triangles_aabb = tri_box;
        n.centroids_aabb = cen_box;
    }

    uint32_t make_leaf(uint32_t idx, int why)
    {
        nodes[idx].is_leaf = 1;
        nodes[idx].comments = why;
        if (depth > max_depth) max_depth = depth;
        return idx + 1;
    }

    // BVH_BuildStructure, BVH.cpp:109-356.  Returns the index after the subtree.
    uint32_t build(uint32_t idx)
    {
        ptmi_node* N = &nodes[idx];
        // (boxes whose extents overflow make the cost comparisons meaningless - a split can then leave a side empty and never end;
        // a tree this deep is of no use to the integrator either: PTMI_BVH_MAX_DEPTH)
        if (failed || depth > 8 * PTMI_BVH_MAX_DEPTH) { failed = true; return idx + 1; }

        if (N->nb_triangles <= kLeafMaxSize) return make_leaf(idx, PTMI_NODE_LEAF_MAX_SIZE);
        {
            const ptmi_float4 &a = N->centroids_aabb.p_max, &b = N->centroids_aabb.p_min;
            const float dx = b.x - a.x, dy = b.y - a.y, dz = b.z - a.z;
            if ((dx * dx) + (dy * dy) + (dz * dz) < kLeafMinDiag) return make_leaf(idx, PTMI_NODE_LEAF_MIN_DIAG);
        }

        const int first = (int)N->triangle_start_index;
        const int last = first + (int)N->nb_triangles - 1;
        const ptmi_float4 cmin = N->centroids_aabb.p_min, cmax = N->centroids_aabb.p_max;

        for (int axis = 0; axis < 3; axis++)
            for (int i = 0; i < kBins; i++) {
                bin_box[axis][i].is_empty = 1;
                bin_count[axis][i] = 0;
                if (i < kBins - 1) sah[axis][i] = (float)INT_MAX;
            }

        for (int axis = 0; axis < 3; axis++) {
            const double cut_length = axis_of(cmax, axis) - axis_of(cmin, axis);
            if (cut_length < kLeafMinDiag) continue;
            k1[axis] = (float)((float)kBins * (0.999f) / cut_length);

            const float lo = axis_of(cmin, axis);
            for (int i = first; i <= last; i++) {
                const float scaled = k1[axis] * (axis_of(tris[i].aabb.centroid, axis) - lo);
                // (the reference ASSERTs triangleBin < const__K, BVH.cpp:199, and indexes out of bounds without it: centroids so
                // far apart that their difference overflows - the entry point below already refuses non-finite ones)
                if (!(scaled >= 0.0f && scaled < (float)kBins)) { failed = true; return idx + 1; }
                const int bin = (int)scaled;
                bin_count[axis][bin]++;
                unite(bin_box[axis][bin], tris[i].aabb);
            }
            for (int i = 0; i < kBins; i++) {
                l2r_box[axis][i] = bin_box[axis][i];
                r2l_box[axis][i] = bin_box[axis][i];
                l2r_count[axis][i] = bin_count[axis][i];
                r2l_count[axis][i] = bin_count[axis][i];
            }
            for (int i = 1, j = kBins - 2; i < kBins; i++, j--) {
                unite(l2r_box[axis][i], l2r_box[axis][i - 1]);
                l2r_count[axis][i] += l2r_count[axis][i - 1];
                unite(r2l_box[axis][j], r2l_box[axis][j + 1]);
                r2l_count[axis][j] += r2l_count[axis][j + 1];
            }
            for (int i = 0; i < kBins - 1; i++)
                sah[axis][i] = (float)(l2r_count[axis][i] * half_area(l2r_box[axis][i]) +
                                       r2l_count[axis][i + 1] * half_area(r2l_box[axis][i + 1]));
        }

        int best_axis = 0, best_index = 0;
        float best_sah = sah[0][0];
        for (int axis = 0; axis < 3; axis++)
            for (int i = 1; i < kBins - 1; i++)
                if (sah[axis][i] < best_sah) { best_axis = axis; best_sah = sah[axis][i]; best_index = i; }

        if (kKI * best_sah + kKT > N->nb_triangles * half_area(N->triangles_aabb))
            return make_leaf(idx, PTMI_NODE_BAD_SAH);

        N->cut_axis = (uint32_t)best_axis;

        int left = first, right = last;
        {
            const float lo = axis_of(cmin, best_axis), k = k1[best_axis];
            const int cut = best_index + 1;
            while (left < right) {
                while ((k * (axis_of(tris[left].aabb.centroid, best_axis) - lo)) < cut && left < right) left++;
                while ((k * (axis_of(tris[right].aabb.centroid, best_axis) - lo)) >= cut && left < right) right--;
                if (left < right) std::swap(tris[left], tris[right]);
            }
        }

        ptmi_bounding_box left_cen, right_cen;
        std::memset(&left_cen, 0, sizeof left_cen);
        std::memset(&right_cen, 0, sizeof right_cen);
        left_cen.is_empty = 1; right_cen.is_empty = 1;
        for (int i = first; i < left; i++) add_point(left_cen, tris[i].aabb.centroid);
        for (int i = last; i >= left; i--) add_point(right_cen, tris[i].aabb.centroid);

        // saved before recursing: the scratch arrays are shared (BVH.cpp:322-325)
        const uint32_t son2_start = (uint32_t)(first + l2r_count[best_axis][best_index]);
        const uint32_t son2_count = (uint32_t)r2l_count[best_axis][best_index + 1];
        const ptmi_bounding_box son2_box = r2l_box[best_axis][best_index + 1];

        // A split always leaves triangles on both sides when the costs are numbers, and the tree then has at most 2n - 1 nodes -
        // what the caller allocated (BVH.cpp:20).  With boxes whose extents overflow it need not: refuse instead of writing past
        // the array (the reference would).
        if (l2r_count[best_axis][best_index] <= 0 || son2_count == 0 || (uint64_t)size + 2 > capacity) { failed = true; return idx + 1; }
        depth++;
        size += 2;

        const uint32_t son1 = idx + 1;
        N->son1_id = son1;
        make_node(son1, (uint32_t)first, (uint32_t)l2r_count[best_axis][best_index], l2r_box[best_axis][best_index],
                  left_cen);
        const uint32_t son2 = build(son1);
        if (failed || son2 >= capacity) { failed = true; depth--; return idx + 1; }
        nodes[idx].son2_id = son2;
        make_node(son2, son2_start, son2_count, son2_box, right_cen);
        const uint32_t next = build(son2);
        depth--;
        return next;
    }
};

}  // namespace

extern "C" int ptmi_bvh_create(ptmi_triangle* triangulation, uint32_t n, ptmi_node* bvh, uint32_t* bvh_size,
                               uint32_t* bvh_max_depth)
{
    if (!triangulation || !bvh || n == 0) {
        ptmi_internal::set_global_error("ptmi_bvh_create: null array or empty triangulation");
        return PTMI_ERR_INVALID_ARGUMENT;
    }
    // The reference's builder has no defined behaviour for boxes that are not numbers (its bin ASSERT fires, BVH.cpp:199; without
    // assertions it writes out of bounds): an error code here.
    for (uint32_t i = 0; i < n; i++) {
        const ptmi_bounding_box& a = triangulation[i].aabb;
        const float v[9] = {a.p_min.x, a.p_min.y, a.p_min.z, a.p_max.x, a.p_max.y, a.p_max.z, a.centroid.x, a.centroid.y, a.centroid.z};
        for (float f : v)
            if (!std::isfinite(f)) {
                ptmi_internal::set_global_error("ptmi_bvh_create: triangle " + std::to_string(i) + " has a bounding box that is not finite");
                return PTMI_ERR_BAD_SCENE;
            }
    }
    // BVH_Create, BVH.cpp:12-37
    ptmi_bounding_box full_tri, full_cen;
    std::memset(&full_tri, 0, sizeof full_tri);
    std::memset(&full_cen, 0, sizeof full_cen);
    full_tri.is_empty = 1; full_cen.is_empty = 1;
    for (uint32_t i = 0; i < n; i++) {
        unite(full_tri, triangulation[i].aabb);
        add_point(full_cen, triangulation[i].aabb.centroid);
    }
    std::vector<Builder> holder(1);  // ~60 KB of scratch: keep it off the stack
    Builder& b = holder[0];
    b.tris = triangulation;
    b.nodes = bvh;
    b.capacity = 2u * n - 1u;
    b.make_node(0, 0, n, full_tri, full_cen);
    b.size = 1;
    b.build(0);
    if (b.failed) {
        ptmi_internal::set_global_error("ptmi_bvh_create: bounding boxes too large to build a tree from (their extents overflow)");
        return PTMI_ERR_BAD_SCENE;
    }
    if (bvh_size) *bvh_size = b.size;
    if (bvh_max_depth) *bvh_max_depth = b.max_depth;
    return PTMI_OK;
}
