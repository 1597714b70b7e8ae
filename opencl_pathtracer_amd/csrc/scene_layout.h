// scene_layout.h - validation and re-layout of a scene for the device: host code without a device call (scene_layout.cpp).
// Used by ptmi_initialize_memory (ptmi_api.cpp) and, on its own, by ptmi_validate_scene - which is also how the tests run it
// under AddressSanitizer on a box without a GPU.
#pragma once

#include <string>
#include <vector>

#include "ptmi.h"
#include "ptmi_internal.h"

namespace ptmi_internal {

// Everything the kernel will index is checked so that a malformed scene is an error code, not a GPU fault.
struct Relayout {
    std::vector<DTri> recs;          // the traversal's one array: DNode and DTri/DTriPre records interleaved
    std::vector<uint32_t> tri_ids;   // per record: index into triangulation[] / shade[] (0xFFFFFFFF for a node)
    std::vector<DTri> tris;          // per input triangle, only a staging area for recs
    std::vector<DShade> shade;
    std::vector<DMat> mats;
    std::vector<DBigLeaf> big_leaves;
    bool tris_precomputed = false;
    bool plain_shading = false; // every material a plain-colour MAT_STANDART, every light a LIGHT_POINT
    bool boxes_ordered = true;  // all non-empty child boxes finite with pMin <= pMax
    std::string literal_kernel_reason;  // scene_needs_literal_kernel()
    uint32_t root_ref = 0;
    uint32_t max_depth = 0;
    // What a closest-hit query makes of a ray whose direction is not a number in ANY component (every comparison of the box
    // test and of the triangle test is false: every non-empty box is "hit", every triangle accepted): it walks the whole tree
    // in one fixed order and returns its LAST triangle.  The walk's box tests, triangle tests and last triangle (a record
    // index; 0xFFFFFFFF: the walk meets no triangle), so that the literal loops can skip it (ptmi_literal_path.hpp).
    uint32_t nan_walk_box_tests = 0, nan_walk_tri_tests = 0, nan_walk_last_tri = 0xFFFFFFFFu;
};

// Returns PTMI_OK or an error code with its message in `err`.  cfg: lights_size and sampler are read.
int build_layout(const ptmi_config& cfg, const ptmi_scene* sc, Relayout& out, std::string& err);

}  // namespace ptmi_internal
