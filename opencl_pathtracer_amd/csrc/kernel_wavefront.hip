// kernel_wavefront.hip - persistent-wavefront path tracer for gfx950 (MI355X).  The default integrator.
//
// Why not one work-item = one path like the reference (Kernel_Main, FullKernel.cl:1180): on a 64-wide
// wavefront a path that ends after one bounce would idle until the slowest of its 63 neighbours has done
// ten, and closest-hit traversal, shadow traversal and shading would run one after the other under
// partial exec masks.  Here instead:
//
//  * The grid is PERSISTENT: as many workgroups as the chip holds, alive for the whole launch.  Work is
//    jobs, job = ONE PATH (pixel, iteration), in 8 queues (8 stripes of 8x8 tiles, tile-major: the iterations
//    of a tile are consecutive jobs); a wave starts on the queue of its workgroup's XCD group, so neighbouring
//    tiles run at the same time on one L2, and takes jobs for all lanes that ran dry with one atomic.
//    A finished path stores its radiance (16 B) and its three statistics bins (4 B) into staging arrays
//    [iteration][pixel]; two trivial follow-up kernels add the staged radiances to the framebuffer pixel by
//    pixel IN ITERATION ORDER (the float sums are the reference's, bit for bit, without atomics) and count
//    the histograms in LDS.  (Jobs of one pixel x all iterations kept the sum in registers but left a tail of
//    one whole job per lane: -12 % at 1080p.  Three global atomics per path for the histograms: -79 % on the
//    Cornell box.)
//  * Every lane is a small STATE MACHINE over the same three kinds of work:
//        I  one inner-node step  (load one 64-byte DNode, two slab tests, push/pop on the LDS stack)
//        T  the triangles of a leaf (64-byte DTri records)
//        P  path logic           (shade a finished closest-hit query, set up / account a shadow ray,
//                                 scatter, finish the path, start the next iteration or fetch a new job)
//    A camera segment and a shadow ray are the same I/T work with another `limit` and exit rule, so lanes
//    in either phase, of any bounce, of any pixel, run together.
//  * LOOP NEST: traversal trips in an inner loop, a path-logic pass in the outer loop when the lanes waiting for
//    it have waited long enough (wait debt) or nothing can traverse.  A trip is either a NODE TRIP - every lane
//    at an inner node takes one step - or a LEAF PASS: a lane that reaches a leaf waits, and when about nineteen
//    lanes wait (or more triangles than there are lanes left at inner nodes) the next <= 64 triangles of all
//    waiting leaves are dealt out one per lane to ALL lanes of the wave, whatever their own state; the owner's ray
//    comes through the cross-lane network (ds_bpermute), the results go back through one 64-bit LDS minimum per
//    owner.  Triangle tests run at 96 % lane utilisation (a leaf of 3 triangles: one pass instead of three trips
//    under a 30 % exec mask); node trips at 66 %.  (PTMI_WF_LEAF_PASS=0 builds round 1's mixed trips, where every
//    traversing lane takes one step per trip, node or triangle.)
//  * Per ray the visit sequence is exactly the reference's (near child first, far child pushed, leaf
//    triangles in index order, limit updated between tests, FullKernel.cl:620-702): only WHEN a lane
//    takes its next step changes, never WHICH tests it makes or what they see (see leaf_pass for the one place
//    where tests of one ray run side by side).  Results are bit-identical to the one-path-per-lane kernel
//    (kernels.hip) and to the CPU checker.
//  * The one thing a leaf pass cannot express is a NaN distance (the reference accepts it and then keeps the LAST triangle
//    that passes, not the nearest): a path whose ray is not a number is GIVEN UP where path logic sets the ray up - a marked
//    radiance goes to its staging slot - and redo_poisoned_kernel, queued behind every staged launch, traces it again with
//    the reference's literal loops (ptmi_literal_path.hpp).  Scenes whose RECORDS yield NaN distances never get here
//    (scene_layout.cpp: scene_needs_literal_kernel).
//
// Exit: a lane dies when it has seen every queue empty; a wave leaves the outer loop when no lane is alive
// (every path is bounded by the ray depth, every traversal by the finite tree, and a pass or trip only runs
// with at least one lane that wants it).
//
// The kernel's limits, measured: DESIGN.md 5 "What binds".  tools/isa_trip.sh prints the instruction mix of the
// traversal loop.
#include <hip/hip_runtime.h>

#include "ptmi_device.hpp"
#include "ptmi_shading.hpp"
#include "ptmi_literal_path.hpp"

#include <atomic>
#include <cstdlib>
#include <cstring>

namespace PTMI_DEV_NS {

constexpr int kWfBlock = 256;  // lanes per workgroup: trees of up to 22 levels (five workgroups = twenty waves per CU)
// ... and for deeper trees (BLOCK template argument): the LDS a workgroup needs - (depth + 9) KB per 256 lanes - is then handed
// out in quarters, so that a CU whose 160 KB hold FOUR wide workgroups (16 waves; 23 to 27 levels) holds 17-19 narrow ones.
// Measured on MI355X, same box, Msamples/s, workgroups of 256 / 128 / 64 lanes: configs[4] stand-in (general shading, depth 23)
// 1240 / 1299 / 1301; 4M triangles (plain, depth 24) in profiles/r04_ab_workgroup_size.txt; 1M triangles (plain, depth 22: twenty
// waves either way) 976 / 931 / 938 - so the narrow form is only launched where the wide one does not get its five workgroups.
constexpr int kWfBlockNarrow = 64;
constexpr int kWfStack = PTMI_BVH_MAX_DEPTH;
#ifndef PTMI_WF_WAIT_DEBT
// lane-trips of waiting a wave tolerates before it spends a pass on path logic: a launch parameter (DWarm::wait_debt),
// 768 from tree depth 16 on, below that 512 (general path logic) or 320 (plain scenes: the cheaper a pass, the sooner it
// pays); PTMI_WF_WAIT_DEBT > 0 fixes it at build time for sweeps.
// Measured on MI355X (Msamples/s: 1M triangles 1080p depth 22 / Cornell box 1080p d8 depth 5 / material mix 4K d16):
//   round 2, plain-scene specialisation (first two scenes):  192: - / 6360 / 2193    256: - / 6620 / 2317    320: - / 6670 / 2403
//                           384: - / 6600 / 2442    512: 960 / 6430 / 2491    640: 963 / 6190 / -    768: 976 / - / -    1024: 959 / - / -
//   round 2, leaf passes:   512: 885 / 5388 / 2344    768: 895 / 5172 / 2314    1024: 895 / 5121 / 2235
//   round 1, mixed trips:   256: - / 4883 / 1876    384: 733 / - / 1969    512: 740 / 4804 / 1986    768: 742 / 4896 / 1944
//                           1024: 737 / 4845 / 1872    2048: - / 4840 / 1718
// (a fixed threshold of 8 waiting lanes instead of a debt measured 474 / 743 with an early build)
#define PTMI_WF_WAIT_DEBT 0
#endif
#ifndef PTMI_WF_MIN_WAVES
// waves per SIMD the register allocator must fit (5 -> 96 VGPRs, the overflow spills to scratch inside the path-logic
// code, which is ~1 % of the loop trips).  Measured on MI355X, same box (Msamples/s, 1M triangles / Cornell / material
// mix 4K): 3: 415, 4: 500 / 2731 / 1382, 5: 533 / 2708 / 1375, 6: 409.
#define PTMI_WF_MIN_WAVES 5
#endif
#ifndef PTMI_WF_MIN_WAVES_GENERAL
// ... of the general shading specialisation (textures, all material and light types: twice the path-logic code of the
// plain one), whose waves spend 40 % of their life in path logic on the material-mix scene
#define PTMI_WF_MIN_WAVES_GENERAL 5
#endif
#ifndef PTMI_WF_QUEUES
#define PTMI_WF_QUEUES 8
#endif
constexpr int kQueues = PTMI_WF_QUEUES;  // job queues (image stripes), one per XCD group of workgroups
#ifndef PTMI_WF_QUEUE_STRIDE
#define PTMI_WF_QUEUE_STRIDE 64
#endif
constexpr int kQueueStride = PTMI_WF_QUEUE_STRIDE;  // dwords between two queue counters (64 = one 256-byte block each)
// word 1 of the job-counter block is the launch's "a path was given up" flag (redo_poisoned_kernel): free only while the
// queue counters are at least two words apart
static_assert(kQueueStride >= 2, "job_counter[1] is the given-up flag: queue counters must not be adjacent words");
constexpr int kWaitDebtFixed = PTMI_WF_WAIT_DEBT;
#ifndef PTMI_WF_HIT_WORDS
// LDS words of the closest-hit record per lane.  8: hit point (4), s, t, triangle | front, found.  4: the ray parameter
// instead of the point - path logic rebuilds the point from the ray it still holds with the very operations of the
// triangle test, bit for bit - s, t, and one word triangle | front | found.  LDS per workgroup = (tree depth + 1 + words)
// KB + 4 KB of leaf-pass keys and items (kLeafPassWords) = (depth + 9) KB with 4 words, plus the static counter block (8 * C_COUNT bytes): trees up to depth 22 keep five
// workgroups per CU (160 KB); the 4M-triangle scene (depth 24: 33 KB) and the 16M one (depth 27) run four.
// Measured on MI355X, same box, 1M triangles 1080p: 4 words 759-761, 8 words 751-753 Msamples/s.
#define PTMI_WF_HIT_WORDS 4
#endif
constexpr int kHitWords = PTMI_WF_HIT_WORDS;
static_assert(kHitWords == 4 || kHitWords == 8, "closest-hit record: 4 or 8 words");
#ifndef PTMI_WF_LEAF_PASS
// 1: LEAF PASSES.  A lane that reaches a leaf does not test its triangles itself, one per trip, under whatever exec mask the
// trip happens to have; it waits, and when enough lanes wait the wave runs one pass in which the triangles of ALL waiting
// leaves are dealt out as work items, one per lane, to all 64 lanes (the ray of an item's owner travels through the
// cross-lane network).  A leaf of 3 triangles costs one pass at full lane utilisation instead of three trips at ~30 %.
// Same tests, same order of acceptance per ray (see leaf_pass): results bit-identical.
#define PTMI_WF_LEAF_PASS 1
#endif
constexpr bool kLeafPass = PTMI_WF_LEAF_PASS != 0;
static_assert(!kLeafPass || kHitWords == 4, "leaf passes write the 4-word closest-hit record");
#ifndef PTMI_WF_LEAF_LANES
#define PTMI_WF_LEAF_LANES 19
#endif
#ifndef PTMI_WF_LEAF_RATIO
#define PTMI_WF_LEAF_RATIO 2
#endif
// a pass (up to 64 triangles, one per lane) runs when this many lanes wait at a leaf (3 triangles each on average), or when
// the waiting triangles are kLeafRatio times as many as the lanes left to take node steps
constexpr int kLeafLanes = PTMI_WF_LEAF_LANES, kLeafRatio = PTMI_WF_LEAF_RATIO;
constexpr int kLeafPassWordsPerLane = kLeafPass ? 4 : 0;  // LDS: one 64-bit key per lane + 64 items of 8 bytes per wave
#ifndef PTMI_WF_TOS
// The top entry of a lane's traversal stack ALSO in a register (round 4): a pop takes the register and the LDS read that refills
// it is not needed before the lane's NEXT pop, instead of an LDS round trip on the critical path of every node step.
// 0: never, 1: every instantiation, 2: the general shading instantiations only.  Measured on MI355X, same box, two rounds,
// Msamples/s (1M triangles plain / 1M triangles general shading forced / Cornell box 1080p d8 plain / material mix 4K d16 general /
// configs[4] stand-in 4K d16 general, four workgroups per CU / 4M triangles plain, four workgroups per CU):
//   0: 973.9 / 916.2 / 7197 / 2511 / 1161 / 517.1      1: 965.7 / 943.1 / 7215 / 2658 / 1211 / 510.7      2: 973.6 / 941.8 / 7263 / 2659 / 1214 / 516.9
// The general instantiations gain 3-6 % (their spilled registers also fall from 67 to 30: the allocator's doing), the plain one
// loses 0.8 %: hence 2.
#define PTMI_WF_TOS 2
#endif

// Scene fields by value (SGPRs): what the traversal trips and EVERY path-logic trip need.  The rarely used
// rest of DScene (sky: only when a path escapes; histogram / RANDOM-sampler / SUPER_SAMPLING buffers; counters)
// is read from a device-memory copy where it is used.  Measured on MI355X: the whole DScene by value (~80 SGPRs)
// spills SGPRs into the hot loop; everything through memory stalls each path-logic trip on ~50 scalar loads
// (-4.5 %); this split is the fastest of the three.
struct DWarm {
    const DNode* nodes;
    const DTri* tris;
    const DBigLeaf* big_leaves;
    const uint32_t* tri_ids;
    const DShade* shade;
    const DMat* mats;
    const ptmi_light* lights;
    const ptmi_texture* textures;
    const ptmi_uchar4* texels;
    uint32_t root_ref;
    uint32_t width, height;
    uint32_t max_depth, n_lights, sampler, tris_precomputed;
    uint32_t histograms;  // hist_depths != nullptr
    uint32_t boxes_ordered;
    uint32_t wide_records;  // the record array is 4 GB or more: 64-bit addressing
    uint32_t russian_roulette;  // PTMI_FLAG_RUSSIAN_ROULETTE (non-parity mode)
    uint32_t source_seed;       // PTMI_FLAG_SOURCE_SEED (non-parity mode)
    uint32_t wait_debt;         // see PTMI_WF_WAIT_DEBT
    uint32_t split_paths;       // DScene::split_paths
};

// A finished path's three histogram bins in one word: depth (6 bits, < kStatDepthBins), box tests and triangle tests
// (13 bits each; anything >= MAX_INTERSECTION_NUMBER = 5000 is not counted by the reference, stored as 8191).
constexpr uint32_t kStatDepthBins = 64;
// A path the wavefront kernel gives up (where it sets up a ray that is not a number, below): its radiance becomes this NaN - the
// payload survives the additions and fused multiply-adds that follow (tools/microbench/nan_payload.hip) -, its counters
// restart at zero, the launch's job-counter block gets a non-zero word 1, and redo_poisoned_kernel traces the path again.
constexpr uint32_t kPoisonMarker = 0x7FC0DEADu;
// The RANDOM sampler stages nothing (its samples land on arbitrary pixels and are added atomically), so a path it gives up is
// not marked in a staging slot: its slot number goes to a LIST in the launch's job-counter block - word 2 counts, the entries
// start at word kGiveUpListFirst, behind the queue counters - and bit 31 of the lane's slot word says "this path adds nothing";
// redo_random_kernel traces the listed paths again and adds them atomically.  A launch that fills the list (7680 paths whose
// ray is not a number: scenes of NaN records never get here, they run the one-path-per-lane kernel with this sampler) lets the
// rest keep the ordered minimum, as every such path did before round 4.
constexpr uint32_t kGivenUp = 0x80000000u, kGiveUpListFirst = (uint32_t)(kQueues * kQueueStride), kGiveUpListCap = 8u * 1024u - kGiveUpListFirst;
static_assert(kQueues * kQueueStride <= 1024, "the give-up list lies behind the queue counters inside the set's 8 x 1024 dwords");
// A launch that renders AHEAD for several calls at once (DScene::split_paths = staging slots per call, ptmi_api.cpp) keeps the
// totals of ptmi_get_counters per call: the paths of call k add theirs to block k - kSplitWords words each in LDS (PATHS .. TRI:
// all a launch without scheduler statistics counts), C_COUNT words apart in the set's counter block.  Any other launch:
// split_paths = 0, everything in block 0.  (The workgroup's LDS block is as large as before round 4 - 24 words, of which the
// production instantiations used six: at 22 stack levels five workgroups fill a CU's LDS to the last allocation granule, and a
// block of 48 words cost the fifth, 974 -> 882 Msamples/s.)
constexpr uint32_t kSplitWords = C_TRI + 1;
static_assert(PTMI_COUNTER_SPLITS * (int)kSplitWords <= 64, "flushed by one wave");
__device__ __forceinline__ uint32_t counter_split_of(uint32_t slot, uint32_t split_paths)
{
    // (three compares instead of a division: at most PTMI_COUNTER_SPLITS = 4 calls share a launch)
    const uint32_t t1 = split_paths ? split_paths : 0xFFFFFFFFu, t2 = split_paths ? 2u * split_paths : 0xFFFFFFFFu,
                   t3 = split_paths ? 3u * split_paths : 0xFFFFFFFFu;
    return (slot >= t1 ? 1u : 0u) + (slot >= t2 ? 1u : 0u) + (slot >= t3 ? 1u : 0u);
}

__device__ __forceinline__ uint32_t pack_path_statistics(uint32_t depth, uint32_t bbx, uint32_t tri)
{
    static_assert(PTMI_MAX_INTERSECTION_NUMBER <= 8191, "13-bit fields");
    const uint32_t b = bbx < PTMI_MAX_INTERSECTION_NUMBER ? bbx : 8191u, t = tri < PTMI_MAX_INTERSECTION_NUMBER ? tri : 8191u;
    return depth | (b << 6) | (t << 19);
}

__device__ __forceinline__ void decode_leaf(const DWarm& sc, uint32_t ref, uint32_t& tri_i, uint32_t& tri_end)
{
    uint32_t count = (ref >> REF_COUNT_SHIFT) & 7u;
    uint32_t start = ref & REF_INDEX_MASK_LEAF;
    if (count == REF_COUNT_BIG) {
        const DBigLeaf bl = sc.big_leaves[start];
        start = bl.start;
        count = bl.count;
    }
    tri_i = start;
    tri_end = start + count;
}

// PLAIN: path logic for scenes of plain-colour MAT_STANDART materials and ONE LIGHT_POINT (DScene::plain_shading) rendered
// with the JITTERED sampler and without Russian roulette: the same operations for those renders, without the code - and
// the registers - of the other material types, light types, samplers and of the light loop (1M triangles +4.9 %, Cornell
// box +8.5 %).
// NANSAFE: for scenes whose RECORDS can make a triangle test compute a NaN distance (scene_layout.cpp:
// scene_needs_literal_kernel - a zero-area triangle as the importer emits it, astronomical coordinates).  The reference accepts
// such a test (its rejections are comparisons, FullKernel.cl:533-567) and from then on keeps the LAST triangle that passes,
// which a minimum over ordered keys cannot express.  This instantiation looks at every ACCEPTED triangle of a closest-hit
// query inside the leaf pass; a NaN distance there marks the owner's key, the owner gives its path up exactly like a path whose
// ray is not a number, and redo_poisoned_kernel traces it again with the literal loops.  Everything else - and every path that
// never reaches such a record - is the ordinary kernel.  (A shadow query needs nothing: it ends at the FIRST accepted triangle
// in index order whatever the distances are, and its limit never changes.)  Clean scenes run the instantiations without it.
// BLOCK: lanes per workgroup (kWfBlock, or kWfBlockNarrow for deep trees: see there).
template <bool STATS, bool PRE, bool SS, bool PLAIN = false, bool NANSAFE = false, int BLOCK = 256>
__global__ void __launch_bounds__(BLOCK, PLAIN ? PTMI_WF_MIN_WAVES : PTMI_WF_MIN_WAVES_GENERAL) render_wavefront_kernel(
                                                                    const DScene* __restrict__ scene_in_memory, const DWarm sc_arg, const uint32_t first_iteration,
                                                                    const uint32_t n_iterations, const uint32_t iteration_stride,
                                                                    const uint32_t n_jobs,
                                                                    uint32_t* __restrict__ job_counter,
                                                                    const uint32_t stack_levels,
                                                                    float* __restrict__ stage,
                                                                    uint32_t* __restrict__ stage_stats)
{
    // traversal stacks: [level][lane], one dword per entry, as many levels as the tree is deep (the reference
    // reserves 30, FullKernel.cl:627; the deepest possible chain of pending far children is the tree depth)
    // PLAIN also fixes the JITTERED sampler, no Russian roulette and ONE light (the launch only picks it then): constants below
    DWarm sc = sc_arg;
    if (PLAIN) { sc.sampler = PTMI_SAMPLER_JITTERED; sc.russian_roulette = 0; sc.n_lights = 1; }  // ... and ONE light
    constexpr bool kOneLight = PLAIN;  // (nothing gathered across shadow queries, no light index: five registers fewer per lane)
    constexpr int kWfBlock = BLOCK;  // (shadows the namespace's: every LDS stride below is the workgroup's own width)
    constexpr bool kTos = PTMI_WF_TOS == 1 || (PTMI_WF_TOS == 2 && !PLAIN);
    extern __shared__ __attribute__((aligned(16))) uint32_t stack_mem[];
    // (the statistics build: one block with every counter, its launches never render for several calls)
    constexpr uint32_t kBlockCounters = STATS ? (uint32_t)C_COUNT : PTMI_COUNTER_SPLITS * kSplitWords;
    static_assert(kBlockCounters <= 64, "zeroed and flushed by one wave");
    __shared__ unsigned long long block_counters[kBlockCounters];

    const uint32_t tid = threadIdx.x;
    if (tid < kBlockCounters) block_counters[tid] = 0;
    __syncthreads();
    // LDS: [closest-hit record: 8][sentinel][stack levels...], each level one dword per lane.  The sentinel below the
    // stack holds REF_NONE: popping an empty stack yields "query finished" without a test (and, sitting behind the
    // hit record, its address minus one level cannot wrap below zero).  Pushes write the slot above the top for
    // every lane and move the top only for pushing lanes; an inner node at depth k has at most k pending entries
    // above it, so that slot is always inside the `stack_levels` = tree depth levels.
    uint32_t* const stack_floor = &stack_mem[kHitWords * kWfBlock + tid];
    *stack_floor = REF_NONE;
    // the rare fields: pointer re-derived through an opaque asm so the loads stay where they are used
    auto cold_scene = [&]() -> const DScene& {
        const DScene* p = scene_in_memory;
        asm volatile("" : "+s"(p));
        return *p;
    };
    // The closest-hit record (point, s, t, triangle, side) changes only when a closer hit is accepted and is read
    // only by path logic: it lives in LDS in front of the stack, [field][lane], not in registers of the hot loop.
    uint32_t* const hit_mem = &stack_mem[tid];
    // the record's triangle word = triangle record index (< 2^27) | kHitFront when the ray met the front side (N . dir < 0)
    // (| kHitFound in the 4-word record, whose last word it is; the 8-word record keeps "found" in a word of its own)
    constexpr uint32_t kHitFront = 0x80000000u, kHitFound = 0x40000000u;
    constexpr int kWordS = kHitWords == 8 ? 4 : 1, kWordT = kHitWords == 8 ? 5 : 2, kWordTri = kHitWords == 8 ? 6 : 3;
    const uint32_t tiles_x = (sc.width + 7u) >> 3;
    const bool owns_pixel = sc.sampler != PTMI_SAMPLER_RANDOM;
    const uint32_t n_tiles = (n_jobs / n_iterations) >> 6;
    uint32_t q_cur = blockIdx.x % (uint32_t)kQueues, q_done = 0;  // wave-uniform: current queue, queues seen empty

    // ---- lane state -----------------------------------------------------------------------------
    // Whether a lane has a path in flight, or is done for good, is kept IN `cur` (REF_IDLE / REF_DEAD) so that the
    // wave's three step masks come from three integer compares; the two bools below only live inside a path-logic trip.
    bool alive = true, need_path = true;
    // The path's job as ONE word: its slot in the staging arrays = it_local * W * H + y * W + x.  Iteration ids of this
    // launch: first_iteration + it_local * iteration_stride, it_local < n_iterations (the stride is the number of devices
    // that share a render: each takes the ids of its own residue class, ptmi_api.cpp).  Pixel and iteration are only needed
    // when the path starts; everything a lane keeps across traversal trips costs a register of the 96 (5 waves per SIMD).
    uint32_t slot = 0;
    // path
    int seed = 1;
    V4 transfer = v4(1, 1, 1, 1), radiance = v4(0, 0, 0, 0);
    uint32_t reflection = 0, p_bbx = 0, p_tri = 0;
    bool in_water = false;
    // current query
    Ray r;
    r.o = v4(0, 0, 0, 0); r.d = v4(0, 0, 0, 0); r.ix = r.iy = r.iz = 0;
    float limit = 0;
    bool shadow = false;
    bool exact_boxes = false;  // this ray needs the literal box test (see box_hit_ordered)
    bool wave_exact = true;    // ... and so does some ray of this wave (wave-uniform, refreshed after every path-logic pass)
    uint32_t cur = REF_IDLE, tri_i = 0, tri_end = 0;
    uint32_t dir_signs = 0;  // bit k: direction component k > 0 (which child of a node cut along k is the near one)
    uint32_t* sp = stack_floor;  // the top entry (the sentinel when the stack is empty)
    uint32_t tos = REF_NONE;     // kTos: ... and its value (PTMI_WF_TOS); invariant tos == *sp
    // the point of the closest hit, as path logic needs it when a query has finished
    auto load_hit_point = [&]() {
        if (kHitWords == 8)
            return v4(__uint_as_float(hit_mem[0 * kWfBlock]), __uint_as_float(hit_mem[1 * kWfBlock]),
                      __uint_as_float(hit_mem[2 * kWfBlock]), __uint_as_float(hit_mem[3 * kWfBlock]));
        // 4-word record.  After a shadow query the lane's ray still starts at the hit point (:932-936 shoot from it without
        // an offset); after a closest-hit query the ray is the one that found the hit, and the point is
        // origin + direction * parameter evaluated as Triangle_Intersects does (FullKernel.cl:536).
        if (shadow) return r.o;
        return mad(r.d, __uint_as_float(hit_mem[0 * kWfBlock]), r.o);
    };
    auto query_found = [&]() {
        return kHitWords == 8 ? (hit_mem[7 * kWfBlock] & 2u) != 0 : (hit_mem[kWordTri * kWfBlock] & kHitFound) != 0;
    };
    // saved across the shadow rays of one surface hit: the direction the surface was reached along and the light gathered
    // so far.  The surface itself (normals, colour, material) is NOT kept: it is a pure function of the hit record in LDS
    // and cam_d and is rebuilt where it is used (once per hit with one light), which frees ~15 registers of every trip.
    V4 cam_d = v4(0, 0, 0, 0), direct = v4(0, 0, 0, 0);
    uint32_t light_idx = 0;
    // wave-uniform scheduler statistics (scalar registers): trips and active lanes per step kind
    uint32_t trips_i = 0, trips_t = 0, trips_p = 0;
    unsigned long long lanes_i = 0, lanes_t = 0, lanes_p = 0;
    unsigned long long cycles_p = 0;
    const unsigned long long loop_start = STATS ? __builtin_amdgcn_s_memtime() : 0ull;

    // the reference's -D LOG_INFO checks (header.cl:21-48), counted in the statistics build instead of printed (ptmi_invariant_checks)
    auto check = [&](bool holds, int slot) {
        if (STATS && !holds) atomicAdd(&block_counters[slot], 1ull);
    };
    // statistics + accumulation of a finished path (FullKernel.cl:1319-1345).  `missed`: the path ended on a sky miss,
    // i.e. it made one closest-hit query more than it has surface hits.
    auto finish_path = [&](bool missed) {
        // totals of the launch (ptmi_get_counters): added per finished path into the workgroup's LDS block - everything is
        // a function of the three per-path counters, so no lane keeps running totals in registers
        // (the index goes through an opaque asm so that the compiler does not see a wave-uniform address: for those it sums
        // the lanes in a scalar loop first - five loops on the critical path of a pass in which two or three lanes finish.
        // Plain LDS atomics: 1M triangles +0.2 %, Cornell box +4.1 %, material mix +1.6 %)
        // (... and the call the path belongs to, when the launch renders ahead for several: counter_split_of)
        uint32_t split_paths = sc.split_paths;
        asm volatile("" : "+s"(split_paths));  // (the three thresholds are made HERE, not kept in scalar registers across the loops)
        uint32_t none = STATS ? 0u : counter_split_of(slot, split_paths) * kSplitWords;
        asm volatile("" : "+v"(none));
        unsigned long long* const totals = &block_counters[none];
        atomicAdd(&totals[C_PATHS], 1ull);
        atomicAdd(&totals[C_HITS], (unsigned long long)reflection);
        atomicAdd(&totals[C_SEGMENTS], (unsigned long long)(reflection + (missed ? 1u : 0u)));
        atomicAdd(&totals[C_BBX], (unsigned long long)p_bbx);
        atomicAdd(&totals[C_TRI], (unsigned long long)p_tri);
        check(p_bbx < PTMI_MAX_INTERSECTION_NUMBER && p_tri < PTMI_MAX_INTERSECTION_NUMBER, C_CHK_STATS_RANGE);  // cl:1325,1330
        const bool given_up = !owns_pixel && (slot & kGivenUp) != 0u;  // (RANDOM sampler: redo_random_kernel adds this path)
        if (sc.histograms && stage_stats == nullptr && !given_up) {
            // RANDOM sampler / very deep paths: the three statistics atomics as the reference issues them (:1319-1331)
            const DScene& cs = cold_scene();
            atomicAdd(&cs.hist_depths[reflection], 1u);
            if (p_bbx < PTMI_MAX_INTERSECTION_NUMBER) atomicAdd(&cs.hist_bbx[p_bbx], 1u);
            if (p_tri < PTMI_MAX_INTERSECTION_NUMBER) atomicAdd(&cs.hist_tri[p_tri], 1u);
        }
        if (owns_pixel) {
            // JITTERED / UNIFORM: the sample lands on the work-item's own pixel (:1333-1336); stage it
            reinterpret_cast<float4*>(stage)[slot] = make_float4(radiance.x, radiance.y, radiance.z, radiance.w);
            // ... and its three histogram bins in one word; histogram_staged_kernel counts them afterwards in LDS.
            // (Three global atomics per path on a handful of hot bins cost 3 % on the 1M-triangle scene and 79 % on the
            // Cornell box: atomics of different XCDs on one address are resolved memory-side.)
            if (stage_stats != nullptr) stage_stats[slot] = pack_path_statistics(reflection, p_bbx, p_tri);
            if (SS) cold_scene().stage_flag[slot] = 1.f;
        } else if (!given_up) {
            // RANDOM sampler: the sample lands on an arbitrary pixel; the reference races there (:1339-1345).  The sample
            // position is drawn again from the path's seed (the first two draws, :1137-1141) instead of being kept.
            const uint32_t n_pixels = sc.width * sc.height;
            const uint32_t it_local = slot / n_pixels, pixel = slot - it_local * n_pixels;
            const uint32_t gy = pixel / sc.width, gx = pixel - gy * sc.width;
            const uint32_t it = first_iteration + it_local * iteration_stride;
            int seed0 = lcg_seed(gx, gy, sc.width, sc.height, it, sc.source_seed != 0);
            float sample_x, sample_y;
            draw_sample(sc, gx, gy, it, seed0, sample_x, sample_y);
            const uint32_t off = sample_pixel(sc, sample_x, sample_y);
            const DScene& cs = cold_scene();
            // (the atomics return what the accumulators held before: sumBefore / nRayBefore of :1339-1342)
            const V4 before = v4(atomicAdd(&cs.image_color[4 * off + 0], radiance.x), atomicAdd(&cs.image_color[4 * off + 1], radiance.y),
                                 atomicAdd(&cs.image_color[4 * off + 2], radiance.z), atomicAdd(&cs.image_color[4 * off + 3], radiance.w));
            const float n_before = atomicAdd(&cs.image_ray_nb[off], 1.f);
            if (SS) {
                // SUPER_SAMPLING with the RANDOM sampler (:1346-1349): the reference read-modify-writes the variance of a pixel
                // other work-items may be updating too; here every update is an atomic add.  As in the reference, a pixel whose
                // first sample arrives after iteration 0 divides 0 by 0 here, keeps a NaN variance and is never skipped (:1168:
                // the comparison with NaN is false) - with this sampler 29 % of the pixels get no sample in iteration 0, so that
                // quirk decides how many samples a render takes and is kept (the staged form guards its one such case instead).
                float* const vp = &cs.image_v[4 * off];
                if (it != 0u) {
                    const V4 after = before + radiance;
                    const float n_after = n_before + 1.f;
                    atomicAdd(&vp[0], (radiance.x - fdiv(before.x, n_before)) * (radiance.x - fdiv(after.x, n_after)));
                    atomicAdd(&vp[1], (radiance.y - fdiv(before.y, n_before)) * (radiance.y - fdiv(after.y, n_after)));
                    atomicAdd(&vp[2], (radiance.z - fdiv(before.z, n_before)) * (radiance.z - fdiv(after.z, n_after)));
                    atomicAdd(&vp[3], (radiance.w - fdiv(before.w, n_before)) * (radiance.w - fdiv(after.w, n_after)));
                }
            }
        }
        need_path = true;
        cur = REF_NONE; tri_i = tri_end = 0;
    };

    // Begin a BVH query at the root.  Invariant kept by every step: outside a leaf's triangle range `cur`
    // is an inner-node reference or REF_NONE; a leaf reference is decoded into [tri_i, tri_end) as soon as
    // the range is free.
    auto start_query = [&]() {
        cur = sc.root_ref; sp = stack_floor; tri_i = tri_end = 0;
        if (kTos) tos = REF_NONE;
        // "found" of THIS query (a shadow query leaves the rest of the closest hit's record alone)
        if (kHitWords == 8) hit_mem[7 * kWfBlock] = 0;
        else if (shadow) atomicAnd(&hit_mem[kWordTri * kWfBlock], ~kHitFound);
        else hit_mem[kWordTri * kWfBlock] = 0;
        exact_boxes = !(sc.boxes_ordered && ray_slabs_are_ordered(r));
        if (cur & REF_LEAF) {  // the whole scene is one leaf
            decode_leaf(sc, cur, tri_i, tri_end);
            cur = REF_NONE;
        }
    };

    // ---- leaf passes (kLeafPass) -----------------------------------------------------------------------------------
    // Per wave: items = the next (up to 4) triangles of every lane that waits at a leaf, numbered owner by owner (prefix sum
    // of the counts from three ballots); a pass takes the first 64 and lane k tests item k with the OWNER's ray, limit and
    // mode (an owner whose triangles did not all fit offers the rest to the next pass).
    // Equivalence with the reference's leaf loop (FullKernel.cl:638-646, limit updated between tests):
    //   closest hit: a triangle is accepted iff it passes every other test and its squared distance <= the current limit,
    //     which then becomes that distance; so after the leaf the record holds the LAST triangle (in index order) among
    //     those that pass against the limit at entry with the smallest distance: one 64-bit minimum per owner over the key
    //     (distance bits, ~index).  Testing against the limit at entry only admits candidates that lose that minimum.
    //   shadow: the FIRST triangle that passes ends the query (:724-727) and only the tests up to it are counted:
    //     a minimum over the index.
    // All LDS traffic of a pass stays inside the wave (LDS operations of one wave execute in order): no barrier.
    unsigned long long* const key_mem = reinterpret_cast<unsigned long long*>(&stack_mem[(kHitWords + 1 + stack_levels) * kWfBlock]);
    uint2* const item_mem = reinterpret_cast<uint2*>(&stack_mem[(kHitWords + 3 + stack_levels) * kWfBlock]) + (tid & ~63u);
    const uint32_t lane = tid & 63u, wave_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid & ~63u));  // (a scalar)
    uint32_t pass_rounds = 0, pass_items = 0, item_violations = 0;
    const uint32_t n_records = STATS ? cold_scene().n_records : 0u;
    auto leaf_pass = [&](bool waits_at_leaf) {
        auto lanes_below = [&](unsigned long long m) {
            return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        };
        // triangles this lane offers (up to 4 of its leaf's), as three bit planes over the wave: prefix sum and total
        const uint32_t left = tri_end - tri_i;
        const uint32_t offered = waits_at_leaf ? (left < 4u ? left : 4u) : 0u;
        const unsigned long long plane0 = __builtin_amdgcn_ballot_w64((offered & 1u) != 0), plane1 = __builtin_amdgcn_ballot_w64((offered & 2u) != 0),
                                 plane2 = __builtin_amdgcn_ballot_w64((offered & 4u) != 0);
        const int n_items = __popcll(plane0) + 2 * __popcll(plane1) + 4 * __popcll(plane2);
        // one pass = one item per lane: the first 64 items; an owner whose triangles do not all fit keeps the rest
        const uint32_t first_item = lanes_below(plane0) + 2u * lanes_below(plane1) + 4u * lanes_below(plane2);
        const uint32_t room = first_item < 64u ? 64u - first_item : 0u;
        const uint32_t cnt = offered < room ? offered : room;
        const uint32_t n_now = n_items < 64 ? (uint32_t)n_items : 64u;  // wave-uniform
        if (STATS) { pass_rounds++; pass_items += n_now; }
        if (cnt != 0u) {
            key_mem[tid] = shadow ? ~0ull : (((unsigned long long)__float_as_uint(limit) << 32) | 0xFFFFFFFFull);
#pragma unroll
            for (uint32_t j = 0; j < 4u; j++)  // item = (triangle record index (< 2^27) | shadow query, owner)
                if (j < cnt) item_mem[first_item + j] = make_uint2((tri_i + j) | (shadow ? 0x80000000u : 0u), lane);
        }
        {
            const bool has_item = lane < n_now;
            // one LDS read tells a lane its triangle and its owner: the record loads go out before the owner's ray is fetched
            // (items of one byte - owner and position - with the triangle index fetched from the owner: 864 -> 896 Msamples/s)
            const uint2 item = has_item ? item_mem[lane] : make_uint2(0u, 0u);
            const uint32_t owner = item.y;
            const int from = (int)(owner << 2);
            auto fetch = [&](float x) { return __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(x))); };
            const uint32_t w_tri = item.x & 0x7FFFFFFFu;
            const bool w_shadow = (item.x >> 31) != 0u;
            // the item protocol's invariant, checked where statistics are collected (tests/test_parity_gpu.py): an item a lane
            // reads was written by its owner in THIS pass - a stale one would name a lane or a record that may not exist
            if (STATS && has_item && (owner >= 64u || w_tri >= n_records)) item_violations++;
            const float4* const rec = reinterpret_cast<const float4*>(&sc.tris[has_item ? w_tri : 0u]);
            constexpr int kE1 = PRE ? 1 : 3, kL0 = PRE ? 2 : 1, kL1 = PRE ? 3 : 2;
            const float4 e0 = rec[0], e1 = rec[kE1];  // (lanes without an item read record 0: in bounds, unused)
            asm volatile("" ::: "memory");  // (keeps the loads above the ray fetches: 860 -> 894 Msamples/s)
            Ray wr;
            wr.o = v4(fetch(r.o.x), fetch(r.o.y), fetch(r.o.z), fetch(r.o.w));
            wr.d = v4(fetch(r.d.x), fetch(r.d.y), fetch(r.d.z), fetch(r.d.w));
            wr.ix = wr.iy = wr.iz = 0;
            const float w_limit = fetch(limit);
            if (has_item) {
                unsigned long long* const owner_key = &key_mem[wave_first + owner];
                // (all four quads up front instead of two now and two for the lanes that pass the distance tests: +0.3 %, not kept)
                tri_test<PRE>(e0, e1, [&](float4& l0, float4& l1) { l0 = rec[kL0]; l1 = rec[kL1]; }, wr, w_limit,
                              [&](const V4&, float ray_t, float s, float t, bool front, float nsd) {
                    // every lane that accepts is here at the same time: one LDS minimum for all of them, then each asks
                    // whether it is (so far) the one its owner keeps
                    // (NANSAFE: an accepted NaN distance of a closest-hit query -> key 0, below every real key: a distance
                    // under 1e-5 is never accepted, cl:543)
                    const bool nan_hit = NANSAFE && !w_shadow && !(nsd == nsd);
                    const unsigned long long key = w_shadow ? ((unsigned long long)w_tri << 32)
                                                   : nan_hit ? 0ull
                                                   : (((unsigned long long)__float_as_uint(nsd) << 32) | (unsigned long long)(~w_tri));
                    atomicMin(owner_key, key);
                    if (!w_shadow && *owner_key == key) {  // the closest so far: its record is the one path logic shades from
                        uint32_t* const rec_out = &stack_mem[wave_first + owner];
                        rec_out[0 * kWfBlock] = __float_as_uint(ray_t);
                        rec_out[kWordS * kWfBlock] = __float_as_uint(s);
                        rec_out[kWordT * kWfBlock] = __float_as_uint(t);
                        rec_out[kWordTri * kWfBlock] = w_tri | (front ? kHitFront : 0u) | kHitFound;
                    }
                });
            }
        }
        if (cnt != 0u) {
            const uint32_t won = (uint32_t)(key_mem[tid] >> 32);
            if (NANSAFE && !shadow && won == 0u) {
                // a triangle of this leaf was accepted with a NaN distance: the path is given up (as where path logic meets
                // a ray that is not a number, below) - marked radiance, counters back to zero, the query ends as a miss -
                // and redo_poisoned_kernel traces it again
                const float m = __uint_as_float(kPoisonMarker);
                radiance = v4(m, m, m, m);
                transfer = v4(1, 1, 1, 1);
                reflection = 0; p_bbx = 0; p_tri = 0;
                hit_mem[kWordTri * kWfBlock] = 0;
                cur = REF_NONE; tri_i = tri_end = 0; sp = stack_floor;
                if (kTos) tos = REF_NONE;
                job_counter[1] = 1u;
            } else
            if (!shadow) {
                limit = __uint_as_float(won);
                p_tri += cnt; tri_i += cnt;
            } else if (won != 0xFFFFFFFFu) {
                p_tri += won - tri_i + 1u;
                hit_mem[kWordTri * kWfBlock] |= kHitFound;
                tri_end = tri_i; cur = REF_NONE;
            } else {
                p_tri += cnt; tri_i += cnt;
            }
            if (tri_i >= tri_end && cur != REF_NONE && (cur & REF_LEAF)) {
                decode_leaf(sc, cur, tri_i, tri_end);
                cur = kTos ? tos : *sp;
                uint32_t* const under = sp - kWfBlock;
                sp = under < stack_floor ? stack_floor : under;
                if (kTos) tos = *sp;
            }
        }
    };
    // one inner-node step of the calling lanes (:660-697): the same code as in the mixed trip below
    // (always the 64-bit address form here: one instruction more than the 32-bit offset of the mixed trip, measured
    // faster - 851 vs 837 Msamples/s - than choosing between the two)
    auto node_step = [&]() {
        const float4* const rec = reinterpret_cast<const float4*>(&sc.tris[cur & REF_INDEX_MASK_INNER]);
        const float4 a = rec[0], b = rec[1], c = rec[2], d = rec[3];
        const float lo1[3] = {a.x, a.y, a.z}, hi1[3] = {a.w, b.x, b.y};
        const float lo2[3] = {b.z, b.w, c.x}, hi2[3] = {c.y, c.z, c.w};
        const uint32_t ref1 = __float_as_uint(d.x), ref2 = __float_as_uint(d.y), axis = __float_as_uint(d.z);
        // dir[cutAxis] > 0 (:663) from a per-ray word of the three signs (one bit test; selected from the three sign masks
        // the box tests hold it took three compares and five scalar instructions)
        const bool fwd = ((dir_signs >> axis) & 1u) != 0;
        bool h1, h2;
        if (!wave_exact) {
            h1 = box_hit_ordered(lo1, hi1, r, limit);
            h2 = box_hit_ordered(lo2, hi2, r, limit);
        } else {
            h1 = box_hit(lo1, hi1, (ref1 & REF_EMPTY) != 0, r, limit);
            h2 = box_hit(lo2, hi2, (ref2 & REF_EMPTY) != 0, r, limit);
        }
        p_bbx += 2;
        // (every choice as a select on the two hit masks themselves: combined into new booleans first - both, neither - the
        // compiler builds them as 0 / 1 integers in vector registers: seven instructions more per step)
        if (kTos) {
            const uint32_t far_ref = fwd ? ref2 : ref1;
            sp[kWfBlock] = far_ref;
            uint32_t* const pushed = sp + kWfBlock;
            const uint32_t child = fwd ? (h1 ? ref1 : ref2) : (h2 ? ref2 : ref1);  // near child if it was hit, else the far one
            // both hit: the far child becomes the top; one hit: nothing moves; none: the top is the next node and the entry
            // under it becomes the top - read from LDS, but needed only when this lane pops the next time
            cur = h1 ? child : (h2 ? child : tos);
            tos = h1 ? (h2 ? far_ref : tos) : tos;
            sp = h1 ? (h2 ? pushed : sp) : sp;
            if (!(h1 | h2)) {
                uint32_t* const below = sp - kWfBlock;
                sp = below < stack_floor ? stack_floor : below;
                tos = *sp;
            }
            if (cur != REF_NONE && (cur & REF_LEAF)) {  // (the triangle range of a lane that takes node steps is free)
                decode_leaf(sc, cur, tri_i, tri_end);
                cur = tos;
                uint32_t* const under = sp - kWfBlock;
                sp = under < stack_floor ? stack_floor : under;
                tos = *sp;
            }
            return;
        }
        sp[kWfBlock] = fwd ? ref2 : ref1;
        uint32_t* const pushed = sp + kWfBlock;
        sp = h1 ? (h2 ? pushed : sp) : sp;
        const uint32_t popped = *sp;
        uint32_t* const below = sp - kWfBlock;
        uint32_t* const under = below < stack_floor ? stack_floor : below;
        const uint32_t child = fwd ? (h1 ? ref1 : ref2) : (h2 ? ref2 : ref1);  // near child if it was hit, else the far one
        cur = h1 ? child : (h2 ? child : popped);
        sp = h1 ? sp : (h2 ? sp : under);
        if (cur != REF_NONE && (cur & REF_LEAF)) {  // (the triangle range of a lane that takes node steps is free)
            decode_leaf(sc, cur, tri_i, tri_end);
            cur = *sp;
            uint32_t* const under = sp - kWfBlock;
            sp = under < stack_floor ? stack_floor : under;
        }
    };

    // Scheduling of path logic: a fixed lane threshold cannot serve both a 21-node scene (queries of ~10 steps,
    // all lanes finish together: waiting for a full wave is nearly free and a threshold of 8 runs the long
    // path-logic code at 12 % utilisation: 743 vs 2897 Msamples/s on the Cornell box) and a million-triangle
    // scene (ragged queries of ~120 steps: waiting starves the traversal, optimum ~10 lanes).  So the wave keeps
    // a WAIT DEBT = sum over traversal trips of the number of lanes that sat waiting, and runs path logic when
    // the debt crosses a bound: bursts of finishers are served together (and leave path logic in lockstep, which
    // keeps the following traversal trips uniform), stragglers are not waited for.
    int wait_debt = 0;
    // lane states: triangles pending (idle and dead lanes keep an empty range) / `cur` is an inner-node reference /
    // neither: the query is finished (REF_NONE) or there is no path (REF_IDLE) -> path logic / REF_DEAD
    bool pending, want_inner, want_post;
    int n_t, n_i, n_p;
    bool any_lane;
    unsigned long long dead_lanes = 0ull;  // (lanes only die in path logic: the mask is taken there)
    auto survey = [&]() {
        pending = tri_i < tri_end;
        const bool has_node = cur < REF_DEAD;
        const unsigned long long b_t = __builtin_amdgcn_ballot_w64(pending), b_n = __builtin_amdgcn_ballot_w64(has_node), b_dead = dead_lanes;
        const unsigned long long m_t = b_t, m_i = b_n & ~b_t, m_p = ~(b_t | b_n | b_dead);
        want_inner = !pending & has_node;
        want_post = !pending & !has_node & (cur != REF_DEAD);
        any_lane = (m_t | m_i | m_p) != 0ull;
        n_t = __popcll(m_t); n_i = __popcll(m_i); n_p = __popcll(m_p);
        // (scalars, and to stay scalars: left alone the compiler decides what the next trip is with packed 16-bit VECTOR
        // multiplies.  Going further - the two decisions as integer max / min arithmetic the loop branches on - measured
        // slower, 924 -> 916 Msamples/s: the scalar unit of a CU is ~45 % busy in this kernel and a scalar instruction more
        // per trip costs more than a plain vector one, tools/microbench/pk_rate.hip and DESIGN.md 5.)
        if (kLeafPass) asm volatile("" : "+s"(n_t), "+s"(n_i));
        wait_debt += n_p;
    };
    // path logic is due / nothing can traverse (then every waiting lane is served)
    // (Tried: a wait debt of its own for the lanes whose shadow query has finished - the expensive kind of path logic - so
    // that each kind is served at a higher lane count: 809 -> 685-700 Msamples/s; lanes kept waiting are lanes that do not traverse.)
    const int wait_debt_bound = kWaitDebtFixed > 0 ? kWaitDebtFixed : (int)sc.wait_debt;
    auto path_logic_due = [&]() { return n_p > 0 && wait_debt >= wait_debt_bound; };
    auto nothing_traverses = [&]() { return n_t == 0 && n_i == 0; };
    // Loop nest: path logic in the outer loop, traversal trips in an inner loop of their own, so that the traversal
    // state is loop-carried through ONE small loop (as one if/else in one loop the two big branches were merged through
    // temporaries: 16 v_mov per trip).  Progress: a trip or a path-logic pass only runs with at least one lane wanting
    // it (the inner loop leaves as soon as nothing can traverse), so every pass advances some lane and the wave drains.
    for (;;) {
        survey();
        if (!any_lane) break;
        if (kLeafPass) {
            // node trips for the lanes at inner nodes; a leaf pass when enough lanes wait at leaves (or as many as run)
            // (node trips in a loop of their own: their state is carried through one small loop without copies)
            auto to_path_logic = [&]() { return path_logic_due() || nothing_traverses(); };
            auto pass_is_due = [&]() { return n_t >= kLeafLanes || (n_t > 0 && 3 * n_t >= kLeafRatio * n_i); };
            // (One loop for both kinds of trip.  Tried: node trips in an inner loop of their own - same schedule, 851 -> 807
            // Msamples/s; the node step of the lanes at inner nodes in the same trip as a pass, its record loads in flight
            // during the pass - 894 -> 791, the sixteen registers held across the pass spill; requesting the record of a lane's
            // next node as soon as the step has chosen it, before the wave has counted its lanes and decided what the next trip
            // is - 895 -> 724: the requests of lanes that turn out to wait are extra L1 traffic, and the loop-carried
            // registers cost sixteen copies per trip.)
            while (!to_path_logic()) {
                if (pass_is_due()) {
                    leaf_pass(pending);
                    survey();
                    continue;
                }
                if (STATS) { trips_i++; lanes_i += n_i; }
                if (want_inner) node_step();
                survey();
            }
        } else
        if (!(path_logic_due() || nothing_traverses())) {
            for (;;) {
                // ===================== traversal trip: EVERY traversing lane takes one step ====================
                // A DNode and a DTri are both one aligned 64-byte record, so node lanes and triangle lanes issue
                // the same four dwordx4 loads and the wave pays the memory latency once for both kinds.
                if (STATS) {
                    trips_i += n_i ? 1u : 0u; lanes_i += n_i;
                    trips_t += n_t ? 1u : 0u; lanes_t += n_t;
                }
                const bool is_tri = pending;
                if (pending || want_inner) {
                    // nodes and triangles live in one array of 64-byte records (nodes == tris): scalar base + 32-bit
                    // byte offset while the array is below 4 GB (the shift drops a node reference's flag bits)
                    // Every lane loads the two quads a triangle lane can reject with (see tri_test); node lanes load
                    // the other two as well, triangle lanes only if the distance tests pass.
                    const float4* rec;
                    if (!sc.wide_records) {
                        const uint32_t off = (is_tri ? tri_i : cur) << 6;
                        rec = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sc.tris) + off);
                    } else {
                        rec = reinterpret_cast<const float4*>(&sc.tris[is_tri ? tri_i : (cur & REF_INDEX_MASK_INNER)]);
                    }
                    constexpr int kE1 = PRE ? 1 : 3, kL0 = PRE ? 2 : 1, kL1 = PRE ? 3 : 2;
                    const float4 e0 = rec[0], e1 = rec[kE1];
                    bool need_pop = false;
                    if (is_tri) {
                        // ---- one triangle test (Triangle_Intersects inside the leaf loop, FullKernel.cl:638-646)
                        p_tri++;
                        tri_test<PRE>(e0, e1, [&](float4& l0, float4& l1) { l0 = rec[kL0]; l1 = rec[kL1]; }, r, limit,
                                      [&](const V4& q, float ray_t, float s, float t, bool front, float nsd) {
                            limit = nsd;
                            if (!shadow) {  // closest hit so far: the record path logic will shade from
                                if (kHitWords == 8) {
                                    hit_mem[0 * kWfBlock] = __float_as_uint(q.x); hit_mem[1 * kWfBlock] = __float_as_uint(q.y);
                                    hit_mem[2 * kWfBlock] = __float_as_uint(q.z); hit_mem[3 * kWfBlock] = __float_as_uint(q.w);
                                    hit_mem[6 * kWfBlock] = tri_i | (front ? kHitFront : 0u); hit_mem[7 * kWfBlock] = 2u;
                                } else {
                                    hit_mem[0 * kWfBlock] = __float_as_uint(ray_t);
                                    hit_mem[kWordTri * kWfBlock] = tri_i | (front ? kHitFront : 0u) | kHitFound;
                                }
                                hit_mem[kWordS * kWfBlock] = __float_as_uint(s); hit_mem[kWordT * kWfBlock] = __float_as_uint(t);
                            } else {
                                // any hit ends a shadow query (:724-727): empty the triangle range, drop the pending
                                // node.  Written as in-place moves (tied asm operands) so that these two registers
                                // are not merged back through the early exits with a select on every trip.
                                if (kHitWords == 8) hit_mem[7 * kWfBlock] = 2u;
                                else atomicOr(&hit_mem[kWordTri * kWfBlock], kHitFound);
                                asm volatile("v_mov_b32 %0, %1" : "+v"(tri_end) : "v"(tri_i));
                                asm volatile("v_mov_b32 %0, -1" : "+v"(cur));
                            }
                        });
                        tri_i++;
                    } else {
                        // ---- one inner-node step (:660-697)
                        const float4 l0 = rec[kL0], l1 = rec[kL1];
                        const float4 a = e0, b = PRE ? e1 : l0, c = PRE ? l0 : l1, d = PRE ? l1 : e1;
                        const float lo1[3] = {a.x, a.y, a.z}, hi1[3] = {a.w, b.x, b.y};
                        const float lo2[3] = {b.z, b.w, c.x}, hi2[3] = {c.y, c.z, c.w};
                        const uint32_t ref1 = __float_as_uint(d.x), ref2 = __float_as_uint(d.y), axis = __float_as_uint(d.z);
                        // dir[cutAxis] > 0 (:663) from the three sign masks the box tests need anyway: lane-mask logic
                        const bool a0 = axis == 0, a1 = axis == 1;
                        const bool fwd = (a0 & (r.d.x > 0)) | (a1 & (r.d.y > 0)) | (!(a0 | a1) & (r.d.z > 0));
                        bool h1, h2;
                        if (!wave_exact) {  // wave-uniform: nearly always
                            // (an empty child needs no flag test here: the upload stores it as an inverted infinite box)
                            h1 = box_hit_ordered(lo1, hi1, r, limit);
                            h2 = box_hit_ordered(lo2, hi2, r, limit);
                        } else {
                            h1 = box_hit(lo1, hi1, (ref1 & REF_EMPTY) != 0, r, limit);
                            h2 = box_hit(lo2, hi2, (ref2 & REF_EMPTY) != 0, r, limit);
                        }
                        p_bbx += 2;
                        // near child = fwd ? son1 : son2 (:663-666); descend into the near one if it was hit, else into the
                        // far one; push the far one when both were hit.  In terms of son1/son2:
                        const uint32_t far_ref = fwd ? ref2 : ref1;
                        const bool both = h1 & h2;
                        // push without a branch (see the LDS layout above)
                        sp[kWfBlock] = far_ref;
                        sp += both ? kWfBlock : 0;
                        cur = fwd ? (h1 ? ref1 : ref2) : (h2 ? ref2 : ref1);  // near child if it was hit, else the far one
                        need_pop = !(h1 | h2);
                    }
                    // ---- common tail of both step kinds, branch-free pops (an LDS read every lane can afford)
                    {
                        const uint32_t popped = *sp;
                        uint32_t* const below = sp - kWfBlock;
                        cur = need_pop ? popped : cur;
                        sp = need_pop ? (below < stack_floor ? stack_floor : below) : sp;
                    }
                    // the triangle range is free and the next node is a leaf: its triangles come next, and the node
                    // after them is whatever is pending on the stack  (kept as a branch: the select form measured -1 %)
                    if (tri_i >= tri_end && cur != REF_NONE && (cur & REF_LEAF)) {
                        decode_leaf(sc, cur, tri_i, tri_end);
                        cur = *sp;
                        uint32_t* const below = sp - kWfBlock;
                        sp = below < stack_floor ? stack_floor : below;
                    }
                }
                survey();
                if (path_logic_due() || nothing_traverses()) break;
            }
        }
        wait_debt = 0;
        // ================================ P: path logic ========================================
        if (STATS) { trips_p++; lanes_p += n_p; }
        const unsigned long long pass_start = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
        // (three divergent regions with the wave-uniform job hand-out between them: the queue state must stay scalar)
        // A shadow ray, a scattered ray and a camera ray all end in the same ray set-up (a normalisation and three divisions)
        // and the same start of a query: their sources leave the direction here and ONE copy at the end of the pass serves
        // the three (a lane has at most one new ray per pass).
        bool new_ray = false;
        V4 new_direction = v4(0, 0, 0, 0);
        if (want_post) {
            need_path = cur == REF_IDLE;
            alive = true;
            bool end_path = false, missed = false;
            bool start_shadow = false, do_scatter = false, lit = false;
            Hit hit;
            hit.point = v4(0, 0, 0, 0); hit.s = hit.t = 0; hit.tri = 0; hit.front = false;
            if (!need_path) {
                hit.point = load_hit_point();
                const bool found = query_found();
                if (!shadow) {
                    // closest-hit query finished (FullKernel.cl:1252-1288)
                    if (found) {
                        cam_d = r.d;
                        if (!kOneLight) {  // (one light: nothing is gathered across queries and the index is a constant)
                            direct = v4(0, 0, 0, 0);
                            light_idx = 0;
                        }
                        if (sc.n_lights > 0) start_shadow = true;
                        else do_scatter = true;
                    } else {
                        radiance = mad(sky_color(cold_scene().sky, sc.texels, r.d), transfer, radiance);  // cl:1287
                        end_path = missed = true;
                    }
                } else {
                    // shadow query finished (Scene_ComputeDirectIllumination, :944-947)
                    lit = !found;
                    if (!kOneLight && light_idx + 1u < sc.n_lights) start_shadow = true;
                    else do_scatter = true;
                }
                if (lit || do_scatter) {
                    // the surface of the closest hit (:1254-1274), rebuilt from the hit record and the arrival direction
                    Surface sf;
                    const uint32_t w6 = hit_mem[kWordTri * kWfBlock];
                    hit.s = __uint_as_float(hit_mem[kWordS * kWfBlock]); hit.t = __uint_as_float(hit_mem[kWordT * kWfBlock]);
                    hit.tri = w6 & REF_INDEX_MASK_LEAF; hit.front = (w6 & kHitFront) != 0;
                    Ray arrival;
                    arrival.o = hit.point; arrival.d = cam_d; arrival.ix = arrival.iy = arrival.iz = 0;
                    load_surface<PLAIN>(sc, arrival, hit, sf);
                    check(dot(cam_d, sf.Ns) < 0 && dot(cam_d, sf.Ng) < 0, C_CHK_NORMALS);  // cl:1275
                    V4 gathered = kOneLight ? v4(0, 0, 0, 0) : direct;
                    if (lit) {
                        ptmi_light light = sc.lights[kOneLight ? 0u : light_idx];
                        if (PLAIN) light.type = PTMI_LIGHT_POINT;
                        const float brdf = material_brdf(sf.mat.type, -r.d, sf.Ns, cam_d);
                        gathered = mad(v4(1, 1, 1, 1) * (light_power_toward(light, hit.point, sf.Ns) * brdf), v4(light.color), gathered);  // cl:945
                        if (!kOneLight) direct = gathered;
                        check(gathered.x >= 0 && gathered.y >= 0 && gathered.z >= 0, C_CHK_RADIANCE);  // cl:951
                    }
                    if (do_scatter) {
                        if (STATS && !PLAIN && !sf.mat.is_simple_color) atomicAdd(&block_counters[C_TEXTURED_HITS], 1ull);
                        r.d = cam_d;
                        V4 out;
                        V4 out_normal = v4(0, 0, 0, 0);
                        bool undefined_in_reference = false;
                        radiance = radiance + scatter_direction(r, seed, in_water, sf, gathered, transfer, out, STATS ? &out_normal : nullptr,
                                                                STATS ? &undefined_in_reference : nullptr);
                        check(dot(out, out_normal) > 0.0f, C_CHK_HEMISPHERE);  // header.cl:243
                        check(!undefined_in_reference, C_UNDEF_REFRACTION);      // (not a check of the reference: see ptmi_invariant_checks)
                        r.o = mad(out, 0.001f, hit.point);  // :880 uses the un-normalised direction
                        reflection++;
                        shadow = false;
                        if (!path_continues(transfer, reflection, seed, sc.russian_roulette != 0) || reflection >= sc.max_depth) {  // :1296-1314, :1248
                            end_path = true;
                        } else {
                            limit = INFINITY;
                            new_direction = out; new_ray = true;
                        }
                    }
                }
                if (!kOneLight && shadow) light_idx++;  // (still set: this pass finished a shadow query and did not scatter)
                if (start_shadow) {
                    // :932-944: ray from the hit point (no offset) towards light `light_idx`
                    const ptmi_light light = sc.lights[kOneLight ? 0u : light_idx];
                    const bool directional = !PLAIN && light.type == PTMI_LIGHT_DIRECTIONNAL;
                    const V4 full = directional ? -v4(light.direction) : v4(light.position) - hit.point;
                    r.o = hit.point;
                    limit = directional ? INFINITY : length(full);  // LINEAR distance in the squared slot
                    shadow = true;
                    new_direction = full; new_ray = true;
                }
                if (end_path) finish_path(missed);
            }
        }

        // ---- job hand-out: one atomic per wave for all lanes that ran dry -----------------
        // kQueues job queues, queue g = the g-th horizontal stripe of tiles; a wave starts on the queue of its
        // workgroup's XCD group (blockIdx % 8: which blocks share an XCD and its L2, not which XCD - a speed matter
        // only) and moves on to the next one when its queue is exhausted, so neighbouring tiles run on one L2.
        bool got_job = false;
        uint32_t gx = 0, gy = 0, it_local = 0;  // of the new job; live only until the path has started, below
        const bool want_job = want_post && need_path;
        const unsigned long long m_job = __ballot(want_job);
        if (m_job != 0ull) {
            const bool open = q_done < (uint32_t)kQueues;
            const uint32_t n_want = (uint32_t)__popcll(m_job);
            const uint32_t tile_lo = (uint32_t)(((unsigned long long)n_tiles * q_cur) / kQueues);
            const uint32_t tile_hi = (uint32_t)(((unsigned long long)n_tiles * (q_cur + 1u)) / kQueues);
            const uint32_t q_size = (tile_hi - tile_lo) * 64u * n_iterations;
            uint32_t base = 0;
            if (open) {
                const int leader = __ffsll((long long)m_job) - 1;
                if ((int)(tid & 63u) == leader) base = atomicAdd(&job_counter[q_cur * kQueueStride], n_want);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl(base, leader));
            }
            if (want_job) {
                const uint32_t rank = __popcll(m_job & ((1ull << (tid & 63u)) - 1ull));
                const uint32_t job = base + rank;
                if (!open) {
                    alive = false;  // every queue has been seen empty
                } else if (job < q_size) {
                    // tile-major: the iterations of one 8x8 tile are consecutive jobs, so the waves that take them (at
                    // about the same time) send their camera rays and first shadow rays through the same part of the tree
                    const uint32_t unit = job >> 6, in_tile = job & 63u;
                    const uint32_t tile_in_q = unit / n_iterations;
                    it_local = unit - tile_in_q * n_iterations;
                    const uint32_t tile = tile_lo + tile_in_q;
                    gx = (tile % tiles_x) * 8u + (in_tile & 7u);
                    gy = (tile / tiles_x) * 8u + (in_tile >> 3);
                    got_job = gx < sc.width && gy < sc.height;  // edge tiles: pixel outside the image, ask again
                }  // else: the queue ran out under this wave; the lane asks again in the next pass
            }
            if (open && base + n_want > q_size) {  // wave-uniform: this queue is (now) exhausted
                q_cur = (q_cur + 1u) % (uint32_t)kQueues;
                q_done++;
            }
        }

        if (want_post) {
            // ---- start the next camera path of this pixel (FullKernel.cl:1208-1215) ------------
            if (got_job) {
                const uint32_t it = first_iteration + it_local * iteration_stride;
                slot = it_local * (sc.width * sc.height) + gy * sc.width + gx;
                seed = lcg_seed(gx, gy, sc.width, sc.height, it, sc.source_seed != 0);
                float sample_x, sample_y;
                draw_sample(sc, gx, gy, it, seed, sample_x, sample_y);
                check(sample_x >= -0.5f && sample_y >= -0.5f && sample_x <= 0.5f && sample_y <= 0.5f, C_CHK_SAMPLE);  // cl:1217
                {
                    const DScene& cs = cold_scene();  // camera: only needed here, once per path
                    r.o = v4(cs.cam_pos);
                    new_direction = mad(v4(cs.cam_up), sample_y, mad(v4(cs.cam_right), sample_x, v4(cs.cam_dir)));  // cl:1213
                }
                radiance = v4(0, 0, 0, 0);
                transfer = v4(1, 1, 1, 1);
                reflection = 0; p_bbx = 0; p_tri = 0;
                in_water = false;
                shadow = false;
                need_path = false;
                bool skip = false;
                if (SS && it > 5u) {
                    // superSamplingStopCriteria, FullKernel.cl:1152-1172 (called at :1219-1222, one launch per
                    // iteration so the accumulators hold iterations < it); draws one random number
                    const DScene& cs = cold_scene();
                    // the pixel the SAMPLE falls on (:1159-1161): the work-item's own with JITTERED / UNIFORM, any with RANDOM
                    const uint32_t off = owns_pixel ? gy * sc.width + gx : sample_pixel(sc, sample_x, sample_y);
                    const float n = cs.image_ray_nb[off];
                    const float4 vv = reinterpret_cast<const float4*>(cs.image_v)[off];
                    const float sigma2_n = fmaxf(fmaxf(fdiv(vv.x, n), fdiv(vv.y, n)), fdiv(vv.z, n));
                    uint32_t idx = (uint32_t)n;
                    if (idx > 1000u) idx = 1000u;  // the reference indexes past its 1001-entry table here
                    skip = (double)lcg_random(seed) > (double)fdiv(100 * sigma2_n, cs.x2inv[idx]) + 0.05;
                }
                if (skip) {
                    if (owns_pixel) cold_scene().stage_flag[gy * sc.width + gx] = 0.f;  // returns before statistics and accumulation
                    need_path = true;
                } else
                if (sc.max_depth > 0) {
                    limit = INFINITY;
                    new_ray = true;
                } else {
                    finish_path(false);  // depth 0: the bounce loop never runs (:1248), radiance 0, depth bin 0
                }
            }
            if (new_ray) {
                ray_set_direction(r, new_direction);
                dir_signs = (r.d.x > 0 ? 1u : 0u) | (r.d.y > 0 ? 2u : 0u) | (r.d.z > 0 ? 4u : 0u);
                // A ray that is not a number - a refraction at |cos| = 1 + 1 ulp takes the square root of a negative (cl:235),
                // a hit on a fake plane 1e30 away overflows - makes every triangle test compute a NaN distance, and the
                // reference ACCEPTS those (its rejections are comparisons, cl:533-567): from then on nothing is "too far" and
                // the LAST triangle that passes wins - not a minimum over ordered keys, which is what a leaf pass keeps.
                // With finite rays from origins below 2^40 and the records the upload admits (scene_layout.cpp:
                // scene_needs_literal_kernel) a distance is always a number.  So such a path is GIVEN UP here: it restarts
                // its counters, takes a marked NaN for its radiance and its query ends before it starts - as a miss, so the
                // next pass finishes the path the ordinary way (one segment, no hit) - and redo_poisoned_kernel, behind the
                // launch, traces it again with the reference's literal loops.  (RANDOM sampler: nothing is staged; the path
                // goes to the launch's give-up list instead and redo_random_kernel adds it, see kGivenUp.)
                const float o_size = __builtin_fabsf(r.o.x) + __builtin_fabsf(r.o.y) + __builtin_fabsf(r.o.z) + __builtin_fabsf(r.o.w);
                const float d_size = __builtin_fabsf(r.d.x) + __builtin_fabsf(r.d.y) + __builtin_fabsf(r.d.z) + __builtin_fabsf(r.d.w);
                bool bad = !((o_size <= 0x1p+40f) & (d_size <= 4.0f));  // (false for a NaN)
                if (!PLAIN && !owns_pixel && __builtin_expect(bad, 0)) {
                    // RANDOM sampler: a place in the launch's give-up list, or (list full) carry on as before round 4
                    const uint32_t place = atomicAdd(&job_counter[2], 1u);
                    bad = place < kGiveUpListCap;
                    if (bad) {
                        job_counter[kGiveUpListFirst + place] = slot;
                        slot |= kGivenUp;
                    }
                }
                if (__builtin_expect(bad, 0)) {
                    const float m = __uint_as_float(kPoisonMarker);
                    radiance = v4(m, m, m, m);
                    transfer = v4(1, 1, 1, 1);
                    reflection = 0; p_bbx = 0; p_tri = 0;
                    shadow = false;
                    job_counter[1] = 1u;  // (the block is zeroed before every launch)
                }
                start_query();
                if (__builtin_expect(bad, 0)) { cur = REF_NONE; tri_i = tri_end = 0; }
            }
            if (need_path) cur = alive ? REF_IDLE : REF_DEAD;
        }
        dead_lanes = __builtin_amdgcn_ballot_w64(cur == REF_DEAD);
        wave_exact = __builtin_amdgcn_ballot_w64(exact_boxes) != 0ull;  // finished lanes keep a stale flag: conservative
        if (STATS) cycles_p += __builtin_amdgcn_s_memtime() - pass_start;
    }

    if (STATS && (tid & 63u) == 0) {  // one lane per wave: the scheduler counters are wave-uniform
        atomicAdd(&block_counters[C_TRIPS_I], (unsigned long long)trips_i);
        atomicAdd(&block_counters[C_LANES_I], lanes_i);
        atomicAdd(&block_counters[C_TRIPS_T], (unsigned long long)(kLeafPass ? pass_rounds : trips_t));
        atomicAdd(&block_counters[C_LANES_T], kLeafPass ? (unsigned long long)pass_items : lanes_t);
        atomicAdd(&block_counters[C_TRIPS_P], (unsigned long long)trips_p);
        atomicAdd(&block_counters[C_LANES_P], lanes_p);
        atomicAdd(&block_counters[C_CYCLES_P], cycles_p);
        atomicAdd(&block_counters[C_CYCLES_LOOP], __builtin_amdgcn_s_memtime() - loop_start);
    }
    if (STATS && item_violations != 0u) atomicAdd(&block_counters[C_ITEM_VIOLATIONS], (unsigned long long)item_violations);
    __syncthreads();
    // every surface hit sends one shadow ray to every light (Scene_ComputeDirectIllumination, :901-954)
    if (tid < (STATS ? 1u : (uint32_t)PTMI_COUNTER_SPLITS)) {
        unsigned long long* const totals = &block_counters[tid * kSplitWords];
        totals[C_SHADOW] = totals[C_HITS] * sc.n_lights;
    }
    __syncthreads();
    if (STATS) {
        if (tid < C_COUNT) atomicAdd(&cold_scene().counters[tid], block_counters[tid]);
    } else if (tid < kBlockCounters) {
        const uint32_t k = tid / kSplitWords, c = tid % kSplitWords;
        if (k == 0u || block_counters[tid] != 0ull) atomicAdd(&cold_scene().counters[k * C_COUNT + c], block_counters[tid]);
    }
}

// The paths a wavefront launch gave up (marked radiance in the staging slot, counted as one segment without a hit), traced
// again by the reference's loops as they are written (ptmi_literal_path.hpp): radiance, statistics word and totals as if the
// launch had traced them.  Runs behind every staged wavefront launch on its stream and returns at once unless the launch
// left a non-zero word 1 in its job-counter block.
// The marked slots are few and scattered (a handful per launch in a clean scene; a few per cent where rays meet a record that
// yields NaN distances), so a wave does not trace the slots it scans: it COLLECTS marked slot numbers in an LDS queue of its
// own while it scans (64 slots per step, chunks dealt out to the waves of the grid in turn) and traces them 64 at a time, all
// lanes busy - round 4; traced where they were found, one lane in fifteen worked and a scene with 7 % of its paths marked ran
// at the one-path-per-lane kernel's pace.
template <bool PRE>
__global__ void __launch_bounds__(kBlock) redo_poisoned_kernel(const DScene sc, const uint32_t first_iteration, const uint32_t n_iterations,
                                                               const uint32_t iteration_stride, float* __restrict__ stage,
                                                               uint32_t* __restrict__ stage_stats, const uint32_t* __restrict__ job_counter)
{
    if (__builtin_nontemporal_load(&job_counter[1]) == 0u) return;  // (the same for every lane of the grid)
    __shared__ uint32_t stack_mem[kStackDepth * kBlock];
    // (two's complement: the one segment the launch counted is taken back; one block per call the launch rendered for)
    __shared__ unsigned long long block_counters[PTMI_COUNTER_SPLITS * kSplitWords];
    __shared__ uint32_t queues[kBlock / 64][128];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    if (tid < PTMI_COUNTER_SPLITS * kSplitWords) block_counters[tid] = 0;
    __syncthreads();
    const uint32_t n_pixels = sc.width * sc.height, n_slots = n_pixels * n_iterations;
    auto trace_slot = [&](const uint32_t slot) {
        const uint32_t it_local = slot / n_pixels, pixel = slot - it_local * n_pixels;
        const uint32_t gy = pixel / sc.width, gx = pixel - gy * sc.width;
        const uint32_t it = first_iteration + it_local * iteration_stride;
        float sx, sy;
        uint32_t depth = 0, n_seg = 0, n_shadow = 0;
        PathCounters pc;
        const V4 radiance = trace_path<PRE>(sc, gx, gy, it, &stack_mem[tid], sx, sy, depth, n_seg, n_shadow, pc, sc.super_sampling != 0 && it > 5u);
        reinterpret_cast<float4*>(stage)[slot] = make_float4(radiance.x, radiance.y, radiance.z, radiance.w);
        if (stage_stats != nullptr) {
            stage_stats[slot] = pack_path_statistics(depth, pc.bbx, pc.tri);
        } else if (sc.hist_depths) {  // FullKernel.cl:1319-1331; the launch counted the path in the three zero bins
            atomicSub(&sc.hist_depths[0], 1u); atomicSub(&sc.hist_bbx[0], 1u); atomicSub(&sc.hist_tri[0], 1u);
            atomicAdd(&sc.hist_depths[depth], 1u);
            if (pc.bbx < PTMI_MAX_INTERSECTION_NUMBER) atomicAdd(&sc.hist_bbx[pc.bbx], 1u);
            if (pc.tri < PTMI_MAX_INTERSECTION_NUMBER) atomicAdd(&sc.hist_tri[pc.tri], 1u);
        }
        // (the launch counted: one path, one segment, no hit)
        unsigned long long* const totals = &block_counters[counter_split_of(slot, sc.split_paths) * kSplitWords];
        atomicAdd(&totals[C_PATHS], 1ull);  // (here: paths traced again)
        atomicAdd(&totals[C_SEGMENTS], (unsigned long long)n_seg - 1ull);
        atomicAdd(&totals[C_HITS], (unsigned long long)depth);
        atomicAdd(&totals[C_SHADOW], (unsigned long long)n_shadow);
        atomicAdd(&totals[C_BBX], (unsigned long long)pc.bbx);
        atomicAdd(&totals[C_TRI], (unsigned long long)pc.tri);
    };
    uint32_t* const queue = queues[tid >> 6];
    uint32_t queued = 0;  // wave-uniform: slots waiting in this wave's queue (< 64 between two steps)
    const uint32_t n_waves = gridDim.x * (kBlock / 64), n_chunks = (n_slots + 63u) / 64u;
    for (uint32_t chunk = blockIdx.x * (kBlock / 64) + (tid >> 6); chunk < n_chunks; chunk += n_waves) {
        const uint32_t slot = chunk * 64u + lane;
        bool marked = false;
        if (slot < n_slots) {
            const uint4 v = reinterpret_cast<const uint4*>(stage)[slot];
            marked = v.x == kPoisonMarker && v.y == kPoisonMarker && v.z == kPoisonMarker && v.w == kPoisonMarker;
            // ... and the path counted nothing (a given-up path restarts its counters): a radiance that merely CARRIES the
            // marker's bits - NaN payloads a caller put into a light or a material colour - belongs to a path that has made at
            // least one box or triangle test
            if (marked && stage_stats != nullptr) marked = stage_stats[slot] == pack_path_statistics(0u, 0u, 0u);
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(marked);
        if (m == 0ull) continue;
        if (marked) queue[queued + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = slot;
        queued += (uint32_t)__popcll(m);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // (the queue is read by other lanes of the wave than wrote it)
        if (queued >= 64u) {
            queued -= 64u;
            const uint32_t mine = queue[queued + lane];
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            trace_slot(mine);
        }
    }
    if (lane < queued) trace_slot(queue[lane]);
    __syncthreads();
    if (tid < PTMI_COUNTER_SPLITS * kSplitWords && block_counters[tid] != 0ull) {
        const uint32_t k = tid / kSplitWords, c = tid % kSplitWords;
        if (c > C_PATHS && c <= C_TRI) atomicAdd(&sc.counters[k * C_COUNT + c], block_counters[tid]);
        if (c == C_PATHS) atomicAdd(&sc.counters[k * C_COUNT + C_RETRACED], block_counters[tid]);
    }
}

// The RANDOM sampler's form of the above: the paths a launch gave up are LISTED in its job-counter block (kGivenUp); traced
// again by the literal loops, each adds its sample where finish_path would have - atomically, on the pixel its sample position
// falls on (FullKernel.cl:1333-1349) - its three histogram bins and its counts.
template <bool PRE>
__global__ void __launch_bounds__(kBlock) redo_random_kernel(const DScene sc, const uint32_t first_iteration, const uint32_t n_iterations,
                                                             const uint32_t iteration_stride, const uint32_t* __restrict__ job_counter)
{
    if (__builtin_nontemporal_load(&job_counter[1]) == 0u) return;
    __shared__ uint32_t stack_mem[kStackDepth * kBlock];
    __shared__ unsigned long long block_counters[C_TRI + 1];
    const uint32_t tid = threadIdx.x;
    if (tid <= C_TRI) block_counters[tid] = 0;
    __syncthreads();
    const uint32_t listed = job_counter[2], n = listed < kGiveUpListCap ? listed : kGiveUpListCap;
    const uint32_t n_pixels = sc.width * sc.height;
    for (uint32_t i = blockIdx.x * kBlock + tid; i < n; i += gridDim.x * kBlock) {
        const uint32_t slot = job_counter[kGiveUpListFirst + i];
        const uint32_t it_local = slot / n_pixels, pixel = slot - it_local * n_pixels;
        const uint32_t gy = pixel / sc.width, gx = pixel - gy * sc.width;
        const uint32_t it = first_iteration + it_local * iteration_stride;
        float sx, sy;
        uint32_t depth = 0, n_seg = 0, n_shadow = 0;
        PathCounters pc;
        const bool ss = sc.super_sampling != 0;
        const V4 radiance = trace_path<PRE>(sc, gx, gy, it, &stack_mem[tid], sx, sy, depth, n_seg, n_shadow, pc, ss && it > 5u);
        if (sc.hist_depths) {  // FullKernel.cl:1319-1331
            atomicAdd(&sc.hist_depths[depth], 1u);
            if (pc.bbx < PTMI_MAX_INTERSECTION_NUMBER) atomicAdd(&sc.hist_bbx[pc.bbx], 1u);
            if (pc.tri < PTMI_MAX_INTERSECTION_NUMBER) atomicAdd(&sc.hist_tri[pc.tri], 1u);
        }
        const uint32_t off = sample_pixel(sc, sx, sy);
        const V4 before = v4(atomicAdd(&sc.image_color[4 * off + 0], radiance.x), atomicAdd(&sc.image_color[4 * off + 1], radiance.y),
                             atomicAdd(&sc.image_color[4 * off + 2], radiance.z), atomicAdd(&sc.image_color[4 * off + 3], radiance.w));
        const float n_before = atomicAdd(&sc.image_ray_nb[off], 1.f);
        if (ss && it != 0u) {  // cl:1346-1349, as finish_path's RANDOM branch
            float* const vp = &sc.image_v[4 * off];
            const V4 after = before + radiance;
            const float n_after = n_before + 1.f;
            atomicAdd(&vp[0], (radiance.x - fdiv(before.x, n_before)) * (radiance.x - fdiv(after.x, n_after)));
            atomicAdd(&vp[1], (radiance.y - fdiv(before.y, n_before)) * (radiance.y - fdiv(after.y, n_after)));
            atomicAdd(&vp[2], (radiance.z - fdiv(before.z, n_before)) * (radiance.z - fdiv(after.z, n_after)));
            atomicAdd(&vp[3], (radiance.w - fdiv(before.w, n_before)) * (radiance.w - fdiv(after.w, n_after)));
        }
        // (the launch counted: one path, one segment, no hit)
        atomicAdd(&block_counters[C_PATHS], 1ull);
        atomicAdd(&block_counters[C_SEGMENTS], (unsigned long long)n_seg - 1ull);
        atomicAdd(&block_counters[C_HITS], (unsigned long long)depth);
        atomicAdd(&block_counters[C_SHADOW], (unsigned long long)n_shadow);
        atomicAdd(&block_counters[C_BBX], (unsigned long long)pc.bbx);
        atomicAdd(&block_counters[C_TRI], (unsigned long long)pc.tri);
    }
    __syncthreads();
    if (tid > C_PATHS && tid <= C_TRI && block_counters[tid] != 0ull) atomicAdd(&sc.counters[tid], block_counters[tid]);
    if (tid == C_PATHS && block_counters[tid] != 0ull) atomicAdd(&sc.counters[C_RETRACED], block_counters[tid]);
}

// Histograms of the paths of one launch from their staged statistics words (FullKernel.cl:1319-1331: depth bin
// always, box / triangle bins only below MAX_INTERSECTION_NUMBER): counted in LDS per workgroup, then one global
// atomic per non-empty bin.  `stage_flag` (SUPER_SAMPLING): 0 = the path was skipped before the statistics.
__global__ void __launch_bounds__(1024) histogram_staged_kernel(uint32_t* __restrict__ hist_depths, uint32_t* __restrict__ hist_bbx,
                                                                uint32_t* __restrict__ hist_tri,
                                                                const uint32_t* __restrict__ stats,
                                                                const float* __restrict__ stage_flag, const uint32_t n_slots)
{
    __shared__ uint32_t bins[kStatDepthBins + 2 * PTMI_MAX_INTERSECTION_NUMBER];
    for (uint32_t i = threadIdx.x; i < kStatDepthBins + 2 * PTMI_MAX_INTERSECTION_NUMBER; i += blockDim.x) bins[i] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += gridDim.x * blockDim.x) {
        if (stage_flag != nullptr && stage_flag[i] == 0.f) continue;
        const uint32_t w = stats[i];
        const uint32_t depth = w & (kStatDepthBins - 1u), bbx = (w >> 6) & 0x1FFFu, tri = w >> 19;
        atomicAdd(&bins[depth], 1u);
        if (bbx < PTMI_MAX_INTERSECTION_NUMBER) atomicAdd(&bins[kStatDepthBins + bbx], 1u);
        if (tri < PTMI_MAX_INTERSECTION_NUMBER) atomicAdd(&bins[kStatDepthBins + PTMI_MAX_INTERSECTION_NUMBER + tri], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kStatDepthBins + 2 * PTMI_MAX_INTERSECTION_NUMBER; i += blockDim.x) {
        const uint32_t n = bins[i];
        if (n == 0) continue;
        if (i < kStatDepthBins) atomicAdd(&hist_depths[i], n);
        else if (i < kStatDepthBins + PTMI_MAX_INTERSECTION_NUMBER) atomicAdd(&hist_bbx[i - kStatDepthBins], n);
        else atomicAdd(&hist_tri[i - kStatDepthBins - PTMI_MAX_INTERSECTION_NUMBER], n);
    }
}

// Adds the staged radiances of one launch to the accumulators, per pixel in iteration order:
// sumAfter = sumBefore + radiance; nRayAfter = nRayBefore + 1 (FullKernel.cl:1339-1345), n_iterations times.
__global__ void __launch_bounds__(256) accumulate_staged_kernel(float* __restrict__ image_color,
                                                                 float* __restrict__ image_ray_nb,
                                                                 const float* __restrict__ stage, const uint32_t n_pixels,
                                                                 const uint32_t n_iterations)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pixels) return;
    float4 sum = reinterpret_cast<const float4*>(image_color)[p];
    float count = image_ray_nb[p];
    for (uint32_t k = 0; k < n_iterations; k++) {
        const float4 r = reinterpret_cast<const float4*>(stage)[(size_t)k * n_pixels + p];
        sum.x = sum.x + r.x; sum.y = sum.y + r.y; sum.z = sum.z + r.z; sum.w = sum.w + r.w;
        count = count + 1.f;
    }
    reinterpret_cast<float4*>(image_color)[p] = sum;
    image_ray_nb[p] = count;
}

// SUPER_SAMPLING form of the accumulation (one iteration per launch): pixels whose path was skipped are left
// alone; the others also update the variance accumulator imageV exactly as FullKernel.cl:1346-1349.
__global__ void __launch_bounds__(256) accumulate_staged_ss_kernel(float* __restrict__ image_color,
                                                                    float* __restrict__ image_ray_nb,
                                                                    float* __restrict__ image_v,
                                                                    const float* __restrict__ stage,
                                                                    const float* __restrict__ stage_flag,
                                                                    const uint32_t n_pixels, const uint32_t iteration)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pixels) return;
    if (stage_flag[p] == 0.f) return;
    const float4 before = reinterpret_cast<const float4*>(image_color)[p];
    const float4 r = reinterpret_cast<const float4*>(stage)[p];
    const float4 after = make_float4(before.x + r.x, before.y + r.y, before.z + r.z, before.w + r.w);
    const float n_before = image_ray_nb[p];
    const float n_after = n_before + 1.f;
    image_ray_nb[p] = n_after;
    reinterpret_cast<float4*>(image_color)[p] = after;
    float4 v = make_float4(0, 0, 0, 0);
    // n_before == 0 with iteration != 0: the first sample THIS context adds to the pixel (a shard of a multi-device
    // render that does not start at iteration 0, or a cleared context).  The reference would divide 0 by 0 there and keep
    // a NaN variance for good; a first sample has no deviation, exactly like iteration 0.
    if (iteration != 0 && n_before != 0.f) {
        v = reinterpret_cast<const float4*>(image_v)[p];
        v.x = mad(r.x - fdiv(before.x, n_before), r.x - fdiv(after.x, n_after), v.x);  // cl:1349
        v.y = mad(r.y - fdiv(before.y, n_before), r.y - fdiv(after.y, n_after), v.y);
        v.z = mad(r.z - fdiv(before.z, n_before), r.z - fdiv(after.z, n_after), v.z);
        v.w = mad(r.w - fdiv(before.w, n_before), r.w - fdiv(after.w, n_after), v.w);
    }
    reinterpret_cast<float4*>(image_v)[p] = v;
}

#if PTMI_DEFAULT_ARITHMETIC
// The reciprocal determinant of a DTriPre record as the reference's default build computes it per test (FullKernel.cl:556:
// fused uv*uv - uu*vv, then the reciprocal through v_rcp_f32): written once per upload, on the device because the
// instruction's value is not reproducible on the host.  Node records (tri_ids == 0xFFFFFFFF) are left alone.
__global__ void __launch_bounds__(256) precompute_denominators_kernel(DTri* __restrict__ records, const uint32_t* __restrict__ tri_ids,
                                                                       const uint32_t n_records)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_records || tri_ids[i] == 0xFFFFFFFFu) return;
    DTriPre* const p = reinterpret_cast<DTriPre*>(&records[i]);
    const V4 u = v4(p->u_den[0], p->u_den[1], p->u_den[2], 0.0f), v = v4(p->v_s1w[0], p->v_s1w[1], p->v_s1w[2], 0.0f);
    const float uv = dot(u, v), uu = dot(u, u), vv = dot(v, v);
    p->u_den[3] = frcp(mad(uv, uv, -(uu * vv)));
}
#endif

}  // namespace PTMI_DEV_NS

namespace ptmi_internal {

#if PTMI_DEFAULT_ARITHMETIC
int launch_precompute_denominators_da(DTri* records, const uint32_t* tri_ids, uint32_t n_records, void* stream, std::string* err)
{
    if (n_records == 0) return PTMI_OK;
    hipLaunchKernelGGL(PTMI_DEV_NS::precompute_denominators_kernel, dim3((n_records + 255u) / 256u), dim3(256), 0, (hipStream_t)stream,
                       records, tri_ids, n_records);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (err) *err = std::string("precompute_denominators_kernel launch: ") + hipGetErrorString(e);
        return PTMI_ERR_HIP;
    }
    return PTMI_OK;
}
#endif

static uint32_t clamp_levels(uint32_t stack_levels)
{
    if (stack_levels < 1) stack_levels = 1;
    if (stack_levels > (uint32_t)PTMI_DEV_NS::kWfStack) stack_levels = PTMI_DEV_NS::kWfStack;
    return stack_levels;
}

static size_t wavefront_lds_bytes(uint32_t stack_levels, int block)
{
    stack_levels = clamp_levels(stack_levels);
    // closest-hit record + sentinel + stack (+ the keys and items of the leaf passes), per lane of the workgroup
    return (size_t)(stack_levels + 1 + PTMI_DEV_NS::kHitWords + PTMI_DEV_NS::kLeafPassWordsPerLane) * block * sizeof(uint32_t);
}

// workgroups of instantiation `kernel` the current device holds at once (the persistent grid): registers and LDS decide
template <class Kernel>
static int resident_blocks_of(Kernel kernel, uint32_t stack_levels, int block)
{
    int device = 0, per_cu = 0, n_cu = 0;
    if (hipGetDevice(&device) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, wavefront_lds_bytes(stack_levels, block)) != hipSuccess)
        return 0;
    if (per_cu < 1) per_cu = 1;
    return per_cu * n_cu;
}

// the grid of the most recent wavefront launch per device: lanes per workgroup << 20 | workgroups the device holds at once
// (ptmi_scheduler_stats.workgroup_lanes / resident_workgroups: what a test of "five wide workgroups per CU" reads)
static std::atomic<uint32_t> g_last_grid[64];
void PTMI_ARITH(last_wavefront_grid)(int device, uint32_t* lanes, uint32_t* resident)
{
    const uint32_t v = device >= 0 && device < 64 ? g_last_grid[device].load(std::memory_order_relaxed) : 0u;
    *lanes = v >> 20;
    *resident = v & 0xFFFFFu;
}

int PTMI_ARITH(launch_render_wavefront)(const DScene& sc, const DScene* scene_in_device_memory, uint32_t first_iteration,
                            uint32_t n_iterations, uint32_t iteration_stride, uint32_t* job_counter, uint32_t stack_levels,
                            bool scheduler_stats, float* stage, uint32_t* stage_stats, void* stream, std::string* err)
{
    if (n_iterations == 0) return PTMI_OK;
    const uint32_t tiles = ((sc.width + 7u) / 8u) * ((sc.height + 7u) / 8u);
    const uint64_t jobs64 = (uint64_t)tiles * 64u * n_iterations;
    if (jobs64 > (sc.sampler == PTMI_SAMPLER_RANDOM ? 0x7FFFFFF0ull : 0xFFFFFFF0ull)) {  // (RANDOM: bit 31 of a slot word is kGivenUp)
        if (err) *err = "too many jobs in one launch";
        return PTMI_ERR_INVALID_ARGUMENT;
    }
    const uint32_t n_jobs = (uint32_t)jobs64;
    hipError_t e = hipMemsetAsync(job_counter, 0, PTMI_DEV_NS::kQueues * PTMI_DEV_NS::kQueueStride * sizeof(uint32_t), (hipStream_t)stream);
    if (e == hipSuccess && sc.sampler == PTMI_SAMPLER_RANDOM) {
        // PTMI_RANDOM_GIVE_UP=0 (tests, A/B): the give-up list starts full, so a path whose ray is not a number carries on with
        // the ordered minimum as before round 4
        const char* off = std::getenv("PTMI_RANDOM_GIVE_UP");
        if (off && off[0] == '0')
            e = hipMemsetD32Async((hipDeviceptr_t)(job_counter + 2), (int)PTMI_DEV_NS::kGiveUpListCap, 1, (hipStream_t)stream);
    }
    if (e == hipSuccess) {
        const uint32_t lv = clamp_levels(stack_levels);
        hipStream_t st = (hipStream_t)stream;
        PTMI_DEV_NS::DWarm warm{};
        warm.nodes = sc.nodes; warm.tris = sc.tris; warm.big_leaves = sc.big_leaves; warm.tri_ids = sc.tri_ids; warm.shade = sc.shade;
        warm.mats = sc.mats; warm.lights = sc.lights; warm.textures = sc.textures; warm.texels = sc.texels;
        warm.root_ref = sc.root_ref; warm.width = sc.width; warm.height = sc.height; warm.max_depth = sc.max_depth;
        warm.n_lights = sc.n_lights; warm.sampler = sc.sampler; warm.tris_precomputed = sc.tris_precomputed;
        warm.histograms = sc.hist_depths != nullptr;
        warm.boxes_ordered = sc.boxes_ordered;
        warm.wide_records = sc.wide_records;
        warm.russian_roulette = sc.russian_roulette;
        warm.source_seed = sc.source_seed;
        warm.split_paths = sc.split_paths;
        constexpr int kMaxCachedDevices = 64;
        int device = 0;
        const bool cached_device = hipGetDevice(&device) == hipSuccess && device >= 0 && device < kMaxCachedDevices;
        const bool plain = sc.tris_precomputed && sc.plain_shading && sc.sampler == PTMI_SAMPLER_JITTERED && !sc.russian_roulette &&
                           sc.n_lights == 1 && !sc.super_sampling && !scheduler_stats;
        warm.wait_debt = lv >= 16u ? 768u : (plain ? 320u : 512u);  // (the cheaper a path-logic pass, the sooner it pays)
        // the persistent grid of the chosen instantiation on the CURRENT device (instantiations differ in registers, devices in
        // CUs and partition mode, hence in workgroups held at once): asked once per (instantiation, device, stack levels);
        // host threads that drive contexts of their own may race for an entry, and then write the same value
#define PTMI_LAUNCH_WF_BLOCK(S, P, A, L, N, B)                                                                             \
    do {                                                                                                                  \
        auto kernel = PTMI_DEV_NS::render_wavefront_kernel<S, P, A, L, N, B>;                                              \
        static std::atomic<int> resident_cache[kMaxCachedDevices][PTMI_BVH_MAX_DEPTH + 1];                                 \
        int resident = cached_device ? resident_cache[device][lv].load(std::memory_order_relaxed) : 0;                     \
        if (resident == 0) {                                                                                               \
            resident = resident_blocks_of(kernel, lv, B);                                                                  \
            if (cached_device) resident_cache[device][lv].store(resident, std::memory_order_relaxed);                      \
        }                                                                                                                  \
        if (cached_device) g_last_grid[device].store((uint32_t)(B) << 20 | (uint32_t)resident, std::memory_order_relaxed); \
        uint32_t nb = (n_jobs + (B) - 1) / (B);                                                                           \
        if (resident > 0 && nb > (uint32_t)resident) nb = (uint32_t)resident;                                             \
        hipLaunchKernelGGL(kernel, dim3(nb), dim3(B), wavefront_lds_bytes(lv, B), st, scene_in_device_memory, warm,        \
                           first_iteration, n_iterations, iteration_stride, n_jobs, job_counter, lv, stage, stage_stats); \
    } while (0)
#define PTMI_LAUNCH_WF_IMPL(S, P, A, L, N) PTMI_LAUNCH_WF_BLOCK(S, P, A, L, N, PTMI_DEV_NS::kWfBlock)
#define PTMI_LAUNCH_WF(S, P, A, N) PTMI_LAUNCH_WF_IMPL(S, P, A, false, N)
        // Deep trees (23 levels and more): the wide workgroup's LDS no longer fits five times into a CU; the two production
        // instantiations are then launched in their narrow form (kWfBlockNarrow).  Asked once per device and depth.
        static std::atomic<int> wide_fits[kMaxCachedDevices][PTMI_BVH_MAX_DEPTH + 1];  // 0 = not asked, 1 = five wide workgroups fit, 2 = they do not
        int fits = cached_device ? wide_fits[device][lv].load(std::memory_order_relaxed) : 0;
        if (fits == 0) {
            int n_cu = 0;
            const int wide = resident_blocks_of(PTMI_DEV_NS::render_wavefront_kernel<false, true, false, true, false>, lv, PTMI_DEV_NS::kWfBlock);
            const bool known = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n_cu > 0 && wide > 0;
            fits = known && wide < 5 * n_cu ? 2 : 1;
            if (cached_device) wide_fits[device][lv].store(fits, std::memory_order_relaxed);
        }
        const bool narrow = fits == 2 && std::getenv("PTMI_WIDE_WORKGROUPS") == nullptr;  // (developer switch: A/B runs)
        // instantiations: the common case (no statistics, no adaptive sampling, records that cannot yield NaN distances) pays
        // for none of them; the statistics and SUPER_SAMPLING builds always carry the NaN check (two instructions per
        // accepted triangle)
        if (sc.super_sampling) {
            if (sc.tris_precomputed) PTMI_LAUNCH_WF(true, true, true, true); else PTMI_LAUNCH_WF(true, false, true, true);
        } else if (scheduler_stats) {
            if (sc.tris_precomputed) PTMI_LAUNCH_WF(true, true, false, true); else PTMI_LAUNCH_WF(true, false, false, true);
        } else if (sc.nan_safe) {
            if (plain) PTMI_LAUNCH_WF_IMPL(false, true, false, true, true);
            else if (sc.tris_precomputed) PTMI_LAUNCH_WF(false, true, false, true);
            else PTMI_LAUNCH_WF(false, false, false, true);
        } else {
            if (plain) {  // the common case, BASELINE's untextured scenes among them
                if (narrow) PTMI_LAUNCH_WF_BLOCK(false, true, false, true, false, PTMI_DEV_NS::kWfBlockNarrow);
                else PTMI_LAUNCH_WF_IMPL(false, true, false, true, false);
            } else if (sc.tris_precomputed) {
                if (narrow) PTMI_LAUNCH_WF_BLOCK(false, true, false, false, false, PTMI_DEV_NS::kWfBlockNarrow);
                else PTMI_LAUNCH_WF(false, true, false, false);
            }
            else PTMI_LAUNCH_WF(false, false, false, false);
        }
#undef PTMI_LAUNCH_WF
#undef PTMI_LAUNCH_WF_IMPL
#undef PTMI_LAUNCH_WF_BLOCK
        e = hipGetLastError();
        if (e == hipSuccess && stage == nullptr && sc.sampler == PTMI_SAMPLER_RANDOM) {
            // RANDOM sampler: behind the launch, the paths on its give-up list (returns at once when there is none)
            if (sc.tris_precomputed)
                hipLaunchKernelGGL(PTMI_DEV_NS::redo_random_kernel<true>, dim3(64), dim3(PTMI_DEV_NS::kBlock), 0, st, sc, first_iteration,
                                   n_iterations, iteration_stride, job_counter);
            else
                hipLaunchKernelGGL(PTMI_DEV_NS::redo_random_kernel<false>, dim3(64), dim3(PTMI_DEV_NS::kBlock), 0, st, sc, first_iteration,
                                   n_iterations, iteration_stride, job_counter);
            e = hipGetLastError();
        }
        if (e == hipSuccess && stage != nullptr) {
            // behind the launch, on its stream: the paths it gave up, if any (returns at once otherwise)
            if (sc.tris_precomputed)
                hipLaunchKernelGGL(PTMI_DEV_NS::redo_poisoned_kernel<true>, dim3(1024), dim3(PTMI_DEV_NS::kBlock), 0, st, sc, first_iteration,
                                   n_iterations, iteration_stride, stage, stage_stats, job_counter);
            else
                hipLaunchKernelGGL(PTMI_DEV_NS::redo_poisoned_kernel<false>, dim3(1024), dim3(PTMI_DEV_NS::kBlock), 0, st, sc, first_iteration,
                                   n_iterations, iteration_stride, stage, stage_stats, job_counter);
            e = hipGetLastError();
        }
    }
    if (e != hipSuccess) {
        if (err) *err = std::string("render_wavefront_kernel launch: ") + hipGetErrorString(e);
        return PTMI_ERR_HIP;
    }
    return PTMI_OK;
}

// What follows a wavefront launch of a sampler that owns its pixels: the staged radiances into the accumulators (per pixel,
// in iteration order) and the staged statistics words into the three histograms.  May run on another stream than the
// launch (the caller orders them with an event): these two small kernels are what keeps launches in iteration order.
int PTMI_ARITH(launch_accumulate_staged)(const DScene& sc, uint32_t first_iteration, uint32_t n_iterations, const float* stage,
                             const uint32_t* stage_stats, bool with_histograms, void* stream, std::string* err)
{
    if (n_iterations == 0 || sc.sampler == PTMI_SAMPLER_RANDOM) return PTMI_OK;
    const uint32_t n_pixels = sc.width * sc.height;
    if (sc.super_sampling)
        hipLaunchKernelGGL(PTMI_DEV_NS::accumulate_staged_ss_kernel, dim3((n_pixels + 255u) / 256u), dim3(256), 0,
                           (hipStream_t)stream, sc.image_color, sc.image_ray_nb, sc.image_v, stage, sc.stage_flag, n_pixels, first_iteration);
    else
        hipLaunchKernelGGL(PTMI_DEV_NS::accumulate_staged_kernel, dim3((n_pixels + 255u) / 256u), dim3(256), 0,
                           (hipStream_t)stream, sc.image_color, sc.image_ray_nb, stage, n_pixels, n_iterations);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (err) *err = std::string("accumulate_staged_kernel launch: ") + hipGetErrorString(e);
        return PTMI_ERR_HIP;
    }
    if (with_histograms && stage_stats != nullptr) return PTMI_ARITH(launch_histogram_staged)(sc, n_iterations, stage_stats, stream, err);
    return PTMI_OK;
}

// the staged statistics words of `n_iterations` iterations into the three histograms
int PTMI_ARITH(launch_histogram_staged)(const DScene& sc, uint32_t n_iterations, const uint32_t* stage_stats, void* stream, std::string* err)
{
    const uint32_t n_slots = sc.width * sc.height * n_iterations;
    uint32_t hb = (n_slots + 1023u) / 1024u;
    if (hb > 512u) hb = 512u;
    hipLaunchKernelGGL(PTMI_DEV_NS::histogram_staged_kernel, dim3(hb), dim3(1024), 0, (hipStream_t)stream, sc.hist_depths,
                       sc.hist_bbx, sc.hist_tri, stage_stats, sc.super_sampling ? sc.stage_flag : nullptr, n_slots);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        if (err) *err = std::string("histogram_staged_kernel launch: ") + hipGetErrorString(e);
        return PTMI_ERR_HIP;
    }
    return PTMI_OK;
}

}  // namespace ptmi_internal
