// ptmi_api.cpp - host side of libptmi.so: the C ABI of include/ptmi.h.
//
// Mirrors the life cycle of the reference backend (Controleur/PathTracer_OpenCL.cpp):
// setup_context -> initialize_memory -> render/read ... -> release, with the
// differences DESIGN.md lists (accumulators are zeroed; a launch covers a range
// of iterations; the scene is validated and re-laid out before upload).
// No CPU fallback exists: without a HIP device nothing here computes.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cmath>
#include <limits>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "ptmi.h"
#include "ptmi_internal.h"
#include "scene_layout.h"

using namespace ptmi_internal;

namespace {
const float kX2inv[1001] = {
#include "x2inv_table.inc"
};
std::mutex g_err_mutex;
std::string g_err;  // failures that have no context yet
}  // namespace

void ptmi_internal::set_global_error(const std::string& msg)
{
    std::lock_guard<std::mutex> lock(g_err_mutex);
    g_err = msg;
}

// One device's share of a render: a full scene replica, its own accumulators and stream (one process drives all of
// them from one host thread: every launch and copy below is asynchronous).
struct DeviceState {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;       // render stream (own_stream unless ptmi_set_stream gave another)
    hipStream_t copy_stream = nullptr;  // devices[0]: readbacks; other devices: their peer copy onto devices[0]
    std::vector<void*> allocations;     // freed with the scene
    float* d_color = nullptr;
    float* d_count = nullptr;
    uint32_t* d_hist = nullptr;  // depths | bbx | tri
    unsigned long long* d_counters = nullptr;
    uint32_t* d_job_counter = nullptr;
    // Staged radiances [iteration][pixel] float4 (+ one statistics word per path) of the launches in flight.  TWO sets, each
    // with a launch stream and job-queue counters of its own: consecutive launches alternate between them, so that the
    // ramp-up of launch k+1 fills the CUs the tail of launch k leaves idle (a persistent launch ends ragged: its last paths
    // finish one by one).  What keeps the results those of sequential launches: the staged values reach the accumulators
    // on the ONE main stream, launch after launch (launch_accumulate_staged), and set k % 2 is reused by launch k + 2
    // only after launch k's have been added (stage_free).
    // (round 4: FOUR sets.  Set 0 is sized for the longest launch and is the one long launches use, on the main stream; short
    // launches - fewer than 4 iterations - take the sets in turn, each with a COUNTER BLOCK and a device copy of the scene
    // record of its own, so that such a launch touches nothing of the context but its set until the main stream ADOPTS it:
    // adds its staged radiances to the accumulators, its statistics words to the histograms, its counter block to the
    // counters.  That is what lets the library RENDER AHEAD of a caller that asks for one image per call and waits for it, as
    // the reference's loop does, OpenCL.cpp:76-107: the launches of the next calls are already running when they are asked
    // for - `ahead` - and are dropped without a trace if the caller asks for something else.)
    static constexpr int kStageSets = 4;
    float* d_stage[kStageSets] = {};
    size_t stage_cap[kStageSets] = {};            // iterations a set holds
    hipStream_t launch_stream[kStageSets] = {};
    hipEvent_t rendered[kStageSets] = {};         // recorded on launch_stream[i] behind the kernel
    hipEvent_t stage_free[kStageSets] = {};       // recorded on the main stream behind the accumulation
    hipEvent_t reuse_after[kStageSets] = {};      // what the set's next launch waits for: stage_free (adopted) or rendered (dropped)
    unsigned long long* d_set_counters[kStageSets] = {};  // [PTMI_COUNTER_SPLITS][C_COUNT] each: one block per call a launch renders for
    DScene* d_scene_set[kStageSets] = {};         // ds with .counters = the set's block
    // a launch that renders for `calls` calls of n iterations each (ids from `first` on), of which `taken` have come and adopted
    // their part
    struct Ahead { uint32_t first, n, stride; int set; uint32_t calls, taken; };
    std::deque<Ahead> ahead;                      // launches in flight that calls have not (all) asked for yet, oldest first
    uint32_t streak = 0;                          // calls in a row that continued where the previous one left off
    uint32_t next_set = 0;
    // the previous ptmi_render call on this device: launches only run ahead of a caller that has been SEEN to continue where
    // it left off (ids first + n, same n), so a caller that jumps around pays nothing
    bool have_last = false;
    uint32_t last_first = 0, last_n = 0, last_stride = 0;
    hipEvent_t previous_call_done = nullptr;  // the event behind the previous call's work on the main stream
    uint32_t launches_issued = 0;
    DScene ds{};
    DScene* d_scene = nullptr;  // device copy of ds (what the wavefront kernel's path logic reads)
    // ptmi_snapshot ring: float[5*W*H] per slot (colour, then count), allocated on first use
    float* d_snapshot[PTMI_MAX_SNAPSHOT_SLOTS] = {};
    hipEvent_t snapshot_ready[PTMI_MAX_SNAPSHOT_SLOTS] = {};
    // Where this device's share of the image of ring slot s lives: slot s itself when its accumulators changed with that image,
    // else the slot of the last image that changed them (a device of a G-device render changes with every G-th image only, so
    // ptmi_render_snapshots copies 41.5 MB per OWN iteration instead of per image); -1 = the slot has never been filled.
    int source_slot[PTMI_MAX_SNAPSHOT_SLOTS];
    // ... so d_snapshot[] / snapshot_ready[] / snapshot_gen[] are BUFFERS, source_slot[s] names the buffer ring slot s shows, and
    // a buffer several slots show is never written: a new snapshot for one of them goes to a buffer no slot shows (there always
    // is one: as many buffers as slots) - the others keep showing what they showed (tests/test_api_fuzz_gpu.py).
    int buffer_refs[PTMI_MAX_SNAPSHOT_SLOTS] = {};
    uint32_t snapshot_gen[PTMI_MAX_SNAPSHOT_SLOTS] = {};  // bumped by every copy into the buffer
    float* d_peer_copy = nullptr;      // devices[0] only: where device k's snapshot lands before the sum, one per device
    hipEvent_t peer_copied = nullptr;  // ... and the event that says it has
    int landed_slot = -1;              // ... and which snapshot it holds: (slot, generation) - a peer sends only what has changed
    uint32_t landed_gen = 0;
    DeviceState() { for (int& s : source_slot) s = -1; }

    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending_events;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> free_events;
    double kernel_ms = 0;
    uint32_t kernel_launches = 0;
};

struct ptmi_ctx {
    ptmi_config cfg{};
    std::vector<DeviceState> dev;  // dev[0] = devices[0]: where partial images are summed and read back from
    bool have_scene = false;
    bool accum_bound = false;  // caller-owned accumulators (single device)
    std::string err;
    uint32_t stack_levels = PTMI_BVH_MAX_DEPTH;
    uint32_t iterations_per_launch = kMaxIterationsPerLaunch;
    // Why the uploaded scene is rendered by the one-path-per-lane kernel although the context did not ask for it (empty: it is
    // not).  See scene_needs_literal_kernel() in scene_layout.cpp.
    std::string literal_kernel_reason;

    // RCCL communicators, one per device of the context (single process, ncclCommInitAll): the sum of the devices' partial
    // images is an ncclReduce over xGMI where librccl is present and the devices are distinct (rccl_reduce_snapshots)
    std::vector<void*> rccl_comms;
    int rccl_state = 0;  // 0 = not tried, 1 = ready, -1 = unavailable (peer copies + a sum kernel instead)

    // on devices[0]
    float* d_reduced = nullptr;    // sum of the devices' snapshots (n_devices > 1)
    uint8_t* d_display = nullptr;  // B,G,R scanlines of ptmi_read_display
    size_t display_bytes = 0;
    // host side of the readbacks
    float* h_staging = nullptr;  // pinned, 5*W*H floats
    struct HostRange { char* p; size_t bytes; };
    std::vector<HostRange> pinned_host;  // caller buffers page-locked by ptmi_pin_host_buffer: readbacks DMA straight into them

    size_t npix() const { return (size_t)cfg.image_width * cfg.image_height; }
    uint32_t n_dev() const { return (uint32_t)dev.size(); }
};

namespace {

// the integrator's entry points in the context's arithmetic mode (ptmi_internal.h)
bool default_arithmetic(const ptmi_ctx* ctx) { return (ctx->cfg.flags & PTMI_FLAG_DEFAULT_ARITHMETIC) != 0; }
#define KERNELS_OF(ctx, name) (default_arithmetic(ctx) ? name##_da : name)

// one path per lane (kernels.hip) instead of the wavefront kernel: asked for, or needed by the scene - records that can yield
// NaN distances (literal_kernel_reason) rendered with the RANDOM sampler, whose samples are not staged, so that a path the
// wavefront kernel gives up could not be traced again; with the other samplers such a scene runs the wavefront kernel's
// NANSAFE instantiation (PTMI_LITERAL_KERNEL=1: the one-path-per-lane kernel as a whole, as before round 4, for A/B runs)
bool one_path_per_lane(const ptmi_ctx* ctx)
{
    if ((ctx->cfg.flags & PTMI_FLAG_MEGAKERNEL) != 0) return true;
    if (ctx->literal_kernel_reason.empty()) return false;
    const char* force = std::getenv("PTMI_LITERAL_KERNEL");
    return ctx->cfg.sampler == PTMI_SAMPLER_RANDOM || (force && force[0] == '1');
}

int fail(ptmi_ctx* ctx, int code, const std::string& msg)
{
    if (ctx) ctx->err = msg;
    else set_global_error(msg);
    return code;
}

#define HIP_TRY(ctx, expr)                                                                           \
    do {                                                                                             \
        hipError_t e__ = (expr);                                                                     \
        if (e__ != hipSuccess)                                                                       \
            return fail(ctx, PTMI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));      \
    } while (0)

// every device call below is made with the target device current
#define ON_DEVICE(ctx, d) HIP_TRY(ctx, hipSetDevice((d).device))

void free_scene_memory(ptmi_ctx* ctx)
{
    for (DeviceState& d : ctx->dev) {
        (void)hipSetDevice(d.device);
        // every stream that may still run a kernel or a copy on this memory - the launch streams too: after a failure between a
        // launch and the main stream's wait for it (render_on_device) a persistent kernel may still be reading the scene
        for (int i = 0; i < DeviceState::kStageSets; i++)
            if (d.launch_stream[i]) (void)hipStreamSynchronize(d.launch_stream[i]);
        d.ahead.clear();
        d.have_last = false;
        d.previous_call_done = nullptr;
        (void)hipStreamSynchronize(d.stream);
        if (d.copy_stream) (void)hipStreamSynchronize(d.copy_stream);
        for (void* p : d.allocations) (void)hipFree(p);
        d.allocations.clear();
        d.d_color = d.d_count = nullptr;
        d.d_hist = nullptr;
        d.d_counters = nullptr;
        d.d_job_counter = nullptr;
        d.d_scene = nullptr;
        for (int i = 0; i < DeviceState::kStageSets; i++) {
            if (d.d_stage[i]) (void)hipFree(d.d_stage[i]);
            d.d_stage[i] = nullptr;
            d.stage_cap[i] = 0;
            d.reuse_after[i] = nullptr;
            d.d_set_counters[i] = nullptr;
            d.d_scene_set[i] = nullptr;
        }
        for (uint32_t k = 0; k < PTMI_MAX_SNAPSHOT_SLOTS; k++) {
            if (d.d_snapshot[k]) (void)hipFree(d.d_snapshot[k]);
            d.d_snapshot[k] = nullptr;
            // ... and the event that says the slot is filled: a slot of the NEXT scene is empty until ptmi_snapshot fills it
            if (d.snapshot_ready[k]) (void)hipEventDestroy(d.snapshot_ready[k]);
            d.snapshot_ready[k] = nullptr;
            d.source_slot[k] = -1;
            d.buffer_refs[k] = 0;
        }
        d.landed_slot = -1;
        if (d.d_peer_copy) (void)hipFree(d.d_peer_copy);
        d.d_peer_copy = nullptr;
    }
    if (!ctx->dev.empty()) (void)hipSetDevice(ctx->dev[0].device);
    if (ctx->d_reduced) (void)hipFree(ctx->d_reduced);
    ctx->d_reduced = nullptr;
    if (ctx->d_display) (void)hipFree(ctx->d_display);
    ctx->d_display = nullptr;
    ctx->display_bytes = 0;
    ctx->accum_bound = false;
    ctx->have_scene = false;
    ctx->literal_kernel_reason.clear();
}

template <class T>
int upload(ptmi_ctx* ctx, DeviceState& d, const T* host, size_t count, const T** out)
{
    const size_t bytes = std::max<size_t>(count * sizeof(T), 16);  // reference uploads >= 1 byte (OpenCL.cpp:165)
    void* p = nullptr;
    HIP_TRY(ctx, hipMalloc(&p, bytes));
    d.allocations.push_back(p);
    // blocking copy: the source is pageable host memory
    if (count) HIP_TRY(ctx, hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T*>(p);
    return PTMI_OK;
}
template <class T>
int upload(ptmi_ctx* ctx, DeviceState& d, const std::vector<T>& host, const T** out)
{
    return upload(ctx, d, host.data(), host.size(), out);
}

int device_alloc(ptmi_ctx* ctx, DeviceState& d, size_t bytes, void** out)
{
    void* p = nullptr;
    HIP_TRY(ctx, hipMalloc(&p, bytes));
    d.allocations.push_back(p);
    *out = p;
    return PTMI_OK;
}

// Scene validation + re-layout: scene_layout.cpp (host-only, also behind ptmi_validate_scene).
int build_layout(ptmi_ctx* ctx, const ptmi_scene* sc, Relayout& out)
{
    std::string err;
    const int rc = ptmi_internal::build_layout(ctx->cfg, sc, out, err);
    return rc == PTMI_OK ? rc : fail(ctx, rc, err);
}

// The device copies of d.ds: the context's, and one per stage set whose launches count into the set's own block.
int upload_scene_records(ptmi_ctx* ctx, DeviceState& d)
{
    HIP_TRY(ctx, hipMemcpy(d.d_scene, &d.ds, sizeof(DScene), hipMemcpyHostToDevice));
    for (int i = 0; i < DeviceState::kStageSets; i++) {
        DScene k = d.ds;
        k.counters = d.d_set_counters[i];
        HIP_TRY(ctx, hipMemcpy(d.d_scene_set[i], &k, sizeof(DScene), hipMemcpyHostToDevice));
    }
    return PTMI_OK;
}

int fold_events(ptmi_ctx* ctx, DeviceState& d)
{
    ON_DEVICE(ctx, d);
    for (auto& ev : d.pending_events) {
        float ms = 0;
        HIP_TRY(ctx, hipEventSynchronize(ev.second));
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ev.first, ev.second));
        d.kernel_ms += ms;
        d.kernel_launches++;
        d.free_events.push_back(ev);
    }
    d.pending_events.clear();
    return PTMI_OK;
}

// Everything of a scene that lives on one device: the re-laid-out records, the accumulators, the statistics.
int upload_scene(ptmi_ctx* ctx, DeviceState& d, const Relayout& lay, const ptmi_scene* sc)
{
    ON_DEVICE(ctx, d);
    DScene& ds = d.ds;
    ds = DScene{};
    if (int rc = upload(ctx, d, lay.recs, &ds.tris)) return rc;
    ds.nodes = reinterpret_cast<const DNode*>(ds.tris);  // same array: a reference is an index of 64-byte records
    ds.n_records = (uint32_t)lay.recs.size();
    ds.wide_records = (lay.recs.size() > (1u << 26) || std::getenv("PTMI_WIDE_RECORDS") != nullptr) ? 1u : 0u;  // env: test switch
    if (int rc = upload(ctx, d, lay.tri_ids, &ds.tri_ids)) return rc;
    if (default_arithmetic(ctx) && lay.tris_precomputed) {
        // the records' reciprocal determinants in the reference's default arithmetic: a device instruction's values
        std::string err;
        if (int rc = launch_precompute_denominators_da(const_cast<DTri*>(ds.tris), ds.tri_ids, ds.n_records, d.stream, &err))
            return fail(ctx, rc, err);
        HIP_TRY(ctx, hipStreamSynchronize(d.stream));
    }
    if (int rc = upload(ctx, d, lay.shade, &ds.shade)) return rc;
    if (int rc = upload(ctx, d, lay.mats, &ds.mats)) return rc;
    if (int rc = upload(ctx, d, lay.big_leaves, &ds.big_leaves)) return rc;
    if (int rc = upload(ctx, d, sc->lights, sc->lights_size, &ds.lights)) return rc;
    if (int rc = upload(ctx, d, sc->textures, sc->textures_size, &ds.textures)) return rc;
    if (int rc = upload(ctx, d, sc->textures_data, sc->textures_data_size, &ds.texels)) return rc;

    const size_t npix = ctx->npix();
    const size_t hist_words = (size_t)ctx->cfg.ray_max_depth + 1 + 2 * PTMI_MAX_INTERSECTION_NUMBER;
    void *dc = nullptr, *dn = nullptr, *dh = nullptr, *dk = nullptr, *dsc = nullptr;
    if (int rc = device_alloc(ctx, d, npix * 16, &dc)) return rc;
    if (int rc = device_alloc(ctx, d, npix * 4, &dn)) return rc;
    if (int rc = device_alloc(ctx, d, hist_words * 4, &dh)) return rc;
    // counters, then one set of job-queue counters per stage set (256-byte aligned, up to 8 x 1024 dwords apart), then the
    // stage sets' counter blocks and scene records
    constexpr size_t kCounterBlock = ((C_COUNT * 8 + 255) / 256) * 256, kSceneBlock = ((sizeof(DScene) + 255) / 256) * 256;
    constexpr size_t kSetCounterBlock = ((PTMI_COUNTER_SPLITS * C_COUNT * 8 + 255) / 256) * 256;
    constexpr int kSets = DeviceState::kStageSets;
    if (int rc = device_alloc(ctx, d, kCounterBlock + 256 + kSets * 8 * 1024 * 4 + kSets * kSetCounterBlock, &dk)) return rc;
    if (int rc = device_alloc(ctx, d, (1 + kSets) * kSceneBlock, &dsc)) return rc;
    d.d_scene = (DScene*)dsc;
    d.d_color = (float*)dc; d.d_count = (float*)dn; d.d_hist = (uint32_t*)dh;
    d.d_counters = (unsigned long long*)dk;
    d.d_job_counter = (uint32_t*)((char*)dk + kCounterBlock);
    for (int i = 0; i < kSets; i++) {
        d.d_set_counters[i] = (unsigned long long*)((char*)dk + kCounterBlock + kSets * 8 * 1024 * 4 + i * kSetCounterBlock);
        d.d_scene_set[i] = (DScene*)((char*)dsc + (1 + i) * kSceneBlock);
    }

    ds.image_color = d.d_color;
    ds.image_ray_nb = d.d_count;
    const bool hist = !(ctx->cfg.flags & PTMI_FLAG_NO_HISTOGRAMS);
    ds.hist_depths = hist ? d.d_hist : nullptr;
    ds.hist_bbx = hist ? d.d_hist + ctx->cfg.ray_max_depth + 1 : nullptr;
    ds.hist_tri = hist ? d.d_hist + ctx->cfg.ray_max_depth + 1 + PTMI_MAX_INTERSECTION_NUMBER : nullptr;
    ds.counters = d.d_counters;
    ds.super_sampling = ctx->cfg.super_sampling ? 1u : 0u;
    if (ds.super_sampling) {
        void *dv = nullptr, *dx = nullptr, *df = nullptr;
        if (int rc = device_alloc(ctx, d, npix * 16, &dv)) return rc;
        if (int rc = device_alloc(ctx, d, sizeof kX2inv, &dx)) return rc;
        if (int rc = device_alloc(ctx, d, npix * 4, &df)) return rc;
        HIP_TRY(ctx, hipMemcpy(dx, kX2inv, sizeof kX2inv, hipMemcpyHostToDevice));
        ds.image_v = (float*)dv; ds.x2inv = (const float*)dx; ds.stage_flag = (float*)df;
    }
    ds.sky = *sc->sky;
    std::memcpy(ds.cam_pos, &sc->camera_position, 16);
    std::memcpy(ds.cam_dir, &sc->camera_direction, 16);
    std::memcpy(ds.cam_right, &sc->camera_right, 16);
    std::memcpy(ds.cam_up, &sc->camera_up, 16);
    ds.tris_precomputed = lay.tris_precomputed ? 1u : 0u;
    ds.plain_shading = lay.plain_shading ? 1u : 0u;
    ds.nan_safe = lay.literal_kernel_reason.empty() ? 0u : 1u;
    ds.nan_walk_box_tests = lay.nan_walk_box_tests; ds.nan_walk_tri_tests = lay.nan_walk_tri_tests; ds.nan_walk_last_tri = lay.nan_walk_last_tri;
    if (std::getenv("PTMI_WALK_NAN_RAYS") != nullptr) ds.nan_walk_box_tests = ds.nan_walk_tri_tests = 0xFFFFFFFFu;  // developer switch: A/B and tests
    ds.boxes_ordered = (lay.boxes_ordered && std::getenv("PTMI_GENERIC_BOXES") == nullptr) ? 1u : 0u;  // env: developer switch for A/B runs
    ds.root_ref = lay.root_ref;
    ds.width = ctx->cfg.image_width;
    ds.height = ctx->cfg.image_height;
    ds.max_depth = ctx->cfg.ray_max_depth;
    ds.n_lights = ctx->cfg.lights_size;
    ds.sampler = ctx->cfg.sampler;
    ds.russian_roulette = (ctx->cfg.flags & PTMI_FLAG_RUSSIAN_ROULETTE) ? 1u : 0u;
    ds.source_seed = (ctx->cfg.flags & PTMI_FLAG_SOURCE_SEED) ? 1u : 0u;
    if (int rc = upload_scene_records(ctx, d)) return rc;
    return PTMI_OK;
}

// Iteration ids [first, first + n) that device k of G takes: those congruent to k modulo G.
void device_share(uint32_t first, uint32_t n, uint32_t k, uint32_t G, uint32_t* first_k, uint32_t* n_k)
{
    const uint32_t skip = (k + G - first % G) % G;  // ids to skip from `first` to the first one of class k
    *first_k = first + skip;
    *n_k = skip < n ? (n - skip + G - 1) / G : 0;
}

constexpr uint32_t kUserSlots = PTMI_MAX_SNAPSHOT_SLOTS - 1;  // the last slot is the library's own

// Make ring slot `slot` of device `d` show buffer `b` (-1: nothing).
void point_slot(DeviceState& d, uint32_t slot, int b)
{
    if (d.source_slot[slot] >= 0) d.buffer_refs[d.source_slot[slot]]--;
    d.source_slot[slot] = b;
    if (b >= 0) d.buffer_refs[b]++;
}

// Queue, behind everything device `d` has been given so far, a copy of its accumulators for ring slot `slot`: into the buffer
// the slot shows if no other slot shows it too, else into one that no slot shows.  *buffer = where it went.
int snapshot_device(ptmi_ctx* ctx, DeviceState& d, uint32_t slot, int* buffer = nullptr)
{
    const size_t npix = ctx->npix();
    ON_DEVICE(ctx, d);
    int b = d.source_slot[slot];
    if (b < 0 || d.buffer_refs[b] > 1) {
        b = d.buffer_refs[slot] == 0 ? (int)slot : -1;  // (its own, as long as nobody else has taken it)
        for (int k = 0; b < 0 && k < (int)PTMI_MAX_SNAPSHOT_SLOTS; k++)
            if (d.buffer_refs[k] == 0) b = k;
        if (b < 0) return fail(ctx, PTMI_ERR_STATE, "snapshot ring: no free buffer");  // (cannot happen: as many buffers as slots)
    }
    if (!d.d_snapshot[b]) {
        void* p = nullptr;
        HIP_TRY(ctx, hipMalloc(&p, npix * 20));
        d.d_snapshot[b] = (float*)p;
    }
    if (!d.snapshot_ready[b]) HIP_TRY(ctx, hipEventCreateWithFlags(&d.snapshot_ready[b], hipEventDisableTiming));
    HIP_TRY(ctx, hipMemcpyAsync(d.d_snapshot[b], d.ds.image_color, npix * 16, hipMemcpyDeviceToDevice, d.stream));
    HIP_TRY(ctx, hipMemcpyAsync(d.d_snapshot[b] + 4 * npix, d.ds.image_ray_nb, npix * 4, hipMemcpyDeviceToDevice, d.stream));
    HIP_TRY(ctx, hipEventRecord(d.snapshot_ready[b], d.stream));
    point_slot(d, slot, b);
    d.snapshot_gen[b]++;
    if (buffer) *buffer = b;
    return PTMI_OK;
}
int snapshot_all(ptmi_ctx* ctx, uint32_t slot)
{
    for (DeviceState& d : ctx->dev)
        if (int rc = snapshot_device(ctx, d, slot)) return rc;
    return PTMI_OK;
}

// ptmi_render_snapshots: an image after EVERY iteration of the call although the iterations share launches.  Global
// iteration first + k (k < n) goes to slot (first_slot + k) % PTMI_MAX_USER_SLOTS.  A device's share of image k is whatever
// it has accumulated by then (its own ids up to first + k): it COPIES its accumulators only when they have changed since its
// last copy of this call - once per own iteration, plus once at the start of the call (so that every image of a call is
// served from slots of that call: a caller may be overwriting the previous call's) - and lets the other images of the call
// point at that copy (source_slot).
struct SnapshotPlan {
    uint32_t first, n, first_slot;
    uint32_t next = 0;    // next global k to provide on this device
    int last_slot = -1;   // this device's latest copy of this call: the BUFFER it went to
    bool changed = true;  // accumulators changed since (or no copy of this call yet)
};
int snapshots_up_to(ptmi_ctx* ctx, DeviceState& d, SnapshotPlan& plan, uint32_t k_end)
{
    for (; plan.next < k_end && plan.next < plan.n; plan.next++) {
        const uint32_t slot = (plan.first_slot + plan.next) % kUserSlots;
        if (plan.changed || plan.last_slot < 0) {
            if (int rc = snapshot_device(ctx, d, slot, &plan.last_slot)) return rc;
            plan.changed = false;
        } else {
            point_slot(d, slot, plan.last_slot);
        }
    }
    return PTMI_OK;
}

// How many launches the library keeps in flight AHEAD of a caller that renders one short call after the other and waits for
// each (DeviceState::ahead): PTMI_RENDER_AHEAD, default 2, 0 = never.
int render_ahead_depth()
{
    const char* e = std::getenv("PTMI_RENDER_AHEAD");  // (read per call: the tests switch it between contexts)
    const int v = e ? std::atoi(e) : 2;
    // (at most kStageSets - 2: beside them one launch whose calls are coming, and one set for a call that finds nothing)
    return v < 0 ? 0 : (v > DeviceState::kStageSets - 2 ? DeviceState::kStageSets - 2 : v);
}

// ... and how many CALLS one of those launches may render for (PTMI_RENDER_AHEAD_CALLS, default PTMI_COUNTER_SPLITS = 4, 1 = one
// launch per call): a persistent launch of ONE iteration spends a fifth of its time in its ragged end, and two such launches
// side by side share the CUs only as the first one's workgroups retire - at the very end of that tail.  One launch for the next
// four calls has one tail in four; each call adopts its quarter of the staging arrays and its own block of counters.
// (Eight calls per launch would need eight blocks of totals in the workgroup's LDS, which has room for four: as 32-bit words with
// a carry into memory they cost the kernel itself 1 - 3 %, 973 -> 965 Msamples/s on 1M triangles, material mix 2626 -> 2550.)
constexpr uint32_t kAheadIterations = PTMI_COUNTER_SPLITS;  // iterations of such a launch at most (calls x iterations per call)
uint32_t render_ahead_calls()
{
    const char* e = std::getenv("PTMI_RENDER_AHEAD_CALLS");
    const int v = e ? std::atoi(e) : PTMI_COUNTER_SPLITS;
    return v < 1 ? 1u : (v > PTMI_COUNTER_SPLITS ? (uint32_t)PTMI_COUNTER_SPLITS : (uint32_t)v);
}

// Stage set `set` able to hold `iterations` iterations (radiance float4 + one statistics word per path).  Growing it waits
// for whatever may still use the old arrays.
int ensure_stage_set(ptmi_ctx* ctx, DeviceState& d, int set, size_t iterations)
{
    if (d.stage_cap[set] >= iterations) return PTMI_OK;
    for (auto it = d.ahead.begin(); it != d.ahead.end();)
        it = it->set == set ? d.ahead.erase(it) : it + 1;
    if (d.launch_stream[set]) HIP_TRY(ctx, hipStreamSynchronize(d.launch_stream[set]));
    HIP_TRY(ctx, hipStreamSynchronize(d.stream));
    if (d.d_stage[set]) (void)hipFree(d.d_stage[set]);
    d.d_stage[set] = nullptr;
    d.stage_cap[set] = 0;
    d.reuse_after[set] = nullptr;
    void* p = nullptr;
    HIP_TRY(ctx, hipMalloc(&p, iterations * ctx->npix() * 20));
    d.d_stage[set] = (float*)p;
    d.stage_cap[set] = iterations;
    return PTMI_OK;
}

// One device's launches for its share of a ptmi_render call, bracketed by an event pair for ptmi_kernel_time.
int render_on_device(ptmi_ctx* ctx, DeviceState& d, uint32_t first, uint32_t n, uint32_t stride, SnapshotPlan* plan = nullptr)
{
    if (n == 0) return plan ? snapshots_up_to(ctx, d, *plan, plan->n) : PTMI_OK;
    ON_DEVICE(ctx, d);
    if (d.pending_events.size() >= 512)
        if (int rc = fold_events(ctx, d)) return rc;
    const bool megakernel = one_path_per_lane(ctx);
    const bool staged = !megakernel && ctx->cfg.sampler != PTMI_SAMPLER_RANDOM;
    // launch streams of their own: only where the launch itself neither reads nor writes the accumulators (staged results, no
    // adaptive sampling), on the context's own stream, and unless switched off (PTMI_SERIAL_LAUNCHES: developer A/B switch)
    static const bool serial_env = std::getenv("PTMI_SERIAL_LAUNCHES") != nullptr;
    const bool may_overlap = staged && !ctx->cfg.super_sampling && d.stream == d.own_stream && !serial_env;
    const size_t npix = ctx->npix();
    constexpr uint32_t kShort = 4;  // launches of fewer iterations run beside their neighbours (see below)
    // Rendering ahead: the call is ONE short launch on this device, nothing but staged results leaves the kernel (the histograms
    // of very deep paths are atomics inside it), and the caller has not asked for an image per iteration.  Per device: in a
    // context of G devices a caller that asks for one image per call comes to this device with every G-th call (ids first,
    // first + G, ...: the same pattern with stride G), and without launches ahead only ONE of the G devices would work at a time.
    const bool can_run_ahead = may_overlap && render_ahead_depth() > 0 && !plan && n < kShort &&
                               !(d.ds.hist_depths && ctx->cfg.ray_max_depth >= 64);
    if (!can_run_ahead) d.ahead.clear();  // (their sets are free again once their kernels have ended: reuse_after)
    // the caller comes back for the next ids with the same count
    const bool continues = d.have_last && d.last_n == n && d.last_stride == stride &&
                           (uint64_t)d.last_first + (uint64_t)n * stride == (uint64_t)first;
    if (staged) {
        // staging arrays: set 0 for the longest launch of this call, every set for a short one; grown on demand
        const size_t want = n < ctx->iterations_per_launch ? n : ctx->iterations_per_launch;
        if (int rc = ensure_stage_set(ctx, d, 0, want)) return rc;
        if (may_overlap && (n % ctx->iterations_per_launch) != 0 && (n % ctx->iterations_per_launch) < kShort)
            for (int i = can_run_ahead && continues ? 0 : 1; i < DeviceState::kStageSets; i++) {
                // (room for launches ahead of several calls: only once they are due, and only if the device has it - a launch
                // ahead renders for as many calls as its set holds)
                const size_t small = kShort - 1, large = can_run_ahead && continues && kAheadIterations > small ? kAheadIterations : small;
                int rc = ensure_stage_set(ctx, d, i, large);
                if (rc != PTMI_OK && large > small) {
                    (void)hipGetLastError();
                    rc = ensure_stage_set(ctx, d, i, small);
                }
                if (rc != PTMI_OK) return rc;
            }
        for (int i = 0; i < DeviceState::kStageSets && may_overlap; i++) {
            if (!d.launch_stream[i]) HIP_TRY(ctx, hipStreamCreateWithFlags(&d.launch_stream[i], hipStreamNonBlocking));
            if (!d.rendered[i]) HIP_TRY(ctx, hipEventCreateWithFlags(&d.rendered[i], hipEventDisableTiming));
            if (!d.stage_free[i]) HIP_TRY(ctx, hipEventCreateWithFlags(&d.stage_free[i], hipEventDisableTiming));
        }
    }
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    if (!d.free_events.empty()) {
        ev = d.free_events.back();
        d.free_events.pop_back();
    } else {
        HIP_TRY(ctx, hipEventCreate(&ev.first));
        if (hipEventCreate(&ev.second) != hipSuccess) {
            (void)hipEventDestroy(ev.first);
            return fail(ctx, PTMI_ERR_HIP, "hipEventCreate failed");
        }
    }
    std::string err;
    int rc = PTMI_OK;
    hipError_t e = hipSuccess;
    const bool stats_build = (ctx->cfg.flags & PTMI_FLAG_SCHEDULER_STATS) != 0;
    // a stage set no launch in flight ahead of the caller holds (there always is one: fewer launches ahead than sets)
    auto pick_set = [&]() {
        int set = 0;
        for (int tries = 0; tries < DeviceState::kStageSets; tries++) {
            set = (int)(d.next_set++ % DeviceState::kStageSets);
            bool held = false;
            for (const DeviceState::Ahead& a : d.ahead) held = held || a.set == set;
            if (!held) break;
        }
        return set;
    };
    // where the statistics words of a set's launches go: staged per path and counted after the launch, unless there is no
    // histogram (PTMI_FLAG_NO_HISTOGRAMS) or a depth that does not fit the 6-bit field
    auto stats_of = [&](int set) -> uint32_t* {
        if (!(d.d_stage[set] && d.ds.hist_depths && ctx->cfg.ray_max_depth < 64)) return nullptr;
        return reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(d.d_stage[set]) + d.stage_cap[set] * npix * 16);
    };
    // A SHORT launch on stage set `set`: on the set's own stream, counting into the set's own block - it touches nothing else
    // of the context, whether a call has asked for it or not.
    // `calls` > 1: the launch renders for that many calls of m / calls iterations each, and counts per call.
    auto launch_on_set = [&](int set, uint32_t f, uint32_t m, uint32_t calls) {
        hipStream_t ls = d.launch_stream[set];
        if (d.reuse_after[set]) e = hipStreamWaitEvent(ls, d.reuse_after[set], 0);
        if (e == hipSuccess) e = hipMemsetAsync(d.d_set_counters[set], 0, PTMI_COUNTER_SPLITS * C_COUNT * 8, ls);
        if (e != hipSuccess) return;
        DScene on_set = d.ds;
        on_set.counters = d.d_set_counters[set];
        on_set.split_paths = calls > 1 ? (uint32_t)((m / calls) * npix) : 0u;
        rc = KERNELS_OF(ctx, launch_render_wavefront)(on_set, d.d_scene_set[set], f, m, stride, d.d_job_counter + set * 8 * 1024, ctx->stack_levels,
                                                      stats_build, d.d_stage[set], stats_of(set), ls, &err);
        if (rc != PTMI_OK) return;
        e = hipEventRecord(d.rendered[set], ls);
        d.reuse_after[set] = d.rendered[set];  // (until the main stream adopts it)
        d.launches_issued++;
    };
    // Both events on the MAIN stream: [previous launch accumulated, this one accumulated].  With launch streams of their own
    // the intervals still tile the time line (no double counting of the overlap).
    e = hipEventRecord(ev.first, d.stream);
    if (e == hipSuccess) {
        if (megakernel) {
            rc = KERNELS_OF(ctx, launch_render)(d.ds, first, n, stride, d.stream, &err);
        } else {
            // one launch per chunk of iterations
            for (uint32_t done = 0; done < n && rc == PTMI_OK && e == hipSuccess;) {
                // SUPER_SAMPLING: the stop criterion of iteration k reads the accumulators after k-1 => one per launch
                const uint32_t cap = ctx->cfg.super_sampling ? 1u : ctx->iterations_per_launch;
                const uint32_t m = n - done < cap ? n - done : cap;
                const uint32_t f = first + done * stride;
                // Only SHORT launches get streams of their own (measured on MI355X, 1M triangles 1080p: one image per launch 631 -> 657
                // Msamples/s with the overlap; 16 images per launch 763 -> 753: two long persistent launches side by side only
                // get in each other's way, and their ragged ends are 1 % of their length anyway)
                const bool on_own_stream = may_overlap && m < kShort;
                int set = 0;
                uint32_t part = 0;  // which of the calls a launch that ran ahead rendered for this one is
                if (on_own_stream) {
                    // a launch that ran ahead for this call?  (the oldest first; anything else the caller did not come back for)
                    bool found = false;
                    while (can_run_ahead && !d.ahead.empty() && !found) {
                        DeviceState::Ahead& a = d.ahead.front();
                        found = a.n == m && a.stride == stride && (uint64_t)a.first + (uint64_t)a.taken * a.n * a.stride == (uint64_t)f;
                        set = a.set;
                        part = a.taken;
                        if (!found || ++a.taken == a.calls) d.ahead.pop_front();
                    }
                    if (!found) {
                        part = 0;
                        set = pick_set();
                        launch_on_set(set, f, m, 1);
                        if (rc != PTMI_OK || e != hipSuccess) break;
                    }
                    e = hipStreamWaitEvent(d.stream, d.rendered[set], 0);
                    if (e != hipSuccess) break;
                } else {
                    if (d.reuse_after[0]) e = hipStreamWaitEvent(d.stream, d.reuse_after[0], 0);  // (a short launch nobody adopted)
                    if (e != hipSuccess) break;
                    rc = KERNELS_OF(ctx, launch_render_wavefront)(d.ds, d.d_scene, f, m, stride, d.d_job_counter, ctx->stack_levels, stats_build,
                                                                  staged ? d.d_stage[0] : nullptr, staged ? stats_of(0) : nullptr, d.stream, &err);
                    if (rc != PTMI_OK) break;
                    d.launches_issued++;
                }
                float* const stage = staged ? d.d_stage[set] + (size_t)part * m * npix * 4 : nullptr;
                uint32_t* const stage_stats = staged && stats_of(set) ? stats_of(set) + (size_t)part * m * npix : nullptr;
                if (!plan) {
                    rc = KERNELS_OF(ctx, launch_accumulate_staged)(d.ds, f, m, stage, stage_stats, true, d.stream, &err);
                } else {
                    // one accumulation per iteration, each followed by the snapshots of the global iterations up to it
                    for (uint32_t j = 0; j < m && rc == PTMI_OK; j++) {
                        const uint32_t id = first + (done + j) * stride;
                        rc = snapshots_up_to(ctx, d, *plan, id - plan->first);  // images before this device's next own one
                        if (rc != PTMI_OK) break;
                        rc = KERNELS_OF(ctx, launch_accumulate_staged)(d.ds, id, 1, stage + (size_t)j * npix * 4,
                                                                       stage_stats ? stage_stats + (size_t)j * npix : nullptr, false, d.stream, &err);
                        plan->changed = true;
                        if (rc == PTMI_OK) rc = snapshots_up_to(ctx, d, *plan, id - plan->first + 1);
                    }
                    if (rc == PTMI_OK && stage_stats)
                        rc = KERNELS_OF(ctx, launch_histogram_staged)(d.ds, m, stage_stats, d.stream, &err);
                    if (rc != PTMI_OK && err.empty()) err = ctx->err;
                }
                if (rc != PTMI_OK) break;
                if (on_own_stream) rc = launch_add_counters(d.d_counters, d.d_set_counters[set] + (size_t)part * C_COUNT, C_COUNT, d.stream, &err);
                if (rc != PTMI_OK) break;
                if (staged && may_overlap) {  // (also behind a launch on the main stream: a later short launch may take set 0)
                    e = hipEventRecord(d.stage_free[set], d.stream);
                    d.reuse_after[set] = d.stage_free[set];
                }
                done += m;
            }
            // keep the next launches of a caller that comes back for one short call after the other in flight
            // ... and only ahead of a caller that WAITS: if what the previous call asked for was still running when this call
            // came, the caller keeps the GPU busy by itself (launches queued ahead of its readbacks) and more launches in
            // flight would only be in its way
            const bool caller_waits = d.previous_call_done == nullptr || hipEventQuery(d.previous_call_done) == hipSuccess;
            (void)hipGetLastError();  // (hipErrorNotReady is not an error)
            d.streak = continues ? d.streak + 1 : 0;
            if (can_run_ahead && continues && caller_waits && rc == PTMI_OK && e == hipSuccess) {
                uint64_t next = d.ahead.empty() ? (uint64_t)first + (uint64_t)n * stride
                                                : (uint64_t)d.ahead.back().first + (uint64_t)d.ahead.back().calls * n * stride;
                // (a launch whose calls have begun to come no longer counts: what replaces it starts as soon as it has ended)
                const int untouched = (int)d.ahead.size() - (!d.ahead.empty() && d.ahead.front().taken != 0u ? 1 : 0);
                // (Tried: the two launches ahead side by side with HALF of the persistent grid each, so that one's steady state fills
                // the other's ragged end - 1M triangles 149.9 -> 132.4 Mpaths/s, Cornell box 1080p 1352 -> 1211: two persistent
                // grids do not share the CUs evenly.  They take turns with whole grids.)
                for (int have = untouched; have < render_ahead_depth() && rc == PTMI_OK && e == hipSuccess; have++) {
                    // one launch for the next `calls` calls: 1, 2, 4 as the caller keeps coming back, four iterations at most
                    // (the statistics build counts per launch: one call each)
                    uint32_t calls = stats_build ? 1u : render_ahead_calls();
                    if (calls > kAheadIterations / n) calls = kAheadIterations / n;
                    if (calls * n > ctx->iterations_per_launch) calls = ctx->iterations_per_launch / n;
                    if (d.streak < 4 && calls > (1u << (d.streak - 1))) calls = 1u << (d.streak - 1);
                    while (calls > 1 && next + ((uint64_t)calls * n - 1) * stride > 0xFFFFFFFFull) calls--;
                    if (calls < 1 || next + (uint64_t)(n - 1) * stride > 0xFFFFFFFFull) break;
                    const int set = pick_set();
                    if ((size_t)n * calls > d.stage_cap[set]) calls = (uint32_t)(d.stage_cap[set] / n);
                    if (calls < 1) break;
                    launch_on_set(set, (uint32_t)next, n * calls, calls);
                    if (rc == PTMI_OK && e == hipSuccess) d.ahead.push_back({(uint32_t)next, n, stride, set, calls, 0u});
                    next += (uint64_t)calls * n * stride;
                }
            }
        }
        if (e == hipSuccess && rc == PTMI_OK && plan) rc = snapshots_up_to(ctx, d, *plan, plan->n);  // images after its last own one
        if (e == hipSuccess) e = hipEventRecord(ev.second, d.stream);
    }
    if (e != hipSuccess || rc != PTMI_OK) {
        d.free_events.push_back(ev);  // never timed: back to the pool
        if (rc != PTMI_OK) return fail(ctx, rc, err);
        return fail(ctx, PTMI_ERR_HIP, std::string("launch sequencing: ") + hipGetErrorString(e));
    }
    d.pending_events.push_back(ev);
    d.previous_call_done = ev.second;  // (stays valid in the pool: fold_events only moves the pair to free_events)
    d.have_last = true;
    d.last_first = first; d.last_n = n; d.last_stride = stride;
    return PTMI_OK;
}

// Is [p, p + bytes) inside a buffer the caller has page-locked with ptmi_pin_host_buffer?  Then a readback is one DMA into
// it; any other destination goes through the context's pinned staging buffer and a host memcpy.
bool host_is_pinned(ptmi_ctx* ctx, void* p, size_t bytes)
{
    for (auto& r : ctx->pinned_host)
        if ((char*)p >= r.p && (char*)p + bytes <= r.p + r.bytes) return true;
    return false;
}

int ensure_staging(ptmi_ctx* ctx)
{
    if (!ctx->h_staging) {
        void* p = nullptr;
        HIP_TRY(ctx, hipHostMalloc(&p, ctx->npix() * 20, hipHostMallocDefault));
        ctx->h_staging = (float*)p;
    }
    return PTMI_OK;
}

// Device float[4*npix] / float[npix] -> the caller's buffers over devices[0]'s `stream`, then wait for that stream.
int copy_out(ptmi_ctx* ctx, hipStream_t stream, const float* d_color, const float* d_count, float* image_color, float* image_ray_nb)
{
    const size_t npix = ctx->npix();
    const bool pin_c = image_color && host_is_pinned(ctx, image_color, npix * 16);
    const bool pin_n = image_ray_nb && host_is_pinned(ctx, image_ray_nb, npix * 4);
    if ((image_color && !pin_c) || (image_ray_nb && !pin_n))
        if (int rc = ensure_staging(ctx)) return rc;
    if (image_color)
        HIP_TRY(ctx, hipMemcpyAsync(pin_c ? image_color : ctx->h_staging, d_color, npix * 16, hipMemcpyDeviceToHost, stream));
    if (image_ray_nb)
        HIP_TRY(ctx, hipMemcpyAsync(pin_n ? image_ray_nb : ctx->h_staging + 4 * npix, d_count, npix * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    if (image_color && !pin_c) std::memcpy(image_color, ctx->h_staging, npix * 16);
    if (image_ray_nb && !pin_n) std::memcpy(image_ray_nb, ctx->h_staging + 4 * npix, npix * 4);
    return PTMI_OK;
}

// ---- RCCL, loaded at run time (the library has no link-time dependency on it) --------------------------------------------
// north_star: "samples-per-pixel shard across the GPUs of one node with an RCCL reduce of the framebuffer over xGMI".  One
// process drives all devices of a context, so the communicators come from ncclCommInitAll and the G reduce calls of an image
// are one group.  OPT-IN (PTMI_REDUCE=rccl) until the collective has run on a node with two GPUs: the default sum is peer copies
// + sum_images_kernel, whose order of additions is the device order ptmi.h documents; RCCL's order for more than two devices is
// its algorithm's, so the image's last bits depend on the choice.  PTMI_REDUCE=rccl-always sends even a one-device context
// through a one-rank communicator (how the tests exercise this code on a one-GPU box).  Any failure - library absent or of
// another major version, initialisation refused, a run-time error of ncclReduce / ncclGroupEnd - falls back to the peer path for
// the rest of the context's life (ptmi_rccl_state tells which path a context uses).
struct RcclApi {
    void* lib = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Reduce)(const void* send, void* recv, size_t count, int datatype, int op, int root, void* comm, hipStream_t stream) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int* version) = nullptr;
    int version = 0;
    bool ok = false;
};
RcclApi& rccl_api()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
        auto sym = [&](const char* n) { return dlsym(api.lib, n); };
        api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.Reduce = reinterpret_cast<decltype(api.Reduce)>(sym("ncclReduce"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(sym("ncclGetVersion"));
        // the enum values below are those of the NCCL 2 API (rccl.h of ROCm 7.2 reports 2.2x): another major version is refused
        if (api.GetVersion && api.GetVersion(&api.version) != 0) api.version = 0;
        const int major = api.version >= 10000 ? api.version / 10000 : api.version / 1000;
        api.ok = api.CommInitAll && api.CommDestroy && api.GroupStart && api.GroupEnd && api.Reduce && major == 2;
    });
    return api;
}
constexpr int kNcclFloat32 = 7, kNcclSum = 0;  // rccl.h: ncclDataType_t / ncclRedOp_t

const char* reduce_mode()
{
    const char* e = std::getenv("PTMI_REDUCE");
    return e ? e : "";
}

// devices[0]'s ctx->d_reduced = sum over the devices of their share of the image of ring slot `slot`, by ONE ncclReduce per
// device (root = devices[0]), each on its device's copy stream behind that device's snapshot.  PTMI_ERR_UNSUPPORTED = this
// context cannot use RCCL (library absent, a device listed twice, initialisation refused): the caller falls back to peer copies.
int rccl_reduce_snapshots(ptmi_ctx* ctx, uint32_t slot)
{
    if (ctx->rccl_state < 0 || std::strcmp(reduce_mode(), "peer") == 0) return PTMI_ERR_UNSUPPORTED;
    RcclApi& api = rccl_api();
    const int G = (int)ctx->n_dev();
    if (ctx->rccl_state == 0) {
        ctx->rccl_state = -1;
        if (!api.ok) return PTMI_ERR_UNSUPPORTED;
        std::vector<int> devs;
        for (DeviceState& d : ctx->dev) {
            for (int o : devs)
                if (o == d.device) return PTMI_ERR_UNSUPPORTED;  // RCCL wants distinct devices
            devs.push_back(d.device);
        }
        ctx->rccl_comms.assign((size_t)G, nullptr);
        if (api.CommInitAll(ctx->rccl_comms.data(), G, devs.data()) != 0) {
            ctx->rccl_comms.clear();
            (void)hipGetLastError();
            return PTMI_ERR_UNSUPPORTED;
        }
        ctx->rccl_state = 1;
    }
    const size_t count = ctx->npix() * 5;
    for (DeviceState& d : ctx->dev) {
        ON_DEVICE(ctx, d);
        HIP_TRY(ctx, hipStreamWaitEvent(d.copy_stream, d.snapshot_ready[d.source_slot[slot]], 0));
    }
    int rc = api.GroupStart();
    for (int k = 0; k < G && rc == 0; k++) {
        DeviceState& d = ctx->dev[(size_t)k];
        ON_DEVICE(ctx, d);
        float* send = d.d_snapshot[d.source_slot[slot]];
        rc = api.Reduce(send, k == 0 ? (void*)ctx->d_reduced : (void*)send, count, kNcclFloat32, kNcclSum, 0, ctx->rccl_comms[(size_t)k], d.copy_stream);
    }
    const int rc_end = api.GroupEnd();
    if (rc == 0) rc = rc_end;
    ON_DEVICE(ctx, ctx->dev[0]);
    if (rc != 0) {
        // a run-time refusal: remember why, never try again in this context, and let the caller sum through peer copies
        ctx->err = std::string("ncclReduce: ") + (api.GetErrorString ? api.GetErrorString(rc) : "error") + " (falling back to peer copies)";
        (void)hipGetLastError();
        ctx->rccl_state = -1;
        return PTMI_ERR_UNSUPPORTED;
    }
    return PTMI_OK;
}

// The image of ring slot `slot` on devices[0], ordered on dev[0].copy_stream: the slot itself for one device; for several,
// every other device's snapshot copied over (each on its own stream, so the transfers use their own xGMI links at the same
// time) and the sum of all of them, in device order, in ctx->d_reduced.
int gather_snapshot(ptmi_ctx* ctx, uint32_t slot, const float** image)
{
    DeviceState& lead = ctx->dev[0];
    const size_t npix = ctx->npix();
    for (DeviceState& d : ctx->dev)
        if (d.source_slot[slot] < 0 || !d.snapshot_ready[d.source_slot[slot]] || !d.d_snapshot[d.source_slot[slot]])
            return fail(ctx, PTMI_ERR_STATE, "ptmi_read_snapshot of a slot no ptmi_snapshot has filled");
    ON_DEVICE(ctx, lead);
    const int lead_src = lead.source_slot[slot];
    HIP_TRY(ctx, hipStreamWaitEvent(lead.copy_stream, lead.snapshot_ready[lead_src], 0));
    if (ctx->n_dev() == 1 && std::strcmp(reduce_mode(), "rccl-always") != 0) {
        *image = lead.d_snapshot[lead_src];
        return PTMI_OK;
    }
    if (!ctx->d_reduced) {
        void* p = nullptr;
        HIP_TRY(ctx, hipMalloc(&p, npix * 20));
        ctx->d_reduced = (float*)p;
    }
    // Which path: peer copies + the sum kernel in device order, unless the caller opted into the collective (PTMI_REDUCE=rccl /
    // rccl-always, above).  The images of a progressive per-image loop (ptmi_render_snapshots: ONE device's share is new per
    // image) are cheaper through the incremental peer copies anyway, which move 1 / (G - 1) of what a reduce would.
    const bool collective = std::strncmp(reduce_mode(), "rccl", 4) == 0;
    if (int rc = collective ? rccl_reduce_snapshots(ctx, slot) : (int)PTMI_ERR_UNSUPPORTED) {
        if (rc != PTMI_ERR_UNSUPPORTED) return rc;
        if (ctx->n_dev() == 1) {  // (rccl-always on a box without the library)
            *image = lead.d_snapshot[lead_src];
            return PTMI_OK;
        }
    } else {
        *image = ctx->d_reduced;
        return PTMI_OK;
    }
    // peer copies + one sum kernel (also the path of a context that lists one device several times, which RCCL refuses)
    const float* parts[PTMI_MAX_DEVICES];
    parts[0] = lead.d_snapshot[lead_src];
    // the previous sum must have read the landing buffers before they are overwritten: the peers' copies wait for the lead's
    // copy stream as it stands now
    hipEvent_t& gate = lead.peer_copied;
    if (!gate) HIP_TRY(ctx, hipEventCreateWithFlags(&gate, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventRecord(gate, lead.copy_stream));
    for (uint32_t k = 1; k < ctx->n_dev(); k++) {
        DeviceState& d = ctx->dev[k];
        if (!d.d_peer_copy) {  // on devices[0] (the lead device is current)
            void* p = nullptr;
            HIP_TRY(ctx, hipMalloc(&p, npix * 20));
            d.d_peer_copy = (float*)p;
        }
        parts[k] = d.d_peer_copy;
    }
    for (uint32_t k = 1; k < ctx->n_dev(); k++) {  // each peer pushes its snapshot over its own link, on a stream and with an event of its own device
        DeviceState& d = ctx->dev[k];
        const int src = d.source_slot[slot];
        // ... unless the landing buffer already holds that very snapshot: consecutive images of a G-device render differ in
        // ONE device's share, so an image costs one 41.5 MB peer copy, not G - 1
        if (d.landed_slot == src && d.landed_gen == d.snapshot_gen[src]) continue;
        ON_DEVICE(ctx, d);
        if (!d.peer_copied) HIP_TRY(ctx, hipEventCreateWithFlags(&d.peer_copied, hipEventDisableTiming));
        HIP_TRY(ctx, hipStreamWaitEvent(d.copy_stream, gate, 0));
        HIP_TRY(ctx, hipStreamWaitEvent(d.copy_stream, d.snapshot_ready[src], 0));
        HIP_TRY(ctx, hipMemcpyPeerAsync(d.d_peer_copy, lead.device, d.d_snapshot[src], d.device, npix * 20, d.copy_stream));
        HIP_TRY(ctx, hipEventRecord(d.peer_copied, d.copy_stream));
        d.landed_slot = src;
        d.landed_gen = d.snapshot_gen[src];
    }
    ON_DEVICE(ctx, lead);
    for (uint32_t k = 1; k < ctx->n_dev(); k++)
        if (ctx->dev[k].peer_copied) HIP_TRY(ctx, hipStreamWaitEvent(lead.copy_stream, ctx->dev[k].peer_copied, 0));
    std::string err;
    if (int rc = launch_sum_images(ctx->d_reduced, parts, ctx->n_dev(), npix * 5, lead.copy_stream, &err)) return fail(ctx, rc, err);
    *image = ctx->d_reduced;
    return PTMI_OK;
}

constexpr uint32_t kInternalSlot = PTMI_MAX_SNAPSHOT_SLOTS - 1;  // ptmi_read_image / ptmi_read_display of a multi-device context
static_assert(kInternalSlot == kUserSlots, "the library's own slot lies behind the callers'");

}  // namespace

extern "C" {

int ptmi_abi_version(void) { return PTMI_ABI_VERSION; }

void ptmi_device_share(uint32_t first_iteration, uint32_t n_iterations, uint32_t k, uint32_t n_devices, uint32_t* first_k,
                       uint32_t* n_k)
{
    uint32_t f = first_iteration, n = 0;
    if (n_devices > 0 && k < n_devices) device_share(first_iteration, n_iterations, k, n_devices, &f, &n);
    if (first_k) *first_k = f;
    if (n_k) *n_k = n;
}

int ptmi_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* ptmi_last_error(const ptmi_ctx* ctx)
{
    if (ctx) return ctx->err.c_str();
    std::lock_guard<std::mutex> lock(g_err_mutex);
    static thread_local std::string copy;
    copy = g_err;
    return copy.c_str();
}

int ptmi_setup_context(ptmi_ctx** out, const ptmi_config* cfg)
{
    if (!out) return fail(nullptr, PTMI_ERR_INVALID_ARGUMENT, "ctx out-pointer is NULL");
    *out = nullptr;
    if (!cfg || cfg->struct_size != sizeof(ptmi_config))
        return fail(nullptr, PTMI_ERR_INVALID_ARGUMENT, "config is NULL or struct_size mismatch (ABI)");
    if (cfg->image_width == 0 || cfg->image_height == 0 || (uint64_t)cfg->image_width * cfg->image_height > 0x3FFFFFFFull)
        return fail(nullptr, PTMI_ERR_INVALID_ARGUMENT, "image size must be in [1, 2^30) pixels");
    if (cfg->sampler > PTMI_SAMPLER_UNIFORM) return fail(nullptr, PTMI_ERR_INVALID_ARGUMENT, "unknown sampler");
    if (cfg->lights_size >= PTMI_MAX_LIGHT_SIZE)  // PathTracer.cpp:60-65
        return fail(nullptr, PTMI_ERR_LIMIT, "lights_size >= 30");
    if (cfg->n_devices > PTMI_MAX_DEVICES) return fail(nullptr, PTMI_ERR_INVALID_ARGUMENT, "n_devices > PTMI_MAX_DEVICES");
    if (cfg->super_sampling && (cfg->flags & PTMI_FLAG_MEGAKERNEL))
        return fail(nullptr, PTMI_ERR_UNSUPPORTED, "SUPER_SAMPLING needs the wavefront kernel");

    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(nullptr, PTMI_ERR_NO_DEVICE, "no HIP device available (there is no CPU fallback)");
    std::vector<int> ordinals;
    if (cfg->n_devices <= 1) ordinals.push_back(cfg->n_devices == 1 ? cfg->devices[0] : cfg->device);
    else ordinals.assign(cfg->devices, cfg->devices + cfg->n_devices);
    for (int o : ordinals)
        if (o < 0 || o >= n) return fail(nullptr, PTMI_ERR_NO_DEVICE, "device ordinal out of range");

    ptmi_ctx* ctx = new ptmi_ctx();
    ctx->cfg = *cfg;
    ctx->dev.resize(ordinals.size());
    hipError_t e = hipSuccess;
    for (size_t k = 0; k < ordinals.size() && e == hipSuccess; k++) {
        DeviceState& d = ctx->dev[k];
        d.device = ordinals[k];
        e = hipSetDevice(d.device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&d.own_stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&d.copy_stream, hipStreamNonBlocking);
        d.stream = d.own_stream;
        if (e == hipSuccess && k > 0 && d.device != ordinals[0]) {
            // direct peer copies over xGMI where the platform allows them (hipMemcpyPeerAsync stages through the host otherwise)
            (void)hipDeviceEnablePeerAccess(ordinals[0], 0);
            (void)hipGetLastError();
        }
    }
    if (e != hipSuccess) {
        const std::string msg = std::string("device/stream setup: ") + hipGetErrorString(e);
        ptmi_release(ctx);
        return fail(nullptr, PTMI_ERR_HIP, msg);
    }
    // iterations per launch: at most 16, fewer for very large images (32-bit job ids, staging array <= 4 GiB)
    {
        const uint64_t tiles = (uint64_t)((cfg->image_width + 7u) / 8u) * ((cfg->image_height + 7u) / 8u);
        const uint64_t by_jobs = (cfg->sampler == PTMI_SAMPLER_RANDOM ? 0x7FFFFFF0ull : 0xFFFFFFF0ull) / (tiles * 64u);  // (kernel_wavefront.hip: kGivenUp)
        const uint64_t by_bytes = (4ull << 30) / ((uint64_t)cfg->image_width * cfg->image_height * 20u);
        uint64_t cap = kMaxIterationsPerLaunch;
        if (by_jobs < cap) cap = by_jobs;
        if (by_bytes < cap) cap = by_bytes;
        ctx->iterations_per_launch = cap < 1 ? 1u : (uint32_t)cap;
    }
    *out = ctx;
    return PTMI_OK;
}

int ptmi_set_stream(ptmi_ctx* ctx, void* hip_stream)
{
    if (!ctx) return PTMI_ERR_INVALID_ARGUMENT;
    if (ctx->n_dev() != 1) return fail(ctx, PTMI_ERR_UNSUPPORTED, "ptmi_set_stream on a multi-device context");
    DeviceState& d = ctx->dev[0];
    ON_DEVICE(ctx, d);
    HIP_TRY(ctx, hipStreamSynchronize(d.stream));
    if (int rc = fold_events(ctx, d)) return rc;
    d.stream = hip_stream ? (hipStream_t)hip_stream : d.own_stream;
    return PTMI_OK;
}

int ptmi_initialize_memory(ptmi_ctx* ctx, const ptmi_scene* sc)
{
    if (!ctx) return PTMI_ERR_INVALID_ARGUMENT;
    if (!sc || sc->struct_size != sizeof(ptmi_scene))
        return fail(ctx, PTMI_ERR_INVALID_ARGUMENT, "scene is NULL or struct_size mismatch (ABI)");
    free_scene_memory(ctx);

    Relayout lay;
    if (int rc = build_layout(ctx, sc, lay)) return rc;
    ctx->literal_kernel_reason = lay.literal_kernel_reason;
    if (one_path_per_lane(ctx) && !(ctx->cfg.flags & PTMI_FLAG_MEGAKERNEL) && ctx->cfg.super_sampling) {
        ctx->literal_kernel_reason.clear();
        return fail(ctx, PTMI_ERR_UNSUPPORTED, "SUPER_SAMPLING needs the wavefront kernel, which cannot reproduce the reference on this scene with "
                                               "the RANDOM sampler: " + lay.literal_kernel_reason);
    }
    // a ray holds at most one pending far child per level it has descended
    ctx->stack_levels = lay.max_depth < 1 ? 1 : lay.max_depth;
    for (DeviceState& d : ctx->dev)
        if (int rc = upload_scene(ctx, d, lay, sc)) {
            const std::string msg = ctx->err;
            free_scene_memory(ctx);
            ctx->err = msg;
            return rc;
        }
    ctx->have_scene = true;
    if (int rc = ptmi_clear(ctx)) return rc;
    return ptmi_synchronize(ctx);
}

int ptmi_clear(ptmi_ctx* ctx)
{
    if (!ctx) return PTMI_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return fail(ctx, PTMI_ERR_STATE, "ptmi_clear before ptmi_initialize_memory");
    const size_t npix = ctx->npix();
    const size_t hist_words = (size_t)ctx->cfg.ray_max_depth + 1 + 2 * PTMI_MAX_INTERSECTION_NUMBER;
    for (DeviceState& d : ctx->dev) {
        ON_DEVICE(ctx, d);
        // (every launch is followed by its accumulation on the main stream, so main-stream order covers the launch streams)
        HIP_TRY(ctx, hipMemsetAsync(d.ds.image_color, 0, npix * 16, d.stream));
        HIP_TRY(ctx, hipMemsetAsync(d.ds.image_ray_nb, 0, npix * 4, d.stream));
        HIP_TRY(ctx, hipMemsetAsync(d.d_hist, 0, hist_words * 4, d.stream));
        HIP_TRY(ctx, hipMemsetAsync(d.d_counters, 0, C_COUNT * 8, d.stream));
        if (d.ds.image_v) HIP_TRY(ctx, hipMemsetAsync(d.ds.image_v, 0, npix * 16, d.stream));
        // a launch queued AFTER this call runs on a launch stream of its own and adds to the counters when it ends: it must
        // not overtake the memsets above
        HIP_TRY(ctx, hipStreamSynchronize(d.stream));
    }
    return PTMI_OK;
}

int ptmi_render(ptmi_ctx* ctx, uint32_t first_iteration, uint32_t n_iterations)
{
    if (!ctx) return PTMI_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return fail(ctx, PTMI_ERR_STATE, "ptmi_render before ptmi_initialize_memory");
    if (n_iterations == 0) return PTMI_OK;
    if ((uint64_t)first_iteration + n_iterations > 0xFFFFFFFFull)
        return fail(ctx, PTMI_ERR_INVALID_ARGUMENT, "iteration range overflows 32 bits");
    const uint32_t G = ctx->n_dev();
    for (uint32_t k = 0; k < G; k++) {
        uint32_t first_k, n_k;
        device_share(first_iteration, n_iterations, k, G, &first_k, &n_k);
        if (int rc = render_on_device(ctx, ctx->dev[k], first_k, n_k, G)) return rc;
    }
    return PTMI_OK;
}

int ptmi_render_snapshots(ptmi_ctx* ctx, uint32_t first_iteration, uint32_t n_iterations, uint32_t first_slot)
{
    if (!ctx) return PTMI_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return fail(ctx, PTMI_ERR_STATE, "ptmi_render_snapshots before ptmi_initialize_memory");
    if (n_iterations == 0) return PTMI_OK;
    if ((uint64_t)first_iteration + n_iterations > 0xFFFFFFFFull)
        return fail(ctx, PTMI_ERR_INVALID_ARGUMENT, "iteration range overflows 32 bits");
    if (n_iterations > kUserSlots || first_slot >= kUserSlots)
        return fail(ctx, PTMI_ERR_INVALID_ARGUMENT, "ptmi_render_snapshots: more iterations than snapshot slots, or slot out of range");
    if (ctx->cfg.super_sampling || ctx->cfg.sampler == PTMI_SAMPLER_RANDOM || (ctx->cfg.flags & PTMI_FLAG_MEGAKERNEL))
        return fail(ctx, PTMI_ERR_UNSUPPORTED, "ptmi_render_snapshots needs staged launches (JITTERED / UNIFORM sampler, wavefront kernel, no "
                                                "super_sampling): call ptmi_render + ptmi_snapshot per iteration instead");
    if (one_path_per_lane(ctx)) {  // a scene that needs the one-path-per-lane kernel: the same images, one launch each
        for (uint32_t k = 0; k < n_iterations; k++) {
            if (int rc = ptmi_render(ctx, first_iteration + k, 1)) return rc;
            if (int rc = snapshot_all(ctx, (first_slot + k) % kUserSlots)) return rc;
        }
        return PTMI_OK;
    }
    const uint32_t G = ctx->n_dev();
    for (uint32_t k = 0; k < G; k++) {
        uint32_t first_k, n_k;
        device_share(first_iteration, n_iterations, k, G, &first_k, &n_k);
        SnapshotPlan plan{first_iteration, n_iterations, first_slot};
        if (int rc = render_on_device(ctx, ctx->dev[k], first_k, n_k, G, &plan)) return rc;
    }
    return PTMI_OK;
}

int ptmi_reduce_path(const ptmi_ctx* ctx, int* rccl_state, int* n_communicators, int* nccl_version)
{
    if (!ctx) return PTMI_ERR_INVALID_ARGUMENT;
    if (rccl_state) *rccl_state = ctx->rccl_state;
    if (n_communicators) {
        int n = 0;
        for (void* c : ctx->rccl_comms) n += c != nullptr;
        *n_communicators = n;
    }
    if (nccl_version) *nccl_version = ctx->rccl_state != 0 ? rccl_api().version : 0;  // (never loads the library by itself)
    return PTMI_OK;
}

const char* ptmi_literal_kernel_reason(const ptmi_ctx* ctx)
{
    return ctx && ctx->have_scene && !ctx->literal_kernel_reason.empty() ? ctx->literal_kernel_reason.c_str() : nullptr;
}

int ptmi_synchronize(ptmi_ctx* ctx)
{
    if (!ctx) return PTMI_ERR_INVALID_ARGUMENT;
    for (DeviceState& d : ctx->dev) {
        ON_DEVICE(ctx, d);
        HIP_TRY(ctx, hipStreamSynchronize(d.stream));
    }
    return PTMI_OK;
}

int ptmi_snapshot(ptmi_ctx* ctx, uint32_t slot)
{
    if (!ctx) return PTMI_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return fail(ctx, PTMI_ERR_STATE, "ptmi_snapshot before ptmi_initialize_memory");
    if (slot >= kUserSlots) return fail(ctx, PTMI_ERR_INVALID_ARGUMENT, "snapshot slot out of range");
    return snapshot_all(ctx, slot);
}

int ptmi_read_snapshot(ptmi_ctx* ctx, uint32_t slot, float* image_color, float* image_ray_nb)
{
    if (!ctx) return PTMI_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return fail(ctx, PTMI_ERR_STATE, "ptmi_read_snapshot before ptmi_initialize_memory");
    if (slot >= PTMI_MAX_SNAPSHOT_SLOTS) return fail(ctx, PTMI_ERR_INVALID_ARGUMENT, "snapshot slot out of range");
    if (!image_color && !image_ray_nb) {  // wait only: the snapshot has been taken on every device (clFinish of that image)
        for (DeviceState& d : ctx->dev) {
            const int src = d.source_slot[slot];
            if (src < 0 || !d.snapshot_ready[src]) return fail(ctx, PTMI_ERR_STATE, "ptmi_read_snapshot of a slot no ptmi_snapshot has filled");
            ON_DEVICE(ctx, d);
            HIP_TRY(ctx, hipEventSynchronize(d.snapshot_ready[src]));
        }
        return PTMI_OK;
    }
    const float* image = nullptr;
    if (int rc = gather_snapshot(ctx, slot, &image)) return rc;
    return copy_out(ctx, ctx->dev[0].copy_stream, image, image + 4 * ctx->npix(), image_color, image_ray_nb);
}

int ptmi_read_image(ptmi_ctx* ctx, float* image_color, float* image_ray_nb)
{
    if (!ctx) return PTMI_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return fail(ctx, PTMI_ERR_STATE, "ptmi_read_image before ptmi_initialize_memory");
    if (ctx->n_dev() == 1) {  // straight from the accumulators, in order on the render stream
        DeviceState& d = ctx->dev[0];
        ON_DEVICE(ctx, d);
        return copy_out(ctx, d.stream, d.ds.image_color, d.ds.image_ray_nb, image_color, image_ray_nb);
    }
    if (int rc = snapshot_all(ctx, kInternalSlot)) return rc;
    return ptmi_read_snapshot(ctx, kInternalSlot, image_color, image_ray_nb);
}

int ptmi_pin_host_buffer(ptmi_ctx* ctx, void* buffer, size_t bytes)
{
    if (!ctx || !buffer || bytes == 0) return PTMI_ERR_INVALID_ARGUMENT;
    if (host_is_pinned(ctx, buffer, bytes)) return PTMI_OK;
    if (ctx->dev.empty()) return PTMI_ERR_STATE;
    ON_DEVICE(ctx, ctx->dev[0]);
    const hipError_t e = hipHostRegister(buffer, bytes, hipHostRegisterPortable);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // not sticky: readbacks into this buffer simply keep using the staging path
        return fail(ctx, PTMI_ERR_HIP, std::string("hipHostRegister: ") + hipGetErrorString(e));
    }
    ctx->pinned_host.push_back({(char*)buffer, bytes});
    return PTMI_OK;
}

int ptmi_unpin_host_buffer(ptmi_ctx* ctx, void* buffer)
{
    if (!ctx || !buffer) return PTMI_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < ctx->pinned_host.size(); i++) {
        if (ctx->pinned_host[i].p != (char*)buffer) continue;
        for (DeviceState& d : ctx->dev) {  // no copy into it may be in flight
            ON_DEVICE(ctx, d);
            HIP_TRY(ctx, hipStreamSynchronize(d.stream));
            if (d.copy_stream) HIP_TRY(ctx, hipStreamSynchronize(d.copy_stream));
        }
        (void)hipHostUnregister(buffer);
        (void)hipGetLastError();
        ctx->pinned_host.erase(ctx->pinned_host.begin() + (long)i);
        return PTMI_OK;
    }
    return fail(ctx, PTMI_ERR_INVALID_ARGUMENT, "ptmi_unpin_host_buffer: not a buffer ptmi_pin_host_buffer has page-locked");
}

int ptmi_write_image(ptmi_ctx* ctx, const float* image_color, const float* image_ray_nb)
{
    if (!ctx) return PTMI_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return fail(ctx, PTMI_ERR_STATE, "ptmi_write_image before ptmi_initialize_memory");
    const size_t npix = ctx->npix();
    // the image goes to devices[0]; the other devices' partial sums restart from zero
    for (uint32_t k = 0; k < ctx->n_dev(); k++) {
        DeviceState& d = ctx->dev[k];
        ON_DEVICE(ctx, d);
        HIP_TRY(ctx, hipStreamSynchronize(d.stream));
        if (k == 0) {
            if (image_color) HIP_TRY(ctx, hipMemcpy(d.ds.image_color, image_color, npix * 16, hipMemcpyHostToDevice));
            if (image_ray_nb) HIP_TRY(ctx, hipMemcpy(d.ds.image_ray_nb, image_ray_nb, npix * 4, hipMemcpyHostToDevice));
        } else {
            if (image_color) HIP_TRY(ctx, hipMemset(d.ds.image_color, 0, npix * 16));
            if (image_ray_nb) HIP_TRY(ctx, hipMemset(d.ds.image_ray_nb, 0, npix * 4));
        }
    }
    return PTMI_OK;
}

int ptmi_read_display(ptmi_ctx* ctx, uint8_t* bgr, uint32_t row_stride)
{
    if (!ctx || !bgr) return PTMI_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return fail(ctx, PTMI_ERR_STATE, "ptmi_read_display before ptmi_initialize_memory");
    const uint32_t w = ctx->cfg.image_width, h = ctx->cfg.image_height;
    if (row_stride < 3u * w || row_stride > 3u * w + 3u)
        return fail(ctx, PTMI_ERR_INVALID_ARGUMENT, "row_stride must be 3*W plus 0..3 padding bytes");
    DeviceState& d = ctx->dev[0];
    const float *color = d.ds.image_color, *count = d.ds.image_ray_nb;
    hipStream_t stream = d.stream;
    if (ctx->n_dev() > 1) {
        if (int rc = snapshot_all(ctx, kInternalSlot)) return rc;
        const float* image = nullptr;
        if (int rc = gather_snapshot(ctx, kInternalSlot, &image)) return rc;
        color = image; count = image + 4 * ctx->npix();
        stream = d.copy_stream;
    }
    ON_DEVICE(ctx, d);
    const size_t bytes = (size_t)h * row_stride;
    if (bytes > ctx->display_bytes) {
        if (ctx->d_display) (void)hipFree(ctx->d_display);
        ctx->d_display = nullptr;
        ctx->display_bytes = 0;
        void* p = nullptr;
        HIP_TRY(ctx, hipMalloc(&p, bytes));
        ctx->d_display = (uint8_t*)p;
        ctx->display_bytes = bytes;
    }
    std::string err;
    if (int rc = launch_display_bgr(color, count, ctx->d_display, w, h, row_stride, stream, &err)) return fail(ctx, rc, err);
    HIP_TRY(ctx, hipMemcpyAsync(bgr, ctx->d_display, bytes, hipMemcpyDeviceToHost, stream));
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    return PTMI_OK;
}

int ptmi_read_statistics(ptmi_ctx* ctx, uint32_t* depths, uint32_t* bbx, uint32_t* tri)
{
    if (!ctx) return PTMI_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return fail(ctx, PTMI_ERR_STATE, "ptmi_read_statistics before ptmi_initialize_memory");
    const uint32_t nd = ctx->cfg.ray_max_depth + 1;
    const size_t words = (size_t)nd + 2 * PTMI_MAX_INTERSECTION_NUMBER;
    std::vector<uint32_t> sum(words, 0u), part(words);
    for (DeviceState& d : ctx->dev) {  // histograms are integer sums over the devices
        ON_DEVICE(ctx, d);
        HIP_TRY(ctx, hipMemcpyAsync(part.data(), d.d_hist, words * 4, hipMemcpyDeviceToHost, d.stream));
        HIP_TRY(ctx, hipStreamSynchronize(d.stream));
        for (size_t i = 0; i < words; i++) sum[i] += part[i];
    }
    if (depths) std::memcpy(depths, sum.data(), nd * 4);
    if (bbx) std::memcpy(bbx, sum.data() + nd, PTMI_MAX_INTERSECTION_NUMBER * 4);
    if (tri) std::memcpy(tri, sum.data() + nd + PTMI_MAX_INTERSECTION_NUMBER, PTMI_MAX_INTERSECTION_NUMBER * 4);
    return PTMI_OK;
}

static int read_counter_block(ptmi_ctx* ctx, unsigned long long* total)
{
    for (int i = 0; i < C_COUNT; i++) total[i] = 0;
    for (DeviceState& d : ctx->dev) {
        ON_DEVICE(ctx, d);
        unsigned long long h[C_COUNT];
        HIP_TRY(ctx, hipMemcpyAsync(h, d.d_counters, sizeof h, hipMemcpyDeviceToHost, d.stream));
        HIP_TRY(ctx, hipStreamSynchronize(d.stream));
        for (int i = 0; i < C_COUNT; i++) total[i] += h[i];
    }
    return PTMI_OK;
}

int ptmi_get_counters(ptmi_ctx* ctx, ptmi_counters* out)
{
    if (!ctx || !out) return PTMI_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return fail(ctx, PTMI_ERR_STATE, "ptmi_get_counters before ptmi_initialize_memory");
    unsigned long long h[C_COUNT];
    if (int rc = read_counter_block(ctx, h)) return rc;
    out->paths = h[C_PATHS]; out->segments = h[C_SEGMENTS]; out->surface_hits = h[C_HITS];
    out->shadow_rays = h[C_SHADOW]; out->box_tests = h[C_BBX]; out->triangle_tests = h[C_TRI];
    return PTMI_OK;
}

int ptmi_get_scheduler_stats(ptmi_ctx* ctx, ptmi_scheduler_stats* out)
{
    if (!ctx || !out) return PTMI_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return fail(ctx, PTMI_ERR_STATE, "ptmi_get_scheduler_stats before ptmi_initialize_memory");
    unsigned long long h[C_COUNT];
    if (int rc = read_counter_block(ctx, h)) return rc;
    out->trips_node = h[C_TRIPS_I]; out->lanes_node = h[C_LANES_I];
    out->trips_triangle = h[C_TRIPS_T]; out->lanes_triangle = h[C_LANES_T];
    out->trips_path = h[C_TRIPS_P]; out->lanes_path = h[C_LANES_P];
    out->cycles_path = h[C_CYCLES_P]; out->cycles_loop = h[C_CYCLES_LOOP];
    out->leaf_item_violations = h[C_ITEM_VIOLATIONS];
    out->paths_retraced = h[C_RETRACED];
    out->textured_hits = h[C_TEXTURED_HITS];
    uint32_t lanes = 0, resident = 0;
    KERNELS_OF(ctx, last_wavefront_grid)(ctx->dev[0].device, &lanes, &resident);
    out->workgroup_lanes = lanes;
    out->resident_workgroups = resident;
    return PTMI_OK;
}

int ptmi_get_invariant_checks(ptmi_ctx* ctx, ptmi_invariant_checks* out)
{
    if (!ctx || !out) return PTMI_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return fail(ctx, PTMI_ERR_STATE, "ptmi_get_invariant_checks before ptmi_initialize_memory");
    unsigned long long h[C_COUNT];
    if (int rc = read_counter_block(ctx, h)) return rc;
    out->sample_out_of_range = h[C_CHK_SAMPLE]; out->normal_not_facing_ray = h[C_CHK_NORMALS];
    out->negative_direct_radiance = h[C_CHK_RADIANCE]; out->scattered_below_surface = h[C_CHK_HEMISPHERE];
    out->statistics_out_of_range = h[C_CHK_STATS_RANGE];
    out->refraction_undefined_in_reference = h[C_UNDEF_REFRACTION];
    return PTMI_OK;
}

int ptmi_kernel_time(ptmi_ctx* ctx, double* total_ms, uint32_t* n_launches)
{
    if (!ctx) return PTMI_ERR_INVALID_ARGUMENT;
    double ms = 0;
    uint32_t n = 0;
    for (DeviceState& d : ctx->dev) {  // summed over the devices: total / launches stays the average launch duration
        if (int rc = fold_events(ctx, d)) return rc;
        ms += d.kernel_ms;
        n += d.kernel_launches;
        d.kernel_ms = 0;
        d.kernel_launches = 0;
    }
    if (total_ms) *total_ms = ms;
    if (n_launches) *n_launches = n;
    return PTMI_OK;
}

int ptmi_device_accumulators(ptmi_ctx* ctx, void** d_color, void** d_count)
{
    if (!ctx) return PTMI_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return fail(ctx, PTMI_ERR_STATE, "ptmi_device_accumulators before ptmi_initialize_memory");
    if (ctx->n_dev() != 1) return fail(ctx, PTMI_ERR_UNSUPPORTED, "ptmi_device_accumulators on a multi-device context (partial sums)");
    if (d_color) *d_color = ctx->dev[0].ds.image_color;
    if (d_count) *d_count = ctx->dev[0].ds.image_ray_nb;
    return PTMI_OK;
}

int ptmi_device_variance(ptmi_ctx* ctx, void** d_image_v)
{
    if (!ctx || !d_image_v) return PTMI_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return fail(ctx, PTMI_ERR_STATE, "ptmi_device_variance before ptmi_initialize_memory");
    if (ctx->n_dev() != 1) return fail(ctx, PTMI_ERR_UNSUPPORTED, "ptmi_device_variance on a multi-device context (per-device moments)");
    if (!ctx->dev[0].ds.image_v) return fail(ctx, PTMI_ERR_STATE, "no variance accumulator: the context was set up without super_sampling");
    *d_image_v = ctx->dev[0].ds.image_v;
    return PTMI_OK;
}

int ptmi_read_variance(ptmi_ctx* ctx, float* image_v)
{
    if (!ctx || !image_v) return PTMI_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return fail(ctx, PTMI_ERR_STATE, "ptmi_read_variance before ptmi_initialize_memory");
    DeviceState& d0 = ctx->dev[0];
    if (!d0.ds.image_v) return fail(ctx, PTMI_ERR_STATE, "no variance accumulator: the context was set up without super_sampling");
    const size_t npix = ctx->npix();
    ON_DEVICE(ctx, d0);
    HIP_TRY(ctx, hipMemcpyAsync(image_v, d0.ds.image_v, npix * 16, hipMemcpyDeviceToHost, d0.stream));
    HIP_TRY(ctx, hipStreamSynchronize(d0.stream));
    if (ctx->n_dev() == 1) return PTMI_OK;
    // Several devices: each kept (sum S, count n, imageV = sum of squared deviations M2) of ITS samples.  S and n add; M2
    // does not:  M2 = M2_a + M2_b + (mean_b - mean_a)^2 * n_a * n_b / (n_a + n_b)   (Chan, Golub, LeVeque).  Merged here on
    // the host, device after device, in fp32 with the operation order of distributed.merge_moments.
    std::vector<float> sum(npix * 4), cnt(npix), sb(npix * 4), nb(npix), vb(npix * 4);
    HIP_TRY(ctx, hipMemcpy(sum.data(), d0.ds.image_color, npix * 16, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(cnt.data(), d0.ds.image_ray_nb, npix * 4, hipMemcpyDeviceToHost));
    for (uint32_t k = 1; k < ctx->n_dev(); k++) {
        DeviceState& d = ctx->dev[k];
        ON_DEVICE(ctx, d);
        HIP_TRY(ctx, hipStreamSynchronize(d.stream));
        HIP_TRY(ctx, hipMemcpy(sb.data(), d.ds.image_color, npix * 16, hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipMemcpy(nb.data(), d.ds.image_ray_nb, npix * 4, hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipMemcpy(vb.data(), d.ds.image_v, npix * 16, hipMemcpyDeviceToHost));
        for (size_t p = 0; p < npix; p++) {
            const float na = cnt[p], nbp = nb[p], n = na + nbp;
            const float sa_ = na > 0 ? na : 1.f, sb_ = nbp > 0 ? nbp : 1.f, sn_ = n > 0 ? n : 1.f;
            const bool both = na > 0 && nbp > 0;
            for (int c = 0; c < 4; c++) {
                const float delta = sb[4 * p + c] / sb_ - sum[4 * p + c] / sa_;
                const float cross = delta * delta * (na * nbp / sn_);
                image_v[4 * p + c] = (image_v[4 * p + c] + vb[4 * p + c]) + (both ? cross : 0.f);
                sum[4 * p + c] = sum[4 * p + c] + sb[4 * p + c];
            }
            cnt[p] = n;
        }
    }
    return PTMI_OK;
}

int ptmi_write_variance(ptmi_ctx* ctx, const float* image_v)
{
    if (!ctx || !image_v) return PTMI_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return fail(ctx, PTMI_ERR_STATE, "ptmi_write_variance before ptmi_initialize_memory");
    if (ctx->n_dev() != 1) return fail(ctx, PTMI_ERR_UNSUPPORTED, "ptmi_write_variance on a multi-device context");
    DeviceState& d = ctx->dev[0];
    if (!d.ds.image_v) return fail(ctx, PTMI_ERR_STATE, "no variance accumulator: the context was set up without super_sampling");
    ON_DEVICE(ctx, d);
    HIP_TRY(ctx, hipStreamSynchronize(d.stream));
    HIP_TRY(ctx, hipMemcpy(d.ds.image_v, image_v, ctx->npix() * 16, hipMemcpyHostToDevice));
    return PTMI_OK;
}

int ptmi_bind_accumulators(ptmi_ctx* ctx, void* d_color, void* d_count)
{
    if (!ctx) return PTMI_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return fail(ctx, PTMI_ERR_STATE, "ptmi_bind_accumulators before ptmi_initialize_memory");
    if (ctx->n_dev() != 1) return fail(ctx, PTMI_ERR_UNSUPPORTED, "ptmi_bind_accumulators on a multi-device context");
    if ((d_color == nullptr) != (d_count == nullptr))
        return fail(ctx, PTMI_ERR_INVALID_ARGUMENT, "bind both accumulators or neither");
    DeviceState& d = ctx->dev[0];
    ON_DEVICE(ctx, d);
    HIP_TRY(ctx, hipStreamSynchronize(d.stream));
    d.ds.image_color = d_color ? (float*)d_color : d.d_color;
    d.ds.image_ray_nb = d_count ? (float*)d_count : d.d_count;
    ctx->accum_bound = d_color != nullptr;
    if (int rc = upload_scene_records(ctx, d)) return rc;
    return PTMI_OK;
}

void ptmi_release(ptmi_ctx* ctx)
{
    if (!ctx) return;
    free_scene_memory(ctx);
    for (void* comm : ctx->rccl_comms)
        if (comm) (void)rccl_api().CommDestroy(comm);
    ctx->rccl_comms.clear();
    for (auto& r : ctx->pinned_host) (void)hipHostUnregister(r.p);
    (void)hipGetLastError();
    if (ctx->h_staging) (void)hipHostFree(ctx->h_staging);
    for (DeviceState& d : ctx->dev) {
        (void)hipSetDevice(d.device);
        for (auto& ev : d.pending_events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
        for (auto& ev : d.free_events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
        for (uint32_t k = 0; k < PTMI_MAX_SNAPSHOT_SLOTS; k++)
            if (d.snapshot_ready[k]) (void)hipEventDestroy(d.snapshot_ready[k]);
        if (d.peer_copied) (void)hipEventDestroy(d.peer_copied);
        for (int i = 0; i < DeviceState::kStageSets; i++) {
            if (d.rendered[i]) (void)hipEventDestroy(d.rendered[i]);
            if (d.stage_free[i]) (void)hipEventDestroy(d.stage_free[i]);
            if (d.launch_stream[i]) (void)hipStreamDestroy(d.launch_stream[i]);
        }
        if (d.own_stream) (void)hipStreamDestroy(d.own_stream);
        if (d.copy_stream) (void)hipStreamDestroy(d.copy_stream);
    }
    delete ctx;
}

}  // extern "C"
