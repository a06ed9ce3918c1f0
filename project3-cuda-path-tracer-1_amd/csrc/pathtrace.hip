// pathtrace.hip — render context and per-iteration driver: the MI355X-native counterpart of
// pathtraceInit / pathtraceFree / pathtrace (reference src/pathtrace.cu:85-125, 637-816).
//
// One iteration = generate + traceDepth bounces, enqueued back-to-back on one HIP stream with the live-path
// counts resident on the device (the reference does 4 blocking 4-byte D2H copies, a cudaDeviceSynchronize,
// a full-frame cudaMemset and a cudaMalloc/cudaFree pair per bounce/iteration).  Nothing here synchronises
// with the host; image / counts are fetched lazily by the accessors.
//
// ITERATIONS IN FLIGHT.  Iterations only interact by addition into the radiance buffer, so the context can hold L
// independent "lanes" (stream + path/hit/flag buffers + counts + one radiance plane each); iteration `it` runs on lane
// (it - 1) % L, and consecutive pathtrace() calls therefore overlap on the GPU.  Every iteration still executes exactly the
// reference's algorithm with the reference's RNG keys; only the order of the per-pixel float additions changes:
// plane l accumulates iterations l+1, l+1+L, ... in order, and the image is ((plane0 + plane1) + plane2) + ...
// L = 1 (default) reproduces the reference's accumulation order.  This matters on MI355X because late bounces, BVH walks
// (chains of dependent loads with ~1 wave per SIMD left) and tile-sharded frames leave most of the 256 CUs idle.
#include "ptk_internal.h"
#include "pathtrace.h"
#ifdef PTK_TEST_HOOKS
#include "../../include/ptk_test_hooks.h"
#endif
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace ptk;

__attribute__((weak)) SampleMode sampleMode = MIS;       // main.cpp:20 (the application's definition wins)

namespace {

constexpr int kMaxLanes = 16;
// PTK_HOST_TIMING=1 (diagnostic): host time spent inside hipGraphLaunch, in the merge's three runtime calls and in the rest of ptk_pathtrace, printed when the context is freed
struct HostTiming { bool on = false; double graph_us = 0, merge_us = 0, call_us = 0, graph_max = 0; long calls = 0; } g_ht;
static inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Iterations in flight the C++ API starts with (pathtraceInit) = one lane per hardware queue of the lanes' pool (lane_stream_create).  Measured against
// three, Grays/s (profiles/r5_ab_lane_queues.txt): cornell 400^2 depth 4 12.9 -> 15.6 (MIS 12.1 -> 13.8), textureEnv 15.9 -> 18.3, obj_test 13.9 -> 15.2,
// mis_test 1024^2 depth 16 MIS 18.3 -> 19.2 / BSDF 24.5 -> 27.7, bunny 6.14 -> 6.56 (MIS 5.62 -> 6.01), metallic 14.7 -> 15.2, cornell 800^2 +-1 %; five or
// more share queues again and lose.
constexpr int kDefaultLanes = 4;

struct Lane {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    PtkPaths bufA = {nullptr, nullptr, nullptr}, bufB = {nullptr, nullptr, nullptr};
    PtkHits hits = {nullptr, nullptr, nullptr}, hitsB = {nullptr, nullptr, nullptr};     // hitsB, hrA/hrB: fused pipeline (hit records ping-pong with the paths)
    int* hrA = nullptr; int* hrB = nullptr;
    Workspace ws;
    float* plane = nullptr;             // float[spp][n_depth][3*P]: the radiance of the launch in flight on this lane, merged into the image when it is done (flush_merges); the image itself with one lane, one sample per launch or tiles in flight
    unsigned char* flags = nullptr;     // touched-tile flags of those planes (ordered accumulation), [spp][n_depth][ceil(P / 64)]
    int last_iter = 0;
    // hipGraph replay of one iteration (all launches of render_iteration): the iteration number lives on the device
    int* iter_dev = nullptr;
    // one captured graph per plane set (batched merges: a lane alternates between two sets of planes, and the film pointers are baked into the launches)
    struct GraphSlot { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; int mode = -1, full = -1, late = -1; } gs[2];
    int cur_set = 0;                    // the plane set (and graph slot) the lane's next iteration renders into
    float* plane_sets[2] = {nullptr, nullptr}; unsigned char* flag_sets[2] = {nullptr, nullptr};
    hipEvent_t set_free[2] = {nullptr, nullptr};      // recorded behind the merge that folded (and zeroed) the set
    bool set_busy[2] = {false, false};  // set_free[k] has been recorded: the next iteration into set k waits for it
    int expected_iter = 0;              // value *iter_dev will hold when the next replay starts
    int* h_counts = nullptr;            // pinned host copy of ws.counts of an earlier iteration (asynchronous read-back, never waited for)
    hipEvent_t counts_ev = nullptr;
    bool counts_pending = false;
    int* seg_done_mem = nullptr;        // Workspace::seg_done points here unless pipeline 7 asks for the separate k_segments launch
    hipEvent_t merged = nullptr;        // ordered accumulation: recorded after a merge that ran on this lane's stream (flush_merges)
    bool wait_resolved = false;         // the next enqueue must wait for lane 0's asynchronous reader of the planes / the image (plane sum, tone map of the previous pbo)
    int fresh_hint = 0;                 // iterations enqueued on this lane since the hint was reset (read-backs every iteration at first)
    // what this lane renders: the whole frame / this rank's tiles (iterations in flight), or ONE band of rows (tiles in flight)
    int n_parts = 1, part = 0, tile_rows = 1, n_pix = 0, capacity = 0;
    int full_bounces = 0;               // bounces enqueued at full width; the rest of an iteration runs in k_tail (sparse_tail())
    int traced_depth = 0;               // of the last read-back of this lane's counts (tiles in flight: the viewer's TracedDepth is the maximum over the bands)
    int gen_late = 0;                   // open scene (fewer than ~half of the camera rays hit, learnt like full_bounces): the camera launch stores the hits' path state only
    // tiles in flight: cumulative hit counts [PTK_CUM_RING][traceDepth + 2] (ptk_internal.h: Workspace::cum_*) and one event per ring slot,
    // recorded after this lane's iteration i in slot i % PTK_CUM_RING
    int* cum = nullptr;
    hipEvent_t ring_ev[PTK_CUM_RING] = {nullptr, nullptr, nullptr, nullptr};
};

struct RenderContext {
    Scene* scene = nullptr;
    GuiDataContainer* gui = nullptr;
    std::vector<Lane> lanes;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t resolved = nullptr;      // recorded on lane 0 after an asynchronous reader of the image (ptk_pathtrace's tone map)
    float* planes = nullptr;            // float[L][spp][3*P]
    float* image = nullptr;             // float[3*P]: the running sum, in iteration order (== planes when there is one plane)
    bool sched_explicit = false;        // ptk_set_iterations_in_flight / _tiles_in_flight / _graph_replay was called: pathtraceInit keeps the caller's schedule
    int want_ordered = -1;              // ptk_set_ordered_accumulation: -1 / 1 ordered (default), 0 planes that accumulate
    bool ordered = false;               // several planes AND ordered accumulation (flush_merges); false: the planes accumulate, resolve_image sums them
    int n_depth = 1;                    // planes per sample: traceDepth + 1 (one per bounce) with ordered accumulation initialised in MIS mode, else 1
    unsigned char* flags = nullptr;     // [L][spp][n_depth][ceil(P / 64)] (ordered accumulation), else null
    hipEvent_t last_merged = nullptr;   // ordered accumulation: the event of the latest merge (some lane's `merged`), nullptr before the first
    // batched merges (round 5): the iterations of up to `merge_batch` consecutive calls — consecutive lanes, the same plane set — are folded into
    // the image by ONE k_merge_planes launch, in call order; a lane alternates between n_sets = 2 sets of planes so that it never waits for a merge
    int merge_batch = 1, n_sets = 1;
    // merge stream (round 5, PTK_MERGE_STREAM): the merges of ALL lanes run on one stream of their own, in call order (= the order of that stream), in
    // the normal-priority pool where its hardware queue is shared with no lane (lane_stream_create); lanes alternate between two plane sets
    hipStream_t merge_stream = nullptr;
    hipEvent_t merge_done = nullptr;    // recorded behind the latest merge on merge_stream (join_on_lane0)
    bool merge_wait_resolved = false;   // the next merge must wait for lane 0's asynchronous reader of the image (image_reader_enqueued)
    int pend_first = 0, pend_n = 0, pend_set = 0;      // lanes pend_first .. pend_first + pend_n - 1 hold unmerged iterations in set pend_set
    unsigned long long* ray_total = nullptr;
    int P = 0, capacity = 0, traceDepth = 0;
    int n_parts = 1, part = 0, tile_rows = 1;
    int pipeline = 0, n_lanes = 1, spp = 1, n_pix = 0;      // n_lanes: EFFECTIVE lane count of the live context (iterations in flight, or bands with tiles in flight)
    int want_lanes = 1;                 // ptk_set_iterations_in_flight: the caller's knob, never overwritten (init / free / init with tiles in flight)
    bool use_graph = false;
    int last_lane = 0;
    int hint_mode = -1;                 // sample mode the lanes' full_bounces / gen_late hints were learnt under
    int want_tiles = 1;                 // ptk_set_tiles_in_flight
    bool tiles = false;                 // tiles in flight: the lanes are bands of rows of ONE iteration stream (see render_iteration)
    int n_shards = 1, shard = 0;        // iteration sharding (ptk_set_iteration_sharding): call `step` renders iteration (step-1)*n_shards + shard + 1
    bool live = false;
    // C++ API (pathtrace.h): scene->state.image is made current on demand (pathtraceResolve), not after every call
    void* host_image_reg = nullptr;     // state.image's storage, pinned in place (hipHostRegister) so that the D2H copy is one asynchronous DMA
    int image_policy = 0;               // C++ API: 0 = automatic (see pathtrace()), 1 = PTK_EAGER_IMAGE=1: resolve after every call as pathtrace.cu:809-810 does, 2 = PTK_LAZY_IMAGE=1: only pathtraceResolve()
    bool host_image_stale = false;      // a pathtrace() call was enqueued since scene->state.image was last made current
} g;

int g_mode_override = -1;               // ptk_set_sample_mode; -1 = follow the `sampleMode` global
int g_debug_slot_shift = 0, g_debug_slot_shift_deep = 0;      // test hooks (libptk_hooks.so only): offsets of the shading RNG key at depth 0 / depth >= 1; always 0 in the product

template <typename T> int dalloc(T** p, size_t count) {
    *p = nullptr;
    hipError_t e = hipMalloc((void**)p, sizeof(T) * std::max<size_t>(count, 1));
    return e == hipSuccess ? 0 : (int)e;
}
template <typename T> void dfree(T*& p) { if (p) (void)hipFree(p); p = nullptr; }
inline int rc_of(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

int alloc_paths(PtkPaths& b, int cap) {
    int rc;
    if ((rc = dalloc(&b.o_pix, 4 * (size_t)cap))) return rc;
    if ((rc = dalloc(&b.d_bnc, 4 * (size_t)cap))) return rc;
    return dalloc(&b.col_pdf, 4 * (size_t)cap);
}
void free_paths(PtkPaths& b) { dfree(b.o_pix); dfree(b.d_bnc); dfree(b.col_pdf); }

void sync_all() { for (Lane& l : g.lanes) if (l.stream) (void)hipStreamSynchronize(l.stream); if (g.merge_stream) (void)hipStreamSynchronize(g.merge_stream); }

int context_free() {
    sync_all();
    if (g_ht.on && g_ht.calls > 0)
        std::fprintf(stderr, "ptk host timing: %ld calls, %.1f us per call on the host: hipGraphLaunch %.1f (max %.0f), merge enqueue %.1f, rest %.1f\n", g_ht.calls, g_ht.call_us / g_ht.calls,
                     g_ht.graph_us / g_ht.calls, g_ht.graph_max, g_ht.merge_us / g_ht.calls, (g_ht.call_us - g_ht.graph_us - g_ht.merge_us) / g_ht.calls);
    g_ht = HostTiming(); g_ht.on = getenv("PTK_HOST_TIMING") && atoi(getenv("PTK_HOST_TIMING")) != 0;
    g_debug_slot_shift = g_debug_slot_shift_deep = 0;      // a test that died between set and reset must not shift the keys of later renders
    for (Lane& l : g.lanes) {
        free_paths(l.bufA); free_paths(l.bufB);
        dfree(l.hits.t_n); dfree(l.hits.p_mat); dfree(l.hits.uv_ids);
        dfree(l.hitsB.t_n); dfree(l.hitsB.p_mat); dfree(l.hitsB.uv_ids); dfree(l.hrA); dfree(l.hrB);
        dfree(l.ws.agg_h); dfree(l.ws.hit_pref); dfree(l.ws.tr_valid); dfree(l.ws.tr_heavy); dfree(l.ws.hcounts); dfree(l.ws.seg_first_h);
        dfree(l.ws.agg_a); dfree(l.ws.agg_b); dfree(l.ws.valid_a); dfree(l.ws.valid_b);
        dfree(l.ws.srcidx); dfree(l.ws.seg_pref); dfree(l.ws.seg_first); dfree(l.seg_done_mem); l.ws.seg_done = nullptr;
        dfree(l.ws.counts); dfree(l.ws.stats); dfree(l.ws.queue); dfree(l.ws.list); dfree(l.iter_dev); dfree(l.cum); dfree(l.ws.look);
        for (hipEvent_t& e : l.ring_ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
        for (Lane::GraphSlot& q : l.gs) { if (q.exec) (void)hipGraphExecDestroy(q.exec); if (q.graph) (void)hipGraphDestroy(q.graph); }
        for (hipEvent_t& e : l.set_free) { if (e) (void)hipEventDestroy(e); e = nullptr; }
        if (l.done) (void)hipEventDestroy(l.done);
        if (l.counts_ev) (void)hipEventDestroy(l.counts_ev);
        if (l.merged) (void)hipEventDestroy(l.merged);
        if (l.h_counts) (void)hipHostFree(l.h_counts);
        if (l.stream) (void)hipStreamDestroy(l.stream);
    }
    g.lanes.clear();
    if (g.merge_stream) { (void)hipStreamDestroy(g.merge_stream); g.merge_stream = nullptr; }
    if (g.merge_done) { (void)hipEventDestroy(g.merge_done); g.merge_done = nullptr; }
    g.merge_wait_resolved = false;
    if (g.host_image_reg) { (void)hipHostUnregister(g.host_image_reg); g.host_image_reg = nullptr; }
    if (g.image != g.planes) dfree(g.image);
    g.image = nullptr;
    dfree(g.planes); dfree(g.flags); dfree(g.ray_total);
    if (g.ev0) { (void)hipEventDestroy(g.ev0); g.ev0 = nullptr; }
    if (g.ev1) { (void)hipEventDestroy(g.ev1); g.ev1 = nullptr; }
    if (g.resolved) { (void)hipEventDestroy(g.resolved); g.resolved = nullptr; }
    g.last_merged = nullptr; g.ordered = false; g.n_depth = 1; g.merge_batch = 1; g.n_sets = 1; g.pend_n = 0;
    g.live = false; g.tiles = false; g.n_lanes = g.want_lanes; g.scene = nullptr; g.P = g.capacity = 0;
    return 0;
}

// BVH walk of the work-list rays (bit-identical hits either way, ptk_internal.h): the wide walk unless pipeline 5 asks for the
// reference's threaded walk.  Measured on the bunny scene, Grays/s, threaded / wide: one iteration at a time 2.47 / 3.28; three in
// flight 4.99 / 5.38 (with 128 staged nodes the wide walk's 25 KB of LDS per block lost there, 4.34; with 16 it does not).
// Window bookkeeping of the segmented streams: by the shading kernel's last block (pipeline 8, default with several iterations in
// flight: one launch less per bounce, +2 % on the headline) or by a separate one-block k_segments launch (pipeline 7, default with
// one iteration at a time, where the serial tail of that last block costs more than the launch: 0.352 vs 0.363 ms per iteration).
// Pipeline 0 / 10: the fused k_shade_trace launches (one per bounce); they always do their own window bookkeeping.
static bool trace_pipeline(int pipeline) { return pipeline == 0 || pipeline == 10; }
// pipeline 11 (round 5): k_trace_shade — trace AND shade a bounce per launch, the hit record never leaves the block (ptk_kernels.hip).  Triangle-free
// scenes without textures only; not with tiles in flight (the bands' cumulative hit counts ride on k_shade_trace), the per-kernel profiler's
// intersect / shade split or the traversal counters.  PTK_TRACE_SHADE=0|1 overrides what pipeline 0 picks (A/B runs).
// k_shade_trace_w (round 5): the fused compact kernel with one wave and one 64-slot window per unit of work.  PTK_WAVE_WINDOWS=0|1 (A/B runs).
static bool wave_windows_choice() {
    if (const char* e = getenv("PTK_WAVE_WINDOWS")) return atoi(e) != 0;
    return false;
}
static bool ts_pipeline_wanted(int pipeline) {
    if (pipeline == 11) return true;
    if (pipeline != 0) return false;
    static const int forced = getenv("PTK_TRACE_SHADE") ? atoi(getenv("PTK_TRACE_SHADE")) : -1;
    return forced == 1;
}
static bool fold_segments(int pipeline, int n_lanes) {
    if (pipeline == 7) return false;
    if (pipeline == 8 || pipeline == 11 || trace_pipeline(pipeline)) return true;
    return n_lanes > 1;
}
static int pick_lpr(int n_lanes) {
    static const int forced = getenv("PTK_WIDE_LPR") ? atoi(getenv("PTK_WIDE_LPR")) : 0;      // (A/B knob: 2 or 4)
    if (forced == 2 || forced == 4) return forced;
    return n_lanes > 1 ? 2 : 4;
}
static int pick_walk(int pipeline, int n_lanes) {
    (void)n_lanes;
    return pipeline == 5 ? PTK_WALK_THREADED : PTK_WALK_WIDE;
}

// The BVH walk inside the intersect kernel (pipelines 1 / 2) instead of the work-list stage (k_list + k_traverse_wide + k_finalize): three
// launches fewer per bounce.  Pipeline 0 picks it for small meshes, and then runs the fused launches with the walk in their trace phase (one
// launch per bounce).  Measured (tools/diag/inline_threshold.py, Grays/s, BSDF / MIS, work list vs in-kernel walk): obj_test.txt (cube.obj, 12
// triangles, BASELINE configs[2]) staged + work list 11.1, staged + inline walk 12.9, fused + inline walk 14.1; the same room with K scattered
// meshes: 36 triangles 10.7 / 9.2 vs 13.8 / 12.9; 288: 5.7 / 5.2 vs 6.6 / 6.5; 320 (sphere.obj): 8.2 / 7.4 vs 8.9 / 8.3; 1 280: 4.7 / 4.5 vs
// 5.1 / 4.9; 5 120: 2.9 / 2.8 vs 3.0 / 3.0 — and the other way round for ONE dense mesh that most rays enter (bunny scene, 10 000 triangles):
// 5.96 / 5.62 vs 4.69 / 4.20.  The triangle count is a crude stand-in for "how many rays walk deep"; 1 024 keeps every measured win and stays
// clear of the bunny class.  Identical results either way (PTK_INLINE_BVH_MAX_TRIS overrides it for A/B runs).
constexpr int kInlineBvhMaxTris = 1024;
bool inline_bvh_choice(int which) {
    if (which == 1) return true;
    static const int max_tris = getenv("PTK_INLINE_BVH_MAX_TRIS") ? atoi(getenv("PTK_INLINE_BVH_MAX_TRIS")) : kInlineBvhMaxTris;      // (A/B knob)
    return which == 0 && g.scene && g.scene->tempDevScene.tri_num > 0 && g.scene->tempDevScene.tri_num <= max_tris;
}

// The lanes' streams and the hardware queues behind them.  The HIP runtime maps a process's streams onto at most GPU_MAX_HW_QUEUES (4) hardware
// queues PER PRIORITY LEVEL, and streams that share a queue serialise behind each other's event waits (profiles/r4_tiles_hw_queues.txt).  The null
// stream and the host's own streams live in the normal-priority pool, so a fourth lane there lands on a queue another stream already uses.
// Round 5: the lanes are created in the HIGH-priority pool, whose four queues then belong to the lanes alone — what GPU_MAX_HW_QUEUES=8 bought, without
// an environment variable the host would have to set before its first HIP call.  Four lanes: 4.96 -> 6.56 Grays/s on the bunny scene (three: 6.14),
// mis_test 1024^2 14.4 -> 19.2 (three: 18.3), cornell 400^2 depth 4 12.9 -> 15.6; with three lanes nothing changes (profiles/r5_ab_lane_queues.txt).
// PTK_LANE_PRIORITY (A/B knob): 1 (default) = as above; 0 = every lane at normal priority (rounds 2-4); 2 = lanes alternate between the two pools (up
// to eight queues; measured worse: the pools are served with different priority).
hipError_t lane_stream_create(hipStream_t* s, int idx) {
    static const int policy = getenv("PTK_LANE_PRIORITY") ? atoi(getenv("PTK_LANE_PRIORITY")) : 1;
    if (policy == 0) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e != hipSuccess) return e;
    int prio = greatest;
    if (policy == 2) prio = (idx & 1) == 0 ? greatest : 0;
    if (policy == 3) prio = idx % 3 == 0 ? greatest : (idx % 3 == 1 ? 0 : least);      // three pools: up to twelve queues
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, prio);
}

int lane_init(Lane& l, int idx) {
    int rc;
    hipError_t e = lane_stream_create(&l.stream, idx);
    if (e != hipSuccess) return (int)e;
    if ((e = hipEventCreateWithFlags(&l.done, hipEventDisableTiming)) != hipSuccess) return (int)e;
    if ((e = hipEventCreateWithFlags(&l.counts_ev, hipEventDisableTiming)) != hipSuccess) return (int)e;
    if ((e = hipEventCreateWithFlags(&l.merged, hipEventDisableTiming)) != hipSuccess) return (int)e;
    if ((e = hipHostMalloc((void**)&l.h_counts, sizeof(int) * 2 * ((size_t)g.traceDepth + 2), hipHostMallocDefault)) != hipSuccess) return (int)e;      // counts, then hcounts
    const Camera& cam = g.scene->state.camera;
    if (g.tiles) {
        // tiles in flight: lane idx renders the idx-th band of rows of EVERY iteration into the one shared plane; its paths are a contiguous
        // range of the reference's pixel order, so its compacted slots continue those of the bands above (Workspace::cum_*)
        l.n_parts = g.n_lanes; l.part = idx; l.tile_rows = (cam.resolution.y + g.n_lanes - 1) / g.n_lanes;
        l.plane = g.planes;
        l.ws.slot_bias = 0;
        if ((rc = dalloc(&l.cum, (size_t)PTK_CUM_RING * ((size_t)g.traceDepth + 2)))) return rc;
        if ((e = hipMemsetAsync(l.cum, 0, sizeof(int) * PTK_CUM_RING * ((size_t)g.traceDepth + 2), l.stream)) != hipSuccess) return (int)e;
        for (hipEvent_t& ev : l.ring_ev) if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return (int)e;
    } else {
        l.n_parts = g.n_parts; l.part = g.part; l.tile_rows = g.tile_rows;
        // planes / flags: [set][lane][sample][bounce]: the lanes of one set are contiguous in lane order = call order within a round of calls
        for (int k = 0; k < g.n_sets; ++k) {
            l.plane_sets[k] = g.planes + ((size_t)k * g.n_lanes + idx) * g.spp * g.n_depth * 3 * (size_t)g.P;
            l.flag_sets[k] = g.flags ? g.flags + ((size_t)k * g.n_lanes + idx) * g.spp * g.n_depth * ((g.P + 63) / 64) : nullptr;
            if (g.n_sets > 1 && (e = hipEventCreateWithFlags(&l.set_free[k], hipEventDisableTiming)) != hipSuccess) return (int)e;
        }
        l.plane = l.plane_sets[0]; l.flags = l.flag_sets[0];
        // under the tile partition every rank used the same (iter, slot, depth) keys; make them rank-unique
        l.ws.slot_bias = g.n_parts > 1 ? g.part * g.P * g.spp : 0;          // fits an int: checked in context_init
    }
    l.n_pix = local_path_count(cam.resolution.x, cam.resolution.y, l.n_parts, l.part, l.tile_rows);
    l.capacity = l.n_pix * g.spp;
    l.full_bounces = g.traceDepth;                  // until live counts of earlier iterations say otherwise (note_counts)
    l.gen_late = 0;
    l.ws.plane_stride = g.spp > 1 ? (unsigned)g.n_depth * 3u * (unsigned)g.P : 0u;      // 0: no sample-plane bits in pixelIndex (film_add, ptk_kernels.hip)
    if (!g.tiles) {
        const unsigned tiles = (unsigned)((g.P + 63) / 64);
        l.ws.depth_stride = g.n_depth > 1 ? 3u * (unsigned)g.P : 0u;
        l.ws.film_flags = l.flags; l.ws.flag_sample_stride = (unsigned)g.n_depth * tiles; l.ws.flag_depth_stride = g.n_depth > 1 ? tiles : 0u;
    }
    l.ws.spp = g.spp;
    if ((rc = alloc_paths(l.bufA, l.capacity))) return rc;
    if ((rc = alloc_paths(l.bufB, l.capacity))) return rc;
    if ((rc = dalloc(&l.hits.t_n, 4 * (size_t)l.capacity))) return rc;
    if ((rc = dalloc(&l.hits.p_mat, 4 * (size_t)l.capacity))) return rc;
    if ((rc = dalloc(&l.hits.uv_ids, 4 * (size_t)l.capacity))) return rc;
    if ((rc = dalloc(&l.hitsB.t_n, 4 * (size_t)l.capacity))) return rc;
    if ((rc = dalloc(&l.hitsB.p_mat, 4 * (size_t)l.capacity))) return rc;
    if ((rc = dalloc(&l.hitsB.uv_ids, 4 * (size_t)l.capacity))) return rc;
    const size_t slots = ((size_t)l.capacity + 255) / 256 * 256 + 256;          // physical slots: whole 256-slot windows
    if ((rc = dalloc(&l.hrA, slots))) return rc;
    if ((rc = dalloc(&l.hrB, slots))) return rc;
    if ((rc = dalloc(&l.ws.tr_valid, slots))) return rc;
    if ((rc = dalloc(&l.ws.tr_heavy, slots))) return rc;
    const size_t blocks = ((size_t)l.capacity + 255) / 256 + 2;
    const size_t wins = ((size_t)l.capacity + 63) / 64 + 8;               // window tables are sized for 64-slot windows (k_shade_trace_w); 256-slot windows use a quarter
    if ((rc = dalloc(&l.ws.agg_h, wins))) return rc;
    if ((rc = dalloc(&l.ws.hit_pref, wins + 16))) return rc;
    if ((rc = dalloc(&l.ws.seg_first_h, wins + 16))) return rc;
    if ((rc = dalloc(&l.ws.hcounts, (size_t)g.traceDepth + 2))) return rc;
    if ((rc = dalloc(&l.ws.agg_a, blocks))) return rc;
    if ((rc = dalloc(&l.ws.agg_b, wins))) return rc;
    if ((rc = dalloc(&l.ws.valid_a, (size_t)l.capacity))) return rc;
    if ((rc = dalloc(&l.ws.valid_b, (size_t)l.capacity))) return rc;
    if ((rc = dalloc(&l.ws.counts, (size_t)g.traceDepth + 2))) return rc;
    if ((rc = dalloc(&l.ws.queue, 1))) return rc;
    if ((rc = dalloc(&l.ws.list, (size_t)l.capacity))) return rc;
    if ((rc = dalloc(&l.iter_dev, 1))) return rc;
    if ((rc = dalloc(&l.ws.srcidx, (size_t)l.capacity))) return rc;
    if ((rc = dalloc(&l.ws.seg_pref, wins + 16))) return rc;
    if ((rc = dalloc(&l.ws.seg_first, wins + 16))) return rc;
    // pipeline 11's look-back tables: one tagged word per block (never cleared) + one accumulating word per group of 32 blocks behind them (zero here,
    // cleared again by every launch's last block)
    if ((rc = dalloc(&l.ws.look, 2 * blocks + 64))) return rc;
    if ((e = hipMemsetAsync(l.ws.look, 0, sizeof(unsigned long long) * (2 * blocks + 64), l.stream)) != hipSuccess) return (int)e;
    l.ws.look_gave_up = l.ws.look + (2 * blocks + 63);            // (the last word of the table: raised by a block that gives its wait up)
    if (const char* v = std::getenv("PTK_LB_POLL_LIMIT")) l.ws.look_limit = std::max(0, std::atoi(v));      // (tests: 0 = never wait)
    const size_t n_done = 32 * ((size_t)blocks / 32 + 3);          // top counter + one per 32 blocks, each on its own 128-B line (segments_tail)
    if ((rc = dalloc(&l.seg_done_mem, n_done))) return rc;
    if ((e = hipMemsetAsync(l.seg_done_mem, 0, sizeof(int) * n_done, l.stream)) != hipSuccess) return (int)e;
    l.ws.seg_done = fold_segments(g.pipeline, g.n_lanes) ? l.seg_done_mem : nullptr;
    if ((e = hipMemsetAsync(l.ws.queue, 0, sizeof(unsigned), l.stream)) != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(l.ws.counts, 0, sizeof(int) * ((size_t)g.traceDepth + 2), l.stream)) != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(l.ws.hcounts, 0, sizeof(int) * ((size_t)g.traceDepth + 2), l.stream)) != hipSuccess) return (int)e;
    l.ws.ray_total = g.ray_total;
    l.ws.inline_bvh = inline_bvh_choice(g.pipeline);
    l.ws.walk = pick_walk(g.pipeline, g.n_lanes);
    l.ws.wide_lpr = pick_lpr(g.n_lanes);            // see k_traverse_wide: pairs when the machine is full, quads for the shortest chain
    l.ws.wide_pool = pick_wide_pool(l.capacity, g.tiles ? 1 : g.n_lanes);
    l.ws.wave_windows = wave_windows_choice();
    l.ws.capacity = l.capacity;
    l.last_iter = 0;
    return 0;
}

int current_mode();

int context_init(Scene* scene) {
    if (!scene || !scene->ok) return PTK_E_NOSCENE;
    if (!scene->tempDevScene.uploaded) { scene->setDevData(); if (scene->devError) return scene->devError; }
    context_free();
    const Camera& cam = scene->state.camera;
    g.scene = scene;
    g.P = cam.resolution.x * cam.resolution.y;
    g.traceDepth = scene->state.traceDepth;
    g.hint_mode = -1;
    if (g.traceDepth < 1 || g.P < 1) return PTK_E_BADARG;
    g.n_pix = local_path_count(cam.resolution.x, cam.resolution.y, g.n_parts, g.part, g.tile_rows);
    g.capacity = g.n_pix * g.spp;
    if (g.spp > 1 && g.P >= (1 << 24)) return PTK_E_BADARG;      // samples per launch: pixelIndex carries the sample plane in bits 24..30
    if (g.P > (1 << 29)) return PTK_E_BADARG;                    // 32-bit float offsets into a plane
    // the rank-unique RNG key is slot + part * P * spp as a 32-bit int; refuse partitions whose largest key would wrap
    if (g.n_parts > 1 && (long long)g.part * g.P * g.spp + (long long)g.capacity > 2147483647LL) return PTK_E_BADARG;
    int rc;
    hipError_t e;
    if ((e = hipEventCreate(&g.ev0)) != hipSuccess) return (int)e;
    if ((e = hipEventCreate(&g.ev1)) != hipSuccess) return (int)e;
    if ((e = hipEventCreateWithFlags(&g.resolved, hipEventDisableTiming)) != hipSuccess) return (int)e;
    // tiles in flight: as many lanes as there are non-empty bands of ceil(H / K) rows; one shared radiance plane
    g.tiles = g.want_tiles > 1;
    g.n_lanes = g.want_lanes;
    if (g.tiles) {
        if (g.want_lanes != 1 || g.n_parts != 1 || g.spp != 1) return PTK_E_BADARG;       // (not combined with iterations in flight / the multi-GPU tile partition / samples per launch)
        if (!trace_pipeline(g.pipeline)) return PTK_E_BADARG;                           // (only the fused launches carry the bands' cumulative hit counts)
        const int rows = (cam.resolution.y + g.want_tiles - 1) / g.want_tiles;
        g.n_lanes = (cam.resolution.y + rows - 1) / rows;
        if (g.n_lanes < 2) { g.tiles = false; g.n_lanes = 1; }
    }
    const size_t plane = 3 * (size_t)g.P;
    // Accumulation (flush_merges): with several planes the default is ORDERED — one plane per sample and, when the context is initialised in MIS mode,
    // per bounce, so that every (pixel, plane) receives one add and the merge can repeat the reference's own sequence of adds
    const size_t n_basic = g.tiles ? 1 : (size_t)g.n_lanes * g.spp;
    g.ordered = n_basic > 1 && g.want_ordered != 0;
    g.n_depth = (g.ordered && current_mode() == PTK_MODE_MIS && g.spp == 1) ? g.traceDepth + 1 : 1;      // (samples per launch: one plane per sample)
    if ((unsigned long long)g.n_depth * plane >= (1ull << 31)) g.n_depth = 1;       // (sample stride must fit 31 bits: then MIS merges the iteration's own sum)
    // batched merges (PTK_MERGE_BATCH=<n>, n <= lanes): OFF by default — built and measured in round 5 (profiles/r5_ab_batched_merge.txt), bit-identical
    // (the whole GPU tier ran with it on) and slower: cornell 21.2 -> 20.0, mis_test 1024^2 19.3 -> 16.5, cornell 400^2 12.7 -> 10.8 Grays/s.  The
    // round-4 upper bound (+2..4 %, two of three merges simply skipped) left out what the real thing costs: the batched merge needs ALL the round's
    // iterations done, so the stream it runs on stalls until the slowest lane of the round has finished, and that lane's next iteration queues behind it.
    g.merge_batch = 1; g.n_sets = 1;
    if (g.ordered && !g.tiles && g.n_lanes > 1) {
        const char* mb = getenv("PTK_MERGE_BATCH");
        g.merge_batch = mb ? std::max(1, std::min(atoi(mb), g.n_lanes)) : 1;
        if (g.merge_batch > 1) g.n_sets = 2;
        const char* ms = getenv("PTK_MERGE_STREAM");
        if (g.merge_batch == 1 && ms && atoi(ms) != 0) {
            g.n_sets = 2;
            if ((e = hipStreamCreateWithFlags(&g.merge_stream, hipStreamNonBlocking)) != hipSuccess) return (int)e;
            if ((e = hipEventCreateWithFlags(&g.merge_done, hipEventDisableTiming)) != hipSuccess) return (int)e;
        }
    }
    g.pend_n = 0;
    const size_t n_planes = n_basic * g.n_depth * g.n_sets;
    if ((rc = dalloc(&g.planes, plane * n_planes))) return rc;
    if ((e = hipMemset(g.planes, 0, sizeof(float) * plane * n_planes)) != hipSuccess) return (int)e;
    if (n_basic > 1) {
        if ((rc = dalloc(&g.image, plane))) return rc;
        if ((e = hipMemset(g.image, 0, sizeof(float) * plane)) != hipSuccess) return (int)e;
    } else g.image = g.planes;
    if (g.ordered) {
        const size_t n_flags = n_planes * (((size_t)g.P + 63) / 64);
        if ((rc = dalloc(&g.flags, n_flags))) return rc;
        if ((e = hipMemset(g.flags, 0, n_flags)) != hipSuccess) return (int)e;
    }
    g.last_merged = nullptr;
    if ((rc = dalloc(&g.ray_total, 1))) return rc;
    if ((e = hipMemset(g.ray_total, 0, sizeof(unsigned long long))) != hipSuccess) return (int)e;
    if (g.pipeline == 5 && scene->tempDevScene.n_copies != 6 && (rc = scene->tempDevScene.upload_threaded(*scene, 6))) return rc;
    g.lanes.assign(g.n_lanes, Lane());
    for (int i = 0; i < g.n_lanes; ++i) if ((rc = lane_init(g.lanes[i], i))) return rc;
    g.last_lane = 0;
    // the two hipMemset above run on the NULL stream and the lanes' streams are non-blocking: nothing else orders the clears before the
    // first iteration's radiance adds (seen once with two processes sharing a GPU: a few pixels of the first iterations lost)
    if ((e = hipDeviceSynchronize()) != hipSuccess) return (int)e;
    g.live = true;
    return 0;
}

int current_mode() { return g_mode_override >= 0 ? g_mode_override : (int)sampleMode; }

// optional per-kernel timing (bench.py's roofline leg): HIP events on the render stream around every launch
struct KernelProfile {
    bool on = false;
    hipStream_t stream = nullptr;
    std::vector<hipEvent_t> ev;          // pairs
    std::vector<int> cls;                // kernel class per pair
} g_prof;

void prof_mark(int cls, bool begin) {
    if (!g_prof.on) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, g_prof.stream);
    g_prof.ev.push_back(e);
    if (begin) g_prof.cls.push_back(cls);
}

// k_tail finishes the iterations of every pipeline but 1 (and never under the per-kernel profiler or with the traversal counters on)
bool sparse_tail() { return g.pipeline != 1 && !g_prof.on &&
                            !(g.lanes.size() && g.lanes[0].ws.stats); }
// the fused shade + trace launches: pipeline 10 always, pipeline 0 (auto) for triangle-free scenes — with a BVH stage behind every launch
// the staged kernels are 1-3 % ahead (bunny scene 5.70 vs 5.62 Grays/s); the traversal counters are only kept by the staged kernels
// ONE ITERATION AT A TIME ONLY.  A block of k_trace_shade spins until every earlier block of its launch has published its hit count.  Alone on the GPU that
// cannot deadlock: each XCD hands out its share of the workgroups in index order, so the lowest unfinished block is always resident and needs nothing.  Beside
// OTHER launches whose blocks wait the same way it can: the XCDs dispatch independently, launch A's waiting blocks can fill the XCD that launch B's next block
// needs and vice versa — observed as a hang with four and more iterations in flight (round 5; the GPU had to be reclaimed).  With several lanes the context
// therefore renders with the default kernels (bit-identical images either way).
bool use_ts_pipeline() {
    return ts_pipeline_wanted(g.pipeline) && !g.tiles && g.n_lanes == 1 && g.scene && trace_shade_usable(dev_view(*g.scene)) && !(g.lanes.size() && g.lanes[0].ws.stats);
}
bool use_trace_pipeline() {
    if (use_ts_pipeline()) return false;
    if (!(trace_pipeline(g.pipeline) || (g.pipeline == 11 && g.n_lanes > 1)) || (g.lanes.size() && g.lanes[0].ws.stats)) return false;      // (11 with several lanes: the fused default)
    if (g.pipeline == 10 || g.tiles || g.scene->tempDevScene.tri_num == 0 || inline_bvh_choice(g.pipeline)) return true;      // (tiny mesh: walked inside the fused kernel; tiles in flight: only the fused launches carry the bands' cumulative hit counts)
    // larger meshes: a BVH stage follows every launch either way.  The staged kernels are ahead in BSDF / DIRECT (bunny scene 5.93 vs 5.85, 7.8 vs
    // 7.0 Grays/s); in MIS the fused kernel is, since its shadow rays run in a balanced phase at six waves per SIMD (5.69 vs 5.55)
    return current_mode() == PTK_MODE_MIS;
}
// the fused launches shade hit-dense unless a BVH stage follows them (SPLIT: meshes too large for the in-kernel walk)
bool hit_dense_pipeline() {
    return use_trace_pipeline() && !(g.scene->tempDevScene.tri_num > 0 && !inline_bvh_choice(g.pipeline));
}
void no_mark(int, bool) {}

int enqueue_iteration(Lane& L, int iter, int mode, bool from_device_iter) {
    const Camera& cam = g.scene->state.camera;
    const SceneView S = dev_view(*g.scene);
    Workspace ws = L.ws;
    ws.iter_dev = from_device_iter ? L.iter_dev : nullptr;
    ws.iter_stride = (g.tiles ? 1 : g.n_lanes) * g.n_shards;       // (tiles in flight: every lane renders every iteration)
    ws.dbg_shift0 = g_debug_slot_shift; ws.dbg_shift_deep = g_debug_slot_shift_deep;
    ws.gen_late = L.gen_late;
    // ordered accumulation: the lane's planes are zero when the launch starts and a (pixel, plane) receives at most one add — a plain store (MIS
    // without planes per bounce — the context was initialised in another mode — adds bounce by bounce into the one plane)
    if (g.ordered && (g.n_depth > 1 || mode != PTK_MODE_MIS)) ws.plane_stride |= PTK_FILM_STORE;
    if (g.tiles) {
        const int li = (int)(&L - &g.lanes[0]);
        ws.cum_prev = li > 0 ? g.lanes[li - 1].cum : nullptr; ws.cum_out = L.cum; ws.cum_stride = g.traceDepth + 2; ws.ring_div = g.n_shards;
    }
    const int full = sparse_tail() ? std::min(std::max(L.full_bounces, 1), g.traceDepth) : g.traceDepth;
    if (use_ts_pipeline()) {
        // launch d traces and shades bounce d (launch 0 generates the camera rays); the survivors ping-pong between the two path buffers
        void (*mark)(int, bool) = g_prof.on ? prof_mark : no_mark;
        PtkPaths cur = L.bufA, nxt = L.bufB;
        int rc = 0;
        for (int d = 0; d < full && !rc; ++d) {
            rc = launch_trace_shade(S, mode, cam, L.n_parts, L.part, L.tile_rows, L.n_pix, g.traceDepth, iter, d, L.capacity, cur, nxt, L.plane, ws, L.stream, d == g.traceDepth - 1, mark);
            std::swap(cur, nxt);
        }
        if (!rc && full < g.traceDepth) {            // every remaining bounce in one single-workgroup launch (k_tail), as after the staged kernels
            prof_mark(1, true);
            rc = launch_tail(S, mode, iter, full, g.traceDepth, cur, nxt, L.plane, ws, L.stream);
            prof_mark(1, false);
        }
        return rc;
    }
    if (use_trace_pipeline()) {
        // launch 0 generates the camera rays and traces bounce 0; launch dt shades bounce dt - 1 and traces bounce dt (k_shade_trace)
        void (*mark)(int, bool) = g_prof.on ? prof_mark : no_mark;
        TraceBufs A = {L.bufA, L.hits, L.hrA}, B = {L.bufB, L.hitsB, L.hrB};
        TraceBufs *cur = &A, *nxt = &B;
        int rc = launch_shade_trace(S, mode, true, cam, L.n_parts, L.part, L.tile_rows, L.n_pix, g.traceDepth, iter, 0, L.capacity, A, A, L.plane, ws, L.stream, false, mark);
        for (int dt = 1; dt <= full && !rc; ++dt) {
            rc = launch_shade_trace(S, mode, false, cam, L.n_parts, L.part, L.tile_rows, L.n_pix, g.traceDepth, iter, dt, L.capacity, *cur, *nxt, L.plane, ws, L.stream,
                                    dt == g.traceDepth, mark);
            std::swap(cur, nxt);
        }
        if (!rc && full < g.traceDepth) {            // the sparse rest of the iteration in one single-workgroup launch; bounce `full` is traced already
            prof_mark(1, true);
            rc = launch_tail(S, mode, iter, full, g.traceDepth, cur->paths, nxt->paths, L.plane, ws, L.stream, &cur->hits, cur->hr);
            prof_mark(1, false);
        }
        return rc;
    }
    prof_mark(0, true);
    int rc = launch_generate(cam, iter, ws.iter_dev, g.traceDepth, L.bufA, L.n_parts, L.part, L.tile_rows, L.n_pix, g.spp, ws.counts, L.stream);
    prof_mark(0, false);
    if (rc) return rc;
    PtkPaths cur = L.bufA, nxt = L.bufB;
    for (int depth = 0; depth < full; ++depth) {
        const bool last = depth == g.traceDepth - 1;      // false for every full-width bounce when k_tail follows: it steps the iteration number
        if (g_prof.on) rc = launch_bounce_staged_profiled(S, mode, iter, depth, L.capacity, cur, nxt, L.hits, L.plane, ws, L.stream, prof_mark, last);
        else rc = launch_bounce_staged(S, mode, iter, depth, L.capacity, cur, nxt, L.hits, L.plane, ws, L.stream, last);
        if (rc) return rc;
        PtkPaths t = cur; cur = nxt; nxt = t;
    }
    if (full < g.traceDepth) {                   // every remaining bounce in one single-workgroup launch (k_tail)
        prof_mark(1, true);
        rc = launch_tail(S, mode, iter, full, g.traceDepth, cur, nxt, L.plane, ws, L.stream);
        prof_mark(1, false);
    }
    return rc;
}

// Sparse tail (k_tail, ptk_kernels.hip).  How many bounces are worth a full-width launch is learnt from the live counts of
// EARLIER iterations: after every iteration counts[] is copied to pinned host memory asynchronously and looked at, without
// waiting, before a later enqueue on the same lane.  Results never depend on it — k_tail is correct for any live count.
constexpr int kTailPaths = 1024;                 // a bounce with at most this many live paths goes to the tail (one chunk of k_tail)
void note_counts(Lane& L) {
    if (!L.counts_pending || hipEventQuery(L.counts_ev) != hipSuccess) return;
    L.counts_pending = false;
    if (g.gui) {                                                       // pathtrace.cu:793-796, one read-back late
        int d = 0; while (d < g.traceDepth && L.h_counts[d] > 0) ++d;
        L.traced_depth = d;
        if (g.tiles) for (const Lane& o : g.lanes) d = std::max(d, o.traced_depth);      // a band is a part of the frame: the frame's depth is the deepest band's
        g.gui->TracedDepth = d;
    }
    if (!sparse_tail()) return;
    // what k_tail looks at in its first bounce: the live paths — or, after a hit-dense launch, only the paths that hit something (mis_test.txt:
    // 1 088 live but 43 hits at bounce 2: the tail takes over one 20-us launch earlier)
    const int* cnt = hit_dense_pipeline() ? L.h_counts + g.traceDepth + 2 : L.h_counts;
    int d_new = g.traceDepth;
    for (int d = 1; d < g.traceDepth; ++d) if (cnt[d] <= kTailPaths) { d_new = d; break; }
    // hysteresis: widen at once, narrow only when the count is clearly below the threshold (no graph re-capture ping-pong)
    if (d_new > L.full_bounces || (d_new < L.full_bounces && cnt[d_new] <= kTailPaths / 2)) L.full_bounces = d_new;
    // camera rays that hit something / camera rays (fused pipeline, compact streams): below ~half, the camera launch stores the path state of
    // the hits only, after its trace phase (k_shade_trace, GenArgs::late_stores); hysteresis as above.  Results never depend on it.
    // (only the hit-dense launches write hcounts[]: after a SPLIT launch h0 is a stale zero and would read as "open scene" — ADVICE r3)
    const long long n0 = L.h_counts[0], h0 = L.h_counts[g.traceDepth + 2];
    if (n0 > 0 && hit_dense_pipeline()) {
        if (!L.gen_late && 100 * h0 < 45 * n0) L.gen_late = 1;
        else if (L.gen_late && 100 * h0 > 55 * n0) L.gen_late = 0;
    }
}
int request_counts(Lane& L) {
    if ((!sparse_tail() && !g.gui) || L.counts_pending) return 0;      // (the viewer's TracedDepth rides on the same read-back)
    // the counts of a static scene barely move: look every iteration at first, then at every 16th (each look is a ~5-us copy kernel on the stream)
    if (++L.fresh_hint > 8 && L.fresh_hint % 16 != 0) return 0;
    hipError_t e = hipMemcpyAsync(L.h_counts, L.ws.counts, sizeof(int) * ((size_t)g.traceDepth + 1), hipMemcpyDeviceToHost, L.stream);
    if (e != hipSuccess) return (int)e;
    if ((e = hipMemcpyAsync(L.h_counts + g.traceDepth + 2, L.ws.hcounts, sizeof(int) * ((size_t)g.traceDepth + 1), hipMemcpyDeviceToHost, L.stream)) != hipSuccess) return (int)e;
    if ((e = hipEventRecord(L.counts_ev, L.stream)) != hipSuccess) return (int)e;
    L.counts_pending = true;
    return 0;
}

// Replay path: the launches of one iteration are captured once per lane into a hipGraph whose kernels read the iteration
// number from L.iter_dev (stepped by n_lanes at the end of every replay), so a replay needs no argument patching.
// returns 0, a launch error of the captured kernels (> 0 or PTK_E_*), or -1000 when the graph machinery itself failed
int build_graph(Lane& L, int iter, int mode) {
    hipError_t e;
    // a recapture happens a handful of times per render (mode switch, full_bounces hint): let the lane's queued replays of the
    // old exec finish before it is destroyed (whether the runtime defers the destroy itself is not documented)
    Lane::GraphSlot& Q = L.gs[L.cur_set];
    if (Q.exec && hipStreamSynchronize(L.stream) != hipSuccess) return -1000;
    if (Q.exec) { (void)hipGraphExecDestroy(Q.exec); Q.exec = nullptr; }
    if (Q.graph) { (void)hipGraphDestroy(Q.graph); Q.graph = nullptr; }
    if ((e = hipStreamBeginCapture(L.stream, hipStreamCaptureModeThreadLocal)) != hipSuccess) return -1000;
    const int rc = enqueue_iteration(L, iter, mode, true);
    e = hipStreamEndCapture(L.stream, &Q.graph);
    if (rc) { if (Q.graph) { (void)hipGraphDestroy(Q.graph); Q.graph = nullptr; } return rc; }
    if (e != hipSuccess || !Q.graph) { Q.graph = nullptr; return -1000; }
    if ((e = hipGraphInstantiate(&Q.exec, Q.graph, nullptr, nullptr, 0)) != hipSuccess) {
        (void)hipGraphDestroy(Q.graph); Q.graph = nullptr; Q.exec = nullptr;
        return -1000;
    }
    return 0;
}

int replay_iteration(Lane& L, int iter, int mode) {
    hipError_t e;
    Lane::GraphSlot& Q = L.gs[L.cur_set];
    if (!Q.exec || Q.mode != mode || Q.full != L.full_bounces || Q.late != L.gen_late) {
        const int rc = build_graph(L, iter, mode);
        if (rc == -1000) {
            // capture / instantiate refused by the runtime: same kernels, launched one by one (identical results)
            (void)hipGetLastError();
            std::fprintf(stderr, "ptk: hipGraph capture failed, falling back to eager launches\n");
            g.use_graph = false;
            L.expected_iter = -1;
            return enqueue_iteration(L, iter, mode, false);
        }
        if (rc) return rc;
        Q.mode = mode; Q.full = L.full_bounces; Q.late = L.gen_late;
        L.expected_iter = -1;
    }
    if (L.expected_iter != iter) {           // first replay, or the caller jumped in the iteration sequence
        if ((e = hipMemcpyAsync(L.iter_dev, &iter, sizeof(int), hipMemcpyHostToDevice, L.stream)) != hipSuccess) return (int)e;
        if ((e = hipStreamSynchronize(L.stream)) != hipSuccess) return (int)e;     // &iter is a stack address
    }
    const double t0 = g_ht.on ? now_us() : 0.0;
    if ((e = hipGraphLaunch(Q.exec, L.stream)) != hipSuccess) return (int)e;
    if (g_ht.on) { const double d = now_us() - t0; g_ht.graph_us += d; if (d > g_ht.graph_max) g_ht.graph_max = d; }
    L.expected_iter = iter + (g.tiles ? 1 : g.n_lanes) * g.n_shards;
    return 0;
}

static void reset_hints() { for (Lane& l : g.lanes) { l.full_bounces = g.traceDepth; l.gen_late = 0; l.counts_pending = false; l.fresh_hint = 0; } }

// ORDERED ACCUMULATION (round 4).  With several planes (iterations in flight, samples per launch) a launch adds its radiance into the lane's own
// planes, zero when it starts — one per sample; one per BOUNCE when the context was initialised in MIS mode — so that every (pixel, plane) receives
// at most one add: BSDF / DIRECT add once per pixel and iteration (pathtrace.cu:398,486 or the ENV add of :303-311), MIS once per pixel and bounce
// (:556 or :572 or the ENV add).  This merge — on the lane's stream, behind the launch — folds them into the image in the reference's order of
// adds, iteration by iteration, bounce by bounce: it waits for the merge of the previous call (another lane's stream; one event edge, no host
// wait).  image = ((image + a_1) + a_2) + ...: the reference's running sum (pathtrace.cu:600), bit-identical to one iteration at a time in every
// sample mode (ptk_kernels.hip: k_merge_planes; touched-tile flags keep the sparse adds of deep bounces cheap).  The lane's next launch follows
// the merge on the same stream and finds its planes zero again.  Exceptions, both valid images with another float association: samples per launch
// in MIS (one plane per sample holds the sample's own sum) and a context initialised in BSDF / DIRECT that is switched to MIS without
// re-initialising.  Measured and dropped: the merges on a stream of their own, so that no lane waits for another — a fourth stream beside three
// lanes shares a hardware queue with one of them, cornell 21.9 -> 15.2 Grays/s.  Cost against planes that accumulate
// (ptk_set_ordered_accumulation(0); profiles/r4_ab_ordered_accumulation.txt): cornell BSDF -3 %, MIS -5 %, mis_test 1024^2 MIS -7 %, BSDF -10 %.
// BATCHED (round 5; opt-in, measured slower — see context_init): with merge_batch = L > 1 the merges of a round of L consecutive calls are ONE launch — it reads the L lanes'
// planes of the round (contiguous: [set][lane][sample][bounce]) in call order and touches the image once per pixel instead of L times; a lane renders
// alternately into two sets of planes, so its next iteration never waits for the merge of its previous one (only, two calls later, for the merge that
// zeroed the set it is about to reuse — long done).  The sequence of adds into every pixel is unchanged: bit-identical images.
static int flush_merges() {
    if (!g.ordered || g.pend_n == 0) return 0;
    hipError_t e;
    if (g.merge_stream) {
        // one iteration per merge, every merge on the merge stream: its order IS the call order; the lane goes on with its other plane set at once
        Lane& L = g.lanes[g.pend_first];
        if ((e = hipEventRecord(L.done, L.stream)) != hipSuccess) return (int)e;
        if ((e = hipStreamWaitEvent(g.merge_stream, L.done, 0)) != hipSuccess) return (int)e;
        if (g.merge_wait_resolved) { if ((e = hipStreamWaitEvent(g.merge_stream, g.resolved, 0)) != hipSuccess) return (int)e; g.merge_wait_resolved = false; }
        const int rc = launch_merge_planes(g.image, L.plane_sets[g.pend_set], L.flag_sets[g.pend_set], g.spp * g.n_depth, g.P, g.merge_stream);
        if (rc) return rc;
        if ((e = hipEventRecord(L.set_free[g.pend_set], g.merge_stream)) != hipSuccess) return (int)e;
        L.set_busy[g.pend_set] = true;
        g.pend_n = 0;
        return 0;
    }
    Lane& M = g.lanes[g.pend_first + g.pend_n - 1];          // the merge runs behind the last iteration it folds, on that lane's stream
    for (int k = 0; k < g.pend_n - 1; ++k) {
        Lane& o = g.lanes[g.pend_first + k];
        if ((e = hipEventRecord(o.done, o.stream)) != hipSuccess) return (int)e;
        if ((e = hipStreamWaitEvent(M.stream, o.done, 0)) != hipSuccess) return (int)e;
    }
    if (g.last_merged && g.last_merged != M.merged && (e = hipStreamWaitEvent(M.stream, g.last_merged, 0)) != hipSuccess) return (int)e;
    Lane& F = g.lanes[g.pend_first];
    const int rc = launch_merge_planes(g.image, F.plane_sets[g.pend_set], F.flag_sets[g.pend_set], g.pend_n * g.spp * g.n_depth, g.P, M.stream);
    if (rc) return rc;
    if ((e = hipEventRecord(M.merged, M.stream)) != hipSuccess) return (int)e;
    g.last_merged = M.merged;
    if (g.n_sets > 1)
        for (int k = 0; k < g.pend_n; ++k) {
            Lane& o = g.lanes[g.pend_first + k];
            if ((e = hipEventRecord(o.set_free[g.pend_set], M.stream)) != hipSuccess) return (int)e;
            o.set_busy[g.pend_set] = true;
        }
    g.pend_n = 0;
    return 0;
}
// the iteration just enqueued on lane `li` (into its current set) joins the pending batch; the batch is merged when it is full, at the end of a round
// of lanes, or when the next call would not continue it (callers that jump in the call sequence)
static int note_unmerged(int li) {
    if (!g.ordered) return 0;
    Lane& L = g.lanes[li];
    if (g.pend_n > 0 && (li != g.pend_first + g.pend_n || L.cur_set != g.pend_set)) { const int rc = flush_merges(); if (rc) return rc; }
    if (g.pend_n == 0) { g.pend_first = li; g.pend_set = L.cur_set; }
    ++g.pend_n;
    if (g.n_sets > 1) L.cur_set ^= 1;
    if (g.pend_n >= g.merge_batch || li == g.n_lanes - 1) return flush_merges();
    return 0;
}

// one iteration of one lane: replayed from its hipGraph or enqueued launch by launch
static int lane_iteration(Lane& L, int iter, int mode) {
    if (L.wait_resolved) {                   // lane 0 may still be summing this lane's planes / tone-mapping the image its merge adds to, for the previous pbo
        const hipError_t e = hipStreamWaitEvent(L.stream, g.resolved, 0);
        if (e != hipSuccess) return (int)e;
        L.wait_resolved = false;
    }
    if (g.n_sets > 1) {
        // the set this iteration renders into was zeroed by the merge that folded the lane's iteration of two calls ago
        if (L.set_busy[L.cur_set]) { const hipError_t e = hipStreamWaitEvent(L.stream, L.set_free[L.cur_set], 0); if (e != hipSuccess) return (int)e; L.set_busy[L.cur_set] = false; }
        L.plane = L.plane_sets[L.cur_set]; L.flags = L.flag_sets[L.cur_set]; L.ws.film_flags = L.flags;
    }
    L.last_iter = iter;
    g_prof.stream = L.stream;
    note_counts(L);
    int rc;
    if (g.use_graph && !g_prof.on && !L.ws.stats) rc = replay_iteration(L, iter, mode);
    else { L.expected_iter = -1; rc = enqueue_iteration(L, iter, mode, false); }
    if (rc || (rc = request_counts(L))) return rc;
    if (g.tiles) return 0;
    const double t0 = g_ht.on ? now_us() : 0.0;
    rc = note_unmerged((int)(&L - &g.lanes[0]));
    if (g_ht.on) g_ht.merge_us += now_us() - t0;
    return rc;
}

int render_iteration(unsigned char* pbo, int step) {
    (void)pbo;
    if (!g.live) return PTK_E_NOSCENE;
    if (step < 1 || (long long)(step - 1) * g.n_shards + g.shard + 1 > 2147483647LL) return PTK_E_BADARG;
    // iteration sharding (multi-GPU validation mode): this process renders iterations shard+1, shard+1+n, ... of the
    // whole frame; lanes are assigned by call number so that every lane is used whatever n_shards is
    const int iter = (step - 1) * g.n_shards + g.shard + 1;
    const int mode = current_mode();
    if (mode != g.hint_mode) {
        // the full-width hint was learnt from another sample mode's live counts (DIRECT dies after one bounce, BSDF
        // does not): start again from "every bounce at full width" and ignore read-backs that are still in flight
        reset_hints();
        g.hint_mode = mode;
    }
    if (g.tiles) {
        // TILES IN FLIGHT.  Lane k renders band k of this iteration.  Its compacted slots — the reference's RNG keys — continue those of the
        // bands above, so it starts the iteration when band k - 1 has FINISHED it (whose cumulative hit counts per bounce are then final,
        // Workspace::cum_*): band k of iteration i runs beside band k - 1 of iteration i + 1, band k - 2 of i + 2, ... — a software pipeline
        // over (band, iteration) that keeps K launches in flight like K iterations in flight do, while every pixel still receives its
        // iterations in order: the image is the reference's own sum, bit for bit.  The counts live in a ring of PTK_CUM_RING iterations:
        // band k - 1 may not start iteration i before band k has finished iteration i - PTK_CUM_RING (it would overwrite what band k reads).
        const int slot = (step - 1) & (PTK_CUM_RING - 1);
        for (int k = 0; k < g.n_lanes; ++k) {
            Lane& L = g.lanes[k];
            hipError_t e;
            if (k > 0 && (e = hipStreamWaitEvent(L.stream, g.lanes[k - 1].ring_ev[slot], 0)) != hipSuccess) return (int)e;                             // band above: this iteration done
            if (k + 1 < g.n_lanes && step > PTK_CUM_RING && (e = hipStreamWaitEvent(L.stream, g.lanes[k + 1].ring_ev[slot], 0)) != hipSuccess) return (int)e;   // band below: iteration i - RING done
            const int rc = lane_iteration(L, iter, mode);
            if (rc) return rc;
            if ((e = hipEventRecord(L.ring_ev[slot], L.stream)) != hipSuccess) return (int)e;
        }
        g.last_lane = g.n_lanes - 1;
        return 0;
    }
    const int li = (step - 1) % g.n_lanes;
    g.last_lane = li;
    return lane_iteration(g.lanes[li], iter, mode);
}

// all lanes' work so far -> visible to lane 0's stream (event edges only, no host wait)
int join_on_lane0() {
    if (g.merge_stream) {
        hipError_t e = hipEventRecord(g.merge_done, g.merge_stream);
        if (e != hipSuccess) return (int)e;
        if ((e = hipStreamWaitEvent(g.lanes[0].stream, g.merge_done, 0)) != hipSuccess) return (int)e;
    }
    for (size_t i = 1; i < g.lanes.size(); ++i) {
        hipError_t e = hipEventRecord(g.lanes[i].done, g.lanes[i].stream);
        if (e != hipSuccess) return (int)e;
        if ((e = hipStreamWaitEvent(g.lanes[0].stream, g.lanes[i].done, 0)) != hipSuccess) return (int)e;
    }
    return 0;
}

// the image as of the last call, on lane 0's stream after joining the other lanes (event edges, no host wait): the running sum itself (one
// plane, tiles in flight, ordered accumulation), or ((plane0 + plane1) + plane2) + ... of planes that accumulate
int resolve_image() {
    int rc = flush_merges();                 // (batched merges: fold what has been rendered so far)
    if (rc) return rc;
    rc = join_on_lane0();
    if (rc) return rc;
    if (g.image != g.planes && !g.ordered) rc = launch_sum_planes(g.planes, g.n_lanes * g.spp, 3 * (size_t)g.P, g.image, g.lanes[0].stream);
    return rc;
}
// asynchronous readers on lane 0 were enqueued (the plane sum above, the tone map of ptk_pathtrace): the other lanes' next launches write the
// planes / merge into the image they read — they wait for them (event edge, no host wait)
int image_reader_enqueued() {
    // (tiles in flight: one shared plane that bands 1..K-1 keep adding the NEXT iteration into — the tone map must have read it first)
    if (g.n_lanes == 1 || (g.image == g.planes && !g.tiles)) return 0;
    const hipError_t e = hipEventRecord(g.resolved, g.lanes[0].stream);
    if (e != hipSuccess) return (int)e;
    for (size_t i = 1; i < g.lanes.size(); ++i) g.lanes[i].wait_resolved = true;
    if (g.merge_stream) g.merge_wait_resolved = true;
    return 0;
}

// pipeline 11: did a block of k_trace_shade give its look-back wait up (ptk_kernels.hip: LookBack)?  Called where the host has just waited for the
// lanes anyway; the image is wrong then, and the caller gets PTK_E_DEVICE instead of it.
int lookback_status() {
    if (!ts_pipeline_wanted(g.pipeline)) return 0;
    for (Lane& l : g.lanes) {
        if (!l.ws.look_gave_up) continue;
        unsigned long long w = 0;
        const hipError_t e = hipMemcpy(&w, l.ws.look_gave_up, sizeof w, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return (int)e;
        if (w) return PTK_E_DEVICE;
    }
    return 0;
}

[[noreturn]] void die(const char* msg, int rc, const char* file, int line) {      // checkCUDAErrorFn, scene.cpp:20-38
    std::fprintf(stderr, "HIP error (%s:%d): %s: %s\n", file, line, msg, rc == PTK_E_DEVICE ? "a kernel gave a device-side wait up (pipeline 11)" : rc >= 1000 ? "ptk argument/scene error" : hipGetErrorString((hipError_t)rc));
    std::exit(EXIT_FAILURE);
}
#define PTK_CHECK(msg, expr) do { int _rc = (expr); if (_rc) die(msg, _rc, __FILE__, __LINE__); } while (0)

} // namespace

// ---------------------------------------------------------------- C ABI (include/ptk.h, render API)
extern "C" {

int ptk_pathtrace_init(PtkScene* s) { if (!s || !s->scene) return PTK_E_NOSCENE; return context_init(s->scene); }
int ptk_pathtrace_free(void) { return context_free(); }

int ptk_pathtrace(unsigned char* pbo, int frame, int iter) {
    (void)frame;
    const double t0 = g_ht.on ? now_us() : 0.0;
    int rc = render_iteration(pbo, iter);
    if (g_ht.on) { g_ht.call_us += now_us() - t0; ++g_ht.calls; }
    if (rc || !pbo) return rc;
    if ((rc = resolve_image())) return rc;
    const Camera& cam = g.scene->state.camera;
    // samples accumulated so far by THIS process (with iteration sharding a call is one sample, whatever its iteration number)
    if ((rc = ptk_tonemap(g.image, cam.resolution.x, cam.resolution.y, iter * g.spp, pbo, g.lanes[0].stream))) return rc;
    return image_reader_enqueued();
}

int ptk_set_sample_mode(int mode) { if (mode < 0 || mode > 2) return PTK_E_BADARG; g_mode_override = mode; return 0; }
int ptk_get_sample_mode(void) { return current_mode(); }

/* Tiles in flight (1..16, default 1 = off; set before ptk_pathtrace_init; not combined with iterations in flight, the tile partition or
 * samples per launch): every iteration is rendered as K bands of rows on K streams, band k of iteration i beside band k - 1 of iteration
 * i + 1 (render_iteration).  The adds go straight into the one image, in the reference's order — bit-identical to the default's without planes
 * or merges (iterations in flight reach the same image through ordered accumulation, and are faster: 22 against 14.5 Grays/s on cornell.txt). */
int ptk_set_tiles_in_flight(int tiles) {
    if (tiles < 1 || tiles > kMaxLanes || g.live) return PTK_E_BADARG;
    g.want_tiles = tiles; g.sched_explicit = true;
    return 0;
}
/* Accumulation with several radiance planes (iterations in flight, samples per launch; set before ptk_pathtrace_init): -1 / 1 (default) = ordered —
 * every launch's planes are folded into the image in the reference's order of adds (flush_merges: bit-identical to one iteration at a time);
 * 0 = the planes accumulate and are summed when the image is read, ((p0 + p1) + p2) + ... (no merge launches: 3-10 % faster, float association
 * differs). */
int ptk_set_ordered_accumulation(int on) {
    if (on < -1 || on > 1 || g.live) return PTK_E_BADARG;
    g.want_ordered = on;
    return 0;
}
int ptk_set_iterations_in_flight(int lanes) {
    if (lanes < 1 || lanes > kMaxLanes || g.live) return PTK_E_BADARG;     // set before ptk_pathtrace_init
    g.want_lanes = lanes; g.sched_explicit = true;
    return 0;
}

static void invalidate_graphs() { for (Lane& l : g.lanes) for (Lane::GraphSlot& q : l.gs) q.mode = -1; }     // captured launches bake these knobs in: recapture
// The reference's own threaded walk as the work-list walk (pipeline 5) and the traversal counters (whose numbers describe that walk, direction
// order included) need all six direction copies; a large mesh is uploaded with one (dev_scene.hip).  Re-upload on demand; captured graphs hold
// the old table's address.
static int need_six_copies() {
    if (!g.scene || !g.scene->tempDevScene.uploaded || g.scene->tempDevScene.n_copies == 6) return 0;
    sync_all();
    const int rc = g.scene->tempDevScene.upload_threaded(*g.scene, 6);
    invalidate_graphs();
    return rc;
}
/* Opt-in fixed cube lights (not reference behaviour, default off): cube lights get a sampled position and a pdf. */
int ptk_set_light_fix(int on) { set_light_fix(on != 0); invalidate_graphs(); return 0; }
/* Replay each iteration from a captured hipGraph (1) instead of launching its ~25 kernels one by one (0, default). */
int ptk_set_graph_replay(int on) { g.use_graph = on != 0; g.sched_explicit = true; return 0; }

/* Samples per launch (1..64, default 1; set before ptk_pathtrace_init): ptk_pathtrace(.., step) then traces iterations
 * (step-1)*spp + 1 .. step*spp of the owned pixels in ONE set of launches (camera jitter keyed per iteration as usual, shading
 * RNG keyed by the batch's first iteration and the compacted slot, one radiance plane per sample). */
int ptk_set_samples_per_launch(int spp) {
    if (spp < 1 || spp > 64 || g.live) return PTK_E_BADARG;
    g.spp = spp;
    return 0;
}

int ptk_get_image(float* host_out) {
    if (!g.live) return PTK_E_NOSCENE;
    if (!host_out) return PTK_E_BADARG;
    int rc = resolve_image();
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(host_out, g.image, sizeof(float) * 3 * (size_t)g.P, hipMemcpyDeviceToHost, g.lanes[0].stream);
    if (e != hipSuccess) return (int)e;
    if ((rc = rc_of(hipStreamSynchronize(g.lanes[0].stream)))) return rc;
    return lookback_status();
}
int ptk_get_image_device(float** out) {
    if (!g.live) return PTK_E_NOSCENE;
    if (!out) return PTK_E_BADARG;
    int rc = resolve_image();
    if (rc) return rc;
    if ((rc = rc_of(hipStreamSynchronize(g.lanes[0].stream)))) return rc;
    *out = g.image;
    return lookback_status();
}
int ptk_copy_image_to_device(float* dst) {
    if (!g.live) return PTK_E_NOSCENE;
    if (!dst) return PTK_E_BADARG;
    int rc = resolve_image();
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(dst, g.image, sizeof(float) * 3 * (size_t)g.P, hipMemcpyDeviceToDevice, g.lanes[0].stream);
    if (e != hipSuccess) return (int)e;
    return rc_of(hipStreamSynchronize(g.lanes[0].stream));
}
int ptk_get_live_counts(int* counts, int* traced_depth) {          // of the LAST ptk_pathtrace call
    if (!g.live) return PTK_E_NOSCENE;
    if (!counts) return PTK_E_BADARG;
    if (g.tiles) {                                                  // the bands of the last iteration together
        std::vector<int> part((size_t)g.traceDepth + 1);
        for (int d = 0; d <= g.traceDepth; ++d) counts[d] = 0;
        for (Lane& l : g.lanes) {
            hipError_t e = hipMemcpyAsync(part.data(), l.ws.counts, sizeof(int) * part.size(), hipMemcpyDeviceToHost, l.stream);
            if (e != hipSuccess) return (int)e;
            if ((e = hipStreamSynchronize(l.stream)) != hipSuccess) return (int)e;
            for (int d = 0; d <= g.traceDepth; ++d) counts[d] += part[d];
        }
        if (traced_depth) { int d = 0; while (d < g.traceDepth && counts[d] > 0) ++d; *traced_depth = d; }
        return 0;
    }
    Lane& L = g.lanes[g.last_lane];
    hipError_t e = hipMemcpyAsync(counts, L.ws.counts, sizeof(int) * ((size_t)g.traceDepth + 1), hipMemcpyDeviceToHost, L.stream);
    if (e != hipSuccess) return (int)e;
    if ((e = hipStreamSynchronize(L.stream)) != hipSuccess) return (int)e;
    if (traced_depth) { int d = 0; while (d < g.traceDepth && counts[d] > 0) ++d; *traced_depth = d; }
    return 0;
}
int ptk_get_ray_total(unsigned long long* rays) {
    if (!g.live) return PTK_E_NOSCENE;
    if (!rays) return PTK_E_BADARG;
    sync_all();
    return rc_of(hipMemcpy(rays, g.ray_total, sizeof(unsigned long long), hipMemcpyDeviceToHost));
}
int ptk_set_tile_partition(int n_parts, int part, int tile_rows) {
    if (n_parts < 1 || part < 0 || part >= n_parts || tile_rows < 1) return PTK_E_BADARG;
    if (g.live) return PTK_E_BADARG;                 // must be set before ptk_pathtrace_init
    g.n_parts = n_parts; g.part = part; g.tile_rows = tile_rows;
    return 0;
}
/* Host-only helpers describing the tile partition (no GPU needed): number of paths rank `part` owns, and the pixel index
 * (x + y*W) that local path j maps to — the same mapping k_generate uses. */
int ptk_partition_count(int W, int H, int n_parts, int part, int tile_rows) {
    if (W < 1 || H < 1 || n_parts < 1 || part < 0 || part >= n_parts || tile_rows < 1) return -1;
    return local_path_count(W, H, n_parts, part, tile_rows);
}
int ptk_partition_pixel(int W, int n_parts, int part, int tile_rows, int j) {
    if (W < 1 || n_parts < 1 || part < 0 || part >= n_parts || tile_rows < 1 || j < 0) return -1;
    const int lr = j / W, x = j - lr * W;
    const int y = ((lr / tile_rows) * n_parts + part) * tile_rows + (lr % tile_rows);
    return x + y * W;
}
/* Iteration sharding (SURVEY 8e validation mode; set before ptk_pathtrace_init): ptk_pathtrace(.., step) renders iteration
 * (step-1)*n_shards + shard + 1 of the WHOLE frame with the reference's own RNG keys (camera key (x, y, iter), pathtrace.cu:148;
 * shading key (iter, compacted slot, depth), :458,532).  Shard g of n then holds the running sum of exactly the iterations g+1, g+1+n,
 * ...; the sum of the shards is the single-process image up to float association. */
int ptk_set_iteration_sharding(int n_shards, int shard) {
    if (n_shards < 1 || shard < 0 || shard >= n_shards || g.live) return PTK_E_BADARG;
    g.n_shards = n_shards; g.shard = shard;
    return 0;
}
/* 1 while the camera launch stores the path state of the hits only (open scene: fewer than ~half of the camera rays of earlier
 * iterations hit something; note_counts).  Diagnostic. */
int ptk_get_open_scene_hint(int* out) {
    if (!g.live) return PTK_E_NOSCENE;
    if (!out) return PTK_E_BADARG;
    *out = g.lanes[0].gen_late;
    return 0;
}
/* Bounces the render loop currently enqueues at full width (the rest of an iteration runs in the single-workgroup k_tail);
 * trace depth until live counts of earlier iterations have been seen.  Diagnostic. */
int ptk_get_full_width_bounces(int* out) {
    if (!g.live) return PTK_E_NOSCENE;
    if (!out) return PTK_E_BADARG;
    int fb = 0;
    for (const Lane& l : g.lanes) fb = std::max(fb, l.full_bounces);
    *out = sparse_tail() ? std::min(std::max(fb, 1), g.traceDepth) : g.traceDepth;
    return 0;
}
#ifdef PTK_TEST_HOOKS
/* TEST-ONLY (include/ptk_test_hooks.h; compiled into libptk_hooks.so only).  ptk_debug_slot_shift(k0, k1): the shading kernels key their RNG by
 * compacted slot + k0 at depth 0 and + k1 at depth >= 1 (the reference's GPU build flipped a few silhouette hit / miss decisions per iteration,
 * which shifts every later slot of that depth by a small constant; tests/golden/make_gallery_fixture.py).  ptk_debug_clear_image(): zero the
 * radiance planes, so that single iterations can be read back one by one. */
int ptk_debug_slot_shift(int k0, int k1) { g_debug_slot_shift = k0; g_debug_slot_shift_deep = k1; invalidate_graphs(); return 0; }
int ptk_debug_clear_image(void) {
    if (!g.live) return PTK_E_NOSCENE;
    sync_all();
    (void)flush_merges(); sync_all();
    int rc = rc_of(hipMemset(g.planes, 0, sizeof(float) * 3 * (size_t)g.P * (g.tiles ? 1 : (size_t)g.n_lanes * g.spp * g.n_depth * g.n_sets)));
    if (!rc && g.flags) rc = rc_of(hipMemset(g.flags, 0, (size_t)g.n_lanes * g.spp * g.n_depth * g.n_sets * (((size_t)g.P + 63) / 64)));
    if (!rc && g.image != g.planes) rc = rc_of(hipMemset(g.image, 0, sizeof(float) * 3 * (size_t)g.P));
    return rc ? rc : rc_of(hipDeviceSynchronize());      // null-stream clear vs the lanes' non-blocking streams
}
#endif
int ptk_set_pipeline(int which) {
    if (which < 0 || which > 11 || which == 2 || which == 3 || which == 4 || which == 9) return PTK_E_BADARG;      // retired A/B layouts: 3 (ticketed fusion, round 3); 2 (wave-uniform primitive loops), 4 (scatter-pass compaction), 9 (no k_tail) in round 4
    if (g.live && g.tiles && !trace_pipeline(which)) return PTK_E_BADARG;
    if (g.live && which == 5) { const int rc = need_six_copies(); if (rc) return rc; }
    g.pipeline = which;
    for (Lane& l : g.lanes) { l.ws.inline_bvh = inline_bvh_choice(which); l.ws.walk = pick_walk(which, g.n_lanes); l.ws.wide_lpr = pick_lpr(g.n_lanes); l.ws.seg_done = fold_segments(which, g.n_lanes) ? l.seg_done_mem : nullptr; }
    invalidate_graphs();
    // the hints were learnt from the other pipeline's counters (hcounts[] is only written by the hit-dense launches): start again, like a
    // sample-mode switch does
    reset_hints();
    return 0;
}
/* Per-kernel timing: begin -> every launch of the following ptk_pathtrace calls is bracketed by HIP events on the
 * render stream; end -> synchronises and returns, per kernel class (0 generate, 1 intersect, 2 shade, 3 scatter),
 * the summed device time in ms and the number of launches.  Not for use inside a timed region. */
int ptk_profile_begin(void) {
    if (!g.live) return PTK_E_NOSCENE;
    for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof.ev.clear(); g_prof.cls.clear(); g_prof.on = true;
    return 0;
}
int ptk_profile_end(float* ms4, int* launches4) {
    if (!g.live) return PTK_E_NOSCENE;
    if (!ms4 || !launches4) return PTK_E_BADARG;
    g_prof.on = false;
    sync_all();
    for (int k = 0; k < 4; ++k) { ms4[k] = 0.f; launches4[k] = 0; }
    for (size_t i = 0; i < g_prof.cls.size() && 2 * i + 1 < g_prof.ev.size(); ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) == hipSuccess) { ms4[g_prof.cls[i]] += ms; launches4[g_prof.cls[i]] += 1; }
    }
    for (hipEvent_t ev : g_prof.ev) (void)hipEventDestroy(ev);
    g_prof.ev.clear(); g_prof.cls.clear();
    return 0;
}
/* Enable (1) / disable (0) the BVH traversal counters; read them with ptk_get_traversal_stats (node visits, triangle tests). */
int ptk_enable_traversal_stats(int on) {
    if (!g.live) return PTK_E_NOSCENE;
    if (on && g.tiles) return PTK_E_BADARG;              // (the counters live in the staged kernels, which tiles in flight cannot use)
    if (on) { const int rc = need_six_copies(); if (rc) return rc; }
    sync_all();
    for (Lane& l : g.lanes) {
        if (on && !l.ws.stats) {
            int rc = dalloc(&l.ws.stats, 2); if (rc) return rc;
            hipError_t e = hipMemset(l.ws.stats, 0, 2 * sizeof(unsigned long long));
            if (e != hipSuccess) return (int)e;
        }
        if (!on && l.ws.stats) dfree(l.ws.stats);
    }
    return 0;
}
int ptk_get_traversal_stats(unsigned long long* out2) {
    if (!g.live) return PTK_E_NOSCENE;
    if (!out2) return PTK_E_BADARG;
    sync_all();
    out2[0] = out2[1] = 0;
    for (Lane& l : g.lanes) {
        if (!l.ws.stats) return PTK_E_NOSCENE;
        unsigned long long v[2];
        hipError_t e = hipMemcpy(v, l.ws.stats, sizeof v, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return (int)e;
        out2[0] += v[0]; out2[1] += v[1];
    }
    return 0;
}
int ptk_stream_sync(void) { if (!g.live) return PTK_E_NOSCENE; sync_all(); return lookback_status(); }
int ptk_timer_start(void) {
    if (!g.live) return PTK_E_NOSCENE;
    sync_all();
    return rc_of(hipEventRecord(g.ev0, g.lanes[0].stream));
}
int ptk_timer_stop(float* ms) {
    if (!g.live) return PTK_E_NOSCENE;
    if (!ms) return PTK_E_BADARG;
    int rc = join_on_lane0();
    if (rc) return rc;
    hipError_t e = hipEventRecord(g.ev1, g.lanes[0].stream);
    if (e != hipSuccess) return (int)e;
    if ((e = hipEventSynchronize(g.ev1)) != hipSuccess) return (int)e;
    return rc_of(hipEventElapsedTime(ms, g.ev0, g.ev1));
}

} // extern "C"

// ---------------------------------------------------------------- C++ API (pathtrace.h)
void InitDataContainer(GuiDataContainer* guiData) { g.gui = guiData; }

// Schedule of the C++ API, chosen at pathtraceInit (the four reference signatures have no room for knobs).  DEFAULT: kDefaultLanes (four) iterations in flight +
// hipGraph replay with ordered accumulation — scene->state.image is bit-identical to one iteration at a time, i.e. to the reference's running sum,
// in every sample mode (flush_merges; MIS gets one plane per bounce, which is decided by the mode at pathtraceInit: the viewer re-initialises on a
// sample-mode change, main.cpp:173-201, preview.cpp:247).  Overrides: PTK_IN_FLIGHT=<L> (1 = the reference's own schedule), PTK_TILES=<K>,
// PTK_GRAPH=0|1, PTK_ORDERED=-1|0|1 (as the ptk_set_* calls, which a host may also make itself before pathtraceInit), PTK_EAGER_IMAGE=1 = refresh
// scene->state.image after EVERY call like pathtrace.cu:809-810 (a 7.7 MB D2H and a host wait per 0.3 ms iteration at 800 x 800).
void pathtraceInit(Scene* scene) {
    if (!g.live) {
        const bool env_sched = getenv("PTK_IN_FLIGHT") || getenv("PTK_TILES") || getenv("PTK_GRAPH");
        if (!g.sched_explicit && !env_sched) {
            g.want_lanes = kDefaultLanes; g.use_graph = true; g.want_tiles = 1;
        }
        if (const char* e = getenv("PTK_IN_FLIGHT")) { const int l = atoi(e); if (l >= 1 && l <= kMaxLanes) g.want_lanes = l; }
        if (const char* e = getenv("PTK_GRAPH")) g.use_graph = atoi(e) != 0;
        if (const char* e = getenv("PTK_TILES")) { const int t = atoi(e); if (t >= 1 && t <= kMaxLanes) g.want_tiles = t; }
        if (const char* e = getenv("PTK_ORDERED")) { const int o = atoi(e); if (o >= -1 && o <= 1) g.want_ordered = o; }
    }
    PTK_CHECK("pathtraceInit", context_init(scene));
    g.image_policy = (getenv("PTK_EAGER_IMAGE") && atoi(getenv("PTK_EAGER_IMAGE")) != 0) ? 1 : ((getenv("PTK_LAZY_IMAGE") && atoi(getenv("PTK_LAZY_IMAGE")) != 0) ? 2 : 0);
    g.host_image_stale = false;
    // the host image lives where the reference keeps it; pinned in place, so that pathtraceResolve's copy is one asynchronous DMA.  The
    // reference never resizes it after pathtraceInit (main.cpp:146-170 only reads it); if pinning fails the copy is a plain pageable one.
    std::vector<ptk::vec3>& img = scene->state.image;
    img.assign((size_t)g.P, ptk::vec3());
    if (hipHostRegister(img.data(), sizeof(ptk::vec3) * img.size(), hipHostRegisterDefault) == hipSuccess) g.host_image_reg = img.data();
    else (void)hipGetLastError();
}

// (the reference's state.image is current at any time, so a caller may read it after pathtraceFree: make it so before the device image goes)
void pathtraceFree() {
    if (g.live && g.host_image_stale && g.image_policy != 2) pathtraceResolve();
    PTK_CHECK("pathtraceFree", context_free());
}

// Not a reference function.  scene->state.image (the running radiance SUM, pathtrace.cu:809-810) and guiData->TracedDepth
// (pathtrace.cu:793-796) as of the last pathtrace() call; waits for the GPU.  SURVEY 8(b) "Outputs": the per-iteration D2H may be lazy as
// long as state.image is current when saveImage() runs — the maintainer calls this at the top of saveImage().
void pathtraceResolve() {
    if (!g.live) return;
    std::vector<ptk::vec3>& img = g.scene->state.image;
    if ((int)img.size() != g.P || (g.host_image_reg && g.host_image_reg != (void*)img.data())) {      // the caller replaced the vector: pin the new one
        if (g.host_image_reg) { (void)hipHostUnregister(g.host_image_reg); g.host_image_reg = nullptr; }
        if ((int)img.size() != g.P) img.assign((size_t)g.P, ptk::vec3());
        if (hipHostRegister(img.data(), sizeof(ptk::vec3) * img.size(), hipHostRegisterDefault) == hipSuccess) g.host_image_reg = img.data();
        else (void)hipGetLastError();
    }
    PTK_CHECK("pathtraceResolve", ptk_get_image(reinterpret_cast<float*>(img.data())));
    g.host_image_stale = false;
    if (g.gui) {
        std::vector<int> counts((size_t)g.traceDepth + 1);
        int depth = 0;
        PTK_CHECK("pathtraceResolve", ptk_get_live_counts(counts.data(), &depth));
        g.gui->TracedDepth = depth;
    }
}

// The reference refreshes scene->state.image after EVERY call (pathtrace.cu:809-810) and its caller reads it with no hook (saveImage, main.cpp:146-170).
// Default policy (round 5) — correct for an UNEDITED caller, at the overlapped speed where that matters:
//   * pbo != nullptr (the viewer's loop, main.cpp:205-219: one call per displayed frame, and the S key may save at any time): resolve after every
//     call, as the reference does;
//   * pbo == nullptr (headless callers): enqueue only — no host wait, no D2H — and resolve after the call with iteration >= state.iterations, the one
//     after which main.cpp:221-222 saves and exits, and in pathtraceFree().  TracedDepth follows from the asynchronous counts read-back (note_counts),
//     a call or more late, and is exact after a resolve.
// pathtraceResolve() remains for a headless caller that reads the image earlier; PTK_EAGER_IMAGE=1 / PTK_LAZY_IMAGE=1 force either behaviour.
void pathtrace(uchar4* pbo, int frame, int iteration) {
    PTK_CHECK("pathtrace", ptk_pathtrace(reinterpret_cast<unsigned char*>(pbo), frame, iteration));
    g.host_image_stale = true;
    const bool last_call = g.scene && iteration >= g.scene->state.iterations;
    if (g.image_policy == 1 || (g.image_policy == 0 && (pbo != nullptr || last_call))) pathtraceResolve();
}
