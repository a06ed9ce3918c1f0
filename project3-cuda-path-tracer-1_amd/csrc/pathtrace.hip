// pathtrace.hip — render context and per-iteration driver: the MI355X-native counterpart of
// pathtraceInit / pathtraceFree / pathtrace (reference src/pathtrace.cu:85-125, 637-816).
//
// One iteration = generate + traceDepth bounces, enqueued back-to-back on one HIP stream with the live-path
// counts resident on the device (the reference does 4 blocking 4-byte D2H copies, a cudaDeviceSynchronize,
// a full-frame cudaMemset and a cudaMalloc/cudaFree pair per bounce/iteration).  Nothing here synchronises
// with the host; image / counts are fetched lazily by the accessors.
#include "ptk_internal.h"
#include "pathtrace.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace ptk;

__attribute__((weak)) SampleMode sampleMode = MIS;       // main.cpp:20 (the application's definition wins)

namespace {

struct RenderContext {
    Scene* scene = nullptr;
    GuiDataContainer* gui = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float* image = nullptr;             // float[3*P] running radiance sum (full frame even when tiled)
    PtkPaths bufA = {nullptr, nullptr, nullptr}, bufB = {nullptr, nullptr, nullptr};
    PtkHits hits = {nullptr, nullptr, nullptr};
    Workspace ws;
    int P = 0, capacity = 0, traceDepth = 0;
    int n_parts = 1, part = 0, tile_rows = 1;
    int pipeline = 0;
    bool live = false;
} g;

int g_mode_override = -1;               // ptk_set_sample_mode; -1 = follow the `sampleMode` global

template <typename T> int dalloc(T** p, size_t count) {
    *p = nullptr;
    hipError_t e = hipMalloc((void**)p, sizeof(T) * std::max<size_t>(count, 1));
    return e == hipSuccess ? 0 : (int)e;
}
template <typename T> void dfree(T*& p) { if (p) (void)hipFree(p); p = nullptr; }

int alloc_paths(PtkPaths& b, int cap) {
    int rc;
    if ((rc = dalloc(&b.o_pix, 4 * (size_t)cap))) return rc;
    if ((rc = dalloc(&b.d_bnc, 4 * (size_t)cap))) return rc;
    return dalloc(&b.col_pdf, 4 * (size_t)cap);
}
void free_paths(PtkPaths& b) { dfree(b.o_pix); dfree(b.d_bnc); dfree(b.col_pdf); }

int context_free() {
    if (g.stream) (void)hipStreamSynchronize(g.stream);
    dfree(g.image);
    free_paths(g.bufA); free_paths(g.bufB);
    dfree(g.hits.t_n); dfree(g.hits.p_mat); dfree(g.hits.uv_ids);
    dfree(g.ws.agg_a); dfree(g.ws.agg_b); dfree(g.ws.valid_a); dfree(g.ws.valid_b);
    dfree(g.ws.counts); dfree(g.ws.ray_total); dfree(g.ws.stats); dfree(g.ws.queue); dfree(g.ws.list);
    if (g.ev0) { (void)hipEventDestroy(g.ev0); g.ev0 = nullptr; }
    if (g.ev1) { (void)hipEventDestroy(g.ev1); g.ev1 = nullptr; }
    if (g.stream) { (void)hipStreamDestroy(g.stream); g.stream = nullptr; }
    g.live = false; g.scene = nullptr; g.P = g.capacity = 0;
    return 0;
}

int context_init(Scene* scene) {
    if (!scene || !scene->ok) return PTK_E_NOSCENE;
    if (!scene->tempDevScene.uploaded) { scene->setDevData(); if (scene->devError) return scene->devError; }
    context_free();
    const Camera& cam = scene->state.camera;
    g.scene = scene;
    g.P = cam.resolution.x * cam.resolution.y;
    g.traceDepth = scene->state.traceDepth;
    if (g.traceDepth < 1) return PTK_E_BADARG;
    g.capacity = local_path_count(cam.resolution.x, cam.resolution.y, g.n_parts, g.part, g.tile_rows);
    int rc;
    hipError_t e = hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking);
    if (e != hipSuccess) return (int)e;
    if ((e = hipEventCreate(&g.ev0)) != hipSuccess) return (int)e;
    if ((e = hipEventCreate(&g.ev1)) != hipSuccess) return (int)e;
    if ((rc = dalloc(&g.image, 3 * (size_t)g.P))) return rc;
    if ((e = hipMemsetAsync(g.image, 0, sizeof(float) * 3 * (size_t)g.P, g.stream)) != hipSuccess) return (int)e;
    if ((rc = alloc_paths(g.bufA, g.capacity))) return rc;
    if ((rc = alloc_paths(g.bufB, g.capacity))) return rc;
    if ((rc = dalloc(&g.hits.t_n, 4 * (size_t)g.capacity))) return rc;
    if ((rc = dalloc(&g.hits.p_mat, 4 * (size_t)g.capacity))) return rc;
    if ((rc = dalloc(&g.hits.uv_ids, 4 * (size_t)g.capacity))) return rc;
    const size_t blocks = ((size_t)g.capacity + 255) / 256;
    if ((rc = dalloc(&g.ws.agg_a, blocks))) return rc;
    if ((rc = dalloc(&g.ws.agg_b, blocks))) return rc;
    if ((rc = dalloc(&g.ws.valid_a, (size_t)g.capacity))) return rc;
    if ((rc = dalloc(&g.ws.valid_b, (size_t)g.capacity))) return rc;
    if ((rc = dalloc(&g.ws.counts, (size_t)g.traceDepth + 2))) return rc;
    if ((rc = dalloc(&g.ws.ray_total, 1))) return rc;
    if ((rc = dalloc(&g.ws.queue, 1))) return rc;
    if ((rc = dalloc(&g.ws.list, (size_t)g.capacity))) return rc;
    if ((e = hipMemsetAsync(g.ws.queue, 0, sizeof(unsigned), g.stream)) != hipSuccess) return (int)e;
    g.ws.inline_bvh = (g.pipeline == 0);
    if ((e = hipMemsetAsync(g.ws.counts, 0, sizeof(int) * ((size_t)g.traceDepth + 2), g.stream)) != hipSuccess) return (int)e;
    if ((e = hipMemsetAsync(g.ws.ray_total, 0, sizeof(unsigned long long), g.stream)) != hipSuccess) return (int)e;
    g.ws.capacity = g.capacity;
    g.live = true;
    return 0;
}

int current_mode() { return g_mode_override >= 0 ? g_mode_override : (int)sampleMode; }

// optional per-kernel timing (bench.py's roofline leg): HIP events on the render stream around every launch
struct KernelProfile {
    bool on = false;
    std::vector<hipEvent_t> ev;          // pairs
    std::vector<int> cls;                // PTK_KCLASS_* per pair
} g_prof;

void prof_mark(int cls, bool begin) {
    if (!g_prof.on) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, g.stream);
    g_prof.ev.push_back(e);
    if (begin) g_prof.cls.push_back(cls);
}

int render_iteration(unsigned char* pbo, int iter) {
    if (!g.live) return PTK_E_NOSCENE;
    if (iter < 1) return PTK_E_BADARG;
    const Camera& cam = g.scene->state.camera;
    const SceneView S = dev_view(*g.scene);
    prof_mark(0, true);
    int rc = launch_generate(cam, iter, g.traceDepth, g.bufA, g.n_parts, g.part, g.tile_rows, g.capacity, g.ws.counts, g.stream);
    prof_mark(0, false);
    if (rc) return rc;
    PtkPaths cur = g.bufA, nxt = g.bufB;
    const int mode = current_mode();
    for (int depth = 0; depth < g.traceDepth; ++depth) {
        if (g_prof.on) {
            rc = launch_bounce_staged_profiled(S, mode, iter, depth, g.capacity, cur, nxt, g.hits, g.image, g.ws, g.stream, prof_mark);
        } else {
            rc = launch_bounce_staged(S, mode, iter, depth, g.capacity, cur, nxt, g.hits, g.image, g.ws, g.stream);
        }
        if (rc) return rc;
        PtkPaths t = cur; cur = nxt; nxt = t;
    }
    if (pbo) rc = ptk_tonemap(g.image, cam.resolution.x, cam.resolution.y, iter, pbo, g.stream);
    return rc;
}

[[noreturn]] void die(const char* msg, int rc, const char* file, int line) {      // checkCUDAErrorFn, scene.cpp:20-38
    std::fprintf(stderr, "HIP error (%s:%d): %s: %s\n", file, line, msg, rc >= 1000 ? "ptk argument/scene error" : hipGetErrorString((hipError_t)rc));
    std::exit(EXIT_FAILURE);
}
#define PTK_CHECK(msg, expr) do { int _rc = (expr); if (_rc) die(msg, _rc, __FILE__, __LINE__); } while (0)

} // namespace

// ---------------------------------------------------------------- C ABI (include/ptk.h, render API)
extern "C" {

int ptk_pathtrace_init(PtkScene* s) { if (!s || !s->scene) return PTK_E_NOSCENE; return context_init(s->scene); }
int ptk_pathtrace_free(void) { return context_free(); }
int ptk_pathtrace(unsigned char* pbo, int frame, int iter) { (void)frame; return render_iteration(pbo, iter); }
int ptk_set_sample_mode(int mode) { if (mode < 0 || mode > 2) return PTK_E_BADARG; g_mode_override = mode; return 0; }
int ptk_get_sample_mode(void) { return current_mode(); }

int ptk_get_image(float* host_out) {
    if (!g.live) return PTK_E_NOSCENE;
    if (!host_out) return PTK_E_BADARG;
    hipError_t e = hipMemcpyAsync(host_out, g.image, sizeof(float) * 3 * (size_t)g.P, hipMemcpyDeviceToHost, g.stream);
    if (e != hipSuccess) return (int)e;
    e = hipStreamSynchronize(g.stream);
    return e == hipSuccess ? 0 : (int)e;
}
int ptk_get_image_device(float** out) { if (!g.live) return PTK_E_NOSCENE; if (!out) return PTK_E_BADARG; *out = g.image; return 0; }
int ptk_copy_image_to_device(float* dst) {
    if (!g.live) return PTK_E_NOSCENE;
    if (!dst) return PTK_E_BADARG;
    hipError_t e = hipMemcpyAsync(dst, g.image, sizeof(float) * 3 * (size_t)g.P, hipMemcpyDeviceToDevice, g.stream);
    if (e != hipSuccess) return (int)e;
    e = hipStreamSynchronize(g.stream);
    return e == hipSuccess ? 0 : (int)e;
}
int ptk_get_live_counts(int* counts, int* traced_depth) {
    if (!g.live) return PTK_E_NOSCENE;
    if (!counts) return PTK_E_BADARG;
    hipError_t e = hipMemcpyAsync(counts, g.ws.counts, sizeof(int) * ((size_t)g.traceDepth + 1), hipMemcpyDeviceToHost, g.stream);
    if (e != hipSuccess) return (int)e;
    if ((e = hipStreamSynchronize(g.stream)) != hipSuccess) return (int)e;
    if (traced_depth) { int d = 0; while (d < g.traceDepth && counts[d] > 0) ++d; *traced_depth = d; }
    return 0;
}
int ptk_get_ray_total(unsigned long long* rays) {
    if (!g.live) return PTK_E_NOSCENE;
    if (!rays) return PTK_E_BADARG;
    hipError_t e = hipMemcpyAsync(rays, g.ws.ray_total, sizeof(unsigned long long), hipMemcpyDeviceToHost, g.stream);
    if (e != hipSuccess) return (int)e;
    e = hipStreamSynchronize(g.stream);
    return e == hipSuccess ? 0 : (int)e;
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
int ptk_set_pipeline(int which) {
    if (which != 0 && which != 1) return PTK_E_BADARG;
    g.pipeline = which; g.ws.inline_bvh = (which == 0);
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
    hipError_t e = hipStreamSynchronize(g.stream);
    if (e != hipSuccess) return (int)e;
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
    if (on && !g.ws.stats) {
        int rc = dalloc(&g.ws.stats, 2); if (rc) return rc;
        hipError_t e = hipMemsetAsync(g.ws.stats, 0, 2 * sizeof(unsigned long long), g.stream);
        return e == hipSuccess ? 0 : (int)e;
    }
    if (!on && g.ws.stats) { (void)hipStreamSynchronize(g.stream); dfree(g.ws.stats); }
    return 0;
}
int ptk_get_traversal_stats(unsigned long long* out2) {
    if (!g.live || !g.ws.stats) return PTK_E_NOSCENE;
    if (!out2) return PTK_E_BADARG;
    hipError_t e = hipMemcpyAsync(out2, g.ws.stats, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, g.stream);
    if (e != hipSuccess) return (int)e;
    e = hipStreamSynchronize(g.stream);
    return e == hipSuccess ? 0 : (int)e;
}
int ptk_stream_sync(void) { if (!g.live) return PTK_E_NOSCENE; hipError_t e = hipStreamSynchronize(g.stream); return e == hipSuccess ? 0 : (int)e; }
int ptk_timer_start(void) { if (!g.live) return PTK_E_NOSCENE; hipError_t e = hipEventRecord(g.ev0, g.stream); return e == hipSuccess ? 0 : (int)e; }
int ptk_timer_stop(float* ms) {
    if (!g.live) return PTK_E_NOSCENE;
    if (!ms) return PTK_E_BADARG;
    hipError_t e = hipEventRecord(g.ev1, g.stream);
    if (e != hipSuccess) return (int)e;
    if ((e = hipEventSynchronize(g.ev1)) != hipSuccess) return (int)e;
    e = hipEventElapsedTime(ms, g.ev0, g.ev1);
    return e == hipSuccess ? 0 : (int)e;
}

} // extern "C"

// ---------------------------------------------------------------- C++ API (pathtrace.h)
void InitDataContainer(GuiDataContainer* guiData) { g.gui = guiData; }

void pathtraceInit(Scene* scene) { PTK_CHECK("pathtraceInit", context_init(scene)); }

void pathtraceFree() { PTK_CHECK("pathtraceFree", context_free()); }

void pathtrace(uchar4* pbo, int frame, int iteration) {
    (void)frame;
    PTK_CHECK("pathtrace", render_iteration(reinterpret_cast<unsigned char*>(pbo), iteration));
    // pathtrace.cu:809-810: the host copy of the running sum is refreshed after every call
    std::vector<ptk::vec3>& img = g.scene->state.image;
    if ((int)img.size() != g.P) img.assign((size_t)g.P, ptk::vec3());
    PTK_CHECK("pathtrace", ptk_get_image(reinterpret_cast<float*>(img.data())));
    if (g.gui) {                                       // pathtrace.cu:793-796
        std::vector<int> counts((size_t)g.traceDepth + 1);
        int depth = 0;
        PTK_CHECK("pathtrace", ptk_get_live_counts(counts.data(), &depth));
        g.gui->TracedDepth = depth;
    }
}
