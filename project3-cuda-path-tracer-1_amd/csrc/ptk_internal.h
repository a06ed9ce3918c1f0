// ptk_internal.h — glue shared by the .hip translation units (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/ptk.h"
#include "scene.h"
#include "ptk_device.hpp"

struct PtkScene {
    Scene* scene = nullptr;
    bool owns = true;
};

namespace ptk {

// BVH walk of the work-list rays (both give bit-identical hits): the reference's threaded walk, one ray per lane (fewest
// instructions, longest dependent chain); four lanes per ray over the 8-wide collapsed tree (default).
enum { PTK_WALK_AUTO = 0, PTK_WALK_THREADED = 1, PTK_WALK_WIDE = 2 };
#define PTK_FILM_STORE 0x80000000u   // bit 31 of Workspace::plane_stride: radiance adds are plain stores (ptk_kernels.hip: pixel_ptr / film_add)
enum { PTK_CUM_RING = 4 };       // iterations a band may run ahead of the band below it (tiles in flight), power of two

void set_light_fix(bool on);            // opt-in fixed cube lights (ptk_set_light_fix)
bool light_fix_enabled();

// device view of an uploaded scene (pointers into DevScene's buffers), passed to kernels by value
inline SceneView dev_view(const ::Scene& s) {
    SceneView d;
    const ::DevScene& t = s.tempDevScene;
    d.geoms = (const DevGeom*)t.dev_geoms;       d.n_geoms = (int)s.geoms.size();
    d.gbox = (const DevGeomBox*)t.dev_gbox;      d.gmask = (const DevGeomMask*)t.dev_gmask;
    d.lds_geoms = (d.n_geoms > 0 && d.n_geoms <= PTK_LDS_GEOMS_MAX) ? d.n_geoms : 0;
    d.mats = (const DevMaterial*)t.dev_materials; d.n_mats = (int)s.materials.size();
    d.nodes = (const DevNode*)t.dev_nodes;       d.bvh_size = t.bvh_size;
    d.n_copies = t.n_copies;                     d.ranks = (const uint32_t*)t.dev_ranks;
    d.trie = (const DevTriE*)t.dev_trie;
    d.wnodes = (const uint4*)t.dev_wnodes;       d.n_wnodes = t.wide_nodes;  d.wide_width = t.wide_width;
    d.wleaves = (const uint4*)t.dev_wleaves;
    d.triv = (const DevTriV*)t.dev_triv;         d.tria = (const DevTriA*)t.dev_tria; d.n_tris = t.tri_num;
    d.lights = (const DevLight*)t.dev_lights;    d.n_lights = (int)s.lights.size();
    d.tex = (const DevTex*)t.dev_textures;       d.texels = (const float*)t.dev_texture_data; d.env_tex = t.envMapID;
    d.n_tex = t.tex_num; d.slim_hits = 0; d.light_fix = light_fix_enabled() ? 1 : 0;
    return d;
}

// The radiance planes a launch adds into (ptk_kernels.hip: film_add).  Layout float[sample][depth][3 P]: one plane per sample of the launch
// (plane_stride = floats per sample, 0 with one sample: pixelIndex then carries no sample bits; bit 31 = PTK_FILM_STORE) and — ordered accumulation
// in MIS mode (pathtrace.hip: merge_lane) — one per bounce (depth_stride = 3 P, else 0).  flags (null = off): one byte per plane and 64 pixels,
// set by every add (flag_sample_stride / flag_depth_stride = bytes per sample / per bounce), so that the merge visits touched tiles only.
struct FilmArgs { float* img; unsigned char* flags; unsigned plane_stride, depth_stride, flag_sample_stride, flag_depth_stride; };

// pipeline-internal launchers (ptk_kernels.hip); all asynchronous on `st`
struct Workspace {               // per-render-context scratch, device memory
    int* agg_a = nullptr;        // per-block hit counts of the intersect stage
    int* agg_b = nullptr;        // per-block survivor counts of the shade stage
    int* valid_a = nullptr;      // rayValid after intersect   [capacity]
    int* valid_b = nullptr;      // rayValid after shading     [capacity]
    int* counts = nullptr;       // live paths entering bounce d, d = 0..traceDepth  (device)
    unsigned long long* ray_total = nullptr;   // sum of counts over all iterations (device)
    unsigned long long* stats = nullptr;       // optional node/triangle counters
    unsigned* queue = nullptr;   // length of the traversal work list (rays that entered the BVH root box)
    int* list = nullptr;         // the work list [capacity]
    int* iter_dev = nullptr;     // device-resident iteration number (hipGraph replay); null = use the launch argument
    int iter_stride = 0;         // added to *iter_dev by the last kernel of an iteration
    unsigned plane_stride = 0;   // floats per sample of the radiance planes (FilmArgs); bit 31 = PTK_FILM_STORE
    unsigned depth_stride = 0;   // floats per bounce plane, 0 = one plane for all bounces
    unsigned char* film_flags = nullptr; unsigned flag_sample_stride = 0, flag_depth_stride = 0;      // touched-tile flags of the planes (FilmArgs), null = off
    int spp = 1;                 // samples (iterations) per launch
    int slot_bias = 0;           // added to the compacted slot in the shading RNG key: part * W * H * spp under the tile partition, so
                                 // that two ranks never share a random stream (0 on one GPU = the reference's key)
    int dbg_shift0 = 0, dbg_shift_deep = 0;   // test hooks only (libptk_hooks.so, include/ptk_test_hooks.h): extra key offset at depth 0 / depth >= 1; 0 in the product
    // segmented path streams (default): physical slot of every dense path, window prefix and first-window tables
    int* srcidx = nullptr; int* seg_pref = nullptr; int* seg_first = nullptr;
    int* seg_done = nullptr;     // blocks of the running shading kernel that have published their survivor count (segments_tail); null:
                                 // a separate k_segments launch does the bookkeeping (pipeline 7, A/B)
    bool inline_bvh = false;     // true: one-ray-per-lane traversal inside k_intersect (A/B comparison)
    // fused pipeline (k_shade_trace): per-window hit counts and their prefix, the SPLIT stage's per-slot flags
    int* agg_h = nullptr; int* hit_pref = nullptr; int* tr_valid = nullptr; int* tr_heavy = nullptr;
    int gen_late = 0;            // host hint (pathtrace.hip: note_counts): the camera launch stores the hits' path state only, after its trace phase
    int* hcounts = nullptr;      // fused pipeline, compact streams: hcounts[d] = paths of bounce d that hit something [depth + 2]
    int* seg_first_h = nullptr;  // ... and the window of hit index 256 B [blocks + 16]
    // tiles in flight (ptk_set_tiles_in_flight, pathtrace.hip): this lane renders one band of pixel rows; its compacted slots continue those of
    // the bands above it.  cum_*[ring][d] = paths of bounce d that hit something in this band and all bands above it, ring = iteration mod
    // PTK_CUM_RING (the band above runs ahead); cum_prev = the band above's table (null for the first band), cum_out = this band's; null = off
    const int* cum_prev = nullptr; int* cum_out = nullptr; int cum_stride = 0, ring_div = 1;
    int wide_lpr = 4;            // lanes per ray of the wide walk (4 or 2; ptk_kernels.hip k_traverse_wide)
    unsigned long long* look = nullptr;      // pipeline 11 (k_trace_shade): one look-back word per block of a launch [blocks + 1]
    unsigned long long* look_gave_up = nullptr;   // ... and the word a block raises when it gives the wait up (ptk_kernels.hip: LookBack); look_limit: polls before that
    int look_limit = 1 << 20;
    bool wave_windows = false;   // fused pipeline, compact streams: k_shade_trace_w — one wave and one 64-slot window per unit of work (window tables in units of 64)
    int wide_pool = 0;           // > 0: the wide walk pools that many work-list rays per block (k_traverse_pool); 0: one ray per lane group (k_traverse_wide)
    int walk = PTK_WALK_THREADED; // how the work-list rays are walked (PTK_WALK_THREADED / PTK_WALK_WIDE; AUTO is resolved by the render loop)
    int capacity = 0;
};

int launch_generate(const ::Camera& cam, int iter, const int* iter_ptr, int traceDepth, PtkPaths paths, int n_parts, int part, int tile_rows,
                    int n_pix, int spp, int* d_count0, hipStream_t st);
int launch_bounce_staged(const SceneView& S, int mode, int iter, int depth, int capacity, PtkPaths cur, PtkPaths nxt,
                         PtkHits hits, float* img, const Workspace& ws, hipStream_t st, bool last_bounce);
int launch_bounce_staged_profiled(const SceneView& S, int mode, int iter, int depth, int capacity, PtkPaths cur, PtkPaths nxt,
                                  PtkHits hits, float* img, const Workspace& ws, hipStream_t st, void (*mark)(int, bool), bool last_bounce);
int launch_tail(const SceneView& S, int mode, int iter, int depth0, int trace_depth, PtkPaths cur, PtkPaths nxt, float* img,
                const Workspace& ws, hipStream_t st, const PtkHits* hits0 = nullptr, const int* hr0 = nullptr);
// fused pipeline (k_shade_trace): shade bounce dt - 1 of `cur` (gen: generate the camera rays instead) and trace bounce dt into `nxt`
struct TraceBufs { PtkPaths paths; PtkHits hits; int* hr; };
int launch_shade_trace(const SceneView& S, int mode, bool gen, const ::Camera& cam, int n_parts, int part, int tile_rows, int n_pix, int trace_depth,
                       int iter, int dt, int capacity, const TraceBufs& cur, const TraceBufs& nxt, float* img, const Workspace& ws, hipStream_t st,
                       bool last, void (*mark)(int, bool));
// pipeline 11 (k_trace_shade): launch `depth` traces and shades bounce `depth` (0: camera rays) of `cur` into `nxt`
bool trace_shade_usable(const SceneView& S);
int launch_trace_shade(const SceneView& S, int mode, const ::Camera& cam, int n_parts, int part, int tile_rows, int n_pix, int trace_depth,
                       int iter, int depth, int capacity, PtkPaths cur, PtkPaths nxt, float* img, const Workspace& ws, hipStream_t st, bool last, void (*mark)(int, bool));
int launch_sum_planes(const float* planes, int n_planes, size_t plane_len, float* out, hipStream_t st);      // out = ((p0 + p1) + p2) + ...
int launch_merge_planes(float* image, float* planes, unsigned char* flags, int n_planes, int n_pixels, hipStream_t st);      // image += the touched tiles of planes[0], planes[1], ... in order; planes = 0, flags = 0
int local_path_count(int W, int H, int n_parts, int part, int tile_rows);
int pick_wide_pool(int capacity, int n_lanes);       // rays per block of the pooled wide walk for a lane of `capacity` paths (0 = off); PTK_WIDE_POOL overrides
inline FilmArgs film_of(float* img, const Workspace& ws) { return FilmArgs{img, ws.film_flags, ws.plane_stride, ws.depth_stride, ws.flag_sample_stride, ws.flag_depth_stride}; }

} // namespace ptk
