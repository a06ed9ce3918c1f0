/* ptk.h — C ABI of the MI355X-native path-tracing kernels ("ptk").
 *
 * This is the drop-in boundary for the per-bounce path-tracing inner loop of
 * zhiyi1801/Project3-CUDA-Path-Tracer-1.  Every entry point names the reference
 * interface it replaces (file:line relative to the reference repo).  Plain C:
 * pointers, sizes and PODs only; no C++ / torch / HIP types in any signature
 * (a HIP stream is passed as void*).  Every function returns 0 on success or a
 * non-zero code (a hipError_t value, or PTK_E_* below) and never throws or exits.
 *
 * The C++ API of the reference (src/pathtrace.h:6-9: InitDataContainer /
 * pathtraceInit / pathtraceFree / pathtrace) is provided on top of this ABI by
 * project3-cuda-path-tracer-1_amd/csrc/pathtrace.h with the same four signatures.
 *
 * Data layout (device, HBM): path and hit state are SoA-of-float4 streams, one
 * 16-byte element per path per stream, so that every wave64 access is a fully
 * coalesced 1 KiB transaction:
 *   paths.o_pix  [i] = (origin.x, origin.y, origin.z, bits(pixelIndex))
 *   paths.d_bnc  [i] = (dir.x,    dir.y,    dir.z,    bits(remainingBounces))
 *   paths.col_pdf[i] = (color.r,  color.g,  color.b,  prevPdf)
 *   hits.t_n     [i] = (t, normal.x, normal.y, normal.z)
 *   hits.p_mat   [i] = (point.x, point.y, point.z, bits(materialId))
 *   hits.uv_ids  [i] = (u, v, bits(geomID), bits(triangleID))
 * These carry exactly the fields of PathSegment / ShadeableIntersection
 * (src/sceneStructs.h:80-99), 48 B each.
 */
#ifndef PTK_H
#define PTK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTK_E_BADARG   1001
#define PTK_E_NOSCENE  1002
#define PTK_E_PARSE    1003
#define PTK_E_NODEVICE 1004
#define PTK_E_DEVICE   1005   /* a kernel gave a device-side wait up (pipeline 11 only: its look-back, see ptk_set_pipeline); the image is not valid */

/* enum SampleMode { BSDF, DirectLi, MIS } — src/preview.h:13 */
enum { PTK_MODE_BSDF = 0, PTK_MODE_DIRECTLI = 1, PTK_MODE_MIS = 2 };
/* enum GeomType — src/sceneStructs.h:11-15 */
enum { PTK_GEOM_SPHERE = 0, PTK_GEOM_CUBE = 1, PTK_GEOM_OBJ = 2 };
/* Material::Type — src/material.h:32-38 */
enum { PTK_MAT_LAMBERTIAN = 0, PTK_MAT_METALLIC = 1, PTK_MAT_DIELECTRIC = 2, PTK_MAT_MICROFACET = 3, PTK_MAT_LIGHT = 4 };

/* ---- interchange PODs (host side; dword-for-dword the reference's structs) ------------------- */

/* struct Camera, src/sceneStructs.h:61-70 (84 B) */
typedef struct PtkCamera {
    int   res[2];
    float position[3], lookAt[3], view[3], up[3], right[3];
    float fov[2], pixelLength[2];
} PtkCamera;

/* struct Geom, src/sceneStructs.h:35-47 without translation/rotation/scale/mesh*; matrices column-major (glm::mat4) */
typedef struct PtkGeom {
    int   type, materialid;
    float transform[16], inverseTransform[16], invTranspose[16];
} PtkGeom;

/* class Material, src/material.h:524-538:
 * albedo = Material::albedo (default 1, used by Light), *_s = devTexSampler::fixVal (default 0, used by the BSDFs when the
 * matching *MapID is < 0), *MapID = index into the scene's texture list (-1 = constant sampler) */
typedef struct PtkMaterial {
    int   type;
    float albedo[3];
    float albedo_s[3];
    float rough_s, metal_s, ior;
    int   albedoMapID, metallicMapID, roughnessMapID, normalMapID;
} PtkMaterial;

/* class Triangle, src/Bounds3.hpp:250-258 (124 B) */
typedef struct PtkTriangle {
    float v[9], n[9], tex[6], tangent[3], bitangent[3];
    int   geomIdx;
} PtkTriangle;

/* struct GpuBVHNode, src/BVH.h:44-50 (40 B); 6 direction-ordered copies back to back (BVH.cpp:180-239) */
typedef struct PtkBvhNode {
    float pMin[3], pMax[3];
    int   start, end, miss, hit;
} PtkBvhNode;

/* class lightPrim, src/lightSample.h:26-34 */
typedef struct PtkLight { int geomID, triangleID, type; } PtkLight;

/* ---- device-side handles ------------------------------------------------------------------- */

typedef struct PtkPaths { float* o_pix; float* d_bnc; float* col_pdf; } PtkPaths;   /* device pointers, 4 floats/path each */
typedef struct PtkHits  { float* t_n;   float* p_mat; float* uv_ids;  } PtkHits;    /* device pointers, 4 floats/path each */

typedef struct PtkScene PtkScene;          /* host scene + its device image (class Scene / DevScene, src/scene.h:28-85) */

/* ---- scene: replaces `new Scene(file)` (src/main.cpp:86, src/scene.cpp:47-81), the camera fix-ups of
 *      src/main.cpp:102-127,173-193, and Scene::setDevData (src/scene.cpp:479-567) ------------------ */

/* Parse the scene file (same grammar and quirks as scene.cpp), build triangles, SAH BVH, threaded 6-direction BVH
 * and the light list.  res_w/res_h/depth <= 0 keep the file's RES / DEPTH.  Host only: needs no GPU. */
int  ptk_scene_load(const char* file, int res_w, int res_h, int depth, PtkScene** out);
void ptk_scene_free(PtkScene* s);                       /* Scene::clear + Resource::clear (src/main.cpp:140-141) */
const char* ptk_scene_error(const PtkScene* s);         /* message of the last PTK_E_PARSE */
/* counts[11] = geoms, materials, triangles, bvh_size (nodes per direction), lights, traceDepth, iterations, res.x, res.y,
 *              textures, envMapID (-1 = no ENV map) */
int  ptk_scene_counts(const PtkScene* s, int* counts);
/* Texture `id` (class image, src/image.h:8-51): size, and its float RGB texels (row 0 = bottom row, as loaded by the
 * reference with stbi_set_flip_vertically_on_load(true)). */
int  ptk_scene_get_texture_size(const PtkScene* s, int id, int* width, int* height);
int  ptk_scene_get_texture(const PtkScene* s, int id, float* rgb_out);
int  ptk_scene_get_camera(const PtkScene* s, PtkCamera* out);
int  ptk_scene_get_geoms(const PtkScene* s, PtkGeom* out);
int  ptk_scene_get_materials(const PtkScene* s, PtkMaterial* out);
int  ptk_scene_get_triangles(const PtkScene* s, PtkTriangle* out);
int  ptk_scene_get_bvh(const PtkScene* s, PtkBvhNode* out);          /* 6 * bvh_size nodes */
int  ptk_scene_get_lights(const PtkScene* s, PtkLight* out);
/* Device bytes of the BVH tables of an uploaded scene: out4 = {threaded nodes (one or six direction-ordered copies of BVH.cpp:180-239's table, 32 B
 * per node), 8-wide nodes, 8-wide leaf records, leaf-rank table}.  A mesh of more than 1 024 triangles is uploaded with the direction-0 copy and
 * the rank table that makes any walk order return the reference's winner on exact ties (csrc/ptk_device.hpp, "ONE COPY INSTEAD OF SIX"; SURVEY
 * 8f-4: removal of the 6x MTBVH duplication); ptk_set_pipeline(5) and the traversal counters re-upload all six. */
int  ptk_scene_bvh_device_bytes(const PtkScene* s, unsigned long long* out4);
/* The 8-wide collapsed form of the same BVH (src/BVH.cpp:13-239 builds the binary tree; the collapse and why it is
 * parity-preserving: csrc/wide_bvh.h).  counts[4] = wide nodes (0: the tree cannot be collapsed, the threaded walk is
 * used), leaf records, levels, per-ray stack bound.  nodes: 144 B each, leaves: 64 B each (layout: csrc/ptk_device.hpp). */
int  ptk_scene_wide_bvh_counts(const PtkScene* s, int* counts);
int  ptk_scene_get_wide_bvh(const PtkScene* s, void* nodes, void* leaves);
/* ... and of the 16-wide variant the environment variable PTK_WIDE_WIDTH=16 makes ptk_scene_upload build instead (round 5: built, bit-identical, slower —
 * DESIGN.md section 5.2): width = 8 or 16; nodes of 144 B / 272 B. */
int  ptk_scene_wide_bvh_counts_w(const PtkScene* s, int width, int* counts);
int  ptk_scene_get_wide_bvh_w(const PtkScene* s, int width, void* nodes, void* leaves);
/* Upload the scene tables to the current HIP device (DevScene::initiate, src/scene.cpp:569-663). */
int  ptk_scene_upload(PtkScene* s);

/* ---- kernels: one entry point per reference kernel (src/pathtrace.cu) ---------------------------
 * All pointers are DEVICE pointers unless noted.  `stream` is a hipStream_t (NULL = default stream).
 * `n` is the number of paths; when `d_n` is non-NULL the kernel reads the count from *d_n instead
 * (device-resident count, no host round trip) and `n` is only the launch upper bound. */

/* generateRayFromCamera, pathtrace.cu:135-163.  Writes res.x*res.y paths. */
int ptk_generate_rays(const PtkCamera* cam /*host*/, int iter, int trace_depth, PtkPaths paths, void* stream);

/* computeIntersections, pathtrace.cu:169-346.  valid[i] = rayValid (0/1).  Misses get t = -1.
 * stats (optional, device, 2 x uint64): accumulates BVH node visits and triangle tests.
 * img (optional, device, float[3*P]): with an ENV map in the scene, misses add color * env(direction) to their pixel
 * (pathtrace.cu:303-311); pass NULL to skip that (paths.col_pdf is only read when img is given). */
int ptk_intersect(const PtkScene* s, int n, const int* d_n, PtkPaths paths, PtkHits hits, int* valid,
                  unsigned long long* stats, float* img, void* stream);

/* compact_rays, pathtrace.cu:614-631: stable (order-preserving) stream compaction of both 48-byte streams by
 * `valid`.  Either stream may be omitted (NULL members).  The new count is written to *d_count_out (device). */
int ptk_compact(int n, const int* d_n, const int* valid, PtkPaths paths_in, PtkHits hits_in,
                PtkPaths paths_out, PtkHits hits_out, int* d_count_out, void* stream);

/* PTkernel / DirectLiPTkernel / MisPTkernel, pathtrace.cu:348-588, selected by `mode` (sampleMode, main.cpp:20).
 * Shades n compacted paths in place; thread index == compacted slot index == RNG key.  img: float[3*P] (+=). */
int ptk_shade(const PtkScene* s, int mode, int iter, int depth, int n, const int* d_n, PtkHits hits,
              PtkPaths paths, int* valid, float* img, void* stream);

/* sendImageToPBO, pathtrace.cu:32-65: ACES + gamma 1/2.2 -> uchar4 (w = 0). */
int ptk_tonemap(const float* img, int res_x, int res_y, int iter, unsigned char* pbo, void* stream);

/* ---- render API: the C form of src/pathtrace.h:6-9 ------------------------------------------------ */

int ptk_pathtrace_init(PtkScene* s);                    /* pathtraceInit(Scene*), pathtrace.cu:85-112 */
int ptk_pathtrace_free(void);                           /* pathtraceFree(), pathtrace.cu:114-125; null-safe, repeatable */
/* pathtrace(uchar4* pbo, int frame, int iter), pathtrace.cu:637-816.  pbo: device pointer or NULL (skip tonemap).
 * iter is 1-based.  Accumulates into the device radiance buffer; no host synchronisation. */
int ptk_pathtrace(unsigned char* pbo, int frame, int iter);
int ptk_set_sample_mode(int mode);                      /* `extern SampleMode sampleMode`, main.cpp:20 / pathtrace.cu:23 */
int ptk_get_sample_mode(void);
/* Tiles in flight (1..16, default 1 = off; before ptk_pathtrace_init; not combined with iterations in flight, the tile partition or samples
 * per launch; needs the fused pipeline 0 / 10).  Every iteration is rendered as K bands of pixel rows on K streams — band k of iteration i beside
 * band k - 1 of iteration i + 1 — so K launches are in flight like with K iterations in flight, but every pixel still receives its iterations in
 * order and every path keeps the reference's RNG key (pathtrace.cu:458,532: a band's compacted slots continue those of the bands above it): the
 * image is BIT-IDENTICAL to the default's, i.e. to the reference's own accumulation order (pathtrace.cu:697-810), without radiance planes or
 * merge launches.  (Iterations in flight, below, reach the same image through ordered accumulation and are the faster of the two.) */
int ptk_set_tiles_in_flight(int tiles);
/* Iterations in flight (1..16, default 1; set before ptk_pathtrace_init).  With L > 1, the call `step` runs on lane
 * (step-1) % L (own stream, buffers and radiance planes), so consecutive ptk_pathtrace calls overlap on the GPU.  Every
 * iteration is computed exactly as with L = 1 into its lane's planes — zero when it starts; one plane, or one per bounce when
 * the context is initialised in MIS mode, so that every (pixel, plane) receives at most one add (pathtrace.cu:398,486,556,572,
 * 303-311) — and the planes are folded into the image when the iteration is done, in call order and bounce by bounce (ordered
 * accumulation): image = ((image + a_1) + a_2) + ..., the reference's own sequence of adds (pathtrace.cu:600).  The image is
 * BIT-IDENTICAL to one iteration at a time in every sample mode.  Not a reference feature.
 * Device memory of the planes: L x S x D x 12 B x W x H, D = traceDepth + 1 when the context is initialised in MIS mode with S = 1, else 1
 * (cornell.txt 800 x 800 depth 8 MIS, L = 3: 208 MB; 2048 x 2048 depth 16 MIS, L = 3: 2.6 GB) + one flag byte per 64 pixels and plane; when a
 * sample's planes would exceed 2^31 floats the context falls back to D = 1 (MIS then merges the iteration's own sum: a valid image with another
 * float association).  ptk_pathtrace_init returns the hipMalloc error if the planes do not fit. */
int ptk_set_iterations_in_flight(int lanes);
/* Accumulation policy with several radiance planes (L iterations in flight and / or S samples per launch; set before ptk_pathtrace_init).
 * -1 / 1 (default): ordered, as described above (one small merge launch per call; measured -3 % on cornell.txt BSDF, -5 % MIS, -7..-10 % on the
 * short iterations of mis_test.txt 1024 x 1024).  Not bit-identical to one iteration at a time, though valid: MIS with samples per launch (one plane
 * per sample) and MIS after ptk_set_sample_mode on a context initialised in another mode (one plane; re-initialise instead).
 * 0: the planes accumulate — plane l sums the calls l+1, l+1+L, ... — and are summed when the image is read: ((plane0 + plane1) + plane2) + ... */
int ptk_set_ordered_accumulation(int on);
/* Samples per launch (1..64, default 1; set before ptk_pathtrace_init).  With spp > 1, ptk_pathtrace(.., step) traces
 * iterations (step-1)*spp + 1 .. step*spp of the owned pixels in one set of launches: camera jitter is keyed per iteration
 * as usual, the shading RNG by (first iteration of the batch, compacted slot, depth), and every sample accumulates into its
 * own radiance plane, folded into the image in sample order when the launch is done or summed when the image is read
 * (ptk_set_ordered_accumulation).  Used with the tile partition so that a rank that owns
 * 1/N of the pixels still launches full-size kernels (multi-GPU weak scaling).  Not a reference feature. */
int ptk_set_samples_per_launch(int spp);
/* Opt-in "fixed" cube lights.  The reference's LightSampler gives cube lights neither a sampled position nor a
 * pdf (lightSample.h:141-248): MIS renders of cube-lit scenes (cornell.txt) are darkened and direct lighting from them is
 * black.  on = 1: a cube light is sampled uniformly over its surface (face in proportion to world-space area) with the
 * area -> solid-angle pdf of a triangle light, and lightPDF returns the same for a BSDF-sampled hit; MIS and BSDF renders
 * then converge to the same image.  Default 0 = reference behaviour.  Applies to the kernels launched afterwards. */
int ptk_set_light_fix(int on);
/* 1: replay every iteration from a hipGraph captured once per lane (the kernels then read the iteration number from device
 * memory); 0 (default): launch the kernels one by one.  Identical results; cuts the host cost per iteration. */
int ptk_set_graph_replay(int on);
/* Copy the running radiance SUM (not mean) into host memory, float[3*P], row-major x + y*W
 * (the per-iteration cudaMemcpy of pathtrace.cu:809-810, made lazy).  Synchronises. */
int ptk_get_image(float* host_out);
/* Device pointer of the radiance buffer (float[3*P]) for collectives; valid until ptk_pathtrace_free. */
int ptk_get_image_device(float** out);
/* Device-to-device copy of the radiance buffer into caller-owned device memory (e.g. a tensor handed to RCCL).  Runs on the render
 * context's own NON-BLOCKING stream and synchronises it before returning; work the caller has queued on dst on another stream (a fill,
 * say) must have completed before the call — the Python binding waits for torch's current stream. */
int ptk_copy_image_to_device(float* dst_device);
/* Per-bounce live-path counts of the LAST ptk_pathtrace call (counts[trace_depth+1], host); synchronises.
 * *traced_depth = number of bounces with live paths (guiData->TracedDepth, pathtrace.cu:793-796). */
int ptk_get_live_counts(int* counts, int* traced_depth);
/* Total rays (sum of live paths entering the intersection stage) since ptk_pathtrace_init; synchronises. */
int ptk_get_ray_total(unsigned long long* rays);
/* Restrict rendering to the pixels of interleaved tiles: pixel (x,y) is traced iff
 * ((y / tile_rows) % n_parts) == part.  (1,0,*) = whole frame.  Multi-GPU sharding, SURVEY §8e. */
int ptk_set_tile_partition(int n_parts, int part, int tile_rows);
/* Iteration sharding — the exact multi-GPU validation mode of SURVEY §8e (set before ptk_pathtrace_init; (1,0) = off).
 * ptk_pathtrace(.., step) then renders iteration (step-1)*n_shards + shard + 1 of the WHOLE frame, with the reference's own
 * RNG keys (camera: (x, y, iter), pathtrace.cu:148 — shading: (iter, compacted slot, depth), pathtrace.cu:458,532), so shard g
 * of n computes exactly the iterations g+1, g+1+n, ... of a single-GPU render: its radiance buffer is the running sum of exactly
 * those iterations, bit for bit, and the sum of the shards is the single-GPU image up to float association (the same
 * per-iteration radiances, added shard by shard instead of in iteration order). */
int ptk_set_iteration_sharding(int n_shards, int shard);
/* Bounces currently enqueued at full width (the remaining ones run in the single-workgroup tail kernel); diagnostic. */
int ptk_get_full_width_bounces(int* out);
/* 1 when the render loop has learnt (hit counts of earlier iterations, fused pipeline with compact streams) that fewer than about half of the
 * camera rays hit anything: the camera launch then stores the path state of the hits only, after its trace phase (generateRayFromCamera,
 * pathtrace.cu:120-167, writes every path; a miss is never read again here).  Diagnostic; results never depend on it. */
int ptk_get_open_scene_hint(int* out);
/* Host-only description of that partition (no GPU): paths owned by `part`, and the pixel index (x + y*W) of its j-th path. */
int ptk_partition_count(int W, int H, int n_parts, int part, int tile_rows);
int ptk_partition_pixel(int W, int n_parts, int part, int tile_rows, int j);
/* Pipeline selection (A/B comparisons; all produce identical results):
 *   0 = default: the fused launches of 10 for scenes without triangles, for meshes of up to 1 024 triangles in all (the reference's walk then runs
 *       inside the fused kernel's trace phase, per lane: no work list, one launch per bounce) and, on larger meshes, in MIS mode (its
 *       shadow rays run in a balanced block phase); the staged kernels of 8 for larger meshes in BSDF / DIRECT mode (a BVH stage behind
 *       every launch leaves the fusion nothing to win there: measured 1-3 % behind);
 *   10 = one launch per bounce: k_shade_trace shades bounce d-1 and traces bounce d for the
 *       survivors while their new rays are still in registers (the first launch of an iteration generates the camera rays
 *       instead of shading); path state and hit record are written together at the survivor's slot in its block's window, the
 *       compacted slot of a hit path (the reference's RNG key) is hit_pref[window] + its rank among the window's hits, and the last
 *       block of every launch builds the window tables.  Triangle-free scenes without textures use compact streams (four 16-byte
 *       records per path, five in MIS: only what shading really reads).  Every layout without a BVH stage shades HIT-DENSE: a launch
 *       enumerates the paths that hit something (the reference's compacted slot is the index itself) through the hit prefix and a per-window
 *       inverse map the tracing block writes; a miss is never loaded again.  Next-event shadow rays (MIS / DIRECT) are tested in a block-cooperative
 *       any-hit phase between shading and tracing.  Scenes with larger meshes continue each launch with the work-list BVH stage
 *       (stable compaction of the rays that enter the root box, BVH walk over the list, finalize).  While the traversal counters
 *       are enabled the staged kernels of 8 run instead (they keep the counters);
 *   11 = (round 5, opt-in) k_trace_shade: a launch TRACES and SHADES one bounce (the first one generates the camera rays), traceDepth launches per
 *       iteration; the hit record never leaves the block, the streams are the reference's PathSegment as three 16-byte records (48 B in, 48 B out per
 *       surviving path-bounce); the compacted slot of a hit path comes from a decoupled look-back over the blocks' hit counts INSIDE the launch
 *       (one tagged 64-bit word per block).  Triangle-free scenes without textures / ENV map; elsewhere it behaves as 8; not with tiles in flight;
 *       ONE ITERATION AT A TIME only — with several iterations in flight the context renders with 0's kernels: blocks that spin for earlier blocks
 *       of their launch can starve another launch's blocks of the XCD they need (a hang was observed with four in flight).
 *       The wait is bounded (PTK_LB_POLL_LIMIT looks, default 2^20): a block that gives up lets the launch drain and the next read of the
 *       image returns PTK_E_DEVICE — what two processes sharing one GPU with this pipeline may get instead of a hung GPU.
 *       Bit-identical; slower than 0 today (every block waits for the slowest earlier block of the launch: 11.4 against 12.2 Grays/s on
 *       cornell.txt one iteration at a time; three in flight, before the restriction: 16.7 / 21.3, 23.7 if the wait cost nothing — profiles/r5_ab_trace_shade.txt);
 *   7 / 8 = staged (the default until round 3): per bounce an intersect stage, shading with block-local compaction of the
 *       survivors, and the window prefix in a separate one-block k_segments launch (7) or in the last block of the shading kernel (8);
 *   1 = like 7, the BVH is walked inside the intersect kernel for every ray, one ray per lane (no k_tail);
 *   5 = like 7/8, the work-list rays walk the reference's threaded 6-direction BVH, one ray per lane (re-uploads all six direction copies; the
 *       others walk the 8-wide collapsed form of the same tree, several lanes per ray, csrc/wide_bvh.h); 6 = the wide walk with the staged kernels;
 *   2, 3, 4, 9 = retired A/B layouts (PTK_E_BADARG): 3 the ticketed one-kernel-per-bounce fusion of rounds 1-2 (round 3: its ticket counter
 *       serialised block starts); 2 wave-uniform primitive loops instead of the LDS pair lists, 4 survivors moved by a stable scatter pass
 *       instead of the window tables, 9 every bounce at full width without k_tail (round 4: each lost its measurement in rounds 1-3). */
int ptk_set_pipeline(int which);
int ptk_stream_sync(void);

/* Per-kernel device time, measured with HIP events ON THE RENDER STREAM around every launch of the ptk_pathtrace calls
 * issued between begin and end.  ms4/launches4: [0] generate, [1] intersect (pipeline 0: the work-list BVH stage and k_tail), [2] shade (pipeline 0: k_shade_trace; its first launch of an iteration counts as generate), [3] compact (k_segments / k_scatter).  Diagnostic only
 * (adds an event pair per launch): never used inside a timed region. */
int ptk_profile_begin(void);
int ptk_profile_end(float* ms4, int* launches4);
/* BVH traversal counters of the intersection kernel: out2 = {node visits, triangle tests} since enabling. */
int ptk_enable_traversal_stats(int on);
int ptk_get_traversal_stats(unsigned long long* out2);

/* HIP events on the render stream, for benchmark timing of the timed region (bench.py). */
int ptk_timer_start(void);
int ptk_timer_stop(float* ms);

#ifdef __cplusplus
}
#endif
#endif /* PTK_H */
