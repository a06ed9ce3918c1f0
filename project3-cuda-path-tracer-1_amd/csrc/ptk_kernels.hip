// ptk_kernels.hip — HIP kernels of the path-tracing hot path for gfx950 + their C-ABI launchers.
//
// Reference kernel (src/pathtrace.cu) -> kernels here:
//   generateRayFromCamera (135-163)                       -> k_generate
//   computeIntersections  (169-346)                       -> k_intersect (+ k_list, k_traverse_wide | k_traverse, k_finalize for scenes with triangles)
//   PTkernel / DirectLiPTkernel / MisPTkernel (348-588)   -> k_shade_binned<M> (render pipeline), k_shade<M> (kernel-level ABI)
//   compact_rays (614-631; thrust scan + 2x scatter_if + 2 blocking D2H), twice per bounce
//                                                         -> nothing moves: virtual compaction of the hits + block-local compaction
//                                                            of the survivors + segments_tail / k_segments (window tables); k_count + k_scatter
//                                                            remain as the stand-alone stable compaction (ptk_compact, pipeline 4)
//   sendImageToPBO (32-65)                                -> k_tonemap
//   (shade bounce d-1 + trace bounce d in one launch)     -> k_shade_trace (default render kernel)
//   (the sparse end of an iteration in one launch)        -> k_tail
// What is specific to MI355X:
//   * 256-thread blocks = 4 wave64; all state streams are float4 SoA (16 B/lane, 1 KiB per wave access);
//   * scene constants are read with wave-uniform indices through the constant address space -> scalar loads / SGPR operands;
//     per-lane primitive and material records come from LDS copies staged once per block;
//   * analytic primitives: conservative candidate test, then the block's (ray, primitive) pairs balanced over all lanes
//     through an LDS list, closest hit by ds_min_u64 (analytic_closest_block);
//   * path counts never leave the device: every kernel takes a device-resident count, so a whole iteration is enqueued
//     (or replayed from a hipGraph) without a single host synchronisation;
//   * the reference's two compactions per bounce are index arithmetic: a hit path's compacted slot (its RNG key) comes from
//     the intersect flags + per-block aggregates, survivors are written once, block-locally, and found again through the
//     window prefix (gather_source).  See DESIGN.md section 4/5.
#include "ptk_internal.h"
#ifdef PTK_TEST_HOOKS
#include "../../include/ptk_test_hooks.h"
#endif

using namespace ptk;

#define PTK_BLOCK 256
#define PTK_SEG_GROUP 32          // blocks per first-level done counter (segments_tail)
#define PTK_WAVES (PTK_BLOCK / 64)

// Diagnostic build only (make EXTRA_HIPFLAGS=-DPTK_PHASE_TIMING, tools/diag/phase_timing.py): where a wave of k_intersect /
// k_shade_binned spends its LIFETIME — s_memtime stamps at phase boundaries (with s_waitcnt(0), so phases that would overlap are
// serialised: upper bounds), added into a per-wave slot with plain stores; one stream only.  Off in the product.
#ifdef PTK_PHASE_TIMING
#define PTK_PHASE_SLOTS 16384
__device__ unsigned long long g_phase_buf[2][PTK_PHASE_SLOTS][8];      // per kernel, per wave slot: cycles per phase ([7] = visits)
#define PTK_PHASE_BEGIN() unsigned long long ph_prev = __builtin_amdgcn_s_memtime(); int ph_k = 0; \
    const unsigned ph_slot = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % PTK_PHASE_SLOTS
#define PTK_PHASE(K) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long ph_now = __builtin_amdgcn_s_memtime(); \
    if ((threadIdx.x & 63) == 0) g_phase_buf[K][ph_slot][ph_k] += ph_now - ph_prev; ph_prev = __builtin_amdgcn_s_memtime(); ++ph_k; } while (0)
#define PTK_PHASE_END(K) do { if ((threadIdx.x & 63) == 0) g_phase_buf[K][ph_slot][7] += 1ull; } while (0)
// (k_tail, end of round 5) a stamp into a FIXED phase index: the phases repeat per bounce
#define PTK_PHASE_AT(K, IDX) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long ph_now = __builtin_amdgcn_s_memtime(); \
    if ((threadIdx.x & 63) == 0) g_phase_buf[K][ph_slot][IDX] += ph_now - ph_prev; ph_prev = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PTK_PHASE_AT(K, IDX) do {} while (0)
#define PTK_PHASE_BEGIN() do {} while (0)
#define PTK_PHASE(K) do {} while (0)
#define PTK_PHASE_END(K) do {} while (0)
#endif

namespace {

// ------------------------------------------------------------------------------------------------------------------
// block-level primitives
// ------------------------------------------------------------------------------------------------------------------
// exclusive prefix of `flag` over the block in thread order; `total` = number of set flags in the block
__device__ __forceinline__ int block_excl_prefix(int flag, int& total) {
    __shared__ int wsum[PTK_WAVES];
    const unsigned long long bal = __ballot(flag != 0);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int within = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wid] = __popcll(bal);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < PTK_WAVES; ++w) { const int c = wsum[w]; off += (w < wid) ? c : 0; tot += c; }
    __syncthreads();
    total = tot;
    return off + within;
}

// Tiles in flight: the compacted slot of a hit path = its rank among the hits of ITS band + the hits of the same bounce in all bands above
// (a band is a contiguous range of the reference's pixel order, and stable compaction keeps it contiguous at every bounce).  The band above has
// finished this iteration before this one starts it (pathtrace.hip), so its cumulative counts are final; `row` = this iteration's ring slot.
__device__ __forceinline__ int cum_row(int step, int ring_div, int cum_stride) { return (((step - 1) / ring_div) & (PTK_CUM_RING - 1)) * cum_stride; }
__device__ __forceinline__ int cum_read(const int* cum_prev, int row, int d) { return cum_prev ? cum_prev[row + d] : 0; }

// sum of agg[0 .. b): every block reduces its predecessors' aggregates itself (<= 10 loads/thread at 800x800)
__device__ __forceinline__ int prior_blocks(const int* __restrict__ agg, int b) {
    __shared__ int red[PTK_WAVES];
    int s = 0;
    for (int k = threadIdx.x; k < b; k += PTK_BLOCK) s += agg[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    int t = 0;
#pragma unroll
    for (int w = 0; w < PTK_WAVES; ++w) t += red[w];
    __syncthreads();
    return t;
}

// exclusive prefix of agg[0 .. nw) into out[0 .. nw], out[nw .. nw + 8) = total (one block; AGENT as in segments_block)
template <int NT, bool AGENT>
__device__ __forceinline__ int prefix_block(int nw, const int* __restrict__ agg, int* __restrict__ out) {
    constexpr int NW = NT / 64;
    __shared__ int s_w[NW];
    __shared__ int s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < nw; base += NT) {
        const int idx = base + tid;
        int v = 0;
        if (idx < nw) v = AGENT ? __hip_atomic_load(agg + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : agg[idx];
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) s_w[wid] = incl;
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { const int cw = s_w[w]; woff += (w < wid) ? cw : 0; tot += cw; }
        const int carry = s_carry;
        if (idx < nw) out[idx] = carry + woff + incl - v;
        __syncthreads();
        if (tid == 0) s_carry = carry + tot;
        __syncthreads();
    }
    const int total = s_carry;
    if (tid < 8) out[nw + tid] = total;
    return total;
}

// pixelIndex -> its three floats in the radiance planes (FilmArgs, ptk_internal.h).  With samples per launch (spp > 1) bits 24..30 of pixelIndex carry
// the sample's plane; a launch with spp == 1 passes plane_stride = 0 and all 31 bits are the pixel (frames of 2^24 pixels and more).  With planes per
// depth (ordered accumulation in MIS, pathtrace.hip: merge_lane) the add of bounce `depth` goes to that bounce's plane.  Wave-uniform selects.
// Bit 31 of plane_stride (PTK_FILM_STORE; 3 P < 2^31, pathtrace.hip: context_init): the plane is zero when the iteration starts and receives at most ONE
// add per pixel — the add is a plain store, no dependent load.  flags (may be null): one byte per 64 pixels and plane, set by every add, so that the
// merge visits touched tiles only.
__device__ __forceinline__ void film_add(const FilmArgs& f, int depth, int pix, f3 c) {
    const unsigned stride = f.plane_stride & ~PTK_FILM_STORE;
    const unsigned mask = stride ? 0xFFFFFFu : 0x7FFFFFFFu;
    const unsigned plane = stride ? ((unsigned)pix >> 24) : 0u;
    const unsigned px = (unsigned)pix & mask;
    const unsigned d = f.depth_stride ? (unsigned)depth : 0u;
    float* p = f.img + (size_t)plane * stride + (size_t)d * f.depth_stride + 3 * (size_t)px;
    if (f.flags) f.flags[(size_t)plane * f.flag_sample_stride + (size_t)d * f.flag_depth_stride + (px >> 6)] = 1;      // wave-uniform branch
    // one live path per (sample plane, pixel): plain read-modify-write, no atomics
    if (f.plane_stride & PTK_FILM_STORE) { p[0] = c.x; p[1] = c.y; p[2] = c.z; }      // wave-uniform
    else { p[0] += c.x; p[1] += c.y; p[2] += c.z; }
}
__device__ __forceinline__ float4 ld4(const float* p, int i) { return reinterpret_cast<const float4*>(p)[i]; }
__device__ __forceinline__ void st4(float* p, int i, float4 v) { reinterpret_cast<float4*>(p)[i] = v; }

struct DevCamera { int W, H; f3 pos, view, up, right; float plx, ply; };

// ------------------------------------------------------------------------------------------------------------------
// generateRayFromCamera, pathtrace.cu:135-163.  Path j -> pixel (x,y) through the tile partition.
// ------------------------------------------------------------------------------------------------------------------
// Samples per launch (spp > 1, not a reference feature): launch `step` traces iterations (step-1)*spp + s + 1, s = 0..spp-1,
// of the owned pixels at once; path j = s * n_pix + (pixel slot), and the sample index rides in bits 24..30 of pixelIndex so
// that each sample accumulates into its own radiance plane (two samples of one pixel may be shaded concurrently).
__global__ void __launch_bounds__(PTK_BLOCK) k_generate(DevCamera cam, int iter_arg, const int* __restrict__ iter_ptr, int traceDepth,
                                                         PtkPaths paths, int n_parts, int part, int tile_rows, int n_pix, int spp, int* count0) {
    const int step = iter_ptr ? *iter_ptr : iter_arg;      // device-resident iteration number when replayed from a hipGraph
    const int n_local = n_pix * spp;
    const int j = blockIdx.x * PTK_BLOCK + threadIdx.x;
    if (j == 0 && count0) *count0 = n_local;
    if (j >= n_local) return;
    const int s = j / n_pix, jp = j - s * n_pix;
    const int iter = (step - 1) * spp + s + 1;
    const int lr = jp / cam.W, x = jp - lr * cam.W;
    const int y = ((lr / tile_rows) * n_parts + part) * tile_rows + (lr % tile_rows);
    Rng rng = make_rng(x, y, iter);                       // argument order of pathtrace.cu:148: (iter:=x, index:=y, depth:=iter)
    const f2 r = sample2D(rng);
    const float fx = static_cast<float>(static_cast<float>(x) + ((double)r.x - 0.5) - cam.W * 0.5f);
    const float fy = static_cast<float>(static_cast<float>(y) + ((double)r.y - 0.5) - cam.H * 0.5f);
    const f3 d = normalize(cam.view - cam.right * cam.plx * fx - cam.up * cam.ply * fy);
    st4(paths.o_pix, j, make_float4(cam.pos.x, cam.pos.y, cam.pos.z, __int_as_float((s << 24) | (x + y * cam.W))));
    st4(paths.d_bnc, j, make_float4(d.x, d.y, d.z, __int_as_float(traceDepth)));
    st4(paths.col_pdf, j, make_float4(1.f, 1.f, 1.f, -1.f));
}

// Closest analytic primitive for every ray of the block, block-cooperatively (all NT threads must call it).
//
// The per-lane candidate loops of analytic_closest run max-over-lanes(candidates) exact tests per wave although the MEAN is
// about half of that (cornell: 1.4 candidates per ray, 2.0 + 1.4 loop trips per wave).  Here the (ray, primitive) candidate
// pairs of the whole block are written to one LDS list — cube pairs from the front, sphere pairs behind them — and thread j
// tests pair j, j + NT, ...: every lane of every wave holds a pair, and at most one wave per round mixes the two types.
// Winner per ray: ds_min_u64 on {t bits : 32 | primitive index : 32} — smaller t, then smaller index, which is the
// reference's "first in index order among equal distances" (strict t_min > t, pathtrace.cu:228), for any pair order;
// t > 0 floats order like their bit patterns.  After a barrier the thread whose key is the ray's minimum stores the hit
// point and the deferred normal for it (a later round can only lower the key, and then overwrites).
// WAVE = true (round 5, k_shade_trace_w): the "block" is ONE wave — NT = 64 rays, the lane is the thread index, every array belongs to the wave alone —
// and a barrier is a compiler fence: a wave's LDS instructions execute in issue order, so a later read of any lane sees the earlier writes of all lanes.
// (No fence instruction: a fence — even at wavefront scope — may make the compiler drain the vector-memory counter, i.e. wait for the path stores that are
// meant to overlap the tracing.  The compiler barrier keeps the program order of the LDS accesses; measured: no difference either way.)
__device__ __forceinline__ void wave_sync() { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); }
template <bool WAVE> __device__ __forceinline__ void group_sync() { if (WAVE) wave_sync(); else __syncthreads(); }
// LDS the block-cooperative closest-hit search works in (24.1 KB for 256 threads)
struct TraceLds {
    float4* ray;                 // [2 * NT]  origins [0, NT), directions [NT, 2 NT) of the block's rays: two planes, so that the 16-byte reads of the pair
                                 //           list's lanes (neighbouring rays) fall on neighbouring 16-byte slots — interleaved (origin, direction) pairs put eight
                                 //           lanes on two LDS rows (round 3's counters: 25 % of the LDS cycles of k_shade_trace were bank conflicts)
    float2* res;                 // [3 * NT]  winner's hit point + deferred normal, 24 B per ray (with 32 B the block's static LDS crosses
                                 //           160 KB / 6: five blocks per CU instead of six)
    unsigned long long* best;    // [NT]
    unsigned short* list;        // [NT * 16] pair list
    int* cnt;                    // [NT / 64]
};
// RAYS_IN_LDS: L.ray already holds the rays (written by other threads of the block, a barrier since); else every thread stores its own
// R3 (round 5, the eight-waves-per-SIMD experiment): the rays are SIX float planes ox, oy, oz, dx, dy, dz of NT entries at L.ray — 24 instead of 32 B per ray
// (the fourth components carry nothing the search reads); RAYS_IN_LDS only.
template <int NT, bool RAYS_IN_LDS, int LP = 16, bool WAVE = false, bool R3 = false>       // LP: primitives per pass of the pair list (list = NT * LP entries)
__device__ __forceinline__ void analytic_closest_lds(const SceneView& S, const GeomStage& gs, const TraceLds& L, bool active, f3 ro, f3 rd, Analytic& a) {
    static_assert(!WAVE || NT == 64, "a wave-level group is one wave");
    static_assert(!R3 || RAYS_IN_LDS, "the plane layout is written by the caller");
    constexpr int NW = NT / 64;
    constexpr int RB = NT == 512 ? 9 : 8;                 // bits of the ray index inside a list entry
    float4* s_ray = L.ray; float2* s_res = L.res; unsigned long long* s_best = L.best; unsigned short* s_list = L.list; int* s_cnt = L.cnt;
    const int tid = WAVE ? (int)(threadIdx.x & 63) : (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (!RAYS_IN_LDS) {
        s_ray[tid] = make_float4(ro.x, ro.y, ro.z, 0.f);
        s_ray[NT + tid] = make_float4(rd.x, rd.y, rd.z, 0.f);
    }
    s_best[tid] = ~0ull;
    for (int base = 0; base < S.n_geoms; base += 32) {
        const int n = (S.n_geoms - base) < 32 ? (S.n_geoms - base) : 32;
#ifdef PTK_DIAG_DOUBLE_CAND      // timing experiments only (tools/diag): the section runs twice on laundered inputs, same results
        f3 ro2 = ro; asm volatile("" : "+v"(ro2.x), "+v"(ro2.y), "+v"(ro2.z));
        const Cand call0 = analytic_candidates(S, base, n, ro2, rd);
        Cand call = analytic_candidates(S, base, n, ro, rd);
        call.cube |= call0.cube & 0u; asm volatile("" :: "v"(call0.cube), "v"(call0.sphere));
#else
        const Cand call = analytic_candidates(S, base, n, ro, rd);
#endif
        for (int half = 0; half * LP < n; ++half) {       // the pair list holds LP primitives x NT rays
            Cand c = call;
            const unsigned hm = ((1u << LP) - 1u) << (LP * half);
            c.cube &= hm; c.sphere &= hm;
            if (!active) { c.cube = 0u; c.sphere = 0u; }
            // block-wide exclusive prefix of (cube count | sphere count << 16); a wave total is at most 64 * 16 per half
            const int v = __popc(c.cube) | (__popc(c.sphere) << 16);
            int incl = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
            if (lane == 63) s_cnt[wid] = incl;
            group_sync<WAVE>();                              // also publishes s_ray / s_best (first pass) and ends the previous pass's reads
            int woff_c = 0, woff_s = 0, Tc = 0, Ts = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const int cw = s_cnt[w];
                if (w < wid) { woff_c += cw & 0xffff; woff_s += cw >> 16; }
                Tc += cw & 0xffff; Ts += cw >> 16;
            }
            int oc = woff_c + ((incl - v) & 0xffff), os = Tc + woff_s + ((incl - v) >> 16);
            for (unsigned m = c.cube; m; m &= m - 1u) s_list[oc++] = (unsigned short)(tid | (__builtin_ctz(m) << RB));
            for (unsigned m = c.sphere; m; m &= m - 1u) s_list[os++] = (unsigned short)(tid | (__builtin_ctz(m) << RB));
            group_sync<WAVE>();
            const int T = Tc + Ts;
            // WAVE: a round of 64 lanes would usually MIX cube and sphere pairs and then run both tests (the 256-lane list mixes them in one wave of
            // four at most): cube rounds first, sphere rounds behind them — the sphere part of the list starts at a multiple of 64 (measured: the
            // mixed rounds cost k_shade_trace_w 27 % more vector instructions than k_shade_trace)
            const int Tc_pad = WAVE ? (Tc + NT - 1) / NT * NT : Tc;
            for (int r0 = 0; r0 < Tc_pad + Ts; r0 += NT) {
                int pi = r0 + tid;
                if (WAVE) pi = pi < Tc_pad ? (pi < Tc ? pi : T) : Tc + (pi - Tc_pad);      // (padding lanes: no pair)
                unsigned long long key = ~0ull;
                int r = 0;
                f3 P = mk3(0, 0, 0), NSv = mk3(0, 0, 0);
                if (pi < T) {
                    const unsigned e = s_list[pi];
                    r = (int)(e & ((1u << RB) - 1u));
                    const int k = (int)(e >> RB);
                    f3 po, pd;
                    if (R3) {
                        const float* s_r3 = reinterpret_cast<const float*>(s_ray);
                        po = mk3(s_r3[r], s_r3[NT + r], s_r3[2 * NT + r]); pd = mk3(s_r3[3 * NT + r], s_r3[4 * NT + r], s_r3[5 * NT + r]);
                    } else {
                        const float4 o4 = s_ray[r], d4 = s_ray[NT + r];
                        po = mk3(o4.x, o4.y, o4.z); pd = mk3(d4.x, d4.y, d4.z);
                    }
                    const GeomXfRef g = ref_xf(gs.xf, base + k, (c.diag >> k) & 1);
#ifdef PTK_DIAG_DOUBLE_TEST      // timing experiments only
                    { f3 po2 = po; asm volatile("" : "+v"(po2.x), "+v"(po2.y), "+v"(po2.z)); f3 P2 = mk3(0, 0, 0), N2 = mk3(0, 0, 0);
                      const float t2 = pi < Tc ? box_test(g, po2, pd, P2, N2) : sphere_test(g, po2, pd, P2, N2);
                      asm volatile("" :: "v"(t2), "v"(P2.x), "v"(P2.y), "v"(P2.z), "v"(N2.x), "v"(N2.y), "v"(N2.z)); }
#endif
                    const float t = pi < Tc ? box_test(g, po, pd, P, NSv) : sphere_test(g, po, pd, P, NSv);
                    if (t > 0.0f && FLT_MAX > t) {
                        key = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned)(base + k);
                        atomicMin(&s_best[r], key);
                    }
                }
                group_sync<WAVE>();
                if (key != ~0ull && s_best[r] == key) {
                    s_res[3 * r] = make_float2(P.x, P.y);
                    s_res[3 * r + 1] = make_float2(P.z, NSv.x);
                    s_res[3 * r + 2] = make_float2(NSv.y, NSv.z);
                }
            }
            group_sync<WAVE>();
        }
    }
    a.t_min = FLT_MAX; a.geom = -1; a.P = mk3(0, 0, 0); a.NS = mk3(0, 0, 0);
    const unsigned long long best = s_best[tid];
    if (best != ~0ull) {
        const float2 r0 = s_res[3 * tid], r1 = s_res[3 * tid + 1], r2 = s_res[3 * tid + 2];
        a.t_min = __uint_as_float((unsigned)(best >> 32)); a.geom = (int)(unsigned)best;
        a.P = mk3(r0.x, r0.y, r1.x); a.NS = mk3(r1.y, r2.x, r2.y);
    }
}
// Any-hit over the analytic primitives for the block's SHADOW rays (the first loop of LightSampler::occulusionTest, lightSample.h:61-88),
// block-cooperatively like analytic_closest_lds: candidate test per ray, one LDS pair list (cube pairs from the front, sphere pairs behind
// them), thread j tests pairs j, j + NT, ... and ORs the reference's occlusion condition into the ray's bit of a 256-bit LDS mask.  The
// result is a boolean OR over the primitives — order-independent, and the primitives the candidate test culls provably miss (returns 0,
// tests/test_candidate_margin.py) — so it equals the per-lane loop of occluded_analytic.  All NT threads must call it; ends with a barrier.
//   L.ray[r] = (origin, minT), L.ray[NT + r] = (direction, -) of shadow ray r, written by thread r before the caller's last barrier.
template <int NT, int LP, bool WAVE = false>
__device__ __forceinline__ void analytic_anyhit_lds(const SceneView& S, const GeomStage& gs, const TraceLds& L, unsigned long long* s_occ, bool active, f3 ro, f3 rd) {
    static_assert(!WAVE || NT == 64, "a wave-level group is one wave");
    constexpr int NW = NT / 64;
    constexpr int RB = NT == 512 ? 9 : 8;
    const float4* s_ray = L.ray; unsigned short* s_list = L.list; int* s_cnt = L.cnt;
    const int tid = WAVE ? (int)(threadIdx.x & 63) : (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int base = 0; base < S.n_geoms; base += 32) {
        const int n = (S.n_geoms - base) < 32 ? (S.n_geoms - base) : 32;
        const Cand call = analytic_candidates(S, base, n, ro, rd);
        for (int half = 0; half * LP < n; ++half) {
            Cand c = call;
            const unsigned hm = ((1u << LP) - 1u) << (LP * half);
            c.cube &= hm; c.sphere &= hm;
            if (!active) { c.cube = 0u; c.sphere = 0u; }
            const int v = __popc(c.cube) | (__popc(c.sphere) << 16);
            int incl = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
            if (lane == 63) s_cnt[wid] = incl;
            group_sync<WAVE>();
            int woff_c = 0, woff_s = 0, Tc = 0, Ts = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const int cw = s_cnt[w];
                if (w < wid) { woff_c += cw & 0xffff; woff_s += cw >> 16; }
                Tc += cw & 0xffff; Ts += cw >> 16;
            }
            int oc = woff_c + ((incl - v) & 0xffff), os = Tc + woff_s + ((incl - v) >> 16);
            for (unsigned m = c.cube; m; m &= m - 1u) s_list[oc++] = (unsigned short)(tid | (__builtin_ctz(m) << RB));
            for (unsigned m = c.sphere; m; m &= m - 1u) s_list[os++] = (unsigned short)(tid | (__builtin_ctz(m) << RB));
            group_sync<WAVE>();
            const int T = Tc + Ts;
            const int Tc_pad = WAVE ? (Tc + NT - 1) / NT * NT : Tc;      // (cube rounds, then sphere rounds: see analytic_closest_lds)
            for (int r0 = 0; r0 < Tc_pad + Ts; r0 += NT) {
                int pi = r0 + tid;
                if (WAVE) pi = pi < Tc_pad ? (pi < Tc ? pi : T) : Tc + (pi - Tc_pad);
                if (pi < T) {
                    const unsigned e = s_list[pi];
                    const int r = (int)(e & ((1u << RB) - 1u));
                    const int k = (int)(e >> RB);
                    const float4 o4 = s_ray[r], d4 = s_ray[NT + r];
                    const GeomXfRef g = ref_xf(gs.xf, base + k, (c.diag >> k) & 1);
                    const f3 po = mk3(o4.x, o4.y, o4.z), pd = mk3(d4.x, d4.y, d4.z);
                    const float minT = o4.w;
                    f3 tp, tn;
                    const float t = pi < Tc ? box_test(g, po, pd, tp, tn) : sphere_test(g, po, pd, tp, tn);
                    if (t > 0.0f && minT - 1e-5f > t && (double)fabsf(t - minT) > 1e-2) atomicOr(&s_occ[r >> 6], 1ull << (r & 63));
                }
            }
            group_sync<WAVE>();                              // the list is rewritten by the next pass; the last one publishes s_occ
        }
    }
}

#ifndef PTK_INTERSECT_LP
#define PTK_INTERSECT_LP 16      // primitives per pass of k_intersect's pair list (8 KB of LDS; the fused kernel uses 8)
#endif
template <int NT>
__device__ __forceinline__ void analytic_closest_block(const SceneView& S, const GeomStage& gs, bool active, f3 ro, f3 rd, Analytic& a) {
    __shared__ float4 s_ray[2 * NT];
    __shared__ float2 s_res[3 * NT];
    __shared__ unsigned long long s_best[NT];
    __shared__ unsigned short s_list[NT * PTK_INTERSECT_LP];
    __shared__ int s_cnt[NT / 64];
    const TraceLds L = {s_ray, s_res, s_best, s_list, s_cnt};
    analytic_closest_lds<NT, false, PTK_INTERSECT_LP>(S, gs, L, active, ro, rd, a);
}

// Segmented path streams (default render pipeline).  The shading kernel writes the survivors of its block contiguously at
// the START of the block's own 256-slot window of the next buffer ("block-local compaction", no global scan needed), and
// k_segments turns the per-block counts into seg_pref (exclusive prefix) and seg_first (for every 256-path range of the
// dense order, the window that holds its first path).  The next bounce's kernels address paths by DENSE index — the
// reference's compacted order, still the RNG key — and find the physical slot here.  Compared with moving every live
// path once more (k_scatter: 52 B read + 48 B written per path and bounce) this costs five scalar loads per block.
struct Gather {
    const int* pref;      // [windows + 8]: dense index of the first path of window w; entries past the last window = total.  nullptr: dense stream
    const int* first;     // [dense 256-ranges]: window containing dense index 256 * B
    int* src;             // out: physical index of every dense path (read by the kernels that follow the intersect kernel)
};
// Many sparse windows in a row (an open scene after a bounce that killed most paths: 1 057 survivors in 768 windows on mis_test.txt): the
// window of dense index i is the LAST w with pref[w] <= i.  Known: pref[lo] <= i, pref is non-decreasing over [0, nw] and pref[nw] = total > i
// (entries beyond nw + 7 are stale).  Galloping + binary search: ~2 log2(distance) dependent loads; the linear walk this replaces cost one
// dependent load per window (185 of them per thread in that launch: 42 us instead of 22).
__device__ __forceinline__ int sparse_window(const int* __restrict__ pref, int lo, int nw, int i, int& base) {
    int hi = nw, step = 4;
    while (lo + step < hi) {
        const int v = pref[lo + step];
        if (v <= i) { lo += step; base = v; step <<= 1; } else { hi = lo + step; break; }
    }
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1, v = pref[mid];
        if (v <= i) { lo = mid; base = v; } else hi = mid;
    }
    return lo;
}
// nw: number of windows of the stream = ceil(paths the writing launch shaded / 256) — counts[depth - 1] wherever a windowed stream is read
__device__ __forceinline__ int gather_source(const Gather& G, int b, int i, int n, int nw) {
    if (!G.pref) return i;
    const int w0 = G.first[b];                                    // block-uniform: scalar loads
    const int p0 = G.pref[w0], p1 = G.pref[w0 + 1], p2 = G.pref[w0 + 2], p3 = G.pref[w0 + 3], p4 = G.pref[w0 + 4];
    if (i >= n) return 0;
    int k = (i >= p1) + (i >= p2) + (i >= p3) + (i >= p4);
    int base = k == 0 ? p0 : (k == 1 ? p1 : (k == 2 ? p2 : (k == 3 ? p3 : p4)));
    if (k == 4) k = sparse_window(G.pref, w0 + 4, nw, i, base) - w0;
    return (w0 + k) * PTK_BLOCK + (i - base);
}

// ------------------------------------------------------------------------------------------------------------------
// Block-cooperative copy of every primitive's record into dynamic LDS as {inverseTransform rows, transform rows, invTranspose,
// diag, materialid, pad} (dwords [2, 36) and 1 of DevGeom: the matrices stay 16-byte aligned); must be reached by all threads of the
// block.  The launchers size the dynamic LDS with stage_bytes(S).
extern __shared__ float4 ptk_dyn_lds[];
// sync = false: the caller guarantees a workgroup barrier between this copy and the first read of gs.xf
__device__ __forceinline__ GeomStage stage_geoms(const SceneView& S, bool sync = true) {
    GeomStage gs;
    gs.xf = reinterpret_cast<const float*>(ptk_dyn_lds);
    gs.on = S.lds_geoms > 0;
    if (gs.on) {
        float* dst = reinterpret_cast<float*>(ptk_dyn_lds);
        const float* src = reinterpret_cast<const float*>(S.geoms);
        const int total = S.n_geoms * PTK_XF_FLOATS;
        for (int j = threadIdx.x; j < total; j += blockDim.x) {
            const int g = j / PTK_XF_FLOATS, k = j - g * PTK_XF_FLOATS;
            dst[j] = src[(size_t)g * (sizeof(DevGeom) / 4) + (k < 34 ? 2 + k : (k == 34 ? 1 : 0))];
        }
        if (sync) __syncthreads();
    }
    return gs;
}
static inline size_t stage_bytes(const SceneView& S) { return S.lds_geoms > 0 ? (size_t)S.n_geoms * PTK_XF_FLOATS * sizeof(float) : 0; }

// computeIntersections, pathtrace.cu:169-346
//
// Triangle-free scenes: k_intersect<*, false> does everything (analytic primitives only; the empty BVH root cannot hit).
// Scenes with triangles run four kernels by default, because per-ray traversal cost is bimodal (bunny scene: 55-70 % of the
// rays miss the mesh's root box and cost 1 node visit, the rest ~40 with a tail beyond 200; with one ray per lane a wave
// runs for its slowest lane and lane utilisation in the walk is ~15 %):
//   k_intersect<*, true>  analytic primitives for every ray (uniform loop, no divergence) + the ROOT box test; rays that
//                         enter the root box are flagged;
//   k_list                stable compaction of the flagged ray indices into a work list (ballot + block prefix, no atomics);
//   k_traverse            the threaded-BVH walk, one listed ray per lane: every lane now holds a "heavy" ray (utilisation
//                         ~30 %).  Rays are independent, so every per-ray result is the same for any schedule;
//   k_finalize            turns (triangle, u, v, t) winners into full hit records (coalesced) and builds the per-block
//                         aggregates the shading kernel's virtual compaction needs.
// Measured on MI355X (bunny scene, ms per iteration): inline walk 1.34 / 0.79 (1 / 3 iterations in flight), this layout
// 1.43 / 0.72.  Alone on the GPU the walk is bounded by its longest dependent-load chain (~240 visits x ~1100 cycles) with
// ~1 wave per SIMD left, so denser waves do not help; with other iterations' kernels filling the idle SIMDs the machine
// becomes VALU-bound and the denser waves win.  Two dropped variants: appending to the list with one wave-aggregated atomic
// per wave (10 k contended atomics on one word: +48 us per launch), and a persistent-wave walk that refilled idle lanes from a
// global ray counter (too few rays per SIMD for refill to reach a steady state: 2.5 ms).
// ------------------------------------------------------------------------------------------------------------------
#define PTK_VALID_TRI 2          // flag value between k_traverse and k_finalize: "closest hit is a triangle, record pending"

#ifndef PTK_INTERSECT_WAVES
#define PTK_INTERSECT_WAVES 1
#endif
template <bool STATS, bool SPLIT>
__global__ void __launch_bounds__(PTK_BLOCK, PTK_INTERSECT_WAVES) k_intersect(SceneView S, int n_arg, const int* __restrict__ d_n, PtkPaths paths,
                                                          PtkHits hits, int* __restrict__ valid, int* __restrict__ agg,
                                                          unsigned long long* ray_total, unsigned long long* stats,
                                                          int* __restrict__ heavy_flag, int* __restrict__ heavy_agg, Gather G) {
    const int n = d_n ? *d_n : n_arg;
    const int b = blockIdx.x;
    if (b * PTK_BLOCK >= n) return;
    const int i = b * PTK_BLOCK + threadIdx.x;
    if (i == 0 && ray_total) atomicAdd(ray_total, (unsigned long long)n);
    PTK_PHASE_BEGIN();
    const int src = gather_source(G, b, i, n, G.pref ? (d_n[-1] + PTK_BLOCK - 1) / PTK_BLOCK : 0);      // (windowed streams come with d_n = counts + depth, depth > 0)
    PTK_PHASE(0);                                    // 0: live count + window tables
    const GeomStage gs = stage_geoms(S);
    PTK_PHASE(0);                                    // 1: primitive records -> LDS
    int flag = 0;
    bool heavy = false;
    unsigned nv = 0, nt = 0;
    const bool active = i < n;
    f3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 0);
    if (active) {
        const float4 o4 = ld4(paths.o_pix, src), d4 = ld4(paths.d_bnc, src);
        ro = mk3(o4.x, o4.y, o4.z); rd = mk3(d4.x, d4.y, d4.z);
        if (G.src) G.src[i] = src;
    }
    PTK_PHASE(0);                                    // 2: ray loads
    Analytic a;
    if (gs.on) analytic_closest_block<PTK_BLOCK>(S, gs, active, ro, rd, a);          // block-cooperative: every thread takes part
    else if (active) analytic_closest(S, gs, ro, rd, a);
    PTK_PHASE(0);                                    // 3: candidates + pair lists + exact tests
    if (active) {
        Hit h;
        if (SPLIT) {
            flag = a.geom >= 0;
            if (flag) { if (gs.on) analytic_record_staged(gs.xf, a, h); else analytic_record(S, a, h); }
            // root box test of this ray's direction tree (the first iteration of pathtrace.cu:249-256)
            const DevNode root = S.nodes[0];                       // (the same box in every direction copy)
            float tEnter;
            heavy = aabb_test(root, ro, rd, mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z), tEnter) && !(tEnter > a.t_min);
            if (STATS && !heavy) nv = 1;
        } else if (gs.on && S.n_tris == 0) {          // analytic-only scene (the empty BVH root cannot hit): record from the staged tables
            flag = a.geom >= 0;
            if (flag) analytic_record_staged(gs.xf, a, h); else h.t = -1.0f;
            if (STATS) nv = 1;                          // the reference still visits the empty root once
        } else {
            flag = closest_hit_finish<STATS>(S, a, ro, rd, h, nv, nt);
        }
        if (flag) {
            st4(hits.t_n, i, make_float4(h.t, h.n.x, h.n.y, h.n.z));
            st4(hits.p_mat, i, make_float4(h.p.x, h.p.y, h.p.z, __int_as_float(h.mat)));
            if (!S.slim_hits) st4(hits.uv_ids, i, make_float4(h.uv.x, h.uv.y, __int_as_float(h.geom), __int_as_float(h.tri)));
        } else {
            st4(hits.t_n, i, make_float4(-1.f, 0.f, 0.f, 0.f));
        }
        valid[i] = flag;
    }
    PTK_PHASE(0);                                    // 4: hit record + stores
    if (SPLIT) {                                    // flag the rays that entered the root box; k_list compacts them into a work list
        if (i < n) heavy_flag[i] = heavy ? 1 : 0;
        int total;
        (void)block_excl_prefix(heavy ? 1 : 0, total);
        if (threadIdx.x == 0) heavy_agg[b] = total;
    }
    if (agg) {
        int total;
        (void)block_excl_prefix(flag, total);
        if (threadIdx.x == 0) agg[b] = total;
    }
    PTK_PHASE(0);                                    // 5: block aggregates
    PTK_PHASE_END(0);
    if (STATS && stats) {
        unsigned long long a = nv, c = nt;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); c += __shfl_down(c, o, 64); }
        if ((threadIdx.x & 63) == 0) { atomicAdd(&stats[0], a); atomicAdd(&stats[1], c); }
    }
}

// stable compaction of the flagged ray indices into the traversal work list (same scheme as k_scatter, 4-byte payload)
__global__ void __launch_bounds__(PTK_BLOCK) k_list(int n_arg, const int* __restrict__ d_n, const int* __restrict__ heavy_flag,
                                                     const int* __restrict__ heavy_agg, int* __restrict__ list, unsigned* __restrict__ list_count, int whole_windows) {
    int n = d_n ? *d_n : n_arg;
    if (whole_windows) n = (n + PTK_BLOCK - 1) / PTK_BLOCK * PTK_BLOCK;      // fused pipeline: flags live at physical slots, every slot of a window is written
    const int b = blockIdx.x;
    if (b * PTK_BLOCK >= n) { if (n <= 0 && b == 0 && threadIdx.x == 0) *list_count = 0u; return; }
    const int i = b * PTK_BLOCK + threadIdx.x;
    const int flag = (i < n) ? heavy_flag[i] : 0;
    int total;
    const int local = block_excl_prefix(flag, total);
    const int prior = prior_blocks(heavy_agg, b);
    if (flag) list[prior + local] = i;
    if (threadIdx.x == 0 && b == (n - 1) / PTK_BLOCK) *list_count = (unsigned)(prior + total);
}

// (round 4, measured and removed — commits 85deb16 / the one after e1334c2: the same list ordered by PREDICTED TRAVERSAL COST, k_list_sorted.
// Root children hit x 3 + a 32^3 leaf-density table sampled along the ray's chord, counting-sorted inside chunks of 1 024 stream slots, the
// heaviest quarter placed at the front of the list so that the long chains start first.  The walk's busy lane share rose from 0.42 to 0.55
// (quads) / 0.36 to 0.48 (pairs) — PTK_WIDE_STATS=2, profiles/r4_lane_busy_sorted_list.txt, exactly what the CPU model tools/diag/coherence_sim.py
// predicts — and k_traverse_wide went from 75.2 to 70.1 us: its duration is the dependent chain of its slowest ray, not its instruction count.
// With the 13-18 us the sort kernel costs the iteration was 4-10 % SLOWER at 800 x 800 and unchanged at 2048 x 2048 (profiles/r4_ab_sorted_list*.txt).)
// BVH traversal of the listed rays, one per lane (pathtrace.cu:236-279).  A triangle winner is parked as (u, v, t, tri) in
// the ray's uv_ids slot with flag PTK_VALID_TRI; k_finalize expands it (and clears the list counter for the next bounce).
template <bool STATS>
__global__ void __launch_bounds__(PTK_BLOCK) k_traverse(SceneView S, const unsigned* __restrict__ list_count, const int* __restrict__ list,
                                                         PtkPaths paths, PtkHits hits, int* __restrict__ valid, unsigned long long* stats,
                                                         const int* __restrict__ srcidx) {
    const int m = (int)*list_count;
    const int j = blockIdx.x * PTK_BLOCK + threadIdx.x;
    if (blockIdx.x * PTK_BLOCK >= m) return;
    unsigned nv = 0, nt = 0;
    if (j < m) {
        const int ray = list[j];
        const int src = srcidx ? srcidx[ray] : ray;                    // physical slot of dense path `ray` (segmented streams)
        const float4 o4 = ld4(paths.o_pix, src), d4 = ld4(paths.d_bnc, src);
        const float t0 = hits.t_n[4 * (size_t)ray];                   // analytic closest t, -1 = none
        Walk w;
        walk_begin(S, mk3(o4.x, o4.y, o4.z), mk3(d4.x, d4.y, d4.z), t0 > 0.f ? t0 : FLT_MAX, w);
        while (w.idx != -1) {
            const int k = walk_step(S, w);
            if (STATS) { nv++; nt += k; }
        }
        if (w.tri >= 0) {
            st4(hits.uv_ids, ray, make_float4(w.u, w.v, w.t_min, __int_as_float(w.tri)));
            valid[ray] = PTK_VALID_TRI;
        }
    }
    if (STATS && stats) {
        unsigned long long a = nv, c = nt;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); c += __shfl_down(c, o, 64); }
        if ((threadIdx.x & 63) == 0) { atomicAdd(&stats[0], a); atomicAdd(&stats[1], c); }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// k_traverse_wide: the same walk over the 8-wide collapsed form of the same tree (wide_bvh.h has the parity argument:
// only leaf visits have effects, inner tests only prune, so any conservative pruning + the reference's leaf order gives
// bit-identical winners).
//
// SEVERAL LANES PER RAY (template LPR: 4 = a "quad", 16 rays per wave64, described below; 2 = a pair, each lane owns four
// children — fewer instructions per ray (frame, ballots, pushes and leaf code are shared by twice the rays), longer chain.
// Measured on the bunny scene, Grays/s at 1 / 3 iterations in flight: LPR 1: 2.56 / 5.40, 2: 3.08 / 5.76, 4: 3.29 / 5.51, 8: 3.08 /
// 4.63 — all bit-identical; the render loop uses 4 with one iteration at a time and 2 when several keep the machine full).  The work list of a bounce holds ~150 k rays on a
// machine with 524 k lane slots: with one ray per lane the walk runs at ~2 waves per SIMD and its time is the dependent
// chain of the slowest lane (measured: one-lane-per-ray over this same tree 185 us per launch, the threaded walk 125 us).
// A quad shortens the chain and quadruples the waves: lane q of a quad owns children q and q + 4 of the node, so a visit is
// 3 independent 16-B loads per lane (header + two child records, no dependent loads) and two child tests instead of eight.
//   * child test: t = q * (2^e / d) + (p - o) / d — one v_cvt_f32_ubyte + one v_fma per plane — widened by
//     m = 2^-20 (|(p - o) / d| + 255 |2^e / d|), which bounds every rounding difference against the exact leaf test
//     t = (plane - o) * (1 / d) (derivation: DESIGN.md section 5);
//   * two ballots give the quad its 8-bit hit mask; a hit child's position among the hit children in the ray's visiting
//     order is popc(hits & before[dir]) (the node stores, per direction tree, which children precede each child), so all
//     hit children are pushed at once, in reverse visiting order, without a loop;
//   * per-ray stack in LDS; the first PTK_WIDE_TOP nodes of the breadth-first node array (the top two levels) are staged
//     in LDS once per block;
//   * leaves run the reference's exact box test, the `tEnter > t_min` cut and the Moeller-Trumbore test (same functions
//     as the threaded walk), redundantly in the four lanes.
// Rays with a zero / denormal direction component (Bounds3.hpp:174-177 containment tests) and rays whose stack would
// overflow take the threaded walk instead — same result by construction.
// ------------------------------------------------------------------------------------------------------------------
// wide nodes staged in LDS per block (the root and the first nodes of the breadth-first array).  A block serves only 64 rays, so the
// copy is not free: measured on the bunny scene (intersect stage, ms per iteration) 1: 0.882, 8: 0.880, 16: 0.877, 64: 0.883,
// 128: 0.926, 256: 1.150 — the upper levels are served by L1/L2 just as well; two levels (2.3 KB) is what pays.
#ifndef PTK_WIDE_TOP
#define PTK_WIDE_TOP 16
#endif
#define PTK_WIDE_STACK 24         // stack entries per ray
#define PTK_WIDE_STRIDE 25        // odd: the 16 rays of a wave spread over the LDS banks
#define PTK_WIDE_LEAF 0x80000000u

struct WideRay { f3 o, d, inv; float t_min; int tri; float u, v; int dir; };

// conservative test of one child record against the ray (see above); A/Bn/Bf are per visit, per axis
// selN / selF: v_perm_b32 selectors, fixed per ray, that gather the three NEAR (resp. FAR) plane bytes of a child record — the low plane
// of an axis the ray travels up, the high plane otherwise — into bytes 0..2 of one word: two instructions per child instead of six selects
struct WideFrame { float Ax, Ay, Az, Bnx, Bny, Bnz, Bfx, Bfy, Bfz; unsigned selN, selF; };
__device__ __forceinline__ void wide_frame_selectors(bool px, bool py, bool pz, WideFrame& f) {
    // record bytes: c.x = lox, loy, loz, hix; c.y = hiy, hiz, ...  (perm(c.y, c.x, sel): selector values 0-3 take bytes of c.x, 4-7 of c.y, 0x0c = 0)
    f.selN = 0x0c000000u | (px ? 0u : 3u) | ((py ? 1u : 4u) << 8) | ((pz ? 2u : 5u) << 16);
    f.selF = 0x0c000000u | (px ? 3u : 0u) | ((py ? 4u : 1u) << 8) | ((pz ? 5u : 2u) << 16);
}
__device__ __forceinline__ bool wide_child_test(const uint4 c, const WideFrame& f, float t_min) {
    const unsigned nb = __builtin_amdgcn_perm(c.y, c.x, f.selN), fb = __builtin_amdgcn_perm(c.y, c.x, f.selF);
    const float nx = (float)(nb & 0xffu), ny = (float)((nb >> 8) & 0xffu), nz = (float)((nb >> 16) & 0xffu);
    const float fx = (float)(fb & 0xffu), fy = (float)((fb >> 8) & 0xffu), fz = (float)((fb >> 16) & 0xffu);
    const float tn = fmaxf(fmaxf(__builtin_fmaf(nx, f.Ax, f.Bnx), __builtin_fmaf(ny, f.Ay, f.Bny)), __builtin_fmaf(nz, f.Az, f.Bnz));
    const float tf = fminf(fminf(__builtin_fmaf(fx, f.Ax, f.Bfx), __builtin_fmaf(fy, f.Ay, f.Bfy)), __builtin_fmaf(fz, f.Az, f.Bfz));
    return tn <= fminf(tf, t_min) && tf >= 0.f;
}
// the lanes of a pair / quad always run together (one ray, one control flow): values cross them through quad permutes
__device__ __forceinline__ int wide_x1(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true); }      // lanes 1 0 3 2
__device__ __forceinline__ int wide_x2(int v) { return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true); }      // lanes 2 3 0 1
template <int LPR> __device__ __forceinline__ unsigned wide_group_or(unsigned v) {
    v |= (unsigned)wide_x1((int)v);
    if (LPR == 4) v |= (unsigned)wide_x2((int)v);
    return v;
}
__device__ __forceinline__ unsigned wide_before(const uint4 c, int dir) {      // before[dir] of a child record
    const unsigned long long w = ((unsigned long long)c.z << 32) | c.y;
    return (unsigned)(w >> (16 + 8 * dir)) & 0xffu;
}

// exact leaf visit (walk_apply's leaf branch on the leaf's own box): pathtrace.cu:249-267
__device__ __forceinline__ int wide_leaf_visit(const uint4 l0, const uint4 l1, const uint4 l2, const uint4 l3, WideRay& r) {
    DevNode nd;
    nd.minx = __uint_as_float(l0.x); nd.miny = __uint_as_float(l0.y); nd.minz = __uint_as_float(l0.z);
    nd.maxx = __uint_as_float(l0.w); nd.maxy = __uint_as_float(l1.x); nd.maxz = __uint_as_float(l1.y);
    nd.a = 0; nd.b = 0;
    float tEnter;
    if (!aabb_test_fast(nd, r.o, r.inv, tEnter) || tEnter > r.t_min) return 0;
    DevTriE T;
    T.v0[0] = __uint_as_float(l1.z); T.v0[1] = __uint_as_float(l1.w); T.v0[2] = __uint_as_float(l2.x);
    T.e1[0] = __uint_as_float(l2.y); T.e1[1] = __uint_as_float(l2.z); T.e1[2] = __uint_as_float(l2.w);
    T.e2[0] = __uint_as_float(l3.x); T.e2[1] = __uint_as_float(l3.y); T.e2[2] = __uint_as_float(l3.z);
    float tt, u, v;
    if (tri_test(T, r.o, r.d, tt, u, v) && r.t_min > tt) { r.t_min = tt; r.tri = (int)l3.w; r.u = u; r.v = v; }
    return 1;
}

// threads per block of the wide walk: a block holds its LDS (and a new block needs all its wave slots at once) until its slowest ray is done
#ifndef PTK_WIDE_BLOCK
#define PTK_WIDE_BLOCK 256
#endif
// W (round 5): children per node — 8, or the 16-wide variant (PTK_WIDE_WIDTH=16; ptk_device.hpp): 17 x uint4 per node, a hit child's place on the stack from
// the positions of the hit children in the ray's visiting order instead of the before masks; everything else is the same walk.
__device__ __forceinline__ unsigned wide_pos16(const uint4 c, int dir) { return (((c.y >> 16) | (c.z << 16)) >> (4 * dir)) & 15u; }
template <bool STATS, int LPR, int W = 8>       // LPR lanes per ray (4: shortest chain; 2: fewer instructions per ray, see the header comment)
__global__ void __launch_bounds__(PTK_WIDE_BLOCK) k_traverse_wide(SceneView S, const unsigned* __restrict__ list_count, const int* __restrict__ list,
                                                              PtkPaths paths, PtkHits hits, int* __restrict__ valid, unsigned long long* stats,
                                                              const int* __restrict__ srcidx, int stats_mode = 0) {
    static_assert(W == 8 || (W == 16 && (LPR == 2 || LPR == 4)), "16-wide nodes: two or four lanes per ray");
    constexpr int RAYS = PTK_WIDE_BLOCK / LPR, NCH = W / LPR, U4 = W + 1;     // rays per block; children per lane; uint4 per node
    __shared__ uint4 s_top[(PTK_WIDE_TOP > 0 ? PTK_WIDE_TOP : 1) * U4];
    __shared__ unsigned s_stack[RAYS * PTK_WIDE_STRIDE];
    const int m = (int)*list_count;
    if (blockIdx.x * RAYS >= m) return;
    const int n_top = S.n_wnodes < PTK_WIDE_TOP ? S.n_wnodes : PTK_WIDE_TOP;
    if (PTK_WIDE_TOP > 0) {
        for (int k = threadIdx.x; k < n_top * U4; k += PTK_WIDE_BLOCK) s_top[k] = S.wnodes[k];
        __syncthreads();
    }
    const int q = threadIdx.x & (LPR - 1);                       // lane within the ray's group
    const int qsh = (threadIdx.x & 63) & ~(LPR - 1);             // first lane of the group within the wave
    const int j = blockIdx.x * RAYS + threadIdx.x / LPR;
    unsigned nv = 0, nt = 0;
    if (j < m) {
        const int ray = list[j];
        const int src = srcidx ? srcidx[ray] : ray;
        const float4 o4 = ld4(paths.o_pix, src), d4 = ld4(paths.d_bnc, src);
        const float t0 = hits.t_n[4 * (size_t)ray];
        WideRay r;
        r.o = mk3(o4.x, o4.y, o4.z); r.d = mk3(d4.x, d4.y, d4.z);
        r.inv = mk3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
        r.t_min = t0 > 0.f ? t0 : FLT_MAX; r.tri = -1; r.u = 0.f; r.v = 0.f;
        r.dir = mtbvh_dir(r.d);
        const float t_init = r.t_min;
        // the fast slab test needs finite non-zero 1/d (Bounds3.hpp:174-177 otherwise); |d_k| >= 1e-30 keeps 2^e / d finite
        bool threaded = !(fabsf(r.d.x) >= 1e-30f && fabsf(r.d.y) >= 1e-30f && fabsf(r.d.z) >= 1e-30f);
        if (!threaded) {
            unsigned* stk = s_stack + (threadIdx.x / LPR) * PTK_WIDE_STRIDE;
            WideFrame f;
            wide_frame_selectors(r.inv.x > 0.f, r.inv.y > 0.f, r.inv.z > 0.f, f);
            int sp = 0;
            unsigned cur = 0u;                                   // root
            for (;;) {
                if (!(cur & PTK_WIDE_LEAF)) {
                    uint4 q0, ch[NCH];                           // lane q owns children q, q + LPR, ...
                    if (PTK_WIDE_TOP > 0 && (int)cur < n_top) {
                        const uint4* np = s_top + cur * U4; q0 = np[0];
#pragma unroll
                        for (int k = 0; k < NCH; ++k) ch[k] = np[1 + q + LPR * k];
                        asm volatile("" : "+v"(q0.x), "+v"(ch[0].x));
                    } else {
                        const uint4* np = S.wnodes + (size_t)cur * U4; q0 = np[0];
#pragma unroll
                        for (int k = 0; k < NCH; ++k) ch[k] = np[1 + q + LPR * k];
                        asm volatile("" : "+v"(q0.x), "+v"(ch[0].x));
                    }
#ifdef PTK_DIAG_PAD_WALK      // timing experiment only (tools/diag/issue_sensitivity.sh): N more fma per node visit on live registers, same results
                    {
                        float p0 = r.o.x, p1 = r.o.y, p2 = r.d.x, p3 = r.d.y;
#pragma unroll
                        for (int i = 0; i < PTK_DIAG_PAD_WALK / 4; ++i) {
                            p0 = __builtin_fmaf(p0, p1, p2); p1 = __builtin_fmaf(p1, p2, p3); p2 = __builtin_fmaf(p2, p3, p0); p3 = __builtin_fmaf(p3, p0, p1);
                            asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
                        }
                        asm volatile("" :: "v"(p0), "v"(p1), "v"(p2), "v"(p3));
                    }
#endif
                    if (STATS) nv += (q == 0);
                    const float sx = __uint_as_float((q0.w & 0xffu) << 23), sy = __uint_as_float(((q0.w >> 8) & 0xffu) << 23), sz = __uint_as_float(((q0.w >> 16) & 0xffu) << 23);
                    f.Ax = sx * r.inv.x; f.Ay = sy * r.inv.y; f.Az = sz * r.inv.z;                     // exact (power-of-two scale)
                    const float Bx = (__uint_as_float(q0.x) - r.o.x) * r.inv.x, By = (__uint_as_float(q0.y) - r.o.y) * r.inv.y, Bz = (__uint_as_float(q0.z) - r.o.z) * r.inv.z;
                    const float K = 9.5367431640625e-07f;                                              // 2^-20
                    // m = 2^-20 (|B| + 255 |A|) + 1e-36, twice the rounding bound: two fused steps per axis (any rounding of m itself is
                    // far inside that factor of two; the walk only prunes, so both library flavours find the same hits)
                    const float K255 = 255.f * K;
                    const float mx = __builtin_fmaf(K255, fabsf(f.Ax), __builtin_fmaf(K, fabsf(Bx), 1e-36f)), my = __builtin_fmaf(K255, fabsf(f.Ay), __builtin_fmaf(K, fabsf(By), 1e-36f)),
                                mz = __builtin_fmaf(K255, fabsf(f.Az), __builtin_fmaf(K, fabsf(Bz), 1e-36f));
                    f.Bnx = Bx - mx; f.Bfx = Bx + mx; f.Bny = By - my; f.Bfy = By + my; f.Bnz = Bz - mz; f.Bfz = Bz + mz;
                    const int nc = (int)(q0.w >> 24);
                    bool hit[NCH];
                    unsigned pos[NCH];                            // (W = 16) the child's position in the ray's visiting order
                    unsigned H = 0u;                              // hit children of this ray (bit = child index; W = 16: bit = position in the visiting order)
#pragma unroll
                    for (int k = 0; k < NCH; ++k) {
                        hit[k] = (q + LPR * k < nc) & wide_child_test(ch[k], f, r.t_min);      // no short circuit: straight-line code
                        if (W == 16) { pos[k] = wide_pos16(ch[k], r.dir); H |= hit[k] ? (1u << pos[k]) : 0u; }
                        else if (LPR == 2 || LPR == 4) H |= hit[k] ? (1u << (q + LPR * k)) : 0u;    // own bits; the group's lanes are ORed below
                        else {
                            const unsigned long long bal = __ballot(hit[k]);
                            H |= ((unsigned)(bal >> qsh) & ((1u << LPR) - 1u)) << (LPR * k);
                        }
                    }
                    if (LPR == 2 || LPR == 4) H = wide_group_or<LPR>(H);
                    // (round 4, measured and removed — commit "leading leaf children tested at the node visit": the hit children that come first in the
                    // ray's order and are leaves tested at once, each by the lane that owns it, the rest culled again with the new t_min.  Bit-exact;
                    // the CPU model (tools/diag/wide_sim.cpp: wide_sim_inline) counts 7.4 instead of 9.7 loop iterations per listed ray and 45-50 instead
                    // of 70-90 for the slowest ray — but k_traverse_wide went from 75.2 to 78.1 us and the bunny scene from 5.85 to 5.34 Grays/s
                    // (profiles/r4_ab_inline_leaves.txt): a node visit with leaves is then TWO dependent fetches (node, then its leaves), so the chain of
                    // dependent round trips of a ray shrinks by ~11 %, not by the iteration count, and the walk pays 78 instead of 64 VGPRs for it.)
                    // (round 4, measured and removed as well — LEAF CLUSTERS: references to all-leaf nodes replaced by "n consecutive leaf records", all
                    // tested at once by the group's lanes without the node fetch, the ordered replay kept for the rare visit with two candidates.
                    // Bit-exact; model chain 9.7 -> 7.7 round trips per ray; one iteration in flight +3 %, three in flight -5.5 %
                    // (profiles/r4_ab_leaf_clusters.txt): n x 64 B and n exact tests per visit instead of 144 B + the ~1.5 leaves the ray meets.)
                    // (round 4, measured and removed — CAPPED TWO-STAGE WALK: every ray walked for at most 12 loop trips, the unfinished ones parked with hit and
                    // stack in 128-B records and resumed by a second launch, packed into waves of their own.  Bit-exact; model wave-iterations x 0.68; bunny
                    // 5.98 -> 5.84 Grays/s, 2048^2 6.56 -> 6.83 (profiles/r4_ab_capped_walk.txt): idle lanes issue no node fetches, the busy lanes' fetches are
                    // the walk's throughput, and no packing changes their number.)
                    const int nh = __popc(H);
                    if (sp + nh > PTK_WIDE_STACK) { threaded = true; break; }
                    // all hit children at once, in reverse visiting order: the child visited first ends on top
#pragma unroll
                    for (int k = 0; k < NCH; ++k)
                        if (hit[k]) stk[sp + nh - 1 - __popc(H & (W == 16 ? (1u << pos[k]) - 1u : wide_before(ch[k], r.dir)))] = ch[k].w;
                    sp += nh;
                } else {
                    const uint4* lp = S.wleaves + (size_t)(cur & ~PTK_WIDE_LEAF) * 4;
                    const uint4 l0 = lp[0], l1 = lp[1], l2 = lp[2], l3 = lp[3];
#ifdef PTK_DIAG_DOUBLE_LEAF      // timing experiments only: the leaf visit runs twice on laundered inputs (same results)
                    { WideRay r2 = r; asm volatile("" : "+v"(r2.o.x), "+v"(r2.d.x), "+v"(r2.inv.x)); (void)wide_leaf_visit(l0, l1, l2, l3, r2);
                      asm volatile("" :: "v"(r2.t_min), "v"(r2.tri), "v"(r2.u), "v"(r2.v)); }
#endif
                    (void)wide_leaf_visit(l0, l1, l2, l3, r);
                    if (STATS) nt += (q == 0);
                }
                if (sp == 0) break;
                --sp; cur = stk[sp];
            }
        }
        if (threaded) {                                          // the reference's own walk (also the overflow fallback)
            Walk w;
            walk_begin(S, r.o, r.d, t_init, w);
            while (w.idx != -1) {
                const int k = walk_step(S, w);
                if (STATS && q == 0) { nv++; nt += k; }
            }
            r.tri = w.tri; r.u = w.u; r.v = w.v; r.t_min = w.t_min;
        }
        if (q == 0 && r.tri >= 0) {
            st4(hits.uv_ids, ray, make_float4(r.u, r.v, r.t_min, __int_as_float(r.tri)));
            valid[ray] = PTK_VALID_TRI;
        }
    }
    if (STATS && stats) {
        unsigned long long a = nv, c = nt;
        if (stats_mode == 2) {
            // PTK_WIDE_STATS=2 (lane-busy diagnostic): stats[0] += the visits of the wave's slowest ray (what the wave runs for), stats[1] += the
            // visits of all its rays; busy lane share = stats[1] / (stats[0] x rays per wave)
            unsigned trips = nv + nt;                            // (counted by lane 0 of every group)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const unsigned t = __shfl_xor(trips, o, 64); trips = t > trips ? t : trips; }
            c = a + c; a = 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
            a = trips;
            if ((threadIdx.x & 63) == 0) { atomicAdd(&stats[0], a); atomicAdd(&stats[1], c); }
            return;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); c += __shfl_down(c, o, 64); }
        if ((threadIdx.x & 63) == 0) { atomicAdd(&stats[0], a); atomicAdd(&stats[1], c); }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// k_traverse_pool (round 5): the same 8-wide walk, with the rays of a block POOLED.  k_traverse_wide gives every group of LPR lanes one ray and a
// wave runs until its slowest ray is done: with 32 rays per wave the wave-iterations are 2.8 x the lane work (busy lane share 0.36; mean 9 loop trips
// per ray, 99th percentile 31, slowest 80-90 — tools/diag/wide_tree_stats.py) and, with several iterations in flight, vector-ALU issue is what the
// walk costs.  Here a block of 256 threads stages a chunk of `chunk` consecutive work-list rays in LDS (one coalesced pass; a refill is an LDS read,
// not a chain of three dependent global loads) and its groups take the next ray from a block counter whenever a quarter of a wave's lanes have
// gone idle (all idle groups of the wave at once: the refill code runs once per ~2.5 loop trips, not in every trip for one or two groups — the
// persistent-threads scheme of Aila & Laine, "Understanding the efficiency of ray traversal on GPUs", HPG 2009, at group granularity).  CPU model:
// 0.58 (chunk 512) / 0.51 (1 024) of k_traverse_wide's wave-iterations.  Every ray runs exactly the loop of k_traverse_wide — same node / leaf
// visits in the same order, same fallbacks — so the results are bit-identical by construction; only which lanes walk which ray changes.
// LDS: rays 32 B x chunk + stacks 12.8 KB + staged top nodes 2.3 KB (chunk 512: 31 KB, five blocks per CU).
// ------------------------------------------------------------------------------------------------------------------
#define PTK_POOL_IDLE_LANES 16    // a wave refills when at least this many of its 64 lanes are idle
// S16 (second session of round 5): the per-ray stacks hold 16-bit child references (15 bits of index + the leaf flag) when the tree has fewer than 32 768 nodes
// and leaves — 6.4 instead of 12.8 KB of LDS per block with pairs, so that a pool of 256 rays fits eight blocks per CU (16.9 KB) like k_traverse_wide does, and
// the kernel is compiled for eight waves per SIMD.  Same references, same order: bit-identical.
__device__ __forceinline__ unsigned short pool_pack16(unsigned w) { return (unsigned short)((w & 0x7fffu) | ((w >> 16) & 0x8000u)); }
__device__ __forceinline__ unsigned pool_unpack16(unsigned h) { return (h & 0x7fffu) | ((h & 0x8000u) << 16); }
template <bool S16> struct PoolStack { typedef unsigned T; };
template <> struct PoolStack<true> { typedef unsigned short T; };
template <int LPR, bool S16>
__global__ void __launch_bounds__(PTK_WIDE_BLOCK, S16 ? 8 : 1) k_traverse_pool(SceneView S, const unsigned* __restrict__ list_count, const int* __restrict__ list,
                                                              PtkPaths paths, PtkHits hits, int* __restrict__ valid, const int* __restrict__ srcidx, int chunk) {
    constexpr int GROUPS = PTK_WIDE_BLOCK / LPR, NCH = 8 / LPR;
    typedef typename PoolStack<S16>::T StackT;
    __shared__ uint4 s_top[(PTK_WIDE_TOP > 0 ? PTK_WIDE_TOP : 1) * 9];
    __shared__ StackT s_stack[GROUPS * PTK_WIDE_STRIDE];
    __shared__ int s_next;
    extern __shared__ float4 s_rays[];                           // [chunk][2]: (origin, t0), (direction, ray index)
    const int m = (int)*list_count;
    const int j0 = blockIdx.x * chunk;
    if (j0 >= m) return;
    const int cnt = m - j0 < chunk ? m - j0 : chunk;
    const int n_top = S.n_wnodes < PTK_WIDE_TOP ? S.n_wnodes : PTK_WIDE_TOP;
    for (int k = threadIdx.x; k < n_top * 9; k += PTK_WIDE_BLOCK) s_top[k] = S.wnodes[k];
    for (int k = threadIdx.x; k < cnt; k += PTK_WIDE_BLOCK) {
        const int ray = list[j0 + k];
        const int src = srcidx ? srcidx[ray] : ray;
        const float4 o4 = ld4(paths.o_pix, src), d4 = ld4(paths.d_bnc, src);
        s_rays[2 * k] = make_float4(o4.x, o4.y, o4.z, hits.t_n[4 * (size_t)ray]);
        s_rays[2 * k + 1] = make_float4(d4.x, d4.y, d4.z, __int_as_float(ray));
    }
    if (threadIdx.x == 0) s_next = 0;
    __syncthreads();
    const int q = threadIdx.x & (LPR - 1);
    const int leader = (threadIdx.x & 63) & ~(LPR - 1);            // first lane of the group within the wave
    StackT* stk = s_stack + (threadIdx.x / LPR) * PTK_WIDE_STRIDE;
    WideRay r; WideFrame f;
    r.o = mk3(0, 0, 0); r.d = mk3(0, 0, 0); r.inv = mk3(0, 0, 0); r.t_min = 0.f; r.tri = -1; r.u = 0.f; r.v = 0.f; r.dir = 0; f.selN = f.selF = 0u;
    int ray = -1, sp = 0;
    float t_init = 0.f;
    unsigned cur = 0u;
    bool active = false, pool_empty = false;                     // pool_empty is wave-uniform
    for (;;) {
        // ---- refill: a wave-uniform decision; all idle groups of the wave draw at once (the first pass hands every group its first ray)
        const int n_idle = 64 - __popcll(__ballot(active));
        if (n_idle >= PTK_POOL_IDLE_LANES && !pool_empty) {
            if (!active) {
                int j = 0;
                if (q == 0) j = atomicAdd(&s_next, 1);
                j = __shfl(j, leader, 64);
                if (j < cnt) {
                    const float4 a = s_rays[2 * j], b = s_rays[2 * j + 1];
                    r.o = mk3(a.x, a.y, a.z); r.d = mk3(b.x, b.y, b.z); ray = __float_as_int(b.w);
                    r.inv = mk3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
                    r.t_min = a.w > 0.f ? a.w : FLT_MAX; r.tri = -1; r.u = 0.f; r.v = 0.f;
                    r.dir = mtbvh_dir(r.d);
                    t_init = r.t_min;
                    wide_frame_selectors(r.inv.x > 0.f, r.inv.y > 0.f, r.inv.z > 0.f, f);
                    sp = 0; active = true;
                    // the fast slab test needs finite non-zero 1/d (Bounds3.hpp:174-177 otherwise): such rays take the reference's own walk at once
                    cur = (fabsf(r.d.x) >= 1e-30f && fabsf(r.d.y) >= 1e-30f && fabsf(r.d.z) >= 1e-30f) ? 0u : 0xffffffffu;
                }
            }
            pool_empty = __any(!active);                         // a group drew past the end of the chunk (s_next only grows)
        }
        if (__ballot(active) == 0ull) break;                     // every group idle: the pool is empty (otherwise the refill above made one busy)
        if (active) {
            // ---- one loop trip of k_traverse_wide for this group's ray
            bool finished = false, overflow = false;
            if (cur == 0xffffffffu) {                            // the reference's own walk (zero direction component, or the stack would overflow)
                Walk w;
                walk_begin(S, r.o, r.d, t_init, w);
                while (w.idx != -1) (void)walk_step(S, w);
                r.tri = w.tri; r.u = w.u; r.v = w.v; r.t_min = w.t_min;
                finished = true;
            } else if (!(cur & PTK_WIDE_LEAF)) {
                uint4 q0, ch[NCH];
                if (PTK_WIDE_TOP > 0 && (int)cur < n_top) {
                    const uint4* np = s_top + cur * 9; q0 = np[0];
#pragma unroll
                    for (int k = 0; k < NCH; ++k) ch[k] = np[1 + q + LPR * k];
                    asm volatile("" : "+v"(q0.x), "+v"(ch[0].x));
                } else {
                    const uint4* np = S.wnodes + (size_t)cur * 9; q0 = np[0];
#pragma unroll
                    for (int k = 0; k < NCH; ++k) ch[k] = np[1 + q + LPR * k];
                    asm volatile("" : "+v"(q0.x), "+v"(ch[0].x));
                }
                const float sx = __uint_as_float((q0.w & 0xffu) << 23), sy = __uint_as_float(((q0.w >> 8) & 0xffu) << 23), sz = __uint_as_float(((q0.w >> 16) & 0xffu) << 23);
                f.Ax = sx * r.inv.x; f.Ay = sy * r.inv.y; f.Az = sz * r.inv.z;
                const float Bx = (__uint_as_float(q0.x) - r.o.x) * r.inv.x, By = (__uint_as_float(q0.y) - r.o.y) * r.inv.y, Bz = (__uint_as_float(q0.z) - r.o.z) * r.inv.z;
                const float K = 9.5367431640625e-07f, K255 = 255.f * K;            // same margins as k_traverse_wide
                const float mx = __builtin_fmaf(K255, fabsf(f.Ax), __builtin_fmaf(K, fabsf(Bx), 1e-36f)), my = __builtin_fmaf(K255, fabsf(f.Ay), __builtin_fmaf(K, fabsf(By), 1e-36f)),
                            mz = __builtin_fmaf(K255, fabsf(f.Az), __builtin_fmaf(K, fabsf(Bz), 1e-36f));
                f.Bnx = Bx - mx; f.Bfx = Bx + mx; f.Bny = By - my; f.Bfy = By + my; f.Bnz = Bz - mz; f.Bfz = Bz + mz;
                const int nc = (int)(q0.w >> 24);
                bool hit[NCH];
                unsigned H = 0u;
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    hit[k] = (q + LPR * k < nc) & wide_child_test(ch[k], f, r.t_min);
                    H |= hit[k] ? (1u << (q + LPR * k)) : 0u;
                }
                H = wide_group_or<LPR>(H);
                const int nh = __popc(H);
                overflow = sp + nh > PTK_WIDE_STACK;             // (the threaded walk starts over from t_init: same result by construction)
                if (!overflow) {
#pragma unroll
                    for (int k = 0; k < NCH; ++k)
                        if (hit[k]) stk[sp + nh - 1 - __popc(H & wide_before(ch[k], r.dir))] = S16 ? (StackT)pool_pack16(ch[k].w) : (StackT)ch[k].w;
                    sp += nh;
                    finished = sp == 0;
                }
            } else {
                const uint4* lp = S.wleaves + (size_t)(cur & ~PTK_WIDE_LEAF) * 4;
                const uint4 l0 = lp[0], l1 = lp[1], l2 = lp[2], l3 = lp[3];
                (void)wide_leaf_visit(l0, l1, l2, l3, r);
                finished = sp == 0;
            }
            if (overflow) cur = 0xffffffffu;
            else if (finished) {
                if (q == 0 && r.tri >= 0) {
                    st4(hits.uv_ids, ray, make_float4(r.u, r.v, r.t_min, __int_as_float(r.tri)));
                    valid[ray] = PTK_VALID_TRI;
                }
                active = false;
            } else { --sp; cur = S16 ? pool_unpack16(stk[sp]) : (unsigned)stk[sp]; }
        }
    }
}

__global__ void __launch_bounds__(PTK_BLOCK) k_finalize(SceneView S, int n_arg, const int* __restrict__ d_n, PtkHits hits,
                                                         int* __restrict__ valid, int* __restrict__ agg) {
    const int n = d_n ? *d_n : n_arg;
    const int b = blockIdx.x;
    if (b * PTK_BLOCK >= n) return;
    const int i = b * PTK_BLOCK + threadIdx.x;
    int flag = 0;
    if (i < n) {
        flag = valid[i];
        if (flag == PTK_VALID_TRI) {
            const float4 r = ld4(hits.uv_ids, i);
            Hit h;
            triangle_record(S, __float_as_int(r.w), r.x, r.y, r.z, h);
            st4(hits.t_n, i, make_float4(h.t, h.n.x, h.n.y, h.n.z));
            st4(hits.p_mat, i, make_float4(h.p.x, h.p.y, h.p.z, __int_as_float(h.mat)));
            if (!S.slim_hits) st4(hits.uv_ids, i, make_float4(h.uv.x, h.uv.y, __int_as_float(h.geom), __int_as_float(h.tri)));
            valid[i] = 1;
            flag = 1;
        }
    }
    if (agg) {
        int total;
        (void)block_excl_prefix(flag, total);
        if (threadIdx.x == 0) agg[b] = total;
    }
}

// k_finalize of the fused pipeline: slots are physical, every slot of the windows [0, ceil(n / 256)) carries valid = -2 (empty) / 0 / 1 /
// PTK_VALID_TRI; writes the hit rank of every slot (hr: >= 0 rank among the hits of its window, -1 miss, -2 empty) and the window's
// hit count, from which its last block builds hit_pref (a triangle hit changes the ranks the shading keys are made of).
__global__ void __launch_bounds__(PTK_BLOCK) k_finalize2(SceneView S, int n_arg, const int* __restrict__ d_n, PtkHits hits, const int* __restrict__ valid,
                                                          int* __restrict__ hr, int* __restrict__ agg_h, int* __restrict__ done, int* __restrict__ hit_pref,
                                                          const int* __restrict__ cum_prev, int* __restrict__ cum_out, int cum_stride, int ring_div, int cum_d,
                                                          int iter_arg, const int* __restrict__ iter_ptr) {
    const int n = d_n ? *d_n : n_arg;
    const int b = blockIdx.x;
    // tiles in flight (cum_out != null): the hits of the traced bounce cum_d in this band and the bands above, final only here
    const int cum_at = cum_out ? cum_row(iter_ptr ? *iter_ptr : iter_arg, ring_div, cum_stride) + cum_d : 0;
    if (b * PTK_BLOCK >= n) {
        if (n <= 0 && b == 0 && threadIdx.x < 8) hit_pref[threadIdx.x] = 0;
        if (n <= 0 && b == 0 && threadIdx.x == 0 && cum_out) cum_out[cum_at] = cum_read(cum_prev, 0, cum_at);
        return;
    }
    const int i = b * PTK_BLOCK + threadIdx.x;
    const int flag = valid[i];
    int mat = 0;
    if (flag == PTK_VALID_TRI) {
        const float4 r = ld4(hits.uv_ids, i);
        Hit h;
        triangle_record(S, __float_as_int(r.w), r.x, r.y, r.z, h);
        st4(hits.t_n, i, make_float4(h.t, h.n.x, h.n.y, h.n.z));
        st4(hits.p_mat, i, make_float4(h.p.x, h.p.y, h.p.z, __int_as_float(h.mat)));
        if (!S.slim_hits) st4(hits.uv_ids, i, make_float4(h.uv.x, h.uv.y, __int_as_float(h.geom), __int_as_float(h.tri)));
        mat = h.mat;
    } else if (flag == 1) mat = __float_as_int(hits.p_mat[4 * (size_t)i + 3]);      // analytic winner: record written by k_shade_trace
    int total;
    const int rank = block_excl_prefix(flag >= 1 ? 1 : 0, total);
    hr[i] = flag >= 1 ? (rank | (S.mats[mat].type << 8)) : (flag == 0 ? (int)0x80000000u : (int)0xC0000000u);      // rank | material type << 8: the next launch's head reads one word; top bits 10 = miss, 11 = empty slot
    // the last block to get here builds hit_pref (two-level done counters as in segments_tail; they are back at zero after every launch)
    __shared__ int s_last;
    const int nb = (n + PTK_BLOCK - 1) / PTK_BLOCK, ng = (nb + PTK_SEG_GROUP - 1) / PTK_SEG_GROUP;
    if (threadIdx.x == 0) {
        __hip_atomic_store(agg_h + b, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int grp = b / PTK_SEG_GROUP;
        const int gsize = min(PTK_SEG_GROUP, nb - grp * PTK_SEG_GROUP);
        int last = 0;
        if (__hip_atomic_fetch_add(done + 32 * (grp + 1), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1)
            last = __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ng - 1;
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    for (int k = threadIdx.x; k <= ng; k += PTK_BLOCK) __hip_atomic_store(done + 32 * k, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int total_h = prefix_block<PTK_BLOCK, true>(nb, agg_h, hit_pref);
    if (threadIdx.x == 0 && cum_out) cum_out[cum_at] = cum_read(cum_prev, 0, cum_at) + total_h;
}
// environment map on a miss, fused pipeline: slot g of the traced stream missed iff the top bits of hr[g] are 10 (00: hit, 11: empty slot)
__global__ void __launch_bounds__(PTK_BLOCK) k_env_miss2(SceneView S, int n_arg, const int* __restrict__ d_n, PtkPaths paths, const int* __restrict__ hr,
                                                          FilmArgs film, int depth) {
    const int n = d_n ? *d_n : n_arg;
    const int g = blockIdx.x * PTK_BLOCK + threadIdx.x;
    if (blockIdx.x * PTK_BLOCK >= n || ((unsigned)hr[g] >> 30) != 2u) return;      // status bits 10 = a traced ray that missed
    const float4 d4 = ld4(paths.d_bnc, g), c4 = ld4(paths.col_pdf, g);
    const int pix = __float_as_int(paths.o_pix[4 * (size_t)g + 3]);
    const f3 c = nan_to_zero(mk3(c4.x, c4.y, c4.z) * env_radiance(S, mk3(d4.x, d4.y, d4.z)));
    film_add(film, depth, pix, c);
}

// environment map on a miss (pathtrace.cu:303-311): img[pixel] += processNAN(color * env(sphere2Plane(direction)))
__global__ void __launch_bounds__(PTK_BLOCK) k_env_miss(SceneView S, int n_arg, const int* __restrict__ d_n, PtkPaths paths,
                                                         const int* __restrict__ valid, FilmArgs film, int depth,
                                                         const int* __restrict__ srcidx) {
    const int n = d_n ? *d_n : n_arg;
    const int g = blockIdx.x * PTK_BLOCK + threadIdx.x;
    if (g >= n || valid[g] != 0) return;
    const int i = srcidx ? srcidx[g] : g;
    const float4 d4 = ld4(paths.d_bnc, i), c4 = ld4(paths.col_pdf, i);
    const int pix = __float_as_int(paths.o_pix[4 * (size_t)i + 3]);
    const f3 c = nan_to_zero(mk3(c4.x, c4.y, c4.z) * env_radiance(S, mk3(d4.x, d4.y, d4.z)));
    film_add(film, depth, pix, c);
}

// ------------------------------------------------------------------------------------------------------------------
// shading: PTkernel (405-498), DirectLiPTkernel (348-403), MisPTkernel (500-588) for one path.
// `slot` is the compacted index of the path == the RNG key of the reference.  Returns rayValid.
// ------------------------------------------------------------------------------------------------------------------
struct PathRegs { f3 o, d, col; int pix, bounces; float prevPdf; };

struct Film { FilmArgs a; int depth; };       // the radiance planes and the bounce whose adds this launch makes
__device__ __forceinline__ void add_radiance(const Film& film, int pix, f3 c) {
    film_add(film.a, film.depth, pix, c);
}

// Material records staged in LDS by the shading kernel (first PTK_LDS_MATS materials; the rest, if any, come from global memory):
// shading otherwise starts with two dependent global loads per path (hit -> material id -> material).
#define PTK_LDS_MATS 64
struct MatTable { const DevMaterial* lds; int n_lds; };
__device__ __forceinline__ DevMaterial fetch_material(const SceneView& S, const MatTable& mt, int id) {
    if (id < mt.n_lds) return mt.lds[id];
    return S.mats[id];
}
__device__ __forceinline__ int fetch_material_type(const SceneView& S, const MatTable& mt, int id) {
    if (id < mt.n_lds) return mt.lds[id].type;
    return S.mats[id].type;
}
template <int NT, int MAXM = PTK_LDS_MATS>
__device__ __forceinline__ MatTable stage_materials(const SceneView& S, DevMaterial* lds) {      // all threads of the block; ends with a barrier
    const int n = S.n_mats < MAXM ? S.n_mats : MAXM;
    uint32_t* dst = reinterpret_cast<uint32_t*>(lds);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(S.mats);
    for (int j = threadIdx.x; j < n * (int)(sizeof(DevMaterial) / 4); j += NT) dst[j] = src[j];
    __syncthreads();
    MatTable mt; mt.lds = lds; mt.n_lds = n;
    return mt;
}

// rq != nullptr (fused kernel, MIS / DIRECT): the next-event shadow ray is NOT tested here — the sample is evaluated as if it were free, its
// contribution goes to rq->contrib instead of the film, and the caller adds it after the block's shadow phase if the ray is unoccluded.  Same
// values, same single read-modify-write of the pixel within the launch: bit-identical to the inline form.
template <int MODE>
__device__ __forceinline__ int shade_path(const SceneView& S, const GeomStage& gs, const MatTable& mt, int iter, int slot, int depth, const Hit& h, PathRegs& p, const Film& img,
                                          ShadowReq* rq = nullptr) {
    if (h.t <= 0.f) { p.col = p.col * 0.f; p.bounces = 0; return 0; }
    const DevMaterial mrec = fetch_material(S, mt, h.mat);
    const MatEval m = eval_material(S, mrec, h.uv);
    Rng rng = make_rng(iter, slot, depth);
    if (MODE == PTK_MODE_DIRECTLI) {
        if (m.type == PTK_MAT_LIGHT) {
            p.col = p.col * m.albedo_member;
            p.bounces = 0;
            add_radiance(img, p.pix, nan_to_zero(p.col));
            return 0;
        }
        LightRec li;
        light_sample(S, gs, S.mats, h.p, rng, li, rq);
        if (li.pdf <= 0) { p.col = p.col * 0.f; if (rq) rq->on = false; return 0; }
        const f3 wi = normalize(li.pos - h.p);
        const f3 f = mat_bsdf(m, h.n, p.d, wi);
        p.col = p.col * (f * li.emit * gmax(dot(wi, h.n), 0.f) / li.pdf);
        if (rq) rq->contrib = nan_to_zero(p.col);
        else add_radiance(img, p.pix, nan_to_zero(p.col));
        return 0;
    }
    Scatter s;
#ifdef PTK_DIAG_DOUBLE_SCATTER   // timing experiments only
    { Rng rng2 = rng; f3 n2 = h.n; asm volatile("" : "+v"(n2.x), "+v"(n2.y), "+v"(n2.z)); Scatter s2; scatter_sample(m, n2, p.d, s2, rng2);
      asm volatile("" :: "v"(s2.pdf), "v"(s2.dir.x), "v"(s2.dir.y), "v"(s2.dir.z), "v"(s2.bsdf.x), "v"(s2.bsdf.y), "v"(s2.bsdf.z)); }
#endif
    scatter_sample(m, h.n, p.d, s, rng);
    if (s.pdf == 0) { p.col = p.col * 0.f; p.bounces = 0; return 0; }
    if (m.type == PTK_MAT_LIGHT) {
        float weight = 1.f;
        if (MODE == PTK_MODE_MIS && p.prevPdf > 0) {
            const float lp = light_pdf(S, p.o, h.p, h.n, h.tri, h.geom);
            weight = power_heuristic(p.prevPdf, lp);
        }
        if (MODE == PTK_MODE_MIS) p.col = p.col * ((s.bsdf / s.pdf) * weight);
        else p.col = p.col * (s.bsdf / s.pdf);
        p.bounces = 0;
        add_radiance(img, p.pix, nan_to_zero(p.col));
        return 0;
    }
    const bool isDelta = (m.type == PTK_MAT_DIELECTRIC);
    if (MODE == PTK_MODE_MIS && !isDelta) {
        LightRec li;
        light_sample(S, gs, S.mats, h.p, rng, li, rq);                    // rng by value: the path's stream does not advance
        if (!li.is_zero) {
            const f3 wi = normalize(li.pos - h.p);
            const float bp = mat_pdf(m, h.n, p.d, wi);
            const float lp = li.pdf;
            const f3 f = mat_bsdf(m, h.n, p.d, wi);
            const float w = power_heuristic(lp, bp);
            const f3 c = nan_to_zero(w * p.col * li.emit * f * gmax(dot(wi, h.n), 0.f) / lp);
            if (rq) rq->contrib = c; else add_radiance(img, p.pix, c);
        }
    }
    const f3 offsetDir = dot(s.dir, h.n) > 0 ? h.n : -h.n;
    p.d = s.dir;
    p.o = h.p + (isDelta ? 1e-3f * offsetDir : 1e-4f * s.dir);
    p.col = p.col * div3_exact(s.bsdf * fabsf(dot(s.dir, h.n)), s.pdf);
    if (MODE == PTK_MODE_MIS) p.prevPdf = isDelta ? -1.f : s.pdf;
    int alive = 1;
    if (--p.bounces == 0) { p.col = mk3(0.f, 0.f, 0.f); alive = 0; }
    return alive;
}

// Thread index = compacted slot: the reference's launch shape.  The kernel behind the kernel-level ptk_shade entry point (the render loops shade
// with k_shade_trace / k_shade_binned).
template <int MODE>
__global__ void __launch_bounds__(PTK_BLOCK) k_shade(SceneView S, int iter_arg, int depth, int n_arg, const int* __restrict__ d_n,
                                                      PtkHits hits, PtkPaths paths, int* __restrict__ valid_b, float* __restrict__ img_base) {
    const int n = d_n ? *d_n : n_arg;
    const Film img = {FilmArgs{img_base, nullptr, 0u, 0u, 0u, 0u}, 0};
    const int b = blockIdx.x;
    if (b * PTK_BLOCK >= n) return;
    GeomStage gs; gs.xf = nullptr; gs.on = false;
    if (MODE != PTK_MODE_BSDF) gs = stage_geoms(S);          // only the light-sampling modes trace shadow rays
    const int i = b * PTK_BLOCK + threadIdx.x;
    int alive = 0;
    if (i < n) {
        const float4 h0 = ld4(hits.t_n, i), h1 = ld4(hits.p_mat, i);
        const float4 h2 = S.slim_hits ? make_float4(0.f, 0.f, __int_as_float(-1), __int_as_float(-1)) : ld4(hits.uv_ids, i);
        const float4 o4 = ld4(paths.o_pix, i), d4 = ld4(paths.d_bnc, i), c4 = ld4(paths.col_pdf, i);
        Hit h;
        h.t = h0.x; h.n = mk3(h0.y, h0.z, h0.w); h.p = mk3(h1.x, h1.y, h1.z); h.mat = __float_as_int(h1.w);
        h.uv.x = h2.x; h.uv.y = h2.y; h.geom = __float_as_int(h2.z); h.tri = __float_as_int(h2.w);
        PathRegs p;
        p.o = mk3(o4.x, o4.y, o4.z); p.pix = __float_as_int(o4.w);
        p.d = mk3(d4.x, d4.y, d4.z); p.bounces = __float_as_int(d4.w);
        p.col = mk3(c4.x, c4.y, c4.z); p.prevPdf = c4.w;
        const MatTable mt = {nullptr, 0};
        alive = shade_path<MODE>(S, gs, mt, iter_arg, i, depth, h, p, img);
        st4(paths.o_pix, i, make_float4(p.o.x, p.o.y, p.o.z, __int_as_float(p.pix)));
        st4(paths.d_bnc, i, make_float4(p.d.x, p.d.y, p.d.z, __int_as_float(p.bounces)));
        st4(paths.col_pdf, i, make_float4(p.col.x, p.col.y, p.col.z, p.prevPdf));
        valid_b[i] = alive;
    }
}

// Segmented streams, bookkeeping between two bounces: exclusive prefix of the per-window survivor counts (seg_pref), the
// window that holds the first path of every dense 256-range (seg_first), the live count of the next bounce, and (last bounce
// of an iteration) the step of the device-resident iteration number.  A window holds at most 256 paths, so it contains at
// most one multiple of 256 of the dense order: seg_first is written by scatter, no search.
// Run by ONE block of NT threads: by k_segments (pipeline 4 / A-B), or — default — by the LAST block of the shading kernel
// to finish (segments_tail below), which saves a one-block launch (7 us + the launch gap) per bounce.  AGENT: agg[] was
// written by other workgroups of the SAME launch: read it with agent-scope (sc1) loads, never from this CU's L1.
template <int NT, bool AGENT>
__device__ __forceinline__ void segments_block(int n_prev, const int* __restrict__ agg, int* __restrict__ pref, int* __restrict__ first,
                                               int* __restrict__ count_out, int* iter_advance, int iter_stride) {
    constexpr int NW = NT / 64;
    __shared__ int s_w[NW];
    __shared__ int s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nw = n_prev > 0 ? (n_prev + PTK_BLOCK - 1) / PTK_BLOCK : 0;          // windows the shading kernel wrote
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < nw; base += 4 * NT) {                        // four consecutive windows per thread
        const int idx = base + 4 * tid;
        int4 c = make_int4(0, 0, 0, 0);
        if (AGENT) {
            if (idx < nw) c.x = __hip_atomic_load(agg + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (idx + 1 < nw) c.y = __hip_atomic_load(agg + idx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (idx + 2 < nw) c.z = __hip_atomic_load(agg + idx + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (idx + 3 < nw) c.w = __hip_atomic_load(agg + idx + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (idx + 3 < nw) c = *reinterpret_cast<const int4*>(agg + idx);   // agg is 16-byte aligned (hipMalloc) and idx % 4 == 0
        else { if (idx < nw) c.x = agg[idx]; if (idx + 1 < nw) c.y = agg[idx + 1]; if (idx + 2 < nw) c.z = agg[idx + 2]; }
        const int v = c.x + c.y + c.z + c.w;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) s_w[wid] = incl;
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { const int cw = s_w[w]; woff += (w < wid) ? cw : 0; tot += cw; }
        const int carry = s_carry;
        int excl = carry + woff + incl - v;
        const int cnt[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (idx + k < nw) {
                pref[idx + k] = excl;
                const int B = (excl + PTK_BLOCK - 1) / PTK_BLOCK;
                if (cnt[k] > 0 && B * PTK_BLOCK < excl + cnt[k]) first[B] = idx + k;
            }
            excl += cnt[k];
        }
        __syncthreads();
        if (tid == 0) s_carry = carry + tot;
        __syncthreads();
    }
    const int total = s_carry;
    if (tid < 8) pref[nw + tid] = total;                   // gather_source reads up to 4 entries past its window
    if (tid == 0) {
        if (count_out) *count_out = total;
        if (iter_advance) *iter_advance += iter_stride;    // last work of an iteration: step the device-resident iteration number
    }
}

// Tail of the shading kernel (segmented streams): every block publishes its survivor count with an agent-scope store,
// waits for it (s_waitcnt vmcnt(0)) and counts itself done; the block that completes the count is the last one, sees every
// count (MI355X_MICROARCH.md, inter-workgroup visibility: one lane per workgroup signals after its stores' vmcnt(0), the
// workgroup whose add came last loads after the add has returned, its other waves after a workgroup barrier; payload stored
// and loaded sc1) and runs segments_block.
// The count is a two-level tree — one counter per PTK_SEG_GROUP blocks, each on its own 128-B line, and a top counter the
// last block of every group adds to — because agent-scope atomics on ONE word serialise at ~11 ns each: 2 500 blocks on a
// single counter cost +17 us per launch (measured), 32 per counter in parallel ~0.4 us.  The last block also clears the
// counters for the next launch.
struct SegTail { int* done; int* pref; int* first; int* count_out; int* iter_advance; int iter_stride; };
template <int NT>
__device__ __forceinline__ void segments_tail(const SegTail& T, int n, int b, int total, int* __restrict__ agg_b, unsigned long long* clear_words = nullptr, int n_clear = 0) {
    __shared__ int s_last;
    const int nb = (n + NT - 1) / NT, ng = (nb + PTK_SEG_GROUP - 1) / PTK_SEG_GROUP;
    if (threadIdx.x == 0) {
        __hip_atomic_store(agg_b + b, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int grp = b / PTK_SEG_GROUP;
        const int gsize = min(PTK_SEG_GROUP, nb - grp * PTK_SEG_GROUP);
        int last = 0;
        if (__hip_atomic_fetch_add(T.done + 32 * (grp + 1), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1)
            last = __hip_atomic_fetch_add(T.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ng - 1;
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    for (int k = threadIdx.x; k <= ng; k += NT) __hip_atomic_store(T.done + 32 * k, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int k = threadIdx.x; k < n_clear; k += NT) __hip_atomic_store(clear_words + k, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (k_trace_shade's group words)
    segments_block<NT, true>(n, agg_b, T.pref, T.first, T.count_out, T.iter_advance, T.iter_stride);
}

// Material-binned shading (pipeline default).  The BSDF switch of Material::scatterSample is the divergent part of shading
// (Lambertian ~350, Microfacet ~600, Dielectric ~250, Light ~50 VALU; a wave with a mix executes them all).  A block of NT
// consecutive paths is counting-sorted by material type in LDS and thread j shades the j-th path of that order, so most waves
// see one type.  Each path keeps its own compacted slot (the RNG key), hit and state, so results are unchanged; only which
// lane does the work moves (within the block's contiguous NT-path window, so global accesses stay line-coalesced).
template <int MODE, int NT>
__global__ void __launch_bounds__(NT, 4) k_shade_binned(SceneView S, int iter_arg, const int* __restrict__ iter_ptr, int depth, int n_arg,
                                                      const int* __restrict__ d_n, PtkHits hits, PtkPaths paths,
                                                      const int* __restrict__ valid_a, const int* __restrict__ agg_a,
                                                      int* __restrict__ agg_b,
                                                      FilmArgs film, int spp, int slot_bias,
                                                      const int* __restrict__ srcidx, PtkPaths seg_out, SegTail seg_tail) {
    // srcidx  : physical slot of every dense path (segmented input stream), nullptr = the stream is dense
    // seg_out : block-local compaction: survivors go to seg_out[b * NT + rank within the block] and agg_b[b] is their number
    //           (until round 4 a null seg_out meant "update in place, k_scatter follows" = pipeline 4, retired)
    constexpr int NW = NT / 64, NKEY = 6, SUB = NT / PTK_BLOCK;      // SUB aggregate entries (256-path ranges) per block
    __shared__ int s_cnt[NW][NKEY];
    __shared__ int s_perm[NT];
    __shared__ int s_slot[NT];
    __shared__ int s_red[NW];
    const int n = d_n ? *d_n : n_arg;
    const int iter = ((iter_ptr ? *iter_ptr : iter_arg) - 1) * spp + 1;
    const Film img = {film, depth};
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (b * NT >= n) {
        // no live path at all: nobody runs the tail, block 0 does the (empty) bookkeeping of the bounce
        if (n <= 0 && b == 0 && seg_tail.done) segments_block<NT, false>(0, agg_b, seg_tail.pref, seg_tail.first, seg_tail.count_out, seg_tail.iter_advance, seg_tail.iter_stride);
        return;
    }
    PTK_PHASE_BEGIN();
    GeomStage gs; gs.xf = nullptr; gs.on = false;
    if (MODE != PTK_MODE_BSDF) gs = stage_geoms(S);
    __shared__ DevMaterial s_mats[PTK_LDS_MATS];
    const MatTable mt = stage_materials<NT>(S, s_mats);
    PTK_PHASE(1);                                    // 0: staging
    const int i = b * NT + tid;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // --- original order: flag, compacted slot, material key
    const int flag = (i < n) ? valid_a[i] : 0;
    __shared__ int s_src[NT];
    s_src[tid] = (srcidx && i < n) ? srcidx[i] : i;          // physical slot of this dense position, handed to whoever shades it
    // the hit's material id is loaded beside the flag, not behind it (one dependent global latency less at the head of the kernel);
    // for a miss the word is stale or never written — it is not used then
    const int mat_word = (i < n) ? __float_as_int(hits.p_mat[4 * (size_t)i + 3]) : 0;
    int key = NKEY - 1;
    if (flag) key = fetch_material_type(S, mt, mat_word);
    unsigned long long bal[NKEY];
#pragma unroll
    for (int k = 0; k < NKEY; ++k) bal[k] = __ballot(key == k);
    if (lane < NKEY) s_cnt[wid][lane] = __popcll(bal[lane]);
    // prior hit count of all earlier 256-path ranges (agg_a is indexed by 256-path range)
    // (16-byte loads, four passes in flight: with one 4-byte load per pass the head of the kernel waited up to ten L2 latencies in a row)
    int prior = 0;
    {
        const int B = b * SUB, B4 = B & ~3;
        int4 acc = make_int4(0, 0, 0, 0);
#pragma unroll 4
        for (int k = 4 * tid; k < B4; k += 4 * NT) { const int4 v = *reinterpret_cast<const int4*>(agg_a + k); acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        prior = (acc.x + acc.y) + (acc.z + acc.w);
        if (tid < B - B4) prior += agg_a[B4 + tid];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) prior += __shfl_down(prior, o, 64);
    if (lane == 0) s_red[wid] = prior;
    __syncthreads();
    prior = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) prior += s_red[w];
    // rank in the hit order (keys 0..4 are hits) and in the sorted order
    int hits_before = 0, rank = 0, n_active = 0;
#pragma unroll
    for (int k = 0; k < NKEY; ++k) {
        int before_waves = 0, total = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { const int c = s_cnt[w][k]; before_waves += (w < wid) ? c : 0; total += c; }
        if (k < NKEY - 1) { hits_before += before_waves + __popcll(bal[k] & lt); n_active += total; }
        if (key == k) rank += before_waves + __popcll(bal[k] & lt);
        else if (key > k) rank += total;
    }
    s_perm[rank] = tid;
    s_slot[tid] = prior + hits_before + slot_bias;      // slot_bias: rank-unique RNG keys under the tile partition (0 on one GPU)
    __syncthreads();
    PTK_PHASE(1);                                    // 1: flags, slot, counting sort
    // --- sorted order: thread `tid` shades item src
    int alive = 0, src = -1;
    PathRegs p;
    p.o = mk3(0, 0, 0); p.d = mk3(0, 0, 0); p.col = mk3(0, 0, 0); p.pix = 0; p.bounces = 0; p.prevPdf = 0.f;
    if (tid < n_active) {
        src = s_perm[tid];
        const int it = b * NT + src;
        const int slot = s_slot[src];
        const float4 h0 = ld4(hits.t_n, it), h1 = ld4(hits.p_mat, it);
        const float4 h2 = S.slim_hits ? make_float4(0.f, 0.f, __int_as_float(-1), __int_as_float(-1)) : ld4(hits.uv_ids, it);
        const int ps = s_src[src];
        const float4 o4 = ld4(paths.o_pix, ps), d4 = ld4(paths.d_bnc, ps), c4 = ld4(paths.col_pdf, ps);
        Hit h;
        h.t = h0.x; h.n = mk3(h0.y, h0.z, h0.w); h.p = mk3(h1.x, h1.y, h1.z); h.mat = __float_as_int(h1.w);
        h.uv.x = h2.x; h.uv.y = h2.y; h.geom = __float_as_int(h2.z); h.tri = __float_as_int(h2.w);
        p.o = mk3(o4.x, o4.y, o4.z); p.pix = __float_as_int(o4.w);
        p.d = mk3(d4.x, d4.y, d4.z); p.bounces = __float_as_int(d4.w);
        p.col = mk3(c4.x, c4.y, c4.z); p.prevPdf = c4.w;
        alive = shade_path<MODE>(S, gs, mt, iter, slot, depth, h, p, img);
    }
    PTK_PHASE(1);                                    // 2: loads + shading
    __syncthreads();                       // everyone is done reading s_slot; reuse it for the survivor flags in original order
    s_slot[tid] = 0;
    __syncthreads();
    if (src >= 0) s_slot[src] = alive;
    __syncthreads();
    const int alive_here = s_slot[tid];
    const unsigned long long ab = __ballot(alive_here != 0);
    if (lane == 0) s_red[wid] = __popcll(ab);
    __syncthreads();
    {                                      // block-local compaction (NT == PTK_BLOCK): survivor rank in original order
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { const int c = s_red[w]; before += (w < wid) ? c : 0; total += c; }
        s_perm[tid] = before + __popcll(ab & lt);
        __syncthreads();
        if (alive) {
            const int s2 = b * NT + s_perm[src];
            st4(seg_out.o_pix, s2, make_float4(p.o.x, p.o.y, p.o.z, __int_as_float(p.pix)));
            st4(seg_out.d_bnc, s2, make_float4(p.d.x, p.d.y, p.d.z, __int_as_float(p.bounces)));
            st4(seg_out.col_pdf, s2, make_float4(p.col.x, p.col.y, p.col.z, p.prevPdf));
        }
        PTK_PHASE(1);                                // 3: survivor compaction + stores
        if (seg_tail.done) segments_tail<NT>(seg_tail, n, b, total, agg_b);       // the last block to get here finishes the bounce
        else if (tid == 0) agg_b[b] = total;
        PTK_PHASE(1);                                // 4: segment bookkeeping (the last block does the work)
        PTK_PHASE_END(1);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// k_shade_trace — the default render kernel (round 3): SHADE bounce d-1 and TRACE bounce d in one launch.
//
// The staged pipeline ran two full-width kernels per bounce (k_intersect, k_shade_binned).  With three iterations in flight the
// machine spent a third of its time between kernels: rocprofv3 shows an average of 1.9 kernels in flight on three streams, every
// kernel boundary is a drain, a dispatch and a ramp (and a cache write-back between the XCDs).  A surviving path's next ray is in
// registers at the end of shading — so the block that shaded it finds its closest hit at once, block-cooperatively like k_intersect
// (analytic_closest_lds), and writes path state AND hit record together.  One iteration = 1 + traceDepth launches instead of
// 1 + 2 x traceDepth (+ traceDepth one-block k_segments launches with one iteration at a time).
//
// What makes this possible without any inter-block communication inside the launch:
//   * survivors stay in their block's window (segmented streams, as before), and the hit record of a survivor is stored at the
//     SAME physical slot as its path state (two hit buffers ping-pong with the two path buffers);
//   * the compacted slot of a hit path — the reference's RNG key (pathtrace.cu:458,532) — is
//         hit_pref[window] + (rank of the path among the hits of its window),
//     the rank written next to the hit record by the block that traced it (`hr`: >= 0 rank, -1 miss, -2 empty slot), hit_pref by
//     the last block of the launch (segments_tail) together with the window prefix.  The O(blocks) re-summation of the per-block
//     hit counts at the head of k_shade_binned is gone;
//   * LDS: the shading phase's tables (materials, counting-sort permutation, slots) and the tracing phase's pair list share 8 KB,
//     so the kernel keeps k_intersect's 24.6 KB = six blocks per CU.
// GEN: the first launch of an iteration generates the camera rays (generateRayFromCamera, pathtrace.cu:135-163) instead of shading.
// SPLIT (scenes with triangles): the trace phase is the analytic part + the root-box test; k_list / k_traverse_wide / k_finalize
// follow on the survivors' physical slots and k_finalize writes `hr` (a triangle hit changes the ranks).
// Results are unchanged: same paths, same keys, same order of additions into every pixel.
// ------------------------------------------------------------------------------------------------------------------
struct TraceIO {
    // the bounce being shaded (unused with GEN): paths, their hit records and hit ranks at PHYSICAL slots, the window tables
    PtkPaths cur; PtkHits hcur; const int* hr_cur;      // hr: the slot's status + (hit-dense shading) the window's inverse map, full-width layouts
    const int* pref; const int* first; const int* hit_pref; const int* first_h;      // first_h: COMPACT (hit-dense shading)
    // the bounce being traced: survivors (or fresh camera rays), their hit records and ranks
    PtkPaths nxt; PtkHits hnxt; int* hr_nxt;
    int* agg_s; int* agg_h;              // per-window survivor / hit counts of this launch
    int* valid_nxt;                      // SPLIT: 0 / 1 / -2 per slot for k_traverse / k_finalize
    int* heavy_flag; int* heavy_agg;     // SPLIT: rays that enter the BVH root box
};
struct SegTail2 { int* done; int* pref; int* first; int* hit_pref; int* count_out; int* iter_advance; int iter_stride; unsigned long long* ray_total;
                  int* first_h; int* hcount_out;       // COMPACT (hit-dense shading): window of hit index 256 B, total hits of the launch
                  const int* cum_prev; int* cum_out; int cum_stride, ring_div; };      // tiles in flight (Workspace::cum_*), null = off
struct GenArgs { DevCamera cam; int trace_depth, n_parts, part, tile_rows, n_pix, late_stores; };

// segments_block for the fused launches: the window prefix of the survivor counts (pref / first / next live count) AND the prefix of the hit
// counts (hit_pref) in ONE pass — the last block of a launch runs alone, its time is on every lane's critical path (two separate scans took
// twice the barriers).  Four consecutive windows per thread; agg_h == nullptr: no hit prefix (SPLIT: k_finalize2 builds it).
template <int NT, int WIN = PTK_BLOCK, int WPT = 4>      // WIN: slots per window (64 for k_shade_trace_w's wave windows); WPT: consecutive windows per thread and pass (the
                                                         // loads of a pass are in flight together: the last block runs alone, each pass is one exposed round trip)
__device__ __forceinline__ void segments_block2(int n_prev, const int* __restrict__ agg_s, const int* __restrict__ agg_h, int* __restrict__ pref, int* __restrict__ first,
                                                int* __restrict__ hit_pref, int* __restrict__ count_out, int* iter_advance, int iter_stride,
                                                int* __restrict__ first_h = nullptr, int* __restrict__ hcount_out = nullptr,
                                                const int* __restrict__ cum_prev = nullptr, int* __restrict__ cum_out = nullptr, int cum_at = 0) {
    constexpr int NW = NT / 64;
    __shared__ int s_ws[NW], s_wh[NW];
    __shared__ int s_carry_s, s_carry_h;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nw = n_prev > 0 ? (n_prev + WIN - 1) / WIN : 0;
    if (tid == 0) { s_carry_s = 0; s_carry_h = 0; }
    __syncthreads();
    // the counts of pass p + 1 are requested before pass p is processed: the passes are serial (carry), their load round trips need not be
    int ns[WPT], nh[WPT];
    auto fetch = [&](int base, int* os, int* oh) {
        const int idx = base + WPT * tid;
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            os[k] = 0; oh[k] = 0;
            if (idx + k < nw) {
                os[k] = __hip_atomic_load(agg_s + idx + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (agg_h) oh[k] = __hip_atomic_load(agg_h + idx + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    fetch(0, ns, nh);
    for (int base = 0; base < nw; base += WPT * NT) {
        const int idx = base + WPT * tid;
        int cs[WPT], ch[WPT];
#pragma unroll
        for (int k = 0; k < WPT; ++k) { cs[k] = ns[k]; ch[k] = nh[k]; }
        if (base + WPT * NT < nw) fetch(base + WPT * NT, ns, nh);
        int vs = 0, vh = 0;
#pragma unroll
        for (int k = 0; k < WPT; ++k) { vs += cs[k]; vh += ch[k]; }
        int is = vs, ih = vh;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int a = __shfl_up(is, o, 64), c = __shfl_up(ih, o, 64); if (lane >= o) { is += a; ih += c; } }
        if (lane == 63) { s_ws[wid] = is; s_wh[wid] = ih; }
        __syncthreads();
        int offs = 0, tots = 0, offh = 0, toth = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { const int a = s_ws[w], c = s_wh[w]; offs += (w < wid) ? a : 0; tots += a; offh += (w < wid) ? c : 0; toth += c; }
        const int carry_s = s_carry_s, carry_h = s_carry_h;
        int es = carry_s + offs + is - vs, eh = carry_h + offh + ih - vh;
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            if (idx + k < nw) {
                pref[idx + k] = es;
                if (agg_h) hit_pref[idx + k] = eh;
                const int B = (es + WIN - 1) / WIN;
                if (cs[k] > 0 && B * WIN < es + cs[k]) first[B] = idx + k;
                if (first_h) {                                   // the same table over the HIT order (the order the next launch shades in)
                    const int Bh = (eh + WIN - 1) / WIN;
                    if (ch[k] > 0 && Bh * WIN < eh + ch[k]) first_h[Bh] = idx + k;
                }
            }
            es += cs[k]; eh += ch[k];
        }
        __syncthreads();
        if (tid == 0) { s_carry_s = carry_s + tots; s_carry_h = carry_h + toth; }
        __syncthreads();
    }
    const int total = s_carry_s, total_h = s_carry_h;
    if (tid < 8) { pref[nw + tid] = total; if (agg_h) hit_pref[nw + tid] = total_h; }
    if (tid == 0) {
        if (count_out) *count_out = total;
        if (hcount_out) *hcount_out = total_h;
        if (cum_out && agg_h) cum_out[cum_at] = cum_read(cum_prev, 0, cum_at) + total_h;      // (SPLIT: k_finalize2 writes it, the hits are final there)
        if (iter_advance) *iter_advance += iter_stride;
    }
}

// end of a k_shade_trace block: publish the window's survivor and hit counts; the last block of the launch builds the tables
template <int NT>
__device__ __forceinline__ void segments_tail2(const SegTail2& T, int n, int b, int survivors, int hits, int* __restrict__ agg_s, int* __restrict__ agg_h, int cum_at) {
    __shared__ int s_last;
    const int nb = (n + NT - 1) / NT, ng = (nb + PTK_SEG_GROUP - 1) / PTK_SEG_GROUP;
    if (threadIdx.x == 0) {
        __hip_atomic_store(agg_s + b, survivors, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (agg_h) __hip_atomic_store(agg_h + b, hits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int grp = b / PTK_SEG_GROUP;
        const int gsize = min(PTK_SEG_GROUP, nb - grp * PTK_SEG_GROUP);
        int last = 0;
        if (__hip_atomic_fetch_add(T.done + 32 * (grp + 1), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1)
            last = __hip_atomic_fetch_add(T.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ng - 1;
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    for (int k = threadIdx.x; k <= ng; k += NT) __hip_atomic_store(T.done + 32 * k, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    segments_block2<NT>(n, agg_s, (T.hit_pref && agg_h) ? agg_h : nullptr, T.pref, T.first, T.hit_pref, T.count_out, T.iter_advance, T.iter_stride,
                        (T.hit_pref && agg_h) ? T.first_h : nullptr, (T.hit_pref && agg_h) ? T.hcount_out : nullptr, T.cum_prev, T.cum_out, cum_at);
    if (threadIdx.x == 0 && T.ray_total) atomicAdd(T.ray_total, (unsigned long long)*T.count_out);     // rays of the bounce just traced
}

// dense index -> (physical slot, window) of the segmented stream (gather_source + the window, whose hit prefix the caller needs)
__device__ __forceinline__ int gather_source_w(const int* __restrict__ pref, const int* __restrict__ first, const int* __restrict__ hit_pref,
                                               int b, int i, int n, int nw, int& hit_base) {
    const int w0 = first[b];                                      // block-uniform: scalar loads
    const int p0 = pref[w0], p1 = pref[w0 + 1], p2 = pref[w0 + 2], p3 = pref[w0 + 3], p4 = pref[w0 + 4];
    const int h0 = hit_pref[w0], h1 = hit_pref[w0 + 1], h2 = hit_pref[w0 + 2], h3 = hit_pref[w0 + 3], h4 = hit_pref[w0 + 4];
    hit_base = 0;
    if (i >= n) return 0;
    int k = (i >= p1) + (i >= p2) + (i >= p3) + (i >= p4);
    int base = k == 0 ? p0 : (k == 1 ? p1 : (k == 2 ? p2 : (k == 3 ? p3 : p4)));
    hit_base = k == 0 ? h0 : (k == 1 ? h1 : (k == 2 ? h2 : (k == 3 ? h3 : h4)));
    if (k == 4) {                                                 // many sparse windows in a row
        k = sparse_window(pref, w0 + 4, nw, i, base) - w0;
        hit_base = hit_pref[w0 + k];
    }
    return (w0 + k) * PTK_BLOCK + (i - base);
}

// COMPACT (triangle-free scenes without textures, BSDF / DIRECT modes): 64 B per path instead of 84.  Shading never reads the ray origin or
// prevPdf outside MIS (lightPDF, pathtrace.cu:547-554) nor the hit distance beyond its sign, and (uv, geomID, triangleID) only with textures
// or MIS — so the four 16-byte streams are  paths.o_pix = (direction, pixelIndex), paths.d_bnc = (color, remainingBounces),
// hits.t_n = (normal, word), hits.p_mat = (point, materialId); col_pdf, uv_ids and the hr array stay untouched.  `word` of slot p: sign = slot p
// itself missed; low bits = the window's inverse map for hit-dense shading (hit number p of the window: its slot | material type << 8).
// MIS adds a fifth stream, paths.col_pdf = (origin, prevPdf), and stores the primitive instead of the material in hits.p_mat.w (lightPDF needs
// geomID; the material id is a field of the primitive's staged record; triangleID is -1 without triangles): 80 B per path instead of 100.
// The timed configuration is partly bound by HBM traffic (+32 B per path costs 5.7 %, measured): −24 % bytes per path-bounce.
template <int MODE, bool GEN, bool SPLIT, bool COMPACT>
#ifndef PTK_RAY3
#define PTK_RAY3 0                  // 1: BSDF-mode k_shade_trace keeps its rays as six float planes in LDS (6 instead of 8 KB), see R3 there
#endif
#ifndef PTK_FUSED_WAVES_BSDF
#define PTK_FUSED_WAVES_BSDF 7
#endif
#ifndef PTK_FUSED_WAVES_MIS
#define PTK_FUSED_WAVES_MIS 6
#endif
#ifndef PTK_FUSED_WAVES_DIRECT
#define PTK_FUSED_WAVES_DIRECT 6
#endif
#ifndef PTK_FUSED_WAVES_GEN
#define PTK_FUSED_WAVES_GEN 0     // > 0: occupancy the camera launches are compiled for; 0: that of their mode (measured: MIS layout 22.2 at six vs 21.1 Grays/s at seven on mis_test)
#endif
__global__ void __launch_bounds__(PTK_BLOCK, ((GEN && PTK_FUSED_WAVES_GEN > 0) ? PTK_FUSED_WAVES_GEN : (MODE == PTK_MODE_MIS ? PTK_FUSED_WAVES_MIS : (MODE == PTK_MODE_DIRECTLI ? PTK_FUSED_WAVES_DIRECT : PTK_FUSED_WAVES_BSDF)))) k_shade_trace(SceneView S, GenArgs gen, int iter_arg, const int* __restrict__ iter_ptr, int depth, int n_arg,
                                                             const int* __restrict__ d_n, TraceIO io, FilmArgs film,
                                                             int spp, int slot_bias, SegTail2 tail) {
    constexpr int NT = PTK_BLOCK, NW = NT / 64, NKEY = 6;
    // Occupancy: 22.1 KB of LDS and <= 72 VGPRs = seven blocks per CU (with the 16-primitive pair list and 64 staged materials of the
    // staged kernels: 25.9 KB = six; measured +x %).  MATS materials are staged (the rest, if any, come from global memory); the pair
    // list takes LP primitives per pass.
    constexpr int MATS = 16, LP = 8;
    constexpr int OVER = (MATS * (int)sizeof(DevMaterial) + 3 * NT * (int)sizeof(int)) > NT * LP * 2 ? (MATS * (int)sizeof(DevMaterial) + 3 * NT * (int)sizeof(int)) : NT * LP * 2;
    // shared by the two phases: shading = materials + permutation / slot / source tables (3 x 1 KB); tracing = pair list
    __shared__ __attribute__((aligned(16))) unsigned char s_over[OVER];
    // the trace phase's LDS: rays (8 KB), the winners' hit points (6 KB), the ds_min keys (2 KB)
    // R3 (-DPTK_RAY3=1, BSDF mode — no shadow phase borrows the upper half): the rays as six float planes, 6 KB; 20.1 instead of 22.1 KB per block, what
    // an eighth block per CU needs (with PTK_FUSED_WAVES_BSDF=8: 64 VGPRs)
    constexpr bool R3 = PTK_RAY3 && MODE == PTK_MODE_BSDF;
    constexpr int RAY_U4 = R3 ? (3 * NT) / 2 : 2 * NT;        // float4 units the rays take
    __shared__ float4 s_stage[RAY_U4 + 2 * NT];
    float4* s_ray = s_stage;
    float* s_r3 = reinterpret_cast<float*>(s_stage);
    float2* s_res = reinterpret_cast<float2*>(s_stage + RAY_U4);
    unsigned long long* s_best = reinterpret_cast<unsigned long long*>(s_stage + RAY_U4 + (3 * NT) / 2);
    __shared__ int s_cnt[NW][NKEY];
    __shared__ int s_red[NW];
    __shared__ int s_tcnt[NW];
    __shared__ unsigned long long s_alive[NT / 64];
    // MIS / DIRECT: next-event shadow rays are deferred to a block-cooperative phase between shading and tracing (DEFER_NEE below)
    constexpr bool DEFER_NEE = !GEN && MODE != PTK_MODE_BSDF;
    __shared__ unsigned long long s_occ[NT / 64];
    __shared__ int s_anyreq;
    DevMaterial* s_mats = reinterpret_cast<DevMaterial*>(s_over);
    int* s_perm = reinterpret_cast<int*>(s_over + MATS * sizeof(DevMaterial));
    int* s_slot = s_perm + NT;
    int* s_src = s_slot + NT;
    const TraceLds L = {s_ray, s_res, s_best, reinterpret_cast<unsigned short*>(s_over), s_tcnt};

    const int step = iter_ptr ? *iter_ptr : iter_arg;
    const int n = GEN ? gen.n_pix * spp : (d_n ? *d_n : n_arg);      // paths of the bounce being shaded (GEN: rays to generate)
    const Film img = {film, depth};               // (depth = the bounce being shaded; -1 in the camera launch, which shades nothing)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if ((int)blockIdx.x * NT >= n) {
        // no live path at all: nobody runs the tail, block 0 does the (empty) bookkeeping of the bounce
        if (n <= 0 && blockIdx.x == 0) {
            if (tail.hit_pref && tid < 8) tail.hit_pref[tid] = 0;
            if (tail.hcount_out && tid == 0) *tail.hcount_out = 0;
            if (tail.cum_out && tid == 0) { const int at = cum_row(step, tail.ring_div, tail.cum_stride) + depth + 1; tail.cum_out[at] = cum_read(tail.cum_prev, 0, at); }
            segments_block<NT, false>(0, io.agg_s, tail.pref, tail.first, tail.count_out, tail.iter_advance, tail.iter_stride);
        }
        return;
    }
    PTK_PHASE_BEGIN();
    // the trace phase always needs the primitives; MIS / DIRECT shading too.  No barrier of its own (round 4: one barrier and one exposed load
    // latency less at the head of every block): a shading launch publishes the copy with stage_materials' barrier, the camera launch with the
    // first barrier of the trace phase — both come before the first read of gs.xf
    const GeomStage gs = stage_geoms(S, false);
    if (!GEN) PTK_PHASE(0);                          // 0: counts + primitive records -> LDS   (diagnostic build only, shading launches)
    // (round 4, measured and removed: an occupancy-sized grid whose blocks loop over the windows b, b + G, ... and stage the primitives once —
    // cornell 12.40 -> 8.74 Grays/s one iteration at a time, 23.17 -> 12.74 with three in flight, profiles/r4_ab_persistent_grid.txt: a resident
    // block that loops keeps its slot through every phase of its next window, where the dispatcher hands a finished block's slot to whichever
    // launch has work, and with several lanes the second lane's grid cannot start before the first one's blocks have looped to the end)
    { const int b = blockIdx.x;
    if (tid < NT / 64) { s_alive[tid] = 0ull; if (DEFER_NEE) { s_occ[tid] = 0ull; s_anyreq = 0; } }      // (published by the barrier in stage_materials)
    const int i = b * NT + tid;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int alive = 0, my_rank = -1;           // this thread's shaded path survives, at rank my_rank of the window
    int total = 0;                         // survivors of the block
    // shadow ray of thread t -> slots 2t, 2t + 1 of the (still idle) upper half of the trace phase's LDS: s_res / s_best are first written
    // inside analytic_closest_lds
    ShadowReq rq; rq.on = false; rq.slot = s_stage + RAY_U4 + tid; rq.dstride = NT; rq.pix = 0; rq.contrib = mk3(0, 0, 0);
    int gen_pixw = 0;                      // GEN + COMPACT: the camera ray's pixel word, stored with the path state after the trace phase
    // ... when the scene is open (gen.late_stores: the host saw fewer than ~half of the camera rays of earlier iterations hit something);
    // otherwise before it, where the stores overlap the tracing (cornell.txt: 22.8 vs 22.6 Grays/s).  Same values either way.  (Deciding here
    // from hcounts[0] of the lane's previous iteration puts a dependent load ahead of the stores: -3 % on cornell.txt DIRECT.)
    const bool gen_late = GEN && COMPACT && gen.late_stores != 0;
    if (GEN) {
        // ---- generateRayFromCamera: path i -> pixel through the tile partition, sample s of the batch (see k_generate)
        total = min(NT, n - b * NT);
        if (i < n) {
            alive = 1; my_rank = tid;
            const int n_pix = gen.n_pix;
            const int sidx = i / n_pix, jp = i - sidx * n_pix;
            const int iter = (step - 1) * spp + sidx + 1;
            const int lr = jp / gen.cam.W, x = jp - lr * gen.cam.W;
            const int y = ((lr / gen.tile_rows) * gen.n_parts + gen.part) * gen.tile_rows + (lr % gen.tile_rows);
            Rng rng = make_rng(x, y, iter);                       // argument order of pathtrace.cu:148: (iter:=x, index:=y, depth:=iter)
            const f2 r = sample2D(rng);
            const float fx = static_cast<float>(static_cast<float>(x) + ((double)r.x - 0.5) - gen.cam.W * 0.5f);
            const float fy = static_cast<float>(static_cast<float>(y) + ((double)r.y - 0.5) - gen.cam.H * 0.5f);
            const f3 d = normalize(gen.cam.view - gen.cam.right * gen.cam.plx * fx - gen.cam.up * gen.cam.ply * fy);
            if (COMPACT) {
                // hit-dense shading never loads a camera ray that missed: in an open scene (gen_late) the path state is stored after the trace
                // phase, for the hits only (mis_test.txt: 200 k of 1 048 k camera rays — 40 of 50 MB less written by this launch)
                gen_pixw = (sidx << 24) | (x + y * gen.cam.W);
                if (!gen_late) {
                    st4(io.nxt.o_pix, i, make_float4(d.x, d.y, d.z, __int_as_float(gen_pixw)));
                    st4(io.nxt.d_bnc, i, make_float4(1.f, 1.f, 1.f, __int_as_float(gen.trace_depth)));
                    if (MODE == PTK_MODE_MIS) st4(io.nxt.col_pdf, i, make_float4(gen.cam.pos.x, gen.cam.pos.y, gen.cam.pos.z, -1.f));
                }
            } else {
                st4(io.nxt.o_pix, i, make_float4(gen.cam.pos.x, gen.cam.pos.y, gen.cam.pos.z, __int_as_float((sidx << 24) | (x + y * gen.cam.W))));
                st4(io.nxt.d_bnc, i, make_float4(d.x, d.y, d.z, __int_as_float(gen.trace_depth)));
                st4(io.nxt.col_pdf, i, make_float4(1.f, 1.f, 1.f, -1.f));
            }
            if (R3) {
                s_r3[tid] = gen.cam.pos.x; s_r3[NT + tid] = gen.cam.pos.y; s_r3[2 * NT + tid] = gen.cam.pos.z;
                s_r3[3 * NT + tid] = d.x; s_r3[4 * NT + tid] = d.y; s_r3[5 * NT + tid] = d.z;
            } else {
                s_ray[tid] = make_float4(gen.cam.pos.x, gen.cam.pos.y, gen.cam.pos.z, 0.f);
                s_ray[NT + tid] = make_float4(d.x, d.y, d.z, 0.f);
            }
        }
    } else {
        // ---- shading phase: k_shade_binned's, with the compacted slot from the window tables
        const int iter = (step - 1) * spp + 1;
        const int slot_off = slot_bias + cum_read(tail.cum_prev, cum_row(step, tail.ring_div, tail.cum_stride), depth);      // tiles in flight: the hits of this bounce in the bands above
        // windows of the stream being read = ceil(paths the previous launch shaded / 256); the camera launch's stream (depth 0) is dense
        const int nw_prev = depth > 0 ? (d_n[-1] + NT - 1) / NT : (gen.n_pix * spp + NT - 1) / NT;
        int src_mine = 0, flag = 0, key = NKEY - 1, slot_mine = 0;
        // (the window tables and the map word are loaded BEFORE the materials are staged: their two dependent round trips overlap the staging copy
        // instead of following its barrier)
        MatTable mt;
        if (SPLIT) mt = stage_materials<NT, MATS>(S, s_mats);
        if (!SPLIT) {
            // HIT-DENSE shading: n and i count the paths of the previous launch that HIT something, in stream order — which is the reference's
            // compacted slot, the RNG key, itself.  A miss is never shaded (h.t <= 0: colour 0, no radiance, dead), so it is never even loaded: an
            // open scene (mis_test.txt: 200 435 hits among 1 048 576 camera rays, 4 953 among the 196 540 rays of bounce 1) runs 783 + 20 full
            // blocks instead of 4 096 + 768 mostly idle ones.  Hit index -> window by the hit prefix (five scalar loads, first_h[b] = the
            // window of hit 256 b); hit r of window w sits at the slot the tracing block recorded as the window's r-th inverse-map word (with
            // the material type, the sort key): in the normal's spare word of slot 256 w + r (compact streams) or in the hr word of that slot.
            // Survivors keep their order, so every later key is unchanged.
            const int w0 = io.first_h[b];
            const int h0 = io.hit_pref[w0], h1 = io.hit_pref[w0 + 1], h2 = io.hit_pref[w0 + 2], h3 = io.hit_pref[w0 + 3], h4 = io.hit_pref[w0 + 4];
            if (i < n) {
                int k = (i >= h1) + (i >= h2) + (i >= h3) + (i >= h4);
                int base = k == 0 ? h0 : (k == 1 ? h1 : (k == 2 ? h2 : (k == 3 ? h3 : h4)));
                if (k == 4) k = sparse_window(io.hit_pref, w0 + 4, nw_prev, i, base) - w0;
                const int at = (w0 + k) * NT + (i - base);
                const int word = (COMPACT ? __float_as_int(io.hcur.t_n[4 * (size_t)at + 3]) : io.hr_cur[at]) & 0x3fffffff;      // in the normal's spare word / the hr word
                src_mine = (w0 + k) * NT + (word & 0xff);
                flag = 1; key = word >> 8; slot_mine = i + slot_off;
            }
            mt = stage_materials<NT, MATS>(S, s_mats);
        } else {
            int hit_base = 0;
            src_mine = gather_source_w(io.pref, io.first, io.hit_pref, b, i, n, nw_prev, hit_base);      // 0 for i >= n
            // one word per path at the head — rank among the window's hits | material type << 8 (the sort key), -1 for a miss, in the hr array.
            // (Reading the material id from the hit record here, a second strided stream ahead of the sort, cost 12 % of the headline.)
            const int hr_word = (i < n) ? io.hr_cur[src_mine] : -1;
            flag = hr_word >= 0;
            // (SPLIT: k_finalize2 writes the word — a triangle hit's material is only known there)
            if (flag) key = hr_word >> 8;
            slot_mine = hit_base + (hr_word & 0xff) + slot_off;
        }
        s_src[tid] = src_mine;
        unsigned long long bal[NKEY];
#pragma unroll
        for (int k = 0; k < NKEY; ++k) bal[k] = __ballot(key == k);
        if (lane < NKEY) s_cnt[wid][lane] = __popcll(bal[lane]);
        __syncthreads();
        int rank = 0, n_active = 0;
#pragma unroll
        for (int k = 0; k < NKEY; ++k) {
            int before_waves = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) { const int c = s_cnt[w][k]; before_waves += (w < wid) ? c : 0; tot += c; }
            if (k < NKEY - 1) n_active += tot;
            if (key == k) rank += before_waves + __popcll(bal[k] & lt);
            else if (key > k) rank += tot;
        }
        s_perm[rank] = tid;
        s_slot[tid] = slot_mine;
        __syncthreads();
        PTK_PHASE(0);                                // 1: materials -> LDS, window tables, map word, counting sort
        int src = -1;
        PathRegs p;
        p.o = mk3(0, 0, 0); p.d = mk3(0, 0, 0); p.col = mk3(0, 0, 0); p.pix = 0; p.bounces = 0; p.prevPdf = 0.f;
        if (tid < n_active) {
            src = s_perm[tid];
            const int slot = s_slot[src];
            const int ps = s_src[src];
            Hit h;
            if (COMPACT) {
                const float4 h0 = ld4(io.hcur.t_n, ps), h1 = ld4(io.hcur.p_mat, ps), a4 = ld4(io.cur.o_pix, ps), b4 = ld4(io.cur.d_bnc, ps);
                h.t = 1.0f; h.n = mk3(h0.x, h0.y, h0.z); h.p = mk3(h1.x, h1.y, h1.z); h.mat = __float_as_int(h1.w);
                h.uv.x = 0.f; h.uv.y = 0.f; h.geom = -1; h.tri = -1;
                p.o = mk3(0, 0, 0); p.pix = __float_as_int(a4.w);
                p.d = mk3(a4.x, a4.y, a4.z); p.bounces = __float_as_int(b4.w);
                p.col = mk3(b4.x, b4.y, b4.z); p.prevPdf = -1.f;
                if (MODE == PTK_MODE_MIS) {
                    const float4 c4 = ld4(io.cur.col_pdf, ps);
                    p.o = mk3(c4.x, c4.y, c4.z); p.prevPdf = c4.w;
                    h.geom = __float_as_int(h1.w);
                    h.mat = gs.on ? __float_as_int(gs.xf[(size_t)h.geom * PTK_XF_FLOATS + 34]) : S.geoms[h.geom].materialid;
                }
            } else {
                const float4 h0 = ld4(io.hcur.t_n, ps), h1 = ld4(io.hcur.p_mat, ps);
                const float4 h2 = S.slim_hits ? make_float4(0.f, 0.f, __int_as_float(-1), __int_as_float(-1)) : ld4(io.hcur.uv_ids, ps);
                const float4 o4 = ld4(io.cur.o_pix, ps), d4 = ld4(io.cur.d_bnc, ps), c4 = ld4(io.cur.col_pdf, ps);
#ifdef PTK_DIAG_EXTRA_TRAFFIC    // timing experiments only: +16 B read and +16 B written per path (the unused third of the slim hit record)
                { const float4 x4 = ld4(io.hcur.uv_ids, ps); asm volatile("" :: "v"(x4.x), "v"(x4.y), "v"(x4.z), "v"(x4.w)); }
#endif
                h.t = h0.x; h.n = mk3(h0.y, h0.z, h0.w); h.p = mk3(h1.x, h1.y, h1.z); h.mat = __float_as_int(h1.w);
                h.uv.x = h2.x; h.uv.y = h2.y; h.geom = __float_as_int(h2.z); h.tri = __float_as_int(h2.w);
                p.o = mk3(o4.x, o4.y, o4.z); p.pix = __float_as_int(o4.w);
                p.d = mk3(d4.x, d4.y, d4.z); p.bounces = __float_as_int(d4.w);
                p.col = mk3(c4.x, c4.y, c4.z); p.prevPdf = c4.w;
            }
            rq.pix = p.pix;
            alive = shade_path<MODE>(S, gs, mt, iter, slot, depth, h, p, img, DEFER_NEE ? &rq : nullptr);
            if (DEFER_NEE && rq.on) s_anyreq = 1;
        }
        PTK_PHASE(0);                                // 2: record loads + shading
        // survivor rank in the ORIGINAL (dense) order — the next bounce's compacted order, hence its RNG keys: every survivor sets the bit of
        // its original position in a 256-bit LDS mask, one barrier, rank = set bits below it (five barriers with a flag array + ballot before)
        if (alive) atomicOr(&s_alive[src >> 6], 1ull << (src & 63));
        __syncthreads();
        {
            const unsigned long long m0 = s_alive[0], m1 = s_alive[1], m2 = s_alive[2], m3 = s_alive[3];
            const int c0 = __popcll(m0), c1 = __popcll(m1), c2 = __popcll(m2);
            total = c0 + c1 + c2 + __popcll(m3);
            if (alive) {
                const int w = src >> 6;
                const unsigned long long mw = w == 0 ? m0 : (w == 1 ? m1 : (w == 2 ? m2 : m3));
                my_rank = (w > 0 ? c0 : 0) + (w > 1 ? c1 : 0) + (w > 2 ? c2 : 0) + __popcll(mw & ((1ull << (src & 63)) - 1ull));
            }
        }
        if (alive) {
            const int s2 = b * NT + my_rank;
            if (COMPACT) {
                st4(io.nxt.o_pix, s2, make_float4(p.d.x, p.d.y, p.d.z, __int_as_float(p.pix)));
                st4(io.nxt.d_bnc, s2, make_float4(p.col.x, p.col.y, p.col.z, __int_as_float(p.bounces)));
                if (MODE == PTK_MODE_MIS) st4(io.nxt.col_pdf, s2, make_float4(p.o.x, p.o.y, p.o.z, p.prevPdf));
            } else {
                st4(io.nxt.o_pix, s2, make_float4(p.o.x, p.o.y, p.o.z, __int_as_float(p.pix)));
                st4(io.nxt.d_bnc, s2, make_float4(p.d.x, p.d.y, p.d.z, __int_as_float(p.bounces)));
                st4(io.nxt.col_pdf, s2, make_float4(p.col.x, p.col.y, p.col.z, p.prevPdf));
            }
            if (R3) {
                s_r3[my_rank] = p.o.x; s_r3[NT + my_rank] = p.o.y; s_r3[2 * NT + my_rank] = p.o.z;
                s_r3[3 * NT + my_rank] = p.d.x; s_r3[4 * NT + my_rank] = p.d.y; s_r3[5 * NT + my_rank] = p.d.z;
            } else {
                s_ray[my_rank] = make_float4(p.o.x, p.o.y, p.o.z, __int_as_float(p.pix));          // hand the new ray to the thread that traces it
                s_ray[NT + my_rank] = make_float4(p.d.x, p.d.y, p.d.z, __int_as_float(p.bounces));
            }
        }
    }
    __syncthreads();                           // shading tables are dead, s_ray is complete: the pair list may overwrite s_over
    if (!GEN) PTK_PHASE(0);                          // 3: survivor ranks + path stores
    if (DEFER_NEE && s_anyreq) {               // block-uniform
        // ---- shadow phase (LightSampler::occulusionTest, lightSample.h:59-139) for the block's next-event samples.  Inside shade_path the
        // test is a per-lane candidate loop that runs max-over-lanes trips per wave and holds the shading state in registers meanwhile
        // (MIS: 109 VGPRs = four waves per SIMD; without it 87 = five).  Here: analytic primitives block-cooperatively, the BVH per lane.
        const TraceLds Lsh = {s_stage + RAY_U4, nullptr, nullptr, reinterpret_cast<unsigned short*>(s_over), s_tcnt};
        bool occ = false;
        f3 so = mk3(0, 0, 0), sd = mk3(0, 0, 0); float s_minT = 0.f; int s_pix = 0;
        if (rq.on) { const float4 o4 = rq.slot[0], d4 = rq.slot[NT]; so = mk3(o4.x, o4.y, o4.z); s_minT = o4.w; sd = mk3(d4.x, d4.y, d4.z); s_pix = __float_as_int(d4.w); }
        if (gs.on) {
            analytic_anyhit_lds<NT, LP>(S, gs, Lsh, s_occ, rq.on, so, sd);
            occ = (s_occ[tid >> 6] >> (tid & 63)) & 1ull;
        } else {
            if (rq.on) occ = occluded_analytic(S, gs, so, sd, s_minT);
            __syncthreads();                   // (analytic_anyhit_lds ends with a barrier: the rays are dead before the trace phase writes s_best)
        }
        if ((SPLIT || !COMPACT) && rq.on && !occ) occ = occluded_bvh(S, so, sd, s_minT);      // (returns at once without triangles)
        if (rq.on && !occ) add_radiance(img, s_pix, rq.contrib);
    }
    // ---- trace phase: thread t < total traces survivor t of the window (computeIntersections, pathtrace.cu:169-346)
    int hits = 0;
    if (total > 0) {                           // block-uniform
        const bool active = tid < total;
        f3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 0);
        if (active) {
            if (R3) { ro = mk3(s_r3[tid], s_r3[NT + tid], s_r3[2 * NT + tid]); rd = mk3(s_r3[3 * NT + tid], s_r3[4 * NT + tid], s_r3[5 * NT + tid]); }
            else { const float4 o4 = s_ray[tid], d4 = s_ray[NT + tid]; ro = mk3(o4.x, o4.y, o4.z); rd = mk3(d4.x, d4.y, d4.z); }
        }
        Analytic a;
#ifdef PTK_DIAG_PAD_VALU     // timing experiment only (tools/diag/issue_sensitivity.sh): N more vector instructions per wave — four independent fma chains on live
        {                        // registers, no memory, no LDS — to read how the launch's time answers to its instruction count.  Same results.
            float p0 = ro.x, p1 = ro.y, p2 = rd.x, p3 = rd.y;
#pragma unroll
            for (int i = 0; i < PTK_DIAG_PAD_VALU / 4; ++i) {
                p0 = __builtin_fmaf(p0, p1, p2); p1 = __builtin_fmaf(p1, p2, p3); p2 = __builtin_fmaf(p2, p3, p0); p3 = __builtin_fmaf(p3, p0, p1);
                asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
            }
            asm volatile("" :: "v"(p0), "v"(p1), "v"(p2), "v"(p3));
        }
#endif
        if (gs.on) analytic_closest_lds<NT, true, LP, false, R3>(S, gs, L, active, ro, rd, a);
        else { a.t_min = FLT_MAX; a.geom = -1; a.P = mk3(0, 0, 0); a.NS = mk3(0, 0, 0); if (active) analytic_closest(S, gs, ro, rd, a); }
        if (!GEN) PTK_PHASE(0);                      // 4: (shadow phase +) closest hit: candidates, pair list, exact tests
        int flag = 0;
        bool heavy = false;
        const int slot = b * NT + tid;
        Hit h;
        h.t = -1.f; h.n = mk3(0, 0, 0); h.p = mk3(0, 0, 0); h.uv.x = h.uv.y = 0.f; h.mat = 0; h.geom = -1; h.tri = -1;
        if (active) {
            flag = a.geom >= 0;
            int wtri = -1;
            if (!SPLIT && !COMPACT && S.n_tris > 0) {
                // tiny mesh (the launcher sends meshes of up to kInlineBvhMaxTris triangles here): the reference's own walk, in this lane,
                // seeded with the analytic winner (pathtrace.cu:236-279) — no work list, no BVH stage, one launch per bounce
                Walk w;
                walk_begin(S, ro, rd, a.t_min, w);
                while (w.idx != -1) (void)walk_step(S, w);
                if (w.tri >= 0) { triangle_record(S, w.tri, w.u, w.v, w.t_min, h); wtri = w.tri; flag = 1; }
            }
            if (flag && wtri < 0) { if (gs.on) analytic_record_staged(gs.xf, a, h); else analytic_record(S, a, h); }
#ifdef PTK_DIAG_DOUBLE_RECORD    // timing experiments only
            if (flag && gs.on) { Analytic a2 = a; asm volatile("" : "+v"(a2.NS.x), "+v"(a2.NS.y), "+v"(a2.NS.z)); Hit h2; analytic_record_staged(gs.xf, a2, h2);
                                 asm volatile("" :: "v"(h2.n.x), "v"(h2.n.y), "v"(h2.n.z)); }
#endif
            if (SPLIT) {                       // root box test of this ray's direction tree (the first iteration of pathtrace.cu:249-256)
                const DevNode root = S.nodes[0];                       // (the same box in every direction copy)
                float tEnter;
                heavy = aabb_test(root, ro, rd, mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z), tEnter) && !(tEnter > a.t_min);
            }
            if (!COMPACT) {
#ifdef PTK_DIAG_EXTRA_TRAFFIC
                st4(io.hnxt.uv_ids, slot, make_float4(ro.x, ro.y, rd.x, rd.y));
#endif
                if (flag) {
                    st4(io.hnxt.t_n, slot, make_float4(h.t, h.n.x, h.n.y, h.n.z));
                    st4(io.hnxt.p_mat, slot, make_float4(h.p.x, h.p.y, h.p.z, __int_as_float(h.mat)));
                    if (!S.slim_hits) st4(io.hnxt.uv_ids, slot, make_float4(h.uv.x, h.uv.y, __int_as_float(h.geom), __int_as_float(h.tri)));
                } else {
                    st4(io.hnxt.t_n, slot, make_float4(-1.f, 0.f, 0.f, 0.f));
                }
            }
        }
        const unsigned long long hb = __ballot(flag != 0);
        if (lane == 0) s_red[wid] = __popcll(hb);
        __syncthreads();
        int before = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { const int c = s_red[w]; before += (w < wid) ? c : 0; hits += c; }
        if (SPLIT) {
            io.valid_nxt[slot] = active ? flag : -2;
            io.heavy_flag[slot] = heavy ? 1 : 0;
            const unsigned long long vb = __ballot(heavy);
            __syncthreads();
            if (lane == 0) s_red[wid] = __popcll(vb);
            __syncthreads();
            if (tid == 0) { int t = 0; for (int w = 0; w < NW; ++w) t += s_red[w]; io.heavy_agg[b] = t; }
        } else if (COMPACT) {
            // The normal's spare word of slot p carries TWO things: its sign says whether slot p itself missed (k_tail reads that when it takes
            // over after this launch), its low bits are the window's inverse map for hit-dense shading — hit number p of this window sits at
            // slot (word & 0xff), material type (word >> 8) & 0x7f — handed from the thread that owns that hit through LDS (the ds_min keys are
            // dead since the barrier above).  No stream of its own: the next launch's head reads the very lines it loads the normals from.
            const int rank_h = before + __popcll(hb & lt), type = flag ? S.mats[h.mat].type : 0;
            int* s_inv = reinterpret_cast<int*>(s_best);
            if (active && flag) s_inv[rank_h] = tid | (type << 8);
            __syncthreads();
            if (active) {
                const int word = (tid < hits ? s_inv[tid] : 0) | (flag ? 0 : (int)0x80000000u);
                st4(io.hnxt.t_n, slot, make_float4(h.n.x, h.n.y, h.n.z, __int_as_float(word)));
                if (flag) {
                    st4(io.hnxt.p_mat, slot, make_float4(h.p.x, h.p.y, h.p.z, __int_as_float(MODE == PTK_MODE_MIS ? h.geom : h.mat)));
                    if (GEN && gen_late) {                               // (thread tid generated and traced camera ray `slot`)
                        st4(io.nxt.o_pix, slot, make_float4(rd.x, rd.y, rd.z, __int_as_float(gen_pixw)));
                        st4(io.nxt.d_bnc, slot, make_float4(1.f, 1.f, 1.f, __int_as_float(gen.trace_depth)));
                        if (MODE == PTK_MODE_MIS) st4(io.nxt.col_pdf, slot, make_float4(gen.cam.pos.x, gen.cam.pos.y, gen.cam.pos.z, -1.f));
                    }
                }
            }
        } else {
            // (textures / ENV map / a tiny mesh walked in this kernel: the full-width streams.  hr: the slot's own status for k_env_miss2 and k_tail)
            // The hr word of slot p does double duty like the compact streams' spare word: bits 31-30 = the slot's own status (00 hit, 10 miss,
            // 11 empty: k_env_miss2 and k_tail read them), the low bits = the window's inverse map for hit-dense shading (hit number p of the
            // window sits at slot (word & 0xff), material type (word >> 8) & 0x7f), handed over through LDS.
            const int rank_h = before + __popcll(hb & lt), type = flag ? S.mats[h.mat].type : 0;
            int* s_inv = reinterpret_cast<int*>(s_best);
            if (active && flag) s_inv[rank_h] = tid | (type << 8);
            __syncthreads();
            io.hr_nxt[slot] = (tid < hits ? s_inv[tid] : 0) | (active ? (flag ? 0 : (int)0x80000000u) : (int)0xC0000000u);
        }
    } else if (SPLIT) {
        io.valid_nxt[b * NT + tid] = -2; io.heavy_flag[b * NT + tid] = 0;
        if (tid == 0) io.heavy_agg[b] = 0;
    } else if (!COMPACT) {
        io.hr_nxt[b * NT + tid] = -2;
    }
    if (!GEN && total > 0) PTK_PHASE(0);             // 5: hit record, inverse map, stores
    segments_tail2<NT>(tail, n, b, total, hits, io.agg_s, SPLIT ? nullptr : io.agg_h, cum_row(step, tail.ring_div, tail.cum_stride) + depth + 1);
    if (!GEN && total > 0) { PTK_PHASE(0); PTK_PHASE_END(0); }      // 6: window counts + done counters (+ the last block's tables)
    }
}

// ------------------------------------------------------------------------------------------------------------------
// k_shade_trace_w (round 5) — k_shade_trace's compact layout with WAVE WINDOWS: the unit of work is one wave and its own 64-slot window.
//
// Why it was built (VERDICT r4 #2b): by the counters (profiles/r5_pmc_wait.json) a wave of k_shade_trace spends 0.58 of its cycles waiting on data or at a
// barrier — 0.42 even in the camera launch, which has no loads at its head and no shading — and the block has ~18 barriers, each of which holds four waves (on
// four SIMDs, each shared with six other waves) for the slowest of them.  Here a block is four INDEPENDENT waves that share only the staged primitives and
// materials (one barrier at the start, one at the end for the window bookkeeping): wave w of block b shades hits [64 (4 b + w), + 64) of the previous bounce,
// keeps its survivors in window 4 b + w of the next buffer, and traces them — every former barrier is a compiler fence (a wave's LDS instructions execute in
// issue order), the counting sort, the survivor ranks and the inverse map are ballots and lane permutes, the pair list of the closest-hit search is balanced
// over the wave's 64 lanes.  Same streams, same tables (in units of 64 slots), same keys, same order of additions: bit-identical to k_shade_trace.
// What it measured (profiles/r5_ab_wave_windows.txt): the barrier wait does drop (-24 %) and comes back as waiting to ISSUE (+65 %) — the launch is bound by
// instruction issue, and this form needs 27 % more vector instructions: 19.3 against 21.0 Grays/s.  Opt-in (PTK_WAVE_WINDOWS=1); k_shade_trace stays the default.
// ------------------------------------------------------------------------------------------------------------------
#define PTK_WW 64
template <int MODE, bool GEN>
__global__ void __launch_bounds__(PTK_BLOCK, ((GEN && PTK_FUSED_WAVES_GEN > 0) ? PTK_FUSED_WAVES_GEN : (MODE == PTK_MODE_MIS ? PTK_FUSED_WAVES_MIS : (MODE == PTK_MODE_DIRECTLI ? PTK_FUSED_WAVES_DIRECT : (PTK_FUSED_WAVES_BSDF > 7 ? 7 : PTK_FUSED_WAVES_BSDF)))))      // (its LDS admits seven blocks per CU)
k_shade_trace_w(SceneView S, GenArgs gen, int iter_arg, const int* __restrict__ iter_ptr, int depth, int n_arg, const int* __restrict__ d_n, TraceIO io, FilmArgs film,
                int spp, int slot_bias, SegTail2 tail) {
    constexpr int NT = PTK_BLOCK, NW = NT / 64, WW = PTK_WW, NKEY = 6, MATS = 16, LP = 8;
    __shared__ DevMaterial s_mats[MATS];                            // shared by the block's waves (read-only after the first barrier)
    __shared__ float4 s_stage_all[NW][4 * WW];                      // per wave: rays (2 KB), winners' hit points + deferred normals (1.5 KB), winner keys (0.5 KB)
    __shared__ unsigned short s_list_all[NW][WW * LP];              // per wave: the pair list of its closest-hit / any-hit search
    __shared__ int s_tcnt[NW];
    __shared__ unsigned long long s_occ_all[NW];
    __shared__ int s_last;
    constexpr bool DEFER_NEE = !GEN && MODE != PTK_MODE_BSDF;

    const int step = iter_ptr ? *iter_ptr : iter_arg;
    const int n = GEN ? gen.n_pix * spp : (d_n ? *d_n : n_arg);      // hits of the bounce being shaded (GEN: rays to generate)
    const Film img = {film, depth};
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, b = blockIdx.x;
    if (b * NT >= n) {
        if (n <= 0 && b == 0) {
            if (tail.hit_pref && tid < 8) tail.hit_pref[tid] = 0;
            if (tail.hcount_out && tid == 0) *tail.hcount_out = 0;
            if (tail.cum_out && tid == 0) { const int at = cum_row(step, tail.ring_div, tail.cum_stride) + depth + 1; tail.cum_out[at] = cum_read(tail.cum_prev, 0, at); }
            segments_block<NT, false>(0, io.agg_s, tail.pref, tail.first, tail.count_out, tail.iter_advance, tail.iter_stride);
        }
        return;
    }
    const GeomStage gs = stage_geoms(S, false);
    const MatTable mt = stage_materials<NT, MATS>(S, s_mats);       // ends with the block's first barrier: primitives and materials are staged
    float4* s_ray = s_stage_all[wid];
    float2* s_res = reinterpret_cast<float2*>(s_ray + 2 * WW);
    unsigned long long* s_best = reinterpret_cast<unsigned long long*>(s_ray + 2 * WW + (3 * WW) / 2);
    // sorted position -> lane (before the shading phase), later hit rank -> lane | type << 8 (after the trace phase): in the wave's pair-list space, which
    // is only in use between the two (23.0 KB per block with the staged primitives of cornell.txt = seven blocks per CU; with a table of its own: six)
    int* s_tab = reinterpret_cast<int*>(s_list_all[wid]);
    const TraceLds L = {s_ray, s_res, s_best, s_list_all[wid], &s_tcnt[wid]};
    const int Wn = __builtin_amdgcn_readfirstlane(b * NW + wid);    // this wave's window
    const bool wave_live = Wn * WW < n;                            // (wave-uniform)
    const int i = Wn * WW + lane;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int total = 0, hits = 0;
    if (wave_live) {
        int alive = 0, my_rank = -1;
        ShadowReq rq; rq.on = false; rq.slot = s_ray + 2 * WW + lane; rq.dstride = WW; rq.pix = 0; rq.contrib = mk3(0, 0, 0);
        int gen_pixw = 0;
        const bool gen_late = GEN && gen.late_stores != 0;
        if (DEFER_NEE && lane == 0) s_occ_all[wid] = 0ull;
        if (GEN) {
            total = min(WW, n - Wn * WW);
            if (i < n) {
                alive = 1; my_rank = lane;
                const int n_pix = gen.n_pix;
                const int sidx = i / n_pix, jp = i - sidx * n_pix;
                const int iter = (step - 1) * spp + sidx + 1;
                const int lr = jp / gen.cam.W, x = jp - lr * gen.cam.W;
                const int y = ((lr / gen.tile_rows) * gen.n_parts + gen.part) * gen.tile_rows + (lr % gen.tile_rows);
                Rng rng = make_rng(x, y, iter);                       // argument order of pathtrace.cu:148: (iter:=x, index:=y, depth:=iter)
                const f2 r = sample2D(rng);
                const float fx = static_cast<float>(static_cast<float>(x) + ((double)r.x - 0.5) - gen.cam.W * 0.5f);
                const float fy = static_cast<float>(static_cast<float>(y) + ((double)r.y - 0.5) - gen.cam.H * 0.5f);
                const f3 d = normalize(gen.cam.view - gen.cam.right * gen.cam.plx * fx - gen.cam.up * gen.cam.ply * fy);
                gen_pixw = (sidx << 24) | (x + y * gen.cam.W);
                if (!gen_late) {
                    st4(io.nxt.o_pix, i, make_float4(d.x, d.y, d.z, __int_as_float(gen_pixw)));
                    st4(io.nxt.d_bnc, i, make_float4(1.f, 1.f, 1.f, __int_as_float(gen.trace_depth)));
                    if (MODE == PTK_MODE_MIS) st4(io.nxt.col_pdf, i, make_float4(gen.cam.pos.x, gen.cam.pos.y, gen.cam.pos.z, -1.f));
                }
                s_ray[lane] = make_float4(gen.cam.pos.x, gen.cam.pos.y, gen.cam.pos.z, 0.f);
                s_ray[WW + lane] = make_float4(d.x, d.y, d.z, 0.f);
            }
        } else {
            // ---- shading phase, hit-dense: lane shades hit i of the previous bounce; its compacted slot — the reference's RNG key — is i itself
            const int iter = (step - 1) * spp + 1;
            const int slot_off = slot_bias + cum_read(tail.cum_prev, cum_row(step, tail.ring_div, tail.cum_stride), depth);
            const int nw_prev = depth > 0 ? (d_n[-1] + WW - 1) / WW : (gen.n_pix * spp + WW - 1) / WW;
            const int w0 = __builtin_amdgcn_readfirstlane(io.first_h[Wn]);
            const int h0 = io.hit_pref[w0], h1 = io.hit_pref[w0 + 1], h2 = io.hit_pref[w0 + 2], h3 = io.hit_pref[w0 + 3], h4 = io.hit_pref[w0 + 4];
            int src_mine = 0, key = NKEY - 1, slot_mine = 0;
            if (i < n) {
                int k = (i >= h1) + (i >= h2) + (i >= h3) + (i >= h4);
                int base = k == 0 ? h0 : (k == 1 ? h1 : (k == 2 ? h2 : (k == 3 ? h3 : h4)));
                if (k == 4) k = sparse_window(io.hit_pref, w0 + 4, nw_prev, i, base) - w0;
                const int at = (w0 + k) * WW + (i - base);
                const int word = __float_as_int(io.hcur.t_n[4 * (size_t)at + 3]) & 0x3fffffff;
                src_mine = (w0 + k) * WW + (word & 0xff);
                key = word >> 8; slot_mine = i + slot_off;
            }
            // counting sort by material type inside the wave: ballots; sorted position -> lane through the wave's table
            int rank = 0, n_active = 0;
#pragma unroll
            for (int k = 0; k < NKEY; ++k) {
                const unsigned long long bk = __ballot(key == k);
                const int tot = __popcll(bk);
                if (k < NKEY - 1) n_active += tot;
                if (key == k) rank += __popcll(bk & lt);
                else if (key > k) rank += tot;
            }
            s_tab[rank] = lane;
            wave_sync();
            const int pj = s_tab[lane];                              // the lane whose path this thread shades (its position in the original order)
            const int slot = __shfl(slot_mine, pj, 64), ps = __shfl(src_mine, pj, 64);
            PathRegs p;
            p.o = mk3(0, 0, 0); p.d = mk3(0, 0, 0); p.col = mk3(0, 0, 0); p.pix = 0; p.bounces = 0; p.prevPdf = 0.f;
            if (lane < n_active) {
                Hit h;
                const float4 q0 = ld4(io.hcur.t_n, ps), q1 = ld4(io.hcur.p_mat, ps), a4 = ld4(io.cur.o_pix, ps), b4 = ld4(io.cur.d_bnc, ps);
                h.t = 1.0f; h.n = mk3(q0.x, q0.y, q0.z); h.p = mk3(q1.x, q1.y, q1.z); h.mat = __float_as_int(q1.w);
                h.uv.x = 0.f; h.uv.y = 0.f; h.geom = -1; h.tri = -1;
                p.pix = __float_as_int(a4.w);
                p.d = mk3(a4.x, a4.y, a4.z); p.bounces = __float_as_int(b4.w);
                p.col = mk3(b4.x, b4.y, b4.z); p.prevPdf = -1.f;
                if (MODE == PTK_MODE_MIS) {
                    const float4 c4 = ld4(io.cur.col_pdf, ps);
                    p.o = mk3(c4.x, c4.y, c4.z); p.prevPdf = c4.w;
                    h.geom = __float_as_int(q1.w);
                    h.mat = gs.on ? __float_as_int(gs.xf[(size_t)h.geom * PTK_XF_FLOATS + 34]) : S.geoms[h.geom].materialid;
                }
                rq.pix = p.pix;
                alive = shade_path<MODE>(S, gs, mt, iter, slot, depth, h, p, img, DEFER_NEE ? &rq : nullptr);
            }
            // survivor rank in the ORIGINAL (dense) order — the next bounce's compacted order, hence its RNG keys: the path of original position L was
            // shaded by the thread at sorted position rank(L)
            const int alive_orig = __shfl(alive, rank, 64);
            const unsigned long long m = __ballot(alive_orig != 0);
            total = __popcll(m);
            if (alive) {
                my_rank = __popcll(m & ((1ull << pj) - 1ull));
                const int s2 = Wn * WW + my_rank;
                st4(io.nxt.o_pix, s2, make_float4(p.d.x, p.d.y, p.d.z, __int_as_float(p.pix)));
                st4(io.nxt.d_bnc, s2, make_float4(p.col.x, p.col.y, p.col.z, __int_as_float(p.bounces)));
                if (MODE == PTK_MODE_MIS) st4(io.nxt.col_pdf, s2, make_float4(p.o.x, p.o.y, p.o.z, p.prevPdf));
                s_ray[my_rank] = make_float4(p.o.x, p.o.y, p.o.z, __int_as_float(p.pix));          // hand the new ray to the lane that traces it
                s_ray[WW + my_rank] = make_float4(p.d.x, p.d.y, p.d.z, __int_as_float(p.bounces));
            }
        }
        wave_sync();
        if (DEFER_NEE && __ballot(rq.on) != 0ull) {                  // (wave-uniform) shadow phase of the wave's next-event samples (lightSample.h:59-139)
            const TraceLds Lsh = {s_ray + 2 * WW, nullptr, nullptr, s_list_all[wid], &s_tcnt[wid]};
            f3 so = mk3(0, 0, 0), sd = mk3(0, 0, 0); int s_pix = 0;
            if (rq.on) { const float4 o4 = rq.slot[0], d4 = rq.slot[WW]; so = mk3(o4.x, o4.y, o4.z); sd = mk3(d4.x, d4.y, d4.z); s_pix = __float_as_int(d4.w); }
            analytic_anyhit_lds<WW, LP, true>(S, gs, Lsh, &s_occ_all[wid], rq.on, so, sd);
            const bool occ = (s_occ_all[wid] >> lane) & 1ull;
            if (rq.on && !occ) add_radiance(img, s_pix, rq.contrib);
            wave_sync();
        }
        // ---- trace phase: lane t < total traces survivor t of the window
        if (total > 0) {                                             // (wave-uniform)
            const bool active = lane < total;
            f3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 0);
            if (active) { const float4 o4 = s_ray[lane], d4 = s_ray[WW + lane]; ro = mk3(o4.x, o4.y, o4.z); rd = mk3(d4.x, d4.y, d4.z); }
            Analytic a;
            analytic_closest_lds<WW, true, LP, true>(S, gs, L, active, ro, rd, a);
            const int flag = active && a.geom >= 0;
            const int slot = Wn * WW + lane;
            Hit h;
            h.t = -1.f; h.n = mk3(0, 0, 0); h.p = mk3(0, 0, 0); h.uv.x = h.uv.y = 0.f; h.mat = 0; h.geom = -1; h.tri = -1;
            if (flag) analytic_record_staged(gs.xf, a, h);
            const unsigned long long hb = __ballot(flag != 0);
            hits = __popcll(hb);
            // the normal's spare word of slot p: sign = slot p itself missed; low bits = the window's inverse map for hit-dense shading (hit number p of
            // this window sits at slot (word & 0xff), material type (word >> 8) & 0x7f)
            const int rank_h = __popcll(hb & lt), type = flag ? S.mats[h.mat].type : 0;
            if (flag) s_tab[rank_h] = lane | (type << 8);
            wave_sync();
            if (active) {
                const int word = (lane < hits ? s_tab[lane] : 0) | (flag ? 0 : (int)0x80000000u);
                st4(io.hnxt.t_n, slot, make_float4(h.n.x, h.n.y, h.n.z, __int_as_float(word)));
                if (flag) {
                    st4(io.hnxt.p_mat, slot, make_float4(h.p.x, h.p.y, h.p.z, __int_as_float(MODE == PTK_MODE_MIS ? h.geom : h.mat)));
                    if (GEN && gen_late) {
                        st4(io.nxt.o_pix, slot, make_float4(rd.x, rd.y, rd.z, __int_as_float(gen_pixw)));
                        st4(io.nxt.d_bnc, slot, make_float4(1.f, 1.f, 1.f, __int_as_float(gen.trace_depth)));
                        if (MODE == PTK_MODE_MIS) st4(io.nxt.col_pdf, slot, make_float4(gen.cam.pos.x, gen.cam.pos.y, gen.cam.pos.z, -1.f));
                    }
                }
            }
        }
        if (lane == 0) {                                             // this window's survivor and hit counts
            __hip_atomic_store(io.agg_s + Wn, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(io.agg_h + Wn, hits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();                                                 // the block's windows are published: the only other barrier of the block
    // the last block of the launch builds the window tables (as segments_tail2, in units of 64-slot windows)
    const int nb = (n + NT - 1) / NT, ng = (nb + PTK_SEG_GROUP - 1) / PTK_SEG_GROUP;
    if (tid == 0) {
        const int grp = b / PTK_SEG_GROUP;
        const int gsize = min(PTK_SEG_GROUP, nb - grp * PTK_SEG_GROUP);
        int last = 0;
        if (__hip_atomic_fetch_add(tail.done + 32 * (grp + 1), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1)
            last = __hip_atomic_fetch_add(tail.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ng - 1;
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    for (int k = tid; k <= ng; k += NT) __hip_atomic_store(tail.done + 32 * k, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    segments_block2<NT, WW, 4>(n, io.agg_s, io.agg_h, tail.pref, tail.first, tail.hit_pref, tail.count_out, tail.iter_advance, tail.iter_stride,
                            tail.first_h, tail.hcount_out, tail.cum_prev, tail.cum_out, cum_row(step, tail.ring_div, tail.cum_stride) + depth + 1);
    if (tid == 0 && tail.ray_total) atomicAdd(tail.ray_total, (unsigned long long)*tail.count_out);
}

// ------------------------------------------------------------------------------------------------------------------
// k_trace_shade (round 5, pipeline 11) — TRACE bounce d and SHADE bounce d in one launch, for triangle-free scenes.
//
// k_shade_trace fuses the other way round (shade d-1, trace d) because the reference keys the shading RNG by the COMPACTED slot of a hit path
// (pathtrace.cu:458,532): the rank of the hit among ALL hits of the bounce, which a block only knows when every earlier block has counted its
// hits — a launch boundary gives that for free, and costs the hit record a round trip through HBM: 32 B written and 32 B read per path-bounce
// of the 128 B the compact streams move, plus the inverse-map word and two dependent loads at the head of every block.  Here the record never
// leaves the block: the block traces its 256 rays, counts its hits, publishes the count and obtains the number of hits in all earlier blocks by
// a DECOUPLED LOOK-BACK (Merrill & Garland, "Single-pass parallel prefix scan with decoupled look-back", 2016): one 64-bit word per block,
// {launch tag : 32 | status : 2 | value : 30}; status 1 = the block's own count, 2 = the inclusive prefix; wave 0 inspects 64 predecessors per
// round and sums counts back to the first inclusive prefix.  A block only ever waits for blocks with a SMALLER index; each XCD hands out its share of
// the workgroups in index order, so with the GPU to itself the lowest unfinished block of a launch is always resident and the wait cannot deadlock.
// Beside other launches that wait the same way it CAN (the XCDs dispatch independently: two launches can fill the XCDs each other's next block needs —
// observed as a hang with four iterations in flight), so the render loop uses this kernel with one iteration at a time only (pathtrace.hip); the tag (iteration x 256 + bounce + 1, unique per
// launch of a lane) makes stale words of earlier launches invisible without a clear.  (Rounds 1-2 tried the CHAINED form — every block waits for
// its predecessor's prefix, pipeline 3 — and lost: 11 ns x 2 500 blocks in series.  Look-back has no chain: a block adds up counts that are
// published before any prefix is.)
//   streams: the reference's PathSegment as three 16-byte records (origin, pixel) (direction, bounces) (colour, prevPdf) — 48 B read + 48 B
//   written per surviving path-bounce (k_shade_trace's compact layout: 64 + 64), in the staged pipeline's segmented windows (Gather), so that
//   k_tail takes over unchanged; one launch per bounce, traceDepth per iteration (k_shade_trace: traceDepth + 1).
//   phases: ray (camera ray or gathered path) -> LDS | closest hit, block-cooperative (analytic_closest_lds) | hit count, publish, look-back
//   (wave 0) beside the counting sort by material type of the others | thread j shades the j-th hit of the sorted order: ray, winner key, hit
//   point and deferred normal come from LDS where the trace phase left them, colour / prevPdf from HBM | [MIS / DIRECT: block-cooperative
//   shadow phase] | survivor ranks, stores into the block's window, window bookkeeping by the last block (segments_tail).
// Same paths, same keys, same order of additions into every pixel as every other pipeline: bit-identical images (tests/test_gpu_parity.py).
// ------------------------------------------------------------------------------------------------------------------
// (Measured and dropped, profiles/r5_ab_trace_shade.txt: a two-level form — one accumulating word per group of 32 blocks, every block sums the
// complete groups before its own and its group's earlier blocks — removes the rounds but makes every waiting block poll the SAME few words:
// 6.8 instead of 16.7 Grays/s.  The flat form's polls go to the block's own predecessors only.)
// One round inspects PTK_LB_SPAN x 64 predecessors: every lane of wave 0 reads PTK_LB_SPAN words that are 64 blocks apart (loads issued together:
// one round trip), so that the first blocks of a launch — nobody has an inclusive prefix yet, all reach the count at about the same time — need
// 1 792 / 256 = 7 rounds at worst instead of 28.
#ifndef PTK_LB_SPAN
#define PTK_LB_SPAN 4
#endif
// EXIT CONDITION of the wait: a block that has looked PTK look-back `limit` times at a predecessor that still has not published gives up — it raises
// *gave_up, takes the missing counts as zero and publishes an inclusive prefix, so that no later block waits for it.  The launch then drains with wrong
// keys and the host reports PTK_E_DEVICE at its next read (pathtrace.hip: lookback_status).  A legitimate wait is bounded by the duration of the launch
// (< 1 ms); the default limit is about a second of polling.  This turns the cross-launch deadlock described at k_trace_shade (two processes sharing a
// GPU, say) from a hung GPU into an error.
struct LookBack { unsigned long long* words; unsigned long long* gave_up; int limit; };       // words: [blocks + 1] per lane
__device__ __forceinline__ unsigned long long lb_word(unsigned tag, unsigned status, unsigned value) { return ((unsigned long long)tag << 32) | ((unsigned long long)status << 30) | value; }
// lane 0 of a wave of block b, as soon as the block's count is known (block 0: that is its inclusive prefix already)
__device__ __forceinline__ void lookback_publish(const LookBack& lb, unsigned tag, int b, int count) {
    __hip_atomic_store(lb.words + b, lb_word(tag, b == 0 ? 2u : 1u, (unsigned)count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wave 0 of block b (b > 0 or not): the number of hits in blocks [0, b); publishes the block's inclusive prefix
__device__ __forceinline__ int lookback_exclusive(const LookBack& lb, unsigned tag, int b, int count, int lane) {
    if (b == 0) return 0;
    int excl = 0, polls = 0;
    for (int j = b - 1; ; ) {
        unsigned long long w[PTK_LB_SPAN];
        unsigned st[PTK_LB_SPAN];
        int first_incl = 64 * PTK_LB_SPAN, first_missing = 64 * PTK_LB_SPAN;      // positions (distance from block j) over the whole span
#pragma unroll
        for (int k = 0; k < PTK_LB_SPAN; ++k) {
            const int idx = j - 64 * k - lane;
            w[k] = lb_word(tag, 2u, 0u);                             // (virtual blocks before block 0: inclusive prefix 0)
            if (idx >= 0) w[k] = __hip_atomic_load(lb.words + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int k = PTK_LB_SPAN - 1; k >= 0; --k) {
            st[k] = ((unsigned)(w[k] >> 32) == tag) ? ((unsigned)w[k] >> 30) : 0u;
            const unsigned long long incl = __ballot(st[k] == 2u), missing = __ballot(st[k] == 0u);
            if (incl) first_incl = 64 * k + __builtin_ctzll(incl);
            if (missing) first_missing = 64 * k + __builtin_ctzll(missing);
        }
        if (first_missing < first_incl) {                         // a predecessor this round needs has not published yet
            if (polls++ < lb.limit) { __builtin_amdgcn_s_sleep(4); continue; }
            if (lane == 0) atomicOr(lb.gave_up, 1ull);            // (wave-uniform: the ballots above are)
            break;
        }
        int v = 0;
#pragma unroll
        for (int k = 0; k < PTK_LB_SPAN; ++k) v += (64 * k + lane <= first_incl) ? (int)((unsigned)w[k] & 0x3fffffffu) : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        excl += v;
        if (first_incl < 64 * PTK_LB_SPAN) break;
        j -= 64 * PTK_LB_SPAN;
    }
    if (lane == 0) __hip_atomic_store(lb.words + b, lb_word(tag, 2u, (unsigned)(excl + count)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}

#ifndef PTK_TS_WAVES_BSDF
#define PTK_TS_WAVES_BSDF 7
#endif
#ifndef PTK_TS_WAVES_NEE
#define PTK_TS_WAVES_NEE 6
#endif
template <int MODE, bool GEN>
__global__ void __launch_bounds__(PTK_BLOCK, (MODE == PTK_MODE_BSDF ? PTK_TS_WAVES_BSDF : PTK_TS_WAVES_NEE))
k_trace_shade(SceneView S, GenArgs gen, int iter_arg, const int* __restrict__ iter_ptr, int depth, int n_arg, const int* __restrict__ d_n,
              PtkPaths cur, Gather G, PtkPaths nxt, int* __restrict__ agg_s, LookBack lb, FilmArgs film, int spp, int slot_bias, SegTail tail,
              int* __restrict__ count0, unsigned long long* __restrict__ ray_total) {
    constexpr int NT = PTK_BLOCK, NW = NT / 64, NKEY = 6, MATS = 16, LP = 8;
    constexpr int OVER = (MATS * (int)sizeof(DevMaterial) + 3 * NT * (int)sizeof(int)) > NT * LP * 2 ? (MATS * (int)sizeof(DevMaterial) + 3 * NT * (int)sizeof(int)) : NT * LP * 2;
    __shared__ __attribute__((aligned(16))) unsigned char s_over[OVER];      // trace phase: pair list; afterwards: materials + sort tables
    __shared__ float4 s_stage[4 * NT];                                       // rays (8 KB), winners' hit points + deferred normals (6 KB), winner keys (2 KB)
    float4* s_ray = s_stage;
    float2* s_res = reinterpret_cast<float2*>(s_stage + 2 * NT);
    unsigned long long* s_best = reinterpret_cast<unsigned long long*>(s_stage + 2 * NT + (3 * NT) / 2);
    __shared__ int s_cnt[NW][NKEY];
    __shared__ int s_red[NW];
    __shared__ int s_tcnt[NW];
    __shared__ unsigned long long s_alive[NT / 64];
    constexpr bool NEE = MODE != PTK_MODE_BSDF;
    __shared__ unsigned long long s_occ[NT / 64];
    __shared__ int s_anyreq;
    __shared__ int s_base;
    DevMaterial* s_mats = reinterpret_cast<DevMaterial*>(s_over);
    int* s_perm = reinterpret_cast<int*>(s_over + MATS * sizeof(DevMaterial));
    int* s_hrank = s_perm + NT;
    int* s_src = s_hrank + NT;
    const TraceLds L = {s_ray, s_res, s_best, reinterpret_cast<unsigned short*>(s_over), s_tcnt};

    const int step = iter_ptr ? *iter_ptr : iter_arg;
    const int n = GEN ? gen.n_pix * spp : (d_n ? *d_n : n_arg);      // rays of this bounce
    const Film img = {film, depth};
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, b = blockIdx.x;
    if (b == 0 && tid == 0) {
        if (GEN && count0) *count0 = n;
        if (ray_total && n > 0) atomicAdd(ray_total, (unsigned long long)n);
    }
    if (b * NT >= n) {
        if (n <= 0 && b == 0) segments_block<NT, false>(0, agg_s, tail.pref, tail.first, tail.count_out, tail.iter_advance, tail.iter_stride);
        return;
    }
    const GeomStage gs = stage_geoms(S, false);          // published by the first barrier of the trace phase
    if (tid < NT / 64) { s_alive[tid] = 0ull; if (NEE) { s_occ[tid] = 0ull; s_anyreq = 0; } }
    const int i = b * NT + tid;
    const bool active = i < n;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // ---- the ray of dense path i: generateRayFromCamera (pathtrace.cu:135-163), or the survivor the previous launch left in its window
    int src_mine = i;
    f3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 0);
    int pixw = 0, bounces = 0;
    if (GEN) {
        if (active) {
            const int n_pix = gen.n_pix;
            const int sidx = i / n_pix, jp = i - sidx * n_pix;
            const int iter = (step - 1) * spp + sidx + 1;
            const int lr = jp / gen.cam.W, x = jp - lr * gen.cam.W;
            const int y = ((lr / gen.tile_rows) * gen.n_parts + gen.part) * gen.tile_rows + (lr % gen.tile_rows);
            Rng rng = make_rng(x, y, iter);                       // argument order of pathtrace.cu:148: (iter:=x, index:=y, depth:=iter)
            const f2 r = sample2D(rng);
            const float fx = static_cast<float>(static_cast<float>(x) + ((double)r.x - 0.5) - gen.cam.W * 0.5f);
            const float fy = static_cast<float>(static_cast<float>(y) + ((double)r.y - 0.5) - gen.cam.H * 0.5f);
            rd = normalize(gen.cam.view - gen.cam.right * gen.cam.plx * fx - gen.cam.up * gen.cam.ply * fy);
            ro = gen.cam.pos;
            pixw = (sidx << 24) | (x + y * gen.cam.W); bounces = gen.trace_depth;
        }
    } else {
        const int nw_prev = (d_n[-1] + NT - 1) / NT;            // windows of the stream being read = ceil(paths the previous launch shaded / 256)
        src_mine = gather_source(G, b, i, n, nw_prev);          // 0 for i >= n
        if (active) {
            const float4 o4 = ld4(cur.o_pix, src_mine), d4 = ld4(cur.d_bnc, src_mine);
            ro = mk3(o4.x, o4.y, o4.z); pixw = __float_as_int(o4.w);
            rd = mk3(d4.x, d4.y, d4.z); bounces = __float_as_int(d4.w);
        }
    }
    s_ray[tid] = make_float4(ro.x, ro.y, ro.z, __int_as_float(pixw));
    s_ray[NT + tid] = make_float4(rd.x, rd.y, rd.z, __int_as_float(bounces));
    // ---- trace phase (computeIntersections, pathtrace.cu:169-346): the winner's key, hit point and deferred normal stay in LDS
    Analytic a;
    analytic_closest_lds<NT, true, LP>(S, gs, L, active, ro, rd, a);
    const int flag = active && a.geom >= 0;
    const unsigned long long hb = __ballot(flag != 0);
    if (lane == 0) s_red[wid] = __popcll(hb);
    // ---- the pair list is dead: materials -> LDS (ends with a barrier, which also publishes s_red)
    const MatTable mt = stage_materials<NT, MATS>(S, s_mats);
    int hits = 0, before = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { const int c = s_red[w]; before += (w < wid) ? c : 0; hits += c; }
    const int iter = (step - 1) * spp + 1;
    // ---- this block's hit count -> the look-back tables (every wave fetches the prefix itself, as late as possible: just before shading)
    const unsigned lb_tag = (unsigned)step * 256u + (unsigned)depth + 1u;
    if (tid == 0) lookback_publish(lb, lb_tag, b, hits);          // as early as possible: later blocks wait for it
    int key = NKEY - 1;
    if (flag) key = fetch_material_type(S, mt, __float_as_int(gs.xf[(size_t)a.geom * PTK_XF_FLOATS + 34]));
    unsigned long long bal[NKEY];
#pragma unroll
    for (int k = 0; k < NKEY; ++k) bal[k] = __ballot(key == k);
    if (lane < NKEY) s_cnt[wid][lane] = __popcll(bal[lane]);
    s_src[tid] = src_mine;
    s_hrank[tid] = before + __popcll(hb & lt);              // rank among the block's hits, in stream order: + s_base = the reference's compacted slot
    __syncthreads();
    int rank = 0, n_active = 0;
#pragma unroll
    for (int k = 0; k < NKEY; ++k) {
        int before_waves = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { const int c = s_cnt[w][k]; before_waves += (w < wid) ? c : 0; tot += c; }
        if (k < NKEY - 1) n_active += tot;
        if (key == k) rank += before_waves + __popcll(bal[k] & lt);
        else if (key > k) rank += tot;
    }
    s_perm[rank] = tid;
    if (wid == 0) {                            // wave 0: hits of all earlier blocks (the barrier below publishes s_base); the count went out before the sort
#ifdef PTK_DIAG_NO_LOOKBACK                    // timing experiments only (wrong keys): what the launch costs without the wait for the earlier blocks' counts
        const int e = b * 200;
#else
        const int e = lookback_exclusive(lb, lb_tag, b, hits, lane);
#endif
        if (lane == 0) s_base = e;
    }
    __syncthreads();
    // ---- shading phase: thread j shades the j-th hit of the sorted order
    int alive = 0, src = -1;
    PathRegs p;
    p.o = mk3(0, 0, 0); p.d = mk3(0, 0, 0); p.col = mk3(0, 0, 0); p.pix = 0; p.bounces = 0; p.prevPdf = 0.f;
    Hit h;
    h.t = -1.f; h.n = mk3(0, 0, 0); h.p = mk3(0, 0, 0); h.uv.x = h.uv.y = 0.f; h.mat = 0; h.geom = -1; h.tri = -1;
    int slot = 0;
    const bool shading = tid < n_active;
    if (shading) {
        src = s_perm[tid];
        slot = s_hrank[src] + slot_bias;
        const float4 o4 = s_ray[src], d4 = s_ray[NT + src];
        const unsigned long long best = s_best[src];
        const float2 r0 = s_res[3 * src], r1 = s_res[3 * src + 1], r2 = s_res[3 * src + 2];
        Analytic w;
        w.t_min = __uint_as_float((unsigned)(best >> 32)); w.geom = (int)(unsigned)best;
        w.P = mk3(r0.x, r0.y, r1.x); w.NS = mk3(r1.y, r2.x, r2.y);
        analytic_record_staged(gs.xf, w, h);
        p.o = mk3(o4.x, o4.y, o4.z); p.pix = __float_as_int(o4.w);
        p.d = mk3(d4.x, d4.y, d4.z); p.bounces = __float_as_int(d4.w);
        if (GEN) { p.col = mk3(1.f, 1.f, 1.f); p.prevPdf = -1.f; }
        else { const float4 c4 = ld4(cur.col_pdf, s_src[src]); p.col = mk3(c4.x, c4.y, c4.z); p.prevPdf = c4.w; }
    }
    ShadowReq rq; rq.on = false; rq.slot = s_stage + 2 * NT + tid; rq.dstride = NT; rq.pix = p.pix; rq.contrib = mk3(0, 0, 0);
    if (NEE) __syncthreads();                  // every shading thread has its inputs in registers: the upper half of s_stage may take the shadow rays
    slot += s_base;                            // hits of all earlier blocks: the reference's compacted slot = that + the rank among the block's hits
    if (shading) {
        alive = shade_path<MODE>(S, gs, mt, iter, slot, depth, h, p, img, NEE ? &rq : nullptr);
        if (NEE && rq.on) s_anyreq = 1;
    }
    // survivor rank in the ORIGINAL (dense) order — the next bounce's compacted order, hence its RNG keys
    if (alive) atomicOr(&s_alive[src >> 6], 1ull << (src & 63));
    __syncthreads();
    int total = 0, my_rank = -1;
    {
        const unsigned long long m0 = s_alive[0], m1 = s_alive[1], m2 = s_alive[2], m3 = s_alive[3];
        const int c0 = __popcll(m0), c1 = __popcll(m1), c2 = __popcll(m2);
        total = c0 + c1 + c2 + __popcll(m3);
        if (alive) {
            const int w = src >> 6;
            const unsigned long long mw = w == 0 ? m0 : (w == 1 ? m1 : (w == 2 ? m2 : m3));
            my_rank = (w > 0 ? c0 : 0) + (w > 1 ? c1 : 0) + (w > 2 ? c2 : 0) + __popcll(mw & ((1ull << (src & 63)) - 1ull));
        }
    }
    if (alive) {
        const int s2 = b * NT + my_rank;
        st4(nxt.o_pix, s2, make_float4(p.o.x, p.o.y, p.o.z, __int_as_float(p.pix)));
        st4(nxt.d_bnc, s2, make_float4(p.d.x, p.d.y, p.d.z, __int_as_float(p.bounces)));
        st4(nxt.col_pdf, s2, make_float4(p.col.x, p.col.y, p.col.z, p.prevPdf));
    }
    if (NEE && s_anyreq) {                     // block-uniform (published by the barrier above)
        // ---- shadow phase (LightSampler::occulusionTest, lightSample.h:59-139) for the block's next-event samples, block-cooperatively
        const TraceLds Lsh = {s_stage + 2 * NT, nullptr, nullptr, reinterpret_cast<unsigned short*>(s_over + MATS * sizeof(DevMaterial)), s_tcnt};
        f3 so = mk3(0, 0, 0), sd = mk3(0, 0, 0); int s_pix = 0;
        if (rq.on) { const float4 o4 = rq.slot[0], d4 = rq.slot[NT]; so = mk3(o4.x, o4.y, o4.z); sd = mk3(d4.x, d4.y, d4.z); s_pix = __float_as_int(d4.w); }
        analytic_anyhit_lds<NT, LP>(S, gs, Lsh, s_occ, rq.on, so, sd);
        const bool occ = (s_occ[tid >> 6] >> (tid & 63)) & 1ull;
        if (rq.on && !occ) add_radiance(img, s_pix, rq.contrib);
    }
    // publish the window's survivor count; the last block of the launch builds the window tables
    segments_tail<NT>(tail, n, b, total, agg_s);
}

// ------------------------------------------------------------------------------------------------------------------
// compact_rays, pathtrace.cu:614-631: k_count (per-block aggregates) + k_scatter (stable scatter, device count out)
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(PTK_BLOCK) k_count(int n_arg, const int* __restrict__ d_n, const int* __restrict__ valid, int* __restrict__ agg) {
    const int n = d_n ? *d_n : n_arg;
    const int b = blockIdx.x;
    if (b * PTK_BLOCK >= n) return;
    const int i = b * PTK_BLOCK + threadIdx.x;
    const int flag = (i < n) ? valid[i] : 0;
    int total;
    (void)block_excl_prefix(flag, total);
    if (threadIdx.x == 0) agg[b] = total;
}

__global__ void __launch_bounds__(PTK_BLOCK) k_scatter(int n_arg, const int* __restrict__ d_n, const int* __restrict__ valid,
                                                        const int* __restrict__ agg, PtkPaths pin, PtkHits hin, PtkPaths pout,
                                                        PtkHits hout, int* __restrict__ count_out) {
    const int n = d_n ? *d_n : n_arg;
    const int b = blockIdx.x;
    if (b * PTK_BLOCK >= n) { if (n <= 0 && b == 0 && threadIdx.x == 0 && count_out) *count_out = 0; return; }
    const int i = b * PTK_BLOCK + threadIdx.x;
    const int flag = (i < n) ? valid[i] : 0;
    int total;
    const int local = block_excl_prefix(flag, total);
    const int prior = prior_blocks(agg, b);
    if (flag) {
        const int s = prior + local;
        if (pin.o_pix) { st4(pout.o_pix, s, ld4(pin.o_pix, i)); st4(pout.d_bnc, s, ld4(pin.d_bnc, i)); st4(pout.col_pdf, s, ld4(pin.col_pdf, i)); }
        if (hin.t_n) { st4(hout.t_n, s, ld4(hin.t_n, i)); st4(hout.p_mat, s, ld4(hin.p_mat, i)); st4(hout.uv_ids, s, ld4(hin.uv_ids, i)); }
    }
    if (count_out && threadIdx.x == 0 && b == (n - 1) / PTK_BLOCK) *count_out = prior + total;
}

template <int NT>
__device__ __forceinline__ int block_excl_prefix_n(int flag, int& total) {
    constexpr int NW = NT / 64;
    __shared__ int wsum[NW];
    const unsigned long long bal = __ballot(flag != 0);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int within = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wid] = __popcll(bal);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { const int c = wsum[w]; off += (w < wid) ? c : 0; tot += c; }
    __syncthreads();
    total = tot;
    return off + within;
}

// ------------------------------------------------------------------------------------------------------------------
// k_tail: ALL remaining bounces of an iteration in ONE single-workgroup launch, for the sparse end of open scenes.
// mis_test.txt at 1024x1024 depth 16 has 1 048 576 / 196 631 / 1 057 / 49 / 0 / 0 ... live paths per bounce: 12 of the 16
// bounces are empty, yet each costs three full-capacity launches whose blocks only read the live count and leave
// (36 of the 49 launches of an iteration).  The render loop therefore enqueues full-width bounces only while the live
// count it has SEEN in earlier iterations is large (asynchronous read-back of counts[], no host wait — pathtrace.hip),
// and finishes every iteration with this kernel: one block of 1024 threads that loops over the remaining bounces and, per
// bounce, over chunks of 1024 paths — closest hit (the in-lane form, as pipeline 1), compacted slot through a block prefix
// (order-preserving: still the reference's RNG key), shading, survivors written densely.  It is correct for ANY live
// count (the prediction only decides how fast the tail is), writes the same counts[] / ray total / radiance as the
// full-width kernels, and exits at the first bounce that has no path left.
// ------------------------------------------------------------------------------------------------------------------
#define PTK_TAIL_BLOCK 1024
struct TailArgs {
    int iter_arg; const int* iter_ptr; int depth0, trace_depth;
    int* counts;                             // counts[d] = live paths entering bounce d (read at depth0, written above)
    const int* nshaded;                      // paths the last full-width launch shaded (its windows = ceil / 256): counts[depth0 - 1], or hcounts[depth0 - 1] after a hit-dense launch
    PtkPaths cur, nxt;                       // cur: the stream entering bounce depth0 (windowed if G.pref != null)
    Gather G;
    FilmArgs film; int spp, slot_bias;
    int dbg_shift0, dbg_shift_deep;          // test hooks only (0 in the product): extra key offset at depth 0 / depth >= 1
    unsigned long long* ray_total;
    int* iter_advance; int iter_stride;
    // fused pipeline: bounce depth0 has been traced already (k_shade_trace); its hit records / ranks sit at the paths' physical slots
    PtkHits hits0; const int* hr0; int have_hits;
    int compact;                             // ... in k_shade_trace's compact stream layout
    // ... by a HIT-DENSE launch (every fused layout but SPLIT): the tail enumerates the hits of bounce depth0 like that launch's successor would
    // (hit prefix + first_h + the inverse-map words), never its misses — and takes over as soon as a bounce has few HITS, however many rays
    int hit_dense; const int* hit_pref; const int* first_h; const int* nhits;
    int win;                                 // slots per window of the traced stream: 256, or 64 after k_shade_trace_w
    int* hcounts;                            // (may be null) hcounts[d] = hits of bounce d, kept up to date for the host's hand-over hint
    const int* cum_prev; int* cum_out; int cum_stride, ring_div;      // tiles in flight (Workspace::cum_*), null = off
};

template <int MODE>
__global__ void __launch_bounds__(PTK_TAIL_BLOCK) k_tail(SceneView S, TailArgs A) {
    constexpr int NT = PTK_TAIL_BLOCK;
    __shared__ DevMaterial s_mats[PTK_LDS_MATS];
    PTK_PHASE_BEGIN();                                            // (diagnostic build: phase stamps of WAVE 0 — the wave that holds a bounce of a few dozen paths — in slot 1 of the phase buffer; the other waves' go to slot 0)
    const int TK = threadIdx.x < 64 ? 1 : 0; (void)TK;
    const GeomStage gs = stage_geoms(S);
    const MatTable mt = stage_materials<NT>(S, s_mats);
    const int step = A.iter_ptr ? *A.iter_ptr : A.iter_arg;
    const int iter = (step - 1) * A.spp + 1;
    PtkPaths cur = A.cur, nxt = A.nxt;
    Gather G = A.G;
    int n = A.hit_dense ? *A.nhits : A.counts[A.depth0];        // paths to look at in the first bounce: its hits (hit-dense) or its live paths
    const int crow = cum_row(step, A.ring_div, A.cum_stride);
    PTK_PHASE_AT(TK, 0);                                           // 0: staging (primitives, materials)
    for (int depth = A.depth0; depth < A.trace_depth; ++depth) {
        const Film img = {A.film, depth};
        if (n <= 0) {                                            // nothing left: the remaining bounces are empty
            for (int d = depth + 1 + threadIdx.x; d <= A.trace_depth; d += NT) A.counts[d] = 0;
            if (A.cum_out) for (int d = depth + threadIdx.x; d <= A.trace_depth; d += NT) if (!(A.have_hits && d == A.depth0)) A.cum_out[crow + d] = cum_read(A.cum_prev, crow, d);
            if (A.hcounts) for (int d = depth + threadIdx.x; d <= A.trace_depth; d += NT) if (!(A.have_hits && d == A.depth0)) A.hcounts[d] = 0;
            break;
        }
        int hits_before = 0, survivors = 0;
        for (int c0 = 0; c0 < n; c0 += NT) {
            const int i = c0 + threadIdx.x;
            int flag = 0, alive = 0;
            unsigned nv = 0, nt = 0;
            Hit h;
            PathRegs p;
            h.t = -1.f; h.n = mk3(0, 0, 0); h.p = mk3(0, 0, 0); h.uv.x = h.uv.y = 0.f; h.mat = 0; h.geom = -1; h.tri = -1;
            p.o = mk3(0, 0, 0); p.d = mk3(0, 0, 0); p.col = mk3(0, 0, 0); p.pix = 0; p.bounces = 0; p.prevPdf = 0.f;
            if (i < n) {
                const bool traced = A.have_hits && depth == A.depth0;      // by the last full-width launch (ENV radiance of its misses added there)
                int src;
                if (traced && A.hit_dense) {
                    // hit i of the traced bounce -> window (hit prefix) -> slot (inverse-map word of the window's (i - base)-th slot)
                    const int nw = (*A.nshaded + A.win - 1) / A.win;
                    const int w0 = A.first_h[i / A.win];
                    int k = 0, base = A.hit_pref[w0];
                    while (k < 4 && i >= A.hit_pref[w0 + k + 1]) { ++k; base = A.hit_pref[w0 + k]; }
                    if (k == 4) k = sparse_window(A.hit_pref, w0 + 4, nw, i, base) - w0;
                    const int w = w0 + k;
                    const int at = w * A.win + (i - base);
                    const int word = (A.compact ? __float_as_int(A.hits0.t_n[4 * (size_t)at + 3]) : A.hr0[at]) & 0x3fffffff;
                    src = w * A.win + (word & 0xff);
                } else src = gather_source(G, i / PTK_BLOCK, i, n, G.pref ? (*A.nshaded + PTK_BLOCK - 1) / PTK_BLOCK : 0);
                if (traced && A.compact) {                                  // k_shade_trace's compact streams: (d, pix) (col, bounces) (n, rank) (p, mat)
                    const float4 a4 = ld4(cur.o_pix, src), b4 = ld4(cur.d_bnc, src), h0 = ld4(A.hits0.t_n, src);
                    p.pix = __float_as_int(a4.w); p.d = mk3(a4.x, a4.y, a4.z); p.bounces = __float_as_int(b4.w);
                    p.col = mk3(b4.x, b4.y, b4.z); p.prevPdf = -1.f;
                    if (MODE == PTK_MODE_MIS) { const float4 c4 = ld4(cur.col_pdf, src); p.o = mk3(c4.x, c4.y, c4.z); p.prevPdf = c4.w; }
                    flag = __float_as_int(h0.w) >= 0;
                    if (flag) {
                        const float4 h1 = ld4(A.hits0.p_mat, src);
                        h.t = 1.0f; h.n = mk3(h0.x, h0.y, h0.z); h.p = mk3(h1.x, h1.y, h1.z); h.mat = __float_as_int(h1.w);
                        if (MODE == PTK_MODE_MIS) {
                            h.geom = h.mat;
                            h.mat = gs.on ? __float_as_int(gs.xf[(size_t)h.geom * PTK_XF_FLOATS + 34]) : S.geoms[h.geom].materialid;
                        }
                    }
                } else {
                const float4 o4 = ld4(cur.o_pix, src), d4 = ld4(cur.d_bnc, src), c4 = ld4(cur.col_pdf, src);
                p.o = mk3(o4.x, o4.y, o4.z); p.pix = __float_as_int(o4.w);
                p.d = mk3(d4.x, d4.y, d4.z); p.bounces = __float_as_int(d4.w);
                p.col = mk3(c4.x, c4.y, c4.z); p.prevPdf = c4.w;
                if (traced) {
                    flag = A.hr0[src] >= 0;
                    if (flag) {
                        const float4 h0 = ld4(A.hits0.t_n, src), h1 = ld4(A.hits0.p_mat, src);
                        const float4 h2 = S.slim_hits ? make_float4(0.f, 0.f, __int_as_float(-1), __int_as_float(-1)) : ld4(A.hits0.uv_ids, src);
                        h.t = h0.x; h.n = mk3(h0.y, h0.z, h0.w); h.p = mk3(h1.x, h1.y, h1.z); h.mat = __float_as_int(h1.w);
                        h.uv.x = h2.x; h.uv.y = h2.y; h.geom = __float_as_int(h2.z); h.tri = __float_as_int(h2.w);
                    }
                } else {
                    PTK_PHASE_AT(TK, 1);                           // 1: gather + path loads
                    flag = closest_hit<false>(S, gs, p.o, p.d, h, nv, nt);
                    if (!flag && S.env_tex >= 0) add_radiance(img, p.pix, nan_to_zero(p.col * env_radiance(S, p.d)));   // pathtrace.cu:303-311
                    PTK_PHASE_AT(TK, 2);                           // 2: closest hit
                }
                }
            }
            PTK_PHASE_AT(TK, 1);                                   // 1: gather + loads (traced bounce: incl. the hit record)
            int total;
            const int local = block_excl_prefix_n<NT>(flag, total);
            PTK_PHASE_AT(TK, 3);                                   // 3: block prefixes
            if (flag) alive = shade_path<MODE>(S, gs, mt, iter, hits_before + local + A.slot_bias + cum_read(A.cum_prev, crow, depth) + (depth >= 1 ? A.dbg_shift_deep : A.dbg_shift0), depth, h, p, img);
            PTK_PHASE_AT(TK, 4);                                   // 4: shading
            int total2;
            const int local2 = block_excl_prefix_n<NT>(alive, total2);
            PTK_PHASE_AT(TK, 3);
            if (alive) {
                const int s = survivors + local2;
                st4(nxt.o_pix, s, make_float4(p.o.x, p.o.y, p.o.z, __int_as_float(p.pix)));
                st4(nxt.d_bnc, s, make_float4(p.d.x, p.d.y, p.d.z, __int_as_float(p.bounces)));
                st4(nxt.col_pdf, s, make_float4(p.col.x, p.col.y, p.col.z, p.prevPdf));
            }
            hits_before += total; survivors += total2;
        }
        if (threadIdx.x == 0) {
            A.counts[depth + 1] = survivors;
            if (A.hcounts && !(A.have_hits && depth == A.depth0)) A.hcounts[depth] = hits_before;      // (the traced bounce's count was written by its launch)
            if (A.cum_out && !(A.have_hits && depth == A.depth0)) A.cum_out[crow + depth] = cum_read(A.cum_prev, crow, depth) + hits_before;
            if (A.ray_total && !(A.have_hits && depth == A.depth0)) atomicAdd(A.ray_total, (unsigned long long)n);     // (counted by the launch that traced it)
        }
        __syncthreads();                                         // this bounce's stores are visible to the whole workgroup before the next bounce reads them
        PTK_PHASE_AT(TK, 5);                                       // 5: survivor stores, counts, end-of-bounce barrier
        const PtkPaths t = cur; cur = nxt; nxt = t;
        G.pref = nullptr; G.first = nullptr; G.src = nullptr;   // survivors were written densely
        n = survivors;
#ifdef PTK_PHASE_TIMING
        if ((threadIdx.x & 63) == 0) g_phase_buf[TK][ph_slot][6] += 1ull;      // 6: bounces run
#endif
    }
    if (threadIdx.x == 0 && A.iter_advance) *A.iter_advance += A.iter_stride;
    PTK_PHASE_END(TK);
}

// stand-alone form of segments_block (one block; used when the shading kernel does not run the tail itself)
#define PTK_SEG_BLOCK 1024
__global__ void __launch_bounds__(PTK_SEG_BLOCK) k_segments(int n_arg, const int* __restrict__ d_n, const int* __restrict__ agg, int* __restrict__ pref,
                                                            int* __restrict__ first, int* __restrict__ count_out, int* iter_advance, int iter_stride) {
    segments_block<PTK_SEG_BLOCK, false>(d_n ? *d_n : n_arg, agg, pref, first, count_out, iter_advance, iter_stride);
}

// ------------------------------------------------------------------------------------------------------------------
// sendImageToPBO, pathtrace.cu:32-65 (ACESFilm + gammaCorrection, utilities.h:35-48)
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(PTK_BLOCK) k_tonemap(const float* __restrict__ img, int P, int iter, uchar4* __restrict__ pbo) {
    const int i = blockIdx.x * PTK_BLOCK + threadIdx.x;
    if (i >= P) return;
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float x = img[3 * (size_t)i + k] / (float)iter;
        float y = gclamp((x * (2.51f * x + 0.03f)) / (x * (2.43f * x + 0.59f) + 0.14f), 0.f, 1.f);
        y = powf(y, 1 / 2.2f);
        c[k] = gclamp(y * 255.0f, 0.f, 255.0f);
    }
    pbo[i] = make_uchar4((unsigned char)c[0], (unsigned char)c[1], (unsigned char)c[2], 0);
}

// unordered accumulation (MIS with several planes, pathtrace.hip: resolve_image): image = ((plane0 + plane1) + plane2) + ...  (fixed order -> deterministic)
__global__ void __launch_bounds__(PTK_BLOCK) k_sum_planes(const float* __restrict__ planes, int n_planes, size_t plane_len, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * PTK_BLOCK + threadIdx.x;
    if (i >= plane_len) return;
    float s = planes[i];
    for (int p = 1; p < n_planes; ++p) s += planes[(size_t)p * plane_len + i];
    out[i] = s;
}

// ORDERED ACCUMULATION (round 4; pathtrace.hip: merge_lane).  With several iterations in flight (or several samples per launch) an iteration adds
// its radiance into planes of its own, zero when it starts — one per sample and, in MIS mode, per bounce — and every (pixel, plane) receives at most
// ONE add: BSDF / DIRECT add once per pixel and iteration (pathtrace.cu:398,486 or the ENV add of :303-311); MIS adds once per pixel and BOUNCE
// (the next-event term of :572, the light hit of :556 or the ENV add, whichever ends or continues the path there).  This kernel folds the planes
// into the image in the order the reference made those adds — sample by sample, bounce by bounce — and leaves them zero for the lane's next launch:
// image = ((image + a_1) + a_2) + ..., the reference's running sum (pathtrace.cu:600), bit for bit in every sample mode.  A wave owns a tile of 64
// pixels; it visits the planes whose touched-tile flag is set (film_add), so the sparse adds of deep bounces and open scenes cost nothing elsewhere.
__global__ void __launch_bounds__(PTK_BLOCK) k_merge_planes(float* __restrict__ image, float* __restrict__ planes, unsigned char* __restrict__ flags,
                                                            int n_planes, int n_pixels, int n_tiles) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * (PTK_BLOCK / 64) + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    const int px = tile * 64 + lane;
    const bool in = px < n_pixels;
    const size_t plane_len = 3 * (size_t)n_pixels;
    // (r5) per pixel: only a plane value that is not zero is an add — the image is read and written for the pixels that received radiance, not for the
    // whole touched tile (BSDF: a path adds when it hits a light, a few percent of the pixels of an iteration, but nearly every 64-pixel tile holds one:
    // the merge used to read and rewrite the whole image every iteration).  v + 0 = v: same image; a plane that is zero needs no clear either.
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    bool have = false;
    for (int q0 = 0; q0 < n_planes; q0 += 64) {
        const int q = q0 + lane;
        unsigned char* fp = flags + (size_t)(q < n_planes ? q : 0) * n_tiles + tile;
        const bool dirty = q < n_planes && *fp != 0;
        if (dirty) *fp = 0;
        unsigned long long m = __ballot(dirty);
        while (m) {                                              // wave-uniform: the touched planes of this tile, in order
            const int k = __builtin_ctzll(m);
            m &= m - 1ull;
            float* p = planes + (size_t)(q0 + k) * plane_len + 3 * (size_t)px;
            if (in) {
                const float a0 = p[0], a1 = p[1], a2 = p[2];
                if (((__float_as_uint(a0) | __float_as_uint(a1) | __float_as_uint(a2)) & 0x7fffffffu) != 0u) {
                    if (!have) { v0 = image[3 * (size_t)px]; v1 = image[3 * (size_t)px + 1]; v2 = image[3 * (size_t)px + 2]; have = true; }
                    v0 += a0; v1 += a1; v2 += a2;
                    p[0] = 0.f; p[1] = 0.f; p[2] = 0.f;
                }
            }
        }
    }
    if (have) { image[3 * (size_t)px] = v0; image[3 * (size_t)px + 1] = v1; image[3 * (size_t)px + 2] = v2; }
}

#ifdef PTK_TEST_HOOKS
// ------------------------------------------------------------------------------------------------------------------
// TEST-ONLY: the exact fast paths of ptk_device.hpp (div_exact / inv_exact / sqrt_exact / inv_sqrt_exact / div3_exact / box_test / sphere_test,
// the packed dot / normalize / xform_*) against the plain IEEE forms (`/`, sqrtf, scalar arithmetic in glm's order), bit for bit,
// on pseudo-random operand sets: any bit pattern, moderate magnitudes, the edges of the fast-path ranges (2^-40, 2^40, 2^-80, 2^80,
// 2^-96), slab numerators next to zero.  bad[k] counts the mismatches of check k (NaN == NaN).  tests/test_gpu_exact_math.py.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float chk_operand(uint32_t h, int cls) {
    const uint32_t mant = h & 0x7fffffu, sign = (h >> 31) << 31;
    switch (cls) {
    case 0: return __uint_as_float(h);                                                        // anything: NaN, inf, denormals, zeros
    case 1: return __uint_as_float(sign | ((121u + ((h >> 23) & 15u)) << 23) | mant);         // 2^-6 .. 2^9
    case 2: { const int e[8] = {-42, -41, -40, -39, 38, 39, 40, 41}; return __uint_as_float(sign | ((uint32_t)(127 + e[(h >> 23) & 7u]) << 23) | ((h >> 26) & 1u ? mant : 0u)); }
    case 3: { const int e[8] = {-97, -96, -95, -81, -80, 79, 80, 81}; return __uint_as_float(sign | ((uint32_t)(127 + e[(h >> 23) & 7u]) << 23) | ((h >> 26) & 1u ? mant : 0u)); }
    default: { const float base = (h >> 30) & 1u ? 0.5f : -0.5f;                              // next to a cube face: numerators near 0
               return __uint_as_float(__float_as_uint(base) + ((h >> 23) & 7u) - 3u); }
    }
}
__device__ __forceinline__ bool chk_same(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }
__global__ void __launch_bounds__(PTK_BLOCK) k_exact_check(unsigned long long n, uint32_t seed, unsigned long long* __restrict__ bad, uint32_t* __restrict__ samples) {
    const unsigned long long i = (unsigned long long)blockIdx.x * PTK_BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t h[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) h[k] = utilhash((uint32_t)i * 8u + (uint32_t)k + seed * 0x9e3779b9u) ^ utilhash((uint32_t)(i >> 29) + 77u * (uint32_t)k);
    const int cls = (int)(utilhash((uint32_t)i + seed) % 5u);
    const float a = chk_operand(h[0], cls), b = chk_operand(h[1], cls), c = chk_operand(h[2], cls), d = chk_operand(h[3], cls == 4 ? 1 : cls);
    const float e = chk_operand(h[4], cls == 4 ? 1 : cls), f = chk_operand(h[5], cls == 4 ? 1 : cls);
    unsigned m = 0u;
    m |= chk_same(div_exact(a, d), a / d) ? 0u : 1u;
    m |= chk_same(inv_exact(d), 1.0f / d) ? 0u : 2u;
    const float pa = fabsf(a);
    m |= chk_same(sqrt_exact(pa), sqrtf(pa)) ? 0u : 4u;
    m |= chk_same(inv_sqrt_exact(pa), 1.0f / sqrtf(pa)) ? 0u : 8u;
    { const f3 q = div3_exact(mk3(a, b, c), d);
      m |= (chk_same(q.x, a / d) && chk_same(q.y, b / d) && chk_same(q.z, c / d)) ? 0u : 16u; }
    { // the exact tests as the kernels call them (optimistic fast form + plain fallback) against the plain form alone, on a random
      // transform (general and diagonal) and a ray made of the operand set: distance, hit point and deferred normal, bit for bit
      struct { float inv[12], fwd[12]; int diag; } g;
#pragma unroll
      for (int k = 0; k < 12; ++k) { g.inv[k] = chk_operand(utilhash(h[7] + (uint32_t)k), (k % 5 == 0 || k == 3 || k == 7 || k == 11) ? 1 : (cls == 0 ? 0 : 1));
                                     g.fwd[k] = chk_operand(utilhash(h[7] + 100u + (uint32_t)k), 1); }
      bool ok = true;
#pragma unroll
      for (int dg = 0; dg < 2; ++dg) {
          g.diag = dg;
          const f3 ro = mk3(a, b, c), rd = mk3(d, e, f);
          f3 P1 = mk3(0, 0, 0), n1 = mk3(0, 0, 0), P2 = mk3(0, 0, 0), n2 = mk3(0, 0, 0);
          bool dummy = true;
          const float t1 = box_test(g, ro, rd, P1, n1), t2 = box_test_impl<false>(g, ro, rd, P2, n2, dummy);
          ok = ok && chk_same(t1, t2) && (!(t2 >= 0.f) || (chk_same(P1.x, P2.x) && chk_same(P1.y, P2.y) && chk_same(P1.z, P2.z) && chk_same(n1.x, n2.x) && chk_same(n1.y, n2.y) && chk_same(n1.z, n2.z)));
          const float u1 = sphere_test(g, ro, rd, P1, n1), u2 = sphere_test_impl<false>(g, ro, rd, P2, n2, dummy);
          ok = ok && chk_same(u1, u2) && (!(u2 >= 0.f) || (chk_same(P1.x, P2.x) && chk_same(P1.y, P2.y) && chk_same(P1.z, P2.z) && chk_same(n1.x, n2.x) && chk_same(n1.y, n2.y) && chk_same(n1.z, n2.z)));
      }
      m |= ok ? 0u : 32u; }
    { const f3 v = mk3(a, b, c), nv = normalize(v);
      const float r = 1.0f / sqrtf(a * a + b * b + c * c);
      m |= (chk_same(nv.x, a * r) && chk_same(nv.y, b * r) && chk_same(nv.z, c * r) && chk_same(length(v), sqrtf(a * a + b * b + c * c)) &&
            chk_same(dot(v, mk3(d, e, f)), a * d + b * e + c * f)) ? 0u : 64u; }
    { float mtx[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) mtx[k] = chk_operand(utilhash(h[6] + (uint32_t)k), 1);
      const f3 v = mk3(a, b, c);
      const f3 p = xform_point(mtx, v), q = xform_dir(mtx, v), r = xform_nrm(mtx, v), pd = xform_point_d(mtx, v, 1), qd = xform_dir_d(mtx, v, 1), rd = xform_nrm_d(mtx, v, 1);
      bool ok = chk_same(p.x, (mtx[0] * a + mtx[1] * b) + (mtx[2] * c + mtx[3])) && chk_same(p.y, (mtx[4] * a + mtx[5] * b) + (mtx[6] * c + mtx[7])) &&
                chk_same(p.z, (mtx[8] * a + mtx[9] * b) + (mtx[10] * c + mtx[11]));
      ok = ok && chk_same(q.x, (mtx[0] * a + mtx[1] * b) + mtx[2] * c) && chk_same(q.y, (mtx[4] * a + mtx[5] * b) + mtx[6] * c) && chk_same(q.z, (mtx[8] * a + mtx[9] * b) + mtx[10] * c);
      ok = ok && chk_same(r.x, (mtx[0] * a + mtx[1] * b) + mtx[2] * c) && chk_same(r.y, (mtx[3] * a + mtx[4] * b) + mtx[5] * c) && chk_same(r.z, (mtx[6] * a + mtx[7] * b) + mtx[8] * c);
      ok = ok && chk_same(pd.x, mtx[0] * a + mtx[3]) && chk_same(pd.y, mtx[5] * b + mtx[7]) && chk_same(pd.z, mtx[10] * c + mtx[11]);
      ok = ok && chk_same(qd.x, mtx[0] * a) && chk_same(qd.y, mtx[5] * b) && chk_same(qd.z, mtx[10] * c);
      ok = ok && chk_same(rd.x, mtx[0] * a) && chk_same(rd.y, mtx[4] * b) && chk_same(rd.z, mtx[8] * c);
      m |= ok ? 0u : 128u; }
    if (m) {
#pragma unroll
        for (int k = 0; k < 8; ++k) if (m & (1u << k)) atomicAdd(&bad[k], 1ull);
        if (samples) {                                   // the first failing operand sets, for the test's message: {mask, a..f}
            const unsigned slot = atomicAdd(&samples[0], 1u);
            if (slot < 15u) { uint32_t* r = samples + 8 * (slot + 1); r[0] = m; r[1] = __float_as_uint(a); r[2] = __float_as_uint(b); r[3] = __float_as_uint(c);
                              r[4] = __float_as_uint(d); r[5] = __float_as_uint(e); r[6] = __float_as_uint(f); r[7] = (uint32_t)cls; }
        }
    }
}

#endif // PTK_TEST_HOOKS

inline int grid_for(int n) { return (n + PTK_BLOCK - 1) / PTK_BLOCK; }

// The 8-wide walk of the work-list rays (g = blocks of 256 stream slots the list may cover).  pool > 0: k_traverse_pool with chunks of `pool` rays
// per block; 0: k_traverse_wide, one ray per group of lpr lanes.  Bit-identical hits either way (pick_wide_pool below chooses).
static void launch_wide_walk(const SceneView& S, int g, int lpr, int pool, const unsigned* queue, const int* list, PtkPaths paths, PtkHits hits, int* valid,
                             const int* srcidx, hipStream_t st) {
    if (pool > 0 && S.wide_width == 8) {             // (the pooled walk exists for 8-wide nodes only)
        const int blocks = (g * PTK_BLOCK + pool - 1) / pool;
        const size_t lds = (size_t)pool * 2 * sizeof(float4);
        const char* s16_env = getenv("PTK_POOL_S16");                                                  // (A/B and test knob: 0 = 32-bit stacks; read per launch, a test flips it)
        const int s16_forced = s16_env ? atoi(s16_env) : -1;
        const bool s16 = S.n_wnodes < 32768 && S.n_tris < 32768 && s16_forced != 0;
        if (lpr == 2) { if (s16) k_traverse_pool<2, true><<<blocks, PTK_WIDE_BLOCK, lds, st>>>(S, queue, list, paths, hits, valid, srcidx, pool);
                        else k_traverse_pool<2, false><<<blocks, PTK_WIDE_BLOCK, lds, st>>>(S, queue, list, paths, hits, valid, srcidx, pool); }
        else { if (s16) k_traverse_pool<4, true><<<blocks, PTK_WIDE_BLOCK, lds, st>>>(S, queue, list, paths, hits, valid, srcidx, pool);
               else k_traverse_pool<4, false><<<blocks, PTK_WIDE_BLOCK, lds, st>>>(S, queue, list, paths, hits, valid, srcidx, pool); }
        return;
    }
    if (S.wide_width == 16) {
        if (lpr == 2) k_traverse_wide<false, 2, 16><<<2 * g * (PTK_BLOCK / PTK_WIDE_BLOCK), PTK_WIDE_BLOCK, 0, st>>>(S, queue, list, paths, hits, valid, nullptr, srcidx);
        else k_traverse_wide<false, 4, 16><<<4 * g * (PTK_BLOCK / PTK_WIDE_BLOCK), PTK_WIDE_BLOCK, 0, st>>>(S, queue, list, paths, hits, valid, nullptr, srcidx);
        return;
    }
    if (lpr == 2) k_traverse_wide<false, 2><<<2 * g * (PTK_BLOCK / PTK_WIDE_BLOCK), PTK_WIDE_BLOCK, 0, st>>>(S, queue, list, paths, hits, valid, nullptr, srcidx);
    else k_traverse_wide<false, 4><<<4 * g * (PTK_BLOCK / PTK_WIDE_BLOCK), PTK_WIDE_BLOCK, 0, st>>>(S, queue, list, paths, hits, valid, nullptr, srcidx);
}
inline int hip_rc(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }
inline int launch_rc() { return hip_rc(hipGetLastError()); }

DevCamera to_dev_camera(const PtkCamera& c) {
    DevCamera d;
    d.W = c.res[0]; d.H = c.res[1];
    d.pos = mk3(c.position[0], c.position[1], c.position[2]);
    d.view = mk3(c.view[0], c.view[1], c.view[2]);
    d.up = mk3(c.up[0], c.up[1], c.up[2]);
    d.right = mk3(c.right[0], c.right[1], c.right[2]);
    d.plx = c.pixelLength[0]; d.ply = c.pixelLength[1];
    return d;
}

int launch_shade_binned(int mode, const SceneView& S, int iter, const int* iter_ptr, int depth, int n, const int* d_n, PtkHits hits, PtkPaths paths,
                        const int* va, const int* aa, int* ab, const FilmArgs& film, int spp, int slot_bias, hipStream_t st,
                        const int* srcidx, PtkPaths seg_out, SegTail seg_tail) {
    constexpr int NT = PTK_BLOCK;
    const int g = (n + NT - 1) / NT;
    if (g == 0) return 0;
    if (!seg_out.o_pix) return PTK_E_BADARG;
    switch (mode) {
    case PTK_MODE_BSDF: k_shade_binned<PTK_MODE_BSDF, NT><<<g, NT, 0, st>>>(S, iter, iter_ptr, depth, n, d_n, hits, paths, va, aa, ab, film, spp, slot_bias, srcidx, seg_out, seg_tail); break;
    case PTK_MODE_DIRECTLI: k_shade_binned<PTK_MODE_DIRECTLI, NT><<<g, NT, stage_bytes(S), st>>>(S, iter, iter_ptr, depth, n, d_n, hits, paths, va, aa, ab, film, spp, slot_bias, srcidx, seg_out, seg_tail); break;
    case PTK_MODE_MIS: k_shade_binned<PTK_MODE_MIS, NT><<<g, NT, stage_bytes(S), st>>>(S, iter, iter_ptr, depth, n, d_n, hits, paths, va, aa, ab, film, spp, slot_bias, srcidx, seg_out, seg_tail); break;
    default: return PTK_E_BADARG;
    }
    return launch_rc();
}

unsigned* g_abi_queue = nullptr;       // traversal list counter + list for the kernel-level ptk_intersect entry point
int* g_abi_list = nullptr;
int g_abi_list_cap = 0;
int* g_abi_hflag = nullptr;             // work-list flags / per-block counts of the same entry point (same layout as the render loop)
int* g_abi_hagg = nullptr;

// scratch for the stand-alone ptk_compact entry point (per-block aggregates), grown on demand
int* g_compact_agg = nullptr;
int g_compact_agg_cap = 0;

// fused pipeline: one launch shades bounce dt - 1 (paths / hits / ranks of `cur`) — or, gen, generates the camera rays — and traces
// bounce dt into `nxt`.  Scenes with triangles continue with the work-list BVH stage on the survivors' physical slots.
template <int MODE, bool GEN>
static int launch_shade_trace_t(const SceneView& S, bool split, bool compact, bool wave_windows, const GenArgs& G, int iter, const int* iter_ptr, int depth, int capacity, const int* d_n,
                                const TraceIO& io, const FilmArgs& film, int spp, int slot_bias, const SegTail2& tail, hipStream_t st) {
    const int g = grid_for(capacity);
    if (split) k_shade_trace<MODE, GEN, true, false><<<g, PTK_BLOCK, stage_bytes(S), st>>>(S, G, iter, iter_ptr, depth, capacity, d_n, io, film, spp, slot_bias, tail);
    else if (compact && wave_windows) k_shade_trace_w<MODE, GEN><<<g, PTK_BLOCK, stage_bytes(S), st>>>(S, G, iter, iter_ptr, depth, capacity, d_n, io, film, spp, slot_bias, tail);
    else if (compact) k_shade_trace<MODE, GEN, false, true><<<g, PTK_BLOCK, stage_bytes(S), st>>>(S, G, iter, iter_ptr, depth, capacity, d_n, io, film, spp, slot_bias, tail);
    else k_shade_trace<MODE, GEN, false, false><<<g, PTK_BLOCK, stage_bytes(S), st>>>(S, G, iter, iter_ptr, depth, capacity, d_n, io, film, spp, slot_bias, tail);
    return launch_rc();
}
} // namespace

// ------------------------------------------------------------------------------------------------------------------
// SECOND TRANSLATION UNIT (round 5): this file is compiled twice.  With -DPTK_TU_BSDF it yields ONLY launch_shade_trace_bsdf — the BSDF-mode
// instantiations of k_shade_trace, the headline's kernel — under -mllvm -amdgpu-sched-strategy=max-memory-clause (Makefile): the same instructions in
// another order (loads grouped into clauses), measured +2...+5 % on cornell.txt and metallic_analytic.txt in BSDF mode, while MIS / DIRECT and the mesh
// kernels are at best unchanged under it (profiles/r5_ab_sched_strategy.txt) — hence per kernel, not per library.  Results are bit-identical: scheduling does
// not touch a single operation.  The argument structs live in this file's anonymous namespace, so they cross the boundary as untyped pointers.
// ------------------------------------------------------------------------------------------------------------------
namespace ptk {
int launch_shade_trace_bsdf(bool gen, const SceneView& S, bool split, bool compact, bool wave_windows, const void* G, int iter, const int* iter_ptr, int depth, int capacity,
                            const int* d_n, const void* io, const FilmArgs& film, int spp, int slot_bias, const void* tail, hipStream_t st);
}
#ifdef PTK_TU_BSDF
namespace ptk {
int launch_shade_trace_bsdf(bool gen, const SceneView& S, bool split, bool compact, bool wave_windows, const void* G, int iter, const int* iter_ptr, int depth, int capacity,
                            const int* d_n, const void* io, const FilmArgs& film, int spp, int slot_bias, const void* tail, hipStream_t st) {
    const GenArgs& g = *static_cast<const GenArgs*>(G); const TraceIO& t = *static_cast<const TraceIO*>(io); const SegTail2& q = *static_cast<const SegTail2*>(tail);
    return gen ? launch_shade_trace_t<PTK_MODE_BSDF, true>(S, split, compact, wave_windows, g, iter, iter_ptr, depth, capacity, d_n, t, film, spp, slot_bias, q, st)
               : launch_shade_trace_t<PTK_MODE_BSDF, false>(S, split, compact, wave_windows, g, iter, iter_ptr, depth, capacity, d_n, t, film, spp, slot_bias, q, st);
}
} // namespace ptk
#else

// ------------------------------------------------------------------------------------------------------------------
// pipeline-internal launchers
// ------------------------------------------------------------------------------------------------------------------
namespace ptk {

// Pooled wide walk or not (k_traverse_pool).  Measured on the bunny scene, Grays/s, three iterations in flight, k_traverse_wide / pool 256 / 512 /
// 1 024 (profiles/r5_ab_pooled_walk.txt): 800 x 800 depth 8 (142 k listed rays per launch): 6.12 / 6.01 / 5.62 / 4.39 — too few rays to keep the
// machine in waves once four share a group; one iteration at a time 3.48 / 2.67 / 2.02 / 1.28; 2 048 x 2 048 depth 16 (0.9 M listed rays): 6.67 /
// 6.91 / 7.08 / 6.53.  With FOUR iterations in flight (the default since the end of round 5; gpurun_out s2_f, profiles/r5_ab_lane_queues.txt section 6), wide /
// 256 / 512: 800^2 6.56 / 6.59 / 6.44; 1 280^2 depth 8 6.84 / 7.00 / 7.20; 2 048^2 depth 16 6.79 / 7.04 / 7.24.  Hence: pools of 512 from 1.5 million
// paths per launch on, and only with iterations in flight.  PTK_WIDE_POOL=<n> forces a chunk (0 = off) for A/B runs and the parity tests.
int pick_wide_pool(int capacity, int n_lanes) {
    if (const char* e = getenv("PTK_WIDE_POOL")) { const int v = atoi(e); return v < 0 ? 0 : (v > 2048 ? 2048 : v); }
    return (n_lanes > 1 && capacity >= (3 << 19)) ? 512 : 0;
}
int local_path_count(int W, int H, int n_parts, int part, int tile_rows) {
    if (n_parts <= 1) return W * H;
    int rows = 0;
    const int n_tiles = (H + tile_rows - 1) / tile_rows;
    for (int t = part; t < n_tiles; t += n_parts) rows += std::min(tile_rows, H - t * tile_rows);
    return rows * W;
}

int launch_generate(const ::Camera& cam, int iter, const int* iter_ptr, int traceDepth, PtkPaths paths, int n_parts, int part, int tile_rows,
                    int n_pix, int spp, int* d_count0, hipStream_t st) {
    DevCamera d;
    d.W = cam.resolution.x; d.H = cam.resolution.y;
    d.pos = mk3(cam.position.x, cam.position.y, cam.position.z);
    d.view = mk3(cam.view.x, cam.view.y, cam.view.z);
    d.up = mk3(cam.up.x, cam.up.y, cam.up.z);
    d.right = mk3(cam.right.x, cam.right.y, cam.right.z);
    d.plx = cam.pixelLength.x; d.ply = cam.pixelLength.y;
    const int g = std::max(1, grid_for(n_pix * spp));
    k_generate<<<g, PTK_BLOCK, 0, st>>>(d, iter, iter_ptr, traceDepth, paths, std::max(1, n_parts), part, std::max(1, tile_rows), n_pix, spp, d_count0);
    return launch_rc();
}

// intersection stage (any of the layouts above), asynchronous on `st`.  agg may be null (kernel-level ABI).
int launch_intersect_stage(const SceneView& S, int n, const int* d_n, PtkPaths cur, PtkHits hits, int* valid, int* agg,
                           unsigned long long* ray_total, unsigned long long* stats, unsigned* queue, int* list, int* hflag, int* hagg,
                           bool inline_bvh, hipStream_t st, void (*mark)(int, bool), const Gather& G, int walk = PTK_WALK_THREADED, int wide_lpr = 4, int wide_pool = 0) {
    const int g = grid_for(n);
    if (g == 0) return 0;
    const bool split = S.n_tris > 0 && !inline_bvh && queue != nullptr && list != nullptr && hflag != nullptr && hagg != nullptr;
    mark(1, true);
    if (!split) {
        if (stats) k_intersect<true, false><<<g, PTK_BLOCK, stage_bytes(S), st>>>(S, n, d_n, cur, hits, valid, agg, ray_total, stats, nullptr, nullptr, G);
        else k_intersect<false, false><<<g, PTK_BLOCK, stage_bytes(S), st>>>(S, n, d_n, cur, hits, valid, agg, ray_total, nullptr, nullptr, nullptr, G);
        mark(1, false);
        return launch_rc();
    }
    static const int wide_stats_mode = getenv("PTK_WIDE_STATS") ? atoi(getenv("PTK_WIDE_STATS")) : 0;      // diagnostics: 1: stats = {wide node visits, leaf visits}; 2: {sum of the waves' slowest rays, all visits}
    if (stats) k_intersect<true, true><<<g, PTK_BLOCK, stage_bytes(S), st>>>(S, n, d_n, cur, hits, valid, nullptr, ray_total, wide_stats_mode == 2 ? nullptr : stats, hflag, hagg, G);
    else k_intersect<false, true><<<g, PTK_BLOCK, stage_bytes(S), st>>>(S, n, d_n, cur, hits, valid, nullptr, ray_total, nullptr, hflag, hagg, G);
    int rc = launch_rc(); if (rc) { mark(1, false); return rc; }
    const bool wide_stats = wide_stats_mode != 0;
    const bool wide = walk == PTK_WALK_WIDE && S.wnodes;
    k_list<<<g, PTK_BLOCK, 0, st>>>(n, d_n, hflag, hagg, list, queue, 0);
    rc = launch_rc(); if (rc) { mark(1, false); return rc; }
    if (wide && stats && wide_stats && S.wide_width == 16 && wide_lpr == 2) k_traverse_wide<true, 2, 16><<<2 * g * (PTK_BLOCK / PTK_WIDE_BLOCK), PTK_WIDE_BLOCK, 0, st>>>(S, queue, list, cur, hits, valid, stats, G.src, wide_stats_mode);
    else if (wide && stats && wide_stats && S.wide_width == 16) k_traverse_wide<true, 4, 16><<<4 * g * (PTK_BLOCK / PTK_WIDE_BLOCK), PTK_WIDE_BLOCK, 0, st>>>(S, queue, list, cur, hits, valid, stats, G.src, wide_stats_mode);
    else if (wide && stats && wide_stats && wide_lpr == 2) k_traverse_wide<true, 2><<<2 * g * (PTK_BLOCK / PTK_WIDE_BLOCK), PTK_WIDE_BLOCK, 0, st>>>(S, queue, list, cur, hits, valid, stats, G.src, wide_stats_mode);
    else if (wide && stats && wide_stats) k_traverse_wide<true, 4><<<4 * g * (PTK_BLOCK / PTK_WIDE_BLOCK), PTK_WIDE_BLOCK, 0, st>>>(S, queue, list, cur, hits, valid, stats, G.src, wide_stats_mode);
    else if (wide && !stats) launch_wide_walk(S, g, wide_lpr, wide_pool, queue, list, cur, hits, valid, G.src, st);
    else if (stats) k_traverse<true><<<g, PTK_BLOCK, 0, st>>>(S, queue, list, cur, hits, valid, stats, G.src);
    else k_traverse<false><<<g, PTK_BLOCK, 0, st>>>(S, queue, list, cur, hits, valid, nullptr, G.src);
    rc = launch_rc(); if (rc) { mark(1, false); return rc; }
    k_finalize<<<g, PTK_BLOCK, 0, st>>>(S, n, d_n, hits, valid, agg);
    mark(1, false);
    return launch_rc();
}

// one bounce of the staged pipeline: intersect -> shade (virtual compaction of misses) -> scatter survivors.
// `capacity` bounds the launch; the live count is read from ws.counts[depth] on the device.
static void no_mark(int, bool) {}
int launch_bounce_staged_profiled(const SceneView& S_in, int mode, int iter, int depth, int capacity, PtkPaths cur, PtkPaths nxt,
                                  PtkHits hits, float* img, const Workspace& ws, hipStream_t st, void (*mark)(int, bool), bool last_bounce) {
    const int g = grid_for(capacity);
    if (g == 0) return 0;
    const int* d_n = ws.counts + depth;
    // (uv, geomID, triangleID) of the hit record feed texture lookups and lightPDF only: without textures and outside MIS
    // the shading kernels never look at them, so that third of the record is neither written nor read
    SceneView S = S_in;
    S.slim_hits = (S.n_tex == 0 && mode != PTK_MODE_MIS) ? 1 : 0;
    // segmented streams: `cur` holds the survivors of the previous bounce window by window (dense at bounce 0)
    if (!ws.seg_pref || !ws.seg_first || !ws.srcidx) return PTK_E_BADARG;
    Gather G = {nullptr, nullptr, nullptr};
    if (depth > 0) { G.pref = ws.seg_pref; G.first = ws.seg_first; G.src = ws.srcidx; }
    // the shading stage's flag/aggregate arrays are free during the intersection stage: they carry the work-list flags
    int rc = launch_intersect_stage(S, capacity, d_n, cur, hits, ws.valid_a, ws.agg_a, ws.ray_total, ws.stats, ws.queue, ws.list,
                                    ws.valid_b, ws.agg_b, ws.inline_bvh, st, mark, G, ws.walk, ws.wide_lpr, ws.wide_pool);
    if (rc) return rc;
    if (S.env_tex >= 0) {
        k_env_miss<<<g, PTK_BLOCK, 0, st>>>(S, capacity, d_n, cur, ws.valid_a, film_of(img, ws), depth, G.src);
        if ((rc = launch_rc())) return rc;
    }
    mark(2, true);
    const bool fold = ws.seg_done != nullptr;                // the shading kernel's last block does the window bookkeeping itself
    const SegTail tail = {fold ? ws.seg_done : nullptr, ws.seg_pref, ws.seg_first, ws.counts + depth + 1, last_bounce ? ws.iter_dev : nullptr, ws.iter_stride};
    rc = launch_shade_binned(mode, S, iter, ws.iter_dev, depth, capacity, d_n, hits, cur, ws.valid_a, ws.agg_a, ws.agg_b, film_of(img, ws), ws.spp,
                             ws.slot_bias + (depth >= 1 ? ws.dbg_shift_deep : ws.dbg_shift0), st, G.src, nxt, tail);
    mark(2, false);
    if (rc || fold) return rc;
    mark(3, true);                          // survivors are already in nxt, window by window: only the bookkeeping is left
    k_segments<<<1, PTK_SEG_BLOCK, 0, st>>>(capacity, d_n, ws.agg_b, ws.seg_pref, ws.seg_first, ws.counts + depth + 1,
                                             last_bounce ? ws.iter_dev : nullptr, ws.iter_stride);
    mark(3, false);
    return launch_rc();
}
int launch_bounce_staged(const SceneView& S, int mode, int iter, int depth, int capacity, PtkPaths cur, PtkPaths nxt,
                         PtkHits hits, float* img, const Workspace& ws, hipStream_t st, bool last_bounce) {
    return launch_bounce_staged_profiled(S, mode, iter, depth, capacity, cur, nxt, hits, img, ws, st, no_mark, last_bounce);
}
// the fused pipeline's compact stream layout (k_shade_trace): no triangles, no textures, no MIS
bool trace_compact(const SceneView& S, int mode) { (void)mode; return S.n_tris == 0 && S.n_tex == 0; }
int launch_shade_trace(const SceneView& S_in, int mode, bool gen, const ::Camera& cam, int n_parts, int part, int tile_rows, int n_pix, int trace_depth,
                       int iter, int dt, int capacity, const TraceBufs& cur, const TraceBufs& nxt, float* img, const Workspace& ws, hipStream_t st,
                       bool last, void (*mark)(int, bool)) {
    const int g = grid_for(capacity);
    if (g == 0) return 0;
    if (!ws.seg_pref || !ws.seg_first || !ws.hit_pref || !ws.agg_h || !ws.agg_b || !ws.seg_done || !ws.tr_valid || !ws.tr_heavy || !ws.hcounts || !ws.seg_first_h) return PTK_E_BADARG;
    SceneView S = S_in;
    S.slim_hits = (S.n_tex == 0 && mode != PTK_MODE_MIS) ? 1 : 0;
    const bool split = S.n_tris > 0 && !ws.inline_bvh && ws.queue && ws.list;       // inline_bvh (tiny mesh): the walk runs inside k_shade_trace's trace phase
    const bool compact = !split && trace_compact(S, mode);
    const bool ww = compact && ws.wave_windows && S.lds_geoms > 0;          // k_shade_trace_w: 64-slot windows, one per wave
    GenArgs G;
    G.cam.W = cam.resolution.x; G.cam.H = cam.resolution.y;
    G.cam.pos = mk3(cam.position.x, cam.position.y, cam.position.z);
    G.cam.view = mk3(cam.view.x, cam.view.y, cam.view.z);
    G.cam.up = mk3(cam.up.x, cam.up.y, cam.up.z);
    G.cam.right = mk3(cam.right.x, cam.right.y, cam.right.z);
    G.cam.plx = cam.pixelLength.x; G.cam.ply = cam.pixelLength.y;
    G.trace_depth = trace_depth; G.n_parts = std::max(1, n_parts); G.part = part; G.tile_rows = std::max(1, tile_rows); G.n_pix = n_pix; G.late_stores = ws.gen_late;
    TraceIO io;
    io.cur = cur.paths; io.hcur = cur.hits; io.hr_cur = cur.hr;
    io.pref = ws.seg_pref; io.first = ws.seg_first; io.hit_pref = ws.hit_pref; io.first_h = ws.seg_first_h;
    io.nxt = nxt.paths; io.hnxt = nxt.hits; io.hr_nxt = nxt.hr;
    io.agg_s = ws.agg_b; io.agg_h = ws.agg_h;
    io.valid_nxt = ws.tr_valid; io.heavy_flag = ws.tr_heavy; io.heavy_agg = ws.agg_a;
    // the launch that shades bounce dt - 1 reads counts[dt - 1]; its last block writes counts[dt] = the rays it traced.  Hit-dense shading (every
    // layout but SPLIT): it shades the hcounts[dt - 1] paths of bounce dt - 1 that hit something, and writes hcounts[dt] = the hits it found
    const int* d_n = gen ? nullptr : (!split ? ws.hcounts : ws.counts) + (dt - 1);
    const SegTail2 tail = {ws.seg_done, ws.seg_pref, ws.seg_first, split ? nullptr : ws.hit_pref, ws.counts + dt, last ? ws.iter_dev : nullptr, ws.iter_stride, ws.ray_total,
                           !split ? ws.seg_first_h : nullptr, !split ? ws.hcounts + dt : nullptr, ws.cum_prev, ws.cum_out, ws.cum_stride, ws.ring_div};
    const int depth = dt - 1;           // RNG key depth of the bounce being shaded
    const int sbias = ws.slot_bias + (depth >= 1 ? ws.dbg_shift_deep : ws.dbg_shift0);
    mark(gen ? 0 : 2, true);
    int rc;
    if (gen) {
        switch (mode) {                 // (the mode only matters for the hit-record layout the next launch expects: same kernel body)
        case PTK_MODE_MIS: rc = launch_shade_trace_t<PTK_MODE_MIS, true>(S, split, compact, ww, G, iter, ws.iter_dev, depth, capacity, d_n, io, film_of(img, ws), ws.spp, sbias, tail, st); break;   // (the MIS layouts carry the origin)
        default: rc = launch_shade_trace_bsdf(true, S, split, compact, ww, &G, iter, ws.iter_dev, depth, capacity, d_n, &io, film_of(img, ws), ws.spp, sbias, &tail, st); break;
        }
    } else {
        switch (mode) {
        case PTK_MODE_BSDF: rc = launch_shade_trace_bsdf(false, S, split, compact, ww, &G, iter, ws.iter_dev, depth, capacity, d_n, &io, film_of(img, ws), ws.spp, sbias, &tail, st); break;
        case PTK_MODE_DIRECTLI: rc = launch_shade_trace_t<PTK_MODE_DIRECTLI, false>(S, split, compact, ww, G, iter, ws.iter_dev, depth, capacity, d_n, io, film_of(img, ws), ws.spp, sbias, tail, st); break;
        case PTK_MODE_MIS: rc = launch_shade_trace_t<PTK_MODE_MIS, false>(S, split, compact, ww, G, iter, ws.iter_dev, depth, capacity, d_n, io, film_of(img, ws), ws.spp, sbias, tail, st); break;
        default: rc = PTK_E_BADARG;
        }
    }
    mark(gen ? 0 : 2, false);
    if (rc) return rc;
    // the traced stream lives in the windows [0, ceil(n_shaded / 256)): the follow-up kernels take the shaded bounce's count as their bound
    const int n_bound = gen ? n_pix * ws.spp : capacity;
    if (split) {
        mark(1, true);
        const bool wide = ws.walk == PTK_WALK_WIDE && S.wnodes;
        k_list<<<g, PTK_BLOCK, 0, st>>>(n_bound, d_n, ws.tr_heavy, ws.agg_a, ws.list, ws.queue, 1);
        if ((rc = launch_rc())) { mark(1, false); return rc; }
        if (wide) launch_wide_walk(S, g, ws.wide_lpr, ws.wide_pool, ws.queue, ws.list, nxt.paths, nxt.hits, ws.tr_valid, nullptr, st);
        else k_traverse<false><<<g, PTK_BLOCK, 0, st>>>(S, ws.queue, ws.list, nxt.paths, nxt.hits, ws.tr_valid, nullptr, nullptr);
        if ((rc = launch_rc())) { mark(1, false); return rc; }
        k_finalize2<<<g, PTK_BLOCK, 0, st>>>(S, n_bound, d_n, nxt.hits, ws.tr_valid, nxt.hr, ws.agg_h, ws.seg_done, ws.hit_pref,
                                             ws.cum_prev, ws.cum_out, ws.cum_stride, ws.ring_div, dt, iter, ws.iter_dev);
        mark(1, false);
        if ((rc = launch_rc())) return rc;
    }
    if (S.env_tex >= 0) {
        k_env_miss2<<<g, PTK_BLOCK, 0, st>>>(S, n_bound, d_n, nxt.paths, nxt.hr, film_of(img, ws), dt);
        if ((rc = launch_rc())) return rc;
    }
    return 0;
}
// pipeline 11 (k_trace_shade): launch `depth` traces AND shades bounce `depth` of the stream in `cur` (depth 0: generates the camera rays instead of
// reading a stream) and leaves the survivors in `nxt`, window by window (the staged pipeline's stream format: k_tail follows unchanged).
bool trace_shade_usable(const SceneView& S) { return S.n_tris == 0 && S.n_tex == 0 && S.env_tex < 0 && S.lds_geoms > 0; }
template <int MODE, bool GEN>
static int launch_trace_shade_t(const SceneView& S, const GenArgs& G, int iter, const int* iter_ptr, int depth, int capacity, const int* d_n, PtkPaths cur, const Gather& Ga,
                                PtkPaths nxt, int* agg_s, LookBack lb, const FilmArgs& film, int spp, int slot_bias, const SegTail& tail, int* count0,
                                unsigned long long* ray_total, hipStream_t st) {
    k_trace_shade<MODE, GEN><<<grid_for(capacity), PTK_BLOCK, stage_bytes(S), st>>>(S, G, iter, iter_ptr, depth, capacity, d_n, cur, Ga, nxt, agg_s, lb, film, spp, slot_bias, tail, count0, ray_total);
    return launch_rc();
}
int launch_trace_shade(const SceneView& S_in, int mode, const ::Camera& cam, int n_parts, int part, int tile_rows, int n_pix, int trace_depth,
                       int iter, int depth, int capacity, PtkPaths cur, PtkPaths nxt, float* img, const Workspace& ws, hipStream_t st, bool last, void (*mark)(int, bool)) {
    if (grid_for(capacity) == 0) return 0;
    if (!ws.seg_pref || !ws.seg_first || !ws.agg_b || !ws.seg_done || !ws.look || !ws.look_gave_up || !trace_shade_usable(S_in)) return PTK_E_BADARG;
    SceneView S = S_in;
    S.slim_hits = 1;
    const bool gen = depth == 0;
    GenArgs G;
    G.cam.W = cam.resolution.x; G.cam.H = cam.resolution.y;
    G.cam.pos = mk3(cam.position.x, cam.position.y, cam.position.z);
    G.cam.view = mk3(cam.view.x, cam.view.y, cam.view.z);
    G.cam.up = mk3(cam.up.x, cam.up.y, cam.up.z);
    G.cam.right = mk3(cam.right.x, cam.right.y, cam.right.z);
    G.cam.plx = cam.pixelLength.x; G.cam.ply = cam.pixelLength.y;
    G.trace_depth = trace_depth; G.n_parts = std::max(1, n_parts); G.part = part; G.tile_rows = std::max(1, tile_rows); G.n_pix = n_pix; G.late_stores = 0;
    const Gather Ga = {gen ? nullptr : ws.seg_pref, gen ? nullptr : ws.seg_first, nullptr};
    const int* d_n = gen ? nullptr : ws.counts + depth;
    const SegTail tail = {ws.seg_done, ws.seg_pref, ws.seg_first, ws.counts + depth + 1, last ? ws.iter_dev : nullptr, ws.iter_stride};
    const int sbias = ws.slot_bias + (depth >= 1 ? ws.dbg_shift_deep : ws.dbg_shift0);
    const LookBack lb = {ws.look, ws.look_gave_up, ws.look_limit};
    const int cap = gen ? n_pix * ws.spp : capacity;
    mark(gen ? 0 : 2, true);
    int rc;
#define PTK_TS(M, GENF) launch_trace_shade_t<M, GENF>(S, G, iter, ws.iter_dev, depth, cap, d_n, cur, Ga, nxt, ws.agg_b, lb, film_of(img, ws), ws.spp, sbias, tail, ws.counts, ws.ray_total, st)
    switch (mode) {
    case PTK_MODE_BSDF: rc = gen ? PTK_TS(PTK_MODE_BSDF, true) : PTK_TS(PTK_MODE_BSDF, false); break;
    case PTK_MODE_DIRECTLI: rc = gen ? PTK_TS(PTK_MODE_DIRECTLI, true) : PTK_TS(PTK_MODE_DIRECTLI, false); break;
    case PTK_MODE_MIS: rc = gen ? PTK_TS(PTK_MODE_MIS, true) : PTK_TS(PTK_MODE_MIS, false); break;
    default: rc = PTK_E_BADARG;
    }
#undef PTK_TS
    mark(gen ? 0 : 2, false);
    return rc;
}
// all bounces from `depth0` on in one single-workgroup launch (k_tail); cur = the stream entering bounce depth0
int launch_tail(const SceneView& S_in, int mode, int iter, int depth0, int trace_depth, PtkPaths cur, PtkPaths nxt, float* img,
                const Workspace& ws, hipStream_t st, const PtkHits* hits0, const int* hr0) {
    SceneView S = S_in;
    TailArgs A;
    A.have_hits = hits0 != nullptr ? 1 : 0;
    A.compact = (hits0 != nullptr && S_in.n_tris == 0 && trace_compact(S_in, mode)) ? 1 : 0;
    A.hits0 = hits0 ? *hits0 : PtkHits{nullptr, nullptr, nullptr}; A.hr0 = hr0;
    if (hits0) S.slim_hits = (S.n_tex == 0 && mode != PTK_MODE_MIS) ? 1 : 0;      // as the launch that wrote them
    A.iter_arg = iter; A.iter_ptr = ws.iter_dev; A.depth0 = depth0; A.trace_depth = trace_depth;
    A.counts = ws.counts; A.cur = cur; A.nxt = nxt;
    // the last full-width launch was a hit-dense k_shade_trace (every fused layout but SPLIT): its windows = ceil(hcounts[depth0 - 1] / 256)
    const bool hit_dense = hits0 != nullptr && ws.hcounts && !(S_in.n_tris > 0 && !ws.inline_bvh && ws.queue && ws.list);
    A.nshaded = (depth0 > 0 ? (hit_dense ? ws.hcounts : ws.counts) + (depth0 - 1) : ws.counts);
    A.hit_dense = (hit_dense && depth0 > 0) ? 1 : 0; A.hit_pref = ws.hit_pref; A.first_h = ws.seg_first_h; A.nhits = ws.hcounts ? ws.hcounts + depth0 : nullptr;
    A.hcounts = ws.hcounts;
    A.win = (A.compact && ws.wave_windows && S_in.lds_geoms > 0) ? PTK_WW : PTK_BLOCK;      // (as launch_shade_trace chose the kernel)
    A.cum_prev = ws.cum_prev; A.cum_out = ws.cum_out; A.cum_stride = ws.cum_stride; A.ring_div = ws.ring_div;
    const bool seg = ws.seg_pref && ws.seg_first && depth0 > 0;
    A.G = Gather{seg ? ws.seg_pref : nullptr, seg ? ws.seg_first : nullptr, nullptr};
    A.film = film_of(img, ws); A.spp = ws.spp; A.slot_bias = ws.slot_bias; A.dbg_shift0 = ws.dbg_shift0; A.dbg_shift_deep = ws.dbg_shift_deep;
    A.ray_total = ws.ray_total; A.iter_advance = ws.iter_dev; A.iter_stride = ws.iter_stride;
    switch (mode) {
    case PTK_MODE_BSDF: k_tail<PTK_MODE_BSDF><<<1, PTK_TAIL_BLOCK, stage_bytes(S), st>>>(S, A); break;
    case PTK_MODE_DIRECTLI: k_tail<PTK_MODE_DIRECTLI><<<1, PTK_TAIL_BLOCK, stage_bytes(S), st>>>(S, A); break;
    case PTK_MODE_MIS: k_tail<PTK_MODE_MIS><<<1, PTK_TAIL_BLOCK, stage_bytes(S), st>>>(S, A); break;
    default: return PTK_E_BADARG;
    }
    return launch_rc();
}
bool g_light_fix = false;
void set_light_fix(bool on) { g_light_fix = on; }
bool light_fix_enabled() { return g_light_fix; }
int launch_sum_planes(const float* planes, int n_planes, size_t plane_len, float* out, hipStream_t st) {
    const int g = (int)((plane_len + PTK_BLOCK - 1) / PTK_BLOCK);
    k_sum_planes<<<g, PTK_BLOCK, 0, st>>>(planes, n_planes, plane_len, out);
    return launch_rc();
}
int launch_merge_planes(float* image, float* planes, unsigned char* flags, int n_planes, int n_pixels, hipStream_t st) {
    const int n_tiles = (n_pixels + 63) / 64;
    const int g = (n_tiles + PTK_BLOCK / 64 - 1) / (PTK_BLOCK / 64);
    k_merge_planes<<<g, PTK_BLOCK, 0, st>>>(image, planes, flags, n_planes, n_pixels, n_tiles);
    return launch_rc();
}

} // namespace ptk

// ------------------------------------------------------------------------------------------------------------------
// C ABI: kernel-level entry points (include/ptk.h)
// ------------------------------------------------------------------------------------------------------------------
extern "C" {

int ptk_generate_rays(const PtkCamera* cam, int iter, int trace_depth, PtkPaths paths, void* stream) {
    if (!cam || !paths.o_pix || !paths.d_bnc || !paths.col_pdf || cam->res[0] <= 0 || cam->res[1] <= 0) return PTK_E_BADARG;
    const int n = cam->res[0] * cam->res[1];
    k_generate<<<grid_for(n), PTK_BLOCK, 0, (hipStream_t)stream>>>(to_dev_camera(*cam), iter, nullptr, trace_depth, paths, 1, 0, 1, n, 1, nullptr);
    return launch_rc();
}

int ptk_intersect(const PtkScene* s, int n, const int* d_n, PtkPaths paths, PtkHits hits, int* valid,
                  unsigned long long* stats, float* img, void* stream) {
    if (!s || !s->scene || !s->scene->tempDevScene.uploaded) return PTK_E_NOSCENE;
    if (n < 0 || !paths.o_pix || !paths.d_bnc || !hits.t_n || !hits.p_mat || !hits.uv_ids || !valid) return PTK_E_BADARG;
    if (n == 0) return 0;
    if (!g_abi_queue) {
        hipError_t e = hipMalloc(&g_abi_queue, sizeof(unsigned)); if (e != hipSuccess) return (int)e;
        e = hipMemset(g_abi_queue, 0, sizeof(unsigned)); if (e != hipSuccess) return (int)e;
    }
    if (n > g_abi_list_cap) {
        for (int** p : {&g_abi_list, &g_abi_hflag, &g_abi_hagg}) { if (*p) (void)hipFree(*p); *p = nullptr; }
        g_abi_list_cap = 0;
        hipError_t e = hipMalloc(&g_abi_list, sizeof(int) * (size_t)n); if (e != hipSuccess) return (int)e;
        if ((e = hipMalloc(&g_abi_hflag, sizeof(int) * (size_t)n)) != hipSuccess) return (int)e;
        if ((e = hipMalloc(&g_abi_hagg, sizeof(int) * (size_t)(grid_for(n) + 1))) != hipSuccess) return (int)e;
        g_abi_list_cap = n;
    }
    if (stats && s->scene->tempDevScene.n_copies != 6) {      // the counters describe the reference's own walk, direction order included
        const int rc6 = s->scene->tempDevScene.upload_threaded(*s->scene, 6);
        if (rc6) return rc6;
    }
    const SceneView S = dev_view(*s->scene);
    // the render loop's own layout: analytic primitives + root-box test, work list, BVH walk (the 8-wide collapsed tree; with `stats`
    // the reference's threaded walk, whose visits the counters describe), finalize
    int rc = launch_intersect_stage(S, n, d_n, paths, hits, valid, nullptr, nullptr, stats, g_abi_queue, g_abi_list,
                                    g_abi_hflag, g_abi_hagg, false, (hipStream_t)stream, [](int, bool) {}, Gather{nullptr, nullptr, nullptr}, PTK_WALK_WIDE, 4, pick_wide_pool(0, 1));
    if (rc || !img || S.env_tex < 0) return rc;
    if (!paths.col_pdf) return PTK_E_BADARG;
    k_env_miss<<<grid_for(n), PTK_BLOCK, 0, (hipStream_t)stream>>>(S, n, d_n, paths, valid, FilmArgs{img, nullptr, 0u, 0u, 0u, 0u}, 0, nullptr);
    return launch_rc();
}

int ptk_compact(int n, const int* d_n, const int* valid, PtkPaths paths_in, PtkHits hits_in,
                PtkPaths paths_out, PtkHits hits_out, int* d_count_out, void* stream) {
    if (n < 0 || !valid) return PTK_E_BADARG;
    if (paths_in.o_pix && (!paths_in.d_bnc || !paths_in.col_pdf || !paths_out.o_pix || !paths_out.d_bnc || !paths_out.col_pdf)) return PTK_E_BADARG;
    if (hits_in.t_n && (!hits_in.p_mat || !hits_in.uv_ids || !hits_out.t_n || !hits_out.p_mat || !hits_out.uv_ids)) return PTK_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int g = grid_for(n);
    if (g == 0) { if (d_count_out) return hip_rc(hipMemsetAsync(d_count_out, 0, sizeof(int), st)); return 0; }
    if (g > g_compact_agg_cap) {
        if (g_compact_agg) { hipError_t e = hipFree(g_compact_agg); if (e != hipSuccess) return (int)e; }
        g_compact_agg = nullptr; g_compact_agg_cap = 0;
        hipError_t e = hipMalloc(&g_compact_agg, sizeof(int) * (size_t)g);
        if (e != hipSuccess) return (int)e;
        g_compact_agg_cap = g;
    }
    k_count<<<g, PTK_BLOCK, 0, st>>>(n, d_n, valid, g_compact_agg);
    int rc = launch_rc(); if (rc) return rc;
    k_scatter<<<g, PTK_BLOCK, 0, st>>>(n, d_n, valid, g_compact_agg, paths_in, hits_in, paths_out, hits_out, d_count_out);
    return launch_rc();
}

int ptk_shade(const PtkScene* s, int mode, int iter, int depth, int n, const int* d_n, PtkHits hits,
              PtkPaths paths, int* valid, float* img, void* stream) {
    if (!s || !s->scene || !s->scene->tempDevScene.uploaded) return PTK_E_NOSCENE;
    if (n < 0 || !paths.o_pix || !paths.d_bnc || !paths.col_pdf || !hits.t_n || !hits.p_mat || !hits.uv_ids || !valid || !img) return PTK_E_BADARG;
    if (n == 0) return 0;
    const SceneView S = dev_view(*s->scene);
    const int g = grid_for(n);
    hipStream_t st = (hipStream_t)stream;
    switch (mode) {
    case PTK_MODE_BSDF: k_shade<PTK_MODE_BSDF><<<g, PTK_BLOCK, 0, st>>>(S, iter, depth, n, d_n, hits, paths, valid, img); break;
    case PTK_MODE_DIRECTLI: k_shade<PTK_MODE_DIRECTLI><<<g, PTK_BLOCK, stage_bytes(S), st>>>(S, iter, depth, n, d_n, hits, paths, valid, img); break;
    case PTK_MODE_MIS: k_shade<PTK_MODE_MIS><<<g, PTK_BLOCK, stage_bytes(S), st>>>(S, iter, depth, n, d_n, hits, paths, valid, img); break;
    default: return PTK_E_BADARG;
    }
    return launch_rc();
}

int ptk_tonemap(const float* img, int res_x, int res_y, int iter, unsigned char* pbo, void* stream) {
    if (!img || !pbo || res_x <= 0 || res_y <= 0 || iter <= 0) return PTK_E_BADARG;
    const int P = res_x * res_y;
    k_tonemap<<<grid_for(P), PTK_BLOCK, 0, (hipStream_t)stream>>>(img, P, iter, reinterpret_cast<uchar4*>(pbo));
    return launch_rc();
}

#ifdef PTK_TEST_HOOKS
/* test-only: see k_exact_check.  bad8 (host): mismatch counts of the 8 checks over n pseudo-random operand sets */
int ptk_debug_exact_math(unsigned long long n, unsigned seed, unsigned long long* bad8, unsigned* samples128) {
    if (!bad8 || n == 0 || n > (1ull << 36)) return PTK_E_BADARG;
    unsigned long long* d = nullptr;
    hipError_t e = hipMalloc(&d, 8 * sizeof(unsigned long long) + 128 * sizeof(unsigned)); if (e != hipSuccess) return (int)e;
    if ((e = hipMemset(d, 0, 8 * sizeof(unsigned long long) + 128 * sizeof(unsigned))) != hipSuccess) { (void)hipFree(d); return (int)e; }
    uint32_t* smp = reinterpret_cast<uint32_t*>(d + 8);
    const unsigned long long chunk = 1ull << 28;                      // bounded launches: a grid of 2^20 blocks each
    for (unsigned long long done = 0; done < n; done += chunk) {
        const unsigned long long m = n - done < chunk ? n - done : chunk;
        k_exact_check<<<(unsigned)((m + PTK_BLOCK - 1) / PTK_BLOCK), PTK_BLOCK>>>(m, seed + (unsigned)(done >> 28) * 7919u, d, smp);
        if ((e = hipGetLastError()) != hipSuccess) { (void)hipFree(d); return (int)e; }
    }
    e = hipMemcpy(bad8, d, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    if (e == hipSuccess && samples128) e = hipMemcpy(samples128, smp, 128 * sizeof(unsigned), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return hip_rc(e);
}
#endif // PTK_TEST_HOOKS

#ifdef PTK_PHASE_TIMING
/* diagnostic build only: out[2][8] = cycles per phase summed over all waves ([7] = wave visits) for k_intersect (0) and
 * k_shade_binned (1); clears the buffer */
int ptk_debug_phases(unsigned long long* out) {
    static unsigned long long host[2][PTK_PHASE_SLOTS][8];
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_phase_buf), sizeof(host)) != hipSuccess) return 1;
    for (int k = 0; k < 2; ++k) for (int j = 0; j < 8; ++j) { unsigned long long t = 0; for (int w = 0; w < PTK_PHASE_SLOTS; ++w) t += host[k][w][j]; out[k * 8 + j] = t; }
    for (int k = 0; k < 2; ++k) for (int w = 0; w < PTK_PHASE_SLOTS; ++w) for (int j = 0; j < 8; ++j) host[k][w][j] = 0;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase_buf), host, sizeof(host)) == hipSuccess ? 0 : 1;
}
#endif

} // extern "C"
#endif // PTK_TU_BSDF
