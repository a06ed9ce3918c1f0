// ptk_device.hpp — device-side math of the path-tracing hot path for gfx950 (wave64).
//
// Scalar-per-lane code: one path per lane, scene constants addressed with wave-uniform
// indices so the compiler keeps them in SGPRs (s_load), per-lane state in VGPRs; records a
// lane picks for itself (its candidate primitive, its material) are read from LDS copies.
//
// NUMERICS CONTRACT.  This file is compiled with -ffp-contract=off and the default
// correctly-rounded fp32 divide/sqrt.  Every expression is written in the evaluation order
// of the reference's host/device headers (file:line cited per function; GLM 0.9.6.3 order
// for dot / mat*vec / normalize / mix), so each fp32 result is the IEEE-754 result of the
// same operation sequence.  sin/cos use the "ptk sincos" specification (below) instead of
// a platform libm: + - * only, so the result is reproducible on any IEEE machine.
//
// Deliberate, value-preserving simplifications (results compare == to the literal form):
//   * mat4 * (v,0): the "+ m[3]*0" term of glm's mat4*vec4 is dropped (adds +-0);
//   * cross((0,0,1), w) is written as (-w.y, w.x, 0) (differs only in the sign of a zero);
//   * triangle attributes are interpolated once for the final closest hit instead of at
//     every accepted candidate (pure function of triangle, u, v);
//   * next-event estimation towards a cube light is skipped: in the reference it always
//     evaluates to 0 or NaN->0 (lightSample.h:171-248 assigns no position/pdf for cubes),
//     unless the opt-in fix (SceneView::light_fix) is on;
//   * analytic primitives that provably miss are not tested (analytic_candidates), and the
//     closest one is chosen by "smaller t, then smaller index" — the winner of the
//     reference's first-found-wins loop in index order, for any visiting order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

namespace ptk {

#define PTK_PI      3.1415926535897932384626422832795028841971f
#define PTK_TWO_PI  6.2831853071795864769252867665590057683943f
#define PTK_ROUGHNESS_MIN 1e-3f     // static_cast<float>(ROUGHNESS_MIN), utilities.h:28
#define PTK_ROUGHNESS_MAX 1.f
#define PTK_DI __host__ __device__ __forceinline__

struct f3 { float x, y, z; };
struct f2 { float x, y; };

PTK_DI f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
typedef float ptk_v2f __attribute__((ext_vector_type(2)));
#if defined(__HIP_DEVICE_COMPILE__)
// x and y of a 3-vector ride in one packed operation (v_pk_add/mul_f32: per-component IEEE), z in a scalar one: 2 instructions per 3-vector op
PTK_DI f3 pk3(ptk_v2f xy, float z) { return mk3(xy.x, xy.y, z); }
PTK_DI ptk_v2f xy_of(f3 a) { ptk_v2f r = {a.x, a.y}; return r; }
PTK_DI f3 operator+(f3 a, f3 b) { return pk3(xy_of(a) + xy_of(b), a.z + b.z); }
PTK_DI f3 operator-(f3 a, f3 b) { return pk3(xy_of(a) - xy_of(b), a.z - b.z); }
PTK_DI f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
PTK_DI f3 operator*(f3 a, f3 b) { return pk3(xy_of(a) * xy_of(b), a.z * b.z); }
PTK_DI f3 operator*(f3 a, float s) { ptk_v2f ss = {s, s}; return pk3(xy_of(a) * ss, a.z * s); }
PTK_DI f3 operator*(float s, f3 a) { ptk_v2f ss = {s, s}; return pk3(ss * xy_of(a), s * a.z); }
#else
PTK_DI f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PTK_DI f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PTK_DI f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
PTK_DI f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
PTK_DI f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
PTK_DI f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
#endif
PTK_DI f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }

// ---- exact fast paths for IEEE divide / sqrt ("same operations, fewer instructions") -----------------------------------------
// The compiler expands a correctly rounded fp32 a / d into 11 instructions: v_div_scale x2 (operand pre-scaling), v_rcp, two
// Newton steps on the reciprocal (e = fma(-d, y, 1); y = fma(e, y, y)), q = a * y, two residual corrections (r = fma(-d, q, a);
// q = fma(r, y, q)), the last one as v_div_fmas, then v_div_fixup; and sqrtf into 16: a pre-scaling of tiny arguments, v_sqrt,
// one step down / one step up tested with two exact residuals, the rescaling, a class test for 0 / inf.  The scaling and fix-up
// steps are the identity when the operands are normal numbers of moderate magnitude — V_DIV_SCALE_F32 scales only when an operand
// or the quotient or 1/d is (nearly) denormal, when the exponents differ by >= 96 or when the numerator's exponent is <= 23; the
// sqrt prologue only below 2^-96 — so for operands in [2^-40, 2^40] (divide) / [2^-80, 2^80] (sqrt, 1/sqrt) the sequences
// below, which are THE SAME OPERATIONS without those steps, return the same bits as `/` and sqrtf.  What that buys:
//   * one reciprocal refinement shared by all numerators over one denominator (the two slab planes of an axis; bsdf.xyz / pdf);
//   * the per-numerator steps as packed operations (v_pk_mul_f32 / v_pk_fma_f32 are per-component IEEE): x and y of a 3-vector,
//     or two numerators, ride in one instruction;
//   * 1 / sqrt(x) (normalize) under one range test instead of two special-case ladders.
// Operands outside the range (zeros, denormals, huge, inf, NaN) take the plain `/` / sqrtf, i.e. the compiler's full expansion.
// tests/test_gpu_exact_math.py compares every helper with `/` and sqrtf on the device, bit for bit, on ~1e8 operand sets including
// the range edges.  On the host (oracle-side builds, the scene layer) the helpers ARE `/` and sqrtf.
#define PTK_EX_LO 9.094947017729282e-13f        // 2^-40
#define PTK_EX_HI 1.099511627776e12f            // 2^40
#define PTK_EX_LO2 8.271806125530277e-25f       // 2^-80
#define PTK_EX_HI2 1.2089258196146292e24f       // 2^80
#if defined(__HIP_DEVICE_COMPILE__)
PTK_DI ptk_v2f splat2(float s) { ptk_v2f r = {s, s}; return r; }
PTK_DI ptk_v2f fma2(ptk_v2f a, ptk_v2f b, ptk_v2f c) { return __builtin_elementwise_fma(a, b, c); }
PTK_DI float rcp_refined(float d) { const float y = __builtin_amdgcn_rcpf(d); const float e = __builtin_fmaf(-d, y, 1.0f); return __builtin_fmaf(e, y, y); }
PTK_DI ptk_v2f rcp_refined2(ptk_v2f d) {
    ptk_v2f y = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const ptk_v2f e = fma2(-d, y, splat2(1.0f));
    return fma2(e, y, y);
}
// the quotient steps of the expansion, given the refined reciprocal y of d
PTK_DI float div_steps(float a, float d, float y) {
    float q = a * y;
    float r = __builtin_fmaf(-d, q, a); q = __builtin_fmaf(r, y, q);
    r = __builtin_fmaf(-d, q, a); return __builtin_fmaf(r, y, q);
}
PTK_DI ptk_v2f div_steps2(ptk_v2f a, ptk_v2f d, ptk_v2f y) {
    ptk_v2f q = a * y;
    ptk_v2f r = fma2(-d, q, a); q = fma2(r, y, q);
    r = fma2(-d, q, a); return fma2(r, y, q);
}
PTK_DI float inv_steps(float d, float y) {          // 1.0f / d: the first product 1 * y is y itself
    float r = __builtin_fmaf(-d, y, 1.0f); float q = __builtin_fmaf(r, y, y);
    r = __builtin_fmaf(-d, q, 1.0f); return __builtin_fmaf(r, y, q);
}
PTK_DI float sqrt_steps(float x) {                  // x in [2^-96, FLT_MAX]
    const float s = __builtin_amdgcn_sqrtf(x);
    const float dn = __int_as_float(__float_as_int(s) - 1), up = __int_as_float(__float_as_int(s) + 1);
    const float e_dn = __builtin_fmaf(-dn, s, x), e_up = __builtin_fmaf(-up, s, x);
    float t = (e_dn <= 0.0f) ? dn : s;
    t = (e_up > 0.0f) ? up : t;
    return t;
}
#ifdef PTK_EX_NOFALLBACK      // TIMING EXPERIMENTS ONLY (tools/diag): the range tests compiled out; never shipped
PTK_DI bool ex_in(float, float, float) { return true; }
#else
PTK_DI bool ex_in(float x, float lo, float hi) { return x >= lo && x <= hi; }      // false for NaN
#endif
#endif
// a / d
PTK_DI float div_exact(float a, float d) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (ex_in(__builtin_fabsf(a), PTK_EX_LO, PTK_EX_HI) && ex_in(__builtin_fabsf(d), PTK_EX_LO, PTK_EX_HI)) return div_steps(a, d, rcp_refined(d));
#endif
    return a / d;
}
// 1 / d
PTK_DI float inv_exact(float d) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (ex_in(__builtin_fabsf(d), PTK_EX_LO, PTK_EX_HI)) return inv_steps(d, rcp_refined(d));
#endif
    return 1.0f / d;
}
// sqrtf(x)
PTK_DI float sqrt_exact(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (ex_in(x, PTK_EX_LO2, PTK_EX_HI2)) return sqrt_steps(x);
#endif
    return sqrtf(x);
}
// 1 / sqrtf(x) (two correctly rounded steps, as glm::normalize / inversesqrt do it)
PTK_DI float inv_sqrt_exact(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (ex_in(x, PTK_EX_LO2, PTK_EX_HI2)) { const float s = sqrt_steps(x); return inv_steps(s, rcp_refined(s)); }
#endif
    return 1.0f / sqrtf(x);
}
// (a.x / d, a.y / d, a.z / d)
PTK_DI f3 div3_exact(f3 a, float d) {
#if defined(__HIP_DEVICE_COMPILE__)
    // explicit range compares, each false for a NaN (min / max chains are not NaN-safe: in IEEE mode a signalling NaN turns the partial
    // result into a quiet NaN, which the next min / max drops together with the operands it had absorbed — found by k_exact_check)
    if (ex_in(__builtin_fabsf(a.x), PTK_EX_LO, PTK_EX_HI) && ex_in(__builtin_fabsf(a.y), PTK_EX_LO, PTK_EX_HI) &&
        ex_in(__builtin_fabsf(a.z), PTK_EX_LO, PTK_EX_HI) && ex_in(__builtin_fabsf(d), PTK_EX_LO, PTK_EX_HI)) {
        const float y = rcp_refined(d);
        const ptk_v2f q = div_steps2(xy_of(a), splat2(d), splat2(y));
        return mk3(q.x, q.y, div_steps(a.z, d, y));
    }
#endif
    return mk3(a.x / d, a.y / d, a.z / d);
}
PTK_DI float dot(f3 a, f3 b);
PTK_DI float length(f3 v) { return sqrt_exact(dot(v, v)); }
PTK_DI f3 normalize(f3 v) { return v * inv_sqrt_exact(dot(v, v)); }              // glm: v * inversesqrt(dot(v, v)), inversesqrt = 1 / sqrt
PTK_DI float gmin(float x, float y) { return x < y ? x : y; }          // glm::min
PTK_DI float gmax(float x, float y) { return x > y ? x : y; }          // glm::max
PTK_DI float smax(float a, float b) { return a < b ? b : a; }          // std::max
PTK_DI float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
#if defined(__HIP_DEVICE_COMPILE__)
PTK_DI float dot(f3 a, f3 b) { const ptk_v2f p = xy_of(a) * xy_of(b); return (p.x + p.y) + a.z * b.z; }      // (ax*bx + ay*by) + az*bz, the products packed
#else
PTK_DI float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
#endif
PTK_DI f3 cross(f3 x, f3 y) { return mk3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
PTK_DI float length(f3 v);
PTK_DI f3 normalize(f3 v);
PTK_DI f3 reflect(f3 I, f3 N) { return I - N * dot(N, I) * 2.0f; }
PTK_DI float mixf(float x, float y, float a) { return x + a * (y - x); }
PTK_DI f3 mixv(f3 x, f3 y, float a) { return x + a * (y - x); }
PTK_DI f3 mixvv(f3 x, f3 y, f3 a) { return x + a * (y - x); }
PTK_DI float comp(f3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

// ---- ptk sincos specification (same text as oracle/pt_oracle.cpp, implemented independently) -------
//   fk = floor(x * 2/pi + 0.5);  r = ((x - fk*P1) - fk*P2) - fk*P3;  z = r*r
//   s = r + r*z*(S1 + z*(S2 + z*S3));  c = (1 - 0.5 z) + z*z*(C1 + z*(C2 + z*C3))
//   quadrant fk&3: 0:(s,c) 1:(c,-s) 2:(-s,-c) 3:(-c,s)
PTK_DI void sincos_ptk(float x, float& sn, float& cs) {
    float fk = floorf(x * 0.636619772367581343f + 0.5f);
    float r = ((x - fk * 1.5703125f) - fk * 4.837512969970703125e-4f) - fk * 7.54978995489188216e-8f;
    float z = r * r;
    float s = r + r * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
    float c = (1.0f - 0.5f * z) + z * z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
    int q = ((int)fk) & 3;
    float a = (q & 1) ? c : s;
    float b = (q & 1) ? s : c;
    sn = (q & 2) ? -a : a;
    cs = (q == 1 || q == 2) ? -b : b;
}

// ---- RNG: utilhash (intersections.h:12-20), makeSeededRandomEngine (pathtrace.cu:25-29),
//      thrust minstd_rand + uniform_real_distribution<float> ------------------------------------------
PTK_DI uint32_t utilhash(uint32_t a) {
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}
struct Rng { uint32_t x; };
PTK_DI Rng make_rng(int iter, int index, int depth) {
    uint32_t h = utilhash(0x80000000u | ((uint32_t)depth << 22) | (uint32_t)iter) ^ utilhash((uint32_t)index);
    Rng r; r.x = h % 2147483647u; if (r.x == 0u) r.x = 1u; return r;
}
PTK_DI float sample1D(Rng& r) {
    // x <- 48271 x mod (2^31 - 1); the 64-bit product form equals thrust's Schrage form (exact integers)
    uint64_t p = (uint64_t)r.x * 48271ull;
    uint32_t lo = (uint32_t)(p & 0x7fffffffull), hi = (uint32_t)(p >> 31);
    uint32_t x = lo + hi; if (x >= 2147483647u) x -= 2147483647u;
    r.x = x;
    return (float)(x - 1u) / 2147483648.0f;
}
// sample2D (utilities.h:54-56): glm::vec2(sample1D(s), sample1D(s)) — unspecified argument order in C++; the reference's
// nvcc device build draws the FIRST argument first, settled against its own 20-spp gallery renders (DESIGN.md §2,
// tests/test_gallery.py).
PTK_DI f2 sample2D(Rng& r) { f2 o; o.x = sample1D(r); o.y = sample1D(r); return o; }

// ---- device scene tables ---------------------------------------------------------------------------
// Analytic primitive, 36 dwords.  Matrices as 3 rows x 4 (row-major) of glm's column-major mat4: m[r*4+c] = M[c][r].
struct DevGeom {
    int type, materialid;
    float inv[12];      // inverseTransform rows 0..2
    float fwd[12];      // transform rows 0..2
    float invT[9];      // upper 3x3 of invTranspose, row-major
    int diag;           // 1: the 3x3 parts of inv, fwd and invT are diagonal (all off-diagonal entries are +-0)
};
// World-space bounds of one analytic primitive for the conservative candidate test (analytic_candidates), 8 dwords, slab
// pairs adjacent: {lo.x, hi.x, lo.y, hi.y, lo.z, hi.z, -, -}.  The box encloses the transformed unit cube, grown on the host
// by PTK_CAND_ABS + PTK_CAND_REL * max|corner coordinate| (dev_scene.hip).  A primitive whose transform is not finite
// gets (-inf, +inf) slabs (always a candidate).
struct DevGeomBox { float s[6]; float qa, qb; };      // spheres: extra pad qa + qb * |origin|^2 (see PTK_CAND_QUAD), cubes 0
// Per chunk of 32 primitives: bit k set <=> primitive base + k is a cube / a sphere / has diagonal 3x3 transforms.
struct DevGeomMask { unsigned cube, sphere, diag, round; };   // round: spheres under a similarity transform (see PTK_CAND_QUAD)
// Margins of the candidate test.  Measured need (tests/test_candidate_margin.py: adversarial grazing rays, exact fp32 test
// vs this test): no exact hit is ever culled from REL = 1e-7 on; shipped values are 20x that.
#define PTK_CAND_ABS 1e-5f
#define PTK_CAND_REL 2e-6f
// Spheres only: the exact test's radicand vd*vd - (dot(o,o) - 0.25) cancels catastrophically for distant origins; it accepts
// rays that pass the unit sphere at up to ~3.6e-7 * |o_obj|^2 (object units) — quadratic in the distance.  The box of a sphere
// is therefore grown per ray by QUAD * s_max / s_min^2 * |o - c|^2, bounded by 2 (|o|^2 + |c|^2): qa = 2 Q |c|^2, qb = 2 Q.
// Measured need: QUAD = 3e-7; shipped 13x that.
#define PTK_CAND_QUAD 4e-6f
#define PTK_CAND_QUAD2 2e-5f
// "Round" spheres (uniform scale: the primitive IS a world-space sphere) get a tighter test than the box: DevGeomBox then holds
// {c.x, c.y, c.z, Rp^2, 2 Rp, -} with Rp = radius + the box margins, and a ray is a candidate unless its line passes the centre
// at a squared distance > (Rp + pr)^2 + QUAD2 * |o - c|^2, or the centre lies behind an origin that is outside that sphere.
// QUAD2 covers the exact test's cancellation (3.6e-7 |o - c|^2) plus this test's own (2.4e-7): measured need 5e-7, shipped 40x.
// (A sphere fills 52 % of its box: in cornell.txt the large sphere's box catches 50 % of the rays, the sphere 9-40 %.)
struct DevMaterial {    // 20 dwords
    int type;
    float albedo[3], albedo_s[3];
    float rough, metal, ior;     // constant-sampler rough/metal already clamped like Material::BSDF does (material.h:486-487)
    int albedo_tex, rough_tex, metal_tex, normal_tex;      // texture index per sampler, -1 = the constant above
    float albedo_pi[3];          // albedo_s / pi, the same IEEE fp32 division done once on the host (Lambertian bsdf, material.h:58)
    float pad[3];
};
struct DevTex { int w, h; unsigned offset; int pad; };    // devTexObj (image.h:53-66): texels at texels[3 * (offset + y*w + x)]
// One node of a direction-ordered threaded BVH, 32 B = two 16-B loads.  a/b encode the reference's (start,end,hit,miss):
//   a >=  0 : inner node, a = hit link (first child in this direction's order), b = miss link
//   a == -1 : empty leaf (start == end; only the root of a triangle-free scene), b = next (hit == miss for leaves)
//   a <= -2 : leaf holding triangle -(a + 2)  (MAX_PRIM = 1, BVH.h:5), b = next
struct DevNode {
    float minx, miny, minz, maxx, maxy, maxz;
    int a, b;
};
// 8-wide collapsed node of the SAME binary tree (host builder + the parity argument: wide_bvh.h), 144 B = 9 x uint4:
//   q[0]     = {p.x, p.y, p.z, ex | ey << 8 | ez << 16 | n << 24}   quantisation origin, biased exponents of the per-axis
//                                                                    scale 2^(e-127), number of children n (1..8)
//   q[1 + c] = child c: {qlo.x | qlo.y << 8 | qlo.z << 16 | qhi.x << 24,
//                        qhi.y | qhi.z << 8 | before[0] << 16 | before[1] << 24,
//                        before[2] | before[3] << 8 | before[4] << 16 | before[5] << 24,
//                        ref}
//              the child spans [p + qlo * 2^e, p + qhi * 2^e] (a superset of its exact box; unused slots are inverted);
//              before[d] = mask of the children a ray of direction tree d (pathtrace.cu:242-244) visits before this one;
//              ref = wide-node index, or 0x80000000 | leaf-record index.
// 16-wide variant (round 5, PTK_WIDE_WIDTH=16; SceneView::wide_width), 272 B = 17 x uint4: the same header (n = 1..16) and sixteen child records whose
// bytes 6..8 hold pos[0..5], four bits each — the child's POSITION in direction d's visiting order — instead of the before masks: a hit child's rank among
// the hit children is the number of hit children at smaller positions (the group ORs 1 << pos of its hit children into one 16-bit word).
// Leaf record, 64 B = 4 x uint4: exact fp32 box of the reference's leaf node, the triangle in edge form (as DevTriE), its index.
struct WideLeaf { float lo[3], hi[3], v0[3], e1[3], e2[3]; int tri; };
struct DevTriE { float v0[3], e1[3], e2[3], pad[3]; };                          // 48 B: v0 and the two edges (e = v - v0 as in Bounds3.hpp:300-301)
struct DevTriV { float v0[3], v1[3], v2[3]; };                                  // 36 B, positions (attribute interpolation, light sampling)
struct DevTriA { float n0[3], n1[3], n2[3]; float t0[2], t1[2], t2[2]; int geomIdx; float tangent[3]; };  // 76 B, shading attributes
struct DevLight { int geomID, triangleID, type; };

struct SceneView {
    const DevGeom* geoms;        int n_geoms;
    const DevGeomBox* gbox;      // [n_geoms]
    const DevGeomMask* gmask;    // [ceil(n_geoms / 32)]
    int lds_geoms;               // > 0: kernels stage the transforms of all n_geoms primitives in LDS (stage_geoms) and run the
                                 // candidate loops; 0: wave-uniform loops over every primitive (scalar loads)
    const DevMaterial* mats;     int n_mats;
    const DevNode* nodes;        int bvh_size;   // [n_copies][bvh_size]: six direction-ordered copies, or the direction-0 copy alone (round 4)
    int n_copies;                // 6 or 1
    const uint32_t* ranks;       // [6][n_tris] with one copy: rank of triangle t in direction d's leaf order (ranks[d * n_tris + t]); null with six
    const DevTriE* trie;         // traversal form of the triangles
    const uint4* wnodes;         int n_wnodes;   // 8-wide collapsed form of the same tree (WideNode, 9 x uint4 = 144 B each); null = none
    int wide_width;              // 8, or 16: the 16-wide variant (17 x uint4 per node)
    const uint4* wleaves;        // WideLeaf records (4 x uint4 each), children of one wide node consecutive
    const DevTriV* triv;         const DevTriA* tria; int n_tris;
    const DevLight* lights;      int n_lights;
    const DevTex* tex;           const float* texels;   int env_tex;     // env_tex = envMapID (-1: none)
    int n_tex;                   // number of textures
    int light_fix;               // opt-in fixed cube lights (ptk_set_light_fix), 0 = reference behaviour
    int slim_hits;               // render pipeline only: 1 = the (uv, geomID, triangleID) third of the hit record is unused
};

// Wave-uniform load of one primitive record into SGPRs.  The scene tables are immutable while a kernel runs, but the
// compiler cannot prove that for a pointer inside a by-value struct, and then falls back to VECTOR loads of uniform
// addresses (33 global_load_dword per primitive per wave).  Reading through the constant address space (4) makes it
// emit s_load_dwordx8/x16 instead, and the matrices become scalar operands of the VALU instructions.
PTK_DI DevGeom load_geom_uniform(const DevGeom* base, int i) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const uint32_t __attribute__((address_space(4)))* cptr;
    cptr p = (cptr)(const uint32_t*)(base + i);
    DevGeom g;
    uint32_t* dst = reinterpret_cast<uint32_t*>(&g);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(DevGeom) / 4); ++k) dst[k] = p[k];
    return g;
#else
    return base[i];
#endif
}

PTK_DI DevGeomBox load_gbox_uniform(const DevGeomBox* base, int i) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const uint32_t __attribute__((address_space(4)))* cptr;
    cptr p = (cptr)(const uint32_t*)(base + i);
    DevGeomBox g;
    uint32_t* dst = reinterpret_cast<uint32_t*>(&g);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(DevGeomBox) / 4); ++k) dst[k] = p[k];
    return g;
#else
    return base[i];
#endif
}

PTK_DI DevGeomMask load_gmask_uniform(const DevGeomMask* base, int i) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const uint32_t __attribute__((address_space(4)))* cptr;
    cptr p = (cptr)(const uint32_t*)(base + i);
    DevGeomMask g;
    g.cube = p[0]; g.sphere = p[1]; g.diag = p[2]; g.round = p[3];
    return g;
#else
    return base[i];
#endif
}

// vec3(M * vec4(v,1)) and vec3(M * vec4(v,0)) in glm's order (type_mat4x4.inl:617-628)
#if defined(__HIP_DEVICE_COMPILE__)
// rows 0 and 1 of the product ride in the two halves of packed operations (same products, same sums, same order per component)
PTK_DI ptk_v2f mk2(float a, float b) { ptk_v2f r = {a, b}; return r; }
PTK_DI f3 xform_point(const float* m, f3 v) {
    const ptk_v2f r = (mk2(m[0], m[4]) * splat2(v.x) + mk2(m[1], m[5]) * splat2(v.y)) + (mk2(m[2], m[6]) * splat2(v.z) + mk2(m[3], m[7]));
    return mk3(r.x, r.y, (m[8] * v.x + m[9] * v.y) + (m[10] * v.z + m[11]));
}
PTK_DI f3 xform_dir(const float* m, f3 v) {
    const ptk_v2f r = (mk2(m[0], m[4]) * splat2(v.x) + mk2(m[1], m[5]) * splat2(v.y)) + mk2(m[2], m[6]) * splat2(v.z);
    return mk3(r.x, r.y, (m[8] * v.x + m[9] * v.y) + (m[10] * v.z));
}
PTK_DI f3 xform_nrm(const float* m, f3 v) {
    const ptk_v2f r = (mk2(m[0], m[3]) * splat2(v.x) + mk2(m[1], m[4]) * splat2(v.y)) + mk2(m[2], m[5]) * splat2(v.z);
    return mk3(r.x, r.y, (m[6] * v.x + m[7] * v.y) + (m[8] * v.z));
}
#else
PTK_DI f3 xform_point(const float* m, f3 v) {
    return mk3((m[0] * v.x + m[1] * v.y) + (m[2] * v.z + m[3]),
               (m[4] * v.x + m[5] * v.y) + (m[6] * v.z + m[7]),
               (m[8] * v.x + m[9] * v.y) + (m[10] * v.z + m[11]));
}
PTK_DI f3 xform_dir(const float* m, f3 v) {
    return mk3((m[0] * v.x + m[1] * v.y) + (m[2] * v.z),
               (m[4] * v.x + m[5] * v.y) + (m[6] * v.z),
               (m[8] * v.x + m[9] * v.y) + (m[10] * v.z));
}
PTK_DI f3 xform_nrm(const float* m, f3 v) {
    return mk3((m[0] * v.x + m[1] * v.y) + (m[2] * v.z),
               (m[3] * v.x + m[4] * v.y) + (m[5] * v.z),
               (m[6] * v.x + m[7] * v.y) + (m[8] * v.z));
}
#endif

// Diagonal fast paths.  With exactly-zero off-diagonal entries, (m0*x + 0*y) + (0*z + m3) == m0*x + m3 and
// (m0*x + 0*y) + 0*z == m0*x for finite x,y,z (adding +-0 changes at most the sign of a zero), so these return values
// that compare == to the general forms above.  `diag` is wave-uniform (it comes from the SGPR-resident geom record).
PTK_DI f3 xform_point_d(const float* m, f3 v, int diag) {
    if (diag) return mk3(m[0], m[5], m[10]) * v + mk3(m[3], m[7], m[11]);      // (x, y) packed, z scalar (operator* / operator+)
    return xform_point(m, v);
}
PTK_DI f3 xform_dir_d(const float* m, f3 v, int diag) {
    if (diag) return mk3(m[0], m[5], m[10]) * v;
    return xform_dir(m, v);
}
PTK_DI f3 xform_nrm_d(const float* m, f3 v, int diag) {
    if (diag) return mk3(m[0], m[4], m[8]) * v;
    return xform_nrm(m, v);
}

// ---- primitives -----------------------------------------------------------------------------------
// The part of a primitive record the exact tests read (inverseTransform, transform, diag), as a per-lane register copy.
// Staged form: kernels copy every DevGeom into LDS once per block (stage_geoms) as {inv[12], fwd[12], invT[9], diag, materialid,
// pad} = 36 floats = 144 B per primitive; each lane fetches the matrices of ITS current candidate with ds_read_b128, and the hit
// record reads invT / materialid from the same copy (analytic_record_staged).
struct GeomXf { float inv[12], fwd[12]; int diag; };
#define PTK_XF_FLOATS 36            // staged record: inv[12], fwd[12], invT[9], diag, materialid, pad (ptk_kernels.hip: stage_geoms)
#define PTK_LDS_GEOMS_MAX 512          // 72 KB of LDS; larger scenes use the wave-uniform loops
struct GeomStage { const float* xf; bool on; };
// the same record by reference: the exact tests read inv / fwd from LDS where they use them (fwd only on a hit), which keeps
// 12-24 registers free compared with the by-value copy
struct GeomXfRef { const float* inv; const float* fwd; int diag; };
PTK_DI GeomXfRef ref_xf(const float* xf, int i, int diag) {
    GeomXfRef g; g.inv = xf + (size_t)i * PTK_XF_FLOATS; g.fwd = g.inv + 12; g.diag = diag;
    return g;
}
PTK_DI GeomXf load_xf(const float* xf, int i, int diag) {
    GeomXf g;
    const float* p = xf + (size_t)i * PTK_XF_FLOATS;
#pragma unroll
    for (int k = 0; k < 12; ++k) { g.inv[k] = p[k]; g.fwd[k] = p[12 + k]; }
    g.diag = diag;
    return g;
}
// boxIntersectionTest, intersections.h:48-90.  Returns world-space distance or -1.
// The normal is NOT computed here: `ns` receives the object-space vector n with N = normalize(invTranspose * n), so the
// caller evaluates it once for the primitive that wins (the reference evaluates it for every candidate and discards it).
// The exact tests in two forms.  FAST (device only): every divide / sqrt / 1/sqrt through the *_steps sequences with NO branch, the
// range conditions of all of them ANDed into `ok` (a handful of compares; a guarded helper per operation costs an exec-mask
// save / restore and a branch each — measured 4-6 % of the headline).  Conditions that follow from earlier ones are not tested again:
// a vector that was just normalised has |v|^2 within a few ulps of 1, so the second normalize of getPointOnRay
// (intersections.h:27-29) is always in range when the first was.  The caller recomputes the whole test in the plain form when
// `ok` comes back false — zero / denormal / huge / NaN operands, a slab numerator that is exactly 0, a grazing sphere radicand.
template <bool FAST> PTK_DI float t_inv_sqrt(float x, bool& ok, float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (FAST) { ok = ok && ex_in(x, lo, hi); const float s = sqrt_steps(x); return inv_steps(s, rcp_refined(s)); }
#endif
    return 1.0f / sqrtf(x);
}
template <bool FAST> PTK_DI float t_inv_sqrt_unit(float x) {       // x = |v|^2 of a vector normalised by a FAST step: in range by construction
#if defined(__HIP_DEVICE_COMPILE__)
    if (FAST) { const float s = sqrt_steps(x); return inv_steps(s, rcp_refined(s)); }
#endif
    return 1.0f / sqrtf(x);
}
template <bool FAST> PTK_DI float t_sqrt(float x, bool& ok, float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (FAST) { ok = ok && ex_in(x, lo, hi); return sqrt_steps(x); }
#endif
    return sqrtf(x);
}
template <bool FAST> PTK_DI void t_box_slabs(f3 o, f3 d, f3& t1, f3& t2, bool& ok) {
    const f3 n1 = mk3(-0.5f, -0.5f, -0.5f) - o, n2 = mk3(0.5f, 0.5f, 0.5f) - o;
#if defined(__HIP_DEVICE_COMPILE__)
    if (FAST) {
        // numerators: |o| <= 2^39 bounds them above; below, -0.5 - o and 0.5 - o are exact next to 0 (Sterbenz), so a numerator is 0 —
        // when |o| == 0.5 — or at least 2^-25.  Denominators: components of a vector normalised by a FAST step (<= 1 + a few ulps): only
        // the lower bound needs a test.  Every compare is false for a NaN.
        const float ox = __builtin_fabsf(o.x), oy = __builtin_fabsf(o.y), oz = __builtin_fabsf(o.z);
        ok = ok && (ox <= 0.5f * PTK_EX_HI && ox != 0.5f) && (oy <= 0.5f * PTK_EX_HI && oy != 0.5f) && (oz <= 0.5f * PTK_EX_HI && oz != 0.5f) &&
             __builtin_fabsf(d.x) >= PTK_EX_LO && __builtin_fabsf(d.y) >= PTK_EX_LO && __builtin_fabsf(d.z) >= PTK_EX_LO;
        const ptk_v2f dxy = xy_of(d), yxy = rcp_refined2(dxy);
        const float yz = rcp_refined(d.z);
        const ptk_v2f a = div_steps2(xy_of(n1), dxy, yxy), b = div_steps2(xy_of(n2), dxy, yxy);
        const ptk_v2f c = div_steps2(mk2(n1.z, n2.z), splat2(d.z), splat2(yz));
        t1 = mk3(a.x, a.y, c.x); t2 = mk3(b.x, b.y, c.y);
        return;
    }
#endif
    t1 = mk3(n1.x / d.x, n1.y / d.y, n1.z / d.z); t2 = mk3(n2.x / d.x, n2.y / d.y, n2.z / d.z);
}

template <bool FAST, class G>
PTK_DI float box_test_impl(const G& g, f3 ro, f3 rd, f3& P, f3& ns, bool& ok) {
    f3 qo = xform_point_d(g.inv, ro, g.diag);
    const f3 qv = xform_dir_d(g.inv, rd, g.diag);
    f3 qd = qv * t_inv_sqrt<FAST>(dot(qv, qv), ok, PTK_EX_LO2, PTK_EX_HI2);
    float tmin = -1e38f, tmax = 1e38f;
    int amin = 0, amax = 0; float smin = 0.f, smax_ = 0.f;     // normal = axis * sign (zero-initialised tmin_n/tmax_n)
    f3 s1, s2;
    t_box_slabs<FAST>(qo, qd, s1, s2, ok);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float t1 = comp(s1, a);
        float t2 = comp(s2, a);
        float ta = gmin(t1, t2), tb = gmax(t1, t2);
        float sg = t2 < t1 ? +1.f : -1.f;
        if (ta > 0 && ta > tmin) { tmin = ta; amin = a; smin = sg; }
        if (tb < tmax) { tmax = tb; amax = a; smax_ = sg; }
    }
    if (tmax >= tmin && tmax > 0) {
        if (tmin <= 0) { tmin = tmax; amin = amax; smin = smax_; }
        ns = mk3(amin == 0 ? smin : 0.f, amin == 1 ? smin : 0.f, amin == 2 ? smin : 0.f);
        f3 p = qo + (tmin - .0001f) * (qd * t_inv_sqrt_unit<FAST>(dot(qd, qd)));      // getPointOnRay, intersections.h:27-29: normalize(qd) again
        P = xform_point_d(g.fwd, p, g.diag);
        const f3 dv = ro - P;
        return t_sqrt<FAST>(dot(dv, dv), ok, PTK_EX_LO2, PTK_EX_HI2);
    }
    return -1.f;
}
// boxIntersectionTest, intersections.h:48-90.  Returns world-space distance or -1.
// The normal is NOT computed here: `ns` receives the object-space vector n with N = normalize(invTranspose * n), so the
// caller evaluates it once for the primitive that wins (the reference evaluates it for every candidate and discards it).
template <class G>
PTK_DI float box_test(const G& g, f3 ro, f3 rd, f3& P, f3& ns) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PTK_EX_PLAIN)
    bool ok = true;
    f3 Pf = mk3(0, 0, 0), nf = mk3(0, 0, 0);
    const float t = box_test_impl<true>(g, ro, rd, Pf, nf, ok);
#ifndef PTK_EX_NOFALLBACK
    if (__builtin_expect(!ok, 0)) { bool dummy = true; return box_test_impl<false>(g, ro, rd, P, ns, dummy); }
#endif
    if (t >= 0.f) { P = Pf; ns = nf; }
    return t;
#else
    bool dummy = true;
    return box_test_impl<false>(g, ro, rd, P, ns, dummy);
#endif
}

// sphereIntersectionTest, intersections.h:102-144 (normal deferred like box_test: ns = object-space hit point)
template <bool FAST, class G>
PTK_DI float sphere_test_impl(const G& g, f3 ro, f3 rd, f3& P, f3& ns, bool& ok) {
    f3 o = xform_point_d(g.inv, ro, g.diag);
    const f3 dv0 = xform_dir_d(g.inv, rd, g.diag);
    f3 d = dv0 * t_inv_sqrt<FAST>(dot(dv0, dv0), ok, PTK_EX_LO2, PTK_EX_HI2);
    float vd = dot(o, d);
    float radicand = vd * vd - (dot(o, o) - 0.25f);
    if (radicand < 0) return -1.f;
    float sq = t_sqrt<FAST>(radicand, ok, PTK_EX_LO2, PTK_EX_HI2);
    float first = -vd;
    float t1 = first + sq, t2 = first - sq;
    float t;
    if (t1 < 0 && t2 < 0) return -1.f;
    else if (t1 > 0 && t2 > 0) t = (t2 < t1) ? t2 : t1;
    else t = (t1 < t2) ? t2 : t1;
    f3 obj = o + (t - .0001f) * (d * t_inv_sqrt_unit<FAST>(dot(d, d)));
    P = xform_point_d(g.fwd, obj, g.diag);
    ns = obj;
    const f3 dv = ro - P;
    return t_sqrt<FAST>(dot(dv, dv), ok, PTK_EX_LO2, PTK_EX_HI2);
}
template <class G>
PTK_DI float sphere_test(const G& g, f3 ro, f3 rd, f3& P, f3& ns) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PTK_EX_PLAIN)
    bool ok = true;
    f3 Pf = mk3(0, 0, 0), nf = mk3(0, 0, 0);
    const float t = sphere_test_impl<true>(g, ro, rd, Pf, nf, ok);
#ifndef PTK_EX_NOFALLBACK
    if (__builtin_expect(!ok, 0)) { bool dummy = true; return sphere_test_impl<false>(g, ro, rd, P, ns, dummy); }
#endif
    if (t >= 0.f) { P = Pf; ns = nf; }
    return t;
#else
    bool dummy = true;
    return sphere_test_impl<false>(g, ro, rd, P, ns, dummy);
#endif
}
// the deferred normal of an analytic primitive: normalize(invTranspose * ns)  (intersections.h:87, 138)
PTK_DI f3 analytic_normal(const DevGeom& g, f3 ns) { return normalize(xform_nrm_d(g.invT, ns, g.diag)); }

// Bounds3::IntersectP(ray, t), Bounds3.hpp:162-189
PTK_DI bool aabb_test(const DevNode& nd, f3 o, f3 d, f3 inv, float& t) {
    float tEnter = -FLT_MAX, tExit = FLT_MAX;
    {
        if (d.x == 0) { if (o.x < nd.minx || o.x > nd.maxx) return false; }
        else { float tMax = (nd.maxx - o.x) * inv.x, tMin = (nd.minx - o.x) * inv.x;
               if (d.x < 0) { float s = tMax; tMax = tMin; tMin = s; }
               tEnter = tMin > tEnter ? tMin : tEnter; tExit = tMax < tExit ? tMax : tExit; }
    }
    {
        if (d.y == 0) { if (o.y < nd.miny || o.y > nd.maxy) return false; }
        else { float tMax = (nd.maxy - o.y) * inv.y, tMin = (nd.miny - o.y) * inv.y;
               if (d.y < 0) { float s = tMax; tMax = tMin; tMin = s; }
               tEnter = tMin > tEnter ? tMin : tEnter; tExit = tMax < tExit ? tMax : tExit; }
    }
    {
        if (d.z == 0) { if (o.z < nd.minz || o.z > nd.maxz) return false; }
        else { float tMax = (nd.maxz - o.z) * inv.z, tMin = (nd.minz - o.z) * inv.z;
               if (d.z < 0) { float s = tMax; tMax = tMin; tMin = s; }
               tEnter = tMin > tEnter ? tMin : tEnter; tExit = tMax < tExit ? tMax : tExit; }
    }
    t = tEnter;
    return tEnter <= tExit && tExit > 0;
}

// Same test for a ray with no zero direction component (the common case), ~23 VALU instead of ~50.  With d_k != 0 the
// reference's "swap if d_k < 0" equals min/max of the two plane distances (boxes of a non-empty BVH have min <= max), and no
// NaN can arise ((plane - o) is finite or +-inf, inv is finite and non-zero), so compare-and-select chains equal min/max.
// The -FLT_MAX / FLT_MAX seeds are kept so that every comparison the caller makes sees the same values.
PTK_DI bool aabb_test_fast(const DevNode& nd, f3 o, f3 inv, float& t) {
    const float ax = (nd.minx - o.x) * inv.x, bx = (nd.maxx - o.x) * inv.x;
    const float ay = (nd.miny - o.y) * inv.y, by = (nd.maxy - o.y) * inv.y;
    const float az = (nd.minz - o.z) * inv.z, bz = (nd.maxz - o.z) * inv.z;
    const float tEnter = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), -FLT_MAX));
    const float tExit = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), FLT_MAX));
    t = tEnter;
    return tEnter <= tExit && tExit > 0;
}

// Triangle::getInterSect, Bounds3.hpp:298-326 (e1, e2 are the same fp32 subtractions, done once on the host)
PTK_DI bool tri_test(const DevTriE& T, f3 o, f3 d, float& t, float& u, float& v) {
    f3 v0 = mk3(T.v0[0], T.v0[1], T.v0[2]);
    f3 e1 = mk3(T.e1[0], T.e1[1], T.e1[2]);
    f3 e2 = mk3(T.e2[0], T.e2[1], T.e2[2]);
    f3 pvec = cross(d, e2);
    float det = dot(e1, pvec);
    if (det == 0) return false;
    float invDet = 1 / det;
    f3 tvec = o - v0;
    u = dot(tvec, pvec); u *= invDet;
    f3 qvec = cross(tvec, e1);
    v = dot(d, qvec); v *= invDet;
    t = dot(e2, qvec); t *= invDet;
    if (t < 0 || u < 0 || v < 0 || (1 - u - v) < 0) return false;
    return true;
}

PTK_DI int mtbvh_dir(f3 d) {          // pathtrace.cu:242-244
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    int k = ((ax > ay) && (ax > az)) ? 0 : (ay > az ? 1 : 2);
    return k + (comp(d, k) > 0 ? 0 : 3);
}

struct Hit {
    float t; f3 n; f3 p; f2 uv; int mat, geom, tri;
};

// ---- textures (image.h:53-110) and the environment map (utilities.h:209-214, pathtrace.cu:303-311) -------------------
PTK_DI f3 texel(const SceneView& S, const DevTex& t, int x, int y) {
    x = x < 0 ? 0 : (x >= t.w ? t.w - 1 : x);                 // the reference indexes unchecked; clamp instead of reading out of bounds
    y = y < 0 ? 0 : (y >= t.h ? t.h - 1 : y);
    const float* p = S.texels + 3 * ((size_t)t.offset + (size_t)y * t.w + x);
    return mk3(p[0], p[1], p[2]);
}
// devTexObj::linearSample, image.h:73-85
PTK_DI f3 tex_sample(const SceneView& S, int id, f2 uv) {
    const DevTex t = S.tex[id];
    const float x = uv.x * (t.w - 1), y = uv.y * (t.h - 1);
    const int lx = (int)x, ux = (x + 1 >= t.w) ? lx : lx + 1;
    const int ly = (int)y, uy = (y + 1 >= t.h) ? ly : ly + 1;
    const float fx = x - floorf(x), fy = y - floorf(y);        // glm::fract
    const f3 p1 = mixv(texel(S, t, lx, ly), texel(S, t, ux, ly), fx);
    const f3 p2 = mixv(texel(S, t, lx, uy), texel(S, t, ux, uy), fx);
    return mixv(p1, p2, fy);
}
// ptk atan2 specification (shared text with oracle/pt_oracle.cpp; cephes atanf/atan2f structure, IEEE + - * / only):
//   atan(x): s = x < 0, x = |x|;  x > tan(3pi/8): y = pi/2, x = -(1/x);  x > tan(pi/8): y = pi/4, x = (x-1)/(x+1);  else y = 0
//            z = x*x;  y += (((8.05374449538e-2 z - 1.38776856032e-1) z + 1.99777106478e-1) z - 3.33329491539e-1) z x + x;  s: y = -y
//   atan2(y,x): x == 0: y < 0 -> -pi/2, y == 0 -> 0, else pi/2;  y == 0: x < 0 -> pi, else 0;
//               w = 0 (x > 0), pi (x < 0, y >= 0), -pi (x < 0, y < 0);  result = w + atan(y/x)
PTK_DI float atan_ptk(float x) {
    const bool neg = x < 0; x = fabsf(x);
    float y;
    if (x > 2.414213562373095f) { y = 1.5707963267948966192f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483096f; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.f;
    const float z = x * x;
    y += (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
    return neg ? -y : y;
}
PTK_DI float atan2_ptk(float y, float x) {
    const float PIF = 3.141592653589793238f, PIO2F = 1.5707963267948966192f;
    if (x == 0) { if (y < 0) return -PIO2F; if (y == 0) return 0.f; return PIO2F; }
    if (y == 0) return x < 0 ? PIF : 0.f;
    const float w = x > 0 ? 0.f : (y < 0 ? -PIF : PIF);
    return w + atan_ptk(y / x);
}
// math::sphere2Plane, utilities.h:209-214 ("* InvPI" expands to "* 1.f/PI")
PTK_DI f2 sphere2plane(f3 d) {
    f2 uv;
    const float a = atan2_ptk(d.z, d.x) * 1.f / PTK_PI * .5f + 1.f;
    uv.x = a - floorf(a);
    uv.y = gmax(atan2_ptk(d.y, sqrtf(d.x * d.x + d.z * d.z)) * 1.f / PTK_PI + 0.5f, 0.f);
    return uv;
}
PTK_DI f3 env_radiance(const SceneView& S, f3 dir) { return tex_sample(S, S.env_tex, sphere2plane(dir)); }

// ---- closest hit, in three pieces so that kernels can schedule them differently -----------------------------------
// (1) analytic primitives (pathtrace.cu:211-235): wave-uniform loop, primitive records in SGPRs
struct Analytic { float t_min; int geom; f3 P, NS; };

// Conservative candidate test.  The reference runs the exact object-space test of EVERY primitive for every ray
// (pathtrace.cu:211-235); a test that misses has no side effect, so primitives that provably miss can be skipped.
// A primitive is a candidate unless the ray misses its padded world-space box: slab test t = (b - o') * (1/d) with
// approximate reciprocals (packed subtract + packed multiply, two planes per instruction), the box grown on the host
// (DevGeomBox) and the origin here by PTK_CAND_REL * (1 + |origin|_inf).  The exact test's rounding moves a hit/miss
// decision by ~1e-7 relative to those magnitudes (measured, see PTK_CAND_REL).  Direction components that are 0 or
// denormal give +-inf slabs of the right sign (the subtraction comes first, so there is no inf - inf); the only NaN is
// 0 * inf — shifted origin exactly on a padded plane, direction parallel to it — which the IEEE min/max turn into a
// miss, correct because the real primitive lies a margin inside.  Rays with NaN components never produce an accepted
// hit in the exact test either (t > 0 fails).  Output: one bit per primitive of the chunk [base, base+n), n <= 32.
struct Cand { unsigned cube, sphere, diag; };
PTK_DI Cand analytic_candidates(const SceneView& S, int base, int n, f3 ro, f3 rd) {
    const float pr = PTK_CAND_REL * (1.0f + fmaxf(fabsf(ro.x), fmaxf(fabsf(ro.y), fabsf(ro.z))));
#if defined(__HIP_DEVICE_COMPILE__)
    const f3 inv = mk3(__builtin_amdgcn_rcpf(rd.x), __builtin_amdgcn_rcpf(rd.y), __builtin_amdgcn_rcpf(rd.z));
#else
    const f3 inv = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
#endif
    // plane k of axis a: t = (s[2a+k] - o'.a[k]) * inv.a   (lo planes see the origin moved up by pr, hi planes moved down)
    const ptk_v2f ix = {inv.x, inv.x}, iy = {inv.y, inv.y}, iz = {inv.z, inv.z};
    const ptk_v2f ox = {ro.x + pr, ro.x - pr}, oy = {ro.y + pr, ro.y - pr}, oz = {ro.z + pr, ro.z - pr};
    const float oo = ro.x * ro.x + ro.y * ro.y + ro.z * ro.z;
    const float dd = rd.x * rd.x + rd.y * rd.y + rd.z * rd.z;
#if defined(__HIP_DEVICE_COMPILE__)
    const float idd = __builtin_amdgcn_rcpf(dd);
#else
    const float idd = 1.0f / dd;
#endif
    const DevGeomMask m = load_gmask_uniform(S.gmask, base >> 5);
    const unsigned mround = m.round;
    unsigned hit = 0u;
    for (int k = 0; k < n; ++k) {
        const DevGeomBox b = load_gbox_uniform(S.gbox, base + k);
        if ((mround >> k) & 1u) {                        // wave-uniform: world-space sphere
            const f3 oc = mk3(b.s[0] - ro.x, b.s[1] - ro.y, b.s[2] - ro.z);
            const float oc2 = oc.x * oc.x + oc.y * oc.y + oc.z * oc.z;
            const float tca = oc.x * rd.x + oc.y * rd.y + oc.z * rd.z;
            const float lim = (b.s[3] + pr * (b.s[4] + pr)) + PTK_CAND_QUAD2 * oc2;
            const float b2 = oc2 - tca * tca * idd;
            const bool cull = (b2 > lim) || (tca < 0.0f && oc2 > lim);
            hit |= cull ? 0u : (1u << k);
            continue;
        }
        ptk_v2f bx = {b.s[0], b.s[1]}, by = {b.s[2], b.s[3]}, bz = {b.s[4], b.s[5]};
        {   // sphere pad qa + qb |o|^2; qa = qb = 0 for cubes (dev_scene.hip), so this is unconditional: the compiler turned the
            // wave-uniform `if (sphere)` into 12 selects per primitive (PMC: the candidate test was 27 % of k_intersect's instructions)
            const float pg = b.qa + b.qb * oo;
            const ptk_v2f pp = {-pg, pg};
            bx += pp; by += pp; bz += pp;
        }
        const ptk_v2f tx = (bx - ox) * ix, ty = (by - oy) * iy, tz = (bz - oz) * iz;
        const float tn = fmaxf(fmaxf(fminf(tx.x, tx.y), fminf(ty.x, ty.y)), fminf(tz.x, tz.y));
        const float tf = fminf(fminf(fmaxf(tx.x, tx.y), fmaxf(ty.x, ty.y)), fmaxf(tz.x, tz.y));
        hit |= (!(tn > tf) && !(tf <= 0.0f)) ? (1u << k) : 0u;
    }
    Cand c; c.cube = hit & m.cube; c.sphere = hit & m.sphere; c.diag = m.diag;
    return c;
}

// Closest analytic primitive.  The reference keeps the FIRST primitive in index order among equal distances (strict
// t_min > t, pathtrace.cu:228).  Staged form: each lane walks its own candidate list, cubes then spheres per chunk of 32
// primitives (two type-pure loops, max-over-lanes(candidates) trips per wave instead of n_geoms), and the same winner is
// selected by the order-independent rule "smaller t, or equal t and smaller index".  Unstaged form: the wave-uniform
// loop over every primitive with the record in SGPRs.
PTK_DI void analytic_closest(const SceneView& S, const GeomStage& gs, f3 ro, f3 rd, Analytic& a) {
    a.t_min = FLT_MAX; a.geom = -1; a.P = mk3(0, 0, 0); a.NS = mk3(0, 0, 0);
    if (!gs.on) {
        float t = -1.f;
        for (int i = 0; i < S.n_geoms; ++i) {
            const DevGeom g = load_geom_uniform(S.geoms, i);
            f3 tp, tn;
            if (g.type == 1) t = box_test(g, ro, rd, tp, tn);
            else if (g.type == 0) t = sphere_test(g, ro, rd, tp, tn);
            if (t > 0.0f && a.t_min > t) { a.t_min = t; a.geom = i; a.P = tp; a.NS = tn; }
        }
        return;
    }
    for (int base = 0; base < S.n_geoms; base += 32) {
        const int n = (S.n_geoms - base) < 32 ? (S.n_geoms - base) : 32;
        const Cand c = analytic_candidates(S, base, n, ro, rd);
        unsigned mc = c.cube, ms = c.sphere;
        while (mc) {
            const int k = __builtin_ctz(mc), i = base + k;
            mc &= mc - 1u;
            const GeomXfRef g = ref_xf(gs.xf, i, (c.diag >> k) & 1);
            f3 tp, tn;
            const float t = box_test(g, ro, rd, tp, tn);
            if (t > 0.0f && (a.t_min > t || (a.t_min == t && i < a.geom))) { a.t_min = t; a.geom = i; a.P = tp; a.NS = tn; }
        }
        while (ms) {
            const int k = __builtin_ctz(ms), i = base + k;
            ms &= ms - 1u;
            const GeomXfRef g = ref_xf(gs.xf, i, (c.diag >> k) & 1);
            f3 tp, tn;
            const float t = sphere_test(g, ro, rd, tp, tn);
            if (t > 0.0f && (a.t_min > t || (a.t_min == t && i < a.geom))) { a.t_min = t; a.geom = i; a.P = tp; a.NS = tn; }
        }
    }
}
// hit record of an analytic winner (pathtrace.cu:315-340 with the constant normal sampler)
PTK_DI void analytic_record(const SceneView& S, const Analytic& a, Hit& h) {
    h.t = a.t_min; h.p = a.P;
    h.n = normalize(analytic_normal(S.geoms[a.geom], a.NS));     // per-lane geom index: one gather, once per ray
    h.uv.x = 0.f; h.uv.y = 0.f;
    h.mat = S.geoms[a.geom].materialid; h.geom = a.geom; h.tri = -1;
}
// the same from the LDS-staged records (no dependent global gather at the end of the kernel)
PTK_DI void analytic_record_staged(const float* xf, const Analytic& a, Hit& h) {
    const float* r = xf + (size_t)a.geom * PTK_XF_FLOATS;
    h.t = a.t_min; h.p = a.P;
    h.n = normalize(normalize(xform_nrm_d(r + 24, a.NS, __float_as_int(r[33]))));
    h.uv.x = 0.f; h.uv.y = 0.f;
    h.mat = __float_as_int(r[34]); h.geom = a.geom; h.tri = -1;
}
// (2) one traversal state per ray: threaded (stackless) walk over the direction-ordered nodes (pathtrace.cu:236-279)
// ONE COPY INSTEAD OF SIX (round 4).  The reference threads its tree six times so that a ray visits the nearer child first (BVH.cpp:180-239); the
// order has two effects: how soon t_min shrinks (speed only) and WHICH triangle wins among exactly equal distances — the first one visited,
// because the update is strict (pathtrace.cu:262-267).  No other effect: a leaf is skipped only when it cannot lower t_min, so the winner is
// argmin (t, position in the direction's leaf order) over the leaves whose own tests pass.  With the position of every triangle in each of the
// six leaf orders in a table (SceneView::ranks), a walk in ANY order returns the reference's winner: accept on t_min > t, or on t_min == t when
// the new triangle comes earlier in the ray's direction order.  The large-mesh default therefore uploads the direction-0 copy only — it serves
// the walks that run beside the 8-wide tree (k_tail, the fallback rays of k_traverse_wide, the threaded A/B walk) and every shadow ray (an
// any-hit: order-free anyway); small meshes walked inside the fused kernel keep six copies (the nearer-first order is their speed).
struct Walk { f3 o, d, inv; float t_min; int tri; float u, v; int idx; const DevNode* nodes; bool general; const uint32_t* rank; };
PTK_DI void walk_begin(const SceneView& S, f3 ro, f3 rd, float t_min, Walk& w) {
    w.o = ro; w.d = rd; w.inv = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
    w.t_min = t_min; w.tri = -1; w.u = 0.f; w.v = 0.f; w.idx = 0;
    const int dir = mtbvh_dir(rd);
    w.nodes = S.nodes + (size_t)(S.n_copies == 6 ? dir : 0) * S.bvh_size;
    w.rank = S.n_copies == 6 ? nullptr : S.ranks + (size_t)dir * S.n_tris;
    w.general = (rd.x == 0) || (rd.y == 0) || (rd.z == 0);       // needs the containment tests of Bounds3.hpp:174-177
}
// one node visit; returns the number of triangle tests it did (0/1).  When w.idx becomes -1 the walk is complete.
PTK_DI int walk_apply(const SceneView& S, Walk& w, const DevNode& nd);
PTK_DI int walk_step(const SceneView& S, Walk& w) {
    const DevNode nd = w.nodes[w.idx];
    return walk_apply(S, w, nd);
}
PTK_DI int walk_apply(const SceneView& S, Walk& w, const DevNode& nd) {
    float tEnter;
    int tests = 0;
    const bool in = w.general ? aabb_test(nd, w.o, w.d, w.inv, tEnter) : aabb_test_fast(nd, w.o, w.inv, tEnter);
    if (!in || tEnter > w.t_min) { w.idx = nd.b; return 0; }
    if (nd.a >= 0) { w.idx = nd.a; return 0; }
    if (nd.a <= -2) {
        const int k = -(nd.a + 2);
        float tt, u, v;
        tests = 1;
        if (tri_test(S.trie[k], w.o, w.d, tt, u, v)) {
            bool take = w.t_min > tt;
            if (!take && w.rank && w.t_min == tt && w.tri >= 0) take = w.rank[k] < w.rank[w.tri];      // exact tie, walked out of the direction's order
            if (take) { w.t_min = tt; w.tri = k; w.u = u; w.v = v; }
        }
    }
    w.idx = nd.b;
    return tests;
}
// (3) hit record of a triangle winner (pathtrace.cu:268-274, 315-340)
PTK_DI void triangle_record(const SceneView& S, int tri, float bu, float bv, float t, Hit& h) {
    const DevTriV& V = S.triv[tri];
    const DevTriA& A = S.tria[tri];
    const float w = 1 - bu - bv;
    f3 P = w * mk3(V.v0[0], V.v0[1], V.v0[2]) + bu * mk3(V.v1[0], V.v1[1], V.v1[2]) + bv * mk3(V.v2[0], V.v2[1], V.v2[2]);
    f3 N = w * mk3(A.n0[0], A.n0[1], A.n0[2]) + bu * mk3(A.n1[0], A.n1[1], A.n1[2]) + bv * mk3(A.n2[0], A.n2[1], A.n2[2]);
    f2 tc; tc.x = w * A.t0[0] + bu * A.t1[0] + bv * A.t2[0]; tc.y = w * A.t0[1] + bu * A.t1[1] + bv * A.t2[1];
    h.uv.x = gclamp(tc.x, 0.f, 1.f);
    h.uv.y = gclamp(tc.y, 0.f, 1.f);
    h.t = t; h.p = P; h.n = normalize(N);
    h.geom = A.geomIdx; h.mat = S.geoms[A.geomIdx].materialid; h.tri = tri;
    // normal mapping, pathtrace.cu:323-340 (sampled with the UNclamped texCoords, like the reference).  With the constant
    // sampler (0.5,0.5,1) localNorm is exactly (0,0,1) and the branch is never taken, so only real maps are evaluated.
    const int nt = S.mats[h.mat].normal_tex;
    if (nt >= 0) {
        const f3 mapped = tex_sample(S, nt, tc);
        const f3 ln = normalize(mk3(mapped.x * 1.f - 0.5f, mapped.y * 1.f - 0.5f, mapped.z * 1.f - 0.5f));
        f3 T = mk3(A.tangent[0], A.tangent[1], A.tangent[2]);
        if (dot(T, T) > 1e-3f && (double)fabsf(ln.z - 1.f) > 1e-5) {
            const f3 nrm = h.n;
            const f3 B = normalize(cross(nrm, T));
            T = normalize(cross(B, nrm));
            h.n = normalize(mk3(T.x * ln.x + B.x * ln.y + nrm.x * ln.z, T.y * ln.x + B.y * ln.y + nrm.y * ln.z, T.z * ln.x + B.z * ln.y + nrm.z * ln.z));
        }
    }
}

// computeIntersections body (pathtrace.cu:191-343) for one ray, everything in the calling lane.  Returns rayValid.
// second half of computeIntersections for one ray: BVH walk seeded with the analytic winner, then the hit record
template <bool STATS>
PTK_DI int closest_hit_finish(const SceneView& S, const Analytic& a, f3 ro, f3 rd, Hit& h, unsigned& nodeVisits, unsigned& triTests) {
    int tri = -1; float bu = 0.f, bv = 0.f, t_min = a.t_min;
    if (S.n_tris > 0) {                 // a triangle-free scene has one empty root per direction: it can never hit
        Walk w;
        walk_begin(S, ro, rd, a.t_min, w);
        while (w.idx != -1) {
            const int k = walk_step(S, w);
            if (STATS) { nodeVisits++; triTests += k; }
        }
        tri = w.tri; bu = w.u; bv = w.v; t_min = w.t_min;
    } else if (STATS) nodeVisits++;     // the reference still visits that empty root once
    if (tri >= 0) { triangle_record(S, tri, bu, bv, t_min, h); return 1; }
    if (a.geom == -1) { h.t = -1.0f; return 0; }
    analytic_record(S, a, h);
    return 1;
}
template <bool STATS>
PTK_DI int closest_hit(const SceneView& S, const GeomStage& gs, f3 ro, f3 rd, Hit& h, unsigned& nodeVisits, unsigned& triTests) {
    Analytic a;
    analytic_closest(S, gs, ro, rd, a);
    return closest_hit_finish<STATS>(S, a, ro, rd, h, nodeVisits, triTests);
}

// LightSampler::occulusionTest, lightSample.h:59-139 — in two parts (analytic primitives, lightSample.h:61-88; the BVH walk, :90-125) so
// that the fused kernel's deferred shadow phase can run the first part block-cooperatively (analytic_anyhit_lds, ptk_kernels.hip) and the
// second per lane.  minT = length(des - ori), computed by the caller exactly as the reference does at the top of the function.
PTK_DI bool occluded_analytic(const SceneView& S, const GeomStage& gs, f3 ori, f3 dir, float minT) {
    if (!gs.on) {
        float t = 0.f;
        for (int i = 0; i < S.n_geoms; ++i) {
            const DevGeom g = load_geom_uniform(S.geoms, i);
            f3 tp, tn;
            if (g.type == 1) t = box_test(g, ori, dir, tp, tn);
            else if (g.type == 0) t = sphere_test(g, ori, dir, tp, tn);
            if (t > 0.0f && minT - 1e-5f > t && (double)fabsf(t - minT) > 1e-2) return true;
        }
    } else {
        // any-hit over the analytic primitives is order-independent, so only the candidates are tested
        for (int base = 0; base < S.n_geoms; base += 32) {
            const int n = (S.n_geoms - base) < 32 ? (S.n_geoms - base) : 32;
            const Cand c = analytic_candidates(S, base, n, ori, dir);
            unsigned mc = c.cube, ms = c.sphere;
            while (mc) {
                const int k = __builtin_ctz(mc);
                mc &= mc - 1u;
                const GeomXfRef g = ref_xf(gs.xf, base + k, (c.diag >> k) & 1);
                f3 tp, tn;
                const float t = box_test(g, ori, dir, tp, tn);
                if (t > 0.0f && minT - 1e-5f > t && (double)fabsf(t - minT) > 1e-2) return true;
            }
            while (ms) {
                const int k = __builtin_ctz(ms);
                ms &= ms - 1u;
                const GeomXfRef g = ref_xf(gs.xf, base + k, (c.diag >> k) & 1);
                f3 tp, tn;
                const float t = sphere_test(g, ori, dir, tp, tn);
                if (t > 0.0f && minT - 1e-5f > t && (double)fabsf(t - minT) > 1e-2) return true;
            }
        }
    }
    return false;
}
PTK_DI bool occluded_bvh(const SceneView& S, f3 ori, f3 dir, float minT) {
    if (S.n_tris == 0) return false;
    const DevNode* nodes = S.nodes;                    // an any-hit is order-free: every shadow ray walks the direction-0 copy (one 640-KB table in L2 instead of six)
    f3 inv = mk3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    const bool general = (dir.x == 0) || (dir.y == 0) || (dir.z == 0);
    int idx = 0;
    while (idx != -1) {
        const DevNode nd = nodes[idx];
        float tEnter;
        const bool in = general ? aabb_test(nd, ori, dir, inv, tEnter) : aabb_test_fast(nd, ori, inv, tEnter);
        if (!in || tEnter > minT) { idx = nd.b; continue; }
        if (nd.a >= 0) { idx = nd.a; continue; }
        if (nd.a <= -2) {
            float tt, u, v;
            if (tri_test(S.trie[-(nd.a + 2)], ori, dir, tt, u, v) && minT - 1e-5f > tt && (double)fabsf(tt - minT) > 1e-4) return true;
        }
        idx = nd.b;
    }
    return false;
}
PTK_DI bool occluded(const SceneView& S, const GeomStage& gs, f3 ori, f3 dir, f3 des) {
    const float minT = length(des - ori);
    return occluded_analytic(S, gs, ori, dir, minT) || occluded_bvh(S, ori, dir, minT);
}

// material parameters at a hit: the four devTexSampler::linearSample(uv) calls of material.h, resolved once
struct MatEval { int type; f3 albedo_member, albedo, albedo_pi; float rough, metal, ior; };
PTK_DI MatEval eval_material(const SceneView& S, const DevMaterial& m, f2 uv) {
    MatEval e;
    e.type = m.type; e.ior = m.ior;
    e.albedo_member = mk3(m.albedo[0], m.albedo[1], m.albedo[2]);
    if (m.albedo_tex >= 0) { e.albedo = tex_sample(S, m.albedo_tex, uv); e.albedo_pi = e.albedo / PTK_PI; }
    else { e.albedo = mk3(m.albedo_s[0], m.albedo_s[1], m.albedo_s[2]); e.albedo_pi = mk3(m.albedo_pi[0], m.albedo_pi[1], m.albedo_pi[2]); }
    e.rough = m.rough_tex >= 0 ? gclamp(tex_sample(S, m.rough_tex, uv).x, PTK_ROUGHNESS_MIN, PTK_ROUGHNESS_MAX) : m.rough;
    e.metal = m.metal_tex >= 0 ? gclamp(tex_sample(S, m.metal_tex, uv).x, 0.f, 1.f) : m.metal;
    return e;
}

// ---- shading math ---------------------------------------------------------------------------------
PTK_DI void pixar_basis(f3 n, f3& xp, f3& yp) {                   // utilities.h:188-200
    float sz = n.z >= 0 ? 1.f : -1.f;
    float a = inv_exact(sz + n.z);
    float sx = sz * n.x;
    float b = n.x * n.y * a;
    xp = mk3(sx * n.x * a - 1.f, sz * b, sx);
    yp = mk3(b, n.y * n.y * a - sz, n.y);
}
PTK_DI f3 sample_hemisphere_cosine(f3 n, f2 r) {                  // utilities.h:224-235 + ONB 136-146
    f3 u, v; f3 w = normalize(n);
    pixar_basis(n, u, v);
    float sinTheta = sqrt_exact(r.x), cosTheta = sqrt_exact(1 - r.x);
    float sp, cp; sincos_ptk(PTK_TWO_PI * r.y, sp, cp);
    float lx = sinTheta * cp, ly = sinTheta * sp;
    return lx * u + ly * v + cosTheta * w;
}
PTK_DI float pow5(float x) { float x2 = x * x; return x2 * x2 * x; }
PTK_DI float nan_to_zero(float x) { return (x != x || __builtin_isinf(x)) ? 0.f : x; }   // math::processNAN utilities.h:275-283
PTK_DI f3 nan_to_zero(f3 v) { return mk3(nan_to_zero(v.x), nan_to_zero(v.y), nan_to_zero(v.z)); }
PTK_DI f3 fresnel_schlick(f3 f0, float c) { return f0 + (mk3(1.f, 1.f, 1.f) - f0) * pow5(1 - c); }
PTK_DI float fresnel_maxwell(float c1, float ior1, float ior2) {   // utilities.h:332-342
    float s1 = sqrtf(gmax(1 - c1 * c1, 0.f));
    float s2 = s1 * ior1 / ior2;
    if (s2 > 1) return 1.f;
    float c2 = sqrtf(gmax(1 - s2 * s2, 0.f));
    float rpa = (ior1 * c2 - ior2 * c1) / (ior1 * c2 + ior2 * c1);
    float rpe = (ior1 * c1 - ior2 * c2) / (ior1 * c1 + ior2 * c2);
    return (rpa * rpa + rpe * rpe) / 2.f;
}
PTK_DI f3 sample_normal_ggx(f3 n, f3 wo, float alpha, f2 r) {      // utilities.h:351-377
    f3 t, b; pixar_basis(n, t, b);
    f3 wl = mk3(t.x * wo.x + t.y * wo.y + t.z * wo.z, b.x * wo.x + b.y * wo.y + b.z * wo.z, n.x * wo.x + n.y * wo.y + n.z * wo.z);
    f3 wh = normalize(mk3(alpha * wl.x, alpha * wl.y, 1.f * wl.z));
    if (wh.z < 0) wh = wh * -1.f;
    f3 T1 = (wh.z < 0.99999f) ? normalize(mk3(-wh.y, wh.x, 0.f)) : mk3(1.f, 0.f, 0.f);
    f3 T2 = cross(wh, T1);
    float radius = sqrtf(r.x);
    float st, ct; sincos_ptk(PTK_TWO_PI * r.y, st, ct);
    float px = radius * ct, py = radius * st;
    float h = sqrtf(1 - px * px);
    float lx = (1 + wh.z) / 2.f;
    py = (1 - lx) * h + lx * py;
    float pz = sqrtf(gmax(0.f, 1.f - (px * px + py * py)));
    f3 nh = px * T1 + py * T2 + pz * wh;
    f3 l = mk3(alpha * nh.x, alpha * nh.y, gmax(1e-6f, nh.z));
    return normalize(mk3(t.x * l.x + b.x * l.y + n.x * l.z, t.y * l.x + b.y * l.y + n.y * l.z, t.z * l.x + b.z * l.y + n.z * l.z));
}
PTK_DI float smith_g2(float a2, float NoV, float NoL) {
    float denom = NoL * sqrtf(NoV * NoV * (1.f - a2) + a2) + NoV * sqrtf(NoL * NoL * (1.f - a2) + a2);
    return (2.f * NoV * NoL) / denom;
}
PTK_DI float smith_g1(float a2, float NoV) {
    float denom = sqrtf(NoV * NoV * (1 - a2) + a2) + NoV;
    return (2 * NoV) / denom;
}
PTK_DI float ggx_d(float c, float a2) {
    if (c < 1e-6f) return 0.f;
    float denom = c * c * (a2 - 1.f) + 1.f;
    denom = denom * denom * PTK_PI;
    return a2 / denom;
}
PTK_DI float power_heuristic(float f, float g) { return (f * f) / (f * f + g * g); }

PTK_DI f3 microfacet_bsdf(f3 n, f3 wo, f3 wi, f3 albedo, float rough) {       // material.h:246-261
    float a2 = rough * rough;
    float cosO = dot(n, wo), cosI = dot(n, wi);
    if ((double)(cosO * cosI) < 1e-7) return mk3(0, 0, 0);
    f3 wm = normalize(wo + wi);
    float D = ggx_d(dot(wm, n), a2);
    float G2 = smith_g2(rough, cosO, cosI);
    f3 F = fresnel_schlick(albedo, dot(wo, wm));
    return F * D * G2 / gmax(4 * cosO * cosI, 1e-8f);
}
PTK_DI float microfacet_pdf(f3 n, f3 wo, f3 wi, float rough) {                // material.h:263-271
    float a2 = rough * rough;
    float cosO = dot(n, wo);
    f3 wm = normalize(wo + wi);
    float D = ggx_d(dot(wm, n), a2);
    float G1 = smith_g1(rough, cosO);
    return G1 * D / gmax(4 * dot(wo, n), 1e-8f);
}
PTK_DI f3 metallic_bsdf(f3 n, f3 wo, f3 wi, f3 albedo, float rough, float metal) {   // material.h:296-310
    float a2 = rough * rough;
    float cosO = dot(n, wo), cosI = dot(n, wi);
    if ((double)(cosO * cosI) < 1e-7) return mk3(0, 0, 0);
    f3 wm = normalize(wo + wi);
    float D = ggx_d(dot(wm, n), a2);
    float G2 = smith_g2(rough, cosO, cosI);
    f3 F = fresnel_schlick(mixv(mk3(0.08f, 0.08f, 0.08f), albedo, metal), dot(wo, wm));
    float spec = D * G2 / gmax(4 * cosO * cosI, 1e-8f);
    f3 diff = (1.f - metal) * albedo * 1.f / PTK_PI;
    return mixvv(diff, mk3(spec, spec, spec), F);
}
PTK_DI float metallic_pdf(f3 n, f3 wo, f3 wi, float rough, float metal) {     // material.h:312-320
    float a2 = rough * rough;
    float cosO = dot(n, wo);
    f3 wm = normalize(wo + wi);
    float D = ggx_d(dot(wm, n), a2);
    float G1 = smith_g1(rough, cosO);
    return mixf(dot(wi, n) * 1.f / PTK_PI, G1 * D / gmax(4 * dot(wo, n), 1e-8f), 1.f / (2.f - metal));
}

struct Scatter { f3 bsdf; float pdf; f3 dir; };

// Material::scatterSample, material.h:450-481.  wo = incident direction (into the surface).
PTK_DI void scatter_sample(const MatEval& m, f3 n, f3 wo, Scatter& s, Rng& rng) {
    s.pdf = 0.f; s.bsdf = mk3(0, 0, 0); s.dir = mk3(0, 0, 0);
    f3 albedo = m.albedo;
    switch (m.type) {
    case 0: {
        s.bsdf = m.albedo_pi;                                    // albedo / pi (material.h:58), divided once per material
        s.dir = sample_hemisphere_cosine(n, sample2D(rng));
        s.pdf = dot(s.dir, n) / PTK_PI;
        break; }
    case 1: {
        float r = sample1D(rng);
        if (r < 1.f / (2.f - m.metal)) {
            f3 wm = sample_normal_ggx(n, -1.f * wo, m.rough * m.rough, sample2D(rng));
            s.dir = reflect(wo, wm);
        } else {
            s.dir = sample_hemisphere_cosine(n, sample2D(rng));
        }
        if (dot(-1.f * wo, n) < 0 || dot(s.dir, n) < 0) { s.bsdf = mk3(0, 0, 0); s.pdf = 0; return; }
        s.bsdf = metallic_bsdf(n, -1.f * wo, s.dir, albedo, m.rough, m.metal);
        s.pdf = metallic_pdf(n, -1.f * wo, s.dir, m.rough, m.metal);
        break; }
    case 2: {
        float ior1, ior2;
        float won = dot(wo, n);
        if (won < 0) { ior1 = 1; ior2 = m.ior; } else { ior2 = 1; ior1 = m.ior; }
        float F = fresnel_maxwell(fabsf(won), ior1, ior2);
        f3 nn = won < 0 ? n : -n;                                  // getReflectDir / getRefractDir, utilities.h:290-320
        if (sample1D(rng) < F) {
            s.dir = wo - 2.f * (nn * dot(wo, nn));
            s.bsdf = albedo;
        } else {
            f3 perp = (wo - dot(wo, nn) * nn) * (ior1 / ior2);
            f3 para = (-1 * sqrtf(1 - (perp.x * perp.x + perp.y * perp.y + perp.z * perp.z))) * nn;
            s.dir = perp + para;
            s.bsdf = albedo * (ior2 * ior2) / (ior1 * ior1);
        }
        s.pdf = 1.f;
        s.bsdf = s.bsdf / fabsf(dot(s.dir, n));
        break; }
    case 3: {
        f3 wm = sample_normal_ggx(n, -1.f * wo, m.rough, sample2D(rng));
        s.dir = reflect(wo, wm);
        if (dot(s.dir, n) * dot(-1.f * wo, n) < 0) { s.bsdf = mk3(0, 0, 0); s.pdf = 0; return; }
        s.bsdf = microfacet_bsdf(n, -1.f * wo, s.dir, albedo, m.rough);
        s.pdf = microfacet_pdf(n, -1.f * wo, s.dir, m.rough);
        break; }
    case 4:
        s.bsdf = m.albedo_member; s.pdf = 1.f; break;
    default: break;
    }
}
PTK_DI f3 mat_bsdf(const MatEval& m, f3 n, f3 wo, f3 wi) {       // Material::BSDF, material.h:483-501
    f3 albedo = m.albedo;
    switch (m.type) {
    case 0: return m.albedo_pi;
    case 3: return microfacet_bsdf(n, -1.f * wo, wi, albedo, m.rough);
    case 1: return metallic_bsdf(n, -1.f * wo, wi, albedo, m.rough, m.metal);
    default: return mk3(0, 0, 0);
    }
}
PTK_DI float mat_pdf(const MatEval& m, f3 n, f3 wo, f3 wi) {     // Material::pdf, material.h:503-521
    switch (m.type) {
    case 0: return dot(wi, n) / PTK_PI;
    case 3: return microfacet_pdf(n, -1.f * wo, wi, m.rough);
    case 1: return metallic_pdf(n, -1.f * wo, wi, m.rough, m.metal);
    default: return 0.f;
    }
}

// ---- light sampling (lightSample.h) -----------------------------------------------------------------
struct LightRec { f3 pos, emit; float pdf; bool is_zero; };
// A shadow ray whose occlusion test is deferred (fused kernel, MIS / DIRECT): light_sample writes the ray and reports the sample
// as unoccluded; the caller stores the contribution it WOULD add in `contrib` and adds it after the block's shadow phase if the ray is free.
// The ray itself is parked in the caller's LDS slot at once — (origin, minT), (direction, pixel) — so that it is not held in registers while the
// BSDF of the sample is evaluated.
struct ShadowReq { bool on; float4* slot; int dstride; int pix; f3 contrib; };      // slot[0] = (origin, minT), slot[dstride] = (direction, pixel)

// LightSampler::lightPDF, lightSample.h:141-169
// Opt-in "fixed" cube lights (ptk_set_light_fix; NOT reference behaviour): world-space areas of the three face
// pairs of the unit cube under the primitive's transform; total = the cube's surface area.
PTK_DI void cube_face_areas(const DevGeom& g, float a[3], float& total) {
    const f3 ex = mk3(g.fwd[0], g.fwd[4], g.fwd[8]), ey = mk3(g.fwd[1], g.fwd[5], g.fwd[9]), ez = mk3(g.fwd[2], g.fwd[6], g.fwd[10]);
    a[0] = length(cross(ey, ez)); a[1] = length(cross(ez, ex)); a[2] = length(cross(ex, ey));
    total = 2.f * ((a[0] + a[1]) + a[2]);
}

PTK_DI float light_pdf(const SceneView& S, f3 viewPos, f3 lightPos, f3 normal, int triID, int geomID) {
    float pdf = -1.f;
    float L = (float)S.n_lights;
    if (triID >= 0) {
        const DevTriV& T = S.triv[triID];
        f3 v0 = mk3(T.v0[0], T.v0[1], T.v0[2]);
        float area = length(cross(mk3(T.v1[0], T.v1[1], T.v1[2]) - v0, mk3(T.v2[0], T.v2[1], T.v2[2]) - v0)) / 2.f;
        pdf = 1.f / L;
        f3 dl = lightPos - viewPos;
        pdf = pdf * dot(dl, dl) / (area * fabsf(dot(normalize(viewPos - lightPos), normal)));
    }
    const DevGeom& g = S.geoms[geomID];
    if (g.type == 0) {
        f3 vl = xform_point(g.inv, viewPos);
        float s2 = (0.5f * 0.5f) / dot(vl, vl);
        float cmax = sqrtf(smax(0.0f, 1.0f - s2));
        pdf = 1.0f / (PTK_TWO_PI * (1 - cmax) * L);
    }
    if (S.light_fix && triID < 0 && g.type == 1) {          // fixed cube light: area pdf over the whole surface -> solid angle
        float a[3], area; cube_face_areas(g, a, area);
        pdf = 1.f / L;
        f3 dl = lightPos - viewPos;
        pdf = pdf * dot(dl, dl) / (area * fabsf(dot(normalize(viewPos - lightPos), normal)));
    }
    return pdf;
}

// LightSampler::lightSample, lightSample.h:171-248.  rng is a COPY (by-value parameter in the reference).
PTK_DI void light_sample(const SceneView& S, const GeomStage& gs, const DevMaterial* mats, f3 viewPos, Rng rng, LightRec& rec, ShadowReq* defer = nullptr) {
    rec.pos = mk3(0, 0, 0); rec.emit = mk3(0, 0, 0); rec.pdf = 0.f; rec.is_zero = true;
    int Li = S.n_lights;
    if (Li == 0) return;
    float L = (float)Li;
    int lightID = (int)gmin(sample1D(rng) * L, L - 1.f);
    DevLight light = S.lights[lightID];
    const DevGeom& geom = S.geoms[light.geomID];
    f3 lightPos = mk3(0, 0, 0); float pdf = 0.f;
    if (light.triangleID >= 0) {
        f2 r = sample2D(rng);
        float tt = sqrtf(r.x);
        float u = 1.f - tt, v = tt * (1.f - r.y);
        const DevTriV& T = S.triv[light.triangleID];
        const DevTriA& A = S.tria[light.triangleID];
        f3 v0 = mk3(T.v0[0], T.v0[1], T.v0[2]), v1 = mk3(T.v1[0], T.v1[1], T.v1[2]), v2 = mk3(T.v2[0], T.v2[1], T.v2[2]);
        float w = 1 - u - v;
        lightPos = u * v0 + v * v1 + w * v2;
        f3 normal = normalize(u * mk3(A.n0[0], A.n0[1], A.n0[2]) + v * mk3(A.n1[0], A.n1[1], A.n1[2]) + w * mk3(A.n2[0], A.n2[1], A.n2[2]));
        float area = length(cross(v1 - v0, v2 - v0)) / 2.f;
        pdf = 1.f / L;
        f3 dl = lightPos - viewPos;
        pdf = pdf * dot(dl, dl) / (area * fabsf(dot(normalize(viewPos - lightPos), normal)));
    } else if (light.type == 0) {
        f2 xi = sample2D(rng);
        f3 vl = xform_point(geom.inv, viewPos);
        f3 c2r = normalize(mk3(0.f - vl.x, 0.f - vl.y, 0.f - vl.z));
        f3 tan, bit; pixar_basis(c2r, tan, bit);
        float s2 = (0.5f * 0.5f) / dot(vl, vl);
        float cmax = sqrtf(smax(0.0f, 1.0f - s2));
        float cosT = (1.0f - xi.x) + xi.x * cmax;
        float sinT = sqrtf(smax(0.f, 1.0f - cosT * cosT));
        float phi = xi.y * PTK_TWO_PI;
        float dc = length(mk3(0.f - vl.x, 0.f - vl.y, 0.f - vl.z));
        float ds = dc * cosT - sqrtf(smax(0.0f, 0.5f * 0.5f - dc * dc * sinT * sinT));
        float sinA = ds * sinT / 0.5f;
        float cosA = sqrtf(gmax(0.f, 1.f - sinA * sinA));
        float sp, cp; sincos_ptk(phi, sp, cp);
        f3 nObj = sinA * cp * tan + sinA * sp * bit + cosA * -c2r;
        f3 pObj = nObj * 0.5f;
        lightPos = xform_point(geom.fwd, pObj);
        pdf = 1.0f / (PTK_TWO_PI * (1 - cmax) * L);
    } else if (S.light_fix && light.type == 1) {
        // fixed cube light: face in proportion to its area (+x -x +y -y +z -z), uniform point on it
        float a[3], area; cube_face_areas(geom, a, area);
        const float uf = sample1D(rng) * area;
        const f2 r = sample2D(rng);
        int axis = 2; float sg = -1.f, acc = 0.f;
        bool done = false;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                if (!done) {
                    acc = acc + a[k];
                    if (uf < acc) { axis = k; sg = s2 == 0 ? 1.f : -1.f; done = true; }
                }
            }
        }
        const float c0 = r.x - 0.5f, c1 = r.y - 0.5f;
        f3 pObj, nObj;
        if (axis == 0) { pObj = mk3(0.5f * sg, c0, c1); nObj = mk3(sg, 0.f, 0.f); }
        else if (axis == 1) { pObj = mk3(c1, 0.5f * sg, c0); nObj = mk3(0.f, sg, 0.f); }
        else { pObj = mk3(c0, c1, 0.5f * sg); nObj = mk3(0.f, 0.f, sg); }
        lightPos = xform_point(geom.fwd, pObj);
        const f3 normal = normalize(xform_nrm(geom.invT, nObj));
        pdf = 1.f / L;
        f3 dl = lightPos - viewPos;
        pdf = pdf * dot(dl, dl) / (area * fabsf(dot(normalize(viewPos - lightPos), normal)));
    } else {
        // cube light: the reference assigns neither position nor pdf; the sample evaluates to 0 / NaN->0 whatever
        // the occlusion test returns (see file header), so the shadow ray is not traced.
        rec.pdf = 0.f; return;
    }
    f3 rayDir = normalize(lightPos - viewPos);
    if (defer) {
        const f3 ori = viewPos + 1e-5f * rayDir;
        defer->slot[0] = make_float4(ori.x, ori.y, ori.z, length(lightPos - ori));
        defer->slot[defer->dstride] = make_float4(rayDir.x, rayDir.y, rayDir.z, __int_as_float(defer->pix));
        defer->on = true;
    } else if (occluded(S, gs, viewPos + 1e-5f * rayDir, rayDir, lightPos)) { rec.emit = mk3(0, 0, 0); rec.pdf = -1.f; return; }
    const DevMaterial& lm = mats[geom.materialid];
    rec.emit = mk3(lm.albedo[0], lm.albedo[1], lm.albedo[2]);
    rec.pdf = pdf; rec.pos = lightPos; rec.is_zero = false;
}

} // namespace ptk
