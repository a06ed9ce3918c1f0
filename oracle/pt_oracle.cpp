// pt_oracle.cpp — CPU ORACLE for the per-bounce path-tracing inner loop.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported, linked or
// executed by the product (project3-cuda-path-tracer-1_amd/).  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg load liboracle.so,
// and only as the checker / reported CPU baseline.
//
// What it is: a single-thread, scalar, plain-C++ restatement of the algorithm of
// zhiyi1801/Project3-CUDA-Path-Tracer-1 for the hot path (camera rays ->
// primitive/BVH intersection -> BSDF/MIS shading -> stable stream compaction)
// plus the host prerequisites that feed it (scene-file parser, OBJ reader, SAH
// BVH build + 6-direction threaded BVH, camera fix-ups of main.cpp).  Every
// function cites the reference file:line it follows.  Own vec/mat types; the
// GLM 0.9.6.3 semantics the reference relies on (zero-init vectors, evaluation
// order of dot / mat*vec / normalize / mix / min / max) are restated in the helpers.
//
// PARITY PINNING: the reference's sources all include <cuda_runtime.h>/<thrust/*>, which this image lacks; building
// them would need stand-in headers, so the reference is treated as unbuildable here and there is no oracle/_ref.
// It has no tests or golden vectors, but it HOLDS SIX RENDERS of a scene on this path — gallery/{Direct,BSDF,MIS}{20,2000}.png
// = scenes/mis_test2.txt in the three sample modes — and this file is pinned to them (tests/test_gallery.py, fixture
// tests/golden/gallery_mis_test2.npz): with the fitted slot-shift table (see sample2D below) it reproduces the 20-spp images
// pixel for pixel on 99.4 % (Direct), 99.4 % (BSDF) and 95.1 % (MIS; 98.9 % within one 8-bit level) of the lit pixels.
// Second anchor: the [probe] values recorded in SURVEY.md §6/§8 (numbers the surveyor obtained from the reference's own
// host-callable code, g++ build); tests/test_oracle_pins.py checks this file against every one of them (RNG, box probe,
// scatterSample records, BVH node table, per-bounce live-path counts, mean radiance).
//
// Third-party arithmetic restated here: thrust::default_random_engine (minstd_rand)
// + uniform_real_distribution<float> (thrust is not vendored by the reference;
// restated from rocThrust 7.2 in /opt/rocm/include/thrust/random).
//
// Trig modes: the reference calls sinf/cosf of the platform (CUDA's on device).
// mode 0 = glibc sinf/cosf (used for the SURVEY pins, which were produced with
// glibc); mode 1 = the "ptk sincos" specification below (Cody-Waite reduction by
// pi/2 + cephes-style minimax polynomials, IEEE + - * only, no FMA), which the HIP
// kernels implement independently so that GPU and oracle agree BIT-EXACTLY.
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off, no -ffast-math).

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cfloat>
#include <cstdint>
#include <string>
#include <vector>
#include <map>
#include <fstream>
#include <sstream>
#include <iterator>
#include <algorithm>
#include <chrono>

// ---------------------------------------------------------------------------------------
// vector / matrix helpers with GLM 0.9.6.3 evaluation order
// ---------------------------------------------------------------------------------------
struct v2 { float x = 0, y = 0; };
struct v3 {
    float x = 0, y = 0, z = 0;                       // glm 0.9.6.3 default ctor zero-inits
    v3() {}
    v3(float a, float b, float c) : x(a), y(b), z(c) {}
    float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
struct m4 { float c[4][4]; };                         // column-major: c[col][row]

static inline v3 operator+(v3 a, v3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 operator-(v3 a, v3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 operator-(v3 a) { return v3(-a.x, -a.y, -a.z); }
static inline v3 operator*(v3 a, v3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 operator*(v3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline v3 operator*(float s, v3 a) { return v3(s * a.x, s * a.y, s * a.z); }
static inline v3 operator/(v3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
static inline v3 operator/(v3 a, v3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
// glm func_common.inl:409-435
static inline float gmin(float x, float y) { return x < y ? x : y; }
static inline float gmax(float x, float y) { return x > y ? x : y; }
static inline float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
static inline v3 vmin(v3 a, v3 b) { return v3(gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z)); }
static inline v3 vmax(v3 a, v3 b) { return v3(gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)); }
// glm func_geometric.inl: compute_dot<tvec3> = tmp.x + tmp.y + tmp.z
static inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 cross(v3 x, v3 y) {
    return v3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
static inline float length(v3 v) { return sqrtf(dot(v, v)); }
// normalize(x) = x * inversesqrt(dot(x,x)); inversesqrt(x) = 1 / sqrt(x)
static inline v3 normalize(v3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
// reflect(I,N) = I - N * dot(N,I) * 2
static inline v3 reflect(v3 I, v3 N) { return I - N * dot(N, I) * 2.0f; }
// mix(x,y,a) = x + a*(y-x)
static inline float mixf(float x, float y, float a) { return x + a * (y - x); }
static inline v3 mixv(v3 x, v3 y, float a) { return x + a * (y - x); }
static inline v3 mixvv(v3 x, v3 y, v3 a) { return x + a * (y - x); }

// vec3(m * vec4(v, w)): glm type_mat4x4.inl:617-628  (m0*v0 + m1*v1) + (m2*v2 + m3*v3)
static inline v3 mulMV(const m4& m, v3 v, float w) {
    v3 r;
    for (int i = 0; i < 3; ++i)
        r[i] = (m.c[0][i] * v.x + m.c[1][i] * v.y) + (m.c[2][i] * v.z + m.c[3][i] * w);
    return r;
}
static inline m4 m4_identity() {
    m4 m; memset(&m, 0, sizeof m);
    m.c[0][0] = m.c[1][1] = m.c[2][2] = m.c[3][3] = 1.f; return m;
}
// glm type_mat4x4.inl:686-704
static m4 m4_mul(const m4& a, const m4& b) {
    m4 r;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            r.c[j][i] = a.c[0][i] * b.c[j][0] + a.c[1][i] * b.c[j][1] + a.c[2][i] * b.c[j][2] + a.c[3][i] * b.c[j][3];
    return r;
}
// glm gtc/matrix_transform.inl translate(): Result[3] = m[0]*v[0] + m[1]*v[1] + m[2]*v[2] + m[3]
static m4 m4_translate(const m4& m, v3 v) {
    m4 r = m;
    for (int i = 0; i < 4; ++i) r.c[3][i] = m.c[0][i] * v.x + m.c[1][i] * v.y + m.c[2][i] * v.z + m.c[3][i];
    return r;
}
// glm gtc/matrix_transform.inl rotate()
static m4 m4_rotate(const m4& m, float angle, v3 v) {
    float a = angle, c = cosf(a), s = sinf(a);
    v3 axis = normalize(v);
    v3 temp = (1.f - c) * axis;
    float R[3][3];
    R[0][0] = c + temp[0] * axis[0];
    R[0][1] = 0 + temp[0] * axis[1] + s * axis[2];
    R[0][2] = 0 + temp[0] * axis[2] - s * axis[1];
    R[1][0] = 0 + temp[1] * axis[0] - s * axis[2];
    R[1][1] = c + temp[1] * axis[1];
    R[1][2] = 0 + temp[1] * axis[2] + s * axis[0];
    R[2][0] = 0 + temp[2] * axis[0] + s * axis[1];
    R[2][1] = 0 + temp[2] * axis[1] - s * axis[0];
    R[2][2] = c + temp[2] * axis[2];
    m4 r;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 4; ++i)
            r.c[j][i] = m.c[0][i] * R[j][0] + m.c[1][i] * R[j][1] + m.c[2][i] * R[j][2];
    for (int i = 0; i < 4; ++i) r.c[3][i] = m.c[3][i];
    return r;
}
static m4 m4_scale(const m4& m, v3 v) {
    m4 r;
    for (int i = 0; i < 4; ++i) { r.c[0][i] = m.c[0][i] * v.x; r.c[1][i] = m.c[1][i] * v.y; r.c[2][i] = m.c[2][i] * v.z; r.c[3][i] = m.c[3][i]; }
    return r;
}
// glm type_mat4x4.inl:37-90 compute_inverse
static m4 m4_inverse(const m4& M) {
    const float (*m)[4] = M.c;
    float Coef00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float Coef02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float Coef03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float Coef04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float Coef06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float Coef07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float Coef08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float Coef10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float Coef11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float Coef12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float Coef14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float Coef15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float Coef16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float Coef18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float Coef19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float Coef20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float Coef22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float Coef23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    float Fac0[4] = {Coef00, Coef00, Coef02, Coef03};
    float Fac1[4] = {Coef04, Coef04, Coef06, Coef07};
    float Fac2[4] = {Coef08, Coef08, Coef10, Coef11};
    float Fac3[4] = {Coef12, Coef12, Coef14, Coef15};
    float Fac4[4] = {Coef16, Coef16, Coef18, Coef19};
    float Fac5[4] = {Coef20, Coef20, Coef22, Coef23};
    float Vec0[4] = {m[1][0], m[0][0], m[0][0], m[0][0]};
    float Vec1[4] = {m[1][1], m[0][1], m[0][1], m[0][1]};
    float Vec2[4] = {m[1][2], m[0][2], m[0][2], m[0][2]};
    float Vec3[4] = {m[1][3], m[0][3], m[0][3], m[0][3]};
    const float SignA[4] = {+1, -1, +1, -1}, SignB[4] = {-1, +1, -1, +1};
    m4 Inv;
    for (int i = 0; i < 4; ++i) {
        float Inv0 = Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i] + Vec3[i] * Fac2[i];
        float Inv1 = Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i] + Vec3[i] * Fac4[i];
        float Inv2 = Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i] + Vec3[i] * Fac5[i];
        float Inv3 = Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i] + Vec2[i] * Fac5[i];
        Inv.c[0][i] = Inv0 * SignA[i];
        Inv.c[1][i] = Inv1 * SignB[i];
        Inv.c[2][i] = Inv2 * SignA[i];
        Inv.c[3][i] = Inv3 * SignB[i];
    }
    float D0 = m[0][0] * Inv.c[0][0], D1 = m[0][1] * Inv.c[1][0], D2 = m[0][2] * Inv.c[2][0], D3 = m[0][3] * Inv.c[3][0];
    float Dot1 = (D0 + D1) + (D2 + D3);
    float OneOverDet = 1.f / Dot1;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) Inv.c[j][i] = Inv.c[j][i] * OneOverDet;
    return Inv;
}
// glm gtc/matrix_inverse.inl:95-147 inverseTranspose(mat4)
static m4 m4_inverse_transpose(const m4& M) {
    const float (*m)[4] = M.c;
    float S00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float S01 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float S02 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float S03 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float S04 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float S05 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float S06 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float S07 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float S08 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float S09 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float S10 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float S11 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float S12 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float S13 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float S14 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float S15 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float S16 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float S17 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float S18 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    m4 I;
    I.c[0][0] = +(m[1][1] * S00 - m[1][2] * S01 + m[1][3] * S02);
    I.c[0][1] = -(m[1][0] * S00 - m[1][2] * S03 + m[1][3] * S04);
    I.c[0][2] = +(m[1][0] * S01 - m[1][1] * S03 + m[1][3] * S05);
    I.c[0][3] = -(m[1][0] * S02 - m[1][1] * S04 + m[1][2] * S05);
    I.c[1][0] = -(m[0][1] * S00 - m[0][2] * S01 + m[0][3] * S02);
    I.c[1][1] = +(m[0][0] * S00 - m[0][2] * S03 + m[0][3] * S04);
    I.c[1][2] = -(m[0][0] * S01 - m[0][1] * S03 + m[0][3] * S05);
    I.c[1][3] = +(m[0][0] * S02 - m[0][1] * S04 + m[0][2] * S05);
    I.c[2][0] = +(m[0][1] * S06 - m[0][2] * S07 + m[0][3] * S08);
    I.c[2][1] = -(m[0][0] * S06 - m[0][2] * S09 + m[0][3] * S10);
    I.c[2][2] = +(m[0][0] * S11 - m[0][1] * S09 + m[0][3] * S12);
    I.c[2][3] = -(m[0][0] * S08 - m[0][1] * S10 + m[0][2] * S12);
    I.c[3][0] = -(m[0][1] * S13 - m[0][2] * S14 + m[0][3] * S15);
    I.c[3][1] = +(m[0][0] * S13 - m[0][2] * S16 + m[0][3] * S17);
    I.c[3][2] = -(m[0][0] * S14 - m[0][1] * S16 + m[0][3] * S18);
    I.c[3][3] = +(m[0][0] * S15 - m[0][1] * S17 + m[0][2] * S18);
    float Det = +m[0][0] * I.c[0][0] + m[0][1] * I.c[0][1] + m[0][2] * I.c[0][2] + m[0][3] * I.c[0][3];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) I.c[j][i] = I.c[j][i] / Det;
    return I;
}

// ---------------------------------------------------------------------------------------
// constants (utilities.h:13-29)
// ---------------------------------------------------------------------------------------
#define PI_F     3.1415926535897932384626422832795028841971f
#define TWO_PI_F 6.2831853071795864769252867665590057683943f
#define ROUGHNESS_MIN 1e-3
#define ROUGHNESS_MAX 1.f
#define MAX_PRIM 1
#define BUCKET_NUM 20

// ---------------------------------------------------------------------------------------
// trig: mode 0 = libm, mode 1 = "ptk sincos" spec (shared spec with the HIP kernels)
// ---------------------------------------------------------------------------------------
static int g_trig_mode = 1;

// ptk sincos spec.  Valid for |x| up to ~1e4 (used on [0, 2pi]).
//   k  = (int)(x * 2/pi + 0.5)                    (x >= 0 on the hot path)
//   r  = ((x - k*P1) - k*P2) - k*P3               (pi/2 split in 3 parts, P1 has 8 significant bits)
//   z  = r*r
//   s  = r + r*z*(S1 + z*(S2 + z*S3))             (cephes sinf minimax on [-pi/4, pi/4])
//   c  = (1 - 0.5*z) + z*z*(C1 + z*(C2 + z*C3))   (cephes cosf minimax)
//   quadrant k&3: 0:(s,c) 1:(c,-s) 2:(-s,-c) 3:(-c,s)
// Every operation is a single IEEE-754 binary32 +,-,* (no FMA, no reassociation).
static inline void ptk_sincos(float x, float* sn, float* cs) {
    const float TWO_OVER_PI = 0.636619772367581343f;
    const float P1 = 1.5703125f;
    const float P2 = 4.837512969970703125e-4f;
    const float P3 = 7.54978995489188216e-8f;
    float fk = floorf(x * TWO_OVER_PI + 0.5f);
    int k = (int)fk;
    float r = ((x - fk * P1) - fk * P2) - fk * P3;
    float z = r * r;
    float s = r + r * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
    float c = (1.0f - 0.5f * z) + z * z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
    switch (k & 3) {
        case 0: *sn = s;  *cs = c;  break;
        case 1: *sn = c;  *cs = -s; break;
        case 2: *sn = -s; *cs = -c; break;
        default: *sn = -c; *cs = s; break;
    }
}
static inline void trig_sincos(float x, float* sn, float* cs) {
    if (g_trig_mode == 0) { *sn = sinf(x); *cs = cosf(x); }
    else ptk_sincos(x, sn, cs);
}

// ptk atan2 spec (cephes atanf / atan2f structure, IEEE + - * / only); mode 0 uses glibc atan2f.
//   atan(x): s = x < 0, x = |x|;  x > tan(3pi/8): y = pi/2, x = -(1/x);  x > tan(pi/8): y = pi/4, x = (x-1)/(x+1);  else y = 0
//            z = x*x;  y += (((8.05374449538e-2 z - 1.38776856032e-1) z + 1.99777106478e-1) z - 3.33329491539e-1) z x + x;  s: y = -y
//   atan2(y,x): x == 0: y < 0 -> -pi/2, y == 0 -> 0, else pi/2;  y == 0: x < 0 -> pi, else 0;
//               w = 0 (x > 0), pi (x < 0, y >= 0), -pi (x < 0, y < 0);  result = w + atan(y/x)
static inline float ptk_atan(float x) {
    bool neg = x < 0; if (neg) x = -x;
    float y;
    if (x > 2.414213562373095f) { y = 1.5707963267948966192f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483096f; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.f;
    float z = x * x;
    y += (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
    return neg ? -y : y;
}
static inline float ptk_atan2(float y, float x) {
    const float PIF = 3.141592653589793238f, PIO2F = 1.5707963267948966192f;
    if (x == 0) { if (y < 0) return -PIO2F; if (y == 0) return 0.f; return PIO2F; }
    if (y == 0) return x < 0 ? PIF : 0.f;
    float w = x > 0 ? 0.f : (y < 0 ? -PIF : PIF);
    return w + ptk_atan(y / x);
}
static inline float trig_atan2(float y, float x) { return g_trig_mode == 0 ? atan2f(y, x) : ptk_atan2(y, x); }

// ---------------------------------------------------------------------------------------
// RNG  (intersections.h:12-20, pathtrace.cu:25-29, thrust minstd_rand + uniform_real<float>)
// ---------------------------------------------------------------------------------------
static inline uint32_t utilhash(uint32_t a) {
    a = (a + 0x7ed55d16) + (a << 12);
    a = (a ^ 0xc761c23c) ^ (a >> 19);
    a = (a + 0x165667b1) + (a << 5);
    a = (a + 0xd3a2646c) ^ (a << 9);
    a = (a + 0xfd7046c5) + (a << 3);
    a = (a ^ 0xb55a4f09) ^ (a >> 16);
    return a;
}
struct Rng { uint32_t x; };
static inline uint32_t seed_hash(int iter, int index, int depth) {
    return utilhash((uint32_t)((1u << 31) | ((uint32_t)depth << 22) | (uint32_t)iter)) ^ utilhash((uint32_t)index);
}
static inline Rng make_rng(int iter, int index, int depth) {
    uint32_t h = seed_hash(iter, index, depth);
    Rng r; r.x = h % 2147483647u; if (r.x == 0) r.x = 1; return r;
}
static inline float sample1D(Rng& r) {
    // Schrage step of linear_congruential_engine<uint32, 48271, 0, 2^31-1> (thrust/random/detail/mod.h)
    const uint32_t a = 48271u, m = 2147483647u, q = m / a, rr = m % a;
    uint32_t x = r.x, t1 = a * (x % q), t2 = rr * (x / q);
    x = (t1 >= t2) ? (t1 - t2) : (m - t2 + t1);
    r.x = x;
    float res = (float)(x - 1u);
    res /= (1.0f + (float)(2147483646u - 1u));
    return res * (1.f - 0.f) + 0.f;
}
// utilities.h:54-56  sample2D = glm::vec2(sample1D(s), sample1D(s)).  C++ leaves the evaluation order of the two
// constructor arguments unspecified, and the two builds of the reference differ:
//  * nvcc device code (what the reference actually renders with) draws the FIRST argument first (x = 1st draw).  Settled
//    against the reference's own renders gallery/{Direct,MIS,BSDF}20.png: with x first, every one of the 20 iterations of
//    each image reproduces the gallery noise pixel for pixel (tests/golden/make_gallery_fixture.py, tests/test_gallery.py);
//    with y first none does.  This is the DEFAULT (g_s2d_xfirst = 1) and what the HIP kernels implement.
//  * the g++ host build the SURVEY [probe] records came from draws the SECOND argument first; orc_set_sample2d_order(0)
//    selects that order so the oracle stays checkable against those records (tests/test_oracle_pins.py).
// g_dbg_slot_shift is an analysis knob of the gallery test only: it offsets the compacted slot in the shading RNG key
// (the reference's GPU build flips a few sphere-silhouette hits per iteration, which shifts every later slot).
static int g_dbg_slot_shift = 0;
static int g_dbg_slot_shift_deep = 0, g_dbg_deep_on = 0;      // diagnostic (tools/diag/mis20_residue.py): a separate offset for depth >= 1
static int g_dbg_min_depth = 0;                                // diagnostic: only contributions made at depth >= this reach the image (MIS)
static inline int dbg_shift(int depth) { return (g_dbg_deep_on && depth >= 1) ? g_dbg_slot_shift_deep : g_dbg_slot_shift; }
static int g_dbg_mis_terms = 3;      // diagnostic: MisPTkernel terms that reach the image (orc_set_debug_mis_terms)
// tile partition (not a reference feature): rank `part` keys its shading RNG with slot + part * W * H * spp so that two ranks never
// share a random stream; 0 without a partition = the reference's key (pathtrace.cu:363,458,532)
static int g_slot_bias = 0;
static int g_s2d_xfirst = 1;
static inline v2 sample2D(Rng& r) { v2 o; if (g_s2d_xfirst) { o.x = sample1D(r); o.y = sample1D(r); } else { o.y = sample1D(r); o.x = sample1D(r); } return o; }

// ---------------------------------------------------------------------------------------
// scene-side PODs (interchange layouts, dword-for-dword what the tests exchange)
// ---------------------------------------------------------------------------------------
enum { GEOM_SPHERE = 0, GEOM_CUBE = 1, GEOM_OBJ = 2 };                     // sceneStructs.h:11-15
enum { MAT_LAMBERTIAN = 0, MAT_METALLIC = 1, MAT_DIELECTRIC = 2, MAT_MICROFACET = 3, MAT_LIGHT = 4 }; // material.h:32-38
enum { MODE_BSDF = 0, MODE_DIRECTLI = 1, MODE_MIS = 2 };                   // preview.h:13

struct Geom {            // 50 dwords
    int type, materialid;
    m4 transform, inverseTransform, invTranspose;
};
struct Material {        // 14 dwords.  albedo = Material::albedo (default 1); *_s = devTexSampler::fixVal (default 0); *MapID: texture or -1
    int type;
    float albedo[3];
    float albedo_s[3];
    float rough_s, metal_s, ior;
    int albedoMapID, metallicMapID, roughnessMapID, normalMapID;
};
struct Texture { int w = 0, h = 0; std::vector<float> rgb; std::string path; };   // class image (image.h:8-51): row 0 = bottom row
struct Triangle {        // 31 dwords = 124 B (Bounds3.hpp:250-258)
    v3 v[3], n[3];
    v2 tex[3];
    v3 tangent, bitangent;
    int geomIdx = -1;
};
struct BvhNode {         // 10 dwords = 40 B (BVH.h:44-50)
    v3 pMin, pMax;
    int start, end, miss, hit;
};
struct LightPrim { int geomID, triangleID, type; };   // lightSample.h:26-34
struct Camera {          // 21 dwords = 84 B (sceneStructs.h:61-70)
    int res[2];
    v3 position, lookAt, view, up, right;
    float fov[2], pixelLength[2];
};
struct PathSegment {     // 12 dwords (sceneStructs.h:80-86)
    v3 o, d, color;
    int pixelIndex, remainingBounces;
    float prevPdf;
};
struct Isect {           // 12 dwords (sceneStructs.h:91-99)
    float t;
    v3 n, p;
    v2 uv;
    int materialId, geomID, triangleID;
};

struct OrcScene {
    std::vector<Geom> geoms;
    std::vector<v3> geomScale;
    std::vector<std::string> geomMeshFile;
    std::vector<Material> materials;
    std::map<std::string, int> materialNameMap, geomNameMap;
    std::map<std::string, std::vector<Triangle>> meshPool;
    std::vector<Triangle> triangles;
    std::vector<BvhNode> bvh;          // 6 * bvh_size
    int bvh_size = 0;
    std::vector<LightPrim> lights;
    std::vector<Texture> textures;     // pixel data is supplied by the test harness (orc_scene_set_texture): the oracle has no image decoder
    int envMapID = -1;
    Camera cam;
    int iterations = 0, traceDepth = 0;
    std::string imageName, sceneDir;
    float theta = 0, phi = 0; bool posInit = true;
    int ovr_w = -1, ovr_h = -1, ovr_depth = -1;
    std::string error;
};

// ---------------------------------------------------------------------------------------
// textures (image.h:53-110)
// ---------------------------------------------------------------------------------------
static inline v3 texel(const Texture& t, int x, int y) {
    x = x < 0 ? 0 : (x >= t.w ? t.w - 1 : x);              // unchecked in the reference; clamped (only differs where that is UB)
    y = y < 0 ? 0 : (y >= t.h ? t.h - 1 : y);
    const float* p = &t.rgb[3 * ((size_t)y * t.w + x)];
    return v3(p[0], p[1], p[2]);
}
static v3 linearSample(const Texture& t, v2 uv) {           // devTexObj::linearSample, image.h:73-85
    float u = uv.x, v = uv.y;
    float x = u * (t.w - 1), y = v * (t.h - 1);
    int lx = x, ux = x + 1 >= t.w ? lx : lx + 1;
    int ly = y, uy = y + 1 >= t.h ? ly : ly + 1;
    float fx = x - floorf(x), fy = y - floorf(y);
    v3 p1 = mixv(texel(t, lx, ly), texel(t, ux, ly), fx);
    v3 p2 = mixv(texel(t, lx, uy), texel(t, ux, uy), fx);
    return mixv(p1, p2, fy);
}
static inline v2 sphere2Plane(v3 dir) {                     // utilities.h:209-214 ("* InvPI" == "* 1.f/PI")
    v2 r;
    float a = trig_atan2(dir.z, dir.x) * 1.f / PI_F * .5f + 1.f;
    r.x = a - floorf(a);
    r.y = gmax(trig_atan2(dir.y, sqrtf(dir.x * dir.x + dir.z * dir.z)) * 1.f / PI_F + 0.5f, 0.f);
    return r;
}

// ---------------------------------------------------------------------------------------
// primitives
// ---------------------------------------------------------------------------------------
struct Ray { v3 origin, direction; };

// intersections.h:27-29
static inline v3 getPointOnRay(const Ray& r, float t) { return r.origin + (t - .0001f) * normalize(r.direction); }

// intersections.h:48-90
static float boxIntersectionTest(const Geom& box, const Ray& r, v3& P, v3& N) {
    Ray q;
    q.origin = mulMV(box.inverseTransform, r.origin, 1.0f);
    q.direction = normalize(mulMV(box.inverseTransform, r.direction, 0.0f));
    float tmin = -1e38f, tmax = 1e38f;
    v3 tmin_n, tmax_n;
    for (int xyz = 0; xyz < 3; ++xyz) {
        float qd = q.direction[xyz];
        float t1 = (-0.5f - q.origin[xyz]) / qd;
        float t2 = (+0.5f - q.origin[xyz]) / qd;
        float ta = gmin(t1, t2), tb = gmax(t1, t2);
        v3 n; n[xyz] = t2 < t1 ? +1.f : -1.f;
        if (ta > 0 && ta > tmin) { tmin = ta; tmin_n = n; }
        if (tb < tmax) { tmax = tb; tmax_n = n; }
    }
    if (tmax >= tmin && tmax > 0) {
        if (tmin <= 0) { tmin = tmax; tmin_n = tmax_n; }
        P = mulMV(box.transform, getPointOnRay(q, tmin), 1.0f);
        N = normalize(mulMV(box.invTranspose, tmin_n, 0.0f));
        return length(r.origin - P);
    }
    return -1;
}

// intersections.h:102-144
static float sphereIntersectionTest(const Geom& sph, const Ray& r, v3& P, v3& N) {
    v3 ro = mulMV(sph.inverseTransform, r.origin, 1.0f);
    v3 rd = normalize(mulMV(sph.inverseTransform, r.direction, 0.0f));
    Ray rt; rt.origin = ro; rt.direction = rd;
    float vDotDirection = dot(rt.origin, rt.direction);
    float radicand = vDotDirection * vDotDirection - (dot(rt.origin, rt.origin) - 0.25f);   // powf(.5,2) == .25 exactly
    if (radicand < 0) return -1;
    float squareRoot = sqrtf(radicand);
    float firstTerm = -vDotDirection;
    float t1 = firstTerm + squareRoot, t2 = firstTerm - squareRoot;
    float t = 0;
    if (t1 < 0 && t2 < 0) return -1;
    else if (t1 > 0 && t2 > 0) t = (t2 < t1) ? t2 : t1;       // min
    else t = (t1 < t2) ? t2 : t1;                             // max
    v3 obj = getPointOnRay(rt, t);
    P = mulMV(sph.transform, obj, 1.f);
    N = normalize(mulMV(sph.invTranspose, obj, 0.f));
    return length(r.origin - P);
}

// Bounds3.hpp:162-189
static bool aabbIntersectP(const v3& pMin, const v3& pMax, const Ray& ray, float& t) {
    v3 o = ray.origin;
    v3 invDir = v3(1.0f / ray.direction.x, 1.0f / ray.direction.y, 1.0f / ray.direction.z);
    float tEnter = -FLT_MAX, tExit = FLT_MAX;
    for (int i = 0; i < 3; ++i) {
        if (ray.direction[i] == 0) {
            if (o[i] < pMin[i] || o[i] > pMax[i]) return false;
        } else {
            float tMax = (pMax[i] - o[i]) * invDir[i];
            float tMin = (pMin[i] - o[i]) * invDir[i];
            if (ray.direction[i] < 0) std::swap(tMax, tMin);
            tEnter = tMin > tEnter ? tMin : tEnter;
            tExit = tMax < tExit ? tMax : tExit;
        }
    }
    t = tEnter;
    return tEnter <= tExit && tExit > 0;
}

// Bounds3.hpp:298-326 (Moller-Trumbore, no cull)
static bool triIntersect(const Triangle& T, const Ray& ray, float& t, float& u, float& v) {
    v3 e1 = T.v[1] - T.v[0], e2 = T.v[2] - T.v[0];
    v3 pvec = cross(ray.direction, e2);
    float det = dot(e1, pvec);
    if (det == 0) return false;
    float invDet = 1 / det;
    v3 tvec = ray.origin - T.v[0];
    u = dot(tvec, pvec); u *= invDet;
    v3 qvec = cross(tvec, e1);
    v = dot(ray.direction, qvec); v *= invDet;
    t = dot(e2, qvec); t *= invDet;
    if (t < 0 || u < 0 || v < 0 || (1 - u - v) < 0) return false;
    return true;
}

static inline int mtbvh_offset(v3 dir, int bvh_size) {       // pathtrace.cu:242-245
    int off = ((fabsf(dir[0]) > fabsf(dir[1])) && (fabsf(dir[0]) > fabsf(dir[2]))) ? 0 : (fabsf(dir[1]) > fabsf(dir[2]) ? 1 : 2);
    off = off + (dir[off] > 0 ? 0 : 3);
    return off * bvh_size;
}

struct TraverseStats { uint64_t nodes = 0, tris = 0; };
static TraverseStats g_stats;
// Optional multi-threading of the per-path loops (bench.py's all-cores CPU baseline): paths are independent inside the
// intersect and shade loops (own RNG key, own pixel), compaction stays serial and ordered, so results are identical.
static int g_threads = 1;
// traversal counters are accumulated per thread and folded into g_stats at the end of each loop (stats_fold)
static thread_local TraverseStats t_stats;
static inline void stats_fold() {
    if (t_stats.nodes | t_stats.tris) {
        __atomic_fetch_add(&g_stats.nodes, t_stats.nodes, __ATOMIC_RELAXED);
        __atomic_fetch_add(&g_stats.tris, t_stats.tris, __ATOMIC_RELAXED);
        t_stats = TraverseStats();
    }
}

// pathtrace.cu:169-346 computeIntersections body for one path.  Returns rayValid.
static int intersect_one(const OrcScene& S, const PathSegment& ps, Isect& out) {
    Ray ray; ray.origin = ps.o; ray.direction = ps.d;
    float t = -1;
    v3 P, N; v2 uv;
    float t_min = FLT_MAX;
    int hit_geom = -1, triId = -1;
    v3 tmpP, tmpN;
    for (int i = 0; i < (int)S.geoms.size(); ++i) {
        const Geom& g = S.geoms[i];
        if (g.type == GEOM_CUBE) t = boxIntersectionTest(g, ray, tmpP, tmpN);
        else if (g.type == GEOM_SPHERE) t = sphereIntersectionTest(g, ray, tmpP, tmpN);
        if (t > 0.0f && t_min > t) { t_min = t; hit_geom = i; P = tmpP; N = tmpN; }
    }
    const BvhNode* cur = S.bvh.data() + mtbvh_offset(ray.direction, S.bvh_size);
    int idx = 0; float tempT = FLT_MAX;
    while (idx != -1) {
        t_stats.nodes++;
        if (!aabbIntersectP(cur[idx].pMin, cur[idx].pMax, ray, tempT) || tempT > t_min) { idx = cur[idx].miss; continue; }
        if (cur[idx].end - cur[idx].start <= MAX_PRIM) {
            for (int i = cur[idx].start; i < cur[idx].end; ++i) {
                const Triangle& T = S.triangles[i];
                float u, v;
                t_stats.tris++;
                bool isHit = triIntersect(T, ray, t, u, v);
                if (isHit && t_min > t) {
                    t_min = t; hit_geom = T.geomIdx; triId = i;
                    float w = 1 - u - v;
                    P = w * T.v[0] + u * T.v[1] + v * T.v[2];
                    N = w * T.n[0] + u * T.n[1] + v * T.n[2];
                    uv.x = w * T.tex[0].x + u * T.tex[1].x + v * T.tex[2].x;
                    uv.y = w * T.tex[0].y + u * T.tex[1].y + v * T.tex[2].y;
                }
            }
        }
        idx = cur[idx].hit;
    }
    out = Isect();                                        // cudaMemset(dev_intersections1, 0, ...) pathtrace.cu:700
    if (hit_geom == -1) { out.t = -1.0f; return 0; }
    out.t = t_min; out.p = P;
    out.materialId = S.geoms[hit_geom].materialid;
    out.uv.x = gclamp(uv.x, 0.f, 1.f); out.uv.y = gclamp(uv.y, 0.f, 1.f);
    out.geomID = hit_geom; out.triangleID = triId;
    // pathtrace.cu:323-340.  The constant normal sampler is (0.5,0.5,1) (scene.cpp:311): localNorm = (0,0,1) -> plain normal.
    v3 normal = normalize(N);
    out.n = normal;
    const Material& hm = S.materials[out.materialId];
    if (hm.normalMapID >= 0 && hm.normalMapID < (int)S.textures.size()) {
        v3 mapped = linearSample(S.textures[hm.normalMapID], uv);               // unclamped texCoords, like the reference
        v3 localNorm = normalize(v3(mapped.x * 1.f - 0.5f, mapped.y * 1.f - 0.5f, mapped.z * 1.f - 0.5f));
        v3 T = triId >= 0 ? S.triangles[triId].tangent : v3();
        if (dot(T, T) > 1e-3f && (double)fabsf(localNorm.z - 1.f) > 1e-5) {
            v3 B = normalize(cross(normal, T));
            T = normalize(cross(B, normal));
            out.n = normalize(v3(T.x * localNorm.x + B.x * localNorm.y + normal.x * localNorm.z,
                                 T.y * localNorm.x + B.y * localNorm.y + normal.y * localNorm.z,
                                 T.z * localNorm.x + B.z * localNorm.y + normal.z * localNorm.z));
        }
    }
    return 1;
}

// lightSample.h:59-139
static bool occlusionTest(const OrcScene& S, v3 ori, v3 dir, v3 des) {
    float minT = length(des - ori);
    Ray ray; ray.origin = ori; ray.direction = dir;
    v3 nor, ip; float t = 0;
    for (int i = 0; i < (int)S.geoms.size(); ++i) {
        const Geom& g = S.geoms[i];
        if (g.type == GEOM_CUBE) t = boxIntersectionTest(g, ray, ip, nor);
        else if (g.type == GEOM_SPHERE) t = sphereIntersectionTest(g, ray, ip, nor);
        if (t > 0.0f && minT - 1e-5f > t && (double)fabsf(t - minT) > 1e-2) return true;
    }
    const BvhNode* cur = S.bvh.data() + mtbvh_offset(dir, S.bvh_size);
    int idx = 0; float tempT = FLT_MAX;
    while (idx != -1) {
        if (!aabbIntersectP(cur[idx].pMin, cur[idx].pMax, ray, tempT) || tempT > minT) { idx = cur[idx].miss; continue; }
        if (cur[idx].end - cur[idx].start <= MAX_PRIM) {
            for (int i = cur[idx].start; i < cur[idx].end; ++i) {
                float u, v;
                bool isHit = triIntersect(S.triangles[i], ray, t, u, v);
                if (isHit && minT - 1e-5f > t && (double)fabsf(t - minT) > 1e-4) return true;
            }
        }
        idx = cur[idx].hit;
    }
    return false;
}

// ---------------------------------------------------------------------------------------
// shading math (utilities.h math::, material.h)
// ---------------------------------------------------------------------------------------
static inline void localRefMatrix_Pixar(v3 n, v3& xp, v3& yp) {      // utilities.h:188-200
    float x = n.x, y = n.y, z = n.z;
    float sz = z >= 0 ? 1.f : -1.f;
    float a = 1.0f / (sz + z);
    float sx = sz * x;
    float b = x * y * a;
    xp = v3(sx * x * a - 1.f, sz * b, sx);
    yp = v3(b, y * y * a - sz, y);
}
static inline v3 sampleHemisphereCosine(v3 n, v2 r) {                // utilities.h:224-235, 136-146
    v3 u, v; v3 unit_w = normalize(n);
    localRefMatrix_Pixar(n, u, v);
    float r1 = r.x, r2 = r.y;
    float sinTheta = sqrtf(r1), cosTheta = sqrtf(1 - r1);
    float phi = TWO_PI_F * r2;
    float sp, cp; trig_sincos(phi, &sp, &cp);
    v3 xyz(sinTheta * cp, sinTheta * sp, cosTheta);
    return xyz.x * u + xyz.y * v + xyz.z * unit_w;
}
static inline float pow5(float x) { float x2 = x * x; return x2 * x2 * x; }
static inline float processNANf(float x) { return (x != x || std::isinf(x)) ? 0 : x; }
static inline v3 processNAN(v3 v) { return v3(processNANf(v.x), processNANf(v.y), processNANf(v.z)); }
static inline float lengthSquared(v3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
static inline v3 getReflectDir(v3 n, v3 wo) {                         // utilities.h:290-294
    v3 normal = dot(wo, n) < 0 ? n : -n;
    return wo - 2.f * (normal * dot(wo, normal));
}
static inline v3 getRefractDir(v3 n, v3 wo, float ior1, float ior2) { // utilities.h:311-320
    v3 normal = dot(wo, n) < 0 ? n : -n;
    v3 r_perp = (wo - dot(wo, normal) * normal) * (ior1 / ior2);
    v3 r_para = (-1 * sqrtf(1 - lengthSquared(r_perp))) * normal;
    return r_perp + r_para;
}
static inline v3 FresnelSchlick(v3 f0, float c) { return f0 + (v3(1.f, 1.f, 1.f) - f0) * pow5(1 - c); }  // utilities.h:327-330
static inline float FresnelMaxwell(float c1, float ior1, float ior2) {                              // utilities.h:332-342
    float s1 = sqrtf(gmax(1 - c1 * c1, 0.f));
    float s2 = s1 * ior1 / ior2;
    if (s2 > 1) return 1;
    float c2 = sqrtf(gmax(1 - s2 * s2, 0.f));
    float r_para = (ior1 * c2 - ior2 * c1) / (ior1 * c2 + ior2 * c1);
    float r_perp = (ior1 * c1 - ior2 * c2) / (ior1 * c1 + ior2 * c2);
    return (r_para * r_para + r_perp * r_perp) / 2.f;
}
static v3 sampleNormalGGX(v3 n, v3 wo, float alpha, v2 r) {           // utilities.h:351-377
    v3 t, b; localRefMatrix_Pixar(n, t, b);                           // local2world = [t b n]
    v3 wl(t.x * wo.x + t.y * wo.y + t.z * wo.z, b.x * wo.x + b.y * wo.y + b.z * wo.z, n.x * wo.x + n.y * wo.y + n.z * wo.z); // transpose(L)*wo
    v3 whL = normalize(v3(alpha, alpha, 1.f) * wl);
    if (whL.z < 0) whL = whL * -1.f;
    v3 T1 = (whL.z < 0.99999f) ? normalize(cross(v3(0.f, 0.f, 1.f), whL)) : v3(1.f, 0.f, 0.f);
    v3 T2 = cross(whL, T1);
    float radius = sqrtf(r.x), theta = TWO_PI_F * r.y;                // sampleUniformDisc utilities.h:296-301
    float st, ct; trig_sincos(theta, &st, &ct);
    float px = radius * ct, py = radius * st;
    float h = sqrtf(1 - px * px);
    float lx = (1 + whL.z) / 2.f;
    py = (1 - lx) * h + lx * py;                                      // Lerp(x,a,b) = (1-x)*a + x*b
    float pz = sqrtf(gmax(0.f, 1.f - (px * px + py * py)));
    v3 nh = px * T1 + py * T2 + pz * whL;
    v3 l(alpha * nh.x, alpha * nh.y, gmax(1e-6f, nh.z));
    v3 w(t.x * l.x + b.x * l.y + n.x * l.z, t.y * l.x + b.y * l.y + n.y * l.z, t.z * l.x + b.z * l.y + n.z * l.z);  // L * l
    return normalize(w);
}
static inline float SmithG2(float a2, float NoV, float NoL) {          // utilities.h:406-411
    float denom = NoL * sqrtf(NoV * NoV * (1.f - a2) + a2) + NoV * sqrtf(NoL * NoL * (1.f - a2) + a2);
    float nom = 2.f * NoV * NoL;
    return nom / denom;
}
static inline float SmithG1(float a2, float NoV) {                     // utilities.h:413-418
    float denom = sqrtf(NoV * NoV * (1 - a2) + a2) + NoV;
    float nom = 2 * NoV;
    return nom / denom;
}
static inline float normalDistribGGX(float cosTheta, float a2) {       // utilities.h:421-430
    if (cosTheta < 1e-6f) return 0.f;
    float nom = a2;
    float denom = cosTheta * cosTheta * (a2 - 1.f) + 1.f;
    denom = denom * denom * PI_F;
    return nom / denom;
}
static inline float powerHeuristic(float f, float g) { return (f * f) / (f * f + g * g); }  // utilities.h:432-436

struct ScatterRec { v3 bsdf; float pdf = 0; v3 dir; };

// the devTexSampler::linearSample(uv) calls of material.h: a bound texture, else the constant
static const OrcScene* g_tex_scene = nullptr;      // scene whose textures the samplers refer to (set by the entry points)
static thread_local v2 g_uv;                       // intersection.texCoords of the hit being shaded
static inline bool has_tex(int id) { return g_tex_scene && id >= 0 && id < (int)g_tex_scene->textures.size(); }
static inline float mat_rough(const Material& m) {
    float r = has_tex(m.roughnessMapID) ? linearSample(g_tex_scene->textures[m.roughnessMapID], g_uv).x : m.rough_s;
    return gclamp(r, (float)ROUGHNESS_MIN, ROUGHNESS_MAX);
}
static inline float mat_metal(const Material& m) {
    float r = has_tex(m.metallicMapID) ? linearSample(g_tex_scene->textures[m.metallicMapID], g_uv).x : m.metal_s;
    return gclamp(r, 0.f, 1.f);
}
static inline v3 mat_albedo_s(const Material& m) {
    return has_tex(m.albedoMapID) ? linearSample(g_tex_scene->textures[m.albedoMapID], g_uv) : v3(m.albedo_s[0], m.albedo_s[1], m.albedo_s[2]);
}

// material.h:246-261
static v3 microfacetBSDF(v3 n, v3 wo, v3 wi, v3 albedo, float rough) {
    float a2 = rough * rough;
    float cosO = dot(n, wo), cosI = dot(n, wi);
    if ((double)(cosO * cosI) < 1e-7) return v3();
    v3 wm = normalize(wo + wi);
    float D = normalDistribGGX(dot(wm, n), a2);
    float G2 = SmithG2(rough, cosO, cosI);
    v3 F = FresnelSchlick(albedo, dot(wo, wm));
    return F * D * G2 / gmax(4 * cosO * cosI, 1e-8f);
}
// material.h:263-271
static float microfacetPDF(v3 n, v3 wo, v3 wi, float rough) {
    float a2 = rough * rough;
    float cosO = dot(n, wo);
    v3 wm = normalize(wo + wi);
    float D = normalDistribGGX(dot(wm, n), a2);
    float G1 = SmithG1(rough, cosO);
    return G1 * D / gmax(4 * dot(wo, n), 1e-8f);
}
// material.h:296-310
static v3 metallicBSDF(v3 n, v3 wo, v3 wi, v3 albedo, float rough, float metal) {
    float a2 = rough * rough;
    float cosO = dot(n, wo), cosI = dot(n, wi);
    if ((double)(cosO * cosI) < 1e-7) return v3();
    v3 wm = normalize(wo + wi);
    float D = normalDistribGGX(dot(wm, n), a2);
    float G2 = SmithG2(rough, cosO, cosI);
    v3 F = FresnelSchlick(mixv(v3(0.08f, 0.08f, 0.08f), albedo, metal), dot(wo, wm));
    float spec = D * G2 / gmax(4 * cosO * cosI, 1e-8f);
    v3 diff = (1.f - metal) * albedo * 1.f / PI_F;                    // "* InvPI" expands to "* 1.f/PI"
    return mixvv(diff, v3(spec, spec, spec), F);
}
// material.h:312-320
static float metallicPDF(v3 n, v3 wo, v3 wi, float rough, float metal) {
    float a2 = rough * rough;
    float cosO = dot(n, wo);
    v3 wm = normalize(wo + wi);
    float D = normalDistribGGX(dot(wm, n), a2);
    float G1 = SmithG1(rough, cosO);
    return mixf(dot(wi, n) * 1.f / PI_F, G1 * D / gmax(4 * dot(wo, n), 1e-8f), 1.f / (2.f - metal));
}

// material.h:450-481 Material::scatterSample.  wo = incident ray direction (points INTO the surface).
static void scatterSample(const Material& m, const Isect& is, v3 wo, ScatterRec& srec, Rng& rng) {
    v3 n = is.n; g_uv = is.uv;
    switch (m.type) {
    case MAT_LAMBERTIAN: {                                             // material.h:56-62
        srec.bsdf = mat_albedo_s(m) / PI_F;
        srec.dir = sampleHemisphereCosine(n, sample2D(rng));
        srec.pdf = dot(srec.dir, n) / PI_F;
        break; }
    case MAT_METALLIC: {                                               // material.h:322-357
        float rough = mat_rough(m), metal = mat_metal(m);
        v3 albedo = mat_albedo_s(m);
        float r = sample1D(rng);
        if (r < 1.f / (2.f - metal)) {
            v3 wm = sampleNormalGGX(n, -1.f * wo, rough * rough, sample2D(rng));
            srec.dir = reflect(wo, wm);
        } else {
            srec.dir = sampleHemisphereCosine(n, sample2D(rng));
        }
        if (dot(-1.f * wo, n) < 0 || dot(srec.dir, n) < 0) { srec.bsdf = v3(); srec.pdf = 0; return; }
        srec.bsdf = metallicBSDF(n, -1.f * wo, srec.dir, albedo, rough, metal);
        srec.pdf = metallicPDF(n, -1.f * wo, srec.dir, rough, metal);
        break; }
    case MAT_DIELECTRIC: {                                             // material.h:81-114
        float ior1, ior2;
        if (dot(wo, n) < 0) { ior1 = 1; ior2 = m.ior; } else { ior2 = 1; ior1 = m.ior; }
        float F = FresnelMaxwell(fabsf(dot(wo, n)), ior1, ior2);
        if (sample1D(rng) < F) {
            srec.pdf = 1.f; srec.dir = getReflectDir(n, wo); srec.bsdf = mat_albedo_s(m);
        } else {
            srec.pdf = 1.f; srec.dir = getRefractDir(n, wo, ior1, ior2);
            srec.bsdf = mat_albedo_s(m) * (ior2 * ior2) / (ior1 * ior1);
        }
        srec.bsdf = srec.bsdf / fabsf(dot(srec.dir, n));
        break; }
    case MAT_MICROFACET: {                                             // material.h:273-294
        float rough = mat_rough(m);
        v3 albedo = mat_albedo_s(m);
        v3 wm = sampleNormalGGX(n, -1.f * wo, rough, sample2D(rng));
        srec.dir = reflect(wo, wm);
        if (dot(srec.dir, n) * dot(-1.f * wo, n) < 0) { srec.bsdf = v3(); srec.pdf = 0; return; }
        srec.bsdf = microfacetBSDF(n, -1.f * wo, srec.dir, albedo, rough);
        srec.pdf = microfacetPDF(n, -1.f * wo, srec.dir, rough);
        break; }
    case MAT_LIGHT:
        srec.bsdf = v3(m.albedo[0], m.albedo[1], m.albedo[2]); srec.pdf = 1.f; break;
    default: break;
    }
}
// material.h:483-501
static v3 matBSDF(const Material& m, const Isect& is, v3 wo, v3 wi) {
    v3 n = is.n; g_uv = is.uv;
    switch (m.type) {
    case MAT_LAMBERTIAN: return mat_albedo_s(m) / PI_F;
    case MAT_MICROFACET: return microfacetBSDF(n, -1.f * wo, wi, mat_albedo_s(m), mat_rough(m));
    case MAT_METALLIC: return metallicBSDF(n, -1.f * wo, wi, mat_albedo_s(m), mat_rough(m), mat_metal(m));
    default: return v3();
    }
}
// material.h:503-521
static float matPDF(const Material& m, const Isect& is, v3 wo, v3 wi) {
    v3 n = is.n; g_uv = is.uv;
    switch (m.type) {
    case MAT_LAMBERTIAN: return dot(wi, n) / PI_F;
    case MAT_MICROFACET: return microfacetPDF(n, -1.f * wo, wi, mat_rough(m));
    case MAT_METALLIC: return metallicPDF(n, -1.f * wo, wi, mat_rough(m), mat_metal(m));
    default: return 0;
    }
}

// ---------------------------------------------------------------------------------------
// light sampling (lightSample.h)
// ---------------------------------------------------------------------------------------
struct LightRec { v3 pos, emit; float pdf = 0; };   // pdf is uninitialised in the reference; 0 here (only read when lightSize==0)

// ---- opt-in "fixed" cube lights (NOT reference behaviour, off by default) -------------------------------
// The reference gives cube lights neither a sampled position nor a pdf (lightSample.h:141-248), which darkens MIS renders
// of scenes lit by cubes (cornell.txt).  With the fix on, a cube light is sampled uniformly over its surface — face chosen
// in proportion to its world-space area, uniform point on the face — with the same area -> solid-angle pdf as a triangle
// light, pdf = (1/L) d^2 / (A |cos|), A = the cube's total area, and lightPDF returns the same for a BSDF-sampled hit.
static int g_light_fix = 0;
// world-space areas of the three face pairs (x, y, z) of a unit cube under `transform`, and their sum times two
static inline void cubeFaceAreas(const Geom& g, float a[3], float& total) {
    v3 ex = mulMV(g.transform, v3(1.f, 0.f, 0.f), 0.f), ey = mulMV(g.transform, v3(0.f, 1.f, 0.f), 0.f), ez = mulMV(g.transform, v3(0.f, 0.f, 1.f), 0.f);
    a[0] = length(cross(ey, ez)); a[1] = length(cross(ez, ex)); a[2] = length(cross(ex, ey));
    total = 2.f * ((a[0] + a[1]) + a[2]);
}

// lightSample.h:141-169
static float lightPDF(const OrcScene& S, v3 viewPos, v3 lightPos, v3 normal, int triID, int geomID) {
    float pdf = -1.f;
    int L = (int)S.lights.size();
    if (triID >= 0) {
        const Triangle& tri = S.triangles[triID];
        float area = length(cross(tri.v[1] - tri.v[0], tri.v[2] - tri.v[0])) / 2.f;
        pdf = 1.f / L;
        v3 dlt = lightPos - viewPos;
        pdf = pdf * dot(dlt, dlt) / (area * fabsf(dot(normalize(viewPos - lightPos), normal)));
    }
    if (S.geoms[geomID].type == GEOM_SPHERE) {
        v3 viewPosL = mulMV(S.geoms[geomID].inverseTransform, viewPos, 1.f);
        float sinThetaMax2 = (0.5f * 0.5f) / dot(viewPosL, viewPosL);      // (viewPosL - 0) == viewPosL
        float cosThetaMax = sqrtf(std::max(0.0f, 1.0f - sinThetaMax2));
        pdf = 1.0f / (TWO_PI_F * (1 - cosThetaMax) * L);
    }
    if (g_light_fix && triID < 0 && S.geoms[geomID].type == GEOM_CUBE) {
        float a[3], area; cubeFaceAreas(S.geoms[geomID], a, area);
        pdf = 1.f / L;
        v3 dlt = lightPos - viewPos;
        pdf = pdf * dot(dlt, dlt) / (area * fabsf(dot(normalize(viewPos - lightPos), normal)));
    }
    return pdf;
}

// lightSample.h:171-248 (rng passed BY VALUE: the caller's stream does not advance)
static void lightSample(const OrcScene& S, v3 viewPos, Rng rng, LightRec& rec) {
    int L = (int)S.lights.size();
    if (L == 0) return;
    int lightID = (int)gmin(sample1D(rng) * L, L - 1.f);
    const LightPrim& light = S.lights[lightID];
    const Geom& geom = S.geoms[light.geomID];
    const Material& lightMat = S.materials[geom.materialid];
    v3 normal, lightPos; float pdf = 0;
    if (light.triangleID >= 0) {
        v2 r = sample2D(rng);
        float tt = sqrtf(r.x);                                              // sampleTriangleUniform utilities.h:218-222
        float u = 1.f - tt, v = tt * (1.f - r.y);
        const Triangle& tri = S.triangles[light.triangleID];
        lightPos = u * tri.v[0] + v * tri.v[1] + (1 - u - v) * tri.v[2];
        normal = normalize(u * tri.n[0] + v * tri.n[1] + (1 - u - v) * tri.n[2]);
        float area = length(cross(tri.v[1] - tri.v[0], tri.v[2] - tri.v[0])) / 2.f;
        pdf = 1.f / L;
        v3 dlt = lightPos - viewPos;
        pdf = pdf * dot(dlt, dlt) / (area * fabsf(dot(normalize(viewPos - lightPos), normal)));
    } else if (light.type == GEOM_SPHERE) {
        v2 xi = sample2D(rng);
        v3 viewPosL = mulMV(geom.inverseTransform, viewPos, 1.f);
        v3 center;                                                          // (0,0,0)
        v3 centerToRef = normalize(center - viewPosL);
        v3 tan, bit; localRefMatrix_Pixar(centerToRef, tan, bit);
        v3 dv = viewPosL - center;
        float sinThetaMax2 = (0.5f * 0.5f) / dot(dv, dv);
        float cosThetaMax = sqrtf(std::max(0.0f, 1.0f - sinThetaMax2));
        float cosTheta = (1.0f - xi.x) + xi.x * cosThetaMax;
        float sinTheta = sqrtf(std::max(0.f, 1.0f - cosTheta * cosTheta));
        float phi = xi.y * TWO_PI_F;
        float dc = length(center - viewPosL);                                // glm::distance(p0,p1) = length(p1-p0)
        float ds = dc * cosTheta - sqrtf(std::max(0.0f, 0.5f * 0.5f - dc * dc * sinTheta * sinTheta));
        float sinAlpha = ds * sinTheta / 0.5f;
        float cosAlpha = sqrtf(gmax(0.f, 1.f - sinAlpha * sinAlpha));
        float sp, cp; trig_sincos(phi, &sp, &cp);
        v3 nObj = sinAlpha * cp * tan + sinAlpha * sp * bit + cosAlpha * -centerToRef;
        v3 pObj = nObj * 0.5f;
        lightPos = mulMV(geom.transform, pObj, 1.0f);
        pdf = 1.0f / (TWO_PI_F * (1 - cosThetaMax) * L);
    }
    else if (g_light_fix && light.type == GEOM_CUBE) {
        float a[3], area; cubeFaceAreas(geom, a, area);
        float uf = sample1D(rng) * area;
        v2 r = sample2D(rng);
        int axis = 2; float sg = -1.f, acc = 0.f;                               // faces in the order +x -x +y -y +z -z
        bool done = false;
        for (int k = 0; k < 3 && !done; ++k) for (int s2 = 0; s2 < 2 && !done; ++s2) {
            acc = acc + a[k];
            if (uf < acc) { axis = k; sg = s2 == 0 ? 1.f : -1.f; done = true; }
        }
        v3 pObj, nObj;
        float c0 = r.x - 0.5f, c1 = r.y - 0.5f;
        if (axis == 0) { pObj = v3(0.5f * sg, c0, c1); nObj = v3(sg, 0.f, 0.f); }
        else if (axis == 1) { pObj = v3(c1, 0.5f * sg, c0); nObj = v3(0.f, sg, 0.f); }
        else { pObj = v3(c0, c1, 0.5f * sg); nObj = v3(0.f, 0.f, sg); }
        lightPos = mulMV(geom.transform, pObj, 1.0f);
        normal = normalize(mulMV(geom.invTranspose, nObj, 0.0f));
        pdf = 1.f / L;
        v3 dlt = lightPos - viewPos;
        pdf = pdf * dot(dlt, dlt) / (area * fabsf(dot(normalize(viewPos - lightPos), normal)));
    }
    // cube lights (reference behaviour): nothing assigned -> lightPos = 0, pdf = 0
    v3 rayDir = normalize(lightPos - viewPos);
    if (occlusionTest(S, viewPos + 1e-5f * rayDir, rayDir, lightPos)) { rec.emit = v3(); rec.pdf = -1.f; return; }
    rec.emit = v3(lightMat.albedo[0], lightMat.albedo[1], lightMat.albedo[2]);
    rec.pdf = pdf; rec.pos = lightPos;
}

// ---------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------
// Radiance address of a path.  Reference: img[pixelIndex].  The build's "samples per launch" extension (not a reference
// feature) carries the sample number in bits 24..30 of pixelIndex and gives every sample its own plane of g_plane_stride
// floats; with spp = 1 the high bits are zero and this is the reference's address.
static size_t g_plane_stride = 0;
static inline float* radiance_px(float* img, int pixelIndex) {
    return img + (size_t)(pixelIndex >> 24) * g_plane_stride + 3 * (size_t)(pixelIndex & 0xFFFFFF);
}
// pathtrace.cu:135-163
static void generate_ray(const Camera& cam, int iter, int traceDepth, int x, int y, PathSegment& seg) {
    seg.o = cam.position;
    seg.color = v3(1.f, 1.f, 1.f);
    Rng rng = make_rng(x, y, iter);                 // NOTE argument order (pathtrace.cu:148)
    v2 r = sample2D(rng);
    float fx = static_cast<float>(static_cast<float>(x) + ((double)r.x - 0.5) - cam.res[0] * 0.5f);
    float fy = static_cast<float>(static_cast<float>(y) + ((double)r.y - 0.5) - cam.res[1] * 0.5f);
    seg.d = normalize(cam.view - cam.right * cam.pixelLength[0] * fx - cam.up * cam.pixelLength[1] * fy);
    seg.pixelIndex = x + y * cam.res[0];
    seg.remainingBounces = traceDepth;
    seg.prevPdf = -1.f;
}

static inline void shade_continue(PathSegment& ps, const Isect& is, const ScatterRec& srec, int mtype, int& valid) {
    v3 offsetDir = dot(srec.dir, is.n) > 0 ? is.n : -is.n;
    ps.d = srec.dir;
    ps.o = is.p + (mtype == MAT_DIELECTRIC ? 1e-3f * offsetDir : 1e-4f * srec.dir);
    ps.color = ps.color * (srec.bsdf * fabsf(dot(srec.dir, is.n)) / srec.pdf);
    valid = 1;
}

// pathtrace.cu:405-498 PTkernel, one path
static int shade_bsdf(const OrcScene& S, int iter, int idx, int depth, const Isect& is, PathSegment& ps, float* img) {
    int valid = 0;
    const Material& m = S.materials[is.materialId];
    if (is.t <= 0.f) { ps.color = ps.color * 0.f; ps.remainingBounces = 0; return 0; }
    Rng rng = make_rng(iter, idx + dbg_shift(depth) + g_slot_bias, depth);
    ScatterRec srec;
    scatterSample(m, is, ps.d, srec, rng);
    if (srec.pdf == 0) { ps.color = ps.color * 0.f; ps.remainingBounces = 0; return 0; }
    if (m.type == MAT_LIGHT) {
        ps.color = ps.color * (srec.bsdf / srec.pdf);
        ps.remainingBounces = 0;
        v3 c = processNAN(ps.color);
        float* px = radiance_px(img, ps.pixelIndex); px[0] += c.x; px[1] += c.y; px[2] += c.z;
        return 0;
    }
    shade_continue(ps, is, srec, m.type, valid);
    if (--ps.remainingBounces == 0) { ps.color = v3(); valid = 0; }
    return valid;
}

// pathtrace.cu:500-588 MisPTkernel, one path
static int shade_mis(const OrcScene& S, int iter, int idx, int depth, const Isect& is, PathSegment& ps, float* img) {
    int valid = 0;
    const Material& m = S.materials[is.materialId];
    if (is.t <= 0.f) { ps.color = ps.color * 0.f; ps.remainingBounces = 0; return 0; }
    bool isDelta = (m.type == MAT_DIELECTRIC);
    Rng rng = make_rng(iter, idx + dbg_shift(depth) + g_slot_bias, depth);
    ScatterRec srec;
    scatterSample(m, is, ps.d, srec, rng);
    v3 viewPos = is.p, viewNor = is.n;
    float* px = radiance_px(img, ps.pixelIndex);
    if (srec.pdf == 0) { ps.color = ps.color * 0.f; ps.remainingBounces = 0; return 0; }
    if (m.type == MAT_LIGHT) {
        float weight = 1.f;
        if (ps.prevPdf > 0) {
            float lightPdf = lightPDF(S, ps.o, viewPos, viewNor, is.triangleID, is.geomID);
            weight = powerHeuristic(ps.prevPdf, lightPdf);
        }
        ps.color = ps.color * ((srec.bsdf / srec.pdf) * weight);
        ps.remainingBounces = 0;
        v3 c = processNAN(ps.color);
        if ((g_dbg_mis_terms & 1) && depth >= g_dbg_min_depth) { px[0] += c.x; px[1] += c.y; px[2] += c.z; }
        return 0;
    }
    if (!isDelta) {
        LightRec Li;
        lightSample(S, viewPos, rng, Li);
        v3 Liwi = normalize(Li.pos - viewPos);
        float bsdfPdf = matPDF(m, is, ps.d, Liwi);
        float lightPdf = Li.pdf;
        v3 Libsdf = matBSDF(m, is, ps.d, Liwi);
        float weight = powerHeuristic(lightPdf, bsdfPdf);
        v3 c = processNAN(weight * ps.color * Li.emit * Libsdf * gmax(dot(Liwi, is.n), 0.f) / lightPdf);
        if ((g_dbg_mis_terms & 2) && depth >= g_dbg_min_depth) { px[0] += c.x; px[1] += c.y; px[2] += c.z; }
    }
    shade_continue(ps, is, srec, m.type, valid);
    ps.prevPdf = isDelta ? -1.f : srec.pdf;
    if (--ps.remainingBounces == 0) { ps.color = v3(); valid = 0; }
    return valid;
}

// pathtrace.cu:348-403 DirectLiPTkernel, one path (always rayValid = 0)
static int shade_direct(const OrcScene& S, int iter, int idx, int depth, const Isect& is, PathSegment& ps, float* img) {
    Rng rng = make_rng(iter, idx + dbg_shift(depth) + g_slot_bias, depth);
    const Material& m = S.materials[is.materialId];
    float* px = radiance_px(img, ps.pixelIndex);
    if (is.t <= 0.f) { ps.color = ps.color * 0.f; ps.remainingBounces = 0; return 0; }
    if (m.type == MAT_LIGHT) {
        ps.color = ps.color * v3(m.albedo[0], m.albedo[1], m.albedo[2]);
        ps.remainingBounces = 0;
        v3 c = processNAN(ps.color); px[0] += c.x; px[1] += c.y; px[2] += c.z;
        return 0;
    }
    LightRec Li;
    lightSample(S, is.p, rng, Li);
    if (Li.pdf <= 0) { ps.color = ps.color * 0.f; return 0; }
    v3 wi = normalize(Li.pos - is.p);
    v3 bsdf = matBSDF(m, is, ps.d, wi);
    ps.color = ps.color * (bsdf * Li.emit * gmax(dot(wi, is.n), 0.f) / Li.pdf);
    v3 c = processNAN(ps.color); px[0] += c.x; px[1] += c.y; px[2] += c.z;
    return 0;
}

// ---------------------------------------------------------------------------------------
// host: parser (scene.cpp:47-337), OBJ (scene.cpp:340-440), setDevData (479-567), BVH (BVH.cpp)
// ---------------------------------------------------------------------------------------
static std::vector<std::string> tokenize(const std::string& s) {
    std::stringstream ss(s);
    std::istream_iterator<std::string> it(ss), end;
    return std::vector<std::string>(it, end);
}
static bool safeGetline(std::istream& is, std::string& t) {   // utilities.cpp:81-112: handles \n, \r\n, \r, EOF
    t.clear();
    std::streambuf* sb = is.rdbuf();
    for (;;) {
        int c = sb->sbumpc();
        switch (c) {
        case '\n': return true;
        case '\r': if (sb->sgetc() == '\n') sb->sbumpc(); return true;
        case EOF: if (t.empty()) is.setstate(std::ios::eofbit); return true;
        default: t += (char)c;
        }
    }
}
static bool file_exists(const std::string& p) { std::ifstream f(p.c_str()); return f.good(); }
static std::string resolve_path(const OrcScene& S, std::string p) {
    std::replace(p.begin(), p.end(), '\\', '/');
    if (file_exists(p)) return p;                       // reference: relative to the process CWD
    std::string q = S.sceneDir + "/" + p;               // fallback: relative to the scene file's directory
    return q;
}

static m4 buildTransformationMatrix(v3 tr, v3 rot, v3 sc) {   // utilities.cpp:65-72
    m4 T = m4_translate(m4_identity(), tr);
    m4 R = m4_rotate(m4_identity(), rot.x * (float)PI_F / 180, v3(1, 0, 0));
    R = m4_mul(R, m4_rotate(m4_identity(), rot.y * (float)PI_F / 180, v3(0, 1, 0)));
    R = m4_mul(R, m4_rotate(m4_identity(), rot.z * (float)PI_F / 180, v3(0, 0, 1)));
    m4 Sc = m4_scale(m4_identity(), sc);
    return m4_mul(m4_mul(T, R), Sc);
}

// minimal OBJ reader standing in for tinyobjloader 2.0.0 LoadObj(triangulate=true) (scene.cpp:340-440):
// v / vn / vt / f with fan triangulation, negative indices supported; everything else ignored.
static bool loadObj(OrcScene& S, const std::string& file, int geomIdx, std::vector<Triangle>& out) {
    std::ifstream in(file.c_str());
    if (!in.good()) { S.error = "cannot open obj " + file; return false; }
    std::vector<v3> V, N; std::vector<v2> T;
    std::string line;
    struct Idx { int v, t, n; };
    while (std::getline(in, line)) {
        const char* p = line.c_str();
        while (*p == ' ' || *p == '\t') ++p;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) { char* e; v3 a; a.x = (float)strtod(p + 2, &e); a.y = (float)strtod(e, &e); a.z = (float)strtod(e, &e); V.push_back(a); }
        else if (p[0] == 'v' && p[1] == 'n') { char* e; v3 a; a.x = (float)strtod(p + 3, &e); a.y = (float)strtod(e, &e); a.z = (float)strtod(e, &e); N.push_back(a); }
        else if (p[0] == 'v' && p[1] == 't') { char* e; v2 a; a.x = (float)strtod(p + 3, &e); a.y = (float)strtod(e, &e); T.push_back(a); }
        else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            std::vector<Idx> f; const char* s = p + 2;
            for (;;) {
                while (*s == ' ' || *s == '\t' || *s == '\r') ++s;
                if (!*s) break;
                Idx id = {0, 0, 0}; char* e;
                id.v = (int)strtol(s, &e, 10); s = e;
                if (*s == '/') { ++s; if (*s != '/') { id.t = (int)strtol(s, &e, 10); s = e; } if (*s == '/') { ++s; id.n = (int)strtol(s, &e, 10); s = e; } }
                id.v = id.v > 0 ? id.v - 1 : (int)V.size() + id.v;
                id.t = id.t > 0 ? id.t - 1 : (id.t < 0 ? (int)T.size() + id.t : -1);
                id.n = id.n > 0 ? id.n - 1 : (id.n < 0 ? (int)N.size() + id.n : -1);
                f.push_back(id);
            }
            bool hasTex = !T.empty(), hasNor = !N.empty();
            for (int i = 0; i + 2 < (int)f.size(); ++i) {
                Idx I[3] = {f[0], f[i + 1], f[i + 2]};
                Triangle tri;
                for (int k = 0; k < 3; ++k) tri.v[k] = V[I[k].v];
                if (hasNor) { for (int k = 0; k < 3; ++k) tri.n[k] = N[I[k].n]; }
                else { v3 fn = normalize(cross(tri.v[1] - tri.v[0], tri.v[2] - tri.v[0])); tri.n[0] = tri.n[1] = tri.n[2] = fn; }
                if (hasTex) for (int k = 0; k < 3; ++k) tri.tex[k] = T[I[k].t];
                tri.geomIdx = geomIdx;
                out.push_back(tri);
            }
        }
    }
    // NOTE hasTex/hasNor are file-global in the reference (attrib.normals.empty()); all vn/vt precede f in the shipped models.
    return true;
}

// Scene::loadTexture, scene.cpp:318-337: returns the texture id when the token names an existing non-empty file (pixel
// data comes later from the harness), -1 otherwise (the caller then falls back to numbers, scene.cpp:276-297).
static int loadTextureRef(OrcScene& S, const std::string& token) {
    std::string path = resolve_path(S, token);
    std::ifstream f(path.c_str(), std::ios::binary);
    if (!f.good() || f.peek() == EOF) return -1;
    for (size_t i = 0; i < S.textures.size(); ++i) if (S.textures[i].path == path) return (int)i;
    Texture t; t.path = path;
    S.textures.push_back(t);
    return (int)S.textures.size() - 1;
}

static int loadMaterial(OrcScene& S, std::istream& in, const std::string& name) {   // scene.cpp:243-316
    if (S.materialNameMap.count(name)) return -1;
    Material m; memset(&m, 0, sizeof m);
    m.type = 0;                                 // uninitialised in the reference; 0 (Lambertian) in practice
    m.albedo[0] = m.albedo[1] = m.albedo[2] = 1.f; m.ior = 1.5f;
    m.albedoMapID = m.metallicMapID = m.roughnessMapID = m.normalMapID = -1;
    for (int i = 0; i < 6; i++) {
        std::string line; safeGetline(in, line);
        std::vector<std::string> tk = tokenize(line);
        if (tk.size() <= 0) break;
        if (tk[0] == "TYPE") {
            static const char* names[5] = {"Lambertian", "MetallicWorkflow", "Dielectric", "Microfacet", "Light"};
            for (int k = 0; k < 5; ++k) if (tk.size() > 1 && tk[1] == names[k]) m.type = k;
        } else if (tk[0] == "ALBEDO" || tk[0] == "METALLIC" || tk[0] == "ROUGHNESS" || tk[0] == "NORMAL" || tk[0] == "IOR") {
            // The reference indexes tokens[1] (and [2], [3] for ALBEDO) unconditionally (scene.cpp:268-309): out of bounds on
            // short lines, e.g. `ALBEDO <missing file>` in camera.txt / normalMap.txt / test.txt / texture_test.txt whose
            // textures are missing blobs.  Guarded like the product: a keyword without a value is skipped, an ALBEDO that is
            // neither an existing file nor three numbers is an error.
            if (tk.size() < 2) continue;
            if (tk[0] == "ALBEDO") {
                m.albedoMapID = loadTextureRef(S, tk[1]);
                if (m.albedoMapID < 0) {
                    if (tk.size() < 4) { S.error = "ALBEDO needs three numbers: " + line; return -2; }
                    for (int k = 0; k < 3; ++k) { m.albedo[k] = (float)atof(tk[1 + k].c_str()); m.albedo_s[k] = m.albedo[k]; }
                }
            } else if (tk[0] == "METALLIC") {
                m.metallicMapID = loadTextureRef(S, tk[1]);
                if (m.metallicMapID < 0) m.metal_s = (float)atof(tk[1].c_str());
            } else if (tk[0] == "ROUGHNESS") {
                m.roughnessMapID = loadTextureRef(S, tk[1]);
                if (m.roughnessMapID < 0) { double r = atof(tk[1].c_str()); m.rough_s = (float)(r > ROUGHNESS_MIN ? r : ROUGHNESS_MIN); }   // glm::max(double,double)
            } else if (tk[0] == "NORMAL") {
                m.normalMapID = loadTextureRef(S, tk[1]);
            } else {
                m.ior = (float)atof(tk[1].c_str());
            }
        }
    }
    S.materialNameMap[name] = (int)S.materials.size();
    S.materials.push_back(m);
    return 1;
}

static int loadGeom(OrcScene& S, std::istream& in, const std::string& name) {       // scene.cpp:83-163
    if (S.geomNameMap.count(name)) return -1;
    Geom g; memset(&g, 0, sizeof g);
    std::string line, mesh;
    v3 tr, rot, sc;
    g.type = 0;                                   // uninitialised in the reference if the type line is unrecognised
    safeGetline(in, line);
    if (!line.empty() && in.good()) {
        if (line == "sphere") g.type = GEOM_SPHERE;
        else if (line == "cube") g.type = GEOM_CUBE;
        else if (line.find(".obj") != line.npos) { g.type = GEOM_OBJ; mesh = resolve_path(S, line); }
    }
    safeGetline(in, line);
    if (!line.empty() && in.good()) {
        std::vector<std::string> tk = tokenize(line);
        // the reference indexes tokens[1] unconditionally (scene.cpp:124-131): out of bounds on a one-token line such as the
        // obsolete-format files' (disney_test.txt).  Guarded here; the id then keeps its initial value.
        if (tk.size() > 1) {
            if (S.materialNameMap.count(tk[1])) g.materialid = S.materialNameMap[tk[1]];
            else g.materialid = atoi(tk[1].c_str());
        }
    }
    // the reference never checks the id (a later dev_materials[id] read is then out of bounds); rejected like the product does
    if (g.materialid < 0 || g.materialid >= (int)S.materials.size()) { S.error = "object " + name + " refers to an unknown material"; return -2; }
    safeGetline(in, line);
    while (!line.empty() && in.good()) {
        std::vector<std::string> tk = tokenize(line);
        if (tk.size() >= 4) {                     // same guard: the reference reads tokens[1..3] unconditionally
            if (tk[0] == "TRANS") tr = v3((float)atof(tk[1].c_str()), (float)atof(tk[2].c_str()), (float)atof(tk[3].c_str()));
            else if (tk[0] == "ROTAT") rot = v3((float)atof(tk[1].c_str()), (float)atof(tk[2].c_str()), (float)atof(tk[3].c_str()));
            else if (tk[0] == "SCALE") sc = v3((float)atof(tk[1].c_str()), (float)atof(tk[2].c_str()), (float)atof(tk[3].c_str()));
        }
        safeGetline(in, line);
    }
    g.transform = buildTransformationMatrix(tr, rot, sc);
    g.inverseTransform = m4_inverse(g.transform);
    g.invTranspose = m4_inverse_transpose(g.transform);
    if (g.type == GEOM_OBJ && !S.meshPool.count(mesh)) {
        std::vector<Triangle> tris;
        if (!loadObj(S, mesh, (int)S.geoms.size(), tris)) return -2;
        S.meshPool[mesh] = tris;
    }
    S.geomNameMap[name] = (int)S.geoms.size();
    S.geoms.push_back(g); S.geomScale.push_back(sc); S.geomMeshFile.push_back(mesh);
    return 1;
}

static int loadCamera(OrcScene& S, std::istream& in) {                              // scene.cpp:165-236
    Camera& cam = S.cam;
    float fovy = 0;
    for (int i = 0; i < 5; i++) {
        std::string line; safeGetline(in, line);
        std::vector<std::string> tk = tokenize(line);
        if (tk.empty()) continue;
        if (tk[0] == "RES") { cam.res[0] = atoi(tk[1].c_str()); cam.res[1] = atoi(tk[2].c_str()); }
        else if (tk[0] == "FOVY") fovy = (float)atof(tk[1].c_str());
        else if (tk[0] == "ITERATIONS") S.iterations = atoi(tk[1].c_str());
        else if (tk[0] == "DEPTH") S.traceDepth = atoi(tk[1].c_str());
        else if (tk[0] == "FILE") S.imageName = tk[1];
    }
    // harness overrides (RES/DEPTH overrides are harness flags; equivalent to editing the RES/DEPTH lines)
    if (S.ovr_w > 0) { cam.res[0] = S.ovr_w; cam.res[1] = S.ovr_h; }
    if (S.ovr_depth > 0) S.traceDepth = S.ovr_depth;
    std::string line; safeGetline(in, line);
    while (!line.empty() && in.good()) {
        std::vector<std::string> tk = tokenize(line);
        if (tk[0] == "EYE") cam.position = v3((float)atof(tk[1].c_str()), (float)atof(tk[2].c_str()), (float)atof(tk[3].c_str()));
        if (tk[0] == "ROTAT") { S.theta = gclamp((float)atof(tk[1].c_str()), -89.f, 89.f); S.phi = (float)atof(tk[2].c_str()); S.posInit = false; }
        else if (tk[0] == "LOOKAT") { cam.lookAt = v3((float)atof(tk[1].c_str()), (float)atof(tk[2].c_str()), (float)atof(tk[3].c_str())); S.posInit = true; }
        else if (tk[0] == "UP") cam.up = v3((float)atof(tk[1].c_str()), (float)atof(tk[2].c_str()), (float)atof(tk[3].c_str()));
        safeGetline(in, line);
    }
    float yscaled = tanf(fovy * (PI_F / 180));
    float xscaled = (yscaled * cam.res[0]) / cam.res[1];
    float fovx = (atanf(xscaled) * 180) / PI_F;
    cam.fov[0] = fovx; cam.fov[1] = fovy;
    cam.view = normalize(cam.lookAt - cam.position);
    cam.right = normalize(cross(cam.view, cam.up));
    cam.pixelLength[0] = 2 * xscaled / (float)cam.res[0];
    cam.pixelLength[1] = 2 * yscaled / (float)cam.res[1];
    return 1;
}

// main.cpp:102-127 (startup) + main.cpp:173-193 (runCuda, camchanged)
static void camera_fixups(OrcScene& S) {
    Camera& cam = S.cam;
    v3 view = cam.view;
    if (S.posInit) {
        S.phi = atan2f(view.z, view.x) * 57.295779513082320876798154814105f;       // glm::degrees
        S.theta = sinf(view.y) * 57.295779513082320876798154814105f;               // [sic] sin, not asin
        S.theta = gclamp(S.theta, -89.f, 89.f);
    } else {
        float rt = S.theta * 0.01745329251994329576923690768489f, rp = S.phi * 0.01745329251994329576923690768489f;
        cam.view = v3(cosf(rt) * cosf(rp), sinf(rt), cosf(rt) * sinf(rp));
        cam.lookAt = cam.position + cam.view;
    }
    cam.right = cross(cam.view, cam.up);
    cam.up = cross(cam.right, cam.view);
    // runCuda(), camchanged == true
    float rt = S.theta * 0.01745329251994329576923690768489f, rp = S.phi * 0.01745329251994329576923690768489f;
    cam.view = v3(cosf(rt) * cosf(rp), sinf(rt), cosf(rt) * sinf(rp));
    v3 v = cam.view, u = v3(0, 1, 0), r = cross(v, u);
    cam.up = normalize(cross(r, v));
    cam.right = normalize(r);
}

// ---- BVH (BVH.cpp) ----
struct Bounds3 {
    v3 pMin, pMax;
    Bounds3() { pMin = v3(INFINITY, INFINITY, INFINITY); pMax = v3(-INFINITY, -INFINITY, -INFINITY); } // vec3(DBL_MAX) -> +inf as float
    Bounds3(v3 a, v3 b) { pMin = vmin(a, b); pMax = vmax(a, b); }
    void Union(const Bounds3& b) { pMin = vmin(pMin, b.pMin); pMax = vmax(pMax, b.pMax); }
    v3 Diagonal() const { return pMax - pMin; }
    v3 Centroid() const { return (pMax + pMin) / 2.0f; }
    int MaxExtent() const { v3 d = Diagonal(); if (d.x > d.y && d.x > d.z) return 0; else if (d.y > d.z) return 1; else return 2; }
    bool isValid() const { if (pMin.x > pMax.x || pMin.y > pMin.y || pMin.z > pMin.z) return false; return true; }   // [sic] Bounds3.hpp:71
    double SurfaceArea() const {
        v3 d = Diagonal();
        double area = 2 * ((double)d.x * (double)d.y + (double)d.y * (double)d.z + (double)d.z * (double)d.x);
        if (!isValid()) return 0;
        return area;
    }
};
static inline Bounds3 triBound(const Triangle& t) {     // Union(Bounds3(v0,v1), v2)  Bounds3.hpp:231-238, 293
    Bounds3 b(t.v[0], t.v[1]); Bounds3 r; r.pMin = vmin(b.pMin, t.v[2]); r.pMax = vmax(b.pMax, t.v[2]); return r;
}
static inline v3 triCentroid(const Triangle& t) { return (t.v[0] + t.v[1] + t.v[2]) / 3.f; }

struct BuildNode { int start, end, left = -1, right = -1, parent = -1; Bounds3 box; };

// BVH.cpp:13-92 recursiveBuildSAH + BVH.cpp:121-147 preorder flatten, fused (preorder index == creation order)
static int buildSAH(std::vector<Triangle>& t, int start, int end, std::vector<BuildNode>& nodes, int parent) {
    int me = (int)nodes.size();
    nodes.push_back(BuildNode());
    nodes[me].start = start; nodes[me].end = end; nodes[me].parent = parent;
    Bounds3 box; v3 cMin(FLT_MAX, FLT_MAX, FLT_MAX), cMax(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (int i = start; i < end; ++i) { box.Union(triBound(t[i])); cMin = vmin(cMin, triCentroid(t[i])); cMax = vmax(cMax, triCentroid(t[i])); }
    nodes[me].box = box;
    Bounds3 centerBox(cMin, cMax);
    int axis = centerBox.MaxExtent();
    if (end - start <= std::max(MAX_PRIM, 1)) return me;
    struct Bucket { int num = 0; Bounds3 box; };
    std::vector<Bucket> buckets(BUCKET_NUM);
    float Loss = FLT_MAX;
    float boxDia = centerBox.pMax[axis] - centerBox.pMin[axis];
    auto bucketOf = [&](const Triangle& tri) {
        float offset = gclamp((triCentroid(tri)[axis] - centerBox.pMin[axis]) / boxDia, 0.f, 1.f);
        return offset == 1.f ? BUCKET_NUM - 1 : static_cast<int>(offset * BUCKET_NUM);
    };
    for (int i = start; i < end; ++i) { int b = bucketOf(t[i]); buckets[b].num += 1; buckets[b].box.Union(triBound(t[i])); }
    int partitionBucket = 0;
    for (int i = 0; i < BUCKET_NUM - 1; ++i) {
        Bounds3 L, R; int nL = 0, nR = 0;
        for (int j = 0; j <= i; ++j) { L.Union(buckets[j].box); nL += buckets[j].num; }
        for (int j = i + 1; j < BUCKET_NUM; ++j) { R.Union(buckets[j].box); nR += buckets[j].num; }
        float tempLoss = (float)((nL * L.SurfaceArea() + nR * R.SurfaceArea()) / box.SurfaceArea());
        if (tempLoss < Loss && nL != 0 && nR != 0) { Loss = tempLoss; partitionBucket = i; }
    }
    // std::partition (libstdc++): order inside each side is implementation-defined; the pins are to libstdc++
    std::vector<Triangle>::iterator bound = std::partition(t.begin() + start, t.begin() + end,
        [&](const Triangle& tri) { return bucketOf(tri) <= partitionBucket; });
    int mid = (int)(bound - t.begin());
    if (mid == start || mid == end) mid = (start + end) / 2;   // GUARD (deviation): the reference recurses forever on degenerate input
    int l = buildSAH(t, start, mid, nodes, me);
    int r = buildSAH(t, mid, end, nodes, me);
    nodes[me].left = l; nodes[me].right = r;
    return me;
}

// BVH.cpp:180-239 buildGpuMTBVH
static void buildMTBVH(const std::vector<BuildNode>& info, std::vector<BvhNode>& out) {
    int n = (int)info.size();
    out.resize(6 * (size_t)n);
    for (int dir = 0; dir < 6; ++dir) {
        int off = dir * n, axis = dir % 3, sign = (dir < 3 ? 1 : -1);
        for (int i = 0; i < n; ++i) {
            BvhNode& o = out[off + i];
            o.pMin = info[i].box.pMin; o.pMax = info[i].box.pMax;
            o.start = info[i].start; o.end = info[i].end;
            int left = info[i].left, right = info[i].right, parent = info[i].parent;
            if (left != -1) {
                if (info[left].box.Centroid()[axis] * sign > info[right].box.Centroid()[axis] * sign) std::swap(left, right);
                o.hit = left;
            } else {
                if (parent == -1) o.hit = -1;
                else if (i == out[off + parent].hit) o.hit = (i == info[parent].left) ? info[parent].right : info[parent].left;
                else o.hit = out[off + parent].miss;
            }
            if (i == 0) o.miss = -1;
            else if (i == out[off + parent].hit) o.miss = (i == info[parent].left) ? info[parent].right : info[parent].left;
            else o.miss = out[off + parent].miss;
        }
    }
}

// scene.cpp:479-567 setDevData (host part)
static void setDevData(OrcScene& S) {
    for (int gi = 0; gi < (int)S.geoms.size(); ++gi) {
        const Geom& g = S.geoms[gi];
        if (g.type != GEOM_OBJ) continue;
        const std::vector<Triangle>& mesh = S.meshPool[S.geomMeshFile[gi]];
        for (const Triangle& t : mesh) {
            Triangle w;
            for (int i = 0; i < 3; ++i) {
                w.v[i] = mulMV(g.transform, t.v[i], 1.0f);
                w.n[i] = normalize(mulMV(g.invTranspose, t.n[i], 0.0f));
                w.tex[i] = t.tex[i];
                // tangent/bitangent (scene.cpp:493-507): evaluated inside the vertex loop with partially filled data; the last pass (i==2) wins
                w.tangent = v3(); w.bitangent = v3();
                v3 e1 = w.v[1] - w.v[0], e2 = w.v[2] - w.v[0];
                float du1 = w.tex[1].x - w.tex[0].x, dv1 = w.tex[1].y - w.tex[0].y;
                float du2 = w.tex[2].x - w.tex[0].x, dv2 = w.tex[2].y - w.tex[0].y;
                float f = (du1 * dv2 - du2 * dv1);
                if ((double)fabsf(f) < 1e-8) continue;
                w.tangent = normalize((dv2 * e1 - dv1 * e2) / f);
                w.bitangent = normalize((-du2 * e1 + du1 * e2) / f);
            }
            w.geomIdx = gi;
            S.triangles.push_back(w);
        }
    }
    std::vector<BuildNode> nodes;
    buildSAH(S.triangles, 0, (int)S.triangles.size(), nodes, -1);
    S.bvh_size = (int)nodes.size();
    buildMTBVH(nodes, S.bvh);
    for (int i = 0; i < (int)S.geoms.size(); ++i)
        if (S.materials[S.geoms[i].materialid].type == MAT_LIGHT && S.geoms[i].type != GEOM_OBJ)
            S.lights.push_back(LightPrim{i, -1, S.geoms[i].type});
    for (int i = 0; i < (int)S.triangles.size(); ++i) {
        int gid = S.triangles[i].geomIdx;
        if (S.materials[S.geoms[gid].materialid].type == MAT_LIGHT) S.lights.push_back(LightPrim{gid, i, GEOM_OBJ});
    }
}

// ---------------------------------------------------------------------------------------
// C API
// ---------------------------------------------------------------------------------------
extern "C" {

void orc_set_trig_mode(int m) { g_trig_mode = m; }
void orc_set_debug_slot_shift(int k) { g_dbg_slot_shift = k; }
// diagnostic (tools/diag/mis20_residue.py): which of MisPTkernel's two radiance terms reach the image — 1 the BSDF-sampled light hits
// (pathtrace.cu:545-556), 2 the next-event samples (:558-571), 3 both (default); the paths themselves are unchanged
void orc_set_debug_mis_terms(int mask) { g_dbg_mis_terms = mask; }
// diagnostics of the same script: a compacted-slot offset of its own for depth >= 1 (on = 0: the common offset applies at every depth), and
// a lower bound on the depth whose MIS contributions reach the image
void orc_set_debug_slot_shift_deep(int on, int k) { g_dbg_deep_on = on; g_dbg_slot_shift_deep = k; }
void orc_set_debug_min_depth(int d) { g_dbg_min_depth = d; }
void orc_set_sample2d_order(int xfirst) { g_s2d_xfirst = xfirst ? 1 : 0; }
void orc_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
void orc_set_light_fix(int on) { g_light_fix = on ? 1 : 0; }
int orc_get_trig_mode() { return g_trig_mode; }

// Scene(file) + main.cpp camera fix-ups + setDevData.  res_w/res_h/depth <= 0 keep the file's values.
OrcScene* orc_scene_load(const char* file, int res_w, int res_h, int depth) {
    OrcScene* S = new OrcScene();
    S->ovr_w = res_w; S->ovr_h = res_h; S->ovr_depth = depth;
    std::string f(file);
    size_t sl = f.find_last_of('/');
    S->sceneDir = sl == std::string::npos ? "." : f.substr(0, sl);
    std::ifstream in(file);
    if (!in.is_open()) { S->error = "cannot open scene file"; return S; }
    S->cam = Camera();
    while (in.good()) {
        std::string line; safeGetline(in, line);
        if (line.empty()) continue;
        std::vector<std::string> tk = tokenize(line);
        if (tk.empty()) continue;
        int rc = 1;
        if (tk[0] == "MATERIAL" && tk.size() > 1) rc = loadMaterial(*S, in, tk[1]);       // tokens[1] is read unconditionally in the reference
        else if (tk[0] == "OBJECT" && tk.size() > 1) rc = loadGeom(*S, in, tk[1]);
        else if (tk[0] == "CAMERA") rc = loadCamera(*S, in);
        else if (tk[0] == "ENV" && tk.size() > 1) S->envMapID = loadTextureRef(*S, tk[1]);
        if (rc == -2) return S;
    }
    camera_fixups(*S);
    setDevData(*S);
    return S;
}
void orc_scene_free(OrcScene* S) { if (g_tex_scene == S) g_tex_scene = nullptr; delete S; }     // never leave the samplers a dangling scene
const char* orc_scene_error(OrcScene* S) { return S->error.c_str(); }

// counts: [0]=geoms [1]=materials [2]=triangles [3]=bvh_size [4]=lights [5]=traceDepth [6]=iterations [7]=res.x [8]=res.y
//         [9]=textures [10]=envMapID
void orc_scene_counts(OrcScene* S, int* out) {
    out[9] = (int)S->textures.size(); out[10] = S->envMapID;
    out[0] = (int)S->geoms.size(); out[1] = (int)S->materials.size(); out[2] = (int)S->triangles.size();
    out[3] = S->bvh_size; out[4] = (int)S->lights.size(); out[5] = S->traceDepth; out[6] = S->iterations;
    out[7] = S->cam.res[0]; out[8] = S->cam.res[1];
}
const char* orc_scene_texture_path(OrcScene* S, int i) { return (i >= 0 && i < (int)S->textures.size()) ? S->textures[i].path.c_str() : ""; }
// float RGB texels, row 0 = bottom row (what stbi_loadf(..., 3) returns after stbi_set_flip_vertically_on_load(true))
void orc_scene_set_texture(OrcScene* S, int i, int w, int h, const float* rgb) {
    Texture& t = S->textures[i]; t.w = w; t.h = h; t.rgb.assign(rgb, rgb + 3 * (size_t)w * h);
}
void orc_scene_get_geoms(OrcScene* S, void* out) { memcpy(out, S->geoms.data(), S->geoms.size() * sizeof(Geom)); }
void orc_scene_get_materials(OrcScene* S, void* out) { memcpy(out, S->materials.data(), S->materials.size() * sizeof(Material)); }
void orc_scene_get_triangles(OrcScene* S, void* out) { memcpy(out, S->triangles.data(), S->triangles.size() * sizeof(Triangle)); }
void orc_scene_get_bvh(OrcScene* S, void* out) { memcpy(out, S->bvh.data(), S->bvh.size() * sizeof(BvhNode)); }
void orc_scene_get_lights(OrcScene* S, void* out) { memcpy(out, S->lights.data(), S->lights.size() * sizeof(LightPrim)); }
void orc_scene_get_camera(OrcScene* S, void* out) { memcpy(out, &S->cam, sizeof(Camera)); }
int orc_sizeof(int what) {
    switch (what) { case 0: return sizeof(Geom); case 1: return sizeof(Material); case 2: return sizeof(Triangle);
    case 3: return sizeof(BvhNode); case 4: return sizeof(LightPrim); case 5: return sizeof(Camera);
    case 6: return sizeof(PathSegment); case 7: return sizeof(Isect); } return -1;
}

// --- primitive-level probes (for the SURVEY pins and per-function parity) ---
uint32_t orc_utilhash(uint32_t a) { return utilhash(a); }
uint32_t orc_seed_hash(int iter, int index, int depth) { return seed_hash(iter, index, depth); }
void orc_rng_uniforms(int iter, int index, int depth, int n, float* out) { Rng r = make_rng(iter, index, depth); for (int i = 0; i < n; ++i) out[i] = sample1D(r); }
void orc_rng_uniforms_seed(uint32_t seed, int n, float* out) { Rng r; r.x = seed % 2147483647u; if (r.x == 0) r.x = 1; for (int i = 0; i < n; ++i) out[i] = sample1D(r); }
void orc_sincos(float x, float* s, float* c) { trig_sincos(x, s, c); }
void orc_ptk_atan2_array(const float* y, const float* x, int n, float* out) { for (int i = 0; i < n; ++i) out[i] = ptk_atan2(y[i], x[i]); }
void orc_ptk_sincos_array(const float* x, int n, float* s, float* c) { for (int i = 0; i < n; ++i) ptk_sincos(x[i], s + i, c + i); }

// geom given as 50 dwords; ray o[3], d[3]; out: t, P[3], N[3]
float orc_box_test(const void* geom, const float* o, const float* d, float* P, float* N) {
    Ray r; r.origin = v3(o[0], o[1], o[2]); r.direction = v3(d[0], d[1], d[2]);
    v3 p, n; float t = boxIntersectionTest(*(const Geom*)geom, r, p, n);
    P[0] = p.x; P[1] = p.y; P[2] = p.z; N[0] = n.x; N[1] = n.y; N[2] = n.z; return t;
}
float orc_sphere_test(const void* geom, const float* o, const float* d, float* P, float* N) {
    Ray r; r.origin = v3(o[0], o[1], o[2]); r.direction = v3(d[0], d[1], d[2]);
    v3 p, n; float t = sphereIntersectionTest(*(const Geom*)geom, r, p, n);
    P[0] = p.x; P[1] = p.y; P[2] = p.z; N[0] = n.x; N[1] = n.y; N[2] = n.z; return t;
}
// batch forms of the two probes above (n rays against one primitive): t only
void orc_box_test_array(const void* geom, const float* o, const float* d, int n, float* t) {
    for (int i = 0; i < n; ++i) {
        Ray r; r.origin = v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]); r.direction = v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
        v3 p, nn; t[i] = boxIntersectionTest(*(const Geom*)geom, r, p, nn);
    }
}
void orc_sphere_test_array(const void* geom, const float* o, const float* d, int n, float* t) {
    for (int i = 0; i < n; ++i) {
        Ray r; r.origin = v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]); r.direction = v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
        v3 p, nn; t[i] = sphereIntersectionTest(*(const Geom*)geom, r, p, nn);
    }
}
int orc_aabb_test(const float* pmin, const float* pmax, const float* o, const float* d, float* t) {
    Ray r; r.origin = v3(o[0], o[1], o[2]); r.direction = v3(d[0], d[1], d[2]);
    return aabbIntersectP(v3(pmin[0], pmin[1], pmin[2]), v3(pmax[0], pmax[1], pmax[2]), r, *t) ? 1 : 0;
}
int orc_tri_test(const void* tri, const float* o, const float* d, float* tuv) {
    Ray r; r.origin = v3(o[0], o[1], o[2]); r.direction = v3(d[0], d[1], d[2]);
    return triIntersect(*(const Triangle*)tri, r, tuv[0], tuv[1], tuv[2]) ? 1 : 0;
}
// unit-transform helper: build a Geom (50 dwords) from TRANS/ROTAT/SCALE like scene.cpp:150-154
void orc_make_geom(int type, int materialid, const float* tr, const float* rot, const float* sc, void* out) {
    Geom g; g.type = type; g.materialid = materialid;
    g.transform = buildTransformationMatrix(v3(tr[0], tr[1], tr[2]), v3(rot[0], rot[1], rot[2]), v3(sc[0], sc[1], sc[2]));
    g.inverseTransform = m4_inverse(g.transform); g.invTranspose = m4_inverse_transpose(g.transform);
    memcpy(out, &g, sizeof g);
}
// scatterSample probe: material (10 dwords), n[3], wo[3], uv ignored; rng key; out: dir[3], bsdf[3], pdf
void orc_scatter_sample(const void* mat, const float* n, const float* wo, int iter, int idx, int depth, float* out) {
    Isect is = Isect(); is.n = v3(n[0], n[1], n[2]); is.t = 1;
    g_tex_scene = nullptr;                       // a bare material record: constant samplers only
    Rng rng = make_rng(iter, idx, depth);
    ScatterRec s; scatterSample(*(const Material*)mat, is, v3(wo[0], wo[1], wo[2]), s, rng);
    out[0] = s.dir.x; out[1] = s.dir.y; out[2] = s.dir.z; out[3] = s.bsdf.x; out[4] = s.bsdf.y; out[5] = s.bsdf.z; out[6] = s.pdf;
}
void orc_mat_bsdf_pdf(const void* mat, const float* n, const float* wo, const float* wi, float* out) {
    Isect is = Isect(); is.n = v3(n[0], n[1], n[2]);
    g_tex_scene = nullptr;
    v3 f = matBSDF(*(const Material*)mat, is, v3(wo[0], wo[1], wo[2]), v3(wi[0], wi[1], wi[2]));
    out[0] = f.x; out[1] = f.y; out[2] = f.z; out[3] = matPDF(*(const Material*)mat, is, v3(wo[0], wo[1], wo[2]), v3(wi[0], wi[1], wi[2]));
}
// light sampling probes: out = pos[3], emit[3], pdf
void orc_light_sample(OrcScene* S, const float* viewPos, int iter, int idx, int depth, int skip, float* out) {
    Rng rng = make_rng(iter, idx, depth);
    for (int i = 0; i < skip; ++i) sample1D(rng);
    LightRec r; lightSample(*S, v3(viewPos[0], viewPos[1], viewPos[2]), rng, r);
    out[0] = r.pos.x; out[1] = r.pos.y; out[2] = r.pos.z; out[3] = r.emit.x; out[4] = r.emit.y; out[5] = r.emit.z; out[6] = r.pdf;
}
float orc_light_pdf(OrcScene* S, const float* viewPos, const float* lightPos, const float* nrm, int triID, int geomID) {
    return lightPDF(*S, v3(viewPos[0], viewPos[1], viewPos[2]), v3(lightPos[0], lightPos[1], lightPos[2]), v3(nrm[0], nrm[1], nrm[2]), triID, geomID);
}

// --- kernel-level entry points on AoS host arrays (12 dwords per path / per hit) ---
void orc_generate_rays(OrcScene* S, int iter, void* paths) {
    PathSegment* p = (PathSegment*)paths; const Camera& cam = S->cam;
    for (int y = 0; y < cam.res[1]; ++y) for (int x = 0; x < cam.res[0]; ++x) generate_ray(cam, iter, S->traceDepth, x, y, p[x + y * cam.res[0]]);
}
// img (optional): with an ENV map, a miss adds color * env(sphere2Plane(direction)) to its pixel (pathtrace.cu:303-311)
void orc_intersect_env(OrcScene* S, int n, const void* paths, void* isects, int* valid, float* img) {
    const PathSegment* p = (const PathSegment*)paths; Isect* is = (Isect*)isects;
    g_tex_scene = S;
    const bool env = img && S->envMapID >= 0 && S->envMapID < (int)S->textures.size();
#pragma omp parallel num_threads(g_threads) if (g_threads > 1)
    {
#pragma omp for schedule(dynamic, 2048)
        for (int i = 0; i < n; ++i) {
            valid[i] = intersect_one(*S, p[i], is[i]);
            if (!valid[i] && env) {
                v3 c = processNAN(p[i].color * linearSample(S->textures[S->envMapID], sphere2Plane(p[i].d)));
                float* px = radiance_px(img, p[i].pixelIndex); px[0] += c.x; px[1] += c.y; px[2] += c.z;
            }
        }
        stats_fold();
    }
}
void orc_intersect(OrcScene* S, int n, const void* paths, void* isects, int* valid) { orc_intersect_env(S, n, paths, isects, valid, nullptr); }
// shade n compacted paths in place; idx = compacted slot index (the RNG key).  img: float[3*P] accumulated.
void orc_shade(OrcScene* S, int mode, int iter, int depth, int n, const void* isects, void* paths, int* valid, float* img) {
    PathSegment* p = (PathSegment*)paths; const Isect* is = (const Isect*)isects;
    g_tex_scene = S;
#pragma omp parallel for schedule(dynamic, 2048) num_threads(g_threads) if (g_threads > 1)
    for (int i = 0; i < n; ++i) {
        if (mode == MODE_BSDF) valid[i] = shade_bsdf(*S, iter, i, depth, is[i], p[i], img);
        else if (mode == MODE_MIS) valid[i] = shade_mis(*S, iter, i, depth, is[i], p[i], img);
        else valid[i] = shade_direct(*S, iter, i, depth, is[i], p[i], img);
    }
}
// stable compaction of both streams (pathtrace.cu:614-631).  In place; returns the new count.
int orc_compact(int n, const int* valid, void* paths, void* isects) {
    PathSegment* p = (PathSegment*)paths; Isect* is = (Isect*)isects; int m = 0;
    for (int i = 0; i < n; ++i) if (valid[i]) { p[m] = p[i]; if (is) is[m] = is[i]; ++m; }
    return m;
}

// One pathtrace() call (pathtrace.cu:637-816): accumulates into img[3*P]; live[d] = paths entering intersect at bounce d
// (live must hold traceDepth+1 ints; unused entries are set to 0).  Returns total rays (sum of live).
// timing[4] (optional, seconds): generate, intersect, shade, compact.
// Tile partition (the build's multi-GPU sharding; (1,0,1) = the reference's full frame): only the pixels of
// rows with ((y / tile_rows) % n_parts) == part are traced, enumerated in row-major order of the owned rows.
// spp > 1 (the build's samples-per-launch extension): launch `step` traces iterations (step-1)*spp + s + 1, s = 0..spp-1, of the
// owned pixels together: path order is sample-major, camera keys use the sample's own iteration number, shading keys use the
// batch's first iteration, and img must hold spp planes of 3*P floats (sample s accumulates into plane s).
long long orc_pathtrace_batch(OrcScene* S, int mode, int step, int spp, int n_parts, int part, int tile_rows, float* img, int* live, double* timing) {
    const Camera& cam = S->cam;
    int W = cam.res[0], H = cam.res[1];
    // the rank-unique key slot + part * W * H * spp is a 32-bit int on both sides: refuse what would wrap (the product returns PTK_E_BADARG)
    if (n_parts > 1 && (long long)part * W * H * spp + ((long long)W * H * spp) / n_parts + (long long)W * tile_rows * spp > 2147483647LL) return -1;
    g_plane_stride = 3 * (size_t)W * H;
    std::vector<PathSegment> paths; paths.reserve((size_t)W * H * spp / n_parts + W * tile_rows);
    using clk = std::chrono::steady_clock;
    auto t0 = clk::now();
    for (int s = 0; s < spp; ++s) {
        const int iter = (step - 1) * spp + s + 1;
        for (int y = 0; y < H; ++y) {
            if (((y / tile_rows) % n_parts) != part) continue;
            for (int x = 0; x < W; ++x) { PathSegment ps; generate_ray(cam, iter, S->traceDepth, x, y, ps); ps.pixelIndex |= (s << 24); paths.push_back(ps); }
        }
    }
    const int key_iter = (step - 1) * spp + 1;
    g_slot_bias = n_parts > 1 ? part * W * H * spp : 0;
    int P = (int)paths.size();
    std::vector<Isect> isects(P); std::vector<int> valid(P);
    auto t1 = clk::now();
    double tg = std::chrono::duration<double>(t1 - t0).count(), ti = 0, ts = 0, tc = 0;
    int n = P, depth = 0; long long rays = 0;
    for (int d = 0; d <= S->traceDepth; ++d) live[d] = 0;
    while (n > 0) {
        if (depth <= S->traceDepth) live[depth] = n;
        rays += n;
        auto a = clk::now();
        orc_intersect_env(S, n, paths.data(), isects.data(), valid.data(), img);
        auto b = clk::now();
        n = orc_compact(n, valid.data(), paths.data(), isects.data());
        auto c = clk::now();
        orc_shade(S, mode, key_iter, depth, n, isects.data(), paths.data(), valid.data(), img);
        auto d2 = clk::now();
        n = orc_compact(n, valid.data(), paths.data(), isects.data());
        auto e = clk::now();
        ti += std::chrono::duration<double>(b - a).count();
        tc += std::chrono::duration<double>(c - b).count() + std::chrono::duration<double>(e - d2).count();
        ts += std::chrono::duration<double>(d2 - c).count();
        depth++;
    }
    g_slot_bias = 0;
    if (timing) { timing[0] += tg; timing[1] += ti; timing[2] += ts; timing[3] += tc; }
    return rays;
}
long long orc_pathtrace_tiled(OrcScene* S, int mode, int iter, int n_parts, int part, int tile_rows, float* img, int* live, double* timing) {
    return orc_pathtrace_batch(S, mode, iter, 1, n_parts, part, tile_rows, img, live, timing);
}
long long orc_pathtrace(OrcScene* S, int mode, int iter, float* img, int* live, double* timing) {
    return orc_pathtrace_tiled(S, mode, iter, 1, 0, 1, img, live, timing);
}
void orc_stats(unsigned long long* out, int reset) { stats_fold(); out[0] = g_stats.nodes; out[1] = g_stats.tris; if (reset) g_stats = TraverseStats(); }

// sendImageToPBO (pathtrace.cu:32-65): img sum -> uchar4 (w = 0)
void orc_tonemap(const float* img, int P, int iter, unsigned char* pbo) {
    for (int i = 0; i < P; ++i) {
        for (int k = 0; k < 3; ++k) {
            float x = img[3 * i + k] / (float)iter;
            float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
            float y = gclamp((x * (a * x + b)) / (x * (c * x + d) + e), 0.f, 1.f);       // ACESFilm utilities.h:35-43
            y = powf(y, 1 / 2.2f);                                                       // gammaCorrection utilities.h:45-48
            y = gclamp(y * 255.0f, 0.f, 255.0f);
            pbo[4 * i + k] = (unsigned char)y;
        }
        pbo[4 * i + 3] = 0;
    }
}

} // extern "C"
