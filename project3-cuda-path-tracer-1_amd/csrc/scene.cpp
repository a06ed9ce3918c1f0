// scene.cpp — host scene layer: scene-file parser, OBJ reader, SAH BVH + 6-direction threaded BVH, light list,
// viewer camera fix-ups.  Follows the behaviour (grammar, quirks, arithmetic order) of the reference's
// src/scene.cpp, src/BVH.cpp, src/utilities.cpp and src/main.cpp so that the tables it produces are identical;
// cited per function.  Pure host C++ (no HIP calls here; the upload lives in dev_scene.cpp).
#include "scene.h"
#include "image.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>

using ptk::vec2;
using ptk::vec3;
using ptk::mat4;

namespace {

// ---------------------------------------------------------------- small linear algebra (GLM 0.9.6.3 operation order)
inline vec3 add(const vec3& a, const vec3& b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 sub(const vec3& a, const vec3& b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 scl(const vec3& a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 divs(const vec3& a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
inline float dot3(const vec3& a, const vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross3(const vec3& x, const vec3& y) {
    return vec3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
inline vec3 unit(const vec3& v) { return scl(v, 1.0f / std::sqrt(dot3(v, v))); }      // x * inversesqrt(dot(x,x))
inline float fminG(float x, float y) { return x < y ? x : y; }
inline float fmaxG(float x, float y) { return x > y ? x : y; }
inline float clampG(float x, float lo, float hi) { return fminG(fmaxG(x, lo), hi); }
inline vec3 vminG(const vec3& a, const vec3& b) { return vec3(fminG(a.x, b.x), fminG(a.y, b.y), fminG(a.z, b.z)); }
inline vec3 vmaxG(const vec3& a, const vec3& b) { return vec3(fmaxG(a.x, b.x), fmaxG(a.y, b.y), fmaxG(a.z, b.z)); }

const float kPi = 3.1415926535897932384626422832795028841971f;

// column c of M scaled and summed the way glm's mat4*mat4 does: ((A0*b0 + A1*b1) + A2*b2) + A3*b3
mat4 matmul(const mat4& A, const mat4& B) {
    mat4 R;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r)
            R.at(c, r) = A.at(0, r) * B.at(c, 0) + A.at(1, r) * B.at(c, 1) + A.at(2, r) * B.at(c, 2) + A.at(3, r) * B.at(c, 3);
    return R;
}

// glm::translate(mat4(), v)
mat4 translation(const vec3& v) {
    mat4 I = mat4::identity(), R = I;
    for (int r = 0; r < 4; ++r) R.at(3, r) = I.at(0, r) * v.x + I.at(1, r) * v.y + I.at(2, r) * v.z + I.at(3, r);
    return R;
}
// glm::rotate(mat4(), angle, axis) (gtc/matrix_transform.inl)
mat4 rotation(float angle, const vec3& axisIn) {
    const float c = std::cos(angle), s = std::sin(angle);
    const vec3 ax = unit(axisIn);
    const vec3 tmp = vec3((1.f - c) * ax.x, (1.f - c) * ax.y, (1.f - c) * ax.z);
    float Rm[3][3];
    Rm[0][0] = c + tmp.x * ax.x;
    Rm[0][1] = 0 + tmp.x * ax.y + s * ax.z;
    Rm[0][2] = 0 + tmp.x * ax.z - s * ax.y;
    Rm[1][0] = 0 + tmp.y * ax.x - s * ax.z;
    Rm[1][1] = c + tmp.y * ax.y;
    Rm[1][2] = 0 + tmp.y * ax.z + s * ax.x;
    Rm[2][0] = 0 + tmp.z * ax.x + s * ax.y;
    Rm[2][1] = 0 + tmp.z * ax.y - s * ax.x;
    Rm[2][2] = c + tmp.z * ax.z;
    const mat4 I = mat4::identity();
    mat4 R;
    for (int c2 = 0; c2 < 3; ++c2)
        for (int r = 0; r < 4; ++r)
            R.at(c2, r) = I.at(0, r) * Rm[c2][0] + I.at(1, r) * Rm[c2][1] + I.at(2, r) * Rm[c2][2];
    for (int r = 0; r < 4; ++r) R.at(3, r) = I.at(3, r);
    return R;
}
// glm::scale(mat4(), v)
mat4 scaling(const vec3& v) {
    const mat4 I = mat4::identity();
    mat4 R;
    for (int r = 0; r < 4; ++r) { R.at(0, r) = I.at(0, r) * v.x; R.at(1, r) = I.at(1, r) * v.y; R.at(2, r) = I.at(2, r) * v.z; R.at(3, r) = I.at(3, r); }
    return R;
}
// utilityCore::buildTransformationMatrix, utilities.cpp:65-72: T * (Rx * Ry * Rz) * S, degrees
mat4 buildTransformationMatrix(const vec3& t, const vec3& rot, const vec3& sc) {
    mat4 T = translation(t);
    mat4 R = rotation(rot.x * (float)kPi / 180, vec3(1, 0, 0));
    R = matmul(R, rotation(rot.y * (float)kPi / 180, vec3(0, 1, 0)));
    R = matmul(R, rotation(rot.z * (float)kPi / 180, vec3(0, 0, 1)));
    return matmul(matmul(T, R), scaling(sc));
}

inline float det2(float a, float b, float c, float d) { return a * b - c * d; }     // a*b - c*d as written in glm

// glm::inverse(mat4) — detail::compute_inverse, type_mat4x4.inl:37-90
mat4 inverse(const mat4& M) {
    auto m = [&](int c, int r) { return M.at(c, r); };
    const float c00 = det2(m(2,2), m(3,3), m(3,2), m(2,3)), c02 = det2(m(1,2), m(3,3), m(3,2), m(1,3)), c03 = det2(m(1,2), m(2,3), m(2,2), m(1,3));
    const float c04 = det2(m(2,1), m(3,3), m(3,1), m(2,3)), c06 = det2(m(1,1), m(3,3), m(3,1), m(1,3)), c07 = det2(m(1,1), m(2,3), m(2,1), m(1,3));
    const float c08 = det2(m(2,1), m(3,2), m(3,1), m(2,2)), c10 = det2(m(1,1), m(3,2), m(3,1), m(1,2)), c11 = det2(m(1,1), m(2,2), m(2,1), m(1,2));
    const float c12 = det2(m(2,0), m(3,3), m(3,0), m(2,3)), c14 = det2(m(1,0), m(3,3), m(3,0), m(1,3)), c15 = det2(m(1,0), m(2,3), m(2,0), m(1,3));
    const float c16 = det2(m(2,0), m(3,2), m(3,0), m(2,2)), c18 = det2(m(1,0), m(3,2), m(3,0), m(1,2)), c19 = det2(m(1,0), m(2,2), m(2,0), m(1,2));
    const float c20 = det2(m(2,0), m(3,1), m(3,0), m(2,1)), c22 = det2(m(1,0), m(3,1), m(3,0), m(1,1)), c23 = det2(m(1,0), m(2,1), m(2,0), m(1,1));
    const float F0[4] = {c00, c00, c02, c03}, F1[4] = {c04, c04, c06, c07}, F2[4] = {c08, c08, c10, c11};
    const float F3[4] = {c12, c12, c14, c15}, F4[4] = {c16, c16, c18, c19}, F5[4] = {c20, c20, c22, c23};
    const float V0[4] = {m(1,0), m(0,0), m(0,0), m(0,0)}, V1[4] = {m(1,1), m(0,1), m(0,1), m(0,1)};
    const float V2[4] = {m(1,2), m(0,2), m(0,2), m(0,2)}, V3[4] = {m(1,3), m(0,3), m(0,3), m(0,3)};
    mat4 Inv;
    for (int r = 0; r < 4; ++r) {
        const float sa = (r & 1) ? -1.f : +1.f, sb = -sa;
        Inv.at(0, r) = (V1[r] * F0[r] - V2[r] * F1[r] + V3[r] * F2[r]) * sa;
        Inv.at(1, r) = (V0[r] * F0[r] - V2[r] * F3[r] + V3[r] * F4[r]) * sb;
        Inv.at(2, r) = (V0[r] * F1[r] - V1[r] * F3[r] + V3[r] * F5[r]) * sa;
        Inv.at(3, r) = (V0[r] * F2[r] - V1[r] * F4[r] + V2[r] * F5[r]) * sb;
    }
    const float d0 = m(0,0) * Inv.at(0,0), d1 = m(0,1) * Inv.at(1,0), d2 = m(0,2) * Inv.at(2,0), d3 = m(0,3) * Inv.at(3,0);
    const float oneOverDet = 1.f / ((d0 + d1) + (d2 + d3));
    for (int i = 0; i < 16; ++i) Inv.m[i] = Inv.m[i] * oneOverDet;
    return Inv;
}

// glm::inverseTranspose(mat4), gtc/matrix_inverse.inl:95-147
mat4 inverseTranspose(const mat4& M) {
    auto m = [&](int c, int r) { return M.at(c, r); };
    const float s00 = det2(m(2,2), m(3,3), m(3,2), m(2,3)), s01 = det2(m(2,1), m(3,3), m(3,1), m(2,3)), s02 = det2(m(2,1), m(3,2), m(3,1), m(2,2));
    const float s03 = det2(m(2,0), m(3,3), m(3,0), m(2,3)), s04 = det2(m(2,0), m(3,2), m(3,0), m(2,2)), s05 = det2(m(2,0), m(3,1), m(3,0), m(2,1));
    const float s06 = det2(m(1,2), m(3,3), m(3,2), m(1,3)), s07 = det2(m(1,1), m(3,3), m(3,1), m(1,3)), s08 = det2(m(1,1), m(3,2), m(3,1), m(1,2));
    const float s09 = det2(m(1,0), m(3,3), m(3,0), m(1,3)), s10 = det2(m(1,0), m(3,2), m(3,0), m(1,2)), s11 = det2(m(1,1), m(3,3), m(3,1), m(1,3));
    const float s12 = det2(m(1,0), m(3,1), m(3,0), m(1,1)), s13 = det2(m(1,2), m(2,3), m(2,2), m(1,3)), s14 = det2(m(1,1), m(2,3), m(2,1), m(1,3));
    const float s15 = det2(m(1,1), m(2,2), m(2,1), m(1,2)), s16 = det2(m(1,0), m(2,3), m(2,0), m(1,3)), s17 = det2(m(1,0), m(2,2), m(2,0), m(1,2));
    const float s18 = det2(m(1,0), m(2,1), m(2,0), m(1,1));
    mat4 I;
    I.at(0,0) = +(m(1,1) * s00 - m(1,2) * s01 + m(1,3) * s02);
    I.at(0,1) = -(m(1,0) * s00 - m(1,2) * s03 + m(1,3) * s04);
    I.at(0,2) = +(m(1,0) * s01 - m(1,1) * s03 + m(1,3) * s05);
    I.at(0,3) = -(m(1,0) * s02 - m(1,1) * s04 + m(1,2) * s05);
    I.at(1,0) = -(m(0,1) * s00 - m(0,2) * s01 + m(0,3) * s02);
    I.at(1,1) = +(m(0,0) * s00 - m(0,2) * s03 + m(0,3) * s04);
    I.at(1,2) = -(m(0,0) * s01 - m(0,1) * s03 + m(0,3) * s05);
    I.at(1,3) = +(m(0,0) * s02 - m(0,1) * s04 + m(0,2) * s05);
    I.at(2,0) = +(m(0,1) * s06 - m(0,2) * s07 + m(0,3) * s08);
    I.at(2,1) = -(m(0,0) * s06 - m(0,2) * s09 + m(0,3) * s10);
    I.at(2,2) = +(m(0,0) * s11 - m(0,1) * s09 + m(0,3) * s12);
    I.at(2,3) = -(m(0,0) * s08 - m(0,1) * s10 + m(0,2) * s12);
    I.at(3,0) = -(m(0,1) * s13 - m(0,2) * s14 + m(0,3) * s15);
    I.at(3,1) = +(m(0,0) * s13 - m(0,2) * s16 + m(0,3) * s17);
    I.at(3,2) = -(m(0,0) * s14 - m(0,1) * s16 + m(0,3) * s18);
    I.at(3,3) = +(m(0,0) * s15 - m(0,1) * s17 + m(0,2) * s18);
    const float det = +m(0,0) * I.at(0,0) + m(0,1) * I.at(0,1) + m(0,2) * I.at(0,2) + m(0,3) * I.at(0,3);
    for (int i = 0; i < 16; ++i) I.m[i] = I.m[i] / det;
    return I;
}

// vec3(M * vec4(v, w)): (M0*x + M1*y) + (M2*z + M3*w), type_mat4x4.inl:617-628
vec3 transform3(const mat4& M, const vec3& v, float w) {
    vec3 r;
    r.x = (M.at(0,0) * v.x + M.at(1,0) * v.y) + (M.at(2,0) * v.z + M.at(3,0) * w);
    r.y = (M.at(0,1) * v.x + M.at(1,1) * v.y) + (M.at(2,1) * v.z + M.at(3,1) * w);
    r.z = (M.at(0,2) * v.x + M.at(1,2) * v.y) + (M.at(2,2) * v.z + M.at(3,2) * w);
    return r;
}

// ---------------------------------------------------------------- text helpers (utilities.cpp:74-112)
std::vector<std::string> splitWs(const std::string& s) {
    std::vector<std::string> out;
    std::istringstream ss(s);
    std::string w;
    while (ss >> w) out.push_back(w);
    return out;
}
// one line, accepting \n, \r\n and \r terminators; sets eofbit when nothing could be read
void readLine(std::istream& is, std::string& line) {
    line.clear();
    std::streambuf* sb = is.rdbuf();
    while (true) {
        const int ch = sb->sbumpc();
        if (ch == '\n') return;
        if (ch == '\r') { if (sb->sgetc() == '\n') sb->sbumpc(); return; }
        if (ch == EOF) { if (line.empty()) is.setstate(std::ios::eofbit); return; }
        line.push_back((char)ch);
    }
}
bool isPlainNumber(const std::string& t) {
    char* e = nullptr;
    std::strtod(t.c_str(), &e);
    return e != t.c_str() && *e == 0;
}
float toF(const std::string& s) { return (float)std::atof(s.c_str()); }
vec3 toV3(const std::vector<std::string>& t) { return vec3(toF(t[1]), toF(t[2]), toF(t[3])); }

// ---------------------------------------------------------------- bounds (Bounds3.hpp)
struct Box {
    vec3 lo, hi;
    Box() : lo(vec3(std::numeric_limits<float>::infinity())), hi(vec3(-std::numeric_limits<float>::infinity())) {}  // vec3(DBL_MAX) as float
    Box(const vec3& a, const vec3& b) : lo(vminG(a, b)), hi(vmaxG(a, b)) {}
    void grow(const Box& b) { lo = vminG(lo, b.lo); hi = vmaxG(hi, b.hi); }
    vec3 diagonal() const { return sub(hi, lo); }
    vec3 centroid() const { return divs(add(hi, lo), 2.0f); }
    int maxExtent() const { const vec3 d = diagonal(); if (d.x > d.y && d.x > d.z) return 0; return d.y > d.z ? 1 : 2; }
    // Bounds3::isValid as written (Bounds3.hpp:69-73): only the x test can ever fire
    bool validAsWritten() const { return !(lo.x > hi.x || lo.y > lo.y || lo.z > lo.z); }
    double surfaceArea() const {
        const vec3 d = diagonal();
        const double a = 2 * ((double)d.x * (double)d.y + (double)d.y * (double)d.z + (double)d.z * (double)d.x);
        return validAsWritten() ? a : 0;
    }
};
Box boundOf(const Triangle& t) {                       // Triangle::getBound = Union(Bounds3(v0, v1), v2)
    Box b(t.v[0], t.v[1]);
    Box r; r.lo = vminG(b.lo, t.v[2]); r.hi = vmaxG(b.hi, t.v[2]);
    return r;
}
vec3 centroidOf(const Triangle& t) { return divs(add(add(t.v[0], t.v[1]), t.v[2]), 3.f); }

// ---------------------------------------------------------------- BVH build (BVH.cpp)
constexpr int kMaxPrim = 1;        // MAX_PRIM, BVH.h:5
constexpr int kBuckets = 20;       // BUCKET_NUM, BVH.h:6

struct FlatNode { int start, end, parent, left, right; Box box; };

// recursiveBuildSAH (BVH.cpp:13-92) writing nodes in preorder, which is the order recursiveBuildGpuBVHInfo (121-147) assigns
int buildNode(std::vector<Triangle>& tris, int start, int end, int parent, std::vector<FlatNode>& out) {
    const int self = (int)out.size();
    out.push_back(FlatNode{start, end, parent, -1, -1, Box()});
    Box bounds;
    vec3 cLo(FLT_MAX), cHi(-FLT_MAX);
    for (int i = start; i < end; ++i) {
        bounds.grow(boundOf(tris[i]));
        cLo = vminG(cLo, centroidOf(tris[i]));
        cHi = vmaxG(cHi, centroidOf(tris[i]));
    }
    out[self].box = bounds;
    if (end - start <= std::max(kMaxPrim, 1)) return self;

    const Box cBox(cLo, cHi);
    const int axis = cBox.maxExtent();
    const float lo = cBox.lo.get(axis), extent = cBox.hi.get(axis) - cBox.lo.get(axis);
    auto bucketIndex = [&](const Triangle& t) {
        const float off = clampG((centroidOf(t).get(axis) - lo) / extent, 0.f, 1.f);
        return off == 1.f ? kBuckets - 1 : static_cast<int>(off * kBuckets);
    };
    int count[kBuckets] = {0};
    Box bbox[kBuckets];
    for (int i = start; i < end; ++i) { const int b = bucketIndex(tris[i]); count[b] += 1; bbox[b].grow(boundOf(tris[i])); }

    float best = FLT_MAX;
    int splitBucket = 0;
    for (int s = 0; s < kBuckets - 1; ++s) {
        Box L, R; int nL = 0, nR = 0;
        for (int j = 0; j <= s; ++j) { L.grow(bbox[j]); nL += count[j]; }
        for (int j = s + 1; j < kBuckets; ++j) { R.grow(bbox[j]); nR += count[j]; }
        const float cost = (float)((nL * L.surfaceArea() + nR * R.surfaceArea()) / bounds.surfaceArea());
        if (cost < best && nL != 0 && nR != 0) { best = cost; splitBucket = s; }
    }
    // std::partition is not stable; the triangle order it leaves is libstdc++'s (the pins are to a libstdc++ build)
    auto midIt = std::partition(tris.begin() + start, tris.begin() + end, [&](const Triangle& t) { return bucketIndex(t) <= splitBucket; });
    int mid = (int)(midIt - tris.begin());
    if (mid == start || mid == end) mid = (start + end) / 2;   // guard: the reference never terminates on degenerate input
    const int l = buildNode(tris, start, mid, self, out);
    const int r = buildNode(tris, mid, end, self, out);
    out[self].left = l; out[self].right = r;
    return self;
}

// buildGpuMTBVH, BVH.cpp:180-239: six copies, children ordered along +x,+y,+z,-x,-y,-z, hit/miss threading
void threadSixWays(const std::vector<FlatNode>& nodes, std::vector<GpuBVHNode>& out) {
    const int n = (int)nodes.size();
    out.assign((size_t)6 * n, GpuBVHNode());
    for (int d = 0; d < 6; ++d) {
        GpuBVHNode* T = out.data() + (size_t)d * n;
        const int axis = d % 3, sign = d < 3 ? 1 : -1;
        for (int i = 0; i < n; ++i) {
            const FlatNode& N = nodes[i];
            T[i].pMin = N.box.lo; T[i].pMax = N.box.hi; T[i].start = N.start; T[i].end = N.end;
            const int parent = N.parent;
            const auto sibling = [&]() { return i == nodes[parent].left ? nodes[parent].right : nodes[parent].left; };
            if (N.left != -1) {
                int first = N.left, second = N.right;
                if (nodes[first].box.centroid().get(axis) * sign > nodes[second].box.centroid().get(axis) * sign) std::swap(first, second);
                T[i].hit = first;
            } else if (parent == -1) T[i].hit = -1;
            else if (i == T[parent].hit) T[i].hit = sibling();
            else T[i].hit = T[parent].miss;

            if (i == 0) T[i].miss = -1;
            else if (i == T[parent].hit) T[i].miss = sibling();
            else T[i].miss = T[parent].miss;
        }
    }
}

std::map<std::string, MeshData*> g_meshPool;           // Resource::meshPool, scene.cpp:43
std::map<std::string, image*> g_texturePool;           // Resource::texturePool, scene.cpp:44

} // namespace

mat4 mat4::identity() { mat4 r; std::memset(r.m, 0, sizeof r.m); r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.f; return r; }

// ---------------------------------------------------------------- OBJ (scene.cpp:340-440)
// Minimal reader replacing tinyobjloader's LoadObj(..., triangulate=true): v / vn / vt / f, fan triangulation,
// vertex normals when the file has any `vn`, else the face normal; texcoords when it has any `vt`.
MeshData* Resource::loadObj(const std::string& filename, const int geomIdx, std::string* err) {
    auto hit = g_meshPool.find(filename);
    if (hit != g_meshPool.end()) return hit->second;
    std::ifstream in(filename.c_str());
    if (!in.good()) { if (err) *err = "cannot open obj file " + filename; return nullptr; }
    std::vector<vec3> pos, nrm;
    std::vector<vec2> uv;
    struct Corner { int p, t, n; };
    std::vector<std::vector<Corner>> faces;
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string tag;
        if (!(ls >> tag)) continue;
        if (tag == "v" || tag == "vn" || tag == "vt") {
            std::string a, b, c;
            ls >> a >> b; if (tag != "vt") ls >> c;
            // tinyobj parses reals as double and narrows to float
            const float x = (float)std::strtod(a.c_str(), nullptr), y = (float)std::strtod(b.c_str(), nullptr);
            if (tag == "vt") uv.push_back(vec2(x, y));
            else { const float z = (float)std::strtod(c.c_str(), nullptr); (tag == "v" ? pos : nrm).push_back(vec3(x, y, z)); }
        } else if (tag == "f") {
            std::vector<Corner> f;
            std::string tok;
            while (ls >> tok) {
                Corner c{0, 0, 0};
                const char* s = tok.c_str(); char* e;
                c.p = (int)std::strtol(s, &e, 10); s = e;
                if (*s == '/') { ++s; if (*s != '/') { c.t = (int)std::strtol(s, &e, 10); s = e; } if (*s == '/') { ++s; c.n = (int)std::strtol(s, &e, 10); } }
                c.p = c.p > 0 ? c.p - 1 : (int)pos.size() + c.p;
                c.t = c.t > 0 ? c.t - 1 : (c.t < 0 ? (int)uv.size() + c.t : -1);
                c.n = c.n > 0 ? c.n - 1 : (c.n < 0 ? (int)nrm.size() + c.n : -1);
                f.push_back(c);
            }
            faces.push_back(f);
        }
    }
    const bool hasTex = !uv.empty(), hasNor = !nrm.empty();
    MeshData* mesh = new MeshData;
    for (const auto& f : faces) {
        for (size_t i = 0; i + 2 < f.size(); ++i) {
            const Corner cs[3] = {f[0], f[i + 1], f[i + 2]};
            Triangle t;
            for (int k = 0; k < 3; ++k) {
                if (cs[k].p < 0 || cs[k].p >= (int)pos.size()) { if (err) *err = "obj index out of range in " + filename; delete mesh; return nullptr; }
                t.v[k] = pos[cs[k].p];
            }
            if (hasNor) { for (int k = 0; k < 3; ++k) if (cs[k].n >= 0 && cs[k].n < (int)nrm.size()) t.n[k] = nrm[cs[k].n]; }
            else { const vec3 fn = unit(cross3(sub(t.v[1], t.v[0]), sub(t.v[2], t.v[0]))); t.n[0] = t.n[1] = t.n[2] = fn; }
            if (hasTex) for (int k = 0; k < 3; ++k) if (cs[k].t >= 0 && cs[k].t < (int)uv.size()) t.tex[k] = uv[cs[k].t];
            t.geomIdx = geomIdx;
            mesh->triangles.push_back(t);
        }
    }
    g_meshPool[filename] = mesh;
    return mesh;
}

void Resource::clear() {
    for (auto& kv : g_meshPool) delete kv.second;
    g_meshPool.clear();
    for (auto& kv : g_texturePool) delete kv.second;
    g_texturePool.clear();
}

// Resource::loadTexture, scene.cpp:465-477.  nullptr + empty *err: no such file (the caller falls back to numbers);
// nullptr + message: the file exists but cannot be decoded by this build (PNG 8-bit and Radiance .hdr only).
image* Resource::loadTexture(const std::string& filename, float gamma, std::string* err) {
    auto hit = g_texturePool.find(filename);
    if (hit != g_texturePool.end()) return hit->second;
    { std::ifstream probe(filename.c_str(), std::ios::binary); if (!probe.good() || probe.peek() == EOF) return nullptr; }   // no such file (or a directory)
    image* tex = new image(filename, gamma);
    if (!tex->ok()) { if (err) *err = "cannot decode image " + filename + ": " + tex->error; delete tex; return nullptr; }
    g_texturePool[filename] = tex;
    return tex;
}

// ---------------------------------------------------------------- Scene (scene.cpp:47-337)
Scene::Scene(const std::string& filename, int res_w, int res_h, int depth) : ovrW(res_w), ovrH(res_h), ovrDepth(depth) {
    const size_t slash = filename.find_last_of('/');
    sceneDir = slash == std::string::npos ? std::string(".") : filename.substr(0, slash);
    std::ifstream in(filename.c_str());
    if (!in.is_open()) { error = "Error reading from file - aborting! (" + filename + ")"; return; }
    while (in.good()) {
        std::string line;
        readLine(in, line);
        if (line.empty()) continue;
        const std::vector<std::string> tk = splitWs(line);
        if (tk.empty()) continue;
        int rc = 1;
        if (tk[0] == "MATERIAL" && tk.size() > 1) rc = loadMaterial(in, tk[1]);
        else if (tk[0] == "OBJECT" && tk.size() > 1) rc = loadGeom(in, tk[1]);
        else if (tk[0] == "CAMERA") rc = loadCamera(in);
        else if (tk[0] == "ENV" && tk.size() > 1) { envMapID = loadTexture(tk[1]); rc = envMapID == -2 ? -2 : 1; if (envMapID < 0) envMapID = -1; }
        if (rc == -2) return;
    }
    ok = true;
}

Scene::~Scene() { clear(); }

std::string Scene::resolvePath(std::string p) const {
    std::replace(p.begin(), p.end(), '\\', '/');
    { std::ifstream f(p.c_str()); if (f.good()) return p; }             // reference: relative to the process CWD
    return sceneDir + "/" + p;                                         // else relative to the scene file
}

int Scene::loadMaterial(std::istream& in, const std::string& name) {  // scene.cpp:243-316
    if (materialNameMap.count(name)) return -1;
    Material m;
    static const char* kTypeNames[5] = {"Lambertian", "MetallicWorkflow", "Dielectric", "Microfacet", "Light"};
    for (int i = 0; i < 6; ++i) {
        std::string line;
        readLine(in, line);
        const std::vector<std::string> tk = splitWs(line);
        if (tk.empty()) break;
        const std::string& key = tk[0];
        if (key == "TYPE" && tk.size() > 1) {
            for (int k = 0; k < 5; ++k) if (tk[1] == kTypeNames[k]) m.type = (Material::Type)k;
        } else if (key == "ALBEDO" || key == "METALLIC" || key == "ROUGHNESS") {
            // The reference first tries the value token as an image file and falls back to numbers (scene.cpp:276-297).
            if (tk.size() < 2) continue;
            const int id = loadTexture(tk[1]);
            if (id == -2) return -2;
            if (key == "ALBEDO") {
                m.albedoMapID = id;
                if (id < 0) {
                    if (tk.size() < 4) { error = "ALBEDO needs three numbers: " + line; return -2; }
                    m.albedo = toV3(tk); m.albedoSampler = m.albedo;
                }
            } else if (key == "METALLIC") {
                m.metallicMapID = id;
                if (id < 0) { m.metallic = toF(tk[1]); m.metallicSampler = m.metallic; }
            } else {
                m.roughnessMapID = id;
                if (id < 0) {
                    const double r = std::atof(tk[1].c_str());
                    m.roughness = (float)(r > 1e-3 ? r : 1e-3);         // glm::max(atof(...), ROUGHNESS_MIN), doubles
                    m.roughnessSampler = m.roughness;
                }
            }
        } else if (key == "NORMAL" && tk.size() > 1) {
            // scene.cpp:303-311: the constant sampler (0.5,0.5,1) is the default either way; a loaded map is bound at upload
            const int id = loadTexture(tk[1]);
            if (id == -2) return -2;
            m.normalMapID = id;
        } else if (key == "IOR" && tk.size() > 1) {
            m.ior = toF(tk[1]);
        }
    }
    materialNameMap[name] = (int)materials.size();
    materials.push_back(m);
    return 1;
}

int Scene::loadTexture(const std::string& fileName, float gamma) {     // scene.cpp:318-337
    std::string err;
    image* tex = Resource::loadTexture(resolvePath(fileName), gamma, &err);
    if (!tex) { if (!err.empty()) { error = err; return -2; } return -1; }
    for (size_t i = 0; i < textures.size(); ++i) if (textures[i] == tex) return (int)i;      // textureIdMap
    textures.push_back(tex);
    return (int)textures.size() - 1;
}

int Scene::loadGeom(std::istream& in, const std::string& name) {      // scene.cpp:83-163
    if (geomNameMap.count(name)) return -1;
    Geom g;
    std::string line;
    readLine(in, line);
    if (!line.empty() && in.good()) {
        if (line == "sphere") g.type = SPHERE;
        else if (line == "cube") g.type = CUBE;
        else if (line.find(".obj") != std::string::npos) {
            g.type = OBJ;
            g.mesh = Resource::loadObj(resolvePath(line), (int)geoms.size(), &error);
            if (!g.mesh) return -2;
        }
    }
    readLine(in, line);
    if (!line.empty() && in.good()) {
        const std::vector<std::string> tk = splitWs(line);
        if (tk.size() > 1) {
            auto it = materialNameMap.find(tk[1]);                     // by name first, else atoi (scene.cpp:124-131)
            g.materialid = it != materialNameMap.end() ? it->second : std::atoi(tk[1].c_str());
        }
    }
    if (g.materialid < 0 || g.materialid >= (int)materials.size()) { error = "object " + name + " refers to an unknown material"; return -2; }
    readLine(in, line);
    while (!line.empty() && in.good()) {
        const std::vector<std::string> tk = splitWs(line);
        if (tk.size() >= 4) {
            if (tk[0] == "TRANS") g.translation = toV3(tk);
            else if (tk[0] == "ROTAT") g.rotation = toV3(tk);
            else if (tk[0] == "SCALE") g.scale = toV3(tk);
        }
        readLine(in, line);
    }
    g.transform = buildTransformationMatrix(g.translation, g.rotation, g.scale);
    g.inverseTransform = inverse(g.transform);
    g.invTranspose = inverseTranspose(g.transform);
    geomNameMap[name] = (int)geoms.size();
    geoms.push_back(g);
    return 1;
}

int Scene::loadCamera(std::istream& in) {                             // scene.cpp:165-236
    Camera& cam = state.camera;
    float fovy = 0.f;
    for (int i = 0; i < 5; ++i) {
        std::string line;
        readLine(in, line);
        const std::vector<std::string> tk = splitWs(line);
        if (tk.size() < 2) continue;
        if (tk[0] == "RES" && tk.size() > 2) { cam.resolution.x = std::atoi(tk[1].c_str()); cam.resolution.y = std::atoi(tk[2].c_str()); }
        else if (tk[0] == "FOVY") fovy = toF(tk[1]);
        else if (tk[0] == "ITERATIONS") state.iterations = std::atoi(tk[1].c_str());
        else if (tk[0] == "DEPTH") state.traceDepth = std::atoi(tk[1].c_str());
        else if (tk[0] == "FILE") state.imageName = tk[1];
    }
    if (ovrW > 0 && ovrH > 0) { cam.resolution.x = ovrW; cam.resolution.y = ovrH; }
    if (ovrDepth > 0) state.traceDepth = ovrDepth;
    std::string line;
    readLine(in, line);
    while (!line.empty() && in.good()) {
        const std::vector<std::string> tk = splitWs(line);
        if (!tk.empty()) {
            if (tk[0] == "EYE" && tk.size() > 3) cam.position = toV3(tk);
            if (tk[0] == "ROTAT" && tk.size() > 2) { theta = clampG(toF(tk[1]), -89.f, 89.f); phi = toF(tk[2]); posInit = false; }
            else if (tk[0] == "LOOKAT" && tk.size() > 3) { cam.lookAt = toV3(tk); posInit = true; }
            else if (tk[0] == "UP" && tk.size() > 3) cam.up = toV3(tk);
        }
        readLine(in, line);
    }
    if (cam.resolution.x <= 0 || cam.resolution.y <= 0) { error = "CAMERA block without a valid RES line"; return -2; }
    // the image is 12 B per pixel on host and device and pixel offsets are 32-bit (samples per launch narrows this to 2^24 pixels:
    // pixelIndex then shares its word with the sample plane, checked in ptk_pathtrace_init)
    if ((long long)cam.resolution.x * (long long)cam.resolution.y > (1ll << 29)) { error = "RES too large (W * H must not exceed 2^29)"; return -2; }
    const float yscaled = std::tan(fovy * (kPi / 180));                 // FOVY is used as a HALF angle
    const float xscaled = (yscaled * cam.resolution.x) / cam.resolution.y;
    const float fovx = (std::atan(xscaled) * 180) / kPi;
    cam.fov = vec2(fovx, fovy);
    cam.view = unit(sub(cam.lookAt, cam.position));
    cam.right = unit(cross3(cam.view, cam.up));
    cam.pixelLength = vec2(2 * xscaled / (float)cam.resolution.x, 2 * yscaled / (float)cam.resolution.y);
    state.image.assign((size_t)cam.resolution.x * cam.resolution.y, vec3());
    return 1;
}

// main.cpp:102-127 (start-up) followed by runCuda()'s camchanged block, main.cpp:173-193
void Scene::applyViewerCamera() {
    Camera& cam = state.camera;
    const float kDeg = 57.295779513082320876798154814105f, kRad = 0.01745329251994329576923690768489f;   // glm::degrees / radians
    if (posInit) {
        phi = std::atan2(cam.view.z, cam.view.x) * kDeg;
        theta = clampG(std::sin(cam.view.y) * kDeg, -89.f, 89.f);       // sin, not asin — as in the reference
    } else {
        const float rt = theta * kRad, rp = phi * kRad;
        cam.view = vec3(std::cos(rt) * std::cos(rp), std::sin(rt), std::cos(rt) * std::sin(rp));
        cam.lookAt = add(cam.position, cam.view);
    }
    cam.right = cross3(cam.view, cam.up);
    cam.up = cross3(cam.right, cam.view);
    const float rt = theta * kRad, rp = phi * kRad;
    cam.view = vec3(std::cos(rt) * std::cos(rp), std::sin(rt), std::cos(rt) * std::sin(rp));
    const vec3 r = cross3(cam.view, vec3(0, 1, 0));
    cam.up = unit(cross3(r, cam.view));
    cam.right = unit(r);
}

// host half of Scene::setDevData, scene.cpp:479-558
void Scene::buildHostData() {
    if (hostDataBuilt) return;
    triangles.clear(); lights.clear();
    for (size_t gi = 0; gi < geoms.size(); ++gi) {
        const Geom& g = geoms[gi];
        if (g.type != OBJ || !g.mesh) continue;
        for (const Triangle& src : g.mesh->triangles) {
            Triangle w;
            for (int i = 0; i < 3; ++i) {
                w.v[i] = transform3(g.transform, src.v[i], 1.0f);
                w.n[i] = unit(transform3(g.invTranspose, src.n[i], 0.0f));
                w.tex[i] = src.tex[i];
                // tangent frame, recomputed on every vertex pass exactly like scene.cpp:493-507 (last pass wins)
                w.tangent = vec3(); w.bitangent = vec3();
                const vec3 e1 = sub(w.v[1], w.v[0]), e2 = sub(w.v[2], w.v[0]);
                const float du1 = w.tex[1].x - w.tex[0].x, dv1 = w.tex[1].y - w.tex[0].y;
                const float du2 = w.tex[2].x - w.tex[0].x, dv2 = w.tex[2].y - w.tex[0].y;
                const float f = du1 * dv2 - du2 * dv1;
                if ((double)std::fabs(f) < 1e-8) continue;
                w.tangent = unit(divs(sub(scl(e1, dv2), scl(e2, dv1)), f));
                w.bitangent = unit(divs(add(scl(e1, -du2), scl(e2, du1)), f));
            }
            w.geomIdx = (int)gi;
            triangles.push_back(w);
        }
    }
    std::vector<FlatNode> nodes;
    buildNode(triangles, 0, (int)triangles.size(), -1, nodes);
    bvhSize = (int)nodes.size();
    threadSixWays(nodes, gpuBVHNodes);
    for (size_t i = 0; i < geoms.size(); ++i)                            // analytic lights first, in geom order
        if (materials[geoms[i].materialid].type == Material::Light && geoms[i].type != OBJ)
            lights.emplace_back((int)i, -1, geoms[i].type);
    for (size_t i = 0; i < triangles.size(); ++i)                        // then every triangle of a Light geom
        if (materials[geoms[triangles[i].geomIdx].materialid].type == Material::Light)
            lights.emplace_back(triangles[i].geomIdx, (int)i, OBJ);
    hostDataBuilt = true;
}

void Scene::setDevData() {
    buildHostData();
    devError = tempDevScene.initiate(*this);
}

void Scene::clear() { tempDevScene.destroy(); }
