// dev_scene.hip — device image of the scene (DevScene::initiate / destroy, reference src/scene.cpp:569-672)
// and the scene part of the C ABI (include/ptk.h).
//
// HBM layout (all read-only during rendering, replicated per GPU; a few MB at most):
//   DevGeom[G]      144 B each : analytic primitives, matrices as 3x4 rows  -> read through scalar loads
//   DevMaterial[M]   80 B each : constant BSDF parameters (roughness/metallic pre-clamped)
//   DevNode[6][B]    32 B each : AABB + the two links of one direction-ordered threaded BVH, leaf/triangle folded into the
//                                link words (two 16-B loads per node visit; the reference's GpuBVHNode is 40 B)
//   DevTriE[T]       48 B each : v0 + the two edges (what a triangle test touches; the reference copies 124 B)
//   DevTriV[T]       36 B each : vertex positions (attribute interpolation of the final hit, light sampling)
//   DevTriA[T]       64 B each : vertex normals, texcoords, geomIdx (touched once per ray, for the final hit)
#include "ptk_internal.h"
#include "image.h"
#include "wide_bvh.h"
#include <cmath>
#include <cstring>
#include <limits>

using namespace ptk;

namespace {

template <typename T>
int upload(void** dst, const std::vector<T>& src) {
    *dst = nullptr;
    const size_t bytes = sizeof(T) * std::max<size_t>(src.size(), 1);     // never a zero-size allocation
    hipError_t e = hipMalloc(dst, bytes);
    if (e != hipSuccess) return (int)e;
    if (!src.empty()) { e = hipMemcpy(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice); if (e != hipSuccess) return (int)e; }
    return 0;
}

void rows3x4(const ptk::mat4& M, float* out) {            // out[r*4+c] = M[c][r], r < 3
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) out[r * 4 + c] = M.at(c, r);
}
float clampG(float x, float lo, float hi) { const float a = x > lo ? x : lo; return a < hi ? a : hi; }   // glm::clamp = min(max(x,lo),hi)

} // namespace

int DevScene::initiate(Scene& scene) {
    destroy();
    std::vector<DevGeom> g(scene.geoms.size());
    for (size_t i = 0; i < g.size(); ++i) {
        const Geom& s = scene.geoms[i];
        std::memset(&g[i], 0, sizeof(DevGeom));
        g[i].type = (int)s.type; g[i].materialid = s.materialid;
        rows3x4(s.inverseTransform, g[i].inv);
        rows3x4(s.transform, g[i].fwd);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) g[i].invT[r * 3 + c] = s.invTranspose.at(c, r);
        bool diag = true;                       // all off-diagonal 3x3 entries exactly +-0 (e.g. ROTAT 0 0 0)
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
            if (r != c && (s.inverseTransform.at(c, r) != 0.f || s.transform.at(c, r) != 0.f || s.invTranspose.at(c, r) != 0.f)) diag = false;
        g[i].diag = diag ? 1 : 0;
    }
    // padded world-space bounds + per-chunk type masks for the conservative candidate test (ptk_device.hpp: analytic_candidates)
    std::vector<DevGeomBox> gb(scene.geoms.size());
    std::vector<DevGeomMask> gm((scene.geoms.size() + 31) / 32 + 1, DevGeomMask{0u, 0u, 0u, 0u});
    for (size_t i = 0; i < gb.size(); ++i) {
        const Geom& s = scene.geoms[i];
        const float inf = std::numeric_limits<float>::infinity();
        DevGeomBox b{{-inf, inf, -inf, inf, -inf, inf}, 0.f, 0.f};
        bool finite = true;
        for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r)
            finite = finite && std::isfinite(s.transform.at(c, r)) && std::isfinite(s.inverseTransform.at(c, r)) && std::isfinite(s.invTranspose.at(c, r));
        if (finite) {
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, mag = 0.0;
            for (int k = 0; k < 8; ++k) {
                const double c[3] = {(k & 1) ? 0.5 : -0.5, (k & 2) ? 0.5 : -0.5, (k & 4) ? 0.5 : -0.5};
                for (int r = 0; r < 3; ++r) {
                    const double w = (double)s.transform.at(0, r) * c[0] + (double)s.transform.at(1, r) * c[1] + (double)s.transform.at(2, r) * c[2] + (double)s.transform.at(3, r);
                    lo[r] = w < lo[r] ? w : lo[r]; hi[r] = w > hi[r] ? w : hi[r];
                    mag = std::fabs(w) > mag ? std::fabs(w) : mag;
                }
            }
            const double pad = (double)PTK_CAND_ABS + (double)PTK_CAND_REL * mag;
            for (int r = 0; r < 3; ++r) {
                b.s[2 * r] = std::nextafter((float)(lo[r] - pad), -inf);
                b.s[2 * r + 1] = std::nextafter((float)(hi[r] + pad), inf);
            }
            bool round = false;
            if (s.type == SPHERE) {                      // PTK_CAND_QUAD: per-ray pad qa + qb * |o|^2
                double smin = 1e300, smax = 0.0, cc = 0.0, col[3][3];
                for (int c = 0; c < 3; ++c) {            // column norms of the 3x3 part = the scale factors (T * R * S)
                    double nrm = 0.0;
                    for (int r = 0; r < 3; ++r) { col[c][r] = (double)s.transform.at(c, r); nrm += col[c][r] * col[c][r]; }
                    nrm = std::sqrt(nrm);
                    smin = nrm < smin ? nrm : smin; smax = nrm > smax ? nrm : smax;
                    cc += (double)s.transform.at(3, c) * (double)s.transform.at(3, c);
                }
                const double Q = smin > 0.0 ? (double)PTK_CAND_QUAD * smax / (smin * smin) : (double)inf;
                b.qa = std::nextafter((float)(2.0 * Q * cc), inf);
                b.qb = std::nextafter((float)(2.0 * Q), inf);
                if (!std::isfinite(b.qa) || !std::isfinite(b.qb)) { b = DevGeomBox{{-inf, inf, -inf, inf, -inf, inf}, 0.f, 0.f}; }
                else if (smin > 0.0 && (smax - smin) <= 1e-6 * smax) {
                    // similarity transform (uniform scale, columns orthogonal): the primitive is a world-space sphere
                    double dev = 0.0;
                    for (int c0 = 0; c0 < 3; ++c0) for (int c1 = c0 + 1; c1 < 3; ++c1) {
                        double d = 0.0;
                        for (int r = 0; r < 3; ++r) d += col[c0][r] * col[c1][r];
                        dev = std::fabs(d) > dev ? std::fabs(d) : dev;
                    }
                    if (dev <= 1e-6 * smax * smax) {
                        const double Rp = 0.5 * smax * (1.0 + 2e-6) + pad;
                        DevGeomBox rb{{s.transform.at(3, 0), s.transform.at(3, 1), s.transform.at(3, 2),
                                       std::nextafter((float)(Rp * Rp), inf), std::nextafter((float)(2.0 * Rp), inf), 0.f}, 0.f, 0.f};
                        if (std::isfinite(rb.s[3])) { b = rb; round = true; }
                    }
                }
            }
            if (round) gm[i / 32].round |= 1u << (i % 32);
        }
        gb[i] = b;
        DevGeomMask& mk = gm[i / 32];
        const unsigned bit = 1u << (i % 32);
        if (s.type == CUBE) mk.cube |= bit;
        if (s.type == SPHERE) mk.sphere |= bit;
        if (g[i].diag) mk.diag |= bit;
    }
    std::vector<DevMaterial> m(scene.materials.size());
    for (size_t i = 0; i < m.size(); ++i) {
        const Material& s = scene.materials[i];
        std::memset(&m[i], 0, sizeof(DevMaterial));
        m[i].type = (int)s.type;
        m[i].albedo[0] = s.albedo.x; m[i].albedo[1] = s.albedo.y; m[i].albedo[2] = s.albedo.z;
        m[i].albedo_s[0] = s.albedoSampler.x; m[i].albedo_s[1] = s.albedoSampler.y; m[i].albedo_s[2] = s.albedoSampler.z;
        m[i].rough = clampG(s.roughnessSampler, static_cast<float>(1e-3), 1.f);     // material.h:274, 486
        m[i].metal = clampG(s.metallicSampler, 0.f, 1.f);                           // material.h:325, 487
        m[i].ior = s.ior;
        for (int k = 0; k < 3; ++k) m[i].albedo_pi[k] = m[i].albedo_s[k] / PTK_PI;      // correctly rounded fp32 division, as on the device
        const int T = (int)scene.textures.size();                                   // scene.cpp:597-618: bind maps that exist
        m[i].albedo_tex = (s.albedoMapID >= 0 && s.albedoMapID < T) ? s.albedoMapID : -1;
        m[i].rough_tex = (s.roughnessMapID >= 0 && s.roughnessMapID < T) ? s.roughnessMapID : -1;
        m[i].metal_tex = (s.metallicMapID >= 0 && s.metallicMapID < T) ? s.metallicMapID : -1;
        m[i].normal_tex = (s.normalMapID >= 0 && s.normalMapID < T) ? s.normalMapID : -1;
    }
    const int B = scene.bvhSize;
    std::vector<DevTriE> te(scene.triangles.size());
    std::vector<DevTriV> tv(scene.triangles.size());
    std::vector<DevTriA> ta(scene.triangles.size());
    for (size_t i = 0; i < tv.size(); ++i) {
        const Triangle& t = scene.triangles[i];
        const ptk::vec3* v = t.v; const ptk::vec3* n = t.n;
        tv[i] = DevTriV{{v[0].x, v[0].y, v[0].z}, {v[1].x, v[1].y, v[1].z}, {v[2].x, v[2].y, v[2].z}};
        te[i] = DevTriE{{v[0].x, v[0].y, v[0].z}, {v[1].x - v[0].x, v[1].y - v[0].y, v[1].z - v[0].z},
                        {v[2].x - v[0].x, v[2].y - v[0].y, v[2].z - v[0].z}, {0.f, 0.f, 0.f}};
        ta[i] = DevTriA{{n[0].x, n[0].y, n[0].z}, {n[1].x, n[1].y, n[1].z}, {n[2].x, n[2].y, n[2].z},
                        {t.tex[0].x, t.tex[0].y}, {t.tex[1].x, t.tex[1].y}, {t.tex[2].x, t.tex[2].y}, t.geomIdx,
                        {t.tangent.x, t.tangent.y, t.tangent.z}};
    }
    std::vector<DevLight> li;
    for (const lightPrim& l : scene.lights) li.push_back(DevLight{l.geomID, l.triangleID, (int)l.type});
    std::vector<DevTex> tx;                                  // devTexObj table + one packed texel array (scene.cpp:575-595)
    std::vector<float> texels;
    for (const image* im : scene.textures) {
        tx.push_back(DevTex{im->width, im->height, (unsigned)(texels.size() / 3), 0});
        const float* src = reinterpret_cast<const float*>(im->data());
        texels.insert(texels.end(), src, src + 3 * (size_t)im->width * im->height);
    }

    int rc;
    if ((rc = upload(&dev_geoms, g))) return rc;
    if ((rc = upload(&dev_gbox, gb))) return rc;
    if ((rc = upload(&dev_gmask, gm))) return rc;
    if ((rc = upload(&dev_materials, m))) return rc;
    if ((rc = upload(&dev_trie, te))) return rc;
    wide_nodes = wide_depth = wide_stack = 0;
    if (!scene.triangles.empty()) {                          // 8-wide collapsed form of the same tree (wide_bvh.h)
        WideBvh wb;
        const char* ww = getenv("PTK_WIDE_WIDTH");
        wide_width = (ww && atoi(ww) == 16) ? 16 : 8;
        if (build_wide_bvh(scene.gpuBVHNodes, B, scene.triangles, wb, wide_width)) {
            if ((rc = wide_width == 16 ? upload(&dev_wnodes, wb.nodes16) : upload(&dev_wnodes, wb.nodes))) return rc;
            if ((rc = upload(&dev_wleaves, wb.leaves))) return rc;
            wide_nodes = (int)(wb.nodes.size() + wb.nodes16.size()); wide_depth = wb.depth; wide_stack = wb.max_stack;
        }
    }
    tri_num = (int)tv.size(); bvh_size = B;
    // Threaded copies (ptk_device.hpp, "ONE COPY INSTEAD OF SIX"): meshes the render loop walks inside the fused kernel (<= 1 024 triangles, where the
    // nearer-first order is the walk's speed) and meshes without a wide tree keep all six; a large mesh gets the direction-0 copy + the rank table.
    // PTK_THREADED_COPIES=1|6 overrides (tests run both); ptk_set_pipeline(5) / the traversal counters re-upload six (upload_threaded).
    // ONE COPY needs more than "large mesh" (ADVICE r4): walking in direction 0's order with the rank tie rule equals the reference's direction-ordered
    // walk only as long as no leaf is cut (`tEnter > t_min`, pathtrace.cu:253) in one order and tested in the other.  tEnter <= t holds for a
    // triangle's own box in exact arithmetic; in fp32 it can fail by an ulp where the hit point lies ON a face of the box — for a triangle with an
    // axis-parallel edge (two vertices sharing a coordinate: the edge lies in that face; a flat box is the extreme case) that is every hit on that
    // edge, and a hit on a SHARED edge is also where two triangles tie or nearly tie.  Meshes made of such triangles (tessellated boxes,
    // architecture) keep six copies.  A scanned mesh has a few by coincidence of its rounded coordinates (bunny10k: 4 of 10 000): it keeps one copy
    // — the residue is a ray of the k_tail / fallback walks that crosses one of those edges to the last bit (~1e-7 of the rays that cross it) AND
    // finds the two triangles within an ulp of each other.
    int axis_parallel = 0;
    for (const Triangle& t : scene.triangles) {
        bool hit = false;
        for (int a = 0; a < 3 && !hit; ++a)
            for (int b = a + 1; b < 3 && !hit; ++b)
                hit = t.v[a].x == t.v[b].x || t.v[a].y == t.v[b].y || t.v[a].z == t.v[b].z;
        axis_parallel += hit;
    }
    const bool axis_parallel_edge = axis_parallel > std::max(8, tri_num / 1000);
    int copies = (wide_nodes > 0 && tri_num > 1024 && !axis_parallel_edge) ? 1 : 6;
    if (const char* e = getenv("PTK_THREADED_COPIES")) { const int c = atoi(e); if (c == 6 || (c == 1 && tri_num > 0)) copies = c; }
    if ((rc = upload_threaded(scene, copies))) return rc;
    if ((rc = upload(&dev_triv, tv))) return rc;
    if ((rc = upload(&dev_tria, ta))) return rc;
    if ((rc = upload(&dev_lights, li))) return rc;
    if ((rc = upload(&dev_textures, tx))) return rc;
    if ((rc = upload(&dev_texture_data, texels))) return rc;
    tex_num = (int)tx.size(); envMapID = scene.envMapID;
    uploaded = true;
    return 0;
}

int DevScene::upload_threaded(Scene& scene, int copies) {
    if (copies != 1 && copies != 6) return PTK_E_BADARG;
    if (n_copies == copies && dev_nodes) return 0;
    const int B = scene.bvhSize, T = (int)scene.triangles.size();
    std::vector<DevNode> nodes((size_t)copies * B);
    for (size_t i = 0; i < nodes.size(); ++i) {
        const GpuBVHNode& n = scene.gpuBVHNodes[i];
        DevNode o{n.pMin.x, n.pMin.y, n.pMin.z, n.pMax.x, n.pMax.y, n.pMax.z, 0, 0};
        if (n.end - n.start <= 1) {                 // leaf (MAX_PRIM = 1): hit link == miss link by construction (BVH.cpp:200-236)
            if (n.hit != n.miss) return PTK_E_BADARG;
            o.a = n.end > n.start ? -(n.start + 2) : -1;
            o.b = n.hit;
        } else { o.a = n.hit; o.b = n.miss; }
        nodes[i] = o;
    }
    std::vector<uint32_t> ranks;
    if (copies == 1) {
        // position of every triangle in each direction's leaf order: follow the hit links of that direction's copy (they visit every node)
        ranks.assign((size_t)6 * T, 0u);
        for (int d = 0; d < 6; ++d) {
            uint32_t r = 0; int i = 0, guard = 0;
            while (i != -1 && guard++ <= B) {
                const GpuBVHNode& n = scene.gpuBVHNodes[(size_t)d * B + i];
                if (n.end - n.start == 1 && n.start >= 0 && n.start < T) ranks[(size_t)d * T + n.start] = r++;
                i = n.hit;
            }
            if ((int)r != T) return upload_threaded(scene, 6);      // not a tree with one triangle per leaf: keep six copies instead
        }
    }
    void* new_nodes = nullptr; void* new_ranks = nullptr;
    int rc;
    if ((rc = upload(&new_nodes, nodes))) return rc;
    if (copies == 1 && (rc = upload(&new_ranks, ranks))) { (void)hipFree(new_nodes); return rc; }
    if (dev_nodes || dev_ranks) (void)hipDeviceSynchronize();      // a re-upload (pipeline 5 / traversal counters): nothing may still be reading the old tables
    if (dev_nodes) (void)hipFree(dev_nodes);
    if (dev_ranks) (void)hipFree(dev_ranks);
    dev_nodes = new_nodes; dev_ranks = new_ranks; n_copies = copies;
    return 0;
}

void DevScene::destroy() {
    void** ptrs[] = {&dev_geoms, &dev_gbox, &dev_gmask, &dev_materials, &dev_nodes, &dev_ranks, &dev_trie, &dev_wnodes, &dev_wleaves, &dev_triv, &dev_tria, &dev_lights, &dev_textures, &dev_texture_data};
    for (void** p : ptrs) { if (*p) (void)hipFree(*p); *p = nullptr; }
    wide_nodes = 0; n_copies = 0;
    uploaded = false;
}

// ------------------------------------------------------------------------------------------------------------------
extern "C" {

int ptk_scene_load(const char* file, int res_w, int res_h, int depth, PtkScene** out) {
    if (!file || !out) return PTK_E_BADARG;
    PtkScene* h = new PtkScene();
    h->scene = new Scene(file, res_w, res_h, depth);
    *out = h;
    if (!h->scene->ok) return PTK_E_PARSE;
    h->scene->applyViewerCamera();
    h->scene->buildHostData();
    return 0;
}

void ptk_scene_free(PtkScene* s) {
    if (!s) return;
    if (s->owns && s->scene) { s->scene->clear(); delete s->scene; }
    delete s;
}

const char* ptk_scene_error(const PtkScene* s) { return (s && s->scene) ? s->scene->error.c_str() : "no scene"; }

int ptk_scene_counts(const PtkScene* s, int* c) {
    if (!s || !s->scene || !c) return PTK_E_BADARG;
    const Scene& S = *s->scene;
    c[0] = (int)S.geoms.size(); c[1] = (int)S.materials.size(); c[2] = (int)S.triangles.size(); c[3] = S.bvhSize;
    c[4] = (int)S.lights.size(); c[5] = S.state.traceDepth; c[6] = (int)S.state.iterations;
    c[7] = S.state.camera.resolution.x; c[8] = S.state.camera.resolution.y;
    c[9] = (int)S.textures.size(); c[10] = S.envMapID;
    return 0;
}

int ptk_scene_get_camera(const PtkScene* s, PtkCamera* o) {
    if (!s || !s->scene || !o) return PTK_E_BADARG;
    const Camera& c = s->scene->state.camera;
    o->res[0] = c.resolution.x; o->res[1] = c.resolution.y;
    const ptk::vec3* src[5] = {&c.position, &c.lookAt, &c.view, &c.up, &c.right};
    float* dst[5] = {o->position, o->lookAt, o->view, o->up, o->right};
    for (int k = 0; k < 5; ++k) { dst[k][0] = src[k]->x; dst[k][1] = src[k]->y; dst[k][2] = src[k]->z; }
    o->fov[0] = c.fov.x; o->fov[1] = c.fov.y; o->pixelLength[0] = c.pixelLength.x; o->pixelLength[1] = c.pixelLength.y;
    return 0;
}

int ptk_scene_get_geoms(const PtkScene* s, PtkGeom* o) {
    if (!s || !s->scene || !o) return PTK_E_BADARG;
    for (size_t i = 0; i < s->scene->geoms.size(); ++i) {
        const Geom& g = s->scene->geoms[i];
        o[i].type = (int)g.type; o[i].materialid = g.materialid;
        std::memcpy(o[i].transform, g.transform.m, 64);
        std::memcpy(o[i].inverseTransform, g.inverseTransform.m, 64);
        std::memcpy(o[i].invTranspose, g.invTranspose.m, 64);
    }
    return 0;
}

int ptk_scene_get_materials(const PtkScene* s, PtkMaterial* o) {
    if (!s || !s->scene || !o) return PTK_E_BADARG;
    for (size_t i = 0; i < s->scene->materials.size(); ++i) {
        const Material& m = s->scene->materials[i];
        o[i].type = (int)m.type;
        o[i].albedo[0] = m.albedo.x; o[i].albedo[1] = m.albedo.y; o[i].albedo[2] = m.albedo.z;
        o[i].albedo_s[0] = m.albedoSampler.x; o[i].albedo_s[1] = m.albedoSampler.y; o[i].albedo_s[2] = m.albedoSampler.z;
        o[i].rough_s = m.roughnessSampler; o[i].metal_s = m.metallicSampler; o[i].ior = m.ior;
        o[i].albedoMapID = m.albedoMapID; o[i].metallicMapID = m.metallicMapID; o[i].roughnessMapID = m.roughnessMapID; o[i].normalMapID = m.normalMapID;
    }
    return 0;
}

int ptk_scene_get_texture_size(const PtkScene* s, int id, int* width, int* height) {
    if (!s || !s->scene || !width || !height || id < 0 || id >= (int)s->scene->textures.size()) return PTK_E_BADARG;
    *width = s->scene->textures[id]->width; *height = s->scene->textures[id]->height;
    return 0;
}
int ptk_scene_get_texture(const PtkScene* s, int id, float* rgb_out) {
    if (!s || !s->scene || !rgb_out || id < 0 || id >= (int)s->scene->textures.size()) return PTK_E_BADARG;
    const image* im = s->scene->textures[id];
    std::memcpy(rgb_out, im->data(), im->byteSize());
    return 0;
}

int ptk_scene_get_triangles(const PtkScene* s, PtkTriangle* o) {
    if (!s || !s->scene || !o) return PTK_E_BADARG;
    for (size_t i = 0; i < s->scene->triangles.size(); ++i) {
        const Triangle& t = s->scene->triangles[i];
        for (int k = 0; k < 3; ++k) {
            o[i].v[3 * k] = t.v[k].x; o[i].v[3 * k + 1] = t.v[k].y; o[i].v[3 * k + 2] = t.v[k].z;
            o[i].n[3 * k] = t.n[k].x; o[i].n[3 * k + 1] = t.n[k].y; o[i].n[3 * k + 2] = t.n[k].z;
            o[i].tex[2 * k] = t.tex[k].x; o[i].tex[2 * k + 1] = t.tex[k].y;
        }
        o[i].tangent[0] = t.tangent.x; o[i].tangent[1] = t.tangent.y; o[i].tangent[2] = t.tangent.z;
        o[i].bitangent[0] = t.bitangent.x; o[i].bitangent[1] = t.bitangent.y; o[i].bitangent[2] = t.bitangent.z;
        o[i].geomIdx = t.geomIdx;
    }
    return 0;
}

int ptk_scene_get_bvh(const PtkScene* s, PtkBvhNode* o) {
    if (!s || !s->scene || !o) return PTK_E_BADARG;
    for (size_t i = 0; i < s->scene->gpuBVHNodes.size(); ++i) {
        const GpuBVHNode& n = s->scene->gpuBVHNodes[i];
        o[i].pMin[0] = n.pMin.x; o[i].pMin[1] = n.pMin.y; o[i].pMin[2] = n.pMin.z;
        o[i].pMax[0] = n.pMax.x; o[i].pMax[1] = n.pMax.y; o[i].pMax[2] = n.pMax.z;
        o[i].start = n.start; o[i].end = n.end; o[i].miss = n.miss; o[i].hit = n.hit;
    }
    return 0;
}

int ptk_scene_wide_bvh_counts_w(const PtkScene* s, int width, int* c) {
    if (!s || !s->scene || !c || (width != 8 && width != 16)) return PTK_E_BADARG;
    c[0] = c[1] = c[2] = c[3] = 0;
    WideBvh wb;
    if (!s->scene->triangles.empty() && build_wide_bvh(s->scene->gpuBVHNodes, s->scene->bvhSize, s->scene->triangles, wb, width)) {
        c[0] = (int)(wb.nodes.size() + wb.nodes16.size()); c[1] = (int)wb.leaves.size(); c[2] = wb.depth; c[3] = wb.max_stack;
    }
    return 0;
}
int ptk_scene_get_wide_bvh_w(const PtkScene* s, int width, void* nodes, void* leaves) {
    if (!s || !s->scene || !nodes || !leaves || (width != 8 && width != 16)) return PTK_E_BADARG;
    WideBvh wb;
    if (s->scene->triangles.empty() || !build_wide_bvh(s->scene->gpuBVHNodes, s->scene->bvhSize, s->scene->triangles, wb, width)) return PTK_E_BADARG;
    if (width == 16) std::memcpy(nodes, wb.nodes16.data(), wb.nodes16.size() * sizeof(WideNode16Host));
    else std::memcpy(nodes, wb.nodes.data(), wb.nodes.size() * sizeof(WideNodeHost));
    std::memcpy(leaves, wb.leaves.data(), wb.leaves.size() * sizeof(WideLeafHost));
    return 0;
}
int ptk_scene_wide_bvh_counts(const PtkScene* s, int* c) { return ptk_scene_wide_bvh_counts_w(s, 8, c); }
int ptk_scene_get_wide_bvh(const PtkScene* s, void* nodes, void* leaves) { return ptk_scene_get_wide_bvh_w(s, 8, nodes, leaves); }

int ptk_scene_bvh_device_bytes(const PtkScene* s, unsigned long long* out4) {
    if (!s || !s->scene || !out4) return PTK_E_BADARG;
    const DevScene& d = s->scene->tempDevScene;
    if (!d.uploaded) return PTK_E_NOSCENE;
    out4[0] = (unsigned long long)d.n_copies * d.bvh_size * sizeof(DevNode);
    out4[1] = (unsigned long long)d.wide_nodes * 144ull;
    out4[2] = d.dev_wleaves ? (unsigned long long)d.tri_num * 64ull : 0ull;
    out4[3] = d.dev_ranks ? 6ull * d.tri_num * sizeof(uint32_t) : 0ull;
    return 0;
}

int ptk_scene_get_lights(const PtkScene* s, PtkLight* o) {
    if (!s || !s->scene || !o) return PTK_E_BADARG;
    for (size_t i = 0; i < s->scene->lights.size(); ++i) {
        const lightPrim& l = s->scene->lights[i];
        o[i].geomID = l.geomID; o[i].triangleID = l.triangleID; o[i].type = (int)l.type;
    }
    return 0;
}

int ptk_scene_upload(PtkScene* s) {
    if (!s || !s->scene || !s->scene->ok) return PTK_E_NOSCENE;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return PTK_E_NODEVICE;
    s->scene->setDevData();
    return s->scene->devError;
}

} // extern "C"
