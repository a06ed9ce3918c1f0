// scene.h — host scene layer: the mirror of the reference's src/scene.h, src/sceneStructs.h, src/BVH.h,
// src/material.h (data members) and src/lightSample.h (lightPrim) for the path-tracing hot path.
//
// Same class and member names as the reference where main.cpp / preview.cpp touch them
// (Scene::state, geoms, materials, triangles, gpuBVHNodes, lights, setDevData(), clear(), Resource::clear()),
// own small vector types instead of GLM (ptk::vec3 zero-initialises like glm 0.9.6.3's default ctor).
// The device image (DevScene) is SoA / split-by-use and NOT the reference's AoS structs — see ptk_device.hpp.
#pragma once
#include <map>
#include <string>
#include <vector>
#include <istream>
#include "../../include/ptk.h"

namespace ptk {
struct vec2 { float x = 0.f, y = 0.f; vec2() {} vec2(float a, float b) : x(a), y(b) {} };
struct ivec2 { int x = 0, y = 0; };
struct vec3 {
    float x = 0.f, y = 0.f, z = 0.f;
    vec3() {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    float get(int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
struct mat4 {                 // column-major like glm::mat4: col(c)[r] = m[4*c + r]
    float m[16];
    static mat4 identity();
    float at(int c, int r) const { return m[4 * c + r]; }
    float& at(int c, int r) { return m[4 * c + r]; }
};
struct DevSceneView;          // ptk_device.hpp's DevScene (device pointers), opaque here
} // namespace ptk

enum GeomType { SPHERE, CUBE, OBJ };                          // sceneStructs.h:11-15
enum SampleMode { BSDF, DirectLi, MIS };                      // preview.h:13

class MeshData;

struct Geom {                                                 // sceneStructs.h:35-47
    GeomType type = SPHERE;
    int materialid = 0;
    ptk::vec3 translation, rotation, scale;
    ptk::mat4 transform, inverseTransform, invTranspose;
    MeshData* mesh = nullptr;
};

struct Camera {                                               // sceneStructs.h:61-70
    ptk::ivec2 resolution;
    ptk::vec3 position, lookAt, view, up, right;
    ptk::vec2 fov, pixelLength;
};

struct RenderState {                                          // sceneStructs.h:72-78
    Camera camera;
    unsigned int iterations = 0;
    int traceDepth = 0;
    std::vector<ptk::vec3> image;
    std::string imageName;
};

class Material {                                              // material.h:28-38, 524-538
public:
    enum Type { Lambertian, MetallicWorkflow, Dielectric, Microfacet, Light };
    Type type = Lambertian;                                   // no default in the reference; 0 in practice
    ptk::vec3 albedo = ptk::vec3(1.f);
    float roughness = .0f, metallic = .0f, ior = 1.5f;
    int albedoMapID = -1, metallicMapID = -1, roughnessMapID = -1, normalMapID = -1;     // index into Scene::textures
    // devTexSampler::fixVal of the samplers (image.h:88-110), used when the corresponding MapID is < 0
    ptk::vec3 albedoSampler;          // default devTexSampler(): fixVal(0)
    float roughnessSampler = 0.f, metallicSampler = 0.f;
};

class Triangle {                                              // Bounds3.hpp:250-258
public:
    ptk::vec3 v[3], n[3];
    ptk::vec2 tex[3];
    ptk::vec3 tangent, bitangent;
    int geomIdx = -1;
};

struct GpuBVHNode {                                           // BVH.h:44-50
    ptk::vec3 pMin, pMax;
    int start = 0, end = 0, miss = -1, hit = -1;
};

class lightPrim {                                             // lightSample.h:26-34
public:
    int geomID, triangleID;
    GeomType type;
    lightPrim(int g, int t, GeomType ty) : geomID(g), triangleID(t), type(ty) {}
};

class MeshData { public: std::vector<Triangle> triangles; };  // scene.h:101-107
class image;                                                  // image.h

class Scene;

class DevScene {                                              // scene.h:28-52 — device image of the scene tables
public:
    int initiate(Scene& scene);                               // returns 0 or a hipError_t
    int upload_threaded(Scene& scene, int copies);            // (re)uploads 1 or 6 direction-ordered copies of the threaded BVH (+ the leaf ranks with 1)
    void destroy();
    int tri_num = 0, bvh_size = 0;
    void* dev_geoms = nullptr;        // ptk::DevGeom[]
    void* dev_gbox = nullptr;         // ptk::DevGeomBox[]
    void* dev_gmask = nullptr;        // ptk::DevGeomMask[]
    void* dev_materials = nullptr;    // ptk::DevMaterial[]
    void* dev_nodes = nullptr;        // ptk::DevNode[n_copies][bvh_size]: the threaded BVH, all six direction orders or direction 0 only
    void* dev_ranks = nullptr;        // uint32[6][tri_num] with ONE copy: position of every triangle in each direction's leaf order (the tie rule, ptk_device.hpp)
    int n_copies = 0;
    void* dev_trie = nullptr;         // ptk::DevTriE[]
    void* dev_wnodes = nullptr;       // 8-wide collapsed tree (wide_bvh.h), 144 B = 9 x uint4 per node; null if the tree could not be collapsed
    void* dev_wleaves = nullptr;      // its leaf records, 64 B each
    int wide_nodes = 0, wide_depth = 0, wide_stack = 0;
    int wide_width = 8;               // children per wide node: 8, or 16 with PTK_WIDE_WIDTH=16 (272-B nodes)
    void* dev_triv = nullptr;         // ptk::DevTriV[]
    void* dev_tria = nullptr;         // ptk::DevTriA[]
    void* dev_lights = nullptr;       // ptk::DevLight[]
    void* dev_textures = nullptr;     // ptk::DevTex[]          (devTexObj, image.h:53-86)
    void* dev_texture_data = nullptr; // float[3 * total texels] (dev_texture_data, scene.cpp:575-590)
    int tex_num = 0, envMapID = -1;
    bool uploaded = false;
};

class Scene {                                                 // scene.h:54-85
public:
    // res_w/res_h/depth <= 0 keep the file's RES / DEPTH lines (harness overrides)
    explicit Scene(const std::string& filename, int res_w = -1, int res_h = -1, int depth = -1);
    ~Scene();
    void setDevData();                                        // scene.cpp:479-567
    void buildHostData();                                     // the host half of setDevData (triangles, BVH, lights)
    void clear();                                             // scene.cpp:444-448
    void applyViewerCamera();                                 // main.cpp:102-127 + 173-193 (camera fix-ups)

    bool ok = false;
    std::string error;
    int devError = 0;

    std::vector<Geom> geoms;
    std::map<std::string, int> geomNameMap;
    std::vector<Material> materials;
    std::map<std::string, int> materialNameMap;
    std::vector<Triangle> triangles;
    std::vector<lightPrim> lights;
    std::vector<image*> textures;                             // scene.h:72 (owned by Resource's texture pool)
    int envMapID = -1;                                        // scene.h:76
    std::vector<GpuBVHNode> gpuBVHNodes;                      // 6 * bvhSize (direction-ordered copies)
    int bvhSize = 0;
    DevScene tempDevScene;
    RenderState state;
    float theta = 0.f, phi = 0.f;                             // globals of main.cpp:24
    bool posInit = true;                                      // main.cpp:25
    bool hostDataBuilt = false;

private:
    int loadMaterial(std::istream& in, const std::string& name);
    int loadGeom(std::istream& in, const std::string& name);
    int loadCamera(std::istream& in);
    int loadTexture(const std::string& fileName, float gamma = 1.f);   // scene.cpp:318-337; -1 = not an image, -2 = undecodable
    std::string resolvePath(std::string p) const;
    std::string sceneDir;
    int ovrW, ovrH, ovrDepth;
};

namespace Resource {                                          // scene.h:88-98
    MeshData* loadObj(const std::string& filename, const int geomIdx, std::string* err);
    image* loadTexture(const std::string& filename, float gamma, std::string* err);   // scene.cpp:465-477 (texturePool)
    void clear();
}
