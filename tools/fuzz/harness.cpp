// ASan/UBSan harness for the host scene layer: load a scene file (parser, OBJ reader, textures, BVH build, host tables)
#include "scene.h"
#include <cstdio>
int DevScene::initiate(Scene&) { return 0; }
void DevScene::destroy() {}
int main(int argc, char** argv) {
    for (int i = 1; i < argc; ++i) {
        Scene s(argv[i], -1, -1, -1);
        if (!s.ok) { std::printf("%s: rejected: %s\n", argv[i], s.error.c_str()); continue; }
        s.applyViewerCamera();
        s.buildHostData();
        std::printf("%s: ok geoms %zu tris %zu bvh %d lights %zu tex %zu\n", argv[i], s.geoms.size(), s.triangles.size(), s.bvhSize, s.lights.size(), s.textures.size());
    }
    Resource::clear();
    return 0;
}
