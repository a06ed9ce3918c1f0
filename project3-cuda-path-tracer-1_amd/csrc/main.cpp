// main.cpp — headless driver that replays the call sequence of the reference's viewer (src/main.cpp:48-227,
// src/preview.cpp:264-296) without GLFW/GL/ImGui:
//   Scene(file) -> camera orbit fix-ups -> InitDataContainer -> setDevData -> loop { runCuda() } -> saveImage
// runCuda(): at iteration 0 pathtraceFree(); pathtraceInit(scene); then iteration++ and pathtrace(pbo, 0, iteration)
// (1-based) until `iterations`, then saveImage(): mean -> ACES -> gamma -> x-flipped 8-bit image.
// Output: 8-bit RGB PNG like image::savePNG (image.cpp:56-73; written here with zlib instead of stb_image_write),
// binary PPM (P6) and the raw float32 radiance sum (.f32).
//
// usage: ptk_headless SCENE.txt [--mode bsdf|direct|mis] [--iters N] [--res W H] [--depth D] [--out BASE]
//        [--warmup W] [--no-save] [--in-flight L] [--tiles K] [--graph 0|1] [--eager] [--reinit-at N]
//        [--ranks N --rank R --id-file PATH]      (multi-GPU: see "RCCL" below)
// The loop below is the reference's, call for call: it uses the four reference signatures only and never calls pathtraceResolve() — the library
// makes scene->state.image current where this loop reads it (round 5: after the call with iteration == state.iterations, and in pathtraceFree;
// pathtrace.hip).  --iters overrides the scene file's ITERATIONS (state.iterations), as editing the file would.
// Timing (bench.py `cpp_api`): --warmup W renders W untimed iterations first (then waits for the GPU); the timed region is the remaining
// pathtrace() calls including the resolve the last one makes — what a viewer that saves the image at the end pays.  --eager = PTK_EAGER_IMAGE=1
// (the image copied after every call, pathtrace.cu:809-810), --in-flight / --tiles / --graph are the ptk_set_* knobs; without them pathtraceInit
// chooses three iterations in flight + graph replay with ordered accumulation in every sample mode (image bit-identical to one at a time).
// --reinit-at N: pathtraceFree() + pathtraceInit() again after iteration N and start over (what the viewer does on a camera / mode change).
#include "pathtrace.h"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <zlib.h>

SampleMode sampleMode = MIS;        // main.cpp:20

// minimal PNG encoder: 8-bit RGB, filter 0 on every scanline, one zlib stream in one IDAT chunk
static bool writePNG(const std::string& path, int W, int H, const unsigned char* rgb) {
    std::vector<unsigned char> raw((size_t)H * (3 * (size_t)W + 1));
    for (int y = 0; y < H; ++y) {
        raw[(size_t)y * (3 * W + 1)] = 0;
        std::memcpy(&raw[(size_t)y * (3 * W + 1) + 1], rgb + (size_t)y * 3 * W, (size_t)3 * W);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    auto be32 = [](unsigned v, unsigned char* o) { o[0] = v >> 24; o[1] = v >> 16; o[2] = v >> 8; o[3] = v; };
    auto chunk = [&](const char* tag, const unsigned char* data, unsigned len) {
        unsigned char hdr[8]; be32(len, hdr); std::memcpy(hdr + 4, tag, 4);
        std::fwrite(hdr, 1, 8, f);
        if (len) std::fwrite(data, 1, len, f);
        uLong c = crc32(0L, (const Bytef*)tag, 4);
        if (len) c = crc32(c, data, len);
        unsigned char crc[4]; be32((unsigned)c, crc); std::fwrite(crc, 1, 4, f);
    };
    const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::fwrite(sig, 1, 8, f);
    unsigned char ihdr[13]; be32((unsigned)W, ihdr); be32((unsigned)H, ihdr + 4);
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;      // 8 bit, colour type 2 (RGB)
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), (unsigned)zlen);
    chunk("IEND", nullptr, 0);
    std::fclose(f);
    return true;
}

static float aces(float x) {        // utilities.h:35-43
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    float y = (x * (a * x + b)) / (x * (c * x + d) + e);
    return y < 0.f ? 0.f : (y > 1.f ? 1.f : y);
}

static void saveImage(const Scene& scene, int iteration, const std::string& base) {   // main.cpp:146-170
    const int W = scene.state.camera.resolution.x, H = scene.state.camera.resolution.y;
    const float samples = (float)iteration;
    std::vector<unsigned char> bytes((size_t)3 * W * H);
    for (int x = 0; x < W; x++)
        for (int y = 0; y < H; y++) {
            const ptk::vec3& s = scene.state.image[x + y * W];
            const float pix[3] = {s.x / samples, s.y / samples, s.z / samples};
            unsigned char* o = &bytes[3 * ((size_t)y * W + (W - 1 - x))];               // setPixel(width-1-x, y, ...)
            for (int k = 0; k < 3; ++k) {
                float v = std::pow(aces(pix[k]), 1 / 2.2f);
                v = (v < 0.f ? 0.f : (v > 1.f ? 1.f : v)) * 255.f;                       // image::savePNG, image.cpp:56-73
                o[k] = (unsigned char)v;
            }
        }
    const std::string png = base + "." + std::to_string(iteration) + "samp.png";
    if (writePNG(png, W, H, bytes.data())) std::printf("Saved %s.\n", png.c_str());
    const std::string ppm = base + "." + std::to_string(iteration) + "samp.ppm";
    if (FILE* f = std::fopen(ppm.c_str(), "wb")) {
        std::fprintf(f, "P6\n%d %d\n255\n", W, H);
        std::fwrite(bytes.data(), 1, bytes.size(), f);
        std::fclose(f);
        std::printf("Saved %s.\n", ppm.c_str());
    }
    const std::string raw = base + "." + std::to_string(iteration) + "samp.f32";
    if (FILE* f = std::fopen(raw.c_str(), "wb")) {
        std::fwrite(scene.state.image.data(), sizeof(ptk::vec3), scene.state.image.size(), f);
        std::fclose(f);
    }
}

int main(int argc, char** argv) {
    if (argc < 2) { std::printf("Usage: %s SCENEFILE.txt [--mode bsdf|direct|mis] [--iters N] [--res W H] [--depth D] [--out BASE]\n", argv[0]); return 1; }
    int iters = -1, rw = -1, rh = -1, depth = -1, warmup = 0, in_flight = 0, graph = -1, tiles = 0, reinit_at = 0;
    bool save = true, eager = false;
    std::string out;
    for (int i = 2; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--mode") && i + 1 < argc) { const char* m = argv[++i]; sampleMode = !std::strcmp(m, "bsdf") ? BSDF : (!std::strcmp(m, "direct") ? DirectLi : MIS); }
        else if (!std::strcmp(argv[i], "--iters") && i + 1 < argc) iters = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--res") && i + 2 < argc) { rw = std::atoi(argv[++i]); rh = std::atoi(argv[++i]); }
        else if (!std::strcmp(argv[i], "--depth") && i + 1 < argc) depth = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--out") && i + 1 < argc) out = argv[++i];
        else if (!std::strcmp(argv[i], "--light-fix")) ptk_set_light_fix(1);        // opt-in fixed cube lights (not reference behaviour)
        else if (!std::strcmp(argv[i], "--warmup") && i + 1 < argc) warmup = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--in-flight") && i + 1 < argc) in_flight = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--graph") && i + 1 < argc) graph = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--tiles") && i + 1 < argc) tiles = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--no-save")) save = false;
        else if (!std::strcmp(argv[i], "--eager")) eager = true;
        else if (!std::strcmp(argv[i], "--reinit-at") && i + 1 < argc) reinit_at = std::atoi(argv[++i]);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { std::printf("No HIP-capable devices were detected.\n"); return 2; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess)
        std::printf("Device 0: \"%s\" %s, %d CUs, core clock %d MHz\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);

    Scene* scene = new Scene(argv[1], rw, rh, depth);
    if (!scene->ok) { std::printf("%s\n", scene->error.c_str()); return 3; }
    GuiDataContainer* guiData = new GuiDataContainer();
    scene->applyViewerCamera();                          // main.cpp:102-127 + runCuda's camchanged block
    InitDataContainer(guiData);
    scene->setDevData();
    if (scene->devError) { std::printf("scene upload failed: %d\n", scene->devError); return 4; }
    if (iters > 0) scene->state.iterations = iters;     // as if the scene file said ITERATIONS <iters>
    const int total = (int)scene->state.iterations;
    if (eager) setenv("PTK_EAGER_IMAGE", "1", 1);
    if (out.empty()) out = scene->state.imageName.empty() ? std::string("render") : scene->state.imageName;

    if (in_flight > 0 && ptk_set_iterations_in_flight(in_flight)) { std::printf("bad --in-flight\n"); return 5; }
    if (tiles > 0 && ptk_set_tiles_in_flight(tiles)) { std::printf("bad --tiles\n"); return 5; }
    if (graph >= 0) ptk_set_graph_replay(graph);
    int iteration = 0;
    auto t0 = std::chrono::steady_clock::now();
    unsigned long long rays0 = 0;
    while (true) {                                       // mainLoop -> runCuda
        if (iteration == 0) { pathtraceFree(); pathtraceInit(scene); }
        if (iteration < total) {
            iteration++; pathtrace(nullptr, 0, iteration);
            if (iteration == warmup) { ptk_stream_sync(); ptk_get_ray_total(&rays0); t0 = std::chrono::steady_clock::now(); }
            if (iteration == reinit_at) { reinit_at = 0; iteration = 0; }      // the viewer's camchanged: runCuda() starts over (main.cpp:173-201)
        }
        else break;                                      // main.cpp:221-225: saveImage(); pathtraceFree(); exit
    }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    unsigned long long rays = 0;
    ptk_get_ray_total(&rays);
    const int timed = iteration - (warmup < iteration ? warmup : 0);
    std::printf("%d iterations in %.3f s (%.2f ms/iteration), TracedDepth %d\n", iteration, sec, 1e3 * sec / std::max(1, timed), guiData->TracedDepth);
    std::printf("cpp_api {\"iterations\": %d, \"warmup\": %d, \"seconds\": %.6f, \"rays\": %llu, \"eager\": %d, \"in_flight\": %d, \"graph\": %d}\n",
                timed, warmup < iteration ? warmup : 0, sec, rays - rays0, eager ? 1 : 0, in_flight > 0 ? in_flight : 1, graph > 0 ? 1 : 0);
    if (save) saveImage(*scene, iteration, out);
    pathtraceFree();
    Resource::clear();
    scene->clear();
    delete scene;
    delete guiData;
    return 0;
}
