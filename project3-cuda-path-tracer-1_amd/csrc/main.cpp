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
// chooses four iterations in flight + graph replay with ordered accumulation in every sample mode (image bit-identical to one at a time).
// --reinit-at N: pathtraceFree() + pathtraceInit() again after iteration N and start over (what the viewer does on a camera / mode change).
//
// RCCL (round 5; north star: "independent pixel tiles shard across the GPUs of one node with an RCCL reduce of the accumulated radiance buffer over
// xGMI").  One process per GPU: `ptk_headless SCENE --ranks N --rank R --id-file PATH [--tile-rows T]`.  Rank 0 writes the ncclUniqueId
// to PATH (the launcher's only job is to start N processes with the same PATH), every rank joins the communicator (ncclCommInitRank), renders its
// row tiles t = R (mod N) (ptk_set_tile_partition: no collective on the data path), and ONE ncclReduce(sum, float32, 3 * W * H) of the device
// images — zero outside a rank's own tiles — puts the frame on rank 0, which saves it.  The clock (after a ncclAllReduce as the barrier) covers the
// pathtrace() calls and the reduce; rank 0 prints the whole job's rays (ncclReduce of the ranks' ray counters) in the cpp_rccl line.  A rank count
// above the visible GPUs is refused with one line and exit code 6: two ranks never share a device.
#include "pathtrace.h"
#include <rccl/rccl.h>
#include <unistd.h>
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

#define RCCL_CHECK(expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) { std::fprintf(stderr, "RCCL error (%s:%d): %s: %s\n", __FILE__, __LINE__, #expr, ncclGetErrorString(_r)); std::exit(EXIT_FAILURE); } } while (0)
#define HIP_CHECK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { std::fprintf(stderr, "HIP error (%s:%d): %s: %s\n", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); std::exit(EXIT_FAILURE); } } while (0)

// the ncclUniqueId through a file: rank 0 writes PATH.tmp and renames it (atomic), the others wait for PATH (at most 120 s); rank 0 removes PATH once the
// communicator exists.  PATH must not exist when the job starts (a file left by a job that died would be read as this job's id).
static void exchange_id(ncclUniqueId& id, int rank, const std::string& path) {
    if (rank == 0) {
        RCCL_CHECK(ncclGetUniqueId(&id));
        const std::string tmp = path + ".tmp";
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(&id, sizeof id, 1, f) != 1) { std::fprintf(stderr, "ptk_headless: cannot write %s\n", tmp.c_str()); std::exit(7); }
        std::fclose(f);
        if (std::rename(tmp.c_str(), path.c_str()) != 0) { std::fprintf(stderr, "ptk_headless: cannot rename %s\n", tmp.c_str()); std::exit(7); }
        return;
    }
    for (int tries = 0; tries < 2400; ++tries) {
        if (FILE* f = std::fopen(path.c_str(), "rb")) {
            const size_t n = std::fread(&id, sizeof id, 1, f);
            std::fclose(f);
            if (n == 1) return;
        }
        usleep(50000);
    }
    std::fprintf(stderr, "ptk_headless: rank %d: no ncclUniqueId in %s after 120 s\n", rank, path.c_str());
    std::exit(7);
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
    int ranks = 1, rank = 0, tile_rows = 8;
    std::string id_file;
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
        else if (!std::strcmp(argv[i], "--ranks") && i + 1 < argc) ranks = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--rank") && i + 1 < argc) rank = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--id-file") && i + 1 < argc) id_file = argv[++i];
        else if (!std::strcmp(argv[i], "--tile-rows") && i + 1 < argc) tile_rows = std::atoi(argv[++i]);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { std::printf("No HIP-capable devices were detected.\n"); return 2; }
    const bool multi = !id_file.empty() || ranks > 1;
    if (multi) {
        if (ranks < 1 || rank < 0 || rank >= ranks || tile_rows < 1 || id_file.empty()) { std::printf("usage: --ranks N --rank R (0 <= R < N) --id-file PATH [--tile-rows T]\n"); return 5; }
        if (ranks > ndev) { std::printf("ptk_headless: %d ranks need %d GPUs, %d visible (one process per GPU; two ranks never share a device)\n", ranks, ranks, ndev); return 6; }
        HIP_CHECK(hipSetDevice(rank));
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, multi ? rank : 0) == hipSuccess)
        std::printf("Device %d: \"%s\" %s, %d CUs, core clock %d MHz\n", multi ? rank : 0, prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
    ncclComm_t comm = nullptr;
    hipStream_t cst = nullptr;
    float* sync_word = nullptr;
    if (multi) {
        ncclUniqueId id;
        exchange_id(id, rank, id_file);
        RCCL_CHECK(ncclCommInitRank(&comm, ranks, id, rank));
        if (rank == 0) std::remove(id_file.c_str());      // every rank has read it (the init is collective): a later job that reuses PATH must not find this id
        HIP_CHECK(hipStreamCreateWithFlags(&cst, hipStreamNonBlocking));
        HIP_CHECK(hipMalloc((void**)&sync_word, sizeof(float)));
        HIP_CHECK(hipMemset(sync_word, 0, sizeof(float)));
        if (ptk_set_tile_partition(ranks, rank, tile_rows)) { std::printf("bad tile partition\n"); return 5; }
    }
    auto barrier = [&]() {          // every rank's GPU work so far is done, and every rank is here
        if (!multi) return;
        ptk_stream_sync();
        RCCL_CHECK(ncclAllReduce(sync_word, sync_word, 1, ncclFloat, ncclSum, comm, cst));
        HIP_CHECK(hipStreamSynchronize(cst));
    };

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
            if (iteration == warmup) { ptk_stream_sync(); barrier(); ptk_get_ray_total(&rays0); t0 = std::chrono::steady_clock::now(); }
            if (iteration == reinit_at) { reinit_at = 0; iteration = 0; }      // the viewer's camchanged: runCuda() starts over (main.cpp:173-201)
        }
        else break;                                      // main.cpp:221-225: saveImage(); pathtraceFree(); exit
    }
    unsigned long long rays = 0;
    ptk_get_ray_total(&rays);
    double reduce_ms = 0.0;
    if (multi) {
        // ONE reduce of the radiance sums (each rank's image is zero outside its own tiles): the frame lands on rank 0, straight from the render
        // context's device image — no host staging
        const size_t n = 3 * (size_t)scene->state.camera.resolution.x * scene->state.camera.resolution.y;
        float* img_dev = nullptr; float* frame = nullptr;
        if (ptk_get_image_device(&img_dev)) { std::printf("ptk_get_image_device failed\n"); return 8; }
        if (rank == 0) HIP_CHECK(hipMalloc((void**)&frame, n * sizeof(float)));
        unsigned long long* rays_dev = nullptr;
        HIP_CHECK(hipMalloc((void**)&rays_dev, 2 * sizeof(unsigned long long)));
        const unsigned long long mine = rays - rays0;
        HIP_CHECK(hipMemcpy(rays_dev, &mine, sizeof mine, hipMemcpyHostToDevice));
        const auto r0 = std::chrono::steady_clock::now();
        RCCL_CHECK(ncclReduce(img_dev, frame, n, ncclFloat, ncclSum, 0, comm, cst));
        RCCL_CHECK(ncclReduce(rays_dev, rays_dev + 1, 1, ncclUint64, ncclSum, 0, comm, cst));
        HIP_CHECK(hipStreamSynchronize(cst));
        reduce_ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - r0).count();
        if (rank == 0) {
            HIP_CHECK(hipMemcpy(scene->state.image.data(), frame, n * sizeof(float), hipMemcpyDeviceToHost));
            unsigned long long all = 0;
            HIP_CHECK(hipMemcpy(&all, rays_dev + 1, sizeof all, hipMemcpyDeviceToHost));
            rays = rays0 + all;
            HIP_CHECK(hipFree(frame));
        }
        HIP_CHECK(hipFree(rays_dev));
    }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const int timed = iteration - (warmup < iteration ? warmup : 0);
    std::printf("%d iterations in %.3f s (%.2f ms/iteration), TracedDepth %d\n", iteration, sec, 1e3 * sec / std::max(1, timed), guiData->TracedDepth);
    std::printf("cpp_api {\"iterations\": %d, \"warmup\": %d, \"seconds\": %.6f, \"rays\": %llu, \"eager\": %d, \"in_flight\": %d, \"graph\": %d}\n",
                timed, warmup < iteration ? warmup : 0, sec, rays - rays0, eager ? 1 : 0, in_flight > 0 ? in_flight : 1, graph > 0 ? 1 : 0);
    if (multi) std::printf("cpp_rccl {\"ranks\": %d, \"rank\": %d, \"tile_rows\": %d, \"iterations\": %d, \"seconds\": %.6f, \"rays\": %llu, \"reduce_ms\": %.4f}\n",
                           ranks, rank, tile_rows, timed, sec, rays - rays0, reduce_ms);
    if (save && rank == 0) saveImage(*scene, iteration, out);
    pathtraceFree();
    if (multi) {
        barrier();
        HIP_CHECK(hipFree(sync_word));
        RCCL_CHECK(ncclCommDestroy(comm));
        HIP_CHECK(hipStreamDestroy(cst));
        ptk_set_tile_partition(1, 0, 1);
    }
    Resource::clear();
    scene->clear();
    delete scene;
    delete guiData;
    return 0;
}
