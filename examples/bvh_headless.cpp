// bvh_headless — the reference's BVH and Pool demos (struct RayTrace + main of /root/reference/BVH/BVH.cpp:207-825 and
// Pool/Pool.cpp:459-759) without a window: the same scene assembly into RayTracing::Model, the same Transform camera, and the frame
// of RayTrace::init() / RayTrace::run() (BVH/BVH.cpp:712-743) with every OpenGL call replaced by the C ABI of include/lrl_trace.h,
// exactly as INTEGRATION.md lists them.  Instead of glfwSwapBuffers the last frame goes through lrl_quantize_rgba8 — the RGBA8,
// clamped default framebuffer the reference draws into (BVH/BVH.cpp:42) — and is written as a 24-bit BMP.
//
//   bvh_headless <resources dir> <out.bmp> [--pool] [--size W H] [--frames N] [--fast] [--device-bvh] [--water-steps n]
//                [--move] [--ppm out.ppm] [--pfm out.pfm]
// --move: Movement::run() before every frame (BVH/BVH.cpp:187-205, switched off at :726 in the shipped demo): two spheres move, the
// host builder makes the tree again (model.dataInit()), and instead of uploading everything the caller hands over the two records
// and the tree (lrl_scene_update_primitives + lrl_scene_update_bvh; with --device-bvh: + lrl_scene_rebuild_bvh).
//
// `lrl::` (raytracing_amd/host/lrl/*.hpp) restates the author's absent <RayTracing/_RayTracing.h>, <_STL.h>, <_BMP.h>, <_Math.h>.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "lrl/Image.hpp"
#include "lrl/RayTracing.hpp"
#include "lrl/Scenes.hpp"
#include "lrl_trace.h"

using namespace lrl;

#define LRL_CHECK(ctx, call)                                                                       \
    do {                                                                                           \
        int rc_ = (call);                                                                          \
        if (rc_ != LRL_OK) { std::fprintf(stderr, "%s: error %d: %s\n", #call, rc_, lrl_last_error(ctx)); std::exit(1); } \
    } while (0)
#define HIP_CHECK(call)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); std::exit(1); } \
    } while (0)

struct RayTrace {
    bool poolVariant, fast, deviceBVH;
    unsigned waterSteps;
    Model model;                                 // RayTracing::Model: the scene, assembled as the reference does
    Transform transform;                         // RayTracing::Transform: the camera
    FrameScale frameScale;
    LrlDecayOrigin decayOriginData{};            // RayTracing::DecayOriginData (SSBO 8)
    LrlContext* lrl = nullptr;
    float* frameDev = nullptr;                   // RGBA32F, row 0 = bottom, as gl_FragCoord
    uint8_t* rgba8Dev = nullptr;                 // the RGBA8 "default framebuffer"
    unsigned waterTriangles = 0;
    bool moving = false;                         // Movement movement (BVH.cpp:243)
    float moveT = 0.0f;

    RayTrace(std::string const& resources, bool pool, bool fastMode, bool devBVH, unsigned steps)
        : poolVariant(pool), fast(fastMode), deviceBVH(devBVH), waterSteps(steps),
          transform({{60.0}, {0.09, 0.9, 0.01}, {0.15}, pool ? Math::vec3<double>{0, 0, 1} : Math::vec3<double>{0, 0, 30}, 700.0}) {   // BVH.cpp:245, Pool.cpp:465
        scenes::Camera cam = pool ? scenes::pool(model, resources) : scenes::bvhDefault(model, resources);      // BVH.cpp:280-703, Pool.cpp:499-606
        (void)cam;
        if (pool) waterTriangles = 2u * (8 * 6 - 1) * (8 * 6 - 1);
    }

    void init(FrameScale const& size) {                                           // BVH/BVH.cpp:712-723
        frameScale = size;
        transform.init(size);
        LRL_CHECK(lrl, lrl_create(0, &lrl));
        if (deviceBVH) model.dataInitWithoutTree(); else model.dataInit();        // the CPU BVH build of model.dataInit() (:722), or none
        if (poolVariant) scenes::poolWaterSync(model);                            // waterSim.positionSync.run() (Pool.cpp:649-650)
        LrlSceneView v = model.sceneView();
        LRL_CHECK(lrl, lrl_scene_upload(lrl, &v));                                // uploads + TrianglePre + CirclePre (:159-168)
        if (poolVariant && waterSteps) {
            std::vector<LrlWaterPoint> pts = scenes::poolWaterPoints();          // WaterData's initial state (Pool.cpp:97-112)
            LRL_CHECK(lrl, lrl_water_init(lrl, 48, 48, 0.0005f, 0.05f, pts.data(), 0, v.trianglesOrigin));
        }
        HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&frameDev), size_t(size.w) * size.h * 16));
        HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&rgba8Dev), size_t(size.w) * size.h * 4));
    }

    void move() {                                                                 // Movement::run + sphereRun (BVH/BVH.cpp:187-205)
        moveT += 0.1f;
        model.spheres.data.spheres[0].sphere[1] += std::sin(moveT);
        model.spheres.data.spheres[1].sphere[1] += std::sin(moveT);
        model.spheres.upToDate = false;
        LRL_CHECK(lrl, lrl_scene_update_primitives(lrl, LRL_PRIM_SPHERE, 0, 2, model.spheres.data.spheres.data));
        if (deviceBVH) LRL_CHECK(lrl, lrl_scene_rebuild_bvh(lrl));
        else {
            model.dataInit();                                                     // the CPU BVH build for the new positions (:203)
            LrlSceneView v = model.sceneView();
            LRL_CHECK(lrl, lrl_scene_update_bvh(lrl, v.bvh, v.bvhNodeNum));
        }
        model.moved = true;
    }

    void run() {                                                                  // BVH/BVH.cpp:724-743
        if (moving) move();                                                       // //movement.run(); (:726)
        transform.operate();
        if (transform.updated) {                                                  // transUniform.refreshData() (:736)
            LRL_CHECK(lrl, lrl_set_camera(lrl, static_cast<LrlTrans const*>(transform.bufferData.pointer()), unsigned(frameScale.w), unsigned(frameScale.h)));
            transform.updated = false;
        }
        if (poolVariant && waterSteps) {                                          // waterSim.run() + TrianglePre (Pool.cpp:654-655)
            LRL_CHECK(lrl, lrl_water_step(lrl, waterSteps));
            if (deviceBVH) LRL_CHECK(lrl, lrl_scene_rebuild_bvh(lrl));
            model.moved = true;
        }
        if (transform.moved || model.moved) {                                     // decayOriginCalc.run() (:169-172)
            LRL_CHECK(lrl, lrl_compute_decay_origin(lrl, poolVariant ? LRL_VARIANT_POOL : LRL_VARIANT_BVH, &decayOriginData));
            transform.moved = false;
            model.moved = false;
        }
        LrlRenderParams p{};                                                      // renderer.use(); renderer.run() (:741-742)
        p.maxDepth = poolVariant ? 8 : 10;                                        // RayTraceDepth of the two shader copies
        p.variant = poolVariant ? LRL_VARIANT_POOL : LRL_VARIANT_BVH;
        p.mode = fast ? LRL_MODE_FAST : LRL_MODE_PARITY;
        p.spp = 1;
        p.flags = LRL_FLAG_OUT_DEVICE;
        LRL_CHECK(lrl, lrl_render(lrl, &p, frameDev, nullptr));                   // synchronous, like glDrawArrays + glFinish (:42, :815)
    }

    // what the window would show: RGBA8, clamped; rows bottom-up as the BMP stores them
    void writeBMP(std::string const& path) {
        const size_t px = size_t(frameScale.w) * frameScale.h;
        LRL_CHECK(lrl, lrl_quantize_rgba8(lrl, frameDev, rgba8Dev, px));
        LRL_CHECK(lrl, lrl_sync(lrl));
        std::vector<uint8_t> rgba(px * 4);
        HIP_CHECK(hipMemcpy(rgba.data(), rgba8Dev, px * 4, hipMemcpyDeviceToHost));
        BMP out;
        out.header.width = frameScale.w; out.header.height = frameScale.h;
        out.data.resize(px * 3);
        for (size_t i = 0; i < px; ++i) { out.data[i * 3] = rgba[i * 4 + 2]; out.data[i * 3 + 1] = rgba[i * 4 + 1]; out.data[i * 3 + 2] = rgba[i * 4]; }
        out.write(path);
    }

    void writePPMFile(std::string const& path) {                                  // the same RGBA8 frame as binary PPM
        const size_t px = size_t(frameScale.w) * frameScale.h;
        LRL_CHECK(lrl, lrl_quantize_rgba8(lrl, frameDev, rgba8Dev, px));
        LRL_CHECK(lrl, lrl_sync(lrl));
        std::vector<uint8_t> rgba(px * 4);
        HIP_CHECK(hipMemcpy(rgba.data(), rgba8Dev, px * 4, hipMemcpyDeviceToHost));
        writePPM(path, frameScale.w, frameScale.h, rgba.data());
    }
    void writePFMFile(std::string const& path) {                                  // the unclamped float frame lrl_render wrote
        const size_t px = size_t(frameScale.w) * frameScale.h;
        std::vector<float> rgba(px * 4);
        HIP_CHECK(hipMemcpy(rgba.data(), frameDev, px * 16, hipMemcpyDeviceToHost));
        writePFM(path, frameScale.w, frameScale.h, rgba.data());
    }
    void writeEXRFile(std::string const& path) {                                  // the same float frame as OpenEXR (with alpha)
        const size_t px = size_t(frameScale.w) * frameScale.h;
        std::vector<float> rgba(px * 4);
        HIP_CHECK(hipMemcpy(rgba.data(), frameDev, px * 16, hipMemcpyDeviceToHost));
        writeEXR(path, frameScale.w, frameScale.h, rgba.data());
    }

    ~RayTrace() {
        if (frameDev) (void)hipFree(frameDev);
        if (rgba8Dev) (void)hipFree(rgba8Dev);
        if (lrl) lrl_destroy(lrl);
    }
};

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <resources dir> <out.bmp> [--pool] [--size W H] [--frames N] [--fast] [--device-bvh] [--water-steps n] [--move] [--ppm f] [--pfm f] [--exr f]\n", argv[0]);
        return 2;
    }
    bool pool = false, fast = false, deviceBVH = false, moveSpheres = false;
    const char *ppm = nullptr, *pfm = nullptr, *exr = nullptr;
    int W = 720, H = 480, frames = 1;                                             // the reference's window: 720 x 480 (BVH.cpp:799)
    unsigned waterSteps = 0;
    for (int i = 3; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--pool")) pool = true;
        else if (!std::strcmp(argv[i], "--fast")) fast = true;
        else if (!std::strcmp(argv[i], "--device-bvh")) deviceBVH = true;
        else if (!std::strcmp(argv[i], "--move")) moveSpheres = true;
        else if (!std::strcmp(argv[i], "--ppm") && i + 1 < argc) ppm = argv[++i];
        else if (!std::strcmp(argv[i], "--pfm") && i + 1 < argc) pfm = argv[++i];
        else if (!std::strcmp(argv[i], "--exr") && i + 1 < argc) exr = argv[++i];
        else if (!std::strcmp(argv[i], "--size") && i + 2 < argc) { W = std::atoi(argv[i + 1]); H = std::atoi(argv[i + 2]); i += 2; }
        else if (!std::strcmp(argv[i], "--frames") && i + 1 < argc) frames = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--water-steps") && i + 1 < argc) waterSteps = unsigned(std::atoi(argv[++i]));
        else { std::fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    try {
        RayTrace test(argv[1], pool, fast, deviceBVH, waterSteps);
        FrameScale size; size.w = W; size.h = H;
        test.init(size);
        test.moving = moveSpheres && !pool;
        auto t0 = std::chrono::steady_clock::now();
        for (int f = 0; f < frames; ++f) {                                        // the loop of main(), BVH.cpp:813-822, without events and swap
            test.run();
            if ((f + 1) % 10 == 0) {
                const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                std::printf("\r%.2lf", 10.0 / s);                                 // the reference's FPS counter prints every ten frames (:818-822)
                std::fflush(stdout);
                t0 = std::chrono::steady_clock::now();
            }
        }
        if (frames >= 10) std::printf("\n");
        test.writeBMP(argv[2]);
        if (ppm) test.writePPMFile(ppm);
        if (pfm) test.writePFMFile(pfm);
        if (exr) test.writeEXRFile(exr);
        std::printf("%s: %d x %d, %d frame(s), decayOrigin (%g, %g, %g, %g)\n", argv[2], W, H, frames, test.decayOriginData.decayOrigin[0],
                    test.decayOriginData.decayOrigin[1], test.decayOriginData.decayOrigin[2], test.decayOriginData.decayOrigin[3]);
    } catch (std::exception const& e) {
        std::fprintf(stderr, "bvh_headless: %s\n", e.what());
        return 1;
    }
    return 0;
}
