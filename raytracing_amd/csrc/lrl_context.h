// Host side of liblrltrace.so, shared by its three host translation units:
//   lrl_context.hip   life cycle, tuning, camera, measurement and information entry points; the helpers every unit uses
//   lrl_scene.hip     scene upload and updates: records, the caller's tree and its four-wide repack, the fit to the primitives' own boxes,
//                     the layout pass, the device build, the water simulation
//   lrl_schedule.hip  the frame: the wavefront pipeline (generate, trace + shade per ray segment), its scheduling decisions, lrl_render
// The kernels live in kernels_parity.hip / kernels_fast.hip (lrl_kernels.inc, lrl_wavefront.inc, lrl_lean.inc) and lrl_build.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "lrl_device.h"

#define LRL_WF_MAX_ITERATIONS 4096u   /* ray segments of one path the wavefront pipeline follows at most */

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct ColorKey {
    uint32_t w[20];
    bool operator==(ColorKey const& o) const { return std::memcmp(w, o.w, sizeof(w)) == 0; }
};
struct ColorHash {
    size_t operator()(ColorKey const& k) const {
        uint64_t h = 1469598103934665603ull;
        for (uint32_t v : k.w) { h ^= v; h *= 1099511628211ull; }
        return size_t(h);
    }
};

struct LrlContext {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t evStart = nullptr, evStop = nullptr;
    std::string err;

    // scene
    bool haveScene = false, haveCamera = false;
    bool deviceBuilt = false;      // the tree was built by lrl_build.hip: primMin / primMax hold the primitives' boxes
    uint32_t primTotal = 0;
    LrlGeometryNum num{};
    uint32_t nodeNum = 0, recNum = 0, bvhDepth = 0;
    DevBuf primMin, primMax, buildScratch, buildInfo;   // device BVH build (lrl_build.hip)
    DevBuf waterPoints, waterTris, waterMat;   // Pool water simulation (lrl_water_*)
    uint32_t waterGX = 0, waterGY = 0, waterFirst = 0;
    float waterK = 0, waterDt = 0;
    DevBuf nodes, nodes32, triHot, triCold, materials, spheres, circles, cylinders, cones, lights, tex, cube, triStage, matStage;
    // the four-wide tree as the rays walk it (lrl_layout_node4): nodes32 holds the laid-out nodes, triLeaf the hot records in the
    // order of their leaves, posOfTri where each triangle's record went
    DevBuf nodes32Raw, layoutScratch, triLeaf, posOfTri;
    // a caller's tree on the device (lrl_caller_tree): its nodes and where each four-wide item's box comes from are kept while the tree is
    // fitted to the triangles' own boxes (fitActive), for the refit that follows them (lrl_refit_node4)
    DevBuf bvhDev, src4, src4Raw, callerScratch, refitScratch;
    bool fitActive = false;
    bool refitted = false;        // the count of tightly bounded primitives is the last refit's (in refitScratch), not the upload's
    bool triBoxes = false;        // primMin / primMax hold every triangle's box as it is now
    float fitPad = 0.0f;
    bool leafOrder = false;        // triLeaf / posOfTri are in use (else the four-wide kernels read triHot itself)
    uint32_t wideNodes = 0;        // four-wide nodes in use (nodes32)
    uint32_t texW = 0, texH = 0, texL = 0, cubeSize = 0;
    std::vector<LrlColor> materialTable;
    std::unordered_map<ColorKey, uint32_t, ColorHash> materialIndex;
    // callers' trees: the triangles' own boxes (six floats each, min > max: not known), kept current by the update calls; the
    // four-wide tree is fitted to them (fitOwnBoxes)
    uint32_t parkedTight = 0;     // primitives that nodes32 bounds more tightly than the caller's tree does (lrl_caller_tree)

    // frame
    LrlTrans trans{};
    uint32_t width = 0, height = 0;
    float decayOrigin[4] = {0, 0, 0, 1};
    uint32_t node4Depth = 0;   // depth of the four-wide tree (nodes32)
    DevBuf transDev, outDev, hitDev, counterDev, decayDev, timelineDev;
    DevBuf wfArena;                     // wavefront pipeline: path state, ray records, masks, refraction stacks, overflow stacks
    uint32_t* aliveHost = nullptr;      // pinned, 2 x LRL_WF_MAX_ITERATIONS (alternating by sample): per-iteration "any path alive" words the shade kernel writes
    uint32_t* aliveHostDev = nullptr;   // the device's address of the same memory
    // feedback for the hand-over thresholds of the FAST trace kernels: rays handed over per iteration of the last finished
    // sample (pinned copy, read when its event has completed: never waited for)
    uint32_t* contHost = nullptr;       // pinned, [LRL_FEEDBACK_ITERS][8]: {rays handed over, LRL_WORK_COUNTERS work counters} written by the shade kernel
    uint32_t* contHostDev = nullptr;    // the device's address of the same memory
    uint32_t contSlotsInFlight = 0;     // hand-over slots of the pass whose feedback is in flight
    hipEvent_t evCont = nullptr;
    // triangle trips of the lean kernel (lrl_lean.inc, PH): decided per scene and frame shape by timing one frame each way
    struct PhasedTrial {
        int state = 0;                  // 0..2: frames that let the hand-over thresholds settle; 3..6: timed frames, without / with / without / with; 7: waiting for the events; 8: decided
        uint64_t signature = 0;         // frame shape the trial belongs to
        bool choice = false;
        hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};      // start, end of the four timed frames
    } phasedTrial;
    hipEvent_t evSceneReady = nullptr;  // recorded on the context's stream when a render starts on another one, which waits for it
    hipEvent_t evLastRender = nullptr;  // recorded behind every render: a render on another stream waits for it (the arena is shared)
    hipStream_t lastRenderStream = nullptr;
    bool renderedOnce = false;
    LrlTuning tuning{};
    LrlRenderInfo renderInfo{};
    void* noallocScratch = nullptr;     // this device's copy of the static staging array (lrl_scene_update_triangles_noalloc)
    unsigned char* noallocPinned = nullptr;   // pinned host mirror of that array's layout (records, then material indices), allocated in lrl_create: the
                                        // entry point copies the caller's records here, so that its host-to-device copies are plain DMA from pinned memory
                                        // (a copy from pageable memory makes the runtime stage, and may make it allocate, behind the caller's back)
    std::vector<float> workPerRay;      // per iteration: frontier entries per handed-over ray, last feedback
    uint32_t expectedIters = 0;         // iterations the last pass of a scene with refraction needed (renderWavefront: enqueued without looking)
    uint32_t feedbackTick = 0;          // samples rendered since the thresholds were reset
    uint32_t contIters = 0;             // words of contHost in flight / ready
    std::vector<LrlTrans> transUploaded;   // what transDev holds (per-sample cameras of the last render)
    std::vector<uint32_t> handoverThr;  // per iteration index; empty until first use
    uint32_t handoverRegime = 0;        // what the base thresholds were chosen for: 0 a sample in the caches, 1 a large launch, 2 a scene beyond the caches
    std::vector<uint32_t> swarmGrid;    // waves launched for the handed-over rays of that iteration (a grid-stride kernel)
    std::vector<uint8_t> handoverIdle;  // per iteration: the last feedback saw no ray reach the threshold (the hand-over launch is left out until the next feedback pass)
    hipEvent_t evIter[4] = {nullptr, nullptr, nullptr, nullptr};
    bool hasTransmission = false;       // some material has t != 0: refraction stacks are needed
    bool triNeedUV = false;             // some triangle material has a texture or the checker: a hit on a triangle needs its texture coordinates
    uint32_t lastIterations = 0;
    float lastMs = 0;
    bool timed = false;
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timingEvents;
    uint32_t timingUsed = 0;
    LrlCounters counters{};
};

namespace lrl_host {

int fail(LrlContext* c, int code, std::string const& msg);
#define HIP_TRY(c, expr)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(c, e_ == hipErrorOutOfMemory ? LRL_ERR_OUT_OF_MEMORY : LRL_ERR_DEVICE, \
                                          std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// device buffers that grow on demand, uploads on the context's stream, release
int ensure(LrlContext* c, DevBuf& b, size_t bytes);
int upload(LrlContext* c, DevBuf& b, const void* src, size_t bytes);
void release(DevBuf& b);
// A call that changes the scene waits (on the device) for the last render, which may have run on a stream of the caller's
int orderAfterRenders(LrlContext* c);
void fillKScene(const LrlContext* c, KScene& k);
LrlTuning defaultTuning();
// lrl_scene.hip
hipError_t noallocScratchFor(int device, void** out);      // this device's copy of the static staging array of lrl_scene_update_triangles_noalloc

}  // namespace lrl_host
