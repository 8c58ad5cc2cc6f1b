// liblrltrace.so — the C ABI of include/lrl_trace.h: life cycle, tuning, camera, decay origin, measurement and information entry points,
// and the helpers every host unit uses (lrl_context.h).  Host side only.
#include "lrl_context.h"

namespace lrl_host {

int fail(LrlContext* c, int code, std::string const& msg) {
    if (c) c->err = msg;
    return code;
}

int ensure(LrlContext* c, DevBuf& b, size_t bytes) {
    if (bytes <= b.bytes && b.p) return LRL_OK;
    if (b.p) { HIP_TRY(c, hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
    size_t want = std::max<size_t>(bytes, 256);
    HIP_TRY(c, hipMalloc(&b.p, want));
    b.bytes = want;
    return LRL_OK;
}
int upload(LrlContext* c, DevBuf& b, const void* src, size_t bytes) {
    int rc = ensure(c, b, bytes);
    if (rc) return rc;
    if (bytes) HIP_TRY(c, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
    return LRL_OK;
}
void release(DevBuf& b) { if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.bytes = 0; } }

// A call that changes the scene works on the context's stream (or copies after waiting for it); the last render may have run on a stream of
// the caller's (lrl_render_async) and may still be reading what the call is about to change: the context's stream waits for it, on the device.
// The other direction is renderImpl's: a render on a caller's stream waits for what the context's stream still has queued.
int orderAfterRenders(LrlContext* c) {
    if (c->renderedOnce && c->lastRenderStream != c->stream) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->evLastRender, 0));
    return LRL_OK;
}

void fillKScene(const LrlContext* c, KScene& k) {
    k.recs = static_cast<const float4*>(c->nodes.p);
    k.node4 = static_cast<const uint4*>(c->nodes32.p);
    k.triHot = static_cast<const float4*>(c->triHot.p);
    k.triLeaf = static_cast<const float4*>(c->leafOrder ? c->triLeaf.p : c->triHot.p);
    k.triCold = static_cast<const LrlTriCold*>(c->triCold.p);
    k.triNeedUV = c->triNeedUV ? 1u : 0u;
    k.materials = static_cast<const LrlColor*>(c->materials.p);
    k.spheres = static_cast<const LrlSphere*>(c->spheres.p);
    k.circles = static_cast<const LrlCircle*>(c->circles.p);
    k.cylinders = static_cast<const LrlCylinder*>(c->cylinders.p);
    k.cones = static_cast<const LrlCone*>(c->cones.p);
    k.lights = static_cast<const LrlPointLight*>(c->lights.p);
    k.tex = static_cast<const uchar4*>(c->tex.p);
    k.cube = static_cast<const uchar4*>(c->cube.p);
    k.lightNum = c->num.pointLightNum;
    k.texW = c->texW; k.texH = c->texH; k.texL = c->texL;
    k.cubeSize = c->cubeSize;
    std::memcpy(k.decayOrigin, c->decayOrigin, 16);
}

LrlTuning defaultTuning() {
    LrlTuning t{};
    t.handoverEarly = LRL_HANDOVER_AUTO; t.handoverLate = LRL_HANDOVER_AUTO; t.handoverListLen = 0;
    t.adapt = 1; t.groupLanes = 0; t.contSlots = 0; t.groupCapX = 0; t.sparseFrom = 5; t.compactFrom = -1; t.coopFetch = 0; t.groupQuad = 0; t.streamWords = 0; t.refillMin = 16; t.phasedTrips = 0; t.deviceBuilder = 0;
    t.workLow = 12.0f; t.pollLag = 1; t.traceRef = 0; t.batchSlots = 0; t.debug = 0; t.treeLayout = 0; t.deterministic = 0;
    t.tailFrom = 0;
    return t;
}

#ifdef LRL_DIAG
void tuningFromEnv(LrlTuning& t) {
    auto u = [](const char* name, uint32_t& v) { if (const char* e = std::getenv(name)) v = uint32_t(std::strtoul(e, nullptr, 10)); };
    u("LRL_HANDOVER", t.handoverEarly); u("LRL_HANDOVER_LATE", t.handoverLate); u("LRL_ADAPT", t.adapt); u("LRL_SWARM_GROUP", t.groupLanes);
    u("LRL_CONT_SLOTS", t.contSlots); u("LRL_GROUP_CAP_X", t.groupCapX); u("LRL_SPARSE_FROM", t.sparseFrom); u("LRL_POLL_LAG", t.pollLag);
    u("LRL_BATCH_SLOTS", t.batchSlots); u("LRL_DEBUG", t.debug);
    if (const char* e = std::getenv("LRL_COMPACT_FROM")) t.compactFrom = std::atoi(e);
    if (const char* e = std::getenv("LRL_WORK_LOW")) t.workLow = float(std::atof(e));
    if (const char* e = std::getenv("LRL_TRACE")) t.traceRef = !std::strcmp(e, "ref");
    if (const char* tl = std::getenv("LRL_HANDOVER_LIST")) {
        t.handoverListLen = 0;
        for (const char* q = tl; *q && t.handoverListLen < LRL_TUNING_LIST_MAX;) {
            t.handoverList[t.handoverListLen++] = uint32_t(std::strtoul(q, const_cast<char**>(&q), 10));
            if (*q == ',') ++q; else break;
        }
    }
}
#endif

}  // namespace lrl_host

using namespace lrl_host;

extern "C" {

int lrl_version(void) { return 2; }

const char* lrl_last_error(const LrlContext* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int lrl_create(int device, LrlContext** out) {
    if (!out) return LRL_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return LRL_ERR_DEVICE;
    if (device < 0 || device >= n) return LRL_ERR_INVALID_ARG;
    LrlContext* c = new (std::nothrow) LrlContext();
    if (!c) return LRL_ERR_OUT_OF_MEMORY;
    c->device = device;
    c->tuning = defaultTuning();
    // everything the render path needs besides device memory exists from here on: pinned words the host polls / the device reports
    // into, and the events (creating events or pinned memory inside a timed loop stalled next to a collective's worker thread)
    bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreate(&c->evStart) == hipSuccess && hipEventCreate(&c->evStop) == hipSuccess &&
              hipHostMalloc(reinterpret_cast<void**>(&c->aliveHost), LRL_WF_MAX_ITERATIONS * 4 * 2, hipHostMallocDefault) == hipSuccess &&
              hipHostGetDevicePointer(reinterpret_cast<void**>(&c->aliveHostDev), c->aliveHost, 0) == hipSuccess &&
              hipHostMalloc(reinterpret_cast<void**>(&c->contHost), LRL_FEEDBACK_ITERS * 32, hipHostMallocDefault) == hipSuccess &&
              hipHostGetDevicePointer(reinterpret_cast<void**>(&c->contHostDev), c->contHost, 0) == hipSuccess &&
              hipEventCreateWithFlags(&c->evCont, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->evLastRender, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->evSceneReady, hipEventDisableTiming) == hipSuccess;
    for (hipEvent_t& e : c->evIter) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    for (hipEvent_t& e : c->phasedTrial.ev) ok = ok && hipEventCreate(&e) == hipSuccess;
    ok = ok && noallocScratchFor(device, &c->noallocScratch) == hipSuccess;
    ok = ok && hipHostMalloc(reinterpret_cast<void**>(&c->noallocPinned), size_t(LRL_NOALLOC_MAX_TRIANGLES) * (sizeof(LrlTriangle) + 4u), hipHostMallocDefault) == hipSuccess;
    if (!ok) {
        (void)lrl_destroy(c);
        return LRL_ERR_DEVICE;
    }
#ifdef LRL_DIAG
    tuningFromEnv(c->tuning);      // diagnostic builds (tools/build_variant.sh): knobs from the environment, read once, here
#endif
    *out = c;
    return LRL_OK;
}

int lrl_get_tuning(const LrlContext* c, LrlTuning* out) {
    if (!c || !out) return LRL_ERR_INVALID_ARG;
    *out = c->tuning;
    return LRL_OK;
}

int lrl_set_tuning(LrlContext* c, const LrlTuning* t) {
    if (!c || !t) return LRL_ERR_INVALID_ARG;
    if (t->handoverListLen > LRL_TUNING_LIST_MAX) return fail(c, LRL_ERR_INVALID_ARG, "handoverListLen > LRL_TUNING_LIST_MAX");
    if (t->groupLanes != 0 && t->groupLanes != 8 && t->groupLanes != 16 && t->groupLanes != 32 && t->groupLanes != 64) return fail(c, LRL_ERR_INVALID_ARG, "groupLanes must be 0, 8, 16, 32 or 64");
    if ((t->groupCapX > 2 && t->groupCapX != 4) || t->compactFrom < -1 || !(t->workLow >= 0.0f)) return fail(c, LRL_ERR_INVALID_ARG, "groupCapX must be 0, 1, 2 or 4, compactFrom >= -1, workLow >= 0");
    if (t->contSlots > (1u << 22)) return fail(c, LRL_ERR_LIMIT, "contSlots > 2^22");
    if (t->coopFetch > 1 || t->groupQuad > 1) return fail(c, LRL_ERR_INVALID_ARG, "coopFetch and groupQuad must be 0 or 1");
    if (t->phasedTrips > 2) return fail(c, LRL_ERR_INVALID_ARG, "phasedTrips must be 0 (by launch size), 1 (on) or 2 (off)");
    if (t->deviceBuilder > 2) return fail(c, LRL_ERR_INVALID_ARG, "deviceBuilder must be 0 (PLOC), 1 (LBVH) or 2 (PLOC)");
    if (t->streamWords > 4096 || t->refillMin > 64) return fail(c, LRL_ERR_INVALID_ARG, "streamWords <= 4096, refillMin <= 64");
    if (t->deterministic > 1 || t->treeLayout > 1) return fail(c, LRL_ERR_INVALID_ARG, "deterministic and treeLayout must be 0 or 1");
    c->tuning = *t;
    c->phasedTrial.state = 0; c->phasedTrial.signature = 0;
    c->handoverThr.clear();      // the adaptive thresholds start again from the new base values
    c->feedbackTick = 0;
    c->contIters = 0;
    return LRL_OK;
}

int lrl_destroy(LrlContext* c) {
    if (!c) return LRL_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    // a render issued with lrl_render_async on a stream of the caller's may still be in flight: it reads the buffers freed below and
    // writes its feedback into the pinned words
    if (c->renderedOnce && c->evLastRender) (void)hipEventSynchronize(c->evLastRender);
    if (c->evCont) (void)hipEventSynchronize(c->evCont);
    for (DevBuf* b : {&c->nodes32Raw, &c->layoutScratch, &c->bvhDev, &c->src4, &c->src4Raw, &c->callerScratch, &c->refitScratch, &c->triLeaf, &c->posOfTri, &c->primMin, &c->primMax, &c->buildScratch, &c->buildInfo, &c->waterPoints, &c->waterTris, &c->waterMat, &c->nodes, &c->nodes32, &c->triHot, &c->triCold, &c->materials, &c->spheres, &c->circles, &c->cylinders, &c->cones, &c->lights,
                      &c->tex, &c->cube, &c->triStage, &c->matStage, &c->transDev, &c->outDev, &c->hitDev, &c->counterDev, &c->decayDev, &c->timelineDev, &c->wfArena})
        release(*b);
    for (auto& e : c->timingEvents) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (c->aliveHost) (void)hipHostFree(c->aliveHost);
    if (c->contHost) (void)hipHostFree(c->contHost);
    if (c->noallocPinned) (void)hipHostFree(c->noallocPinned);
    if (c->evCont) (void)hipEventDestroy(c->evCont);
    if (c->evLastRender) (void)hipEventDestroy(c->evLastRender);
    if (c->evSceneReady) (void)hipEventDestroy(c->evSceneReady);
    for (hipEvent_t e : c->evIter) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->phasedTrial.ev) if (e) (void)hipEventDestroy(e);
    if (c->evStart) (void)hipEventDestroy(c->evStart);
    if (c->evStop) (void)hipEventDestroy(c->evStop);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return LRL_OK;
}

int lrl_set_camera(LrlContext* c, const LrlTrans* trans, uint32_t width, uint32_t height) {
    if (!c || !trans || !width || !height || width > 65536 || height > 65536) return LRL_ERR_INVALID_ARG;
    c->trans = *trans;
    c->width = width; c->height = height;
    c->haveCamera = true;
    return LRL_OK;
}

int lrl_set_decay_origin(LrlContext* c, const float d[4]) {
    if (!c || !d) return LRL_ERR_INVALID_ARG;
    std::memcpy(c->decayOrigin, d, 16);
    return LRL_OK;
}

int lrl_compute_decay_origin(LrlContext* c, uint32_t variant, LrlDecayOrigin* out) {
    if (!c || variant > LRL_VARIANT_POOL) return LRL_ERR_INVALID_ARG;
    if (!c->haveScene || !c->haveCamera) return fail(c, LRL_ERR_NO_SCENE, "scene and camera must be set first");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = upload(c, c->transDev, &c->trans, sizeof(LrlTrans));
    c->transUploaded.clear();      // transDev no longer holds a render's per-sample cameras
    if (rc) return rc;
    if ((rc = ensure(c, c->decayDev, sizeof(LrlDecayOrigin)))) return rc;
    KScene k;
    fillKScene(c, k);
    HIP_TRY(c, lrl_launch_decay_origin(k, static_cast<const LrlTrans*>(c->transDev.p), std::min(c->bvhDepth + 1, LRL_LDS_STACK_ENTRIES), variant == LRL_VARIANT_POOL,
                                       static_cast<LrlDecayOrigin*>(c->decayDev.p), c->stream));
    LrlDecayOrigin h;
    HIP_TRY(c, hipMemcpyAsync(&h, c->decayDev.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::memcpy(c->decayOrigin, h.decayOrigin, 16);
    if (out) *out = h;
    return LRL_OK;
}

int lrl_sync(LrlContext* c) {
    if (!c) return LRL_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return LRL_OK;
}

int lrl_quantize_rgba8(LrlContext* c, const float* rgbaDev, uint8_t* outDev, uint64_t pixels) {
    if (!c || (pixels && (!rgbaDev || !outDev)) || pixels > (1ull << 32)) return LRL_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, lrl_launch_quantize(reinterpret_cast<const float4*>(rgbaDev), reinterpret_cast<uchar4*>(outDev), pixels, c->stream));
    return LRL_OK;
}

int lrl_get_counters(LrlContext* c, LrlCounters* out) {
    if (!c || !out) return LRL_ERR_INVALID_ARG;
    *out = c->counters;
    return LRL_OK;
}

int lrl_last_kernel_ms(LrlContext* c, float* ms) {
    if (!c || !ms) return LRL_ERR_INVALID_ARG;
    if (!c->timed) return fail(c, LRL_ERR_NO_SCENE, "nothing rendered yet");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventSynchronize(c->evStop));
    HIP_TRY(c, hipEventElapsedTime(ms, c->evStart, c->evStop));
    return LRL_OK;
}

int lrl_timing_begin(LrlContext* c) {
    if (!c) return LRL_ERR_INVALID_ARG;
    // the event pairs of a typical timed loop exist before it starts: creating events while another thread of the process
    // sits in a HIP wait (a host-side collective's worker, as in bench.py's gloo rehearsal) stalled the loop
    HIP_TRY(c, hipSetDevice(c->device));
    while (c->timingEvents.size() < 256) {
        hipEvent_t a, b;
        HIP_TRY(c, hipEventCreate(&a));
        HIP_TRY(c, hipEventCreate(&b));
        c->timingEvents.emplace_back(a, b);
    }
    c->timing = true;
    c->timingUsed = 0;
    return LRL_OK;
}

int lrl_timing_end(LrlContext* c, double* totalMs, uint32_t* launches) {
    if (!c || !totalMs || !launches) return LRL_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    double sum = 0;
    for (uint32_t i = 0; i < c->timingUsed; ++i) {
        float ms = 0;
        HIP_TRY(c, hipEventSynchronize(c->timingEvents[i].second));
        HIP_TRY(c, hipEventElapsedTime(&ms, c->timingEvents[i].first, c->timingEvents[i].second));
        sum += ms;
    }
    *totalMs = sum;
    *launches = c->timingUsed;
    c->timing = false;
    c->timingUsed = 0;
    return LRL_OK;
}

int lrl_scene_info(const LrlContext* c, LrlSceneInfo* out) {
    if (!c || !out) return LRL_ERR_INVALID_ARG;
    if (!c->haveScene) return LRL_ERR_NO_SCENE;
    out->bvhNodes = c->nodeNum; out->bvhRecords = c->recNum; out->bvhDepth = c->bvhDepth; out->wideDepth = c->node4Depth;
    out->deviceBuilt = c->deviceBuilt ? 1u : 0u; out->materials = uint32_t(c->materialTable.size());
    out->ownBoxes = c->parkedTight;
    if (c->fitActive && c->refitted) {      // the tree has followed its triangles since it came: the last refit's count (this call waits for it)
        uint32_t tight = 0;
        if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess ||
            hipMemcpy(&tight, lrl_refit_tight_count(c->refitScratch.p, c->wideNodes), 4, hipMemcpyDeviceToHost) != hipSuccess) return LRL_ERR_DEVICE;
        out->ownBoxes = tight;
    }
    out->wideNodes = c->wideNodes; out->leafOrder = c->leafOrder ? 1u : 0u;
    return LRL_OK;
}

int lrl_last_render_info(const LrlContext* c, LrlRenderInfo* out) {
    if (!c || !out) return LRL_ERR_INVALID_ARG;
    *out = c->renderInfo;
    return LRL_OK;
}

int lrl_scene_device_bytes(LrlContext* c, uint64_t out[4]) {
    if (!c || !out) return LRL_ERR_INVALID_ARG;
    out[0] = uint64_t(c->recNum) * 64 + uint64_t(c->wideNodes) * 64;      // binary records + the four-wide nodes in use
    out[1] = uint64_t(c->num.triangleNum) * 64 * (c->leafOrder ? 2 : 1);      // hot records by triangle number (+ the copy in leaf order)
    out[2] = uint64_t(c->num.triangleNum) * sizeof(LrlTriCold) + c->materialTable.size() * sizeof(LrlColor);
    out[3] = uint64_t(c->num.sphereNum + c->num.circleNum + c->num.cylinderNum + c->num.coneNum) * 128 + uint64_t(c->num.pointLightNum) * 32 +
             uint64_t(c->texW) * c->texH * c->texL * 4 + uint64_t(c->cubeSize) * c->cubeSize * 24;
    return LRL_OK;
}

}  // extern "C"
