// liblrltrace.so — implementation of the C ABI declared in include/lrl_trace.h.
// Host side only: validation, the AoS -> device-layout repack, uploads, launches, timing.
// The kernels live in kernels_parity.hip / kernels_fast.hip (lrl_kernels.inc).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "lrl_device.h"

#define LRL_WF_MAX_ITERATIONS 4096u   /* ray segments of one path the wavefront pipeline follows at most */

// ---- lrl_scene_update_triangles_noalloc: its staging area ----
// lrl_scene_update_triangles stages the caller's origin records and their material indices in device buffers it allocates
// (triStage, matStage) for trianglePreKernel to read.  The no-allocation variant stages them here instead: one array per GPU,
// part of the code object, held from the library's load until the process ends.  Layout: LRL_NOALLOC_MAX_TRIANGLES origin
// records (160 B each), then as many material indices (4 B each).  Host code never uses the array's name as an address: lrl_create
// asks hipGetSymbolAddress for this device's copy (noallocScratchFor) and everything below works on that pointer.
static __device__ __align__(256) unsigned char scratch[size_t(LRL_NOALLOC_MAX_TRIANGLES) * (sizeof(LrlTriangle) + 4u)];

namespace {

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

}  // namespace

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
    DevBuf nodes, nodes32, nodes32Loose, triHot, triCold, materials, spheres, circles, cylinders, cones, lights, tex, cube, triStage, matStage;
    // the four-wide tree as the rays walk it (lrl_layout_node4): nodes32 / nodes32Loose hold the laid-out nodes, triLeaf the hot records in the
    // order of their leaves, posOfTri where each triangle's record went
    DevBuf nodes32Raw, layoutScratch, layoutSizes, triLeaf, posOfTri;
    bool leafOrder = false;        // triLeaf / posOfTri are in use (else the four-wide kernels read triHot itself)
    uint32_t wideNodes = 0;        // four-wide nodes in use (nodes32)
    uint32_t texW = 0, texH = 0, texL = 0, cubeSize = 0;
    std::vector<LrlColor> materialTable;
    std::unordered_map<ColorKey, uint32_t, ColorHash> materialIndex;
    // callers' trees: the triangles' own boxes (six floats each, min > max: not known), kept current by the update calls; the
    // four-wide tree is fitted to them (fitOwnBoxes)
    std::vector<float> ownBox;
    uint32_t parkedTight = 0;     // triangles that nodes32 bounds by their own boxes (fitOwnBoxes)
    bool haveLoose = false;       // nodes32Loose holds the same four-wide tree with the caller's boxes only
    bool looseInUse = false;      // primitives changed under the caller's tree since it came: its boxes alone are the contract (ownBoxesStale)
    std::vector<LrlBVHNode> callerTree;   // the caller's tree, kept where fitting pays: a scene that has come to rest is fitted again (refitAtRest)
    uint64_t pathsAtRest = 0;     // paths (pixels x samples) rendered since the last change under the caller's tree

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
    uint32_t feedbackTick = 0;          // samples rendered since the thresholds were reset
    uint32_t contIters = 0;             // words of contHost in flight / ready
    std::vector<LrlTrans> transUploaded;   // what transDev holds (per-sample cameras of the last render)
    std::vector<uint32_t> handoverThr;  // per iteration index; empty until first use
    uint32_t handoverRegime = 0;        // what the base thresholds were chosen for: 0 a sample in the caches, 1 a large launch, 2 a scene beyond the caches
    std::vector<uint32_t> swarmGrid;    // waves launched for the handed-over rays of that iteration (a grid-stride kernel)
    std::vector<uint8_t> handoverIdle;  // per iteration: the last feedback saw no ray reach the threshold (the hand-over launch is left out until the next feedback pass)
    hipEvent_t evIter[4] = {nullptr, nullptr, nullptr, nullptr};
    bool hasTransmission = false;       // some material has t != 0: refraction stacks are needed
    uint32_t lastIterations = 0;
    float lastMs = 0;
    bool timed = false;
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timingEvents;
    uint32_t timingUsed = 0;
    LrlCounters counters{};
};

namespace {

int fail(LrlContext* c, int code, std::string const& msg) {
    if (c) c->err = msg;
    return code;
}
#define HIP_TRY(c, expr)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(c, e_ == hipErrorOutOfMemory ? LRL_ERR_OUT_OF_MEMORY : LRL_ERR_DEVICE, \
                                          std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

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

bool transmits(const LrlColor& col) { return !(col.t[0] == 0.0f && col.t[1] == 0.0f && col.t[2] == 0.0f); }   // NaN counts as transmitting

uint32_t internMaterial(LrlContext* c, const LrlColor& col) {
    if (transmits(col)) c->hasTransmission = true;
    ColorKey k;
    std::memcpy(k.w, &col, sizeof(k.w));
    auto it = c->materialIndex.find(k);
    if (it != c->materialIndex.end()) return it->second;
    uint32_t id = uint32_t(c->materialTable.size());
    c->materialTable.push_back(col);
    c->materialIndex.emplace(k, id);
    return id;
}

// Structural checks of the caller's BVH so that no lane can index outside a buffer or overrun its stack.
int validateBVH(LrlContext* c, const LrlSceneView* s, uint32_t& depthOut) {
    const uint32_t N = s->bvhNodeNum;
    if (N == 0 || N >= LRL_REF_LEAF) return fail(c, LRL_ERR_INVALID_ARG, "bvhNodeNum out of range");
    const uint32_t counts[7] = {0, 0, s->num.triangleNum, s->num.sphereNum, s->num.circleNum, s->num.cylinderNum, s->num.coneNum};
    for (uint32_t i = 0; i < N; ++i) {
        const LrlBVHNode& n = s->bvh[i];
        if (n.bound.leftChild != 0 && i + 1 >= N) return fail(c, LRL_ERR_INVALID_ARG, "BVH: left child past the end");
        if (n.bound.rightChild != 0 && (n.bound.rightChild < 0 || uint32_t(n.bound.rightChild) >= N || uint32_t(n.bound.rightChild) <= i))
            return fail(c, LRL_ERR_INVALID_ARG, "BVH: right child index invalid (must be a later pre-order index)");
        if (n.bound.rightChild != 0 && n.bound.leftChild == 0) return fail(c, LRL_ERR_INVALID_ARG, "BVH: right child without left child");
        if (n.axis > 2) return fail(c, LRL_ERR_INVALID_ARG, "BVH: axis > 2");
        if (n.geometry != 0) {
            if (n.geometry < 2 || n.geometry > 6) return fail(c, LRL_ERR_INVALID_ARG, "BVH: unknown geometry type");
            if (n.geometryNum >= counts[n.geometry] || n.geometryNum > LRL_PRIM_INDEX_MASK)
                return fail(c, LRL_ERR_INVALID_ARG, "BVH: geometryNum out of range");
        }
    }
    // depth by an explicit-stack walk; children always have larger indices, so this terminates
    std::vector<std::pair<uint32_t, uint32_t>> st;
    st.emplace_back(0u, 0u);
    uint32_t depth = 0;
    size_t visited = 0;
    while (!st.empty()) {
        auto [i, d] = st.back();
        st.pop_back();
        depth = std::max(depth, d);
        if (++visited > size_t(N)) return fail(c, LRL_ERR_INVALID_ARG, "BVH: a node is reachable twice");
        const LrlBVHNode& n = s->bvh[i];
        if (n.bound.rightChild) st.emplace_back(uint32_t(n.bound.rightChild), d + 1);
        if (n.bound.leftChild) st.emplace_back(i + 1, d + 1);
    }
    depthOut = depth;
    if (depth + 1 > LRL_MAX_BVH_DEPTH) return fail(c, LRL_ERR_BVH_DEPTH, "BVH deeper than LRL_MAX_BVH_DEPTH");
    return LRL_OK;
}

// A call that changes the scene works on the context's stream (or copies after waiting for it); the last render may have run on a stream of
// the caller's (lrl_render_async) and may still be reading what the call is about to change: the context's stream waits for it, on the device.
// The other direction is renderImpl's: a render on a caller's stream waits for what the context's stream still has queued.
int orderAfterRenders(LrlContext* c) {
    if (c->renderedOnce && c->lastRenderStream != c->stream) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->evLastRender, 0));
    return LRL_OK;
}

void fillKScene(const LrlContext* c, KScene& k) {
    k.recs = static_cast<const float4*>(c->nodes.p);
    k.node4 = static_cast<const uint4*>(c->looseInUse ? c->nodes32Loose.p : c->nodes32.p);
    k.triHot = static_cast<const float4*>(c->triHot.p);
    k.triLeaf = static_cast<const float4*>(c->leafOrder ? c->triLeaf.p : c->triHot.p);
    k.triCold = static_cast<const LrlTriCold*>(c->triCold.p);
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

std::vector<uchar4> bgrToRgbx(const uint8_t* bgr, size_t texels) {
    std::vector<uchar4> o(texels);
    for (size_t i = 0; i < texels; ++i) o[i] = make_uchar4(bgr[i * 3 + 2], bgr[i * 3 + 1], bgr[i * 3 + 0], 255);
    return o;
}


// The caller's tree refitted to the primitives' own boxes, for the four-wide tree (buildNode4).
// The reference tests a primitive parked in an inner node whenever the ray enters that node's box, which spans a whole subtree, and the
// Pool's tree has leaf boxes as tall as the highest wave (Pool/Pool.cpp:250-263).  While the primitives are where they were when the tree
// came, a box fitted to the primitive says the same for every ray that can hit it and turns most of the others away:
//   park[b] = the parked primitive's own box                      (its node's box where the own box is not known or pokes out of it)
//   node[b] = leaf: the primitive's own box (same rule); inner: union of park[b] and the children's node[] boxes
// Only for a nested tree (every child's box inside its parent's) and own boxes inside the caller's: then a ray that hits a primitive reaches
// it through either set of boxes, and the fitted ones (nested as well, a margin wider than the own boxes) only turn away rays that cannot hit.  `count`: primitives bounded more tightly than by the caller's tree.
struct FittedBoxes { std::vector<float> node, park; size_t count = 0; };
bool fitOwnBoxes(const LrlSceneView* s, const std::vector<float>& ownBox, FittedBoxes& f) {
    const LrlBVHNode* bvh = s->bvh;
    const uint32_t N = s->bvhNodeNum;
    f = FittedBoxes();
    if (ownBox.empty()) return false;
    auto callers = [&](uint32_t b, float* q) { for (int a = 0; a < 3; ++a) { q[a] = bvh[b].bound.min[a]; q[3 + a] = bvh[b].bound.max[a]; } };
    float scale = 0.0f;       // of the coordinates the slab arithmetic works with
    for (int a = 0; a < 3 && N; ++a) scale = std::max(scale, std::max(std::fabs(bvh[0].bound.min[a]), std::fabs(bvh[0].bound.max[a])));
    const float pad = std::ldexp(scale, -14);
    auto inside = [](const float* q, const float* outer) {
        for (int a = 0; a < 3; ++a) if (!(q[a] <= q[3 + a]) || q[a] < outer[a] || q[3 + a] > outer[3 + a]) return false;
        return true;
    };
    // own box of node b's primitive into q if it is known and lies in the caller's box; else the caller's box.  -> strictly tighter
    auto own = [&](uint32_t b, float* q, bool parked) {
        const LrlBVHNode& n = bvh[b];
        float outer[6];
        callers(b, outer);
        std::memcpy(q, outer, sizeof(outer));
        // triangles only: the other kinds go through the kernels' cold path, which is cheapest when the lanes of a wave take it together,
        // on arrival at the node (C1, where every kind is parked near the root: 1.53 ms of trace kernels as the caller's tree has them,
        // 1.63 ms with own boxes for all kinds)
        if (n.geometry != 2) return false;
        const std::vector<float>& o = ownBox;
        const size_t at = size_t(n.geometryNum) * 6;
        if (at + 6 > o.size() || !inside(&o[at], outer)) return false;
        // with a margin on every side: a hit on the box's face (every hit on a triangle that lies in an axis plane, every hit on an edge
        // that two neighbours' boxes share) must enter the box safely before the hit's distance, or one of two primitives at exactly
        // the same distance would be turned away by the rounding of the slab arithmetic instead of being decided by its code
        // (tests/test_gpu_parity.py::test_hand_over_kernels_agree, the Pool seen along its grid's diagonal)
        for (int a = 0; a < 3; ++a) { q[a] = o[at + a] - pad; q[3 + a] = o[at + 3 + a] + pad; }
        return parked || std::memcmp(&o[at], outer, sizeof(outer)) != 0;
    };
    // who gains, before anything is allocated (few gainers: not worth the second copy of the tree that ownBoxesStale falls back to; C2's
    // tree parks 0.5 % of its triangles)
    size_t gainers = 0;
    for (uint32_t b = 0; b < N; ++b) {
        float q[6];
        if (bvh[b].geometry) gainers += own(b, q, bvh[b].bound.leftChild != 0) ? 1 : 0;
    }
    if (gainers == 0 || gainers * 64 < ownBox.size() / 6) return false;
    f.node.assign(size_t(N) * 6, 0.0f);
    f.park.assign(size_t(N) * 6, 0.0f);
    for (uint32_t b = N; b-- > 0;) {      // pre-order: children come after their parent
        const LrlBVHNode& n = bvh[b];
        float* q = &f.node[size_t(b) * 6];
        if (!n.bound.leftChild) {
            if (n.geometry) f.count += own(b, q, false) ? 1 : 0;
            else callers(b, q);
            continue;
        }
        float outer[6];
        callers(b, outer);
        bool any = false;
        auto join = [&](const float* c) {
            for (int a = 0; a < 3; ++a) { q[a] = any ? std::min(q[a], c[a]) : c[a]; q[3 + a] = any ? std::max(q[3 + a], c[3 + a]) : c[3 + a]; }
            any = true;
        };
        if (n.geometry) { f.count += own(b, &f.park[size_t(b) * 6], true) ? 1 : 0; join(&f.park[size_t(b) * 6]); }
        const uint32_t kids[2] = {b + 1, uint32_t(n.bound.rightChild)};
        for (int k = 0; k < (n.bound.rightChild ? 2 : 1); ++k) {
            float cb[6];
            callers(kids[k], cb);
            if (!inside(cb, outer)) { f = FittedBoxes(); return false; }      // not nested: the caller's boxes as they are
            join(&f.node[size_t(kids[k]) * 6]);
        }
    }
    return true;
}

// Four-wide, compressed repack of the caller's binary tree for the FAST lean kernel (lrl_lean.inc).  A node's items are
// its parked primitive and its children; an inner child is replaced by ITS items while the node stays within four (largest
// box first).  Boxes: the caller's, or the fitted ones (fitOwnBoxes).  The node of binary inner node b is stored at b's record index
// (recOf[b], the index `recs` uses), so refs mean
// the same in both trees and a ray can change from one to the other at any ref; record indices of binary nodes that were
// merged into their parent stay unused.  64 B per node:
//   {origin.x, origin.y, origin.z, unit.x}                       origin = min corner of the items' boxes; unit = per-axis
//                                                                length of one byte step, a power of two (as a float)
//   {lo.x of items 0..3 (one byte each), lo.y, lo.z, hi.x}       box = origin + byte * unit, rounded outwards
//   {hi.y, hi.z, ref0, ref1}  {ref2, ref3, unit.y, unit.z}       ref: record index, LRL_REF_LEAF | primitive, or LRL_REF_EMPTY4 with an
//                                                                inverted box (lo 255, hi 0) for an empty slot
std::vector<uint4> buildNode4(const LrlSceneView* s, const FittedBoxes* fitted, const std::vector<uint32_t>& recOf, uint32_t recCount, uint32_t& depthOut) {
    const LrlBVHNode* bvh = s->bvh;
    struct Item { bool inner; uint32_t ref; float lo[3], hi[3]; };      // inner: binary node index; else primitive bits
    auto primOf = [&](const LrlBVHNode& n) { return (uint32_t(n.geometry) << LRL_PRIM_TYPE_SHIFT) | (n.geometryNum & LRL_PRIM_INDEX_MASK); };
    auto boxed = [&](bool inner, uint32_t ref, uint32_t boxOf, bool parked) {
        Item it{inner, ref, {0, 0, 0}, {0, 0, 0}};
        if (fitted) {
            const float* q = parked ? &fitted->park[size_t(boxOf) * 6] : &fitted->node[size_t(boxOf) * 6];
            for (int a = 0; a < 3; ++a) { it.lo[a] = q[a]; it.hi[a] = q[3 + a]; }
        } else
            for (int a = 0; a < 3; ++a) { it.lo[a] = bvh[boxOf].bound.min[a]; it.hi[a] = bvh[boxOf].bound.max[a]; }
        return it;
    };
    auto asItem = [&](uint32_t cnode, Item* out, int& k) {
        if (bvh[cnode].bound.leftChild) out[k++] = boxed(true, cnode, cnode, false);
        else if (bvh[cnode].geometry) out[k++] = boxed(false, primOf(bvh[cnode]), cnode, false);
    };
    auto contents = [&](uint32_t b, Item* out) -> int {
        const LrlBVHNode& n = bvh[b];
        int k = 0;
        if (n.geometry) out[k++] = boxed(false, primOf(n), b, true);
        if (n.bound.leftChild) {
            asItem(b + 1, out, k);
            if (n.bound.rightChild) asItem(uint32_t(n.bound.rightChild), out, k);
        }
        return k;
    };
    auto area = [&](uint32_t b) {
        const Item q = boxed(true, b, b, false);
        float dx = q.hi[0] - q.lo[0], dy = q.hi[1] - q.lo[1], dz = q.hi[2] - q.lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    auto fb = [](float f) { uint32_t v; std::memcpy(&v, &f, 4); return v; };
    std::vector<uint4> out(size_t(recCount) * 4, make_uint4(0, 0, LRL_REF_EMPTY4, LRL_REF_EMPTY4));
    struct Work { uint32_t bin; bool virt; uint32_t depth; };
    std::vector<Work> work;
    work.push_back({0u, true, 0u});
    uint32_t depth = 0;
    while (!work.empty()) {
        const Work w = work.back();
        work.pop_back();
        depth = std::max(depth, w.depth);
        Item items[8];
        int n = 0;
        if (w.virt) asItem(0u, items, n);      // record 0: the virtual parent of the root
        else n = contents(w.bin, items);
        for (;;) {
            int best = -1;
            float bestArea = -1.0f;
            Item sub[3];
            for (int i = 0; i < n; ++i) {
                if (!items[i].inner) continue;
                const int add = contents(items[i].ref, sub);
                if (n - 1 + add > 4) continue;
                const float a = area(items[i].ref);
                if (a > bestArea) { bestArea = a; best = i; }
            }
            if (best < 0) break;
            const int add = contents(items[best].ref, sub);
            for (int i = best; i + 1 < n; ++i) items[i] = items[i + 1];
            --n;
            for (int i = 0; i < add; ++i) items[n++] = sub[i];
        }
        uint4* q = &out[size_t(w.virt ? 0u : recOf[w.bin]) * 4];
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (int i = 0; i < n; ++i)
            for (int a = 0; a < 3; ++a) {
                lo[a] = std::min(lo[a], double(items[i].lo[a]));
                hi[a] = std::max(hi[a], double(items[i].hi[a]));
            }
        float origin[3] = {0, 0, 0};
        uint32_t expo[3] = {127, 127, 127};
        double unit[3] = {1, 1, 1};
        for (int a = 0; a < 3 && n; ++a) {
            origin[a] = float(lo[a]);
            if (double(origin[a]) > lo[a]) origin[a] = std::nextafter(origin[a], -INFINITY);
            const double ext = hi[a] - double(origin[a]);
            int e = ext > 0 ? int(std::ceil(std::log2(ext / 255.0))) : -126;
            e = std::min(127, std::max(-126, e));
            while (e < 127 && std::ceil(ext / std::ldexp(1.0, e)) > 255.0) ++e;
            expo[a] = uint32_t(e + 127);
            unit[a] = std::ldexp(1.0, e);
        }
        uint32_t bytes[6] = {0, 0, 0, 0, 0, 0};      // lo.x lo.y lo.z hi.x hi.y hi.z, item i in byte i
        uint32_t refs[4] = {LRL_REF_EMPTY4, LRL_REF_EMPTY4, LRL_REF_EMPTY4, LRL_REF_EMPTY4};
        for (int i = n; i < 4; ++i)
            for (int a = 0; a < 3; ++a) bytes[a] |= LRL_EMPTY4_LO << (8 * i);      // empty slots: inverted boxes
        for (int i = 0; i < n; ++i) {
            for (int a = 0; a < 3; ++a) {
                const double ql = std::floor((double(items[i].lo[a]) - double(origin[a])) / unit[a]);
                const double qh = std::ceil((double(items[i].hi[a]) - double(origin[a])) / unit[a]);
                bytes[a] |= uint32_t(std::min(255.0, std::max(0.0, ql))) << (8 * i);
                bytes[3 + a] |= uint32_t(std::min(255.0, std::max(0.0, qh))) << (8 * i);
            }
            if (items[i].inner) {
                refs[i] = recOf[items[i].ref];
                work.push_back({items[i].ref, false, w.depth + 1});
            } else refs[i] = LRL_REF_LEAF | items[i].ref;
        }
        q[0] = make_uint4(fb(origin[0]), fb(origin[1]), fb(origin[2]), expo[0] << 23);
        q[1] = make_uint4(bytes[0], bytes[1], bytes[2], bytes[3]);
        q[2] = make_uint4(bytes[4], bytes[5], refs[0], refs[1]);
        q[3] = make_uint4(refs[2], refs[3], expo[1] << 23, expo[2] << 23);
    }
    depthOut = depth;
    return out;
}

// ---- the four-wide tree in the order the rays walk it (lrl_layout_node4, lrl_build.hip) ----
// subtree sizes of a host-built four-wide tree by record index: sizes[r] = four-wide nodes, sizes[recCount + r] = triangle leaves (a node's
// items are records with larger indices: the binary tree is in pre-order)
void node4Sizes(const std::vector<uint4>& n4, uint32_t recCount, std::vector<uint32_t>& sizes) {
    sizes.assign(size_t(recCount) * 2, 0u);
    for (uint32_t r = recCount; r-- > 0;) {
        const uint32_t refs[4] = {n4[size_t(r) * 4 + 2].z, n4[size_t(r) * 4 + 2].w, n4[size_t(r) * 4 + 3].x, n4[size_t(r) * 4 + 3].y};
        uint32_t sN = 1, sT = 0;
        for (uint32_t ref : refs) {
            if (ref != LRL_REF_NONE && !(ref & LRL_REF_LEAF)) { if (ref > r && ref < recCount) { sN += sizes[ref]; sT += sizes[size_t(recCount) + ref]; } }
            else if ((ref >> LRL_PRIM_TYPE_SHIFT) == ((LRL_REF_LEAF >> LRL_PRIM_TYPE_SHIFT) | 2u)) ++sT;
        }
        sizes[r] = sN; sizes[size_t(recCount) + r] = sT;
    }
}
int ensureLeafOrder(LrlContext* c, uint32_t recCount, uint32_t triCount) {
    int rc;
    if ((rc = ensure(c, c->layoutScratch, lrl_layout_scratch_bytes(recCount)))) return rc;
    if ((rc = ensure(c, c->posOfTri, size_t(triCount) * 4))) return rc;
    return ensure(c, c->triLeaf, size_t(triCount) * 64);
}
void releaseLayoutTemporaries(LrlContext* c) {      // (hipFree waits for the device)
    if (c->nodes32Raw.bytes > (64u << 20)) release(c->nodes32Raw);
    if (c->layoutScratch.bytes > (64u << 20)) release(c->layoutScratch);
    if (c->layoutSizes.bytes > (64u << 20)) release(c->layoutSizes);
}
// a host-built four-wide tree (by record index) -> laid out in `dst`; assign: the triangles' places in triLeaf are made anew (posOfTri)
int layoutFromHost(LrlContext* c, const std::vector<uint4>& node4, uint32_t recCount, uint32_t depth4, DevBuf& dst, bool assign, uint32_t triCount, uint32_t& wideNodes) {
    std::vector<uint32_t> sizes;
    node4Sizes(node4, recCount, sizes);
    if (sizes[size_t(recCount)] > triCount) return fail(c, LRL_ERR_INVALID_ARG, "BVH: a triangle appears in more than one node");
    int rc;
    if ((rc = upload(c, c->nodes32Raw, node4.data(), node4.size() * sizeof(uint4)))) return rc;
    if ((rc = upload(c, c->layoutSizes, sizes.data(), sizes.size() * 4))) return rc;
    if ((rc = ensureLeafOrder(c, recCount, triCount))) return rc;
    if ((rc = ensure(c, dst, size_t(sizes[0]) * 64))) return rc;
    const uint32_t* sz = static_cast<const uint32_t*>(c->layoutSizes.p);
    // (a caller's tree that names a triangle twice leaves a place of triLeaf without a record: an all-zero record is hit by no ray)
    if (assign && triCount) HIP_TRY(c, hipMemsetAsync(c->triLeaf.p, 0, size_t(triCount) * 64, c->stream));
    HIP_TRY(c, lrl_layout_node4(static_cast<const uint4*>(c->nodes32Raw.p), recCount, sz, sz + recCount, depth4 + 2, c->layoutScratch.p, static_cast<uint4*>(dst.p),
                                static_cast<uint32_t*>(c->posOfTri.p), triCount, assign, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));      // `node4` and `sizes` are borrowed only for this call
    wideNodes = sizes[0];
    return LRL_OK;
}
int gatherTriLeaf(LrlContext* c, uint32_t triCount) {
    HIP_TRY(c, lrl_launch_tri_leaf_gather(static_cast<const float4*>(c->triHot.p), static_cast<const uint32_t*>(c->posOfTri.p), 0, triCount,
                                          static_cast<float4*>(c->triLeaf.p), c->stream));
    return LRL_OK;
}

// the primitives' own boxes, for buildNode4 (callers' trees); both return whether a primitive left the box it had
bool ownBoxPut(float* q, const float mn[3], const float mx[3]) {
    bool left = false;
    for (int a = 0; a < 3; ++a) {
        left |= !(q[a] <= mn[a] && mx[a] <= q[3 + a]);
        q[a] = mn[a]; q[3 + a] = mx[a];
    }
    return left;
}
bool ownTriangleBoxes(LrlContext* c, uint32_t first, uint32_t count, const LrlTriangle* tris) {
    std::vector<float>& own = c->ownBox;
    if (own.size() < size_t(first + count) * 6) return true;      // (not kept for this scene)
    bool left = false;
    for (uint32_t i = 0; i < count; ++i) {
        const LrlTriangle& t = tris[i];
        float mn[3], mx[3];
        for (int a = 0; a < 3; ++a) {
            mn[a] = std::min(t.p1[a], std::min(t.p2[a], t.p3[a]));
            mx[a] = std::max(t.p1[a], std::max(t.p2[a], t.p3[a]));
        }
        left |= ownBoxPut(&own[size_t(first + i) * 6], mn, mx);
    }
    return left;
}
// Primitives changed under a caller's tree.  Its boxes are the contract again (the Pool demo never rebuilds: the water's leaf boxes
// were made tall enough for every wave, Pool/Pool.cpp:250-263), until the next tree arrives.
void ownBoxesStale(LrlContext* c) {
    if (c->haveLoose) c->looseInUse = true;
    c->pathsAtRest = 0;
}

// TrianglePre for [first, first+count): materials interned on the host, geometry on the device.
int runTrianglePre(LrlContext* c, uint32_t first, uint32_t count, const LrlTriangle* tris) {
    if (!count) return LRL_OK;
    std::vector<uint32_t> mat(count);
    size_t before = c->materialTable.size();
    for (uint32_t i = 0; i < count; ++i) mat[i] = internMaterial(c, tris[i].color);
    if (c->materialTable.size() != before || !c->materials.p) {
        int rc = upload(c, c->materials, c->materialTable.data(), c->materialTable.size() * sizeof(LrlColor));
        if (rc) return rc;
    }
    int rc = upload(c, c->triStage, tris, size_t(count) * sizeof(LrlTriangle));
    if (rc) return rc;
    rc = upload(c, c->matStage, mat.data(), size_t(count) * 4);
    if (rc) return rc;
    HIP_TRY(c, lrl_launch_triangle_pre(static_cast<const LrlTriangle*>(c->triStage.p), static_cast<const uint32_t*>(c->matStage.p), first,
                                       count, static_cast<float4*>(c->triHot.p), static_cast<LrlTriCold*>(c->triCold.p), static_cast<float4*>(c->triLeaf.p),
                                       c->leafOrder ? static_cast<const uint32_t*>(c->posOfTri.p) : nullptr, c->stream));
    if (c->deviceBuilt && c->haveScene)      // keep the boxes current for lrl_scene_rebuild_bvh
        HIP_TRY(c, lrl_launch_triangle_boxes(static_cast<const LrlTriangle*>(c->triStage.p), first, count, static_cast<float4*>(c->primMin.p),
                                             static_cast<float4*>(c->primMax.p), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // `mat` and the caller's array are borrowed only for this call
    return LRL_OK;
}

// boxes of the few primitives that are no triangles, with the leaf formulas of the host builder (SURVEY App. A:
// BVH.exe@0x140005180 spheres, @0x1400052e0 discs, @0x140005560 cylinders, @0x140005910 cones; 5 % slack on radii)
void boxFromDisc(const float c[3], const float n[3], float R, float mn[3], float mx[3]) {
    const float nl = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    for (int i = 0; i < 3; ++i) {
        const float ni = nl > 0 ? n[i] / nl : 0;
        const float e = float(1.05 * double(R) * std::sqrt(std::max(0.0, 1.0 - double(ni) * ni)));
        mn[i] = c[i] - e; mx[i] = c[i] + e;
    }
}

// leaf box of record i of a primitive class that is no triangle (kind = the BVH geometry code 3..6)
static void otherBox(uint32_t kind, const void* records, uint32_t i, float mn[3], float mx[3]) {
    if (kind == 3) {
        const LrlSphere& q = static_cast<const LrlSphere*>(records)[i];
        const float d = float(1.05 * double(std::sqrt(q.sphere[3])));
        for (int k = 0; k < 3; ++k) { mn[k] = q.sphere[k] - d; mx[k] = q.sphere[k] + d; }
    } else if (kind == 4) {
        const LrlCircle& q = static_cast<const LrlCircle*>(records)[i];
        boxFromDisc(q.sphere, q.plane, std::sqrt(q.sphere[3]), mn, mx);
    } else if (kind == 5) {
        const LrlCylinder& q = static_cast<const LrlCylinder*>(records)[i];
        boxFromDisc(q.c, q.n, std::sqrt(q.r2), mn, mx);
        for (int k = 0; k < 3; ++k) {
            const float sh = q.n[k] * q.l;
            mn[k] = std::min(mn[k], mn[k] + sh);
            mx[k] = std::max(mx[k], mx[k] + sh);
        }
    } else {
        const LrlCone& q = static_cast<const LrlCone*>(records)[i];
        const float h = std::sqrt(q.c2 * q.l2), R = std::sqrt((1 - q.c2) * q.l2);
        const float base[3] = {q.c[0] + q.n[0] * h, q.c[1] + q.n[1] * h, q.c[2] + q.n[2] * h};
        boxFromDisc(base, q.n, R, mn, mx);
        for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], q.c[k]); mx[k] = std::max(mx[k], q.c[k]); }
    }
}

int runDeviceBuild(LrlContext* c, uint32_t prims, uint32_t& recCount, uint32_t& depth);

int buildOnDevice(LrlContext* c, const LrlSceneView* s, uint32_t prims, uint32_t& recCount, uint32_t& depth) {
    const LrlGeometryNum& g = s->num;
    auto asf = [](uint32_t v) { float f; std::memcpy(&f, &v, 4); return f; };
    // the non-triangles' boxes, appended after the triangles'
    const uint32_t others = prims - g.triangleNum;
    std::vector<float4> omin(others), omax(others);
    uint32_t o = 0;
    auto put = [&](const float mn[3], const float mx[3], uint32_t type, uint32_t i) {
        omin[o] = make_float4(mn[0], mn[1], mn[2], asf((type << LRL_PRIM_TYPE_SHIFT) | i));
        omax[o] = make_float4(mx[0], mx[1], mx[2], 0.0f);
        ++o;
    };
    const void* const arrays[7] = {nullptr, nullptr, nullptr, s->spheres, s->circles, s->cylinders, s->cones};
    const uint32_t counts[7] = {0, 0, 0, g.sphereNum, g.circleNum, g.cylinderNum, g.coneNum};
    for (uint32_t kind = 3; kind <= 6; ++kind)
        for (uint32_t i = 0; i < counts[kind]; ++i) {
            float mn[3], mx[3];
            otherBox(kind, arrays[kind], i, mn, mx);
            put(mn, mx, kind, i);
        }
    if (others) {
        HIP_TRY(c, hipMemcpy(static_cast<float4*>(c->primMin.p) + g.triangleNum, omin.data(), size_t(others) * 16, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(static_cast<float4*>(c->primMax.p) + g.triangleNum, omax.data(), size_t(others) * 16, hipMemcpyHostToDevice));
    }
    return runDeviceBuild(c, prims, recCount, depth);
}

// the build proper, from the boxes in primMin / primMax (also lrl_scene_rebuild_bvh)
int runDeviceBuild(LrlContext* c, uint32_t prims, uint32_t& recCount, uint32_t& depth) {
    auto asf = [](uint32_t v) { float f; std::memcpy(&f, &v, 4); return f; };
    int rc;
    recCount = std::max(prims, 1u);
    // large trees are laid out in the order the rays walk them (lrl_layout_node4): built into a temporary by record index, then compacted
    const bool layout = prims >= LRL_LAYOUT_FROM && c->tuning.treeLayout == 0;
    DevBuf& raw = layout ? c->nodes32Raw : c->nodes32;
    if ((rc = ensure(c, c->nodes, size_t(recCount) * 64))) return rc;
    if ((rc = ensure(c, raw, size_t(recCount) * 64))) return rc;
    c->leafOrder = false;
    c->wideNodes = recCount;
    if (prims == 1) {      // one primitive: the virtual parent of a leaf
        float4 mn, mx;
        HIP_TRY(c, hipStreamSynchronize(c->stream));      // the boxes may come from a triangle update still in flight (lrl_scene_rebuild_bvh)
        HIP_TRY(c, hipMemcpy(&mn, c->primMin.p, 16, hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(&mx, c->primMax.p, 16, hipMemcpyDeviceToHost));
        uint32_t bits; std::memcpy(&bits, &mn.w, 4);
        const float4 rec[4] = {make_float4(mn.x, mn.y, mn.z, asf(LRL_REF_LEAF | bits)), make_float4(mx.x, mx.y, mx.z, asf(LRL_REF_NONE)),
                               make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
        uint32_t o3[3]; std::memcpy(o3, &mn, 12);
        // a single item: its box is the node's box, one byte step = the whole extent
        uint4 n4[4];
        uint32_t ex[3];
        const float lo[3] = {mn.x, mn.y, mn.z}, hi[3] = {mx.x, mx.y, mx.z};
        uint32_t hb[3];
        for (int a = 0; a < 3; ++a) {
            const double ext = double(hi[a]) - double(lo[a]);
            int e = -126;
            if (ext > 0) (void)std::frexp(ext / 255.0, &e);
            e = std::min(127, std::max(-126, e));
            ex[a] = uint32_t(e + 127) << 23;
            hb[a] = uint32_t(std::min(255.0, std::ceil(ext / std::ldexp(1.0, e))));
        }
        n4[0] = make_uint4(o3[0], o3[1], o3[2], ex[0]);
        n4[1] = make_uint4(0xffffff00u, 0xffffff00u, 0xffffff00u, hb[0]);      // slots 1..3 empty: inverted boxes
        n4[2] = make_uint4(hb[1], hb[2], LRL_REF_LEAF | bits, LRL_REF_EMPTY4);
        n4[3] = make_uint4(LRL_REF_EMPTY4, LRL_REF_EMPTY4, ex[1], ex[2]);
        HIP_TRY(c, hipMemcpy(c->nodes.p, rec, 64, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(c->nodes32.p, n4, 64, hipMemcpyHostToDevice));
        depth = 0;
        c->node4Depth = 1;
        return LRL_OK;
    }
    const size_t buildBytes = lrl_build_scratch_bytes(prims);
    if ((rc = ensure(c, c->buildScratch, buildBytes))) return rc;
    if ((rc = ensure(c, c->buildInfo, 256))) return rc;
    const uint32_t* sizes[2] = {nullptr, nullptr};
    HIP_TRY(c, lrl_build_bvh(static_cast<const float4*>(c->primMin.p), static_cast<const float4*>(c->primMax.p), prims, static_cast<float4*>(c->nodes.p),
                             static_cast<uint4*>(raw.p), c->buildScratch.p, buildBytes, static_cast<uint32_t*>(c->buildInfo.p),
                             c->tuning.deviceBuilder != 1u, sizes, c->stream));
    uint32_t info[5] = {0, 0, 0, 0, 0};
    HIP_TRY(c, hipMemcpyAsync(info, c->buildInfo.p, 20, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (info[2]) return fail(c, LRL_ERR_INVALID_ARG, "a primitive has a coordinate that is not finite (NaN or infinity): no tree can bound it");
    if (info[0] == 0 || info[1] == 0) return fail(c, LRL_ERR_DEVICE, "device BVH build did not reach the root");
    if (info[0] + 1 > LRL_MAX_BVH_DEPTH) return fail(c, LRL_ERR_BVH_DEPTH, "device-built BVH deeper than LRL_MAX_BVH_DEPTH");
    depth = info[0];
    c->node4Depth = info[1];
    if (layout) {
        const uint32_t triCount = c->num.triangleNum;
        if (!sizes[0] || info[3] == 0 || info[3] > recCount || info[4] > triCount) return fail(c, LRL_ERR_DEVICE, "device BVH build: the four-wide subtree sizes are not what a tree over these primitives has");
        if ((rc = ensure(c, c->nodes32, size_t(info[3]) * 64))) return rc;
        if ((rc = ensureLeafOrder(c, recCount, triCount))) return rc;
        HIP_TRY(c, lrl_layout_node4(static_cast<const uint4*>(raw.p), recCount, sizes[0], sizes[1], info[1] + 1, c->layoutScratch.p, static_cast<uint4*>(c->nodes32.p),
                                    static_cast<uint32_t*>(c->posOfTri.p), triCount, true, c->stream));
        if ((rc = gatherTriLeaf(c, triCount))) return rc;
        c->leafOrder = true;
        c->wideNodes = info[3];
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        releaseLayoutTemporaries(c);
    }
    if (buildBytes > (64u << 20)) release(c->buildScratch);      // the tree is built: only the records stay (small scratch is kept for rebuilds)
    return LRL_OK;
}

// this device's copy of `scratch` (each GPU holds its own, at its own address); resolved once per device, with that device current
constexpr int kMaxDevices = 64;
std::mutex g_scratchMutex;
void* g_scratchPtr[kMaxDevices] = {};
hipError_t noallocScratchFor(int device, void** out) {
    *out = nullptr;
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratchMutex);
    if (!g_scratchPtr[device]) {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return e;
        void* ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!ptr) return hipErrorInvalidSymbol;
        g_scratchPtr[device] = ptr;
    }
    *out = g_scratchPtr[device];
    return hipSuccess;
}

LrlTuning defaultTuning() {
    LrlTuning t{};
    t.handoverEarly = LRL_HANDOVER_AUTO; t.handoverLate = LRL_HANDOVER_AUTO; t.handoverListLen = 0;
    t.adapt = 1; t.groupLanes = 0; t.contSlots = 0; t.groupCapX = 0; t.sparseFrom = 5; t.compactFrom = -1; t.coopFetch = 0; t.groupQuad = 0; t.streamWords = 0; t.refillMin = 16; t.phasedTrips = 0; t.deviceBuilder = 0;
    t.workLow = 12.0f; t.pollLag = 1; t.traceRef = 0; t.batchSlots = 0; t.debug = 0; t.treeLayout = 0; t.deterministic = 0;
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

}  // namespace

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
    for (DevBuf* b : {&c->nodes32Raw, &c->layoutScratch, &c->layoutSizes, &c->triLeaf, &c->posOfTri, &c->primMin, &c->primMax, &c->buildScratch, &c->buildInfo, &c->waterPoints, &c->waterTris, &c->waterMat, &c->nodes, &c->nodes32, &c->nodes32Loose, &c->triHot, &c->triCold, &c->materials, &c->spheres, &c->circles, &c->cylinders, &c->cones, &c->lights,
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

// The caller's tree (pre-order 64-B own-box nodes) -> the two device layouts: 64-B inner records that hold both children's boxes, and the
// four-wide compressed nodes.  Also lrl_scene_update_bvh.
static int uploadCallerTree(LrlContext* c, const LrlSceneView* s, uint32_t& recCount) {
    int rc = LRL_OK;
    recCount = 1;
    const uint32_t N = s->bvhNodeNum;
    std::vector<uint32_t> recOf(N, 0);
    for (uint32_t i = 0; i < N; ++i) if (s->bvh[i].bound.leftChild) recOf[i] = recCount++;
    auto refOf = [&](uint32_t i) -> uint32_t {
        const LrlBVHNode& n = s->bvh[i];
        if (n.bound.leftChild) return recOf[i];
        return LRL_REF_LEAF | (n.geometry << LRL_PRIM_TYPE_SHIFT) | (n.geometryNum & LRL_PRIM_INDEX_MASK);
    };
    auto asf = [](uint32_t v) { float f; std::memcpy(&f, &v, 4); return f; };
    std::vector<float4> packed(size_t(recCount) * 4);
    {   // record 0: virtual parent of the root
        const LrlBVHNode& r = s->bvh[0];
        packed[0] = make_float4(r.bound.min[0], r.bound.min[1], r.bound.min[2], asf(refOf(0)));
        packed[1] = make_float4(r.bound.max[0], r.bound.max[1], r.bound.max[2], asf(LRL_REF_NONE));
        packed[2] = make_float4(0, 0, 0, asf(0));
        packed[3] = make_float4(0, 0, 0, asf(0));
    }
    for (uint32_t i = 0; i < N; ++i) {
        const LrlBVHNode& n = s->bvh[i];
        if (!n.bound.leftChild) continue;
        float4* q = &packed[size_t(recOf[i]) * 4];
        const LrlBVHNode& l = s->bvh[i + 1];
        uint32_t own = n.geometry ? ((n.geometry << LRL_PRIM_TYPE_SHIFT) | (n.geometryNum & LRL_PRIM_INDEX_MASK)) : 0u;
        q[0] = make_float4(l.bound.min[0], l.bound.min[1], l.bound.min[2], asf(refOf(i + 1)));
        if (n.bound.rightChild) {
            const LrlBVHNode& r = s->bvh[n.bound.rightChild];
            q[1] = make_float4(l.bound.max[0], l.bound.max[1], l.bound.max[2], asf(refOf(uint32_t(n.bound.rightChild))));
            q[2] = make_float4(r.bound.min[0], r.bound.min[1], r.bound.min[2], asf(own));
            q[3] = make_float4(r.bound.max[0], r.bound.max[1], r.bound.max[2], asf(n.axis));
        } else {
            q[1] = make_float4(l.bound.max[0], l.bound.max[1], l.bound.max[2], asf(LRL_REF_NONE));
            q[2] = make_float4(0, 0, 0, asf(own));
            q[3] = make_float4(0, 0, 0, asf(n.axis));
        }
    }
    rc = upload(c, c->nodes, packed.data(), packed.size() * sizeof(float4));
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // the same tree collapsed to four children per node with 8-bit child boxes, for the FAST lean kernel (lrl_lean.inc)
    uint32_t depth4 = 0;
    FittedBoxes fitted;
    const bool fit = fitOwnBoxes(s, c->ownBox, fitted);
    std::vector<uint4> node4 = buildNode4(s, fit ? &fitted : nullptr, recOf, recCount, depth4);
    const size_t tightened = fit ? fitted.count : 0;
    fitted = FittedBoxes();
    c->node4Depth = depth4;
    const uint32_t triCount = s->num.triangleNum;
    const bool layout = c->tuning.treeLayout == 0;
    if (layout) {
        if ((rc = layoutFromHost(c, node4, recCount, depth4, c->nodes32, true, triCount, c->wideNodes))) return rc;
    } else {
        if ((rc = upload(c, c->nodes32, node4.data(), node4.size() * sizeof(uint4)))) return rc;
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->wideNodes = recCount;
    }
    c->leafOrder = layout;
    c->haveLoose = c->looseInUse = false;
    c->parkedTight = uint32_t(tightened);
    if (fit) {      // the same tree with the caller's boxes, for when the primitives move under it (ownBoxesStale)
        node4 = buildNode4(s, nullptr, recOf, recCount, depth4);
        c->node4Depth = std::max(c->node4Depth, depth4);      // (the stacks are sized for either copy)
        uint32_t looseNodes = 0;
        if (layout) {
            if ((rc = layoutFromHost(c, node4, recCount, depth4, c->nodes32Loose, false, triCount, looseNodes))) return rc;      // (the triangles keep the places the fitted tree gave them)
        } else {
            if ((rc = upload(c, c->nodes32Loose, node4.data(), node4.size() * sizeof(uint4)))) return rc;
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
        c->haveLoose = true;
        if (c->callerTree.data() != s->bvh) c->callerTree.assign(s->bvh, s->bvh + s->bvhNodeNum);
    } else if (c->callerTree.data() != s->bvh) { c->callerTree.clear(); c->callerTree.shrink_to_fit(); }
    c->pathsAtRest = 0;
    node4 = std::vector<uint4>();
    releaseLayoutTemporaries(c);
    if (c->tuning.debug) std::fprintf(stderr, "lrl: the four-wide tree bounds %zu primitives by their own boxes; %u of %u node slots in use\n", tightened, c->wideNodes, recCount);
    return LRL_OK;
}

// A scene whose triangles moved under the caller's tree is fitted again when the rendering since the move is worth it: the fit costs about
// 0.2 us per node of the tree (on the host) and saves a third of a frame that costs about 0.5 ns per path, so it is done before the render
// that takes the paths since the move past 2048 per node (the Pool: 13.7 M paths, seven samples of 1080p: the paused Pool a few frames after
// the init-time sync of its water, Pool/Pool.cpp:648-650; a 64-sample frame right after a water step; never a loop of one-sample frames
// with a step each): the water's triangles, which moved on the device, are read back for
// their boxes (the others' are kept current by the update calls), and the fitted copy of the four-wide tree is made anew.  Renders in flight
// walk the other copy.
static int refitAtRest(LrlContext* c, uint64_t paths) {
    if (!c->looseInUse || c->callerTree.empty() || c->deviceBuilt || c->tuning.deterministic) return LRL_OK;
    c->pathsAtRest += paths;
    if (c->pathsAtRest < 2048ull * c->callerTree.size()) return LRL_OK;
    if (int rc0 = orderAfterRenders(c)) return rc0;      // (a frame in flight on another stream may still walk the copy that is about to be replaced)
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->waterGX && c->ownBox.size() >= size_t(c->num.triangleNum) * 6) {
        const uint32_t count = 2u * (c->waterGX - 1) * (c->waterGY - 1);
        std::vector<LrlTriangle> t(count);
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipMemcpy(t.data(), c->waterTris.p, size_t(count) * sizeof(LrlTriangle), hipMemcpyDeviceToHost));
        (void)ownTriangleBoxes(c, c->waterFirst, count, t.data());
    }
    LrlSceneView v{};
    v.num = c->num; v.bvh = c->callerTree.data(); v.bvhNodeNum = uint32_t(c->callerTree.size());
    uint32_t recCount = 1;
    std::vector<uint32_t> recOf(v.bvhNodeNum, 0);
    for (uint32_t i = 0; i < v.bvhNodeNum; ++i) if (v.bvh[i].bound.leftChild) recOf[i] = recCount++;
    FittedBoxes fitted;
    if (!fitOwnBoxes(&v, c->ownBox, fitted)) { c->pathsAtRest = 0; c->callerTree.clear(); c->callerTree.shrink_to_fit(); return LRL_OK; }      // (nothing to gain any more)
    uint32_t depth4 = 0;
    std::vector<uint4> node4 = buildNode4(&v, &fitted, recOf, recCount, depth4);
    int rc;
    if (c->leafOrder) rc = layoutFromHost(c, node4, recCount, depth4, c->nodes32, false, c->num.triangleNum, c->wideNodes);      // (the triangles stay where they are in triLeaf)
    else {
        rc = upload(c, c->nodes32, node4.data(), node4.size() * sizeof(uint4));
        if (!rc) HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    if (rc) return rc;
    c->node4Depth = std::max(c->node4Depth, depth4);
    c->parkedTight = uint32_t(fitted.count);
    c->looseInUse = false;
    c->pathsAtRest = 0;
    if (c->tuning.debug) std::fprintf(stderr, "lrl: the scene is at rest: the four-wide tree is fitted again (%zu own boxes)\n", fitted.count);
    return LRL_OK;
}

int lrl_scene_upload(LrlContext* c, const LrlSceneView* s) {
    if (c) { c->handoverThr.clear(); c->feedbackTick = 0; c->contIters = 0; c->phasedTrial.state = 0; c->phasedTrial.signature = 0; }      // the hand-over thresholds start again with a new scene (feedback still in flight is the old scene's: dropped)
    if (!c || !s) return LRL_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc0 = orderAfterRenders(c)) return rc0;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const LrlGeometryNum& g = s->num;
    if (g.triangleNum && !s->trianglesOrigin == !s->trianglesGPU)
        return fail(c, LRL_ERR_INVALID_ARG, "exactly one of trianglesOrigin / trianglesGPU must be given");
    if ((g.sphereNum && !s->spheres) || (g.circleNum && !s->circles) || (g.cylinderNum && !s->cylinders) || (g.coneNum && !s->cones) ||
        (g.pointLightNum && !s->pointLights))
        return fail(c, LRL_ERR_INVALID_ARG, "null array with non-zero count");
    const bool deviceBuild = s->bvh == nullptr;      // no tree from the caller: built on the device (lrl_build.hip)
    if (deviceBuild && g.triangleNum && !s->trianglesOrigin) return fail(c, LRL_ERR_INVALID_ARG, "the device BVH build needs trianglesOrigin");
    const uint64_t primTotal = uint64_t(g.triangleNum) + g.sphereNum + g.circleNum + g.cylinderNum + g.coneNum;
    if (deviceBuild && (primTotal == 0 || primTotal >= LRL_PRIM_INDEX_MASK)) return fail(c, LRL_ERR_INVALID_ARG, "the device BVH build needs 1 .. 2^28 primitives");
    if (g.triangleNum > LRL_PRIM_INDEX_MASK) return fail(c, LRL_ERR_LIMIT, "more than 2^28 triangles");
    if (s->texLayers && (!s->texBGR || !s->texWidth || !s->texHeight)) return fail(c, LRL_ERR_INVALID_ARG, "texture array incomplete");
    if (s->cubeSize) for (int i = 0; i < 6; ++i) if (!s->cubeBGR[i]) return fail(c, LRL_ERR_INVALID_ARG, "cube face missing");
    uint32_t depth = 0;
    int rc = deviceBuild ? LRL_OK : validateBVH(c, s, depth);
    if (rc) return rc;
    c->haveScene = false;
    c->deviceBuilt = false;
    c->waterGX = c->waterGY = 0;      // a new scene: lrl_water_init again

    // nodes: the reference's 64-B own-box records -> 64-B inner records that hold both children's boxes
    uint32_t recCount = 1;
    c->ownBox.clear(); c->ownBox.shrink_to_fit();
    if (deviceBuild) { c->callerTree.clear(); c->callerTree.shrink_to_fit(); }
    c->haveLoose = c->looseInUse = false;
    c->parkedTight = 0;
    c->leafOrder = false;      // (the device build lays its tree out after the triangles are there; a caller's tree before: TrianglePre then writes both copies)
    if (!deviceBuild) {
        if (g.triangleNum && s->trianglesOrigin) {
            c->ownBox.assign(size_t(g.triangleNum) * 6, 0.0f);
            (void)ownTriangleBoxes(c, 0, g.triangleNum, s->trianglesOrigin);
        }
    }
    if (!deviceBuild && (rc = uploadCallerTree(c, s, recCount))) return rc;
    if (!deviceBuild && c->tuning.debug) std::fprintf(stderr, "lrl: binary nodes %u (depth %u), four-wide tree depth %u\n", s->bvhNodeNum, depth, c->node4Depth);
    if (deviceBuild) {
        if ((rc = ensure(c, c->primMin, size_t(primTotal) * 16))) return rc;
        if ((rc = ensure(c, c->primMax, size_t(primTotal) * 16))) return rc;
    }
    c->materialTable.clear();
    c->materialIndex.clear();
    c->hasTransmission = false;
    for (uint32_t i = 0; i < g.sphereNum; ++i) if (transmits(s->spheres[i].color)) c->hasTransmission = true;
    for (uint32_t i = 0; i < g.circleNum; ++i) if (transmits(s->circles[i].color)) c->hasTransmission = true;
    for (uint32_t i = 0; i < g.cylinderNum; ++i) if (transmits(s->cylinders[i].color)) c->hasTransmission = true;
    for (uint32_t i = 0; i < g.coneNum; ++i) if (transmits(s->cones[i].color)) c->hasTransmission = true;
    rc = ensure(c, c->triHot, size_t(g.triangleNum) * 64);
    if (rc) return rc;
    rc = ensure(c, c->triCold, size_t(g.triangleNum) * sizeof(LrlTriCold));
    if (rc) return rc;
    rc = ensure(c, c->materials, sizeof(LrlColor));
    if (rc) return rc;
    c->num = g;
    if (g.triangleNum) {
        if (s->trianglesOrigin) {
            // in slabs, so the staging buffer stays modest for very large meshes
            const uint32_t slab = 1u << 22;
            for (uint32_t first = 0; first < g.triangleNum; first += slab) {
                const uint32_t cnt = std::min(slab, g.triangleNum - first);
                rc = runTrianglePre(c, first, cnt, s->trianglesOrigin + first);
                if (rc) return rc;
                if (deviceBuild) {      // the slab is still in the staging buffer
                    HIP_TRY(c, lrl_launch_triangle_boxes(static_cast<const LrlTriangle*>(c->triStage.p), first, cnt, static_cast<float4*>(c->primMin.p),
                                                         static_cast<float4*>(c->primMax.p), c->stream));
                    HIP_TRY(c, hipStreamSynchronize(c->stream));
                }
            }
        } else {
            std::vector<float4> hot(size_t(g.triangleNum) * 4);
            std::vector<LrlTriCold> cold(g.triangleNum);
            for (uint32_t i = 0; i < g.triangleNum; ++i) {
                const LrlTriangleGPU& t = s->trianglesGPU[i];
                hot[size_t(i) * 4 + 0] = make_float4(t.plane[0], t.plane[1], t.plane[2], t.plane[3]);
                hot[size_t(i) * 4 + 1] = make_float4(t.p1[0], t.p1[1], t.p1[2], t.k1[0]);
                hot[size_t(i) * 4 + 2] = make_float4(t.k1[1], t.k1[2], t.k2[0], t.k2[1]);
                float id; std::memcpy(&id, &i, 4);
                hot[size_t(i) * 4 + 3] = make_float4(t.k2[2], id, 0, 0);      // (the triangle's number travels in its hot record)
                LrlTriCold& q = cold[i];
                std::memcpy(q.uv1, t.uv1, 8); std::memcpy(q.uv2, t.uv2, 8); std::memcpy(q.uv3, t.uv3, 8);
                q.material = internMaterial(c, t.color); q._pad = 0;
            }
            rc = upload(c, c->triHot, hot.data(), hot.size() * sizeof(float4));
            if (rc) return rc;
            rc = upload(c, c->triCold, cold.data(), cold.size() * sizeof(LrlTriCold));
            if (rc) return rc;
            rc = upload(c, c->materials, c->materialTable.data(), c->materialTable.size() * sizeof(LrlColor));
            if (rc) return rc;
            if (c->leafOrder && (rc = gatherTriLeaf(c, g.triangleNum))) return rc;
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
    }
    if ((rc = upload(c, c->spheres, s->spheres, size_t(g.sphereNum) * sizeof(LrlSphere)))) return rc;
    if ((rc = upload(c, c->circles, s->circles, size_t(g.circleNum) * sizeof(LrlCircle)))) return rc;
    if ((rc = upload(c, c->cylinders, s->cylinders, size_t(g.cylinderNum) * sizeof(LrlCylinder)))) return rc;
    if ((rc = upload(c, c->cones, s->cones, size_t(g.coneNum) * sizeof(LrlCone)))) return rc;
    if ((rc = upload(c, c->lights, s->pointLights, size_t(g.pointLightNum) * sizeof(LrlPointLight)))) return rc;
    HIP_TRY(c, lrl_launch_circle_pre(static_cast<LrlCircle*>(c->circles.p), g.circleNum, c->stream));
    c->texW = c->texH = c->texL = 0;
    if (s->texLayers) {
        size_t texels = size_t(s->texWidth) * s->texHeight * s->texLayers;
        std::vector<uchar4> t = bgrToRgbx(s->texBGR, texels);
        if ((rc = upload(c, c->tex, t.data(), texels * 4))) return rc;
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->texW = s->texWidth; c->texH = s->texHeight; c->texL = s->texLayers;
    } else if ((rc = ensure(c, c->tex, 4))) return rc;
    c->cubeSize = 0;
    if (s->cubeSize) {
        size_t face = size_t(s->cubeSize) * s->cubeSize;
        std::vector<uchar4> t(face * 6);
        for (int f = 0; f < 6; ++f) {
            std::vector<uchar4> one = bgrToRgbx(s->cubeBGR[f], face);
            std::copy(one.begin(), one.end(), t.begin() + size_t(f) * face);
        }
        if ((rc = upload(c, c->cube, t.data(), t.size() * 4))) return rc;
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->cubeSize = s->cubeSize;
    } else if ((rc = ensure(c, c->cube, 4))) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (deviceBuild) {
        if ((rc = buildOnDevice(c, s, uint32_t(primTotal), recCount, depth))) return rc;
        if (c->tuning.debug) std::fprintf(stderr, "lrl: device-built tree over %llu primitives: %u records, height %u, four-wide height %u\n",
                                                   (unsigned long long)primTotal, recCount, depth, c->node4Depth);
    }
    c->deviceBuilt = deviceBuild;
    c->primTotal = uint32_t(primTotal);
    c->nodeNum = deviceBuild ? uint32_t(2 * primTotal - 1) : s->bvhNodeNum;
    c->recNum = recCount;
    c->bvhDepth = depth;
    c->haveScene = true;
    return LRL_OK;
}

int lrl_scene_update_triangles(LrlContext* c, uint32_t first, uint32_t count, const LrlTriangle* tris) {
    if (!c || (count && !tris)) return LRL_ERR_INVALID_ARG;
    if (!c->haveScene) return fail(c, LRL_ERR_NO_SCENE, "no scene uploaded");
    if (uint64_t(first) + count > c->num.triangleNum) return fail(c, LRL_ERR_INVALID_ARG, "triangle range out of bounds");
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc0 = orderAfterRenders(c)) return rc0;
    if (count && ownTriangleBoxes(c, first, count, tris)) ownBoxesStale(c);      // (the same triangles again, as the Pool demo sends them every frame: nothing moved)
    return runTrianglePre(c, first, count, tris);
}

int lrl_scene_update_triangles_noalloc(LrlContext* c, uint32_t first, uint32_t count, const LrlTriangle* tris) {
    if (!c || (count && !tris)) return LRL_ERR_INVALID_ARG;
    if (!c->haveScene) return fail(c, LRL_ERR_NO_SCENE, "no scene uploaded");
    if (uint64_t(first) + count > c->num.triangleNum) return fail(c, LRL_ERR_INVALID_ARG, "triangle range out of bounds");
    if (count > LRL_NOALLOC_MAX_TRIANGLES) return fail(c, LRL_ERR_LIMIT, "lrl_scene_update_triangles_noalloc: count exceeds LRL_NOALLOC_MAX_TRIANGLES (the static staging area); update in pieces or use lrl_scene_update_triangles");
    if (!count) return LRL_OK;
    if (!c->noallocScratch || !c->noallocPinned) return fail(c, LRL_ERR_DEVICE, "the static staging area was not resolved for this device");
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc0 = orderAfterRenders(c)) return rc0;      // (a stream-level wait: allocates nothing)
    // materials: looked up only; a new one would grow the table (host and device allocations)
    LrlTriangle* const pinTris = reinterpret_cast<LrlTriangle*>(c->noallocPinned);
    uint32_t* const pinMat = reinterpret_cast<uint32_t*>(c->noallocPinned + size_t(LRL_NOALLOC_MAX_TRIANGLES) * sizeof(LrlTriangle));
    for (uint32_t i = 0; i < count; ++i) {
        ColorKey k;
        std::memcpy(k.w, &tris[i].color, sizeof(k.w));
        auto it = c->materialIndex.find(k);
        if (it == c->materialIndex.end())
            return fail(c, LRL_ERR_LIMIT, "lrl_scene_update_triangles_noalloc: a triangle carries a material that is not in the scene's table (adding one allocates); use lrl_scene_update_triangles");
        pinMat[i] = it->second;
    }
    std::memcpy(pinTris, tris, size_t(count) * sizeof(LrlTriangle));
    if (ownTriangleBoxes(c, first, count, tris)) ownBoxesStale(c);      // (written in place: the vector has its size since the scene came)
    unsigned char* const area = static_cast<unsigned char*>(c->noallocScratch);
    LrlTriangle* const stageTris = reinterpret_cast<LrlTriangle*>(area);
    uint32_t* const stageMat = reinterpret_cast<uint32_t*>(area + size_t(LRL_NOALLOC_MAX_TRIANGLES) * sizeof(LrlTriangle));
    // the area keeps the last call's records beyond `count`: nothing reads them, the kernels stop at `count`
    HIP_TRY(c, hipMemcpyAsync(stageTris, pinTris, size_t(count) * sizeof(LrlTriangle), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(stageMat, pinMat, size_t(count) * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, lrl_launch_triangle_pre(stageTris, stageMat, first, count, static_cast<float4*>(c->triHot.p), static_cast<LrlTriCold*>(c->triCold.p),
                                       static_cast<float4*>(c->triLeaf.p), c->leafOrder ? static_cast<const uint32_t*>(c->posOfTri.p) : nullptr, c->stream));
    if (c->deviceBuilt)      // keep the boxes current for lrl_scene_rebuild_bvh
        HIP_TRY(c, lrl_launch_triangle_boxes(stageTris, first, count, static_cast<float4*>(c->primMin.p), static_cast<float4*>(c->primMax.p), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // the caller's array is borrowed only for this call; the staging area is free again
    return LRL_OK;
}

int lrl_scene_update_primitives(LrlContext* c, uint32_t kind, uint32_t first, uint32_t count, const void* records) {
    if (!c || (count && !records)) return LRL_ERR_INVALID_ARG;
    if (!c->haveScene) return fail(c, LRL_ERR_NO_SCENE, "no scene uploaded");
    const LrlGeometryNum& g = c->num;
    DevBuf* buf = nullptr;
    uint32_t have = 0;
    size_t size = 0;
    switch (kind) {
        case LRL_PRIM_SPHERE: buf = &c->spheres; have = g.sphereNum; size = sizeof(LrlSphere); break;
        case LRL_PRIM_CIRCLE: buf = &c->circles; have = g.circleNum; size = sizeof(LrlCircle); break;
        case LRL_PRIM_CYLINDER: buf = &c->cylinders; have = g.cylinderNum; size = sizeof(LrlCylinder); break;
        case LRL_PRIM_CONE: buf = &c->cones; have = g.coneNum; size = sizeof(LrlCone); break;
        case LRL_PRIM_LIGHT: buf = &c->lights; have = g.pointLightNum; size = sizeof(LrlPointLight); break;
        default: return fail(c, LRL_ERR_INVALID_ARG, "kind must be LRL_PRIM_SPHERE, _CIRCLE, _CYLINDER, _CONE or _LIGHT");
    }
    if (uint64_t(first) + count > have) return fail(c, LRL_ERR_INVALID_ARG, "primitive range out of bounds");
    if (!count) return LRL_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc0 = orderAfterRenders(c)) return rc0;
    HIP_TRY(c, hipStreamSynchronize(c->stream));      // frames in flight read these records
    HIP_TRY(c, hipMemcpy(static_cast<char*>(buf->p) + size_t(first) * size, records, size_t(count) * size, hipMemcpyHostToDevice));
    if (kind == LRL_PRIM_LIGHT) return LRL_OK;
    const char* const rec = static_cast<const char*>(records);
    for (uint32_t i = 0; i < count; ++i) {               // every geometry record ends with its Color
        const LrlColor& col = *reinterpret_cast<const LrlColor*>(rec + size_t(i + 1) * size - sizeof(LrlColor));
        if (transmits(col)) c->hasTransmission = true;
    }
    if (kind == LRL_PRIM_CIRCLE) {                       // CirclePre on the new records only (normalising twice is not the identity)
        HIP_TRY(c, lrl_launch_circle_pre(static_cast<LrlCircle*>(c->circles.p) + first, count, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    if (c->deviceBuilt) {                                // their leaf boxes, for lrl_scene_rebuild_bvh
        const uint32_t before[7] = {0, 0, 0, 0, g.sphereNum, g.sphereNum + g.circleNum, g.sphereNum + g.circleNum + g.cylinderNum};
        auto asf = [](uint32_t v) { float f; std::memcpy(&f, &v, 4); return f; };
        std::vector<float4> mnv(count), mxv(count);
        for (uint32_t i = 0; i < count; ++i) {
            float mn[3], mx[3];
            otherBox(kind, records, i, mn, mx);
            mnv[i] = make_float4(mn[0], mn[1], mn[2], asf((kind << LRL_PRIM_TYPE_SHIFT) | (first + i)));
            mxv[i] = make_float4(mx[0], mx[1], mx[2], 0.0f);
        }
        const size_t at = size_t(g.triangleNum) + before[kind] + first;
        HIP_TRY(c, hipMemcpy(static_cast<float4*>(c->primMin.p) + at, mnv.data(), size_t(count) * 16, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(static_cast<float4*>(c->primMax.p) + at, mxv.data(), size_t(count) * 16, hipMemcpyHostToDevice));
    }
    return LRL_OK;
}

int lrl_scene_update_bvh(LrlContext* c, const LrlBVHNode* bvh, uint32_t bvhNodeNum) {
    if (!c || !bvh) return LRL_ERR_INVALID_ARG;
    if (!c->haveScene) return fail(c, LRL_ERR_NO_SCENE, "no scene uploaded");
    if (c->deviceBuilt) return fail(c, LRL_ERR_INVALID_ARG, "the library built this scene's tree: lrl_scene_rebuild_bvh");
    HIP_TRY(c, hipSetDevice(c->device));
    LrlSceneView v{};
    v.num = c->num; v.bvh = bvh; v.bvhNodeNum = bvhNodeNum;
    uint32_t depth = 0, recCount = 1;
    int rc = validateBVH(c, &v, depth);
    if (rc) return rc;
    if ((rc = orderAfterRenders(c))) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));      // frames in flight walk the old tree
    c->haveScene = false;                             // (a failed upload leaves no half-written tree in use)
    if ((rc = uploadCallerTree(c, &v, recCount))) return rc;
    if (c->leafOrder && (rc = gatherTriLeaf(c, c->num.triangleNum))) return rc;      // the new tree has its own order of leaves
    c->nodeNum = bvhNodeNum; c->recNum = recCount; c->bvhDepth = depth;
    c->handoverThr.clear(); c->feedbackTick = 0; c->contIters = 0;
    c->phasedTrial.state = 0; c->phasedTrial.signature = 0;
    c->haveScene = true;
    return LRL_OK;
}

int lrl_scene_read_triangles_gpu(LrlContext* c, uint32_t first, uint32_t count, LrlTriangleGPU* out) {
    if (!c || (count && !out)) return LRL_ERR_INVALID_ARG;
    if (!c->haveScene) return fail(c, LRL_ERR_NO_SCENE, "no scene uploaded");
    if (uint64_t(first) + count > c->num.triangleNum) return fail(c, LRL_ERR_INVALID_ARG, "triangle range out of bounds");
    HIP_TRY(c, hipSetDevice(c->device));
    std::vector<float4> hot(size_t(count) * 4);
    std::vector<LrlTriCold> cold(count);
    HIP_TRY(c, hipStreamSynchronize(c->stream));      // TrianglePre of an update or a water step may still be in flight (the stream is non-blocking)
    if (count) {
        HIP_TRY(c, hipMemcpy(hot.data(), static_cast<const float4*>(c->triHot.p) + size_t(first) * 4, hot.size() * sizeof(float4), hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(cold.data(), static_cast<const LrlTriCold*>(c->triCold.p) + first, cold.size() * sizeof(LrlTriCold), hipMemcpyDeviceToHost));
    }
    for (uint32_t i = 0; i < count; ++i) {
        LrlTriangleGPU& t = out[i];
        std::memset(&t, 0, sizeof(t));
        const float4* h = &hot[size_t(i) * 4];
        t.plane[0] = h[0].x; t.plane[1] = h[0].y; t.plane[2] = h[0].z; t.plane[3] = h[0].w;
        t.p1[0] = h[1].x; t.p1[1] = h[1].y; t.p1[2] = h[1].z;
        t.k1[0] = h[1].w; t.k1[1] = h[2].x; t.k1[2] = h[2].y;
        t.k2[0] = h[2].z; t.k2[1] = h[2].w; t.k2[2] = h[3].x;
        std::memcpy(t.uv1, cold[i].uv1, 8); std::memcpy(t.uv2, cold[i].uv2, 8); std::memcpy(t.uv3, cold[i].uv3, 8);
        if (cold[i].material >= c->materialTable.size()) return fail(c, LRL_ERR_DEVICE, "material index corrupt");
        t.color = c->materialTable[cold[i].material];
    }
    return LRL_OK;
}

int lrl_scene_read_circles(LrlContext* c, uint32_t first, uint32_t count, LrlCircle* out) {
    if (!c || (count && !out)) return LRL_ERR_INVALID_ARG;
    if (!c->haveScene) return fail(c, LRL_ERR_NO_SCENE, "no scene uploaded");
    if (uint64_t(first) + count > c->num.circleNum) return fail(c, LRL_ERR_INVALID_ARG, "circle range out of bounds");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));      // CirclePre runs on the context's (non-blocking) stream
    if (count) HIP_TRY(c, hipMemcpy(out, static_cast<const LrlCircle*>(c->circles.p) + first, size_t(count) * sizeof(LrlCircle), hipMemcpyDeviceToHost));
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


// ---- wavefront pipeline: generate, then (trace, shade) per ray segment until no path is alive ----
#define LRL_LIST_KINDS 8u        /* ray kinds (closest + one shadow ray per light) the compacted lists are kept for */

// LrlTuning::debug >= 2: a line on stderr before and after every HIP call of the render path, so that a call that blocks names itself
#define CRUMB_TRY(c, what, expr)                                                        \
    do {                                                                                \
        if ((c)->tuning.debug >= 2) { std::fprintf(stderr, "lrl[%d]: > %s\n", (c)->device, what); std::fflush(stderr); } \
        HIP_TRY(c, expr);                                                               \
        if ((c)->tuning.debug >= 2) { std::fprintf(stderr, "lrl[%d]: < %s\n", (c)->device, what); std::fflush(stderr); } \
    } while (0)

// where the blocks of one pass (bsz samples of Ppix path slots each) sit in the arena; every size grows with bsz, so the layout of a
// smaller last batch fits the arena of a full one
struct WfLayout {
    size_t state, flags, rayQ, maskC, maskS, refr, ovf, alive, contHead, contRefs, rayList, listCount, total;
};
static WfLayout wfLayout(size_t P, uint32_t L, uint32_t refrEntries, uint32_t ovfN, uint32_t contSlots, uint32_t contStride, bool special) {
    WfLayout o{};
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t at = off; off += (bytes + 255) / 256 * 256; return at; };
    const size_t words = P / 64, chunks = words * (1 + size_t(L)) / LRL_WF_CHUNK_WORDS;
    o.state = take(P * 16 * 6); o.flags = take(P * 4); o.rayQ = take(P * (1 + size_t(L)) * 32);
    o.maskC = take(words * 8); o.maskS = take(words * 8);
    o.refr = take(P * 64 * refrEntries); o.ovf = take(chunks * ovfN * 512);
    o.alive = take(LRL_WF_MAX_ITERATIONS * 4 * 2 + LRL_FEEDBACK_ITERS * LRL_WORK_COUNTERS * 128);
    o.contHead = take(size_t(contSlots) * 16); o.contRefs = take(size_t(contSlots) * contStride * 4);
    o.rayList = take(special ? P * (1 + size_t(L)) * 4 : 0);
    o.listCount = take(special ? size_t(LRL_WF_MAX_ITERATIONS) * LRL_LIST_KINDS * 4 : 0);
    o.total = off;
    return o;
}

static int renderWavefront(LrlContext* c, const LrlRenderParams* p, const KScene& k, KFrame f, float* rgbaDev, uint32_t* hitDev,
                           unsigned long long* ctr, hipStream_t stream, uint32_t rows, hipEvent_t e0, hipEvent_t e1) {
    const LrlTuning& tn = c->tuning;
    const bool fast = p->mode == LRL_MODE_FAST, pool = p->variant == LRL_VARIANT_POOL;
    const uint32_t L = c->num.pointLightNum;
    const uint32_t tilesX = (c->width + 7) / 8, tilesY = (rows + 7) / 8;
    const uint64_t Ppix64 = (uint64_t(tilesX) * tilesY * 64 + LRL_WF_SLOT_ALIGN - 1) / LRL_WF_SLOT_ALIGN * LRL_WF_SLOT_ALIGN;
    if (Ppix64 * (1 + uint64_t(L)) >= (1ull << 31)) return fail(c, LRL_ERR_LIMIT, "pixels x (1 + lights) of one render exceeds 2^31 ray slots: render in row stripes");
    const size_t Ppix = size_t(Ppix64);
    // samples traced together: one generate / trace / shade pass carries `batch` samples (slot = b * Ppix + pixel slot), so that
    // the launches of a high-spp frame fill the chip (a 1080p sample is 2 M paths; the chip wants about 8 M rays per launch) and a frame
    // pays the tails of its launches once per batch instead of once per sample
    const uint32_t nSamples = f.sampleEnd - f.sampleBegin;
    // (measured, 1080p: C3 132.9 / 131.4 / 131.2 / 128.2 ms per 64-spp frame with batches of 4 / 8 / 16 / 32 samples, 148.8 with one; C5
    // 67.6 / 63.1 / 60.3 ms per 16-spp frame with 4 / 8 / 16).  Default: 2^25 path slots per launch, as long as the arena stays within 8 GiB
    const uint64_t batchSlots = tn.batchSlots ? tn.batchSlots : (1u << 25);
    uint32_t batch = uint32_t(std::min<uint64_t>(nSamples, std::max<uint64_t>(1, batchSlots / Ppix64)));
    while (batch > 1 && Ppix64 * batch * (1 + uint64_t(L)) >= (1ull << 31)) --batch;
    if (!tn.batchSlots) {
        const uint64_t perSlot = 16 * 6 + 4 + (1 + uint64_t(L)) * 36 + 64ull * (c->hasTransmission ? p->maxDepth : 0u);
        while (batch > 1 && Ppix64 * batch * perSlot > (8ull << 30)) --batch;
    }
    // Trace kernel per iteration.  PARITY and counting renders: wfTraceKernel (the reference's visit order).  FAST: the lean
    // one-lane-per-ray kernel over the four-wide tree, which hands rays that are still going after handoverTrips trips to the
    // lanes-per-ray kernel.
    const bool special = fast && !tn.traceRef && (!ctr || (p->flags & LRL_FLAG_COUNT_FETCHES));
    // deepest stack: one parked child per level of the binary tree (wfTraceKernel), three per level of the four-wide tree (lean)
    const uint32_t stackNeed = special ? std::max(c->bvhDepth + 1, 3 * c->node4Depth + 4) : c->bvhDepth + 1;
    const uint32_t ldsRows = special ? uint32_t(LRL_LEAN_STACK) : uint32_t(LRL_WF_STACK);
    const uint32_t ovfN = stackNeed > ldsRows ? stackNeed - ldsRows : 0u;
    const uint32_t refrEntries = c->hasTransmission ? p->maxDepth : 0u;
    const uint32_t contStride = stackNeed + 1;
    const uint32_t contSlots = tn.contSlots ? std::max(1024u, tn.contSlots)
                                            : uint32_t(std::min<uint64_t>(1u << 22, 262144ull * ((Ppix64 * batch + (1u << 21) - 1) >> 21)));
    const WfLayout full = wfLayout(Ppix * batch, L, refrEntries, ovfN, contSlots, contStride, special);
    int rc = ensure(c, c->wfArena, full.total);
    if (rc) return rc;
    char* const base = static_cast<char*>(c->wfArena.p);
    // base thresholds of the hand-over: a scene in the caches is best off with early hand-overs (C2: a plateau around 48, 48, 32, 16, 16);
    // beyond them a trip of the lean kernel and a round of the hand-over kernel both cost an HBM miss, and the lean kernel's ordered walk
    // needs fewer of them per ray: later (C5 with 16 lanes per ray: 40.7 ms with 48 / 16, 35.3 with 64 / 32, 34.5 with 96 / 48, 35.3 with 128 / 64)
    // A launch of many waves per slot (batches of samples, 4K frames) is bound less by its chains than a 1080p sample: in between (C4, 2^25
    // path slots a launch: 865 ms with 48 / 16, 797 with 64 / 32, 807 with 96 / 48, 844 without hand-over).
    const bool beyondCachesThr = size_t(c->recNum) * 64 > (size_t(1) << 30), bigLaunch = Ppix64 * batch >= (1ull << 23);
    const uint32_t regime = beyondCachesThr ? 2u : bigLaunch ? 1u : 0u;
    if (c->handoverRegime != regime) { c->handoverRegime = regime; c->handoverThr.clear(); c->feedbackTick = 0; c->contIters = 0; }
    const uint32_t handoverEarly = tn.handoverEarly != LRL_HANDOVER_AUTO ? tn.handoverEarly : regime == 2 ? 96u : regime == 1 ? 64u : 48u;
    const uint32_t handoverLate = tn.handoverLate != LRL_HANDOVER_AUTO ? tn.handoverLate : regime == 2 ? 48u : regime == 1 ? 32u : 16u;
    // The hand-over pays for rays with much work left.  Where most rays are of middling length (Pool: flat tree of tall
    // boxes) the slots run out, or the rays that leave are nearly done and cost more to move than to finish, so the
    // threshold of an iteration follows what that iteration did in an earlier pass: doubled (x4 when far beyond
    // the slots) while (nearly) all slots were asked for or the handed-over rays needed fewer than workLow frontier
    // entries each.  A ratchet: thresholds that came back down made the Pool oscillate between two states from sample to
    // sample (160 ms a frame against 150 for any pinned state); a new scene starts from the defaults.  adapt = 0 pins them.
    // lanes per handed-over ray: groups of 16 when a group's share of the frontier holds the hand-over plus a depth-first
    // descent (lrl_lean.inc), else a wave per ray
    // Round 2 measured 49.1 ms on C5 with a wave per ray against 60.0 with 16 lanes, and made the wave the default beyond the caches; with the
    // later hand-over above and 512-entry frontiers (capX below) it is the other way round, because what bounds the rounds there is their latency
    // and four rays a wave are four times the rays in flight: C5 44.3 ms with a wave per ray, 34.5 with 16 lanes (32 lanes: 35.0; 8: 34.3).
    const bool beyondCaches = size_t(c->recNum) * 64 > (size_t(1) << 30);
    uint32_t swarmGroup = tn.groupLanes ? tn.groupLanes : 16u;
    // Four lanes to a record fetch (lrl_lean.inc).  A uniformly random gather out of 16 GiB runs 2.3x faster that way (address translations
    // per lane bound it: tools/probes/hbm_gather.hip), but the rays of C5 are not that incoherent: 54.2 against 48.5 ms per frame (every kernel
    // of the frame ~15 % slower: idle lanes fetch too, 8 shuffles + 16 swaps a trip).  Off unless asked for.
    const bool coop = tn.coopFetch == 1;
    const bool groupQuad = tn.groupQuad == 1;      // lanes-per-ray kernel: four lanes to a frontier entry (the wave-per-ray kernel switches per round by itself)
    // Triangle trips (lrl_lean.inc, PH): -7 % on the Pool and -3 % on C5, +3-5 % on the coherent mirror rays of C2 / C4, and nothing the host knows
    // beforehand tells the two apart.  The frames are the same bit for bit either way, so the library times frames each way (events on the
    // caller's stream, read back without waiting) and keeps the faster for this scene and frame shape (two frames each way, alternating, after three that let the thresholds settle).  phasedTrips = 1 / 2 pins it.
    LrlContext::PhasedTrial& trial = c->phasedTrial;
    hipEvent_t trialEnd = nullptr;
    bool phased = false;
    uint32_t trialState = 0;
    const bool deterministic = tn.deterministic != 0;
    if (special && !coop) {
        if (tn.phasedTrips) phased = tn.phasedTrips == 1;
        else if (deterministic) phased = false;
        else {
            const uint64_t signature = (Ppix64 * batch) ^ (uint64_t(f.sampleEnd - f.sampleBegin) << 32) ^ (uint64_t(p->maxDepth) << 48) ^ (uint64_t(p->flags & LRL_FLAG_DIFFUSE_BOUNCE) << 52) ^
                                       (uint64_t(pool) << 57) ^ (uint64_t(L) << 58) ^ (ctr ? 1ull << 63 : 0ull);
            if (trial.signature != signature) { trial.signature = signature; trial.state = 0; }
            if (trial.state == 7 && hipEventQuery(trial.ev[7]) == hipSuccess) {
                float t[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                bool ok = true;
                for (int i = 0; i < 4; ++i) ok = ok && hipEventElapsedTime(&t[i], trial.ev[2 * i], trial.ev[2 * i + 1]) == hipSuccess;
                trial.choice = ok && t[1] + t[3] < 0.99f * (t[0] + t[2]);
                trial.state = 8;
                if (tn.debug == 1 || tn.debug >= 3) std::fprintf(stderr, "lrl: triangle trips %s (frames %.3f, %.3f ms with, %.3f, %.3f ms without)\n", trial.choice ? "on" : "off", t[1], t[3], t[0], t[2]);
            }
            if (trial.state >= 3 && trial.state <= 6) {
                const int k = trial.state - 3;
                phased = (k & 1) != 0;
                CRUMB_TRY(c, "hipEventRecord(trial start)", hipEventRecord(trial.ev[2 * k], stream));
                trialEnd = trial.ev[2 * k + 1];
            } else if (trial.state == 8) phased = trial.choice;
            if (trial.state < 7) ++trial.state;
            trialState = trial.state == 8 ? 2u : 1u;
        }
    }
    // lane refill (wfTraceLeanStreamKernel): waves that own streamWords words of a launch
    const uint32_t streamWords = tn.streamWords, refillMin = tn.refillMin ? std::min(64u, tn.refillMin) : 16u;
    if (swarmGroup != 8 && swarmGroup != 16 && swarmGroup != 32) swarmGroup = 64;
    // a ray's frontier: LRL_GROUP_CAP entries, twice that where the hand-over record + depth-first reserve (contStride) would
    // leave less than 2/5 of it for full rounds (groups of 16 lanes only)
    const uint32_t capX = tn.groupCapX ? tn.groupCapX : (swarmGroup == 16 && 2 * contStride + 3 * swarmGroup > LRL_GROUP_CAP - 32) ? 2u : 1u;      // C2 (stride 77): 256 is 1.3 % faster; 100 M device-built tree (95): 512 is 1.85x faster on C5
    const bool instantiated = (swarmGroup == 16 && (capX == 1 || capX == 2 || capX == 4)) || (swarmGroup == 32 && (capX == 1 || capX == 2 || capX == 4)) || (swarmGroup == 8 && capX == 1);
    const uint32_t groupCap = LRL_GROUP_CAP * (instantiated ? capX : 1u);
    if (swarmGroup != 64 && (!instantiated || 2 * contStride + 3 * swarmGroup > groupCap)) swarmGroup = 64;
    const bool handoverFits = 2 * contStride + 192 <= LRL_SWARM_CAP;      // else (trees hundreds of levels deep) the lean kernel finishes every ray itself
    const uint32_t sparseFrom = tn.sparseFrom;      // (measured: C2 best with none of its 5, C1 with 8 of its 56, C3 with 4 of its 10)
    const bool adapt = tn.adapt != 0 && !deterministic;
    auto baseThr = [&](uint32_t i) { return tn.handoverListLen ? tn.handoverList[std::min<uint32_t>(i, tn.handoverListLen - 1)] : i < 2 ? handoverEarly : i == 2 ? (handoverEarly + handoverLate) / 2 : handoverLate; };   // few rays left after the first bounces: the waves are free sooner (C2: a plateau around 48,48,32,16,16)
    if (c->handoverThr.empty()) {
        c->handoverThr.resize(LRL_FEEDBACK_ITERS);
        for (uint32_t i = 0; i < LRL_FEEDBACK_ITERS; ++i) c->handoverThr[i] = baseThr(i);
        c->swarmGrid.assign(LRL_FEEDBACK_ITERS, contSlots);
        c->handoverIdle.assign(LRL_FEEDBACK_ITERS, 0);
    }
    const float workLow = tn.workLow;
    c->workPerRay.resize(LRL_FEEDBACK_ITERS, 0.0f);
    // The counters of a pass reach the host without a copy: the shade kernel's first block stores them into pinned memory
    // (WFrame::feedbackHost) and an event recorded behind the pass says when they are all there.  (Round 1 copied them with
    // hipMemcpyAsync on the caller's stream; a device-to-host copy is the one call of the render path that may wait for the
    // stream to drain inside the runtime, and a caller's stream may itself be waiting for a collective.)
    auto feedback = [&]() {
        if (!c->contIters || hipEventQuery(c->evCont) != hipSuccess) return;
        const uint32_t slotsThen = c->contSlotsInFlight;
        for (uint32_t i = 0; i < c->contIters; ++i) {
            uint32_t& t = c->handoverThr[i];
            const volatile uint32_t* h = c->contHost + size_t(i) * 8;
            const uint32_t asked = h[0];
            c->handoverIdle[i] = asked == 0u;
            if (t == 0) continue;
            uint64_t work = 0;
            for (uint32_t w = 0; w < LRL_WORK_COUNTERS; ++w) work += h[1 + w];
            const uint32_t served = std::min(asked, slotsThen);
            const float perRay = served ? 16.0f * float(work) / float(served) : 0.0f;       // one wave in 16 reports
            c->workPerRay[i] = perRay;
            // a few dozen rays that were nearly done (the default scene: 20-90 rays an iteration with 2-5 entries each) are not worth a launch
            if (served < 1024u && perRay < workLow) c->handoverIdle[i] = 1;
            // rays that were nearly done when they left (a few frontier entries each) cost more to hand over than to finish
            // where they were: Pool's rays are all of middling length; the mesh scenes' long rays need 30-40 entries
            if (!adapt) {}
            else if (served >= 1024 && perRay < workLow) t = std::min(t * 2, 4096u);
            else if (asked > 4 * uint64_t(slotsThen)) t = std::min(t * 4, 4096u);
            else if (asked > slotsThen / 16 * 15) t = std::min(t * 2, 4096u);
            // dispatching 65536 workgroups that find nothing costs ~15 us a launch: twice what was asked for last time is
            // plenty (the kernel strides over the slots whatever the grid)
            c->swarmGrid[i] = std::max(2048u, uint32_t(std::min<uint64_t>(1u << 22, (2ull * asked + 1023u) / 1024u * 1024u)));
        }
        c->contIters = 0;
    };
    // compaction of the live rays before the trace launches of the late segments (compactFrom: first such iteration,
    // 0 = never).  It pays where the survivors are scattered over the mask words or a trip is expensive: after diffuse
    // bounces (C2 with them: 2.07 -> 1.60 ms) and on scenes far beyond the caches (100 M triangles: 2.84 -> 2.62 ms; C5:
    // 141.5 -> 132.6 ms); on C2 itself, the Pool and the default scene the extra kernel per iteration costs 1.5-10 %.
    // With more kinds of rays than LRL_LIST_KINDS (lights + 1) the masks are used throughout.
    const uint32_t compactFrom = (1 + L > LRL_LIST_KINDS) ? 0u : tn.compactFrom >= 0 ? uint32_t(tn.compactFrom)
                                 : (f.diffuse || beyondCaches) ? 2u : 0u;
#ifdef LRL_DIAG
    const char* tlPath = std::getenv("LRL_TIMELINE");          // diagnostic builds only: timeline of ONE trace launch
    const char* tlIter = std::getenv("LRL_TIMELINE_ITER");
    const uint32_t tlWanted = tlIter ? uint32_t(std::atoi(tlIter)) : 0u;
    const size_t tlEntries = std::max<size_t>(Ppix * batch * (1 + size_t(L)) / 16, 131072u + 65536u);      // lean: one per mask word; wave-per-ray: 131072 + slot
    if (tlPath && *tlPath && (rc = ensure(c, c->timelineDev, tlEntries * 48))) return rc;
#endif
    // every path ends after at most 2^(maxDepth+1) segments (each hit continues one reflection and parks one refraction)
    // (without a transmitting material nothing is ever parked: a path is primary + at most maxDepth reflections, and its
    // last segment casts no shadow rays, so iteration maxDepth finalises whatever is left)
    const uint32_t bound = c->hasTransmission ? uint32_t(std::min<uint64_t>(LRL_WF_MAX_ITERATIONS, (2ull << p->maxDepth) + 2)) : p->maxDepth + 1;
    const bool perKernel = c->timing && (p->flags & LRL_FLAG_TIME_KERNELS);
    const uint32_t pollLag = std::min(3u, std::max(1u, tn.pollLag));
    const bool poll = c->hasTransmission;    // refraction stacks make the number of segments data dependent
    if (e0) CRUMB_TRY(c, "hipEventRecord(frame start)", hipEventRecord(e0, stream));
    uint32_t iterations = 0, passIndex = 0;
    for (uint32_t s = f.sampleBegin; s < f.sampleEnd; ++passIndex) {
        const uint32_t bsz = std::min(batch, f.sampleEnd - s);
        const size_t P = Ppix * bsz;
        const uint32_t words = uint32_t(P / 64);
        const WfLayout lay = wfLayout(P, L, refrEntries, ovfN, contSlots, contStride, special);
        WFrame wf{};
        wf.P = uint32_t(P); wf.words = words; wf.tilesX = tilesX; wf.tilesY = tilesY; wf.overflowEntries = ovfN;
        wf.sample = s; wf.batch = bsz; wf.Ppix = uint32_t(Ppix); wf.blocksPerSample = uint32_t(Ppix / 256);
        float4* st = reinterpret_cast<float4*>(base + lay.state);
        wf.answer = st; wf.ratio = st + P; wf.decay = st + 2 * P; wf.hit = st + 3 * P;
        wf.pendA = st + 4 * P; wf.pendB = st + 5 * P;
        wf.flags = reinterpret_cast<uint32_t*>(base + lay.flags);
        wf.rayQ = reinterpret_cast<float4*>(base + lay.rayQ);
        wf.maskC = reinterpret_cast<unsigned long long*>(base + lay.maskC);
        wf.maskS = reinterpret_cast<unsigned long long*>(base + lay.maskS);
        wf.refr = refrEntries ? reinterpret_cast<float4*>(base + lay.refr) : nullptr;
        wf.overflow = reinterpret_cast<uint2*>(base + lay.ovf);
        wf.alive = reinterpret_cast<uint32_t*>(base + lay.alive);
        uint32_t* const contCounts = wf.alive + LRL_WF_MAX_ITERATIONS;      // one per iteration, zeroed with `alive`
        uint32_t* const contWorkAll = contCounts + LRL_WF_MAX_ITERATIONS;   // [iteration < LRL_FEEDBACK_ITERS][LRL_WORK_COUNTERS] 128 B apart, zeroed with `alive`
        wf.handoverTrips = handoverEarly;
        wf.contSlots = contSlots; wf.contStride = contStride; wf.groupCap = groupCap;
        wf.contHead = reinterpret_cast<uint4*>(base + lay.contHead);
        wf.contRefs = reinterpret_cast<uint32_t*>(base + lay.contRefs);
        wf.contCount = contCounts;
        wf.contWork = contWorkAll;
        wf.rayList = reinterpret_cast<uint32_t*>(base + lay.rayList);
        uint32_t* const listCounts = reinterpret_cast<uint32_t*>(base + lay.listCount);      // [iteration][LRL_LIST_KINDS]
        wf.listCount = listCounts;
        wf.out = reinterpret_cast<float4*>(rgbaDev);
        wf.hitOut = reinterpret_cast<uint2*>(hitDev);
        wf.counters = ctr;
        wf.timeline = nullptr;
        if (special && !deterministic) feedback();
        // the feedback need not be fresh: the first passes after a scene upload, then one pass in eight (and never two in flight)
        const bool report = special && !deterministic && c->contIters == 0 && (c->feedbackTick++ < 8 || (c->feedbackTick & 7u) == 0);
        wf.feedbackHost = report ? c->contHostDev : nullptr;
        if (report) std::memset(c->contHost, 0, size_t(LRL_FEEDBACK_ITERS) * 32);
        // one fill (a size that is no multiple of 16 B costs a second fill kernel): alive, hand-over counts, work counters
        CRUMB_TRY(c, "hipMemsetAsync(alive, counters)", hipMemsetAsync(wf.alive, 0, special ? size_t(LRL_WF_MAX_ITERATIONS) * 8 + size_t(LRL_FEEDBACK_ITERS) * LRL_WORK_COUNTERS * 128
                                                                                            : size_t(LRL_WF_MAX_ITERATIONS) * 4, stream));
        if (special && compactFrom != 0 && bound > compactFrom)
            CRUMB_TRY(c, "hipMemsetAsync(list counts)", hipMemsetAsync(listCounts, 0, size_t(bound) * LRL_LIST_KINDS * 4, stream));
        // polled words: the half of the pinned array that pass - 2 used (the host has seen that pass end)
        uint32_t* const aliveWords = c->aliveHost + size_t(passIndex & 1u) * LRL_WF_MAX_ITERATIONS;
        wf.aliveHost = poll ? c->aliveHostDev + size_t(passIndex & 1u) * LRL_WF_MAX_ITERATIONS : nullptr;
        if (poll) std::memset(aliveWords, 0, size_t(bound) * 4);
        CRUMB_TRY(c, "generate", fast ? lrl_fast::wfLaunchGenerate(k, f, wf, stream) : lrl_parity::wfLaunchGenerate(k, f, wf, stream));
        uint32_t iter = 0;
        for (; iter < bound; ++iter) {
#ifdef LRL_DIAG
            const bool tl = tlPath && *tlPath && iter == tlWanted && s == f.sampleBegin;
            if (tl) { HIP_TRY(c, hipMemsetAsync(c->timelineDev.p, 0, tlEntries * 48, stream)); wf.timeline = static_cast<unsigned long long*>(c->timelineDev.p); }
#endif
            hipEvent_t ta = nullptr, tb = nullptr;
            if (perKernel && c->timingUsed < LRL_MAX_TIMED_LAUNCHES) {
                if (c->timingUsed == c->timingEvents.size()) {
                    HIP_TRY(c, hipEventCreate(&ta));
                    HIP_TRY(c, hipEventCreate(&tb));
                    c->timingEvents.emplace_back(ta, tb);
                }
                ta = c->timingEvents[c->timingUsed].first; tb = c->timingEvents[c->timingUsed].second;
                ++c->timingUsed;
                HIP_TRY(c, hipEventRecord(ta, stream));
            }
            if (special) {
                const uint32_t fi = std::min(iter, LRL_FEEDBACK_ITERS - 1);
                wf.contCount = contCounts + (iter < LRL_WF_MAX_ITERATIONS ? iter : 0);
                // an iteration whose rays all finished below the threshold at the last feedback runs without hand-over (the same rays, finished
                // by the lean kernel as they were anyway) and without the launch that would find nothing: the default scene has 56 such
                // iterations a frame, 9 us each; every feedback pass looks again
                wf.handoverTrips = (handoverFits && (report || !c->handoverIdle[fi])) ? c->handoverThr[fi] : 0u;
                wf.contWork = contWorkAll + size_t(fi) * LRL_WORK_COUNTERS * 32;
                const bool compact = compactFrom != 0 && iter >= compactFrom && iter < LRL_WF_MAX_ITERATIONS;
                wf.listCount = listCounts + size_t(iter < LRL_WF_MAX_ITERATIONS ? iter : 0) * LRL_LIST_KINDS;
                CRUMB_TRY(c, "trace (lean)", lrl_fast::wfLaunchTraceLean(k, wf, iter >= sparseFrom, compact, coop, phased, streamWords, refillMin, stream));
                if (wf.handoverTrips) CRUMB_TRY(c, "trace (hand-over)", lrl_fast::wfLaunchTraceSwarm(k, wf, std::min(contSlots, c->swarmGrid[fi]), swarmGroup, coop, groupQuad, stream));
            }
            else CRUMB_TRY(c, "trace", fast ? lrl_fast::wfLaunchTrace(k, wf, stream) : lrl_parity::wfLaunchTrace(k, wf, stream));
            if (tb) HIP_TRY(c, hipEventRecord(tb, stream));
#ifdef LRL_DIAG
            if (tl) {
                wf.timeline = nullptr;
                std::vector<unsigned long long> h(tlEntries * 6);
                HIP_TRY(c, hipStreamSynchronize(stream));
                HIP_TRY(c, hipMemcpy(h.data(), c->timelineDev.p, h.size() * 8, hipMemcpyDeviceToHost));
                if (FILE* fp = std::fopen(tlPath, "wb")) { std::fwrite(h.data(), 8, h.size(), fp); std::fclose(fp); }
            }
#endif
            CRUMB_TRY(c, "shade", fast ? lrl_fast::wfLaunchShade(k, f, wf, iter, pool, stream) : lrl_parity::wfLaunchShade(k, f, wf, iter, pool, stream));
            if (!poll) continue;   // exact bound: nothing to ask the device, the whole frame is enqueued without a host wait
            CRUMB_TRY(c, "hipEventRecord(iteration)", hipEventRecord(c->evIter[iter & 3], stream));
            if (iter >= pollLag) {      // look at an earlier iteration while this one runs: the pipeline ends pollLag (empty) iterations late
                CRUMB_TRY(c, "hipEventSynchronize(iteration)", hipEventSynchronize(c->evIter[(iter - pollLag) & 3]));
                if (reinterpret_cast<volatile uint32_t*>(aliveWords)[iter - pollLag] == 0) { ++iter; break; }
            }
        }
        if (poll && iter == bound) {      // the last iteration has not been looked at yet
            CRUMB_TRY(c, "hipEventSynchronize(last iteration)", hipEventSynchronize(c->evIter[(bound - 1) & 3]));
            if (reinterpret_cast<volatile uint32_t*>(aliveWords)[bound - 1] != 0) return fail(c, LRL_ERR_LIMIT, "wavefront pipeline did not finish within its iteration bound");
        }
        if (bsz > 1) CRUMB_TRY(c, "accumulate", fast ? lrl_fast::wfLaunchAccumulate(f, wf, stream) : lrl_parity::wfLaunchAccumulate(f, wf, stream));
        iterations += iter;
        if (report) {
            c->contIters = std::min(iter, LRL_FEEDBACK_ITERS);
            c->contSlotsInFlight = contSlots;
            CRUMB_TRY(c, "hipEventRecord(feedback)", hipEventRecord(c->evCont, stream));
        }
        if ((tn.debug == 1 || tn.debug >= 3) && special) {      // (2: only the lines around the HIP calls, no added waits)
            std::vector<uint32_t> cc(iter);
            HIP_TRY(c, hipStreamSynchronize(stream));
            HIP_TRY(c, hipMemcpy(cc.data(), contCounts, iter * 4, hipMemcpyDeviceToHost));
            std::fprintf(stderr, "lrl: samples %u..%u, handed-over rays per iteration (threshold, entries per ray at the last feedback):", s, s + bsz);
            for (uint32_t i = 0; i < iter; ++i) std::fprintf(stderr, " %u (%u, %.0f)", cc[i], c->handoverThr[std::min(i, LRL_FEEDBACK_ITERS - 1)], c->workPerRay[std::min(i, LRL_FEEDBACK_ITERS - 1)]);
            std::fprintf(stderr, "  [%u lanes per handed-over ray, stride %u, %u slots]\n", swarmGroup, contStride, contSlots);
        }
        s += bsz;
    }
    if (trialEnd) CRUMB_TRY(c, "hipEventRecord(trial end)", hipEventRecord(trialEnd, stream));
    if (e1) CRUMB_TRY(c, "hipEventRecord(frame end)", hipEventRecord(e1, stream));
    c->lastIterations = iterations;
    LrlRenderInfo& ri = c->renderInfo;
    ri = LrlRenderInfo{};
    ri.iterations = iterations; ri.passes = passIndex; ri.batch = batch;
    if (special) {
        ri.compactFrom = compactFrom; ri.coopFetch = coop ? 1u : 0u; ri.streamWords = coop ? 0u : streamWords; ri.phasedTrips = phased ? 1u : 0u; ri.groupLanes = swarmGroup; ri.frontierEntries = swarmGroup == 64 ? LRL_SWARM_CAP : groupCap;
        ri.contSlots = contSlots; ri.contStride = contStride;
        for (uint32_t i = 0; i < 8; ++i) ri.handoverTrips[i] = handoverFits ? c->handoverThr[i] : 0u;
        ri.trialState = trialState;
    }
    return LRL_OK;
}

static int renderImpl(LrlContext* c, const LrlRenderParams* p, float* rgbaDev, uint32_t* hitDev, hipStream_t stream, uint32_t rows) {
    const uint32_t spp = p->spp;
    uint32_t s0 = p->sampleBegin, s1 = p->sampleEnd;
    if (s0 == 0 && s1 == 0) s1 = spp;
    if (int rcFit = refitAtRest(c, uint64_t(s1 - s0) * c->width * rows)) return rcFit;
    // per-sample camera: the jitter is the shear of `trans`; computed here in float like the host helper
    std::vector<LrlTrans> tr(spp, c->trans);
    if (p->jitter) {
        for (uint32_t s = 0; s < spp; ++s) {
            float dx = p->jitter[s * 2], dy = p->jitter[s * 2 + 1];
            if (dx == 0.0f && dy == 0.0f) continue;
            for (int i = 0; i < 3; ++i)
                tr[s].trans[i][2] = c->trans.trans[i][2] + (c->trans.trans[i][0] * (2.0f * dx / c->trans.z0) + c->trans.trans[i][1] * (2.0f * dy / c->trans.z0));
        }
    }
    int rc = LRL_OK;
    // one arena, one set of per-sample cameras and of polled words per context: a render on another stream than the last one
    // waits (on the device) for that one to end
    if (c->renderedOnce && stream != c->lastRenderStream) CRUMB_TRY(c, "hipStreamWaitEvent(last render)", hipStreamWaitEvent(stream, c->evLastRender, 0));
    // scene changes still queued on the context's stream (lrl_water_step, lrl_scene_rebuild_bvh return before their kernels end)
    if (stream != c->stream) {
        CRUMB_TRY(c, "hipEventRecord(scene ready)", hipEventRecord(c->evSceneReady, c->stream));
        CRUMB_TRY(c, "hipStreamWaitEvent(scene ready)", hipStreamWaitEvent(stream, c->evSceneReady, 0));
    }
    // the per-sample cameras go to the device only when they changed: an unchanged camera costs no copy and no host wait,
    // so consecutive frames queue back to back
    if (c->transUploaded.size() != tr.size() || std::memcmp(c->transUploaded.data(), tr.data(), tr.size() * sizeof(LrlTrans)) != 0) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));      // renders in flight still read the old ones
        if (stream != c->stream) HIP_TRY(c, hipStreamSynchronize(stream));
        c->transUploaded.clear();
        if ((rc = ensure(c, c->transDev, tr.size() * sizeof(LrlTrans)))) return rc;
        HIP_TRY(c, hipMemcpyAsync(c->transDev.p, tr.data(), tr.size() * sizeof(LrlTrans), hipMemcpyHostToDevice, stream));
        HIP_TRY(c, hipStreamSynchronize(stream));   // `tr` is a local
        c->transUploaded = tr;
    }
    unsigned long long* ctr = nullptr;
    if (p->flags & (LRL_FLAG_COUNTERS | LRL_FLAG_COUNT_FETCHES)) {
        if ((rc = ensure(c, c->counterDev, LRL_CTR_COUNT * 8))) return rc;
        HIP_TRY(c, hipMemsetAsync(c->counterDev.p, 0, LRL_CTR_COUNT * 8, stream));
        ctr = static_cast<unsigned long long*>(c->counterDev.p);
    }
    KScene k;
    fillKScene(c, k);
    KFrame f;
    f.trans = static_cast<const LrlTrans*>(c->transDev.p);
    f.width = c->width; f.height = c->height;
    f.rowBegin = p->rowBegin; f.rowEnd = p->rowBegin + rows;
    f.rows = rows;
    f.interleave = p->rowInterleave > 1 ? p->rowInterleave : 0u;
    f.shard = f.interleave ? p->rowShard : 0u;
    if (f.interleave) { f.rowBegin = 0; f.rowEnd = c->height; }
    f.alpha = float(s1 - s0) / float(spp);      // 1 exactly for a whole frame
    f.sampleBegin = s0; f.sampleEnd = s1; f.spp = spp;
    f.maxDepth = p->maxDepth;
    f.stackEntries = std::min(c->bvhDepth + 1, LRL_LDS_STACK_ENTRIES);
    f.timeline = nullptr;
    f.diffuse = (p->flags & LRL_FLAG_DIFFUSE_BOUNCE) ? 1u : 0u;
    if (f.diffuse && (p->flags & LRL_FLAG_MEGAKERNEL)) return fail(c, LRL_ERR_INVALID_ARG, "LRL_FLAG_DIFFUSE_BOUNCE is implemented in the wavefront pipeline only");
#ifdef LRL_DIAG
    const char* tlPath = std::getenv("LRL_TIMELINE");   // diagnostic builds: per-wave start/end ticks of this launch, dumped to the named file
#else
    const char* tlPath = nullptr;
#endif
    size_t tlWaves = 0;
    if (tlPath && *tlPath && (p->flags & LRL_FLAG_MEGAKERNEL)) {
        uint32_t tx = (c->width + LRL_TILE - 1) / LRL_TILE, ty = (rows + LRL_TILE - 1) / LRL_TILE;
        tlWaves = size_t(tx) * ty * (LRL_BLOCK_THREADS / 64);
        if ((rc = ensure(c, c->timelineDev, tlWaves * 8 * LRL_TIMELINE_WORDS))) return rc;
        HIP_TRY(c, hipMemsetAsync(c->timelineDev.p, 0, tlWaves * 8 * LRL_TIMELINE_WORDS, stream));
        f.timeline = static_cast<unsigned long long*>(c->timelineDev.p);
    }
    const bool mega = (p->flags & LRL_FLAG_MEGAKERNEL) != 0;
    const bool perKernel = !mega && c->timing && (p->flags & LRL_FLAG_TIME_KERNELS);
    hipEvent_t e0 = c->evStart, e1 = c->evStop;
    if (c->timing && !perKernel && c->timingUsed < LRL_MAX_TIMED_LAUNCHES) {
        if (c->timingUsed == c->timingEvents.size()) {
            hipEvent_t a, b;
            HIP_TRY(c, hipEventCreate(&a));
            HIP_TRY(c, hipEventCreate(&b));
            c->timingEvents.emplace_back(a, b);
        }
        e0 = c->timingEvents[c->timingUsed].first; e1 = c->timingEvents[c->timingUsed].second;
        ++c->timingUsed;
    }
    auto finish = [&]() -> int {
        CRUMB_TRY(c, "hipEventRecord(last render)", hipEventRecord(c->evLastRender, stream));
        c->lastRenderStream = stream;
        c->renderedOnce = true;
        return LRL_OK;
    };
    if (!mega) {
        if ((rc = renderWavefront(c, p, k, f, rgbaDev, hitDev, ctr, stream, rows, e0, e1))) return rc;
        c->timed = (e0 == c->evStart);
        return finish();
    }
    HIP_TRY(c, hipEventRecord(e0, stream));
    if (p->mode == LRL_MODE_FAST)
        HIP_TRY(c, lrl_launch_render_fast(k, f, reinterpret_cast<float4*>(rgbaDev), reinterpret_cast<uint2*>(hitDev), ctr, p->variant == LRL_VARIANT_POOL, stream));
    else
        HIP_TRY(c, lrl_launch_render_parity(k, f, reinterpret_cast<float4*>(rgbaDev), reinterpret_cast<uint2*>(hitDev), ctr, p->variant == LRL_VARIANT_POOL, stream));
    HIP_TRY(c, hipEventRecord(e1, stream));
    c->timed = (e0 == c->evStart);
    if (f.timeline) {
        std::vector<unsigned long long> h(tlWaves * LRL_TIMELINE_WORDS);
        HIP_TRY(c, hipStreamSynchronize(stream));
        HIP_TRY(c, hipMemcpy(h.data(), c->timelineDev.p, tlWaves * 8 * LRL_TIMELINE_WORDS, hipMemcpyDeviceToHost));
        if (FILE* fp = std::fopen(tlPath, "wb")) { std::fwrite(h.data(), 8, h.size(), fp); std::fclose(fp); }
    }
    return finish();
}

static int checkRender(LrlContext* c, const LrlRenderParams* p, uint32_t& rows) {
    if (!c || !p) return LRL_ERR_INVALID_ARG;
    if (!c->haveScene || !c->haveCamera) return fail(c, LRL_ERR_NO_SCENE, "scene and camera must be set before rendering");
    if (p->maxDepth < 1 || p->maxDepth > LRL_MAX_RAY_DEPTH) return fail(c, LRL_ERR_LIMIT, "maxDepth must be 1..LRL_MAX_RAY_DEPTH");
    if (p->variant > LRL_VARIANT_POOL || p->mode > LRL_MODE_FAST) return fail(c, LRL_ERR_INVALID_ARG, "unknown variant / mode");
    if ((p->flags & LRL_FLAG_COUNT_FETCHES) && (p->mode != LRL_MODE_FAST || (p->flags & (LRL_FLAG_COUNTERS | LRL_FLAG_MEGAKERNEL))))
        return fail(c, LRL_ERR_INVALID_ARG, "LRL_FLAG_COUNT_FETCHES goes with LRL_MODE_FAST, without LRL_FLAG_COUNTERS / LRL_FLAG_MEGAKERNEL");
    if (p->spp < 1 || p->spp > LRL_MAX_SPP) return fail(c, LRL_ERR_LIMIT, "spp must be 1..LRL_MAX_SPP");
    uint32_t r0 = p->rowBegin, r1 = p->rowEnd;
    if (r0 == 0 && r1 == 0) r1 = c->height;
    if (r0 >= r1 || r1 > c->height) return fail(c, LRL_ERR_INVALID_ARG, "row range invalid");
    if (p->rowInterleave > 1) {
        if (p->rowBegin || p->rowEnd) return fail(c, LRL_ERR_INVALID_ARG, "rowInterleave goes with rowBegin = rowEnd = 0");
        if (p->rowShard >= p->rowInterleave) return fail(c, LRL_ERR_INVALID_ARG, "rowShard must be below rowInterleave");
        if (p->flags & LRL_FLAG_MEGAKERNEL) return fail(c, LRL_ERR_INVALID_ARG, "rowInterleave is implemented in the wavefront pipeline only");
        r0 = 0;
        r1 = lrl_shard_rows(c->height, p->rowInterleave, p->rowShard);
        if (r1 == 0) return fail(c, LRL_ERR_INVALID_ARG, "this shard holds no rows of the frame (more shards than stripes)");
    }
    if (!(p->sampleBegin == 0 && p->sampleEnd == 0) && (p->sampleBegin >= p->sampleEnd || p->sampleEnd > p->spp))
        return fail(c, LRL_ERR_INVALID_ARG, "sample range invalid");
    rows = r1 - r0;
    return LRL_OK;
}

uint32_t lrl_shard_rows(uint32_t height, uint32_t interleave, uint32_t shard) {
    if (interleave <= 1) return shard == 0 ? height : 0u;
    if (shard >= interleave) return 0u;
    const uint32_t stripes = (height + LRL_STRIPE_ROWS - 1) / LRL_STRIPE_ROWS;
    if (shard >= stripes) return 0u;
    const uint32_t mine = (stripes - shard + interleave - 1) / interleave;      // stripes shard, shard + N, ...
    const uint32_t last = shard + (mine - 1) * interleave;                      // this shard's last stripe
    uint32_t rows = mine * LRL_STRIPE_ROWS;
    if (last == stripes - 1) rows -= stripes * LRL_STRIPE_ROWS - height;       // the frame's last stripe may be short
    return rows;
}

int lrl_render_async(LrlContext* c, const LrlRenderParams* p, float* rgbaDev, uint32_t* hitDev, void* hipStream) {
    uint32_t rows = 0;
    int rc = checkRender(c, p, rows);
    if (rc) return rc;
    if (!rgbaDev) return fail(c, LRL_ERR_INVALID_ARG, "rgba_out is null");
    HIP_TRY(c, hipSetDevice(c->device));
    return renderImpl(c, p, rgbaDev, hitDev, hipStream ? static_cast<hipStream_t>(hipStream) : c->stream, rows);
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

int lrl_render(LrlContext* c, const LrlRenderParams* p, float* rgba, uint32_t* hit) {
    uint32_t rows = 0;
    int rc = checkRender(c, p, rows);
    if (rc) return rc;
    if (!rgba) return fail(c, LRL_ERR_INVALID_ARG, "rgba_out is null");
    HIP_TRY(c, hipSetDevice(c->device));
    const bool dev = (p->flags & LRL_FLAG_OUT_DEVICE) != 0;
    const size_t px = size_t(rows) * c->width;
    float* o = rgba;
    uint32_t* h = hit;
    if (!dev) {
        if ((rc = ensure(c, c->outDev, px * 16))) return rc;
        o = static_cast<float*>(c->outDev.p);
        if (hit) {
            if ((rc = ensure(c, c->hitDev, px * 8))) return rc;
            h = static_cast<uint32_t*>(c->hitDev.p);
        }
    }
    rc = renderImpl(c, p, o, h, c->stream, rows);
    if (rc) return rc;
    if (!dev) {
        HIP_TRY(c, hipMemcpyAsync(rgba, o, px * 16, hipMemcpyDeviceToHost, c->stream));
        if (hit) HIP_TRY(c, hipMemcpyAsync(hit, h, px * 8, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (p->flags & (LRL_FLAG_COUNTERS | LRL_FLAG_COUNT_FETCHES)) {
        unsigned long long v[LRL_CTR_COUNT];
        HIP_TRY(c, hipMemcpy(v, c->counterDev.p, sizeof(v), hipMemcpyDeviceToHost));
        c->counters.raysPrimary = v[LRL_CTR_PRIMARY]; c->counters.raysReflect = v[LRL_CTR_REFLECT];
        c->counters.raysRefract = v[LRL_CTR_REFRACT]; c->counters.raysShadow = v[LRL_CTR_SHADOW];
        c->counters.nodeVisits = v[LRL_CTR_NODES]; c->counters.triTests = v[LRL_CTR_TRI]; c->counters.sphereTests = v[LRL_CTR_SPH];
        c->counters.otherTests = v[LRL_CTR_OTHER]; c->counters.primaryHits = v[LRL_CTR_PRIMARY_HITS];
        c->counters.recordFetches = v[LRL_CTR_FETCH]; c->counters.laneTrips = v[LRL_CTR_LANE_TRIPS]; c->counters.waveTrips = v[LRL_CTR_WAVE_TRIPS];
    }
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

int lrl_scene_rebuild_bvh(LrlContext* c) {
    if (!c) return LRL_ERR_INVALID_ARG;
    if (!c->haveScene) return fail(c, LRL_ERR_NO_SCENE, "no scene uploaded");
    if (!c->deviceBuilt) return fail(c, LRL_ERR_INVALID_ARG, "the tree came from the caller (LrlSceneView.bvh): it is never rebuilt here");
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc0 = orderAfterRenders(c)) return rc0;
    uint32_t recCount = 0, depth = 0;
    const int rc = runDeviceBuild(c, c->primTotal, recCount, depth);
    if (rc) return rc;
    c->recNum = recCount;
    c->bvhDepth = depth;
    return LRL_OK;
}

int lrl_water_init(LrlContext* c, uint32_t gx, uint32_t gy, float k, float dt, const LrlWaterPoint* points, uint32_t firstTriangle,
                   const LrlTriangle* triangles) {
    if (!c || !points || !triangles) return LRL_ERR_INVALID_ARG;
    if (!c->haveScene) return fail(c, LRL_ERR_NO_SCENE, "no scene uploaded");
    if (gx < 2 || gy < 2 || uint64_t(gx) * gy > 16384u) return fail(c, LRL_ERR_LIMIT, "water grid must be 2 x 2 .. 16384 points");
    const uint64_t count = 2ull * (gx - 1) * (gy - 1);
    if (uint64_t(firstTriangle) + count > c->num.triangleNum) return fail(c, LRL_ERR_INVALID_ARG, "water triangles out of range");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = upload(c, c->waterPoints, points, size_t(gx) * gy * sizeof(LrlWaterPoint));
    if (rc) return rc;
    if ((rc = upload(c, c->waterTris, triangles, size_t(count) * sizeof(LrlTriangle)))) return rc;
    std::vector<uint32_t> mat(count);
    const size_t before = c->materialTable.size();
    for (uint64_t i = 0; i < count; ++i) mat[i] = internMaterial(c, triangles[i].color);
    if (c->materialTable.size() != before && (rc = upload(c, c->materials, c->materialTable.data(), c->materialTable.size() * sizeof(LrlColor)))) return rc;
    if ((rc = upload(c, c->waterMat, mat.data(), mat.size() * 4))) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));      // the host arrays are borrowed only for this call
    c->waterGX = gx; c->waterGY = gy; c->waterFirst = firstTriangle; c->waterK = k; c->waterDt = dt;
    return LRL_OK;
}

int lrl_water_step(LrlContext* c, uint32_t substeps) {
    if (!c) return LRL_ERR_INVALID_ARG;
    if (!c->waterGX) return fail(c, LRL_ERR_NO_SCENE, "lrl_water_init first");
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc0 = orderAfterRenders(c)) return rc0;
    const uint32_t count = 2u * (c->waterGX - 1) * (c->waterGY - 1);
    if (c->ownBox.size() >= size_t(c->waterFirst + count) * 6) {      // where the water's triangles are now is known on the device only
        for (uint32_t i = 0; i < count; ++i) { float* q = &c->ownBox[size_t(c->waterFirst + i) * 6]; q[0] = 1.0f; q[3] = 0.0f; }
    }
    ownBoxesStale(c);
    HIP_TRY(c, lrl_launch_water_run(static_cast<float4*>(c->waterPoints.p), c->waterGX, c->waterGY, c->waterK, c->waterDt, substeps,
                                    static_cast<LrlTriangle*>(c->waterTris.p), c->stream));
    HIP_TRY(c, lrl_launch_triangle_pre(static_cast<const LrlTriangle*>(c->waterTris.p), static_cast<const uint32_t*>(c->waterMat.p), c->waterFirst, count,
                                       static_cast<float4*>(c->triHot.p), static_cast<LrlTriCold*>(c->triCold.p), static_cast<float4*>(c->triLeaf.p),
                                       c->leafOrder ? static_cast<const uint32_t*>(c->posOfTri.p) : nullptr, c->stream));
    if (c->deviceBuilt)      // keep the boxes current for lrl_scene_rebuild_bvh
        HIP_TRY(c, lrl_launch_triangle_boxes(static_cast<const LrlTriangle*>(c->waterTris.p), c->waterFirst, count, static_cast<float4*>(c->primMin.p),
                                             static_cast<float4*>(c->primMax.p), c->stream));
    return LRL_OK;
}

int lrl_water_read(LrlContext* c, LrlWaterPoint* points, LrlTriangle* triangles) {
    if (!c) return LRL_ERR_INVALID_ARG;
    if (!c->waterGX) return fail(c, LRL_ERR_NO_SCENE, "lrl_water_init first");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (points) HIP_TRY(c, hipMemcpy(points, c->waterPoints.p, size_t(c->waterGX) * c->waterGY * sizeof(LrlWaterPoint), hipMemcpyDeviceToHost));
    if (triangles) HIP_TRY(c, hipMemcpy(triangles, c->waterTris.p, size_t(2) * (c->waterGX - 1) * (c->waterGY - 1) * sizeof(LrlTriangle), hipMemcpyDeviceToHost));
    return LRL_OK;
}

int lrl_scene_info(const LrlContext* c, LrlSceneInfo* out) {
    if (!c || !out) return LRL_ERR_INVALID_ARG;
    if (!c->haveScene) return LRL_ERR_NO_SCENE;
    out->bvhNodes = c->nodeNum; out->bvhRecords = c->recNum; out->bvhDepth = c->bvhDepth; out->wideDepth = c->node4Depth;
    out->deviceBuilt = c->deviceBuilt ? 1u : 0u; out->materials = uint32_t(c->materialTable.size());
    out->ownBoxes = c->looseInUse ? 0u : c->parkedTight;
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
    out[0] = uint64_t(c->recNum) * 64 + uint64_t(c->wideNodes) * 64 * (c->haveLoose ? 2 : 1);      // binary records + the four-wide nodes in use (two copies where a caller's tree is fitted)
    out[1] = uint64_t(c->num.triangleNum) * 64 * (c->leafOrder ? 2 : 1);      // hot records by triangle number (+ the copy in leaf order)
    out[2] = uint64_t(c->num.triangleNum) * sizeof(LrlTriCold) + c->materialTable.size() * sizeof(LrlColor);
    out[3] = uint64_t(c->num.sphereNum + c->num.circleNum + c->num.cylinderNum + c->num.coneNum) * 128 + uint64_t(c->num.pointLightNum) * 32 +
             uint64_t(c->texW) * c->texH * c->texL * 4 + uint64_t(c->cubeSize) * c->cubeSize * 24;
    return LRL_OK;
}

}  // extern "C"
