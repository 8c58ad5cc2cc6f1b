// liblrltrace.so — the C ABI of include/lrl_trace.h: the scene.  Validation of a caller's tree, uploads, updates, the Pool's water
// simulation, and the host side of what lrl_build.hip does on the device: the library's own tree (build, layout), a caller's tree (records,
// fit to the triangles' own boxes, four-wide collapse, layout, refit after every move).  Host side only; the kernels live in
// kernels_parity.hip (TrianglePre, CirclePre, water) and lrl_build.hip.
#include "lrl_context.h"

// ---- lrl_scene_update_triangles_noalloc: its staging area ----
// lrl_scene_update_triangles stages the caller's origin records and their material indices in device buffers it allocates
// (triStage, matStage) for trianglePreKernel to read.  The no-allocation variant stages them here instead: one array per GPU,
// part of the code object, held from the library's load until the process ends.  Layout: LRL_NOALLOC_MAX_TRIANGLES origin
// records (160 B each), then as many material indices (4 B each).  Host code never uses the array's name as an address: lrl_create
// asks hipGetSymbolAddress for this device's copy (noallocScratchFor) and everything below works on that pointer.
static __device__ __align__(256) unsigned char scratch[size_t(LRL_NOALLOC_MAX_TRIANGLES) * (sizeof(LrlTriangle) + 4u)];

namespace lrl_host {

bool transmits(const LrlColor& col) { return !(col.t[0] == 0.0f && col.t[1] == 0.0f && col.t[2] == 0.0f); }   // NaN counts as transmitting
// the shade kernels read u, v of a hit for texture layers (>= 0) and the checker (-2) only
bool textured(const LrlColor& col) { return col.texR >= 0 || col.texT >= 0 || col.texD >= 0 || col.texG >= 0 || col.texD == -2 || col.texG == -2; }

uint32_t internMaterial(LrlContext* c, const LrlColor& col) {
    if (transmits(col)) c->hasTransmission = true;
    if (textured(col)) c->triNeedUV = true;
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
int validateBVH(LrlContext* c, const LrlSceneView* s, uint32_t& depthOut, uint32_t& recCountOut) {
    const uint32_t N = s->bvhNodeNum;
    if (N == 0 || N >= LRL_REF_LEAF) return fail(c, LRL_ERR_INVALID_ARG, "bvhNodeNum out of range");
    const uint32_t counts[7] = {0, 0, s->num.triangleNum, s->num.sphereNum, s->num.circleNum, s->num.cylinderNum, s->num.coneNum};
    uint32_t inner = 0;
    // (a node with two parents: the device passes give every node ONE place in the packed tree; a shared subtree would leave one of its
    // parents pointing at a slot nobody writes)
    std::vector<uint8_t> hasParent(N, 0);
    auto adopt = [&](uint32_t child) { const bool twice = hasParent[child] != 0; hasParent[child] = 1; return twice; };
    for (uint32_t i = 0; i < N; ++i) {
        const LrlBVHNode& n = s->bvh[i];
        inner += n.bound.leftChild != 0 ? 1u : 0u;
        if (n.bound.leftChild != 0 && i + 1 >= N) return fail(c, LRL_ERR_INVALID_ARG, "BVH: left child past the end");
        if (n.bound.rightChild != 0 && (n.bound.rightChild < 0 || uint32_t(n.bound.rightChild) >= N || uint32_t(n.bound.rightChild) <= i))
            return fail(c, LRL_ERR_INVALID_ARG, "BVH: right child index invalid (must be a later pre-order index)");
        if (n.bound.rightChild != 0 && n.bound.leftChild == 0) return fail(c, LRL_ERR_INVALID_ARG, "BVH: right child without left child");
        if (n.bound.rightChild != 0 && uint32_t(n.bound.rightChild) == i + 1) return fail(c, LRL_ERR_INVALID_ARG, "BVH: left and right child are the same node");
        if ((n.bound.leftChild != 0 && adopt(i + 1)) || (n.bound.rightChild != 0 && adopt(uint32_t(n.bound.rightChild))))
            return fail(c, LRL_ERR_INVALID_ARG, "BVH: a node has two parents");
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
    recCountOut = 1u + inner;      // record 0 is the root's virtual parent
    if (depth + 1 > LRL_MAX_BVH_DEPTH) return fail(c, LRL_ERR_BVH_DEPTH, "BVH deeper than LRL_MAX_BVH_DEPTH");
    return LRL_OK;
}

std::vector<uchar4> bgrToRgbx(const uint8_t* bgr, size_t texels) {
    std::vector<uchar4> o(texels);
    for (size_t i = 0; i < texels; ++i) o[i] = make_uchar4(bgr[i * 3 + 2], bgr[i * 3 + 1], bgr[i * 3 + 0], 255);
    return o;
}

// ---- the four-wide tree in the order the rays walk it (lrl_layout_node4, lrl_build.hip) ----
int ensureLeafOrder(LrlContext* c, uint32_t recCount, uint32_t triCount) {
    int rc;
    if ((rc = ensure(c, c->layoutScratch, lrl_layout_scratch_bytes(recCount)))) return rc;
    if ((rc = ensure(c, c->posOfTri, size_t(triCount) * 4))) return rc;
    return ensure(c, c->triLeaf, size_t(triCount) * 64);
}
void releaseLayoutTemporaries(LrlContext* c) {      // (hipFree waits for the device)
    if (c->nodes32Raw.bytes > (64u << 20)) release(c->nodes32Raw);
    if (c->src4Raw.bytes > (64u << 20)) release(c->src4Raw);
    if (c->layoutScratch.bytes > (64u << 20)) release(c->layoutScratch);
    if (c->callerScratch.bytes > (64u << 20)) release(c->callerScratch);
}
int gatherTriLeaf(LrlContext* c, uint32_t triCount) {
    if (!triCount) return LRL_OK;
    HIP_TRY(c, lrl_launch_tri_leaf_gather(static_cast<const float4*>(c->triHot.p), static_cast<const uint32_t*>(c->posOfTri.p), 0, triCount,
                                          static_cast<float4*>(c->triLeaf.p), c->stream));
    return LRL_OK;
}

// A caller's four-wide tree that is fitted to its triangles (uploadCallerTree) follows them wherever they go, on the device and in stream
// order (lrl_refit_node4): after every update of triangles and every water step.  The reference never refits: the Pool's leaf boxes were made
// tall enough for every wave (/root/reference/Pool/Pool.cpp:250-263), and those boxes stay the contract here too (a triangle that leaves the box
// the caller's tree gives it is bounded by that box, as for the reference).
int refitOnDevice(LrlContext* c) {
    if (!c->fitActive) return LRL_OK;
    HIP_TRY(c, lrl_refit_node4(static_cast<uint4*>(c->nodes32.p), c->wideNodes, static_cast<const uint4*>(c->src4.p), static_cast<const LrlBVHNode*>(c->bvhDev.p),
                               static_cast<const float4*>(c->primMin.p), static_cast<const float4*>(c->primMax.p), c->num.triangleNum, c->fitPad, c->node4Depth + 1u,
                               c->refitScratch.p, c->stream));
    c->refitted = true;
    return LRL_OK;
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
    if (c->triBoxes)      // keep the boxes current: for lrl_scene_rebuild_bvh, for the refit of a caller's tree
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
void otherBox(uint32_t kind, const void* records, uint32_t i, float mn[3], float mx[3]) {
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
                                    static_cast<uint32_t*>(c->posOfTri.p), triCount, true, nullptr, nullptr, c->stream));
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

// The caller's tree (pre-order 64-B own-box nodes, validated) -> the two device layouts: 64-B inner records that hold both children's boxes, and
// the four-wide compressed nodes, fitted to the triangles' own boxes where that pays, in the order the rays walk them.  All of it on the device
// (lrl_caller_tree, lrl_layout_node4); the triangles are there already.  Also lrl_scene_update_bvh.
int uploadCallerTree(LrlContext* c, const LrlBVHNode* bvh, uint32_t N, uint32_t recCount, uint32_t depth) {
    int rc;
    const uint32_t triCount = c->num.triangleNum;
    const bool layout = c->tuning.treeLayout == 0;
    c->fitActive = false;
    c->leafOrder = false;
    c->parkedTight = 0;
    if ((rc = upload(c, c->bvhDev, bvh, size_t(N) * sizeof(LrlBVHNode)))) return rc;
    DevBuf& raw = layout ? c->nodes32Raw : c->nodes32;
    DevBuf& srcRaw = layout ? c->src4Raw : c->src4;
    if ((rc = ensure(c, c->nodes, size_t(recCount) * 64))) return rc;
    if ((rc = ensure(c, raw, size_t(recCount) * 64))) return rc;
    if ((rc = ensure(c, srcRaw, size_t(recCount) * 16))) return rc;
    if ((rc = ensure(c, c->callerScratch, lrl_caller_tree_scratch_bytes(N, recCount)))) return rc;
    if ((rc = ensure(c, c->buildInfo, 256))) return rc;
    float scale = 0.0f;       // of the coordinates the slab arithmetic works with
    for (int a = 0; a < 3; ++a) scale = std::max(scale, std::max(std::fabs(bvh[0].bound.min[a]), std::fabs(bvh[0].bound.max[a])));
    c->fitPad = std::isfinite(scale) ? std::ldexp(scale, -14) : 0.0f;
    const bool own = c->triBoxes && triCount;
    const uint32_t* sizes[2] = {nullptr, nullptr};
    HIP_TRY(c, lrl_caller_tree(static_cast<const LrlBVHNode*>(c->bvhDev.p), N, recCount, depth + 1u, own ? static_cast<const float4*>(c->primMin.p) : nullptr,
                               own ? static_cast<const float4*>(c->primMax.p) : nullptr, triCount, c->fitPad, c->callerScratch.p, static_cast<float4*>(c->nodes.p),
                               static_cast<uint4*>(raw.p), static_cast<uint4*>(srcRaw.p), sizes, static_cast<uint32_t*>(c->buildInfo.p), c->stream));
    uint32_t info[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIP_TRY(c, hipMemcpyAsync(info, c->buildInfo.p, sizeof(info), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));      // (`bvh` is borrowed only for this call)
    if (info[6] == 0 || info[0] == 0 || info[3] == 0 || info[3] > recCount) return fail(c, LRL_ERR_DEVICE, "the caller's tree did not reach its root on the device");
    if (info[4] > triCount) return fail(c, LRL_ERR_INVALID_ARG, "BVH: a triangle appears in more than one node");
    c->node4Depth = info[0];
    c->wideNodes = recCount;
    if (layout) {
        if ((rc = ensure(c, c->nodes32, size_t(info[3]) * 64))) return rc;
        if ((rc = ensure(c, c->src4, size_t(info[3]) * 16))) return rc;
        if ((rc = ensureLeafOrder(c, recCount, triCount))) return rc;
        // (a caller's tree that leaves a triangle out leaves a place of triLeaf without a record: an all-zero record is hit by no ray)
        if (triCount) HIP_TRY(c, hipMemsetAsync(c->triLeaf.p, 0, size_t(triCount) * 64, c->stream));
        HIP_TRY(c, lrl_layout_node4(static_cast<const uint4*>(raw.p), recCount, sizes[0], sizes[1], info[0] + 1u, c->layoutScratch.p, static_cast<uint4*>(c->nodes32.p),
                                    static_cast<uint32_t*>(c->posOfTri.p), triCount, true, static_cast<const uint4*>(srcRaw.p), static_cast<uint4*>(c->src4.p), c->stream));
        if ((rc = gatherTriLeaf(c, triCount))) return rc;
        c->leafOrder = true;
        c->wideNodes = info[3];
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    releaseLayoutTemporaries(c);
    c->fitActive = info[7] != 0;
    c->parkedTight = info[5];
    c->refitted = false;
    if (c->fitActive) {
        if ((rc = ensure(c, c->refitScratch, lrl_refit_scratch_bytes(c->wideNodes)))) return rc;
    } else { release(c->bvhDev); release(c->src4); release(c->refitScratch); }      // (nothing follows the triangles)
    if (c->tuning.debug) std::fprintf(stderr, "lrl: the four-wide tree bounds %u primitives by their own boxes (%u would gain, nested %d); %u of %u node slots in use\n",
                                      info[5], info[1], info[2] ? 0 : 1, c->wideNodes, recCount);
    return LRL_OK;
}

}  // namespace lrl_host

using namespace lrl_host;

extern "C" {

int lrl_scene_upload(LrlContext* c, const LrlSceneView* s) {
    if (c) { c->expectedIters = 0; c->handoverThr.clear(); c->feedbackTick = 0; c->contIters = 0; c->phasedTrial.state = 0; c->phasedTrial.signature = 0; }      // the hand-over thresholds start again with a new scene (feedback still in flight is the old scene's: dropped)
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
    uint32_t recCount = 1;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {      // (LrlTuning.debug: where an upload's time goes)
        if (!c->tuning.debug) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "lrl: upload: %s %.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    };
    int rc = deviceBuild ? LRL_OK : validateBVH(c, s, depth, recCount);
    if (rc) return rc;
    lap("tree validated");
    c->haveScene = false;
    c->deviceBuilt = false;
    c->waterGX = c->waterGY = 0;      // a new scene: lrl_water_init again

    c->fitActive = false;
    c->parkedTight = 0;
    c->leafOrder = false;      // (the tree is made and laid out after the triangles are there; TrianglePre writes both copies of their records from then on)
    // the triangles' own boxes: what the device build sorts, what a caller's tree is fitted to (trianglesGPU does not give the corners)
    c->triBoxes = g.triangleNum && s->trianglesOrigin;
    if (deviceBuild || c->triBoxes) {
        const uint64_t boxes = deviceBuild ? primTotal : uint64_t(g.triangleNum);
        if ((rc = ensure(c, c->primMin, size_t(boxes) * 16))) return rc;
        if ((rc = ensure(c, c->primMax, size_t(boxes) * 16))) return rc;
    }
    c->materialTable.clear();
    c->materialIndex.clear();
    c->hasTransmission = false;
    c->triNeedUV = false;
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
                rc = runTrianglePre(c, first, cnt, s->trianglesOrigin + first);      // (and their boxes)
                if (rc) return rc;
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
                LrlTriCold& q = cold[i];
                std::memcpy(q.uv1, t.uv1, 8); std::memcpy(q.uv2, t.uv2, 8); std::memcpy(q.uv3, t.uv3, 8);
                q.material = internMaterial(c, t.color); q._pad = 0;
                float mat; std::memcpy(&mat, &q.material, 4);
                hot[size_t(i) * 4 + 3] = make_float4(t.k2[2], id, mat, 0);    // (the triangle's number and its material travel in its hot record)
            }
            rc = upload(c, c->triHot, hot.data(), hot.size() * sizeof(float4));
            if (rc) return rc;
            rc = upload(c, c->triCold, cold.data(), cold.size() * sizeof(LrlTriCold));
            if (rc) return rc;
            rc = upload(c, c->materials, c->materialTable.data(), c->materialTable.size() * sizeof(LrlColor));
            if (rc) return rc;
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
    }
    lap("triangles");
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
    lap("other primitives, textures");
    if (!deviceBuild) {
        if ((rc = uploadCallerTree(c, s->bvh, s->bvhNodeNum, recCount, depth))) return rc;
        if (c->tuning.debug) std::fprintf(stderr, "lrl: binary nodes %u (depth %u), four-wide tree height %u\n", s->bvhNodeNum, depth, c->node4Depth);
    } else {
        if ((rc = buildOnDevice(c, s, uint32_t(primTotal), recCount, depth))) return rc;
        if (c->tuning.debug) std::fprintf(stderr, "lrl: device-built tree over %llu primitives: %u records, height %u, four-wide height %u\n",
                                                   (unsigned long long)primTotal, recCount, depth, c->node4Depth);
    }
    lap("tree");
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
    if (int rc = runTrianglePre(c, first, count, tris)) return rc;
    return count ? refitOnDevice(c) : LRL_OK;
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
    unsigned char* const area = static_cast<unsigned char*>(c->noallocScratch);
    LrlTriangle* const stageTris = reinterpret_cast<LrlTriangle*>(area);
    uint32_t* const stageMat = reinterpret_cast<uint32_t*>(area + size_t(LRL_NOALLOC_MAX_TRIANGLES) * sizeof(LrlTriangle));
    // the area keeps the last call's records beyond `count`: nothing reads them, the kernels stop at `count`
    HIP_TRY(c, hipMemcpyAsync(stageTris, pinTris, size_t(count) * sizeof(LrlTriangle), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(stageMat, pinMat, size_t(count) * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, lrl_launch_triangle_pre(stageTris, stageMat, first, count, static_cast<float4*>(c->triHot.p), static_cast<LrlTriCold*>(c->triCold.p),
                                       static_cast<float4*>(c->triLeaf.p), c->leafOrder ? static_cast<const uint32_t*>(c->posOfTri.p) : nullptr, c->stream));
    if (c->triBoxes)      // keep the boxes current: for lrl_scene_rebuild_bvh, for the refit of a caller's tree
        HIP_TRY(c, lrl_launch_triangle_boxes(stageTris, first, count, static_cast<float4*>(c->primMin.p), static_cast<float4*>(c->primMax.p), c->stream));
    if (int rcFit = refitOnDevice(c)) return rcFit;      // (its scratch was allocated when the tree came)
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
    int rc = validateBVH(c, &v, depth, recCount);
    if (rc) return rc;
    if ((rc = orderAfterRenders(c))) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));      // frames in flight walk the old tree
    c->haveScene = false;                             // (a failed upload leaves no half-written tree in use)
    if ((rc = uploadCallerTree(c, bvh, bvhNodeNum, recCount, depth))) return rc;      // (the new tree has its own order of leaves: triLeaf is made anew)
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
    HIP_TRY(c, lrl_launch_water_run(static_cast<float4*>(c->waterPoints.p), c->waterGX, c->waterGY, c->waterK, c->waterDt, substeps,
                                    static_cast<LrlTriangle*>(c->waterTris.p), c->stream));
    HIP_TRY(c, lrl_launch_triangle_pre(static_cast<const LrlTriangle*>(c->waterTris.p), static_cast<const uint32_t*>(c->waterMat.p), c->waterFirst, count,
                                       static_cast<float4*>(c->triHot.p), static_cast<LrlTriCold*>(c->triCold.p), static_cast<float4*>(c->triLeaf.p),
                                       c->leafOrder ? static_cast<const uint32_t*>(c->posOfTri.p) : nullptr, c->stream));
    if (c->triBoxes)      // keep the boxes current: for lrl_scene_rebuild_bvh, for the refit of a caller's tree
        HIP_TRY(c, lrl_launch_triangle_boxes(static_cast<const LrlTriangle*>(c->waterTris.p), c->waterFirst, count, static_cast<float4*>(c->primMin.p),
                                             static_cast<float4*>(c->primMax.p), c->stream));
    return refitOnDevice(c);
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

}  // extern "C"
