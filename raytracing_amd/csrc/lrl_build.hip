// lrl_build.hip — BVH build on the device.
//
// The reference builds its tree on one host thread (missing source; BVH.exe@0x140005c90, SURVEY App. A: recursive mean
// split with per-level heap vectors), which is what dominates end to end once the mesh is large (C5: minutes).  When the
// caller passes no tree (LrlSceneView::bvh == NULL) the library builds one here, directly in the two device layouts the
// trace kernels read (lrl_device.h: `recs`, the binary child-box records, and `node4`, the four-wide compressed nodes),
// without ever making the reference's 64-B own-box nodes:
//   1. bounds of the primitive-box centres (block reduction + 6 atomics per block)
//   2. 63-bit Morton codes of the centres, radix sort of (code, primitive) pairs (rocPRIM)
//   3. the binary radix tree over the sorted codes (Karras 2012: every inner node finds its range and split by itself)
//   4. boxes bottom-up (the second thread to reach a node does the work), which also writes the node's 64-B record, its
//      height, and orders its children by the axis along which their centres differ most (the reference-order kernel
//      visits "left" first when the ray goes up that axis)
//   5. the four-wide node of EVERY record, each as if it were the root of a collapse (two expansions of the larger inner
//      item): they are addressed by record index, so the ones no ray reaches cost nothing but their memory
//   6. the height of the four-wide tree, bottom-up again (stack sizes of the trace kernels)
// From LRL_BUILD_LEVELS_FROM primitives on, steps 4-6 are one kernel launched once per level of readiness (fitLevelKernel).
// Step 3 has a second form (LrlTuning.deviceBuilder): PLOC, below.
// Not the reference's tree: images agree with a host-built tree except where two primitives are hit at exactly the same
// distance (tests/test_gpu_parity.py::test_device_built_bvh*).
// Also here, because they share the node writer and the level-by-level scheme: the four-wide tree in the order the rays walk it
// (lrl_layout_node4), a CALLER's tree from its 64-B nodes to the two device layouts with the fit to the triangles' own boxes
// (lrl_caller_tree), and the refit of such a tree after its triangles moved (lrl_refit_node4).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "lrl_device.h"

// primitives from which steps 4-6 run level by level (fitLevelKernel) instead of as a climb with atomics.  Measured, ms per rebuild,
// climb / levels: 4446 primitives 0.23 / 0.45, 77 560: 1.22 / 0.98, 310 k: 4.45 / 1.43, 1.24 M: 17.5 / 2.8, 11.2 M: 164.9 / 16.4
#define LRL_BUILD_LEVELS_FROM LRL_LAYOUT_FROM

namespace lrl_build {

#define DI static __device__ __forceinline__

DI uint32_t fkey(float f) { const uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }   // order-preserving
DI float fkeyInv(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
DI float ldf(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }      // written by another wave of this launch
DI uint32_t ldu(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DI float4 ld4(const float4* p) {
    const float* f = reinterpret_cast<const float*>(p);
    return make_float4(ldf(f), ldf(f + 1), ldf(f + 2), ldf(f + 3));
}

// boxes of the triangles of one staged slab: exact min / max of the three corners, like the reference's leaves
// (BVH.exe@0x140004d00); w of the min corner carries the primitive bits
__global__ __launch_bounds__(256) void triBoxKernel(const LrlTriangle* __restrict__ in, uint32_t first, uint32_t count, float4* __restrict__ pmin,
                                                    float4* __restrict__ pmax) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const float4* q = reinterpret_cast<const float4*>(in + i);
    const float4 a = q[0], b = q[1], c = q[2];
    pmin[first + i] = make_float4(fminf(a.x, fminf(b.x, c.x)), fminf(a.y, fminf(b.y, c.y)), fminf(a.z, fminf(b.z, c.z)),
                                  __uint_as_float((2u << LRL_PRIM_TYPE_SHIFT) | (first + i)));
    pmax[first + i] = make_float4(fmaxf(a.x, fmaxf(b.x, c.x)), fmaxf(a.y, fmaxf(b.y, c.y)), fmaxf(a.z, fmaxf(b.z, c.z)), 0.0f);
}

__global__ __launch_bounds__(256) void boundsKernel(const float4* __restrict__ pmin, const float4* __restrict__ pmax, uint32_t n, uint32_t* bounds) {
    __shared__ uint32_t sh[6];
    if (threadIdx.x < 3) sh[threadIdx.x] = 0xffffffffu;
    else if (threadIdx.x < 6) sh[threadIdx.x] = 0u;
    __syncthreads();
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const float4 a = pmin[i], b = pmax[i];
        const float cx = 0.5f * (a.x + b.x), cy = 0.5f * (a.y + b.y), cz = 0.5f * (a.z + b.z);
        atomicMin(&sh[0], fkey(cx)); atomicMin(&sh[1], fkey(cy)); atomicMin(&sh[2], fkey(cz));
        atomicMax(&sh[3], fkey(cx)); atomicMax(&sh[4], fkey(cy)); atomicMax(&sh[5], fkey(cz));
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&bounds[threadIdx.x], sh[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&bounds[threadIdx.x], sh[threadIdx.x]);
}

DI uint64_t spread21(uint32_t v) {      // 21 bits -> every third bit
    uint64_t x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ __launch_bounds__(256) void mortonKernel(const float4* __restrict__ pmin, const float4* __restrict__ pmax, uint32_t n,
                                                    const uint32_t* __restrict__ bounds, uint64_t* __restrict__ keys, uint32_t* __restrict__ idx, uint32_t* info) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    {   // a box with a NaN or an infinity: the build goes on (it ends whatever the boxes), the host refuses the scene (info[2])
        const float4 a = pmin[i], b = pmax[i];
        const float s = (a.x + a.y + a.z) + (b.x + b.y + b.z);
        if (!(fabsf(s) < INFINITY)) info[2] = 1u;
    }
    const float lo[3] = {fkeyInv(bounds[0]), fkeyInv(bounds[1]), fkeyInv(bounds[2])};
    const float hi[3] = {fkeyInv(bounds[3]), fkeyInv(bounds[4]), fkeyInv(bounds[5])};
    const float4 a = pmin[i], b = pmax[i];
    const float c[3] = {0.5f * (a.x + b.x), 0.5f * (a.y + b.y), 0.5f * (a.z + b.z)};
    // one cell size for the three axes (the longest extent): with per-axis scaling a flat scene would spend every third
    // bit on slicing its thin direction into pancakes
    const float ext = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        const float t = ext > 0.0f ? (c[k] - lo[k]) / ext : 0.0f;
        q[k] = uint32_t(fminf(fmaxf(t * 2097152.0f, 0.0f), 2097151.0f));
    }
    keys[i] = spread21(q[0]) << 2 | spread21(q[1]) << 1 | spread21(q[2]);
    idx[i] = i;
}

// common leading bits of the codes at sorted positions i and j; equal codes are told apart by position
DI int delta(const uint64_t* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t x = keys[i] ^ keys[j];
    return x ? __clzll((long long)x) : 64 + __clz(i ^ j);
}

// Karras 2012.  children[i] = {left, right}: an inner node index, or LRL_REF_LEAF | sorted position of a leaf.
__global__ __launch_bounds__(256) void treeKernel(const uint64_t* __restrict__ keys, uint32_t n, uint2* __restrict__ children,
                                                  uint32_t* __restrict__ parentInner, uint32_t* __restrict__ parentLeaf) {
    const int i = int(blockIdx.x * 256u + threadIdx.x);
    const int N = int(n);
    if (i >= N - 1) return;
    const int d = delta(keys, N, i, i + 1) - delta(keys, N, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, N, i, i - d);
    int lmax = 2;
    while (delta(keys, N, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, N, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, N, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (delta(keys, N, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    uint2 c;
    if (lo == gamma) { c.x = LRL_REF_LEAF | uint32_t(gamma); parentLeaf[gamma] = uint32_t(i); }
    else { c.x = uint32_t(gamma); parentInner[gamma] = uint32_t(i); }
    if (hi == gamma + 1) { c.y = LRL_REF_LEAF | uint32_t(gamma + 1); parentLeaf[gamma + 1] = uint32_t(i); }
    else { c.y = uint32_t(gamma + 1); parentInner[gamma + 1] = uint32_t(i); }
    children[i] = c;
    if (i == 0) parentInner[0] = 0xffffffffu;
}

// bottom-up: boxes, heights and the 64-B records.  Inner node i is record i + 1; record 0 is the virtual parent of the root.
__global__ __launch_bounds__(256) void fitKernel(const float4* __restrict__ pmin, const float4* __restrict__ pmax, const uint32_t* __restrict__ idx, uint32_t n,
                                                 const uint2* __restrict__ children, const uint32_t* __restrict__ parentInner,
                                                 const uint32_t* __restrict__ parentLeaf, uint32_t* counters, float4* nodeMin, float4* nodeMax,
                                                 uint32_t* heights, float4* recs, uint32_t* info) {
    const uint32_t leaf = blockIdx.x * 256u + threadIdx.x;
    if (leaf >= n) return;
    uint32_t cur = parentLeaf[leaf];
    for (;;) {
        __threadfence();
        if (atomicAdd(&counters[cur], 1u) == 0u) return;      // the sibling subtree is not finished: its last thread goes on
        __threadfence();
        const uint2 c = children[cur];
        float4 mn[2], mx[2];
        uint32_t ref[2], h[2];
        const uint32_t cc[2] = {c.x, c.y};
        for (int k = 0; k < 2; ++k) {
            if (cc[k] & LRL_REF_LEAF) {
                const uint32_t p = idx[cc[k] & ~LRL_REF_LEAF];
                mn[k] = pmin[p]; mx[k] = pmax[p];
                ref[k] = LRL_REF_LEAF | __float_as_uint(mn[k].w);
                h[k] = 0;
            } else {
                mn[k] = ld4(nodeMin + cc[k]); mx[k] = ld4(nodeMax + cc[k]);
                ref[k] = cc[k] + 1u;
                h[k] = ldu(heights + cc[k]);
            }
        }
        // split axis: where the children's centres differ most; "left" = the lower one
        const float dx = (mn[1].x + mx[1].x) - (mn[0].x + mx[0].x), dy = (mn[1].y + mx[1].y) - (mn[0].y + mx[0].y),
                    dz = (mn[1].z + mx[1].z) - (mn[0].z + mx[0].z);
        uint32_t axis = fabsf(dx) >= fabsf(dy) ? (fabsf(dx) >= fabsf(dz) ? 0u : 2u) : (fabsf(dy) >= fabsf(dz) ? 1u : 2u);
        const float dd = axis == 0 ? dx : (axis == 1 ? dy : dz);
        const int L = dd < 0.0f ? 1 : 0, R = 1 - L;
        const float4 bmin = make_float4(fminf(mn[0].x, mn[1].x), fminf(mn[0].y, mn[1].y), fminf(mn[0].z, mn[1].z), 0.0f);
        const float4 bmax = make_float4(fmaxf(mx[0].x, mx[1].x), fmaxf(mx[0].y, mx[1].y), fmaxf(mx[0].z, mx[1].z), 0.0f);
        const uint32_t height = 1u + max(h[0], h[1]);
        float4* q = recs + size_t(cur + 1u) * 4;
        q[0] = make_float4(mn[L].x, mn[L].y, mn[L].z, __uint_as_float(ref[L]));
        q[1] = make_float4(mx[L].x, mx[L].y, mx[L].z, __uint_as_float(ref[R]));
        q[2] = make_float4(mn[R].x, mn[R].y, mn[R].z, __uint_as_float(0u));
        q[3] = make_float4(mx[R].x, mx[R].y, mx[R].z, __uint_as_float(axis));
        nodeMin[cur] = bmin; nodeMax[cur] = bmax;
        heights[cur] = height;
        if (cur == 0u) {
            recs[0] = make_float4(bmin.x, bmin.y, bmin.z, __uint_as_float(1u));
            recs[1] = make_float4(bmax.x, bmax.y, bmax.z, __uint_as_float(LRL_REF_NONE));
            recs[2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            recs[3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            info[0] = height;
            return;
        }
        cur = parentInner[cur];
    }
}

struct Item4 { float3 lo, hi; uint32_t ref; };
DI float area(const Item4& t) {
    const float dx = t.hi.x - t.lo.x, dy = t.hi.y - t.lo.y, dz = t.hi.z - t.lo.z;
    return dx * dy + dy * dz + dz * dx;
}
DI bool inner(uint32_t ref) { return ref != LRL_REF_NONE && !(ref & LRL_REF_LEAF); }
DI bool triLeaf(uint32_t ref) { return (ref >> LRL_PRIM_TYPE_SHIFT) == ((LRL_REF_LEAF >> LRL_PRIM_TYPE_SHIFT) | 2u); }

// the items of the four-wide node of record r: its children, the larger inner one replaced by ITS children, twice
DI int expand4(const float4* recs, Item4* it, int n) {
    while (n < 4) {
        int best = -1;
        float bestArea = -1.0f;
        for (int i = 0; i < n; ++i)
            if (inner(it[i].ref) && area(it[i]) > bestArea) { bestArea = area(it[i]); best = i; }
        if (best < 0) break;
        const float4* q = recs + size_t(it[best].ref) * 4;
        const float4 a = q[0], b = q[1], c = q[2], d = q[3];
        for (int i = best; i + 1 < n; ++i) it[i] = it[i + 1];
        --n;
        it[n++] = {make_float3(a.x, a.y, a.z), make_float3(b.x, b.y, b.z), __float_as_uint(a.w)};
        it[n++] = {make_float3(c.x, c.y, c.z), make_float3(d.x, d.y, d.z), __float_as_uint(b.w)};
    }
    return n;
}
DI int gather4(const float4* __restrict__ recs, uint32_t r, Item4* it) {
    int n = 0;
    const float4* q = recs + size_t(r) * 4;
    const float4 a = q[0], b = q[1], c = q[2], d = q[3];
    it[n++] = {make_float3(a.x, a.y, a.z), make_float3(b.x, b.y, b.z), __float_as_uint(a.w)};
    if (__float_as_uint(b.w) != LRL_REF_NONE) it[n++] = {make_float3(c.x, c.y, c.z), make_float3(d.x, d.y, d.z), __float_as_uint(b.w)};
    return expand4(recs, it, n);
}

// the four-wide node of record r from its items (lrl_device.h: 64 B, byte boxes rounded outwards)
DI void writeNode4(const Item4* it, int n, uint32_t r, uint4* node4) {
    float origin[3];
    uint32_t expo[3];
    double unit[3];
    for (int a = 0; a < 3; ++a) {
        float lo = INFINITY, hi = -INFINITY;
        for (int i = 0; i < n; ++i) {
            lo = fminf(lo, a == 0 ? it[i].lo.x : (a == 1 ? it[i].lo.y : it[i].lo.z));
            hi = fmaxf(hi, a == 0 ? it[i].hi.x : (a == 1 ? it[i].hi.y : it[i].hi.z));
        }
        origin[a] = lo;
        const double ext = double(hi) - double(lo);
        int e = -126;
        if (ext > 0.0) { (void)frexp(ext / 255.0, &e); }      // ext / 255 = m * 2^e with m in [0.5, 1): 2^e > ext / 255
        e = min(127, max(-126, e));
        while (e < 127 && ceil(ext / ldexp(1.0, e)) > 255.0) ++e;
        expo[a] = uint32_t(e + 127);
        unit[a] = ldexp(1.0, e);
    }
    uint32_t bytes[6] = {0, 0, 0, 0, 0, 0};
    uint32_t refs[4] = {LRL_REF_EMPTY4, LRL_REF_EMPTY4, LRL_REF_EMPTY4, LRL_REF_EMPTY4};
    for (int i = n; i < 4; ++i)
        for (int a = 0; a < 3; ++a) bytes[a] |= LRL_EMPTY4_LO << (8 * i);      // empty slots: inverted boxes
    for (int i = 0; i < n; ++i) {
        const float lo[3] = {it[i].lo.x, it[i].lo.y, it[i].lo.z}, hi[3] = {it[i].hi.x, it[i].hi.y, it[i].hi.z};
        for (int a = 0; a < 3; ++a) {
            const double ql = floor((double(lo[a]) - double(origin[a])) / unit[a]);
            const double qh = ceil((double(hi[a]) - double(origin[a])) / unit[a]);
            bytes[a] |= uint32_t(fmin(255.0, fmax(0.0, ql))) << (8 * i);
            bytes[3 + a] |= uint32_t(fmin(255.0, fmax(0.0, qh))) << (8 * i);
        }
        refs[i] = it[i].ref;
    }
    uint4* q = node4 + size_t(r) * 4;
    q[0] = make_uint4(__float_as_uint(origin[0]), __float_as_uint(origin[1]), __float_as_uint(origin[2]), expo[0] << 23);
    q[1] = make_uint4(bytes[0], bytes[1], bytes[2], bytes[3]);
    q[2] = make_uint4(bytes[4], bytes[5], refs[0], refs[1]);
    q[3] = make_uint4(refs[2], refs[3], expo[1] << 23, expo[2] << 23);
}
__global__ __launch_bounds__(256) void node4Kernel(const float4* __restrict__ recs, uint32_t recCount, uint4* __restrict__ node4) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= recCount) return;
    Item4 it[5];
    const int n = gather4(recs, r, it);
    writeNode4(it, n, r, node4);
}

DI uint32_t height4Of(const uint4* __restrict__ node4, const uint32_t* h4, uint32_t r) {
    const uint4 a = node4[size_t(r) * 4 + 2], b = node4[size_t(r) * 4 + 3];
    const uint32_t refs[4] = {a.z, a.w, b.x, b.y};
    uint32_t h = 0;
    for (int i = 0; i < 4; ++i)
        if (inner(refs[i])) h = max(h, ldu(h4 + refs[i]));
    return h + 1u;
}

// height of the four-wide tree, bottom-up over the binary tree: when both subtrees of a node are done, so are the
// four-wide nodes of everything below it
__global__ __launch_bounds__(256) void height4Kernel(uint32_t n, const uint32_t* __restrict__ parentInner, const uint32_t* __restrict__ parentLeaf,
                                                     uint32_t* counters, const uint4* __restrict__ node4, uint32_t* h4, uint32_t* info) {
    const uint32_t leaf = blockIdx.x * 256u + threadIdx.x;
    if (leaf >= n) return;
    uint32_t cur = parentLeaf[leaf];
    for (;;) {
        __threadfence();
        if (atomicAdd(&counters[cur], 1u) == 0u) return;
        __threadfence();
        h4[cur + 1u] = height4Of(node4, h4, cur + 1u);
        if (cur == 0u) {
            __threadfence();
            info[1] = height4Of(node4, h4, 0u);
            return;
        }
        cur = parentInner[cur];
    }
}

// Steps 4-6 for large trees, one tree level of readiness per launch: a node is fitted by the first launch that finds both children
// done by an EARLIER launch (doneAt < launch: what this launch writes is not read by it, and a kernel boundary makes the rest visible),
// and writes its record, its four-wide node and that node's height in one go.  No fences, no atomics: the climb above pays a
// device-scope release / acquire pair per node and child, which on eight L2s is a write-back each (100 M triangles: 873 + 664 ms for
// fitKernel + height4Kernel; this kernel: a few ms a launch for the first launches, then the cost of reading doneAt).
__global__ __launch_bounds__(256) void fitLevelKernel(const float4* __restrict__ pmin, const float4* __restrict__ pmax, const uint32_t* __restrict__ idx, uint32_t n,
                                                      const uint2* __restrict__ children, uint32_t* doneAt, uint32_t launch, float4* nodeMin, float4* nodeMax,
                                                      uint32_t* heights, float4* recs, uint4* node4, uint32_t* h4, uint32_t* sizeN, uint32_t* sizeT, uint32_t* info) {
    // (a grid-stride loop: late launches find nearly every node done, and dispatching 390 000 workgroups for 100 M nodes costs more than reading their stamps)
    for (uint32_t cur = blockIdx.x * 256u + threadIdx.x; cur + 1u < n; cur += gridDim.x * 256u) {
    if (doneAt[cur] != 0u) continue;
    const uint2 c = children[cur];
    const uint32_t cc[2] = {c.x, c.y};
    bool ready = true;
    for (int k = 0; k < 2; ++k)
        if (!(cc[k] & LRL_REF_LEAF)) {
            const uint32_t d = doneAt[cc[k]];
            ready = ready && d != 0u && d < launch;
        }
    if (!ready) continue;
    float4 mn[2], mx[2];
    uint32_t ref[2], h[2];
    for (int k = 0; k < 2; ++k) {
        if (cc[k] & LRL_REF_LEAF) {
            const uint32_t p = idx[cc[k] & ~LRL_REF_LEAF];
            mn[k] = pmin[p]; mx[k] = pmax[p];
            ref[k] = LRL_REF_LEAF | __float_as_uint(mn[k].w);
            h[k] = 0;
        } else {
            mn[k] = nodeMin[cc[k]]; mx[k] = nodeMax[cc[k]];
            ref[k] = cc[k] + 1u;
            h[k] = heights[cc[k]];
        }
    }
    // (as fitKernel) split axis: where the children's centres differ most; "left" = the lower one
    const float dx = (mn[1].x + mx[1].x) - (mn[0].x + mx[0].x), dy = (mn[1].y + mx[1].y) - (mn[0].y + mx[0].y),
                dz = (mn[1].z + mx[1].z) - (mn[0].z + mx[0].z);
    uint32_t axis = fabsf(dx) >= fabsf(dy) ? (fabsf(dx) >= fabsf(dz) ? 0u : 2u) : (fabsf(dy) >= fabsf(dz) ? 1u : 2u);
    const float dd = axis == 0 ? dx : (axis == 1 ? dy : dz);
    const int L = dd < 0.0f ? 1 : 0, R = 1 - L;
    const float4 bmin = make_float4(fminf(mn[0].x, mn[1].x), fminf(mn[0].y, mn[1].y), fminf(mn[0].z, mn[1].z), 0.0f);
    const float4 bmax = make_float4(fmaxf(mx[0].x, mx[1].x), fmaxf(mx[0].y, mx[1].y), fmaxf(mx[0].z, mx[1].z), 0.0f);
    const uint32_t height = 1u + max(h[0], h[1]);
    float4* q = recs + size_t(cur + 1u) * 4;
    q[0] = make_float4(mn[L].x, mn[L].y, mn[L].z, __uint_as_float(ref[L]));
    q[1] = make_float4(mx[L].x, mx[L].y, mx[L].z, __uint_as_float(ref[R]));
    q[2] = make_float4(mn[R].x, mn[R].y, mn[R].z, __uint_as_float(0u));
    q[3] = make_float4(mx[R].x, mx[R].y, mx[R].z, __uint_as_float(axis));
    nodeMin[cur] = bmin; nodeMax[cur] = bmax;
    heights[cur] = height;
    // the four-wide node of this record: its two children, then the expansions, which read records of earlier launches only
    Item4 it[5];
    it[0] = {make_float3(mn[L].x, mn[L].y, mn[L].z), make_float3(mx[L].x, mx[L].y, mx[L].z), ref[L]};
    it[1] = {make_float3(mn[R].x, mn[R].y, mn[R].z), make_float3(mx[R].x, mx[R].y, mx[R].z), ref[R]};
    const int cnt = expand4(recs, it, 2);
    writeNode4(it, cnt, cur + 1u, node4);
    uint32_t hw = 0, sN = 1u, sT = 0u;      // height, four-wide nodes and triangle leaves of the four-wide subtree (for the layout pass below)
    for (int i = 0; i < cnt; ++i)
        if (inner(it[i].ref)) { hw = max(hw, h4[it[i].ref]); sN += sizeN[it[i].ref]; sT += sizeT[it[i].ref]; }
        else if (triLeaf(it[i].ref)) ++sT;
    h4[cur + 1u] = hw + 1u;
    sizeN[cur + 1u] = sN; sizeT[cur + 1u] = sT;
    if (cur == 0u) {      // record 0, the virtual parent of the root: the same four items (the root replaced by its children, then as above)
        sizeN[0] = sN; sizeT[0] = sT;
        info[3] = sN; info[4] = sT;
        recs[0] = make_float4(bmin.x, bmin.y, bmin.z, __uint_as_float(1u));
        recs[1] = make_float4(bmax.x, bmax.y, bmax.z, __uint_as_float(LRL_REF_NONE));
        recs[2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        recs[3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        writeNode4(it, cnt, 0u, node4);
        info[1] = hw + 1u;
        info[0] = height;
    }
    doneAt[cur] = launch;
    }
}

// ---- step 3, the other way: clusters merged bottom-up with their nearest neighbour in Morton order (PLOC, Meister & Bittner 2018) ----
// Every cluster looks LRL_PLOC_RADIUS places to either side for the neighbour with which it would make the smallest box; pairs that chose
// each other merge into a new inner node, the rest wait; the survivors are compacted and the round repeats until one cluster is left.
// Inner nodes are numbered from the root down (root = 0, as the fit kernels expect): the t-th merge of the build is node n - 2 - t.
#define LRL_PLOC_RADIUS 8       /* (16, 32 and 64 give the same frame times within 1 %, and longer builds) */
DI float unionArea(const float4& amin, const float4& amax, const float4& bmin, const float4& bmax) {
    const float dx = fmaxf(amax.x, bmax.x) - fminf(amin.x, bmin.x), dy = fmaxf(amax.y, bmax.y) - fminf(amin.y, bmin.y),
                dz = fmaxf(amax.z, bmax.z) - fminf(amin.z, bmin.z);
    return dx * dy + dy * dz + dz * dx;
}
__global__ __launch_bounds__(256) void plocInitKernel(const float4* __restrict__ pmin, const float4* __restrict__ pmax, const uint32_t* __restrict__ idx, uint32_t n,
                                                      uint32_t* __restrict__ ref, float4* __restrict__ cmin, float4* __restrict__ cmax) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = idx[i];
    ref[i] = LRL_REF_LEAF | i;
    cmin[i] = pmin[p]; cmax[i] = pmax[p];
}
__global__ __launch_bounds__(256) void plocNearestKernel(const float4* __restrict__ cmin, const float4* __restrict__ cmax, uint32_t cnt, uint32_t* __restrict__ nn) {
    constexpr int RAD = LRL_PLOC_RADIUS;
    __shared__ float4 smin[256 + 2 * RAD], smax[256 + 2 * RAD];
    const int base = int(blockIdx.x * 256u) - RAD;
    for (int t = int(threadIdx.x); t < 256 + 2 * RAD; t += 256) {
        const int g = base + t;
        if (g >= 0 && g < int(cnt)) { smin[t] = cmin[g]; smax[t] = cmax[g]; }
    }
    __syncthreads();
    const int i = int(blockIdx.x * 256u + threadIdx.x);
    if (i >= int(cnt)) return;
    const int li = i - base;
    const float4 mn = smin[li], mx = smax[li];
    float best = INFINITY;
    int bestJ = (i ^ 1) < int(cnt) ? (i ^ 1) : i - 1;      // (boxes with a NaN compare with nothing: they pair off with their buddies in the end)
    // nearest places first, so that among equal boxes the nearer neighbour wins; the buddy i ^ 1 wins every tie (a run of identical boxes
    // pairs off in one round instead of forming a chain that merges one pair a round)
    for (int d = 1; d <= RAD; ++d)
        for (int sgn = 0; sgn < 2; ++sgn) {
            const int j = sgn ? i - d : i + d;
            if (j < 0 || j >= int(cnt)) continue;
            const float a = unionArea(mn, mx, smin[j - base], smax[j - base]);
            if (a < best || (a == best && j == (i ^ 1))) { best = a; bestJ = j; }
        }
    nn[i] = uint32_t(bestJ);
}
__global__ __launch_bounds__(256) void plocMarkKernel(const uint32_t* __restrict__ nn, uint32_t cnt, uint32_t* __restrict__ merge, uint32_t* __restrict__ keep) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= cnt) return;
    const uint32_t j = nn[i];
    const bool mutual = j != i && nn[j] == i;
    merge[i] = (mutual && i < j) ? 1u : 0u;
    keep[i] = (mutual && i > j) ? 0u : 1u;
}
__global__ __launch_bounds__(256) void plocApplyKernel(const uint32_t* __restrict__ refIn, const float4* __restrict__ cminIn, const float4* __restrict__ cmaxIn,
                                                       const uint32_t* __restrict__ nn, const uint32_t* __restrict__ merge, const uint32_t* __restrict__ keep,
                                                       const uint32_t* __restrict__ mOff, const uint32_t* __restrict__ kOff, uint32_t cnt, uint32_t nodeBase,
                                                       uint32_t n, uint32_t* __restrict__ refOut, float4* __restrict__ cminOut, float4* __restrict__ cmaxOut,
                                                       uint2* __restrict__ children, uint32_t* __restrict__ totals) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= cnt) return;
    if (i == cnt - 1u) { totals[0] = kOff[i] + keep[i]; totals[1] = mOff[i] + merge[i]; }
    if (!keep[i]) return;
    const uint32_t o = kOff[i];
    float4 mn = cminIn[i], mx = cmaxIn[i];
    uint32_t r = refIn[i];
    if (merge[i]) {
        const uint32_t j = nn[i];
        const float4 bn = cminIn[j], bx = cmaxIn[j];
        const uint32_t id = n - 2u - (nodeBase + mOff[i]);
        children[id] = make_uint2(r, refIn[j]);
        r = id;
        mn = make_float4(fminf(mn.x, bn.x), fminf(mn.y, bn.y), fminf(mn.z, bn.z), 0.0f);
        mx = make_float4(fmaxf(mx.x, bx.x), fmaxf(mx.y, bx.y), fmaxf(mx.z, bx.z), 0.0f);
    }
    refOut[o] = r; cminOut[o] = mn; cmaxOut[o] = mx;
}

// The same for small scenes (a rebuild per frame of the Pool's 4446 triangles): every round in ONE workgroup, so that a build is one launch
// with no host round trips.  Thread t owns a contiguous run of the clusters; the runs' merge / keep counts are scanned in LDS.
#define LRL_PLOC_SMALL_THREADS 1024
__global__ __launch_bounds__(LRL_PLOC_SMALL_THREADS) void plocSmallKernel(const float4* __restrict__ pmin, const float4* __restrict__ pmax, const uint32_t* __restrict__ idx,
                                                                          uint32_t n, uint32_t* refA, float4* minA, float4* maxA, uint32_t* refB, float4* minB,
                                                                          float4* maxB, uint32_t* nn, uint2* children, uint32_t* parentInner, uint32_t* parentLeaf) {
    __shared__ uint32_t sM[LRL_PLOC_SMALL_THREADS / 64], sK[LRL_PLOC_SMALL_THREADS / 64];
    const uint32_t T = LRL_PLOC_SMALL_THREADS, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t i = tid; i < n; i += T) {
        const uint32_t p = idx[i];
        refA[i] = LRL_REF_LEAF | i; minA[i] = pmin[p]; maxA[i] = pmax[p];
    }
    if (tid == 0) parentInner[0] = 0xffffffffu;
    __syncthreads();
    uint32_t cnt = n, nodeBase = 0;
    // A round merges at least one pair as long as unionArea orders the candidate pairs (the smallest one chose each other).  Should a round
    // ever merge nothing (boxes that compare with nothing: non-finite coordinates), the next one pairs the clusters off by position
    // (i with i ^ 1), which always merges: the loop ends after at most log2(n) such rounds instead of spinning in a one-workgroup kernel.
    bool byPosition = false;
    while (cnt > 1u) {
        const uint32_t chunk = (cnt + T - 1u) / T, lo = min(cnt, tid * chunk), hi = min(cnt, lo + chunk);      // chunk <= 64
        for (uint32_t i = lo; i < hi; ++i) {
            const float4 mn = minA[i], mx = maxA[i];
            float best = INFINITY;
            uint32_t bestJ = (i ^ 1u) < cnt ? (i ^ 1u) : i - 1u;
            if (byPosition) { nn[i] = (i ^ 1u) < cnt ? (i ^ 1u) : i; continue; }
            for (int d = 1; d <= LRL_PLOC_RADIUS; ++d)
                for (int sgn = 0; sgn < 2; ++sgn) {
                    const long long j = sgn ? (long long)i - d : (long long)i + d;
                    if (j < 0 || j >= (long long)cnt) continue;
                    const float a = unionArea(mn, mx, minA[j], maxA[j]);
                    if (a < best || (a == best && uint32_t(j) == (i ^ 1u))) { best = a; bestJ = uint32_t(j); }
                }
            nn[i] = bestJ;
        }
        __syncthreads();
        // which of my clusters merge (the lower of a mutual pair) and which go (the upper): one bit each
        unsigned long long mergeBits = 0ull, goneBits = 0ull;
        for (uint32_t i = lo; i < hi; ++i) {
            const uint32_t j = nn[i];
            const bool mutual = j != i && nn[j] == i;
            if (mutual && i < j) mergeBits |= 1ull << (i - lo);
            if (mutual && i > j) goneBits |= 1ull << (i - lo);
        }
        const uint32_t m = uint32_t(__popcll(mergeBits)), k = (hi - lo) - uint32_t(__popcll(goneBits));
        // exclusive scans over the threads: within the wave by shuffles, across the sixteen waves through LDS
        uint32_t im = m, ik = k;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t am = uint32_t(__shfl_up(int(im), off, 64)), ak = uint32_t(__shfl_up(int(ik), off, 64));
            if (lane >= uint32_t(off)) { im += am; ik += ak; }
        }
        if (lane == 63u) { sM[wave] = im; sK[wave] = ik; }
        __syncthreads();
        uint32_t mo = im - m, ko = ik - k, merges = 0, keeps = 0;
        for (uint32_t w = 0; w < T / 64u; ++w) {
            const uint32_t wm = sM[w], wk = sK[w];
            if (w < wave) { mo += wm; ko += wk; }
            merges += wm; keeps += wk;
        }
        for (uint32_t i = lo; i < hi; ++i) {
            if ((goneBits >> (i - lo)) & 1ull) continue;
            float4 mn = minA[i], mx = maxA[i];
            uint32_t r = refA[i];
            if ((mergeBits >> (i - lo)) & 1ull) {
                const uint32_t j = nn[i];
                const float4 bn = minA[j], bx = maxA[j];
                const uint32_t id = n - 2u - (nodeBase + mo++), rj = refA[j];
                children[id] = make_uint2(r, rj);
                if (r & LRL_REF_LEAF) parentLeaf[r & ~LRL_REF_LEAF] = id; else parentInner[r] = id;
                if (rj & LRL_REF_LEAF) parentLeaf[rj & ~LRL_REF_LEAF] = id; else parentInner[rj] = id;
                r = id;
                mn = make_float4(fminf(mn.x, bn.x), fminf(mn.y, bn.y), fminf(mn.z, bn.z), 0.0f);
                mx = make_float4(fmaxf(mx.x, bx.x), fmaxf(mx.y, bx.y), fmaxf(mx.z, bx.z), 0.0f);
            }
            refB[ko] = r; minB[ko] = mn; maxB[ko] = mx;
            ++ko;
        }
        __syncthreads();
        byPosition = merges == 0u;                           // (the same sum in every thread)
        cnt = keeps; nodeBase += merges;
        uint32_t* tr = refA; refA = refB; refB = tr;
        float4* t4 = minA; minA = minB; minB = t4;
        t4 = maxA; maxA = maxB; maxB = t4;
    }
}

// ---- the four-wide tree in the order the rays walk it ----
// The builders address a four-wide node by the record index of the binary node it replaces: about half of those slots are not reachable
// from the root (their binary nodes were absorbed by an ancestor's node), and a node's children lie wherever the build numbered them.
// Every L2 miss of the trace kernels pulls a 128-B line for a 64-B record (TCC_EA0_RDREQ_128B = TCC_EA0_RDREQ), so half of the HBM
// traffic of a scene beyond the caches carried slots nobody asked for; and beyond ~2 GiB a gather is bound by address translations, so
// records scattered over 6 GB of slots cost more than the same records packed.  This pass writes the reachable nodes, and only
// those, in a grouped depth-first order: the (up to four) inner children of a node side by side (a ray that enters a node usually
// looks at several of them: they share lines), then the subtree of each child, one after the other (what a ray touches near its hit
// lies within a few pages).  The triangles' hot records get the matching order (a node's triangle leaves side by side, subtree by
// subtree: `triLeaf`, with the primitive's number inside the record), so that a leaf node's triangles share lines too.
// Positions are a pure function of the subtree sizes (sizeN: four-wide nodes, sizeT: triangle leaves, by record index):
//   node X at pos(X), children group at gs(X), own triangles from ts(X):  child i -> gs(X) + i;  gs(child i) = gs(X) + k + sum_{j<i} (sizeN(c_j) - 1);
//   ts(child i) = ts(X) + m + sum_{j<i} sizeT(c_j)                        (k inner children, m triangle leaves of X)
// so one launch per level of the tree assigns them top-down: a record stamped by the previous launch writes its node at its position,
// with its refs renamed, and stamps its children.  The visit order of a ray does not depend on where a record lies: same frames, bit for bit.
__global__ void layoutRootKernel(uint32_t* stamp, uint32_t* pos, uint32_t* gs, uint32_t* ts) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamp[0] = 1u; pos[0] = 0u; gs[0] = 1u; ts[0] = 0u; }
}
// assign: triangle leaves get their positions here (posOfTri written); else they are looked up (a second tree over the same triangles)
__global__ __launch_bounds__(256) void layoutLevelKernel(const uint4* __restrict__ in, uint32_t nRecs, const uint32_t* __restrict__ sizeN, const uint32_t* __restrict__ sizeT,
                                                         uint32_t* stamp, uint32_t* pos, uint32_t* gs, uint32_t* ts, uint32_t level, uint4* __restrict__ out,
                                                         uint32_t* posOfTri, uint32_t triCount, int assign, const uint4* __restrict__ auxIn, uint4* __restrict__ auxOut) {
    for (uint32_t r = blockIdx.x * 256u + threadIdx.x; r < nRecs; r += gridDim.x * 256u) {
        if (stamp[r] != level) continue;
        const uint4 q0 = in[size_t(r) * 4], q1 = in[size_t(r) * 4 + 1];
        uint4 q2 = in[size_t(r) * 4 + 2], q3 = in[size_t(r) * 4 + 3];
        uint32_t refs[4] = {q2.z, q2.w, q3.x, q3.y};
        uint32_t k = 0, m = 0;
        for (int i = 0; i < 4; ++i) { k += inner(refs[i]) ? 1u : 0u; m += triLeaf(refs[i]) ? 1u : 0u; }
        const uint32_t g = gs[r], t = ts[r];
        uint32_t run = g + k, trun = t + m, ki = 0, mi = 0;
        for (int i = 0; i < 4; ++i) {
            const uint32_t ref = refs[i];
            if (inner(ref)) {
                if (ref >= nRecs) { refs[i] = LRL_REF_EMPTY4; continue; }      // (cannot happen for a tree the builders made)
                const uint32_t at = g + ki++;
                pos[ref] = at; gs[ref] = run; ts[ref] = trun;
                run += sizeN[ref] - 1u; trun += sizeT[ref];
                stamp[ref] = level + 1u;
                refs[i] = at;
            } else if (triLeaf(ref)) {
                const uint32_t prim = ref & LRL_PRIM_INDEX_MASK;
                uint32_t p = LRL_REF_NONE;
                if (prim < triCount) {
                    if (assign) { p = t + mi++; posOfTri[prim] = p; }
                    else p = posOfTri[prim];
                }
                refs[i] = p < triCount ? (LRL_REF_LEAF | (2u << LRL_PRIM_TYPE_SHIFT) | p) : LRL_REF_EMPTY4;      // (a triangle the first tree did not hold: no ray finds it)
            }
        }
        q2.z = refs[0]; q2.w = refs[1]; q3.x = refs[2]; q3.y = refs[3];
        uint4* o = out + size_t(pos[r]) * 4;
        o[0] = q0; o[1] = q1; o[2] = q2; o[3] = q3;
        if (auxIn) auxOut[pos[r]] = auxIn[r];          // what travels with a node (a caller's tree: where each item's box comes from, for the refit)
    }
}

// ---- a caller's tree (LrlSceneView.bvh: the reference's pre-order array of 64-B nodes, each with its own box, a primitive parked in any
// node; /root/reference/RayTracing/RayTracing.h: BVH::Node) -> the two device layouts, on the device ----
// What the host did until round 3 (validation stays there: it guards these kernels' indexing):
//   1. callerScanKernel: which nodes are inner (-> record numbers by a prefix sum: recOf; binOf the other way); who would gain from a box of
//      its own; whether the tree is nested (every child's box inside its parent's)
//   2. callerDecideKernel: the fit applies when the tree is nested and at least 1 triangle in 64 gains
//   3. callerLevelKernel, one launch per level of readiness, bottom-up: a record whose inner children were done by an EARLIER launch gets its
//      fitted box, its binary record (caller's boxes: the PARITY kernels walk the tree as the reference does), its four-wide node as if it
//      were the root of a collapse, where each item's box comes from (src4, for the refit below), and its four-wide subtree's sizes and height
// The fit: the reference tests a primitive parked in an inner node whenever the ray enters that node's box, which spans a whole subtree, and
// the Pool's tree has leaf boxes as tall as the highest wave (/root/reference/Pool/Pool.cpp:250-263).  A box fitted to the primitive says the
// same for every ray that can hit it and turns most of the others away: a triangle's own box (primMin / primMax, kept current by the update
// calls and the water step) with a margin `pad` on every side, where it lies inside the box the caller's tree gives that primitive; the
// caller's box otherwise (other primitive classes: they go through the kernels' cold path, cheapest when the lanes of a wave take it together,
// on arrival at the node; a triangle that has left its box: the caller's boxes are the contract, as for the reference); an inner node: the
// union of its items.  Only for a nested tree: then a ray that hits a primitive reaches it through either set of boxes.  The margin: a hit on
// the box's face (every hit on a triangle that lies in an axis plane, every hit on an edge that two neighbours' boxes share) must enter the box
// safely before the hit's distance, or one of two primitives at exactly the same distance would be turned away by the rounding of the slab
// arithmetic instead of being decided by its code (tests/test_gpu_parity.py::test_hand_over_kernels_agree).
struct CNode { float3 lo, hi; uint32_t left, right, axis, geometry, num; };
DI CNode ldNode(const LrlBVHNode* __restrict__ bvh, uint32_t b) {
    const uint4* q = reinterpret_cast<const uint4*>(bvh + b);
    const uint4 a = q[0], c = q[1], d = q[2];
    return {make_float3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z)), make_float3(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z)),
            a.w, c.w, d.y, d.z, d.w};
}
DI uint32_t primBits(const CNode& n) { return (n.geometry << LRL_PRIM_TYPE_SHIFT) | (n.num & LRL_PRIM_INDEX_MASK); }
struct OwnBoxes { const float4* pmin; const float4* pmax; uint32_t triCount; float pad; };
// box of node n's primitive -> 0: the caller's; 1: its own, which is the caller's plus the margin; 2: its own, tighter
DI int primBox(const OwnBoxes& o, const CNode& n, float3& lo, float3& hi) {
    lo = n.lo; hi = n.hi;
    if (!o.pmin || n.geometry != 2u || n.num >= o.triCount) return 0;
    const float4 a = o.pmin[n.num], b = o.pmax[n.num];
    const bool inside = a.x <= b.x && a.y <= b.y && a.z <= b.z && a.x >= lo.x && a.y >= lo.y && a.z >= lo.z && b.x <= hi.x && b.y <= hi.y && b.z <= hi.z;
    if (!inside) return 0;
    const bool same = a.x == lo.x && a.y == lo.y && a.z == lo.z && b.x == hi.x && b.y == hi.y && b.z == hi.z;
    lo = make_float3(a.x - o.pad, a.y - o.pad, a.z - o.pad); hi = make_float3(b.x + o.pad, b.y + o.pad, b.z + o.pad);
    return same ? 1 : 2;
}
DI bool within(const CNode& c, const CNode& p) {
    return c.lo.x <= c.hi.x && c.lo.y <= c.hi.y && c.lo.z <= c.hi.z && c.lo.x >= p.lo.x && c.lo.y >= p.lo.y && c.lo.z >= p.lo.z && c.hi.x <= p.hi.x && c.hi.y <= p.hi.y && c.hi.z <= p.hi.z;
}
// info words of the caller's-tree kernels
enum { CT_HEIGHT4 = 0, CT_GAINERS = 1, CT_NOT_NESTED = 2, CT_SIZE_N = 3, CT_SIZE_T = 4, CT_FITTED = 5, CT_ROOT_DONE = 6, CT_FIT = 7 };
__global__ __launch_bounds__(256) void callerScanKernel(const LrlBVHNode* __restrict__ bvh, uint32_t N, OwnBoxes own, uint32_t* __restrict__ flags, uint32_t* info) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    bool gain = false, loose = false;
    if (i < N) {
        const CNode n = ldNode(bvh, i);
        flags[i] = n.left ? 1u : 0u;
        if (own.pmin) {
            if (n.geometry) { float3 lo, hi; const int k = primBox(own, n, lo, hi); gain = n.left ? k != 0 : k == 2; }
            if (n.left) {
                loose = !within(ldNode(bvh, i + 1u), n);
                if (n.right) loose = loose || !within(ldNode(bvh, n.right), n);
            }
        }
    }
    const uint32_t gainers = uint32_t(__popcll(__ballot(gain)));
    const bool anyLoose = __ballot(loose) != 0ull;
    if ((threadIdx.x & 63u) == 0u) {
        if (gainers) atomicAdd(&info[CT_GAINERS], gainers);
        if (anyLoose) atomicOr(&info[CT_NOT_NESTED], 1u);
    }
}
__global__ __launch_bounds__(256) void callerBinOfKernel(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ recOf, uint32_t N, uint32_t* __restrict__ binOf) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < N && flags[i]) binOf[recOf[i]] = i;
}
__global__ void callerDecideKernel(uint32_t* info, uint32_t triCount, int haveOwn) {
    if (threadIdx.x == 0 && blockIdx.x == 0)
        info[CT_FIT] = (haveOwn && info[CT_NOT_NESTED] == 0u && info[CT_GAINERS] != 0u && uint64_t(info[CT_GAINERS]) * 64u >= triCount) ? 1u : 0u;
}
struct CItem { float3 lo, hi; uint32_t ref, src; };      // ref: record index or LRL_REF_LEAF | primitive; src: binary node (bit 31: the primitive parked in it)
struct CallerTree {
    const LrlBVHNode* bvh; const uint32_t* recOf; const float4* nodeMin; const float4* nodeMax; OwnBoxes own;
};
DI void childItem(const CallerTree& t, uint32_t c, CItem* out, int& k, uint32_t* fitted) {
    const CNode n = ldNode(t.bvh, c);
    if (n.left) {
        const uint32_t r = t.recOf[c];
        const float4 a = t.nodeMin[r], b = t.nodeMax[r];
        out[k++] = {make_float3(a.x, a.y, a.z), make_float3(b.x, b.y, b.z), r, c};
    } else if (n.geometry) {
        float3 lo, hi;
        const int own = primBox(t.own, n, lo, hi);
        if (fitted && own == 2) ++*fitted;
        out[k++] = {lo, hi, LRL_REF_LEAF | primBits(n), c};
    }
}
// the items of binary node b: its parked primitive, its children (a leaf without a primitive is none)
DI int contents(const CallerTree& t, uint32_t b, const CNode& n, CItem* out, uint32_t* fitted) {
    int k = 0;
    if (n.geometry) {
        float3 lo, hi;
        const int own = primBox(t.own, n, lo, hi);
        if (fitted && own != 0) ++*fitted;
        out[k++] = {lo, hi, LRL_REF_LEAF | primBits(n), b | 0x80000000u};
    }
    if (n.left) {
        childItem(t, b + 1u, out, k, fitted);
        if (n.right) childItem(t, n.right, out, k, fitted);
    }
    return k;
}
DI float area(const CItem& t) {
    const float dx = t.hi.x - t.lo.x, dy = t.hi.y - t.lo.y, dz = t.hi.z - t.lo.z;
    return dx * dy + dy * dz + dz * dx;
}
DI void writeCallerNode4(const CItem* it, int n, uint32_t r, uint4* node4, uint4* src4) {
    if (n) {
        Item4 q[4];
        for (int i = 0; i < n; ++i) q[i] = {it[i].lo, it[i].hi, it[i].ref};
        writeNode4(q, n, r, node4);
    } else {      // nothing below: four empty slots (inverted boxes)
        uint4* q = node4 + size_t(r) * 4;
        q[0] = make_uint4(0u, 0u, 0u, 127u << 23);
        q[1] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0u);
        q[2] = make_uint4(0u, 0u, LRL_REF_EMPTY4, LRL_REF_EMPTY4);
        q[3] = make_uint4(LRL_REF_EMPTY4, LRL_REF_EMPTY4, 127u << 23, 127u << 23);
    }
    src4[r] = make_uint4(n > 0 ? it[0].src : 0u, n > 1 ? it[1].src : 0u, n > 2 ? it[2].src : 0u, n > 3 ? it[3].src : 0u);
}
DI uint32_t leafRefOf(const CNode& n) { return LRL_REF_LEAF | (n.geometry << LRL_PRIM_TYPE_SHIFT) | (n.num & LRL_PRIM_INDEX_MASK); }
// records 1 .. recCount-1 (record r replaces binary node binOf[r]); the thread that does the root's also writes record 0, the root's virtual parent
__global__ __launch_bounds__(256) void callerLevelKernel(const LrlBVHNode* __restrict__ bvh, const uint32_t* __restrict__ recOf, const uint32_t* __restrict__ binOf, uint32_t recCount,
                                                         OwnBoxes own, uint32_t* doneAt, uint32_t launch, float4* nodeMin, float4* nodeMax, float4* recs, uint4* node4,
                                                         uint4* src4, uint32_t* h4, uint32_t* sizeN, uint32_t* sizeT, uint32_t* info) {
    const bool fit = info[CT_FIT] != 0u;
    if (!fit) own.pmin = nullptr;
    const CallerTree t{bvh, recOf, nodeMin, nodeMax, own};
    uint32_t fitted = 0;
    for (uint32_t r = 1u + blockIdx.x * 256u + threadIdx.x; r < recCount; r += gridDim.x * 256u) {
        if (doneAt[r] != 0u) continue;
        const uint32_t b = binOf[r];
        const CNode n = ldNode(bvh, b);
        const CNode l = ldNode(bvh, b + 1u);
        CNode rt = l;
        if (n.right) rt = ldNode(bvh, n.right);
        bool ready = true;
        if (l.left) { const uint32_t d = doneAt[recOf[b + 1u]]; ready = ready && d != 0u && d < launch; }
        if (n.right && rt.left) { const uint32_t d = doneAt[recOf[n.right]]; ready = ready && d != 0u && d < launch; }
        if (!ready) continue;
        // the binary record: both children's boxes as the caller has them
        {
            float4* q = recs + size_t(r) * 4;
            const uint32_t ownPrim = n.geometry ? primBits(n) : 0u;
            const uint32_t lref = l.left ? recOf[b + 1u] : leafRefOf(l);
            q[0] = make_float4(l.lo.x, l.lo.y, l.lo.z, __uint_as_float(lref));
            if (n.right) {
                const uint32_t rref = rt.left ? recOf[n.right] : leafRefOf(rt);
                q[1] = make_float4(l.hi.x, l.hi.y, l.hi.z, __uint_as_float(rref));
                q[2] = make_float4(rt.lo.x, rt.lo.y, rt.lo.z, __uint_as_float(ownPrim));
                q[3] = make_float4(rt.hi.x, rt.hi.y, rt.hi.z, __uint_as_float(n.axis));
            } else {
                q[1] = make_float4(l.hi.x, l.hi.y, l.hi.z, __uint_as_float(LRL_REF_NONE));
                q[2] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(ownPrim));
                q[3] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(n.axis));
            }
        }
        CItem it[8];
        int cnt = contents(t, b, n, it, &fitted);
        // this node's box: fitted = the union of its items; else the caller's
        float3 lo = n.lo, hi = n.hi;
        if (fit && cnt) {
            lo = it[0].lo; hi = it[0].hi;
            for (int i = 1; i < cnt; ++i) {
                lo = make_float3(fminf(lo.x, it[i].lo.x), fminf(lo.y, it[i].lo.y), fminf(lo.z, it[i].lo.z));
                hi = make_float3(fmaxf(hi.x, it[i].hi.x), fmaxf(hi.y, it[i].hi.y), fmaxf(hi.z, it[i].hi.z));
            }
        }
        nodeMin[r] = make_float4(lo.x, lo.y, lo.z, 0.0f); nodeMax[r] = make_float4(hi.x, hi.y, hi.z, 0.0f);
        // the collapse: an inner item is replaced by ITS items while the node stays within four, largest box first
        for (;;) {
            int best = -1, bestAdd = 0;
            float bestArea = -1.0f;
            CItem sub[3], bestSub[3];
            for (int i = 0; i < cnt; ++i) {
                if (!inner(it[i].ref)) continue;
                const float a = area(it[i]);
                if (!(a > bestArea)) continue;
                const int add = contents(t, it[i].src, ldNode(bvh, it[i].src), sub, nullptr);
                if (cnt - 1 + add > 4) continue;
                bestArea = a; best = i; bestAdd = add;
                for (int k = 0; k < add; ++k) bestSub[k] = sub[k];
            }
            if (best < 0) break;
            for (int i = best; i + 1 < cnt; ++i) it[i] = it[i + 1];
            --cnt;
            for (int k = 0; k < bestAdd; ++k) it[cnt++] = bestSub[k];
        }
        writeCallerNode4(it, cnt, r, node4, src4);
        uint32_t hw = 0, sN = 1u, sT = 0u;
        for (int i = 0; i < cnt; ++i)
            if (inner(it[i].ref)) { hw = max(hw, h4[it[i].ref]); sN += sizeN[it[i].ref]; sT += sizeT[it[i].ref]; }
            else if (triLeaf(it[i].ref)) ++sT;
        h4[r] = hw + 1u; sizeN[r] = sN; sizeT[r] = sT;
        if (b == 0u) {      // record 0: the root replaced by its items, then as above: the same node
            recs[0] = make_float4(n.lo.x, n.lo.y, n.lo.z, __uint_as_float(r));
            recs[1] = make_float4(n.hi.x, n.hi.y, n.hi.z, __uint_as_float(LRL_REF_NONE));
            recs[2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            recs[3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            writeCallerNode4(it, cnt, 0u, node4, src4);
            h4[0] = hw + 1u; sizeN[0] = sN; sizeT[0] = sT;
            info[CT_HEIGHT4] = hw + 1u; info[CT_SIZE_N] = sN; info[CT_SIZE_T] = sT; info[CT_ROOT_DONE] = launch;
        }
        doneAt[r] = launch;
    }
    // (each primitive is counted once: by the launch that does the node it is an item of, before any collapse)
    for (int o = 32; o; o >>= 1) fitted += __shfl_xor(fitted, o);
    if ((threadIdx.x & 63u) == 0u && fitted) atomicAdd(&info[CT_FITTED], fitted);
}
// a tree whose root is a leaf: record 0 alone
__global__ void callerLeafRootKernel(const LrlBVHNode* bvh, OwnBoxes own, float4* recs, uint4* node4, uint4* src4, uint32_t* h4, uint32_t* sizeN, uint32_t* sizeT, uint32_t* info) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (info[CT_FIT] == 0u) own.pmin = nullptr;
    const CNode n = ldNode(bvh, 0u);
    recs[0] = make_float4(n.lo.x, n.lo.y, n.lo.z, __uint_as_float(leafRefOf(n)));
    recs[1] = make_float4(n.hi.x, n.hi.y, n.hi.z, __uint_as_float(LRL_REF_NONE));
    recs[2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    recs[3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    CItem it[1];
    int cnt = 0;
    if (n.geometry) {
        float3 lo, hi;
        if (primBox(own, n, lo, hi) == 2) info[CT_FITTED] = 1u;
        it[cnt++] = {lo, hi, LRL_REF_LEAF | primBits(n), 0u};
    }
    writeCallerNode4(it, cnt, 0u, node4, src4);
    h4[0] = 1u; sizeN[0] = 1u; sizeT[0] = (cnt && triLeaf(it[0].ref)) ? 1u : 0u;
    info[CT_HEIGHT4] = 1u; info[CT_SIZE_N] = 1u; info[CT_SIZE_T] = sizeT[0]; info[CT_ROOT_DONE] = 1u;
}

// ---- refit of a caller's four-wide tree to where its triangles are now ("a refit kernel for Pool's moving water instead
// of tall static boxes", /root/reference/Pool/Pool.cpp:250-263) ----
// The topology stays (refs, layout, depth); every node's child boxes are made again, bottom-up, by the rule above (primBox; an inner child:
// the union of its own items).  src4 = per item the binary node of the caller's tree whose box that is (bit 31: the primitive parked in that node).
// One launch per level of readiness, as fitLevelKernel: a node is done by the first launch that finds its inner children done earlier.
// (one pass over nodes first, first + stride, ...: -> primitives of the nodes done by it that are bounded more tightly than by the caller's tree)
DI uint32_t refitPass(uint4* node4, uint32_t nodes, const uint4* __restrict__ src4, const LrlBVHNode* __restrict__ bvh, const OwnBoxes& own, float4* boxLo, float4* boxHi,
                      uint32_t* doneAt, uint32_t launch, uint32_t first, uint32_t stride) {
    uint32_t fitted = 0;
    for (uint32_t r = first; r < nodes; r += stride) {
        if (doneAt[r] != 0u) continue;
        const uint4 q2 = node4[size_t(r) * 4 + 2], q3 = node4[size_t(r) * 4 + 3], s = src4[r];
        const uint32_t refs[4] = {q2.z, q2.w, q3.x, q3.y}, srcs[4] = {s.x, s.y, s.z, s.w};
        bool ready = true;
        for (int i = 0; i < 4; ++i)
            if (inner(refs[i])) {
                const uint32_t d = refs[i] < nodes ? doneAt[refs[i]] : 1u;
                ready = ready && d != 0u && d < launch;
            }
        if (!ready) continue;
        Item4 it[4];
        int n = 0;
        float3 nlo = make_float3(INFINITY, INFINITY, INFINITY), nhi = make_float3(-INFINITY, -INFINITY, -INFINITY);
        for (int i = 0; i < 4; ++i) {
            const uint32_t ref = refs[i];
            if (ref == LRL_REF_EMPTY4) continue;                  // (empty slots are the last ones)
            float3 lo, hi;
            if (inner(ref) && ref < nodes) { const float4 a = boxLo[ref], b = boxHi[ref]; lo = make_float3(a.x, a.y, a.z); hi = make_float3(b.x, b.y, b.z); }
            else {
                const int k = primBox(own, ldNode(bvh, srcs[i] & 0x7fffffffu), lo, hi);
                fitted += ((srcs[i] >> 31) ? k != 0 : k == 2) ? 1u : 0u;
            }
            it[n++] = {lo, hi, ref};
            nlo = make_float3(fminf(nlo.x, lo.x), fminf(nlo.y, lo.y), fminf(nlo.z, lo.z));
            nhi = make_float3(fmaxf(nhi.x, hi.x), fmaxf(nhi.y, hi.y), fmaxf(nhi.z, hi.z));
        }
        if (n) writeNode4(it, n, r, node4);
        boxLo[r] = make_float4(nlo.x, nlo.y, nlo.z, 0.0f); boxHi[r] = make_float4(nhi.x, nhi.y, nhi.z, 0.0f);
        doneAt[r] = launch;
    }
    return fitted;
}
__global__ __launch_bounds__(256) void refitLevelKernel(uint4* node4, uint32_t nodes, const uint4* __restrict__ src4, const LrlBVHNode* __restrict__ bvh,
                                                        OwnBoxes own, float4* boxLo, float4* boxHi, uint32_t* doneAt, uint32_t launch, uint32_t* tight) {
    uint32_t fitted = refitPass(node4, nodes, src4, bvh, own, boxLo, boxHi, doneAt, launch, blockIdx.x * 256u + threadIdx.x, gridDim.x * 256u);
    for (int o = 32; o; o >>= 1) fitted += __shfl_xor(fitted, o);
    if ((threadIdx.x & 63u) == 0u && fitted) atomicAdd(tight, fitted);      // primitives bounded more tightly than by the caller's tree
}
// a small tree (the Pool: 3.3 k nodes, refitted after every water step of a one-sample frame): every level in ONE workgroup, a barrier
// between levels instead of a launch (what a workgroup writes is visible to it after the barrier)
#define LRL_REFIT_SMALL_NODES 8192u
__global__ __launch_bounds__(1024) void refitSmallKernel(uint4* node4, uint32_t nodes, const uint4* __restrict__ src4, const LrlBVHNode* __restrict__ bvh,
                                                         OwnBoxes own, float4* boxLo, float4* boxHi, uint32_t* doneAt, uint32_t levels, uint32_t* tight) {
    for (uint32_t r = threadIdx.x; r < nodes; r += 1024u) doneAt[r] = 0u;
    if (threadIdx.x == 0) *tight = 0u;
    __syncthreads();
    uint32_t fitted = 0;
    for (uint32_t launch = 1; launch <= levels; ++launch) {
        fitted += refitPass(node4, nodes, src4, bvh, own, boxLo, boxHi, doneAt, launch, threadIdx.x, 1024u);
        __syncthreads();
    }
    for (int o = 32; o; o >>= 1) fitted += __shfl_xor(fitted, o);
    if ((threadIdx.x & 63u) == 0u && fitted) atomicAdd(tight, fitted);
}
// triLeaf[posOfTri[i]] = triHot[i], with the triangle's number in the record (the kernels report hits by it)
__global__ __launch_bounds__(256) void triLeafGatherKernel(const float4* __restrict__ triHot, const uint32_t* __restrict__ posOfTri, uint32_t first, uint32_t count,
                                                           float4* __restrict__ triLeaf) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const uint32_t p = posOfTri[first + i];
    if (p == LRL_REF_NONE) return;
    const float4* s = triHot + size_t(first + i) * 4;
    float4* d = triLeaf + size_t(p) * 4;
    float4 h3 = s[3];
    h3.y = __uint_as_float(first + i);
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = h3;
}

}  // namespace lrl_build

size_t lrl_layout_scratch_bytes(uint32_t nRecs) { return (size_t(nRecs) * 4 + 255) / 256 * 256 * 4; }

// in: nRecs four-wide nodes by record index (root = record 0); sizeN / sizeT: subtree sizes by record index (device); levels: height of the
// four-wide tree (or more); out: sizeN[0] nodes; posOfTri: triCount entries (assign: written, LRL_REF_NONE where the tree holds no such
// triangle; else read).  scratch: lrl_layout_scratch_bytes(nRecs).
hipError_t lrl_layout_node4(const uint4* in, uint32_t nRecs, const uint32_t* sizeN, const uint32_t* sizeT, uint32_t levels, void* scratch, uint4* out,
                            uint32_t* posOfTri, uint32_t triCount, bool assign, const uint4* auxIn, uint4* auxOut, hipStream_t stream) {
    const size_t part = (size_t(nRecs) * 4 + 255) / 256 * 256;
    char* p = static_cast<char*>(scratch);
    uint32_t* stamp = reinterpret_cast<uint32_t*>(p);
    uint32_t* pos = reinterpret_cast<uint32_t*>(p + part);
    uint32_t* gs = reinterpret_cast<uint32_t*>(p + 2 * part);
    uint32_t* ts = reinterpret_cast<uint32_t*>(p + 3 * part);
    hipError_t e;
    if ((e = hipMemsetAsync(stamp, 0, size_t(nRecs) * 4, stream)) != hipSuccess) return e;
    if (assign && triCount && (e = hipMemsetAsync(posOfTri, 0xff, size_t(triCount) * 4, stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(lrl_build::layoutRootKernel, dim3(1), dim3(64), 0, stream, stamp, pos, gs, ts);
    const dim3 grid(std::min<uint32_t>((nRecs + 255u) / 256u, 16384u));
    for (uint32_t level = 1; level <= levels; ++level)
        hipLaunchKernelGGL(lrl_build::layoutLevelKernel, grid, dim3(256), 0, stream, in, nRecs, sizeN, sizeT, stamp, pos, gs, ts, level, out, posOfTri, triCount, assign ? 1 : 0, auxIn, auxOut);
    return hipGetLastError();
}

namespace {
struct CallerScratch { uint32_t *flags, *recOf, *binOf, *doneAt, *h4, *sizeN, *sizeT; float4 *nodeMin, *nodeMax; void* scanBuf; size_t scanTmp; size_t bytes; };
CallerScratch callerScratch(void* scratch, uint32_t N, uint32_t recCount) {
    char* p = static_cast<char*>(scratch);
    auto take = [&](size_t bytes) { char* o = p; p += (bytes + 255) / 256 * 256; return o; };
    CallerScratch c{};
    c.flags = reinterpret_cast<uint32_t*>(take(size_t(N) * 4));
    c.recOf = reinterpret_cast<uint32_t*>(take(size_t(N) * 4));
    c.binOf = reinterpret_cast<uint32_t*>(take(size_t(recCount) * 4));
    c.doneAt = reinterpret_cast<uint32_t*>(take(size_t(recCount) * 4));
    c.h4 = reinterpret_cast<uint32_t*>(take(size_t(recCount) * 4));
    c.sizeN = reinterpret_cast<uint32_t*>(take(size_t(recCount) * 4));
    c.sizeT = reinterpret_cast<uint32_t*>(take(size_t(recCount) * 4));
    c.nodeMin = reinterpret_cast<float4*>(take(size_t(recCount) * 16));
    c.nodeMax = reinterpret_cast<float4*>(take(size_t(recCount) * 16));
    uint32_t* v = nullptr;
    (void)rocprim::exclusive_scan(nullptr, c.scanTmp, v, v, 1u, size_t(N), rocprim::plus<uint32_t>(), hipStream_t(nullptr));
    c.scanBuf = take(c.scanTmp);
    c.bytes = size_t(p - static_cast<char*>(scratch));
    return c;
}
}  // namespace
size_t lrl_caller_tree_scratch_bytes(uint32_t N, uint32_t recCount) { return callerScratch(nullptr, N, recCount).bytes; }
// bvh: the caller's N nodes (validated on the host, on the device here); recCount = 1 + its inner nodes; levels: the binary tree's height (or
// more); pmin / pmax: the triangles' own boxes (NULL: none known, no fit).  Out: recs, node4 (by record index), src4: recCount entries each;
// sizes[0..1]: the four-wide subtree sizes in `scratch`, for lrl_layout_node4; info (host-visible once the stream is synchronised):
// {four-wide height, gainers, 1 if not nested, four-wide nodes, triangle leaves, primitives fitted more tightly, launch that did the root (0: not reached), 1 if fitted}
hipError_t lrl_caller_tree(const LrlBVHNode* bvh, uint32_t N, uint32_t recCount, uint32_t levels, const float4* pmin, const float4* pmax, uint32_t triCount, float pad,
                           void* scratch, float4* recs, uint4* node4, uint4* src4, const uint32_t** sizes, uint32_t* infoDev, hipStream_t stream) {
    if (!N || !recCount) return hipErrorInvalidValue;
    const CallerScratch c = callerScratch(scratch, N, recCount);
    sizes[0] = c.sizeN; sizes[1] = c.sizeT;
    hipError_t e;
    if ((e = hipMemsetAsync(infoDev, 0, 32, stream)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(c.doneAt, 0, size_t(recCount) * 4, stream)) != hipSuccess) return e;
    const lrl_build::OwnBoxes own{pmin, pmax, triCount, pad};
    const dim3 blk(256), grd((N + 255u) / 256u);
    hipLaunchKernelGGL(lrl_build::callerScanKernel, grd, blk, 0, stream, bvh, N, own, c.flags, infoDev);
    size_t tmp = c.scanTmp;
    if ((e = rocprim::exclusive_scan(c.scanBuf, tmp, c.flags, c.recOf, 1u, size_t(N), rocprim::plus<uint32_t>(), stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(lrl_build::callerBinOfKernel, grd, blk, 0, stream, c.flags, c.recOf, N, c.binOf);
    hipLaunchKernelGGL(lrl_build::callerDecideKernel, dim3(1), dim3(64), 0, stream, infoDev, triCount, pmin ? 1 : 0);
    if (recCount == 1u) {
        hipLaunchKernelGGL(lrl_build::callerLeafRootKernel, dim3(1), dim3(64), 0, stream, bvh, own, recs, node4, src4, c.h4, c.sizeN, c.sizeT, infoDev);
        return hipGetLastError();
    }
    const dim3 lgrd(std::min<uint32_t>((recCount + 255u) / 256u, 32768u));
    for (uint32_t launch = 1; launch <= levels; ++launch)
        hipLaunchKernelGGL(lrl_build::callerLevelKernel, lgrd, blk, 0, stream, bvh, c.recOf, c.binOf, recCount, own, c.doneAt, launch, c.nodeMin, c.nodeMax, recs, node4, src4,
                           c.h4, c.sizeN, c.sizeT, infoDev);
    return hipGetLastError();
}

size_t lrl_refit_scratch_bytes(uint32_t nodes) { return size_t(nodes) * (16 + 16 + 4) + 1024; }
// (-> the word after the nodes' stamps: how many primitives the refitted tree bounds more tightly than the caller's does)
const uint32_t* lrl_refit_tight_count(const void* scratch, uint32_t nodes) {
    return reinterpret_cast<const uint32_t*>(static_cast<const char*>(scratch) + size_t(nodes) * 32) + nodes;
}
// node4: `nodes` four-wide nodes (laid out or by record index) of a caller's tree `bvh`; src4: per node, the binary node behind each of its
// items (bit 31: its parked primitive); pmin / pmax: the triangles' own boxes as they are now; levels: height of the four-wide tree (or more)
hipError_t lrl_refit_node4(uint4* node4, uint32_t nodes, const uint4* src4, const LrlBVHNode* bvh, const float4* pmin, const float4* pmax, uint32_t triCount,
                           float pad, uint32_t levels, void* scratch, hipStream_t stream) {
    char* p = static_cast<char*>(scratch);
    float4* boxLo = reinterpret_cast<float4*>(p);
    float4* boxHi = boxLo + nodes;
    uint32_t* doneAt = reinterpret_cast<uint32_t*>(boxHi + nodes);
    if (nodes <= LRL_REFIT_SMALL_NODES) {
        hipLaunchKernelGGL(lrl_build::refitSmallKernel, dim3(1), dim3(1024), 0, stream, node4, nodes, src4, bvh, lrl_build::OwnBoxes{pmin, pmax, triCount, pad}, boxLo, boxHi, doneAt,
                           levels, doneAt + nodes);
        return hipGetLastError();
    }
    hipError_t e = hipMemsetAsync(doneAt, 0, size_t(nodes) * 4 + 4, stream);      // (and the count behind them)
    if (e != hipSuccess) return e;
    const dim3 grid(std::min<uint32_t>((nodes + 255u) / 256u, 16384u));
    for (uint32_t launch = 1; launch <= levels; ++launch)
        hipLaunchKernelGGL(lrl_build::refitLevelKernel, grid, dim3(256), 0, stream, node4, nodes, src4, bvh, lrl_build::OwnBoxes{pmin, pmax, triCount, pad}, boxLo, boxHi, doneAt, launch, doneAt + nodes);
    return hipGetLastError();
}

hipError_t lrl_launch_tri_leaf_gather(const float4* triHot, const uint32_t* posOfTri, uint32_t first, uint32_t count, float4* triLeaf, hipStream_t stream) {
    if (!count) return hipSuccess;
    hipLaunchKernelGGL(lrl_build::triLeafGatherKernel, dim3((count + 255u) / 256u), dim3(256), 0, stream, triHot, posOfTri, first, count, triLeaf);
    return hipGetLastError();
}

hipError_t lrl_launch_triangle_boxes(const LrlTriangle* in, uint32_t first, uint32_t count, float4* pmin, float4* pmax, hipStream_t stream) {
    if (!count) return hipSuccess;
    hipLaunchKernelGGL(lrl_build::triBoxKernel, dim3((count + 255) / 256), dim3(256), 0, stream, in, first, count, pmin, pmax);
    return hipGetLastError();
}

size_t lrl_build_scratch_bytes(uint32_t n) {
    size_t sortTmp = 0;
    uint64_t* k = nullptr;
    uint32_t* v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, sortTmp, k, k, v, v, size_t(n), 0, 63, hipStream_t(nullptr));
    const size_t N = n;
    // keys x2, idx x2, children, parents x2, counters, nodeMin, nodeMax, heights, h4, bounds + info
    size_t scanTmp = 0;
    (void)rocprim::exclusive_scan(nullptr, scanTmp, v, v, 0u, size_t(n), rocprim::plus<uint32_t>(), hipStream_t(nullptr));
    // + PLOC: refs x2, boxes x2 x2, nearest, merge, keep, two offsets
    return sortTmp + scanTmp + N * (16 + 8 + 8 + 8 + 4 + 32 + 4 + 4 + 8) + N * (8 + 64 + 20) + 4096 + 34 * 256;
}


// pmin / pmax: n primitive boxes (w of pmin = primitive bits).  recs / node4: n records each (n >= 2).  info (host-visible
// after the stream is synchronised): {height of the binary tree, height of the four-wide tree, 1 if a primitive box is not finite}.
// sizes[0..1] (may be NULL): where the build left the four-wide subtree sizes (nodes, triangle leaves; by record index) in `scratch`, for
// lrl_layout_node4; NULL pointers when this build did not make them (fewer than LRL_BUILD_LEVELS_FROM primitives).
hipError_t lrl_build_bvh(const float4* pmin, const float4* pmax, uint32_t n, float4* recs, uint4* node4, void* scratch, size_t scratchBytes,
                         uint32_t* infoDev, bool ploc, const uint32_t** sizes, hipStream_t stream) {
    if (sizes) sizes[0] = sizes[1] = nullptr;
    if (n < 2) return hipErrorInvalidValue;
    char* p = static_cast<char*>(scratch);
    auto take = [&](size_t bytes) { char* o = p; p += (bytes + 255) / 256 * 256; return o; };
    const size_t N = n;
    uint64_t* keysA = reinterpret_cast<uint64_t*>(take(N * 8));
    uint64_t* keysB = reinterpret_cast<uint64_t*>(take(N * 8));
    uint32_t* idxA = reinterpret_cast<uint32_t*>(take(N * 4));
    uint32_t* idxB = reinterpret_cast<uint32_t*>(take(N * 4));
    uint2* children = reinterpret_cast<uint2*>(take(N * 8));
    uint32_t* parentInner = reinterpret_cast<uint32_t*>(take(N * 4));
    uint32_t* parentLeaf = reinterpret_cast<uint32_t*>(take(N * 4));
    uint32_t* counters = reinterpret_cast<uint32_t*>(take(N * 4));
    float4* nodeMin = reinterpret_cast<float4*>(take(N * 16));
    float4* nodeMax = reinterpret_cast<float4*>(take(N * 16));
    uint32_t* heights = reinterpret_cast<uint32_t*>(take(N * 4));
    uint32_t* h4 = reinterpret_cast<uint32_t*>(take(N * 4));
    uint32_t* sizeN = reinterpret_cast<uint32_t*>(take(N * 4));      // four-wide subtree sizes by record index (builds by levels), for lrl_layout_node4
    uint32_t* sizeT = reinterpret_cast<uint32_t*>(take(N * 4));
    uint32_t* bounds = reinterpret_cast<uint32_t*>(take(256));
    size_t sortTmp = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, sortTmp, keysA, keysB, idxA, idxB, N, 0, 63, stream);
    if (e != hipSuccess) return e;
    void* sortBuf = take(sortTmp);
    if (size_t(p - static_cast<char*>(scratch)) > scratchBytes) return hipErrorOutOfMemory;

    const uint32_t initBounds[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    if ((e = hipMemcpyAsync(bounds, initBounds, sizeof(initBounds), hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(counters, 0, N * 4, stream)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(infoDev, 0, 32, stream)) != hipSuccess) return e;
    const dim3 blk(256), grd((n + 255) / 256);
    hipLaunchKernelGGL(lrl_build::boundsKernel, dim3(std::min<uint32_t>(grd.x, 4096u)), blk, 0, stream, pmin, pmax, n, bounds);
    hipLaunchKernelGGL(lrl_build::mortonKernel, grd, blk, 0, stream, pmin, pmax, n, bounds, keysA, idxA, infoDev);
    if ((e = rocprim::radix_sort_pairs(sortBuf, sortTmp, keysA, keysB, idxA, idxB, N, 0, 63, stream)) != hipSuccess) return e;
    if (ploc) {
        uint32_t* refA = reinterpret_cast<uint32_t*>(take(N * 4)); uint32_t* refB = reinterpret_cast<uint32_t*>(take(N * 4));
        float4* minA = reinterpret_cast<float4*>(take(N * 16)); float4* minB = reinterpret_cast<float4*>(take(N * 16));
        float4* maxA = reinterpret_cast<float4*>(take(N * 16)); float4* maxB = reinterpret_cast<float4*>(take(N * 16));
        uint32_t* nn = reinterpret_cast<uint32_t*>(take(N * 4));
        uint32_t* merge = reinterpret_cast<uint32_t*>(take(N * 4)); uint32_t* keep = reinterpret_cast<uint32_t*>(take(N * 4));
        uint32_t* mOff = reinterpret_cast<uint32_t*>(take(N * 4)); uint32_t* kOff = reinterpret_cast<uint32_t*>(take(N * 4));
        uint32_t* totals = reinterpret_cast<uint32_t*>(take(256));
        size_t scanTmp = 0;
        if ((e = rocprim::exclusive_scan(nullptr, scanTmp, merge, mOff, 0u, N, rocprim::plus<uint32_t>(), stream)) != hipSuccess) return e;
        void* scanBuf = take(scanTmp);
        if (size_t(p - static_cast<char*>(scratch)) > scratchBytes) return hipErrorOutOfMemory;
        const bool small = n < LRL_BUILD_LEVELS_FROM;
        if (small) hipLaunchKernelGGL(lrl_build::plocSmallKernel, dim3(1), dim3(LRL_PLOC_SMALL_THREADS), 0, stream, pmin, pmax, idxB, n, refA, minA, maxA, refB, minB, maxB,
                                      nn, children, parentInner, parentLeaf);
        else hipLaunchKernelGGL(lrl_build::plocInitKernel, grd, blk, 0, stream, pmin, pmax, idxB, n, refA, minA, maxA);
        uint32_t cnt = small ? 1u : n, nodeBase = 0, rounds = 0;
        while (cnt > 1u) {
            if (++rounds > 4096u) return hipErrorUnknown;
            const dim3 g((cnt + 255u) / 256u);
            hipLaunchKernelGGL(lrl_build::plocNearestKernel, g, blk, 0, stream, minA, maxA, cnt, nn);
            hipLaunchKernelGGL(lrl_build::plocMarkKernel, g, blk, 0, stream, nn, cnt, merge, keep);
            if ((e = rocprim::exclusive_scan(scanBuf, scanTmp, merge, mOff, 0u, size_t(cnt), rocprim::plus<uint32_t>(), stream)) != hipSuccess) return e;
            if ((e = rocprim::exclusive_scan(scanBuf, scanTmp, keep, kOff, 0u, size_t(cnt), rocprim::plus<uint32_t>(), stream)) != hipSuccess) return e;
            hipLaunchKernelGGL(lrl_build::plocApplyKernel, g, blk, 0, stream, refA, minA, maxA, nn, merge, keep, mOff, kOff, cnt, nodeBase, n, refB, minB, maxB,
                               children, totals);
            uint32_t tot[2] = {0, 0};
            if ((e = hipMemcpyAsync(tot, totals, 8, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
            if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
            if (tot[1] == 0u || tot[0] + tot[1] != cnt) return hipErrorUnknown;
            cnt = tot[0]; nodeBase += tot[1];
            std::swap(refA, refB); std::swap(minA, minB); std::swap(maxA, maxB);
        }
    } else
    hipLaunchKernelGGL(lrl_build::treeKernel, grd, blk, 0, stream, keysB, n, children, parentInner, parentLeaf);
    const bool byLevels = n >= LRL_BUILD_LEVELS_FROM;
    if (!byLevels) {
        hipLaunchKernelGGL(lrl_build::fitKernel, grd, blk, 0, stream, pmin, pmax, idxB, n, children, parentInner, parentLeaf, counters, nodeMin, nodeMax,
                           heights, recs, infoDev);
        hipLaunchKernelGGL(lrl_build::node4Kernel, grd, blk, 0, stream, recs, n, node4);
        if ((e = hipMemsetAsync(counters, 0, N * 4, stream)) != hipSuccess) return e;
        hipLaunchKernelGGL(lrl_build::height4Kernel, grd, blk, 0, stream, n, parentInner, parentLeaf, counters, node4, h4, infoDev);
        return hipGetLastError();
    }
    if (sizes) { sizes[0] = sizeN; sizes[1] = sizeT; }
    // by levels: `counters` (zeroed above) serves as doneAt.  The tree's height is not known beforehand: after a first run of launches the
    // host looks at info[0] (the root's height, written by the launch that fits the root) every few launches
    uint32_t rootHeight = 0;
    for (uint32_t launch = 1; launch <= LRL_MAX_BVH_DEPTH + 1u; ++launch) {
        hipLaunchKernelGGL(lrl_build::fitLevelKernel, dim3(std::min<uint32_t>(grd.x, 32768u)), blk, 0, stream, pmin, pmax, idxB, n, children, counters, launch, nodeMin, nodeMax, heights, recs,
                           node4, h4, sizeN, sizeT, infoDev);
        if (launch >= 16u && (launch & 7u) == 0u) {
            if ((e = hipMemcpyAsync(&rootHeight, infoDev, 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
            if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
            if (rootHeight) break;
        }
    }
    return hipGetLastError();
}
