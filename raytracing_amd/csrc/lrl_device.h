// Device-side scene layout and kernel launch interface shared by the C-ABI implementation
// (lrl_context.hip, lrl_scene.hip, lrl_schedule.hip) and the two kernel translation units (kernels_parity.hip, kernels_fast.hip).
//
// HBM layout (built by lrl_scene_upload from the reference's AoS records):
//   recs      4 x float4 per INNER node (64 B, one half cache line), boxes of both children in the parent:
//               {left.min, leftRef} {left.max, rightRef} {right.min, own primitive} {right.max, split axis}
//             a child ref is LRL_REF_NONE, an inner-record index, or LRL_REF_LEAF | primitive, where
//             primitive = type << 28 | index; "own primitive" is the primitive the reference's builder
//             parks in an inner node (0 = none).  Record 0 is a virtual parent of the root.  Leaves have no
//             record: a leaf visit costs no node fetch.  father / blank of the reference's 64-B node are
//             dropped: traversal keeps a per-lane stack of pending far children in LDS instead of the
//             reference's parent-pointer trail.
//   node4     4 x uint4 per node of the four-wide collapse of the same tree (64 B), child boxes as bytes relative to the
//             node's origin and power-of-two units (layout at buildNode4, lrl_scene.hip); indexed by the record index of
//             the binary node it replaces, so refs are shared with `recs`.  FAST lean kernel only.  Why: the vector L1
//             charges per lane and per 16-byte piece (tools/probes/l1_gather.hip: 171 CU-cycles for a wave whose lanes
//             each read their own 64 B from L1) and that, not arithmetic, bounds a one-lane-per-ray traversal: four
//             children per 64 B halve both the fetches and the chain of dependent trips of a ray.
//             Laid out by lrl_layout_node4 (lrl_build.hip): reachable nodes only, a node's children side by side, subtree after subtree.
//   triHot    4 x float4 per triangle (64 B, one half cache line): plane | p1,k1.x | k1.yz,k2.xy | k2.z, triangle number, material index
//   triLeaf   the same records in the order of node4's leaves (a leaf node's triangles side by side): what the four-wide kernels read
//   triCold   32 B per triangle: uv1 uv2 uv3, material index   (read once per accepted hit)
//   materials 80-B Color records, de-duplicated on upload (a mesh usually has a handful)
//   spheres / circles / cylinders / cones / lights: the reference's 128-B / 32-B records as they are
//   tex / cube: RGBX8 texels (converted from the BMP's BGR), 4 B per texel
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lrl_trace.h"
#include "../../include/lrl_types.h"

struct LrlTriCold {
    float uv1[2], uv2[2], uv3[2];
    uint32_t material;
    uint32_t _pad;
};
static_assert(sizeof(LrlTriCold) == 32, "TriCold");

#ifdef LRL_DIAG
#define LRL_TIMELINE_WORDS 6
#else
#define LRL_TIMELINE_WORDS 3
#endif
#define LRL_REF_NONE 0xffffffffu
#define LRL_REF_LEAF 0x80000000u
// An empty child slot of a FOUR-WIDE node: an inverted byte box (near planes 255, far planes 0), which fails the interval test by itself, so
// the FAST kernels need no "is there a child" test per slot; its ref is a leaf of type 0, which no test block accepts: should rounding ever
// let a ray through an inverted box (an origin 2^24 node sizes away), the lane spends one empty trip on it.
#define LRL_REF_EMPTY4 LRL_REF_LEAF
#define LRL_EMPTY4_LO 0xffu
#define LRL_PRIM_TYPE_SHIFT 28
#define LRL_PRIM_INDEX_MASK 0x0fffffffu
#ifndef LRL_LDS_STACK_ENTRIES
#define LRL_LDS_STACK_ENTRIES 12u   /* pending far children per lane kept in LDS; deeper entries spill to scratch */
#endif
#ifndef LRL_MIN_WAVES_PER_SIMD
#define LRL_MIN_WAVES_PER_SIMD 4    /* __launch_bounds__ second argument of the trace kernel (register budget) */
#endif

struct KScene {
    const float4* recs;
    const uint4* node4;      // four-wide compressed nodes, 4 x uint4 each, indexed like recs (FAST lean kernel)
    const float4* triHot;
    const float4* triLeaf;   // the hot records in the order of node4's leaves (the triangle's number in y of the last quarter); == triHot when the tree is not laid out
    const LrlTriCold* triCold;
    uint32_t triNeedUV;      // 0: no triangle material has a texture or the checker, so a hit on a triangle needs its normal and its material only:
                             // both are in the hot record (the wavefront shade kernels then leave the cold one alone: one line per hit instead of two)
    const LrlColor* materials;
    const LrlSphere* spheres;
    const LrlCircle* circles;
    const LrlCylinder* cylinders;
    const LrlCone* cones;
    const LrlPointLight* lights;
    const uchar4* tex;
    const uchar4* cube;      // 6 faces, face-major
    uint32_t lightNum;
    uint32_t texW, texH, texL;
    uint32_t cubeSize;
    float decayOrigin[4];
};

struct KFrame {
    const LrlTrans* trans;   // device array, one per sample (jitter shear already applied)
    uint32_t width, height;
    uint32_t rowBegin, rowEnd;
    uint32_t rows;                     // rows of this render's output (rowEnd - rowBegin, or the rows of the shard's stripes)
    uint32_t interleave, shard;        // > 1: this render covers the stripes s = shard (mod interleave) of LRL_STRIPE_ROWS rows (wavefront pipeline)
    float alpha;                       // alpha of the output: the share of the frame's samples this render sums (1 for all of them)
    uint32_t sampleBegin, sampleEnd;   // samples rendered by this launch
    uint32_t spp;                      // divisor of the mean
    uint32_t maxDepth;
    uint32_t stackEntries;             // LDS traversal-stack entries per lane: min(BVH depth + 1, LRL_LDS_STACK_ENTRIES)
    uint32_t diffuse;                  // LRL_FLAG_DIFFUSE_BOUNCE: cosine-weighted continuation instead of the mirror direction (wavefront pipeline)
    unsigned long long* timeline;      // diagnostic (env LRL_TIMELINE): per wave {start, end} in 100 MHz ticks + hardware id, else NULL
};

enum { LRL_CTR_PRIMARY = 0, LRL_CTR_REFLECT, LRL_CTR_REFRACT, LRL_CTR_SHADOW, LRL_CTR_NODES, LRL_CTR_TRI, LRL_CTR_SPH,
       LRL_CTR_OTHER, LRL_CTR_PRIMARY_HITS, LRL_CTR_FETCH, LRL_CTR_LANE_TRIPS, LRL_CTR_WAVE_TRIPS, LRL_CTR_GROUP_LANES, LRL_CTR_GROUP_ROUNDS, LRL_CTR_COUNT };

#ifndef LRL_BLOCK_THREADS
#define LRL_BLOCK_THREADS 256
#endif
#if LRL_BLOCK_THREADS == 256
#define LRL_TILE 16   // a 256-thread workgroup renders a 16x16 pixel tile, each wave an 8x8 quadrant
#elif LRL_BLOCK_THREADS == 64
#define LRL_TILE 8    // one-wave workgroups: an 8x8 tile each
#else
#error "LRL_BLOCK_THREADS must be 64 or 256"
#endif

// ---- wavefront form (lrl_wavefront.inc): one kernel launch per ray SEGMENT instead of one lane per pixel path ----
// Path slots are pixels in 8x8-tile-major order (slot = tile * 64 + y_in_tile * 8 + x_in_tile), so 64 consecutive
// slots are one wave's tile.  Ray index space: [0, P) = the closest-hit ray of slot r; [P * (1 + li) + slot] = the
// shadow ray of that slot towards light li.  A ray exists when its bit is set in maskC / maskS (one 64-bit word
// per 64 slots, written whole by one wave of the shade kernel: no atomics, no compaction, fixed addresses).
#ifndef LRL_WF_CHUNK_WORDS
#define LRL_WF_CHUNK_WORDS 1u     /* mask words (64 ray slots each) per one-wave trace workgroup: <= 4 rays per lane */
#endif
#ifndef LRL_WF_STACK
#define LRL_WF_STACK 10           /* pending-far-child entries per lane in LDS; deeper ones go to WFrame::overflow */
#endif
#ifndef LRL_WF_TRACE_WAVES
#define LRL_WF_TRACE_WAVES 8      /* __launch_bounds__ waves per SIMD of the trace kernel */
#endif
#ifndef LRL_WF_LEAN_WAVES
#define LRL_WF_LEAN_WAVES 8       /* __launch_bounds__ waves per SIMD of the lean trace kernel (64 VGPRs) */
#endif
#ifndef LRL_LEAN_BLOCK_WAVES
#define LRL_LEAN_BLOCK_WAVES 4    /* waves per workgroup of the lean trace kernel in sparse iterations (independent of each other) */
#endif
#ifndef LRL_LEAN_STACK
#define LRL_LEAN_STACK 8          /* its LDS rows of parked children per lane (8 rows + 1: 32 waves per CU fit in 160 KB) */
#endif
#ifndef LRL_SWARM_CAP
#define LRL_SWARM_CAP 1024         /* frontier entries of the wave-per-ray continuation kernel (LDS) */
#endif
#define LRL_LAYOUT_FROM (1u << 16)    /* device-built trees from this many primitives on are laid out (lrl_layout_node4); = LRL_BUILD_LEVELS_FROM of lrl_build.hip */
#define LRL_FEEDBACK_ITERS 64u        /* iterations (ray segments) whose hand-over thresholds adapt */
#define LRL_WORK_COUNTERS 4u         /* per iteration, 128 B apart: sampled work of the hand-over kernels (feedback) */
#ifndef LRL_GROUP_CAP
#define LRL_GROUP_CAP 256u           /* smallest frontier of one ray in the lanes-per-ray kernel (tests build it smaller); doubled for deep trees */
#endif
#define LRL_WF_SLOT_ALIGN (64u * LRL_WF_CHUNK_WORDS > 256u ? 64u * LRL_WF_CHUNK_WORDS : 256u)

#define LRL_WF_ALIVE 1u       /* path not finished */
#define LRL_WF_HASRAY 2u      /* a closest-hit ray of this path was traced in this iteration */
#define LRL_WF_PENDING 4u     /* shadow rays of the previous segment were traced: add their light now */
#define LRL_WF_FIRST 8u       /* the segment in flight is the primary ray */
#define LRL_WF_DECAY 16u      /* the path has come out of a refraction: its decay is in WFrame::decay (else it is the scene's decayOrigin) */
#define LRL_WF_DEPTH_SHIFT 8  /* 4 bits: depth */
#define LRL_WF_SP_SHIFT 12    /* 4 bits: refraction stack entries */

struct WFrame {
    uint32_t P;              // path slots of this pass = batch x Ppix, multiple of LRL_WF_SLOT_ALIGN
    uint32_t words;          // P / 64
    uint32_t tilesX;         // 8x8 tiles per row of tiles
    uint32_t sample;         // first sample index rendered by this pass
    uint32_t batch;          // samples traced together by this pass: slot = b * Ppix + pixel slot, sample = `sample` + b
    uint32_t Ppix;           // path slots of one sample (pixels in 8x8-tile-major order, padded)
    uint32_t blocksPerSample;   // Ppix / 256: a 256-thread block of generate / shade never straddles two samples
    uint32_t* feedbackHost;  // pinned host memory (device address), [iteration][8]: hand-over count + work counters of this pass, written by the
                             // shade kernel's first block (no copy on the caller's stream); NULL: not this pass
    uint32_t overflowEntries;   // per-lane entries of `overflow` (0: the LDS stack always suffices)
    uint32_t tilesY;         // rows of tiles
    float4* answer;          // xyz accumulated colour, w = v of the pending light terms
    float4* ratio;           // ratioNow
    float4* decay;           // decayNow (kept only when a material transmits: otherwise it never leaves decayOrigin)
    float2* hit;             // closest-hit result of the segment in flight: t, primitive bits
    float* hitAux;           //   cylinders and cones: which part was hit (written and read for those two kinds only)
    float* shadow;           // [light][P]: the light factor k of the segment's shadow ray, set to -1 by the trace kernels when it is occluded
    float4* pendA;           // pending light terms: diffuse colour (FAST: times the ratio at the hit), texD
    float4* pendB;           //                      PARITY: ratio at the hit, u (FAST keeps u in ratio.w)
    uint32_t* flags;
    float4* rayQ;            // closest-hit rays, 24 B each in two arrays: {origin, direction.x} here,
    float2* rayD;            //   {direction.y, direction.z} here (a closest-hit ray starts with t = -1: nothing to store)
    float4* shadowQ;         // [light][P] shadow rays: {origin, distance to the light}
    unsigned long long* maskC;
    unsigned long long* maskS;
    unsigned long long* maskL;   // per 64 path slots: which paths are unfinished (the shade kernel's own; flags of the others are not read)
    float4* refr;            // refraction stack [entry][4][P], NULL when no material transmits
    uint2* overflow;         // [trace workgroup][overflowEntries][64]
    uint32_t* alive;         // [iteration]: set to 1 by the shade kernel while any path is unfinished
    uint32_t zeroWords;      // words from `alive` on (alive, hand-over counts, work counters) that the generate kernel clears for its pass
    uint32_t zeroListWords;  // the same from `listCount` on (the generate kernel sees the array's start there)
    uint32_t* aliveHost;     // the same words in pinned host memory when the host polls them (scenes with transmission), else NULL
    // hand-over of long chains (lrl_lean.inc -> wfTraceSwarmKernel): a ray still going after `handoverTrips` trips leaves
    // its pending work here and a whole wave finishes it
    uint32_t* rayList;       // [kind][P]: the live rays of the iteration, compacted (wfCompactKernel), for the late, sparse segments
    uint32_t* listCount;     // [kind] for this iteration, zeroed with the frame's counters
    uint32_t groupCap;       // frontier entries per ray of the lanes-per-ray kernel (LRL_GROUP_CAP or twice that: picks the instantiation)
    uint32_t* contWork;      // LRL_WORK_COUNTERS counters, 128 B apart: frontier entries the hand-over kernel tested in this iteration
    uint32_t handoverTrips;  // 0: never hand over
    uint32_t contSlots;      // capacity of contHead / contRefs
    uint32_t contStride;     // refs per slot (>= deepest stack + 1)
    uint32_t* contCount;     // slots taken by this launch (zero before it)
    uint4* contHead;         // {ray index, tFar bits (ray.t, < 0: nothing hit), hit primitive, number of refs}
    uint32_t* contRefs;      // [slot][contStride]
    float4* out;
    uint2* hitOut;
    unsigned long long* counters;
    unsigned long long* timeline;   // diagnostic (env LRL_TIMELINE): per trace workgroup {start, end, hw id, max trips of one ray, sum of trips, wave trips}
};

#define LRL_WF_DECLARE(ns)                                                                                            \
    namespace ns {                                                                                                    \
    hipError_t wfLaunchGenerate(const KScene& sc, const KFrame& fr, const WFrame& wf, hipStream_t stream);            \
    hipError_t wfLaunchTrace(const KScene& sc, const WFrame& wf, hipStream_t stream);                                 \
    hipError_t wfLaunchShade(const KScene& sc, const KFrame& fr, const WFrame& wf, uint32_t iter, bool pool, hipStream_t stream); \
    hipError_t wfLaunchAccumulate(const KFrame& fr, const WFrame& wf, hipStream_t stream);                            \
    }
LRL_WF_DECLARE(lrl_parity)
LRL_WF_DECLARE(lrl_fast)
namespace lrl_fast {
hipError_t wfLaunchTraceLean(const KScene& sc, const WFrame& wf, bool sparse, bool compact, bool coop, bool phased, uint32_t streamWords, uint32_t refillMin, hipStream_t stream);   // phased: one block a trip   // sparse: few rays in many words (larger workgroups)
hipError_t wfLaunchTail(const KScene& sc, const KFrame& fr, const WFrame& wf, uint32_t iterFirst, uint32_t iterBound, bool pool, hipStream_t stream);   // the rest of a pass in one launch (lrl_lean.inc)
hipError_t wfLaunchTraceSwarm(const KScene& sc, const WFrame& wf, uint32_t waves, uint32_t group, bool coop, bool quad, hipStream_t stream);   // coop: four lanes to a record fetch (lrl_lean.inc)
}

// kernels_parity.hip / kernels_fast.hip
hipError_t lrl_launch_render_parity(const KScene& sc, const KFrame& fr, float4* out, uint2* hit, unsigned long long* counters,
                                    bool pool, hipStream_t stream);
hipError_t lrl_launch_render_fast(const KScene& sc, const KFrame& fr, float4* out, uint2* hit, unsigned long long* counters,
                                  bool pool, hipStream_t stream);
hipError_t lrl_launch_decay_origin(const KScene& sc, const LrlTrans* trans, uint32_t stackEntries, bool pool, LrlDecayOrigin* out,
                                   hipStream_t stream);
// triLeaf / posOfTri: the copy of the hot records in the order of the four-wide tree's leaves, also written when posOfTri != NULL
hipError_t lrl_launch_triangle_pre(const LrlTriangle* in, const uint32_t* material, uint32_t first, uint32_t count, float4* triHot,
                                   LrlTriCold* triCold, float4* triLeaf, const uint32_t* posOfTri, hipStream_t stream);
hipError_t lrl_launch_circle_pre(LrlCircle* circles, uint32_t count, hipStream_t stream);
// Device BVH build (lrl_build.hip): boxes of a staged slab of triangles; scratch size; the build itself (n >= 2 boxes, w of
// the min corner = primitive bits; writes n records into recs and node4, {height, four-wide height} into info)
hipError_t lrl_launch_triangle_boxes(const LrlTriangle* in, uint32_t first, uint32_t count, float4* pmin, float4* pmax, hipStream_t stream);
size_t lrl_build_scratch_bytes(uint32_t n);
hipError_t lrl_build_bvh(const float4* pmin, const float4* pmax, uint32_t n, float4* recs, uint4* node4, void* scratch, size_t scratchBytes,
                         uint32_t* infoDev, bool ploc, const uint32_t** sizes, hipStream_t stream);
// The four-wide tree in the order the rays walk it (lrl_build.hip): reachable nodes only, a node's children side by side, subtree after
// subtree, and the matching order of the triangles' hot records (triLeaf[posOfTri[i]] = triHot[i] with i inside the record).
size_t lrl_layout_scratch_bytes(uint32_t nRecs);
hipError_t lrl_layout_node4(const uint4* in, uint32_t nRecs, const uint32_t* sizeN, const uint32_t* sizeT, uint32_t levels, void* scratch, uint4* out,
                            uint32_t* posOfTri, uint32_t triCount, bool assign, const uint4* auxIn, uint4* auxOut, hipStream_t stream);
// A caller's tree (validated) -> binary records, four-wide nodes by record index, their items' sources (src4), subtree sizes; the fit to the
// triangles' own boxes where it applies (lrl_build.hip)
size_t lrl_caller_tree_scratch_bytes(uint32_t N, uint32_t recCount);
hipError_t lrl_caller_tree(const LrlBVHNode* bvh, uint32_t N, uint32_t recCount, uint32_t levels, const float4* pmin, const float4* pmax, uint32_t triCount, float pad,
                           void* scratch, float4* recs, uint4* node4, uint4* src4, const uint32_t** sizes, uint32_t* infoDev, hipStream_t stream);
// Refit of a caller's four-wide tree to the triangles' current boxes, in place, on the device (lrl_build.hip)
size_t lrl_refit_scratch_bytes(uint32_t nodes);
const uint32_t* lrl_refit_tight_count(const void* scratch, uint32_t nodes);
hipError_t lrl_refit_node4(uint4* node4, uint32_t nodes, const uint4* src4, const LrlBVHNode* bvh, const float4* pmin, const float4* pmax, uint32_t triCount,
                           float pad, uint32_t levels, void* scratch, hipStream_t stream);
hipError_t lrl_launch_tri_leaf_gather(const float4* triHot, const uint32_t* posOfTri, uint32_t first, uint32_t count, float4* triLeaf, hipStream_t stream);
hipError_t lrl_launch_quantize(const float4* in, uchar4* out, unsigned long long pixels, hipStream_t stream);   // RGBA32F -> RGBA8 (unorm)
// Pool water simulation: `substeps` x (acceleration, position) on gx * gy points {z, v, a, blank} (<= 16384), then the
// heights into the corners of the 2 (gx-1) (gy-1) water triangles `tris` (origin records on the device; NULL: no sync)
hipError_t lrl_launch_water_run(float4* water, uint32_t gx, uint32_t gy, float k, float dt, uint32_t substeps, LrlTriangle* tris, hipStream_t stream);
