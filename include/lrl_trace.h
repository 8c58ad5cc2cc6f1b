/*
 * lrl_trace.h — C ABI of liblrltrace.so, the MI355X (gfx950) implementation of the per-pixel
 * ray-tracing path of LRLVEC/RayTracing.  It replaces exactly the device side of one frame of
 * the reference (all citations relative to /root/reference):
 *
 *   reference call site                                   replaced by
 *   ----------------------------------------------------  ---------------------------------
 *   model.dataInit()                BVH/BVH.cpp:722        lrl_scene_upload
 *   TrianglePre / CirclePre         BVH/BVH.cpp:159-168    (inside lrl_scene_upload / _update_triangles)
 *   triangles re-upload each frame  Pool/Pool.cpp:654-655  lrl_scene_update_triangles
 *   Movement::run + dataInit        BVH/BVH.cpp:187-205    lrl_scene_update_primitives (+ lrl_scene_update_bvh)
 *   frameSizeUniform / transUniform BVH/BVH.cpp:718-720,   lrl_set_camera
 *     .dataInit()/.refreshData()        730,736
 *   decayOriginStorage.dataInit()   BVH/BVH.cpp:721        lrl_set_decay_origin
 *   decayOriginCalc.run()           BVH/BVH.cpp:130-138    lrl_compute_decay_origin
 *   renderer.use(); renderer.run()  BVH/BVH.cpp:741-742    lrl_render
 *     -> glDrawArrays(TRIANGLE_FAN) BVH/BVH.cpp:42
 *   glFinish()                      BVH/BVH.cpp:815        lrl_render is synchronous (lrl_render_async + lrl_sync otherwise)
 *
 * Conventions: plain pointers and sizes; caller owns every host array (borrowed for the call);
 * the library owns device memory; every function returns an int status (LRL_OK == 0), never throws.
 * One host thread per context; one context per GPU.
 * Output image: RGBA float32, W*H*4, row 0 = bottom row (gl_FragCoord convention), unclamped.
 */
#ifndef LRL_TRACE_H
#define LRL_TRACE_H

#include <stddef.h>
#include <stdint.h>

#include "lrl_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct LrlContext LrlContext;

enum LrlStatus {
    LRL_OK = 0,
    LRL_ERR_INVALID_ARG = 1,   /* null pointer, zero size, index out of range, malformed BVH */
    LRL_ERR_DEVICE = 2,        /* HIP runtime error; text via lrl_last_error */
    LRL_ERR_NO_SCENE = 3,      /* render before scene upload / camera set */
    LRL_ERR_BVH_DEPTH = 4,     /* BVH deeper than LRL_MAX_BVH_DEPTH */
    LRL_ERR_OUT_OF_MEMORY = 5,
    LRL_ERR_LIMIT = 6          /* a stated static limit exceeded (depth, spp, scratch) */
};

#ifndef LRL_MAX_RAY_DEPTH
#define LRL_MAX_RAY_DEPTH 10u   /* RayTraceDepth, BVH/shaders/RendererFragment0.cpp:2 (Pool: 8) */
#endif
#ifndef LRL_MAX_BVH_DEPTH
#define LRL_MAX_BVH_DEPTH 96u   /* traversal stack entries per lane the kernels can be launched with */
#endif
#define LRL_MAX_SPP 4096u

/* Shader variant: the Pool copy of the fragment shader lacks the tex == -2 checkerboard
 * (diff of Pool/shaders/RendererFragment0.cpp vs BVH/shaders/RendererFragment0.cpp:648-649,713-714). */
#define LRL_VARIANT_BVH 0u
#define LRL_VARIANT_POOL 1u

/* Arithmetic mode.  PARITY keeps the reference's operation order (per-node IEEE divides in the slab
 * test, no fused multiply-add); FAST hoists reciprocals and lets the compiler contract. */
#define LRL_MODE_PARITY 0u
#define LRL_MODE_FAST 1u

#define LRL_FLAG_COUNTERS 1u      /* accumulate LrlCounters for this render (slower kernel) */
#define LRL_FLAG_OUT_DEVICE 2u    /* rgba_out / hit_out are device pointers */
#define LRL_FLAG_MEGAKERNEL 4u    /* one launch, one lane per pixel path (the first design; kept for A/B and as the   */
                                  /* second schedule of the same arithmetic).  Default: wavefront pipeline, one       */
                                  /* trace + one shade launch per ray segment                                        */
#define LRL_FLAG_TIME_KERNELS 8u  /* between lrl_timing_begin/_end: time every trace-kernel launch on its own instead */
                                  /* of the frame as a whole (adds event records between the launches)               */
#define LRL_FLAG_COUNT_FETCHES 32u /* FAST mode: run the FAST trace kernels themselves in a counting instantiation (ray counters + */
                                  /* LrlCounters.recordFetches; node / primitive-test counts stay 0: LRL_FLAG_COUNTERS counts those  */
                                  /* in the reference's visit order).  Slower: for the roofline's byte count, not for timing         */
#define LRL_FLAG_DIFFUSE_BOUNCE 16u /* SURVEY 8d, C5 (no counterpart in the reference's shader, whose a11 step            */
                                  /* `RendererFragment0.cpp:718-740` always mirrors): the ray that continues after a   */
                                  /* hit takes a cosine-weighted direction about the normal on the incoming side, so    */
                                  /* that secondary rays are incoherent.  Philox4x32-10, counter (pixel = y*width + x,  */
                                  /* sample, depth of the segment that hit, 0), key (0, 0): u1, u2 = top 24 bits;       */
                                  /* dir = sqrt(u1)(cos(2 pi u2) T + sin(2 pi u2) B) + sqrt(1-u1) N.  Everything else   */
                                  /* (materials, lights, refraction, `ratioNow *= r`) is unchanged.  Not with           */
                                  /* LRL_FLAG_MEGAKERNEL.  Oracle: oracle_render with OracleParams.diffuse.             */

typedef struct LrlSceneView {
    /* geometry in the reference's record layouts; counts in `num` */
    LrlGeometryNum        num;
    const LrlTriangle*    trianglesOrigin;   /* origin form (TrianglePre runs on the device) ...            */
    const LrlTriangleGPU* trianglesGPU;      /* ... or already in GPU form; exactly one may be non-NULL     */
    const LrlSphere*      spheres;
    const LrlCircle*      circles;           /* un-normalised, CirclePre runs on the device                 */
    const LrlCylinder*    cylinders;
    const LrlCone*        cones;
    const LrlPointLight*  pointLights;
    const LrlBVHNode*     bvh;               /* pre-order array produced by the CPU builder (model.bvh, SSBO 9); NULL (with
                                              * bvhNodeNum 0): the library builds a tree itself, on the device (SURVEY 8f
                                              * rank 1: LBVH; needs trianglesOrigin).  Not the reference's tree: the images
                                              * agree unless two primitives are hit at exactly the same distance        */
    uint32_t              bvhNodeNum;
    /* textures: BMP payload convention, BGR8, rows bottom-up */
    uint32_t              texWidth, texHeight, texLayers;
    const uint8_t*        texBGR;            /* texLayers * texHeight * texWidth * 3, may be NULL           */
    uint32_t              cubeSize;          /* 0 = no cube map (miss colour 0)                             */
    const uint8_t*        cubeBGR[6];        /* +X,-X,+Y,-Y,+Z,-Z                                           */
} LrlSceneView;

typedef struct LrlRenderParams {
    uint32_t     maxDepth;       /* RayTraceDepth: 10 (BVH), 8 (Pool), 4 (headline config); 1..LRL_MAX_RAY_DEPTH */
    uint32_t     variant;        /* LRL_VARIANT_*                                                            */
    uint32_t     mode;           /* LRL_MODE_*                                                               */
    uint32_t     flags;          /* LRL_FLAG_*                                                               */
    uint32_t     spp;            /* >= 1; image = mean over samples, summed in sample order                  */
    const float* jitter;         /* spp*2 sub-pixel offsets (pixels), NULL = pixel centres; applied as the   */
                                 /* shear of `trans` described in SURVEY §8d                                 */
    uint32_t     rowBegin, rowEnd; /* render rows [rowBegin,rowEnd); 0,0 = whole frame (multi-GPU stripes)   */
    uint32_t     sampleBegin, sampleEnd; /* samples [begin,end) of the jitter table; 0,0 = all; the output   */
                                 /* is then the SUM over that range divided by spp (partial means add up;    */
                                 /* alpha = the range's share of spp, so the partial alphas add up to 1)     */
    uint32_t     rowInterleave, rowShard; /* rowInterleave = N > 1 (with rowBegin = rowEnd = 0): the frame is cut  */
                                 /* into stripes of LRL_STRIPE_ROWS rows and this render covers the stripes    */
                                 /* s = rowShard (mod N), one after the other in its output (lrl_shard_rows    */
                                 /* rows): the N shards of a frame balance its load and are all-gathered by    */
                                 /* the caller (multi-GPU strong scaling).  0 or 1: rowBegin / rowEnd apply.   */
                                 /* Not with LRL_FLAG_MEGAKERNEL.                                             */
} LrlRenderParams;
#define LRL_STRIPE_ROWS 16u
/* rows of shard `shard` of `interleave` of a frame `height` rows high */
uint32_t lrl_shard_rows(uint32_t height, uint32_t interleave, uint32_t shard);

typedef struct LrlCounters {
    uint64_t raysPrimary, raysReflect, raysRefract, raysShadow;
    uint64_t nodeVisits;         /* box tests (one per node visit, closest-hit + any-hit traversals)         */
    uint64_t triTests, sphereTests, otherTests;
    uint64_t primaryHits;
    uint64_t recordFetches;      /* LRL_FLAG_COUNT_FETCHES: 64-B records (four-wide nodes, triangle hot      */
                                 /* records) the FAST trace kernels fetched; 0 otherwise                      */
    uint64_t laneTrips;          /* LRL_FLAG_COUNT_FETCHES: lanes that carried a live ray (or a frontier entry), summed over the     */
    uint64_t waveTrips;          /* trips (rounds) of the FAST trace kernels' waves, and those trips: laneTrips / (64 x waveTrips)   */
                                 /* is the share of the issued vector work that was some ray's own                                    */
    uint64_t groupLanes;         /* the hand-over kernels' part of those two sums (lanes with a frontier entry, rounds of their waves) */
    uint64_t groupRounds;
} LrlCounters;

/* Scheduling knobs of the FAST-mode pipeline (no counterpart in the reference, whose driver schedules its fragment shader).
 * None of them changes which primitive a ray hits except at exact ties (FAST bars, tests/test_gpu_parity.py); PARITY renders
 * ignore all but batchSlots / pollLag / debug.  lrl_create sets the defaults; lrl_set_tuning replaces them (and resets the
 * adaptive thresholds).  The library reads no environment variable on the render path. */
#define LRL_TUNING_LIST_MAX 8u
#define LRL_HANDOVER_AUTO 0xffffffffu
#define LRL_TAIL_NEVER 0xffffffffu
typedef struct LrlTuning {
    uint32_t handoverEarly;      /* lean kernel: trips after which a ray still going is handed to the lanes-per-ray kernel, */
    uint32_t handoverLate;       /*   first two segments / later segments; 0 = never hand over; LRL_HANDOVER_AUTO (default) = */
                                 /*   48 / 16; 64 / 32 for launches of 2^23 path slots and more (batches of samples, 4K);       */
                                 /*   96 / 48 for scenes beyond the caches (more than 1 GiB of node records)                    */
    uint32_t handoverListLen;    /* > 0: explicit thresholds per segment instead (the last one repeats)                     */
    uint32_t handoverList[LRL_TUNING_LIST_MAX];
    uint32_t adapt;              /* 1: thresholds ratchet up from the sampled work counters of earlier frames               */
    uint32_t groupLanes;         /* lanes per handed-over ray: 8, 16, 32, or 64 (a wave per ray); 0 = 16 (a wave per ray for */
                                 /*   trees too deep for a group's frontier)                                                  */
    uint32_t contSlots;          /* hand-over records per launch; 0 = 262144 per 2^21 path slots of a launch                */
    uint32_t groupCapX;          /* frontier entries per ray = 256 x this (1, 2 or 4); 0 = by tree depth                    */
    uint32_t sparseFrom;         /* first segment traced with four-wave workgroups (5)                                      */
    int32_t  compactFrom;        /* first segment traced from compacted ray lists; 0 never; -1 = by regime                  */
    float    workLow;            /* frontier entries per handed-over ray below which a threshold doubles (12)               */
    uint32_t pollLag;            /* scenes with transmission: iterations the host polls behind the queue (1..3) once a pass
                                  * outlasts the iterations the last pass needed (those are enqueued without looking);
                                  * 0: look after every iteration, one behind the queue                                    */
    uint32_t traceRef;           /* 1: FAST arithmetic on the reference-order kernel (diagnostic)                           */
    uint32_t batchSlots;         /* spp > 1: samples are traced together up to this many path slots per launch;             */
                                 /*   0 = 2^26, less where the ray state of a pass would pass 24 GiB; passes equally long   */
    uint32_t coopFetch;          /* 1: the trace kernels fetch a 64-B record with four lanes (same records, same frame bit for bit).  A uniformly */
                                 /*   random gather out of more than ~2 GiB is bound by address translations per lane and gains 2.3x that way    */
                                 /*   (tools/probes/hbm_gather.hip); the rays of the measured scenes are more coherent and lose ~12 %.  0: off   */
    uint32_t groupQuad;          /* 1: the lanes-per-ray hand-over kernel (groupLanes 16 / 32) puts four lanes to a frontier entry, one child each */
    uint32_t streamWords;        /* > 1: the lean trace kernel runs as waves that own this many 64-ray words of a launch and hand the next rays  */
    uint32_t refillMin;          /*   of their words to idle lanes whenever at least refillMin (16) lanes are idle; 0 / 1: one word per wave        */
    uint32_t phasedTrips;        /* triangle trips of the lean kernel: a trip on which many lanes hold a triangle runs the triangle block alone and the  */
                                 /*   node lanes wait with their record (same frame bit for bit; faster on flat trees and incoherent rays, slower on     */
                                 /*   coherent ones): 1 on, 2 off, 0 = the library times one frame each way per scene and frame shape and keeps the faster */
    uint32_t deviceBuilder;      /* the library's own tree (lrl_scene_upload without a tree, lrl_scene_rebuild_bvh): 0 / 2 = clusters merged with   */
                                 /*   their nearest neighbour in Morton order (PLOC); 1 = binary radix tree over the Morton codes (LBVH): half the   */
                                 /*   build time (0.23 against 0.48 ms for the Pool) for frames 5-10 % slower, the better choice only for a small      */
                                 /*   scene that is rebuilt for every one- or two-sample frame                                                         */
    uint32_t debug;              /* 1: per-frame diagnostics on stderr (adds host waits); 2: a line around every HIP call of the   */
                                 /*   render path instead (no added waits); 3: both                                                */
    uint32_t treeLayout;         /* the four-wide tree of the FAST kernels: 0 = laid out in the order the rays walk it at upload / build (reachable   */
                                 /*   nodes only, a node's children side by side, subtree after subtree; the triangles' hot records in the order of   */
                                 /*   the leaves: fewer and fuller cache lines, fewer pages; every caller's tree, library-built trees from 65 536      */
                                 /*   primitives on); 1 = left as built, addressed by binary record index (for A/B measurements: same frames bit for  */
                                 /*   bit).  Read by lrl_scene_upload, lrl_scene_update_bvh and lrl_scene_rebuild_bvh                                  */
    uint32_t deterministic;      /* 1: the library decides nothing from measurements: no timed trial of triangle trips (off unless  */
                                 /*   phasedTrips = 1), no adaptation of the hand-over thresholds from earlier frames' counters, no   */
                                 /*   hand-over launches left out.  Every frame of a scene and                                        */
                                 /*   frame shape then issues the same launches, the first like the hundredth (the image is the same  */
                                 /*   bit for bit either way; what is given up is the few per cent the adaptation finds).  0: adaptive */
    uint32_t tailFrom;           /* from this iteration (ray segment) on the rest of a pass is ONE launch in which every workgroup    */
                                 /*   traces and shades its own 64 path slots until they are finished (same frame bit for bit; pays    */
                                 /*   where tens of late iterations carry a few hundred paths: C1 2.11 -> 1.24 ms).  0 = 5 for a scene  */
                                 /*   in which a material transmits, never otherwise; LRL_TAIL_NEVER = never                           */
} LrlTuning;

/* ---- life cycle ---- */
int lrl_create(int device, LrlContext** out);
int lrl_destroy(LrlContext* ctx);
const char* lrl_last_error(const LrlContext* ctx);   /* never NULL */
int lrl_version(void);
int lrl_get_tuning(const LrlContext* ctx, LrlTuning* out);
int lrl_set_tuning(LrlContext* ctx, const LrlTuning* tuning);

/* ---- scene (model.dataInit) ---- */
int lrl_scene_upload(LrlContext* ctx, const LrlSceneView* scene);
/* Pool: water triangles rewritten every frame; re-runs TrianglePre on [first, first+count), and refits a caller's fitted four-wide
 * tree to where the triangles are now (on the device, in stream order; LrlSceneInfo.ownBoxes). */
int lrl_scene_update_triangles(LrlContext* ctx, uint32_t first, uint32_t count, const LrlTriangle* tris);
/* The same update for callers that cannot allocate (a tight serving loop, a harness that forbids hipMalloc, host code that runs
 * beside a HIP graph capture).  No counterpart in the reference.  Same arguments, same results; it stages the records in a static
 * device array of the library (LRL_NOALLOC_MAX_TRIANGLES x 164 B = 10.25 MiB per GPU, held until the process ends) instead of
 * buffers sized to the call, through a pinned host mirror of the same size that lrl_create made, and on its success path calls no
 * hipMalloc / hipFree or any other allocating routine, device or host (an error return composes its message, which may allocate
 * host memory).  It is not itself capturable: it returns when its kernels are done (hipStreamSynchronize), because the caller's
 * array is borrowed only for the call.  Limits, refused with
 * LRL_ERR_LIMIT: count > LRL_NOALLOC_MAX_TRIANGLES; a triangle whose material is not yet in the scene's table (growing the table
 * allocates).  Every context on one GPU shares that GPU's array: calls that overlap on one GPU (two contexts, two threads)
 * race; calls on one context follow each other (the call returns when its kernels are done). */
#define LRL_NOALLOC_MAX_TRIANGLES 65536u
int lrl_scene_update_triangles_noalloc(LrlContext* ctx, uint32_t first, uint32_t count, const LrlTriangle* tris);
/* Movement::run (BVH/BVH.cpp:187-205: spheres and a point light move, then model.dataInit() uploads everything again, :203; disabled at
 * :726 in the shipped demo): replaces records [first, first + count) of ONE primitive class in place, in the reference's record layout
 * (CirclePre runs on new circles).  kind = the BVH geometry code of the class, or LRL_PRIM_LIGHT.  Lights need nothing else.  Geometry is
 * found through the tree, so when a primitive may have left its leaf box follow with
 *   lrl_scene_update_bvh(ctx, model.bvh, n)  caller's tree: the tree the host builder made for the new positions, over the SAME primitive
 *                                            counts (what model.dataInit() uploads to SSBO 9); validated like lrl_scene_upload's
 *   lrl_scene_rebuild_bvh(ctx)               library's tree: this call has already refreshed the leaf boxes of the new records */
#define LRL_PRIM_SPHERE 3u
#define LRL_PRIM_CIRCLE 4u
#define LRL_PRIM_CYLINDER 5u
#define LRL_PRIM_CONE 6u
#define LRL_PRIM_LIGHT 0x100u
int lrl_scene_update_primitives(LrlContext* ctx, uint32_t kind, uint32_t first, uint32_t count, const void* records);
int lrl_scene_update_bvh(LrlContext* ctx, const LrlBVHNode* bvh, uint32_t bvhNodeNum);
/* read back device results of the pre-compute kernels (tests / debugging) */
int lrl_scene_read_triangles_gpu(LrlContext* ctx, uint32_t first, uint32_t count, LrlTriangleGPU* out);
int lrl_scene_read_circles(LrlContext* ctx, uint32_t first, uint32_t count, LrlCircle* out);

/* ---- per-frame state ---- */
int lrl_set_camera(LrlContext* ctx, const LrlTrans* trans, uint32_t width, uint32_t height);
int lrl_set_decay_origin(LrlContext* ctx, const float decayOrigin[4]);
/* runs DecayOriginPre (8 probe rays) + DecayOrigin (mode vote) on the device for the current camera;
 * `variant` selects the BVH (sin-hash directions) or Pool (+z) probe; result stored and optionally returned */
int lrl_compute_decay_origin(LrlContext* ctx, uint32_t variant, LrlDecayOrigin* out /* may be NULL */);

/* ---- the frame (renderer.run) ---- */
int lrl_render(LrlContext* ctx, const LrlRenderParams* params,
               float* rgba_out /* W*(rowEnd-rowBegin)*4 floats */,
               uint32_t* hit_out /* optional: per pixel {type, index} of the primary hit (of the first sample rendered), 0 = miss */);
int lrl_render_async(LrlContext* ctx, const LrlRenderParams* params, float* rgba_out_dev, uint32_t* hit_out_dev,
                     void* hip_stream);
int lrl_sync(LrlContext* ctx);
/* The image the reference shows: its fragment output lands in the default framebuffer, RGBA8, clamped to [0, 1] (SURVEY 8b
 * "Output"; BVH/BVH.cpp:42 draws into the window).  Converts `pixels` RGBA32F values on the device (rgba_dev: e.g. the output
 * of lrl_render_async) to 8-bit unsigned normalised values, round(clamp(v, 0, 1) * 255) as OpenGL's float -> unorm conversion,
 * into out_dev (4 bytes per pixel, device memory), on the context's stream.  For A/B against a screenshot of the reference. */
int lrl_quantize_rgba8(LrlContext* ctx, const float* rgba_dev, uint8_t* out_dev, uint64_t pixels);

/* ---- measurement ---- */
int lrl_get_counters(LrlContext* ctx, LrlCounters* out);   /* of the last render with LRL_FLAG_COUNTERS */
/* duration of the trace kernel(s) of the last render, HIP events on the stream they ran on */
int lrl_last_kernel_ms(LrlContext* ctx, float* ms);
/* accumulate device time of every render between _begin and _end (HIP event pairs on the stream the kernels ran
 * on; _end waits for them): per frame (first to last kernel of the render), or per trace-kernel launch when the
 * render carries LRL_FLAG_TIME_KERNELS; at most LRL_MAX_TIMED_LAUNCHES pairs are kept */
#define LRL_MAX_TIMED_LAUNCHES 4096u
int lrl_timing_begin(LrlContext* ctx);
int lrl_timing_end(LrlContext* ctx, double* total_ms, uint32_t* launches);
/* Rebuild the tree the library built itself (scene uploaded with LrlSceneView.bvh == NULL) from the CURRENT triangles, after
 * lrl_scene_update_triangles or lrl_water_step moved some: the reference has no counterpart (its Pool keeps tall static boxes
 * around the moving water, /root/reference/Pool/Pool.cpp:250-263; SURVEY 8f rank 1 asks for a refit instead; a rebuild of a
 * small scene costs the same and gives tight boxes).  Error when the caller supplied the tree: such a tree keeps its topology and is
 * refitted to the moved triangles by the update calls and lrl_water_step themselves (a kernel per level, in stream order). */
int lrl_scene_rebuild_bvh(LrlContext* ctx);

/* ---- Pool water simulation (replaces WaterSimulation of /root/reference/Pool/Pool.cpp:46-235 and the
 * shaders Pool/shaders/WaterAccelerationCompute0.cpp, WaterPositionCompute0.cpp, WaterPositionSyncCompute0.cpp) ----
 * The water is a gx x gy grid of points (Pool/Pool.cpp:84-90, buffer binding 10); its 2 (gx-1) (gy-1) triangles are
 * triangles [firstTriangle, ...) of the uploaded scene (the reference keeps them at the head, :128-129).
 * lrl_water_init   = water.dataInit() (:236-239): points and parameters {k, dt} (UBO 4) to the device, plus the origin
 *                    records of the water triangles (the sync writes their corner heights; `triangles` = the same records
 *                    that were uploaded with the scene, borrowed during the call).
 * lrl_water_step   = WaterSimulation::run() (:224-235): n x (acceleration, position), then the position sync, followed by
 *                    TrianglePre of the water triangles (what model.upToDate + tracerInit.run() do next, :654-655); all
 *                    enqueued on the context's stream, no host wait.  n = 0 only syncs (the reference's init, :649-650).
 * lrl_water_read   : points and / or the water triangles' origin records back to the host (tests; either may be NULL).
 * gx * gy <= 16384.  Results are bit-identical to the reference's shaders (tests/test_gpu_parity.py). */
int lrl_water_init(LrlContext* ctx, uint32_t gx, uint32_t gy, float k, float dt, const LrlWaterPoint* points,
                   uint32_t firstTriangle, const LrlTriangle* triangles);
int lrl_water_step(LrlContext* ctx, uint32_t substeps);
int lrl_water_read(LrlContext* ctx, LrlWaterPoint* points, LrlTriangle* triangles);

/* what the library holds / did, for reports and tests */
typedef struct LrlSceneInfo {
    uint32_t bvhNodes;       /* nodes of the binary tree (the caller's, or 2 n - 1 of the device-built one)               */
    uint32_t bvhRecords;     /* 64-B inner-node records (and four-wide node slots)                                        */
    uint32_t bvhDepth;       /* depth of the binary tree                                                                   */
    uint32_t wideDepth;      /* depth of the four-wide tree                                                                */
    uint32_t deviceBuilt;    /* 1: built by the library on the device                                                      */
    uint32_t materials;      /* distinct Color records                                                                     */
    uint32_t ownBoxes;       /* a caller's tree: triangles that the four-wide tree of the FAST kernels bounds by their own boxes,
                                more tightly than the caller's tree does (one parked in an inner node, which the reference tests
                                whenever the ray enters the node; one whose leaf box was made larger than it is).  The tree follows
                                its triangles: after update calls and water steps it is refitted on the device, in stream order,
                                and this is the last refit's count (the call then waits for it).  A triangle that has left the
                                box the caller's tree gives it is bounded by that box again, as the reference's Pool bounds its
                                moving water by boxes made tall enough (Pool/Pool.cpp:250-263), and is not counted.          */
    uint32_t wideNodes;      /* four-wide nodes in use: the FAST kernels' tree is stored compactly, in the order the rays walk it (a   */
                             /* node's children side by side, subtree after subtree)                                                  */
    uint32_t leafOrder;      /* 1: the FAST kernels read a copy of the triangles' hot records in the order of that tree's leaves      */
} LrlSceneInfo;
int lrl_scene_info(const LrlContext* ctx, LrlSceneInfo* out);
typedef struct LrlRenderInfo {       /* of the last wavefront-pipeline render */
    uint32_t iterations;     /* trace + shade rounds, summed over the passes                                               */
    uint32_t passes;         /* generate ... accumulate passes (batches of samples)                                        */
    uint32_t batch;          /* samples per full pass                                                                      */
    uint32_t compactFrom;    /* first segment traced from compacted lists (0: none)                                        */
    uint32_t groupLanes;     /* lanes per handed-over ray actually used (64: a wave per ray)                               */
    uint32_t frontierEntries;/* frontier entries per handed-over ray (256 or 512)                                          */
    uint32_t contSlots, contStride;
    uint32_t coopFetch;      /* 1: the trace kernels fetched records with four lanes to a record                            */
    uint32_t streamWords;    /* words per wave of the lean kernel with lane refill (0: one-word waves)                      */
    uint32_t phasedTrips;    /* 1: the lean kernel ran with triangle trips                                                  */
    uint32_t handoverTrips[8]; /* thresholds of the first eight segments as the last pass used them (0: no hand-over)      */
    uint32_t trialState;     /* the timed trial of triangle trips for this scene and frame shape: 0 none (pinned by phasedTrips,   */
                             /* deterministic, or not a FAST render), 1 under way (frames are still being timed each way: a        */
                             /* benchmark warms up until this changes), 2 decided                                                   */
} LrlRenderInfo;
int lrl_last_render_info(const LrlContext* ctx, LrlRenderInfo* out);

/* device memory held for the scene (bytes), by class: nodes (binary records + four-wide nodes), triangle hot,
 * triangle cold, other */
int lrl_scene_device_bytes(LrlContext* ctx, uint64_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* LRL_TRACE_H */
