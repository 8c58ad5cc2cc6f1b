// Device-side scene layout and kernel launch interface shared by the C-ABI implementation
// (lrl_trace.hip) and the two kernel translation units (kernels_parity.hip, kernels_fast.hip).
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
//   triHot    4 x float4 per triangle (64 B, one half cache line): plane | p1,k1.x | k1.yz,k2.xy | k2.z
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
    const float4* triHot;
    const LrlTriCold* triCold;
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
    uint32_t sampleBegin, sampleEnd;   // samples rendered by this launch
    uint32_t spp;                      // divisor of the mean
    uint32_t maxDepth;
    uint32_t stackEntries;             // LDS traversal-stack entries per lane: min(BVH depth + 1, LRL_LDS_STACK_ENTRIES)
    unsigned long long* timeline;      // diagnostic (env LRL_TIMELINE): per wave {start, end} in 100 MHz ticks + hardware id, else NULL
};

enum { LRL_CTR_PRIMARY = 0, LRL_CTR_REFLECT, LRL_CTR_REFRACT, LRL_CTR_SHADOW, LRL_CTR_NODES, LRL_CTR_TRI, LRL_CTR_SPH,
       LRL_CTR_OTHER, LRL_CTR_PRIMARY_HITS, LRL_CTR_COUNT };

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

// kernels_parity.hip / kernels_fast.hip
hipError_t lrl_launch_render_parity(const KScene& sc, const KFrame& fr, float4* out, uint2* hit, unsigned long long* counters,
                                    bool pool, hipStream_t stream);
hipError_t lrl_launch_render_fast(const KScene& sc, const KFrame& fr, float4* out, uint2* hit, unsigned long long* counters,
                                  bool pool, hipStream_t stream);
hipError_t lrl_launch_decay_origin(const KScene& sc, const LrlTrans* trans, uint32_t stackEntries, bool pool, LrlDecayOrigin* out,
                                   hipStream_t stream);
hipError_t lrl_launch_triangle_pre(const LrlTriangle* in, const uint32_t* material, uint32_t first, uint32_t count, float4* triHot,
                                   LrlTriCold* triCold, hipStream_t stream);
hipError_t lrl_launch_circle_pre(LrlCircle* circles, uint32_t count, hipStream_t stream);
