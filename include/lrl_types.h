/*
 * lrl_types.h — plain-C mirrors of the GPU-visible data layouts of the LRLVEC/RayTracing
 * BVH / Pool tracers.  These ARE the reference's layouts (std430 / std140 as declared in the
 * reference's GLSL); callers keep building their scene in these records and hand pointers to
 * the C ABI in lrl_trace.h.  Any repacking for the MI355X kernels happens behind the ABI.
 *
 * Reference declarations (all in /root/reference/BVH/shaders/RendererFragment0.cpp):
 *   Color :17-29   Bound :30-36   BVHNode :37-45   Plane :54-58   TriangleGPU :59-69
 *   Sphere :70-76  Circle :77-84  Cylinder :85-93  Cone :94-102   PointLight :103-107
 *   UBO Size :118-121   UBO Trans :122-127   UBO GeometryNum :130-139   SSBO DecayOrigin :169-173
 * Triangle (origin form): BVH/shaders/TrianglePreCompute0.cpp:17-26
 *   (Pool adds `bool changed` in the padding at offset 72: Pool/shaders/WaterPositionSyncCompute0.cpp:24-34)
 */
#ifndef LRL_TYPES_H
#define LRL_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct LrlColor {
    float   r[3];  int32_t texR;          /* mirror reflectance, texture layer or -1            */
    float   t[3];  int32_t texT;          /* transmittance                                      */
    float   d[3];  int32_t texD;          /* diffuse; texD == -2 -> 20x20 checker (BVH only)    */
    float   g[3];  int32_t texG;          /* glow (emission); texG == -2 -> checker (BVH only)  */
    float   decayFactor[3];               /* Beer-Lambert absorption of the medium behind       */
    float   n;                            /* refractive index of the medium behind              */
} LrlColor;                               /* 80 B */

typedef struct LrlBound {
    float   min[3]; int32_t leftChild;    /* leftChild: 0 = leaf, 1 = child at index+1          */
    float   max[3]; int32_t rightChild;   /* absolute index, 0 = none                           */
} LrlBound;                               /* 32 B */

typedef struct LrlBVHNode {
    LrlBound bound;
    uint32_t father;
    uint32_t axis;
    uint32_t geometry;                    /* 0 none, 2 triangle, 3 sphere, 4 circle, 5 cylinder, 6 cone */
    uint32_t geometryNum;                 /* index inside that primitive array                  */
    float    blank[4];
} LrlBVHNode;                             /* 64 B */

typedef struct LrlPlane {
    float    plane[4];
    LrlColor color;
} LrlPlane;                               /* 96 B — dead in the reference shader (:230-241) */

typedef struct LrlTriangle {              /* "origin" form, input of TrianglePre */
    float    p1[3]; float _pad0;
    float    p2[3]; float _pad1;
    float    p3[3]; float _pad2;
    float    uv1[2], uv2[2], uv3[2];
    int32_t  changed;                     /* Pool only; padding otherwise */
    int32_t  _pad3;
    LrlColor color;
} LrlTriangle;                            /* 160 B */

typedef struct LrlTriangleGPU {           /* output of TrianglePre, read by the tracer */
    float    plane[4];
    float    p1[3]; float _pad0;
    float    k1[3]; float _pad1;
    float    k2[3]; float _pad2;
    float    uv1[2], uv2[2], uv3[2];
    float    _pad3[2];
    LrlColor color;
} LrlTriangleGPU;                         /* 176 B */

typedef struct LrlSphere {
    float    sphere[4];                   /* centre, R^2 */
    float    e1[3]; float _pad0;          /* pole */
    float    e2[3]; float _pad1;          /* x direction */
    LrlColor color;
} LrlSphere;                              /* 128 B */

typedef struct LrlCircle {
    float    plane[4];
    float    sphere[4];                   /* centre, R^2 */
    float    e1[3]; int32_t tex;
    LrlColor color;
} LrlCircle;                              /* 128 B */

typedef struct LrlCylinder {
    float    c[3];  float r2;
    float    n[3];  float l;
    float    e1[3]; float _pad0;
    LrlColor color;
} LrlCylinder;                            /* 128 B */

typedef struct LrlCone {
    float    c[3];  float c2;             /* apex, cos^2(half angle) */
    float    n[3];  float l2;             /* axis, slant length^2    */
    float    e1[3]; float _pad0;
    LrlColor color;
} LrlCone;                                /* 128 B */

typedef struct LrlPointLight {
    float    color[3]; float _pad0;
    float    p[3];     float _pad1;
} LrlPointLight;                          /* 32 B */

typedef struct LrlSize {                  /* UBO 0, std140 */
    uint32_t size[2]; uint32_t _pad[2];
} LrlSize;                                /* 16 B */

typedef struct LrlTrans {                 /* UBO 1, std140 row_major */
    float    trans[3][4];                 /* rows of the 3x3, 12 B used of each 16 */
    float    r0[3];
    float    z0;
} LrlTrans;                               /* 64 B */

typedef struct LrlGeometryNum {           /* UBO 3, std140 */
    uint32_t planeNum, triangleNum, sphereNum, circleNum, cylinderNum, coneNum, pointLightNum;
    uint32_t _pad;
} LrlGeometryNum;                         /* 32 B */

typedef struct LrlDecayOrigin {           /* SSBO 8 */
    float    decayOrigins[8][4];
    float    decayOrigin[4];              /* xyz absorption at the eye, w refractive-index product */
} LrlDecayOrigin;                         /* 144 B */

typedef struct LrlWaterPoint {            /* Pool SSBO 10, Pool/Pool.cpp:84-90 */
    float z, v, a, blank;
} LrlWaterPoint;

typedef struct LrlWaterParams {           /* Pool UBO 4, Pool/Pool.cpp:50-63 */
    float    k, dt;
    int32_t  groupSizeX, groupSizeY, groupNumX, groupNumY;
    float    z0, dx, dzMax;
} LrlWaterParams;

#ifdef __cplusplus
}
static_assert(sizeof(LrlColor) == 80, "Color");
static_assert(sizeof(LrlBound) == 32, "Bound");
static_assert(sizeof(LrlBVHNode) == 64, "BVHNode");
static_assert(sizeof(LrlPlane) == 96, "Plane");
static_assert(sizeof(LrlTriangle) == 160, "Triangle");
static_assert(sizeof(LrlTriangleGPU) == 176, "TriangleGPU");
static_assert(sizeof(LrlSphere) == 128 && sizeof(LrlCircle) == 128, "Sphere/Circle");
static_assert(sizeof(LrlCylinder) == 128 && sizeof(LrlCone) == 128, "Cylinder/Cone");
static_assert(sizeof(LrlPointLight) == 32, "PointLight");
static_assert(sizeof(LrlTrans) == 64 && sizeof(LrlGeometryNum) == 32, "UBOs");
static_assert(sizeof(LrlDecayOrigin) == 144, "DecayOrigin");
#endif

#endif /* LRL_TYPES_H */
