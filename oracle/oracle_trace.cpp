// oracle_trace.cpp — TEST INFRASTRUCTURE: CPU restatement of the reference's per-pixel tracer.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
// product path (raytracing_amd/csrc) never does.
//
// What it restates (all in /root/reference):
//   rayAlloctor / getPlaneT / getTriangleUV / triangleTest / judgeHitBox / judgeHit / rayTrace / main
//                                   BVH/shaders/RendererFragment0.cpp:180-746 (Pool copy: Pool/shaders/…, checker absent)
//   TrianglePre                     BVH/shaders/TrianglePreCompute0.cpp:55-80
//   CirclePre                       BVH/shaders/CirclePreCompute0.cpp:35-44
//   DecayOriginPre / DecayOrigin    BVH/shaders/DecayOriginPreCompute0.cpp:181-419 (Pool: :188-190 fixed +z),
//                                   BVH/shaders/DecayOriginCompute0.cpp:12-48
// Arithmetic follows what Mesa llvmpipe 23.2.1 does with that GLSL, established with the probes in
// tools/probes/ (tools/probe_llvmpipe.py): float32 throughout, no fused multiply-add, dot products
// summed from the LAST component to the first, normalize(v) = v * (1 / sqrt(dot(v, v))), IEEE divide
// and sqrt, min/max return the non-NaN operand.  exp / pow / acos are libm here and polynomial
// approximations there (relative error <= 5e-7 for exp/pow, absolute 1.6e-4 for acos — uv only).
// Texture filtering: bilinear for both magnification and minification (the reference sets no
// sampler state; the goldens were rendered with GL_TEXTURE_MIN_FILTER = GL_LINEAR to make that so).
// Parity pin: tests/test_oracle_golden.py compares this file's output with the llvmpipe renders of
// the reference's shader committed under tests/golden/.
// Build: g++ -O2 -ffp-contract=off -fopenmp (oracle/Makefile).
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/lrl_trace.h"
#include "../include/lrl_types.h"

namespace {

struct V3 { float x, y, z; };
inline V3 v3(float x, float y, float z) { return {x, y, z}; }
inline V3 v3(const float* p) { return {p[0], p[1], p[2]}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3 operator/(V3 a, V3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline float dot(V3 a, V3 b) { return (a.z * b.z + a.y * b.y) + a.x * b.x; }            // llvmpipe order
inline float dot4(const float* p, V3 q, float qw) { return ((p[3] * qw + p[2] * q.z) + p[1] * q.y) + p[0] * q.x; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3 normalize(V3 a) { float r = 1.0f / std::sqrt(dot(a, a)); return a * r; }
inline float sgn(float x) { return x > 0 ? 1.0f : (x < 0 ? -1.0f : 0.0f); }
inline float comp(V3 a, unsigned i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
inline bool anyGE(V3 a, float m) { return a.x >= m || a.y >= m || a.z >= m; }

constexpr float kPi = 3.14159265359f;      // #define Pi, RendererFragment0.cpp:3
constexpr float kOffset = 0.005f;          // :4
constexpr float kMinColor = 0.01f;         // :5

struct Ray { V3 p0; V3 n; float t; };

struct Scene {
    LrlGeometryNum num{};
    std::vector<LrlTriangleGPU> tris;
    const LrlSphere* spheres = nullptr;
    std::vector<LrlCircle> circles;
    const LrlCylinder* cylinders = nullptr;
    const LrlCone* cones = nullptr;
    const LrlPointLight* lights = nullptr;
    const LrlBVHNode* bvh = nullptr;
    uint32_t nodeNum = 0;
    uint32_t texW = 0, texH = 0, texL = 0; const uint8_t* tex = nullptr;
    uint32_t cubeSize = 0; const uint8_t* cube[6] = {};
    uint32_t variant = 0, maxDepth = 10;
    float decayOrigin[4] = {0, 0, 0, 1};
    bool diffuse = false;       // include/lrl_trace.h LRL_FLAG_DIFFUSE_BOUNCE (not in the reference's shader)
    uint32_t width = 0;         // of the full frame, for the pixel index of the random stream
};

struct Counters { uint64_t primary = 0, reflect = 0, refract = 0, shadow = 0, nodes = 0, tri = 0, sph = 0, other = 0, primaryHits = 0; };

// ---- TrianglePre (TrianglePreCompute0.cpp:55-80) ----
void trianglePre(const LrlTriangle& o, LrlTriangleGPU& g) {
    std::memset(&g, 0, sizeof(g));
    g.color = o.color;
    V3 p1 = v3(o.p1);
    g.p1[0] = p1.x; g.p1[1] = p1.y; g.p1[2] = p1.z;
    V3 e1 = v3(o.p2) - p1, e2 = v3(o.p3) - p1;
    V3 n = cross(e1, e2);
    float s = dot(n, n);
    n = normalize(n);
    g.plane[0] = n.x; g.plane[1] = n.y; g.plane[2] = n.z; g.plane[3] = -dot(n, p1);
    float d = dot(e1, e2);
    V3 k1 = (dot(e2, e2) * e1 - d * e2) / s;
    V3 k2 = (dot(e1, e1) * e2 - d * e1) / s;
    g.k1[0] = k1.x; g.k1[1] = k1.y; g.k1[2] = k1.z;
    g.k2[0] = k2.x; g.k2[1] = k2.y; g.k2[2] = k2.z;
    for (int i = 0; i < 2; ++i) { g.uv1[i] = o.uv1[i]; g.uv2[i] = o.uv2[i]; g.uv3[i] = o.uv3[i]; }
}
// ---- CirclePre (CirclePreCompute0.cpp:35-44) ----
void circlePre(LrlCircle& c) {
    V3 n = normalize(v3(c.plane));
    c.plane[0] = n.x; c.plane[1] = n.y; c.plane[2] = n.z; c.plane[3] = -dot(n, v3(c.sphere));
    V3 e = normalize(v3(c.e1));
    c.e1[0] = e.x; c.e1[1] = e.y; c.e1[2] = e.z;
}

// ---- texture sampling (GL 4.5 §8.13/8.14 semantics, bilinear, REPEAT, seamless cube) ----
inline V3 texel2D(const Scene& s, int x, int y, int layer) {
    const uint8_t* p = s.tex + ((size_t(layer) * s.texH + y) * s.texW + x) * 3;
    const float k = 1.0f / 255.0f;
    return {p[2] * k, p[1] * k, p[0] * k};
}
V3 texture2DArray(const Scene& s, float u, float v, int layer) {
    if (!s.tex || s.texL == 0) return {0, 0, 0};
    layer = layer < 0 ? 0 : (layer >= int(s.texL) ? int(s.texL) - 1 : layer);
    float fu = u * float(s.texW) - 0.5f, fv = v * float(s.texH) - 0.5f;
    float flu = std::floor(fu), flv = std::floor(fv);
    float wu = fu - flu, wv = fv - flv;
    auto wrap = [](long i, long n) { long m = i % n; return int(m < 0 ? m + n : m); };
    int x0 = wrap(long(flu), s.texW), x1 = wrap(long(flu) + 1, s.texW);
    int y0 = wrap(long(flv), s.texH), y1 = wrap(long(flv) + 1, s.texH);
    V3 t00 = texel2D(s, x0, y0, layer), t10 = texel2D(s, x1, y0, layer), t01 = texel2D(s, x0, y1, layer), t11 = texel2D(s, x1, y1, layer);
    V3 a = t00 + (t10 - t00) * wu, b = t01 + (t11 - t01) * wu;
    return a + (b - a) * wv;
}
inline V3 cubeTexel(const Scene& s, int face, int x, int y) {
    const uint8_t* p = s.cube[face] + (size_t(y) * s.cubeSize + x) * 3;
    const float k = 1.0f / 255.0f;
    return {p[2] * k, p[1] * k, p[0] * k};
}
// direction -> (face, s, t) per the GL cube-map table
inline void cubeFace(V3 r, int& face, float& sc, float& tc) {
    float ax = std::fabs(r.x), ay = std::fabs(r.y), az = std::fabs(r.z);
    float ma, s, t;
    if (ax >= ay && ax >= az) { face = r.x >= 0 ? 0 : 1; ma = ax; s = r.x >= 0 ? -r.z : r.z; t = -r.y; }
    else if (ay >= az)        { face = r.y >= 0 ? 2 : 3; ma = ay; s = r.x; t = r.y >= 0 ? r.z : -r.z; }
    else                      { face = r.z >= 0 ? 4 : 5; ma = az; s = r.z >= 0 ? r.x : -r.x; t = -r.y; }
    float ima = 0.5f / ma;
    sc = s * ima + 0.5f; tc = t * ima + 0.5f;
}
// inverse: (face, s, t in [0,1] possibly outside) -> direction on the (extended) face plane
inline V3 cubeDir(int face, float s, float t) {
    float a = 2 * s - 1, b = 2 * t - 1;
    switch (face) {
        case 0: return {1, -b, -a};
        case 1: return {-1, -b, a};
        case 2: return {a, 1, b};
        case 3: return {a, -1, -b};
        case 4: return {a, -b, 1};
        default: return {-a, -b, -1};
    }
}
V3 textureCube(const Scene& s, V3 dir) {
    if (!s.cubeSize) return {0, 0, 0};
    int face; float sc, tc;
    cubeFace(dir, face, sc, tc);
    int N = int(s.cubeSize);
    float fu = sc * float(N) - 0.5f, fv = tc * float(N) - 0.5f;
    float flu = std::floor(fu), flv = std::floor(fv);
    float wu = fu - flu, wv = fv - flv;
    int x0 = int(flu), y0 = int(flv);
    V3 tap[4]; bool corner[4];
    int nCorner = 0;
    for (int k = 0; k < 4; ++k) {
        int x = x0 + (k & 1), y = y0 + (k >> 1);
        bool ox = x < 0 || x >= N, oy = y < 0 || y >= N;
        corner[k] = ox && oy;
        if (corner[k]) { ++nCorner; continue; }
        if (!ox && !oy) { tap[k] = cubeTexel(s, face, x, y); continue; }
        // seamless: the tap lies across an edge; fetch the texel of the neighbouring face it falls on
        int f2; float s2, t2;
        cubeFace(cubeDir(face, (float(x) + 0.5f) / float(N), (float(y) + 0.5f) / float(N)), f2, s2, t2);
        int xx = int(std::floor(s2 * float(N))), yy = int(std::floor(t2 * float(N)));
        xx = xx < 0 ? 0 : (xx >= N ? N - 1 : xx); yy = yy < 0 ? 0 : (yy >= N ? N - 1 : yy);
        tap[k] = cubeTexel(s, f2, xx, yy);
    }
    if (nCorner) {   // cube corner: the missing texel is the average of the other three
        V3 sum = {0, 0, 0};
        for (int k = 0; k < 4; ++k) if (!corner[k]) sum = sum + tap[k];
        for (int k = 0; k < 4; ++k) if (corner[k]) tap[k] = sum * (1.0f / 3.0f);
    }
    V3 a = tap[0] + (tap[1] - tap[0]) * wu, b = tap[2] + (tap[3] - tap[2]) * wu;
    return a + (b - a) * wv;
}

// ---- judgeHitBox (:198-214) ----
inline bool judgeHitBox(const Ray& ray, const LrlBound& b) {
    V3 mn = v3(b.min), mx = v3(b.max), p = ray.p0;
    if (mn.x <= p.x && mn.y <= p.y && mn.z <= p.z && p.x <= mx.x && p.y <= mx.y && p.z <= mx.z) return true;
    V3 tmin = (mn - p) / ray.n;
    V3 tmax = (mx - p) / ray.n;
    V3 mintt = {std::fmin(tmin.x, tmax.x), std::fmin(tmin.y, tmax.y), std::fmin(tmin.z, tmax.z)};
    tmax = {std::fmax(tmin.x, tmax.x), std::fmax(tmin.y, tmax.y), std::fmax(tmin.z, tmax.z)};
    float maxt = std::fmax(mintt.x, std::fmax(mintt.y, mintt.z));
    if (maxt > 0) {
        float mint = std::fmin(tmax.x, std::fmin(tmax.y, tmax.z));
        return (maxt <= mint) && (ray.t < 0 || maxt <= ray.t);
    }
    return false;
}
inline float getPlaneT(const Ray& ray, const float* plane) { return -dot4(plane, ray.p0, 1.0f) / dot(v3(plane), ray.n); }

// Trail-based traversal bookkeeping exactly as the shader's (bvhStack / bvhState bit masks, :407-410).
// 64-bit masks instead of the shader's 32-bit ones: identical up to depth 32 (where the shader's shift
// is still defined), and trees deeper than that (the median builder never is) still traverse correctly.
struct Trail {
    uint32_t now = 0; uint64_t stack = 0, state = 0; int sp = -1;
    inline void descend(const LrlBVHNode* bvh, const Ray& ray) {
        ++sp; stack |= (1ull << sp);
        if (comp(ray.n, bvh[now].axis) >= 0 || bvh[now].bound.rightChild == 0) { ++now; state &= ~(1ull << sp); }
        else { now = uint32_t(bvh[now].bound.rightChild); state |= (1ull << sp); }
    }
    inline void ascend(const LrlBVHNode* bvh) {   // :611-630
        while (sp >= 0) {
            now = bvh[now].father;
            if (stack & (1ull << sp)) {
                if (state & (1ull << sp)) { stack &= ~(1ull << sp); ++now; break; }
                else if (bvh[now].bound.rightChild != 0) { stack &= ~(1ull << sp); now = uint32_t(bvh[now].bound.rightChild); break; }
            }
            --sp;
        }
    }
};

// ---- judgeHit: any-hit for shadow rays (:215-391) ----
bool judgeHit(const Scene& sc, const Ray& ray, Counters& cn) {
    Trail tr;
    const LrlBVHNode* bvh = sc.bvh;
    for (;;) {
        ++cn.nodes;
        if (judgeHitBox(ray, bvh[tr.now].bound)) {
            if (bvh[tr.now].geometry != 0) {
                uint32_t n = bvh[tr.now].geometryNum;
                switch (bvh[tr.now].geometry) {
                    case 2: {
                        ++cn.tri;
                        const LrlTriangleGPU& T = sc.tris[n];
                        float tt = getPlaneT(ray, T.plane);
                        if (tt > 0 && tt < ray.t) {
                            V3 d = (ray.p0 + ray.n * tt) - v3(T.p1);
                            float u = dot(d, v3(T.k1)), v = dot(d, v3(T.k2));
                            if (u >= 0 && v >= 0 && (u + v <= 1)) return true;
                        }
                        break;
                    }
                    case 3: {
                        ++cn.sph;
                        const LrlSphere& S = sc.spheres[n];
                        V3 d = v3(S.sphere) - ray.p0;
                        V3 c = cross(d, ray.n);
                        float s = S.sphere[3] - dot(c, c);
                        if (s >= 0) {
                            s = std::sqrt(s);
                            float k = dot(d, ray.n);
                            float tt = -1;
                            if (k + s > 0) tt = k + s;
                            if (k - s > 0) tt = k - s;
                            if (tt > 0 && tt < ray.t) return true;
                        }
                        break;
                    }
                    case 4: {
                        ++cn.other;
                        const LrlCircle& C = sc.circles[n];
                        float tt = getPlaneT(ray, C.plane);
                        if (tt > 0 && tt < ray.t) {
                            V3 d = (ray.p0 + ray.n * tt) - v3(C.sphere);
                            if (dot(d, d) <= C.sphere[3]) return true;
                        }
                        break;
                    }
                    case 5: {
                        ++cn.other;
                        const LrlCylinder& C = sc.cylinders[n];
                        V3 cnv = v3(C.n);
                        float nn0 = dot(ray.n, cnv);
                        float cnn02 = 1 - nn0 * nn0;
                        if (cnn02 != 0) {
                            V3 d = ray.p0 - v3(C.c);
                            float nd = dot(cnv, d);
                            V3 j = d - nd * cnv;
                            float n0j = dot(ray.n, j);
                            float k = n0j * n0j + cnn02 * (C.r2 - dot(j, j));
                            if (k > 0) {
                                k = std::sqrt(k);
                                float tt = -1;
                                if (k - n0j > 0) {
                                    tt = (k - n0j) / cnn02;
                                    float v = nd + nn0 * tt;
                                    if (v > C.l || v < 0) tt = -1;
                                }
                                if (k + n0j < 0) {
                                    float ttt = -(k + n0j) / cnn02;
                                    float ut = nd + nn0 * ttt;
                                    if (ut <= C.l && ut >= 0) tt = ttt;
                                }
                                if (tt > 0 && tt < ray.t) return true;
                            }
                        }
                        break;
                    }
                    case 6: {   // the older algebraic form of the cone test (:315-348)
                        ++cn.other;
                        const LrlCone& C = sc.cones[n];
                        V3 cnv = v3(C.n);
                        V3 d = ray.p0 - v3(C.c);
                        float nn0 = dot(ray.n, cnv);
                        float dn0 = dot(d, cnv);
                        float dn = dot(d, ray.n);
                        float d2 = dot(d, d);
                        float a = C.c2 - nn0 * nn0;
                        float b = nn0 * dn0 - dn * C.c2;
                        float c = d2 * C.c2 - dn0 * dn0;
                        float s = b * b - a * c;
                        if (s > 0) {
                            s = std::sqrt(s);
                            float tt = (b + s) / a;
                            if (tt > 0) {
                                float r2 = d2 + tt * tt + 2 * dn * tt;
                                float k = dn0 + nn0 * tt;
                                if (r2 > C.l2 || k < 0) tt = -1;
                            }
                            float ttt = (b - s) / a;
                            if (ttt > 0 && (ttt < tt || (tt < 0))) {
                                float r2t = d2 + ttt * ttt + 2 * dn * ttt;
                                float k = dn0 + nn0 * ttt;
                                if (r2t <= C.l2 && k > 0) tt = ttt;
                            }
                            if (tt > 0 && tt < ray.t) return true;
                        }
                        break;
                    }
                }
            }
            if (bvh[tr.now].bound.leftChild != 0) { tr.descend(bvh, ray); continue; }
        }
        tr.ascend(bvh);
        if (tr.sp < 0) break;
    }
    return false;
}

struct Hit { uint32_t type = 0, idx = 0; V3 n{0, 0, 0}; float u = 0, v = 0; };

// one primitive against the ray: the body of the shader's switch (RendererFragment0.cpp:419-593); Tp = the triangle's GPU form when geometry == 2
inline void testPrimitive(const Scene& sc, uint32_t geometry, uint32_t n, const LrlTriangleGPU* Tp, Ray& ray, Hit& hit, Counters& cn) {
    switch (geometry) {
        case 2: {
            ++cn.tri;
            const LrlTriangleGPU& T = *Tp;
            float tt = getPlaneT(ray, T.plane);
            if (tt > 0 && (tt < ray.t || ray.t < 0)) {
                V3 d = (ray.p0 + ray.n * tt) - v3(T.p1);
                float u = dot(d, v3(T.k1)), v = dot(d, v3(T.k2));
                if (u >= 0 && v >= 0 && (u + v <= 1)) {
                    ray.t = tt;
                    hit.type = geometry; hit.idx = n;
                    hit.n = v3(T.plane);
                    float w = 1 - u - v;
                    hit.u = w * T.uv1[0] + u * T.uv2[0] + v * T.uv3[0];
                    hit.v = w * T.uv1[1] + u * T.uv2[1] + v * T.uv3[1];
                }
            }
            break;
        }
        case 3: {
            ++cn.sph;
            const LrlSphere& S = sc.spheres[n];
            V3 d = v3(S.sphere) - ray.p0;
            V3 c = cross(d, ray.n);
            float s = S.sphere[3] - dot(c, c);
            if (s >= 0) {
                s = std::sqrt(s);
                float k = dot(d, ray.n);
                float tt = -1;
                if (k + s > 0) tt = k + s;
                if (k - s > 0) tt = k - s;
                if (tt > 0 && (tt < ray.t || ray.t < 0)) {
                    ray.t = tt;
                    hit.type = geometry; hit.idx = n;
                    hit.n = ((ray.p0 + ray.t * ray.n) - v3(S.sphere)) / std::sqrt(S.sphere[3]);
                    V3 e1 = v3(S.e1), e2 = v3(S.e2);
                    float ne1 = dot(hit.n, e1);
                    float ac = std::acos(dot(e2, hit.n) / std::sqrt(1 - ne1 * ne1)) / (2 * kPi);
                    float u = dot(hit.n, cross(e1, e2)) >= 0 ? ac : 1 - ac;
                    hit.u = u; hit.v = 1 - std::acos(ne1) / kPi;
                }
            }
            break;
        }
        case 4: {
            ++cn.other;
            const LrlCircle& C = sc.circles[n];
            float tt = getPlaneT(ray, C.plane);
            if (tt > 0 && (tt < ray.t || ray.t < 0)) {
                V3 d = (ray.p0 + ray.n * tt) - v3(C.sphere);
                if (dot(d, d) <= C.sphere[3]) {
                    ray.t = tt;
                    hit.type = geometry; hit.idx = n;
                    hit.n = v3(C.plane);
                    V3 e1 = v3(C.e1);
                    V3 e2 = cross(hit.n, e1);
                    float R = std::sqrt(C.sphere[3]);
                    hit.u = (1 + dot(e1, d) / R) / 2;
                    hit.v = (1 + dot(e2, d) / R) / 2;
                }
            }
            break;
        }
        case 5: {
            ++cn.other;
            const LrlCylinder& C = sc.cylinders[n];
            V3 cnv = v3(C.n);
            float nn0 = dot(ray.n, cnv);
            float cnn02 = 1 - nn0 * nn0;
            if (cnn02 != 0) {
                V3 d = ray.p0 - v3(C.c);
                float nd = dot(cnv, d);
                V3 j = d - nd * cnv;
                float n0j = dot(ray.n, j);
                float k = n0j * n0j + cnn02 * (C.r2 - dot(j, j));
                if (k > 0) {
                    k = std::sqrt(k);
                    float tt = -1, v = 0;
                    if (k - n0j > 0) {
                        tt = (k - n0j) / cnn02;
                        v = nd + nn0 * tt;
                        if (v > C.l || v < 0) tt = -1;
                    }
                    if (k + n0j < 0) {
                        float ttt = -(k + n0j) / cnn02;
                        float ut = nd + nn0 * ttt;
                        if (ut <= C.l && ut >= 0) { tt = ttt; v = ut; }
                    }
                    if (tt > 0 && (tt < ray.t || ray.t < 0)) {
                        ray.t = tt;
                        hit.type = geometry; hit.idx = n;
                        hit.n = normalize((d + ray.n * ray.t) - cnv * v);
                        V3 e1 = v3(C.e1);
                        V3 e2 = cross(cnv, e1);
                        float ac = std::acos(dot(e1, hit.n)) / (2 * kPi);
                        hit.u = dot(hit.n, e2) >= 0 ? ac : 1 - ac;
                        hit.v = v / C.l;
                    }
                }
            }
            break;
        }
        case 6: {
            ++cn.other;
            const LrlCone& C = sc.cones[n];
            V3 cnv = v3(C.n);
            V3 d = ray.p0 - v3(C.c);
            float nn0 = dot(ray.n, cnv);
            float dn = dot(d, cnv);
            V3 j = dn * cnv - C.c2 * d;
            float c = dot(d, j);
            float b = dot(ray.n, j);
            float a = nn0 * nn0 - C.c2;
            float s = b * b - a * c;
            if (s > 0) {
                s = std::sqrt(s);
                float r2 = 0;
                float tt = (s - b) / a;
                float d2 = dot(d, d);
                float dn0 = dot(d, ray.n);
                if (tt > 0) {
                    r2 = d2 + tt * tt + 2 * dn0 * tt;
                    float k = dn + nn0 * tt;
                    if (r2 > C.l2 || k < 0) tt = -1;
                }
                float ttt = (-b - s) / a;
                if (ttt > 0 && (ttt < tt || (tt < 0))) {
                    float r2t = d2 + ttt * ttt + 2 * dn0 * ttt;
                    float k = dn + nn0 * ttt;
                    if (r2t <= C.l2 && k > 0) { tt = ttt; r2 = r2t; }
                }
                if (tt > 0 && (tt < ray.t || ray.t < 0)) {
                    ray.t = tt;
                    hit.type = geometry; hit.idx = n;
                    V3 q = d + ray.n * ray.t;
                    hit.n = (q * std::sqrt(C.c2 / r2) - cnv) / std::sqrt(1 - C.c2);
                    V3 nxy = normalize(q - cnv * std::sqrt(r2 * C.c2));
                    V3 e1 = v3(C.e1);
                    float ac = std::acos(dot(e1, nxy)) / (2 * kPi);
                    hit.u = dot(nxy, cross(cnv, e1)) >= 0 ? ac : 1 - ac;
                    hit.v = 1 - std::sqrt(r2 / C.l2);
                }
            }
            break;
        }
    }
}

// ---- closest-hit traversal (:406-632) ----
void closestHit(const Scene& sc, Ray& ray, Hit& hit, Counters& cn) {
    Trail tr;
    const LrlBVHNode* bvh = sc.bvh;
    for (;;) {
        ++cn.nodes;
        if (judgeHitBox(ray, bvh[tr.now].bound)) {
            uint32_t geometry = bvh[tr.now].geometry;
            if (geometry != 0) {
                uint32_t n = bvh[tr.now].geometryNum;
                testPrimitive(sc, geometry, n, geometry == 2 ? &sc.tris[n] : nullptr, ray, hit, cn);
            }
            if (bvh[tr.now].bound.leftChild != 0) { tr.descend(bvh, ray); continue; }
        }
        tr.ascend(bvh);
        if (tr.sp < 0) break;
    }
}

inline const LrlColor* colorOf(const Scene& sc, const Hit& h) {
    switch (h.type) {
        case 2: return &sc.tris[h.idx].color;
        case 3: return &sc.spheres[h.idx].color;
        case 4: return &sc.circles[h.idx].color;
        case 5: return &sc.cylinders[h.idx].color;
        case 6: return &sc.cones[h.idx].color;
    }
    return nullptr;
}
inline float checker(float u, float v) { return float((uint32_t(20 * u) + uint32_t(20 * v)) % 2u); }

struct StackEntry { V3 p0; V3 n; int depth; V3 ratio; V3 decay; };

// ---- rayTrace (:393-741) ----
// ---- diffuse bounce as include/lrl_trace.h defines it (LRL_FLAG_DIFFUSE_BOUNCE): the library's own extension for
// SURVEY 8d's C5, restated here independently of the device code ----
static void philox4x32_10(const uint32_t ctr[4], uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]}, key[2] = {0u, 0u};
    for (int round = 0; round < 10; ++round) {
        uint64_t a = uint64_t(0xD2511F53u) * c[0], b = uint64_t(0xCD9E8D57u) * c[2];
        uint32_t n[4] = {uint32_t(b >> 32) ^ c[1] ^ key[0], uint32_t(b), uint32_t(a >> 32) ^ c[3] ^ key[1], uint32_t(a)};
        std::memcpy(c, n, sizeof n);
        key[0] += 0x9E3779B9u; key[1] += 0xBB67AE85u;
    }
    std::memcpy(out, c, sizeof c);
}
static float sinQuarterTurn(float x) {      // sin(pi/2 x) on [0, 1], the polynomial the definition names
    float x2 = x * x;
    return x * (1.57079632679f + x2 * (-0.64596409750f + x2 * (0.07969262624f + x2 * (-0.00468175413f + x2 * 0.00016044118f))));
}
static V3 diffuseDirection(V3 N, float cosi1, uint32_t pixel, uint32_t sample, uint32_t depth) {
    uint32_t ctr[4] = {pixel, sample, depth, 0u}, r[4];
    philox4x32_10(ctr, r);
    float u1 = float(r[0] >> 8) * (1.0f / 16777216.0f), u2 = float(r[1] >> 8) * (1.0f / 16777216.0f);
    float q = u2 * 4.0f;
    int quad = int(q);
    float f = q - float(quad);
    float sq = sinQuarterTurn(f), cq = sinQuarterTurn(1.0f - f), c, s;
    switch (quad) {
        case 0: c = cq; s = sq; break;
        case 1: c = -sq; s = cq; break;
        case 2: c = -cq; s = -sq; break;
        default: c = sq; s = -cq; break;
    }
    V3 n = cosi1 > 0 ? V3{-N.x, -N.y, -N.z} : N;
    float sg = n.z >= 0 ? 1.0f : -1.0f;
    float a = -1.0f / (sg + n.z), b = (n.x * n.y) * a;
    V3 T = {1.0f + (sg * (n.x * n.x)) * a, sg * b, -(sg * n.x)};
    V3 B = {b, sg + (n.y * n.y) * a, -n.y};
    float rr = std::sqrt(u1), h = std::sqrt(1.0f - u1);
    return normalize(((rr * c) * T + (rr * s) * B) + h * n);
}

V3 rayTrace(const Scene& sc, Ray ray, Counters& cn, uint32_t* primaryHit, uint32_t rngPixel = 0, uint32_t rngSample = 0) {
    StackEntry stack[LRL_MAX_RAY_DEPTH];
    int sp = -1, depth = 0;
    const int maxDepth = int(sc.maxDepth);
    V3 ratioNow = {1, 1, 1}, answer = {0, 0, 0};
    Hit hit;                                            // explicit zero-init of the shader's uninitialised hitObj/tempN/tempUV
    V3 decayNow = v3(sc.decayOrigin);
    bool first = true;
    ++cn.primary;
    for (;;) {
        ray.t = -1;
        closestHit(sc, ray, hit, cn);
        if (first) {
            first = false;
            if (ray.t >= 0) ++cn.primaryHits;
            if (primaryHit) { primaryHit[0] = ray.t < 0 ? 0 : hit.type; primaryHit[1] = ray.t < 0 ? 0 : hit.idx; }
        }
        LrlColor col;
        std::memset(&col, 0, sizeof(col));
        col.texG = -1;
        if (hit.type != 0) col = *colorOf(sc, hit);
        V3 g = v3(col.g), r = v3(col.r), t = v3(col.t), dcol = v3(col.d);
        float nIdx = col.n;
        if (col.texG >= 0) g = g * texture2DArray(sc, hit.u, hit.v, col.texG);
        if (sc.variant == LRL_VARIANT_BVH && col.texG == -2) g = g * checker(hit.u, hit.v);
        if (ray.t < 0) { g = textureCube(sc, ray.n); ray.t = 0; }
        V3 e = decayNow * ray.t;
        ratioNow = ratioNow * v3(std::exp(e.x), std::exp(e.y), std::exp(e.z));
        answer = answer + g * ratioNow;
        if (ray.t > 0 && depth < maxDepth) {
            if (col.texT >= 0) t = t * texture2DArray(sc, hit.u, hit.v, col.texT);
            t = t * ratioNow;
            float cosi1 = dot(ray.n, hit.n);
            if (anyGE(t, kMinColor)) {
                if (cosi1 > 0) nIdx = 1 / nIdx;
                float sini1 = std::sqrt(1 - cosi1 * cosi1);
                float sini2 = sini1 / nIdx;
                if (sini2 < 1) {
                    float cosi2 = std::sqrt(1 - sini2 * sini2);
                    if (sini2 <= 0.01f) {
                        float nadd1 = 1 / (nIdx + 1);
                        float q = (nIdx - 1) * nadd1;
                        r = r * (q * q);
                        float q2 = 2 * nadd1;
                        t = t * ((q2 * q2) * nIdx);
                    } else {
                        float ac = std::fabs(cosi1);
                        float a1 = nIdx * ac + cosi2;
                        float a2 = ac + nIdx * cosi2;
                        float f1 = (cosi2 - nIdx * ac) / a1, f2 = (nIdx * cosi2 - ac) / a2;
                        r = r * ((f1 * f1 + f2 * f2) / 2);
                        t = t * (2 * cosi2 * (1 / (a1 * a1) + 1 / (a2 * a2)) * nIdx * ac);
                    }
                    if (anyGE(t, kMinColor)) {
                        StackEntry& s = stack[++sp];
                        s.decay = decayNow - sgn(cosi1) * v3(col.decayFactor);
                        s.p0 = ray.p0 + (ray.t + kOffset) * ray.n;
                        s.n = (ray.n + (nIdx * sgn(cosi1) * cosi2 - cosi1) * hit.n) / nIdx;
                        s.ratio = t;
                        s.depth = depth + 1;
                    }
                } else {
                    r = {1, 1, 1};
                }
            }
            ray.p0 = ray.p0 + (ray.t - kOffset) * ray.n;
            for (uint32_t n = 0; n < sc.num.pointLightNum; ++n) {
                V3 dn = v3(sc.lights[n].p) - ray.p0;
                float tt = dot(dn, dn);
                dn = normalize(dn);
                Ray sh{ray.p0, dn, std::sqrt(tt)};
                ++cn.shadow;
                if (!judgeHit(sc, sh, cn)) {
                    if (col.texD >= 0) dcol = dcol * texture2DArray(sc, hit.u, hit.v, col.texD);
                    if (sc.variant == LRL_VARIANT_BVH && col.texD == -2) dcol = dcol * checker(hit.u, hit.v);
                    float k = std::fmax(-20 * sgn(dot(ray.n, hit.n)) * dot(hit.n, dn) / tt, 0.0f);
                    answer = answer + ((k * v3(sc.lights[n].color)) * dcol) * ratioNow;
                }
            }
            if (col.texR >= 0) r = r * texture2DArray(sc, hit.u, hit.v, col.texR);
            ratioNow = ratioNow * r;
            if (anyGE(ratioNow, kMinColor)) {
                if (sc.diffuse) ray.n = diffuseDirection(hit.n, cosi1, rngPixel, rngSample, uint32_t(depth));
                else ray.n = ray.n - (2 * cosi1) * hit.n;
                ++depth;
                ++cn.reflect;
                continue;
            }
        }
        if (sp < 0) return answer / sc.decayOrigin[3];
        depth = stack[sp].depth;
        ray.p0 = stack[sp].p0;
        ray.n = stack[sp].n;
        ratioNow = stack[sp].ratio;
        decayNow = stack[sp].decay;
        --sp;
        ++cn.refract;
    }
}

void buildScene(const LrlSceneView* sv, Scene& sc) {
    sc.num = sv->num;
    sc.tris.resize(sv->num.triangleNum);
    if (sv->trianglesGPU) std::memcpy(sc.tris.data(), sv->trianglesGPU, sizeof(LrlTriangleGPU) * sv->num.triangleNum);
    else for (uint32_t i = 0; i < sv->num.triangleNum; ++i) trianglePre(sv->trianglesOrigin[i], sc.tris[i]);
    sc.circles.assign(sv->circles, sv->circles + sv->num.circleNum);
    for (auto& c : sc.circles) circlePre(c);
    sc.spheres = sv->spheres; sc.cylinders = sv->cylinders; sc.cones = sv->cones; sc.lights = sv->pointLights;
    sc.bvh = sv->bvh; sc.nodeNum = sv->bvhNodeNum;
    sc.texW = sv->texWidth; sc.texH = sv->texHeight; sc.texL = sv->texLayers; sc.tex = sv->texBGR;
    sc.cubeSize = sv->cubeSize;
    for (int i = 0; i < 6; ++i) sc.cube[i] = sv->cubeBGR[i];
}

inline Ray primaryRay(const LrlTrans& tr, uint32_t W, uint32_t H, uint32_t px, uint32_t py) {
    // rayAlloctor (:180-183): gl_FragCoord = pixel centre
    float fx = 2 * (float(px) + 0.5f) - float(W), fy = 2 * (float(py) + 0.5f) - float(H);
    V3 v = {fx, fy, tr.z0};
    // mat3 * vec3 is summed first component first on llvmpipe (tools/probes/matvec.comp), unlike dot()
    auto row = [&](int i) { return (tr.trans[i][0] * v.x + tr.trans[i][1] * v.y) + tr.trans[i][2] * v.z; };
    V3 d = {row(0), row(1), row(2)};
    return Ray{v3(tr.r0), normalize(d), -1};
}

}  // namespace

extern "C" {

struct OracleParams {
    uint32_t width, height, maxDepth, variant, nSamples, rowBegin, rowEnd, threads;
    float decayOrigin[4];
    uint32_t diffuse;        // LRL_FLAG_DIFFUSE_BOUNCE semantics
};

// the two pieces of the diffuse bounce, for known-answer tests
void oracle_philox4x32_10(const uint32_t ctr[4], uint32_t out[4]) { philox4x32_10(ctr, out); }
void oracle_diffuse_direction(const float N[3], float cosi1, uint32_t pixel, uint32_t sample, uint32_t depth, float out[3]) {
    V3 d = diffuseDirection(V3{N[0], N[1], N[2]}, cosi1, pixel, sample, depth);
    out[0] = d.x; out[1] = d.y; out[2] = d.z;
}

int oracle_triangle_pre(const LrlTriangle* in, uint32_t n, LrlTriangleGPU* out) {
    for (uint32_t i = 0; i < n; ++i) trianglePre(in[i], out[i]);
    return 0;
}
int oracle_circle_pre(LrlCircle* c, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i) circlePre(c[i]);
    return 0;
}

// image = (sum over samples, in order) / nSamples; hit = primary hit {type, idx} of sample 0
int oracle_render(const LrlSceneView* sv, const LrlTrans* trans, const OracleParams* p, float* rgba, uint32_t* hit, LrlCounters* counters) {
    if (!sv || !trans || !p || !rgba || p->maxDepth < 1 || p->maxDepth > LRL_MAX_RAY_DEPTH || !p->nSamples) return 1;
    Scene sc;
    buildScene(sv, sc);
    sc.variant = p->variant; sc.maxDepth = p->maxDepth;
    std::memcpy(sc.decayOrigin, p->decayOrigin, 16);
    sc.diffuse = p->diffuse != 0; sc.width = p->width;
    uint32_t r0 = p->rowBegin, r1 = p->rowEnd ? p->rowEnd : p->height;
    Counters total;
    int threads = p->threads ? int(p->threads) : 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
    {
        Counters cn;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
        for (int64_t y = r0; y < int64_t(r1); ++y) {
            for (uint32_t x = 0; x < p->width; ++x) {
                V3 acc = {0, 0, 0};
                size_t o = (size_t(y - r0) * p->width + x);
                for (uint32_t s = 0; s < p->nSamples; ++s) {
                    Ray ray = primaryRay(trans[s], p->width, p->height, x, uint32_t(y));
                    V3 c = rayTrace(sc, ray, cn, (hit && s == 0) ? hit + o * 2 : nullptr, uint32_t(y) * p->width + x, s);
                    acc = p->nSamples == 1 ? c : acc + c;
                }
                if (p->nSamples > 1) acc = acc / float(p->nSamples);
                rgba[o * 4 + 0] = acc.x; rgba[o * 4 + 1] = acc.y; rgba[o * 4 + 2] = acc.z; rgba[o * 4 + 3] = 1.0f;
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        {
            total.primary += cn.primary; total.reflect += cn.reflect; total.refract += cn.refract; total.shadow += cn.shadow;
            total.nodes += cn.nodes; total.tri += cn.tri; total.sph += cn.sph; total.other += cn.other; total.primaryHits += cn.primaryHits;
        }
    }
    (void)threads;
    if (counters) {
        counters->raysPrimary = total.primary; counters->raysReflect = total.reflect; counters->raysRefract = total.refract;
        counters->raysShadow = total.shadow; counters->nodeVisits = total.nodes; counters->triTests = total.tri;
        counters->sphereTests = total.sph; counters->otherTests = total.other; counters->primaryHits = total.primaryHits;
        counters->recordFetches = 0; counters->laneTrips = 0; counters->waveTrips = 0; counters->groupLanes = 0; counters->groupRounds = 0;
    }
    return 0;
}

// Tree-free closest hit of the PRIMARY rays of `n` pixels {x, y}: every primitive of the scene is tested against every ray, with the
// same test arithmetic as closestHit above (testPrimitive; triangles through TrianglePre computed on the fly, so a 100 M-triangle
// scene needs no second copy and no BVH).  It is what checks a scene too large for a host-built tree: the answer does
// not depend on any tree, node index or byte offset.  Among primitives at exactly the same distance the lowest index wins (the
// reference takes the first one its traversal meets): callers judge such pixels as floating-point ties.
// hits: {type, idx} per pixel (0, 0 = miss); tOut: the hit distance (-1 = miss).
int oracle_bruteforce_primary(const LrlSceneView* sv, const LrlTrans* trans, uint32_t width, uint32_t height, uint32_t n, const uint32_t* pixels,
                              uint32_t threads, uint32_t* hits, float* tOut) {
    if (!sv || !trans || !pixels || !hits || !tOut || !n || (sv->num.triangleNum && !sv->trianglesOrigin && !sv->trianglesGPU)) return 1;
    std::vector<Ray> rays(n);
    for (uint32_t r = 0; r < n; ++r) rays[r] = primaryRay(trans[0], width, height, pixels[2 * r], pixels[2 * r + 1]);
    // direction components as arrays: the plane distances of one triangle against all rays are one vectorisable loop
    std::vector<float> nx(n), ny(n), nz(n);
    for (uint32_t r = 0; r < n; ++r) { nx[r] = rays[r].n.x; ny[r] = rays[r].n.y; nz[r] = rays[r].n.z; }
    const V3 p0 = rays[0].p0;      // every primary ray starts at the eye
    const uint32_t T = sv->num.triangleNum;
    const int nthreads = threads ? int(threads) : 1;
    std::vector<float> bestT(size_t(nthreads) * n, -1.0f);
    std::vector<uint32_t> bestI(size_t(nthreads) * n, 0xffffffffu);
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    {
#ifdef _OPENMP
        const int me = omp_get_thread_num();
#else
        const int me = 0;
#endif
        float* bt = &bestT[size_t(me) * n];
        uint32_t* bi = &bestI[size_t(me) * n];
        std::vector<float> tt(n);
        std::vector<int32_t> okv(n);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int64_t i = 0; i < int64_t(T); ++i) {      // (static schedule, ascending within a thread: the lowest index wins a tie)
            LrlTriangleGPU G;
            const LrlTriangleGPU* g = sv->trianglesGPU ? sv->trianglesGPU + i : &G;
            if (!sv->trianglesGPU) trianglePre(sv->trianglesOrigin[i], G);
            const float num = -dot4(g->plane, p0, 1.0f);                       // getPlaneT (:184-187), its numerator the same for every ray
            const float px = g->plane[0], py = g->plane[1], pz = g->plane[2];
            const float ax = g->p1[0], ay = g->p1[1], az = g->p1[2];
            const float k1x = g->k1[0], k1y = g->k1[1], k1z = g->k1[2], k2x = g->k2[0], k2y = g->k2[1], k2z = g->k2[2];
            float* t = tt.data();
            int32_t* ok = okv.data();
            const float *NX = nx.data(), *NY = ny.data(), *NZ = nz.data();
            // the triangle test of closestHit (:433-448) for all rays at once, element by element in the shader's operation order
            // (dot(): last component first; no contraction): plane distance, point, affine coordinates, inside test
#ifdef _OPENMP
#pragma omp simd
#endif
            for (uint32_t r = 0; r < n; ++r) {
                const float q = num / ((pz * NZ[r] + py * NY[r]) + px * NX[r]);
                const float dx = (p0.x + NX[r] * q) - ax, dy = (p0.y + NY[r] * q) - ay, dz = (p0.z + NZ[r] * q) - az;
                const float u = (dz * k1z + dy * k1y) + dx * k1x, v = (dz * k2z + dy * k2y) + dx * k2x;
                t[r] = q;
                ok[r] = (q > 0) & (u >= 0) & (v >= 0) & (u + v <= 1);
            }
            for (uint32_t r = 0; r < n; ++r)
                if (ok[r] && (t[r] < bt[r] || bt[r] < 0)) { bt[r] = t[r]; bi[r] = uint32_t(i); }
        }
    }
    Scene sc;      // the few primitives that are no triangles, through the shader's own tests
    sc.num = sv->num;
    sc.circles.assign(sv->circles, sv->circles + sv->num.circleNum);
    for (auto& c : sc.circles) circlePre(c);
    sc.spheres = sv->spheres; sc.cylinders = sv->cylinders; sc.cones = sv->cones;
    const uint32_t counts[7] = {0, 0, 0, sv->num.sphereNum, sv->num.circleNum, sv->num.cylinderNum, sv->num.coneNum};
    Counters cn;
    for (uint32_t r = 0; r < n; ++r) {
        float t = -1.0f;
        uint32_t idx = 0xffffffffu;
        for (int th = 0; th < nthreads; ++th) {
            const float q = bestT[size_t(th) * n + r];
            const uint32_t qi = bestI[size_t(th) * n + r];
            if (q > 0 && (t < 0 || q < t || (q == t && qi < idx))) { t = q; idx = qi; }
        }
        Ray ray = rays[r];
        ray.t = t;
        Hit hit;
        if (t > 0) { hit.type = 2; hit.idx = idx; }
        for (uint32_t geometry = 3; geometry <= 6; ++geometry)
            for (uint32_t k = 0; k < counts[geometry]; ++k) testPrimitive(sc, geometry, k, nullptr, ray, hit, cn);
        hits[2 * r] = hit.type; hits[2 * r + 1] = hit.type ? hit.idx : 0u;
        tOut[r] = ray.t;
    }
    return 0;
}

// For judging id mismatches as floating-point ties (north_star: "mismatches allowed only at floating-point tie edges"): for each of n
// {pixel, primitive} pairs, where the pixel's primary ray meets the primitive's carrier surface and how far that point is from the
// primitive's own patch.  Triangles and discs in double precision from the records: tOut = distance along the ray to the plane (-1:
// behind the eye or parallel), sdOut = signed in-plane distance of that point to the patch's boundary (negative inside), tolOut = the
// world-space rounding band of a float32 hit point there: 8 ulp of the largest coordinate involved, over the cosine of incidence (a
// plane distance at grazing incidence is that much less certain).  Spheres, cylinders, cones: the shader's own float test; sd = -inf on
// a hit, +inf on a miss.  Type 0 (a miss): t = -1, sd = +inf.
int oracle_tie_probe(const LrlSceneView* sv, const LrlTrans* trans, uint32_t width, uint32_t height, uint32_t n, const uint32_t* pixels,
                     const uint32_t* prims, float* tOut, float* sdOut, float* tolOut) {
    if (!sv || !trans || !pixels || !prims || !tOut || !sdOut || !tolOut) return 1;
    Scene sc;
    sc.num = sv->num;
    sc.circles.assign(sv->circles, sv->circles + sv->num.circleNum);
    for (auto& c : sc.circles) circlePre(c);
    sc.spheres = sv->spheres; sc.cylinders = sv->cylinders; sc.cones = sv->cones;
    const uint32_t counts[7] = {0, 0, sv->num.triangleNum, sv->num.sphereNum, sv->num.circleNum, sv->num.cylinderNum, sv->num.coneNum};
    const double eps = 8.0 * 1.1920928955078125e-7;
    Counters cn;
    for (uint32_t i = 0; i < n; ++i) {
        const Ray ray = primaryRay(trans[0], width, height, pixels[2 * i], pixels[2 * i + 1]);
        const uint32_t type = prims[2 * i], idx = prims[2 * i + 1];
        tOut[i] = -1.0f; sdOut[i] = INFINITY; tolOut[i] = 0.0f;
        if (type < 2 || type > 6 || idx >= counts[type]) continue;
        const double o[3] = {ray.p0.x, ray.p0.y, ray.p0.z}, d[3] = {ray.n.x, ray.n.y, ray.n.z};
        auto band = [&](double t, double cosi) {
            double scale = t;
            for (int a = 0; a < 3; ++a) scale = std::max(scale, std::max(std::fabs(o[a]), std::fabs(o[a] + d[a] * t)));
            return float(eps * scale / std::max(std::fabs(cosi), 1e-3));
        };
        if (type == 2) {
            if (!sv->trianglesOrigin) continue;
            const LrlTriangle& T = sv->trianglesOrigin[idx];
            const double p[3][3] = {{T.p1[0], T.p1[1], T.p1[2]}, {T.p2[0], T.p2[1], T.p2[2]}, {T.p3[0], T.p3[1], T.p3[2]}};
            double e1[3], e2[3], N[3];
            for (int a = 0; a < 3; ++a) { e1[a] = p[1][a] - p[0][a]; e2[a] = p[2][a] - p[0][a]; }
            N[0] = e1[1] * e2[2] - e1[2] * e2[1]; N[1] = e1[2] * e2[0] - e1[0] * e2[2]; N[2] = e1[0] * e2[1] - e1[1] * e2[0];
            const double nl = std::sqrt(N[0] * N[0] + N[1] * N[1] + N[2] * N[2]);
            if (!(nl > 0)) continue;
            for (int a = 0; a < 3; ++a) N[a] /= nl;
            const double den = N[0] * d[0] + N[1] * d[1] + N[2] * d[2];
            if (den == 0) continue;
            const double t = (N[0] * (p[0][0] - o[0]) + N[1] * (p[0][1] - o[1]) + N[2] * (p[0][2] - o[2])) / den;
            if (!(t > 0)) continue;
            const double P[3] = {o[0] + d[0] * t, o[1] + d[1] * t, o[2] + d[2] * t};
            // signed distance to the boundary in the plane: inside = on the inner side of all three edges
            bool inside = true;
            double nearestOut = INFINITY, nearestIn = INFINITY;
            for (int k = 0; k < 3; ++k) {
                const double* A = p[k];
                const double* B = p[(k + 1) % 3];
                double ab[3], ap[3];
                for (int a = 0; a < 3; ++a) { ab[a] = B[a] - A[a]; ap[a] = P[a] - A[a]; }
                // side: (ab x ap) . N >= 0 inside (counter-clockwise about N)
                const double cx = ab[1] * ap[2] - ab[2] * ap[1], cy = ab[2] * ap[0] - ab[0] * ap[2], cz = ab[0] * ap[1] - ab[1] * ap[0];
                const double side = cx * N[0] + cy * N[1] + cz * N[2];
                const double l2 = ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2];
                if (!(l2 > 0)) continue;
                if (side < 0) inside = false;
                nearestIn = std::min(nearestIn, std::fabs(side) / std::sqrt(l2));                  // distance to the edge's line
                const double s = std::min(1.0, std::max(0.0, (ab[0] * ap[0] + ab[1] * ap[1] + ab[2] * ap[2]) / l2));
                double q2 = 0;
                for (int a = 0; a < 3; ++a) { const double w = ap[a] - ab[a] * s; q2 += w * w; }
                nearestOut = std::min(nearestOut, std::sqrt(q2));                                  // distance to the edge segment
            }
            tOut[i] = float(t); sdOut[i] = float(inside ? -nearestIn : nearestOut); tolOut[i] = band(t, den);
        } else if (type == 4) {
            const LrlCircle& C = sc.circles[idx];
            const double den = double(C.plane[0]) * d[0] + double(C.plane[1]) * d[1] + double(C.plane[2]) * d[2];
            if (den == 0) continue;
            const double t = -(double(C.plane[0]) * o[0] + double(C.plane[1]) * o[1] + double(C.plane[2]) * o[2] + double(C.plane[3])) / den;
            if (!(t > 0)) continue;
            double r2 = 0;
            for (int a = 0; a < 3; ++a) { const double w = o[a] + d[a] * t - double(C.sphere[a]); r2 += w * w; }
            tOut[i] = float(t); sdOut[i] = float(std::sqrt(r2) - std::sqrt(double(C.sphere[3]))); tolOut[i] = band(t, den);
        } else {
            Ray r = ray;
            Hit h;
            testPrimitive(sc, type, idx, nullptr, r, h, cn);
            if (h.type) { tOut[i] = r.t; sdOut[i] = -INFINITY; tolOut[i] = band(double(r.t), 1.0); }
        }
    }
    return 0;
}

// DecayOriginPre + DecayOrigin: which medium is the eye in (8 probe rays, mode vote).
int oracle_decay_origin(const LrlSceneView* sv, const LrlTrans* trans, uint32_t variant, LrlDecayOrigin* out) {
    Scene sc;
    buildScene(sv, sc);
    const float off = 0.001f;    // #define offset, DecayOriginPreCompute0.cpp:3
    auto rnd = [](float a, float b) {   // random(), :175-178
        float d = (b + 43758.5453123f) * 78.233f + (a + 43758.5453123f) * 12.9898f;   // dot, last component first
        float s = std::sin(d) * 43758.5453123f;
        return 2 * (s - std::floor(s)) - 1;
    };
    Counters cn;
    for (uint32_t i = 0; i < 8; ++i) {
        Ray ray;
        ray.p0 = v3(trans->r0);
        if (variant == LRL_VARIANT_POOL) ray.n = normalize(v3(0, 0, 1));   // Pool/shaders/DecayOriginPreCompute0.cpp:188-190
        else {
            float b0 = float(i & 1u), b1 = float((i & 2u) / 2u), b2 = float((i & 4u) / 4u);
            ray.n = normalize(v3(rnd(trans->r0[0] + b0, trans->r0[1] + b0), rnd(trans->r0[1] + b1, trans->r0[2] + b1),
                                 rnd(trans->r0[2] + b2, trans->r0[0] + b2)));
        }
        V3 decayTemp = {0, 0, 0};
        float nTemp = 1;
        for (int guard = 0; guard < 4096; ++guard) {
            ray.t = -1;
            Hit hit;
            closestHit(sc, ray, hit, cn);
            if (ray.t < 0) break;
            // the probe shader recomputes the surface normal per primitive (:236,257,273,316,361); only its
            // sign against the ray matters, and for cones it uses its own axis-scaled form
            V3 nrm = hit.n;
            if (hit.type == 6) {
                const LrlCone& C = sc.cones[hit.idx];
                V3 d = ray.p0 - v3(C.c);
                V3 q = d + ray.n * ray.t;
                float r2 = dot(q, q);
                nrm = normalize(q - v3(C.n) * std::sqrt(r2 / C.c2));
            }
            float dn = dot(ray.n, nrm);
            const LrlColor* col = colorOf(sc, hit);
            decayTemp = decayTemp + v3(col->decayFactor) * sgn(dn);
            nTemp *= dn > 0 ? col->n : 1 / col->n;
            ray.p0 = ray.p0 + ray.n * (ray.t + off);
        }
        out->decayOrigins[i][0] = decayTemp.x; out->decayOrigins[i][1] = decayTemp.y; out->decayOrigins[i][2] = decayTemp.z;
        out->decayOrigins[i][3] = nTemp;
    }
    // DecayOrigin: most frequent exactly-equal vec4, first wins ties (DecayOriginCompute0.cpp:14-47)
    bool judged[8] = {}; int num[8] = {};
    for (int a = 0; a < 8; ++a) {
        if (judged[a]) continue;
        judged[a] = true; num[a]++;
        for (int b = 0; b < 8; ++b)
            if (!judged[b] && out->decayOrigins[b][0] == out->decayOrigins[a][0] && out->decayOrigins[b][1] == out->decayOrigins[a][1] &&
                out->decayOrigins[b][2] == out->decayOrigins[a][2] && out->decayOrigins[b][3] == out->decayOrigins[a][3]) { num[a]++; judged[b] = true; }
    }
    int index = 0, mx = 0;
    for (int a = 0; a < 8; ++a) if (num[a] > mx) { mx = num[a]; index = a; }
    std::memcpy(out->decayOrigin, out->decayOrigins[index], 16);
    return 0;
}


// ---- Pool water simulation (next row f2) ----
// One substep = WaterAccelerationCompute0.cpp:22-41 then WaterPositionCompute0.cpp:49-55, in the shaders' expression order
// (this file is compiled without contraction).  The acceleration pass reads z of its neighbours and writes only a, the
// position pass is point-local, so sweeping the grid in place is the same as the two dispatches.  gx = gl_NumWorkGroups.x * 8
// (row length), gy = gl_NumWorkGroups.y * 8; the shader strides rows by gx in `index` but reaches its y neighbours with
// gy (`offset`, :33): kept as it is (the reference only uses square grids).
struct OracleWaterPoint { float z, v, a, blank; };
void oracle_water_run(OracleWaterPoint* w, uint32_t gx, uint32_t gy, float k, float dt, uint32_t substeps) {
    for (uint32_t s = 0; s < substeps; ++s) {
        for (uint32_t y = 0; y < gy; ++y)
            for (uint32_t x = 0; x < gx; ++x) {
                float dv = 0;
                const uint32_t index = x + gx * y;
                if (x != 0 && x != gx - 1) dv += 2 * w[index].z - w[index - 1].z - w[index + 1].z;
                else if (x != 0) dv += w[index].z - w[index - 1].z;
                else dv += w[index].z - w[index + 1].z;
                const uint32_t offset = gy;
                if (y != 0 && y != offset - 1) dv += 2 * w[index].z - w[index - offset].z - w[index + offset].z;
                else if (y != 0) dv += w[index].z - w[index - offset].z;
                else dv += w[index].z - w[index + offset].z;
                w[index].a = k * dt * dv;
            }
        for (uint32_t i = 0; i < gx * gy; ++i) {
            w[i].z -= dt * (w[i].v + w[i].a / 2);
            w[i].v += w[i].a;
            w[i].v *= 0.999985f;
        }
    }
}
// WaterPositionSyncCompute0.cpp:49-75: the height of point (x, y) goes into the up to six triangle corners that share it
void oracle_water_sync(const OracleWaterPoint* w, uint32_t gx, uint32_t gy, LrlTriangle* tris) {
    for (uint32_t y = 0; y < gy; ++y)
        for (uint32_t x = 0; x < gx; ++x) {
            const float z = w[x + gx * y].z;
            const uint32_t t = x + y * 2 * (gx - 1);
            if (x != 0) {
                if (y != 0) tris[t - gx].p1[2] = z;
                if (y != gy - 1) { tris[t - 1].p2[2] = z; tris[t + gx - 2].p3[2] = z; }
            }
            if (x != gx - 1) {
                if (y != 0) { tris[t - 2 * (gx - 1)].p3[2] = z; tris[t - gx + 1].p2[2] = z; }
                if (y != gy - 1) tris[t].p1[2] = z;
            }
        }
}

}  // extern "C"
