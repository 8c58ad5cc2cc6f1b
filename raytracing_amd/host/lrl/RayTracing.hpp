// Host-side scene / BVH / camera library: a restatement of the reference's external header
// <RayTracing/_RayTracing.h> (absent from /root/reference; its machine code is inside BVH/BVH.exe).
// What is restated, and from where (SURVEY §2.3 + Appendix A):
//   * member / method names               from the call sites in BVH/BVH.cpp:159-173,189-197,245-252,
//                                         280-710,716-750 and Pool/Pool.cpp:242-298,465,499-606
//   * record layouts                      from the GLSL declarations (include/lrl_types.h)
//   * addSTL / addCylinder / addCone      BVH.exe@0x140008c30 / @0x140007b78 / @0x140007e80
//   * BVH leaves, split, flatten          BVH.exe@0x140006320 / @0x140005c90 / @0x1400061d0
//   * camera z0                           BVH.exe@0x1400049a8
// GL objects (Buffer, BufferConfig, binding points) have no counterpart: the device side is the
// C ABI of include/lrl_trace.h, fed from Model::sceneView().
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "../../../include/lrl_trace.h"
#include "../../../include/lrl_types.h"
#include "BMP.hpp"
#include "Math.hpp"
#include "STL.hpp"
#include "Vector.hpp"

namespace lrl {

using vec2f = Math::vec2<float>;
using vec3f = Math::vec3<float>;
using vec4f = Math::vec4<float>;

struct FrameScale {
    int w = 0, h = 0;
    struct { unsigned int data[2]; } scale{{0, 0}};
};

// Full-screen quad of the reference's Renderer program (BVH/BVH.cpp:17-31); kept for API parity only.
struct View {
    vec2f vertices[4] = {{-1, -1}, {1, -1}, {1, 1}, {-1, 1}};
};

struct DecayOriginData {
    LrlDecayOrigin data{};
    DecayOriginData() { data.decayOrigin[3] = 1.0f; for (auto& d : data.decayOrigins) d[3] = 1.0f; }
    void* pointer() { return &data; }
    unsigned int size() const { return 144; }
};

// ------------------------------------------------------------------------------------------------
// Camera.  Ctor argument order as used at BVH/BVH.cpp:245:
//   { {fovy}, {scroll increaseDelta, decreaseRatio, threshold}, {key ratio}, {initial position}, depth }
// z0 = -float(persp.y / tan(fovy*pi/360)) (BVH.exe@0x1400049a8); trans starts as identity.
// operate(): the reference's mouse/keyboard maths was not recovered (interactive only, SURVEY App. A
// "Not recovered"); the version here is this library's own: keys translate in the camera frame,
// scroll dollies along the view axis, left-drag rotates about the point `depth` in front of the eye.
// ------------------------------------------------------------------------------------------------
struct Transform {
    struct Data {
        struct Perspective { double fovy; int y = 0; bool updated = false; } persp;
        struct Scroll { double increaseDelta, decreaseRatio, threshold; } scroll;
        struct Key { double ratio; } key;
        Math::vec3<double> initialPosition;
        double depth;
    };
    struct Perspective { double fovy; int y; bool updated; };
    struct Scroll {
        double increaseDelta, decreaseRatio, threshold, total = 0;
        void refresh(double d) { total += d * increaseDelta; }
        double operate() {
            if (std::fabs(total) > threshold) { double r = total; total *= decreaseRatio; return r; }
            total = 0; return 0;
        }
    };
    struct Key {
        bool left = false, right = false, up = false, down = false;
        double ratio;
        void refresh(int k, bool pressed) { (k == 0 ? left : k == 1 ? right : k == 2 ? up : down) = pressed; }
        Math::vec2<double> operate() const { return {ratio * (int(right) - int(left)), ratio * (int(up) - int(down))}; }
    };
    struct Mouse {
        Math::vec2<double> now, pre;
        bool left = false, middle = false, right = false;
        void refreshPos(double x, double y) { if (left) { pre = now; } now = {x, y}; }
        void refreshButton(int b, bool pressed) { (b == 0 ? left : b == 1 ? middle : right) = pressed; pre = now; }
        Math::vec2<double> operate() { Math::vec2<double> d{now[0] - pre[0], now[1] - pre[1]}; pre = now; return left ? d : Math::vec2<double>{0, 0}; }
    };
    struct BufferData {
        LrlTrans data{};
        void* pointer() { return &data; }
        unsigned int size() const { return 64; }
    };

    Perspective persp;
    Scroll scroll;
    Key key;
    Mouse mouse;
    BufferData bufferData;
    Math::vec3<double> dr;       // eye position (double, narrowed into bufferData.r0)
    Math::mat3<double> trans;    // camera rotation
    double depth;
    bool moved, updated;

    Transform(Data const& d)
        : persp{d.persp.fovy, d.persp.y, false}, scroll{d.scroll.increaseDelta, d.scroll.decreaseRatio, d.scroll.threshold},
          key{false, false, false, false, d.key.ratio}, dr(d.initialPosition), trans(Math::mat3<double>::id()),
          depth(d.depth), moved(true), updated(false) {}

    void refreshBuffer() {
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) bufferData.data.trans[i][j] = float(trans.data[i][j]);
            bufferData.data.trans[i][3] = 0;
            bufferData.data.r0[i] = float(dr[i]);
        }
    }
    void calcZ0() { bufferData.data.z0 = -float(double(persp.y) / std::tan(persp.fovy * Math::Pi / 360.0)); }
    void init(FrameScale const& size) {
        persp.y = size.h;
        persp.updated = false;
        calcZ0();
        refreshBuffer();
        updated = true;
    }
    void operate() {
        bool changed = false;
        if (persp.updated) { calcZ0(); persp.updated = false; updated = true; }
        Math::vec2<double> k = key.operate();
        double s = scroll.operate();
        Math::vec2<double> m = mouse.operate();
        if (k[0] != 0 || k[1] != 0 || s != 0) {
            Math::vec3<double> d = trans * Math::vec3<double>(k[0], k[1], -s);
            dr = dr + d;
            changed = true;
        }
        if (m[0] != 0 || m[1] != 0) {
            double l = std::sqrt(m[0] * m[0] + m[1] * m[1]);
            double ang = l / depth, c = std::cos(ang), sn = std::sin(ang);
            // rotation axis in camera space: perpendicular to the drag, in the image plane
            double ax = -m[1] / l, ay = -m[0] / l;
            Math::mat3<double> r{};
            double axis[3] = {ax, ay, 0};
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j)
                    r.data[i][j] = (i == j ? c : 0) + (1 - c) * axis[i] * axis[j];
            r.data[0][1] += -sn * axis[2]; r.data[0][2] += sn * axis[1];
            r.data[1][0] += sn * axis[2];  r.data[1][2] += -sn * axis[0];
            r.data[2][0] += -sn * axis[1]; r.data[2][1] += sn * axis[0];
            trans = trans * r;
            changed = true;
        }
        if (changed) { moved = true; refreshBuffer(); updated = true; }   // the frame driver clears `moved` after DecayOriginCalc
    }
    // Sub-pixel jitter as a shear of trans: S(x,y,z0) = (x+2dx, y+2dy, z0).
    static LrlTrans sheared(LrlTrans const& t, float dx, float dy) {
        LrlTrans o = t;
        for (int i = 0; i < 3; ++i)
            o.trans[i][2] = t.trans[i][2] + (t.trans[i][0] * (2.0f * dx / t.z0) + t.trans[i][1] * (2.0f * dy / t.z0));
        return o;
    }
};

// ------------------------------------------------------------------------------------------------
struct Model {
    struct alignas(16) Color {
        vec3f r; int texR;
        vec3f t; int texT;
        vec3f d; int texD;
        vec3f g; int texG;
        vec3f decayFactor; float n;
    };
    static_assert(sizeof(Color) == 80, "Color layout");

    struct Planes {
        struct Plane { vec4f plane; Color color; };
        struct PlaneData { Vector<Plane> planes; } data;
        bool numChanged = false, upToDate = false, GPUUpToDate = false;
    };
    struct Triangles {
        struct TriangleOriginData {
            struct TriangleOrigin {
                vec4f vertices[3];                 // xyz used, 16-byte stride as in std430
                vec2f uv1, uv2, uv3;
                Color color;                       // alignas(16) leaves the 8 bytes Pool uses for `changed`
            };
            static_assert(sizeof(TriangleOrigin) == 160, "Triangle layout");
            Vector<TriangleOrigin> trianglesOrigin;
        } trianglesOrigin;
        bool numChanged = false, upToDate = false, GPUUpToDate = false;
    };
    struct Spheres {
        struct Sphere { vec4f sphere; vec4f e1; vec4f e2; Color color; };
        struct SphereData { Vector<Sphere> spheres; } data;
        bool numChanged = false, upToDate = false, GPUUpToDate = false;
    };
    struct Circles {
        struct Circle { vec4f plane; vec3f sphere; float r2; vec3f e1; Color color; };
        struct CircleData { Vector<Circle> circles; } data;
        bool numChanged = false, upToDate = false, GPUUpToDate = false;
    };
    struct Cylinders {
        struct Cylinder { vec3f c; float r2; vec3f n; float l; vec3f e1; Color color; };
        struct CylinderData { Vector<Cylinder> cylinders; } data;
        bool numChanged = false, upToDate = false, GPUUpToDate = false;
    };
    struct Cones {
        struct Cone { vec3f c; float c2; vec3f n; float l2; vec3f e1; Color color; };
        struct ConeData { Vector<Cone> cones; } data;
        bool numChanged = false, upToDate = false, GPUUpToDate = false;
    };
    struct PointLights {
        struct PointLight { vec4f color; vec4f p; };
        struct PointLightData { Vector<PointLight> pointLights; } data;
        bool numChanged = false, upToDate = false, GPUUpToDate = false;
    };
    static_assert(sizeof(Spheres::Sphere) == 128 && sizeof(Circles::Circle) == 128, "layout");
    static_assert(sizeof(Cylinders::Cylinder) == 128 && sizeof(Cones::Cone) == 128, "layout");
    static_assert(sizeof(PointLights::PointLight) == 32 && sizeof(Planes::Plane) == 96, "layout");

    struct GeometryNum {
        struct Data { LrlGeometryNum num{}; } data;
    };

    struct BVH {
        struct BVHData { Vector<LrlBVHNode> nodes; } data;
        unsigned int maxDepth = 0;      // edges on the longest root-to-leaf path
        unsigned int fallbackSplits = 0; // times the median-by-index guard fired (0 on the reference's scenes)
    };

    Planes planes;
    Triangles triangles;
    Spheres spheres;
    Circles circles;
    Cylinders cylinders;
    Cones cones;
    PointLights pointLights;
    GeometryNum geometryNum;
    BVH bvh;
    bool moved = true;

    // textures travel with the model on this side of the ABI
    std::vector<uint8_t> texBGR; unsigned texWidth = 0, texHeight = 0, texLayers = 0;
    std::vector<uint8_t> cubeBGR[6]; unsigned cubeSize = 0;

    // -------- scene assembly --------
    // BVH.exe@0x140008c30: winding fixed so the geometric normal agrees with the stored STL normal;
    // default uv (0,0),(1,0),(0,1).
    void addSTL(STL const& stl, Color const& color, unsigned int n) {
        n = std::min(n, stl.triangles.length);
        for (unsigned int i = 0; i < n; ++i) {
            STL::Triangle const& t = stl.triangles[i];
            vec3f v1(t.vertices[0][0], t.vertices[0][1], t.vertices[0][2]);
            vec3f v2(t.vertices[1][0], t.vertices[1][1], t.vertices[1][2]);
            vec3f v3(t.vertices[2][0], t.vertices[2][1], t.vertices[2][2]);
            vec3f nrm(t.normal[0], t.normal[1], t.normal[2]);
            float s = (v2 - v1).cross(v3 - v1).dot(nrm);
            Triangles::TriangleOriginData::TriangleOrigin o{};
            o.vertices[0] = {v1[0], v1[1], v1[2]};
            if (s > 0) { o.vertices[1] = {v2[0], v2[1], v2[2]}; o.vertices[2] = {v3[0], v3[1], v3[2]}; }
            else       { o.vertices[1] = {v3[0], v3[1], v3[2]}; o.vertices[2] = {v2[0], v2[1], v2[2]}; }
            o.uv1 = {0, 0}; o.uv2 = {1, 0}; o.uv3 = {0, 1};
            o.color = color;
            triangles.trianglesOrigin.trianglesOrigin.pushBack(o);
        }
    }
    // BVH.exe@0x140007b78: two cap circles + the side.
    void addCylinder(Cylinders::Cylinder const& c) {
        Circles::Circle a{};
        a.plane = {c.n[0], c.n[1], c.n[2], 0};
        a.sphere = c.c; a.r2 = c.r2; a.e1 = c.e1; a.color = c.color;
        Circles::Circle b = a;
        b.sphere = c.c + c.n * c.l;
        circles.data.circles.pushBack(a);
        circles.data.circles.pushBack(b);
        cylinders.data.cylinders.pushBack(c);
    }
    // BVH.exe@0x140007e80: base circle (centre c + n*sqrt(c2*l2), R^2 = (1-c2)*l2) + the side.
    void addCone(Cones::Cone const& c) {
        Circles::Circle a{};
        a.plane = {c.n[0], c.n[1], c.n[2], 0};
        a.sphere = c.c + c.n * std::sqrt(c.c2 * c.l2);
        a.r2 = (1 - c.c2) * c.l2;
        a.e1 = c.e1; a.color = c.color;
        circles.data.circles.pushBack(a);
        cones.data.cones.pushBack(c);
    }
    void setTextureArray(std::vector<BMP const*> const& layers) {
        texLayers = unsigned(layers.size());
        if (!texLayers) { texWidth = texHeight = 0; texBGR.clear(); return; }
        texWidth = layers[0]->header.width; texHeight = layers[0]->header.height;
        texBGR.assign(size_t(texWidth) * texHeight * 3 * texLayers, 0);
        for (unsigned l = 0; l < texLayers; ++l) {
            // a smaller bitmap is uploaded into the corner of the layer (Pool floor.bmp is 1016x1024 in a
            // 1024^2 layer, Pool/Pool.cpp:491-492; the remaining texels are undefined in GL, zero here)
            BMP const& b = *layers[l];
            unsigned w = std::min<unsigned>(b.header.width, texWidth), h = std::min<unsigned>(b.header.height, texHeight);
            for (unsigned y = 0; y < h; ++y)
                std::memcpy(&texBGR[(size_t(l) * texHeight + y) * texWidth * 3], &b.data[size_t(y) * b.header.width * 3], size_t(w) * 3);
        }
    }
    void setCube(BMPCubeData const& c) {
        cubeSize = c.bmp[0].header.width;
        for (int i = 0; i < 6; ++i) cubeBGR[i] = c.bmp[i].data;
    }

    void upToDate() {
        planes.upToDate = triangles.upToDate = spheres.upToDate = circles.upToDate = cylinders.upToDate = cones.upToDate =
            pointLights.upToDate = true;
        planes.GPUUpToDate = triangles.GPUUpToDate = spheres.GPUUpToDate = circles.GPUUpToDate = cylinders.GPUUpToDate =
            cones.GPUUpToDate = pointLights.GPUUpToDate = true;
        planes.numChanged = triangles.numChanged = spheres.numChanged = circles.numChanged = cylinders.numChanged =
            cones.numChanged = pointLights.numChanged = false;
        moved = false;
    }

    // model.dataInit() (BVH/BVH.cpp:722): counts + full CPU BVH rebuild.  The upload half is
    // lrl_scene_upload(ctx, &model.sceneView()).
    // counts only, no tree: for callers that let the tracer library build the BVH on the device (LrlSceneView::bvh = NULL)
    void dataInitWithoutTree() {
        dataInit(false);
        bvh.data.nodes.clear();
    }
    void dataInit(bool buildTree = true) {
        LrlGeometryNum& g = geometryNum.data.num;
        g.planeNum = planes.data.planes.length;
        g.triangleNum = triangles.trianglesOrigin.trianglesOrigin.length;
        g.sphereNum = spheres.data.spheres.length;
        g.circleNum = circles.data.circles.length;
        g.cylinderNum = cylinders.data.cylinders.length;
        g.coneNum = cones.data.cones.length;
        g.pointLightNum = pointLights.data.pointLights.length;
        if (buildTree) buildBVH();
        triangles.GPUUpToDate = false;
        circles.GPUUpToDate = false;
        moved = true;
    }

    LrlSceneView sceneView() const {
        LrlSceneView v{};
        v.num = geometryNum.data.num;
        v.trianglesOrigin = reinterpret_cast<LrlTriangle const*>(triangles.trianglesOrigin.trianglesOrigin.data);
        v.spheres = reinterpret_cast<LrlSphere const*>(spheres.data.spheres.data);
        v.circles = reinterpret_cast<LrlCircle const*>(circles.data.circles.data);
        v.cylinders = reinterpret_cast<LrlCylinder const*>(cylinders.data.cylinders.data);
        v.cones = reinterpret_cast<LrlCone const*>(cones.data.cones.data);
        v.pointLights = reinterpret_cast<LrlPointLight const*>(pointLights.data.pointLights.data);
        v.bvh = bvh.data.nodes.data;
        v.bvhNodeNum = bvh.data.nodes.length;
        v.texWidth = texWidth; v.texHeight = texHeight; v.texLayers = texLayers;
        v.texBGR = texBGR.empty() ? nullptr : texBGR.data();
        v.cubeSize = cubeSize;
        for (int i = 0; i < 6; ++i) v.cubeBGR[i] = cubeSize ? cubeBGR[i].data() : nullptr;
        return v;
    }

    // -------- CPU BVH builder --------
    struct Leaf {
        float min[3], max[3], centre[3], weight;
        uint32_t geometry, geometryNum;
    };
    enum class Builder { Reference, Median, Sah };
    Builder builder = Builder::Reference;

    static void boxFromDisc(float const c[3], float const n[3], float R, float mn[3], float mx[3]) {
        float nl = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        for (int i = 0; i < 3; ++i) {
            float ni = nl > 0 ? n[i] / nl : 0;
            float e = float(1.05 * double(R) * std::sqrt(std::max(0.0, 1.0 - double(ni) * ni)));
            mn[i] = c[i] - e; mx[i] = c[i] + e;
        }
    }
    void makeLeaves(std::vector<Leaf>& leaves) const {
        leaves.clear();
        auto const& T = triangles.trianglesOrigin.trianglesOrigin;
        leaves.reserve(size_t(T.length) + spheres.data.spheres.length + circles.data.circles.length +
                       cylinders.data.cylinders.length + cones.data.cones.length);
        for (unsigned i = 0; i < T.length; ++i) {           // BVH.exe@0x140004d00
            Leaf l{};
            for (int k = 0; k < 3; ++k) {
                float a = T[i].vertices[0][k], b = T[i].vertices[1][k], c = T[i].vertices[2][k];
                l.min[k] = std::min(a, std::min(b, c));
                l.max[k] = std::max(a, std::max(b, c));
                l.centre[k] = 0.5f * (l.min[k] + l.max[k]);
            }
            vec3f p1(T[i].vertices[0][0], T[i].vertices[0][1], T[i].vertices[0][2]);
            vec3f p2(T[i].vertices[1][0], T[i].vertices[1][1], T[i].vertices[1][2]);
            vec3f p3(T[i].vertices[2][0], T[i].vertices[2][1], T[i].vertices[2][2]);
            vec3f cr = (p2 - p1).cross(p3 - p1);
            l.weight = 0.5f * std::sqrt(cr.dot(cr));
            l.geometry = 2; l.geometryNum = i;
            leaves.push_back(l);
        }
        for (unsigned i = 0; i < spheres.data.spheres.length; ++i) {   // BVH.exe@0x140005180
            auto const& s = spheres.data.spheres[i];
            Leaf l{};
            float d = float(1.05 * double(std::sqrt(s.sphere[3])));
            for (int k = 0; k < 3; ++k) { l.min[k] = s.sphere[k] - d; l.max[k] = s.sphere[k] + d; l.centre[k] = s.sphere[k]; }
            l.weight = float(4 * Math::Pi * double(s.sphere[3]));
            l.geometry = 3; l.geometryNum = i;
            leaves.push_back(l);
        }
        for (unsigned i = 0; i < circles.data.circles.length; ++i) {   // BVH.exe@0x1400052e0
            auto const& c = circles.data.circles[i];
            Leaf l{};
            float n[3] = {c.plane[0], c.plane[1], c.plane[2]};
            boxFromDisc(c.sphere.data, n, std::sqrt(c.r2), l.min, l.max);
            for (int k = 0; k < 3; ++k) l.centre[k] = c.sphere[k];
            l.weight = float(Math::Pi * double(c.r2));
            l.geometry = 4; l.geometryNum = i;
            leaves.push_back(l);
        }
        for (unsigned i = 0; i < cylinders.data.cylinders.length; ++i) {   // BVH.exe@0x140005560
            auto const& c = cylinders.data.cylinders[i];
            Leaf l{};
            float R = std::sqrt(c.r2);
            boxFromDisc(c.c.data, c.n.data, R, l.min, l.max);
            for (int k = 0; k < 3; ++k) {
                float sh = c.n[k] * c.l;
                l.min[k] = std::min(l.min[k], l.min[k] + sh);
                l.max[k] = std::max(l.max[k], l.max[k] + sh);
                l.centre[k] = c.c[k] + 0.5f * sh;
            }
            l.weight = float(2 * Math::Pi * double(R) * c.l);
            l.geometry = 5; l.geometryNum = i;
            leaves.push_back(l);
        }
        for (unsigned i = 0; i < cones.data.cones.length; ++i) {   // BVH.exe@0x140005910 (restated from geometry)
            auto const& c = cones.data.cones[i];
            Leaf l{};
            float h = std::sqrt(c.c2 * c.l2), R = std::sqrt((1 - c.c2) * c.l2);
            float base[3] = {c.c[0] + c.n[0] * h, c.c[1] + c.n[1] * h, c.c[2] + c.n[2] * h};
            boxFromDisc(base, c.n.data, R, l.min, l.max);
            for (int k = 0; k < 3; ++k) {
                l.min[k] = std::min(l.min[k], c.c[k]);
                l.max[k] = std::max(l.max[k], c.c[k]);
                l.centre[k] = c.c[k] + 0.75f * h * c.n[k];
            }
            l.weight = float(Math::Pi * double(R) * std::sqrt(double(c.l2)));
            l.geometry = 6; l.geometryNum = i;
            leaves.push_back(l);
        }
    }

   private:
    static LrlBVHNode leafNode(Leaf const& l, uint32_t father) {
        LrlBVHNode n{};
        for (int k = 0; k < 3; ++k) { n.bound.min[k] = l.min[k]; n.bound.max[k] = l.max[k]; }
        n.father = father; n.axis = 0; n.geometry = l.geometry; n.geometryNum = l.geometryNum;
        return n;
    }
    struct BuildCtx {
        std::vector<Leaf> const* leaves;
        std::vector<uint32_t> idx, tmp;
        Vector<LrlBVHNode>* out;
        unsigned maxDepth = 0, fallback = 0;
        Builder builder;
    };
    // emits the subtree over idx[b,e) in pre-order; returns its root index
    static uint32_t emit(BuildCtx& cx, size_t b, size_t e, uint32_t father, unsigned depth) {
        cx.maxDepth = std::max(cx.maxDepth, depth);
        std::vector<Leaf> const& L = *cx.leaves;
        if (e - b == 1) {
            cx.out->pushBack(leafNode(L[cx.idx[b]], father));
            return cx.out->length - 1;
        }
        uint32_t self = cx.out->length;
        cx.out->pushBack(LrlBVHNode{});
        float mn[3], mx[3];
        for (int k = 0; k < 3; ++k) { mn[k] = L[cx.idx[b]].min[k]; mx[k] = L[cx.idx[b]].max[k]; }
        double wsum = 0, mean[3] = {0, 0, 0};
        for (size_t i = b; i < e; ++i) {
            Leaf const& l = L[cx.idx[i]];
            for (int k = 0; k < 3; ++k) {
                mn[k] = std::min(mn[k], l.min[k]); mx[k] = std::max(mx[k], l.max[k]);
                mean[k] += double(l.weight) * l.centre[k];
            }
            wsum += l.weight;
        }
        if (wsum > 0) for (int k = 0; k < 3; ++k) mean[k] /= wsum;
        else {  // all-degenerate set: unweighted mean keeps the split meaningful
            for (int k = 0; k < 3; ++k) mean[k] = 0;
            for (size_t i = b; i < e; ++i) for (int k = 0; k < 3; ++k) mean[k] += L[cx.idx[i]].centre[k];
            for (int k = 0; k < 3; ++k) mean[k] /= double(e - b);
        }
        double var[3] = {0, 0, 0};
        for (size_t i = b; i < e; ++i) {
            Leaf const& l = L[cx.idx[i]];
            double w = wsum > 0 ? l.weight : 1.0;
            for (int k = 0; k < 3; ++k) { double d = l.centre[k] - mean[k]; var[k] += w * d * d; }
        }
        // tie order of BVH.exe@0x140005fa5-0x140005fc3
        unsigned axis = (var[1] > var[0]) ? (var[1] <= var[2] ? 2u : 1u) : (var[0] > var[2] ? 0u : 2u);
        float meanAxis = float(mean[axis]);
        if (cx.builder == Builder::Sah) {
            // binned surface-area heuristic (not in the reference; a yardstick for what the tree's quality is worth to the tracer):
            // 16 bins over the centres' extent on each axis, one primitive per leaf, no in-node primitives
            constexpr int B = 16;
            float cmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, cmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
            for (size_t i = b; i < e; ++i)
                for (int k = 0; k < 3; ++k) { cmn[k] = std::min(cmn[k], L[cx.idx[i]].centre[k]); cmx[k] = std::max(cmx[k], L[cx.idx[i]].centre[k]); }
            auto half = [](const float* lo, const float* hi) { float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2]; return dx * dy + dy * dz + dz * dx; };
            double best = 1e300; int bestAxis = -1, bestBin = 0;
            for (int k = 0; k < 3; ++k) {
                if (!(cmx[k] > cmn[k])) continue;
                struct Bin { float lo[3], hi[3]; size_t n; } bins[B];
                for (auto& q : bins) { for (int a = 0; a < 3; ++a) { q.lo[a] = FLT_MAX; q.hi[a] = -FLT_MAX; } q.n = 0; }
                const float scale = float(B) / (cmx[k] - cmn[k]);
                for (size_t i = b; i < e; ++i) {
                    Leaf const& l = L[cx.idx[i]];
                    int bi = std::min(B - 1, int((l.centre[k] - cmn[k]) * scale));
                    for (int a = 0; a < 3; ++a) { bins[bi].lo[a] = std::min(bins[bi].lo[a], l.min[a]); bins[bi].hi[a] = std::max(bins[bi].hi[a], l.max[a]); }
                    ++bins[bi].n;
                }
                double rightCost[B]; size_t rightN[B];
                float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX}; size_t n = 0;
                for (int j = B - 1; j > 0; --j) {
                    for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], bins[j].lo[a]); hi[a] = std::max(hi[a], bins[j].hi[a]); }
                    n += bins[j].n; rightN[j] = n; rightCost[j] = n ? double(half(lo, hi)) * double(n) : 0.0;
                }
                for (int a = 0; a < 3; ++a) { lo[a] = FLT_MAX; hi[a] = -FLT_MAX; }
                n = 0;
                for (int j = 0; j + 1 < B; ++j) {
                    for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], bins[j].lo[a]); hi[a] = std::max(hi[a], bins[j].hi[a]); }
                    n += bins[j].n;
                    if (!n || !rightN[j + 1]) continue;
                    const double c = double(half(lo, hi)) * double(n) + rightCost[j + 1];
                    if (c < best) { best = c; bestAxis = k; bestBin = j; }
                }
            }
            size_t mid;
            if (bestAxis < 0) { mid = b + (e - b) / 2; bestAxis = 0; }      // all centres coincide: by index
            else {
                const float scale = float(B) / (cmx[bestAxis] - cmn[bestAxis]);
                auto it = std::stable_partition(cx.idx.begin() + b, cx.idx.begin() + e, [&](uint32_t x) {
                    return std::min(B - 1, int((L[x].centre[bestAxis] - cmn[bestAxis]) * scale)) <= bestBin; });
                mid = size_t(it - cx.idx.begin());
            }
            LrlBVHNode node{};
            for (int k = 0; k < 3; ++k) { node.bound.min[k] = mn[k]; node.bound.max[k] = mx[k]; }
            node.father = father; node.axis = unsigned(bestAxis);
            node.bound.leftChild = 1;
            emit(cx, b, mid, self, depth + 1);
            node.bound.rightChild = int32_t(emit(cx, mid, e, self, depth + 1));
            (*cx.out)[self] = node;
            return self;
        }
        if (cx.builder == Builder::Median) {
            // balanced alternative (not in the reference): object median on the axis of largest centre extent,
            // no in-node primitives; depth = ceil(log2 N), for scenes whose mean-split tree would pass 32 levels
            float cmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, cmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
            for (size_t i = b; i < e; ++i)
                for (int k = 0; k < 3; ++k) {
                    cmn[k] = std::min(cmn[k], L[cx.idx[i]].centre[k]);
                    cmx[k] = std::max(cmx[k], L[cx.idx[i]].centre[k]);
                }
            axis = 0;
            for (unsigned k = 1; k < 3; ++k) if (cmx[k] - cmn[k] > cmx[axis] - cmn[axis]) axis = k;
            size_t half = (e - b) / 2;
            std::nth_element(cx.idx.begin() + b, cx.idx.begin() + b + half, cx.idx.begin() + e, [&](uint32_t x, uint32_t y) {
                float cxv = L[x].centre[axis], cyv = L[y].centre[axis];
                return cxv < cyv || (cxv == cyv && x < y);
            });
            LrlBVHNode node{};
            for (int k = 0; k < 3; ++k) { node.bound.min[k] = mn[k]; node.bound.max[k] = mx[k]; }
            node.father = father; node.axis = axis;
            node.bound.leftChild = 1;
            emit(cx, b, b + half, self, depth + 1);
            node.bound.rightChild = int32_t(emit(cx, b + half, e, self, depth + 1));
            (*cx.out)[self] = node;
            return self;
        }

        // partition (stable): absorbed leaf -> node itself; centre < mean -> A (left), else B (right)
        uint32_t geometry = 0, geometryNum = 0;
        cx.tmp.clear();
        size_t w = b;
        for (size_t i = b; i < e; ++i) {
            uint32_t id = cx.idx[i];
            Leaf const& l = L[id];
            if (geometry == 0 && std::memcmp(l.min, mn, 12) == 0 && std::memcmp(l.max, mx, 12) == 0) {
                geometry = l.geometry; geometryNum = l.geometryNum;
                continue;
            }
            if (meanAxis > l.centre[axis]) cx.idx[w++] = id; else cx.tmp.push_back(id);
        }
        size_t nA = w - b, nB = cx.tmp.size();
        std::copy(cx.tmp.begin(), cx.tmp.end(), cx.idx.begin() + w);
        size_t e2 = w + nB;   // idx[b, b+nA) = A, idx[b+nA, e2) = B
        if ((nA == 0 || nB == 0) && geometry == 0 && (nA + nB) > 1) {
            // termination guard (SURVEY App. A "Termination hazard"): every centre on one side and nothing
            // absorbed -> the reference would recurse forever; split the list in half by index instead
            nA = (nA + nB) / 2; nB = (e2 - b) - nA;
            ++cx.fallback;
        }
        LrlBVHNode node{};
        for (int k = 0; k < 3; ++k) { node.bound.min[k] = mn[k]; node.bound.max[k] = mx[k]; }
        node.father = father; node.axis = axis; node.geometry = geometry; node.geometryNum = geometryNum;
        size_t lb = b, le = b + nA, rb = b + nA, re = e2;
        if (nA == 0) { lb = rb; le = re; rb = re = 0; }          // B takes the left slot
        if (le > lb) { node.bound.leftChild = 1; emit(cx, lb, le, self, depth + 1); }
        if (re > rb) { node.bound.rightChild = int32_t(emit(cx, rb, re, self, depth + 1)); }
        LrlBVHNode& slot = (*cx.out)[self];
        slot = node;
        return self;
    }

   public:
    void buildBVH() {
        std::vector<Leaf> leaves;
        makeLeaves(leaves);
        Vector<LrlBVHNode>& out = bvh.data.nodes;
        out.clear();
        bvh.maxDepth = 0; bvh.fallbackSplits = 0;
        if (leaves.empty()) {   // the reference would upload an empty buffer and read bvh[0] anyway: give it a dead node
            LrlBVHNode n{};
            for (int k = 0; k < 3; ++k) { n.bound.min[k] = 0; n.bound.max[k] = 0; }
            out.pushBack(n);
            return;
        }
        if (leaves.size() > 0x0fffffffu) throw std::runtime_error("buildBVH: too many primitives");
        BuildCtx cx;
        cx.leaves = &leaves; cx.out = &out; cx.builder = builder;
        cx.idx.resize(leaves.size());
        for (size_t i = 0; i < leaves.size(); ++i) cx.idx[i] = uint32_t(i);
        out.reserve(unsigned(2 * leaves.size()));
        emit(cx, 0, leaves.size(), 0, 0);
        bvh.maxDepth = cx.maxDepth; bvh.fallbackSplits = cx.fallback;
    }
};

}  // namespace lrl
