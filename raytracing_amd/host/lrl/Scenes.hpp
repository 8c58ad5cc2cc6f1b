// Scene fixtures: the reference demos' hard-coded scenes as functions over lrl::Model.
//   bvhDefault : the BVH demo scene, numbers from BVH/BVH.cpp:245,280-703 (94 triangles from 11 STL
//                files, 5 spheres, 1 + 10 + 5 circles, 5 cylinders, 5 cones, 3 lights), camera (0,0,30).
//   pool       : the Pool demo, Pool/Pool.cpp:465,484,499-606 + initModel :240-299 + water initial
//                condition :97-112 scattered into the water triangles as WaterPositionSync does
//                (Pool/shaders/WaterPositionSyncCompute0.cpp:49-74), which the reference runs once at
//                init (Pool/Pool.cpp:649-650).
//   mesh       : SURVEY §8d config C2 — one STL mesh (n levels of 1->4 subdivision), mirror-ish
//                material, ground disc, one light, camera framing the bounding box.  No counterpart
//                in the reference's apps (RenderSTL is a rasteriser); numbers fixed here.
// The materials are written through small constructors instead of the reference's 10-field brace
// lists; values are the reference's.
#pragma once
#include <stdexcept>
#include <string>

#include "RayTracing.hpp"

namespace lrl {
namespace scenes {

using Color = Model::Color;

inline Color material(vec3f r, vec3f t, vec3f d, vec3f g, vec3f decay, float n) {
    Color c{};
    c.r = r; c.texR = -1; c.t = t; c.texT = -1; c.d = d; c.texD = -1; c.g = g; c.texG = -1;
    c.decayFactor = decay; c.n = n;
    return c;
}
inline Color glass(vec3f decay, double n) { return material(1, 1, 0, 0, decay, float(n)); }
inline Color mirror() { return material(0.7, 0, 0, 0, 0, 1); }
inline Color matte() { return material(0, 0, 1, 0, 0, 1); }

struct Camera {
    double fovy = 60.0;
    Math::vec3<double> r0{0, 0, 30};
    Math::mat3<double> trans = Math::mat3<double>::id();
    LrlTrans toTrans(unsigned height) const {
        Transform t({{fovy}, {0.09, 0.9, 0.01}, {0.15}, r0, 700.0});
        t.trans = trans;
        FrameScale fs; fs.w = 0; fs.h = int(height);
        t.init(fs);
        return t.bufferData.data;
    }
};

// resdir = directory holding box/, three/, mirror.stl and (optionally) room/ — BVH/resources of the reference
inline Camera bvhDefault(Model& m, std::string const& resdir, bool loadCube = true) {
    auto stl = [&](char const* rel) { return STL::readSTL(resdir + "/" + rel); };
    {
        Model::Circles::Circle ground{};
        ground.plane = {0, 0, 1, 0}; ground.sphere = {0, 0, 0}; ground.r2 = 5000; ground.e1 = {1, 0, 0};
        ground.color = material(0, 0, 1, 0.1, 0, 1);
        ground.color.texD = -2;                                         // checkerboard
        m.circles.data.circles.pushBack(ground);
    }
    {
        struct S { double x, r2; Color c; };
        const S spheres[5] = {
            {0, 16, glass({-0.15, -0.15, 0}, 1.5)},
            {0, 16 * 0.95 * 0.95, glass({0.15, 0.15, 0}, 1 / 1.5)},   // air bubble inside the first
            {10, 16, mirror()},
            {20, 16, matte()},
            {30, 16, glass({-0.15, -0.15, 0}, 1.5)},
        };
        for (S const& s : spheres) {
            Model::Spheres::Sphere sp{};
            sp.sphere = {s.x, 0, 10, s.r2}; sp.e1 = {0, 0, 1, 0}; sp.e2 = {1, 0, 0, 0}; sp.color = s.c;
            m.spheres.data.spheres.pushBack(sp);
        }
    }
    m.pointLights.data.pointLights += {
        {{4000, 4000, 4000}, {0, 0, 300}},
        {{10, 10, 10}, {-10, 20, 20}},
        {{10, 10, 10}, {30, 20, 20}},
    };
    auto add = [&](char const* rel, Color const& c) { STL s = stl(rel); m.addSTL(s, c, s.triangles.length); };
    add("box/box1.stl", glass({0, 0, -0.03}, 1.5));
    add("box/box2.stl", mirror());
    add("box/box3.stl", matte());
    add("box/boxtransparent1.stl", glass({0, 0, -0.03}, 1.5));
    add("box/boxtransparent2.stl", glass({0, 0, 0.03}, 1.0 / 1.5));
    add("mirror.stl", mirror());
    {
        struct C { double y, z, r2, l; Color c; };
        const C cyl[5] = {
            {35, 1, 1, 2, glass({0, 0, -0.1}, 1.5)},
            {37.5, 1, 1, 2, mirror()},
            {32.5, 1, 1, 2, matte()},
            {30, 1, 1, 2, glass({0, 0, -0.01}, 1.5)},
            {30, 1.1, 1 * 0.95 * 0.95, 1.9, glass({0, 0, 0.01}, 1 / 1.5)},
        };
        for (C const& c : cyl) {
            Model::Cylinders::Cylinder k{};
            k.c = {-15, c.y, c.z}; k.r2 = float(c.r2); k.n = {0, 0, 1}; k.l = float(c.l); k.e1 = {1, 0, 0}; k.color = c.c;
            m.addCylinder(k);
        }
        struct K { double y, z, l2; Color c; };
        const K cone[5] = {
            {35, 7, 4, glass({-0.2, 0, -0.2}, 1.5)},
            {37, 7, 4, mirror()},
            {33, 7, 4, matte()},
            {31, 7, 4, glass({-0.2, 0, -0.2}, 1.5)},
            {31, 7 - 0.1, 4 - 0.5, glass({0.2, 0, 0.2}, 1 / 1.5)},
        };
        for (K const& c : cone) {
            Model::Cones::Cone k{};
            k.c = {-7, c.y, c.z}; k.c2 = 0.8f; k.n = {0, 0, -1}; k.l2 = float(c.l2); k.e1 = {1, 0, 0}; k.color = c.c;
            m.addCone(k);
        }
    }
    add("three/three1.stl", glass({0, 0, -0.03}, 1.5));
    add("three/three2.stl", mirror());
    add("three/three3.stl", matte());
    add("three/three transparent1.stl", glass({0, 0, -0.1}, 1.3));
    add("three/three transparent2.stl", glass({0, 0, 0.1}, 1.0 / 1.3));
    if (loadCube) m.setCube(BMPCubeData(resdir + "/room/"));
    m.planes.numChanged = m.triangles.numChanged = m.spheres.numChanged = m.circles.numChanged = true;
    m.cylinders.numChanged = m.cones.numChanged = m.pointLights.numChanged = true;
    return Camera{};
}

// C2: a single STL mesh on a ground disc.  The mesh is lifted so its lowest point rests on z = 0.
// grid > 1 (towards C5): grid x grid copies of the mesh side by side, vertices replicated (the format has no
// instancing), on one disc, framed by the same camera recipe.
inline Camera mesh(Model& m, STL const& in, unsigned subdivisions, unsigned grid = 1) {
    STL s = in;
    for (unsigned i = 0; i < subdivisions; ++i) s = s.subdivided();
    if (grid > 1) {
        float lo[2] = {FLT_MAX, FLT_MAX}, hi[2] = {-FLT_MAX, -FLT_MAX};
        for (unsigned i = 0; i < s.triangles.length; ++i)
            for (int v = 0; v < 3; ++v)
                for (int k = 0; k < 2; ++k) {
                    lo[k] = std::min(lo[k], s.triangles[i].vertices[v][k]);
                    hi[k] = std::max(hi[k], s.triangles[i].vertices[v][k]);
                }
        const float pitch = 1.15f * std::max(hi[0] - lo[0], hi[1] - lo[1]);
        const unsigned n0 = s.triangles.length;
        for (unsigned gy = 0; gy < grid; ++gy)
            for (unsigned gx = 0; gx < grid; ++gx) {
                if (gx == 0 && gy == 0) continue;
                for (unsigned i = 0; i < n0; ++i) {
                    auto t = s.triangles[i];
                    for (int v = 0; v < 3; ++v) { t.vertices[v][0] += pitch * gx; t.vertices[v][1] += pitch * gy; }
                    s.triangles.pushBack(t);
                }
            }
    }
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (unsigned i = 0; i < s.triangles.length; ++i)
        for (int v = 0; v < 3; ++v)
            for (int k = 0; k < 3; ++k) {
                mn[k] = std::min(mn[k], s.triangles[i].vertices[v][k]);
                mx[k] = std::max(mx[k], s.triangles[i].vertices[v][k]);
            }
    float lift = -mn[2];
    for (unsigned i = 0; i < s.triangles.length; ++i)
        for (int v = 0; v < 3; ++v) s.triangles[i].vertices[v][2] += lift;
    mx[2] += lift; mn[2] = 0;
    Color body = material(0.7, 0, 1, 0, 0, 1);                 // mirror-ish + diffuse so that 4 bounces occur
    m.addSTL(s, body, s.triangles.length);
    double cx = 0.5 * (mn[0] + mx[0]), cy = 0.5 * (mn[1] + mx[1]), cz = 0.5 * mx[2];
    double ext = std::max({double(mx[0] - mn[0]), double(mx[1] - mn[1]), double(mx[2])});
    {
        Model::Circles::Circle ground{};
        ground.plane = {0, 0, 1, 0}; ground.sphere = {cx, cy, 0}; ground.r2 = float(25 * ext * ext); ground.e1 = {1, 0, 0};
        ground.color = material(0.3, 0, 1, 0.05, 0, 1);
        ground.color.texD = -2;
        m.circles.data.circles.pushBack(ground);
    }
    double lp = 2.5 * ext;
    m.pointLights.data.pointLights += {{{lp * lp * 0.05, lp * lp * 0.05, lp * lp * 0.05}, {cx + 0.6 * lp, cy - 0.5 * lp, lp}}};
    // camera: close on the +y side, slightly above, looking at the bbox centre (the mesh fills about a third of
    // the frame); the camera looks down its -z
    Camera cam;
    cam.fovy = 60;
    Math::vec3<double> target(cx, cy, 0.9 * cz), eye(cx + 0.15 * ext, cy + 0.72 * ext, cz + 0.36 * ext);
    Math::vec3<double> f = target - eye;                       // forward
    double fl = std::sqrt(f.dot(f)); f = f * (1 / fl);
    Math::vec3<double> up(0, 0, 1);
    Math::vec3<double> r = f.cross(up); r = r * (1 / std::sqrt(r.dot(r)));
    Math::vec3<double> u = r.cross(f);
    for (int i = 0; i < 3; ++i) { cam.trans.data[i][0] = r[i]; cam.trans.data[i][1] = u[i]; cam.trans.data[i][2] = -f[i]; }
    cam.r0 = eye;
    return cam;
}


// Water heights at rest (WaterData, Pool/Pool.cpp:97-112; paused, :463) scattered into the triangle vertices the way
// WaterPositionSync does (Pool/shaders/WaterPositionSyncCompute0.cpp:49-74).  The reference runs it once at init AFTER
// model.dataInit() (Pool/Pool.cpp:649-650): the BVH is built over the tall proxy triangles of initModel, whose boxes
// (z0 +- dzMax) hold every height the water can take.  Call it after Model::dataInit().
inline void poolWaterSync(Model& m) {
    const unsigned gx = 8 * 6, gy = 8 * 6;
    const float z0 = 0.4f, dzMax = 0.15f;
    auto& tris = m.triangles.trianglesOrigin.trianglesOrigin;
    if (tris.length < 2 * (gx - 1) * (gy - 1)) throw std::runtime_error("poolWaterSync: not the pool scene");
    const float cX = float((gx - 1) / 2), cY = float((gy - 1) / 2);       // integer division, as in the reference
    const double pi = 3.14159265358979323846;
    for (unsigned y = 0; y < gy; ++y)
        for (unsigned x = 0; x < gx; ++x) {
            const float z = z0 + dzMax * 0.5f * sinf(float(2 * pi * x / cX)) * sinf(float(2 * pi * y / cY));
            const unsigned ti = x + y * 2 * (gx - 1);
            if (x != 0) {
                if (y != 0) tris[ti - gx].vertices[0][2] = z;
                if (y != gy - 1) { tris[ti - 1].vertices[1][2] = z; tris[ti + gx - 2].vertices[2][2] = z; }
            }
            if (x != gx - 1) {
                if (y != 0) { tris[ti - 2 * (gx - 1)].vertices[2][2] = z; tris[ti - gx + 1].vertices[1][2] = z; }
                if (y != gy - 1) tris[ti].vertices[0][2] = z;
            }
        }
}

// The same heights as the water grid's points {z, v = 0, a = 0} (WaterData's initial state, Pool/Pool.cpp:97-112): what
// lrl_water_init takes.
inline std::vector<LrlWaterPoint> poolWaterPoints() {
    const unsigned gx = 8 * 6, gy = 8 * 6;
    const float z0 = 0.4f, dzMax = 0.15f;
    const float cX = float((gx - 1) / 2), cY = float((gy - 1) / 2);
    const double pi = 3.14159265358979323846;
    std::vector<LrlWaterPoint> w(size_t(gx) * gy);
    for (unsigned y = 0; y < gy; ++y)
        for (unsigned x = 0; x < gx; ++x)
            w[x + size_t(gx) * y] = LrlWaterPoint{z0 + dzMax * 0.5f * sinf(float(2 * pi * x / cX)) * sinf(float(2 * pi * y / cY)), 0.0f, 0.0f, 0.0f};
    return w;
}

// Pool demo: 48 x 48 water grid as 2 * 47 * 47 tall proxy triangles (initModel, Pool/Pool.cpp:240-299; the heights come
// from poolWaterSync after the BVH is built), then the 28 triangles of pool.stl with the per-triangle materials and uv of Pool/Pool.cpp:515-606,
// one light, four texture layers (pool, brick, ceiling, floor) and the room cube map.
// resdir = directory holding pool.stl, pool.bmp, brick.bmp, ceiling.bmp, floor.bmp and room/.
inline Camera pool(Model& m, std::string const& resdir, bool loadTextures = true) {
    using Tri = Model::Triangles::TriangleOriginData::TriangleOrigin;
    const unsigned gx = 8 * 6, gy = 8 * 6;                      // groupSize * groupNum, Pool/Pool.cpp:484
    const float z0 = 0.4f, dx = float((1.0 - 0.01) / (8 * 6 - 1)), dzMax = 0.15f;
    const float x0 = float(-0.5 + 0.005), y0 = float(-0.5 + 0.005);
    m.pointLights.data.pointLights += {{{0.15, 0.15, 0.15}, {0, 0, 2.6}}};
    Color water = material(1, 1, 0, 0, -1, 1.33f);
    auto& tris = m.triangles.trianglesOrigin.trianglesOrigin;
    for (unsigned c0 = 0; c0 + 1 < gy; ++c0) {
        for (unsigned c1 = 0; c1 + 1 < gx; ++c1) {
            Tri t{};
            t.vertices[0] = {x0 + dx * c1, y0 + dx * c0, z0 - dzMax};
            t.vertices[1] = {x0 + dx * (c1 + 1), y0 + dx * c0, z0 - dzMax};
            t.vertices[2] = {x0 + dx * c1, y0 + dx * (c0 + 1), z0 + dzMax};
            t.uv1 = {0, 0}; t.uv2 = {0, 0}; t.uv3 = {0, 0};
            t.color = water;
            tris.pushBack(t);
        }
        for (unsigned c1 = 0; c1 + 1 < gx; ++c1) {
            Tri t{};
            t.vertices[0] = {x0 + dx * (c1 + 1), y0 + dx * (c0 + 1), z0 - dzMax};
            t.vertices[1] = {x0 + dx * c1, y0 + dx * (c0 + 1), z0 - dzMax};
            t.vertices[2] = {x0 + dx * (c1 + 1), y0 + dx * c0, z0 + dzMax};
            t.uv1 = {0, 0}; t.uv2 = {0, 0}; t.uv3 = {0, 0};
            t.color = water;
            tris.pushBack(t);
        }
    }
    const unsigned k = tris.length;
    {
        STL s = STL::readSTL(resdir + "/pool.stl");
        m.addSTL(s, material(0, 0, 1, 0, 0, 1), s.triangles.length);
    }
    if (tris.length < k + 28) throw std::runtime_error("pool.stl: 28 triangles expected");
    auto uv = [&](unsigned i, float a, float b, float c, float d, float e, float f) { tris[i].uv1 = {a, b}; tris[i].uv2 = {c, d}; tris[i].uv3 = {e, f}; };
    for (unsigned c0 = 0; c0 < 4; ++c0) {                        // pool walls: tiled pool texture as glow, behind water
        uv(k + 2 * c0, 5, 0, 5, 1.5f, 0, 0);
        uv(k + 2 * c0 + 1, 0, 1.5f, 0, 0, 5, 1.5f);
        for (unsigned i : {k + 2 * c0, k + 2 * c0 + 1}) { tris[i].color.d = 0; tris[i].color.g = 1; tris[i].color.texG = 0; tris[i].color.n = 1.33f; }
    }
    for (unsigned c0 = 8; c0 < 16; ++c0) { tris[k + c0].color.d = 1; tris[k + c0].color.g = 0; tris[k + c0].color.texD = 3; tris[k + c0].color.n = 1; }
    uv(k + 16, 5, 0, 0, 0, 5, 5);                                // pool bottom
    uv(k + 17, 0, 5, 5, 5, 0, 0);
    for (unsigned i : {k + 16, k + 17}) { tris[i].color.g = 1; tris[i].color.texG = 0; tris[i].color.decayFactor = -1; tris[i].color.n = 1.33f; }
    for (unsigned c0 = 9; c0 < 13; ++c0) {                       // room walls: brick
        uv(k + 2 * c0, 1, 0, 1, 1, 0, 0);
        uv(k + 2 * c0 + 1, 0, 1, 0, 0, 1, 1);
        for (unsigned i : {k + 2 * c0, k + 2 * c0 + 1}) { tris[i].color.d = 1; tris[i].color.g = 0; tris[i].color.texD = 1; tris[i].color.n = 1; }
    }
    uv(k + 26, 1, 1, 0, 0, 1, 0);                                // ceiling
    uv(k + 27, 1, 1, 0, 1, 0, 0);
    for (unsigned i : {k + 26, k + 27}) { tris[i].color.d = 1; tris[i].color.g = 0; tris[i].color.texD = 2; tris[i].color.n = 1; }
    if (loadTextures) {
        BMPData poolBMP(resdir + "/pool.bmp"), wallBMP(resdir + "/brick.bmp"), ceilingBMP(resdir + "/ceiling.bmp"), floorBMP(resdir + "/floor.bmp");
        m.setTextureArray({&poolBMP.bmp, &wallBMP.bmp, &ceilingBMP.bmp, &floorBMP.bmp});
        m.setCube(BMPCubeData(resdir + "/room/"));
    }
    m.planes.numChanged = m.triangles.numChanged = m.spheres.numChanged = m.circles.numChanged = true;
    m.cylinders.numChanged = m.cones.numChanged = m.pointLights.numChanged = true;
    Camera cam;
    cam.r0 = {0, 0, 1};                                          // Pool/Pool.cpp:465
    return cam;
}

}  // namespace scenes
}  // namespace lrl
