// liblrlhost.so — C entry points over the host-side C++ scene library (lrl/RayTracing.hpp), so the
// Python tests, bench.py and tools can drive it through ctypes.  Not part of the device ABI
// (include/lrl_trace.h); these are conveniences for building the reference's scenes and BVH.
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>

#include "lrl/RayTracing.hpp"
#include "lrl/Scenes.hpp"

using namespace lrl;

namespace {
thread_local std::string g_err;
template <class F> int guarded(F&& f) {
    try { f(); return 0; }
    catch (std::exception const& e) { g_err = e.what(); return 1; }
    catch (...) { g_err = "unknown C++ exception"; return 1; }
}
struct Handle {
    Model model;
    scenes::Camera cam;
    bool poolWaterPending = false;   // scenes::pool was built; the water heights follow the BVH build
};
}  // namespace

extern "C" {

const char* lrlh_last_error() { return g_err.c_str(); }

Handle* lrlh_model_create() { return new (std::nothrow) Handle(); }
void lrlh_model_destroy(Handle* h) { delete h; }

int lrlh_scene_bvh_default(Handle* h, const char* resdir, int loadCube) {
    return guarded([&] { h->cam = scenes::bvhDefault(h->model, resdir, loadCube != 0); });
}
int lrlh_scene_mesh(Handle* h, const char* stlPath, unsigned subdivisions) {
    return guarded([&] { h->cam = scenes::mesh(h->model, STL::readSTL(stlPath), subdivisions); });
}
int lrlh_scene_mesh_grid(Handle* h, const char* stlPath, unsigned subdivisions, unsigned grid) {
    return guarded([&] { h->cam = scenes::mesh(h->model, STL::readSTL(stlPath), subdivisions, grid); });
}
int lrlh_scene_pool(Handle* h, const char* resdir, int loadTextures) {
    return guarded([&] { h->cam = scenes::pool(h->model, resdir, loadTextures != 0); h->poolWaterPending = true; });
}

// generic adders (records in the layouts of include/lrl_types.h)
int lrlh_add_triangles(Handle* h, const LrlTriangle* t, unsigned n) {
    return guarded([&] {
        auto& v = h->model.triangles.trianglesOrigin.trianglesOrigin;
        for (unsigned i = 0; i < n; ++i) {
            Model::Triangles::TriangleOriginData::TriangleOrigin o;
            static_assert(sizeof(o) == sizeof(LrlTriangle), "layout");
            std::memcpy(static_cast<void*>(&o), t + i, sizeof(o));
            v.pushBack(o);
        }
    });
}
#define LRLH_ADDER(NAME, CTYPE, HTYPE, MEMBER)                                   \
    int NAME(Handle* h, const CTYPE* p, unsigned n) {                            \
        return guarded([&] {                                                     \
            for (unsigned i = 0; i < n; ++i) {                                   \
                HTYPE o;                                                         \
                static_assert(sizeof(o) == sizeof(CTYPE), "layout");             \
                std::memcpy(static_cast<void*>(&o), p + i, sizeof(o));           \
                h->model.MEMBER.pushBack(o);                                     \
            }                                                                    \
        });                                                                      \
    }
LRLH_ADDER(lrlh_add_spheres, LrlSphere, Model::Spheres::Sphere, spheres.data.spheres)
LRLH_ADDER(lrlh_add_circles, LrlCircle, Model::Circles::Circle, circles.data.circles)
LRLH_ADDER(lrlh_add_cylinders_raw, LrlCylinder, Model::Cylinders::Cylinder, cylinders.data.cylinders)
LRLH_ADDER(lrlh_add_cones_raw, LrlCone, Model::Cones::Cone, cones.data.cones)
LRLH_ADDER(lrlh_add_point_lights, LrlPointLight, Model::PointLights::PointLight, pointLights.data.pointLights)
#undef LRLH_ADDER

// composite adders as the reference's Model::addCylinder / addCone (caps pushed as circles)
int lrlh_add_cylinder(Handle* h, const LrlCylinder* c) {
    return guarded([&] { Model::Cylinders::Cylinder o; std::memcpy(static_cast<void*>(&o), c, sizeof(o)); h->model.addCylinder(o); });
}
int lrlh_add_cone(Handle* h, const LrlCone* c) {
    return guarded([&] { Model::Cones::Cone o; std::memcpy(static_cast<void*>(&o), c, sizeof(o)); h->model.addCone(o); });
}
int lrlh_add_stl(Handle* h, const char* path, const LrlColor* color, unsigned n) {
    return guarded([&] {
        STL s = STL::readSTL(path);
        Model::Color c; std::memcpy(static_cast<void*>(&c), color, sizeof(c));
        h->model.addSTL(s, c, n ? n : s.triangles.length);
    });
}
int lrlh_set_cube_dir(Handle* h, const char* dir) { return guarded([&] { h->model.setCube(BMPCubeData(dir)); }); }
int lrlh_set_cube_raw(Handle* h, unsigned size, const uint8_t* const faces[6]) {
    return guarded([&] {
        h->model.cubeSize = size;
        for (int i = 0; i < 6; ++i) h->model.cubeBGR[i].assign(faces[i], faces[i] + size_t(size) * size * 3);
    });
}
int lrlh_set_tex_raw(Handle* h, unsigned w, unsigned hh, unsigned layers, const uint8_t* bgr) {
    return guarded([&] {
        h->model.texWidth = w; h->model.texHeight = hh; h->model.texLayers = layers;
        h->model.texBGR.assign(bgr, bgr + size_t(w) * hh * layers * 3);
    });
}

// model.dataInit(): counts + CPU BVH build.  builder: 0 = reference algorithm, 1 = balanced median.
int lrlh_data_init(Handle* h, int builder) {
    return guarded([&] {
        h->model.builder = builder == 1 ? Model::Builder::Median : (builder == 3 ? Model::Builder::Sah : Model::Builder::Reference);
        if (builder == 2) h->model.dataInitWithoutTree();      // counts only: the tracer library builds the tree on the device
        else h->model.dataInit();
        if (h->poolWaterPending) { scenes::poolWaterSync(h->model); h->poolWaterPending = false; }   // Pool/Pool.cpp:649-650: after dataInit
    });
}
int lrlh_scene_view(Handle* h, LrlSceneView* out) { return guarded([&] { *out = h->model.sceneView(); }); }
int lrlh_bvh_info(Handle* h, unsigned* nodes, unsigned* maxDepth, unsigned* fallbackSplits) {
    *nodes = h->model.bvh.data.nodes.length; *maxDepth = h->model.bvh.maxDepth; *fallbackSplits = h->model.bvh.fallbackSplits;
    return 0;
}
// the fixture camera of the scene built last (or identity/(0,0,30) by default) for a frame of `height` rows
int lrlh_camera(Handle* h, unsigned height, LrlTrans* out) { return guarded([&] { *out = h->cam.toTrans(height); }); }
int lrlh_camera_set(Handle* h, double fovy, const double r0[3], const double trans[9]) {
    h->cam.fovy = fovy;
    for (int i = 0; i < 3; ++i) { h->cam.r0[i] = r0[i]; for (int j = 0; j < 3; ++j) h->cam.trans.data[i][j] = trans[i * 3 + j]; }
    return 0;
}
void lrlh_sheared(const LrlTrans* t, float dx, float dy, LrlTrans* out) { *out = Transform::sheared(*t, dx, dy); }

// Scene dump for the llvmpipe harness (oracle/gl_harness.c) and for fixtures: little-endian,
//   "LRLS" u32 version=1, LrlGeometryNum(32), u32 bvhNodeNum, u32 texW,texH,texLayers, u32 cubeSize, u32 pad[2],
//   then triangles(160) spheres circles cylinders cones (128 each) pointLights(32) bvh(64) tex(BGR) cube(6 faces BGR)
int lrlh_scene_dump(Handle* h, const char* path) {
    return guarded([&] {
        LrlSceneView v = h->model.sceneView();
        FILE* f = std::fopen(path, "wb");
        if (!f) throw std::runtime_error(std::string("cannot write ") + path);
        uint32_t head[10] = {0x534c524cu, 1u, 0, 0, 0, 0, 0, 0, 0, 0};
        std::fwrite(head, 4, 2, f);
        std::fwrite(&v.num, sizeof(v.num), 1, f);
        uint32_t meta[8] = {v.bvhNodeNum, v.texWidth, v.texHeight, v.texLayers, v.cubeSize, 0, 0, 0};
        std::fwrite(meta, 4, 8, f);
        auto put = [&](void const* p, size_t rec, size_t n) { if (n) std::fwrite(p, rec, n, f); };
        put(v.trianglesOrigin, 160, v.num.triangleNum);
        put(v.spheres, 128, v.num.sphereNum);
        put(v.circles, 128, v.num.circleNum);
        put(v.cylinders, 128, v.num.cylinderNum);
        put(v.cones, 128, v.num.coneNum);
        put(v.pointLights, 32, v.num.pointLightNum);
        put(v.bvh, 64, v.bvhNodeNum);
        put(v.texBGR, 1, size_t(v.texWidth) * v.texHeight * v.texLayers * 3);
        for (int i = 0; i < 6; ++i) put(v.cubeBGR[i], 1, size_t(v.cubeSize) * v.cubeSize * 3);
        std::fclose(f);
    });
}

}  // extern "C"
