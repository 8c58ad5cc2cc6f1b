// gl_harness — TEST INFRASTRUCTURE, not product code.
// Runs the REFERENCE's own GLSL programs (read at run time from a directory given on the command
// line, e.g. /root/reference/BVH/shaders — nothing of them is copied into this repository) headless
// on Mesa llvmpipe through the raw DRI swrast interface, on a scene dump written
// by liblrlhost (lrlh_scene_dump).  It is how the CPU oracle (oracle/oracle_trace.cpp) is pinned and
// how the golden fixtures under tests/golden/ were produced (tools/make_golden.py).
//
// Textual changes made to the reference's shader text at load time, and only these:
//   * fragment shader: after line 1, declare `out vec4 lrl_FragColor` and `#define gl_FragColor
//     lrl_FragColor` (`#version 450 core` has no gl_FragColor and Mesa rejects it);
//   * `#define RayTraceDepth N` is re-valued when --depth is given (SURVEY §8d "4-bounce");
//   * with --ids, a second colour output receives {hit type, hit index, bits of t} of the primary
//     ray, assigned just before the first `Color tempColor;` of rayTrace() when depth == 0.
//
// Modes:
//   gl_harness render --shaders DIR --scene F --trans F --size W H [--depth N] [--min-linear]
//        [--out F] [--ids F] [--tri-out F] [--circle-out F] [--decay x y z w | --decay-compute] [--decay-out F]
//   gl_harness probe  --shader FILE --in F --out F --count N        (arithmetic probes, tools/probes/*.comp)
#include <GL/glcorearb.h>
#include <GL/internal/dri_interface.h>
#include <dlfcn.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

static int g_w = 64, g_h = 64;

static void getDrawableInfo(__DRIdrawable*, int* x, int* y, int* w, int* h, void*) { *x = *y = 0; *w = g_w; *h = g_h; }
static void putImage(__DRIdrawable*, int, int, int, int, int, char*, void*) {}
static void getImage(__DRIdrawable*, int, int, int w, int h, char* d, void*) { memset(d, 0, size_t(w) * h * 4); }
static void putImage2(__DRIdrawable*, int, int, int, int, int, int, char*, void*) {}
static void getImage2(__DRIdrawable*, int, int, int, int h, int stride, char* d, void*) { memset(d, 0, size_t(stride) * h); }

#define GLF(type, name) static type name
GLF(PFNGLGETSTRINGPROC, glGetString_);
GLF(PFNGLCREATESHADERPROC, glCreateShader_);
GLF(PFNGLSHADERSOURCEPROC, glShaderSource_);
GLF(PFNGLCOMPILESHADERPROC, glCompileShader_);
GLF(PFNGLGETSHADERIVPROC, glGetShaderiv_);
GLF(PFNGLGETSHADERINFOLOGPROC, glGetShaderInfoLog_);
GLF(PFNGLCREATEPROGRAMPROC, glCreateProgram_);
GLF(PFNGLATTACHSHADERPROC, glAttachShader_);
GLF(PFNGLLINKPROGRAMPROC, glLinkProgram_);
GLF(PFNGLGETPROGRAMIVPROC, glGetProgramiv_);
GLF(PFNGLGETPROGRAMINFOLOGPROC, glGetProgramInfoLog_);
GLF(PFNGLUSEPROGRAMPROC, glUseProgram_);
GLF(PFNGLGENBUFFERSPROC, glGenBuffers_);
GLF(PFNGLBINDBUFFERPROC, glBindBuffer_);
GLF(PFNGLBUFFERDATAPROC, glBufferData_);
GLF(PFNGLBUFFERSUBDATAPROC, glBufferSubData_);
GLF(PFNGLBINDBUFFERBASEPROC, glBindBufferBase_);
GLF(PFNGLGETBUFFERSUBDATAPROC, glGetBufferSubData_);
GLF(PFNGLGENVERTEXARRAYSPROC, glGenVertexArrays_);
GLF(PFNGLBINDVERTEXARRAYPROC, glBindVertexArray_);
GLF(PFNGLENABLEVERTEXATTRIBARRAYPROC, glEnableVertexAttribArray_);
GLF(PFNGLVERTEXATTRIBPOINTERPROC, glVertexAttribPointer_);
GLF(PFNGLGENTEXTURESPROC, glGenTextures_);
GLF(PFNGLBINDTEXTUREPROC, glBindTexture_);
GLF(PFNGLACTIVETEXTUREPROC, glActiveTexture_);
GLF(PFNGLTEXSTORAGE2DPROC, glTexStorage2D_);
GLF(PFNGLTEXSTORAGE3DPROC, glTexStorage3D_);
GLF(PFNGLTEXSUBIMAGE2DPROC, glTexSubImage2D_);
GLF(PFNGLTEXSUBIMAGE3DPROC, glTexSubImage3D_);
GLF(PFNGLTEXPARAMETERIPROC, glTexParameteri_);
GLF(PFNGLPIXELSTOREIPROC, glPixelStorei_);
GLF(PFNGLGENFRAMEBUFFERSPROC, glGenFramebuffers_);
GLF(PFNGLBINDFRAMEBUFFERPROC, glBindFramebuffer_);
GLF(PFNGLFRAMEBUFFERTEXTURE2DPROC, glFramebufferTexture2D_);
GLF(PFNGLCHECKFRAMEBUFFERSTATUSPROC, glCheckFramebufferStatus_);
GLF(PFNGLDRAWBUFFERSPROC, glDrawBuffers_);
GLF(PFNGLVIEWPORTPROC, glViewport_);
GLF(PFNGLCLEARCOLORPROC, glClearColor_);
GLF(PFNGLCLEARPROC, glClear_);
GLF(PFNGLDRAWARRAYSPROC, glDrawArrays_);
GLF(PFNGLFINISHPROC, glFinish_);
GLF(PFNGLREADBUFFERPROC, glReadBuffer_);
GLF(PFNGLREADPIXELSPROC, glReadPixels_);
GLF(PFNGLDISPATCHCOMPUTEPROC, glDispatchCompute_);
GLF(PFNGLMEMORYBARRIERPROC, glMemoryBarrier_);
GLF(PFNGLENABLEPROC, glEnable_);
GLF(PFNGLGETERRORPROC, glGetError_);

static void die(const char* msg) { fprintf(stderr, "gl_harness: %s\n", msg); exit(2); }

static void initGL(int w, int h) {
    g_w = w; g_h = h;
    void* glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!glapi) die("cannot dlopen libglapi.so.0");
    void* drv = dlopen("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so", RTLD_NOW | RTLD_GLOBAL);
    if (!drv) die("cannot dlopen swrast_dri.so");
    typedef const __DRIextension** (*GetExt)(void);
    GetExt getExt = (GetExt)dlsym(drv, "__driDriverGetExtensions_swrast");
    if (!getExt) die("no __driDriverGetExtensions_swrast");
    const __DRIextension** exts = getExt();
    const __DRIcoreExtension* core = nullptr;
    const __DRIswrastExtension* swrast = nullptr;
    for (int i = 0; exts[i]; ++i) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) core = (const __DRIcoreExtension*)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) swrast = (const __DRIswrastExtension*)exts[i];
    }
    if (!core || !swrast) die("DRI_Core / DRI_SWRast missing");
    static __DRIswrastLoaderExtension loader;
    memset(&loader, 0, sizeof(loader));
    loader.base.name = __DRI_SWRAST_LOADER;
    loader.base.version = 3;
    loader.getDrawableInfo = getDrawableInfo;
    loader.putImage = putImage;
    loader.getImage = getImage;
    loader.putImage2 = putImage2;
    loader.getImage2 = getImage2;
    static const __DRIextension* loaderExts[] = {&loader.base, nullptr};
    const __DRIconfig** configs = nullptr;
    __DRIscreen* scr = swrast->createNewScreen2(0, loaderExts, exts, &configs, nullptr);
    if (!scr || !configs) die("createNewScreen2 failed");
    const __DRIconfig* cfg = nullptr;
    for (int i = 0; configs[i]; ++i) {
        unsigned r = 0, a = 0, db = 1;
        core->getConfigAttrib(configs[i], __DRI_ATTRIB_RED_SIZE, &r);
        core->getConfigAttrib(configs[i], __DRI_ATTRIB_ALPHA_SIZE, &a);
        core->getConfigAttrib(configs[i], __DRI_ATTRIB_DOUBLE_BUFFER, &db);
        if (r == 8 && a == 8 && !db) { cfg = configs[i]; break; }
    }
    if (!cfg) cfg = configs[0];
    uint32_t attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 5};
    unsigned err = 0;
    __DRIcontext* ctx = swrast->createContextAttribs(scr, __DRI_API_OPENGL_CORE, cfg, nullptr, 2, attribs, &err, nullptr);
    if (!ctx) die("createContextAttribs(4.5 core) failed");
    __DRIdrawable* draw = swrast->createNewDrawable(scr, cfg, nullptr);
    if (!draw || !core->bindContext(ctx, draw, draw)) die("bindContext failed");
    typedef void* (*GPA)(const char*);
    GPA gpa = (GPA)dlsym(glapi, "_glapi_get_proc_address");
    if (!gpa) die("no _glapi_get_proc_address");
#define LOAD(var, type, name) var = (type)gpa(name); if (!var) die("missing GL entry point " name)
    LOAD(glGetString_, PFNGLGETSTRINGPROC, "glGetString");
    LOAD(glCreateShader_, PFNGLCREATESHADERPROC, "glCreateShader");
    LOAD(glShaderSource_, PFNGLSHADERSOURCEPROC, "glShaderSource");
    LOAD(glCompileShader_, PFNGLCOMPILESHADERPROC, "glCompileShader");
    LOAD(glGetShaderiv_, PFNGLGETSHADERIVPROC, "glGetShaderiv");
    LOAD(glGetShaderInfoLog_, PFNGLGETSHADERINFOLOGPROC, "glGetShaderInfoLog");
    LOAD(glCreateProgram_, PFNGLCREATEPROGRAMPROC, "glCreateProgram");
    LOAD(glAttachShader_, PFNGLATTACHSHADERPROC, "glAttachShader");
    LOAD(glLinkProgram_, PFNGLLINKPROGRAMPROC, "glLinkProgram");
    LOAD(glGetProgramiv_, PFNGLGETPROGRAMIVPROC, "glGetProgramiv");
    LOAD(glGetProgramInfoLog_, PFNGLGETPROGRAMINFOLOGPROC, "glGetProgramInfoLog");
    LOAD(glUseProgram_, PFNGLUSEPROGRAMPROC, "glUseProgram");
    LOAD(glGenBuffers_, PFNGLGENBUFFERSPROC, "glGenBuffers");
    LOAD(glBindBuffer_, PFNGLBINDBUFFERPROC, "glBindBuffer");
    LOAD(glBufferData_, PFNGLBUFFERDATAPROC, "glBufferData");
    LOAD(glBufferSubData_, PFNGLBUFFERSUBDATAPROC, "glBufferSubData");
    LOAD(glBindBufferBase_, PFNGLBINDBUFFERBASEPROC, "glBindBufferBase");
    LOAD(glGetBufferSubData_, PFNGLGETBUFFERSUBDATAPROC, "glGetBufferSubData");
    LOAD(glGenVertexArrays_, PFNGLGENVERTEXARRAYSPROC, "glGenVertexArrays");
    LOAD(glBindVertexArray_, PFNGLBINDVERTEXARRAYPROC, "glBindVertexArray");
    LOAD(glEnableVertexAttribArray_, PFNGLENABLEVERTEXATTRIBARRAYPROC, "glEnableVertexAttribArray");
    LOAD(glVertexAttribPointer_, PFNGLVERTEXATTRIBPOINTERPROC, "glVertexAttribPointer");
    LOAD(glGenTextures_, PFNGLGENTEXTURESPROC, "glGenTextures");
    LOAD(glBindTexture_, PFNGLBINDTEXTUREPROC, "glBindTexture");
    LOAD(glActiveTexture_, PFNGLACTIVETEXTUREPROC, "glActiveTexture");
    LOAD(glTexStorage2D_, PFNGLTEXSTORAGE2DPROC, "glTexStorage2D");
    LOAD(glTexStorage3D_, PFNGLTEXSTORAGE3DPROC, "glTexStorage3D");
    LOAD(glTexSubImage2D_, PFNGLTEXSUBIMAGE2DPROC, "glTexSubImage2D");
    LOAD(glTexSubImage3D_, PFNGLTEXSUBIMAGE3DPROC, "glTexSubImage3D");
    LOAD(glTexParameteri_, PFNGLTEXPARAMETERIPROC, "glTexParameteri");
    LOAD(glPixelStorei_, PFNGLPIXELSTOREIPROC, "glPixelStorei");
    LOAD(glGenFramebuffers_, PFNGLGENFRAMEBUFFERSPROC, "glGenFramebuffers");
    LOAD(glBindFramebuffer_, PFNGLBINDFRAMEBUFFERPROC, "glBindFramebuffer");
    LOAD(glFramebufferTexture2D_, PFNGLFRAMEBUFFERTEXTURE2DPROC, "glFramebufferTexture2D");
    LOAD(glCheckFramebufferStatus_, PFNGLCHECKFRAMEBUFFERSTATUSPROC, "glCheckFramebufferStatus");
    LOAD(glDrawBuffers_, PFNGLDRAWBUFFERSPROC, "glDrawBuffers");
    LOAD(glViewport_, PFNGLVIEWPORTPROC, "glViewport");
    LOAD(glClearColor_, PFNGLCLEARCOLORPROC, "glClearColor");
    LOAD(glClear_, PFNGLCLEARPROC, "glClear");
    LOAD(glDrawArrays_, PFNGLDRAWARRAYSPROC, "glDrawArrays");
    LOAD(glFinish_, PFNGLFINISHPROC, "glFinish");
    LOAD(glReadBuffer_, PFNGLREADBUFFERPROC, "glReadBuffer");
    LOAD(glReadPixels_, PFNGLREADPIXELSPROC, "glReadPixels");
    LOAD(glDispatchCompute_, PFNGLDISPATCHCOMPUTEPROC, "glDispatchCompute");
    LOAD(glMemoryBarrier_, PFNGLMEMORYBARRIERPROC, "glMemoryBarrier");
    LOAD(glEnable_, PFNGLENABLEPROC, "glEnable");
    LOAD(glGetError_, PFNGLGETERRORPROC, "glGetError");
#undef LOAD
    fprintf(stderr, "gl_harness: %s | %s\n", (const char*)glGetString_(GL_VERSION), (const char*)glGetString_(GL_RENDERER));
}

static std::string readText(std::string const& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { fprintf(stderr, "gl_harness: cannot read %s\n", path.c_str()); exit(2); }
    std::stringstream ss; ss << f.rdbuf();
    std::string s = ss.str();
    std::string o; o.reserve(s.size());
    for (char c : s) if (c != '\r') o.push_back(c);
    return o;
}
static std::vector<uint8_t> readBin(std::string const& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { fprintf(stderr, "gl_harness: cannot read %s\n", path.c_str()); exit(2); }
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void writeBin(std::string const& path, void const* p, size_t n) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f || fwrite(p, 1, n, f) != n) { fprintf(stderr, "gl_harness: cannot write %s\n", path.c_str()); exit(2); }
    fclose(f);
}

static GLuint compile(GLenum type, std::string const& src, const char* what) {
    GLuint s = glCreateShader_(type);
    const char* p = src.c_str();
    glShaderSource_(s, 1, &p, nullptr);
    glCompileShader_(s);
    GLint ok = 0; glGetShaderiv_(s, GL_COMPILE_STATUS, &ok);
    if (!ok) {
        char log[8192]; GLsizei n = 0; glGetShaderInfoLog_(s, sizeof(log), &n, log);
        fprintf(stderr, "gl_harness: compile of %s failed:\n%s\n", what, log); exit(2);
    }
    return s;
}
static GLuint link(std::vector<GLuint> const& shaders, const char* what) {
    GLuint p = glCreateProgram_();
    for (GLuint s : shaders) glAttachShader_(p, s);
    glLinkProgram_(p);
    GLint ok = 0; glGetProgramiv_(p, GL_LINK_STATUS, &ok);
    if (!ok) {
        char log[8192]; GLsizei n = 0; glGetProgramInfoLog_(p, sizeof(log), &n, log);
        fprintf(stderr, "gl_harness: link of %s failed:\n%s\n", what, log); exit(2);
    }
    return p;
}
static GLuint makeBuffer(GLenum target, GLuint binding, void const* data, size_t bytes) {
    GLuint b; glGenBuffers_(1, &b);
    glBindBuffer_(target, b);
    glBufferData_(target, GLsizeiptr(bytes ? bytes : 16), nullptr, GL_DYNAMIC_DRAW);
    if (bytes && data) glBufferSubData_(target, 0, GLsizeiptr(bytes), data);
    glBindBufferBase_(target, binding, b);
    return b;
}

static void replaceFirst(std::string& s, std::string const& a, std::string const& b, const char* what) {
    size_t p = s.find(a);
    if (p == std::string::npos) { fprintf(stderr, "gl_harness: anchor not found: %s\n", what); exit(2); }
    s.replace(p, a.size(), b);
}

struct Args {
    std::vector<std::string> v;
    bool has(const char* k) const { for (auto& s : v) if (s == k) return true; return false; }
    const char* get(const char* k, int off = 1) const {
        for (size_t i = 0; i + off < v.size(); ++i) if (v[i] == k) return v[i + off].c_str();
        return nullptr;
    }
};

static int probeMain(Args const& a) {
    initGL(16, 16);
    std::string src = readText(a.get("--shader"));
    std::vector<uint8_t> in = readBin(a.get("--in"));
    int count = atoi(a.get("--count"));
    GLuint prog = link({compile(GL_COMPUTE_SHADER, src, "probe")}, "probe");
    size_t outBytes = size_t(count) * 16;
    GLuint bi = makeBuffer(GL_SHADER_STORAGE_BUFFER, 0, in.data(), in.size());
    GLuint bo = makeBuffer(GL_SHADER_STORAGE_BUFFER, 1, nullptr, outBytes);
    (void)bi;
    glUseProgram_(prog);
    glDispatchCompute_(GLuint(count), 1, 1);
    glMemoryBarrier_(GL_ALL_BARRIER_BITS);
    glFinish_();
    std::vector<uint8_t> out(outBytes);
    glBindBuffer_(GL_SHADER_STORAGE_BUFFER, bo);
    glGetBufferSubData_(GL_SHADER_STORAGE_BUFFER, 0, GLsizeiptr(outBytes), out.data());
    writeBin(a.get("--out"), out.data(), out.size());
    return 0;
}

// The reference's water simulation (Pool/Pool.cpp:224-235): n x (WaterAcceleration, WaterPosition), then WaterPositionSync,
// on the reference's own compute shaders.  --water: gx*gy points of 16 B {z, v, a, blank}; --tris: trianglesOrigin (160 B).
static int waterMain(Args const& a) {
    initGL(16, 16);
    std::string dir = std::string(a.get("--shaders")) + "/";
    std::vector<uint8_t> water = readBin(a.get("--water")), tris = readBin(a.get("--tris"));
    const GLuint groupsX = GLuint(atoi(a.get("--groups", 1))), groupsY = GLuint(atoi(a.get("--groups", 2)));
    const int steps = atoi(a.get("--steps"));
    float para[4] = {float(atof(a.get("--k"))), float(atof(a.get("--dt"))), 0, 0};
    if (water.size() != size_t(groupsX) * 8 * groupsY * 8 * 16) die("water: --water must hold (8 groupsX) x (8 groupsY) points of 16 bytes");
    GLuint acc = link({compile(GL_COMPUTE_SHADER, readText(dir + "WaterAccelerationCompute0.cpp"), "WaterAcceleration")}, "WaterAcceleration");
    GLuint pos = link({compile(GL_COMPUTE_SHADER, readText(dir + "WaterPositionCompute0.cpp"), "WaterPosition")}, "WaterPosition");
    GLuint sync = link({compile(GL_COMPUTE_SHADER, readText(dir + "WaterPositionSyncCompute0.cpp"), "WaterPositionSync")}, "WaterPositionSync");
    makeBuffer(GL_UNIFORM_BUFFER, 4, para, 16);                                            // Pool/Pool.cpp:484 {4, 10}
    GLuint triBuf = makeBuffer(GL_SHADER_STORAGE_BUFFER, 1, tris.data(), tris.size());
    GLuint waterBuf = makeBuffer(GL_SHADER_STORAGE_BUFFER, 10, water.data(), water.size());
    for (int i = 0; i < steps; ++i) {
        glUseProgram_(acc); glDispatchCompute_(groupsX, groupsY, 1); glMemoryBarrier_(GL_SHADER_STORAGE_BARRIER_BIT);
        glUseProgram_(pos); glDispatchCompute_(groupsX, groupsY, 1); glMemoryBarrier_(GL_SHADER_STORAGE_BARRIER_BIT);
    }
    glUseProgram_(sync); glDispatchCompute_(groupsX, groupsY, 1); glMemoryBarrier_(GL_ALL_BARRIER_BITS);
    glFinish_();
    glBindBuffer_(GL_SHADER_STORAGE_BUFFER, waterBuf);
    glGetBufferSubData_(GL_SHADER_STORAGE_BUFFER, 0, GLsizeiptr(water.size()), water.data());
    glBindBuffer_(GL_SHADER_STORAGE_BUFFER, triBuf);
    glGetBufferSubData_(GL_SHADER_STORAGE_BUFFER, 0, GLsizeiptr(tris.size()), tris.data());
    writeBin(a.get("--water-out"), water.data(), water.size());
    writeBin(a.get("--tri-out"), tris.data(), tris.size());
    GLenum e = glGetError_();
    if (e) fprintf(stderr, "gl_harness: GL error 0x%x\n", e);
    return 0;
}

int main(int argc, char** argv) {
    Args a; for (int i = 1; i < argc; ++i) a.v.push_back(argv[i]);
    if (a.v.empty()) die("usage: gl_harness render|probe|water ...");
    if (a.v[0] == "probe") return probeMain(a);
    if (a.v[0] == "water") return waterMain(a);
    if (a.v[0] != "render") die("unknown mode");
    const char* shaderDir = a.get("--shaders");
    const char* scenePath = a.get("--scene");
    const char* transPath = a.get("--trans");
    if (!shaderDir || !scenePath || !transPath || !a.get("--size", 2)) die("render: --shaders --scene --trans --size W H required");
    int W = atoi(a.get("--size", 1)), H = atoi(a.get("--size", 2));
    bool ids = a.has("--ids");
    initGL(W, H);

    // ---- scene dump ----
    std::vector<uint8_t> sc = readBin(scenePath);
    if (sc.size() < 72 || memcmp(sc.data(), "LRLS", 4) != 0) die("bad scene dump");
    uint32_t num[8]; memcpy(num, sc.data() + 8, 32);   // plane, triangle, sphere, circle, cylinder, cone, light, pad
    uint32_t meta[8]; memcpy(meta, sc.data() + 40, 32);
    uint32_t nodeNum = meta[0], texW = meta[1], texH = meta[2], texL = meta[3], cubeSize = meta[4];
    size_t off = 72;
    auto take = [&](size_t bytes) { const uint8_t* p = sc.data() + off; off += bytes; if (off > sc.size()) die("scene dump truncated"); return p; };
    const uint8_t* tris = take(size_t(num[1]) * 160);
    const uint8_t* spheres = take(size_t(num[2]) * 128);
    const uint8_t* circles = take(size_t(num[3]) * 128);
    const uint8_t* cylinders = take(size_t(num[4]) * 128);
    const uint8_t* cones = take(size_t(num[5]) * 128);
    const uint8_t* lights = take(size_t(num[6]) * 32);
    const uint8_t* nodes = take(size_t(nodeNum) * 64);
    const uint8_t* tex = take(size_t(texW) * texH * texL * 3);
    const uint8_t* cube[6];
    for (int i = 0; i < 6; ++i) cube[i] = take(size_t(cubeSize) * cubeSize * 3);

    std::vector<uint8_t> transAll = readBin(transPath);
    if (transAll.size() < 64 || transAll.size() % 64) die("--trans must hold n*64 bytes");
    size_t nSamples = transAll.size() / 64;

    // ---- buffers at the reference's binding points (BVH/BVH.cpp:246,250-252) ----
    uint32_t sizeUBO[4] = {uint32_t(W), uint32_t(H), 0, 0};
    makeBuffer(GL_UNIFORM_BUFFER, 0, sizeUBO, 16);
    GLuint transBuf = makeBuffer(GL_UNIFORM_BUFFER, 1, transAll.data(), 64);
    makeBuffer(GL_UNIFORM_BUFFER, 3, num, 32);
    makeBuffer(GL_SHADER_STORAGE_BUFFER, 0, nullptr, 96);                                   // planes (dead)
    makeBuffer(GL_SHADER_STORAGE_BUFFER, 1, tris, size_t(num[1]) * 160);
    GLuint triGPU = makeBuffer(GL_SHADER_STORAGE_BUFFER, 2, nullptr, size_t(num[1]) * 176);
    makeBuffer(GL_SHADER_STORAGE_BUFFER, 3, spheres, size_t(num[2]) * 128);
    GLuint circleBuf = makeBuffer(GL_SHADER_STORAGE_BUFFER, 4, circles, size_t(num[3]) * 128);
    makeBuffer(GL_SHADER_STORAGE_BUFFER, 5, cylinders, size_t(num[4]) * 128);
    makeBuffer(GL_SHADER_STORAGE_BUFFER, 6, cones, size_t(num[5]) * 128);
    makeBuffer(GL_SHADER_STORAGE_BUFFER, 7, lights, size_t(num[6]) * 32);
    float decay[36]; memset(decay, 0, sizeof(decay));
    for (int i = 0; i < 9; ++i) decay[i * 4 + 3] = 1.0f;
    if (a.has("--decay")) for (int i = 0; i < 4; ++i) decay[32 + i] = float(atof(a.get("--decay", 1 + i)));
    GLuint decayBuf = makeBuffer(GL_SHADER_STORAGE_BUFFER, 8, decay, 144);
    makeBuffer(GL_SHADER_STORAGE_BUFFER, 9, nodes, size_t(nodeNum) * 64);

    // ---- textures (no glTexParameter in the reference: GL defaults; --min-linear pins the filter) ----
    glPixelStorei_(GL_UNPACK_ALIGNMENT, 1);
    GLuint t2d, tcube;
    glGenTextures_(1, &t2d);
    glActiveTexture_(GL_TEXTURE1);
    glBindTexture_(GL_TEXTURE_2D_ARRAY, t2d);
    if (texL) {
        glTexStorage3D_(GL_TEXTURE_2D_ARRAY, 1, GL_RGBA32F, texW, texH, texL);
        glTexSubImage3D_(GL_TEXTURE_2D_ARRAY, 0, 0, 0, 0, texW, texH, texL, GL_BGR, GL_UNSIGNED_BYTE, tex);
    } else {
        uint8_t z[3] = {0, 0, 0};
        glTexStorage3D_(GL_TEXTURE_2D_ARRAY, 1, GL_RGBA32F, 1, 1, 1);
        glTexSubImage3D_(GL_TEXTURE_2D_ARRAY, 0, 0, 0, 0, 1, 1, 1, GL_BGR, GL_UNSIGNED_BYTE, z);
    }
    if (a.has("--min-linear")) glTexParameteri_(GL_TEXTURE_2D_ARRAY, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
    glGenTextures_(1, &tcube);
    glActiveTexture_(GL_TEXTURE2);
    glBindTexture_(GL_TEXTURE_CUBE_MAP, tcube);
    {
        uint32_t cs = cubeSize ? cubeSize : 1;
        uint8_t z[3] = {0, 0, 0};
        glTexStorage2D_(GL_TEXTURE_CUBE_MAP, 1, GL_RGBA32F, cs, cs);
        for (int i = 0; i < 6; ++i)
            glTexSubImage2D_(GL_TEXTURE_CUBE_MAP_POSITIVE_X + i, 0, 0, 0, cs, cs, GL_BGR, GL_UNSIGNED_BYTE, cubeSize ? cube[i] : z);
    }
    if (a.has("--min-linear")) glTexParameteri_(GL_TEXTURE_CUBE_MAP, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
    glEnable_(GL_TEXTURE_CUBE_MAP_SEAMLESS);

    // ---- programs, from the reference's files ----
    std::string dir = std::string(shaderDir) + "/";
    std::string fs = readText(dir + "RendererFragment0.cpp");
    {
        size_t nl = fs.find('\n');
        std::string prelude = ids ? "\nlayout(location = 0) out vec4 lrl_FragColor;\nlayout(location = 1) out uvec4 lrl_Hit;\n#define gl_FragColor lrl_FragColor"
                                  : "\nout vec4 lrl_FragColor;\n#define gl_FragColor lrl_FragColor";
        fs.insert(nl, prelude);
        if (a.get("--depth")) {
            size_t p = fs.find("#define RayTraceDepth");
            if (p == std::string::npos) die("RayTraceDepth define not found");
            size_t e = fs.find('\n', p);
            fs.replace(p, e - p, std::string("#define RayTraceDepth ") + a.get("--depth"));
        }
        if (ids)
            replaceFirst(fs, "\t\tColor tempColor;",
                         "\t\tif (depth == 0) lrl_Hit = uvec4(ray.t < 0 ? uvec2(0) : hitObj, floatBitsToUint(ray.t), 0u);\n\t\tColor tempColor;",
                         "Color tempColor");
    }
    GLuint renderer = link({compile(GL_VERTEX_SHADER, readText(dir + "RendererVertex0.cpp"), "RendererVertex0"),
                            compile(GL_FRAGMENT_SHADER, fs, "RendererFragment0")}, "Renderer");
    GLuint triPre = link({compile(GL_COMPUTE_SHADER, readText(dir + "TrianglePreCompute0.cpp"), "TrianglePre")}, "TrianglePre");
    GLuint circlePre = link({compile(GL_COMPUTE_SHADER, readText(dir + "CirclePreCompute0.cpp"), "CirclePre")}, "CirclePre");

    // ---- tracerInit.run() (BVH/BVH.cpp:157-174) ----
    auto t0 = std::chrono::steady_clock::now();
    if (num[1]) { glUseProgram_(triPre); glDispatchCompute_((num[1] + 1023) / 1024, 1, 1); }
    if (num[3]) { glUseProgram_(circlePre); glDispatchCompute_((num[3] + 1023) / 1024, 1, 1); }
    glMemoryBarrier_(GL_SHADER_STORAGE_BARRIER_BIT);
    if (a.has("--decay-compute")) {
        GLuint dPre = link({compile(GL_COMPUTE_SHADER, readText(dir + "DecayOriginPreCompute0.cpp"), "DecayOriginPre")}, "DecayOriginPre");
        GLuint dO = link({compile(GL_COMPUTE_SHADER, readText(dir + "DecayOriginCompute0.cpp"), "DecayOrigin")}, "DecayOrigin");
        glUseProgram_(dPre); glDispatchCompute_(8, 1, 1);
        glMemoryBarrier_(GL_SHADER_STORAGE_BARRIER_BIT);
        glUseProgram_(dO); glDispatchCompute_(1, 1, 1);
        glMemoryBarrier_(GL_SHADER_STORAGE_BARRIER_BIT);
    }
    glFinish_();
    double preMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

    if (a.get("--tri-out")) {
        std::vector<uint8_t> o(size_t(num[1]) * 176);
        glBindBuffer_(GL_SHADER_STORAGE_BUFFER, triGPU);
        if (!o.empty()) glGetBufferSubData_(GL_SHADER_STORAGE_BUFFER, 0, GLsizeiptr(o.size()), o.data());
        writeBin(a.get("--tri-out"), o.data(), o.size());
    }
    if (a.get("--circle-out")) {
        std::vector<uint8_t> o(size_t(num[3]) * 128);
        glBindBuffer_(GL_SHADER_STORAGE_BUFFER, circleBuf);
        if (!o.empty()) glGetBufferSubData_(GL_SHADER_STORAGE_BUFFER, 0, GLsizeiptr(o.size()), o.data());
        writeBin(a.get("--circle-out"), o.data(), o.size());
    }
    if (a.get("--decay-out")) {
        float o[36];
        glBindBuffer_(GL_SHADER_STORAGE_BUFFER, decayBuf);
        glGetBufferSubData_(GL_SHADER_STORAGE_BUFFER, 0, 144, o);
        writeBin(a.get("--decay-out"), o, 144);
    }
    if (a.has("--no-render")) return 0;

    // ---- render target + full-screen quad (BVH/BVH.cpp:17-43) ----
    GLuint fbo, colorTex, idTex = 0;
    glActiveTexture_(GL_TEXTURE0);
    glGenTextures_(1, &colorTex);
    glBindTexture_(GL_TEXTURE_2D, colorTex);
    glTexStorage2D_(GL_TEXTURE_2D, 1, GL_RGBA32F, W, H);
    glGenFramebuffers_(1, &fbo);
    glBindFramebuffer_(GL_FRAMEBUFFER, fbo);
    glFramebufferTexture2D_(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, colorTex, 0);
    if (ids) {
        glGenTextures_(1, &idTex);
        glBindTexture_(GL_TEXTURE_2D, idTex);
        glTexStorage2D_(GL_TEXTURE_2D, 1, GL_RGBA32UI, W, H);
        glFramebufferTexture2D_(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT1, GL_TEXTURE_2D, idTex, 0);
        GLenum bufs[2] = {GL_COLOR_ATTACHMENT0, GL_COLOR_ATTACHMENT1};
        glDrawBuffers_(2, bufs);
    }
    if (glCheckFramebufferStatus_(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) die("FBO incomplete");
    GLuint vao, vbo;
    float quad[8] = {-1, -1, 1, -1, 1, 1, -1, 1};
    glGenVertexArrays_(1, &vao); glBindVertexArray_(vao);
    glGenBuffers_(1, &vbo); glBindBuffer_(GL_ARRAY_BUFFER, vbo);
    glBufferData_(GL_ARRAY_BUFFER, sizeof(quad), quad, GL_STATIC_DRAW);
    glEnableVertexAttribArray_(0);
    glVertexAttribPointer_(0, 2, GL_FLOAT, GL_FALSE, 8, nullptr);
    glViewport_(0, 0, W, H);
    glUseProgram_(renderer);

    std::vector<float> acc(size_t(W) * H * 4, 0.0f), img(size_t(W) * H * 4);
    std::vector<uint32_t> idImg;
    double renderMs = 0;
    for (size_t s = 0; s < nSamples; ++s) {
        glBindBuffer_(GL_UNIFORM_BUFFER, transBuf);
        glBufferSubData_(GL_UNIFORM_BUFFER, 0, 64, transAll.data() + s * 64);
        auto r0 = std::chrono::steady_clock::now();
        glClearColor_(0, 0, 0, 0);
        glClear_(GL_COLOR_BUFFER_BIT);
        glDrawArrays_(GL_TRIANGLE_FAN, 0, 4);
        glFinish_();
        renderMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r0).count();
        glReadBuffer_(GL_COLOR_ATTACHMENT0);
        glReadPixels_(0, 0, W, H, GL_RGBA, GL_FLOAT, img.data());
        if (nSamples == 1) acc = img;
        else for (size_t i = 0; i < acc.size(); ++i) acc[i] += img[i];
        if (ids && s == 0) {
            idImg.resize(size_t(W) * H * 4);
            glReadBuffer_(GL_COLOR_ATTACHMENT1);
            glReadPixels_(0, 0, W, H, GL_RGBA_INTEGER, GL_UNSIGNED_INT, idImg.data());
        }
    }
    if (nSamples > 1) { float inv = float(nSamples); for (float& v : acc) v /= inv; }
    GLenum e = glGetError_();
    if (e) fprintf(stderr, "gl_harness: GL error 0x%x\n", e);
    if (a.get("--out")) writeBin(a.get("--out"), acc.data(), acc.size() * 4);
    if (ids) writeBin(a.get("--ids"), idImg.data(), idImg.size() * 4);
    printf("{\"pre_ms\": %.3f, \"render_ms\": %.3f, \"samples\": %zu, \"width\": %d, \"height\": %d}\n", preMs, renderMs, nSamples, W, H);
    return 0;
}
