// Binary STL reader/writer.  Stand-in for the reference's external <_STL.h> (absent); use sites
// BVH/BVH.cpp:222-232,255-265 (`STL`, `.triangles.length`, `readSTL()`), RenderSTL/RenderSTL.cpp:68.
// File format: 80-byte header, u32 count, then 50-byte records {normal[3], v1[3], v2[3], v3[3], u16 attr}
// (all 27 STL files shipped with the reference are binary).  The in-memory triangle keeps the file's
// 50-byte packing, like the reference's (BVH.exe@0x140008c30).
#pragma once
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "Math.hpp"
#include "Vector.hpp"

namespace lrl {

struct STL {
#pragma pack(push, 1)
    struct Triangle {
        float normal[3];
        float vertices[3][3];
        uint16_t attrib;
    };
#pragma pack(pop)
    static_assert(sizeof(Triangle) == 50, "STL record");

    std::string name;
    Vector<Triangle> triangles;

    STL() = default;
    explicit STL(std::string const& path) { *this = readSTL(path); }

    static STL readSTL(std::string const& path) {
        STL s;
        FILE* f = std::fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("STL: cannot open " + path);
        char header[80];
        uint32_t n = 0;
        if (std::fread(header, 1, 80, f) != 80 || std::fread(&n, 4, 1, f) != 1) {
            std::fclose(f);
            throw std::runtime_error("STL: truncated header in " + path);
        }
        std::fseek(f, 0, SEEK_END);
        long sz = std::ftell(f);
        if (sz < 0 || uint64_t(sz) < 84ull + 50ull * n) {
            std::fclose(f);
            throw std::runtime_error("STL: truncated body (not a binary STL?) in " + path);
        }
        std::fseek(f, 84, SEEK_SET);
        s.name = path;
        s.triangles.reserve(n);
        std::vector<Triangle> buf(n);
        if (n && std::fread(buf.data(), 50, n, f) != n) {
            std::fclose(f);
            throw std::runtime_error("STL: short read in " + path);
        }
        std::fclose(f);
        for (uint32_t i = 0; i < n; ++i) s.triangles.pushBack(buf[i]);
        return s;
    }

    void writeSTL(std::string const& path) const {
        FILE* f = std::fopen(path.c_str(), "wb");
        if (!f) throw std::runtime_error("STL: cannot write " + path);
        char header[80] = "lrltrace binary STL";
        uint32_t n = triangles.length;
        std::fwrite(header, 1, 80, f);
        std::fwrite(&n, 4, 1, f);
        if (n) std::fwrite(triangles.data, 50, n, f);
        std::fclose(f);
    }

    // 1 -> 4 midpoint subdivision (deterministic; SURVEY 8d C2: bunny x 2 levels = 1 240 960 triangles).
    STL subdivided() const {
        STL o;
        o.name = name + "+sub";
        o.triangles.reserve(triangles.length * 4);
        for (unsigned i = 0; i < triangles.length; ++i) {
            Triangle const& t = triangles[i];
            float m[3][3];
            for (int e = 0; e < 3; ++e)
                for (int k = 0; k < 3; ++k) m[e][k] = 0.5f * (t.vertices[e][k] + t.vertices[(e + 1) % 3][k]);
            auto emit = [&](float const* a, float const* b, float const* c) {
                Triangle q = t;
                for (int k = 0; k < 3; ++k) { q.vertices[0][k] = a[k]; q.vertices[1][k] = b[k]; q.vertices[2][k] = c[k]; }
                o.triangles.pushBack(q);
            };
            emit(t.vertices[0], m[0], m[2]);
            emit(m[0], t.vertices[1], m[1]);
            emit(m[2], m[1], t.vertices[2]);
            emit(m[0], m[1], m[2]);
        }
        return o;
    }
};

}  // namespace lrl
