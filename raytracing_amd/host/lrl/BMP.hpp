// 24-bit BMP reader/writer.  Stand-in for the reference's external <_BMP.h> (absent); use sites
// BVH/BVH.cpp:220-221,253-254,266-274 (`BMPData`, `BMPCubeData`, `.bmp.header.width/height`),
// Pool/Pool.cpp:446-450,473-494.  Payload is kept exactly as stored in the file: BGR bytes, rows
// bottom-up, row stride padded to 4 bytes is removed on load (the reference uploads it to GL as
// GL_BGR/GL_UNSIGNED_BYTE without flipping — SURVEY §2.3 "Textures").
// Cube-face file order front, back, down, up, right, left -> +X,-X,+Y,-Y,+Z,-Z (BVH.exe@0x140008590).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace lrl {

struct BMP {
    struct Header { int32_t width = 0, height = 0; } header;
    std::vector<uint8_t> data;  // BGR, bottom-up, tightly packed (width*3 per row)

    static BMP read(std::string const& path) {
        BMP b;
        FILE* f = std::fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("BMP: cannot open " + path);
        uint8_t h[54];
        if (std::fread(h, 1, 54, f) != 54 || h[0] != 'B' || h[1] != 'M') {
            std::fclose(f);
            throw std::runtime_error("BMP: bad header in " + path);
        }
        uint32_t off; int32_t w, hh; uint16_t bpp; uint32_t comp;
        std::memcpy(&off, h + 10, 4); std::memcpy(&w, h + 18, 4); std::memcpy(&hh, h + 22, 4);
        std::memcpy(&bpp, h + 28, 2); std::memcpy(&comp, h + 30, 4);
        if (bpp != 24 || comp != 0 || w <= 0 || hh == 0) {
            std::fclose(f);
            throw std::runtime_error("BMP: only uncompressed 24-bit supported: " + path);
        }
        bool topDown = hh < 0;
        int32_t H = topDown ? -hh : hh;
        size_t stride = (size_t(w) * 3 + 3) & ~size_t(3);
        b.header.width = w; b.header.height = H;
        b.data.resize(size_t(w) * 3 * H);
        std::vector<uint8_t> row(stride);
        std::fseek(f, long(off), SEEK_SET);
        for (int32_t y = 0; y < H; ++y) {
            if (std::fread(row.data(), 1, stride, f) != stride) {
                std::fclose(f);
                throw std::runtime_error("BMP: short read in " + path);
            }
            int32_t dst = topDown ? (H - 1 - y) : y;
            std::memcpy(b.data.data() + size_t(dst) * w * 3, row.data(), size_t(w) * 3);
        }
        std::fclose(f);
        return b;
    }

    void write(std::string const& path) const {
        FILE* f = std::fopen(path.c_str(), "wb");
        if (!f) throw std::runtime_error("BMP: cannot write " + path);
        size_t stride = (size_t(header.width) * 3 + 3) & ~size_t(3);
        uint32_t imgSize = uint32_t(stride * header.height), fileSize = 54 + imgSize, off = 54, hs = 40;
        uint16_t planes = 1, bpp = 24;
        uint8_t h[54] = {0};
        h[0] = 'B'; h[1] = 'M';
        std::memcpy(h + 2, &fileSize, 4); std::memcpy(h + 10, &off, 4); std::memcpy(h + 14, &hs, 4);
        std::memcpy(h + 18, &header.width, 4); std::memcpy(h + 22, &header.height, 4);
        std::memcpy(h + 26, &planes, 2); std::memcpy(h + 28, &bpp, 2); std::memcpy(h + 34, &imgSize, 4);
        std::fwrite(h, 1, 54, f);
        std::vector<uint8_t> row(stride, 0);
        for (int32_t y = 0; y < header.height; ++y) {
            std::memcpy(row.data(), data.data() + size_t(y) * header.width * 3, size_t(header.width) * 3);
            std::fwrite(row.data(), 1, stride, f);
        }
        std::fclose(f);
    }
};

struct BMPData {
    BMP bmp;
    BMPData() = default;
    explicit BMPData(std::string const& path) : bmp(BMP::read(path)) {}
};

struct BMPCubeData {
    BMP bmp[6];  // file order front, back, down, up, right, left == GL faces +X,-X,+Y,-Y,+Z,-Z
    BMPCubeData() = default;
    explicit BMPCubeData(std::string const& dir) {
        static const char* names[6] = {"front.bmp", "back.bmp", "down.bmp", "up.bmp", "right.bmp", "left.bmp"};
        for (int i = 0; i < 6; ++i) bmp[i] = BMP::read(dir + names[i]);
        for (int i = 1; i < 6; ++i)
            if (bmp[i].header.width != bmp[0].header.width || bmp[i].header.height != bmp[0].header.height)
                throw std::runtime_error("BMPCubeData: faces differ in size under " + dir);
        if (bmp[0].header.width != bmp[0].header.height) throw std::runtime_error("BMPCubeData: faces must be square");
    }
};

}  // namespace lrl
