// Frame dumps besides the 24-bit BMP ("PPM/EXR dump"): binary PPM (P6, 8 bits, what the reference's RGBA8 window
// shows), PFM (PF, 32-bit float RGB) and OpenEXR (scan lines, 32-bit float A B G R channels, no compression: a file every EXR reader
// opens, without a compressor this tree has no other use for): the unclamped values lrl_render writes, the high-dynamic-range dumps.
// All take the tracer's row order, row 0 = bottom (gl_FragCoord): PFM stores rows bottom-up as they come, PPM and EXR top-down.
#pragma once
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

namespace lrl {

// rgba8: width * height * 4 bytes (lrl_quantize_rgba8), row 0 = bottom
inline void writePPM(std::string const& path, int width, int height, const uint8_t* rgba8) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("PPM: cannot write " + path);
    std::fprintf(f, "P6\n%d %d\n255\n", width, height);
    std::vector<uint8_t> row(size_t(width) * 3);
    for (int y = height - 1; y >= 0; --y) {
        const uint8_t* src = rgba8 + size_t(y) * width * 4;
        for (int x = 0; x < width; ++x) { row[size_t(x) * 3] = src[size_t(x) * 4]; row[size_t(x) * 3 + 1] = src[size_t(x) * 4 + 1]; row[size_t(x) * 3 + 2] = src[size_t(x) * 4 + 2]; }
        std::fwrite(row.data(), 1, row.size(), f);
    }
    std::fclose(f);
}

// rgba: width * height * 4 floats (lrl_render's output), row 0 = bottom; little-endian (scale -1.0)
inline void writePFM(std::string const& path, int width, int height, const float* rgba) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("PFM: cannot write " + path);
    std::fprintf(f, "PF\n%d %d\n-1.0\n", width, height);
    std::vector<float> row(size_t(width) * 3);
    for (int y = 0; y < height; ++y) {
        const float* src = rgba + size_t(y) * width * 4;
        for (int x = 0; x < width; ++x) { row[size_t(x) * 3] = src[size_t(x) * 4]; row[size_t(x) * 3 + 1] = src[size_t(x) * 4 + 1]; row[size_t(x) * 3 + 2] = src[size_t(x) * 4 + 2]; }
        std::fwrite(row.data(), sizeof(float), row.size(), f);
    }
    std::fclose(f);
}

// rgba: width * height * 4 floats (lrl_render's output), row 0 = bottom.  OpenEXR 2, single part, scan lines in increasing y
// (y = 0 is the top row), one line per chunk, channels in alphabetical order as the format demands; little-endian throughout.
inline void writeEXR(std::string const& path, int width, int height, const float* rgba) {
    if (width <= 0 || height <= 0) throw std::runtime_error("EXR: empty image");
    std::vector<uint8_t> h;
    auto bytes = [&](const void* p, size_t n) { h.insert(h.end(), static_cast<const uint8_t*>(p), static_cast<const uint8_t*>(p) + n); };
    auto str = [&](const char* s) { bytes(s, std::char_traits<char>::length(s) + 1); };
    auto i32 = [&](int32_t v) { bytes(&v, 4); };
    auto f32 = [&](float v) { bytes(&v, 4); };
    auto attr = [&](const char* name, const char* type, int32_t size) { str(name); str(type); i32(size); };
    const uint8_t magic[8] = {0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0};
    bytes(magic, 8);
    attr("channels", "chlist", 4 * 18 + 1);
    for (const char* c : {"A", "B", "G", "R"}) { str(c); i32(2); const uint8_t lin[4] = {0, 0, 0, 0}; bytes(lin, 4); i32(1); i32(1); }      // FLOAT, sampling 1 x 1
    h.push_back(0);
    attr("compression", "compression", 1); h.push_back(0);
    attr("dataWindow", "box2i", 16); i32(0); i32(0); i32(width - 1); i32(height - 1);
    attr("displayWindow", "box2i", 16); i32(0); i32(0); i32(width - 1); i32(height - 1);
    attr("lineOrder", "lineOrder", 1); h.push_back(0);
    attr("pixelAspectRatio", "float", 4); f32(1.0f);
    attr("screenWindowCenter", "v2f", 8); f32(0.0f); f32(0.0f);
    attr("screenWindowWidth", "float", 4); f32(1.0f);
    h.push_back(0);
    const uint64_t line = 8 + uint64_t(width) * 16, first = h.size() + uint64_t(height) * 8;
    for (int y = 0; y < height; ++y) { const uint64_t at = first + uint64_t(y) * line; bytes(&at, 8); }
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("EXR: cannot write " + path);
    std::fwrite(h.data(), 1, h.size(), f);
    std::vector<float> row(size_t(width) * 4);
    static const int channelOf[4] = {3, 2, 1, 0};      // A B G R out of r g b a
    for (int y = 0; y < height; ++y) {
        const float* src = rgba + size_t(height - 1 - y) * width * 4;
        for (int c = 0; c < 4; ++c)
            for (int x = 0; x < width; ++x) row[size_t(c) * width + x] = src[size_t(x) * 4 + channelOf[c]];
        const int32_t head[2] = {y, int32_t(width * 16)};
        std::fwrite(head, 4, 2, f);
        std::fwrite(row.data(), sizeof(float), row.size(), f);
    }
    std::fclose(f);
}

}  // namespace lrl
