// Frame dumps besides the 24-bit BMP ("PPM/EXR dump"): binary PPM (P6, 8 bits, what the reference's RGBA8 window
// shows) and PFM (PF, 32-bit float RGB, the unclamped values lrl_render writes: the high-dynamic-range dump; EXR would need a
// compressor this tree has no use for).  Both take the tracer's row order, row 0 = bottom (gl_FragCoord): PFM stores rows
// bottom-up as they come, PPM top-down.
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

}  // namespace lrl
