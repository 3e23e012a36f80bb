// image_action.cpp — see image_action.h.
#include "image_action.h"
#include <fstream>
#include <vector>
#include "error.h"
#include "image_io.h"

namespace rthost {

static uint32_t f64_as_u32(double x) { // Rust `as u32`
    if (!(x > 0.0)) return 0u;         // NaN and negatives
    if (x >= 4294967295.0) return 4294967295u;
    return (uint32_t)x;
}

void pack_rgba8(const double *rgb, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; ++i) {
        uint32_t red = f64_as_u32(rgb[3 * i] * 255.0);
        uint32_t green = f64_as_u32(rgb[3 * i + 1] * 255.0);
        uint32_t blue = f64_as_u32(rgb[3 * i + 2] * 255.0);
        uint32_t word = (red << 24) | (green << 16) | (blue << 8) | 255u;
        out[4 * i] = (uint8_t)(word >> 24);
        out[4 * i + 1] = (uint8_t)(word >> 16);
        out[4 * i + 2] = (uint8_t)(word >> 8);
        out[4 * i + 3] = (uint8_t)word;
    }
}

std::string save_png(const double *rgb, int width, int height, const std::string &dir) {
    size_t n = (size_t)width * (size_t)height;
    std::vector<uint8_t> rgba(n * 4);
    pack_rgba8(rgb, n, rgba.data());
    std::string path = dir;
    if (!path.empty() && path.back() != '/') path += '/';
    path += sha256_hex_upper(rgba.data(), rgba.size()) + ".png";
    std::vector<uint8_t> file;
    std::string why;
    if (!encode_png_rgba8(rgba.data(), width, height, file, why)) throw TracerError::ImageSave(why);
    std::ofstream f(path, std::ios::binary);
    if (!f) throw TracerError::ImageSave("cannot create " + path);
    f.write(reinterpret_cast<const char *>(file.data()), (std::streamsize)file.size());
    if (!f) throw TracerError::ImageSave("short write to " + path);
    return path;
}

} // namespace rthost
