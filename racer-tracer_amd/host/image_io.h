// image_io.h — the small amount of image I/O the path's neighbours need:
// decode a texture file to RGBA8 (what `image::open(..).into_rgba8()` gives
// the reference, texture/image.rs:18-24), encode an RGBA8 PNG and SHA-256 its
// pixels (image_action/png.rs:33-47).  The `image 0.24.5` and `sha2 0.10.6`
// crates are not in /root/reference; JPEG/PNG/SHA-256 are public standards
// (ITU T.81, RFC 2083, FIPS 180-4) and are implemented from those.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace rthost {

bool file_exists(const std::string &path);
bool read_file(const std::string &path, std::vector<uint8_t> &out);

// Baseline/extended-sequential Huffman JPEG (8-bit, 1 or 3 components) and
// non-interlaced 8-bit PNG.  Returns false with `why` set on failure.
bool decode_image_rgba8(const std::string &path, std::vector<uint8_t> &rgba, int &width, int &height,
                        std::string &why);
bool decode_jpeg_rgba8(const std::vector<uint8_t> &file, std::vector<uint8_t> &rgba, int &width, int &height,
                       std::string &why);
bool decode_png_rgba8(const std::vector<uint8_t> &file, std::vector<uint8_t> &rgba, int &width, int &height,
                      std::string &why);

// RGBA8 -> PNG file bytes (colour type 6, zlib-compressed).
bool encode_png_rgba8(const uint8_t *rgba, int width, int height, std::vector<uint8_t> &out, std::string &why);

// SHA-256 as upper-case hex, the `{:X}` of png.rs:39.
std::string sha256_hex_upper(const uint8_t *data, size_t len);

} // namespace rthost
