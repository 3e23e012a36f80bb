// image_action.h — SavePng (racer-tracer/src/image_action/png.rs).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

namespace rthost {

// png.rs:21-31: per channel `(v * 255.0) as u32` (saturating, NaN -> 0), then
// `(red << 24) | green << 16 | blue << 8 | 255` as big-endian bytes.  Channels
// above 255 wrap (red) or spill into the next-higher channel, as in the
// reference.
void pack_rgba8(const double *rgb, size_t n_pixels, uint8_t *out_rgba);

// png.rs:33-55: file name = upper-hex SHA-256 of the RGBA bytes + ".png"
// inside `dir`.  Returns the path; throws TracerError::ImageSave.
std::string save_png(const double *rgb, int width, int height, const std::string &dir);

} // namespace rthost
