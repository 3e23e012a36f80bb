// rt_post_kernel.hip — tone map + RGBA8 packing of a finished frame on the
// device: ScreenBuffer::update (racer-tracer/src/image_buffer.rs:147-153,
// tone_map/{none,reinhard,hable,aces}.rs) fused with SavePng's packing
// (image_action/png.rs:21-31).
//
// HBM-bound streaming kernel: 24 B read + 4 B written per pixel (28 B; 52 B
// with the optional tone-mapped float output).  Floating-point contraction is
// OFF in this file so every operation is the same IEEE f64 operation the host
// implementation (host/tone_map.cpp, host/image_action.cpp) performs, which
// makes the packed bytes equal on both sides.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rt_abi.h"

#pragma clang fp contract(off)

namespace rtdev {

struct c3 {
    double x, y, z;
};

__device__ __forceinline__ c3 mat_mul(const double *m, c3 c) { // aces.rs:18-23
    return c3{m[0] * c.x + m[1] * c.y + m[2] * c.z, m[3] * c.x + m[4] * c.y + m[5] * c.z,
              m[6] * c.x + m[7] * c.y + m[8] * c.z};
}

__device__ __forceinline__ double hable_partial(double color, const double *d, double toe_angle) { // hable.rs:52-62
    return ((color * (d[0] * color + d[2] * d[1]) + d[3] * d[4]) / (color * (d[0] * color + d[1]) + d[3] * d[5])) - toe_angle;
}

__device__ __forceinline__ c3 tone_map(const RtToneMap &tm, c3 c) {
    switch (tm.kind) {
    case RT_TM_REINHARD: { // reinhard.rs:16-41
        const double mwp = tm.max_white * tm.max_white;
        const double l_old = c.x * 0.2126 + c.y * 0.7152 + c.z * 0.0722;
        const double numerator = l_old * (1.0 + (l_old / mwp));
        const double l_new = numerator / (1.0 + l_old);
        const double s = l_new / l_old;
        return c3{c.x * s, c.y * s, c.z * s};
    }
    case RT_TM_HABLE: { // hable.rs:38-80
        const double toe_angle = tm.hable[4] / tm.hable[5];
        const double white_scale = 1.0 / hable_partial(tm.linear_white, tm.hable, toe_angle);
        return c3{hable_partial(c.x * tm.exposure_bias, tm.hable, toe_angle) * white_scale,
                  hable_partial(c.y * tm.exposure_bias, tm.hable, toe_angle) * white_scale,
                  hable_partial(c.z * tm.exposure_bias, tm.hable, toe_angle) * white_scale};
    }
    case RT_TM_ACES: { // aces.rs:25-55
        const c3 v = mat_mul(tm.aces_in, c);
        const c3 a = c3{v.x * (v.x + 0.0245786) - 0.000090537, v.y * (v.y + 0.0245786) - 0.000090537,
                        v.z * (v.z + 0.0245786) - 0.000090537};
        const c3 b = c3{v.x * (0.983729 * v.x + 0.4329510) + 0.238081, v.y * (0.983729 * v.y + 0.4329510) + 0.238081,
                        v.z * (0.983729 * v.z + 0.4329510) + 0.238081};
        return mat_mul(tm.aces_out, c3{a.x / b.x, a.y / b.y, a.z / b.z});
    }
    default: return c; // none.rs
    }
}

__device__ __forceinline__ uint32_t f64_as_u32(double x) { // Rust `as u32`: saturating, NaN -> 0
    if (!(x > 0.0)) return 0u;
    if (x >= 4294967295.0) return 4294967295u;
    return (uint32_t)x;
}

__global__ __launch_bounds__(256) void k_post_rgba8(const RtToneMap tm, const double *__restrict__ rgb,
                                                    size_t n_pixels, uint32_t *__restrict__ rgba,
                                                    double *__restrict__ mapped) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pixels; i += (size_t)gridDim.x * blockDim.x) {
        const c3 c = tone_map(tm, c3{rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]});
        if (mapped) {
            mapped[3 * i] = c.x;
            mapped[3 * i + 1] = c.y;
            mapped[3 * i + 2] = c.z;
        }
        const uint32_t red = f64_as_u32(c.x * 255.0), green = f64_as_u32(c.y * 255.0), blue = f64_as_u32(c.z * 255.0);
        const uint32_t word = (red << 24) | (green << 16) | (blue << 8) | 255u; // png.rs:24-28
        rgba[i] = __builtin_bswap32(word);                                       // to_be_bytes on a little-endian store
    }
}

} // namespace rtdev

extern "C" hipError_t rtdev_launch_post_rgba8(const RtToneMap *tm, const double *rgb, size_t n_pixels, uint8_t *rgba,
                                              double *mapped, hipStream_t stream) {
    if (n_pixels == 0) return hipSuccess;
    size_t want = (n_pixels + 255) / 256;
    unsigned blocks = (unsigned)(want > 4096 ? 4096 : want);
    hipLaunchKernelGGL(rtdev::k_post_rgba8, dim3(blocks), dim3(256), 0, stream, *tm, rgb, n_pixels,
                       reinterpret_cast<uint32_t *>(rgba), mapped);
    return hipGetLastError();
}
