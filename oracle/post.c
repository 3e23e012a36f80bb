/* post.c — what happens to a tile after the renderer emitted it:
 * tone map (image_buffer.rs:135-170 calling tone_map/{aces,hable,reinhard,none}.rs) and the RGBA8
 * packing of SavePng (image_action/png.rs:19-31).
 *
 * TEST INFRASTRUCTURE (see oracle.h).
 */
#include "oracle.h"
#include <math.h>
#include <string.h>

/* tone_map.rs:18-66 defaults */
void orc_tone_map_defaults(int kind, OrcToneMap *out) {
    static const double aces_in[9] = { 0.59719, 0.35458, 0.04823, 0.07600, 0.90834, 0.01566,
                                       0.02840, 0.13383, 0.83777 };
    static const double aces_out[9] = { 1.60475, -0.53108, -0.07367, -0.10208, 1.10813, -0.00605,
                                        -0.00327, -0.07276, 1.07602 };
    memset(out, 0, sizeof *out);
    out->kind = kind;
    out->max_white = 25.0;
    out->hable[0] = 0.15; out->hable[1] = 0.5; out->hable[2] = 0.1;
    out->hable[3] = 0.2; out->hable[4] = 0.02; out->hable[5] = 0.3;
    out->exposure_bias = 2.0;
    out->linear_white = 11.2;
    memcpy(out->aces_in, aces_in, sizeof aces_in);
    memcpy(out->aces_out, aces_out, sizeof aces_out);
}

/* aces.rs:18-23 */
static void mat_mul(const double m[9], const double c[3], double out[3]) {
    out[0] = m[0] * c[0] + m[1] * c[1] + m[2] * c[2];
    out[1] = m[3] * c[0] + m[4] * c[1] + m[5] * c[2];
    out[2] = m[6] * c[0] + m[7] * c[1] + m[8] * c[2];
}

/* hable.rs:52-62 */
static double hable_partial(double color, const double d[6], double toe_angle) {
    double a = d[0], b = d[1], c = d[2], dd = d[3], e = d[4], f = d[5];
    return ((color * (a * color + c * b) + dd * e) / (color * (a * color + b) + dd * f)) - toe_angle;
}

void orc_tone_map_apply(const OrcToneMap *tm, const double *in, double *out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        const double *c = in + 3 * i;
        double *o = out + 3 * i;
        switch (tm->kind) {
        case ORC_TM_REINHARD: { /* reinhard.rs:16-41; max_white_pow = max_white^2 */
            double mwp = tm->max_white * tm->max_white;
            double l_old = c[0] * 0.2126 + c[1] * 0.7152 + c[2] * 0.0722;
            double numerator = l_old * (1.0 + (l_old / mwp));
            double l_new = numerator / (1.0 + l_old);
            double s = l_new / l_old; /* change_luminance: color * (l_new / l_old) */
            o[0] = c[0] * s; o[1] = c[1] * s; o[2] = c[2] * s;
            break;
        }
        case ORC_TM_HABLE: { /* hable.rs:38-80 */
            double toe_angle = tm->hable[4] / tm->hable[5];
            double white_scale = 1.0 / hable_partial(tm->linear_white, tm->hable, toe_angle);
            for (int k = 0; k < 3; ++k)
                o[k] = hable_partial(c[k] * tm->exposure_bias, tm->hable, toe_angle) * white_scale;
            break;
        }
        case ORC_TM_ACES: { /* aces.rs:25-55 */
            double v[3], fit[3];
            mat_mul(tm->aces_in, c, v);
            for (int k = 0; k < 3; ++k) {
                double a = v[k] * (v[k] + 0.0245786) - 0.000090537;
                double b = v[k] * (0.983729 * v[k] + 0.4329510) + 0.238081;
                fit[k] = a / b;
            }
            mat_mul(tm->aces_out, fit, o);
            break;
        }
        default: /* none.rs */
            o[0] = c[0]; o[1] = c[1]; o[2] = c[2];
            break;
        }
    }
}

/* Rust `f64 as u32`: saturating, NaN -> 0 */
static uint32_t f64_as_u32(double x) {
    if (x != x || x <= 0.0) return 0u;
    if (x >= 4294967295.0) return 4294967295u;
    return (uint32_t)x;
}

/* png.rs:21-31: (red << 24) | green << 16 | blue << 8 | 255, big-endian bytes.
 * Shifts are on u32, so a channel above 255 loses its high bits (red) or
 * ORs them into the next-higher channel (green, blue). */
void orc_pack_rgba8(const double *rgb, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; ++i) {
        uint32_t red = f64_as_u32(rgb[3 * i + 0] * 255.0);
        uint32_t green = f64_as_u32(rgb[3 * i + 1] * 255.0);
        uint32_t blue = f64_as_u32(rgb[3 * i + 2] * 255.0);
        uint32_t word = (red << 24) | (green << 16) | (blue << 8) | 255u;
        out[4 * i + 0] = (uint8_t)(word >> 24);
        out[4 * i + 1] = (uint8_t)(word >> 16);
        out[4 * i + 2] = (uint8_t)(word >> 8);
        out[4 * i + 3] = (uint8_t)word;
    }
}
