// tone_map.cpp — see tone_map.h.
#include "tone_map.h"
#include <cstring>

namespace rthost {

Color Reinhard::tone_map(const Color &color) const { // reinhard.rs:16-41
    double l_old = color.dot(Color(0.2126, 0.7152, 0.0722));
    double numerator = l_old * (1.0 + (l_old / max_white_pow));
    double l_new = numerator / (1.0 + l_old);
    return color * (l_new / l_old);
}

double Hable::partial(double color, const HableData &d, double toe_angle) { // hable.rs:52-62
    double a = d.shoulder_strength, b = d.linear_strength, c = d.linear_angle;
    double dd = d.toe_strength, e = d.toe_numerator, f = d.toe_denominator;
    return ((color * (a * color + c * b) + dd * e) / (color * (a * color + b) + dd * f)) - toe_angle;
}

Hable::Hable(HableData d, double bias, double white_point) // hable.rs:41-50
    : data(d), toe_angle(d.toe_numerator / d.toe_denominator), exposure_bias(bias), linear_white_point(white_point),
      white_scale(1.0 / partial(white_point, d, d.toe_numerator / d.toe_denominator)) {}

static RtToneMap blank_tone_map(int kind) {
    RtToneMap t;
    memset(&t, 0, sizeof t);
    t.kind = kind;
    return t;
}
RtToneMap ToneMapNone::describe() const { return blank_tone_map(RT_TM_NONE); }
RtToneMap Reinhard::describe() const {
    RtToneMap t = blank_tone_map(RT_TM_REINHARD);
    t.max_white = max_white;
    return t;
}
RtToneMap Hable::describe() const {
    RtToneMap t = blank_tone_map(RT_TM_HABLE);
    const double d[6] = {data.shoulder_strength, data.linear_strength, data.linear_angle,
                         data.toe_strength, data.toe_numerator, data.toe_denominator};
    memcpy(t.hable, d, sizeof d);
    t.exposure_bias = exposure_bias;
    t.linear_white = linear_white_point;
    return t;
}
RtToneMap Aces::describe() const {
    RtToneMap t = blank_tone_map(RT_TM_ACES);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            t.aces_in[3 * r + c] = input_matrix[(size_t)r][c];
            t.aces_out[3 * r + c] = output_matrix[(size_t)r][c];
        }
    return t;
}

Color Hable::tone_map(const Color &color) const { // hable.rs:72-80
    Color c = color * exposure_bias;
    return Color(partial(c.x(), data, toe_angle) * white_scale, partial(c.y(), data, toe_angle) * white_scale,
                 partial(c.z(), data, toe_angle) * white_scale);
}

static Color mat_mul(const std::array<Color, 3> &m, const Color &c) { // aces.rs:18-23
    return Color(m[0][0] * c[0] + m[0][1] * c[1] + m[0][2] * c[2], m[1][0] * c[0] + m[1][1] * c[1] + m[1][2] * c[2],
                 m[2][0] * c[0] + m[2][1] * c[1] + m[2][2] * c[2]);
}

Color Aces::tone_map(const Color &color) const { // aces.rs:25-55
    Color v = mat_mul(input_matrix, color);
    Color a = v * (v + 0.0245786) - 0.000090537;
    Color b = v * (0.983729 * v + 0.4329510) + 0.238081;
    return mat_mul(output_matrix, Color(a.x() / b.x(), a.y() / b.y(), a.z() / b.z()));
}

std::unique_ptr<ToneMap> make_tone_map(const ToneMapConfig &t) { // tone_map.rs:18-66
    switch (t.kind) {
    case ToneMapConfig::Reinhard: return std::make_unique<Reinhard>(t.max_white.value_or(25.0));
    case ToneMapConfig::Hable:
        return std::make_unique<Hable>(
            HableData{t.shoulder_strength.value_or(0.15), t.linear_strength.value_or(0.5), t.linear_angle.value_or(0.1),
                      t.toe_strength.value_or(0.2), t.toe_numerator.value_or(0.02), t.toe_denominator.value_or(0.3)},
            t.exposure_bias.value_or(2.0), t.linear_white_point.value_or(11.2));
    case ToneMapConfig::Aces: {
        std::array<Color, 3> in = {Color(0.59719, 0.35458, 0.04823), Color(0.07600, 0.90834, 0.01566),
                                   Color(0.02840, 0.13383, 0.83777)};
        std::array<Color, 3> out = {Color(1.60475, -0.53108, -0.07367), Color(-0.10208, 1.10813, -0.00605),
                                    Color(-0.00327, -0.07276, 1.07602)};
        return std::make_unique<Aces>(t.input_matrix.value_or(in), t.output_matrix.value_or(out));
    }
    default: return std::make_unique<ToneMapNone>();
    }
}

} // namespace rthost
