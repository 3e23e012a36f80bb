// tone_map.h — ToneMap trait and its four implementations
// (racer-tracer/src/tone_map.rs, tone_map/{none,reinhard,hable,aces}.rs).
// Applied to the gamma-encoded frame AFTER the renderer, like
// ScreenBuffer::update does (image_buffer.rs:135-170).
#pragma once
#include <array>
#include <memory>
#include "../../include/rt_abi.h"
#include "config.h"
#include "vec3.h"

namespace rthost {

class ToneMap {
  public:
    virtual ~ToneMap() = default;
    virtual Color tone_map(const Color &color) const = 0;
    virtual const char *name() const = 0;
    virtual RtToneMap describe() const = 0; // POD form for the device-side post pass (rt_post_rgba8_device)
};

class ToneMapNone : public ToneMap { // tone_map/none.rs
  public:
    Color tone_map(const Color &c) const override { return c; }
    const char *name() const override { return "None"; }
    RtToneMap describe() const override;
};

class Reinhard : public ToneMap { // tone_map/reinhard.rs
  public:
    explicit Reinhard(double max_white) : max_white(max_white), max_white_pow(max_white * max_white) {}
    Color tone_map(const Color &c) const override;
    const char *name() const override { return "Reinhard"; }
    RtToneMap describe() const override;
    double max_white, max_white_pow;
};

struct HableData { // tone_map/hable.rs:5-12
    double shoulder_strength, linear_strength, linear_angle, toe_strength, toe_numerator, toe_denominator;
};

class Hable : public ToneMap { // tone_map/hable.rs
  public:
    Hable(HableData data, double exposure_bias, double linear_white_point);
    Color tone_map(const Color &c) const override;
    const char *name() const override { return "Hable"; }
    RtToneMap describe() const override;
    static double partial(double color, const HableData &d, double toe_angle);
    HableData data;
    double toe_angle, exposure_bias, linear_white_point, white_scale;
};

class Aces : public ToneMap { // tone_map/aces.rs
  public:
    Aces(std::array<Color, 3> in, std::array<Color, 3> out) : input_matrix(in), output_matrix(out) {}
    Color tone_map(const Color &c) const override;
    const char *name() const override { return "Aces"; }
    RtToneMap describe() const override;
    std::array<Color, 3> input_matrix, output_matrix;
};

// impl From<&ToneMapConfig> for Box<dyn ToneMap> (tone_map.rs:18-66)
std::unique_ptr<ToneMap> make_tone_map(const ToneMapConfig &cfg);

} // namespace rthost
