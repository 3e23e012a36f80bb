// config.h — Config / Args of racer-tracer (racer-tracer/src/config.rs), same
// keys, same precedence (CLI over file, config.rs:30-67).
#pragma once
#include <optional>
#include <string>
#include "vec3.h"
#include "yaml.h"

namespace rthost {

struct ScreenConfig { // config.rs:69-73
    size_t height = 0, width = 0;
};

struct RenderConfig { // config.rs:75-82
    size_t samples = 0, max_depth = 0, num_threads_width = 0, num_threads_height = 0, scale = 0;
};

struct SceneLoaderConfig { // config.rs:84-93
    enum Kind { None, Yml, Random, Sandbox } kind = None;
    std::string path;
};

enum class ImageActionConfig { None, SavePng }; // config.rs:95-100

// config.rs:108-113 plus the one new variant a GPU backend adds
enum class RendererConfig { Cpu, CpuPreview, Hip };

struct ToneMapConfig { // config.rs:136-167
    enum Kind { None, Reinhard, Hable, Aces } kind = None;
    std::optional<double> max_white;
    std::optional<double> shoulder_strength, linear_strength, linear_angle, toe_strength, toe_numerator,
        toe_denominator, exposure_bias, linear_white_point;
    std::optional<std::array<Vec3, 3>> input_matrix, output_matrix;
};

struct CameraConfig { // config.rs:169-178
    std::optional<double> vfov, aperture, focus_distance;
    std::optional<Vec3> pos, look_at;
    std::optional<double> speed, sensitivity;
};

struct Config { // config.rs:180-214
    RenderConfig preview, render;
    ScreenConfig screen;
    SceneLoaderConfig loader;
    ImageActionConfig image_action = ImageActionConfig::None;
    std::optional<std::string> image_output_dir;
    RendererConfig renderer = RendererConfig::Hip;
    RendererConfig preview_renderer = RendererConfig::CpuPreview;
    CameraConfig camera;
    ToneMapConfig tone_map;

    static Config from_file(const std::string &file); // config.rs:216-225
};

struct Args { // config.rs:12-28
    std::string config = "./config.yml"; // -c/--config, env CONFIG
    std::optional<std::string> scene;    // -s/--scene <file.yml|random|sandbox>
    std::optional<ImageActionConfig> image_action; // --image-action <png|none>
    // additive (the reference has no headless mode and no seed):
    uint64_t seed = 1;
    int device = 0;
    int devices = 1; // --devices N: render the frame on devices device .. device+N-1 (rt_render_frame_multi)
    bool help = false;

    static Args parse(int argc, const char *const *argv);
};

ImageActionConfig image_action_from_str(const std::string &s); // config.rs:119-129
Config config_try_from(const Args &args);                      // config.rs:30-67

// Shared YAML -> value helpers (also used by the scene loader).
double yaml_f64(const YamlNode &n, const std::string &file, const std::string &what);
Vec3 yaml_vec3(const YamlNode &n, const std::string &file, const std::string &what);
Vec3 yaml_vec3_flat(const YamlNode &parent, const std::string &file, const std::string &what);
CameraConfig yaml_camera(const YamlNode &n, const std::string &file);
ToneMapConfig yaml_tone_map(const YamlNode &n, const std::string &file);
// Externally tagged enum: "Name" (unit) or {Name: body}.  Returns the variant
// name; *body is the payload or nullptr.
std::string yaml_variant(const YamlNode &n, const YamlNode **body, const std::string &file,
                         const std::string &what);

} // namespace rthost
