// config.cpp — see config.h.
#include "config.h"
#include <array>
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include "error.h"

namespace rthost {

double yaml_f64(const YamlNode &n, const std::string &file, const std::string &what) {
    if (!n.is_scalar()) throw TracerError::Configuration(file, "expected a number for " + what);
    const char *s = n.scalar.c_str();
    char *end = nullptr;
    errno = 0;
    double v = std::strtod(s, &end);
    if (end == s || *end != '\0') throw TracerError::Configuration(file, "invalid number \"" + n.scalar + "\" for " + what);
    return v;
}

static size_t yaml_usize(const YamlNode &n, const std::string &file, const std::string &what) {
    double v = yaml_f64(n, file, what);
    if (v < 0 || v != (double)(size_t)v) throw TracerError::Configuration(file, "expected a non-negative integer for " + what);
    return (size_t)v;
}

// vec3.rs:12-16: a map with key `pos` (alias `color`) holding [x, y, z]
Vec3 yaml_vec3_flat(const YamlNode &parent, const std::string &file, const std::string &what) {
    const YamlNode *arr = parent.find("pos");
    if (!arr) arr = parent.find("color");
    if (!arr || !arr->is_seq() || arr->seq.size() != 3)
        throw TracerError::Configuration(file, "expected `pos: [x, y, z]` for " + what);
    return Vec3(yaml_f64(arr->seq[0], file, what), yaml_f64(arr->seq[1], file, what), yaml_f64(arr->seq[2], file, what));
}

Vec3 yaml_vec3(const YamlNode &n, const std::string &file, const std::string &what) {
    if (!n.is_map()) throw TracerError::Configuration(file, "expected a map with `pos`/`color` for " + what);
    return yaml_vec3_flat(n, file, what);
}

std::string yaml_variant(const YamlNode &n, const YamlNode **body, const std::string &file, const std::string &what) {
    *body = nullptr;
    if (n.is_scalar()) return n.scalar;
    if (n.is_map() && n.map.size() == 1) {
        *body = &n.map[0].second;
        return n.map[0].first;
    }
    throw TracerError::Configuration(file, "expected an enum variant for " + what);
}

static std::optional<double> opt_f64(const YamlNode &m, const char *key, const std::string &file) {
    const YamlNode *n = m.find(key);
    if (!n || n->is_null()) return std::nullopt;
    return yaml_f64(*n, file, key);
}

CameraConfig yaml_camera(const YamlNode &n, const std::string &file) {
    CameraConfig c;
    if (!n.is_map()) return c;
    c.vfov = opt_f64(n, "vfov", file);
    c.aperture = opt_f64(n, "aperture", file);
    c.focus_distance = opt_f64(n, "focus_distance", file);
    c.speed = opt_f64(n, "speed", file);
    c.sensitivity = opt_f64(n, "sensitivity", file);
    if (const YamlNode *p = n.find("pos")) c.pos = yaml_vec3(*p, file, "camera.pos");
    if (const YamlNode *p = n.find("look_at")) c.look_at = yaml_vec3(*p, file, "camera.look_at");
    return c;
}

static std::array<Vec3, 3> yaml_matrix(const YamlNode &n, const std::string &file) { // config.rs:131-134
    const YamlNode *colors = n.find("colors");
    if (!colors || !colors->is_seq() || colors->seq.size() != 3)
        throw TracerError::Configuration(file, "expected `colors:` with three vectors");
    std::array<Vec3, 3> m;
    for (int i = 0; i < 3; ++i) m[(size_t)i] = yaml_vec3(colors->seq[(size_t)i], file, "matrix row");
    return m;
}

ToneMapConfig yaml_tone_map(const YamlNode &n, const std::string &file) {
    ToneMapConfig t;
    const YamlNode *body = nullptr;
    std::string name = yaml_variant(n, &body, file, "tone_map");
    static const YamlNode empty;
    const YamlNode &b = body && body->is_map() ? *body : empty;
    if (iequals(name, "None")) {
        t.kind = ToneMapConfig::None;
    } else if (iequals(name, "Reinhard")) {
        t.kind = ToneMapConfig::Reinhard;
        t.max_white = opt_f64(b, "max_white", file);
    } else if (iequals(name, "Hable")) {
        t.kind = ToneMapConfig::Hable;
        t.shoulder_strength = opt_f64(b, "shoulder_strength", file);
        t.linear_strength = opt_f64(b, "linear_strength", file);
        t.linear_angle = opt_f64(b, "linear_angle", file);
        t.toe_strength = opt_f64(b, "toe_strength", file);
        t.toe_numerator = opt_f64(b, "toe_numerator", file);
        t.toe_denominator = opt_f64(b, "toe_denominator", file);
        t.exposure_bias = opt_f64(b, "exposure_bias", file);
        t.linear_white_point = opt_f64(b, "linear_white_point", file);
    } else if (iequals(name, "Aces")) {
        t.kind = ToneMapConfig::Aces;
        if (const YamlNode *m = b.find("input_matrix")) t.input_matrix = yaml_matrix(*m, file);
        if (const YamlNode *m = b.find("output_matrix")) t.output_matrix = yaml_matrix(*m, file);
    } else {
        throw TracerError::Configuration(file, "unknown tone_map variant \"" + name + "\"");
    }
    return t;
}

static RenderConfig yaml_render(const YamlNode &n, const std::string &file) {
    RenderConfig r;
    if (!n.is_map()) return r;
    auto get = [&](const char *k, size_t &dst) {
        const YamlNode *v = n.find(k);
        if (!v) throw TracerError::Configuration(file, std::string("missing field `") + k + "`");
        dst = yaml_usize(*v, file, k);
    };
    get("samples", r.samples);
    get("max_depth", r.max_depth);
    get("num_threads_width", r.num_threads_width);
    get("num_threads_height", r.num_threads_height);
    get("scale", r.scale);
    return r;
}

static RendererConfig yaml_renderer(const YamlNode &n, const std::string &file) {
    const YamlNode *body;
    std::string name = yaml_variant(n, &body, file, "renderer");
    if (iequals(name, "Cpu")) return RendererConfig::Cpu;
    if (iequals(name, "CpuPreview")) return RendererConfig::CpuPreview;
    if (iequals(name, "Hip")) return RendererConfig::Hip;
    throw TracerError::Configuration(file, "unknown renderer \"" + name + "\"");
}

Config Config::from_file(const std::string &file) {
    YamlNode root = parse_yaml_file(file);
    Config c;
    if (root.is_null()) return c;
    if (!root.is_map()) throw TracerError::Configuration(file, "top level must be a mapping");
    if (const YamlNode *n = root.find("preview")) c.preview = yaml_render(*n, file);
    if (const YamlNode *n = root.find("render")) c.render = yaml_render(*n, file);
    if (const YamlNode *n = root.find("screen")) {
        const YamlNode *w = n->find("width"), *h = n->find("height");
        if (!w || !h) throw TracerError::Configuration(file, "screen needs width and height");
        c.screen.width = yaml_usize(*w, file, "screen.width");
        c.screen.height = yaml_usize(*h, file, "screen.height");
    }
    if (const YamlNode *n = root.find("loader")) {
        const YamlNode *body;
        std::string name = yaml_variant(*n, &body, file, "loader");
        if (iequals(name, "None")) c.loader.kind = SceneLoaderConfig::None;
        else if (iequals(name, "Random")) c.loader.kind = SceneLoaderConfig::Random;
        else if (iequals(name, "Sandbox")) c.loader.kind = SceneLoaderConfig::Sandbox;
        else if (iequals(name, "Yml")) {
            const YamlNode *p = body ? body->find("path") : nullptr;
            if (!p || !p->is_scalar()) throw TracerError::Configuration(file, "loader Yml needs `path`");
            c.loader.kind = SceneLoaderConfig::Yml;
            c.loader.path = p->scalar;
        } else throw TracerError::Configuration(file, "unknown loader \"" + name + "\"");
    }
    if (const YamlNode *n = root.find("image_action")) {
        const YamlNode *body;
        std::string name = yaml_variant(*n, &body, file, "image_action");
        if (iequals(name, "SavePng")) c.image_action = ImageActionConfig::SavePng;
        else if (iequals(name, "None")) c.image_action = ImageActionConfig::None;
        else throw TracerError::Configuration(file, "unknown image_action \"" + name + "\"");
    }
    if (const YamlNode *n = root.find("image_output_dir"))
        if (n->is_scalar()) c.image_output_dir = n->scalar;
    if (const YamlNode *n = root.find("renderer")) c.renderer = yaml_renderer(*n, file);
    if (const YamlNode *n = root.find("preview_renderer")) c.preview_renderer = yaml_renderer(*n, file);
    if (const YamlNode *n = root.find("camera")) c.camera = yaml_camera(*n, file);
    if (const YamlNode *n = root.find("tone_map")) c.tone_map = yaml_tone_map(*n, file);
    return c;
}

ImageActionConfig image_action_from_str(const std::string &s) { // config.rs:119-129
    if (s == "png") return ImageActionConfig::SavePng;
    return ImageActionConfig::None; // "none" and every other string
}

Args Args::parse(int argc, const char *const *argv) {
    Args a;
    if (const char *env = std::getenv("CONFIG")) a.config = env;
    auto value = [&](int &i, const std::string &flag) -> std::string {
        if (i + 1 >= argc) throw TracerError::ArgumentParsingError("missing value for " + flag);
        return argv[++i];
    };
    for (int i = 1; i < argc; ++i) {
        std::string s = argv[i];
        std::string inline_val;
        size_t eq = s.find('=');
        bool has_inline = s.rfind("--", 0) == 0 && eq != std::string::npos;
        if (has_inline) {
            inline_val = s.substr(eq + 1);
            s = s.substr(0, eq);
        }
        auto val = [&]() { return has_inline ? inline_val : value(i, s); };
        if (s == "-c" || s == "--config") a.config = val();
        else if (s == "-s" || s == "--scene") a.scene = val();
        else if (s == "--image-action") a.image_action = image_action_from_str(val());
        else if (s == "--seed") a.seed = std::strtoull(val().c_str(), nullptr, 0);
        else if (s == "--device") a.device = std::atoi(val().c_str());
        else if (s == "--devices") a.devices = std::atoi(val().c_str());
        else if (s == "-h" || s == "--help") a.help = true;
        else throw TracerError::ArgumentParsingError("unknown argument " + s);
    }
    return a;
}

Config config_try_from(const Args &args) { // config.rs:30-67
    Config cfg = Config::from_file(args.config);
    if (args.image_action) cfg.image_action = *args.image_action;
    if (args.scene) {
        const std::string &scene = *args.scene;
        if (scene == "random") {
            cfg.loader.kind = SceneLoaderConfig::Random;
        } else if (scene == "sandbox") {
            cfg.loader.kind = SceneLoaderConfig::Sandbox;
        } else {
            size_t slash = scene.find_last_of('/');
            size_t dot = scene.find_last_of('.');
            if (dot == std::string::npos || (slash != std::string::npos && dot < slash) || dot + 1 == scene.size())
                throw TracerError::ArgumentParsingError("Could not get extension from scene file: " + scene);
            if (scene.substr(dot + 1) != "yml")
                throw TracerError::ArgumentParsingError("Could not find a suitable scene loader for file: " + scene);
            cfg.loader.kind = SceneLoaderConfig::Yml;
            cfg.loader.path = scene;
        }
    }
    return cfg;
}

} // namespace rthost
