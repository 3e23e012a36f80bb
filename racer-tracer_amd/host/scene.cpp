// scene.cpp — see scene.h.  YAML scene schema and two-pass resolution follow
// racer-tracer/src/scene/yml.rs:49-457.
#include "scene.h"
#include <atomic>
#include <cmath>
#include <cstring>
#include "../../include/rt_rng.h"
#include "error.h"
#include "image_io.h"

namespace rthost {

// ------------------------------------------------------------ host-side RNG
// Philox4x32-R, R = RT_PHILOX_ROUNDS (include/rt_rng.h); used only to fill the Perlin gradient
// table, which the reference fills from thread_rng (noise.rs:45-47).
static void philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < RT_PHILOX_ROUNDS; ++r) {
        uint64_t p0 = (uint64_t)RT_PHILOX_M0 * c[0], p1 = (uint64_t)RT_PHILOX_M1 * c[2];
        uint32_t n[4] = {(uint32_t)(p1 >> 32) ^ c[1] ^ k0, (uint32_t)p1, (uint32_t)(p0 >> 32) ^ c[3] ^ k1, (uint32_t)p0};
        memcpy(c, n, sizeof c);
        k0 += RT_PHILOX_W0;
        k1 += RT_PHILOX_W1;
    }
    memcpy(out, c, sizeof c);
}
static double u53(uint32_t hi, uint32_t lo) {
    return (double)((((uint64_t)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
}

Perlin::Perlin(uint64_t seed, uint32_t perlin_index) {
    uint32_t key[2] = {(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32)};
    for (uint32_t i = 0; i < 256; ++i) { // noise.rs:45-47: Vec3::random_range(-1, 1).unit_vector()
        uint32_t c0[4] = {i, RT_RNG_SAMPLE_TABLE, (perlin_index << 8) | RT_RNG_PERLIN, 0}, o0[4], o1[4];
        uint32_t c1[4] = {i, RT_RNG_SAMPLE_TABLE, (perlin_index << 8) | RT_RNG_PERLIN, 1};
        philox4x32(c0, key, o0);
        philox4x32(c1, key, o1);
        Vec3 v(-1.0 + 2.0 * u53(o0[0], o0[1]), -1.0 + 2.0 * u53(o0[2], o0[3]), -1.0 + 2.0 * u53(o1[0], o1[1]));
        Vec3 g = v.unit_vector();
        for (int k = 0; k < 3; ++k) table.ranvec[i][k] = g[k];
        // noise.rs:111-130: `for i in (count-1)..0` is an empty range, so the
        // permutation tables stay the identity (SURVEY B-9)
        table.perm_x[i] = table.perm_y[i] = table.perm_z[i] = (int32_t)i;
    }
}

// ------------------------------------------------------------- flattening
int SceneFlattener::texture_index(const std::shared_ptr<const Texture> &t) {
    auto it = texture_ids_.find(t.get());
    if (it != texture_ids_.end()) return it->second;
    RtTexture rec = t->describe(*this); // children first
    int idx = (int)textures.size();
    textures.push_back(rec);
    texture_ids_[t.get()] = idx;
    keep_alive_.push_back(t);
    return idx;
}

int SceneFlattener::material_index(const std::shared_ptr<const Material> &m) {
    auto it = material_ids_.find(m.get());
    if (it != material_ids_.end()) return it->second;
    RtMaterial rec = m->describe(*this);
    int idx = (int)materials.size();
    materials.push_back(rec);
    material_ids_[m.get()] = idx;
    keep_alive_.push_back(m);
    return idx;
}

static RtTexture blank_texture(int kind) {
    RtTexture t;
    memset(&t, 0, sizeof t);
    t.kind = kind;
    t.tex_even = t.tex_odd = t.image = t.perlin = -1;
    return t;
}

RtTexture SolidColor::describe(SceneFlattener &) const {
    RtTexture t = blank_texture(RT_TEX_SOLID_COLOR);
    for (int k = 0; k < 3; ++k) t.color[k] = color[k];
    return t;
}

RtTexture Checkered::describe(SceneFlattener &f) const {
    RtTexture t = blank_texture(RT_TEX_CHECKERED);
    t.tex_even = f.texture_index(even);
    t.tex_odd = f.texture_index(odd);
    return t;
}

std::shared_ptr<TextureImage> TextureImage::try_new(const std::string &path) {
    auto img = std::make_shared<TextureImage>();
    std::string why;
    if (!decode_image_rgba8(path, img->rgba, img->width, img->height, why))
        throw TracerError::FailedToOpenImage(path, why);
    return img;
}

RtTexture TextureImage::describe(SceneFlattener &f) const {
    RtTexture t = blank_texture(RT_TEX_IMAGE);
    RtImage im;
    im.rgba = rgba.data();
    im.width = width;
    im.height = height;
    t.image = (int)f.images.size();
    f.images.push_back(im);
    return t;
}

RtTexture Noise::describe(SceneFlattener &f) const {
    RtTexture t = blank_texture(RT_TEX_NOISE);
    for (int k = 0; k < 3; ++k) t.color[k] = color[k];
    t.scale = scale;
    t.depth = depth;
    t.perlin = (int)f.perlins.size();
    f.perlins.push_back(perlin.table);
    return t;
}

static RtMaterial blank_material(int kind) {
    RtMaterial m;
    memset(&m, 0, sizeof m);
    m.kind = kind;
    m.texture = -1;
    return m;
}

RtMaterial Lambertian::describe(SceneFlattener &f) const {
    RtMaterial m = blank_material(RT_MAT_LAMBERTIAN);
    m.texture = f.texture_index(texture);
    return m;
}
RtMaterial Metal::describe(SceneFlattener &f) const {
    RtMaterial m = blank_material(RT_MAT_METAL);
    m.texture = f.texture_index(texture);
    m.fuzz = fuzz;
    return m;
}
RtMaterial Dialectric::describe(SceneFlattener &) const {
    RtMaterial m = blank_material(RT_MAT_DIELECTRIC);
    m.refraction_index = refraction_index;
    return m;
}
RtMaterial DiffuseLight::describe(SceneFlattener &f) const {
    RtMaterial m = blank_material(RT_MAT_DIFFUSE_LIGHT);
    m.texture = f.texture_index(texture);
    return m;
}

// ------------------------------------------------------------ scene objects
static std::atomic<size_t> g_scene_object_id{1}; // scene.rs:51

SceneObject::SceneObject(Vec3 pos, std::shared_ptr<const Material> material,
                         std::shared_ptr<const HittableSceneObject> hittable)
    : pos(pos), material(std::move(material)), hittable(std::move(hittable)), obj_id(g_scene_object_id.fetch_add(1)) {}

RtPrimitive SceneObject::describe(SceneFlattener &f) const {
    RtPrimitive p;
    memset(&p, 0, sizeof p);
    p.rot_cos = 1.0;
    hittable->describe(p);
    p.material = f.material_index(material);
    p.obj_id = (int32_t)obj_id;
    return p;
}

void Sphere::describe(RtPrimitive &out) const {
    out.kind = RT_PRIM_SPHERE;
    out.p[0] = center.x(); out.p[1] = center.y(); out.p[2] = center.z(); out.p[3] = radius;
}

void MovingSphere::describe(RtPrimitive &out) const {
    out.kind = RT_PRIM_MOVING_SPHERE;
    out.p[0] = pos_a.x(); out.p[1] = pos_a.y(); out.p[2] = pos_a.z(); out.p[3] = radius;
    for (int k = 0; k < 3; ++k) out.center_b[k] = pos_b[k];
    out.time_a = time_a;
    out.time_b = time_b;
}

void AxisRect::describe(RtPrimitive &out) const {
    out.kind = kind;
    out.p[0] = a0; out.p[1] = a1; out.p[2] = b0; out.p[3] = b1; out.p[4] = k;
}

void Boxx::describe(RtPrimitive &out) const {
    out.kind = RT_PRIM_BOX;
    for (int i = 0; i < 3; ++i) {
        out.p[i] = box_min[i];
        out.p[3 + i] = box_max[i];
    }
}

RotateY::RotateY(SceneObject obj, double degrees) : object(std::move(obj)) { // rotate_y.rs:19-28
    double radians = degrees * 3.14159265358979323846264338327950288 / 180.0; // util.rs:5-7
    sin_theta = std::sin(radians);
    cos_theta = std::cos(radians);
}

void RotateY::describe(RtPrimitive &out) const {
    object.hittable->describe(out);
    if (out.flags & (RT_PRIM_HAS_ROTATE_Y | RT_PRIM_HAS_TRANSLATE))
        throw TracerError(RT_ERR_UNSUPPORTED, "RotateY around an already rotated/translated object has no flat form");
    out.flags |= RT_PRIM_HAS_ROTATE_Y;
    out.rot_sin = sin_theta;
    out.rot_cos = cos_theta;
}

void Translate::describe(RtPrimitive &out) const {
    object.hittable->describe(out);
    if (out.flags & RT_PRIM_HAS_TRANSLATE)
        throw TracerError(RT_ERR_UNSUPPORTED, "Translate of a Translate has no flat form");
    out.flags |= RT_PRIM_HAS_TRANSLATE;
    for (int k = 0; k < 3; ++k) out.translate[k] = offset[k];
}

// geometry_creation.rs
SceneObject create_sphere(std::shared_ptr<const Material> m, Vec3 pos, double radius) {
    return SceneObject(pos, std::move(m), std::make_shared<Sphere>(pos, radius));
}
SceneObject create_movable_sphere(std::shared_ptr<const Material> m, Vec3 pos_a, Vec3 pos_b, double radius,
                                  double time_a, double time_b) {
    return SceneObject(pos_a, std::move(m), std::make_shared<MovingSphere>(pos_a, pos_b, radius, time_a, time_b));
}
SceneObject create_xy_rect(std::shared_ptr<const Material> m, double x0, double x1, double y0, double y1, double k) {
    return SceneObject(Vec3(x0, y0, k), std::move(m), std::make_shared<AxisRect>(RT_PRIM_XY_RECT, x0, x1, y0, y1, k));
}
SceneObject create_xz_rect(std::shared_ptr<const Material> m, double x0, double x1, double z0, double z1, double k) {
    return SceneObject(Vec3(x0, k, z0), std::move(m), std::make_shared<AxisRect>(RT_PRIM_XZ_RECT, x0, x1, z0, z1, k));
}
SceneObject create_yz_rect(std::shared_ptr<const Material> m, double y0, double y1, double z0, double z1, double k) {
    return SceneObject(Vec3(k, y0, z0), std::move(m), std::make_shared<AxisRect>(RT_PRIM_YZ_RECT, y0, y1, z0, z1, k));
}
SceneObject create_box(std::shared_ptr<const Material> m, Vec3 mn, Vec3 mx) {
    for (int i = 0; i < 6; ++i) g_scene_object_id.fetch_add(1); // box.rs:22-71: six side SceneObjects take ids first
    return SceneObject(mn, std::move(m), std::make_shared<Boxx>(mn, mx));
}
SceneObject create_translate(Vec3 offset, SceneObject obj) {
    Vec3 pos = obj.pos;
    auto material = obj.material;
    return SceneObject(pos, material, std::make_shared<Translate>(offset, std::move(obj)));
}
SceneObject create_rotate_y(double degrees, SceneObject obj) {
    Vec3 pos = obj.pos;
    auto material = obj.material;
    return SceneObject(pos, material, std::make_shared<RotateY>(std::move(obj), degrees));
}

RtBackground Sky::describe() const {
    RtBackground b;
    memset(&b, 0, sizeof b);
    b.kind = RT_BG_SKY;
    for (int k = 0; k < 3; ++k) {
        b.top[k] = top[k];
        b.bottom[k] = bottom[k];
    }
    return b;
}
RtBackground SolidBackgroundColor::describe() const {
    RtBackground b;
    memset(&b, 0, sizeof b);
    b.kind = RT_BG_SOLID;
    for (int k = 0; k < 3; ++k) b.top[k] = color[k];
    return b;
}

// ------------------------------------------------------------------ loaders
namespace {

template <class T> T *find_named(std::vector<std::pair<std::string, T>> &v, const std::string &key) {
    for (auto &kv : v)
        if (kv.first == key) return &kv.second;
    return nullptr;
}

const YamlNode &need(const YamlNode &m, const char *key, const std::string &file, const std::string &owner) {
    const YamlNode *n = m.is_map() ? m.find(key) : nullptr;
    if (!n) throw TracerError::Configuration(file, "missing field `" + std::string(key) + "` in " + owner);
    return *n;
}

std::string need_str(const YamlNode &m, const char *key, const std::string &file, const std::string &owner) {
    const YamlNode &n = need(m, key, file, owner);
    if (!n.is_scalar()) throw TracerError::Configuration(file, "`" + std::string(key) + "` of " + owner + " must be a string");
    return n.scalar;
}

// texture image paths are relative to the process CWD in the reference
// (texture/image.rs:18); as a convenience a path that does not exist there is
// retried relative to the scene file's directory.
std::string resolve_path(const std::string &path, const std::string &scene_file) {
    if (!path.empty() && path[0] == '/') return path;
    if (file_exists(path)) return path;
    size_t slash = scene_file.find_last_of('/');
    std::string dir = slash == std::string::npos ? std::string(".") : scene_file.substr(0, slash);
    std::string alt = dir + "/" + path;
    return file_exists(alt) ? alt : path;
}

} // namespace

SceneLoadData YmlLoader::load() const {
    const std::string &file = path_;
    YamlNode root = parse_yaml_file(file);
    if (!root.is_map()) throw TracerError::Configuration(file, "top level must be a mapping");
    static const YamlNode empty_map = [] { YamlNode n; n.kind = YamlNode::Map; return n; }();
    const YamlNode &ytex = need(root, "textures", file, "scene");
    const YamlNode &ymat = need(root, "materials", file, "scene");
    const YamlNode &ygeo = need(root, "geometry", file, "scene");

    // ---- textures (yml.rs:175-243): Checkered resolved after everything else
    std::vector<std::pair<std::string, std::shared_ptr<const Texture>>> textures;
    std::vector<std::pair<std::string, std::pair<std::string, std::string>>> checkered;
    uint32_t perlin_count = 0;
    for (const auto &kv : (ytex.is_map() ? ytex : empty_map).map) {
        const YamlNode *body;
        std::string kind = yaml_variant(kv.second, &body, file, "texture " + kv.first);
        if (!body || !body->is_map()) throw TracerError::Configuration(file, "texture " + kv.first + " needs fields");
        if (iequals(kind, "SolidColor")) {
            textures.emplace_back(kv.first, std::make_shared<SolidColor>(yaml_vec3(need(*body, "color", file, kv.first), file, kv.first)));
        } else if (iequals(kind, "Checkered")) {
            checkered.emplace_back(kv.first, std::make_pair(need_str(*body, "texture_a", file, kv.first),
                                                            need_str(*body, "texture_b", file, kv.first)));
        } else if (iequals(kind, "Image")) {
            textures.emplace_back(kv.first, TextureImage::try_new(resolve_path(need_str(*body, "path", file, kv.first), file)));
        } else if (iequals(kind, "Noise")) {
            double scale = yaml_f64(need(*body, "scale", file, kv.first), file, "scale");
            int depth = (int)yaml_f64(need(*body, "depth", file, kv.first), file, "depth");
            Color color = yaml_vec3(need(*body, "color", file, kv.first), file, kv.first);
            textures.emplace_back(kv.first, std::make_shared<Noise>(scale, depth, color, seed_, perlin_count++));
        } else {
            throw TracerError::Configuration(file, "unknown texture variant \"" + kind + "\"");
        }
    }
    for (const auto &c : checkered) {
        auto *a = find_named(textures, c.second.first);
        auto *b = find_named(textures, c.second.second);
        if (!a || !b)
            throw TracerError::SceneLoad("Checkered texture \"" + c.first + "\" expected texture \"" +
                                         (a ? c.second.second : c.second.first) + "\" to exist.");
        if (dynamic_cast<const Checkered *>(a->get()) || dynamic_cast<const Checkered *>(b->get()))
            throw TracerError::SceneLoad("Checkered texture \"" + c.first + "\" refers to another Checkered texture.");
        textures.emplace_back(c.first, std::make_shared<Checkered>(*a, *b));
    }

    // ---- materials (yml.rs:245-286)
    std::vector<std::pair<std::string, std::shared_ptr<const Material>>> materials;
    auto texture_of = [&](const YamlNode &body, const std::string &mat_kind, const std::string &key) {
        const YamlNode *t = body.find("texture");
        if (!t) t = body.find("texture_key"); // #[serde(alias = "texture")] texture_key
        if (!t || !t->is_scalar()) throw TracerError::Configuration(file, "material " + key + " needs `texture`");
        auto *tex = find_named(textures, t->scalar);
        if (!tex)
            throw TracerError::SceneLoad("Failed to find texture \"" + t->scalar + "\" for " + mat_kind + " material \"" + key + "\"");
        return *tex;
    };
    for (const auto &kv : (ymat.is_map() ? ymat : empty_map).map) {
        const YamlNode *body;
        std::string kind = yaml_variant(kv.second, &body, file, "material " + kv.first);
        if (!body || !body->is_map()) throw TracerError::Configuration(file, "material " + kv.first + " needs fields");
        if (iequals(kind, "Lambertian")) {
            materials.emplace_back(kv.first, std::make_shared<Lambertian>(texture_of(*body, "lambertian", kv.first)));
        } else if (iequals(kind, "Metal")) {
            double fuzz = yaml_f64(need(*body, "fuzz", file, kv.first), file, "fuzz");
            materials.emplace_back(kv.first, std::make_shared<Metal>(texture_of(*body, "metal", kv.first), fuzz));
        } else if (iequals(kind, "Dialectric")) {
            materials.emplace_back(kv.first, std::make_shared<Dialectric>(
                                                 yaml_f64(need(*body, "refraction_index", file, kv.first), file, "refraction_index")));
        } else if (iequals(kind, "DiffuseLight")) {
            materials.emplace_back(kv.first, std::make_shared<DiffuseLight>(texture_of(*body, "diffuse light", kv.first)));
        } else {
            throw TracerError::Configuration(file, "unknown material variant \"" + kind + "\"");
        }
    }

    // ---- geometry (yml.rs:288-439): primitives, then RotateY wrappers, then Translate wrappers
    std::vector<std::pair<std::string, SceneObject>> geometry;
    std::vector<std::pair<std::string, double>> rotations;   // child key -> degrees
    std::vector<std::pair<std::string, Vec3>> translations;  // child key -> offset
    auto material_of = [&](const YamlNode &body, const std::string &key) {
        std::string name = need_str(body, "material", file, key);
        auto *m = find_named(materials, name);
        if (!m) throw TracerError::UnknownMaterial(name);
        return *m;
    };
    auto insert = [&](const std::string &key, SceneObject obj) {
        if (find_named(geometry, key))
            throw TracerError::SceneLoad("The object \"" + key + "\" was already present in the scene.");
        geometry.emplace_back(key, std::move(obj));
    };
    auto f = [&](const YamlNode &body, const char *k, const std::string &owner) {
        return yaml_f64(need(body, k, file, owner), file, k);
    };
    for (const auto &kv : (ygeo.is_map() ? ygeo : empty_map).map) {
        const YamlNode *body;
        std::string kind = yaml_variant(kv.second, &body, file, "geometry " + kv.first);
        if (!body || !body->is_map()) throw TracerError::Configuration(file, "geometry " + kv.first + " needs fields");
        const YamlNode &b = *body;
        const std::string &key = kv.first;
        if (iequals(kind, "Sphere")) {
            insert(key, create_sphere(material_of(b, key), yaml_vec3_flat(b, file, key), f(b, "radius", key)));
        } else if (iequals(kind, "XyRect")) {
            insert(key, create_xy_rect(material_of(b, key), f(b, "x0", key), f(b, "x1", key), f(b, "y0", key), f(b, "y1", key), f(b, "k", key)));
        } else if (iequals(kind, "XzRect")) {
            insert(key, create_xz_rect(material_of(b, key), f(b, "x0", key), f(b, "x1", key), f(b, "z0", key), f(b, "z1", key), f(b, "k", key)));
        } else if (iequals(kind, "YzRect")) {
            insert(key, create_yz_rect(material_of(b, key), f(b, "y0", key), f(b, "y1", key), f(b, "z0", key), f(b, "z1", key), f(b, "k", key)));
        } else if (iequals(kind, "Box")) {
            insert(key, create_box(material_of(b, key), yaml_vec3(need(b, "min", file, key), file, key),
                                   yaml_vec3(need(b, "max", file, key), file, key)));
        } else if (iequals(kind, "RotateY")) { // its own name is discarded; `key` names the child (SURVEY B-21)
            std::string child = need_str(b, "key", file, key);
            double deg = f(b, "degrees", key);
            if (auto *r = find_named(rotations, child)) *r = deg; else rotations.emplace_back(child, deg);
        } else if (iequals(kind, "Translate")) {
            std::string child = need_str(b, "key", file, key);
            Vec3 off = yaml_vec3_flat(b, file, key);
            if (auto *t = find_named(translations, child)) *t = off; else translations.emplace_back(child, off);
        } else {
            throw TracerError::Configuration(file, "unknown geometry variant \"" + kind + "\"");
        }
    }
    auto take = [&](const std::string &key) -> std::optional<SceneObject> {
        for (size_t i = 0; i < geometry.size(); ++i)
            if (geometry[i].first == key) {
                SceneObject o = std::move(geometry[i].second);
                geometry.erase(geometry.begin() + (long)i);
                return o;
            }
        return std::nullopt;
    };
    for (const auto &r : rotations) {
        size_t at = 0;
        while (at < geometry.size() && geometry[at].first != r.first) ++at;
        auto child = take(r.first);
        if (!child)
            throw TracerError::SceneLoad("Rotation_Y \"" + r.first + "\" did not have any child with key \"" + r.first + "\"");
        geometry.insert(geometry.begin() + (long)at, std::make_pair(r.first, create_rotate_y(r.second, std::move(*child))));
    }
    for (const auto &t : translations) {
        size_t at = 0;
        while (at < geometry.size() && geometry[at].first != t.first) ++at;
        auto child = take(t.first);
        if (!child)
            throw TracerError::SceneLoad("Translation \"" + t.first + "\" did not have any child with key \"" + t.first + "\"");
        geometry.insert(geometry.begin() + (long)at, std::make_pair(t.first, create_translate(t.second, std::move(*child))));
    }

    SceneLoadData out;
    for (auto &kv : geometry) out.objects.push_back(std::move(kv.second));
    // ---- background (yml.rs:442-453)
    if (const YamlNode *bg = root.find("background")) {
        const YamlNode *body;
        std::string kind = yaml_variant(*bg, &body, file, "background");
        if (iequals(kind, "Sky")) {
            if (!body) throw TracerError::Configuration(file, "Sky needs top and bottom");
            out.background = std::make_unique<Sky>(yaml_vec3(need(*body, "top", file, "Sky"), file, "Sky.top"),
                                                   yaml_vec3(need(*body, "bottom", file, "Sky"), file, "Sky.bottom"));
        } else if (iequals(kind, "SolidColor")) {
            if (!body) throw TracerError::Configuration(file, "SolidColor background needs a colour");
            out.background = std::make_unique<SolidBackgroundColor>(yaml_vec3(*body, file, "background"));
        } else {
            throw TracerError::Configuration(file, "unknown background variant \"" + kind + "\"");
        }
    } else {
        out.background = std::make_unique<Sky>();
    }
    if (const YamlNode *cam = root.find("camera")) out.camera = yaml_camera(*cam, file);
    if (const YamlNode *tm = root.find("tone_map")) out.tone_map = make_tone_map(yaml_tone_map(*tm, file));
    return out;
}

SceneLoadData SandboxLoader::load() const { // scene/sandbox.rs:39-80
    SceneLoadData data = YmlLoader(path_, seed_).load();
    std::shared_ptr<const Material> white = Lambertian::new_with_color(Color(0.63, 0.63, 0.63));
    SceneObject box1 = create_box(white, Vec3(0.0, 0.0, 0.0), Vec3(165.0, 330.0, 165.0));
    data.objects.push_back(create_translate(Vec3(265.0, 0.0, 295.0), create_rotate_y(15.0, std::move(box1))));
    SceneObject box2 = create_box(white, Vec3(0.0, 0.0, 0.0), Vec3(165.0, 165.0, 165.0));
    data.objects.push_back(create_translate(Vec3(130.0, 0.0, 65.0), create_rotate_y(-18.0, std::move(box2))));
    data.background = std::make_unique<SolidBackgroundColor>(Color(0.0, 0.0, 0.0));
    CameraConfig cam;
    cam.vfov = 40.0;
    cam.aperture = 0.0;
    cam.focus_distance = 10000.0;
    cam.pos = Vec3(278.0, 278.0, -800.0);
    cam.look_at = Vec3(278.0, 278.0, 0.0);
    data.camera = cam;
    data.tone_map.reset();
    return data;
}

SceneLoadData RandomLoader::load() const { // scene/random.rs:25-96
    // the n-th random_double() of the loader = d0 of block 0 at pixel = n (include/rt_rng.h, RT_RNG_SCENE)
    uint32_t key[2] = {(uint32_t)(seed_ & 0xffffffffu), (uint32_t)(seed_ >> 32)};
    uint32_t n = 0;
    auto random_double = [&]() {
        uint32_t ctr[4] = {n++, RT_RNG_SAMPLE_TABLE, RT_RNG_SCENE, 0}, out[4];
        philox4x32(ctr, key, out);
        return u53(out[0], out[1]);
    };
    auto random_range = [&](double a, double b) { return a + (b - a) * random_double(); };
    auto random_color = [&]() { // Vec3::random(): x, y, z in that order (vec3.rs:95-99)
        double x = random_double(), y = random_double(), z = random_double();
        return Color(x, y, z);
    };
    SceneLoadData data;
    auto checkered = std::make_shared<Checkered>(std::make_shared<SolidColor>(Color(0.2, 0.3, 0.1)),
                                                 std::make_shared<SolidColor>(Color(0.9, 0.9, 0.9)));
    data.objects.push_back(create_sphere(std::make_shared<Lambertian>(checkered), Vec3(0.0, -1000.0, 0.0), 1000.0));
    for (int a = -11; a < 11; ++a)
        for (int b = -11; b < 11; ++b) {
            double choose_mat = random_double();
            double cx = (double)a + 0.9 * random_double();
            double cz = (double)b + 0.9 * random_double();
            Vec3 center(cx, 0.2, cz);
            if ((center - Vec3(4.0, 0.2, 0.0)).length() > 0.9) {
                if (choose_mat < 0.8) { // diffuse, moving
                    Color c1 = random_color();
                    Color c2 = random_color();
                    Color albedo = c1 * c2;
                    Vec3 center2 = center + Vec3(0.0, random_range(0.0, 0.5), 0.0);
                    data.objects.push_back(create_movable_sphere(Lambertian::new_with_color(albedo), center, center2, 0.2, 0.0, 1.0));
                } else if (choose_mat > 0.95) { // metal
                    double r = random_range(0.5, 1.0), g = random_range(0.5, 1.0), bl = random_range(0.5, 1.0);
                    double fuzz = random_range(0.0, 0.5);
                    data.objects.push_back(create_sphere(
                        std::make_shared<Metal>(std::make_shared<SolidColor>(Color(r, g, bl)), fuzz), center, 0.2));
                } else { // glass
                    data.objects.push_back(create_sphere(std::make_shared<Dialectric>(1.5), center, 0.2));
                }
            }
        }
    data.objects.push_back(create_sphere(std::make_shared<Dialectric>(1.5), Vec3(0.0, 1.0, 0.0), 1.0));
    data.objects.push_back(create_sphere(Lambertian::new_with_color(Color(0.4, 0.2, 0.1)), Vec3(-4.0, 1.0, 0.0), 1.0));
    data.objects.push_back(create_sphere(
        std::make_shared<Metal>(std::make_shared<SolidColor>(Color(0.7, 0.6, 0.5)), 0.0), Vec3(4.0, 1.0, 0.0), 1.0));
    data.background = std::make_unique<Sky>();
    CameraConfig cam;
    cam.vfov = 20.0;
    cam.aperture = 0.1;
    cam.focus_distance = 10.0;
    cam.pos = Vec3(0.0, 2.0, 10.0);
    cam.look_at = Vec3(0.0, 0.0, 0.0);
    cam.speed = 0.000002;
    data.camera = cam;
    return data;
}

SceneLoadData NoneLoader::load() const { // scene/none.rs
    SceneLoadData data;
    data.background = std::make_unique<Sky>();
    return data;
}

std::unique_ptr<SceneLoader> make_loader(const SceneLoaderConfig &cfg, uint64_t seed) { // main.rs:74-79
    switch (cfg.kind) {
    case SceneLoaderConfig::Yml: return std::make_unique<YmlLoader>(cfg.path, seed);
    case SceneLoaderConfig::Sandbox: return std::make_unique<SandboxLoader>("../resources/scenes/cornell_box.yml", seed);
    case SceneLoaderConfig::None: return std::make_unique<NoneLoader>();
    default: return std::make_unique<RandomLoader>(seed);
    }
}

std::unique_ptr<FlatScene> flatten_scene(const SceneLoadData &data) {
    auto flat = std::make_unique<FlatScene>();
    SceneFlattener &t = flat->tables;
    for (const SceneObject &o : data.objects) {
        RtPrimitive p = o.describe(t);
        t.primitives.push_back(p);
    }
    RtSceneDesc &d = flat->desc;
    memset(&d, 0, sizeof d);
    d.primitives = t.primitives.data();
    d.n_primitives = (int32_t)t.primitives.size();
    d.materials = t.materials.data();
    d.n_materials = (int32_t)t.materials.size();
    d.textures = t.textures.data();
    d.n_textures = (int32_t)t.textures.size();
    d.images = t.images.data();
    d.n_images = (int32_t)t.images.size();
    d.perlins = t.perlins.data();
    d.n_perlins = (int32_t)t.perlins.size();
    d.background = data.background ? data.background->describe() : Sky().describe();
    return flat;
}

} // namespace rthost
