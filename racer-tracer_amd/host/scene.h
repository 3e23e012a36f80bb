// scene.h — host-side scene model: the reference's trait surface
// (Texture / Material / HittableSceneObject / BackgroundColor, SceneObject,
// SceneLoadData, SceneLoader) as C++ classes.
//
// What is different from the Rust traits, and why: `hit`, `scatter`,
// `color_emitted`, `value` and `color` are what the HIP kernel evaluates, so
// the host classes do not re-implement them on the CPU (there is no CPU
// render path in this library).  Instead every class has the one extra
// method SURVEY.md 8(b) calls for — `describe()` — which emits the POD record
// of include/rt_abi.h that the kernel consumes.  `&dyn Hittable` is opaque in
// the reference (geometry.rs:59-63), so a GPU backend needs exactly this.
#pragma once
#include <map>
#include <memory>
#include <optional>
#include <string>
#include <vector>
#include "../../include/rt_abi.h"
#include "config.h"
#include "tone_map.h"
#include "vec3.h"

namespace rthost {

class Texture;
class Material;

// Collects the POD tables while the object graph describes itself.  Shared
// textures/materials (the reference shares them through Arc) are emitted once.
class SceneFlattener {
  public:
    std::vector<RtTexture> textures;
    std::vector<RtMaterial> materials;
    std::vector<RtPrimitive> primitives;
    std::vector<RtImage> images;
    std::vector<RtPerlin> perlins;

    int texture_index(const std::shared_ptr<const Texture> &t);
    int material_index(const std::shared_ptr<const Material> &m);

  private:
    std::map<const void *, int> texture_ids_, material_ids_;
    std::vector<std::shared_ptr<const void>> keep_alive_;
};

// ---- texture.rs:8-10 ----
class Texture {
  public:
    virtual ~Texture() = default;
    virtual RtTexture describe(SceneFlattener &f) const = 0;
};

class SolidColor : public Texture { // texture/solid_color.rs
  public:
    explicit SolidColor(Color c) : color(c) {}
    RtTexture describe(SceneFlattener &f) const override;
    Color color;
};

class Checkered : public Texture { // texture/checkered.rs (checker_size is fixed at 10)
  public:
    Checkered(std::shared_ptr<const Texture> even, std::shared_ptr<const Texture> odd)
        : even(std::move(even)), odd(std::move(odd)) {}
    RtTexture describe(SceneFlattener &f) const override;
    std::shared_ptr<const Texture> even, odd;
};

class TextureImage : public Texture { // texture/image.rs
  public:
    static std::shared_ptr<TextureImage> try_new(const std::string &path); // FailedToOpenImage on error
    RtTexture describe(SceneFlattener &f) const override;
    std::vector<uint8_t> rgba; // row 0 = top
    int width = 0, height = 0;
};

class Perlin { // texture/noise.rs:36-55; gradients drawn under include/rt_rng.h (RT_RNG_PERLIN)
  public:
    Perlin(uint64_t seed, uint32_t perlin_index);
    RtPerlin table;
};

class Noise : public Texture { // texture/noise.rs:8-33
  public:
    Noise(double scale, std::optional<int> depth, Color color, uint64_t seed, uint32_t perlin_index)
        : perlin(seed, perlin_index), depth(depth.value_or(7)), color(color), scale(scale) {}
    RtTexture describe(SceneFlattener &f) const override;
    Perlin perlin;
    int depth;
    Color color;
    double scale;
};

// ---- material.rs:10-15 ----
class Material {
  public:
    virtual ~Material() = default;
    virtual RtMaterial describe(SceneFlattener &f) const = 0;
};

class Lambertian : public Material {
  public:
    explicit Lambertian(std::shared_ptr<const Texture> t) : texture(std::move(t)) {}
    static std::shared_ptr<Lambertian> new_with_color(Color c) { return std::make_shared<Lambertian>(std::make_shared<SolidColor>(c)); }
    RtMaterial describe(SceneFlattener &f) const override;
    std::shared_ptr<const Texture> texture;
};

class Metal : public Material {
  public:
    Metal(std::shared_ptr<const Texture> t, double fuzz) : texture(std::move(t)), fuzz(fuzz) {}
    RtMaterial describe(SceneFlattener &f) const override;
    std::shared_ptr<const Texture> texture;
    double fuzz;
};

class Dialectric : public Material { // (sic) the reference's spelling
  public:
    explicit Dialectric(double refraction_index) : refraction_index(refraction_index) {}
    RtMaterial describe(SceneFlattener &f) const override;
    double refraction_index;
};

class DiffuseLight : public Material {
  public:
    explicit DiffuseLight(std::shared_ptr<const Texture> t) : texture(std::move(t)) {}
    RtMaterial describe(SceneFlattener &f) const override;
    std::shared_ptr<const Texture> texture;
};

// ---- scene.rs:24-107 ----
class SceneObject;

class HittableSceneObject {
  public:
    virtual ~HittableSceneObject() = default;
    // Fill kind / p[] / wrapper fields of `out` (material and obj_id are the SceneObject's).
    virtual void describe(RtPrimitive &out) const = 0;
};

class SceneObject {
  public:
    SceneObject(Vec3 pos, std::shared_ptr<const Material> material, std::shared_ptr<const HittableSceneObject> hittable);
    RtPrimitive describe(SceneFlattener &f) const;
    Vec3 pos;
    std::shared_ptr<const Material> material;
    std::shared_ptr<const HittableSceneObject> hittable;
    size_t obj_id; // scene.rs:51,60: global counter starting at 1
};

class Sphere : public HittableSceneObject { // geometry/sphere.rs (centre = SceneObject.pos)
  public:
    Sphere(Vec3 center, double radius) : center(center), radius(radius) {}
    void describe(RtPrimitive &out) const override;
    Vec3 center;
    double radius;
};

class MovingSphere : public HittableSceneObject { // geometry/moving_sphere.rs (pos_a = SceneObject.pos)
  public:
    MovingSphere(Vec3 pos_a, Vec3 pos_b, double radius, double time_a, double time_b)
        : pos_a(pos_a), pos_b(pos_b), radius(radius), time_a(time_a), time_b(time_b) {}
    void describe(RtPrimitive &out) const override;
    Vec3 pos_a, pos_b;
    double radius, time_a, time_b;
};

class AxisRect : public HittableSceneObject { // geometry/{xy,xz,yz}_rect.rs
  public:
    AxisRect(int kind, double a0, double a1, double b0, double b1, double k) : kind(kind), a0(a0), a1(a1), b0(b0), b1(b1), k(k) {}
    void describe(RtPrimitive &out) const override;
    int kind; // RT_PRIM_XY_RECT / XZ / YZ
    double a0, a1, b0, b1, k;
};

class Boxx : public HittableSceneObject { // geometry/box.rs
  public:
    Boxx(Vec3 mn, Vec3 mx) : box_min(mn), box_max(mx) {}
    void describe(RtPrimitive &out) const override;
    Vec3 box_min, box_max;
};

class RotateY : public HittableSceneObject { // geometry/rotate_y.rs
  public:
    RotateY(SceneObject object, double degrees);
    void describe(RtPrimitive &out) const override;
    double sin_theta, cos_theta;
    SceneObject object;
};

class Translate : public HittableSceneObject { // geometry/translate.rs
  public:
    Translate(Vec3 offset, SceneObject object) : offset(offset), object(std::move(object)) {}
    void describe(RtPrimitive &out) const override;
    Vec3 offset;
    SceneObject object;
};

// geometry_creation.rs
SceneObject create_sphere(std::shared_ptr<const Material> m, Vec3 pos, double radius);
SceneObject create_movable_sphere(std::shared_ptr<const Material> m, Vec3 pos_a, Vec3 pos_b, double radius,
                                  double time_a, double time_b);
SceneObject create_xy_rect(std::shared_ptr<const Material> m, double x0, double x1, double y0, double y1, double k);
SceneObject create_xz_rect(std::shared_ptr<const Material> m, double x0, double x1, double z0, double z1, double k);
SceneObject create_yz_rect(std::shared_ptr<const Material> m, double y0, double y1, double z0, double z1, double k);
SceneObject create_box(std::shared_ptr<const Material> m, Vec3 mn, Vec3 mx);
SceneObject create_translate(Vec3 offset, SceneObject obj);
SceneObject create_rotate_y(double degrees, SceneObject obj);

// ---- background_color.rs ----
class BackgroundColor {
  public:
    virtual ~BackgroundColor() = default;
    virtual RtBackground describe() const = 0;
};
class Sky : public BackgroundColor {
  public:
    Sky() : top(1.0, 1.0, 1.0), bottom(0.5, 0.7, 1.0) {} // background_color.rs:18-25
    Sky(Color top, Color bottom) : top(top), bottom(bottom) {}
    RtBackground describe() const override;
    Color top, bottom;
};
class SolidBackgroundColor : public BackgroundColor {
  public:
    explicit SolidBackgroundColor(Color c) : color(c) {}
    RtBackground describe() const override;
    Color color;
};

// ---- scene.rs:109-118 ----
struct SceneLoadData {
    std::vector<SceneObject> objects;
    std::unique_ptr<BackgroundColor> background;
    std::optional<CameraConfig> camera;
    std::unique_ptr<ToneMap> tone_map; // null = "use the config's" (main.rs:84-86)
};

class SceneLoader {
  public:
    virtual ~SceneLoader() = default;
    virtual SceneLoadData load() const = 0;
};

class YmlLoader : public SceneLoader { // scene/yml.rs
  public:
    YmlLoader(std::string path, uint64_t seed) : path_(std::move(path)), seed_(seed) {}
    SceneLoadData load() const override;

  private:
    std::string path_;
    uint64_t seed_;
};

class SandboxLoader : public SceneLoader { // scene/sandbox.rs:39-80
  public:
    SandboxLoader(std::string cornell_path, uint64_t seed) : path_(std::move(cornell_path)), seed_(seed) {}
    SceneLoadData load() const override;

  private:
    std::string path_;
    uint64_t seed_;
};

// scene/random.rs: the "book cover" scene, 22 x 22 small spheres (diffuse ones
// move) + three big ones.  The reference draws it from thread_rng; here the
// draws come from the contract's RT_RNG_SCENE stream, in the reference's order,
// so a seed names one scene.
class RandomLoader : public SceneLoader {
  public:
    explicit RandomLoader(uint64_t seed) : seed_(seed) {}
    SceneLoadData load() const override;

  private:
    uint64_t seed_;
};

class NoneLoader : public SceneLoader { // scene/none.rs
  public:
    SceneLoadData load() const override;
};

// main.rs:74-79
std::unique_ptr<SceneLoader> make_loader(const SceneLoaderConfig &cfg, uint64_t seed);

// The POD scene the C ABI takes, with the storage that backs its pointers.
struct FlatScene {
    SceneFlattener tables;
    RtSceneDesc desc;
};
std::unique_ptr<FlatScene> flatten_scene(const SceneLoadData &data);

} // namespace rthost
