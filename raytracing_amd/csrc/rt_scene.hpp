// rt_scene.hpp -- host-side scene construction with the reference's names and
// argument order, plus the flattener that turns the object graph into the
// rt_scene buffers of include/rt_abi.h.
//
// Mirrors (reference file:line): Sphere::new hitable.rs:40, XYRect::new :151,
// XZRect::new :204, YZRect::new :278, FlipNormals::new_xy/xz/yz :330-338,
// Cuboid::new :380, Translate::new :433, RotateY::new :465, ConstantMedium::new
// :551 (commented out there), HitableList::new :100, BvhNode::new bvh.rs:106,
// Lambertian::new / new_from_color material.rs:57,61, Metal::new :93,
// Dielectric::new :121, DiffuseLight::new :189, Isotropic::new :213,
// Texture::new_constant / new_checker / new_noise texture.rs:36-52,
// Perlin::new perlin.rs:20-27, Camera::new camera.rs:17-51.
#ifndef RT_SCENE_HPP
#define RT_SCENE_HPP

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"

namespace rt {

struct Vec3 {
  double x, y, z;
  Vec3() : x(0), y(0), z(0) {}
  Vec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
  double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
  double& at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 scalar(double f) { return Vec3(f, f, f); }

// bvh.rs:9-12
struct Aabb {
  Vec3 min, max;
  Aabb() {}
  Aabb(Vec3 mn, Vec3 mx) : min(mn), max(mx) {}
};
Aabb surrounding_box(const Aabb& box0, const Aabb& box1);  // bvh.rs:80-94

struct SceneError : std::runtime_error {
  int code;
  SceneError(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

// Deterministic host generator standing in for rand::thread_rng() during scene
// construction (BVH split axes, Perlin gradients): splitmix64, uniform in [0,1).
class HostRng {
 public:
  explicit HostRng(uint64_t seed) : state_(seed) {}
  uint64_t next_u64();
  double gen_f64();  // 53-bit mantissa, like rand 0.7's Standard for f64
 private:
  uint64_t state_;
};

// Tables shared by everything built for one scene.
struct Tables {
  std::vector<rt_texture> textures;
  std::vector<rt_material> materials;
  std::vector<rt_perlin> perlins;
  std::vector<rt_light> lights;
};

// texture.rs:5-17.  A Texture is an index into the scene's texture table.
struct Texture {
  int id;
  static Texture new_constant(Tables& t, Vec3 color);
  static Texture new_checker(Tables& t, Texture odd, Texture even);
  static Texture new_noise(Tables& t, HostRng& rng, double scale);
};

// perlin.rs:67-100.
void perlin_generate(HostRng& rng, rt_perlin* out);

// material.rs:11-17.  A Material is an index into the scene's material table.
struct Material {
  int id;
};
struct Lambertian {
  static Material new_(Tables& t, Texture texture);
  static Material new_from_color(Tables& t, Vec3 albedo);
};
struct Metal { static Material new_(Tables& t, Vec3 albedo, double fuzz); };
struct Dielectric { static Material new_(Tables& t, double ref_idx); };
struct DiffuseLight { static Material new_(Tables& t, Texture emit); };
struct Isotropic { static Material new_(Tables& t, Texture albedo); };

// hitable.rs:19-30 (the construction-time half: bounding_box).
class Hitable {
 public:
  enum Kind { SPHERE, XY_RECT, XZ_RECT, YZ_RECT, FLIP_NORMALS, CUBOID, TRANSLATE, ROTATE_Y, CONSTANT_MEDIUM, HITABLE_LIST, BVH_NODE };
  virtual ~Hitable() {}
  virtual Kind kind() const = 0;
  virtual bool bounding_box(Aabb* out) const = 0;
};
typedef std::shared_ptr<Hitable> HitablePtr;

class Sphere : public Hitable {
 public:
  Vec3 center; double radius; Material material;
  Sphere(Vec3 c, double r, Material m) : center(c), radius(r), material(m) {}
  Kind kind() const override { return SPHERE; }
  bool bounding_box(Aabb* out) const override;  // hitable.rs:86-91
};

// One class for the three rects; `axis` is the fixed coordinate (2 = XYRect,
// 1 = XZRect, 0 = YZRect); a0,a1,b0,b1,k in constructor order.
class Rect : public Hitable {
 public:
  int axis; double a0, a1, b0, b1, k; Material material;
  Rect(int axis_, double a0_, double a1_, double b0_, double b1_, double k_, Material m)
      : axis(axis_), a0(a0_), a1(a1_), b0(b0_), b1(b1_), k(k_), material(m) {}
  Kind kind() const override { return axis == 2 ? XY_RECT : (axis == 1 ? XZ_RECT : YZ_RECT); }
  bool bounding_box(Aabb* out) const override;  // hitable.rs:185-190, 238-243, 312-317
};
inline HitablePtr XYRect(double x0, double x1, double y0, double y1, double k, Material m) { return std::make_shared<Rect>(2, x0, x1, y0, y1, k, m); }
inline HitablePtr XZRect(double x0, double x1, double z0, double z1, double k, Material m) { return std::make_shared<Rect>(1, x0, x1, z0, z1, k, m); }
inline HitablePtr YZRect(double y0, double y1, double z0, double z1, double k, Material m) { return std::make_shared<Rect>(0, y0, y1, z0, z1, k, m); }

// hitable.rs:320-361: wraps one of the three rects only.
class FlipNormals : public Hitable {
 public:
  std::shared_ptr<Rect> rect;
  explicit FlipNormals(HitablePtr h);
  Kind kind() const override { return FLIP_NORMALS; }
  bool bounding_box(Aabb* out) const override { return rect->bounding_box(out); }
};

// hitable.rs:363-424: six rects in source order, top face lifted by 0.01.
class Cuboid : public Hitable {
 public:
  Vec3 pmin, pmax; std::vector<HitablePtr> list;
  Cuboid(Vec3 p0, Vec3 p1, Material m);
  Kind kind() const override { return CUBOID; }
  bool bounding_box(Aabb* out) const override { *out = Aabb(pmin, pmax); return true; }
};

class Translate : public Hitable {
 public:
  HitablePtr hitable; Vec3 offset;
  Translate(HitablePtr h, Vec3 off) : hitable(h), offset(off) {}
  Kind kind() const override { return TRANSLATE; }
  bool bounding_box(Aabb* out) const override;  // hitable.rs:447-452
};

class RotateY : public Hitable {
 public:
  HitablePtr hitable; double sin_theta, cos_theta; bool hasbox; Aabb boxy;
  RotateY(HitablePtr h, double angle_degrees);  // hitable.rs:465-505
  Kind kind() const override { return ROTATE_Y; }
  bool bounding_box(Aabb* out) const override { if (hasbox) *out = boxy; return hasbox; }
};

class ConstantMedium : public Hitable {
 public:
  HitablePtr boundary; double density; Material phase_function;
  ConstantMedium(Tables& t, HitablePtr b, double d, Texture a) : boundary(b), density(d), phase_function(Isotropic::new_(t, a)) {}
  Kind kind() const override { return CONSTANT_MEDIUM; }
  bool bounding_box(Aabb* out) const override { return boundary->bounding_box(out); }
};

class HitableList : public Hitable {
 public:
  std::vector<HitablePtr> list;
  explicit HitableList(std::vector<HitablePtr> l) : list(std::move(l)) {}
  Kind kind() const override { return HITABLE_LIST; }
  bool bounding_box(Aabb* out) const override;  // hitable.rs:119-128
};

// bvh.rs:97-147: random axis per node, sort by box minimum on it, median split,
// one hitable per leaf.
class BvhNode : public Hitable {
 public:
  HitablePtr contents; std::shared_ptr<BvhNode> left, right; Aabb boxy;
  BvhNode(std::vector<HitablePtr> l, HostRng& rng);
  Kind kind() const override { return BVH_NODE; }
  bool bounding_box(Aabb* out) const override { *out = boxy; return true; }
};

// The flattened scene; `view()` points into the vectors.
struct FlatScene {
  std::vector<rt_node> nodes;
  std::vector<rt_prim> prims;
  std::vector<rt_object> objects;
  Tables tables;
  rt_scene view() const;
};

// What src/main.rs:85-90 does per thread (a BvhNode over world.list), then the
// flattening: breadth-first nodes, primitives in leaf order, objects for
// anything wrapped in Translate/RotateY/ConstantMedium or holding a sub-tree.
void flatten_world(const HitableList& world, const Tables& tables, HostRng& rng, FlatScene* out);

}  // namespace rt

#endif
