// rt_scene.cpp -- host scene construction, BVH build and flattening.
// See rt_scene.hpp for the reference lines each class follows.
#include "rt_scene.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <deque>

#include "../../include/rt_scene_api.h"
#include "rt_internal.hpp"

namespace rt {

namespace {
thread_local std::string g_last_error;
const double kPi = 3.14159265358979323846264338327950288;
}  // namespace

int set_error(int code, const std::string& text) {
  g_last_error = text;
  return code;
}

// ---- bvh.rs:64-94 ----
static double ffmin(double a, double b) { return a < b ? a : b; }
static double ffmax(double a, double b) { return a > b ? a : b; }
Aabb surrounding_box(const Aabb& box0, const Aabb& box1) {
  return Aabb(Vec3(ffmin(box0.min.x, box1.min.x), ffmin(box0.min.y, box1.min.y), ffmin(box0.min.z, box1.min.z)),
              Vec3(ffmax(box0.max.x, box1.max.x), ffmax(box0.max.y, box1.max.y), ffmax(box0.max.z, box1.max.z)));
}

uint64_t HostRng::next_u64() {
  uint64_t z = (state_ += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
double HostRng::gen_f64() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }

// ---- texture.rs:36-52 ----
Texture Texture::new_constant(Tables& t, Vec3 color) {
  rt_texture x;
  std::memset(&x, 0, sizeof(x));
  x.type = RT_TEX_CONSTANT;
  x.color[0] = (float)color.x; x.color[1] = (float)color.y; x.color[2] = (float)color.z;
  x.odd = x.even = x.perlin = -1;
  t.textures.push_back(x);
  return Texture{(int)t.textures.size() - 1};
}
Texture Texture::new_checker(Tables& t, Texture odd, Texture even) {
  rt_texture x;
  std::memset(&x, 0, sizeof(x));
  x.type = RT_TEX_CHECKER;
  x.odd = odd.id; x.even = even.id; x.perlin = -1;
  t.textures.push_back(x);
  return Texture{(int)t.textures.size() - 1};
}
Texture Texture::new_noise(Tables& t, HostRng& rng, double scale) {
  rt_perlin p;
  perlin_generate(rng, &p);
  t.perlins.push_back(p);
  rt_texture x;
  std::memset(&x, 0, sizeof(x));
  x.type = RT_TEX_NOISE;
  x.odd = x.even = -1;
  x.scale = (float)scale;
  x.perlin = (int)t.perlins.size() - 1;
  t.textures.push_back(x);
  return Texture{(int)t.textures.size() - 1};
}

// ---- perlin.rs:67-100 ----
// permute() computes `(rng.gen::<f64>() as usize) * (i + 1)`: the cast comes
// before the multiply, so the swap target is always 0.  Kept as written.
static void perlin_generate_perm(HostRng& rng, int32_t* p) {
  for (int i = 0; i < 256; ++i) p[i] = i;
  const int n = 256;
  for (int ii = 0; ii < n; ++ii) {
    const int i = n - ii - 1;
    const size_t target = (size_t)rng.gen_f64() * (size_t)(i + 1);
    const int32_t tmp = p[i];
    p[i] = p[target];
    p[target] = tmp;
  }
}
void perlin_generate(HostRng& rng, rt_perlin* out) {
  for (int i = 0; i < 256; ++i) {
    const double x = -1.0 + 2.0 * rng.gen_f64();
    const double y = -1.0 + 2.0 * rng.gen_f64();
    const double z = -1.0 + 2.0 * rng.gen_f64();
    const double k = 1.0 / std::sqrt(x * x + y * y + z * z);  // vec3.rs:188-190
    out->ranvec[i][0] = (float)(x * k);
    out->ranvec[i][1] = (float)(y * k);
    out->ranvec[i][2] = (float)(z * k);
  }
  perlin_generate_perm(rng, out->perm_x);
  perlin_generate_perm(rng, out->perm_y);
  perlin_generate_perm(rng, out->perm_z);
}

// ---- material.rs constructors ----
static Material push_material(Tables& t, int type, int texture, Vec3 albedo, double fuzz, double ref_idx) {
  rt_material m;
  std::memset(&m, 0, sizeof(m));
  m.type = type;
  m.texture = texture;
  m.albedo[0] = (float)albedo.x; m.albedo[1] = (float)albedo.y; m.albedo[2] = (float)albedo.z;
  m.fuzz = (float)fuzz;
  m.ref_idx = (float)ref_idx;
  t.materials.push_back(m);
  return Material{(int)t.materials.size() - 1};
}
Material Lambertian::new_(Tables& t, Texture texture) { return push_material(t, RT_MAT_LAMBERTIAN, texture.id, Vec3(), 0, 0); }
Material Lambertian::new_from_color(Tables& t, Vec3 albedo) { return new_(t, Texture::new_constant(t, albedo)); }
Material Metal::new_(Tables& t, Vec3 albedo, double fuzz) { return push_material(t, RT_MAT_METAL, -1, albedo, fuzz < 1.0 ? fuzz : 1.0, 0); }
Material Dielectric::new_(Tables& t, double ref_idx) { return push_material(t, RT_MAT_DIELECTRIC, -1, Vec3(), 0, ref_idx); }
Material DiffuseLight::new_(Tables& t, Texture emit) { return push_material(t, RT_MAT_DIFFUSE_LIGHT, emit.id, Vec3(), 0, 0); }
Material Isotropic::new_(Tables& t, Texture albedo) { return push_material(t, RT_MAT_ISOTROPIC, albedo.id, Vec3(), 0, 0); }

// ---- bounding boxes ----
bool Sphere::bounding_box(Aabb* out) const {
  *out = Aabb(center - scalar(radius), center + scalar(radius));
  return true;
}
bool Rect::bounding_box(Aabb* out) const {
  if (axis == 2) *out = Aabb(Vec3(a0, b0, k - 0.0001), Vec3(a1, b1, k + 0.0001));
  else if (axis == 1) *out = Aabb(Vec3(a0, k - 0.0001, b0), Vec3(a1, k + 0.0001, b1));
  else *out = Aabb(Vec3(k - 0.0001, a0, b0), Vec3(k + 0.0001, a1, b1));
  return true;
}

FlipNormals::FlipNormals(HitablePtr h) {
  rect = std::dynamic_pointer_cast<Rect>(h);
  if (!rect) throw SceneError(RT_ERR_INVALID, "FlipNormals wraps XYRect, XZRect or YZRect only (hitable.rs:320-325)");
}

// hitable.rs:380-423
Cuboid::Cuboid(Vec3 p0, Vec3 p1, Material m) : pmin(p0), pmax(p1) {
  list.push_back(XYRect(p0.x, p1.x, p0.y, p1.y, p1.z, m));
  list.push_back(std::make_shared<FlipNormals>(XYRect(p0.x, p1.x, p0.y, p1.y, p0.z, m)));
  list.push_back(XZRect(p0.x, p1.x, p0.z, p1.z, p1.y + 0.01, m));
  list.push_back(std::make_shared<FlipNormals>(XZRect(p0.x, p1.x, p0.z, p1.z, p0.y, m)));
  list.push_back(YZRect(p0.y, p1.y, p0.z, p1.z, p1.x, m));
  list.push_back(std::make_shared<FlipNormals>(YZRect(p0.y, p1.y, p0.z, p1.z, p0.x, m)));
}

bool Translate::bounding_box(Aabb* out) const {
  Aabb b;
  if (!hitable->bounding_box(&b)) return false;
  *out = Aabb(b.min + offset, b.max + offset);
  return true;
}

// hitable.rs:465-505
RotateY::RotateY(HitablePtr h, double angle) : hitable(h) {
  const double radians = (kPi / 180.0) * angle;
  sin_theta = std::sin(radians);
  cos_theta = std::cos(radians);
  Aabb bbox;
  hasbox = hitable->bounding_box(&bbox);
  if (!hasbox) throw SceneError(RT_ERR_INVALID, "RotateY::new needs a child with a bounding box (hitable.rs:476)");
  // The box of the rotated child: a turn about y leaves heights alone, so it is the child's y range over the x/z extent
  // of the box's four rotated vertical edges (the same eight products and sums the source makes for its eight corners).
  const double edge_x[2] = {bbox.min.x, bbox.max.x}, edge_z[2] = {bbox.min.z, bbox.max.z};
  double lo_x = DBL_MAX, hi_x = -DBL_MAX, lo_z = DBL_MAX, hi_z = -DBL_MAX;
  for (int xi = 0; xi < 2; ++xi)
    for (int zi = 0; zi < 2; ++zi) {
      const double turned_x = cos_theta * edge_x[xi] + sin_theta * edge_z[zi];
      const double turned_z = -sin_theta * edge_x[xi] + cos_theta * edge_z[zi];
      lo_x = std::min(lo_x, turned_x); hi_x = std::max(hi_x, turned_x);
      lo_z = std::min(lo_z, turned_z); hi_z = std::max(hi_z, turned_z);
    }
  const Vec3 mn(lo_x, std::min(bbox.min.y, bbox.max.y), lo_z), mx(hi_x, std::max(bbox.min.y, bbox.max.y), hi_z);
  boxy = Aabb(mn, mx);
}

bool HitableList::bounding_box(Aabb* out) const {
  if (list.empty()) return false;
  Aabb boxy;
  if (!list[0]->bounding_box(&boxy)) return false;
  for (size_t i = 1; i < list.size(); ++i) {
    Aabb b;
    if (!list[i]->bounding_box(&b)) return false;
    boxy = surrounding_box(boxy, b);
  }
  *out = boxy;
  return true;
}

// bvh.rs:105-147.  The reference's comparators never return Equal
// (bvh.rs:158-162); a strict `<` with a stable sort is used instead so the
// result is the same on every platform.
BvhNode::BvhNode(std::vector<HitablePtr> l, HostRng& rng) {
  if (l.empty()) throw SceneError(RT_ERR_INVALID, "BvhNode::new over an empty list");
  const int axis = (int)(3.0 * rng.gen_f64());
  std::vector<std::pair<double, HitablePtr> > keyed;
  for (size_t i = 0; i < l.size(); ++i) {
    Aabb b;
    if (!l[i]->bounding_box(&b)) throw SceneError(RT_ERR_INVALID, "No bounding box in BvhNode::new() (bvh.rs:153-156)");
    keyed.push_back(std::make_pair(b.min[axis], l[i]));
  }
  std::stable_sort(keyed.begin(), keyed.end(),
                   [](const std::pair<double, HitablePtr>& a, const std::pair<double, HitablePtr>& b) { return a.first < b.first; });
  for (size_t i = 0; i < l.size(); ++i) l[i] = keyed[i].second;
  if (l.size() == 1) {
    contents = l[0];
    if (!l[0]->bounding_box(&boxy)) throw SceneError(RT_ERR_INVALID, "No bounding box in BvhNode::new() 1 (bvh.rs:124)");
  } else {
    const size_t half = l.size() / 2;
    left = std::make_shared<BvhNode>(std::vector<HitablePtr>(l.begin(), l.begin() + half), rng);
    right = std::make_shared<BvhNode>(std::vector<HitablePtr>(l.begin() + half, l.end()), rng);
    boxy = surrounding_box(left->boxy, right->boxy);
  }
}

// ---------------------------------------------------------------------------
// Flattening
// ---------------------------------------------------------------------------
rt_scene FlatScene::view() const {
  rt_scene s;
  std::memset(&s, 0, sizeof(s));
  s.nodes = nodes.data(); s.n_nodes = (int32_t)nodes.size();
  s.prims = prims.data(); s.n_prims = (int32_t)prims.size();
  s.objects = objects.data(); s.n_objects = (int32_t)objects.size();
  s.materials = tables.materials.data(); s.n_materials = (int32_t)tables.materials.size();
  s.textures = tables.textures.data(); s.n_textures = (int32_t)tables.textures.size();
  s.perlins = tables.perlins.data(); s.n_perlins = (int32_t)tables.perlins.size();
  s.lights = tables.lights.data(); s.n_lights = (int32_t)tables.lights.size();
  return s;
}

namespace {

struct Flattener {
  FlatScene* out;
  int n_media;
  std::vector<std::pair<int, const BvhNode*> > pending;  // object index -> sub-tree to emit

  int nesting;
  explicit Flattener(FlatScene* o) : out(o), n_media(0), nesting(0) {}

  void push_rect(const Rect* r, bool flip) {
    rt_prim p;
    std::memset(&p, 0, sizeof(p));
    p.type = (r->axis == 2 ? RT_PRIM_XY_RECT : (r->axis == 1 ? RT_PRIM_XZ_RECT : RT_PRIM_YZ_RECT)) | (flip ? RT_PRIM_FLIP : 0);
    p.material = r->material.id;
    p.f[0] = (float)r->a0; p.f[1] = (float)r->a1; p.f[2] = (float)r->b0; p.f[3] = (float)r->b1; p.f[4] = (float)r->k;
    out->prims.push_back(p);
  }

  // Appends the primitives of a hitable made only of spheres, rects, flipped
  // rects, cuboids and lists of those.  Returns false (emitting nothing) otherwise.
  bool is_simple(const Hitable* h) const {
    switch (h->kind()) {
      case Hitable::SPHERE: case Hitable::XY_RECT: case Hitable::XZ_RECT: case Hitable::YZ_RECT:
      case Hitable::FLIP_NORMALS: case Hitable::CUBOID:
        return true;
      case Hitable::HITABLE_LIST: {
        const HitableList* l = static_cast<const HitableList*>(h);
        if (l->list.empty()) return false;
        for (size_t i = 0; i < l->list.size(); ++i)
          if (!is_simple(l->list[i].get())) return false;
        return true;
      }
      default:
        return false;
    }
  }
  void emit_simple(const Hitable* h) {
    switch (h->kind()) {
      case Hitable::SPHERE: {
        const Sphere* s = static_cast<const Sphere*>(h);
        rt_prim p;
        std::memset(&p, 0, sizeof(p));
        p.type = RT_PRIM_SPHERE;
        p.material = s->material.id;
        p.f[0] = (float)s->center.x; p.f[1] = (float)s->center.y; p.f[2] = (float)s->center.z; p.f[3] = (float)s->radius;
        out->prims.push_back(p);
        break;
      }
      case Hitable::XY_RECT: case Hitable::XZ_RECT: case Hitable::YZ_RECT:
        push_rect(static_cast<const Rect*>(h), false);
        break;
      case Hitable::FLIP_NORMALS:
        push_rect(static_cast<const FlipNormals*>(h)->rect.get(), true);
        break;
      case Hitable::CUBOID: {
        const Cuboid* c = static_cast<const Cuboid*>(h);
        for (size_t i = 0; i < c->list.size(); ++i) emit_simple(c->list[i].get());
        break;
      }
      case Hitable::HITABLE_LIST: {
        const HitableList* l = static_cast<const HitableList*>(h);
        for (size_t i = 0; i < l->list.size(); ++i) emit_simple(l->list[i].get());
        break;
      }
      default:
        break;
    }
  }

  // Fills the leaf fields of `node` for the hitable a BvhNode leaf holds.  Anything that is not plain primitives
  // becomes an object: [ConstantMedium(] [Translate(] [RotateY(] inner, where inner is primitives (RT_OBJ_LIST /
  // RT_OBJ_MEDIUM), a BvhNode (RT_OBJ_BVH / RT_OBJ_MEDIUM_BVH: its tree follows later) or anything else the
  // constructors allow (hitable.rs:433,465 take any Box<dyn Hitable>): a further Translate or RotateY in any order, a
  // HitableList holding such things.  That last case becomes a sub-tree too, made here: one leaf per item, in the list's
  // order, every box infinite (no such box test exists in the reference: Translate::hit hands the ray straight to its
  // child), and each leaf an object of its own, so `RotateY(Translate(x))` is an object in an object.
  // `top_level`: the hitable is an item of the world itself.  A ConstantMedium must be one: the kernels walk the media
  // after the tree, in the world's frame.
  void emit_leaf(const Hitable* h, bool top_level, rt_node* node) {
    if (is_simple(h)) {
      const int first = (int)out->prims.size();
      emit_simple(h);
      node->left = ~first;
      node->right = (int)out->prims.size() - first;
      return;
    }
    rt_object ob;
    std::memset(&ob, 0, sizeof(ob));
    ob.cos_theta = 1.0f;
    ob.material = -1;
    ob.medium_index = -1;
    const Hitable* cur = h;
    bool medium = false;
    if (cur->kind() == Hitable::CONSTANT_MEDIUM) {
      if (!top_level)
        throw SceneError(RT_ERR_UNSUPPORTED, "a ConstantMedium must be an item of the world's list: inside a Translate, a RotateY, a HitableList or a BvhNode it is not supported");
      const ConstantMedium* m = static_cast<const ConstantMedium*>(cur);
      medium = true;
      ob.density = (float)m->density;
      ob.material = m->phase_function.id;
      ob.medium_index = n_media++;
      cur = m->boundary.get();
    }
    if (cur->kind() == Hitable::TRANSLATE) {
      const Translate* t = static_cast<const Translate*>(cur);
      ob.xform |= RT_XFORM_TRANSLATE;
      ob.offset[0] = (float)t->offset.x; ob.offset[1] = (float)t->offset.y; ob.offset[2] = (float)t->offset.z;
      cur = t->hitable.get();
    }
    if (cur->kind() == Hitable::ROTATE_Y) {
      const RotateY* r = static_cast<const RotateY*>(cur);
      ob.xform |= RT_XFORM_ROTATE_Y;
      ob.sin_theta = (float)r->sin_theta;
      ob.cos_theta = (float)r->cos_theta;
      cur = r->hitable.get();
    }
    // (the object takes its place in the array before its inside is emitted: the inside may add objects of its own)
    const int obj_index = (int)out->objects.size();
    out->objects.push_back(ob);
    if (is_simple(cur)) {
      ob.kind = medium ? RT_OBJ_MEDIUM : RT_OBJ_LIST;
      ob.first = (int)out->prims.size();
      emit_simple(cur);
      ob.count = (int)out->prims.size() - ob.first;
    } else if (cur->kind() == Hitable::BVH_NODE) {
      ob.kind = medium ? RT_OBJ_MEDIUM_BVH : RT_OBJ_BVH;
      ob.first = -1;  // patched when the sub-tree is emitted
      pending.push_back(std::make_pair(obj_index, static_cast<const BvhNode*>(cur)));
    } else {
      if (++nesting > 64) throw SceneError(RT_ERR_UNSUPPORTED, "hitables nested more than 64 deep");
      ob.kind = medium ? RT_OBJ_MEDIUM_BVH : RT_OBJ_BVH;
      ob.first = emit_items(cur);
      --nesting;
    }
    out->objects[(size_t)obj_index] = ob;
    node->left = ~obj_index;
    node->right = 0;
  }

  // A tree over the items of a HitableList in the list's order (or over the one hitable), every box infinite; returns its root.
  int emit_items(const Hitable* h) {
    std::vector<const Hitable*> items;
    if (h->kind() == Hitable::HITABLE_LIST) {
      const HitableList* l = static_cast<const HitableList*>(h);
      for (size_t i = 0; i < l->list.size(); ++i) items.push_back(l->list[i].get());
      if (items.empty()) throw SceneError(RT_ERR_INVALID, "an empty HitableList inside another hitable");
    } else {
      items.push_back(h);
    }
    return emit_item_range(items, 0, items.size());
  }
  int emit_item_range(const std::vector<const Hitable*>& items, size_t begin, size_t end) {
    const int at = (int)out->nodes.size();
    out->nodes.push_back(rt_node());
    rt_node node;
    std::memset(&node, 0, sizeof(node));
    for (int c = 0; c < 3; ++c) { node.bmin[c] = -INFINITY; node.bmax[c] = INFINITY; }
    if (end - begin == 1) {
      emit_leaf(items[begin], false, &node);
    } else {
      const size_t mid = begin + (end - begin) / 2;
      node.left = emit_item_range(items, begin, mid);
      node.right = emit_item_range(items, mid, end);
    }
    out->nodes[(size_t)at] = node;
    return at;
  }

  // The largest float not above v / the smallest not below v.
  static float float_below(double v) {
    float f = (float)v;
    if ((double)f > v) f = std::nextafter(f, -INFINITY);
    return f;
  }
  static float float_above(double v) {
    float f = (float)v;
    if ((double)f < v) f = std::nextafter(f, INFINITY);
    return f;
  }

  // Breadth-first emission of one tree; returns the index of its root.
  int emit_tree(const BvhNode* root, bool top_level) {
    std::deque<std::pair<const BvhNode*, int> > queue;
    const int root_index = (int)out->nodes.size();
    out->nodes.push_back(rt_node());
    queue.push_back(std::make_pair(root, root_index));
    while (!queue.empty()) {
      const BvhNode* n = queue.front().first;
      const int at = queue.front().second;
      queue.pop_front();
      rt_node node;
      std::memset(&node, 0, sizeof(node));
      // The f64 box goes to fp32 rounded outwards, so the fp32 box contains it: rounded to nearest, the 0.0001 a rect's box
      // is padded with (hitable.rs:185-190) disappears from |k| = 2048 on and the slab test rejects the zero-width box.
      node.bmin[0] = float_below(n->boxy.min.x); node.bmin[1] = float_below(n->boxy.min.y); node.bmin[2] = float_below(n->boxy.min.z);
      node.bmax[0] = float_above(n->boxy.max.x); node.bmax[1] = float_above(n->boxy.max.y); node.bmax[2] = float_above(n->boxy.max.z);
      if (n->contents) {
        emit_leaf(n->contents.get(), top_level, &node);
      } else {
        node.left = (int)out->nodes.size();
        node.right = node.left + 1;
        out->nodes.push_back(rt_node());
        out->nodes.push_back(rt_node());
        queue.push_back(std::make_pair(n->left.get(), node.left));
        queue.push_back(std::make_pair(n->right.get(), node.right));
      }
      out->nodes[at] = node;
    }
    return root_index;
  }
};

}  // namespace

void flatten_world(const HitableList& world, const Tables& tables, HostRng& rng, FlatScene* out) {
  out->nodes.clear();
  out->prims.clear();
  out->objects.clear();
  out->tables = tables;
  // main.rs:85-90: a BvhNode over the items of world.list.  An item that is itself a BvhNode (or holds one under
  // Translate / RotateY) becomes a leaf holding a sub-tree; sub-trees may hold further ones.
  BvhNode top(world.list, rng);
  Flattener f(out);
  f.emit_tree(&top, true);
  for (size_t i = 0; i < f.pending.size(); ++i) {
    const int at = f.pending[i].first;
    const int root = f.emit_tree(f.pending[i].second, false);
    out->objects[(size_t)at].first = root;
  }
}

}  // namespace rt

// ---------------------------------------------------------------------------
// C surface
// ---------------------------------------------------------------------------
struct rt_builder {
  rt::Tables tables;
  rt::HostRng rng;
  std::vector<rt::HitablePtr> hitables;
  rt::FlatScene flat;
  explicit rt_builder(uint64_t seed) : rng(seed) {}
};

namespace {

template <class F> int guarded(F f) {
  try {
    return f();
  } catch (const rt::SceneError& e) {
    return rt::set_error(e.code, e.what());
  } catch (const std::exception& e) {
    return rt::set_error(RT_ERR_INVALID, e.what());
  }
}

bool bad_texture(rt_builder* b, int t) { return t < 0 || t >= (int)b->tables.textures.size(); }
bool bad_material(rt_builder* b, int m) { return m < 0 || m >= (int)b->tables.materials.size(); }
bool bad_hitable(rt_builder* b, int h) { return h < 0 || h >= (int)b->hitables.size(); }

int push(rt_builder* b, rt::HitablePtr h) {
  b->hitables.push_back(h);
  return (int)b->hitables.size() - 1;
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return rt::g_last_error.c_str(); }

int rt_camera_init(rt_camera* out, const double lookfrom[3], const double lookat[3], const double vup[3],
                   double vfov, double aspect, double aperture, double focus_dist) {
  if (!out || !lookfrom || !lookat || !vup) return rt::set_error(RT_ERR_INVALID, "rt_camera_init: null argument");
  // camera.rs:26-50
  const double lens_radius = aperture / 2.0;
  const double theta = vfov * rt::kPi / 180.0;
  const double half_height = std::tan(theta / 2.0);
  const double half_width = aspect * half_height;
  const double ox = lookfrom[0], oy = lookfrom[1], oz = lookfrom[2];
  // w = (lookfrom - lookat).unit()
  double wx = lookfrom[0] - lookat[0], wy = lookfrom[1] - lookat[1], wz = lookfrom[2] - lookat[2];
  double k = 1.0 / std::sqrt(wx * wx + wy * wy + wz * wz);
  wx *= k; wy *= k; wz *= k;
  // u = vup.cross(w).unit()   (vec3.rs:180-186)
  double ux = vup[1] * wz - vup[2] * wy, uy = -(vup[0] * wz - vup[2] * wx), uz = vup[0] * wy - vup[1] * wx;
  k = 1.0 / std::sqrt(ux * ux + uy * uy + uz * uz);
  ux *= k; uy *= k; uz *= k;
  // v = w.cross(u)
  const double vx = wy * uz - wz * uy, vy = -(wx * uz - wz * ux), vz = wx * uy - wy * ux;
  const double hw = half_width * focus_dist, hh = half_height * focus_dist;
  out->lower_left_corner[0] = ox - hw * ux - hh * vx - focus_dist * wx;
  out->lower_left_corner[1] = oy - hw * uy - hh * vy - focus_dist * wy;
  out->lower_left_corner[2] = oz - hw * uz - hh * vz - focus_dist * wz;
  out->horizontal[0] = 2.0 * hw * ux; out->horizontal[1] = 2.0 * hw * uy; out->horizontal[2] = 2.0 * hw * uz;
  out->vertical[0] = 2.0 * hh * vx; out->vertical[1] = 2.0 * hh * vy; out->vertical[2] = 2.0 * hh * vz;
  out->origin[0] = ox; out->origin[1] = oy; out->origin[2] = oz;
  out->u[0] = ux; out->u[1] = uy; out->u[2] = uz;
  out->v[0] = vx; out->v[1] = vy; out->v[2] = vz;
  out->w[0] = wx; out->w[1] = wy; out->w[2] = wz;
  out->lens_radius = lens_radius;
  return RT_OK;
}

int rtb_create(uint64_t seed, rt_builder** out) {
  if (!out) return rt::set_error(RT_ERR_INVALID, "rtb_create: null out");
  *out = new rt_builder(seed);
  return RT_OK;
}
int rtb_destroy(rt_builder* b) {
  delete b;
  return RT_OK;
}

int rtb_texture_constant(rt_builder* b, double r, double g, double bl) {
  return guarded([&]() { return rt::Texture::new_constant(b->tables, rt::Vec3(r, g, bl)).id; });
}
int rtb_texture_checker(rt_builder* b, int odd, int even) {
  if (bad_texture(b, odd) || bad_texture(b, even)) return rt::set_error(RT_ERR_INVALID, "rtb_texture_checker: bad texture handle");
  return guarded([&]() { return rt::Texture::new_checker(b->tables, rt::Texture{odd}, rt::Texture{even}).id; });
}
int rtb_texture_noise(rt_builder* b, double scale) {
  return guarded([&]() { return rt::Texture::new_noise(b->tables, b->rng, scale).id; });
}

int rtb_lambertian(rt_builder* b, int texture) {
  if (bad_texture(b, texture)) return rt::set_error(RT_ERR_INVALID, "rtb_lambertian: bad texture handle");
  return rt::Lambertian::new_(b->tables, rt::Texture{texture}).id;
}
int rtb_metal(rt_builder* b, double r, double g, double bl, double fuzz) { return rt::Metal::new_(b->tables, rt::Vec3(r, g, bl), fuzz).id; }
int rtb_dielectric(rt_builder* b, double ref_idx) { return rt::Dielectric::new_(b->tables, ref_idx).id; }
int rtb_diffuse_light(rt_builder* b, int texture) {
  if (bad_texture(b, texture)) return rt::set_error(RT_ERR_INVALID, "rtb_diffuse_light: bad texture handle");
  return rt::DiffuseLight::new_(b->tables, rt::Texture{texture}).id;
}
int rtb_isotropic(rt_builder* b, int texture) {
  if (bad_texture(b, texture)) return rt::set_error(RT_ERR_INVALID, "rtb_isotropic: bad texture handle");
  return rt::Isotropic::new_(b->tables, rt::Texture{texture}).id;
}

int rtb_sphere(rt_builder* b, const double c[3], double radius, int material) {
  if (bad_material(b, material)) return rt::set_error(RT_ERR_INVALID, "rtb_sphere: bad material handle");
  return push(b, std::make_shared<rt::Sphere>(rt::Vec3(c[0], c[1], c[2]), radius, rt::Material{material}));
}
int rtb_xy_rect(rt_builder* b, double x0, double x1, double y0, double y1, double k, int material) {
  if (bad_material(b, material)) return rt::set_error(RT_ERR_INVALID, "rtb_xy_rect: bad material handle");
  return push(b, rt::XYRect(x0, x1, y0, y1, k, rt::Material{material}));
}
int rtb_xz_rect(rt_builder* b, double x0, double x1, double z0, double z1, double k, int material) {
  if (bad_material(b, material)) return rt::set_error(RT_ERR_INVALID, "rtb_xz_rect: bad material handle");
  return push(b, rt::XZRect(x0, x1, z0, z1, k, rt::Material{material}));
}
int rtb_yz_rect(rt_builder* b, double y0, double y1, double z0, double z1, double k, int material) {
  if (bad_material(b, material)) return rt::set_error(RT_ERR_INVALID, "rtb_yz_rect: bad material handle");
  return push(b, rt::YZRect(y0, y1, z0, z1, k, rt::Material{material}));
}
int rtb_flip_normals(rt_builder* b, int rect) {
  if (bad_hitable(b, rect)) return rt::set_error(RT_ERR_INVALID, "rtb_flip_normals: bad hitable handle");
  return guarded([&]() { return push(b, std::make_shared<rt::FlipNormals>(b->hitables[rect])); });
}
int rtb_cuboid(rt_builder* b, const double p0[3], const double p1[3], int material) {
  if (bad_material(b, material)) return rt::set_error(RT_ERR_INVALID, "rtb_cuboid: bad material handle");
  return push(b, std::make_shared<rt::Cuboid>(rt::Vec3(p0[0], p0[1], p0[2]), rt::Vec3(p1[0], p1[1], p1[2]), rt::Material{material}));
}
int rtb_translate(rt_builder* b, int hitable, const double off[3]) {
  if (bad_hitable(b, hitable)) return rt::set_error(RT_ERR_INVALID, "rtb_translate: bad hitable handle");
  return push(b, std::make_shared<rt::Translate>(b->hitables[hitable], rt::Vec3(off[0], off[1], off[2])));
}
int rtb_rotate_y(rt_builder* b, int hitable, double angle) {
  if (bad_hitable(b, hitable)) return rt::set_error(RT_ERR_INVALID, "rtb_rotate_y: bad hitable handle");
  return guarded([&]() { return push(b, std::make_shared<rt::RotateY>(b->hitables[hitable], angle)); });
}
int rtb_constant_medium(rt_builder* b, int boundary, double density, int texture) {
  if (bad_hitable(b, boundary)) return rt::set_error(RT_ERR_INVALID, "rtb_constant_medium: bad hitable handle");
  if (bad_texture(b, texture)) return rt::set_error(RT_ERR_INVALID, "rtb_constant_medium: bad texture handle");
  return push(b, std::make_shared<rt::ConstantMedium>(b->tables, b->hitables[boundary], density, rt::Texture{texture}));
}
static int collect(rt_builder* b, const int* hs, int n, std::vector<rt::HitablePtr>* out) {
  if (n < 0 || (n > 0 && !hs)) return rt::set_error(RT_ERR_INVALID, "bad hitable array");
  for (int i = 0; i < n; ++i) {
    if (bad_hitable(b, hs[i])) return rt::set_error(RT_ERR_INVALID, "bad hitable handle in array");
    out->push_back(b->hitables[hs[i]]);
  }
  return RT_OK;
}
int rtb_hitable_list(rt_builder* b, const int* hs, int n) {
  std::vector<rt::HitablePtr> l;
  const int rc = collect(b, hs, n, &l);
  if (rc != RT_OK) return rc;
  return push(b, std::make_shared<rt::HitableList>(l));
}
int rtb_bvh_node(rt_builder* b, const int* hs, int n) {
  std::vector<rt::HitablePtr> l;
  const int rc = collect(b, hs, n, &l);
  if (rc != RT_OK) return rc;
  return guarded([&]() { return push(b, std::make_shared<rt::BvhNode>(l, b->rng)); });
}

int rtb_light_xz_rect(rt_builder* b, double x0, double x1, double z0, double z1, double k) {
  rt_light l;
  std::memset(&l, 0, sizeof(l));
  l.type = RT_LIGHT_XZ_RECT;
  l.f[0] = (float)x0; l.f[1] = (float)x1; l.f[2] = (float)z0; l.f[3] = (float)z1; l.f[4] = (float)k;
  b->tables.lights.clear();  // color() samples one shape (main.rs:177-178): registering again replaces it
  b->tables.lights.push_back(l);
  return (int)b->tables.lights.size() - 1;
}

int rtb_light_from_world(rt_builder* b, int world_list) {
  if (bad_hitable(b, world_list) || b->hitables[world_list]->kind() != rt::Hitable::HITABLE_LIST)
    return rt::set_error(RT_ERR_INVALID, "rtb_light_from_world: world must be a HitableList handle");
  const rt::HitableList* world = static_cast<const rt::HitableList*>(b->hitables[world_list].get());
  int found = 0;
  for (const rt::HitablePtr& h : world->list) {
    const rt::Rect* rect = nullptr;
    if (h->kind() == rt::Hitable::XZ_RECT) rect = static_cast<const rt::Rect*>(h.get());
    else if (h->kind() == rt::Hitable::FLIP_NORMALS) rect = static_cast<const rt::FlipNormals*>(h.get())->rect.get();
    if (!rect || rect->axis != 1) continue;
    if (b->tables.materials[(size_t)rect->material.id].type != RT_MAT_DIFFUSE_LIGHT) continue;
    if (found == 0) {
      b->tables.lights.clear();
      (void)rtb_light_xz_rect(b, rect->a0, rect->a1, rect->b0, rect->b1, rect->k);
    }
    ++found;
  }
  return found;
}

int rtb_light_add(rt_builder* b, int type, double a, double bb, double c, double d, double e) {
  if (type < RT_LIGHT_XZ_RECT || type > RT_LIGHT_SPHERE) return rt::set_error(RT_ERR_INVALID, "rtb_light_add: unknown light type");
  if (b->tables.lights.size() >= (size_t)RT_MAX_LIGHTS) return rt::set_error(RT_ERR_UNSUPPORTED, "rtb_light_add: more than RT_MAX_LIGHTS shapes");
  rt_light l;
  std::memset(&l, 0, sizeof(l));
  l.type = type;
  l.f[0] = (float)a; l.f[1] = (float)bb; l.f[2] = (float)c; l.f[3] = (float)d; l.f[4] = (float)e;
  b->tables.lights.push_back(l);
  return (int)b->tables.lights.size();
}

int rtb_lights_from_world(rt_builder* b, int world_list) {
  if (bad_hitable(b, world_list) || b->hitables[world_list]->kind() != rt::Hitable::HITABLE_LIST)
    return rt::set_error(RT_ERR_INVALID, "rtb_lights_from_world: world must be a HitableList handle");
  const rt::HitableList* world = static_cast<const rt::HitableList*>(b->hitables[world_list].get());
  b->tables.lights.clear();
  for (const rt::HitablePtr& h : world->list) {
    if (b->tables.lights.size() >= (size_t)RT_MAX_LIGHTS) break;
    if (h->kind() == rt::Hitable::SPHERE) {
      const rt::Sphere* sp = static_cast<const rt::Sphere*>(h.get());
      if (b->tables.materials[(size_t)sp->material.id].type != RT_MAT_DIFFUSE_LIGHT || !(sp->radius > 0.0)) continue;
      (void)rtb_light_add(b, RT_LIGHT_SPHERE, sp->center.x, sp->center.y, sp->center.z, sp->radius, 0.0);
      continue;
    }
    const rt::Rect* rect = nullptr;
    if (h->kind() == rt::Hitable::XY_RECT || h->kind() == rt::Hitable::XZ_RECT || h->kind() == rt::Hitable::YZ_RECT) rect = static_cast<const rt::Rect*>(h.get());
    else if (h->kind() == rt::Hitable::FLIP_NORMALS) rect = static_cast<const rt::FlipNormals*>(h.get())->rect.get();
    if (!rect || b->tables.materials[(size_t)rect->material.id].type != RT_MAT_DIFFUSE_LIGHT) continue;
    if (!(rect->a0 < rect->a1) || !(rect->b0 < rect->b1)) continue;
    const int type = rect->axis == 2 ? RT_LIGHT_XY_RECT : (rect->axis == 1 ? RT_LIGHT_XZ_RECT : RT_LIGHT_YZ_RECT);
    (void)rtb_light_add(b, type, rect->a0, rect->a1, rect->b0, rect->b1, rect->k);
  }
  return (int)b->tables.lights.size();
}

int rtb_flatten(rt_builder* b, int world_list, rt_scene* out) {
  if (!out) return rt::set_error(RT_ERR_INVALID, "rtb_flatten: null out");
  if (bad_hitable(b, world_list) || b->hitables[world_list]->kind() != rt::Hitable::HITABLE_LIST)
    return rt::set_error(RT_ERR_INVALID, "rtb_flatten: world must be a HitableList handle");
  return guarded([&]() {
    const rt::HitableList* world = static_cast<const rt::HitableList*>(b->hitables[world_list].get());
    rt::flatten_world(*world, b->tables, b->rng, &b->flat);
    *out = b->flat.view();
    return RT_OK;
  });
}

}  // extern "C"
