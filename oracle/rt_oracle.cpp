/*
 * rt_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the per-pixel path-tracing loop of
 * jesseyoungmann/raytracing, written from its Rust source with the quirks of
 * that source kept (each function cites the lines it follows).  It evaluates
 * the same flattened scene (include/rt_abi.h) the device library receives, but
 * walks it the way the reference walks its object tree: recursive color(),
 * recursive unpruned BvhNode::hit, HitableList with a shrinking t_max.
 *
 * Parity status: PARITY UNPINNED by the reference's own tests (it has none);
 * the single pin is examples/cornell_box.png, checked in tests/test_oracle.py
 * (block means at 20x20 and 10x10 pixels, wall colour ratios) and, for the device,
 * in tests/test_gpu_render.py.  See rt_oracle.h.
 *
 * Random numbers: the reference uses rand::thread_rng() (OS seeded); this
 * restatement draws from the Philox4x32-10 stream fixed in include/rt_abi.h so
 * that it can be compared with the device sample for sample.  Philox itself is
 * restated from Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy
 * as 1, 2, 3" (SC'11) and checked against that paper's known-answer vectors.
 *
 * Build with -ffp-contract=off: every fused multiply-add below is explicit.
 */
#include "rt_oracle.h"

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------------------
// Philox4x32-10
// ---------------------------------------------------------------------------
inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int round = 0; round < 10; ++round) {
    const uint64_t p0 = (uint64_t)M0 * c0;
    const uint64_t p1 = (uint64_t)M1 * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// ---------------------------------------------------------------------------
// Arithmetic policies
// ---------------------------------------------------------------------------
struct PolicyF64 { typedef double Real; static const bool fused = false; };
struct PolicyF32 { typedef float Real; static const bool fused = false; };
struct PolicyF32Fused { typedef float Real; static const bool fused = true; };

// a*b + c : two roundings in the reference's order, one rounding on the device.
template <class P> inline typename P::Real madd(typename P::Real a, typename P::Real b, typename P::Real c) {
  if (P::fused) return std::fma(a, b, c);
  return a * b + c;
}

// vec3.rs:6-11
template <class R> struct V3 {
  R x, y, z;
  V3() : x(0), y(0), z(0) {}
  V3(R x_, R y_, R z_) : x(x_), y(y_), z(z_) {}
  R operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
  void set(int i, R v) { if (i == 0) x = v; else if (i == 1) y = v; else z = v; }
};
// vec3.rs:53-113
template <class R> inline V3<R> operator+(V3<R> a, V3<R> b) { return V3<R>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class R> inline V3<R> operator-(V3<R> a, V3<R> b) { return V3<R>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <class R> inline V3<R> operator*(V3<R> a, V3<R> b) { return V3<R>(a.x * b.x, a.y * b.y, a.z * b.z); }
template <class R> inline V3<R> operator/(V3<R> a, V3<R> b) { return V3<R>(a.x / b.x, a.y / b.y, a.z / b.z); }
template <class R> inline V3<R> operator-(V3<R> a) { return V3<R>(-a.x, -a.y, -a.z); }
template <class R> inline V3<R> scalar(R f) { return V3<R>(f, f, f); }  // vec3.rs:16-18

// vec3.rs:176-178.  Fused order: z*z' + (y*y' + x*x').
template <class P> inline typename P::Real dot(V3<typename P::Real> a, V3<typename P::Real> b) {
  if (P::fused) return std::fma(a.z, b.z, std::fma(a.y, b.y, a.x * b.x));
  return a.x * b.x + a.y * b.y + a.z * b.z;
}
template <class P> inline typename P::Real squared_length(V3<typename P::Real> a) { return dot<P>(a, a); }  // vec3.rs:172-174
template <class P> inline typename P::Real length(V3<typename P::Real> a) { return std::sqrt(dot<P>(a, a)); }  // vec3.rs:168-170
// vec3.rs:188-190: multiply by the reciprocal of the length.
template <class P> inline V3<typename P::Real> unit(V3<typename P::Real> a) {
  typedef typename P::Real R;
  return a * scalar<R>(R(1) / length<P>(a));
}
// vec3.rs:180-186
template <class R> inline V3<R> cross(V3<R> a, V3<R> b) {
  return V3<R>(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x);
}

// ray.rs:3-21 (+ the reciprocal direction the fused policy uses)
template <class P> struct Ray {
  typedef typename P::Real R;
  V3<R> a, b, inv;
  Ray() {}
  Ray(V3<R> a_, V3<R> b_) : a(a_), b(b_) {
    if (P::fused) inv = V3<R>(R(1) / b.x, R(1) / b.y, R(1) / b.z);
  }
  V3<R> origin() const { return a; }
  V3<R> direction() const { return b; }
  V3<R> point_at_parameter(R t) const {
    if (P::fused) return V3<R>(std::fma(t, b.x, a.x), std::fma(t, b.y, a.y), std::fma(t, b.z, a.z));
    return a + b * scalar<R>(t);
  }
};

// hitable.rs:9-17
template <class R> struct HitRecord {
  R t, u, v;
  V3<R> p, normal;
  int material;
  int prim;
  HitRecord() : t(0), u(0), v(0), material(-1), prim(-1) {}
};

template <class R> inline R pi() { return R(3.14159265358979323846264338327950288); }

struct Counters {
  uint64_t paths, rays, hits, node_tests, rect_tests, sphere_tests, instance_ops, shade_events, light_tests;
  Counters() { std::memset(this, 0, sizeof(*this)); }
  void add(const Counters& o) {
    paths += o.paths; rays += o.rays; hits += o.hits; node_tests += o.node_tests; rect_tests += o.rect_tests;
    sphere_tests += o.sphere_tests; instance_ops += o.instance_ops; shade_events += o.shade_events;
    light_tests += o.light_tests;
  }
};

// Per-sample random stream (include/rt_abi.h, "Random numbers").
struct Sampler {
  uint32_t key[2];
  uint32_t pixel, sample;
  bool enabled;  // false: deterministic queries, media are skipped
  void draw(uint32_t slot, uint32_t sub, uint32_t out[4]) const {
    const uint32_t ctr[4] = {pixel, sample, slot, sub};
    philox4x32_10(ctr, key, out);
  }
  template <class R> static R uniform(uint32_t x) { return R(x >> 8) * R(1.0 / 16777216.0); }
};

// ---------------------------------------------------------------------------
// The scene, evaluated the reference's way
// ---------------------------------------------------------------------------
template <class P> struct World {
  typedef typename P::Real R;
  typedef V3<R> Vec;
  const rt_scene* s;
  mutable Counters cnt;

  explicit World(const rt_scene* scene) : s(scene) {}

  static R real_max() { return std::numeric_limits<R>::max(); }

  // ---- bvh.rs:46-61 Aabb::hit ----
  bool aabb_hit(const rt_node& n, const Ray<P>& r, R t_min, R t_max) const {
    cnt.node_tests++;
    for (int a = 0; a < 3; ++a) {
      const R inv_d = R(1) / r.direction()[a];
      R t0 = (R(n.bmin[a]) - r.origin()[a]) * inv_d;
      R t1 = (R(n.bmax[a]) - r.origin()[a]) * inv_d;
      if (inv_d < R(0)) { R tmp = t0; t0 = t1; t1 = tmp; }
      t_min = t0 > t_min ? t0 : t_min;
      t_max = t1 < t_max ? t1 : t_max;
      if (sizeof(R) == 8 && t_max <= t_min) return false;
    }
    // fp32 only, not in the reference: a rect's box is 0.0002 thick, and from an origin a few thousand units away
    // (bound - origin) rounds to the same float for both faces, so t0 == t1 and the reference's `<=` rejects a box the
    // ray goes through (in f64 that needs distances of 1e12).  The exit distance is widened by two units in the last
    // place, which never rejects a box the exact test accepts.
    if (sizeof(R) == 4) return !(t_max * R(1.00000024f) <= t_min);
    return true;
  }

  // ---- hitable.rs:163-183, 216-236, 290-310: the three rects ----
  // axis c is the fixed one, (ca, cb) the two in-plane axes in constructor order.
  bool rect_hit(int idx, const Ray<P>& r, R t0, R t1, HitRecord<R>& rec) const {
    cnt.rect_tests++;
    const rt_prim& q = s->prims[idx];
    const int type = q.type & RT_PRIM_TYPE_MASK;
    int c, ca, cb;
    if (type == RT_PRIM_XY_RECT) { c = 2; ca = 0; cb = 1; }
    else if (type == RT_PRIM_XZ_RECT) { c = 1; ca = 0; cb = 2; }
    else { c = 0; ca = 1; cb = 2; }
    const R a0 = q.f[0], a1 = q.f[1], b0 = q.f[2], b1 = q.f[3], k = q.f[4];
    R t;
    if (P::fused) t = (k - r.origin()[c]) * r.inv[c];
    else t = (k - r.origin()[c]) / r.direction()[c];
    if (t < t0 || t > t1) return false;
    const R a = madd<P>(t, r.direction()[ca], r.origin()[ca]);
    const R b = madd<P>(t, r.direction()[cb], r.origin()[cb]);
    if (a < a0 || a > a1 || b < b0 || b > b1) return false;
    rec.u = (a - a0) / (a1 - a0);
    rec.v = (b - b0) / (b1 - b0);
    rec.t = t;
    rec.material = q.material;
    rec.prim = idx;
    rec.p = r.point_at_parameter(t);
    Vec nrm(0, 0, 0);
    nrm.set(c, R(1));
    // hitable.rs:341-353 FlipNormals
    if (q.type & RT_PRIM_FLIP) nrm = -nrm;
    rec.normal = nrm;
    return true;
  }

  // ---- hitable.rs:131-138 ----
  static void get_sphere_uv(Vec p, R& u, R& v) {
    const R phi = std::atan2(p.z, p.x);
    const R theta = std::asin(p.y);
    const R PI = pi<R>();
    u = R(1) - (phi + PI) / (R(2) * PI);
    v = (theta + PI / R(2)) / PI;
  }

  // The quadratic of hitable.rs:50-61: discriminant and both roots (the roots are meaningless when it is not positive).
  static void sphere_roots(Vec center, R radius, const Ray<P>& r, R& discriminant, R& near_t, R& far_t) {
    const Vec oc = r.origin() - center;
    const R a = dot<P>(r.direction(), r.direction());
    const R b = dot<P>(oc, r.direction());
    if (sizeof(R) == 4) {
      // fp32 only, not in the reference: b*b - a*c cancels badly when the origin is far from a small sphere (at 2000
      // units from a sphere of radius 2.4 its absolute error is a tenth of the discriminant's whole range, so grazing
      // hits come and go and t is off by up to 3e-4 relative).  The same quantity from the component of oc that is
      // perpendicular to the ray, l = oc - (b/a) d:  b*b - a*c = a * (r*r - l.l), whose terms are of the sphere's own
      // size.  Roots: t = -b/a -+ sqrt((r*r - l.l) / a).  The sign test is the reference's (a > 0).
      const R inv_a = R(1) / a;
      const R k = b * inv_a;
      const Vec dir = r.direction();
      R h;
      if (P::fused) {
        const Vec l(std::fma(-k, dir.x, oc.x), std::fma(-k, dir.y, oc.y), std::fma(-k, dir.z, oc.z));
        h = std::fma(radius, radius, -dot<P>(l, l));
      } else {
        const Vec l = oc - dir * scalar<R>(k);
        h = radius * radius - dot<P>(l, l);
      }
      discriminant = h;
      const R sq = std::sqrt(h * inv_a);
      near_t = -k - sq;
      far_t = -k + sq;
    } else {
      const R c = dot<P>(oc, oc) - radius * radius;
      discriminant = b * b - a * c;
      const R sq = std::sqrt(discriminant);
      near_t = (-b - sq) / a;
      far_t = (-b + sq) / a;
    }
  }

  // ---- hitable.rs:49-84 Sphere::hit ----
  bool sphere_hit(int idx, const Ray<P>& r, R t_min, R t_max, HitRecord<R>& rec) const {
    cnt.sphere_tests++;
    const rt_prim& q = s->prims[idx];
    const Vec center(q.f[0], q.f[1], q.f[2]);
    const R radius = q.f[3];
    R discriminant, near_t, far_t;
    sphere_roots(center, radius, r, discriminant, near_t, far_t);
    if (discriminant > R(0)) {
      for (int root = 0; root < 2; ++root) {
        const R temp = root == 0 ? near_t : far_t;
        if (temp < t_max && temp > t_min) {
          rec.t = temp;
          rec.p = r.point_at_parameter(rec.t);
          if (sizeof(R) == 4) {
            // fp32 only, not in the reference: ray(t) lies off the sphere by the rounding of t * d (1e-4 at a
            // few thousand units), enough for the next ray to hit the same sphere again from inside; in f64
            // that never shows.  The point is put back at distance |radius| from the centre.
            const Vec dd = rec.p - center;
            const R len = std::sqrt(dot<P>(dd, dd));
            if (P::fused) {
              const R scale = std::fabs(radius) * (R(1) / len);
              rec.p = Vec(std::fma(dd.x, scale, center.x), std::fma(dd.y, scale, center.y), std::fma(dd.z, scale, center.z));
            } else {
              rec.p = center + dd * scalar<R>(std::fabs(radius) / len);
            }
          }
          get_sphere_uv(rec.p, rec.u, rec.v);
          if (P::fused) rec.normal = (rec.p - center) * scalar<R>(R(1) / radius);
          else rec.normal = (rec.p - center) / scalar<R>(radius);
          rec.material = q.material;
          rec.prim = idx;
          return true;
        }
      }
    }
    return false;
  }

  bool prim_hit(int idx, const Ray<P>& r, R t_min, R t_max, HitRecord<R>& rec) const {
    if ((s->prims[idx].type & RT_PRIM_TYPE_MASK) == RT_PRIM_SPHERE) return sphere_hit(idx, r, t_min, t_max, rec);
    return rect_hit(idx, r, t_min, t_max, rec);
  }

  // ---- hitable.rs:105-117 HitableList::hit ----
  bool list_hit(int first, int count, const Ray<P>& r, R t_min, R t_max, HitRecord<R>& out) const {
    bool hit_anything = false;
    R closest_so_far = t_max;
    for (int i = first; i < first + count; ++i) {
      HitRecord<R> rec;
      if (prim_hit(i, r, t_min, closest_so_far, rec)) {
        closest_so_far = rec.t;
        out = rec;
        hit_anything = true;
      }
    }
    return hit_anything;
  }

  // Inner hitable of an object in the object's frame.
  bool inner_hit(const rt_object& ob, const Ray<P>& r, R t_min, R t_max, const Sampler& smp, int depth,
                 HitRecord<R>& rec) const {
    if (ob.kind == RT_OBJ_BVH) return node_hit(ob.first, r, t_min, t_max, smp, depth, rec);
    return list_hit(ob.first, ob.count, r, t_min, t_max, rec);
  }

  // ---- hitable.rs:438-446 Translate::hit around hitable.rs:508-532 RotateY::hit ----
  bool xform_hit(const rt_object& ob, const Ray<P>& r, R t_min, R t_max, const Sampler& smp, int depth,
                 HitRecord<R>& rec) const {
    Ray<P> local = r;
    const Vec offset(ob.offset[0], ob.offset[1], ob.offset[2]);
    const R cos_theta = ob.cos_theta, sin_theta = ob.sin_theta;
    if (ob.xform) cnt.instance_ops++;
    if (ob.xform & RT_XFORM_TRANSLATE) local = Ray<P>(local.origin() - offset, local.direction());
    if (ob.xform & RT_XFORM_ROTATE_Y) {
      Vec origin = local.origin(), direction = local.direction();
      if (P::fused) {
        origin.x = std::fma(cos_theta, local.origin().x, -(sin_theta * local.origin().z));
        origin.z = std::fma(sin_theta, local.origin().x, cos_theta * local.origin().z);
        direction.x = std::fma(cos_theta, local.direction().x, -(sin_theta * local.direction().z));
        direction.z = std::fma(sin_theta, local.direction().x, cos_theta * local.direction().z);
      } else {
        origin.x = cos_theta * local.origin().x - sin_theta * local.origin().z;
        origin.z = sin_theta * local.origin().x + cos_theta * local.origin().z;
        direction.x = cos_theta * local.direction().x - sin_theta * local.direction().z;
        direction.z = sin_theta * local.direction().x + cos_theta * local.direction().z;
      }
      local = Ray<P>(origin, direction);
    }
    if (!inner_hit(ob, local, t_min, t_max, smp, depth, rec)) return false;
    if (ob.xform & RT_XFORM_ROTATE_Y) {
      Vec p = rec.p, normal = rec.normal;
      if (P::fused) {
        p.x = std::fma(cos_theta, rec.p.x, sin_theta * rec.p.z);
        p.z = std::fma(-sin_theta, rec.p.x, cos_theta * rec.p.z);
        normal.x = std::fma(cos_theta, rec.normal.x, sin_theta * rec.normal.z);
        normal.z = std::fma(-sin_theta, rec.normal.x, cos_theta * rec.normal.z);
      } else {
        p.x = cos_theta * rec.p.x + sin_theta * rec.p.z;
        p.z = -sin_theta * rec.p.x + cos_theta * rec.p.z;
        normal.x = cos_theta * rec.normal.x + sin_theta * rec.normal.z;
        normal.z = -sin_theta * rec.normal.x + cos_theta * rec.normal.z;
      }
      rec.p = p;
      rec.normal = normal;
    }
    if (ob.xform & RT_XFORM_TRANSLATE) rec.p = rec.p + offset;
    return true;
  }

  // ---- hitable.rs:560-598 ConstantMedium::hit (commented out in the reference) ----
  bool medium_hit(int obj_index, const rt_object& ob, const Ray<P>& r, R t_min, R t_max, const Sampler& smp,
                  int depth, HitRecord<R>& rec) const {
    if (!smp.enabled) return false;
    rt_object boundary = ob;  // (the boundary is any hitable: primitives, or a sub-tree)
    boundary.kind = ob.kind == RT_OBJ_MEDIUM_BVH ? RT_OBJ_BVH : RT_OBJ_LIST;
    HitRecord<R> rec1, rec2;
    if (!xform_hit(boundary, r, -real_max(), real_max(), smp, depth, rec1)) return false;
    R t_next = rec1.t + R(0.0001);
    // fp32 only, not in the reference: beyond t = 2048 the 0.0001 is less than half a unit in the last place, the
    // second search would start at rec1.t itself and a rect (whose test accepts t == t_min) would be found again
    if (sizeof(R) == 4 && !(t_next > rec1.t)) t_next = std::nextafter(rec1.t, std::numeric_limits<R>::infinity());
    if (!xform_hit(boundary, r, t_next, real_max(), smp, depth, rec2)) return false;
    if (rec1.t < t_min) rec1.t = t_min;
    if (rec2.t > t_max) rec2.t = t_max;
    if (rec1.t >= rec2.t) return false;
    if (rec1.t < R(0)) rec1.t = R(0);
    const R dlen = length<P>(r.direction());
    const R distance_inside_boundary = (rec2.t - rec1.t) * dlen;
    uint32_t w[4];
    smp.draw((uint32_t)depth, 4096u + (uint32_t)ob.medium_index, w);
    const R xi = Sampler::uniform<R>(w[0]);
    const R hit_distance = -(R(1) / R(ob.density)) * std::log(xi);
    if (hit_distance < distance_inside_boundary) {
      const R t = rec1.t + hit_distance / dlen;
      rec.t = t;
      rec.u = 0;
      rec.v = 0;
      rec.p = r.point_at_parameter(t);
      rec.normal = Vec(1, 0, 0);
      rec.material = ob.material;
      rec.prim = s->n_prims + obj_index;
      return true;
    }
    return false;
  }

  bool leaf_hit(const rt_node& n, const Ray<P>& r, R t_min, R t_max, const Sampler& smp, int depth,
                HitRecord<R>& rec) const {
    const int ref = ~n.left;
    if (n.right > 0) return list_hit(ref, n.right, r, t_min, t_max, rec);
    const rt_object& ob = s->objects[ref];
    if (ob.kind == RT_OBJ_MEDIUM || ob.kind == RT_OBJ_MEDIUM_BVH) return medium_hit(ref, ob, r, t_min, t_max, smp, depth, rec);
    return xform_hit(ob, r, t_min, t_max, smp, depth, rec);
  }

  // ---- bvh.rs:197-228 BvhNode::hit: both children, same t_max, right wins ties ----
  bool node_hit(int idx, const Ray<P>& r, R t_min, R t_max, const Sampler& smp, int depth,
                HitRecord<R>& rec) const {
    const rt_node& n = s->nodes[idx];
    if (!aabb_hit(n, r, t_min, t_max)) return false;
    if (n.left < 0) return leaf_hit(n, r, t_min, t_max, smp, depth, rec);
    HitRecord<R> hl, hr;
    const bool hit_left = node_hit(n.left, r, t_min, t_max, smp, depth, hl);
    const bool hit_right = node_hit(n.right, r, t_min, t_max, smp, depth, hr);
    if (hit_left && hit_right) {
      rec = (hl.t < hr.t) ? hl : hr;
      return true;
    }
    if (hit_left) { rec = hl; return true; }
    if (hit_right) { rec = hr; return true; }
    return false;
  }

  // Every leaf of the world tree in depth-first order as one HitableList:
  // the `color(&r, &*world, depth)` alternative at main.rs:109.
  void collect_leaves(int idx, std::vector<int>& out) const {
    const rt_node& n = s->nodes[idx];
    if (n.left < 0) { out.push_back(idx); return; }
    collect_leaves(n.left, out);
    collect_leaves(n.right, out);
  }
  bool world_list_hit(const Ray<P>& r, R t_min, R t_max, const Sampler& smp, int depth, HitRecord<R>& out) const {
    std::vector<int> leaves;
    collect_leaves(0, leaves);
    bool any = false;
    R closest = t_max;
    for (size_t i = 0; i < leaves.size(); ++i) {
      HitRecord<R> rec;
      if (leaf_hit(s->nodes[leaves[i]], r, t_min, closest, smp, depth, rec)) {
        closest = rec.t;
        out = rec;
        any = true;
      }
    }
    return any;
  }

  bool world_hit(const Ray<P>& r, R t_min, R t_max, const Sampler& smp, int depth, HitRecord<R>& rec) const {
    cnt.rays++;
    if (s->n_nodes <= 0) return false;
    const bool hit = node_hit(0, r, t_min, t_max, smp, depth, rec);
    if (hit) cnt.hits++;
    return hit;
  }

  // ---- perlin.rs:102-124 ----
  static R perlin_interpolate(const Vec c[2][2][2], R u, R v, R w) {
    const R uu = u * u * (R(3) - R(2) * u);
    const R vv = v * v * (R(3) - R(2) * v);
    const R ww = w * w * (R(3) - R(2) * w);
    R accum = 0;
    for (int ii = 0; ii < 2; ++ii) {
      const R i = R(ii);
      for (int jj = 0; jj < 2; ++jj) {
        const R j = R(jj);
        for (int kk = 0; kk < 2; ++kk) {
          const R k = R(kk);
          const Vec weight_v(u - i, v - j, w - k);
          accum += (i * uu + (R(1) - i) * (R(1) - uu)) * (j * vv + (R(1) - j) * (R(1) - vv)) *
                   (k * ww + (R(1) - k) * (R(1) - ww)) * dot<P>(c[ii][jj][kk], weight_v);
        }
      }
    }
    return accum;
  }

  // `f64 as usize` of current Rust: saturating, NaN -> 0 (perlin.rs:34-36).
  static uint64_t as_usize(R f) {
    if (!(f > R(0))) return 0;
    if (f >= R(18446744073709551615.0)) return ~(uint64_t)0;
    return (uint64_t)f;
  }

  // ---- perlin.rs:29-50 ----
  static R perlin_noise(const rt_perlin& pn, Vec p) {
    const R u = p.x - std::floor(p.x);
    const R v = p.y - std::floor(p.y);
    const R w = p.z - std::floor(p.z);
    const uint64_t i = as_usize(std::floor(p.x));
    const uint64_t j = as_usize(std::floor(p.y));
    const uint64_t k = as_usize(std::floor(p.z));
    Vec c[2][2][2];
    for (int di = 0; di < 2; ++di)
      for (int dj = 0; dj < 2; ++dj)
        for (int dk = 0; dk < 2; ++dk) {
          const int idx = pn.perm_x[(i + di) & 255] ^ pn.perm_y[(j + dj) & 255] ^ pn.perm_z[(k + dk) & 255];
          c[di][dj][dk] = Vec(pn.ranvec[idx][0], pn.ranvec[idx][1], pn.ranvec[idx][2]);
        }
    return perlin_interpolate(c, u, v, w);
  }

  // ---- perlin.rs:52-63 ----
  static R perlin_turb(const rt_perlin& pn, Vec p, int depth) {
    R accum = 0;
    Vec temp_p = p;
    R weight = 1;
    for (int i = 0; i < depth; ++i) {
      accum += weight * perlin_noise(pn, temp_p);
      weight *= R(0.5);
      temp_p = temp_p * scalar<R>(R(2));
    }
    return std::fabs(accum);
  }

  // ---- texture.rs:22-35 ----
  Vec texture_value(int tex, R u, R v, Vec p) const {
    const rt_texture& t = s->textures[tex];
    if (t.type == RT_TEX_CHECKER) {
      const R sines = std::sin(R(10) * p.x) * std::sin(R(10) * p.y) * std::sin(R(10) * p.z);
      return texture_value(sines < R(0) ? t.odd : t.even, u, v, p);
    }
    if (t.type == RT_TEX_NOISE) {
      const R turb = perlin_turb(s->perlins[t.perlin], p, 7);
      return scalar<R>(R(0.5)) * scalar<R>(R(1) + std::sin(R(t.scale) * p.z + R(10) * turb));
    }
    return Vec(t.color[0], t.color[1], t.color[2]);
  }

  // ---- main.rs:211-221 random_in_unit_sphere ----
  Vec random_in_unit_sphere(const Sampler& smp, int depth) const {
    for (uint32_t attempt = 0;; ++attempt) {
      uint32_t w[4];
      smp.draw((uint32_t)depth, 256u + attempt, w);
      const Vec p = scalar<R>(R(2)) * Vec(Sampler::uniform<R>(w[0]), Sampler::uniform<R>(w[1]), Sampler::uniform<R>(w[2])) - scalar<R>(R(1));
      if (squared_length<P>(p) < R(1)) return p;
      if (attempt > 4096) std::abort();
    }
  }

  // ---- main.rs:239-249 random_cosine_direction (x and y carry the source's factor 2) ----
  static Vec random_cosine_direction(R r1, R r2) {
    const R z = std::sqrt(R(1) - r2);
    const R phi = R(2) * pi<R>() * r1;
    const R r2_sqrt = std::sqrt(r2);
    const R x = std::cos(phi) * R(2) * r2_sqrt;
    const R y = std::sin(phi) * R(2) * r2_sqrt;
    return Vec(x, y, z);
  }

  // ---- onb.rs:9-26 ----
  struct ONB {
    Vec u, v, w;
    static ONB build_from_w(Vec n) {
      ONB o;
      o.w = unit<P>(n);
      const Vec a = std::fabs(o.w.x) > R(0.9) ? Vec(0, 1, 0) : Vec(1, 0, 0);
      o.v = unit<P>(cross(o.w, a));
      o.u = cross(o.w, o.v);
      return o;
    }
    Vec local(Vec a) const { return scalar<R>(a.x) * u + scalar<R>(a.y) * v + scalar<R>(a.z) * w; }
  };

  // ---- Hitable::pdf_value (hitable.rs:23-25, 245-254) of a light shape ----
  // Only XZRect overrides it in the reference (hitable.rs:245-254).  The XY and YZ rects get the same formula with
  // their own axes; the sphere gets the book's ("Ray Tracing: The Rest of Your Life", Sphere::pdf_value): uniform over
  // the cone the sphere subtends, zero for directions that miss it.  Those are this build's additions: the reference
  // has no code for them.
  R light_pdf_value(const rt_light& lt, Vec o, Vec v) const {
    cnt.light_tests++;
    if (lt.type == RT_LIGHT_SPHERE) {
      const Vec center(lt.f[0], lt.f[1], lt.f[2]);
      const R radius = lt.f[3];
      R discriminant, near_t, far_t;
      sphere_roots(center, radius, Ray<P>(o, v), discriminant, near_t, far_t);
      if (!(discriminant > R(0))) return 0;
      const bool near_ok = near_t < real_max() && near_t > R(0.001), far_ok = far_t < real_max() && far_t > R(0.001);
      if (!near_ok && !far_ok) return 0;
      const R cos_theta_max = std::sqrt(R(1) - radius * radius / squared_length<P>(center - o));
      const R solid_angle = R(2) * pi<R>() * (R(1) - cos_theta_max);
      return R(1) / solid_angle;
    }
    // the fixed axis C and the in-plane axes A, B in constructor order (hitable.rs:151, 204, 278)
    const int C = lt.type == RT_LIGHT_XZ_RECT ? 1 : (lt.type == RT_LIGHT_XY_RECT ? 2 : 0);
    const int A = lt.type == RT_LIGHT_YZ_RECT ? 1 : 0;
    const int B = lt.type == RT_LIGHT_XY_RECT ? 1 : 2;
    const R a0 = lt.f[0], a1 = lt.f[1], b0 = lt.f[2], b1 = lt.f[3], k = lt.f[4];
    // XZRect::hit(&Ray::new(o, v), 0.001, f64::MAX), hitable.rs:217-236
    const R t = (k - o[C]) / v[C];
    if (t < R(0.001) || t > real_max()) return 0;
    const R a = o[A] + t * v[A];
    const R b = o[B] + t * v[B];
    if (a < a0 || a > a1 || b < b0 || b > b1) return 0;
    const R area = (a1 - a0) * (b1 - b0);
    const R distance_squared = t * t * squared_length<P>(v);
    const R cosine = std::fabs(v[C] / length<P>(v));  // dot(v, normal) with the rect's own axis normal
    return distance_squared / (cosine * area);
  }

  // ---- Hitable::random (hitable.rs:27-29, 256-264) of a light shape ----
  static Vec light_random(const rt_light& lt, Vec o, R xi1, R xi2) {
    if (lt.type == RT_LIGHT_SPHERE) {
      // the book's Sphere::random: a direction inside the cone towards the sphere
      const Vec center(lt.f[0], lt.f[1], lt.f[2]);
      const R radius = lt.f[3];
      const Vec direction = center - o;
      const R distance_squared = squared_length<P>(direction);
      const ONB uvw = ONB::build_from_w(direction);
      const R z = R(1) + xi2 * (std::sqrt(R(1) - radius * radius / distance_squared) - R(1));
      const R phi = R(2) * pi<R>() * xi1;
      const R sxy = std::sqrt(R(1) - z * z);
      return uvw.local(Vec(std::cos(phi) * sxy, std::sin(phi) * sxy, z));
    }
    const R a0 = lt.f[0], a1 = lt.f[1], b0 = lt.f[2], b1 = lt.f[3], k = lt.f[4];
    const R a = a0 + xi1 * (a1 - a0), b = b0 + xi2 * (b1 - b0);
    const Vec random_point = lt.type == RT_LIGHT_XZ_RECT ? Vec(a, k, b) : (lt.type == RT_LIGHT_XY_RECT ? Vec(a, b, k) : Vec(k, a, b));
    return random_point - o;
  }

  // ---- material.rs:162-180 (commented out in the reference) ----
  static Vec reflect(Vec v, Vec n) { return v - scalar<R>(R(2)) * scalar<R>(dot<P>(v, n)) * n; }
  static bool refract(Vec v, Vec n, R ni_over_nt, Vec& out) {
    const Vec uv = unit<P>(v);
    const R dt = dot<P>(uv, n);
    const R discriminant = R(1) - ni_over_nt * ni_over_nt * (R(1) - dt * dt);
    if (discriminant > R(0)) {
      out = scalar<R>(ni_over_nt) * (uv - n * scalar<R>(dt)) - n * scalar<R>(std::sqrt(discriminant));
      return true;
    }
    return false;
  }
  static R schlick(R cosine, R ref_idx) {
    R r0 = (R(1) - ref_idx) / (R(1) + ref_idx);
    r0 = r0 * r0;
    return r0 + (R(1) - r0) * std::pow(R(1) - cosine, R(5));
  }

  Vec sky(const Ray<P>& r, int mode) const {
    if (mode == RT_SKY_GRADIENT) {  // main.rs:203-205
      const Vec unit_direction = unit<P>(r.direction());
      const R t = R(0.5) * (unit_direction.y + R(1));
      return scalar<R>(R(1) - t) * scalar<R>(R(1)) + scalar<R>(t) * Vec(R(0.5), R(0.7), R(1.0));
    }
    return scalar<R>(R(0));  // main.rs:32,208
  }

  // ---- main.rs:165-209 color ----
  Vec color(const Ray<P>& r, int depth, const Sampler& smp, const rt_render_params& prm) const {
    HitRecord<R> rec;
    if (!world_hit(r, R(prm.t_min), std::numeric_limits<R>::infinity(), smp, depth, rec)) return sky(r, prm.sky);
    const rt_material& m = s->materials[rec.material];
    // material.rs:43-48, 197-203
    Vec emitted = scalar<R>(R(0));
    if (m.type == RT_MAT_DIFFUSE_LIGHT && dot<P>(rec.normal, r.direction()) < R(0))
      emitted = texture_value(m.texture, rec.u, rec.v, rec.p);
    if (depth >= prm.max_depth) return emitted;

    if (m.type == RT_MAT_LAMBERTIAN) {
      cnt.shade_events++;
      // material.rs:67-74: only the albedo of scatter() is used by color().
      const Vec albedo = texture_value(m.texture, rec.u, rec.v, rec.p);
      uint32_t w[4];
      smp.draw((uint32_t)depth, 0u, w);
      const R xi1 = Sampler::uniform<R>(w[2]), xi2 = Sampler::uniform<R>(w[3]);
      const ONB uvw = ONB::build_from_w(rec.normal);  // pdf.rs:13-17
      Vec dir;
      R pdf_val;
      if (s->n_lights > 0) {
        // HitablePDF over the light shapes.  The reference samples one shape (main.rs:177-178).  With several, the
        // shapes form a list whose random() picks one uniformly and whose pdf_value() is the mean of theirs (the
        // book's HitableList; the reference's list keeps the defaults 0.0 and (1,0,0), hitable.rs:23-29): one more
        // draw, counter (pixel, sample, depth, 8192), word 0, picks the shape.
        int pick = 0;
        if (s->n_lights > 1) {
          uint32_t w2[4];
          smp.draw((uint32_t)depth, 8192u, w2);
          pick = (int)(((uint64_t)w2[0] * (uint64_t)s->n_lights) >> 32);
        }
        // pdf.rs:69-75: gen::<bool>() is the top bit of a fresh 32-bit word.
        if (w[1] >> 31) dir = light_random(s->lights[pick], rec.p, xi1, xi2);
        else dir = uvw.local(random_cosine_direction(xi1, xi2));
        // pdf.rs:65-67, 19-26
        const R cosine = dot<P>(unit<P>(dir), uvw.w);
        const R cos_pdf = cosine > R(0) ? cosine / pi<R>() : R(0);
        R light_pdf = 0;
        for (int k = 0; k < s->n_lights; ++k) light_pdf += light_pdf_value(s->lights[k], rec.p, dir);
        if (s->n_lights > 1) light_pdf /= R(s->n_lights);
        pdf_val = R(0.5) * light_pdf + R(0.5) * cos_pdf;
      } else {
        dir = uvw.local(random_cosine_direction(xi1, xi2));
        const R cosine = dot<P>(unit<P>(dir), uvw.w);
        pdf_val = cosine > R(0) ? cosine / pi<R>() : R(0);
      }
      const Ray<P> scattered(rec.p, dir);  // main.rs:186
      // material.rs:76-82
      R cosine = dot<P>(rec.normal, unit<P>(scattered.direction()));
      if (cosine < R(0)) cosine = R(0);
      const R scattering_pdf = cosine / pi<R>();
      // main.rs:189-197
      return emitted + albedo * scalar<R>(scattering_pdf) * color(scattered, depth + 1, smp, prm) / scalar<R>(pdf_val);
    }
    if (m.type == RT_MAT_METAL) {  // material.rs:100-112
      const Vec reflected = reflect(unit<P>(r.direction()), rec.normal);
      const Ray<P> scattered(rec.p, reflected + scalar<R>(R(m.fuzz)) * random_in_unit_sphere(smp, depth));
      if (dot<P>(scattered.direction(), rec.normal) > R(0)) {
        const Vec attenuation(m.albedo[0], m.albedo[1], m.albedo[2]);
        return emitted + attenuation * color(scattered, depth + 1, smp, prm);
      }
      return emitted;
    }
    if (m.type == RT_MAT_DIELECTRIC) {  // material.rs:125-159
      const Vec reflected = reflect(r.direction(), rec.normal);
      const R ref_idx = m.ref_idx;
      Vec outward_normal;
      R ni_over_nt, cosine;
      if (dot<P>(r.direction(), rec.normal) > R(0)) {
        outward_normal = -rec.normal;
        ni_over_nt = ref_idx;
        cosine = ref_idx * dot<P>(r.direction(), rec.normal) / length<P>(r.direction());
      } else {
        outward_normal = rec.normal;
        ni_over_nt = R(1) / ref_idx;
        cosine = -dot<P>(r.direction(), rec.normal) / length<P>(r.direction());
      }
      Vec refracted;
      R reflect_prob;
      const bool can_refract = refract(r.direction(), outward_normal, ni_over_nt, refracted);
      reflect_prob = can_refract ? schlick(cosine, ref_idx) : R(1);
      uint32_t w[4];
      smp.draw((uint32_t)depth, 0u, w);
      const Ray<P> scattered = (Sampler::uniform<R>(w[1] << 1) < reflect_prob) ? Ray<P>(rec.p, reflected) : Ray<P>(rec.p, refracted);
      return emitted + color(scattered, depth + 1, smp, prm);  // attenuation (1,1,1)
    }
    if (m.type == RT_MAT_ISOTROPIC) {  // material.rs:217-221
      const Ray<P> scattered(rec.p, random_in_unit_sphere(smp, depth));
      const Vec attenuation = texture_value(m.texture, rec.u, rec.v, rec.p);
      return emitted + attenuation * color(scattered, depth + 1, smp, prm);
    }
    return emitted;  // DiffuseLight::scatter -> None, material.rs:193-195
  }

  // ---- camera.rs:53-75 ----
  Ray<P> get_ray(const rt_camera& cam, R s_, R t_, const Sampler& smp) const {
    Vec p;
    for (uint32_t attempt = 0;; ++attempt) {
      uint32_t w[4];
      smp.draw(0u, 1u + attempt, w);
      p = scalar<R>(R(2)) * Vec(Sampler::uniform<R>(w[0]), Sampler::uniform<R>(w[1]), R(0)) - Vec(R(1), R(1), R(0));
      if (dot<P>(p, p) < R(1)) break;
      if (attempt > 4096) std::abort();
    }
    const Vec origin(R(cam.origin[0]), R(cam.origin[1]), R(cam.origin[2]));
    const Vec llc(R(cam.lower_left_corner[0]), R(cam.lower_left_corner[1]), R(cam.lower_left_corner[2]));
    const Vec horizontal(R(cam.horizontal[0]), R(cam.horizontal[1]), R(cam.horizontal[2]));
    const Vec vertical(R(cam.vertical[0]), R(cam.vertical[1]), R(cam.vertical[2]));
    const Vec cu(R(cam.u[0]), R(cam.u[1]), R(cam.u[2]));
    const Vec cv(R(cam.v[0]), R(cam.v[1]), R(cam.v[2]));
    const Vec rd = scalar<R>(R(cam.lens_radius)) * p;
    const Vec offset = cu * scalar<R>(rd.x) + cv * scalar<R>(rd.y);
    return Ray<P>(origin + offset, llc + scalar<R>(s_) * horizontal + scalar<R>(t_) * vertical - origin - offset);
  }

  // One sample of main.rs:101-109 (before de_nan).
  Vec sample_pixel(const rt_camera& cam, const rt_render_params& prm, int i, int j, int sample, uint64_t seed) const {
    Sampler smp;
    smp.key[0] = (uint32_t)seed;
    smp.key[1] = (uint32_t)(seed >> 32);
    smp.pixel = (uint32_t)(j * prm.nx + i);
    smp.sample = (uint32_t)sample;
    smp.enabled = true;
    uint32_t w[4];
    smp.draw(0u, 0u, w);
    const R u = (R(i) + R(w[0] >> 16) * R(1.0 / 65536.0)) / R(prm.nx);
    const R v = (R(j) + R(w[0] & 0xffffu) * R(1.0 / 65536.0)) / R(prm.ny);
    const Ray<P> r = get_ray(cam, u, v, smp);
    cnt.paths++;
    return color(r, 0, smp, prm);
  }
};

// main.rs:252-264
template <class R> inline V3<R> de_nan(V3<R> c) {
  if (std::isnan(c.x)) c.x = 0;
  if (std::isnan(c.y)) c.y = 0;
  if (std::isnan(c.z)) c.z = 0;
  return c;
}

// rt_shard_width of include/rt_abi.h, restated (the oracle does not link the library it checks).
static int shard_grid_width(int nx, int shard_count) {
  int w = (nx + 7) / 8;
  if (shard_count <= 1) return w;
  for (;; ++w) {
    int a = w, b = shard_count;
    while (b) { const int r = a % b; a = b; b = r; }
    if (a == 1) return w;
  }
}

template <class P>
void render_rows(const rt_scene* scene, const rt_camera* cam, const rt_render_params* prm, int j_begin, int j_end,
                 int sample_begin, int sample_end, double* sum, double* sumsq, Counters* counters) {
  typedef typename P::Real R;
  World<P> world(scene);
  const int nx = prm->nx, ny = prm->ny;
  const int shard_width = shard_grid_width(nx, prm->shard_count);
  // main.rs:96-116
  for (int j = j_end - 1; j >= j_begin; --j) {
    for (int i = 0; i < nx; ++i) {
      if (prm->shard_count > 1) {
        const int tile = ((ny - 1 - j) / 8) * shard_width + i / 8;  // numbered as include/rt_abi.h describes
        if (tile % prm->shard_count != prm->shard_index) continue;
      }
      double acc[3] = {0, 0, 0}, acc2[3] = {0, 0, 0};
      for (int sidx = sample_begin; sidx < sample_end; ++sidx) {
        const V3<R> c = de_nan(world.sample_pixel(*cam, *prm, i, j, sidx, prm->seed));
        acc[0] += (double)c.x; acc[1] += (double)c.y; acc[2] += (double)c.z;
        acc2[0] += (double)c.x * (double)c.x; acc2[1] += (double)c.y * (double)c.y; acc2[2] += (double)c.z * (double)c.z;
      }
      const size_t at = ((size_t)(ny - 1 - j) * nx + i) * 3;
      for (int c = 0; c < 3; ++c) {
        sum[at + c] += acc[c];
        if (sumsq) sumsq[at + c] += acc2[c];
      }
    }
  }
  *counters = world.cnt;
}

template <class P>
int render_impl(const rt_scene* scene, const rt_camera* cam, const rt_render_params* prm, int n_threads,
                int split_samples, double* out_sum, double* out_sumsq, rt_stats* stats) {
  const size_t n = (size_t)prm->nx * prm->ny * 3;
  std::memset(out_sum, 0, n * sizeof(double));
  if (out_sumsq) std::memset(out_sumsq, 0, n * sizeof(double));
  if (n_threads < 1) n_threads = 1;
  std::vector<Counters> counters(n_threads);
  std::vector<std::thread> threads;
  const int s0 = prm->sample_begin, s1 = prm->sample_begin + prm->spp;
  if (split_samples) {
    // main.rs:70-133: each thread renders every pixel for its share of the samples.
    std::vector<std::vector<double> > part(n_threads), part2(n_threads);
    for (int t = 0; t < n_threads; ++t) {
      part[t].assign(n, 0.0);
      if (out_sumsq) part2[t].assign(n, 0.0);
      const int a = s0 + (int)((int64_t)prm->spp * t / n_threads);
      const int b = s0 + (int)((int64_t)prm->spp * (t + 1) / n_threads);
      threads.push_back(std::thread(render_rows<P>, scene, cam, prm, 0, prm->ny, a, b, part[t].data(),
                                    out_sumsq ? part2[t].data() : (double*)0, &counters[t]));
    }
    for (size_t t = 0; t < threads.size(); ++t) threads[t].join();
    for (int t = 0; t < n_threads; ++t)
      for (size_t k = 0; k < n; ++k) {
        out_sum[k] += part[t][k];
        if (out_sumsq) out_sumsq[k] += part2[t][k];
      }
  } else {
    // Interleave rows so threads finish together.
    std::atomic<int> next_row(0);
    for (int t = 0; t < n_threads; ++t) {
      threads.push_back(std::thread([&, t]() {
        Counters total;
        for (;;) {
          const int j = next_row.fetch_add(1);
          if (j >= prm->ny) break;
          Counters c;
          render_rows<P>(scene, cam, prm, j, j + 1, s0, s1, out_sum, out_sumsq, &c);
          total.add(c);
        }
        counters[t] = total;
      }));
    }
    for (size_t t = 0; t < threads.size(); ++t) threads[t].join();
  }
  if (stats) {
    Counters total;
    for (int t = 0; t < n_threads; ++t) total.add(counters[t]);
    std::memset(stats, 0, sizeof(*stats));
    stats->paths = total.paths; stats->rays = total.rays; stats->hits = total.hits; stats->node_tests = total.node_tests;
    stats->rect_tests = total.rect_tests; stats->sphere_tests = total.sphere_tests;
    stats->instance_ops = total.instance_ops; stats->shade_events = total.shade_events;
    stats->light_tests = total.light_tests;
  }
  return RT_OK;
}

template <class P>
int trace_impl(const rt_scene* scene, int traversal, const float* o, const float* d, size_t n, double t_min,
               double t_max, int32_t* prim_id, double* t, double* normal, double* p) {
  typedef typename P::Real R;
  World<P> world(scene);
  Sampler smp;
  std::memset(&smp, 0, sizeof(smp));
  smp.enabled = false;
  for (size_t i = 0; i < n; ++i) {
    const Ray<P> r(V3<R>(o[3 * i], o[3 * i + 1], o[3 * i + 2]), V3<R>(d[3 * i], d[3 * i + 1], d[3 * i + 2]));
    HitRecord<R> rec;
    const R tmax = std::isinf(t_max) ? std::numeric_limits<R>::infinity() : R(t_max);
    bool hit;
    if (traversal == ORC_TRAVERSE_LIST) hit = world.world_list_hit(r, R(t_min), tmax, smp, 0, rec);
    else hit = world.world_hit(r, R(t_min), tmax, smp, 0, rec);
    prim_id[i] = hit ? rec.prim : -1;
    t[i] = hit ? (double)rec.t : 0.0;
    if (normal) { normal[3 * i] = hit ? rec.normal.x : 0; normal[3 * i + 1] = hit ? rec.normal.y : 0; normal[3 * i + 2] = hit ? rec.normal.z : 0; }
    if (p) { p[3 * i] = hit ? rec.p.x : 0; p[3 * i + 1] = hit ? rec.p.y : 0; p[3 * i + 2] = hit ? rec.p.z : 0; }
  }
  return RT_OK;
}

}  // namespace

extern "C" {

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox4x32_10(ctr, key, out); }

int orc_trace_closest(const rt_scene* scene, int mode, int traversal, const float* o, const float* d, size_t n,
                      double t_min, double t_max, int32_t* prim_id, double* t, double* normal, double* p) {
  if (mode == ORC_F64) return trace_impl<PolicyF64>(scene, traversal, o, d, n, t_min, t_max, prim_id, t, normal, p);
  if (mode == ORC_F32) return trace_impl<PolicyF32>(scene, traversal, o, d, n, t_min, t_max, prim_id, t, normal, p);
  if (mode == ORC_F32_FUSED) return trace_impl<PolicyF32Fused>(scene, traversal, o, d, n, t_min, t_max, prim_id, t, normal, p);
  return RT_ERR_INVALID;
}

int orc_render(const rt_scene* scene, const rt_camera* cam, const rt_render_params* params, int mode, int n_threads,
               int split_samples, double* out_sum, double* out_sumsq, rt_stats* stats) {
  if (mode == ORC_F64) return render_impl<PolicyF64>(scene, cam, params, n_threads, split_samples, out_sum, out_sumsq, stats);
  if (mode == ORC_F32) return render_impl<PolicyF32>(scene, cam, params, n_threads, split_samples, out_sum, out_sumsq, stats);
  if (mode == ORC_F32_FUSED) return render_impl<PolicyF32Fused>(scene, cam, params, n_threads, split_samples, out_sum, out_sumsq, stats);
  return RT_ERR_INVALID;
}

// main.rs:136-155
void orc_tonemap(const double* sum_rgb, int32_t nx, int32_t ny, int32_t spp_total, uint8_t* rgb8) {
  const size_t n = (size_t)nx * ny * 3;
  for (size_t k = 0; k < n; ++k) {
    double c = sum_rgb[k] / (double)spp_total;
    c = std::sqrt(c);
    if (c > 1.0) c = 1.0;
    const double x = 255.99 * c;
    // `as u8`: saturating, NaN -> 0
    uint8_t b;
    if (!(x > 0.0)) b = 0;
    else if (x >= 255.0) b = 255;
    else b = (uint8_t)x;
    rgb8[k] = b;
  }
}

double orc_perlin_noise(const rt_perlin* perlin, const double p[3], int mode) {
  if (mode == ORC_F64) return World<PolicyF64>::perlin_noise(*perlin, V3<double>(p[0], p[1], p[2]));
  return World<PolicyF32>::perlin_noise(*perlin, V3<float>((float)p[0], (float)p[1], (float)p[2]));
}

double orc_perlin_turb(const rt_perlin* perlin, const double p[3], int depth, int mode) {
  if (mode == ORC_F64) return World<PolicyF64>::perlin_turb(*perlin, V3<double>(p[0], p[1], p[2]), depth);
  return World<PolicyF32>::perlin_turb(*perlin, V3<float>((float)p[0], (float)p[1], (float)p[2]), depth);
}

void orc_texture_value(const rt_scene* scene, int texture, double u, double v, const double p[3], int mode,
                       double out_rgb[3]) {
  if (mode == ORC_F64) {
    World<PolicyF64> w(scene);
    const V3<double> c = w.texture_value(texture, u, v, V3<double>(p[0], p[1], p[2]));
    out_rgb[0] = c.x; out_rgb[1] = c.y; out_rgb[2] = c.z;
  } else {
    World<PolicyF32> w(scene);
    const V3<float> c = w.texture_value(texture, (float)u, (float)v, V3<float>((float)p[0], (float)p[1], (float)p[2]));
    out_rgb[0] = c.x; out_rgb[1] = c.y; out_rgb[2] = c.z;
  }
}

void orc_camera_ray(const rt_camera* cam, double s, double t, const rt_render_params* params, int pixel, int sample,
                    int mode, double out_o[3], double out_d[3]) {
  Sampler smp;
  smp.key[0] = (uint32_t)params->seed;
  smp.key[1] = (uint32_t)(params->seed >> 32);
  smp.pixel = (uint32_t)pixel;
  smp.sample = (uint32_t)sample;
  smp.enabled = true;
  uint32_t w[4];
  smp.draw(0u, 0u, w);
  rt_scene empty;
  std::memset(&empty, 0, sizeof(empty));
  if (mode == ORC_F64) {
    World<PolicyF64> world(&empty);
    const Ray<PolicyF64> r = world.get_ray(*cam, s, t, smp);
    out_o[0] = r.a.x; out_o[1] = r.a.y; out_o[2] = r.a.z;
    out_d[0] = r.b.x; out_d[1] = r.b.y; out_d[2] = r.b.z;
  } else {
    World<PolicyF32> world(&empty);
    const Ray<PolicyF32> r = world.get_ray(*cam, (float)s, (float)t, smp);
    out_o[0] = r.a.x; out_o[1] = r.a.y; out_o[2] = r.a.z;
    out_d[0] = r.b.x; out_d[1] = r.b.y; out_d[2] = r.b.z;
  }
}

void orc_random_cosine_direction(double r1, double r2, double out[3]) {
  const V3<double> v = World<PolicyF64>::random_cosine_direction(r1, r2);
  out[0] = v.x; out[1] = v.y; out[2] = v.z;
}

void orc_light_pdf_value(const rt_light* light, const double o[3], const double* dirs, size_t n, double* out) {
  const World<PolicyF64> world(nullptr);
  for (size_t i = 0; i < n; ++i)
    out[i] = world.light_pdf_value(*light, V3<double>(o[0], o[1], o[2]), V3<double>(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]));
}

void orc_light_random(const rt_light* light, const double o[3], const double* xi, size_t n, double* out_dirs) {
  for (size_t i = 0; i < n; ++i) {
    const V3<double> d = World<PolicyF64>::light_random(*light, V3<double>(o[0], o[1], o[2]), xi[2 * i], xi[2 * i + 1]);
    out_dirs[3 * i] = d.x; out_dirs[3 * i + 1] = d.y; out_dirs[3 * i + 2] = d.z;
  }
}

void orc_onb_local(const double n[3], const double a[3], double out[3]) {
  const World<PolicyF64>::ONB o = World<PolicyF64>::ONB::build_from_w(V3<double>(n[0], n[1], n[2]));
  const V3<double> v = o.local(V3<double>(a[0], a[1], a[2]));
  out[0] = v.x; out[1] = v.y; out[2] = v.z;
}

}  // extern "C"
